"""The timed region again: four repeats (spread), the same K steps from an idle device, and -- same box, same images --
the reference's launch order (ScaleDown chain first) against the pyramid-in-detection sequence."""
import time


def run(R):
    args, capi, out, exs, d_imgs, K = R.args, R.capi, R.out, R.exs, R.d_imgs, R.args.steps
    elapsed, total_pix, run_pipelined, leg_guard = R.elapsed, R.total_pix, R.run_pipelined, R.leg_guard
    reps_in_order = [run_pipelined(d_imgs, K, warm=0) for _ in range(4)]
    reps = sorted(reps_in_order)
    allr = sorted(reps + [elapsed / K * 1e3])
    out["ms_per_step_spread"] = {"min": round(allr[0], 4), "median": round(allr[len(allr) // 2], 4),
                                 "max": round(allr[-1], 4), "regions": len(allr),
                                 "in_order": [round(elapsed / K * 1e3, 4)] + [round(r, 4) for r in reps_in_order],
                                 "note": "the timed region (`ms_per_step`) and 4 repeats of it, K steps each"}
    time.sleep(0.05)  # what a region costs that starts from an idle device (no pre-flight, no warm-up)
    idle_ms = run_pipelined(d_imgs, K, warm=0)
    out["ms_per_step_spread"]["ms_per_step_from_idle"] = round(idle_ms, 4)
    # beside `value`, at the top level: the same K steps started from an idle device, no pre-flight, no warm-up
    out["ms_per_step_from_idle"] = round(idle_ms, 4)
    out["value_from_idle_mpix_per_s"] = round(total_pix / idle_ms / 1e3, 1)
    # same box, same images, the reference's order (ScaleDown chain first, coarsest octave searched first):
    # what the pyramid-in-detection sequence is worth here
    if args.pyramid_in_detect == -1:
        with leg_guard("pyramid_policy_ab"):
            ab = {}
            for pol, name in ((0, "scale_down_chain_first"), (-1, "pyramid_in_detect (default)")):
                for x in exs:
                    x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, pol)
                r = sorted(run_pipelined(d_imgs, K) for _ in range(3))
                ab[name] = {"ms_per_step_median_of_3": round(r[1], 4), "Mpix_per_s": round(total_pix / r[1] / 1e3, 1)}
            out["pyramid_policy_ab"] = ab
        for x in exs:
            x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, -1)

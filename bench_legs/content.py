"""How much of the rate is the images: other content (`blobs`, un-pre-blurred `tile`), initBlur = 0 declared, ragged
widths -- each single stream AND pipelined like the timed region."""
from cusift_amd.batch import BatchExtractor


def run_content(R):
    args, torch, capi, synth, out, ex, d_imgs, K = R.args, R.torch, R.capi, R.synth, R.out, R.ex, R.d_imgs, R.args.steps
    B, w, h, E, local_kp, seeds, stage = R.B, R.w, R.h, R.E, R.local_kp, R.seeds, R.stage
    run_single_stream, run_pipelined, make_images = R.run_single_stream, R.run_pipelined, R.make_images
    cl = {}
    cl["tile_preblurred (the timed workload)"] = content_stats(torch, capi, ex, d_imgs, None, w, h, B, args, K)
    def pipelined_rate(d, imgs):
        ex.params.concurrent_batches = E
        ms = run_pipelined(imgs, max(8, K // 2))
        ex.params.concurrent_batches = 1
        d["ms_per_step_pipelined"] = round(ms, 4)
        d["Mpix_per_s_pipelined"] = round(B * w * h / (ms * 1e-3) / 1e6, 1)
        d["keypoints_per_s_pipelined"] = round(d["keypoints_per_step"] / (ms * 1e-3), 1)
        return d["Mpix_per_s_pipelined"]

    raw = ex.images_from_numpy(make_images(lambda s: synth.tile(s, w, h, 0.0), seeds))
    name = "tile_raw (SURVEY 8d primary generator as written: no pre-blur; initBlur=%.1f still declared)" % args.init_blur
    cl[name] = content_stats(torch, capi, ex, raw, run_single_stream, w, h, B, args, max(4, K // 2))
    out["value_tile_raw_mpix_per_s"] = pipelined_rate(cl[name], raw)
    del raw
    blob = ex.images_from_numpy(make_images(lambda s: synth.blobs(s, w, h), seeds))
    name = "blobs (SURVEY 8d secondary generator)"
    cl[name] = content_stats(torch, capi, ex, blob, run_single_stream, w, h, B, args, max(4, K // 2))
    out["value_blobs_mpix_per_s"] = pipelined_rate(cl[name], blob)
    del blob
    if stage is not None:
        cl["tile_preblurred (the timed workload)"].update(
            {"ms_per_step_single_stream": out["single_stream_leg"]["ms_per_step"],
             "keypoints_per_step": local_kp})
    out["content_legs"] = cl


def run_initblur0(R):
    args, torch, out, ex, d_imgs, K = R.args, R.torch, R.out, R.ex, R.d_imgs, R.args.steps
    B, w, h, E = R.B, R.w, R.h, R.E
    run_single_stream, run_pipelined = R.run_single_stream, R.run_pipelined
    saved_blur = ex.params.init_blur
    ex.params.init_blur = 0.0
    i_ms, i_st = run_single_stream(ex, d_imgs, max(4, K // 2))
    i_kp = int(ex.valid_counts().sum().item())
    raw_cnt = torch.clamp(ex.counts, min=0)
    ex.params.init_blur = saved_blur
    ex.params.concurrent_batches = E
    p_ms = run_pipelined(d_imgs, max(8, K // 2), init_blur=0.0)
    ex.params.concurrent_batches = 1
    n_steps = max(4, K // 2)
    out["initblur0_leg"] = {
        "workload": "the timed images, initBlur = 0.0 declared: all 8 levels of octave 0 are filtered (no "
                    "identity pass-through), the detector sees more and finer structure",
        "ms_per_step_single_stream": round(i_ms, 4), "ms_per_step_pipelined": round(p_ms, 4),
        "Mpix_per_s_pipelined": round(B * w * h / (p_ms * 1e-3) / 1e6, 1), "keypoints_per_step": i_kp,
        "keypoints_per_s_pipelined": round(i_kp / (p_ms * 1e-3), 1),
        "images_saturating_max_pts": int((raw_cnt >= ex.max_pts).sum().item()),
        "stage_ms_per_step": {k: round(i_st[k][0] / n_steps, 4) for k in ("scale_down", "detect_multi",
                                                                          "describe_all")}}
    out["value_initblur0_mpix_per_s"] = out["initblur0_leg"]["Mpix_per_s_pipelined"]


def run_ragged(R):
    args, synth, out, K, B, w, h, seeds, prm_kw = R.args, R.synth, R.out, R.args.steps, R.B, R.w, R.h, R.seeds, R.prm_kw
    run_single_stream, make_images = R.run_single_stream, R.make_images
    rw, rh = 1366, 768
    rex = BatchExtractor(B, rw, rh, **prm_kw)
    rimgs = rex.images_from_numpy(make_images(lambda s: synth.tile(s, rw, rh, args.init_blur), seeds))
    r_ms, r_st = run_single_stream(rex, rimgs, max(4, K // 2))
    rate = B * rw * rh / (r_ms * 1e-3) / 1e6
    leg = {"workload": "%d x %dx%d (octave widths 1366, 683, 341, 170, 85: none a multiple of 4)" % (B, rw, rh),
           "ms_per_step_single_stream": round(r_ms, 4), "Mpix_per_s_single_stream": round(rate, 1),
           "stage_ms_per_step": {k: round(r_st[k][0] / max(4, K // 2), 4)
                                 for k in ("scale_down", "detect_multi", "describe_all")},
           "detect_launches_fused": r_st["detect_multi"][1], "laplace_launches": r_st["laplace_multi"][1],
           "keypoints_per_step": int(rex.valid_counts().sum().item())}
    if "single_stream_leg" in out:
        base = B * w * h / (out["single_stream_leg"]["ms_per_step"] * 1e-3) / 1e6
        leg["per_pixel_rate_vs_1080p"] = round(rate / base, 3)
    out["ragged_width_leg"] = leg
    rex.close()
    del rimgs


def content_stats(torch, capi, ex, d_imgs, run_single_stream, w, h, B, args, steps):
    """Keypoints per step and the fraction of octave-0 wave-rows (240 columns x 1 row, the fused kernel's unit) in
    which no DoG centre of the 5 searchable scales exceeds the threshold -- the rows the pre-test skips (measured from
    the DoG planes of image 0 through the two-stage entry point); plus the single-stream rate on this content."""
    out = {}
    p = ex.pitch
    dog = torch.empty((7, h, p), dtype=torch.float32, device=d_imgs.device)
    ex.ctx.laplace_multi(d_imgs.data_ptr(), w, h, p, args.init_blur, dog.data_ptr())
    torch.cuda.synchronize()
    big = (dog[1:6, 1:h - 1, :w].abs() > args.thresh).any(dim=0)   # [h-2, w]: any scale above threshold, centre rows
    big[:, 0] = False  # border columns are never centres
    big[:, w - 1] = False
    strips = -(-w // 240)  # the kernel's strips: columns [240 s, 240 s + 240)
    pad = torch.zeros((big.shape[0], strips * 240), dtype=torch.bool, device=big.device)
    pad[:, :w] = big
    rows_with = pad.view(big.shape[0], strips, 240).any(dim=2)
    out["pretest_skip_frac_octave0"] = round(1.0 - float(rows_with.float().mean().item()), 4)
    out["pixels_above_thresh_frac_octave0"] = round(float(big.float().mean().item()), 5)
    if run_single_stream is not None:
        ms, st = run_single_stream(ex, d_imgs, steps)
        out["ms_per_step_single_stream"] = round(ms, 4)
        out["Mpix_per_s_single_stream"] = round(B * w * h / (ms * 1e-3) / 1e6, 1)
        out["keypoints_per_step"] = int(ex.valid_counts().sum().item())
        out["stage_ms_per_step"] = {k: round(st[k][0] / steps, 4) for k in ("scale_down", "detect_multi", "describe_all")}
        raw = torch.clamp(ex.counts, min=0)
        out["images_saturating_max_pts"] = int((raw >= ex.max_pts).sum().item())
    return out

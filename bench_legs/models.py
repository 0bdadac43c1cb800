"""Committed PREDICTIONS of the multi-GPU steps.  No multi-GPU node was reachable from the build box: the first measured
scaling curve (configs[3]) and the first tiled run on several devices (configs[4]) are to be read against these numbers.
Pure arithmetic, checked by tests/test_bench_model.py."""

XGMI_LINK_GBPS_PER_DIRECTION = 76.8  # one xGMI link of an MI355X: 153.6 GB/s bidirectional = 76.8 GB/s each way


def gather_model(kp_per_rank_step, rec_bytes, step_ms, lag_steps):
    """A PREDICTION of the all-gatherv step at 2 / 4 / 8 ranks, committed before any multi-GPU hardware has run it (no
    8-GPU node was reachable from the build box): the first measured scaling curve is to be read against these numbers.
    Exchange of one step: every rank sends its whole shard to each of its W - 1 peers, one dedicated point-to-point xGMI
    link per peer, all links at once (one ncclGroup of ncclSend / ncclRecv) -- so the time is one shard over one link,
    whatever W >= 2, and the same number of bytes arrives over the link's other direction."""
    shard = kp_per_rank_step * rec_bytes
    out = {"records_per_rank_per_step": int(kp_per_rank_step), "record_bytes": int(rec_bytes),
           "bytes_per_rank_per_step": int(shard), "bytes_per_peer_link_per_direction_per_step": int(shard),
           "assumed_link_GBps_per_direction": XGMI_LINK_GBPS_PER_DIRECTION,
           "assumed_rccl_p2p_efficiency": [1.0, 0.7],
           "extraction_ms_per_step": round(step_ms, 4), "finish_lags_begin_by_steps": lag_steps, "ranks": {}}
    for W in (2, 4, 8):
        row = {"bytes_received_per_rank_per_step": int(shard * (W - 1))}
        for eff in (1.0, 0.7):
            ex_ms = shard / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9 * eff) * 1e3
            row["eff_%.1f" % eff] = {
                "exchange_ms": round(ex_ms, 4),
                # own stream, finish lagging begin: latency is hidden, bandwidth is not -- a step cannot be shorter than
                # its exchange
                "ms_per_step_overlapped": round(max(step_ms, ex_ms), 4),
                "weak_scaling_efficiency_overlapped": round(step_ms / max(step_ms, ex_ms), 4),
                "ms_per_step_serial": round(step_ms + ex_ms, 4),
                "weak_scaling_efficiency_serial": round(step_ms / (step_ms + ex_ms), 4)}
        out["ranks"][str(W)] = row
    ex1 = shard / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3
    out["verdict"] = ("link-bound: one shard over one link takes %.2f ms at link peak against %.2f ms of extraction -- the "
                      "exchange, not the GPU, sets the step from 2 ranks up" % (ex1, step_ms)) if ex1 > step_ms else (
                      "extraction-bound at link peak (%.2f ms exchange against %.2f ms); link-bound below %.0f %% RCCL "
                      "efficiency" % (ex1, step_ms, 100.0 * ex1 / step_ms))
    out["not_modelled"] = ("the counts all-gather (a few tens of microseconds, hidden by the lag), the CUs RCCL's send / "
                           "receive kernels take from the extraction, HBM traffic of the arriving shards (%.2f GB per step "
                           "at 8 ranks: ~0.1 ms of HBM time)" % (shard * 7 / 1e9))
    out["options"] = {"compact 160-byte wire record (--gather-compact; 8-bit descriptor, lossy)":
                      round(kp_per_rank_step * 160 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3, 4),
                      "trimmed 540-byte wire record (the N > 1 default since round 5: the 135 floats extraction writes, EXACT, "
                      "expanded on arrival to 588-byte SiftPoint records; --gather-exact keeps 588 on the wire: %.4f ms)"
                      % (kp_per_rank_step * 588 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3):
                      round(kp_per_rank_step * 540 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3, 4),
                      "unit": "exchange ms per step at link peak"}
    return out


TILED_HALO_ROWS = 48  # cusift_amd.tiling.HALO: rows of neighbour data above / below a rank's rows in every tiled octave


def tiled_model(W, H, n_oct, whole_ms, keypoints, rec_bytes=540, halo=TILED_HALO_ROWS, launch_floor_ms=0.006):
    """A PREDICTION of BASELINE configs[4] -- one W x H image strip-tiled over P ranks with a halo exchange per octave and
    an all-gatherv of the merged SiftData (cusift_tiled_extract + cusift_allgatherv, cusift_amd/csrc/sift_tiled.hip) --
    committed before any multi-GPU hardware has run it.  `whole_ms` and `keypoints` are MEASURED on one GPU (the same image
    through the whole-image driver); everything else is arithmetic on the plan:
      per tiled octave o (every rank owns >= `halo` rows of it): `halo` rows x w_o x 4 bytes to each neighbour and as many
        back, one grouped ncclSend/ncclRecv per octave -- a chain: ScaleDown of octave o waits for octave o - 1's rows, the
        exchange for the ScaleDown, so the per-exchange latency is paid n_tiled times, not hidden;
      the first octave some rank owns fewer than `halo` rows of collapses onto rank 0 (one more exchange, the owned rows);
      kernels: the whole image's time / P x 1.15 (bands overlap by the blur's rows; the busiest strip sets the time) plus a
        floor per launch (n_tiled - 1 ScaleDowns, detection, description, pack, counts, expand);
      merge: one counts all-gather + every rank's records over its own link to every peer at once (one shard over one link).
    Two corners per P: link peak with 20 us per exchange, and 70 % of link peak with 60 us."""
    dims = [(W, H)]
    for _ in range(1, n_oct):
        dims.append((dims[-1][0] // 2, dims[-1][1] // 2))
    out = {"image": "%dx%d, %d octaves" % (W, H, n_oct), "halo_rows": halo, "record_bytes": rec_bytes,
           "measured_one_gpu_whole_image_ms": round(whole_ms, 4), "measured_keypoints": int(keypoints),
           "assumed_link_GBps_per_direction": XGMI_LINK_GBPS_PER_DIRECTION,
           "assumed_corners": [{"exchange_latency_us": 20, "rccl_p2p_efficiency": 1.0},
                               {"exchange_latency_us": 60, "rccl_p2p_efficiency": 0.7}],
           "assumed_launch_floor_ms": launch_floor_ms, "assumed_strip_imbalance": 1.15, "ranks": {}}
    for P in (2, 4, 8):
        own0 = H // P
        collapse = n_oct
        for o in range(n_oct):
            if (own0 >> o) < halo or dims[o][0] < 4 or dims[o][1] < 3:
                collapse = o
                break
        n_tiled = collapse
        halo_bytes = [halo * dims[o][0] * 4 for o in range(n_tiled)]
        collapse_bytes = (own0 >> collapse) * dims[collapse][0] * 4 if collapse < n_oct else 0
        launches = max(0, n_tiled - 1) + 5
        kernel_ms = whole_ms / P * 1.15 + launches * launch_floor_ms
        shard = keypoints / P * rec_bytes
        row = {"tiled_octaves": n_tiled, "collapse_octave": collapse if collapse < n_oct else None,
               "halo_bytes_per_neighbour_per_direction_by_octave": halo_bytes,
               "halo_bytes_per_neighbour_per_direction_total": int(sum(halo_bytes)),
               "exchanges_in_the_chain": n_tiled + (1 if collapse_bytes else 0),
               "collapse_bytes_per_rank": int(collapse_bytes), "merge_bytes_per_rank": int(shard),
               "kernel_ms_per_rank": round(kernel_ms, 4)}
        for lat_us, eff in ((20, 1.0), (60, 0.7)):
            bw = XGMI_LINK_GBPS_PER_DIRECTION * 1e9 * eff
            lat = lat_us * 1e-3
            halo_ms = sum(lat + b / bw * 1e3 for b in halo_bytes)
            coll_ms = (lat + collapse_bytes * (P - 1) / bw * 1e3) if collapse_bytes else 0.0  # P - 1 senders, one receiver
            merge_ms = lat + shard / bw * 1e3
            total = kernel_ms + halo_ms + coll_ms + merge_ms
            row["latency_%dus_eff_%.1f" % (lat_us, eff)] = {
                "halo_exchange_ms": round(halo_ms, 4), "collapse_ms": round(coll_ms, 4), "merge_ms": round(merge_ms, 4),
                "predicted_ms_per_image": round(total, 4), "predicted_speedup_over_one_gpu": round(whole_ms / total, 3),
                "predicted_Mpix_per_s": round(W * H / total / 1e3, 1)}
        out["ranks"][str(P)] = row
    best = out["ranks"]["8"]["latency_20us_eff_1.0"]["predicted_speedup_over_one_gpu"]
    worst = out["ranks"]["8"]["latency_60us_eff_0.7"]["predicted_speedup_over_one_gpu"]
    out["verdict"] = ("latency-bound: at 8 ranks the image's kernels shrink to %.2f ms per rank while the chain of %d "
                      "exchanges and the merge cost %.2f-%.2f ms -- predicted speed-up over one GPU %.2fx-%.2fx, NOT 8x; a "
                      "first hardware run inside that range is as designed, below it is a defect"
                      % (out["ranks"]["8"]["kernel_ms_per_rank"], out["ranks"]["8"]["exchanges_in_the_chain"] + 1,
                         out["ranks"]["8"]["latency_20us_eff_1.0"]["predicted_ms_per_image"] - out["ranks"]["8"]["kernel_ms_per_rank"],
                         out["ranks"]["8"]["latency_60us_eff_0.7"]["predicted_ms_per_image"] - out["ranks"]["8"]["kernel_ms_per_rank"],
                         worst, best))
    out["not_modelled"] = ("host enqueue time of the per-octave calls (hidden while the device works), the CUs RCCL's "
                           "send / receive kernels occupy, HBM traffic of the arriving records")
    return out

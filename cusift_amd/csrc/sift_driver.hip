// sift_driver.hip -- the octave driver of the C ABI (include/cusift_amd.h): cusift_extract_batch (SiftData::Extract +
// ExtractSiftLoop, cuSIFT.cu:61-120,175-270, for a batch), its recorded hipGraph and the blocking single-image entry points.
#include "sift_host.h"

// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
// Bytes of DoG planes the two-stage path needs: the largest searched octave that does not take the fused detection
// (0 when every octave does).  `arena_base`: where octaves >= 1 live (their alignment is what matters).
size_t two_stage_dog_bytes(const cusift_ctx *ctx, const Plan &pl, const cusift_params *prm, const float *d_imgs,
                                  size_t image_stride, const char *arena_base, int n_images) {
  size_t need = 0;
  const bool generic = ctx->knobs.force_generic;
  for (int o = 0; o < pl.n_oct; ++o) {
    if (!(prm->lowest_scale < pl.sub[o] * 2.0f)) continue;
    const float *b = o == 0 ? d_imgs : (const float *)(arena_base + pl.base_off[o]);
    const size_t st = o == 0 ? image_stride : (size_t)pl.h[o] * pl.p[o];
    if (!prm->fused_detect || generic || !detect_fused_ok(b, pl.w[o], pl.h[o], pl.p[o], st))
      need = std::max(need, (size_t)n_images * kNumDog * pl.h[o] * pl.p[o] * sizeof(float));
  }
  return need;
}

extern "C" int cusift_extract_batch(cusift_ctx *ctx, const float *d_imgs, int n_images, int w, int h, int pitch,
                                    size_t image_stride, const cusift_params *prm, cusift_point *d_points,
                                    unsigned int *d_counters) {
  TRY(enter(ctx));
  if (!d_imgs || !d_points || !d_counters) return fail(CUSIFT_ERR_INVALID, "extract: missing data");
  if (n_images > 1 && image_stride < (size_t)h * pitch) return fail(CUSIFT_ERR_INVALID, "image_stride too small");
  Plan pl;
  TRY(make_plan(pl, n_images, w, h, pitch, prm, wants_side_stream(ctx, prm, n_images, w, h),
                wants_stage_all(ctx, prm, n_images, w, h), stage_all_limit(ctx)));
  TRY(ensure_arena(ctx, pl.total));
  if (const size_t dog_need = two_stage_dog_bytes(ctx, pl, prm, d_imgs, image_stride, ctx->arena, n_images))
    TRY(ensure_dog(ctx, dog_need));  // sized once for the largest two-stage octave, before anything is enqueued

  StageTimer total(ctx, CUSIFT_STAGE_TOTAL);

  const float *base[kMaxOctaves];
  size_t stride[kMaxOctaves];
  base[0] = d_imgs;
  stride[0] = image_stride;
  for (int o = 1; o < pl.n_oct; ++o) {
    base[o] = (const float *)(ctx->arena + pl.base_off[o]);
    stride[o] = (size_t)pl.h[o] * pl.p[o];
  }
  unsigned int *first = (unsigned int *)(ctx->arena + pl.first_off);
  // With fused_detect the keypoint stages run once, after the last octave's detection, over the flattened list
  // of all keypoints of the batch (describe_all_kernel); otherwise per octave like the reference.  The DETECTION
  // kernel is chosen per octave: the fused one wherever it applies (16-byte aligned rows, w >= 4, h >= 3), the
  // two-stage pair for an octave where it does not (a 2x1 coarsest octave, a caller's odd pitch) -- one such octave
  // no longer demotes the others.
  const bool generic = ctx->knobs.force_generic;
  const bool flat = prm->fused_detect && n_images <= kMaxFlatImages && !generic;
  auto searched = [&](int o) { return prm->lowest_scale < pl.sub[o] * 2.0f; };  // cuSIFT.cu:194
  auto fused_ok = [&](int o) { return detect_fused_ok(base[o], pl.w[o], pl.h[o], pl.p[o], stride[o]); };

  // Where the octaves' keypoints go (sift_types.h: SegmentTable).  One stream searching coarsest first leaves SiftData
  // in list order by itself; the moment two detections may overlap -- octave 0 on the side stream, the coarser octaves
  // in one launch -- they append to lists of their own (record heads in the arena) and describe_all_kernel joins them.
  //   stage_all   every searched octave to its own list: needs the fused kernel for every searched octave
  //   forked      octave 0 to a list of its own and to the side stream; the coarser ones in place (or staged too)
  bool stage_all = flat && pl.staged_octaves == pl.n_oct;
  bool any_coarser = false;
  for (int o = 0; o < pl.n_oct; ++o) {
    if (!searched(o)) continue;
    stage_all = stage_all && fused_ok(o);
    any_coarser = any_coarser || o > 0;
  }
  bool forked = flat && pl.fork && pl.staged_octaves >= 1 && searched(0) && fused_ok(0) && any_coarser;
  if (forked && ensure_side_stream(ctx) != CUSIFT_OK) forked = false;  // no stream runs beside this one: one stream
  const size_t list_bytes = (size_t)n_images * prm->max_pts * kStagedRecBytes;
  char *const lists = ctx->arena + pl.staged_off;
  unsigned int *const seg_counts = first;  // [octave][image]: `first` is free when the keypoint stages run once
  unsigned int *const seg_end = (unsigned int *)(ctx->arena + pl.seg_end_off);
  auto list_of = [&](int o) { return reinterpret_cast<cusift_point *>(lists + (size_t)o * list_bytes); };
  SegmentTable G;
  memset(&G, 0, sizeof(G));
  if (stage_all) {
    G.n_seg = pl.n_oct;
    for (int r = 0; r < pl.n_oct; ++r) {  // list order: coarsest octave first
      const int o = pl.n_oct - 1 - r;
      G.base[r] = reinterpret_cast<const char *>(list_of(o));
      G.count[r] = seg_counts + (size_t)o * n_images;
    }
  } else if (forked) {
    G.n_seg = 2;
    G.base[0] = nullptr;  // the coarser octaves: in place, the caller's counter
    G.count[0] = d_counters;
    G.base[1] = reinterpret_cast<const char *>(list_of(0));
    G.count[1] = seg_counts;
  }
  // The pyramid as a by-product of the detection (CUSIFT_POLICY_PYRAMID_IN_DETECT): octaves [0, chain_end) are searched
  // finest first by detections that also write the next octave's image -- images 1 .. chain_end come from there, not
  // from ScaleDown launches.  Needs a list per octave (the detections no longer run in list order) and octave 0 on the
  // context's own stream (octave 1 waits for it anyway).
  int chain_end = 0;
  if (stage_all && !forked) {
    const int mode = wants_pyramid_in_detect(ctx, prm, n_images, w, h);
    while (chain_end < pl.n_oct - 1 && (mode == 2 || (mode == 1 && chain_end == 0)) && searched(chain_end)) ++chain_end;
  }
  // A small call's dispatches are most of its time, so its housekeeping rides along: the ScaleDown chain in one launch
  // (which also clears the lists' counters), all octaves in one detection launch (which also clears describe_all's work
  // cursors), and describe_all_kernel joins the lists itself -- pyramid, detection, description: three dispatches.
  const bool small_pyramid = pl.n_oct >= 2 && chain_end == 0 && wants_small_pyramid(ctx, n_images, w, h);
  const int first_rest = std::max(forked ? 1 : 0, chain_end);  // the octaves from here on: the ScaleDown chain, then searched
  int n_one_launch = 0;  // octaves the one detection launch would take
  if (stage_all && !ctx->knobs.no_multi)
    for (int o = first_rest; o < pl.n_oct && n_one_launch < kMaxMultiOctaves; ++o) n_one_launch += searched(o) ? 1 : 0;
  const bool one_launch = n_one_launch >= 2;
  // no join_counts_kernel: describe_all_kernel joins (small calls: wants_self_join)
  const bool self_join = stage_all && one_launch && wants_self_join(ctx, n_images, w, h);
  const bool pyramid_clears = small_pyramid && stage_all && !forked;  // (a forked octave 0 may count before the pyramid runs)
  // cuSIFT.cu:69: point counter = 0 (with every octave staged the join writes it instead)
  if (!stage_all) HIP_TRY(hipMemsetAsync(d_counters, 0, sizeof(unsigned int) * n_images, ctx->stream));
  const size_t n_seg_counts = (size_t)n_images * (stage_all ? pl.n_oct : 1);
  // join_counts_kernel reads every list's counter exactly once and leaves it zero (join_clears): when the previous
  // extraction of this context did that for these very counters, nothing has to be cleared now -- in a loop of equal
  // batches no memset is dispatched at all.  Not relied upon inside a recording (a replay cannot know what ran before it).
  const bool join_clears = stage_all && !self_join && !forked;
  const size_t seg_bytes = std::min(align_up_sz(sizeof(unsigned int) * n_seg_counts, 64),
                                    sizeof(unsigned int) * n_images * kMaxOctaves);
  const bool seg_clean = !ctx->recording && ctx->seg_clean_ptr == (const void *)seg_counts && ctx->seg_clean_bytes >= seg_bytes;
  ctx->seg_clean_ptr = nullptr;  // whatever follows writes the arena; set again once the clearing join is enqueued
  if (G.n_seg && !pyramid_clears && !seg_clean) {
    // (a multiple of 64 bytes: the runtime fills an odd size with two dispatches; the region is kMaxOctaves x n_images)
    HIP_TRY(hipMemsetAsync(seg_counts, 0, seg_bytes, ctx->stream));
  }

  if (forked) {
    ctx->forks++;
    HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    const int rc = detect_impl(ctx, d_imgs, w, h, pitch, image_stride, (float)pl.blur[0], prm->peak_thresh,
                               prm->edge_thresh, pl.sub[0], list_of(0), prm->max_pts, seg_counts, n_images,
                               RowWindow{0, h}, 0, h, 1, true, true);
    const hipError_t e = hipEventRecord(ctx->ev_join, ctx->side);
    if (rc != CUSIFT_OK || e != hipSuccess) {
      (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);  // never leave the side stream forked (a capture would not end)
      if (rc != CUSIFT_OK) return rc;
      HIP_TRY(e);
    }
  }
  // the side stream rejoins the context's stream however the work in between ends
  // CUSIFT_LAB only (CUSIFT_UNORDERED_COARSE; round 6): what would an octave hand-over INSIDE one launch be
  // worth to a lone caller at best?  Everything behind octave 0's detection -- the short ScaleDown chain and the one
  // launch that searches octaves 1.. -- is put on the side stream WITHOUT waiting for octave 0's detection: it reads the
  // octave-1 image the PREVIOUS call left in the arena (the same pixels when the same batch is extracted in a loop, as the
  // benchmark does -- a timing experiment, not an extraction), so the coarser octaves' waves fill octave 0's tail with
  // perfect overlap and no synchronisation cost at all.  profiles/r06/lone_caller_handover.md holds the result.
  bool unordered = false;
#ifdef CUSIFT_LAB
  unordered = ctx->knobs.unordered_coarse && !forked && chain_end == 1 && !ctx->timing && !ctx->recording &&
              ensure_side_stream(ctx) == CUSIFT_OK;
#endif
  hipStream_t const main_stream = ctx->stream;
  auto on_main = [&]() -> int {
    // ExtractSiftLoop, cuSIFT.cu:175-192: build the pyramid finest -> coarsest -- a small call's first levels in one launch
    int built = 0;
    if (unordered) {  // the side stream starts behind whatever this call has enqueued so far (the counters' memset)
      HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
      HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    }
    for (int o = 0; o < chain_end; ++o) {  // finest first: octave o's detection writes octave o + 1
      DownOut dn;
      dn.dst = const_cast<float *>(base[o + 1]);
      dn.pitch = pl.p[o + 1];
      dn.stride = (long)stride[o + 1];
      scale_down_taps(dn.T, 0.5f);  // cuSIFT.cu:185
      TRY(detect_impl(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], prm->peak_thresh,
                      prm->edge_thresh, pl.sub[o], list_of(o), prm->max_pts, seg_counts + (size_t)o * n_images, n_images,
                      RowWindow{0, pl.h[o]}, 0, pl.h[o], prm->concurrent_batches, true, false, &dn));
      built = o + 1;
    }
    if (unordered) ctx->stream = ctx->side;  // (the rest of on_main enqueues there; put back below)
    if (small_pyramid) {
      built = std::min(pl.n_oct - 1, kMaxPyramidLevels);
      TRY(pyramid_small_impl(ctx, base, pl.w, pl.h, pl.p, stride, built, n_images, 0.5f,
                             pyramid_clears ? seg_counts : nullptr, pyramid_clears ? (int)n_seg_counts : 0));
    }
    for (int o = built + 1; o < pl.n_oct; ++o)
      TRY(cusift_scale_down(ctx, const_cast<float *>(base[o]), pl.p[o], stride[o], base[o - 1], pl.w[o - 1], pl.h[o - 1],
                            pl.p[o - 1], stride[o - 1], n_images, 0.5f));  // cuSIFT.cu:185
    // With a list per octave all octaves (but a forked octave 0) are searched by ONE launch, largest first
    bool in_one_launch[kMaxOctaves] = {false};
    if (one_launch) {
      MultiOctave mo[kMaxMultiOctaves];
      int n_mo = 0;
      for (int o = first_rest; o < pl.n_oct && n_mo < kMaxMultiOctaves; ++o)
        if (searched(o)) {
          mo[n_mo++] = MultiOctave{base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], pl.sub[o], list_of(o),
                                   seg_counts + (size_t)o * n_images};
          in_one_launch[o] = true;
        }
      TRY(detect_multi_impl(ctx, mo, n_mo, prm->peak_thresh, prm->edge_thresh, prm->max_pts, n_images,
                            forked ? 1 : prm->concurrent_batches, ctx->d_queue));
    }
    // ... and search it coarsest first (the recursion unwinds: cuSIFT.cu:190-196)
    for (int o = pl.n_oct - 1; o >= first_rest; --o) {
      if (!searched(o) || in_one_launch[o]) continue;
      // ExtractSiftOctave, cuSIFT.cu:204-270
      unsigned int *fst = first + (size_t)o * n_images;  // cuSIFT.cu:243 (fstPts), kept on the device
      if (!flat)
        HIP_TRY(hipMemcpyAsync(fst, d_counters, sizeof(unsigned int) * n_images, hipMemcpyDeviceToDevice, ctx->stream));
      if (prm->fused_detect && !generic && fused_ok(o)) {
        TRY(detect_impl(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], prm->peak_thresh,
                        prm->edge_thresh, pl.sub[o], stage_all ? list_of(o) : d_points, prm->max_pts,
                        stage_all ? seg_counts + (size_t)o * n_images : d_counters, n_images, RowWindow{0, pl.h[o]}, 0,
                        pl.h[o], forked ? 1 : prm->concurrent_batches, stage_all));
      } else {
        const size_t dstride = (size_t)kNumDog * pl.h[o] * pl.p[o];
        float *dog = ctx->dog;
        TRY(cusift_laplace_multi(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], (float)pl.blur[o], dog, dstride,
                                 n_images));
        TRY(cusift_find_points_multi(ctx, dog, pl.w[o], pl.h[o], pl.p[o], dstride, prm->peak_thresh, prm->edge_thresh,
                                     pl.sub[o], d_points, prm->max_pts, d_counters, n_images));
      }
      if (flat) continue;
      TRY(cusift_compute_orientations(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], d_points, prm->max_pts, fst,
                                      d_counters, prm->tex_frac_bits, n_images));
      TRY(descriptors_impl(ctx, base[o], pl.w[o], pl.h[o], pl.p[o], stride[o], d_points, prm->max_pts, fst, d_counters,
                           pl.sub[o], prm->tex_frac_bits, n_images, RowWindow{0, pl.h[o]}, prm->root_sift));
    }
    return CUSIFT_OK;
  };
  const int rc_main = on_main();
  if (unordered) {
    ctx->stream = main_stream;
    (void)hipEventRecord(ctx->ev_join, ctx->side);
    (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
  }
  if (forked) {
    const hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    if (rc_main == CUSIFT_OK) HIP_TRY(e);
  }
  if (rc_main != CUSIFT_OK) return rc_main;
  if (flat) {
    OctaveTable T;
    memset(&T, 0, sizeof(T));
    T.n_oct = pl.n_oct;
    for (int o = 0; o < pl.n_oct; ++o) {
      T.base[o] = base[o];
      T.stride[o] = (long)stride[o];
      T.w[o] = pl.w[o];
      T.h[o] = pl.h[o];
      T.pitch[o] = pl.p[o];
      T.sub[o] = pl.sub[o];
    }
    float q, inv_q;
    frac_consts(prm->tex_frac_bits, q, inv_q);
    // persistent grid = exactly the blocks that are resident at once (a larger static grid would run in
    // rounds and leave the second round's items waiting); items are interleaved over the blocks
    if (ctx->describe_grid == 0) {
      int per_cu = 0, cus = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, describe_all_kernel, 64, 0));
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
      ctx->describe_grid = std::max(1, per_cu) * std::max(1, cus);
    }
    const long cap = (long)n_images * prm->max_pts;
    // a multiple of the shard count (the kernel deals items to shards by workgroup index)
    const long want = std::max(1L, std::min(cap, (long)ctx->describe_grid));
    dim3 grid((unsigned int)std::max<long>(kQueueShards, want / kQueueShards * kQueueShards));
    unsigned int *queue = ctx->d_queue;  // the kernel's work cursors, zero at launch
    if (self_join) {
      // (the detection launch cleared the cursors; describe_all_kernel joins the lists itself)
    } else if (G.n_seg) {
      hipLaunchKernelGGL(join_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, d_counters, G, seg_end, n_images,
                         prm->max_pts, queue, join_clears ? 1 : 0);
      TRY(check_launch("join_counts"));
      if (join_clears && !ctx->recording) {  // (a recording enqueues nothing: the counters are as they were)
        ctx->seg_clean_ptr = seg_counts;
        ctx->seg_clean_bytes = seg_bytes;
      }
    } else {
      HIP_TRY(hipMemsetAsync(queue, 0, kQueueShards * 128, ctx->stream));
    }
    StageTimer t(ctx, CUSIFT_STAGE_DESCRIBE_ALL);
    hipLaunchKernelGGL(describe_all_kernel, grid, dim3(64), 0, ctx->stream, T, d_points, prm->max_pts, d_counters,
                       n_images, q, inv_q, prm->root_sift, queue, G,
                       self_join ? (const unsigned int *)nullptr : (const unsigned int *)seg_end);
    TRY(check_launch("describe_all"));
  }
  return CUSIFT_OK;
}

// ------------------------------------------------------------------------------------------------
// replayable extraction: the launch sequence of cusift_extract_batch recorded once as a hipGraph
// ------------------------------------------------------------------------------------------------
struct cusift_graph {
  cusift_ctx *ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  unsigned long scratch_gen = 0;  // the recording refers to the context's scratch (arena, DoG planes of the
                                  // two-stage path) as it was: any later re-allocation invalidates it
  int nodes = 0;
};

extern "C" int cusift_graph_create(cusift_ctx *ctx, cusift_graph **out, const float *d_imgs, int n_images, int w, int h,
                                   int pitch, size_t image_stride, const cusift_params *prm, cusift_point *d_points,
                                   unsigned int *d_counters) {
  if (!ctx || !out) return fail(CUSIFT_ERR_INVALID, "NULL argument");
  TRY(enter(ctx));
  *out = nullptr;
  if (!ctx->stream) return fail(CUSIFT_ERR_INVALID, "graph capture needs a real stream (the context borrows the null stream)");
  // For the length of this call the context is "recording": no stage-timer events (they are not part of a recording)
  // and the one-stream launch sequence (wants_side_stream) -- the plan below is the one cusift_extract_batch will make.
  struct Recording {
    cusift_ctx *c;
    bool timing;
    explicit Recording(cusift_ctx *ctx) : c(ctx), timing(ctx->timing) { c->timing = false; c->recording = true; }
    ~Recording() { c->timing = timing; c->recording = false; }
  } recording(ctx);
  Plan pl;
  // the side stream is found (and probed: that waits) before the capture starts; the fork and the join become edges
  const bool fork = wants_side_stream(ctx, prm, n_images, w, h) && ensure_side_stream(ctx) == CUSIFT_OK;
  TRY(make_plan(pl, n_images, w, h, pitch, prm, fork, wants_stage_all(ctx, prm, n_images, w, h), stage_all_limit(ctx)));
  // everything that allocates or synchronises happens before the capture starts
  TRY(ensure_arena(ctx, pl.total));
  // the DoG planes of every octave that takes the two-stage path (see cusift_extract_batch), sized up front
  const size_t dog_need = two_stage_dog_bytes(ctx, pl, prm, d_imgs, image_stride, ctx->arena, n_images);
  if (dog_need) TRY(ensure_dog(ctx, dog_need));
  if (ctx->describe_grid == 0) {
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, describe_all_kernel, 64, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->describe_grid = std::max(1, per_cu) * std::max(1, cus);
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  cusift_graph *g = new cusift_graph();
  g->ctx = ctx;
  hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    delete g;
    return fail(CUSIFT_ERR_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(e));
  }
  const int rc = cusift_extract_batch(ctx, d_imgs, n_images, w, h, pitch, image_stride, prm, d_points, d_counters);
  e = hipStreamEndCapture(ctx->stream, &g->graph);
  if (rc != CUSIFT_OK || e != hipSuccess || !g->graph) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    if (rc != CUSIFT_OK) return rc;
    return fail(CUSIFT_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g->graph);
    delete g;
    return fail(CUSIFT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  size_t n_nodes = 0;
  (void)hipGraphGetNodes(g->graph, nullptr, &n_nodes);
  g->nodes = (int)n_nodes;
  g->scratch_gen = ctx->scratch_gen;
  *out = g;
  return CUSIFT_OK;
}

extern "C" int cusift_graph_launch(cusift_graph *g) {
  if (!g || !g->exec) return fail(CUSIFT_ERR_INVALID, "graph is NULL");
  if (g->ctx->scratch_gen != g->scratch_gen)
    return fail(CUSIFT_ERR_INVALID,
                "the context's scratch (arena / DoG planes) was re-allocated after this graph was recorded; record it again");
  HIP_TRY(hipSetDevice(g->ctx->device));
  g->ctx->seg_clean_ptr = nullptr;  // the recording may leave its lists' counters in any state
  HIP_TRY(hipGraphLaunch(g->exec, g->ctx->stream));
  return CUSIFT_OK;
}

extern "C" int cusift_graph_nodes(cusift_graph *g) { return g ? g->nodes : 0; }

extern "C" int cusift_graph_destroy(cusift_graph *g) {
  if (!g) return CUSIFT_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return CUSIFT_OK;
}

extern "C" int cusift_extract(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, const cusift_params *prm,
                              cusift_point *d_points, cusift_point *h_points, int *num_pts) {
  TRY(enter(ctx));
  if (!num_pts) return fail(CUSIFT_ERR_INVALID, "num_pts is NULL");
  *num_pts = 0;
  TRY(cusift_extract_batch(ctx, d_img, 1, w, h, pitch, (size_t)h * pitch, prm, d_points, ctx->d_counter1));
  // The count travels to a pinned word of the context (a pageable destination is a staged copy under a runtime-wide
  // lock: callers on several threads serialise on it), and the records the caller most likely wants travel WITH it: as
  // many as the context's previous call returned plus an eighth, copied before the count is known -- one wait for the
  // device instead of two (the reference's order, cuSIFT.cu:107-114: count, then Synchronize()).  Whatever the guess
  // missed follows in a second copy.  Rows at and beyond numPts of the caller's buffer are unspecified, as after the
  // reference's malloc (cuSIFT.cu:24); never beyond its max_pts rows.
  HIP_TRY(hipMemcpyAsync(ctx->h_counter1, ctx->d_counter1, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
  const int guess = h_points ? (ctx->extract_guess < prm->max_pts ? ctx->extract_guess : prm->max_pts) : 0;
  if (guess > 0)
    HIP_TRY(hipMemcpyAsync(h_points, d_points, sizeof(cusift_point) * (size_t)guess, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const unsigned int cnt = *(volatile unsigned int *)ctx->h_counter1;
  // cuSIFT.cu:107-110
  const int n = cnt < (unsigned int)prm->max_pts ? (int)cnt : prm->max_pts;
  *num_pts = n;
  if (h_points && n > guess) {  // SiftData::Synchronize, cuSIFT.cu:52-59
    HIP_TRY(hipMemcpyAsync(h_points + guess, d_points + guess, sizeof(cusift_point) * (size_t)(n - guess),
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (h_points) ctx->extract_guess = n + n / 8 + 16;
  return CUSIFT_OK;
}

extern "C" int cusift_extract_host(cusift_ctx *ctx, const float *h_img, int w, int h, const cusift_params *prm,
                                   cusift_point *d_points, cusift_point *h_points, int *num_pts) {
  TRY(enter(ctx));
  if (!h_img) return fail(CUSIFT_ERR_INVALID, "image is NULL");
  if (w < 1 || h < 1) return fail(CUSIFT_ERR_INVALID, "bad image size %dx%d", w, h);
  const int pitch = ialign_up(w, 128);  // cuImage::AllocateWithHostMemory, cuImage.cu:11-13
  Plan pl;
  TRY(make_plan(pl, 1, w, h, pitch, prm, wants_side_stream(ctx, prm, 1, w, h),
                wants_stage_all(ctx, prm, 1, w, h), stage_all_limit(ctx)));  // the plan cusift_extract_batch will make
  const size_t img_bytes = align_up_sz((size_t)h * pitch * sizeof(float), 256);
  TRY(ensure_arena(ctx, pl.total + img_bytes));
  float *d_img = (float *)(ctx->arena + pl.total);
  HIP_TRY(hipMemcpy2DAsync(d_img, sizeof(float) * pitch, h_img, sizeof(float) * w, sizeof(float) * w, h,
                           hipMemcpyHostToDevice, ctx->stream));
  return cusift_extract(ctx, d_img, w, h, pitch, prm, d_points, h_points, num_pts);
}


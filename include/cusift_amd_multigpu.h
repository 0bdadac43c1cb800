/*
 * cusift_amd_multigpu.h -- more than one GPU: SiftData on the wire, the communicator over RCCL, all-gatherv, halo exchange, strip tiling.
 * Part of the C ABI of libcusift_amd.so; conventions and the map of the four headers: cusift_amd.h.
 */
#ifndef CUSIFT_AMD_MULTIGPU_H
#define CUSIFT_AMD_MULTIGPU_H

#include "cusift_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Packs a batch's SiftData for an exchange (all-gatherv over RCCL): the valid records of all images back to back in
 * image order into d_packed (room for `capacity` records; anything beyond is dropped) and the exclusive prefix sums
 * of the valid counts into d_offsets[0 .. n_images] (may be NULL).  n_images <= 256.  Asynchronous. */
int cusift_pack_points(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters, int n_images,
                       int max_pts, cusift_point *d_packed, size_t capacity, unsigned int *d_offsets);

/* Compact wire format for SiftData that has to cross PCIe or a network (new; optional -- the exact 588-byte records
 * stay the default everywhere).  Extraction writes 7 header fields + the 128-float descriptor of a record; the other 12
 * floats are left as they were (SURVEY: cuSIFT.cu:24,29).  A compact record carries the 7 fields EXACTLY and the
 * descriptor as 128 bytes with one quantisation step per record: data[i] ~= q[i] * desc_step, desc_step =
 * max(data) / 255, q[i] = min(255, floor(data[i] / desc_step + 0.5)) -- 160 B instead of 588, |error| <= desc_step / 2
 * per element (<= 1e-3 for a SIFT descriptor, whose elements are <= ~0.5: an L2 distance of a few 1e-3, NOT within the
 * 1e-4 parity bar, which is why this is a wire format and not the SiftData).  A descriptor without a finite positive
 * maximum (flat patch: NaN) travels as desc_step = that maximum (NaN or 0) and q = 0.
 * cusift_pack_points_compact: as cusift_pack_points, compacting on the way (d_packed holds `capacity` compact records).
 * cusift_expand_points_host: compact records -> SiftPoint records on the host (data[i] = q[i] * desc_step; a NaN step
 *   gives the NaN descriptor back; the 12 fields extraction never writes are zeroed). */
typedef struct cusift_compact_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation;
  float subsampling;
  float desc_step;
  unsigned char q[128];
} cusift_compact_point; /* 160 bytes */
int cusift_pack_points_compact(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                               int n_images, int max_pts, cusift_compact_point *d_packed, size_t capacity,
                               unsigned int *d_offsets);
int cusift_expand_points_host(const cusift_compact_point *h_compact, size_t n, cusift_point *h_points);

/* Trimmed wire format (new; optional): the 135 floats of a record that extraction WRITES -- the seven header fields and
 * the descriptor -- and nothing else: 540 B instead of 588, bit-exact (the other 12 floats of a SiftPoint are whatever
 * the caller's buffer held, cuSIFT.cu:24,29, so nothing is lost).  8 % fewer bytes per record over PCIe / xGMI.
 * cusift_pack_points_trimmed: as cusift_pack_points.  cusift_expand_trimmed (device, asynchronous) /
 * cusift_expand_trimmed_host: trimmed -> SiftPoint records, the 12 unwritten floats zeroed. */
typedef struct cusift_trimmed_point {
  float coords2D[2];
  float scale;
  float sharpness;
  float edgeness;
  float orientation;
  float subsampling;
  float data[128];
} cusift_trimmed_point; /* 540 bytes */
int cusift_pack_points_trimmed(cusift_ctx *ctx, const cusift_point *d_points, const unsigned int *d_counters,
                               int n_images, int max_pts, cusift_trimmed_point *d_packed, size_t capacity,
                               unsigned int *d_offsets);
int cusift_expand_trimmed(cusift_ctx *ctx, const cusift_trimmed_point *d_trimmed, size_t n, cusift_point *d_points);
int cusift_expand_trimmed_host(const cusift_trimmed_point *h_trimmed, size_t n, cusift_point *h_points);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (BASELINE configs[3], [4]) --------------------------------
 * New functionality: the reference is single-GPU, single-image (no collective call sites).
 * A communicator wraps one ncclComm_t bound to a context: every exchange is enqueued on that context's stream.
 * RCCL is loaded at run time from the directory of the process's HIP runtime (or $CUSIFT_RCCL_LIB, or the library
 * named by cusift_comm_use_library -- any library exporting the nine nccl* entry points used here; the tests bind an
 * in-process transport that way, next to the real RCCL); a program that never creates a communicator never needs it.
 *   rank 0:  cusift_comm_get_unique_id(id)  -> hand the 128 bytes to every rank (MPI_Bcast, a TCP store, a file ...)
 *   all:     cusift_comm_create(&comm, ctx, id, rank, world)      (collective: ncclCommInitRank)
 */
#define CUSIFT_UNIQUE_ID_BYTES 128
typedef struct cusift_comm cusift_comm;
/* The library cusift_comm_get_unique_id / cusift_comm_create bind from now on (NULL or "": the default search). */
int cusift_comm_use_library(const char *path);
int cusift_comm_get_unique_id(char id[CUSIFT_UNIQUE_ID_BYTES]);
int cusift_comm_create(cusift_comm **out, cusift_ctx *ctx, const char id[CUSIFT_UNIQUE_ID_BYTES], int rank, int world);
int cusift_comm_destroy(cusift_comm *comm);
int cusift_comm_rank(cusift_comm *comm, int *rank, int *world);
cusift_ctx *cusift_comm_ctx(cusift_comm *comm); /* the context (device + stream) the communicator is bound to */
/* What the bound LIBRARY reports for this communicator: ncclCommCount, ncclCommUserRank, ncclGetVersion (major * 10000 +
 * minor * 100 + patch) -- each -1 if the library does not export the call.  A benchmark line that carries lib_ranks
 * proves "RCCL saw N ranks" without trusting the caller's own bookkeeping (cusift_comm_rank). */
int cusift_comm_info(cusift_comm *comm, int *lib_ranks, int *lib_rank, int *lib_version);
/* Path of the library bound last ("" before the first communicator call). */
const char *cusift_comm_library(void);
/* Tests / world == 1: route the local shard (and rows addressed to this rank) through ncclSend/ncclRecv to self too,
 * so that one GPU exercises the grouped p2p path.  Also settable with CUSIFT_COMM_SELF_P2P=1. */
int cusift_comm_set_self_p2p(cusift_comm *comm, int on);
/* Pre-sizes what the all-gatherv needs so that no call of the loop allocates or synchronises: `tickets` exchanges may
 * be in flight at once (begin() without its finish(); default 4), each with n_images_max count slots per rank;
 * stage_records only matters with self_p2p (the staging copy of the local shard). */
int cusift_comm_reserve(cusift_comm *comm, int n_images_max, int tickets, size_t stage_records);
/* 0 (default): exact sizes -- finish() reads the gathered counts on the host and posts ncclSend/ncclRecv of exactly the
 * valid records.  1: whole regions travel (region_cap records per peer whatever the counts), posted by begin(): the
 * exchange needs no host read at all, at the price of the bytes; for small capacities (a tiled image's merge). */
int cusift_comm_set_fixed_size(cusift_comm *comm, int on);
/* Wire format of the gathered records.  0 (default): cusift_point (588 B, exact).  1: they travel -- and arrive -- as
 * cusift_compact_point (160 B: exact header fields, 8-bit descriptor; see cusift_pack_points_compact): 3.7x fewer bytes
 * over xGMI when the exchange, not the extraction, bounds a step.  2: as cusift_trimmed_point (540 B: the 135 floats
 * extraction writes, EXACT; cusift_expand_trimmed makes SiftPoint records of them).  d_gathered then holds
 * world * region_cap records of that format. */
int cusift_comm_set_wire_format(cusift_comm *comm, int format);
/* Diagnostic: how many finish() calls found their counts not yet there, i.e. the host was ahead of the GPU (in a
 * GPU-bound pipelined loop that is the normal case and costs nothing: the device has the caller's other steps queued). */
unsigned long long cusift_comm_host_waits(cusift_comm *comm);
/* ... and the wall time those finish() calls spent waiting, in milliseconds. */
double cusift_comm_host_wait_ms(cusift_comm *comm);
/* Diagnostic: synchronising HIP calls (hipStreamSynchronize) this communicator has made so far -- they only happen while
 * its buffers are (re)sized: after cusift_comm_reserve the number stays put through any number of begin / finish. */
unsigned long long cusift_comm_hip_syncs(cusift_comm *comm);

/* All-gatherv of SiftData: every rank ends up with the valid records of ALL ranks' images.  d_gathered is `world`
 * REGIONS of region_cap records (cusift_point, or cusift_compact_point after cusift_comm_set_wire_format(comm, 1));
 * region r holds rank r's records packed back to back in image order (h_totals[r] of them).  Fixed region starts are what lets a rank pack its shard into place before anybody's counts are known.
 *   begin  (asynchronous, no host wait): orders the exchange after everything enqueued on `producer` so far (the
 *          context that extracted d_points; NULL: the caller has ordered the streams), clamps the per-image counters on
 *          the device, packs the local shard straight into region `rank` of d_gathered -- after which d_points /
 *          d_counters are free again: cusift_ctx_wait(producer, cusift_comm_ctx(comm)) before overwriting them --,
 *          ncclAllGather of the counts (n_images_max slots per rank: the largest image count of any rank, the same value
 *          on every rank) and a kernel that publishes them to pinned host memory.
 *   finish (of the oldest begin): READS the counts on the host -- the sizes of ncclSend/ncclRecv are host arguments --
 *          which is no wait when the caller has enqueued a step or more of other work since begin() (up to `tickets`
 *          begins may be outstanding); writes h_counts[world][n_images_max] and h_totals[world] (either may be NULL) and
 *          posts ONE ncclGroup of ncclSend/ncclRecv: each shard travels directly to each peer over its xGMI link.
 *          CUSIFT_ERR_NOMEM (on every rank alike, nothing sent) if a rank's total exceeds region_cap.
 * Nothing here allocates once cusift_comm_reserve() has been called.  n_images <= 256.
 * cusift_allgatherv() = begin + finish.  cusift_compact_gathered: the regions back to back in rank order (world
 * asynchronous device copies on ctx's stream), for consumers that want one list. */
int cusift_allgatherv_begin(cusift_comm *comm, cusift_ctx *producer, const cusift_point *d_points,
                            const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                            void *d_gathered, size_t region_cap);
int cusift_allgatherv_finish(cusift_comm *comm, unsigned int *h_counts, size_t *h_totals);
int cusift_allgatherv(cusift_comm *comm, cusift_ctx *producer, const cusift_point *d_points,
                      const unsigned int *d_counters, int n_images, int max_pts, int n_images_max,
                      void *d_gathered, size_t region_cap, unsigned int *h_counts, size_t *h_totals);
int cusift_compact_gathered(cusift_ctx *ctx, const cusift_point *d_gathered, size_t region_cap, int world,
                            const size_t *h_totals, cusift_point *d_out, size_t capacity);
/* Expand on arrival: regions gathered in the trimmed wire format (cusift_comm_set_wire_format(comm, 2): 540 B per record
 * over xGMI) -> the same regions as SiftPoint records (588 B, what the reference's SiftData holds; the 12 floats
 * extraction never writes -- uninitialised in the reference, cuSIFT.cu:24,29 -- are zero), region r at
 * d_points + r * region_cap, h_totals[r] records each (as cusift_allgatherv_finish returned them).  One launch for all
 * ranks on the communicator's stream, behind the exchange; asynchronous.  Every rank then ends the step holding the
 * SiftData of all ranks' images exactly as with the 588-byte wire format, for 8 % fewer bytes per link. */
int cusift_expand_gathered(cusift_comm *comm, const cusift_trimmed_point *d_gathered, size_t region_cap,
                           const size_t *h_totals, cusift_point *d_points);

/* Rows of a pitched float image between ranks, as one ncclGroup: op i sends send_rows[i] rows starting at local row
 * send_row[i] of d_band to peers[i] and receives recv_rows[i] rows from it into local row recv_row[i] (`pitch` floats
 * per row; the peer must post the mirror image; d_band holds `band_rows` rows and every op is checked against that --
 * RCCL reads and writes the rows on the device, where a range outside the allocation is a fault).  Asynchronous.
 * cusift_exchange_halos is the strip tiling's per-octave step (cusift_*_band entry points, BASELINE configs[4]): a band
 * is [top_halo rows of the neighbour above][own_rows][bottom_halo rows of the neighbour below]; the first / last
 * `send_rows` owned rows go to rank-1 / rank+1 and their counterparts arrive in the halo rows (every interior rank
 * uses the same send_rows == its neighbours' halo depth; rank 0 has top_halo = 0, the last rank bottom_halo = 0). */
int cusift_exchange_rows(cusift_comm *comm, float *d_band, int pitch, int band_rows, int n_ops, const int *peers,
                         const int *send_row, const int *send_rows, const int *recv_row, const int *recv_rows);
int cusift_exchange_halos(cusift_comm *comm, float *d_band, int pitch, int top_halo, int own_rows, int bottom_halo,
                          int send_rows);

/* Pre-size the arena for cusift_extract_bands over up to n_bands bands of one image with max_pts records (the strip
 * tiling's per-rank step; cusift_tiled_create calls it on every rank so that no extraction allocates). */
int cusift_ctx_reserve_bands(cusift_ctx *ctx, int n_bands, int max_pts);

/* ---- band ("tile") forms: one large image strip-tiled over several GPUs (BASELINE configs[4]) -- */
/* A band is `h` rows of device memory whose local row 0 is row `row0` of a global image with `h_global` rows
 * (same width).  Row addressing is "clamp to the global image, then translate", so a band that carries enough
 * halo rows gives bit-identical results to the whole image for the rows it owns.  New functionality: the
 * reference has no tiling (its scratch arena is sized for the whole image, cuSIFT.cu:81-98).
 * cusift_scale_down_band: computes global rows [r_begin, r_end) of the half-size image into a destination band
 *   that starts at global row dst_row0, from a source band {src_row0, h_src_global}; needs source rows
 *   2r-1 .. 2r+3 (cuSIFT_D.cu:75,123-125) inside the source band.
 * cusift_detect_band: fused LaplaceMulti+FindPointsMulti with extremum centres restricted to global rows
 *   [cy_begin, cy_end); keypoint rows are written in global coordinates.  Needs >= 5 halo rows of true data on each
 *   side that is not the image border (4 blur + 1 extremum; CUSIFT_ERR_INVALID otherwise).
 * cusift_describe_band: ComputeOrientations + ExtractSiftDescriptors for keypoints in global coordinates
 *   (root_sift as cusift_params.root_sift).
 *   d_flags (may be NULL): one counter, incremented for every keypoint whose sampling footprint (orientation window,
 *   rotated descriptor grid, +-1 px taps, bilinear 2x2) reaches beyond the band on a side that is not the image
 *   border -- such a keypoint samples clamped rows instead of the neighbour's and would differ from the whole image;
 *   the caller must treat a non-zero count as an error (cusift_amd.tiling.StripExtractor.check does). */
int cusift_scale_down_band(cusift_ctx *ctx, float *d_dst, int dst_pitch, int dst_row0, int r_begin, int r_end,
                           const float *d_src, int w, int h_src, int src_pitch, int src_row0, int h_src_global,
                           float variance);
int cusift_detect_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0, int h_global,
                       int cy_begin, int cy_end, float init_blur, float peak_thresh, float edge_thresh,
                       float subsampling, cusift_point *d_points, int max_pts, unsigned int *d_counter);
int cusift_describe_band(cusift_ctx *ctx, const float *d_img, int w, int h, int pitch, int row0, int h_global,
                         cusift_point *d_points, int max_pts, const unsigned int *d_first,
                         const unsigned int *d_counter, float subsampling, int tex_frac_bits, int root_sift,
                         unsigned int *d_flags);
/* One octave of a strip-tiled image as a rank holds it: `h` local rows (global rows row0 .. row0 + h - 1 of h_global),
 * extremum centres restricted to global rows [cy_begin, cy_end). */
typedef struct cusift_band {
  const float *d_img;
  int w, h, pitch;
  int row0, h_global, cy_begin, cy_end;
  float init_blur, subsampling;
} cusift_band;
/* cusift_detect_band + cusift_describe_band for SEVERAL consecutive octaves of one image (bands[0] the finest) with one
 * detection launch and one description launch.  *d_counter (device) counts what d_points holds already -- the coarser
 * octaves of a root rank, described: left alone -- and the bands' keypoints are appended behind it, coarsest band
 * first; the same records as the per-octave calls leave.  At most 8 bands.  d_flags as in cusift_describe_band. */
int cusift_extract_bands(cusift_ctx *ctx, const cusift_band *bands, int n_bands, float peak_thresh, float edge_thresh,
                         cusift_point *d_points, int max_pts, unsigned int *d_counter, int tex_frac_bits,
                         int root_sift, unsigned int *d_flags);

/* ---- one large image strip-tiled over the ranks (BASELINE configs[4]) -------------------------------------------
 * The rank-side driver of the tiling: plan, bands, per-octave ScaleDown -> halo exchange -> band detection and
 * description, coarse-octave collapse onto rank 0, footprint check.  Mirrors the octave loop of cuSIFT.cu:175-202 (the
 * reference itself has no tiling); the union of the ranks' SiftData equals the whole-image extraction bit for bit.
 * Rank k owns base rows [k*H/world, (k+1)*H/world).  Everything runs on ctx's stream; `comm` (NULL for world == 1 or
 * for extractors driven with cusift_tiled_exchange_virtual) must be bound to the same context or stream.
 *   cusift_tiled_create(&t, ctx, comm, rank, world, W, H, &params, 0)     allocates the bands (halo_rows 0 = 48)
 *   cusift_tiled_extract(t, d_strip, strip_pitch, d_points, d_counter)    collective, asynchronous: my owned base rows
 *                                                                         in, my SiftData (params.max_pts records) out
 *   cusift_tiled_check(t, &flagged)                                       blocking; an error if a keypoint's sampling
 *                                                                         footprint left the halo
 *   cusift_allgatherv(comm, ...)                                          merged SiftData on every rank
 * cusift_tiled_extract is cusift_tiled_load, then for each octave o cusift_tiled_build_octave(o) (o > 0) and
 * cusift_tiled_exchange(o), then cusift_tiled_process; the steps are public so that P extractors of one process can be
 * stepped together with cusift_tiled_exchange_virtual (device copies instead of RCCL: a plan run on one GPU).
 * cusift_tiled_plan is the row geometry alone (no GPU): octave sizes, the first collapsed octave (== n_octaves: none),
 * the rows [own_begin, own_end) rank `rank` owns in `octave` and the rows [band_begin, band_end) its band holds. */
#define CUSIFT_TILED_DEFAULT_HALO 48
typedef struct cusift_tiled cusift_tiled;
int cusift_tiled_plan(int W, int H, int world, int num_octaves, int halo_rows, int rank, int octave, int *n_octaves,
                      int *collapse_octave, int *w, int *h, int *pitch, int *own_begin, int *own_end, int *band_begin,
                      int *band_end);
int cusift_tiled_create(cusift_tiled **out, cusift_ctx *ctx, cusift_comm *comm, int rank, int world, int W, int H,
                        const cusift_params *p, int halo_rows);
int cusift_tiled_destroy(cusift_tiled *t);
int cusift_tiled_info(cusift_tiled *t, int *n_octaves, int *collapse_octave, int *root, int *halo_rows);
int cusift_tiled_band(cusift_tiled *t, int octave, float **d_band, int *w, int *h_global, int *pitch, int *own_begin,
                      int *own_end, int *band_begin, int *band_end);
int cusift_tiled_load(cusift_tiled *t, const float *d_strip, int strip_pitch);
int cusift_tiled_build_octave(cusift_tiled *t, int octave);
int cusift_tiled_exchange(cusift_tiled *t, int octave);
int cusift_tiled_exchange_virtual(cusift_tiled **ranks, int n, int octave);
int cusift_tiled_process(cusift_tiled *t, cusift_point *d_points, unsigned int *d_counter);
int cusift_tiled_extract(cusift_tiled *t, const float *d_strip, int strip_pitch, cusift_point *d_points,
                         unsigned int *d_counter);
int cusift_tiled_check(cusift_tiled *t, unsigned int *flagged);


#ifdef __cplusplus
}
#endif
#endif /* CUSIFT_AMD_MULTIGPU_H */

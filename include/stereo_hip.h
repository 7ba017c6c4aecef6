/*
 * stereo_hip.h -- C ABI of libstereo_hip.so, the MI355X (gfx950) implementation of the
 * dense-stereo hot path: weighted-census cost -> per-disparity edge-aware filter -> WTA.
 *
 * The reference is pure Python; its "operator API" is the set of module-level functions in
 * census.py / guided_filter.py / filters.py / stereo.py.  Each entry point below names the
 * reference function it replaces.  All arrays are row-major and contiguous; sizes are in
 * elements.  Functions return 0 on success or a negative STEREO_E* code;
 * stereo_hip_last_error() returns a thread-local description of the last failure.
 *
 * *_host entry points take HOST pointers, stage data through device memory they allocate
 * themselves and block until the result is in the caller's output buffer.
 * *_dev entry points take DEVICE pointers and a HIP stream (passed as void*, i.e. a
 * hipStream_t), allocate nothing and return after enqueueing.
 */
#ifndef STEREO_HIP_H
#define STEREO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEREO_OK 0
#define STEREO_EINVAL (-1)       /* argument violates the reference's own validation */
#define STEREO_EUNSUPPORTED (-2) /* valid for the reference, not implemented on the GPU path */
#define STEREO_EHIP (-3)         /* HIP runtime error (no device, launch failure, ...) */
#define STEREO_ENOMEM (-4)       /* device or host allocation failed */
#define STEREO_ENONFINITE (-5)   /* non-finite input where the reference raises ValueError */

/* element types accepted for gray images (the census compares in the caller's dtype) */
#define STEREO_DT_F32 0
#define STEREO_DT_F64 1
#define STEREO_DT_U8 2
#define STEREO_DT_U16 3 /* image files only (to_gray) */

/* aggregation filter (stereo.py:136-165) */
#define STEREO_FILTER_GUIDED 0
#define STEREO_FILTER_MEDIAN 1     /* exact median of the float window: scipy/vectorized/naive */
#define STEREO_FILTER_MEDIAN_U8Q 2 /* "opencv": uint8 min-max quantised median (filters.py:139-180) */
#define STEREO_FILTER_GAUSSIAN 3
#define STEREO_FILTER_BILATERAL 4

/* Parameters of stereo.aggregate_and_wta / stereo.compute_disparity (stereo.py:94-245). */
typedef struct StereoParams {
    int32_t wct_radius;     /* census radius, 1..8 on the GPU path (reference default 4) */
    double base_weight;     /* weight of the r=1 ring (reference default 8.0) */
    int32_t filter_kind;    /* STEREO_FILTER_* */
    int32_t guided_radius;  /* guided-filter window radius (default 3) */
    double guided_eps;      /* guided-filter regulariser; rounded to float32 as the reference does */
    int32_t median_radius;  /* median window radius (default 3) */
    double gaussian_sigma;  /* default 1.0; window radius = max(ceil(3 sigma), 1) */
    double bilateral_sigma; /* default 1.0; same sigma for space and range */
} StereoParams;

/* ---- library state -------------------------------------------------------------------- */
int stereo_hip_init(int device);             /* select the device used by *_host calls */
int stereo_hip_device_count(int *count);
const char *stereo_hip_last_error(void);
const char *stereo_hip_version(void);

/* ---- census.py ------------------------------------------------------------------------- */
/* census.generate_offsets (census.py:20-45): out[8*radius][3] = (dy, dx, r). Host only. */
int stereo_generate_offsets(int radius, int32_t *out);
/* census.compute_weights (census.py:48-62): float32(base_weight / 2**(r-1)). Host only. */
int stereo_compute_weights(const int32_t *offsets, int n, double base_weight, float *out);
/* large_value of census.py:240 for (radius, base_weight). Host only. */
int stereo_large_value(int radius, double base_weight, float *out);
/* census.compute_wct_cost_volume (census.py:205-269): out_dsi[H][W][dmax] float32. */
int stereo_wct_cost_volume_host(const void *left, const void *right, int dtype, int height, int width,
                                int dmax, int radius, double base_weight, float *out_dsi);

/* ---- guided_filter.py ------------------------------------------------------------------ */
/* _integral_image_numba / README's integral_image (guided_filter.py:21-47): out[(H+1)][(W+1)]. */
int stereo_integral_image_host(const double *image, int height, int width, double *out);
/* box_filter_mean (guided_filter.py:141-179): clipped-window mean, radius >= 0. */
int stereo_box_filter_mean_host(const double *image, int height, int width, int radius, double *out);
/* box_filter_mean_with_indices (guided_filter.py:155-179): caller-supplied window bounds
 * y0/y1[H], x0/x1[W] (inclusive) and per-pixel area[H][W]. */
int stereo_box_filter_mean_indexed_host(const double *image, int height, int width, const int32_t *y0,
                                        const int32_t *y1, const int32_t *x0, const int32_t *x1,
                                        const double *area, double *out);
/* prepare_guided_filter (guided_filter.py:182-217): mean_guide, mean_gg, var_g, each [H][W]. */
int stereo_guided_prepare_host(const double *guide, int height, int width, int radius, double *mean_guide,
                               double *mean_gg, double *var_g);
/* guided_filter (guided_filter.py:256-275) = prepare_guided_filter + guided_filter_with_precompute. */
int stereo_guided_filter_host(const double *guide, const float *src, int height, int width, int radius,
                              double eps, float *out);

/* prepare_guided_filter + guided_filter_with_precompute (guided_filter.py:182-253) as the reference uses them: the
 * guide's statistics once per scene, then one call per layer (stereo.py:143-150).  stereo_guided_plan_create uploads
 * the guide and keeps it, its box mean and 1 / (var + eps) on the current device; stereo_guided_plan_apply moves one
 * float32 layer in, filters it and moves the result out (same values as stereo_guided_filter_host);
 * stereo_guided_plan_destroy releases the plan.  A plan belongs to the device it was made on and to one caller at a
 * time. */
int stereo_guided_plan_create(const double *guide, int height, int width, int radius, double eps, void **plan);
int stereo_guided_plan_apply(void *plan, const float *src, float *out);
int stereo_guided_plan_destroy(void *plan);

/* ---- filters.py ------------------------------------------------------------------------ */
/* median_filter (filters.py:227-256). mode: STEREO_FILTER_MEDIAN or STEREO_FILTER_MEDIAN_U8Q. */
int stereo_median_filter_host(const float *image, int height, int width, int radius, int mode, float *out);
/* median_filter(method="opencv") on a uint8 image: no rescaling (filters.py:158-161). */
int stereo_median_filter_u8_host(const uint8_t *image, int height, int width, int radius, float *out);
/* gaussian_filter (filters.py:259-276); numerics follow the reference's naive path (:191-204). */
int stereo_gaussian_filter_host(const float *image, int height, int width, double sigma, float *out);
/* bilateral_filter (filters.py:279-310). */
int stereo_bilateral_filter_host(const float *image, int height, int width, double sigma, float *out);

/* ---- stereo.py ------------------------------------------------------------------------- */
/* aggregate_and_wta (stereo.py:94-174) on a materialised cost volume dsi[H][W][D].
 * guide_dtype: STEREO_DT_F32 or STEREO_DT_F64. Layers d in [d_begin, d_end) are aggregated. */
int stereo_aggregate_wta_host(const float *dsi, const void *guide, int guide_dtype, int height, int width,
                              int dmax, const StereoParams *params, int d_begin, int d_end,
                              int32_t *out_disparity, float *out_min_cost);
/* compute_disparity (stereo.py:177-245).  float32 / uint8 images take a fused kernel -- the cost volume is never
 * materialised -- for the guided filter (radius 1..8), the uint8-quantised median, the Gaussian and the bilateral
 * filter (window radius 1..12) and, for census weights whose sums are exact in float32, the exact median
 * (radius 1..8).  Other combinations (float64 images, larger windows, the exact median with other weights) build
 * the cost layers on the device and filter them one at a time.  Disparities d in [d_begin, d_end) are searched. */
int stereo_compute_disparity_host(const void *left, const void *right, int dtype, int height, int width,
                                  int dmax, const StereoParams *params, int d_begin, int d_end,
                                  int32_t *out_disparity, float *out_min_cost);

/* ---- stereo_io.py / metrics: the steps either side of the path ------------------------------ */
/* to_gray (stereo_io.py:47-75): image[H][W][channels] (channels 1 or >= 3, the first three are R, G, B; uint8, uint16 or
 * float32) -> float32 gray[H][W]; normalize != 0 divides by float32(scale) and clips to [0, 1]. */
int stereo_to_gray_host(const void *image, int dtype, int height, int width, int channels, int normalize,
                        double scale, float *out_gray);
/* nanmin / nanmax of a float32 buffer (_resolve_normalization_scale, stereo_io.py:38-39).
 * Returns +inf / -inf when every element is NaN or n == 0. */
int stereo_nan_minmax_host(const float *values, size_t n, float *out_min, float *out_max);
/* _compute_pbm_rms (stereo.py:532-577): over pixels with finite ground truth and mask != 0.
 * out[0] = pbm (percent), out[1] = rms, out[2] = valid pixel count. EINVAL when no pixel is valid. */
int stereo_pbm_rms_host(const int32_t *disparity, const float *ground_truth, const uint8_t *mask, int height,
                        int width, double bad_threshold, double *out);

/* ---- device-pointer path (bench, multi-GPU sharding) ------------------------------------ */
/* Scratch the fused path needs for one (H, W) frame, in bytes. */
int stereo_fused_workspace_bytes(int height, int width, const StereoParams *params, size_t *bytes);
/* Fused compute_disparity over d in [d_begin, d_end) on float32 device images.  Writes one
 * packed key per pixel: (order-preserving uint32 image of the float32 min cost) << 32 | d, so
 * that an unsigned 64-bit MIN over keys of different disparity shards is the reference's
 * "lowest cost, ties to the lowest disparity" rule (stereo.py:167-170).
 * Covers the guided filter (radius 1..8), the uint8-quantised median (radius 1..12), the exact median
 * (radius 1..8, census weights whose sums are exact in float32) and the Gaussian / bilateral filters
 * (max(ceil(3 sigma), 1) <= 12); other parameters return STEREO_EUNSUPPORTED (the host entry point falls back to
 * the layer-at-a-time path for them). */
int stereo_compute_disparity_dev(const float *d_left, const float *d_right, int height, int width, int dmax,
                                 const StereoParams *params, int d_begin, int d_end, void *d_workspace,
                                 size_t workspace_bytes, uint64_t *d_keys, void *stream);
/* The two-call form of the path on device buffers, the data flow the reference's public API defines:
 * census.compute_wct_cost_volume (census.py:205-269) writes d_dsi[H][W][dmax] float32 (D innermost, census.py:250)
 * from float32 device images, and stereo.aggregate_and_wta (stereo.py:94-174, guided filter, radius 1..8,
 * float32 guide, finite costs) reads it back layer by layer (stereo.py:146) and writes one packed key per pixel.
 * d_workspace / workspace_bytes as for stereo_compute_disparity_dev (params->wct_radius sizes it). */
int stereo_wct_cost_volume_dev(const float *d_left, const float *d_right, int height, int width, int dmax, int radius,
                               double base_weight, void *d_workspace, size_t workspace_bytes, float *d_dsi,
                               void *stream);
int stereo_aggregate_wta_dev(const float *d_dsi, const float *d_guide, int height, int width, int dmax,
                             const StereoParams *params, int d_begin, int d_end, void *d_workspace,
                             size_t workspace_bytes, uint64_t *d_keys, void *stream);
/* Workspace of stereo_aggregate_wta_dev that lets it run at memory speed: the fused workspace plus room for the
 * volume's "cost codes".  The volume's layout (census.py:250, D innermost) puts a pixel's disparities side by side,
 * the per-layer filter of stereo.py:145-166 wants a layer's pixels side by side; with this workspace the entry
 * point first streams the volume once through an encoder that transposes it into one [H][W] layer per disparity of
 * 16-bit whole numbers of the volume's cost unit (census costs are multiples of a power of two, census.py:58-62),
 * and the two-column kernel of the single-call path filters those.  A volume whose costs are not such multiples (or
 * negative, not finite, 2^16 units and above), and guided radii 5..8, are served by the general kernel reading the
 * float32 volume in place -- as is every volume when the workspace only has stereo_fused_workspace_bytes. */
int stereo_aggregate_workspace_bytes(int height, int width, int dmax, const StereoParams *params, size_t *bytes);
/* The two steps of stereo_aggregate_wta_dev on their own, for callers that aggregate one volume several times
 * (stereo.aggregate_and_wta with different filter parameters on the same dsi, README:144-175): encode layers
 * [d_begin, d_end) of d_dsi[H][W][dmax] once into a buffer of stereo_cost_codes_bytes(H, W, d_end - d_begin), then
 * aggregate from the codes (the volume is still passed: it is what the general kernel reads if the codes turn out
 * unusable).  d_begin / d_end of the two calls must agree. */
int stereo_cost_codes_bytes(int height, int width, int n_disparities, size_t *bytes);
int stereo_encode_cost_volume_dev(const float *d_dsi, int height, int width, int dmax, int d_begin, int d_end,
                                  void *d_codes, size_t codes_bytes, void *stream);
int stereo_aggregate_wta_codes_dev(const void *d_codes, const float *d_dsi, const float *d_guide, int height, int width,
                                   int dmax, const StereoParams *params, int d_begin, int d_end, void *d_workspace,
                                   size_t workspace_bytes, uint64_t *d_keys, void *stream);
/* stereo_compute_disparity_dev without the workspace argument.  No counterpart in the reference; it is for
 * callers that cannot allocate device memory or carry a workspace: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The scratch (census words, guide statistics, per-layer ranges) is a
 * static device array inside the library, about 34 MB per GPU, sized for frames of at most
 * STEREO_NOALLOC_MAX_PIXELS pixels (any census radius); larger frames return STEREO_EINVAL.  The call allocates
 * nothing (no hipMalloc / hipFree, no library routine that allocates) and, like the other *_dev entry points,
 * only enqueues work on `stream`.  All calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (two streams, two threads) race on it -- run them one after the other. */
#define STEREO_NOALLOC_MAX_PIXELS 1048576
int stereo_compute_disparity_noalloc_dev(const float *d_left, const float *d_right, int height, int width, int dmax,
                                         const StereoParams *params, int d_begin, int d_end, uint64_t *d_keys,
                                         void *stream);
/* keys[i] = min(keys[i], other[i]) -- local merge of two disparity shards. */
int stereo_merge_keys_dev(uint64_t *d_keys, const uint64_t *d_other, size_t n, void *stream);
/* The same over n_slices equal-length slices laid end to end (what an all-to-all of key slices
 * delivers): d_slices[i] = min_j d_slices[j * chunk + i], one pass. */
int stereo_merge_key_slices_dev(uint64_t *d_slices, size_t chunk, int n_slices, void *stream);
/* Map keys to a signed-int64-comparable form and back (torch/RCCL reduce int64, not uint64). */
int stereo_keys_to_i64_dev(const uint64_t *d_keys, int64_t *d_out, size_t n, void *stream);
int stereo_keys_from_i64_dev(const int64_t *d_in, uint64_t *d_keys, size_t n, void *stream);
/* Unpack keys into int32 disparity and float32 min cost (never-updated pixels: 0 and +inf). */
int stereo_unpack_keys_dev(const uint64_t *d_keys, size_t n, int32_t *d_disparity, float *d_min_cost,
                           void *stream);

/* ---- row-band progress: overlap of the multi-GPU exchange with the search ------------------ */
/* No counterpart in the reference (a single-process CPU program); for callers that merge
 * disparity shards across GPUs and want to ship the top rows of the frame while the kernel is
 * still working on the rest.
 *
 * stereo_progress_plan splits the frame into at most n_bands row bands (<= 16) whose borders
 * suit the kernel these parameters select: row_end[b] is the end row (exclusive) of band b,
 * *bands_out the number of bands actually formed.
 * stereo_compute_disparity_progress_dev is stereo_compute_disparity_dev plus reporting:
 * d_progress points to 48 zero-initialised uint32 words owned by the caller; once all keys of
 * the rows of band b are final, word 16 + b holds `epoch` (use a new, increasing epoch for every
 * call).  Kernels without in-flight reporting set every flag when they have finished.
 * n_bands / row_end must come from stereo_progress_plan for the same frame and parameters.
 * stereo_wait_progress_dev makes `stream` wait (a polling wavefront, bounded to about ten
 * seconds; a time-out stores `epoch` in word 32 + band) until band `band` carries `epoch`;
 * enqueue it AFTER the producing call. */
int stereo_progress_plan(int height, int width, const StereoParams *params, int n_bands, int32_t *row_end,
                         int *bands_out);
int stereo_compute_disparity_progress_dev(const float *d_left, const float *d_right, int height, int width, int dmax,
                                          const StereoParams *params, int d_begin, int d_end, void *d_workspace,
                                          size_t workspace_bytes, uint64_t *d_keys, int n_bands,
                                          uint32_t *d_progress, uint32_t epoch, void *stream);
int stereo_wait_progress_dev(const uint32_t *d_progress, int band, uint32_t epoch, void *stream);
/* Whether work on stream_b can run while a kernel enqueued earlier on stream_a is still running.  The HIP runtime
 * maps streams onto a few hardware queues, and two streams on the same queue run in submission order: a band wait
 * enqueued on such a stream would only start after the search.  A caller that sets up a side stream for
 * stereo_wait_progress_dev tries candidates with this call (it runs two tiny kernels, synchronises both streams and
 * allocates a few words: set-up time only) and keeps one that reports *overlap = 1.  Do not give the side stream a
 * higher priority instead: a polling wait on a high-priority queue was seen to hold back the search it waits for. */
int stereo_streams_overlap(void *stream_a, void *stream_b, int *overlap);

/* The host entry points keep the device buffers they allocate in a per-process cache (at most 8 GiB
 * idle) so that a caller processing frame after frame does not pay hipMalloc / hipFree per call;
 * no counterpart in the reference.  This returns every cached buffer to the driver. */
int stereo_hip_release_cache(void);

/* The test hook and the profile below, like the buffer cache above, are process-global state: ONE caller at a
 * time per process (the reference's own contract: one synchronous call per scene from one thread,
 * stereo.py:665-681).  None of them is thread-safe. */
/* Test hook, no counterpart in the reference: selects the fused guided kernel compute_disparity
 * uses, so the variants can be compared on the same input.  0 = the best one that applies
 * (default), 1 = the general kernel (float64 sums exchanged through LDS), 2 = the fast kernel
 * with one image column per lane (the default fast kernel gives every lane two columns), 3 = the
 * two-column kernel's precise variant, float64 where the fast one has float32 (the default for every eps below
 * 8e-3 times max(1, max|guide|^2), guided radius <= 4).  Mode 1 also makes the fused uint8-median kernel take its
 * plain bisection instead of the bit-sliced count. */
int stereo_debug_force_general_kernel(int mode);

/* Timing of the dominant (fused) kernel with HIP events recorded on the launch stream, for
 * bench.py's roofline figure; no counterpart in the reference.  stereo_profile_begin creates up
 * to max_launches event pairs and arms recording; stereo_profile_end synchronises, returns the
 * summed kernel time in milliseconds and the number of launches seen, and disarms. */
int stereo_profile_begin(int max_launches);
int stereo_profile_end(double *total_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* STEREO_HIP_H */

// Shared host/device helpers for libstereo_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "stereo_hip.h"

namespace shp {

// ---- error reporting ---------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SHP_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) return ::shp::hip_fail(e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define SHP_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != STEREO_OK) return rc__; \
    } while (0)

// Check for launch errors right after a kernel launch.
#define SHP_LAUNCH_CHECK() SHP_HIP(hipGetLastError())

// ---- RAII device buffer (host-pointer entry points only) -----------------------------------
// Device allocations of the host entry points come from a small per-process cache: a frame-sized
// hipMalloc / hipFree pair costs about as much as a kernel, and a caller that processes frame after
// frame asks for the same sizes again and again.  stereo_hip_release_cache() returns everything.
void *pool_alloc(size_t bytes, size_t *capacity);   // nullptr on failure (error text set)
void pool_free(void *p, size_t capacity);
void pool_release_all();

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    size_t capacity = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 1;
        p = pool_alloc(n, &capacity);
        if (!p) return STEREO_ENOMEM;
        bytes = n;
        return STEREO_OK;
    }
    void release() {
        if (p) pool_free(p, capacity);
        p = nullptr;
        bytes = 0;
        capacity = 0;
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- packed (cost, disparity) keys ----------------------------------------------------------
// Order-preserving map float32 -> uint32 (negative values reverse, positive get the top bit).
__host__ __device__ static inline uint32_t f32_to_ordered(float f) {
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(f);
#else
    memcpy(&b, &f, 4);
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ static inline float ordered_to_f32(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}
// Key of a pixel no layer has improved yet: (+inf, d = 0), as stereo.py:134-135 initialises.
#define SHP_KEY_INIT ((uint64_t(0xFF800000u) << 32))
__host__ __device__ static inline uint64_t make_key(float cost, uint32_t d) {
    return (uint64_t(f32_to_ordered(cost)) << 32) | uint64_t(d);
}

// ---- census parameters shared by the cost kernels -------------------------------------------
struct CensusCfg {
    int radius;      // 1..8
    int word64;      // 0: 32-bit words (radius <= 4), 1: 64-bit words (radius 5..8)
    int exact;       // 1: every partial sum of the reference's float32 adds is exact
    float unit;      // cost of one unit of the packed integer sum (base/8 or base/128)
    float large;     // census.py:240
    float w[8];      // w[r-1] = float32(base_weight / 2**(r-1))
    float hw[8];     // w[r-1] / 2 (exact), for the branch-free form of the sequential sum
    int seq_fma;     // 1: every w is finite and every hw a normal number (the branch-free form applies)
};
int make_census_cfg(int radius, double base_weight, CensusCfg *cfg);

// ---- kernel launchers (device pointers, stream) ---------------------------------------------
int launch_census_pack(const void *d_img, int dtype, int H, int W, int radius, void *d_words, hipStream_t s);
int launch_census_pack_pair(const void *d_img0, const void *d_img1, int dtype, int H, int W, int radius, void *d_words0,
                            void *d_words1, hipStream_t s);
// the pre-pass of the fused guided path as one launch (census.hip): applies to float32 images with these radii
bool census_stats_merged(int census_radius, int guided_radius);
int launch_census_stats(const float *d_left, const float *d_right, int H, int W, int census_radius, int guided_radius,
                        double eps, void *d_lw, void *d_rw, double *d_mean, double *d_invden, void *d_gmax, hipStream_t s);
int launch_dsi(const void *d_lw, const void *d_rw, int H, int W, int dmax, const CensusCfg &cfg, float *d_dsi,
               hipStream_t s);
int launch_unpack_keys(const uint64_t *keys, size_t n, int32_t *disp, float *cost, hipStream_t s);
int launch_merge_keys(uint64_t *keys, const uint64_t *other, size_t n, hipStream_t s);
int launch_merge_key_slices(uint64_t *slices, size_t chunk, int n_slices, hipStream_t s);
int launch_fill_keys(uint64_t *keys, size_t n, hipStream_t s);
int launch_keys_to_i64(const uint64_t *keys, int64_t *out, size_t n, hipStream_t s);
int launch_keys_from_i64(const int64_t *in, uint64_t *keys, size_t n, hipStream_t s);

// guided filter
// d_gmax (optional, 8 bytes): receives the largest finite |guide| as a float32, then a word that is
// non-zero when some guide pixel is NaN or infinite
int launch_guide_stats(const void *d_guide, int guide_dtype, int H, int W, int radius, double eps,
                       double *d_mean, double *d_invden, void *d_gmax, hipStream_t s);
size_t guide_stats_scratch_bytes(int H, int W, int radius);
bool fused_guided_supported(int guided_radius);
// cost source: census words (d_lw/d_rw, cfg) when d_dsi == nullptr, else the materialised volume.
int launch_fused_guided(const void *d_lw, const void *d_rw, const CensusCfg *cfg, const float *d_dsi,
                        const void *d_guide, int guide_dtype, const double *d_mean, const double *d_invden,
                        int H, int W, int dmax, int guided_radius, int d_begin, int d_end, uint64_t *d_keys,
                        float *d_q_out, int atomic_out, hipStream_t s,
                        const float *gate_gmax = nullptr, float gate_eps = 0.f, float gate_min = 0.f,
                        const uint32_t *gate_code_stats = nullptr);


// fast fused kernel (guided_fast.hip): exact census weights, float32 guide, guided radius <= 4
bool fused_pair_supported(const CensusCfg &cfg, int guided_radius);
int fused_pair_slices(int H, int W, int guided_radius, bool word64, int n_disp);
// Progress words of a launch that reports finished row bands (see FastArgs::progress).
constexpr int PROGRESS_MAX_BANDS = 16;
struct BandProgress {
    uint32_t *words;      // device, 2 * PROGRESS_MAX_BANDS words
    int band_tile_rows;   // tile rows per band (0: the kernel does not report, flags are set after it)
    uint32_t epoch;
};
int pair_tile_height(int guided_radius, bool word64);
int launch_fused_guided_pair(const void *d_lw, const void *d_rw, const CensusCfg &cfg, const float *d_guide,
                             const double *d_mean, const double *d_invden, const float *d_gmax, int H, int W,
                             int dmax, int guided_radius, int d_begin, int d_end, int n_slices, uint64_t *d_keys,
                             float *d_q_out, hipStream_t s, const BandProgress *progress = nullptr,
                             float gate_eps = 0.f, float gate_min = 0.f, bool precise = false, float gate_low = 0.f,
                             const uint16_t *d_codes = nullptr, const uint32_t *d_code_stats = nullptr, int code_d0 = 0);
// Cost codes of a caller-supplied volume (guided_pair_codes.hip): layers [d_begin, d_end) of dsi[H*W][D] as 16-bit
// whole numbers of the volume's cost unit, one [H*W] layer per disparity, and the words that tell the kernels
// reading them whether they may (PAIR_CODE_STATS_WORDS uint32 at d_stats).
size_t cost_codes_bytes(int H, int W, int n_disp);
int launch_encode_cost_codes(const float *d_dsi, int H, int W, int D, int d_begin, int d_end, uint16_t *d_codes,
                             uint32_t *d_stats, hipStream_t s);
int launch_progress_finish(uint32_t *words, int n_bands, uint32_t epoch, hipStream_t s);
int launch_progress_wait(const uint32_t *words, int band, uint32_t epoch, hipStream_t s);
int streams_overlap(hipStream_t a, hipStream_t b, bool *overlap);   // set-up time only: synchronises and allocates
bool fused_fast_supported(const CensusCfg &cfg, int guided_radius);
int fused_fast_slices(int H, int W, int guided_radius, int n_disp, bool word64 = false);
int launch_fused_guided_fast(const void *d_lw, const void *d_rw, const CensusCfg &cfg, const float *d_guide,
                             const double *d_mean, const double *d_invden, const float *d_gmax, int H, int W,
                             int dmax, int guided_radius, int d_begin, int d_end, int n_slices, uint64_t *d_keys,
                             float *d_q_out, hipStream_t s, float gate_eps = 0.f, float gate_min = 0.f);

// volume-free median (uint8-quantised) / Gaussian / bilateral compute_disparity (fused_filters.hip)
bool fused_filter_supported(int filter_kind, int radius);
size_t fused_filter_minmax_bytes(int n_disp);
int launch_fused_filter(const void *d_lw, const void *d_rw, const CensusCfg &cfg, int H, int W, int filter_kind,
                        int radius, double sigma, int d_begin, int d_end, uint32_t *d_mm, uint64_t *d_keys,
                        hipStream_t s, bool plain_median = false);

int launch_dsi_layers(const void *d_lw, const void *d_rw, int H, int W, int d_begin, int d_end, const CensusCfg &cfg,
                      float *d_layers, hipStream_t s);

// summed-area tables / box means / layer-at-a-time guided filter (box.hip)
int launch_integral(const void *d_img, int dtype, int H, int W, double *d_tmp, double *d_integral, hipStream_t s);
int launch_box_from_integral(const double *d_integral, int H, int W, const int *y0, const int *y1, const int *x0,
                             const int *x1, const double *area, int radius, double *d_out, hipStream_t s);
int launch_box_mean(const void *d_img, int dtype, int H, int W, int radius, double *d_tmp, double *d_integral,
                    double *d_out, hipStream_t s);
size_t layered_guided_ws_bytes(int H, int W);
int layered_guided_prepare(const void *d_guide, int guide_dtype, int H, int W, int radius, void *ws, hipStream_t s);
void layered_guided_planes(void *ws, int H, int W, double **g64, double **mean_g, double **mean_gg, double **var);
int layered_guided_apply(const float *d_src, int H, int W, int radius, double eps, void *ws, float *d_out,
                         hipStream_t s);

// single-layer filters and layered-loop helpers (filters.hip)
int radius_from_sigma(double sigma);
void gaussian_kernel_host(int radius, double sigma, float *k);
size_t filter_kernel_table_bytes(int r);
int launch_gaussian(const float *d_img, int H, int W, int r, const float *d_kern, float *d_out, hipStream_t s);
int launch_bilateral(const float *d_img, int H, int W, int r, const float *d_kern, double sigma, float *d_out,
                     hipStream_t s);
int launch_median_f32(const float *d_img, int H, int W, int r, float *d_out, hipStream_t s);
int launch_median_u8(const uint8_t *d_img, int H, int W, int r, float inv_scale, float offset, int rescale,
                     float *d_out, hipStream_t s);
int launch_median_u8q(const float *d_img, int H, int W, int r, uint32_t *d_mm, uint8_t *d_q, float *d_sc,
                      float *d_out, hipStream_t s);
int launch_dsi_to_layers(const float *d_dsi, size_t npix, int D, float *d_layers, hipStream_t s);
int launch_wta_update(const float *d_layer, size_t n, int d, uint64_t *d_keys, hipStream_t s);


// gray conversion and metrics (io_ops.hip)
int launch_to_gray(const void *d_img, int dtype, size_t npix, int channels, int normalize, float scale, float *d_out,
                   hipStream_t s);
int launch_nan_minmax(const float *d_v, size_t n, uint32_t *d_mm, hipStream_t s);
int launch_metrics(const int32_t *d_disp, const float *d_gt, const uint8_t *d_mask, size_t n, float threshold,
                   unsigned long long *d_acc, double *d_acc_sq, hipStream_t s);

}  // namespace shp

// Median / Gaussian / bilateral filters on one 2-D layer (filters.py:95-310) and the helpers of
// the layer-at-a-time aggregation loop (stereo.py:145-170).
//
// Every filter kernel stages a (TH + 2r) x (TW + 2r) tile of the layer in LDS with the
// reference's border rule already applied (edge replication for the median, reflect-101 for
// Gaussian and bilateral), then each thread produces output pixels from its window.
#include "common.hpp"

#pragma clang fp contract(off)

namespace shp {

constexpr int FT_TW = 64, FT_TH = 16;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

enum { BORDER_EDGE = 0, BORDER_REFLECT101 = 1 };

template <typename T, int BORDER>
__device__ __forceinline__ void stage_tile(const T *__restrict__ img, int H, int W, int r, int x0, int y0, T *tile) {
    const int LW = FT_TW + 2 * r, LH = FT_TH + 2 * r;
    for (int i = threadIdx.x; i < LW * LH; i += blockDim.x) {
        int ly = i / LW, lx = i - ly * LW;
        int gy = y0 - r + ly, gx = x0 - r + lx;
        if (BORDER == BORDER_EDGE) { gy = clampi(gy, 0, H - 1); gx = clampi(gx, 0, W - 1); }
        else { gy = reflect101(gy, H); gx = reflect101(gx, W); }
        tile[i] = img[(size_t)gy * W + gx];
    }
}

// ---- Gaussian (filters.py:191-204): out = sum(window * kernel), float32 ---------------------
__global__ __launch_bounds__(256) void gaussian_kernel2d(const float *__restrict__ img, int H, int W, int r,
                                                         const float *__restrict__ kern, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tile = reinterpret_cast<float *>(smem_raw);
    const int LW = FT_TW + 2 * r, K = 2 * r + 1;
    float *ks = tile + LW * (FT_TH + 2 * r);
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    stage_tile<float, BORDER_REFLECT101>(img, H, W, r, x0, y0, tile);
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) ks[i] = kern[i];
    __syncthreads();
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < FT_TH; ly += 4) {
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float acc = 0.f;
        for (int dy = 0; dy < K; ++dy)
            for (int dx = 0; dx < K; ++dx) acc += tile[(ly + dy) * LW + lx + dx] * ks[dy * K + dx];
        out[(size_t)gy * W + gx] = acc;
    }
}

// ---- bilateral (filters.py:279-310) -----------------------------------------------------------
__global__ __launch_bounds__(256) void bilateral_kernel2d(const float *__restrict__ img, int H, int W, int r,
                                                          const float *__restrict__ kern, float denom,
                                                          float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tile = reinterpret_cast<float *>(smem_raw);
    const int LW = FT_TW + 2 * r, K = 2 * r + 1;
    float *ks = tile + LW * (FT_TH + 2 * r);
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    stage_tile<float, BORDER_REFLECT101>(img, H, W, r, x0, y0, tile);
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) ks[i] = kern[i];
    __syncthreads();
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < FT_TH; ly += 4) {
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float c = tile[(ly + r) * LW + lx + r];
        float wsum = 0.f, vsum = 0.f;
        for (int dy = 0; dy < K; ++dy)
            for (int dx = 0; dx < K; ++dx) {
                float v = tile[(ly + dy) * LW + lx + dx];
                float diff = v - c;
                float w = ks[dy * K + dx] * expf(-(diff * diff) / denom);
                wsum += w;
                vsum += v * w;
            }
        out[(size_t)gy * W + gx] = (wsum <= 0.f) ? c : vsum / wsum;
    }
}

// ---- median -------------------------------------------------------------------------------------
// Exact median of the (2r+1)^2 window by bisection on the order-preserving integer image of
// the values: the median is the smallest key k with count(window <= k) > n/2.
__global__ __launch_bounds__(256) void median_f32_kernel(const float *__restrict__ img, int H, int W, int r,
                                                         float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem_raw);
    float *ftile = reinterpret_cast<float *>(smem_raw);
    const int LW = FT_TW + 2 * r, LH = FT_TH + 2 * r, K = 2 * r + 1;
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    stage_tile<float, BORDER_EDGE>(img, H, W, r, x0, y0, ftile);
    __syncthreads();
    for (int i = threadIdx.x; i < LW * LH; i += blockDim.x) tile[i] = f32_to_ordered(ftile[i]);
    __syncthreads();
    const int need = (K * K) / 2 + 1;
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < FT_TH; ly += 4) {
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        // np.median (filters.py:105,124) returns NaN for a window that holds a NaN; the keys of NaNs lie
        // above +inf's (0xff800000) or below -inf's (0x007fffff)
        bool has_nan = false;
        for (int dy = 0; dy < K; ++dy)
            for (int dx = 0; dx < K; ++dx) {
                const uint32_t k = tile[(ly + dy) * LW + lx + dx];
                has_nan = has_nan || k > 0xff800000u || k < 0x007fffffu;
            }
        uint32_t lo = 0u, hi = 0xffffffffu;  // invariant: count(<= hi) >= need
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
            for (int dy = 0; dy < K; ++dy)
                for (int dx = 0; dx < K; ++dx) cnt += (tile[(ly + dy) * LW + lx + dx] <= mid) ? 1 : 0;
            if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        out[(size_t)gy * W + gx] = has_nan ? __int_as_float(0x7fc00000) : ordered_to_f32(lo);
    }
}

// uint8 median with replicate border (what cv2.medianBlur computes), then the reference's
// rescale: float32(m) * inv_scale + offset (filters.py:179).
__global__ __launch_bounds__(256) void median_u8_kernel(const uint8_t *__restrict__ img, int H, int W, int r,
                                                        float inv_scale, float offset, int rescale,
                                                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint8_t *tile = smem_raw;
    const int LW = FT_TW + 2 * r, K = 2 * r + 1;
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    stage_tile<uint8_t, BORDER_EDGE>(img, H, W, r, x0, y0, tile);
    __syncthreads();
    const int need = (K * K) / 2 + 1;
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < FT_TH; ly += 4) {
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        int lo = 0, hi = 255;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int dy = 0; dy < K; ++dy)
                for (int dx = 0; dx < K; ++dx) cnt += ((int)tile[(ly + dy) * LW + lx + dx] <= mid) ? 1 : 0;
            if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        float m = (float)lo;
        out[(size_t)gy * W + gx] = rescale ? (m * inv_scale + offset) : m;
    }
}

// min / max / finiteness of a layer, as order-preserving integers (filters.py:141-144)
__global__ void minmax_kernel(const float *__restrict__ img, size_t n, uint32_t *__restrict__ mm /* min,max,nonfinite */) {
    uint32_t lo = 0xffffffffu, hi = 0u, bad = 0u;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = img[i];
        if (!(fabsf(v) <= 3.402823466e38f)) bad = 1u;
        uint32_t k = f32_to_ordered(v);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo2 = __shfl_down(lo, off, 64), hi2 = __shfl_down(hi, off, 64), b2 = __shfl_down(bad, off, 64);
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
        bad |= b2;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
        if (bad) atomicOr(&mm[2], 1u);
    }
}
// mm[2] (non-finite seen) is sticky across layers; the caller clears it once.
__global__ void minmax_init_kernel(uint32_t *mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

// filters.py:148-151: uint8(rint(clip((x - min) * scale, 0, 255))), scale = 255 / (max - min).
// min/max come from the device-side reduction so the layered loop never syncs with the host.
__global__ void quantise_kernel(const float *__restrict__ img, size_t n, const uint32_t *__restrict__ mm,
                                uint8_t *__restrict__ q, float *__restrict__ scale_out /* inv_scale, offset, flat */) {
    const float lo = ordered_to_f32(mm[0]), hi = ordered_to_f32(mm[1]);
    const double dlo = (double)lo, dhi = (double)hi;
    const bool flat = (dhi == dlo);
    const float scale = flat ? 0.f : (float)(255.0 / (dhi - dlo));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale_out[0] = flat ? 0.f : (float)((dhi - dlo) / 255.0);
        scale_out[1] = lo;
        scale_out[2] = flat ? 1.f : 0.f;
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = (img[i] - lo) * scale;
        s = fminf(fmaxf(s, 0.f), 255.f);
        q[i] = flat ? (uint8_t)0 : (uint8_t)rintf(s);
    }
}
// rescale variant reading inv_scale/offset from device memory (constant layer -> full(offset))
__global__ __launch_bounds__(256) void median_u8_dev_scale_kernel(const uint8_t *__restrict__ img, int H, int W, int r,
                                                                  const float *__restrict__ sc, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint8_t *tile = smem_raw;
    const int LW = FT_TW + 2 * r, K = 2 * r + 1;
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    stage_tile<uint8_t, BORDER_EDGE>(img, H, W, r, x0, y0, tile);
    __syncthreads();
    const float inv_scale = sc[0], offset = sc[1];
    const bool flat = sc[2] != 0.f;
    const int need = (K * K) / 2 + 1;
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < FT_TH; ly += 4) {
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        int lo = 0, hi = 255;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int dy = 0; dy < K; ++dy)
                for (int dx = 0; dx < K; ++dx) cnt += ((int)tile[(ly + dy) * LW + lx + dx] <= mid) ? 1 : 0;
            if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        out[(size_t)gy * W + gx] = flat ? offset : ((float)lo * inv_scale + offset);
    }
}

// ---- layered aggregation helpers ----------------------------------------------------------------
// (H, W, D) -> (D, H, W): tiles of 64 pixels x 64 disparities through LDS, coalesced both ways.
__global__ __launch_bounds__(256) void dsi_to_layers_kernel(const float *__restrict__ dsi, size_t npix, int D,
                                                            float *__restrict__ layers) {
    __shared__ float t[64][65];
    const size_t p0 = (size_t)blockIdx.x * 64;
    const int d0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    for (int k = ly0; k < 64; k += 4) {  // k: pixel within tile, lx: disparity
        size_t p = p0 + k;
        int d = d0 + lx;
        t[k][lx] = (p < npix && d < D) ? dsi[p * D + d] : 0.f;
    }
    __syncthreads();
    for (int k = ly0; k < 64; k += 4) {  // k: disparity within tile, lx: pixel
        size_t p = p0 + lx;
        int d = d0 + k;
        if (p < npix && d < D) layers[(size_t)d * npix + p] = t[lx][k];
    }
}

// stereo.py:167-170 on packed keys: strict '<', NaN never wins.
__global__ void wta_update_kernel(const float *__restrict__ layer, size_t n, uint32_t d, uint64_t *__restrict__ keys) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = layer[i];
        if (v == v) {
            uint64_t k = make_key(v, d), old = keys[i];
            if (k < old) keys[i] = k;
        }
    }
}

static inline int ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- host-side filter tables ----------------------------------------------------------------------
int radius_from_sigma(double sigma) {  // filters.py:42-47
    int r = (int)std::ceil(3.0 * sigma);
    return r > 1 ? r : 1;
}

// filters.py:24-39 evaluated in float32 like NumPy does (pairwise sum for the normalisation).
static float np_sum_f32_any(const float *a, int n) {
    if (n < 8) { float r = 0.f; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        float r[8]; int i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum_f32_any(a, n2) + np_sum_f32_any(a + n2, n - n2);
}
void gaussian_kernel_host(int radius, double sigma, float *k) {
    const int size = 2 * radius + 1;
    const float denom = (float)(2.0 * sigma * sigma);
    for (int y = -radius; y <= radius; ++y)
        for (int x = -radius; x <= radius; ++x) {
            float fx = (float)x, fy = (float)y;
            k[(y + radius) * size + (x + radius)] = std::exp(-(fx * fx + fy * fy) / denom);
        }
    float s = np_sum_f32_any(k, size * size);
    for (int i = 0; i < size * size; ++i) k[i] = k[i] / s;
}

constexpr int MAX_FILTER_RADIUS = 32;
size_t filter_kernel_table_bytes(int r) { return sizeof(float) * (size_t)(2 * r + 1) * (2 * r + 1); }

static int check_radius(int r) {
    if (r < 1) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    if (r > MAX_FILTER_RADIUS) { set_error("filter radius %d > %d is not supported on the GPU path", r, MAX_FILTER_RADIUS); return STEREO_EUNSUPPORTED; }
    return STEREO_OK;
}

template <typename K, typename... Args>
static int launch_tile_filter(K kern, int H, int W, size_t smem, hipStream_t s, Args... args) {
    if (smem > 64 * 1024)
        SHP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(ceil_div(W, FT_TW), ceil_div(H, FT_TH));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, args...);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

int launch_gaussian(const float *d_img, int H, int W, int r, const float *d_kern, float *d_out, hipStream_t s) {
    SHP_TRY(check_radius(r));
    size_t smem = sizeof(float) * ((size_t)(FT_TW + 2 * r) * (FT_TH + 2 * r) + (size_t)(2 * r + 1) * (2 * r + 1));
    return launch_tile_filter(gaussian_kernel2d, H, W, smem, s, d_img, H, W, r, d_kern, d_out);
}
int launch_bilateral(const float *d_img, int H, int W, int r, const float *d_kern, double sigma, float *d_out,
                     hipStream_t s) {
    SHP_TRY(check_radius(r));
    const float denom = (float)(2.0 * (double)(sigma * sigma));
    size_t smem = sizeof(float) * ((size_t)(FT_TW + 2 * r) * (FT_TH + 2 * r) + (size_t)(2 * r + 1) * (2 * r + 1));
    return launch_tile_filter(bilateral_kernel2d, H, W, smem, s, d_img, H, W, r, d_kern, denom, d_out);
}
int launch_median_f32(const float *d_img, int H, int W, int r, float *d_out, hipStream_t s) {
    SHP_TRY(check_radius(r));
    size_t smem = sizeof(float) * (size_t)(FT_TW + 2 * r) * (FT_TH + 2 * r);
    return launch_tile_filter(median_f32_kernel, H, W, smem, s, d_img, H, W, r, d_out);
}
int launch_median_u8(const uint8_t *d_img, int H, int W, int r, float inv_scale, float offset, int rescale,
                     float *d_out, hipStream_t s) {
    SHP_TRY(check_radius(r));
    size_t smem = align_up((size_t)(FT_TW + 2 * r) * (FT_TH + 2 * r), 16);
    return launch_tile_filter(median_u8_kernel, H, W, smem, s, d_img, H, W, r, inv_scale, offset, rescale, d_out);
}
// "opencv" median entirely on the device: d_mm (3 x uint32), d_q (H*W bytes), d_sc (3 floats) are scratch.
int launch_median_u8q(const float *d_img, int H, int W, int r, uint32_t *d_mm, uint8_t *d_q, float *d_sc,
                      float *d_out, hipStream_t s) {
    SHP_TRY(check_radius(r));
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, d_mm);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(minmax_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d_img, n, d_mm);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(quantise_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d_img, n, d_mm, d_q, d_sc);
    SHP_LAUNCH_CHECK();
    size_t smem = align_up((size_t)(FT_TW + 2 * r) * (FT_TH + 2 * r), 16);
    return launch_tile_filter(median_u8_dev_scale_kernel, H, W, smem, s, d_q, H, W, r, d_sc, d_out);
}
int launch_dsi_to_layers(const float *d_dsi, size_t npix, int D, float *d_layers, hipStream_t s) {
    dim3 grid((unsigned)((npix + 63) / 64), (unsigned)ceil_div(D, 64));
    hipLaunchKernelGGL(dsi_to_layers_kernel, grid, dim3(256), 0, s, d_dsi, npix, D, d_layers);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_wta_update(const float *d_layer, size_t n, int d, uint64_t *d_keys, hipStream_t s) {
    hipLaunchKernelGGL(wta_update_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d_layer, n, (uint32_t)d, d_keys);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace shp

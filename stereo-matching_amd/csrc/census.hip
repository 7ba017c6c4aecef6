// Weighted census transform: per-pixel packed census words and the (H, W, D) cost volume.
//
// Reference semantics (census.py:107-153, :156-202): bit(idx) = img[y+dy, x+dx] > img[y, x]
// (strict, caller's dtype, 0 when the neighbour is outside the image); the cost of (y, x, d) is
// the weight-sum over offsets whose left/right bits differ, or `large` when x < d or either
// centre pixel is within `radius` of an image border.
//
// Packing: offsets come in 8 directions x radius rings with weight base/2^(r-1).  Direction k
// owns a 4-bit (radius <= 4) or 8-bit (radius <= 8) field whose bit for ring r has place value
// 2^(field_bits - r), so the field of L^R read as an integer IS that direction's weighted
// Hamming cost in units of base/2^(field_bits-1).  The whole cost is then the sum of the
// fields: one v_dot8_u32_u4 (or two v_dot4_u32_u8) per evaluation.
#include "fast_common.hpp"

namespace shp {

// ---- host-side census configuration ---------------------------------------------------------

// NumPy's pairwise float32 sum for n <= 128 (what np.sum(weights) evaluates, census.py:240).
static float np_sum_f32(const float *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    float r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

int make_census_cfg(int radius, double base_weight, CensusCfg *cfg) {
    if (radius <= 0) {
        set_error("radius must be a positive integer");
        return STEREO_EINVAL;
    }
    if (radius > 8) {
        set_error("census radius %d > 8 is not supported by the GPU path", radius);
        return STEREO_EUNSUPPORTED;
    }
    cfg->radius = radius;
    cfg->word64 = radius > 4;
    float all[64];
    for (int r = 1; r <= 8; ++r) cfg->w[r - 1] = (float)(base_weight / std::ldexp(1.0, r - 1));
    for (int k = 0; k < 8; ++k)
        for (int r = 1; r <= radius; ++r) all[k * radius + (r - 1)] = cfg->w[r - 1];
    cfg->large = (float)((double)np_sum_f32(all, 8 * radius) * 10.0);
    const int field_bits = cfg->word64 ? 8 : 4;
    cfg->unit = cfg->w[field_bits - 1];  // base / 2^(field_bits-1), value of the lowest bit
    // The reference adds float32 weights one by one.  Every partial sum is m * unit with
    // m <= 8 * (2^field_bits - 1); all are exact if mantissa_bits(base) + bits(m_max) <= 24.
    float bf = (float)base_weight;
    int exact = 0;
    if (std::isfinite(bf) && bf != 0.f && std::fabs(cfg->unit) >= 1.17549435e-38f) {
        uint32_t bits;
        memcpy(&bits, &bf, 4);
        uint32_t mant = (bits & 0x7fffffu) | 0x800000u;
        int tz = __builtin_ctz(mant);
        int sig_bits = 24 - tz;
        // `large` = 10 * sum(weights) must be an exact multiple of the unit as well:
        // 10 * 120 < 2^11 (4-bit fields), 10 * 2040 < 2^15 (8-bit fields)
        int m_bits = cfg->word64 ? 15 : 11;
        exact = (sig_bits + m_bits <= 24);
    }
    cfg->exact = exact;
    cfg->seq_fma = 1;
    for (int r = 0; r < 8; ++r) {
        cfg->hw[r] = cfg->w[r] * 0.5f;
        if (!std::isfinite(cfg->w[r]) || !(std::fabs(cfg->hw[r]) >= 1.17549435e-38f)) cfg->seq_fma = 0;
    }
    return STEREO_OK;
}

// ---- census packing ---------------------------------------------------------------------------

constexpr int CP_TW = 64;  // tile width  (one wavefront per row)
constexpr int CP_TH = 16;  // tile height

// lowest value of the image type: a neighbour outside the image compares "not greater" (census.py:140-146)
template <typename T> __device__ __forceinline__ T census_lowest();
template <> __device__ __forceinline__ float census_lowest<float>() { return -__builtin_inff(); }
template <> __device__ __forceinline__ double census_lowest<double>() { return -__builtin_inf(); }
template <> __device__ __forceinline__ uint8_t census_lowest<uint8_t>() { return 0; }

// word * 2 + (n > c): the next census bit enters at the low end.
template <typename T, typename WordT>
__device__ __forceinline__ WordT shift_in_greater(WordT word, T n, T c) {
    return (WordT)((word << 1) | (WordT)(n > c ? 1 : 0));
}
// float32 pixels, 32-bit words: compare into the carry, then word + word + carry -- two instructions per
// neighbour where the generic form compiles to three (compare, select 0 / 1, shift-or); the kernel is bound by
// its vector instructions.
template <>
__device__ __forceinline__ uint32_t shift_in_greater<float, uint32_t>(uint32_t word, float n, float c) {
    uint32_t out;
    asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(out) : "v"(n), "v"(c), "v"(word) : "vcc");
    return out;
}

// 64-bit words: the carry ripples through the two halves.
template <>
__device__ __forceinline__ uint64_t shift_in_greater<float, uint64_t>(uint64_t word, float n, float c) {
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32), out_lo, out_hi;
    asm("v_cmp_gt_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %4, %4, vcc\n\tv_addc_co_u32 %1, vcc, %5, %5, vcc"
        : "=&v"(out_lo), "=v"(out_hi) : "v"(n), "v"(c), "v"(lo), "v"(hi) : "vcc");
    return ((uint64_t)out_hi << 32) | out_lo;
}

// One packed census word per pixel.  blockIdx.z selects the image, so the left and the right image
// of a pair are packed by one launch.  The tile and its halo are staged in LDS with out-of-image
// cells set to the lowest value; bits are shifted in from the most significant one down (direction 7
// ring 1 first), so that every neighbour costs one compare and one shift-or.
template <typename T, typename WordT, int R>
__global__ __launch_bounds__(256) void census_pack_kernel(const T *__restrict__ img0, const T *__restrict__ img1,
                                                          int H, int W, WordT *__restrict__ words0,
                                                          WordT *__restrict__ words1) {
    constexpr int LW = CP_TW + 2 * R, LH = CP_TH + 2 * R;
    constexpr int FB = (sizeof(WordT) == 8) ? 8 : 4;
    __shared__ T tile[LH * LW];
    const T *__restrict__ img = blockIdx.z ? img1 : img0;
    WordT *__restrict__ words = blockIdx.z ? words1 : words0;
    const int x0 = blockIdx.x * CP_TW, y0 = blockIdx.y * CP_TH;
    // every load of the tile is issued before the first one is used (addresses clamped into the image, so
    // no branch separates the loads)
    constexpr int NLD = (LW * LH + 255) / 256;
    T v[NLD];
    bool in[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = min((int)threadIdx.x + u * 256, LW * LH - 1);
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        in[u] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[u] = img[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u)
        if ((int)threadIdx.x + u * 256 < LW * LH) tile[threadIdx.x + u * 256] = in[u] ? v[u] : census_lowest<T>();
    __syncthreads();
    constexpr int dys[8] = {1, -1, 0, 0, 1, 1, -1, -1};
    constexpr int dxs[8] = {0, 0, 1, -1, 1, -1, 1, -1};
    const int lx = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < CP_TH / 4; ++i) {
        const int ly = (threadIdx.x >> 6) + 4 * i;
        const int gy = y0 + ly, gx = x0 + lx;
        const T *c0 = tile + (ly + R) * LW + (lx + R);
        const T c = *c0;
        WordT word = 0;
#pragma unroll
        for (int k = 7; k >= 0; --k) {
#pragma unroll
            for (int r = 1; r <= R; ++r) word = shift_in_greater(word, c0[dys[k] * r * LW + dxs[k] * r], c);
            if (FB > R) word <<= (k > 0 ? FB - R : 0);
        }
        // ring r of direction k sits at bit FB * k + (FB - r)
        if (FB > R) word <<= (FB - R);
        if (gy < H && gx < W) words[(size_t)gy * W + gx] = word;
    }
}

// The pre-pass of the fused guided path in one launch: census words of both images and, from the left image's tile
// while it is in LDS, the guide statistics of the guided filter (guided_filter.py:202-216: box mean of the guide and
// 1 / (var + eps32), float64, clipped windows; the largest finite |guide| and a non-finite flag for the kernels'
// scaling).  float32 images; the guided radius may not exceed the census radius (the tile's halo).  Window means
// multiply by the reciprocal of the window area (a compile-time constant for full windows) instead of dividing: one
// float64 division per pixel instead of three, at most one unit in the last place of the mean.
template <typename WordT, int R, int RG>
__global__ __launch_bounds__(256) void census_stats_kernel(const float *__restrict__ img0, const float *__restrict__ img1,
                                                           int H, int W, WordT *__restrict__ words0,
                                                           WordT *__restrict__ words1, double eps32,
                                                           double *__restrict__ mean, double *__restrict__ invden,
                                                           unsigned int *gmax_bits) {
    static_assert(RG >= 1 && RG <= R, "the statistics' windows must fit the census tile's halo");
    constexpr int LW = CP_TW + 2 * R, LH = CP_TH + 2 * R;
    constexpr int FB = (sizeof(WordT) == 8) ? 8 : 4;
    constexpr int SH = CP_TH + 2 * RG;           // rows of horizontal window sums
    __shared__ float tile[LH * LW];               // census: cells outside the image hold the lowest value
    __shared__ float gtile[LH * LW];              // statistics: cells outside the image hold 0 (clipped windows)
    __shared__ double hs1[SH * CP_TW], hs2[SH * CP_TW];
    const bool left = blockIdx.z == 0;
    const float *__restrict__ img = left ? img0 : img1;
    WordT *__restrict__ words = left ? words0 : words1;
    const int x0 = blockIdx.x * CP_TW, y0 = blockIdx.y * CP_TH;
    constexpr int NLD = (LW * LH + 255) / 256;
    float v[NLD];
    bool in[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = min((int)threadIdx.x + u * 256, LW * LH - 1);
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        in[u] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[u] = img[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u)
        if ((int)threadIdx.x + u * 256 < LW * LH) {
            tile[threadIdx.x + u * 256] = in[u] ? v[u] : census_lowest<float>();
            if (left) gtile[threadIdx.x + u * 256] = in[u] ? v[u] : 0.f;
        }
    __syncthreads();
    constexpr int dys[8] = {1, -1, 0, 0, 1, 1, -1, -1};
    constexpr int dxs[8] = {0, 0, 1, -1, 1, -1, 1, -1};
    const int lx = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < CP_TH / 4; ++i) {
        const int ly = (threadIdx.x >> 6) + 4 * i;
        const int gy = y0 + ly, gx = x0 + lx;
        const float *c0 = tile + (ly + R) * LW + (lx + R);
        const float c = *c0;
        WordT word = 0;
#pragma unroll
        for (int k = 7; k >= 0; --k) {
#pragma unroll
            for (int r = 1; r <= R; ++r) word = shift_in_greater(word, c0[dys[k] * r * LW + dxs[k] * r], c);
            if (FB > R) word <<= (k > 0 ? FB - R : 0);
        }
        if (FB > R) word <<= (FB - R);
        if (gy < H && gx < W) words[(size_t)gy * W + gx] = word;
    }
    if (!left) return;
    // horizontal window sums of G and G * G for the rows the tile's vertical windows reach
    for (int i = threadIdx.x; i < SH * CP_TW; i += 256) {
        const int r = i >> 6, cx = i & 63;
        const float *row = gtile + (r + R - RG) * LW + (cx + R - RG);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k <= 2 * RG; ++k) {
            const double g = (double)row[k];
            s1 += g;
            s2 = fma(g, g, s2);
        }
        hs1[i] = s1;
        hs2[i] = s2;
    }
    __syncthreads();
    float amax = 0.f;   // largest finite |guide| of the tile
    bool nonfinite = false;
    constexpr double RA_FULL = 1.0 / (double)((2 * RG + 1) * (2 * RG + 1));
#pragma unroll
    for (int i = 0; i < CP_TH / 4; ++i) {
        const int ly = (threadIdx.x >> 6) + 4 * i;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        const float gv = gtile[(ly + R) * LW + lx + R];
        if (fabsf(gv) <= 3.402823466e38f) amax = fmaxf(amax, fabsf(gv));
        else nonfinite = true;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k <= 2 * RG; ++k) {
            s1 += hs1[(ly + k) * CP_TW + lx];
            s2 += hs2[(ly + k) * CP_TW + lx];
        }
        const int cy = min(gy + RG, H - 1) - max(gy - RG, 0) + 1;
        const int cx = min(gx + RG, W - 1) - max(gx - RG, 0) + 1;
        double m, inv;
        guide_mean_invden(s1, s2, cy, cx, (2 * RG + 1) * (2 * RG + 1), RA_FULL, eps32, &m, &inv);
        mean[(size_t)gy * W + gx] = m;
        invden[(size_t)gy * W + gx] = inv;
    }
    unsigned int b = __float_as_uint(amax);
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = __shfl_down(b, off, 64);
        b = o > b ? o : b;
    }
    // read first: almost every wavefront sees a value already at least as large
    if ((threadIdx.x & 63) == 0 && b > *static_cast<volatile unsigned int *>(gmax_bits)) atomicMax(gmax_bits, b);
    if (nonfinite) gmax_bits[1] = 1u;   // second word: some guide pixel is NaN or infinite
}

bool census_stats_merged(int census_radius, int guided_radius) {
    return (census_radius == 4 || census_radius == 8) && guided_radius >= 1 && guided_radius <= 4;
}

// census words of both float32 images + the left image's guide statistics, one launch (see census_stats_kernel);
// d_gmax: 8 bytes, cleared here
int launch_census_stats(const float *d_left, const float *d_right, int H, int W, int census_radius, int guided_radius,
                        double eps, void *d_lw, void *d_rw, double *d_mean, double *d_invden, void *d_gmax, hipStream_t s) {
    if (H <= 0 || W <= 0) return STEREO_OK;
    if (!census_stats_merged(census_radius, guided_radius)) {
        set_error("no merged pre-pass for census radius %d, guided radius %d", census_radius, guided_radius);
        return STEREO_EUNSUPPORTED;
    }
    unsigned int *gmax_bits = static_cast<unsigned int *>(d_gmax);
    SHP_HIP(hipMemsetAsync(gmax_bits, 0, 8, s));
    const double eps32 = (double)(float)eps;   // guided_filter.py:245 rounds eps to float32
    dim3 grid(ceil_div(W, CP_TW), ceil_div(H, CP_TH), 2);
#define SHP_CS_CASE(WT, RR, GG)                                                                                          \
    if (census_radius == RR && guided_radius == GG)                                                                      \
        hipLaunchKernelGGL((census_stats_kernel<WT, RR, GG>), grid, dim3(256), 0, s, d_left, d_right, H, W,               \
                           static_cast<WT *>(d_lw), static_cast<WT *>(d_rw), eps32, d_mean, d_invden, gmax_bits);
    SHP_CS_CASE(uint32_t, 4, 1) SHP_CS_CASE(uint32_t, 4, 2) SHP_CS_CASE(uint32_t, 4, 3) SHP_CS_CASE(uint32_t, 4, 4)
    SHP_CS_CASE(uint64_t, 8, 1) SHP_CS_CASE(uint64_t, 8, 2) SHP_CS_CASE(uint64_t, 8, 3) SHP_CS_CASE(uint64_t, 8, 4)
#undef SHP_CS_CASE
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

template <typename T, typename WordT>
static int launch_census_pack_r(const T *img0, const T *img1, int H, int W, int R, WordT *w0, WordT *w1, hipStream_t s) {
    dim3 grid(ceil_div(W, CP_TW), ceil_div(H, CP_TH), img1 ? 2 : 1);
#define SHP_CP_CASE(RR) \
    case RR: hipLaunchKernelGGL((census_pack_kernel<T, WordT, RR>), grid, dim3(256), 0, s, img0, img1, H, W, w0, w1); break;
    switch (R) {
        SHP_CP_CASE(1) SHP_CP_CASE(2) SHP_CP_CASE(3) SHP_CP_CASE(4) SHP_CP_CASE(5) SHP_CP_CASE(6) SHP_CP_CASE(7) SHP_CP_CASE(8)
        default: set_error("census radius %d is outside 1..8", R); return STEREO_EUNSUPPORTED;
    }
#undef SHP_CP_CASE
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

template <typename T>
static int launch_census_pack_t(const T *img0, const T *img1, int H, int W, int R, void *w0, void *w1, hipStream_t s) {
    if (R <= 4)
        return launch_census_pack_r<T, uint32_t>(img0, img1, H, W, R, static_cast<uint32_t *>(w0), static_cast<uint32_t *>(w1), s);
    return launch_census_pack_r<T, uint64_t>(img0, img1, H, W, R, static_cast<uint64_t *>(w0), static_cast<uint64_t *>(w1), s);
}

// d_img1 / d_words1 may be null (one image)
int launch_census_pack_pair(const void *d_img0, const void *d_img1, int dtype, int H, int W, int radius, void *d_words0,
                            void *d_words1, hipStream_t s) {
    if (H <= 0 || W <= 0) return STEREO_OK;
    switch (dtype) {
        case STEREO_DT_F32:
            return launch_census_pack_t(static_cast<const float *>(d_img0), static_cast<const float *>(d_img1), H, W, radius,
                                        d_words0, d_words1, s);
        case STEREO_DT_F64:
            return launch_census_pack_t(static_cast<const double *>(d_img0), static_cast<const double *>(d_img1), H, W, radius,
                                        d_words0, d_words1, s);
        case STEREO_DT_U8:
            return launch_census_pack_t(static_cast<const uint8_t *>(d_img0), static_cast<const uint8_t *>(d_img1), H, W,
                                        radius, d_words0, d_words1, s);
        default: set_error("unsupported image dtype %d", dtype); return STEREO_EINVAL;
    }
}

int launch_census_pack(const void *d_img, int dtype, int H, int W, int radius, void *d_words, hipStream_t s) {
    return launch_census_pack_pair(d_img, nullptr, dtype, H, W, radius, d_words, nullptr, s);
}

// ---- materialised cost volume, D innermost (census.py:250) -----------------------------------

constexpr int DSI_XB = 64;  // pixels of one row handled by a block

template <typename WordT>
__global__ __launch_bounds__(256) void dsi_kernel(const WordT *__restrict__ lw, const WordT *__restrict__ rw,
                                                  int H, int W, int D, CensusCfg cfg, float *__restrict__ dsi) {
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * DSI_XB;
    const int npx = min(DSI_XB, W - x0);
    const int R = cfg.radius;
    const bool row_ok = (y >= R) && (y < H - R);
    const WordT *lrow = lw + (size_t)y * W;
    const WordT *rrow = rw + (size_t)y * W;
    float *out = dsi + ((size_t)y * W + x0) * D;
    const int total = npx * D;  // <= 64 * D
    // walk the flat (pixel, d) index with d fastest so that stores are fully coalesced
    const int step_q = 256 / D, step_r = 256 % D;
    int j = threadIdx.x;
    int xl = j / D, d = j - xl * D;
    for (; j < total; j += 256) {
        int x = x0 + xl;
        float c = cfg.large;
        if (row_ok && x >= R && x < W - R && x - d >= R) c = pair_cost<WordT>(lrow[x], rrow[x - d], cfg);
        out[j] = c;
        xl += step_q;
        d += step_r;
        if (d >= D) { d -= D; ++xl; }
    }
}

int launch_dsi(const void *d_lw, const void *d_rw, int H, int W, int dmax, const CensusCfg &cfg, float *d_dsi,
               hipStream_t s) {
    if (H <= 0 || W <= 0) return STEREO_OK;
    if ((long long)DSI_XB * dmax > 0x7fffffffLL) {
        set_error("dmax %d too large", dmax);
        return STEREO_EUNSUPPORTED;
    }
    dim3 grid(ceil_div(W, DSI_XB), H);
    if (cfg.word64)
        hipLaunchKernelGGL(dsi_kernel<uint64_t>, grid, dim3(256), 0, s, static_cast<const uint64_t *>(d_lw),
                           static_cast<const uint64_t *>(d_rw), H, W, dmax, cfg, d_dsi);
    else
        hipLaunchKernelGGL(dsi_kernel<uint32_t>, grid, dim3(256), 0, s, static_cast<const uint32_t *>(d_lw),
                           static_cast<const uint32_t *>(d_rw), H, W, dmax, cfg, d_dsi);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

// Same costs in (D, H, W) order: the layout the layer-at-a-time aggregation reads.
template <typename WordT>
__global__ __launch_bounds__(256) void dsi_layers_kernel(const WordT *__restrict__ lw, const WordT *__restrict__ rw,
                                                         int H, int W, int d_begin, CensusCfg cfg,
                                                         float *__restrict__ layers) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int d = d_begin + blockIdx.z;
    if (x >= W) return;
    const int R = cfg.radius;
    float c = cfg.large;
    if (y >= R && y < H - R && x >= R && x < W - R && x - d >= R)
        c = pair_cost<WordT>(lw[(size_t)y * W + x], rw[(size_t)y * W + x - d], cfg);
    layers[((size_t)blockIdx.z * H + y) * W + x] = c;
}

// layers[k] = cost layer d_begin + k, k in [0, d_end - d_begin)
int launch_dsi_layers(const void *d_lw, const void *d_rw, int H, int W, int d_begin, int d_end, const CensusCfg &cfg,
                      float *d_layers, hipStream_t s) {
    if (H <= 0 || W <= 0 || d_begin >= d_end) return STEREO_OK;
    for (int d0 = d_begin; d0 < d_end; d0 += 32768) {
        int nd = min(32768, d_end - d0);
        dim3 grid(ceil_div(W, 256), H, nd);
        float *out = d_layers + (size_t)(d0 - d_begin) * H * W;
        if (cfg.word64)
            hipLaunchKernelGGL(dsi_layers_kernel<uint64_t>, grid, dim3(256), 0, s, static_cast<const uint64_t *>(d_lw),
                               static_cast<const uint64_t *>(d_rw), H, W, d0, cfg, out);
        else
            hipLaunchKernelGGL(dsi_layers_kernel<uint32_t>, grid, dim3(256), 0, s, static_cast<const uint32_t *>(d_lw),
                               static_cast<const uint32_t *>(d_rw), H, W, d0, cfg, out);
        SHP_LAUNCH_CHECK();
    }
    return STEREO_OK;
}

// ---- packed-key utilities ----------------------------------------------------------------------

__global__ void unpack_keys_kernel(const uint64_t *__restrict__ keys, size_t n, int32_t *__restrict__ disp,
                                   float *__restrict__ cost) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t k = keys[i];
        disp[i] = (int32_t)(uint32_t)k;
        cost[i] = ordered_to_f32((uint32_t)(k >> 32));
    }
}
__global__ void merge_keys_kernel(uint64_t *__restrict__ keys, const uint64_t *__restrict__ other, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t a = keys[i], b = other[i];
        keys[i] = b < a ? b : a;
    }
}
// keys[i] = min over the n_slices equal-length slices laid end to end; the result lands in slice 0
__global__ void merge_key_slices_kernel(uint64_t *__restrict__ slices, size_t chunk, int n_slices) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < chunk; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t m = slices[i];
        for (int j = 1; j < n_slices; ++j) {
            const uint64_t v = slices[(size_t)j * chunk + i];
            m = v < m ? v : m;
        }
        slices[i] = m;
    }
}
__global__ void fill_keys_kernel(uint64_t *__restrict__ keys, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        keys[i] = SHP_KEY_INIT;
}
// uint64 order -> int64 order: flip the top bit (and back).
__global__ void keys_flip_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i] ^ 0x8000000000000000ull;
}

static inline int ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int launch_unpack_keys(const uint64_t *keys, size_t n, int32_t *disp, float *cost, hipStream_t s) {
    if (n == 0) return STEREO_OK;
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(ew_grid(n)), dim3(256), 0, s, keys, n, disp, cost);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_merge_keys(uint64_t *keys, const uint64_t *other, size_t n, hipStream_t s) {
    if (n == 0) return STEREO_OK;
    hipLaunchKernelGGL(merge_keys_kernel, dim3(ew_grid(n)), dim3(256), 0, s, keys, other, n);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_merge_key_slices(uint64_t *slices, size_t chunk, int n_slices, hipStream_t s) {
    if (chunk == 0 || n_slices <= 1) return STEREO_OK;
    hipLaunchKernelGGL(merge_key_slices_kernel, dim3(ew_grid(chunk)), dim3(256), 0, s, slices, chunk, n_slices);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_fill_keys(uint64_t *keys, size_t n, hipStream_t s) {
    if (n == 0) return STEREO_OK;
    hipLaunchKernelGGL(fill_keys_kernel, dim3(ew_grid(n)), dim3(256), 0, s, keys, n);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_keys_to_i64(const uint64_t *keys, int64_t *out, size_t n, hipStream_t s) {
    if (n == 0) return STEREO_OK;
    hipLaunchKernelGGL(keys_flip_kernel, dim3(ew_grid(n)), dim3(256), 0, s, keys, reinterpret_cast<uint64_t *>(out), n);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
int launch_keys_from_i64(const int64_t *in, uint64_t *keys, size_t n, hipStream_t s) {
    if (n == 0) return STEREO_OK;
    hipLaunchKernelGGL(keys_flip_kernel, dim3(ew_grid(n)), dim3(256), 0, s, reinterpret_cast<const uint64_t *>(in), keys, n);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace shp

// Guided-filter aggregation fused with the census cost and the running winner-take-all.
//
// Reference semantics (guided_filter.py:182-253, stereo.py:145-170), all in float64:
//   box(X)  = clipped-window sum of X / clipped-window area
//   mG = box(G), var = max(box(G*G) - mG^2, 0), den = var + float32(eps)      (once per scene)
//   per layer p:  mP = box(p), mGP = box(G*p), a = (mGP - mG*mP)/den, b = mP - a*mG
//                 q = float32(box(a)*G + box(b));  WTA: q < min  (strict, ties keep lower d)
// The reference gets its box sums from global float64 integral images; here they are
// tile-local float64 sliding sums (the integer cost sums are exact), so no layer of the
// (H, W, D) volume is ever written to memory.
//
// Work decomposition of the fused kernel: a workgroup owns a TH x (64 - 4r) pixel tile and all
// disparities of its slice; each wavefront takes every NW-th disparity and marches down the
// tile rows with one lane per column (2r halo columns on each side):
//   row i:  cost p(i)                                  -> vertical running sums C1 = sum p, C2 = sum G*p
//           7-tap horizontal sums across lanes (LDS)   -> a, b at row i - r
//           vertical running sums A = sum a, B = sum b -> 7-tap horizontal sums -> q at row i - 2r
//           q -> packed (cost, d) key, ds_min_u64 into the tile's key array in LDS.
// The per-pixel constants of the tile (census words, guide, mG, 1/den) live in LDS for the whole
// disparity loop; the values leaving a sliding window are kept in registers (the row loop is
// unrolled by the window length so the ring index is static).
#include <type_traits>

#include "fast_common.hpp"

namespace shp {

// ---- per-scene guide statistics -----------------------------------------------------------------

constexpr int GS_TW = 64, GS_TH = 16;

// RT: the radius at compile time (window loops unrolled, tile pitch a constant), 0 = use the argument
template <typename GT, int RT>
__global__ __launch_bounds__(256) void guide_stats_kernel(const GT *__restrict__ guide, int H, int W, int r_arg,
                                                          double eps32, double *__restrict__ mean,
                                                          double *__restrict__ invden, unsigned int *gmax_bits) {
    const int R = RT > 0 ? RT : r_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *tile = reinterpret_cast<double *>(smem_raw);  // [LH][LW] guide, 0 outside the image
    const int LW = GS_TW + 2 * R, LH = GS_TH + 2 * R;
    double *hs1 = tile + LW * LH;          // [LH][GS_TW] horizontal sums of G
    double *hs2 = hs1 + LH * GS_TW;        // [LH][GS_TW] horizontal sums of G*G
    const int x0 = blockIdx.x * GS_TW, y0 = blockIdx.y * GS_TH;
    // four loads per thread are in flight at a time (addresses clamped into the image, so no branch
    // separates them); cells outside the image are stored as 0
    for (int i0 = threadIdx.x; i0 < LW * LH; i0 += 4 * 256) {
        GT v[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, LW * LH - 1);
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = y0 - R + ly, gx = x0 - R + lx;
            in[u] = gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[u] = guide[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 256 < LW * LH) tile[i0 + u * 256] = in[u] ? (double)v[u] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LH * GS_TW; i += blockDim.x) {
        int ly = i >> 6, lx = i & 63;
        double s1 = 0.0, s2 = 0.0;
        _Pragma("unroll") for (int k = 0; k <= 2 * R; ++k) {
            double g = tile[ly * LW + lx + k];
            s1 += g;
            s2 = fma(g, g, s2);
        }
        hs1[i] = s1;
        hs2[i] = s2;
    }
    __syncthreads();
    float amax = 0.f;  // max finite |guide| over the tile, as float32
    bool nonfinite = false;
    for (int i = threadIdx.x; i < GS_TH * GS_TW; i += blockDim.x) {
        int ly = i >> 6, lx = i & 63;
        int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        const double gv = tile[(ly + R) * LW + lx + R];
        const float ag = fabsf((float)gv);
        if (fabs(gv) <= 1.7976931348623157e308) amax = fmaxf(amax, fminf(ag, 3.402823466e38f));
        else nonfinite = true;
        double s1 = 0.0, s2 = 0.0;
        _Pragma("unroll") for (int k = 0; k <= 2 * R; ++k) {
            s1 += hs1[(ly + k) * GS_TW + lx];
            s2 += hs2[(ly + k) * GS_TW + lx];
        }
        int cy = min(gy + R, H - 1) - max(gy - R, 0) + 1;
        int cx = min(gx + R, W - 1) - max(gx - R, 0) + 1;
        const int full_area = (2 * R + 1) * (2 * R + 1);
        double m, inv;
        guide_mean_invden(s1, s2, cy, cx, full_area, 1.0 / (double)full_area, eps32, &m, &inv);
        mean[(size_t)gy * W + gx] = m;
        invden[(size_t)gy * W + gx] = inv;
    }
    if (gmax_bits) {
        unsigned int b = __float_as_uint(amax);
        for (int off = 32; off > 0; off >>= 1) {
            unsigned int o = __shfl_down(b, off, 64);
            b = o > b ? o : b;
        }
        // read first: almost every wavefront sees a value already at least as large
        if ((threadIdx.x & 63) == 0 && b > *static_cast<volatile unsigned int *>(gmax_bits)) atomicMax(gmax_bits, b);
        if (nonfinite) gmax_bits[1] = 1u;   // second word: some guide pixel is NaN or infinite
    }
}

size_t guide_stats_scratch_bytes(int, int, int) { return 0; }

int launch_guide_stats(const void *d_guide, int guide_dtype, int H, int W, int radius, double eps,
                       double *d_mean, double *d_invden, void *d_gmax, hipStream_t s) {
    if (H <= 0 || W <= 0) return STEREO_OK;
    unsigned int *gmax_bits = static_cast<unsigned int *>(d_gmax);
    if (gmax_bits) SHP_HIP(hipMemsetAsync(gmax_bits, 0, 8, s));
    if (radius < 1 || radius > 16) {
        set_error("guide statistics kernel supports radius 1..16, got %d", radius);
        return STEREO_EUNSUPPORTED;
    }
    const double eps32 = (double)(float)eps;  // guided_filter.py:245 rounds eps to float32
    dim3 grid(ceil_div(W, GS_TW), ceil_div(H, GS_TH));
    const int LW = GS_TW + 2 * radius, LH = GS_TH + 2 * radius;
    size_t smem = sizeof(double) * ((size_t)LW * LH + 2 * (size_t)LH * GS_TW);
    if (guide_dtype == STEREO_DT_F32) {
        auto kern = radius == 3 ? guide_stats_kernel<float, 3> : radius == 2 ? guide_stats_kernel<float, 2>
                    : radius == 4 ? guide_stats_kernel<float, 4> : guide_stats_kernel<float, 0>;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s,
                           static_cast<const float *>(d_guide), H, W, radius, eps32, d_mean, d_invden, gmax_bits);
    } else if (guide_dtype == STEREO_DT_F64)
        hipLaunchKernelGGL((guide_stats_kernel<double, 0>), grid, dim3(256), smem, s,
                           static_cast<const double *>(d_guide), H, W, radius, eps32, d_mean, d_invden, gmax_bits);
    else {
        set_error("guide dtype must be float32 or float64");
        return STEREO_EINVAL;
    }
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

// ---- fused cost + guided filter + WTA -------------------------------------------------------------

enum { SRC_CENSUS32 = 0, SRC_CENSUS64 = 1, SRC_DSI = 2 };

struct FusedArgs {
    const void *lw;        // left census words   (census sources)
    const void *rw;        // right census words
    const float *dsi;      // [H][W][D] cost volume (SRC_DSI)
    const void *guide;     // [H][W] float or double
    const double *mean;    // [H][W] box mean of the guide
    const double *invden;  // [H][W] 1 / (var + eps32)
    uint64_t *keys;        // [H][W] packed (cost, d)
    float *q_out;          // optional [D][H][W] filtered layers (nullptr: not written)
    CensusCfg cfg;
    int H, W, D;
    int d_begin, d_end;
    int atomic_out;        // 1: merge into keys with atomicMin, 0: plain store
    // set when a fast kernel was launched ahead with the opposite test: run only if it declined
    const float *gate_gmax;
    float gate_eps, gate_min;
    // set when the two-column kernel was launched ahead on cost codes of the volume: run only if it declined
    const uint32_t *gate_code_stats;
};

// wavefront-local ordering of LDS traffic: lanes exchange values through LDS without a
// workgroup barrier; the fence keeps the compiler from moving the reads above the writes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int RG, int SRC, bool INT_UNITS, typename GT, int TH, int NW>
__global__ __launch_bounds__(NW * 64) void fused_guided_kernel(FusedArgs a) {
    constexpr int WIN = 2 * RG + 1;   // box window
    constexpr int HALO = 2 * RG;      // halo columns/rows on each side of the tile
    constexpr int NI = TH + 4 * RG;   // input (cost) rows per tile
    constexpr int NJ = TH + 2 * RG;   // rows of (a, b)
    constexpr int TWV = 64 - 4 * RG;  // output columns per tile
    constexpr int EXN = 64 + 2 * RG;  // exchange row length
    using WordT = typename std::conditional<SRC == SRC_CENSUS64, uint64_t, uint32_t>::type;
    using PT = typename std::conditional<INT_UNITS, uint32_t, double>::type;  // cost sums
    static_assert(SRC != SRC_DSI || !INT_UNITS, "a materialised volume holds arbitrary floats");

    if (a.gate_gmax) {
        const float g = *a.gate_gmax;
        if (a.gate_eps >= a.gate_min * fmaxf(1.f, g * g)) return;   // the fast kernel took the frame
    }
    if (a.gate_code_stats && pair_codes_accepted(a.gate_code_stats, RG)) return;   // the two-column kernel took the volume
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // carve (all offsets multiples of 16)
    unsigned char *sp = smem_raw;
    double *mG_s = reinterpret_cast<double *>(sp); sp += sizeof(double) * NJ * 64;
    double *inv_s = reinterpret_cast<double *>(sp); sp += sizeof(double) * NJ * 64;
    uint64_t *key_s = reinterpret_cast<uint64_t *>(sp); sp += sizeof(uint64_t) * TH * 64;
    double *invtab = reinterpret_cast<double *>(sp); sp += sizeof(double) * 16 * ((WIN + 1 + 15) / 16);
    GT *G_s = reinterpret_cast<GT *>(sp); sp += sizeof(GT) * NI * 64;
    WordT *L_s = reinterpret_cast<WordT *>(sp);
    if (SRC != SRC_DSI) sp += sizeof(WordT) * NI * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-wave exchange rows
    constexpr int EX_BYTES = ((EXN * (int)(3 * sizeof(double) + sizeof(PT))) + 15) / 16 * 16;
    unsigned char *exb = sp + (size_t)wave * EX_BYTES;
    double *exC2 = reinterpret_cast<double *>(exb);
    double *exA = exC2 + EXN;
    double *exB = exA + EXN;
    PT *exC1 = reinterpret_cast<PT *>(exB + EXN);

    const int H = a.H, W = a.W, D = a.D;
    const int tx0 = blockIdx.x * TWV, ty0 = blockIdx.y * TH;
    const GT *guide = static_cast<const GT *>(a.guide);
    const WordT *lw = static_cast<const WordT *>(a.lw);
    const WordT *rw = static_cast<const WordT *>(a.rw);

    // ---- stage the tile constants --------------------------------------------------------
    for (int idx = threadIdx.x; idx < NI * 64; idx += NW * 64) {
        int i = idx >> 6, l = idx & 63;
        int gy = ty0 - HALO + i, gx = tx0 - HALO + l;
        bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        G_s[idx] = in ? guide[(size_t)gy * W + gx] : GT(0);
        if (SRC != SRC_DSI) L_s[idx] = in ? lw[(size_t)gy * W + gx] : WordT(0);
    }
    for (int idx = threadIdx.x; idx < NJ * 64; idx += NW * 64) {
        int j = idx >> 6, l = idx & 63;
        int gy = ty0 - RG + j, gx = tx0 - HALO + l;
        bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        mG_s[idx] = in ? a.mean[(size_t)gy * W + gx] : 0.0;
        inv_s[idx] = in ? a.invden[(size_t)gy * W + gx] : 0.0;
    }
    for (int idx = threadIdx.x; idx < TH * 64; idx += NW * 64) key_s[idx] = SHP_KEY_INIT;
    if (threadIdx.x <= WIN) invtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;
    for (int idx = lane; idx < EXN; idx += 64) {
        exC2[idx] = 0.0; exA[idx] = 0.0; exB[idx] = 0.0; exC1[idx] = PT(0);
    }
    __syncthreads();

    const int x = tx0 - HALO + lane;  // image column of this lane
    const bool x_in = x >= 0 && x < W;
    const int cx = x_in ? (min(x + RG, W - 1) - max(x - RG, 0) + 1) : 0;
    const double invAX = invtab[cx];
    const int R = a.cfg.radius;
    const bool x_valid = x_in && x >= R && x < W - R;  // census centre valid in x (left image)
    const bool out_lane = lane >= HALO && lane < 64 - HALO && x < W;
    const double unit = INT_UNITS ? (double)a.cfg.unit : 1.0;
    const PT large_p = INT_UNITS ? PT(a.cfg.large / a.cfg.unit) : PT(a.cfg.large);

    for (int d = a.d_begin + wave; d < a.d_end; d += NW) {
        PT C1 = PT(0);
        double C2 = 0.0, A = 0.0, B = 0.0;
        PT pring[WIN];
        double gpring[WIN], aring[WIN], bring[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) { pring[k] = PT(0); gpring[k] = 0.0; aring[k] = 0.0; bring[k] = 0.0; }
        const bool xd_valid = x_valid && (x - d >= R);

        for (int base = 0; base < NI; base += WIN) {
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const int i = base + k;
                if (i < NI) {
                    const int yi = ty0 - HALO + i;
                    const bool row_in = yi >= 0 && yi < H;
                    // 1. raw cost of (yi, x, d); pixels outside the image contribute nothing
                    PT p = PT(0);
                    double gp = 0.0;
                    if (row_in && x_in) {
                        if (SRC == SRC_DSI) {
                            p = (PT)a.dsi[((size_t)yi * W + x) * D + d];
                        } else {
                            const bool valid = xd_valid && yi >= R && yi < H - R;
                            p = large_p;
                            if (valid) {
                                WordT xw = L_s[i * 64 + lane] ^ rw[(size_t)yi * W + (x - d)];
                                if (INT_UNITS) p = (PT)field_sum(xw);
                                else p = (PT)sequential_cost<WordT>(xw, a.cfg);
                            }
                        }
                        gp = (double)G_s[i * 64 + lane] * (double)p;
                    }
                    // 2. vertical running sums over rows [i - 2r, i]
                    C1 += p - pring[k];
                    C2 += gp - gpring[k];
                    pring[k] = p;
                    gpring[k] = gp;
                    if (i >= HALO) {
                        // 3. horizontal sums across lanes -> box sums centred on row i - r
                        wave_lds_sync();
                        exC1[RG + lane] = C1;
                        exC2[RG + lane] = C2;
                        wave_lds_sync();
                        PT S1 = PT(0);
                        double S2 = 0.0;
#pragma unroll
                        for (int t = 0; t < WIN; ++t) { S1 += exC1[lane + t]; S2 += exC2[lane + t]; }
                        // 4. a, b at stage-1 row j (image row ty0 - r + j)
                        const int j = i - HALO;
                        const int ym = ty0 - RG + j;
                        double av = 0.0, bv = 0.0;
                        if (ym >= 0 && ym < H && x_in) {
                            const int cy = min(ym + RG, H - 1) - max(ym - RG, 0) + 1;
                            const double inv_area = invtab[cy] * invAX;
                            const double mP = (double)S1 * inv_area;
                            const double mGP = S2 * inv_area;
                            const double mG = mG_s[j * 64 + lane];
                            const double cov = mGP - mG * mP;
                            av = cov * inv_s[j * 64 + lane];
                            bv = mP - av * mG;
                        }
                        // 5. vertical running sums over stage-1 rows [j - 2r, j]
                        const int k2 = (k + 1) % WIN;  // (i - 2r) mod WIN
                        A += av - aring[k2];
                        B += bv - bring[k2];
                        aring[k2] = av;
                        bring[k2] = bv;
                        if (i >= 2 * HALO) {
                            // 6. horizontal sums -> box(a), box(b) centred on output row o
                            wave_lds_sync();
                            exA[RG + lane] = A;
                            exB[RG + lane] = B;
                            wave_lds_sync();
                            double SA = 0.0, SB = 0.0;
#pragma unroll
                            for (int t = 0; t < WIN; ++t) { SA += exA[lane + t]; SB += exB[lane + t]; }
                            // 7. q and the running WTA
                            const int o = i - 2 * HALO;
                            const int yo = ty0 + o;
                            if (out_lane && yo < H) {
                                const int cy = min(yo + RG, H - 1) - max(yo - RG, 0) + 1;
                                const double inv_area = invtab[cy] * invAX;
                                const double g = (double)G_s[(o + HALO) * 64 + lane];
                                const double q = (SA * g + SB) * inv_area * unit;
                                const float qf = (float)q;
                                if (a.q_out) a.q_out[((size_t)d * H + yo) * W + x] = qf;
                                if (qf == qf) atomicMin(reinterpret_cast<unsigned long long *>(&key_s[o * 64 + lane]),
                                                        (unsigned long long)make_key(qf, (uint32_t)d));
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- write the tile's keys -------------------------------------------------------------
    for (int idx = threadIdx.x; idx < TH * 64; idx += NW * 64) {
        int o = idx >> 6, l = idx & 63;
        int gy = ty0 + o, gx = tx0 - HALO + l;
        if (l >= HALO && l < 64 - HALO && gy < H && gx < W) {
            uint64_t key = key_s[idx];
            if (a.atomic_out) atomicMin(reinterpret_cast<unsigned long long *>(&a.keys[(size_t)gy * W + gx]),
                                        (unsigned long long)key);
            else a.keys[(size_t)gy * W + gx] = key;
        }
    }
}

constexpr int FG_TH = 32;
constexpr int FG_NW = 16;

// Radii above 4 keep rings of up to 17 rows per lane: they run eight wavefronts per workgroup (two per SIMD, 256
// registers each) instead of sixteen.
template <int RG>
struct FusedGeometry { static constexpr int NW = RG <= 4 ? FG_NW : 8; };

template <int RG, int SRC, bool INT_UNITS, typename GT>
static int launch_fused_t(const FusedArgs &a, hipStream_t s) {
    constexpr int FG_NW = FusedGeometry<RG>::NW;
    constexpr int WIN = 2 * RG + 1, NI = FG_TH + 4 * RG, NJ = FG_TH + 2 * RG, TWV = 64 - 4 * RG, EXN = 64 + 2 * RG;
    using WordT = typename std::conditional<SRC == SRC_CENSUS64, uint64_t, uint32_t>::type;
    using PT = typename std::conditional<INT_UNITS, uint32_t, double>::type;
    constexpr int EX_BYTES = ((EXN * (int)(3 * sizeof(double) + sizeof(PT))) + 15) / 16 * 16;
    size_t smem = sizeof(double) * NJ * 64 * 2 + sizeof(uint64_t) * FG_TH * 64 +
                  sizeof(double) * 16 * ((WIN + 1 + 15) / 16) + sizeof(GT) * NI * 64 +
                  (SRC != SRC_DSI ? sizeof(WordT) * NI * 64 : 0) + (size_t)FG_NW * EX_BYTES;
    auto kern = fused_guided_kernel<RG, SRC, INT_UNITS, GT, FG_TH, FG_NW>;
    SHP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    dim3 grid(ceil_div(a.W, TWV), ceil_div(a.H, FG_TH));
    hipLaunchKernelGGL(kern, grid, dim3(FG_NW * 64), smem, s, a);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

template <int RG>
static int launch_fused_rg(const FusedArgs &a, int src, bool int_units, int guide_dtype, hipStream_t s) {
    if (src == SRC_DSI) {
        if (guide_dtype == STEREO_DT_F32) return launch_fused_t<RG, SRC_DSI, false, float>(a, s);
        return launch_fused_t<RG, SRC_DSI, false, double>(a, s);
    }
    if (guide_dtype != STEREO_DT_F32) {
        set_error("the census-fused kernel takes a float32 guide");
        return STEREO_EINVAL;
    }
    if (src == SRC_CENSUS32) {
        if (int_units) return launch_fused_t<RG, SRC_CENSUS32, true, float>(a, s);
        return launch_fused_t<RG, SRC_CENSUS32, false, float>(a, s);
    }
    if (int_units) return launch_fused_t<RG, SRC_CENSUS64, true, float>(a, s);
    return launch_fused_t<RG, SRC_CENSUS64, false, float>(a, s);
}

bool fused_guided_supported(int guided_radius) { return guided_radius >= 1 && guided_radius <= 8; }

int launch_fused_guided(const void *d_lw, const void *d_rw, const CensusCfg *cfg, const float *d_dsi,
                        const void *d_guide, int guide_dtype, const double *d_mean, const double *d_invden,
                        int H, int W, int dmax, int guided_radius, int d_begin, int d_end, uint64_t *d_keys,
                        float *d_q_out, int atomic_out, hipStream_t s, const float *gate_gmax, float gate_eps,
                        float gate_min, const uint32_t *gate_code_stats) {
    if (H <= 0 || W <= 0 || d_begin >= d_end) return STEREO_OK;
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    a.gate_gmax = gate_gmax; a.gate_eps = gate_eps; a.gate_min = gate_min; a.gate_code_stats = gate_code_stats;
    a.lw = d_lw; a.rw = d_rw; a.dsi = d_dsi; a.guide = d_guide; a.mean = d_mean; a.invden = d_invden;
    a.keys = d_keys; a.q_out = d_q_out;
    if (cfg) a.cfg = *cfg;
    a.H = H; a.W = W; a.D = dmax; a.d_begin = d_begin; a.d_end = d_end; a.atomic_out = atomic_out;
    int src = d_dsi ? SRC_DSI : (cfg && cfg->word64 ? SRC_CENSUS64 : SRC_CENSUS32);
    bool int_units = (src != SRC_DSI) && cfg && cfg->exact;
    switch (guided_radius) {
        case 1: return launch_fused_rg<1>(a, src, int_units, guide_dtype, s);
        case 2: return launch_fused_rg<2>(a, src, int_units, guide_dtype, s);
        case 3: return launch_fused_rg<3>(a, src, int_units, guide_dtype, s);
        case 4: return launch_fused_rg<4>(a, src, int_units, guide_dtype, s);
        case 5: return launch_fused_rg<5>(a, src, int_units, guide_dtype, s);
        case 6: return launch_fused_rg<6>(a, src, int_units, guide_dtype, s);
        case 7: return launch_fused_rg<7>(a, src, int_units, guide_dtype, s);
        case 8: return launch_fused_rg<8>(a, src, int_units, guide_dtype, s);
        default:
            set_error("fused guided kernel supports radius 1..8, got %d", guided_radius);
            return STEREO_EUNSUPPORTED;
    }
}

}  // namespace shp

// Fast path of the fused census-cost + guided-filter + WTA kernel (float32 guide, exactly
// summable census weights, guided radius 1..4 and 6..8; the two-column kernel serves 1..5 by default).  Same decomposition as guided.hip -- a
// workgroup owns a TH x (64 - 4r) tile, each wavefront marches down the tile rows for one
// disparity at a time with one lane per column -- but the horizontal box sums never touch LDS:
//
//  * Stage 1 (box sums of p and G*p) is exact 32-bit arithmetic.  p is an integer number of
//    cost units; G is split as k * 2^-s + g_lo with k a 16-bit integer chosen so that the
//    window sum of k*p fits an int32, and g_lo = G - k*2^-s (exact in float32, |g_lo| <= 2^-s-1).
//    sum(G*p) = 2^-s * sum(k*p) [exact integers] + sum(g_lo*p) [tiny float32 terms].
//    The three 32-bit running sums are summed across 2r+1 neighbouring lanes with a chain of
//    v_add_*_dpp wave_shr:1 instructions (s_t = v + shift(s_{t-1})).
//  * a, b and their vertical running sums stay float64 as in the reference.
//  * Stage 2 (box sums of a and b) avoids summing the ill-conditioned pair (sum a, sum b) across
//    lanes.  With A_j, B_j the vertical sums of column j and G_j the guide of column j in the
//    output row:  sum_j (A_j*G(x) + B_j) = sum_j E_j + sum_j A_j*(G(x) - G_j),  E_j = A_j*G_j + B_j.
//    E_j is the (7-row) filter output column j would get from its own guide value: no
//    cancellation, so a float32 lane chain suffices.  The second sum is evaluated term by term
//    (A_j shifted across lanes, guide differences from LDS); each term is small and
//    well-conditioned, so float32 suffices there as well.  The chains run the other way
//    (wave_shl:1) so that outputs land back on their own lane.
//  * The final scaling by 1/area is done in float64 so that a constant window returns the
//    constant exactly (ties between all-`large` layers must stay ties).
#include <cstdlib>
#include <type_traits>

#include "fast_common.hpp"

namespace shp {

// LDS image of one tile (pointers into the dynamic shared segment)
template <typename WordT>
struct TileLds {
    double *mG;      // [NJ][64] box mean of the guide
    uint64_t *key;   // [TH][64] packed (cost, d); every lane owns a slot, only the inner ones are written out
    double *iay1;    // [NJ] 1 / (rows of the clipped window), 0 outside the image
    float *cy2;      // [TH] rows of the clipped window of the output rows (0 outside the image)
    double *invtab;  // [32] 1/n
    WordT *L;        // [NI][64] left census words
    float *inv;      // [NJ][64] 1 / ((var + eps) * window area)
    float *glo;      // [NI][64] guide - k * 2^-s
    float *gout;     // [TH][64] (+pad) guide at the output pixels
    int *next_d;     // [0] next disparity to hand out, [1] lowest disparity whose costs are all `large`
    short *k;        // [NI][64] round(guide * 2^s)
};

// One disparity of one tile.  INTERIOR: every row the tile touches is inside the image and has a
// valid census centre, so the per-row tests vanish.
template <int RG, bool WORD64, int TH, bool QOUT, bool INTERIOR>
__device__ __forceinline__ void fast_disparity_pass(
    const FastArgs &a, const TileLds<typename std::conditional<WORD64, uint64_t, uint32_t>::type> &t, const int d,
    const int lane, const int x, const int ty0, const bool x_in, const bool x_valid, const double invAX1,
    const float cxf, const int lane1, const double inv_scale_d) {
    constexpr int WIN = 2 * RG + 1;
    constexpr int HALO = 2 * RG;
    constexpr int NI = TH + 4 * RG;
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;
    const int H = a.H, W = a.W, R = a.radius;
    const WordT *rw = static_cast<const WordT *>(a.rw);
    const uint32_t large_lane = x_in ? a.large_units : 0u;  // cost of an invalid pixel; 0 outside the image
    // tile-row ranges: census centre valid for iv_lo <= i < iv_hi, row inside the image for ii_lo <= i < ii_hi
    const int iv_lo = max(R - (ty0 - HALO), 0), iv_hi = min(H - R - (ty0 - HALO), NI);
    const int ii_lo = max(-(ty0 - HALO), 0), ii_hi = min(H - (ty0 - HALO), NI);

    uint32_t C1 = 0u;
    int I2 = 0;
    float F2 = 0.f;
    double A = 0.0, B = 0.0;
    uint32_t pr[WIN];
    int kr[WIN];
    float gr[WIN];
    double ar[WIN], br[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) { pr[k] = 0u; kr[k] = 0; gr[k] = 0.f; ar[k] = 0.0; br[k] = 0.0; }
    const bool xd_valid = x_valid && (x - d >= R);
    // Right census words: lanes without a valid match read a clamped in-range column (value unused).
    const int xr = min(max(x - d, 0), W - 1);
    const WordT *rp = rw + xr;
    auto row_of = [&](int i) -> int {  // image row of tile row i, clamped into the image on border tiles
        int y = ty0 - HALO + i;
        return INTERIOR ? y : min(max(y, 0), H - 1);
    };

    const double ia_interior = invAX1 * (1.0 / (double)WIN);
    const float area_interior = cxf * (float)WIN;
    const float rcp_interior = __builtin_amdgcn_rcpf(area_interior);
    // One group of WIN consecutive tile rows.  STEADY groups start at base >= 4r, where both box
    // stages are live for every row; the first two groups resolve their stage tests at compile
    // time (base is then the literal 0 or WIN), so no group has a branch in its body.
    // The operands a row needs first are fetched one row ahead; the head of the next row is carried
    // from one row group into the next so that no group starts by waiting for its own loads.
    struct RowHead {
        WordT L, R; int k; float glo;
    };
    RowHead cur;
    bool have_head = false;
    auto row_group = [&](const int base, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        // lane pointers of this group: the per-row accesses below are then immediate offsets
        LDS_AS const WordT *Lg = lds_lane_ptr(t.L + base * 64 + lane);
        LDS_AS const short *kg = lds_lane_ptr(t.k + base * 64 + lane);
        LDS_AS const float *glog = lds_lane_ptr(t.glo + base * 64 + lane);
        LDS_AS const double *mGg = lds_lane_ptr(t.mG + (base - HALO) * 64 + lane1);
        LDS_AS const float *invg = lds_lane_ptr(t.inv + (base - HALO) * 64 + lane1);
        LDS_AS const float *goutg = lds_lane_ptr(t.gout + (base - 2 * HALO) * 64 + lane);
        LDS_AS uint64_t *keyg = lds_lane_ptr(t.key + (base - 2 * HALO) * 64 + lane);
        LDS_AS const double *iay1g = lds_lane_ptr(t.iay1 + (base - HALO));
        LDS_AS const float *cy2g = lds_lane_ptr(t.cy2 + (base - 2 * HALO));
        // The operands a row needs first (census words, guide split) are fetched one row ahead;
        // the ones it needs late (guide mean, 1/den, guide of the output row) at the start of the
        // row itself, so that every LDS / L2 latency overlaps arithmetic.
        auto fetch_head = [&](int k) {
            RowHead r;
            r.L = Lg[k * 64];
            r.k = (int)kg[k * 64];
            r.glo = glog[k * 64];
            r.R = rp[(size_t)row_of(min(base + k, NI - 1)) * W];
            return r;
        };
        if (!have_head) { cur = fetch_head(0); have_head = true; }
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const int i = base + k;
            const RowHead nxt = fetch_head(k + 1);
            const double mG = mGg[k * 64];
            const float inv_f = invg[k * 64];
            float g[WIN];
#pragma unroll
            for (int u = 0; u < WIN; ++u) g[u] = goutg[k * 64 + u];
            // the row leaving the window [i - 2r, i] retires first, so that its ring registers are free
            // before the new row's values are formed (no register copies at the loop edge)
            C1 -= pr[k];
            I2 -= kr[k];
            F2 -= gr[k];
            // 1. cost in units; k*p and g_lo*p
            const uint32_t fs = field_sum((WordT)(cur.L ^ cur.R));
            uint32_t p;
            if (INTERIOR) {
                p = xd_valid ? fs : large_lane;
            } else {
                const bool rowv = i >= iv_lo && i < iv_hi;
                const bool rowi = i >= ii_lo && i < ii_hi;
                p = (rowv && xd_valid) ? fs : large_lane;
                p = rowi ? p : 0u;
            }
            const int kp = __mul24(cur.k, (int)p);
            const float gl = cur.glo * (float)p;
            // 2. vertical running sums over rows [i - 2r, i]
            C1 += p;
            I2 += kp;
            F2 += gl;
            pr[k] = p; kr[k] = kp; gr[k] = gl;
            if (STEADY || i >= HALO) {
                // 3. sums over lanes [l - 2r, l]: the box sums of column x - r
                uint32_t s1 = C1;
                int si = I2;
                float sf = F2;
#pragma unroll
                for (int u = 0; u < 2 * RG; ++u) {
                    s1 = C1 + (uint32_t)dpp_i<DPP_WAVE_SHR1>((int)s1);
                    si = I2 + dpp_i<DPP_WAVE_SHR1>(si);
                    sf = F2 + dpp_f<DPP_WAVE_SHR1>(sf);
                }
                // 4. a, b at (row ty0 - r + j, column x - r); 1/area is 0 outside the image
                //    a = (sum(G p) - mG * sum(p)) / (area * den),  b = sum(p) / area - a * mG
                const double ia = INTERIOR ? ia_interior : iay1g[k] * invAX1;
                const double s1d = (double)s1;
                const double u = fma(-mG, s1d, fma((double)si, inv_scale_d, (double)sf));
                const double av = u * (double)inv_f;
                const double bv = fma(-av, mG, s1d * ia);
                // 5. vertical running sums over stage-1 rows [j - 2r, j]
                const int k2 = (k + 1) % WIN;
                A = (A - ar[k2]) + av;
                B = (B - br[k2]) + bv;
                ar[k2] = av;
                br[k2] = bv;
                if (STEADY || i >= 2 * HALO) {
                    // 6. sums over lanes [l, l + 2r] (stage-1 columns x - r .. x + r)
                    const int o = i - 2 * HALO;
                    const float gx = g[0];
                    const float ef = (float)fma(A, (double)gx, B);  // column's own-guide output (7 rows)
                    float ash = (float)A;
                    float se = ef, dsum = 0.f;
#pragma unroll
                    for (int u2 = 1; u2 <= 2 * RG; ++u2) {
                        se = ef + dpp_f<DPP_WAVE_SHL1>(se);
                        ash = dpp_mov_f<DPP_WAVE_SHL1>(ash);          // A of column x + u
                        dsum = fmaf(ash, gx - g[u2], dsum);           // A_j * (G(x) - G_j)
                    }
                    // 7. q and the running WTA (NaN never wins: it is replaced by +inf)
                    //    t32 / area by one Newton step on the reciprocal: exact whenever the quotient is
                    //    representable, so a constant window returns the constant (ties stay ties)
                    const float t32 = se + dsum;
                    const float areaf = INTERIOR ? area_interior : cy2g[k] * cxf;
                    const float r0 = INTERIOR ? rcp_interior : __builtin_amdgcn_rcpf(areaf);
                    const float q0 = t32 * r0;
                    float qf = fmaf(fmaf(-q0, areaf, t32), r0, q0) * a.unit;
                    if (QOUT) {
                        const int yo = ty0 + o;
                        if (lane >= HALO && lane < 64 - HALO && x < W && yo < H) a.q_out[((size_t)d * H + yo) * W + x] = qf;
                    }
                    qf = (qf == qf) ? qf : __int_as_float(0x7f800000);
                    const unsigned long long key = ((unsigned long long)ordered_bits(qf) << 32) | (uint32_t)d;
                    __hip_atomic_fetch_min(reinterpret_cast<LDS_AS unsigned long long *>(keyg + k * 64), key,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            cur = nxt;
        }
    };
    static_assert(2 * WIN >= 2 * HALO, "two peeled groups cover the start-up rows");
    row_group(0, std::false_type{});
    row_group(WIN, std::false_type{});
    for (int base = 2 * WIN; base < NI; base += WIN) row_group(base, std::true_type{});
}

template <int RG, bool WORD64, int TH, int NW, bool QOUT>
__global__ __launch_bounds__(NW * 64) void fused_guided_fast_kernel(FastArgs a) {
    constexpr int WIN = 2 * RG + 1;
    constexpr int HALO = 2 * RG;
    constexpr int NI = TH + 4 * RG;
    constexpr int NJ = TH + 2 * RG;
    constexpr int TWV = 64 - 4 * RG;
    static_assert(NI % WIN == 0, "the row loop is unrolled by the window length without a tail");
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;

    if (a.gate_min > 0.f && !eps_gate_takes_fast(a.gate_eps, a.gate_min, *a.gmax)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *sp = smem_raw;
    TileLds<WordT> t;
    t.mG = reinterpret_cast<double *>(sp); sp += sizeof(double) * NJ * 64;  // (read-ahead past the last row lands in the next array)
    t.key = reinterpret_cast<uint64_t *>(sp); sp += sizeof(uint64_t) * TH * 64;
    t.iay1 = reinterpret_cast<double *>(sp); sp += sizeof(double) * ((NJ + 1) / 2 * 2);
    t.cy2 = reinterpret_cast<float *>(sp); sp += sizeof(double) * ((TH + 1) / 2 * 2);
    t.invtab = reinterpret_cast<double *>(sp); sp += sizeof(double) * 32;
    t.L = reinterpret_cast<WordT *>(sp); sp += sizeof(WordT) * NI * 64;
    t.inv = reinterpret_cast<float *>(sp); sp += sizeof(float) * NJ * 64;
    t.glo = reinterpret_cast<float *>(sp); sp += sizeof(float) * NI * 64;
    t.gout = reinterpret_cast<float *>(sp); sp += sizeof(float) * (TH * 64 + 16);
    t.next_d = reinterpret_cast<int *>(sp); sp += 16;
    t.k = reinterpret_cast<short *>(sp);

    const int lane = threadIdx.x & 63;
    const int H = a.H, W = a.W;
    const int tx0 = blockIdx.x * TWV, ty0 = blockIdx.y * TH;
    const WordT *lw = static_cast<const WordT *>(a.lw);

    // guide quantisation: |k| <= kmax keeps the (2r+1)^2-window sum of k*p inside int32
    const float gmax = *a.gmax;
    int kmax = (int)(2147483647u / ((uint32_t)(WIN * WIN) * max(a.large_units, 1u)));
    kmax = min(kmax, 32767);
    int sh = 0;
    if (gmax > 0.f && gmax < 3.0e38f) sh = min(max(ilogbf((float)kmax / gmax), -120), 120);
    const float scale_f = ldexpf(1.f, sh);
    const float inv_scale_f = ldexpf(1.f, -sh);
    const double inv_scale_d = (double)inv_scale_f;

    // Staging: every global load of the tile is issued before the first result is used (one memory round
    // trip; the workgroup owns the compute unit's LDS, so nothing else would hide the latency).  Addresses
    // are clamped into the image, cells outside it are zeroed when they are stored.
    {
        const int l = threadIdx.x & 63, i_first = threadIdx.x >> 6;
        constexpr int N1 = (NI + NW - 1) / NW, N2 = (NJ + NW - 1) / NW;
        const int gx = tx0 - HALO + l;
        const bool gx_in = gx >= 0 && gx < W;
        const int gxc = min(max(gx, 0), W - 1);
        float g_in[N1];
        WordT l_in[N1];
        double m_in[N2], d_in[N2];
#pragma unroll
        for (int u = 0; u < N1; ++u) {
            const int gyc = min(max(ty0 - HALO + i_first + u * NW, 0), H - 1);
            g_in[u] = a.guide[(size_t)gyc * W + gxc];
            l_in[u] = lw[(size_t)gyc * W + gxc];
        }
#pragma unroll
        for (int u = 0; u < N2; ++u) {
            const int gyc = min(max(ty0 - RG + i_first + u * NW, 0), H - 1);
            m_in[u] = a.mean[(size_t)gyc * W + gxc];
            d_in[u] = a.invden[(size_t)gyc * W + gxc];
        }
#pragma unroll
        for (int u = 0; u < N1; ++u) {
            const int i = i_first + u * NW;
            if (i >= NI) break;
            const int gy = ty0 - HALO + i, idx = i * 64 + l;
            const bool in = gx_in && gy >= 0 && gy < H;
            const float g = in ? g_in[u] : 0.f;
            const int kk = __float2int_rn(g * scale_f);
            t.k[idx] = (short)kk;
            t.glo[idx] = fmaf(-(float)kk, inv_scale_f, g);
            t.L[idx] = in ? l_in[u] : WordT(0);
            const int o = i - HALO;
            if (o >= 0 && o < TH) t.gout[o * 64 + l] = g;
        }
#pragma unroll
        for (int u = 0; u < N2; ++u) {
            const int j = i_first + u * NW;
            if (j >= NJ) break;
            const int gy = ty0 - RG + j, idx = j * 64 + l;
            const bool in = gx_in && gy >= 0 && gy < H;
            t.mG[idx] = in ? m_in[u] : 0.0;
            const int cyy = min(gy + RG, H - 1) - max(gy - RG, 0) + 1, cxx = min(gx + RG, W - 1) - max(gx - RG, 0) + 1;
            t.inv[idx] = in ? (float)(d_in[u] / (double)(cyy * cxx)) : 0.f;
        }
    }
    for (int idx = threadIdx.x; idx < TH * 64; idx += NW * 64) t.key[idx] = SHP_KEY_INIT;
    if (threadIdx.x < 16) t.gout[TH * 64 + threadIdx.x] = 0.f;
    for (int j = threadIdx.x; j < NJ; j += NW * 64) {
        int y = ty0 - RG + j;
        t.iay1[j] = (y >= 0 && y < H) ? 1.0 / (double)(min(y + RG, H - 1) - max(y - RG, 0) + 1) : 0.0;
    }
    for (int o = threadIdx.x; o < TH; o += NW * 64) {
        int y = ty0 + o;
        t.cy2[o] = (y < H) ? (float)(min(y + RG, H - 1) - max(y - RG, 0) + 1) : 0.f;
    }
    if (threadIdx.x <= WIN) t.invtab[threadIdx.x] = threadIdx.x ? 1.0 / (double)threadIdx.x : 0.0;
    if (threadIdx.x == 0) { t.next_d[0] = 0; t.next_d[1] = 0x7fffffff; }
    __syncthreads();

    const int x = tx0 - HALO + lane;   // column of this lane's cost and of its final output
    const int x1 = x - RG;             // column of this lane's stage-1 result (a, b)
    const bool x_in = x >= 0 && x < W;
    const bool x1_in = x1 >= 0 && x1 < W && lane >= RG;
    const int cx = x_in ? (min(x + RG, W - 1) - max(x - RG, 0) + 1) : 0;
    const int cx1 = x1_in ? (min(x1 + RG, W - 1) - max(x1 - RG, 0) + 1) : 0;
    const double invAX1 = t.invtab[cx1];                   // 0 when column x1 is outside the image
    const float cxf = (float)cx;
    const int R = a.radius;
    const bool x_valid = x_in && x >= R && x < W - R;
    const int lane1 = max(lane - RG, 0);
    const bool interior = (ty0 - HALO >= R) && (ty0 + TH + HALO <= H - R);
    const int n_disp = a.d_end - a.d_begin;

    // disparities are handed to the wavefronts of the workgroup one at a time
    for (;;) {
        // Every lane takes part in the LDS atomic (the counter advances by 64 per wavefront) and no
        // lane-dependent branch surrounds the hand-out: a `lane == 0` guard here lets the optimiser
        // specialise the loop for the other lanes, for which the guarded value is a constant.
        const int dn = __builtin_amdgcn_readfirstlane(atomicAdd(t.next_d, 1)) >> 6;
        const int d = a.d_begin + dn * a.n_slices + (int)blockIdx.z;
        if (d >= a.d_end || dn >= n_disp) break;
        // Every column this wavefront reads lies left of the first valid match (x - d < radius):
        // all costs are `large`, and a constant window filters to exactly that constant.  Only
        // the lowest such disparity can win a tie, so remember it and skip the pass.
        if (!QOUT && tx0 - HALO + 63 - d < R) {
            atomicMin(&t.next_d[1], d);
            continue;
        }
        if (interior)
            fast_disparity_pass<RG, WORD64, TH, QOUT, true>(a, t, d, lane, x, ty0, x_in, x_valid, invAX1, cxf, lane1,
                                                            inv_scale_d);
        else
            fast_disparity_pass<RG, WORD64, TH, QOUT, false>(a, t, d, lane, x, ty0, x_in, x_valid, invAX1, cxf, lane1,
                                                             inv_scale_d);
    }
    __syncthreads();
    const int skipped_d = t.next_d[1];
    const float large_f = (float)a.large_units * a.unit;
    for (int idx = threadIdx.x; idx < TH * 64; idx += NW * 64) {
        int o = idx >> 6, l = idx & 63;
        int gy = ty0 + o, gx = tx0 - HALO + l;
        if (l >= HALO && l < 64 - HALO && gy < H && gx < W) {
            uint64_t key = t.key[idx];
            if (skipped_d != 0x7fffffff) {
                const uint64_t kl = make_key(large_f, (uint32_t)skipped_d);
                key = kl < key ? kl : key;
            }
            if (a.atomic_out) atomicMin(reinterpret_cast<unsigned long long *>(&a.keys[(size_t)gy * W + gx]),
                                        (unsigned long long)key);
            else a.keys[(size_t)gy * W + gx] = key;
        }
    }
}

// tile height per (guided radius, census word size): TH + 4r is a multiple of 2r + 1 and the
// tile image fits the 160 KiB LDS
template <int RG, bool WORD64> struct FastTile;
template <> struct FastTile<1, false> { static constexpr int TH = 71; };
template <> struct FastTile<2, false> { static constexpr int TH = 67; };
template <> struct FastTile<3, false> { static constexpr int TH = 65; };
template <> struct FastTile<4, false> { static constexpr int TH = 65; };
template <> struct FastTile<1, true> { static constexpr int TH = 62; };
template <> struct FastTile<2, true> { static constexpr int TH = 62; };
template <> struct FastTile<3, true> { static constexpr int TH = 58; };
template <> struct FastTile<4, true> { static constexpr int TH = 56; };
// radii 6 .. 8: 13- to 17-row windows, 40 / 36 / 32 output columns per wavefront; the rings of the vertical windows
// need the registers of two wavefronts per SIMD (workgroups of 8 wavefronts)
template <> struct FastTile<6, false> { static constexpr int TH = 54; };
template <> struct FastTile<7, false> { static constexpr int TH = 47; };
template <> struct FastTile<8, false> { static constexpr int TH = 53; };
template <> struct FastTile<6, true> { static constexpr int TH = 41; };
template <> struct FastTile<7, true> { static constexpr int TH = 47; };
template <> struct FastTile<8, true> { static constexpr int TH = 36; };
#ifndef FF_NW
#define FF_NW 16
#endif

template <int RG, bool WORD64, bool QOUT, int NWV = FF_NW>
static int launch_fast_q(const FastArgs &a, hipStream_t s) {
    constexpr int TH = FastTile<RG, WORD64>::TH;
    constexpr int NI = TH + 4 * RG, NJ = TH + 2 * RG, TWV = 64 - 4 * RG;
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;
    constexpr size_t smem = sizeof(double) * NJ * 64 + sizeof(uint64_t) * TH * 64 + sizeof(double) * ((NJ + 1) / 2 * 2) +
                            sizeof(double) * ((TH + 1) / 2 * 2) + sizeof(double) * 32 + sizeof(WordT) * NI * 64 +
                            sizeof(float) * NJ * 64 + sizeof(float) * NI * 64 + sizeof(float) * (TH * 64 + 16) + 16 +
                            sizeof(short) * NI * 64;
    static_assert(smem <= 160 * 1024, "tile image exceeds the LDS");
    auto kern = fused_guided_fast_kernel<RG, WORD64, TH, NWV, QOUT>;
    SHP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    dim3 grid(ceil_div(a.W, TWV), ceil_div(a.H, TH), a.n_slices);
    hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), smem, s, a);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
template <int RG, bool WORD64>
static int launch_fast_t(const FastArgs &a, hipStream_t s) {
    constexpr int NWV = RG >= 5 ? 8 : FF_NW;
    return a.q_out ? launch_fast_q<RG, WORD64, true, NWV>(a, s) : launch_fast_q<RG, WORD64, false, NWV>(a, s);
}

bool fused_fast_supported(const CensusCfg &cfg, int guided_radius) {
    return cfg.exact && guided_radius >= 1 && guided_radius <= 8 && guided_radius != 5 && cfg.unit > 0.f;
}

static int fast_tile_height(int guided_radius, bool w64) {
    switch (guided_radius) {
        case 1: return w64 ? FastTile<1, true>::TH : FastTile<1, false>::TH;
        case 2: return w64 ? FastTile<2, true>::TH : FastTile<2, false>::TH;
        case 3: return w64 ? FastTile<3, true>::TH : FastTile<3, false>::TH;
        case 4: return w64 ? FastTile<4, true>::TH : FastTile<4, false>::TH;
        case 6: return w64 ? FastTile<6, true>::TH : FastTile<6, false>::TH;
        case 7: return w64 ? FastTile<7, true>::TH : FastTile<7, false>::TH;
        case 8: return w64 ? FastTile<8, true>::TH : FastTile<8, false>::TH;
        default: return 65;
    }
}

// Number of disparity slices (gridDim.z) that gives the chip enough workgroups.
int fused_fast_slices(int H, int W, int guided_radius, int n_disp, bool word64) {
    int tiles = ceil_div(W, 64 - 4 * guided_radius) * ceil_div(H, fast_tile_height(guided_radius, word64));
    const int nw = guided_radius >= 5 ? 8 : FF_NW;
    int s = 1;
    const char *e = getenv("STEREO_FAST_SLICES");
    if (e) return atoi(e) > 0 ? atoi(e) : 1;
    while (tiles * s < 1024 && s * nw * 4 <= n_disp) s *= 2;
    return s;
}

int launch_fused_guided_fast(const void *d_lw, const void *d_rw, const CensusCfg &cfg, const float *d_guide,
                             const double *d_mean, const double *d_invden, const float *d_gmax, int H, int W,
                             int dmax, int guided_radius, int d_begin, int d_end, int n_slices, uint64_t *d_keys,
                             float *d_q_out, hipStream_t s, float gate_eps, float gate_min) {
    if (H <= 0 || W <= 0 || d_begin >= d_end) return STEREO_OK;
    FastArgs a;
    memset(&a, 0, sizeof(a));
    a.gate_eps = gate_eps; a.gate_min = gate_min;
    a.lw = d_lw; a.rw = d_rw; a.guide = d_guide; a.mean = d_mean; a.invden = d_invden; a.gmax = d_gmax;
    a.keys = d_keys; a.q_out = d_q_out;
    a.H = H; a.W = W; a.D = dmax; a.d_begin = d_begin; a.d_end = d_end;
    a.n_slices = n_slices < 1 ? 1 : n_slices;
    a.radius = cfg.radius; a.unit = cfg.unit;
    a.large_units = (uint32_t)std::lround((double)cfg.large / (double)cfg.unit);
    a.atomic_out = a.n_slices > 1;
    const bool w64 = cfg.word64 != 0;
    switch (guided_radius) {
        case 1: return w64 ? launch_fast_t<1, true>(a, s) : launch_fast_t<1, false>(a, s);
        case 2: return w64 ? launch_fast_t<2, true>(a, s) : launch_fast_t<2, false>(a, s);
        case 3: return w64 ? launch_fast_t<3, true>(a, s) : launch_fast_t<3, false>(a, s);
        case 4: return w64 ? launch_fast_t<4, true>(a, s) : launch_fast_t<4, false>(a, s);
        case 6: return w64 ? launch_fast_t<6, true>(a, s) : launch_fast_t<6, false>(a, s);
        case 7: return w64 ? launch_fast_t<7, true>(a, s) : launch_fast_t<7, false>(a, s);
        case 8: return w64 ? launch_fast_t<8, true>(a, s) : launch_fast_t<8, false>(a, s);
        default: set_error("one-column fused guided kernel supports radius 1..4 and 6..8"); return STEREO_EUNSUPPORTED;
    }
}

}  // namespace shp

// Kernel templates of the two-columns-per-lane fused guided-filter + WTA kernel (see guided_pair.hip for the
// description).  Included by guided_pair.hip (costs from census words) and guided_pair_codes.hip (costs from
// 16-bit cost codes of a caller-supplied volume), which instantiate it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <type_traits>
#include <vector>


#include "fast_common.hpp"

namespace shp {


// Empty asm statements pin values in registers against re-association (volatile: the plain form measured no faster).
#define FP_PIN asm volatile
template <typename WordT>
struct PairLds {
    double *mG;      // [NJ][128] box mean of the guide, rows shifted by r (element c + r = tile column c)
    uint64_t *key;   // [TH][128] packed (cost, d)
    WordT *L;        // [NI][128] left census words.  32-bit words: L and glo are ONE array of 16-byte records per
                     // lane and row, (L(2l), L(2l+1), glo(2l), glo(2l+1)), so that a lane's four values are one LDS read
    float *inv;      // [NJ][128] cost unit / ((var + eps) * window area): a and b come out in cost units; rows shifted like mG
    float *glo;      // [NI][128] guide - k * 2^-s
    float *gout;     // [TH][128] (+pad) guide at the output pixels
    int *next_d;     // [0] next disparity to hand out, [1] lowest disparity whose costs are all `large`
    short *k;        // [NI][128] round(guide * 2^s)
};

// 1 / n for the window lengths a clipped box filter can have (n = 0: outside the image).  Row counts are the
// same for every lane, so these are scalar loads; the table replaces three small LDS arrays.
static __constant__ double pair_inv_n[16] = {0.0,       1.0,       1.0 / 2.0,  1.0 / 3.0,  1.0 / 4.0,  1.0 / 5.0,  1.0 / 6.0,  1.0 / 7.0,
                                      1.0 / 8.0, 1.0 / 9.0, 1.0 / 10.0, 1.0 / 11.0, 1.0 / 12.0, 1.0 / 13.0, 1.0 / 14.0, 1.0 / 15.0};
// rows (or columns) of the radius-r window centred on `y` that lie inside [0, n); 0 when y itself is outside
__device__ __forceinline__ int clipped_len(int y, int r, int n) {
    return (y >= 0 && y < n) ? min(y + r, n - 1) - max(y - r, 0) + 1 : 0;
}
// A tile's key array holds (cost, d) as float64 patterns (see the end of pair_disparity_pass): high word = the
// cost's float32 pattern, low word = (d << 7) xor the sign-extended top byte of that pattern.  The write-out turns
// them into the library's ordered integer keys.  Start value: (+inf, d = 0), which SHP_KEY_INIT stands for.
#define PAIR_TILE_KEY_INIT ((uint64_t(0x7F800000u) << 32) | uint64_t(0x7Fu))
__device__ __forceinline__ uint64_t pair_tile_key_to_key(uint64_t tile_key) {
    const uint32_t hi = (uint32_t)(tile_key >> 32), lo = (uint32_t)tile_key;
    return make_key(__uint_as_float(hi), (lo ^ (uint32_t)((int32_t)hi >> 24)) >> 7);
}
template <typename T>
using Pair = T __attribute__((ext_vector_type(2)));  // the two slots of a lane, loaded with one LDS read
// The float32 arithmetic of the two slots is written on pairs: with two wavefronts per SIMD every
// vector instruction costs one issue slot, so a packed instruction halves the cost of the pair.
using f2 = Pair<float>;
using LdsRec = uint32_t __attribute__((ext_vector_type(4)));   // (L(2l), L(2l+1), glo(2l), glo(2l+1)) of a lane and row
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// {a.y * b.x + c.y, a.y * b.y + c.y}: the high halves of a and c serve both results.  Written as the instruction
// with its operand selects: from the vector expression the compiler first copies a.y and c.y into low halves.
__device__ __forceinline__ f2 fma2_hi(f2 a, f2 b, f2 c) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// a float64 value of the lane CTRL away (both halves move; lanes without a source read zero)
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Tile column c inside a de-interleaved 128-column LDS row: even columns first, then odd ones.  Lane l
// owns columns 2l and 2l + 1; with this order the lanes' accesses to either parity are consecutive
// (no two-way bank conflict from the stride of two elements).
__device__ __forceinline__ int deint(int c) { return ((c & 1) << 6) | (c >> 1); }

// Per-lane constants of the two slots.
struct PairLane {
    int x[2];             // image column of the slot's cost and of its final output
    bool x_in[2];         // column inside the image
    bool x_valid[2];      // census centre valid in x
    double invAX1[2];     // cost unit / (columns of the clipped stage-1 window), 0 outside the image
    float cxf[2];         // columns of the clipped output window
    double rcx[2];        // 1 / that (0 outside the image): the precise variant divides in float64
    int c1, c1b;          // de-interleaved positions of the stage-1 columns of slot 0 and slot 1 (clamped to >= 0)
};

// One disparity of one tile.  MODE 0: general.  MODE 1: every row the tile touches is inside the
// image and has a valid census centre, so the per-row tests vanish.  MODE 2: in addition every
// column of the tile (halo included) has a valid match at this disparity and full filter
// windows, and the tile's inputs are finite: no cost select, constant window area, no NaN test.
// PRECISE: float64 where the fast variant has float32 -- the sums of (guide's low-order part x cost) in the first
// stage and the whole second box stage (lane shifts move both halves of every partial sum): 40 % more vector
// instructions, no limit on eps.
// SRC: the costs come from 16-bit cost codes (units of the cost unit) of a caller-supplied volume, one [H][W] layer per
// disparity, instead of from census words: no invalid matches (every in-image cost counts as it is), no skipped
// disparities; a.radius is 0 then.
template <int RG, bool WORD64, int TH, bool QOUT, int MODE, bool PRECISE, bool SRC = false>
__device__ __forceinline__ void pair_disparity_pass(
    const FastArgs &a, const PairLds<typename std::conditional<WORD64, uint64_t, uint32_t>::type> &t, const int d,
    const int lane, const int ty0, const PairLane &pl, const double inv_scale_d, const double inv_post) {
    constexpr int WIN = 2 * RG + 1;
    constexpr int HALO = 2 * RG;
    constexpr int NI = TH + 4 * RG;
    constexpr int TC = 128;
    constexpr bool INTERIOR = MODE >= 1;
    constexpr bool LEAN = MODE == 2;
    // Tiles whose windows are all full stage the plain 1 / (var + eps) (see the kernel).  Their lean passes then
    // work on a and b scaled by (2r+1)^2 / unit -- b' = sum(p) - a' mean(G) needs no product for its first
    // term -- and scale the result once per output; their other passes multiply `inv_post` back in.
    constexpr bool SCALED = LEAN && RG <= 3;
    // 32-bit census words (radius <= 4): a cost is at most 1200 units, so the vertical sums (<= 8400) and the
    // window sums (<= 58800) of the two slots share one register, 16 bits each: half the adds and lane shifts.
    // 64-bit census words (radius <= 8): a valid cost is at most 2040 units, so passes without `large` costs (the
    // lean ones) pack as well up to guided radius 3 ((r+1) lanes x (2r+1) rows x 2040 < 2^16).
    // Guided radius 5 (32-bit words): (r+1) (2r+1) 1200 exceeds 16 bits, so only its passes without `large` costs pack.
    constexpr bool PACK16 = WORD64 ? (LEAN && RG <= 3) : (RG <= 4 || LEAN);
    static_assert(!SRC || (!WORD64 && RG <= 4 && !QOUT), "cost codes: packed 16-bit window sums in every pass");
    constexpr bool MAGIC = PACK16 && !PRECISE;
    constexpr uint32_t MAGIC_BITS = MAGIC ? 0x4B000000u : 0u;
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;
    const int H = a.H, W = a.W, R = a.radius;
    const WordT *rw = static_cast<const WordT *>(a.rw);
    const int iv_lo = max(R - (ty0 - HALO), 0), iv_hi = min(H - R - (ty0 - HALO), NI);
    const int ii_lo = max(-(ty0 - HALO), 0), ii_hi = min(H - (ty0 - HALO), NI);

    // d in the position it has in the low word of a tile key (see PAIR_TILE_KEY_INIT)
    const uint32_t d_sh = (uint32_t)d << 7;
    // The factor from a full window's sum to the filtered cost, in a scalar register (one operand of the two
    // multiplies that write the keys' high words).  Unscaled passes: fl(1 / (2r+1)^2).
    // Scaled passes: the window sums are (2r+1)^4 / unit times the result.  fl(1 / (2r+1)^4) has a relative error
    // below 2^-25 for r = 1..3, so a window of equal costs (the sum is (2r+1)^4 times the cost in units, exact)
    // still gives exactly that cost; the cost unit, a power of two, goes into the same factor (with any other
    // unit the kernel runs these tiles without lean passes).
    // (the factor comes from the host as a kernel argument: a.lean_scale)
    float rcp_full_s = a.lean_scale;
    asm volatile("" : "+s"(rcp_full_s));
    const double lean_scale_d = SCALED ? (double)a.unit / (double)(WIN * WIN * WIN * WIN) : 1.0 / (double)(WIN * WIN);
    // (the selectable costs carry the bits of 2^23 like the computed ones, see MAGIC)
    uint32_t large_lane[2];
    bool xd_valid[2];
    unsigned roff[2];  // byte offset of the right census word inside its row
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        large_lane[j] = (pl.x_in[j] ? a.large_units : 0u) | MAGIC_BITS;
        xd_valid[j] = SRC ? pl.x_in[j] : (pl.x_valid[j] && (pl.x[j] - d >= R));
        // (cost codes: the lane's two columns are one aligned 32-bit word -- slot 0's column is even -- inside one
        //  group of 128 columns, so one load serves both; columns outside the image are not used)
        const int xc = min(max(pl.x[0], 0), (W - 1) & ~1);
        roff[j] = SRC ? (unsigned)(xc / PAIR_CODE_GROUP) * (unsigned)(a.code_dn * PAIR_CODE_GROUP * 2) + (unsigned)(xc % PAIR_CODE_GROUP) * 2u
                      : (unsigned)min(max(pl.x[j] - d, 0), W - 1) * (unsigned)sizeof(WordT);
    }

    uint32_t C1[2] = {0u, 0u};
    uint32_t C1p = 0u, prp[WIN];
    int I2[2] = {0, 0};
    f2 F2 = {0.f, 0.f};
    double F2d[2] = {0.0, 0.0};   // PRECISE: the same sums (guide's low-order part times cost) in float64
    double A[2] = {0.0, 0.0}, B[2] = {0.0, 0.0};
    // rings of the rows inside the two vertical windows; a slot is first read one window length after it was
    // first written (the start-up row groups below are peeled and skip the retiring step), so none is cleared
    uint32_t pr[2][WIN];
    int kr[2][WIN];
    f2 gr[WIN];
    double grd[2][WIN];
    double ar[2][WIN], br[2][WIN];

    auto row_of = [&](int i) -> int {
        int y = ty0 - HALO + i;
        return INTERIOR ? y : min(max(y, 0), H - 1);
    };
    double ia_interior[2];
    float area_interior[2], rcp_interior[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ia_interior[j] = pl.invAX1[j] * (1.0 / (double)WIN);   // (invAX1 carries the cost unit)
        area_interior[j] = pl.cxf[j] * (float)WIN;
        rcp_interior[j] = __builtin_amdgcn_rcpf(area_interior[j]);
    }
    const f2 area_mode = f2{area_interior[0], area_interior[1]};
    const f2 rcp_mode = f2{rcp_interior[0], rcp_interior[1]};
    const double ia_full = (double)a.unit * (1.0 / (double)(WIN * WIN));

    // Operands of a row are fetched one row ahead; the head of the next row is carried from one
    // row group into the next so that no group starts by waiting for its own loads.
    struct RowHead {
        WordT L[2]; int k[2]; f2 glo; double mG[2]; float inv_f[2]; double inv_d[2];
    };
    RowHead cur;
    bool have_head = false;
    // right census words come from global memory (L2): they are requested RPF rows ahead
    constexpr int RPF = SRC ? PAIR_CODES_ROWS_AHEAD : 3;   // (census words, from L2: 2 and 4 measured within 0.3 % of 3)
    WordT rq[RPF][2];
    // buffer loads: the row offset travels in a scalar register and the lane's column offset in
    // one 32-bit register, so no 64-bit address arithmetic is spent per row
    // (cost codes: the descriptor starts at this disparity's codes of the tile's first row; a row of the code array
    //  is code_row_bytes long, so the offsets of the tile's rows stay small)
    const int code_row0 = max(ty0 - HALO, 0);
    const int code_row_bytes = pair_code_groups(W) * a.code_dn * PAIR_CODE_GROUP * 2;
    const __amdgpu_buffer_rsrc_t rsrc = SRC
        ? __builtin_amdgcn_make_buffer_rsrc(
              const_cast<uint16_t *>(a.codes + ((size_t)code_row0 * pair_code_groups(W) * (size_t)a.code_dn + (size_t)(d - a.code_d0)) * PAIR_CODE_GROUP),
              0, 0x7ffffff0, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<WordT *>(rw), 0, (int)((size_t)H * W * sizeof(WordT)), 0x00020000);
    auto load_right = [&](int i, WordT *dst) {
#ifdef PAIR_CODES_TIMING_ONLY_SAME_ROW   // (timing experiment: every row of a pass reads the same codes, from cache)
        const int soff = SRC ? 0 : row_of(min(i, NI - 1)) * W * (int)sizeof(WordT);
#else
        const int soff = SRC ? (row_of(min(i, NI - 1)) - code_row0) * code_row_bytes : row_of(min(i, NI - 1)) * W * (int)sizeof(WordT);
#endif
        if (SRC) {
            dst[0] = (WordT)(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)roff[0], soff, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (WORD64) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)roff[j], soff, 0);
                dst[j] = (WordT)(((unsigned long long)(uint32_t)v[1] << 32) | (uint32_t)v[0]);
            } else {
                dst[j] = (WordT)(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)roff[j], soff, 0);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < RPF; ++u) load_right(u, rq[u]);
    // PRECISE: 1 / (var + eps) in float64 from the statistics array itself (the tile image keeps it in float32),
    // fetched with the other operands of a row, one row ahead
    const __amdgpu_buffer_rsrc_t rsrc_inv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(a.invden), 0, (int)((size_t)H * W * sizeof(double)), 0x00020000);
    unsigned ioff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tc1 = max(2 * lane - RG, 0) + j;   // tile column of the slot's stage-1 result
        ioff[j] = (unsigned)min(max(pl.x[0] - 2 * lane + tc1, 0), W - 1) * (unsigned)sizeof(double);
    }
    auto row_group = [&](const int base, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        LDS_AS const Pair<WordT> *Lg = WORD64 ? lds_lane_ptr(reinterpret_cast<const Pair<WordT> *>(t.L + base * TC + 2 * lane)) : nullptr;
        LDS_AS const LdsRec *LGg = WORD64 ? nullptr : lds_lane_ptr(reinterpret_cast<const LdsRec *>(reinterpret_cast<const uint32_t *>(t.L) + base * 2 * TC + 4 * lane));
        LDS_AS const uint32_t *kg = lds_lane_ptr(reinterpret_cast<const uint32_t *>(t.k + base * TC + 2 * lane));
        LDS_AS const Pair<float> *glog = WORD64 ? lds_lane_ptr(reinterpret_cast<const Pair<float> *>(t.glo + base * TC + 2 * lane)) : nullptr;
        // stage-1 columns 2l - r and 2l - r + 1 of the two slots (r odd: odd column first)
        // (rows of mG and inv are stored shifted by r: element 2l + j is stage-1 column 2l - r + j, so a lane's two
        //  values are one 16-byte / 8-byte read)
        LDS_AS const Pair<double> *mGp = lds_lane_ptr(reinterpret_cast<const Pair<double> *>(t.mG + (base - HALO) * TC + 2 * lane));
        LDS_AS const Pair<float> *invp = lds_lane_ptr(reinterpret_cast<const Pair<float> *>(t.inv + (base - HALO) * TC + 2 * lane));
        LDS_AS const Pair<float> *goutg = lds_lane_ptr(reinterpret_cast<const Pair<float> *>(t.gout + (base - 2 * HALO) * TC + 2 * lane));
        LDS_AS uint64_t *keyg = lds_lane_ptr(t.key + (base - 2 * HALO) * TC + lane);
        auto fetch_head = [&](int k) {
            RowHead r;
            const uint32_t kk = kg[k * (TC / 2)];
            if constexpr (WORD64) {
                const Pair<WordT> lp = Lg[k * (TC / 2)];
                r.L[0] = lp.x; r.L[1] = lp.y;
                r.glo = glog[k * (TC / 2)];
            } else {
                const LdsRec rec = LGg[k * (TC / 2)];
                r.L[0] = rec.x; r.L[1] = rec.y;
                r.glo = f2{__uint_as_float(rec.z), __uint_as_float(rec.w)};
            }
            r.k[0] = (int)(short)(kk & 0xffffu); r.k[1] = (int)kk >> 16;
            const Pair<double> mp = mGp[k * (TC / 2)];
            r.mG[0] = mp.x; r.mG[1] = mp.y;
            if (PRECISE) {
                const int gy = min(max(ty0 - RG + base + k - HALO, 0), H - 1);
                const int soff = gy * W * (int)sizeof(double);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_inv, (int)ioff[j], soff, 0);
                    r.inv_d[j] = __hiloint2double((int)v[1], (int)v[0]);
                }
            } else {
                const Pair<float> ip = invp[k * (TC / 2)];
                r.inv_f[0] = ip.x; r.inv_f[1] = ip.y;
            }
            return r;
        };
        if (!have_head) { cur = fetch_head(0); have_head = true; }
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const int i = base + k;
            const RowHead nxt = fetch_head(k + 1);
            const WordT Rcur[2] = {rq[0][0], SRC ? WordT(0) : rq[0][1]};
#pragma unroll
            for (int u = 0; u + 1 < RPF; ++u) { rq[u][0] = rq[u + 1][0]; if (!SRC) rq[u][1] = rq[u + 1][1]; }
            load_right(i + RPF, rq[RPF - 1]);
            const double mG[2] = {cur.mG[0], cur.mG[1]};
            const float inv_f[2] = {cur.inv_f[0], cur.inv_f[1]};
            f2 gq[RG + 1];  // guide of the output row at tile columns 2(l - u), 2(l - u) + 1  (lanes l < u read the
                            // floats in front of the row: their terms leave the wavefront with the shifts)
#pragma unroll
            for (int u = 0; u <= RG; ++u) gq[u] = goutg[k * (TC / 2) - u];
            // 2a. the row leaving the window [i - 2r, i] retires first, so that its ring registers are
            //     free before the new row's values are formed (no register copies at the loop edge)
            if (STEADY || i >= WIN) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!PACK16) C1[j] -= pr[j][k];
                    I2[j] -= kr[j][k];
                }
                if (PACK16) C1p -= prp[k];
                if (PRECISE) { F2d[0] -= grd[0][k]; F2d[1] -= grd[1][k]; }
                else F2 -= gr[k];
            }
            f2 pf;
            uint32_t pq[2];
            // 1. costs in units; k*p and g_lo*p; 2b. the new row enters the vertical running sums
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // MAGIC: the cost is formed on top of the pattern of 2^23, so that the same register read as a float32
                // is 2^23 + cost exactly (no integer-to-float conversion per slot: one packed subtraction for both);
                // the 24-bit multiply and the 16-bit packing below read the low bits, which are the cost itself
                // (cost codes: the loaded word holds the codes of both slots)
                const uint32_t fs = SRC ? ((j == 0 ? (uint32_t)Rcur[0] & 0xffffu : (uint32_t)Rcur[0] >> 16) | MAGIC_BITS)
                                        : field_sum((WordT)(cur.L[j] ^ Rcur[j]), MAGIC_BITS);
                uint32_t p;
                if (LEAN) {
                    p = fs;
                } else if (INTERIOR) {
                    p = xd_valid[j] ? fs : large_lane[j];
                } else {
                    const bool rowv = i >= iv_lo && i < iv_hi;
                    const bool rowi = i >= ii_lo && i < ii_hi;
                    p = (rowv && xd_valid[j]) ? fs : large_lane[j];
                    p = rowi ? p : MAGIC_BITS;
                }
                int kp = __mul24(cur.k[j], (int)p);
                FP_PIN("" : "+v"(kp));  // one multiply: the running sum adds kp instead of re-deriving it
                pf[j] = MAGIC ? __uint_as_float(p) : (float)p;
                if (!PACK16) { C1[j] += p; pr[j][k] = p; }
                I2[j] += kp;
                kr[j][k] = kp;
                pq[j] = p;
            }
            if (MAGIC) pf -= f2{8388608.f, 8388608.f};
            if (PACK16) {
                // (cost codes, no selects: the loaded word is the packed pair)
                const uint32_t pp = (SRC && LEAN) ? (uint32_t)Rcur[0]
                                    : MAGIC       ? __builtin_amdgcn_perm(pq[1], pq[0], 0x05040100u) : (pq[0] | (pq[1] << 16));
                C1p += pp;
                prp[k] = pp;
            }
            if (PRECISE) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double gl = (double)cur.glo[j] * (double)pq[j];   // (exact: 24 x 11 bits)
                    F2d[j] += gl;
                    grd[j][k] = gl;
                }
            } else {
                const f2 gl = cur.glo * pf;
                F2 += gl;
                gr[k] = gl;
            }
            if (STEADY || i >= HALO) {
                // 3. sums over the 2r + 1 columns ending in each slot: the box sums of column x - r
                uint32_t e1 = C1[0], o1 = C1[1], e1r = C1[0], o1r = C1[1];
                uint32_t cp = C1p, cpr = C1p;
                // k p and g_lo p: with P = (even + odd column of a lane), the window ending in the odd column is
                // P(l) + ... + P(l-r+1) + odd(l-r) and the one ending in the even column even(l) + P(l-1) + ... + P(l-r):
                // one add for P and r shifted adds per slot (2r + 1 instead of 2r + 2 instructions)
                const int Pi = I2[0] + I2[1];
                const float Pf = F2.x + F2.y;
                const double Pd = F2d[0] + F2d[1];
                int ci = I2[1], di = Pi;
                float cf = F2.y, df = Pf;
                double cd = F2d[1], dd = Pd;
#pragma unroll
                for (int u = 0; u < RG; ++u) {
                    e1r = e1; o1r = o1;   // partial sums of u + 1 lanes
                    if (PACK16) {
                        cpr = cp;
                        cp = C1p + (uint32_t)dpp_i<DPP_WAVE_SHR1>((int)cp);
                    } else {
                        e1 = C1[0] + (uint32_t)dpp_i<DPP_WAVE_SHR1>((int)e1);
                        o1 = C1[1] + (uint32_t)dpp_i<DPP_WAVE_SHR1>((int)o1);
                    }
                    ci = Pi + dpp_i<DPP_WAVE_SHR1>(ci);
                    if (PRECISE) cd = Pd + dpp_d<DPP_WAVE_SHR1>(cd);
                    else cf = Pf + dpp_f<DPP_WAVE_SHR1>(cf);
                    if (u + 1 < RG) {
                        di = Pi + dpp_i<DPP_WAVE_SHR1>(di);
                        if (PRECISE) dd = Pd + dpp_d<DPP_WAVE_SHR1>(dd);
                        else df = Pf + dpp_f<DPP_WAVE_SHR1>(df);
                    }
                }
                // (the integer partial sums are pinned: re-associated, each three-operand add would need its
                //  lane-shifted operands as separate move instructions)
                //  (only the sums the variant uses: a pinned unused one costs a move of its zero per row)
                if (PACK16) FP_PIN("" : "+v"(ci), "+v"(di), "+v"(cp));
                else FP_PIN("" : "+v"(e1), "+v"(o1), "+v"(ci), "+v"(di));
                uint32_t s1[2];
                int si[2];
                float sf[2];
                if (PACK16) {
                    const uint32_t tp = (uint32_t)__builtin_amdgcn_mov_dpp((int)cpr, DPP_WAVE_SHR1, 0xf, 0xf, true);
                    s1[0] = (cp & 0xffffu) + (tp >> 16);
                    s1[1] = (cp >> 16) + (cpr & 0xffffu);
                } else {
                    s1[0] = e1 + (uint32_t)dpp_i<DPP_WAVE_SHR1>((int)o1r);
                    s1[1] = o1 + e1r;
                }
                si[0] = I2[0] + dpp_i<DPP_WAVE_SHR1>(di);
                double sfd[2];
                if (PRECISE) {
                    sfd[0] = F2d[0] + dpp_d<DPP_WAVE_SHR1>(dd);
                    sfd[1] = cd;
                } else {
                    sf[0] = F2.x + dpp_f<DPP_WAVE_SHR1>(df);
                    sf[1] = cf;
                    sfd[0] = (double)sf[0];
                    sfd[1] = (double)sf[1];
                }
                si[1] = ci;
                // 4. a, b; 5. their vertical running sums over stage-1 rows [j - 2r, j]
                const int k2 = (k + 1) % WIN;
                if (STEADY || i >= HALO + WIN) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        A[j] -= ar[j][k2];
                        B[j] -= br[j][k2];
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double ia = LEAN ? ia_full : INTERIOR ? ia_interior[j]
                                                     : pair_inv_n[clipped_len(ty0 - HALO + i - RG, RG, H)] * pl.invAX1[j];
                    const double s1d = (double)s1[j];
                    const double u = fma(-mG[j], s1d, fma((double)si[j], inv_scale_d, sfd[j]));
                    const double av = PRECISE ? (SCALED ? u * cur.inv_d[j] : u * cur.inv_d[j] * ia)
                                      : LEAN  ? u * (double)inv_f[j] : u * (double)inv_f[j] * inv_post;
                    const double bv = SCALED ? fma(-av, mG[j], s1d) : fma(-av, mG[j], s1d * ia);
                    A[j] += av;
                    B[j] += bv;
                    ar[j][k2] = av;
                    br[j][k2] = bv;
                }
                if (STEADY || i >= 2 * HALO) {
                    // 6. sums over the 2r + 1 stage-1 columns starting in each slot.  Both columns of a lane are
                    //    evaluated at ONE reference guide value c = G(2l): E = A c + B in float64, and the window sum
                    //    of A_j G(x) + B_j is  sum_j [E_j + A_j (G(x) - c_j)]  with c_j the reference of column j's lane.
                    //    Every term is small where the guide is locally smooth and every difference is between guide
                    //    values a few columns apart, so float32 suffices; and because a lane's two columns share
                    //    their reference, a lane contributes ONE term  (E_even + E_odd) + (A_even + A_odd)(G(x) - c)
                    //    to an output whose window holds both of its columns.
                    //    The terms are formed in the lane that owns the columns (it reads the guide of the outputs
                    //    to its left from the tile image) and travel down the lanes inside the shifted adds: lane m's
                    //    term for the output u lanes to its left enters the chain at step u, so one chain of r
                    //    shifted adds per slot sums values and corrections together.
                    //    Slot 0 of lane l (x = 2l):     lanes l .. l+r-1 whole (its own lane without correction:
                    //                                   G(x) = c), the even column of lane l+r.
                    //    Slot 1 of lane l (x = 2l + 1): the odd column of lane l, lanes l+1 .. l+r whole.
                    const int o = i - 2 * HALO;
                    f2 qv;
                    if (PRECISE) {
                        // the same windows on the float64 sums themselves: with P = (even + odd column of a lane),
                        // slot 0 is P(l) + ... + P(l+r-1) + even(l+r) and slot 1 odd(l) + P(l+1) + ... + P(l+r)
                        const double PAd = A[0] + A[1], PBd = B[0] + B[1];
                        double cA = A[0], cB = B[0], dA = PAd, dB = PBd;
#pragma unroll
                        for (int u = 0; u < RG; ++u) {
                            cA = PAd + dpp_d<DPP_WAVE_SHL1>(cA);
                            cB = PBd + dpp_d<DPP_WAVE_SHL1>(cB);
                            if (u + 1 < RG) {
                                dA = PAd + dpp_d<DPP_WAVE_SHL1>(dA);
                                dB = PBd + dpp_d<DPP_WAVE_SHL1>(dB);
                            }
                        }
                        const double sA[2] = {cA, A[1] + dpp_d<DPP_WAVE_SHL1>(dA)};
                        const double sB[2] = {cB, B[1] + dpp_d<DPP_WAVE_SHL1>(dB)};
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const double td = fma(sA[j], (double)gq[0][j], sB[j]);
                            const double rs = LEAN ? lean_scale_d
                                              : INTERIOR ? pair_inv_n[WIN] * pl.rcx[j]
                                                         : pair_inv_n[clipped_len(ty0 + o, RG, H)] * pl.rcx[j];
                            qv[j] = (float)(td * rs);
                        }
                    } else {
                    const float c = gq[0].x;
                    const double cref = (double)c;
                    const float Ex = (float)fma(A[0], cref, B[0]);
                    const float Ey = (float)fma(A[1], cref, B[1]);
                    const float a0 = (float)A[0], a1 = (float)A[1];
                    const float S = Ex + Ey;
                    const f2 PA = f2{a0, a0 + a1}, PE = f2{Ex, S};   // (even column, whole lane)
                    const f2 cc = f2{c, c};
                    f2 sum = fma2(PA, gq[RG] - cc, PE);
#pragma unroll
                    for (int u = RG - 1; u >= 1; --u) {
                        const f2 X = fma2_hi(PA, gq[u] - cc, PE);
                        sum.x = X.x + dpp_f<DPP_WAVE_SHL1>(sum.x);
                        sum.y = X.y + dpp_f<DPP_WAVE_SHL1>(sum.y);
                    }
                    sum.x = S + dpp_f<DPP_WAVE_SHL1>(sum.x);
                    sum.y = fmaf(a1, gq[0].y - c, Ey) + dpp_f<DPP_WAVE_SHL1>(sum.y);
                    // 7. q and the running WTA.  a and b already carry the cost unit.  Full windows: one multiply by
                    //    fl(1 / (2r+1)^2), whose relative error is below 2^-25 for r = 1..4, so a window of equal
                    //    costs c (t32 = (2r+1)^2 c exactly) rounds to exactly c; clipped windows: one Newton step on
                    //    the reciprocal, exact whenever the quotient is representable.  Either way constant `large`
                    //    windows stay exact ties.
                    const f2 t32 = sum;
                    if (LEAN && RG <= 4) {   // (fl(1 / 121) is not close enough: radius 5 divides like the clipped windows)
                        // (two plain multiplies: each result is the high word of its slot's key, see below)
                        qv.x = t32.x * rcp_full_s;
                        qv.y = t32.y * rcp_full_s;
                    } else {
                        f2 areav, r0v;
                        if (INTERIOR) {
                            areav = area_mode; r0v = rcp_mode;
                        } else {
                            const float cy = (float)clipped_len(ty0 + o, RG, H);
                            areav = f2{cy * pl.cxf[0], cy * pl.cxf[1]};
                            r0v = f2{__builtin_amdgcn_rcpf(areav.x), __builtin_amdgcn_rcpf(areav.y)};
                        }
                        const f2 q0 = t32 * r0v;
                        qv = fma2(fma2(-q0, areav, t32), r0v, q0);
                    }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float tf = qv[j];
                        if (QOUT) {
                            const int yo = ty0 + o;
                            const int tc = 2 * lane + j;
                            if (tc >= HALO && tc < TC - HALO && pl.x[j] >= 0 && pl.x[j] < W && yo < H)
                                a.q_out[((size_t)d * H + yo) * W + pl.x[j]] = tf;
                        }
                        if (!LEAN) tf = (tf == tf) ? tf : __int_as_float(0x7f800000);
                        // The tile's keys are float64 patterns, merged with the LDS unit's float64 minimum: it
                        // compares sign and magnitude, so the cost's float32 pattern goes into the high word as it
                        // is (no order-preserving integer map), and the patterns are finite float64 values for
                        // every float32 cost that is not NaN (a float32 infinity has the exponent 0x7f8 of 0x7ff).
                        // The low word breaks ties of equal costs in favour of the lower d: for a negative
                        // cost the larger magnitude is the smaller number, so d has to enter inverted.  One
                        // instruction does it: (d << 7) xor the cost's top byte sign-extended (an operand
                        // select of the xor) -- bits 7.. hold d or ~d by the cost's sign, bits 0..6 exponent bits,
                        // which are the same for all keys of one cost.
                        const uint32_t cbits = __float_as_uint(tf);
                        const uint32_t klo = d_sh ^ (uint32_t)((int32_t)cbits >> 24);
                        const double key = __hiloint2double((int)cbits, (int)klo);
                        __hip_atomic_fetch_min(reinterpret_cast<LDS_AS double *>(keyg + k * TC + j * 64), key,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            cur = nxt;
        }
    };
    static_assert(2 * WIN >= 2 * HALO, "two peeled groups cover the start-up rows");
    row_group(0, std::false_type{});
    row_group(WIN, std::false_type{});
    for (int base = 2 * WIN; base < NI; base += WIN) {
        // (bottom tiles: a row group all of whose output rows lie below the image has nothing left to produce)
        if (!INTERIOR && ty0 + base - 2 * HALO >= H) break;
        row_group(base, std::true_type{});
    }
}

// Tile order.  Workgroups start in the order of their index.  Rows are grouped in bands (the bands of the
// progress report, or one band for the whole frame); inside a band the tiles run column by column from the
// right edge to the left.  Tiles near the left edge skip the disparities beyond their columns and are the
// cheapest, so they fill the end of the launch, where a compute unit that gets a long tile late would leave
// the others idle; for the same reason the bottom row of tiles (clipped windows, the slower code variant)
// starts the last band.
struct TilePos { int x, y, band, band_rows; };
// The tile columns cover nx * (tile width) >= W image columns.  The slack goes to the left edge -- the first tile
// column starts this many columns left of the image -- because a tile skips every disparity beyond its rightmost
// column: at the left edge the slack shortens the disparity loops of the first tile columns, at the right edge the
// partly empty tile would cost as much as a full one.  Even, so that an image column keeps its slot in the lane.
__host__ __device__ inline int pair_x_shift(int W, int tile_width) {
    const int nx = (W + tile_width - 1) / tile_width;
    return (nx * tile_width - W) & ~1;
}
__host__ __device__ inline TilePos pair_tile_of_index(int index, int nx, int ny, int band_rows_max) {
    TilePos p;
    p.band = index / (band_rows_max * nx);
    const int first_row = p.band * band_rows_max;
    p.band_rows = band_rows_max < ny - first_row ? band_rows_max : ny - first_row;
    const int in_band = index - first_row * nx;
    p.x = nx - 1 - in_band / p.band_rows;
    p.y = first_row + in_band % p.band_rows;
    if (first_row + p.band_rows == ny) {
        if (in_band < nx) {
            p.x = nx - 1 - in_band;
            p.y = ny - 1;
        } else {
            const int j = in_band - nx, rows = p.band_rows - 1;
            p.x = nx - 1 - j / rows;
            p.y = first_row + j % rows;
        }
    }
    return p;
}

// Start value of the keys of the tiles from index `first` on (one workgroup per tile): the tiles whose disparity
// slices merge through atomicMin.
static __global__ __launch_bounds__(256) void pair_fill_split_keys_kernel(uint64_t *__restrict__ keys, int H, int W, int th,
                                                                   int twv, int nx, int ny, int band_rows_max, int first) {
    const TilePos tp = pair_tile_of_index(first + (int)blockIdx.x, nx, ny, band_rows_max);
    const int x0 = tp.x * twv - pair_x_shift(W, twv), y0 = tp.y * th;
    for (int i = threadIdx.x; i < th * twv; i += blockDim.x) {
        const int y = y0 + i / twv, x = x0 + i % twv;
        if (y < H && x >= 0 && x < W) keys[(size_t)y * W + x] = SHP_KEY_INIT;
    }
}

// (one workgroup of 8 wavefronts per compute unit -- the tile image fills the LDS -- is two wavefronts per SIMD:
//  said to the compiler, so that its scheduler does not hold back for an occupancy the kernel cannot have)
template <int RG, bool WORD64, int TH, int NW, bool QOUT, bool SRC = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fused_guided_pair_kernel(FastArgs a) {
    constexpr int WIN = 2 * RG + 1;
    constexpr int HALO = 2 * RG;
    constexpr int NI = TH + 4 * RG;
    constexpr int NJ = TH + 2 * RG;
    constexpr int TC = 128;
    constexpr int TWV = TC - 4 * RG;
    static_assert(NI % WIN == 0, "the row loop is unrolled by the window length without a tail");
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *sp = smem_raw;
    PairLds<WordT> t;
    t.mG = reinterpret_cast<double *>(sp); sp += sizeof(double) * NJ * TC;
    t.key = reinterpret_cast<uint64_t *>(sp); sp += sizeof(uint64_t) * TH * TC;
    t.L = reinterpret_cast<WordT *>(sp); sp += sizeof(WordT) * NI * TC;
    t.glo = reinterpret_cast<float *>(sp); sp += sizeof(float) * NI * TC;   // (32-bit words: one array of records with L, see PairLds)
    t.inv = reinterpret_cast<float *>(sp); sp += sizeof(float) * NJ * TC;
    t.gout = reinterpret_cast<float *>(sp); sp += sizeof(float) * (TH * TC + 16);
    t.next_d = reinterpret_cast<int *>(sp); sp += 16;
    t.k = reinterpret_cast<short *>(sp);

    // eps against the guide's range: the float32 second stage serves eps >= gate_min * max(1, max|G|^2), the
    // precise variant the range below it down to gate_low * max(1, max|G|^2) (the kernel that declines returns)
    // (one launch holds both variants: a second launch that only declines would still start every workgroup)
    if (SRC) {
        // Cost codes: the encoder left the cost unit (a power of two), the largest code and whether every cost of the
        // volume is a whole number of units in the device words a.code_stats.  The kernel serves the volume if so and
        // if the packed 16-bit window sums cannot overflow; otherwise it returns and the general kernel, enqueued
        // behind it with the opposite test, does the frame.
        if (!pair_codes_accepted(a.code_stats, RG)) return;
        // (the unit is a power of two: its float32 pattern and that of the scale factor are formed with integer
        //  arithmetic, which stays in scalar registers like the kernel arguments these values replace)
        const int e = (int)a.code_stats[0] - PAIR_CODE_FRACTION_BITS;   // -4 .. 0
        a.unit = __uint_as_float((uint32_t)(127 + e) << 23);
        a.unit_pow2 = 1;
        a.large_units = max(a.code_stats[1], 1u);
        constexpr float lean_base = RG <= 3 ? 1.0f / (float)(WIN * WIN * WIN * WIN) : 1.0f / (float)(WIN * WIN);
        a.lean_scale = RG <= 3 ? __uint_as_float(__float_as_uint(lean_base) + ((uint32_t)e << 23)) : lean_base;
        a.radius = 0;
    }
    bool precise = a.precise != 0;    // without a test (gate_min == 0): the variant the caller names
    if (a.gate_min > 0.f) {
        const bool fast_range = eps_gate_takes_fast(a.gate_eps, a.gate_min, *a.gmax);
        precise = !fast_range;
        // (no lower end, gate_low <= 0: no test -- for a guide whose square overflows float32 the product
        //  gate_low * inf would be NaN and the comparison false, and no kernel would take the frame)
        if (precise && !(a.precise != 0 && (a.gate_low <= 0.f || eps_gate_takes_fast(a.gate_eps, a.gate_low, *a.gmax)))) return;
    }
    const int lane = threadIdx.x & 63;
    const int H = a.H, W = a.W;
    const int nx = (a.W + TWV - 1) / TWV, ny = (a.H + TH - 1) / TH;
    // whole tiles first, then the slices of the split tiles
    int tile_index = (int)blockIdx.x, slice = (int)blockIdx.z, n_slices = a.n_slices;
    bool atomic_out = a.atomic_out != 0;
    if (a.split_slices > 1 && tile_index >= a.split_first) {
        const int r = tile_index - a.split_first;
        tile_index = a.split_first + r / a.split_slices;
        slice = r % a.split_slices;
        n_slices = a.split_slices;
        atomic_out = true;
    }
    const TilePos tp = pair_tile_of_index(tile_index, nx, ny, a.band_tile_rows > 0 ? a.band_tile_rows : ny);
    const int tile_x = tp.x, tile_y = tp.y, tile_band = tp.band, tile_band_rows = tp.band_rows;
    const int tx0 = tile_x * TWV - pair_x_shift(W, TWV), ty0 = tile_y * TH;
    const WordT *lw = static_cast<const WordT *>(a.lw);

    const float gmax = *a.gmax;
    int kmax = (int)(2147483647u / ((uint32_t)(WIN * WIN) * max(a.large_units, 1u)));
    kmax = min(kmax, 32767);
    int sh = 0;
    if (gmax > 0.f && gmax < 3.0e38f) sh = min(max(ilogbf((float)kmax / gmax), -120), 120);
    const float scale_f = ldexpf(1.f, sh);
    const float inv_scale_f = ldexpf(1.f, -sh);
    const double inv_scale_d = (double)inv_scale_f;

    if (threadIdx.x == 0) { t.next_d[0] = 0; t.next_d[1] = 0x7fffffff; t.next_d[2] = 0; }
    __syncthreads();
    // Tiles all of whose columns and rows (halo included) have full filter windows and valid census centres:
    // their lean passes work on scaled a and b (pair_disparity_pass), so 1 / (var + eps) is staged as it is
    // and the factor unit / (2r+1)^2 that the other tiles fold into it is applied by their other passes.
    const bool lean_geometry = !QOUT && (ty0 - HALO >= a.radius) && (ty0 + TH + HALO <= H - a.radius) &&
                               tx0 - HALO >= max(a.radius, 2 * RG) && tx0 - HALO + TC - 1 <= W - 1 - max(a.radius, 2 * RG);
    const bool scaled_tile = lean_geometry && RG <= 3 && a.unit_pow2 != 0;
    const double inv_post = scaled_tile ? (double)a.unit / (double)(WIN * WIN) : 1.0;
    bool finite_in = __float_as_uint(a.gmax[1]) == 0u;
    // Staging.  A workgroup owns its compute unit's whole LDS, so nothing else runs while the tile is
    // fetched: every global load of the tile is issued before the first result is used (one memory
    // round trip instead of one per loop iteration).  Addresses are clamped into the image so that the
    // loads need no branch; cells outside the image are zeroed when they are stored.
    {
        constexpr int NT = NW * 64;
        static_assert(NT % TC == 0, "a thread keeps its tile column in every staging step");
        constexpr int RSTEP = NT / TC;                      // tile rows between two steps of a thread
        constexpr int N1 = (NI + RSTEP - 1) / RSTEP, N2 = (NJ + RSTEP - 1) / RSTEP;
        const int c = threadIdx.x % TC, i_first = threadIdx.x / TC;
        const int gx = tx0 - HALO + c;
        const bool x_in = gx >= 0 && gx < W;
        const int gxc = min(max(gx, 0), W - 1);
        float g_in[N1];
        WordT l_in[N1];
        double m_in[N2], d_in[N2];
#pragma unroll
        for (int u = 0; u < N1; ++u) {
            const int gyc = min(max(ty0 - HALO + i_first + u * RSTEP, 0), H - 1);
            g_in[u] = a.guide[(size_t)gyc * W + gxc];
            l_in[u] = SRC ? WordT(0) : lw[(size_t)gyc * W + gxc];
        }
#pragma unroll
        for (int u = 0; u < N2; ++u) {
            const int gyc = min(max(ty0 - RG + i_first + u * RSTEP, 0), H - 1);
            m_in[u] = a.mean[(size_t)gyc * W + gxc];
            d_in[u] = a.invden[(size_t)gyc * W + gxc];
        }
#pragma unroll
        for (int u = 0; u < N1; ++u) {
            const int i = i_first + u * RSTEP;
            if (i >= NI) break;
            const int gy = ty0 - HALO + i, idx = i * TC + c;
            const bool in = x_in && gy >= 0 && gy < H;
            const float g = in ? g_in[u] : 0.f;
            finite_in = finite_in && fabsf(g) <= 3.0e38f;
            const int kk = __float2int_rn(g * scale_f);
            t.k[idx] = (short)kk;
            const float glo_v = fmaf(-(float)kk, inv_scale_f, g);
            if constexpr (WORD64) {
                t.glo[idx] = glo_v;
                t.L[idx] = in ? l_in[u] : WordT(0);
            } else {
                uint32_t *rec = reinterpret_cast<uint32_t *>(t.L) + i * 2 * TC + (c >> 1) * 4 + (c & 1);
                rec[0] = in ? (uint32_t)l_in[u] : 0u;
                rec[2] = __float_as_uint(glo_v);
            }
            const int o = i - HALO;
            if (o >= 0 && o < TH) t.gout[o * TC + c] = g;
        }
#pragma unroll
        for (int u = 0; u < N2; ++u) {
            const int j = i_first + u * RSTEP;
            if (j >= NJ) break;
            const int gy = ty0 - RG + j;
            const bool in = x_in && gy >= 0 && gy < H;
            const double mg = in ? m_in[u] : 0.0;
            // stage-1 column c is element c + r of its row (see the row loop); the first r elements, which only the
            // unused results of the first lanes read, are zero
            if (c + RG < TC) t.mG[j * TC + c + RG] = mg;
            if (c < RG) t.mG[j * TC + c] = 0.0;
            const int cyy = min(gy + RG, H - 1) - max(gy - RG, 0) + 1, cxx = min(gx + RG, W - 1) - max(gx - RG, 0) + 1;
            const float iv = !in ? 0.f : scaled_tile ? (float)d_in[u] : (float)(d_in[u] * (double)a.unit / (double)(cyy * cxx));
            if (c + RG < TC) t.inv[j * TC + c + RG] = iv;
            if (c < RG) t.inv[j * TC + c] = 0.f;
            finite_in = finite_in && fabs(mg) <= 1.0e300 && fabsf(iv) <= 3.0e38f;
        }
    }
    if (!finite_in) t.next_d[2] = 1;  // (after the barrier below it reads 0 unless some thread saw a non-finite input)
    for (int idx = threadIdx.x; idx < TH * TC; idx += NW * 64) t.key[idx] = PAIR_TILE_KEY_INIT;
    if (threadIdx.x < 16) t.gout[TH * TC + threadIdx.x] = 0.f;
    __syncthreads();

    const int R = a.radius;
    PairLane pl;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tc = 2 * lane + j;       // tile column of the slot
        const int x = tx0 - HALO + tc;     // column of the slot's cost and of its final output
        const int x1 = x - RG;             // column of the slot's stage-1 result (a, b)
        const bool x_in = x >= 0 && x < W;
        const bool x1_in = x1 >= 0 && x1 < W && tc >= RG;
        const int cx = x_in ? (min(x + RG, W - 1) - max(x - RG, 0) + 1) : 0;
        const int cx1 = x1_in ? (min(x1 + RG, W - 1) - max(x1 - RG, 0) + 1) : 0;
        pl.x[j] = x;
        pl.x_in[j] = x_in;
        pl.x_valid[j] = x_in && x >= R && x < W - R;
        pl.invAX1[j] = pair_inv_n[cx1] * (double)a.unit;
        pl.cxf[j] = (float)cx;
        pl.rcx[j] = pair_inv_n[cx];
    }
    pl.c1 = deint(max(2 * lane - RG, 0));
    pl.c1b = deint(max(2 * lane - RG, 0) + 1);
    const bool interior = (ty0 - HALO >= R) && (ty0 + TH + HALO <= H - R);
    // columns tx0 - 2r .. tx0 - 2r + 127: full filter windows, valid census centres, finite inputs
    const int xlo = tx0 - HALO, xhi = tx0 - HALO + TC - 1;
    const int xmargin = max(R, 2 * RG);
    const bool lean_tile = !QOUT && interior && xlo >= xmargin && xhi <= W - 1 - xmargin && t.next_d[2] == 0 &&
                           (RG > 3 || a.unit_pow2 != 0);   // (= lean_geometry, finite inputs, and a unit the scaled passes can fold)
    const int n_disp = a.d_end - a.d_begin;

    for (;;) {
        // Every lane takes part in the LDS atomic (the counter advances by 64 per wavefront) and no
        // lane-dependent branch surrounds the hand-out: a `lane == 0` guard here lets the optimiser
        // specialise the loop for the other lanes, for which the guarded value is a constant.
        const int dn = __builtin_amdgcn_readfirstlane(atomicAdd(t.next_d, 1)) >> 6;
        // (cost codes: d addresses the layer, through a buffer descriptor in scalar registers)
        const int d = SRC ? __builtin_amdgcn_readfirstlane(a.d_begin + dn * n_slices + slice) : a.d_begin + dn * n_slices + slice;
        if (d >= a.d_end || dn >= n_disp) break;
        // every column of the tile lies left of the first valid match: all costs are `large`
        if (!QOUT && !SRC && tx0 - HALO + (TC - 1) - d < R) {
            atomicMin(&t.next_d[1], d);
            continue;
        }
        if (RG <= 4 && precise) {   // (radius 5 has no precise variant: its rings would not fit the registers)
            if (lean_tile && (SRC || xlo - d >= R))
                pair_disparity_pass<RG, WORD64, TH, QOUT, 2, true, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
            else if (interior)
                pair_disparity_pass<RG, WORD64, TH, QOUT, 1, true, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
            else
                pair_disparity_pass<RG, WORD64, TH, QOUT, 0, true, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
        } else if (lean_tile && (SRC || xlo - d >= R))
            pair_disparity_pass<RG, WORD64, TH, QOUT, 2, false, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
        else if (interior)
            pair_disparity_pass<RG, WORD64, TH, QOUT, 1, false, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
        else
            pair_disparity_pass<RG, WORD64, TH, QOUT, 0, false, SRC>(a, t, d, lane, ty0, pl, inv_scale_d, inv_post);
    }
    __syncthreads();
    const int skipped_d = t.next_d[1];
    const float large_f = (float)a.large_units * a.unit;
    for (int idx = threadIdx.x; idx < TH * TC; idx += NW * 64) {
        int o = idx / TC, c = idx % TC;
        int gy = ty0 + o, gx = tx0 - HALO + c;
        if (c >= HALO && c < TC - HALO && gy < H && gx >= 0 && gx < W) {
            const uint64_t tile_key = t.key[o * TC + deint(c)];
            uint64_t key = pair_tile_key_to_key(tile_key);
            if (skipped_d != 0x7fffffff) {
                const uint64_t kl = make_key(large_f, (uint32_t)skipped_d);
                key = kl < key ? kl : key;
            }
            if (atomic_out) atomicMin(reinterpret_cast<unsigned long long *>(&a.keys[(size_t)gy * W + gx]),
                                        (unsigned long long)key);
            // device-scope store: written through this XCD's L2, so that a reader on another XCD sees it
            // without this workgroup flushing the whole L2 (which a device-scope fence would do)
            else if (a.progress) __hip_atomic_store(&a.keys[(size_t)gy * W + gx], key, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            else a.keys[(size_t)gy * W + gx] = key;
        }
    }
    if (a.progress) {
        // This workgroup's keys are final.  Every thread waits for its own stores to be performed, the
        // workgroup meets, and one thread counts the workgroup into its row band; the last workgroup of
        // the band publishes the band's flag.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the stores above have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            // workgroups of the band: its tiles, the split ones (all in the last band) counted once per slice
            uint32_t total = (uint32_t)tile_band_rows * nx * gridDim.z;
            if (a.split_slices > 1 && tile_band * a.band_tile_rows + tile_band_rows == ny)
                total += (uint32_t)(nx * ny - a.split_first) * (uint32_t)(a.split_slices - 1);
            if (atomicAdd(&a.progress[tile_band], 1u) + 1u == total) {
                __hip_atomic_store(&a.progress[PROGRESS_MAX_BANDS + tile_band], a.epoch, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Tile height per (guided radius, census word size): the largest TH with TH + 4r a multiple
// of 2r + 1 whose tile image fits the 160 KiB LDS.
template <int RG, bool WORD64>
struct PairTile {
    static constexpr size_t bytes(int th) {
        const size_t ni = th + 4 * RG, nj = th + 2 * RG, tc = 128, w = WORD64 ? 8 : 4;
        return sizeof(double) * nj * tc + sizeof(uint64_t) * th * tc + w * ni * tc + sizeof(float) * nj * tc +
               sizeof(float) * ni * tc + sizeof(float) * (th * tc + 16) + 16 + sizeof(short) * (ni + 1) * tc;  // one pad row: rows are fetched one ahead
    }
    static constexpr int best() {
        int th = 0;
        for (int c = 1; c < 200; ++c)
            if ((c + 4 * RG) % (2 * RG + 1) == 0 && bytes(c) <= 160 * 1024) th = c;
        return th;
    }
    static constexpr int TH = best();
};

#ifndef FP_NW
#define FP_NW 8
#endif

struct FastArgs;
// the instantiations on cost codes (guided_pair_codes.hip)
int launch_pair_codes(const FastArgs &a, int guided_radius, hipStream_t s);

template <int RG, bool WORD64, bool QOUT, bool SRC = false>
static int launch_pair_q(const FastArgs &a, hipStream_t s) {
    constexpr int TH = PairTile<RG, WORD64>::TH;
    static_assert(TH >= 2 * RG + 1, "tile too small");
    constexpr size_t smem = PairTile<RG, WORD64>::bytes(TH);
    static_assert(smem <= 160 * 1024, "tile image exceeds the LDS");
    auto kern = fused_guided_pair_kernel<RG, WORD64, TH, FP_NW, QOUT, SRC>;
    SHP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    const int tiles = ceil_div(a.W, 128 - 4 * RG) * ceil_div(a.H, TH);
    dim3 grid(a.split_slices > 1 ? a.split_first + (tiles - a.split_first) * a.split_slices : tiles, 1, a.n_slices);
    hipLaunchKernelGGL(kern, grid, dim3(FP_NW * 64), smem, s, a);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace shp

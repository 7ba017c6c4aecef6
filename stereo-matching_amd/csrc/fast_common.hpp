// Pieces shared by the fast fused census-cost + guided-filter + WTA kernels
// (guided_fast.hip: one column per lane, guided_pair.hip: two columns per lane).
#pragma once

#include "census_dev.hpp"

namespace shp {

struct FastArgs {
    const void *lw;         // left census words
    const void *rw;         // right census words
    const float *guide;     // [H][W]
    const double *mean;     // [H][W] box mean of the guide
    const double *invden;   // [H][W] 1 / (var + eps32)
    const float *gmax;      // max |guide| (device scalar)
    uint64_t *keys;         // [H][W] packed (cost, d)
    float *q_out;           // optional [D][H][W]
    int H, W, D;
    int d_begin, d_end;     // disparity range of the launch
    int n_slices;           // gridDim.z; slice z handles every n_slices-th chunk of the range
    int radius;             // census radius
    float unit;             // value of one cost unit
    int unit_pow2;          // the cost unit is a power of two (two-column kernel: it then goes into the final scale factor)
    float lean_scale;       // two-column kernel: factor from a full window's sum to the filtered cost
    uint32_t large_units;   // `large` in cost units
    int atomic_out;
    // Tail split (two-column kernel): workgroups from index split_first on are slices of tiles -- tile
    // split_first + (b - split_first) / split_slices, slice (b - split_first) % split_slices of split_slices --
    // so that the last, partly filled round of workgroups is made of short ones.  split_slices <= 1: none.
    int split_first, split_slices;
    // Row-band progress for callers that ship finished rows while the rest is searched:
    // words[b] counts the finished workgroups of band b, words[16 + b] receives `epoch` when the
    // band's rows are final in `keys`.  nullptr = no reporting.
    uint32_t *progress;
    int band_tile_rows;     // tile rows per band
    uint32_t epoch;
    // The float32 second stage is accurate to 1e-5 of a cost unit only while eps is not too small
    // against the guide's range: the kernel runs iff eps >= gate_min * max(1, gmax^2) and leaves the
    // frame to the float64 kernel (launched behind it with the opposite test) otherwise.  0: no test.
    float gate_eps, gate_min;
    float gate_low;         // two-column kernel, precise variant: lower end of its eps range (same scaling)
    int precise;            // two-column kernel: with a test, the precise variant (float64 second stage) may take
                            // the frame below the fast range; without one, it is the variant that runs
    // Two-column kernel on a caller-supplied cost volume: the costs as 16-bit codes (whole numbers of the cost
    // unit) in the layout of pair_code_index below, for code_dn disparities from code_d0 on, and the encoder's
    // device words (see pair_codes_accepted).  nullptr: costs from the census words.
    const uint16_t *codes;
    const uint32_t *code_stats;
    int code_d0, code_dn;
};

// Layout of the cost codes: [row y][group g of 128 columns][disparity][128 columns].  All disparities of 128
// neighbouring pixels lie side by side: the encoder, which reads the volume pixel by pixel (disparities innermost,
// as the reference lays it out), writes each such block as ONE contiguous piece, and the kernel that filters
// layer by layer reads 256-byte rows of it.  The last group of a row is padded to 128 columns.
constexpr int PAIR_CODE_GROUP = 128;
__host__ __device__ inline int pair_code_groups(int W) { return (W + PAIR_CODE_GROUP - 1) / PAIR_CODE_GROUP; }
__host__ __device__ inline size_t pair_code_index(int y, int x, int d, int W, int dn) {
    return (((size_t)y * pair_code_groups(W) + (size_t)(x / PAIR_CODE_GROUP)) * (size_t)dn + (size_t)d) * PAIR_CODE_GROUP +
           (size_t)(x % PAIR_CODE_GROUP);
}

// Words the cost-code encoder leaves for the kernels that read the codes:
//   [0] s: the cost unit is 2^(s - PAIR_CODE_FRACTION_BITS)   [1] largest code   [2] non-zero: some cost of the volume
//   is not a whole, non-negative number of units below 2^16 (or not finite): the codes are unusable
//   [3] (encoder's scratch) OR of the sampled costs in units of 2^-PAIR_CODE_FRACTION_BITS   [4] sample found an unusable cost
constexpr int PAIR_CODE_FRACTION_BITS = 4;
constexpr int PAIR_CODE_STATS_WORDS = 8;
#ifndef PAIR_CODES_ROWS_AHEAD
#define PAIR_CODES_ROWS_AHEAD 3   // rows the two-column kernel requests its codes ahead (5, 8, 12, 16 measured: no faster)
#endif

// Whether the two-column kernel serves the encoded volume at guided radius rg: usable codes, and the window sums
// of (rg + 1) lanes x (2 rg + 1) rows of two packed 16-bit halves cannot overflow.  The general kernel runs with
// the opposite test.
__host__ __device__ inline uint32_t pair_code_limit(int rg) { return 65535u / (uint32_t)((rg + 1) * (2 * rg + 1)); }
__device__ __forceinline__ bool pair_codes_accepted(const uint32_t *stats, int rg) {
    return stats[2] == 0u && stats[1] <= pair_code_limit(rg);
}

__device__ __forceinline__ bool eps_gate_takes_fast(float eps, float gate_min, float gmax) {
    return eps >= gate_min * fmaxf(1.f, gmax * gmax);
}

// Box mean of the guide and 1 / (var + eps32) from a clipped window's sums of G and G * G over cy x cx pixels
// (guided_filter.py:202-216, :245-248).  The means multiply by the reciprocal of the window area -- for a full
// window of `full_area` pixels a value formed once -- instead of dividing: one float64 division per pixel instead of
// three.  Every kernel that forms guide statistics goes through here, so that they agree to the last bit.
__device__ __forceinline__ void guide_mean_invden(double s1, double s2, int cy, int cx, int full_area, double rcp_full_area,
                                                  double eps32, double *mean, double *invden) {
    double ra = rcp_full_area;
    if (cy * cx != full_area) ra = 1.0 / (double)(cy * cx);
    const double m = s1 * ra;
    double var = fma(-m, m, s2 * ra);
    var = var > 0.0 ? var : (var != var ? var : 0.0);
    double den = var + eps32;
    if (den <= 0.0) den = eps32;
    *mean = m;
    *invden = 1.0 / den;
}

#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// LDS pointer whose value the optimiser cannot see through: accesses relative to it keep their
// small row offsets as instruction immediates instead of re-adding the (> 64 KiB) array base.
#define LDS_AS __attribute__((address_space(3)))
template <typename T>
__device__ __forceinline__ LDS_AS T *lds_lane_ptr(T *generic) {
    unsigned addr = (unsigned)(size_t)(LDS_AS T *)generic;
    asm volatile("" : "+v"(addr));
    return (LDS_AS T *)(size_t)addr;
}

// pure lane shift (no accumulate): lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

// order-preserving key of a float32 that is not NaN
__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

}  // namespace shp

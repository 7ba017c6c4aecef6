// Device-side cost of one (left, right) census word pair.  See census.hip for the packing.
#pragma once

#include "common.hpp"

namespace shp {

// Sum of the weighted direction fields of x, in units of cfg.unit.
// `acc` is added by the dot instruction itself (its accumulator operand).
__device__ __forceinline__ uint32_t field_sum(uint32_t x, uint32_t acc = 0u) {
    return __builtin_amdgcn_udot8(x, 0x11111111u, acc, false);
}
__device__ __forceinline__ uint32_t field_sum(uint64_t x, uint32_t acc = 0u) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return __builtin_amdgcn_udot4(hi, 0x01010101u, __builtin_amdgcn_udot4(lo, 0x01010101u, acc, false), false);
}

// The reference's own evaluation order (census.py:198-201): float32 adds over
// idx = direction * radius + (r - 1).  Used when the weights are not exactly summable.
// Branch-free form: a bit is moved to bit 30, where it alone is the float32 2.0 (or 0.0), and
// fma(2.0 or 0.0, w / 2, c) rounds w + c (or returns c) exactly as the reference's add does (or does not): three
// instructions per ring bit, no divergence, no weight loads.  Rings beyond the radius are zero bits.  Weights
// whose half is not a normal float32 (cfg.seq_fma == 0) take the loop.
template <typename WordT>
__device__ __forceinline__ float sequential_cost(WordT x, const CensusCfg &cfg) {
    constexpr int FB = (sizeof(WordT) == 8) ? 8 : 4;
    float c = 0.f;
    if (cfg.seq_fma) {
        const uint32_t half[2] = {(uint32_t)x, sizeof(WordT) == 8 ? (uint32_t)((uint64_t)x >> 32) : 0u};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int r = 1; r <= FB; ++r) {
                const int p = FB * k + (FB - r), q = p & 31;
                const uint32_t word = half[p >> 5];
                const uint32_t two = (q == 31 ? word >> 1 : word << (30 - q)) & 0x40000000u;
                c = fmaf(__uint_as_float(two), cfg.hw[r - 1], c);
            }
        }
        return c;
    }
    for (int k = 0; k < 8; ++k)
        for (int r = 1; r <= cfg.radius; ++r)
            if ((x >> (FB * k + (FB - r))) & WordT(1)) c = c + cfg.w[r - 1];
    return c;
}

template <typename WordT>
__device__ __forceinline__ float pair_cost(WordT l, WordT r, const CensusCfg &cfg) {
    WordT x = l ^ r;
    if (cfg.exact) return (float)field_sum(x) * cfg.unit;
    return sequential_cost<WordT>(x, cfg);
}

}  // namespace shp

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
template <typename WordT>
__device__ __forceinline__ float sequential_cost(WordT x, const CensusCfg &cfg) {
    const int FB = (sizeof(WordT) == 8) ? 8 : 4;
    float c = 0.f;
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

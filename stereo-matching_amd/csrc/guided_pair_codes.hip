// The two-columns-per-lane fused guided-filter + WTA kernel on a CALLER-SUPPLIED cost volume: the second call of
// the reference's two-call API, stereo.aggregate_and_wta(dsi, guide, ...) (stereo.py:94-174), where the volume
// dsi[H][W][D] float32 (D innermost, census.py:250) exists in memory and every layer dsi[:, :, d] is read back
// (stereo.py:146).
//
// The volume's layout puts a pixel's disparities side by side; the kernel wants one disparity of a tile row side by
// side.  Read in place, a wavefront's row of 128 costs touches 128 cache lines for 4 useful bytes each.  So the
// volume is first streamed ONCE, contiguously, through an encoder that transposes it through LDS into one [H][W]
// layer per disparity and stores each cost as a 16-bit whole number of the volume's cost unit ("cost code"): half
// the bytes to write and to read back, and the form in which the kernel's exact integer window sums apply.
//
// Census cost volumes are whole multiples of a power-of-two unit (census.py:58-62 with the default base weight:
// 1 for radius <= 4, 1/16 for radius 8).  Nothing about the volume is known to the callee, so a sampling pass reads
// one 4 KiB block in 64 and takes the finest unit it sees (never finer than 2^-4); the encoder then checks EVERY
// cost against that unit (whole, non-negative, below 2^16 units, finite) and records the largest code.  A volume
// that fails the check, or whose codes could overflow the kernel's packed 16-bit window sums, is left to the
// general float64 kernel, which is enqueued behind the two-column one with the opposite test on the same device
// words: exactly one of them does the frame, and nothing synchronises with the host.
#include "guided_pair_kernel.hpp"

namespace shp {

namespace {

constexpr int ENC_THREADS = 512;    // encoder: one workgroup per (row, group of 128 columns)
constexpr int SAMPLE_THREADS = 256;
constexpr int SAMPLE_FLOATS = 1024;   // floats per sampled block
constexpr int SAMPLE_STRIDE = 64;     // one block in this many is sampled

// cost in units of 2^-PAIR_CODE_FRACTION_BITS if it is a whole, non-negative, finite number of them below 2^20
__device__ __forceinline__ bool cost_in_sixteenths(float v, uint32_t *out) {
    const float u = v * (float)(1 << PAIR_CODE_FRACTION_BITS);   // exact (power of two) unless it overflows to inf
    const bool ok = u >= 0.f && u < 1048576.f && u == truncf(u);   // (NaN fails every comparison)
    *out = ok ? (uint32_t)u : 0u;
    return ok;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void cost_unit_sample_kernel(const float *__restrict__ dsi, size_t n_floats,
                                                                       uint32_t *__restrict__ stats) {
    uint32_t bits = 0u, bad = 0u;
    const size_t n_blocks = (n_floats + SAMPLE_FLOATS - 1) / SAMPLE_FLOATS;
    for (size_t b = (size_t)blockIdx.x * SAMPLE_STRIDE; b < n_blocks; b += (size_t)gridDim.x * SAMPLE_STRIDE) {
        for (int k = threadIdx.x; k < SAMPLE_FLOATS; k += SAMPLE_THREADS) {
            const size_t i = b * SAMPLE_FLOATS + k;
            if (i >= n_floats) break;
            uint32_t u;
            if (cost_in_sixteenths(dsi[i], &u)) bits |= u; else bad = 1u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bits |= (uint32_t)__shfl_xor((int)bits, off, 64);
        bad |= (uint32_t)__shfl_xor((int)bad, off, 64);
    }
    // (atomics on one word serialise -- 32 000 wavefronts spent 0.37 ms on them: a wavefront reports only bits the
    //  word does not hold yet; a stale read costs a needless atomic at worst)
    if ((threadIdx.x & 63) == 0) {
        if (bits & ~__hip_atomic_load(&stats[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats[3], bits);
        if (bad) atomicOr(&stats[4], 1u);
    }
}

// stats[0] = s (unit 2^(s - 4)) from the sample's OR: the number of trailing zero bits all sampled costs share
__global__ void cost_unit_from_sample_kernel(uint32_t *stats) {
    const uint32_t bits = stats[3];
    stats[0] = bits ? (uint32_t)min(__ffs((int)bits) - 1, PAIR_CODE_FRACTION_BITS) : (uint32_t)PAIR_CODE_FRACTION_BITS;
    stats[1] = 0u;
    stats[2] = stats[4];
}

// One workgroup: 128 neighbouring pixels of a row (a group) x all disparities, a contiguous slab of the volume.  The
// slab is read with consecutive lanes on consecutive 16-byte pieces (4-byte pieces where the slab is not 16-byte
// aligned), eight loads in flight per thread before the first is used, encoded, written to LDS pixel-major, and
// written out as [disparity][128 columns]: one contiguous piece of the code array (pair_code_index).
__global__ __launch_bounds__(ENC_THREADS) void cost_encode_kernel(const float *__restrict__ dsi, int W, int D, int d_begin,
                                                                  int d_end, uint16_t *__restrict__ codes,
                                                                  uint32_t *__restrict__ stats, float rcp_d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t *t = reinterpret_cast<uint16_t *>(smem_raw);   // [128][Dp], Dp odd
    const int Dn = d_end - d_begin, Dp = Dn | 1;
    const int y = blockIdx.y, g = blockIdx.x, x_first = g * PAIR_CODE_GROUP;
    const int np = min(PAIR_CODE_GROUP, W - x_first);
    const uint32_t s = stats[0];
    const float to_units = ldexpf(1.f, PAIR_CODE_FRACTION_BITS - (int)s);
    const float *slab = dsi + ((size_t)y * W + x_first) * (size_t)D;
    const int n = np * D;
    uint32_t bad = 0u, top = 0u;
    auto put = [&](int px, int d, float v) {
        if (d < d_begin || d >= d_end) return;
        const float u = v * to_units;
        const bool ok = u >= 0.f && u < 65536.f && u == truncf(u);   // (NaN fails every comparison)
        const uint32_t c = ok ? (uint32_t)u : 0u;
        bad |= ok ? 0u : 1u;
        top = max(top, c);
        t[px * Dp + (d - d_begin)] = (uint16_t)c;
    };
    // (pixel, d) of flat index i: i < 128 * D fits the float's 24 bits whenever D < 2^17
    auto locate = [&](int i, int *px, int *d) {
        int q = (int)(((float)i + 0.5f) * rcp_d);
        int r = i - q * D;
        if (r < 0) { --q; r += D; } else if (r >= D) { ++q; r -= D; }
        *px = q; *d = r;
    };
    constexpr int U = 8;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(slab) & 15u) == 0) {
        const float4 *slab4 = reinterpret_cast<const float4 *>(slab);
        const int n4 = n >> 2;
        for (int base = threadIdx.x; base < n4; base += ENC_THREADS * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * ENC_THREADS;
                v[u] = idx < n4 ? slab4[idx] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * ENC_THREADS;
                if (idx >= n4) break;
                int px, d;
                locate(idx * 4, &px, &d);
                const float c[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    put(px, d, c[k]);
                    if (++d == D) { d = 0; ++px; }
                }
            }
        }
        done = n4 << 2;
    }
    for (int base = done + threadIdx.x; base < n; base += ENC_THREADS * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * ENC_THREADS;
            v[u] = i < n ? slab[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * ENC_THREADS;
            if (i >= n) break;
            int px, d;
            locate(i, &px, &d);
            put(px, d, v[u]);
        }
    }
    __syncthreads();
    // a wavefront writes the 128 codes of one disparity per step (256 contiguous bytes); columns beyond the image
    // (the padding of a row's last group) are written as zero
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pa = 2 * lane, pb = 2 * lane + 1;
    uint32_t *dst = reinterpret_cast<uint32_t *>(codes + pair_code_index(y, x_first, 0, W, Dn));
    for (int d = wave; d < Dn; d += ENC_THREADS / 64) {
        const uint32_t lo = pa < np ? (uint32_t)t[pa * Dp + d] : 0u, hi = pb < np ? (uint32_t)t[pb * Dp + d] : 0u;
        dst[d * (PAIR_CODE_GROUP / 2) + lane] = lo | (hi << 16);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad |= (uint32_t)__shfl_xor((int)bad, off, 64);
        top = max(top, (uint32_t)__shfl_xor((int)top, off, 64));
    }
    // (atomics on one word serialise: 400 000 wavefronts of a frame would spend milliseconds on them -- only a
    //  wavefront that raises the largest code seen so far reports it; a stale read costs a needless atomic at worst)
    if ((threadIdx.x & 63) == 0) {
        if (top > __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[1], top);
        if (bad) atomicOr(&stats[2], 1u);
    }
}

template <int RG>
int launch_pair_codes_t(const FastArgs &a, hipStream_t s) {
    return launch_pair_q<RG, false, false, true>(a, s);
}

}  // namespace

size_t cost_codes_bytes(int H, int W, int n_disp) {
    return align_up((size_t)H * pair_code_groups(W) * (size_t)std::max(n_disp, 0) * PAIR_CODE_GROUP * 2, 256);
}

int launch_encode_cost_codes(const float *d_dsi, int H, int W, int D, int d_begin, int d_end, uint16_t *d_codes,
                             uint32_t *d_stats, hipStream_t s) {
    if (H <= 0 || W <= 0 || d_begin >= d_end) return STEREO_OK;
    const int Dn = d_end - d_begin;
    const size_t smem = (size_t)PAIR_CODE_GROUP * (Dn | 1) * 2;
    if (D >= (1 << 17) || smem > 160 * 1024) {
        set_error("cost codes: at most %d disparities per launch", (160 * 1024 / (PAIR_CODE_GROUP * 2)) - 1);
        return STEREO_EUNSUPPORTED;
    }
    SHP_HIP(hipMemsetAsync(d_stats, 0, sizeof(uint32_t) * PAIR_CODE_STATS_WORDS, s));
    const size_t n_floats = (size_t)H * W * (size_t)D;
    const size_t sample_blocks = (n_floats + (size_t)SAMPLE_FLOATS * SAMPLE_STRIDE - 1) / ((size_t)SAMPLE_FLOATS * SAMPLE_STRIDE);
    hipLaunchKernelGGL(cost_unit_sample_kernel, dim3((unsigned)std::min<size_t>(sample_blocks, 8192)), dim3(SAMPLE_THREADS), 0, s,
                       d_dsi, n_floats, d_stats);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(cost_unit_from_sample_kernel, dim3(1), dim3(1), 0, s, d_stats);
    SHP_LAUNCH_CHECK();
    SHP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cost_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    hipLaunchKernelGGL(cost_encode_kernel, dim3((unsigned)pair_code_groups(W), (unsigned)H), dim3(ENC_THREADS), smem, s,
                       d_dsi, W, D, d_begin, d_end, d_codes, d_stats, 1.0f / (float)D);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

int launch_pair_codes(const FastArgs &a, int guided_radius, hipStream_t s) {
    switch (guided_radius) {
        case 1: return launch_pair_codes_t<1>(a, s);
        case 2: return launch_pair_codes_t<2>(a, s);
        case 3: return launch_pair_codes_t<3>(a, s);
        case 4: return launch_pair_codes_t<4>(a, s);
        default: set_error("two-column kernel on cost codes: guided radius 1..4"); return STEREO_EUNSUPPORTED;
    }
}

}  // namespace shp

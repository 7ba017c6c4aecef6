// Issue-rate microbenchmark for the vector instructions the fused stereo kernels are made of.
// For each instruction: cycles per wave-instruction seen by one wave, with 1, 2 and 4 waves per SIMD
// (s_memtime around a loop of independent instructions).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4000
#define REP8(x) x x x x x x x x

#define DEFKERNEL(NAME, TYPE, INIT, ASM)                                                         \
    __global__ void NAME(unsigned long long *out, TYPE seed) {                                   \
        TYPE a0 = INIT + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
             a6 = a0 + 6, a7 = a0 + 7;                                                           \
        __syncthreads();                                                                         \
        unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int it = 0; it < ITERS; ++it) {                                                     \
            asm volatile(REP8(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7))            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                           "+v"(a7));                                                            \
        }                                                                                        \
        unsigned long long t1 = __builtin_readcyclecounter();                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == (TYPE)12345) out[0] = 1;                     \
    }

#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_ADD32(i) "v_add_f32 %" #i ", %" #i ", %" #i "\n"
#define A_ADDDPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADDDPPROW(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADDU32DPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADDU32(i) "v_add_u32 %" #i ", %" #i ", %" #i "\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_MUL24(i) "v_mul_i32_i24 %" #i ", %" #i ", %" #i "\n"
#define A_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_DOT8(i) "v_dot8_u32_u4 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_CVTF32U32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %" #i ", vcc\n"
#define A_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %" #i ", s[20:21]\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_MIN32(i) "v_min_f32 %" #i ", %" #i ", %" #i "\n"
#define A_CMP(i) "v_cmp_o_f32 vcc, %" #i ", %" #i "\n"
#define A_CMPCND(i) "v_cmp_o_f32 vcc, %" #i ", %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %" #i ", vcc\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %" #i ", %" #i " bitop3:0x36\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %" #i "\n"
#define A_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %" #i "\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %" #i ", %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %" #i "\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %" #i "\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %" #i "\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %" #i "\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %" #i "\n"
#define A_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define A_MOV64(i) "v_mov_b64 %" #i ", %" #i "\n"

DEFKERNEL(k_fma32, float, 1.0f, A_FMA32)
DEFKERNEL(k_add32, float, 1.0f, A_ADD32)
DEFKERNEL(k_adddpp, float, 1.0f, A_ADDDPP)
DEFKERNEL(k_adddpprow, float, 1.0f, A_ADDDPPROW)
DEFKERNEL(k_movdpp, float, 1.0f, A_MOVDPP)
DEFKERNEL(k_addu32dpp, unsigned, 1u, A_ADDU32DPP)
DEFKERNEL(k_addu32, unsigned, 1u, A_ADDU32)
DEFKERNEL(k_add3, unsigned, 1u, A_ADD3)
DEFKERNEL(k_mul24, int, 1, A_MUL24)
DEFKERNEL(k_mad24, int, 1, A_MAD24)
DEFKERNEL(k_dot8, unsigned, 1u, A_DOT8)
DEFKERNEL(k_cvtf32u32, unsigned, 1u, A_CVTF32U32)
DEFKERNEL(k_cndmask, unsigned, 1u, A_CNDMASK)
DEFKERNEL(k_cndmask64, unsigned, 1u, A_CNDMASK64)
DEFKERNEL(k_andor, unsigned, 1u, A_ANDOR)
DEFKERNEL(k_bfi, unsigned, 1u, A_BFI)
DEFKERNEL(k_min32, float, 1.0f, A_MIN32)
DEFKERNEL(k_cmp, float, 1.0f, A_CMP)
DEFKERNEL(k_cmpcnd, float, 1.0f, A_CMPCND)
DEFKERNEL(k_bitop3, unsigned, 1u, A_BITOP3)
DEFKERNEL(k_ashr, int, 1, A_ASHR)
DEFKERNEL(k_xor, unsigned, 1u, A_XOR)
DEFKERNEL(k_subf, float, 1.0f, A_SUBF)
DEFKERNEL(k_mulf, float, 1.0f, A_MULF)
DEFKERNEL(k_fmac, float, 1.0f, A_FMAC)
DEFKERNEL(k_mov, float, 1.0f, A_MOV)
DEFKERNEL(k_lshladd64, double, 1.0, A_LSHLADD64)
DEFKERNEL(k_fma64, double, 1.0, A_FMA64)
DEFKERNEL(k_add64, double, 1.0, A_ADD64)
DEFKERNEL(k_mul64, double, 1.0, A_MUL64)
DEFKERNEL(k_pkfma, double, 1.0, A_PKFMA)
DEFKERNEL(k_pkadd, double, 1.0, A_PKADD)
DEFKERNEL(k_pkmul, double, 1.0, A_PKMUL)
DEFKERNEL(k_mov64, double, 1.0, A_MOV64)

// conversions: source and destination widths differ, so they get their own bodies
#define CVTKERNEL(NAME, DT, ST, ASM)                                                              \
    __global__ void NAME(unsigned long long *out, ST seed) {                                      \
        ST s0 = seed + 1, s1 = seed + 2, s2 = seed + 3, s3 = seed + 4;                            \
        DT d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;                        \
        __syncthreads();                                                                         \
        unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int it = 0; it < ITERS; ++it) {                                                     \
            asm volatile(REP8(ASM " %0, %8\n" ASM " %1, %9\n" ASM " %2, %10\n" ASM " %3, %11\n"    \
                              ASM " %4, %8\n" ASM " %5, %9\n" ASM " %6, %10\n" ASM " %7, %11\n")    \
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7) \
                         : "v"(s0), "v"(s1), "v"(s2), "v"(s3));                                   \
        }                                                                                        \
        unsigned long long t1 = __builtin_readcyclecounter();                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == (DT)12345) out[0] = 1;                       \
    }
CVTKERNEL(k_cvt_f64_f32, double, float, "v_cvt_f64_f32")
CVTKERNEL(k_cvt_f32_f64, float, double, "v_cvt_f32_f64")
CVTKERNEL(k_cvt_f64_i32, double, int, "v_cvt_f64_i32")
CVTKERNEL(k_cvt_f64_u32, double, unsigned, "v_cvt_f64_u32")

template <typename K, typename T>
static void run(const char *name, K kern, T seed) {
    unsigned long long *d;
    hipMalloc(&d, 256 * 16 * 8);
    printf("%-16s", name);
    for (int wps : {1, 2, 4}) {
        int threads = 64 * 4 * wps;
        hipMemset(d, 0, 256 * 16 * 8);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, seed);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * wps);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        double per = mx / (double)(ITERS * 64);
        printf("  wps=%d: %6.2f ticks/instr/wave (%5.2f per SIMD-instr)", wps, per, per / wps);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on this part; calibrate against v_fma_f32
    run("v_fma_f32", k_fma32, 0.0f);
    run("v_add_f32", k_add32, 0.0f);
    run("v_add_f32_dpp", k_adddpp, 0.0f);
    run("v_add_dpp_row", k_adddpprow, 0.0f);
    run("v_mov_b32_dpp", k_movdpp, 0.0f);
    run("v_add_u32_dpp", k_addu32dpp, 0u);
    run("v_add_u32", k_addu32, 0u);
    run("v_add3_u32", k_add3, 0u);
    run("v_mul_i32_i24", k_mul24, 0);
    run("v_mad_i32_i24", k_mad24, 0);
    run("v_dot8_u32_u4", k_dot8, 0u);
    run("v_cvt_f32_u32", k_cvtf32u32, 0u);
    run("v_cndmask_b32", k_cndmask, 0u);
    run("v_cndmask_e64", k_cndmask64, 0u);
    run("v_and_or_b32", k_andor, 0u);
    run("v_bfi_b32", k_bfi, 0u);
    run("v_min_f32", k_min32, 0.0f);
    run("v_cmp_o_f32", k_cmp, 0.0f);
    run("cmp+cndmask", k_cmpcnd, 0.0f);
    run("v_bitop3_b32", k_bitop3, 0u);
    run("v_ashrrev_i32", k_ashr, 0);
    run("v_xor_b32", k_xor, 0u);
    run("v_sub_f32", k_subf, 0.0f);
    run("v_mul_f32", k_mulf, 0.0f);
    run("v_fmac_f32", k_fmac, 0.0f);
    run("v_mov_b32", k_mov, 0.0f);
    run("v_lshl_add_u64", k_lshladd64, 0.0);
    run("v_fma_f64", k_fma64, 0.0);
    run("v_add_f64", k_add64, 0.0);
    run("v_mul_f64", k_mul64, 0.0);
    run("v_pk_fma_f32", k_pkfma, 0.0);
    run("v_pk_add_f32", k_pkadd, 0.0);
    run("v_pk_mul_f32", k_pkmul, 0.0);
    run("v_mov_b64", k_mov64, 0.0);
    run("v_cvt_f64_f32", k_cvt_f64_f32, 0.0f);
    run("v_cvt_f32_f64", k_cvt_f32_f64, 0.0);
    run("v_cvt_f64_i32", k_cvt_f64_i32, 0);
    run("v_cvt_f64_u32", k_cvt_f64_u32, 0u);
    return 0;
}

// Microbenchmark: issue cost of the VALU instructions the streaming kernel is made of, on gfx950 at the kernel's
// occupancy (768-thread workgroups, 2 per CU = 6 waves per SIMD).  Each kernel runs 8 independent chains of one
// instruction; the result is wall time per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o build/bench_valu_issue tools/bench_valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define DEF_KERNEL3(NAME, INSTR)                                                                                   \
    __global__ __launch_bounds__(768) void NAME(float* out, float a, float b) {                                     \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < N_ITER; i++) {                                                                          \
            asm volatile(INSTR " %0, %0, %8, %9\n" INSTR " %1, %1, %8, %9\n" INSTR " %2, %2, %8, %9\n" INSTR " %3, %3, %8, %9\n" \
                         INSTR " %4, %4, %8, %9\n" INSTR " %5, %5, %8, %9\n" INSTR " %6, %6, %8, %9\n" INSTR " %7, %7, %8, %9\n" \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                         \
    }
#define DEF_KERNEL2(NAME, INSTR)                                                                                   \
    __global__ __launch_bounds__(768) void NAME(float* out, float a, float b) {                                     \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < N_ITER; i++) {                                                                          \
            asm volatile(INSTR " %0, %0, %8\n" INSTR " %1, %1, %8\n" INSTR " %2, %2, %8\n" INSTR " %3, %3, %8\n"     \
                         INSTR " %4, %4, %8\n" INSTR " %5, %5, %8\n" INSTR " %6, %6, %8\n" INSTR " %7, %7, %8\n"     \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                         \
    }
#define DEF_KERNEL1(NAME, INSTR)                                                                                   \
    __global__ __launch_bounds__(768) void NAME(float* out, float a, float b) {                                     \
        float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < N_ITER; i++) {                                                                          \
            asm volatile(INSTR " %0, %0\n" INSTR " %1, %1\n" INSTR " %2, %2\n" INSTR " %3, %3\n"                     \
                         INSTR " %4, %4\n" INSTR " %5, %5\n" INSTR " %6, %6\n" INSTR " %7, %7\n"                     \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                         \
    }
// compare into vcc + select on it: the pair the traversal logic is made of
__global__ __launch_bounds__(768) void k_cmp_cndmask(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                     "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ __launch_bounds__(768) void k_pk_fma(float* out, float a, float b) {
    float t = threadIdx.x;
    v2f x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7}, x4 = {t + 8, t + 9}, x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15};
    v2f aa = {a, a}, bb = {b, b};
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(768) void k_pk_mul(float* out, float a, float b) {
    float t = threadIdx.x;
    v2f x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7}, x4 = {t + 8, t + 9}, x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15};
    v2f aa = {a, a};
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa));
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
DEF_KERNEL3(k_fma, "v_fma_f32")
DEF_KERNEL2(k_fmac, "v_fmac_f32")
DEF_KERNEL2(k_mul, "v_mul_f32")
DEF_KERNEL2(k_add, "v_add_f32")
DEF_KERNEL2(k_min, "v_min_f32")
DEF_KERNEL3(k_max3, "v_max3_f32")
DEF_KERNEL2(k_xor, "v_xor_b32")
DEF_KERNEL2(k_addu, "v_add_u32")
DEF_KERNEL2(k_lshl, "v_lshlrev_b32")
DEF_KERNEL3(k_alignbit, "v_alignbit_b32")
DEF_KERNEL2(k_mullo, "v_mul_lo_u32")
DEF_KERNEL2(k_mulhi, "v_mul_hi_u32")
DEF_KERNEL2(k_mul24, "v_mul_u32_u24")
DEF_KERNEL1(k_mov, "v_mov_b32")
DEF_KERNEL1(k_rcp, "v_rcp_f32")
DEF_KERNEL1(k_sqrt, "v_sqrt_f32")
DEF_KERNEL1(k_cvt, "v_cvt_f32_u32")
DEF_KERNEL3(k_divfixup, "v_div_fixup_f32")
DEF_KERNEL2(k_sub, "v_sub_f32")
DEF_KERNEL2(k_max, "v_max_f32")
DEF_KERNEL3(k_med3, "v_med3_f32")
DEF_KERNEL2(k_and, "v_and_b32")
DEF_KERNEL2(k_or, "v_or_b32")
DEF_KERNEL3(k_bfi, "v_bfi_b32")
DEF_KERNEL2(k_subu, "v_sub_u32")
DEF_KERNEL3(k_lshl_add, "v_lshl_add_u32")
DEF_KERNEL3(k_add3, "v_add3_u32")
DEF_KERNEL3(k_and_or, "v_and_or_b32")
DEF_KERNEL3(k_xad, "v_xad_u32")
DEF_KERNEL2(k_lshr, "v_lshrrev_b32")
DEF_KERNEL2(k_ashr, "v_ashrrev_i32")
DEF_KERNEL3(k_bfe, "v_bfe_u32")
DEF_KERNEL3(k_perm, "v_perm_b32")
DEF_KERNEL3(k_mad24, "v_mad_u32_u24")
DEF_KERNEL2(k_minu, "v_min_u32")
DEF_KERNEL1(k_fract, "v_fract_f32")
DEF_KERNEL1(k_floor, "v_floor_f32")
DEF_KERNEL1(k_cvtu, "v_cvt_u32_f32")
DEF_KERNEL1(k_rsq, "v_rsq_f32")
DEF_KERNEL1(k_sin, "v_sin_f32")
DEF_KERNEL2(k_ldexp, "v_ldexp_f32")
DEF_KERNEL2(k_mul_legacy, "v_mul_legacy_f32")
// compare alone (result to an SGPR pair) and select alone (fixed condition)
__global__ __launch_bounds__(768) void k_cmp(float* out, float a, float b) {
    float x0 = threadIdx.x;
    unsigned long long acc = 0;
    for (int i = 0; i < N_ITER; i++) {
        unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
        asm volatile("v_cmp_lt_f32 %0, %8, %9\n v_cmp_lt_f32 %1, %9, %8\n v_cmp_le_f32 %2, %8, %9\n v_cmp_le_f32 %3, %9, %8\n"
                     "v_cmp_gt_f32 %4, %8, %9\n v_cmp_gt_f32 %5, %9, %8\n v_cmp_ge_f32 %6, %8, %9\n v_cmp_ge_f32 %7, %9, %8\n"
                     : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7) : "v"(x0), "v"(a));
        acc ^= m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(acc & 0xff);
}
__global__ __launch_bounds__(768) void k_cndmask(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long m = (threadIdx.x & 64) ? 0x5555555555555555ull : 0x3333333333333333ull;
    m = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                     "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// v_mad_u64_u32: full 64-bit product (hi and lo of a Philox multiply in one instruction)
__global__ __launch_bounds__(768) void k_mad64(float* out, float a, float b) {
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    unsigned k = __float_as_uint(a);
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                     "v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(k) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)((p0 ^ p1 ^ p2 ^ p3) & 0xffff);
}
// LDS reads: 16-B and 4-B per lane, random rows of a 32 KB table (the node fetch pattern)
__global__ __launch_bounds__(768) void k_ds_read_b128(float* out, float a, float b) {
    __shared__ float4 tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 768) tab[i] = make_float4(i, a, b, 1.0f);
    __syncthreads();
    unsigned idx = threadIdx.x * 2654435761u;
    float acc = 0;
    for (int i = 0; i < N_ITER; i++) {
        float4 v0 = tab[(idx >> 8) & 2047], v1 = tab[(idx >> 12) & 2047];
        acc += v0.x + v1.y;
        idx = idx * 1664525u + 1013904223u + (unsigned)v0.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}


// scalar ALU: is it hidden behind the vector issue or a second budget?  (a) SALU alone, (b) 8 v_fma + N s_and_b64 interleaved
__global__ __launch_bounds__(768) void k_salu(float* out, float a, float b) {
    unsigned long long m0 = __builtin_amdgcn_readfirstlane(threadIdx.x) + 1, m1 = m0 * 3, m2 = m0 * 5, m3 = m0 * 7;
    unsigned long long k = 0x00ff00ff00ff00ffull | (unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(a));
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("s_and_b64 %0, %0, %4\n s_or_b64 %1, %1, %4\n s_xor_b64 %2, %2, %4\n s_andn2_b64 %3, %3, %4\n"
                     "s_or_b64 %0, %0, %4\n s_and_b64 %1, %1, %4\n s_andn2_b64 %2, %2, %4\n s_xor_b64 %3, %3, %4\n"
                     : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "s"(k) : "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)((m0 ^ m1 ^ m2 ^ m3) & 0xff);
}
#define DEF_MIX(NAME, SALU_TEXT)                                                                                    \
    __global__ __launch_bounds__(768) void NAME(float* out, float a, float b) {                                     \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        unsigned long long m0 = __builtin_amdgcn_readfirstlane(threadIdx.x) + 1, m1 = m0 * 3;                       \
        unsigned long long k = 0x00ff00ff00ff00ffull | (unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(a)); \
        for (int i = 0; i < N_ITER; i++) {                                                                          \
            asm volatile("v_fma_f32 %0, %0, %10, %11\n" SALU_TEXT "v_fma_f32 %1, %1, %10, %11\n" SALU_TEXT "v_fma_f32 %2, %2, %10, %11\n" SALU_TEXT \
                         "v_fma_f32 %3, %3, %10, %11\n" SALU_TEXT "v_fma_f32 %4, %4, %10, %11\n" SALU_TEXT "v_fma_f32 %5, %5, %10, %11\n" SALU_TEXT \
                         "v_fma_f32 %6, %6, %10, %11\n" SALU_TEXT "v_fma_f32 %7, %7, %10, %11\n" SALU_TEXT                 \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(m0), "+s"(m1) \
                         : "v"(a), "v"(b), "s"(k) : "scc");                                                         \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)((m0 ^ m1) & 0xff); \
    }
DEF_MIX(k_mix_8v_4s, "s_and_b64 %8, %8, %12\n")

DEF_MIX(k_mix_8v_8s, "s_and_b64 %8, %8, %12\n s_or_b64 %9, %9, %12\n")
DEF_MIX(k_mix_8v_0s, "")

static int g_clock_khz = 2400000;
template <typename K> static void run(const char* name, K kern, float* out, int blocks_per_cu, int instr_per_iter = 8) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    kern<<<grid, 768>>>(out, 0.999f, 0.001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<grid, 768>>>(out, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = blocks_per_cu * 12.0 / 4.0;
    double instr = waves_per_simd * N_ITER * (double)instr_per_iter;
    printf("%-18s %.3f ms  %.3f ns per wave-instruction per SIMD  (%.2f cycles at %.2f GHz)\n", name, ms, ms * 1e6 / instr, ms * 1e6 / instr * g_clock_khz * 1e-6, g_clock_khz * 1e-6);
}
int main() {
    float* out; (void)hipMalloc(&out, sizeof(float) * 768 * 512);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    g_clock_khz = prop.clockRate;
    printf("device %s, %d CUs, clockRate %d kHz; 2 x 768-thread workgroups per CU (6 waves per SIMD)\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int b = 2;
    run("v_mov_b32", k_mov, out, b);
    run("v_add_f32", k_add, out, b);
    run("v_mul_f32", k_mul, out, b);
    run("v_fmac_f32", k_fmac, out, b);
    run("v_fma_f32", k_fma, out, b);
    run("v_min_f32", k_min, out, b);
    run("v_max3_f32", k_max3, out, b);
    run("v_cmp+v_cndmask", k_cmp_cndmask, out, b);
    run("v_xor_b32", k_xor, out, b);
    run("v_add_u32", k_addu, out, b);
    run("v_lshlrev_b32", k_lshl, out, b);
    run("v_alignbit_b32", k_alignbit, out, b);
    run("v_mul_u32_u24", k_mul24, out, b);
    run("v_mul_lo_u32", k_mullo, out, b);
    run("v_mul_hi_u32", k_mulhi, out, b);
    run("v_cvt_f32_u32", k_cvt, out, b);
    run("v_rcp_f32", k_rcp, out, b);
    run("v_sqrt_f32", k_sqrt, out, b);
    run("v_div_fixup_f32", k_divfixup, out, b);
    run("v_sub_f32", k_sub, out, b);
    run("v_max_f32", k_max, out, b);
    run("v_med3_f32", k_med3, out, b);
    run("v_and_b32", k_and, out, b);
    run("v_or_b32", k_or, out, b);
    run("v_bfi_b32", k_bfi, out, b);
    run("v_sub_u32", k_subu, out, b);
    run("v_lshl_add_u32", k_lshl_add, out, b);
    run("v_add3_u32", k_add3, out, b);
    run("v_and_or_b32", k_and_or, out, b);
    run("v_xad_u32", k_xad, out, b);
    run("v_lshrrev_b32", k_lshr, out, b);
    run("v_ashrrev_i32", k_ashr, out, b);
    run("v_bfe_u32", k_bfe, out, b);
    run("v_perm_b32", k_perm, out, b);
    run("v_mad_u32_u24", k_mad24, out, b);
    run("v_min_u32", k_minu, out, b);
    run("v_fract_f32", k_fract, out, b);
    run("v_floor_f32", k_floor, out, b);
    run("v_cvt_u32_f32", k_cvtu, out, b);
    run("v_rsq_f32", k_rsq, out, b);
    run("v_sin_f32", k_sin, out, b);
    run("v_ldexp_f32", k_ldexp, out, b);
    run("v_mul_legacy_f32", k_mul_legacy, out, b);
    run("v_cmp_*_f32 ->sgpr", k_cmp, out, b);
    run("v_cndmask_b32", k_cndmask, out, b);
    run("v_mad_u64_u32", k_mad64, out, b);
    run("ds_read_b128 x2+", k_ds_read_b128, out, b, 2);
    run("SALU b64 logic x8", k_salu, out, b);
    run("8 v_fma (+0 salu)", k_mix_8v_0s, out, b);
    run("8 v_fma + 8 salu", k_mix_8v_4s, out, b);
    run("8 v_fma + 16 salu", k_mix_8v_8s, out, b);
    run("v_pk_mul_f32", k_pk_mul, out, b);
    run("v_pk_fma_f32", k_pk_fma, out, b);
    return 0;
}

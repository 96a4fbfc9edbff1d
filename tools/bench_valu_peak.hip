// Microbenchmark: what issue rate of full-rate fp32 vector instructions a SIMD of gfx950 really sustains, without the loop overhead that
// tools/bench_valu_issue.hip carries (3 scalar instructions per 8 vector ones): 8 independent chains x 16 = 128 v_fma_f32 per loop iteration,
// at 4 / 6 / 8 waves per SIMD, with the shader clock measured during the run (s_memtime against the 100-MHz wall clock).
//   hipcc --offload-arch=gfx950 -O3 -o build/bench_valu_peak tools/bench_valu_peak.hip && build/bench_valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define R8(S) S S S S S S S S
#define R16(S) R8(S) R8(S)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fma(float* out, unsigned long long* clk, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < N_ITER; i++) {
        asm volatile(R16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int BLOCK> static void run(float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 2;
    k_fma<BLOCK><<<grid, BLOCK>>>(out, clk, 0.999f, 0.001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_fma<BLOCK><<<grid, BLOCK>>>(out, clk, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double waves_per_simd = 2.0 * (BLOCK / 64) / 4.0;
    const double instr = waves_per_simd * N_ITER * 128.0;          // wave-instructions per SIMD
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memtime ticks per second of the 100-MHz wall clock
    printf("%4d-thread workgroups x 2 per CU = %.0f waves per SIMD: %.3f ms, %.3f ns per wave-instruction per SIMD = %.2f cycles at the nominal %.2f GHz, "
           "s_memtime rate %.3f GHz\n", BLOCK, waves_per_simd, ms, ms * 1e6 / instr, ms * 1e6 / instr * prop.clockRate * 1e-6, prop.clockRate * 1e-6, ghz);
}
int main() {
    float* out; (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    run<256>(out, clk); run<512>(out, clk); run<768>(out, clk); run<1024>(out, clk);
    return 0;
}

// Developer micro-benchmark (not part of the product): issue cost of the fp64 instructions the CImg resize restatement
// uses, per wave instruction on one SIMD (independent chains, 8 waves per SIMD resident).
// hipcc --offload-arch=gfx950 -O3 f64_rates.hip -o f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kIters = 4096, kChains = 8;
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, unsigned seed) {
  double a[kChains]; unsigned u[kChains]; float f[kChains];
  for (int i = 0; i < kChains; ++i) { a[i] = 1.0 + (threadIdx.x + i) * 1e-3; u[i] = threadIdx.x * 7 + i + seed; f[i] = (float)a[i]; }
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kChains; ++i) {
      if (OP == 0) a[i] = a[i] * 1.0000001;                       // v_mul_f64
      if (OP == 1) a[i] = a[i] + 1e-9;                            // v_add_f64
      if (OP == 2) { a[i] = (double)u[i]; u[i] += (unsigned)it; asm volatile("" : "+v"(a[i])); }  // v_cvt_f64_u32 (+ v_add_u32)
      if (OP == 3) { u[i] = (unsigned)a[i]; asm volatile("" : "+v"(u[i])); }                       // v_cvt_u32_f64
      if (OP == 4) { f[i] = f[i] * 1.0000001f; }                   // v_mul_f32
      if (OP == 5) { a[i] = __builtin_fma(a[i], 1.0000001, 1e-9); } // v_fma_f64
      if (OP == 6) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); }                        // v_cvt_f64_f32
      if (OP == 7) { a[i] = __builtin_trunc(a[i]); asm volatile("" : "+v"(a[i])); }               // v_trunc_f64
      if (OP == 8) { a[i] = (double)(int)u[i]; asm volatile("" : "+v"(a[i])); }                   // v_cvt_f64_i32
    }
  }
  double s = 0; for (int i = 0; i < kChains; ++i) s += a[i] + u[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> static float run(double* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 1u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 2u);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 8 waves x kIters x kChains instructions
  const double instr = 8.0 * kIters * kChains;
  return (float)(ms * 1e-3 * 2.4e9 / instr);  // cycles per wave instruction at 2.4 GHz
}
int main() {
  double* out; CK(hipMalloc((void**)&out, 256 * 8 * 256 * sizeof(double)));
  printf("cycles per wave instruction (2.4 GHz assumed):\n");
  printf("v_mul_f64 %.1f | v_add_f64 %.1f | v_fma_f64 %.1f | v_mul_f32 %.1f\n", run<0>(out), run<1>(out), run<5>(out), run<4>(out));
  printf("v_cvt_f64_u32(+add_u32) %.1f | v_cvt_u32_f64 %.1f | v_cvt_f64_f32 %.1f | v_trunc_f64 %.1f | v_cvt_f64_i32 %.1f\n", run<2>(out), run<3>(out), run<6>(out), run<7>(out), run<8>(out));
  return 0;
}

// Developer micro-benchmark (not part of the product): issue cost of the vector instructions the background preparation
// and compose are made of, per wave instruction on one SIMD, with 1 / 2 / 4 / 5 waves resident per SIMD (independent
// instructions, 8 destinations in rotation).  Cycles are the shader clock (s_memtime), so no clock is assumed.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kIters = 2048;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, float* out, float seed) {
  float a[8]; f32x2 p[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; p[i] = f32x2{a[i], a[i] + 1.f}; u[i] = (unsigned)a[i] * 2654435761u; }
  const float c = seed * 0.5f; const f32x2 cp = {c, c + 0.25f}; const unsigned cu = (unsigned)seed + 77u; unsigned long long msk = (unsigned long long)cyc[0] | 0x5555ull; asm volatile("" : "+s"(msk)); unsigned sr = 0; if (OP == 34) asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#define OP0(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define OP1(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(cp));
#define OP2(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(cp));
#define OP3(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define OP4(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(cu));
#define OP5(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP6(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP7(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define OP8(i) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(u[i]) : "v"(cu));
#define OP9(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define OP10(i) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
#define OP11(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "v"(cp));
#define OP12(i) asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(u[i]) : "v"(cu));
#define OP13(i) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP14(i) asm volatile("v_perm_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(cu));
#define OP15(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define OP16(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP17(i) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP18(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
#define OP19(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define OP20(i) asm volatile("v_add3_u32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(cu));
#define OP22(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(u[i]) : "v"(cu), "s"(msk));
#define OP23(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(cu) : "vcc");
#define OP24(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(cu) : "vcc");
#define OP25(i) asm volatile("v_max_u32 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP26(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(cu));
#define OP27(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
#define OP28(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define OP29(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define OP30(i) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(msk) : "v"(u[i]), "v"(cu));
#define OP31(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
#define OP32(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(cu));
#define OP33(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(cu));
#define OP34(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(cu));
#define OP35(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr) : "v"(u[i]));
#define OP36(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define OP21(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(*(double*)&p[i]));
    if (OP == 0) { REP8(OP0) REP8(OP0) } if (OP == 1) { REP8(OP1) REP8(OP1) } if (OP == 2) { REP8(OP2) REP8(OP2) }
    if (OP == 3) { REP8(OP3) REP8(OP3) } if (OP == 4) { REP8(OP4) REP8(OP4) } if (OP == 5) { REP8(OP5) REP8(OP5) }
    if (OP == 6) { REP8(OP6) REP8(OP6) } if (OP == 7) { REP8(OP7) REP8(OP7) } if (OP == 8) { REP8(OP8) REP8(OP8) }
    if (OP == 9) { REP8(OP9) REP8(OP9) } if (OP == 10) { REP8(OP10) REP8(OP10) } if (OP == 11) { REP8(OP11) REP8(OP11) }
    if (OP == 12) { REP8(OP12) REP8(OP12) } if (OP == 13) { REP8(OP13) REP8(OP13) } if (OP == 14) { REP8(OP14) REP8(OP14) }
    if (OP == 15) { REP8(OP15) REP8(OP15) } if (OP == 16) { REP8(OP16) REP8(OP16) } if (OP == 17) { REP8(OP17) REP8(OP17) }
    if (OP == 18) { REP8(OP18) REP8(OP18) } if (OP == 19) { REP8(OP19) REP8(OP19) } if (OP == 20) { REP8(OP20) REP8(OP20) }
    if (OP == 21) { REP8(OP21) REP8(OP21) }
    if (OP == 22) { REP8(OP22) REP8(OP22) } if (OP == 23) { REP8(OP23) REP8(OP23) } if (OP == 24) { REP8(OP24) }
    if (OP == 25) { REP8(OP25) REP8(OP25) } if (OP == 26) { REP8(OP26) REP8(OP26) } if (OP == 27) { REP8(OP27) REP8(OP27) }
    if (OP == 28) { REP8(OP28) REP8(OP28) } if (OP == 29) { REP8(OP29) REP8(OP29) } if (OP == 30) { REP8(OP30) REP8(OP30) }
    if (OP == 31) { REP8(OP31) REP8(OP31) }
    if (OP == 32) { REP8(OP32) REP8(OP32) }
    if (OP == 33) { asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[0]), "v"(cu) : "vcc"); REP8(OP33) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[1]), "v"(cu) : "vcc"); REP8(OP33) }
    if (OP == 34) { REP8(OP34) REP8(OP34) }
    if (OP == 35) { REP8(OP35) REP8(OP35) } if (OP == 36) { REP8(OP36) REP8(OP36) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sr; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> static void run(const char* name, unsigned long long* cyc, float* out) {
  printf("%-22s", name);
  for (int w : {1, 2, 4, 5, 8}) {
    const int grid = 256 * w;  // w blocks of 4 waves per CU: w waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, cyc, out, 3.f);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < grid; ++i) sum += (double)h[i];
    // s_memtime counts at 100 MHz on this part? report raw ticks per instruction per wave AND per SIMD (ticks * 1 / (16 * kIters) / w)
    const double per_wave = sum / grid / (16.0 * kIters);
    printf("  %dw: %6.2f/wave %5.2f/SIMD", w, per_wave, per_wave / w);
  }
  printf("\n");
}
int main() {
  unsigned long long* cyc; float* out;
  CK(hipMalloc((void**)&cyc, 256 * 8 * sizeof(unsigned long long))); CK(hipMalloc((void**)&out, 256 * 8 * 256 * sizeof(float)));
  printf("ticks of the cycle counter per wave instruction, per wave and per SIMD (w waves resident per SIMD):\n");
  run<0>("v_add_f32", cyc, out); run<9>("v_mul_f32", cyc, out); run<10>("v_fma_f32", cyc, out);
  run<1>("v_pk_add_f32", cyc, out); run<2>("v_pk_mul_f32", cyc, out); run<11>("v_pk_fma_f32", cyc, out);
  run<3>("v_cvt_f32_ubyte1", cyc, out); run<7>("v_cvt_i32_f32", cyc, out); run<15>("v_cvt_f32_i32", cyc, out); run<18>("v_cvt_f32_i32_sdwa", cyc, out);
  run<4>("v_cndmask_b32", cyc, out); run<16>("v_add_u32", cyc, out); run<20>("v_add3_u32", cyc, out); run<8>("v_lshl_or_b32", cyc, out);
  run<19>("v_bfe_u32", cyc, out); run<14>("v_perm_b32", cyc, out); run<17>("v_pk_add_u16", cyc, out);
  run<5>("v_mul_u32_u24", cyc, out); run<12>("v_mad_u32_u24", cyc, out); run<13>("v_mul_hi_u32_u24", cyc, out); run<6>("v_mul_lo_u32", cyc, out);
  run<21>("v_mul_f64", cyc, out);
  run<22>("v_cndmask_b32_e64 sgpr", cyc, out); run<23>("v_cmp_gt_u32 vcc", cyc, out); run<30>("v_cmp_gt_u32_e64 sgpr", cyc, out); run<24>("v_cmp+v_cndmask (x8: per pair)", cyc, out);
  run<25>("v_max_u32", cyc, out); run<26>("v_and_b32", cyc, out); run<27>("v_lshlrev_b32", cyc, out); run<28>("v_sub_f32", cyc, out);
  run<32>("v_cndmask_e64 vcc", cyc, out); run<33>("1 v_cmp : 8 v_cndmask vcc", cyc, out); run<34>("v_cndmask vcc (s_mov vcc once)", cyc, out);
  run<35>("v_readlane_b32", cyc, out); run<36>("v_mov_b32 dpp row_shr", cyc, out);
  run<29>("v_cvt_u32_f32", cyc, out); run<31>("v_cvt_pk_u8_f32", cyc, out);
  return 0;
}

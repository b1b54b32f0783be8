// Developer micro-benchmark (not part of the product): does the SHAPE of the strip a wave renders matter to the
// memory system?  One wave = 256 pixels of one sample: 64 x 4 (the compose kernel's shape: four 256-byte row
// segments per plane and store), 128 x 2, or 256 x 1 (one contiguous kilobyte per plane).  Each lane reads two
// 16-byte background-like texel groups from a random image of a 3 GB pool and writes 8 fp32 planes with
// non-temporal 16-byte stores, as compose does.   hipcc --offload-arch=gfx950 -O3 strip_shape.hip -o strip_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 512, H = 384, B = 32, PW = 1024, PH = 768, NPOOL = 1000;

template <int SW, int SH, bool READS>
__global__ __launch_bounds__(64) void strip_kernel(const uint32_t* __restrict__ pool, float* __restrict__ out, int salt) {
  constexpr int per_row = W / SW, per_sample = per_row * (H / SH);
  // the compose kernel's XCD interleave: 32 consecutive strips per XCD turn
  int wg = blockIdx.x;
  { const int xcd = wg & 7, slot = wg >> 3; wg = (((slot >> 5) * 8 + xcd) << 5) + (slot & 31); }
  const int s = wg / per_sample, t = wg - s * per_sample;
  const int lane = threadIdx.x;
  constexpr int lanes_x = SW / 4;
  const int x0 = (t % per_row) * SW + (lane % lanes_x) * 4, y = (t / per_row) * SH + lane / lanes_x;
  uint4 a = make_uint4(1, 2, 3, 4), b = a;
  if (READS) {
    const uint32_t img = ((uint32_t)(s * 2654435761u + salt * 40503u) >> 7) % NPOOL;
    const uint32_t* tex = pool + (size_t)img * PW * PH;
    a = *reinterpret_cast<const uint4*>(tex + (size_t)(y + PH / 4) * PW + x0 + PW / 4);
    b = *reinterpret_cast<const uint4*>(tex + (size_t)(y + PH / 4 + 9) * PW + x0 + PW / 4 + 12);
  }
  const size_t plane = (size_t)W * H, o = (size_t)y * W + x0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const uint32_t sh = 8 * (p % 3);
    const uint4 q = (p & 1) ? b : a;
    f32x4 v = {(float)((q.x >> sh) & 255u), (float)((q.y >> sh) & 255u), (float)((q.z >> sh) & 255u), (float)((q.w >> sh) & 255u)};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + ((size_t)s * 8 + p) * plane + o));
  }
}

// the ceiling: the same bytes as one contiguous fill, 16 bytes per lane, grid-stride (NT: non-temporal stores)
template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ out, size_t n4, float v) {
  f32x4 val = {v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(out) + i);
    else reinterpret_cast<f32x4*>(out)[i] = val;
  }
}
template <bool NT>
static float run_fill(float* out, size_t bytes, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill_kernel<NT>), dim3(256 * 16), dim3(256), 0, 0, out, bytes / 16, (float)i);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill_kernel<NT>), dim3(256 * 16), dim3(256), 0, 0, out, bytes / 16, (float)i);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

template <int SW, int SH, bool READS>
static float run(const uint32_t* pool, float* out, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = B * (W / SW) * (H / SH);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((strip_kernel<SW, SH, READS>), dim3(grid), dim3(64), 0, 0, pool, out, i);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((strip_kernel<SW, SH, READS>), dim3(grid), dim3(64), 0, 0, pool, out, 5 + i);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  uint32_t* pool = nullptr;
  float* out = nullptr;
  const size_t pool_bytes = (size_t)NPOOL * PW * PH * 4, out_bytes = (size_t)B * 8 * W * H * 4;
  CK(hipMalloc((void**)&pool, pool_bytes));
  CK(hipMalloc((void**)&out, out_bytes));
  CK(hipMemset(pool, 0x5A, pool_bytes));
  CK(hipDeviceSynchronize());
  const double mb_w = out_bytes / 1e6, mb_r = (double)B * W * H * 32 / 4 / 1e6;
  printf("per launch: %.1f MB written, %.1f MB of texel reads requested\n", mb_w, mb_r);
  printf("contiguous fill of the same bytes: non-temporal stores %6.1f us, plain stores %6.1f us\n", run_fill<true>(out, out_bytes, 100), run_fill<false>(out, out_bytes, 100));
  for (int round = 0; round < 2; ++round) {
    printf("stores only : 64x4 %6.1f us   128x2 %6.1f us   256x1 %6.1f us\n", run<64, 4, false>(pool, out, 100), run<128, 2, false>(pool, out, 100), run<256, 1, false>(pool, out, 100));
    printf("with reads  : 64x4 %6.1f us   128x2 %6.1f us   256x1 %6.1f us\n", run<64, 4, true>(pool, out, 100), run<128, 2, true>(pool, out, 100), run<256, 1, true>(pool, out, 100));
  }
  return 0;
}

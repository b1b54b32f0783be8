// Developer micro-benchmark (not part of the product): what do the FOREGROUND texture reads of compose cost the memory
// system by themselves?  64 x 4 strips as in compose: every wave reads background-like texels (two 16-byte groups per
// lane from its sample's image) and writes 8 fp32 planes with non-temporal stores; a fraction of the waves ("visits",
// 38 % with one, a third of those with two) additionally reads an object-like window - 16 B per lane at the strip's own
// position plus eight 8-byte tap loads around it - from ANOTHER random image of the pool:
//   independent   issued together with the background reads
//   dependent     issued after the background reads have returned (compose: coverage -> ballot -> taps)
//   warm          the object images come from the first 64 pool images (Infinity-Cache resident)
// Round 3: LAYOUT = how a pool image is laid out (0 row-major, 1 tiles of 16 x 4 texels, 2 tiles of 8 x 8 texels: 256 B each,
// tiles in row-major order) and ROT = the object windows are read along a line rotated by 30 degrees (real objects carry
// any rotation; the background never more than 10 degrees).
// hipcc --offload-arch=gfx950 -O3 fg_reads.hip -o fg_reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 512, H = 384, B = 32, PW = 1024, PH = 768, NPOOL = 1000;

__device__ __forceinline__ uint32_t hash32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// MODE 0: no visits; 1: independent; 2: dependent; 3: independent, the window as two wide row loads (6 rows x 272 B:
// what staging the window in LDS would issue); WARM: object images from the first 64
// byte offset of texel (x, y) inside an image
template <int LAYOUT>
__device__ __forceinline__ size_t texel_at(int x, int y) {
  if (LAYOUT == 0) return ((size_t)y * PW + x) * 4;
  if (LAYOUT == 1) return ((size_t)((y >> 2) * (PW / 16) + (x >> 4)) * 64 + ((y & 3) * 16 + (x & 15))) * 4;
  return ((size_t)((y >> 3) * (PW / 8) + (x >> 3)) * 64 + ((y & 7) * 8 + (x & 7))) * 4;
}
// compose's memory operations on a LAYOUT pool: background texels of both frames (4 px per lane), stores, and object
// windows (frame 0: the lane's 4 texels; frame 1: two texel pairs on two rows per pixel, along a rotated line if ROT)
template <int LAYOUT, bool ROT, bool WARM, bool VISITS>
__global__ __launch_bounds__(64) void layout_kernel(const uint32_t* __restrict__ pool_, float* __restrict__ out, int salt) {
  const char* pool = reinterpret_cast<const char*>(pool_);
  constexpr int per_row = W / 64, per_sample = per_row * (H / 4);
  int wg = blockIdx.x;
  { const int xcd = wg & 7, slot = wg >> 3; wg = (((slot >> 5) * 8 + xcd) << 5) + (slot & 31); }
  const int s = wg / per_sample, t = wg - s * per_sample;
  const int lane = threadIdx.x;
  const int x0 = (t % per_row) * 64 + (lane & 15) * 4, y = (t / per_row) * 4 + (lane >> 4);
  const uint32_t img = (hash32((uint32_t)s * 2654435761u + (uint32_t)salt * 40503u) >> 7) % NPOOL;
  const char* tex = pool + (size_t)img * PW * PH * 4;
  const uint32_t cell = hash32((uint32_t)(s * 131 + (t % per_row) * 17 + (t / per_row) / 12) * 2246822519u + (uint32_t)salt);
  const int visits = !VISITS ? 0 : ((cell % 100u) < 38u ? ((cell >> 8) % 3u == 0 ? 2 : 1) : 0);  // wave-uniform
  // background: frame 0 = 4 consecutive texels (16 B: inside one tile for both tilings), frame 1 = taps shifted by (12.x, 9.x)
  const uint4 a = *reinterpret_cast<const uint4*>(tex + texel_at<LAYOUT>(x0 + PW / 4, y + PH / 4));
  uint2 bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bt[k] = *reinterpret_cast<const uint2*>(tex + texel_at<LAYOUT>(x0 + PW / 4 + 12 + (k & 3), y + PH / 4 + 9 + (k >> 2)));
  uint32_t acc = 0;
  for (int v = 0; v < visits; ++v) {
    const uint32_t oimg = (hash32(cell + 77u * (uint32_t)v) >> 5) % (WARM ? 64u : (uint32_t)NPOOL);
    const char* ot = pool + (size_t)oimg * PW * PH * 4;
    const int ox = PW / 2 - W / 2, oy = PH / 2 - H / 2;
    const uint4 q = *reinterpret_cast<const uint4*>(ot + texel_at<LAYOUT>(ox + x0, oy + y));
    uint2 tp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int tx = x0 + 2 + (k & 3), ty = y + 1 + (k >> 2);
      if (ROT) {  // rotate about the frame centre by 30 degrees (integer arithmetic: 887 / 1024, 512 / 1024)
        const int dx = tx - W / 2, dy = ty - H / 2;
        tx = W / 2 + ((887 * dx - 512 * dy) >> 10); ty = H / 2 + ((512 * dx + 887 * dy) >> 10);
        tx = min(max(tx, -ox), PW - ox - 2); ty = min(max(ty, -oy), PH - oy - 1);
      }
      tp[k] = *reinterpret_cast<const uint2*>(ot + texel_at<LAYOUT>(ox + tx, oy + ty));
    }
    acc ^= q.x + q.y + q.z + q.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += tp[k].x * 3u + tp[k].y;
    asm volatile("" : "+v"(acc));
  }
  uint4 b = make_uint4(bt[0].x ^ bt[4].y, bt[1].x ^ bt[5].y, bt[2].x ^ bt[6].y, bt[3].x ^ bt[7].y);
  b.x ^= acc;
  const size_t plane = (size_t)W * H, o = (size_t)y * W + x0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const uint32_t sh = 8 * (p % 3);
    const uint4 q = (p & 1) ? b : a;
    f32x4 vv = {(float)((q.x >> sh) & 255u), (float)((q.y >> sh) & 255u), (float)((q.z >> sh) & 255u), (float)((q.w >> sh) & 255u)};
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(out + ((size_t)s * 8 + p) * plane + o));
  }
}
template <int LAYOUT, bool ROT, bool WARM, bool VISITS>
static float run_layout(const uint32_t* pool, float* out, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = B * (W / 64) * (H / 4);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((layout_kernel<LAYOUT, ROT, WARM, VISITS>), dim3(grid), dim3(64), 0, 0, pool, out, i);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((layout_kernel<LAYOUT, ROT, WARM, VISITS>), dim3(grid), dim3(64), 0, 0, pool, out, 5 + i);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

template <int MODE, bool WARM, bool STORES, int BPT>
__global__ __launch_bounds__(64) void fg_kernel(const uint32_t* __restrict__ pool_, float* __restrict__ out, int salt) {
  const char* pool = reinterpret_cast<const char*>(pool_);
  constexpr int per_row = W / 64, per_sample = per_row * (H / 4);
  int wg = blockIdx.x;
  { const int xcd = wg & 7, slot = wg >> 3; wg = (((slot >> 5) * 8 + xcd) << 5) + (slot & 31); }
  const int s = wg / per_sample, t = wg - s * per_sample;
  const int lane = threadIdx.x;
  const int x0 = (t % per_row) * 64 + (lane & 15) * 4, y = (t / per_row) * 4 + (lane >> 4);
  const uint32_t img = (hash32((uint32_t)s * 2654435761u + (uint32_t)salt * 40503u) >> 7) % NPOOL;
  const char* tex = pool + (size_t)img * PW * PH * BPT;
  // objects are blobs of ~48 x 48 px: the strips of one 64 x 48 cell share their fate and their object image
  const uint32_t cell = hash32((uint32_t)(s * 131 + (t % per_row) * 17 + (t / per_row) / 12) * 2246822519u + (uint32_t)salt);
  const int visits = MODE == 0 ? 0 : ((cell % 100u) < 38u ? ((cell >> 8) % 3u == 0 ? 2 : 1) : 0);  // wave-uniform
  uint4 a = make_uint4(0, 0, 0, 0), b = a;
  if (BPT == 4) {
    a = *reinterpret_cast<const uint4*>(tex + ((size_t)(y + PH / 4) * PW + x0 + PW / 4) * 4);
    b = *reinterpret_cast<const uint4*>(tex + ((size_t)(y + PH / 4 + 9) * PW + x0 + PW / 4 + 12) * 4);
  } else {  // 4 texels = 12 bytes, 4-byte aligned (x0 is a multiple of 4)
    const uint3 a3 = *reinterpret_cast<const uint3*>(tex + ((size_t)(y + PH / 4) * PW + x0 + PW / 4) * 3);
    const uint3 b3 = *reinterpret_cast<const uint3*>(tex + ((size_t)(y + PH / 4 + 9) * PW + x0 + PW / 4 + 12) * 3);
    a = make_uint4(a3.x, a3.y, a3.z, a3.x ^ a3.y); b = make_uint4(b3.x, b3.y, b3.z, b3.y ^ b3.z);
  }
  uint32_t acc = 0;
  if (MODE == 2) {  // the visit's loads wait for the background's
    acc = a.x ^ b.y;
    asm volatile("" : "+v"(acc));
  }
  for (int v = 0; v < visits; ++v) {
    uint32_t oimg = (hash32(cell + 77u * (uint32_t)v) >> 5) % (WARM ? 64u : (uint32_t)NPOOL);
    const char* ot = pool + ((size_t)oimg * PW * PH + (size_t)(PH / 2 - H / 2) * PW + (PW / 2 - W / 2)) * BPT;
    uint4 q;
    if (MODE == 3) {
      const int y4 = y & ~3, xb = x0 & ~63;       // the strip's origin
      uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0;
      const int r = lane / 17, cc = lane - r * 17;  // 3 rows of 17 x 16 B per load
      if (r < 3) {
        w0 = *reinterpret_cast<const uint4*>(ot + ((size_t)(y4 + r) * PW + xb) * 4 + cc * 16);
        w1 = *reinterpret_cast<const uint4*>(ot + ((size_t)(y4 + 3 + r) * PW + xb) * 4 + cc * 16);
      }
      acc ^= w0.x + w0.y + w0.z + w0.w + w1.x + w1.y + w1.z + w1.w;
      continue;
    }
    if (BPT == 4) q = *reinterpret_cast<const uint4*>(ot + ((size_t)y * PW + x0) * 4);
    else { const uint3 q3 = *reinterpret_cast<const uint3*>(ot + ((size_t)y * PW + x0) * 3); q = make_uint4(q3.x, q3.y, q3.z, q3.x + q3.z); }
    uint2 tp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tp[k] = *reinterpret_cast<const uint2*>(ot + ((size_t)(y + 1 + (k >> 2) + ((acc >> 20) & 1)) * PW + x0 + 2 + (k & 3)) * BPT);  // (BPT 3: byte-aligned 8-byte loads)
    acc ^= q.x + q.y + q.z + q.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += tp[k].x * 3u + tp[k].y;
    if (MODE == 2) asm volatile("" : "+v"(acc));  // the next visit depends on this one
  }
  a.x ^= acc;
  if (!STORES) {
    if ((a.x ^ b.x ^ a.w) == 0x12345u) out[lane] = 1.f;  // (keeps the loads alive)
    return;
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

template <int MODE, bool WARM, bool STORES, int BPT = 4>
static float run(const uint32_t* pool, float* out, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = B * (W / 64) * (H / 4);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fg_kernel<MODE, WARM, STORES, BPT>), dim3(grid), dim3(64), 0, 0, pool, out, i);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fg_kernel<MODE, WARM, STORES, BPT>), dim3(grid), dim3(64), 0, 0, pool, out, 5 + i);
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
  for (int round = 0; round < 2; ++round) {
    printf("pool layout (stores + background texels and taps; visits cold / cold rotated 30 deg / warm rotated):\n");
    printf("  row-major      : no visits %6.1f us | %6.1f / %6.1f / %6.1f\n", run_layout<0, false, false, false>(pool, out, 100),
           run_layout<0, false, false, true>(pool, out, 100), run_layout<0, true, false, true>(pool, out, 100), run_layout<0, true, true, true>(pool, out, 100));
    printf("  tiles 16 x 4   : no visits %6.1f us | %6.1f / %6.1f / %6.1f\n", run_layout<1, false, false, false>(pool, out, 100),
           run_layout<1, false, false, true>(pool, out, 100), run_layout<1, true, false, true>(pool, out, 100), run_layout<1, true, true, true>(pool, out, 100));
    printf("  tiles 8 x 8    : no visits %6.1f us | %6.1f / %6.1f / %6.1f\n", run_layout<2, false, false, false>(pool, out, 100),
           run_layout<2, false, false, true>(pool, out, 100), run_layout<2, true, false, true>(pool, out, 100), run_layout<2, true, true, true>(pool, out, 100));
    printf("with stores : no visits %6.1f us | independent cold %6.1f  warm %6.1f | dependent cold %6.1f  warm %6.1f\n",
           run<0, false, true>(pool, out, 100), run<1, false, true>(pool, out, 100), run<1, true, true>(pool, out, 100),
           run<2, false, true>(pool, out, 100), run<2, true, true>(pool, out, 100));
    printf("3-byte texel: no visits %6.1f us | independent cold %6.1f  warm %6.1f | dependent cold %6.1f  warm %6.1f\n",
           run<0, false, true, 3>(pool, out, 100), run<1, false, true, 3>(pool, out, 100), run<1, true, true, 3>(pool, out, 100),
           run<2, false, true, 3>(pool, out, 100), run<2, true, true, 3>(pool, out, 100));
    printf("wide loads  : independent cold %6.1f  warm %6.1f   (two 16-byte row loads per visit instead of 16 B + eight taps)\n",
           run<3, false, true>(pool, out, 100), run<3, true, true>(pool, out, 100));
    printf("reads only  : no visits %6.1f us | independent cold %6.1f  warm %6.1f | dependent cold %6.1f  warm %6.1f\n",
           run<0, false, false>(pool, out, 100), run<1, false, false>(pool, out, 100), run<1, true, false>(pool, out, 100),
           run<2, false, false>(pool, out, 100), run<2, true, false>(pool, out, 100));
  }
  return 0;
}

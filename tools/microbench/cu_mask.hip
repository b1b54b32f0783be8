// Developer micro-benchmark (not part of the product): how does a compose-like kernel (fg_reads.hip's memory operations:
// stores + background texels + taps + cold object windows) scale with the number of CUs its stream may use
// (hipExtStreamCreateWithCUMask), and how fast does a streaming-read kernel run on the remaining CUs?
// hipcc --offload-arch=gfx950 -O3 cu_mask.hip -o cu_mask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W = 512, H = 384, B = 32, PW = 1024, PH = 768, NPOOL = 1000;
__device__ __forceinline__ uint32_t hash32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

__global__ __launch_bounds__(64) void compose_like(const uint32_t* __restrict__ pool_, float* __restrict__ out, int salt) {
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
  const int visits = (cell % 100u) < 38u ? ((cell >> 8) % 3u == 0 ? 2 : 1) : 0;
  auto at = [](int x, int yy) { return ((size_t)yy * PW + x) * 4; };
  const uint4 a = *reinterpret_cast<const uint4*>(tex + at(x0 + PW / 4, y + PH / 4));
  uint2 bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bt[k] = *reinterpret_cast<const uint2*>(tex + at(x0 + PW / 4 + 12 + (k & 3), y + PH / 4 + 9 + (k >> 2)));
  uint32_t acc = 0;
  for (int v = 0; v < visits; ++v) {
    const uint32_t oimg = (hash32(cell + 77u * (uint32_t)v) >> 5) % (uint32_t)NPOOL;
    const char* ot = pool + (size_t)oimg * PW * PH * 4;
    const int ox = PW / 2 - W / 2, oy = PH / 2 - H / 2;
    const uint4 q = *reinterpret_cast<const uint4*>(ot + at(ox + x0, oy + y));
    uint2 tp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int tx = x0 + 2 + (k & 3), ty = y + 1 + (k >> 2);
      const int dx = tx - W / 2, dy = ty - H / 2;
      tx = W / 2 + ((887 * dx - 512 * dy) >> 10); ty = H / 2 + ((512 * dx + 887 * dy) >> 10);
      tx = min(max(tx, -ox), PW - ox - 2); ty = min(max(ty, -oy), PH - oy - 1);
      tp[k] = *reinterpret_cast<const uint2*>(ot + at(ox + tx, oy + ty));
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
// streaming read of `bytes` from a random offset of the pool (one dword per 128-byte line per lane, like the warm-up launch)
__global__ __launch_bounds__(64) void stream_read(const uint32_t* __restrict__ pool, size_t first_line, int lines, uint32_t* sink) {
  uint32_t acc = 0;
  for (int i = blockIdx.x * 64 + threadIdx.x; i < lines; i += gridDim.x * 64) acc ^= pool[(first_line + (size_t)i) * 32];
  if (acc == 0x1234567u) *sink = acc;
}

static hipStream_t masked_stream(int first_cu, int n_cus) {
  // CU mask bits: bit i = CU i in the runtime's linear order (XCDs interleaved); take every CU with first <= index < first + n
  std::vector<uint32_t> mask(8, 0u);  // 256 CUs
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return nullptr; }
  return s;
}
static float time_kernel(hipStream_t s, int reps, const uint32_t* pool, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = B * (W / 64) * (H / 4);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(compose_like, dim3(grid), dim3(64), 0, s, pool, out, i);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(compose_like, dim3(grid), dim3(64), 0, s, pool, out, 5 + i);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  uint32_t* pool = nullptr; float* out = nullptr; uint32_t* sink = nullptr;
  const size_t pool_bytes = (size_t)NPOOL * PW * PH * 4, out_bytes = (size_t)B * 8 * W * H * 4;
  CK(hipMalloc((void**)&pool, pool_bytes)); CK(hipMalloc((void**)&out, out_bytes)); CK(hipMalloc((void**)&sink, 4));
  CK(hipMemset(pool, 0x5A, pool_bytes)); CK(hipDeviceSynchronize());
  for (int n : {256, 224, 192, 160, 128}) {
    hipStream_t s = masked_stream(0, n);
    if (!s) return 1;
    printf("compose-like kernel on %3d CUs: %6.1f us\n", n, time_kernel(s, 60, pool, out));
    (void)hipStreamDestroy(s);
  }
  // the two together: compose-like on CUs [0, 192), 50 MB streaming reads per launch on CUs [192, 256)
  for (int n_read : {64, 32}) {
    hipStream_t sc = masked_stream(0, 256 - n_read), sr = masked_stream(256 - n_read, n_read);
    hipEvent_t e0, e1, r0, r1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&r0); (void)hipEventCreate(&r1);
    const int grid = B * (W / 64) * (H / 4), lines = 50 * 1000 * 1000 / 128, reps = 60;
    (void)hipEventRecord(r0, sr);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_read, dim3(n_read * 8), dim3(64), 0, sr, pool, (size_t)(i * 7919 % 900) * 6144 * 4, lines, sink);
    (void)hipEventRecord(r1, sr);
    (void)hipEventRecord(e0, sc);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(compose_like, dim3(grid), dim3(64), 0, sc, pool, out, 100 + i);
    (void)hipEventRecord(e1, sc);
    (void)hipDeviceSynchronize();
    float mc = 0, mr = 0;
    (void)hipEventElapsedTime(&mc, e0, e1); (void)hipEventElapsedTime(&mr, r0, r1);
    printf("together: compose-like on %3d CUs %6.1f us per launch | 50 MB streaming read on %2d CUs %6.1f us per launch\n", 256 - n_read, mc * 1e3f / reps, n_read, mr * 1e3f / reps);
  }
  {  // the same with no masks at all
    hipStream_t sc, sr;
    (void)hipStreamCreateWithFlags(&sc, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sr, hipStreamNonBlocking);
    hipEvent_t e0, e1, r0, r1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&r0); (void)hipEventCreate(&r1);
    const int grid = B * (W / 64) * (H / 4), lines = 50 * 1000 * 1000 / 128, reps = 60;
    (void)hipEventRecord(r0, sr);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_read, dim3(512), dim3(64), 0, sr, pool, (size_t)(i * 7919 % 900) * 6144 * 4, lines, sink);
    (void)hipEventRecord(r1, sr);
    (void)hipEventRecord(e0, sc);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(compose_like, dim3(grid), dim3(64), 0, sc, pool, out, 100 + i);
    (void)hipEventRecord(e1, sc);
    (void)hipDeviceSynchronize();
    float mc = 0, mr = 0;
    (void)hipEventElapsedTime(&mc, e0, e1); (void)hipEventElapsedTime(&mr, r0, r1);
    printf("together, no masks: compose-like %6.1f us per launch | 50 MB streaming read %6.1f us per launch\n", mc * 1e3f / reps, mr * 1e3f / reps);
  }
  return 0;
}

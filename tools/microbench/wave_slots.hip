// Developer microbenchmark (round 5): how many single-wave workgroups does a SIMD / a CU hold at once, by kernel shape?
// Every wave notes HW_ID / XCC_ID and its start and end time, and spins ~30 us; the host counts the waves alive together per
// SIMD and per CU.  Shapes: registers 24 / 64 / 88 per lane, LDS 0 / 1 KB / 10 KB per workgroup, 64 / 128 / 256 threads.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/wave_slots.hip -o tools/microbench/wave_slots
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
template <int kLds, int kRegs>
__global__ void spin(Rec* out, int spin_ticks) {
  __shared__ int lds[kLds > 0 ? kLds / 4 : 1];
  if (kLds > 0) lds[threadIdx.x % (kLds / 4)] = threadIdx.x;
  if (kRegs >= 64) asm volatile("; pad" ::: "v63");
  if (kRegs >= 88) asm volatile("; pad" ::: "v87");
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.t0 = t0; r.t1 = wall_clock64();
    r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
    out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = r;
  }
  if (kLds > 0 && lds[0] == -12345) out[0].hw = 0;
}
template <int kLds, int kRegs>
static void run(const char* name, int threads) {
  const int grid = 256 * 4 * 12 * 64 / threads, waves = grid * (threads / 64);
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * waves);
  hipLaunchKernelGGL((spin<kLds, kRegs>), dim3(grid), dim3(threads), 0, 0, d, 3000);
  hipDeviceSynchronize();
  std::vector<Rec> r(waves);
  hipMemcpy(r.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
  hipFree(d);
  // waves alive together per SIMD / CU: sweep over start / end events
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> per_simd, per_cu;
  unsigned max_wave_id = 0;
  for (const Rec& x : r) {
    const unsigned cu = (x.xcc << 16) | (x.hw & 0xff00u), simd = cu | ((x.hw >> 4) & 3u);
    per_simd[simd].push_back({x.t0, 1}); per_simd[simd].push_back({x.t1, -1});
    per_cu[cu].push_back({x.t0, 1}); per_cu[cu].push_back({x.t1, -1});
    max_wave_id = std::max(max_wave_id, x.hw & 15u);
  }
  auto peak = [](std::map<unsigned, std::vector<std::pair<unsigned long long, int>>>& m) {
    int best = 0;
    for (auto& kv : m) {
      std::sort(kv.second.begin(), kv.second.end());
      int cur = 0;
      for (auto& e : kv.second) { cur += e.second; best = std::max(best, cur); }
    }
    return best;
  };
  const int ps = peak(per_simd), pc = peak(per_cu);
  std::printf("%-44s %4zu CUs %5zu SIMDs: at most %2d waves together on a SIMD, %2d on a CU; largest wave id %u\n", name, per_cu.size(), per_simd.size(), ps, pc, max_wave_id);
}
// two kernels of different register sizes on two streams: do their waves share a SIMD's register file?
template <int kRegsA, int kRegsB>
static void mixed(const char* name) {
  const int wa = 256 * 4 * 6, wb = 256 * 4 * 4;
  Rec *da, *db;
  hipMalloc(&da, sizeof(Rec) * wa); hipMalloc(&db, sizeof(Rec) * wb);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipLaunchKernelGGL((spin<0, kRegsA>), dim3(wa), dim3(64), 0, s1, da, 6000);
  hipLaunchKernelGGL((spin<0, kRegsB>), dim3(wb), dim3(64), 0, s2, db, 3000);
  hipDeviceSynchronize();
  std::vector<Rec> ra(wa), rb(wb);
  hipMemcpy(ra.data(), da, sizeof(Rec) * wa, hipMemcpyDeviceToHost);
  hipMemcpy(rb.data(), db, sizeof(Rec) * wb, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> per_simd;
  for (const Rec& x : ra) { const unsigned k = (x.xcc << 16) | (x.hw & 0xff00u) | ((x.hw >> 4) & 3u); per_simd[k].push_back({x.t0, 1}); per_simd[k].push_back({x.t1, -1}); }
  for (const Rec& x : rb) { const unsigned k = (x.xcc << 16) | (x.hw & 0xff00u) | ((x.hw >> 4) & 3u); per_simd[k].push_back({x.t0, 100}); per_simd[k].push_back({x.t1, -100}); }
  int best = 0, best_a = 0, best_b = 0;
  for (auto& kv : per_simd) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0;
    for (auto& e : kv.second) { cur += e.second; if (cur % 100 + cur / 100 > best) { best = cur % 100 + cur / 100; best_a = cur % 100; best_b = cur / 100; } }
  }
  unsigned long long a0 = ~0ull, b0 = ~0ull, b1 = 0;
  for (const Rec& x : ra) a0 = std::min(a0, x.t0);
  for (const Rec& x : rb) { b0 = std::min(b0, x.t0); b1 = std::max(b1, x.t0); }
  std::printf("%-44s at most %d waves together on a SIMD (%d of A + %d of B); B's first wave starts %.1f us after A's first, its last %.1f us\n", name, best, best_a, best_b,
              (double)(long long)(b0 - a0) / 100.0, (double)(long long)(b1 - a0) / 100.0);
  hipFree(da); hipFree(db);
}
int main() {
  mixed<88, 64>("A: 88 registers x 6144, B: 64 registers x 4096");
  mixed<88, 24>("A: 88 registers x 6144, B: 24 registers x 4096");
  mixed<64, 88>("A: 64 registers x 6144, B: 88 registers x 4096");
  run<0, 24>("64 threads, ~24 registers, no LDS", 64);
  run<1024, 24>("64 threads, ~24 registers, 1 KB LDS", 64);
  run<0, 64>("64 threads, 64 registers, no LDS", 64);
  run<10240, 64>("64 threads, 64 registers, 10 KB LDS", 64);
  run<0, 88>("64 threads, 88 registers, no LDS", 64);
  run<0, 24>("256 threads, ~24 registers, no LDS", 256);
  run<1024, 24>("256 threads, ~24 registers, 1 KB LDS", 256);
  return 0;
}

// Developer microbenchmark: what a step's SUBMISSION costs on this HIP stack - four kernel launches on an in-order
// stream against one hipGraphLaunch of the same four kernels, with empty kernels (host-bound: config 1)
// and with kernels of ~20 us (device-bound: the gaps between a chain's kernels).
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/launch_cost tools/microbench/launch_cost.hip && ./tools/microbench/launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// spins for `ticks` of the 100 MHz wall clock (bounded: at most 1 << 22 polls), then leaves a mark
__global__ void spin_kernel(int ticks, int* out, long long a, long long b, long long c, long long d) {
  const long long t0 = wall_clock64();
  int guard = 0;
  while (wall_clock64() - t0 < ticks && ++guard < (1 << 22)) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = guard + (int)(a + b + c + d);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Result { double issue_us, total_us; };

template <class F>
static Result run(int iters, const std::vector<hipStream_t>& streams, F submit) {
  CK(hipDeviceSynchronize());
  const double t0 = now_us();
  for (int i = 0; i < iters; ++i) submit(i, streams[i % streams.size()]);
  const double t1 = now_us();
  CK(hipDeviceSynchronize());
  const double t2 = now_us();
  return {(t1 - t0) / iters, (t2 - t0) / iters};
}

int main() {
  int* d_out = nullptr;
  CK(hipMalloc((void**)&d_out, 64));
  void* h_pinned = nullptr;
  CK(hipHostMalloc(&h_pinned, 1 << 16, hipHostMallocDefault));
  void* d_buf = nullptr;
  CK(hipMalloc(&d_buf, 1 << 16));
  std::vector<hipStream_t> one(1), four(4);
  CK(hipStreamCreateWithFlags(&one[0], hipStreamNonBlocking));
  for (auto& s : four) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));

  for (int ticks : {0, 2000}) {  // empty kernels; 20 us kernels
    for (int nk : {3, 4, 5}) {
      for (int use_four = 0; use_four < 2; ++use_four) {
        const std::vector<hipStream_t>& st = use_four ? four : one;
        const int iters = ticks ? 200 : 2000;
        // (a) plain launches
        Result a = run(iters, st, [&](int, hipStream_t s) {
          for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(spin_kernel, dim3(k == nk - 1 ? 1024 : 64), dim3(64), 0, s, ticks, d_out, 1LL, 2LL, 3LL, 4LL);
        });
        // (b) hipExtLaunchKernelGGL (what the library uses: events on the kernels' own packets)
        Result b = run(iters, st, [&](int, hipStream_t s) {
          for (int k = 0; k < nk; ++k) hipExtLaunchKernelGGL(spin_kernel, dim3(k == nk - 1 ? 1024 : 64), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d_out, 1LL, 2LL, 3LL, 4LL);
        });
        // (c) one graph per stream, captured once
        std::vector<hipGraphExec_t> execs;
        for (hipStream_t s : st) {
          hipGraph_t g;
          CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
          for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(spin_kernel, dim3(k == nk - 1 ? 1024 : 64), dim3(64), 0, s, ticks, d_out, 1LL, 2LL, 3LL, 4LL);
          CK(hipStreamEndCapture(s, &g));
          hipGraphExec_t ge;
          CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
          CK(hipGraphDestroy(g));
          execs.push_back(ge);
        }
        for (size_t i = 0; i < execs.size(); ++i) CK(hipGraphLaunch(execs[i], st[i]));  // (the first launch uploads the graph)
        Result c = run(iters, st, [&](int i, hipStream_t s) { CK(hipGraphLaunch(execs[i % execs.size()], s)); });
        for (auto ge : execs) CK(hipGraphExecDestroy(ge));
        std::printf("%s kernels x %d on %d stream(s): launches issue %.1f total %.1f | ext launches issue %.1f total %.1f | graph issue %.1f total %.1f  (us per step)\n",
                    ticks ? "20-us" : "empty", nk, (int)st.size(), a.issue_us, a.total_us, b.issue_us, b.total_us, c.issue_us, c.total_us);
      }
    }
  }
  // single calls
  {
    Result m = run(2000, one, [&](int, hipStream_t s) { CK(hipMemcpyAsync(d_buf, h_pinned, 4096, hipMemcpyHostToDevice, s)); });
    Result m3 = run(2000, one, [&](int, hipStream_t s) { for (int k = 0; k < 3; ++k) CK(hipMemcpyAsync((char*)d_buf + 8192 * k, (char*)h_pinned + 8192 * k, 4096, hipMemcpyHostToDevice, s)); });
    Result e = run(2000, one, [&](int, hipStream_t s) { CK(hipEventRecord(ev, s)); });
    Result q = run(2000, one, [&](int, hipStream_t) { (void)hipEventQuery(ev); });
    Result w = run(2000, four, [&](int i, hipStream_t s) { CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(four[(i + 1) % 4], ev, 0)); });
    std::printf("hipMemcpyAsync H2D 4 KB pinned: issue %.1f total %.1f | three of them: issue %.1f total %.1f | hipEventRecord %.1f | hipEventQuery %.2f | record + wait on another stream %.1f (us per call)\n",
                m.issue_us, m.total_us, m3.issue_us, m3.total_us, e.issue_us, q.issue_us, w.issue_us);
    // a graph of [H2D copy, 3 kernels]: config 1's step as one submission
    hipGraph_t g;
    CK(hipStreamBeginCapture(one[0], hipStreamCaptureModeThreadLocal));
    CK(hipMemcpyAsync(d_buf, h_pinned, 4096, hipMemcpyHostToDevice, one[0]));
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, one[0], 0, d_out, 1LL, 2LL, 3LL, 4LL);
    CK(hipStreamEndCapture(one[0], &g));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, one[0]));
    Result c = run(2000, one, [&](int, hipStream_t s) { CK(hipGraphLaunch(ge, s)); });
    Result p = run(2000, one, [&](int, hipStream_t s) {
      CK(hipMemcpyAsync(d_buf, h_pinned, 4096, hipMemcpyHostToDevice, s));
      for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, 0, d_out, 1LL, 2LL, 3LL, 4LL);
    });
    std::printf("[H2D 4 KB + 3 empty kernels] plain: issue %.1f total %.1f | graph: issue %.1f total %.1f (us per step)\n", p.issue_us, p.total_us, c.issue_us, c.total_us);
  }
  return 0;
}

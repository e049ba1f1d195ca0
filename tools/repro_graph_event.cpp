// Stand-alone reproducer for the hazard DESIGN.md section 6 describes: in the piecewise-graph data-parallel step a
// HIP-graph launch was followed DIRECTLY by the cross-stream event record / wait pair of a collective, and 40-60 % of 13-step runs
// ended in NaN parameters unless one ordinary kernel was launched between the graph launch and the event record.  No engine, no
// RCCL, no torch here: one captured graph of slow kernels on stream A, then hipEventRecord(A) -> hipStreamWaitEvent(B) -> a checker
// kernel on B that must see everything the graph wrote -> event back to A.  If the event recorded right behind hipGraphLaunch can
// complete before the graph's last kernel has, the checker counts stale elements.
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro_graph_event.cpp -o /tmp/repro_graph_event && /tmp/repro_graph_event [iters] [kernels]
//
// Prints, per variant, how many of the iterations saw stale data:
//   direct          graph launch, event record
//   kernel-between  graph launch, one trivial eager kernel, event record            (the engine's workaround)
//   eager           the same kernels launched eagerly, event record                 (the control)
// each with default events and with hipEventDisableTiming events (what torch records), graphs captured with one stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void bump(unsigned* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1; }

// every thread spins for `spin` clock ticks, then writes the step number: the longer the kernel, the wider the window in which a
// prematurely completed event lets the checker run ahead
__global__ void slow_write(unsigned* buf, size_t n, const unsigned* counter, long long spin) {
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  const unsigned v = *counter;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v;
}

__global__ void check(const unsigned* buf, size_t n, const unsigned* counter, unsigned* stale) {
  const unsigned v = *counter;
  unsigned bad = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += buf[i] != v;
  if (bad) atomicAdd(stale, bad);
}

__global__ void trivial(unsigned* p) { if (threadIdx.x == 1000) *p = 0; }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, nk = argc > 2 ? atoi(argv[2]) : 4;
  const size_t n = 1 << 22;
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  unsigned *buf, *counter, *stale, *dummy;
  CK(hipMalloc(&buf, nk * n * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&stale, 4)); CK(hipMalloc(&dummy, 4));
  const long long spin = 200000;                             // ~0.1 ms per kernel
  auto launch_all = [&](hipStream_t s) {
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, counter);
    for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(slow_write, dim3(256), dim3(256), 0, s, buf + k * n, n, counter, spin);
  };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
  launch_all(A);
  CK(hipStreamEndCapture(A, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  int total_bad = 0;
  for (int evflags = 0; evflags < 2; ++evflags) {
    for (int variant = 0; variant < 3; ++variant) {
      CK(hipMemset(counter, 0, 4)); CK(hipMemset(buf, 0xff, nk * n * 4));
      CK(hipDeviceSynchronize());
      int bad_iters = 0;
      std::vector<hipEvent_t> evs;
      for (int it = 0; it < iters; ++it) {
        hipEvent_t e1, e2;
        CK(hipEventCreateWithFlags(&e1, evflags ? hipEventDisableTiming : hipEventDefault));
        CK(hipEventCreateWithFlags(&e2, evflags ? hipEventDisableTiming : hipEventDefault));
        evs.push_back(e1); evs.push_back(e2);
        CK(hipMemsetAsync(stale, 0, 4, B));
        if (variant == 2) launch_all(A); else CK(hipGraphLaunch(ge, A));
        if (variant == 1) hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, A, dummy);
        CK(hipEventRecord(e1, A));
        CK(hipStreamWaitEvent(B, e1, 0));
        for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, B, buf + k * n, n, counter, stale);
        CK(hipEventRecord(e2, B));
        CK(hipStreamWaitEvent(A, e2, 0));                    // the next graph launch must not overwrite what B is still checking
        unsigned h = 0;
        CK(hipMemcpyAsync(&h, stale, 4, hipMemcpyDeviceToHost, B));
        CK(hipStreamSynchronize(B));
        bad_iters += h != 0;
        if ((it & 255) == 255) { for (hipEvent_t e : evs) CK(hipEventDestroy(e)); evs.clear(); }
      }
      for (hipEvent_t e : evs) CK(hipEventDestroy(e));
      CK(hipDeviceSynchronize());
      static const char* names[3] = {"direct", "kernel-between", "eager"};
      printf("events %-14s %-15s stale iterations: %d of %d\n", evflags ? "disable-timing" : "default", names[variant], bad_iters, iters);
      fflush(stdout);
      total_bad += variant != 2 ? bad_iters : 0;
    }
  }
  // ---- second pattern: what the piecewise step really does.  Graph G1 (writers) and graph G2 (its dependent readers) are launched on
  // the SAME stream; between them only the event choreography of an asynchronous collective: record on A, wait on B, a small kernel
  // on B (the one-rank all-reduce), and - as in the engine - A does NOT wait for B before launching G2.  Stream order alone must make
  // G2 see what G1 wrote.
  hipGraph_t g2; hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, A, buf + k * n, n, counter, stale);
  CK(hipStreamEndCapture(A, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  int total_bad2 = 0;
  for (int variant = 0; variant < 4; ++variant) {
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(buf, 0xff, nk * n * 4)); CK(hipMemset(stale, 0, 4));
    CK(hipDeviceSynchronize());
    int bad_iters = 0;
    std::vector<hipEvent_t> evs;
    for (int it = 0; it < iters; ++it) {
      hipEvent_t e1, e2;
      CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      evs.push_back(e1); evs.push_back(e2);
      CK(hipGraphLaunch(ge, A));
      if (variant == 1 || variant == 3) hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, A, dummy);
      if (variant != 2) {                                    // variant 2: nothing between the two graph launches (control)
        CK(hipEventRecord(e1, A));
        CK(hipStreamWaitEvent(B, e1, 0));
        hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, B, dummy);
        CK(hipEventRecord(e2, B));
        if (variant == 3) CK(hipStreamWaitEvent(A, e2, 0));
      }
      CK(hipGraphLaunch(ge2, A));
      unsigned h = 0;
      CK(hipMemcpyAsync(&h, stale, 4, hipMemcpyDeviceToHost, A));
      CK(hipStreamSynchronize(A));
      CK(hipStreamSynchronize(B));
      bad_iters += h != 0;
      if (h) CK(hipMemset(stale, 0, 4));
      if ((it & 255) == 255) { for (hipEvent_t e : evs) CK(hipEventDestroy(e)); evs.clear(); }
    }
    for (hipEvent_t e : evs) CK(hipEventDestroy(e));
    static const char* names2[4] = {"G1, events, G2", "G1, kernel, events, G2", "G1, G2 (control)", "G1, kernel, events, wait back, G2"};
    printf("graph -> graph  %-36s stale iterations: %d of %d\n", names2[variant], bad_iters, iters);
    fflush(stdout);
    total_bad2 += bad_iters;
  }
  printf("RESULT %s ; %s\n", total_bad ? "event completes before the graph: reproduced" : "event after graph: not reproduced",
         total_bad2 ? "dependent graph runs ahead of its predecessor: reproduced" : "graph -> graph order: not reproduced");
  return 0;
}

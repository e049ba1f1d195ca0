// How fast can one CU take bytes into LDS by LDS-DMA, in the access shapes conv_dmap uses?  (The conv_dmap ablations of round 3 - the kernel
// with its DMA instructions but no MFMAs runs as long as the whole kernel, and out-of-range (zero-fill) DMAs cost as much as real ones - say the
// per-instruction rate bounds it: ~26 ns per 1-KiB wave instruction and CU.)  One workgroup per CU, W waves, a ring of S stages of I instructions
// per wave, counted vmcnt, optional barrier per stage; sources: an L2-resident table shared by every workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_rate.cpp -o /tmp/ldsdma_rate && /tmp/ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((address_space(3))) void* lds_void_p;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// SHAPE 0: every instruction 1 KiB contiguous; 1: 16 rows x 64 B at `stride` bytes; 2: 8 rows x 128 B at `stride`; 3: all lanes out of range
template <int I, int NBUF, int SHAPE, bool BARRIER, bool GLOBAL>
__global__ __launch_bounds__(512) void fill(const unsigned char* src, unsigned bytes, int stages, unsigned stride, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t r = make_rsrc(src, bytes);
  const unsigned stage_bytes = (unsigned)(nw * I * 1024);
  unsigned off;
  if (SHAPE == 1) off = (lane >> 2) * stride + (lane & 3) * 16;
  else if (SHAPE == 2) off = (lane >> 3) * stride + (lane & 7) * 16;
  else off = lane * 16;
  const unsigned rows = SHAPE == 1 ? 16 : SHAPE == 2 ? 8 : 0;
  unsigned pos = (blockIdx.x * 7919u) % 64u * 1024u;                  // workgroups start at different places of the shared table
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < I; ++i) {
      unsigned o = SHAPE == 3 ? 0x80000000u : (pos + (unsigned)((wid * I + i) * (rows ? rows * stride : 1024)) + off) % (bytes - 1024);
      o &= ~15u;
      if (GLOBAL) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o), (lds_void_p)(smem + buf * stage_bytes + (wid * I + i) * 1024), 16, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_p)(smem + buf * stage_bytes + (wid * I + i) * 1024), 16, o, 0, 0, 0);
    }
    pos += stage_bytes * 3;
  };
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) issue(b);
  int buf = NBUF - 1;
  for (int s = 0; s < stages; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * I) : "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    issue(buf);
    buf = buf + 1 == NBUF ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[17] == 0x5a) sink[0] = 1;
}

template <int I, int NBUF, int SHAPE, bool BARRIER, bool GLOBAL>
static void run(const char* what, int waves, const unsigned char* src, unsigned bytes, unsigned stride, unsigned* sink) {
  const int stages = 400, smem = NBUF * waves * I * 1024;
  if (smem > 160 * 1024) return;
  auto k = fill<I, NBUF, SHAPE, BARRIER, GLOBAL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), smem, 0, src, bytes, stages, stride, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double per_cu = (double)(stages + NBUF - 1) * waves * I * 1024 / (best * 1e-3) / 1e9;
  printf("%-34s waves %d  instr/wave/stage %2d  ring %d (%3d KiB)  %s  %6.1f GB/s per CU  %5.1f ns per 1-KiB instruction\n", what, waves, I, NBUF, smem >> 10,
         BARRIER ? "barrier" : "no barr", per_cu, 1024.0 / per_cu);
  fflush(stdout);
}

int main() {
  const unsigned bytes = 2u << 20;                                   // 2 MiB table: resident in every XCD's L2
  unsigned char* src; unsigned* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 4));
#define ROW(I, NB, SH, BAR, GL, what, stride) for (int w : {1, 2, 4, 8}) run<I, NB, SH, BAR, GL>(what, w, src, bytes, stride, sink);
  ROW(8, 3, 0, true, false, "1 KiB contiguous, buffer", 0)
  ROW(8, 3, 1, true, false, "16 rows x 64 B @256, buffer", 256)
  ROW(8, 3, 2, true, false, "8 rows x 128 B @256, buffer", 256)
  ROW(8, 3, 3, true, false, "all lanes out of range", 0)
  ROW(8, 3, 0, true, true, "1 KiB contiguous, global_load_lds", 0)
  ROW(8, 3, 1, true, true, "16 rows x 64 B @256, global_load_lds", 256)
  ROW(8, 3, 0, false, false, "1 KiB contiguous, buffer", 0)
  ROW(4, 5, 0, true, false, "1 KiB contiguous, buffer", 0)
  ROW(4, 5, 1, true, false, "16 rows x 64 B @256, buffer", 256)
  ROW(2, 8, 0, true, false, "1 KiB contiguous, buffer", 0)
  ROW(16, 3, 0, true, false, "1 KiB contiguous, buffer", 0)
  ROW(8, 3, 1, true, false, "16 rows x 64 B @2048, buffer", 2048)
  return 0;
}

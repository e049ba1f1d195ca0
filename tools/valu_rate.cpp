// Issue cost of the vector instructions of the in-place BatchNorm pass on gfx950, per wave64 instruction: one wave per SIMD runs 64 independent
// copies of one instruction per loop iteration (inline asm, so nothing is folded away), s_memtime around 256 iterations.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.cpp -o scratch/valu_rate && scratch/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP> __global__ __launch_bounds__(1024) void rate(unsigned long long* out, int waves_per_simd) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  float a = threadIdx.x * 1.0f, b = 1.5f, c = 0.25f;
  f2 p = {a, b}, q = {b, c}, r = {c, a};
  unsigned u = threadIdx.x, w = 0, w1 = 0, w2 = 0, w3 = 0;
  float a1 = a + 1, a2 = a + 2, a3 = a + 3;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(r));) }
    if (OP == 2) { REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(b), "v"(c));) }
    if (OP == 3) { REP64(asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(w) : "v"(u), "v"(u));) }
    if (OP == 4) { REP64(asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w) : "v"(u));) }
    if (OP == 5) { REP64(asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w) : "v"(u));) }
    if (OP == 6) { REP64(asm volatile("v_xor_b32 %0, 32, %1" : "=v"(w) : "v"(u));) }
    if (OP == 7) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w) : "v"(u), "v"(u));) }
    if (OP == 8) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));) }
    if (OP == 9) { REP64(asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 10) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");) }
    if (OP == 11) { REP64(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(q), "v"(r));) }
    if (OP == 12) { REP64(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(u), "v"(u), "v"(u));) }
    if (OP == 13) { REP64(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 14) { unsigned sw_ = 0; REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(sw_));) w += sw_; }
    if (OP == 15) { REP64(asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(u));) }
#define REP16x4(X0, X1, X2, X3) REP8(X0 X1 X2 X3) REP8(X0 X1 X2 X3)
    if (OP == 16) { REP16x4(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c));,
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c));, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 17) { REP16x4(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(b), "v"(c));, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(b), "v"(c));,
                            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(b), "v"(c));, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(b), "v"(c));) }
    if (OP == 18) { REP16x4(asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w) : "v"(u));, asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w1) : "v"(u));,
                            asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w2) : "v"(u));, asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w3) : "v"(u));) }
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (a + a1 + a2 + a3 + p.x + (float)(w + w1 + w2 + w3) + (float)u == 12345.678f) out[1] = 1;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_pk_max_i16", "v_lshlrev_b32", "v_and_b32 (literal)", "v_xor_b32", "v_cndmask_b32", "v_permlane32_swap_b32",
                         "v_add_f32", "v_cmp_lt_f32", "v_pk_mul_f32", "v_perm_b32", "v_fmac_f32", "s_add_u32", "v_mov_b32", "v_fma_f32 x4 chains", "v_cvt_pk_bf16_f32 x4 dst", "v_lshlrev_b32 x4 dst"};
  for (int threads = 256; threads <= 1024; threads *= 2)
    for (int op = 0; op < 19; ++op) {
      unsigned long long h[2] = {0, 0};
      hipMemset(d, 0, 16);
#define GO(N) case N: hipLaunchKernelGGL(rate<N>, dim3(1), dim3(threads), 0, 0, d, threads / 256); break;
      switch (op) { GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) GO(9) GO(10) GO(11) GO(12) GO(13) GO(14) GO(15) GO(16) GO(17) GO(18) }
      hipDeviceSynchronize();
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%d wave(s) per SIMD  %-24s %6.2f clocks per instruction and wave\n", threads / 256, names[op], (double)h[0] / (256.0 * 64.0));
    }
  return 0;
}

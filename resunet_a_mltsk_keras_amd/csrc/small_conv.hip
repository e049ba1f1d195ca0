// Few-channel 1x1 convolutions that do not fill an MFMA tile: the stem (3/6/7 bands -> 32,
// model2.py:101) and the heads (32 -> num_classes / 3 with softmax / sigmoid, model2.py:145-188).
// Both are HBM-bound; weights live in LDS / registers, one thread per pixel (or pixel x piece).
#include "common.h"
#include <stdlib.h>

template <typename T> __device__ __forceinline__ void st8(unsigned char* base, size_t elem_off, const float* f) {
  if constexpr (sizeof(T) == 2) stg16(base + elem_off * 2, ET<T>::pack(f));
  else { stg16(base + elem_off * 4, ET<T>::pack(f)); stg16(base + elem_off * 4 + 16, ET<T>::pack(f + 4)); }
}
template <typename T> __device__ __forceinline__ void ld8(const unsigned char* base, size_t elem_off, float* f) {
  if constexpr (sizeof(T) == 2) ET<T>::unpack(ldg16(base + elem_off * 2), f);
  else { ET<T>::unpack(ldg16(base + elem_off * 4), f); ET<T>::unpack(ldg16(base + elem_off * 4 + 16), f + 4); }
}

// ---- stem ------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float stored_value(float v);          // v as the storage type holds it
template <> __device__ __forceinline__ float stored_value<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored_value<bf16_t>(float v) { return (float)(__bf16)v; }
template <typename T, bool STATS>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        unsigned char* y, long long M, int Cin, int Cout, double* stats, int replicas) {
  extern __shared__ float sw[];                       // [Cout][Cin] then [Cout]; STATS: then [4 waves][2][Cout] partial sums
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) sw[Cout * Cin + i] = b ? b[i] : 0.f;
  __syncthreads();
  const int CG8 = Cout / 8;
  const long long total = M * CG8;
  const bool pow2 = (CG8 & (CG8 - 1)) == 0;
  const int cgsh = __builtin_ctz((unsigned)CG8);
  // STATS (rua_stem_fwd_stats): per-channel sum / sum of squares of the output AS STORED (rounded to the storage type), the statistics the
  // first BatchNorm of the encoder wants - the rua_col_stats pass over the tensor just written disappears.  CG8 divides 256 and the grid
  // stride, so a thread keeps its channel group for the whole sweep and carries the 16 partial sums in registers
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = pow2 ? (i >> cgsh) : i / CG8; const int cg = (int)(i - m * CG8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = sw[Cout * Cin + cg * 8 + j];
    for (int c = 0; c < Cin; ++c) {
      const float xv = x[m * Cin + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(xv, sw[(cg * 8 + j) * Cin + c], o[j]);
    }
    st8<T>(y, (size_t)m * Cout + cg * 8, o);
    if constexpr (STATS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = stored_value<T>(o[j]); s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
    }
  }
  if constexpr (STATS) {
    // lanes l and l ^ o (o = CG8 .. 32) hold the same channel group: fold with shuffles, then over the four waves through LDS, one fp64 atomic per
    // channel, sum and block into the block's replica
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, cg = threadIdx.x % CG8;
    for (int o = CG8; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
    }
    float* red = sw + Cout * Cin + Cout;
    if (lane < CG8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[(wid * 2) * Cout + cg * 8 + j] = s1[j]; red[(wid * 2 + 1) * Cout + cg * 8 + j] = s2[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cout; i += 256) {
      const float t = (red[i] + red[2 * Cout + i]) + (red[4 * Cout + i] + red[6 * Cout + i]);
      unsafeAtomicAdd(&stats[(size_t)(blockIdx.x % replicas) * 2 * Cout + i], (double)t);
    }
  }
}

// Register form of the stem (Cin <= 8, Cout / 8 dividing 256: the reference's 3 / 6 / 7 bands -> 32): a thread keeps its channel group for the whole
// sweep, so its 8 x Cin weights live in registers (the LDS form above issues 8 x Cin broadcast reads per pixel piece) and the input values of the NEXT
// pixel are loaded before the arithmetic of this one.  CINT: Cin as a template constant (3, 6, 7; 0: runtime - every `c < Cin` test of a runtime Cin
// became a branch around one scalar load).  Same order of operations per output value: bit-identical to the LDS form.
// xpack (optional, Cin <= 7): the input once more as bf16 [M][16] = { hi(x_0..x_7) | lo(x_0..x_6), 1 } with hi = bf16(x), lo = bf16(x - hi) - the operand
// the weight gradient of the stem takes on the matrix pipe (rua_stem_bwd_fold): dW = dy^T . (hi + lo) to ~16 mantissa bits, the column of ones gives db.
template <typename T, bool STATS, int CINT>
__global__ __launch_bounds__(256) void stem_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            unsigned char* y, long long M, int Cin_rt, int Cout, double* stats, int replicas,
                                                            unsigned char* xpack) {
  constexpr int CINP = 8;
  const int Cin = CINT > 0 ? CINT : Cin_rt;
  extern __shared__ float sw[];                       // STATS: [4 waves][2][Cout] partial sums
  const int CG8 = Cout / 8;
  const int cg = threadIdx.x % CG8;
  float wr[8][CINP], bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bias[j] = b ? b[cg * 8 + j] : 0.f;
#pragma unroll
    for (int c = 0; c < CINP; ++c) wr[j][c] = c < Cin ? w[(cg * 8 + j) * Cin + c] : 0.f;
  }
  const long long total = M * CG8;
  const long long stride = (long long)gridDim.x * 256;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float xv[CINP], xn[CINP];
  const int cgsh = __builtin_ctz((unsigned)CG8);      // CG8 divides 256: a power of two (m = i / CG8 as a 64-bit division was ~100 vector instructions per piece)
  auto fetch = [&](long long ii, float* v) {
    const long long m = ii >> cgsh;
    if constexpr (CINT == 6) {                          // 24-byte rows: three 8-byte loads
      const float2* q = reinterpret_cast<const float2*>(x + m * 6);
      const float2 a = q[0], b2 = q[1], c2 = q[2];
      v[0] = a.x; v[1] = a.y; v[2] = b2.x; v[3] = b2.y; v[4] = c2.x; v[5] = c2.y; v[6] = 0.f; v[7] = 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < CINP; ++c) v[c] = c < Cin ? x[m * Cin + c] : 0.f;
    }
  };
#pragma unroll
  for (int c = 0; c < CINP; ++c) { xv[c] = 0.f; xn[c] = 0.f; }
  if (i < total) fetch(i, xv);
  for (; i < total; i += stride) {
    if (i + stride < total) fetch(i + stride, xn);
    const long long m = i >> cgsh;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bias[j];
#pragma unroll
    for (int c = 0; c < CINP; ++c) {
      if (c < Cin) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(xv[c], wr[j][c], o[j]);
      }
    }
    st8<T>(y, (size_t)m * Cout + cg * 8, o);
    if (xpack && cg < 2) {                              // threads cg = 0 / 1 of the pixel write the two 16-byte halves of its packed row
      float hf[8], lf[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { hf[c] = stored_value<bf16_t>(xv[c]); lf[c] = xv[c] - hf[c]; }
      lf[7] = 1.f;
      st8<bf16_t>(xpack, (size_t)m * 16 + cg * 8, cg == 0 ? hf : lf);
    }
    if constexpr (STATS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = stored_value<T>(o[j]); s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
    }
#pragma unroll
    for (int c = 0; c < CINP; ++c) xv[c] = xn[c];
  }
  if constexpr (STATS) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = CG8; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
    }
    float* red = sw;
    if (lane < CG8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[(wid * 2) * Cout + cg * 8 + j] = s1[j]; red[(wid * 2 + 1) * Cout + cg * 8 + j] = s2[j]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * Cout; k += 256) {
      const float t = (red[k] + red[2 * Cout + k]) + (red[4 * Cout + k] + red[6 * Cout + k]);
      unsafeAtomicAdd(&stats[(size_t)(blockIdx.x % replicas) * 2 * Cout + k], (double)t);
    }
  }
}

// dW[co][c] += tmp[co][c] + tmp[co][8 + c], db[co] += tmp[co][15], tmp := 0 (tmp [Cout][16]: what the 1x1 weight gradient of dy against the packed
// input of stem_fwd_reg_kernel leaves: hi and lo halves of x, the column of ones)
__global__ void stem_bwd_fold_kernel(float* tmp, float* dw, float* db, int Cin, int Cout) {
  for (int i = threadIdx.x; i < Cout * 16; i += blockDim.x) {
    const int co = i >> 4, c = i & 15;
    const float v = tmp[i];
    if (c < Cin) dw[co * Cin + c] += v + tmp[co * 16 + 8 + c];
    else if (c == 15 && db) db[co] += v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cout * 16; i += blockDim.x) tmp[i] = 0.f;
}
extern "C" int rua_stem_bwd_fold(float* tmp, float* dw, float* db, int Cin, int Cout, void* stream) {
  RUA_CHECK_ARG(tmp && dw && Cin >= 1 && Cin <= 7 && Cout >= 1, "rua_stem_bwd_fold: bad arguments (Cin=%d must be in 1..7)", Cin);
  hipLaunchKernelGGL(stem_bwd_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tmp, dw, db, Cin, Cout);
  RUA_LAUNCH_CHECK("rua_stem_bwd_fold");
  return RUA_OK;
}

// STEM_NT threads per block: one block per CU (every block ends with one float atomic per weight, and same-address atomics
// serialise), so the block is what fills the CU: 16 waves keep 4x the loads in flight of the former 4 (40 -> 15 us, 46 MB).
template <typename T, int CINP, int STEM_NT>      // 1024 threads for Cin <= 8 (64 accumulators per thread), 256 for Cin <= 16 (128: would spill)
__global__ __launch_bounds__(STEM_NT) void stem_bwd_kernel(const float* __restrict__ x, const unsigned char* dy, float* dw, float* db,
                                                            long long M, int Cin, int Cout, int rows_per_block) {
  extern __shared__ float red[];                      // [STEM_NT / 64 waves][Cout][CINP+1]
  const int CG8 = Cout / 8;                           // power of two <= 32 (checked by the launcher)
  const int cg = threadIdx.x % CG8, pl = threadIdx.x / CG8, PL = STEM_NT / CG8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int NE = Cout * (CINP + 1);
  float acc[8][CINP], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bs[j] = 0.f;
#pragma unroll
    for (int c = 0; c < CINP; ++c) acc[j][c] = 0.f; }
  long long r = (long long)blockIdx.x * rows_per_block + pl;
  long long rend = (long long)(blockIdx.x + 1) * rows_per_block; if (rend > M) rend = M;
  auto fma_row = [&](const float* g, const float* xv) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bs[j] += g[j];
#pragma unroll
      for (int c = 0; c < CINP; ++c) acc[j][c] = fmaf(g[j], xv[c], acc[j][c]); }
  };
  for (; r + PL < rend; r += 2 * PL) {                 // two rows per iteration, all loads before the first use
    float g0[8], g1[8], x0[CINP], x1[CINP];
    ld8<T>(dy, (size_t)r * Cout + cg * 8, g0);
    ld8<T>(dy, (size_t)(r + PL) * Cout + cg * 8, g1);
#pragma unroll
    for (int c = 0; c < CINP; ++c) { x0[c] = c < Cin ? x[r * Cin + c] : 0.f; x1[c] = c < Cin ? x[(r + PL) * Cin + c] : 0.f; }
    fma_row(g0, x0); fma_row(g1, x1);
  }
  for (; r < rend; r += PL) {
    float g[8], xv[CINP];
    ld8<T>(dy, (size_t)r * Cout + cg * 8, g);
#pragma unroll
    for (int c = 0; c < CINP; ++c) xv[c] = c < Cin ? x[r * Cin + c] : 0.f;
    fma_row(g, xv);
  }
  // lanes l and l ^ o (o = CG8 .. 32) hold the same 8 output channels: fold with shuffles (the previous version sent 72 LDS
  // float atomics per thread into 9 x Cout addresses: 88 us for a 46 MB pass), then one LDS slot per wave
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int c = 0; c < CINP; ++c)
      for (int o = CG8; o < 64; o <<= 1) acc[j][c] += __shfl_xor(acc[j][c], o, 64);
    for (int o = CG8; o < 64; o <<= 1) bs[j] += __shfl_xor(bs[j], o, 64);
  }
  if (lane < CG8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int c = 0; c < CINP; ++c) red[wid * NE + (lane * 8 + j) * (CINP + 1) + c] = acc[j][c];
      red[wid * NE + (lane * 8 + j) * (CINP + 1) + CINP] = bs[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NE; i += STEM_NT) {
    const int co = i / (CINP + 1), c = i - co * (CINP + 1);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < STEM_NT / 64; ++w) v += red[w * NE + i];
    if (c < Cin) unsafeAtomicAdd(&dw[co * Cin + c], v);
    else if (c == CINP && db) unsafeAtomicAdd(&db[co], v);
  }
}

static int stem_fwd_launch(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, double* stats, int replicas, void* stream,
                           const char* who, void* xpack = nullptr) {
  RUA_CHECK_ARG(x && w && y && M > 0, "%s: bad arguments", who);
  RUA_CHECK_ARG(Cin >= 1 && Cin <= 16, "%s: Cin=%d must be in 1..16", who, Cin);
  RUA_CHECK_ARG(Cout % 8 == 0 && Cout <= 256, "%s: Cout=%d must be a multiple of 8 (<=256)", who, Cout);
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (M * (Cout / 8) + 255) / 256;
  const bool reg_ok = Cin <= 8 && 256 % (Cout / 8) == 0;
  RUA_CHECK_ARG(!xpack || (reg_ok && Cin <= 7 && Cout >= 16), "%s: the packed input needs Cin <= 7 and Cout / 8 a power of two >= 2", who);
  if (reg_ok && (g_tune.stem_reg || xpack)) {
    RUA_CHECK_ARG(!stats || replicas >= 1, "%s: replicas=%d", who, replicas);
    const int64_t cap = (stats ? 4 : 8) * (int64_t)rua_cu_count();      // STATS: every block ends with 2 Cout fp64 atomics into its replica
    if (g > cap) g = cap;
    const size_t smem = (size_t)(8 * Cout) * 4;
#define RUA_STEM_GO(T_, ST_, CI_) hipLaunchKernelGGL((stem_fwd_reg_kernel<T_, ST_, CI_>), dim3((int)g), dim3(256), smem, st, x, w, b, (unsigned char*)y, (long long)M, Cin, Cout, \
                                                     stats, stats ? replicas : 1, (unsigned char*)xpack)
#define RUA_STEM_CI(T_, ST_) do { if (Cin == 6) RUA_STEM_GO(T_, ST_, 6); else if (Cin == 3) RUA_STEM_GO(T_, ST_, 3); else if (Cin == 7) RUA_STEM_GO(T_, ST_, 7); else RUA_STEM_GO(T_, ST_, 0); } while (0)
    if (stats) { if (dtype == RUA_BF16) RUA_STEM_CI(bf16_t, true); else RUA_STEM_CI(float, true); }
    else { if (dtype == RUA_BF16) RUA_STEM_CI(bf16_t, false); else RUA_STEM_CI(float, false); }
#undef RUA_STEM_CI
#undef RUA_STEM_GO
    RUA_LAUNCH_CHECK(who);
    return RUA_OK;
  }
  if (stats) {
    // (every block ends with 2 Cout fp64 atomics into its replica: few, long blocks - two per CU)
    RUA_CHECK_ARG(replicas >= 1 && 256 % (Cout / 8) == 0, "%s: replicas=%d, Cout=%d", who, replicas, Cout);
    const int64_t cap = 2 * (int64_t)rua_cu_count();
    if (g > cap) g = cap;
    const size_t smem = (size_t)(Cout * Cin + Cout + 8 * Cout) * 4;
    if (dtype == RUA_BF16) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, true>), dim3((int)g), dim3(256), smem, st, x, w, b, (unsigned char*)y, (long long)M, Cin, Cout, stats, replicas);
    else hipLaunchKernelGGL((stem_fwd_kernel<float, true>), dim3((int)g), dim3(256), smem, st, x, w, b, (unsigned char*)y, (long long)M, Cin, Cout, stats, replicas);
  } else {
    if (g > 4096) g = 4096;
    const size_t smem = (size_t)(Cout * Cin + Cout) * 4;
    if (dtype == RUA_BF16) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, false>), dim3((int)g), dim3(256), smem, st, x, w, b, (unsigned char*)y, (long long)M, Cin, Cout, nullptr, 1);
    else hipLaunchKernelGGL((stem_fwd_kernel<float, false>), dim3((int)g), dim3(256), smem, st, x, w, b, (unsigned char*)y, (long long)M, Cin, Cout, nullptr, 1);
  }
  RUA_LAUNCH_CHECK(who);
  return RUA_OK;
}
extern "C" int rua_stem_fwd(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, void* stream) {
  return stem_fwd_launch(x, w, b, y, M, Cin, Cout, dtype, nullptr, 1, stream, "rua_stem_fwd");
}
extern "C" int rua_stem_fwd_stats(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, double* stats, int replicas, void* stream) {
  RUA_CHECK_ARG(stats != nullptr, "rua_stem_fwd_stats: stats is NULL");
  return stem_fwd_launch(x, w, b, y, M, Cin, Cout, dtype, stats, replicas, stream, "rua_stem_fwd_stats");
}
extern "C" int rua_stem_fwd_pack(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, double* stats, int replicas,
                                 void* xpack, void* stream) {
  RUA_CHECK_ARG(xpack != nullptr, "rua_stem_fwd_pack: xpack is NULL");
  return stem_fwd_launch(x, w, b, y, M, Cin, Cout, dtype, stats, replicas, stream, "rua_stem_fwd_pack", xpack);
}

extern "C" int rua_stem_bwd(const float* x, const void* dy, float* dw, float* db, int64_t M, int Cin, int Cout, int dtype, void* stream) {
  RUA_CHECK_ARG(x && dy && dw && M > 0, "rua_stem_bwd: bad arguments");
  RUA_CHECK_ARG(Cin >= 1 && Cin <= 16, "rua_stem_bwd: Cin=%d must be in 1..16", Cin);
  RUA_CHECK_ARG(Cout % 8 == 0 && Cout <= 256 && 256 % (Cout / 8) == 0, "rua_stem_bwd: unsupported Cout=%d", Cout);
  // every block ends with one float atomic per weight: same-address float atomics serialise (~25 ns each), so few blocks
  const int64_t nblk = g_tune.stem_blocks > 0 ? g_tune.stem_blocks : rua_cu_count();
  int64_t blocks = nblk; int64_t rpb = (M + blocks - 1) / blocks; if (rpb < 64) rpb = 64;
  const int g = (int)((M + rpb - 1) / rpb);
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* d = (const unsigned char*)dy;
  if (Cin <= 8) {
    const size_t smem = (size_t)16 * Cout * 9 * 4;
    if (smem <= 64 * 1024 && 1024 % (Cout / 8) == 0) {
      if (dtype == RUA_BF16) hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 8, 1024>), dim3(g), dim3(1024), smem, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
      else hipLaunchKernelGGL((stem_bwd_kernel<float, 8, 1024>), dim3(g), dim3(1024), smem, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
    } else {
      if (dtype == RUA_BF16) hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 8, 256>), dim3(g), dim3(256), smem / 4, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
      else hipLaunchKernelGGL((stem_bwd_kernel<float, 8, 256>), dim3(g), dim3(256), smem / 4, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
    }
  } else {
    const size_t smem = (size_t)4 * Cout * 17 * 4;
    if (dtype == RUA_BF16) hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 16, 256>), dim3(g), dim3(256), smem, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
    else hipLaunchKernelGGL((stem_bwd_kernel<float, 16, 256>), dim3(g), dim3(256), smem, st, x, d, dw, db, (long long)M, Cin, Cout, (int)rpb);
  }
  RUA_LAUNCH_CHECK("rua_stem_bwd");
  return RUA_OK;
}

// ---- heads -------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const unsigned char* x, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* z, float* p, long long M, int Cin, int Cout, int act) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float sw[];                       // [Cout][Cin] + [Cout]
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) sw[Cout * Cin + i] = b ? b[i] : 0.f;
  __syncthreads();
  const int CGI = Cin / VEC;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = co < Cout ? sw[Cout * Cin + co] : 0.f;
    for (int cp = 0; cp < CGI; ++cp) {
      float xv[VEC];
      ET<T>::unpack(ldg16(x + ((size_t)m * CGI + cp) * 16), xv);
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        if (co < Cout) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[co] = fmaf(xv[j], sw[co * Cin + cp * VEC + j], acc[co]);
        }
      }
    }
    float pr[8];
    if (act == RUA_ACT_SOFTMAX) {
      float mx = -INFINITY;
#pragma unroll
      for (int co = 0; co < 8; ++co) if (co < Cout) mx = fmaxf(mx, acc[co]);
      float s = 0.f;
#pragma unroll
      for (int co = 0; co < 8; ++co) { pr[co] = co < Cout ? expf(acc[co] - mx) : 0.f; s += pr[co]; }
      const float inv = 1.f / s;
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] *= inv;
    } else if (act == RUA_ACT_SIGMOID) {
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] = 1.f / (1.f + expf(-acc[co]));
    } else {
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] = acc[co];
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) if (co < Cout) { if (z) z[m * Cout + co] = acc[co]; p[m * Cout + co] = pr[co]; }
  }
}

// ---- head forward, second form (Cin = 32: the reference's heads) -------------------------------------------------------------
// CGI = Cin / VEC lanes share a pixel: each loads ONE 16-byte piece (a wave reads 1 KiB runs), keeps its slice of the weights in
// registers (the first form re-read them from LDS: 192 broadcast reads per pixel, 22 us for a 59 MB pass), the partial dot products
// are folded over the CGI lanes by xor shuffles, every lane finishes the activation, and lane cp stores / accounts for the classes
// cp, cp + CGI: logits, probabilities, and - LOSS - the Tanimoto moments and the accuracy / confusion counts of those classes.
template <typename T, int CO, bool LOSS>
__global__ __launch_bounds__(256) void head_fwd2_kernel(const unsigned char* x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* z, float* p, const float* __restrict__ y, double* sums, double* metrics,
                                                         long long HW, int pix_per_block, int Cin, int Cout, int act) {
  constexpr int VEC = ET<T>::VEC;
  constexpr int CGI = 32 / VEC;                       // 4 (bf16) or 8 (fp32) lanes per pixel
  constexpr int PL = 256 / CGI;                       // pixels per pass and block
  constexpr int NOWN = (8 + CGI - 1) / CGI;           // classes a lane owns: cp, cp + CGI
  __shared__ float sh[4 * 56];
  const int cp = threadIdx.x % CGI, pl = threadIdx.x / CGI;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = blockIdx.y;
  float wr[CO][VEC], bias[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    bias[co] = (co < Cout && b) ? b[co] : 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) wr[co][j] = co < Cout ? w[co * Cin + cp * VEC + j] : 0.f;
  }
  float ts[NOWN][6], mt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < NOWN; ++o)
#pragma unroll
    for (int k = 0; k < 6; ++k) ts[o][k] = 0.f;
  long long i = (long long)blockIdx.x * pix_per_block + pl;
  long long iend = (long long)(blockIdx.x + 1) * pix_per_block; if (iend > HW) iend = HW;
  // the loads of pass k + 1 are issued before the arithmetic of pass k (a block makes ~8 passes, each a chain load -> dot products ->
  // shuffles -> exp -> stores: with one pass in flight per wave the launch ran at 2 TB/s)
  uint4 xq = make_uint4(0, 0, 0, 0);
  float yv[NOWN];
#pragma unroll
  for (int o = 0; o < NOWN; ++o) yv[o] = 0.f;
  auto fetch = [&](long long ii, uint4& q, float* yy) {
    const long long mm_ = (long long)n * HW + ii;
    q = ldg16(x + ((size_t)mm_ * CGI + cp) * 16);
    if constexpr (LOSS) {
#pragma unroll
      for (int o = 0; o < NOWN; ++o) yy[o] = (cp + o * CGI) < Cout ? y[mm_ * Cout + cp + o * CGI] : 0.f;
    }
  };
  if (i < iend) fetch(i, xq, yv);
  for (; i < iend; i += PL) {
    const long long m = (long long)n * HW + i;
    uint4 xq_n = make_uint4(0, 0, 0, 0);
    float yv_n[NOWN];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) yv_n[o] = 0.f;
    if (i + PL < iend) fetch(i + PL, xq_n, yv_n);
    float xv[VEC], acc[CO];
    ET<T>::unpack(xq, xv);
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) a = fmaf(xv[j], wr[co][j], a);
#pragma unroll
      for (int o = 1; o < CGI; o <<= 1) a += __shfl_xor(a, o, 64);
      acc[co] = a + bias[co];
    }
    float pr[CO];
    if (act == RUA_ACT_SOFTMAX) {
      float mx = -INFINITY;
#pragma unroll
      for (int co = 0; co < CO; ++co) if (co < Cout) mx = fmaxf(mx, acc[co]);
      float s_ = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) { pr[co] = co < Cout ? expf(acc[co] - mx) : 0.f; s_ += pr[co]; }
      const float inv = 1.f / s_;
#pragma unroll
      for (int co = 0; co < CO; ++co) pr[co] *= inv;
    } else if (act == RUA_ACT_SIGMOID) {
#pragma unroll
      for (int co = 0; co < CO; ++co) pr[co] = 1.f / (1.f + expf(-acc[co]));
    } else {
#pragma unroll
      for (int co = 0; co < CO; ++co) pr[co] = acc[co];
    }
    // this lane's classes: select from the register arrays with compile-time indices
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
      float zo = 0.f, po = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) if (co % CGI == cp && co / CGI == o) { zo = acc[co]; po = pr[co]; }
      const int c = cp + o * CGI;
      if (c < Cout) {
        if (z) z[m * Cout + c] = zo;
        p[m * Cout + c] = po;
        if constexpr (LOSS) {
          const float a = po, l = yv[o], q = 1.f - a, mm = 1.f - l;
          ts[o][0] += a; ts[o][1] += mm; ts[o][2] = fmaf(a, l, ts[o][2]);
          ts[o][3] += a * a + l * l; ts[o][4] = fmaf(q, mm, ts[o][4]); ts[o][5] += q * q + mm * mm;
          const bool t = l > 0.5f, qq = a > 0.5f;
          mt[1] += (t && qq); mt[2] += (!t && qq); mt[3] += (!t && !qq); mt[4] += (t && !qq);
        }
      }
    }
    if constexpr (LOSS) {
      if (metrics) {                                  // categorical accuracy: argmax p (every lane has all of p) vs argmax y (spread over the lanes)
        int ip = 0; float bp = pr[0];
#pragma unroll
        for (int co = 1; co < CO; ++co) if (co < Cout && pr[co] > bp) { bp = pr[co]; ip = co; }
        float by = -INFINITY; int iy = 1 << 20;
#pragma unroll
        for (int o = 0; o < NOWN; ++o) { const int c = cp + o * CGI; if (c < Cout && (yv[o] > by)) { by = yv[o]; iy = c; } }
#pragma unroll
        for (int o = 1; o < CGI; o <<= 1) {
          const float ov = __shfl_xor(by, o, 64); const int oi = __shfl_xor(iy, o, 64);
          if (ov > by || (ov == by && oi < iy)) { by = ov; iy = oi; }     // first maximum wins, as a scan in class order
        }
        if (cp == 0) mt[0] += (ip == iy);
      }
    }
    xq = xq_n;
#pragma unroll
    for (int o = 0; o < NOWN; ++o) yv[o] = yv_n[o];
  }
  if constexpr (LOSS) {
    // lanes with the same cp (stride CGI in the wave) hold partial sums of the same classes
#pragma unroll
    for (int o = 0; o < NOWN; ++o)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int of = CGI; of < 64; of <<= 1) ts[o][k] += __shfl_xor(ts[o][k], of, 64);
      }
#pragma unroll
    for (int k = 0; k < 5; ++k) mt[k] = wave_sum(mt[k]);
    if (lane < CGI) {
#pragma unroll
      for (int o = 0; o < NOWN; ++o) {
        const int c = cp + o * CGI;
        if (c < 8) {
#pragma unroll
          for (int k = 0; k < 6; ++k) sh[wid * 56 + c * 6 + k] = ts[o][k];
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sh[wid * 56 + 48 + k] = mt[k];
    }
    __syncthreads();
    if (threadIdx.x < 53) {
      const int e = threadIdx.x;
      const double t = (double)sh[e] + (double)sh[56 + e] + (double)sh[112 + e] + (double)sh[168 + e];
      if (e < 48) { const int c = e / 6, k = e % 6; if (c < Cout && sums) unsafeAtomicAdd(&sums[((size_t)n * Cout + c) * 6 + k], t); }
      else if (metrics) unsafeAtomicAdd(&metrics[e - 48], t);
    }
  }
}

// ---- head forward, third form (bf16 storage, Cin = 32): the 32 -> Cout product on the matrix pipe ---------------------------------
// head_fwd2 is bound by vector-instruction issue (ISA census: ~430 vector instructions per 16 pixels and wave - four lanes share a pixel and each
// of them redoes the softmax; PMC: 4 waves per SIMD, each 28 % active): a wave takes 64 pixels per pass instead,
//   D^T[co][px] = W[co][ci] . X^T[ci][px]  as  mfma_f32_32x32x16_bf16  (A = the weights, rows >= Cout zero; B = 16-byte pieces of the pixels' rows:
//   lane = (pixel & 31, ci half) - two k-steps per 32 pixels),
// with the fp32 weights split into bf16 hi + lo parts (two products: the weights enter with ~16 mantissa bits, the inputs are bf16 anyway, so
// the logits agree with the vector form's to ~1e-5 relative - nothing like a bf16 rounding of the weights).  Pixels 0 - 31 of a pass go
// through one accumulator, pixels 32 - 63 through a second one; class rows 0 - 3 sit in lanes 0 - 31 and rows 4 - 7 in lanes 32 - 63, so four
// v_permlane32_swap leave EVERY lane with all classes of ITS pixel (lane l: pixel l of the pass): bias, softmax / sigmoid, stores,
// Tanimoto moments and confusion counts run once per pixel, one lane each.  The loads of pass k + 1 are issued before the arithmetic of pass k.
template <int CO, bool LOSS>
__global__ __launch_bounds__(256) void head_fwd3_kernel(const unsigned char* x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* z, float* p, const float* __restrict__ y, double* sums, double* metrics,
                                                         long long HW, int pix_per_block, int Cout, int act, int replicas) {
  __shared__ float sh[4 * 56];
  constexpr bool EXACT = CO != 8;                     // CO = 6 / 3 are launched for Cout == CO only: no runtime per-class tests
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int half = lane >> 5, pl = lane & 31;
  if (LOSS && sums) sums += (size_t)(blockIdx.x % replicas) * gridDim.y * Cout * 6;      // this block's copy of sums[B][Cout][6]
  const int n = blockIdx.y;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  bf16x8 whi[2], wlo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = pl < Cout ? w[pl * 32 + ks * 16 + half * 8 + j] : 0.f;
      const __bf16 h = (__bf16)v;
      whi[ks][j] = h;
      wlo[ks][j] = (__bf16)(v - (float)h);
    }
  float bias[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) bias[c] = ((EXACT || c < Cout) && b) ? b[c] : 0.f;
  float ts[CO][6], mt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) ts[c][k] = 0.f;
  const long long blk0 = (long long)blockIdx.x * pix_per_block;
  long long blk_end = blk0 + pix_per_block; if (blk_end > HW) blk_end = HW;
  const unsigned char* xn = x + (size_t)n * HW * 64;
  const size_t row0 = (size_t)n * HW;
  // B fragments of a pass: [pixel group][k-step]; labels of this lane's pixel
  auto fetch = [&](long long g0, uint4 (&q)[2][2], float (&yy)[CO]) {
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
      const long long pb = g0 + 32 * pg + pl;
      const bool ok = pb < blk_end;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) q[pg][ks] = ok ? ldg16(xn + (size_t)pb * 64 + ks * 32 + half * 16) : make_uint4(0, 0, 0, 0);
    }
    if constexpr (LOSS) {
      const long long pm = g0 + lane;
      const bool ok = pm < blk_end;
      const float* yr = y + (row0 + (size_t)pm) * Cout;
      if (CO == 6) {                        // 24-byte rows: three 8-byte loads
#pragma unroll
        for (int c = 0; c < 6; c += 2) { const float2 t = ok ? *reinterpret_cast<const float2*>(yr + c) : make_float2(0.f, 0.f); yy[c] = t.x; yy[c + 1] = t.y; }
      } else {
#pragma unroll
        for (int c = 0; c < CO; ++c) yy[c] = (ok && (EXACT || c < Cout)) ? yr[c] : 0.f;
      }
    }
  };
  uint4 xq[2][2], xq_n[2][2];
  float yv[CO], yv_n[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) { yv[c] = 0.f; yv_n[c] = 0.f; }
  long long g0 = blk0 + (long long)wid * 64;
  if (g0 < blk_end) fetch(g0, xq, yv);
  for (; g0 < blk_end; g0 += 256) {
    if (g0 + 256 < blk_end) fetch(g0 + 256, xq_n, yv_n);
    f32x16 d1, d2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { d1[i] = 0.f; d2[i] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 f1 = __builtin_bit_cast(bf16x8, xq[0][ks]), f2 = __builtin_bit_cast(bf16x8, xq[1][ks]);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], f1, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], f2, d2, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ks], f1, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ks], f2, d2, 0, 0, 0);
    }
    // d1[r] / d2[r], r < 4: class r + 4 * half of pixel pl (first / second group).  v_permlane32_swap(A, B) exchanges the upper half of A with the
    // lower half of B: afterwards lo[r] = class r and hi[r] = class 4 + r of pixel `lane` of the pass, in every lane.  (Inline asm as in conv_pw: the
    // builtin form was miscompiled; the s_nops cover the wait states between an MFMA - or the VALU copy out of an accumulator register - and the
    // exchange that reads its destination, which hipcc does not track for inline asm: without them one class of one pixel came out stale.)
    float lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = d1[r], bb = d2[r];
      if (r == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bb));
      else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bb));     // (s_nop: hipcc put the v_accvgpr_read of the operands right in front of the exchange)
      lo[r] = a; hi[r] = bb;
    }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = (c < 4 ? lo[c & 3] : hi[c & 3]) + bias[c];
    float pr[CO];
    if (act == RUA_ACT_SOFTMAX) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < CO; ++c) if (EXACT || c < Cout) mx = fmaxf(mx, acc[c]);
      float s_ = 0.f;
#pragma unroll
      for (int c = 0; c < CO; ++c) { pr[c] = (EXACT || c < Cout) ? expf(acc[c] - mx) : 0.f; s_ += pr[c]; }
      const float inv = 1.f / s_;
#pragma unroll
      for (int c = 0; c < CO; ++c) pr[c] *= inv;
    } else if (act == RUA_ACT_SIGMOID) {
#pragma unroll
      for (int c = 0; c < CO; ++c) pr[c] = 1.f / (1.f + expf(-acc[c]));
    } else {
#pragma unroll
      for (int c = 0; c < CO; ++c) pr[c] = acc[c];
    }
    const long long pm = g0 + lane;
    if (pm < blk_end) {
      const size_t m = row0 + (size_t)pm;
      if (CO == 6) {
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          if (z) *reinterpret_cast<float2*>(z + m * 6 + c) = make_float2(acc[c], acc[c + 1]);
          *reinterpret_cast<float2*>(p + m * 6 + c) = make_float2(pr[c], pr[c + 1]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < CO; ++c) if (EXACT || c < Cout) { if (z) z[m * Cout + c] = acc[c]; p[m * Cout + c] = pr[c]; }
      }
      if constexpr (LOSS) {
        int ip = 0, iy = 0; float bp = pr[0], by = yv[0];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
          if (EXACT || c < Cout) {
            const float a = pr[c], l = yv[c], q = 1.f - a, mm = 1.f - l;
            ts[c][0] += a; ts[c][1] += mm; ts[c][2] = fmaf(a, l, ts[c][2]);
            ts[c][3] += a * a + l * l; ts[c][4] = fmaf(q, mm, ts[c][4]); ts[c][5] += q * q + mm * mm;
            if (metrics) {
              if (a > bp) { bp = a; ip = c; }
              if (l > by) { by = l; iy = c; }
              const bool t = l > 0.5f, qq = a > 0.5f;
              mt[1] += (t && qq); mt[2] += (!t && qq); mt[3] += (!t && !qq); mt[4] += (t && !qq);
            }
          }
        }
        if (metrics) mt[0] += (ip == iy);
      }
    }
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xq[pg][ks] = xq_n[pg][ks];
#pragma unroll
    for (int c = 0; c < CO; ++c) yv[c] = yv_n[c];
  }
  if constexpr (LOSS) {
    // the wave's CO * 6 + 5 per-lane partial sums: through an LDS tile [lane][column] (odd row stride: the row writes and the column reads are
    // conflict-free), column j summed by lane j - 41 writes + 64 reads per wave.  (41 butterfly sums, 6 ds_bpermute each, were what a block cost at
    // its end: ~3.7 us of the CU's LDS pipe per block, the launch got SLOWER with more blocks per CU.)
    constexpr int NCOL = CO * 6 + 5, RS = NCOL | 1;
    __shared__ float red[4][64 * RS];
    float* rw = red[wid];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
      for (int k = 0; k < 6; ++k) rw[lane * RS + c * 6 + k] = ts[c][k];
#pragma unroll
    for (int k = 0; k < 5; ++k) rw[lane * RS + CO * 6 + k] = mt[k];
    __syncthreads();
    if (lane < NCOL) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int r = 0; r < 64; r += 4) { t0 += rw[r * RS + lane]; t1 += rw[(r + 1) * RS + lane]; t2 += rw[(r + 2) * RS + lane]; t3 += rw[(r + 3) * RS + lane]; }
      sh[wid * 56 + (lane < CO * 6 ? lane : 48 + lane - CO * 6)] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    if (threadIdx.x < 53) {
      const int e = threadIdx.x;
      if (e >= 48 || e < CO * 6) {
        const double t = (double)sh[e] + (double)sh[56 + e] + (double)sh[112 + e] + (double)sh[168 + e];
        if (e < 48) { const int c = e / 6, k = e % 6; if (c < Cout && sums) unsafeAtomicAdd(&sums[((size_t)n * Cout + c) * 6 + k], t); }
        else if (metrics) unsafeAtomicAdd(&metrics[e - 48], t);
      }
    }
  }
}

template <bool LOSS>
static void launch_head_fwd3(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* sums, double* metrics,
                             int B, int64_t HW, int Cout, int act, int replicas, hipStream_t st) {
  const int bpc = g_tune.head_fwd3_bpc > 0 ? g_tune.head_fwd3_bpc : (LOSS ? 3 : 4);       // blocks per CU over the batch (LOSS: a block holds a 42 KB reduction tile and ends in <= 53 fp64 atomics)
  int64_t per_sample = ((int64_t)bpc * rua_cu_count() + B - 1) / B; if (per_sample < 1) per_sample = 1;
  int64_t ppb = (HW + per_sample - 1) / per_sample; if (ppb < 256) ppb = 256;
  ppb = (ppb + 255) / 256 * 256;
  const int gx = (int)((HW + ppb - 1) / ppb);
#define RUA_HEAD3(CO_) hipLaunchKernelGGL((head_fwd3_kernel<CO_, LOSS>), dim3(gx, B), dim3(256), 0, st, (const unsigned char*)x, w, b, z, p, y, sums, metrics, (long long)HW, (int)ppb, Cout, act, replicas)
  if (Cout == 6) RUA_HEAD3(6); else if (Cout == 3) RUA_HEAD3(3); else RUA_HEAD3(8);
#undef RUA_HEAD3
}

template <typename T, bool LOSS>
static void launch_head_fwd2(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* sums, double* metrics,
                             int B, int64_t HW, int Cin, int Cout, int act, hipStream_t st) {
  int64_t per_sample = (4 * (int64_t)rua_cu_count() + B - 1) / B; if (per_sample < 1) per_sample = 1;      // ~4 blocks per CU over the batch
  constexpr int PL = 256 / (32 / ET<T>::VEC);
  int64_t ppb = (HW + per_sample - 1) / per_sample; if (ppb < PL) ppb = PL;
  ppb = (ppb + PL - 1) / PL * PL;
  const int gx = (int)((HW + ppb - 1) / ppb);
#define RUA_HEAD2(CO_) hipLaunchKernelGGL((head_fwd2_kernel<T, CO_, LOSS>), dim3(gx, B), dim3(256), 0, st, (const unsigned char*)x, w, b, z, p, y, sums, metrics, (long long)HW, (int)ppb, Cin, Cout, act)
  if (Cout == 6) RUA_HEAD2(6); else if (Cout == 3) RUA_HEAD2(3); else RUA_HEAD2(8);
#undef RUA_HEAD2
}

// head_fwd with the loss moments in its epilogue: the probabilities are in registers when the Tanimoto sums (and the accuracy /
// confusion counts of the 'seg' head) need them, so rua_tanimoto_sums / rua_seg_metrics do not read p again (four + one passes
// over 12.6 MB and five launches per step).  A block stays inside ONE sample (blockIdx.y): sums[n][c][6] as rua_tanimoto_sums
// defines them, metrics[5] as rua_seg_metrics.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_loss_kernel(const unsigned char* x, const float* __restrict__ w, const float* __restrict__ b,
                                                             float* z, float* p, const float* __restrict__ y, double* sums, double* metrics,
                                                             long long HW, int pix_per_block, int Cin, int Cout, int act) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float sw[];                       // [Cout][Cin] + [Cout], then [4 waves][53]
  float* sh = sw + Cout * Cin + Cout;
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) sw[Cout * Cin + i] = b ? b[i] : 0.f;
  __syncthreads();
  const int CGI = Cin / VEC;
  const int n = blockIdx.y;
  float ts[8][6], mt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) ts[c][k] = 0.f;
  long long i = (long long)blockIdx.x * pix_per_block + threadIdx.x;
  long long iend = (long long)(blockIdx.x + 1) * pix_per_block; if (iend > HW) iend = HW;
  for (; i < iend; i += 256) {
    const long long m = (long long)n * HW + i;
    float yv[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) yv[co] = co < Cout ? y[m * Cout + co] : 0.f;
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = co < Cout ? sw[Cout * Cin + co] : 0.f;
    for (int cp = 0; cp < CGI; ++cp) {
      float xv[VEC];
      ET<T>::unpack(ldg16(x + ((size_t)m * CGI + cp) * 16), xv);
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        if (co < Cout) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[co] = fmaf(xv[j], sw[co * Cin + cp * VEC + j], acc[co]);
        }
      }
    }
    float pr[8];
    if (act == RUA_ACT_SOFTMAX) {
      float mx = -INFINITY;
#pragma unroll
      for (int co = 0; co < 8; ++co) if (co < Cout) mx = fmaxf(mx, acc[co]);
      float s_ = 0.f;
#pragma unroll
      for (int co = 0; co < 8; ++co) { pr[co] = co < Cout ? expf(acc[co] - mx) : 0.f; s_ += pr[co]; }
      const float inv = 1.f / s_;
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] *= inv;
    } else if (act == RUA_ACT_SIGMOID) {
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] = 1.f / (1.f + expf(-acc[co]));
    } else {
#pragma unroll
      for (int co = 0; co < 8; ++co) pr[co] = acc[co];
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) if (co < Cout) { if (z) z[m * Cout + co] = acc[co]; p[m * Cout + co] = pr[co]; }
    // moments (rua_tanimoto_sums) and counts (rua_seg_metrics) of this pixel
    int ip = 0, iy = 0; float bp = pr[0], by = yv[0];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < Cout) {
        const float a = pr[c], l = yv[c], q = 1.f - a, mm = 1.f - l;
        ts[c][0] += a; ts[c][1] += mm; ts[c][2] = fmaf(a, l, ts[c][2]);
        ts[c][3] += a * a + l * l; ts[c][4] = fmaf(q, mm, ts[c][4]); ts[c][5] += q * q + mm * mm;
        if (a > bp) { bp = a; ip = c; }
        if (l > by) { by = l; iy = c; }
        const bool t = l > 0.5f, qq = a > 0.5f;
        mt[1] += (t && qq); mt[2] += (!t && qq); mt[3] += (!t && !qq); mt[4] += (t && !qq);
      }
    }
    mt[0] += (ip == iy);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();                                    // (the weights in sw are dead; sh is behind them anyway)
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (c < Cout) { const float v = wave_sum(ts[c][k]); if (lane == 0) sh[wid * 53 + c * 6 + k] = v; }
    }
  if (metrics) {
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = wave_sum(mt[k]); if (lane == 0) sh[wid * 53 + 48 + k] = v; }
  }
  __syncthreads();
  if (threadIdx.x < 53) {
    const int e = threadIdx.x;
    const double t = (double)sh[e] + (double)sh[53 + e] + (double)sh[106 + e] + (double)sh[159 + e];
    if (e < 48) { const int c = e / 6, k = e % 6; if (c < Cout && sums) unsafeAtomicAdd(&sums[((size_t)n * Cout + c) * 6 + k], t); }
    else if (metrics) unsafeAtomicAdd(&metrics[e - 48], t);
  }
}

template <typename T, int CO>      // CO: output channels held in registers (exact for the reference's 6 classes / 3 colour bands, else 8)
__global__ __launch_bounds__(256) void head_bwd_kernel(const unsigned char* x, const float* __restrict__ dz, const float* __restrict__ w,
                                                        unsigned char* dx, int accumulate_dx, float* dw, float* db, float* partial,
                                                        long long M, int Cin, int Cout, int rows_per_block, int mask_dx, float* dxsum) {
  constexpr int VEC = ET<T>::VEC;
  constexpr bool EXACT = CO != 8;                     // the launcher picks CO = 6 / 3 only for Cout == CO: no per-channel `co < Cout` tests (runtime tests became a branch around
                                                      // every one of the six dz loads of a row)
  extern __shared__ float red[];                      // [4 waves][Cout*Cin + Cout (+ Cin: per-channel sums of dx)]
  const int NE = Cout * Cin + Cout + (dxsum ? Cin : 0);
  const int CGI = Cin / VEC;
  const int cp = threadIdx.x % CGI, pl = threadIdx.x / CGI, PL = 256 / CGI;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float wr[CO][VEC], acc[CO][VEC], bs[CO], so[VEC];   // so: sums of the dx values this thread writes (dxsum: the bias gradient of the conv that produced x)
#pragma unroll
  for (int j = 0; j < VEC; ++j) so[j] = 0.f;
#pragma unroll
  for (int co = 0; co < CO; ++co) { bs[co] = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { wr[co][j] = (EXACT || co < Cout) ? w[co * Cin + cp * VEC + j] : 0.f; acc[co][j] = 0.f; } }
  long long r = (long long)blockIdx.x * rows_per_block + pl;
  long long rend = (long long)(blockIdx.x + 1) * rows_per_block; if (rend > M) rend = M;
  auto row = [&](const uint4& xq, const float* g, const uint4& oq, long long rr) {
    float xv[VEC], o[VEC];
    ET<T>::unpack(xq, xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
    if (dx && accumulate_dx) ET<T>::unpack(oq, o);
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      bs[co] += g[co];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { o[j] = fmaf(g[co], wr[co][j], o[j]); acc[co][j] = fmaf(g[co], xv[j], acc[co][j]); }
    }
    if (mask_dx) {                                      // x is the output of a fused ReLU: dx *= (x > 0)
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = xv[j] > 0.f ? o[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) so[j] += o[j];
    if (dx) stg16(dx + ((size_t)rr * CGI + cp) * 16, ET<T>::pack(o));
  };
  // the Cout gradients of row rr (CO = 6: a 24-byte row as three 8-byte loads)
  auto load_g = [&](long long rr, float* g) {
    if constexpr (CO == 6) {
      const float2* q = reinterpret_cast<const float2*>(dz + rr * 6);
      const float2 a = q[0], b2 = q[1], c2 = q[2];
      g[0] = a.x; g[1] = a.y; g[2] = b2.x; g[3] = b2.y; g[4] = c2.x; g[5] = c2.y;
    } else {
#pragma unroll
      for (int co = 0; co < CO; ++co) g[co] = (EXACT || co < Cout) ? dz[rr * Cout + co] : 0.f;
    }
  };
  // two rows per iteration, software-pipelined: the loads of the NEXT pair are issued before the arithmetic of this one (two blocks per
  // CU are resident: with one pair in flight a CU had 16 KB of x outstanding, a third of what the HBM latency asks for; four rows
  // loaded at once cost the second block its registers).  A thread still meets its rows in the same order (same sums, bit for bit).
  if (r + PL < rend) {
    uint4 xa = ldg16(x + ((size_t)r * CGI + cp) * 16), xb = ldg16(x + ((size_t)(r + PL) * CGI + cp) * 16);
    float ga[CO], gb[CO];
    load_g(r, ga); load_g(r + PL, gb);
    uint4 oa = make_uint4(0, 0, 0, 0), ob = oa;
    if (dx && accumulate_dx) { oa = ldg16(dx + ((size_t)r * CGI + cp) * 16); ob = ldg16(dx + ((size_t)(r + PL) * CGI + cp) * 16); }
    for (; r + PL < rend; r += 2 * PL) {
      const long long n0 = r + 2 * PL, n1 = r + 3 * PL;
      const bool more = n1 < rend;
      uint4 xc = make_uint4(0, 0, 0, 0), xd = xc, oc = xc, od = xc;
      float gc[CO], gd[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) { gc[co] = 0.f; gd[co] = 0.f; }
      if (more) {
        xc = ldg16(x + ((size_t)n0 * CGI + cp) * 16); xd = ldg16(x + ((size_t)n1 * CGI + cp) * 16);
        load_g(n0, gc); load_g(n1, gd);
        if (dx && accumulate_dx) { oc = ldg16(dx + ((size_t)n0 * CGI + cp) * 16); od = ldg16(dx + ((size_t)n1 * CGI + cp) * 16); }
      }
      row(xa, ga, oa, r);
      row(xb, gb, ob, r + PL);
      xa = xc; xb = xd; oa = oc; ob = od;
#pragma unroll
      for (int co = 0; co < CO; ++co) { ga[co] = gc[co]; gb[co] = gd[co]; }
    }
  }
  for (; r < rend; r += PL) {
    const uint4 x0 = ldg16(x + ((size_t)r * CGI + cp) * 16);
    float g0[CO];
    load_g(r, g0);
    uint4 o0 = make_uint4(0, 0, 0, 0);
    if (dx && accumulate_dx) o0 = ldg16(dx + ((size_t)r * CGI + cp) * 16);
    row(x0, g0, o0, r);
  }
  // The block's sums: every thread's CO * VEC + CO + VEC partial sums go into an LDS tile [thread][value] (odd row stride: conflict-free row
  // writes and column reads), output element e is the sum of one column over the PL threads that share its channel piece - all reads
  // independent.  (Before: 62 butterflies of 4 dependent ds_bpermute each per wave - latency-bound, ~8 us of the CU's time per block: the
  // launch took 33 us with 512 blocks and 50 us with 1024.)
  constexpr int NV = CO * VEC + CO + VEC, RS = NV | 1;
  (void)lane; (void)wid;
  float* row_ = red + (size_t)threadIdx.x * RS;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) row_[co * VEC + j] = acc[co][j];
    row_[CO * VEC + co] = bs[co];
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) row_[CO * VEC + CO + j] = so[j];
  __syncthreads();
  for (int i = threadIdx.x; i < NE; i += 256) {
    int col, cpi;                                       // column of the tile, channel piece (the threads t = pl * CGI + cpi hold it)
    if (i < Cout * Cin) { const int co = i / Cin, ci = i - co * Cin; cpi = ci / VEC; col = co * VEC + (ci - cpi * VEC); }
    else if (i < Cout * Cin + Cout) { cpi = 0; col = CO * VEC + (i - Cout * Cin); }
    else { const int ci = i - Cout * Cin - Cout; cpi = ci / VEC; col = CO * VEC + CO + (ci - cpi * VEC); }
    const float* cp_ = red + (size_t)cpi * RS + col;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int q = 0; q < PL; q += 4) {
      t0 += cp_[(size_t)(q + 0) * CGI * RS]; t1 += cp_[(size_t)(q + 1) * CGI * RS];
      t2 += cp_[(size_t)(q + 2) * CGI * RS]; t3 += cp_[(size_t)(q + 3) * CGI * RS];
    }
    const float v = (t0 + t1) + (t2 + t3);
    if (partial) partial[(size_t)blockIdx.x * NE + i] = v;                       // deterministic two-stage path
    else if (i < Cout * Cin) unsafeAtomicAdd(&dw[i], v);
    else if (i < Cout * Cin + Cout) { if (db) unsafeAtomicAdd(&db[i - Cout * Cin], v); }
    else unsafeAtomicAdd(&dxsum[i - Cout * Cin - Cout], v);
  }
}

// dst[e] += sum_b partial[b][e]; 4 elements x 64 slices per block (a thread's loads are independent: unrolled by 8), slices
// folded in a fixed order
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int ne, int nparts, float* dw, int ndw, float* db,
                                                             int ndb = 1 << 30, float* dsum = nullptr) {
  __shared__ float sh[256];
  const int el = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int e = blockIdx.x * 4 + el;
  float s = 0.f;
  if (e < ne) {
    int b = sl;
    for (; b + 7 * 64 < nparts; b += 8 * 64) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(b + k * 64) * ne + e];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; b < nparts; b += 64) s += partial[(size_t)b * ne + e];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && e < ne) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) t += sh[k * 4 + el];
    if (e < ndw) dw[e] += t;
    else if (e < ndw + ndb) { if (db) db[e - ndw] += t; }
    else if (dsum) dsum[e - ndw - ndb] += t;
  }
}

extern "C" int rua_head_fwd(const void* x, const float* w, const float* b, float* z, float* p, int64_t M, int Cin, int Cout, int act, int dtype, void* stream) {
  RUA_CHECK_ARG(x && w && p && M > 0, "rua_head_fwd: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(Cin % vec == 0 && Cin <= 256, "rua_head_fwd: Cin=%d must be a multiple of %d (<=256)", Cin, vec);
  RUA_CHECK_ARG(Cout >= 1 && Cout <= 8, "rua_head_fwd: Cout=%d must be in 1..8", Cout);
  const size_t smem = (size_t)(Cout * Cin + Cout) * 4;
  int64_t g = (M + 255) / 256; if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 32 && dtype == RUA_BF16 && g_tune.head_fwd3 && (M * 64) < (1ll << 40)) {     // bf16 storage: the product on the matrix pipe, a lane per pixel
    launch_head_fwd3<false>(x, w, b, z, p, nullptr, nullptr, nullptr, 1, M, Cout, act, 1, st);
    RUA_LAUNCH_CHECK("rua_head_fwd");
    return RUA_OK;
  }
  if (Cin == 32 && g_tune.head_fwd2) {                  // the reference's heads: weights in registers, a pixel shared by Cin / VEC lanes
    if (dtype == RUA_BF16) launch_head_fwd2<bf16_t, false>(x, w, b, z, p, nullptr, nullptr, nullptr, 1, M, Cin, Cout, act, st);
    else launch_head_fwd2<float, false>(x, w, b, z, p, nullptr, nullptr, nullptr, 1, M, Cin, Cout, act, st);
    RUA_LAUNCH_CHECK("rua_head_fwd");
    return RUA_OK;
  }
  if (dtype == RUA_BF16) hipLaunchKernelGGL((head_fwd_kernel<bf16_t>), dim3((int)g), dim3(256), smem, st, (const unsigned char*)x, w, b, z, p, (long long)M, Cin, Cout, act);
  else hipLaunchKernelGGL((head_fwd_kernel<float>), dim3((int)g), dim3(256), smem, st, (const unsigned char*)x, w, b, z, p, (long long)M, Cin, Cout, act);
  RUA_LAUNCH_CHECK("rua_head_fwd");
  return RUA_OK;
}

extern "C" int rua_head_fwd_loss_rep(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* tanimoto_sums,
                                     int sums_replicas, double* metrics, int B, int64_t HW, int Cin, int Cout, int act, int dtype, void* stream);
extern "C" int rua_head_fwd_loss(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* tanimoto_sums,
                                 double* metrics, int B, int64_t HW, int Cin, int Cout, int act, int dtype, void* stream) {
  return rua_head_fwd_loss_rep(x, w, b, z, p, y, tanimoto_sums, 1, metrics, B, HW, Cin, Cout, act, dtype, stream);
}
// tanimoto_sums [sums_replicas][B][Cout][6]: a block adds into copy blockIdx.x % sums_replicas (every block ends in <= 48 fp64 atomics on its sample's
// sums: with one copy the ~64 blocks of a sample queue up on each of them - 11 us of a 24 us launch); rua_tanimoto_finalize_rep adds the copies
extern "C" int rua_head_fwd_loss_rep(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* tanimoto_sums,
                                     int sums_replicas, double* metrics, int B, int64_t HW, int Cin, int Cout, int act, int dtype, void* stream) {
  RUA_CHECK_ARG(sums_replicas >= 1 && sums_replicas <= 64, "rua_head_fwd_loss_rep: sums_replicas=%d must be in 1..64", sums_replicas);
  RUA_CHECK_ARG(x && w && p && y && B > 0 && HW > 0, "rua_head_fwd_loss: bad arguments");
  RUA_CHECK_ARG(tanimoto_sums || metrics, "rua_head_fwd_loss: nothing to accumulate (use rua_head_fwd)");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(Cin % vec == 0 && Cin <= 256, "rua_head_fwd_loss: Cin=%d must be a multiple of %d (<=256)", Cin, vec);
  RUA_CHECK_ARG(Cout >= 1 && Cout <= 8, "rua_head_fwd_loss: Cout=%d must be in 1..8", Cout);
  const size_t smem = (size_t)(Cout * Cin + Cout + 4 * 53) * 4;
  // ~2 blocks per CU over the batch; every block ends in <= 53 fp64 atomics on its sample's sums
  int64_t per_sample = (2 * (int64_t)rua_cu_count() + B - 1) / B; if (per_sample < 1) per_sample = 1;
  int64_t ppb = (HW + per_sample - 1) / per_sample; if (ppb < 256) ppb = 256;
  ppb = (ppb + 255) / 256 * 256;
  const int gx = (int)((HW + ppb - 1) / ppb);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 32 && dtype == RUA_BF16 && g_tune.head_fwd3) {
    launch_head_fwd3<true>(x, w, b, z, p, y, tanimoto_sums, metrics, B, HW, Cout, act, sums_replicas, st);
    RUA_LAUNCH_CHECK("rua_head_fwd_loss");
    return RUA_OK;
  }
  if (Cin == 32 && g_tune.head_fwd2) {
    if (dtype == RUA_BF16) launch_head_fwd2<bf16_t, true>(x, w, b, z, p, y, tanimoto_sums, metrics, B, HW, Cin, Cout, act, st);
    else launch_head_fwd2<float, true>(x, w, b, z, p, y, tanimoto_sums, metrics, B, HW, Cin, Cout, act, st);
    RUA_LAUNCH_CHECK("rua_head_fwd_loss");
    return RUA_OK;
  }
  if (dtype == RUA_BF16) hipLaunchKernelGGL((head_fwd_loss_kernel<bf16_t>), dim3(gx, B), dim3(256), smem, st, (const unsigned char*)x, w, b, z, p, y, tanimoto_sums, metrics, (long long)HW, (int)ppb, Cin, Cout, act);
  else hipLaunchKernelGGL((head_fwd_loss_kernel<float>), dim3(gx, B), dim3(256), smem, st, (const unsigned char*)x, w, b, z, p, y, tanimoto_sums, metrics, (long long)HW, (int)ppb, Cin, Cout, act);
  RUA_LAUNCH_CHECK("rua_head_fwd_loss");
  return RUA_OK;
}

extern "C" int rua_head_bwd_sums(const void* x, const float* dz, const float* w, void* dx, int accumulate_dx, float* dw, float* db, float* dxsum,
                                 float* scratch, int64_t scratch_bytes, int64_t M, int Cin, int Cout, int dtype, int mask_dx, void* stream) {
  RUA_CHECK_ARG(x && dz && w && dw && M > 0, "rua_head_bwd: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(Cin % vec == 0 && Cin <= 256 && 256 % (Cin / vec) == 0, "rua_head_bwd: unsupported Cin=%d", Cin);
  RUA_CHECK_ARG(Cout >= 1 && Cout <= 8, "rua_head_bwd: Cout=%d must be in 1..8", Cout);
  const int blocks_env = g_tune.head_blocks;
  // two blocks per CU are resident (VGPRs), so 512 blocks run in one round: measured 128 blocks 49 us, 256: 37, 512: 35,
  // 640: 52, 1024: 51, 2048: 85 (kernel + partial reduce, 256x256x32 -> 6)
  int64_t blocks = blocks_env > 0 ? blocks_env : 2 * rua_cu_count(); int64_t rpb = (M + blocks - 1) / blocks; if (rpb < 64) rpb = 64;
  const int g = (int)((M + rpb - 1) / rpb);
  RUA_CHECK_ARG(!dxsum || dx, "rua_head_bwd_sums: the sums are those of dx");
  const int ne = Cout * Cin + Cout + (dxsum ? Cin : 0);
  const int co_t = Cout == 6 ? 6 : (Cout == 3 ? 3 : 8);
  const size_t smem = (size_t)256 * ((co_t * vec + co_t + vec) | 1) * 4;      // the block's [thread][value] reduction tile
  float* partial = (scratch && scratch_bytes >= (int64_t)g * ne * 4) ? scratch : nullptr;   // else: fp32 atomics
  hipStream_t st = (hipStream_t)stream;
#define RUA_HEAD_BWD(TT, CO_) do { static RuaPerDevFlag af; bool& a_ = af.get(); \
    if (!a_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<TT, CO_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); a_ = true; } \
    hipLaunchKernelGGL((head_bwd_kernel<TT, CO_>), dim3(g), dim3(256), smem, st, (const unsigned char*)x, dz, w, \
    (unsigned char*)dx, accumulate_dx, dw, db, partial, (long long)M, Cin, Cout, (int)rpb, mask_dx, dxsum); } while (0)
  if (dtype == RUA_BF16) { if (Cout == 6) RUA_HEAD_BWD(bf16_t, 6); else if (Cout == 3) RUA_HEAD_BWD(bf16_t, 3); else RUA_HEAD_BWD(bf16_t, 8); }
  else { if (Cout == 6) RUA_HEAD_BWD(float, 6); else if (Cout == 3) RUA_HEAD_BWD(float, 3); else RUA_HEAD_BWD(float, 8); }
#undef RUA_HEAD_BWD
  RUA_LAUNCH_CHECK("rua_head_bwd");
  if (partial) {
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((ne + 3) / 4), dim3(256), 0, st, (const float*)partial, ne, g, dw, Cout * Cin, db, Cout, dxsum);
    RUA_LAUNCH_CHECK("partial_reduce_kernel");
  }
  return RUA_OK;
}

// dxsum (optional, fp32 [Cin], +=): the per-channel sums of the values written to dx - the bias gradient of the convolution that produced x when dx is
// its output's complete gradient (the 3x3 + ReLU convs in front of the heads, model2.py:153-171) - taken in the same pass
extern "C" int rua_head_bwd(const void* x, const float* dz, const float* w, void* dx, int accumulate_dx, float* dw, float* db,
                            float* scratch, int64_t scratch_bytes, int64_t M, int Cin, int Cout, int dtype, int mask_dx, void* stream) {
  return rua_head_bwd_sums(x, dz, w, dx, accumulate_dx, dw, db, nullptr, scratch, scratch_bytes, M, Cin, Cout, dtype, mask_dx, stream);
}

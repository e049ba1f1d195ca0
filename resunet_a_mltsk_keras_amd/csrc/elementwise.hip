// HBM-bound kernels: BatchNorm statistics / apply / backward, pooling, resampling, n-ary add.
// All tensors are channels-last [M][C]; every thread moves 16-byte pieces (8 bf16 / 4 fp32
// channels), per-channel coefficient tables are staged in LDS once per block.
#include "common.h"
#include <stdlib.h>

static inline int grid_for(int64_t pieces) {
  int64_t g = (pieces + 255) / 256;
  if (g > 2048) g = 2048;          // 256 CUs x 8 blocks; grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------
// per-channel statistics
struct StatsK { const unsigned char* x; const unsigned char* g; const float* ms; const float* mt; int masked;
                long long M; int C, CG, TX, TY, rows_per_block; double* stats; int R; };

template <typename T, int MODE>   // MODE 1: sum x, sum x^2 ; MODE 2: sum g*m, sum g*m*x
__global__ __launch_bounds__(256) void col_stats_kernel(const StatsK p) {
  // CG (16-byte pieces per row) is a power of two <= 256, so thread t of every block owns column piece t % CG for the whole
  // grid-stride sweep: partial sums stay in registers, 8 independent 16-byte loads in flight per thread, lanes that share
  // a piece are folded with wave shuffles, the four waves through LDS, and ONE fp64 atomic per channel and block goes to
  // replica (block % R).  Same-address fp64 atomics serialise at ~180 ns each (measured), so the launcher keeps the grid
  // at <= 512 blocks and the caller sizes R for <= ~16 adds per address.
  constexpr int VEC = ET<T>::VEC;
  __shared__ float sred[256 * (2 * VEC + 1)];
  const int tid = threadIdx.x;
  const int CGA = p.CG;                                  // pieces per row (power of two)
  const int CG = CGA < 256 ? CGA : 256;                  // pieces handled by one block (blockIdx.y picks the 256-piece slab)
  const int cp0 = blockIdx.y * CG;
  const int cp = cp0 + (tid & (CG - 1));
  const int rpb = 256 / CG;                              // rows covered by one block per step
  float s1[VEC], s2[VEC], ms[VEC], mt[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; ms[j] = 1.f; mt[j] = 0.f; }
  if (MODE == 2 && p.masked) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { if (p.ms) ms[j] = p.ms[cp * VEC + j]; if (p.mt) mt[j] = p.mt[cp * VEC + j]; }
  }
  const long long rstep = (long long)gridDim.x * rpb;
  long long r = (long long)blockIdx.x * rpb + tid / CG;
  constexpr int UN = (MODE == 1) ? 8 : 4;
  for (; r + (UN - 1) * rstep < p.M; r += UN * rstep) {
    uint4 q[UN], qg[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      q[u] = ldg16(p.x + ((size_t)(r + u * rstep) * CGA + cp) * 16);
      if (MODE == 2) qg[u] = ldg16(p.g + ((size_t)(r + u * rstep) * CGA + cp) * 16);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      float xv[VEC];
      ET<T>::unpack(q[u], xv);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s1[j] += xv[j]; s2[j] = fmaf(xv[j], xv[j], s2[j]); }
      } else {
        float gv[VEC];
        ET<T>::unpack(qg[u], gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float m = (!p.masked || fmaf(ms[j], xv[j], mt[j]) > 0.f) ? gv[j] : 0.f;
          s1[j] += m; s2[j] = fmaf(m, xv[j], s2[j]);
        }
      }
    }
  }
  for (; r < p.M; r += rstep) {
    float xv[VEC];
    ET<T>::unpack(ldg16(p.x + ((size_t)r * CGA + cp) * 16), xv);
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s1[j] += xv[j]; s2[j] = fmaf(xv[j], xv[j], s2[j]); }
    } else {
      float gv[VEC];
      ET<T>::unpack(ldg16(p.g + ((size_t)r * CGA + cp) * 16), gv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float m = (!p.masked || fmaf(ms[j], xv[j], mt[j]) > 0.f) ? gv[j] : 0.f;
        s1[j] += m; s2[j] = fmaf(m, xv[j], s2[j]);
      }
    }
  }
  // The 256 / CG threads that share a piece meet in an LDS tile [thread][2 VEC + 1] and ONE thread per (piece, value) sums its column with independent reads.
  // (Until round 5: wave shuffles first - at C = 32 64 dependent ds_bpermute per wave at the end of every block, the LDS-pipe tail of the round-4 heads:
  // rua_col_stats2 on 2 x 33.5 MB took 28.6 us.)
  constexpr int TW = 2 * VEC + 1;
  float* mine = sred + tid * TW;
#pragma unroll
  for (int j = 0; j < VEC; ++j) { mine[j] = s1[j]; mine[VEC + j] = s2[j]; }
  __syncthreads();
  double* st = p.stats + (size_t)(blockIdx.x & (p.R - 1)) * 2 * p.C;
  for (int t = tid; t < CG * 2 * VEC; t += 256) {
    const int cpi = t / (2 * VEC), k = t % (2 * VEC);     // piece, (which sum, channel in piece)
    float a0 = 0.f, a1 = 0.f;
    int rr = cpi;
    for (; rr + CG < 256; rr += 2 * CG) { a0 += sred[rr * TW + k]; a1 += sred[(rr + CG) * TW + k]; }
    if (rr < 256) a0 += sred[rr * TW + k];
    unsafeAtomicAdd(&st[(k / VEC) * p.C + (cp0 + cpi) * VEC + (k % VEC)], (double)(a0 + a1));
  }
}

extern "C" int rua_stats_replicas(int64_t blocks) {
  int r = 1;
  while (r < 32 && (int64_t)r * 8 < blocks) r *= 2;      // <= ~8 adds per replica address up to the cap of 32 replicas:
  return r;                                              // same-address fp64 atomics serialise at ~180 ns each
}

template <int MODE>
static int launch_stats(const void* g, const void* x, const float* ms, const float* mt, int masked,
                        int64_t M, int C, double* stats, int replicas, int dtype, void* stream, const char* name) {
  RUA_CHECK_ARG(replicas >= 1 && (replicas & (replicas - 1)) == 0, "%s: replicas must be a power of two", name);
  RUA_CHECK_ARG(x && stats && M > 0, "%s: bad arguments", name);
  RUA_CHECK_ARG(dtype == RUA_F32 || dtype == RUA_BF16, "%s: bad dtype", name);
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(C % vec == 0, "%s: C=%d not a multiple of %d", name, C, vec);
  StatsK k;
  k.x = (const unsigned char*)x; k.g = (const unsigned char*)g; k.ms = ms; k.mt = mt; k.masked = masked;
  k.M = M; k.C = C; k.CG = C / vec;
  RUA_CHECK_ARG((k.CG & (k.CG - 1)) == 0, "%s: C/%d = %d must be a power of two", name, vec, k.CG);
  const int gy = k.CG > 256 ? k.CG / 256 : 1;
  k.TX = k.CG; k.TY = 256 / k.CG; k.rows_per_block = 0;
  k.stats = stats; k.R = replicas;
  const int64_t pieces = M * k.CG;
  int64_t gx = pieces / (256 * 8) / gy;                  // >= 8 pieces per thread
  const int64_t scap = g_tune.stats_blocks > 0 ? g_tune.stats_blocks : 1024;     // (512 until round 4: 1024 measured -0.03 ms per step, same-box A/B)
  if (gx > scap / gy) gx = scap / gy;
  if (gx < 1) gx = 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((col_stats_kernel<bf16_t, MODE>), dim3((unsigned)gx, gy), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((col_stats_kernel<float, MODE>), dim3((unsigned)gx, gy), dim3(256), 0, st, k);
  RUA_LAUNCH_CHECK(name);
  return RUA_OK;
}

extern "C" int rua_col_stats(const void* x, int64_t M, int C, double* stats, int replicas, int dtype, void* stream) {
  return launch_stats<1>(nullptr, x, nullptr, nullptr, 0, M, C, stats, replicas, dtype, stream, "rua_col_stats");
}
extern "C" int rua_col_stats2(const void* g, const void* x, const float* mscale, const float* mshift, int masked,
                              int64_t M, int C, double* stats, int replicas, int dtype, void* stream) {
  RUA_CHECK_ARG(g != nullptr, "rua_col_stats2: null g");
  return launch_stats<2>(g, x, mscale, mshift, masked, M, C, stats, replicas, dtype, stream, "rua_col_stats2");
}

// ---------------------------------------------------------------------------------------

__global__ void bn_finalize_kernel(const double* stats, int R, double count, double bessel_n, const float* gamma, const float* beta,
                                   float* mmean, float* mvar, float momentum, float eps, int training,
                                   float* scale, float* shift, float* mean_o, float* rstd_o, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (training) {
    double s1, s2;
    replica_sum(stats, R, C, c, s1, s2);
    mean = s1 / count;
    var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    if (mmean) {
      const double unb = bessel_n > 1 ? var * (bessel_n / (bessel_n - 1)) : var;
      mmean[c] = (float)((double)mmean[c] * momentum + mean * (1.0 - momentum));
      mvar[c] = (float)((double)mvar[c] * momentum + unb * (1.0 - momentum));
    }
  } else {
    mean = mmean[c]; var = mvar[c];
  }
  const double r = 1.0 / sqrt(var + (double)eps);
  const double s = (double)gamma[c] * r;
  scale[c] = (float)s;
  shift[c] = (float)((double)beta[c] - mean * s);
  if (mean_o) mean_o[c] = (float)mean;
  if (rstd_o) rstd_o[c] = (float)r;
}

extern "C" int rua_bn_finalize(const double* stats, int replicas, double count, double bessel_n, const float* gamma, const float* beta,
                               float* moving_mean, float* moving_var, float momentum, float eps, int training,
                               float* scale, float* shift, float* mean, float* rstd, int C, void* stream) {
  RUA_CHECK_ARG(gamma && beta && scale && shift && C > 0, "rua_bn_finalize: bad arguments");
  RUA_CHECK_ARG(training ? (stats != nullptr && count > 0) : (moving_mean && moving_var), "rua_bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(rua_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream,
                     stats, replicas < 1 ? 1 : replicas, count, bessel_n, gamma, beta, moving_mean, moving_var, momentum, eps, training, scale, shift, mean, rstd, C);
  RUA_LAUNCH_CHECK("rua_bn_finalize");
  return RUA_OK;
}

__global__ void bn_bwd_finalize_kernel(const double* st2, int R, double count, const float* gamma, const float* mean, const float* rstd,
                                       float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sg, sgx;
  replica_sum(st2, R, C, c, sg, sgx);
  const double mu = mean[c], r = rstd[c], gm = gamma[c];
  const double dbe = sg;
  const double dga = r * (sgx - mu * sg);
  const double s = gm * r;
  if (dgamma) dgamma[c] += (float)dga;      // gradients accumulate (the optimizer zeroes the flat buffer)
  if (dbeta) dbeta[c] += (float)dbe;
  cA[c] = (float)s;
  cB[c] = (float)(-s * r * dga / count);
  cC[c] = (float)(-s * dbe / count + s * r * mu * dga / count);
}

extern "C" int rua_bn_bwd_finalize(const double* stats2, int replicas, double count, const float* gamma, const float* mean, const float* rstd,
                                   float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC, int C, void* stream) {
  RUA_CHECK_ARG(stats2 && gamma && mean && rstd && coefA && coefB && coefC && C > 0 && count > 0, "rua_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(rua_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream,
                     stats2, replicas < 1 ? 1 : replicas, count, gamma, mean, rstd, dgamma, dbeta, coefA, coefB, coefC, C);
  RUA_LAUNCH_CHECK("rua_bn_bwd_finalize");
  return RUA_OK;
}

// per-channel sums (fp64) -> n fp32 destinations (bias gradients: d b = sum over pixels of dy)
__global__ void stats_to_f32_kernel(const double* stats, int R, int C, float* d0, float* d1, float* d2, float* d3, int n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a, unused;
  replica_sum(stats, R, C, c, a, unused);
  const float v = (float)a;
  d0[c] += v; if (n > 1) d1[c] += v; if (n > 2) d2[c] += v; if (n > 3) d3[c] += v;
}
extern "C" int rua_stats_to_f32(const double* stats, int replicas, int C, float* const* dst, int n, void* stream) {
  RUA_CHECK_ARG(stats && dst && n >= 1 && n <= 4 && C > 0, "rua_stats_to_f32: bad arguments");
  for (int i = 0; i < n; ++i) RUA_CHECK_ARG(dst[i], "rua_stats_to_f32: null destination");
  hipLaunchKernelGGL(stats_to_f32_kernel, dim3(rua_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream, stats, replicas < 1 ? 1 : replicas, C,
                     dst[0], n > 1 ? dst[1] : nullptr, n > 2 ? dst[2] : nullptr, n > 3 ? dst[3] : nullptr, n);
  RUA_LAUNCH_CHECK("rua_stats_to_f32");
  return RUA_OK;
}

// ---------------------------------------------------------------------------------------
struct ApplyK { const unsigned char* x; int nb; const float* scale[RUA_MAX_BRANCH]; const float* shift[RUA_MAX_BRANCH];
                unsigned char* out[RUA_MAX_BRANCH]; int relu; long long pieces; int C, CG; };

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const ApplyK p) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float tab[];                       // [nb][2][C]
  for (int i = threadIdx.x; i < p.nb * p.C; i += 256) {
    const int b = i / p.C, c = i - b * p.C;
    tab[(b * 2) * p.C + c] = p.scale[b][c];
    tab[(b * 2 + 1) * p.C + c] = p.shift[b][c];
  }
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.pieces; i += stride) {
    const int c = (int)(i % p.CG) * VEC;
    float xv[VEC];
    ET<T>::unpack(ldg16(p.x + i * 16), xv);
    for (int b = 0; b < p.nb; ++b) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        o[j] = fmaf(tab[(b * 2) * p.C + c + j], xv[j], tab[(b * 2 + 1) * p.C + c + j]);
        if (p.relu) o[j] = fmaxf(o[j], 0.f);
      }
      stg16(p.out[b] + i * 16, ET<T>::pack(o));
    }
  }
}

extern "C" int rua_bn_apply(const void* x, int nb, const float* const* scale, const float* const* shift, int relu,
                            void* const* out, int64_t M, int C, int dtype, void* stream) {
  RUA_CHECK_ARG(x && scale && shift && out && nb >= 1 && nb <= RUA_MAX_BRANCH && M > 0, "rua_bn_apply: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(C % vec == 0, "rua_bn_apply: C=%d not a multiple of %d", C, vec);
  ApplyK k;
  k.x = (const unsigned char*)x; k.nb = nb; k.relu = relu; k.C = C; k.CG = C / vec; k.pieces = M * k.CG;
  for (int b = 0; b < nb; ++b) { k.scale[b] = scale[b]; k.shift[b] = shift[b]; k.out[b] = (unsigned char*)out[b];
    RUA_CHECK_ARG(scale[b] && shift[b] && out[b], "rua_bn_apply: null branch pointer"); }
  const size_t smem = (size_t)nb * 2 * C * 4;
  RUA_CHECK_ARG(smem <= 64 * 1024, "rua_bn_apply: coefficient table too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), dim3(grid_for(k.pieces)), dim3(256), smem, st, k);
  else hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(grid_for(k.pieces)), dim3(256), smem, st, k);
  RUA_LAUNCH_CHECK("rua_bn_apply");
  return RUA_OK;
}

struct BwdApplyK { int nb; const unsigned char* g[RUA_MAX_BRANCH]; const float* cA[RUA_MAX_BRANCH]; const float* cB[RUA_MAX_BRANCH];
                   const float* cC[RUA_MAX_BRANCH]; const float* ms[RUA_MAX_BRANCH]; const float* mt[RUA_MAX_BRANCH]; int masked;
                   const unsigned char* x; const unsigned char* dskip; unsigned char* dx; int accumulate; long long pieces; int C, CG; };

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdApplyK p) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float tab[];                       // [nb][3][C] : A, ms, mt ; then [2][C] : sumB, sumC
  float* tB = tab + p.nb * 3 * p.C;
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float sb = 0.f, sc = 0.f;
    for (int b = 0; b < p.nb; ++b) {
      tab[(b * 3) * p.C + c] = p.cA[b][c];
      tab[(b * 3 + 1) * p.C + c] = (p.masked && p.ms[b]) ? p.ms[b][c] : 1.f;
      tab[(b * 3 + 2) * p.C + c] = (p.masked && p.mt[b]) ? p.mt[b][c] : 0.f;
      sb += p.cB[b][c]; sc += p.cC[b][c];
    }
    tB[c] = sb; tB[p.C + c] = sc;
  }
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.pieces; i += stride) {
    const int c = (int)(i % p.CG) * VEC;
    float xv[VEC], acc[VEC];
    ET<T>::unpack(ldg16(p.x + i * 16), xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(tB[c + j], xv[j], tB[p.C + c + j]);
    if (p.dskip) {
      float d[VEC];
      ET<T>::unpack(ldg16(p.dskip + i * 16), d);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += d[j];
    }
    if (p.accumulate) {
      float d[VEC];
      ET<T>::unpack(ldg16(p.dx + i * 16), d);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += d[j];
    }
    for (int b = 0; b < p.nb; ++b) {
      float gv[VEC];
      ET<T>::unpack(ldg16(p.g[b] + i * 16), gv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const bool on = !p.masked || fmaf(tab[(b * 3 + 1) * p.C + c + j], xv[j], tab[(b * 3 + 2) * p.C + c + j]) > 0.f;
        acc[j] = fmaf(tab[(b * 3) * p.C + c + j], on ? gv[j] : 0.f, acc[j]);
      }
    }
    stg16(p.dx + i * 16, ET<T>::pack(acc));
  }
}

extern "C" int rua_bn_bwd_apply(int nb, const void* const* g, const float* const* coefA, const float* const* coefB,
                                const float* const* coefC, const float* const* mscale, const float* const* mshift, int masked,
                                const void* x, const void* dskip, void* dx, int accumulate, int64_t M, int C, int dtype, void* stream) {
  RUA_CHECK_ARG(nb >= 1 && nb <= RUA_MAX_BRANCH && g && coefA && coefB && coefC && x && dx && M > 0, "rua_bn_bwd_apply: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(C % vec == 0, "rua_bn_bwd_apply: C=%d not a multiple of %d", C, vec);
  BwdApplyK k;
  k.nb = nb; k.masked = masked; k.x = (const unsigned char*)x; k.dskip = (const unsigned char*)dskip; k.dx = (unsigned char*)dx;
  k.accumulate = accumulate; k.C = C; k.CG = C / vec; k.pieces = M * k.CG;
  for (int b = 0; b < nb; ++b) {
    RUA_CHECK_ARG(g[b] && coefA[b] && coefB[b] && coefC[b], "rua_bn_bwd_apply: null branch pointer");
    k.g[b] = (const unsigned char*)g[b]; k.cA[b] = coefA[b]; k.cB[b] = coefB[b]; k.cC[b] = coefC[b];
    k.ms[b] = mscale ? mscale[b] : nullptr; k.mt[b] = mshift ? mshift[b] : nullptr;
  }
  const size_t smem = (size_t)(nb * 3 + 2) * C * 4;
  RUA_CHECK_ARG(smem <= 64 * 1024, "rua_bn_bwd_apply: coefficient table too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(grid_for(k.pieces)), dim3(256), smem, st, k);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(grid_for(k.pieces)), dim3(256), smem, st, k);
  RUA_LAUNCH_CHECK("rua_bn_bwd_apply");
  return RUA_OK;
}

// ---------------------------------------------------------------------------------------
// Fused BatchNorm: statistics -> coefficients in the prologue of every block (no separate finalize launch);
// block 0 also publishes the coefficients for later kernels (ReLU masks of data-gradient epilogues, backward) and
// updates the moving statistics / parameter gradients.
// NB > 0: the branch count, coefficients in registers (needs 256 % CG == 0); NB == 0: any shape, coefficients read from the LDS table
template <typename T, int NB>
__device__ __forceinline__ void bn_fwd_body(const rua_bn_fwd_desc& p, long long pieces, int CG) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float tab[];                       // [nb][2][C] scale/shift, then [2][C] mean/rstd
  const int C = p.C;
  float* mr = tab + p.nb * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    double mean, var;
    if (p.training && p.stats) {
      double s1, s2;
      replica_sum(p.stats, p.replicas, C, c, s1, s2);
      mean = s1 / p.count;
      var = s2 / p.count - mean * mean;
      if (var < 0) var = 0;
    } else { mean = 0; var = 1; }
    for (int b = 0; b < p.nb; ++b) {
      const rua_bn_branch& br = p.br[b];
      double m = mean, v = var;
      if (p.training && br.stats) {                    // this branch normalises its own tensor (coefficient-only launches)
        double s1, s2;
        replica_sum(br.stats, br.replicas, C, c, s1, s2);
        m = s1 / p.count;
        v = s2 / p.count - m * m;
        if (v < 0) v = 0;
      }
      if (!p.training) { m = br.moving_mean[c]; v = br.moving_var[c]; }
      const double r = 1.0 / sqrt(v + (double)p.eps);
      const double sc = (double)br.gamma[c] * r;
      tab[(b * 2) * C + c] = (float)sc;
      tab[(b * 2 + 1) * C + c] = (float)((double)br.beta[c] - m * sc);
      if (blockIdx.x == 0) {
        br.scale[c] = (float)sc; br.shift[c] = (float)((double)br.beta[c] - m * sc);
        if (br.mean) br.mean[c] = (float)m;
        if (br.rstd) br.rstd[c] = (float)r;
        if (p.training && br.out_stats) {               // statistics of the OUTPUT (no ReLU), from the coefficients: exact up to the output's rounding
          const double be = (double)br.beta[c];
          br.out_stats[c] = p.count * be;
          br.out_stats[C + c] = p.count * (be * be + sc * sc * v);
        }
        if (p.training && br.moving_mean) {
          const double unb = p.bessel_n > 1 ? v * (p.bessel_n / (p.bessel_n - 1)) : v;
          br.moving_mean[c] = (float)((double)br.moving_mean[c] * p.momentum + m * (1.0 - p.momentum));
          br.moving_var[c] = (float)((double)br.moving_var[c] * p.momentum + unb * (1.0 - p.momentum));
        }
      }
    }
    (void)mr;
  }
  __syncthreads();
  if (p.br[0].out == nullptr) return;                  // coefficients only
  const unsigned char* x = (const unsigned char*)p.x;
  const long long stride = (long long)gridDim.x * 256;
  if constexpr (NB > 0) {
    // 256 % CG == 0 (the launcher checked): a thread stays on ONE channel group for the whole sweep, so its coefficients live in
    // registers - the per-piece reads of the LDS table (8 consecutive floats per lane, lanes on different groups: the 0.65 LDS
    // bank-conflict share of the round-2 profile) are gone from the loop
    const int c = (int)(threadIdx.x % CG) * VEC;
    float sc[NB][VEC], sh[NB][VEC];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { sc[b][j] = tab[(b * 2) * C + c + j]; sh[b][j] = tab[(b * 2 + 1) * C + c + j]; }
    unsigned char* outs[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) outs[b] = (unsigned char*)p.br[b].out;
    const bool relu = p.relu != 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += stride) {
      float xv[VEC];
      ET<T>::unpack(ldg16(x + i * 16), xv);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          o[j] = fmaf(sc[b][j], xv[j], sh[b][j]);
          if (relu) o[j] = fmaxf(o[j], 0.f);
        }
        stg16(outs[b] + i * 16, ET<T>::pack(o));
      }
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += stride) {
      const int c = (int)(i % CG) * VEC;
      float xv[VEC];
      ET<T>::unpack(ldg16(x + i * 16), xv);
      for (int b = 0; b < p.nb; ++b) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          o[j] = fmaf(tab[(b * 2) * C + c + j], xv[j], tab[(b * 2 + 1) * C + c + j]);
          if (p.relu) o[j] = fmaxf(o[j], 0.f);
        }
        stg16((unsigned char*)p.br[b].out + i * 16, ET<T>::pack(o));
      }
    }
  }
}

template <typename T, int NB>
__global__ __launch_bounds__(256) void bn_fwd_kernel(const rua_bn_fwd_desc p, long long pieces, int CG) { bn_fwd_body<T, NB>(p, pieces, CG); }
// rua_bn_fwd_group: independent one-branch BatchNorm applications of equal shape (the second BatchNorms of a ResBlock's dilation branches
// where they are materialised: model2.py:21, levels 3 - 6) as ONE grid - blockIdx.y picks the member
struct BnFwdG { rua_bn_fwd_desc k[RUA_MAX_BRANCH]; long long pieces[RUA_MAX_BRANCH]; };
static_assert(sizeof(BnFwdG) <= 4096, "grouped launch: kernel arguments are limited to 4 KiB");
template <typename T>
__global__ __launch_bounds__(256) void bn_fwd_kernel_g(const BnFwdG g, int CG) { bn_fwd_body<T, 1>(g.k[blockIdx.y], g.pieces[blockIdx.y], CG); }

extern "C" int rua_bn_fwd(const rua_bn_fwd_desc* d, void* stream);
static thread_local int g_bn_fwd_group_last = 0;
extern "C" int rua_bn_fwd_group_last_grids(void) { return g_bn_fwd_group_last; }
extern "C" int rua_bn_fwd_group(const rua_bn_fwd_desc* d, int n, void* stream) {
  RUA_CHECK_ARG(d && n >= 1 && n <= RUA_MAX_BRANCH, "rua_bn_fwd_group: 1..%d members", RUA_MAX_BRANCH);
  const int vec = d[0].dtype == RUA_BF16 ? 8 : 4;
  bool one = n > 1 && g_tune.bn_bwd_group && d[0].C % vec == 0 && d[0].C / vec <= 256 && 256 % (d[0].C / vec) == 0 && g_tune.bn_regs &&
             (d[0].dtype == RUA_F32 || d[0].dtype == RUA_BF16) && (size_t)4 * d[0].C * 4 <= 64 * 1024;
  for (int i = 0; i < n && one; ++i) {
    const rua_bn_fwd_desc& m = d[i];
    const rua_bn_branch& br = m.br[0];
    one = m.x && m.nb == 1 && m.M > 0 && m.C == d[0].C && m.dtype == d[0].dtype && m.training == d[0].training && m.relu == d[0].relu && br.out &&
          br.gamma && br.beta && br.scale && br.shift && (!m.training || (m.count > 0 && (br.stats ? br.replicas >= 1 : (m.stats && m.replicas >= 1)))) &&
          (m.training || (br.moving_mean && br.moving_var)) && (br.out_stats == nullptr || (m.relu == 0 && m.training));
  }
  g_bn_fwd_group_last = n;
  if (!one) {                                            // same results member by member
    for (int i = 0; i < n; ++i) { const int rc = rua_bn_fwd(d + i, stream); if (rc != RUA_OK) return rc; }
    return RUA_OK;
  }
  const int CG = d[0].C / vec;
  const size_t smem = (size_t)(1 * 2 + 2) * d[0].C * 4;
  BnFwdG a;
  long long pieces = 0;
  for (int i = 0; i < n; ++i) { a.k[i] = d[i]; a.pieces[i] = d[i].M * CG; if (a.pieces[i] > pieces) pieces = a.pieces[i]; }
  int g = grid_for(pieces);
  const int cap = 2 * rua_cu_count() / (n > 2 ? 2 : 1);
  if (g > cap) g = cap;
  hipStream_t st = (hipStream_t)stream;
  if (d[0].dtype == RUA_BF16) hipLaunchKernelGGL((bn_fwd_kernel_g<bf16_t>), dim3(g, n), dim3(256), smem, st, a, CG);
  else hipLaunchKernelGGL((bn_fwd_kernel_g<float>), dim3(g, n), dim3(256), smem, st, a, CG);
  RUA_LAUNCH_CHECK("rua_bn_fwd_group");
  g_bn_fwd_group_last = 1;
  return RUA_OK;
}

extern "C" int rua_bn_fwd(const rua_bn_fwd_desc* d, void* stream) {
  RUA_CHECK_ARG(d && d->nb >= 1 && d->nb <= RUA_MAX_BRANCH && d->M > 0 && d->C > 0, "rua_bn_fwd: bad arguments");
  const bool coef_only = d->br[0].out == nullptr;
  RUA_CHECK_ARG(coef_only || d->x, "rua_bn_fwd: null input");
  RUA_CHECK_ARG(d->dtype == RUA_F32 || d->dtype == RUA_BF16, "rua_bn_fwd: bad dtype");
  const int vec = d->dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(d->C % vec == 0, "rua_bn_fwd: C=%d not a multiple of %d", d->C, vec);
  RUA_CHECK_ARG(!d->training || d->count > 0, "rua_bn_fwd: training needs the element count");
  for (int b = 0; b < d->nb; ++b) {
    const rua_bn_branch& br = d->br[b];
    RUA_CHECK_ARG(br.gamma && br.beta && br.scale && br.shift, "rua_bn_fwd: null branch pointer");
    RUA_CHECK_ARG((br.out == nullptr) == coef_only, "rua_bn_fwd: either every branch has an output or none (coefficients only)");
    RUA_CHECK_ARG(!d->training || (br.stats ? br.replicas >= 1 : (d->stats && d->replicas >= 1)), "rua_bn_fwd: training needs statistics");
    RUA_CHECK_ARG(d->training || (br.moving_mean && br.moving_var), "rua_bn_fwd: inference needs moving statistics");
    RUA_CHECK_ARG(br.out_stats == nullptr || (d->relu == 0 && d->training), "rua_bn_fwd: out_stats is the statistics of a training-mode BatchNorm WITHOUT ReLU");
  }
  const size_t smem = (size_t)(d->nb * 2 + 2) * d->C * 4;
  RUA_CHECK_ARG(smem <= 64 * 1024, "rua_bn_fwd: coefficient table too large");
  const int CG = d->C / vec;
  const long long pieces = d->M * CG;
  int g = grid_for(pieces);
  const int cap = g_tune.bn_grid > 0 ? g_tune.bn_grid : 2 * rua_cu_count();      // 512 on MI355X
  if ((long long)d->replicas * d->C >= 512 && g > cap) g = cap;      // the prologue re-reads replicas*C*2 doubles per block
  if (coef_only) g = 1;
  hipStream_t st = (hipStream_t)stream;
  const int nbk = (!coef_only && CG <= 256 && 256 % CG == 0 && g_tune.bn_regs) ? d->nb : 0;
#define RUA_BN_FWD_GO(T_, NB_) hipLaunchKernelGGL((bn_fwd_kernel<T_, NB_>), dim3(g), dim3(256), smem, st, *d, pieces, CG)
  if (d->dtype == RUA_BF16) {
    switch (nbk) { case 1: RUA_BN_FWD_GO(bf16_t, 1); break; case 2: RUA_BN_FWD_GO(bf16_t, 2); break; case 3: RUA_BN_FWD_GO(bf16_t, 3); break;
                   case 4: RUA_BN_FWD_GO(bf16_t, 4); break; default: RUA_BN_FWD_GO(bf16_t, 0); }
  } else {
    switch (nbk) { case 1: RUA_BN_FWD_GO(float, 1); break; case 2: RUA_BN_FWD_GO(float, 2); break; case 3: RUA_BN_FWD_GO(float, 3); break;
                   case 4: RUA_BN_FWD_GO(float, 4); break; default: RUA_BN_FWD_GO(float, 0); }
  }
#undef RUA_BN_FWD_GO
  RUA_LAUNCH_CHECK("rua_bn_fwd");
  return RUA_OK;
}

// (Measured and rejected, same-box A/B: issuing the sweep's first loads before the coefficient prologue and double-buffering the
// sweep in registers - 9.70 vs 9.68 ms per step; the blocks of one launch already overlap each other's prologue.)
// DXS: the launch also takes the per-channel sums of dx and of dx * x (rua_bn_bwd_desc.dx_stats).  A template flag: the 16 accumulators take the
// four-branch form from 124 to 144 registers (four waves per SIMD -> three: 42 -> 55 us on the 256 x 256 x 32 launch), which only the launches that ask pay.
template <typename T, int NB, bool MASKED, bool DXS = false>
__device__ __forceinline__ void bn_bwd_body(const rua_bn_bwd_desc& p, long long pieces, int CG) {
  constexpr int VEC = ET<T>::VEC;
  extern __shared__ float tab[];                       // [nb][3][C] : A, ms, mt ; then [2][C] : sumB, sumC ; then [nb][2][C] : the branches' terms of sumB, sumC
  const int C = p.C;
  float* tB = tab + p.nb * 3 * C;
  float* tT = tB + 2 * C;
  // one thread per (branch, channel): the branches' statistics are fetched side by side - one chain of dependent round trips for
  // the block instead of one per branch (a four-branch launch began with ~8 of them before its first tensor load)
  for (int it = threadIdx.x; it < p.nb * C; it += 256) {
    const int b = it / C, c = it - b * C;
    const rua_bn_bwd_branch& br = p.br[b];
    double sg, sgx;
    replica_sum(br.stats2, br.replicas, C, c, sg, sgx);
    const double mu = br.mean[c], r = br.rstd[c], gm = br.gamma[c];
    if (br.stats2_out) {                               // slot 1 holds sum g * (scale x + shift): back to sum g * x
      const double sc = br.scale[c], sh = br.shift[c];
      sgx = sc != 0.0 ? (sgx - sh * sg) / sc : mu * sg;
    }
    const double dga = r * (sgx - mu * sg);
    const double s = gm * r;
    tab[(b * 3) * C + c] = (float)s;
    tab[(b * 3 + 1) * C + c] = (p.masked && br.scale) ? br.scale[c] : 1.f;
    tab[(b * 3 + 2) * C + c] = (p.masked && br.shift) ? br.shift[c] : 0.f;
    tT[(b * 2) * C + c] = (float)(-s * r * dga / p.count);
    tT[(b * 2 + 1) * C + c] = (float)(-s * sg / p.count + s * r * mu * dga / p.count);
    if (blockIdx.x == 0) {
      if (br.dgamma) br.dgamma[c] += (float)dga;
      if (br.dbeta) br.dbeta[c] += (float)sg;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {         // the branches' terms in branch order (the sums of the sequential form, bit for bit)
    float sb = 0.f, sc_ = 0.f;
    for (int b = 0; b < p.nb; ++b) { sb += tT[(b * 2) * C + c]; sc_ += tT[(b * 2 + 1) * C + c]; }
    tB[c] = sb; tB[C + c] = sc_;
  }
  __syncthreads();
  const unsigned char* x = (const unsigned char*)p.x;
  const unsigned char* dskip = (const unsigned char*)p.dskip;
  unsigned char* dx = (unsigned char*)p.dx;
  const long long stride = (long long)gridDim.x * 256;
  // per-channel sum of dskip (256 is a multiple of CG, so a thread stays on one channel group for the whole sweep)
  float sk[VEC], so[VEC], so2[VEC];                    // so / so2: per-channel sums of dx and of dx * x over what this thread writes (p.dx_stats)
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sk[j] = 0.f; so[j] = 0.f; so2[j] = 0.f; }
  if constexpr (NB > 0) {
    // coefficients of this thread's channel group in registers for the whole sweep (see bn_fwd_kernel): no LDS reads in the loop
    const int c = (int)(threadIdx.x % CG) * VEC;
    float cA[NB][VEC], cS[MASKED ? NB : 1][VEC], cT[MASKED ? NB : 1][VEC], cB[VEC], cC[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { cB[j] = tB[c + j]; cC[j] = tB[C + c + j]; }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        cA[b][j] = tab[(b * 3) * C + c + j];
        if constexpr (MASKED) { cS[b][j] = tab[(b * 3 + 1) * C + c + j]; cT[b][j] = tab[(b * 3 + 2) * C + c + j]; }
      }
    const unsigned char* gp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gp[b] = (const unsigned char*)p.br[b].g;
    const bool accum = p.accumulate != 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += stride) {
      // every load of the piece is issued before the first use
      const uint4 xr = ldg16(x + i * 16);
      uint4 gr[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) gr[b] = ldg16(gp[b] + i * 16);
      uint4 dr = make_uint4(0, 0, 0, 0), ar = make_uint4(0, 0, 0, 0);
      if (dskip) dr = ldg16(dskip + i * 16);
      if (accum) ar = ldg16(dx + i * 16);
      float xv[VEC], acc[VEC];
      ET<T>::unpack(xr, xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(cB[j], xv[j], cC[j]);
      if (dskip) {
        float dd[VEC];
        ET<T>::unpack(dr, dd);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { acc[j] += dd[j]; sk[j] += dd[j]; }
      }
      if (accum) {
        float dd[VEC];
        ET<T>::unpack(ar, dd);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += dd[j];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float gv[VEC];
        ET<T>::unpack(gr[b], gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          bool on = true;
          if constexpr (MASKED) on = fmaf(cS[b][j], xv[j], cT[b][j]) > 0.f;
          acc[j] = fmaf(cA[b][j], on ? gv[j] : 0.f, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) if constexpr (DXS) { so[j] += acc[j]; so2[j] = fmaf(acc[j], xv[j], so2[j]); }
      stg16(dx + i * 16, ET<T>::pack(acc));
    }
  } else {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += stride) {
    const int c = (int)(i % CG) * VEC;
    float xv[VEC], acc[VEC];
    ET<T>::unpack(ldg16(x + i * 16), xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(tB[c + j], xv[j], tB[C + c + j]);
    if (dskip) {
      float dd[VEC];
      ET<T>::unpack(ldg16(dskip + i * 16), dd);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { acc[j] += dd[j]; sk[j] += dd[j]; }
    }
    if (p.accumulate) {
      float dd[VEC];
      ET<T>::unpack(ldg16(dx + i * 16), dd);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += dd[j];
    }
    for (int b = 0; b < p.nb; ++b) {
      float gv[VEC];
      ET<T>::unpack(ldg16((const unsigned char*)p.br[b].g + i * 16), gv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const bool on = !p.masked || fmaf(tab[(b * 3 + 1) * C + c + j], xv[j], tab[(b * 3 + 2) * C + c + j]) > 0.f;
        acc[j] = fmaf(tab[(b * 3) * C + c + j], on ? gv[j] : 0.f, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) if constexpr (DXS) { so[j] += acc[j]; so2[j] = fmaf(acc[j], xv[j], so2[j]); }
    stg16(dx + i * 16, ET<T>::pack(acc));
  }
  }
  // Per-channel sums of the block (sums of dskip, of dx, of dx * x): every thread holds VEC channels of ONE channel group (256 % CG == 0), so the lanes of a wave
  // that share a group are folded with shuffles first (CG <= 32: 64 / CG lanes each), the four waves through [4][CG][VEC] floats of LDS, one fp64 add per channel
  // and block.  (The first form staged all 256 x VEC partials in LDS and let C threads walk 256 / CG of them each: +9 us on the 256 x 256 x 32 launch.)
  auto fold = [&](float* v, double* dst, int replicas, int slot) {
    float* red = tab;                                    // the coefficient table is dead
    __syncthreads();
    if (CG <= 32) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        for (int o = CG; o < 64; o <<= 1) v[j] += __shfl_xor(v[j], o, 64);
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      if (lane < CG) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[(wv * CG + lane) * VEC + j] = v[j];
      }
      __syncthreads();
      for (int ch = threadIdx.x; ch < C; ch += 256) {
        const float t = (red[ch] + red[C + ch]) + (red[2 * C + ch] + red[3 * C + ch]);       // [wave][CG][VEC] = [wave][C]
        unsafeAtomicAdd(&dst[(size_t)(blockIdx.x & (replicas - 1)) * 2 * C + slot * C + ch], (double)t);
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = v[j];
      __syncthreads();
      const int per = 256 / CG;                          // threads per channel group (2 or 4)
      for (int ch = threadIdx.x; ch < C; ch += 256) {
        const int cg = ch / VEC, j = ch - cg * VEC;
        float t = 0.f;
        for (int k = 0; k < per; ++k) t += red[(k * CG + cg) * VEC + j];
        unsafeAtomicAdd(&dst[(size_t)(blockIdx.x & (replicas - 1)) * 2 * C + slot * C + ch], (double)t);
      }
    }
  };
  if (p.skip_stats && dskip) fold(sk, p.skip_stats, p.skip_replicas, 0);      // (uniform branches)
  if constexpr (DXS) { fold(so, p.dx_stats, p.dx_replicas, 0); fold(so2, p.dx_stats, p.dx_replicas, 1); }
}

template <typename T, int NB, bool MASKED, bool DXS = false>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const rua_bn_bwd_desc p, long long pieces, int CG) { bn_bwd_body<T, NB, MASKED, DXS>(p, pieces, CG); }
// rua_bn_bwd_group: independent one-branch BatchNorm backwards of equal shape (the second BatchNorms of a ResBlock's dilation branches:
// model2.py:21-22, one per branch, each with its own gradient, input and output) as ONE grid - blockIdx.y picks the member
struct BnBwdG { rua_bn_bwd_desc k[RUA_MAX_BRANCH]; long long pieces[RUA_MAX_BRANCH]; };     // members of unequal pixel counts (the PSPPooling branches) sweep their own range
static_assert(sizeof(BnBwdG) <= 4096, "grouped launch: kernel arguments are limited to 4 KiB");
template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void bn_bwd_kernel_g(const BnBwdG g, int CG) { bn_bwd_body<T, 1, MASKED>(g.k[blockIdx.y], g.pieces[blockIdx.y], CG); }

extern "C" int rua_bn_bwd(const rua_bn_bwd_desc* d, void* stream);
static thread_local int g_bn_bwd_group_last = 0;
extern "C" int rua_bn_bwd_group_last_grids(void) { return g_bn_bwd_group_last; }     // grids the calling thread's latest rua_bn_bwd_group issued
extern "C" int rua_bn_bwd_group(const rua_bn_bwd_desc* d, int n, void* stream) {
  RUA_CHECK_ARG(d && n >= 1 && n <= RUA_MAX_BRANCH, "rua_bn_bwd_group: 1..%d members", RUA_MAX_BRANCH);
  const int vec = d[0].dtype == RUA_BF16 ? 8 : 4;
  bool one = n > 1 && g_tune.bn_bwd_group && d[0].C % vec == 0 && d[0].C / vec <= 256 && 256 % (d[0].C / vec) == 0 && g_tune.bn_regs;
  for (int i = 0; i < n && one; ++i) {
    const rua_bn_bwd_desc& m = d[i];
    one = m.x && m.dx && m.nb == 1 && m.M > 0 && m.C == d[0].C && m.dtype == d[0].dtype && m.masked == d[0].masked && !m.skip_stats && !m.dx_stats && m.count > 0 &&
          m.br[0].g && m.br[0].stats2 && m.br[0].replicas >= 1 && m.br[0].gamma && m.br[0].mean && m.br[0].rstd &&
          (d[0].dtype == RUA_F32 || d[0].dtype == RUA_BF16);
  }
  g_bn_bwd_group_last = n;
  if (!one) {                                            // same results member by member
    for (int i = 0; i < n; ++i) { const int rc = rua_bn_bwd(d + i, stream); if (rc != RUA_OK) return rc; }
    return RUA_OK;
  }
  const int CG = d[0].C / vec;
  const size_t smem = (size_t)(1 * 5 + 2) * d[0].C * 4;
  BnBwdG a;
  long long pieces = 0;                                   // the grid is sized for the largest member (the sweep is grid-stride)
  for (int i = 0; i < n; ++i) { a.k[i] = d[i]; a.pieces[i] = d[i].M * CG; if (a.pieces[i] > pieces) pieces = a.pieces[i]; }
  int g = grid_for(pieces);
  const int cap = (pieces >= (1ll << 20) ? 4 : 2) * rua_cu_count() / (n > 2 ? 2 : 1);    // the members fill the chip together
  if (g > cap) g = cap;
  hipStream_t st = (hipStream_t)stream;
  const bool mk = d[0].masked != 0;
  if (d[0].dtype == RUA_BF16) {
    if (mk) hipLaunchKernelGGL((bn_bwd_kernel_g<bf16_t, true>), dim3(g, n), dim3(256), smem, st, a, CG);
    else hipLaunchKernelGGL((bn_bwd_kernel_g<bf16_t, false>), dim3(g, n), dim3(256), smem, st, a, CG);
  } else {
    if (mk) hipLaunchKernelGGL((bn_bwd_kernel_g<float, true>), dim3(g, n), dim3(256), smem, st, a, CG);
    else hipLaunchKernelGGL((bn_bwd_kernel_g<float, false>), dim3(g, n), dim3(256), smem, st, a, CG);
  }
  RUA_LAUNCH_CHECK("rua_bn_bwd_group");
  g_bn_bwd_group_last = 1;
  return RUA_OK;
}

extern "C" int rua_bn_bwd(const rua_bn_bwd_desc* d, void* stream) {
  RUA_CHECK_ARG(d && d->x && d->dx && d->nb >= 1 && d->nb <= RUA_MAX_BRANCH && d->M > 0 && d->C > 0 && d->count > 0, "rua_bn_bwd: bad arguments");
  RUA_CHECK_ARG(d->dtype == RUA_F32 || d->dtype == RUA_BF16, "rua_bn_bwd: bad dtype");
  const int vec = d->dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(d->C % vec == 0, "rua_bn_bwd: C=%d not a multiple of %d", d->C, vec);
  int rmax = 1;
  for (int b = 0; b < d->nb; ++b) {
    const rua_bn_bwd_branch& br = d->br[b];
    RUA_CHECK_ARG(br.g && br.stats2 && br.replicas >= 1 && br.gamma && br.mean && br.rstd, "rua_bn_bwd: null branch pointer");
    if (br.replicas > rmax) rmax = br.replicas;
  }
  size_t smem = (size_t)(d->nb * 5 + 2) * d->C * 4;
  RUA_CHECK_ARG(smem <= 160 * 1024, "rua_bn_bwd: coefficient table too large");      // (5 nb + 2) C floats: nb = 4 up to C = 1861, nb = 1 up to C = 5851
  const int CG = d->C / vec;
  if (d->skip_stats) {
    RUA_CHECK_ARG(d->dskip && d->skip_replicas >= 1 && (d->skip_replicas & (d->skip_replicas - 1)) == 0 && CG <= 256 && 256 % CG == 0,
                  "rua_bn_bwd: skip_stats needs dskip, a power-of-two replica count and C / %d dividing 256", vec);
    if (smem < (size_t)256 * vec * 4) smem = (size_t)256 * vec * 4;
  }
  if (d->dx_stats) {
    RUA_CHECK_ARG(d->dx_replicas >= 1 && (d->dx_replicas & (d->dx_replicas - 1)) == 0 && CG <= 256 && 256 % CG == 0,
                  "rua_bn_bwd: dx_stats needs a power-of-two replica count and C / %d dividing 256", vec);
    if (smem < (size_t)256 * vec * 4) smem = (size_t)256 * vec * 4;
  }
  const long long pieces = d->M * CG;
  int g = grid_for(pieces);
  // measured (4 branches): 256x256x32 53 -> 46 us, 128x128x64 35 -> 31 us with 1024 blocks; smaller tensors prefer 512
  const int cap_env = g_tune.bn_grid;
  const int cap = cap_env > 0 ? cap_env : (pieces >= (1ll << 20) ? 4 : 2) * rua_cu_count();      // 1024 / 512 on MI355X
  if ((long long)rmax * d->C * d->nb >= 512 && g > cap) g = cap;
  hipStream_t st = (hipStream_t)stream;
  const int nbk = (CG <= 256 && 256 % CG == 0 && g_tune.bn_regs) ? d->nb : 0;
  const bool mk = d->masked != 0;
  // a table beyond 64 KB (wider variants than the shipped configurations) needs the kernel's dynamic-LDS limit raised: a rare path, set per launch
#define RUA_BN_BWD_GO1(T_, NB_, MK_, DX_) do { if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)bn_bwd_kernel<T_, NB_, MK_, DX_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
                                               hipLaunchKernelGGL((bn_bwd_kernel<T_, NB_, MK_, DX_>), dim3(g), dim3(256), smem, st, *d, pieces, CG); } while (0)
#define RUA_BN_BWD_GO(T_, NB_, MK_) do { if (d->dx_stats) RUA_BN_BWD_GO1(T_, NB_, MK_, true); else RUA_BN_BWD_GO1(T_, NB_, MK_, false); } while (0)
#define RUA_BN_BWD_SW(T_) switch (nbk * 2 + (mk ? 1 : 0)) { \
    case 2: RUA_BN_BWD_GO(T_, 1, false); break; case 3: RUA_BN_BWD_GO(T_, 1, true); break; \
    case 4: RUA_BN_BWD_GO(T_, 2, false); break; case 5: RUA_BN_BWD_GO(T_, 2, true); break; \
    case 6: RUA_BN_BWD_GO(T_, 3, false); break; case 7: RUA_BN_BWD_GO(T_, 3, true); break; \
    case 8: RUA_BN_BWD_GO(T_, 4, false); break; case 9: RUA_BN_BWD_GO(T_, 4, true); break; \
    default: RUA_BN_BWD_GO(T_, 0, false); }
  if (d->dtype == RUA_BF16) { RUA_BN_BWD_SW(bf16_t) } else { RUA_BN_BWD_SW(float) }
#undef RUA_BN_BWD_SW
#undef RUA_BN_BWD_GO
#undef RUA_BN_BWD_GO1
  RUA_LAUNCH_CHECK("rua_bn_bwd");
  return RUA_OK;
}

// argmax bytes of one piece (VEC = 8: one 8-byte access, VEC = 4: one 4-byte access) instead of VEC byte accesses
template <int VEC> __device__ __forceinline__ void idx_load(const uint8_t* p, int* out) {
  if constexpr (VEC == 8) { const uint2 q = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[j] = (q.x >> (8 * j)) & 255; out[4 + j] = (q.y >> (8 * j)) & 255; } }
  else { const unsigned q = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (q >> (8 * j)) & 255; }
}
template <int VEC> __device__ __forceinline__ void idx_store(uint8_t* p, const int* v) {
  if constexpr (VEC == 8) { uint2 q; q.x = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
    q.y = (unsigned)v[4] | ((unsigned)v[5] << 8) | ((unsigned)v[6] << 16) | ((unsigned)v[7] << 24);
    *reinterpret_cast<uint2*>(p) = q; }
  else *reinterpret_cast<unsigned*>(p) = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
}

// ---------------------------------------------------------------------------------------
// pooling / resampling: one thread per output (or input) piece
// K > 0: window size known at compile time - the K loads of a window row are issued together (the runtime-k loop did
// k*k dependent loads per output: 22 us for the 8x8 windows of the PSP pyramids); K == 0: any k.
template <typename T, int K>
__global__ void maxpool_fwd_kernel(const unsigned char* x, unsigned char* y, uint8_t* idx, int N, int H, int W, int CG, int k_rt) {
  constexpr int VEC = ET<T>::VEC;
  const int k = K > 0 ? K : k_rt;
  const int Ho = H / k, Wo = W / k;
  const long long total = (long long)N * Ho * Wo * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
    float best[VEC]; int bi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    if constexpr (K > 0) {
      for (int a = 0; a < K; ++a) {
        uint4 q[K];
#pragma unroll
        for (int b = 0; b < K; ++b) q[b] = ldg16(x + ((((size_t)n * H + ho * K + a) * W + wo * K + b) * CG + cp) * 16);
#pragma unroll
        for (int b = 0; b < K; ++b) {
          float v[VEC];
          ET<T>::unpack(q[b], v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) if (v[j] > best[j]) { best[j] = v[j]; bi[j] = a * K + b; }   // first maximum wins
        }
      }
    } else {
      for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
          float v[VEC];
          ET<T>::unpack(ldg16(x + ((((size_t)n * H + ho * k + a) * W + wo * k + b) * CG + cp) * 16), v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) if (v[j] > best[j]) { best[j] = v[j]; bi[j] = a * k + b; }   // first maximum wins
        }
    }
    stg16(y + i * 16, ET<T>::pack(best));
    idx_store<VEC>(idx + i * VEC, bi);
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const unsigned char* dy, const uint8_t* idx, unsigned char* dx, int accumulate,
                                   int N, int H, int W, int CG, int k) {
  constexpr int VEC = ET<T>::VEC;
  const int Ho = H / k, Wo = W / k;
  const long long total = (long long)N * H * W * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); const int n = (int)(r / H);
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
    if (accumulate) ET<T>::unpack(ldg16(dx + i * 16), o);
    if (h / k < Ho && w / k < Wo) {
      const size_t pi = (((size_t)n * Ho + h / k) * Wo + w / k) * CG + cp;
      float g[VEC];
      ET<T>::unpack(ldg16(dy + pi * 16), g);
      const int pos = (h % k) * k + (w % k);
      int ib[VEC];
      idx_load<VEC>(idx + pi * VEC, ib);
#pragma unroll
      for (int j = 0; j < VEC; ++j) if (ib[j] == pos) o[j] += g[j];
    }
    stg16(dx + i * 16, ET<T>::pack(o));
  }
}

template <typename T, int K>
__global__ void sumpool_kernel(const unsigned char* x, unsigned char* y, int N, int H, int W, int CG, int k_rt) {
  constexpr int VEC = ET<T>::VEC;
  const int k = K > 0 ? K : k_rt;
  const int Ho = H / k, Wo = W / k;
  const long long total = (long long)N * Ho * Wo * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    if constexpr (K > 0) {
      for (int a = 0; a < K; ++a) {
        uint4 q[K];
#pragma unroll
        for (int b = 0; b < K; ++b) q[b] = ldg16(x + ((((size_t)n * H + ho * K + a) * W + wo * K + b) * CG + cp) * 16);
#pragma unroll
        for (int b = 0; b < K; ++b) {
          float v[VEC];
          ET<T>::unpack(q[b], v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) s[j] += v[j];
        }
      }
    } else {
      for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
          float v[VEC];
          ET<T>::unpack(ldg16(x + ((((size_t)n * H + ho * k + a) * W + wo * k + b) * CG + cp) * 16), v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) s[j] += v[j];
        }
    }
    stg16(y + i * 16, ET<T>::pack(s));
  }
}

template <typename T>
__global__ void upsample_kernel(const unsigned char* x, unsigned char* y, int N, int H, int W, int CG, int k) {
  const int Ho = H * k, Wo = W * k;
  const long long total = (long long)N * Ho * Wo * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
    stg16(y + i * 16, ldg16(x + ((((size_t)n * H + ho / k) * W + wo / k) * CG + cp) * 16));
  }
}

// ---- the 2 / 4 / 8 pooling pyramid of a PSPPooling (model2.py:47-60) ------------------------------------------------------------
// Max-pool with window 2k from the max-pool with window k of the same tensor and its argmax bytes: a thread compares the four
// k x k winners of its 2k x 2k window; on equal values the smaller row-major position inside the 2k x 2k window wins, which is
// what a direct scan of the window (first maximum wins) finds.  The 4- and 8-windows of the pyramid come from the 2-windows this
// way: x is read once, by the k = 2 pass (a thread per 8x8 window doing all three levels in one pass was tried: 31 us against 21
// for three separate passes - 1/64 of the threads, 64 dependent compare rounds each).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_derive_kernel(const unsigned char* yp, const uint8_t* ip, unsigned char* y, uint8_t* idx,
                                                             int N, int Hp, int Wp, int CG, int kp) {
  constexpr int VEC = ET<T>::VEC;
  const int Ho = Hp / 2, Wo = Wp / 2, k = 2 * kp;
  const long long total = (long long)N * Ho * Wo * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
    uint4 q[4]; size_t o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o[u] = (((size_t)n * Hp + ho * 2 + (u >> 1)) * Wp + wo * 2 + (u & 1)) * CG + cp;
      q[u] = ldg16(yp + o[u] * 16);
    }
    float best[VEC]; int bi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = k * k; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[VEC];
      ET<T>::unpack(q[u], v);
      int ib[VEC];
      idx_load<VEC>(ip + o[u] * VEC, ib);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int pp = ib[j];
        const int pos = ((u >> 1) * kp + pp / kp) * k + (u & 1) * kp + pp % kp;
        if (v[j] > best[j] || (v[j] == best[j] && pos < bi[j])) { best[j] = v[j]; bi[j] = pos; }
      }
    }
    stg16(y + i * 16, ET<T>::pack(best));
    idx_store<VEC>(idx + i * VEC, bi);
  }
}

// dx (=|+=) the scatter of up to three pooled gradients (k = ks[0..n)) through their argmax bytes: ONE read-modify-write of dx
// instead of one per pooling size (45 -> 16 us at 256x256x32).
struct PoolPyr { const unsigned char* dy[3]; const uint8_t* idx[3]; int k[3]; int n; };
// POW2: W, H, the channel groups and every window are powers of two (every level of the reference network): the index arithmetic is shifts and masks - with run-time divisors it
// was a dozen integer divisions per 16 bytes of dx, and the pass ran at 2.6 TB/s
template <typename T, bool POW2>
__global__ __launch_bounds__(256) void maxpool_bwd_multi_kernel(const PoolPyr q, unsigned char* dx, int accumulate, int N, int H, int W, int CG, int cgs, int ws, int hs) {
  constexpr int VEC = ET<T>::VEC;
  const long long total = (long long)N * H * W * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int cp, w, h, n;
    if constexpr (POW2) {
      const unsigned ii = (unsigned)i;                 // host: total < 2^31
      cp = (int)(ii & (unsigned)(CG - 1)); const unsigned r = ii >> cgs;
      w = (int)(r & (unsigned)(W - 1)); h = (int)((r >> ws) & (unsigned)(H - 1)); n = (int)(r >> (ws + hs));
    } else {
      cp = (int)(i % CG); long long r = i / CG;
      w = (int)(r % W); r /= W;
      h = (int)(r % H); n = (int)(r / H);
    }
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
    if (accumulate) ET<T>::unpack(ldg16(dx + i * 16), o);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (u >= q.n) break;
      const int k = q.k[u];
      size_t pi; int pos;
      if constexpr (POW2) {
        const int ks = 31 - __clz(k);
        pi = ((((size_t)n << (hs - ks)) + (h >> ks)) << (ws - ks)) + (w >> ks); pi = (pi << cgs) + cp;
        pos = ((h & (k - 1)) << ks) + (w & (k - 1));
      } else {
        const int Ho = H / k, Wo = W / k;
        pi = (((size_t)n * Ho + h / k) * Wo + w / k) * CG + cp;
        pos = (h % k) * k + (w % k);
      }
      float g[VEC];
      ET<T>::unpack(ldg16(q.dy[u] + pi * 16), g);
      int ib[VEC];
      idx_load<VEC>(q.idx[u] + pi * VEC, ib);
#pragma unroll
      for (int j = 0; j < VEC; ++j) if (ib[j] == pos) o[j] += g[j];
    }
    stg16(dx + i * 16, ET<T>::pack(o));
  }
}

// sums over the 2x2, 4x4 and 8x8 windows of x in one pass (the adjoint of the folded nearest upsampling of the three pooled
// PSPPooling branches): fp32 partial sums carried up the pyramid, each level rounded once
template <typename T>
__global__ __launch_bounds__(256) void sumpool_pyramid_kernel(const unsigned char* x, unsigned char* y2, unsigned char* y4, unsigned char* y8,
                                                              int N, int H, int W, int CG) {
  constexpr int VEC = ET<T>::VEC;
  const int H8 = H / 8, W8 = W / 8;
  const long long total = (long long)N * H8 * W8 * CG;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CG); long long r = i / CG;
    const int w8 = (int)(r % W8); r /= W8;
    const int h8 = (int)(r % H8); const int n = (int)(r / H8);
    float s8[VEC], s4[2][VEC], s2[4][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s8[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      uint4 q[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) q[b] = ldg16(x + ((((size_t)n * H + h8 * 8 + a) * W + w8 * 8 + b) * CG + cp) * 16);
      if ((a & 1) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) s2[u][j] = 0.f;
      }
      if ((a & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) s4[u][j] = 0.f;
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        float v[VEC];
        ET<T>::unpack(q[b], v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s2[b >> 1][j] += v[j];
      }
      if (a & 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t o = (((size_t)n * (H / 2) + h8 * 4 + (a >> 1)) * (W / 2) + w8 * 4 + u) * CG + cp;
          stg16(y2 + o * 16, ET<T>::pack(s2[u]));
#pragma unroll
          for (int j = 0; j < VEC; ++j) s4[u >> 1][j] += s2[u][j];
        }
      }
      if ((a & 3) == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const size_t o = (((size_t)n * (H / 4) + h8 * 2 + (a >> 2)) * (W / 4) + w8 * 2 + u) * CG + cp;
          stg16(y4 + o * 16, ET<T>::pack(s4[u]));
#pragma unroll
          for (int j = 0; j < VEC; ++j) s8[j] += s4[u][j];
        }
      }
    }
    stg16(y8 + i * 16, ET<T>::pack(s8));
  }
}

#define POOL_ARGS_CHECK(name) \
  RUA_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && k >= 1, name ": bad arguments"); \
  RUA_CHECK_ARG(dtype == RUA_F32 || dtype == RUA_BF16, name ": bad dtype"); \
  const int vec = dtype == RUA_BF16 ? 8 : 4; \
  RUA_CHECK_ARG(C % vec == 0, name ": C=%d not a multiple of %d", C, vec); \
  const int CG = C / vec; hipStream_t st = (hipStream_t)stream;

extern "C" int rua_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int k, int dtype, void* stream) {
  POOL_ARGS_CHECK("rua_maxpool_fwd");
  RUA_CHECK_ARG(idx && H % k == 0 && W % k == 0 && k * k <= 256, "rua_maxpool_fwd: H,W must be divisible by k");
  RUA_CHECK_ARG(((size_t)idx & 7) == 0, "rua_maxpool_fwd: idx must be 8-byte aligned");
  const int g = grid_for((int64_t)N * (H / k) * (W / k) * CG);
#define RUA_MAXPOOL(TT, KK) hipLaunchKernelGGL((maxpool_fwd_kernel<TT, KK>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y, idx, N, H, W, CG, k)
  if (dtype == RUA_BF16) { if (k == 2) RUA_MAXPOOL(bf16_t, 2); else if (k == 4) RUA_MAXPOOL(bf16_t, 4); else if (k == 8) RUA_MAXPOOL(bf16_t, 8); else RUA_MAXPOOL(bf16_t, 0); }
  else { if (k == 2) RUA_MAXPOOL(float, 2); else if (k == 4) RUA_MAXPOOL(float, 4); else if (k == 8) RUA_MAXPOOL(float, 8); else RUA_MAXPOOL(float, 0); }
#undef RUA_MAXPOOL
  RUA_LAUNCH_CHECK("rua_maxpool_fwd");
  return RUA_OK;
}
extern "C" int rua_maxpool_bwd(const void* x, const uint8_t* idx, void* y, int accumulate, int N, int H, int W, int C, int k, int dtype, void* stream) {
  POOL_ARGS_CHECK("rua_maxpool_bwd");
  RUA_CHECK_ARG(idx && H % k == 0 && W % k == 0, "rua_maxpool_bwd: H,W must be divisible by k");
  RUA_CHECK_ARG(((size_t)idx & 7) == 0, "rua_maxpool_bwd: idx must be 8-byte aligned");
  const int g = grid_for((int64_t)N * H * W * CG);
  if (dtype == RUA_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, idx, (unsigned char*)y, accumulate, N, H, W, CG, k);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, idx, (unsigned char*)y, accumulate, N, H, W, CG, k);
  RUA_LAUNCH_CHECK("rua_maxpool_bwd");
  return RUA_OK;
}
extern "C" int rua_sumpool(const void* x, void* y, int N, int H, int W, int C, int k, int dtype, void* stream) {
  POOL_ARGS_CHECK("rua_sumpool");
  RUA_CHECK_ARG(H % k == 0 && W % k == 0, "rua_sumpool: H,W must be divisible by k");
  const int g = grid_for((int64_t)N * (H / k) * (W / k) * CG);
#define RUA_SUMPOOL(TT, KK) hipLaunchKernelGGL((sumpool_kernel<TT, KK>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y, N, H, W, CG, k)
  if (dtype == RUA_BF16) { if (k == 2) RUA_SUMPOOL(bf16_t, 2); else if (k == 4) RUA_SUMPOOL(bf16_t, 4); else if (k == 8) RUA_SUMPOOL(bf16_t, 8); else RUA_SUMPOOL(bf16_t, 0); }
  else { if (k == 2) RUA_SUMPOOL(float, 2); else if (k == 4) RUA_SUMPOOL(float, 4); else if (k == 8) RUA_SUMPOOL(float, 8); else RUA_SUMPOOL(float, 0); }
#undef RUA_SUMPOOL
  RUA_LAUNCH_CHECK("rua_sumpool");
  return RUA_OK;
}

extern "C" int rua_maxpool_derive(const void* y_half, const uint8_t* idx_half, void* y, uint8_t* idx, int N, int H_half, int W_half, int C,
                                  int k_half, int dtype, void* stream) {
  const void* x = y_half; const int H = H_half, W = W_half, k = k_half;
  POOL_ARGS_CHECK("rua_maxpool_derive");
  RUA_CHECK_ARG(idx_half && idx && H % 2 == 0 && W % 2 == 0 && 4 * k * k <= 256, "rua_maxpool_derive: even input grid, window 2k with 4 k^2 <= 256");
  RUA_CHECK_ARG((((size_t)idx | (size_t)idx_half) & 7) == 0, "rua_maxpool_derive: idx buffers must be 8-byte aligned");
  const int g = grid_for((long long)N * (H / 2) * (W / 2) * CG);
  if (dtype == RUA_BF16) hipLaunchKernelGGL((maxpool_derive_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const unsigned char*)y_half, idx_half, (unsigned char*)y, idx, N, H, W, CG, k);
  else hipLaunchKernelGGL((maxpool_derive_kernel<float>), dim3(g), dim3(256), 0, st, (const unsigned char*)y_half, idx_half, (unsigned char*)y, idx, N, H, W, CG, k);
  RUA_LAUNCH_CHECK("rua_maxpool_derive");
  return RUA_OK;
}

extern "C" int rua_maxpool_bwd_multi(int n, const void* const* dy, const uint8_t* const* idx, const int* ks, void* dx, int accumulate,
                                     int N, int H, int W, int C, int dtype, void* stream) {
  const void* x = dx; const void* y = dx; const int k = 1;
  POOL_ARGS_CHECK("rua_maxpool_bwd_multi");
  RUA_CHECK_ARG(n >= 1 && n <= 3 && dy && idx && ks, "rua_maxpool_bwd_multi: 1..3 pooled gradients");
  PoolPyr q;
  q.n = n;
  for (int u = 0; u < 3; ++u) {
    q.dy[u] = u < n ? (const unsigned char*)dy[u] : nullptr; q.idx[u] = u < n ? idx[u] : nullptr; q.k[u] = u < n ? ks[u] : 1;
    if (u < n) RUA_CHECK_ARG(dy[u] && idx[u] && ((size_t)idx[u] & 7) == 0 && ks[u] >= 1 && H % ks[u] == 0 && W % ks[u] == 0, "rua_maxpool_bwd_multi: member %d: H, W must be divisible by k, idx 8-byte aligned", u);
  }
  const int g = grid_for((long long)N * H * W * CG);
  auto lg2 = [](int v) { int s_ = 0; while ((1 << s_) < v) ++s_; return (1 << s_) == v ? s_ : -1; };
  const int cgs = lg2(CG), ws = lg2(W), hs = lg2(H);
  bool p2 = cgs >= 0 && ws >= 0 && hs >= 0 && (long long)N * H * W * CG < (1ll << 31);
  for (int u = 0; u < n; ++u) p2 = p2 && lg2(ks[u]) >= 0;
  if (p2) {
    if (dtype == RUA_BF16) hipLaunchKernelGGL((maxpool_bwd_multi_kernel<bf16_t, true>), dim3(g), dim3(256), 0, st, q, (unsigned char*)dx, accumulate, N, H, W, CG, cgs, ws, hs);
    else hipLaunchKernelGGL((maxpool_bwd_multi_kernel<float, true>), dim3(g), dim3(256), 0, st, q, (unsigned char*)dx, accumulate, N, H, W, CG, cgs, ws, hs);
  } else {
    if (dtype == RUA_BF16) hipLaunchKernelGGL((maxpool_bwd_multi_kernel<bf16_t, false>), dim3(g), dim3(256), 0, st, q, (unsigned char*)dx, accumulate, N, H, W, CG, 0, 0, 0);
    else hipLaunchKernelGGL((maxpool_bwd_multi_kernel<float, false>), dim3(g), dim3(256), 0, st, q, (unsigned char*)dx, accumulate, N, H, W, CG, 0, 0, 0);
  }
  RUA_LAUNCH_CHECK("rua_maxpool_bwd_multi");
  return RUA_OK;
}

extern "C" int rua_sumpool_pyramid(const void* x, void* y2, void* y4, void* y8, int N, int H, int W, int C, int dtype, void* stream) {
  const void* y = y2; const int k = 8;
  POOL_ARGS_CHECK("rua_sumpool_pyramid");
  RUA_CHECK_ARG(y4 && y8 && H % 8 == 0 && W % 8 == 0, "rua_sumpool_pyramid: H, W must be multiples of 8, every output given");
  const int g = grid_for((long long)N * (H / 8) * (W / 8) * CG);
  if (dtype == RUA_BF16) hipLaunchKernelGGL((sumpool_pyramid_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y2, (unsigned char*)y4, (unsigned char*)y8, N, H, W, CG);
  else hipLaunchKernelGGL((sumpool_pyramid_kernel<float>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y2, (unsigned char*)y4, (unsigned char*)y8, N, H, W, CG);
  RUA_LAUNCH_CHECK("rua_sumpool_pyramid");
  return RUA_OK;
}
extern "C" int rua_upsample_nearest(const void* x, void* y, int N, int H, int W, int C, int k, int dtype, void* stream) {
  POOL_ARGS_CHECK("rua_upsample_nearest");
  const int g = grid_for((int64_t)N * H * k * W * k * CG);
  if (dtype == RUA_BF16) hipLaunchKernelGGL((upsample_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y, N, H, W, CG, k);
  else hipLaunchKernelGGL((upsample_kernel<float>), dim3(g), dim3(256), 0, st, (const unsigned char*)x, (unsigned char*)y, N, H, W, CG, k);
  RUA_LAUNCH_CHECK("rua_upsample_nearest");
  return RUA_OK;
}

// ---------------------------------------------------------------------------------------
struct AddK { int n; const unsigned char* in[8]; unsigned char* out; int accumulate; long long pieces; };
template <typename T>
__global__ void add_n_kernel(const AddK p) {
  constexpr int VEC = ET<T>::VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.pieces; i += (long long)gridDim.x * blockDim.x) {
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    if (p.accumulate) ET<T>::unpack(ldg16(p.out + i * 16), s);
    for (int b = 0; b < p.n; ++b) {
      float v[VEC];
      ET<T>::unpack(ldg16(p.in[b] + i * 16), v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += v[j];
    }
    stg16(p.out + i * 16, ET<T>::pack(s));
  }
}
extern "C" int rua_add_n(int n, const void* const* in, void* out, int accumulate, int64_t elems, int dtype, void* stream) {
  RUA_CHECK_ARG(n >= 1 && n <= 8 && in && out && elems > 0, "rua_add_n: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(elems % vec == 0, "rua_add_n: element count must be a multiple of %d", vec);
  AddK k; k.n = n; k.out = (unsigned char*)out; k.accumulate = accumulate; k.pieces = elems / vec;
  for (int b = 0; b < n; ++b) { RUA_CHECK_ARG(in[b], "rua_add_n: null input"); k.in[b] = (const unsigned char*)in[b]; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((add_n_kernel<bf16_t>), dim3(grid_for(k.pieces)), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((add_n_kernel<float>), dim3(grid_for(k.pieces)), dim3(256), 0, st, k);
  RUA_LAUNCH_CHECK("rua_add_n");
  return RUA_OK;
}

template <typename T, int MODE>   // 0: dy *= (y>0) in place ; 1: out = relu(x)
__global__ void relu_kernel(unsigned char* a, const unsigned char* b, long long pieces) {
  constexpr int VEC = ET<T>::VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (long long)gridDim.x * blockDim.x) {
    float u[VEC], v[VEC];
    ET<T>::unpack(ldg16(b + i * 16), v);
    if (MODE == 0) {
      ET<T>::unpack(ldg16(a + i * 16), u);
#pragma unroll
      for (int j = 0; j < VEC; ++j) u[j] = v[j] > 0.f ? u[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) u[j] = fmaxf(v[j], 0.f);
    }
    stg16(a + i * 16, ET<T>::pack(u));
  }
}
extern "C" int rua_relu_mask(void* dy, const void* y, int64_t elems, int dtype, void* stream) {
  RUA_CHECK_ARG(dy && y && elems > 0, "rua_relu_mask: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(elems % vec == 0, "rua_relu_mask: element count must be a multiple of %d", vec);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((relu_kernel<bf16_t, 0>), dim3(grid_for(elems / vec)), dim3(256), 0, st, (unsigned char*)dy, (const unsigned char*)y, (long long)(elems / vec));
  else hipLaunchKernelGGL((relu_kernel<float, 0>), dim3(grid_for(elems / vec)), dim3(256), 0, st, (unsigned char*)dy, (const unsigned char*)y, (long long)(elems / vec));
  RUA_LAUNCH_CHECK("rua_relu_mask");
  return RUA_OK;
}
extern "C" int rua_relu(const void* x, void* y, int64_t elems, int dtype, void* stream) {
  RUA_CHECK_ARG(x && y && elems > 0, "rua_relu: bad arguments");
  const int vec = dtype == RUA_BF16 ? 8 : 4;
  RUA_CHECK_ARG(elems % vec == 0, "rua_relu: element count must be a multiple of %d", vec);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) hipLaunchKernelGGL((relu_kernel<bf16_t, 1>), dim3(grid_for(elems / vec)), dim3(256), 0, st, (unsigned char*)y, (const unsigned char*)x, (long long)(elems / vec));
  else hipLaunchKernelGGL((relu_kernel<float, 1>), dim3(grid_for(elems / vec)), dim3(256), 0, st, (unsigned char*)y, (const unsigned char*)x, (long long)(elems / vec));
  RUA_LAUNCH_CHECK("rua_relu");
  return RUA_OK;
}

__global__ void cast_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = (bf16_t)x[i];
}
__global__ void cast_from_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
extern "C" int rua_cast_f32_to(const float* x, void* y, int64_t elems, int dtype, void* stream) {
  RUA_CHECK_ARG(x && y && elems > 0, "rua_cast_f32_to: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) {
    hipLaunchKernelGGL(cast_to_bf16_kernel, dim3(grid_for(elems)), dim3(256), 0, st, x, (bf16_t*)y, (long long)elems);
    RUA_LAUNCH_CHECK("rua_cast_f32_to");
    return RUA_OK;
  }
  hipError_t e = hipMemcpyAsync(y, x, elems * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { rua_set_error("rua_cast_f32_to: %s", hipGetErrorString(e)); return RUA_ERR_LAUNCH; }
  return RUA_OK;
}
extern "C" int rua_cast_to_f32(const void* x, float* y, int64_t elems, int dtype, void* stream) {
  RUA_CHECK_ARG(x && y && elems > 0, "rua_cast_to_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RUA_BF16) {
    hipLaunchKernelGGL(cast_from_bf16_kernel, dim3(grid_for(elems)), dim3(256), 0, st, (const bf16_t*)x, y, (long long)elems);
    RUA_LAUNCH_CHECK("rua_cast_to_f32");
    return RUA_OK;
  }
  hipError_t e = hipMemcpyAsync(y, x, elems * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { rua_set_error("rua_cast_to_f32: %s", hipGetErrorString(e)); return RUA_ERR_LAUNCH; }
  return RUA_OK;
}
// zero fill as an ordinary kernel (16-byte stores; head / tail bytes one by one): see rua_fill_zero
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* p, long long bytes) {
  const long long head = ((16 - ((size_t)p & 15)) & 15) < bytes ? ((16 - ((size_t)p & 15)) & 15) : bytes;
  const long long n16 = (bytes - head) / 16;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) q[i] = z;
  if (blockIdx.x == 0) {
    for (long long i = threadIdx.x; i < head; i += blockDim.x) p[i] = 0;
    for (long long i = head + n16 * 16 + threadIdx.x; i < bytes; i += blockDim.x) p[i] = 0;
  }
}
extern "C" int rua_fill_zero(void* p, int64_t bytes, void* stream) {
  RUA_CHECK_ARG(p && bytes >= 0, "rua_fill_zero: bad arguments");
  if (bytes == 0) return RUA_OK;
  if (g_tune.fill_kernel) {
    // A KERNEL, not hipMemsetAsync: captured into a HIP graph a memset becomes a memset node, and graphs that hold memset nodes,
    // replayed back to back with only stream-event operations in between, corrupted the piecewise data-parallel step on this stack
    // (tools/dp_graph_check.py)
    long long blocks = (bytes / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4 * rua_cu_count()) blocks = 4 * rua_cu_count();
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, (long long)bytes);
    RUA_LAUNCH_CHECK("rua_fill_zero");
    return RUA_OK;
  }
  hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) { rua_set_error("rua_fill_zero: %s", hipGetErrorString(e)); return RUA_ERR_LAUNCH; }
  return RUA_OK;
}

// Loss heads, metrics and optimizers.  Everything here is fp32 on [B][HW][C] probabilities and
// labels with fp64 accumulators; per-pixel work is HBM-bound, reductions are wavefront
// shuffles + one fp64 atomic per block and value.
#include "common.h"

__device__ __forceinline__ void block_atomic_add(double* dst, float v, float* sh /*[4]*/, int slot) {
  // every thread calls; wave shuffle -> LDS -> one fp64 atomic
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sh[slot * 4 + wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(dst, (double)sh[slot * 4 + 0] + (double)sh[slot * 4 + 1] + (double)sh[slot * 4 + 2] + (double)sh[slot * 4 + 3]);
  __syncthreads();
}

// ---- Tanimoto with complement (multitasking_utils.py:38-85) ------------------------------
// sums[n][c][6] = { sum p, sum (1-l), sum p*l, sum p^2+l^2, sum (1-p)(1-l), sum (1-p)^2+(1-l)^2 }
__global__ __launch_bounds__(256) void tanimoto_sums_kernel(const float* __restrict__ p, const float* __restrict__ y, long long HW, int C,
                                                            int pix_per_block, double* sums) {
  __shared__ float sh[4 * 48];
  const int n = blockIdx.y;
  float acc[8][6];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[c][k] = 0.f;
  long long i = (long long)blockIdx.x * pix_per_block + threadIdx.x;
  long long iend = (long long)(blockIdx.x + 1) * pix_per_block; if (iend > HW) iend = HW;
  for (; i < iend; i += 256) {
    const float* pp = p + ((size_t)n * HW + i) * C;
    const float* yy = y + ((size_t)n * HW + i) * C;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < C) {
        const float a = pp[c], l = yy[c], q = 1.f - a, m = 1.f - l;
        acc[c][0] += a; acc[c][1] += m; acc[c][2] = fmaf(a, l, acc[c][2]);
        acc[c][3] += a * a + l * l; acc[c][4] = fmaf(q, m, acc[c][4]); acc[c][5] += q * q + m * m;
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float v = wave_sum(acc[c][k]);
      if (lane == 0) sh[wid * 48 + c * 6 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < 48) {
    const int c = threadIdx.x / 6, k = threadIdx.x % 6;
    if (c < C) {
      const double t = (double)sh[threadIdx.x] + (double)sh[48 + threadIdx.x] + (double)sh[96 + threadIdx.x] + (double)sh[144 + threadIdx.x];
      unsafeAtomicAdd(&sums[((size_t)n * C + c) * 6 + k], t);
    }
  }
}

// Same sums with 16-byte loads for the class counts the reference's heads use: a thread takes G pixels (G * C floats = NV
// float4) per iteration from each tensor, all loads issued before the first use; the class of every element is a
// compile-time constant.  (The scalar kernel above streams at ~1.5 TB/s.)
template <int C, int G>
__global__ __launch_bounds__(256) void tanimoto_sums_vec(const float* __restrict__ p, const float* __restrict__ y, long long HW,
                                                         int groups_per_block, double* sums) {
  constexpr int NV = G * C / 4;
  static_assert(G * C % 4 == 0, "group must be whole float4s");
  __shared__ float sh[4 * 48];
  const int n = blockIdx.y;
  float acc[C][6];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[c][k] = 0.f;
  const long long ngroups = HW / G;
  long long gi = (long long)blockIdx.x * groups_per_block + threadIdx.x;
  long long gend = (long long)(blockIdx.x + 1) * groups_per_block; if (gend > ngroups) gend = ngroups;
  const float4* pp = reinterpret_cast<const float4*>(p + (size_t)n * HW * C);
  const float4* yy = reinterpret_cast<const float4*>(y + (size_t)n * HW * C);
  for (; gi < gend; gi += 256) {
    float4 a4[NV], l4[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { a4[v] = pp[gi * NV + v]; l4[v] = yy[gi * NV + v]; }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float av[4] = {a4[v].x, a4[v].y, a4[v].z, a4[v].w}, lv[4] = {l4[v].x, l4[v].y, l4[v].z, l4[v].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (v * 4 + j) % C;
        const float a = av[j], l = lv[j], q = 1.f - a, m = 1.f - l;
        acc[c][0] += a; acc[c][1] += m; acc[c][2] = fmaf(a, l, acc[c][2]);
        acc[c][3] += a * a + l * l; acc[c][4] = fmaf(q, m, acc[c][4]); acc[c][5] += q * q + m * m;
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float v = wave_sum(acc[c][k]);
      if (lane == 0) sh[wid * 48 + c * 6 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * 6) {
    const double t = (double)sh[threadIdx.x] + (double)sh[48 + threadIdx.x] + (double)sh[96 + threadIdx.x] + (double)sh[144 + threadIdx.x];
    unsafeAtomicAdd(&sums[(size_t)n * C * 6 + threadIdx.x], t);
  }
}

template <int C, int G>
static void launch_tanimoto_vec(const float* p, const float* y, int B, int64_t HW, double* sums, hipStream_t st) {
  const int64_t ngroups = HW / G;
  int64_t gpb = (ngroups + 63) / 64; if (gpb < 256) gpb = 256;         // ~64 blocks per sample
  gpb = (gpb + 255) / 256 * 256;
  const int gx = (int)((ngroups + gpb - 1) / gpb);
  hipLaunchKernelGGL((tanimoto_sums_vec<C, G>), dim3(gx, B), dim3(256), 0, st, p, y, (long long)HW, (int)gpb, sums);
}

extern "C" int rua_tanimoto_sums(const float* p, const float* y, int B, int64_t HW, int C, double* sums, void* stream) {
  RUA_CHECK_ARG(p && y && sums && B > 0 && HW > 0, "rua_tanimoto_sums: bad arguments");
  RUA_CHECK_ARG(C >= 1 && C <= 8, "rua_tanimoto_sums: C=%d must be in 1..8", C);
  const int vec_on = g_tune.tani_vec;
  if (vec_on && HW % 4 == 0 && ((size_t)p & 15) == 0 && ((size_t)y & 15) == 0 && (C == 6 || C == 3 || C == 2)) {
    if (C == 6) launch_tanimoto_vec<6, 2>(p, y, B, HW, sums, (hipStream_t)stream);
    else if (C == 3) launch_tanimoto_vec<3, 4>(p, y, B, HW, sums, (hipStream_t)stream);
    else launch_tanimoto_vec<2, 2>(p, y, B, HW, sums, (hipStream_t)stream);
    RUA_LAUNCH_CHECK("rua_tanimoto_sums");
    return RUA_OK;
  }
  int64_t ppb = (HW + 127) / 128; if (ppb < 1024) ppb = 1024;      // measured: 64..256 blocks per sample all ~17 us, 16: 21, 8: 35
  ppb = (ppb + 255) / 256 * 256;
  const int gx = (int)((HW + ppb - 1) / ppb);
  hipLaunchKernelGGL(tanimoto_sums_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, p, y, (long long)HW, C, (int)ppb, sums);
  RUA_LAUNCH_CHECK("rua_tanimoto_sums");
  return RUA_OK;
}

// One block.  Follows Tanimoto_dual_loss: loss1 = T(label:=pred, pred:=label) so the class weights of
// the first term come from the PREDICTION volumes (and carry gradient); loss2 = T(1-label, 1-pred).
// single != 0: Tanimoto_loss(label, pred) itself (multitasking_utils.py:38-68) for sums taken with p := label, y := pred -
// the first term's ratio (N1 + 1e-5) / (D1 + 1e-5) with the class weights from the volumes of the FIRST argument.
__device__ __forceinline__ void tanimoto_finalize_body(const double* sums, int B, int C, float grad_scale, double* loss_out, float* coef, float* per_sample,
                                                       int single, int replicas) {
  __shared__ double w1[8], w2[8], v1[8], kap[8];
  __shared__ double E1[256], F1[256], E2[256], F2[256];
  __shared__ int inf1[8];
  const int t = threadIdx.x;
  const double smooth = 1e-5;
  if (replicas > 1) {
    // sums[replicas + 1][B][C][6]: the producers added into `replicas` copies (rua_head_fwd_loss_rep: fewer same-address atomics in a row); their sum,
    // in replica order, goes into the extra slot behind them by plain stores (idempotent: calling this twice gives the same result) and is what the
    // rest of the kernel reads
    const int ne = B * C * 6;
    double* folded = const_cast<double*>(sums) + (size_t)replicas * ne;
    for (int i = t; i < ne; i += blockDim.x) {
      double a = 0;
      for (int r = 0; r < replicas; ++r) a += sums[(size_t)r * ne + i];
      folded[i] = a;
    }
    __threadfence_block();
    __syncthreads();
    sums = folded;
  }
  if (t < C) {
    double a = 0, b = 0;
    for (int n = 0; n < B; ++n) { a += sums[((size_t)n * C + t) * 6 + 0]; b += sums[((size_t)n * C + t) * 6 + 1]; }
    a /= B; b /= B;
    // the reference squares and takes the reciprocal in float32: overflow to inf happens only at V == 0
    v1[t] = a;
    const float a2 = (float)a * (float)a, b2 = (float)b * (float)b;
    w1[t] = a2 == 0.f ? INFINITY : 1.0 / (a * a);
    w2[t] = b2 == 0.f ? INFINITY : 1.0 / (b * b);
  }
  __syncthreads();
  if (t == 0) {
    double m1 = 0, m2 = 0;
    for (int c = 0; c < C; ++c) { if (!isinf(w1[c]) && w1[c] > m1) m1 = w1[c]; if (!isinf(w2[c]) && w2[c] > m2) m2 = w2[c]; }
    for (int c = 0; c < C; ++c) {
      inf1[c] = isinf(w1[c]);
      if (isinf(w1[c])) w1[c] = m1;
      if (isinf(w2[c])) w2[c] = m2;
    }
  }
  __syncthreads();
  double lsum = 0;
  for (int n = t; n < B; n += blockDim.x) {
    double N1 = 0, D1 = 0, N2 = 0, D2 = 0;
    for (int c = 0; c < C; ++c) {
      const double* s = &sums[((size_t)n * C + c) * 6];
      N1 += w1[c] * s[2]; D1 += w1[c] * (s[3] - s[2]);
      N2 += w2[c] * s[4]; D2 += w2[c] * (s[5] - s[4]);
    }
    E1[n] = D1 + smooth; F1[n] = N1 + smooth; E2[n] = D2 + smooth; F2[n] = N2 + smooth;
    const double ln = single ? F1[n] / E1[n] : 1.0 - 0.5 * (F1[n] / E1[n] + F2[n] / E2[n]);
    if (per_sample) per_sample[n] = (float)ln;
    lsum += ln;
  }
  // block sum of lsum (B <= 256, blockDim 64: one wave)
  for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (t == 0) loss_out[0] = lsum / B;
  __syncthreads();
  if (t < C) {
    // d loss1[n] / d w1_c = [Spl (D1+e) - (N1+e)(SS - Spl)] / (D1+e)^2 ; d w1_c / d p = -2 / (V^3 B)
    double R = 0;
    for (int n = 0; n < B; ++n) {
      const double* s = &sums[((size_t)n * C + t) * 6];
      R += (s[2] * E1[n] - F1[n] * (s[3] - s[2])) / (E1[n] * E1[n]);
    }
    kap[t] = inf1[t] ? 0.0 : (-2.0 / (v1[t] * v1[t] * v1[t] * B)) * R;
  }
  __syncthreads();
  const double sc = -0.5 * (double)grad_scale;      // d(1 - .5(l1+l2)); grad_scale = loss_weight / B
  if (coef)
  for (int i = t; i < B * C; i += blockDim.x) {
    const int n = i / C, c = i - n * C;
    const double a1 = w1[c] / (E1[n] * E1[n]), a2 = w2[c] / (E2[n] * E2[n]);
    const double c0 = kap[c] + a2 * (F2[n] - E2[n]);
    const double c1 = -2.0 * a1 * F1[n] - 2.0 * a2 * F2[n];
    const double c2 = a1 * (E1[n] + F1[n]) + a2 * (E2[n] + F2[n]);
    coef[i * 3 + 0] = (float)(sc * c0); coef[i * 3 + 1] = (float)(sc * c1); coef[i * 3 + 2] = (float)(sc * c2);
  }
}
__global__ void tanimoto_finalize_kernel(const double* sums, int B, int C, float grad_scale, double* loss_out, float* coef, float* per_sample,
                                         int single, int replicas) {
  tanimoto_finalize_body(sums, B, C, grad_scale, loss_out, coef, per_sample, single, replicas);
}
// the heads of a multitask model in ONE launch (a block per head): four one-block launches of ~5 us each were ~20 us of a 7 ms step
struct TaniMulti { rua_tani_head h[RUA_MAX_HEADS]; };
__global__ void tanimoto_finalize_multi_kernel(const TaniMulti a) {
  const rua_tani_head& h = a.h[blockIdx.x];
  tanimoto_finalize_body(h.sums, h.B, h.C, h.grad_scale, h.loss_out, h.coef, h.per_sample, 0, h.replicas);
}
extern "C" int rua_tanimoto_finalize_multi(const rua_tani_head* heads, int n, void* stream) {
  RUA_CHECK_ARG(heads && n >= 1 && n <= RUA_MAX_HEADS, "rua_tanimoto_finalize_multi: 1..%d heads", RUA_MAX_HEADS);
  TaniMulti a;
  for (int i = 0; i < n; ++i) {
    const rua_tani_head& h = heads[i];
    RUA_CHECK_ARG(h.sums && h.loss_out && h.B > 0 && h.B <= 256 && h.C >= 1 && h.C <= 8 && h.replicas >= 1 && h.replicas <= 64,
                  "rua_tanimoto_finalize_multi: head %d: B must be in 1..256, C in 1..8, replicas in 1..64", i);
    a.h[i] = h;
  }
  hipLaunchKernelGGL(tanimoto_finalize_multi_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, a);
  RUA_LAUNCH_CHECK("rua_tanimoto_finalize_multi");
  return RUA_OK;
}

extern "C" int rua_tanimoto_finalize_rep(double* sums, int replicas, int B, int64_t HW, int C, float grad_scale, double* loss_out, float* coef,
                                         float* per_sample, void* stream) {
  RUA_CHECK_ARG(sums && loss_out && B > 0 && B <= 256, "rua_tanimoto_finalize: B must be in 1..256");
  RUA_CHECK_ARG(C >= 1 && C <= 8, "rua_tanimoto_finalize: C=%d must be in 1..8", C);
  RUA_CHECK_ARG(replicas >= 1 && replicas <= 64, "rua_tanimoto_finalize_rep: replicas=%d must be in 1..64", replicas);
  (void)HW;
  hipLaunchKernelGGL(tanimoto_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)sums, B, C, grad_scale, loss_out, coef, per_sample, 0, replicas);
  RUA_LAUNCH_CHECK("rua_tanimoto_finalize");
  return RUA_OK;
}
extern "C" int rua_tanimoto_finalize(const double* sums, int B, int64_t HW, int C, float grad_scale, double* loss_out, float* coef,
                                     float* per_sample, void* stream) {
  return rua_tanimoto_finalize_rep(const_cast<double*>(sums), 1, B, HW, C, grad_scale, loss_out, coef, per_sample, stream);      // one replica: sums is only read
}

extern "C" int rua_tanimoto_ratio(const double* sums, int B, int C, double* mean_out, float* per_sample, void* stream) {
  RUA_CHECK_ARG(sums && mean_out && per_sample && B > 0 && B <= 256, "rua_tanimoto_ratio: B must be in 1..256");
  RUA_CHECK_ARG(C >= 1 && C <= 8, "rua_tanimoto_ratio: C=%d must be in 1..8", C);
  hipLaunchKernelGGL(tanimoto_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, B, C, 0.f, mean_out, (float*)nullptr, per_sample, 1, 1);
  RUA_LAUNCH_CHECK("rua_tanimoto_ratio");
  return RUA_OK;
}

// ---- per-pixel losses (utils.py:481-490; Keras CategoricalCrossentropy / BinaryCrossentropy /
// MeanSquaredError as train_ISPRS.py:411-428 selects them) -----------------------------------
#define KERAS_EPS 1e-7f
__global__ __launch_bounds__(256) void pixel_loss_kernel(int kind, const float* __restrict__ p, const float* __restrict__ z,
                                                         const float* __restrict__ y, const float* __restrict__ cw,
                                                         long long M, int C, double* out, float* per_pixel) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    float l = 0.f;
    if (kind == RUA_LOSS_WCE) {
      float S = 0.f;
      for (int c = 0; c < C; ++c) S += p[m * C + c];
      for (int c = 0; c < C; ++c) {
        const float u = fminf(fmaxf(p[m * C + c] / S, KERAS_EPS), 1.f - KERAS_EPS);
        l -= y[m * C + c] * logf(u) * cw[c];
      }
    } else if (kind == RUA_LOSS_CE_LOGITS) {
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[m * C + c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(z[m * C + c] - mx);
      const float lse = mx + logf(s);
      for (int c = 0; c < C; ++c) l -= y[m * C + c] * (z[m * C + c] - lse);
    } else if (kind == RUA_LOSS_BCE_LOGITS) {
      for (int c = 0; c < C; ++c) {
        const float zz = z[m * C + c];
        l += fmaxf(zz, 0.f) - zz * y[m * C + c] + log1pf(expf(-fabsf(zz)));
      }
      l /= C;
    } else {
      for (int c = 0; c < C; ++c) { const float d = y[m * C + c] - p[m * C + c]; l = fmaf(d, d, l); }
      l /= C;
    }
    if (per_pixel) per_pixel[m] = l;
    acc += l;
  }
  block_atomic_add(out, acc, sh, 0);
}

extern "C" int rua_pixel_loss(int kind, const float* p, const float* z, const float* y, const float* class_w,
                              int64_t M, int C, double* loss_out, float* per_pixel, void* stream) {
  RUA_CHECK_ARG(p && y && loss_out && M > 0 && C >= 1 && C <= 64, "rua_pixel_loss: bad arguments");
  RUA_CHECK_ARG(kind >= RUA_LOSS_WCE && kind <= RUA_LOSS_MSE, "rua_pixel_loss: kind %d is not a per-pixel loss", kind);
  RUA_CHECK_ARG(kind != RUA_LOSS_WCE || class_w, "rua_pixel_loss: weighted CE needs class weights");
  RUA_CHECK_ARG((kind != RUA_LOSS_CE_LOGITS && kind != RUA_LOSS_BCE_LOGITS) || z, "rua_pixel_loss: logits needed");
  int64_t g = (M + 255) / 256; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(pixel_loss_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, kind, p, z, y, class_w, (long long)M, C, loss_out, per_pixel);
  RUA_LAUNCH_CHECK("rua_pixel_loss");
  return RUA_OK;
}

// d(total loss)/d(logits), one thread per pixel
// CT: the class count as a template constant (6, 3; 0: runtime, <= 8) - with a runtime C every `c < C` became a branch around one scalar access
template <int CT>
__device__ __forceinline__ void head_dz_body(int kind, int act, const float* __restrict__ p, const float* __restrict__ y,
                                             const float* __restrict__ coef, const float* __restrict__ cw, float gs,
                                             long long HW, int C_rt, float* dz) {
  const int C = CT > 0 ? CT : C_rt;
  auto load_row = [&](const float* src, long long m, float* v) {
    if constexpr (CT == 6) {                              // 24-byte rows: three 8-byte accesses
      const float2* q = reinterpret_cast<const float2*>(src + m * 6);
      const float2 a = q[0], b = q[1], c = q[2];
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = 0.f; v[7] = 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c < C ? src[m * C + c] : 0.f;
    }
  };
  auto store_row = [&](long long m, const float* v) {
    if constexpr (CT == 6) {
      float2* q = reinterpret_cast<float2*>(dz + m * 6);
      q[0] = make_float2(v[0], v[1]); q[1] = make_float2(v[2], v[3]); q[2] = make_float2(v[4], v[5]);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < C) dz[m * C + c] = v[c];
    }
  };
  // blockIdx.y = sample (the Tanimoto coefficients are per sample and class: m / HW per pixel was a 64-bit division - ~100 vector instructions
  // in a pass that moves 72 bytes per pixel)
  const int n = blockIdx.y;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
    const long long m = (long long)n * HW + i;
    float pv[8], yv[8], g[8], o[8];
    load_row(p, m, pv); load_row(y, m, yv);
#pragma unroll
    for (int c = 0; c < 8; ++c) { g[c] = 0.f; o[c] = 0.f; }
    if (kind == RUA_LOSS_CE_LOGITS) {
      float sy = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) sy += yv[c];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = (pv[c] * sy - yv[c]) * gs;
      store_row(m, o);
      continue;
    }
    if (kind == RUA_LOSS_BCE_LOGITS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = (pv[c] - yv[c]) * gs / C;
      store_row(m, o);
      continue;
    }
    if (kind == RUA_LOSS_TANIMOTO) {
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < C) { const float* k = coef + ((size_t)n * C + c) * 3; g[c] = fmaf(k[1], pv[c], fmaf(k[2], yv[c], k[0])); }
    } else if (kind == RUA_LOSS_WCE) {
      float S = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) S += pv[c];
      float h[8], hu = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        h[c] = 0.f;
        if (c < C) {
          const float u = pv[c] / S;
          const bool inside = u >= KERAS_EPS && u <= 1.f - KERAS_EPS;     // clip_by_value passes gradient inside only
          h[c] = inside ? -yv[c] * cw[c] / u : 0.f;
          hu = fmaf(h[c], u, hu);
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = (h[c] - hu) / S * gs;
    } else {   // MSE
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = 2.f * (pv[c] - yv[c]) * gs / C;
    }
    if (act == RUA_ACT_SOFTMAX) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) dot = fmaf(g[c], pv[c], dot);
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = pv[c] * (g[c] - dot);
      store_row(m, o);
    } else if (act == RUA_ACT_SIGMOID) {
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = g[c] * pv[c] * (1.f - pv[c]);
      store_row(m, o);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = g[c];
      store_row(m, o);
    }
  }
}

__global__ __launch_bounds__(256) void head_dz_kernel(int kind, int act, const float* __restrict__ p, const float* __restrict__ y,
                                                      const float* __restrict__ coef, const float* __restrict__ cw, float gs,
                                                      long long HW, long long M, int C, float* dz) {
  (void)M;
  if (C == 6) head_dz_body<6>(kind, act, p, y, coef, cw, gs, HW, C, dz);
  else if (C == 3) head_dz_body<3>(kind, act, p, y, coef, cw, gs, HW, C, dz);
  else head_dz_body<0>(kind, act, p, y, coef, cw, gs, HW, C, dz);
}
// d(loss)/d(logits) of several heads in ONE launch: blockIdx.z = head (the four heads of the multitask model: four launches of 4 - 12 us)
struct DzMulti { rua_dz_head h[RUA_MAX_HEADS]; };
__global__ __launch_bounds__(256) void head_dz_multi_kernel(const DzMulti a) {
  const rua_dz_head& h = a.h[blockIdx.z];
  if ((int)blockIdx.y >= h.B) return;
  if (h.C == 6) head_dz_body<6>(h.kind, h.act, h.p, h.y, h.coef, h.class_w, h.grad_scale, (long long)h.HW, h.C, h.dz);
  else if (h.C == 3) head_dz_body<3>(h.kind, h.act, h.p, h.y, h.coef, h.class_w, h.grad_scale, (long long)h.HW, h.C, h.dz);
  else head_dz_body<0>(h.kind, h.act, h.p, h.y, h.coef, h.class_w, h.grad_scale, (long long)h.HW, h.C, h.dz);
}
extern "C" int rua_head_dz_multi(const rua_dz_head* heads, int n, void* stream) {
  RUA_CHECK_ARG(heads && n >= 1 && n <= RUA_MAX_HEADS, "rua_head_dz_multi: 1..%d heads", RUA_MAX_HEADS);
  DzMulti a;
  int64_t g = 1; int Bm = 1;
  for (int i = 0; i < n; ++i) {
    const rua_dz_head& h = heads[i];
    RUA_CHECK_ARG(h.p && h.y && h.dz && h.B > 0 && h.B <= 65535 && h.HW > 0 && h.C >= 1 && h.C <= 8 && h.kind >= 0 && h.kind <= 4 && h.act >= 0 && h.act <= 2,
                  "rua_head_dz_multi: head %d: bad arguments", i);
    RUA_CHECK_ARG(h.kind != RUA_LOSS_TANIMOTO || h.coef, "rua_head_dz_multi: Tanimoto needs coefficients");
    RUA_CHECK_ARG(h.kind != RUA_LOSS_WCE || h.class_w, "rua_head_dz_multi: weighted CE needs class weights");
    a.h[i] = h;
    const int64_t gi = (h.HW + 255) / 256; if (gi > g) g = gi;
    if (h.B > Bm) Bm = h.B;
  }
  const int64_t cap = 4096 / ((int64_t)Bm * n) > 0 ? 4096 / ((int64_t)Bm * n) : 1;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(head_dz_multi_kernel, dim3((int)g, Bm, n), dim3(256), 0, (hipStream_t)stream, a);
  RUA_LAUNCH_CHECK("rua_head_dz_multi");
  return RUA_OK;
}

extern "C" int rua_head_dz(int kind, int act, const float* p, const float* y, const float* coef, const float* class_w,
                           float grad_scale, int B, int64_t HW, int C, float* dz, void* stream) {
  RUA_CHECK_ARG(p && y && dz && B > 0 && HW > 0, "rua_head_dz: bad arguments");
  RUA_CHECK_ARG(C >= 1 && C <= 8, "rua_head_dz: C=%d must be in 1..8", C);
  RUA_CHECK_ARG(kind >= 0 && kind <= 4 && act >= 0 && act <= 2, "rua_head_dz: bad kind/act");
  RUA_CHECK_ARG(kind != RUA_LOSS_TANIMOTO || coef, "rua_head_dz: Tanimoto needs coefficients");
  RUA_CHECK_ARG(kind != RUA_LOSS_WCE || class_w, "rua_head_dz: weighted CE needs class weights");
  const int64_t M = (int64_t)B * HW;
  int64_t g = (HW + 255) / 256;
  const int64_t cap = 4096 / B > 0 ? 4096 / B : 1;       // ~4096 blocks over the batch; blockIdx.y = sample
  if (g > cap) g = cap;
  RUA_CHECK_ARG(B <= 65535, "rua_head_dz: B=%d too large", B);
  hipLaunchKernelGGL(head_dz_kernel, dim3((int)g, B), dim3(256), 0, (hipStream_t)stream, kind, act, p, y, coef, class_w, grad_scale,
                     (long long)HW, (long long)M, C, dz);
  RUA_LAUNCH_CHECK("rua_head_dz");
  return RUA_OK;
}

// accuracy + TP/FP/TN/FN (Keras 'accuracy' = categorical accuracy; confusion counts at .5)
__global__ __launch_bounds__(256) void seg_metrics_kernel(const float* __restrict__ p, const float* __restrict__ y, long long M, int C, double* out) {
  __shared__ float sh[5 * 4];
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    int ip = 0, iy = 0; float bp = p[m * C], by = y[m * C];
    for (int c = 0; c < C; ++c) {
      const float pv = p[m * C + c], yv = y[m * C + c];
      if (pv > bp) { bp = pv; ip = c; }
      if (yv > by) { by = yv; iy = c; }
      const bool t = yv > 0.5f, q = pv > 0.5f;
      a[1] += (t && q); a[2] += (!t && q); a[3] += (!t && !q); a[4] += (t && !q);
    }
    a[0] += (ip == iy);
  }
  for (int k = 0; k < 5; ++k) block_atomic_add(&out[k], a[k], sh, k);
}

// 16-byte loads: G pixels (G * C floats = NV float4) per thread per iteration, element classes are compile-time constants
template <int C, int G>
__global__ __launch_bounds__(256) void seg_metrics_vec(const float* __restrict__ p, const float* __restrict__ y, long long M, double* out) {
  constexpr int NV = G * C / 4;
  __shared__ float sh[5 * 4];
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const float4* pp = reinterpret_cast<const float4*>(p);
  const float4* yy = reinterpret_cast<const float4*>(y);
  const long long ngroups = M / G;
  for (long long gi = (long long)blockIdx.x * 256 + threadIdx.x; gi < ngroups; gi += (long long)gridDim.x * 256) {
    float pv[NV * 4], yv[NV * 4];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float4 a4 = pp[gi * NV + v], l4 = yy[gi * NV + v];
      pv[v * 4] = a4.x; pv[v * 4 + 1] = a4.y; pv[v * 4 + 2] = a4.z; pv[v * 4 + 3] = a4.w;
      yv[v * 4] = l4.x; yv[v * 4 + 1] = l4.y; yv[v * 4 + 2] = l4.z; yv[v * 4 + 3] = l4.w;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int ip = 0, iy = 0; float bp = pv[g * C], by = yv[g * C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float q_ = pv[g * C + c], t_ = yv[g * C + c];
        if (q_ > bp) { bp = q_; ip = c; }
        if (t_ > by) { by = t_; iy = c; }
        const bool t = t_ > 0.5f, q = q_ > 0.5f;
        a[1] += (t && q); a[2] += (!t && q); a[3] += (!t && !q); a[4] += (t && !q);
      }
      a[0] += (ip == iy);
    }
  }
  for (int k = 0; k < 5; ++k) block_atomic_add(&out[k], a[k], sh, k);
}

extern "C" int rua_seg_metrics(const float* p, const float* y, int64_t M, int C, double* out, void* stream) {
  RUA_CHECK_ARG(p && y && out && M > 0 && C >= 1, "rua_seg_metrics: bad arguments");
  if ((C == 6 || C == 2) && M % 2 == 0 && ((size_t)p & 15) == 0 && ((size_t)y & 15) == 0) {
    int64_t g = (M / 2 + 255) / 256; if (g > 256) g = 256;
    if (C == 6) hipLaunchKernelGGL((seg_metrics_vec<6, 2>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, p, y, (long long)M, out);
    else hipLaunchKernelGGL((seg_metrics_vec<2, 2>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, p, y, (long long)M, out);
    RUA_LAUNCH_CHECK("rua_seg_metrics");
    return RUA_OK;
  }
  // the five counters share one cache line and every block ends in five same-line fp64 atomics (serialised): measured
  // 64 blocks 36 us, 128: 23, 256: 24, 512: 37
  const int cap = g_tune.metrics_blocks > 0 ? g_tune.metrics_blocks : rua_cu_count() / 2;      // 128 on MI355X
  int64_t g = (M + 255) / 256; if (g > cap) g = cap;
  hipLaunchKernelGGL(seg_metrics_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, p, y, (long long)M, C, out);
  RUA_LAUNCH_CHECK("rua_seg_metrics");
  return RUA_OK;
}

// ---- optimizers on the flat parameter buffer ---------------------------------------------------
// Four parameters per thread (16-byte accesses; every slice of the flat buffers is 64-byte aligned and n a multiple of 4 - the tail, if any, element
// by element); WC: the updated weights also leave as bf16 (wcopy: the forward-layout copy the convolutions read - its index space is the master's -, so
// rua_weight_prep has only the data-gradient layout left to build, from this copy: rua_weight_prep_dgrad).  Same arithmetic per element as before.
template <bool WC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ th, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float lr_t, const float* __restrict__ lr_dev, float b1, float b2, float eps, float gs, int zero,
                                                   bf16_t* __restrict__ wcopy) {
  if (lr_dev) lr_t = lr_dev[0];
  const long long n4 = n >> 2;
  auto upd = [&](float gg_, float& mm, float& vv, float& tt) {
    const float gg = gg_ * gs;
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    tt -= lr_t * mm / (sqrtf(vv) + eps);
  };
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], t4 = reinterpret_cast<float4*>(th)[i];
    upd(g4.x, m4.x, v4.x, t4.x); upd(g4.y, m4.y, v4.y, t4.y); upd(g4.z, m4.z, v4.z, t4.z); upd(g4.w, m4.w, v4.w, t4.w);
    reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4; reinterpret_cast<float4*>(th)[i] = t4;
    if (zero) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (WC) reinterpret_cast<uint2*>(wcopy)[i] = make_uint2(ET<bf16_t>::pk(t4.x, t4.y), ET<bf16_t>::pk(t4.z, t4.w));
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float mm = m[i], vv = v[i], tt = th[i];
    upd(g[i], mm, vv, tt);
    m[i] = mm; v[i] = vv; th[i] = tt;
    if (zero) g[i] = 0.f;
    if constexpr (WC) wcopy[i] = (bf16_t)tt;
  }
}
template <bool WC>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ th, float* __restrict__ g, float* __restrict__ vel,
                                                  long long n, float lr, const float* __restrict__ lr_dev, float mu, float gs, int zero, bf16_t* __restrict__ wcopy) {
  if (lr_dev) lr = lr_dev[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float vv = mu * vel[i] - lr * g[i] * gs;
    vel[i] = vv;
    const float tt = th[i] + vv;
    th[i] = tt;
    if (zero) g[i] = 0.f;
    if constexpr (WC) wcopy[i] = (bf16_t)tt;
  }
}
// Step-dependent learning rate on the device, so that a captured training step needs no host-side scalar update between
// replays: state[0] = optimizer steps taken (advanced here), state[1] = base learning rate; lr_out[0] = the rate the
// optimizer kernel launched next reads (Adam: Keras' lr * sqrt(1 - beta2^t) / (1 - beta1^t), train_ISPRS.py:404-407).
__global__ void lr_step_kernel(double* state, float* lr_out, int adam, double beta1, double beta2) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double t = state[0] + 1.0;
    state[0] = t;
    double lr = state[1];
    if (adam) lr = lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t));
    lr_out[0] = (float)lr;
  }
}
extern "C" int rua_lr_step(double* state, float* lr_out, int adam, double beta1, double beta2, void* stream) {
  RUA_CHECK_ARG(state && lr_out, "rua_lr_step: null pointer");
  hipLaunchKernelGGL(lr_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr_out, adam, beta1, beta2);
  RUA_LAUNCH_CHECK("rua_lr_step");
  return RUA_OK;
}
extern "C" int rua_adam_step_w(float* theta, float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                               float eps, float grad_scale, int zero_grad, void* wcopy_bf16, void* stream) {
  RUA_CHECK_ARG(theta && g && m && v && n > 0, "rua_adam_step: bad arguments");
  RUA_CHECK_ARG((((size_t)theta | (size_t)g | (size_t)m | (size_t)v) & 15) == 0 && ((size_t)wcopy_bf16 & 7) == 0, "rua_adam_step: buffers must be 16-byte aligned");
  int64_t gr = (n / 4 + 255) / 256; if (gr > 4096) gr = 4096; if (gr < 1) gr = 1;
  if (wcopy_bf16) hipLaunchKernelGGL((adam_kernel<true>), dim3((int)gr), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, (long long)n, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale, zero_grad, (bf16_t*)wcopy_bf16);
  else hipLaunchKernelGGL((adam_kernel<false>), dim3((int)gr), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, (long long)n, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale, zero_grad, (bf16_t*)nullptr);
  RUA_LAUNCH_CHECK("rua_adam_step");
  return RUA_OK;
}
extern "C" int rua_adam_step(float* theta, float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                             float eps, float grad_scale, int zero_grad, void* stream) {
  return rua_adam_step_w(theta, g, m, v, n, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale, zero_grad, nullptr, stream);
}
extern "C" int rua_sgd_step_w(float* theta, float* g, float* vel, int64_t n, float lr, const float* lr_dev, float momentum, float grad_scale,
                              int zero_grad, void* wcopy_bf16, void* stream) {
  RUA_CHECK_ARG(theta && g && vel && n > 0, "rua_sgd_step: bad arguments");
  int64_t gr = (n + 255) / 256; if (gr > 4096) gr = 4096;
  if (wcopy_bf16) hipLaunchKernelGGL((sgd_kernel<true>), dim3((int)gr), dim3(256), 0, (hipStream_t)stream, theta, g, vel, (long long)n, lr, lr_dev, momentum, grad_scale, zero_grad, (bf16_t*)wcopy_bf16);
  else hipLaunchKernelGGL((sgd_kernel<false>), dim3((int)gr), dim3(256), 0, (hipStream_t)stream, theta, g, vel, (long long)n, lr, lr_dev, momentum, grad_scale, zero_grad, (bf16_t*)nullptr);
  RUA_LAUNCH_CHECK("rua_sgd_step");
  return RUA_OK;
}
extern "C" int rua_sgd_step(float* theta, float* g, float* vel, int64_t n, float lr, const float* lr_dev, float momentum, float grad_scale,
                            int zero_grad, void* stream) {
  return rua_sgd_step_w(theta, g, vel, n, lr, lr_dev, momentum, grad_scale, zero_grad, nullptr, stream);
}

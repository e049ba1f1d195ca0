// Shared device/host helpers for the gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/rua_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

void rua_set_error(const char* fmt, ...);
#define RUA_CHECK_ARG(cond, ...) do { if (!(cond)) { rua_set_error(__VA_ARGS__); return RUA_ERR_ARG; } } while (0)
#define RUA_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    rua_set_error("%s: %s", name, hipGetErrorString(e_)); return RUA_ERR_LAUNCH; } } while (0)

// ---- element traits: a "piece" is one 16-byte vector of VEC channels -------------------
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct ET<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pk(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;      // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
  }
};

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

// Buffer loads with hardware range checking: an offset >= num_records returns zeros, so zero padding and
// ragged edges need no branch (a predicated global load costs a branch + a drained vmcnt per load in hipcc).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define RUA_OOB 0xFFFFFFF0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bufload16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Tuning switches for experiments and A/B runs.  The launchers never read the environment: values change only through
// rua_set_tuning() (include/rua_hip.h); 0 in a "*_blocks" / "*_target" / "*_grid" field = derive it from the device's CU count.
struct RuaTuning {
  int conv_force_bn = 0, conv_force_bm = 0, conv_dma = -1, conv_pw = 1;
  long long conv_pw_minm = 65536;
  int conv_pw_blocks = 0, conv_halo = 1, halo64_maxd = 0 /* conv_halo<64> at d = 1 was 28 vs 34 us alone, but as the 4th member of the
                                                              grouped conv_igemm grid the d = 1 branch costs less: +0.35 % on the step */, conv_dmap = 1, dmap_target = 0, dmap_fused_finish = 0, dmap_rowb = 64, dmap_bm64 = 1;
  int wgrad_pw = 3 /* bit 1 (round 5): block partials + the batched reduction instead of replicas, tickets and a finishing block */, wgpw_blocks = 0, wgpw_r = 0, wgd_blocks = 0, wgrad_dmap = 1, wgd_mintiles = 10 /* 9 took the 64x64x128 level too: a wash alone, -0.55 % in the step since its K slabs (28 x 0.6 MB) */, wgrad_blocks = 0;
  int bn_grid = 0, tani_vec = 1, metrics_blocks = 0, stem_blocks = 0, head_blocks = 0;
  int conv_strip = 1, wgrad_slabs = 1, strip_narrow_maxd = 0, strip_group_share = 1;
  // grouped launches of a ResBlock's dilation branches, one bit per kernel family (0: every member launches on its own)
  int conv_group = 63;                  // 1 conv_strip, 2 conv_igemm<bf16,256,64>, 4 conv_dmap<128,128>, 8 conv_dmap<64,128>, 16 conv_small (unequal grids), 32 conv_pw with K <= 32 (unequal grids)
  int stats_blocks = 0;                 // col_stats grid cap (0: 1024)
  int stem_reg = 1;                     // stem with Cin <= 8: weights in registers, next pixel prefetched (0: the LDS-weights kernel)
  int head_fwd3 = 1;                    // bf16 heads with Cin = 32: the MFMA form, a lane per pixel (0: head_fwd2)
  int head_fwd3_bpc = 0;                // its blocks per CU over the batch (0: 2)
  int head_fwd2 = 1;                    // heads with Cin = 32: the register-weights kernel (0: the LDS-weights one)
  int conv_img2 = 1;                    // conv_img2 (round 5): the 3x3 convolutions of the 8 x 8 / 16 x 16 levels with whole images resident, coalesced weight rows, weights in registers, K split by input chunks + conv_splitk_finish (0: conv_dmap)
  int conv_img = 0;                     // 1: conv_img - the 3x3 convolutions of the 8 x 8 / 16 x 16 levels with a whole image resident in LDS instead of conv_dmap + split K (measured level: 28 - 29 us either way)
  int conv_small = 4096;                // conv_small serves 1x1 convolutions of at most this many output pixels (0: off)
  int bn_bwd_group = 1;                 // rua_bn_bwd_group: the one-branch BatchNorm backwards of a ResBlock in one grid
  int dmap_spread = 1;                  // conv_dmap: 0 the DMA instructions of a stage in one burst behind its barrier, 1 spread between the MFMAs (conv_dmap_s), 2 issued by waves of their own (conv_dmap_w; | 4: the 64-row tiles too).
                                        // Launch by launch 2 is the fastest (8 x 64 x 64 x 128: 18.5 - 20.2 us against 19.3 - 20.7 and 22 - 23; conv_dmap<128,128> 0.665 against 0.72 ms per step summed over
                                        // its launches), but the STEP is slower with it (8.05 - 8.07 against 8.02 ms on one box): 512-thread blocks of 239 registers leave no room on a CU for the blocks of
                                        // the launches the step's graph runs beside them
  int epi_fast = 1;                     // conv_dmap: compile-time forms of the epilogue for whole tiles (conv_epilogue KIND)
  int dmap_chain = 0;                   // grouped conv_dmap members back to back inside one block (conv_dmap_chain): bit 0 the 128-row tiles, bit 1 the 64-row tiles
  int dmap_group_bm128 = 0;             // grouped conv_dmap members keep 128-row tiles where the group as a whole fills the chip: measured no gain (8.308 vs 8.290 ms per step), off
  int wgrad_kernel_share = 0;           // the same for the K-split weight gradients (wgrad_kernel) of a group: measured SLOWER (8.47 vs 8.28 ms per step: that kernel is not persistent, fewer K slices = fewer blocks to hide latency with), off
  int wgrad_taps_share = 1;             // all-taps weight gradients of a group share one round of blocks (rua_wgrad_desc.group_members)
  int bn_regs = 1;                      // BatchNorm sweeps with the thread's coefficients in registers (0: read from the LDS table per piece)
  int fill_kernel = 1;                  // rua_fill_zero as a kernel, not hipMemsetAsync: no memset nodes in captured graphs (0: experiments, tools/dp_graph_check.py)
  int band_dbg = 0;                     // experiments only (tools/bench_conv_band.py): 1 rows from an L2-resident region, 2 no BatchNorm pass
  int conv_band64m = 1;                 // the independent 3x3 convs of a C = 64 ResBlock (first convs, data gradients) as ONE conv_band64m launch
  int conv_band64 = 1;                  // ... and of a C = 64 ResBlock (conv_band64)
  long long dbg_ptr = 0;                // diagnostic builds only (-DRUA_B128_STAMPS): device buffer for in-kernel stamps
  int conv_band128m = 13;               // conv_band128m for the independent 3x3 convs of a ResBlock (first convs, data gradients) as ONE launch: bit 0 at C = 128 (else grouped conv_dmap:
                                        // 37 - 43 vs 50 - 58 us, step 6.49 -> 6.39 ms), bit 1 at C = 64 (else conv_band64m; off: level with it launch by launch - 54 / 65 vs 60 / 65 us warm / cold
                                        // for plain first convolutions, 68 / 88 vs 69 / 92 for data gradients - and SLOWER in the step, 6.62 vs 6.53 ms: its in-place BatchNorm pass runs behind
                                        // the MFMAs of a stage, conv_band64m's between them, and the level-2 first convolutions normalise on load), bit 2 at C = 256 on 32-pixel rows (else grouped conv_dmap),
                                        // bit 3: a rua_conv_fwd of several 3x3 segments at those channel counts (the summed second convolutions of levels 3 - 4) in the same form with the sum kept on chip
                                        // (else conv_dmap_chain: 40 -> 32 us at level 3, 50 -> 44 at level 4, step -0.024 ms)
  int conv_band = 1;                    // rua_conv_fwd_sum: the branches' second convs of a C = 32 ResBlock as ONE launch with the sum kept on chip (conv_band32)
  int wgd_ks_slow = 1;                  // wgrad_dmap block order: K slice slowest (blocks that read the same pixels share an XCD's L2)
  int wgrad_rows = 127;                 // bit 0: wgrad_rows32 (the all-taps weight gradient at C = 32 on whole rows, W = 256 / 128, one shared LDS-DMA ring), bit 1: wgrad_rows64 (C = 64, W = 128), bit 2: wgrad_rows128, bits 3 - 4: wgrad_img / wgrad_imgs, bit 5: wgrad_rowsx<1> (C = 256 on 32-pixel rows, was wgrad_dmap), bit 6: wgrad_rowsx<0> instead of wgrad_rows128, bit 7 (off): wgrad_rows32 / wgrad_rows64 deal their rows as a slot stream too (WgSlots) - measured: level 2 48 - 50 -> 46 - 50 us per group, level 1 55 - 59 -> 56 - 59 (at d = 1 the cursor work costs the stage loop 10 %, what the short chains of d = 31 gain); 0: wgrad_taps_kernel
  int cu_reserve = 0;                   // CUs the one-round grids leave free (rua_cu_count() = CUs - cu_reserve): room for RCCL's kernels under data parallel
  int strip_seglen = 0;                 // experiments (tools/bench_conv3x3.py): rows per block of conv_strip, 0 = one round of blocks
  int band_stag = 1;                    // conv_band32s (staggered halves) for full-width BatchNorm + ReLU sums (0: conv_band32; >= 4: that many ring slots)
  int strip_stag = 1;                   // conv_strip32s: full-width strips with the two halves of a block half a stage apart (0: conv_strip32 everywhere)
  int wgrad_group = 23;                 // 1 wgrad_kernel, 2 wgrad_taps<32>, 4 wgrad_taps<64>, 8 wgrad_dmap (off: three members at once thrash the L2, 27.6 vs 24.6 us each), 16 wgrad_pw (members with workspaces of their own)
};
extern RuaTuning g_tune;
int rua_cu_count();          // compute units of the current device (queried once per device, cached)
int rua_device_index();      // the calling thread's current HIP device
// "done once" flags of the launchers (hipFuncSetAttribute is per device): one flag per device, not per process - a process that drives
// several devices (not this package's model: one process per GPU) would otherwise configure a kernel on the first device only
struct RuaPerDevFlag { bool f[64] = {}; bool& get() { return f[rua_device_index() & 63]; } };

// Sum of the R replicas of stats[.][2][C] for channel c: the R loads are issued together (independent
// addresses, unrolled) instead of a dependent chain, so a finalize launch costs one memory round trip.
__device__ __forceinline__ void replica_sum(const double* __restrict__ stats, int R, int C, int c, double& s1, double& s2) {
  double a1 = 0, a2 = 0, b1 = 0, b2 = 0;
  int r = 0;
  for (; r + 8 <= R; r += 8) {                           // 16 independent loads in flight: one round trip per 8 replicas
    double x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = stats[(size_t)(r + u) * 2 * C + c]; y[u] = stats[(size_t)(r + u) * 2 * C + C + c]; }
    a1 += (x[0] + x[1]) + (x[2] + x[3]); b1 += (x[4] + x[5]) + (x[6] + x[7]);
    a2 += (y[0] + y[1]) + (y[2] + y[3]); b2 += (y[4] + y[5]) + (y[6] + y[7]);
  }
  for (; r + 4 <= R; r += 4) {
    const double x0 = stats[(size_t)(r + 0) * 2 * C + c], y0 = stats[(size_t)(r + 0) * 2 * C + C + c];
    const double x1 = stats[(size_t)(r + 1) * 2 * C + c], y1 = stats[(size_t)(r + 1) * 2 * C + C + c];
    const double x2 = stats[(size_t)(r + 2) * 2 * C + c], y2 = stats[(size_t)(r + 2) * 2 * C + C + c];
    const double x3 = stats[(size_t)(r + 3) * 2 * C + c], y3 = stats[(size_t)(r + 3) * 2 * C + C + c];
    a1 += x0 + x1; b1 += x2 + x3; a2 += y0 + y1; b2 += y2 + y3;
  }
  for (; r < R; ++r) { a1 += stats[(size_t)r * 2 * C + c]; a2 += stats[(size_t)r * 2 * C + C + c]; }
  s1 = a1 + b1; s2 = a2 + b2;
}

// p[c], or 0 for a null pointer, without a branch: a buffer load whose range is empty for a null pointer.  Several of these are in flight at once;
// behind `if (p)` every load waits for the one before it (a prologue of four dependent round trips instead of one).  p must be wave-uniform.
__device__ __forceinline__ float ld_f32_or_zero(const float* p, int c, int n) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(p, p ? (unsigned)(n * 4) : 0u), (unsigned)(c * 4), 0, 0));
}

// in_fold of up to RUA_MAX_BRANCH members at once - the prologue of the row-streaming kernels (conv_strip32s, conv_band32s, conv_band64): every
// block derives scale / shift of each member's BatchNorm from the replicated fp64 statistics of its input.  The NT / CH thread groups are dealt to
// the members (gpm groups each); a thread sums the replicas gi, gi + gpm, ... of its channel with ALL its loads in flight at once - buffer loads
// whose range ends at the replica count, so a replica beyond it reads zeros and there is no loop to put a memory round trip behind every pair
// of loads (in-kernel stamps, tools/band_phases.py: the four members of a conv_band32 launch one after the other, two block barriers and a
// dependent load loop each, were 6.5 - 7.7 us of an 83 us launch) -, gamma / beta of the finishing thread are fetched beside them; the first
// group of a member adds the groups' shares in a fixed order, finishes mean / variance -> scale / shift and hands them to store(m, c, scale,
// shift); `publish` (one block of the grid) also writes the coefficients out and updates the moving statistics.  red: LDS scratch of
// NT / CH * 2 * CH doubles.  Ends with a block barrier.
// getf(m): the member's rua_bn_fold by reference (a kernel argument: no pointer to it may escape, or hipcc copies the argument block to scratch and
// every descriptor in it turns into a vector value); has(m): false for a member whose coefficients were given.
template <int NT, int CH, typename GetF, typename Has, typename Store>
__device__ __forceinline__ void rua_fold_members(GetF&& getf, Has&& has, int nmem, bool publish, double* red, int tid, Store&& store) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  constexpr int NG = NT / CH, gpm = NG / RUA_MAX_BRANCH;
  static_assert(gpm >= 1, "thread groups per member");
  constexpr int KMAX = (32 + gpm - 1) / gpm;            // rua_stats_replicas() hands out at most 32
  constexpr int GPW = 64 / CH;                          // thread groups per wave: a wave's groups belong to ONE member (its descriptor stays in SGPRs)
  static_assert(CH <= 64 && gpm % GPW == 0, "a wave must not straddle two members");
  const int c = tid % CH, grp = tid / CH;
  const int m = __builtin_amdgcn_readfirstlane(tid >> 6) / (gpm / GPW), gi = grp - m * gpm;
  const int mc = m < nmem ? m : 0;
  const bool act = m < nmem && has(mc);
  const rua_bn_fold& f = getf(mc);
  float gam = 0.f, bet = 0.f;
  if (act) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(f.stats, (unsigned)(f.replicas * 2 * CH * 8));
    u32x2_t x[KMAX], y[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const unsigned off = (unsigned)(((gi + k * gpm) * 2 * CH + c) * 8);
      x[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      y[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, off + (unsigned)(CH * 8), 0, 0);
    }
    if (gi == 0) { gam = f.gamma[c]; bet = f.beta[c]; }
    double a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { a1 += __builtin_bit_cast(double, x[k]); a2 += __builtin_bit_cast(double, y[k]); }
    for (int r = gi + KMAX * gpm; r < f.replicas; r += gpm) { a1 += f.stats[(size_t)r * 2 * CH + c]; a2 += f.stats[(size_t)r * 2 * CH + CH + c]; }
    red[(grp * 2) * CH + c] = a1; red[(grp * 2 + 1) * CH + c] = a2;
  }
  __syncthreads();
  if (act && gi == 0) {
    double s1 = 0, s2 = 0;
    for (int g = 0; g < gpm; ++g) { s1 += red[((grp + g) * 2) * CH + c]; s2 += red[((grp + g) * 2 + 1) * CH + c]; }
    const double mean = s1 / f.count;
    double v = s2 / f.count - mean * mean;
    if (v < 0) v = 0;
    const double r = 1.0 / sqrt(v + (double)f.eps);
    const double sc = (double)gam * r;
    const float scf = (float)sc, shf = (float)((double)bet - mean * sc);
    store(m, c, scf, shf);
    if (publish) {
      f.scale[c] = scf; f.shift[c] = shf;
      if (f.mean) f.mean[c] = (float)mean;
      if (f.rstd) f.rstd[c] = (float)r;
      if (f.moving_mean) {
        const double unb = f.bessel_n > 1 ? v * (f.bessel_n / (f.bessel_n - 1)) : v;
        f.moving_mean[c] = (float)((double)f.moving_mean[c] * f.momentum + mean * (1.0 - f.momentum));
        f.moving_var[c] = (float)((double)f.moving_var[c] * f.momentum + unb * (1.0 - f.momentum));
      }
    }
  }
  __syncthreads();
}

// ---- kernel-side view of a rua_conv_desc (filled by rua_conv_fwd, shared by the conv kernels of conv_mfma.hip / conv_strip.hip)
struct SegK { const unsigned char* x; const unsigned char* w; int C, Hs, Ws, up, dil, taps, nchunk, ubegin; unsigned xbytes, wbytes; };
struct ConvK {
  SegK seg[RUA_MAX_SEG];
  int nseg, nunits;
  int N, H, W, Cout, stride;
  long long M;
  const float* bias; const float* bias_more[3]; const unsigned char* aux; int aux_mode; const float* mscale; const float* mshift;
  int out_relu, accumulate; unsigned char* y; int out_stride, OH, OW; double* stats; int stats_mode; int stats_R;
  int nbn, nbm, ksplit, stages_per_split; float* ws;
  int* cnt;      // per-tile ticket counters (all zero between launches) for the in-launch split-K reduction, or null
  const float* in_scale; const float* in_shift; int in_relu;     // per-input-channel affine (+ ReLU) applied to segment 0 on load (conv_strip)
  int epi_fast;  // tuning key epi_fast: whole tiles of the LDS-DMA kernels leave through the compile-time forms of the epilogue
};
typedef __attribute__((address_space(3))) void* lds_void_p;
struct ConvKG { ConvK k[RUA_MAX_BRANCH]; };           // members of a grouped launch: blockIdx.y picks one
static_assert(sizeof(ConvKG) <= 4096, "grouped launch: kernel arguments are limited to 4 KiB");
// capture mode of the launchers (rua_conv_fwd_group): a groupable launch is recorded instead of issued
struct ConvGroupCapture {
  int members;   // convolutions in the group being captured (launchers that size their grid by the CU count share the chip)
  int n; int kind[RUA_MAX_BRANCH]; unsigned grid[RUA_MAX_BRANCH]; int smem[RUA_MAX_BRANCH]; ConvK k[RUA_MAX_BRANCH];
  bool add(int kd, unsigned g, int sm, const ConvK& kk) {
    if (n >= RUA_MAX_BRANCH) return false;
    kind[n] = kd; grid[n] = g; smem[n] = sm; k[n] = kk; ++n;
    return true;
  }
};
extern thread_local ConvGroupCapture* g_conv_group;
int rua_strip_group_flush(hipStream_t st, int* grids); // conv_strip.hip: issues the conv_strip members captured since the group began (+1 on *grids per launch)
int rua_strip_group_pending(void);                     // members captured and not yet issued
void rua_strip_group_reset(void);                      // drops captured members (group entry, error paths)
// conv_band64.hip
bool rua_band64_ok(const rua_conv_desc* d, int n);
int rua_launch_band64(const rua_conv_desc* d, int n, hipStream_t st);
bool rua_band64m_ok(const rua_conv_desc* d, int n);    // independent members (rua_conv_fwd_group) as one conv_band64m launch
int rua_launch_band64m(const rua_conv_desc* d, int n, hipStream_t st);
// conv_band128.hip
bool rua_band128m_ok(const rua_conv_desc* d, int n);   // independent members at C = Cout = 128 on 64-pixel rows as one conv_band128m launch
int rua_launch_band128m(const rua_conv_desc* d, int n, hipStream_t st);
bool rua_band128_sum_ok(const rua_conv_desc* d);         // rua_conv_fwd with several 3x3 segments (the summed second convolutions of a level-3 / level-4 ResBlock) in conv_band128m's form
int rua_launch_band128_sum(const rua_conv_desc* d, hipStream_t st);
// conv_img2.hip
int rua_pick_img2(const rua_conv_desc* d);             // K slices (0: not this kernel)
int rua_launch_conv_img2(ConvK& k, const rua_conv_desc* d, int KS, hipStream_t st);
int rua_splitk_finish_bf16(const ConvK& k, hipStream_t st);   // conv_mfma.hip: conv_splitk_finish over the K slices' slabs
// conv_strip.hip
bool rua_pick_strip(const rua_conv_desc* d);
int rua_launch_conv_strip(const ConvK& k, const rua_conv_desc* d, hipStream_t st);

static inline int rua_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

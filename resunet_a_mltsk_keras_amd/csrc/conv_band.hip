// conv_band32<NW, BR, R>: the n-ary Add of a ResBlock's dilation branches kept ON CHIP (model2.py:26-31) at the top level of the
// network (C = Cout = 32, bf16):
//
//     y = residual + sum_b ( bias_b + W_b (*)_{d_b} relu(scale_b * x_b + shift_b) )          b = 0 .. nb-1 (<= 4), d_b <= 32
//
// i.e. the SECOND convolutions of all branches of the d6 residual atrous block (ResBlock(32,[1,3,15,31]) at full resolution,
// model2.py:102) in one launch.  As separate accumulating launches (conv_strip) the output tensor is written four times and read
// back three times - 7 of the block's 22 forward tensor passes; here it is written once.
//
// A block owns a BAND of BR consecutive output rows of one image (all SW = 32 * NW columns of a strip) and keeps the band's fp32
// accumulators in registers (BR x one 32-pixel x 32-channel MFMA tile per wave) while it walks the 3 * nb PHASES (branch b, kernel
// row ty): in a phase the BR input rows h0 + r + (ty - 1) * d_b, r = 0 .. BR-1, of x_b stream through a ring of R LDS row slots
// (LDS-DMA, counted vmcnt, one barrier per row - the conv_strip pipeline), each row is normalised in place as it lands (BatchNorm +
// ReLU on load: no normalised copy of x_b in HBM) and feeds the three taps of kernel row ty (6 MFMAs per wave) into the accumulator
// of output row r.  96 back-to-back stages per block instead of conv_strip's ~10 per launch: the DMA ring stays full for the whole
// kernel.  The rows a band needs from x_b at dilation d are the bands d rows above and below: blocks of one image run on one XCD (job
// order) in the same phase at about the same time, so those re-reads are L2 / Infinity-Cache hits (y1 was written by the launch
// before this one).  Weights: the 18 KB of branch b + 1 arrive by LDS-DMA during the last phase of branch b (one DMA instruction per
// wave and stage, part of the uniform operation count the vmcnt waits rely on) and the six fragments of a kernel row are read into
// registers at the start of each phase.
//
// Slot layout: pixel j of a slot row is image column x0 - 32 + j for every branch (the halo is the maximum dilation), so DMA
// addressing and the in-place transform do not depend on the branch; columns outside the image are out-of-range DMA lanes (zeros,
// no memory traffic): for full-width strips the halo costs nothing.
#include "common.h"
#include <type_traits>

struct BandK {
  const unsigned char* x[RUA_MAX_BRANCH];
  const unsigned char* w[RUA_MAX_BRANCH];
  const float* bias[RUA_MAX_BRANCH];
  const float* in_scale[RUA_MAX_BRANCH];
  const float* in_shift[RUA_MAX_BRANCH];
  rua_bn_fold f[RUA_MAX_BRANCH];
  int has_fold, has_bn, in_relu, nb;
  int d[RUA_MAX_BRANCH];
  const unsigned char* res;
  unsigned char* y;
  int N, H, W, strips, bands, njobs;
  unsigned xbytes;
  int dbg;
};
static_assert(sizeof(BandK) <= 4096, "kernel arguments are limited to 4 KiB");

#ifdef RUA_BAND_TS                                    // (debug build: 100 MHz wall-clock stamps of three blocks, read by rua_band_debug_ts; tools/band_phases.py)
__device__ unsigned long long g_band_ts[3 * 32 + 32];
#define RUA_BST(i) do { if (r == 3) st_[i] = clock64(); } while (0)
#define RUA_BTS(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == gridDim.x - 1)) \
    g_band_ts[(blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : 2)) * 32 + (i)] = wall_clock64(); } while (0)
extern "C" int rua_band_debug_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_band_ts), sizeof(g_band_ts)); }
#else
#define RUA_BTS(i) do { } while (0)
#define RUA_BST(i) do { } while (0)
#endif

// FULLW: the strip is the whole image row (W == SW).  The 32 halo pixels either side of a slot row are then zero padding for every
// branch: they are zeroed ONCE and the row DMAs cover the interior only - 2 instead of 3 DMA instructions and transformed pieces
// per wave and row (the kernel is bound by its per-stage instruction issue - VALU of the in-place BatchNorm, DMA issue - not by HBM).
template <int NW, int BR, int R, bool FULLW>
__device__ __forceinline__ void conv_band32_body(const BandK& q) {
  typedef bf16_t T;
  constexpr int C = 32, NT = NW * 64, SW = NW * 32, HALO = 32;
  constexpr int SPX = SW + 2 * HALO, SLOT = SPX * 64;
  constexpr int DMA0 = FULLW ? HALO : 0;                // first slot pixel the row DMAs write
  constexpr int SLOT_INST = (FULLW ? SW : SPX) * 64 / 1024;
  constexpr int NPX = (SLOT_INST + NW - 1) / NW;        // x-row DMA instructions per wave and row
  // operations issued after the DMAs of row s + 1 when stage s waits for them (behind its own DMA issue): the rows s + 2 .. s + R - 1.
  // The weight DMAs of the last phase of a branch come on top in a few stages: the count then waits for a little more than
  // necessary, never for less
  constexpr int NWAIT = (R - 2) * NPX;
  constexpr int WPIECES = 18;                           // 9 taps x 2 k-steps, 1 KiB each (64 lanes x 16 bytes)
  constexpr int WST = (WPIECES + NW - 1) / NW;          // stages it takes the block to issue them
  static_assert(WST <= BR && R - 1 <= BR && R >= 4, "pipeline depths");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem + R * SLOT;
  unsigned char* sDump = sW + WPIECES * 1024;
  float* tab = reinterpret_cast<float*>(sDump + NW * 1024);           // [nb][2][32] scale, shift ; [32] bias sum at 4 * 64

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform: descriptors selected by it stay in SGPRs
  const int pl = lane & 31, kh = lane >> 5;
  const int H = q.H, W = q.W, nb = q.nb;

  const int nwg = q.njobs, bid = blockIdx.x;
  if (bid >= nwg) return;
  RUA_BTS(0);
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // consecutive jobs share an XCD's L2
  const int band = job % q.bands, tq = job / q.bands;
  const int x0 = (tq % q.strips) * SW, n_ = tq / q.strips;
  const int h0 = band * BR;
  const int S = nb * 3 * BR;                            // stages

  // ---- per-channel tables (ordinary loads: all consumed before the first LDS-DMA is issued) ------------------------------
  if (tid < 32) {
    float b = 0.f;
    for (int i = 0; i < nb; ++i) if (q.bias[i]) b += q.bias[i][tid];
    tab[4 * 64 + tid] = b;
  }
  if (q.has_fold) {
    // in_fold: every block derives the BatchNorm coefficients of all branches from the replicated fp64 statistics of their inputs
    // (conv_strip's prologue, once per branch); job 0 publishes them and updates the moving statistics.  The ring is still empty.
    constexpr int NG = NT / 32;
    double* red = reinterpret_cast<double*>(smem);
    const int c = tid & 31, grp = tid >> 5;
    for (int i = 0; i < nb; ++i) {
      const rua_bn_fold& f = q.f[i];
      double a1 = 0, a2 = 0;
      for (int r = grp; r < f.replicas; r += NG) { a1 += f.stats[(size_t)r * 64 + c]; a2 += f.stats[(size_t)r * 64 + 32 + c]; }
      red[(grp * 2) * 32 + c] = a1; red[(grp * 2 + 1) * 32 + c] = a2;
      __syncthreads();
      if (tid < 32) {
        double s1 = 0, s2 = 0;
        for (int g = 0; g < NG; ++g) { s1 += red[(g * 2) * 32 + tid]; s2 += red[(g * 2 + 1) * 32 + tid]; }
        const double m = s1 / f.count;
        double v = s2 / f.count - m * m;
        if (v < 0) v = 0;
        const double rs = 1.0 / sqrt(v + (double)f.eps);
        const double sc = (double)f.gamma[tid] * rs;
        const float scf = (float)sc, shf = (float)((double)f.beta[tid] - m * sc);
        tab[i * 64 + tid] = scf; tab[i * 64 + 32 + tid] = shf;
        if (job == 0) {
          f.scale[tid] = scf; f.shift[tid] = shf;
          if (f.mean) f.mean[tid] = (float)m;
          if (f.rstd) f.rstd[tid] = (float)rs;
          if (f.moving_mean) {
            const double unb = f.bessel_n > 1 ? v * (f.bessel_n / (f.bessel_n - 1)) : v;
            f.moving_mean[tid] = (float)((double)f.moving_mean[tid] * f.momentum + m * (1.0 - f.momentum));
            f.moving_var[tid] = (float)((double)f.moving_var[tid] * f.momentum + unb * (1.0 - f.momentum));
          }
        }
      }
      __syncthreads();
    }
  } else if (tid < 32) {
    for (int i = 0; i < nb; ++i) {
      tab[i * 64 + tid] = q.in_scale[i] ? q.in_scale[i][tid] : 1.f;
      tab[i * 64 + 32 + tid] = q.in_shift[i] ? q.in_shift[i][tid] : 0.f;
    }
  }
  __syncthreads();
  RUA_BTS(1);
  const bool bn = q.has_bn != 0 && !(q.dbg & 2);
  const int d0 = q.d[0], d1 = q.d[1], d2 = q.d[2], d3 = q.d[3];
  auto dil_of = [&](int b) { return b == 0 ? d0 : (b == 1 ? d1 : (b == 2 ? d2 : d3)); };

  // ---- DMA addressing: lane l of instruction `inst` moves the 16-byte piece psrc of slot pixel DMA0 + inst * 16 + (l >> 2) into slot
  // position (l & 3) = psrc ^ ((pixel >> 2) & 3): conflict-free ds_read_b128 of the b-operand at every tap shift (conv_strip).
  // The kernel is bound by instruction issue (measured: rows from an L2-resident region run as fast as from HBM), so everything a
  // stage needs is a per-kernel or per-phase constant plus ONE add: out-of-range rows / columns are encoded in the offsets themselves
  // (row base 0x80000000, column offset 0x7FFFFF00: either makes the sum >= num_records, the DMA then writes zeros).
  const int psrc = (lane & 3) ^ ((lane >> 4) & 3);
  constexpr unsigned ROW_OOB = 0x80000000u, COL_OOB = 0x7FFFFF00u;
  unsigned xrel[NPX]; unsigned pdst[NPX]; bool xok[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int inst = k * NW + wv;
    const int j = DMA0 + inst * 16 + (lane >> 2);       // slot pixel
    const int x = x0 - HALO + j;
    const bool in_slot = inst < SLOT_INST;              // wave-uniform
    xok[k] = in_slot && x >= 0 && x < W;
    xrel[k] = xok[k] ? (unsigned)((x * C + psrc * 8) * 2) : COL_OOB;
    // this lane's piece of instruction k inside a slot (the wave's dump KiB for the instructions beyond the slot: they still issue,
    // the operation counts of the vmcnt waits are uniform)
    pdst[k] = in_slot ? (unsigned)(DMA0 * 64 + inst * 1024 + lane * 16) : (unsigned)((sDump - smem) + wv * 1024 + lane * 16);
  }
  const unsigned smem_a = (unsigned)(size_t)(lds_void_p)smem;
  const unsigned rowbytes = (unsigned)(W * C * 2), imgbase = (unsigned)(n_ * H) * rowbytes;
  const int o = wv * 32 + pl;                           // this lane's output pixel inside the strip

  // per-phase constants: phase ph = 3 b + ty reads the rows hb + r, r = 0 .. BR-1, of branch b
  struct Phase { int hb; bool valid; __amdgpu_buffer_rsrc_t rx; unsigned ca; };
  auto phase = [&](int ph) {
    Phase p;
    const int b = ph / 3, ty = ph - 3 * b;
    p.valid = ph < 3 * nb;
    p.hb = h0 + (ty - 1) * dil_of(b);
    p.rx = make_rsrc(q.x[p.valid ? b : 0], q.xbytes);
    p.ca = smem_a + (unsigned)((unsigned char*)(tab + (p.valid ? b : 0) * 64 + psrc * 8) - smem);
    return p;
  };
  // row DMAs of (phase p, row r) into the slot at byte offset so
  auto issue_x = [&](const Phase& p, int r, unsigned so) {
    const int h = p.hb + r;
    const bool ok = p.valid && (unsigned)h < (unsigned)H;
    const unsigned base = ok ? ((q.dbg & 1) ? (unsigned)(h & 7) * rowbytes : imgbase + (unsigned)h * rowbytes) : ROW_OOB;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int inst = k * NW + wv;
      unsigned char* dst = inst < SLOT_INST ? smem + so + DMA0 * 64 + inst * 1024 : sDump + wv * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(p.rx, (lds_void_p)dst, 16, base + xrel[k], 0, 0, 0);
    }
  };
  // weight piece idx (one KiB: tap idx >> 1, k-step idx & 1, a fragment per lane) of branch b into sW; wave-uniform condition
  auto issue_w = [&](int b, int idx) {
    if (idx < WPIECES && b < nb) {
      const __amdgpu_buffer_rsrc_t rw = make_rsrc(q.w[b], (unsigned)(9 * C * C * 2));
      const unsigned off = (unsigned)((((idx >> 1) * C * C) + pl * C + (idx & 1) * 16 + kh * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_p)(sW + idx * 1024), 16, off, 0, 0, 0);
    }
  };
  // BatchNorm (+ ReLU) of a landed row, in place, on the pieces this thread's own DMA instructions wrote (padding stays zero), in
  // three steps so that the arithmetic of row s + 1 runs beside the MFMAs of row s: tr_read (LDS -> registers), tr_math, tr_write.
  // RAW LDS accesses: in front of a C++ LDS load / store hipcc waits for every LDS-DMA in flight (seen in the ISA as vmcnt(0) here
  // and as vmcnt(2) in conv_strip) - the ring would drain at every stage; these pieces' own DMAs have landed (counted wait).
  auto tr_valid = [&](const Phase& p, int r) { return bn && p.valid && (unsigned)(p.hb + r) < (unsigned)H; };      // block-uniform
  auto tr_read = [&](const Phase& p, unsigned so, f32x4& sa, f32x4& sb, f32x4& ha, f32x4& hb, u32x4_t* rw) {
    const unsigned a0 = smem_a + so + pdst[0];          // instructions 0 and 1 are always slot pieces
    static_assert(NPX == 2 || NPX == 3, "the asm blocks below read two or three pieces");
    if constexpr (NPX == 3) {
      const unsigned a2 = smem_a + (pdst[2] < (unsigned)(R * SLOT) ? so : 0u) + pdst[2];      // instruction 2 may be a dump piece
      asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:128\n\tds_read_b128 %3, %7 offset:144\n\t"
                   "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:%c10\n\tds_read_b128 %6, %9\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sa), "=&v"(sb), "=&v"(ha), "=&v"(hb), "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2])
                   : "v"(p.ca), "v"(a0), "v"(a2), "n"(NW * 1024) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:128\n\tds_read_b128 %3, %6 offset:144\n\t"
                   "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:%c8\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sa), "=&v"(sb), "=&v"(ha), "=&v"(hb), "=&v"(rw[0]), "=&v"(rw[1])
                   : "v"(p.ca), "v"(a0), "n"(NW * 1024) : "memory");
    }
  };
  auto tr_math = [&](const f32x4& sa, const f32x4& sb, const f32x4& ha, const f32x4& hb, u32x4_t* rw) {
    const float sc8[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
    const float sh8[8] = {ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      float f[8];
      ET<T>::unpack(make_uint4(rw[k][0], rw[k][1], rw[k][2], rw[k][3]), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaf(sc8[j], f[j], sh8[j]);
      uint4 pk = ET<T>::pack(f);
      if (q.in_relu) {
        // ReLU on the packed pairs: a negative bf16 is a negative int16 (sign bit), max with 0 clears it: one v_pk_max_i16 per pair
        typedef __attribute__((ext_vector_type(2))) short s16x2;
        const s16x2 z = {0, 0};
        auto relu2 = [&](unsigned v) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z)); };
        pk.x = relu2(pk.x); pk.y = relu2(pk.y); pk.z = relu2(pk.z); pk.w = relu2(pk.w);
      }
      rw[k][0] = pk.x; rw[k][1] = pk.y; rw[k][2] = pk.z; rw[k][3] = pk.w;
    }
  };
  auto tr_write = [&](unsigned so, const u32x4_t* rw) {
#pragma unroll
    for (int k = 0; k < NPX; ++k)
      if (FULLW || xok[k]) {                            // (a full-width strip has no out-of-image columns in its DMA range)
        const unsigned la = smem_a + so + pdst[k];
        asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(rw[k]) : "memory");
      }
  };

  f32x16 acc[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;

  // ---- prologue: weights of branch 0 and rows 0 .. R-2 in flight, then everything landed -------------------------------------
  if constexpr (FULLW) {                                 // the halo pixels of every slot: zero once, never written again
    constexpr int HB = HALO * 64;                        // bytes per halo side
    for (int i = tid; i < R * 2 * HB / 16; i += NT) {
      const int sl = i / (2 * HB / 16), k = i - sl * (2 * HB / 16);
      unsigned char* p = smem + sl * SLOT + (k < HB / 16 ? k * 16 : (HALO + SW) * 64 + (k - HB / 16) * 16);
      *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  }
  Phase cur = phase(0), nxt = phase(1);
#pragma unroll
  for (int i = 0; i < WST; ++i) issue_w(0, i * NW + wv);
#pragma unroll
  for (int s = 0; s <= R - 2; ++s) issue_x(cur, s, (unsigned)(s * SLOT));      // R - 1 <= BR: all in phase 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tr_valid(cur, 0)) {                                // row 0; row s + 1 is normalised during stage s
    f32x4 sa, sb, ha, hb; u32x4_t rw[3];
    tr_read(cur, 0u, sa, sb, ha, hb, rw); tr_math(sa, sb, ha, hb, rw); tr_write(0u, rw);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  RUA_BTS(2);

  // slot byte offsets of the row being consumed, the next one (normalised meanwhile) and the one the DMAs of this stage fill
  unsigned so_cur = 0, so_nxt = SLOT, so_iss = (R - 1) * SLOT;
  for (int ph = 0; ph < 3 * nb; ++ph) {
    const int b = ph / 3, ty = ph - 3 * b;
    const int d = dil_of(b);
    // b-operand fragment addresses inside a slot row: output pixel o, tap column tx reads slot pixel HALO + o + (tx - 1) d
    unsigned boff[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int j = HALO + o + (tx - 1) * d;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) boff[tx][ks] = (unsigned)(j * 64 + (((ks * 2 + kh) ^ ((j >> 2) & 3)) * 16));
    }
    bf16x8 wf[3][2];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      __builtin_amdgcn_s_barrier();                      // everyone's pieces of row s are normalised; every wave is done with row s - 1
      if (r == 0) {
        // kernel row ty of this branch's weights (landed before the phase began: issued during the last phase of branch b - 1,
        // which every wave has waited out; visible since the barrier)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            wf[tx][ks] = *reinterpret_cast<const bf16x8*>(sW + (((ty * 3 + tx) * 2 + ks) * 1024) + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // weights of the next branch while the last kernel row of this one runs (the pieces overwritten are not read any more)
      if (r < WST && ty == 2) issue_w(b + 1, r * NW + wv);
      // row s + R - 1 into the slot of row s - 1
      if (!(q.dbg & 4)) {
        if (r + R - 1 < BR) issue_x(cur, r + R - 1, so_iss);
        else issue_x(nxt, r + R - 1 - BR, so_iss);
      }

      const unsigned char* row = smem + so_cur;
      bf16x8 fx[3][2];
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx[tx][ks] = *reinterpret_cast<const bf16x8*>(row + boff[tx][ks]);
      // row s + 1: this wave's own pieces have landed; its BatchNorm runs beside the MFMAs of row s and is published by the next barrier
      const bool tv = r + 1 < BR ? tr_valid(cur, r + 1) : tr_valid(nxt, 0);
      f32x4 sa, sb, ha, hb; u32x4_t rw[3];
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");
      if (tv) tr_read(r + 1 < BR ? cur : nxt, so_nxt, sa, sb, ha, hb, rw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tx][ks], fx[tx][ks], acc[r], 0, 0, 0);
      if (tv) { tr_math(sa, sb, ha, hb, rw); tr_write(so_nxt, rw); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      so_iss = so_cur; so_cur = so_nxt; so_nxt = so_nxt + SLOT == (unsigned)(R * SLOT) ? 0u : so_nxt + SLOT;
    }
    cur = nxt;
    nxt = phase(ph + 2);
    RUA_BTS(3 + ph);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");       // the over-issued DMAs of the last stages
  RUA_BTS(16);

  // ---- epilogue: bias sum + residual, one write of the band ------------------------------------------------------------------
  // acc[r][k]: channel (k & 3) + 8 * (k >> 2) + 4 * kh of pixel pl; one half-wave exchange per register pair -> this lane holds
  // channels 16 g + 8 kh .. + 7 of its pixel (conv_strip / conv_pw)
  const size_t pix0 = (size_t)((n_ * H + h0) * W + x0 + o);
  uint4 rv[2][2];
  auto load_res = [&](int r, uint4* dst) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
      dst[g] = q.res ? ldg16(q.res + ((pix0 + (size_t)r * W) * C + 16 * g + 8 * kh) * 2) : make_uint4(0, 0, 0, 0);
  };
  load_res(0, rv[0]);
#pragma unroll
  for (int r = 0; r < BR; ++r) {
    if (r + 1 < BR) load_res(r + 1, rv[(r + 1) & 1]);
    float v[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[r][(2 * g) * 4 + j], b2 = acc[r][(2 * g + 1) * 4 + j];
        if (g == 0 && j == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        v[g][j] = a;
        v[g][4 + j] = b2;
      }
    unsigned char* yrow = q.y + ((pix0 + (size_t)r * W) * C) * 2;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = 16 * g + 8 * kh;
      float a8[8];
      ET<T>::unpack(rv[r & 1][g], a8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[g][j] += tab[4 * 64 + co + j] + a8[j];
      stg16(yrow + co * 2, ET<T>::pack(v[g]));
    }
  }
  RUA_BTS(17);
#ifdef RUA_BAND_TS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RUA_BTS(18);
#endif
}

// (the body is a __device__ template behind __global__ wrappers, like conv_strip's: the host pass then never has to instantiate it -
// with the body itself as a __global__ template the host-side stubs silently went missing whenever the body used a device-only
// construct the host pass rejects)
template <int NW, int BR, int R, bool FULLW> __global__ __launch_bounds__(NW * 64) void conv_band32(const BandK q) { conv_band32_body<NW, BR, R, FULLW>(q); }

// ---- conv_band32s<NW, R>: the same launch in the staggered form of conv_strip32s (round 4) ------------------------------------------
// In-kernel stamps of conv_band32 (tools/band_phases.py, 8 x 256 x 256 x 32 x 4 branches, 83 us): 7.7 us tables + BatchNorm fold (the four
// branches one after the other, two block barriers each), 1.4 - 2.2 us ring fill, 12 phases x 5.44 us = 0.68 us per stage of 6 MFMAs per wave
// (0.43 where the rows lie outside the image and nothing is normalised), 4.2 - 5.3 us epilogue, ~5 us launch ramp + drain.  A stage's MFMA time is
// 0.16 us (two waves per SIMD): both waves of a SIMD leave the barrier together, issue their six dependent MFMAs together - in-order issue: neither
// gets to its vector work while the pipe is busy - and then normalise row s + 1 together with the pipe idle.  Here the halves of the block run the two
// independent parts of a stage in OPPOSITE order (waves 0 .. NW/2-1: fragments, MFMAs, then the in-place pass of row s + 1; the others: in-place
// pass, then fragments and MFMAs; a workgroup's waves i and i + 4 share a SIMD), every LDS access is raw asm with counted waits, the zero rows
// above / below the image pass through the arithmetic with zero coefficients instead of a branch, and the fold of all branches runs at once.
// Full-width strips with BatchNorm + ReLU on load only (the form the engine records); everything else stays on conv_band32.
template <int NW, int R>
__device__ __forceinline__ void conv_band32s_body(const BandK& q) {
  typedef bf16_t T;
  constexpr int C = 32, NT = NW * 64, SW = NW * 32, HALO = 32, BR = 8;
  constexpr int SPX = SW + 2 * HALO, SLOT = SPX * 64;
  constexpr int NPX = 2;                                // SW * 64 / 1024 = 2 NW row DMA instructions: two per wave
  constexpr int WPIECES = 18, WST = (WPIECES + NW - 1) / NW;
  // the weights of branch b + 1 are issued in stages 2 .. WST+1 of the last phase of branch b and read after the barrier that ends it: at the wait of
  // that phase's last stage (2 NPX operations may remain) the row DMAs of stage 6 - 2 NPX of them, issued after the weights - lie behind them
  static_assert(WST + 2 <= BR - 1 && R == 7 && R <= BR, "pipeline depths");
  constexpr unsigned ROW_OOB = 0x80000000u;
  constexpr int TAB_BIAS = 4 * 64, TAB_ZERO = 4 * 64 + 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem + R * SLOT;
  float* tab = reinterpret_cast<float*>(sW + WPIECES * 1024);          // [nb][2][32] scale, shift ; [32] bias sum ; [64] zeros

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 31, kh = lane >> 5;
  const int H = q.H, W = q.W, nb = q.nb;
  const int nwg = q.njobs, bid = blockIdx.x;
  if (bid >= nwg) return;
  RUA_BTS(0);
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int band = job % q.bands, n_ = job / q.bands;
  const int h0 = band * BR;

  // ---- tables; the BatchNorm fold of ALL branches at once (conv_strip32s: the thread groups are dealt to the branches) --------------
  if (tid < 32) {
    const float b = ((ld_f32_or_zero(q.bias[0], tid, 32) + ld_f32_or_zero(nb > 1 ? q.bias[1] : nullptr, tid, 32)) + ld_f32_or_zero(nb > 2 ? q.bias[2] : nullptr, tid, 32)) +
                    ld_f32_or_zero(nb > 3 ? q.bias[3] : nullptr, tid, 32);
    tab[TAB_BIAS + tid] = b;
    tab[TAB_ZERO + tid] = 0.f; tab[TAB_ZERO + 32 + tid] = 0.f;
  }
  // ---- per-lane constants (conv_strip32s) ----------------------------------------------------------------------------------------
  const int psrc = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned smem_a = (unsigned)(size_t)(lds_void_p)smem;
  const unsigned tab_a = (unsigned)(size_t)(lds_void_p)tab + (unsigned)(psrc * 32);
  const unsigned sw_a = (unsigned)(size_t)(lds_void_p)sW + (unsigned)(lane * 16);
  const unsigned rowbytes = (unsigned)(W * C * 2), imgbase = (unsigned)(n_ * H) * rowbytes;
  const unsigned xrel = (unsigned)(((wv * 16 + (lane >> 2)) * C + psrc * 8) * 2);
  const unsigned pdst = (unsigned)(HALO * 64 + wv * 1024 + lane * 16);
  const int o = wv * 32 + pl;
  const int d0 = q.d[0], d1 = q.d[1], d2 = q.d[2], d3 = q.d[3];
  // ---- the weights of branch 0 and the rows 1 .. R-2 of phase 0 go out BEFORE the fold (its scratch, 8 KB, lies in slot 0; the row DMAs write the
  //      interior pixels of their slots only, the halo zeroing below the others): the ring fill and the fold's dependent loads share one memory round
  //      trip instead of following one another; row 0 is issued in the prologue of run()
  {
    const __amdgpu_buffer_rsrc_t rw0 = make_rsrc(q.w[0], (unsigned)(9 * C * C * 2));
#pragma unroll
    for (int i = 0; i < WST; ++i) {
      const int idx = i * NW + wv;
      if (idx < WPIECES) {
        const unsigned off = (unsigned)((((idx >> 1) * C * C) + pl * C + (idx & 1) * 16 + kh * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw0, (lds_void_p)(sW + idx * 1024), 16, off, 0, 0, 0);
      }
    }
    const __amdgpu_buffer_rsrc_t rx0 = make_rsrc(q.x[0], q.xbytes);
    const int hb0 = h0 - d0;                            // phase 0: kernel row 0 of branch 0
#pragma unroll
    for (int s_ = 1; s_ < R - 1; ++s_) {
      const unsigned base = ((unsigned)(hb0 + s_) < (unsigned)H) ? imgbase + (unsigned)(hb0 + s_) * rowbytes : ROW_OOB;
#pragma unroll
      for (int k_ = 0; k_ < NPX; ++k_)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_void_p)(smem + s_ * SLOT + HALO * 64 + (k_ * NW + wv) * 1024), 16, (base + (unsigned)(k_ * NW * 1024)) + xrel, 0, 0, 0);
    }
  }
  if (q.has_fold) {
    rua_fold_members<NT, 32>([&](int m) -> const rua_bn_fold& { return q.f[m]; }, [](int) { return true; }, nb, job == 0, reinterpret_cast<double*>(smem), tid,
                             [&](int m, int c, float scf, float shf) { tab[m * 64 + c] = scf; tab[m * 64 + 32 + c] = shf; });
  } else if (tid < 32) {
    for (int i = 0; i < nb; ++i) { tab[i * 64 + tid] = q.in_scale[i][tid]; tab[i * 64 + 32 + tid] = q.in_shift[i] ? q.in_shift[i][tid] : 0.f; }
  }
  __syncthreads();                                      // tables; the fold's scratch (the ring) is dead
  for (int i = tid; i < R * 2 * (HALO * 64 / 16); i += NT) {            // the halo pixels of every slot: zero once, never written again
    const int sl = i / (2 * HALO * 4), k = i - sl * (2 * HALO * 4);
    unsigned char* z = smem + sl * SLOT + (k < HALO * 4 ? k * 16 : (HALO + SW) * 64 + (k - HALO * 4) * 16);
    *reinterpret_cast<uint4*>(z) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  RUA_BTS(1);

  auto dil_of = [&](int b) { return b == 0 ? d0 : (b == 1 ? d1 : (b == 2 ? d2 : d3)); };
#ifdef RUA_BAND_ABLATE
  const bool dbg_nox = (q.dbg & 4) != 0, dbg_notr = (q.dbg & 2) != 0, dbg_nomfma = (q.dbg & 16) != 0;
#else
  constexpr bool dbg_nox = false, dbg_notr = false, dbg_nomfma = false;
#endif

  // everything from here on exists once per half of the block
  auto run = [&](auto GBc) {
    constexpr bool GB = decltype(GBc)::value;
    struct Phase { int hb; bool valid; __amdgpu_buffer_rsrc_t rx; unsigned ca; };
    auto phase = [&](int ph) {
      Phase p;
      const int b = ph / 3, ty = ph - 3 * b;
      p.valid = ph < 3 * nb;
      p.hb = h0 + (ty - 1) * dil_of(b);
      p.rx = make_rsrc(q.x[p.valid ? b : 0], q.xbytes);
      p.ca = tab_a + (unsigned)((p.valid ? b : 0) * 64 * 4);
      return p;
    };
    auto row_valid = [&](const Phase& p, int r) { return p.valid && (unsigned)(p.hb + r) < (unsigned)H; };         // block-uniform
    auto issue_x = [&](const Phase& p, int r, unsigned so) {
      const unsigned base = row_valid(p, r) ? imgbase + (unsigned)(p.hb + r) * rowbytes : ROW_OOB;
#pragma unroll
      for (int k = 0; k < NPX; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(p.rx, (lds_void_p)(smem + so + HALO * 64 + (k * NW + wv) * 1024), 16, (base + (unsigned)(k * NW * 1024)) + xrel, 0, 0, 0);
    };
    auto issue_w = [&](int b, int idx) {
      if (idx < WPIECES && b < nb) {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(q.w[b], (unsigned)(9 * C * C * 2));
        const unsigned off = (unsigned)((((idx >> 1) * C * C) + pl * C + (idx & 1) * 16 + kh * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_p)(sW + idx * 1024), 16, off, 0, 0, 0);
      }
    };
    // the in-place pass of a landed row on this thread's own two DMA pieces; the coefficients of the lane's eight channels come with the pieces
    // (broadcast reads); a row outside the image reads zeros for them: relu(0 * 0 + 0) leaves the zero padding zero
    struct TrRegs { u32x4_t r0, r1; f32x4 sa, sb, ha, hb; };
    auto tr_read = [&](unsigned so, unsigned ca, TrRegs& t) {
      const unsigned a = smem_a + so + pdst;
      asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:%c8\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
                   "ds_read_b128 %4, %7 offset:128\n\tds_read_b128 %5, %7 offset:144"
                   : "=&v"(t.r0), "=&v"(t.r1), "=&v"(t.sa), "=&v"(t.sb), "=&v"(t.ha), "=&v"(t.hb) : "v"(a), "v"(ca), "n"(NW * 1024) : "memory");
    };
    auto tr_wait0 = [&](TrRegs& t) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.r0), "+v"(t.r1), "+v"(t.sa), "+v"(t.sb), "+v"(t.ha), "+v"(t.hb) :: "memory"); };
    auto tr_wait6 = [&](TrRegs& t) { asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(t.r0), "+v"(t.r1), "+v"(t.sa), "+v"(t.sb), "+v"(t.ha), "+v"(t.hb) :: "memory"); };
    auto tr_math1 = [&](u32x4_t& rw, const TrRegs& t) {
      float f[8];
      ET<T>::unpack(make_uint4(rw[0], rw[1], rw[2], rw[3]), f);
      f[0] = fmaf(t.sa[0], f[0], t.ha[0]); f[1] = fmaf(t.sa[1], f[1], t.ha[1]); f[2] = fmaf(t.sa[2], f[2], t.ha[2]); f[3] = fmaf(t.sa[3], f[3], t.ha[3]);
      f[4] = fmaf(t.sb[0], f[4], t.hb[0]); f[5] = fmaf(t.sb[1], f[5], t.hb[1]); f[6] = fmaf(t.sb[2], f[6], t.hb[2]); f[7] = fmaf(t.sb[3], f[7], t.hb[3]);
      typedef __attribute__((ext_vector_type(2))) short s16x2;
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      const s16x2 z = {0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2_t p2 = {f[2 * k], f[2 * k + 1]};
        const bf16x2_t b2 = __builtin_convertvector(p2, bf16x2_t);
        rw[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, b2), z));
      }
    };
    auto tr_write = [&](unsigned so, TrRegs& t) {
      const unsigned a = smem_a + so + pdst;
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%c3" :: "v"(a), "v"(t.r0), "v"(t.r1), "n"(NW * 1024) : "memory");
    };
    auto frag_read = [&](unsigned so, const unsigned* boff, bf16x8* f) {
      unsigned a[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k] = (boff[k >> 1] + so) ^ ((k & 1) ? 32u : 0u);
      asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
                   : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]) : "memory");
    };
    auto f_wait0 = [&](bf16x8* f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) :: "memory"); };
    auto f_wait6 = [&](bf16x8* f) { asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) :: "memory"); };

    f32x16 acc[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;

    // The pipeline of stage s (row r of phase ph), between barrier s and barrier s + 1 - nothing in it waits for an LDS read it has just issued:
    //   MFMAs of row s           fragments in registers since stage s - 1
    //   fragment reads, row s+1  its in-place pass was published by barrier s
    //   in-place pass, row s+2   arithmetic on the pieces read at the end of stage s - 1, written back (published by barrier s + 1)
    //   reads for the pass, s+3  own DMA pieces: behind the wave's own counted wait, no barrier needed
    //   DMA issue, row s+R       into the slot of row s (its fragments were read by everyone before barrier s)
    // first half: MFMAs, fragment reads, then the pass; second half: the pass first.
    struct Coef { f32x4 sa, sb, ha, hb; };
    auto coef_read = [&](unsigned ca, Coef& c) {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c.sa), "=&v"(c.sb), "=&v"(c.ha), "=&v"(c.hb) : "v"(ca) : "memory");
    };
    auto row_read = [&](unsigned so, u32x4_t& r0, u32x4_t& r1) {
      const unsigned a = smem_a + so + pdst;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%c3" : "=&v"(r0), "=&v"(r1) : "v"(a), "n"(NW * 1024) : "memory");
    };
    auto row_math1 = [&](u32x4_t& rw, const Coef& c) {
      float f[8];
      ET<T>::unpack(make_uint4(rw[0], rw[1], rw[2], rw[3]), f);
      f[0] = fmaf(c.sa[0], f[0], c.ha[0]); f[1] = fmaf(c.sa[1], f[1], c.ha[1]); f[2] = fmaf(c.sa[2], f[2], c.ha[2]); f[3] = fmaf(c.sa[3], f[3], c.ha[3]);
      f[4] = fmaf(c.sb[0], f[4], c.hb[0]); f[5] = fmaf(c.sb[1], f[5], c.hb[1]); f[6] = fmaf(c.sb[2], f[6], c.hb[2]); f[7] = fmaf(c.sb[3], f[7], c.hb[3]);
      typedef __attribute__((ext_vector_type(2))) short s16x2;
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      const s16x2 z = {0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2_t p2 = {f[2 * k], f[2 * k + 1]};
        const bf16x2_t b2 = __builtin_convertvector(p2, bf16x2_t);
        rw[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, b2), z));
      }
    };
    auto row_write = [&](unsigned so, u32x4_t& r0, u32x4_t& r1) {
      const unsigned a = smem_a + so + pdst;
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%c3" :: "v"(a), "v"(r0), "v"(r1), "n"(NW * 1024) : "memory");
    };
    auto lds_wait_rows = [&](u32x4_t& r0, u32x4_t& r1) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1) :: "memory"); };
    auto boff_of = [&](int d, unsigned* boff) {          // b-operand fragments: tap column tx reads slot pixel 32 + o + (tx - 1) d (k-step 1: address ^ 32)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int j = HALO + o + (tx - 1) * d;
        boff[tx] = smem_a + (unsigned)(j * 64 + ((kh ^ ((j >> 2) & 3)) * 16));
      }
    };
    auto slot_next = [&](unsigned so) { return so + SLOT == (unsigned)(R * SLOT) ? 0u : so + SLOT; };

    // ---- prologue: weights of branch 0 and rows 0 .. R-2 in flight, everything landed; rows 0 .. 2 normalised, row 3 read --------------
    Phase cur = phase(0), nxt = phase(1);
    issue_x(cur, 0, 0u);                                // (the weights of branch 0 and the rows 1 .. R-2 went out in front of the fold)
    Coef cf;
    coef_read(cur.ca, cf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32x4_t ta0, ta1, tb0, tb1;                          // the pieces of the row being normalised / of the row after it
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (row_valid(cur, s) && !dbg_notr) {
        row_read((unsigned)(s * SLOT), ta0, ta1); lds_wait_rows(ta0, ta1);
        row_math1(ta0, cf); row_math1(ta1, cf); row_write((unsigned)(s * SLOT), ta0, ta1);
      }
    row_read(3u * SLOT, ta0, ta1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fx[6];
    {
      unsigned boff[3];
      boff_of(d0, boff);
      frag_read(0u, boff, fx);
      f_wait0(fx);
    }
    RUA_BTS(2);

    // slot byte offsets of the rows s - 1 and s (refilled with the rows s + R - 1 and s + R at the even stages), s + 1 (fragments), s + 3 (written),
    // s + 4 (read)
    unsigned so_m1 = (R - 1) * SLOT, so_0 = 0, so1 = SLOT, so2 = 2 * SLOT, so3 = 3 * SLOT, so4 = 4 * SLOT;
    for (int ph = 0; ph < 3 * nb; ++ph) {
      const int b = ph / 3, ty = ph - 3 * b;
      unsigned boff[3], boffn[3];                        // ... of this phase and of the next one (row s + 1 of the last stage belongs to it)
      boff_of(dil_of(b), boff);
      boff_of(dil_of(ty == 2 ? b + 1 : b), boffn);
      bf16x8 wf[6];
      auto valid_at = [&](int k) { return (k < BR ? row_valid(cur, k) : row_valid(nxt, k - BR)) && !dbg_notr; };       // row k of this phase, k < 2 BR; block-uniform
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        // ONE barrier per two stages: what stage s reads of other waves' work - the in-place pass of row s + 1 - was written two stages
        // earlier, i.e. before the last barrier whichever parity s has; the slots of the rows s - 1 and s are refilled behind the barrier of
        // an even stage s (their fragments were read in the stages s - 2 and s - 1)
        if ((r & 1) == 0) __builtin_amdgcn_s_barrier();
        if (r == 0) {
          // kernel row ty of this branch's weights (issued during the last phase of the branch before, waited out by every wave, published by the barrier)
          const unsigned wa = sw_a + (unsigned)(ty * 6 * 1024);
          asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\tds_read_b128 %3, %6 offset:3072\n\t"
                       "ds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]), "=&v"(wf[3]), "=&v"(wf[4]), "=&v"(wf[5]) : "v"(wa) : "memory");
        }
        if (r == 5 && ty == 2 && b + 1 < nb) coef_read(nxt.ca, cf);     // rows s + 3 of the stages r >= 5 belong to the next phase: a new branch
        // the next branch's weights from stage 2 on: the pieces of kernel row 2 are overwritten by the second round of instructions, and another
        // wave may still be reading them at stage 0 until the barrier of stage 2
        if (r >= 2 && r < 2 + WST && ty == 2) issue_w(b + 1, (r - 2) * NW + wv);
        if ((r & 1) == 0 && !dbg_nox) {
          if (r + R - 1 < BR) issue_x(cur, r + R - 1, so_m1); else issue_x(nxt, r + R - 1 - BR, so_m1);
          if (r + R < BR) issue_x(cur, r + R, so_0); else issue_x(nxt, r + R - BR, so_0);
        }
        const bool tvm = valid_at(r + 3), tvr = valid_at(r + 4);
        u32x4_t& m0 = (r & 1) ? tb0 : ta0; u32x4_t& m1 = (r & 1) ? tb1 : ta1;       // read at the end of the stage before
        u32x4_t& n0 = (r & 1) ? ta0 : tb0; u32x4_t& n1 = (r & 1) ? ta1 : tb1;
        auto mfmas = [&]() {
          if (dbg_nomfma) return;
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k], fx[k], acc[r], 0, 0, 0);
        };
        auto pass = [&]() {
          // row s + 4 has landed (this wave's pieces): behind it lie the rows s + 5 .. s + 7 at an even stage, s + 5 and s + 6 at an odd one
          if ((r & 1) == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NPX) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPX) : "memory");
          if (tvr) row_read(so4, n0, n1);
          if (tvm) { row_math1(m0, cf); row_math1(m1, cf); row_write(so3, m0, m1); }
        };
        if constexpr (!GB) {
          mfmas();
          __builtin_amdgcn_sched_barrier(0);
          frag_read(so1, r + 1 < BR ? boff : boffn, fx);
          pass();
        } else {
          pass();
          __builtin_amdgcn_sched_barrier(0);
          mfmas();
          __builtin_amdgcn_sched_barrier(0);
          frag_read(so1, r + 1 < BR ? boff : boffn, fx);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]), "+v"(fx[3]), "+v"(fx[4]), "+v"(fx[5]), "+v"(n0), "+v"(n1) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        so_m1 = so_0; so_0 = so1; so1 = so2; so2 = so3; so3 = so4; so4 = slot_next(so4);
      }
      cur = nxt;
      nxt = phase(ph + 2);
      RUA_BTS(3 + ph);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");     // the over-issued DMAs of the last stages
    RUA_BTS(16);

    // ---- epilogue: bias sum + residual, one write of the band (conv_band32) ---------------------------------------------------------
    const size_t pix0 = (size_t)((n_ * H + h0) * W + o);
    uint4 rv[BR][2];
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        rv[r][g] = q.res ? ldg16(q.res + ((pix0 + (size_t)r * W) * C + 16 * g + 8 * kh) * 2) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      float v[2][8];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = acc[r][(2 * g) * 4 + j], b2 = acc[r][(2 * g + 1) * 4 + j];
          if (g == 0 && j == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
          else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
          v[g][j] = a;
          v[g][4 + j] = b2;
        }
      unsigned char* yrow = q.y + ((pix0 + (size_t)r * W) * C) * 2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int co = 16 * g + 8 * kh;
        float a8[8];
        ET<T>::unpack(rv[r][g], a8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[g][j] += tab[TAB_BIAS + co + j] + a8[j];
        stg16(yrow + co * 2, ET<T>::pack(v[g]));
      }
    }
    RUA_BTS(17);
  };
  if (wv < NW / 2 || (q.dbg & 32)) run(std::false_type{});
  else run(std::true_type{});
}

template <int NW, int R> __global__ __launch_bounds__(NW * 64) void conv_band32s(const BandK q) { conv_band32s_body<NW, R>(q); }

// ---- host side ---------------------------------------------------------------------------------------------------------
static thread_local int g_sum_last_kernel = 0;
extern "C" int rua_conv_sum_last_kernel(void) { return g_sum_last_kernel; }   // the calling thread's latest rua_conv_fwd_sum: 1 conv_band32, 2 conv_band64, 0 member by member

static bool band_ok(const rua_conv_desc* d, int n) {
  if (!g_tune.conv_band || n < 1 || n > RUA_MAX_BRANCH) return false;
  const rua_conv_desc& a = d[0];
  if (a.dtype != RUA_BF16 || a.W % 128 != 0 || a.H % 8 != 0 || (long long)a.N * a.H * a.W < 65536) return false;
  if (a.aux_mode != 0 && a.aux_mode != 1) return false;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    const rua_conv_seg& g = m.seg[0];
    if (m.nseg != 1 || m.dtype != RUA_BF16 || g.taps != 9 || g.up_shift != 0 || g.C != 32 || m.Cout != 32 || m.stride != 1 ||
        m.out_stride != 1 || m.OH != m.H || m.OW != m.W || g.Hs != m.H || g.Ws != m.W || g.dil < 1 || g.dil > 32) return false;
    if (m.N != a.N || m.H != a.H || m.W != a.W || m.y != a.y) return false;
    if (m.stats_mode != 0 || m.out_relu != 0 || m.bias_more[0] || m.bias_more[1] || m.bias_more[2]) return false;
    if (i > 0 && (m.aux_mode != 0 || !m.accumulate)) return false;
    if (i == 0 && m.accumulate) return false;
    if ((m.in_fold != nullptr) != (a.in_fold != nullptr) || (m.in_scale != nullptr) != (a.in_scale != nullptr) || m.in_relu != a.in_relu) return false;
    if (m.in_fold && (m.in_scale || m.in_shift)) return false;
  }
  return true;
}

// which kernel rua_conv_fwd_sum would run for these members, without launching: 0 member by member, 1 conv_band32, 2 conv_band64
extern "C" int rua_conv_sum_kernel(const rua_conv_desc* d, int n) {
  if (!d || n < 1 || n > RUA_MAX_BRANCH) return 0;
  return rua_band64_ok(d, n) ? 2 : (band_ok(d, n) ? 1 : 0);
}

extern "C" int rua_conv_fwd_sum(const rua_conv_desc* d, int n, void* stream) {
  RUA_CHECK_ARG(d && n >= 1 && n <= RUA_MAX_BRANCH, "rua_conv_fwd_sum: 1..%d members", RUA_MAX_BRANCH);
  hipStream_t st = (hipStream_t)stream;
  g_sum_last_kernel = 0;
  if (rua_band64_ok(d, n)) {                              // the C = 64 level (conv_band64.hip)
    const int rc = rua_launch_band64(d, n, st);
    if (rc == RUA_OK) g_sum_last_kernel = 2;
    return rc;
  }
  if (!band_ok(d, n)) {                                   // same results from the members' own launches (member i > 0 accumulates)
    for (int i = 0; i < n; ++i) {
      RUA_CHECK_ARG(d[i].y == d[0].y && (i == 0 || d[i].accumulate), "rua_conv_fwd_sum: members after the first must accumulate into the same output");
      const int rc = rua_conv_fwd(d + i, stream);
      if (rc != RUA_OK) return rc;
    }
    return RUA_OK;
  }
  BandK q;
  memset(&q, 0, sizeof(q));
  q.nb = n;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    q.x[i] = (const unsigned char*)m.seg[0].x; q.w[i] = (const unsigned char*)m.seg[0].w; q.bias[i] = m.bias;
    q.in_scale[i] = m.in_scale; q.in_shift[i] = m.in_shift; q.d[i] = m.seg[0].dil;
    if (m.in_fold) {
      q.f[i] = *m.in_fold;
      const rua_bn_fold& f = q.f[i];
      RUA_CHECK_ARG(f.stats && f.replicas >= 1 && f.count > 0 && f.gamma && f.beta && f.scale && f.shift, "rua_conv_fwd_sum: incomplete in_fold");
      RUA_CHECK_ARG((f.moving_mean == nullptr) == (f.moving_var == nullptr), "rua_conv_fwd_sum: in_fold needs both moving statistics or neither");
    }
  }
  for (int i = n; i < RUA_MAX_BRANCH; ++i) q.d[i] = 1;
  const rua_conv_desc& a = d[0];
  q.dbg = g_tune.band_dbg;
  q.has_fold = a.in_fold ? 1 : 0;
  q.has_bn = (a.in_fold || a.in_scale) ? 1 : 0;
  q.in_relu = a.in_relu;
  q.res = a.aux_mode == 1 ? (const unsigned char*)a.aux : nullptr;
  q.y = (unsigned char*)a.y;
  q.N = a.N; q.H = a.H; q.W = a.W;
  q.xbytes = (unsigned)((size_t)a.N * a.H * a.W * 32 * 2);
  RUA_CHECK_ARG((size_t)a.N * a.H * a.W * 32 * 2 < 0x80000000ull, "rua_conv_fwd_sum: tensor of 2 GiB or more");
  constexpr int BR = 8, R = 4;                          // a 4-slot ring: 2 rows in flight measured FASTER than 4 (6 slots): 78 - 82 vs 84 - 93 us
  const int sw = a.W % 256 == 0 ? 256 : 128, nw = sw / 32;
  q.strips = a.W / sw;
  q.bands = a.H / BR;
  q.njobs = a.N * q.strips * q.bands;
  const int smem = R * (sw + 64) * 64 + 18 * 1024 + nw * 1024 + (4 * 64 + 32) * 4;
  RUA_CHECK_ARG(smem <= 160 * 1024 && smem >= nw * 2 * 2 * 32 * 8, "conv_band: %d bytes of LDS", smem);
  const bool fullw = q.strips == 1;
  static RuaPerDevFlag attr[4];                         // per device (a hipFuncSetAttribute is): not per thread
#define RUA_BAND_GO(NW_, FW_, SLOT_) do { \
    if (!attr[SLOT_].get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band32<NW_, 8, 4, FW_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[SLOT_].get() = true; } \
    hipLaunchKernelGGL((conv_band32<NW_, 8, 4, FW_>), dim3(q.njobs), dim3(NW_ * 64), smem, st, q); } while (0)
  if (g_tune.band_stag && fullw && q.has_bn && q.in_relu) {
    // conv_band32s: R slots by tuning key (band_stag = number of ring slots; 1 = the default of the shape)
    const int want = g_tune.band_stag >= 4 ? g_tune.band_stag : (nw == 8 ? 7 : 6);
    static RuaPerDevFlag sattr[4];
#define RUA_BANDS_GO(NW_, R_, SLOT_) do { \
      const int smem_s = R_ * (sw + 64) * 64 + 18 * 1024 + (4 * 64 + 32 + 64) * 4; \
      bool& a_ = sattr[SLOT_].get(); \
      if (!a_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band32s<NW_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a_ = true; } \
      hipLaunchKernelGGL((conv_band32s<NW_, R_>), dim3(q.njobs), dim3(NW_ * 64), smem_s, st, q); } while (0)
    (void)want;
    if (nw == 8) RUA_BANDS_GO(8, 7, 0);
    else RUA_BANDS_GO(4, 7, 2);
#undef RUA_BANDS_GO
  } else if (nw == 8 && fullw && (q.dbg & 8)) {                 // experiment: the 6-slot ring (4 rows in flight instead of 2)
    static RuaPerDevFlag a6f;
    bool& a6 = a6f.get();
    const int smem6 = 6 * (sw + 64) * 64 + 18 * 1024 + nw * 1024 + (4 * 64 + 32) * 4;
    if (!a6) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band32<8, 8, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a6 = true; }
    hipLaunchKernelGGL((conv_band32<8, 8, 6, true>), dim3(q.njobs), dim3(512), smem6, st, q);
  } else if (nw == 8 && fullw) RUA_BAND_GO(8, true, 0);
  else if (nw == 8) RUA_BAND_GO(8, false, 1);
  else if (fullw) RUA_BAND_GO(4, true, 2);
  else RUA_BAND_GO(4, false, 3);
#undef RUA_BAND_GO
  RUA_LAUNCH_CHECK("conv_band32");
  g_sum_last_kernel = 1;
  return RUA_OK;
}

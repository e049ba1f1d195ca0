// conv_strip<NW>: streaming 3x3 dilated convolution for the top level of the network (C = Cout = 32, bf16): the d6 residual
// atrous block of the north star (ResBlock(32,[1,3,15,31]) at full resolution, model2.py:102) and its data gradients.
//
// Those convolutions move 67-100 MB for 9.7 GFLOP: their floor is the HBM time of ONE read of the input and ONE write of the
// output.  conv_halo (conv_mfma.hip) reaches the minimum byte count but runs tile by tile - load, wait, compute, transpose,
// store are serial inside a block - at 2-3 TB/s.  Here a block STREAMS: it owns a strip of SW = 32 * NW pixels of one image
// and walks down the rows h = r, r + d, r + 2d, ... of one residue class mod d (for those rows the dilated conv is a plain
// 3x3 conv over a sliding window of three rows), so that
//   * every input row enters LDS once, by LDS-DMA (HBM -> LDS, no registers), into a ring of R row slots; the DMA of row
//     i + R - 2 is issued at stage i and waited for with a COUNTED vmcnt at stage i + R - 3: R - 3 rows (40-60 KB per CU)
//     are in flight all the time, across the one barrier per stage;
//   * the BatchNorm + ReLU that precedes the conv in the reference graph (model2.py:17-24) is applied as the row lands: the
//     wave that issued a DMA rewrites its own 16-byte pieces in place (scale * x + shift, ReLU; zero padding stays zero)
//     before the barrier - the normalised copy of the input is never written to HBM;
//   * the MFMA runs the TRANSPOSED product D^T[co][px] = W[co][k] * X^T[k][px] (weights = a-operand, in registers for the
//     whole kernel; pixels = b-operand, one ds_read_b128 per tap and k-step), so an accumulator holds 4-channel groups of ONE
//     pixel per lane and one 32-lane exchange per register pair gives 8-channel pieces: bias, mask / residual / accumulate,
//     ReLU, statistics and the 16-byte stores happen in registers, no LDS round trip and no second barrier;
//   * the tensor the epilogue needs per pixel (ReLU-mask source of a data gradient, residual, or the old output when
//     accumulating) streams through a second LDS ring the same way.
// Every vector-memory operation of the loop is either an LDS-DMA or a store, issued unconditionally by every wave, so the
// counted waits are exact (loads, stores and LDS-DMA retire in issue order on one counter: MI355X_MICROARCH.md).
// Measured (tools/bench_conv3x3.py, 8 x 256 x 256 x 32): 23-24 us plain, 31-33 us with BatchNorm on load + statistics or with
// a ReLU mask, at every dilation (conv_halo: 25-30 / 31-38 without the BatchNorm) - in-kernel s_memtime stamps put the
// streaming part within ~25 % of the HBM time of its bytes (a block reads 8 rows + 2 window-fill rows).  A ping-pong variant
// (waves w / w + 4 of a SIMD alternate MFMA phase and epilogue, exact per-interval wait counts) overlapped the two phases as
// designed and was no faster (24.4 us plain, step 9.88 vs 9.78 ms): the rows arrive no sooner.
#include "common.h"
#include <type_traits>

struct StripK {
  ConvK c;
  const unsigned char* ep;                // epilogue stream: aux (aux_mode 1 / 2) or the old output (accumulate); null: none
  unsigned epbytes;
  int d, strips, spc, seglen, njobs, nchains;
  int slot_bytes, SWH;                    // bytes of one x-ring slot (multiple of 1024), pixels per slot row (SW + 2 d)
  int dbg;                                // experiments (tuning key band_dbg; conv_strip32s): 1 no stages (prologue + tail only), 2 no row DMAs in the prologue
  int has_fold; rua_bn_fold f;            // BatchNorm coefficients derived in the prologue from the input's statistics (in_fold)
};

template <int NW, bool HAS_EP, int R>
__device__ __forceinline__ void conv_strip32_body(const StripK& q) {
  typedef bf16_t T;
  constexpr int C = 32, NT = NW * 64, SW = NW * 32;
  constexpr int NPX = 3, NPA = HAS_EP ? 2 : 0, RA = 3, NST = 2;
  constexpr int PER = NPX + NPA + NST;
  // ops issued after the youngest DMA a stage needs: x(i+1) was issued at stage i - (R - 3), ep(i) at stage i - (RA - 2)
  constexpr int NWX = NPA + NST + (R - 4) * PER, NWE = NST + (RA - 2) * PER;
  constexpr int NWAIT = HAS_EP ? (NWX < NWE ? NWX : NWE) : NWX;
  constexpr int EP_SLOT = SW * 64;
  constexpr unsigned OOB = 0x80000000u;
  const ConvK& p = q.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sEP = smem + R * q.slot_bytes;
  unsigned char* sDump = sEP + (HAS_EP ? RA * EP_SLOT : 0);
  float* tab = reinterpret_cast<float*>(sDump + NW * 1024);         // [5][32]: bias sum, mask scale, mask shift, folded BN scale, shift

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = lane & 31, kh = lane >> 5;
  const int H = p.H, W = p.W, d = q.d;

  // ---- job: (image, strip, residue class, segment of the chain) ------------------------------------------------------
  // (grouped launch: the grid is the largest member's, padded to a multiple of 8 so that block x of every member runs on XCD
  // x & 7; the renumbering below is over THIS member's jobs - over the whole grid a short member would land on the first XCDs only)
  const int nwg = q.njobs, bid = blockIdx.x;
  if (bid >= nwg) return;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);     // neighbours in a chain share an XCD's L2
  const int chain = job / q.spc, seg = job - chain * q.spc;
  const int r_ = chain % d, tq = chain / d;
  const int x0 = (tq % q.strips) * SW, n_ = tq / q.strips;
  const int ny = (H - r_ + d - 1) / d;
  const int i0 = seg * q.seglen;
  int nit = ny - i0; if (nit > q.seglen) nit = q.seglen; if (nit < 0) nit = 0;

  // ---- per-thread constants --------------------------------------------------------------------------------------------
  if (tid < 32) {
    float b = 0.f, ms = 1.f, mt = 0.f;
    if (p.bias) { b = p.bias[tid]; for (int r = 0; r < 3; ++r) if (p.bias_more[r]) b += p.bias_more[r][tid]; }
    if (p.aux_mode == 2) { if (p.mscale) ms = p.mscale[tid]; if (p.mshift) mt = p.mshift[tid]; }
    tab[tid] = b; tab[32 + tid] = ms; tab[64 + tid] = mt;
  }
  // weights: a-operand row = output channel pl, k-slice = input channels ks * 16 + kh * 8 .. + 8, all nine taps
  bf16x8 wf[9][2];
  {
    const unsigned char* wl = p.seg[0].w + ((size_t)pl * C + kh * 8) * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const bf16x8*>(wl + ((size_t)t * C * C + ks * 16) * 2);
  }
  // the 16-byte piece this lane moves in every DMA instruction holds input channels psrc * 8 .. + 8 (slot = piece ^ ((pixel >> 2) & 3):
  // conflict-free ds_read_b128 of the b-operand - the four pixels of one residue mod 4 in a 16-lane read group get four
  // different slots, at every tap shift; the permutation is applied on the SOURCE side, the LDS image is lane-linear)
  const int psrc = (lane & 3) ^ ((lane >> 4) & 3);
  const bool bn = p.in_scale != nullptr || q.has_fold;
  float sc8[8], sh8[8];
  if (q.has_fold) {
    // in_fold: the coefficient launch folded into this prologue.  NT / 32 thread groups each sum a share of the replicated fp64
    // statistics (one or two loads deep), the first 32 threads add the shares in a fixed order and finish mean / variance ->
    // scale / shift; job 0 publishes them and updates the moving statistics.  The x ring is still empty: it serves as scratch.
    const rua_bn_fold& f = q.f;
    constexpr int NG = NT / 32;
    double* red = reinterpret_cast<double*>(smem);                   // [NG][2][32]
    const int c = tid & 31, grp = tid >> 5;
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
      const double r = 1.0 / sqrt(v + (double)f.eps);
      const double sc = (double)f.gamma[tid] * r;
      const float scf = (float)sc, shf = (float)((double)f.beta[tid] - m * sc);
      tab[96 + tid] = scf; tab[128 + tid] = shf;
      if (job == 0) {
        f.scale[tid] = scf; f.shift[tid] = shf;
        if (f.mean) f.mean[tid] = (float)m;
        if (f.rstd) f.rstd[tid] = (float)r;
        if (f.moving_mean) {
          const double unb = f.bessel_n > 1 ? v * (f.bessel_n / (f.bessel_n - 1)) : v;
          f.moving_mean[tid] = (float)((double)f.moving_mean[tid] * f.momentum + m * (1.0 - f.momentum));
          f.moving_var[tid] = (float)((double)f.moving_var[tid] * f.momentum + unb * (1.0 - f.momentum));
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc8[j] = tab[96 + psrc * 8 + j]; sh8[j] = tab[128 + psrc * 8 + j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc8[j] = bn ? p.in_scale[psrc * 8 + j] : 1.f; sh8[j] = (bn && p.in_shift) ? p.in_shift[psrc * 8 + j] : 0.f; }
  }
  // all ordinary loads are consumed HERE, before the first LDS-DMA is issued (hipcc waits vmcnt(0) at the first use of a
  // pending load: inside the loop that would drain the DMA ring)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(wf[t][ks]));
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(sc8[j]), "+v"(sh8[j]));
  __syncthreads();                                      // tab
  float bias16[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) bias16[k] = tab[(k & 3) + 8 * (k >> 2) + 4 * kh];

  unsigned xrel[NPX]; int xdst[NPX]; bool xok[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int inst = k * NW + wv;                       // wave-instruction index inside the slot: 1 KiB = 16 pixels each
    const int j = inst * 16 + (lane >> 2);              // pixel of the slot row (0 .. SWH-1), image column x0 - d + j
    const int x = x0 - d + j;
    const bool in_slot = inst * 1024 < q.slot_bytes;    // wave-uniform
    xdst[k] = in_slot ? inst * 1024 : -1;               // -1: the instruction still issues (uniform op counts) into the wave's dump KiB
    xok[k] = in_slot && j < q.SWH && x >= 0 && x < W;
    xrel[k] = (unsigned)((x * C + psrc * 8) * 2);
  }
  unsigned erel[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) erel[k] = (unsigned)(((x0 + (k * NW + wv) * 16 + (lane >> 2)) * C + psrc * 8) * 2);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.seg[0].x, p.seg[0].xbytes);
  const __amdgpu_buffer_rsrc_t re = make_rsrc(HAS_EP ? q.ep : p.seg[0].x, HAS_EP ? q.epbytes : 0u);

  auto row_ok = [&](int rho) { const int h = r_ + (i0 + rho) * d; return rho <= nit && h >= 0 && h < H; };
  auto x_slot = [&](int rho) { return smem + ((rho + 1 + R) % R) * q.slot_bytes; };        // rho >= -1
  auto issue_x = [&](int rho) {
    const bool ok = row_ok(rho);
    const unsigned base = (unsigned)(((n_ * H + r_ + (i0 + rho) * d) * W) * C * 2);
    unsigned char* slot = x_slot(rho);
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      unsigned char* dst = xdst[k] >= 0 ? slot + xdst[k] : sDump + wv * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_p)dst, 16, (unsigned)((ok && xok[k]) ? base + xrel[k] : OOB), 0, 0, 0);
    }
  };
  auto issue_ep = [&](int e) {
    if constexpr (HAS_EP) {
      const bool ok = e < nit;
      const unsigned base = (unsigned)(((n_ * H + r_ + (i0 + e) * d) * W) * C * 2);
      unsigned char* slot = sEP + (e % RA) * EP_SLOT;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (lds_void_p)(slot + (k * NW + wv) * 1024), 16, (unsigned)(ok ? base + erel[k] : OOB), 0, 0, 0);
    }
  };
  // BatchNorm (+ ReLU) of a landed row, in place, on the pieces this thread's own DMA instructions wrote (padding stays zero)
  auto transform = [&](int rho) {
    if (!row_ok(rho)) return;
    unsigned char* slot = x_slot(rho);
    // all pieces are read first (one LDS round trip for the row instead of one per piece: piece by piece the dependent
    // read -> arithmetic -> write chains cost ~1000 cycles per row and wave), then normalised, then written back
    // RAW reads, one wait for the three: in front of a C++ LDS load hipcc waits for the LDS-DMAs issued before it (seen in the ISA
    // as `s_waitcnt vmcnt(2)` right behind the counted wait: everything but the last stage's two stores, i.e. the whole ring,
    // drained at every stage with BatchNorm on load); these pieces' own DMAs have landed (counted wait above)
    uint4 raw[NPX];
    {
      static_assert(NPX == 3, "the asm block reads three pieces");
      unsigned pa[NPX];
#pragma unroll
      for (int k = 0; k < NPX; ++k)
        pa[k] = (unsigned)(size_t)(lds_void_p)(xok[k] ? slot + xdst[k] + lane * 16 : sDump + wv * 1024 + lane * 16);
      u32x4_t rw0, rw1, rw2;
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(rw0), "=&v"(rw1), "=&v"(rw2) : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]) : "memory");
      raw[0] = make_uint4(rw0[0], rw0[1], rw0[2], rw0[3]); raw[1] = make_uint4(rw1[0], rw1[1], rw1[2], rw1[3]);
      raw[2] = make_uint4(rw2[0], rw2[1], rw2[2], rw2[3]);
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      float f[8];
      ET<T>::unpack(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaf(sc8[j], f[j], sh8[j]);
      raw[k] = ET<T>::pack(f);
      if (p.in_relu) {
        // ReLU on the packed pairs (the kernel is bound by vector issue, not by HBM): a negative bf16 is a negative int16, max with 0
        // clears it - one v_pk_max_i16 per pair instead of two v_max_f32; relu(round(x)) == round(relu(x))
        typedef __attribute__((ext_vector_type(2))) short s16x2;
        const s16x2 z = {0, 0};
        auto relu2 = [&](unsigned v) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z)); };
        raw[k].x = relu2(raw[k].x); raw[k].y = relu2(raw[k].y); raw[k].z = relu2(raw[k].z); raw[k].w = relu2(raw[k].w);
      }
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k)
      if (xok[k]) {
        // raw write: hipcc orders a C++ LDS store behind every LDS-DMA in flight (s_waitcnt vmcnt(0): the ring would drain at
        // every stage); this piece's own DMA has landed (counted wait above) and nobody else touches it before the barrier
        const u32x4_t pv = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
        const unsigned la = (unsigned)(size_t)(lds_void_p)(slot + xdst[k] + lane * 16);
        asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(pv) : "memory");
      }
  };

  // b-operand fragment offsets inside a slot row: output pixel o = wv * 32 + pl, tap column tx reads slot pixel o + tx * d
  const int o = wv * 32 + pl;
  int boff[3][2];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int j = o + tx * d;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[tx][ks] = j * 64 + (((ks * 2 + kh) ^ ((j >> 2) & 3)) * 16);
  }
  int eoff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) eoff[g] = o * 64 + (((2 * g + kh) ^ ((o >> 2) & 3)) * 16);

  float s1[2][8], s2[2][8];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[g][j] = 0.f; s2[g][j] = 0.f; }

  // ---- prologue: rows -1 .. R-3 and the first RA-1 epilogue rows in flight, then everything landed ------------------------
#pragma unroll
  for (int rho = -1; rho <= R - 3; ++rho) issue_x(rho);
#pragma unroll
  for (int e = 0; e < RA - 1; ++e) issue_ep(e);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bn) { transform(-1); transform(0); }

  for (int i = 0; i < nit; ++i) {
    if (i > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");       // row i + 1 and epilogue row i have landed (mine)
    if (bn) transform(i + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // everyone's; and every wave is done with stage i - 1
    issue_x(i + R - 2);                                  // into the slot of row i - 2
    issue_ep(i + RA - 1);                                // into the slot of epilogue row i - 1

    // the accumulator starts at the bias of its channels (acc[k]: channel (k & 3) + 8 * (k >> 2) + 4 * kh): the 16 adds of the
    // epilogue and their LDS table reads are gone
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = bias16[k];
    // the six fragments of window row ty + 1 are read while the six MFMAs of row ty execute (left alone, hipcc reads one
    // fragment, waits for it, issues one MFMA: the LDS latency 18 times per stage)
    bf16x8 fx[2][3][2];
    auto read_row = [&](int ty, bf16x8 (*f)[2]) {
      const unsigned char* row = x_slot(i - 1 + ty);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f[tx][ks] = *reinterpret_cast<const bf16x8*>(row + boff[tx][ks]);
    };
    read_row(0, fx[0]);
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      if (ty < 2) read_row(ty + 1, fx[(ty + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ty * 3 + tx][ks], fx[ty & 1][tx][ks], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // acc[r]: channel (r & 3) + 8 * (r >> 2) + 4 * kh of pixel pl; one half-wave exchange per register pair -> this lane holds
    // channels 16 g + 8 kh .. + 7 of its pixel (see conv_pw)
    float v[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[(2 * g) * 4 + j], b = acc[(2 * g + 1) * 4 + j];
        if (g == 0 && j == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v[g][j] = a;
        v[g][4 + j] = b;
      }
    const int h = r_ + (i0 + i) * d;
    unsigned char* yrow = p.y + ((size_t)((n_ * H + h) * W + x0 + o) * C) * 2;
    const unsigned char* erow = sEP + (i % RA) * EP_SLOT;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = 16 * g + 8 * kh;
      float a8[8];
      if constexpr (HAS_EP) {
        ET<T>::unpack(*reinterpret_cast<const uint4*>(erow + eoff[g]), a8);
        if (p.aux_mode == 0) {                            // the stream is the old output: accumulate
#pragma unroll
          for (int j = 0; j < 8; ++j) v[g][j] += a8[j];
        } else if (p.aux_mode == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[g][j] += a8[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[g][j] = (fmaf(tab[32 + co + j], a8[j], tab[64 + co + j]) > 0.f) ? v[g][j] : 0.f;
        }
      }
      if (p.out_relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[g][j] = fmaxf(v[g][j], 0.f);
      }
      const uint4 packed = ET<T>::pack(v[g]);
      if (p.stats_mode == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], v[g][j], s2[g][j]); }
      } else if (p.stats_mode == 2) {
        if constexpr (HAS_EP) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], a8[j], s2[g][j]); }
        }
      }
      stg16(yrow + co * 2, packed);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");       // the over-issued DMAs of the last stages
  if (p.stats_mode != 0) {
    // 32 partial sums per lane, to be folded over the 32 pixels of a half-wave: through LDS (the ring is dead) - every lane writes
    // its values as a column of [wave][value row][33], thread (wave, row) adds the row.  (Butterfly shuffles: 320 ds_bpermute per
    // wave, ~2 us at the end of every launch with statistics.)
    __syncthreads();
    float* sred = reinterpret_cast<float*>(smem);       // [NW][64][33]
    float* sw = sred + NW * 64 * 33;                    // [NW][64]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = (kh * 16 + g * 8 + j) * 2;
        sred[(wv * 64 + row) * 33 + pl] = s1[g][j];
        sred[(wv * 64 + row + 1) * 33 + pl] = s2[g][j];
      }
    __syncthreads();
    {
      const float* rowp = sred + tid * 33;              // tid = wave * 64 + row
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) { t0 += rowp[k]; t1 += rowp[k + 1]; t2 += rowp[k + 2]; t3 += rowp[k + 3]; }
      sw[tid] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    if (tid < 64) {
      const int c = tid >> 1, k = tid & 1;
      const int row = (((c >> 3) & 1) * 16 + (c >> 4) * 8 + (c & 7)) * 2 + k;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += sw[w * 64 + row];
      unsafeAtomicAdd(&p.stats[(size_t)(job & (p.stats_R - 1)) * 2 * C + k * C + c], (double)t);
    }
  }
}

struct StripKG { StripK k[RUA_MAX_BRANCH]; };
static_assert(sizeof(StripKG) <= 4096, "a grouped launch passes its members by value: HIP kernel arguments are limited to 4 KiB");
template <int NW, bool HAS_EP, int R> __global__ __launch_bounds__(NW * 64) void conv_strip32(const StripK q) { conv_strip32_body<NW, HAS_EP, R>(q); }
// grouped launch (rua_conv_fwd_group): the four dilation branches of the d6 block in one grid, blockIdx.y = branch
template <int NW, bool HAS_EP, int R> __global__ __launch_bounds__(NW * 64) void conv_strip32_g(const StripKG g) { conv_strip32_body<NW, HAS_EP, R>(g.k[blockIdx.y]); }

#include "conv_strip2.inc"

// ---- host side ---------------------------------------------------------------------------------------------------------
// strip width: 256 pixels (8 waves) where the row allows it; tuning key strip_narrow_maxd: dilations up to it take 128-pixel
// strips instead (twice the chains, half the window-fill overhead, four waves per block)
static int strip_width(const rua_conv_desc* d) {
  if (d->W % 256 == 0 && d->seg[0].dil > g_tune.strip_narrow_maxd) return 256;
  return d->W % 128 == 0 ? 128 : 0;
}

bool rua_pick_strip(const rua_conv_desc* d) {
  if (!g_tune.conv_strip || d->dtype != RUA_BF16 || d->nseg != 1) return false;
  const rua_conv_seg& g = d->seg[0];
  const int sw = strip_width(d);
  if (!(g.taps == 9 && g.up_shift == 0 && g.C == 32 && d->Cout == 32 && d->stride == 1 && d->out_stride == 1 && d->OH == d->H &&
        d->OW == d->W && g.Hs == d->H && g.Ws == d->W && sw != 0 && g.dil >= 1 && g.dil <= 31 && g.dil < d->H)) return false;
  if ((long long)d->N * d->H * d->W < 65536) return false;
  if (d->aux_mode == 3 || (d->aux_mode != 0 && d->accumulate)) return false;      // one epilogue stream
  if (d->stats_mode == 2 && d->aux_mode == 0) return false;
  return true;
}

// members captured while rua_conv_fwd_group runs its members' dispatch (see conv_mfma.hip)
struct StripCapture { int n; int variant[RUA_MAX_BRANCH]; int smem[RUA_MAX_BRANCH]; StripK k[RUA_MAX_BRANCH]; };
static thread_local StripCapture g_strip_cap = {0, {0}, {0}, {}};

template <int NW, bool HAS_EP, int R> static int launch_strip_members(const StripK* ks, const int* smems, int m, hipStream_t st) {
  static RuaPerDevFlag attr_, attr_g_;
  bool &attr = attr_.get(), &attr_g = attr_g_.get();
  if (m == 1) {
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip32<NW, HAS_EP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((conv_strip32<NW, HAS_EP, R>), dim3(ks[0].njobs), dim3(NW * 64), smems[0], st, ks[0]);
  } else {
    if (!attr_g) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip32_g<NW, HAS_EP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_g = true; }
    StripKG g;
    int grid = 0, smem = 0;
    for (int i = 0; i < m; ++i) { g.k[i] = ks[i]; if (ks[i].njobs > grid) grid = ks[i].njobs; if (smems[i] > smem) smem = smems[i]; }
    hipLaunchKernelGGL((conv_strip32_g<NW, HAS_EP, R>), dim3((grid + 7) / 8 * 8, m), dim3(NW * 64), smem, st, g);
  }
  RUA_LAUNCH_CHECK("conv_strip32");
  return RUA_OK;
}
// conv_strip32s: ONE grid for all members (a lone convolution is a group of one); the members' row stages form one line of cost units
// that the blocks - one per CU - cut into equal pieces (conv_strip2.inc)
template <int NW, int IN, int EP, int ST, int ORELU, int R, int BIAS> static int launch_strip_s(const StripK* ks, const int* smems, int m, hipStream_t st) {
  static RuaPerDevFlag attr_;
  bool& attr = attr_.get();
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip32s_g<NW, IN, EP, ST, ORELU, R, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip32s<NW, IN, EP, ST, ORELU, R, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  StripSG g;
  memset(&g, 0, sizeof(g));
  int smem = 0;
  g.nmem = m; g.N = ks[0].c.N; g.H = ks[0].c.H; g.dbg = ks[0].dbg;
  g.c0 = 3;                                             // a restart: three rows of DMA latency, their BatchNorm, two barriers - about three row stages
  g.tbytes = (unsigned)((size_t)ks[0].c.M * 64);
  int units = 0;
  for (int i = 0; i < m; ++i) {
    const StripK& k = ks[i];
    RUA_CHECK_ARG(k.c.N == g.N && k.c.H == g.H && k.c.M == ks[0].c.M, "conv_strip32s: the members of a group differ in shape");
    StripSM& M = g.m[i];
    M.x = k.c.seg[0].x; M.w = k.c.seg[0].w; M.y = k.c.y; M.ep = k.ep;
    M.bias = k.c.bias; for (int r = 0; r < 3; ++r) M.bias_more[r] = k.c.bias_more[r];
    M.mscale = k.c.mscale; M.mshift = k.c.mshift; M.in_scale = k.c.in_scale; M.in_shift = k.c.in_shift;
    M.stats = k.c.stats; M.stats_R = k.c.stats_R > 0 ? k.c.stats_R : 1; M.d = k.d; M.has_fold = k.has_fold; M.f = k.f;
    g.ustart[i] = units;
    units += g.H + k.d * g.c0;
    if (smems[i] > smem) smem = smems[i];
  }
  g.ustart[m] = units;                                  // the units of one image
  units *= g.N;
  int nb = units / (g.c0 + 5);                          // a piece is worth a block's setup from about five row stages on
  if (nb > rua_cu_count()) nb = rua_cu_count();
  if (nb < 1) nb = 1;
  g.nblocks = nb;
  if (m == 1) hipLaunchKernelGGL((conv_strip32s<NW, IN, EP, ST, ORELU, R, BIAS>), dim3((nb + 7) / 8 * 8), dim3(NW * 64), smem, st, g);
  else hipLaunchKernelGGL((conv_strip32s_g<NW, IN, EP, ST, ORELU, R, BIAS>), dim3((nb + 7) / 8 * 8), dim3(NW * 64), smem, st, g);
  RUA_LAUNCH_CHECK("conv_strip32s");
  return RUA_OK;
}
// the (IN, EP, ST, ORELU) forms the step uses; everything else stays on conv_strip32
struct StripSForm { int in, ep, st, orelu, R, bias; };
static const StripSForm kStripS[] = {
  {1, 0, 1, 0, 6, 1},   // first convs of a ResBlock: BatchNorm + ReLU on load, statistics of the output
  {1, 0, 0, 0, 6, 1},   // ... in evaluation mode
  {0, 2, 2, 0, 5, 0},   // data gradients (no bias): ReLU mask from aux, sums of g and g * aux
  {0, 2, 0, 0, 5, 0},   // ... mask only
  {0, 0, 0, 1, 6, 1},   // 3x3 + ReLU of the heads
  {0, 0, 0, 0, 6, 1},   // plain
  {1, 1, 0, 0, 5, 1},   // BatchNorm on load, accumulating / residual (second convs member by member)
  {0, 1, 0, 0, 5, 0},   // accumulating data gradient
};
static int strip_s_form(const rua_conv_desc* d) {
  const int in = (d->in_scale || d->in_fold) ? 1 : 0;
  if (in && !d->in_relu) return -1;
  const int ep = d->aux_mode == 2 ? 2 : ((d->aux_mode == 1 || d->accumulate) ? 1 : 0);
  const bool has_bias = d->bias || d->bias_more[0] || d->bias_more[1] || d->bias_more[2];
  for (int i = 0; i < (int)(sizeof(kStripS) / sizeof(kStripS[0])); ++i)
    if (kStripS[i].in == in && kStripS[i].ep == ep && kStripS[i].st == d->stats_mode && kStripS[i].orelu == (d->out_relu ? 1 : 0) &&
        (kStripS[i].bias || !has_bias)) return i;
  return -1;
}
static int launch_strip_variant(int variant, const StripK* ks, const int* smems, int m, hipStream_t st) {
  if (variant >= 16) {
    const int form = (variant - 16) >> 1;
    const bool w4 = (variant & 1) != 0;
#define RUA_STRIP_S(F_, IN_, EP_, ST_, OR_, R_, B_) case F_: return w4 ? launch_strip_s<4, IN_, EP_, ST_, OR_, R_, B_>(ks, smems, m, st) : launch_strip_s<8, IN_, EP_, ST_, OR_, R_, B_>(ks, smems, m, st)
    switch (form) {
      RUA_STRIP_S(0, 1, 0, 1, 0, 6, 1);
      RUA_STRIP_S(1, 1, 0, 0, 0, 6, 1);
      RUA_STRIP_S(2, 0, 2, 2, 0, 5, 0);
      RUA_STRIP_S(3, 0, 2, 0, 0, 5, 0);
      RUA_STRIP_S(4, 0, 0, 0, 1, 6, 1);
      RUA_STRIP_S(5, 0, 0, 0, 0, 6, 1);
      RUA_STRIP_S(6, 1, 1, 0, 0, 5, 1);
      default: RUA_STRIP_S(7, 0, 1, 0, 0, 5, 0);
    }
#undef RUA_STRIP_S
  }
  switch (variant) {
    case 0: return launch_strip_members<8, true, 5>(ks, smems, m, st);
    case 1: return launch_strip_members<8, false, 7>(ks, smems, m, st);
    case 2: return launch_strip_members<4, true, 5>(ks, smems, m, st);
    default: return launch_strip_members<4, false, 7>(ks, smems, m, st);
  }
}
int rua_strip_group_pending(void) { return g_strip_cap.n; }
void rua_strip_group_reset(void) { g_strip_cap.n = 0; }
int rua_strip_group_flush(hipStream_t st, int* grids) {
  StripCapture& c = g_strip_cap;
  bool done[RUA_MAX_BRANCH] = {false};
  int rc = RUA_OK;
  for (int i = 0; i < c.n && rc == RUA_OK; ++i) {
    if (done[i]) continue;
    StripK ks[RUA_MAX_BRANCH]; int sm[RUA_MAX_BRANCH], m = 0;
    // one grid takes the largest member's LDS size for every block: only members that keep their own number of blocks per CU
    // under it go together (128-pixel strips at d = 1 run two blocks per CU, at d = 31 one: grouped, d = 1 lost half its occupancy)
    const int per_cu = (160 * 1024) / c.smem[i];
    for (int j = i; j < c.n; ++j)
      if (!done[j] && c.variant[j] == c.variant[i] && (160 * 1024) / c.smem[j] == per_cu) { ks[m] = c.k[j]; sm[m] = c.smem[j]; ++m; done[j] = true; }
    rc = launch_strip_variant(c.variant[i], ks, sm, m, st);
    if (grids) ++*grids;
  }
  c.n = 0;
  return rc;
}

int rua_launch_conv_strip(const ConvK& k, const rua_conv_desc* d, hipStream_t st) {
  StripK q;
  q.c = k;
  const int sw = strip_width(d), nw = sw / 32, dil = d->seg[0].dil;
  const bool has_ep = d->aux_mode != 0 || d->accumulate;
  q.ep = has_ep ? (d->aux_mode != 0 ? (const unsigned char*)d->aux : (const unsigned char*)d->y) : nullptr;
  q.epbytes = (unsigned)((size_t)k.M * 32 * 2);
  q.d = dil;
  q.dbg = g_tune.band_dbg;
  q.strips = d->W / sw;
  q.SWH = sw + 2 * dil;
  q.slot_bytes = (q.SWH * 64 + 1023) / 1024 * 1024;
  RUA_CHECK_ARG(q.slot_bytes <= 3 * nw * 1024, "conv_strip: dilation %d too large for a %d-pixel strip", dil, sw);
  q.nchains = d->N * q.strips * dil;
  const int ny = (d->H + dil - 1) / dil;               // lattice rows of the longest chain
  // one round of blocks where the chains allow it; the members of a grouped launch share that round (tuning key strip_group_share:
  // four members of 256 blocks each ran four blocks per CU back to back, every one with its own prologue, window fill and
  // statistics fold - longer segments amortise them)
  const int share = (g_conv_group && (g_tune.conv_group & 1) && g_tune.strip_group_share) ? g_conv_group->members : 1;
  int spc = rua_cu_count() / (q.nchains * share);
  if (spc < 1) spc = 1;
  if (spc > (ny + 3) / 4) spc = (ny + 3) / 4;          // >= 4 rows per segment (a segment re-reads two window rows)
  if (spc < 1) spc = 1;
  q.seglen = (ny + spc - 1) / spc;
  if (g_tune.strip_seglen > 0) q.seglen = g_tune.strip_seglen;
  q.spc = (ny + q.seglen - 1) / q.seglen;
  q.njobs = q.nchains * q.spc;
  // full-width strips: conv_strip32s (the two halves of a block half a stage apart, fragments kept in registers)
  const int sform = (g_tune.strip_stag && q.strips == 1 && (size_t)k.M * 64 < 0x80000000ull) ? strip_s_form(d) : -1;
  const int R = sform >= 0 ? kStripS[sform].R : (has_ep ? 5 : 7);
  const int smem = sform >= 0 ? R * (sw + 64) * 64 + (has_ep ? 3 * sw * 64 : 0) + RUA_MAX_BRANCH * 7 * 32 * 4
                              : R * q.slot_bytes + (has_ep ? 3 * sw * 64 : 0) + nw * 1024 + 5 * 32 * 4;
  RUA_CHECK_ARG(smem >= nw * (64 * 33 + 64) * 4, "conv_strip: no room for the statistics fold");
  q.has_fold = d->in_fold ? 1 : 0;
  if (d->in_fold) {
    q.f = *d->in_fold;
    RUA_CHECK_ARG(!d->in_scale && !d->in_shift, "conv_strip: in_fold excludes in_scale / in_shift");
    RUA_CHECK_ARG(q.f.stats && q.f.replicas >= 1 && q.f.count > 0 && q.f.gamma && q.f.beta && q.f.scale && q.f.shift, "conv_strip: incomplete in_fold");
    RUA_CHECK_ARG((q.f.moving_mean == nullptr) == (q.f.moving_var == nullptr), "conv_strip: in_fold needs both moving statistics or neither");
    RUA_CHECK_ARG(nw * 2 * 2 * 32 * 8 <= R * q.slot_bytes, "conv_strip: no room for the in_fold scratch");
  } else memset(&q.f, 0, sizeof(q.f));
  RUA_CHECK_ARG(smem <= 160 * 1024, "conv_strip: %d bytes of LDS", smem);
  const int variant = sform >= 0 ? 16 + sform * 2 + (nw == 4 ? 1 : 0) : (nw == 8 ? 0 : 2) + (has_ep ? 0 : 1);
  if (g_conv_group && (g_tune.conv_group & 1)) {        // capture mode: issued by rua_strip_group_flush, grouped with its siblings
    StripCapture& c = g_strip_cap;
    RUA_CHECK_ARG(c.n < RUA_MAX_BRANCH, "conv_strip: group capture overflow");
    c.variant[c.n] = variant; c.smem[c.n] = smem; c.k[c.n] = q; ++c.n;
    return RUA_OK;
  }
  return launch_strip_variant(variant, &q, &smem, 1, st);
}

// conv_band64<BR, R, FULLW>: conv_band32's scheme (conv_band.hip) for the C = Cout = 64 level of the network (ResBlock(64,[1,3,15,31]) at
// half resolution, model2.py:104): the second convolutions of all branches and the Add over them (model2.py:22-31) in ONE launch,
//
//     y = residual + sum_b ( bias_b + W_b (*)_{d_b} relu(scale_b * x_b + shift_b) )          b = 0 .. nb-1 (<= 4), d_b <= 32
//
// every branch normalised on load by its own (folded) BatchNorm - no normalised copy of y1_b in HBM, `out` written once.  As four
// rua_bn_fwd launches + one 4-segment implicit GEMM that stage costs 8 tensor passes of BatchNorm traffic beside the convolution.
//
// What differs from the 32-channel kernel: a row of 128 pixels x 64 channels is the same 16 KB, but the product per row is twice as
// large - 8 waves = 4 pixel tiles x 2 output-channel halves, 12 MFMAs per wave and row (3 taps x 4 k-steps) against 6: more matrix
// work per barrier, which is what these issue-bound kernels lack.  A band is 4 rows (256 blocks at 8 x 128 x 128).  The weights of a
// whole branch (72 KB) do not fit beside the ring: they are staged per KERNEL ROW (24 KB: 3 taps x 4 k-steps x 2 halves, a fragment
// per lane) into two alternating buffers, the row of phase p + 2 arriving by LDS-DMA during stages 1 .. 3 of phase p (buffer p & 1 is
// free once every wave has read its fragments of phase p at stage 0 - the barrier of stage 1 says so) and consumed >= 5 stages later,
// long after the counted waits have retired it.  LDS image: pixel stride 128 B, 16-byte slot = piece ^ ((pixel >> 1) & 7) - the 16
// lanes of a ds_read_b128 group see 16 distinct slots of the 256-byte bank row at every tap shift (conv_dmap's swizzle).
#include "common.h"

struct Band64K {
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
  // MULTI (rua_conv_fwd_group): every member has its own output and epilogue
  unsigned char* ym[RUA_MAX_BRANCH];
  const unsigned char* aux[RUA_MAX_BRANCH];          // ReLU-mask source (aux_mode 2) or null
  const float* mscale[RUA_MAX_BRANCH]; const float* mshift[RUA_MAX_BRANCH];
  double* stats[RUA_MAX_BRANCH]; int stats_mode[RUA_MAX_BRANCH]; int stats_R[RUA_MAX_BRANCH];
  int N, H, W, strips, bands, njobs;
  unsigned xbytes;
  int dbg;                                  // experiments (tuning key band_dbg): 2 no BatchNorm pass, 4 no row DMAs in the loop
};
static_assert(sizeof(Band64K) <= 4096, "kernel arguments are limited to 4 KiB");

// MULTI: the members are INDEPENDENT convolutions (the first convolutions of the branches, or their data gradients: rua_conv_fwd_group) -
// each with its own output, bias, optional ReLU mask from an aux tensor and statistics; the band's accumulators are written out and
// cleared after every member's three phases.  Not MULTI: the members are summed into one output (rua_conv_fwd_sum).
template <int BR, int R, bool FULLW, bool MULTI>
__device__ __forceinline__ void conv_band64_body(const Band64K& q) {
  typedef bf16_t T;
  constexpr int C = 64, NW = 8, NT = NW * 64, SW = 128, HALO = 32, PXB = C * 2;
  constexpr int SPX = SW + 2 * HALO, SLOT = SPX * PXB;                // 192 pixels, 24 KiB
  constexpr int DMA0 = FULLW ? HALO : 0;                              // first slot pixel the row DMAs write
  constexpr int SLOT_INST = (FULLW ? SW : SPX) * PXB / 1024;          // 16 / 24 instructions of 8 pixels each
  constexpr int NPX = SLOT_INST / NW;                                 // 2 / 3: every instruction of every wave is a slot piece
  static_assert(SLOT_INST % NW == 0, "row DMAs divide evenly over the waves");
  constexpr int NWAIT = (R - 2) * NPX;                                // x operations issued after those of row s + 1 (weight DMAs on top: the wait is then stricter, never laxer)
  constexpr int WPIECES = 24, WBUF = WPIECES * 1024;                  // one kernel row: 3 taps x 4 k-steps x 2 output-channel halves
  static_assert(BR >= 4 && (BR - 1) * NW >= WPIECES && R - 1 <= BR && R >= 4, "pipeline depths");
  constexpr unsigned ROW_OOB = 0x80000000u, COL_OOB = 0x7FFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem + R * SLOT;                                // [2][WPIECES][64 lanes][16 B]
  float* tab = reinterpret_cast<float*>(sW + 2 * WBUF);               // [nb][2][64] scale, shift ; [64] bias sum at 4 * 128
  float* tabm = tab + 4 * 128 + 64;                                   // MULTI: [nb][3][64] bias, mask scale, mask shift
  float* sred = tabm + 4 * 192;                                       // MULTI: [8 waves][64] statistics partials of a member

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 31, kh = lane >> 5;
  const int pxt = wv >> 1, coh = wv & 1;                              // this wave's pixel tile (32 pixels) and output-channel half
  const int H = q.H, W = q.W, nb = q.nb;

  const int nwg = q.njobs, bid = blockIdx.x;
  if (bid >= nwg) return;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // consecutive jobs share an XCD's L2
  const int band = job % q.bands, tq = job / q.bands;
  const int x0 = (tq % q.strips) * SW, n_ = tq / q.strips;
  const int h0 = band * BR;

  // ---- per-channel tables (ordinary loads: all consumed before the first LDS-DMA is issued) ------------------------------
  if (tid < 64) {
    float b = 0.f;
    for (int i = 0; i < nb; ++i) if (q.bias[i]) b += q.bias[i][tid];
    tab[4 * 128 + tid] = b;
  }
  if (q.has_fold) {
    rua_fold_members<NT, 64>([&](int m) -> const rua_bn_fold& { return q.f[m]; }, [](int) { return true; }, nb, job == 0, reinterpret_cast<double*>(smem), tid,
                             [&](int m, int c, float scf, float shf) { tab[m * 128 + c] = scf; tab[m * 128 + 64 + c] = shf; });
  } else if (tid < 64) {
    for (int i = 0; i < nb; ++i) {
      tab[i * 128 + tid] = q.in_scale[i] ? q.in_scale[i][tid] : 1.f;
      tab[i * 128 + 64 + tid] = q.in_shift[i] ? q.in_shift[i][tid] : 0.f;
    }
  }
  if constexpr (MULTI) {
    if (tid < 64)
      for (int i = 0; i < nb; ++i) {
        tabm[i * 192 + tid] = q.bias[i] ? q.bias[i][tid] : 0.f;
        tabm[i * 192 + 64 + tid] = (q.aux[i] && q.mscale[i]) ? q.mscale[i][tid] : 1.f;
        tabm[i * 192 + 128 + tid] = (q.aux[i] && q.mshift[i]) ? q.mshift[i][tid] : 0.f;
      }
  }
  __syncthreads();
  const bool bn = q.has_bn != 0 && !(q.dbg & 2);
  const int d0 = q.d[0], d1 = q.d[1], d2 = q.d[2], d3 = q.d[3];
  auto dil_of = [&](int b) { return b == 0 ? d0 : (b == 1 ? d1 : (b == 2 ? d2 : d3)); };

  // ---- DMA addressing: lane l of instruction `inst` (1 KiB = 8 pixels) moves piece psrc of slot pixel DMA0 + inst * 8 + (l >> 3)
  // into slot position (l & 7) = psrc ^ ((pixel >> 1) & 7).  Instructions k of a wave are 64 pixels apart: psrc is the same for all.
  const int pix0 = DMA0 + wv * 8 + (lane >> 3);
  const int psrc = (lane & 7) ^ ((pix0 >> 1) & 7);
  unsigned xrel[NPX]; bool xok[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int x = x0 - HALO + pix0 + k * NW * 8;
    xok[k] = x >= 0 && x < W;
    xrel[k] = xok[k] ? (unsigned)((x * C + psrc * 8) * 2) : COL_OOB;
  }
  const unsigned pdst0 = (unsigned)(DMA0 * PXB + wv * 1024 + lane * 16);     // this lane's piece of instruction 0 inside a slot; k: + k * NW * 1024
  const unsigned smem_a = (unsigned)(size_t)(lds_void_p)smem;
  const unsigned rowbytes = (unsigned)(W * C * 2), imgbase = (unsigned)(n_ * H) * rowbytes;

  struct Phase { int hb; bool valid; __amdgpu_buffer_rsrc_t rx; unsigned ca; };
  auto phase = [&](int ph) {
    Phase p;
    const int b = ph / 3, ty = ph - 3 * b;
    p.valid = ph < 3 * nb;
    p.hb = h0 + (ty - 1) * dil_of(b);
    p.rx = make_rsrc(q.x[p.valid ? b : 0], q.xbytes);
    p.ca = smem_a + (unsigned)((unsigned char*)(tab + (p.valid ? b : 0) * 128 + psrc * 8) - smem);
    return p;
  };
  auto issue_x = [&](const Phase& p, int r, unsigned so) {
    const int h = p.hb + r;
    const bool ok = p.valid && (unsigned)h < (unsigned)H;
    const unsigned base = ok ? imgbase + (unsigned)h * rowbytes : ROW_OOB;
#pragma unroll
    for (int k = 0; k < NPX; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(p.rx, (lds_void_p)(smem + so + DMA0 * PXB + (k * NW + wv) * 1024), 16, base + xrel[k], 0, 0, 0);
  };
  // weight piece idx of kernel row (phase) ph into buffer ph & 1.  Round 5: the buffer holds the kernel row as [tap column 3][output channel 64][input
  // channel 64] - 128-byte rows with the pixel slots' swizzle (piece ^ ((row >> 1) & 7)) on the DMA's source side -, an instruction moves EIGHT WHOLE
  // weight rows (coalesced).  Rounds 3 - 4 staged a fragment per lane (32 rows x 32 bytes per instruction: 32 partly used lines, ~30 ns per instruction
  // in the texture path against ~12.5 ns for a row instruction - measured on conv_band128m, conv_band128.hip).
  const int wrl = (lane >> 3);                                                // row of the instruction's eight, piece slot lane & 7
  auto issue_w = [&](int ph, int idx) {
    const int b = ph / 3, ty = ph - 3 * b;
    if (idx < WPIECES && ph < 3 * nb) {
      const __amdgpu_buffer_rsrc_t rw = make_rsrc(q.w[b], (unsigned)(9 * C * C * 2));
      const int t = ty * 3 + (idx >> 3), row = 8 * (idx & 7) + wrl;
      const unsigned off = (unsigned)(((t * C + row) * C + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_p)(sW + (ph & 1) * WBUF + idx * 1024), 16, off, 0, 0, 0);
    }
  };
  // this wave's fragment of tap column tx, k-step ks: row coh * 32 + pl of the image, piece 2 ks + kh at slot ^ ((row >> 1) & 7)
  const unsigned wfo = (unsigned)((coh * 32 + pl) * 128);
  const int wfs = (pl >> 1) & 7;
  auto wfrag = [&](int buf, int tx, int ks) {
    return *reinterpret_cast<const bf16x8*>(sW + buf * WBUF + tx * 8192 + wfo + (((2 * ks + kh) ^ wfs) * 16));
  };
  // BatchNorm (+ ReLU) of a landed row, in place, on this thread's own DMA pieces (raw LDS accesses: see conv_band.hip)
  auto tr_valid = [&](const Phase& p, int r) { return bn && p.valid && (unsigned)(p.hb + r) < (unsigned)H; };
  auto tr_read = [&](const Phase& p, unsigned so, f32x4& sa, f32x4& sb, f32x4& ha, f32x4& hb, u32x4_t* rw) {
    const unsigned a0 = smem_a + so + pdst0;
    if constexpr (NPX == 3)
      asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:256\n\tds_read_b128 %3, %7 offset:272\n\t"
                   "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:%c9\n\tds_read_b128 %6, %8 offset:%c10\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sa), "=&v"(sb), "=&v"(ha), "=&v"(hb), "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2])
                   : "v"(p.ca), "v"(a0), "n"(NW * 1024), "n"(2 * NW * 1024) : "memory");
    else
      asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:256\n\tds_read_b128 %3, %6 offset:272\n\t"
                   "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:%c8\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sa), "=&v"(sb), "=&v"(ha), "=&v"(hb), "=&v"(rw[0]), "=&v"(rw[1])
                   : "v"(p.ca), "v"(a0), "n"(NW * 1024) : "memory");
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
      if (FULLW || xok[k]) {
        const unsigned la = smem_a + so + pdst0 + (unsigned)(k * NW * 1024);
        asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(rw[k]) : "memory");
      }
  };

  const int o = pxt * 32 + pl;                                        // this lane's output pixel inside the strip
  const size_t pixg = (size_t)((n_ * H + h0) * W + x0 + o);           // ... and in the tensor (row h0 of the band)
  // MULTI: a member's per-channel statistics, summed over the block (the waves' partials are in sred since the member's epilogue)
  auto stats_flush = [&](int b) {
    if (tid < 128 && q.stats_mode[b] != 0) {
      const int ch = tid >> 6, ln = tid & 63, idx = ln & 31, khh = ln >> 5;
      float t = 0.f;
#pragma unroll
      for (int p4 = 0; p4 < 4; ++p4) t += sred[(p4 * 2 + ch) * 64 + ln];
      const int st = idx >> 4, c = ch * 32 + 16 * ((idx >> 3) & 1) + 8 * khh + (idx & 7);
      unsafeAtomicAdd(&q.stats[b][(size_t)(job & (q.stats_R[b] - 1)) * 2 * C + st * C + c], (double)t);
    }
  };
  f32x16 acc[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;

  // ---- prologue ------------------------------------------------------------------------------------------------------------
  if constexpr (FULLW) {                                              // the halo pixels of every slot: zero once, never written again
    constexpr int HB = HALO * PXB;
    for (int i = tid; i < R * 2 * HB / 16; i += NT) {
      const int sl = i / (2 * HB / 16), k = i - sl * (2 * HB / 16);
      unsigned char* p = smem + sl * SLOT + (k < HB / 16 ? k * 16 : (HALO + SW) * PXB + (k - HB / 16) * 16);
      *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  }
  Phase cur = phase(0), nxt = phase(1);
#pragma unroll
  for (int i = 0; i < 3; ++i) { issue_w(0, i * NW + wv); issue_w(1, i * NW + wv); }      // kernel rows of phases 0 and 1
#pragma unroll
  for (int s = 0; s <= R - 2; ++s) issue_x(cur, s, (unsigned)(s * SLOT));               // R - 1 <= BR: all in phase 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tr_valid(cur, 0)) {
    f32x4 sa, sb, ha, hb; u32x4_t rw[3];
    tr_read(cur, 0u, sa, sb, ha, hb, rw); tr_math(sa, sb, ha, hb, rw); tr_write(0u, rw);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  unsigned so_cur = 0, so_nxt = SLOT, so_iss = (R - 1) * SLOT;
  for (int ph = 0; ph < 3 * nb; ++ph) {
    const int d = dil_of(ph / 3);
    // b-operand fragment addresses: output pixel o, tap column tx reads slot pixel HALO + o + (tx - 1) d; piece 2 ks + kh
    unsigned boff[3];
    int bsw[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int j = HALO + o + (tx - 1) * d;
      boff[tx] = (unsigned)(j * PXB);
      bsw[tx] = (j >> 1) & 7;
    }
    bf16x8 wf[3][4];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      __builtin_amdgcn_s_barrier();                      // everyone's pieces of row s are normalised; every wave is done with row s - 1
      if constexpr (MULTI) {
        if (r == 0 && ph > 0 && ph % 3 == 0) stats_flush(ph / 3 - 1);      // the member that just finished: its partials are visible now
      }
      if (r == 0) {
        // this wave's fragments of the kernel row (its output-channel half): landed long ago (issued two phases back), visible since the barrier
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) wf[tx][ks] = wfrag(ph & 1, tx, ks);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (r <= 3) {
        issue_w(ph + 2, (r - 1) * NW + wv);              // the kernel row two phases ahead into the buffer every wave has just finished reading
      }
      if (!(q.dbg & 4)) {
        if (r + R - 1 < BR) issue_x(cur, r + R - 1, so_iss);
        else issue_x(nxt, r + R - 1 - BR, so_iss);
      }

      const unsigned char* row = smem + so_cur;
      const bool tv = r + 1 < BR ? tr_valid(cur, r + 1) : tr_valid(nxt, 0);
      f32x4 sa, sb, ha, hb; u32x4_t rw[3];
      // first half of the row's product (k-steps 0, 1), the in-place BatchNorm of row s + 1 beside it, then the second half
      bf16x8 fx[3][2];
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx[tx][ks] = *reinterpret_cast<const bf16x8*>(row + boff[tx] + ((((ks * 2 + kh) ^ bsw[tx])) * 16));
      if (MULTI && r < 2 && ph > 0 && ph % 3 == 0)       // the 2 BR stores of the member epilogue are younger than row s + 1's DMAs for two stages
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT + 2 * BR) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");
      if (tv) tr_read(r + 1 < BR ? cur : nxt, so_nxt, sa, sb, ha, hb, rw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tx][ks], fx[tx][ks], acc[r], 0, 0, 0);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx[tx][ks] = *reinterpret_cast<const bf16x8*>(row + boff[tx] + (((((ks + 2) * 2 + kh) ^ bsw[tx])) * 16));
      if (tv) { tr_math(sa, sb, ha, hb, rw); tr_write(so_nxt, rw); }
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tx][ks + 2], fx[tx][ks], acc[r], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      so_iss = so_cur; so_cur = so_nxt; so_nxt = so_nxt + SLOT == (unsigned)(R * SLOT) ? 0u : so_nxt + SLOT;
    }
    if constexpr (MULTI) {
      if (ph % 3 == 2) {
        // ---- member epilogue: bias, ReLU mask from the aux tensor, statistics, one write of the member's band; accumulators cleared
        const int b = ph / 3;
        const unsigned char* auxp = q.aux[b];
        unsigned char* yp = q.ym[b];
        const int smode = q.stats_mode[b];
        const float* tb = tabm + b * 192;
        float s1[2][8], s2[2][8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[g][j] = 0.f; s2[g][j] = 0.f; }
        uint4 av[2][2];
        auto load_aux = [&](int r, uint4* dst) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            dst[g] = auxp ? ldg16(auxp + ((pixg + (size_t)r * W) * C + coh * 32 + 16 * g + 8 * kh) * 2) : make_uint4(0, 0, 0, 0);
        };
        load_aux(0, av[0]);
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          if (r + 1 < BR) load_aux(r + 1, av[(r + 1) & 1]);
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
          unsigned char* yrow = yp + ((pixg + (size_t)r * W) * C) * 2;
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int co = coh * 32 + 16 * g + 8 * kh;
            float a8[8];
            ET<T>::unpack(av[r & 1][g], a8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g][j] += tb[co + j];
            if (auxp) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[g][j] = (fmaf(tb[64 + co + j], a8[j], tb[128 + co + j]) > 0.f) ? v[g][j] : 0.f;
            }
            if (smode == 1) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], v[g][j], s2[g][j]); }
            } else if (smode == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], a8[j], s2[g][j]); }
            }
            stg16(yrow + co * 2, ET<T>::pack(v[g]));
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[r][k] = 0.f;
        }
        if (smode != 0) {
          // 32 partial sums per lane over its pixel's rows -> per-channel sums over the wave's 32 pixels by a transposing butterfly (31
          // exchanges: after the step with distance k a lane keeps the half of the values its bit k selects), lane l ends with value l
          float vals[32];
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) { vals[g * 8 + j] = s1[g][j]; vals[16 + g * 8 + j] = s2[g][j]; }
#pragma unroll
          for (int k = 16; k >= 1; k >>= 1) {
            const bool hi = (pl & k) != 0;
#pragma unroll
            for (int i = 0; i < k; ++i) {
              const float send = hi ? vals[i] : vals[i + k];
              const float keep = hi ? vals[i + k] : vals[i];
              vals[i] = keep + __shfl_xor(send, k, 64);
            }
          }
          sred[wv * 64 + lane] = vals[0];
        }
      }
    }
    cur = nxt;
    nxt = phase(ph + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");       // the over-issued DMAs of the last stages

  if constexpr (MULTI) {                                               // the last member's statistics
    __syncthreads();
    stats_flush(nb - 1);
    return;
  }
  // ---- epilogue: bias sum + residual, one write of the band (this wave: 32 pixels x its 32 output channels) -------------------
  uint4 rv[2][2];
  auto load_res = [&](int r, uint4* dst) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
      dst[g] = q.res ? ldg16(q.res + ((pixg + (size_t)r * W) * C + coh * 32 + 16 * g + 8 * kh) * 2) : make_uint4(0, 0, 0, 0);
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
    unsigned char* yrow = q.y + ((pixg + (size_t)r * W) * C) * 2;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = coh * 32 + 16 * g + 8 * kh;
      float a8[8];
      ET<T>::unpack(rv[r & 1][g], a8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[g][j] += tab[4 * 128 + co + j] + a8[j];
      stg16(yrow + co * 2, ET<T>::pack(v[g]));
    }
  }
}

template <int BR, int R, bool FULLW> __global__ __launch_bounds__(512) void conv_band64(const Band64K q) { conv_band64_body<BR, R, FULLW, false>(q); }
template <int BR, int R, bool FULLW> __global__ __launch_bounds__(512) void conv_band64m(const Band64K q) { conv_band64_body<BR, R, FULLW, true>(q); }

// ---- host side (called by rua_conv_fwd_sum, conv_band.hip) ---------------------------------------------------------------------
bool rua_band64_ok(const rua_conv_desc* d, int n) {
  if (!g_tune.conv_band64 || n < 1 || n > RUA_MAX_BRANCH) return false;
  const rua_conv_desc& a = d[0];
  if (a.dtype != RUA_BF16 || a.W % 128 != 0 || a.H % 4 != 0 || (long long)a.N * a.H * a.W < 16384) return false;
  if (a.aux_mode != 0 && a.aux_mode != 1) return false;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    const rua_conv_seg& g = m.seg[0];
    if (m.nseg != 1 || m.dtype != RUA_BF16 || g.taps != 9 || g.up_shift != 0 || g.C != 64 || m.Cout != 64 || m.stride != 1 ||
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

int rua_launch_band64(const rua_conv_desc* d, int n, hipStream_t st) {
  Band64K q;
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
  RUA_CHECK_ARG((size_t)a.N * a.H * a.W * 64 * 2 < 0x7FFFFF00ull, "rua_conv_fwd_sum: tensor of 2 GiB or more");
  q.xbytes = (unsigned)((size_t)a.N * a.H * a.W * 64 * 2);
  constexpr int BR = 4, R = 4;
  q.strips = a.W / 128;
  q.bands = a.H / BR;
  q.njobs = a.N * q.strips * q.bands;
  const int smem = R * 192 * 128 + 2 * 24 * 1024 + (4 * 128 + 64 + 4 * 192 + 8 * 64) * 4;
  static_assert(4 * 192 * 128 + 2 * 24 * 1024 + (4 * 128 + 64 + 4 * 192 + 8 * 64) * 4 <= 160 * 1024, "LDS budget");
  static RuaPerDevFlag attr[2];
  if (q.strips == 1) {
    if (!attr[0].get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band64<BR, R, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[0].get() = true; }
    hipLaunchKernelGGL((conv_band64<BR, R, true>), dim3(q.njobs), dim3(512), smem, st, q);
  } else {
    if (!attr[1].get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band64<BR, R, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[1].get() = true; }
    hipLaunchKernelGGL((conv_band64<BR, R, false>), dim3(q.njobs), dim3(512), smem, st, q);
  }
  RUA_LAUNCH_CHECK("conv_band64");
  return RUA_OK;
}

// ---- MULTI: independent members (rua_conv_fwd_group) ---------------------------------------------------------------------------
bool rua_band64m_ok(const rua_conv_desc* d, int n) {
  if (!g_tune.conv_band64m || n < 2 || n > RUA_MAX_BRANCH) return false;
  const rua_conv_desc& a = d[0];
  if (a.dtype != RUA_BF16 || a.W % 128 != 0 || a.H % 4 != 0 || (long long)a.N * a.H * a.W < 16384) return false;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    const rua_conv_seg& g = m.seg[0];
    if (m.nseg != 1 || m.dtype != RUA_BF16 || g.taps != 9 || g.up_shift != 0 || g.C != 64 || m.Cout != 64 || m.stride != 1 ||
        m.out_stride != 1 || m.OH != m.H || m.OW != m.W || g.Hs != m.H || g.Ws != m.W || g.dil < 1 || g.dil > 32) return false;
    if (m.N != a.N || m.H != a.H || m.W != a.W || !m.y) return false;
    for (int j = 0; j < i; ++j) if (d[j].y == m.y) return false;               // independent outputs
    if (m.accumulate || m.out_relu || m.bias_more[0] || m.bias_more[1] || m.bias_more[2]) return false;
    if (!(m.aux_mode == 0 || (m.aux_mode == 2 && m.aux))) return false;
    if (m.stats_mode != 0 && (!m.stats || m.stats_replicas < 1 || (m.stats_replicas & (m.stats_replicas - 1)))) return false;
    if (m.stats_mode == 2 && m.aux_mode != 2) return false;
    if (m.stats_mode < 0 || m.stats_mode > 2) return false;
    if ((m.in_fold != nullptr) != (a.in_fold != nullptr) || (m.in_scale != nullptr) != (a.in_scale != nullptr) || m.in_relu != a.in_relu) return false;
    if (m.in_fold && (m.in_scale || m.in_shift)) return false;
  }
  return true;
}

int rua_launch_band64m(const rua_conv_desc* d, int n, hipStream_t st) {
  Band64K q;
  memset(&q, 0, sizeof(q));
  q.nb = n;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    q.x[i] = (const unsigned char*)m.seg[0].x; q.w[i] = (const unsigned char*)m.seg[0].w; q.bias[i] = m.bias;
    q.in_scale[i] = m.in_scale; q.in_shift[i] = m.in_shift; q.d[i] = m.seg[0].dil;
    q.ym[i] = (unsigned char*)m.y;
    q.aux[i] = m.aux_mode == 2 ? (const unsigned char*)m.aux : nullptr;
    q.mscale[i] = m.mscale; q.mshift[i] = m.mshift;
    q.stats[i] = m.stats; q.stats_mode[i] = m.stats ? m.stats_mode : 0; q.stats_R[i] = m.stats_replicas > 0 ? m.stats_replicas : 1;
    if (m.in_fold) {
      q.f[i] = *m.in_fold;
      const rua_bn_fold& f = q.f[i];
      RUA_CHECK_ARG(f.stats && f.replicas >= 1 && f.count > 0 && f.gamma && f.beta && f.scale && f.shift, "rua_conv_fwd_group: incomplete in_fold");
      RUA_CHECK_ARG((f.moving_mean == nullptr) == (f.moving_var == nullptr), "rua_conv_fwd_group: in_fold needs both moving statistics or neither");
    }
  }
  for (int i = n; i < RUA_MAX_BRANCH; ++i) { q.d[i] = 1; q.stats_R[i] = 1; }
  const rua_conv_desc& a = d[0];
  q.dbg = g_tune.band_dbg;
  q.has_fold = a.in_fold ? 1 : 0;
  q.has_bn = (a.in_fold || a.in_scale) ? 1 : 0;
  q.in_relu = a.in_relu;
  q.N = a.N; q.H = a.H; q.W = a.W;
  RUA_CHECK_ARG((size_t)a.N * a.H * a.W * 64 * 2 < 0x7FFFFF00ull, "rua_conv_fwd_group: tensor of 2 GiB or more");
  q.xbytes = (unsigned)((size_t)a.N * a.H * a.W * 64 * 2);
  constexpr int BR = 4, R = 4;
  q.strips = a.W / 128;
  q.bands = a.H / BR;
  q.njobs = a.N * q.strips * q.bands;
  const int smem = R * 192 * 128 + 2 * 24 * 1024 + (4 * 128 + 64 + 4 * 192 + 8 * 64) * 4;
  static RuaPerDevFlag attr[2];
  if (q.strips == 1) {
    if (!attr[0].get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band64m<BR, R, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[0].get() = true; }
    hipLaunchKernelGGL((conv_band64m<BR, R, true>), dim3(q.njobs), dim3(512), smem, st, q);
  } else {
    if (!attr[1].get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band64m<BR, R, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[1].get() = true; }
    hipLaunchKernelGGL((conv_band64m<BR, R, false>), dim3(q.njobs), dim3(512), smem, st, q);
  }
  RUA_LAUNCH_CHECK("conv_band64m");
  return RUA_OK;
}

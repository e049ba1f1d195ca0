// conv_band128m<CP, W>: the row-streaming scheme of conv_band64m (conv_band64.hip) for the C = Cout = 128 level of the network
// (ResBlock(128,[1,3,15]) at 64 x 64, model2.py:105-106 and its decoder mirror :128-131): the INDEPENDENT 3x3 convolutions of a group
// (rua_conv_fwd_group: the branches' first convolutions, or their data gradients) as ONE launch.  Rounds 1 - 4 ran these on conv_dmap, an
// implicit GEMM that re-stages every input pixel nine times (once per tap) and whose 128 x 128 x 64 stages move 32 KB through the LDS-DMA
// path per 2.1 MFLOP - the path accepts 1 KiB per ~10.6 ns and CU, so a stage was DMA-bound at 1.6x its MFMA time before any latency.
//
// Geometry (one 512-thread block per CU, 8 x 16 x 2 = 256 jobs at 8 x 64 x 64 x 128):
//   job   = (image, band of BR = 4 output rows, 64-channel slice of the OUTPUT channels)
//   stage = 128 pixels = RPS = 2 whole rows of 64 pixels x all KC = 128 input channels = 32 KB, the pixels of a kernel ROW:
//           a landed row serves the three taps of its kernel row by shifted fragment reads (a third of the input bytes of the
//           implicit GEMM), the weights of one kernel row x the block's 64 output channels (3 taps x 64 x 128 = 48 KB) serve the
//           SPP = 2 stages of a phase.  Per stage and wave: 24 MFMAs 32x32x16 (conv_band64m: 12) on 56 KB of DMA per 6.3 MFLOP =
//           8.9 KB / MFLOP (conv_dmap: 15.3) - just under the 9.6 KB / MFLOP at which the DMA path and the matrix pipe take equal time.
//   waves = 4 pixel tiles of 32 (row of the stage x half row) x 2 output-channel tiles of 32; transposed product D^T[co][px] = W . X^T
//           with the 24 weight fragments of the wave's kernel row IN REGISTERS for the whole phase (96 VGPRs; one LDS read per MFMA).
//   LDS   = ring of R = 3 stage slots (32 KB + one zero pixel each) + ONE 48 KB weight buffer + tables = 154 KB.  Pixel stride 256 B = one
//           bank row: the 16-byte pieces of a pixel are XOR-swizzled with the pixel index (piece ^ (pixel & 15)) on the DMA's SOURCE
//           side, so the 16 lanes of a ds_read_b128 group (16 different pixels, same piece) hit 16 different slots at every tap shift.
//           No halo: a tap column that falls outside the row reads the slot's zero pixel (the fragment address is a per-phase constant).
//   weights: two buffers do not fit.  The fragments of phase p are read into registers behind the phase's first barrier, a second
//           barrier frees the buffer, the DMAs of phase p + 1 are issued right there and have both stages of phase p to land.
//   members: own input, weights, dilation, bias, optional normalise-on-load coefficients (in_scale / in_shift), optional ReLU mask from an aux
//           tensor, statistics (sum v, sum v^2 or sum g, sum g * aux), exactly as conv_band64m; the accumulators leave after a member's three phases.
#include "common.h"

struct Band128K {
  const unsigned char* x[RUA_MAX_BRANCH];
  const unsigned char* w[RUA_MAX_BRANCH];
  const float* bias[RUA_MAX_BRANCH];
  const float* in_scale[RUA_MAX_BRANCH];
  const float* in_shift[RUA_MAX_BRANCH];
  rua_bn_fold f[RUA_MAX_BRANCH];
  unsigned char* ym[RUA_MAX_BRANCH];
  const unsigned char* aux[RUA_MAX_BRANCH];          // ReLU-mask source (aux_mode 2) or null
  const float* mscale[RUA_MAX_BRANCH]; const float* mshift[RUA_MAX_BRANCH];
  double* stats[RUA_MAX_BRANCH]; int stats_mode[RUA_MAX_BRANCH]; int stats_R[RUA_MAX_BRANCH];
  int d[RUA_MAX_BRANCH];
  int has_fold, has_bn, in_relu, nb;
  int N, H, bands, njobs;
  int sum;                                  // 1: the members are the segments of ONE convolution (rua_conv_fwd with several 3x3 segments: the summed second convolutions of a ResBlock) - the accumulators run over all members, one epilogue (bias sum, residual from aux) after the last
  unsigned xbytes;
  int dbg;                                  // experiments (tuning key band_dbg): 2 no BatchNorm pass
  unsigned long long* stamps;               // RUA_B128_STAMPS builds: [njobs][8] section cycle sums (tuning key dbg_ptr), else null
};
static_assert(sizeof(Band128K) <= 4096, "kernel arguments are limited to 4 KiB");
// ablation builds (-DRUA_B128_ABLATE=<bits>, tools/band128_phases.py): 4 no row DMAs in the loop, 8 no MFMAs, 16 no weight DMAs in the loop
#ifndef RUA_B128_ABLATE
#define RUA_B128_ABLATE 0
#endif
// -DRUA_B128_STAMPS: wave 0 of every block sums s_memrealtime (100 MHz) intervals per section of the kernel into stamps[block][8] (tuning key dbg_ptr;
// tools/band128_phases.py).  Diagnostic build only: no stamp executes in the shipped kernel.
#ifdef RUA_B128_STAMPS
#define B128_T(i) do { const unsigned long long t__ = __builtin_amdgcn_s_memrealtime(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
// timeline of every wave through the two stages of phase 4: [job][wave][stage][event 0 barrier left, 1 stage set up, 2 MFMAs done, 3 rows waited for][realtime, shader clock]
#define B128_E(ev) do { if (ph == 4 && lane == 0 && q.stamps) { unsigned long long* e__ = q.stamps + (size_t)q.njobs * 8 + ((((size_t)job * 8 + wv) * 2 + sp) * 4 + (ev)) * 2; \
    e__[0] = __builtin_amdgcn_s_memrealtime(); e__[1] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define B128_T(i) do { } while (0)
#define B128_E(ev) do { } while (0)
#endif

template <int V> struct B128IC { static constexpr int value = V; };

// CP = C = Cout, W = row width (a stage is SPX / W whole rows), KC = input channels staged per phase (CP, or 128 of 256: a kernel row then takes two phases),
// COS = output channels per block: 64 = two 32-channel tiles over the waves (128: one pixel tile per wave; 64: two), or 32 = ONE tile with the k-steps of a
// phase split over the two waves that share a pixel tile (C = 256 on 32-pixel rows: 8 images x 4 bands x 8 slices = 256 jobs; the halves are added through LDS
// in the member epilogue)
template <int CP, int W, int KC = CP, int COS = 64>
__device__ __forceinline__ void conv_band128_body(const Band128K& q) {
  typedef bf16_t T;
  constexpr int NW = 8, NT = NW * 64, R = 3;
  constexpr int PXB = KC * 2, PPP = PXB / 16;                                 // bytes / 16-byte pieces per staged pixel (256 / 16 or 128 / 8)
  constexpr int SLOTB = 32768, SPX = SLOTB / PXB, RPS = SPX / W, SPP = 2, BR = SPP * RPS;     // a stage = 32 KB = 128 or 256 pixels = RPS rows; a band = two stages
  constexpr int TPW = SPX / 128;                                              // pixel tiles (of 32) per wave: 4 pixel groups x 2 (output-channel tiles or k halves) = 8 waves
  constexpr int ZOFF = SLOTB, SLOT = SLOTB + 256;                             // the stage tile + one zero pixel
  constexpr int NINST = SLOTB / 1024, NPX = NINST / NW, PPI = 64 / PPP;       // DMA instructions per stage / per wave, pixels per instruction
  constexpr int NCH = CP / KC, PPM = 3 * NCH;                                 // input-channel chunks; phases (kernel row x chunk) per member
  constexpr int KS = KC / 16, KQ = 64 / COS, KSW = KS / KQ;                   // k-steps per tap and phase; waves sharing a pixel tile (1: they differ in the output-channel tile); k-steps per wave
  constexpr int TAPB = COS * PXB, WBUF = 3 * TAPB, WPIECES = WBUF / 1024, WPW = WPIECES / NW, RPI = 1024 / PXB;   // weight image [tap column][COS rows][KC]; rows per DMA instruction
  constexpr int NCS = CP / COS;
  static_assert((CP == 256 || CP == 128 || CP == 64) && (COS == 64 || COS == 32) && CP % KC == 0 && SPX % W == 0 && RPS >= 1 && W % (NW * PPI) == 0 && W % 32 == 0, "geometry");
  static_assert(KQ == 1 || TPW == 1, "the k split is for one pixel tile per wave");
  static_assert(NINST % NW == 0 && WPIECES % NW == 0 && SLOT % 256 == 0 && (R * SLOT) % 256 == 0, "geometry");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem + R * SLOT;                                        // the weight image of one phase
  float* tab = reinterpret_cast<float*>(sW + WBUF);                           // [nb][2][CP] scale, shift
  float* tabm = tab + RUA_MAX_BRANCH * 2 * CP;                                // [nb][3][64] bias, mask scale, mask shift of the block's output channels
  float* sred = tabm + RUA_MAX_BRANCH * 192;                                  // [8 waves][64] statistics partials of a member
  float* xbuf = sred + NW * 64;                                               // KQ == 2: [4 pixel tiles][16 registers][64 lanes] - the k halves' accumulators meet here

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 31, kh = lane >> 5;
  const int pg = wv >> 1, coh = KQ == 1 ? (wv & 1) : 0, kq = KQ == 1 ? 0 : (wv & 1);   // this wave's pixel group (TPW tiles of the stage), output-channel tile, k half
  const int H = q.H, nb = q.nb;
  auto swz = [](int i) { return PPP == 16 ? (i & 15) : ((i >> 1) & 7); };     // piece slot = piece ^ swz(row or pixel index): 16 pieces per 256-byte bank row, or 8 per half of it

  const int nwg = q.njobs, bid = blockIdx.x;
  if (bid >= nwg) return;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // consecutive jobs (the slices of one band first) share an XCD's L2
  const int cs = job % NCS, tq = job / NCS;
  const int band = tq % q.bands, n_ = tq / q.bands;
  const int h0 = band * BR, co0 = cs * COS;

#ifdef RUA_B128_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- per-channel tables (ordinary loads: all consumed before the first LDS-DMA is issued) ------------------------------
  if (tid < COS)
    for (int i = 0; i < nb; ++i) {
      tabm[i * 192 + tid] = (q.bias[i] ? q.bias[i][co0 + tid] : 0.f) + ((q.sum && i > 0) ? tabm[(i - 1) * 192 + tid] : 0.f);      // (sum: the last member's row holds the sum of the biases)
      tabm[i * 192 + 64 + tid] = (q.aux[i] && q.mscale[i]) ? q.mscale[i][co0 + tid] : 1.f;
      tabm[i * 192 + 128 + tid] = (q.aux[i] && q.mshift[i]) ? q.mshift[i][co0 + tid] : 0.f;
    }
  if constexpr (CP == 64) {
    if (q.has_fold) {                                                         // every member derives (and job 0 publishes) its BatchNorm coefficients from the replicated statistics
      rua_fold_members<NT, 64>([&](int m) -> const rua_bn_fold& { return q.f[m]; }, [](int) { return true; }, nb, job == 0, reinterpret_cast<double*>(smem), tid,
                               [&](int m, int c, float scf, float shf) { tab[m * 2 * CP + c] = scf; tab[m * 2 * CP + CP + c] = shf; });
    }
  }
  if (q.has_bn && !q.has_fold)
    for (int i = tid; i < nb * CP; i += NT) {
      const int m = i / CP, c = i - m * CP;
      tab[m * 2 * CP + c] = q.in_scale[m] ? q.in_scale[m][c] : 1.f;
      tab[m * 2 * CP + CP + c] = q.in_shift[m] ? q.in_shift[m][c] : 0.f;
    }
  if (tid < R * 16) *reinterpret_cast<uint4*>(smem + (tid >> 4) * SLOT + ZOFF + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);   // the zero pixels: never written again
  __syncthreads();
  const bool bn = q.has_bn != 0 && !(q.dbg & 2);
  const int d0 = q.d[0], d1 = q.d[1], d2 = q.d[2], d3 = q.d[3];
  auto dil_of = [&](int b) { return b == 0 ? d0 : (b == 1 ? d1 : (b == 2 ? d2 : d3)); };

  // ---- DMA addressing: lane l of instruction i (1 KiB = PPI pixels x PPP pieces) moves piece psrc of stage pixel i * PPI + l / PPP into slot
  // position l % PPP = psrc ^ swz(pixel).  The instructions of a wave are NW * PPI pixels apart (a multiple of 16): psrc is the same for all of them.
  const int qd0 = wv * PPI + lane / PPP;                                      // pixel of instruction 0 (< NW * PPI <= W); instruction k: + k * NW * PPI
  const int psrc = (lane % PPP) ^ swz(qd0);
  unsigned xrel[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) xrel[k] = (unsigned)((((qd0 + k * NW * PPI) % W) * CP + psrc * 8) * 2);
  auto row_of = [&](int k) { return (k * NW * PPI) / W; };                    // wave-uniform: the row validity below is a scalar select, never a branch around a DMA
  const unsigned pdst0 = (unsigned)(wv * 1024 + lane * 16);                   // this lane's piece of instruction 0 inside a slot; k: + k * NW * 1024
  const unsigned smem_a = (unsigned)(size_t)(lds_void_p)smem;
  const unsigned rowbytes = (unsigned)(W * CP * 2), imgbase = (unsigned)(n_ * H) * rowbytes;

  const int nph = nb * PPM;
  struct Phase { int hb; bool valid; __amdgpu_buffer_rsrc_t rx; unsigned ca, choff; };
  auto phase = [&](int ph) {
    Phase p;
    const int b = ph / PPM, pm = ph - b * PPM, ty = pm / NCH, ch = pm - ty * NCH;
    p.valid = ph < nph;
    const int bb = p.valid ? b : 0;
    p.hb = h0 + (ty - 1) * dil_of(bb);
    p.rx = make_rsrc(q.x[bb], q.xbytes);
    p.choff = (unsigned)(ch * KC * 2);
    p.ca = smem_a + (unsigned)((unsigned char*)(tab + bb * 2 * CP + ch * KC + psrc * 8) - smem);
    return p;
  };
  // one DMA instruction at a time (the loop deals them between its MFMAs: ten issued in one burst behind the barrier stood 0.5 - 1.6 us in the
  // texture path's queue with the wave's MFMAs waiting behind them in program order)
  auto issue_x1 = [&](const Phase& p, int sp, int k, unsigned so) {
    const int h = p.hb + sp * RPS + row_of(k);
    const bool ok = p.valid && (unsigned)h < (unsigned)H;
    const unsigned base = ok ? imgbase + (unsigned)h * rowbytes + p.choff : OOB;        // scalar select; OOB + xrel is still out of range: the lanes write zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(p.rx, (lds_void_p)(smem + so + (k * NW + wv) * 1024), 16, base + xrel[k], 0, 0, 0);
  };
  auto issue_x = [&](const Phase& p, int sp, unsigned so) {
#pragma unroll
    for (int k = 0; k < NPX; ++k) issue_x1(p, sp, k, so);
  };
  // The weights of a phase in LDS: [tap column 3][output channel 64][input channel CP] = the image of the pixel tiles (PXB-byte rows, 16-byte pieces
  // XOR-swizzled with the row index on the DMA's source side), so a DMA instruction moves RPI whole weight rows (fully coalesced).  The first
  // build staged them fragment by fragment - a lane per (output channel, 16 bytes of a 32-byte segment): 32 partly used lines per instruction, and the
  // stamps showed ~30 ns per weight instruction against ~12.5 ns per row instruction in the texture path.
  constexpr int IP16 = 16 / RPI;                                              // instructions per 16 weight rows (the swizzle's period): idx % IP16 == wv % IP16 for every instruction of the wave
  const int wrow0 = RPI * (wv % IP16) + lane / PPP;                           // the lane's weight row modulo 16
  const unsigned wsrc = (unsigned)((wrow0 * CP + (((lane % PPP) ^ swz(wrow0)) * 8)) * 2);
  struct WPhase { __amdgpu_buffer_rsrc_t rw; unsigned base; };
  auto wphase = [&](int ph) {
    WPhase w;
    const bool ok = ph < nph;
    const int pp = ok ? ph : 0;
    const int b = pp / PPM, pm = pp - b * PPM, ty = pm / NCH, ch = pm - ty * NCH;
    w.rw = make_rsrc(q.w[b], ok ? (unsigned)(9 * CP * CP * 2) : 0u);           // past the last phase: every lane out of range (zeros into a dead buffer; the counted waits stay uniform)
    w.base = (unsigned)((((ty * 3) * CP + co0) * CP + ch * KC) * 2) + wsrc;
    return w;
  };
  auto issue_w1 = [&](const WPhase& w, int i) {
    const int idx = i * NW + wv;                                                // tap column idx / (WPIECES / 3), weight rows RPI * (idx % (WPIECES / 3)) ..
    const int tx = idx / (WPIECES / 3), j = idx % (WPIECES / 3);
    const unsigned off = w.base + (unsigned)(((tx * CP + 16 * (j / IP16)) * CP) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w.rw, (lds_void_p)(sW + idx * 1024), 16, off, 0, 0, 0);
  };
  auto issue_w = [&](int ph) {
    const WPhase w = wphase(ph);
#pragma unroll
    for (int i = 0; i < WPW; ++i) issue_w1(w, i);
  };
  // BatchNorm (+ ReLU) of a landed stage, in place, on this thread's own DMA pieces (raw LDS accesses: conv_band.hip says why)
  auto tr_stage = [&](const Phase& p, int sp, unsigned so) {
    f32x4 sa, sb, ha, hb; u32x4_t rw[NPX];
    const unsigned a0 = smem_a + so + pdst0;
    static_assert(NPX == 4, "four pieces per wave and stage");
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:%c10\n\tds_read_b128 %3, %8 offset:%c11\n\t"
                 "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:%c12\n\tds_read_b128 %6, %9 offset:%c13\n\tds_read_b128 %7, %9 offset:%c14\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(sa), "=&v"(sb), "=&v"(ha), "=&v"(hb), "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2]), "=&v"(rw[3])
                 : "v"(p.ca), "v"(a0), "n"(CP * 4), "n"(CP * 4 + 16), "n"(NW * 1024), "n"(2 * NW * 1024), "n"(3 * NW * 1024) : "memory");
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
      const int h = p.hb + sp * RPS + row_of(k);
      if (p.valid && (unsigned)h < (unsigned)H) {            // a row of zero padding stays zero
        const u32x4_t v = {pk.x, pk.y, pk.z, pk.w};
        const unsigned la = a0 + (unsigned)(k * NW * 1024);
        asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(v) : "memory");
      }
    }
  };

  // this lane's output pixels: tile t of the wave = stage pixel (pg * TPW + t) * 32 + pl = row jr (the same for both tiles: 64 | W), column xo + 32 t
  const int qo = pg * TPW * 32 + pl, jr = qo / W, xo = qo - jr * W;
  // a member's per-channel statistics, summed over the block (the waves' partials are in sred since the member's epilogue)
  auto stats_flush = [&](int b) {
    if (tid < COS * 2 && q.stats_mode[b] != 0) {
      const int ch = tid >> 6, ln = tid & 63, idx = ln & 31, khh = ln >> 5;
      float t = 0.f;
      if constexpr (KQ == 1) {
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) t += sred[(p4 * 2 + ch) * 64 + ln];
      } else {                                               // one output-channel tile: all eight waves hold partials of it
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) t += sred[w8 * 64 + ln];
      }
      const int st = idx >> 4, c = co0 + ch * 32 + 16 * ((idx >> 3) & 1) + 8 * khh + (idx & 7);
      unsafeAtomicAdd(&q.stats[b][(size_t)(job & (q.stats_R[b] - 1)) * 2 * CP + st * CP + c], (double)t);
    }
  };
  f32x16 acc[SPP][TPW];
#pragma unroll
  for (int r = 0; r < SPP; ++r)
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.f;

  // ---- prologue: the weights of phase 0 (-> registers), stages 0 and 1 -------------------------------------------------------------
  Phase cur = phase(0), nxt = phase(1);
  issue_w(0);
  issue_x(cur, 0, 0u);
  issue_x(cur, 1, (unsigned)SLOT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bn) tr_stage(cur, 0, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                              // everyone's pieces of the weights have landed
  // Fragment addresses with ONE vector instruction per read: slot = (2 ks + kh) ^ sw = ((ks ^ (sw >> 1)) << 1) | ((kh ^ sw) & 1), so with
  // e = base + row * PXB + ((sw >> 1) << 5) + (((kh ^ sw) & 1) << 4) the address of k-step ks is e ^ (ks << 5): row * PXB and the bases (multiples of
  // 256 from LDS address 0: no static LDS in this kernel) leave bits 5 .. to the swizzle.  The same for pixel rows and weight rows.
  auto frag = [&](unsigned e, int ks) { return *reinterpret_cast<const bf16x8*>(smem + (e ^ (unsigned)(ks << 5))); };
  auto eaddr = [&](unsigned base, int row) {
    const int sw = swz(row);
    return base + (unsigned)(row * PXB + ((sw >> 1) << 5) + (((kh ^ sw) & 1) << 4));
  };
  const unsigned ew0 = eaddr((unsigned)(R * SLOT), coh * 32 + pl);           // this wave's weight fragments: row coh * 32 + pl of a tap column's image
  auto wfrag = [&](int tx, int j) { return frag(ew0 + (unsigned)(tx * TAPB), kq * KSW + j); };      // this wave's k-step j of the tap column
  bf16x8 wf[3][KSW];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) wf[tx][ks] = wfrag(tx, ks);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  unsigned so_cur = 0, so_nxt = SLOT, so_iss = 2 * SLOT;
  B128_T(0);
  // The weight buffer is single (two do not fit at C = 128): the fragments of phase p + 1 are read into the registers of phase p's fragments as the LAST
  // stage of phase p retires them, k-step by k-step under its MFMAs; every wave has read them when it reaches the barrier of phase p + 1's first
  // stage, which is where the DMAs of phase p + 2 are issued.  They are waited for at the end of that stage (one stage to land, from L2) and made
  // visible by the barrier of the stage that reads them.
  for (int ph = 0; ph < nph; ++ph) {
    const int b = ph / PPM;
    const int d = dil_of(b);
    // b-operand fragment addresses: output pixel (jr, xo + 32 t), tap column tx reads stage pixel jr * W + xo + 32 t + (tx - 1) d - or the slot's zero pixel
    unsigned eoff[3][TPW];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int xx = xo + 32 * t + (tx - 1) * d;
        const bool in = (unsigned)xx < (unsigned)W;
        eoff[tx][t] = in ? eaddr(0u, jr * W + xx) : (unsigned)ZOFF + (unsigned)(kh << 4);      // (zero pixel: swizzle 0)
      }
    const bool last = ph % PPM == PPM - 1 && (!q.sum || b == nb - 1);      // the phase behind which a member's (sum: the only) epilogue runs
    // the epilogue's mask source: fetched two stages early (a dependent HBM round trip in front of the epilogue otherwise)
    uint4 av[SPP][TPW][2];
    const size_t pixg = (size_t)((n_ * H + h0 + jr) * W + xo);         // tile 0 in stage 0 of the band; tile t: + 32 t, stage sp: + sp * RPS rows
    auto stage = [&](auto spc) {
      constexpr int sp = decltype(spc)::value;
      __builtin_amdgcn_s_barrier();                          // this stage is complete and normalised; every wave is done with the stage before (and, sp = 0, holds its weight fragments)
      B128_E(0);
      if constexpr (sp == 0) {
        B128_T(1);
        if (ph > 0 && ph % PPM == 0) stats_flush(b - 1);     // the member that just finished: its partials are visible now
      } else {
        B128_T(5);
      }
      if (sp == 0 && last) {
        const unsigned char* auxp = q.aux[b];
#pragma unroll
        for (int r = 0; r < SPP; ++r)
          if (KQ == 1 || r == kq)                            // (k split: a wave finishes the tile of stage kq only)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
              for (int g = 0; g < 2; ++g)
                av[r][t][g] = auxp ? ldg16(auxp + ((pixg + (size_t)(r * RPS) * W + 32 * t) * CP + co0 + coh * 32 + 16 * g + 8 * kh) * 2) : make_uint4(0, 0, 0, 0);
      }
      B128_E(1);
      unsigned e[3][TPW];
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int t = 0; t < TPW; ++t) e[tx][t] = so_cur + eoff[tx][t];
      const WPhase wn = wphase(ph + 1);
      // 3 KS k-steps of TPW MFMAs in program order, PF fragment reads ahead of the product that consumes them, the stage's DMA instructions dealt between
      // them: stage 0 the WPW weight pieces of the next phase (in FRONT of the rows: the wait at the end of the stage then leaves only the rows in
      // flight) and the NPX row pieces of the stage two ahead (into the slot the stage before this one has just left), stage 1 its NPX row pieces -
      // and, as a k-step's fragment retires, the next phase's weight fragment into its registers.  (An LDS-DMA is an LDS write the compiler cannot
      // tell from the fragment reads' addresses: it keeps both in source order, which is the order wanted here.)
      constexpr int NI = 3 * KSW, PFK = 4 / TPW;             // this wave's k-steps; k-steps of fragment reads in flight (four fragments)
      constexpr int NDMA = sp == 0 ? WPW + NPX : NPX, D0 = 1, DSTEP = (NI - 1 - D0) / (NDMA > 1 ? NDMA - 1 : 1) < 5 ? (NI - 1 - D0) / (NDMA > 1 ? NDMA - 1 : 1) : 5;
      static_assert(DSTEP >= 1 && D0 + (NDMA - 1) * DSTEP < NI, "every DMA has its k-step");
      // A stage whose rows ALL lie outside the image multiplies zeros (d = 15 on 32 rows: the upper kernel row of the two top bands, the lower one of the two
      // bottom bands - a quarter of that member's stages): its DMAs (out of range: zero fill), waits and barriers stay, its fragment reads and MFMAs do not.
      const int hs_ = cur.hb + sp * RPS;
      const bool dead = CP == 256 && (q.dbg & 32) == 0 && (hs_ + RPS <= 0 || hs_ >= H);      // (C = 128: the second loop body costs 30 spilled registers - and scratch traffic would be counted by the vmcnt waits)
      if (dead) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int tx = i / KSW, ks = i % KSW;
          if constexpr (sp == SPP - 1) wf[tx][ks] = wfrag(tx, ks);
          if (i >= D0 && (i - D0) % DSTEP == 0 && (i - D0) / DSTEP < NDMA) {
            const int j = (i - D0) / DSTEP;
            if constexpr (sp == 0) {
              if (j < WPW) { if constexpr (!(RUA_B128_ABLATE & 16)) issue_w1(wn, j); }
              else if constexpr (!(RUA_B128_ABLATE & 4)) issue_x1(nxt, sp, j - WPW, so_iss);
            } else {
              if constexpr (!(RUA_B128_ABLATE & 4)) issue_x1(nxt, sp, j, so_iss);
            }
          }
        }
      } else {
      bf16x8 fr[8][TPW];                                   // ring of 8 k-steps (> PFK: a slot is rewritten only after its product has been issued)
#pragma unroll
      for (int i = 0; i < PFK; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) fr[i][t] = frag(e[i / KSW][t], kq * KSW + i % KSW);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int tx = i / KSW, ks = i % KSW;
        if (i + PFK < NI) {
#pragma unroll
          for (int t = 0; t < TPW; ++t) fr[(i + PFK) & 7][t] = frag(e[(i + PFK) / KSW][t], kq * KSW + (i + PFK) % KSW);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          if constexpr (!(RUA_B128_ABLATE & 8)) acc[sp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tx][ks], fr[i & 7][t], acc[sp][t], 0, 0, 0);
          else asm volatile("" :: "v"(fr[i & 7][t]), "v"(wf[tx][ks]));
        }
        if constexpr (sp == SPP - 1) wf[tx][ks] = wfrag(tx, ks);   // retired: the next phase's fragment takes its registers
        if (i >= D0 && (i - D0) % DSTEP == 0 && (i - D0) / DSTEP < NDMA) {
          const int j = (i - D0) / DSTEP;
          if constexpr (sp == 0) {
            if (j < WPW) { if constexpr (!(RUA_B128_ABLATE & 16)) issue_w1(wn, j); }
            else if constexpr (!(RUA_B128_ABLATE & 4)) issue_x1(nxt, sp, j - WPW, so_iss);
          } else {
            if constexpr (!(RUA_B128_ABLATE & 4)) issue_x1(nxt, sp, j, so_iss);
          }
        }
      }
      if constexpr (!(RUA_B128_ABLATE & 8)) {
        // the interleave, spelled out for the scheduler (left alone it sinks every fragment read to one MFMA in front of its use: an LDS round trip per product)
        __builtin_amdgcn_sched_group_barrier(0x100, PFK * TPW, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          constexpr int wfr = sp == SPP - 1 ? 1 : 0;
          __builtin_amdgcn_sched_group_barrier(0x008, TPW, 0);
          if (i + PFK < NI) __builtin_amdgcn_sched_group_barrier(0x100, TPW + wfr, 0);
          else if (wfr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (i >= D0 && (i - D0) % DSTEP == 0 && (i - D0) / DSTEP < NDMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
      }
      B128_T(3);
      B128_E(2);
      // the next stage's rows: issued one stage ago, waited for only now (two stage times to land).  Younger than them: the NPX row DMAs this stage issued
      // (stage 0: the weights of the next phase were issued in FRONT of its rows and are waited for here as well)
      if constexpr (RUA_B128_ABLATE & (4 | 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPX) : "memory");
      if (bn) {
        if (sp + 1 < SPP) tr_stage(cur, sp + 1, so_nxt);
        else tr_stage(nxt, 0, so_nxt);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the last stage: the next phase's weight fragments are in registers)
      so_iss = so_cur; so_cur = so_nxt; so_nxt = so_nxt + SLOT == (unsigned)(R * SLOT) ? 0u : so_nxt + SLOT;
      B128_T(4);
      B128_E(3);
    };
    stage(B128IC<0>{});
    stage(B128IC<1>{});
    if (last) {
      // ---- member epilogue: bias, ReLU mask from the aux tensor, statistics, one write of the member's band; accumulators cleared
      const unsigned char* auxp = q.aux[b];
      unsigned char* yp = q.ym[b];
      const int smode = q.stats_mode[b];
      const float* tb = tabm + b * 192;
      if constexpr (KQ == 2) {
        // the two k halves of a pixel tile meet: wave kq = 1 hands its stage-0 accumulator to wave kq = 0 and takes that wave's stage-1 accumulator
        // (each wave then finishes ONE of the two tiles: the epilogue's work stays on eight waves)
        float* xb = xbuf + pg * 1024 + lane;
        if (kq == 1) {
#pragma unroll
          for (int k = 0; k < 16; ++k) xb[k * 64] = acc[0][0][k];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[0][0][k] += xb[k * 64];
#pragma unroll
          for (int k = 0; k < 16; ++k) xb[k * 64] = acc[1][0][k];
        }
        __syncthreads();
        if (kq == 1) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[1][0][k] += xb[k * 64];
        }
      }
      float s1[2][8], s2[2][8];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[g][j] = 0.f; s2[g][j] = 0.f; }
#pragma unroll
      for (int r = 0; r < SPP; ++r)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          if (KQ == 2 && r != kq) {                          // the partner finishes this tile
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.f;
            continue;
          }
          float v[2][8];
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float a = acc[r][t][(2 * g) * 4 + j], b2 = acc[r][t][(2 * g + 1) * 4 + j];
              if (g == 0 && j == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
              else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
              v[g][j] = a;
              v[g][4 + j] = b2;
            }
          unsigned char* yrow = yp + ((pixg + (size_t)(r * RPS) * W + 32 * t) * CP) * 2;
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int cl = coh * 32 + 16 * g + 8 * kh;        // channel inside the block's slice
            float a8[8];
            ET<T>::unpack(av[r][t][g], a8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g][j] += tb[cl + j];
            if (auxp) {
              if (q.sum) {                                   // the residual
#pragma unroll
                for (int j = 0; j < 8; ++j) v[g][j] += a8[j];
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[g][j] = (fmaf(tb[64 + cl + j], a8[j], tb[128 + cl + j]) > 0.f) ? v[g][j] : 0.f;
              }
            }
            if (smode == 1) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], v[g][j], s2[g][j]); }
            } else if (smode == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[g][j] += v[g][j]; s2[g][j] = fmaf(v[g][j], a8[j], s2[g][j]); }
            }
            stg16(yrow + (co0 + cl) * 2, ET<T>::pack(v[g]));
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.f;
        }
      if (smode != 0) {
        // 32 partial sums per lane -> per-channel sums over the wave's 32 pixel columns by a transposing butterfly (conv_band64m's), lane l ends with value l
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
      B128_T(6);                                             // (the stores are younger than everything the counted waits wait for: they only make those waits stricter)
    }
    cur = nxt;
    nxt = phase(ph + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");       // the over-issued DMAs of the last stages
  __syncthreads();
  stats_flush(nb - 1);
#ifdef RUA_B128_STAMPS
  B128_T(7);
  if (tid == 0 && q.stamps)
    for (int i = 0; i < 8; ++i) q.stamps[(size_t)job * 8 + i] = tacc[i];
#endif
}

// (Measured and not kept, round 5: the same job with the waves SPECIALISED - waves 4 - 7 issue every DMA of the block, wait for the landings and do the in-place
// BatchNorm, waves 0 - 3 only read fragments and multiply, a whole row x 32 output channels each (conv_band128w, 250 lines).  Bit-identical; 38 - 40 us against 37 - 39 us
// for the first-convolution group and 50 against 43 us for the data-gradient group, whose member epilogue - mask, statistics, stores - then runs on four waves
// instead of eight.  What the stamps had blamed on DMA instructions holding up the MFMAs behind them was mostly the WEIGHT DMAs: staged a fragment per lane they
// touched 32 partly used lines per instruction and cost ~30 ns each in the texture path against ~12.5 ns for a row instruction; as whole swizzled rows - the
// form above - the first form went 43 - 46 -> 37 - 39 us and the specialised one lost its reason.)

template <int CP, int W, int KC, int COS> __global__ __launch_bounds__(512) void conv_band128m(const Band128K q) { conv_band128_body<CP, W, KC, COS>(q); }

// ---- host side (called by rua_conv_fwd_group, conv_mfma.hip) -----------------------------------------------------------------------
static int band128_rows(int C, int W) { return 2 * ((32768 / ((C > 128 ? 128 : C) * 2)) / W); }          // rows per band (two stages of 32 KB of <= 128 staged channels)
bool rua_band128m_ok(const rua_conv_desc* d, int n) {
  if (n < 1 || n > RUA_MAX_BRANCH) return false;
  const rua_conv_desc& a = d[0];
  const int Cc = a.Cout;
  if (Cc == 128) { if (!(g_tune.conv_band128m & 1) || (a.W != 64 && a.W != 128)) return false; }
  else if (Cc == 64) { if (!(g_tune.conv_band128m & 2) || (a.W != 64 && a.W != 128 && a.W != 256)) return false; }
  else if (Cc == 256) { if (!(g_tune.conv_band128m & 4) || a.W != 32) return false; }
  else return false;
  if (a.dtype != RUA_BF16 || a.H % band128_rows(Cc, a.W) != 0 || (long long)a.N * a.H * a.W < 1024) return false;
  for (int i = 0; i < n; ++i) {
    const rua_conv_desc& m = d[i];
    const rua_conv_seg& g = m.seg[0];
    if (m.nseg != 1 || m.dtype != RUA_BF16 || g.taps != 9 || g.up_shift != 0 || g.C != Cc || m.Cout != Cc || m.stride != 1 ||
        m.out_stride != 1 || m.OH != m.H || m.OW != m.W || g.Hs != m.H || g.Ws != m.W || g.dil < 1) return false;
    if (m.N != a.N || m.H != a.H || m.W != a.W || !m.y) return false;
    for (int j = 0; j < i; ++j) if (d[j].y == m.y) return false;               // independent outputs
    if (m.accumulate || m.out_relu || m.bias_more[0] || m.bias_more[1] || m.bias_more[2]) return false;
    if (!(m.aux_mode == 0 || (m.aux_mode == 2 && m.aux))) return false;
    if (m.stats_mode != 0 && (!m.stats || m.stats_replicas < 1 || (m.stats_replicas & (m.stats_replicas - 1)))) return false;
    if (m.stats_mode == 2 && m.aux_mode != 2) return false;
    if (m.stats_mode < 0 || m.stats_mode > 2) return false;
    if (m.in_fold && Cc != 64) return false;                                   // (C >= 128: the coefficients are given - rua_bn_fwd makes them at those levels)
    if ((m.in_fold != nullptr) != (a.in_fold != nullptr) || (m.in_scale != nullptr) != (a.in_scale != nullptr) || m.in_relu != a.in_relu) return false;
    if (m.in_fold && (m.in_scale || m.in_shift)) return false;
    if ((m.in_shift != nullptr) != (m.in_scale != nullptr)) return false;
  }
  return true;
}

template <int CP, int W, int KC, int COS>
static int band128_launch(const Band128K& q, hipStream_t st) {
  constexpr int smem = 3 * (32768 + 256) + 3 * COS * KC * 2 + (RUA_MAX_BRANCH * 2 * CP + RUA_MAX_BRANCH * 192 + 8 * 64 + (COS == 32 ? 4 * 1024 : 0)) * 4;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static RuaPerDevFlag attr;
  if (!attr.get()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band128m<CP, W, KC, COS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr.get() = true; }
  hipLaunchKernelGGL((conv_band128m<CP, W, KC, COS>), dim3(q.njobs), dim3(512), smem, st, q);
  RUA_LAUNCH_CHECK("conv_band128m");
  return RUA_OK;
}

int rua_launch_band128m(const rua_conv_desc* d, int n, hipStream_t st) {
  Band128K q;
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
  const int Cc = a.Cout;
  q.dbg = g_tune.band_dbg;
  q.stamps = (unsigned long long*)(uintptr_t)g_tune.dbg_ptr;
  q.has_fold = a.in_fold ? 1 : 0;
  q.has_bn = (a.in_fold || a.in_scale) ? 1 : 0;
  q.in_relu = a.in_relu;
  q.N = a.N; q.H = a.H;
  RUA_CHECK_ARG((size_t)a.N * a.H * a.W * Cc * 2 < 0x7FFFFF00ull, "rua_conv_fwd_group: tensor of 2 GiB or more");
  q.xbytes = (unsigned)((size_t)a.N * a.H * a.W * Cc * 2);
  q.bands = a.H / band128_rows(Cc, a.W);
  q.njobs = a.N * q.bands * (Cc == 256 ? 8 : Cc / 64);
  if (Cc == 256) return band128_launch<256, 32, 128, 32>(q, st);
  if (Cc == 128) return a.W == 64 ? band128_launch<128, 64, 128, 64>(q, st) : band128_launch<128, 128, 128, 64>(q, st);
  return a.W == 64 ? band128_launch<64, 64, 64, 64>(q, st) : (a.W == 128 ? band128_launch<64, 128, 64, 64>(q, st) : band128_launch<64, 256, 64, 64>(q, st));
}

// The summed second convolutions of a level-3 / level-4 ResBlock (model2.py:26-31: out = x + sum_b conv_b(a2_b); rua_conv_fwd with one 3x3 segment per branch; was
// conv_dmap_chain, K x 3) in conv_band128m's form: the accumulators run over the members, ONE epilogue (the sum of the biases, the residual) - tuning key conv_band128m, bit 3.
bool rua_band128_sum_ok(const rua_conv_desc* d) {
  if (!(g_tune.conv_band128m & 8) || d->nseg < 2 || d->nseg > RUA_MAX_BRANCH || d->dtype != RUA_BF16) return false;
  const int Cc = d->Cout;
  if (Cc == 128) { if (!(g_tune.conv_band128m & 1) || (d->W != 64 && d->W != 128)) return false; }
  else if (Cc == 256) { if (!(g_tune.conv_band128m & 4) || d->W != 32) return false; }
  else return false;
  if (d->H % band128_rows(Cc, d->W) != 0 || (long long)d->N * d->H * d->W < 1024 || !d->y) return false;
  if (d->stride != 1 || d->out_stride != 1 || d->OH != d->H || d->OW != d->W || d->accumulate || d->out_relu) return false;
  if (d->in_scale || d->in_shift || d->in_fold) return false;
  if (!(d->aux_mode == 0 || (d->aux_mode == 1 && d->aux))) return false;
  if (d->stats_mode != 0 && d->stats) return false;
  for (int i = 0; i < d->nseg; ++i) {
    const rua_conv_seg& g = d->seg[i];
    if (g.taps != 9 || g.up_shift != 0 || g.C != Cc || g.Hs != d->H || g.Ws != d->W || g.dil < 1 || !g.x || !g.w) return false;
  }
  for (int i = d->nseg - 1; i < 3; ++i) if (d->bias_more[i]) return false;      // at most one bias per segment
  return (size_t)d->N * d->H * d->W * Cc * 2 < 0x7FFFFF00ull;
}

int rua_launch_band128_sum(const rua_conv_desc* d, hipStream_t st) {
  Band128K q;
  memset(&q, 0, sizeof(q));
  const int n = d->nseg, Cc = d->Cout;
  q.nb = n; q.sum = 1;
  for (int i = 0; i < n; ++i) {
    q.x[i] = (const unsigned char*)d->seg[i].x; q.w[i] = (const unsigned char*)d->seg[i].w; q.d[i] = d->seg[i].dil;
    q.bias[i] = i == 0 ? d->bias : d->bias_more[i - 1];
    q.ym[i] = (unsigned char*)d->y; q.stats_R[i] = 1;
  }
  q.aux[n - 1] = d->aux_mode == 1 ? (const unsigned char*)d->aux : nullptr;
  for (int i = n; i < RUA_MAX_BRANCH; ++i) { q.d[i] = 1; q.stats_R[i] = 1; }
  q.dbg = g_tune.band_dbg;
  q.stamps = nullptr;
  q.N = d->N; q.H = d->H;
  q.xbytes = (unsigned)((size_t)d->N * d->H * d->W * Cc * 2);
  q.bands = d->H / band128_rows(Cc, d->W);
  q.njobs = d->N * q.bands * (Cc == 256 ? 8 : Cc / 64);
  if (Cc == 256) return band128_launch<256, 32, 128, 32>(q, st);
  return d->W == 64 ? band128_launch<128, 64, 128, 64>(q, st) : band128_launch<128, 128, 128, 64>(q, st);
}

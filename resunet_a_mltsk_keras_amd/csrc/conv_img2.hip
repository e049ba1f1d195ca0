// conv_img2<W>: the 3x3 (dilation 1) convolutions of the two deepest levels - 8 x 8 x 1024 and 16 x 16 x 512 at batch 8 (model2.py:109-112 and the decoder's
// mirror :120-123; forward and data gradient: 12 launches of ~28 us per step on conv_dmap + conv_splitk_finish, 10 - 13 % of the bf16 peak) - with what round 5
// learned on conv_band128m (conv_band128.hip):
//   * WHOLE IMAGES RESIDENT: a block owns 256 output pixels = whole images (four of 8 x 8, one of 16 x 16) x a 64-channel slice of the outputs x a 128-channel
//     chunk of the inputs; its [256 pixels][128 channels] input tile (64 KB) enters LDS ONCE and serves all nine taps by shifted fragment reads (a tap that leaves
//     the image reads a zero pixel) - conv_dmap staged every input pixel nine times;
//   * the weights of a kernel row x the slice x the chunk (3 x 64 x 128 = 48 KB) are staged as whole, swizzled, COALESCED 256-byte rows (conv_img, round 4, gathered
//     32-byte fragment pieces straight into registers: ~30 ns per instruction in the texture path, and no gain over conv_dmap) and live in REGISTERS for the phase
//     (24 fragments per wave; one LDS read per MFMA: the pixel fragment);
//   * phases = kernel rows; a phase has two stages = the wave's two pixel tiles: the DMAs of the next phase's weights are dealt between the MFMAs of stage 0, the
//     fragments of the next phase replace the retired ones under the MFMAs of stage 1 (conv_band128m's scheme with the tile in place of the ring slot);
//   * K is split over blocks by input-channel chunks (8 x 2 x 16 = 256 blocks at 8 x 8 x 1024, 4 x 8 x 8 at 16 x 16 x 512): every block stores its fp32 partial tile
//     into its slice's slab, conv_splitk_finish (conv_mfma.hip) sums the slabs in a fixed order and runs the shared epilogue (bias, mask, statistics, store) - the
//     same finisher, slab layout and bit-reproducibility as the split-K path it replaces.
// (Measured and not kept: TWO weight images - the input tile + 2 x 48 KB are exactly the 160 KB, so the zero pixel went and a tap that leaves the image read its own
// pixel and was cleared by a per-lane mask, four v_and per fragment - with the weights of phase p + 2 issued in phase p: 28.8 vs 25.8 us per convolution, step
// -0.047 vs -0.07 ms.  The ANDs sit between a fragment read and its MFMA.)
#include "common.h"

struct Img2K {
  const unsigned char* x; const unsigned char* w; float* ws;
  int C, Cout, H, M, ngrp, nco, ksplit, cpb;     // ngrp pixel groups of 256, nco output slices of 64, cpb input chunks of 128 per block
  unsigned xbytes, wbytes;
};
template <int V> struct Img2IC { static constexpr int value = V; };

template <int W>
__device__ __forceinline__ void conv_img2_body(const Img2K& q) {
  constexpr int NW = 8, HW = W * W, PXB = 256, NPX = 256;
  constexpr int XB = NPX * PXB, ZOFF = XB, SWO = XB + 256;                    // input tile, zero pixel, weight image (multiples of 256 from LDS address 0)
  constexpr int KS = 8, TAPB = 64 * PXB, WBUF = 3 * TAPB, WPW = (WBUF / 1024) / NW, XPW = (XB / 1024) / NW;   // 6 weight / 8 input DMA instructions per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 31, kh = lane >> 5;
  const int pg = wv >> 1, coh = wv & 1;                                       // the wave's 64 pixels (two tiles) and output-channel tile
  const int nwg = (int)gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // consecutive jobs (the output slices of one input tile first) share an XCD's L2
  const int cs = job % q.nco, t1 = job / q.nco;
  const int ks = t1 % q.ksplit, grp = t1 / q.ksplit;
  const int m0 = grp * NPX, co0 = cs * 64, ch0 = ks * q.cpb;
  const int C = q.C;
  if (tid < 16) *reinterpret_cast<uint4*>(smem + ZOFF + tid * 16) = make_uint4(0, 0, 0, 0);

  // ---- DMA addressing (conv_band128m's): lane l of instruction i moves piece psrc of tile pixel 4 i + (l >> 4) into slot (l & 15) = psrc ^ (pixel & 15)
  const int qd0 = wv * 4 + (lane >> 4);
  const int psrc = (lane & 15) ^ (qd0 & 15);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.x, q.xbytes), rw = make_rsrc(q.w, q.wbytes);
  const unsigned xsrc = (unsigned)(((m0 + qd0) * C + psrc * 8) * 2);         // instruction k: + 32 k pixels
  auto issue_x = [&](int ch) {
#pragma unroll
    for (int k = 0; k < XPW; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_p)(smem + (k * NW + wv) * 1024), 16, xsrc + (unsigned)((k * 32 * C + ch * 128) * 2), 0, 0, 0);
  };
  const int wrow0 = 4 * (wv & 3) + (lane >> 4);                               // the lane's weight row modulo 16 (four rows per instruction)
  const unsigned wsrc = (unsigned)(((co0 + wrow0) * C + (((lane & 15) ^ wrow0) * 8)) * 2);
  // weight instruction i (of the wave) of phase (chunk ch, kernel row ty): tap column idx >> 4, rows 16 * ((idx & 15) >> 2) + wrow0
  auto issue_w1 = [&](int ch, int ty, int i, bool ok) {
    const int idx = i * NW + wv;
    const unsigned off = ok ? wsrc + (unsigned)((((ty * 3 + (idx >> 4)) * q.Cout + 16 * ((idx & 15) >> 2)) * C + ch * 128) * 2) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_p)(smem + SWO + idx * 1024), 16, off, 0, 0, 0);
  };
  // fragment addresses: e = base + row * 256 + ((sw >> 1) << 5) + (((kh ^ sw) & 1) << 4), k-step ks at e ^ (ks << 5) (sw = row & 15)
  auto frag = [&](unsigned e, int k) { return *reinterpret_cast<const bf16x8*>(smem + (e ^ (unsigned)(k << 5))); };
  auto eaddr = [&](unsigned base, int row) {
    const int sw = row & 15;
    return base + (unsigned)(row * PXB + ((sw >> 1) << 5) + (((kh ^ sw) & 1) << 4));
  };
  const unsigned ew0 = eaddr((unsigned)SWO, coh * 32 + pl);
  auto wfrag = [&](int tx, int k) { return frag(ew0 + (unsigned)(tx * TAPB), k); };
  // the wave's output pixels: tile t = tile pixel pg * 64 + 32 t + pl = (image, row y, column x) of the block's whole images
  int qy[2], qx[2], qq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { qq[t] = pg * 64 + t * 32 + pl; qy[t] = (qq[t] % HW) / W; qx[t] = qq[t] % W; }

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

  // ---- prologue: the input tile of the first chunk and the weights of its first kernel row ---------------------------------------------
  issue_x(ch0);
#pragma unroll
  for (int i = 0; i < WPW; ++i) issue_w1(ch0, 0, i, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 wf[3][KS];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int k = 0; k < KS; ++k) wf[tx][k] = wfrag(tx, k);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int nph = 3 * q.cpb;
  for (int ph = 0; ph < nph; ++ph) {
    const int chl = ph / 3, ty = ph - 3 * chl;
    const int nch = (ph + 1) / 3, nty = (ph + 1) - 3 * nch;                   // the next phase
    const bool nok = ph + 1 < nph;
    // tap (ty, tx) of output pixel (y, x): input pixel (y + ty - 1, x + tx - 1) of the same image, or the zero pixel
    unsigned eoff[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int yy = qy[t] + ty - 1, xx = qx[t] + tx - 1;
        const bool in = (unsigned)yy < (unsigned)W && (unsigned)xx < (unsigned)W;
        eoff[tx][t] = in ? eaddr(0u, qq[t] + (ty - 1) * W + (tx - 1)) : (unsigned)ZOFF + (unsigned)(kh << 4);
      }
    auto stage = [&](auto spc) {
      constexpr int sp = decltype(spc)::value;
      __builtin_amdgcn_s_barrier();                          // sp = 0: every wave holds this phase's fragments (the weight image is free); sp = 1: the next phase's weights have landed
      constexpr int NI = 3 * KS, PF = 4, NDMA = sp == 0 ? WPW : 0, D0 = 1, DSTEP = 3;
      bf16x8 fr[8];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = frag(eoff[i / KS][sp], i % KS);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int tx = i / KS, k = i % KS;
        if (i + PF < NI) fr[(i + PF) & 7] = frag(eoff[(i + PF) / KS][sp], (i + PF) % KS);
        acc[sp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tx][k], fr[i & 7], acc[sp], 0, 0, 0);
        if constexpr (sp == 1) wf[tx][k] = wfrag(tx, k);     // retired: the next phase's fragment takes its registers
        if (NDMA > 0 && i >= D0 && (i - D0) % DSTEP == 0 && (i - D0) / DSTEP < NDMA) issue_w1(nch + ch0, nty, (i - D0) / DSTEP, nok);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + PF < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1 + sp, 0);
        else if (sp == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (NDMA > 0 && i >= D0 && (i - D0) % DSTEP == 0 && (i - D0) / DSTEP < NDMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    };
    stage(Img2IC<0>{});
    stage(Img2IC<1>{});
    if (ty == 2 && chl + 1 < q.cpb) {                        // next input chunk (blocks of several chunks: the deepest level of a d7 network): the tile is reloaded in place
      __builtin_amdgcn_s_barrier();
      issue_x(ch0 + chl + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  // ---- the partial tile -> this K slice's slab [ks][M][Cout] fp32 (conv_splitk_finish sums the slices and runs the epilogue) -------------------
  float* slab = q.ws + ((size_t)ks * q.M + m0) * q.Cout + co0 + coh * 32 + 8 * kh;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float* row = slab + (size_t)qq[t] * q.Cout;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 lo, hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[t][(2 * g) * 4 + j], b2 = acc[t][(2 * g + 1) * 4 + j];
        if (g == 0 && j == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        lo[j] = a; hi[j] = b2;
      }
      *reinterpret_cast<f32x4*>(row + 16 * g) = lo;
      *reinterpret_cast<f32x4*>(row + 16 * g + 4) = hi;
    }
  }
}
template <int W> __global__ __launch_bounds__(512) void conv_img2(const Img2K q) { conv_img2_body<W>(q); }

// eligibility + K slices (0: not this kernel)
int rua_pick_img2(const rua_conv_desc* d) {
  if (!g_tune.conv_img2 || g_conv_group || d->dtype != RUA_BF16 || d->nseg != 1 || d->in_scale || d->in_fold) return 0;
  const rua_conv_seg& g = d->seg[0];
  if (g.taps != 9 || g.dil != 1 || d->stride != 1 || d->out_stride != 1 || g.up_shift != 0 || g.Hs != d->H || g.Ws != d->W || d->H != d->W || (d->W != 8 && d->W != 16) ||
      d->OH != d->H || d->OW != d->W) return 0;
  if (g.C % 128 || d->Cout % 64) return 0;
  const long long M = (long long)d->N * d->H * d->W;
  if (M % 256 || M * d->Cout * 4 >= (1ll << 31) || M * g.C * 2 >= 0x7FFFFF00ll || 9ll * d->Cout * g.C * 2 >= 0x7FFFFF00ll) return 0;
  const int nchunk = g.C / 128, groups = (int)(M / 256) * (d->Cout / 64);
  const size_t ws_usable = d->workspace_bytes > 4096 ? (size_t)d->workspace_bytes - 4096 : 0;
  const long long slabs = d->workspace ? (long long)(ws_usable / ((size_t)M * d->Cout * sizeof(float))) : 0;
  int ks = (rua_cu_count() + groups - 1) / groups;          // K slices: blocks to cover the chip once ...
  if (ks > nchunk) ks = nchunk;
  if (ks > slabs) ks = (int)slabs;                           // ... that the workspace holds slabs for ...
  while (ks > 1 && nchunk % ks) --ks;                        // ... of whole input chunks
  if (ks < 2) return 0;                                      // (the epilogue lives in the finisher)
  return ks;
}

int rua_launch_conv_img2(ConvK& k, const rua_conv_desc* d, int KS, hipStream_t st) {
  const rua_conv_seg& g = d->seg[0];
  Img2K q;
  q.x = (const unsigned char*)g.x; q.w = (const unsigned char*)g.w; q.ws = (float*)d->workspace;
  q.C = g.C; q.Cout = d->Cout; q.H = d->H; q.M = (int)k.M; q.ngrp = (int)(k.M / 256); q.nco = d->Cout / 64; q.ksplit = KS; q.cpb = (g.C / 128) / KS;
  q.xbytes = (unsigned)((size_t)k.M * g.C * 2); q.wbytes = (unsigned)((size_t)9 * d->Cout * g.C * 2);
  k.nbn = (d->Cout + 63) / 64; k.nbm = (int)(k.M / 256); k.ksplit = KS; k.stages_per_split = 0; k.ws = (float*)d->workspace; k.cnt = nullptr;
  const unsigned grid = (unsigned)(q.ngrp * q.nco * KS);
  constexpr int smem = 256 * 256 + 256 + 3 * 64 * 256;
  static RuaPerDevFlag attr;
  if (!attr.get()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_img2<8>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_img2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr.get() = true;
  }
  if (d->W == 8) hipLaunchKernelGGL((conv_img2<8>), dim3(grid), dim3(512), smem, st, q);
  else hipLaunchKernelGGL((conv_img2<16>), dim3(grid), dim3(512), smem, st, q);
  RUA_LAUNCH_CHECK("conv_img2");
  return rua_splitk_finish_bf16(k, st);
}

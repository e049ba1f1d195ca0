/* librua_hip.so — C ABI of the MI355X (gfx950) ResUnet-a training-path kernels.
 *
 * The reference (thimabru1010/ResUnet-a_mltsk_keras) has NO native/FFI interface: its hot
 * path is the list of Keras layer call sites that TensorFlow executes.  Every entry point
 * below therefore cites the reference call site(s) (file:line under /root/reference) whose
 * arithmetic it replaces.  All tensors are channels-last (NHWC, the reference's own layout,
 * train_ISPRS.py:73-92) in device memory owned by the caller; `dtype` selects the storage
 * type of activations (RUA_F32 or RUA_BF16); accumulation is always fp32 (statistics fp64).
 * Every function returns 0 on success or a negative code (text via rua_last_error()), takes
 * the HIP stream as its last argument (`void*` = hipStream_t), allocates nothing and keeps no
 * global state besides the opt-in tuning switches (rua_set_tuning).  Nothing here depends on PyTorch.
 */
#ifndef RUA_HIP_H
#define RUA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RUA_F32 0
#define RUA_BF16 1
#define RUA_MAX_SEG 6
#define RUA_MAX_BRANCH 4
#define RUA_MAX_WGRAD_GROUP 8   /* members of one rua_conv_wgrad_group call (both convolutions of every branch of a ResBlock) */

#define RUA_OK 0
#define RUA_ERR_ARG (-1)      /* shape/alignment precondition violated */
#define RUA_ERR_LAUNCH (-2)   /* HIP launch error */

int rua_version(void);
const char* rua_last_error(void);
int rua_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---- segmented implicit-GEMM convolution (MFMA) --------------------------------------
 * y[n,h,w,co] = sum over segments s, taps t, channels c of
 *               src_s[n, (h*stride+dy_t*dil_s)>>up_s, (w*stride+dx_t*dil_s)>>up_s, c] * w_s[t][co][c]
 * Replaces: KL.Conv2D 3x3 dilated 'same' (model2.py:19-24), 1x1 / strided 1x1 (:101-111),
 * Concatenate+Conv2D of PSPPooling/combine (:73-78,:83-85) with the concat never
 * materialised, UpSampling2D-nearest folded into the read (:55-60,:91), the n-ary Add of the
 * ResBlock (:27-31) as the `aux` residual, and — with transposed/flipped weights — the
 * data-gradient of all of them.  Channel counts must be multiples of 8 (bf16) / 4 (f32). */
typedef struct rua_conv_seg {
  const void* x;       /* source activations [N][Hs][Ws][C] */
  const void* w;       /* weights [taps][Cout][C], same dtype as activations */
  int32_t C, Hs, Ws;
  int32_t up_shift;    /* nearest upsample by 2^up_shift folded into the read */
  int32_t dil;         /* dilation of the 3x3 taps */
  int32_t taps;        /* 1 or 9 */
} rua_conv_seg;

/* model2.py:17-24 BatchNormalization in training mode, folded into the consuming convolution (rua_conv_desc.in_fold) */
typedef struct rua_bn_fold {
  const double* stats;          /* [replicas][2][C] fp64: sum x, sum x^2 of the conv input, complete when the conv starts */
  int32_t replicas, pad;
  double count, bessel_n;       /* elements per channel; count for the unbiased moving variance (count if no replication) */
  float eps, momentum;
  const float* gamma; const float* beta;
  float* moving_mean; float* moving_var;         /* updated once per call (NULL: not) */
  float* scale; float* shift; float* mean; float* rstd;   /* published by one workgroup for later kernels (NULL mean / rstd: not) */
} rua_bn_fold;

typedef struct rua_conv_desc {
  rua_conv_seg seg[RUA_MAX_SEG];
  int32_t nseg;
  int32_t N, H, W, Cout;       /* logical output grid (GEMM rows = N*H*W) */
  int32_t stride;              /* input coordinate = output coordinate * stride */
  int32_t dtype;
  const float* bias;           /* [Cout] or NULL */
  const void* aux;             /* [N][H][W][Cout] or NULL */
  int32_t aux_mode;            /* 0 none, 1 y += aux (residual), 2 y *= (mscale*aux+mshift > 0) (ReLU mask), 3 statistics only */
  const float* mscale;         /* per-channel, NULL => 1 */
  const float* mshift;         /* per-channel, NULL => 0 */
  int32_t out_relu;
  int32_t accumulate;          /* y += result instead of y = result */
  void* y;                     /* [N][OH][OW][Cout], written at (h*out_stride, w*out_stride) */
  int32_t out_stride, OH, OW;
  double* stats;               /* [stats_replicas][2][Cout] atomically accumulated, or NULL */
  int32_t stats_mode;          /* 1: sum v, sum v^2   2: sum v, sum v*aux */
  void* workspace;             /* optional fp32 scratch, >= 2 * rua_conv_workspace_bytes() + 4096: enables split-K for small
                                  output grids (one N*H*W*Cout slab per K slice, plain stores, summed in a fixed order:
                                  deterministic).  The LAST 4096 bytes are reserved for the tile ticket counters of the
                                  optional in-launch reduction (RUA_DMAP_FUSED_FINISH=1): zero them once (e.g. allocate
                                  the workspace zero-filled); every call leaves them zero.  The rest: contents on entry
                                  irrelevant, undefined on return */
  int64_t workspace_bytes;
  int32_t stats_replicas;      /* power of two >= 1: block b adds into replica b % R (spreads atomic contention);
                                  the finalize kernels sum the replicas */
  const float* bias_more[3];   /* further [Cout] bias vectors (or NULL) added after `bias`, in this order: the branch-final convs
                                  of a ResBlock run as ONE concatenated conv whose bias is the sum of the branches' biases */
  /* Normalise on load (model2.py:17-24: BatchNormalization -> ReLU -> Conv2D): segment 0 is read as
   * [relu](in_scale[c] * x + in_shift[c]) - applied as the tile lands in LDS, zero padding stays zero - so the normalised
   * copy of the conv input never exists in HBM.  NULL: plain read.  Only where rua_conv_fused_input_ok() says 1. */
  const float* in_scale;
  const float* in_shift;
  int32_t in_relu;
  int32_t pad_fold;
  /* Training-mode BatchNorm whose coefficients the CONSUMER derives itself (no coefficient launch between the producer of the
   * statistics and this convolution): every workgroup sums the replicated fp64 statistics of the input in its prologue, one
   * designated workgroup publishes scale / shift / mean / rstd (ReLU masks, backward) and updates the moving statistics.
   * Non-NULL: in_scale / in_shift must be NULL; in_relu applies.  Same shapes as in_scale (rua_conv_fused_input_ok). */
  const struct rua_bn_fold* in_fold;
} rua_conv_desc;
int rua_conv_fwd(const rua_conv_desc* d, void* stream);
/* n (<= RUA_MAX_BRANCH) INDEPENDENT convolutions - the dilation branches of a ResBlock (model2.py:26-31) - with the results of n
 * rua_conv_fwd calls; members that land on the same kernel are issued as ONE grid (no drain / launch gap between the branches) */
int rua_conv_fwd_group(const rua_conv_desc* d, int n, void* stream);
int rua_conv_group_last_grids(void);             /* grids the calling thread's latest rua_conv_fwd_group issued (1: all members in one) */
/* C = Cout = 64, 3x3, bf16, W % 128 == 0, H % 4 == 0: the members run as ONE conv_band64m launch (row streaming, 4-row bands, every
 * member normalised on load by its own in_fold / in_scale, its own bias / ReLU mask / statistics epilogue) - the only path on which
 * normalise-on-load is served at 64 channels.  rua_conv_group_band_ok asks without launching; ..._last_band reports the last call. */
int rua_conv_group_band_ok(const rua_conv_desc* d, int n);
int rua_conv_group_last_band(void);
/* Members on conv_dmap (bf16, Cout >= 128, C % 64 == 0, no K split) with the same tiles run BACK TO BACK inside one grid (conv_dmap_chain:
 * a block walks every member over its pixel tile, the LDS-DMA ring never drains between members); ..._last_chain = how many did (0: none). */
int rua_conv_group_last_chain(void);
/* The n-ary Add of a ResBlock (model2.py:26-31: out = x_input + sum of the branches) with the sum kept ON CHIP: n (<= RUA_MAX_BRANCH)
 * single-segment convolutions into the SAME output y = aux_0 + sum_i conv_i - the results of n rua_conv_fwd calls of which member 0
 * writes (accumulate 0, optional residual aux_mode 1) and members i > 0 accumulate (accumulate 1, no aux).  Every member carries its
 * own normalise-on-load (in_fold, or in_scale / in_shift, + in_relu: the same kind for all members) and bias.  Where the shape allows
 * (bf16, C = Cout = 32, 3x3, dilation <= 32, W % 128 == 0, H % 8 == 0, >= 65536 pixels, no statistics / output ReLU) this is ONE
 * launch of conv_band32 that writes y once; otherwise the members are launched one by one (same results). */
int rua_conv_fwd_sum(const rua_conv_desc* d, int n, void* stream);
int rua_conv_sum_last_kernel(void);              /* the calling thread's latest rua_conv_fwd_sum: 1 one conv_band32 launch, 2 one conv_band64 launch (C = Cout = 64,
                                                    W % 128 == 0, H % 4 == 0), 0 member by member */
int rua_conv_sum_kernel(const rua_conv_desc* d, int n);   /* the same answer for a set of members, without launching anything */
int rua_conv_smem_bytes(const rua_conv_desc* d);
int64_t rua_conv_workspace_bytes(const rua_conv_desc* d);   /* bytes of ONE slab (N*H*W*Cout fp32); split-K uses up to 32 */
/* profiling only (bench.py): timing events without the system-scope release a default event performs when recorded */
void* rua_prof_event_create(void);
int rua_prof_event_record(void* ev, void* hip_stream);
int rua_prof_event_elapsed_us(void* start, void* stop, double* us);   /* waits for `stop` */
void rua_prof_event_destroy(void* ev);
void rua_profile_mid_event(void* hip_event);
int rua_profile_mid_event_fired(void);          /* 1 if the call made since the event was armed recorded it (had a second launch) */    /* profiling only: the calling thread's NEXT two-launch call (split-K conv, all-taps weight
                                                   gradient) records this hipEvent_t between its main kernel and its second launch; one shot */
int rua_conv_last_ksplit(void);                  /* K slices of the calling thread's latest rua_conv_fwd (1: single pass, no finisher) */
int rua_conv_tile_bn(const rua_conv_desc* d);   /* 32 / 64 / 128 and */
int rua_conv_kernel_id(const rua_conv_desc* d); /* 0: conv_igemm (register-staged), 1: conv_dma (LDS-DMA, bf16), 2: conv_dmap (LDS-DMA, pipelined across the stage barrier),
                                                   3: conv_halo (input + halo resident in LDS), 4: conv_pw (narrow 1x1, per-wave streaming),
                                                   5: conv_strip (row-streaming 3x3 at C = Cout = 32, BatchNorm + ReLU applied on load) */
int rua_conv_fused_input_ok(const rua_conv_desc* d);   /* 1: this shape runs on a kernel that honours in_scale / in_shift / in_relu */
int rua_conv_tile_bm(const rua_conv_desc* d);   /* 128 / 256: which conv_igemm<T,BM,BN> instantiation a descriptor launches */

/* ---- weight gradient (MFMA, split over pixels, fp32 atomic accumulation) ----------------
 * dW[t][co][c] += sum_pixels dy[n,h,w,co] * a[n, h*stride+dy_t*dil, w*stride+dx_t*dil, c]
 * Replaces the kernel-gradient of every KL.Conv2D above (Keras autodiff inside
 * train_on_batch, train_ISPRS.py:131,148). */
typedef struct rua_wgrad_desc {
  const void* a;  int32_t C, Hs, Ws;        /* conv input  [N][Hs][Ws][C]  */
  const void* dy; int32_t Cout, H, W;       /* out-gradient [N][H][W][Cout] */
  int32_t N, stride, dil, taps, dtype;
  float* dw;                                /* [taps][Cout][C] fp32, accumulated */
  void* workspace;                          /* optional fp32 scratch of rua_wgrad_workspace_bytes(): enables the all-taps kernel of */
  int64_t workspace_bytes;                  /* the two top levels (C = Cout in {32,64}, 3x3, W % 64 == 0, bf16; uses the front of it,
                                               contents on entry irrelevant) and the per-wave kernel of the narrow 1x1 convolutions
                                               (taps 1, C * Cout <= 4096, bf16), whose replica accumulators and ticket counters are the
                                               LAST 264 KiB (16 * 16 KiB + 8 KiB): zero them once before the first call, every call
                                               leaves them zero.  One workspace per concurrently running stream. */
  /* Normalise on load, as in rua_conv_desc: a is read as [relu](in_scale[c] * a + in_shift[c]), zero padding stays zero.
   * Only the all-taps kernel honours it (rua_wgrad_kind() == 1); other shapes reject a non-NULL in_scale. */
  const float* in_scale;
  const float* in_shift;
  int32_t in_relu;
  /* Deferred reduction: with defer != 0 a call whose kernel leaves partial sums (the all-taps block partials, the K slices'
   * slabs) does NOT launch its reduction; `workspace` must then stay untouched until rua_wgrad_reduce_batch has consumed the
   * record rua_wgrad_plan() returns for this descriptor (one workspace per deferred call; size: rua_wgrad_workspace_bytes). */
  int32_t defer;
  /* Members of a rua_conv_wgrad_group call: how many weight gradients share the launch (0 / 1: this one alone).  The all-taps
   * kernel sizes its persistent grid for a share of the chip - one round of blocks for the whole group, 1 / group_members of the
   * block partials to write and to reduce.  Set it on every member (rua_wgrad_plan must see the same value as the launch). */
  int32_t group_members;
  int32_t pad_group;
  /* overwrite_dev (optional): a DEVICE int32 read when the kernel runs.  Non-zero: the caller guarantees that dw holds zeros and that this call is its only
   * writer before it is read (the step's gradient arena: zeroed by the optimizer, every weight has one producer) - kernels that end in a read-modify-write of dw
   * (one K slice per element, the slab / block-partial reductions) then STORE instead: dw = sum, one pass over dw spared.  Zero / NULL: dw += sum (gradient
   * accumulation over several backward passes).  A device flag, so that a captured step and an eager accumulating pass can share one recorded plan.  Kernels that
   * add with atomics or replicas ignore it; the result is the same either way. */
  const int32_t* overwrite_dev;
} rua_wgrad_desc;
typedef struct rua_wgrad_pending {
  int32_t kind;                /* 0: nothing to reduce (single writer / wgrad_pw), 1: all-taps partials, 2: K-slice slabs,
                                  3 (built by the caller): partials = replicated fp64 statistics [parts][2][n], dw[c] += sum of slot 0
                                  (what rua_stats_to_f32 does: bias gradients), blocks = ceil(n / 256) */
  int32_t parts;               /* partial buffers to sum (blocks of the all-taps kernel per output-channel half / K slices) */
  int64_t n;                   /* elements of dW */
  const float* partials;
  float* dw;
  int32_t CC;                  /* kind 1: channels */
  int32_t blocks;              /* 256-thread blocks the reduction of this record takes */
  int32_t block_begin, pad;    /* filled by the caller: first block of this record in the batched launch (prefix sum of `blocks`) */
  const int32_t* overwrite_dev; /* as rua_wgrad_desc.overwrite_dev (copied from the descriptor by rua_wgrad_plan) */
} rua_wgrad_pending;
/* the record a deferred rua_conv_wgrad(d) leaves behind - computed without launching anything */
int rua_wgrad_plan(const rua_wgrad_desc* d, rua_wgrad_pending* out);
/* dw += partial sums for every record, in ONE launch, each in its fixed order (bit-reproducible): items in device memory, sorted by
 * block_begin, total_blocks = sum of their `blocks` */
int rua_wgrad_reduce_batch(const rua_wgrad_pending* items_dev, int n_items, int total_blocks, void* stream);
int rua_conv_wgrad(const rua_wgrad_desc* d, void* stream);
/* n (<= RUA_MAX_WGRAD_GROUP) INDEPENDENT weight gradients - the dilation branches of a ResBlock (model2.py:26-31), first and second convolutions - with the results of n
 * rua_conv_wgrad calls; members on the same kernel run as ONE grid.  Members must not share dw or partial-sum workspace (a group
 * that does runs member by member). */
int rua_conv_wgrad_group(const rua_wgrad_desc* d, int n, void* stream);
int rua_wgrad_group_last_grids(void);            /* grids the calling thread's latest rua_conv_wgrad_group issued */
int64_t rua_wgrad_workspace_bytes(const rua_wgrad_desc* d);
int rua_wgrad_kind(const rua_wgrad_desc* d);   /* 0: generic tiled kernel, 1: all-taps kernel + deterministic partial reduce,
                                                  2: wgrad_dmap (wide levels), 3: wgrad_pw (narrow 1x1) */
int rua_wgrad_img_kind(const rua_wgrad_desc* d);   /* within kind 0: the whole-image kernels of the 8 x 8 / 16 x 16 levels (3x3, dilation 1, bf16, channels % 64 == 0 - % 32 for wgrad_imgs -,
                                                      N H W % 512 == 0): 1 wgrad_img (64 x 64 tiles of dW, 512-pixel chunks as K slices), 2 wgrad_imgs (more than 512 pixels and at
                                                      least half as many 32 x 32 tiles as CUs: the chunks streamed by one block, no K slices), 0 the generic tiles */

/* Master fp32 weights [taps][Cout][C] -> activation-dtype copies: forward layout (same) and
 * data-gradient layout [taps reversed][C][Cout].  One launch for the whole parameter table. */
typedef struct rua_wprep_item { int64_t src_off, dst_off; int32_t taps, Cout, C, pad; } rua_wprep_item;
int rua_weight_prep(const float* master, void* w_fwd, void* w_dgrad, const rua_wprep_item* items_dev,
                    int n_items, int max_elems, int dtype, void* stream);
/* The data-gradient layout alone from the forward-layout bf16 copy (dtype must be RUA_BF16; items as for rua_weight_prep, dst_off = the offset in both copies):
 * used when the optimizer has just written that copy itself (rua_adam_step_w / rua_sgd_step_w with wcopy_bf16 = the forward copy, whose index space
 * must then be the master's: dst_off == src_off). */
int rua_weight_prep_dgrad(const void* w_fwd, void* w_dgrad, const rua_wprep_item* items_dev, int n_items, int max_elems, const int32_t* blockmap_dev,
                          int n_blocks, int dtype, void* stream);
/* blockmap_dev (optional, NULL: a (<= 256, n_items) grid like rua_weight_prep's): int32 [n_blocks][2] = (item, first tile) - item i contributes
 * rua_wprep_blocks(taps, Cout, C) consecutive entries whose first tiles are 0, 8, 16, ...: the grid is as long as the tensors are large. */
int rua_wprep_blocks(int taps, int Cout, int C);

/* ---- few-channel 1x1 convolutions (VALU; the stem and the heads) -------------------------
 * stem: KL.Conv2D(32,(1,1)) on the 3/6/7-band input (model2.py:101).  x fp32 [M][Cin<=16]. */
int rua_stem_fwd(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, void* stream);
/* rua_stem_fwd with the per-channel statistics of its output in the epilogue: stats [replicas][2][Cout] fp64 (sum, sum of squares of y as stored;
 * zeroed by the caller) - what rua_col_stats would compute from y in a pass of its own (the first BatchNorm of the encoder, model2.py:102-103). */
int rua_stem_fwd_stats(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, double* stats, int replicas, void* stream);
int rua_stem_bwd(const float* x, const void* dy, float* dw, float* db, int64_t M, int Cin, int Cout, int dtype, void* stream);
/* rua_stem_fwd(_stats) that also writes the input once more as bf16 xpack [M][16] = { hi(x_0..x_7) | lo(x_0..x_6), 1 } (hi = bf16(x), lo = bf16(x - hi); Cin <= 7;
 * stats may be NULL): with it the stem's weight gradient is a 1x1 weight gradient on the matrix pipe - rua_conv_wgrad(a = xpack, C = 16, dy, dw = tmp [Cout][16],
 * tmp zero) - followed by rua_stem_bwd_fold: dW[co][c] += tmp[co][c] + tmp[co][8 + c], db[co] += tmp[co][15] (NULL: not), tmp := 0.  The vector form
 * (rua_stem_bwd) keeps 72 partial sums per thread and ended every block in their butterflies and 288 same-address atomics: 38 us for a 46 MB pass. */
int rua_stem_fwd_pack(const float* x, const float* w, const float* b, void* y, int64_t M, int Cin, int Cout, int dtype, double* stats, int replicas,
                      void* xpack, void* stream);
int rua_stem_bwd_fold(float* tmp, float* dw, float* db, int Cin, int Cout, void* stream);
/* heads: Conv2D(num_classes,(1,1)) + softmax / sigmoid (model2.py:145-146,160-162,169-171,181-183,186-188).
 * act: 0 none, 1 softmax over channels, 2 sigmoid.  z (logits) and p are fp32 [M][Cout<=8]. */
int rua_head_fwd(const void* x, const float* w, const float* b, float* z, float* p, int64_t M, int Cin, int Cout, int act, int dtype, void* stream);
/* rua_head_fwd with the loss moments of its output in the epilogue (the probabilities are in registers): tanimoto_sums
 * [B][Cout][6] fp64 as rua_tanimoto_sums accumulates them against the labels y [B*HW][Cout] (NULL: not), metrics[5] fp64 as
 * rua_seg_metrics (NULL: not); both zeroed by the caller.  Replaces those passes over p (train_ISPRS.py:456-461). */
int rua_head_fwd_loss(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* tanimoto_sums,
                      double* metrics, int B, int64_t HW, int Cin, int Cout, int act, int dtype, void* stream);
/* the same with tanimoto_sums kept in sums_replicas (1..64) copies, [sums_replicas][B][Cout][6]: a block adds into ONE of them, so ~64 blocks of a sample
 * do not queue up on the same 36 addresses at the end of the launch; rua_tanimoto_finalize_rep adds the copies. */
int rua_head_fwd_loss_rep(const void* x, const float* w, const float* b, float* z, float* p, const float* y, double* tanimoto_sums,
                          int sums_replicas, double* metrics, int B, int64_t HW, int Cin, int Cout, int act, int dtype, void* stream);
/* scratch (optional, >= 1024*(Cout*Cin+Cout)*4 bytes): per-block partials + fixed-order reduce instead of fp32 atomics.
 * mask_dx: x is the output of a fused ReLU (the heads' 3x3 conv + relu, model2.py:153-158): dx *= (x > 0), i.e. the ReLU's
 * backward is applied here instead of in a pass of its own */
int rua_head_bwd(const void* x, const float* dz, const float* w, void* dx, int accumulate_dx, float* dw, float* db,
                 float* scratch, int64_t scratch_bytes, int64_t M, int Cin, int Cout, int dtype, int mask_dx, void* stream);
/* the same with dxsum (fp32 [Cin], +=, optional): the per-channel sums of the values written to dx - the bias gradient of the convolution that produced x
 * where dx is its output's complete gradient (the 3x3 + ReLU convs in front of the heads: model2.py:153-171) - taken in the same pass */
int rua_head_bwd_sums(const void* x, const float* dz, const float* w, void* dx, int accumulate_dx, float* dw, float* db, float* dxsum,
                      float* scratch, int64_t scratch_bytes, int64_t M, int Cin, int Cout, int dtype, int mask_dx, void* stream);

/* ---- BatchNormalization (model2.py:17,21,38,86,93; Keras eps 1e-3, momentum .99) -------- */
/* per-channel sum / sum of squares over all rows of x [M][C] -> stats[R][2][C] (fp64, accumulated over R replicas) */
int rua_col_stats(const void* x, int64_t M, int C, double* stats, int replicas, int dtype, void* stream);
/* replicas to use for a statistics buffer fed by a kernel of `blocks` workgroups */
int rua_stats_replicas(int64_t blocks);
/* sum g*m and sum g*m*x with m = (mscale*x+mshift > 0) when masked, else 1 */
int rua_col_stats2(const void* g, const void* x, const float* mscale, const float* mshift, int masked,
                   int64_t M, int C, double* stats, int replicas, int dtype, void* stream);
/* training: stats -> scale/shift (+ mean, rstd, moving-stat update with the Bessel factor
 * bessel_n/(bessel_n-1), bessel_n = element count Keras sees, i.e. after nearest upsampling);
 * inference: moving stats -> scale/shift */
int rua_bn_finalize(const double* stats, int replicas, double count, double bessel_n, const float* gamma, const float* beta,
                    float* moving_mean, float* moving_var, float momentum, float eps, int training,
                    float* scale, float* shift, float* mean, float* rstd, int C, void* stream);
/* dst_i[c] += (float)stats[c] for n (<=4) fp32 vectors: bias gradients from per-channel sums of dy */
int rua_stats_to_f32(const double* stats, int replicas, int C, float* const* dst, int n, void* stream);
/* out_b = [relu](scale_b * x + shift_b) for b < nb (all branches of a ResBlock read x once) */
int rua_bn_apply(const void* x, int nb, const float* const* scale, const float* const* shift, int relu,
                 void* const* out, int64_t M, int C, int dtype, void* stream);
/* backward statistics -> dgamma += , dbeta += , and the coefficients A,B,Cc of dx = A*g + B*x + Cc */
int rua_bn_bwd_finalize(const double* stats2, int replicas, double count, const float* gamma, const float* mean, const float* rstd,
                        float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC, int C, void* stream);
/* dx (=|+=) [dskip] + sum_b (A_b * g_b * m_b + B_b * x + C_b),  m_b = ReLU mask of branch b (or 1) */
int rua_bn_bwd_apply(int nb, const void* const* g, const float* const* coefA, const float* const* coefB,
                     const float* const* coefC, const float* const* mscale, const float* const* mshift, int masked,
                     const void* x, const void* dskip, void* dx, int accumulate, int64_t M, int C, int dtype, void* stream);

/* Fused forms (one launch each; the engine uses these): statistics -> coefficients in every block's prologue, block 0
 * publishes scale/shift/mean/rstd, updates the moving statistics (forward) or adds dgamma/dbeta (backward). */
typedef struct rua_bn_branch {
  const float* gamma; const float* beta; float* moving_mean; float* moving_var;   /* [C] */
  float* scale; float* shift; float* mean; float* rstd;                           /* [C] published coefficients */
  void* out;                                                                      /* [M][C]; NULL in EVERY branch: coefficients only
                                                                                     (one block, nothing applied: the consumer
                                                                                     normalises on load, rua_conv_desc.in_scale) */
  const double* stats; int32_t replicas, pad;                                     /* this branch's own statistics (else the shared ones) */
  double* out_stats;   /* optional [2][C] (training, relu == 0 only): per-channel sum / sum of squares of THIS BatchNorm's output, written by block 0 from the
                          coefficients - sum = count * beta, sum of squares = count * (beta^2 + gamma^2 var / (var + eps)) exactly, so a BatchNorm that
                          follows (the first BatchNorms of the next ResBlock: model2.py:17 behind model2.py:86) needs no statistics pass over the tensor */
} rua_bn_branch;
typedef struct rua_bn_fwd_desc {
  const void* x; int64_t M; int32_t C, dtype, nb, relu, training, replicas;
  const double* stats;                       /* [replicas][2][C], shared by all branches (they normalise the same x) */
  double count, bessel_n; float momentum, eps;
  rua_bn_branch br[RUA_MAX_BRANCH];
} rua_bn_fwd_desc;
int rua_bn_fwd(const rua_bn_fwd_desc* d, void* stream);
/* n (<= RUA_MAX_BRANCH) independent one-branch BatchNorm applications of equal channel count (any pixel counts) as ONE grid (the results of n rua_bn_fwd calls; members that
 * cannot share a grid are launched one by one).  Tuning key bn_bwd_group switches both this and rua_bn_bwd_group. */
int rua_bn_fwd_group(const rua_bn_fwd_desc* d, int n, void* stream);
int rua_bn_fwd_group_last_grids(void);
typedef struct rua_bn_bwd_branch {
  const void* g; const double* stats2; int32_t replicas;
  int32_t stats2_out;   /* 1: slot 1 of stats2 is sum g * OUT (this BatchNorm's output, no ReLU: what rua_bn_bwd_desc.dx_stats of the launch that wrote g
                           provides) instead of sum g * x; the kernel converts with x = (out - shift) / scale (scale, shift must be set) */
  const float* gamma; const float* mean; const float* rstd; const float* scale; const float* shift;
  float* dgamma; float* dbeta;
} rua_bn_bwd_branch;
typedef struct rua_bn_bwd_desc {
  const void* x; const void* dskip; void* dx; int64_t M; int32_t C, dtype, nb, masked, accumulate, pad; double count;
  rua_bn_bwd_branch br[RUA_MAX_BRANCH];
  double* skip_stats;          /* optional [skip_replicas][2][C]: per-channel sum of dskip added into slot 0 (the bias gradient of */
  int32_t skip_replicas, pad2; /* the convs whose output the skip tensor is the gradient of: model2.py:27-31) - saves a pass over dskip */
  double* dx_stats;            /* optional [dx_replicas][2][C]: per-channel sum of the values written to dx added into slot 0 - the bias gradient of the */
  int32_t dx_replicas, pad3;   /* convolution that produced x (the stride-2 1x1 convs in front of the encoder ResBlocks: model2.py:103-111), without a pass over dx;
                                  slot 1 += sum dx * x: with stats2_out the statistics of the BatchNorm backward of the BatchNorm that produced x (model2.py:86) */
} rua_bn_bwd_desc;
int rua_bn_bwd(const rua_bn_bwd_desc* d, void* stream);
/* n (<= RUA_MAX_BRANCH) independent one-branch BatchNorm backwards of equal channel count (any pixel counts) - the second BatchNorms of a
 * ResBlock's dilation branches (model2.py:21-22), the branch BatchNorms of a PSPPooling (model2.py:56-66), each with its own gradient, input and output - as ONE grid (the results of n rua_bn_bwd calls; members
 * that cannot share a grid - several branches, skip statistics, unequal shapes - are launched one by one).  ..._last_grids: 1 or n. */
int rua_bn_bwd_group(const rua_bn_bwd_desc* d, int n, void* stream);
int rua_bn_bwd_group_last_grids(void);

/* ---- pooling / resampling (PSPPooling model2.py:47-60; decoder model2.py:91) ---------- */
int rua_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int k, int dtype, void* stream);
int rua_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int accumulate, int N, int H, int W, int C, int k, int dtype, void* stream);
/* y[n,h,w,c] = sum over the k x k block (adjoint of nearest upsampling) */
int rua_sumpool(const void* x, void* y, int N, int H, int W, int C, int k, int dtype, void* stream);
/* The 2 / 4 / 8 pyramid of one PSPPooling: the max-pool with window 2k from the max-pool with window k and its argmax bytes
 * (values and argmax bytes of a direct rua_maxpool_fwd with 2k; y_half is [N][H_half][W_half][C]: x is read once, by the k = 2
 * pass), the scatter of up to three pooled gradients into dx (one read-modify-write of dx), and the three window sums in one pass
 * (fp32 partial sums carried up the pyramid). */
int rua_maxpool_derive(const void* y_half, const uint8_t* idx_half, void* y, uint8_t* idx, int N, int H_half, int W_half, int C,
                       int k_half, int dtype, void* stream);
int rua_maxpool_bwd_multi(int n, const void* const* dy, const uint8_t* const* idx, const int* ks, void* dx, int accumulate,
                          int N, int H, int W, int C, int dtype, void* stream);
int rua_sumpool_pyramid(const void* x, void* y2, void* y4, void* y8, int N, int H, int W, int C, int dtype, void* stream);
int rua_upsample_nearest(const void* x, void* y, int N, int H, int W, int C, int k, int dtype, void* stream);

/* ---- elementwise ------------------------------------------------------------------------- */
int rua_add_n(int n, const void* const* in, void* out, int accumulate, int64_t elems, int dtype, void* stream);
int rua_relu_mask(void* dy, const void* y, int64_t elems, int dtype, void* stream);   /* dy *= (y>0) */
int rua_relu(const void* x, void* y, int64_t elems, int dtype, void* stream);
int rua_cast_f32_to(const float* x, void* y, int64_t elems, int dtype, void* stream);
int rua_cast_to_f32(const void* x, float* y, int64_t elems, int dtype, void* stream);
int rua_fill_zero(void* p, int64_t bytes, void* stream);

/* ---- losses and metrics (multitasking_utils.py:38-85, utils.py:466-491, train_ISPRS.py:411-461)
 * p, y, z, dz are fp32 [B][HW][C]. */
#define RUA_LOSS_TANIMOTO 0
#define RUA_LOSS_WCE 1
#define RUA_LOSS_CE_LOGITS 2
#define RUA_LOSS_BCE_LOGITS 3
#define RUA_LOSS_MSE 4
#define RUA_ACT_NONE 0
#define RUA_ACT_SOFTMAX 1
#define RUA_ACT_SIGMOID 2
/* six moments per (sample, class): sum p, sum (1-l), sum p*l, sum p^2+l^2, sum (1-p)(1-l), sum (1-p)^2+(1-l)^2 */
int rua_tanimoto_sums(const float* p, const float* y, int B, int64_t HW, int C, double* sums, void* stream);
/* loss_out[0] = mean_n Tanimoto_dual ; coef[B][C][3] (optional): dLoss/dp = c0 + c1*p + c2*l (already * grad_scale);
 * per_sample[B] (optional): the (B,) vector the reference's loss function returns (multitasking_utils.py:84) */
int rua_tanimoto_finalize(const double* sums, int B, int64_t HW, int C, float grad_scale, double* loss_out, float* coef,
                          float* per_sample, void* stream);
#define RUA_MAX_HEADS 8
typedef struct rua_tani_head {
  double* sums; int32_t replicas, B, C; float grad_scale; double* loss_out; float* coef; float* per_sample;   /* as rua_tanimoto_finalize_rep */
} rua_tani_head;
/* rua_tanimoto_finalize_rep for n (<= RUA_MAX_HEADS) heads in ONE launch (the seg / bound / dist / color heads of the multitask model, train_ISPRS.py:417-421) */
int rua_tanimoto_finalize_multi(const rua_tani_head* heads, int n, void* stream);
typedef struct rua_dz_head {
  int32_t kind, act; const float* p; const float* y; const float* coef; const float* class_w; float grad_scale; int32_t B; int64_t HW; int32_t C, pad; float* dz;
} rua_dz_head;
/* rua_head_dz for n (<= RUA_MAX_HEADS) heads in ONE launch */
int rua_head_dz_multi(const rua_dz_head* heads, int n, void* stream);
/* sums [replicas + 1][B][C][6]: `replicas` copies filled by rua_head_fwd_loss_rep; their sum is stored into the extra slot behind them (plain stores:
 * idempotent) and finalised as above. */
int rua_tanimoto_finalize_rep(double* sums, int replicas, int B, int64_t HW, int C, float grad_scale, double* loss_out, float* coef,
                              float* per_sample, void* stream);
/* Tanimoto_loss(label, pred) itself (multitasking_utils.py:38-68), shape (B,): take the sums with p := label, y := pred
 * (rua_tanimoto_sums(label, pred, ...)), then per_sample[n] = (sum_c w_c*Spl + 1e-5) / (sum_c w_c*(Ssq - Spl) + 1e-5) with
 * w_c = 1 / (mean_n volume of the FIRST argument)^2, inf -> largest finite weight (:46-53); mean_out[0] = mean_n. */
int rua_tanimoto_ratio(const double* sums, int B, int C, double* mean_out, float* per_sample, void* stream);
/* loss_out[0] += sum over pixels of the per-pixel loss of kind 1..4 (caller divides);
 * per_pixel[M] (optional): the (B,H,W) map the reference's loss function returns (utils.py:486) */
int rua_pixel_loss(int kind, const float* p, const float* z, const float* y, const float* class_w,
                   int64_t M, int C, double* loss_out, float* per_pixel, void* stream);
/* dz = d(total)/d(logits) for every loss kind; coef only for Tanimoto; grad_scale = loss_weight/denominator */
int rua_head_dz(int kind, int act, const float* p, const float* y, const float* coef, const float* class_w,
                float grad_scale, int B, int64_t HW, int C, float* dz, void* stream);
/* out[5] += {#argmax matches, TP, FP, TN, FN at threshold .5} */
int rua_seg_metrics(const float* p, const float* y, int64_t M, int C, double* out, void* stream);

/* ---- optimizers on the flat parameter buffer (train_ISPRS.py:404-407) --------------------- */
/* Keras Adam: theta -= lr_t * m / (sqrt(v) + eps); g is read as g*grad_scale and zeroed if zero_grad.
 * lr_t_dev (optional): device scalar overriding lr_t, so a captured HIP graph can be replayed every step */
/* state[0] += 1 (optimizer steps taken), lr_out[0] = state[1] (base rate) [* sqrt(1 - beta2^t) / (1 - beta1^t) for Adam]:
 * the step-dependent rate is produced on the device so that a captured step replays without host-side scalar updates */
int rua_lr_step(double* state, float* lr_out, int adam, double beta1, double beta2, void* stream);
int rua_adam_step(float* theta, float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                  float eps, float grad_scale, int zero_grad, void* stream);
int rua_sgd_step(float* theta, float* g, float* vel, int64_t n, float lr, const float* lr_dev, float momentum, float grad_scale,
                 int zero_grad, void* stream);
/* rua_adam_step / rua_sgd_step that also store the updated parameters as bf16 into wcopy_bf16[i] (NULL: not) - the convolutions' forward-layout weight copy */
int rua_adam_step_w(float* theta, float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                    float eps, float grad_scale, int zero_grad, void* wcopy_bf16, void* stream);
int rua_sgd_step_w(float* theta, float* g, float* vel, int64_t n, float lr, const float* lr_dev, float momentum, float grad_scale,
                   int zero_grad, void* wcopy_bf16, void* stream);

/* ---- data parallel (train_ISPRS.py:347,432: the implicit NCCL all-reduce of tf.distribute.MirroredStrategy).  The library exports
 * no collective: gradients live in ONE flat fp32 buffer in parameter order, so the all-reduce is ncclAllReduce (RCCL) on contiguous
 * slices of it, issued by the host as the backward completes them (the Python engine: torch.distributed, dist.py; a C embedder:
 * INTEGRATION.md section 2).  rua_adam_step / rua_sgd_step take grad_scale = 1 / replicas. */

/* kernel nodes / all nodes of a captured hipGraph_t (diagnostics: dispatches per whole-step graph) */
int rua_graph_kernel_nodes(void* graph, int* kernels, int* total);

/* ---- tuning switches (experiments, A/B runs).  The launchers never read the environment and keep no other global
 * state: a heuristic changes only through this call.  Keys: rua_tuning_key(0..) until NULL.  Grid-size keys
 * ("*_blocks", "*_target", "*_grid") default to 0 = derived from the device's compute-unit count. */
int rua_set_tuning(const char* key, int64_t value);
int rua_get_tuning(const char* key, int64_t* value);
const char* rua_tuning_key(int index);

#ifdef __cplusplus
}
#endif
#endif

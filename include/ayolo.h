/*
 * ayolo.h -- C ABI of libayolo_hip.so: the MI355X (gfx950) hot path of AYolov2's detection pipeline.
 *
 * The reference (j-marple-dev/AYolov2) has NO FFI/plugin interface for this path: its operator boundary is
 * Python objects whose arithmetic is delegated to torch/cuDNN, torchvision and the
 * un-vendored `kindle` model builder.  Every entry point below therefore cites the reference call site(s)
 * whose delegated op it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - plain C, no torch types; every tensor is a raw DEVICE pointer + explicit dims; the caller owns every
 *     buffer (including workspaces whose size is queried first);
 *   - every function takes a hipStream_t (passed as void*) and is asynchronous on it;
 *   - returns 0 on success, a negative AYOLO_E* code otherwise; ayolo_last_error() gives the message
 *     (thread-local);
 *   - activations are NHWC ("channels last"), dtype AYOLO_F16 (fp16 storage, fp32 MFMA accumulate) or
 *     AYOLO_F32 (exact fp32 MFMA, used for the 1e-4 parity mode);
 *   - conv weights are KRSC = [Cout][kh][kw][Cin] (a torch OIHW tensor in channels_last memory format).
 */
#ifndef AYOLO_H
#define AYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AYOLO_OK 0
#define AYOLO_EINVAL (-1)   /* bad argument / unsupported shape */
#define AYOLO_ELAUNCH (-2)  /* HIP launch or runtime error      */
#define AYOLO_ENOSPC (-3)   /* caller-supplied buffer too small */

#define AYOLO_F16 0
#define AYOLO_F32 1

typedef void* ayolo_stream; /* hipStream_t */

int ayolo_version(void);
const char* ayolo_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution (kindle `Conv.forward` / `YOLOHead.conv[i]`: yolov5s.yaml:21-57; autograd backward of the
 * same: scripts/train/yolo_trainer.py:329).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ayolo_conv_desc {
    int dtype;            /* AYOLO_F16 | AYOLO_F32 (x, w, y element type)                            */
    int B, H, W;          /* input spatial dims                                                      */
    int Cin, ldx;         /* channels read, channel stride (elements) of the x buffer (>= Cin)       */
    int Cout, ldy;        /* channels written, channel stride of the y buffer: both multiples of 8   *
                           * for fp16 (tiles leave in 16-byte stores), of 4 for fp32 (EPI_HEAD: any Cout) */
    int kh, kw, sh, sw, ph, pw;
    int Ho, Wo;           /* output spatial dims                                                     */
} ayolo_conv_desc;

/* epilogue selectors for ayolo_conv_fwd */
#define AYOLO_EPI_NONE 0      /* y = conv                        (+ optional per-channel sum/sumsq)   */
#define AYOLO_EPI_AFFINE 1    /* y = conv*scale[c] + shift[c]                                          */
#define AYOLO_EPI_AFFINE_SILU 2 /* y = silu(conv*scale[c] + shift[c])  (eval / fused-BN inference)     */
#define AYOLO_EPI_HEAD 3      /* y(fp32)[pixel][ldy] = conv + shift[c] (YOLOHead logits; ldy = Cout rounded up
                               * to 8; the (B,na,ny,nx,no) tensor is a strided view: c = a*no + o)           */
#define AYOLO_EPI_AFFINE_RES 4      /* y += conv*scale[c] + shift[c]        (in-place residual: Bottleneck shortcut)  */
#define AYOLO_EPI_AFFINE_SILU_RES 5 /* y += silu(conv*scale[c] + shift[c])  (inference: `x + cv2(cv1(x))` over x)     */

/* y = conv(x, w).  `stats` (nullable, EPI_NONE only): double[stat_reps][2*Cout] zero-initialised by the caller;
 * receives per-channel sum and sum of squares of the fp32 accumulators rounded to the output dtype (training-mode
 * BN), spread over stat_reps replicas (workgroup b adds into replica b % stat_reps) to avoid serialising L2
 * atomics; ayolo_bn_finalize sums the replicas.  Every BatchNorm accumulator of this library (`stats` here, `sums` of
 * the backward passes) is DOUBLE from the workgroup level on: the order in which workgroups add is not defined, and only
 * at 1e-16 does that order stay below one fp32 ulp of the statistics derived from the totals -- a train step's
 * statistics, and with them the fp16 rounding of every activation, then repeat from run to run.
 * scale/shift: float[Cout] (nullable where unused).  head_no: `no` for AYOLO_EPI_HEAD (y is fp32). */
int ayolo_conv_fwd(const ayolo_conv_desc* d, const void* x, const void* w, void* y, int epilogue,
                   const float* scale, const float* shift, double* stats, int stat_reps, int head_no, ayolo_stream s);

/* dx (+)= conv_transpose(dy, w).  wt is the transposed weight [Cin][kh][kw][Cout] (see ayolo_cast_weight).
 * accumulate != 0 adds into the existing dx. d describes the FORWARD conv (x:B,H,W,Cin  y:B,Ho,Wo,Cout);
 * dy uses d->ldy, dx uses d->ldx. */
int ayolo_conv_dgrad(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                     ayolo_stream s);

/* The same dgrad with the FIRST pass of the BatchNorm + activation backward of the layer(s) that produced x folded into
 * its epilogue (autograd's native_batch_norm_backward + silu_backward of scripts/train/yolo_trainer.py:329; what
 * ayolo_bn_act_bwd_reduce computes in a pass of its own): dx is the gradient `da` of up to two Conv-BN-act blocks that
 * lie side by side in dx's channels (a concat buffer).  For block k (channels [c0, c0 + C) of dx) the kernel reads z at
 * the pixel it has just produced (z: the block's pre-BatchNorm conv output, channel stride ldz), forms
 * du = da * act'(bn(z)) from the rounded value it stores (for accumulate != 0: the total), and adds sum(du) and
 * sum(du * xhat) into sums[replica][0..C) / [C..2C) exactly as the separate pass would (replicas as for `stats` of
 * ayolo_conv_fwd; zeroed by the caller).  dx is written unchanged, so ayolo_bn_act_bwd_apply and any other reader of da
 * are unaffected.  Only valid when this call is the LAST writer of those channels of dx.
 * fp16 only; c0 and C multiples of 8 and every 32-channel block of dx inside one segment; nseg in {1, 2}. */
typedef struct ayolo_bn_seg {
    const void* z;            /* conv output the BatchNorm normalised, NHWC, same pixels as dx        */
    const float* mean_invstd; /* float[2*C]: save_mean | save_invstd of the forward pass               */
    const float* gamma;       /* float[C] or NULL (= 1)                                                */
    const float* beta;        /* float[C] or NULL (= 0)                                                */
    double* sums;             /* double[reps][2*C] accumulators                                        */
    int ldz, c0, C, reserved;
} ayolo_bn_seg;
int ayolo_conv_dgrad_bn(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                        const ayolo_bn_seg* segs, int nseg, int act, int sum_reps, ayolo_stream s);

/* Forward of a 1x1 / stride-1 conv whose input is (partly) VIRTUAL (transform on load; res/configs/model/yolov5s.yaml:21-33:
 * kindle `Conv` = conv -> BatchNorm -> SiLU, whose activation has exactly one reader here).  The conv's input channels are one or
 * two SEGMENTS side by side (two: C3's cv3 over [last Bottleneck output | cv2 half]); a `virt` segment is the producing
 * block's pre-activation z and its operand act(z[p][c] * xscale[c] + xshift[c]) is formed on the way to the MFMAs, with the
 * arithmetic of ayolo_bn_train_act / ayolo_affine_act (fma, sigmoid by v_exp / v_rcp, one fp16 rounding) -- the result equals
 * the conv over the materialised activation bit for bit and that pass is never launched; a plain segment is an activation as
 * it lies.  xscale / xshift: float[Cin] over the conv's input channels (ignored for plain segments).  fp16; the first of two
 * segments must end on a multiple of 32 channels; epilogue AYOLO_EPI_NONE (+ `stats`) or AYOLO_EPI_HEAD (+ `shift` = bias). */
typedef struct ayolo_xf_seg {
    const void* x;            /* first channel of the segment at pixel 0 */
    int ld, C;                /* channel stride of its buffer, channels */
    int act, virt;            /* SiLU on / off; 1: transform (x is z), 0: plain activation */
} ayolo_xf_seg;
/* nfin > 0: the BatchNorm FINALIZE of the virtual segments happens inside this launch too (what ayolo_bn_finalize_ld computes, from
 * the producing conv's batch statistics): every workgroup derives the scale / shift of channels [c0, c0 + C) of the conv's input,
 * workgroup 0 writes them to xscale / xshift (the weight gradient reads them later) and updates the saved / running statistics
 * -- a block whose activation is read on load then costs no launch of its own at all.  nfin == 0: xscale / xshift are inputs. */
typedef struct ayolo_xf_fin {
    const double* stats;      /* [reps][2][sld] accumulators of the producing conv, this block's channels first */
    int reps, sld, C, c0;     /* replicas, accumulator channel stride, channels, first input channel of the consumer */
    double count;             /* pixels per channel */
    const float* gamma; const float* beta;
    float eps, momentum;
    float* running_mean; float* running_var; float* save_mean; float* save_invstd;   /* each nullable */
} ayolo_xf_fin;
/* xa != NULL (store-back): the workgroups of the first output-channel tile also WRITE the activation they form to xa[p][c]
 * (channel stride ldxa, the conv's input channels side by side; plain segments are not touched) -- exactly the tensor the
 * BatchNorm + activation pass would have written, for later readers that want it materialised (the conv's weight gradient:
 * transforming on load there too costs more on the weight-gradient stream than the write costs here, profiles/r04_ab_xf_*). */
int ayolo_conv_fwd_xf(const ayolo_conv_desc* d, const ayolo_xf_seg* segs, int nseg, float* xscale, float* xshift,
                      const ayolo_xf_fin* fin, int nfin, void* xa, int ldxa, const void* w, void* y, int epilogue, const float* shift,
                      double* stats, int stat_reps, int head_no, ayolo_stream s);

/* Weight gradient (autograd's ConvolutionBackward weight leg behind scripts/train/yolo_trainer.py:329):
 * dw[Cout][kh][kw][Cin] (fp32) += alpha * sum_pixels dy (x) x.  The pixel reduction is split over workgroups; every split
 * STORES its tile sums into its own slot of the caller's workspace `ws` (ayolo_conv_wgrad_workspace(d) bytes, 16-byte
 * aligned) and a second kernel adds the slots in a fixed order -- no atomics, bit-reproducible.  The packed stem needs no
 * workspace (0 bytes). */
size_t ayolo_conv_wgrad_workspace(const ayolo_conv_desc* d);
int ayolo_conv_wgrad(const ayolo_conv_desc* d, const void* x, const void* dy, float* dw, float alpha, void* ws, size_t ws_bytes,
                     ayolo_stream s);
/* The weight gradients of SEVERAL layers as one launch per tile class + one reduction (they have no mutual dependencies:
 * in the reference they are ~60 independent cuDNN wgrad launches inside loss.backward(), yolo_trainer.py:329; here a group
 * = the layers of one reverse-layer gradient bucket, scripts/train/train_model_builder.py:75-78).  Planning (pixel splits,
 * the per-XCD item queues) happens ONCE: _size reports the bytes of the group table and of the split-K workspace, _build
 * fills the table in HOST memory, the caller keeps that copy and uploads another to the device; _run launches from both.
 * dy_slot >= 0: the job's dy is dy_override[dy_slot] of the run call (YOLOHead levels: the loss hands over a different buffer
 * each step); overwrite: dw = alpha * sum instead of dw += alpha * sum; xscale / xshift / xact: transform on load.  Workspaces of groups that run on the same stream may
 * be shared.  The packed stem is not a group job (ayolo_conv_wgrad / ayolo_stem_bn_wgrad). */
typedef struct ayolo_wgrad_job {
    ayolo_conv_desc conv;     /* the forward conv's descriptor with ldy = channel stride of dy */
    const void* x; const void* dy; float* dw;
    float alpha;
    int dy_slot, overwrite;
    int xact;                 /* transform on load (1x1 / stride-1 consumers of a virtual activation, see ayolo_conv_fwd_xf): */
    const float* xscale;      /* x is the producer's pre-activation z, the operand is act(z * xscale[c] + xshift[c]);         */
    const float* xshift;      /* both NULL: plain x                                                                           */
    int dw_ld, reserved;      /* > 0: dw is a column block of a wider matrix with this row stride (the input segments of a
                               * two-segment transform-on-load conv are two jobs over one weight); 0: dense                   */
} ayolo_wgrad_job;
int ayolo_wgrad_group_size(const ayolo_wgrad_job* jobs, int njobs, size_t* table_bytes, size_t* ws_bytes);
int ayolo_wgrad_group_build(const ayolo_wgrad_job* jobs, int njobs, void* table_host, size_t table_bytes);
int ayolo_wgrad_group_run(const void* table_host, const void* table_dev, void* ws, size_t ws_bytes,
                          const void* const* dy_override, int n_override, ayolo_stream s);
/* introspection of a host table (tests / tools): out[0..5] = k_wgrad jobs (batch halves count as jobs), items of tile class 32 / 64 /
 * 128, reduction blocks, workspace floats; job >= 0: out[6..11] = its tile class, column tiles, channel tiles, pixel splits,
 * pixels per split, first workspace slot.  _item: {job or -1 (queue padding), tile, split} of item i of a tile class;
 * block i of a class's launch runs on XCD i % 8. */
int ayolo_wgrad_group_info(const void* table_host, int job, long long* out, int nout);
int ayolo_wgrad_group_item(const void* table_host, int cls, long long i, long long* out);
/* The fp16 3x3 / pad 1 / stride 1 or 2 layers (the Conv rows res/configs/model/yolov5s.yaml:22,25,28,31,46,50 and every Bottleneck's
 * 3x3 conv) run a dedicated kernel behind both entries above: the input patch of a step is staged once and all nine taps read it
 * at shifted LDS addresses (csrc/wgrad3.hip).  In a group table they are item class 3 and jobs `out[0]` .. `out[0] + out[13] - 1`
 * (nout >= 14: out[12] = items of class 3, out[13] = such jobs; job fields: tile class 0, tiles along C, tiles along N, splits,
 * virtual rows per split, first slot; nout >= 20: out[14..19] = strip width, rows per step, strips, n-blocks and c-blocks per
 * workgroup tile, LDS bytes per stage).  ayolo_wgrad3_geometry: the step geometry chosen for a layer (tests restate the kernel's
 * index algebra from it), out[0..23] = TC, RPS, PX, nsub, strips, NB, CB, NP, SL, tn, tc, nrows, ppr, rowpitch, plo, ple, xstage,
 * stage, UP, XP, NU, x_bytes, y_bytes, LDS bytes; AYOLO_EINVAL for any other layer. */
int ayolo_wgrad3_geometry(const ayolo_conv_desc* d, long long* out, int nout);
/* Backward of the STEM block (kindle Conv row 0, res/configs/model/yolov5s.yaml:21: Conv-BN-SiLU on the image) in one launch:
 * the BatchNorm + activation backward of its output gradient da and the weight gradient of its conv.  The stem has no input
 * gradient, so its dz has no other reader: the kernel forms dz = bn_act_backward(da, z; sums) on the way from HBM to LDS
 * instead of a separate ayolo_bn_act_bwd_apply pass writing it (autograd: NativeBatchNormBackward + SiluBackward +
 * ConvolutionBackward's weight leg, yolo_trainer.py:327-329).  `d`: the packed-stem descriptor (fp16, image as pixel pairs,
 * 6 x 3 taps of 8 halves, stride (2, 1), pad (2, 1), Cout <= 64) with ldy = row stride of da; z: the conv output (row stride
 * ldz); sums: [sum_reps][2][Cout] doubles as left by ayolo_bn_act_bwd_reduce / ayolo_conv_dgrad_bn; dw += alpha * gradient
 * ([Cout][6][3][8] fp32); dgamma / dbeta (may be NULL) = grad_scale * the two sums. */
int ayolo_stem_bn_wgrad(const ayolo_conv_desc* d, const void* x, const void* z, int ldz, const void* da,
                        const float* save_mean, const float* save_invstd, const float* gamma, const float* beta, int act,
                        const double* sums, int sum_reps, float* dw, float* dgamma, float* dbeta, float alpha,
                        float grad_scale, ayolo_stream s);

/* fp32 KRSC master weight [Cout][kh][kw][Cin] -> compute-dtype copy `w` [Cout_pad][kh][kw][Cin_pad] and
 * transposed copy `wt` [Cin_pad][kh][kw][Cout_pad] (either nullable); padding rows/channels are zero
 * (stem: Cin 3 -> 4, see ayolo_pack_input; head: Cout 255 -> 256). */
int ayolo_cast_weight(const float* w32, int Cout, int kh, int kw, int Cin, int Cout_pad, int Cin_pad, int dtype,
                      void* w, void* wt, ayolo_stream s);

/* ModelEMA.update (scripts/utils/torch_utils.py:405-416) for every floating tensor of a state dict in one launch:
 * ema = ema*decay; ema += (1-decay)*src.  `jobs_dev` is a DEVICE array written once by the caller. */
typedef struct ayolo_ema_job { float* ema; const float* src; int64_t n; } ayolo_ema_job;
int ayolo_ema_update(const ayolo_ema_job* jobs_dev, int njobs, float decay, ayolo_stream s);

/* Multi-tensor SGD step (torch.optim.SGD arithmetic: weight decay, momentum, dampening, nesterov) for every parameter
 * tensor of a model in ONE launch, with torch.amp.GradScaler's contract: gradients are divided by *grad_scale
 * (nullable) and the whole step is skipped when *found_inf != 0 (nullable).  Replaces the optimiser kernels behind
 * `scaler.step(optimizer)` (scripts/train/yolo_trainer.py:332-338).  `jobs_dev` is a DEVICE array; hyper-parameters
 * travel by value per parameter group (the learning rate changes every step); job.first != 0, or a NaN momentum entry,
 * means the momentum buffer is still uninitialised (buf = g, as torch clones the gradient on the first step).
 * One grid row per job: split multi-million-element tensors into jobs of ~32 K elements for parallelism. */
typedef struct ayolo_sgd_job { float* p; const float* g; float* buf; int64_t n; int group; int first; } ayolo_sgd_job;
typedef struct ayolo_sgd_group { float lr, momentum, weight_decay, dampening; int nesterov, reserved; } ayolo_sgd_group;
#define AYOLO_SGD_MAX_GROUPS 8
typedef struct ayolo_sgd_groups { ayolo_sgd_group g[AYOLO_SGD_MAX_GROUPS]; } ayolo_sgd_groups;
int ayolo_sgd_step(const ayolo_sgd_job* jobs_dev, int njobs, const ayolo_sgd_groups* groups, const float* grad_scale,
                   const float* found_inf, ayolo_stream s);

/* The same cast for every layer of a model in one launch.  `jobs_dev` is a DEVICE array (written once by the caller;
 * each entry must have Cout_pad*taps*Cin_pad < 2^32). */
typedef struct ayolo_cast_job {
    const float* w32; void* w; void* wt;      /* as ayolo_cast_weight (w / wt nullable)                           */
    int Cout, taps, Cin, Cout_pad, Cin_pad;
    int wt_ld;                                /* row stride of wt in elements (0 = Cout_pad): wt may be a column slice */
} ayolo_cast_job;
int ayolo_cast_weights(const ayolo_cast_job* jobs_dev, int njobs, int dtype, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm (training statistics) + SiLU, NHWC.  kindle Conv = Conv2d -> BatchNorm2d -> SiLU.
 * ---------------------------------------------------------------------------------------------- */
/* From stats (sum,sumsq over `count` elements per channel): mean/invstd, running-stat update
 * (momentum, unbiased var), scale = gamma*invstd, shift = beta - mean*scale.  save_mean/save_invstd/scale/
 * shift: float[C].  running_* nullable. */
int ayolo_bn_finalize(const double* stats, int stat_reps, int C, double count, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                      float* save_invstd, float* scale, float* shift, ayolo_stream s);
/* the same over a channel slice of wider accumulators ([reps][2][stat_ld], this layer's channels first at `stats`: C3's
 * cv1 | cv2 run as one conv) -- what ayolo_bn_train_act's prologue does, as a launch of its own for layers whose activation is
 * never materialised (ayolo_conv_fwd_xf) */
int ayolo_bn_finalize_ld(const double* stats, int stat_reps, int stat_ld, int C, double count, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         float* save_mean, float* save_invstd, float* scale, float* shift, ayolo_stream s);
/* ayolo_bn_finalize + ayolo_affine_act(_res) in one pass over z: a = act(batchnorm_train(z)) (+ residual), running
 * statistics updated and save_mean / save_invstd written (all four nullable) by the kernel itself.
 * stats: double[stat_reps][2][stat_ld] as accumulated by ayolo_conv_fwd, pointing at this layer's first channel;
 * stat_ld (0 = C) is the channel count of the conv that produced z when this layer is a channel slice of it. */
int ayolo_bn_train_act(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C, const double* stats,
                       int stat_reps, int stat_ld, double count, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act,
                       const void* residual, int ldr, ayolo_stream s);
/* a = act(z*scale[c] + shift[c]); act: 0 identity, 1 SiLU.  z: npix x C (ldz), a: npix x C (lda). */
int ayolo_affine_act(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C, const float* scale,
                     const float* shift, int act, ayolo_stream s);
/* Backward of a = silu(bn(z)): pass 1 accumulates sums[r][0:C] = sum(du), sums[r][C:2C] = sum(du * xhat) over
 * sum_reps replicas (double[sum_reps][2*C], zeroed by caller); pass 2 sums the replicas, writes dz and dgamma/dbeta. */
int ayolo_bn_act_bwd_reduce(int dtype, const void* z, int ldz, const void* da, int ldda, int64_t npix, int C,
                            const float* save_mean, const float* save_invstd, const float* gamma,
                            const float* beta, int act, double* sums, int sum_reps, ayolo_stream s);
int ayolo_bn_act_bwd_apply(int dtype, const void* z, int ldz, const void* da, int ldda, void* dz, int lddz,
                           int64_t npix, int C, const float* save_mean, const float* save_invstd,
                           const float* gamma, const float* beta, int act, const double* sums, int sum_reps,
                           float* dgamma, float* dbeta, float grad_scale, ayolo_stream s);
/* the same pass for a block whose output also fed a shortcut add (Bottleneck, kindle `Bottleneck.forward`: x + cv2(cv1(x));
 * autograd's AddBackward): dres[p][0:C] (+)= da[p][0:C] is written from the da values the pass reads anyway, instead of a
 * separate ayolo_copy2d over da.  dres == NULL: identical to ayolo_bn_act_bwd_apply. */
int ayolo_bn_act_bwd_apply_res(int dtype, const void* z, int ldz, const void* da, int ldda, void* dz, int lddz,
                               int64_t npix, int C, const float* save_mean, const float* save_invstd, const float* gamma,
                               const float* beta, int act, const double* sums, int sum_reps, float* dgamma, float* dbeta,
                               float grad_scale, void* dres, int lddres, int res_accumulate, ayolo_stream s);
/* the same pass for TWO blocks whose pre-activations lie side by side in one buffer (kindle C3: cv1 and cv2 read the same input,
 * res/configs/model/yolov5s.yaml:23-52; this repo runs them as ONE conv, so z / dz of the two blocks are the channel slices
 * [0, C0) and [C0, C0 + C1) of shared rows starting at `z` / `dz`): one launch over whole rows instead of two over half rows.
 * Every block keeps its own output-gradient buffer, saved statistics, affine parameters, sums and dgamma / dbeta.  Results equal
 * two ayolo_bn_act_bwd_apply calls bit for bit. */
typedef struct ayolo_bn_apply_seg {
    const void* da;           /* output gradient of the block, NHWC, channel stride ldda                 */
    const float* save_mean; const float* save_invstd;     /* float[C] each                               */
    const float* gamma;       /* float[C] or NULL (= 1)                                                  */
    const float* beta;        /* float[C] or NULL (= 0)                                                  */
    const double* sums;       /* double[sum_reps][2*C] (ayolo_bn_act_bwd_reduce / ayolo_conv_dgrad_bn)   */
    float* dgamma; float* dbeta;                          /* float[C] each or NULL                        */
    int ldda, C;
} ayolo_bn_apply_seg;
int ayolo_bn_act_bwd_apply2(int dtype, const void* z, int ldz, void* dz, int lddz, int64_t npix, const ayolo_bn_apply_seg* seg0,
                            const ayolo_bn_apply_seg* seg1, int act, int sum_reps, float grad_scale, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * Small NHWC ops: kindle SPPF's MaxPool2d(5,1,2), UpSample(None,2) nearest, input packing, bias grad.
 * ---------------------------------------------------------------------------------------------- */
/* k x k / stride 1 / pad k/2 max-pool.  argmax (nullable): uint8[B*H*W*C] window position (dy*k+dx) of the first
 * maximum in row-major scan order (torch tie rule), consumed by the backward. */
int ayolo_maxpool_fwd(int dtype, const void* x, int ldx, void* y, int ldy, unsigned char* argmax, int B, int H,
                      int W, int C, int k, ayolo_stream s);
/* dx (+)= gather of dy through argmax */
int ayolo_maxpool_bwd(int dtype, const unsigned char* argmax, const void* dy, int lddy, void* dx, int lddx, int B,
                      int H, int W, int C, int k, int accumulate, ayolo_stream s);
/* kindle SPPF's three chained MaxPool2d(5, 1, 2) (res/configs/model/yolov5s.yaml:33) in one launch per direction, fp16, on the
 * module's concat buffer `cat` = [npix][ld] with the four C-channel slices x | y1 | y2 | y3 at channel offsets 0, C, 2C, 3C:
 * forward reads slice 0 and writes slices 1..3 plus (argmax != NULL) the three window-position planes uint8[3][B*H*W*C] in
 * ayolo_maxpool_fwd's format (first maximum in row-major scan order, torch's NaN rule);  backward takes the gradient of the
 * concat buffer and leaves d(x) = d0 + pool_bwd(d1 + pool_bwd(d2 + pool_bwd(d3))) in slice 0 (intermediate sums rounded to fp16
 * as three ayolo_maxpool_bwd launches would store them; slices 1..3 are left as they were).  The map must fit a workgroup's
 * LDS: ayolo_sppf_pool_supported() > 0 (its value: the 16-byte channel groups a workgroup takes, 4 / 2 / 1 -- the width of the
 * runs it reads and writes per pixel row), else use the per-pool entry points. */
int ayolo_sppf_pool_fwd(int dtype, void* cat, int ld, unsigned char* argmax, int B, int H, int W, int C, ayolo_stream s);
int ayolo_sppf_pool_bwd(int dtype, const unsigned char* argmax, void* dcat, int ld, int B, int H, int W, int C, ayolo_stream s);
int ayolo_sppf_pool_supported(int dtype, int H, int W, int C);
int ayolo_upsample2x_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C,
                         ayolo_stream s);
int ayolo_upsample2x_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C,
                         int accumulate, ayolo_stream s);
/* NCHW fp32 image batch (B,3,H,W) -> NHWC with 4 channels (4th = 0) of `dtype`: the stem's 6x6/s2 conv then
 * runs as a 6x3 / stride (2,1) conv over the (B, H, W/2, 8) view of the same memory. */
int ayolo_pack_input(const float* x, int B, int C, int H, int W, int dtype, void* y, int Cpad, ayolo_stream s);
/* YOLOHead backward entry: d(raw) (B,na,ny,nx,no) fp32 -> NHWC dz[pix][ldz] of `dtype` (channel = a*no+o, channels
 * >= na*no zero) and dbias[c] += sum over pixels (fp32[na*no], zeroed by the caller, nullable). */
int ayolo_head_grad_pack(const float* draw, int B, int na, int ny, int nx, int no, int dtype, void* dz, int ldz,
                         float* dbias, ayolo_stream s);
/* strided 2-D copy / add (concat slices, residual):  y[p][0:C] (+)= x[p][0:C] */
int ayolo_copy2d(int dtype, const void* x, int ldx, void* y, int ldy, int64_t npix, int C, int accumulate,
                 ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * YOLOHead eval decode (layout: scripts/loss/losses.py:245-256,350; scripts/utils/tta_utils.py:52-58):
 * raw (B,na,ny,nx,no) fp32 logits (any b/a/y/x strides, o contiguous) -> out[b][row_off + (a*ny+y)*nx+x][0:no]:
 *   xy = (sig*2-0.5+grid)*stride, wh = (sig*2)^2*anchor_px, rest = sig.
 * ---------------------------------------------------------------------------------------------- */
int ayolo_head_decode(const float* raw, const int64_t* raw_strides /* host int64[4] element strides of (b,a,y,x);
                      NULL = contiguous */, int B, int na, int ny, int nx, int no, const float* anchors_px,
                      float stride, float* out, int64_t rows_total, int64_t row_off, ayolo_stream s);
/* The decode of ONE augmented forward of test-time augmentation, written where the merged prediction wants it
 * (scripts/utils/tta_utils.py:15-37 `descale_pred`, :40-59 `clip_augmented`, :62-86 `inference_with_tta`): the values of
 * ayolo_head_decode, then xywh / scale (true division), flip 2: y = flip_extent - y (flip_extent = image height),
 * flip 3: x = flip_extent - x (image width), 0: none; a row is stored only if its destination index
 * row_off + (a*ny+y)*nx+x lies in [win_lo, win_hi) -- the clipped tails are never written (row_off may be negative). */
int ayolo_head_decode_aug(const float* raw, const int64_t* raw_strides, int B, int na, int ny, int nx, int no,
                          const float* anchors_px, float stride, float* out, int64_t rows_total, int64_t row_off,
                          float scale, int flip, float flip_extent, int64_t win_lo, int64_t win_hi, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * YOLO loss, value and analytic gradient (scripts/loss/losses.py:227-300 `ComputeLoss.__call__`; box term
 * scripts/utils/metrics.py:60-135 `bbox_iou(..., c_iou=True)`; BCEWithLogitsLoss(pos_weight) for objectness / class).
 * One entry per detection level; the matched rows come from `build_targets` (losses.py:303-391).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ayolo_loss_level {
    const float* pred;         /* logits (B,na,ny,nx,no) fp32: element strides of (b,a,y,x) below, o contiguous     */
    int64_t sb, sa, sy, sx;
    int B, na, ny, nx, no;
    int n;                     /* matched target rows of this level                                               */
    const int64_t* b; const int64_t* a; const int64_t* gj; const int64_t* gi;   /* [n] cell of each row             */
    const int64_t* tcls;       /* [n] class of each row                                                          */
    const float* tbox;         /* [n][4] target box (x, y relative to the cell; w, h in grid units)              */
    const float* anch;         /* [n][2] anchor (grid units) of each row                                         */
    int* own;                  /* [B*na*ny*nx] zeroed by the caller before _fwd: 1 + index of the row that owns the
                                * cell's objectness target (the last row in order, as a sequential index_put)    */
    float* score;              /* [n] objectness target of each row: (1-gr) + gr*clamp(ciou, 0), written by _fwd  */
    float balance;             /* objectness balance of the level (losses.py:204-206)                            */
    float* grad;               /* _bwd (dense): d(out[0]) / d pred, (B,na,ny,nx,no) CONTIGUOUS fp32, fully written   */
    /* packed backward (ayolo_yolo_loss_bwd_packed): the gradient goes straight into the head conv's backward operand */
    int* head; int* next;      /* [B*na*ny*nx] zeroed by the caller before _fwd / [n]: per-cell list of its rows     */
    float* rowbox;             /* [n][4] scratch: box-logit gradients of each row                                   */
    void* dz; int ldz; int dz_dtype;   /* NHWC [B*ny*nx][ldz] of dz_dtype (channel = a*no + o, channels >= na*no zero) */
    float* dbias;              /* [na*no] fp32, zeroed by the caller: sum over pixels (nullable)                    */
} ayolo_loss_level;
/* out[5] = {loss*B, lbox*h_box, lobj*h_obj, lcls*h_cls, loss}; acc: scratch double[3*nl*64] (zeroed by the call).
 * cp / cn: smoothed positive / negative class targets; gr: IoU ratio of the objectness target. */
int ayolo_yolo_loss_fwd(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw, float gr,
                        float h_box, float h_obj, float h_cls, double* acc, float* out, ayolo_stream s);
/* grad_out: DEVICE pointer to d(objective)/d(out[0]) (e.g. the GradScaler scale) -- no host synchronisation.
 * Needs own / score as left by _fwd on the same inputs. */
int ayolo_yolo_loss_bwd(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw, float gr,
                        float h_box, float h_obj, float h_cls, const float* grad_out, ayolo_stream s);
/* Same gradient, written directly as the YOLOHead conv's backward operand (what ayolo_head_grad_pack would produce from
 * the dense gradient): lv[l].dz / dbias instead of lv[l].grad.  Needs head / next as left by _fwd. */
int ayolo_yolo_loss_bwd_packed(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw, float gr,
                               float h_box, float h_obj, float h_cls, const float* grad_out, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * NMS (scripts/utils/metrics.py:285-443 `non_max_suppression`, scripts/utils/nms.py:15-116 `batched_nms`,
 * torchvision.ops.nms / ops.boxes.batched_nms call sites metrics.py:385,394,421 nms.py:66,71,102,
 * scripts/utils/metrics.py:138-164 `box_iou`).
 * ---------------------------------------------------------------------------------------------- */
/* Stage A: stream pred (B,N,no) fp32 once; emit one candidate per (proposal, class) with
 * obj > conf (if require_obj) and obj*cls > conf [multi_label] or the best class [otherwise], optionally
 * restricted to classes whose bit is set in class_mask (uint64[(nc+63)/64], nullable).
 * rows: nullable int32 [B][rows_per_img] proposal indices to visit instead of 0..N-1 (batched_nms top-k).
 * Outputs: det[slot][6] = x1,y1,x2,y2,conf,cls ; keys[slot] = img | ~conf | seq (unique, sort ascending =
 * image, conf descending, candidate order ascending); counters[0] = total, counters[1+b] = per image.
 * Candidates beyond `capacity` are counted but not stored (caller re-runs with a larger buffer). */
int ayolo_nms_candidates(const float* pred, int B, int N, int no, float conf_thres, int multi_label,
                         int require_obj, const uint64_t* class_mask, const int32_t* rows, int rows_per_img,
                         float* det, uint64_t* keys, uint32_t* counters, uint32_t capacity, int order_by_seq,
                         ayolo_stream s);
/* layout of the candidate key: seq occupies the low seq_bits, then (unless order_by_seq) 32 bits of ~conf, then
 * the image index; returns AYOLO_EINVAL when more than 64 bits would be needed. */
int ayolo_nms_key_bits(int B, int rows_per_img, int nc_eff, int order_by_seq, int* seq_bits, int* total_bits);
int ayolo_iota_u32(uint32_t* v, uint32_t n, ayolo_stream s);
/* torchvision batched_nms coordinate trick: out[b] = max over the segment's box coordinates + 1 */
int ayolo_seg_max_coord(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, int B, float* out,
                        ayolo_stream s);
/* 64-bit key / 32-bit value radix sort (rocPRIM); ws_bytes queried with ws == NULL. */
int ayolo_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                         uint32_t* vals_out, uint32_t n, int begin_bit, int end_bit, void* ws, size_t* ws_bytes,
                         ayolo_stream s);
/* objectness keys for batched_nms' per-image top-k: keys[b*N+i] = b | ~obj | i, vals = i */
int ayolo_nms_obj_keys(const float* pred, int B, int N, int no, uint64_t* keys, uint32_t* vals, ayolo_stream s);
/* gather det rows into sorted order: sdet[t] = det[order[t]] */
int ayolo_gather_rows(const float* src, const uint32_t* order, float* dst, uint32_t n, int width, ayolo_stream s);
/* Stage C: suppression bit matrix for image segments.  sdet: sorted candidates; seg_off[B+1] (device):
 * segment b = [seg_off[b], seg_off[b]+seg_n[b]).  mask[b]: rows of `words` uint64 at mask + mask_off[b].
 * bit j of row i set iff j > i and IoU(i,j) > thr (mode 0), with boxes offset by cls*offset_scale;
 * class_aware != 0 restricts to cls_i == cls_j (per-class NMS).  thr_f is the float threshold such that
 * (ovr > thr_f) == ((double)ovr > thr_d). */
int ayolo_nms_mask(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, const uint64_t* mask_off,
                   int B, uint32_t max_n, float thr_f, float offset_scale, const float* per_img_offset,
                   int class_aware, uint64_t* mask, ayolo_stream s);
/* Stage D: greedy scan (one wavefront per image) + output gather.
 * out[b][k][0:6] = sdet row of the k-th kept candidate (k < max_out), out_idx[b][k] = its index in the
 * segment, out_count[b] = number kept (capped at max_out). */
int ayolo_nms_reduce(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, const uint64_t* mask_off,
                     const uint64_t* mask, int B, uint32_t max_out, float* out, int32_t* out_idx,
                     uint32_t* out_count, uint32_t max_n, ayolo_stream s);
/* Per-class route of the `nms` branch (metrics.py:383-388: torchvision nms on boxes + cls * 4096).  When the box
 * coordinates span less than 4096 the offset makes classes disjoint, so the greedy scan runs per (image, class)
 * segment.  class_keys: compact the selected rows of each image (segment b of sdet starts at seg_off[b]; its first
 * sel_off[b+1]-sel_off[b] rows take part) into rows1[tot][6], emit key = b*nc + cls and val = compact index for a
 * stable sort, and atomically max the order codes of max(coord) / max(-coord) into span[2] (zeroed by the caller;
 * code = bits|0x80000000 for non-negative floats, ~bits otherwise). */
int ayolo_nms_class_keys(const float* sdet, const uint32_t* seg_off, const uint32_t* sel_off, int B, int nc,
                         uint32_t tot, float* rows1, uint64_t* keys, uint32_t* vals, uint32_t* span, ayolo_stream s);
/* class_layout: from the sorted keys, segment sg = run of key == sg: seg_off2 / seg_n2 / mask_off (the arguments
 * ayolo_nms_mask and ayolo_nms_reduce take) and summary = {max segment length, total mask words}. */
int ayolo_nms_class_layout(const uint64_t* keys_sorted, uint32_t tot, int nseg, uint32_t* seg_off2, uint32_t* seg_n2,
                           uint64_t* mask_off, int64_t* summary, ayolo_stream s);
/* class_merge: kept rows of all segments (out_idx / out_count of ayolo_nms_reduce, perm = sorted vals) back into
 * per-image confidence order, first max_det per image: out[B][max_det][6], kept[B].  flags[tot] zeroed by the
 * caller, scan[tot] scratch; ws_bytes queried with ws == NULL. */
int ayolo_nms_class_merge(const float* rows1, const int32_t* out_idx, const uint32_t* out_count,
                          const uint32_t* seg_off2, const uint32_t* perm, int nseg, uint32_t max_out,
                          const uint32_t* sel_off, int B, uint32_t tot, uint32_t max_det, uint32_t* flags,
                          uint32_t* scan, float* out, uint32_t* kept, void* ws, size_t* ws_bytes, ayolo_stream s);

/* The whole class-aware `nms` branch of non_max_suppression (scripts/utils/metrics.py:313-388: confidence filter, multi-label
 * expansion, boxes offset by cls * 4096, torchvision.ops.nms, max_det) in ONE call without library sorts or intermediate host
 * reads: candidates -> per (image, class) grouping -> one workgroup per segment sorts and scans its candidates in LDS -> one
 * workgroup per image merges the kept boxes in confidence order.  pred: (B, N, no) fp32 xywh + obj + nc class scores;
 * out: (B, max_det, 6) rows [x1, y1, x2, y2, conf, cls]; status (device uint32[2 + 2B]): [0] flags -- non-zero means a limit of
 * this path did not hold (bit 0: `capacity` candidates were not enough, [1] holds the number needed; bit 1: coordinates span
 * >= 4096 so classes are not separable by the offset; bit 2: a segment above 2048 candidates, or more than 8192 keys in the
 * histogram bucket that decides the max_nms cut; bit 3: an image kept more than 8192 boxes) and the caller takes the general
 * path (ayolo_nms_candidates ... ayolo_nms_reduce).  An image with more than max_nms candidates keeps its max_nms most
 * confident ones before NMS (metrics.py:378-379), selected exactly on the device; [2 .. 2 + B) rows per image in `out`; [2 + B ..) candidates per image.
 * ws == NULL: *ws_bytes receives the workspace size for `capacity`.  iou_thres_f as for ayolo_nms_mask. */
int ayolo_nms_class_fast(const float* pred, int B, int N, int no, float conf_thres, int multi_label,
                         const uint64_t* class_mask, float iou_thres_f, uint32_t max_det, uint32_t max_nms,
                         uint32_t capacity, void* ws, size_t* ws_bytes, float* out, uint32_t* status, ayolo_stream s);
/* Fixed-shape batched NMS with the TensorRT BatchedNMS_TRT contract of the reference's engines
 * (scripts/model_converter/model_converter.py:268-388; outputs read by train_utils.py:262-283): per (image, class)
 * the topK boxes with score = obj*cls > scoreThreshold, greedy NMS with the plugin's jaccard (isNormalized = 0: +1 on
 * every extent), then the keepTopK best of an image.  All sizes are fixed by (B, N, nc, topK, keepTopK, capacity) and
 * nothing is read back by the host.  Sequence: trt_nms_candidates -> sort_pairs_u64 (capacity keys, vals = iota) ->
 * gather_rows -> trt_nms_layout -> trt_nms_mask -> nms_reduce -> trt_nms_final_keys -> sort_pairs_u64 -> trt_nms_emit.
 * key = image | class | ~score | row; unused slots hold ~0 and sort last; counters[0] > capacity means overflow
 * (candidates were dropped: rerun with a larger capacity). */
int ayolo_trt_nms_key_bits(int B, int N, int nc, int* row_bits, int* cls_bits, int* img_bits);
int ayolo_trt_nms_candidates(const float* pred, int B, int N, int no, float score_thres, int box_xyxy, float* det,
                             uint64_t* keys, uint32_t* counters, uint32_t capacity, ayolo_stream s);
int ayolo_trt_nms_layout(const uint64_t* keys_sorted, uint32_t capacity, int B, int N, int nc, uint32_t top_k,
                         uint32_t* seg_off2, uint32_t* seg_n2, uint64_t* mask_off, ayolo_stream s);
int ayolo_trt_nms_mask(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, const uint64_t* mask_off,
                       int nseg, uint32_t max_n, float iou_thres, uint64_t* mask, ayolo_stream s);
/* out == NULL: only report the key width in *total_bits */
int ayolo_trt_nms_final_keys(const float* out, const uint32_t* out_count, int B, int nc, uint32_t max_out,
                             uint64_t* fkeys, uint32_t* fvals, int* total_bits, ayolo_stream s);
/* num_det[B], boxes[B][keep][4], scores[B][keep], classes[B][keep] (padding: 0 / 0 / -1) */
int ayolo_trt_nms_emit(const uint64_t* fkeys_sorted, const uint32_t* fvals_sorted, const float* out, int B, int nc,
                       uint32_t max_out, uint32_t keep_top_k, int32_t* num_det, float* boxes, float* scores,
                       float* classes, ayolo_stream s);
/* Dense IoU (metrics.py:138-164): out[N][M]. */
int ayolo_box_iou(const float* a, int64_t N, const float* b, int64_t M, float* out, ayolo_stream s);
/* fast_nms / matrix_nms column reductions over the upper-triangular IoU of n boxes (never materialised):
 * colmax[j] = max_{i<j} iou(i,j) (0 for j = 0; NaN propagates); decay (nullable) [j] =
 * min_i exp(-(iou(i,j)^2 - colmax[i]^2)/0.5) with iou(i,j)=0 for i>=j. */
int ayolo_iou_colmax(const float* boxes, const float* cls, float offset_scale, uint32_t n, float* colmax,
                     ayolo_stream s);
int ayolo_matrix_nms_decay(const float* boxes, const float* cls, float offset_scale, uint32_t n,
                           const float* colmax, float* decay, ayolo_stream s);
/* merge_nms (metrics.py:418-435): for kept row k, weights = (iou(off_box[kept[k]], off_box[:]) > thr) * conf;
 * merged[k][0:4] = (weights @ box) / sum(weights); redundant[k] = (#iou > thr) > 1.  det: n x 6 rows. */
int ayolo_merge_boxes(const float* det, uint32_t n, float offset_scale, const int32_t* kept, uint32_t nk,
                      float thr_f32, float* merged, int32_t* redundant, ayolo_stream s);

/* a = act(z*scale + shift) + residual (residual nullable: Bottleneck shortcut fused into the activation pass) */
int ayolo_affine_act_res(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C, const float* scale,
                         const float* shift, int act, const void* residual, int ldr, ayolo_stream s);
/* inference BatchNorm folded to an affine: scale = gamma/sqrt(var+eps), shift = beta - mean*scale (+ bias*scale) */
int ayolo_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         const float* conv_bias, float eps, int C, float* scale, float* shift, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * Validator matching (scripts/utils/train_utils.py:294-333 `YoloValidator.process_batch`) for all images of a batch:
 * det [N][6] = x1,y1,x2,y2,conf,cls (each image's rows in NMS output order), det_img[N] image of each row,
 * lab [M][5] = cls,x1,y1,x2,y2 grouped by image with lab_off[B+1]; iouv_dev: niou ascending thresholds (device).
 * correct [N][niou] (uint8).  Scratch: best_l[N], best_iou[N], owner[M].  Same result as the reference's
 * IoU-sorted np.unique-by-detection then np.unique-by-label, without its device->host copies.
 * ---------------------------------------------------------------------------------------------- */
int ayolo_match_detections(const float* det, const int* det_img, int64_t N, const float* lab, const int* lab_off, int64_t M,
                           const float* iouv_dev, int niou, int* best_l, float* best_iou, int* owner,
                           unsigned char* correct, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * Result rows of the COCO json (scripts/utils/multi_queue.py:204-305 `_add_outputs` / `add_predicted_box`; un-letterbox =
 * scripts/utils/general.py:324-358 `scale_coords`): det[n][6] = [x1,y1,x2,y2,conf,cls] of a whole batch, img_of_row[n] =
 * image index of every row, letterbox[B][6] = {gain, pad_w, pad_h, w0, h0, apply (0: leave the box as it is)},
 * cat_table[ncat] class -> category id (nullable).  out[n][6] = [x, y, width, height, score, category id].
 * ---------------------------------------------------------------------------------------------- */
int ayolo_coco_rows(const float* det, const int* img_of_row, int64_t n, const float* letterbox, const int* cat_table,
                    int ncat, float* out, ayolo_stream s);

/* ------------------------------------------------------------------------------------------------
 * Batched launch: a pre-compiled straight-line program of the calls above (one model forward or backward over
 * static buffers) enqueued by ONE host call.  Field use per kind: see csrc/plan.hip.
 * ---------------------------------------------------------------------------------------------- */
#define AYOLO_OP_SIDE 0x100   /* OR-ed into kind: run on the executor's side stream (after everything enqueued so far;
                               * joined at the end of the list) */
enum {
    AYOLO_OP_NOP = 0,          /* skipped (a slot the host disabled for this run) */
    AYOLO_OP_CONV_FWD = 1, AYOLO_OP_CONV_DGRAD, AYOLO_OP_CONV_WGRAD, AYOLO_OP_CAST_WEIGHT, AYOLO_OP_BN_FINALIZE,
    AYOLO_OP_AFFINE_ACT, AYOLO_OP_BN_BWD_REDUCE, AYOLO_OP_BN_BWD_APPLY, AYOLO_OP_MAXPOOL_FWD, AYOLO_OP_MAXPOOL_BWD,
    AYOLO_OP_UPSAMPLE_FWD, AYOLO_OP_UPSAMPLE_BWD, AYOLO_OP_PACK_INPUT, AYOLO_OP_HEAD_GRAD_PACK, AYOLO_OP_COPY2D,
    AYOLO_OP_MEMSET, AYOLO_OP_BN_EVAL_AFFINE, AYOLO_OP_BN_TRAIN_ACT, AYOLO_OP_CAST_WEIGHTS, AYOLO_OP_HEAD_DECODE,
    AYOLO_OP_JOIN_SIDE,         /* the caller's stream waits for everything enqueued so far on the side stream */
    AYOLO_OP_STEM_BN_WGRAD,     /* ayolo_stem_bn_wgrad */
    AYOLO_OP_WGRAD_GROUP,       /* ayolo_wgrad_group_run: p[0] table (host), p[1] table (device), p[2] workspace, l[0] its bytes,
                                 * p[3..6] dy overrides, i[0] their count */
    AYOLO_OP_BN_BWD_APPLY2,     /* ayolo_bn_act_bwd_apply2: i[0] dtype, i[1] ldz, i[2] lddz, i[3] act, i[4] sum_reps, i[5 + k] / i[7 + k] =
                                 * C / ldda of block k; l[0] npix; f[0] grad_scale; p[0] z, p[1] dz, p[2 + 7 k ..] = da, save_mean (invstd
                                 * follows at + C), gamma, beta, sums, dgamma, dbeta of block k */
    AYOLO_OP_SPPF_FWD,          /* ayolo_sppf_pool_fwd: i[0] dtype, i[1] ld, i[2..5] B H W C; p[0] cat, p[1] argmax */
    AYOLO_OP_SPPF_BWD           /* ayolo_sppf_pool_bwd: i[] as above; p[0] argmax, p[1] dcat */
};
typedef struct ayolo_op {
    int kind;
    int i[12];
    float f[2];
    double d[1];
    int64_t l[1];
    void* p[16];
    ayolo_conv_desc conv;
} ayolo_op;
int ayolo_run_ops(const ayolo_op* ops, int n, ayolo_stream s);
/* flags: AYOLO_RUN_NO_JOIN = do not make `s` wait for the side stream at the end (the list is a segment of a longer
 * program: the next segment's ops keep overlapping with it; the LAST segment, or ayolo_side_stream_join, joins). */
#define AYOLO_RUN_NO_JOIN 1
int ayolo_run_ops_ex(const ayolo_op* ops, int n, ayolo_stream s, int flags);
/* measurement mode: the same list on the same streams with a HIP event before / after every op on the op's stream;
 * ms[k] (host float[n]) = duration of op k in situ.  Blocks until the list has finished. */
int ayolo_run_ops_timed(const ayolo_op* ops, int n, ayolo_stream s, float* ms);
/* stream `waiter` waits for everything enqueued so far on the calling thread's side stream of the current device */
int ayolo_side_stream_join(ayolo_stream waiter);
/* The executor keeps a side stream + two events per (calling host thread, device), created on first use -- no
 * process-global state, so several devices / threads can run op lists concurrently.  Frees the calling thread's. */
int ayolo_release_thread_state(void);
/* zero `bytes` bytes at a 16-byte aligned device pointer with an ordinary kernel on stream s */
int ayolo_fill_zero(void* ptr, size_t bytes, ayolo_stream s);

#ifdef __cplusplus
}
#endif
#endif /* AYOLO_H */

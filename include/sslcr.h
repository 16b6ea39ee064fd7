/* sslcr.h -- C ABI of the MI355X (gfx950) engine for the SSL_CR_Histo ResNet18 step path.
 *
 * The reference (srinidhiPY/SSL_CR_Histo) is pure Python on torch/cuDNN and has NO FFI of its own; the seam
 * this library plugs into is "the torch ops behind models/net.py + the step body of train()/validate()".
 * Each entry point below names the reference code it replaces.  Conventions:
 *   - return 0 on success, negative on error; sslcr_last_error() gives the message (thread local);
 *     no C++ exception crosses this boundary;
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise; the engine never frees it;
 *   - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream); all calls are asynchronous;
 *   - activations are NHWC, conv weights [K][R][S][C] ("KRSC"), dtype 0 = fp32 (exact-parity mode,
 *     v_mfma_f32_16x16x4_f32), 1 = bf16 storage with fp32 accumulate (v_mfma_f32_16x16x32_bf16);
 *   - one sslcr_ctx per process per device, not thread safe (single training thread, like the reference).
 */
#ifndef SSLCR_H_
#define SSLCR_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the entry points declared between this push and the pop at the end of
 * the file are exported (tests/test_abi_cpu.py compares `nm -D` with this header) */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SSLCR_F32 0
#define SSLCR_BF16 1
#define SSLCR_FP8 2   /* engine mode only (sslcr_create): bf16 storage and backward, fp8 e4m3 forward for the eligible 3x3 convs */

/* the library round (6 here).  ABI notes below name the version a behaviour changed in: 4 = sslcr_bn_bwd_reduce overwrites its sums. */
int sslcr_version(void);
const char* sslcr_last_error(void);

/* ---- conv2d forward / dgrad (replaces nn.Conv2d inside torchvision resnet18: models/net.py:32,77;
 *      BasicBlock convs K2-K4 of SURVEY 2b, and their autograd dgrad K13) */
typedef struct sslcr_conv_desc {
  const void* x;          /* gathered tensor, NHWC [N][H][W][C] */
  const void* w;          /* [K][R][S][C] */
  void* y;                /* NHWC [N][OH][OW][K] */
  const float* in_scale;  /* [C] or NULL: prologue x*scale+shift (+relu) on in-bounds pixels (producer BN fused in the load) */
  const float* in_shift;
  const float* bias;      /* [K] or NULL */
  const void* residual;   /* like y, or NULL */
  float* stats;           /* [sslcr_conv2d_partial_rows][2][K] fp32 or NULL: per-channel (sum, sumsq) partials */
  int N, H, W, C, K, R, S, stride, pad;
  int PH, PW;             /* pixel space the GEMM M dimension enumerates */
  int OH, OW, osh;        /* physical output dims; pixel (ph,pw) is stored at (ph*osh, pw*osh) */
  int transposed;         /* 0: src = p*stride-pad+r ; 1 (dgrad): src = (p+pad-r)/stride when divisible */
  int in_relu, relu, accumulate;
  int pix_mul, pix_off_h, pix_off_w;   /* sub-lattice of the pixel space: pixel (i,j) of the PH x PW grid is (i*mul+off_h, j*mul+off_w);
                                          mul = 0 means 1.  With tap_mask this runs ONE parity class of a strided dgrad */
  unsigned tap_mask;                   /* bit (r*S+s) set = visit that tap; 0 = all taps */
  /* BatchNorm-backward front end of a dgrad (optional; needs stats, excludes in_scale / bias / residual / relu; 3x3 stride 1 on
     16x16-tileable maps only -- sslcr_conv2d fails otherwise).  y is the gradient w.r.t. relu(bn(mask_x)); with these set the
     kernel writes g = y * (mask_scale[k]*mask_x + mask_shift[k] > 0) instead of y, and the stats rows hold
     (sum g, sum g*(mask_x - mask_mean[k])) -- the two sums of sslcr_bn_bwd_reduce -- instead of (sum y, sum y^2): the reduce
     pass over (dy, x) of that BatchNorm is not needed. */
  const void* mask_x; const float* mask_scale; const float* mask_shift; const float* mask_mean;
  int par4;               /* stride-2 3x3 pad-1 dgrad (transposed = 1, pix_mul = 2, PH x PW = the even-sized input / 2): all four
                             output-parity classes in ONE launch (class = grid z; offsets and tap subsets derived in the kernel)
                             instead of four launches with pix_off / tap_mask.  DMA-gather kernel shapes only. */
  /* Segments (models/net.py:50-66, TripletNet: three branches through ONE backbone, BatchNorm statistics per branch): with
     seg_images > 0 the N images are N / seg_images independent batches in one launch -- segment s reads its prologue from
     in_scale + s * seg_stride / in_shift + s * seg_stride, and the stats rows of segment s are rows
     [s * rows / nseg, (s + 1) * rows / nseg) of sslcr_conv2d_partial_rows (no workgroup's rows mix segments).  With mask_x the
     segment's BatchNorm is mask_scale / mask_shift / mask_mean + s * seg_stride.  bf16 only; sslcr_conv2d fails where the kernel
     serving the shape has no segment form (sslcr_conv2d_segments_ok). */
  int seg_images, seg_stride;
  /* library version >= 6.  [K] or NULL: y = epilogue(acc * out_scale[k] + bias[k] ...) -- eval-mode BatchNorm with its scale kept OUT
     of the filters: w is then the plain (bf16-rounded) filter, bias the BatchNorm shift (sslcr_pack_desc.scale_out).  The bf16 engine
     runs the teacher / validate() forward this way: round(w) followed by the fp32 scale leaves 3.6 x less SYSTEMATIC error in the
     logits than round(w * scale), and the consistency loss -- an average over the unlabeled batch -- keeps exactly that part
     (full-size iteration: 1.75e-3 -> 4.9e-4, tools/bf16_teacher_fold_experiment.py, profiles/r06_bf16_teacher_fold_experiment.txt).
     Only with bias (the eval forms); not with stats / mask_x. */
  const float* out_scale;
} sslcr_conv_desc;
int sslcr_conv2d(int dtype, const sslcr_conv_desc* d, void* stream);
int sslcr_conv2d_partial_rows(const sslcr_conv_desc* d);
int sslcr_conv2d_segments_ok(int dtype, const sslcr_conv_desc* d);      /* 1: this descriptor's seg_images is served */
/* name of the kernel instance sslcr_conv2d would launch for this descriptor, spelled as rocprofv3 prints it (static string;
   lets tests and profiles tie a shape to the code path that serves it) */
const char* sslcr_conv2d_kernel_name(int dtype, const sslcr_conv_desc* d);
/* A downsampling BasicBlock's two convolutions of ONE input in one launch (torchvision BasicBlock.forward: `out = self.conv1(x)` and
   `identity = self.downsample(x)` -- downsample[0] is the 1x1 / stride-2 projection -- of layer{2,3,4}[0], reached through
   models/net.py:32,77): c1 is the 3x3 / stride 2 / pad 1 descriptor, ds the 1x1 / stride 2 / pad 0 descriptor with the same x, N, H, W,
   C and K.  Both must be in the same mode: stats (train forward: raw output + BatchNorm partial rows, sslcr_conv2d_partial_rows(c1)
   rows each) or bias (eval forward with the BatchNorm folded; relu honoured per descriptor).  Served where
   sslcr_conv2d_s2_pair_ok returns 1 (bf16, 16x16-tileable OUTPUT maps, C % 64 == 0, K % 128 == 0, no prologue / residual); the
   results equal two sslcr_conv2d calls. */
int sslcr_conv2d_s2_pair(int dtype, const sslcr_conv_desc* c1, const sslcr_conv_desc* ds, void* stream);
int sslcr_conv2d_s2_pair_ok(int dtype, const sslcr_conv_desc* c1, const sslcr_conv_desc* ds);

/* ---- fp8 (OCP e4m3) forward conv path: BASELINE config 5, eval_Camelyon_SSL_CR.py:33-157 "fp8 MFMA conv path".
 *      Serves 3x3 / stride 1 / pad 1 convs with C % 128 == 0, K % 128 == 0 on 16x16-tileable maps (or 8x8 maps with N % 4 == 0):
 *      ResNet18 layers 2-4.  x / y / residual are bf16 NHWC exactly as in sslcr_conv2d(SSLCR_BF16, ...) (in_scale / in_shift /
 *      in_relu, bias, residual, relu, stats all honoured); the descriptor's `w` is ignored in favour of the e4m3 pack.
 *      Quantisation: activations x * x_scale clamped to +-448, round-to-nearest-even, on the way into LDS; weights
 *      w * w_scale[k] (power of two, amax -> (224, 448]) by sslcr_pack_conv_fp8; fp32 accumulate; result * w_dequant[k] / x_scale. */
typedef struct sslcr_fp8_desc {
  const uint8_t* w8;        /* [K][3][3][C] e4m3 */
  const float* w_dequant;   /* [K]: 1 / w_scale[k] */
  float x_scale;            /* per-tensor activation scale (> 0); 1 = activations used as they are (post-BatchNorm values sit in
                               e4m3's normal range [2^-6, 448]) */
  const float* x_scale_dev; /* optional device float: read instead of x_scale (delayed scaling without a host sync) */
  float* amax_out;          /* optional device float (>= 0 on entry): atomically raised to max |transformed activation| seen by this
                               launch, before scaling and clamping -- the engine turns it into the NEXT step's x_scale
                               (scale = 2^floor(log2(448 / (2 amax))): a factor 2 of headroom), the delayed-scaling recipe */
} sslcr_fp8_desc;
int sslcr_conv2d_fp8(const sslcr_conv_desc* d, const sslcr_fp8_desc* q, void* stream);
int sslcr_conv2d_fp8_partial_rows(const sslcr_conv_desc* d);      /* rows of d->stats the fp8 kernel writes; 0 = shape not served */
typedef struct sslcr_pack_fp8_desc {
  const float* w;           /* [K][C][3][3] fp32 (PyTorch layout) */
  uint8_t* w8;              /* [K][3][3][C] e4m3 out */
  float* w_dequant;         /* [K] out */
  const float* gamma; const float* beta; const float* rmean; const float* rvar; float eps; float* bias_out;   /* optional: fold BatchNorm (eval) */
  int K, C;
} sslcr_pack_fp8_desc;
int sslcr_pack_conv_fp8(const sslcr_pack_fp8_desc* d, void* stream);

/* ---- conv2d weight gradient (autograd wgrad of the same convs) */
typedef struct sslcr_wgrad_desc {
  const void* x;          /* conv input NHWC [N][H][W][C] (the raw producer output when in_scale != NULL) */
  const void* dy;         /* NHWC [N][OH][OW][K] */
  float* dw;              /* [K][R][S][C] fp32, ACCUMULATED into */
  const float* in_scale;
  const float* in_shift;
  int in_relu;
  int N, H, W, C, K, R, S, stride, pad, OH, OW;
  int seg_images;         /* optional (> 0, divides N; 3x3 stride-1 halo kernel shapes only): the batch is N / seg_images segments   */
  int seg_stride;         /* with their own producer BatchNorm -- segment s uses in_scale + s*seg_stride, in_shift + s*seg_stride    */
                          /* (the TripletNet branches as one batch: dW is linear in the pixels, their statistics are per branch)    */
} sslcr_wgrad_desc;
int sslcr_conv2d_wgrad(int dtype, const sslcr_wgrad_desc* d, void* stream);
/* name of the kernel instance sslcr_conv2d_wgrad would launch for this descriptor (static string, as sslcr_conv2d_kernel_name) */
const char* sslcr_conv2d_wgrad_kernel_name(int dtype, const sslcr_wgrad_desc* d);

/* diagnostics: lane l of one wave performs ds_read_b64_tr_b16 at byte_addr[l] of a 2 KiB LDS copy of `in`; out[l][0..3] */
int sslcr_probe_tr16(const uint16_t* in, const int* byte_addr, uint16_t* out, void* stream);

/* ---- stem: uint8/fp32 NCHW ingestion fused with conv1 7x7/2 (eval_BreastPathQ_SSL_CR.py:68-74 .float()/reshape
 *      + resnet18.conv1) */
typedef struct sslcr_stem_desc {
  const void* x;          /* NCHW [N][3][H][W], uint8 (in_f32=0) or fp32 (in_f32=1), values 0..255 un-normalised */
  const void* w;          /* packed [64][7][8][4] (s=7 and c=3 are zero) -- see sslcr_pack_stem */
  void* y;                /* NHWC [N][OH][OW][64] */
  const float* bias;      /* [64] or NULL */
  float* stats;           /* [sslcr_stem_partial_rows][2][64] or NULL */
  int N, H, W, OH, OW, in_f32, relu;
  const void* x2;         /* optional second input segment: images n >= n_split are read from x2[n - n_split] -- the
                             reference's torch.cat((inputs_x, inputs_u_s)) (eval_BreastPathQ_SSL_CR.py:82) without the copy */
  int n_split;
  const float* out_scale; /* [64] or NULL, as sslcr_conv_desc.out_scale (library version >= 6) */
} sslcr_stem_desc;
int sslcr_stem_conv(int dtype, const sslcr_stem_desc* d, void* stream);
int sslcr_stem_partial_rows(const sslcr_stem_desc* d);
/* eval-mode stem in one launch: conv1 (BatchNorm folded into d->w / d->bias) -> ReLU -> maxpool 3x3/2 pad 1, i.e. resnet18.conv1 ..
 * .maxpool as the teacher forward, validate() and test_Camelyon16.py reach them (models/net.py:32,77 under model.eval()).  d->y is the
 * POOLED output [N][POH][POW][64] (POH = OH / 2, POW = OW / 2); the conv output never reaches HBM.  bf16, d->relu set, d->stats NULL,
 * OH and OW multiples of 16 with 32 <= OW <= 256; other shapes fail (sslcr_stem_conv + sslcr_bn_relu_maxpool serve them, with the same
 * bits where both apply). */
int sslcr_stem_conv_pool(int dtype, const sslcr_stem_desc* d, int POH, int POW, void* stream);
typedef struct sslcr_stem_wgrad_desc {
  const void* x; const void* dy;
  float* dw;              /* [64][3][7][7] fp32 (PyTorch layout), accumulated */
  int N, H, W, OH, OW, in_f32;
  const void* x2; int n_split;   /* as in sslcr_stem_desc */
} sslcr_stem_wgrad_desc;
int sslcr_stem_wgrad(int dtype, const sslcr_stem_wgrad_desc* d, void* stream);

/* ---- BatchNorm2d (train: batch stats, running update; replaces nn.BatchNorm2d x20) */
typedef struct sslcr_bn_finalize_desc {
  const float* partials;  /* [rows][2][C] */
  int rows, C;
  double count;           /* elements per channel (global) */
  const float* gamma; const float* beta;
  float* scale; float* shift; float* mean; float* invstd;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* may be NULL */
  float momentum, eps; int replay;    /* replay=3 reproduces TripletNet_Finetune's 3 identical passes (models/net.py:88-90) */
  double* sums_out;       /* [2][C]: only reduce the partial rows (sharded runs all-reduce this, then call again with sums_in) */
  const double* sums_in;
  double* stage;          /* workspace [32][2][C] doubles for the two-stage row reduction ([nseg][32][2][C] with segments) */
  /* segments (sslcr_conv_desc.seg_images): nseg > 1 finalizes nseg BatchNorm batches in one call -- segment s owns partial rows
     [s * rows / nseg, (s + 1) * rows / nseg) and writes scale / shift / mean / invstd at + s * seg_stride floats; the running
     statistics take the nseg updates one after the other, segment 0 first, like nseg successive forward calls (count = elements
     per channel of ONE segment).  With sums_out (rows -> sums only) segment s writes sums_out + s * seg_stride doubles.  Not
     with sums_in. */
  int nseg, seg_stride;
  /* library version >= 6.  NULL: two launches (row reduction, finalize).  Else ceil(C / 32) ints, ZERO before the first call and
     left at zero by every call, not shared with a launch that may run concurrently: the row reduction's last workgroup per
     channel block finalizes it in the same launch (same arithmetic, same bits, one launch less per BatchNorm). */
  int* tickets;
} sslcr_bn_finalize_desc;
int sslcr_bn_finalize(const sslcr_bn_finalize_desc* d, void* stream);

typedef struct sslcr_bn_act_desc {      /* y = relu(x*scale+shift [+ res*rscale+rshift | + res])  (K5 apply, K6, K7) */
  const void* x; const float* scale; const float* shift;
  const void* res; const float* rscale; const float* rshift;
  void* y; size_t pixels; int C; int relu;
  int nseg, seg_stride;   /* nseg > 1: the pixels are nseg equal segments, segment s uses (r)scale / (r)shift + s * seg_stride */
  uint8_t* ybits;         /* optional (bf16 only): one bit per element of y in NHWC order, bit (e & 7) of byte e / 8 = (y[e] > 0) -- the
                             ReLU mask of a residual block's output as BatchNorm backward wants it (sslcr_bn_bwd_desc.yact_bits):
                             a sixteenth of the bytes of y */
} sslcr_bn_act_desc;
int sslcr_bn_act(int dtype, const sslcr_bn_act_desc* d, void* stream);

typedef struct sslcr_pool_fwd_desc {    /* maxpool3x3/2 pad 1 of relu(bn(x))   (K8); scale = shift = argmax = NULL: plain
                                           max-pool of x as it is (the eval path: BatchNorm folded into conv1, ReLU in its epilogue) */
  const void* x; const float* scale; const float* shift; void* y; uint8_t* argmax; int N, H, W, C, OH, OW;
} sslcr_pool_fwd_desc;
int sslcr_bn_relu_maxpool(int dtype, const sslcr_pool_fwd_desc* d, void* stream);
typedef struct sslcr_pool_bwd_desc {
  const void* dy; const uint8_t* argmax; const void* x; const float* scale; const float* shift; void* dx; int N, H, W, C, OH, OW;
} sslcr_pool_bwd_desc;
int sslcr_maxpool_relu_bwd(int dtype, const sslcr_pool_bwd_desc* d, void* stream);

int sslcr_avgpool_fwd(int dtype, const void* x, float* y, int N, int HW, int C, void* stream);          /* K9 */
int sslcr_avgpool_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, void* stream);

typedef struct sslcr_bn_bwd_desc {
  const void* dy; const void* x; const void* yact;
  const float* scale; const float* shift; const float* mean; const float* invstd;
  double* sums;           /* [2][C]: OVERWRITTEN by the reduce pass (ordered sum of its workgroups' rows: same bits every run) */
  void* dx; void* gout;
  size_t pixels; int C; int relu_from_x; double count;
  const void* pool_dy;    /* optional: dy is not given directly but through maxpool3x3/2 pad 1 -- pooled gradient [N][pOH][pOW][C] ... */
  const uint8_t* pool_argmax;   /* ... and the argmax codes recorded by sslcr_bn_relu_maxpool (window position 0..8; 9 = maximum not
                                   positive, no gradient); x is then [N][pH][pW][C] */
  int pH, pW, pOH, pOW;
  const void* pool_y;     /* optional with pool_dy: the max-pool OUTPUT saved by the forward, [N][pOH][pOW][C].  With it the reduce
                             pass reads only the pooled tensors: a pooled gradient lands on exactly one input pixel, whose
                             relu(bn(x)) IS the pooled output, so (x - mean) = (y - shift) / scale - mean where y > 0 */
  int g_in_reduce;        /* with yact and gout: the REDUCE pass writes the masked gradient g = dy * (yact > 0) to gout, and the
                             apply pass reads it back instead of dy and yact (one tensor read less; the identity path of a
                             residual block needs g anyway).  Not with pool_dy. */
  float* dgamma; float* dbeta;   /* optional: the apply pass also accumulates the affine gradients (what sslcr_bn_param_grads does: */
  float pg_scale;                /*   dgamma += pg_scale * sums[1] * invstd, dbeta += pg_scale * sums[0]) -- one launch less per BatchNorm */
  /* segments (sslcr_conv_desc.seg_images): nseg > 1 runs nseg BatchNorm batches in one launch -- `pixels` is then the total and
     each segment an equal share of it (tensors contiguous across segments); segment s uses scale / shift / mean / invstd
     + s * seg_stride floats and sums + s * sums_stride doubles; count = elements per channel of ONE segment; dgamma / dbeta take
     the segments' contributions one after the other, segment 0 first.  Not with pool_dy. */
  int nseg, seg_stride, sums_stride;
  const uint8_t* yact_bits;   /* instead of yact (bf16 only): its sign mask as written by sslcr_bn_act (ybits) -- g = dy where the bit is
                                 set, 0 elsewhere: the same g, bit for bit, for 1/16 of yact's bytes */
} sslcr_bn_bwd_desc;
/* ABI note (library version >= 4): sslcr_bn_bwd_reduce OVERWRITES d->sums (until version 3 it accumulated into a buffer the caller had
 * zeroed) -- several reduce calls into one sums buffer keep only the last one; add them on the caller's side.  The reduce pass and the
 * weight-gradient entry points (sslcr_conv2d_wgrad, sslcr_stem_wgrad*) keep per-stream scratch for their ordered folds: the first call
 * on a stream (and any call that needs more than before) allocates with hipMalloc / hipStreamSynchronize INSIDE the call, so these
 * entry points must not be issued under hipStreamBeginCapture unless an earlier un-captured call on that stream has sized the scratch.
 * The scratch is released by sslcr_destroy. */
int sslcr_bn_bwd_reduce(int dtype, const sslcr_bn_bwd_desc* d, void* stream);
int sslcr_bn_bwd_apply(int dtype, const sslcr_bn_bwd_desc* d, void* stream);
int sslcr_bn_param_grads(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, void* stream);
/* conv1 wgrad with bn0's backward apply pass computed on the fly (autograd of resnet18.conv1 <- bn1 <- relu <- maxpool,
 * models/net.py:32,77): `bn` is a pool-form descriptor (pool_dy, pool_argmax, x = the raw conv1 output, pH/pW = w->OH/OW) whose
 * reduce pass has run (sums final, all-reduced when sharded); the un-pooled gradient (bn->dx, w->dy) is never written or read --
 * each workgroup derives its 8x16 dY tile in LDS.  dgamma/dbeta are accumulated as sslcr_bn_bwd_apply would.  Same result, bit for
 * bit, as sslcr_bn_bwd_apply followed by sslcr_stem_wgrad. */
int sslcr_stem_wgrad_pool(int dtype, const sslcr_stem_wgrad_desc* w, const sslcr_bn_bwd_desc* bn, void* stream);

/* ---- heads and losses (models/net.py:12-15,35-36,111; F.mse_loss / F.cross_entropy at
 *      eval_BreastPathQ_SSL_CR.py:92-95, eval_Camelyon_SSL_CR.py:110-116, pretrain_BreastPathQ.py:56) */
int sslcr_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, void* stream);
int sslcr_linear_bwd(const float* x, const float* w, const float* dy, const float* yact, float* dx, float* dw, float* db,
                     int M, int N, int K, int dx_accumulate, float* scratch, void* stream);
typedef struct sslcr_loss_desc {
  int kind;               /* 0 mse+mse (BreastPathQ CR) ; 1 ce + hard-pseudo-label ce (Camelyon/Kather CR) ; 2 ce ; 3 mse */
  const float* logits; const float* logits_t;
  const float* target_f; const int64_t* target_i;
  float* dlogits;
  float* out;             /* [4]: loss, loss_x, loss_u, #correct */
  int nx, nu, C; float lambda_u;
  float inv_nx_global, inv_nu_global;
} sslcr_loss_desc;
int sslcr_loss(const sslcr_loss_desc* d, void* stream);
/* out[i] = softmax(logits[i, 0..C-1])[col]: the per-tile 'tumor' probability of test_Camelyon16.py:58-60 (row f3) */
int sslcr_softmax_col(const float* logits, float* out, int n, int C, int col, void* stream);

/* ---- optimizers (torch.optim.Adam / SGD nesterov as the reference configures them,
 *      eval_BreastPathQ_SSL_CR.py:481, eval_Camelyon_SSL_CR.py:514, pretrain_BreastPathQ.py:245; Lookahead lookahead.py:81-106) */
typedef struct sslcr_tensor_desc {
  float* p; float* g; float* s1; float* s2;
  int n;
  int K, C, RS;           /* conv weight: g is [K][RS][C] while p/s1/s2 are [K][C][RS] ; K=0: same layout */
  void* w_fwd;            /* optional (K > 0): the update also writes the conv kernels' shadow weights of the new value -- */
  void* w_dgrad;          /* forward pack [K][RS][C] and dgrad pack [C][RS][K] (sslcr_pack_conv layouts), element type pack_dtype */
  int pack_dtype;         /* 0 fp32, 1 bf16 */
  int dgrad_flip;         /* dgrad pack with the taps reversed (stride-1 3x3 dgrad runs as a plain conv of dY) */
} sslcr_tensor_desc;
typedef struct sslcr_opt_desc {
  int kind;               /* 0 adam, 1 sgd-nesterov */
  float lr, beta1, beta2, eps, wd, momentum;
  float bc1, bc2;
  int first_step;
  float grad_scale;
} sslcr_opt_desc;
int sslcr_optimizer_step(const sslcr_tensor_desc* device_descs, int ntensors, int max_n, const sslcr_opt_desc* o, void* stream);
int sslcr_axpby(float* p, float* q, size_t n, float alpha, int copy_back, void* stream);
int sslcr_fill(float* p, size_t n, float v, void* stream);

typedef struct sslcr_pack_desc {
  const float* w; void* w_fwd; void* w_dgrad;
  const float* gamma; const float* beta; const float* rmean; const float* rvar; float eps; float* bias_out;
  int K, C, R, S;
  int dgrad_flip;         /* w_dgrad taps reversed ([C][R-1-r][S-1-s][K]): a stride-1 dgrad then IS a plain conv of dY */
  float* scale_out;       /* library version >= 6, with gamma: [K] <- gamma / sqrt(rvar + eps) and w_fwd keeps the PLAIN filter (the fold's
                             scale goes to the conv's epilogue: sslcr_conv_desc.out_scale); bias_out is the shift either way.  NULL: folded */
} sslcr_pack_desc;
int sslcr_pack_conv(int dtype, const sslcr_pack_desc* d, void* stream);
int sslcr_pack_stem(int dtype, const sslcr_pack_desc* d, void* stream);

/* ---- device-side weak augmentation ("next" row f1): TransformFix.weak = RandomHorizontalFlip() + RandomCrop(image_size)
 *      (dataset.py:663-677), applied to a uint8 batch that is already in HBM.  The random draws stay on the host, in the
 *      reference's order per sample (flip: torch.rand(1) < 0.5 ; crop: randint(0, SH-OH+1), randint(0, SW-OW+1)), so a run
 *      is reproducible against the CPU transform; the kernel is the deterministic gather
 *        dst[n][c][i][j] = src[n][c][top+i][flip ? SW-1-(left+j) : left+j]
 *      writing the NCHW uint8 batch the stem ingests (eval_BreastPathQ_SSL_CR.py:68-74). */
typedef struct sslcr_weak_aug_desc {
  const uint8_t* src;     /* [N][3][SH][SW] (src_hwc=0) or [N][SH][SW][3] (src_hwc=1, the PIL/numpy layout) */
  uint8_t* dst;           /* [N][3][OH][OW] */
  const int32_t* params;  /* [N][3] device ints: flip (0/1), top, left */
  int N, SH, SW, OH, OW, src_hwc;
} sslcr_weak_aug_desc;
int sslcr_weak_augment(const sslcr_weak_aug_desc* d, void* stream);

/* ---- device-side strong augmentation, the colour ops of the reference's RandAugment pool ("next" row f4) on a uint8 batch in HBM.
 *      As with sslcr_weak_augment the random draws stay on the host, in the reference's order (ssl_cr_histo_amd/augment.py);
 *      the kernels are the deterministic per-pixel arithmetic.
 *
 *      sslcr_hed_colour_augment: models/randaugment.py:17-48 colour_augmentation() -- rgb2hed, per-image shift of the three stain
 *      channels, hed2rgb, (x * 255).astype(uint8) -- in float64, operation for operation as scikit-image 0.15.0 (requirements.txt:369)
 *      computes it (see csrc/augment.hip); replaces the reference's per-pixel Python loop (:35-38). */
typedef struct sslcr_colour_aug_desc {
  const uint8_t* src;       /* [N][3][H][W] (hwc=0, what the stem ingests) or [N][H][W][3] (hwc=1, the PIL/numpy layout) */
  uint8_t* dst;             /* same layout; may alias src */
  const double* shift;      /* [N][3] device doubles: (hmod, dmod, emod) of :30-32 */
  const uint8_t* apply;     /* [N] or NULL: 0 = the image is copied unchanged (RandAugment did not pick this op for it) */
  double hed_from_rgb[9];   /* row-major 3x3: skimage.color.hed_from_rgb = inv(rgb_from_hed) */
  double rgb_from_hed[9];
  int N, H, W, hwc;
} sslcr_colour_aug_desc;
int sslcr_hed_colour_augment(const sslcr_colour_aug_desc* d, void* stream);
/*      sslcr_brightness_contrast: models/randaugment.py:93-103 Brightness() / Contrast() = albumentations 0.1.8
 *      (requirements.txt:10) RandomBrightnessContrast: clip(float32(img) * alpha + beta * mean(img), 0, max(img)).astype(uint8),
 *      mean and max over the whole image. */
typedef struct sslcr_brightness_contrast_desc {
  const uint8_t* src; uint8_t* dst;      /* [N][3][H][W] or [N][H][W][3]: the map is layout-blind; dst may alias src */
  const double* alpha_beta; /* [N][2] device doubles: alpha = 1 + contrast draw, beta = brightness draw */
  const uint8_t* apply;     /* [N] or NULL */
  unsigned long long* stats;/* workspace [N][2] (sum, max), zeroed by the call */
  int N, H, W;
} sslcr_brightness_contrast_desc;
int sslcr_brightness_contrast(const sslcr_brightness_contrast_desc* d, void* stream);

/* ==================================================================================================
 * Engine: the whole ResNet18 TripletNet(_Finetune)+head graph, forward / backward / update, orchestrated
 * natively (one C call per step).  Replaces the step bodies of the reference's train()/validate():
 *   eval_BreastPathQ_SSL_CR.py:65-100, eval_Camelyon_SSL_CR.py:94-121, eval_Kather_SSL_CR.py:56-105,
 *   pretrain_BreastPathQ.py:42-61 and :110-128, eval_Camelyon_SSL.py:52-98, eval_BreastPathQ_SSL.py:52-84.
 * ================================================================================================== */
typedef struct sslcr_ctx sslcr_ctx;
typedef struct sslcr_net sslcr_net;

int sslcr_create(sslcr_ctx** out, int device, int dtype);
int sslcr_destroy(sslcr_ctx* ctx);
/* one process per GPU: rank 0 makes the 256-byte id pair, the host broadcasts it (torch.distributed), every rank inits.
 * With a communicator: gradients are all-reduced (bucketed, overlapped with backward) and train-mode BatchNorm uses
 * GLOBAL batch statistics (all-reduce of per-channel sums), which is what makes N ranks equal the single-device
 * reference (the reference's nn.DataParallel BN is per-replica, eval_BreastPathQ_SSL_CR.py:474-477). */
int sslcr_comm_unique_id(void* id256);       /* two 128-byte RCCL ids: [0] BatchNorm sums (compute stream), [1] gradient buckets (side stream) */
int sslcr_comm_init(sslcr_ctx* ctx, const void* id256, int rank, int world);
/* what the communicator itself reports (ncclCommUserRank / ncclCommCount when RCCL is in use): rank, world, transport
 * (0 none, 1 RCCL, 2 virtual ranks) -- bench.py prints it as ranks_seen. */
int sslcr_comm_info(sslcr_ctx* ctx, int* rank, int* world, int* transport);
/* in-place SUM of `n` floats over the ranks of ctx's communicator (RCCL or virtual) on `stream`; nothing happens without one.
 * The logging reduction of the step functions: per-rank loss shares -> global values, once per print_freq / epoch end (the
 * reference's meters, eval_BreastPathQ_SSL_CR.py:103-110, read .item() of losses nn.DataParallel had already gathered). */
int sslcr_comm_all_reduce_f32(sslcr_ctx* ctx, float* buf, size_t n, void* stream);
/* "Virtual ranks" (test infrastructure for single-GPU boxes; RCCL refuses two ranks on one device): `world` contexts of ONE
 * process on ONE device, one host thread and one stream each, exchange through a sslcr_vcomm instead of RCCL.  Every sharded
 * code path of the engine -- synced BatchNorm sums forward and backward, global-count loss scaling, bucketed gradient sums on
 * the side stream with their event joins -- runs exactly as with RCCL; only the transport differs (device-to-device copy +
 * rank-ordered sum kernel behind a host rendezvous, asynchronous on the callers' streams).  The nn.DataParallel seam this
 * replaces: eval_BreastPathQ_SSL_CR.py:474-477. */
typedef struct sslcr_vcomm sslcr_vcomm;
int sslcr_vcomm_create(sslcr_vcomm** out, int world);
int sslcr_vcomm_destroy(sslcr_vcomm* v);
int sslcr_comm_init_virtual(sslcr_ctx* ctx, sslcr_vcomm* v, int rank);
/* on (default): synced BatchNorm as described above.  off: every rank normalises with its own shard's statistics -- the
 * semantics of the reference's nn.DataParallel replicas -- and only the gradient buckets are exchanged. */
int sslcr_set_bn_sync(sslcr_ctx* ctx, int on);
/* on: sslcr_step_ssl_cr runs the (frozen, eval-mode) teacher forward on a second HIP stream next to the student forward --
 * the workgroups of one fill the tail rounds of the other's kernels (measured -0.4 ms of a 20.4 ms step in round 1; in round 6,
 * five boxes: together with sslcr_set_wgrad_stream every box runs the step in 15.5-15.6 ms where the serial step takes 15.2-15.75 ms
 * by box -- a gain on slow boxes, a loss on fast ones).  Off by default.
 * Forced off while sslcr_profile() is on: concurrent launches share the CUs, which makes per-kernel durations meaningless as a
 * statement about the kernel, and bench.py's roofline is built from those. */
int sslcr_set_aux_stream(sslcr_ctx* ctx, int on);
/* on: in backward the weight-gradient launches run on a second HIP stream behind an event -- they depend only on a
 * BatchNorm-backward output and feed only the gradient buffer, and they are MFMA-bound while the chain they leave behind
 * alternates MFMA-bound dgrads with HBM-bound BatchNorm passes.  Results are bit-identical.  OFF by default: measured
 * neutral on MI355X (18.84 vs 18.76 ms/step) -- the 4096-workgroup elementwise kernels fill every wave slot of the CUs, so
 * the 512-thread wgrad workgroups do not become co-resident and the two streams serialise.  Forced off while
 * sslcr_profile() is on. */
int sslcr_set_wgrad_stream(sslcr_ctx* ctx, int on);

/* measurement: bracket every conv launch of this ctx with HIP events on its own stream (bench.py roofline leg).
 * which = 0: conv_igemm (forward + dgrad), 1: wgrad.  out4 = {launches, total ms, algorithmic FLOPs, algorithmic bytes}.
 * sslcr_profile(ctx, 1) also clears the previous records. */
int sslcr_profile(sslcr_ctx* ctx, int enable);
int sslcr_profile_read(sslcr_ctx* ctx, int which, double* out4);
/* per kernel template instance: lines "name|launches|total_ms|algorithmic_flops|algorithmic_bytes" (names as rocprofv3 prints them) */
int sslcr_profile_dump(sslcr_ctx* ctx, char* buf, size_t n);

typedef struct sslcr_net_desc {
  float* const* params;                    /* [nparams] device pointers, named_parameters() order of models/net.py:
                                              0..59 resnet18 backbone, 60..63 fc.0/fc.2, then the classifier (2 or 4) */
  int nparams;
  float* const* bn_running_mean;           /* [20] in graph order */
  float* const* bn_running_var;
  int64_t* const* bn_num_batches_tracked;
  const uint8_t* requires_grad;            /* [nparams] host flags (freeze-by-index, eval_BreastPathQ_SSL_CR.py:408-441) */
  int head_kind;                           /* 0 FinetuneResNet 768->C (models/net.py:107-115); 1 Classifier 768->128->C (:8-20) */
  int num_classes;
  int triplet;                             /* 0 TripletNet_Finetune (models/net.py:70-103); 1 TripletNet (:25-66) */
} sslcr_net_desc;
int sslcr_net_create(sslcr_ctx* ctx, const sslcr_net_desc* d, sslcr_net** out);
int sslcr_net_destroy(sslcr_net* net);
int sslcr_net_set_requires_grad(sslcr_net* net, const uint8_t* flags);
/* re-derive the shadow weights from the bound fp32 parameters: mode bit0 = train packs (KRSC + dgrad CRSK),
 * bit1 = eval packs (BatchNorm running stats folded into KRSC weights + bias). Call after any parameter change.
 * SSLCR_FP8 contexts: mode 3 also drops the calibrated activation scales -- the first forward of the net in each mode (train,
 * eval) after sslcr_net_create or after a mode-3 pack runs the backbone TWICE (a calibration pass at scale 1 that only records
 * amax -- running statistics untouched, synced-BatchNorm all-reduces included -- then the real pass), so time steps after it. */
int sslcr_net_pack(sslcr_net* net, int mode, void* stream);

/* forward.  x2/x3 only for triplet nets.  train=0: eval-mode BN (folded), nothing saved.  train=1: batch-stat BN,
 * running stats updated (x3 for TripletNet_Finetune), activations kept for sslcr_net_backward. */
int sslcr_net_forward(sslcr_net* net, int train, const void* x1, const void* x2, const void* x3, int in_f32,
                      int N, int H, int W, float* feats, float* logits, void* stream);
/* backward of the last train forward from d(loss)/d(logits); zeroes then fills the engine's gradient buffer;
 * all-reduces it when a communicator is set. */
int sslcr_net_backward(sslcr_net* net, const float* dlogits, void* stream);
int sslcr_net_grad(sslcr_net* net, int param_index, float* out, void* stream);      /* -> PyTorch layout */
/* diagnostics for the layer-wise backward parity test (tests/test_backward_replay_gpu.py): autograd of one BasicBlock of
 * torchvision resnet18 (reached from models/net.py:32,77) checked kernel by kernel at the size the step really runs.  With the tap
 * on, sslcr_net_backward keeps a copy of every block's transient gradient tensors of pass 0; sslcr_net_debug_tensor copies one
 * tensor of block 0..7 (layer1.0 .. layer4.1) into `out` (device memory, engine storage dtype, NHWC) and reports its dims:
 *   kind 0 G     = d(block output) * (output > 0)           1 dRaw2 = gradient at conv2's raw output (after bn2 backward)
 *        2 dAct1 = conv2's dgrad (flags bit 0: bn1's ReLU mask already applied)    3 dRaw1 = gradient at conv1's raw output
 *        4 dRawD = gradient at the projection conv's raw output                    5 dXin  = gradient at the block input
 *        6 raw1, 7 raw2, 8 rawd (saved raw conv outputs), 9 y (block output), 10 x (block input)
 *       11..14 bn1's saved scale, shift, mean, invstd (fp32 [C]; dims = {C,1,1,1})
 * out == NULL: only dims4 / flags are filled. */
int sslcr_net_debug_tap(sslcr_net* net, int on);
/* 1 when the last train-mode forward of a TripletNet ran its three branches as segments of one launch per layer
 * (sslcr_conv_desc.seg_images; bf16, unsharded), 0 when it ran them pass by pass */
int sslcr_net_segments_used(const sslcr_net* net);
int sslcr_net_debug_tensor(sslcr_net* net, int block, int kind, void* out, size_t out_bytes, int* dims4, int* flags, void* stream);
/* fused multi-tensor update of every requires_grad parameter; state1/state2 = per-parameter optimizer state
 * (exp_avg/exp_avg_sq or momentum_buffer) owned by the caller, NULL entries for frozen parameters */
int sslcr_net_optimizer_step(sslcr_net* net, const sslcr_opt_desc* o, float* const* state1, float* const* state2, void* stream);
int sslcr_net_lookahead(sslcr_net* net, float* const* cached, float alpha, void* stream);   /* lookahead.py:93-97 */
int sslcr_net_ema_from(sslcr_net* teacher, sslcr_net* student, float decay, void* stream);  /* decay 0 == deepcopy (:515-516) */

typedef struct sslcr_ssl_cr_desc {
  int kind;                 /* 0 mse+mse (BreastPathQ) ; 1 ce + hard-pseudo-label ce (Camelyon, Kather) */
  const void* x_student;    /* [nx+nu,3,H,W]: labeled then strong-augmented unlabeled (torch.cat at :82) */
  const void* x_teacher;    /* [nu,3,H,W]: weak-augmented unlabeled */
  int in_f32, nx, nu, H, W;
  const float* target_f; const int64_t* target_i;
  float lambda_u;
  int nx_global, nu_global; /* global batch counts (== nx, nu on one GPU) */
  float* feats;             /* [nx+nu,768] out */
  float* logits;            /* [nx+nu,C] out */
  float* logits_t;          /* [nu,C] out */
  float* losses;            /* [4] out: loss, loss_x, loss_u, #correct */
  int backward;
  const void* x_student2;   /* optional: then x_student holds only the nx labeled images and x_student2 the nu strong-augmented
                               unlabeled ones (the concatenation of :82 happens in the stem kernel's addressing) */
} sslcr_ssl_cr_desc;
int sslcr_step_ssl_cr(sslcr_net* teacher, sslcr_net* student, const sslcr_ssl_cr_desc* d, void* stream);

typedef struct sslcr_sup_desc {
  int kind;                 /* 2 cross-entropy ; 3 mse */
  const void* x1; const void* x2; const void* x3;   /* x2,x3: TripletNet (RSP) only */
  int in_f32, n, H, W;
  const float* target_f; const int64_t* target_i;
  int n_global;
  float* feats; float* logits; float* losses;
  int train;                /* 1: train-mode BN ; 0: validate() */
  int backward;
} sslcr_sup_desc;
int sslcr_step_supervised(sslcr_net* net, const sslcr_sup_desc* d, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SSLCR_H_ */

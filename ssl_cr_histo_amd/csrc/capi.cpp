// extern "C" boundary of libsslcr.so: argument checks, error strings, no exceptions across the ABI.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "kernels.hpp"

namespace sslcr {
thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
int check(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  return fail("%s: %s", what, hipGetErrorString(e));
}
}  // namespace sslcr

using namespace sslcr;

#define NEED(cond, what) \
  do {                   \
    if (!(cond)) return fail("%s: invalid argument (%s)", __func__, what); \
  } while (0)
#define DT_OK(dt) NEED((dt) == SSLCR_F32 || (dt) == SSLCR_BF16, "dtype")

extern "C" {

int sslcr_version(void) { return 6; }      // = the build round; the ABI notes in include/sslcr.h name the version a behaviour changed in
const char* sslcr_last_error(void) { return g_err; }

int sslcr_conv2d(int dtype, const sslcr_conv_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->w && d->y, "null tensor");
  NEED(d->N > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->stride > 0, "shape");
  NEED(d->C % (dtype == SSLCR_BF16 ? 64 : 32) == 0, "C must be a multiple of the 128-byte channel slab");
  NEED(d->K % 64 == 0, "K % 64");
  NEED(d->osh >= 1 && d->PH > 0 && d->PW > 0, "pixel space");
  NEED(conv_segments_ok(dtype, *d), "seg_images: no segment form for this shape / dtype (sslcr_conv2d_segments_ok)");
  return check(launch_conv(dtype, *d, (hipStream_t)stream), "conv2d");
}
int sslcr_conv2d_partial_rows(const sslcr_conv_desc* d) { return d ? conv_partials_rows(*d) : -1; }
int sslcr_conv2d_segments_ok(int dtype, const sslcr_conv_desc* d) { return (d && (dtype == DT_F32 || dtype == DT_BF16) && conv_segments_ok(dtype, *d)) ? 1 : 0; }
const char* sslcr_conv2d_kernel_name(int dtype, const sslcr_conv_desc* d) { return d ? conv_kernel_name(dtype, *d) : ""; }
int sslcr_conv2d_s2_pair_ok(int dtype, const sslcr_conv_desc* c1, const sslcr_conv_desc* ds) {
  return (c1 && ds && conv_s2_pair_ok(dtype, *c1, *ds)) ? 1 : 0;
}
int sslcr_conv2d_s2_pair(int dtype, const sslcr_conv_desc* c1, const sslcr_conv_desc* ds, void* stream) {
  DT_OK(dtype);
  NEED(c1 && ds && c1->x && c1->w && c1->y && ds->w && ds->y, "null tensor");
  NEED(conv_s2_pair_ok(dtype, *c1, *ds), "pair not served (sslcr_conv2d_s2_pair_ok)");
  return check(launch_conv_s2(*c1, ds, (hipStream_t)stream), "conv2d_s2_pair");
}

int sslcr_conv2d_fp8(const sslcr_conv_desc* d, const sslcr_fp8_desc* q, void* stream) {
  NEED(d && q && d->x && d->y && q->w8 && q->w_dequant, "null tensor");
  NEED(q->x_scale > 0.f, "x_scale must be positive");
  NEED(conv_fp8_mode(*d) != 0, "shape not served by the fp8 kernel (3x3/1, C % 128, K % 128, 16x16-tileable or 8x8 with N % 4)");
  return check(launch_conv_fp8(*d, *q, (hipStream_t)stream), "conv2d_fp8");
}
int sslcr_conv2d_fp8_partial_rows(const sslcr_conv_desc* d) { return (d && conv_fp8_mode(*d)) ? conv_fp8_rows(*d) : 0; }
int sslcr_pack_conv_fp8(const sslcr_pack_fp8_desc* d, void* stream) {
  NEED(d && d->w && d->w8 && d->w_dequant && d->K > 0 && d->C > 0, "args");
  NEED(!d->gamma || (d->beta && d->rmean && d->rvar), "BatchNorm fold needs gamma, beta, running mean and var");
  return check(launch_pack_fp8(*d, (hipStream_t)stream), "pack_conv_fp8");
}

const char* sslcr_conv2d_wgrad_kernel_name(int dtype, const sslcr_wgrad_desc* d) { return d ? wgrad_kernel_name(dtype, *d) : ""; }
int sslcr_conv2d_wgrad(int dtype, const sslcr_wgrad_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->dy && d->dw, "null tensor");
  NEED(d->C % 64 == 0 && d->K % 64 == 0, "C,K % 64");
  NEED((d->R == 3 && d->S == 3) || (d->R == 1 && d->S == 1), "3x3 or 1x1");
  return check(launch_wgrad(dtype, *d, (hipStream_t)stream), "conv2d_wgrad");
}

int sslcr_probe_tr16(const uint16_t* in, const int* byte_addr, uint16_t* out, void* stream) {
  NEED(in && byte_addr && out, "null");
  return check(launch_probe_tr16(in, byte_addr, out, (hipStream_t)stream), "probe_tr16");
}

int sslcr_stem_conv(int dtype, const sslcr_stem_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->w && d->y, "null tensor");
  NEED(d->OH == (d->H + 6 - 7) / 2 + 1 && d->OW == (d->W + 6 - 7) / 2 + 1, "7x7/2 pad 3 output dims");
  NEED(!d->x2 || (d->n_split >= 0 && d->n_split <= d->N), "n_split outside the batch");
  return check(launch_stem(dtype, *d, (hipStream_t)stream), "stem_conv");
}
int sslcr_stem_conv_pool(int dtype, const sslcr_stem_desc* d, int POH, int POW, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->w && d->y, "null");
  NEED(stem_pool_ok(dtype, *d, POH, POW), "shape / mode not served by the fused stem + max-pool kernel");
  return check(launch_stem_pool(dtype, *d, POH, POW, (hipStream_t)stream), "stem_conv_pool");
}
int sslcr_stem_partial_rows(const sslcr_stem_desc* d) { return d ? stem_partials_rows(*d) : -1; }
int sslcr_stem_wgrad(int dtype, const sslcr_stem_wgrad_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->dy && d->dw, "null tensor");
  NEED(!d->x2 || (d->n_split >= 0 && d->n_split <= d->N), "n_split outside the batch");
  return check(launch_stem_wgrad(dtype, *d, (hipStream_t)stream), "stem_wgrad");
}
int sslcr_stem_wgrad_pool(int dtype, const sslcr_stem_wgrad_desc* w, const sslcr_bn_bwd_desc* bn, void* stream) {
  DT_OK(dtype);
  NEED(w && w->x && w->dw && bn && bn->pool_dy && bn->pool_argmax && bn->x && bn->sums && bn->mean && bn->invstd && bn->scale && bn->shift, "null tensor");
  NEED(!w->x2 || (w->n_split >= 0 && w->n_split <= w->N), "n_split outside the batch");
  NEED(bn->C == 64 && bn->pH == w->OH && bn->pW == w->OW && bn->pixels == (size_t)w->N * w->OH * w->OW, "bn descriptor is not this stem's");
  NEED(!bn->g_in_reduce && !bn->gout, "pool form has no g output");
  return check(launch_stem_wgrad_pool(dtype, *w, *bn, (hipStream_t)stream), "stem_wgrad_pool");
}

int sslcr_bn_finalize(const sslcr_bn_finalize_desc* d, void* stream) {
  NEED(d && d->C > 0, "desc");
  NEED(d->sums_in || (d->partials && d->stage && d->rows > 0), "partials/stage");
  NEED(d->sums_out || (d->gamma && d->beta && d->scale && d->shift && d->count > 0), "finalize outputs");
  return check(launch_bn_finalize(*d, (hipStream_t)stream), "bn_finalize");
}
int sslcr_bn_act(int dtype, const sslcr_bn_act_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->y && d->scale && d->shift, "null");
  NEED(d->C % 8 == 0, "C % 8");
  return check(launch_bn_act(dtype, *d, (hipStream_t)stream), "bn_act");
}
int sslcr_bn_relu_maxpool(int dtype, const sslcr_pool_fwd_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->y, "null");
  if (!d->scale || !d->shift) {
    const int cols = d->C / (dtype == SSLCR_BF16 ? 8 : 4);
    NEED(!d->scale && !d->shift && !d->argmax, "plain max-pool (scale = shift = NULL) records no argmax");
    NEED(cols >= 1 && cols <= 256 && 256 % cols == 0 && d->C % (dtype == SSLCR_BF16 ? 8 : 4) == 0, "plain max-pool: unsupported channel count");
  }
  return check(launch_bn_relu_maxpool(dtype, *d, (hipStream_t)stream), "bn_relu_maxpool");
}
int sslcr_maxpool_relu_bwd(int dtype, const sslcr_pool_bwd_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->x && d->dy && d->dx && d->argmax, "null");
  return check(launch_maxpool_relu_bwd(dtype, *d, (hipStream_t)stream), "maxpool_relu_bwd");
}
int sslcr_avgpool_fwd(int dtype, const void* x, float* y, int N, int HW, int C, void* stream) {
  DT_OK(dtype);
  NEED(x && y && N > 0 && HW > 0 && C % 8 == 0, "args");
  return check(launch_avgpool_fwd(dtype, x, y, N, HW, C, (hipStream_t)stream), "avgpool_fwd");
}
int sslcr_avgpool_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, void* stream) {
  DT_OK(dtype);
  NEED(dy && dx && N > 0 && HW > 0 && C % 8 == 0, "args");
  return check(launch_avgpool_bwd(dtype, dy, dx, N, HW, C, (hipStream_t)stream), "avgpool_bwd");
}
int sslcr_bn_bwd_reduce(int dtype, const sslcr_bn_bwd_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && (d->dy || (d->pool_dy && d->pool_argmax)) && d->x && d->sums && d->mean, "null");
  NEED(!d->g_in_reduce || (d->yact && d->gout && d->dy && !d->pool_dy), "g_in_reduce needs dy, yact and gout");
  return check(launch_bn_bwd_reduce(dtype, *d, (hipStream_t)stream), "bn_bwd_reduce");
}
int sslcr_bn_bwd_apply(int dtype, const sslcr_bn_bwd_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && (d->dy || (d->pool_dy && d->pool_argmax)) && d->x && d->sums && d->dx && d->mean && d->invstd && d->scale, "null");
  NEED(!d->g_in_reduce || (d->yact && d->gout && d->dy && !d->pool_dy), "g_in_reduce needs dy, yact and gout");
  return check(launch_bn_bwd_apply(dtype, *d, (hipStream_t)stream), "bn_bwd_apply");
}
int sslcr_bn_param_grads(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, void* stream) {
  NEED(sums && invstd && dgamma && dbeta, "null");
  return check(launch_bn_param_grads(sums, invstd, dgamma, dbeta, C, (hipStream_t)stream), "bn_param_grads");
}

int sslcr_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, void* stream) {
  NEED(x && w && y && M > 0 && N > 0 && K > 0, "args");
  return check(launch_linear_fwd(x, w, b, y, M, N, K, relu, (hipStream_t)stream), "linear_fwd");
}
int sslcr_linear_bwd(const float* x, const float* w, const float* dy, const float* yact, float* dx, float* dw, float* db,
                     int M, int N, int K, int dx_accumulate, float* scratch, void* stream) {
  NEED(x && w && dy && scratch && M > 0 && N > 0 && K > 0, "args");
  return check(launch_linear_bwd(x, w, dy, yact, dx, dw, db, M, N, K, dx_accumulate, scratch, (hipStream_t)stream), "linear_bwd");
}
int sslcr_loss(const sslcr_loss_desc* d, void* stream) {
  NEED(d && d->logits && d->out && d->nx > 0 && d->C > 0 && d->C <= 64, "args");
  return check(launch_loss(*d, (hipStream_t)stream), "loss");
}

int sslcr_softmax_col(const float* logits, float* out, int n, int C, int col, void* stream) {
  NEED(logits && out && n > 0 && C > 0 && C <= 64 && col >= 0 && col < C, "args");
  return check(launch_softmax_col(logits, out, n, C, col, (hipStream_t)stream), "softmax_col");
}

int sslcr_optimizer_step(const sslcr_tensor_desc* device_descs, int ntensors, int max_n, const sslcr_opt_desc* o, void* stream) {
  NEED(device_descs && o && ntensors > 0 && max_n > 0, "args");
  return check(launch_optimizer(device_descs, ntensors, max_n, *o, (hipStream_t)stream), "optimizer_step");
}
int sslcr_axpby(float* p, float* q, size_t n, float alpha, int copy_back, void* stream) {
  NEED(p && q, "null");
  return check(launch_axpby(p, q, n, alpha, copy_back, (hipStream_t)stream), "axpby");
}
int sslcr_fill(float* p, size_t n, float v, void* stream) {
  NEED(p, "null");
  return check(launch_fill(p, n, v, (hipStream_t)stream), "fill");
}
int sslcr_pack_conv(int dtype, const sslcr_pack_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->w && (d->w_fwd || d->w_dgrad), "null");
  return check(launch_pack_conv(dtype, *d, (hipStream_t)stream), "pack_conv");
}
int sslcr_weak_augment(const sslcr_weak_aug_desc* d, void* stream) {
  NEED(d && d->src && d->dst && d->params, "null");
  NEED(d->N > 0 && d->OH > 0 && d->OW > 0 && d->OH <= d->SH && d->OW <= d->SW, "crop larger than the source");
  return check(launch_weak_augment(*d, (hipStream_t)stream), "weak_augment");
}
int sslcr_hed_colour_augment(const sslcr_colour_aug_desc* d, void* stream) {
  NEED(d && d->src && d->dst && d->shift, "null");
  NEED(d->N > 0 && d->H > 0 && d->W > 0, "empty batch");
  return check(launch_hed_colour(*d, (hipStream_t)stream), "hed_colour_augment");
}
int sslcr_brightness_contrast(const sslcr_brightness_contrast_desc* d, void* stream) {
  NEED(d && d->src && d->dst && d->alpha_beta && d->stats, "null");
  NEED(d->N > 0 && d->H > 0 && d->W > 0, "empty batch");
  return check(launch_brightness_contrast(*d, (hipStream_t)stream), "brightness_contrast");
}
int sslcr_pack_stem(int dtype, const sslcr_pack_desc* d, void* stream) {
  DT_OK(dtype);
  NEED(d && d->w && d->w_fwd && d->K == 64 && d->C == 3 && d->R == 7 && d->S == 7, "stem shape");
  return check(launch_pack_stem(dtype, *d, (hipStream_t)stream), "pack_stem");
}

}  // extern "C"

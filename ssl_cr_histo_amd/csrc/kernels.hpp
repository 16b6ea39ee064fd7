// Internal kernel launch interface of the engine (host side).  The argument structs ARE the C-ABI
// descriptors of include/sslcr.h.  Every launcher is asynchronous on `st`; no allocation, no sync.
#pragma once
#include "common.hpp"
#include "../../include/sslcr.h"

namespace sslcr {

using ConvArgs = sslcr_conv_desc;
using WgradArgs = sslcr_wgrad_desc;
using StemArgs = sslcr_stem_desc;
using StemWgradArgs = sslcr_stem_wgrad_desc;
using BnFinalizeArgs = sslcr_bn_finalize_desc;
using BnActArgs = sslcr_bn_act_desc;
using PoolFwdArgs = sslcr_pool_fwd_desc;
using PoolBwdArgs = sslcr_pool_bwd_desc;
using BnBwdArgs = sslcr_bn_bwd_desc;
using LossArgs = sslcr_loss_desc;
using TensorDesc = sslcr_tensor_desc;
using OptArgs = sslcr_opt_desc;
using PackArgs = sslcr_pack_desc;
using Fp8Args = sslcr_fp8_desc;
using PackFp8Args = sslcr_pack_fp8_desc;

// conv_igemm.hip
hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t st);
int conv_tile_bp(const ConvArgs& a);
const char* conv_kernel_name(int dtype, const ConvArgs& a);
const char* wgrad_kernel_name(int dtype, const WgradArgs& a);
int conv_partials_rows(const ConvArgs& a);
bool conv_segments_ok(int dtype, const ConvArgs& a);
int device_cus();      // compute units of the current device (asked once per process, thread-safely; 256 if the query fails)
// conv_halo.hip
int conv_halo_tw(int dtype, const ConvArgs& a);
int conv_halo_tiles(const ConvArgs& a, int tw);
hipError_t launch_conv_halo(int dtype, const ConvArgs& a, int tw, hipStream_t st);
// conv_halo256.hip
int conv_halo256_mode(int dtype, const ConvArgs& a);
int conv_halo256_tiles(const ConvArgs& a, int mode);
hipError_t launch_conv_halo256(int dtype, const ConvArgs& a, int mode, hipStream_t st);
// conv_h16.hip
bool conv_h16_ok(int dtype, const ConvArgs& a);
int conv_h16_rows(const ConvArgs& a);
hipError_t launch_conv_h16(int dtype, const ConvArgs& a, hipStream_t st);
const char* conv_h16_name(int dtype, const ConvArgs& a);
// conv_pp64.hip: ping-pong form of the bf16 64 -> 64 resident-filter shape (same partial-row count as conv_h16_rows)
bool conv_pp64_ok(int dtype, const ConvArgs& a);
hipError_t launch_conv_pp64(const ConvArgs& a, hipStream_t st);
const char* conv_pp64_name(const ConvArgs& a);
int conv_pp64_rows(const ConvArgs& a);
// conv_dma.hip
int conv_dma_bp(int dtype, const ConvArgs& a);
int conv_dma_rows(const ConvArgs& a, int bp);
hipError_t launch_conv_dma(int dtype, const ConvArgs& a, int bp, hipStream_t st);
const char* conv_dma_name(int dtype, int bp);
// conv_s2.hip: 3x3 / 2 on 16x16 output tiles by plane-gathering LDS DMA, optionally with the 1x1 / 2 projection of the same input (bf16)
bool conv_s2_ok(int dtype, const ConvArgs& a);
bool conv_s2_pair_ok(int dtype, const ConvArgs& a, const ConvArgs& d);
int conv_s2_rows(const ConvArgs& a);
hipError_t launch_conv_s2(const ConvArgs& a, const ConvArgs* d, hipStream_t st);
const char* conv_s2_name(const ConvArgs& a, bool pair);
// conv_s2d.hip: the stride-2 3x3 dgrad, all four output-parity classes in one pass over dY (the par4 descriptor; bf16, 16x16-tileable dY)
bool conv_s2d_ok(int dtype, const ConvArgs& a);
hipError_t launch_conv_s2d(const ConvArgs& a, hipStream_t st);
const char* conv_s2d_name();
// conv_fp8.hip
int conv_fp8_mode(const ConvArgs& a);
int conv_fp8_rows(const ConvArgs& a);
const char* conv_fp8_name(const ConvArgs& a);
hipError_t launch_conv_fp8(const ConvArgs& a, const Fp8Args& q, hipStream_t st);
hipError_t launch_pack_fp8(const PackFp8Args& a, hipStream_t st);
hipError_t launch_fp8_scale_update(float* slots, int n, hipStream_t st);
// conv_wgrad.hip
hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t st);
int wgrad_halo_tw(const WgradArgs& a);
// per-stream scratch for partial results that a follow-up launch on the SAME stream folds in a fixed order: the accumulator slabs of
// the weight-gradient kernels (wgrad_fold_kernel) and the per-workgroup rows of the BatchNorm-backward reduce pass (bn_bwd_sums_kernel);
// the launches that fold them, wgrad_halo.hip
void* stream_scratch(hipStream_t st, size_t bytes);
void stream_scratch_release();      // frees every stream's scratch (sslcr_destroy, after a device synchronise)
hipError_t launch_wgrad_fold(const void* slabs, float* dw, int C, int gx, int gy, int splits, int taps, int kh_n, hipStream_t st);
hipError_t launch_stem_wgrad_fold(const void* slabs, float* dw, int nwg, hipStream_t st);
hipError_t launch_wgrad_halo(int dtype, const WgradArgs& a, int tw, hipStream_t st);
// wgrad_dma.hip: the halo kernel's 128-kout bf16 instances with the operands staged by LDS DMA (same tiles, slabs and bits)
bool wgrad_dma_ok(int dtype, const WgradArgs& a, int splits);
bool wgrad_dma_used(int dtype, const WgradArgs& a);
hipError_t launch_wgrad_dma(const WgradArgs& a, int tw, int tps, int ntiles, int splits, void* slabs, hipStream_t st);
hipError_t launch_probe_tr16(const uint16_t* in, const int* byte_addr, uint16_t* out, hipStream_t st);
// wgrad_s2.hip: 3x3 / 2 weight gradient with the input region staged once per tile as parity planes (bf16, no producer transform)
bool wgrad_s2_ok(int dtype, const WgradArgs& a);
hipError_t launch_wgrad_s2(const WgradArgs& a, hipStream_t st);
// stem.hip
hipError_t launch_stem(int dtype, const StemArgs& a, hipStream_t st);
int stem_partials_rows(const StemArgs& a);
hipError_t launch_stem_wgrad(int dtype, const StemWgradArgs& a, hipStream_t st);
hipError_t launch_stem_wgrad_pool(int dtype, const StemWgradArgs& a, const sslcr_bn_bwd_desc& b, hipStream_t st);
// stem_pool.hip: eval-mode stem + max-pool in one kernel (bf16)
bool stem_pool_ok(int dtype, const StemArgs& a, int POH, int POW);
hipError_t launch_stem_pool(int dtype, const StemArgs& a, int POH, int POW, hipStream_t st);
// bn_eltwise.hip
hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t st);
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t st);
hipError_t launch_bn_relu_maxpool(int dtype, const PoolFwdArgs& a, hipStream_t st);
hipError_t launch_maxpool_relu_bwd(int dtype, const PoolBwdArgs& a, hipStream_t st);
hipError_t launch_avgpool_fwd(int dtype, const void* x, float* y, int N, int HW, int C, hipStream_t st);
hipError_t launch_avgpool_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, hipStream_t st);
hipError_t launch_bn_bwd_reduce(int dtype, const BnBwdArgs& a, hipStream_t st);
hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t st);
// a downsampling block's two BatchNorms on one gradient (bn2 + the projection's): one reduce and one apply pass for both
bool bn_bwd_pair_ok(const BnBwdArgs& a, const BnBwdArgs& b);
// keep_g / from_g = 0: g is never written; the apply pass forms it again from (dy, mask)
hipError_t launch_bn_bwd_reduce_pair(int dtype, const BnBwdArgs& a, const BnBwdArgs& b, int keep_g, hipStream_t st);
hipError_t launch_bn_bwd_apply_pair(int dtype, const BnBwdArgs& a, const BnBwdArgs& b, int from_g, hipStream_t st);
hipError_t launch_bn_param_grads(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, hipStream_t st);
hipError_t launch_bn_param_grads_scaled(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, float scale, hipStream_t st);
// heads.hip
hipError_t launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, hipStream_t st);
hipError_t launch_linear_bwd(const float* x, const float* w, const float* dy, const float* yact, float* dx, float* dw, float* db,
                             int M, int N, int K, int dx_accumulate, float* scratch, hipStream_t st);
hipError_t launch_loss(const LossArgs& a, hipStream_t st);
hipError_t launch_softmax_col(const float* logits, float* out, int n, int C, int col, hipStream_t st);
// augment.hip
hipError_t launch_weak_augment(const sslcr_weak_aug_desc& a, hipStream_t st);
hipError_t launch_hed_colour(const sslcr_colour_aug_desc& a, hipStream_t st);
hipError_t launch_brightness_contrast(const sslcr_brightness_contrast_desc& a, hipStream_t st);
// optim.hip
hipError_t launch_optimizer(const TensorDesc* d_descs, int ntensors, int max_n, const OptArgs& o, hipStream_t st);
constexpr int OPT_CHUNK = 2048;     // elements per work-list entry
constexpr int OPT_TILE = 16 * 16 * 9; // ... and per LDS-transposed 3x3 filter tile (chunk.y = -(tile + 1))
hipError_t launch_optimizer_chunks(const TensorDesc* d_descs, const void* d_chunks /* int2 {tensor, first element} */, int nchunks,
                                   const OptArgs& o, hipStream_t st);
hipError_t launch_axpby(float* p, float* q, size_t n, float alpha, int copy_back, hipStream_t st);
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t st);
hipError_t launch_pack_conv(int dtype, const PackArgs& a, hipStream_t st);
hipError_t launch_pack_stem(int dtype, const PackArgs& a, hipStream_t st);
hipError_t launch_copy2d(float* dst, long ldd, const float* src, long lds, int rows, int w, int accumulate, hipStream_t st);
hipError_t launch_copy2d_multi(float* dst, long ldd, int ndst, long dstep, const float* src, long lds, int nsrc, long sstep, int rows, int w,
                               hipStream_t st);
hipError_t launch_unpack_grad(const float* g, float* out, int K, int C, int RS, hipStream_t st);
// capi.cpp
int fail(const char* fmt, ...);
int check(hipError_t e, const char* what);

}  // namespace sslcr

// Device-side TransformFix.weak (dataset.py:663-677): per-sample horizontal flip + crop of a uint8 batch, NCHW out.
// HBM-bound byte gather: 3 bytes read and 3 written per output pixel; a thread produces four consecutive output bytes of one
// row (one dword store), reading its four source bytes in either direction.  Parameters (flip, top, left) come from the host
// so that the random stream is the reference's.
#include "kernels.hpp"

namespace sslcr {

__global__ __launch_bounds__(256) void weak_augment_kernel(const sslcr_weak_aug_desc a) {
  const int qw = (a.OW + 3) / 4;                                  // dwords per output row
  const size_t total = (size_t)a.N * 3 * a.OH * qw;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int jq = (int)(t % qw);
    size_t r = t / qw;
    const int i = (int)(r % a.OH); r /= a.OH;
    const int c = (int)(r % 3);
    const int n = (int)(r / 3);
    const int flip = a.params[n * 3 + 0], top = a.params[n * 3 + 1], left = a.params[n * 3 + 2];
    const int y = top + i;
    uint32_t v = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = jq * 4 + e;
      if (j < a.OW) {
        const int x = flip ? a.SW - 1 - (left + j) : left + j;
        const size_t s = a.src_hwc ? (((size_t)n * a.SH + y) * a.SW + x) * 3 + c : (((size_t)n * 3 + c) * a.SH + y) * a.SW + x;
        v |= (uint32_t)a.src[s] << (8 * e);
      }
    }
    uint8_t* d = a.dst + (((size_t)n * 3 + c) * a.OH + i) * a.OW + jq * 4;
    if ((a.OW & 3) == 0) {
      *reinterpret_cast<uint32_t*>(d) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (jq * 4 + e < a.OW) d[e] = (uint8_t)(v >> (8 * e));
    }
  }
}

hipError_t launch_weak_augment(const sslcr_weak_aug_desc& a, hipStream_t st) {
  const size_t total = (size_t)a.N * 3 * a.OH * ((a.OW + 3) / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(weak_augment_kernel, dim3((int)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ================================================================================================================
// Strong augmentation, the colour ops the reference's RandAugment pool applies with its OWN arithmetic (row f4):
//
// hed_colour_kernel -- models/randaugment.py:17-48 colour_augmentation(): skimage.color.rgb2hed, a per-image shift of the
// (H, E, D) stain channels, hed2rgb, (x * 255).astype(uint8).  The reference walks the pixels in a Python loop (its largest
// wall-time sink); here one thread does one pixel in float64, operation for operation as scikit-image 0.15.0
// (requirements.txt:369) does it:
//     rgb  = u8 * (1/255) + 2                     img_as_float, then "rgb += 2"
//     s    = -log(rgb) . hed_from_rgb             separate_stains
//     s   += (hmod, dmod, emod)
//     rgb2 = exp(-s . rgb_from_hed)               combine_stains
//     c    = clip(rgb2 - 2, -1, 1);  v = ((c + 1) / 2) * 2 - 1      rescale_intensity(in_range=(-1, 1)) onto the float range (-1, 1)
//     out  = (uint8)(int)(v * 255)                numpy's float64 -> uint8 cast: truncation, wrap-around modulo 256
// HBM traffic is 3 bytes in + 3 out per pixel; the float64 log / exp make it VALU-bound (~0.3 ms for 640 images of 256x256).
__global__ __launch_bounds__(256) void hed_colour_kernel(const sslcr_colour_aug_desc a) {
  const size_t hw = (size_t)a.H * a.W;
  const size_t total = (size_t)a.N * hw;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int n = (int)(t / hw);
    const size_t p = t - (size_t)n * hw;
    size_t i0, st;
    if (a.hwc) { i0 = ((size_t)n * hw + p) * 3; st = 1; } else { i0 = (size_t)n * 3 * hw + p; st = hw; }
    const uint8_t r = a.src[i0], g = a.src[i0 + st], b = a.src[i0 + 2 * st];
    if (a.apply && !a.apply[n]) { a.dst[i0] = r; a.dst[i0 + st] = g; a.dst[i0 + 2 * st] = b; continue; }
    const double inv255 = 1.0 / 255.0;
    // (no fused multiply-add anywhere: numpy rounds the product, then the sum)
    const double l0 = -log(__dadd_rn(__dmul_rn((double)r, inv255), 2.0)), l1 = -log(__dadd_rn(__dmul_rn((double)g, inv255), 2.0)),
                 l2 = -log(__dadd_rn(__dmul_rn((double)b, inv255), 2.0));
    double s[3], o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {      // row vector times matrix, left to right like the 3-term dot product
      double acc = __dmul_rn(l0, a.hed_from_rgb[0 * 3 + j]);
      acc = __dadd_rn(acc, __dmul_rn(l1, a.hed_from_rgb[1 * 3 + j]));
      acc = __dadd_rn(acc, __dmul_rn(l2, a.hed_from_rgb[2 * 3 + j]));
      s[j] = -(acc + a.shift[n * 3 + j]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = __dmul_rn(s[0], a.rgb_from_hed[0 * 3 + j]);
      acc = __dadd_rn(acc, __dmul_rn(s[1], a.rgb_from_hed[1 * 3 + j]));
      acc = __dadd_rn(acc, __dmul_rn(s[2], a.rgb_from_hed[2 * 3 + j]));
      double c = exp(acc) - 2.0;
      c = fmin(fmax(c, -1.0), 1.0);
      o[j] = __dadd_rn(__dmul_rn(__dadd_rn(c, 1.0) / 2.0, 2.0), -1.0);
    }
    a.dst[i0] = (uint8_t)(int)(o[0] * 255.0);
    a.dst[i0 + st] = (uint8_t)(int)(o[1] * 255.0);
    a.dst[i0 + 2 * st] = (uint8_t)(int)(o[2] * 255.0);
  }
}

hipError_t launch_hed_colour(const sslcr_colour_aug_desc& a, hipStream_t st) {
  const size_t total = (size_t)a.N * a.H * a.W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(hed_colour_kernel, dim3((int)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

// brightness / contrast -- models/randaugment.py:93-103 Brightness() / Contrast() = albumentations 0.1.8 (requirements.txt:10)
// RandomBrightnessContrast -> functional.brightness_contrast_adjust under its @clipped decorator:
//     out = clip(float32(img) * alpha + beta * mean(img), 0, max(img)).astype(uint8)
// mean (float64) and max are over the WHOLE image (all three channels); the arithmetic is float32 with separate roundings of
// the product and of the sum, the cast truncates.  Two launches: per-image (sum, max) by INTEGER atomics into `stats` (order-independent), then the map.
__global__ __launch_bounds__(256) void image_sum_max_kernel(const sslcr_brightness_contrast_desc a) {
  const size_t per = (size_t)3 * a.H * a.W;
  const int n = blockIdx.y;
  const uint8_t* src = a.src + (size_t)n * per;
  unsigned long long s = 0;
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
    const unsigned v = src[i];
    s += v;
    m = v > m ? v : m;
  }
  __shared__ unsigned long long ss[256];
  __shared__ unsigned sm[256];
  ss[threadIdx.x] = s; sm[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      sm[threadIdx.x] = sm[threadIdx.x + o] > sm[threadIdx.x] ? sm[threadIdx.x + o] : sm[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(a.stats + 2 * n, ss[0]);
    atomicMax(a.stats + 2 * n + 1, (unsigned long long)sm[0]);
  }
}

__global__ __launch_bounds__(256) void brightness_contrast_kernel(const sslcr_brightness_contrast_desc a) {
  const size_t per = (size_t)3 * a.H * a.W;
  const size_t total = (size_t)a.N * per;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int n = (int)(t / per);
    const uint8_t v = a.src[t];
    if (a.apply && !a.apply[n]) { a.dst[t] = v; continue; }
    const double mean = (double)a.stats[2 * n] / (double)per;
    const float maxv = (float)a.stats[2 * n + 1];
    const float alpha = (float)a.alpha_beta[2 * n];                       // numpy: float32 array * python float -> float32
    const float add = (float)(a.alpha_beta[2 * n + 1] * mean);            // beta * np.mean(img) is a float64 scalar, added as float32
    float x = __fadd_rn(__fmul_rn((float)v, alpha), add);
    x = fminf(fmaxf(x, 0.f), maxv);
    a.dst[t] = (uint8_t)(int)x;
  }
}

hipError_t launch_brightness_contrast(const sslcr_brightness_contrast_desc& a, hipStream_t st) {
  hipError_t e = hipMemsetAsync(a.stats, 0, (size_t)a.N * 2 * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  const size_t per = (size_t)3 * a.H * a.W;
  int bx = (int)((per + 256 * 16 - 1) / (256 * 16));
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(image_sum_max_kernel, dim3(bx, a.N), dim3(256), 0, st, a);
  size_t blocks = ((size_t)a.N * per + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(brightness_contrast_kernel, dim3((int)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace sslcr

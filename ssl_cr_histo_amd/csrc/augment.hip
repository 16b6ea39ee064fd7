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

}  // namespace sslcr

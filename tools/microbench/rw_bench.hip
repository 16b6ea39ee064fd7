// 2-read 1-write bf16 streaming kernel (the shape of bn_bwd_apply: dY, x -> g) against the shape of the live window:
// a persistent grid of G workgroups; workgroup w belongs to region w % R (R = 8: one region per XCD) and walks that region's n/R
// bytes with the G/R workgroups of its region.  R = 1 is the plain grid-stride loop.  U = loads in flight per thread and tensor.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void k(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, u32x4_t* __restrict__ c, size_t n16, int R) {
  const int region = blockIdx.x % R, wr = blockIdx.x / R, gr = gridDim.x / R;
  const size_t per = n16 / R, base = (size_t)region * per, stride = (size_t)gr * 256;
  size_t i = (size_t)wr * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < per; i += U * stride) {
    u32x4_t x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = a[base + i + u * stride]; y[u] = b[base + i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) c[base + i + u * stride] = x[u] ^ y[u];
  }
  for (; i < per; i += stride) c[base + i] = a[base + i] ^ b[base + i];
}
int main() {
  const size_t bytes = (size_t)640 * 64 * 64 * 64 * 2;       // a layer1 activation at N = 640: 335 MB per tensor
  u32x4_t *a, *b, *c; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {768, 2048, 4096})
    for (int R : {1, 2, 4, 8, 16, 64}) {
      k<2><<<grid, 256>>>(a, b, c, bytes / 16, R); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) k<2><<<grid, 256>>>(a, b, c, bytes / 16, R);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d regions %4d: %7.1f us  %.2f TB/s\n", grid, R, ms * 50, 3 * bytes / (ms / 20 * 1e-3) / 1e12);
    }
  return 0;
}

// practical HBM ceiling for a 2-read 1-write bf16 streaming kernel (the shape of bn_bwd_apply)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, u32x4_t* __restrict__ c, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  // (round 4: the remainder loop below was missing -- with n = 5 grid strides at grid = 16384 the U = 2 / 4 instances skipped a fifth
  //  of the tensor and their "6.7-6.9 TB/s" of round 1 were 5.4-5.6 TB/s of bytes actually moved)
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4_t x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
      y[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4_t r = x[u] ^ y[u];
      if (NT) __builtin_nontemporal_store(r, c + i + u * stride); else c[i + u * stride] = r;
    }
  }
  for (; i < n; i += stride) {
    const u32x4_t x = NT ? __builtin_nontemporal_load(a + i) : a[i], y = NT ? __builtin_nontemporal_load(b + i) : b[i];
    const u32x4_t r = x ^ y;
    if (NT) __builtin_nontemporal_store(r, c + i); else c[i] = r;
  }
}
template <int U, bool NT>
void run(int grid, size_t n, u32x4_t* a, u32x4_t* b, u32x4_t* c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, c, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, c, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("U=%d nt=%d grid=%5d : %7.1f GB/s\n", U, (int)NT, grid, 3.0 * n * 16 * 10 / ms / 1e6);
}
int main() {
  const size_t n = (size_t)640 * 64 * 64 * 64 * 2 / 16;   // 335 MB per tensor (layer1 activation at N=640)
  u32x4_t *a, *b, *c;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
  hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
  for (int grid : {1024, 2048, 4096, 8192, 16384, 20480, 40960, 81920}) {     // (20480 x 256 x 4 = n: exactly one U = 4 pass)
    run<1, false>(grid, n, a, b, c);
    run<2, false>(grid, n, a, b, c);
    run<4, false>(grid, n, a, b, c);
    run<1, true>(grid, n, a, b, c);
    run<4, true>(grid, n, a, b, c);
  }
  return 0;
}

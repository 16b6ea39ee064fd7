// write-only HBM rate against the SHAPE of the live window: a persistent grid of G workgroups writes a 1.34 GB tensor; workgroup w
// belongs to region w % R and walks that region's n/R bytes with the G/R workgroups of its region (R = 1: one contiguous moving
// window of G x 4 KB; R > 1: R windows of G/R x 4 KB, n/R bytes apart).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(u32x4_t* __restrict__ c, size_t n16, int R, unsigned seed) {
  const int region = blockIdx.x % R, wr = blockIdx.x / R, gr = gridDim.x / R;
  const size_t per = n16 / R, base = (size_t)region * per, stride = (size_t)gr * 256;
  const u32x4_t v = {seed, seed + 1, seed + 2, seed + threadIdx.x};
  for (size_t i = (size_t)wr * 256 + threadIdx.x; i < per; i += stride) c[base + i] = v;
}
int main() {
  const size_t bytes = (size_t)640 * 128 * 128 * 64 * 2;
  u32x4_t* c; hipMalloc(&c, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1024, 2048, 4096})
    for (int R : {1, 2, 4, 8, 16, 64, 256}) {
      k<<<grid, 256>>>(c, bytes / 16, R, 1); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) k<<<grid, 256>>>(c, bytes / 16, R, r);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d regions %4d: %7.1f us  %.2f TB/s\n", grid, R, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}

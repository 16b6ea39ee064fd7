// write-only HBM ceiling: what a kernel that only STORES a 1.34 GB bf16 tensor (the stem's raw conv output at N = 640) can reach,
// (a) fully coalesced 16-byte stores, (b) the stem output stage's pattern: a wave stores 16 pixels x 64 bytes (half of each
// pixel's 128-byte channel vector) per instruction, the other half with the next instruction; plain and non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT, bool HALF>
__global__ __launch_bounds__(256) void k(u32x4_t* __restrict__ c, size_t n16, unsigned seed) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u32x4_t v = {seed, seed + 1, seed + 2, seed + threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    size_t j = i;
    if (HALF) {   // lane (li = pixel 0..15, g = 16-byte chunk 0..3): instruction pair (q = 0, 1) covers 16 pixels x 128 B
      const size_t w = i >> 6, l = i & 63;                   // wave-instruction index, lane
      const size_t pair = w >> 1, q = w & 1, li = l & 15, g = l >> 4;
      j = pair * 128 + li * 8 + q * 4 + g;
    }
    if (NT) __builtin_nontemporal_store(v, c + j); else c[j] = v;
  }
}
template <bool NT, bool HALF>
void run(u32x4_t* c, size_t bytes, int grid, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NT, HALF><<<grid, 256>>>(c, bytes / 16, 1); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) k<NT, HALF><<<grid, 256>>>(c, bytes / 16, r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s grid %6d: %7.1f us  %.2f TB/s\n", name, grid, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
}
int main() {
  const size_t bytes = (size_t)640 * 128 * 128 * 64 * 2;
  u32x4_t* c; hipMalloc(&c, bytes);
  for (int grid : {1024, 2048, 8192, 65536}) {
    run<false, false>(c, bytes, grid, "coalesced plain");
    run<true, false>(c, bytes, grid, "coalesced nt");
    run<false, true>(c, bytes, grid, "half-line pairs plain");
    run<true, true>(c, bytes, grid, "half-line pairs nt");
  }
  return 0;
}

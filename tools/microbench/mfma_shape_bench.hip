// MFMA-only ceilings with random operands: 16x16x32 vs 32x32x16 bf16 (power-limited part: which shape sustains more FLOP/s?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ u32x4_t rnd4(unsigned h) { u32x4_t v; for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; v[e] = (0x3c00u + ((h >> 8) & 0x7ffu) + ((h >> 3) & 0x8000u)) | ((0x3c00u + ((h >> 20) & 0x7ffu) + ((h >> 1) & 0x8000u)) << 16); } return v; }
template <int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8_t A[4], B[4];
  for (int i = 0; i < 4; ++i) { A[i] = __builtin_bit_cast(bf16x8_t, rnd4(tid * 31 + i * 7 + blockIdx.x)); B[i] = __builtin_bit_cast(bf16x8_t, rnd4(tid * 17 + i * 13 + 5 + blockIdx.x)); }
  float s = 0;
  if (SHAPE == 16) {
    f32x4_t acc[4][4];
    for (int t = 0; t < 4; ++t) for (int p = 0; p < 4; ++p) acc[t][p] = f32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t], B[p], acc[t][p], 0, 0, 0);
    for (int t = 0; t < 4; ++t) for (int p = 0; p < 4; ++p) s += acc[t][p][0] + acc[t][p][3];
  } else {
    f32x16_t acc[2][2];
    for (int t = 0; t < 2; ++t) for (int p = 0; p < 2; ++p) for (int e = 0; e < 16; ++e) acc[t][p][e] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p) acc[t][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t + 2 * kk], B[p + 2 * kk], acc[t][p], 0, 0, 0);
    for (int t = 0; t < 2; ++t) for (int p = 0; p < 2; ++p) s += acc[t][p][0] + acc[t][p][15];
  }
  out[blockIdx.x * WAVES * 64 + tid] = s;
}
template <int SHAPE, int WAVES> void run(const char* name) {
  float* out; hipMalloc(&out, 4 << 20);
  const int iters = 20000, grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, 100); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * WAVES * iters * (SHAPE == 16 ? 16 * 16384.0 : 8 * 32768.0);
  printf("%-36s %8.3f ms %7.1f TF/s\n", name, ms, fl / ms / 1e9);
}
int main() {
  run<16, 8>("16x16x32 bf16, 8 waves/CU");
  run<32, 8>("32x32x16 bf16, 8 waves/CU");
  run<16, 4>("16x16x32 bf16, 4 waves/CU");
  run<32, 4>("32x32x16 bf16, 4 waves/CU");
  run<16, 8>("16x16x32 bf16, 8 waves/CU (again)");
  return 0;
}

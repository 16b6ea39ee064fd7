// micro-benchmark of the conv inner loop: 8 (or 4) waves, per step 4+4 (or 8+4) ds_read_b128 fragments and TK*TP MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4_t ld16(const void* p) { return *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ void mma(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// MODE 0: reads(next) ; fence ; mfmas(cur)      MODE 1: interleaved by sched_group_barrier      MODE 2: mode 0 + setprio
// MODE 3: no LDS reads at all (MFMA ceiling)     MODE 4: reads only
// NV > 0: NV extra independent VALU fmas per step on a separate register bank (stand-in for a deferred epilogue / halo transform):
//   MODE 0 -> after the step's MFMAs in program order ; MODE 1 -> spread between the MFMAs by sched_group_barrier
template <int WAVES, int TK, int TP, int MODE, int NV = 0, int MINW = 2>
__global__ __launch_bounds__(WAVES * 64, MINW) void loop_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  for (int i = tid; i < 140 * 1024 / 16; i += WAVES * 64) { unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; u32x4_t v; for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; unsigned lo = 0x3c00u + ((h >> 8) & 0x7ffu) + ((h >> 3) & 0x8000u); unsigned hi = 0x3c00u + ((h >> 20) & 0x7ffu) + ((h >> 1) & 0x8000u); v[e] = lo | (hi << 16); } reinterpret_cast<u32x4_t*>(smem)[i] = v; }
  __syncthreads();
  const char* hb = smem;            // "halo": 55 KB
  const char* wbuf = smem + 55296;  // "ring"
  int Bb[2], Ab[2];
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
    Bb[kk] = (((wave & 3) * 4) * 24 + li) * 128 + ((ci ^ (li & 7)) << 4);
    Ab[kk] = ((wave >> 2) * 16 * TK + li) * 128 + ((ci ^ (li & 7)) << 4);
  }
  f32x4_t acc[TK][TP];
  for (int t = 0; t < TK; ++t) for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0, 0, 0, 0};
  float ev[32];
  for (int e = 0; e < 32; ++e) ev[e] = 1.f + 0.001f * (tid + e);
  u32x4_t A[2][TK], B[2][TP];
  auto frags = [&](int buf, int step) {
    const int kk = step & 1, s = (step >> 1) % 3;
#pragma unroll
    for (int t = 0; t < TK; ++t) A[buf][t] = ld16(wbuf + Ab[kk] + s * 16384 + t * 2048);
#pragma unroll
    for (int p = 0; p < TP; ++p) B[buf][p] = ld16(hb + Bb[kk] + s * 128 + p * 3072);
  };
  if (MODE == 3) { for (int t = 0; t < TK; ++t) A[0][t] = A[1][t] = ld16(smem + (tid * 16 + t * 8192) % 100000 / 16 * 16); for (int p = 0; p < TP; ++p) B[0][p] = B[1][p] = ld16(smem + (tid * 16 + p * 8192 + 40000) % 100000 / 16 * 16); }
  else frags(0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (MODE != 3) frags((i + 1) & 1, i + 1);
      if (MODE == 0 || MODE == 2 || MODE == 4) __builtin_amdgcn_sched_barrier(0);
      if (MODE == 2) __builtin_amdgcn_s_setprio(1);
      if (MODE != 4) {
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
          for (int p = 0; p < TP; ++p) mma(A[i & 1][t], B[i & 1][p], acc[t][p]);
      } else {
#pragma unroll
        for (int t = 0; t < TK; ++t) asm volatile("" ::"v"(A[i & 1][t]));
#pragma unroll
        for (int p = 0; p < TP; ++p) asm volatile("" ::"v"(B[i & 1][p]));
      }
      if (MODE == 2) __builtin_amdgcn_s_setprio(0);
      if (NV > 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) ev[e & 31] = fmaf(ev[e & 31], 1.0001f, 0.5f);
      }
      if (MODE == 1 && NV > 0) {
#pragma unroll
        for (int q = 0; q < TK + TP; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, TK * TP / (TK + TP), 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NV / (TK + TP), 0);
        }
      } else
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < TK + TP; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // 1 DS read
          __builtin_amdgcn_sched_group_barrier(0x008, TK * TP / (TK + TP), 0);  // MFMAs
        }
      }
      if (MODE == 0 || MODE == 2 || MODE == 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int e = 0; e < 32; ++e) s += ev[e];
  for (int t = 0; t < TK; ++t) for (int p = 0; p < TP; ++p) s += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
  out[blockIdx.x * WAVES * 64 + tid] = s;
}
template <int WAVES, int TK, int TP, int MODE, int NV = 0, int MINW = 2>
void run(const char* name, int grid_mul) {
  float* out; hipMalloc(&out, 4 << 20);
  auto k = loop_kernel<WAVES, TK, TP, MODE, NV, MINW>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 2000, grid = 256 * grid_mul;
  const size_t lds = grid_mul == 1 ? 150 * 1024 : 75 * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * WAVES * iters * 6 * TK * TP * 16384.0;
  printf("%-44s %8.3f ms  %7.1f TF/s  (err=%s)\n", name, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
  hipFree(out);
}
int main() {
  run<8, 4, 4, 3>("8w 4x4 MFMA only (no LDS)", 1);
  run<8, 4, 4, 4>("8w 4x4 LDS reads only", 1);
  run<8, 4, 4, 0>("8w 4x4 reads(next)|mfma(cur)", 1);
  run<8, 4, 4, 1>("8w 4x4 interleaved 1 read : 2 mfma", 1);
  run<8, 4, 4, 2>("8w 4x4 mode0 + setprio", 1);
  run<4, 4, 4, 0>("4w 4x4 (1 wave/SIMD) mode0", 1);
  run<4, 4, 4, 0>("4w 4x4 x2 WG/CU mode0", 2);
  run<4, 8, 4, 0>("4w 8x4 (1 wave/SIMD) mode0", 1);
  run<4, 8, 4, 1>("4w 8x4 interleaved", 1);
  run<4, 8, 4, 3>("4w 8x4 MFMA only", 1);
  run<8, 4, 4, 0, 64>("8w 4x4 mode0 + 64 VALU/step after", 1);
  run<8, 4, 4, 1, 64>("8w 4x4 interleaved + 64 VALU/step spread", 1);
  run<8, 4, 4, 0, 128>("8w 4x4 mode0 + 128 VALU/step after", 1);
  run<8, 4, 4, 1, 128>("8w 4x4 interleaved + 128 VALU/step spread", 1);
  run<4, 8, 4, 0, 0, 1>("4w 8x4 512-reg budget mode0", 1);
  run<4, 8, 4, 1, 0, 1>("4w 8x4 512-reg budget interleaved", 1);
  run<4, 8, 4, 0, 128, 1>("4w 8x4 512-reg + 128 VALU/step after", 1);
  run<4, 8, 4, 1, 132, 1>("4w 8x4 512-reg + 132 VALU/step spread", 1);
  run<4, 8, 4, 1, 264, 1>("4w 8x4 512-reg + 264 VALU/step spread", 1);
  run<8, 2, 4, 0>("8w 2x4 mode0", 1);
  run<8, 2, 4, 1>("8w 2x4 interleaved", 1);
  return 0;
}

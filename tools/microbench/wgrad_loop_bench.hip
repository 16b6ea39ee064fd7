// micro-benchmark of the wgrad inner loop: per 32-pixel depth step 8 A + 18 B transpose reads and 36 MFMAs per wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short h16x4_t __attribute__((ext_vector_type(4)));
typedef short h16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t tr_pair(const char* p0, const char* p1) {
  h16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4_t*)(p0));
  h16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4_t*)(p1));
  h16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}
// MODE 0: pipelined B frags with fences (as the kernel)  1: no fences  2: MFMA only  3: B frags 2 taps ahead
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  for (int i = tid; i < 46 * 1024 / 16; i += 256) { unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; u32x4_t v; for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; v[e] = (0x3c00u + ((h >> 8) & 0x7ffu)) | ((0x3c00u + ((h >> 20) & 0x7ffu)) << 16); } reinterpret_cast<u32x4_t*>(smem)[i] = v; }
  __syncthreads();
  const char* yb = smem; const char* hb = smem + 16384;
  constexpr int RB = 128, PITCH = 24;
  const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
  int Aoff[4], Boff[3];
  for (int t4 = 0; t4 < 4; ++t4) Aoff[t4] = pl * RB + ((t4 ^ ((pl >> 1) & 3)) << 5) + (li & 3) * 8;
  for (int sx = 0; sx < 3; ++sx) { const int hp = (pl >> 4) * PITCH + (pl & 15) + sx; Boff[sx] = hp * RB + ((wave ^ ((hp >> 1) & 3)) << 5) + (li & 3) * 8; }
  f32x4_t acc[9][4];
  for (int t = 0; t < 9; ++t) for (int c = 0; c < 4; ++c) acc[t][c] = f32x4_t{0, 0, 0, 0};
  bf16x8_t bfr[3];
  auto bfrag_of = [&](int q, int t) { const int qoff = ((q * 2 + t / 3) * PITCH) * RB; return tr_pair(hb + Boff[t % 3] + qoff, hb + Boff[t % 3] + qoff + 8 * RB); };
  bf16x8_t afc[4];
  for (int t4 = 0; t4 < 4; ++t4) afc[t4] = tr_pair(yb + Aoff[t4], yb + Aoff[t4] + 8 * RB);
  bfr[0] = bfrag_of(0, 0); bfr[1] = bfrag_of(0, 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * q2 + qq, par = qq;
        bf16x8_t af[4];
        if (MODE == 2) { for (int t4 = 0; t4 < 4; ++t4) af[t4] = afc[t4]; }
        else for (int t4 = 0; t4 < 4; ++t4) af[t4] = tr_pair(yb + q * 32 * RB + Aoff[t4], yb + q * 32 * RB + Aoff[t4] + 8 * RB);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (MODE == 0 || MODE == 1) bfr[(t + 1 + par) & 1] = t < 8 ? bfrag_of(q, t + 1) : bfrag_of(q < 3 ? q + 1 : 3, 0);
          if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
            acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t4], bfr[MODE == 2 ? 0 : (t + par) & 1], acc[t][t4], 0, 0, 0);
          if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float s = 0;
  for (int t = 0; t < 9; ++t) for (int c = 0; c < 4; ++c) s += acc[t][c][0] + acc[t][c][1] + acc[t][c][2] + acc[t][c][3];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 4 << 20);
  auto kern = k<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  const int iters = 500, grid = 512; const size_t lds = 56 * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %8.3f ms %7.1f TF/s (%s)\n", name, ms, (double)grid * 4 * iters * 144 * 16384.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() { run<2>("MFMA only"); run<0>("pipelined B + fences (kernel)"); run<1>("no fences"); return 0; }

// How fast can a CU fill LDS from L2-resident data?  global_load_lds_dwordx4 (DMA, no registers) against
// global_load_dwordx4 + ds_write_b128 (through VGPRs), alone and underneath a stream of MFMAs.
//   hipcc --offload-arch=gfx950 -O3 lds_fill_bench.hip -o lds_fill_bench && ./lds_fill_bench
// One 8-wave workgroup per CU; per iteration it stages KB_PER_ITER KiB (the conv3x3_h16 stage stages 144 KiB of weights) out
// of a 2.4 MB buffer shared by all workgroups (L2-resident), with all loads of an iteration issued before one wait + barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE 0: DMA   1: registers   2: half DMA, half registers   3: DMA, one instruction after every mfma*4/PER_WAVE MFMAs instead of all up
// front ; MFMA > 0: that many independent MFMAs per wave and iteration
template <int MODE, int PER_WAVE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t src_bytes, float* out, int iters, int mfma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  u32x4_t fa = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, fb = fa;
  size_t off = ((size_t)blockIdx.x * 7919u * 1024u) % src_bytes;
  u32x4_t keep = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4_t r[(MODE == 0 || MODE == 3) ? 1 : PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const size_t o = (off + (size_t)(i * 8 + wave) * 1024u) % src_bytes;
      const bool dma = MODE == 0 || (MODE == 2 && (i & 1) == 0);
      if (MODE == 3) continue;
      if (dma) __builtin_amdgcn_global_load_lds((gptr_t)(src + o + lane * 16), (lptr_t)(smem + (i * 8 + wave) * 1024), 16, 0, 0);
      else r[(MODE == 0 || MODE == 3) ? 0 : i] = *reinterpret_cast<const u32x4_t*>(src + o + lane * 16);
    }
    if (MODE == 3) {
      const int per = mfma / PER_WAVE;            // groups of 4 MFMAs between two DMA instructions
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i) {
        const size_t o = (off + (size_t)(i * 8 + wave) * 1024u) % src_bytes;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + o + lane * 16), (lptr_t)(smem + (i * 8 + wave) * 1024), 16, 0, 0);
        for (int m = 0; m < per; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc[q], 0, 0, 0);
        }
      }
    } else
    for (int m = 0; m < mfma; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc[q], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i)
        if (MODE == 1 || (i & 1)) *reinterpret_cast<u32x4_t*>(smem + (i * 8 + wave) * 1024 + lane * 16) = r[i];
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    keep[0] ^= *reinterpret_cast<const unsigned*>(smem + ((tid * 16 + it * 64) & 0xffff));
    off = (off + (size_t)PER_WAVE * 8 * 1024u) % src_bytes;
  }
  if (keep[0] == 0x12345678u || acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) out[tid] = acc[0][0];
}

template <int MODE, int PER_WAVE>
static void run(const char* name, const char* src, size_t bytes, float* out, int mfma) {
  auto k = fill_kernel<MODE, PER_WAVE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 400, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), PER_WAVE * 8 * 1024, 0, src, bytes, out, 20, mfma);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), PER_WAVE * 8 * 1024, 0, src, bytes, out, iters, mfma);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double kb = PER_WAVE * 8.0, us_it = ms * 1e3 / iters;
  printf("%-28s %3.0f KiB/iter  mfma/wave/iter %4d : %7.2f us/iter  %6.1f B/clk/CU (2.4 GHz)  %5.2f TB/s aggregate  MFMA-only time %.2f us\n",
         name, kb, mfma * 4, us_it, kb * 1024 / (us_it * 2400.0), kb * 1024 * grid / (us_it * 1e-6) / 1e12, mfma * 4 * 16 / 2400.0);
}

int main() {
  const size_t bytes = 2400 * 1024;
  char* src; float* out;
  hipMalloc(&src, bytes + 4096); hipMalloc(&out, 4096);
  hipMemset(src, 1, bytes + 4096);
  for (int mfma : {0, 72}) {      // 72 x 4 = 288 MFMAs per wave and iteration = one conv3x3_h16 stage
    run<0, 18>("DMA  (global_load_lds x4)", src, bytes, out, mfma);
    run<1, 18>("regs (global_load + ds_write)", src, bytes, out, mfma);
    run<2, 18>("half DMA, half regs", src, bytes, out, mfma);
  }
  run<3, 18>("DMA spread between MFMAs", src, bytes, out, 72);
  run<3, 18>("DMA spread, no MFMA", src, bytes, out, 0);
  run<0, 9>("DMA, 72 KiB", src, bytes, out, 72);
  run<0, 4>("DMA, 32 KiB", src, bytes, out, 72);
  return 0;
}

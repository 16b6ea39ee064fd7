// What does an instruction issued by the OTHER wave of a SIMD cost a wave that streams MFMAs?  (round 4)
// Workgroup = 8 waves: waves 0-3 (one per SIMD) issue N back-to-back independent v_mfma_f32_16x16x32_bf16 and time themselves with
// s_memtime; waves 4-7 (their SIMD partners) run a loop of ONE instruction type for longer than that.  The MFMA waves' elapsed cycles
// minus their time alone, divided by the number of partner instructions issued meanwhile, is the cost of one partner instruction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/partner_cost_bench.hip -o partner_cost_bench && ./partner_cost_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

enum { K_NONE, K_VADD, K_VFMA_PK, K_VCNDMASK, K_VDPP, K_VCVT, K_SALU, K_SNOP, K_DSREAD, K_DSWRITE, K_GLOAD, K_GSTORE, K_MFMA, K_COUNT };
static const char* NAMES[K_COUNT] = {"(partner idle)", "v_add_f32", "v_pk_fma_f32", "v_cndmask_b32", "v_add_f32_dpp", "v_cvt_pk_bf16_f32", "s_add_u32", "s_nop 0",
                                     "ds_read_b128", "ds_write_b128", "global_load_dwordx4 (L2 hit)", "global_store_dwordx4", "v_mfma (partner too)"};

__device__ unsigned long long g_out[8][4];

template <int KIND, bool SWAP>
__global__ __launch_bounds__(512) void k(const u32x4_t* gsrc, u32x4_t* gdst, int n_mfma_iters, int n_partner_iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  __shared__ volatile int done_flag;
  // SWAP: the MFMA stream runs on the YOUNGER half (hardware waves 4-7), the partner instructions on the older half, which wins arbitration
  const int hw_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = SWAP ? (hw_wave ^ 4) : hw_wave;
  if (threadIdx.x == 0) done_flag = 0;
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  if (wave < 4) {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(lds + lane * 16), b = *reinterpret_cast<const u32x4_t*>(lds + 1024 + lane * 16);
    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, a), bv = __builtin_bit_cast(bf16x8_t, b);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { g_out[wave][0] = t1 - t0; done_flag = 1; }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.678f) gdst[0] = u32x4_t{1, 2, 3, 4};
  } else {
    // partner: n_partner_iters x 32 instructions of one kind, independent operands; counts what it issued until the MFMA waves finish
    float v[32];
    u32x4_t q[8];
    for (int i = 0; i < 32; ++i) v[i] = (float)(lane + i);
    for (int i = 0; i < 8; ++i) q[i] = u32x4_t{(uint32_t)lane, 1u, 2u, 3u};
    unsigned issued = 0, at_done = 0;
    uint32_t sacc = 0;
    const unsigned long long p0 = __builtin_readcyclecounter();
    const char* lp = lds + 4096 + lane * 16;
    const u32x4_t* gp = gsrc + (blockIdx.x * 256 + (threadIdx.x & 255));
    for (int it = 0; it < n_partner_iters; ++it) {
      if constexpr (KIND == K_VADD) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
      } else if constexpr (KIND == K_VFMA_PK) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[i]))); asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[i]))); }
      } else if constexpr (KIND == K_VCNDMASK) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(v[i]));
      } else if constexpr (KIND == K_VDPP) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 16) & 31]));
      } else if constexpr (KIND == K_VCVT) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(v[i]) : "v"(v[(i + 16) & 31]));
      } else if constexpr (KIND == K_SALU) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
      } else if constexpr (KIND == K_SNOP) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("s_nop 0");
      } else if constexpr (KIND == K_DSREAD) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 7]) : "v"((uint32_t)(size_t)(__attribute__((address_space(3))) const char*)lp), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (KIND == K_DSWRITE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("ds_write_b128 %0, %1" : : "v"((uint32_t)(size_t)(__attribute__((address_space(3))) const char*)lp), "v"(q[i & 7]));
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (KIND == K_GLOAD) {
#pragma unroll
        for (int i = 0; i < 32; ++i) q[i & 7] = __builtin_nontemporal_load(gp + ((it * 32 + i) & 63) * 512);
      } else if constexpr (KIND == K_GSTORE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) gdst[(blockIdx.x * 512 + threadIdx.x) + (size_t)((it * 32 + i) & 63) * 262144] = q[i & 7];
      } else if constexpr (KIND == K_MFMA) {
        f32x4_t c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = f32x4_t{v[i], 0.f, 0.f, 0.f};
        const bf16x8_t av = __builtin_bit_cast(bf16x8_t, q[0]), bv = __builtin_bit_cast(bf16x8_t, q[1]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = c[i][0];
      }
      if constexpr (KIND != K_NONE) issued += 32;
      if (!at_done && done_flag) at_done = issued;
      if constexpr (KIND == K_NONE) { if (done_flag) break; __builtin_amdgcn_s_sleep(8); }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += (float)q[i][0];
    const unsigned long long p1 = __builtin_readcyclecounter();
    if (lane == 0) { g_out[wave][1] = at_done; g_out[wave][2] = issued; g_out[wave][3] = (unsigned long long)(s + sacc); g_out[wave][0] = p1 - p0; }
  }
}

template <int KIND, bool SWAP>
void run(const u32x4_t* gsrc, u32x4_t* gdst, double* alone) {
  const int iters = 2000;                       // 32 000 MFMAs per wave = 512 000 pipe cycles alone
  unsigned long long z[8][4] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(g_out), z, sizeof(z));
  hipLaunchKernelGGL((k<KIND, SWAP>), dim3(1), dim3(512), 0, 0, gsrc, gdst, iters, KIND == K_MFMA ? iters * 4 : iters * 40);
  hipDeviceSynchronize();
  unsigned long long o[8][4];
  hipMemcpyFromSymbol(o, HIP_SYMBOL(g_out), sizeof(o));
  double cyc = 0, part = 0;
  for (int w = 0; w < 4; ++w) { cyc += o[w][0] / 4.0; part += o[w + 4][1] / 4.0; }
  if (KIND == K_NONE) *alone = cyc;
  const double n_mfma = iters * 16.0;
  printf("%-30s MFMA wave: %8.0f cycles = %5.2f per MFMA", NAMES[KIND], cyc, cyc / n_mfma);
  if (KIND != K_NONE && part > 0) printf("  | partner issued %8.0f meanwhile: %+6.2f cycles per partner instruction (%.2f partner instr per MFMA)", part, (cyc - *alone) / part, part / n_mfma);
  printf("\n");
}

int main() {
  u32x4_t *gsrc, *gdst;
  hipMalloc(&gsrc, 64ull * 512 * 16 * 16); hipMalloc(&gdst, 64ull * 262144 * 16 + 4096);
  hipMemset(gsrc, 1, 64ull * 512 * 16 * 16);
  double alone = 0;
  printf("one workgroup on one CU; four waves (one per SIMD) stream 32 000 independent 16x16x32 bf16 MFMAs each, their SIMD partners loop over one instruction type\n");
  printf("== MFMA stream on the OLDER waves 0-3 (they win the issue arbitration), partner = waves 4-7\n");
  run<K_NONE, false>(gsrc, gdst, &alone);
  run<K_VADD, false>(gsrc, gdst, &alone); run<K_VFMA_PK, false>(gsrc, gdst, &alone); run<K_VCNDMASK, false>(gsrc, gdst, &alone); run<K_VDPP, false>(gsrc, gdst, &alone);
  run<K_VCVT, false>(gsrc, gdst, &alone); run<K_SALU, false>(gsrc, gdst, &alone); run<K_SNOP, false>(gsrc, gdst, &alone); run<K_DSREAD, false>(gsrc, gdst, &alone);
  run<K_DSWRITE, false>(gsrc, gdst, &alone); run<K_GLOAD, false>(gsrc, gdst, &alone); run<K_GSTORE, false>(gsrc, gdst, &alone); run<K_MFMA, false>(gsrc, gdst, &alone);
  printf("== MFMA stream on the YOUNGER waves 4-7, partner = waves 0-3 (the partner wins the arbitration)\n");
  run<K_NONE, true>(gsrc, gdst, &alone);
  run<K_VADD, true>(gsrc, gdst, &alone); run<K_VFMA_PK, true>(gsrc, gdst, &alone); run<K_VCNDMASK, true>(gsrc, gdst, &alone); run<K_VDPP, true>(gsrc, gdst, &alone);
  run<K_VCVT, true>(gsrc, gdst, &alone); run<K_SALU, true>(gsrc, gdst, &alone); run<K_SNOP, true>(gsrc, gdst, &alone); run<K_DSREAD, true>(gsrc, gdst, &alone);
  run<K_DSWRITE, true>(gsrc, gdst, &alone); run<K_GLOAD, true>(gsrc, gdst, &alone); run<K_GSTORE, true>(gsrc, gdst, &alone); run<K_MFMA, true>(gsrc, gdst, &alone);
  return 0;
}

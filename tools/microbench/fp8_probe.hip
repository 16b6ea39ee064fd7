// probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x e4m3, unit MX scales) and v_cvt_pk_fp8_f32 on gfx950:
//   (1) operand layout: lane (i = l & 15, g = l >> 4) supplies k = 32 g .. 32 g + 31 of row i (A) / column i (B)?
//   (2) cvt rounding / saturation / OCP e4m3 decoding
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/fp8_probe.hip -o tools/microbench/fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* A, const uint8_t* B, float* D) {
  int l = threadIdx.x, i = l & 15, g = l >> 4;
  v8i a, b;
  for (int e = 0; e < 8; ++e) { a[e] = ((const int*)(A + i * 128 + 32 * g))[e]; b[e] = ((const int*)(B + i * 128 + 32 * g))[e]; }
  f32x4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];      // D[row = kout i of A][col = pixel of B]
}
__global__ void cvt(const float* x, uint32_t* y, int n) {
  int i = threadIdx.x;
  if (i < n) y[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
}
static float e4m3(uint8_t v) {           // OCP e4m3fn
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 15 && m == 7) f = NAN;
  else if (e == 0) f = ldexpf((float)m, -9);
  else f = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = rand() & 0x7f; if ((hA[i] & 0x7f) == 0x7f) hA[i] = 0x30; if (rand() & 1) hA[i] |= 0x80;
                                       hB[i] = rand() & 0x77; if (rand() & 1) hB[i] |= 0x80; }
  uint8_t *dA, *dB; float* dD; float hD[256];
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0;
    for (int kk = 0; kk < 128; ++kk) s += (double)e4m3(hA[i * 128 + kk]) * e4m3(hB[j * 128 + kk]);
    worst = fmax(worst, fabs(s - hD[i * 16 + j])); scale = fmax(scale, fabs(s));
  }
  printf("mfma 16x16x128 fp8: max |D - ref| = %.3e (max |ref| %.3e) -> layout %s\n", worst, scale, worst <= 1e-4 * scale ? "CONFIRMED" : "WRONG");
  float hx[16] = {0.1f, 1.0f, 447.0f, 448.0f, 449.0f, 500.0f, 1e6f, -1e6f, 0.0019f, 0.001f, 0.0009f, 3.3f, -0.07f, 17.0f, 464.0f, 480.0f};
  float* dx; uint32_t* dy; uint32_t hy[8];
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dy, sizeof(hy));
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dy, 8);
  hipMemcpy(hy, dy, sizeof(hy), hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) { uint8_t b = (hy[i / 2] >> (8 * (i & 1))) & 0xff; printf("  cvt %12.5g -> 0x%02x = %g\n", hx[i], b, e4m3(b)); }
  return 0;
}

// Prototype of the "one wave per SIMD, weights straight into registers, no barrier in the tap loop" 3x3 / 1 conv (round-4 review item 3)
// against conv3x3_h16 on the same tensors: correctness (bit-for-bit is not expected: another accumulation order) and time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/r4_conv_bench.hip -o r4_conv_bench ;  ./r4_conv_bench [N] [H=W] [C=K]
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../ssl_cr_histo_amd/csrc/conv_h16.hip"
#include "conv_r4_proto.hip"
namespace sslcr {
int conv_halo256_mode(int, const ConvArgs&) { return 16; }
bool conv_pp64_ok(int, const ConvArgs&) { return false; }
int device_cus() { return 256; }
hipError_t launch_conv_pp64(const ConvArgs&, hipStream_t) { return hipErrorInvalidValue; }
const char* conv_pp64_name(const ConvArgs&) { return ""; }
int conv_pp64_rows(const ConvArgs&) { return 0; }
}
using namespace sslcr;

static float bf2f_h(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 128;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;       // 0 plain, 1 bias + residual + relu
  const size_t elems = (size_t)N * H * H * C;
  uint16_t *x, *y, *y2, *r, *w;
  float* vec;
  hipMalloc(&x, elems * 2); hipMalloc(&y, elems * 2); hipMalloc(&y2, elems * 2); hipMalloc(&r, elems * 2); hipMalloc(&w, (size_t)C * 9 * C * 2);
  hipMalloc(&vec, 4 * C * 4);
  std::vector<uint16_t> hx(1 << 20);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (rand() & 1) ? 0 : (uint16_t)(0x3c00 + (rand() & 0x3ff));
  for (size_t o = 0; o < elems; o += hx.size()) hipMemcpy(x + o, hx.data(), (elems - o < hx.size() ? elems - o : hx.size()) * 2, hipMemcpyHostToDevice);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (uint16_t)(0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (size_t o = 0; o < elems; o += hx.size()) hipMemcpy(r + o, hx.data(), (elems - o < hx.size() ? elems - o : hx.size()) * 2, hipMemcpyHostToDevice);
  for (size_t o = 0; o < (size_t)C * 9 * C; o += hx.size()) hipMemcpy(w + o, hx.data(), ((size_t)C * 9 * C - o < hx.size() ? (size_t)C * 9 * C - o : hx.size()) * 2, hipMemcpyHostToDevice);
  std::vector<float> hv(4 * C);
  for (auto& v : hv) v = 0.01f * (rand() % 100 - 50);
  hipMemcpy(vec, hv.data(), 4 * C * 4, hipMemcpyHostToDevice);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = H; a.C = C; a.K = C; a.R = 3; a.S = 3; a.stride = 1; a.pad = 1;
  a.PH = H; a.PW = H; a.OH = H; a.OW = H; a.osh = 1;
  if (mode == 1) { a.bias = vec; a.residual = r; a.relu = 1; }
  ConvArgs b = a; b.y = y2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 3; ++i) { if (k == 0) launch_conv_h16(DT_BF16, a, 0); else launch_conv_r4(b, 0); }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) { if (k == 0) launch_conv_h16(DT_BF16, a, 0); else launch_conv_r4(b, 0); }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[k], e0, e1);
  }
  hipError_t err = hipGetLastError();
  std::vector<uint16_t> o1(1 << 22), o2(1 << 22);
  const size_t ncmp = elems < o1.size() ? elems : o1.size();
  hipMemcpy(o1.data(), y, ncmp * 2, hipMemcpyDeviceToHost);
  hipMemcpy(o2.data(), y2, ncmp * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxv = 0; size_t nbad = 0;
  for (size_t i = 0; i < ncmp; ++i) {
    const double p = bf2f_h(o1[i]), q = bf2f_h(o2[i]);
    maxv = fmax(maxv, fabs(p));
    const double e = fabs(p - q);
    if (e > maxerr) maxerr = e;
    if (e > 0.02 * (fabs(p) + 1)) ++nbad;
  }
  const double fl = 2.0 * elems * C * 9;
  printf("N=%d %dx%d C=K=%d mode=%d: h16 %.1f us (%.0f TF/s) | r4 %.1f us (%.0f TF/s) | max |diff| %.4g of max %.4g, %zu of %zu off by > 2%% (%s)\n", N, H, H, C, mode,
         ms[0] * 100, fl / (ms[0] * 1e-4) / 1e12, ms[1] * 100, fl / (ms[1] * 1e-4) / 1e12, maxerr, maxv, nbad, ncmp, hipGetErrorString(err));
  return 0;
}

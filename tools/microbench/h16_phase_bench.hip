// Where does a stage's time go in the persistent 16x16-tile conv (csrc/conv_h16.hip, the step's dominant kernel)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSLCR_H16_PROF tools/microbench/h16_phase_bench.hip -o h16_phase_bench
//   ./h16_phase_bench [N] [H=W] [C=K] [op: 0 plain + stats, 1 bias + residual + relu, 3 prologue + stats]
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../ssl_cr_histo_amd/csrc/conv_h16.hip"
namespace sslcr {                                  // what conv_h16.hip takes from its neighbours
int conv_halo256_mode(int, const ConvArgs&) { return 16; }
bool conv_pp64_ok(int, const ConvArgs&) { return false; }
int device_cus() { return 256; }
hipError_t launch_conv_pp64(const ConvArgs&, hipStream_t) { return hipErrorInvalidValue; }
const char* conv_pp64_name(const ConvArgs&) { return ""; }
}
using namespace sslcr;

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 128;
  const int op = argc > 4 ? atoi(argv[4]) : 0;
  // data: 0 random bf16 around +-1 (dense), 1 the same with half of the INPUT values zero (what a ReLU leaves), 2 all input values equal
  const int data = argc > 5 ? atoi(argv[5]) : 0;
  const size_t elems = (size_t)N * H * H * C;
  uint16_t *x, *y, *r, *w;
  float *stats, *vec;
  hipMalloc(&x, elems * 2); hipMalloc(&y, elems * 2); hipMalloc(&r, elems * 2); hipMalloc(&w, (size_t)C * 9 * C * 2);
  hipMalloc(&stats, 256 * 4 * 2 * C * 4 * 2); hipMalloc(&vec, 4 * C * 4);
  std::vector<uint16_t> hx(1 << 20);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));     // random bf16 around +-1
  std::vector<uint16_t> hw(hx);                                    // weights stay dense random
  if (data == 1) for (size_t i = 0; i < hx.size(); ++i) if (rand() & 1) hx[i] = 0;
  if (data == 2) for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3f80;
  for (size_t o = 0; o < elems; o += hx.size()) {
    const size_t n = elems - o < hx.size() ? elems - o : hx.size();
    hipMemcpy(x + o, hx.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(r + o, hx.data(), n * 2, hipMemcpyHostToDevice);
  }
  for (size_t o = 0; o < (size_t)C * 9 * C; o += hx.size()) {
    const size_t n = (size_t)C * 9 * C - o < hx.size() ? (size_t)C * 9 * C - o : hx.size();
    hipMemcpy(w + o, hw.data(), n * 2, hipMemcpyHostToDevice);
  }
  std::vector<float> hv(4 * C, 0.5f);
  hipMemcpy(vec, hv.data(), 4 * C * 4, hipMemcpyHostToDevice);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = H; a.C = C; a.K = C; a.R = 3; a.S = 3; a.stride = 1; a.pad = 1;
  a.PH = H; a.PW = H; a.OH = H; a.OW = H; a.osh = 1;
  if (op == 0) a.stats = stats;
  if (op == 1) { a.residual = r; a.bias = vec; a.relu = 1; }
  if (op == 3) { a.in_scale = vec; a.in_shift = vec + C; a.in_relu = 1; a.stats = stats; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_conv_h16(DT_BF16, a, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch_conv_h16(DT_BF16, a, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long prof[16][8];
  hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_h16_prof), sizeof(prof));
  // implied shader clock: workgroup 0's cycle count over the launch's duration (the launch is one persistent walk per workgroup)
  printf("N=%d %dx%d C=K=%d op=%d data=%d: %.1f us/launch, %.1f TF/s, %.2f GHz implied (MFMA-only floor of a stage: 2 waves x 288 MFMAs x 16 cycles = 9216 cycles per SIMD)\n",
         N, H, H, C, op, data, ms * 1e3 / reps, 2.0 * elems * C * 9 / (ms / reps * 1e-3) / 1e12, (double)prof[0][6] / (ms * 1e3 / reps) * 1e-3);
  for (int wv = 0; wv < 8; wv += 3) {
    const double st = (double)prof[wv][5];
    printf("  wave %d: %4.0f stages, per stage %7.0f cycles: weight-DMA wait %6.0f, P barriers %6.0f, F barriers %6.0f, halo swap %6.0f, epilogue %6.0f\n", wv, st,
           prof[wv][6] / st, prof[wv][0] / st, prof[wv][1] / st, prof[wv][2] / st, prof[wv][3] / st, prof[wv][4] / st);
  }
  return 0;
}

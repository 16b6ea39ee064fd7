// Where does a tap's time go in the plane-gather stride-2 conv (csrc/conv_s2.hip)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSLCR_S2_PROF tools/microbench/s2_phase_bench.hip -o s2_phase_bench
//   ./s2_phase_bench [N] [H=W] [C] [K] [pair 0/1] [eval 0/1]
#include <cstdio>
#include <cstring>
#include <vector>
// the product kernel (its SSLCR_S2_PROF phase counters); -DS2_ABL_COPY: the round-5 measurement copy that carries the timing-only S2_ABL_* switches
#ifdef S2_ABL_COPY
#include "conv_s2_abl.hip"
#else
#include "../../ssl_cr_histo_amd/csrc/conv_s2.hip"
#endif
namespace sslcr {
int device_cus() { return 256; }
}
using namespace sslcr;

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 128;
  const int K = argc > 4 ? atoi(argv[4]) : 256, pair = argc > 5 ? atoi(argv[5]) : 1, eval = argc > 6 ? atoi(argv[6]) : 0;
  const size_t xe = (size_t)N * H * H * C, ye = (size_t)N * (H / 2) * (H / 2) * K;
  uint16_t *x, *y, *yd, *w, *wd;
  float *stats, *statsd, *vec;
  hipMalloc(&x, xe * 2); hipMalloc(&y, ye * 2); hipMalloc(&yd, ye * 2); hipMalloc(&w, (size_t)K * 9 * C * 2); hipMalloc(&wd, (size_t)K * C * 2);
  const size_t rows = (size_t)N * (H / 32) * (H / 32) * 4;
  hipMalloc(&stats, rows * 2 * K * 4); hipMalloc(&statsd, rows * 2 * K * 4); hipMalloc(&vec, K * 4);
  std::vector<uint16_t> hx(1 << 20);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (rand() & 1) ? 0 : (uint16_t)(0x3c00 + (rand() & 0x3ff));     // what a ReLU leaves
  for (size_t o = 0; o < xe; o += hx.size()) hipMemcpy(x + o, hx.data(), (xe - o < hx.size() ? xe - o : hx.size()) * 2, hipMemcpyHostToDevice);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (size_t o = 0; o < (size_t)K * 9 * C; o += hx.size()) hipMemcpy(w + o, hx.data(), ((size_t)K * 9 * C - o < hx.size() ? (size_t)K * 9 * C - o : hx.size()) * 2, hipMemcpyHostToDevice);
  hipMemcpy(wd, hx.data(), (size_t)K * C * 2, hipMemcpyHostToDevice);
  std::vector<float> hv(K, 0.5f);
  hipMemcpy(vec, hv.data(), K * 4, hipMemcpyHostToDevice);
  ConvArgs a, d;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = H; a.C = C; a.K = K; a.R = 3; a.S = 3; a.stride = 2; a.pad = 1;
  a.PH = H / 2; a.PW = H / 2; a.OH = H / 2; a.OW = H / 2; a.osh = 1;
  d = a; d.w = wd; d.y = yd; d.R = 1; d.S = 1; d.pad = 0;
  if (eval) { a.bias = vec; d.bias = vec; a.relu = 1; } else { a.stats = stats; d.stats = statsd; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_conv_s2(a, pair ? &d : nullptr, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch_conv_s2(a, pair ? &d : nullptr, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long prof[8][8];
  hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_s2_prof), sizeof(prof));
  const double fl = 2.0 * ye * C * (pair ? 10 : 9);
  const int taps = (pair ? 10 : 9) * (C / 64);
  printf("N=%d %dx%d C=%d K=%d pair=%d eval=%d: %.1f us/launch, %.1f TF/s, %.2f GHz implied (MFMA-only floor of a tap: 2 waves x 32 MFMAs x 16 cycles = 1024 cycles per SIMD)\n",
         N, H, H, C, K, pair, eval, ms * 1e3 / reps, fl / (ms / reps * 1e-3) / 1e12, (double)prof[0][4] / (ms * 1e3 / reps) * 1e-3);
  for (int wv = 0; wv < 8; ++wv) {
    const double it = (double)prof[wv][3];
    printf("  wave %d (%s): %4.0f items, per item %7.0f cycles = %5.0f per tap: DMA wait %6.0f, barrier %6.0f, output stage %6.0f per item\n", wv, wv < 4 ? "planes " : "weights", it,
           prof[wv][4] / it, prof[wv][4] / it / taps, prof[wv][0] / it, prof[wv][1] / it, prof[wv][2] / it);
  }
  return 0;
}

// Where does a tile's time go in the ping-pong layer1 conv (csrc/conv_pp64.hip)?  Per wave of workgroup 0: shader cycles in the M
// phase, at the barrier behind it, in the W phase, at the barrier behind that.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSLCR_PP_PROF tools/microbench/pp64_phase_bench.hip -o pp64_phase_bench
#include "../../ssl_cr_histo_amd/csrc/conv_pp64.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace sslcr { int device_cus() { return 256; } }
using namespace sslcr;

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 640, H = 64, W = 64;
  const int op = argc > 2 ? atoi(argv[2]) : 0;     // 0 plain+stats, 1 residual+relu+bias, 2 mask, 3 prologue+stats
  const size_t elems = (size_t)N * H * W * 64;
  uint16_t *x, *y, *r, *w;
  float *stats, *vec;
  hipMalloc(&x, elems * 2); hipMalloc(&y, elems * 2); hipMalloc(&r, elems * 2); hipMalloc(&w, 64 * 9 * 64 * 2);
  hipMalloc(&stats, 256 * 8 * 2 * 64 * 4); hipMalloc(&vec, 4 * 64 * 4);
  std::vector<uint16_t> hx(1 << 20);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));     // random bf16 around +-1
  for (size_t o = 0; o < elems; o += hx.size()) {
    const size_t n = elems - o < hx.size() ? elems - o : hx.size();
    hipMemcpy(x + o, hx.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(r + o, hx.data(), n * 2, hipMemcpyHostToDevice);
  }
  hipMemcpy(w, hx.data(), 64 * 9 * 64 * 2, hipMemcpyHostToDevice);
  std::vector<float> hv(256, 0.5f);
  hipMemcpy(vec, hv.data(), 256 * 4, hipMemcpyHostToDevice);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = W; a.C = 64; a.K = 64; a.R = 3; a.S = 3; a.stride = 1; a.pad = 1;
  a.PH = H; a.PW = W; a.OH = H; a.OW = W; a.osh = 1;
  if (op == 0) a.stats = stats;
  if (op == 1) { a.residual = r; a.bias = vec; a.relu = 1; }
  if (op == 2) { a.mask_x = r; a.mask_scale = vec; a.mask_shift = vec + 64; a.mask_mean = vec + 128; a.stats = stats; }
  if (op == 3) { a.in_scale = vec; a.in_shift = vec + 64; a.in_relu = 1; a.stats = stats; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_conv_pp64(a, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch_conv_pp64(a, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long prof[8][8];
  hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_pp_prof), sizeof(prof));
  const int tiles = N * (H / 16) * (W / 16);
  const double stages = (double)((tiles + 511) / 512);
  printf("N=%d op=%d: %.1f us/launch, %.1f TF/s; stages per group %.0f\n", N, op, ms * 1e3 / reps, 2.0 * elems * 64 * 9 / (ms / reps * 1e-3) / 1e12, stages);
  for (int wv = 0; wv < 8; ++wv)
    printf("  wave %d (group %d): per stage  M %7.0f  wait %7.0f  W %7.0f (halo xform+store %6.0f, output stage %6.0f, rest %6.0f)  wait %7.0f cycles\n", wv, wv >> 2,
           prof[wv][0] / stages, prof[wv][1] / stages, prof[wv][2] / stages, prof[wv][4] / stages, prof[wv][5] / stages,
           (prof[wv][2] - prof[wv][4] - prof[wv][5]) / stages, prof[wv][3] / stages);
  return 0;
}

// What does rocprofv3's FETCH_SIZE report for the read patterns of the conv kernels?  MI355X_MICROARCH.md calibrates ONE pattern
// (wide coalesced streaming reads, 16 B per lane: the counter shows half the bytes) and says to calibrate any other on a known byte
// count.  Each kernel below reads a known number of bytes from a 2 GiB buffer (8x the 256 MiB Infinity Cache) exactly once:
//   stream16      1 KiB contiguous per wave-load (16 B per lane)                                  -- the guide's pattern
//   half128       128 B pieces at 256 B stride, 8 pieces per wave-load (16 B per lane): a C=128 tensor's 64-channel slab, the
//                 halo loads of conv3x3_h16<128> and the pixel rows of conv_dma at C >= 128
//   half128_both  the same, first halves then second halves in one launch (both slabs of a tile, far apart in time)
//   row128        128 B rows, contiguous (a C=64 tensor): 8 rows per wave-load = the same 1 KiB, as the halo loader addresses it
//   halo18        18x18-pixel halos of 16x16 tiles of a 32x32 x (256 B per pixel) image, first 128 B of each pixel, tiles walked
//                 like conv3x3_h16 does (XCD x takes a contiguous run of tiles): 1.27x the tensor's half if no overlap is caught
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fc -- tools/microbench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sink(u32x4 v, unsigned* out) {
  if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x12345678u) *out = 1;      // never true for the zero-filled buffer; keeps the loads
}

__global__ __launch_bounds__(256) void stream16(const char* p, size_t bytes, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16) acc ^= *(const u32x4*)(p + o);
  sink(acc, out);
}
// record = 256 B; thread t of the grid-wide index reads 16 B chunk (t & 7) of half `half` of record t >> 3
__global__ __launch_bounds__(256) void half128(const char* p, size_t records, int half, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < records * 8; t += (size_t)gridDim.x * 256)
    acc ^= *(const u32x4*)(p + (t >> 3) * 256 + half * 128 + (t & 7) * 16);
  sink(acc, out);
}
__global__ __launch_bounds__(256) void half128_both(const char* p, size_t records, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (int half = 0; half < 2; ++half)
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < records * 8; t += (size_t)gridDim.x * 256)
      acc ^= *(const u32x4*)(p + (t >> 3) * 256 + half * 128 + (t & 7) * 16);
  sink(acc, out);
}
__global__ __launch_bounds__(256) void row128(const char* p, size_t rows, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < rows * 8; t += (size_t)gridDim.x * 256)
    acc ^= *(const u32x4*)(p + (t >> 3) * 128 + (t & 7) * 16);
  sink(acc, out);
}
// images of 32x32 pixels x 256 B; item = (image, tile row, tile column) of 16x16 tiles; a workgroup (512 threads) loads the
// in-image part of the 18x18 halo, 128 B per pixel, then takes item + gridDim.x (XCD-major first item like conv3x3_h16)
__global__ __launch_bounds__(512) void halo18(const char* p, int images, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const int G = gridDim.x, n_items = images * 4;
  const int first = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  for (int item = first; item < n_items; item += G) {
    const int n = item >> 2, h0 = ((item >> 1) & 1) * 16, w0 = (item & 1) * 16;
    for (int i = threadIdx.x; i < 18 * 18 * 8; i += 512) {
      const int px = i >> 3, hh = h0 - 1 + px / 18, ww = w0 - 1 + px % 18;
      if (hh >= 0 && hh < 32 && ww >= 0 && ww < 32) acc ^= *(const u32x4*)(p + ((size_t)(n * 32 + hh) * 32 + ww) * 256 + (i & 7) * 16);
    }
  }
  sink(acc, out);
}

int main() {
  const size_t bytes = 2ull << 30;
  char* p; unsigned* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(p, 0, bytes); hipMemset(out, 0, 4);
  const int grid = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](const char* name, double mib, auto launch) {
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s reads %8.1f MiB  %7.1f us  %5.2f TB/s\n", name, mib, ms * 1e3, mib * 1048576.0 / (ms * 1e-3) / 1e12);
  };
  const size_t rec = bytes / 256;
  timed("stream16", bytes / 1048576.0, [&] { hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, p, bytes, out); });
  timed("half128", rec * 128 / 1048576.0, [&] { hipLaunchKernelGGL(half128, dim3(grid), dim3(256), 0, 0, p, rec, 0, out); });
  timed("half128_both", rec * 256 / 1048576.0, [&] { hipLaunchKernelGGL(half128_both, dim3(grid), dim3(256), 0, 0, p, rec, out); });
  timed("row128", bytes / 1048576.0, [&] { hipLaunchKernelGGL(row128, dim3(grid), dim3(256), 0, 0, p, bytes / 128, out); });
  const int images = (int)(bytes / (32 * 32 * 256));
  timed("halo18", images * (4.0 * 17 * 17) * 128 / 1048576.0, [&] { hipLaunchKernelGGL(halo18, dim3(256), dim3(512), 0, 0, p, images, out); });
  printf("halo18: tensor half = %.1f MiB, halo requests = %.1f MiB (17x17 in-image pixels per tile)\n", images * 1024.0 * 128 / 1048576.0,
         images * (4.0 * 17 * 17) * 128 / 1048576.0);
  return 0;
}

// Eval-mode ResNet18 stem in ONE kernel: uint8/fp32 NCHW ingestion -> conv1 7x7/2 (BatchNorm folded: bias) -> ReLU -> maxpool 3x3/2
// pad 1, bf16 out.  What the teacher forward, validate() and the WSI inference run (models/net.py:32,77 reached in eval mode from
// eval_BreastPathQ_SSL_CR.py:43-44,77-79, test_Camelyon16.py:30-70).  The two-kernel form (stem.hip + the plain max-pool) writes the
// 128x128x64 conv output (2.1 MB per image in bf16: 940 MB for the benchmark's 448 teacher images) and reads it straight back;
// here it never leaves the CU.
//
// A workgroup (8 waves) takes whole images.  It walks an image in 16x16-pixel tiles of conv outputs, band by band (top to bottom),
// left to right inside a band; the MFMA part of a tile is stem_fwd_kernel's (same packed weights, same K order, same input halo
// with the next tile's bytes requested before this tile's MFMAs).  The output stage adds the bias, clamps, packs to bf16 and writes
// the tile into an LDS plane P[17][17] whose row 0 / column 0 hold the conv outputs just above / left of the tile: column 0 is the
// previous tile's last column, row 0 comes from a band-wide row buffer R that every tile refreshes with its own last row -- so a
// max-pool window that straddles tiles finds all nine inputs in LDS and no conv output is computed twice.  512 threads then pool
// the tile's 8x8 windows x 8 sixteen-byte channel chunks, one each: nine ds_read_b128 and 32 packed signed 16-bit maxima (post-
// ReLU bf16 values order like integers; signed, so that a -0 cannot win), one 16-byte store.  P is double-buffered: while the
// pooling threads read this tile's plane, the borders of the next tile's plane are filled in -- ONE barrier per tile.
//
// Result: bit-identical to sslcr_stem_conv (bias, relu) followed by sslcr_bn_relu_maxpool (scale = shift = NULL): the same fp32
// accumulation order, the same rounding to bf16 before the maximum.  Zero padding of the pool is a 0 in P (all inputs are >= 0).
#include <stdlib.h>

#include "kernels.hpp"

namespace sslcr {

namespace {
constexpr int SP_TH = 16, SP_TW = 16;                // conv-output tile
constexpr int SP_HR = 2 * SP_TH + 5;                 // 37 input halo rows
constexpr int SP_HC = 2 * SP_TW + 6;                 // 38 input halo columns (even; covers the zero-weight tap s = 7)
constexpr int SP_NT = 512;
constexpr int SP_WROW = 224 * 2;                     // bytes per packed weight row (bf16), unpadded + swizzled as in stem_fwd_kernel
constexpr int SP_PP = 17 * 17 * 128;                 // one plane P: [17 rows][17 cols][64 ch bf16]
// kout owned by MFMA tile t, accumulator row group q, element j (stem.hip STEM_CH): two 8-channel runs 32 channels apart
#define SP_CH(t, q, j) ((((t) >> 1) * 32) + ((q) * 8) + (((t) & 1) * 4) + (j))

typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct SpRaw { uint32_t d[3]; };

__device__ __forceinline__ const void* sp_seg(const StemArgs& a, int& n) {
  if (a.x2 && n >= a.n_split) { n -= a.n_split; return a.x2; }
  return a.x;
}
// uint8 fast path (W % 4 == 0): thread (row rr = tid / 10, dword d = tid % 10) of the first 370 loads one aligned dword per colour plane
__device__ __forceinline__ SpRaw sp_issue4(const void* xv, int n, int H, int W, int hi0, int wi0, int tid) {
  SpRaw r{{0u, 0u, 0u}};
  if (tid < SP_HR * 10) {
    const int rr = tid / 10, d = tid - rr * 10;
    const int h = hi0 + rr, w = wi0 - 1 + 4 * d;
    if (h >= 0 && h < H && w >= 0 && w < W) {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(xv) + ((size_t)(n * 3) * H + h) * W + w;
      const size_t plane = (size_t)H * W;
#pragma unroll
      for (int c = 0; c < 3; ++c) r.d[c] = *reinterpret_cast<const uint32_t*>(p + c * plane);
    }
  }
  return r;
}
__device__ __forceinline__ void sp_commit4(bf16_t* halo, const SpRaw& r, int tid) {
  if (tid >= SP_HR * 10) return;
  const int rr = tid / 10, d = tid - rr * 10;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = 4 * d - 1 + j;
    if (cc < 0 || cc >= SP_HC) continue;
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = (float)((r.d[c] >> (8 * j)) & 0xffu);
    // 0..255 are exact in bf16: the upper half of the fp32 pattern
    const uint32_t lo = (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
    const uint32_t hi = __float_as_uint(f[2]) >> 16;
    *reinterpret_cast<u32x2_t*>(halo + (rr * SP_HC + cc) * 4) = u32x2_t{lo, hi};
  }
}
// generic path (fp32 input, odd widths): element by element, synchronous
template <bool INF32>
__device__ __forceinline__ void sp_load_halo(bf16_t* halo, const void* xv, int n, int H, int W, int hi0, int wi0, int tid) {
  for (int idx = tid; idx < 3 * SP_HR * SP_HC; idx += SP_NT) {
    const int c = idx / (SP_HR * SP_HC), rem = idx - c * (SP_HR * SP_HC);
    const int rr = rem / SP_HC, cc = rem - rr * SP_HC;
    const int h = hi0 + rr, w = wi0 + cc;
    float v = 0.f;
    if (h >= 0 && w >= 0 && h < H && w < W) {
      const size_t o = ((size_t)(n * 3 + c) * H + h) * W + w;
      v = INF32 ? reinterpret_cast<const float*>(xv)[o] : (float)reinterpret_cast<const uint8_t*>(xv)[o];
    }
    halo[(rr * SP_HC + cc) * 4 + c] = f2bf(v);
  }
}
// byte offset of (row, col, 16-byte chunk) inside a plane: the chunk index is XORed with (col & 7), so that the 16 lanes of an
// output-stage store (consecutive columns, one chunk) and of a pooling read hit 16 different bank groups
__device__ __forceinline__ int sp_off(int row, int col, int chunk) { return (row * 17 + col) * 128 + ((chunk ^ (col & 7)) << 4); }
}  // namespace

template <bool INF32>
__global__ __launch_bounds__(SP_NT) void stem_pool_fwd_kernel(const StemArgs a, int POH, int POW) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_lds = smem;                                                   // [64][448 B]
  T* halo0 = reinterpret_cast<T*>(smem + 64 * SP_WROW);                  // two input halos of SP_HR * SP_HC * 4 elements
  constexpr int HALO_B = SP_HR * SP_HC * 4 * 2;
  char* P0 = smem + 64 * SP_WROW + 2 * HALO_B;                           // two planes
  char* Rrow = P0 + 2 * SP_PP;                                           // [OW columns][128 B]: the last conv row of the band above

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int bands = a.OH / SP_TH, tcols = a.OW / SP_TW;
  const int tiles_img = bands * tcols;
  const bool fast = !INF32 && (a.W & 3) == 0;

  {   // packed weights [64][224] once; zero the halos (4th channel stays 0), both planes and the row buffer
    const char* wg = reinterpret_cast<const char*>(a.w);
    constexpr int CH = 224 * 2 / 16;
    for (int i = tid; i < 64 * CH; i += SP_NT) {
      const int k = i / CH, c = i - k * CH;
      st16(w_lds + k * SP_WROW + (c ^ ((k >> 3) & 2)) * 16, ld16(wg + (size_t)i * 16));
    }
    const u32x4_t z{0u, 0u, 0u, 0u};
    for (int i = tid; i < (2 * HALO_B + 2 * SP_PP + a.OW * 128) / 16; i += SP_NT) st16(smem + 64 * SP_WROW + i * 16, z);
  }
  float bias[16], osc[16];                   // osc: sslcr_stem_desc.out_scale (1 where the BatchNorm scale sits in the filters: x * 1 + b is x + b)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    bias[j] = a.bias ? a.bias[SP_CH(j >> 2, g, j & 3)] : 0.f;
    osc[j] = a.out_scale ? a.out_scale[SP_CH(j >> 2, g, j & 3)] : 1.f;
  }
  __syncthreads();

  // the walk: item = image * tiles_img + band * tcols + tile column, this workgroup's images blockIdx.x, + gridDim.x, ...
  const int n_first = blockIdx.x;
  if (n_first >= a.N) return;
  auto origin = [&](int item, int& n, int& tb, int& tc) {
    const int k = item / tiles_img, rem = item - k * tiles_img;
    n = n_first + k * (int)gridDim.x;
    tb = rem / tcols; tc = rem - tb * tcols;
  };
  const int n_imgs = (a.N - n_first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_items = n_imgs * tiles_img;
  SpRaw raw{{0u, 0u, 0u}};
  int cur = 0;
  {
    int n, tb, tc;
    origin(0, n, tb, tc);
    int ns = n;
    const void* xseg = sp_seg(a, ns);
    if (fast) { raw = sp_issue4(xseg, ns, a.H, a.W, 2 * tb * SP_TH - 3, 2 * tc * SP_TW - 3, tid); sp_commit4(halo0, raw, tid); }
    else sp_load_halo<INF32>(halo0, xseg, ns, a.H, a.W, 2 * tb * SP_TH - 3, 2 * tc * SP_TW - 3, tid);
  }
  __syncthreads();

  for (int item = 0; item < n_items; ++item) {
    int n, tb, tc;
    origin(item, n, tb, tc);
    const T* halo = halo0 + cur * (SP_HR * SP_HC * 4);
    char* P = P0 + (item & 1) * SP_PP;
    char* Pn = P0 + ((item + 1) & 1) * SP_PP;
    const bool more = item + 1 < n_items;
    int nn = 0, ntb = 0, ntc = 0;
    if (more) {
      origin(item + 1, nn, ntb, ntc);
      if (fast) {
        int ns = nn;
        const void* xseg = sp_seg(a, ns);
        raw = sp_issue4(xseg, ns, a.H, a.W, 2 * ntb * SP_TH - 3, 2 * ntc * SP_TW - 3, tid);
      }
    }
    // ---- conv tile: wave w owns output rows 2w, 2w+1 (x 16 columns x 64 kouts); K = (r, s8, c4) as in stem_fwd_kernel
    f32x4_t acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      u32x4_t af[4], bfr[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = SP_CH(t, li >> 2, li & 3);
        af[t] = ld16(w_lds + k * SP_WROW + (r * 8 + 2 * (g ^ ((li >> 3) << 1))) * 4 * 2);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int hr = 2 * (2 * wave + p) + r;
        bfr[p] = ld16(reinterpret_cast<const char*>(halo) + ((hr * SP_HC + 2 * li + 2 * g) * 4) * 2);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[t]), __builtin_bit_cast(bf16x8_t, bfr[p]), acc[t][p], 0, 0, 0);
    }
    // ---- output stage: bias, clamp, bf16 -> plane rows 1..16, columns 1..16
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = fmaf(acc[t][p][j], osc[t * 4 + j], bias[t * 4 + j]);
          v[t * 4 + j] = a.relu ? relu0(o) : o;
        }
      const int prow = 1 + 2 * wave + p, pcol = 1 + li;
#pragma unroll
      for (int q = 0; q < 2; ++q) st16(P + sp_off(prow, pcol, q * 4 + g), PackH<T>::run(v + q * 8));
    }
    if (more) {
      if (fast) sp_commit4(halo0 + (cur ^ 1) * (SP_HR * SP_HC * 4), raw, tid);
      else {
        int ns = nn;
        const void* xseg = sp_seg(a, ns);
        sp_load_halo<INF32>(halo0 + (cur ^ 1) * (SP_HR * SP_HC * 4), xseg, ns, a.H, a.W, 2 * ntb * SP_TH - 3, 2 * ntc * SP_TW - 3, tid);
      }
    }
    __syncthreads();                            // plane complete (borders were filled one tile ago), next input halo complete
    cur ^= 1;
    // ---- pooling: thread = (pooled row pa, pooled column pb, chunk): window rows 2pa..2pa+2, columns 2pb..2pb+2 of the plane
    {
      const int pa = tid >> 6, pb = (tid >> 3) & 7, ch = tid & 7;
      s16x8_t m = __builtin_bit_cast(s16x8_t, ld16(P + sp_off(2 * pa, 2 * pb, ch)));
#pragma unroll
      for (int e = 1; e < 9; ++e) {
        const s16x8_t x = __builtin_bit_cast(s16x8_t, ld16(P + sp_off(2 * pa + e / 3, 2 * pb + e % 3, ch)));
        m = __builtin_elementwise_max(m, x);
      }
      const int i = tb * 8 + pa, j = tc * 8 + pb;
      if (i < POH && j < POW)
        st16(reinterpret_cast<char*>(a.y) + ((((size_t)n * POH + i) * POW + j) * 64 + ch * 8) * sizeof(T), __builtin_bit_cast(u32x4_t, m));
    }
    // ---- borders of the next tile's plane and the row buffer (they touch neither this plane's readers nor the MFMA operands)
    {
      const bool same_band = more && nn == n && ntb == tb;      // the next tile continues this band: its column 0 is my column 16
      const bool has_above = more && nn == n && ntb > 0;        // ... and its row 0 is the band above (the row buffer), else padding
      const u32x4_t z{0u, 0u, 0u, 0u};
      if (tid < 17 * 8) {                        // column 0 of the next plane <- column 16 of this one (rows 0..16); padding when
        const int r = tid >> 3, ch = tid & 7;    // the next tile starts a band or an image
        st16(Pn + sp_off(r, 0, ch), same_band ? ld16(P + sp_off(r, 16, ch)) : z);
      } else if (tid < 17 * 8 + 128) {           // row buffer <- my last row (columns 1..16), for the band below
        const int c = (tid - 17 * 8) >> 3, ch = tid & 7;
        st16(Rrow + (tc * 16 + c) * 128 + (ch << 4), ld16(P + sp_off(16, 1 + c, ch)));
      } else if (tid < 17 * 8 + 256) {           // row 0 of the next plane (columns 1..16) <- the row buffer above the next tile
        const int c = (tid - 17 * 8 - 128) >> 3, ch = tid & 7;
        u32x4_t v = z;
        if (has_above) v = ld16(Rrow + (ntc * 16 + c) * 128 + (ch << 4));
        st16(Pn + sp_off(0, 1 + c, ch), v);
      }
    }
  }
}

// shapes the fused kernel serves: bf16, eval form, conv output tileable by 16 (256x256, 224x224, 64x64 ... inputs)
bool stem_pool_ok(int dtype, const StemArgs& a, int POH, int POW) {
  static const bool on = [] { const char* e = getenv("SSLCR_STEM_POOL"); return !e || atoi(e) != 0; }();
  // (ReLU: the pooling compares bf16 bit patterns as integers; two tile columns at least: a tile refreshes its own part of the row
  //  buffer in the phase in which the next tile's part is read)
  return on && dtype == DT_BF16 && !a.stats && a.relu && a.OH % 16 == 0 && a.OW % 16 == 0 && a.OW >= 32 && a.OW <= 256 && POH == a.OH / 2 &&
         POW == a.OW / 2;
}

hipError_t launch_stem_pool(int dtype, const StemArgs& a, int POH, int POW, hipStream_t st) {
  if (!stem_pool_ok(dtype, a, POH, POW)) return hipErrorInvalidValue;
  const size_t lds = 64 * SP_WROW + 2 * (SP_HR * SP_HC * 4 * 2) + 2 * SP_PP + (size_t)a.OW * 128;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int cus = device_cus();
  const int grid = a.N < cus ? a.N : cus;
  if (a.in_f32) hipLaunchKernelGGL(stem_pool_fwd_kernel<true>, dim3(grid), dim3(SP_NT), lds, st, a, POH, POW);
  else hipLaunchKernelGGL(stem_pool_fwd_kernel<false>, dim3(grid), dim3(SP_NT), lds, st, a, POH, POW);
  return hipGetLastError();
}

}  // namespace sslcr

// ResNet18 stem on MFMA: uint8/fp32 NCHW ingestion fused with conv1 (7x7 stride 2 pad 3, 3->64), and its wgrad.
// Replaces `.float()` + `reshape` + resnet18.conv1 (eval_BreastPathQ_SSL_CR.py:68-74, models/net.py:32,77).
//
// The 0..255 input values are exact in bf16, so the cast is free.  A workgroup stages a (2*8+5) x (2*16+6) x 4-channel
// halo of the planar NCHW image in LDS (4th channel = 0) and produces an 8x16 tile of output pixels x 64 kouts.
// The GEMM K dimension is ordered (r, s8, c4): for each filter row r, 8 taps x 4 channels = 32 = one bf16 MFMA depth,
// and a lane's 8 K-elements are two horizontally adjacent halo pixels = 16 contiguous LDS bytes (tap s=7 and c=3
// carry zero weights: 147 useful of 224 MACs).  Workgroups are persistent over tiles: the 28 KiB packed weight block
// is staged once, and BatchNorm (sum, sumsq) partials are accumulated in registers across tiles.
#include "kernels.hpp"
#include <type_traits>

namespace sslcr {

constexpr int TH = 8, TW = 16;                 // output tile
constexpr int HR = 2 * TH + 5;                 // 21 halo rows
constexpr int HC = 2 * TW + 6;                 // 38 halo cols (even, covers the zero-weight tap s=7)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

static int stem_grid(int N, int OH, int OW) {
  const int tiles = N * cdiv(OH, TH) * cdiv(OW, TW);
  return tiles < 1024 ? tiles : 1024;
}
int stem_partials_rows(const StemArgs& a) { return stem_grid(a.N, a.OH, a.OW) * 4; }

template <typename T, bool INF32>
__device__ __forceinline__ void stem_load_halo(T* halo, const void* xv, int n, int H, int W, int hi0, int wi0, int tid = threadIdx.x,
                                               int nthreads = 256) {
  for (int idx = tid; idx < 3 * HR * HC; idx += nthreads) {
    const int c = idx / (HR * HC), rem = idx - c * (HR * HC);
    const int rr = rem / HC, cc = rem - rr * HC;
    const int h = hi0 + rr, w = wi0 + cc;
    float v = 0.f;
    if (h >= 0 && w >= 0 && h < H && w < W) {
      const size_t o = ((size_t)(n * 3 + c) * H + h) * W + w;
      v = INF32 ? reinterpret_cast<const float*>(xv)[o] : (float)reinterpret_cast<const uint8_t*>(xv)[o];
    }
    Elem<T>::st(halo + (rr * HC + cc) * 4 + c, v);
  }
}

// Split halo staging for the forward kernel: issue() puts the next tile's input bytes in flight (one value per role, roles
// fixed per thread), commit() converts and writes them into the OTHER LDS halo buffer after the current tile's MFMAs --
// the HBM latency of the planar uint8 gather hides under compute instead of sitting between two barriers.
// kout owned by MFMA tile t, fragment row group q (= lane>>2 for the A fragment, lane>>4 for the accumulator), element j:
// a lane's 16 channels form two 8-channel runs 32 channels apart, so the four lane groups of one pixel write contiguous
// 64-byte segments (16 consecutive channels per lane would leave every 16-byte store half of a 32-byte stride)
#define STEM_CH(t, q, j) ((((t) >> 1) * 32) + ((q) * 8) + (((t) & 1) * 4) + (j))
constexpr int STEM_NEL = (3 * HR * HC + 255) / 256;     // 10 values per thread
// Image n of a (possibly two-segment) input batch: the reference's torch.cat((inputs_x, inputs_u_s)) is an address select here.
template <typename A>
__device__ __forceinline__ const void* stem_seg(const A& a, int& n) {
  if (a.x2 && n >= a.n_split) { n -= a.n_split; return a.x2; }
  return a.x;
}
template <bool INF32>
__device__ __forceinline__ void stem_issue(float (&pv)[STEM_NEL], const int (&role)[STEM_NEL], const void* xv, int n, int H, int W,
                                           int hi0, int wi0) {
#pragma unroll
  for (int i = 0; i < STEM_NEL; ++i) {
    float v = 0.f;
    if (role[i] >= 0) {
      const int c = role[i] >> 20, rr = (role[i] >> 10) & 1023, cc = role[i] & 1023;
      const int h = hi0 + rr, w = wi0 + cc;
      if (h >= 0 && w >= 0 && h < H && w < W) {
        const size_t o = ((size_t)(n * 3 + c) * H + h) * W + w;
        v = INF32 ? reinterpret_cast<const float*>(xv)[o] : (float)reinterpret_cast<const uint8_t*>(xv)[o];
      }
    }
    pv[i] = v;
  }
}
template <typename T>
__device__ __forceinline__ void stem_commit(T* halo, const float (&pv)[STEM_NEL], const int (&role)[STEM_NEL]) {
#pragma unroll
  for (int i = 0; i < STEM_NEL; ++i)
    if (role[i] >= 0) {
      const int c = role[i] >> 20, rr = (role[i] >> 10) & 1023, cc = role[i] & 1023;
      Elem<T>::st(halo + (rr * HC + cc) * 4 + c, pv[i]);
    }
}

// uint8 fast path (W % 4 == 0): the halo window starts 3 pixels left of a 32-pixel boundary, so the aligned dwords from
// one pixel further left cover it exactly: thread (row rr = tid/10, dword d = tid%10) of the first 210 loads ONE dword per
// colour plane = 4 pixels x 3 channels, and writes them as four 8-byte (c0,c1,c2,0) pixels.  12 bytes per load-triple and
// 4 LDS stores per thread per tile instead of 10 byte loads + 10 two-byte stores with per-element address arithmetic.
struct StemRaw { uint32_t d[3]; };
__device__ __forceinline__ StemRaw stem_issue4(const void* xv, int n, int H, int W, int hi0, int wi0, int tid = threadIdx.x) {
  StemRaw r{{0u, 0u, 0u}};
  if (tid < HR * 10) {
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
template <typename T>
__device__ __forceinline__ void stem_commit4(T* halo, const StemRaw& r, int tid = threadIdx.x) {
  if (tid >= HR * 10) return;
  const int rr = tid / 10, d = tid - rr * 10;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = 4 * d - 1 + j;
    if (cc < 0 || cc >= HC) continue;
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = (float)((r.d[c] >> (8 * j)) & 0xffu);
    T* dst = halo + (rr * HC + cc) * 4;
    if constexpr (sizeof(T) == 2) {
      // 0..255 are exact in bf16: the upper half of the fp32 pattern
      const uint32_t lo = (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
      const uint32_t hi = __float_as_uint(f[2]) >> 16;
      *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{lo, hi};
    } else {
      *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{f[0], f[1], f[2], 0.f};
    }
  }
}

template <typename T, bool INF32>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemArgs a, int tiles_h, int tiles_w, int ntiles) {
  constexpr bool BF = Elem<T>::DT == DT_BF16;
  // bf16: unpadded 448-byte rows with the 16-byte chunk index XORed with ((k >> 3) & 2).  A lane group of a ds_read_b128
  // holds rows {0-3, 24-27 | 8-11, 16-19} (+ tile offsets) of two k-chunks; with any row PADDING two of them share a bank
  // slot (2-way conflict on every weight read = 44 % of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT); this swizzle is
  // conflict-free (enumerated over all tiles / filter rows / lane groups) and costs nothing: for a lane it is g ^ 2*(li>>3)
  constexpr int WROW = BF ? 224 * (int)sizeof(T) : 224 * (int)sizeof(T) + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_lds = smem;
  T* halo0 = reinterpret_cast<T*>(smem + 64 * WROW);     // two halo buffers of HR*HC*4 elements

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int role[STEM_NEL];
#pragma unroll
  for (int i = 0; i < STEM_NEL; ++i) {
    const int idx = tid + 256 * i;
    if (idx < 3 * HR * HC) {
      const int c = idx / (HR * HC), rem = idx - c * (HR * HC);
      role[i] = (c << 20) | ((rem / HC) << 10) | (rem % HC);
    } else {
      role[i] = -1;
    }
  }
  float pv[STEM_NEL];
  StemRaw raw{{0u, 0u, 0u}};
  const bool fast = !INF32 && (a.W & 3) == 0;
  // stage the packed weights [64][224] once
  {
    const char* wg = reinterpret_cast<const char*>(a.w);
    constexpr int CH = 224 * sizeof(T) / 16;     // 16-byte chunks per row
    for (int i = tid; i < 64 * CH; i += 256) {
      const int k = i / CH, c = i - k * CH;
      st16(w_lds + k * WROW + (BF ? (c ^ ((k >> 3) & 2)) : c) * 16, ld16(wg + (size_t)i * 16));
    }
    for (int i = tid; i < 2 * HR * HC * 4; i += 256) Elem<T>::st(halo0 + i, 0.f);
  }
  __syncthreads();
  int cur = 0;
  // XCD-aware walk (workgroups land on XCD blockIdx % 8): each XCD takes a contiguous run of tiles per round, so that neighbouring
  // tiles' shared halo columns -- and the 64-byte sectors that a 37-byte uint8 row segment only partly uses -- hit one L2
  const int G = gridDim.x;
  const int vb = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  if (vb < ntiles) {
    const int t0 = vb;
    const int n = t0 / (tiles_h * tiles_w), rem = t0 - n * tiles_h * tiles_w;
    int ns = n;
    const void* xseg = stem_seg(a, ns);
    if (fast) {
      raw = stem_issue4(xseg, ns, a.H, a.W, 2 * (rem / tiles_w) * TH - 3, 2 * (rem % tiles_w) * TW - 3);
      stem_commit4<T>(halo0, raw);
    } else {
      stem_issue<INF32>(pv, role, xseg, ns, a.H, a.W, 2 * (rem / tiles_w) * TH - 3, 2 * (rem % tiles_w) * TW - 3);
      stem_commit<T>(halo0, pv, role);
    }
  }
  __syncthreads();
  float s1[16], s2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  float bias[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bias[j] = a.bias ? a.bias[STEM_CH(j >> 2, g, j & 3)] : 0.f;

  for (int tile = vb; tile < ntiles; tile += G) {
    const int n = tile / (tiles_h * tiles_w), rem = tile - n * tiles_h * tiles_w;
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int ho0 = th * TH, wo0 = tw * TW;
    const T* halo = halo0 + cur * (HR * HC * 4);
    const int nxt = tile + G;
    if (nxt < ntiles) {
      int nn = nxt / (tiles_h * tiles_w);
      const int nrem = nxt - nn * tiles_h * tiles_w;
      const void* xseg = stem_seg(a, nn);
      if (fast) raw = stem_issue4(xseg, nn, a.H, a.W, 2 * (nrem / tiles_w) * TH - 3, 2 * (nrem % tiles_w) * TW - 3);
      else stem_issue<INF32>(pv, role, xseg, nn, a.H, a.W, 2 * (nrem / tiles_w) * TH - 3, 2 * (nrem % tiles_w) * TW - 3);
    }

    f32x4_t acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      if constexpr (BF) {
        u32x4_t af[4], bfr[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = STEM_CH(t, li >> 2, li & 3);
          af[t] = ld16(w_lds + k * WROW + (r * 8 + 2 * (g ^ ((li >> 3) << 1))) * 4 * sizeof(T));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int hr = 2 * (2 * wave + p) + r;
          bfr[p] = ld16(reinterpret_cast<const char*>(halo) + ((hr * HC + 2 * li + 2 * g) * 4) * sizeof(T));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[t]), __builtin_bit_cast(bf16x8_t, bfr[p]), acc[t][p], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
          float av[4], bv[2];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int k = STEM_CH(t, li >> 2, li & 3);
            av[t] = *reinterpret_cast<const float*>(w_lds + k * WROW + ((r * 8 + s) * 4 + g) * 4);
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int hr = 2 * (2 * wave + p) + r;
            bv[p] = reinterpret_cast<const float*>(halo)[(hr * HC + 2 * li + s) * 4 + g];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[p], acc[t][p], 0, 0, 0);
        }
      }
    }
    // epilogue: lane holds kouts g*16 .. g*16+15 of pixel (ho0+2*wave+p, wo0+li)
    // (the kernel is VALU-bound -- 56 MFMAs against ~650 VALU per tile and wave -- so the common training case, a full tile
    // with no bias/ReLU, takes a path without the per-element selects, adds and the integer bf16 rounding)
    if (ho0 + TH <= a.OH && wo0 + TW <= a.OW && !a.bias && !a.relu) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int ho = ho0 + 2 * wave + p, wo = wo0 + li;
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float q = acc[t][p][j];
            s1[t * 4 + j] += q;
            s2[t * 4 + j] = fmaf(q, q, s2[t * 4 + j]);
            v[t * 4 + j] = q;
          }
        char* yp = reinterpret_cast<char*>(a.y) + ((((size_t)n * a.OH + ho) * a.OW + wo) * 64) * sizeof(T);
        constexpr int EPC = Elem<T>::EPC;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q) st16(yp + STEM_CH((q * EPC) >> 2, g, 0) * sizeof(T), PackH<T>::run(v + q * EPC));
      }
    } else if (ho0 + TH <= a.OH && wo0 + TW <= a.OW && a.bias && a.relu && !a.stats) {
      // the eval-mode (teacher / validate) case -- full tile, folded-BatchNorm bias, ReLU, no statistics: add, one v_med3 clamp,
      // pack, store; the general path below spends a select, two statistics updates and a second max per element on it
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int ho = ho0 + 2 * wave + p, wo = wo0 + li;
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[t * 4 + j] = relu0(acc[t][p][j] + bias[t * 4 + j]);
        char* yp = reinterpret_cast<char*>(a.y) + ((((size_t)n * a.OH + ho) * a.OW + wo) * 64) * sizeof(T);
        constexpr int EPC = Elem<T>::EPC;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q) st16(yp + STEM_CH((q * EPC) >> 2, g, 0) * sizeof(T), PackH<T>::run(v + q * EPC));
      }
    } else
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int ho = ho0 + 2 * wave + p, wo = wo0 + li;
      const bool valid = ho < a.OH && wo < a.OW;
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float q = valid ? acc[t][p][j] : 0.f;
          s1[t * 4 + j] += q;
          s2[t * 4 + j] = fmaf(q, q, s2[t * 4 + j]);
          float o = q + bias[t * 4 + j];
          v[t * 4 + j] = a.relu ? fmaxf(o, 0.f) : o;
        }
      if (valid) {
        char* yp = reinterpret_cast<char*>(a.y) + ((((size_t)n * a.OH + ho) * a.OW + wo) * 64) * sizeof(T);
        constexpr int EPC = Elem<T>::EPC;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q)         // v[q*EPC ..] are EPC consecutive channels starting at STEM_CH(.)
          st16(yp + STEM_CH((q * EPC) >> 2, g, 0) * sizeof(T), Elem<T>::pack(v + q * EPC));
      }
    }
    if (nxt < ntiles) {
      if (fast) stem_commit4<T>(halo0 + (cur ^ 1) * (HR * HC * 4), raw);
      else stem_commit<T>(halo0 + (cur ^ 1) * (HR * HC * 4), pv, role);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1[j] = row16_sum(s1[j]); s2[j] = row16_sum(s2[j]); }
    if (li == 0) {
      float* sp = a.stats + ((size_t)(blockIdx.x * 4 + wave) * 2) * 64;
#pragma unroll
      for (int j = 0; j < 16; ++j) { sp[STEM_CH(j >> 2, g, j & 3)] = s1[j]; sp[64 + STEM_CH(j >> 2, g, j & 3)] = s2[j]; }
    }
  }
}

template <typename T, bool INF32>
static hipError_t launch_stem_t(const StemArgs& a, hipStream_t st) {
  const int th = cdiv(a.OH, TH), tw = cdiv(a.OW, TW);
  const int ntiles = a.N * th * tw;
  const size_t lds = 64 * (224 * sizeof(T) + 16) + 2 * HR * HC * 4 * sizeof(T);
  auto kern = stem_fwd_kernel<T, INF32>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(stem_grid(a.N, a.OH, a.OW)), dim3(256), lds, st, a, th, tw, ntiles);
  return hipGetLastError();
}

hipError_t launch_stem(int dtype, const StemArgs& a, hipStream_t st) {
  // sslcr_stem_desc.out_scale exists in the fused conv + max-pool kernel only (stem_pool.hip): here 16 more live values in the
  // output stage took stem_fwd_kernel from 232 to 272 registers -- one wave per SIMD instead of two for the TRAIN forward too
  if (a.out_scale) return hipErrorInvalidValue;
  if (dtype == DT_BF16) return a.in_f32 ? launch_stem_t<bf16_t, true>(a, st) : launch_stem_t<bf16_t, false>(a, st);
  return a.in_f32 ? launch_stem_t<float, true>(a, st) : launch_stem_t<float, false>(a, st);
}

// ------------------------------------------------------------------ stem wgrad
// D[kout][feature (r,s8,c4)] = sum_pixels dY^T[kout][pixel] * patch[pixel][feature]; the patch operand is read straight
// out of the halo with per-lane addresses (ds_read_b64_tr_b16 delivers the pixel-major -> K-major transpose for free).
//
// POOL form (sslcr_stem_wgrad_pool): dY is never in memory.  The stem's gradient arrives through maxpool3x3/2 -> ReLU -> bn0, and
// the BatchNorm-backward apply pass (bn_bwd_apply_pool_kernel: pooled gradient + argmax codes + raw conv output -> dY) is run on
// the tile while it is staged: thread = (2x2 pixel block, 16-byte channel chunk) of the 8x16 tile, exactly the work item of that
// kernel, writing its four dY chunks into the LDS tile instead of 2.3 GB of HBM that this kernel would read straight back.
struct StemPoolRaw { u32x4_t dv[4]; uint32_t am[4][2]; u32x4_t xv[4]; };

template <typename T, bool INF32, bool POOL>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemWgradArgs a, const BnBwdArgs b, int tiles_h, int tiles_w, int ntiles, f32x4_t* partials) {
  constexpr bool BF = Elem<T>::DT == DT_BF16;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int PS = BF ? 32 : 16;               // pixels per MFMA depth step
  constexpr int RB = 64 * sizeof(T);
  constexpr int CPR = RB / 16;
  constexpr int YL = TH * TW * CPR / 256;        // dY staging loads per thread and tile (4 / 8)
  constexpr int YBUF = TH * TW * RB, HBUF = HR * HC * 4 * sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two (dY tile, input halo) buffer pairs: the next tile's global loads are in flight during this tile's MFMAs and are
  // written to the other pair afterwards -- one barrier per tile, no global latency between barriers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int row = tid / CPR, chunk = tid % CPR;
  const bool fast = !INF32 && (a.W & 3) == 0;
  for (int i = tid; i < 2 * (YBUF + HBUF) / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0u;

  f32x4_t acc[14];
#pragma unroll
  for (int f = 0; f < 14; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int IT = POOL ? (TH / 2) * (TW / 2) * CPR / 256 : 1;     // POOL work items per thread and tile (1 / 2)
  u32x4_t yreg[POOL ? 1 : YL];
  StemPoolRaw praw[IT];
  float cA[EPC], cB[EPC], cC[EPC], rsh[EPC];     // POOL: dY = cA g + cB x + cC for this thread's channel chunk (bn_bwd_apply_pool_kernel)
  if constexpr (POOL) {
    const float invM = (float)(1.0 / b.count);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = chunk * EPC + e;
      const float is = b.invstd[c], sc = b.scale[c];
      const float m0 = (float)b.sums[c] * invM, m1 = (float)b.sums[64 + c] * invM;
      cA[e] = sc;
      cB[e] = -sc * is * is * m1;
      cC[e] = -sc * m0 - cB[e] * b.mean[c];
      rsh[e] = b.shift[c];
    }
    if (blockIdx.x == 0 && b.dgamma && b.dbeta && tid < 64) {        // affine gradients, as workgroup 0 of the apply pass does
      b.dgamma[tid] += (float)(b.sums[64 + tid] * (double)b.invstd[tid] * (double)b.pg_scale);
      b.dbeta[tid] += (float)(b.sums[tid] * (double)b.pg_scale);
    }
  }
  StemRaw raw{{0u, 0u, 0u}};
  auto issue = [&](int tile) {
    const int n = tile / (tiles_h * tiles_w), rem = tile - n * tiles_h * tiles_w;
    const int ho0 = (rem / tiles_w) * TH, wo0 = (rem % tiles_w) * TW;
    if (fast) {
      int ns = n;
      const void* xseg = stem_seg(a, ns);
      raw = stem_issue4(xseg, ns, a.H, a.W, 2 * ho0 - 3, 2 * wo0 - 3);
    }
    if constexpr (POOL) {
      const char* xg = reinterpret_cast<const char*>(b.x);
      const char* dyg = reinterpret_cast<const char*>(b.pool_dy);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int blk = (tid + 256 * it) / CPR;
        const int bh = (ho0 >> 1) + blk / (TW / 2), bw = (wo0 >> 1) + blk % (TW / 2);
        // the four windows and the four pixels: unconditional loads from clamped addresses, validity applied in commit
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int oh = bh + (q >> 1), ow = bw + (q & 1);
          const int ohc = oh < b.pOH ? oh : b.pOH - 1, owc = ow < b.pOW ? ow : b.pOW - 1;
          const size_t o = (((size_t)n * b.pOH + ohc) * b.pOW + owc) * CPR + chunk;
          praw[it].dv[q] = ld16_nt(dyg + o * 16);
          if constexpr (EPC == 8) {
            const u32x2_t v = *reinterpret_cast<const u32x2_t*>(b.pool_argmax + o * EPC);
            praw[it].am[q][0] = v[0]; praw[it].am[q][1] = v[1];
          } else {
            praw[it].am[q][0] = *reinterpret_cast<const uint32_t*>(b.pool_argmax + o * EPC); praw[it].am[q][1] = 0;
          }
          const int h = 2 * bh + (q >> 1), w = 2 * bw + (q & 1);
          const int hc = h < a.OH ? h : a.OH - 1, wc = w < a.OW ? w : a.OW - 1;
          praw[it].xv[q] = ld16_nt(xg + ((((size_t)n * a.OH + hc) * a.OW + wc) * CPR + chunk) * 16);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < YL; ++i) {
        const int p = row + (256 / CPR) * i;
        const int ho = ho0 + p / TW, wo = wo0 + p % TW;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (ho < a.OH && wo < a.OW)
          v = ld16(reinterpret_cast<const char*>(a.dy) + ((((size_t)n * a.OH + ho) * a.OW + wo) * 64 + chunk * EPC) * sizeof(T));
        yreg[i] = v;
      }
    }
  };
  auto commit = [&](int tile, int buf) {
    char* yt = smem + buf * (YBUF + HBUF);
    T* hl = reinterpret_cast<T*>(yt + YBUF);
    if (fast) {
      stem_commit4<T>(hl, raw);
    } else {
      int n = tile / (tiles_h * tiles_w);
      const int rem = tile - n * tiles_h * tiles_w;
      const void* xseg = stem_seg(a, n);
      stem_load_halo<T, INF32>(hl, xseg, n, a.H, a.W, 2 * (rem / tiles_w) * TH - 3, 2 * (rem % tiles_w) * TW - 3);
    }
    if constexpr (POOL) {
      const int rem = tile % (tiles_h * tiles_w);
      const int ho0 = (rem / tiles_w) * TH, wo0 = (rem % tiles_w) * TW;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int blk = (tid + 256 * it) / CPR;
        const int lh = 2 * (blk / (TW / 2)), lw = 2 * (blk % (TW / 2));
        // branch-free throughout: a window outside the pooled map contributes zeros, the argmax match is a select
        float dw[4][EPC];
        uint32_t cd[4][EPC];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool wok = ((ho0 + lh) >> 1) + (q >> 1) < b.pOH && ((wo0 + lw) >> 1) + (q & 1) < b.pOW;
          u32x4_t v = praw[it].dv[q];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = wok ? v[j] : 0u;
          Elem<T>::unpack(v, dw[q]);
#pragma unroll
          for (int e = 0; e < EPC; ++e) cd[q][e] = (praw[it].am[q][e >> 2] >> (8 * (e & 3))) & 0xffu;
        }
        const float thr = b.relu_from_x ? 0.f : -__builtin_inff();   // ReLU between the BatchNorm and the pool: g lives where y > thr
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) {
            float xf[EPC], g[EPC], d[EPC];
            Elem<T>::unpack(praw[it].xv[pi * 2 + pj], xf);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = 0.f;
#pragma unroll
            for (int di = 0; di <= pi; ++di)
#pragma unroll
              for (int dj = 0; dj <= pj; ++dj) {
                const uint32_t code = (uint32_t)((pi - 2 * di + 1) * 3 + (pj - 2 * dj + 1));
#pragma unroll
                for (int e = 0; e < EPC; ++e) g[e] += cd[di * 2 + dj][e] == code ? dw[di * 2 + dj][e] : 0.f;
              }
            const bool pok = ho0 + lh + pi < a.OH && wo0 + lw + pj < a.OW;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const float ge = fmaf(xf[e], cA[e], rsh[e]) > thr ? g[e] : 0.f;
              const float de = fmaf(cA[e], ge, fmaf(cB[e], xf[e], cC[e]));
              d[e] = pok ? de : 0.f;
            }
            st16(yt + ((lh + pi) * TW + lw + pj) * RB + chunk * 16, Elem<T>::pack(d));
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < YL; ++i) st16(yt + (row + (256 / CPR) * i) * RB + chunk * 16, yreg[i]);
    }
  };

  __syncthreads();                               // zero fill (4th halo channel stays 0 on the slow path)
  int cur = 0;
  if ((int)blockIdx.x < ntiles) {
    issue(blockIdx.x);
    commit(blockIdx.x, 0);
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) issue(nxt);
    const char* ytile = smem + cur * (YBUF + HBUF);
    const T* halo = reinterpret_cast<const T*>(ytile + YBUF);
#pragma unroll 1
    for (int step = 0; step < TH * TW / PS; ++step) {
      const char* ystep = ytile + step * PS * RB;
      if constexpr (BF) {
        // A: dY^T, kouts 16*wave + li, pixels 8g..8g+7
        const char* pa = ystep + (8 * g + (li >> 2)) * RB + (16 * wave + (li & 3) * 4) * 2;
        s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa));
        s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa + 4 * RB));
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
        // this lane's source pixels for the transpose read: j = li>>2 (+4), feature quad q = li&3
        const int p0 = step * PS + 8 * g + (li >> 2), p1 = p0 + 4;
        const int h0 = 2 * (p0 / TW), w0 = 2 * (p0 % TW), h1 = 2 * (p1 / TW), w1 = 2 * (p1 % TW);
#pragma unroll
        for (int f = 0; f < 14; ++f) {
          const int r = f >> 1, s0 = (f & 1) * 4;
          const char* b0 = reinterpret_cast<const char*>(halo) + (((h0 + r) * HC + w0 + s0 + (li & 3)) * 4) * 2;
          const char* b1 = reinterpret_cast<const char*>(halo) + (((h1 + r) * HC + w1 + s0 + (li & 3)) * 4) * 2;
          s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b0));
          s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b1));
          const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag, acc[f], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < PS / 4; ++q) {
          const int pl = 4 * q + g;
          const float av = *reinterpret_cast<const float*>(ystep + pl * RB + (16 * wave + li) * 4);
          const int p = step * PS + pl;
          const int hh = 2 * (p / TW), ww = 2 * (p % TW);
#pragma unroll
          for (int f = 0; f < 14; ++f) {
            const int r = f >> 1, s = (f & 1) * 4 + (li >> 2), c = li & 3;
            const float bv = reinterpret_cast<const float*>(halo)[((hh + r) * HC + ww + s) * 4 + c];
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[f], 0, 0, 0);
          }
        }
      }
    }
    if (nxt < ntiles) commit(nxt, cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (partials) {
    // every workgroup adds into the SAME 64 x 147 weights: as atomics that is up to 768 adds per address.  Slab + fold launch
    // instead (see wgrad_halo.hip)
    f32x4_t* sp = partials + (size_t)blockIdx.x * 14 * 256 + threadIdx.x;
#pragma unroll
    for (int f = 0; f < 14; ++f) sp[f * 256] = acc[f];
    return;
  }
  // D[row = kout 16*wave+4g+j][col = feature li -> (s = s0 + li>>2, c = li&3)] -> dW[k][c][r][s]
#pragma unroll
  for (int f = 0; f < 14; ++f) {
    const int r = f >> 1, s = (f & 1) * 4 + (li >> 2), c = li & 3;
    if (s < 7 && c < 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {             // a one-workgroup launch: the only writer
        const int k = 16 * wave + 4 * g + j;
        a.dw[((k * 3 + c) * 7 + r) * 7 + s] += acc[f][j];
      }
    }
  }
}

// ---- bf16 pool form, role-split (the fp32 parity mode keeps the POOL instantiation above: its tiles do not fit three stages)
// The apply-pass arithmetic is ~570 VALU instructions per (2x2 block, chunk) item against 56 MFMAs per wave and tile: run one
// after the other in the same wave (above) neither the matrix pipe nor the VALU is busy half the time.  Here a workgroup is
// 12 waves: waves 0-3 ("matrix") stage the tile by DMA (global_load_lds: no registers, issued two tiles ahead), stage the image
// halo and run the MFMAs; waves 4-11 ("apply") turn the raw staged tile -- conv output x, pooled gradient windows, argmax codes
// -- into dY IN PLACE over x, one tile ahead of the MFMAs.  Three stages rotate: dY(t) under the MFMAs | raw(t+1) under the
// apply pass | DMA(t+2) in flight; one barrier per tile.
typedef const __attribute__((address_space(1))) void* stem_gptr_t;
typedef __attribute__((address_space(3))) void* stem_lptr_t;
constexpr int P2_XB = TH * TW * 128;                  // x / dY tile: [pixel][64 bf16]
constexpr int P2_WIN = (TH / 2 + 1) * (TW / 2 + 1);   // 45 pooling windows touch the tile
constexpr int P2_WB = 384 * 16;                       // 45 x 128 B of pooled gradient, padded to 6 wave-loads
constexpr int P2_AB = 192 * 16;                       // 45 x 64 B of argmax codes, padded to 3 wave-loads
constexpr int P2_DUMP = P2_XB + P2_WB + P2_AB;       // 1 KiB nobody reads: target of the loads that pad every wave to 7 per tile
constexpr int P2_STAGE = P2_DUMP + 1024;
constexpr int P2_NS = 5;
constexpr int P2_HBUF = HR * HC * 4 * 2;

// LDS transpose read as inline asm.  Through the intrinsic, the compiler's wait-count pass (no usable memory operand on it) puts
// s_waitcnt vmcnt(0) in front of every such read while ANY global_load_lds is outstanding -- which drains the three tiles of DMA
// this kernel keeps in flight (plain ds_read_b128 after global_load_lds, conv_dma.hip, does not get that wait).  The asm is
// opaque to that pass, so the LDS counter is handled by hand next to the reads (see the MFMA loop).
template <int OFF>
__device__ __forceinline__ u32x2_t p2_tr16(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t p2_lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}
// Workgroup barrier that leaves the vector-memory counter alone: __syncthreads() is a workgroup-scope fence, which the compiler
// lowers to s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier -- and vmcnt(0) here means "wait for the three tiles of DMA in flight".
// What crosses this barrier between waves is LDS data only; the DMA writes it publishes were waited for explicitly.
__device__ __forceinline__ void p2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define P2_BOFF(f) ((((f) >> 1) * HC + ((f) & 1) * 4) * 8)
__device__ __forceinline__ bf16x8_t p2_frag(const u32x2_t& lo, const u32x2_t& hi) {
  return __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
}

template <bool INF32>
__global__ __launch_bounds__(768) void stem_wgrad_pool2_kernel(const StemWgradArgs a, const BnBwdArgs b, int tiles_h, int tiles_w, int ntiles,
                                                               f32x4_t* partials) {
  using T = bf16_t;
  constexpr int RB = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem + P2_NS * P2_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool matrix = wave < 4;
  const int li = lane & 15, g = lane >> 4;
  const bool fast = !INF32 && (a.W & 3) == 0;
  for (int i = tid; i < 2 * P2_HBUF / 4; i += 768) reinterpret_cast<uint32_t*>(halo_base)[i] = 0u;

  // ---- apply role (8 waves): this thread's item is (2x2 pixel block blk, 4 channels = half a 16-byte chunk) of every tile
  const int pt = tid >= 256 ? tid - 256 : 0;
  const int cq = pt & 15, blk = pt >> 4;           // channel quad 0..15, block 0..31
  const int lh = 2 * (blk >> 3), lw = 2 * (blk & 7);
  float cA[4], cB[4], cC[4];
  if (!matrix) {
    const float invM = (float)(1.0 / b.count);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = cq * 4 + e;
      const float is = b.invstd[c], sc = b.scale[c];
      const float m0 = (float)b.sums[c] * invM, m1 = (float)b.sums[64 + c] * invM;
      cA[e] = sc;
      cB[e] = -sc * is * is * m1;
      cC[e] = -sc * m0 - cB[e] * b.mean[c];
    }
    if (blockIdx.x == 0 && b.dgamma && b.dbeta && pt < 64) {          // affine gradients, as workgroup 0 of the apply pass does
      b.dgamma[pt] += (float)(b.sums[64 + pt] * (double)b.invstd[pt] * (double)b.pg_scale);
      b.dbeta[pt] += (float)(b.sums[pt] * (double)b.pg_scale);
    }
  }
  // No ReLU test here: sslcr_bn_relu_maxpool records code 9 ("nobody") for a window whose maximum is not positive, and the pixel
  // a live window names has y = that maximum > 0.  EDGE = the tile touches the border of the map (ragged tile, or its last
  // window row / column lies outside the pooled map): validity by select; interior tiles skip those selects.
  auto apply = [&](int tile, char* st, auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    const int rem = tile % (tiles_h * tiles_w);
    const int ho0 = (rem / tiles_w) * TH, wo0 = (rem % tiles_w) * TW;
    u32x2_t dv[4], xv[4];
    uint32_t am[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int win = ((lh >> 1) + (q >> 1)) * (TW / 2 + 1) + (lw >> 1) + (q & 1);
      dv[q] = *reinterpret_cast<const u32x2_t*>(st + P2_XB + win * RB + cq * 8);
      am[q] = *reinterpret_cast<const uint32_t*>(st + P2_XB + P2_WB + win * 64 + cq * 4);
      xv[q] = *reinterpret_cast<const u32x2_t*>(st + ((lh + (q >> 1)) * TW + lw + (q & 1)) * RB + cq * 8);
    }
    float dw[4][4];
    uint32_t cd[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x2_t v = dv[q];
      if constexpr (EDGE) {
        const bool wok = ((ho0 + lh) >> 1) + (q >> 1) < b.pOH && ((wo0 + lw) >> 1) + (q & 1) < b.pOW;
        v[0] = wok ? v[0] : 0u; v[1] = wok ? v[1] : 0u;
      }
      dw[q][0] = bf_lo(v[0]); dw[q][1] = bf_hi(v[0]); dw[q][2] = bf_lo(v[1]); dw[q][3] = bf_hi(v[1]);
#pragma unroll
      for (int e = 0; e < 4; ++e) cd[q][e] = (am[q] >> (8 * e)) & 0xffu;
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        const u32x2_t xw = xv[pi * 2 + pj];
        const float xf[4] = {bf_lo(xw[0]), bf_hi(xw[0]), bf_lo(xw[1]), bf_hi(xw[1])};
        float gg[4] = {0.f, 0.f, 0.f, 0.f}, d[4];
#pragma unroll
        for (int di = 0; di <= pi; ++di)
#pragma unroll
          for (int dj = 0; dj <= pj; ++dj) {
            const uint32_t code = (uint32_t)((pi - 2 * di + 1) * 3 + (pj - 2 * dj + 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) gg[e] += cd[di * 2 + dj][e] == code ? dw[di * 2 + dj][e] : 0.f;
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = fmaf(cA[e], gg[e], fmaf(cB[e], xf[e], cC[e]));
        if constexpr (EDGE) {
          const bool pok = ho0 + lh + pi < a.OH && wo0 + lw + pj < a.OW;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = pok ? d[e] : 0.f;
        }
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t o;                                         // hardware RNE pack: the same bits as Elem<bf16_t>::pack for finite values
        o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{d[0], d[1]}, bf16x2_t));
        o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{d[2], d[3]}, bf16x2_t));
        *reinterpret_cast<u32x2_t*>(st + ((lh + pi) * TW + lw + pj) * RB + cq * 8) = o;
      }
  };
  auto apply_tile = [&](int tile, char* st) {
    const int rem = tile % (tiles_h * tiles_w);
    const int ho0 = (rem / tiles_w) * TH, wo0 = (rem % tiles_w) * TW;
    const bool interior = ho0 + TH <= a.OH && wo0 + TW <= a.OW && (ho0 >> 1) + TH / 2 < b.pOH && (wo0 >> 1) + TW / 2 < b.pOW;
    if (interior) apply(tile, st, std::false_type{});
    else apply(tile, st, std::true_type{});
  };

  // ---- matrix role: DMA of a raw tile into a stage (25 wave-loads of 1 KiB over the 4 waves), image halo, MFMAs
  const char* const xg = reinterpret_cast<const char*>(b.x);
  const char* const dyg = reinterpret_cast<const char*>(b.pool_dy);
  const char* const amg = reinterpret_cast<const char*>(b.pool_argmax);
  // per-lane byte offsets of this wave's 7 loads inside an interior tile (tile-invariant: the tile only moves a scalar base)
  int woff[2], aoff;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int win = ((wave + 4 * i) * 64 + lane) >> 3;
    win = win < P2_WIN ? win : P2_WIN - 1;
    woff[i] = ((win / (TW / 2 + 1)) * b.pOW + win % (TW / 2 + 1)) * 128 + (lane & 7) * 16;
  }
  {
    int win = (wave * 64 + lane) >> 2;
    win = win < P2_WIN ? win : P2_WIN - 1;
    aoff = ((win / (TW / 2 + 1)) * b.pOW + win % (TW / 2 + 1)) * 64 + (lane & 3) * 16;
  }
  auto dma = [&](int tile, char* st) {
    const int n = tile / (tiles_h * tiles_w), rem = tile - n * tiles_h * tiles_w;
    const int ho0 = (rem / tiles_w) * TH, wo0 = (rem % tiles_w) * TW;
    const bool interior = ho0 + TH <= a.OH && wo0 + TW <= a.OW && (ho0 >> 1) + TH / 2 < b.pOH && (wo0 >> 1) + TW / 2 < b.pOW;
    if (interior) {
      // no clamping anywhere in the tile: uniform base + the precomputed lane offset (the address arithmetic of the general
      // form below, ~40 VALU instructions per load, was half of this role's instruction stream)
      const char* xt = xg + (((size_t)n * a.OH + ho0) * a.OW + wo0) * 128;
      const char* dt = dyg + (((size_t)n * b.pOH + (ho0 >> 1)) * b.pOW + (wo0 >> 1)) * 128;
      const char* at = amg + (((size_t)n * b.pOH + (ho0 >> 1)) * b.pOW + (wo0 >> 1)) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((stem_gptr_t)(xt + (size_t)(j >> 1) * a.OW * 128 + (j & 1) * 1024 + (uint32_t)(lane * 16)),
                                         (stem_lptr_t)(st + j * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int l = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((stem_gptr_t)(dt + (uint32_t)woff[i]), (stem_lptr_t)(st + (l < 6 ? P2_XB + l * 1024 : P2_DUMP)), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((stem_gptr_t)(at + (uint32_t)aoff), (stem_lptr_t)(st + (wave < 3 ? P2_XB + P2_WB + wave * 1024 : P2_DUMP)), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // x: wave-load j = 8 pixels of tile row j/2
      const int j = wave + 4 * i;
      const int h = ho0 + (j >> 1), w = wo0 + (j & 1) * 8 + (lane >> 3);
      const int hc = h < a.OH ? h : a.OH - 1, wc = w < a.OW ? w : a.OW - 1;
      const char* src = xg + ((((size_t)n * a.OH + hc) * a.OW + wc) * 8 + (lane & 7)) * 16;
      __builtin_amdgcn_global_load_lds((stem_gptr_t)src, (stem_lptr_t)(st + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                 // pooled gradient: slot = (window, chunk), 45 x 8 of 384; l = 6, 7 pad
      const int l = wave + 4 * i;
      int win = (l * 64 + lane) >> 3;
      win = win < P2_WIN ? win : P2_WIN - 1;
      const int oh = (ho0 >> 1) + win / (TW / 2 + 1), ow = (wo0 >> 1) + win % (TW / 2 + 1);
      const int ohc = oh < b.pOH ? oh : b.pOH - 1, owc = ow < b.pOW ? ow : b.pOW - 1;
      const char* src = dyg + ((((size_t)n * b.pOH + ohc) * b.pOW + owc) * 8 + (lane & 7)) * 16;
      __builtin_amdgcn_global_load_lds((stem_gptr_t)src, (stem_lptr_t)(st + (l < 6 ? P2_XB + l * 1024 : P2_DUMP)), 16, 0, 0);
    }
    {                                             // argmax codes: slot = (window, quarter), 45 x 4 of 192; wave 3 pads
      int win = (wave * 64 + lane) >> 2;
      win = win < P2_WIN ? win : P2_WIN - 1;
      const int oh = (ho0 >> 1) + win / (TW / 2 + 1), ow = (wo0 >> 1) + win % (TW / 2 + 1);
      const int ohc = oh < b.pOH ? oh : b.pOH - 1, owc = ow < b.pOW ? ow : b.pOW - 1;
      const char* src = amg + ((((size_t)n * b.pOH + ohc) * b.pOW + owc) * 4 + (lane & 3)) * 16;
      __builtin_amdgcn_global_load_lds((stem_gptr_t)src, (stem_lptr_t)(st + (wave < 3 ? P2_XB + P2_WB + wave * 1024 : P2_DUMP)), 16, 0, 0);
    }
  };
  // image halo: by the apply waves (their vector-memory counter carries nothing else, so waiting for these loads never
  // touches the DMA queue of the matrix waves), issued ahead of the tile's apply pass and committed after it
  auto halo_issue = [&](int tile) -> StemRaw {
    if (!fast || tile >= ntiles) return StemRaw{{0u, 0u, 0u}};
    int n = tile / (tiles_h * tiles_w);
    const int rem = tile - n * tiles_h * tiles_w;
    const void* xseg = stem_seg(a, n);
    return stem_issue4(xseg, n, a.H, a.W, 2 * (rem / tiles_w) * TH - 3, 2 * (rem % tiles_w) * TW - 3, pt);
  };
  auto halo_commit = [&](int tile, int buf, const StemRaw& raw) {
    T* hl = reinterpret_cast<T*>(halo_base + buf * P2_HBUF);
    if (fast) {
      stem_commit4<T>(hl, raw, pt);
    } else {
      int n = tile / (tiles_h * tiles_w);
      const int rem = tile - n * tiles_h * tiles_w;
      const void* xseg = stem_seg(a, n);
      stem_load_halo<T, INF32>(hl, xseg, n, a.H, a.W, 2 * (rem / tiles_w) * TH - 3, 2 * (rem % tiles_w) * TW - 3, pt, 512);
    }
  };
  auto stage = [&](int k) -> char* { return smem + k * P2_STAGE; };

  // XCD-aware walk like stem_fwd_kernel: an XCD takes a contiguous run of tiles per round (shared image halos and pooling windows)
  const int gs = gridDim.x;
  const int t0 = (gs & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (gs >> 3) + (int)(blockIdx.x >> 3);
  __syncthreads();                    // halo zero fill
  if (matrix) {
    if (t0 < ntiles) dma(t0, stage(0));
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
  } else if (t0 < ntiles) {
    const StemRaw r0 = halo_issue(t0);
    halo_commit(t0, 0, r0);
  }
  __syncthreads();
  if (matrix) {
#pragma unroll
    for (int k = 1; k < P2_NS - 1; ++k)
      if (t0 + k * gs < ntiles) dma(t0 + k * gs, stage(k));
    __builtin_amdgcn_s_waitcnt(0x0f70);
  } else if (t0 < ntiles) {
    apply_tile(t0, stage(0));
  }
  __syncthreads();
  // One loop per role, the same number of barriers in each: with the roles as two branches of one loop body the accumulators
  // of the matrix waves (56 registers) stay allocated through the apply code and the kernel does not fit 3 waves per SIMD.
  if (!matrix) {
    int cur = 0, i5 = 0;              // i5 = iteration mod 5: dY(t) in stage i5, raw(t+1) in i5+1, DMA(t+4) into i5+4
    for (int tile = t0; tile < ntiles; tile += gs) {
      const int nxt = tile + gs;
      if (nxt < ntiles) {
        const StemRaw r = halo_issue(nxt);
        apply_tile(nxt, stage(i5 + 1 < P2_NS ? i5 + 1 : 0));
        halo_commit(nxt, cur ^ 1, r);
      }
      p2_barrier();
      cur ^= 1;
      i5 = i5 + 1 < P2_NS ? i5 + 1 : 0;
    }
    return;
  }
  f32x4_t acc[14];
#pragma unroll
  for (int f = 0; f < 14; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int cur = 0, i5 = 0;
  for (int tile = t0; tile < ntiles; tile += gs) {
    {
      const int nx4 = tile + 4 * gs;
      if (nx4 < ntiles) dma(nx4, stage(i5 + 4 < P2_NS ? i5 + 4 : i5 + 4 - P2_NS));
      const char* sA = stage(i5);
      const char* halo = halo_base + cur * P2_HBUF;
#pragma unroll 1
      for (int step = 0; step < TH * TW / 32; ++step) {
        // A: dY^T, kouts 16*wave + li, pixels 8g..8g+7; B: this lane's source pixels j = li>>2 (+4), feature quad li&3
        const uint32_t pa = p2_lds_addr(sA + step * 32 * RB + (8 * g + (li >> 2)) * RB + (16 * wave + (li & 3) * 4) * 2);
        const int p0 = step * 32 + 8 * g + (li >> 2), p1 = p0 + 4;
        const uint32_t q0 = p2_lds_addr(halo + ((2 * (p0 / TW) * HC + 2 * (p0 % TW) + (li & 3)) * 4) * 2);
        const uint32_t q1 = p2_lds_addr(halo + ((2 * (p1 / TW) * HC + 2 * (p1 % TW) + (li & 3)) * 4) * 2);
        u32x2_t alo = p2_tr16<0>(pa), ahi = p2_tr16<4 * RB>(pa);
        // the 7 MFMA pairs of a step run three register sets deep: while pair k multiplies, the reads of pairs k+1 and k+2 are in
        // flight (LDS latency ~130 cycles against 32 cycles of MFMA per pair).  LDS reads return in order, so "pair k has
        // arrived" is s_waitcnt lgkmcnt(<reads issued after it>); the wait names pair k's registers, which orders its MFMAs
        // behind it
        u32x2_t x0, x1, x2, x3, y0, y1, y2, y3, z0, z1, z2, z3;
#define P2_RD(f, r0, r1, r2, r3)                                                                       \
        r0 = p2_tr16<P2_BOFF(f)>(q0); r1 = p2_tr16<P2_BOFF(f)>(q1);                                    \
        r2 = p2_tr16<P2_BOFF(f + 1)>(q0); r3 = p2_tr16<P2_BOFF(f + 1)>(q1);
#define P2_MM(f, r0, r1, r2, r3)                                                                       \
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, p2_frag(r0, r1), acc[f], 0, 0, 0);       \
        acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, p2_frag(r2, r3), acc[f + 1], 0, 0, 0);
#define P2_WAIT(n, r0, r1, r2, r3) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        P2_RD(0, x0, x1, x2, x3)
        P2_RD(2, y0, y1, y2, y3)
        P2_RD(4, z0, z1, z2, z3)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(alo), "+v"(ahi), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        const bf16x8_t af = p2_frag(alo, ahi);
        P2_MM(0, x0, x1, x2, x3)   P2_RD(6, x0, x1, x2, x3)
        P2_WAIT(8, y0, y1, y2, y3) P2_MM(2, y0, y1, y2, y3)   P2_RD(8, y0, y1, y2, y3)
        P2_WAIT(8, z0, z1, z2, z3) P2_MM(4, z0, z1, z2, z3)   P2_RD(10, z0, z1, z2, z3)
        P2_WAIT(8, x0, x1, x2, x3) P2_MM(6, x0, x1, x2, x3)   P2_RD(12, x0, x1, x2, x3)
        P2_WAIT(8, y0, y1, y2, y3) P2_MM(8, y0, y1, y2, y3)
        P2_WAIT(4, z0, z1, z2, z3) P2_MM(10, z0, z1, z2, z3)
        P2_WAIT(0, x0, x1, x2, x3) P2_MM(12, x0, x1, x2, x3)
#undef P2_RD
#undef P2_MM
#undef P2_WAIT
      }
      // DMA(t+2) -- the apply pass of the next iteration reads it -- has landed: only DMA(t+3) and DMA(t+4), 7 loads each, may
      // still be out
      if (nx4 < ntiles) __builtin_amdgcn_s_waitcnt(0x0f70 | 14);   // vmcnt(14)
      else __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    p2_barrier();
    cur ^= 1;
    i5 = i5 + 1 < P2_NS ? i5 + 1 : 0;
  }
  if (partials) {
    f32x4_t* sp = partials + (size_t)blockIdx.x * 14 * 256 + tid;
#pragma unroll
    for (int f = 0; f < 14; ++f) sp[f * 256] = acc[f];
    return;
  }
#pragma unroll
  for (int f = 0; f < 14; ++f) {
    const int r = f >> 1, s = (f & 1) * 4 + (li >> 2), c = li & 3;
    if (s < 7 && c < 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {             // a one-workgroup launch: the only writer
        const int k = 16 * wave + 4 * g + j;
        a.dw[((k * 3 + c) * 7 + r) * 7 + s] += acc[f][j];
      }
    }
  }
}

template <typename T, bool INF32, bool POOL>
static hipError_t launch_stem_wgrad_t(const StemWgradArgs& a, const BnBwdArgs& b, hipStream_t st) {
  const int th = cdiv(a.OH, TH), tw = cdiv(a.OW, TW);
  const int ntiles = a.N * th * tw;
  const size_t lds = 2 * (TH * TW * 64 * sizeof(T) + HR * HC * 4 * sizeof(T));
  auto kern = stem_wgrad_kernel<T, INF32, POOL>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  int grid = ntiles < 768 ? ntiles : 768;
  // every workgroup holds a partial of the SAME 64 x 147 weights: slabs + the ordered fold (wgrad_halo.hip), in every dtype
  f32x4_t* slabs = nullptr;
  if (grid > 1) {
    slabs = reinterpret_cast<f32x4_t*>(stream_scratch(st, (size_t)grid * 14 * 256 * sizeof(f32x4_t)));
    if (!slabs) return hipErrorOutOfMemory;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, b, th, tw, ntiles, slabs);
  if (slabs) return launch_stem_wgrad_fold(slabs, a.dw, grid, st);
  return hipGetLastError();
}

hipError_t launch_stem_wgrad(int dtype, const StemWgradArgs& a, hipStream_t st) {
  const BnBwdArgs b{};
  if (dtype == DT_BF16) return a.in_f32 ? launch_stem_wgrad_t<bf16_t, true, false>(a, b, st) : launch_stem_wgrad_t<bf16_t, false, false>(a, b, st);
  return a.in_f32 ? launch_stem_wgrad_t<float, true, false>(a, b, st) : launch_stem_wgrad_t<float, false, false>(a, b, st);
}

// conv1 wgrad with the max-pool + ReLU + bn0 backward apply pass computed on the fly (b: a pool-form descriptor whose reduce
// pass has run; b.dx is not written)
template <bool INF32>
static hipError_t launch_stem_wgrad_pool2(const StemWgradArgs& a, const BnBwdArgs& b, hipStream_t st) {
  const int th = cdiv(a.OH, TH), tw = cdiv(a.OW, TW);
  const int ntiles = a.N * th * tw;
  const size_t lds = P2_NS * P2_STAGE + 2 * P2_HBUF;
  auto kern = stem_wgrad_pool2_kernel<INF32>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  int grid = ntiles < 256 ? ntiles : 256;          // one 12-wave workgroup per CU (143 KiB of LDS)
  f32x4_t* slabs = nullptr;
  if (grid > 1) {
    slabs = reinterpret_cast<f32x4_t*>(stream_scratch(st, (size_t)grid * 14 * 256 * sizeof(f32x4_t)));
    if (!slabs) return hipErrorOutOfMemory;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, st, a, b, th, tw, ntiles, slabs);
  if (slabs) return launch_stem_wgrad_fold(slabs, a.dw, grid, st);
  return hipGetLastError();
}

static int stem_pool_form() {                      // SSLCR_STEM_POOL_FORM=1: the single-role kernel in bf16 mode too (A/B runs)
  static const int v = [] { const char* e = getenv("SSLCR_STEM_POOL_FORM"); return e ? atoi(e) : 2; }();
  return v;
}

hipError_t launch_stem_wgrad_pool(int dtype, const StemWgradArgs& a, const BnBwdArgs& b, hipStream_t st) {
  if (!b.pool_dy || !b.pool_argmax || !b.x || b.C != 64 || b.pH != a.OH || b.pW != a.OW || b.g_in_reduce || b.gout) return hipErrorInvalidValue;
  if (dtype == DT_BF16 && stem_pool_form() == 2) return a.in_f32 ? launch_stem_wgrad_pool2<true>(a, b, st) : launch_stem_wgrad_pool2<false>(a, b, st);
  if (dtype == DT_BF16) return a.in_f32 ? launch_stem_wgrad_t<bf16_t, true, true>(a, b, st) : launch_stem_wgrad_t<bf16_t, false, true>(a, b, st);
  return a.in_f32 ? launch_stem_wgrad_t<float, true, true>(a, b, st) : launch_stem_wgrad_t<float, false, true>(a, b, st);
}

}  // namespace sslcr

// Direct (im2col-free) NHWC convolution on MFMA for gfx950 -- forward, dgrad and eval-mode fused forms.
//
// Replaces the cuDNN/ATen conv2d calls behind torchvision resnet18 (reference call sites
// models/net.py:32,77).  GEMM view: D[kout][pixel] = W[kout][(r,s,c)] * X[(r,s,c)][pixel],
// i.e. the WEIGHTS are the MFMA A operand and the PIXELS the B operand, so that a lane ends up holding
// 4*TK consecutive output channels of ONE pixel -> 16/32-byte channel-contiguous NHWC stores, and the
// per-channel BatchNorm statistics are a 16-lane DPP row reduction of the fp32 accumulators.
//
//   prologue (on the global->LDS path): optional per-channel scale/shift(+ReLU) = the producer's BatchNorm
//            applied in the consumer's load path (zero padding stays zero);
//   main loop: one (tap, 128-byte channel slab) per step, register-prefetched, double-buffered LDS with an
//            XOR-swizzled 16-byte chunk index (conflict-free ds_read_b128 fragment reads);
//   epilogue: optional bias / residual / ReLU / accumulate / strided scatter, per-channel (sum, sumsq) partials.
//
// T = bf16 (v_mfma_f32_16x16x32_bf16) or fp32 (v_mfma_f32_16x16x4_f32, the exact-fp32 parity mode).
#include "kernels.hpp"

namespace sslcr {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

template <typename T, int BP, int BKO>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;   // elements per 16-byte chunk
  constexpr int CE = 8 * EPC;         // channels per 128-byte LDS row = k-step width
  constexpr int TP = BP / 32;         // 16-pixel MFMA tiles per wave (wave owns BP/2 pixels)
  constexpr int TK = BKO / 32;        // 16-kout  MFMA tiles per wave (wave owns BKO/2 kouts)
  constexpr int PR = BP / 32;         // pixel rows each thread stages
  constexpr int WR = BKO / 32;        // weight rows each thread stages
  constexpr int BUF = (BP + BKO) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_scale = reinterpret_cast<float*>(smem + 2 * BUF);
  float* s_shift = s_scale + a.C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 1, wk = wave >> 1;
  const int m0 = blockIdx.x * BP, k0 = blockIdx.y * BKO;
  const int PHW = a.PH * a.PW;
  const int M = a.N * PHW;
  const int chunk = tid & 7, lrow = tid >> 3;
  const bool xform = a.in_scale != nullptr;
  const int pmul = a.pix_mul ? a.pix_mul : 1;

  if (xform) {
    for (int c = tid; c < a.C; c += 256) { s_scale[c] = a.in_scale[c]; s_shift[c] = a.in_shift[c]; }
  }

  int pbase[PR], hb[PR], wb[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    int m = m0 + lrow + 32 * i;
    if (m < M) {
      int n = m / PHW, rem = m - n * PHW;
      int ph = rem / a.PW, pw = rem - ph * a.PW;
      ph = ph * pmul + a.pix_off_h; pw = pw * pmul + a.pix_off_w;
      pbase[i] = n * a.H * a.W;
      hb[i] = a.transposed ? ph + a.pad : ph * a.stride - a.pad;
      wb[i] = a.transposed ? pw + a.pad : pw * a.stride - a.pad;
    } else {
      pbase[i] = -1; hb[i] = 0; wb[i] = 0;
    }
  }
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* wg = reinterpret_cast<const char*>(a.w);
  const int RS = a.R * a.S;
  const int cslabs = a.C / CE;
  const unsigned tmask = a.tap_mask ? a.tap_mask : ((1u << RS) - 1u);
  const int nsteps = __builtin_popcount(tmask) * cslabs;
  int it_tap = __builtin_ctz(tmask), it_slab = 0;      // load_regs() is called for steps 0,1,2,... in order

  u32x4_t preg[PR], wreg[WR];
  unsigned inb_mask = 0;
  int cur_c0 = 0;

  auto load_regs = [&](int) {
    const int tap = it_tap;
    const int c0 = it_slab * CE;
    if (++it_slab == cslabs) {
      it_slab = 0;
      do { ++it_tap; } while (it_tap < RS && !((tmask >> it_tap) & 1u));
    }
    int r = tap / a.S, s = tap - r * a.S;
    cur_c0 = c0;
    inb_mask = 0;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      int h, w; bool ok = pbase[i] >= 0;
      if (a.transposed) {
        int th = hb[i] - r, tw = wb[i] - s;
        h = th / a.stride; w = tw / a.stride;
        ok = ok && th >= 0 && tw >= 0 && (th - h * a.stride) == 0 && (tw - w * a.stride) == 0;
      } else {
        h = hb[i] + r; w = wb[i] + s;
        ok = ok && h >= 0 && w >= 0;
      }
      ok = ok && h < a.H && w < a.W;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (ok) {
        size_t off = ((size_t)(pbase[i] + h * a.W + w) * a.C + c0 + chunk * EPC) * sizeof(T);
        v = ld16(xg + off);
        inb_mask |= 1u << i;
      }
      preg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      int krow = k0 + lrow + 32 * i;
      size_t off = (((size_t)krow * RS + tap) * a.C + c0 + chunk * EPC) * sizeof(T);
      wreg[i] = ld16(wg + off);
    }
  };

  auto store_lds = [&](int buf) {
    char* pb = smem + buf * BUF;
    char* wbuf = pb + BP * 128;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      int row = lrow + 32 * i;
      u32x4_t v = preg[i];
      if (xform && ((inb_mask >> i) & 1u)) {
        float f[EPC];
        Elem<T>::unpack(v, f);
        const int cb = cur_c0 + chunk * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float t = fmaf(f[e], s_scale[cb + e], s_shift[cb + e]);
          f[e] = a.in_relu ? fmaxf(t, 0.f) : t;
        }
        v = Elem<T>::pack(f);
      }
      st16(pb + row * 128 + ((chunk ^ (row & 7)) << 4), v);
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      int row = wperm<TK>(lrow + 32 * i);
      st16(wbuf + row * 128 + ((chunk ^ (row & 7)) << 4), wreg[i]);
    }
  };

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (xform) __syncthreads();          // scale/shift visible before the first transform
  load_regs(0);
  store_lds(0);
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const bool more = step + 1 < nsteps;
    if (more) load_regs(step + 1);
    const char* pb = smem + (step & 1) * BUF;
    const char* wbuf = pb + BP * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ci = kk * 4 + g;
      u32x4_t af[TK], bfr[TP];
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        int row = wk * (BKO / 2) + t * 16 + li;            // fragment order (see wperm)
        af[t] = ld16(wbuf + row * 128 + ((ci ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        int row = wp * (BP / 2) + p * 16 + li;
        bfr[p] = ld16(pb + row * 128 + ((ci ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) Mma<T>::run(af[t], bfr[p], acc[t][p]);
    }
    if (more) store_lds((step + 1) & 1);
    __syncthreads();
  }

  // ---------------- epilogue: lane (li = pixel within 16-tile, g) holds kouts kb .. kb+4*TK-1 of its pixels
  const int kb = k0 + wk * (BKO / 2) + g * (4 * TK);
  if (a.out_scale) conv_scale_acc<TK, TP>(acc, a.out_scale + kb);
  float bias[4 * TK];
#pragma unroll
  for (int j = 0; j < 4 * TK; ++j) bias[j] = a.bias ? a.bias[kb + j] : 0.f;
  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = reinterpret_cast<const char*>(a.residual);
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    int m = m0 + wp * (BP / 2) + p * 16 + li;
    if (m >= M) continue;
    int n = m / PHW, rem = m - n * PHW;
    int ph = rem / a.PW, pw = rem - ph * a.PW;
    ph = ph * pmul + a.pix_off_h; pw = pw * pmul + a.pix_off_w;
    size_t opix = ((size_t)n * a.OH + (size_t)ph * a.osh) * a.OW + (size_t)pw * a.osh;
    size_t off = (opix * a.K + kb) * sizeof(T);
    float v[4 * TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[t * 4 + j] = acc[t][p][j] + bias[t * 4 + j];
#pragma unroll
    for (int q = 0; q < 4 * TK / EPC; ++q) {
      float* vq = v + q * EPC;
      if (rg) {
        float rr[EPC];
        Elem<T>::unpack(ld16(rg + off + q * 16), rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
      }
      if (a.accumulate) {
        float rr[EPC];
        Elem<T>::unpack(ld16(yg + off + q * 16), rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
      }
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] = fmaxf(vq[e], 0.f);
      }
      st16(yg + off + q * 16, Elem<T>::pack(vq));
    }
  }

  if (a.stats) {
    // rows >= M were staged as zeros -> contribute 0.  Sum over this wave's TP*16 pixels.
    float s1[4 * TK], s2[4 * TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int p = 0; p < TP; ++p) { float q = acc[t][p][j]; x1 += q; x2 = fmaf(q, q, x2); }
        s1[t * 4 + j] = row16_sum(x1);
        s2[t * 4 + j] = row16_sum(x2);
      }
    if (li == 0) {
      float* sp = a.stats + ((size_t)(blockIdx.x * 2 + wp) * 2) * a.K + kb;
#pragma unroll
      for (int j = 0; j < 4 * TK; ++j) { sp[j] = s1[j]; sp[a.K + j] = s2[j]; }
    }
  }
}

template <typename T, int BP, int BKO>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t st) {
  const int M = a.N * a.PH * a.PW;
  dim3 grid(cdiv(M, BP), a.K / BKO);
  size_t lds = 2 * (BP + BKO) * 128 + 2 * a.C * sizeof(float);
  auto kern = conv_igemm_kernel<T, BP, BKO>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

static int g_conv_dtype_hint = DT_BF16;   // conv_partials_rows() is asked before the launch: both dtypes tile identically
int conv_partials_rows(const ConvArgs& a) {
  if (conv_h16_ok(g_conv_dtype_hint, a)) return conv_h16_rows(a);
  const int q = conv_halo256_mode(g_conv_dtype_hint, a);
  if (q) return conv_halo256_tiles(a, q) * 4;
  const int tw = conv_halo_tw(g_conv_dtype_hint, a);
  if (tw) return conv_halo_tiles(a, tw) * 2;
  const int bp = conv_dma_bp(g_conv_dtype_hint, a);
  if (bp) return conv_dma_rows(a, bp);
  const int M = a.N * a.PH * a.PW;
  return cdiv(M, conv_tile_bp(a)) * 2;
}

int conv_tile_bp(const ConvArgs& a) {
  const int M = a.N * a.PH * a.PW;
  return M >= 128 * 512 ? 128 : 64;     // keep >= ~512 workgroups on the 256 CUs when M is small
}

template <typename T>
static hipError_t launch_t(const ConvArgs& a, hipStream_t st) {
  constexpr int CE = 8 * Elem<T>::EPC;
  if (a.C % CE != 0 || a.K % 64 != 0) return hipErrorInvalidValue;
  const int bp = conv_tile_bp(a);
  if (bp == 128) {
    if (a.K % 128 == 0) return launch_cfg<T, 128, 128>(a, st);
    return launch_cfg<T, 128, 64>(a, st);
  }
  return launch_cfg<T, 64, 64>(a, st);
}

// the template instance launch_conv() will pick, spelled like rocprofv3 prints it (minus the argument list)
const char* conv_kernel_name(int dtype, const ConvArgs& a) {
  const bool bf = dtype == DT_BF16;
  const bool wide = a.K % 128 == 0;
  if (conv_h16_ok(dtype, a)) return conv_h16_name(dtype, a);
  const int q = conv_halo256_mode(dtype, a);
  if (q && conv_halo256_mode(DT_BF16, a) == q) {
    const bool one = a.C == (bf ? 64 : 32);
    if (q == 16) {
      if (wide) return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 16, 128, 2, false>" : "sslcr::conv3x3_halo256_kernel<float, 16, 128, 2, false>";
      if (one) return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 16, 64, 1, true>" : "sslcr::conv3x3_halo256_kernel<float, 16, 64, 1, true>";
      return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 16, 64, 2, false>" : "sslcr::conv3x3_halo256_kernel<float, 16, 64, 2, false>";
    }
    if (wide) return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 8, 128, 2, false>" : "sslcr::conv3x3_halo256_kernel<float, 8, 128, 2, false>";
    if (one) return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 8, 64, 1, true>" : "sslcr::conv3x3_halo256_kernel<float, 8, 64, 1, true>";
    return bf ? "sslcr::conv3x3_halo256_kernel<unsigned short, 8, 64, 2, false>" : "sslcr::conv3x3_halo256_kernel<float, 8, 64, 2, false>";
  }
  const int tw = q ? 0 : conv_halo_tw(dtype, a);
  if (tw && conv_halo_tw(DT_BF16, a) == tw) return bf ? "sslcr::conv3x3_halo_kernel<unsigned short, ...>" : "sslcr::conv3x3_halo_kernel<float, ...>";
  if (!q && !tw && conv_s2_ok(dtype, a)) return conv_s2_name(a, false);
  if (!q && !tw && conv_s2d_ok(dtype, a)) return conv_s2d_name();
  const int dbp = (q || tw) ? 0 : conv_dma_bp(dtype, a);
  if (dbp && conv_dma_bp(DT_BF16, a) == dbp) return conv_dma_name(dtype, dbp);
  const int bp = conv_tile_bp(a);
  if (bp == 128) {
    if (wide) return bf ? "sslcr::conv_igemm_kernel<unsigned short, 128, 128>" : "sslcr::conv_igemm_kernel<float, 128, 128>";
    return bf ? "sslcr::conv_igemm_kernel<unsigned short, 128, 64>" : "sslcr::conv_igemm_kernel<float, 128, 64>";
  }
  return bf ? "sslcr::conv_igemm_kernel<unsigned short, 64, 64>" : "sslcr::conv_igemm_kernel<float, 64, 64>";
}

int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

// Segments (sslcr_conv_desc.seg_images): which kernels have the form, and where the row ranges come out right.
//   conv3x3_h16 / conv3x3_pp64: the grid is split into nseg groups of workgroups (their statistics rows are per workgroup)
//   conv3x3_halo256: one workgroup per tile, rows in tile order; a tile's images must not straddle a segment
//   conv_dma: rows in pixel order, no prologue; a pixel block must not straddle a segment
bool conv_segments_ok(int dtype, const ConvArgs& a) {
  if (a.seg_images <= 0) return true;
  if (dtype != DT_BF16 || a.N % a.seg_images != 0 || a.transposed || a.par4) return false;
  const int nseg = a.N / a.seg_images;
  if (nseg > 8) return false;
  if (conv_h16_ok(dtype, a)) return true;          // (with mask_x: mask_scale / mask_shift / mask_mean + s * seg_stride)
  if (a.mask_x) return false;
  const int q = conv_halo256_mode(dtype, a);
  if (q) return q == 16 || a.seg_images % 4 == 0;
  if (conv_halo_tw(dtype, a)) return false;
  const int bp = conv_dma_bp(dtype, a);
  if (bp) return !a.in_scale && ((long)a.seg_images * a.PH * a.PW) % bp == 0;
  return false;
}

hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t st) {
  if (!conv_segments_ok(dtype, a)) return hipErrorInvalidValue;
  if (a.out_scale && (!a.bias || a.stats || a.mask_x)) return hipErrorInvalidValue;      // the output scale exists in the bias (eval) epilogues only
  if (conv_h16_ok(dtype, a)) return launch_conv_h16(dtype, a, st);
  if (a.mask_x) return hipErrorInvalidValue;        // the BatchNorm-backward front end exists in the 16x16-tile kernel only
  const int q = conv_halo256_mode(dtype, a);
  if (q && conv_halo256_mode(DT_BF16, a) == q) return launch_conv_halo256(dtype, a, q, st);
  const int tw = q ? 0 : conv_halo_tw(dtype, a);
  if (tw && conv_halo_tw(DT_BF16, a) == tw) return launch_conv_halo(dtype, a, tw, st);   // (same tiling in both dtypes)
  if (!q && !tw && conv_s2_ok(dtype, a)) return launch_conv_s2(a, nullptr, st);       // 3x3 / 2 on 16x16 output tiles (rows as conv_dma's)
  if (!q && !tw && conv_s2d_ok(dtype, a)) return launch_conv_s2d(a, st);             // ... and its dgrad, the four parity classes in one pass
  const int dbp = (q || tw) ? 0 : conv_dma_bp(dtype, a);
  if (dbp && conv_dma_bp(DT_BF16, a) == dbp) return launch_conv_dma(dtype, a, dbp, st);
  if (a.par4) return hipErrorInvalidValue;          // the one-launch parity form exists in the DMA-gather kernel only
  return dtype == DT_BF16 ? launch_t<bf16_t>(a, st) : launch_t<float>(a, st);
}

}  // namespace sslcr

// Weight gradient of the NHWC convolutions on MFMA (autograd wgrad of the resnet18 BasicBlock convs).
//
//   dW[k][r][s][c] += sum_{n,ho,wo} dY[n,ho,wo,k] * act(X)[n, ho*stride-pad+r, wo*stride-pad+s, c]
//
// GEMM view per tap: D[kout][cin] = sum_pixels dY^T[kout][pixel] * X_tap[pixel][cin].  Both operands live in LDS
// pixel-major ([pixel][64 channels], exactly as they sit in NHWC HBM) and the reduction dimension (pixels) is the
// slow one, so bf16 fragments are fetched with the gfx950 transpose read ds_read_b64_tr_b16; fp32 uses scalar reads
// into v_mfma_f32_16x16x4_f32.  One workgroup owns a 64(kout) x 64(cin) x all-taps tile for a slice of the pixels
// and leaves its fp32 result in a slab; the ordered fold of wgrad_halo.hip adds the slices into dW (no atomics: same bits every run).
// The producer's BatchNorm+ReLU is applied to X on the load path, like in the forward conv.
#include "kernels.hpp"

namespace sslcr {

typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct WgFrag;
template <> struct WgFrag<bf16_t> {
  static constexpr int PS = 32;          // pixels per step = one 16x16x32 MFMA depth
  // tile: [PS][128B], the 32-byte column group XORed with bits 1..2 of the row (swz_off); (col0 = first of 16 channels) -> fragment of
  // lane (li, g): eight pixels of channel col0 + li.  Which eight is free as long as dY and X use the same map: rows pl and pl + 8
  // with pl = 16 (g >> 1) + 4 (g & 1) + (li >> 2), so that the eight rows a 32-lane half presents to the transpose read are eight
  // CONSECUTIVE rows -- with the swizzle conflict-free (wgrad_halo.hip).  (Until round 5 this kernel read rows 8 g + (li >> 2) and
  // + 4 of an unswizzled tile: SQ_LDS_BANK_CONFLICT = 65 % of its LDS cycles, profiles/r05_s2_lds_bank_conflicts.txt.)
  __device__ static __forceinline__ int swz_off(int row, int ch) { return (((ch >> 1) ^ ((row >> 1) & 3)) << 5) | ((ch & 1) << 4); }
  __device__ static __forceinline__ bf16x8_t load(const char* tile, int col0, int li, int g) {
    const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
    const char* p = tile + pl * 128 + (((col0 >> 4) ^ ((pl >> 1) & 3)) << 5) + (li & 3) * 8;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 8 * 128));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
};

// KH = 64-kout halves per workgroup (256 threads each).  KH = 2 reads X once per 128 kouts instead of once per 64: these
// launches are HBM-bound on that re-read (r03: 874 MB per launch against 503 MB of tensors at KH = 1).
template <typename T, int TAPS, int KH>
__global__ __launch_bounds__(256 * KH, 2) void wgrad_kernel(const WgradArgs a, int steps_per_split, f32x4_t* partials) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr bool BF = Elem<T>::DT == DT_BF16;
  constexpr int PS = BF ? 32 : 16;             // pixels per step
  constexpr int RB = 64 * sizeof(T);           // bytes per LDS row (64 channels)
  constexpr int CPR = RB / 16;                 // 16-byte chunks per row
  constexpr int TILE = PS * RB;                // 4 KiB
  constexpr int BUF = (KH + TAPS) * TILE;      // KH dY tiles, then one X tile per tap
  constexpr int NT = 256 * KH;
  constexpr int NX = (TAPS + KH - 1) / KH;     // X taps staged per thread: tap KH*i + kh
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kh = tid >> 8;
  const int li = lane & 15, g = lane >> 4;
  // grid x = (kout block, cin block, pixel split) triples, the gx * gy workgroups of one pixel split neighbours on one XCD
  // (wgrad_halo.hip): their re-reads of the same dY / X rows hit its L2
  const int gx = a.K / (64 * KH), gy = a.C / 64, GT = gx * gy, nsplit = (int)gridDim.x / GT;
  int bz, bt;
  if ((nsplit & 7) == 0) {
    const int w = blockIdx.x, grp = w / (8 * GT), r = w - grp * 8 * GT;
    bz = grp * 8 + (r & 7); bt = r >> 3;
  } else {
    bz = (int)blockIdx.x / GT; bt = (int)blockIdx.x - bz * GT;
  }
  const int by = bt / gx, bx = bt - by * gx;
  const int k0 = bx * (64 * KH), c0 = by * 64;
  const int OHW = a.OH * a.OW;
  const int M = a.N * OHW;
  const int row = (tid & 255) / CPR, chunk = (tid & 255) % CPR;
  const bool xform = a.in_scale != nullptr;
  // this thread always stages the same EPC channels: keep their BN scale/shift in registers (LDS is exactly 2 x 40 KiB,
  // so two workgroups share a CU's 160 KiB)
  float r_scale[EPC], r_shift[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    r_scale[e] = xform ? a.in_scale[c0 + chunk * EPC + e] : 1.f;
    r_shift[e] = xform ? a.in_shift[c0 + chunk * EPC + e] : 0.f;
  }

  const int step0 = bz * steps_per_split;
  const int total_steps = (M + PS - 1) / PS;
  int nsteps = total_steps - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;
  if (nsteps <= 0) {                            // (the launcher sizes the splits so that none is empty; a slab must still be defined)
    if (partials) {
      const size_t wg = ((size_t)bz * gy + by) * gx + bx;
      for (int e = 0; e < TAPS * 4; ++e) partials[(wg * (TAPS * 4) + e) * NT + tid] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }

  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* dyg = reinterpret_cast<const char*>(a.dy);

  // pixel -> (image, row, column) once per step and thread: a reciprocal estimate plus one correction instead of
  // integer divisions
  constexpr int TS = TAPS == 9 ? 3 : 1;        // filter width, known from the tap count
  const float rcp_ohw = 1.f / (float)OHW, rcp_ow = 1.f / (float)a.OW;
  auto divmod = [](int x, int d, float rcp, int& q, int& r) {
    q = (int)((float)x * rcp);
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
  };

  u32x4_t yreg, xreg[NX];
  unsigned inb = 0;
  auto load_regs = [&](int s) {
    const int m = (step0 + s) * PS + row;
    inb = 0;
    yreg = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NX; ++i) xreg[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (m < M) {
      yreg = ld16(dyg + ((size_t)m * a.K + k0 + 64 * kh + chunk * EPC) * sizeof(T));
      int n, rem, ho, wo;
      divmod(m, OHW, rcp_ohw, n, rem);
      divmod(rem, a.OW, rcp_ow, ho, wo);
      if constexpr (KH == 2) {
        // address of the (possibly out-of-image) tap (0, 0) position once; a tap adds a uniform offset
        const int h0 = ho * a.stride - a.pad, w0 = wo * a.stride - a.pad;
        const char* xp = xg + (((long)n * a.H + h0) * a.W + w0) * (long)(a.C * (int)sizeof(T)) + (c0 + chunk * EPC) * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int t = KH * i + kh;
          const int r = (t >= TS) + (t >= 2 * TS), s_ = t - r * TS;
          if (t < TAPS && (unsigned)(h0 + r) < (unsigned)a.H && (unsigned)(w0 + s_) < (unsigned)a.W) {
            xreg[i] = ld16(xp + (long)(r * a.W + s_) * (a.C * (int)sizeof(T)));
            inb |= 1u << i;
          }
        }
      } else {                                    // (the narrow form sits at the register limit: per-tap addresses from scratch)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int r = t / TS, s_ = t - r * TS;
          const int h = ho * a.stride - a.pad + r, w = wo * a.stride - a.pad + s_;
          if (h >= 0 && w >= 0 && h < a.H && w < a.W) {
            xreg[t] = ld16(xg + ((size_t)((n * a.H + h) * a.W + w) * a.C + c0 + chunk * EPC) * sizeof(T));
            inb |= 1u << t;
          }
        }
      }
    }
  };
  auto store_lds = [&](int buf) {
    char* b = smem + buf * BUF;
    const int coff = BF ? WgFrag<bf16_t>::swz_off(row, chunk) : chunk * 16;      // (fp32 tiles: scalar reads, no swizzle)
    st16(b + kh * TILE + row * RB + coff, yreg);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int t = KH * i + kh;
      if (TAPS % KH != 0 && t >= TAPS) continue;
      u32x4_t v = xreg[i];
      if (xform && ((inb >> i) & 1u)) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float q = fmaf(f[e], r_scale[e], r_shift[e]);
          f[e] = a.in_relu ? fmaxf(q, 0.f) : q;
        }
        v = Elem<T>::pack(f);
      }
      st16(b + (KH + t) * TILE + row * RB + coff, v);
    }
  };

  f32x4_t acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  load_regs(0);
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) load_regs(s + 1);
    const char* b = smem + (s & 1) * BUF;
    // wave (kh, w) owns cin tile w and the four kout tiles of half kh: 4 A + TAPS B fragments feed 4*TAPS MFMAs per step
    if constexpr (BF) {
      bf16x8_t af[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) af[t4] = WgFrag<bf16_t>::load(b + kh * TILE, 16 * t4, li, g);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const bf16x8_t bfrag = WgFrag<bf16_t>::load(b + (KH + t) * TILE, 16 * wave, li, g);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t4], bfrag, acc[t][t4], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < PS / 4; ++q) {
        const int prow = 4 * q + g;
        float av[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) av[t4] = *reinterpret_cast<const float*>(b + kh * TILE + prow * RB + (16 * t4 + li) * 4);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const float bv = *reinterpret_cast<const float*>(b + (KH + t) * TILE + prow * RB + (16 * wave + li) * 4);
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], bv, acc[t][t4], 0, 0, 0);
        }
      }
    }
    if (more) store_lds((s + 1) & 1);
    __syncthreads();
  }

  // acc[tap][t4]: D[row = kout 64*kh+16*t4+4g+j][col = cin 16*wave+li]  ->  dW[k][tap][c]
  const int RS = a.R * a.S;
  if (partials) {                               // accumulator slab + ordered fold launch, see wgrad_halo.hip
    const size_t wg = ((size_t)bz * gy + by) * gx + bx;
    f32x4_t* sp = partials + wg * (TAPS * 4) * NT + tid;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) sp[(size_t)(t * 4 + t4) * NT] = acc[t][t4];
    return;
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int j = 0; j < 4; ++j) {               // one pixel split: this workgroup is the only writer of its block of dW
        const int k = k0 + 64 * kh + 16 * t4 + 4 * g + j;
        a.dw[((size_t)k * RS + t) * a.C + c0 + 16 * wave + li] += acc[t][t4][j];
      }
}

static bool wgrad_wide(int dtype, const WgradArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
  return on && dtype == DT_BF16 && a.K % 128 == 0;
}

template <typename T, int TAPS, int KH>
static hipError_t launch_w(const WgradArgs& a, hipStream_t st) {
  constexpr int PS = Elem<T>::DT == DT_BF16 ? 32 : 16;
  const int M = a.N * a.OH * a.OW;
  const int total_steps = cdiv(M, PS);
  const int tiles = (a.K / (64 * KH)) * (a.C / 64);
  // one resident round (160 KiB of LDS per CU = two narrow or one wide workgroup): the pixel split sets how many slabs
  // are written and folded, see wgrad_halo.hip
  const int cus = device_cus();
  int splits = cdiv((TAPS == 1 ? 4 * cus : 2 * cus) / KH, tiles);     // 1x1: a small slab per workgroup, more parallel slices pay
  const int max_splits = cdiv(total_steps, 8);      // at least 8 steps per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int sps = cdiv(total_steps, splits);
  splits = cdiv(total_steps, sps);
  const size_t lds = 2 * (KH + TAPS) * 4096;
  auto kern = wgrad_kernel<T, TAPS, KH>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int gx = a.K / (64 * KH), gy = a.C / 64;
  if (TAPS != a.R * a.S) return hipErrorInvalidValue;
  f32x4_t* slabs = nullptr;                      // accumulator slabs + the ordered fold whenever there is more than one pixel split
  if (splits > 1) {
    slabs = reinterpret_cast<f32x4_t*>(stream_scratch(st, (size_t)gx * gy * splits * TAPS * 4 * 256 * KH * sizeof(f32x4_t)));
    if (!slabs) return hipErrorOutOfMemory;
  }
  hipLaunchKernelGGL(kern, dim3(gx * gy * splits), dim3(256 * KH), lds, st, a, sps, slabs);
  if (slabs) return launch_wgrad_fold(slabs, a.dw, a.C, gx, gy, splits, TAPS, KH, st);
  return hipGetLastError();
}

// diagnostics: raw semantics of ds_read_b64_tr_b16 -- lane l reads at byte_addr[l] of a 2 KiB LDS image
__global__ void probe_tr16_kernel(const uint16_t* in, const int* byte_addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  const char* p = reinterpret_cast<const char*>(lds) + byte_addr[threadIdx.x];
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
hipError_t launch_probe_tr16(const uint16_t* in, const int* byte_addr, uint16_t* out, hipStream_t st) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, st, in, byte_addr, out);
  return hipGetLastError();
}

const char* wgrad_kernel_name(int dtype, const WgradArgs& a) {
  const bool bf = dtype == DT_BF16;
  const int tw = wgrad_halo_tw(a);
  const bool wide = bf && a.K % 128 == 0;          // the 128-kout block, 8-wave form (wgrad_halo.hip)
  if (tw && wgrad_dma_used(dtype, a)) {
    const bool xf = a.in_scale != nullptr;
    if (tw == 16) return xf ? "sslcr::wgrad3x3_dma_kernel<16, true>" : "sslcr::wgrad3x3_dma_kernel<16, false>";
    return xf ? "sslcr::wgrad3x3_dma_kernel<8, true>" : "sslcr::wgrad3x3_dma_kernel<8, false>";
  }
  if (tw == 16) return bf ? (wide ? "sslcr::wgrad3x3_halo_kernel<unsigned short, 16, 2>" : "sslcr::wgrad3x3_halo_kernel<unsigned short, 16, 1>") : "sslcr::wgrad3x3_halo_kernel<float, 16, 1>";
  if (tw == 8) return bf ? (wide ? "sslcr::wgrad3x3_halo_kernel<unsigned short, 8, 2>" : "sslcr::wgrad3x3_halo_kernel<unsigned short, 8, 1>") : "sslcr::wgrad3x3_halo_kernel<float, 8, 1>";
  if (wgrad_s2_ok(dtype, a)) return a.OW % 16 == 0 ? "sslcr::wgrad_s2_kernel<16>" : "sslcr::wgrad_s2_kernel<8>";
  const bool kw = wgrad_wide(dtype, a);
  if (a.R == 3) return bf ? (kw ? "sslcr::wgrad_kernel<unsigned short, 9, 2>" : "sslcr::wgrad_kernel<unsigned short, 9, 1>") : "sslcr::wgrad_kernel<float, 9, 1>";
  return bf ? (kw ? "sslcr::wgrad_kernel<unsigned short, 1, 2>" : "sslcr::wgrad_kernel<unsigned short, 1, 1>") : "sslcr::wgrad_kernel<float, 1, 1>";
}

hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t st) {
  const int tw = wgrad_halo_tw(a);
  if (tw) return launch_wgrad_halo(dtype, a, tw, st);
  if (a.seg_images > 0 && a.seg_images < a.N) return hipErrorInvalidValue;      // per-segment prologue: halo kernel only
  if (wgrad_s2_ok(dtype, a)) return launch_wgrad_s2(a, st);                     // 3x3 / 2 on 4x16-tileable output maps: parity-plane halo form
  const bool three = a.R == 3;
  if (wgrad_wide(dtype, a)) return three ? launch_w<bf16_t, 9, 2>(a, st) : launch_w<bf16_t, 1, 2>(a, st);
  if (dtype == DT_BF16) return three ? launch_w<bf16_t, 9, 1>(a, st) : launch_w<bf16_t, 1, 1>(a, st);
  return three ? launch_w<float, 9, 1>(a, st) : launch_w<float, 1, 1>(a, st);
}

}  // namespace sslcr

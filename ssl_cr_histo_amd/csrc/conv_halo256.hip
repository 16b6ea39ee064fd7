// 3x3 / stride 1 / pad 1 NHWC convolution, 256-pixel LDS-halo form: the main MFMA kernel of the engine
// (ResNet18 layer1-4 block convs forward, and their dgrads through tap-flipped [C][R][S][K] packs).
//
// Why this shape.  At 2048 MAC/clk/CU the operand paths are the limit, not the matrix pipe: global->LDS moves 64 B/clk,
// LDS writes ~80 B/clk, LDS reads 256 B/clk.  A workgroup computing P pixels x Kb kouts needs 2/P bytes of weights and
// ~2*1.3/(9*Kb) bytes of activations per MAC, so the tile is made big in P: 512 threads (8 waves, 2 per SIMD) own a
// 16x16-pixel (or 4 images x 8x8) x BKO-kout tile.  Per 128-byte channel slab the 18x18 input halo is staged in LDS once
// (producer BatchNorm+ReLU applied on the way; zero padding stays zero) and serves all nine taps at shifted offsets; the
// weights stream through a double-buffered LDS ring one tap (BKO x 128 B) at a time: 16 B/clk of global->LDS traffic per
// CU instead of 64.  Fragment reads are XOR-swizzled ds_read_b128; epilogue conventions are those of conv_igemm.hip.
#include "kernels.hpp"

namespace sslcr {

template <typename T> struct MmaQ;
template <> struct MmaQ<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaQ<float> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// WK = waves along kouts (2: 512 threads, each wave 64 px x BKO/2 kouts; 1: 256 threads, each wave 64 px x BKO kouts --
// used for K = 64 so that a wave still owns a 64x64 register tile and LDS reads stay at 16 MAC per byte)
// ONE = the layer has a single channel slab (C == 64 bf16): no halo prefetch registers are kept across the tap loop
template <typename T, int TW, int BKO, int WK, bool ONE>
__global__ __launch_bounds__(256 * WK, 2) void conv3x3_halo256_kernel(const ConvArgs a, const int tile0) {
  constexpr int NT = 256 * WK;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CE = 8 * EPC;
  constexpr int TH = TW;                      // 16x16 tile of one image, or 8x8 tiles of four images
  constexpr int NI = 256 / (TH * TW);
  constexpr int HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;           // 324 or 400 halo pixels
  constexpr int NLD = (HP * 8 + NT - 1) / NT;   // 16-byte halo staging loads per thread
  constexpr int WLD = BKO * 8 / NT;          // 16-byte weight staging loads per thread per tap (1 or 2)
  constexpr int TK = BKO / (16 * WK), TP = 4;
  constexpr int HBUF = HP * 128, WBUF = BKO * 128;
  // TPB = taps per barrier.  512-thread configs are alone on their CU (registers), so LDS is free: the weight ring holds
  // 2 x 3 taps and the workgroup synchronises once per filter ROW instead of once per tap; a prefetched tap is written into
  // the other half of the ring without a barrier, so an L2 round trip only ever stalls the wave that issued it.
  constexpr int TPB = (WK == 2 && !ONE) ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_halo = smem;
  char* s_w = smem + HBUF;                    // [2][BKO][128 B]
  float* s_scale = reinterpret_cast<float*>(smem + HBUF + 2 * TPB * WBUF);
  float* s_shift = s_scale + a.C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 3, wk = wave >> 2;
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  int tile = blockIdx.x + tile0;             // (tile0: a launch may cover a sub-range of the tiles, see launch_qt)
  const int tw_i = tile % tiles_w; tile /= tiles_w;
  const int th_i = tile % tiles_h;
  const int n0 = (tile / tiles_h) * NI;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int k0 = blockIdx.y * BKO;
  const bool xform = a.in_scale != nullptr;
  if (xform) {
    // segments (sslcr_conv_desc.seg_images): a tile's NI images sit in one segment (seg_images % NI == 0, conv_segments_ok)
    const size_t so = a.seg_images > 0 ? (size_t)(n0 / a.seg_images) * a.seg_stride : 0;
    for (int c = tid; c < a.C; c += NT) { s_scale[c] = a.in_scale[so + c]; s_shift[c] = a.in_shift[so + c]; }
  }

  const int chunk = tid & 7;
  int src_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int hp = (tid >> 3) + (NT / 8) * i;
    if (hp < HP) {
      const int ni = hp / (HH * HWD), rem = hp - ni * (HH * HWD);
      const int hr = rem / HWD, hc = rem - hr * HWD;
      const int h = h0 - 1 + hr, w = w0 - 1 + hc;
      src_off[i] = (h >= 0 && w >= 0 && h < a.H && w < a.W) ? ((n0 + ni) * a.H + h) * a.W + w : -1;
    } else {
      src_off[i] = -2;
    }
  }
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* wg = reinterpret_cast<const char*>(a.w);
  const int nslabs = a.C / CE;

  u32x4_t hreg[NLD], wreg[WLD];
  auto load_halo = [&](int slab) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (src_off[i] >= 0) v = ld16(xg + ((size_t)src_off[i] * a.C + slab * CE + chunk * EPC) * sizeof(T));
      hreg[i] = v;
    }
  };
  auto halo_key = [&](int hp) {               // see the comment at hbase below
    if (TW == 16) return hp & 7;
    return (hp % HWD) & 7;                     // HH * HWD is a multiple of HWD: the halo column
  };
  auto store_halo = [&](int slab) {
    // this thread's EPC channels of the slab: read scale/shift ONCE (the compiler cannot hoist LDS reads over the LDS stores)
    float sc[EPC], sh[EPC];
    if (xform) {
      const int cb = slab * CE + chunk * EPC;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc[e] = s_scale[cb + e]; sh[e] = s_shift[cb + e]; }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (src_off[i] == -2) continue;
      u32x4_t v = hreg[i];
      if (xform && src_off[i] >= 0) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float t = fmaf(f[e], sc[e], sh[e]);
          f[e] = a.in_relu ? fmaxf(t, 0.f) : t;
        }
        v = Elem<T>::pack(f);
      }
      const int hp = (tid >> 3) + (NT / 8) * i;
      st16(s_halo + hp * 128 + ((chunk ^ halo_key(hp)) << 4), v);
    }
  };
  // weights of one (slab, tap): rows k0 .. k0+BKO-1, 128 B each; thread -> (row = tid>>3 (+64), chunk)
  auto load_w = [&](int slab, int tap) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = (tid >> 3) + (NT / 8) * i;
      wreg[i] = ld16(wg + (((size_t)(k0 + row) * 9 + tap) * a.C + slab * CE + chunk * EPC) * sizeof(T));
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = wperm<TK>((tid >> 3) + (NT / 8) * i);
      st16(s_w + buf * WBUF + row * 128 + ((chunk ^ (row & 7)) << 4), wreg[i]);
    }
  };

  auto load_w_to = [&](u32x4_t (&dst)[WLD], int slab, int tap) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = (tid >> 3) + (NT / 8) * i;
      dst[i] = ld16(wg + (((size_t)(k0 + row) * 9 + tap) * a.C + slab * CE + chunk * EPC) * sizeof(T));
    }
  };
  auto store_w_from = [&](const u32x4_t (&src)[WLD], int buf) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = wperm<TK>((tid >> 3) + (NT / 8) * i);
      st16(s_w + buf * WBUF + row * 128 + ((chunk ^ (row & 7)) << 4), src[i]);
    }
  };

  // XOR key of a halo pixel's 16-byte chunks.  16-wide tiles: pixel index & 7 (rows are 18 pixels = even, a fragment's 16 lanes
  // are 16 consecutive pixels).  8-wide tiles: a fragment's 16 lanes are TWO rows of 8 pixels one 10-pixel halo row apart, and
  // with the linear key the second row's keys are the first row's shifted by 2 -- every fragment read was a 2-way bank conflict
  // (27 % of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT).  key = halo COLUMN & 7 is conflict-free for every tap
  // (enumerated over all lane groups of ds_read_b128, like the 18-pixel-pitch finding in conv_h16.hip).
  int hbase[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int pg = wp * 4 + p;                 // 16-pixel group 0..15 of the tile
    if (TW == 16) {
      hbase[p] = pg * HWD + li;
    } else {
      hbase[p] = (pg >> 2) * (HH * HWD) + (2 * (pg & 3) + (li >> 3)) * HWD + (li & 7);
    }
  }
  int arow[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) arow[t] = wk * (BKO / WK) + t * 16 + li;      // fragment order (see wperm)

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Weight pipeline: the tap computed now sits in LDS, the NEXT tap waits in registers (wnxt) and the one after that is in
  // flight (wreg) -- an L2 round trip gets two tap-times (~2x512 MFMA cycles per wave) instead of one to land.
  // Two register sets alternate by tap parity (static indices in the unrolled tap loop; no register copies, which would
  // force a wait on the load in flight); nine taps per slab is odd, so the sets are swapped once per slab boundary.
  u32x4_t wset[2][WLD];
  const int jtot = nslabs * 9;
  if (xform) __syncthreads();
  load_halo(0);
  store_halo(0);
#pragma unroll
  for (int t = 0; t < TPB; ++t) {                // ring half 0 <- taps 0..TPB-1
    load_w(0, t);
    store_w_from(wreg, t);
  }
  constexpr bool DEEP = ONE;     // 2-tap-ahead prefetch only where registers allow it (multi-slab configs spill with it)
  if (DEEP) {
    load_w(0, 1);                                // jtot >= 9
#pragma unroll
    for (int i = 0; i < WLD; ++i) wset[0][i] = wreg[i];
  }
  __syncthreads();

  int wb = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const bool more = ONE ? false : (slab + 1 < nslabs);
    if (more) load_halo(slab + 1);            // in flight during the nine taps of this slab
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int j = slab * 9 + tap;
      const bool wnext = tap < 8 || more;        // tap j+1 exists (held in wnxt)
      const bool wnext2 = j + 2 < jtot;          // tap j+2 exists: put it in flight now
      u32x4_t (&held)[WLD] = wset[DEEP ? (tap & 1) : 0];          // tap j+1 (DEEP: loaded one tap ago)
      u32x4_t (&flight)[WLD] = wset[DEEP ? ((tap & 1) ^ 1) : 0];  // DEEP: tap j+2 goes in flight now
      if (DEEP) {
        if (wnext2) {
          if (tap < 7) load_w_to(flight, slab, tap + 2); else load_w_to(flight, slab + 1, tap - 7);
        }
      } else if (j + TPB < jtot) {                 // tap j+TPB: same ring slot, other half
        if (tap + TPB < 9) load_w_to(held, slab, tap + TPB); else load_w_to(held, slab + 1, tap + TPB - 9);
      }
      const int r = tap / 3, s = tap - 3 * r;
      const int toff = r * HWD + s;
      const int slot = tap % TPB;
      const char* wbuf = s_w + (wb * TPB + slot) * WBUF;
#pragma unroll((TW == 8 && !ONE && BKO == 128) ? 1 : 2)      // the 4-image x 8x8 wide config is register-bound
      for (int kk = 0; kk < 2; ++kk) {
        const int ci = kk * 4 + g;
        u32x4_t af[TK], bfr[TP];
#pragma unroll
        for (int t = 0; t < TK; ++t) af[t] = ld16(wbuf + arow[t] * 128 + ((ci ^ (arow[t] & 7)) << 4));
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const int hp = hbase[p] + toff;
          const int key = TW == 16 ? (hp & 7) : (((li & 7) + s) & 7);
          bfr[p] = ld16(s_halo + hp * 128 + ((ci ^ key) << 4));
        }
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
          for (int p = 0; p < TP; ++p) MmaQ<T>::run(af[t], bfr[p], acc[t][p]);
      }
      if (tap == 8 && more) {
        __syncthreads();                      // every wave is done with this slab's halo
        store_halo(slab + 1);
      }
      if (DEEP) {
        if (wnext) store_w_from(held, wb ^ 1);
      } else if (j + TPB < jtot) {
        store_w_from(held, (wb ^ 1) * TPB + slot);
      }
      if (slot == TPB - 1) {
        __syncthreads();
        wb ^= 1;
      }
    }
    if (DEEP && more) {      // after 9 taps the next tap's weights sit in wset[1]: make them wset[0] again
#pragma unroll
      for (int i = 0; i < WLD; ++i) { const u32x4_t t = wset[0][i]; wset[0][i] = wset[1][i]; wset[1][i] = t; }
    }
  }

  // ---------------- epilogue
  const int kb = k0 + wk * (BKO / WK) + g * (4 * TK);
  if (a.out_scale) conv_scale_acc<TK, TP>(acc, a.out_scale + kb);
  float bias[4 * TK];
  if (a.bias) {
#pragma unroll
    for (int q = 0; q < TK; ++q) {
      const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(a.bias + kb + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) bias[4 * q + j] = b4[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4 * TK; ++j) bias[j] = 0.f;
  }
  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = reinterpret_cast<const char*>(a.residual);
  size_t off[TP];
  bool ok[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int pg = wp * 4 + p;
    int n, h, w;
    if (TW == 16) { n = n0; h = h0 + pg; w = w0 + li; }
    else { n = n0 + (pg >> 2); h = h0 + 2 * (pg & 3) + (li >> 3); w = w0 + (li & 7); }
    off[p] = ((((size_t)n * a.H + h) * a.W + w) * a.K + kb) * sizeof(T);
    ok[p] = true;
  }
  conv_store_tile<T, TK, TP>(acc, bias, off, ok, yg, rg, false, a.relu != 0);
  if (a.stats) {
    float s1[4 * TK], s2[4 * TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int p = 0; p < TP; ++p) { float q = acc[t][p][j]; x1 += q; x2 = fmaf(q, q, x2); }
        s1[t * 4 + j] = x1;
        s2[t * 4 + j] = x2;
      }
    float* sp = a.stats + ((size_t)((blockIdx.x + tile0) * 4 + wp) * 2) * a.K + kb;
    if constexpr (4 * TK == 16) {
      // sums over the 16 lanes of the DPP row, sixteen values at a time (common.hpp: 32 DPP adds per set where 16 row16_sum calls are
      // 128 instructions as compiled): quad q of the row ends up with values 4q..4q+3 and stores that 16-byte piece
      row16_fold16(s1);
      row16_fold16(s2);
      if ((li & 3) == 0) {
        const int q4 = (li >> 2) * 4;
        *reinterpret_cast<f32x4_t*>(sp + q4) = f32x4_t{s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4_t*>(sp + a.K + q4) = f32x4_t{s2[0], s2[1], s2[2], s2[3]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4 * TK; ++j) { s1[j] = row16_sum(s1[j]); s2[j] = row16_sum(s2[j]); }
      if (li == 0) {
#pragma unroll
        for (int j = 0; j < 4 * TK; ++j) { sp[j] = s1[j]; sp[a.K + j] = s2[j]; }
      }
    }
  }
}

// 0: not applicable; 16: 16x16 tiles; 8: four images x 8x8
int conv_halo256_mode(int dtype, const ConvArgs& a) {
  if (a.pix_mul > 1 || a.tap_mask) return 0;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.transposed || a.accumulate || a.osh != 1) return 0;
  if (a.PH != a.H || a.PW != a.W || a.OH != a.H || a.OW != a.W) return 0;
  const int ce = dtype == DT_BF16 ? 64 : 32;
  if (a.C % ce != 0 || a.K % 64 != 0) return 0;
  if (a.H % 16 == 0 && a.W % 16 == 0) return 16;
  if (a.H == 8 && a.W == 8 && a.N % 4 == 0) return 8;
  return 0;
}

int conv_halo256_tiles(const ConvArgs& a, int mode) {
  return mode == 16 ? a.N * (a.H / 16) * (a.W / 16) : a.N / 4;
}

template <typename T, int TW, int BKO, int WK, bool ONE>
static hipError_t launch_q(const ConvArgs& a, hipStream_t st, int tile0 = 0, int ntiles = -1) {
  constexpr int HP = (256 / (TW * TW)) * (TW + 2) * (TW + 2);
  constexpr int TPB = (WK == 2 && !ONE) ? 3 : 1;
  const size_t lds = HP * 128 + 2 * TPB * BKO * 128 + 2 * a.C * sizeof(float);
  auto kern = conv3x3_halo256_kernel<T, TW, BKO, WK, ONE>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  dim3 grid(ntiles < 0 ? conv_halo256_tiles(a, TW) : ntiles, a.K / BKO);
  hipLaunchKernelGGL(kern, grid, dim3(256 * WK), lds, st, a, tile0);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_qt(const ConvArgs& a, int mode, hipStream_t st) {
  const bool wide = a.K % 128 == 0;
  const bool one = a.C == 8 * Elem<T>::EPC;
  if (mode == 16) {
    if (wide) return launch_q<T, 16, 128, 2, false>(a, st);
    return one ? launch_q<T, 16, 64, 1, true>(a, st) : launch_q<T, 16, 64, 2, false>(a, st);
  }
  if (wide) {
    // one workgroup per CU and (tile, 128-kout block) item: when the last round would occupy at most half the CUs (N=640 at
    // layer4: 640 items = 2.5 rounds of 256), its tiles run as 64-kout half-items on all of them instead -- ~0.6 of a round
    // instead of a whole one
    const int cus = device_cus();
    const int tiles = conv_halo256_tiles(a, 8), kb = a.K / 128, items = tiles * kb, rem = items % cus;
    if (rem != 0 && 2 * rem <= cus && rem % kb == 0) {          // (also the whole launch when it has at most cus / 2 items: N=128)
      const int tail = rem / kb;
      if (tiles > tail) {
        hipError_t e = launch_q<T, 8, 128, 2, false>(a, st, 0, tiles - tail);
        if (e != hipSuccess) return e;
      }
      return launch_q<T, 8, 64, 2, false>(a, st, tiles - tail, tail);
    }
    return launch_q<T, 8, 128, 2, false>(a, st);
  }
  return one ? launch_q<T, 8, 64, 1, true>(a, st) : launch_q<T, 8, 64, 2, false>(a, st);
}

hipError_t launch_conv_halo256(int dtype, const ConvArgs& a, int mode, hipStream_t st) {
  return dtype == DT_BF16 ? launch_qt<bf16_t>(a, mode, st) : launch_qt<float>(a, mode, st);
}

}  // namespace sslcr

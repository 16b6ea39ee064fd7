// 3x3 / stride 1 / pad 1 NHWC convolution, LDS-staged halo form (im2col-free direct conv): the 12 (of 20) ResNet18 convs
// that carry 77 % of the forward FLOPs, and -- with tap-flipped [C][R][S][K] weights -- their dgrads.
//
// A workgroup owns an (NI x 8 x TW)-pixel x BKO-kout output tile.  Per 128-byte channel slab the (8+2) x (TW+2) input
// halo is staged in LDS ONCE (producer BatchNorm+ReLU applied on the way, zero padding stays zero) and all nine taps read
// their MFMA B fragments from it at shifted pixel offsets, so the activation operand crosses L2->LDS once instead of nine
// times.  The weight (A) fragments never touch LDS: each lane loads its own 16-byte pieces straight from the L2-resident
// [K][R][S][C] pack, one tap ahead of use.  Epilogue and fragment/kout-permutation conventions are those of conv_igemm.hip.
#include "kernels.hpp"

namespace sslcr {

template <typename T> struct MmaH;
template <> struct MmaH<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaH<float> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

template <typename T, int TW, int BKO>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CE = 8 * EPC;                 // channels per 128-byte slab
  constexpr int TH = 8;
  constexpr int NI = 128 / (TH * TW);         // images per tile (1 for TW=16, 2 for TW=8)
  constexpr int HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;           // halo pixels
  constexpr int NLD = (HP * 8 + 255) / 256;   // 16-byte staging loads per thread
  constexpr int TK = BKO / 32, TP = 4;
  constexpr int HBUF = HP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_scale = reinterpret_cast<float*>(smem + 2 * HBUF);
  float* s_shift = s_scale + a.C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 1, wk = wave >> 1;
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  int tile = blockIdx.x;
  const int tw_i = tile % tiles_w; tile /= tiles_w;
  const int th_i = tile % tiles_h;
  const int n0 = (tile / tiles_h) * NI;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int k0 = blockIdx.y * BKO;
  const bool xform = a.in_scale != nullptr;
  if (xform)
    for (int c = tid; c < a.C; c += 256) { s_scale[c] = a.in_scale[c]; s_shift[c] = a.in_shift[c]; }

  // ---- halo staging plan of this thread: element idx = tid + 256*i -> (halo pixel, 16-byte chunk)
  int src_off[NLD];          // pixel index in the source tensor, -1 = zero padding, -2 = no element
  int dst_off[NLD];          // LDS byte offset (swizzled)
  const int chunk = tid & 7;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int hp = (tid >> 3) + 32 * i;
    if (hp < HP) {
      const int ni = hp / (HH * HWD), rem = hp - ni * (HH * HWD);
      const int hr = rem / HWD, hc = rem - hr * HWD;
      const int h = h0 - 1 + hr, w = w0 - 1 + hc;
      src_off[i] = (h >= 0 && w >= 0 && h < a.H && w < a.W) ? ((n0 + ni) * a.H + h) * a.W + w : -1;
      dst_off[i] = hp * 128 + ((chunk ^ (hp & 7)) << 4);
    } else {
      src_off[i] = -2; dst_off[i] = 0;
    }
  }
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* wg = reinterpret_cast<const char*>(a.w);
  const int nslabs = a.C / CE;

  u32x4_t hreg[NLD];
  auto load_halo = [&](int slab) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (src_off[i] >= 0) v = ld16(xg + ((size_t)src_off[i] * a.C + slab * CE + chunk * EPC) * sizeof(T));
      hreg[i] = v;
    }
  };
  auto store_halo = [&](int slab, int buf) {
    char* hb = smem + buf * HBUF;
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
      st16(hb + dst_off[i], v);
    }
  };

  // ---- fragment addressing
  int hbase[TP];             // halo pixel index of this lane's pixel for tap (0,0), per pixel group
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int pg = wp * 4 + p;
    if (TW == 16) {
      hbase[p] = pg * HWD + li;
    } else {
      hbase[p] = (pg >> 2) * (HH * HWD) + (2 * (pg & 3) + (li >> 3)) * HWD + (li & 7);
    }
  }
  size_t wrow[TK];           // byte offset of this lane's weight row (tap 0, channel 0)
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    const int kout = k0 + wk * (BKO / 2) + (li >> 2) * (4 * TK) + t * 4 + (li & 3);
    wrow[t] = (size_t)kout * 9 * a.C * sizeof(T);
  }
  auto load_a = [&](u32x4_t (&af)[TK][2], int slab, int tap) {
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        af[t][kk] = ld16(wg + wrow[t] + ((size_t)tap * a.C + slab * CE + (kk * 4 + g) * EPC) * sizeof(T));
  };

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t afA[TK][2], afB[TK][2];
  if (xform) __syncthreads();
  load_halo(0);
  load_a(afA, 0, 0);
  store_halo(0, 0);
  __syncthreads();

  for (int slab = 0; slab < nslabs; ++slab) {
    const bool more = slab + 1 < nslabs;
    if (more) load_halo(slab + 1);
    const char* hb = smem + (slab & 1) * HBUF;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // prefetch the next tap's weight fragments (next slab's tap 0 after tap 8) into the other register set
      u32x4_t (&cur)[TK][2] = (tap & 1) ? afB : afA;
      u32x4_t (&nxt)[TK][2] = (tap & 1) ? afA : afB;
      if (tap < 8) load_a(nxt, slab, tap + 1);
      else if (more) load_a(nxt, slab + 1, 0);
      const int r = tap / 3, s = tap - 3 * r;
      const int toff = r * HWD + s;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ci = kk * 4 + g;
        u32x4_t bfr[TP];
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const int hp = hbase[p] + toff;
          bfr[p] = ld16(hb + hp * 128 + ((ci ^ (hp & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
          for (int p = 0; p < TP; ++p) MmaH<T>::run(cur[t][kk], bfr[p], acc[t][p]);
      }
    }
    // 9 taps = odd count: the register set holding the next slab's tap 0 alternates; swap so that afA is always "current"
    if (more) {
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { u32x4_t tmp = afA[t][kk]; afA[t][kk] = afB[t][kk]; afB[t][kk] = tmp; }
      store_halo(slab + 1, (slab + 1) & 1);
    }
    __syncthreads();
  }

  // ---------------- epilogue (same conventions as conv_igemm_kernel)
  const int kb = k0 + wk * (BKO / 2) + g * (4 * TK);
  if (a.out_scale) conv_scale_acc<TK, TP>(acc, a.out_scale + kb);
  float bias[4 * TK];
#pragma unroll
  for (int j = 0; j < 4 * TK; ++j) bias[j] = a.bias ? a.bias[kb + j] : 0.f;
  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = reinterpret_cast<const char*>(a.residual);
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int pg = wp * 4 + p;
    int n, h, w;
    if (TW == 16) { n = n0; h = h0 + pg; w = w0 + li; }
    else { n = n0 + (pg >> 2); h = h0 + 2 * (pg & 3) + (li >> 3); w = w0 + (li & 7); }
    const size_t off = ((((size_t)n * a.H + h) * a.W + w) * a.K + kb) * sizeof(T);
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
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] = fmaxf(vq[e], 0.f);
      }
      st16(yg + off + q * 16, Elem<T>::pack(vq));
    }
  }
  if (a.stats) {
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

// usable when: plain 3x3/1 pad 1, same-size output, no scatter/accumulate, spatial dims tile exactly
int conv_halo_tw(int dtype, const ConvArgs& a) {
  if (a.pix_mul > 1 || a.tap_mask) return 0;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.transposed || a.accumulate || a.osh != 1) return 0;
  if (a.PH != a.H || a.PW != a.W || a.OH != a.H || a.OW != a.W) return 0;
  const int ce = dtype == DT_BF16 ? 64 : 32;
  if (a.C % ce != 0 || a.K % 64 != 0 || a.H % 8 != 0) return 0;
  if (a.W % 16 == 0) return 16;
  if (a.W % 8 == 0 && a.N % 2 == 0) return 8;
  return 0;
}

int conv_halo_tiles(const ConvArgs& a, int tw) {
  const int ni = 128 / (8 * tw);
  return (a.N / ni) * (a.H / 8) * (a.W / tw);
}

template <typename T, int TW, int BKO>
static hipError_t launch_h(const ConvArgs& a, hipStream_t st) {
  constexpr int HP = (128 / (8 * TW)) * 10 * (TW + 2);
  const size_t lds = 2 * HP * 128 + 2 * a.C * sizeof(float);
  dim3 grid(conv_halo_tiles(a, TW), a.K / BKO);
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, TW, BKO>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_ht(const ConvArgs& a, int tw, hipStream_t st) {
  const bool wide = a.K % 128 == 0;
  if (tw == 16) return wide ? launch_h<T, 16, 128>(a, st) : launch_h<T, 16, 64>(a, st);
  return wide ? launch_h<T, 8, 128>(a, st) : launch_h<T, 8, 64>(a, st);
}

hipError_t launch_conv_halo(int dtype, const ConvArgs& a, int tw, hipStream_t st) {
  return dtype == DT_BF16 ? launch_ht<bf16_t>(a, tw, st) : launch_ht<float>(a, tw, st);
}

}  // namespace sslcr

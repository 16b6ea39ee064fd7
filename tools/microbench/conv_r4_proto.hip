// 3x3 / stride 1 / pad 1 NHWC convolution on 16x16 tiles, "one wave per SIMD" form (round-4 review, item 3): four waves of 512
// registers, no barrier and no LDS traffic for the weights in the tap loop.
//
//   * a wave owns 128 pixels (8 rows of the tile) x 64 kouts: TK = 4, TP = 8 -- 12 fragments per 32 MFMAs where the eight-wave
//     form (conv_h16.hip, 64 x 64 per wave) reads 8 per 16;
//   * the WEIGHTS never touch LDS: a lane loads its 16 bytes of an A fragment straight from the [K][R][S][C] pack (L2-resident;
//     global_load_dwordx4, three k-steps ahead in registers) -- only the pixel fragments are ds_read_b128, 8 per 32 MFMAs;
//   * LDS holds two 18x24-pitch halos: the next stage's halo is loaded into registers early in a stage, (transformed and) written to
//     the other buffer in the middle of it; two bare barriers per stage (one frees the buffer before the writes, one publishes it),
//     neither drains the MFMA stream -- fragments are in registers a step ahead;
//   * a lone wave has nobody to cover its output stage, so there are TWO accumulator sets: the finished item's set is packed and
//     stored in pieces between the next item's MFMAs.
// Serves (bf16): C % 64 == 0, K % 128 == 0, 16x16-tileable maps, no input transform (the teacher's folded convs, conv1 of a block's
// train forward, conv1's dgrad with the identity gradient as residual).
#include <type_traits>

#include "../../ssl_cr_histo_amd/csrc/kernels.hpp"

namespace sslcr {

#define R4_BAR_PUBLISH() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define R4_BAR_FREE() asm volatile("s_barrier" ::: "memory")

// MODE 0: plain store (+ statistics rows when a.stats); 1: bias (+ residual) (+ ReLU)
template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_r4_kernel(const ConvArgs a, const int tiles_total, const int n_items, const int kshift) {
  typedef bf16_t T;
  constexpr int BKO = 128, TK = 4, TP = 8, CE = 64, PITCH = 24;
  constexpr int HBUF = 18 * PITCH * 128;
  constexpr int NLD = 11;                      // 324 halo pixels x 8 chunks over 256 threads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_halo = smem;                         // [2][HBUF]
  float* s_bias = reinterpret_cast<float*>(smem + 2 * HBUF);      // MODE 1: [K]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 1, wk = wave >> 1;
  const int tiles_w = a.W / 16, tiles_h = a.H / 16;
  const int G = gridDim.x, lb = blockIdx.x;
  const int first = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);
  if (first >= n_items) return;
  const int nslabs = a.C / CE;
  if (MODE == 1) {
    for (int i = tid; i < a.K; i += 256) s_bias[i] = a.bias ? a.bias[i] : 0.f;
  }
  const float out_lo = a.relu ? 0.f : -__builtin_inff();

  const bool kfast = kshift >= 0;
  struct Geo { int origin, k0, tile, n0, h0, w0; unsigned long long out; };
  auto geom = [&](int item) {
    Geo q;
    const int kbi = kfast ? item & ((1 << kshift) - 1) : item / tiles_total;
    q.tile = kfast ? item >> kshift : item - kbi * tiles_total;
    q.k0 = kbi * BKO;
    int t = q.tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    q.n0 = t / tiles_h;
    q.h0 = th_i * 16; q.w0 = tw_i * 16;
    q.origin = (q.n0 * a.H + q.h0) * a.W + q.w0;
    q.out = (unsigned long long)((q.h0 == 0) | ((q.h0 + 16 >= a.H) << 1) | ((q.w0 == 0) << 2) | ((q.w0 + 16 >= a.W) << 3)) * 0x1111111111111111ull;
    return q;
  };

  // ---- halo staging roles (as conv3x3_h16: pitch 24, swizzle key = halo pixel & 7)
  const int chunk = tid & 7;
  int rel[NLD], st_off[NLD];
  unsigned long long edge = 0;
  unsigned hvalid = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int sp = (tid >> 3) + 32 * i;
    rel[i] = 0; st_off[i] = 0;
    if (sp < 324) {
      const int hr = sp / 18, hc = sp - hr * 18;
      rel[i] = (hr - 1) * a.W + hc - 1;
      const int hp = hr * PITCH + hc;
      st_off[i] = hp * 128 + ((chunk ^ (hp & 7)) << 4);
      hvalid |= 1u << i;
      edge |= (unsigned long long)((hr == 0) | ((hr == 17) << 1) | ((hc == 0) << 2) | ((hc == 17) << 3)) << (4 * i);
    }
  }
  const char* xg = reinterpret_cast<const char*>(a.x) + (size_t)chunk * 16;
  u32x4_t hreg[NLD];
  auto load_halo = [&](const Geo& q, int slab) {
    const unsigned long long bad = edge & q.out;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      const int idx = q.origin + (ok ? rel[i] : 0);
      u32x4_t v = ld16_nt(xg + ((size_t)idx * a.C + slab * CE) * sizeof(T));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
      hreg[i] = v;
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if ((hvalid >> i) & 1u) st16(s_halo + buf * HBUF + st_off[i], hreg[i]);
  };

  // ---- fragment addresses.  B: halo row (wp * 8 + p + r), column li + s; A: the lane's 16 bytes of kout row
  // wk * 64 + (li >> 2) * 16 + t * 4 + (li & 3) (so that a lane ends up with 16 consecutive kouts), channels (kk * 4 + g) * 8 of the slab
  int Bb[3][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
#pragma unroll
    for (int s = 0; s < 3; ++s) Bb[s][kk] = ((wp * 8) * PITCH + li + s) * 128 + ((ci ^ ((li + s) & 7)) << 4);
  }
  unsigned avoff[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) avoff[t] = (unsigned)(((wk * 64 + (li >> 2) * 16 + t * 4 + (li & 3)) * 9 * a.C + g * 8) * (int)sizeof(T));
  const char* wg = reinterpret_cast<const char*>(a.w);

  // acc: the running item (the MFMAs' set); park: the finished item, copied out of acc at the item's end and stored piece by piece
  // between the next item's MFMAs (one code path; the copy is 128 moves per item)
#ifdef R4_NOPARK
  f32x4_t acc[TK][TP];
#define park acc
#else
  f32x4_t acc[TK][TP], park[TK][TP];
#endif
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t A[3][TK], B[2][TP];

  auto lda = [&](int buf, int k0, int slab, int step) {      // step = tap * 2 + kk
    const char* wb = wg + ((size_t)(k0 * 9 + (step >> 1)) * a.C + slab * CE + (step & 1) * 32) * sizeof(T);
#pragma unroll
    for (int t = 0; t < TK; ++t) A[buf][t] = ld16(wb + avoff[t]);
  };
  auto ldb = [&](int buf, int hbuf, int step) {
    const int tap = step >> 1, kk = step & 1;
    const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
    for (int p = 0; p < TP; ++p) B[buf][p] = ld16(s_halo + hbuf * HBUF + Bb[s][kk] + (p + r) * (PITCH * 128));
  };

  // ---- output stage of accumulator set z (item geometry q), in TP pieces (one pixel row each) so that it can be issued between MFMAs
  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = reinterpret_cast<const char*>(a.residual);
  u32x4_t rres[2][2];                          // residual of the piece two steps ahead (MODE 1)
  auto out_off = [&](const Geo& q, int p) {
    const size_t pix = ((size_t)q.n0 * a.H + q.h0 + wp * 8 + p) * a.W + q.w0 + li;
    return (pix * a.K + q.k0 + wk * 64 + g * 16) * sizeof(T);
  };
  auto res_load = [&](const Geo& q, int p) {
    if (MODE == 1 && rg) {
      const size_t off = out_off(q, p);
      rres[p & 1][0] = ld16(rg + off);
      rres[p & 1][1] = ld16(rg + off + 16);
    }
  };
  auto out_piece = [&](const Geo& q, int p) {
    const int kb = q.k0 + wk * 64 + g * 16;
    const size_t off = out_off(q, p);
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      float vq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = qq * 8 + e;
        vq[e] = park[idx >> 2][p][idx & 3];
      }
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vq[e] += s_bias[kb + qq * 8 + e];
        if (rg) {
          float rr[8];
          Elem<T>::unpack(rres[p & 1][qq], rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) vq[e] += rr[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) vq[e] = clamp_lo(vq[e], out_lo);
      }
      st16(yg + off + qq * 16, PackH<T>::run(vq));
    }
  };

  // ---- pipeline fill
  Geo cur = geom(first);
  load_halo(cur, 0);
  store_halo(0);
  lda(0, cur.k0, 0, 0);
  lda(1, cur.k0, 0, 1);
  __syncthreads();
  ldb(0, 0, 0);

  int item = first, slab = 0, hb = 0;
  bool have_prev = false;                            // park holds a finished item (prv) whose output stage is still to be issued
  Geo prv = cur;
  for (;;) {
    const bool last = slab + 1 == nslabs;
    const bool done = last && item + G >= n_items;
    const int nslab = last ? 0 : slab + 1;
    const Geo nxt = (last && !done) ? geom(item + G) : cur;
#ifdef R4_NOPARK
    const bool flush = false;
#else
    const bool flush = have_prev && slab == 0;       // the previous item's output stage rides in this stage
#endif
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      // weights two steps ahead (steps 16, 17 request the next stage's first two), pixels one step ahead
      if (i < 16) lda((i + 2) % 3, cur.k0, slab, i + 2);
      else lda((i + 2) % 3, nxt.k0, nslab, i - 16);
      if (i < 17) ldb((i + 1) & 1, hb, i + 1);
      else ldb(0, hb ^ 1, 0);
      if (flush && i >= 1 && i < 1 + TP) res_load(prv, i - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p)
          acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A[i % 3][t]), __builtin_bit_cast(bf16x8_t, B[i & 1][p]),
                                                               acc[t][p], 0, 0, 0);
      // the previous item's output stage, one pixel row per step (steps 3..10; its residual was requested two steps earlier)
      if (flush && i >= 3 && i < 3 + TP) out_piece(prv, i - 3);
      __builtin_amdgcn_sched_barrier(0);
      if (i == 1) {
        R4_BAR_FREE();                               // every wave has left the previous stage: its buffer may be overwritten
        load_halo(nxt, nslab);
      }
      if (i == 12) store_halo(hb ^ 1);
      if (i == 15) R4_BAR_PUBLISH();                 // the next halo is complete before step 17 reads its first fragments
      __builtin_amdgcn_sched_barrier(0);
    }
    hb ^= 1;
    if (last) {
#ifdef R4_NOPARK
#pragma unroll
      for (int p = 0; p < TP; ++p) { res_load(cur, p); out_piece(cur, p); }
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) { park[t][p] = acc[t][p]; acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#endif
      have_prev = true;
      prv = cur;
      if (done) break;
      item += G;
      cur = nxt;
    }
    slab = nslab;
  }
  // the last item's output stage
#ifndef R4_NOPARK
#pragma unroll
  for (int p = 0; p < TP; ++p) { res_load(prv, p); out_piece(prv, p); }
#endif
}

bool conv_r4_ok(int dtype, const ConvArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_R4"); return e && atoi(e) != 0; }();      // opt-in (A/B runs)
  if (!on || dtype != DT_BF16) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.transposed || a.par4 || a.tap_mask || a.pix_mul > 1) return false;
  if (a.in_scale || a.mask_x || a.accumulate || a.stats || a.osh != 1 || a.seg_images > 0) return false;
  if (a.H % 16 != 0 || a.W % 16 != 0 || a.PH != a.H || a.PW != a.W || a.OH != a.H || a.OW != a.W) return false;
  if (a.C % 64 != 0 || a.K % 128 != 0) return false;
  return true;
}

hipError_t launch_conv_r4(const ConvArgs& a, hipStream_t st) {
  const bool m1 = a.bias || a.residual || a.relu;
  const size_t lds = 2 * 18 * 24 * 128 + (m1 ? a.K * sizeof(float) : 0);
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_r4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_r4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = a.N * (a.H / 16) * (a.W / 16);
  const int kbn = a.K / 128;
  const int n_items = tiles * kbn;
  const int cus = device_cus();
  const int grid = n_items < cus ? n_items : cus;
  const int kshift = (kbn > 1 && (kbn & (kbn - 1)) == 0 && (grid & (kbn - 1)) == 0) ? __builtin_ctz(kbn) : -1;
  if (m1) hipLaunchKernelGGL(conv3x3_r4_kernel<1>, dim3(grid), dim3(256), lds, st, a, tiles, n_items, kshift);
  else hipLaunchKernelGGL(conv3x3_r4_kernel<0>, dim3(grid), dim3(256), lds, st, a, tiles, n_items, kshift);
  return hipGetLastError();
}

}  // namespace sslcr

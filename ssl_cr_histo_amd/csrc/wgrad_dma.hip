// Weight gradient of the 3x3 / stride 1 / pad 1 convs, LDS-halo form with the operands staged by LDS DMA (bf16, K % 128 == 0).
//
//   dW[k][r][s][c] += sum_{pixels} dY[pix][k] * act(X)[pix + (r-1, s-1)][c]        (autograd of final_loss.backward(),
//                                                                                    eval_BreastPathQ_SSL_CR.py:98-100)
//
// Same tiling, register tile, MFMA order, slabs and fold as wgrad3x3_halo_kernel<bf16, TW, 2> (wgrad_halo.hip) -- the result is
// the same bits -- but in the dominant conv's form:
//   * the dY tile (128 pixels x 128 kouts) never touches registers: `buffer_load_dwordx4 ... lds`, four 1 KB pieces per wave and
//     tile, the lane picking its SOURCE chunk so that the linear DMA placement is the swizzled tile the transpose reads expect;
//   * where the input needs no transform (XF = false: conv1 of a block reads the previous block's output) the halo goes the same
//     way, padding by the buffer range check; where the producer's BatchNorm + ReLU is applied on the way (XF = true: conv2 reads
//     raw1) the halo stays register-staged as in the halo kernel;
//   * behind an LDS DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 INTRINSIC, so the transpose
//     reads are inline asm and the LDS counter is kept by hand: the two reads of tap t + 1 are issued before the four MFMAs of
//     tap t, which wait with lgkmcnt(2).
// The DMA of tile t + 1 goes into the other buffer while tile t computes: a request costs its wave some hundred cycles of issue, so a
// wave's nine requests are spread over the odd taps of the first half of the MFMA loop (all nine in front of the loop: +0.14 ms per
// step); `s_waitcnt vmcnt(0)` + a bare barrier end the tile.
#include <stdlib.h>

#include <atomic>

#include "kernels.hpp"

namespace sslcr {

namespace {
template <int OFF>
__device__ __forceinline__ u32x2_t wd_tr(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t wd_lds(const void* p) { return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)(const char*)p; }
__device__ __forceinline__ bf16x8_t wd_frag(const u32x2_t& lo, const u32x2_t& hi) {
  return __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
}
__device__ __forceinline__ void wd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int WD_RB = 128;                         // LDS bytes per pixel row (64 bf16 channels)
__device__ __forceinline__ int wd_swz(int row) { return (row >> 1) & 3; }
// logical 16-byte chunk that lives in physical slot j of row `row` (the XOR swizzle of wgrad_halo.hip is its own inverse)
__device__ __forceinline__ int wd_chunk_of_slot(int row, int j) { return (((j >> 1) ^ wd_swz(row)) << 1) | (j & 1); }

// B fragment of (depth step Q, tap T): two transpose reads at a compile-time offset from the lane's per-column base
template <int TW, int Q, int T>
__device__ __forceinline__ void wd_issue_b(u32x2_t (&b)[2], const uint32_t (&bbase)[3]) {
  constexpr int PITCH = TW == 16 ? 24 : 16, HH = 10;
  constexpr int HPIX = TW == 16 ? Q * 2 * PITCH : (Q >> 1) * (HH * PITCH) + ((4 * Q) & 7) * PITCH;
  constexpr int OFF = (HPIX + (T / 3) * PITCH) * WD_RB, HI = (TW == 16 ? 8 : PITCH) * WD_RB;
  b[0] = wd_tr<OFF>(bbase[T % 3]);
  b[1] = wd_tr<OFF + HI>(bbase[T % 3]);
}
template <int Q>
__device__ __forceinline__ void wd_issue_a(u32x2_t (&alo)[4], u32x2_t (&ahi)[4], const uint32_t (&abase)[4]) {
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    alo[t4] = wd_tr<Q * 32 * WD_RB>(abase[t4]);
    ahi[t4] = wd_tr<Q * 32 * WD_RB + 8 * WD_RB>(abase[t4]);
  }
}
template <int N>
__device__ __forceinline__ void wd_wait2(u32x2_t& r0, u32x2_t& r1) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wd_wait10(u32x2_t (&alo)[4], u32x2_t (&ahi)[4], u32x2_t& r0, u32x2_t& r1) {
  asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]),
               "+v"(ahi[3]), "+v"(r0), "+v"(r1) : "n"(N));
}

// The next tile's DMA requests, one piece at a time: a request costs its wave some hundred cycles of issue (NOTES r05), so the nine
// pieces of a wave are spread over the odd taps of the first half of the MFMA loop instead of standing in front of it.
template <int TW, bool XF>
struct WdNext {
  static constexpr int PITCH = TW == 16 ? 24 : 16, NI = 128 / (8 * TW);
  static constexpr int YH = 128 * WD_RB, YBUF = 2 * YH, NHPIECE = NI * 10 * PITCH / 8, HPW = (NHPIECE + 7) / 8;
  LdsDma dy, x;
  int vy[4], vh[HPW];
  int soff, wave8;
  char* yb;
  bool more;
  template <int K>
  __device__ __forceinline__ void piece() const {
    if (!more) return;
    if constexpr (K < 4) {
      dy.load16(yb + (K >> 1) * YH + (wave8 + 8 * (K & 1)) * 1024, vy[K], soff);
    } else if constexpr (!XF && K - 4 < HPW) {
      if (wave8 + 8 * (K - 4) < NHPIECE) x.load16(yb + YBUF + (wave8 + 8 * (K - 4)) * 1024, vh[K - 4], 0);
    }
  }
};

// One tap (TAU = 9 q + t of the tile's 36) of the MFMA loop.  The LDS round trip under the load of eight waves is ~350 cycles (r05
// ablation: with a quarter of the MFMAs the loop still ran at 75 % of its time) against 64 cycles of MFMAs per tap and wave, so the
// B fragment is requested D taps ahead (ring of D + 1 register pairs) and, ADBL, the A fragments of the next depth step at tap 4
// of this one.  Issue order per tap: B(TAU + D), then A(q + 1) at t = 4; the wait counts the reads YOUNGER than B(TAU).
template <int TW, int D, bool ADBL, int TAU, typename NX>
__device__ __forceinline__ void wd_tap(f32x4_t (&acc)[9][4], u32x2_t (&alo)[2][4], u32x2_t (&ahi)[2][4], u32x2_t (&bf)[D + 1][2],
                                       const uint32_t (&abase)[4], const uint32_t (&bbase)[3], const NX& nx, bool spread) {
  constexpr int Q = TAU / 9, T = TAU % 9;
  if constexpr ((TAU & 1) && TAU < 18) {
    if (spread) nx.template piece<(TAU >> 1)>();
  }
  constexpr int AI = ADBL ? (Q & 1) : 0;
  if constexpr (!ADBL && T == 0) wd_issue_a<Q>(alo[0], ahi[0], abase);
  if constexpr (TAU + D < 36) wd_issue_b<TW, (TAU + D) / 9, (TAU + D) % 9>(bf[(TAU + D) % (D + 1)], bbase);
  if constexpr (ADBL && T == 4 && Q < 3) wd_issue_a<Q + 1>(alo[(Q + 1) & 1], ahi[(Q + 1) & 1], abase);
  constexpr int AHEAD = 35 - TAU < D ? 35 - TAU : D;
  constexpr int YOUNGER_B = 2 * AHEAD + ((ADBL && Q < 3 && T >= 4 && T <= 4 + D) ? 8 : 0);
  // (!ADBL, t = 0: the A reads sit between B(TAU + D - 1) and B(TAU + D): only the latter may stay outstanding)
  constexpr int CNT = (!ADBL && T == 0) ? (TAU + D < 36 ? 2 : 0) : YOUNGER_B;
  static_assert(CNT <= 15, "lgkmcnt is a 4-bit field");
  if constexpr (T == 0) wd_wait10<CNT>(alo[AI], ahi[AI], bf[TAU % (D + 1)][0], bf[TAU % (D + 1)][1]);
  else wd_wait2<CNT>(bf[TAU % (D + 1)][0], bf[TAU % (D + 1)][1]);
  __builtin_amdgcn_sched_barrier(0);
  const bf16x8_t b = wd_frag(bf[TAU % (D + 1)][0], bf[TAU % (D + 1)][1]);
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
    acc[T][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd_frag(alo[AI][t4], ahi[AI][t4]), b, acc[T][t4], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}
template <int TW, int D, bool ADBL, int TAU0, int N, typename NX>
__device__ __forceinline__ void wd_taps(f32x4_t (&acc)[9][4], u32x2_t (&alo)[2][4], u32x2_t (&ahi)[2][4], u32x2_t (&bf)[D + 1][2],
                                        const uint32_t (&abase)[4], const uint32_t (&bbase)[3], const NX& nx, bool spread) {
  if constexpr (N > 0) {
    wd_tap<TW, D, ADBL, TAU0>(acc, alo, ahi, bf, abase, bbase, nx, spread);
    wd_taps<TW, D, ADBL, TAU0 + 1, N - 1>(acc, alo, ahi, bf, abase, bbase, nx, spread);
  }
}
template <int TW, int D, int I>
__device__ __forceinline__ void wd_prologue_b(u32x2_t (&bf)[D + 1][2], const uint32_t (&bbase)[3]) {
  if constexpr (I < D) {
    wd_issue_b<TW, I / 9, I % 9>(bf[I % (D + 1)], bbase);
    wd_prologue_b<TW, D, I + 1>(bf, bbase);
  }
}
}  // namespace

template <int TW, bool XF>
__global__ __launch_bounds__(512, 2) void wgrad3x3_dma_kernel(const WgradArgs a, int tiles_per_split, int ntiles, f32x4_t* partials, int stagger) {
  using T = bf16_t;
  constexpr int EPC = 8, RB = WD_RB, CPR = 8;
  constexpr int TH = 8, NI = 128 / (TH * TW), HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;
  constexpr int NT = 512;
  constexpr int HL = (HP * CPR + NT - 1) / NT, HROWS = NT / CPR;
  constexpr int PITCH = TW == 16 ? 24 : 16;
  constexpr int YH = 128 * RB, YBUF = 2 * YH, HROWS_LDS = NI * HH * PITCH, HBUF = HROWS_LDS * RB, BUF = YBUF + HBUF;
  constexpr int NHPIECE = HROWS_LDS / 8, HPW = (NHPIECE + 7) / 8;     // halo pieces of 8 rows: 30 / 40; per wave 4 / 5
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  const int wave = wave8 & 3, kh = tid >> 8;              // cin tile, kout half
  const int li = lane & 15, g = lane >> 4;
  const int gx = a.K / 128, gy = a.C / 64, GT = gx * gy, splits = (int)gridDim.x / GT;
  int bz, bt;
  if ((splits & 7) == 0) {                                // the gx * gy workgroups of one pixel split: neighbours on one XCD
    const int w = blockIdx.x, grp = w / (8 * GT), r = w - grp * 8 * GT;
    bz = grp * 8 + (r & 7); bt = r >> 3;
  } else {
    bz = (int)blockIdx.x / GT; bt = (int)blockIdx.x - bz * GT;
  }
  const int by = bt / gx, bx = bt - by * gx;
  const int k0 = bx * 128, c0 = by * 64;
  const int chunk = tid % CPR, prow = tid / CPR;          // register-staged halo role (XF)
  float* s_aff = reinterpret_cast<float*>(smem + 2 * BUF);
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  if constexpr (XF) {
    for (int i = tid; i < nseg * 64; i += NT) {
      const int sg = i >> 6, ch = i & 63;
      s_aff[sg * 128 + ch] = a.in_scale[(size_t)sg * a.seg_stride + c0 + ch];
      s_aff[sg * 128 + 64 + ch] = a.in_shift[(size_t)sg * a.seg_stride + c0 + ch];
    }
  }
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  const int t_begin = bz * tiles_per_split;
  int t_end = t_begin + tiles_per_split;
  if (t_end > ntiles) t_end = ntiles;
  if (t_begin >= t_end) {
    if (partials) {
      const size_t wg = ((size_t)bz * gy + by) * gx + bx;
      for (int e = 0; e < 36; ++e) partials[(wg * 36 + e) * NT + tid] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }

  LdsDma dy_dma, x_dma;
  dy_dma.init(a.dy, (unsigned)((size_t)a.N * a.H * a.W * a.K * 2));
  x_dma.init(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.C * 2));
  // ---- DMA roles.  dY: pieces wave8 + 8 i (i < 4): kout half i >> 1, pixel rows 8 * (wave8 + 8 (i & 1)) .. + 7
  int voff_y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = 8 * (wave8 + 8 * (i & 1)) + (lane >> 3);
    int rel;
    if (TW == 16) rel = (p >> 4) * a.W + (p & 15);
    else rel = ((p >> 6) * a.H + ((p >> 3) & 7)) * a.W + (p & 7);
    voff_y[i] = (rel * a.K + 64 * (i >> 1)) * 2 + wd_chunk_of_slot(p, lane & 7) * 16;
  }
  // halo (XF = false): pieces wave8 + 8 i of 8 pitched rows; per lane the row's place in the halo and its byte offset from the
  // halo's (0, 0) pixel
  int voff_h[XF ? 1 : HPW], hrc[XF ? 1 : HPW];
  if constexpr (!XF) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      const int hpl = 8 * (wave8 + 8 * i) + (lane >> 3);
      const int ni = hpl / (HH * PITCH), rem = hpl - ni * (HH * PITCH);
      const int hr = rem / PITCH, hc = rem - hr * PITCH;
      hrc[i] = (hc < HWD && hpl < HROWS_LDS) ? ((hr << 8) | hc) : ((0x3fff << 8) | 0xff);     // pitch padding: fails the range tests
      voff_h[i] = ((ni * a.H + hr) * a.W + hc) * a.C * 2 + wd_chunk_of_slot(hpl, lane & 7) * 16;
    }
  }
  auto tile_origin = [&](int tile, int& h0, int& w0, int& n0) {
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    n0 = (t / tiles_h) * NI;
    h0 = th_i * TH; w0 = tw_i * TW;
    return (n0 * a.H + h0) * a.W + w0;
  };
  // the requests of `tile` into buffer `buf`: per-tile offsets now, the nine pieces all at once (now = true) or by the MFMA loop
  WdNext<TW, XF> nx;
  nx.dy = dy_dma; nx.x = x_dma; nx.wave8 = wave8; nx.more = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) nx.vy[i] = voff_y[i];
  auto prepare_dma = [&](int tile, int buf, bool now) {
    int h0, w0, n0;
    const int origin = tile_origin(tile, h0, w0, n0);
    nx.yb = smem + buf * BUF;
    nx.soff = (origin * a.K + k0) * 2;
    nx.more = true;
    if constexpr (!XF) {
      const int sbase = ((origin - a.W - 1) * a.C + c0) * 2;        // byte offset of the halo's (0, 0) pixel (negative at the first tile)
#pragma unroll
      for (int i = 0; i < HPW; ++i) {
        const int hr = hrc[i] >> 8, hc = hrc[i] & 255;
        const bool ok = (unsigned)(h0 + hr - 1) < (unsigned)a.H && (unsigned)(w0 + hc - 1) < (unsigned)a.W;
        nx.vh[i] = ok ? voff_h[i] + sbase : (int)0xfffffff0;
      }
    }
    if (now) {
      nx.template piece<0>(); nx.template piece<1>(); nx.template piece<2>(); nx.template piece<3>(); nx.template piece<4>();
      nx.template piece<5>(); nx.template piece<6>(); nx.template piece<7>(); nx.template piece<8>();
    }
  };

  // ---- register-staged halo with the producer's BatchNorm + ReLU (XF), as wgrad3x3_halo_kernel
  int rel_h[XF ? HL : 1];
  unsigned long long edge = 0;
  unsigned hvalid = 0;
  if constexpr (XF) {
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int hp = prow + HROWS * i;
      rel_h[i] = 0;
      if (hp < HP) {
        const int ni = hp / (HH * HWD), rem = hp - ni * (HH * HWD);
        const int hr = rem / HWD, hc = rem - hr * HWD;
        rel_h[i] = (ni * a.H + hr - 1) * a.W + hc - 1;
        hvalid |= 1u << i;
        edge |= (unsigned long long)((hr == 0) | ((hr == HH - 1) << 1) | ((hc == 0) << 2) | ((hc == HWD - 1) << 3)) << (4 * i);
      }
    }
  }
  const char* xg = reinterpret_cast<const char*>(a.x);
  const size_t xbase = ((size_t)c0 + chunk * EPC) * sizeof(T);
  u32x4_t hreg[XF ? HL : 1];
  unsigned hin = 0;
  int seg_ld = 0;
  auto load_regs = [&](int tile) {
    int h0, w0, n0;
    const int origin = tile_origin(tile, h0, w0, n0);
    seg_ld = a.seg_images > 0 ? n0 / a.seg_images : 0;
    const unsigned long long out =
        (unsigned long long)((h0 == 0) | ((h0 + TH >= a.H) << 1) | ((w0 == 0) << 2) | ((w0 + TW >= a.W) << 3)) * 0x1111111111111111ull;
    hin = hvalid;
    const unsigned long long bad = edge & out;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      // (measured and not kept here, same box: the branch-free form of wgrad_halo.hip -- 190-193 -> 197-203 us per launch --, and the
      //  halo conversion of waves 0-3 moved in front of their MFMA loop so that SIMD partners convert at opposite ends of a tile: +-0)
      u32x4_t v = {0u, 0u, 0u, 0u};
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      if (ok) v = ld16(xg + (size_t)(origin + rel_h[i]) * a.C * sizeof(T) + xbase);
      else hin &= ~(1u << i);
      hreg[i] = v;
    }
  };
  auto store_halo = [&](int buf) {
    char* hb = smem + buf * BUF + YBUF;
    float r_scale[EPC], r_shift[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { r_scale[e] = s_aff[seg_ld * 128 + chunk * EPC + e]; r_shift[e] = s_aff[seg_ld * 128 + 64 + chunk * EPC + e]; }
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int hp = prow + HROWS * i;
      if (hp >= HP) continue;
      const int hni = hp / (HH * HWD), hrem = hp - hni * (HH * HWD);
      const int hpl = (hni * HH + hrem / HWD) * PITCH + hrem % HWD;
      u32x4_t v = hreg[i];
      if ((hin >> i) & 1u) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float q = fmaf(f[e], r_scale[e], r_shift[e]);
          f[e] = a.in_relu ? fmaxf(q, 0.f) : q;
        }
        v = Elem<T>::pack(f);
      }
      st16(hb + hpl * RB + ((((chunk >> 1) ^ wd_swz(hpl)) << 5) | ((chunk & 1) << 4)), v);
    }
  };

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto hpix = [&](int p) {
    if (TW == 16) return (p >> 4) * PITCH + (p & 15);
    return (p >> 6) * (HH * PITCH) + ((p >> 3) & 7) * PITCH + (p & 7);
  };
  // this lane's source pixels within a 32-pixel depth step: pl and pl + 8 (wgrad_halo.hip: the 8 pixel rows of a 32-lane group are
  // consecutive, the 32-byte column group is XORed with bits 1..2 of the row)
  const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
  uint32_t aoff[4], boff[3];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) aoff[t4] = wd_lds(smem) + kh * YH + pl * RB + ((t4 ^ wd_swz(pl)) << 5) + (li & 3) * 8;
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) {
    const int hp = hpix(pl) + sx;
    boff[sx] = wd_lds(smem) + YBUF + hp * RB + ((wave ^ wd_swz(hp)) << 5) + (li & 3) * 8;
  }
  constexpr int D = 1;                         // B fragments one tap ahead (two or three taps, and the next depth step's A fragments
  constexpr bool ADBL = false;                 // at tap 4, measured the same: NOTES r05)
  const bool spread = stagger != 0;
  prepare_dma(t_begin, 0, true);
  if constexpr (XF) {
    load_regs(t_begin);
    __syncthreads();                            // s_aff
    store_halo(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wd_barrier();
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    nx.more = more;
    if (more) {
      prepare_dma(tile + 1, buf ^ 1, !spread);
      if constexpr (XF) load_regs(tile + 1);
    }
    uint32_t abase[4], bbase[3];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) abase[t4] = aoff[t4] + buf * BUF;
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) bbase[sx] = boff[sx] + buf * BUF;
    u32x2_t bf[D + 1][2], alo[2][4], ahi[2][4];
    if constexpr (ADBL) wd_issue_a<0>(alo[0], ahi[0], abase);
    wd_prologue_b<TW, D, 0>(bf, bbase);
    wd_taps<TW, D, ADBL, 0, 36>(acc, alo, ahi, bf, abase, bbase, nx, spread);
    if constexpr (XF) {
      if (more) store_halo(buf ^ 1);            // the compiler's wait for hreg also retires the (older) DMA requests
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wd_barrier();
    buf ^= 1;
  }

  const size_t wg = ((size_t)bz * gy + by) * gx + bx;
  f32x4_t* sp = partials + wg * 36 * NT + tid;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) sp[(size_t)(t * 4 + t4) * NT] = acc[t][t4];
}

// bf16, 128-kout blocks, tensors below 2 GiB (32-bit DMA offsets), more than one pixel split (the kernel always writes slabs);
// SSLCR_WG_DMA=0: wgrad3x3_halo_kernel (A/B runs)
bool wgrad_dma_ok(int dtype, const WgradArgs& a, int splits) {
  static const bool on = [] { const char* e = getenv("SSLCR_WG_DMA"); return !e || atoi(e) != 0; }();
  if (!on || dtype != DT_BF16 || a.K % 128 != 0 || splits < 2) return false;
  const long long px = (long long)a.N * a.H * a.W;
  return px * a.K * 2 < (1ll << 31) && px * a.C * 2 < (1ll << 31);
}

template <int TW, bool XF>
static hipError_t launch_wd_t(const WgradArgs& a, int tps, int ntiles, int splits, f32x4_t* slabs, hipStream_t st) {
  constexpr int NI = 128 / (8 * TW), PITCH = TW == 16 ? 24 : 16;
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const size_t lds = (size_t)2 * (256 + NI * 10 * PITCH) * 128 + 512 * nseg;
  auto kern = wgrad3x3_dma_kernel<TW, XF>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int gx = a.K / 128, gy = a.C / 64;
  static const int stagger = [] { const char* e = getenv("SSLCR_WG_STAGGER"); return e ? atoi(e) : 1; }();
  hipLaunchKernelGGL(kern, dim3(gx * gy * splits), dim3(512), lds, st, a, tps, ntiles, slabs, stagger);
  return hipGetLastError();
}

hipError_t launch_wgrad_dma(const WgradArgs& a, int tw, int tps, int ntiles, int splits, void* slabs, hipStream_t st) {
  f32x4_t* s = reinterpret_cast<f32x4_t*>(slabs);
  const bool xf = a.in_scale != nullptr;
  if (tw == 16) return xf ? launch_wd_t<16, true>(a, tps, ntiles, splits, s, st) : launch_wd_t<16, false>(a, tps, ntiles, splits, s, st);
  return xf ? launch_wd_t<8, true>(a, tps, ntiles, splits, s, st) : launch_wd_t<8, false>(a, tps, ntiles, splits, s, st);
}

}  // namespace sslcr

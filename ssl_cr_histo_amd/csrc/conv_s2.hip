// 3x3 / stride 2 / pad 1 NHWC convolution (ResNet18 layer{2,3}.0.conv1 via models/net.py:32,77) on 16x16 OUTPUT tiles, with the block's
// 1x1 / stride 2 projection (downsample.0) riding in the same launch on a second accumulator set: the persistent, all-DMA form of
// the stride-1 pipeline (conv_h16.hip) for the shapes the gather kernel (conv_dma.hip) served at 0.3-0.55 PF/s.
//
// A stride-2 3x3 conv is four stride-1 sub-convolutions over the input's (row, column) PARITY PLANES: output (i, j) reads input
// (2i + r - 1, 2j + s - 1), so r = 1 reads the even-row plane at i, r = 0 / 2 the odd-row plane at i / i + 1 (same for s):
//   plane OO (odd rows, odd columns; 17 x 17 per tile)  taps (0,0) (0,2) (2,0) (2,2)
//   plane OE (odd rows, even columns; 17 x 16)           taps (0,1) (2,1)
//   plane EO (even rows, odd columns; 16 x 17)           taps (1,0) (1,2)
//   plane EE (even rows, even columns; 16 x 16)          tap  (1,1) -- and the whole 1x1 / 2 projection
// 9 C MACs per output and kout, no zero work, and within a plane a fragment's 16 pixels are 16 CONSECUTIVE LDS rows.  The planes
// are not a layout in HBM: the input stays NHWC, and the LDS DMA (buffer_load_dwordx4 ... lds) GATHERS a plane -- every lane
// gives its own global address, the LDS placement is the instruction's linear 1 KiB -- so no producer or consumer changes.
//
// What bounds the design is LDS: a 64-channel slab of the 33 x 33 input region is 139 KB, next to 16 KB of weights per tap.  So
//   * a STAGE is one plane of one slab (37 KB, 2-4 taps); three plane buffers roll: stage S computes on buffer S % 3 while stage
//     S + 1 has landed and stage S + 2 is in flight -- no stage-end halo swap, no staging registers, two stages (>= 4 taps) of lead;
//   * weights go through a ring of three single-tap slots (16 KB each), one tap of lead;
//   * ONE bare s_barrier per tap, in the middle of it (between its two 32-channel k-steps): it publishes the next tap's weights
//     (and, at a stage's last tap, the next stage's plane) and frees the previous tap's slot (at a stage's first tap, the previous
//     stage's buffer) -- whose last reads fed the MFMAs of the step before the barrier;
//   * a DMA instruction costs its wave 100-500 cycles of issue (measured: the same walk without any DMA runs 1617 cycles per tap;
//     with four waves gathering planes and four streaming weights 2620-2860, with every wave issuing its share of both 2290), so all
//     eight waves issue, two weight pieces and one to three plane pieces per tap each, and the waves of a SIMD pair issue their plane
//     pieces in different phases -- waves 0-3 in front of their 32 MFMAs, waves 4-7 behind theirs;
//   * s_waitcnt vmcnt retires IN ORDER: a wave's requests are ordered (weights of tap t + 2, then plane pieces) and every barrier's
//     wait is counted -- the plane pieces of the previous tap may stay outstanding, so a weight tap has one tap of lead and a plane
//     piece two; an item's output stores are issued behind the requests they could delay and counted too (vmcnt(n + 16)).
// 8 waves, wave (wp, wk) = 4 output rows x 64 kouts (TK = TP = 4), two accumulator sets (conv1, projection) = 128 registers,
// fragments double-buffered across the barriers.  Output stage: pack + store (+ BatchNorm partial sums by the row16_fold16 tree)
// for the train forward, bias (+ ReLU) for the eval forward with the BatchNorm folded (EVAL instance).  bf16 only; the fp32 parity
// mode and the shapes that are not 16x16-tileable stay on conv_dma.
#include "kernels.hpp"

namespace sslcr {

#define S2_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S2_BARRIER() asm volatile("s_barrier" ::: "memory")
#ifdef SSLCR_S2_PROF
__device__ unsigned long long g_s2_prof[8][8];
#define S2_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define S2_ACC(i, d) s2_t[i] += (d)
#else
#define S2_T(v)
#define S2_ACC(i, d)
#endif

template <bool PAIR, bool EVAL>
__global__ __launch_bounds__(512, 2) void conv_s2_kernel(const ConvArgs a, const ConvArgs d, const int tiles_total, const int n_items, const int kshift) {
  typedef bf16_t T;
  constexpr int BKO = 128, TK = 4, TP = 4, CE = 64;
  constexpr int HB = 289 * 128, WS = BKO * 128, NHB = 3, NWS = 3;
  constexpr int NTAP = PAIR ? 10 : 9;
  constexpr int NST = PAIR ? 16 : 8;          // output stores of a wave per item (statistics rows not counted: the wait only gets longer)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_halo = smem;
  char* s_w = smem + NHB * HB;
  // EVAL: [2 (bias, out_scale)][2 (conv1, projection)][128] of the kout block bk0 (reloaded in the output stage of an item whose block differs)
  float* s_bias = reinterpret_cast<float*>(s_w + NWS * WS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 3, wk = wave >> 2;
  const bool late = wave >= 4;                // waves 4-7 issue their plane pieces BEHIND their MFMAs, waves 0-3 in front
  const int OH = a.PH, OW = a.PW;
  const int tiles_w = OW / 16, tiles_h = OH / 16;
  const int G = gridDim.x;
  const int lb = blockIdx.x;
  const int first = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);        // XCD-contiguous runs of tiles, as conv3x3_h16
  if (first >= n_items) return;
  const int nslabs = a.C / CE;
  const float out_lo = a.relu ? 0.f : -__builtin_inff();
  const float out_lo_d = (PAIR && d.relu) ? 0.f : -__builtin_inff();

  // bias (the folded BatchNorm's shift) and sslcr_conv_desc.out_scale (1 where the scale sits in the filters) of one 128-kout block
  auto load_affine = [&](int k0) {
    if (tid < 128) {
      s_bias[tid] = a.bias ? a.bias[k0 + tid] : 0.f;
      s_bias[128 + tid] = (PAIR && d.bias) ? d.bias[k0 + tid] : 0.f;
      s_bias[256 + tid] = a.out_scale ? a.out_scale[k0 + tid] : 1.f;
      s_bias[384 + tid] = (PAIR && d.out_scale) ? d.out_scale[k0 + tid] : 1.f;
    }
  };
  const bool osc_on = EVAL && (a.out_scale != nullptr || (PAIR && d.out_scale != nullptr));

  const bool kfast = kshift >= 0;
  struct Geo { int k0, tile, n0, oh0, ow0; };
  auto geom = [&](int item) {
    Geo q;
    const int kbi = kfast ? item & ((1 << kshift) - 1) : item / tiles_total;
    q.tile = kfast ? item >> kshift : item - kbi * tiles_total;
    q.k0 = kbi * BKO;
    int t = q.tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    q.n0 = t / tiles_h;
    q.oh0 = th_i * 16; q.ow0 = tw_i * 16;
    return q;
  };

  // ---- plane gather.  A plane is `rows` pixels = ceil(rows / 8) PIECES of one DMA instruction (8 pixels x 128 B); piece grp fills
  // LDS rows [grp * 8, +8) of the plane buffer; lane -> (row = plane pixel, 16-byte slot); the lane loads the channel chunk
  // (slot ^ (plane column & 7)) so that the linear placement is the swizzled tile.  A pixel in the zero padding (top row / left
  // column of the image) gets an offset beyond the resource.  A call issues pieces base + 8 i + wave, i < HP, of the plane.
  const char* xg = reinterpret_cast<const char*>(a.x);
  const size_t img_bytes = (size_t)a.H * a.W * a.C * sizeof(T);
  constexpr int OOR = (int)0xfffffff0u;
  auto issue_halo = [&](const Geo& q, int slab, int plane, int buf, int nw, int w, int i0, int ni) {
    // pieces i0 .. i0 + ni - 1 of lane group w of nw: piece i is plane group i * nw + w
    const int rows = plane == 0 ? 289 : (plane == 3 ? 256 : 272);
    const bool p17 = (plane & 1) == 0;                                           // planes 0 (OO) and 2 (EO): 17 columns
    const int rodd = plane < 2 ? 1 : 0, codd = p17 ? 1 : 0;
    const int ngrp = (rows + 7) >> 3;
    // the tile's share of the address goes into the resource base (it is negative by a row and a pixel where the tile touches the
    // image's top / left edge: those pixels are padding and get the out-of-range offset)
    const long tile_off = ((long)(2 * q.oh0 - rodd) * a.W + (2 * q.ow0 - codd)) * a.C * (long)sizeof(T);
    LdsDma xd;
    xd.init(xg + (size_t)q.n0 * img_bytes + tile_off, 0x7fffffffu);
    const bool top = rodd && q.oh0 == 0, left = codd && q.ow0 == 0;
    const int soff = slab * 128;
    const int rstep = 2 * a.W * a.C * (int)sizeof(T), cstep = 2 * a.C * (int)sizeof(T);
    char* dst0 = s_halo + buf * HB;
    // (the lane id goes through an opaque move: everything below depends on lane and the piece only, and left visible the compiler
    //  hoists the (row, column, swizzle) triples of all pieces out of the walk and keeps them in registers)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int grp = i * nw + w;
      if (i >= i0 && i < i0 + ni && grp < ngrp) {
        const int qq = grp * 8 + (ln >> 3);
        int ridx, cidx;
        if (p17) { ridx = (qq * 241) >> 12; cidx = qq - 17 * ridx; }
        else { ridx = qq >> 4; cidx = qq & 15; }
        const bool pad = (top && ridx == 0) || (left && cidx == 0);
        const int voff = pad ? OOR : ridx * rstep + cidx * cstep + (((ln & 7) ^ (cidx & 7)) << 4);
        if (qq < rows) xd.load16(dst0 + grp * 1024, voff, soff);
      }
    }
  };
  // ---- weight taps: 16 pieces per tap (8 LDS rows each, in fragment order -- wperm<4>); lane group w of nw issues pieces
  // w, w + nw, ...: one per-lane source offset for piece w, the later pieces' share is uniform
  auto wperm_inv4 = [](int rr) { const int blk = rr >> 6, x = rr & 63; return blk * 64 + ((x >> 2) & 3) * 16 + (x >> 4) * 4 + (x & 3); };
  auto wsrc_of = [&](int w, int rowbytes) {
    const int rr = w * 8 + (lane >> 3);
    return (wperm_inv4(rr) * rowbytes + (((lane & 7) ^ (rr & 7)) << 4));
  };
  const int wsrc0 = wsrc_of(wave, 9 * a.C * (int)sizeof(T)), wsrc0d = wsrc_of(wave, a.C * (int)sizeof(T));     // pieces wave, wave + 8
  LdsDma wdma, wdmad;
  wdma.init(a.w, 0x7fffffffu);
  wdmad.init(PAIR ? d.w : a.w, 0x7fffffffu);
  auto issue_w = [&](int k0, int slab, int tap, int slot, int nw, int w, int src, int srcd) {
    // tap order of a slab: OO (0,0) (0,2) (2,0) (2,2) | OE (0,1) (2,1) | EO (1,0) (1,2) | EE (1,1) [projection]
    const bool ds = PAIR && tap == 9;
    const int tid9 = (int)((0x453718620ull >> (4 * tap)) & 15ull);
    const int soff = ds ? (k0 * a.C + slab * CE) * (int)sizeof(T) : ((k0 * 9 + tid9) * a.C + slab * CE) * (int)sizeof(T);
    const int kstep = (ds ? a.C : 9 * a.C) * (int)sizeof(T);
    char* dst0 = s_w + slot * WS + w * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i * nw < 16) {
        // piece i * nw + w: LDS rows (i * nw + w) * 8; the kout rows of piece i * nw lie wperm_inv4(i * nw * 8) above piece 0's
        if (ds) wdmad.load16(dst0 + i * nw * 1024, srcd, soff + wperm_inv4(i * nw * 8) * kstep);
        else wdma.load16(dst0 + i * nw * 1024, src, soff + wperm_inv4(i * nw * 8) * kstep);
      }
    }
  };
  // ---- request cursors: hc = the plane stage being requested (two stages ahead of the MFMAs), wc = the last weight tap requested
  struct HCur { int item, slab, plane; bool valid; Geo q; } hc;
  struct WCur { int item, slab, tap; bool valid; int k0; } wc;
  auto hc_next = [&]() {
    if (++hc.plane == 4) {
      hc.plane = 0;
      if (++hc.slab == nslabs) {
        hc.slab = 0;
        hc.item += G;
        hc.valid = hc.item < n_items;
        if (hc.valid) hc.q = geom(hc.item);
      }
    }
  };
  auto wc_next = [&]() {
    if (++wc.tap == NTAP) {
      wc.tap = 0;
      if (++wc.slab == nslabs) {
        wc.slab = 0;
        wc.item += G;
        wc.valid = wc.item < n_items;
        if (wc.valid) wc.k0 = geom(wc.item).k0;
      }
    }
  };

  // ---- fragment addresses (CURRENT: slot / buffer offset included; a rotation adds the uniform difference).  Weights: the wave's
  // 64 rows of the slot, 16 consecutive rows per MFMA tile.  Pixels: plane row (wp * 4 + p + dr), column li + dc; the swizzle key is
  // the plane column & 7, so dc picks one of two registers and the rows are immediates.  The registers hold the pitch-17 form; a
  // pitch-16 plane subtracts wp * 4 rows' worth of the 128-byte difference.
  int Ac[2], Bc[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
    Ac[kk] = (wk * 64 + li) * 128 + ((ci ^ (li & 7)) << 4);
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) Bc[dc][kk] = (wp * 4 * 17 + li + dc) * 128 + ((ci ^ ((li + dc) & 7)) << 4);
  }
  int a_off = 0, b_off = 0;

  f32x4_t acc[PAIR ? 2 : 1][TK][TP];
#pragma unroll
  for (int s = 0; s < (PAIR ? 2 : 1); ++s)
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int p = 0; p < TP; ++p) acc[s][t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t A[2][TK], B[2][TP];

  // ---- pipeline fill: planes of stages 0 and 1 (OO, OE of the first item's slab 0), weight taps 0 and 1
  Geo cur = geom(first);
  int bk0 = cur.k0;
  if (EVAL) load_affine(bk0);                  // (in front of the DMA requests: the barrier below publishes it)
  hc.item = first; hc.slab = 0; hc.plane = 1; hc.valid = true; hc.q = cur;
  wc.item = first; wc.slab = 0; wc.tap = 1; wc.valid = true; wc.k0 = cur.k0;
  issue_w(cur.k0, 0, 0, 0, 8, wave, wsrc0, wsrc0d);
  issue_w(cur.k0, 0, 1, 1, 8, wave, wsrc0, wsrc0d);
  issue_halo(cur, 0, 0, 0, 8, wave, 0, 5);
  issue_halo(cur, 0, 1, 1, 8, wave, 0, 5);
  S2_VMCNT(0);
  __syncthreads();

  int hb = 0, ws = 0;                          // plane buffer of the current stage, weight slot of the current tap
  bool prev_ok = false;                        // the previous tap's site issued its plane pieces (uniform): its count may stay outstanding
  bool after_epi = false;                      // the output stores of the previous item are the youngest requests of this wave
  auto set_a = [&](int slot) {
    const int o = slot * WS, dlt = o - a_off;
    a_off = o;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) Ac[kk] += dlt;
  };
  auto set_b = [&](int buf, bool p17) {
    const int o = buf * HB - (p17 ? 0 : wp * 4 * 128), dlt = o - b_off;
    b_off = o;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int dc = 0; dc < 2; ++dc) Bc[dc][kk] += dlt;
  };
  // fragments of one k-step into register set `buf`
#define S2_FRAGS(buf, kk, DR, DC, P17)                                                                      \
  do {                                                                                                      \
    A[buf][0] = ld16(s_w + Ac[kk]);                                                                         \
    _Pragma("unroll") for (int p = 0; p < TP; ++p) B[buf][p] = ld16(s_halo + Bc[DC][kk] + (p + (DR)) * ((P17) ? 17 * 128 : 16 * 128)); \
    _Pragma("unroll") for (int t = 1; t < TK; ++t) A[buf][t] = ld16(s_w + Ac[kk] + t * 2048);             \
  } while (0)
#define S2_MFMA(buf, SET)                                                                                   \
  do {                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < TK; ++t)                                                              \
      _Pragma("unroll") for (int p = 0; p < TP; ++p)                                                        \
        acc[SET][t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A[buf][t]),   \
                                                                 __builtin_bit_cast(bf16x8_t, B[buf][p]), acc[SET][t][p], 0, 0, 0); \
    _Pragma("unroll") for (int q = 0; q < TK + TP; ++q) {                                                   \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, (TK * TP) / (TK + TP), 0);                                \
    }                                                                                                       \
  } while (0)
#define S2_WAIT(n)                                                                                          \
  do {                                                                                                      \
    if ((n) == 0) S2_VMCNT(0); else if ((n) == 1) S2_VMCNT(1); else if ((n) == 2) S2_VMCNT(2); else if ((n) == 3) S2_VMCNT(3);      \
    else if ((n) == 4) S2_VMCNT(4); else if ((n) == 5) S2_VMCNT(5); else if ((n) == 7) S2_VMCNT(7); else if ((n) == 8) S2_VMCNT(8); else if ((n) == 9) S2_VMCNT(9); else if ((n) == 10) S2_VMCNT(10); \
    else if ((n) == 11) S2_VMCNT(11); else if ((n) == 12) S2_VMCNT(12); else if ((n) == 16) S2_VMCNT(16); else if ((n) == 17) S2_VMCNT(17); \
    else if ((n) == 18) S2_VMCNT(18); else if ((n) == 19) S2_VMCNT(19); else if ((n) == 20) S2_VMCNT(20); else S2_VMCNT(0);       \
  } while (0)
  // One tap = two k-steps with the barrier BETWEEN them (at the tap boundary it measured 4-10 % slower: the second k-step's MFMAs are
  // what runs while the partner wave issues).  Behind the barrier every wave requests its two pieces of the weight tap two ahead,
  // then pieces I0 .. I0 + HP - 1 (piece i = plane group 8 i + wave) of the plane two stages ahead (TPL) -- waves 0-3 at once, waves
  // 4-7 behind the second k-step's MFMAs, so that on every SIMD one wave feeds the matrix pipe while the other sits in its DMA issue.
  // A wave's requests retire in order: the wait lets the NPREV plane pieces every wave issued behind the previous barrier stay
  // outstanding -- the next tap's weights and every plane piece requested two barriers ago or earlier have landed.  So a plane's
  // pieces go out at least two barriers before the one that needs them (in the last tap of the stage before).
  // FIRST / LAST: position in the stage (FIRST moves the plane cursor on, LAST rotates the plane buffer); N*: the next tap's fragments.
#define S2_TAP(DR, DC, P17, SET, FIRST, LAST, NPREV, TPL, I0, HP, NDR, NDC, NP17, ITEM_FIRST)               \
  do {                                                                                                      \
    S2_FRAGS(1, 1, DR, DC, P17);                                                                            \
    S2_MFMA(0, SET);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    {                                                                                                       \
      S2_T(tb0);                                                                                            \
      if (!prev_ok) S2_VMCNT(0);                                                                            \
      else if ((ITEM_FIRST) && after_epi) S2_WAIT((NPREV) + NST);                                           \
      else S2_WAIT(NPREV);                                                                                  \
      S2_T(tb1);                                                                                            \
      S2_BARRIER();                                                                                         \
      S2_T(tb2);                                                                                            \
      S2_ACC(0, tb1 - tb0); S2_ACC(1, tb2 - tb1);                                                           \
    }                                                                                                       \
    if (ITEM_FIRST) after_epi = false;                                                                      \
    wc_next();                                                                                              \
    if (wc.valid) issue_w(wc.k0, wc.slab, wc.tap, ws == 0 ? 2 : ws - 1, 8, wave, wsrc0, wsrc0d);            \
    if (FIRST) hc_next();                                                                                   \
    prev_ok = hc.valid;                                                                                     \
    const int hb_req = hb == 0 ? 2 : hb - 1;                                                                \
    if (!late && hc.valid && (HP) > 0) issue_halo(hc.q, hc.slab, TPL, hb_req, 8, wave, I0, HP);             \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    ws = ws == 2 ? 0 : ws + 1;                                                                              \
    set_a(ws);                                                                                              \
    if (LAST) { hb = hb == 2 ? 0 : hb + 1; set_b(hb, NP17); }                                               \
    S2_FRAGS(0, 0, NDR, NDC, NP17);                                                                         \
    S2_MFMA(1, SET);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (late && hc.valid && (HP) > 0) issue_halo(hc.q, hc.slab, TPL, hb_req, 8, wave, I0, HP);              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)

  char* yg = reinterpret_cast<char*>(a.y);
  char* ydg = reinterpret_cast<char*>(PAIR ? d.y : a.y);
  int item = first, slab = 0;
#ifdef SSLCR_S2_PROF
  unsigned long long s2_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long s2_begin = __builtin_readcyclecounter();
#endif
  S2_FRAGS(0, 0, 0, 0, true);
  // Plane pieces per tap (a plane is 37 / 34 / 34 / 32 pieces = at most 5 per wave).  NPREV of a tap = the number of i in the
  // PREVIOUS tap's range with 8 i + 7 < pieces (what every wave issued there).
  constexpr int NP0 = PAIR ? 1 : 4;            // ... of a slab's last tap, seen by the next slab's first
  for (;;) {
    //     DR DC P17    SET FIRST  LAST   NPREV TPL I0 HP  next: DR DC P17
    S2_TAP(0, 0, true, 0, true, false, NP0, 2, 0, 2, 0, 1, true, true);         // OO (0,0)   EO pieces 0-1
    S2_TAP(0, 1, true, 0, false, false, 2, 2, 2, 1, 1, 0, true, false);         // OO (0,2)             2
    S2_TAP(1, 0, true, 0, false, false, 1, 2, 3, 1, 1, 1, true, false);         // OO (2,0)             3
    S2_TAP(1, 1, true, 0, false, true, 1, 2, 4, 1, 0, 0, false, false);         // OO (2,2)             4 (groups 32-33)
    S2_TAP(0, 0, false, 0, true, false, 0, 3, 0, 2, 1, 0, false, false);        // OE (0,1)   EE pieces 0-1
    S2_TAP(1, 0, false, 0, false, true, 2, 3, 2, 2, 0, 0, true, false);         // OE (2,1)             2-3
    if constexpr (PAIR) {
      S2_TAP(0, 0, true, 0, true, false, 2, 0, 0, 3, 0, 1, true, false);        // EO (1,0)   next OO pieces 0-2
      S2_TAP(0, 1, true, 0, false, true, 3, 0, 3, 2, 0, 0, false, false);       // EO (1,2)             3-4
      S2_TAP(0, 0, false, 0, true, false, 1, 1, 0, 3, 0, 0, false, false);      // EE (1,1)   next OE pieces 0-2
      S2_TAP(0, 0, false, 1, false, true, 3, 1, 3, 2, 0, 0, true, false);       // EE projection        3-4
    } else {
      S2_TAP(0, 0, true, 0, true, false, 2, 0, 0, 5, 0, 1, true, false);        // EO (1,0)   next OO pieces 0-4 (the EE stage is one tap)
      S2_TAP(0, 1, true, 0, false, true, 4, 0, 0, 0, 0, 0, false, false);       // EO (1,2)
      S2_TAP(0, 0, false, 0, true, true, 0, 1, 0, 5, 0, 0, true, false);        // EE (1,1)   next OE pieces 0-4
    }
    if (++slab < nslabs) continue;
    slab = 0;
    // ---------------- output stage of the finished item (the next item's first fragments are in flight)
    S2_T(te0);
    if (EVAL && cur.k0 != bk0) {               // uniform; a kout-block-major walk crosses a block boundary a few times per launch
      __syncthreads();                         // (drains this wave's DMA once: rare)
      bk0 = cur.k0;
      load_affine(bk0);
      __syncthreads();
      prev_ok = false;                         // the next barrier waits vmcnt(0): the counted waits assumed the requests above were the only ones
    }
    {
      const int kb = cur.k0 + wk * 64 + g * 16;
      const int kl = wk * 64 + g * 16;
#pragma unroll
      for (int s = 0; s < (PAIR ? 2 : 1); ++s) {
        char* yo = s ? ydg : yg;
        float* st = s ? d.stats : a.stats;
        const float lo = s ? out_lo_d : out_lo;
        float bias[16];
        if (EVAL) {
          if (osc_on) {                        // eval-mode BatchNorm scale kept out of the filters: acc * scale in place
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const float sj = s_bias[256 + s * 128 + kl + j];
#pragma unroll
              for (int p = 0; p < TP; ++p) acc[s][j >> 2][p][j & 3] *= sj;
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) bias[j] = s_bias[s * 128 + kl + j];
        }
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const size_t pix = ((size_t)cur.n0 * OH + cur.oh0 + wp * 4 + p) * OW + cur.ow0 + li;
          const size_t off = (pix * a.K + kb) * sizeof(T);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float vq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int idx = q * 8 + e;
              const float v = acc[s][idx >> 2][p][idx & 3];
              vq[e] = EVAL ? clamp_lo(v + bias[idx], lo) : v;
            }
            st16(yo + off + q * 16, PackH<T>::run(vq));
          }
        }
        if (!EVAL && st) {
          float s1[16], s2[16];
#pragma unroll
          for (int t = 0; t < TK; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float x1 = 0.f, x2 = 0.f;
#pragma unroll
              for (int p = 0; p < TP; ++p) { const float v = acc[s][t][p][j]; x1 += v; x2 = fmaf(v, v, x2); }
              s1[t * 4 + j] = x1; s2[t * 4 + j] = x2;
            }
          row16_fold16(s1);
          row16_fold16(s2);
          if ((li & 3) == 0) {
            float* sp = st + ((size_t)(cur.tile * 4 + wp) * 2) * a.K + kb + (li >> 2) * 4;
            *reinterpret_cast<f32x4_t*>(sp) = f32x4_t{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4_t*>(sp + a.K) = f32x4_t{s2[0], s2[1], s2[2], s2[3]};
          }
        }
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
          for (int p = 0; p < TP; ++p) acc[s][t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef SSLCR_S2_PROF
    { S2_T(te1); S2_ACC(2, te1 - te0); S2_ACC(3, 1); }
#endif
    after_epi = true;
    item += G;
    if (item >= n_items) break;
    cur = geom(item);
  }
#ifdef SSLCR_S2_PROF
  if (blockIdx.x == 0 && lane == 0) {
    s2_t[4] = __builtin_readcyclecounter() - s2_begin;
    for (int i = 0; i < 8; ++i) g_s2_prof[wave][i] = s2_t[i];
  }
#endif
#undef S2_TAP
#undef S2_WAIT
#undef S2_MFMA
#undef S2_FRAGS
}

// ---- host side
// the 3x3 / 2 descriptor this kernel serves (bf16; fp32 and everything else stay on the gather kernel)
bool conv_s2_ok(int dtype, const ConvArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_S2"); return !e || atoi(e) != 0; }();     // 0: the gather kernel keeps these shapes (A/B runs)
  if (!on || dtype != DT_BF16) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 2 || a.pad != 1 || a.transposed || a.par4 || a.tap_mask || a.pix_mul > 1 || a.pix_off_h || a.pix_off_w) return false;
  if (a.in_scale || a.residual || a.accumulate || a.mask_x || a.osh != 1) return false;
  if (a.H % 32 != 0 || a.W % 32 != 0 || a.PH != a.H / 2 || a.PW != a.W / 2 || a.OH != a.PH || a.OW != a.PW) return false;
  if (a.C % 64 != 0 || a.K % 128 != 0) return false;
  if ((size_t)a.H * a.W * a.C * 2 >= 0x7fffffffull) return false;
  if (a.bias && a.stats) return false;
  if (a.relu && !a.bias) return false;          // the output clamp lives in the EVAL instance (chosen by the bias): conv_dma takes relu without one
  if (a.out_scale && !a.bias) return false;
  if (a.seg_images > 0 && a.N % a.seg_images != 0) return false;
  // the statistics rows are asked for without a dtype (sslcr_conv2d_partial_rows) and the fp32 mode runs these shapes on the gather
  // kernel: served only where that kernel's row count is this one's (M / 64: its 128-pixel tiles, M >= 2048)
  if ((long)a.N * a.PH * a.PW < 2048) return false;
  return true;
}
// ... and the 1x1 / 2 projection that may ride with it: same input, same output shape, same mode
bool conv_s2_pair_ok(int dtype, const ConvArgs& a, const ConvArgs& d) {
  if (!conv_s2_ok(dtype, a)) return false;
  if (d.R != 1 || d.S != 1 || d.stride != 2 || d.pad != 0 || d.transposed || d.par4 || d.tap_mask || d.pix_mul > 1) return false;
  if (d.in_scale || d.residual || d.accumulate || d.mask_x || d.osh != 1) return false;
  if (d.x != a.x || d.N != a.N || d.H != a.H || d.W != a.W || d.C != a.C || d.K != a.K || d.PH != a.PH || d.PW != a.PW) return false;
  if ((a.bias != nullptr) != (d.bias != nullptr) || (a.stats != nullptr) != (d.stats != nullptr)) return false;
  if (d.relu && !d.bias) return false;
  if (d.out_scale && !d.bias) return false;
  return true;
}
int conv_s2_rows(const ConvArgs& a) { return a.N * (a.PH / 16) * (a.PW / 16) * 4; }

template <bool PAIR, bool EVAL>
static hipError_t launch_s2(const ConvArgs& a, const ConvArgs& d, hipStream_t st) {
  const size_t lds = 3 * 289 * 128 + 3 * 128 * 128 + (EVAL ? 4 * 128 * sizeof(float) : 0);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  auto kern = conv_s2_kernel<PAIR, EVAL>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = a.N * (a.PH / 16) * (a.PW / 16);
  const int kbn = a.K / 128;
  const int n_items = tiles * kbn;
  const int cus = device_cus();
  const int grid = n_items < cus ? n_items : cus;
  const int kshift = (kbn > 1 && (kbn & (kbn - 1)) == 0 && (grid & (kbn - 1)) == 0) ? __builtin_ctz(kbn) : -1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, d, tiles, n_items, kshift);
  return hipGetLastError();
}

// d == nullptr: the 3x3 / 2 conv alone
hipError_t launch_conv_s2(const ConvArgs& a, const ConvArgs* d, hipStream_t st) {
  const bool eval = a.bias != nullptr;
  if (d) return eval ? launch_s2<true, true>(a, *d, st) : launch_s2<true, false>(a, *d, st);
  return eval ? launch_s2<false, true>(a, a, st) : launch_s2<false, false>(a, a, st);
}

const char* conv_s2_name(const ConvArgs& a, bool pair) {
  const bool eval = a.bias != nullptr;
  if (pair) return eval ? "sslcr::conv_s2_kernel<true, true>" : "sslcr::conv_s2_kernel<true, false>";
  return eval ? "sslcr::conv_s2_kernel<false, true>" : "sslcr::conv_s2_kernel<false, false>";
}

}  // namespace sslcr

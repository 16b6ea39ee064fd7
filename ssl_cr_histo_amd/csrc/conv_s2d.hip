// Input gradient of the 3x3 / stride 2 / pad 1 convolution (autograd of layer{2,3}.0.conv1, reached from final_loss.backward(),
// eval_BreastPathQ_SSL_CR.py:98-100): the four output-parity classes of dX in ONE pass over dY, on the persistent all-DMA pipeline
// of conv_s2.hip.  (The gather kernel ran the classes as four sub-launches re-gathering dY per tap: 0.36 PF/s on the layer2.0 shape.)
//
//   dX[2i + a][2j + b][c] = sum over the taps (r, s) with r = a + 1 (mod 2), s = b + 1 (mod 2) of  W[k][r][s][c] . dY[i + dr][j + dc][k],
//   dr = (r == 0), dc = (s == 0):   class (0,0): tap (1,1)      class (0,1): taps (1,0) (1,2)
//                                   class (1,0): taps (0,1) (2,1)   class (1,1): taps (0,0) (0,2) (2,0) (2,2)
// -- a stride-1 2x2-window convolution of dY with 4 C virtual output channels, every tap used once.  A workgroup takes a 16x16 block
// of (i, j) and 64 channels c: ONE 17x17 halo of dY per 64-channel slab of k (37 KB, zero beyond the bottom / right edge), nine
// 8 KB weight taps per slab, four accumulator sets (one per class, 256 px x 64 c: TK = 2) = 128 registers, output = a 32x32 patch of
// dX written as 128-byte channel runs.
//
// Pipeline: as conv_s2.hip, at the granularity of a GROUP of three taps (48 MFMAs per wave): a ring of three group slots
// (3 x 24 KB), two halo buffers; one bare barrier per group, in its middle -- it publishes the next group's weights (and, in a slab's
// last group, the next slab's halo) and frees the previous group's slot (in a slab's first group, the previous slab's buffer).
// Behind the barrier every wave requests its three pieces of the group two ahead and, in a slab's first group, its (up to) five
// pieces of the next slab's halo -- waves 0-3 at once, waves 4-7 behind their MFMAs.  Requests retire in order and every wait is
// counted: the halo pieces of the previous barrier may stay outstanding.  The taps are ordered by their dY offset, and the B
// fragments are kept per k-step, so a tap with its predecessor's offset reads no pixels at all (10-14 fragment reads per 24 MFMAs).
#include "kernels.hpp"

namespace sslcr {

#define S2D_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S2D_BARRIER() asm volatile("s_barrier" ::: "memory")

__global__ __launch_bounds__(512, 2) void conv_s2d_kernel(const ConvArgs a, const int tiles_total, const int n_items, const int kshift) {
  typedef bf16_t T;
  constexpr int BC = 64, TK = 2, TP = 4, CE = 64;
  constexpr int HB = 289 * 128, WT = BC * 128, WG = 3 * WT, NHB = 2, NWS = 3;
  constexpr int NST = 16;                     // output stores of a wave per item
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_halo = smem;
  char* s_w = smem + NHB * HB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 3, wk = wave >> 2;
  const bool late = wave >= 4;
  // descriptor (sslcr_conv_desc, par4 form): x = dY [N][H][W][C] (C = the conv's output channels: the contraction), w = [K][3][3][C]
  // (K = the conv's input channels), y = dX [N][2H][2W][K]
  const int IH = a.H, IW = a.W;
  const int tiles_w = IW / 16, tiles_h = IH / 16;
  const int G = gridDim.x;
  const int lb = blockIdx.x;
  const int first = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);
  if (first >= n_items) return;
  const int nslabs = a.C / CE;

  const bool kfast = kshift >= 0;
  struct Geo { int c0, n0, i0, j0; };
  auto geom = [&](int item) {
    Geo q;
    const int cbi = kfast ? item & ((1 << kshift) - 1) : item / tiles_total;
    int t = kfast ? item >> kshift : item - cbi * tiles_total;
    q.c0 = cbi * BC;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    q.n0 = t / tiles_h;
    q.i0 = th_i * 16; q.j0 = tw_i * 16;
    return q;
  };

  // ---- dY halo: 289 pixels = 37 pieces of 8 pixels x 128 B; piece grp fills LDS rows [grp * 8, +8); the lane loads channel chunk
  // (slot ^ (halo column & 7)).  Rows / columns beyond the map (bottom / right edge tiles) are zero: out-of-range offset.
  const char* xg = reinterpret_cast<const char*>(a.x);
  const size_t img_bytes = (size_t)IH * IW * a.C * sizeof(T);
  constexpr int OOR = (int)0xfffffff0u;
  auto issue_halo = [&](const Geo& q, int slab, int buf) {
    LdsDma xd;
    xd.init(xg + (size_t)q.n0 * img_bytes + ((size_t)q.i0 * IW + q.j0) * a.C * sizeof(T), 0x7fffffffu);
    const int rmax = IH - q.i0, cmax = IW - q.j0;      // halo rows / columns >= these are padding (16 on a bottom / right edge tile)
    const int soff = slab * 128;
    const int rstep = IW * a.C * (int)sizeof(T), cstep = a.C * (int)sizeof(T);
    char* dst0 = s_halo + buf * HB;
    int ln = lane;
    asm volatile("" : "+v"(ln));                       // (opaque: keeps the per-piece address arithmetic from being hoisted into registers)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int grp = i * 8 + wave;
      if (grp < 37) {
        const int qq = grp * 8 + (ln >> 3);
        const int ridx = (qq * 241) >> 12, cidx = qq - 17 * ridx;
        const int voff = (ridx >= rmax || cidx >= cmax) ? OOR : ridx * rstep + cidx * cstep + (((ln & 7) ^ (cidx & 7)) << 4);
        if (qq < 289) xd.load16(dst0 + grp * 1024, voff, soff);
      }
    }
  };
  // ---- weights: a tap is 64 channel rows x 128 B = 8 pieces; wave w issues piece w of each of the group's three taps.  LDS row
  // t * 16 + q * 4 + j of a 32-row half holds channel q * 8 + t * 4 + j (wperm<2>: a lane ends up with 8 consecutive channels).
  int wsrc0;
  {
    const int rr = wave * 8 + (lane >> 3);
    const int blk = rr >> 5, x = rr & 31;
    const int crow = blk * 32 + ((x >> 2) & 3) * 8 + (x >> 4) * 4 + (x & 3);
    wsrc0 = (crow * 9 * a.C + (((lane & 7) ^ (rr & 7)) << 3)) * (int)sizeof(T);
  }
  LdsDma wdma;
  wdma.init(a.w, 0x7fffffffu);
  // tap order of a slab (by dY offset): G0 (1,1) (1,2) (2,1) | G1 (2,2) (1,0) (2,0) | G2 (0,1) (0,2) (0,0)
  auto issue_w = [&](int c0, int slab, int grp, int slot) {
    char* dst0 = s_w + slot * WG + wave * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = (int)((0x021638754ull >> (4 * (grp * 3 + i))) & 15ull);
      wdma.load16(dst0 + i * WT, wsrc0, ((c0 * 9 + tap) * a.C + slab * CE) * (int)sizeof(T));
    }
  };

  struct HCur { int item, slab; bool valid; Geo q; } hc;       // the last halo requested
  struct WCur { int item, slab, grp; bool valid; int c0; } wc;   // the last weight group requested
  auto hc_next = [&]() {
    if (++hc.slab == nslabs) {
      hc.slab = 0;
      hc.item += G;
      hc.valid = hc.item < n_items;
      if (hc.valid) hc.q = geom(hc.item);
    }
  };
  auto wc_next = [&]() {
    if (++wc.grp == 3) {
      wc.grp = 0;
      if (++wc.slab == nslabs) {
        wc.slab = 0;
        wc.item += G;
        wc.valid = wc.item < n_items;
        if (wc.valid) wc.c0 = geom(wc.item).c0;
      }
    }
  };

  // fragment addresses (current slot / buffer offset included)
  int Ac[2], Bc[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
    Ac[kk] = (wk * 32 + li) * 128 + ((ci ^ (li & 7)) << 4);
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) Bc[dc][kk] = (wp * 4 * 17 + li + dc) * 128 + ((ci ^ ((li + dc) & 7)) << 4);
  }

  f32x4_t acc[4][TK][TP];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int p = 0; p < TP; ++p) acc[s][t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t A[2][TK], B[2][TP];                  // A: by step parity; B: by k-step (kept across taps with one dY offset)

  Geo cur = geom(first);
  hc.item = first; hc.slab = 0; hc.valid = true; hc.q = cur;
  wc.item = first; wc.slab = 0; wc.grp = 1; wc.valid = true; wc.c0 = cur.c0;
  issue_w(cur.c0, 0, 0, 0);
  issue_w(cur.c0, 0, 1, 1);
  issue_halo(cur, 0, 0);
  S2D_VMCNT(0);
  __syncthreads();

  int hb = 0, ws = 0;
  bool prev_halo = false, after_epi = false;
  auto rot_a = [&]() {
    const int dlt = ws == 2 ? -2 * WG : WG;
    ws = ws == 2 ? 0 : ws + 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) Ac[kk] += dlt;
  };
  auto rot_b = [&]() {
    const int dlt = hb ? -HB : HB;
    hb ^= 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int dc = 0; dc < 2; ++dc) Bc[dc][kk] += dlt;
  };
#define S2D_FA(buf, kk, TI)                                                                                 \
  do {                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < TK; ++t) A[buf][t] = ld16(s_w + Ac[kk] + (TI) * WT + t * 2048);   \
  } while (0)
#define S2D_FB(kk, DR, DC)                                                                                  \
  do {                                                                                                      \
    _Pragma("unroll") for (int p = 0; p < TP; ++p) B[kk][p] = ld16(s_halo + Bc[DC][kk] + (p + (DR)) * (17 * 128)); \
  } while (0)
#define S2D_MFMA(abuf, kk, CLS)                                                                             \
  do {                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < TK; ++t)                                                          \
      _Pragma("unroll") for (int p = 0; p < TP; ++p)                                                        \
        acc[CLS][t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A[abuf][t]),  \
                                                                 __builtin_bit_cast(bf16x8_t, B[kk][p]), acc[CLS][t][p], 0, 0, 0); \
  } while (0)
  // one k-step: request the next step's fragments (A always; B only where the next step's offset differs from what B[nkk] holds),
  // then this step's 8 MFMAs
#define S2D_STEP(abuf, kk, CLS, NTI, NKK, NLOADB, NDR, NDC, ROT)                                            \
  do {                                                                                                      \
    if (ROT) rot_a();                                                                                       \
    if ((ROT) == 2) rot_b();                                                                                \
    S2D_FA((abuf) ^ 1, NKK, NTI);                                                                           \
    if (NLOADB) S2D_FB(NKK, NDR, NDC);                                                                      \
    S2D_MFMA(abuf, kk, CLS);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
  // the barrier in the middle of a group: NPREV = halo pieces every wave issued behind the previous barrier (4 behind a slab's first)
#define S2D_SYNC(NPREV_HALO, SLAB_FIRST, ITEM_FIRST)                                                        \
  do {                                                                                                      \
    if ((NPREV_HALO) && prev_halo) { if ((ITEM_FIRST) && after_epi) S2D_VMCNT(20); else S2D_VMCNT(4); }     \
    else if ((ITEM_FIRST) && after_epi) S2D_VMCNT(16);                                                      \
    else S2D_VMCNT(0);                                                                                      \
    S2D_BARRIER();                                                                                          \
    if (ITEM_FIRST) after_epi = false;                                                                      \
    wc_next();                                                                                              \
    if (wc.valid) issue_w(wc.c0, wc.slab, wc.grp, ws == 0 ? 2 : ws - 1);                                    \
    if (SLAB_FIRST) { hc_next(); prev_halo = hc.valid; }                                                    \
    if ((SLAB_FIRST) && !late && hc.valid) issue_halo(hc.q, hc.slab, hb ^ 1);                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)

  char* yg = reinterpret_cast<char*>(a.y);
  int item = first, slab = 0;
  S2D_FA(0, 0, 0);
  S2D_FB(0, 0, 0);
  for (;;) {
    // B[kk] holds offset (dr, dc) of the running tap; classes: 0 = (0,0), 1 = (0,1), 2 = (1,0), 3 = (1,1)
    //        abuf kk CLS | next: tap-in-group, kk, load B?, dr, dc | rotate (1: weight slot, 2: + halo buffer)
    // ---- G0: (1,1)->c0, (1,2)->c1, (2,1)->c2, all at offset (0,0)
    S2D_STEP(0, 0, 0, 0, 1, true, 0, 0, 0);      // (1,1) k0   next (1,1) k1: B[1] <- (0,0)
    S2D_STEP(1, 1, 0, 1, 0, false, 0, 0, 0);     // (1,1) k1   next (1,2) k0: same offset
    S2D_STEP(0, 0, 1, 1, 1, false, 0, 0, 0);     // (1,2) k0
    S2D_SYNC(false, true, true);
    S2D_STEP(1, 1, 1, 2, 0, false, 0, 0, 0);     // (1,2) k1
    S2D_STEP(0, 0, 2, 2, 1, false, 0, 0, 0);     // (2,1) k0
    S2D_STEP(1, 1, 2, 0, 0, false, 0, 0, 1);     // (2,1) k1   next G1 (2,2) k0: offset (0,0) still
    if (late && hc.valid) issue_halo(hc.q, hc.slab, hb ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- G1: (2,2)->c3 at (0,0); (1,0)->c1, (2,0)->c3 at (0,1)
    S2D_STEP(0, 0, 3, 0, 1, false, 0, 0, 0);     // (2,2) k0
    S2D_STEP(1, 1, 3, 1, 0, true, 0, 1, 0);      // (2,2) k1   next (1,0) k0: B[0] <- (0,1)
    S2D_STEP(0, 0, 1, 1, 1, true, 0, 1, 0);      // (1,0) k0   next (1,0) k1: B[1] <- (0,1)
    S2D_SYNC(true, false, false);
    S2D_STEP(1, 1, 1, 2, 0, false, 0, 1, 0);     // (1,0) k1
    S2D_STEP(0, 0, 3, 2, 1, false, 0, 1, 0);     // (2,0) k0
    S2D_STEP(1, 1, 3, 0, 0, true, 1, 0, 1);      // (2,0) k1   next G2 (0,1) k0: B[0] <- (1,0)
    // ---- G2: (0,1)->c2, (0,2)->c3 at (1,0); (0,0)->c3 at (1,1)
    S2D_STEP(0, 0, 2, 0, 1, true, 1, 0, 0);      // (0,1) k0   next (0,1) k1: B[1] <- (1,0)
    S2D_STEP(1, 1, 2, 1, 0, false, 1, 0, 0);     // (0,1) k1
    S2D_STEP(0, 0, 3, 1, 1, false, 1, 0, 0);     // (0,2) k0
    S2D_SYNC(false, false, false);
    S2D_STEP(1, 1, 3, 2, 0, true, 1, 1, 0);      // (0,2) k1   next (0,0) k0: B[0] <- (1,1)
    S2D_STEP(0, 0, 3, 2, 1, true, 1, 1, 0);      // (0,0) k0   next (0,0) k1: B[1] <- (1,1)
    S2D_STEP(1, 1, 3, 0, 0, true, 0, 0, 2);      // (0,0) k1   next slab / item G0 (1,1) k0: B[0] <- (0,0) of the other buffer
    if (++slab < nslabs) continue;
    slab = 0;
    // ---------------- output stage: class (a, b) of pixel (i, j) goes to dX (2i + a, 2j + b); a lane holds 8 consecutive channels
    {
      const int OW2 = 2 * IW;
      const int cb = cur.c0 + wk * 32 + g * 8;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const size_t pix = ((size_t)cur.n0 * (2 * IH) + 2 * (cur.i0 + wp * 4 + p) + (cls >> 1)) * OW2 + 2 * (cur.j0 + li) + (cls & 1);
          float vq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) vq[e] = acc[cls][e >> 2][p][e & 3];
          st16(yg + (pix * a.K + cb) * sizeof(T), PackH<T>::run(vq));
        }
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
          for (int p = 0; p < TP; ++p) acc[cls][t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    after_epi = true;
    item += G;
    if (item >= n_items) break;
    cur = geom(item);
  }
#undef S2D_SYNC
#undef S2D_STEP
#undef S2D_MFMA
#undef S2D_FB
#undef S2D_FA
}

// the par4 descriptor this kernel serves (sslcr_conv_desc: transposed, stride 2, pix_mul 2, PH x PW = dY's map)
bool conv_s2d_ok(int dtype, const ConvArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_S2D"); return !e || atoi(e) != 0; }();    // 0: the gather kernel keeps the shape (A/B runs)
  if (!on || dtype != DT_BF16) return false;
  if (!a.par4 || !a.transposed || a.stride != 2 || a.R != 3 || a.S != 3 || a.pad != 1 || a.pix_mul != 2 || a.tap_mask) return false;
  if (a.in_scale || a.residual || a.accumulate || a.mask_x || a.bias || a.out_scale || a.relu || a.stats || a.osh != 1 || a.seg_images > 0) return false;
  if (a.H % 16 != 0 || a.W % 16 != 0 || a.PH != a.H || a.PW != a.W || a.OH != 2 * a.H || a.OW != 2 * a.W) return false;
  if (a.C % 64 != 0 || a.K % 64 != 0) return false;
  if ((size_t)a.H * a.W * a.C * 2 >= 0x7fffffffull) return false;
  return true;
}

hipError_t launch_conv_s2d(const ConvArgs& a, hipStream_t st) {
  const size_t lds = 2 * 289 * 128 + 3 * 3 * 64 * 128;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = a.N * (a.H / 16) * (a.W / 16);
  const int cbn = a.K / 64;
  const int n_items = tiles * cbn;
  const int cus = device_cus();
  const int grid = n_items < cus ? n_items : cus;
  const int kshift = (cbn > 1 && (cbn & (cbn - 1)) == 0 && (grid & (cbn - 1)) == 0) ? __builtin_ctz(cbn) : -1;
  hipLaunchKernelGGL(conv_s2d_kernel, dim3(grid), dim3(512), lds, st, a, tiles, n_items, kshift);
  return hipGetLastError();
}

const char* conv_s2d_name() { return "sslcr::conv_s2d_kernel"; }

}  // namespace sslcr

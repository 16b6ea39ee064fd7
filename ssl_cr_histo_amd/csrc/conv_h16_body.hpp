// The body of conv3x3_h16_kernel / conv3x3_h16s_kernel (conv_h16.hip), textually included into both kernel definitions: in scope are the
// template parameters T, BKO, WK, XF, WR, RAW, TW, the constant OSC and the kernel arguments a, tiles_total, n_items, kshift, row0.
// (As a __device__ function called from two __global__ wrappers the plain instance came out at 256 registers + 88 B of scratch where the
//  kernel written out has 251 + 0: the include keeps the compiler's view of the dominant instance what it was.)
  constexpr int NT = 256 * WK;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CE = 8 * EPC;                 // channels per 128-byte slab
  static_assert(TW == 16 || (TW == 8 && !WR), "tile forms");
  constexpr int TH = TW, NI = 256 / (TW * TH), HH = TH + 2, HWD = TW + 2, PITCH = TW == 16 ? 24 : 10;
  constexpr int HP = TW == 16 ? HH * HWD : 256;   // staged halo pixels: the 18x18 halo, or the four images' interiors
  constexpr int NLD = (HP * 8 + NT - 1) / NT; // 16-byte halo loads per thread per stage
  constexpr int TK = BKO / (16 * WK), TP = 4;
  constexpr int HBUF = NI * HH * PITCH * 128, WBUF = BKO * 128, TPB = 3;
  static_assert(NLD <= 16, "staging shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_halo = smem;
  constexpr int NRING = WR ? 3 : 2;           // tap groups held in LDS
  char* s_w = smem + HBUF;                    // [NRING][TPB][BKO][128 B]
  float* s_scale = reinterpret_cast<float*>(smem + HBUF + NRING * TPB * WBUF);
  float* s_shift = s_scale + a.C;
  // BatchNorm (sum, sumsq) of this workgroup's current kout block, per 64-pixel wave row: [4][2][BKO].  Items add into
  // it in place (each entry has exactly one writer lane, so the order -- and the fp32 result -- is deterministic);
  // it is written out as ONE set of four partial rows per workgroup and kout block instead of four rows per tile
  // (40960 partial rows -> 1024 for the layer1 shape: the second-stage row reduction was 2.4 % of the step).
  float* s_stat = s_shift + a.C;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 3, wk = wave >> 2;
  const int tiles_w = TW == 16 ? a.W / TW : 1, tiles_h = TW == 16 ? a.H / TH : 1;
  // segments (sslcr_conv_desc.seg_images): the grid is nseg equal groups of workgroups, group s walks the tiles of images
  // [s * seg_images, (s + 1) * seg_images) with that segment's prologue -- tiles_total / n_items are then PER SEGMENT.  A
  // workgroup's four statistics rows (index blockIdx.x * 4 + ...) therefore belong to one segment.
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int G = gridDim.x / nseg;
  const int seg = nseg > 1 ? (int)blockIdx.x / G : 0, lb = (int)blockIdx.x - seg * G;
  const int seg_n0 = seg * a.seg_images;
  if (XF) {
    const float* isc = a.in_scale + (size_t)seg * a.seg_stride;
    const float* ish = a.in_shift + (size_t)seg * a.seg_stride;
    for (int c = tid; c < a.C; c += NT) { s_scale[c] = isc[c]; s_shift[c] = ish[c]; }
  }
  for (int i = tid; i < 8 * BKO; i += NT) s_stat[i] = 0.f;
  // the (folded-BatchNorm) bias of all K outputs: read from LDS in the epilogue.  As global loads -- even skipped ones, when
  // there is no bias -- they put a compiler vmcnt(0) in front of the output stores
  float* s_bias = s_stat + 8 * BKO;
  // BatchNorm-backward front end (sslcr_conv_desc.mask_x): s_bias holds the BatchNorm's scale, two more arrays its shift and mean
  const bool mk = !OSC && !XF && !RAW && a.mask_x != nullptr;      // (the output-scale instance carries no mask body)
  float* s_msh = s_bias + a.K;
  float* s_mmu = s_msh + a.K;
  // eval-mode BatchNorm scale kept out of the filters (sslcr_conv_desc.out_scale; never together with the mask): the array behind s_bias
  float* s_osc = s_msh;
  for (int i = tid; i < a.K; i += NT) {
    const size_t mo = (size_t)seg * a.seg_stride + i;          // the mask's BatchNorm is the segment's own
    s_bias[i] = mk ? a.mask_scale[mo] : (a.bias ? a.bias[i] : 0.f);
    if (mk) { s_msh[i] = a.mask_shift[mo]; s_mmu[i] = a.mask_mean[mo]; }
    if constexpr (OSC) s_osc[i] = a.out_scale[i];
  }
  const float relu_lo = a.in_relu ? 0.f : -__builtin_inff();
  const float out_lo = a.relu ? 0.f : -__builtin_inff();

  // XCD-aware walk: blocks land on XCD (blockIdx % 8); each XCD takes a contiguous run of tiles per round so that
  // neighbouring tiles' shared halo rows hit the same L2.
  const int first = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);
  if (first >= n_items) return;

  // ---- per-thread staging roles, fixed for the whole walk
  const int chunk = tid & 7;
  int rel[NLD], st_off[NLD];
  unsigned long long edge = 0;                // 4 bits per entry: on the top / bottom / left / right halo ring
  unsigned hvalid = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int sp = (tid >> 3) + (NT / 8) * i;
    rel[i] = 0; st_off[i] = 0;
    if (TW == 8) {
      // interior pixel sp of the tile's four images (contiguous in NHWC): image sp >> 6, row (sp >> 3) & 7, column sp & 7
      rel[i] = sp;
      const int hc = (sp & 7) + 1;
      const int hp = ((sp >> 6) * HH + ((sp >> 3) & 7) + 1) * PITCH + hc;
      st_off[i] = hp * 128 + ((chunk ^ (hc & 7)) << 4);
      hvalid |= 1u << i;
    } else if (sp < HP) {
      const int hr = sp / HWD, hc = sp - hr * HWD;
      rel[i] = (hr - 1) * a.W + hc - 1;
      const int hp = hr * PITCH + hc;
      st_off[i] = hp * 128 + ((chunk ^ (hp & 7)) << 4);
      hvalid |= 1u << i;
      edge |= (unsigned long long)((hr == 0) | ((hr == HH - 1) << 1) | ((hc == 0) << 2) | ((hc == HWD - 1) << 3)) << (4 * i);
    }
  }
  // weight DMA.  Only ONE wave of each SIMD's pair issues it -- the YOUNGER one (waves 4-7; with WK = 1 there is no pair and all
  // four load).  The older wave of a pair wins the matrix pipe (profiles/r04_partner_instruction_cost.txt: 94 % of it), so after
  // every barrier the younger wave sits out the older one's MFMAs anyway: that is where its 1 KiB DMA instructions (60-180 cycles
  // of issue each, MI355X_MICROARCH.md) cost nothing, instead of both waves of a SIMD issuing theirs at the same moment with the
  // pipe idle.  Instruction i of loader wave lw fills LDS rows [i*32 + lw*8, +8) of a tap; lane -> (row, 16-byte slot).  wperm_inv
  // permutes bit fields, so row i*32 + r comes from kout row wperm_inv(i*32) + wperm_inv(r): ONE per-lane source offset, the
  // instruction's share is wave-uniform (soffset).
  const bool loader = WK == 1 || wave >= 4;
  constexpr int WLI = BKO / 32;               // DMA instructions per loader wave per tap
  const int lw = wave & 3;
  constexpr int WROWS = BKO / WLI;            // LDS rows between a wave's consecutive instructions
  int wsrc0;
  {
    const int rr = lw * 8 + (lane >> 3);
    const int krow = wperm_inv<TK>(rr);
    const int c16 = (lane & 7) ^ (rr & 7);
    wsrc0 = (int)(((size_t)krow * 9 * a.C + c16 * EPC) * sizeof(T));
  }
  // fragment addresses: everything but these 8 registers is an immediate offset
  int Bb[3][2], Ab[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      Bb[s][kk] = TW == 16 ? ((wp * 4) * PITCH + li + s) * 128 + ((ci ^ ((li + s) & 7)) << 4)
                           : ((wp * HH + (li >> 3)) * PITCH + (li & 7) + s) * 128 + ((ci ^ (((li & 7) + s) & 7)) << 4);
    Ab[kk] = (wk * (BKO / WK) + li) * 128 + ((ci ^ (li & 7)) << 4);
  }
  const char* xg = reinterpret_cast<const char*>(a.x) + (size_t)chunk * EPC * sizeof(T);
  const char* wg = reinterpret_cast<const char*>(a.w);
  const int nslabs = a.C / CE;

  // Item order.  kout-block-major (item = kb * tiles + tile) keeps a workgroup on one kout block for a long run of items, but the
  // KBn kout blocks of a tile -- which read the same halo -- are then half a kernel apart and every one fetches it again
  // (layer3, K = 256: the input was read twice, r03 PMC).  Where the walk stride G is a multiple of KBn the items go kout-block-
  // FASTEST instead (item = tile * KBn + kb): a workgroup still sees ONE kout block (item % KBn = first % KBn for all its items,
  // so its statistics rows are published once), and a tile's KBn blocks sit next to each other in one XCD's run of the round.
  // KBn = K / BKO.
  // (kshift >= 0 from the launcher: power-of-two block counts only -- mask and shift, no division; -1 = kout-block-major)
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
    q.n0 = (t / tiles_h) * NI + seg_n0;
    q.h0 = th_i * TH; q.w0 = tw_i * TW;
    q.origin = (q.n0 * a.H + q.h0) * a.W + q.w0;
    q.out = TW == 8 ? 0ull : (unsigned long long)((q.h0 == 0) | ((q.h0 + TH >= a.H) << 1) | ((q.w0 == 0) << 2) | ((q.w0 + TW >= a.W) << 3)) *
                             0x1111111111111111ull;
    return q;
  };

  u32x4_t hreg[NLD];
  unsigned hin = 0;                           // hreg[i] holds image data (not zero padding)
  // branch-free: padding entries load the tile origin (a valid address) and are zeroed when staged
  auto load_halo = [&](const Geo& q, int slab) {
    const unsigned long long bad = edge & q.out;
    hin = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      const int idx = q.origin + (ok ? rel[i] : 0);
      // (non-temporal: the input streams through once per kout block, and what should stay in the caches is the OUTPUT, which the
      //  next kernel reads -- r04, same box, six alternations: -0.06 ms per step; non-temporal output stores are +0.33 ms)
      hreg[i] = ld16_nt(xg + ((size_t)idx * a.C + slab * CE) * sizeof(T));
      hin |= (ok ? 1u : 0u) << i;
    }
  };
  float sc[EPC], sh[EPC];
  auto load_affine = [&](int slab) {
    if (XF) {
      const int cb = slab * CE + chunk * EPC;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc[e] = s_scale[cb + e]; sh[e] = s_shift[cb + e]; }
    }
  };
  auto xform_one = [&](int i) {                // hreg[i] -> what LDS must hold
    u32x4_t v = hreg[i];
    if (XF) {
      float f[EPC];
      Elem<T>::unpack(v, f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = clamp_lo(fmaf(f[e], sc[e], sh[e]), relu_lo);
      v = PackH<T>::run(f);
    }
    const bool ok = (hin >> i) & 1u;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
    hreg[i] = v;
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if ((hvalid >> i) & 1u) st16(s_halo + st_off[i], hreg[i]);
  };
  // DMA the three taps tap0..tap0+2 of (kout block k0, slab) into ring half `half`
  // the DMA is the MUBUF form (LdsDma, common.hpp): behind global_load_lds 12 of a stage's 18 steps began with s_waitcnt lgkmcnt(0)
  LdsDma wdma;
  wdma.init(wg, 0x7fffffffu);
  auto dma_w = [&](int k0, int slab, int tap0, int half) {
    if (!loader) return;
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
      const int soff = (int)(((size_t)k0 * 9 * a.C + (size_t)(tap0 + tt) * a.C + slab * CE) * sizeof(T));
#pragma unroll
      for (int i = 0; i < WLI; ++i) {
        char* dst = s_w + (half * TPB + tt) * WBUF + (i * WROWS + lw * 8) * 128;
        wdma.load16(dst, wsrc0, soff + wperm_inv<TK>(i * WROWS) * 9 * a.C * (int)sizeof(T));
      }
    }
  };

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragments are double-buffered across the WHOLE walk: step i (tap i/2, 64-byte half kk = i&1) requests the fragments of
  // step i+1 before it runs its 16*TK/4 MFMAs, across barriers too
  u32x4_t A[2][TK], B[2][TP];
  auto frags = [&](int buf, int step, const char* ringg) {     // ringg: the ring half holding tap (step>>1)'s group
    const int tap = step >> 1, kk = step & 1;
    const int r = tap / 3, s = tap - 3 * r;
    // read order = the order in which the step's MFMAs (t-major) first need a fragment: A0, then every B, then A1.. -- the reads
    // are issued one per two MFMAs of the previous step, so B[TP-1] is requested 11 MFMAs before its first use instead of 5
    A[buf][0] = ld16(ringg + Ab[kk] + s * WBUF);
#pragma unroll
    for (int p = 0; p < TP; ++p) B[buf][p] = ld16(s_halo + Bb[s][kk] + ((TW == 16 ? p : 2 * p) + r) * (PITCH * 128));
#pragma unroll
    for (int t = 1; t < TK; ++t) A[buf][t] = ld16(ringg + Ab[kk] + s * WBUF + t * 2048);
  };

  if (TW == 8) {          // the padding ring (and everything else) once; a stage rewrites the interiors only
    for (int i = tid; i < HBUF / 16; i += NT) st16(s_halo + i * 16, u32x4_t{0u, 0u, 0u, 0u});
    __syncthreads();
  }
  // ---- pipeline fill: halo of (first item, slab 0) and ring half 0 <- taps 0..2
  Geo cur = geom(first);
  dma_w(cur.k0, 0, 0, 0);
  if constexpr (WR) { dma_w(cur.k0, 0, 3, 1); dma_w(cur.k0, 0, 6, 2); }
  load_halo(cur, 0);
  if (XF) __syncthreads();
  load_affine(0);
#pragma unroll
  for (int i = 0; i < NLD; ++i) xform_one(i);
  store_halo();
  SSLCR_WAIT_VM0();
  __syncthreads();
  frags(0, 0, s_w);

  char* yg = reinterpret_cast<char*>(a.y);
  // no residual with an input transform (conv_h16_ok): a residual load in the epilogue -- even one skipped at run time --
  // made the compiler put a vmcnt(0) in front of every output store, i.e. eight serial write round trips per item
  const char* rg = (XF || RAW) ? nullptr : reinterpret_cast<const char*>(a.mask_x ? a.mask_x : a.residual);   // same shape, same prefetch
  // bf16 residual (teacher conv2 / the skip gradient of a block's first dgrad): requested with the next halo in the middle
  // of the item's LAST stage, so its HBM round trip sits under six steps of MFMAs instead of in front of the epilogue
  // (the epilogue-time load cost 57-80 us per layer1 launch, one exposed latency per tile)
  constexpr bool RPRE = sizeof(T) == 2 && !XF && !RAW;
  constexpr int RQ = 4 * TK / EPC;
  u32x4_t rres[RPRE ? TP : 1][RPRE ? RQ : 1];
  auto out_off = [&](const Geo& q, int p) {
    const int h = q.h0 + wp * 4 + p, w = q.w0 + li;
    const size_t pix = TW == 16 ? ((size_t)q.n0 * a.H + h) * a.W + w : (size_t)(q.n0 + wp) * 64 + p * 16 + li;
    return (pix * a.K + q.k0 + wk * (BKO / WK) + g * (4 * TK)) * sizeof(T);
  };
  int wb = 0, item = first, slab = 0;
  unsigned pub = 0;                            // kout blocks whose statistics rows this workgroup has published (bit per block)
  // (a static s_setprio 1 for waves 4-7 -- the arbitration losers of every contended issue slot -- measured 0.00 ms on the step, r04)
#ifdef SSLCR_H16_PROF
  unsigned long long h16_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long h16_begin = __builtin_readcyclecounter();
#endif
  for (;;) {
    // the stage after this one: next slab of this tile, or slab 0 of the next item (the last stage of the walk re-requests
    // itself: branch-free, and nobody reads what it stages)
    const bool last = slab + 1 == nslabs;
    const bool done = last && item + G >= n_items;
    const int nslab = last ? 0 : slab + 1;
    const Geo nxt = (last && !done) ? geom(item + G) : cur;
    const char* ring0 = s_w + wb * (TPB * WBUF);          // taps 0-2 and 6-8 of this stage
    const char* ring1 = s_w + (wb ^ 1) * (TPB * WBUF);    // taps 3-5, and taps 0-2 of the next stage

    // Schedule of a stage (18 steps, 3 tap groups G0 G1 G2 on alternating ring halves):
    //   group start : DMA the NEXT group's three taps into the half the previous group just released
    //   mid group   : vmcnt(0) + barrier P  -> the next group's weights are published one and a half taps before they are
    //                 needed, so the fragment prefetch of its first step does not wait behind a barrier
    //   group end   : barrier F            -> everybody is done with this group's half; it may be overwritten
    //   mid G1      : request the next stage's halo (HBM), mid G2 it has landed (same vmcnt(0)); steps 15-17 transform it
    //   stage end   : barrier, six ds_write_b128, barrier -- the only place the fragment pipeline drains
    auto load_res = [&]() {
      if constexpr (RPRE) {
        if (rg && last) {
#pragma unroll
          for (int p = 0; p < TP; ++p)
#pragma unroll
            for (int q = 0; q < RQ; ++q) rres[p][q] = ld16(rg + out_off(cur, p) + q * 16);
        }
      }
    };
    if constexpr (WR) {
      load_res();
      load_halo(nxt, 0);
    } else {
      dma_w(cur.k0, slab, 3, wb ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      // the scheduler fences keep "request the next fragments, then run this step's MFMAs (with the halo transform under
      // them)" in that order; left alone the compiler serialises read -> wait -> MFMA, sinks prefetches down to their
      // first use and moves the VALU work into the barrier-to-barrier section of the stage boundary
      if (i < 17) frags((i + 1) & 1, i + 1, WR ? s_w + ((i + 1) / 6) * (TPB * WBUF) : ((((i + 1) / 6) & 1) ? ring1 : ring0));
      if (sizeof(T) != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) MmaH<T>::run(A[i & 1][t], B[i & 1][p], acc[t][p]);
      if (i < 17 && sizeof(T) == 2) {
        // bf16: the next step's fragment reads are spread between this step's MFMAs (one read per ~TK*TP/(TK+TP) MFMAs)
        // instead of all being issued first: +2..7 % on every shape (micro-benchmark of the bare loop: +5 %)
#pragma unroll
        for (int q = 0; q < TK + TP; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (TK * TP) / (TK + TP), 0);
        }
      }
      if (i >= 15) {
#pragma unroll
        for (int j = (i - 15); j < NLD; j += 3) xform_one(j);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (WR && i == 14) {
        SSLCR_WAIT_VM0();                                 // halo + residual (and the previous epilogue's stores) have landed
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!WR && i % 6 == 2) {
        H16_T(tp0);
        SSLCR_WAIT_VM0();
        H16_T(tp1);
        SSLCR_BARE_BARRIER();                             // P
        H16_T(tp2);
        H16_ACC(0, tp1 - tp0); H16_ACC(1, tp2 - tp1);
        if (i == 8) {
          load_res();
          load_halo(nxt, nslab);
          load_affine(nslab);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!WR && (i == 5 || i == 11)) {
        H16_T(tf0);
        SSLCR_BARE_BARRIER();                             // F (the reads of the released half fed MFMAs that have been issued)
        H16_T(tf1);
        H16_ACC(2, tf1 - tf0);
        if (i == 5) dma_w(cur.k0, slab, 6, wb); else dma_w(nxt.k0, nslab, 0, wb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    H16_T(ts0);
    SSLCR_BARE_BARRIER();                     // every wave is done with this stage's halo (its last reads fed step 17's MFMAs)
    store_halo();
    __syncthreads();
    H16_T(ts1);
    H16_ACC(3, ts1 - ts0); H16_ACC(5, 1);
    wb ^= 1;
    frags(0, 0, WR ? s_w : s_w + wb * (TPB * WBUF));      // first fragments of the next stage: in flight under the epilogue
    __builtin_amdgcn_sched_barrier(0);

    H16_T(te0);
    if (last) {
      // ---------------- epilogue of the finished item; its stores drain under the next item's taps
      const int kb = cur.k0 + wk * (BKO / WK) + g * (4 * TK);
      float s1[4 * TK], s2[4 * TK];
      if (mk) {
        // g = y * (scale*x + shift > 0) ; partial sums of g and g*(x - mean) over this wave's 64 pixels
        // one 16-byte chunk of channels at a time: its 3 x EPC constants live only while the four pixel rows are processed
#pragma unroll
        for (int q = 0; q < 4 * TK / EPC; ++q) {
          float msc[EPC], msh[EPC], mmu[EPC], a1[EPC], a2[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            msc[e] = s_bias[kb + q * EPC + e]; msh[e] = s_msh[kb + q * EPC + e]; mmu[e] = s_mmu[kb + q * EPC + e];
            a1[e] = 0.f; a2[e] = 0.f;
          }
#pragma unroll
          for (int p = 0; p < TP; ++p) {
            const size_t off = out_off(cur, p);
            float xr[EPC], vq[EPC];
            if constexpr (RPRE) Elem<T>::unpack(rres[p][q], xr);
            else Elem<T>::unpack(ld16(rg + off + q * 16), xr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const int idx = q * EPC + e;
              const float y = acc[idx >> 2][p][idx & 3];
              const float gv = fmaf(xr[e], msc[e], msh[e]) > 0.f ? y : 0.f;
              vq[e] = gv;
              a1[e] += gv;
              a2[e] = fmaf(gv, xr[e] - mmu[e], a2[e]);
            }
            st16(yg + off + q * 16, PackH<T>::run(vq));
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) { s1[q * EPC + e] = a1[e]; s2[q * EPC + e] = a2[e]; }
        }
      } else {
      if constexpr (RAW) {
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const size_t off = out_off(cur, p);
#pragma unroll
          for (int q = 0; q < 4 * TK / EPC; ++q) {
            float vq[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) vq[e] = acc[(q * EPC + e) >> 2][p][(q * EPC + e) & 3];
            st16(yg + off + q * 16, PackH<T>::run(vq));
          }
        }
      } else if constexpr (OSC) {
#pragma unroll
      for (int q = 0; q < 4 * TK / EPC; ++q) {
        float bq[EPC], sq[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { bq[e] = s_bias[kb + q * EPC + e]; sq[e] = s_osc[kb + q * EPC + e]; }
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const size_t off = out_off(cur, p);
          float vq[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const int idx = q * EPC + e;
            vq[e] = fmaf(acc[idx >> 2][p][idx & 3], sq[e], bq[e]);
          }
          if (rg) {
            float rr[EPC];
            if constexpr (RPRE) Elem<T>::unpack(rres[p][q], rr);
            else Elem<T>::unpack(ld16(rg + off + q * 16), rr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) vq[e] = clamp_lo(vq[e], out_lo);
          st16(yg + off + q * 16, PackH<T>::run(vq));
        }
      }
      } else {
      float bias[4 * TK];
#pragma unroll
      for (int j = 0; j < 4 * TK; ++j) bias[j] = s_bias[kb + j];
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const size_t off = out_off(cur, p);
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
            if constexpr (RPRE) Elem<T>::unpack(rres[p][q], rr);
            else Elem<T>::unpack(ld16(rg + off + q * 16), rr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
          }
          // (one v_med3 per value; written as `if (a.relu) fmaxf` the compiler if-converted it to a compare + select pair)
#pragma unroll
          for (int e = 0; e < EPC; ++e) vq[e] = clamp_lo(vq[e], out_lo);
          st16(yg + off + q * 16, PackH<T>::run(vq));
        }
      }
      }
      if (a.stats) {
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
      }
      }
      // s1 / s2: this lane's sums over its four pixel rows; now over the 16 pixel columns (the lanes of a DPP row)
      constexpr bool FOLD = (4 * TK) % 16 == 0;
      if (a.stats) {
        if constexpr (FOLD) {
          // per 16 values: quad q of the row ends up with the sums of kouts 4q..4q+3 in s[0..3] (row16_fold16: 32 DPP adds per 16
          // values where 16 row16_sum calls are 128 instructions as compiled); its first lane adds them into the workgroup's sums
#pragma unroll
          for (int f = 0; f < 4 * TK / 16; ++f) {
            row16_fold16(*reinterpret_cast<float (*)[16]>(s1 + 16 * f));
            row16_fold16(*reinterpret_cast<float (*)[16]>(s2 + 16 * f));
          }
          if ((li & 3) == 0) {
            float* sp = s_stat + (wp * 2) * BKO + wk * (BKO / WK) + g * (4 * TK) + (li >> 2) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
              for (int f = 0; f < 4 * TK / 16; ++f) {
                const float* sv = (h ? s2 : s1) + 16 * f;
                f32x4_t* slot = reinterpret_cast<f32x4_t*>(sp + h * BKO + 16 * f);
                f32x4_t v = *slot;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sv[e];
                *slot = v;
              }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4 * TK; ++i) { s1[i] = row16_sum(s1[i]); s2[i] = row16_sum(s2[i]); }
        }
        if (!FOLD && li == 0) {
          // each (wave row, kout) entry has exactly one writer LANE in the workgroup, so the running sums are updated with plain
          // 16-byte reads and writes: the 8 * TK LDS float atomics this replaces held the LDS pipe ~40 cycles each (measured on the
          // ping-pong form, tools/microbench/pp64_phase_bench.hip: 1400 -> 250 cycles per tile)
          float* sp = s_stat + (wp * 2) * BKO + wk * (BKO / WK) + g * (4 * TK);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float* sv = h ? s2 : s1;
#pragma unroll
            for (int c4 = 0; c4 < TK; ++c4) {
              f32x4_t* slot = reinterpret_cast<f32x4_t*>(sp + h * BKO + c4 * 4);
              f32x4_t v = *slot;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += sv[c4 * 4 + e];
              *slot = v;
            }
          }
        }
        if (done || nxt.k0 != cur.k0) {           // last item of this kout block: publish the four rows (uniform branch)
          pub |= 1u << (cur.k0 / BKO);
          __syncthreads();
          for (int i = tid; i < 8 * BKO; i += NT) {
            const int rw = i / BKO, c = i - rw * BKO;        // rw = wave row * 2 + (0: sum, 1: sumsq)
            a.stats[((size_t)(row0 + blockIdx.x * 4 + (rw >> 1)) * 2 + (rw & 1)) * a.K + cur.k0 + c] = s_stat[i];
            s_stat[i] = 0.f;
          }
          __syncthreads();
        }
      }
#ifdef SSLCR_H16_PROF
      { H16_T(te1); H16_ACC(4, te1 - te0); }
#endif
      if (done) break;
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      item += G;
      cur = nxt;
    }
    slab = nslab;
  }
#ifdef SSLCR_H16_PROF
  if (blockIdx.x == 0 && lane == 0) {
    h16_t[6] = __builtin_readcyclecounter() - h16_begin;
    for (int i = 0; i < 8; ++i) g_h16_prof[wave][i] = h16_t[i];
  }
#endif
  // the rows of the kout blocks this workgroup never reached are zeros.  (Kept as a set, not a range: where a segment has fewer
  // tiles than workgroups -- three TripletNet branches of 32 four-image tiles on 85 workgroups each -- a kout-block-major walk
  // b, b + G skips blocks.)
  if (a.stats) {
    for (int kbi = 0; kbi < a.K / BKO; ++kbi) {
      if ((pub >> kbi) & 1u) continue;
      for (int i = tid; i < 8 * BKO; i += NT) {
        const int rw = i / BKO, c = i - rw * BKO;
        a.stats[((size_t)(row0 + blockIdx.x * 4 + (rw >> 1)) * 2 + (rw & 1)) * a.K + kbi * BKO + c] = 0.f;
      }
    }
  }

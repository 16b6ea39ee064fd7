// 3x3 / stride 1 / pad 1 NHWC convolution for the bf16 128- and 256-channel maps (ResNet18 layers 2-3: forward, eval-fused forward,
// both dgrads; torchvision BasicBlock convs reached from models/net.py:32,77): the PING-PONG of conv_pp64.hip with a filter bank
// that does not fit LDS -- one 64-kout x 64-channel bank (9 taps, 72 KB) that is REFILLED tap group by tap group behind its last
// reader.
//
// Why (round 4).  conv3x3_h16<bf16,128> is the step's dominant kernel and sat at 0.41 of the MFMA roof for three rounds.  Its phase
// timing (profiles/r03_final_h16_phase.txt) says the matrix pipe is busy 9216 of the 17 900 cycles of a stage; the rest is the output
// stage (3000-4000 cycles per stage, every wave in it at the same time, no MFMA anywhere on the CU), the stage-end halo swap (1600)
// and start-up bubbles behind its seven barriers.  All eight waves walk through the same phase at the same time.  The layer1 kernel
// (conv_pp64.hip) fixed exactly that with two wave groups half a period apart, but it relies on the whole filter bank being resident.
//
// Here: the workgroup's eight waves are two groups of four (one wave per SIMD each).  A group owns a 16x16-pixel x 64-kout tile
// (wave = 4 rows x 16 pixels x 64 kouts) of ONE kout block -- the workgroup's kout block is fixed for the whole walk (items go
// kout-block-fastest and the grid is a multiple of the block count, as in conv_h16.hip) -- and alternates
//     M  one 64-channel slab: 18 steps (9 taps x 2 K-halves) of MFMAs out of its own halo buffer and the shared bank;
//     W  everything else: BatchNorm+ReLU transform of the next stage's halo and its ds_writes, and after the LAST slab of a tile
//        the output stage (bias / residual / ReLU or the BatchNorm-backward front end, pack, stores, BatchNorm partial sums).
// Group 1 runs one phase behind group 0, so the bank of stage i (kout block, slab i mod C/64) is read by group 0 in phase 2i and by
// group 1 in phase 2i+1, and stage i+1's bank must be there in phase 2i+2.  Every phase is three barrier intervals of three taps.
// Group 1, the bank's LAST reader, refills it as it goes: after the barrier that ends its interval j it DMAs the three taps of
// interval j for stage i+1 (global_load_lds, no registers) and waits for them before the barrier that ends its M phase; the last
// three taps can only go once group 1 has left M -- group 0 issues them at the start of its next M phase and waits (counted vmcnt:
// only its own, younger halo loads may stay out) before the barrier in front of the interval that prefetches them.  One bank serves
// two pixel tiles, so a tile sees half the weight bytes per MFMA of conv3x3_h16 (32 B against 64 B: the staged-bytes term of
// DESIGN section 4).  A barrier of the M group is a bare s_barrier (it publishes nothing, and every LDS read of the taps the barrier
// releases has fed an MFMA that was issued before it); the W group's last barrier carries lgkmcnt(0) for its halo writes.
//
// LDS: bank 72 KB + two 18x18-pixel halos at an 18-pixel pitch 81 KB + scale/shift of all C channels, this block's bias / mask
// constants and the two groups' statistics <= 7 KB: 160 KB for C = 256.
#include <stdlib.h>

#include "kernels.hpp"

namespace sslcr {

typedef const __attribute__((address_space(1))) void* gptr_pr;
typedef __attribute__((address_space(3))) void* lptr_pr;

// W side: LDS visibility + rendezvous without the workgroup-scope fence of __syncthreads() (with global loads and stores in flight
// the fence becomes s_waitcnt vmcnt(0)); M side: rendezvous only
#define SSLCR_PR_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SSLCR_PR_BAR_BARE() asm volatile("s_barrier" ::: "memory")

// phase timing for tools/microbench/ppr_phase_bench.hip (-DSSLCR_PR_PROF): per wave of workgroup 0, shader cycles spent in M, at the
// barrier behind M, in W, at the barrier behind W
#ifdef SSLCR_PR_PROF
__device__ unsigned long long g_pr_prof[8][8];
#define PR_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define PR_ACC(i, d) pr_t[i] += (d)
#else
#define PR_T(v)
#define PR_ACC(i, d)
#endif

// Which kout (within the workgroup's 64-kout block) a lane's accumulator element holds: MFMA tile t, accumulator row group g
// (= lane >> 4), element j -- the mapping of conv_pp64.hip: a lane's 16 kouts are two runs of 8 consecutive channels, 32 channels
// apart, so the four row groups of one pixel write one contiguous 64-byte segment per store instruction.
#define PR_CH(t, g, j) ((((t) >> 1) * 32) + ((g) * 8) + (((t) & 1) * 4) + (j))
// LDS row of the bank -> kout row of the block: fragment row 4g + j of tile t feeds accumulator (t, g, j)
__device__ __forceinline__ int pr_row_kout(int rr) {
  const int t = rr >> 4, gq = (rr >> 2) & 3, j = rr & 3;
  return PR_CH(t, gq, j);
}

// XF: the producer's BatchNorm(+ReLU) is applied to the input on its way into LDS (a.in_scale != nullptr); never with a residual
// OP: 0 plain output stage (bias / ReLU / statistics), 1 + residual (a.residual), 2 BatchNorm-backward front end (a.mask_x).
template <bool XF, int OP>
__global__ __launch_bounds__(512, 2) void conv3x3_ppr_kernel(const ConvArgs a, const int tiles_total, const int hgs, const int kshift) {
  typedef bf16_t T;
  constexpr int EPC = 8, BKO = 64, TK = 4, TP = 4;
  constexpr int TW = 16, TH = 16, HH = 18, HWD = 18, PITCH = 18, HP = HH * HWD;
  constexpr int GT = 256;                              // threads per group
  constexpr int NLD = (HP * 8 + GT - 1) / GT;          // 11 sixteen-byte halo chunks per thread and stage
  constexpr int WBUF = BKO * 128, HBUF = HH * PITCH * 128;
  constexpr bool RPRE = OP != 0;                       // a residual / mask operand (never with an input transform)
  static_assert(!(XF && OP != 0), "no residual with an input transform");
  constexpr int RQ = 4 * TK / EPC;                     // 2 sixteen-byte chunks of a lane's 16 kouts
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;                                    // [9 taps][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int gtid = tid & (GT - 1);
  const int g = lane >> 4, li = lane & 15;
  char* s_halo = smem + 9 * WBUF + grp * HBUF;
  float* s_f = reinterpret_cast<float*>(smem + 9 * WBUF + 2 * HBUF);
  float* s_bias = s_f;                                 // bias of this kout block, or the BatchNorm scale of the mask_x front end
  float* s_msh = s_f + 64;
  float* s_mmu = s_f + 128;
  float* s_stat0 = s_f + 192;                          // [2 groups][4 wave rows][2][64] partial (sum, sumsq)
  float* s_stat = s_stat0 + grp * 512;
  float* s_scale = s_f + 192 + 1024;                   // [C], then s_shift [C]
  float* s_shift = s_scale + a.C;

  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  const int nslab = a.C / 64;
  // segments (sslcr_conv_desc.seg_images): nseg equal groups of workgroups, group s walks the tiles of its own images with its own
  // prologue / mask BatchNorm; tiles_total is then PER SEGMENT.  Statistics rows: conv3x3_h16 would launch hgs workgroups per
  // segment and the caller sized the rows for that; workgroup lb of segment s owns row block s * hgs + lb (conv_h16.hip)
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int G = gridDim.x / nseg;
  const int seg = nseg > 1 ? (int)blockIdx.x / G : 0, lb = (int)blockIdx.x - seg * G;
  const int seg_n0 = seg * a.seg_images;
  const int rb = seg * hgs + lb;
  // XCD-aware walk (blocks land on XCD blockIdx % 8): each XCD takes a contiguous run of items per round, so the K / 64 kout
  // blocks of one tile pair -- which read the same halos -- are neighbours on one XCD
  const int vb = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);
  // item = pair * KBn + kb (kout block fastest); the walk stride G is a multiple of KBn = 1 << kshift: kb is this workgroup's own
  const int kbi = vb & ((1 << kshift) - 1);
  const int k0 = kbi * BKO;
  const int pair0 = vb >> kshift, pstep = G >> kshift;
  const int npairs = (tiles_total + 1) >> 1;
  if (pair0 >= npairs) {
    // (the launcher sizes the grid so that every workgroup has a pair; the statistics rows must be defined all the same)
    if (a.stats)
      for (int i = tid; i < 8 * a.K; i += 512) a.stats[(size_t)rb * 8 * a.K + i] = 0.f;
    return;
  }
  const int npw = (npairs - pair0 + pstep - 1) / pstep;               // tile pairs of this workgroup
  const int nst = npw * nslab;                                        // stages per group
  constexpr bool mk = OP == 2;
  const float relu_lo = a.in_relu ? 0.f : -__builtin_inff();

  // ---- weights: lane offset of the row this lane fetches (the lane picks its SOURCE chunk so that the linear DMA placement is the
  //      fragment-ordered, swizzled tile); the stage's slab and the tap are uniform displacements
  const char* wg = reinterpret_cast<const char*>(a.w) + (size_t)k0 * 9 * a.C * sizeof(T);
  const unsigned tap_bytes = (unsigned)a.C * (unsigned)sizeof(T);
  {
    // first bank (slab 0): all eight waves, wave w fills rows [8w, 8w+8) of every tap
    const int rr = wave * 8 + (lane >> 3);
    const unsigned off = ((unsigned)pr_row_kout(rr) * 9u * (unsigned)a.C + (unsigned)(((lane & 7) ^ (rr & 7)) * EPC)) * (unsigned)sizeof(T);
#pragma unroll
    for (int tt = 0; tt < 9; ++tt)
      __builtin_amdgcn_global_load_lds((gptr_pr)(wg + off + tt * tap_bytes), (lptr_pr)(s_w + tt * WBUF + (wave * 8) * 128), 16, 0, 0);
  }
  // refills: the four waves of ONE group fill a tap, wave wq rows [16 wq, 16 wq + 16) as two instructions
  unsigned wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = wq * 16 + i * 8 + (lane >> 3);
    wsrc[i] = ((unsigned)pr_row_kout(rr) * 9u * (unsigned)a.C + (unsigned)(((lane & 7) ^ (rr & 7)) * EPC)) * (unsigned)sizeof(T);
  }
  auto dma_taps = [&](int tap0, int slab) {            // three taps of (this kout block, slab) into their bank slots
    const char* src = wg + (unsigned)slab * 128u;
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr_pr)(src + (tap0 + tt) * tap_bytes + wsrc[i]), (lptr_pr)(s_w + (tap0 + tt) * WBUF + (wq * 16 + i * 8) * 128),
                                         16, 0, 0);
  };

  if (XF)
    for (int c = tid; c < a.C; c += 512) { s_scale[c] = a.in_scale[(size_t)seg * a.seg_stride + c]; s_shift[c] = a.in_shift[(size_t)seg * a.seg_stride + c]; }
  if (tid < 64) {
    const size_t mo = (size_t)seg * a.seg_stride + k0 + tid;     // the mask's BatchNorm is the segment's own
    s_bias[tid] = mk ? a.mask_scale[mo] : (a.bias ? a.bias[k0 + tid] : 0.f);
    if (mk) { s_msh[tid] = a.mask_shift[mo]; s_mmu[tid] = a.mask_mean[mo]; }
  }
  for (int i = tid; i < 1024; i += 512) s_stat0[i] = 0.f;

  // ---- per-thread halo staging roles (group-local), fixed for the whole walk
  const int chunk = gtid & 7;
  // one register per entry: bits 0-15 = pixel offset from the tile origin, biased by W + 1; bits 16-27 = LDS byte offset / 16
  int role[NLD];
  const int rel_bias = a.W + 1;
  unsigned long long edge = 0;                 // 4 bits per entry: on the top / bottom / left / right halo ring
  unsigned hvalid = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int sp = (gtid >> 3) + (GT / 8) * i;
    role[i] = rel_bias;
    if (sp < HP) {
      const int hr = sp / HWD, hc = sp - hr * HWD;
      role[i] = ((hr - 1) * a.W + hc - 1 + rel_bias) | ((((hr * PITCH + hc) * 128 + ((chunk ^ (hc & 7)) << 4)) >> 4) << 16);
      hvalid |= 1u << i;
      edge |= (unsigned long long)((hr == 0) | ((hr == HH - 1) << 1) | ((hc == 0) << 2) | ((hc == HWD - 1) << 3)) << (4 * i);
    }
  }
  // fragment addresses: everything but these 8 registers is an immediate offset
  int Bb[3][2], Ab[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
#pragma unroll
    for (int s = 0; s < 3; ++s) Bb[s][kk] = ((wq * 4) * PITCH + li + s) * 128 + ((ci ^ ((li + s) & 7)) << 4);
    Ab[kk] = li * 128 + ((ci ^ (li & 7)) << 4);
  }
  const char* xg = reinterpret_cast<const char*>(a.x) + (size_t)chunk * EPC * sizeof(T);
  const unsigned xrow = (unsigned)a.C * (unsigned)sizeof(T);           // bytes per pixel of x

  struct Geo { int origin, n0, h0, w0; unsigned long long out; };
  auto geom = [&](int tile) {
    Geo q;
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    q.n0 = t / tiles_h + seg_n0;
    q.h0 = th_i * TH; q.w0 = tw_i * TW;
    q.origin = (q.n0 * a.H + q.h0) * a.W + q.w0;
    q.out = (unsigned long long)((q.h0 == 0) | ((q.h0 + TH >= a.H) << 1) | ((q.w0 == 0) << 2) | ((q.w0 + TW >= a.W) << 3)) *
            0x1111111111111111ull;
    return q;
  };
  // the tile of this group's k-th pair; past the end (an odd tile count: group 1's last tile) it re-walks the last tile of the
  // tensor with its stores and statistics suppressed (same barrier count)
  auto tile_of = [&](int k, bool& live) {
    const int t = 2 * (pair0 + k * pstep) + grp;
    live = t < tiles_total;
    return live ? t : tiles_total - 1;
  };

  u32x4_t hreg[NLD];
  unsigned hin = 0;                            // hreg[i] holds image data (not zero padding)
  auto load_halo = [&](const Geo& q, int slab) {   // branch-free: padding entries load the tile origin and are zeroed when staged
    const unsigned long long bad = edge & q.out;
    hin = 0;
    const char* xs = xg + (unsigned)slab * 128u;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      int rl = role[i];
      asm volatile("" : "+v"(rl));               // keep the unpacked form out of the loop-invariant set (it would cost 11 registers)
      const int idx = q.origin + (ok ? (rl & 0xffff) - rel_bias : 0);
      hreg[i] = ld16(xs + (unsigned)idx * xrow);                       // 32-bit offsets from a uniform base (tensors < 4 GB)
      hin |= (ok ? 1u : 0u) << i;
    }
  };
  auto xform_store = [&](int slab) {           // hreg -> what LDS must hold -> this group's halo buffer
    float sc[EPC], sh[EPC];
    if (XF) {                                  // re-read per stage (four ds_read_b128): 16 registers the M phase does not carry
      const float* ps = s_scale + slab * 64 + chunk * EPC;
      const float* ph = s_shift + slab * 64 + chunk * EPC;
      const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(ps), c1 = *reinterpret_cast<const f32x4_t*>(ps + 4);
      const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(ph), h1 = *reinterpret_cast<const f32x4_t*>(ph + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = c0[e]; sc[4 + e] = c1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      u32x4_t v = hreg[i];
      if (XF) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = __builtin_amdgcn_fmed3f(fmaf(f[e], sc[e], sh[e]), relu_lo, __builtin_inff());
        v = PackH<T>::run(f);
      }
      const bool ok = (hin >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
      int rl = role[i];
      asm volatile("" : "+v"(rl));
      if ((hvalid >> i) & 1u) st16(s_halo + ((rl >> 12) & 0xffff0), v);
    }
  };

  char* yg = reinterpret_cast<char*>(a.y) + (size_t)k0 * sizeof(T);
  const char* rg = OP == 0 ? nullptr : reinterpret_cast<const char*>(OP == 2 ? a.mask_x : a.residual) + (size_t)k0 * sizeof(T);   // same shape as y
  const unsigned yrow = (unsigned)a.K * (unsigned)sizeof(T);
  u32x4_t rres[RPRE ? TP : 1][RPRE ? RQ : 1];
  auto out_off = [&](const Geo& q, int p) {
    const int h = q.h0 + wq * 4 + p, w = q.w0 + li;
    return (unsigned)((q.n0 * a.H + h) * a.W + w) * yrow + (unsigned)(g * 8) * (unsigned)sizeof(T);      // + q * 64 bytes for the second run
  };
  auto load_res = [&](const Geo& q) {
    if constexpr (RPRE) {
#pragma unroll
      for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int qq = 0; qq < RQ; ++qq) rres[p][qq] = ld16(rg + out_off(q, p) + qq * 64);
    }
  };

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t A[2][TK], B[2][TP];
  auto frags = [&](int buf, int step) {
    const int tap = step >> 1, kk = step & 1;
    const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
    for (int t = 0; t < TK; ++t) A[buf][t] = ld16(s_w + Ab[kk] + tap * WBUF + t * 2048);
#pragma unroll
    for (int p = 0; p < TP; ++p) B[buf][p] = ld16(s_halo + Bb[s][kk] + (p + r) * (PITCH * 128));
  };

  // stage st of a group = (its pair st / nslab, slab st % nslab); the halo requested at the top of M(st) is stage st + 1's
  auto stage_geo = [&](int st, bool& live) { return geom(tile_of(st / nslab, live)); };

  // ---- prologue: first bank landed, constants visible; the first halo staged
  bool live = false, live_n = false;
  Geo cur = stage_geo(0, live);
  load_halo(cur, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the DMA'd bank (and this halo)
  SSLCR_PR_BAR();
  xform_store(0);
  SSLCR_PR_BAR();
  if (grp == 1) { SSLCR_PR_BAR_BARE(); SSLCR_PR_BAR_BARE(); SSLCR_PR_BAR_BARE(); }     // group 1 idles through group 0's first M phase

#ifdef SSLCR_PR_PROF
  unsigned long long pr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  int slab = 0;
  for (int st = 0; st < nst; ++st) {
    const bool last = slab + 1 == nslab;               // the output stage follows this M phase
    const int nslab_i = last ? 0 : slab + 1;           // slab of stage st + 1
    const bool more = st + 1 < nst;
    // ================================================================ M: 288 MFMAs per wave + the bank refill
    PR_T(t0);
    // taps 6-8 of THIS stage: group 1 left the previous stage's M phase at the barrier just passed
    if (grp == 0 && st > 0) dma_taps(6, slab);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);         // the halo loads below stay BEHIND the DMA in issue order (the counted wait relies on it)
    // next stage's halo: a whole M phase to land (younger than the DMA above: the counted wait below does not cover it)
    Geo nxt = cur;
    live_n = live;
    if (last) nxt = stage_geo(more ? st + 1 : st, live_n);
    load_halo(nxt, nslab_i);
    frags(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      if (i < 17) frags((i + 1) & 1, i + 1);
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const bf16x8_t av = __builtin_bit_cast(bf16x8_t, A[i & 1][t]), bv = __builtin_bit_cast(bf16x8_t, B[i & 1][p]);
          acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[t][p], 0, 0, 0);
        }
      if (i < 17) {
        // the next step's eight fragment reads are spread between this step's sixteen MFMAs instead of all being issued first
#pragma unroll
        for (int q = 0; q < TK + TP; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (TK * TP) / (TK + TP), 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 5) {
        // group 0: its taps 6-8 are in (only the NLD halo loads issued after them may still be out) -- published by this barrier,
        // one interval before the step that prefetches tap 6
        if (grp == 0 && st > 0) __builtin_amdgcn_s_waitcnt(0x0f70 | NLD);
        SSLCR_PR_BAR_BARE();
        if (grp == 1 && more) dma_taps(0, nslab_i);    // group 1 is the last reader of taps 0-2 of this stage
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 11) {
        SSLCR_PR_BAR_BARE();
        if (grp == 1 && more) dma_taps(3, nslab_i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (grp == 1) __builtin_amdgcn_s_waitcnt(0x0f70);  // its six taps of the next bank (and its halo) have landed
    PR_T(t1);
    SSLCR_PR_BAR_BARE();                       // every wave of this group is done with its halo buffer; group 1: bank published
    PR_T(t2);
    PR_ACC(0, t1 - t0); PR_ACC(1, t2 - t1);

    // ================================================================ W (three barrier intervals, like M)
    // (0) the residual / mask operand of the tile just finished is requested in this phase: its round trip sits under the partner's MFMAs
    // (1) next stage's halo: landed during M -> transform -> LDS
    xform_store(nslab_i);
    if (last) load_res(cur);                   // (behind the transform: the 44 halo registers are free again)
    PR_T(ta);
    PR_ACC(4, ta - t2);
    SSLCR_PR_BAR_BARE();
    // (2) output stage of the tile just finished
    if (last) {
      const int kb = g * 8;                      // this lane's channels within the block: kb + q * 32 + e
      float s1[4 * TK], s2[4 * TK];
      if constexpr (mk) {
        // g = y * (scale*x + shift > 0) ; partial sums of g and g*(x - mean) over this wave's 64 pixels
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
          float msc[EPC], msh[EPC], mmu[EPC], a1[EPC], a2[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            msc[e] = s_bias[kb + q * 32 + e]; msh[e] = s_msh[kb + q * 32 + e]; mmu[e] = s_mmu[kb + q * 32 + e];
            a1[e] = 0.f; a2[e] = 0.f;
          }
#pragma unroll
          for (int p = 0; p < TP; ++p) {
            float xr[EPC], vq[EPC];
            Elem<T>::unpack(rres[RPRE ? p : 0][RPRE ? q : 0], xr);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const int idx = q * EPC + e;
              const float y = acc[idx >> 2][p][idx & 3];
              const float gv = fmaf(xr[e], msc[e], msh[e]) > 0.f ? y : 0.f;
              vq[e] = gv;
              a1[e] += gv;
              a2[e] = fmaf(gv, xr[e] - mmu[e], a2[e]);
            }
            if (live) st16(yg + out_off(cur, p) + q * 64, PackH<T>::run(vq));      // (uniform)
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) { s1[q * EPC + e] = a1[e]; s2[q * EPC + e] = a2[e]; }
        }
      } else {
        if (live) {
          if (OP == 0 && !a.bias && !a.relu) {     // train-mode forward: pack and store
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
              for (int q = 0; q < RQ; ++q) {
                float vq[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) vq[e] = acc[(q * EPC + e) >> 2][p][e & 3];
                st16(yg + out_off(cur, p) + q * 64, PackH<T>::run(vq));
              }
          } else {
            float bias[4 * TK];
#pragma unroll
            for (int j = 0; j < 4 * TK; ++j) bias[j] = s_bias[kb + (j >> 3) * 32 + (j & 7)];
            const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
              for (int q = 0; q < RQ; ++q) {
                float vq[EPC], rr[EPC];
                if constexpr (OP == 1) Elem<T>::unpack(rres[RPRE ? p : 0][RPRE ? q : 0], rr);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                  float v = acc[(q * EPC + e) >> 2][p][e & 3] + bias[q * EPC + e];
                  if constexpr (OP == 1) v += rr[e];
                  vq[e] = __builtin_amdgcn_fmed3f(v, lo, __builtin_inff());
                }
                st16(yg + out_off(cur, p) + q * 64, PackH<T>::run(vq));
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
      PR_T(tc);
      PR_ACC(6, tc - ta);
      SSLCR_PR_BAR_BARE();
      if (a.stats) {
        // sums over the 16 pixel columns (the lanes of a DPP row): quad q of the row ends up with values 4q..4q+3 in s[0..3]
        // (row16_fold16: 32 DPP adds per 16 values); value idx = t * 4 + j is channel kb + (t >> 1) * 32 + (t & 1) * 4 + j, so quad
        // q = t owns one 16-byte piece.  Its first lane adds it into the group's running sums: every (wave row, kout) entry has
        // exactly ONE writer lane in the workgroup -- plain 16-byte reads and writes, a fixed order
        row16_fold16(s1);
        row16_fold16(s2);
        if (live && (li & 3) == 0) {
          const int t = li >> 2;
          float* sp = s_stat + (wq * 2) * BKO + kb + (t >> 1) * 32 + (t & 1) * 4;
#pragma unroll
          for (int h = 0; h < 2; ++h) {                      // h = 0: sums, 1: sums of squares (or of g * (x - mean))
            const float* sv = h ? s2 : s1;
            f32x4_t* slot = reinterpret_cast<f32x4_t*>(sp + h * BKO);
            f32x4_t v = *slot;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += sv[e];
            *slot = v;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else {
      SSLCR_PR_BAR_BARE();
    }
    PR_T(tb);
    PR_ACC(5, tb - ta);
    // (3) advance
    cur = nxt;
    live = live_n;
    slab = nslab_i;
    PR_T(t3);
    if (!(grp == 1 && st == nst - 1)) SSLCR_PR_BAR();     // this group's new halo is visible to it
    PR_T(t4);
    PR_ACC(2, t3 - t2); PR_ACC(3, t4 - t3);
  }
#ifdef SSLCR_PR_PROF
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) g_pr_prof[wave][i] = pr_t[i];
#endif
  // group 0 idles through group 1's last W phase (three barriers; the last one closes that phase: both groups' sums are final)
  if (grp == 0) { SSLCR_PR_BAR_BARE(); SSLCR_PR_BAR_BARE(); }
  SSLCR_PR_BAR();
  if (a.stats) {
    // four partial rows per workgroup, group 0 + group 1 in that order (deterministic); this workgroup holds ONE kout block: the
    // other blocks' columns of its rows are zero.  Rows of the buffer that no workgroup of this grid owns are zeroed too (the
    // caller sized it with conv_h16_rows before it knew which kernel serves the shape)
    for (int i = tid; i < 8 * a.K; i += 512) {
      const int rw = i / a.K, c = i - rw * a.K;            // rw = wave row * 2 + (0: sum, 1: sumsq)
      const int cl = c - k0;
      float v = 0.f;
      if (cl >= 0 && cl < 64) v = s_stat0[rw * 64 + cl] + s_stat0[512 + rw * 64 + cl];
      a.stats[((size_t)(rb * 4 + (rw >> 1)) * 2 + (rw & 1)) * a.K + c] = v;
    }
    for (int eb = lb + G; eb < hgs; eb += G)
      for (int i = tid; i < 8 * a.K; i += 512) a.stats[(size_t)(seg * hgs + eb) * 8 * a.K + i] = 0.f;
  }
}

// workgroups per segment: a multiple of the kout-block count, at most what conv3x3_h16 would launch per segment (hgs: its rows size
// the caller's statistics buffer) -- which is at most one per CU over all segments
static int ppr_grid_seg(const ConvArgs& a, int hgs) {
  const int kbn = a.K / 64;
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int tiles = (a.N / nseg) * (a.H / 16) * (a.W / 16);
  const int items = ((tiles + 1) / 2) * kbn;
  const int g = items < hgs ? items : hgs;
  return g / kbn * kbn;
}

// bf16, C and K multiples of 64 with a power-of-two kout-block count, 16x16-tileable maps, every operand combination the h16 kernel
// serves for these shapes; SSLCR_PPR=0 keeps conv3x3_h16 for same-box A/B runs.  h16_grid: the workgroups conv3x3_h16 would launch
bool conv_ppr_ok(int dtype, const ConvArgs& a, int h16_grid) {
  static const bool on = [] { const char* e = getenv("SSLCR_PPR"); return !e || atoi(e) != 0; }();
  if (!on || dtype != DT_BF16) return false;
  if (a.C % 64 != 0 || a.K % 64 != 0 || a.C > 256 || (a.C == 64 && a.K == 64)) return false;
  const int kbn = a.K / 64;
  if ((kbn & (kbn - 1)) != 0) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.transposed || a.H % 16 != 0 || a.W % 16 != 0) return false;
  if (a.in_scale && (a.residual || a.mask_x)) return false;
  if ((size_t)a.N * a.H * a.W * (a.C > a.K ? a.C : a.K) * 2 >= ((size_t)1 << 32)) return false;     // 32-bit byte offsets
  if (a.W > 2048) return false;                          // halo roles pack a pixel offset of up to 17 W + 17 into 16 bits
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  if (a.seg_images > 0 && (a.N % a.seg_images != 0 || h16_grid % nseg != 0)) return false;
  return ppr_grid_seg(a, h16_grid / nseg) >= kbn;
}

hipError_t launch_conv_ppr(const ConvArgs& a, int h16_grid, hipStream_t st) {
  const size_t lds = 9 * 64 * 128 + 2 * 18 * 18 * 128 + (192 + 1024 + 2 * (size_t)a.C) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    const void* ks[4] = {reinterpret_cast<const void*>(conv3x3_ppr_kernel<false, 0>), reinterpret_cast<const void*>(conv3x3_ppr_kernel<false, 1>),
                         reinterpret_cast<const void*>(conv3x3_ppr_kernel<false, 2>), reinterpret_cast<const void*>(conv3x3_ppr_kernel<true, 0>)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    attr_done = true;
  }
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int tiles = (a.N / nseg) * (a.H / 16) * (a.W / 16);       // per segment
  const int hgs = h16_grid / nseg;
  const int grid = ppr_grid_seg(a, hgs) * nseg;
  const int kshift = __builtin_ctz(a.K / 64);
  if (a.mask_x && !a.stats) return hipErrorInvalidValue;
  if (a.in_scale) hipLaunchKernelGGL((conv3x3_ppr_kernel<true, 0>), dim3(grid), dim3(512), lds, st, a, tiles, hgs, kshift);
  else if (a.mask_x) hipLaunchKernelGGL((conv3x3_ppr_kernel<false, 2>), dim3(grid), dim3(512), lds, st, a, tiles, hgs, kshift);
  else if (a.residual) hipLaunchKernelGGL((conv3x3_ppr_kernel<false, 1>), dim3(grid), dim3(512), lds, st, a, tiles, hgs, kshift);
  else hipLaunchKernelGGL((conv3x3_ppr_kernel<false, 0>), dim3(grid), dim3(512), lds, st, a, tiles, hgs, kshift);
  return hipGetLastError();
}

const char* conv_ppr_name(const ConvArgs& a) {
  if (a.in_scale) return "sslcr::conv3x3_ppr_kernel<true, 0>";
  return a.mask_x ? "sslcr::conv3x3_ppr_kernel<false, 2>" : (a.residual ? "sslcr::conv3x3_ppr_kernel<false, 1>" : "sslcr::conv3x3_ppr_kernel<false, 0>");
}

}  // namespace sslcr

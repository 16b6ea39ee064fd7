// 3x3 / stride 1 / pad 1 NHWC convolution, 64 -> 64 channels in bf16 on 16x16-pixel tiles (ResNet18 layer1: forward, eval-fused
// forward and both dgrads; torchvision BasicBlock convs reached from models/net.py:32,77) as a PING-PONG of two wave groups.
//
// Why.  The resident-filter form of conv_h16.hip runs this shape at 0.84 PF/s with the matrix pipes 39 % busy, and it is not memory:
// the time per image is the same for a batch that fits the 256 MB Infinity Cache (N = 128: 0.34 us) and for one that streams from HBM
// (N = 1280: 0.32 us).  Its ablation (DESIGN, round 1) showed the parts of a tile -- 144 MFMAs per wave, halo load / transform /
// LDS write, output stage, BatchNorm sums -- ADDING UP instead of hiding under each other: all eight waves of the workgroup walk
// through the same phase at the same time, so while they transform a halo or store a tile the matrix pipe idles, and with two waves
// per SIMD there is no third wave to fill the gap.
//
// Here the workgroup's eight waves are two groups of four (waves w and w+4 share a SIMD, so each group has one wave on every
// SIMD).  A group owns a whole 16x16-pixel x 64-kout tile (wave = 4 rows x 16 pixels x 64 kouts: 16 MFMAs per step, half the
// fragment reads per MFMA of the 8-wave split) and alternates two phases separated by workgroup barriers:
//     M  18 steps (9 taps x 2 K-halves) of MFMAs out of its own halo buffer and the shared resident filter bank -- no VALU, no
//        global memory instruction, no LDS write;
//     W  everything else: output stage of the tile just finished (bias / residual / ReLU or the BatchNorm-backward front end,
//        pack, stores, BatchNorm partial sums), BatchNorm+ReLU transform of the NEXT tile's halo (requested one phase earlier,
//        landed during M) and its six..eleven ds_write_b128, request of the halo after that and of the next residual.
// The groups run half a period apart: while one is in M the other is in W, so on every SIMD one wave feeds the matrix pipe while
// its partner issues VALU / VMEM / LDS-write work (separate pipes, MI355X_MICROARCH.md "Two waves per SIMD").  A tile costs
// max(M, W) instead of M + W.  Every barrier is the whole workgroup's s_barrier; both groups execute the same number of them.
//
// LDS: filter bank 9 x 64 x 128 B = 72 KB (fragment order, XOR-swizzled), two 18x18-pixel halos at an 18-pixel pitch (the swizzle
// key is the halo COLUMN & 7, so the bank pattern does not depend on the pitch) = 81 KB, scale/shift/bias/statistics 5 KB.
#include <stdlib.h>

#include "kernels.hpp"

namespace sslcr {

typedef const __attribute__((address_space(1))) void* gptr_pp;
typedef __attribute__((address_space(3))) void* lptr_pp;

// LDS visibility + rendezvous without the workgroup-scope fence of __syncthreads(): with global loads and stores in flight the
// fence becomes s_waitcnt vmcnt(0), i.e. "wait for the halo of the next tile and for the stores of the last one" at every barrier
#define SSLCR_PP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// phase timing for tools/microbench/pp64_phase_bench.hip (-DSSLCR_PP_PROF): per wave of workgroup 0, shader cycles spent in M, at the
// barrier behind M, in W, at the barrier behind W
#ifdef SSLCR_PP_PROF
__device__ unsigned long long g_pp_prof[8][8];
#define PP_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define PP_ACC(i, d) pp_t[i] += (d)
#else
#define PP_T(v)
#define PP_ACC(i, d)
#endif

// Which kout a lane's accumulator element holds: MFMA tile t, accumulator row group g (= lane >> 4), element j.  A lane's 16 kouts
// are TWO runs of 8 consecutive channels, 32 channels apart, so that the four row groups of one pixel write one contiguous 64-byte
// segment per store instruction (the layout of conv_h16 -- 16 consecutive kouts per lane -- makes every 16-byte store half of a
// 32-byte stride: twice the write requests, and the output stage is bound by exactly their issue rate).
#define PP_CH(t, g, j) ((((t) >> 1) * 32) + ((g) * 8) + (((t) & 1) * 4) + (j))
// LDS row of the resident filter bank -> kout row: fragment row 4g + j of tile t feeds accumulator (t, g, j)
__device__ __forceinline__ int pp_row_kout(int rr) {
  const int t = rr >> 4, gq = (rr >> 2) & 3, j = rr & 3;
  return PP_CH(t, gq, j);
}

// XF: the producer's BatchNorm(+ReLU) is applied to the input on its way into LDS (a.in_scale != nullptr); never with a residual
// OP: 0 plain output stage (bias / ReLU / statistics), 1 + residual (a.residual), 2 BatchNorm-backward front end (a.mask_x).
// Compile-time, so that each instance carries only its own operand registers through the loop
template <bool XF, int OP>
__global__ __launch_bounds__(512, 2) void conv3x3_pp64_kernel(const ConvArgs a, const int tiles_total, const int rows_total) {
  typedef bf16_t T;
  constexpr int EPC = 8, BKO = 64, TK = 4, TP = 4;
  constexpr int TW = 16, TH = 16, HH = 18, HWD = 18, PITCH = 18, HP = HH * HWD;
  constexpr int GT = 256;                              // threads per group
  constexpr int NLD = (HP * 8 + GT - 1) / GT;          // 11 sixteen-byte halo chunks per thread and tile
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
  float* s_scale = s_f;
  float* s_shift = s_f + 64;
  float* s_bias = s_f + 128;                           // bias, or the BatchNorm scale of the mask_x front end
  float* s_msh = s_f + 192;
  float* s_mmu = s_f + 256;
  float* s_stat = s_f + 320 + grp * 512;               // this group's [4 wave rows][2][64] partial (sum, sumsq)

  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  // segments (sslcr_conv_desc.seg_images): nseg equal groups of workgroups, group s walks the tiles of its own images with its
  // own prologue; tiles_total is then PER SEGMENT and a workgroup's statistics rows belong to one segment (conv_h16.hip)
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int G = gridDim.x / nseg;
  const int seg = nseg > 1 ? (int)blockIdx.x / G : 0, lb = (int)blockIdx.x - seg * G;
  const int seg_n0 = seg * a.seg_images;
  // XCD-aware walk (blocks land on XCD blockIdx % 8): each XCD takes a contiguous run of tiles per round
  const int vb = (G & 7) ? lb : (lb & 7) * (G >> 3) + (lb >> 3);
  const int first0 = 2 * vb;
  if (first0 >= tiles_total) return;
  const int nst = (tiles_total - first0 + 2 * G - 1) / (2 * G);       // stages of group 0 (group 1 may have one live stage less)
  constexpr bool mk = OP == 2;
  const float relu_lo = a.in_relu ? 0.f : -__builtin_inff();

  // ---- resident filter bank: wave w fills rows [8w, 8w+8) of every tap by DMA; the lane picks its SOURCE chunk so that the
  //      linear placement is the fragment-ordered, swizzled tile
  {
    const int rr = wave * 8 + (lane >> 3);
    const int krow = pp_row_kout(rr);
    const int c16 = (lane & 7) ^ (rr & 7);
    const char* wg = reinterpret_cast<const char*>(a.w) + ((size_t)krow * 9 * 64 + c16 * EPC) * sizeof(T);
#pragma unroll
    for (int tt = 0; tt < 9; ++tt)
      __builtin_amdgcn_global_load_lds((gptr_pp)(wg + (size_t)tt * 64 * sizeof(T)), (lptr_pp)(s_w + tt * WBUF + (wave * 8) * 128), 16, 0, 0);
  }
  if (tid < 64) {
    if (XF) { s_scale[tid] = a.in_scale[(size_t)seg * a.seg_stride + tid]; s_shift[tid] = a.in_shift[(size_t)seg * a.seg_stride + tid]; }
    const size_t mo = (size_t)seg * a.seg_stride + tid;        // the mask's BatchNorm is the segment's own
    s_bias[tid] = mk ? a.mask_scale[mo] : (a.bias ? a.bias[tid] : 0.f);
    if (mk) { s_msh[tid] = a.mask_shift[mo]; s_mmu[tid] = a.mask_mean[mo]; }
    else if (a.out_scale) s_msh[tid] = a.out_scale[tid];       // sslcr_conv_desc.out_scale (eval forms: never with the mask) in the mask's array
  }
  const bool osc_on = !mk && !XF && a.out_scale != nullptr;
  for (int i = tid; i < 1024; i += 512) s_f[320 + i] = 0.f;

  // ---- per-thread halo staging roles (group-local), fixed for the whole walk
  const int chunk = gtid & 7;
  // one register per entry: bits 0-15 = pixel offset from the tile origin, biased by W + 1; bits 16-27 = LDS byte offset / 16
  // (the M phase holds 64 accumulators, 64 fragment registers and the 44 of the halo in flight: every register counts)
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
  // this group's k-th tile; past the end it re-walks the last tile of the tensor with its stores suppressed (same barrier count)
  auto tile_of = [&](int k, bool& live) {
    const int t = first0 + grp + k * 2 * G;
    live = t < tiles_total;
    return live ? t : tiles_total - 1;
  };

  u32x4_t hreg[NLD];
  unsigned hin = 0;                            // hreg[i] holds image data (not zero padding)
  auto load_halo = [&](const Geo& q) {         // branch-free: padding entries load the tile origin and are zeroed when staged
    const unsigned long long bad = edge & q.out;
    hin = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      int rl = role[i];
      asm volatile("" : "+v"(rl));               // keep the unpacked form out of the loop-invariant set (it would cost 11 registers)
      const int idx = q.origin + (ok ? (rl & 0xffff) - rel_bias : 0);
      hreg[i] = ld16_nt(xg + (unsigned)idx * (unsigned)(64 * sizeof(T)));       // 32-bit offsets from a uniform base (tensors < 4 GB)
      hin |= (ok ? 1u : 0u) << i;
    }
  };
  auto xform_store = [&]() {                   // hreg -> what LDS must hold -> this group's halo buffer
    float sc[EPC], sh[EPC];
    if (XF) {                                  // re-read per tile (four ds_read_b128): 16 registers the M phase does not carry
      const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(s_scale + chunk * EPC), c1 = *reinterpret_cast<const f32x4_t*>(s_scale + chunk * EPC + 4);
      const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(s_shift + chunk * EPC), h1 = *reinterpret_cast<const f32x4_t*>(s_shift + chunk * EPC + 4);
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
        for (int e = 0; e < EPC; ++e) f[e] = __builtin_amdgcn_fmed3f(fmaf(f[e], sc[e], sh[e]), relu_lo, __builtin_inff());   // one clamp, no canonicalising max pair
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

  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = OP == 0 ? nullptr : reinterpret_cast<const char*>(OP == 2 ? a.mask_x : a.residual);   // same shape
  u32x4_t rres[RPRE ? TP : 1][RPRE ? RQ : 1];
  auto out_off = [&](const Geo& q, int p) {
    const int h = q.h0 + wq * 4 + p, w = q.w0 + li;
    return (unsigned)(((q.n0 * a.H + h) * a.W + w) * 64 + g * 8) * (unsigned)sizeof(T);      // + q * 64 bytes for the second run
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
  u32x4_t A[2][TK], B[2][TP];
  auto frags = [&](int buf, int step) {
    const int tap = step >> 1, kk = step & 1;
    const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
    for (int t = 0; t < TK; ++t) A[buf][t] = ld16(s_w + Ab[kk] + tap * WBUF + t * 2048);
#pragma unroll
    for (int p = 0; p < TP; ++p) B[buf][p] = ld16(s_halo + Bb[s][kk] + (p + r) * (PITCH * 128));
  };

  // ---- prologue: filter bank landed, constants visible; first halo staged; second halo (and the first residual) requested
  bool live = false, live_n = false;
  Geo cur = geom(tile_of(0, live));
  load_halo(cur);
  __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the DMA'd filter bank (and this halo)
  SSLCR_PP_BAR();
  xform_store();
  Geo nxt = geom(tile_of(1, live_n));
  load_halo(nxt);
  SSLCR_PP_BAR();
  if (grp == 1) SSLCR_PP_BAR();                // group 1 idles through group 0's first M phase: from here on the groups alternate

#ifdef SSLCR_PP_PROF
  unsigned long long pp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int st = 0; st < nst; ++st) {
    // ================================================================ M: 288 MFMAs per wave, nothing else
    PP_T(t0);
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
          if (i == 0) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[t][p], 0, 0, 0);
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
    }
    PP_T(t1);
    SSLCR_PP_BAR();                            // every wave of this group is done with its halo buffer
    PP_T(t2);
    PP_ACC(0, t1 - t0); PP_ACC(1, t2 - t1);

    // ================================================================ W
    // (0) the residual / mask operand of the tile just finished is requested HERE, not a phase ahead: its round trip sits under
    //     the partner group's MFMAs like the rest of W, and the M phase does not carry its 32 registers
    load_res(cur);
    // (1) next tile's halo: landed during M (vmcnt counts in order: waiting for it does not wait for the younger residual
    //     loads) -> transform -> LDS
    xform_store();
    PP_T(ta);
    PP_ACC(4, ta - t2);
    // (2) output stage of the tile just finished
    {
      const int kb = g * 8;                      // this lane's channels: kb + q * 32 + e (q = 16-byte chunk, e = 0..7)
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
        // The phase is bound by its VALU instruction COUNT (a wave issues one per ~4 cycles and nothing hides them, see the
        // header), so the two uniform cases are separate straight-line bodies: the train-mode forward has neither bias nor ReLU
        // nor residual -- pack and store -- and the others clamp with ONE v_med3 per element instead of a branch per chunk.
        if (live) {
          if (OP == 0 && !a.bias && !a.relu) {
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
            if (osc_on) {                          // eval-mode BatchNorm scale kept out of the filters: acc * scale in place (uniform)
#pragma unroll
              for (int j = 0; j < 4 * TK; ++j) {
                const float sj = s_msh[kb + (j >> 3) * 32 + (j & 7)];
#pragma unroll
                for (int p = 0; p < TP; ++p) acc[j >> 2][p][j & 3] *= sj;
              }
            }
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
      if (a.stats) {
        // s1 / s2: this lane's sums over its four pixel rows; now over the 16 pixel columns (the lanes of a DPP row).  row16_fold16
        // (32 DPP adds per 16 values; row16_sum4 took 64) leaves quad q of the row with values 4q..4q+3 in s[0..3]: value idx = t * 4 + j
        // is channel kb + (t >> 1) * 32 + (t & 1) * 4 + j, so quad q = t owns one 16-byte piece of the running sums.  Every
        // (wave row, kout) entry has exactly ONE writer lane in the whole workgroup: plain 16-byte reads and writes, a fixed order
        // (32 LDS float atomics per tile cost ~1400 cycles, each holds the LDS pipe ~40)
        row16_fold16(s1);
        row16_fold16(s2);
        if (live && (li & 3) == 0) {
          const int t = li >> 2;
          float* sp = s_stat + (wq * 2) * BKO + kb + (t >> 1) * 32 + (t & 1) * 4;
#pragma unroll
          for (int h = 0; h < 2; ++h) {                    // h = 0: sums, 1: sums of squares (or of g * (x - mean))
            const float* sv = h ? s2 : s1;
            f32x4_t* slot = reinterpret_cast<f32x4_t*>(sp + h * BKO);
            f32x4_t v = *slot;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += sv[e];
            *slot = v;
          }
        }
      }
    }
    PP_T(tb);
    PP_ACC(5, tb - ta);
    // (3) advance; request the halo after next: a whole M phase to land
    cur = nxt;
    live = live_n;
    nxt = geom(tile_of(st + 2, live_n));
    load_halo(nxt);
    PP_T(t3);
    if (!(grp == 1 && st == nst - 1)) SSLCR_PP_BAR();     // this group's new halo is visible to it
    PP_T(t4);
    PP_ACC(2, t3 - t2); PP_ACC(3, t4 - t3);
  }
#ifdef SSLCR_PP_PROF
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) g_pp_prof[wave][i] = pp_t[i];
#endif
  // group 0 idles through group 1's last W phase: that phase's closing barrier is this one, after which both groups' sums are final
  SSLCR_PP_BAR();
  if (a.stats) {
    // four partial rows per workgroup, group 0 + group 1 in that order (deterministic); the row count is the one conv3x3_h16
    // would write for this shape (conv_h16_rows: the caller sized the buffer before it knew the dtype), so rows this grid does
    // not own are zeroed
    const int rw = tid >> 6, c = tid & 63;                 // rw = wave row * 2 + (0: sum, 1: sumsq)
    a.stats[((size_t)((int)blockIdx.x * 4 + (rw >> 1)) * 2 + (rw & 1)) * 64 + c] = s_f[320 + tid] + s_f[320 + 512 + tid];
    const int eb = (int)blockIdx.x + (int)gridDim.x;
    if (eb * 4 < rows_total) a.stats[((size_t)(eb * 4 + (rw >> 1)) * 2 + (rw & 1)) * 64 + c] = 0.f;
  }
}

static int pp64_grid(const ConvArgs& a) {
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int tiles = (a.N / nseg) * (a.H / 16) * (a.W / 16);       // per segment
  const int pairs = (tiles + 1) / 2, per = device_cus() / nseg;
  return (pairs < per ? pairs : per) * nseg;
}

// bf16 64 -> 64 on 16x16-tileable maps, every operand combination conv3x3_h16's resident-filter form serves (its caller has
// already checked conv_h16_ok); SSLCR_PP64=0 keeps the old kernel for same-box A/B runs
bool conv_pp64_ok(int dtype, const ConvArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_PP64"); return !e || atoi(e) != 0; }();
  if ((size_t)a.N * a.H * a.W * 64 * 2 >= ((size_t)1 << 32)) return false;       // the kernel addresses with 32-bit byte offsets
  if (a.W > 2048) return false;                                                  // halo roles pack a pixel offset of up to 17 W + 17 into 16 bits
  return on && dtype == DT_BF16 && a.C == 64 && a.K == 64 && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && !a.transposed &&
         a.H % 16 == 0 && a.W % 16 == 0 && !(a.in_scale && (a.residual || a.mask_x)) &&
         !(a.out_scale && (a.in_scale || a.mask_x || a.stats || !a.bias));
}
// rows of a.stats: what conv3x3_h16 would write for this shape (one 64-kout block; the caller sized the buffer before it knew the
// dtype); with segments -- a bf16-only form -- exactly this grid's rows, so that segment s owns rows [s, s + 1) * rows / nseg
int conv_pp64_rows(const ConvArgs& a) {
  if (a.seg_images > 0) return pp64_grid(a) * 4;
  const int tiles = a.N * (a.H / 16) * (a.W / 16);
  return (tiles < device_cus() ? tiles : device_cus()) * 4;
}

hipError_t launch_conv_pp64(const ConvArgs& a, hipStream_t st) {
  constexpr size_t lds = 9 * 64 * 128 + 2 * 18 * 18 * 128 + (320 + 1024) * sizeof(float);
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    const void* ks[4] = {reinterpret_cast<const void*>(conv3x3_pp64_kernel<false, 0>), reinterpret_cast<const void*>(conv3x3_pp64_kernel<false, 1>),
                         reinterpret_cast<const void*>(conv3x3_pp64_kernel<false, 2>), reinterpret_cast<const void*>(conv3x3_pp64_kernel<true, 0>)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    attr_done = true;
  }
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int tiles = (a.N / nseg) * (a.H / 16) * (a.W / 16);       // per segment
  const int grid = pp64_grid(a);
  if (a.mask_x && !a.stats) return hipErrorInvalidValue;
  const int rows = conv_pp64_rows(a);
  if (a.in_scale) hipLaunchKernelGGL((conv3x3_pp64_kernel<true, 0>), dim3(grid), dim3(512), lds, st, a, tiles, rows);
  else if (a.mask_x) hipLaunchKernelGGL((conv3x3_pp64_kernel<false, 2>), dim3(grid), dim3(512), lds, st, a, tiles, rows);
  else if (a.residual) hipLaunchKernelGGL((conv3x3_pp64_kernel<false, 1>), dim3(grid), dim3(512), lds, st, a, tiles, rows);
  else hipLaunchKernelGGL((conv3x3_pp64_kernel<false, 0>), dim3(grid), dim3(512), lds, st, a, tiles, rows);
  return hipGetLastError();
}

const char* conv_pp64_name(const ConvArgs& a) {
  if (a.in_scale) return "sslcr::conv3x3_pp64_kernel<true, 0>";
  return a.mask_x ? "sslcr::conv3x3_pp64_kernel<false, 2>" : (a.residual ? "sslcr::conv3x3_pp64_kernel<false, 1>" : "sslcr::conv3x3_pp64_kernel<false, 0>");
}

}  // namespace sslcr

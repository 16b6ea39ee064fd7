// fp8 (OCP e4m3) forward path of the 3x3 / stride 1 / pad 1 block convolutions with C % 128 == 0 and K % 128 == 0 (ResNet18
// layers 2-4: 9 of the 20 convs, 57 % of the forward FLOPs) on v_mfma_scale_f32_16x16x128_f8f6f4 -- the only gfx950 matrix
// instruction that runs fp8 at twice the bf16 rate (the non-scaled 16x16x32 fp8 form runs at the bf16 rate).  BASELINE config 5
// (eval_Camelyon_SSL_CR.py:33-157 "fp8 MFMA conv path").
//
// What is fp8 and what is not.  Activations stay bf16 in HBM (the backward pass -- bf16 dgrad / wgrad from the saved tensors --
// is unchanged); they are quantised on the way into LDS: the 18x18 halo of a 128-channel slab is loaded as bf16, the producer's
// BatchNorm scale/shift + ReLU applied in fp32 (train mode; eval mode reads activated tensors), multiplied by the per-tensor
// x_scale, clamped to +-448 (v_cvt_pk_fp8_f32 does not saturate: 480 -> NaN, tools/microbench/fp8_probe.hip) and converted
// round-to-nearest-even -- 128 bytes per halo pixel instead of 256.  Weights are e4m3 shadow packs [K][9][C] with a
// power-of-two scale per output channel (pack_fp8_kernel; written next to the bf16 packs after every update) and the MX block
// scales of the instruction are all 2^0.  Accumulation is fp32; the epilogue multiplies by 1 / (w_scale[k] * x_scale) and then
// is the bf16 kernels' epilogue: bias / residual / ReLU, bf16 store, per-channel (sum, sum^2) partial rows from the fp32 values.
//
// Shape.  The 256-pixel LDS-halo form of conv_halo256.hip (8 waves, wave = 64 px x 64 kout, weights = MFMA A operand, kout rows
// permuted so that a lane owns 16 consecutive output channels, 3-tap weight ring halves, XOR-swizzled 16-byte units) with
// K = 128 per instruction: lane (i, g) feeds 32 channels of kout row / pixel i as two ds_read_b128 (operand layout verified by
// tools/microbench/fp8_probe.hip) -- the 16-byte units g and g + 4 of the 128-byte slab row, NOT 2g and 2g + 1: any assignment
// works as long as A and B agree, and with adjacent units per lane group every fragment read was a 2-way bank conflict (lane
// groups g and g + 1 of one LDS pass then meet in the same unit two pixels apart, i.e. in the same half of the 64 banks; with
// units g, g + 4 they meet one pixel apart, in different halves -- SQ_LDS_BANK_CONFLICT 43 % -> see profiles/r02_fp8_pmc.md).  Per tap a wave issues 16 MFMAs (512 matrix-pipe cycles) against 16 KB of
// fragment reads -- the same ratio as the bf16 kernel at twice the channels.  Workgroups are PERSISTENT over tiles: the next
// (tile, slab) stage's halo is requested before the nine taps of the current one and converted + written after them, so that
// also single-slab layers (C = 128) overlap the HBM round trip with matrix work.
// Roof: with bf16 activations in HBM the layer2 shape (C = K = 128) moves 512 B per pixel for 295 kFLOP -> 576 FLOP/B, i.e.
// about 3.4 PF at 6 TB/s; layers 3 and 4 are matrix-pipe bound (5 PF).
#include "kernels.hpp"

namespace sslcr {

typedef int v8i_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t cvt4_fp8(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}

// XF = the producer's BatchNorm scale/shift (+ ReLU) is applied on the load path (train mode).  Compile-time: a run-time branch
// inside the tap loop splits it into basic blocks, and the MFMAs were then sunk below the staging code with every fragment
// spilled in between.
template <int TW, bool XF>
__global__ __launch_bounds__(512) void conv3x3_fp8_kernel(const ConvArgs a, const Fp8Args q, const int ntiles) {
  constexpr int NT = 512, BKO = 128, TK = 4, TP = 4;
  constexpr int TH = TW;
  constexpr int NI = 256 / (TH * TW);
  constexpr int HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;               // 324 or 400 halo pixels
  constexpr int NLD = (HP * 16 + NT - 1) / NT;    // 16-byte (8 x bf16) staging loads per thread and stage: 11 or 13
  constexpr int HD = 1;                           // taps between the request of a halo batch and its conversion
  constexpr int BATCH = (NLD + 8 - HD) / (9 - HD); // ... issued in 9 - HD batches under taps 0 .., converted + written HD taps later
  constexpr int WLD = BKO * 8 / NT;               // 16-byte weight loads per thread and tap: 2
  constexpr int HBUF = ((HP * 128 + 1023) / 1024) * 1024, WBUF = BKO * 128;
  // LDS: the fp8 halo is DOUBLE-buffered (the next stage is written while this one is read, and the staging registers live
  // for two taps instead of nine: with one buffer and 11-13 chunks held across the tap loop the kernel spilled 800+ registers
  // at the 256-register budget of two waves per SIMD); weights: a ring of three single-tap slots, one barrier per tap.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem + 2 * HBUF;                    // [3][BKO][128 B]
  float* s_scale = reinterpret_cast<float*>(smem + 2 * HBUF + 3 * WBUF);
  float* s_shift = s_scale + a.C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave & 3, wk = wave >> 2;
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  const int k0 = blockIdx.y * BKO;
  constexpr bool xform = XF;
  const float xs = q.x_scale_dev ? *q.x_scale_dev : q.x_scale;
  float amax = 0.f;                                // of the transformed activations this thread converts, in scaled units
  if (xform)
    for (int c = tid; c < a.C; c += NT) { s_scale[c] = a.in_scale[c] * xs; s_shift[c] = a.in_shift[c] * xs; }

  const int chunk = tid & 15;                     // 8 bf16 channels of the slab: fp8 bytes 8*chunk .. 8*chunk+7 of the pixel
  const int unit = chunk >> 1, half = chunk & 1;
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* wg = reinterpret_cast<const char*>(q.w8);
  const int nslabs = a.C / 128;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_tiles <= 0) return;
  const int nst = my_tiles * nslabs;

  auto tile_coords = [&](int t, int& n0, int& h0, int& w0) {
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    n0 = (t / tiles_h) * NI; h0 = th_i * TH; w0 = tw_i * TW;
  };
  // source pixel of halo chunk i of this thread for the tile at (n0, h0, w0): -1 = zero padding, -2 = past the halo
  // The tap loop below is kept free of control flow (with branches around the prefetches the register allocator spilled the
  // accumulators): chunk indices past the halo are clamped to its last pixel (two threads then write the same bytes), padding
  // pixels load pixel 0 and are zeroed by a select, and the last stage of a workgroup re-stages its own halo.
  auto hp_of = [&](int i) {
    int t = tid;
    asm volatile("" : "+v"(t));                   // derive per use from an opaque copy of the thread index: hoisted out of the stage
    const int hp = (t >> 4) + (NT / 16) * i;      // loop, the per-chunk pixel decompositions cost ~40 registers (then spilled, and a
    return hp < HP ? hp : HP - 1;                 // scratch reload is a vmcnt(0) that also waits for every HBM load in flight)
  };
  auto src_of = [&](int hp, int n0, int h0, int w0) {
    const int ni = hp / (HH * HWD), rem = hp - ni * (HH * HWD);
    const int hr = rem / HWD, hc = rem - hr * HWD;
    const int h = h0 - 1 + hr, w = w0 - 1 + hc;
    const int lin = ((n0 + ni) * a.H + h) * a.W + w;                // computed unconditionally, selected: no exec-masked block
    const bool in = ((h | w) >= 0) & (h < a.H) & (w < a.W);
    return in ? lin : -1;
  };
  // XOR key of a halo pixel's 16-byte units = its halo COLUMN & 7: the key of a fragment read then depends on the lane and the
  // filter column only, so the 9 taps x 4 pixel groups address LDS as (per-lane base) + immediate (conflict-free: a pixel is one
  // full 128-byte bank row, the 16 lanes of a fragment read 16 / 2 x 8 consecutive columns)
  auto halo_key = [&](int hp) { return (hp % HWD) & 7; };
  int aff_off = chunk * 8;                        // this thread's 8 channels of the slab being staged (scale / shift read from LDS
  auto set_affine = [&](int slab) { aff_off = slab * 128 + chunk * 8; };     // at use: 16 registers less across the tap loop)
  // global loads go through buffer resources: (uniform descriptor) + 32-bit lane offset + uniform offset -- ONE address register
  // per load.  With flat 64-bit addresses the compiler kept a pointer pair per (tap, row) across the stage loop, spilled them,
  // and every scratch reload is a vmcnt(0) that also drains the HBM loads in flight.  The launcher keeps tensors below 4 GiB.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xg), 0, (int)((size_t)a.N * a.H * a.W * a.C * 2), 0x00020000);
  auto load_chunk = [&](int src, int slab) {
    const int sidx = src < 0 ? 0 : src;
    const uint32_t lane_off = (uint32_t)sidx * (uint32_t)(a.C * 2) + (uint32_t)(chunk * 16);
    return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)lane_off, slab * 256, 0));
  };
  const float lo_clamp = a.in_relu ? 0.f : -448.f;
  auto store_chunk = [&](char* hbuf, int hp, int src, const u32x4_t& v) {
    float f[8];
    Elem<bf16_t>::unpack(v, f);
    if (xform) {
      const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(s_scale + aff_off), c1 = *reinterpret_cast<const f32x4_t*>(s_scale + aff_off + 4);
      const f32x4_t d0 = *reinterpret_cast<const f32x4_t*>(s_shift + aff_off), d1 = *reinterpret_cast<const f32x4_t*>(s_shift + aff_off + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[e] = fmaf(f[e], c0[e], d0[e]); f[4 + e] = fmaf(f[4 + e], c1[e], d1[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= xs;
    }
    {   // amax for delayed scaling, tracked unconditionally (no branch in the tap loop; padding pixels and, with ReLU, the negative
        // side are included: harmless headroom).  The accumulate is an inline v_max_f32: written as fmaxf() the compiler
        // restructured the staging code around the loop-carried value and spilled ~100 registers instead of ~20.
      const float m = fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))),
                            fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
      asm volatile("v_max_f32 %0, %0, %1" : "+v"(amax) : "v"(m));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], lo_clamp, 448.f);     // ReLU and the e4m3 range in one instruction
    u32x2_t o;
    o[0] = cvt4_fp8(f[0], f[1], f[2], f[3]);
    o[1] = cvt4_fp8(f[4], f[5], f[6], f[7]);
    if (src < 0) o = u32x2_t{0u, 0u};              // padding pixels are zero AFTER the producer transform
    *reinterpret_cast<u32x2_t*>(hbuf + hp * 128 + ((unit ^ halo_key(hp)) << 4) + half * 8) = o;
  };
  const int wchunk = tid & 7;
  const uint32_t wlane = (uint32_t)((tid >> 3) * 9) * (uint32_t)a.C + (uint32_t)(wchunk * 16);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wg) + (size_t)k0 * 9 * a.C, 0, 128 * 9 * a.C, 0x00020000);
  auto load_w_to = [&](u32x4_t (&dst)[WLD], int slab, int tap) {
#pragma unroll
    for (int i = 0; i < WLD; ++i)
      dst[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)wlane, (i * (NT / 8) * 9 + tap) * a.C + slab * 128, 0));
  };
  auto store_w_from = [&](const u32x4_t (&src)[WLD], int buf) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = wperm<TK>((tid >> 3) + (NT / 8) * i);
      st16(s_w + buf * WBUF + row * 128 + ((wchunk ^ (row & 7)) << 4), src[i]);
    }
  };

  int pbase[TP];                                  // byte offset of this lane's pixel of group p in the halo, filter tap (0, 0)
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int pg = wp * 4 + p;
    if (TW == 16) pbase[p] = (pg * HWD + li) * 128;
    else pbase[p] = ((pg >> 2) * (HH * HWD) + (2 * (pg & 3) + (li >> 3)) * HWD + (li & 7)) * 128;
  }
  const int lcol = TW == 16 ? li : (li & 7);
  int xlo[3], xhi[3];                             // swizzled unit offsets of this lane's two 16-byte units, per filter column
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) {
    xlo[sx] = (g ^ ((lcol + sx) & 7)) << 4;
    xhi[sx] = ((g + 4) ^ ((lcol + sx) & 7)) << 4;
  }
  // weight rows of this lane: wk * 64 + t * 16 + li -> row & 7 = li & 7 for every t: one base + t * 2048
  const int abase_lo = (wk * 64 + li) * 128 + ((g ^ (li & 7)) << 4);
  const int abase_hi = (wk * 64 + li) * 128 + (((g + 4) ^ (li & 7)) << 4);

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t held[WLD], wq[2][WLD];
  if (xform) __syncthreads();
  {                                                // stage 0: halo buffer 0, ring slots 0 and 1 <- taps 0, 1
    int n0, h0, w0;
    tile_coords((int)blockIdx.x, n0, h0, w0);
    set_affine(0);
#pragma unroll 1
    for (int i = 0; i < NLD; ++i) {
      const int hp = hp_of(i);
      const int src = src_of(hp, n0, h0, w0);
      store_chunk(smem, hp, src, load_chunk(src, 0));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      load_w_to(held, 0, t);
      store_w_from(held, t);
    }
    load_w_to(wq[0], 0, 2);                        // tap 2: written to slot 2 at the end of tap 0
  }
  __syncthreads();

  for (int s = 0; s < nst; ++s) {
    const int ti = s / nslabs, slab = s - ti * nslabs;
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const bool more = s + 1 < nst;
    const int nslab = (slab + 1 == nslabs) ? 0 : slab + 1;
    const char* hcur = smem + (s & 1) * HBUF;
    char* hnext = smem + ((s & 1) ^ 1) * HBUF;
    // next stage: the following slab of this tile, else slab 0 of this workgroup's next tile; the last stage re-stages itself
    const int ntile = more ? (nslab == 0 ? tile + (int)gridDim.x : tile) : tile;
    const int pslab = more ? nslab : slab;
    int nn0, nh0, nw0;
    tile_coords(ntile, nn0, nh0, nw0);
    if (nslabs > 1) set_affine(pslab);              // (this thread's channels of the NEXT slab; single-slab layers keep theirs)
    // the fragment addresses (pixel base + swizzled unit + tap immediate) are summed per read: as stage-loop invariants all 24
    // (pixel group, filter column, lo / hi) sums were kept in registers and spilled
    asm volatile("" : "+v"(pbase[0]), "+v"(pbase[1]), "+v"(pbase[2]), "+v"(pbase[3]));
    u32x4_t hq[HD + 1][BATCH];
    int hsrc[HD + 1][BATCH];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int j = s * 9 + tap;
      // Global loads are issued two taps ahead of their use (vmcnt retires in order: a wait on a young load also waits for every
      // older one, so the weights are requested BEFORE the halo batch of the same tap and the compiler's counted waits leave the
      // younger halo batches in flight).  Weights: tap j sits in ring slot j % 3; tap j+3 is requested now into the register
      // set wq[tap & 1 ^ 1]... -- concretely: the set loaded at tap j-1 (tap j+2's weights) is written at the end of THIS tap
      // to slot (j+2) % 3, the slot of tap j-1, whose readers all passed the barrier that ended tap j-1.
      // (nine taps per stage is odd: at tap 8 the held set is written first and reloaded in place, so that every stage starts
      // with the same register roles)
      u32x4_t (&wnew)[WLD] = wq[tap == 8 ? 0 : ((tap + 1) & 1)];
      u32x4_t (&wold)[WLD] = wq[tap & 1];
      if (tap == 8) store_w_from(wold, (j + 2) % 3);
      if (tap + 3 < 9) load_w_to(wnew, slab, tap + 3); else load_w_to(wnew, pslab, tap + 3 - 9);
      if (tap < 9 - HD) {                          // next stage's halo, batch `tap`: requested now, converted HD taps later
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
          const int i = tap * BATCH + b;
          if (i < NLD) {
            const int src = src_of(hp_of(i), nn0, nh0, nw0);
            hsrc[tap % (HD + 1)][b] = src;
            hq[tap % (HD + 1)][b] = load_chunk(src, pslab);
          }
        }
      }
      // (no scheduling fence)
      const int r = tap / 3, sx = tap - 3 * r;
      const int toff = (r * HWD + sx) * 128;
      const char* wbuf = s_w + (j % 3) * WBUF;
      u32x4_t blo[TP], bhi[TP];
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        blo[p] = ld16(hcur + (pbase[p] + xlo[sx]) + toff);
        bhi[p] = ld16(hcur + (pbase[p] + xhi[sx]) + toff);
      }
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        const u32x4_t alo = ld16(wbuf + abase_lo + t * 2048);
        const u32x4_t ahi = ld16(wbuf + abase_hi + t * 2048);
        const v8i_t av = {(int)alo[0], (int)alo[1], (int)alo[2], (int)alo[3], (int)ahi[0], (int)ahi[1], (int)ahi[2], (int)ahi[3]};
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const v8i_t bv = {(int)blo[p][0], (int)blo[p][1], (int)blo[p][2], (int)blo[p][3], (int)bhi[p][0], (int)bhi[p][1], (int)bhi[p][2], (int)bhi[p][3]};
          acc[t][p] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[t][p], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      }
      // (no scheduling fence)
      if (tap >= HD) {                             // batch tap-HD of the next halo: fp8 into the OTHER halo buffer
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
          const int i = (tap - HD) * BATCH + b;
          if (i < NLD) store_chunk(hnext, hp_of(i), hsrc[(tap - HD) % (HD + 1)][b], hq[(tap - HD) % (HD + 1)][b]);
        }
      }
      if (tap != 8) store_w_from(wold, (j + 2) % 3);
      __syncthreads();
    }
    if (slab + 1 < nslabs) continue;

    // ---------------- epilogue of this tile.  Its lane constants are re-derived from an opaque copy of the thread index: as
    // loop invariants of the persistent stage loop (pointers into dequant / bias / stats / y, 64 bits each) they were hoisted in
    // front of it and spilled -- 400+ dwords of scratch traffic in the tap loop.
    int n0, h0, w0;
    tile_coords(tile, n0, h0, w0);
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int g = lane_e >> 4, li = lane_e & 15, wp = wave_e & 3, wk = wave_e >> 2;
    const int kb = k0 + wk * 64 + g * 16;
    float bias[16];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(q.w_dequant + kb + 4 * qd);
      f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) b4 = *reinterpret_cast<const f32x4_t*>(a.bias + kb + 4 * qd);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        bias[4 * qd + jj] = b4[jj];
        const float dq = d4[jj] / xs;
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[qd][p][jj] *= dq;
      }
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
      off[p] = ((((size_t)n * a.H + h) * a.W + w) * a.K + kb) * 2;
      ok[p] = true;
    }
    conv_store_tile<bf16_t, TK, TP>(acc, bias, off, ok, yg, rg, false, a.relu != 0);
    if (a.stats) {
      float s1[16], s2[16];
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          float x1 = 0.f, x2 = 0.f;
#pragma unroll
          for (int p = 0; p < TP; ++p) { const float v = acc[t][p][jj]; x1 += v; x2 = fmaf(v, v, x2); }
          s1[t * 4 + jj] = row16_sum(x1);
          s2[t * 4 + jj] = row16_sum(x2);
        }
      if (li == 0) {
        float* sp = a.stats + ((size_t)(tile * 4 + wp) * 2) * a.K + kb;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) { sp[jj] = s1[jj]; sp[a.K + jj] = s2[jj]; }
      }
    }
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  if (q.amax_out) {
    float m = amax / xs;
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
    m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(q.amax_out), __float_as_uint(m));     // non-negative floats order like their bits
  }
}

// 0: not served; 16: 16x16 tiles; 8: four images x 8x8
int conv_fp8_mode(const ConvArgs& a) {
  if (a.pix_mul > 1 || a.tap_mask || a.mask_x || a.par4 || a.out_scale) return 0;      // (its per-kout scale is w_dequant)
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.transposed || a.accumulate || a.osh != 1) return 0;
  if (a.PH != a.H || a.PW != a.W || a.OH != a.H || a.OW != a.W) return 0;
  if (a.C % 128 != 0 || a.K % 128 != 0 || a.C > 512) return 0;
  if (a.H % 16 == 0 && a.W % 16 == 0) return 16;
  if (a.H == 8 && a.W == 8 && a.N % 4 == 0) return 8;
  return 0;
}
int conv_fp8_rows(const ConvArgs& a) {             // per-tile partial (sum, sum^2) rows: 4 per 256-pixel tile
  const int m = conv_fp8_mode(a);
  return 4 * (m == 16 ? a.N * (a.H / 16) * (a.W / 16) : a.N / 4);
}
const char* conv_fp8_name(const ConvArgs& a) {
  if (a.in_scale) return conv_fp8_mode(a) == 16 ? "sslcr::conv3x3_fp8_kernel<16, true>" : "sslcr::conv3x3_fp8_kernel<8, true>";
  return conv_fp8_mode(a) == 16 ? "sslcr::conv3x3_fp8_kernel<16, false>" : "sslcr::conv3x3_fp8_kernel<8, false>";
}

template <int TW, bool XF>
static hipError_t launch_f8(const ConvArgs& a, const Fp8Args& q, hipStream_t st) {
  constexpr int HP = (256 / (TW * TW)) * (TW + 2) * (TW + 2);
  constexpr int HBUF = ((HP * 128 + 1023) / 1024) * 1024;
  const size_t lds = 2 * HBUF + 3 * 128 * 128 + (XF ? 2 * a.C * sizeof(float) : 0);
  auto kern = conv3x3_fp8_kernel<TW, XF>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int cus = device_cus();
  const int ntiles = TW == 16 ? a.N * (a.H / 16) * (a.W / 16) : a.N / 4;
  if ((size_t)a.N * a.H * a.W * a.C * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;      // 32-bit lane offsets (see load_chunk)
  const int gy = a.K / 128;
  int gx = cus / gy;                               // one persistent workgroup per CU over (tile walkers) x (128-kout blocks)
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(512), lds, st, a, q, ntiles);
  return hipGetLastError();
}

hipError_t launch_conv_fp8(const ConvArgs& a, const Fp8Args& q, hipStream_t st) {
  const int m = conv_fp8_mode(a);
  if (m == 0 || !q.w8 || !q.w_dequant || !(q.x_scale > 0.f)) return hipErrorInvalidValue;
  if (a.in_scale) return m == 16 ? launch_f8<16, true>(a, q, st) : launch_f8<8, true>(a, q, st);
  return m == 16 ? launch_f8<16, false>(a, q, st) : launch_f8<8, false>(a, q, st);
}

// ---- e4m3 shadow pack of a 3x3 filter bank: [K][C][3][3] fp32 -> [K][9][C] fp8 with a power-of-two scale per output channel
// (amax * scale in (224, 448]); eval form folds BatchNorm like pack_conv_kernel.  One workgroup per output channel.
__global__ __launch_bounds__(256) void pack_fp8_kernel(const PackFp8Args a) {
  __shared__ float sm[4];
  const int k = blockIdx.x;
  const int n = a.C * 9;
  const float f = a.gamma ? a.gamma[k] / sqrtf(a.rvar[k] + a.eps) : 1.f;
  const float* w = a.w + (size_t)k * n;
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i] * f));
  m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
  m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  // scale = 2^e with amax * scale <= 448 < 2 * amax * scale; an all-zero filter keeps scale 1
  float scale = 1.f;
  if (m > 0.f) {
    int e;
    (void)frexpf(448.f / m, &e);                  // 448 / m = fr * 2^e, fr in [0.5, 1)  ->  2^(e-1) <= 448 / m
    scale = ldexpf(1.f, e - 1);
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i / 9, rs = i - 9 * c;
    const float v = __builtin_amdgcn_fmed3f(w[i] * f * scale, -448.f, 448.f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
    a.w8[((size_t)k * 9 + rs) * a.C + c] = (uint8_t)(pk & 0xff);
  }
  if (threadIdx.x == 0) {
    a.w_dequant[k] = 1.f / scale;
    if (a.gamma && a.bias_out) a.bias_out[k] = a.beta[k] - a.rmean[k] * f;
  }
}

// delayed scaling: slots [n][2] = {x_scale, amax seen since the last update}: scale <- 2^floor(log2(448 / (2 amax))), amax <- 0
__global__ void fp8_scale_update_kernel(float* slots, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = slots[2 * i + 1];
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(224.f / m, &e);
    slots[2 * i] = fminf(fmaxf(ldexpf(1.f, e - 1), 1.0f / 65536.f), 65536.f);
  }
  slots[2 * i + 1] = 0.f;
}
hipError_t launch_fp8_scale_update(float* slots, int n, hipStream_t st) {
  hipLaunchKernelGGL(fp8_scale_update_kernel, dim3((n + 63) / 64), dim3(64), 0, st, slots, n);
  return hipGetLastError();
}

hipError_t launch_pack_fp8(const PackFp8Args& a, hipStream_t st) {
  hipLaunchKernelGGL(pack_fp8_kernel, dim3(a.K), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace sslcr

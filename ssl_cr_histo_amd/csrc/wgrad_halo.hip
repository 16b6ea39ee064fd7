// Weight gradient of the 3x3 / stride 1 / pad 1 convs, LDS-halo form.
//
//   dW[k][r][s][c] += sum_{pixels} dY[pix][k] * act(X)[pix + (r-1, s-1)][c]
//
// The gather form (conv_wgrad.hip) re-stages the shifted input once per tap: 40 KiB of global->LDS traffic per 32 pixels,
// which is the LDS-write / L1 limit, not the matrix pipe.  Here a workgroup walks 128-pixel tiles (8x16, or two 8x8
// images); per tile it stages the dY tile and ONE (8+2)x(TW+2) input halo (producer BatchNorm+ReLU applied on the way) and
// all nine taps fetch their B fragments from the halo at shifted pixel addresses -- ds_read_b64_tr_b16 takes a per-lane
// address, so the shift is free.  4.3x less staging traffic per MAC.  Same 64(kout) x 64(cin) x 9 register tile, wave w
// owning cin tile w; the pixel splits' accumulators go to slabs that wgrad_fold_kernel adds into dW in a fixed order.
#include <cstdio>
#include <mutex>

#include "kernels.hpp"

namespace sslcr {

typedef short h16x4_t __attribute__((ext_vector_type(4)));
typedef short h16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t tr_pair(const char* p0, const char* p1) {
  h16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4_t*)(p0));
  h16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4_t*)(p1));
  h16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// KH = 1: a workgroup of four waves owns a 64(kout) x 64(cin) block, two workgroups share a CU and cover each other's staging
//      phases (single LDS buffer).
// KH = 2 (bf16, K % 128 == 0): eight waves own a 128(kout) x 64(cin) block -- wave = (cin tile, kout half).  The input halo
//      and its producer-BatchNorm transform are staged once per 128 kouts instead of once per 64 (the halo is the operand
//      that is re-read K/64 times: PMC traffic 2.3x / 3.4x algorithmic on layers 2 / 3), and with ONE workgroup per CU the
//      (dY, halo) buffer is doubled: the next tile is transformed and written to the other buffer in the MIDDLE of this
//      tile's MFMA loop, one barrier per tile.
template <typename T, int TW, int KH>
__global__ __launch_bounds__(256 * KH, 2) void wgrad3x3_halo_kernel(const WgradArgs a, int tiles_per_split, int ntiles, f32x4_t* partials) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr bool BF = Elem<T>::DT == DT_BF16;
  constexpr int RB = 64 * sizeof(T);            // LDS bytes per pixel row (64 channels)
  constexpr int CPR = RB / 16;                  // 16-byte chunks per row (8 / 16)
  constexpr int TH = 8;
  constexpr int NI = 128 / (TH * TW);
  constexpr int HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;             // 180 / 200 halo pixels staged per tile
  constexpr int NT = 256 * KH;
  constexpr int YL = 128 * CPR / 256;           // dY staging loads per thread (4 / 8): 128 pixels x KH*64 kouts over NT threads
  constexpr int HL = (HP * CPR + NT - 1) / NT;  // halo staging loads per thread
  constexpr int YROWS = 256 / CPR, HROWS = NT / CPR;   // pixel rows covered per staging pass
  // bf16: halo rows are pitched to 24 (16-wide tiles) / 16 pixels -- multiples of 8 -- so that bits 1..2 of a halo pixel
  // index depend on the lane and the filter COLUMN only: the bank swizzle below is then a per-lane constant per column and
  // every transpose read is base register + immediate (3 + 4 address registers for all nine taps)
  constexpr int PITCH = BF ? (TW == 16 ? 24 : 16) : HWD;
  constexpr int YH = 128 * RB;                  // one 64-kout half of the dY tile
  constexpr int YBUF = KH * YH, HBUF = NI * HH * PITCH * RB;
  constexpr int BUF = YBUF + HBUF;
  constexpr int NBUF = KH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3, kh = tid >> 8;         // cin tile, kout half
  const int li = lane & 15, g = lane >> 4;
  // grid x = (kout block, cin block, pixel split) triples.  The gx * gy workgroups of ONE pixel split read the same dY / X tiles
  // (each its own channel slice, but a dY row is re-read by every cin block and an X row by every kout block): they are made
  // neighbours on one XCD (linear ids w, w + 8, ...; workgroups land on XCD id % 8) so that those re-reads hit its L2
  const int gx = a.K / (64 * KH), gy = a.C / 64, GT = gx * gy, splits = (int)gridDim.x / GT;
  int bz, bt;
  if ((splits & 7) == 0) {
    const int w = blockIdx.x, grp = w / (8 * GT), r = w - grp * 8 * GT;
    bz = grp * 8 + (r & 7); bt = r >> 3;
  } else {
    bz = (int)blockIdx.x / GT; bt = (int)blockIdx.x - bz * GT;
  }
  const int by = bt / gx, bx = bt - by * gx;
  const int k0 = bx * (64 * KH), c0 = by * 64;
  const int chunk = tid % CPR, prow = tid / CPR;          // halo staging role
  const int chunky = tid % (CPR * KH), prowy = tid / (CPR * KH);   // dY staging role: 16-byte chunk of the KH*64 kouts, pixel row
  const bool xform = a.in_scale != nullptr;
  // producer BN scale/shift of this workgroup's 64 input channels: kept in LDS (behind the tile buffer) and read when a
  // halo is staged -- as registers they cost 16 VGPRs through the MFMA loop, which is register-bound
  // (one set per segment of seg_images images when the batch carries several producer BatchNorms, see sslcr_wgrad_desc)
  float* s_aff = reinterpret_cast<float*>(smem + NBUF * BUF);
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  for (int i = tid; i < nseg * 64; i += NT) {
    const int sg = i >> 6, ch = i & 63;
    s_aff[sg * 128 + ch] = xform ? a.in_scale[(size_t)sg * a.seg_stride + c0 + ch] : 1.f;
    s_aff[sg * 128 + 64 + ch] = xform ? a.in_shift[(size_t)sg * a.seg_stride + c0 + ch] : 0.f;
  }
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  const int t_begin = bz * tiles_per_split;
  int t_end = t_begin + tiles_per_split;
  if (t_end > ntiles) t_end = ntiles;
  if (t_begin >= t_end) {                       // (the launcher sizes the splits so that none is empty; a slab must still be defined)
    if (partials) {
      const size_t wg = ((size_t)bz * gy + by) * gx + bx;
      for (int e = 0; e < 36; ++e) partials[(wg * 36 + e) * NT + tid] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* dyg = reinterpret_cast<const char*>(a.dy);

  // Staging roles are fixed per thread: pixel offsets relative to the tile origin and "touches the top/bottom/left/right
  // halo ring" flags are computed ONCE, so a tile costs one add per load and a mask test instead of div/mod chains (the
  // kernel is VALU-issue bound: 735 VALU per 144 MFMAs before this).
  int rel_y[YL], rel_h[HL];
  static_assert(HL <= 16, "edge mask holds 16 halo entries");
  unsigned long long edge = 0;               // 4 bits per halo entry: top, bottom, left, right
  unsigned hvalid = 0;                       // entry exists (hp < HP)
#pragma unroll
  for (int i = 0; i < YL; ++i) {
    const int p = prowy + YROWS * i;
    const int ni = p / (TH * TW), rem = p - ni * (TH * TW);
    const int ph = rem / TW, pw = rem - ph * TW;
    rel_y[i] = (ni * a.H + ph) * a.W + pw;
  }
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
  const size_t ybase = ((size_t)k0 + chunky * EPC) * sizeof(T), xbase = ((size_t)c0 + chunk * EPC) * sizeof(T);

  u32x4_t yreg[YL], hreg[HL];
  unsigned hin = 0;
  int seg_ld = 0;                               // segment of the tile whose operands sit in yreg / hreg
  auto load_regs = [&](int tile) {
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    const int n0 = (t / tiles_h) * NI;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    seg_ld = a.seg_images > 0 ? n0 / a.seg_images : 0;
    const int origin = (n0 * a.H + h0) * a.W + w0;                    // pixel index of the tile's (0,0)
    // which halo rings fall outside the image for this tile (uniform)
    const unsigned long long out =
        (unsigned long long)((h0 == 0) | ((h0 + TH >= a.H) << 1) | ((w0 == 0) << 2) | ((w0 + TW >= a.W) << 3)) * 0x1111111111111111ull;
#pragma unroll
    for (int i = 0; i < YL; ++i)
      yreg[i] = ld16(dyg + (size_t)(origin + rel_y[i]) * a.K * sizeof(T) + ybase);
    hin = hvalid;
    const unsigned long long bad = edge & out;                                   // entry i is padding iff any of its 4 bits is set
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      // branch-free: a padding entry loads the tile origin (a valid address) and is zeroed when staged (as `if (ok) v = load` every
      // load was followed by its own s_waitcnt vmcnt(0), the merge of the two definitions: one exposed round trip per entry)
      const bool ok = ((hvalid >> i) & 1u) && !((bad >> (4 * i)) & 0xfull);
      hreg[i] = ld16(xg + (size_t)(origin + (ok ? rel_h[i] : 0)) * a.C * sizeof(T) + xbase);
      if (!ok) hin &= ~(1u << i);
    }
  };
  // Bank swizzle of the bf16 tiles.  A transpose read presents, per 32 lanes, 8 pixel rows x 32 bytes at ONE column offset;
  // rows are 128 B = half the banks apart, so unswizzled the four same-parity rows collide 4-way (measured: 68 % of the
  // kernel's LDS cycles were bank conflicts).  The lanes are mapped so that those 8 rows are 8 CONSECUTIVE pixels of one
  // image row, and the 32-byte column group is XORed with bits 1..2 of the row index.
  auto swz = [](int row) { return (row >> 1) & 3; };
  auto chunk_off_of = [&](int row, int ch) {    // byte offset of 16-byte chunk `ch` inside its (128-byte, bf16) row
    if (!BF) return ch * 16;
    return (((ch >> 1) ^ swz(row)) << 5) | ((ch & 1) << 4);
  };
  auto chunk_off = [&](int row) { return chunk_off_of(row, chunk); };
  auto store_lds = [&](int buf) {
    char* yb = smem + buf * BUF;
    char* hb = yb + YBUF;
    float r_scale[EPC], r_shift[EPC];
    if (xform) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { r_scale[e] = s_aff[seg_ld * 128 + chunk * EPC + e]; r_shift[e] = s_aff[seg_ld * 128 + 64 + chunk * EPC + e]; }
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      const int p = prowy + YROWS * i;
      st16(yb + (chunky / CPR) * YH + p * RB + chunk_off_of(p, chunky % CPR), yreg[i]);
    }
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int hp = prow + HROWS * i;
      if (hp >= HP) continue;
      const int hni = hp / (HH * HWD), hrem = hp - hni * (HH * HWD);
      const int hpl = (hni * HH + hrem / HWD) * PITCH + hrem % HWD;      // pitched LDS pixel index
      u32x4_t v = hreg[i];
      if (!((hin >> i) & 1u)) {
        v = u32x4_t{0u, 0u, 0u, 0u};              // padding
      } else if (xform) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float q = fmaf(f[e], r_scale[e], r_shift[e]);
          f[e] = a.in_relu ? fmaxf(q, 0.f) : q;
        }
        v = Elem<T>::pack(f);
      }
      st16(hb + hpl * RB + chunk_off(hpl), v);
    }
  };

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // halo pixel index of tile pixel p, tap (0,0)
  auto hpix = [&](int p) {
    if (TW == 16) return (p >> 4) * PITCH + (p & 15);
    return (p >> 6) * (HH * PITCH) + ((p >> 3) & 7) * PITCH + (p & 7);
  };

  load_regs(t_begin);
  __syncthreads();                              // s_aff
  store_lds(0);
  __syncthreads();
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) load_regs(tile + 1);
    const char* yb = smem + buf * BUF;
    const char* hb = yb + YBUF;
    if constexpr (BF) {
      // 32-pixel MFMA depth steps of the 128-pixel tile.  The B fragment of the NEXT tap (or of tap 0 of the next depth
      // step) is requested before the four MFMAs of this tap; scheduler fences keep it that way (left alone the compiler
      // emits read -> wait -> 4 MFMA per tap and the matrix pipe idles for an LDS round trip nine times per step).
      // this lane's source pixels within a 32-pixel depth step: pl and pl + 8 (any lane -> pixel map works as long as dY and
      // X use the same one; this one keeps the 8 pixel rows of a 32-lane group consecutive, see swz)
      const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
      constexpr int HI = (TW == 16 ? 8 : PITCH) * RB;       // halo byte offset of pixel pl + 8 (8-wide tiles: next image row)
      int Aoff[4], Boff[3];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) Aoff[t4] = pl * RB + ((t4 ^ swz(pl)) << 5) + (li & 3) * 8;
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        const int hp = hpix(pl) + sx;
        Boff[sx] = hp * RB + ((wave ^ swz(hp)) << 5) + (li & 3) * 8;
      }
      bf16x8_t bfr[2];
      auto bfrag_of = [&](int q, int t) {
        const int qoff = (hpix(q * 32) + (t / 3) * PITCH) * RB;       // depth step and filter row: uniform byte offset
        return tr_pair(hb + Boff[t % 3] + qoff, hb + Boff[t % 3] + qoff + HI);
      };
      auto qstep = [&](const int q, const int par) {       // par: which of bfr[] holds (q, tap 0)
        bf16x8_t af[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const char* pa = yb + kh * YH + q * 32 * RB + Aoff[t4];
          af[t4] = tr_pair(pa, pa + 8 * RB);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          bfr[(t + 1 + par) & 1] = t < 8 ? bfrag_of(q, t + 1) : bfrag_of(q < 3 ? q + 1 : 3, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
            acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t4], bfr[(t + par) & 1], acc[t][t4], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      bfr[0] = bfrag_of(0, 0);
#pragma unroll 1
      for (int q2 = 0; q2 < 2; ++q2) {           // nine taps flip the buffer parity: two depth steps per trip
        qstep(2 * q2, 0);
        qstep(2 * q2 + 1, 1);
      }
    } else {
#pragma unroll 2
      for (int q = 0; q < 32; ++q) {             // 4-pixel fp32 MFMA depth steps
        const int p = 4 * q + g;
        float av[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) av[t4] = *reinterpret_cast<const float*>(yb + p * RB + (16 * t4 + li) * 4);
        const int hp0 = hpix(p);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int toff = (t / 3) * PITCH + (t % 3);
          const float bv = *reinterpret_cast<const float*>(hb + (hp0 + toff) * RB + (16 * wave + li) * 4);   // (fp32: KH == 1)
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], bv, acc[t][t4], 0, 0, 0);
        }
      }
    }
    if (NBUF == 1) {
      __syncthreads();                          // single buffer: everyone is done reading before it is overwritten
      if (more) store_lds(0);
    } else if (more) {
      // double buffer: no barrier between this tile's MFMAs and the staging of the next one into the other buffer -- a wave
      // that finishes early transforms and writes its share while the others still compute (staging it in the MIDDLE of
      // the MFMA loop instead spilled 38-121 registers: the loop body sits at 252 of 256)
      store_lds(buf ^ 1);
    }
    __syncthreads();
    if (NBUF == 2) buf ^= 1;
  }

  if (partials) {
    // the workgroup's 36 accumulator vectors per thread go to its slab with plain coalesced 16-byte stores; wgrad_fold_kernel
    // adds the slabs into dW.  As fp32 atomics straight into dW (one resident round of workgroups = 75 MB of 4-byte atomics
    // per launch, whatever the layer) they were 20-26 % of the kernel: 248 -> 175 us without them on the layer2 shape,
    // 186 us with these stores.
    const size_t wg = ((size_t)bz * gy + by) * gx + bx;
    f32x4_t* sp = partials + wg * 36 * NT + tid;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) sp[(size_t)(t * 4 + t4) * NT] = acc[t][t4];
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int j = 0; j < 4; ++j) {               // one pixel split: this workgroup is the only writer of its block of dW
        const int k = k0 + 64 * kh + 16 * t4 + 4 * g + j;
        a.dw[((size_t)k * 9 + t) * a.C + c0 + 16 * wave + li] += acc[t][t4][j];
      }
}

// dW += sum over the pixel splits of the workgroups' accumulator slabs (layout of the store above), in ONE fixed order: no atomics,
// so a weight gradient is the same bits run after run (the exact-parity fp32 mode is held to a 24-iteration trajectory, and Adam
// turns a changed last bit into an lr-sized step).  Block = 16 / 32 / 64 slab vectors (x) by ZG split lanes (y): split lane y adds the
// vectors of splits y, y + ZG, ... in rising order, the ZG lane sums are added in lane order by lane 0, which is the only writer of
// its four dW elements.  mode 0: the (kout block, cin block, tap, thread) layout of wgrad3x3_halo_kernel / wgrad_kernel;
// mode 1: stem_wgrad's (feature vector f, thread) layout, dW in PyTorch's [64][3][7][7].
__global__ __launch_bounds__(1024) void wgrad_fold_kernel(const f32x4_t* __restrict__ partials, float* dw, int C, int gx, int gy, int splits,
                                                         int ev, int kh_n, int mode) {
  __shared__ f32x4_t red[16][64];
  const int NT = 256 * kh_n, taps = ev >> 2;                        // threads per workgroup; ev = accumulator vectors per thread
  const size_t per_wg = (size_t)ev * NT;
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (by, bx, e, tid) flattened, tid fastest; the total is a multiple of 64
  const int ZG = blockDim.y, zl = threadIdx.y;
  const int tid = (int)(v % NT);
  const int e = (int)((v / NT) % ev);
  const int bxy = (int)(v / per_wg);
  const int bx = bxy % gx, by = bxy / gx;
  f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
  const size_t zstride = (size_t)gy * gx * per_wg;
  const f32x4_t* p = partials + (size_t)zl * zstride + ((size_t)by * gx + bx) * per_wg + (size_t)e * NT + tid;
  for (int z = zl; z < splits; z += ZG, p += (size_t)ZG * zstride) {
    const f32x4_t q = __builtin_nontemporal_load(p);
    sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2]; sum[3] += q[3];
  }
  if (ZG > 1) {
    red[zl][threadIdx.x] = sum;
    __syncthreads();
    if (zl != 0) return;
    for (int y = 1; y < ZG; ++y) {
      const f32x4_t q = red[y][threadIdx.x];
      sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2]; sum[3] += q[3];
    }
  }
  const int lane = tid & 63, wave = (tid >> 6) & 3, kh = tid >> 8;
  const int li = lane & 15, g = lane >> 4;
  if (mode == 1) {
    const int r = e >> 1, s = (e & 1) * 4 + (li >> 2), c = li & 3;
    if (s < 7 && c < 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[(((16 * wave + 4 * g + j) * 3 + c) * 7 + r) * 7 + s] += sum[j];
    }
    return;
  }
  const int t = e >> 2, t4 = e & 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = bx * 64 * kh_n + 64 * kh + 16 * t4 + 4 * g + j;
    dw[((size_t)k * taps + t) * C + by * 64 + 16 * wave + li] += sum[j];
  }
}
static hipError_t fold_launch(const void* slabs, float* dw, int C, int gx, int gy, int splits, int ev, int kh_n, int mode, hipStream_t st) {
  const size_t nvec = (size_t)gx * gy * ev * 256 * kh_n;
  int zg = 1;                                                       // >= 4 vectors per split lane, at most 16 lanes
  while (zg < 16 && splits >= 8 * zg) zg *= 2;
  int vx = 64;                                                      // vectors per block: fewer where the launch would not fill the chip
  while (vx > 16 && nvec / vx < (size_t)(2 * device_cus())) vx >>= 1;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)(nvec / vx)), dim3(vx, zg), 0, st, reinterpret_cast<const f32x4_t*>(slabs), dw, C, gx, gy,
                     splits, ev, kh_n, mode);
  return hipGetLastError();
}
hipError_t launch_wgrad_fold(const void* slabs, float* dw, int C, int gx, int gy, int splits, int taps, int kh_n, hipStream_t st) {
  return fold_launch(slabs, dw, C, gx, gy, splits, taps * 4, kh_n, 0, st);
}
// stem form: `nwg` slabs of 14 accumulator vectors x 256 threads
hipError_t launch_stem_wgrad_fold(const void* slabs, float* dw, int nwg, hipStream_t st) {
  return fold_launch(slabs, dw, 0, 1, 1, nwg, 14, 1, 1, st);
}

// slabs of the launches on one stream (a launch's fold has consumed them before the next launch on that stream writes; the
// BatchNorm-backward reduce pass keeps its per-workgroup rows here too).
// One entry per stream, 64 entries, least-recently-used eviction (logged once: it costs a device synchronise): a 65th stream (virtual-rank tests create a stream per context
// and drop it) takes over the oldest entry after a device synchronise -- never a silent fall-back to the atomic path, whose
// summation order differs -- and an evicted or destroyed stream's slab is freed instead of leaking.
namespace {
struct Slab { hipStream_t st; void* p; size_t cap; unsigned long long used; };
constexpr int NSLAB = 64;
Slab g_slabs[NSLAB];
int g_nslabs = 0;
unsigned long long g_slab_tick = 0;
std::mutex g_slab_mu;                            // host threads driving different streams
}  // namespace
// sslcr_destroy: every stream's slab is freed (the device has been synchronised; a later launch allocates again)
void stream_scratch_release() {
  std::lock_guard<std::mutex> lock(g_slab_mu);
  for (int i = 0; i < g_nslabs; ++i)
    if (g_slabs[i].p) (void)hipFree(g_slabs[i].p);
  g_nslabs = 0;
}
void* stream_scratch(hipStream_t st, size_t bytes) {
  Slab* const slabs = g_slabs;
  int& n = g_nslabs;
  unsigned long long& tick = g_slab_tick;
  std::lock_guard<std::mutex> lock(g_slab_mu);
  Slab* e = nullptr;
  for (int i = 0; i < n && !e; ++i)
    if (slabs[i].st == st) e = &slabs[i];
  if (!e) {
    if (n < NSLAB) {
      e = &slabs[n++];
    } else {
      e = &slabs[0];
      for (int i = 1; i < NSLAB; ++i)
        if (slabs[i].used < e->used) e = &slabs[i];
      static bool warned = false;
      if (!warned) {
        warned = true;
        fprintf(stderr, "sslcr: more than %d streams have launched weight-gradient / BatchNorm-backward kernels; the least recently used "
                        "stream's slab is evicted after a device synchronise (slow when it happens per launch)\n", NSLAB);
      }
      (void)hipDeviceSynchronize();              // the evicted stream may be gone: wait for the device, not for the stream
      if (e->p) (void)hipFree(e->p);
    }
    *e = Slab{st, nullptr, 0, 0};
  }
  e->used = ++tick;
  if (e->cap < bytes) {
    if (e->p) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(e->p);
    }
    e->p = nullptr; e->cap = 0;
    if (hipMalloc(&e->p, bytes) != hipSuccess) return nullptr;
    e->cap = bytes;
  }
  return e->p;
}

int wgrad_halo_tw(const WgradArgs& a) {
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.OH != a.H || a.OW != a.W) return 0;
  if (a.C % 64 != 0 || a.K % 64 != 0 || a.H % 8 != 0) return 0;
  if (a.W % 16 == 0) return 16;
  if (a.W % 8 == 0 && a.N % 2 == 0) return 8;
  return 0;
}

// pixel splits of a launch: exactly one resident round (two workgroups per CU for KH = 1, one 8-wave workgroup for KH = 2), at least
// 16 tiles per workgroup -- below that the slab of its 64x64x9 block outweighs the parallelism (RSP N=128: +4 % step)
static int wh_splits(const WgradArgs& a, int TW, int KH, int* tps_out, int* ntiles_out) {
  const int NI = 128 / (8 * TW);
  const int ntiles = (a.N / NI) * (a.H / 8) * (a.W / TW);
  const int kc = (a.K / (64 * KH)) * (a.C / 64);
  int splits = cdiv((KH == 1 ? 2 : 1) * device_cus(), kc);
  const int max_splits = cdiv(ntiles, 16);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int tps = cdiv(ntiles, splits);
  if (tps_out) *tps_out = tps;
  if (ntiles_out) *ntiles_out = ntiles;
  return cdiv(ntiles, tps);
}
// does the DMA form (wgrad_dma.hip) serve this launch?
bool wgrad_dma_used(int dtype, const WgradArgs& a) {
  const int tw = wgrad_halo_tw(a);
  if (!tw || dtype != DT_BF16 || a.K % 128 != 0) return false;
  return wgrad_dma_ok(dtype, a, wh_splits(a, tw, 2, nullptr, nullptr));
}

template <typename T, int TW, int KH>
static hipError_t launch_wh(const WgradArgs& a, hipStream_t st) {
  constexpr bool BF = Elem<T>::DT == DT_BF16;
  constexpr int NI = 128 / (8 * TW);
  int tps, ntiles;
  const int splits = wh_splits(a, TW, KH, &tps, &ntiles);
  const int pitch = BF ? (TW == 16 ? 24 : 16) : TW + 2;
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  if (nseg > 8 || (a.seg_images > 0 && (a.N % a.seg_images != 0 || a.seg_images % NI != 0))) return hipErrorInvalidValue;
  const size_t lds = (size_t)KH * (128 * KH + NI * 10 * pitch) * 64 * sizeof(T) + 512 * nseg;     // NBUF = KH buffers of (KH dY halves + halo)
  auto kern = wgrad3x3_halo_kernel<T, TW, KH>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, KH == 1 ? 96 * 1024 : 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int gx = a.K / (64 * KH), gy = a.C / 64;
  // accumulator slabs + the ordered fold when there is more than one split (every dtype)
  f32x4_t* slabs = nullptr;
  if (splits > 1) {
    slabs = reinterpret_cast<f32x4_t*>(stream_scratch(st, (size_t)gx * gy * splits * 36 * 256 * KH * sizeof(f32x4_t)));
    if (!slabs) return hipErrorOutOfMemory;        // (no atomic path to fall back to: its summation order would differ)
  }
  if (KH == 2 && BF && wgrad_dma_ok(Elem<T>::DT, a, splits)) {      // operands by LDS DMA, same tiles / slabs / bits (wgrad_dma.hip)
    hipError_t e = launch_wgrad_dma(a, TW, tps, ntiles, splits, slabs, st);
    if (e != hipSuccess) return e;
    return launch_wgrad_fold(slabs, a.dw, a.C, gx, gy, splits, 9, KH, st);
  }
  hipLaunchKernelGGL(kern, dim3(gx * gy * splits), dim3(256 * KH), lds, st, a, tps, ntiles, slabs);
  if (slabs) return launch_wgrad_fold(slabs, a.dw, a.C, gx, gy, splits, 9, KH, st);
  return hipGetLastError();
}

hipError_t launch_wgrad_halo(int dtype, const WgradArgs& a, int tw, hipStream_t st) {
  if (dtype == DT_BF16) {
    if (a.K % 128 == 0) return tw == 16 ? launch_wh<bf16_t, 16, 2>(a, st) : launch_wh<bf16_t, 8, 2>(a, st);
    return tw == 16 ? launch_wh<bf16_t, 16, 1>(a, st) : launch_wh<bf16_t, 8, 1>(a, st);
  }
  return tw == 16 ? launch_wh<float, 16, 1>(a, st) : launch_wh<float, 8, 1>(a, st);
}

}  // namespace sslcr

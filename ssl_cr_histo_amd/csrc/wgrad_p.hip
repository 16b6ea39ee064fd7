// Weight gradient of the 3x3 / stride 1 / pad 1 convs, bf16, one-wave-per-SIMD double-buffered form.
//
//   dW[k][r][s][c] += sum_{pixels} dY[pix][k] * act(X)[pix + (r-1, s-1)][c]
//
// Same math and fragment scheme as wgrad_halo.hip (dY^T and X^T fragments by ds_read_b64_tr_b16 out of pixel-major LDS
// tiles, fp32 atomics into dW), but built around what the measurements of that kernel said: its inner loop alone runs at
// 1.37 PF/s, the kernel at 0.7 -- the other half is the per-tile staging (global loads -> BN+ReLU -> LDS) sitting between
// two barriers with the matrix pipe idle, and at 256 VGPRs there is no room to overlap it.  Here
//   * a workgroup is FOUR waves, one per SIMD, each with the full 512-register budget: a wave owns 64 kout x 32 cin x 9 taps
//     (288 fp32 accumulators), so every dY fragment feeds twice the MFMAs and nothing spills;
//   * the CU holds one workgroup with two (dY, halo) LDS buffers.  Staging is a rolling register pipeline: the registers
//     hold tile t+1 while tile t is computed; after the MFMAs of a filter tap ONE 16-byte item is transformed (producer
//     BN+ReLU) and stored into the other buffer and its register is immediately re-loaded for tile t+2.  One barrier per
//     tile, nothing but LDS fragment reads between barriers, a full tile of latency cover for every global load;
//   * 256-byte halo rows (CB = 128) are swizzled by (row & 7) over their eight 32-byte column groups, 128-byte rows by
//     ((row >> 1) & 3) over four: conflict-free transpose reads; rows pitched to a multiple of 8 so that the key is a
//     lane constant per filter column and every read is base + immediate; A and B fragments are double-buffered.
#include <cstdlib>
#include "kernels.hpp"

namespace sslcr {

typedef short p16x4_t __attribute__((ext_vector_type(4)));
typedef short p16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t trp(const char* p0, const char* p1) {
  p16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) p16x4_t*)(p0));
  p16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) p16x4_t*)(p1));
  p16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int TW, int CB>
__global__ __launch_bounds__(256, 1) void wgrad3x3_p_kernel(const WgradArgs a, int tiles_per_split, int ntiles) {
  using T = bf16_t;
  constexpr int EPC = 8;
  constexpr int TH = 8, NI = 128 / (TH * TW), HH = TH + 2, HWD = TW + 2;
  constexpr int HP = NI * HH * HWD;             // staged halo pixels per tile (180 / 200)
  constexpr int PITCH = TW == 16 ? 24 : 16;     // LDS halo row pitch in pixels (multiple of 8, see header)
  constexpr int YROW = 128, HROW = CB * 2;      // bytes per dY / halo pixel row
  constexpr int CPRH = CB / 8;                  // 16-byte chunks per halo row
  constexpr int YBUF = 128 * YROW, HBUF = NI * HH * PITCH * HROW, BUF = YBUF + HBUF;
  constexpr int YL = 4, HL = (HP * CPRH + 255) / 256;
  constexpr int NIT = HL + YL;                  // staging items per thread and tile
  constexpr int NCP = CB / 32;                  // cin pairs (a wave owns two 16-channel tiles)
  constexpr int QW = 4 / NCP, QPW = 4 / QW;     // wave groups along the tile's four 32-pixel depth steps; steps per wave
  static_assert(NIT < 9 * QPW, "one staging item per tap slot (+1 for the trailing reload)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_aff = reinterpret_cast<float*>(smem + 2 * BUF);      // [scale CB][shift CB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int cp = wave % NCP, qh = wave / NCP;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * CB;
  const bool xform = a.in_scale != nullptr;
  const float relu_lo = a.in_relu ? 0.f : -__builtin_inff();
  for (int c = tid; c < CB; c += 256) {
    s_aff[c] = xform ? a.in_scale[c0 + c] : 1.f;
    s_aff[CB + c] = xform ? a.in_shift[c0 + c] : 0.f;
  }
  const int tiles_w = a.W / TW, tiles_h = a.H / TH;
  const int t_begin = blockIdx.z * tiles_per_split;
  int t_end = t_begin + tiles_per_split;
  if (t_end > ntiles) t_end = ntiles;
  if (t_begin >= t_end) return;
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* dyg = reinterpret_cast<const char*>(a.dy);

  auto swz_y = [](int p) { return (p >> 1) & 3; };
  auto swz_h = [](int hp) { return CB == 128 ? (hp & 7) : ((hp >> 1) & 3); };

  // ---- staging roles (fixed per thread): items 0..HL-1 halo chunks, HL..NIT-1 dY chunks
  const int ychunk = tid & 7, yp0 = tid >> 3;                    // dY: pixels yp0 + 32 * i
  const int hchunk = tid % CPRH, hs0 = tid / CPRH;               // halo: staged pixels hs0 + (256 / CPRH) * i
  int rel_y0, rel_ystep;
  {
    const int ni = yp0 / (TH * TW), rem = yp0 - ni * (TH * TW);
    rel_y0 = (ni * a.H + rem / TW) * a.W + rem % TW;
    rel_ystep = TW == 16 ? 2 * a.W : 4 * a.W;                     // 32 pixels further: 2 / 4 image rows
  }
  int rel_h[HL], hpl[HL];
  unsigned long long edge = 0;                                    // 4 bits per entry: top, bottom, left, right
  unsigned hvalid = 0;
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hs = hs0 + (256 / CPRH) * i;
    rel_h[i] = 0; hpl[i] = 0;
    if (hs < HP) {
      const int ni = hs / (HH * HWD), rem = hs - ni * (HH * HWD);
      const int hr = rem / HWD, hc = rem - hr * HWD;
      rel_h[i] = (ni * a.H + hr - 1) * a.W + hc - 1;
      hpl[i] = (ni * HH + hr) * PITCH + hc;
      hvalid |= 1u << i;
      edge |= (unsigned long long)((hr == 0) | ((hr == HH - 1) << 1) | ((hc == 0) << 2) | ((hc == HWD - 1) << 3)) << (4 * i);
    }
  }
  const size_t ybase = ((size_t)k0 + ychunk * EPC) * sizeof(T), xbase = ((size_t)c0 + hchunk * EPC) * sizeof(T);

  struct Geo { int origin; unsigned long long bad; };
  auto geom = [&](int tile) {
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    const int n0 = (t / tiles_h) * NI;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    Geo q;
    q.origin = (n0 * a.H + h0) * a.W + w0;
    q.bad = edge & ((unsigned long long)((h0 == 0) | ((h0 + TH >= a.H) << 1) | ((w0 == 0) << 2) | ((w0 + TW >= a.W) << 3)) *
                    0x1111111111111111ull);
    return q;
  };
  float sc[EPC], sh[EPC];                       // producer BN scale/shift of this thread's halo chunk (set after s_aff is visible)
  u32x4_t sreg[NIT];                            // the rolling staging registers
  unsigned hin_cur = 0, hin_nxt = 0;            // halo item holds image data (not zero padding): held tile / tile being loaded
  auto load_item = [&](int j, const Geo& q) {   // (re)load item j for the tile with geometry q
    if (j < HL) {
      const bool ok = ((hvalid >> j) & 1u) && !((q.bad >> (4 * j)) & 0xfull);
      sreg[j] = ld16(xg + (size_t)(q.origin + (ok ? rel_h[j] : 0)) * a.C * sizeof(T) + xbase);    // padding: any valid address
      hin_nxt = (hin_nxt & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    } else {
      sreg[j] = ld16(dyg + (size_t)(q.origin + rel_y0 + (j - HL) * rel_ystep) * a.K * sizeof(T) + ybase);
    }
  };
  auto stage_item = [&](int j, char* dst) {     // transform + store item j of the held tile into buffer dst
    if (j < HL) {
      if (!((hvalid >> j) & 1u)) return;
      u32x4_t v = sreg[j];
      if (xform) {
        float f[EPC];
        Elem<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), relu_lo);
        v = PackH<T>::run(f);
      }
      const bool ok = (hin_cur >> j) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
      const int row = hpl[j];
      st16(dst + YBUF + row * HROW + ((((hchunk >> 1) ^ swz_h(row)) << 5) | ((hchunk & 1) << 4)), v);
    } else {
      const int p = yp0 + 32 * (j - HL);
      st16(dst + p * YROW + ((((ychunk >> 1) ^ swz_y(p)) << 5) | ((ychunk & 1) << 4)), sreg[j]);
    }
  };

  f32x4_t acc[9][2][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][u][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto hpix = [&](int p) {
    if (TW == 16) return (p >> 4) * PITCH + (p & 15);
    return (p >> 6) * (HH * PITCH) + ((p >> 3) & 7) * PITCH + (p & 7);
  };
  // this lane's source pixels within a 32-pixel depth step: pl and pl + 8 (dY and X use the same map; it keeps the 8 pixel
  // rows of a 32-lane transpose read consecutive, which is what the swizzles need)
  const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
  constexpr int HI = (TW == 16 ? 8 : PITCH) * HROW;
  int Aoff[4], Boff[3][2];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) Aoff[t4] = pl * YROW + ((t4 ^ swz_y(pl)) << 5) + (li & 3) * 8;
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) {
    const int hp = hpix(pl) + sx;
#pragma unroll
    for (int u = 0; u < 2; ++u) Boff[sx][u] = YBUF + hp * HROW + (((2 * cp + u) ^ swz_h(hp)) << 5) + (li & 3) * 8;
  }

  // ---- fill: first tile straight into buffer 0, then the registers take the second tile
  {
    const Geo q0 = geom(t_begin);
#pragma unroll
    for (int j = 0; j < NIT; ++j) load_item(j, q0);
    hin_cur = hin_nxt;
    __syncthreads();                            // s_aff
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = s_aff[hchunk * EPC + e]; sh[e] = s_aff[CB + hchunk * EPC + e]; }
#pragma unroll
    for (int j = 0; j < NIT; ++j) stage_item(j, smem);
    if (t_begin + 1 < t_end) {
      const Geo q1 = geom(t_begin + 1);
#pragma unroll
      for (int j = 0; j < NIT; ++j) load_item(j, q1);
    }
    hin_cur = hin_nxt;
    __syncthreads();
  }

  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const bool has1 = tile + 1 < t_end, has2 = tile + 2 < t_end;
    const Geo q2 = geom(has2 ? tile + 2 : tile);
    const char* cb = smem + buf * BUF;
    char* nb = smem + (buf ^ 1) * BUF;
    bf16x8_t af[2][4], bfr[2][2];
    auto afrag = [&](int ab, int q) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const char* pa = cb + q * 32 * YROW + Aoff[t4];
        af[ab][t4] = trp(pa, pa + 8 * YROW);
      }
    };
    auto bfrag = [&](int bb, int q, int t) {
      const int qoff = (hpix(q * 32) + (t / 3) * PITCH) * HROW;
#pragma unroll
      for (int u = 0; u < 2; ++u) bfr[bb][u] = trp(cb + Boff[t % 3][u] + qoff, cb + Boff[t % 3][u] + qoff + HI);
    };
    const int q0 = QPW * qh;
    afrag(0, q0);
    bfrag(0, q0, 0);
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq) {
      const int q = q0 + qq, qn = qq + 1 < QPW ? q + 1 : q;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int slot = qq * 9 + t;            // 9 * QPW (odd * even) slots: fragment buffer parity is slot & 1
        if (t < 8) bfrag((slot + 1) & 1, q, t + 1); else bfrag((slot + 1) & 1, qn, 0);
        if (t == 8) afrag((qq + 1) & 1, qn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
            acc[t][u][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[qq & 1][t4], bfr[slot & 1][u], acc[t][u][t4], 0, 0, 0);
        // item `slot` goes to LDS; the register of the item stored one tap EARLIER is re-loaded (its ds_write has left the
        // register by now -- re-loading the register just stored would make the wave wait for the whole LDS queue,
        // fragment prefetches included)
        if (slot < NIT && has1) stage_item(slot, nb);
        if (slot >= 1 && slot <= NIT && has2) load_item(slot - 1, q2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    hin_cur = hin_nxt;
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + 16 * t4 + 4 * g + j;
          atomicAdd(a.dw + ((size_t)k * 9 + t) * a.C + c0 + 16 * (2 * cp + u) + li, acc[t][u][t4][j]);
        }
}

// 0: not applicable; else the cin block (128 / 64)
int wgrad_p_cb(int dtype, const WgradArgs& a, int tw) {
  if (dtype != DT_BF16 || !tw) return 0;
  static const bool on = getenv("SSLCR_WGRAD_P") != nullptr;       // experimental: slower than wgrad_halo so far
  if (!on) return 0;
  (void)a;
  return 64;          // the 128-channel block (288 accumulators + 16 staging items) does not fit 256 + 256 registers yet
}

template <int TW, int CB>
static hipError_t launch_wp(const WgradArgs& a, hipStream_t st) {
  constexpr int NI = 128 / (8 * TW);
  constexpr int PITCH = TW == 16 ? 24 : 16;
  const int ntiles = (a.N / NI) * (a.H / 8) * (a.W / TW);
  const int kc = (a.K / 64) * (a.C / CB);
  int cus = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  int splits = cdiv(cus, kc);                             // one workgroup per CU
  const int max_splits = cdiv(ntiles, 2);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int tps = cdiv(ntiles, splits);
  splits = cdiv(ntiles, tps);
  const size_t lds = 2 * (size_t)(128 * 128 + NI * 10 * PITCH * CB * 2) + 2 * CB * sizeof(float);
  auto kern = wgrad3x3_p_kernel<TW, CB>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.K / 64, a.C / CB, splits), dim3(256), lds, st, a, tps, ntiles);
  return hipGetLastError();
}

hipError_t launch_wgrad_p(const WgradArgs& a, int tw, int cb, hipStream_t st) {
  (void)cb;
  return tw == 16 ? launch_wp<16, 64>(a, st) : launch_wp<8, 64>(a, st);
}

const char* wgrad_p_name(int tw, int cb) {
  if (tw == 16) return cb == 128 ? "sslcr::wgrad3x3_p_kernel<16, 128>" : "sslcr::wgrad3x3_p_kernel<16, 64>";
  return "sslcr::wgrad3x3_p_kernel<8, 64>";
}

}  // namespace sslcr

// Weight gradient of the 3x3 / stride 2 / pad 1 convs (autograd of layer{2,3}.0.conv1, final_loss.backward() of
// eval_BreastPathQ_SSL_CR.py:98-100), LDS-halo form:
//
//   dW[k][r][s][c] += sum over output pixels (i, j) of dY[i][j][k] * X[2i + r - 1][2j + s - 1][c]
//
// The gather form (conv_wgrad.hip) stages the shifted input once per tap: 36 KB of X per 32 pixels, which is the L2 -> LDS fill
// limit, not the matrix pipe (0.58-0.64 PF/s).  Here a workgroup walks 64-pixel tiles (4 rows x 16 columns of outputs); per tile it
// stages the dY tile (128 kouts) and ONE input region of 9 x 33 pixels, sorted into its four (row, column) parity planes -- within a
// plane the pixels a tap needs for consecutive outputs are consecutive LDS rows (what the transpose read's bank swizzle wants), and
// a tap is a plane + a one-pixel shift in its per-lane address.  38 KB of X per 64 pixels instead of 72.  Everything else is
// wgrad3x3_halo_kernel's: 128(kout) x 64(cin) x 9 register tile over eight waves (wave = cin tile x kout half),
// ds_read_b64_tr_b16 fragments with the B fragment of the next tap requested before this tap's MFMAs, double-buffered LDS with one
// barrier per tile, accumulator slabs + the ordered fold (launch_wgrad_fold: same bits every run).
#include "kernels.hpp"

namespace sslcr {

typedef short w2_h16x4_t __attribute__((ext_vector_type(4)));
typedef short w2_h16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t w2_tr_pair(const char* p0, const char* p1) {
  w2_h16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w2_h16x4_t*)(p0));
  w2_h16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w2_h16x4_t*)(p1));
  w2_h16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// TW = 16: tiles of 4 rows x 16 columns; TW = 8: a tile is 8 rows x 8 columns (layer4.0 at 256x256 input: one whole 8x8 map) -- a depth
// step is then four rows of eight, the lane's second read is the next ROW instead of eight columns on
template <int TW>
__global__ __launch_bounds__(512, 2) void wgrad_s2_kernel(const WgradArgs a, int tiles_per_split, int ntiles, f32x4_t* partials) {
  typedef bf16_t T;
  constexpr int RB = 128;                       // LDS bytes per pixel row (64 channels)
  constexpr int NT = 512;
  constexpr int YH = 64 * RB;                   // one 64-kout half of the dY tile: [64 px][128 B]
  constexpr int YBUF = 2 * YH;
  // planes of the input region (plane pixels, row-major): TW 16: 9 x 33 -> OO 5 x 17, OE 5 x 16, EO 4 x 17, EE 4 x 16;
  // TW 8: 17 x 17 -> OO 9 x 9, OE 9 x 8, EO 8 x 9, EE 8 x 8
  constexpr int TH = 64 / TW, PO = TW + 1, PE = TW;          // tile rows; plane pitches (odd / even columns)
  constexpr int P_OO = 0, P_OE = (TH + 1) * PO, P_EO = P_OE + (TH + 1) * PE, P_EE = P_EO + TH * PO, HP = P_EE + TH * PE;
  static_assert(HP <= 297, "plane buffer");
  constexpr int HBUF = 297 * RB;
  constexpr int BUF = YBUF + HBUF;
  constexpr int YL = 2, HL = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3, kh = tid >> 8;          // cin tile, kout half
  const int li = lane & 15, g = lane >> 4;
  // grid x = (kout block, cin block, pixel split) triples, the gx * gy workgroups of one pixel split neighbours on one XCD (wgrad_halo.hip)
  const int gx = a.K / 128, gy = a.C / 64, GT = gx * gy, splits = (int)gridDim.x / GT;
  int bz, bt;
  if ((splits & 7) == 0) {
    const int w = blockIdx.x, grp = w / (8 * GT), r = w - grp * 8 * GT;
    bz = grp * 8 + (r & 7); bt = r >> 3;
  } else {
    bz = (int)blockIdx.x / GT; bt = (int)blockIdx.x - bz * GT;
  }
  const int by = bt / gx, bx = bt - by * gx;
  const int k0 = bx * 128, c0 = by * 64;
  const int tiles_w = a.OW / TW, tiles_h = a.OH / TH;
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
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* dyg = reinterpret_cast<const char*>(a.dy);

  // bank swizzle of the transpose reads (wgrad_halo.hip): the 32-byte column group is XORed with a key that takes four values over the
  // four same-parity rows of any eight consecutive rows.  dY rows: the pixel index; plane rows: the plane COLUMN (a fragment's eight
  // rows are eight consecutive columns of one plane row), so that a tap's shift is part of a per-lane constant
  auto coff = [](int key, int ch) { return (((ch >> 1) ^ ((key >> 1) & 3)) << 5) | ((ch & 1) << 4); };

  // ---- staging roles, fixed per thread
  const int chunky = tid & 15, prowy = tid >> 4;            // dY: 16-byte chunk of the 128 kouts, pixel row (of 32 per pass)
  const int chunk = tid & 7, prow = tid >> 3;               // X: chunk of the 64 channels, plane pixel (of 64 per pass)
  int rel_y[YL], sty[YL], rel_h[HL], sth[HL];
  unsigned hvalid = 0, htop = 0, hleft = 0;
#pragma unroll
  for (int i = 0; i < YL; ++i) {
    const int p = prowy + 32 * i;
    rel_y[i] = (p / TW) * a.OW + (p % TW);
    sty[i] = (chunky >> 3) * YH + p * RB + coff(p, chunky & 7);
  }
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hp = prow + 64 * i;
    rel_h[i] = 0; sth[i] = 0;
    if (hp < HP) {
      int ridx, cidx, rodd, codd;
      if (hp < P_OE) { ridx = hp / PO; cidx = hp - ridx * PO; rodd = 1; codd = 1; }
      else if (hp < P_EO) { const int q = hp - P_OE; ridx = q / PE; cidx = q - ridx * PE; rodd = 1; codd = 0; }
      else if (hp < P_EE) { const int q = hp - P_EO; ridx = q / PO; cidx = q - ridx * PO; rodd = 0; codd = 1; }
      else { const int q = hp - P_EE; ridx = q / PE; cidx = q - ridx * PE; rodd = 0; codd = 0; }
      rel_h[i] = (2 * ridx - rodd) * a.W + 2 * cidx - codd;         // relative to input pixel (2 h0, 2 w0)
      sth[i] = YBUF + hp * RB + coff(cidx, chunk);
      hvalid |= 1u << i;
      if (rodd && ridx == 0) htop |= 1u << i;
      if (codd && cidx == 0) hleft |= 1u << i;
    }
  }
  const size_t ybase = ((size_t)k0 + chunky * 8) * sizeof(T), xbase = ((size_t)c0 + chunk * 8) * sizeof(T);

  u32x4_t yreg[YL], hreg[HL];
  auto load_regs = [&](int tile) {
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    const int n0 = t / tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int oorg = (n0 * a.OH + h0) * a.OW + w0;                     // output pixel of the tile's (0, 0)
    const int xorg = (n0 * a.H + 2 * h0) * a.W + 2 * w0;               // input pixel (2 h0, 2 w0)
    const unsigned bad = (h0 == 0 ? htop : 0u) | (w0 == 0 ? hleft : 0u);
#pragma unroll
    for (int i = 0; i < YL; ++i) yreg[i] = ld16(dyg + (size_t)(oorg + rel_y[i]) * a.K * sizeof(T) + ybase);
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (((hvalid >> i) & 1u) && !((bad >> i) & 1u)) v = ld16(xg + (size_t)(xorg + rel_h[i]) * a.C * sizeof(T) + xbase);
      hreg[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
    char* b = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < YL; ++i) st16(b + sty[i], yreg[i]);
#pragma unroll
    for (int i = 0; i < HL; ++i)
      if ((hvalid >> i) & 1u) st16(b + sth[i], hreg[i]);
  };

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // this lane's source pixels within a 32-pixel depth step: pl and pl + 8 (wgrad_halo.hip: the eight pixel rows of a 32-lane group
  // are consecutive); pl = output row (pl >> 4) of the step's two, columns 0..7 (+ 8)
  const int pl = 16 * (g >> 1) + 4 * (g & 1) + (li >> 2);
  const int il = pl / TW, jl = pl % TW;                     // TW 16: row 0 / 1 of the step's two, columns 0..7; TW 8: row 0 / 2 of its four
  constexpr int RSTEP = 32 / TW;                            // tile rows per depth step
  int Aoff[4], B17[2], B16;                                 // (B17: the odd-column planes, pitch PO; B16: the even-column ones, pitch PE)
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) Aoff[t4] = pl * RB + ((t4 ^ ((pl >> 1) & 3)) << 5) + (li & 3) * 8;
#pragma unroll
  for (int dc = 0; dc < 2; ++dc) B17[dc] = (il * PO + jl + dc) * RB + ((wave ^ (((jl + dc) >> 1) & 3)) << 5) + (li & 3) * 8;
  B16 = (il * PE + jl) * RB + ((wave ^ ((jl >> 1) & 3)) << 5) + (li & 3) * 8;

  load_regs(t_begin);
  store_lds(0);
  __syncthreads();
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) load_regs(tile + 1);
    const char* yb = smem + buf * BUF;
    const char* hb = yb + YBUF;
    // B fragment of tap t (r = t / 3, s = t % 3) at depth step q: plane by the parities of r, s; plane row RSTEP q + il + (r == 2),
    // column jl + (s == 2)
    auto bfrag_of = [&](int q, int t) {
      const int r = t / 3, s = t - 3 * r;
      const bool rodd = r != 1, codd = s != 1;
      const int dr = r == 2, dc = s == 2;
      const int pbase = (rodd ? (codd ? P_OO : P_OE) : (codd ? P_EO : P_EE)) * RB;
      const int pitch = codd ? PO : PE;
      const char* p = hb + pbase + (codd ? B17[dc] : B16) + (RSTEP * q + dr) * pitch * RB;
      return w2_tr_pair(p, p + (TW == 16 ? 8 : pitch) * RB);        // the lane's pixel pl + 8: eight columns on, or the next row
    };
    bf16x8_t bfr[2];
    auto qstep = [&](const int q, const int par) {       // par: which of bfr[] holds (q, tap 0)
      bf16x8_t af[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const char* pa = yb + kh * YH + q * 32 * RB + Aoff[t4];
        af[t4] = w2_tr_pair(pa, pa + 8 * RB);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        bfr[(t + 1 + par) & 1] = t < 8 ? bfrag_of(q, t + 1) : bfrag_of(q < 1 ? q + 1 : 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          acc[t][t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t4], bfr[(t + par) & 1], acc[t][t4], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    bfr[0] = bfrag_of(0, 0);
    qstep(0, 0);
    qstep(1, 1);
    if (more) store_lds(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  if (partials) {
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

bool wgrad_s2_ok(int dtype, const WgradArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_S2W"); return !e || atoi(e) != 0; }();     // 0: the gather kernel keeps the shape (A/B runs)
  if (!on || dtype != DT_BF16) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 2 || a.pad != 1 || a.in_scale) return false;
  if (a.H != 2 * a.OH || a.W != 2 * a.OW || a.C % 64 != 0 || a.K % 128 != 0) return false;
  if (!((a.OH % 4 == 0 && a.OW % 16 == 0) || (a.OH % 8 == 0 && a.OW == 8))) return false;
  if (a.seg_images > 0 && a.seg_images < a.N) return false;
  return true;
}

hipError_t launch_wgrad_s2(const WgradArgs& a, hipStream_t st) {
  const bool w8 = a.OW % 16 != 0;
  const int ntiles = w8 ? a.N * (a.OH / 8) : a.N * (a.OH / 4) * (a.OW / 16);
  const int kc = (a.K / 128) * (a.C / 64);
  const int cus = device_cus();
  int splits = cdiv(cus, kc);                            // one 8-wave workgroup per CU
  const int max_splits = cdiv(ntiles, 16);                // at least 16 tiles per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int tps = cdiv(ntiles, splits);
  splits = cdiv(ntiles, tps);
  const size_t lds = 2 * (2 * 64 * 128 + 297 * 128);
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_s2_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_s2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int gx = a.K / 128, gy = a.C / 64;
  f32x4_t* slabs = nullptr;
  if (splits > 1) {
    slabs = reinterpret_cast<f32x4_t*>(stream_scratch(st, (size_t)gx * gy * splits * 36 * 512 * sizeof(f32x4_t)));
    if (!slabs) return hipErrorOutOfMemory;
  }
  if (w8) hipLaunchKernelGGL(wgrad_s2_kernel<8>, dim3(gx * gy * splits), dim3(512), lds, st, a, tps, ntiles, slabs);
  else hipLaunchKernelGGL(wgrad_s2_kernel<16>, dim3(gx * gy * splits), dim3(512), lds, st, a, tps, ntiles, slabs);
  if (slabs) return launch_wgrad_fold(slabs, a.dw, a.C, gx, gy, splits, 9, 2, st);
  return hipGetLastError();
}

}  // namespace sslcr

// Gather-form NHWC convolution fed entirely by global->LDS DMA: the stride-2 3x3 convs of ResNet18 (layer2.0/3.0/4.0 conv1),
// their dgrads (four output-parity classes, each a tap subset -- see sslcr_conv_desc.pix_mul / tap_mask) and the 1x1
// stride-2 downsample convs.  None of these has a producer BatchNorm in its load path (their input is a materialised block
// output), so neither operand needs registers on the way to LDS:
//
//   * per step (one tap, one 128-byte channel slab) each lane issues an LDS DMA (buffer_load_dwordx4 ... lds) for its share of the BP pixel
//     rows and BKO weight rows.  The lane picks its SOURCE 16-byte chunk so that the linear DMA placement is the XOR-
//     swizzled (and, for weights, fragment-ordered) tile; a pixel that falls into the zero padding (or past M) gets an
//     offset beyond the buffer resource's range, which reads zeros -- no branches around memory operations, no ds_write, no staging VGPRs;
//   * step j+1 is requested before the MFMAs of step j and waited for (vmcnt(0)) right before the barrier that ends
//     step j; scheduler fences keep the requests where they are written (the compiler otherwise sinks them to the wait);
//   * 256 threads, each wave a 64 px x 64 kout register tile (16 MFMA per 8 fragment reads), two workgroups per CU so one
//     workgroup's barrier/epilogue phases sit under the other's MFMAs.
//
// Replaces conv_igemm_kernel for these shapes (274-370 TF/s forward at N=640 before).  Epilogue conventions (bias /
// residual / ReLU / accumulate / strided scatter / per-channel (sum, sumsq) partial rows) are those of conv_igemm.hip.
#include "kernels.hpp"

namespace sslcr {

template <typename T> struct MmaD;
template <> struct MmaD<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaD<float> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

template <typename T, int BP, int BKO>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(const ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CE = 8 * EPC;                 // channels per 128-byte row = K depth of a step
  constexpr int WP = BP / 64, WKN = 4 / WP;   // waves along pixels / kouts
  static_assert(BKO == 64 * WKN, "each wave owns a 64 px x 64 kout tile");
  constexpr int TK = 4, TP = 4;
  constexpr int BB = BP * 128, AB = BKO * 128, STG = BB + AB;
  constexpr int PLD = BP * 8 / 256, WLD = BKO * 8 / 256;     // DMA instructions per thread per step
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wp = wave % WP, wk = wave / WP;
  // grid x = (pixel block, kout block) pairs; workgroups land on XCD (linear id % 8), each with its own L2.  The KB kout blocks of
  // one pixel block read the same input rows, so they are made neighbours ON ONE XCD (ids w, w + 8, ...): KB - 1 of the KB reads hit
  // that L2 instead of going out to the fabric (r03: layer4.0.conv1 fetched its 84 MB input four times).
  const int KB = a.K / BKO, PB = (int)gridDim.x / KB;
  int pb, kb_i;
  if ((PB & 7) == 0) {
    const int w = blockIdx.x, grp = w / (8 * KB), r = w - grp * 8 * KB;
    pb = grp * 8 + (r & 7); kb_i = r >> 3;
  } else {
    kb_i = (int)blockIdx.x / PB; pb = (int)blockIdx.x - kb_i * PB;
  }
  const int m0 = pb * BP, k0 = kb_i * BKO;
  const int PHW = a.PH * a.PW;
  const int M = a.N * PHW;
  const int pmul = a.pix_mul ? a.pix_mul : 1;
  const int RS = a.R * a.S;
  const int cslabs = a.C / CE;
  // par4: the four output-parity classes of a stride-2 3x3 dgrad in one launch, heaviest class (4 taps) first
  int off_h = a.pix_off_h, off_w = a.pix_off_w;
  unsigned tm4 = 0;
  if (a.par4) {
    const int cls = 3 - (int)blockIdx.z;
    off_h = cls >> 1; off_w = cls & 1;
    const unsigned mr = off_h ? 5u : 2u, ms = off_w ? 5u : 2u;      // rows / columns r with (off + 1 - r) even
    tm4 = ((mr & 1u) ? ms : 0u) | ((mr & 2u) ? ms << 3 : 0u) | ((mr & 4u) ? ms << 6 : 0u);
  }
  const unsigned tmask = a.par4 ? tm4 : (a.tap_mask ? a.tap_mask : ((1u << RS) - 1u));
  const int nsteps = __builtin_popcount(tmask) * cslabs;

  // ---- DMA roles: instruction i of this wave fills LDS rows [i*32 + wave*8, +8); lane -> (row, 16-byte slot).
  // A lane keeps the base coordinates (h0, w0) of its pixel and the ADDRESS of that (possibly out-of-image) position; a tap
  // then adds a displacement (dh, dw) that is uniform over the workgroup -- one 64-bit add and two range checks per DMA
  // instead of the per-lane multiply chain (r03: this kernel ran ~6 VALU instructions per MFMA, most of them addresses and
  // integer divisions).  Transposed stride 2 needs a uniform pixel parity for that: pix_mul even (conv_dma_bp).
  const float rcp_phw = 1.f / (float)PHW, rcp_pw = 1.f / (float)a.PW;
  auto divmod = [](int x, int d, float rcp, int& q, int& r) {      // reciprocal estimate + one correction
    q = (int)((float)x * rcp);
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
  };
  const int tsh = (a.transposed && a.stride == 2) ? 1 : 0;
  const int e_h = (off_h + a.pad) & 1, e_w = (off_w + a.pad) & 1;      // transposed stride 2: parity of (p + pad)
  const char* xg = reinterpret_cast<const char*>(a.x);
  const int slot = lane & 7;
  int h0[PLD], w0[PLD];
  // Both operands come in by the MUBUF form of the DMA (LdsDma, common.hpp: behind global_load_lds every fragment wait of the step
  // was lgkmcnt(0)).  The input's resource starts at the FIRST IMAGE of this workgroup's pixel block, so a lane's byte offset fits
  // 32 bits whatever the batch (config 5's layer2.0 input is 2.3 GB); a position in the zero padding (or past M) gets an offset
  // beyond the resource's range and reads zeros -- the range check is the padding page.
  const int n_first = m0 / PHW;
  const size_t img_bytes = (size_t)a.H * a.W * a.C * sizeof(T);
  const size_t left = (size_t)(a.N - n_first) * img_bytes;
  LdsDma xdma, wdma;
  xdma.init(xg + (size_t)n_first * img_bytes, left < 0x80000000ull ? (unsigned)left : 0x80000000u);
  constexpr int OOR = (int)0xfffffff0u;
  int rowoff[PLD];
#pragma unroll
  for (int i = 0; i < PLD; ++i) {
    const int rr = i * 32 + wave * 8 + (lane >> 3);
    const int m = m0 + rr;
    const int pc16 = ((slot ^ (rr & 7)) * EPC) * (int)sizeof(T);
    if (m < M) {
      int n, rem, ph, pw;
      divmod(m, PHW, rcp_phw, n, rem);
      divmod(rem, a.PW, rcp_pw, ph, pw);
      ph = ph * pmul + off_h; pw = pw * pmul + off_w;
      h0[i] = a.transposed ? (ph + a.pad) >> tsh : ph * a.stride - a.pad;
      w0[i] = a.transposed ? (pw + a.pad) >> tsh : pw * a.stride - a.pad;
      rowoff[i] = (int)(((long)(n - n_first) * a.H * a.W + (long)h0[i] * a.W + w0[i]) * (long)(a.C * (int)sizeof(T))) + pc16;
    } else {
      h0[i] = -(1 << 24); w0[i] = 0; rowoff[i] = 0;          // never in range
    }
  }
  int wsrc[WLD];
#pragma unroll
  for (int i = 0; i < WLD; ++i) {
    const int rr = i * 32 + wave * 8 + (lane >> 3);
    // inverse of wperm<TK>: LDS row t*16 + q*4 + j of a 64-row block holds kout row q*16 + t*4 + j
    const int blk = rr >> 6, x = rr & 63;
    const int krow = blk * 64 + ((x >> 2) & 3) * 16 + (x >> 4) * 4 + (x & 3);
    wsrc[i] = (int)((((size_t)(k0 + krow) * RS) * a.C + (slot ^ (rr & 7)) * EPC) * sizeof(T));
  }
  wdma.init(a.w, (unsigned)((size_t)a.K * RS * a.C * sizeof(T)));

  int it_tap = __builtin_ctz(tmask), it_slab = 0;      // issue() is called for steps 0,1,2,... in order
  auto issue = [&](int stage) {
    const int tap = it_tap, c0 = it_slab * CE;
    if (++it_slab == cslabs) {
      it_slab = 0;
      do { ++it_tap; } while (it_tap < RS && !((tmask >> it_tap) & 1u));
    }
    int r = 0, s = tap;                          // (uniform; a subtract loop instead of a division per step)
    while (s >= a.S) { s -= a.S; ++r; }
    // the tap's displacement from (h0, w0): r, s forward; -r, -s transposed; (parity - r) / 2 where that divides, stride 2
    const int th = e_h - r, tw = e_w - s;
    const int dh = a.transposed ? (tsh ? th >> 1 : -r) : r, dw = a.transposed ? (tsh ? tw >> 1 : -s) : s;
    const bool tap_ok = !tsh || ((th | tw) & 1) == 0;
    const int uoff = ((dh * a.W + dw) * a.C + c0) * (int)sizeof(T);
    char* sb = smem + stage * STG;
#pragma unroll
    for (int i = 0; i < PLD; ++i) {
      const bool ok = tap_ok && (unsigned)(h0[i] + dh) < (unsigned)a.H && (unsigned)(w0[i] + dw) < (unsigned)a.W;
      xdma.load16(sb + (i * 32 + wave * 8) * 128, ok ? rowoff[i] + uoff : OOR, 0);
    }
    const int wtap = (tap * a.C + c0) * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < WLD; ++i) wdma.load16(sb + BB + (i * 32 + wave * 8) * 128, wsrc[i], wtap);
  };

  // fragment addresses: per-lane base (row & 7 == li & 7 for every fragment row) + immediates
  int Ab[2], Bb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ci = kk * 4 + g;
    Ab[kk] = BB + (wk * 64 + li) * 128 + ((ci ^ (li & 7)) << 4);
    Bb[kk] = (wp * 64 + li) * 128 + ((ci ^ (li & 7)) << 4);
  }

  f32x4_t acc[TK][TP];
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[t][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue(0);
  __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) issue((step + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const char* sb = smem + (step & 1) * STG;
    u32x4_t A[2][TK], B[2][TP];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int t = 0; t < TK; ++t) A[kk][t] = ld16(sb + Ab[kk] + t * 2048);
#pragma unroll
      for (int p = 0; p < TP; ++p) B[kk][p] = ld16(sb + Bb[kk] + p * 2048);
      if (kk == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int p = 0; p < TP; ++p) MmaD<T>::run(A[kk][t], B[kk][p], acc[t][p]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }

  // ---------------- epilogue: lane (li = pixel within 16-tile, g) holds kouts kb .. kb+15 of its pixels
  const int kb = k0 + wk * 64 + g * (4 * TK);
  if (a.out_scale) conv_scale_acc<TK, TP>(acc, a.out_scale + kb);
  float bias[4 * TK];
  if (a.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(a.bias + kb + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) bias[4 * q + j] = b4[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4 * TK; ++j) bias[j] = 0.f;
  }
  char* yg = reinterpret_cast<char*>(a.y);
  const char* rg = reinterpret_cast<const char*>(a.residual);
  size_t off[TP];
  bool ok[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int mm = m0 + wp * 64 + p * 16 + li;
    ok[p] = mm < M;
    const int m = ok[p] ? mm : M - 1;          // a valid pixel for the (unconditional) operand loads
    int n, rem, ph, pw;
    divmod(m, PHW, rcp_phw, n, rem);
    divmod(rem, a.PW, rcp_pw, ph, pw);
    ph = ph * pmul + off_h; pw = pw * pmul + off_w;
    const size_t opix = ((size_t)n * a.OH + (size_t)ph * a.osh) * a.OW + (size_t)pw * a.osh;
    off[p] = (opix * a.K + kb) * sizeof(T);
  }
  if (a.bias) conv_store_tile<T, TK, TP>(acc, bias, off, ok, yg, rg, a.accumulate != 0, a.relu != 0);
  else conv_store_tile_nobias<T, TK, TP>(acc, bias, off, ok, yg, rg, a.accumulate != 0, a.relu != 0);

  if (a.stats) {
    // rows >= M were staged as zeros -> contribute 0.  Sum over this wave's 64 pixels.
    float s1[4 * TK], s2[4 * TK];
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
    float* sp = a.stats + ((size_t)(pb * WP + wp) * 2) * a.K + kb;
    if constexpr (4 * TK == 16) {
      // sixteen values over the 16 lanes of the DPP row in 32 instructions (row16_fold16, common.hpp): quad q of the row ends up with
      // values 4q..4q+3 and stores that 16-byte piece
      row16_fold16(s1);
      row16_fold16(s2);
      if ((li & 3) == 0) {
        const int q4 = (li >> 2) * 4;
        *reinterpret_cast<f32x4_t*>(sp + q4) = f32x4_t{s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4_t*>(sp + a.K + q4) = f32x4_t{s2[0], s2[1], s2[2], s2[3]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4 * TK; j += 4) {      // DPP folded into the adds: half the instructions of row16_sum()
        row16_sum4(s1[j], s1[j + 1], s1[j + 2], s1[j + 3]);
        row16_sum4(s2[j], s2[j + 1], s2[j + 2], s2[j + 3]);
      }
      if (li == 0) {
#pragma unroll
        for (int j = 0; j < 4 * TK; ++j) { sp[j] = s1[j]; sp[a.K + j] = s2[j]; }
      }
    }
  }
}

// 0: not applicable; else the pixel tile (128: 128 px x 128 kout, 256: 256 px x 64 kout)
int conv_dma_bp(int dtype, const ConvArgs& a) {
  if (a.in_scale != nullptr) return 0;
  if (a.transposed && a.stride != 1 && a.stride != 2) return 0;
  if (a.transposed && a.stride == 2 && ((a.pix_mul ? a.pix_mul : 1) & 1)) return 0;      // the kernel wants one pixel parity per workgroup (pix_mul 0 = 1)
  const int ce = dtype == DT_BF16 ? 64 : 32;
  if (a.C % ce != 0 || a.K % 64 != 0 || a.R * a.S > 31) return 0;
  const long M = (long)a.N * a.PH * a.PW;
  if (M < 2048) return 0;                        // tiny problems stay on the 64x64-tile kernel
  // the input's buffer resource starts at a workgroup's first image and its lane offsets are 32-bit: the images one pixel block can
  // span (256 pixels at most, plus the partial images at both ends) must stay below 2 GB
  if (((long)(256 / (a.PH * a.PW)) + 2) * a.H * a.W * a.C * (dtype == DT_BF16 ? 2 : 4) >= (1l << 31)) return 0;
  return a.K % 128 == 0 ? 128 : 256;
}
int conv_dma_rows(const ConvArgs& a, int bp) {
  const int M = a.N * a.PH * a.PW;
  return cdiv(M, bp) * (bp / 64);
}

template <typename T, int BP, int BKO>
static hipError_t launch_d(const ConvArgs& a, hipStream_t st) {
  const int M = a.N * a.PH * a.PW;
  const size_t lds = 2 * (BP + BKO) * 128;
  auto kern = conv_dma_kernel<T, BP, BKO>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (a.par4 && (!a.transposed || a.stride != 2 || a.pix_mul != 2 || a.R != 3 || a.S != 3 || a.pad != 1 || a.stats)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(cdiv(M, BP) * (a.K / BKO), 1, a.par4 ? 4 : 1), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_conv_dma(int dtype, const ConvArgs& a, int bp, hipStream_t st) {
  if (dtype == DT_BF16) return bp == 128 ? launch_d<bf16_t, 128, 128>(a, st) : launch_d<bf16_t, 256, 64>(a, st);
  return bp == 128 ? launch_d<float, 128, 128>(a, st) : launch_d<float, 256, 64>(a, st);
}

const char* conv_dma_name(int dtype, int bp) {
  if (dtype == DT_BF16) return bp == 128 ? "sslcr::conv_dma_kernel<unsigned short, 128, 128>" : "sslcr::conv_dma_kernel<unsigned short, 256, 64>";
  return bp == 128 ? "sslcr::conv_dma_kernel<float, 128, 128>" : "sslcr::conv_dma_kernel<float, 256, 64>";
}

}  // namespace sslcr

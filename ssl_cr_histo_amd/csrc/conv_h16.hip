// 3x3 / stride 1 / pad 1 NHWC convolution on 256-pixel tiles -- a 16x16 patch of one image, or (TW = 8) four whole 8x8 images:
// persistent, DMA-fed form of the LDS-halo kernel (ResNet18 layer1-4 block convs forward and their dgrads through tap-flipped
// packs; conv_halo256.hip is the fallback for the shapes this one does not take).
//
// What the previous form lost, measured by ablation on the layer2 shape (N=640, 32x32, 128->128: 220 us, MFMA-only ~95 us):
// removing the MFMAs left 137 us, the halo loads 47 us, the weight loads 50 us, the epilogue 52 us, the barriers 31 us --
// i.e. the parts ran back to back instead of under each other.  The ISA showed why: (1) the compiler sank every weight
// prefetch down to its LDS store, so each tap waited a full L2 round trip; (2) s_waitcnt vmcnt counts IN ORDER, so the first
// wait on a (young, short) weight load also waited for the (older, long) HBM halo loads and for the epilogue stores of the
// previous tile; (3) ~55 VGPRs held per-tap LDS addresses, leaving no room to double-buffer fragments, so the wave ran
// ds_read -> wait -> 4 MFMA -> ds_read ...
//
// This kernel is built around those three facts:
//   * weights go global -> LDS by DMA (buffer_load_dwordx4 ... lds -- LdsDma, common.hpp; no registers, no ds_write; issued by the
//     younger wave of each SIMD's pair): a whole ring half (3 taps) is
//     issued right after the barrier that frees it and waited for once, just before the barrier that publishes it, three
//     taps later.  The lane picks its SOURCE chunk so that the linear DMA placement is the swizzled, fragment-ordered tile.
//   * the halo of the next stage is requested after the first weight wait of a stage, so the only vmcnt waits in the
//     stream are >= 3 taps (~3000 clk) behind every load and store they cover; BatchNorm+ReLU of the producer is applied to
//     the halo IN REGISTERS under the MFMAs of the last three taps, and only six ds_write_b128 sit between the two barriers
//     of a stage boundary.
//   * the halo rows are pitched 24 pixels (18 used) so that the XOR swizzle key (pixel & 7) depends on the lane and the
//     filter column only: all nine taps, four pixel groups and TK weight tiles are immediate offsets on 6 + 2 address
//     registers, and fragments are double-buffered in the registers this frees.
//   * persistent workgroups (one per CU) walk items (tile, kout block) b, b+G, ...: the epilogue's stores drain under the
//     next item's taps and the next item's first halo is already in LDS when the last tap retires.
#include "kernels.hpp"

namespace sslcr {

template <typename T> struct MmaH;
template <> struct MmaH<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaH<float> {
  __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// inverse of wperm<TK> (common.hpp): LDS row -> kout row of the block
template <int TK>
__device__ __forceinline__ int wperm_inv(int rr) {
  constexpr int B = 16 * TK;
  const int blk = rr / B, x = rr - blk * B;
  const int t = x >> 4, q = (x >> 2) & 3, j = x & 3;
  return blk * B + q * (4 * TK) + t * 4 + j;
}

#define SSLCR_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0f70) /* vmcnt(0), lgkmcnt/expcnt untouched */
// A barrier WITHOUT the workgroup fence of __syncthreads() (which is s_waitcnt lgkmcnt(0) first -- a wait for the fragment read
// issued two MFMAs earlier).  Enough wherever no ds_write is pending and what the barrier orders are (a) this wave's completed DMA
// (explicit vmcnt(0) in front) or (b) fragment reads whose MFMAs have already been issued, i.e. whose data has arrived.  The
// "memory" clobber keeps the compiler from moving LDS accesses across it.
#define SSLCR_BARE_BARRIER() asm volatile("s_barrier" ::: "memory")

// phase timing for tools/microbench/h16_phase_bench.hip (-DSSLCR_H16_PROF; compiled out otherwise): per wave of workgroup 0, shader
// cycles of a stage spent waiting for the weight DMA, at the publish (P) and free (F) barriers, in the stage-end halo swap and in
// the item epilogue
#ifdef SSLCR_H16_PROF
__device__ unsigned long long g_h16_prof[16][8];
#define H16_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define H16_ACC(i, d) h16_t[i] += (d)
#else
#define H16_T(v)
#define H16_ACC(i, d)
#endif

// XF: the producer's BatchNorm(+ReLU) is applied to the input on its way into LDS (a.in_scale != nullptr)
// WR: the whole filter bank stays resident in LDS (C == one slab and K == BKO, i.e. the 64->64 layer1 convs: 9 x 64 x 128 B
//     = 72 KB next to the 54 KB halo).  A stage then has no weight DMA, no publish/free barriers and no vmcnt wait before its
//     15th step, so the next halo (and the residual) is requested at the TOP of the stage and has ~14 steps to land; with
//     the ring, the short 8-MFMA steps of this shape left the HBM round trip of the halo half exposed and paid 8 barriers
//     per 144 MFMAs.
// RAW: the output stage of the train-mode forward -- no bias, no residual, no ReLU, no mask, statistics wanted (checked by the
//     launcher): pack + store + the BatchNorm partial sums, as an instance of its own.  An output-stage instruction costs ~4 cycles
//     (no MFMA runs beside it), so the 64 bias adds and 64 clamps of the general body are worth an instance; as run-time cases
//     INSIDE one instance the duplicated bodies spilled (round 3).
// TW = 8 (round 4): the tile is FOUR WHOLE 8x8 IMAGES (ResNet18 layer4 at 256x256 input) instead of a 16x16 patch of one -- the same
//     256 pixels x BKO kouts, wave wp owns image wp, a fragment's 16 lanes are two image rows.  Every halo-ring pixel is padding, so
//     the ring is zeroed once and a stage stages the 256 interior pixels only (4 loads per thread, no edge logic); halo rows are
//     pitched 10 pixels with the swizzle key = halo column & 7 (conflict-free over the lane groups of ds_read_b128, enumerated for
//     conv3x3_halo256's 8-wide form).  row0: first statistics row of this launch (a shape served by two launches, see launch_ht8).
// OSC (round 6): sslcr_conv_desc.out_scale -- eval-mode BatchNorm with its scale kept out of the filters, y = epilogue(acc * scale + bias).
//     An instance (and a kernel name, conv3x3_h16s_kernel) of its own: as a run-time case inside the plain instance the sixteen scale
//     values took the dominant instance of the step from 251 registers to 256 + 120 B of scratch (all of its launches, the dgrads too).
//     Its output stage walks 16-byte chunks with the chunk's bias and scale loaded from LDS next to each other -- the same live set as
//     the plain body's bias[16] + v[16].
template <typename T, int BKO, int WK, bool XF, bool WR, bool RAW = false, int TW = 16>
__global__ __launch_bounds__(256 * WK, WK == 1 ? 1 : 2) void conv3x3_h16_kernel(const ConvArgs a, const int tiles_total, const int n_items, const int kshift,
                                                                                const int row0) {
  constexpr bool OSC = false;
#include "conv_h16_body.hpp"
}
// ... with sslcr_conv_desc.out_scale (no input transform, no statistics: the eval forms)
template <typename T, int BKO, int WK, bool WR, int TW>
__global__ __launch_bounds__(256 * WK, WK == 1 ? 1 : 2) void conv3x3_h16s_kernel(const ConvArgs a, const int tiles_total, const int n_items, const int kshift,
                                                                                 const int row0) {
  constexpr bool OSC = true, XF = false, RAW = false;
#include "conv_h16_body.hpp"
}

// 16: 16x16 tiles of one image; 8: four whole 8x8 images per tile (128-kout blocks: ResNet18 layer4 at 256x256 input); 0: not served.
// The same answer for both dtypes (sslcr_conv2d_partial_rows has no dtype: the launches must tile identically).
static int h16_mode(int dtype, const ConvArgs& a) {
  const int q = conv_halo256_mode(dtype, a);
  if (q == 16 && conv_halo256_mode(DT_BF16, a) == 16) return 16;
  static const bool on8 = [] { const char* e = getenv("SSLCR_H16_TW8"); return !e || atoi(e) != 0; }();   // 0: conv3x3_halo256 keeps the shape (A/B runs)
  if (q == 8 && on8 && conv_halo256_mode(DT_BF16, a) == 8 && a.K % 128 == 0 && !a.mask_x && (a.seg_images <= 0 || a.seg_images % 4 == 0)) return 8;
  return 0;
}
bool conv_h16_ok(int dtype, const ConvArgs& a) {
  if (a.in_scale && a.residual) return false;          // not a ResNet combination; the older halo kernels take it
  if (a.mask_x) {
    // (a.stats is checked at launch: sslcr_conv2d_partial_rows asks before the rows buffer exists)
    if (!a.mask_scale || !a.mask_shift || !a.mask_mean || a.in_scale || a.bias || a.residual || a.relu || a.out_scale) return false;
    if (18 * 24 * 128 + 2 * 3 * (a.K % 128 == 0 ? 128 : 64) * 128 + 2 * a.C * 4 + 8 * 128 * 4 + 3 * a.K * 4 > 160 * 1024) return false;
  }
  if (a.out_scale) {
    // one more K-float array in LDS (the input-transform and train-forward instances have no output scale: bias forms only)
    if (a.in_scale || a.stats || !a.bias) return false;
    const int mode = h16_mode(dtype, a);
    if (mode == 0) return false;
    const int bko = a.K % 128 == 0 ? 128 : 64;
    if (!(a.C == 64 && a.K == 64) &&
        (mode == 16 ? 18 * 24 : 4 * 10 * 10) * 128 + 2 * 3 * bko * 128 + 2 * a.C * 4 + 8 * bko * 4 + 2 * a.K * 4 > 160 * 1024) return false;
  }
  return h16_mode(dtype, a) != 0;
}
static int h16_tiles(const ConvArgs& a, int nseg) {                     // per segment
  return a.H == 8 ? (a.N / nseg) / 4 : (a.N / nseg) * (a.H / 16) * (a.W / 16);
}
// partial-statistics rows the launch will write: four per workgroup (see s_stat)
// workgroups of the launch: one per CU, or per item where there are fewer; with segments, nseg equal groups
static int h16_grid(const ConvArgs& a, int bko) {
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int n_items = h16_tiles(a, nseg) * (a.K / bko);       // per segment
  const int per = device_cus() / nseg;
  return (n_items < per ? n_items : per) * nseg;
}
// Four-image tiles (layer4): with one workgroup per CU walking (tile, 128-kout block) items, a last round that would occupy at most
// half the CUs (N = 640: 640 items = 2.5 rounds of 256) runs its tiles as 64-kout items on all of them instead, in a second launch
// over the tail images (the split conv3x3_halo256 makes for this shape).  -> tiles of the tail launch (0: one launch)
static int h16_tail8(const ConvArgs& a) {
  if (a.H != 8 || a.seg_images > 0) return 0;
  const int cus = device_cus(), tiles = a.N / 4, kb = a.K / 128, rem = (tiles * kb) % cus;
  if (rem == 0 || 2 * rem > cus || rem % kb != 0 || tiles <= rem / kb) return 0;
  return rem / kb;
}
static ConvArgs h16_head8(const ConvArgs& a, int tail) { ConvArgs h = a; h.N = a.N - 4 * tail; return h; }
static ConvArgs h16_tailargs8(const ConvArgs& a, int tail, int dtype) {
  ConvArgs t = a;
  const int n0 = a.N - 4 * tail;
  const size_t es = dtype == DT_BF16 ? 2 : 4, px = (size_t)n0 * a.H * a.W;
  t.N = 4 * tail;
  t.x = reinterpret_cast<const char*>(a.x) + px * a.C * es;
  t.y = reinterpret_cast<char*>(a.y) + px * a.K * es;
  if (a.residual) t.residual = reinterpret_cast<const char*>(a.residual) + px * a.K * es;
  return t;
}
int conv_h16_rows(const ConvArgs& a) {
  if (a.seg_images > 0 && conv_pp64_ok(DT_BF16, a)) return conv_pp64_rows(a);      // segments are bf16-only: the ping-pong kernel's own grid
  const int tail = h16_tail8(a);
  if (tail) return (h16_grid(h16_head8(a, tail), 128) + h16_grid(h16_tailargs8(a, tail, DT_BF16), 64)) * 4;
  return h16_grid(a, a.K % 128 == 0 ? 128 : 64) * 4;
}

// bf16 64 -> 64: one 128-byte slab of input channels and one kout block, the filter bank fits LDS whole
static bool h16_resident(const ConvArgs& a) { return a.C == 64 && a.K == 64; }
// the train-mode forward's output stage (the RAW instance); SSLCR_H16_RAW=0 keeps the general body for same-box A/B runs
static bool h16_raw(const ConvArgs& a) {
  static const bool on = [] { const char* e = getenv("SSLCR_H16_RAW"); return !e || atoi(e) != 0; }();
  return on && a.stats && !a.bias && !a.relu && !a.residual && !a.mask_x;
}

template <typename T, int BKO, int WK, bool XF, bool WR = false, bool RAW = false, int TW = 16, bool OSC = false>
static hipError_t launch_h(const ConvArgs& a, hipStream_t st, int row0 = 0) {
  const size_t lds = (TW == 16 ? 18 * 24 : 4 * 10 * 10) * 128 + (WR ? 3 : 2) * 3 * BKO * 128 + 2 * a.C * sizeof(float) + 8 * BKO * sizeof(float) +
                     (a.mask_x ? 3 : (a.out_scale ? 2 : 1)) * a.K * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static_assert(!OSC || (!XF && !RAW), "output scale: eval forms only");
  void (*kern)(const ConvArgs, const int, const int, const int, const int);
  if constexpr (OSC) kern = conv3x3_h16s_kernel<T, BKO, WK, WR, TW>;
  else kern = conv3x3_h16_kernel<T, BKO, WK, XF, WR, RAW, TW>;
  if (OSC != (a.out_scale != nullptr)) return hipErrorInvalidValue;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int nseg = a.seg_images > 0 ? a.N / a.seg_images : 1;
  const int tiles = h16_tiles(a, nseg);                                 // per segment, like n_items
  const int n_items = tiles * (a.K / BKO);
  const int grid = h16_grid(a, BKO);                                    // one 8-wave workgroup per CU
  const int kbn = a.K / BKO, gseg = grid / nseg;
  const int kshift = (kbn > 1 && (kbn & (kbn - 1)) == 0 && (gseg & (kbn - 1)) == 0) ? __builtin_ctz(kbn) : -1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * WK), lds, st, a, tiles, n_items, kshift, row0);
  return hipGetLastError();
}

// four-image tiles, K % 128 == 0 (h16_mode == 8)
template <typename T>
static hipError_t launch_ht8(const ConvArgs& a, hipStream_t st) {
  const bool xf = a.in_scale != nullptr;
  auto wide = [&](const ConvArgs& q) {
    if (q.out_scale) return launch_h<T, 128, 2, false, false, false, 8, true>(q, st);
    if constexpr (sizeof(T) == 2)
      if (h16_raw(q)) return xf ? launch_h<T, 128, 2, true, false, true, 8>(q, st) : launch_h<T, 128, 2, false, false, true, 8>(q, st);
    return xf ? launch_h<T, 128, 2, true, false, false, 8>(q, st) : launch_h<T, 128, 2, false, false, false, 8>(q, st);
  };
  const int tail = h16_tail8(a);
  if (!tail) return wide(a);
  const ConvArgs head = h16_head8(a, tail), tl = h16_tailargs8(a, tail, Elem<T>::DT);
  hipError_t e = wide(head);
  if (e != hipSuccess) return e;
  const int row0 = h16_grid(head, 128) * 4;                             // the tail's statistics rows follow the head's
  if (tl.out_scale) return launch_h<T, 64, 2, false, false, false, 8, true>(tl, st, row0);
  return xf ? launch_h<T, 64, 2, true, false, false, 8>(tl, st, row0) : launch_h<T, 64, 2, false, false, false, 8>(tl, st, row0);
}

template <typename T>
static hipError_t launch_ht(const ConvArgs& a, hipStream_t st) {
  const bool xf = a.in_scale != nullptr;
  if (h16_mode(Elem<T>::DT, a) == 8) return launch_ht8<T>(a, st);
  if (a.K % 128 == 0) {
    if (a.out_scale) return launch_h<T, 128, 2, false, false, false, 16, true>(a, st);
    if constexpr (sizeof(T) == 2)
      if (h16_raw(a)) return xf ? launch_h<T, 128, 2, true, false, true>(a, st) : launch_h<T, 128, 2, false, false, true>(a, st);
    return xf ? launch_h<T, 128, 2, true>(a, st) : launch_h<T, 128, 2, false>(a, st);
  }
  if constexpr (sizeof(T) == 2)
    if (h16_resident(a)) {
      if (conv_pp64_ok(DT_BF16, a)) return launch_conv_pp64(a, st);        // ping-pong form (conv_pp64.hip)
      if (a.out_scale) return launch_h<T, 64, 2, false, true, false, 16, true>(a, st);
      return xf ? launch_h<T, 64, 2, true, true>(a, st) : launch_h<T, 64, 2, false, true>(a, st);
    }
  if (a.out_scale) return launch_h<T, 64, 2, false, false, false, 16, true>(a, st);
  return xf ? launch_h<T, 64, 2, true>(a, st) : launch_h<T, 64, 2, false>(a, st);
}

hipError_t launch_conv_h16(int dtype, const ConvArgs& a, hipStream_t st) {
  if (a.mask_x && !a.stats) return hipErrorInvalidValue;
  return dtype == DT_BF16 ? launch_ht<bf16_t>(a, st) : launch_ht<float>(a, st);
}

// (the profiler's name of the launch -- of its first, 128-kout launch where a four-image shape takes two)
const char* conv_h16_name(int dtype, const ConvArgs& a) {
  const bool bf = dtype == DT_BF16, xf = a.in_scale != nullptr;
  if (a.out_scale) {                          // conv3x3_h16s_kernel<T, BKO, WK, WR, TW>
    if (h16_mode(dtype, a) == 8) return bf ? "sslcr::conv3x3_h16s_kernel<unsigned short, 128, 2, false, 8>" : "sslcr::conv3x3_h16s_kernel<float, 128, 2, false, 8>";
    if (a.K % 128 == 0) return bf ? "sslcr::conv3x3_h16s_kernel<unsigned short, 128, 2, false, 16>" : "sslcr::conv3x3_h16s_kernel<float, 128, 2, false, 16>";
    if (bf && h16_resident(a) && conv_pp64_ok(DT_BF16, a)) return conv_pp64_name(a);
    if (bf && h16_resident(a)) return "sslcr::conv3x3_h16s_kernel<unsigned short, 64, 2, true, 16>";
    return bf ? "sslcr::conv3x3_h16s_kernel<unsigned short, 64, 2, false, 16>" : "sslcr::conv3x3_h16s_kernel<float, 64, 2, false, 16>";
  }
  if (h16_mode(dtype, a) == 8) {
    if (!bf) return xf ? "sslcr::conv3x3_h16_kernel<float, 128, 2, true, false, false, 8>" : "sslcr::conv3x3_h16_kernel<float, 128, 2, false, false, false, 8>";
    if (h16_raw(a))
      return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, true, false, true, 8>" : "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, false, false, true, 8>";
    return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, true, false, false, 8>" : "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, false, false, false, 8>";
  }
  if (a.K % 128 == 0) {
    if (bf && h16_raw(a))
      return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, true, false, true, 16>" : "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, false, false, true, 16>";
    if (bf) return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, true, false, false, 16>" : "sslcr::conv3x3_h16_kernel<unsigned short, 128, 2, false, false, false, 16>";
    return xf ? "sslcr::conv3x3_h16_kernel<float, 128, 2, true, false, false, 16>" : "sslcr::conv3x3_h16_kernel<float, 128, 2, false, false, false, 16>";
  }
  if (bf && h16_resident(a) && conv_pp64_ok(DT_BF16, a)) return conv_pp64_name(a);
  if (bf && h16_resident(a))
    return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 64, 2, true, true, false, 16>" : "sslcr::conv3x3_h16_kernel<unsigned short, 64, 2, false, true, false, 16>";
  if (bf) return xf ? "sslcr::conv3x3_h16_kernel<unsigned short, 64, 2, true, false, false, 16>" : "sslcr::conv3x3_h16_kernel<unsigned short, 64, 2, false, false, false, 16>";
  return xf ? "sslcr::conv3x3_h16_kernel<float, 64, 2, true, false, false, 16>" : "sslcr::conv3x3_h16_kernel<float, 64, 2, false, false, false, 16>";
}

}  // namespace sslcr

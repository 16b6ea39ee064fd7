// BatchNorm statistics/finalize, fused BN(+residual)+ReLU, max/avg pooling and BN backward for NHWC tensors.
// All HBM-bound: 16-byte vector accesses, channel-contiguous, grid-stride.  Replaces nn.BatchNorm2d / ReLU /
// residual add / MaxPool2d / AdaptiveAvgPool2d of torchvision resnet18 and their autograd (K13).
#include <stdlib.h>

#include "kernels.hpp"

namespace sslcr {

// ------------------------------------------------------------------ statistics: partial rows -> sums -> scale/shift
// stage 1: grid (C/32, SPLITS, nseg); block 256 = 32 channels x 8 row lanes.  Segment z owns rows [z * rows, (z + 1) * rows) of
// `part` and the z-th [splits][2][C] slab of `out` (rows = rows per segment)
__global__ __launch_bounds__(256) void bn_reduce_rows_kernel(const float* __restrict__ part, int rows, int C, double* __restrict__ out, int splits) {
  __shared__ double sm[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  const int per = (rows + splits - 1) / splits;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  part += (size_t)blockIdx.z * rows * 2 * C;
  out += (size_t)blockIdx.z * splits * 2 * C;
  double s = 0.0, ss = 0.0;
  if (c < C) {
    int r = r0 + rl;
    for (; r + 8 < r1; r += 16) {               // two rows in flight per trip
      const float* p = part + (size_t)r * 2 * C;
      const float* q = p + (size_t)16 * C;
      const float a0 = p[c], a1 = p[C + c], b0 = q[c], b1 = q[C + c];
      s += (double)a0 + (double)b0;
      ss += (double)a1 + (double)b1;
    }
    for (; r < r1; r += 8) {
      const float* p = part + (size_t)r * 2 * C;
      s += (double)p[c];
      ss += (double)p[C + c];
    }
  }
  sm[0][rl][threadIdx.x & 31] = s;
  sm[1][rl][threadIdx.x & 31] = ss;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, cc = threadIdx.x & 31;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += sm[which][i][cc];
    const int ch = blockIdx.x * 32 + cc;
    if (ch < C) out[((size_t)blockIdx.y * 2 + which) * C + ch] = t;
  }
}

// stage 2 (+ finalize): one thread per channel; with segments (a.nseg > 1) the thread finalizes them one after the other, so the
// running statistics see the updates in segment order -- the order of the reference's successive forward calls.
// cb = channel block (32 channels), 256 threads = 32 channels x 8 lanes.  SC1: the stage rows were published by OTHER workgroups of
// this launch with write-through stores (bn_reduce_finalize_kernel): read them past this CU's L1
template <bool SC1>
__device__ __forceinline__ void bn_finalize_block(const double* __restrict__ stage, int splits, const BnFinalizeArgs& a, int cb) {
  // 8 lanes per channel share the split rows (a serial loop over 32 splits is 64 dependent loads = 10 us per BatchNorm)
  const int c = cb * 32 + (threadIdx.x >> 3);
  const int part = threadIdx.x & 7;
  const int cc = c < a.C ? c : a.C - 1;
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  float rm = 0.f, rv = 0.f;
  const bool lead = c < a.C && part == 0;
  if (lead && a.running_mean) { rm = a.running_mean[c]; rv = a.running_var[c]; }
  for (int z = 0; z < nseg; ++z) {
    double s = 0.0, ss = 0.0;
    if (a.sums_in) {
      s = a.sums_in[cc];
      ss = a.sums_in[a.C + cc];
    } else {
      const double* stz = stage + (size_t)z * splits * 2 * a.C;
      for (int i = part; i < splits; i += 8) {
        const double* p0 = stz + ((size_t)i * 2) * a.C + cc;
        if (SC1) {
          s += __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ss += __hip_atomic_load(p0 + a.C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          s += p0[0];
          ss += p0[a.C];
        }
      }
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) { s += __shfl_xor(s, m); ss += __shfl_xor(ss, m); }
    }
    if (!lead) continue;
    if (a.sums_out) {
      double* o = a.sums_out + (size_t)z * a.seg_stride;
      o[c] = s;
      o[a.C + c] = ss;
      continue;
    }
    const size_t so = (size_t)z * a.seg_stride;
    const double mean = s / a.count;
    double var = ss / a.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const double sc = (double)a.gamma[c] * invstd;
    a.scale[so + c] = (float)sc;
    a.shift[so + c] = (float)((double)a.beta[c] - mean * sc);
    if (a.mean) a.mean[so + c] = (float)mean;
    if (a.invstd) a.invstd[so + c] = (float)invstd;
    if (a.running_mean) {
      const double unb = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
      for (int i = 0; i < a.replay; ++i) {       // same arithmetic as `replay` sequential nn.BatchNorm2d updates
        rm = (1.f - a.momentum) * rm + a.momentum * (float)mean;
        rv = (1.f - a.momentum) * rv + a.momentum * (float)unb;
      }
    }
  }
  if (lead && a.running_mean && !a.sums_out) {
    a.running_mean[c] = rm;
    a.running_var[c] = rv;
    if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += (int64_t)a.replay * nseg;
  }
}
__global__ void bn_finalize_kernel(const double* __restrict__ stage, int splits, const BnFinalizeArgs a) {
  bn_finalize_block<false>(stage, splits, a, blockIdx.x);
}

// Both stages in ONE launch (a.tickets != nullptr; round 6): the grid and every thread's arithmetic are bn_reduce_rows_kernel's, and
// the workgroup that arrives LAST at its channel block's ticket (splits x nseg arrivals) runs bn_finalize_block for the block --
// the same lanes, loads and adds as the second launch: the same bits, one launch (and one dependent launch boundary) less per
// BatchNorm.  Cross-workgroup visibility without an L2 write-back (a release fence at agent scope is `buffer_wbl2`, and the conv
// that has just run left megabytes of dirty output in every XCD's L2 -- what made the round-2 ticket form slower than two launches):
// the 64 stage values of a workgroup are published with write-through (sc1) stores, the publishing wave waits vmcnt(0) before its
// lane 0 takes the ticket, and the last arriver reads the stage with sc1 loads (past its L1; no XCD's L2 can hold an older copy of
// these lines: kernel start invalidated them and nobody reads them before the last ticket).  The ticket word is left at 0.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ part, int rows, double* __restrict__ stage, int splits,
                                                                 const BnFinalizeArgs a) {
  __shared__ double sm[2][8][32];
  __shared__ int s_last;
  const int C = a.C;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  const int per = (rows + splits - 1) / splits;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  part += (size_t)blockIdx.z * rows * 2 * C;
  double* out = stage + (size_t)blockIdx.z * splits * 2 * C;
  double s = 0.0, ss = 0.0;
  if (c < C) {
    int r = r0 + rl;
    for (; r + 8 < r1; r += 16) {               // two rows in flight per trip
      const float* p = part + (size_t)r * 2 * C;
      const float* q = p + (size_t)16 * C;
      const float a0 = p[c], a1 = p[C + c], b0 = q[c], b1 = q[C + c];
      s += (double)a0 + (double)b0;
      ss += (double)a1 + (double)b1;
    }
    for (; r < r1; r += 8) {
      const float* p = part + (size_t)r * 2 * C;
      s += (double)p[c];
      ss += (double)p[C + c];
    }
  }
  sm[0][rl][threadIdx.x & 31] = s;
  sm[1][rl][threadIdx.x & 31] = ss;
  __syncthreads();
  if (threadIdx.x < 64) {                       // wave 0: publish, drain, ticket
    const int which = threadIdx.x >> 5, cc = threadIdx.x & 31;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += sm[which][i][cc];
    const int ch = blockIdx.x * 32 + cc;
    if (ch < C) __hip_atomic_store(&out[((size_t)blockIdx.y * 2 + which) * C + ch], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      const int total = (int)(gridDim.y * gridDim.z);
      const int old = __hip_atomic_fetch_add(a.tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == total - 1;
      if (old == total - 1) __hip_atomic_store(a.tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    }
  }
  __syncthreads();
  if (!s_last) return;
  bn_finalize_block<true>(stage, splits, a, blockIdx.x);
}

static hipError_t launch_bn_finalize_once(const BnFinalizeArgs& a, hipStream_t st);
hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t st) {
  // SSLCR_BN_REPEAT=n (measurement only: the running statistics take n updates): the launch(es) n times back to back -- the step
  // time difference / ((n - 1) x launches) is what one BatchNorm finalize costs IN the stream, without a profiler's per-kernel overhead
  static const int rep = [] { const char* e = getenv("SSLCR_BN_REPEAT"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
  hipError_t e = hipSuccess;
  for (int i = 0; i < rep && e == hipSuccess; ++i) e = launch_bn_finalize_once(a, st);
  return e;
}
static hipError_t launch_bn_finalize_once(const BnFinalizeArgs& a, hipStream_t st) {
  constexpr int SPLITS = 32;
  double* stage = a.stage;
  int splits = 0;
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  if (nseg > 1 && (a.sums_in || a.rows % nseg != 0)) return hipErrorInvalidValue;
  if (!a.sums_in) {
    const int rows = a.rows / nseg;             // per segment
    splits = rows / 32;                         // >= 4 dependent row loads per thread before it is worth another block row
    if (splits > SPLITS) splits = SPLITS;
    if (splits < 1) splits = 1;
    // OFF by default (round 6, profiles/r06_bn_one_launch_ab.txt): in the stream a finalize pair costs 4.35 us (two launches) and the
    // ticketed launch 5.9 us -- the write-through publish, the returned atomic and the sc1 reads are dearer than a launch boundary
    static const bool one = [] { const char* e = getenv("SSLCR_BN_ONE_LAUNCH"); return e && atoi(e) != 0; }();
    if (a.tickets && one) {
      hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(cdiv(a.C, 32), splits, nseg), dim3(256), 0, st, a.partials, rows, stage, splits, a);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_reduce_rows_kernel, dim3(cdiv(a.C, 32), splits, nseg), dim3(256), 0, st, a.partials, rows, a.C, stage, splits);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(a.C, 32)), dim3(256), 0, st, stage, splits, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ y = relu(bn(x) [+ bn(res) | + res])
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const BnActArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  // segments (sslcr_bn_act_desc.nseg): grid y = segment, an equal share of the pixels with its own constants
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  const size_t total = a.pixels / nseg * cols;
  const size_t seg_bytes = total * 16 * blockIdx.y, so = (size_t)blockIdx.y * a.seg_stride;
  const char* x = reinterpret_cast<const char*>(a.x) + seg_bytes;
  const char* r = a.res ? reinterpret_cast<const char*>(a.res) + seg_bytes : nullptr;
  char* y = reinterpret_cast<char*>(a.y) + seg_bytes;
  // grid stride is a multiple of cols: the thread's EPC channels (and their constants) are fixed for the whole loop
  const int cb = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % cols) * EPC;
  float sc[EPC], sh[EPC], rsc[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sc[e] = a.scale[so + cb + e]; sh[e] = a.shift[so + cb + e];
    rsc[e] = a.rscale ? a.rscale[so + cb + e] : 1.f; rsh[e] = a.rscale ? a.rshift[so + cb + e] : 0.f;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float f[EPC], g[EPC];
    Elem<T>::unpack(ld16_nt(x + i * 16), f);
    if (r) Elem<T>::unpack(ld16_nt(r + i * 16), g);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = fmaf(f[e], sc[e], sh[e]);
      if (r) v += fmaf(g[e], rsc[e], rsh[e]);
      f[e] = a.relu ? fmaxf(v, 0.f) : v;
    }
    const u32x4_t packed = Elem<T>::pack(f);
    st16_nt(y + i * 16, packed);
    if constexpr (EPC == 8) {
      // the sign mask of the STORED values (a positive float below half the smallest bf16 has become 0): what (y > 0) will read
      if (a.ybits) {
        unsigned b = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned lo = packed[d] & 0xffffu, hi = packed[d] >> 16;
          b |= ((lo - 1u) < 0x7fffu ? 1u : 0u) << (2 * d);          // 0x0001 .. 0x7fff: a positive bf16 (and a positive-signed NaN, the one
          b |= ((hi - 1u) < 0x7fffu ? 1u : 0u) << (2 * d + 1);      //   case where (y > 0) says otherwise)
        }
        a.ybits[(size_t)blockIdx.y * total + i] = (uint8_t)b;
      }
    }
  }
}

static inline int ew_grid(size_t work_items) {
  // THREE workgroups per CU (768 on the MI355X).  Same-box sweeps of the default step, r04 (gpurun_out/cap): 256 -> 17.0 ms,
  // 512 15.86, 640 15.82, 768 15.73-15.80, 896 15.87, 1024 +0.03 over 768, 1536 / 2048 (the cap of rounds 2-4) / 3072 15.86-15.87,
  // 40960 and up +2 ms (the per-workgroup constants).  tools/microbench/hbm_bench.hip agrees for a 2-read-1-write stream: 1024
  // workgroups 6.16 TB/s, 2048 5.8, 4096 5.5 -- and more only again with one vector per thread and no prologue at all
  // (81920 workgroups: 6.6 TB/s).  Initialised once, thread-safely (virtual ranks launch from several host threads);
  // SSLCR_EW_CAP is read as a signed value and clamped to [256, 65535]
  static const size_t cap = [] {
    const char* e = getenv("SSLCR_EW_CAP");
    const long dflt = 3L * device_cus();
    long v = e ? strtol(e, nullptr, 10) : dflt;
    if (v < 256) v = e ? 256 : dflt;
    if (v > 65535) v = 65535;
    return (size_t)v;
  }();
  size_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t st) {
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  if (a.pixels % nseg != 0) return hipErrorInvalidValue;
  const size_t per = a.pixels / nseg;
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(ew_grid(per * (a.C / 8)), nseg), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(bn_act_kernel<float>, dim3(ew_grid(per * (a.C / 4)), nseg), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ stem: maxpool3x3/2 p1 over relu(bn(x)), with argmax
// argmax code = window position r * 3 + s of the first maximum, or 9 where the maximum is not positive: the ReLU in front of
// the pool passes no gradient into such a window (torch: relu'(x <= 0) = 0), so the backward kernels need no ReLU test.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const PoolFwdArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const size_t total = (size_t)a.N * a.OH * a.OW * cols;
  const char* x = reinterpret_cast<const char*>(a.x);
  char* y = reinterpret_cast<char*>(a.y);
  float psc[EPC], psh[EPC];                  // fixed channels per thread (grid stride is a multiple of cols)
  {
    const int cb0 = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % cols) * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { psc[e] = a.scale[cb0 + e]; psh[e] = a.shift[cb0 + e]; }
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    size_t pix = i / cols;
    const int ow = (int)(pix % a.OW);
    pix /= a.OW;
    const int oh = (int)(pix % a.OH);
    const int n = (int)(pix / a.OH);
    float best[EPC];
    int arg[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
    for (int wi = 0; wi < 9; ++wi) {
      const int h = 2 * oh - 1 + wi / 3, w = 2 * ow - 1 + wi % 3;
      if (h < 0 || w < 0 || h >= a.H || w >= a.W) continue;
      float f[EPC];
      Elem<T>::unpack(ld16(x + (((size_t)n * a.H + h) * a.W + w) * a.C * sizeof(T) + (size_t)col * 16), f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float v = fmaxf(fmaf(f[e], psc[e], psh[e]), 0.f);
        if (v > best[e]) { best[e] = v; arg[e] = wi; }
      }
    }
    st16_nt(y + i * 16, Elem<T>::pack(best));
#pragma unroll
    for (int e = 0; e < EPC; ++e) arg[e] = best[e] > 0.f ? arg[e] : 9;      // code 9: no pixel of this window receives gradient
    if (a.argmax) {
      uint8_t* ap = a.argmax + i * EPC;
      if (EPC == 8) {
        uint32_t lo = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
        uint32_t hi = arg[4 % EPC] | (arg[5 % EPC] << 8) | (arg[6 % EPC] << 16) | (arg[7 % EPC] << 24);
        *reinterpret_cast<u32x2_t*>(ap) = u32x2_t{lo, hi};
      } else {
        *reinterpret_cast<uint32_t*>(ap) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
      }
    }
  }
}

// Row form (256 % (C / EPC) == 0, every ResNet width): a workgroup walks output rows (n, oh); a thread keeps its channel
// chunk, so the index arithmetic is one uniform division per ROW instead of six 64-bit divisions per 16-byte chunk, and the
// nine window loads are unconditional (clamped address, validity by select) so they are all in flight together.
// PLAIN (scale == nullptr): x is pooled as it is -- the eval path, where conv1's epilogue already applied the folded BatchNorm and
// the ReLU.  No affine, no ReLU, no argmax, and no validity selects either: a clamped address re-reads a pixel of the same window,
// and a duplicate does not change a maximum.  9 x (unpack + max) instead of 9 x (unpack + fma + max + compare + 2 selects).
// Which output rows a workgroup of the row-form pooling kernels walks.  bands (round 6, SSLCR_POOL_BANDS=1; measured slower, off): workgroup b runs on XCD b % 8; XCD x owns the
// x-th eighth of the rows and each of its workgroups a contiguous run of that eighth, so the input row that two consecutive output rows
// share (2 oh + 1 = 2 (oh + 1) - 1) is read once -- by the same workgroup -- instead of by two workgroups on two XCDs (rows b, b + G, ...:
// the input crossed the fabric 1.5 x).  Same arithmetic per row, same bits.
__device__ __forceinline__ void pool_row_range(int rows, int bands, int& r_begin, int& r_end, int& r_step) {
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  if (!bands || (G & 7)) { r_begin = b; r_end = rows; r_step = G; return; }
  const int x = b & 7, j = b >> 3, gpx = G >> 3;
  const int x0 = (int)((long)rows * x / 8), x1 = (int)((long)rows * (x + 1) / 8);
  r_begin = x0 + (int)((long)(x1 - x0) * j / gpx);
  r_end = x0 + (int)((long)(x1 - x0) * (j + 1) / gpx);
  r_step = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_rows_kernel(const PoolFwdArgs a, const int bands) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const int ppp = 256 / cols;
  const int col = threadIdx.x % cols, pw0 = threadIdx.x / cols;
  const char* x = reinterpret_cast<const char*>(a.x) + (size_t)col * 16;
  const int rows = a.N * a.OH;
  const size_t rowb = (size_t)a.W * a.C * sizeof(T);
  int r_begin, r_end, r_step;
  pool_row_range(rows, bands, r_begin, r_end, r_step);
  for (int r = r_begin; r < r_end; r += r_step) {
    const int n = r / a.OH, oh = r - n * a.OH;
    const int h0 = 2 * oh - 1;
    const char* xn = x + (size_t)n * a.H * rowb;
    for (int ow = pw0; ow < a.OW; ow += ppp) {
      const int w0 = 2 * ow - 1;
      u32x4_t v[9];
#pragma unroll
      for (int wi = 0; wi < 9; ++wi) {
        int h = h0 + wi / 3, w = w0 + wi % 3;
        h = h < 0 ? 0 : (h >= a.H ? a.H - 1 : h);
        w = w < 0 ? 0 : (w >= a.W ? a.W - 1 : w);
        v[wi] = ld16(xn + (size_t)h * rowb + (size_t)w * a.C * sizeof(T));
      }
      float best[EPC];
      Elem<T>::unpack(v[0], best);
#pragma unroll
      for (int wi = 1; wi < 9; ++wi) {
        float f[EPC];
        Elem<T>::unpack(v[wi], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) best[e] = fmaxf(best[e], f[e]);
      }
      const size_t i = ((size_t)r * a.OW + ow) * cols + col;
      st16_nt(reinterpret_cast<char*>(a.y) + i * 16, PackH<T>::run(best));     // (exact: the maximum is one of the inputs)
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_rows_kernel(const PoolFwdArgs a, const int bands) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const int ppp = 256 / cols;                 // output pixels per pass
  const int col = threadIdx.x % cols, pw0 = threadIdx.x / cols;
  const char* x = reinterpret_cast<const char*>(a.x) + (size_t)col * 16;
  float psc[EPC], psh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { psc[e] = a.scale[col * EPC + e]; psh[e] = a.shift[col * EPC + e]; }
  const int rows = a.N * a.OH;
  const size_t rowb = (size_t)a.W * a.C * sizeof(T);
  int r_begin, r_end, r_step;
  pool_row_range(rows, bands, r_begin, r_end, r_step);
  for (int r = r_begin; r < r_end; r += r_step) {
    const int n = r / a.OH, oh = r - n * a.OH;
    const int h0 = 2 * oh - 1;
    const char* xn = x + (size_t)n * a.H * rowb;
    for (int ow = pw0; ow < a.OW; ow += ppp) {
      const int w0 = 2 * ow - 1;
      u32x4_t v[9];
#pragma unroll
      for (int wi = 0; wi < 9; ++wi) {
        int h = h0 + wi / 3, w = w0 + wi % 3;
        h = h < 0 ? 0 : (h >= a.H ? a.H - 1 : h);
        w = w < 0 ? 0 : (w >= a.W ? a.W - 1 : w);
        v[wi] = ld16(xn + (size_t)h * rowb + (size_t)w * a.C * sizeof(T));
      }
      float best[EPC];
      int arg[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
      for (int wi = 0; wi < 9; ++wi) {
        const int h = h0 + wi / 3, w = w0 + wi % 3;
        const bool ok = h >= 0 && w >= 0 && h < a.H && w < a.W;
        float f[EPC];
        Elem<T>::unpack(v[wi], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float q = fmaxf(fmaf(f[e], psc[e], psh[e]), 0.f);
          if (ok && q > best[e]) { best[e] = q; arg[e] = wi; }
        }
      }
      const size_t i = ((size_t)r * a.OW + ow) * cols + col;
      st16_nt(reinterpret_cast<char*>(a.y) + i * 16, Elem<T>::pack(best));
#pragma unroll
      for (int e = 0; e < EPC; ++e) arg[e] = best[e] > 0.f ? arg[e] : 9;    // code 9: no pixel of this window receives gradient
      if (a.argmax) {
        uint8_t* ap = a.argmax + i * EPC;
        if (EPC == 8) {
          uint32_t lo = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
          uint32_t hi = arg[4 % EPC] | (arg[5 % EPC] << 8) | (arg[6 % EPC] << 16) | (arg[7 % EPC] << 24);
          *reinterpret_cast<u32x2_t*>(ap) = u32x2_t{lo, hi};
        } else {
          *reinterpret_cast<uint32_t*>(ap) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
        }
      }
    }
  }
}

hipError_t launch_bn_relu_maxpool(int dtype, const PoolFwdArgs& a, hipStream_t st) {
  // OFF: measured +0.07 ms per step on one box (15.39 -> 15.46 ms, three alternations) -- the input row two output rows share comes from
  // L2 / the Infinity Cache either way, and interleaved rows keep more channels busy
  static const int bands = [] { const char* e = getenv("SSLCR_POOL_BANDS"); return (e && atoi(e) != 0) ? 1 : 0; }();
  size_t px = (size_t)a.N * a.OH * a.OW;
  const int cols = a.C / (dtype == DT_BF16 ? 8 : 4);
  if (!a.scale || !a.shift) {          // plain max-pool (capi.cpp admits it only without argmax and for the row form's widths)
    const int rows = a.N * a.OH;
    const int grid = rows < 256 * 16 ? rows : 256 * 16;
    if (dtype == DT_BF16) hipLaunchKernelGGL(maxpool_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a, bands);
    else hipLaunchKernelGGL(maxpool_rows_kernel<float>, dim3(grid), dim3(256), 0, st, a, bands);
    return hipGetLastError();
  }
  if (cols >= 1 && cols <= 256 && 256 % cols == 0) {
    const int rows = a.N * a.OH;
    const int grid = rows < 256 * 16 ? rows : 256 * 16;
    if (dtype == DT_BF16) hipLaunchKernelGGL(bn_relu_maxpool_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a, bands);
    else hipLaunchKernelGGL(bn_relu_maxpool_rows_kernel<float>, dim3(grid), dim3(256), 0, st, a, bands);
    return hipGetLastError();
  }
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(ew_grid(px * (a.C / 8))), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_grid(px * (a.C / 4))), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// dx[n,h,w,c] = (bn(x)>0) * sum_{windows containing (h,w) whose argmax is (h,w)} dy[window]
template <typename T>
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const PoolBwdArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const size_t total = (size_t)a.N * a.H * a.W * cols;
  const char* x = reinterpret_cast<const char*>(a.x);
  const char* dy = reinterpret_cast<const char*>(a.dy);
  char* dx = reinterpret_cast<char*>(a.dx);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    size_t pix = i / cols;
    const int w = (int)(pix % a.W);
    pix /= a.W;
    const int h = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int cb = col * EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    const int oh0 = h >> 1, ow0 = w >> 1;          // candidates: oh in {h/2, (h+1)/2}
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const int oh = oh0 + dh;
      const int wi_h = h - (2 * oh - 1);
      if (oh >= a.OH || wi_h < 0 || wi_h > 2 || (dh == 1 && ((h & 1) == 0))) continue;
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int ow = ow0 + dw;
        const int wi_w = w - (2 * ow - 1);
        if (ow >= a.OW || wi_w < 0 || wi_w > 2 || (dw == 1 && ((w & 1) == 0))) continue;
        const int code = wi_h * 3 + wi_w;
        const size_t o = (((size_t)n * a.OH + oh) * a.OW + ow) * cols + col;
        float g[EPC];
        Elem<T>::unpack(ld16(dy + o * 16), g);
        uint32_t am[2];
        if (EPC == 8) { const u32x2_t v = *reinterpret_cast<const u32x2_t*>(a.argmax + o * EPC); am[0] = v[0]; am[1] = v[1]; }
        else { am[0] = *reinterpret_cast<const uint32_t*>(a.argmax + o * EPC); am[1] = 0; }
#pragma unroll
        for (int e = 0; e < EPC; ++e)
          if (((am[e >> 2] >> (8 * (e & 3))) & 0xffu) == (uint32_t)code) acc[e] += g[e];
      }
    }
    float f[EPC];
    Elem<T>::unpack(ld16_nt(x + i * 16), f);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(fmaf(f[e], a.scale[cb + e], a.shift[cb + e]) > 0.f)) acc[e] = 0.f;
    st16_nt(dx + i * 16, Elem<T>::pack(acc));
  }
}

hipError_t launch_maxpool_relu_bwd(int dtype, const PoolBwdArgs& a, hipStream_t st) {
  size_t px = (size_t)a.N * a.H * a.W;
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel<bf16_t>, dim3(ew_grid(px * (a.C / 8))), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel<float>, dim3(ew_grid(px * (a.C / 4))), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ global average pool
// block = 64 channel chunks x 4 pixel lanes: lane q sums pixels q, q + 4, ... of its (image, chunk) -- four times the loads in
// flight of the one-thread-per-chunk form (33 us for 640 x 64 x 512 bf16 = 42 MB on 160 workgroups; round 6) -- and the four lane
// sums are added in lane order by lane 0 (a fixed order: the same bits run after run)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const void* xv, float* y, int N, int HW, int C) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float sm[3][64][EPC + 1];
  const int cols = C / EPC;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  const bool ok = i < N * cols;
  const int n = ok ? i / cols : 0, col = ok ? i - n * cols : 0;
  const char* x = reinterpret_cast<const char*>(xv) + ((size_t)n * HW * C) * sizeof(T) + (size_t)col * 16;
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  if (ok) {
#pragma unroll 4
    for (int p = q; p < HW; p += 4) {
      float f[EPC];
      Elem<T>::unpack(ld16(x + (size_t)p * C * sizeof(T)), f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += f[e];
    }
  }
  if (q > 0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) sm[q - 1][cl][e] = acc[e];
  }
  __syncthreads();
  if (q > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] += sm[w][cl][e];
  const float inv = 1.f / (float)HW;
#pragma unroll
  for (int e = 0; e < EPC; ++e) y[(size_t)n * C + col * EPC + e] = acc[e] * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* dy, void* dxv, int N, int HW, int C) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = C / EPC;
  const size_t total = (size_t)N * HW * cols;
  char* dx = reinterpret_cast<char*>(dxv);
  const float inv = 1.f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    const int n = (int)(i / ((size_t)HW * cols));
    float f[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] = dy[(size_t)n * C + col * EPC + e] * inv;
    st16_nt(dx + i * 16, Elem<T>::pack(f));
  }
}

hipError_t launch_avgpool_fwd(int dtype, const void* x, float* y, int N, int HW, int C, hipStream_t st) {
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(cdiv(N * (C / 8), 64)), dim3(256), 0, st, x, y, N, HW, C);
  } else {
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(cdiv(N * (C / 4), 64)), dim3(256), 0, st, x, y, N, HW, C);
  }
  return hipGetLastError();
}
hipError_t launch_avgpool_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, hipStream_t st) {
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(ew_grid((size_t)N * HW * (C / 8))), dim3(256), 0, st, dy, dx, N, HW, C);
  } else {
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(ew_grid((size_t)N * HW * (C / 4))), dim3(256), 0, st, dy, dx, N, HW, C);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ BatchNorm backward
template <typename T>
__device__ __forceinline__ void bn_bwd_g(const BnBwdArgs& a, size_t i, const float* rsc, const float* rsh, float* g, float* xf) {
  constexpr int EPC = Elem<T>::EPC;
  Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.x) + i * 16), xf);
  if (a.pool_dy) {
    // the gradient arrives through a 3x3/2 pad-1 max-pool (the stem): gather it from the <= 4 pooled windows that contain
    // this pixel and whose recorded argmax is this pixel -- the un-pooled gradient tensor is never materialised
    const int cols = a.C / EPC, col = (int)(i % cols);
    size_t pix = i / cols;
    const int w = (int)(pix % a.pW);
    pix /= a.pW;
    const int h = (int)(pix % a.pH), n = (int)(pix / a.pH);
#pragma unroll
    for (int e = 0; e < EPC; ++e) g[e] = 0.f;
    // all four candidate windows are loaded UNCONDITIONALLY (clamped addresses) so the loads go out back to back; the
    // window is then kept or dropped by a select -- no divergent branches around memory operations
    u32x4_t dv[4];
    uint32_t am[4][2];
    bool ok[4];
    int code[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int dh = k >> 1, dw = k & 1;
      const int oh = (h >> 1) + dh, ow = (w >> 1) + dw;
      ok[k] = (dh == 0 || (h & 1)) && (dw == 0 || (w & 1)) && oh < a.pOH && ow < a.pOW;
      code[k] = (h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1));
      const int ohc = oh < a.pOH ? oh : a.pOH - 1, owc = ow < a.pOW ? ow : a.pOW - 1;
      const size_t o = (((size_t)n * a.pOH + ohc) * a.pOW + owc) * cols + col;
      dv[k] = ld16(reinterpret_cast<const char*>(a.pool_dy) + o * 16);
      if (EPC == 8) { const u32x2_t v = *reinterpret_cast<const u32x2_t*>(a.pool_argmax + o * EPC); am[k][0] = v[0]; am[k][1] = v[1]; }
      else { am[k][0] = *reinterpret_cast<const uint32_t*>(a.pool_argmax + o * EPC); am[k][1] = 0; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d[EPC];
      Elem<T>::unpack(dv[k], d);
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        if (ok[k] && ((am[k][e >> 2] >> (8 * (e & 3))) & 0xffu) == (uint32_t)code[k]) g[e] += d[e];
    }
  } else {
    Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.dy) + i * 16), g);
  }
  if (EPC == 8 && a.yact_bits) {
    const unsigned b = a.yact_bits[i];
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!((b >> e) & 1u)) g[e] = 0.f;
  } else if (a.yact) {
    float ya[EPC];
    Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.yact) + i * 16), ya);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(ya[e] > 0.f)) g[e] = 0.f;
  } else if (a.relu_from_x) {
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(fmaf(xf[e], rsc[e], rsh[e]) > 0.f)) g[e] = 0.f;
  }
}

// affine gradients from the finished sums, by workgroup 0 of the apply pass (same arithmetic as bn_param_grads_kernel); with
// segments the contributions are added one after the other, like the per-segment launches would
__device__ __forceinline__ void bn_bwd_param_grads(const BnBwdArgs& a) {
  if (blockIdx.x != 0 || blockIdx.y != 0 || !a.dgamma || !a.dbeta) return;
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float dg = a.dgamma[c], db = a.dbeta[c];
    for (int z = 0; z < nseg; ++z) {
      const double* sums = a.sums + (size_t)z * a.sums_stride;
      dg += (float)(sums[a.C + c] * (double)a.invstd[(size_t)z * a.seg_stride + c] * (double)a.pg_scale);
      db += (float)(sums[c] * (double)a.pg_scale);
    }
    a.dgamma[c] = dg;
    a.dbeta[c] = db;
  }
}
// the descriptor of segment blockIdx.y (sslcr_bn_bwd_desc.nseg): its share of the tensors, its constants, its sums
template <typename T>
__device__ __forceinline__ BnBwdArgs bn_bwd_segment(const BnBwdArgs& a) {
  BnBwdArgs b = a;
  if (a.nseg > 1) {
    const int z = blockIdx.y;
    b.pixels = a.pixels / a.nseg;
    const size_t off = (size_t)z * b.pixels * a.C * sizeof(T);
    if (a.dy) b.dy = reinterpret_cast<const char*>(a.dy) + off;
    if (a.x) b.x = reinterpret_cast<const char*>(a.x) + off;
    if (a.yact) b.yact = reinterpret_cast<const char*>(a.yact) + off;
    if (a.yact_bits) b.yact_bits = a.yact_bits + (size_t)z * b.pixels * a.C / 8;
    if (a.dx) b.dx = reinterpret_cast<char*>(a.dx) + off;
    if (a.gout) b.gout = reinterpret_cast<char*>(a.gout) + off;
    const size_t so = (size_t)z * a.seg_stride;
    b.scale = a.scale + so; b.shift = a.shift + so; b.mean = a.mean + so; b.invstd = a.invstd + so;
    b.sums = a.sums + (size_t)z * a.sums_stride;
  }
  return b;
}

// NB threads per workgroup.  A workgroup leaves its 2C per-channel sums as one fp64 row of `rows` ([segment][workgroup][2][C]);
// bn_bwd_sums_kernel adds the rows in a fixed order.  (Until round 4 the sums ended in fp64 atomics on 2C addresses: an order that
// changes run to run, and atomics on one address serialise -- ~18 us at the end of a 1024-workgroup launch.)
template <typename T, int NB>
__global__ __launch_bounds__(NB) void bn_bwd_reduce_kernel(const BnBwdArgs a0, double* __restrict__ rows) {
  const BnBwdArgs a = bn_bwd_segment<T>(a0);
  constexpr int EPC = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) char bn_red_smem[];
  float (*sm)[2 * EPC + 1] = reinterpret_cast<float (*)[2 * EPC + 1]>(bn_red_smem);
  const int cols = a.C / EPC;                 // <= 256 and a power of two for resnet18
  const int rpp = NB / cols;                  // pixel rows per pass
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int cb = col * EPC;
  // per-channel constants of this thread's EPC channels live in registers for the whole grid-stride loop (re-reading them per
  // 16-byte chunk costs more L1 traffic than the tensor data itself)
  float s0[EPC], s1[EPC], mean[EPC], rsc[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { s0[e] = 0.f; s1[e] = 0.f; mean[e] = a.mean[cb + e]; rsc[e] = a.scale[cb + e]; rsh[e] = a.shift[cb + e]; }
  if (rl < rpp) {
    for (size_t p = (size_t)blockIdx.x * rpp + rl; p < a.pixels; p += (size_t)gridDim.x * rpp) {
      float g[EPC], xf[EPC];
      bn_bwd_g<T>(a, p * cols + col, rsc, rsh, g, xf);
      if (a.g_in_reduce) st16(reinterpret_cast<char*>(a.gout) + (p * cols + col) * 16, Elem<T>::pack(g));
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s0[e] += g[e]; s1[e] = fmaf(g[e], xf[e] - mean[e], s1[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sm[threadIdx.x][e] = s0[e]; sm[threadIdx.x][EPC + e] = s1[e]; }
  __syncthreads();
  // thread t < cols*2*EPC: (col, which/e) sums over rl
  for (int t = threadIdx.x; t < cols * 2 * EPC; t += NB) {
    const int cc = t / (2 * EPC), q = t - cc * 2 * EPC;
    double acc = 0.0;
    for (int r = 0; r < rpp; ++r) acc += (double)sm[r * cols + cc][q];
    const int which = q / EPC, e = q - which * EPC;
    rows[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + which) * a.C + cc * EPC + e] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a0) {
  const BnBwdArgs a = bn_bwd_segment<T>(a0);
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const size_t total = a.pixels * cols;
  const float invM = (float)(1.0 / a.count);
  // the grid stride (gridDim.x*256) is a multiple of cols, so a thread always works on the same EPC channels:
  // dx = cA*g + cB*x + cC with cA = scale, cB = -scale*invstd^2*mean(g*(x-mu)), cC = -scale*mean(g) - cB*mu
  const int cb = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % cols) * EPC;
  float cA[EPC], cB[EPC], cC[EPC], rsc[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = cb + e;
    const float is = a.invstd[c], sc = a.scale[c];
    const float m0 = (float)a.sums[c] * invM, m1 = (float)a.sums[a.C + c] * invM;
    cA[e] = sc;
    cB[e] = -sc * is * is * m1;
    cC[e] = -sc * m0 - cB[e] * a.mean[c];
    rsc[e] = sc; rsh[e] = a.shift[c];
  }
  bn_bwd_param_grads(a0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float g[EPC], xf[EPC], d[EPC];
    bn_bwd_g<T>(a, i, rsc, rsh, g, xf);
#pragma unroll
    for (int e = 0; e < EPC; ++e) d[e] = fmaf(cA[e], g[e], fmaf(cB[e], xf[e], cC[e]));
    st16_nt(reinterpret_cast<char*>(a.dx) + i * 16, Elem<T>::pack(d));
    if (a.gout) st16_nt(reinterpret_cast<char*>(a.gout) + i * 16, Elem<T>::pack(g));
  }
}

// ---- a downsampling block's TWO BatchNorms that receive the same gradient (bn2 of the residual branch and the projection
// shortcut's BatchNorm, torchvision BasicBlock: out = bn2(conv2(..)) + downsample(x) -> both get g = dOut * (y > 0)) in one reduce and
// one apply pass: g is formed once (a's dy / mask), written once (a.gout) and read once by the apply pass, where the two separate
// BatchNorm-backward passes read it twice each.  b is the second BatchNorm's descriptor (x, mean, scale, invstd, sums, dx, dgamma ...;
// its dy is a.gout by construction and is not read).  Same grid, same per-thread element order as the single kernels: both
// BatchNorms' sums come out with the bits of the separate launches.
// keep_g = 0: g is not written at all -- the apply pass forms it again from (dy, mask bits): one tensor write and one read less for
// 1/16 of a read (the engine's choice where nothing else reads g: a downsampling block has no identity path)
template <typename T, int NB>
__global__ __launch_bounds__(NB) void bn_bwd_reduce_pair_kernel(const BnBwdArgs a, const BnBwdArgs b, double* __restrict__ rows, const int keep_g) {
  constexpr int EPC = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) char bn_red_smem[];
  float (*sm)[3 * EPC + 1] = reinterpret_cast<float (*)[3 * EPC + 1]>(bn_red_smem);
  const int cols = a.C / EPC;
  const int rpp = NB / cols;
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int cb = col * EPC;
  float s0[EPC], s1[EPC], s1b[EPC], mean[EPC], meanb[EPC], rsc[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    s0[e] = 0.f; s1[e] = 0.f; s1b[e] = 0.f;
    mean[e] = a.mean[cb + e]; meanb[e] = b.mean[cb + e]; rsc[e] = a.scale[cb + e]; rsh[e] = a.shift[cb + e];
  }
  if (rl < rpp) {
    for (size_t p = (size_t)blockIdx.x * rpp + rl; p < a.pixels; p += (size_t)gridDim.x * rpp) {
      float g[EPC], xf[EPC], xb[EPC];
      const size_t i = p * cols + col;
      bn_bwd_g<T>(a, i, rsc, rsh, g, xf);
      Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(b.x) + i * 16), xb);
      if (keep_g) st16(reinterpret_cast<char*>(a.gout) + i * 16, Elem<T>::pack(g));
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        s0[e] += g[e];
        s1[e] = fmaf(g[e], xf[e] - mean[e], s1[e]);
        s1b[e] = fmaf(g[e], xb[e] - meanb[e], s1b[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sm[threadIdx.x][e] = s0[e]; sm[threadIdx.x][EPC + e] = s1[e]; sm[threadIdx.x][2 * EPC + e] = s1b[e]; }
  __syncthreads();
  // rows: [BatchNorm (a, b)][workgroup][2][C]; the second BatchNorm's sum of g is the first one's
  const size_t bstride = (size_t)gridDim.x * 2 * a.C;
  for (int t = threadIdx.x; t < cols * 3 * EPC; t += NB) {
    const int cc = t / (3 * EPC), q = t - cc * 3 * EPC;
    double acc = 0.0;
    for (int r = 0; r < rpp; ++r) acc += (double)sm[r * cols + cc][q];
    const int which = q / EPC, e = q - which * EPC;
    double* ra = rows + ((size_t)blockIdx.x * 2) * a.C + cc * EPC + e;
    if (which == 0) { ra[0] = acc; ra[bstride] = acc; }
    else if (which == 1) ra[a.C] = acc;
    else ra[bstride + a.C] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pair_kernel(const BnBwdArgs a, const BnBwdArgs b, const int from_g) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const size_t total = a.pixels * cols;
  const int cb = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % cols) * EPC;
  float cA[EPC], cB[EPC], cC[EPC], dA[EPC], dB[EPC], dC[EPC], rsc[EPC], rsh[EPC];
  {
    const float invM = (float)(1.0 / a.count), invMb = (float)(1.0 / b.count);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = cb + e;
      float is = a.invstd[c], sc = a.scale[c];
      float m0 = (float)a.sums[c] * invM, m1 = (float)a.sums[a.C + c] * invM;
      cA[e] = sc; cB[e] = -sc * is * is * m1; cC[e] = -sc * m0 - cB[e] * a.mean[c];
      rsc[e] = sc; rsh[e] = a.shift[c];
      is = b.invstd[c]; sc = b.scale[c];
      m0 = (float)b.sums[c] * invMb; m1 = (float)b.sums[b.C + c] * invMb;
      dA[e] = sc; dB[e] = -sc * is * is * m1; dC[e] = -sc * m0 - dB[e] * b.mean[c];
    }
  }
  bn_bwd_param_grads(a);
  bn_bwd_param_grads(b);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float g[EPC], xa[EPC], xb[EPC], d[EPC];
    if (from_g) {
      Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.gout) + i * 16), g);
      Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.x) + i * 16), xa);
    } else {
      bn_bwd_g<T>(a, i, rsc, rsh, g, xa);          // (dy, mask) again: the reduce pass kept no g
    }
    Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(b.x) + i * 16), xb);
#pragma unroll
    for (int e = 0; e < EPC; ++e) d[e] = fmaf(cA[e], g[e], fmaf(cB[e], xa[e], cC[e]));
    st16_nt(reinterpret_cast<char*>(a.dx) + i * 16, Elem<T>::pack(d));
#pragma unroll
    for (int e = 0; e < EPC; ++e) d[e] = fmaf(dA[e], g[e], fmaf(dB[e], xb[e], dC[e]));
    st16_nt(reinterpret_cast<char*>(b.dx) + i * 16, Elem<T>::pack(d));
  }
}

// ---- stem form: the gradient arrives through maxpool3x3/2 (pool_dy + argmax codes), see sslcr_bn_bwd_desc.
// Reduce pass on the POOLED tensors only (4x fewer elements, no 1.3 GB read of x): every pooled gradient lands on exactly one
// input pixel -- the argmax -- and relu(bn(x)) of that pixel IS the pooled output y, so for y > 0
//     g = dy_pool,   (x - mean) = (y - shift) / scale - mean
// and for y == 0 the ReLU mask kills the gradient.  A channel with scale == 0 (gamma exactly 0) cannot be inverted: those
// lanes fetch x at the argmax position instead.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const BnBwdArgs a, double* __restrict__ rows) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float sm[256][2 * EPC + 1];
  const int cols = a.C / EPC;
  const int rpp = 256 / cols;
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int cb = col * EPC;
  float s0[EPC], s1[EPC], mean[EPC], rinv[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    s0[e] = 0.f; s1[e] = 0.f; mean[e] = a.mean[cb + e]; rsh[e] = a.shift[cb + e];
    const float sc = a.scale[cb + e];
    rinv[e] = sc != 0.f ? 1.f / sc : 0.f;
  }
  const size_t ppix = (a.pixels / ((size_t)a.pH * a.pW)) * a.pOH * a.pOW;      // pooled pixels
  if (rl < rpp) {
    for (size_t p = (size_t)blockIdx.x * rpp + rl; p < ppix; p += (size_t)gridDim.x * rpp) {
      float d[EPC], y[EPC];
      Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.pool_dy) + (p * cols + col) * 16), d);
      Elem<T>::unpack(ld16_nt(reinterpret_cast<const char*>(a.pool_y) + (p * cols + col) * 16), y);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        if (a.relu_from_x && !(y[e] > 0.f)) continue;
        float xm;
        if (rinv[e] != 0.f) {
          xm = (y[e] - rsh[e]) * rinv[e] - mean[e];
        } else {
          const int code = a.pool_argmax[p * a.C + cb + e];
          size_t q = p;
          const int ow = (int)(q % a.pOW); q /= a.pOW;
          const int oh = (int)(q % a.pOH); const size_t n = q / a.pOH;
          const int h = 2 * oh - 1 + code / 3, w = 2 * ow - 1 + code % 3;
          xm = Elem<T>::ld(reinterpret_cast<const T*>(a.x) + ((n * a.pH + h) * a.pW + w) * a.C + cb + e) - mean[e];
        }
        s0[e] += d[e];
        s1[e] = fmaf(d[e], xm, s1[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sm[threadIdx.x][e] = s0[e]; sm[threadIdx.x][EPC + e] = s1[e]; }
  __syncthreads();
  for (int t = threadIdx.x; t < cols * 2 * EPC; t += 256) {
    const int cc = t / (2 * EPC), q = t - cc * 2 * EPC;
    double acc = 0.0;
    for (int r = 0; r < rpp; ++r) acc += (double)sm[r * cols + cc][q];
    const int which = q / EPC, e = q - which * EPC;
    rows[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + which) * a.C + cc * EPC + e] = acc;
  }
}

// Apply pass of the stem form on 2x2 input-pixel blocks: the four pixels (2a+i, 2b+j) share the four pooling windows
// (a+di, b+dj), so a thread loads 4 windows (gradient chunk + argmax codes) for 4 pixels instead of 4 per pixel, and the
// argmax code a window must carry to select pixel (i,j) is a compile-time constant (i - 2di + 1) * 3 + (j - 2dj + 1).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const BnBwdArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cols = a.C / EPC;
  const int BH = (a.pH + 1) / 2, BW = (a.pW + 1) / 2;
  const size_t N = a.pixels / ((size_t)a.pH * a.pW);
  const size_t total = N * BH * BW * cols;
  const float invM = (float)(1.0 / a.count);
  const int cb = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % cols) * EPC;      // grid stride is a multiple of cols
  float cA[EPC], cB[EPC], cC[EPC], rsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = cb + e;
    const float is = a.invstd[c], sc = a.scale[c];
    const float m0 = (float)a.sums[c] * invM, m1 = (float)a.sums[a.C + c] * invM;
    cA[e] = sc;
    cB[e] = -sc * is * is * m1;
    cC[e] = -sc * m0 - cB[e] * a.mean[c];
    rsh[e] = a.shift[c];
  }
  bn_bwd_param_grads(a);
  const char* xg = reinterpret_cast<const char*>(a.x);
  const char* dyg = reinterpret_cast<const char*>(a.pool_dy);
  char* dxg = reinterpret_cast<char*>(a.dx);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    size_t q = i / cols;
    const int bw = (int)(q % BW); q /= BW;
    const int bh = (int)(q % BH); const size_t n = q / BH;
    // the four windows: unconditional loads from clamped addresses, validity applied by select
    u32x4_t dv[2][2];
    uint32_t am[2][2][2];
    bool wok[2][2];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const int oh = bh + di, ow = bw + dj;
        wok[di][dj] = oh < a.pOH && ow < a.pOW;
        const int ohc = oh < a.pOH ? oh : a.pOH - 1, owc = ow < a.pOW ? ow : a.pOW - 1;
        const size_t o = ((n * a.pOH + ohc) * a.pOW + owc) * cols + col;
        dv[di][dj] = ld16_nt(dyg + o * 16);
        if (EPC == 8) { const u32x2_t v = *reinterpret_cast<const u32x2_t*>(a.pool_argmax + o * EPC); am[di][dj][0] = v[0]; am[di][dj][1] = v[1]; }
        else { am[di][dj][0] = *reinterpret_cast<const uint32_t*>(a.pool_argmax + o * EPC); am[di][dj][1] = 0; }
      }
    u32x4_t xv[2][2];
    bool pok[2][2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        const int h = 2 * bh + pi, w = 2 * bw + pj;
        pok[pi][pj] = h < a.pH && w < a.pW;
        const int hc = h < a.pH ? h : a.pH - 1, wc = w < a.pW ? w : a.pW - 1;
        xv[pi][pj] = ld16_nt(xg + (((n * a.pH + hc) * a.pW + wc) * cols + col) * 16);
      }
    float dw[2][2][EPC];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) Elem<T>::unpack(dv[di][dj], dw[di][dj]);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        float xf[EPC], g[EPC], d[EPC];
        Elem<T>::unpack(xv[pi][pj], xf);
#pragma unroll
        for (int e = 0; e < EPC; ++e) g[e] = 0.f;
#pragma unroll
        for (int di = 0; di <= pi; ++di)
#pragma unroll
          for (int dj = 0; dj <= pj; ++dj) {
            const uint32_t code = (uint32_t)((pi - 2 * di + 1) * 3 + (pj - 2 * dj + 1));
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              if (wok[di][dj] && ((am[di][dj][e >> 2] >> (8 * (e & 3))) & 0xffu) == code) g[e] += dw[di][dj][e];
          }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          if (a.relu_from_x && !(fmaf(xf[e], cA[e], rsh[e]) > 0.f)) g[e] = 0.f;   // ReLU between the BatchNorm and the pool
          d[e] = fmaf(cA[e], g[e], fmaf(cB[e], xf[e], cC[e]));
        }
        if (pok[pi][pj]) {
          const size_t o = (((n * a.pH + 2 * bh + pi) * a.pW + 2 * bw + pj) * cols + col) * 16;
          st16_nt(dxg + o, Elem<T>::pack(d));
          if (a.gout) st16_nt(reinterpret_cast<char*>(a.gout) + o, Elem<T>::pack(g));
        }
      }
  }
}

// sums[segment][2][C] = the workgroups' rows added in ONE fixed order: block = 64 consecutive values of a [2][C] row (x) by 16 row
// lanes (y); row lane y adds rows y, y + 16, ... in rising order, lane 0 adds the 16 lane sums in lane order and writes.
__global__ __launch_bounds__(1024) void bn_bwd_sums_kernel(const double* __restrict__ rows, int nrows, int C2, double* __restrict__ sums, int sums_stride) {
  __shared__ double red[16][64];
  const int v = blockIdx.x * 64 + threadIdx.x, yl = threadIdx.y;
  const bool ok = v < C2;
  const double* p = rows + ((size_t)blockIdx.y * nrows) * C2 + v;
  double s = 0.0;
  if (ok) {
#pragma unroll 16
    for (int r = yl; r < nrows; r += 16) s += __builtin_nontemporal_load(p + (size_t)r * C2);     // (loads independent: all in flight at once)
  }
  red[yl][threadIdx.x] = s;
  __syncthreads();
  if (yl != 0 || !ok) return;
  for (int y = 1; y < 16; ++y) s += red[y][threadIdx.x];
  sums[(size_t)blockIdx.y * sums_stride + v] = s;
}

// the reduce pass: per-workgroup rows, then their ordered sum OVERWRITES a.sums (no atomics: the same bits run after run)
hipError_t launch_bn_bwd_reduce(int dtype, const BnBwdArgs& a, hipStream_t st) {
  const int epc = dtype == DT_BF16 ? 8 : 4;
  const int cols = a.C / epc;
  if (cols > 256 || (256 % cols) != 0) return hipErrorInvalidValue;
  const int nseg = a.nseg > 1 ? a.nseg : 1;      // grid y; the kernels take their segment's share (bn_bwd_segment)
  if (nseg > 1 && (a.pool_dy || a.pixels % nseg != 0)) return hipErrorInvalidValue;
  const size_t pixels = a.pixels / nseg;
  const int rpp = 256 / cols;
  size_t blocks = (pixels + rpp - 1) / rpp;
  blocks = (blocks + 7) / 8;                              // >= 8 passes per block
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const bool big = blocks >= 512 && !(a.pool_dy && a.pool_y);      // big tensors: a quarter of the workgroups, four times the threads each
  size_t b4 = 0;
  if (big) {
    b4 = ((pixels + 1024 / cols - 1) / (1024 / cols) + 7) / 8;
    if (b4 > 256) b4 = 256;
  }
  const int nrows = (int)(big ? b4 : blocks);
  double* rows = reinterpret_cast<double*>(stream_scratch(st, (size_t)nseg * nrows * 2 * a.C * sizeof(double)));
  if (!rows) return hipErrorOutOfMemory;
  auto fold = [&]() {
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(2 * a.C, 64), nseg), dim3(64, 16), 0, st, rows, nrows, 2 * a.C, a.sums, nseg > 1 ? a.sums_stride : 0);
    return hipGetLastError();
  };
  if (a.pool_dy && a.pool_y) {
    if (dtype == DT_BF16) hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, st, a, rows);
    else hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<float>, dim3((int)blocks), dim3(256), 0, st, a, rows);
    return fold();
  }
  if (big) {
    constexpr int NB = 1024;
    const size_t lds = (size_t)NB * (2 * epc + 1) * sizeof(float);
    static std::atomic<bool> attr_done{false};
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_bwd_reduce_kernel<bf16_t, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_bwd_reduce_kernel<float, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_done = true;
    }
    if (dtype == DT_BF16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 1024>), dim3((int)b4, nseg), dim3(NB), lds, st, a, rows);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1024>), dim3((int)b4, nseg), dim3(NB), lds, st, a, rows);
    return fold();
  }
  const size_t lds = (size_t)256 * (2 * epc + 1) * sizeof(float);
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 256>), dim3((int)blocks, nseg), dim3(256), lds, st, a, rows);
  } else {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 256>), dim3((int)blocks, nseg), dim3(256), lds, st, a, rows);
  }
  return fold();
}

// the two-BatchNorm forms (bn_bwd_reduce_pair_kernel): a = the residual branch's bn2 (g_in_reduce: its reduce pass writes g to gout),
// b = the projection shortcut's BatchNorm on the same g.  b.sums must be a.sums + 2 C (one fold launch, one all-reduce for both).
bool bn_bwd_pair_ok(const BnBwdArgs& a, const BnBwdArgs& b) {
  static const bool on = [] { const char* e = getenv("SSLCR_BN_PAIR"); return !e || atoi(e) != 0; }();     // 0: two passes each (A/B runs)
  return on && a.g_in_reduce && a.gout && b.dy == a.gout && !b.yact && !b.yact_bits && !b.relu_from_x && !b.gout && !a.pool_dy && !b.pool_dy &&
         a.nseg <= 1 && b.nseg <= 1 && a.C == b.C && a.pixels == b.pixels && b.sums == a.sums + 2 * a.C && a.dx && b.dx && b.x && a.x;
}
hipError_t launch_bn_bwd_reduce_pair(int dtype, const BnBwdArgs& a, const BnBwdArgs& b, int keep_g, hipStream_t st) {
  const int epc = dtype == DT_BF16 ? 8 : 4;
  const int cols = a.C / epc;
  if (cols > 256 || (256 % cols) != 0 || !bn_bwd_pair_ok(a, b)) return hipErrorInvalidValue;
  // the grid of launch_bn_bwd_reduce for this tensor (same rows, same per-thread order: the same sums)
  const int rpp = 256 / cols;
  size_t blocks = (a.pixels + rpp - 1) / rpp;
  blocks = (blocks + 7) / 8;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const bool big = blocks >= 512;
  size_t b4 = 0;
  if (big) {
    b4 = ((a.pixels + 1024 / cols - 1) / (1024 / cols) + 7) / 8;
    if (b4 > 256) b4 = 256;
  }
  const int nrows = (int)(big ? b4 : blocks);
  double* rows = reinterpret_cast<double*>(stream_scratch(st, (size_t)2 * nrows * 2 * a.C * sizeof(double)));
  if (!rows) return hipErrorOutOfMemory;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_bwd_reduce_pair_kernel<bf16_t, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_bwd_reduce_pair_kernel<float, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_done = true;
  }
  const int NB = big ? 1024 : 256;
  const size_t lds = (size_t)NB * (3 * epc + 1) * sizeof(float);
  if (big) {
    if (dtype == DT_BF16) hipLaunchKernelGGL((bn_bwd_reduce_pair_kernel<bf16_t, 1024>), dim3(nrows), dim3(1024), lds, st, a, b, rows, keep_g);
    else hipLaunchKernelGGL((bn_bwd_reduce_pair_kernel<float, 1024>), dim3(nrows), dim3(1024), lds, st, a, b, rows, keep_g);
  } else {
    if (dtype == DT_BF16) hipLaunchKernelGGL((bn_bwd_reduce_pair_kernel<bf16_t, 256>), dim3(nrows), dim3(256), lds, st, a, b, rows, keep_g);
    else hipLaunchKernelGGL((bn_bwd_reduce_pair_kernel<float, 256>), dim3(nrows), dim3(256), lds, st, a, b, rows, keep_g);
  }
  // one fold for both BatchNorms: "segment" y = the BatchNorm, sums 2 C doubles apart
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(2 * a.C, 64), 2), dim3(64, 16), 0, st, rows, nrows, 2 * a.C, a.sums, 2 * a.C);
  return hipGetLastError();
}
hipError_t launch_bn_bwd_apply_pair(int dtype, const BnBwdArgs& a, const BnBwdArgs& b, int from_g, hipStream_t st) {
  if (!bn_bwd_pair_ok(a, b)) return hipErrorInvalidValue;
  if (dtype == DT_BF16) hipLaunchKernelGGL(bn_bwd_apply_pair_kernel<bf16_t>, dim3(ew_grid(a.pixels * (a.C / 8))), dim3(256), 0, st, a, b, from_g);
  else hipLaunchKernelGGL(bn_bwd_apply_pair_kernel<float>, dim3(ew_grid(a.pixels * (a.C / 4))), dim3(256), 0, st, a, b, from_g);
  return hipGetLastError();
}

hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a0, hipStream_t st) {
  BnBwdArgs a = a0;
  if (a.g_in_reduce) {      // the reduce pass left the masked gradient in gout: plain dy from here on
    a.dy = a.gout; a.yact = nullptr; a.yact_bits = nullptr; a.relu_from_x = 0; a.gout = nullptr; a.g_in_reduce = 0;
  }
  if (a.pool_dy && a.nseg > 1) return hipErrorInvalidValue;
  if (a.pool_dy) {
    const int epc = dtype == DT_BF16 ? 8 : 4;
    const size_t items = (a.pixels / ((size_t)a.pH * a.pW)) * ((a.pH + 1) / 2) * ((a.pW + 1) / 2) * (a.C / epc);
    if (dtype == DT_BF16) hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<bf16_t>, dim3(ew_grid(items)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<float>, dim3(ew_grid(items)), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  if (a.pixels % nseg != 0) return hipErrorInvalidValue;
  const size_t per = a.pixels / nseg;
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(ew_grid(per * (a.C / 8)), nseg), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid(per * (a.C / 4)), nseg), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

__global__ void bn_param_grads_kernel(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, float scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dgamma[c] += (float)(sums[C + c] * (double)invstd[c] * (double)scale);
  dbeta[c] += (float)(sums[c] * (double)scale);
}
hipError_t launch_bn_param_grads_scaled(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, float scale, hipStream_t st) {
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, sums, invstd, dgamma, dbeta, C, scale);
  return hipGetLastError();
}
hipError_t launch_bn_param_grads(const double* sums, const float* invstd, float* dgamma, float* dbeta, int C, hipStream_t st) {
  return launch_bn_param_grads_scaled(sums, invstd, dgamma, dbeta, C, 1.f, st);
}

}  // namespace sslcr

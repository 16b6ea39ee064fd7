// Fused multi-tensor optimizers, Lookahead/EMA axpby and weight packing (incl. eval-mode BN folding).
// Replaces torch.optim.Adam / SGD(nesterov) per-tensor kernel storms, Lookahead (lookahead.py:81-106),
// teacher refresh (K15) and prepares the KRSC / CRSK shadow weights the conv kernels consume.
#include "kernels.hpp"

namespace sslcr {

// one parameter element: i indexes p / s1 / s2, gi the gradient.  Returns the new value.
__device__ __forceinline__ float opt_update_g(const TensorDesc& d, const OptArgs& o, int i, float graw);
__device__ __forceinline__ float opt_update(const TensorDesc& d, const OptArgs& o, int i, int gi) {
  return opt_update_g(d, o, i, d.g[gi]);
}
// ... with the raw gradient element already in hand
__device__ __forceinline__ float opt_update_g(const TensorDesc& d, const OptArgs& o, int i, float graw) {
  float p = d.p[i];
  const float g = fmaf(o.wd, p, graw * o.grad_scale);
  if (o.kind == 0) {                  // Adam, L2 decay in the gradient, eps outside the sqrt
    float m = d.s1[i], v = d.s2[i];
    m = fmaf(o.beta1, m, (1.f - o.beta1) * g);
    v = fmaf(o.beta2, v, (1.f - o.beta2) * g * g);
    d.s1[i] = m;
    d.s2[i] = v;
    const float denom = sqrtf(v) / sqrtf(o.bc2) + o.eps;
    p -= (o.lr / o.bc1) * (m / denom);
  } else {                            // SGD momentum, nesterov
    float buf = o.first_step ? g : fmaf(o.momentum, d.s1[i], g);
    d.s1[i] = buf;
    p -= o.lr * fmaf(o.momentum, buf, g);
  }
  d.p[i] = p;
  return p;
}
__device__ __forceinline__ void opt_pack(const TensorDesc& d, int k, int c, int rs, int gi, float p) {
  if (d.w_fwd) {
    if (d.pack_dtype == DT_BF16) Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(d.w_fwd) + gi, p);
    else reinterpret_cast<float*>(d.w_fwd)[gi] = p;
  }
  if (d.w_dgrad) {
    const size_t o = ((size_t)c * d.RS + (d.dgrad_flip ? d.RS - 1 - rs : rs)) * d.K + k;
    if (d.pack_dtype == DT_BF16) Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(d.w_dgrad) + o, p);
    else reinterpret_cast<float*>(d.w_dgrad)[o] = p;
  }
}

// generic form (sslcr_optimizer_step): a fixed number of workgroups per tensor
__global__ __launch_bounds__(256) void optimizer_kernel(const TensorDesc* __restrict__ descs, const OptArgs o) {
  const TensorDesc d = descs[blockIdx.y];
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += stride) {
    int gi = i, k = 0, c = 0, rs = 0;
    if (d.K > 0) {                      // param [K][C][RS]  <-  grad [K][RS][C]
      const int crs = d.C * d.RS;
      k = i / crs;
      const int rem = i - k * crs;
      c = rem / d.RS; rs = rem - c * d.RS;
      gi = (k * d.RS + rs) * d.C + c;
    }
    const float p = opt_update(d, o, i, gi);
    if (d.K > 0) opt_pack(d, k, c, rs, gi, p);
  }
}

hipError_t launch_optimizer(const TensorDesc* d_descs, int ntensors, int max_n, const OptArgs& o, hipStream_t st) {
  int bx = cdiv(max_n, 256 * 8);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(optimizer_kernel, dim3(bx, ntensors), dim3(256), 0, st, d_descs, o);
  return hipGetLastError();
}

// engine form: a work list of chunks, so that the four 2.4 M-element layer4 filters (85 % of the parameters) get 85 % of the
// workgroups instead of 64 each.  Chunks run in PARAMETER order (p, s1, s2 coalesced; the gradient gather and the two
// bf16 shadow-weight writes are the strided streams).  Gradient order -- one output channel per chunk, c fastest -- was
// measured 0.3 ms/step slower: the 36-byte stride it puts on the three fp32 state arrays costs more than it saves.
// The shadow weights of the updated value are written here, which removes the 19 pack launches that followed every step
// (step 20.33 -> 20.24 ms).
__global__ __launch_bounds__(256) void optimizer_chunks_kernel(const TensorDesc* __restrict__ descs, const int2* __restrict__ chunks,
                                                               const OptArgs o) {
  const int2 ch = chunks[blockIdx.x];
  const TensorDesc d = descs[ch.x];
  if (ch.y < 0) {
    // 3x3 filter with shadow weights, tile (16 kout) x (16 cin) x 9 taps = 2304 elements through LDS, so that EVERY stream is
    // a run of >= 32 bytes: gradient [K][9][C] in (64-byte runs of 16 cin), parameter / moments [K][C][9] (576-byte runs),
    // forward weights [K][9][C] and dgrad weights [C][9][K] out (32-byte runs).  In parameter order three of the five
    // streams were 2- and 4-byte accesses a filter row apart (the kernel ran at 1.6 TB/s).
    __shared__ float sm[16 * 145];               // parameter order, one pad word per kout: kl * 145 + cl * 9 + rs
    const int tile = -ch.y - 1, ctiles = d.C / 16;
    const int k0 = (tile / ctiles) * 16, c0 = (tile % ctiles) * 16;
#pragma unroll
    for (int j = 0; j < OPT_TILE / 256; ++j) {     // gradient order in
      const int e = threadIdx.x + 256 * j;
      const int cl = e & 15, t2 = e >> 4, kl = t2 / 9, rs = t2 - kl * 9;
      sm[kl * 145 + cl * 9 + rs] = d.g[((size_t)(k0 + kl) * 9 + rs) * d.C + c0 + cl];
    }
    __syncthreads();
    float pv[OPT_TILE / 256];
#pragma unroll
    for (int j = 0; j < OPT_TILE / 256; ++j) {     // parameter order: the update
      const int e = threadIdx.x + 256 * j;
      const int kl = e / 144, rem = e - kl * 144;
      pv[j] = opt_update_g(d, o, ((k0 + kl) * d.C + c0) * 9 + rem, sm[e + kl]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OPT_TILE / 256; ++j) { const int e = threadIdx.x + 256 * j; sm[e + e / 144] = pv[j]; }
    __syncthreads();
    const bool bf = d.pack_dtype == DT_BF16;
#pragma unroll
    for (int j = 0; j < OPT_TILE / 256; ++j) {     // forward weights out: [k][rs][c]
      const int e = threadIdx.x + 256 * j;
      const int cl = e & 15, t2 = e >> 4, kl = t2 / 9, rs = t2 - kl * 9;
      const float p = sm[kl * 145 + cl * 9 + rs];
      const size_t oo = ((size_t)(k0 + kl) * 9 + rs) * d.C + c0 + cl;
      if (bf) Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(d.w_fwd) + oo, p);
      else reinterpret_cast<float*>(d.w_fwd)[oo] = p;
    }
#pragma unroll
    for (int j = 0; j < OPT_TILE / 256; ++j) {     // dgrad weights out: [c][rs'][k]
      const int e = threadIdx.x + 256 * j;
      const int kl = e & 15, t2 = e >> 4, cl = t2 / 9, rs = t2 - cl * 9;
      const float p = sm[kl * 145 + cl * 9 + rs];
      const size_t oo = ((size_t)(c0 + cl) * 9 + (d.dgrad_flip ? 8 - rs : rs)) * d.K + k0 + kl;
      if (bf) Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(d.w_dgrad) + oo, p);
      else reinterpret_cast<float*>(d.w_dgrad)[oo] = p;
    }
  } else if (d.K > 0) {
    // parameter order (p, s1, s2 coalesced); the gradient gather and the two shadow-weight writes are the strided streams
    const int crs = d.C * d.RS;
    const int end = ch.y + OPT_CHUNK < d.n ? ch.y + OPT_CHUNK : d.n;
    for (int i = ch.y + threadIdx.x; i < end; i += 256) {
      const int k = i / crs, rem = i - k * crs;
      const int c = rem / d.RS, rs = rem - c * d.RS;
      const int gi = (k * d.RS + rs) * d.C + c;
      opt_pack(d, k, c, rs, gi, opt_update(d, o, i, gi));
    }
  } else {
    const int end = ch.y + OPT_CHUNK < d.n ? ch.y + OPT_CHUNK : d.n;
    for (int i = ch.y + threadIdx.x; i < end; i += 256) opt_update(d, o, i, i);
  }
}
hipError_t launch_optimizer_chunks(const TensorDesc* d_descs, const void* d_chunks, int nchunks, const OptArgs& o, hipStream_t st) {
  if (nchunks < 1) return hipSuccess;
  hipLaunchKernelGGL(optimizer_chunks_kernel, dim3(nchunks), dim3(256), 0, st, d_descs, reinterpret_cast<const int2*>(d_chunks), o);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void axpby_kernel(float* p, float* q, size_t n, float alpha, int copy_back) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = alpha * p[i] + (1.f - alpha) * q[i];
    p[i] = v;
    if (copy_back) q[i] = v;
  }
}
hipError_t launch_axpby(float* p, float* q, size_t n, float alpha, int copy_back, hipStream_t st) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(axpby_kernel, dim3((int)b), dim3(256), 0, st, p, q, n, alpha, copy_back);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t st) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(fill_kernel, dim3((int)b), dim3(256), 0, st, p, n, v);
  return hipGetLastError();
}

// ------------------------------------------------------------------ weight packing
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_kernel(const PackArgs a) {
  const int RS = a.R * a.S;
  const int total = a.K * a.C * RS;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    // i enumerates the fwd layout [K][RS][C]
    const int k = i / (RS * a.C), rem = i - k * RS * a.C;
    const int rs = rem / a.C, c = rem - rs * a.C;
    const float w = a.w[((size_t)k * a.C + c) * RS + rs];
    if (a.w_fwd) {
      float f = 1.f;
      if (a.gamma && !a.scale_out) f = a.gamma[k] / sqrtf(a.rvar[k] + a.eps);      // (scale_out: the scale goes to the conv's epilogue)
      Elem<T>::st(reinterpret_cast<T*>(a.w_fwd) + i, w * f);
    }
    if (a.w_dgrad) Elem<T>::st(reinterpret_cast<T*>(a.w_dgrad) + ((size_t)c * RS + (a.dgrad_flip ? RS - 1 - rs : rs)) * a.K + k, w);
  }
  if (a.gamma && a.bias_out) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < a.K; k += gridDim.x * 256) {
      const float f = a.gamma[k] / sqrtf(a.rvar[k] + a.eps);
      a.bias_out[k] = a.beta[k] - a.rmean[k] * f;
      if (a.scale_out) a.scale_out[k] = f;
    }
  }
}
hipError_t launch_pack_conv(int dtype, const PackArgs& a, hipStream_t st) {
  int b = cdiv(a.K * a.C * a.R * a.S, 256);
  if (b > 2048) b = 2048;
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(b), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(b), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// stem: [64][3][7][7] -> [64][7][8][4] (s = 7 and c = 3 are zero padding of the MFMA K dimension)
template <typename T>
__global__ __launch_bounds__(256) void pack_stem_kernel(const PackArgs a) {
  const int total = 64 * 7 * 8 * 4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % 7, k = i / 224;
    float w = 0.f;
    if (c < 3 && s < 7) {
      w = a.w[((k * 3 + c) * 7 + r) * 7 + s];
      if (a.gamma && !a.scale_out) w *= a.gamma[k] / sqrtf(a.rvar[k] + a.eps);
    }
    Elem<T>::st(reinterpret_cast<T*>(a.w_fwd) + i, w);
  }
  if (a.gamma && a.bias_out && blockIdx.x == 0 && threadIdx.x < 64) {
    const int k = threadIdx.x;
    const float f = a.gamma[k] / sqrtf(a.rvar[k] + a.eps);
    a.bias_out[k] = a.beta[k] - a.rmean[k] * f;
    if (a.scale_out) a.scale_out[k] = f;
  }
}
hipError_t launch_pack_stem(int dtype, const PackArgs& a, hipStream_t st) {
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(pack_stem_kernel<bf16_t>, dim3(56), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(pack_stem_kernel<float>, dim3(56), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

}  // namespace sslcr

namespace sslcr {

// dst[r*ldd + c] (+)= src[r*lds + c], c < w   (concat / tile / slice-sum glue for the fp32 heads)
__global__ __launch_bounds__(256) void copy2d_kernel(float* dst, long ldd, const float* src, long lds, int rows, int w, int accumulate) {
  const long total = (long)rows * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / w, c = i - r * w;
    const float v = src[r * lds + c];
    float* d = dst + r * ldd + c;
    *d = accumulate ? *d + v : v;
  }
}
// dst[j*dstep + r*ldd + c] = sum_i src[i*sstep + r*lds + c] for j < ndst, i < nsrc, left to right: the tile ([f, f, f],
// [E, E]) and slice-sum (their gradients) forms of the single-branch head in ONE launch each instead of two or three
__global__ __launch_bounds__(256) void copy2d_multi_kernel(float* dst, long ldd, int ndst, long dstep, const float* src, long lds, int nsrc,
                                                           long sstep, int rows, int w) {
  const long total = (long)rows * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / w, c = i - r * w;
    float v = src[r * lds + c];
    for (int s = 1; s < nsrc; ++s) v += src[s * sstep + r * lds + c];
    for (int j = 0; j < ndst; ++j) dst[j * dstep + r * ldd + c] = v;
  }
}
hipError_t launch_copy2d_multi(float* dst, long ldd, int ndst, long dstep, const float* src, long lds, int nsrc, long sstep, int rows, int w,
                               hipStream_t st) {
  long b = ((long)rows * w + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(copy2d_multi_kernel, dim3((int)b), dim3(256), 0, st, dst, ldd, ndst, dstep, src, lds, nsrc, sstep, rows, w);
  return hipGetLastError();
}
hipError_t launch_copy2d(float* dst, long ldd, const float* src, long lds, int rows, int w, int accumulate, hipStream_t st) {
  long b = ((long)rows * w + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(copy2d_kernel, dim3((int)b), dim3(256), 0, st, dst, ldd, src, lds, rows, w, accumulate);
  return hipGetLastError();
}

// engine gradient layout [K][RS][C] -> PyTorch [K][C][RS]
__global__ __launch_bounds__(256) void unpack_grad_kernel(const float* g, float* out, int K, int C, int RS) {
  const int total = K * C * RS;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i / (C * RS), rem = i - k * C * RS;
    const int c = rem / RS, rs = rem - c * RS;
    out[i] = g[(k * RS + rs) * C + c];
  }
}
hipError_t launch_unpack_grad(const float* g, float* out, int K, int C, int RS, hipStream_t st) {
  int b = cdiv(K * C * RS, 256);
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(unpack_grad_kernel, dim3(b), dim3(256), 0, st, g, out, K, C, RS);
  return hipGetLastError();
}

}  // namespace sslcr

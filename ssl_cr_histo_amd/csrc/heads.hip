// fc / classifier heads (exact fp32 on v_mfma_f32_16x16x4_f32) and the fused loss forward+backward.
// Replaces nn.Linear/ReLU of models/net.py:12-15,35-36,111 (K10/K11) and F.mse_loss / F.cross_entropy /
// softmax+max pseudo-labels of eval_BreastPathQ_SSL_CR.py:92-95, eval_Camelyon_SSL_CR.py:110-116 (K12).
#include "kernels.hpp"

namespace sslcr {

// C[M][N] (+)= A(M x K) * B(K x N); element (i,k) of A at A[i*sa_i + k*sa_k], (k,j) of B at B[k*sb_k + j*sb_j].
// One wave per 16x16 tile; lane (li, g) feeds k = kb + 4g + e for MFMA e (any k-permutation is a valid sum order).
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sa_i, long sa_k,
                                                       const float* __restrict__ B, long sb_k, long sb_j,
                                                       float* __restrict__ C, int M, int N, int K,
                                                       const float* __restrict__ bias, int relu, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  // one 16x16 tile per workgroup, K split over its four waves in interleaved 64-wide chunks and folded through LDS: these
  // GEMMs are latency-bound (a dependent memory round trip per 64 of K, M*N of a few hundred tiles), and a wave per tile
  // walked K = 1024 in 16 serial trips
  __shared__ float red[3][64][4];
  const int tiles_n = (N + 15) / 16;
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int i = tm * 16 + li, j = tn * 16 + li;
  const bool iv = i < M, jv = j < N;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  // 64 of K per trip: all 32 operand values are requested before the first MFMA (one k-step per trip left the wave waiting
  // a full memory round trip per 4 MFMAs); operands contiguous in k are fetched as one 16-byte load per lane
  const bool avec = sa_k == 1 && (sa_i & 3) == 0 && ((size_t)A & 15) == 0;
  const bool bvec = sb_k == 1 && (sb_j & 3) == 0 && ((size_t)B & 15) == 0;
  for (int kb = wave * 64; kb < K; kb += 256) {
    float a[4][4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb + 16 * u + 4 * g;
      if (avec && iv && k0 + 3 < K) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(A + (long)i * sa_i + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[u][e] = v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[u][e] = (iv && k0 + e < K) ? A[(long)i * sa_i + (long)(k0 + e) * sa_k] : 0.f;
      }
      if (bvec && jv && k0 + 3 < K) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(B + (long)j * sb_j + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = (jv && k0 + e < K) ? B[(long)(k0 + e) * sb_k + (long)j * sb_j] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[w][lane][r];
  // D[row = 4g + r][col = li]
  if (jv) {
    const float bj = bias ? bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + 4 * g + r;
      if (row < M) {
        float v = acc[r] + bj;
        float* c = C + (long)row * N + j;
        if (accumulate) v += *c;
        if (relu) v = fmaxf(v, 0.f);
        *c = v;
      }
    }
  }
}

// LDS-staged form for the shapes that matter (M, N multiples of 32, K of 64; round 6).  What bounded the form above at 22 us for
// fc.0 (0.67 GFLOP: 30 TF/s on a 157 TF/s pipe) is the operand FETCH, not its volume: a lane loads its own row's 16 bytes, so one
// load instruction touches 64 separate 16-byte segments of 16 rows and the CU's address coalescer serialises them (2x2 tiles per wave,
// operand prefetch and both together changed nothing -- 22-25 us each: tools/heads_bench.py, profiles/r06_heads_gemm.txt).  Here a workgroup owns a
// 32 x 32 block of C and walks K in chunks of 64: the two 32 x 64 operand tiles are fetched with whole 256-byte rows per 16 lanes
// (k-contiguous operands, AKC / BKC) or 128-byte rows per 8 lanes (k-strided ones, transposed on their way into LDS), double-buffered
// through LDS with one barrier per chunk, and wave (wm, wn) runs the 16 x 16 tile's MFMAs from 16-byte fragment reads (row pitch 68
// floats: the 16 rows of a fragment read hit 64 different banks).  Each output element sums its K in rising chunk order.
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_lds_kernel(const float* __restrict__ A, long sa, const float* __restrict__ B, long sb,
                                                           float* __restrict__ C, int M, int N, int K,
                                                           const float* __restrict__ bias, int relu, int accumulate) {
  constexpr int PITCH = 68;                                    // floats per LDS row (64 k + 4)
  __shared__ __attribute__((aligned(16))) float lds[2][2][32 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / 32;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x - bm * tiles_n;
  // sa / sb: the stride of the operand's NON-contiguous index (k-contiguous: elements per row i / j; k-strided: elements per k)
  const float* Ab = AKC ? A + (long)(bm * 32) * sa : A + bm * 32;
  const float* Bb = BKC ? B + (long)(bn * 32) * sb : B + bn * 32;
  // operand registers of TWO chunks: chunk ch + 2 is requested at the top of chunk ch and staged at the end of chunk ch + 1 (with one
  // chunk of lead the ~600 cycles of a chunk's MFMAs covered a third of the fetch: 15.5 us for fc.0 against a 3.4 us MFMA floor)
  f32x4_t ra[2][2], rb[2][2];
  auto fetch = [&](int kb, f32x4_t (&xa)[2], f32x4_t (&xb)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int c = tid + 256 * p;
      if (AKC) xa[p] = *reinterpret_cast<const f32x4_t*>(Ab + (long)(c >> 4) * sa + kb + 4 * (c & 15));
      else xa[p] = *reinterpret_cast<const f32x4_t*>(Ab + (long)(kb + (c >> 3)) * sa + 4 * (c & 7));
      if (BKC) xb[p] = *reinterpret_cast<const f32x4_t*>(Bb + (long)(c >> 4) * sb + kb + 4 * (c & 15));
      else xb[p] = *reinterpret_cast<const f32x4_t*>(Bb + (long)(kb + (c >> 3)) * sb + 4 * (c & 7));
    }
  };
  auto stage = [&](int buf, const f32x4_t (&xa)[2], const f32x4_t (&xb)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int c = tid + 256 * p;
      if (AKC) *reinterpret_cast<f32x4_t*>(&lds[buf][0][(c >> 4) * PITCH + 4 * (c & 15)]) = xa[p];
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[buf][0][(4 * (c & 7) + e) * PITCH + (c >> 3)] = xa[p][e];
      }
      if (BKC) *reinterpret_cast<f32x4_t*>(&lds[buf][1][(c >> 4) * PITCH + 4 * (c & 15)]) = xb[p];
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[buf][1][(4 * (c & 7) + e) * PITCH + (c >> 3)] = xb[p][e];
      }
    }
  };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const float* la = &lds[buf][0][(wm * 16 + li) * PITCH + 4 * g];
    const float* lb = &lds[buf][1][(wn * 16 + li) * PITCH + 4 * g];
    f32x4_t fa[4], fb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fa[u] = *reinterpret_cast<const f32x4_t*>(la + 16 * u);
      fb[u] = *reinterpret_cast<const f32x4_t*>(lb + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[u][e], acc, 0, 0, 0);
  };
  const int nchunks = K / 64;
  fetch(0, ra[0], rb[0]);
  if (nchunks > 1) fetch(64, ra[1], rb[1]);
  stage(0, ra[0], rb[0]);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ch += 2) {
    // even chunk: LDS buffer 0, registers set 0 is free (staged), set 1 holds chunk ch + 1
    if (ch + 2 < nchunks) fetch((ch + 2) * 64, ra[0], rb[0]);
    compute(0);
    if (ch + 1 < nchunks) stage(1, ra[1], rb[1]);
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    if (ch + 3 < nchunks) fetch((ch + 3) * 64, ra[1], rb[1]);
    compute(1);
    if (ch + 2 < nchunks) stage(0, ra[0], rb[0]);
    __syncthreads();
  }
  // D[row = 4g + q][col = li]
  const int j = bn * 32 + wn * 16 + li;
  const float bj = bias ? bias[j] : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = bm * 32 + wm * 16 + 4 * g + q;
    float v = acc[q] + bj;
    float* cp = C + (long)row * N + j;
    if (accumulate) v += *cp;
    if (relu) v = fmaxf(v, 0.f);
    *cp = v;
  }
}

static hipError_t gemm(const float* A, long sa_i, long sa_k, const float* B, long sb_k, long sb_j, float* C, int M, int N, int K,
                       const float* bias, int relu, int accumulate, hipStream_t st) {
  static const bool lds_on = [] { const char* e = getenv("SSLCR_GEMM_LDS"); return !e || atoi(e) != 0; }();     // 0: the direct-fetch form for everything (A/B runs)
  const bool akc = sa_k == 1, bkc = sb_k == 1;
  const long sa = akc ? sa_i : sa_k, sb = bkc ? sb_j : sb_k;
  const bool lds_ok = lds_on && M % 32 == 0 && N % 32 == 0 && K % 64 == 0 && (akc || sa_i == 1) && (bkc || sb_j == 1) && (sa & 3) == 0 &&
                      (sb & 3) == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0;
  if (lds_ok) {
    const dim3 grid((M / 32) * (N / 32));
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_lds_kernel<true, true>), grid, dim3(256), 0, st, A, sa, B, sb, C, M, N, K, bias, relu, accumulate);
    else if (akc) hipLaunchKernelGGL((gemm_f32_lds_kernel<true, false>), grid, dim3(256), 0, st, A, sa, B, sb, C, M, N, K, bias, relu, accumulate);
    else if (bkc) hipLaunchKernelGGL((gemm_f32_lds_kernel<false, true>), grid, dim3(256), 0, st, A, sa, B, sb, C, M, N, K, bias, relu, accumulate);
    else hipLaunchKernelGGL((gemm_f32_lds_kernel<false, false>), grid, dim3(256), 0, st, A, sa, B, sb, C, M, N, K, bias, relu, accumulate);
    return hipGetLastError();
  }
  const int tiles = cdiv(M, 16) * cdiv(N, 16);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, st, A, sa_i, sa_k, B, sb_k, sb_j, C, M, N, K, bias, relu, accumulate);
  return hipGetLastError();
}

hipError_t launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, hipStream_t st) {
  // y = x[M][K] * w[N][K]^T  ->  B(k,j) = w[j*K + k]
  return gemm(x, K, 1, w, 1, K, y, M, N, K, b, relu, 0, st);
}

__global__ __launch_bounds__(1024) void relu_mask_colsum_kernel(const float* dy, const float* yact, float* dym, float* db, int M, int N) {
  // block = 16 columns x 64 row lanes over ALL rows: masks dy by (yact > 0) into dym; the column sums are added in a fixed order
  // (row lane by row lane, then the 64 lane sums in lane order) into db by their single owner -- the same bits run after run
  __shared__ float sm[64][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (n < N) {
    for (int m = rl; m < M; m += 64) {
      float v = dy[(long)m * N + n];
      if (yact && !(yact[(long)m * N + n] > 0.f)) v = 0.f;
      if (dym) dym[(long)m * N + n] = v;
      s += v;
    }
  }
  sm[rl][cl] = s;
  __syncthreads();
  if (db && rl == 0 && n < N) {
    float t = sm[0][cl];
    for (int r = 1; r < 64; ++r) t += sm[r][cl];
    db[n] += t;
  }
}

hipError_t launch_linear_bwd(const float* x, const float* w, const float* dy, const float* yact, float* dx, float* dw, float* db,
                             int M, int N, int K, int dx_accumulate, float* scratch, hipStream_t st) {
  const float* g = dy;
  if (yact || db) {
    hipLaunchKernelGGL(relu_mask_colsum_kernel, dim3(cdiv(N, 16)), dim3(1024), 0, st, dy, yact, yact ? scratch : nullptr, db, M, N);
    if (yact) g = scratch;
  }
  hipError_t e = hipSuccess;
  if (dx) {   // dx[M][K] = g[M][N] * w[N][K]
    e = gemm(g, N, 1, w, K, 1, dx, M, K, N, nullptr, 0, dx_accumulate, st);
    if (e != hipSuccess) return e;
  }
  if (dw) {   // dw[N][K] += g^T[N][M] * x[M][K]
    e = gemm(g, 1, N, x, K, 1, dw, N, K, M, nullptr, 0, 1, st);
  }
  return e;
}

// ------------------------------------------------------------------ losses (single block; a few thousand rows at most)
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ int row_argmax(const float* l, int C) {
  int b = 0;
  float m = l[0];
  for (int c = 1; c < C; ++c)
    if (l[c] > m) { m = l[c]; b = c; }
  return b;
}

// returns -log softmax(l)[y]; writes (softmax - onehot)*w into dl when dl != null
__device__ __forceinline__ float ce_row(const float* l, int C, int y, float* dl, float w) {
  float m = l[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(l[c] - m);
  const float lse = m + logf(s);
  if (dl)
    for (int c = 0; c < C; ++c) dl[c] = (expf(l[c] - lse) - (c == y ? 1.f : 0.f)) * w;
  return lse - l[y];
}

__global__ __launch_bounds__(256) void loss_kernel(const LossArgs a) {
  __shared__ float sm[4];
  const int C = a.C;
  const bool ce = (a.kind == 1 || a.kind == 2);
  float lx = 0.f, lu = 0.f, correct = 0.f;
  for (int i = threadIdx.x; i < a.nx; i += 256) {
    const float* l = a.logits + (long)i * C;
    float* dl = a.dlogits ? a.dlogits + (long)i * C : nullptr;
    if (ce) {
      const int y = (int)a.target_i[i];
      lx += ce_row(l, C, y, dl, a.inv_nx_global);
      correct += (row_argmax(l, C) == y) ? 1.f : 0.f;
    } else {                                   // mse: target is [nx] broadcast against [nx][1] (C == 1 in the reference)
      for (int c = 0; c < C; ++c) {
        const float d = l[c] - a.target_f[i];
        lx += d * d;
        if (dl) dl[c] = 2.f * d * a.inv_nx_global / (float)C;
      }
    }
  }
  if (a.kind == 0 || a.kind == 1) {
    for (int i = threadIdx.x; i < a.nu; i += 256) {
      const float* l = a.logits + (long)(a.nx + i) * C;
      const float* t = a.logits_t + (long)i * C;
      float* dl = a.dlogits ? a.dlogits + (long)(a.nx + i) * C : nullptr;
      if (a.kind == 1) {
        lu += ce_row(l, C, row_argmax(t, C), dl, a.lambda_u * a.inv_nu_global);    // hard pseudo label, no threshold
      } else {
        for (int c = 0; c < C; ++c) {
          const float d = l[c] - t[c];          // F.mse_loss(logits_u_w, logits_u_s): teacher side carries no grad
          lu += d * d;
          if (dl) dl[c] = a.lambda_u * 2.f * d * a.inv_nu_global / (float)C;
        }
      }
    }
  }
  const float sx = block_sum(lx, sm);
  const float su = block_sum(lu, sm);
  const float sc = block_sum(correct, sm);
  if (threadIdx.x == 0) {
    const float fx = ce ? a.inv_nx_global : a.inv_nx_global / (float)C;
    const float fu = (a.kind == 1) ? a.inv_nu_global : a.inv_nu_global / (float)C;
    const float loss_x = sx * fx, loss_u = su * fu;
    a.out[0] = loss_x + a.lambda_u * loss_u;
    a.out[1] = loss_x;
    a.out[2] = loss_u;
    a.out[3] = sc;
  }
}

// out[i] = softmax(logits[i, :])[col] -- the 'tumor' probability of test_Camelyon16.py:58-60
__global__ __launch_bounds__(256) void softmax_col_kernel(const float* logits, float* out, int n, int C, int col) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* l = logits + (long)i * C;
  float m = l[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(l[c] - m);
  out[i] = expf(l[col] - m) / s;
}
hipError_t launch_softmax_col(const float* logits, float* out, int n, int C, int col, hipStream_t st) {
  hipLaunchKernelGGL(softmax_col_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, logits, out, n, C, col);
  return hipGetLastError();
}

hipError_t launch_loss(const LossArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace sslcr

// Common device/host helpers for the gfx950 (MI355X, CDNA4) engine.  HIP only -- no CUDA paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace sslcr {

typedef uint16_t bf16_t;                                    // raw bf16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

enum { DT_F32 = 0, DT_BF16 = 1 };

// ---- bf16 <-> f32 (round to nearest even, NaN-preserving enough for activations)
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- element traits: a 16-byte chunk holds EPC elements
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;
  static constexpr int DT = DT_F32;
  __device__ static __forceinline__ void unpack(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
  }
  __device__ static __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
    return v;
  }
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int DT = DT_BF16;
  __device__ static __forceinline__ void unpack(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v[i]); f[2 * i + 1] = bf_hi(v[i]); }
  }
  __device__ static __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
  }
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// fp32 -> storage type, 8 (bf16) / 4 (fp32) values per 16-byte chunk.  bf16 uses the hardware RNE pack (v_cvt_pk_bf16_f32):
// for VALU-exposed epilogues (conv_h16, stem): one instruction per pair instead of ~7.  (Used selectively: as the
// default pack it made the register-bound wgrad kernels slower.)
template <typename T> struct PackH {
  __device__ static __forceinline__ u32x4_t run(const float* f) { return Elem<T>::pack(f); }
};
template <> struct PackH<bf16_t> {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ u32x4_t run(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * i], f[2 * i + 1]}, bf16x2_t));
    return v;
  }
};

// ---- DPP all-reduce (sum) across the 16 lanes of a DPP row: quad xor1, quad xor2, half-mirror, mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}
// Sum over the 16 lanes of a DPP row, four values at a time, with the lane permutation folded into the add (v_add_f32_dpp): 16
// instructions per four values where row16_sum() (a v_mov_b32_dpp + an add per step) takes 32.  Interleaving the four keeps three
// instructions between a register's write and its next DPP read (the hardware wants two wait states); the leading s_nop covers the
// compiler's instruction in front of the block.
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// Sums of SIXTEEN values over the 16 lanes of a DPP row in 32 instructions (row16_sum: 16 x 4 fused, 128 as the compiler emits
// them): a reduction tree that halves the VALUES a lane carries while it doubles the lanes they cover.  Step 1 pairs lane i with
// i ^ 8 (row_ror:8): lanes 0-7 add up values 0-7, lanes 8-15 values 8-15 (the partner's copy of the same register) -- the two
// halves are the DPP bank masks 0x3 / 0xc, so no select is needed and the result lands in v[0..7] for every lane.  Step 2 pairs
// a lane with its mirror in the half row (row_half_mirror, which flips lane bit 2: banks 0x5 / 0xa) -> v[0..3]; steps 3-4 are the
// two quad exchanges on those four.  Afterwards lane l holds in v[0..3] the complete sums of values 4 (l >> 2) + 0..3: quad q of the
// row owns values 4q..4q+3 (all four lanes of the quad hold the same four).  Every DPP read is >= 3 instructions behind the write
// of its register (the hardware wants two wait states); the leading s_nop covers the compiler's instruction in front.
__device__ __forceinline__ void row16_fold16(float (&v)[16]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      // (early-clobber: %0..%7 are written before all of %8..%15 have been read -- without it the compiler may give an input the
      //  register of a read-write operand it can prove equal, e.g. two zeros)
      : "+&v"(v[0]), "+&v"(v[1]), "+&v"(v[2]), "+&v"(v[3]), "+&v"(v[4]), "+&v"(v[5]), "+&v"(v[6]), "+&v"(v[7])
      : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
}
// full 64-lane sum, result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// clamps as ONE v_med3_f32: fmaxf(x, 0) compiles to a canonicalising v_max x, x followed by the max, and in the output stages of
// the conv kernels every VALU instruction is ~4 exposed cycles (tools/microbench/pp64_phase_bench.hip)
__device__ __forceinline__ float clamp_lo(float x, float lo) { return __builtin_amdgcn_fmed3f(x, lo, __builtin_inff()); }
__device__ __forceinline__ float relu0(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }

__device__ __forceinline__ u32x4_t ld16(const void* p) { return *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4_t& v) { *reinterpret_cast<u32x4_t*>(p) = v; }
// streaming (non-temporal) forms for the read-once / write-once tensors of the HBM-bound elementwise kernels
__device__ __forceinline__ u32x4_t ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ void st16_nt(void* p, const u32x4_t& v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p)); }

// global -> LDS DMA in its MUBUF form (buffer_load_dwordx4 ... lds): lane l's 16 bytes at (base + voff + soff) land at dst + 16 l, no
// registers on the way.  Why not global_load_lds: that is a FLAT-family instruction, and after a FLAT instruction that touches LDS the
// compiler's wait-count pass is in its "pending flat" state, in which EVERY wait for an LDS read becomes s_waitcnt lgkmcnt(0) --
// including the fragment read issued one instruction earlier -- until the next vmcnt(0) + lgkmcnt(0) pair.  In the DMA-fed conv
// kernels that put a full LDS round trip in front of the first MFMA of two steps out of three (round 4, from the ISA); behind a
// buffer load the waits stay counted (lgkmcnt(4), (3), (3), (2) ... in front of the MFMA that needs the fragment).  A voff at or
// beyond `bytes` reads zeros (raw-buffer range check): the padding page of the gather kernels for free.
// (device pass only: the resource type does not exist in the host pass)
struct LdsDma {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#endif
  __device__ __forceinline__ void init(const void* base, unsigned bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
#endif
  }
  // voff: per-lane byte offset (VGPR), soff: wave-uniform byte offset (SGPR); dst: wave-uniform LDS address (goes to M0)
  __device__ __forceinline__ void load16(void* dst, int voff, int soff) const {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
#endif
  }
};

// Weight tiles sit in LDS in MFMA-fragment order: kout row r = q*(4TK) + t*4 + j of a 16*TK-row block (q = fragment lane>>2,
// t = MFMA tile, j = lane&3 -- the permutation that gives a lane 4*TK consecutive output channels) is stored at row
// t*16 + q*4 + j, so the 16 lanes of a fragment read 16 CONSECUTIVE LDS rows and the (row&7) XOR swizzle is conflict-free
// (kept in kout order the 16 rows alias 4-fold on the swizzle key).
template <int TK>
__device__ __forceinline__ int wperm(int r) {
  constexpr int B = 16 * TK;
  const int blk = r / B, x = r - blk * B;
  const int q = x / (4 * TK), y = x - q * (4 * TK);
  return blk * B + (y >> 2) * 16 + q * 4 + (y & 3);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- output stage shared by the convolution kernels.  A lane owns the 4*TK consecutive kouts starting at its kb of TP
// pixels: acc[TK][TP], byte offsets off[p] of (pixel, kb) in y (a valid address also where !ok[p]).
// The optional operands are COMPILE-TIME here and all their loads are issued before the first store.  Written as run-time
// `if (residual) load` inside the store loop, the compiler put an s_waitcnt vmcnt(0) in front of every store -- also when
// the loads were skipped -- and vmcnt retires in order, so each 16-byte store waited out the full write round trip of the
// one before it (conv_dma: a quarter of the kernel; ISA: store, load, vmcnt(0), store, ...).
template <typename T, int TK, int TP, bool RES, bool ACC, bool RELU, bool BIAS = true>
__device__ __forceinline__ void conv_store_tile_t(const f32x4_t (&acc)[TK][TP], const float (&bias)[4 * TK], const size_t (&off)[TP],
                                                  const bool (&ok)[TP], char* yg, const char* rg) {
  constexpr int EPC = Elem<T>::EPC, RQ = 4 * TK / EPC;
  u32x4_t r1[RES ? TP : 1][RES ? RQ : 1], r2[ACC ? TP : 1][ACC ? RQ : 1];
  if constexpr (RES) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
      for (int q = 0; q < RQ; ++q) r1[p][q] = ld16(rg + off[p] + q * 16);
  }
  if constexpr (ACC) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
      for (int q = 0; q < RQ; ++q) r2[p][q] = ld16(yg + off[p] + q * 16);
  }
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    float v[4 * TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[t * 4 + j] = BIAS ? acc[t][p][j] + bias[t * 4 + j] : acc[t][p][j];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      float* vq = v + q * EPC;
      if constexpr (RES) {
        float rr[EPC];
        Elem<T>::unpack(r1[p][q], rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
      }
      if constexpr (ACC) {
        float rr[EPC];
        Elem<T>::unpack(r2[p][q], rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] += rr[e];
      }
      if constexpr (RELU) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) vq[e] = relu0(vq[e]);
      }
      if (ok[p]) st16(yg + off[p] + q * 16, PackH<T>::run(vq));
    }
  }
}
// sslcr_conv_desc.out_scale (eval-mode BatchNorm scale kept out of the filters): acc <- acc * out_scale[kb + 4 t + j], in place, in front
// of the bias epilogue.  osc_kb = out_scale + kb (16-byte aligned: kb is a multiple of 4 TK); a uniform branch at the call sites, so the
// forms without it (train forward, every dgrad) pay nothing.
template <int TK, int TP>
__device__ __forceinline__ void conv_scale_acc(f32x4_t (&acc)[TK][TP], const float* osc_kb) {
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(osc_kb + 4 * t);
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][p][j] *= s4[j];
  }
}

template <typename T, int TK, int TP>
__device__ __forceinline__ void conv_store_tile(const f32x4_t (&acc)[TK][TP], const float (&bias)[4 * TK], const size_t (&off)[TP],
                                                const bool (&ok)[TP], char* yg, const char* rg, bool accumulate, bool relu) {
  // uniform three-way dispatch: eight straight-line bodies
  if (rg) {
    if (accumulate) { if (relu) conv_store_tile_t<T, TK, TP, true, true, true>(acc, bias, off, ok, yg, rg); else conv_store_tile_t<T, TK, TP, true, true, false>(acc, bias, off, ok, yg, rg); }
    else { if (relu) conv_store_tile_t<T, TK, TP, true, false, true>(acc, bias, off, ok, yg, rg); else conv_store_tile_t<T, TK, TP, true, false, false>(acc, bias, off, ok, yg, rg); }
  } else {
    if (accumulate) { if (relu) conv_store_tile_t<T, TK, TP, false, true, true>(acc, bias, off, ok, yg, rg); else conv_store_tile_t<T, TK, TP, false, true, false>(acc, bias, off, ok, yg, rg); }
    else { if (relu) conv_store_tile_t<T, TK, TP, false, false, true>(acc, bias, off, ok, yg, rg); else conv_store_tile_t<T, TK, TP, false, false, false>(acc, bias, off, ok, yg, rg); }
  }
}
// the same with the caller's knowledge that there is no bias (train-mode forward, every dgrad): the plain body is pack + store
template <typename T, int TK, int TP>
__device__ __forceinline__ void conv_store_tile_nobias(const f32x4_t (&acc)[TK][TP], const float (&bias)[4 * TK], const size_t (&off)[TP],
                                                       const bool (&ok)[TP], char* yg, const char* rg, bool accumulate, bool relu) {
  if (!rg && !accumulate && !relu) conv_store_tile_t<T, TK, TP, false, false, false, false>(acc, bias, off, ok, yg, rg);
  else if (rg && !accumulate && !relu) conv_store_tile_t<T, TK, TP, true, false, false, false>(acc, bias, off, ok, yg, rg);
  else conv_store_tile<T, TK, TP>(acc, bias, off, ok, yg, rg, accumulate, relu);
}

}  // namespace sslcr

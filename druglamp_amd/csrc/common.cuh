// common.cuh — shared device/host helpers for libdruglamp_hip (gfx950 only).
//
// MFMA conventions used by every matrix kernel in this library
// -----------------------------------------------------------
// One "fragment" is a 16-byte packet per lane (u32x4).  For a 16x16 MFMA tile
//     D[i][j] = sum_k A[i][k] * B[k][j]
// lane l = (il = l & 15, g = l >> 4) supplies A[il][slots of g] and B[slots of g][il] and
// receives D[4g + r][il] in accumulator register r (r = 0..3).
//   bf16: one fragment = 8 bf16 = the 8 contraction slots of lane-group g of ONE
//         v_mfma_f32_16x16x32_bf16 (KF = 32 contraction indices per fragment).
//   f32 : one fragment = 4 floats = one slot each of FOUR v_mfma_f32_16x16x4_f32
//         (KF = 16 contraction indices per fragment); exact fp32 (fmaf chain).
// The hardware multiplies slot s of group g of A with slot s of group g of B, so any slot ->
// contraction-index map is legal as long as both operands use the same one.  Two maps occur:
//   CONTIG: slot (g, s) <-> index g*(KF/4) + s        (operand read with K contiguous)
//   CTILE : slot (g, s) <-> index 16*(s/4) + 4g + s%4 (operand taken from accumulator tiles,
//           where lane (il, g) holds rows 4g..4g+3 of each 16-row tile)
// For f32 the two maps coincide.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/druglamp_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define DL_LDS __attribute__((address_space(3)))

// Study switches (tile forms that were measured and rejected, "skip the stores" timing decompositions ...) exist only
// in a -DDL_STUDY build (`python -m druglamp_amd.build --study` -> libdruglamp_hip_study.so, used by tools/).  The
// product library never reads the environment: dl_study_env() is the constant default there, so no stray variable
// can change which kernel runs or make a kernel skip work.
#ifdef DL_STUDY
#include <stdlib.h>
static inline int dl_study_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static constexpr int dl_study_env(const char*, int dflt) { return dflt; }
#endif
// kernel-side view of a study bit field: the constant 0 in the product build, so the study branches are compiled out
#ifdef DL_STUDY
#define DL_DBG(p) ((p).dbg)
#else
#define DL_DBG(p) 0
#endif

extern "C" void dl_set_error(const char* fmt, ...);
// profiling hooks (api.hip)
void dl_prof_before(int family, hipStream_t s);
void dl_prof_after(int family, hipStream_t s, double flops, double bytes, int tag = 0);

#define DL_CHECK_ARG(cond, code, ...)                 \
  do {                                                \
    if (!(cond)) {                                    \
      dl_set_error(__VA_ARGS__);                      \
      return (code);                                  \
    }                                                 \
  } while (0)

#define DL_CHECK_LAUNCH(what)                                               \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dl_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));   \
      return DL_ERR_LAUNCH;                                                 \
    }                                                                       \
  } while (0)

template <typename T> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr int value = DL_F32; };
template <> struct DTypeOf<bf16_t> { static constexpr int value = DL_BF16; };

static inline size_t dl_dtype_size(int dt) { return dt == DL_BF16 ? 2 : 4; }

// ---- scalar conversions ---------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) {
  return __builtin_bit_cast(float, w & 0xffff0000u);
}

// An fp32 quad rounded to the pipeline's storage type and back: the value a consumer of the STORED tensor would read.
// STUDY BUILDS ONLY (round 6, tools/trickle_bench.py): gemm_trickle_kernel parks its finished tile in LDS as bf16, so its GELU /
// gelu' / dropout / residual epilogues act on the rounded pre-activation; in a -DDL_STUDY build gemm_kernel and gemm_big_kernel
// round at the same point so that the three kernels can be compared bit for bit.  The product library keeps the fp32 value
// (one rounding per output): there dl_round_store is the identity.
template <typename T> __device__ __forceinline__ f32x4 dl_round_store(f32x4 v) {
#ifdef DL_STUDY
  if constexpr (sizeof(T) == 2) {
    const uint32_t a = pack_bf16x2(v[0], v[1]), b = pack_bf16x2(v[2], v[3]);
    return f32x4{bf16lo(a), bf16hi(a), bf16lo(b), bf16hi(b)};
  }
#endif
  return v;
}

// 4 consecutive elements of T <-> f32x4 (global memory, vector access)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
  u32x2 w = *reinterpret_cast<const u32x2*>(p);
  f32x4 r = {bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
  return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
  u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  *reinterpret_cast<u32x2*>(p) = w;
}

// ---- MFMA wrappers ----------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KF = 32;  // contraction indices per fragment
  static constexpr int EPC = 8;  // elements per 16-byte chunk
  __device__ static __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KF = 16;
  static constexpr int EPC = 4;
  __device__ static __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    // NOTE: written out element by element on f32x4 views — indexing the u32x4 inside an unrolled loop
    // and bit-casting each lane value made hipcc (ROCm 7.2) feed element 0 to all four MFMAs.
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
};

// sum of the contraction slots a lane holds in one fragment (bf16: four v_dot2c_f32_bf16 against 1.0; f32: adds)
template <typename T> __device__ __forceinline__ float frag_slot_sum(u32x4 f, float acc);
template <> __device__ __forceinline__ float frag_slot_sum<bf16_t>(u32x4 f, float acc) {
  // NOTE: pairs are taken by shufflevector from ONE bf16x8 view — indexing the u32x4 per element inside an
  // unrolled loop and bit-casting each word made hipcc (ROCm 7.2) use word 0 four times (cf. Mma<float>).
  const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3F803F80u);
  const bf16x8 v = __builtin_bit_cast(bf16x8, f);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 0, 1), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 2, 3), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 4, 5), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 6, 7), one, acc, false);
  return acc;
}
template <> __device__ __forceinline__ float frag_slot_sum<float>(u32x4 f, float acc) {
  const f32x4 v = __builtin_bit_cast(f32x4, f);
  return acc + ((v[0] + v[1]) + (v[2] + v[3]));
}

// Accumulator tiles -> fragment in the CTILE slot map.
//   bf16: two 16-row tiles (t0: indices 0..15, t1: 16..31) -> 8 bf16
//   f32 : one 16-row tile -> 4 floats
__device__ __forceinline__ u32x4 ctile_frag_bf16(f32x4 t0, f32x4 t1) {
  u32x4 r = {pack_bf16x2(t0[0], t0[1]), pack_bf16x2(t0[2], t0[3]), pack_bf16x2(t1[0], t1[1]),
             pack_bf16x2(t1[2], t1[3])};
  return r;
}
__device__ __forceinline__ u32x4 ctile_frag_f32(f32x4 t0) { return __builtin_bit_cast(u32x4, t0); }

// ---- LDS access -------------------------------------------------------------------------
__device__ __forceinline__ u32x4 lds_read16(const char* base, uint32_t byte_off) {
  return *reinterpret_cast<const u32x4*>(base + byte_off);
}
__device__ __forceinline__ void lds_write16(char* base, uint32_t byte_off, u32x4 v) {
  *reinterpret_cast<u32x4*>(base + byte_off) = v;
}
__device__ __forceinline__ float lds_read_f32(const char* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(base + byte_off);
}
// ds_read_b64_tr_b16: within each 16-lane group, lane t supplies the address of 4 contiguous
// bf16 belonging to row (t>>2), column chunk (t&3) of a [4][16] block; lane t receives column t
// of that block (rows 0..3).  Row placement is free (each lane has its own address).
__device__ __forceinline__ u32x2 lds_read_tr16(const char* base, uint32_t byte_off) {
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (DL_LDS bf16x4*)(base + byte_off));
  return __builtin_bit_cast(u32x2, v);
}

// ---- wave reductions (64 lanes) -----------------------------------------------------------
// Within each 16-lane row by DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS crossbar), across the
// four rows by v_readlane.  __shfl_xor compiles to ds_bpermute_b32 — six dependent LDS round trips per reduction —
// and was the critical path of the one-wave-per-row LayerNorm kernels.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
// reduce across the 4 lane groups (lanes il, il+16, il+32, il+48)
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float group4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Output stores of the GEMM epilogues carry the `nt` (streaming) policy: 65536x2048x512 plain 189 -> 153 us, 65536x1024x256
// 52 -> 49, 65536x512x2048 131 -> 119 (tools/epi_bench.py with DL_NT_MODE; sc1 / sc0 sc1 write-through forms are slower
// than plain).  The store phases of the persistent grid arrive as 33 MB bursts (one 128 KB tile per CU); with the default
// policy the lines are allocated in the XCD's 4 MB L2 and evict the operand panels the next main loops re-read.
// mode 1 (product): the compiler's non-temporal store; study modes: 2 = inline assembly with nt (within 1-3 % of the builtin;
// needs s_nop — the hazard recogniser cannot see into the asm and a VALU write of the data registers directly behind a
// store of more than 8 bytes corrupts single outputs), 3 = inline assembly with the default policy (as slow as the plain
// store: it is the policy, not the compiler's bookkeeping of the store)
__device__ __forceinline__ void store16_nt(void* dst, u32x4 v, int mode = 1) {
#ifdef DL_STUDY
  if (mode == 2) { asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 2" ::"v"(dst), "v"(v) : "memory"); return; }
  if (mode == 3) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(dst), "v"(v) : "memory"); return; }
#endif
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
}
__device__ __forceinline__ u32x4 load16_nt(const void* src) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src)); }
// The same policy per kernel family where a same-box A/B of the training step says it pays (DL_NT_MASK: 1 LayerNorm, 2 BatchNorm
// apply passes, 4 attention outputs and gradients, 8 split-K slabs of the weight-gradient tiles, 16 = non-temporal LOAD of the saved
// activation in the LayerNorm backward).
#ifndef DL_NT_MASK
#define DL_NT_MASK 0
#endif
__device__ __forceinline__ void store8_nt(void* dst, u32x2 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(dst)); }
template <int FAM> __device__ __forceinline__ void store16_fam(void* dst, u32x4 v) {
  if constexpr ((DL_NT_MASK & FAM) != 0) store16_nt(dst, v);
  else *reinterpret_cast<u32x4*>(dst) = v;
}
template <int FAM, typename T> __device__ __forceinline__ void store4_fam(T* p, f32x4 v) {
  if constexpr ((DL_NT_MASK & FAM) != 0 && sizeof(T) == 2) {
    store8_nt(p, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
  } else {
    store4<T>(p, v);
  }
}


// ---- math ---------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// packed fp32 helpers (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per instruction)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bc2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// GELU of the bf16 pipelines (round 6): gelu(x) = x Phi(x), Phi(x) - 1/2 = x R(x^2) with R a degree-8 minimax polynomial on |x| <=
// 4.3 (linear program over 3000 points, weighted for the error of the PRODUCT x Phi), x clamped to the interval: |error| <= 3.8e-5 on
// |x| <= 4.3 and <= 7.1e-5 on |x| <= 12 (beyond the interval it grows by 6e-6 per unit: x (1 - Phi(4.3))) — below half a bf16 ulp of
// every output above 0.02 in magnitude.  13 VALU slots per PAIR (2 v_med3, 11 packed) against 28 for the rational erf above
// (whose two v_rcp_f32 are quarter rate): the GELU + pre-activation epilogue of the fc1 products is VALU-bound (27 -> 19 slots per
// output element with the address arithmetic of gemm_big.cuh), 65536x2048x512: see DESIGN section 7.  (Rounds 1-5: an odd rational
// minimax erf, |err| <= 4.5e-7 — two orders below what a bf16 output can show.)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.3f, 4.3f), __builtin_amdgcn_fmed3f(x[1], -4.3f, 4.3f)};
  const f32x2 u = xc * xc;
  f32x2 r = fma2(bc2(4.198948828e-11f), u, bc2(-4.250320984e-09f));
  r = fma2(r, u, bc2(1.902729281e-07f));
  r = fma2(r, u, bc2(-5.007155778e-06f));
  r = fma2(r, u, bc2(8.712943963e-05f));
  r = fma2(r, u, bc2(-1.071470790e-03f));
  r = fma2(r, u, bc2(9.695499204e-03f));
  r = fma2(r, u, bc2(-6.615635008e-02f));
  r = fma2(r, u, bc2(3.988027275e-01f));
  return x * fma2(xc, r, bc2(0.5f));
}
// 4-wide GELU / GELU' keyed on the pipeline's storage type: fp32 pipelines keep libm erff (parity runs),
// bf16 pipelines take the packed polynomial form.
template <typename T> __device__ __forceinline__ f32x4 gelu4(f32x4 v) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    return v;
  } else {
    const f32x2 ga = gelu_fast2(f32x2{v[0], v[1]}), gb = gelu_fast2(f32x2{v[2], v[3]});
    return f32x4{ga[0], ga[1], gb[0], gb[1]};
  }
}
// gelu'(x) - 1/2 is odd; on |x| <= 5.5 it is x * P(x^2) / Q(x^2) with P cubic and Q cubic in x^2 to 1.0e-4 absolute
// (least-squares fit, checked in fp32 arithmetic over [-9, 9]) — a factor 40 below the bf16 rounding of the product it
// feeds.  One reciprocal and 8 packed FMAs per pair instead of erf + exp (the gelu' data-gradient epilogue is VALU-bound).
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
  x = f32x2{__builtin_amdgcn_fmed3f(x[0], -5.5f, 5.5f), __builtin_amdgcn_fmed3f(x[1], -5.5f, 5.5f)};   // (v_med3_f32: one slot per element; min + max: two)
  const f32x2 x2 = x * x;
  f32x2 p = fma2(bc2(1.6454146817e-04f), x2, bc2(1.4818409354e-02f));
  p = fma2(p, x2, bc2(-2.9252399590e-02f));
  p = fma2(p, x2, bc2(7.9844805043e-01f));
  f32x2 q = fma2(bc2(5.5159476634e-03f), x2, bc2(3.8876839848e-02f));
  q = fma2(q, x2, bc2(3.0011850475e-01f));
  q = fma2(q, x2, bc2(1.0f));
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  return fma2(x * p, r, bc2(0.5f));
}
template <typename T> __device__ __forceinline__ f32x4 gelu_grad4(f32x4 v) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_grad(v[e]);
    return v;
  } else {
    const f32x2 a = gelu_grad_fast2(f32x2{v[0], v[1]}), b = gelu_grad_fast2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
  }
}

// ---- dropout RNG: counter hash keyed by (seed, group index); one draw of 64 bits serves 4 consecutive
// elements (16 bits each).  keep[j] = bits16[j] >= thr16 with thr16 = round(p * 65536).  Two murmur3
// finalisers on decorrelated 32-bit keys (4 integer multiplies per 4 elements); the mask is regenerated
// from the same (seed, index) in backward, never stored.
__device__ __forceinline__ uint32_t dl_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint64_t dl_splitmix(uint64_t seed, uint64_t idx) {
  const uint32_t x = (uint32_t)idx + (uint32_t)seed;
  const uint32_t hi = (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32);
  const uint32_t a = dl_fmix32(x ^ hi);
  const uint32_t b = dl_fmix32((x + 0x9E3779B9u) ^ __builtin_rotateleft32(hi, 13) ^ 0x7F4A7C15u);
  return ((uint64_t)a << 32) | b;
}
__host__ __device__ __forceinline__ uint32_t dl_dropout_thr16(float p) {
  float t = p * 65536.0f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
// Effective dropout seed: the by-value seed of the call plus an optional DEVICE-resident offset.  A training step
// captured in a hipGraph bakes every by-value argument; the offset (one uint64 the host-side step loop bumps with a
// tiny in-graph kernel) is what gives each replay fresh masks, and forward / backward of one replay read the same value.
__device__ __forceinline__ uint64_t dl_eff_seed(uint64_t seed, const uint64_t* __restrict__ off) {
  return off ? seed + *off : seed;
}
// element (row, col) of a logical [rows][ncols] tensor, col % 4 == 0: returns 4 keep flags scaled
__device__ __forceinline__ f32x4 dl_dropout4(f32x4 v, uint64_t seed, uint64_t row, uint64_t col,
                                             uint64_t ncols, uint32_t thr16, float inv_keep) {
  const uint64_t bits = dl_splitmix(seed, (row * ncols + col) >> 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t b = (uint32_t)(bits >> (16 * j)) & 0xffffu;
    v[j] = (b >= thr16) ? v[j] * inv_keep : 0.0f;
  }
  return v;
}

// the same draw keyed by the GROUP index ((row * ncols + col) >> 2) directly: callers that walk rows in constant steps keep a running
// index instead of a 64-bit multiply per quad (gemm_big.cuh)
__device__ __forceinline__ f32x4 dl_dropout4_idx(f32x4 v, uint64_t seed, uint64_t group, uint32_t thr16, float inv_keep) {
  const uint64_t bits = dl_splitmix(seed, group);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t b = (uint32_t)(bits >> (16 * j)) & 0xffffu;
    v[j] = (b >= thr16) ? v[j] * inv_keep : 0.0f;
  }
  return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that are consecutive in the LOGICAL
// order land on the same XCD (observed placement: physical block b runs on XCD b % 8).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
  const uint32_t nx = 8;
  if (nblocks < nx) return bid;
  const uint32_t q = nblocks / nx, r = nblocks % nx;
  const uint32_t xcd = bid % nx, loc = bid / nx;
  const uint32_t start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + loc;
}

// Second stage of the two-stage column reductions: out[c] (+)= sum_k partial[k*stride + c], c < ncols.
// One 1024-thread workgroup per DL_REDUCE_COLS columns (16 columns x 64 row-lanes); deterministic (fixed
// summation order).
constexpr int DL_REDUCE_COLS = 16;   // columns per workgroup of dl_reduce_partials_kernel (launch with 1024 threads)
__global__ void dl_reduce_partials_kernel(const float* __restrict__ partial, int chunks, int64_t stride, int ncols,
                                          float* __restrict__ out, int accumulate);

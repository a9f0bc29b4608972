// tiles.cuh — LDS tile image + MFMA fragment loaders shared by the attention and NT-Xent kernels.
// Tiles are [rows][HD] with 16-byte chunks XOR-swizzled by ATile::swz(row); the same image serves
// ds_read_b128 (K-contiguous fragments) and ds_read_b64_tr_b16 / ds_read_b32 (transposed fragments).
#pragma once
#include <math.h>
#include "common.cuh"

namespace dltile {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int ATT_THREADS = 256;

template <typename T, int HD> struct ATile {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int RB = HD * ES;       // bytes per row
  static constexpr int CPR = RB / 16;      // 16-byte chunks per row
  static constexpr int EPC = 16 / ES;
  // XOR swizzle of a row's 16-byte chunks.  128-byte rows (bf16, head_dim 64): row & 7 — the row parity supplies
  // the fourth bank-space bit.  256-byte rows (bf16, head_dim 128) fill the 256-byte bank space on their own: the
  // 16 lanes of a ds_read_b128 group (rows il, chunk pair g) and the 32 lanes of a ds_read_b64_tr_b16 group (rows
  // 4g + q, adjacent chunks h) need FOUR swizzle bits, and row & 7 left both 2-way conflicted (SQ_LDS_BANK_CONFLICT
  // = 42-46 % of the LDS cycles of the head_dim-128 kernels): row bits 0..1 -> chunk bits 1..2, row bit 2 -> bit 3,
  // chunk bit 0 stays with g / h.
  __device__ static __forceinline__ int swz(int row) {
    if constexpr (ES == 2 && CPR == 16) return ((row & 3) << 1) | (((row >> 2) & 1) << 3);
    else return row & 7;
  }
  __device__ static __forceinline__ int off(int row, int col) {
    const int b = col * ES;
    return row * RB + ((((b >> 4) ^ swz(row))) << 4) + (b & 15);
  }
};

// ---- cooperative global -> LDS staging of NR rows (all 256 threads) -----------------------
// rows >= row_limit are zero filled.  NCH = chunks per thread.
template <typename T, int HD, int NR> struct Stager {
  using TL = ATile<T, HD>;
  static constexpr int NCH = (NR * TL::CPR + ATT_THREADS - 1) / ATT_THREADS;
  u32x4 r[NCH];
  __device__ __forceinline__ void load(const T* base, int64_t row_stride, int row0, int row_limit) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + i * ATT_THREADS;
      const int row = c / TL::CPR, ch = c % TL::CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < NR * TL::CPR && row0 + row < row_limit)
        v = *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + row) * row_stride + ch * TL::EPC);
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + i * ATT_THREADS;
      const int row = c / TL::CPR, ch = c % TL::CPR;
      if (c < NR * TL::CPR) lds_write16(lds, row * TL::RB + ((ch ^ TL::swz(row)) << 4), r[i]);
    }
  }
};

// K-contiguous fragment: rows rowbase+il, contraction chunk (kf, g)
template <typename T, int HD>
__device__ __forceinline__ u32x4 frag_kc(const char* lds, int rowbase, int kf, int il, int g) {
  using TL = ATile<T, HD>;
  const int row = rowbase + il;
  return lds_read16(lds, row * TL::RB + (((kf * 4 + g) ^ TL::swz(row)) << 4));
}
// K-contiguous fragment straight from global memory (kept in registers for a whole kernel)
template <typename T>
__device__ __forceinline__ u32x4 frag_global(const T* rowptr, bool valid, int kf, int g) {
  constexpr int KF = Mma<T>::KF;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (valid) v = *reinterpret_cast<const u32x4*>(rowptr + kf * KF + g * (KF / 4));
  return v;
}
// Transposed fragment: A[i = colbase + il][slots <-> tile rows rbase + CTILE map]
// (covers KF rows: 32 for bf16, 16 for f32)
template <typename T, int HD>
__device__ __forceinline__ u32x4 frag_tr(const char* lds, int rbase, int colbase, int il, int g) {
  using TL = ATile<T, HD>;
  if constexpr (sizeof(T) == 2) {
    const int r0 = rbase + 4 * g + (il >> 2);
    const int col = colbase + (il & 3) * 4;
    const u32x2 a = lds_read_tr16(lds, TL::off(r0, col));
    const u32x2 b = lds_read_tr16(lds, TL::off(r0 + 16, col));
    u32x4 r = {a[0], a[1], b[0], b[1]};
    return r;
  } else {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      r[j] = __builtin_bit_cast(uint32_t, lds_read_f32(lds, TL::off(rbase + 4 * g + j, colbase + il)));
    return r;
  }
}
// accumulator tiles -> fragment (CTILE map); `t` points at CT consecutive tiles
template <typename T>
__device__ __forceinline__ u32x4 frag_from_acc(const f32x4* t) {
  if constexpr (sizeof(T) == 2) return ctile_frag_bf16(t[0], t[1]);
  else return ctile_frag_f32(t[0]);
}


// ---- LDS-DMA as inline assembly (invisible to the compiler's wait-count model: counted vmcnt waits stay counted) ----
// one instruction = 64 lanes x 16 (4) bytes -> 1 KB (256 B) of LDS at lds_off + 16 (4) * lane
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace dltile

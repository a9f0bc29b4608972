// gemm.hip — dl_gemm / dl_colsum: MFMA GEMM with fused epilogues (see include/druglamp_hip.h).
//
// Tiling: 128(m) x 128(n) output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA 16x16 tiles), contraction consumed 128 BYTES per operand row per step (64 bf16 /
// 32 f32), two LDS buffers, global->register->LDS staging with the next tile's global loads in
// flight under the current tile's MFMAs.  Operand tiles in LDS:
//   K-contiguous operand: [128 rows][128 B], 16-byte chunks XOR-swizzled by (row & 7) so that the
//     ds_read_b128 fragment reads of 16 consecutive rows spread over 8 slots.
//   K-slow operand (contraction index is the slow memory dim: dgrad weights, both wgrad
//     operands): [k rows][128 elems] with a padded pitch, fragments by ds_read_b64_tr_b16
//     (bf16) or 4 x ds_read_b32 (f32) — the hardware transpose read.
// The MFMA is issued "swapped" (A = W fragment, B = X fragment) so each lane ends up holding 4
// CONSECUTIVE output columns of one output row: 8-byte (bf16) / 16-byte (f32) stores.
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include "common.cuh"

namespace {

constexpr int BKB = 128;  // bytes of contraction per operand row per step
// TW = MFMA 16x16 tiles per wave per dimension: the workgroup tile is (32*TW)^2 — 128x128 (TW=4, default)
// or 64x64 (TW=2, used for weight gradients with few output tiles so that split-K slabs stay small).
constexpr int NTHREADS = 256;

template <typename T, int TW> struct TileGeom {
  static constexpr int BT = 32 * TW;                                // tile rows / cols
  static constexpr int BKE = BKB / (int)sizeof(T);                 // contraction elems per step
  static constexpr int PITCH_KSLOW = BT * (int)sizeof(T);           // bytes per k-row (unpadded: XOR-swizzled chunks)
  static constexpr int CPR_KSLOW = PITCH_KSLOW / 16;                // 16-byte chunks per k-row
  static constexpr int TILE_KCONTIG = BT * BKB;                     // 16384 at TW=4
  static constexpr int TILE_KSLOW = BKE * PITCH_KSLOW;              // 18432 (bf16) / 17408 (f32)
  static constexpr int TILE_BYTES = TILE_KSLOW > TILE_KCONTIG ? TILE_KSLOW : TILE_KCONTIG;
};

struct GemmP {
  const char* X; const char* W; char* C;
  int64_t ldx, ldw, ldc;
  int M, N, K, k_per_split, mt, nt, splits;
  const float* bias;
  const char* res; int64_t ldr; int res_row_mod; int res_before_dropout;
  int act;
  char* pre_out; int64_t ldp;
  const char* dact_pre; int64_t lddp;
  uint32_t drop_thr16; float drop_inv_keep; uint64_t seed; const uint64_t* seed_off;
  int* tickets;   // gemm_big_kernel: dynamic tile hand-out (dl_gemm_args.tile_tickets) or nullptr
  int accumulate;
  float* slabs;   // split mode: [splits][M][N] f32
  float* cs_slabs; // split mode, optional: [splits][M] partial column sums of the K-slow X operand
  int dbg;
  int nt_c = 0, nt_pre = 0, nt_small = 0, nt_ext = 0;   // large-tile kernels: streaming (`nt`) policy for the output / pre-activation stores
  // paired launch (gemm_kernel only): byte offsets from problem 0's pointers to problem 1's, and the seed difference
  int nprob = 1;
  int64_t dX = 0, dW = 0, dC = 0, dBias = 0, dRes = 0, dPre = 0, dDact = 0;
  uint64_t dSeed = 0;
};

// ---- global -> registers (4 chunks of 16 B per thread per operand) ------------------------
template <typename T, bool KSLOW, int TW>
__device__ __forceinline__ void load_tile(const char* base, int64_t ld, int row0, int nrows, int k0,
                                          int kend, u32x4 (&r)[TW]) {
  constexpr int EPC = Mma<T>::EPC;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    const int c = tid + i * NTHREADS;
    u32x4 v = {0u, 0u, 0u, 0u};
    if constexpr (!KSLOW) {
      const int row = c >> 3, kc = c & 7;
      const int gr = row0 + row, gk = k0 + kc * EPC;
      if (gr < nrows && gk < kend)
        v = *reinterpret_cast<const u32x4*>(base + ((int64_t)gr * ld + gk) * (int64_t)sizeof(T));
    } else {
      constexpr int CPR = 32 * TW / EPC;  // chunks per k-row
      const int krow = c / CPR, cc = c % CPR;
      const int gk = k0 + krow, gr = row0 + cc * EPC;
      if (gk < kend && gr < nrows)
        v = *reinterpret_cast<const u32x4*>(base + ((int64_t)gk * ld + gr) * (int64_t)sizeof(T));
    }
    r[i] = v;
  }
}

// XOR swizzle of the 16-byte chunks of k-row k of a K-slow operand tile.  bf16 tiles are read by ds_read_b64_tr_b16:
// the 32 lanes of a lane group address k-rows {8g' + q : g' = g & 1, q = 0..3} and, per k-row, the 8-byte halves of two
// adjacent chunks (h = 0, 1).  The swizzle has to turn (q, g', h) into distinct 16-byte slots of the 256-byte bank
// space: with the former k & (CPR - 1), q and h both landed on chunk bit 0 (SQ_LDS_BANK_CONFLICT = half of all LDS
// cycles of the weight-gradient kernels).  16 chunks per row (pitch 256 B): q -> bits 1..2, g' -> bit 3, h stays on
// bit 0.  8 chunks per row (pitch 128 B: the k-row parity picks the bank-space half): q >> 1 -> bit 1, g' -> bit 2.
template <typename T, int TW>
__device__ __forceinline__ constexpr int kslow_swz(int k) {
  constexpr int CPR = TileGeom<T, TW>::CPR_KSLOW;
  if constexpr (sizeof(T) == 2 && CPR == 16) return ((k & 3) << 1) | (((k >> 3) & 1) << 3);
  else if constexpr (sizeof(T) == 2 && CPR == 8) return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);
  else return k & (CPR - 1);
}

template <typename T, bool KSLOW, int TW>
__device__ __forceinline__ void store_tile(char* lds, const u32x4 (&r)[TW]) {
  constexpr int EPC = Mma<T>::EPC;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    const int c = tid + i * NTHREADS;
    if constexpr (!KSLOW) {
      const int row = c >> 3, kc = c & 7;
      lds_write16(lds, row * BKB + ((kc ^ (row & 7)) << 4), r[i]);
    } else {
      constexpr int CPR = 32 * TW / EPC;
      const int krow = c / CPR, cc = c % CPR;
      lds_write16(lds, krow * TileGeom<T, TW>::PITCH_KSLOW + ((cc ^ kslow_swz<T, TW>(krow)) << 4), r[i]);
    }
  }
}

// K-contiguous operand tile straight into LDS (LDS-DMA, global_load_lds_dwordx4): no VGPR round trip and
// no ds_write.  The LDS destination of one wave-instruction is linear (wave-uniform base + lane*16), so
// the XOR swizzle is applied on the SOURCE side: LDS slot (row, physical chunk pc) receives the row's
// logical chunk pc ^ (row & 7).  Rows beyond the operand are clamped to its last row (their products
// only reach output rows/columns that are never stored); the K range must be whole steps.
template <typename T, bool KSLOW, int TW>
__device__ __forceinline__ void dma_tile(const char* base, int64_t ld, int row0, int nrows, int k0, char* lds) {
  constexpr int EPC = Mma<T>::EPC;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    const int c = tid + i * NTHREADS;
    const char* src;
    if constexpr (!KSLOW) {
      const int row = c >> 3, kc = (c & 7) ^ (row & 7);
      int gr = row0 + row;
      gr = gr < nrows ? gr : nrows - 1;
      src = base + ((int64_t)gr * ld + k0 + kc * EPC) * (int64_t)sizeof(T);
    } else {
      // K-slow operand: LDS slot (k-row, physical chunk pc) receives the k-row's logical chunk pc ^ (krow & (CPR-1))
      constexpr int CPR = TileGeom<T, TW>::CPR_KSLOW;
      const int krow = c / CPR, cc = (c % CPR) ^ kslow_swz<T, TW>(krow);
      int gr = row0 + cc * EPC;
      gr = gr < nrows ? gr : nrows - EPC;
      src = base + ((int64_t)(k0 + krow) * ld + gr) * (int64_t)sizeof(T);
    }
    const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((c - (tid & 63)) * 16));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
  }
}

// fragment for 16 tile rows starting at rb, fragment index kf within the step
template <typename T, bool KSLOW, int TW>
__device__ __forceinline__ u32x4 read_frag(const char* lds, int rb, int kf, int il, int g) {
  if constexpr (!KSLOW) {
    const int row = rb + il;
    return lds_read16(lds, row * BKB + (((kf * 4 + g) ^ (row & 7)) << 4));
  } else if constexpr (sizeof(T) == 2) {
    constexpr int P = TileGeom<T, TW>::PITCH_KSLOW, CM = TileGeom<T, TW>::CPR_KSLOW - 1;
    const int kidx = kf * 32 + g * 8 + (il >> 2);
    const int cb = (rb + (il & 3) * 4) * 2;                 // byte column; chunk = cb >> 4
    const u32x2 a = lds_read_tr16(lds, kidx * P + (((cb >> 4) ^ kslow_swz<T, TW>(kidx)) << 4) + (cb & 15));
    const u32x2 b = lds_read_tr16(lds, (kidx + 4) * P + (((cb >> 4) ^ kslow_swz<T, TW>(kidx + 4)) << 4) + (cb & 15));
    u32x4 r = {a[0], a[1], b[0], b[1]};
    return r;
  } else {
    constexpr int P = TileGeom<T, TW>::PITCH_KSLOW, CM = TileGeom<T, TW>::CPR_KSLOW - 1;
    const int kidx = kf * 16 + g * 4;
    const int cb = (rb + il) * 4;
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      r[j] = __builtin_bit_cast(uint32_t, lds_read_f32(lds, (kidx + j) * P + ((((cb >> 4) ^ ((kidx + j) & CM))) << 4) + (cb & 15)));
    return r;
  }
}

// EPI: 0 bias only | 1 general (all runtime flags) | 2 bias+pre_out+GELU(+dropout) | 3 bias(+dropout)+residual |
//      4 gelu'(dact_pre)(+dropout) | 5 bias+ReLU.   EPI != 1 need N % 8 == 0 and take 16-byte accesses only.
template <typename T, typename TO, bool XS, bool WS, bool SPLIT, bool DMA, int EPI, int TW, bool CS = false>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmP p0) {
  GemmP p = p0;            // (local copy: a paired launch re-points the epilogue fields per tile, see below)
  const uint64_t seed_off_v = (p0.drop_thr16 && p0.seed_off) ? *p0.seed_off : 0;   // the device-side step offset, read once per launch (see gemm_big_kernel)
  constexpr bool SIMPLE = (EPI == 0);
  constexpr int BM = 32 * TW, BN = 32 * TW;
  constexpr bool XD = DMA, WD = DMA;                 // operands staged by LDS-DMA (both layouts)
  constexpr int BKE = TileGeom<T, TW>::BKE;
  constexpr int NFRAG = BKE / Mma<T>::KF;
  constexpr int TB = TileGeom<T, TW>::TILE_BYTES;
  constexpr int EPB = 32 * TW * (32 * TW * 4 + 16);      // fp32 epilogue tile incl. row padding
  constexpr int SMEM = (2 * 2 * TB > EPB) ? 2 * 2 * TB : EPB;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int EP = BN * 4 + 16;                       // padded row pitch of the epilogue tile (bytes)
  static_assert(BM * EP <= SMEM, "epilogue tile must fit the LDS allocation");

  // Persistent tile loop: the grid holds at most two workgroups per CU; each walks the logical tile
  // list with stride gridDim.x.  Within one round the XCD-aware remap keeps neighbouring logical tiles
  // (same X row panel) on one XCD / L2.  The first operand tiles of the NEXT output tile are requested
  // before the epilogue of the current one, so their HBM latency and the epilogue's stores overlap.
  // Paired launch (dl_gemm_pair, nprob = 2): two problems of identical shape whose pointers differ by the byte deltas
  // d* — the two streams of a paired block.  Problem 1's tiles follow problem 0's in the tile list; operand pointers are
  // chosen when a tile is located, the epilogue's when it runs.
  const uint32_t per_prob = (uint32_t)p.mt * p.nt * p.splits;
  const uint32_t ntiles = per_prob * (uint32_t)p0.nprob;
  const uint32_t G = gridDim.x;
  const char* Xb = p0.X;
  const char* Wb = p0.W;
  int prob = 0;
  auto locate = [&](uint32_t it, int& split, int& m0, int& n0, int& kbeg, int& kend) {
    const uint32_t round0 = (it / G) * G;
    const uint32_t span = min(G, ntiles - round0);
    uint32_t t = round0 + xcd_remap(it - round0, span);
    prob = (t >= per_prob) ? 1 : 0;
    t -= prob ? per_prob : 0u;
    Xb = p0.X + (prob ? p0.dX : 0);
    Wb = p0.W + (prob ? p0.dW : 0);
    split = t / (p.mt * p.nt);
    const int tile = t % (p.mt * p.nt);
    m0 = (tile / p.nt) * BM; n0 = (tile % p.nt) * BN;
    kbeg = split * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  };
  u32x4 rx[TW], rw[TW];
  uint32_t it = blockIdx.x;
  if (it >= ntiles) return;
  if ((DL_DBG(p) & 4) && (blockIdx.x >= gridDim.x / 2)) { for (int z = 0; z < (DL_DBG(p) >> 4); ++z) __builtin_amdgcn_s_sleep(127); }
  int split, m0, n0, kbeg, kend;
  locate(it, split, m0, n0, kbeg, kend);
  if constexpr (!XD) load_tile<T, XS, TW>(Xb, p.ldx, m0, p.M, kbeg, kend, rx);
  if constexpr (!WD) load_tile<T, WS, TW>(Wb, p.ldw, n0, p.N, kbeg, kend, rw);
  for (;;) {
  f32x4 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // CS (weight-gradient form with bias gradient): the waves that own columns 0..63 of the first column tile also
  // sum their X fragments over k, one float per 16-row fragment.  The k loop exists in two copies selected by a
  // wave-uniform branch OUTSIDE it, so the other waves' schedule is untouched (a branch inside the loop cost
  // the plain kernel 60%, summing in every wave 20%).
  float cs[TW];
#pragma unroll
  for (int i = 0; i < TW; ++i) cs[i] = 0.f;
  const bool do_cs = CS && n0 == 0 && wn == 0;

  const int nk = (kend - kbeg + BKE - 1) / BKE;
  if constexpr (XD) dma_tile<T, XS, TW>(Xb, p.ldx, m0, p.M, kbeg, smem);
  else store_tile<T, XS, TW>(smem, rx);
  if constexpr (WD) dma_tile<T, WS, TW>(Wb, p.ldw, n0, p.N, kbeg, smem + TB);
  else store_tile<T, WS, TW>(smem + TB, rw);
  __syncthreads();
  auto kloop = [&](auto with_cs) {
    constexpr bool WCS = decltype(with_cs)::value;
    for (int kt = 0; kt < nk; ++kt) {
      char* cur = smem + (kt & 1) * 2 * TB;
      char* nxt = smem + ((kt + 1) & 1) * 2 * TB;
      const bool more = (kt + 1 < nk);
      if (more && !(DL_DBG(p) & 2)) {
        if constexpr (XD) dma_tile<T, XS, TW>(Xb, p.ldx, m0, p.M, kbeg + (kt + 1) * BKE, nxt);
        else load_tile<T, XS, TW>(Xb, p.ldx, m0, p.M, kbeg + (kt + 1) * BKE, kend, rx);
        if constexpr (WD) dma_tile<T, WS, TW>(Wb, p.ldw, n0, p.N, kbeg + (kt + 1) * BKE, nxt + TB);
        else load_tile<T, WS, TW>(Wb, p.ldw, n0, p.N, kbeg + (kt + 1) * BKE, kend, rw);
      }
#pragma unroll
      for (int kf = 0; kf < NFRAG; ++kf) {
        u32x4 fx[TW], fw[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i) fx[i] = read_frag<T, XS, TW>(cur, wm * 16 * TW + i * 16, kf, il, g);
#pragma unroll
        for (int j = 0; j < TW; ++j) fw[j] = read_frag<T, WS, TW>(cur + TB, wn * 16 * TW + j * 16, kf, il, g);
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[i][j] = Mma<T>::mma(fw[j], fx[i], acc[i][j]);
        if constexpr (WCS) {
#pragma unroll
          for (int i = 0; i < TW; ++i) cs[i] = frag_slot_sum<T>(fx[i], cs[i]);
        }
      }
      if (more) {
        if constexpr (!XD) store_tile<T, XS, TW>(nxt, rx);
        if constexpr (!WD) store_tile<T, WS, TW>(nxt + TB, rw);
      }
      __syncthreads();
    }
  };
  if constexpr (CS) {
    if (do_cs) kloop(std::true_type{}); else kloop(std::false_type{});
  } else {
    kloop(std::false_type{});
  }
  // request the next output tile's first operand tiles now (registers only)
  const int cm0 = m0, cn0 = n0, csplit = split;
  if (p0.nprob > 1) {                    // this tile's epilogue pointers (wave-uniform)
    const int64_t f = prob;
    p.C = p0.C + f * p0.dC;
    p.bias = p0.bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p0.bias) + f * p0.dBias) : nullptr;
    p.res = p0.res ? p0.res + f * p0.dRes : nullptr;
    p.pre_out = p0.pre_out ? p0.pre_out + f * p0.dPre : nullptr;
    p.dact_pre = p0.dact_pre ? p0.dact_pre + f * p0.dDact : nullptr;
    p.seed = p0.seed + (uint64_t)f * p0.dSeed;
  }
  const uint32_t itn = it + G;
  const bool have_next = itn < ntiles;
  if (have_next) {
    locate(itn, split, m0, n0, kbeg, kend);
    if constexpr (!XD) load_tile<T, XS, TW>(Xb, p.ldx, m0, p.M, kbeg, kend, rx);
    if constexpr (!WD) load_tile<T, WS, TW>(Wb, p.ldw, n0, p.N, kbeg, kend, rw);
  }

  if constexpr (CS) {
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const float v = group4_sum(cs[i]);
        const int m = cm0 + wm * 16 * TW + i * 16 + il;
        if (g == 0 && m < p.M) p.cs_slabs[(int64_t)csplit * p.M + m] = v;
      }
    }
  }
  // ---- epilogue -------------------------------------------------------------------------------
  // The accumulators (lane: row il, 4 consecutive columns 4g..4g+3 per tile) are first parked in LDS
  // as an fp32 [128][128] tile (the operand buffers are free now), then every thread takes 8
  // CONSECUTIVE columns of one row: bias / residual / pre-activation traffic and the final store are
  // full 16-byte (bf16) or 2x16-byte (f32) accesses, 16 lanes per 256-byte row -> whole cache lines.
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
      *reinterpret_cast<f32x4*>(smem + (wm * 16 * TW + i * 16 + il) * EP + (wn * 16 * TW + j * 16 + 4 * g) * 4) = acc[i][j];
  __syncthreads();
  const bool vec_ok = (p.N & 7) == 0;
  // this lane's 8 columns are the same in every piece of the tile (NTHREADS is a multiple of the BN / 8 pieces of a row): its
  // bias values are read ONCE per tile here.  (Round 5, late: read per piece they sat behind the previous piece's stores — the
  // wait for the load is a wait for those stores as well, vmcnt counts both in order: a store round trip per piece.)
  static_assert(NTHREADS % (BN / 8) == 0, "a lane keeps its columns across the pieces of a tile");
  const int n_lane = cn0 + (tid % (BN / 8)) * 8;
  f32x4 hb0 = {0.f, 0.f, 0.f, 0.f}, hb1 = hb0;
  const bool hb_ok = p.bias != nullptr && vec_ok && n_lane + 8 <= p.N;
  if (hb_ok) { hb0 = *reinterpret_cast<const f32x4*>(p.bias + n_lane); hb1 = *reinterpret_cast<const f32x4*>(p.bias + n_lane + 4); }
#pragma unroll 2
  for (int it = 0; it < TW * TW / 2; ++it) {
    const int c = tid + it * NTHREADS;
    const int row = c / (BN / 8), cc = (c % (BN / 8)) * 8;
    const int m = cm0 + row, n = cn0 + cc;
    if (m >= p.M || n >= p.N) continue;
    if ((DL_DBG(p) & 1)) continue;
    if constexpr (SIMPLE && !SPLIT) {
      // bias-only epilogue, N % 8 == 0: straight LDS -> (bias) -> convert -> one 16-byte store
      f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4 + 16);
      a0 += hb0; a1 += hb1;                        // (zeros without a bias; N % 8 == 0 here: every piece is full)
      TO* dst = reinterpret_cast<TO*>(p.C) + (int64_t)m * p.ldc + n;
      if constexpr (sizeof(TO) == 2) {
        u32x4 o = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
        if (p.nt_c && p.nt_small) store16_nt(dst, o, p.nt_c);
        else *reinterpret_cast<u32x4*>(dst) = o;
      } else {
        *reinterpret_cast<f32x4*>(dst) = a0;
        *reinterpret_cast<f32x4*>(dst + 4) = a1;
      }
      continue;
    }
    if constexpr (!SPLIT && EPI >= 2) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4 + 16);
      if (EPI != 4) { a0 += hb0; a1 += hb1; }
      if constexpr ((EPI == 2 || EPI == 3 || EPI == 4) && sizeof(T) == 2 && sizeof(TO) == 2) {       // (common.cuh: as the large-tile kernels)
        a0 = dl_round_store<T>(a0); a1 = dl_round_store<T>(a1);
      }
      if constexpr (EPI == 2) {
        T* pd = reinterpret_cast<T*>(p.pre_out) + (int64_t)m * p.ldp + n;
        store4<T>(pd, a0); store4<T>(pd + 4, a1);
        a0 = gelu4<T>(a0); a1 = gelu4<T>(a1);
      } else if constexpr (EPI == 5) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = fmaxf(a0[r], 0.f); a1[r] = fmaxf(a1[r], 0.f); }
      } else if constexpr (EPI == 4) {
        const T* src = reinterpret_cast<const T*>(p.dact_pre) + (int64_t)m * p.lddp + n;
        const f32x4 q0 = load4<T>(src), q1 = load4<T>(src + 4);
        a0 *= gelu_grad4<T>(q0); a1 *= gelu_grad4<T>(q1);
      }
      if (EPI != 5 && p.drop_thr16) {
        a0 = dl_dropout4(a0, p.seed + seed_off_v, (uint64_t)m, (uint64_t)n, (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
        a1 = dl_dropout4(a1, p.seed + seed_off_v, (uint64_t)m, (uint64_t)(n + 4), (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
      }
      if constexpr (EPI == 3) {
        const T* src = reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n;
        a0 += load4<T>(src); a1 += load4<T>(src + 4);
      }
      TO* dst = reinterpret_cast<TO*>(p.C) + (int64_t)m * p.ldc + n;
      if constexpr (sizeof(TO) == 2) {
        u32x4 o = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
        if (p.nt_c && p.nt_small) store16_nt(dst, o, p.nt_c);
        else *reinterpret_cast<u32x4*>(dst) = o;
      } else {
        *reinterpret_cast<f32x4*>(dst) = a0;
        *reinterpret_cast<f32x4*>(dst + 4) = a1;
      }
      continue;
    }
    float v[8];
    {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + row * EP + cc * 4 + 16);
      v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
    }
    const int nvalid = min(8, p.N - n);
    const bool full = vec_ok && nvalid == 8;
    if constexpr (SPLIT) {
      float* dst = p.slabs + ((int64_t)csplit * p.M + m) * p.N + n;
      if (full) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        for (int r = 0; r < nvalid; ++r) dst[r] = v[r];
      }
    } else {
      if (p.bias) {
        if (full) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += hb0[r]; v[4 + r] += hb1[r]; }
        } else {
          for (int r = 0; r < nvalid; ++r) v[r] += p.bias[n + r];
        }
      }
      if (p.pre_out) {
        T* dst = reinterpret_cast<T*>(p.pre_out) + (int64_t)m * p.ldp + n;
        if (full) { store4<T>(dst, f32x4{v[0], v[1], v[2], v[3]}); store4<T>(dst + 4, f32x4{v[4], v[5], v[6], v[7]}); }
        else for (int r = 0; r < nvalid; ++r) dst[r] = from_f32<T>(v[r]);
      }
      if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.0f);
      }
      if (p.dact_pre) {
        const T* src = reinterpret_cast<const T*>(p.dact_pre) + (int64_t)m * p.lddp + n;
        float pre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (full) {
          const f32x4 q0 = load4<T>(src), q1 = load4<T>(src + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { pre[r] = q0[r]; pre[4 + r] = q1[r]; }
        } else {
          for (int r = 0; r < nvalid; ++r) pre[r] = to_f32(src[r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= gelu_erf_grad(pre[r]);
      }
      float resv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.res) {
        const int rr = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
        const T* src = reinterpret_cast<const T*>(p.res) + (int64_t)rr * p.ldr + n;
        if (full) {
          const f32x4 q0 = load4<T>(src), q1 = load4<T>(src + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { resv[r] = q0[r]; resv[4 + r] = q1[r]; }
        } else {
          for (int r = 0; r < nvalid; ++r) resv[r] = to_f32(src[r]);
        }
        if (p.res_before_dropout) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += resv[r];
        }
      }
      if (p.drop_thr16) {
        f32x4 d0 = {v[0], v[1], v[2], v[3]}, d1 = {v[4], v[5], v[6], v[7]};
        d0 = dl_dropout4(d0, p.seed + seed_off_v, (uint64_t)m, (uint64_t)n, (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
        d1 = dl_dropout4(d1, p.seed + seed_off_v, (uint64_t)m, (uint64_t)(n + 4), (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = d0[r]; v[4 + r] = d1[r]; }
      }
      if (p.res && !p.res_before_dropout) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += resv[r];
      }
      TO* dst = reinterpret_cast<TO*>(p.C) + (int64_t)m * p.ldc + n;
      if (p.accumulate) {
        if (full) {
          const f32x4 o0 = load4<TO>(dst), o1 = load4<TO>(dst + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += o0[r]; v[4 + r] += o1[r]; }
        } else {
          for (int r = 0; r < nvalid; ++r) v[r] += to_f32(dst[r]);
        }
      }
      if (full) {
        if constexpr (sizeof(TO) == 2) {
          u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          *reinterpret_cast<u32x4*>(dst) = o;
        } else {
          *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
      } else {
        for (int r = 0; r < nvalid; ++r) dst[r] = from_f32<TO>(v[r]);
      }
    }
  }
  if (!have_next) break;
  it = itn;
  __syncthreads();          // the epilogue tile aliases the operand buffers
  }
}

#include "gemm_big.cuh"
#ifdef DL_STUDY          // (round 6 study kernel: measured and NOT adopted — tools/trickle_bench.py, profiles/r6_trickle_study.txt)
#include "gemm_trickle.cuh"
#endif

// out[idx] (+)= sum_z slabs[z][idx]
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, TO* __restrict__ out,
                                     int64_t mn, int64_t ldc, int N, int splits, int accumulate,
                                     const float* __restrict__ cs_slabs, float* __restrict__ cs_out, int M, uint32_t main_blocks) {
  if (blockIdx.x >= main_blocks) {
    // trailing workgroups: x_colsum[m] = sum_z cs_slabs[z][m]
    const int m = (int)(blockIdx.x - main_blocks) * blockDim.x + threadIdx.x;
    if (m < M) {
      float a0 = 0.f, a1 = 0.f;
      int z = 0;
      for (; z + 1 < splits; z += 2) { a0 += cs_slabs[(int64_t)z * M + m]; a1 += cs_slabs[(int64_t)(z + 1) * M + m]; }
      if (z < splits) a0 += cs_slabs[(int64_t)z * M + m];
      cs_out[m] = a0 + a1;
    }
    return;
  }
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= mn) return;
  // four independent accumulators keep 4 slab loads in flight per thread (fixed order -> deterministic)
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int z = 0;
  for (; z + 3 < splits; z += 4) {
    s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * mn + i4);
    s1 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 1) * mn + i4);
    s2 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 2) * mn + i4);
    s3 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 3) * mn + i4);
  }
  for (; z < splits; ++z) s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * mn + i4);
  f32x4 s = (s0 + s1) + (s2 + s3);
  const int64_t m = i4 / N, n = i4 % N;
  TO* dst = out + m * ldc + n;
  if (accumulate) { const f32x4 o = load4<TO>(dst); s += o; }
  store4<TO>(dst, s);
}

// tile width (MFMA tiles per wave per dim): 64x64 workgroup tiles for weight gradients with few output tiles
int pick_tw(const dl_gemm_args* a) {
  if (a->x_kslow && a->w_kslow && a->split_k >= 0) {
    const int64_t tiles128 = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    // (whole k-steps only: the register-staged 64x64 form is twice slower than the 128-tile one, 202 vs 111 us on the
    //  128x384x591870 conv gradient)
    if (tiles128 <= 4 && a->M >= 64 && a->N >= 64 && a->K % (BKB / (int)dl_dtype_size(a->in_dtype)) == 0) return 2;
  }
  return 4;
}

int auto_split(int64_t M, int64_t N, int64_t K, int bke, int bt) {
  const int64_t tiles = ((M + bt - 1) / bt) * ((N + bt - 1) / bt);
  if (tiles >= 192) return 1;
  const int round_wgs = dl_study_env("DL_SPLIT_ROUND", 512), max_sp = dl_study_env("DL_SPLIT_MAX", 256);
  int64_t want = round_wgs / tiles;        // floor: tiles * splits must fit ONE round of 512 resident workgroups (a
                                           // 516-workgroup plan ran 4 of them alone in a second round: 768x256, +25 %)
  if (want < 1) want = 1;
  int64_t ksteps = (K + bke - 1) / bke;
  int64_t maxs = ksteps / 4;  // at least 4 k-steps per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > max_sp) want = max_sp;
  return (int)(want < 1 ? 1 : want);
}

int big_tt_plan(const dl_gemm_args* a, int* bm_out);
// no k-range may be empty: the DMA path requests a slab's first operand tile before it looks at the range, and an
// empty range would start past the last row of the operands
static int trim_splits(int64_t K, int step, int sp) {
  const int64_t ksteps = (K + step - 1) / step;
  if (sp > ksteps) sp = (int)ksteps;
  if (sp < 1) sp = 1;
  const int64_t per = (ksteps + sp - 1) / sp;
  return (int)((ksteps + per - 1) / per);
}
int resolve_split(const dl_gemm_args* a) {
  { const int sp = big_tt_plan(a, nullptr); if (sp > 0) return trim_splits(a->K, 64, sp); }
  const int bke = BKB / (int)dl_dtype_size(a->in_dtype);
  if (a->split_k > 0) return trim_splits(a->K, bke, a->split_k);
  if (a->split_k < 0) return 1;
  // auto: only legal for plain f32 outputs
  const bool plain = !a->bias && !a->residual && !a->act && !a->pre_out && !a->dact_pre &&
                     a->dropout_p <= 0.f && a->out_dtype == DL_F32 && (a->N % 4 == 0);
  return plain ? trim_splits(a->K, bke, auto_split(a->M, a->N, a->K, bke, 32 * pick_tw(a))) : 1;
}

// specialised epilogue id (see gemm_kernel), 1 = general
int pick_epi(const GemmP& p, bool split) {
  int epi = 1;
  if (!split && (p.N % 8 == 0) && !p.accumulate && p.res_row_mod == 0 && !(p.res && p.res_before_dropout)) {
    const bool drop = p.drop_thr16 != 0;
    if (!p.res && !p.act && !p.pre_out && !p.dact_pre && !drop) epi = 0;
    else if (p.act == 1 && p.pre_out && !p.res && !p.dact_pre) epi = 2;
    else if (p.res && !p.act && !p.pre_out && !p.dact_pre) epi = 3;
    else if (p.dact_pre && !p.res && !p.act && !p.pre_out && !p.bias) epi = 4;
    else if (p.act == 2 && !p.res && !p.pre_out && !p.dact_pre && !drop) epi = 5;
  }
  return epi;
}

// Large-tile path (gemm_big.cuh): bf16 in/out, both operands K-contiguous, specialised epilogue, whole k-steps,
// N wide enough for a 256-column tile and enough tiles to give every CU one.  Measured against the
// alternatives the template allows (64-byte rows x 4 stages; 256x128 tiles, two workgroups per CU) the
// 128-byte x 2-stage 256x256 form won on every shape of the path; narrow outputs (N <= 128) stay on
// gemm_kernel.  dl_gemm_args.algo = DL_GEMM_ALGO_TILE128 keeps a call off the path (bitwise A/B tests).
bool big_eligible(const dl_gemm_args* a, const GemmP& p, int sp) {
  if (a->algo == DL_GEMM_ALGO_TILE128) return false;
  if (sp > 1 || a->in_dtype != DL_BF16 || a->out_dtype != DL_BF16 || a->x_kslow || a->w_kslow) return false;
  if (pick_epi(p, false) == 1) return false;
  if (a->K % 64 != 0 || a->N <= 128) return false;
  const int64_t tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256);
  return tiles >= 192;
}
int big_cfg() { return dl_study_env("DL_GEMM_BIGCFG", 0); }
#ifdef DL_STUDY
// Trickle form (gemm_trickle.cuh; STUDY LIBRARY ONLY, opt-in through DL_GEMM_TRICKLE=1): 256x128 tiles whose epilogue drains from
// LDS inside the next tile's main loop.  Returns the pieces per k-step (0 = not eligible).  Needs whole tiles (M % 256, N % 128),
// K = 256 or a multiple of 512 (the piece schedule), static tile order, one of the epilogues {plain / bias, bias + GELU +
// pre-activation, bias + residual, gelu'}.
int trickle_ppk(const dl_gemm_args* a, const GemmP& p, int sp) {
  if (dl_study_env("DL_GEMM_TRICKLE", 0) == 0) return 0;
  if (!big_eligible(a, p, sp) || p.tickets) return 0;
  const int epi = pick_epi(p, false);
  if (!(epi == 0 || epi == 2 || epi == 3 || epi == 4)) return 0;
  if (a->N % 128 != 0 || a->N / 128 > 256 || a->M % 256 != 0) return 0;
  const int64_t kmax = dl_study_env("DL_GEMM_TRICKLE_MAXK", 2048);
  if (a->K > kmax) return 0;
  if (a->K == 256) return 2;
  return a->K % 512 == 0 ? 1 : 0;
}
void launch_trickle(const GemmP& p, hipStream_t s, int ppk) {
  const uint32_t ntiles = (uint32_t)p.mt * p.nt;
  uint32_t nblocks = (256u / (uint32_t)p.nt) * (uint32_t)p.nt;            // a workgroup keeps its column tile: bias / columns fixed per launch
  if (nblocks > ntiles) nblocks = ntiles;
#define DL_TRK(E)                                                                                            \
  do {                                                                                                       \
    if (ppk == 2) hipLaunchKernelGGL((gemm_trickle_kernel<E, 2>), dim3(nblocks), dim3(512), 0, s, p);      \
    else hipLaunchKernelGGL((gemm_trickle_kernel<E, 1>), dim3(nblocks), dim3(512), 0, s, p);               \
  } while (0)
  switch (pick_epi(p, false)) {
    case 0: DL_TRK(0); break;
    case 2: DL_TRK(2); break;
    case 3: DL_TRK(3); break;
    default: DL_TRK(4); break;
  }
#undef DL_TRK
}
#else
static inline int trickle_ppk(const dl_gemm_args*, const GemmP&, int) { return 0; }
static inline void launch_trickle(const GemmP&, hipStream_t, int) {}
#endif
// Few-tile ("latency") form of the same kernel: a 128x128 tile per 4-wave workgroup with a DEEP stage ring.  When the
// whole output is at most a round or two of tiles (strong-scaling batches: M = 8192 ... 32768 rows) every workgroup walks
// its k-steps alone on its CU, and gemm_kernel's two-buffer ring pays one full memory round trip per k-step; three
// steps in flight divide that by three.  Returns the variant (0 = not eligible).
int lat_cfg() { return dl_study_env("DL_GEMM_LATCFG", 1); }
int lat_eligible(const dl_gemm_args* a, const GemmP& p, int sp) {
  const int cfg = lat_cfg();
  if (cfg == 0 || a->algo == DL_GEMM_ALGO_TILE128) return 0;
  if (sp > 1 || a->in_dtype != DL_BF16 || a->out_dtype != DL_BF16 || a->x_kslow || a->w_kslow) return 0;
  if (pick_epi(p, false) == 1) return 0;
  // measured (tools/latency_gemm_bench.py, M = 8192 / 16384 / 32768): 10-25 % faster than the two-buffer ring when the
  // tiles fit ONE round of one-workgroup-per-CU and there are >= 8 k-steps; with two rounds, or at K = 256, the
  // two-workgroups-per-CU ring of gemm_kernel wins
  if (a->K % 64 != 0 || a->K < dl_study_env("DL_GEMM_LATMINK", 512) || a->N < 128) return 0;
  const int64_t tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
  if (tiles > dl_study_env("DL_GEMM_LATMAX", 256)) return 0;
  return cfg;
}
void launch_lat(const GemmP& p, hipStream_t s, int cfg) {
  const uint32_t ntiles = (uint32_t)p.mt * p.nt;
#define DL_LAT_SW(XF_, RB_, NS_, WGS_)                                                                              \
  {                                                                                                                  \
    const uint32_t nblocks = ntiles < (WGS_) ? ntiles : (WGS_);                                                      \
    switch (pick_epi(p, false)) {                                                                                    \
      case 0: hipLaunchKernelGGL((gemm_big_kernel<XF_, 2, 2, RB_, NS_, 0>), dim3(nblocks), dim3(256), 0, s, p); break; \
      case 2: hipLaunchKernelGGL((gemm_big_kernel<XF_, 2, 2, RB_, NS_, 2>), dim3(nblocks), dim3(256), 0, s, p); break; \
      case 3: hipLaunchKernelGGL((gemm_big_kernel<XF_, 2, 2, RB_, NS_, 3>), dim3(nblocks), dim3(256), 0, s, p); break; \
      case 4: hipLaunchKernelGGL((gemm_big_kernel<XF_, 2, 2, RB_, NS_, 4>), dim3(nblocks), dim3(256), 0, s, p); break; \
      default: hipLaunchKernelGGL((gemm_big_kernel<XF_, 2, 2, RB_, NS_, 5>), dim3(nblocks), dim3(256), 0, s, p); break; \
    }                                                                                                                \
  }
#ifdef DL_STUDY
  if (cfg == 2) { DL_LAT_SW(4, 128, 3, 256u); return; }       // 96 KB ring, one workgroup per CU
  if (cfg == 3) { DL_LAT_SW(4, 64, 4, 512u); return; }        // 64-byte rows: 64 KB ring, two workgroups per CU
#endif
  DL_LAT_SW(4, 128, 4, 256u);                                 // 128 KB ring, one workgroup per CU, three steps in flight
#undef DL_LAT_SW
}
void launch_big(const GemmP& p, hipStream_t s) {
  const uint32_t ntiles = (uint32_t)p.mt * p.nt;
#ifdef DL_STUDY        // (the three rejected tile forms are instantiated in the study library only: 15 kernels the product never launches)
  const int cfg = big_cfg();
  // tile studies (DL_GEMM_BIGCFG): 1 = 256x128 tiles, 64-byte rows, 3 stages, two 4-wave workgroups per CU;
  // 2 = 256x256 tiles, 64-byte rows, 4 stages
  if (cfg == 1) {
    const uint32_t nblocks = ntiles < 512u ? ntiles : 512u;
#define DL_BIG(E) hipLaunchKernelGGL((gemm_big_kernel<8, 2, 2, 64, 3, E>), dim3(nblocks), dim3(256), 0, s, p)
    switch (pick_epi(p, false)) {
      case 0: DL_BIG(0); break;
      case 2: DL_BIG(2); break;
      case 3: DL_BIG(3); break;
      case 4: DL_BIG(4); break;
      default: DL_BIG(5); break;
    }
#undef DL_BIG
    return;
  }
#endif
  const uint32_t nblocks = ntiles < 256u ? ntiles : 256u;       // one 128 KB workgroup per CU
#ifdef DL_STUDY
  if (cfg == 3) {               // 16 waves of 64x64 (four per SIMD) on the same 256x256 tile and stage ring
#define DL_BIG(E) hipLaunchKernelGGL((gemm_big_kernel<4, 4, 4, 128, 2, E>), dim3(nblocks), dim3(1024), 0, s, p)
    switch (pick_epi(p, false)) {
      case 0: DL_BIG(0); break;
      case 2: DL_BIG(2); break;
      case 3: DL_BIG(3); break;
      case 4: DL_BIG(4); break;
      default: DL_BIG(5); break;
    }
#undef DL_BIG
    return;
  }
  if (cfg == 2) {
#define DL_BIG(E) hipLaunchKernelGGL((gemm_big_kernel<8, 2, 4, 64, 4, E>), dim3(nblocks), dim3(512), 0, s, p)
    switch (pick_epi(p, false)) {
      case 0: DL_BIG(0); break;
      case 2: DL_BIG(2); break;
      case 3: DL_BIG(3); break;
      case 4: DL_BIG(4); break;
      default: DL_BIG(5); break;
    }
#undef DL_BIG
    return;
  }
#endif
#define DL_BIG(E) hipLaunchKernelGGL((gemm_big_kernel<8, 2, 4, 128, 2, E>), dim3(nblocks), dim3(512), 0, s, p)
  switch (pick_epi(p, false)) {
    case 0: DL_BIG(0); break;
    case 2: DL_BIG(2); break;
    case 3: DL_BIG(3); break;
    case 4: DL_BIG(4); break;
    default: DL_BIG(5); break;
  }
#undef DL_BIG
}

// Large-tile weight-gradient path (gemm_big_tt2_kernel): bf16 operands, both K-slow, plain fp32 output through
// split-K slabs.  Returns the slab count (0 = not eligible); *bm_out is the tile height (256 or 128; 256 columns).
int big_tt_plan(const dl_gemm_args* a, int* bm_out) {
  if (a->algo == DL_GEMM_ALGO_TILE128) return 0;
  if (a->in_dtype != DL_BF16 || !a->x_kslow || !a->w_kslow || a->split_k != 0) return 0;
  const bool plain = !a->bias && !a->residual && !a->act && !a->pre_out && !a->dact_pre && a->dropout_p <= 0.f;
  if (!plain || a->M % 8 != 0 || a->N % 8 != 0 || a->N < 192 || a->M < 96 || a->K < 4096) return 0;
  // One round of 256 workgroups writes 256 fp32 tiles of slabs whatever the problem, so a tile only pays when the
  // operand stream dwarfs that.  256x256 tiles from 640K outputs (2048x512, 1536x512); below that 128x256 tiles
  // (same slab bytes as the 128-tile kernel's 512-workgroup plan) where tools/tt_study.py measured a win:
  // 1024x256, 256x1024, 768x256, 512x512, 256x648 at K = 65536, 256x392 at K = 131072 (5-17 %), the
  // 128 x {768, 1152} x 591864 conv gradients (20-25 %); NOT 256x512 / 256x256 at K = 65536 or 128x384x591870.
  const bool big = a->M > 128 && a->M * a->N >= 640 * 1024;
  const int bm = big ? 256 : 128;
  if (!big) {
    const double work = (double)a->M * (double)a->N * (double)a->K;
    if (a->M > 128 ? work < 1.0e10 : (a->N < 640 || a->K < 262144)) return 0;
  }
  const int64_t tiles = ((a->M + bm - 1) / bm) * ((a->N + 255) / 256);
  if (tiles > 256) return 0;
  int64_t sp = 256 / tiles;                          // one round of at most 256 workgroups
  const int64_t ksteps = (a->K + 63) / 64;
  if (sp > ksteps / 4) sp = ksteps / 4;
  if (sp < 1) sp = 1;
  if (bm_out) *bm_out = bm;
  return (int)sp;
}
template <int XF, int KS, int NS, bool PF>
void launch_big_tt2(const GemmP& p, hipStream_t s, uint32_t nblocks) {
  if (p.cs_slabs) hipLaunchKernelGGL((gemm_big_tt2_kernel<XF, 2, 4, true, KS, NS, PF>), dim3(nblocks), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((gemm_big_tt2_kernel<XF, 2, 4, false, KS, NS, PF>), dim3(nblocks), dim3(512), 0, s, p);
}
void launch_big_tt(const GemmP& p, hipStream_t s, int bm) {
  const uint32_t ntiles = (uint32_t)p.mt * p.nt * p.splits;
  const uint32_t nblocks = ntiles < 256u ? ntiles : 256u;
#ifdef DL_STUDY
  // rejected forms (tools/tt_study.py): two 64-row stages (one step in flight), the L2 prefetch, five 32-row stages
  switch (dl_study_env("DL_GEMM_TTCFG", 0)) {
    case 1: if (bm == 256) launch_big_tt2<8, 64, 2, false>(p, s, nblocks); else launch_big_tt2<4, 64, 2, false>(p, s, nblocks); return;
    case 2: if (bm == 256) launch_big_tt2<8, 64, 2, true>(p, s, nblocks); else launch_big_tt2<4, 64, 2, true>(p, s, nblocks); return;
    case 4: if (bm == 256) launch_big_tt2<8, 32, 4, true>(p, s, nblocks); else launch_big_tt2<4, 64, 3, true>(p, s, nblocks); return;
    default: break;
  }
#endif
  // 256x256: four 32-row stages (three steps in flight); 128x256: three 64-row stages
  if (bm == 256) launch_big_tt2<8, 32, 4, false>(p, s, nblocks); else launch_big_tt2<4, 64, 3, false>(p, s, nblocks);
}

template <typename T, typename TO, bool XS, bool WS, bool SPLIT, bool DMA>
void launch(const GemmP& p, hipStream_t s, int tw) {
  const int epi = pick_epi(p, SPLIT);
  const uint32_t ntiles = (uint32_t)p.mt * p.nt * p.splits;
  const uint32_t nblocks = ntiles < 512u ? ntiles : 512u;      // 256 CUs x 2 resident workgroups (LDS-limited)
#define DL_LAUNCH(E) hipLaunchKernelGGL((gemm_kernel<T, TO, XS, WS, SPLIT, DMA, E, 4>), dim3(nblocks), dim3(NTHREADS), 0, s, p)
  if constexpr (SPLIT) {
    if constexpr (XS && WS) {
      if (p.cs_slabs) {
        if (tw == 2) hipLaunchKernelGGL((gemm_kernel<T, TO, XS, WS, SPLIT, DMA, 1, 2, true>), dim3(nblocks), dim3(NTHREADS), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<T, TO, XS, WS, SPLIT, DMA, 1, 4, true>), dim3(nblocks), dim3(NTHREADS), 0, s, p);
        return;
      }
      if (tw == 2) {
        hipLaunchKernelGGL((gemm_kernel<T, TO, XS, WS, SPLIT, DMA, 1, 2>), dim3(nblocks), dim3(NTHREADS), 0, s, p);
        return;
      }
    }
    DL_LAUNCH(1);
  } else if constexpr (!DMA) {
    // the register-staged layouts keep two variants only
    if (epi == 0) DL_LAUNCH(0); else DL_LAUNCH(1);
  } else {
    switch (epi) {
      case 0: DL_LAUNCH(0); break;
      case 2: DL_LAUNCH(2); break;
      case 3: DL_LAUNCH(3); break;
      case 4: DL_LAUNCH(4); break;
      case 5: DL_LAUNCH(5); break;
      default: DL_LAUNCH(1); break;
    }
  }
#undef DL_LAUNCH
}

template <typename T, typename TO, bool SPLIT>
int dispatch_layout(const dl_gemm_args* a, const GemmP& p, hipStream_t s, int tw) {
  // LDS-DMA staging needs whole K steps per split and at least one K-contiguous operand
  const int bke = BKB / (int)sizeof(T);
  const bool dma = (a->K % bke == 0) && (p.k_per_split % bke == 0) && !(DL_DBG(p) & 8);
  if (!a->x_kslow && !a->w_kslow) { if (dma) launch<T, TO, false, false, SPLIT, true>(p, s, tw); else launch<T, TO, false, false, SPLIT, false>(p, s, tw); }
  else if (!a->x_kslow && a->w_kslow) { if (dma) launch<T, TO, false, true, SPLIT, true>(p, s, tw); else launch<T, TO, false, true, SPLIT, false>(p, s, tw); }
  else if (a->x_kslow && a->w_kslow) { if (dma) launch<T, TO, true, true, SPLIT, true>(p, s, tw); else launch<T, TO, true, true, SPLIT, false>(p, s, tw); }
  else {
    dl_set_error("dl_gemm: layout x_kslow=1,w_kslow=0 is not instantiated");
    return DL_ERR_UNSUPPORTED;
  }
  return DL_OK;
}

}  // namespace

extern "C" size_t dl_gemm_workspace_bytes(const dl_gemm_args* a) {
  if (!a) return 0;
  const int sp = resolve_split(a);
  const size_t cs = a->x_colsum ? (size_t)sp * (size_t)a->M * sizeof(float) : 0;
  if (big_tt_plan(a, nullptr) > 0 || sp > 1 || a->x_colsum) return (size_t)sp * (size_t)a->M * (size_t)a->N * sizeof(float) + cs;
  return 0;
}

namespace {
// Algorithmic HBM bytes of one product: every DISTINCT operand byte once + the output + the epilogue's extra operands.
// An operand whose row pitch is smaller than its row length (the implicit-im2col A operand of the ProteinCNN
// convolutions, reference model/basic_model.py:155-180: pitch C, K = k * C; the K-slow W operand of their weight
// gradients) overlaps its own rows: it spans (rows - 1) * pitch + row_length elements, not rows * row_length.
static double gemm_algorithmic_bytes(const dl_gemm_args* a) {
  const double es = (double)dl_dtype_size(a->in_dtype), oes = (double)dl_dtype_size(a->out_dtype);
  auto span = [](double rows, double len, double pitch) { return pitch < len ? (rows - 1.0) * pitch + len : rows * len; };
  const double x = a->x_kslow ? span((double)a->K, (double)a->M, (double)a->ldx) : span((double)a->M, (double)a->K, (double)a->ldx);
  const double w = a->w_kslow ? span((double)a->K, (double)a->N, (double)a->ldw) : span((double)a->N, (double)a->K, (double)a->ldw);
  const double mn = (double)a->M * (double)a->N;
  double bytes = (x + w) * es + mn * oes;
  if (a->pre_out) bytes += mn * es;                                   // pre-activation copy (written)
  if (a->dact_pre) bytes += mn * es;                                  // saved pre-activation (read)
  if (a->residual) bytes += (a->res_row_mod > 0 ? (double)a->res_row_mod * (double)a->N : mn) * es;
  if (a->accumulate) bytes += mn * oes;                               // C read back
  if (a->bias) bytes += (double)a->N * 4.0;
  return bytes;
}
constexpr int DL_PAIR_FALLBACK = 1;       // gemm_run: the pair is not on the gemm_kernel path, nothing was launched
// b: nullptr, or a second problem of identical shape / layout / epilogue (checked by dl_gemm_pair) that shares the launch.
int gemm_run(const dl_gemm_args* a, const dl_gemm_args* b, dl_stream stream);
}  // namespace
extern "C" int dl_gemm(const dl_gemm_args* a, dl_stream stream) { return gemm_run(a, nullptr, stream); }

extern "C" int dl_gemm_pair(const dl_gemm_args* a, const dl_gemm_args* b, dl_stream stream) {
  DL_CHECK_ARG(a && b, DL_ERR_ARG, "dl_gemm_pair: null argument block");
  auto same_presence = [](const void* x, const void* y) { return (x == nullptr) == (y == nullptr); };
  const bool twin =
      a->M == b->M && a->N == b->N && a->K == b->K && a->ldx == b->ldx && a->ldw == b->ldw && a->ldc == b->ldc &&
      a->x_kslow == b->x_kslow && a->w_kslow == b->w_kslow && a->in_dtype == b->in_dtype && a->out_dtype == b->out_dtype &&
      a->ldr == b->ldr && a->res_row_mod == b->res_row_mod && a->res_before_dropout == b->res_before_dropout &&
      a->act == b->act && a->ldp == b->ldp && a->lddp == b->lddp && a->dropout_p == b->dropout_p &&
      a->accumulate == b->accumulate && a->split_k == b->split_k && a->algo == b->algo &&
      a->dropout_seed_offset == b->dropout_seed_offset && a->tile_tickets == b->tile_tickets &&
      same_presence(a->bias, b->bias) && same_presence(a->residual, b->residual) && same_presence(a->pre_out, b->pre_out) &&
      same_presence(a->dact_pre, b->dact_pre) && !a->x_colsum && !b->x_colsum && !a->deferred && !b->deferred &&
      (((uintptr_t)b->X | (uintptr_t)b->W | (uintptr_t)b->C | (uintptr_t)b->bias | (uintptr_t)b->residual |
        (uintptr_t)b->pre_out | (uintptr_t)b->dact_pre) & 15) == 0;
  if (twin) {
    const int rc = gemm_run(a, b, stream);
    if (rc != DL_PAIR_FALLBACK) return rc;
  }
  const int rc = gemm_run(a, nullptr, stream);
  return rc != DL_OK ? rc : gemm_run(b, nullptr, stream);
}

namespace {
int gemm_run(const dl_gemm_args* a, const dl_gemm_args* b, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(a && a->X && a->W && a->C, DL_ERR_ARG, "dl_gemm: null operand");
  if (b) DL_CHECK_ARG(b->X && b->W && b->C, DL_ERR_ARG, "dl_gemm_pair: null operand");
  DL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, DL_ERR_SHAPE, "dl_gemm: bad shape M=%ld N=%ld K=%ld",
               (long)a->M, (long)a->N, (long)a->K);
  DL_CHECK_ARG(a->M < (1ll << 30) && a->N < (1ll << 30) && a->K < (1ll << 30), DL_ERR_SHAPE,
               "dl_gemm: dims must fit int32");
  DL_CHECK_ARG(a->in_dtype == DL_F32 || a->in_dtype == DL_BF16, DL_ERR_ARG, "dl_gemm: bad in_dtype");
  DL_CHECK_ARG(a->out_dtype == DL_F32 || a->out_dtype == a->in_dtype, DL_ERR_ARG,
               "dl_gemm: out_dtype must be f32 or in_dtype");
  const int es = (int)dl_dtype_size(a->in_dtype);
  const int epc = 16 / es;
  // 16-byte chunk granularity along the contiguous dim of each operand
  DL_CHECK_ARG(a->ldx % epc == 0 && a->ldw % epc == 0, DL_ERR_ALIGN,
               "dl_gemm: ldx/ldw must be multiples of %d elements", epc);
  DL_CHECK_ARG(((uintptr_t)a->X & 15) == 0 && ((uintptr_t)a->W & 15) == 0, DL_ERR_ALIGN,
               "dl_gemm: X/W must be 16-byte aligned");
  if (!a->x_kslow) DL_CHECK_ARG(a->K % epc == 0, DL_ERR_ALIGN, "dl_gemm: K %% %d != 0", epc);
  else DL_CHECK_ARG(a->M % epc == 0, DL_ERR_ALIGN, "dl_gemm: M %% %d != 0 for k-slow X", epc);
  if (!a->w_kslow) DL_CHECK_ARG(a->K % epc == 0, DL_ERR_ALIGN, "dl_gemm: K %% %d != 0", epc);
  else DL_CHECK_ARG(a->N % epc == 0, DL_ERR_ALIGN, "dl_gemm: N %% %d != 0 for k-slow W", epc);
  const int oes = (int)dl_dtype_size(a->out_dtype);
  if (a->N % 8 == 0) {
    DL_CHECK_ARG(a->ldc % 8 == 0 && ((uintptr_t)a->C % 16) == 0, DL_ERR_ALIGN,
                 "dl_gemm: C / ldc not aligned for 16-byte stores (N %% 8 == 0 path)");
    DL_CHECK_ARG(!a->residual || (a->ldr % 8 == 0 && ((uintptr_t)a->residual % 16) == 0), DL_ERR_ALIGN,
                 "dl_gemm: residual not 16-byte aligned");
    DL_CHECK_ARG(!a->pre_out || (a->ldp % 8 == 0 && ((uintptr_t)a->pre_out % 16) == 0), DL_ERR_ALIGN,
                 "dl_gemm: pre_out not 16-byte aligned");
    DL_CHECK_ARG(!a->dact_pre || (a->lddp % 8 == 0 && ((uintptr_t)a->dact_pre % 16) == 0), DL_ERR_ALIGN,
                 "dl_gemm: dact_pre not 16-byte aligned");
    DL_CHECK_ARG(!a->bias || ((uintptr_t)a->bias % 16) == 0, DL_ERR_ALIGN, "dl_gemm: bias not 16-byte aligned");
  }
  DL_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, DL_ERR_ARG, "dl_gemm: dropout_p out of range");
  DL_CHECK_ARG(a->dropout_p == 0.f || a->N % 8 == 0, DL_ERR_SHAPE, "dl_gemm: dropout needs N %% 8 == 0");
  DL_CHECK_ARG(a->algo == DL_GEMM_ALGO_AUTO || a->algo == DL_GEMM_ALGO_TILE128, DL_ERR_ARG, "dl_gemm: bad algo %d", a->algo);

  const int sp = resolve_split(a);
  int tt_bm = 0;
  const bool big_tt = big_tt_plan(a, &tt_bm) > 0;
  if (a->x_colsum)
    DL_CHECK_ARG(a->x_kslow && a->w_kslow && a->split_k == 0, DL_ERR_UNSUPPORTED,
                 "dl_gemm: x_colsum needs the weight-gradient form (x_kslow, w_kslow, split_k = 0)");
  const bool slab_path = sp > 1 || big_tt || a->x_colsum != nullptr;
  if (slab_path) {
    DL_CHECK_ARG(!a->bias && !a->residual && !a->act && !a->pre_out && !a->dact_pre &&
                     a->dropout_p == 0.f && (a->N % 4 == 0),
                 DL_ERR_UNSUPPORTED, "dl_gemm: split_k supports only the plain epilogue, N %% 4 == 0");
    const size_t need = dl_gemm_workspace_bytes(a);
    DL_CHECK_ARG(a->workspace && a->workspace_bytes >= need, DL_ERR_WORKSPACE,
                 "dl_gemm: split_k=%d needs %zu workspace bytes, got %zu", sp, need, a->workspace_bytes);
  }

  GemmP p;
  p.X = (const char*)a->X; p.W = (const char*)a->W; p.C = (char*)a->C;
  p.ldx = a->ldx; p.ldw = a->ldw; p.ldc = a->ldc;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  const int tw = slab_path ? pick_tw(a) : 4;
  const int bt = 32 * tw;
  p.mt = (int)((a->M + bt - 1) / bt); p.nt = (int)((a->N + bt - 1) / bt); p.splits = sp;
  const int bke = BKB / es;
  {
    const int64_t ksteps = (a->K + bke - 1) / bke;
    const int64_t per = (ksteps + sp - 1) / sp;
    p.k_per_split = (int)(per * bke);
  }
  p.bias = a->bias;
  p.res = (const char*)a->residual; p.ldr = a->ldr; p.res_row_mod = (int)a->res_row_mod;
  p.res_before_dropout = a->res_before_dropout;
  p.act = a->act;
  p.pre_out = (char*)a->pre_out; p.ldp = a->ldp;
  p.dact_pre = (const char*)a->dact_pre; p.lddp = a->lddp;
  p.drop_thr16 = a->dropout_p > 0.f ? dl_dropout_thr16(a->dropout_p) : 0u;
  p.drop_inv_keep = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
  p.seed = a->dropout_seed; p.seed_off = a->dropout_seed_offset;
  p.tickets = a->tile_tickets;
  p.accumulate = a->accumulate;
  p.slabs = (float*)a->workspace;
  p.cs_slabs = a->x_colsum ? (float*)a->workspace + (size_t)sp * a->M * a->N : nullptr;
  p.dbg = dl_study_env("DL_GEMM_DBG", 0);     // 0 in the product build (kernels compile the study branches out)
  {
    const double out_mb = (double)a->M * (double)a->N * oes / 1e6;
    p.nt_c = out_mb >= (double)dl_study_env("DL_NT_MIN_MB", 0) ? dl_study_env("DL_NT_MODE", 1) : 0;
    p.nt_pre = dl_study_env("DL_NT_PRE", 1) ? dl_study_env("DL_NT_MODE", 1) : 0;
    p.nt_small = dl_study_env("DL_NT_SMALL", 1);
    p.nt_ext = dl_study_env("DL_NT_EXT", 0);       // residual / saved pre-activation reads of the large-tile epilogues
  }

  if (b) {
    // a pair shares a launch only on the gemm_kernel path without split-K; anything else runs as two launches
    // (the few-tile deep-ring form beats a shared launch of the two-buffer kernel where it applies: 2 x 12.6 us against
    //  27.6 us for two 8192x256x1024 products, tools/pair_bench.py)
    if (slab_path || big_eligible(a, p, sp) || lat_eligible(a, p, sp)) return DL_PAIR_FALLBACK;
    p.nprob = 2;
    p.dX = (const char*)b->X - (const char*)a->X; p.dW = (const char*)b->W - (const char*)a->W;
    p.dC = (char*)b->C - (char*)a->C;
    p.dBias = a->bias ? (const char*)b->bias - (const char*)a->bias : 0;
    p.dRes = a->residual ? (const char*)b->residual - (const char*)a->residual : 0;
    p.dPre = a->pre_out ? (char*)b->pre_out - (char*)a->pre_out : 0;
    p.dDact = a->dact_pre ? (const char*)b->dact_pre - (const char*)a->dact_pre : 0;
    p.dSeed = b->dropout_seed - a->dropout_seed;
  }
  dl_prof_before(0, s);
  int rc = DL_OK;
  if (big_tt) {
    p.mt = (int)((a->M + tt_bm - 1) / tt_bm); p.nt = (int)((a->N + 255) / 256);
    const int64_t ksteps = (a->K + 63) / 64;
    p.k_per_split = (int)(((ksteps + sp - 1) / sp) * 64);
    launch_big_tt(p, s, tt_bm);
  } else if (const int ppk = trickle_ppk(a, p, sp)) {
    p.mt = (int)((a->M + 255) / 256); p.nt = (int)(a->N / 128);
    launch_trickle(p, s, ppk);
  } else if (big_eligible(a, p, sp)) {
    p.mt = (int)((a->M + 255) / 256); p.nt = (int)((a->N + (big_cfg() == 1 ? 127 : 255)) / (big_cfg() == 1 ? 128 : 256));
    launch_big(p, s);
  } else if (const int lat = b ? 0 : lat_eligible(a, p, sp)) {
    p.mt = (int)((a->M + 127) / 128); p.nt = (int)((a->N + 127) / 128);
    launch_lat(p, s, lat);
  } else if (a->in_dtype == DL_BF16) {
    if (slab_path) rc = dispatch_layout<bf16_t, float, true>(a, p, s, tw);
    else if (a->out_dtype == DL_F32) rc = dispatch_layout<bf16_t, float, false>(a, p, s, tw);
    else rc = dispatch_layout<bf16_t, bf16_t, false>(a, p, s, tw);
  } else {
    if (slab_path) rc = dispatch_layout<float, float, true>(a, p, s, tw);
    else rc = dispatch_layout<float, float, false>(a, p, s, tw);
  }
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_gemm");
  {
    // the timing bracket covers the MFMA kernel only (the split-K slab reduction is a separate, HBM-bound launch)
    const double flops = 2.0 * (double)a->M * (double)a->N * (double)a->K * (b ? 2 : 1);
    dl_prof_after(0, s, flops, gemm_algorithmic_bytes(a) * (b ? 2 : 1), a->prof_tag);
  }
  if (slab_path) {
    const int64_t mn = a->M * a->N;
    if (a->deferred) {
      dl_reduce_item& it = *a->deferred;
      it.kind = DL_REDUCE_SPLITK; it.out_dtype = a->out_dtype;
      it.src = (const float*)a->workspace; it.out = a->C; it.mn = mn; it.ldc = a->ldc; it.N = (int32_t)a->N;
      it.splits = sp; it.accumulate = a->accumulate; it.M = a->x_colsum ? (int32_t)a->M : 0;
      it.cs_slabs = p.cs_slabs; it.cs_out = a->x_colsum;
      return DL_OK;
    }
    const int threads = 256;
    const int64_t blocks = (mn / 4 + threads - 1) / threads;
    const int64_t cs_blocks = a->x_colsum ? (a->M + threads - 1) / threads : 0;   // column sums ride along
    if (a->out_dtype == DL_F32)
      hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((uint32_t)(blocks + cs_blocks)), dim3(threads), 0, s,
                         (const float*)a->workspace, (float*)a->C, mn, a->ldc, (int)a->N, sp,
                         a->accumulate, (const float*)p.cs_slabs, a->x_colsum, (int)a->M, (uint32_t)blocks);
    else
      hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((uint32_t)(blocks + cs_blocks)), dim3(threads), 0, s,
                         (const float*)a->workspace, (bf16_t*)a->C, mn, a->ldc, (int)a->N, sp,
                         a->accumulate, (const float*)p.cs_slabs, a->x_colsum, (int)a->M, (uint32_t)blocks);
    DL_CHECK_LAUNCH("dl_gemm(split reduce)");
  } else if (a->deferred) {
    a->deferred->kind = DL_REDUCE_NONE;
  }
  return DL_OK;
}
}  // namespace

// ---- grouped weight gradients (dl_gemm_group) -------------------------------------------------------------
namespace {
bool group_member_ok(const dl_gemm_args* a) {
  const bool plain = !a->bias && !a->residual && !a->act && !a->pre_out && !a->dact_pre && a->dropout_p <= 0.f && !a->accumulate;
  return a->X && a->W && a->C && a->in_dtype == DL_BF16 && a->out_dtype == DL_F32 && a->x_kslow && a->w_kslow && a->split_k == 0 && plain &&
         a->algo == DL_GEMM_ALGO_AUTO && a->M >= 8 && a->N >= 8 && a->M % 8 == 0 && a->N % 8 == 0 && a->K >= 64 &&
         a->M < (1ll << 30) && a->N < (1ll << 30) && a->K < (1ll << 30) && a->ldx % 8 == 0 && a->ldw % 8 == 0 &&
         (((uintptr_t)a->X | (uintptr_t)a->W | (uintptr_t)a->C) & 15) == 0 && a->ldc % 4 == 0;
}
// 128 x 256 tiles; one common slab count `sp` so that all tiles x slabs make about one round of 256 workgroups, at least
// eight 64-row k-steps per slab; per problem trimmed so that no slab is empty.  Returns 0 when the group is not eligible.
int group_plan(const dl_gemm_args* args, int n, int* splits, int* bm_out) {
  if (!args || n < 1 || n > DL_GROUP_MAX) return 0;
  int64_t min_steps = INT64_MAX, big_mn = 0, all_mn = 0, min_m = INT64_MAX;
  for (int i = 0; i < n; ++i) {
    if (!group_member_ok(&args[i])) return 0;
    min_steps = std::min<int64_t>(min_steps, (args[i].K + 63) / 64);
    const int64_t mn = args[i].M * args[i].N;
    all_mn += mn;
    if (mn >= 640 * 1024) big_mn += mn;
    min_m = std::min<int64_t>(min_m, args[i].M);
  }
  // 256-row tiles (four 32-row stages) when most of the group's outputs belong to large members (the d = 512 blocks) over at
  // least 16384 rows: half the W-operand re-reads of the 128-row tile; elsewhere the 128-row tile's parallelism wins.
  // Batch 256 with every block grouped: 15.80 ms with 128-row tiles only, 15.52 with this rule (15.60 with the large
  // products left on their own launches); batch 128: 8.87 -> 8.77; batch 64: 5.79 -> 5.75; batch 32: no difference
  const int want = dl_study_env("DL_GROUP_BM", 0);       // study: 128 / 256 force a tile height
  const int bm = want == 128 ? 128 : (2 * big_mn >= all_mn && min_m >= 256 && min_steps >= dl_study_env("DL_GROUP_BM_MINSTEPS", 256)) ? 256 : 128;
  int64_t tiles = 0;
  for (int i = 0; i < n; ++i) tiles += ((args[i].M + bm - 1) / bm) * ((args[i].N + 255) / 256);
  int64_t sp = 256 / tiles;
  if (sp > min_steps / 8) sp = min_steps / 8;
  if (sp < 1) sp = 1;
  for (int i = 0; i < n; ++i) splits[i] = trim_splits(args[i].K, 64, (int)sp);
  if (bm_out) *bm_out = bm;
  return 1;
}
}  // namespace

extern "C" int dl_gemm_group_plan(const dl_gemm_args* args, int32_t n, int32_t* splits_out) {
  DL_CHECK_ARG(splits_out, DL_ERR_ARG, "dl_gemm_group_plan: null splits_out");
  int sp[DL_GROUP_MAX];
  if (!group_plan(args, n, sp, nullptr)) {
    dl_set_error("dl_gemm_group_plan: not a group of at most %d bf16 weight-gradient products (x_kslow, w_kslow, f32 plain output, split_k = 0, "
                 "M, N multiples of 8, K >= 64, 16-byte aligned operands)", DL_GROUP_MAX);
    return DL_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < n; ++i) splits_out[i] = sp[i];
  return DL_OK;
}

extern "C" int dl_gemm_group(const dl_gemm_args* args, int32_t n, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  int sp[DL_GROUP_MAX], bm = 128;
  DL_CHECK_ARG(group_plan(args, n, sp, &bm), DL_ERR_UNSUPPORTED, "dl_gemm_group: the group is not eligible (see dl_gemm_group_plan)");
  GemmGroupP gp;
  gp.n = n; gp.dbg = dl_study_env("DL_GEMM_DBG", 0);
  uint32_t end = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const dl_gemm_args* a = &args[i];
    const size_t need = (size_t)sp[i] * (size_t)a->M * ((size_t)a->N + (a->x_colsum ? 1 : 0)) * sizeof(float);
    DL_CHECK_ARG(a->workspace && a->workspace_bytes >= need && ((uintptr_t)a->workspace & 15) == 0, DL_ERR_WORKSPACE,
                 "dl_gemm_group: problem %d needs %zu workspace bytes (16-byte aligned), got %zu", i, need, a->workspace_bytes);
    GroupProb& q = gp.q[i];
    q.X = (const char*)a->X; q.W = (const char*)a->W; q.ldx = a->ldx; q.ldw = a->ldw;
    q.M = (int)a->M; q.N = (int)a->N; q.K = (int)a->K;
    q.mt = (int)((a->M + bm - 1) / bm); q.nt = (int)((a->N + 255) / 256);
    const int64_t ksteps = (a->K + 63) / 64;
    q.k_per_split = (int)(((ksteps + sp[i] - 1) / sp[i]) * 64);
    q.slabs = (float*)a->workspace;
    q.cs_slabs = a->x_colsum ? (float*)a->workspace + (size_t)sp[i] * a->M * a->N : nullptr;
    end += (uint32_t)q.mt * q.nt * sp[i];
    q.end = end; q.pad = 0;
    flops += 2.0 * (double)a->M * (double)a->N * (double)a->K;
    bytes += gemm_algorithmic_bytes(a);
  }
  for (int i = n; i < DL_GROUP_MAX; ++i) { gp.q[i] = gp.q[n - 1]; }
  const uint32_t nblocks = end < 256u ? end : 256u;
  dl_prof_before(0, s);
  if (bm == 256) hipLaunchKernelGGL((gemm_big_tt2_kernel<8, 2, 4, true, 32, 4, false, true>), dim3(nblocks), dim3(512), 0, s, gp);
  else hipLaunchKernelGGL((gemm_big_tt2_kernel<4, 2, 4, true, 64, 3, false, true>), dim3(nblocks), dim3(512), 0, s, gp);
  DL_CHECK_LAUNCH("dl_gemm_group");
  dl_prof_after(0, s, flops, bytes, args[0].prof_tag);
  for (int i = 0; i < n; ++i) {
    const dl_gemm_args* a = &args[i];
    const int64_t mn = a->M * a->N;
    if (a->deferred) {
      dl_reduce_item& it = *a->deferred;
      it.kind = DL_REDUCE_SPLITK; it.out_dtype = a->out_dtype;
      it.src = (const float*)a->workspace; it.out = a->C; it.mn = mn; it.ldc = a->ldc; it.N = (int32_t)a->N;
      it.splits = sp[i]; it.accumulate = 0; it.M = a->x_colsum ? (int32_t)a->M : 0;
      it.cs_slabs = gp.q[i].cs_slabs; it.cs_out = a->x_colsum;
      continue;
    }
    const int threads = 256;
    const int64_t blocks = (mn / 4 + threads - 1) / threads;
    const int64_t cs_blocks = a->x_colsum ? (a->M + threads - 1) / threads : 0;
    hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((uint32_t)(blocks + cs_blocks)), dim3(threads), 0, s,
                       (const float*)a->workspace, (float*)a->C, mn, a->ldc, (int)a->N, sp[i], 0,
                       (const float*)gp.q[i].cs_slabs, a->x_colsum, (int)a->M, (uint32_t)blocks);
    DL_CHECK_LAUNCH("dl_gemm_group(split reduce)");
  }
  return DL_OK;
}

// ---- column sums ------------------------------------------------------------------------------
namespace {
// rows per workgroup are chosen per call so that the grid has >= ~1024 workgroups
static inline int cs_rows_per_block(int64_t M, int64_t N) {
  const int64_t colgroups = (N + 255) / 256;
  int64_t chunks = (1024 + colgroups - 1) / colgroups;
  int64_t rows = (M + chunks - 1) / chunks;
  if (rows < 32) rows = 32;
  return (int)((rows + 3) / 4 * 4);
}

// grid: (ceil(N / 256), row_chunks); block 256 threads = 4 waves; lane -> 4 columns, wave -> row phase
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ X, int64_t ldx, int64_t M, int N,
                                      float* __restrict__ partial, int rows_per_block) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 256 + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    if (n + 4 <= N && (N & 3) == 0) {
      for (int64_t r = r0 + wave; r < r1; r += 4) s += load4<T>(X + r * ldx + n);
    } else {
      for (int64_t r = r0 + wave; r < r1; r += 4)
        for (int j = 0; j < 4 && n + j < N; ++j) s[j] += to_f32(X[r * ldx + n + j]);
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* dst = partial + (int64_t)blockIdx.y * N + n;
    for (int j = 0; j < 4 && n + j < N; ++j) dst[j] = s[j];
  }
}
}  // namespace

extern "C" size_t dl_colsum_workspace_bytes(int64_t M, int64_t N) {
  const int rpb = cs_rows_per_block(M, N);
  const int64_t chunks = (M + rpb - 1) / rpb;
  return (size_t)chunks * (size_t)N * sizeof(float);
}

extern "C" int dl_colsum(const void* X, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* out,
                         int32_t accumulate, void* workspace, size_t workspace_bytes, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(X && out && M > 0 && N > 0, DL_ERR_ARG, "dl_colsum: bad args");
  DL_CHECK_ARG(workspace && workspace_bytes >= dl_colsum_workspace_bytes(M, N), DL_ERR_WORKSPACE,
               "dl_colsum: workspace too small");
  if ((N & 3) == 0)
    DL_CHECK_ARG(ldx % 4 == 0 && ((uintptr_t)X % (4 * dl_dtype_size(dtype))) == 0, DL_ERR_ALIGN,
                 "dl_colsum: X not aligned for 4-wide loads");
  const int rpb = cs_rows_per_block(M, N);
  const int chunks = (int)((M + rpb - 1) / rpb);
  dim3 grid((uint32_t)((N + 255) / 256), (uint32_t)chunks);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)X, ldx, M,
                       (int)N, (float*)workspace, rpb);
  else
    hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)X, ldx, M,
                       (int)N, (float*)workspace, rpb);
  DL_CHECK_LAUNCH("dl_colsum(partial)");
  hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((N + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s,
                     (const float*)workspace, chunks, (int64_t)N, (int)N, out, accumulate);
  DL_CHECK_LAUNCH("dl_colsum(final)");
  return DL_OK;
}

// ffn.hip — chained FFN forward (round 4): out = residual + dropout2(b2 + W2 . dropout1(gelu(b1 + W1 . x)))
// in ONE launch, with the 4d-wide hidden activation kept on the chip (reference model/PMMA/mlp.py:44-50 called from
// model/PMMA/block.py:52-60).  What reaches HBM: the pre-activation (bf16, the backward's gelu' operand) and the output.
// The unchained pair (dl_gemm with the GELU + pre-activation epilogue, then dl_gemm with the residual epilogue) writes the
// post-activation too and reads it back: 2 x M x 4d x 2 bytes per call more.
//
//   workgroup : 512 threads = 8 waves, 128 rows of x per tile, persistent over the row tiles (XCD-aware order);
//               the hidden width is walked in chunks of HC = 128 columns:
//     phase 1   H[128][128] = x[128][D] . W1[chunk][D]^T        waves 4(m) x 2(n), wave tile 32 x 64, k-steps of 128
//     epilogue  + b1 -> pre-activation (bf16) to HBM, GELU, dropout -> Hs[128][128] bf16 in LDS (K-contiguous rows)
//     phase 2   Y[128][D] += Hs[128][128] . W2[:, chunk]^T      waves 2(m) x 4(n), wave tile 64 x D/4, k-steps of 64
//   LDS       : two 64 KB stage buffers (phase 1: x tile 32 KB + W1 chunk 32 KB, 256-byte rows; phase 2: W2 slab, 128-byte
//               rows) filled by LDS-DMA one step ahead + Hs 32 KB = 160 KB; one s_barrier per step
//   registers : Y accumulators stay in registers over the whole tile (D = 512: 128 per lane), H accumulators 32 per lane
//   epilogues : accumulators are held transposed (lane = row, 4 consecutive columns per register quad, as in gemm.hip);
//               lane pairs (g, g ^ 1) exchange quads through ds_bpermute so that every lane owns 8 consecutive columns
//               of a row: 16-byte stores without an LDS staging tile (there is no LDS left for one)
// Arithmetic order per output element is that of the unchained pair (same k order, same fp32 accumulation, same
// rounding points, same dropout keys): results are bit-identical to it (tests/test_ffn_gpu.py).
#include <type_traits>
#include "common.cuh"

namespace {
struct FfnP {
  const char* X; const char* W1; const char* W2; const char* res;
  char* pre; char* out;
  const float* b1; const float* b2;
  int64_t ldx, ldr;              // elements
  int M, Hd;
  uint32_t thr16; float inv_keep; uint64_t seed1, seed2; const uint64_t* seed_off;
  int nt;
};

__device__ __forceinline__ void f_wait0() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void f_waitn() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void f_barrier() { asm volatile("s_barrier" ::: "memory"); }
// LDS-DMA, SGPR base + 32-bit VGPR byte offset (the base is uniform; readfirstlane tells the compiler so)
__device__ __forceinline__ void f_dma(uint32_t voff, const char* sbase_, uint32_t lds_off) {
  const uint64_t b = (uint64_t)sbase_;
  const char* sbase = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ int swz256(int row) { return ((row & 3) << 1) | (((row >> 2) & 1) << 3); }   // 256-byte rows (tiles.cuh)

template <int D>
__global__ __launch_bounds__(512, 2) void ffn_fwd_kernel(const FfnP p) {
  typedef bf16_t T;
  constexpr int BM = 128, HC = 128;
  constexpr int STAGE = 65536, XB1 = 32768;          // phase-1 stage: [x 128 rows x 256 B][W1 128 rows x 256 B]
  constexpr int NK1 = D / 128;                       // phase-1 k-steps of 128 elements
  constexpr int WF2 = D / 64;                        // phase-2 MFMA tiles per wave along n (wave tile 64 x D/4)
  constexpr int W2P = D * 8 / 512;                   // phase-2 DMA pieces per thread (D rows x 8 chunks)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + BM * HC * 2];
  char* const hs = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, g = lane >> 4;
  const int wm1 = wave >> 1, wn1 = wave & 1;         // phase 1: 4 x 2 waves of 32 x 64
  const int wm2 = wave >> 2, wn2 = wave & 3;         // phase 2: 2 x 4 waves of 64 x D/4
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  // a lane's byte offset inside an operand tile (same for every piece and step)
  const int r16 = tid >> 4, c16 = tid & 15, r8 = tid >> 3, c8 = tid & 7;
  const uint32_t xvoff = (uint32_t)(r16 * (int)p.ldx * 2 + ((c16 ^ swz256(r16)) << 4));
  const uint32_t w1voff = (uint32_t)(r16 * D * 2 + ((c16 ^ swz256(r16)) << 4));
  const uint32_t w2voff = (uint32_t)(r8 * p.Hd * 2 + ((c8 ^ (r8 & 7)) << 4));

  const int nchunks = p.Hd / HC;
  const int steps_per_chunk = NK1 + 2;
  const uint32_t ntiles = (uint32_t)(p.M / BM), G = gridDim.x;
  const uint64_t s1 = dl_eff_seed(p.seed1, p.seed_off), s2 = dl_eff_seed(p.seed2, p.seed_off);

  auto locate = [&](uint32_t it) -> int {
    const uint32_t round0 = (it / G) * G;
    const uint32_t span = min(G, ntiles - round0);
    return (int)(round0 + xcd_remap(it - round0, span)) * BM;
  };
  // Piece q (0 .. 7) of the operand request of step s (0 .. nchunks * steps_per_chunk - 1) of the tile at row m0 into stage
  // buffer `buf`: one LDS-DMA instruction.  Issued one per MFMA group (back to back after the barrier each of them costs the
  // wave 100-185 cycles with the matrix pipe idle, gemm_big.cuh).  Phase-1 steps have 8 pieces (4 x, 4 W1), phase-2 steps W2P.
  auto issue_piece = [&](int m0, int s, uint32_t buf, int q) {
    const int c = s / steps_per_chunk, k = s - c * steps_per_chunk;
    const uint32_t sb = smem_lds + buf * STAGE;
    if (k < NK1) {
      const int i = q & 3;
      if (q < 4) f_dma(xvoff, p.X + ((int64_t)m0 * p.ldx) * 2 + k * 256 + (int64_t)(32 * i) * p.ldx * 2,
                       __builtin_amdgcn_readfirstlane(sb + (uint32_t)((wave * 64 + i * 512) * 16)));
      else f_dma(w1voff, p.W1 + ((int64_t)c * HC * D) * 2 + k * 256 + (int64_t)(32 * i) * D * 2,
                 __builtin_amdgcn_readfirstlane(sb + (uint32_t)(XB1 + (wave * 64 + i * 512) * 16)));
    } else if (q < W2P) {
      f_dma(w2voff, p.W2 + ((int64_t)c * HC + (k - NK1) * 64) * 2 + (int64_t)(64 * q) * p.Hd * 2,
            __builtin_amdgcn_readfirstlane(sb + (uint32_t)((wave * 64 + q * 512) * 16)));
    }
  };
  auto issue = [&](int m0, int s, uint32_t buf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(m0, s, buf, q);
  };
  // lanes (il, g) and (il, g ^ 1) exchange a quad: even g ends up with tile `ja`'s columns 4g .. 4g + 7, odd g with tile `jb`'s
  // columns 4(g - 1) .. 4(g - 1) + 7 (lo = the first four of the eight)
  auto pair8 = [&](f32x4 ta, f32x4 tb, f32x4& lo, f32x4& hi) {
    const bool odd = g & 1;
    const f32x4 send = odd ? ta : tb, keep = odd ? tb : ta;
    f32x4 got;
#pragma unroll
    for (int e = 0; e < 4; ++e) got[e] = __shfl_xor(send[e], 16, 64);
    lo = odd ? got : keep;
    hi = odd ? keep : got;
  };

  uint32_t it = blockIdx.x;
  if (it >= ntiles) return;
  int m0 = locate(it);
  const int total = nchunks * steps_per_chunk;
  uint32_t gs = 0;                                   // global step counter: stage buffer = gs & 1
  issue(m0, 0, 0);

  for (;;) {
    f32x4 acc2[4][WF2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < WF2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t itn = it + G;
    const bool have_next = itn < ntiles;
    const int m0n = have_next ? locate(itn) : 0;

    for (int c = 0; c < nchunks; ++c) {
      // ---- phase 1 ---------------------------------------------------------------------------------------------------
      f32x4 acc1[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int k = 0; k < NK1; ++k) {
        // (first step of a later tile: the previous tile's output stores were issued AFTER this step's DMA and may stay in flight)
        if (k == 0 && c == 0 && gs != 0) f_waitn<2 * WF2>(); else f_wait0();
        f_barrier();
        const int snext1 = c * steps_per_chunk + k + 1;             // (the step after the last phase-1 step is phase 2's first)
        const uint32_t nbuf1 = (gs + 1) & 1;
        const char* cur = smem + (gs & 1) * STAGE;
        auto rdx = [&](int kf, int i) { const int row = wm1 * 32 + i * 16 + il; return lds_read16(cur, row * 256 + (((kf * 4 + g) ^ swz256(row)) << 4)); };
        auto rdw = [&](int kf, int j) { const int row = wn1 * 64 + j * 16 + il; return lds_read16(cur + XB1, row * 256 + (((kf * 4 + g) ^ swz256(row)) << 4)); };
        u32x4 fx[2], fw[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) fx[i] = rdx(0, i);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fw[j] = rdw(kf, j);
          u32x4 fxn[2] = {fx[0], fx[1]};
          if (kf + 1 < 4) {                                          // next slice's x fragments under this slice's MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) fxn[i] = rdx(kf + 1, i);
          }
          issue_piece(m0, snext1, nbuf1, 2 * kf);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[0][j] = Mma<T>::mma(fw[j], fx[0], acc1[0][j]);
          issue_piece(m0, snext1, nbuf1, 2 * kf + 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[1][j] = Mma<T>::mma(fw[j], fx[1], acc1[1][j]);
          fx[0] = fxn[0]; fx[1] = fxn[1];
        }
        ++gs;
      }
      // ---- epilogue 1: bias, pre-activation out, GELU, dropout, Hs ------------------------------------------------------------
      {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mloc = wm1 * 32 + i * 16 + il;
          const int m = m0 + mloc;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x4 lo, hi;
            pair8(acc1[i][2 * t], acc1[i][2 * t + 1], lo, hi);
            const int nloc = wn1 * 64 + (2 * t + (g & 1)) * 16 + (g >> 1) * 8;     // first of this lane's 8 columns inside the chunk
            const int n = c * HC + nloc;
            const f32x4 ba = *reinterpret_cast<const f32x4*>(p.b1 + n), bb = *reinterpret_cast<const f32x4*>(p.b1 + n + 4);
            lo += ba; hi += bb;
            const u32x4 pv = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
            T* pd = reinterpret_cast<T*>(p.pre) + (int64_t)m * p.Hd + n;
            if (p.nt) store16_nt(pd, pv); else *reinterpret_cast<u32x4*>(pd) = pv;
            lo = gelu4<T>(lo); hi = gelu4<T>(hi);
            if (p.thr16) {
              lo = dl_dropout4(lo, s1, (uint64_t)m, (uint64_t)n, (uint64_t)p.Hd, p.thr16, p.inv_keep);
              hi = dl_dropout4(hi, s1, (uint64_t)m, (uint64_t)(n + 4), (uint64_t)p.Hd, p.thr16, p.inv_keep);
            }
            const u32x4 av = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
            lds_write16(hs, mloc * 256 + ((((nloc >> 3)) ^ swz256(mloc)) << 4), av);
          }
        }
      }
      // ---- phase 2 ---------------------------------------------------------------------------------------------------
#pragma unroll 1
      for (int k2 = 0; k2 < 2; ++k2) {
        if (k2 == 0) f_waitn<4>(); else f_wait0();         // (the four pre-activation stores of this chunk are younger than the awaited DMA)
        f_barrier();                                        // Hs is complete, the stage of this step has landed
        const int snext = c * steps_per_chunk + NK1 + k2 + 1;
        const bool nxt_tile = snext >= total;
        const int sn = nxt_tile ? 0 : snext, mn = nxt_tile ? m0n : m0;
        const bool do_issue = !nxt_tile || have_next;
        const uint32_t nbuf = (gs + 1) & 1;
        const char* cur = smem + (gs & 1) * STAGE;
        auto rdx = [&](int kf, int i) { const int row = wm2 * 64 + i * 16 + il; const int q = (k2 * 2 + kf) * 4 + g; return lds_read16(hs, row * 256 + ((q ^ swz256(row)) << 4)); };
        auto rdw = [&](int kf, int j) { const int row = wn2 * (16 * WF2) + j * 16 + il; return lds_read16(cur, row * 128 + (((kf * 4 + g) ^ (row & 7)) << 4)); };
        // per 32-wide k slice: the four x (Hs) fragments stay, the W2 fragments stream through two registers quads
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
          u32x4 fx[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) fx[i] = rdx(kf, i);
          u32x4 fw = rdw(kf, 0);
#pragma unroll
          for (int j = 0; j < WF2; ++j) {
            const u32x4 fwn = (j + 1 < WF2) ? rdw(kf, j + 1) : fw;
            if (do_issue && (j % (WF2 / 4)) == 0) issue_piece(mn, sn, nbuf, kf * 4 + j / (WF2 / 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[i][j] = Mma<T>::mma(fw, fx[i], acc2[i][j]);
            fw = fwn;
          }
        }
        ++gs;
      }
    }
    // ---- final epilogue: bias, dropout, residual, store ---------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm2 * 64 + i * 16 + il;
#pragma unroll
      for (int t = 0; t < WF2 / 2; ++t) {
        f32x4 lo, hi;
        pair8(acc2[i][2 * t], acc2[i][2 * t + 1], lo, hi);
        const int n = wn2 * (16 * WF2) + (2 * t + (g & 1)) * 16 + (g >> 1) * 8;
        lo += *reinterpret_cast<const f32x4*>(p.b2 + n);
        hi += *reinterpret_cast<const f32x4*>(p.b2 + n + 4);
        if (p.thr16) {
          lo = dl_dropout4(lo, s2, (uint64_t)m, (uint64_t)n, (uint64_t)D, p.thr16, p.inv_keep);
          hi = dl_dropout4(hi, s2, (uint64_t)m, (uint64_t)(n + 4), (uint64_t)D, p.thr16, p.inv_keep);
        }
        if (p.res) {
          const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n);
          lo += f32x4{bf16lo(r[0]), bf16hi(r[0]), bf16lo(r[1]), bf16hi(r[1])};
          hi += f32x4{bf16lo(r[2]), bf16hi(r[2]), bf16lo(r[3]), bf16hi(r[3])};
        }
        const u32x4 o = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        T* od = reinterpret_cast<T*>(p.out) + (int64_t)m * D + n;
        if (p.nt) store16_nt(od, o); else *reinterpret_cast<u32x4*>(od) = o;
      }
    }
    if (!have_next) break;
    it = itn;
    m0 = m0n;
  }
}
}  // namespace

extern "C" int dl_ffn_fwd(const dl_ffn_fwd_args* a, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(a && a->X && a->W1 && a->W2 && a->b1 && a->b2 && a->pre_out && a->out, DL_ERR_ARG, "dl_ffn_fwd: null pointer");
  DL_CHECK_ARG(a->D == 256 || a->D == 512, DL_ERR_UNSUPPORTED, "dl_ffn_fwd: model width %d not in {256, 512}", a->D);
  DL_CHECK_ARG(a->M > 0 && a->M % 128 == 0 && a->Hd > 0 && a->Hd % 128 == 0 && a->M < (1ll << 31), DL_ERR_SHAPE,
               "dl_ffn_fwd: needs M %% 128 == 0 and Hd %% 128 == 0 (M = %ld, Hd = %d)", (long)a->M, a->Hd);
  DL_CHECK_ARG(a->ldx >= a->D && a->ldx % 8 == 0 && (!a->residual || (a->ldr >= a->D && a->ldr % 8 == 0)), DL_ERR_ALIGN,
               "dl_ffn_fwd: row pitches must be multiples of 8 elements");
  DL_CHECK_ARG((((uintptr_t)a->X | (uintptr_t)a->W1 | (uintptr_t)a->W2 | (uintptr_t)a->residual | (uintptr_t)a->pre_out | (uintptr_t)a->out |
                 (uintptr_t)a->b1 | (uintptr_t)a->b2) & 15) == 0, DL_ERR_ALIGN, "dl_ffn_fwd: 16-byte alignment");
  DL_CHECK_ARG(63ll * a->ldx * 2 + 256 < (1ll << 31) && 63ll * a->Hd * 2 + 128 < (1ll << 31), DL_ERR_SHAPE, "dl_ffn_fwd: pitch too large");
  DL_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, DL_ERR_ARG, "dl_ffn_fwd: dropout_p out of range");
  FfnP p;
  p.X = (const char*)a->X; p.W1 = (const char*)a->W1; p.W2 = (const char*)a->W2; p.res = (const char*)a->residual;
  p.pre = (char*)a->pre_out; p.out = (char*)a->out; p.b1 = a->b1; p.b2 = a->b2;
  p.ldx = a->ldx; p.ldr = a->ldr; p.M = (int)a->M; p.Hd = a->Hd;
  p.thr16 = a->dropout_p > 0.f ? dl_dropout_thr16(a->dropout_p) : 0u;
  p.inv_keep = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
  p.seed1 = a->dropout_seed1; p.seed2 = a->dropout_seed2; p.seed_off = a->dropout_seed_offset;
  p.nt = 1;
  const uint32_t ntiles = (uint32_t)(a->M / 128);
  const uint32_t nblocks = ntiles < 256u ? ntiles : 256u;
  dl_prof_before(0, s);
  if (a->D == 512) hipLaunchKernelGGL((ffn_fwd_kernel<512>), dim3(nblocks), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((ffn_fwd_kernel<256>), dim3(nblocks), dim3(512), 0, s, p);
  DL_CHECK_LAUNCH("dl_ffn_fwd");
  {
    const double M = (double)a->M, D = a->D, H = a->Hd;
    dl_prof_after(0, s, 4.0 * M * D * H, (M * D + 2.0 * D * H + M * H + M * D + (a->residual ? M * D : 0.0)) * 2.0 + (D + H) * 4.0, DL_TAG_FFN);
  }
  return DL_OK;
}

// ntxent.hip — streaming NT-Xent (InfoNCE) kernels: nt_xent_loss of the reference
// (model/self_supervised_learning.py:168-182) without ever materialising the (2n)^2 logit matrix, in a form that also
// serves the GLOBAL-batch loss of the data-parallel path (north star: "RCCL all-gather of embeddings so the contrastive
// denominator sees the full global batch").
//
// Two "sides" of rows, each the concatenation [q rows ; k rows] of n rows per half:
//   A (resident): the rows whose loss (forward) or gradient (backward) a launch produces; their MFMA fragments live in
//                 registers for the whole launch (RT 16-row tiles per wave, 4 waves per workgroup);
//   B (streamed): the rows every A row is scored against, 64 at a time through a double-buffered LDS tile filled by
//                 LDS-DMA (source-side XOR swizzle, tiles.cuh layout).
// A row's identity is its GLOBAL index g in [0, 2 n_g): q half first (g < n_g), then k; row r of a side's q half is
// g = gid_offset + r, of its k half g = n_g + gid_offset + r.  Row g's positive is g +- n_g, its own column is excluded
// (the reference removes the diagonal, :172-176).  Single process: A = B = [q; k], offsets 0, n_g = n.  Data parallel:
// A = this rank's rows (offset rank * n), B = the gathered rows (offset 0, n = n_g).
//
// forward : logits of a 64-column tile as S^T = B_tile . A^T on the matrix pipe (bf16: v_mfma_f32_16x16x32_bf16, fp32:
//           exact v_mfma_f32_16x16x4_f32), online log-sum-exp PER LANE in fp32 (a lane keeps its own running maximum
//           over the columns it sees; the four lane groups of a row are combined once at the end: no cross-lane traffic
//           in the loop), loss_i = lse_i - logit(i, pos_i).
// backward: weights w_ij = [lse_A] exp(s_ij - lse_A[i]) + [lse_B] exp(s_ij - lse_B[j]) - (#softmaxes) [j = pos_i] from the
//           recomputed logits, then dA_i += sum_j w_ij B_j on the matrix pipe with w straight from the accumulators
//           (CTILE slot map) and B_j by the transposing LDS read of the SAME tile.  Both softmax terms at once is the
//           single-process symmetric case (one pass gives the whole gradient); the data-parallel path runs it twice:
//           (A = local rows with their lse, B = gathered) for the query-side gradient and (A = gathered, B = local rows
//           with their lse) for the key-side gradient that the reduce-scatter then returns to the owners.
#include <type_traits>
#include "tiles.cuh"

namespace {
using namespace dltile;

struct NtxP {
  const char *aq, *ak, *bq, *bk;
  int64_t na, nb, offa, offb, ng;
  const float *lsea, *lseb;       // backward: per-row log-sum-exp of a side (natural log), or nullptr
  float inv_t, gscale;
  float *row_lse, *row_loss;      // forward outputs (2 na each)
  float *daq, *dak;               // backward outputs (fp32, na x HD each)
};

__device__ __forceinline__ int64_t ntx_gid(int64_t r, int64_t n, int64_t off, int64_t ng) { return r < n ? off + r : ng + off + (r - n); }

template <typename T, int HD, int RT, bool BWD>
__global__ __launch_bounds__(ATT_THREADS) void ntxent_kernel(const NtxP p) {
  using TL = ATile<T, HD>;
  constexpr int ES = (int)sizeof(T);
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NKT = 4, NDT = HD / 16;
  constexpr int CT = (ES == 2) ? 2 : 1;            // 16-row score tiles per contraction fragment of the second product
  constexpr int TILE = 64 * TL::RB;
  constexpr int NCH = 64 * TL::CPR / ATT_THREADS;   // DMA instructions per thread per tile
  static_assert(64 * TL::CPR % ATT_THREADS == 0, "whole DMA instructions per thread");
  // ring of NST tiles, D = NST - 1 tiles in flight: a 64-column tile is ~1000 cycles of work per wave, less than one
  // L2 round trip, so with ONE tile in flight (round-3 first version: 304 TFLOP/s) every tile waited for its data
  constexpr int NST = (TILE <= 16384) ? 4 : 3, D = NST - 1, PERT = NCH + (BWD ? 1 : 0);
  static_assert(D <= 3 && (D - 1) * PERT < 64, "vmcnt immediate");
  __shared__ __attribute__((aligned(16))) char smem[NST * TILE + (BWD ? NST * 256 : 0)];
  const float* lse_s = reinterpret_cast<const float*>(smem + NST * TILE);
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const int64_t na2 = 2 * p.na, nb2 = 2 * p.nb;
  const float c = p.inv_t * LOG2E;

  // resident rows.  jself / jpos: the STREAMED side's row index of this row's own column / of its positive (or a
  // value no tile contains): the per-score index tests then only run in the few tiles that hold one of them
  u32x4 af[RT][NKF];
  int64_t arow[RT];
  int jself[RT], jpos[RT];
  bool aok[RT];
  auto b_index = [&](int64_t gid) -> int {       // row of the streamed side with this global id, or -2^30
    const bool first = gid < p.ng;
    const int64_t r = (first ? gid : gid - p.ng) - p.offb;
    return (r >= 0 && r < p.nb) ? (int)(first ? r : p.nb + r) : -(1 << 30);
  };
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t a = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT) + rt * 16 + il;
    arow[rt] = a; aok[rt] = a < na2;
    const int64_t ac = aok[rt] ? a : 0;
    const T* src = reinterpret_cast<const T*>(ac < p.na ? p.aq : p.ak) + (ac < p.na ? ac : ac - p.na) * HD;
    const int64_t gs = ntx_gid(ac, p.na, p.offa, p.ng);
    jself[rt] = b_index(gs);
    jpos[rt] = b_index(gs < p.ng ? gs + p.ng : gs - p.ng);
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) af[rt][kf] = frag_global<T>(src, aok[rt], kf, g);
  }

  // streamed tiles: LDS slot (row, physical chunk pc) receives the row's logical chunk pc ^ swz(row); rows past the end
  // are clamped to the last row (their scores are masked).  Backward: the tile's 64 log-sum-exp values ride along as one
  // dword DMA (every wave issues it — same bytes, same place — so that all waves count the same number of requests).
  auto issue = [&](int64_t j0, int buf) {
    const uint32_t dst = smem_lds + buf * TILE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * ATT_THREADS, row = ch / TL::CPR, pc = ch % TL::CPR;
      int64_t j = j0 + row; j = j < nb2 ? j : nb2 - 1;
      const char* src = (j < p.nb ? p.bq + j * (int64_t)TL::RB : p.bk + (j - p.nb) * (int64_t)TL::RB) + ((pc ^ TL::swz(row)) << 4);
      lds_dma16(src, __builtin_amdgcn_readfirstlane(dst + (uint32_t)((ch - (tid & 63)) * 16)));
    }
    if constexpr (BWD) {
      int64_t j = j0 + lane; j = j < nb2 ? j : nb2 - 1;
      const void* src = p.lseb ? (const void*)(p.lseb + j) : (const void*)p.bq;
      lds_dma4(src, smem_lds + (uint32_t)(NST * TILE + buf * 256));
    }
  };

  float m_run[RT], l_run[RT], pos_logit[RT], lse_a[RT];
  f32x4 acc[RT][BWD ? NDT : 1];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m_run[rt] = -INFINITY; l_run[rt] = 0.f; pos_logit[rt] = 0.f;
    lse_a[rt] = (BWD && p.lsea && aok[rt]) ? p.lsea[arow[rt]] * LOG2E : INFINITY;
#pragma unroll
    for (int d = 0; d < (BWD ? NDT : 1); ++d) acc[rt][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float npos = BWD ? ((p.lsea ? 1.f : 0.f) + (p.lseb ? 1.f : 0.f)) : 0.f;
  const int wmode = BWD ? ((p.lsea && p.lseb) ? 0 : p.lsea ? 1 : 2) : 0;      // which softmax terms a weight has

  const int64_t nt = (nb2 + 63) / 64;
#pragma unroll
  for (int t0 = 0; t0 < D; ++t0)
    if (t0 < nt) issue((int64_t)t0 * 64, t0);
  int buf = 0;
  for (int64_t t = 0; t < nt; ++t, buf = (buf + 1 == NST) ? 0 : buf + 1) {
    const int64_t j0 = t * 64;
    // tile t has landed once only the younger tiles' requests are outstanding (counted: they stay in flight across the
    // barrier); past the barrier every wave is also done with tile t - 1, whose buffer takes tile t + D
    const int younger = (int)min((int64_t)(D - 1), nt - 1 - t);
    if (younger <= 0) vm_wait<0>();
    else if (younger == 1) vm_wait<PERT>();
    else vm_wait<2 * PERT>();
    raw_barrier();
    if (t + D < nt) issue((t + D) * 64, (buf + D) % NST);
    const char* Bs = smem + buf * TILE;
    f32x4 s[RT][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) s[rt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const u32x4 bf = frag_kc<T, HD>(Bs, kt * 16, kf, il, g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) s[rt][kt] = Mma<T>::mma(bf, af[rt][kf], s[rt][kt]);
      }
    }
    // does this tile hold the own / positive column of one of the wave's rows, or the end of the streamed side?
    const int j0i = (int)j0;
    bool special = j0 + 64 > nb2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) special |= (unsigned)(jself[rt] - j0i) < 64u || (unsigned)(jpos[rt] - j0i) < 64u || !aok[rt];
    const bool slow = __builtin_amdgcn_ballot_w64(special) != 0;          // wave-uniform
    const int jb = j0i + 4 * g;                                          // this lane's columns: jb + kt*16 + r
    if constexpr (!BWD) {
      // two straight-line copies selected by the wave-uniform `slow` (left to the compiler the index tests became ~500
      // compare / select instructions executed on EVERY tile)
      auto softmax_tile = [&](auto slow_) __attribute__((always_inline)) {
        constexpr bool SLOW = decltype(slow_)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if constexpr (SLOW) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int j = jb + kt * 16 + r;
                if (j == jpos[rt]) pos_logit[rt] = s[rt][kt][r] * p.inv_t;
                if (j == jself[rt] || j >= nb2) s[rt][kt][r] = -INFINITY;
              }
          }
          float mx = fmaxf(fmaxf(s[rt][0][0], s[rt][0][1]), fmaxf(s[rt][0][2], s[rt][0][3]));
#pragma unroll
          for (int kt = 1; kt < NKT; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(s[rt][kt][0], s[rt][kt][1]), fmaxf(s[rt][kt][2], s[rt][kt][3])));
          const float m_new = fmaxf(m_run[rt], mx);
          // (a lane whose columns so far were all masked keeps m = -inf: its terms are exp2(-inf) = 0, no NaN as long as
          //  the subtraction is skipped)
          const float mc = (m_new > -INFINITY) ? m_new * c : 0.f;
          float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            rs0 += __builtin_amdgcn_exp2f(__builtin_fmaf(s[rt][kt][0], c, -mc)) + __builtin_amdgcn_exp2f(__builtin_fmaf(s[rt][kt][1], c, -mc));
            rs1 += __builtin_amdgcn_exp2f(__builtin_fmaf(s[rt][kt][2], c, -mc)) + __builtin_amdgcn_exp2f(__builtin_fmaf(s[rt][kt][3], c, -mc));
          }
          const float alpha = (m_run[rt] > -INFINITY) ? __builtin_amdgcn_exp2f(__builtin_fmaf(m_run[rt], c, -mc)) : 0.f;
          l_run[rt] = l_run[rt] * alpha + (rs0 + rs1);
          m_run[rt] = m_new;
        }
      };
      if (!slow) softmax_tile(std::false_type{}); else softmax_tile(std::true_type{});
    } else {
      const float* lj = lse_s + buf * 64;
      // the weight loop in straight-line copies: which softmax terms exist (wmode) and whether any index test is needed
      // are wave-uniform, so they are decided OUTSIDE the 16 x RT scores of a lane
      auto weights = [&](auto mode, auto slow_) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode)::value;
        constexpr bool SLOW = decltype(slow_)::value;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          f32x4 lb = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MODE != 1) lb = *reinterpret_cast<const f32x4*>(lj + kt * 16 + 4 * g) * LOG2E;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sv = s[rt][kt][r];
              float w;
              if constexpr (MODE == 0) w = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -lse_a[rt])) + __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -lb[r]));
              else if constexpr (MODE == 1) w = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -lse_a[rt]));
              else w = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -lb[r]));
              if constexpr (SLOW) {
                const int j = jb + kt * 16 + r;
                if (j == jpos[rt]) w -= npos;
                if (j == jself[rt] || j >= nb2 || !aok[rt]) w = 0.f;
              }
              s[rt][kt][r] = w;
            }
        }
      };
      typedef std::integral_constant<int, 0> M0; typedef std::integral_constant<int, 1> M1; typedef std::integral_constant<int, 2> M2;
      if (!slow) {
        if (wmode == 0) weights(M0{}, std::false_type{}); else if (wmode == 1) weights(M1{}, std::false_type{}); else weights(M2{}, std::false_type{});
      } else {
        if (wmode == 0) weights(M0{}, std::true_type{}); else if (wmode == 1) weights(M1{}, std::true_type{}); else weights(M2{}, std::true_type{});
      }
#pragma unroll
      for (int kp = 0; kp < NKT / CT; ++kp) {
        u32x4 wf[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wf[rt] = frag_from_acc<T>(&s[rt][kp * CT]);
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 bt = frag_tr<T, HD>(Bs, kp * 16 * CT, d * 16, il, g);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][d] = Mma<T>::mma(bt, wf[rt], acc[rt][d]);
        }
      }
    }
  }

  if constexpr (!BWD) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // combine the four lane groups of a row: each holds (max, sum) over its own columns
      const float m_all = group4_max(m_run[rt]);
      const float part = (m_run[rt] > -INFINITY) ? l_run[rt] * exp2f((m_run[rt] - m_all) * c) : 0.f;
      const float l = group4_sum(part);
      const float pl = group4_sum(pos_logit[rt]);
      if (aok[rt] && g == 0) {
        const float lse = m_all * p.inv_t + logf(l);
        p.row_lse[arow[rt]] = lse;
        p.row_loss[arow[rt]] = lse - pl;
      }
    }
  } else {
    const float sc = p.gscale * p.inv_t;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (!aok[rt]) continue;
      const int64_t a = arow[rt];
      float* drow = a < p.na ? p.daq + a * HD : p.dak + (a - p.na) * HD;
#pragma unroll
      for (int d = 0; d < NDT; ++d) *reinterpret_cast<f32x4*>(drow + d * 16 + 4 * g) = acc[rt][d] * sc;
    }
  }
}

__global__ void ntx_vec_sum_kernel(const float* __restrict__ v, int64_t n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 3 * (int64_t)blockDim.x < n; i += 4 * (int64_t)blockDim.x) {
    s0 += v[i]; s1 += v[i + blockDim.x]; s2 += v[i + 2 * (int64_t)blockDim.x]; s3 += v[i + 3 * (int64_t)blockDim.x];
  }
  for (; i < n; i += blockDim.x) s0 += v[i];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    *out = t * scale;
  }
}

int check_side(const dl_ntxent_side& sd, const char* what) {
  DL_CHECK_ARG(sd.q && sd.k && sd.n > 0 && sd.gid_offset >= 0, DL_ERR_ARG, "dl_ntxent: bad %s side", what);
  DL_CHECK_ARG((((uintptr_t)sd.q | (uintptr_t)sd.k) & 15) == 0, DL_ERR_ALIGN, "dl_ntxent: %s rows must be 16-byte aligned", what);
  return DL_OK;
}

template <bool BWD>
int ntx_launch(const dl_ntxent_args* a, NtxP& p, hipStream_t s) {
  const int64_t rows = 2 * a->a.n;
  const bool bf = a->dtype == DL_BF16;
  // bf16: two 16-row tiles per wave (every streamed fragment feeds two products) once that still fills the chip
  const bool rt2 = bf && rows >= 128 * 512;
  const uint32_t blocks = (uint32_t)((rows + (rt2 ? 127 : 63)) / (rt2 ? 128 : 64));
#define DL_NTX(T_, HD_, RT_) hipLaunchKernelGGL((ntxent_kernel<T_, HD_, RT_, BWD>), dim3(blocks), dim3(ATT_THREADS), 0, s, p)
  if (bf) {
    if (a->d == 128) { if (rt2) DL_NTX(bf16_t, 128, 2); else DL_NTX(bf16_t, 128, 1); }
    else { if (rt2) DL_NTX(bf16_t, 64, 2); else DL_NTX(bf16_t, 64, 1); }
  } else {
    if (a->d == 128) DL_NTX(float, 128, 1); else DL_NTX(float, 64, 1);
  }
#undef DL_NTX
  return DL_OK;
}

int ntx_fill(const dl_ntxent_args* a, NtxP& p, const char* what) {
  DL_CHECK_ARG(a, DL_ERR_ARG, "%s: null argument block", what);
  if (int rc = check_side(a->a, "resident")) return rc;
  if (int rc = check_side(a->b, "streamed")) return rc;
  DL_CHECK_ARG(a->d == 64 || a->d == 128, DL_ERR_UNSUPPORTED, "%s: d must be 64 or 128 (got %ld)", what, (long)a->d);
  DL_CHECK_ARG(a->dtype == DL_F32 || a->dtype == DL_BF16, DL_ERR_ARG, "%s: bad dtype", what);
  DL_CHECK_ARG(a->temperature > 0.f, DL_ERR_ARG, "%s: temperature must be > 0", what);
  DL_CHECK_ARG(a->a.n < (1ll << 29) && a->b.n < (1ll << 29), DL_ERR_SHAPE, "%s: at most 2^29 rows per half", what);
  DL_CHECK_ARG(a->n_global >= a->a.gid_offset + a->a.n && a->n_global >= a->b.gid_offset + a->b.n, DL_ERR_SHAPE,
               "%s: a side does not fit the global batch (n_global %ld)", what, (long)a->n_global);
  p.aq = (const char*)a->a.q; p.ak = (const char*)a->a.k; p.bq = (const char*)a->b.q; p.bk = (const char*)a->b.k;
  p.na = a->a.n; p.nb = a->b.n; p.offa = a->a.gid_offset; p.offb = a->b.gid_offset; p.ng = a->n_global;
  p.lsea = a->a.lse; p.lseb = a->b.lse;
  p.inv_t = 1.0f / a->temperature; p.gscale = 0.f;
  p.row_lse = nullptr; p.row_loss = nullptr; p.daq = nullptr; p.dak = nullptr;
  return DL_OK;
}
}  // namespace

extern "C" int dl_ntxent_fwd_ex(const dl_ntxent_args* a, float* row_lse, float* row_loss, float* loss, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  NtxP p;
  if (int rc = ntx_fill(a, p, "dl_ntxent_fwd_ex")) return rc;
  DL_CHECK_ARG(row_lse && row_loss, DL_ERR_ARG, "dl_ntxent_fwd_ex: null output");
  p.row_lse = row_lse; p.row_loss = row_loss;
  dl_prof_before(5, s);
  ntx_launch<false>(a, p, s);
  DL_CHECK_LAUNCH("dl_ntxent_fwd_ex");
  {
    const double ra = 2.0 * (double)a->a.n, rb = 2.0 * (double)a->b.n, es = (double)dl_dtype_size(a->dtype);
    dl_prof_after(5, s, 2.0 * ra * rb * (double)a->d, (ra + rb) * (double)a->d * es + 2.0 * ra * 4.0);
  }
  if (loss) {
    hipLaunchKernelGGL(ntx_vec_sum_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_loss, 2 * a->a.n,
                       1.0f / (float)(2 * a->a.n), loss);
    DL_CHECK_LAUNCH("dl_ntxent_fwd_ex(sum)");
  }
  return DL_OK;
}

extern "C" int dl_ntxent_bwd_ex(const dl_ntxent_args* a, float grad_scale, float* da_q, float* da_k, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  NtxP p;
  if (int rc = ntx_fill(a, p, "dl_ntxent_bwd_ex")) return rc;
  DL_CHECK_ARG(da_q && da_k, DL_ERR_ARG, "dl_ntxent_bwd_ex: null output");
  DL_CHECK_ARG(a->a.lse || a->b.lse, DL_ERR_ARG, "dl_ntxent_bwd_ex: at least one side must carry its log-sum-exp");
  p.daq = da_q; p.dak = da_k; p.gscale = grad_scale;
  dl_prof_before(6, s);
  ntx_launch<true>(a, p, s);
  DL_CHECK_LAUNCH("dl_ntxent_bwd_ex");
  {
    const double ra = 2.0 * (double)a->a.n, rb = 2.0 * (double)a->b.n, es = (double)dl_dtype_size(a->dtype);
    dl_prof_after(6, s, 4.0 * ra * rb * (double)a->d, (ra + rb) * (double)a->d * es + ra * (double)a->d * 4.0);
  }
  return DL_OK;
}

// ---- the round-1 entry points: one process, fp32 rows, A = B = [q; k] ---------------------------------------------
extern "C" size_t dl_ntxent_workspace_bytes(int64_t n, int64_t d) { (void)d; return (size_t)(2 * n) * sizeof(float); }

extern "C" int dl_ntxent_fwd(const float* q, const float* k, int64_t n, int64_t d, float temperature, float* loss,
                             float* row_lse, void* workspace, size_t workspace_bytes, dl_stream stream) {
  DL_CHECK_ARG(q && k && loss && row_lse && n > 0, DL_ERR_ARG, "dl_ntxent_fwd: bad args");
  DL_CHECK_ARG(d == 64 || d == 128, DL_ERR_UNSUPPORTED, "dl_ntxent_fwd: d must be 64 or 128 (got %ld)", (long)d);
  DL_CHECK_ARG(temperature > 0.f, DL_ERR_ARG, "dl_ntxent_fwd: temperature must be > 0");
  DL_CHECK_ARG(workspace && workspace_bytes >= dl_ntxent_workspace_bytes(n, d), DL_ERR_WORKSPACE,
               "dl_ntxent_fwd: workspace too small");
  dl_ntxent_args a = {};
  a.a.q = q; a.a.k = k; a.a.n = n; a.b = a.a; a.n_global = n; a.d = d; a.dtype = DL_F32; a.temperature = temperature;
  return dl_ntxent_fwd_ex(&a, row_lse, (float*)workspace, loss, stream);
}

extern "C" int dl_ntxent_bwd(const float* q, const float* k, int64_t n, int64_t d, float temperature,
                             const float* row_lse, float grad_out, float* dq, float* dk, dl_stream stream) {
  DL_CHECK_ARG(q && k && row_lse && dq && dk && n > 0, DL_ERR_ARG, "dl_ntxent_bwd: bad args");
  DL_CHECK_ARG(d == 64 || d == 128, DL_ERR_UNSUPPORTED, "dl_ntxent_bwd: d must be 64 or 128");
  dl_ntxent_args a = {};
  a.a.q = q; a.a.k = k; a.a.n = n; a.a.lse = row_lse; a.b = a.a; a.n_global = n; a.d = d; a.dtype = DL_F32;
  a.temperature = temperature;
  return dl_ntxent_bwd_ex(&a, grad_out / (float)(2 * n), dq, dk, stream);
}

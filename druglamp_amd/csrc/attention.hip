// attention.hip — fused (flash-style) attention forward/backward for PMMA paired attention, PMMA
// self attention and PGCA cross attention.  See include/druglamp_hip.h for the problem/segment
// addressing.
//
// All three kernels work on TRANSPOSED score tiles so that per-query statistics are lane-local:
//   forward / dQ kernel : S^T[key][q] = K Q^T  (A = K from LDS, B = Q kept in registers)
//                         lane (il, g) holds S^T[key = 4g+r][q = il]; row max/sum are a register
//                         reduction plus two cross-group shuffles; O^T[d][q] = V^T P^T takes P^T
//                         straight from the accumulators (CTILE slot map, common.cuh) and V^T by
//                         the LDS transpose read.  alpha/l/lse scaling of O^T is lane-local.
//   dK/dV kernel        : S[q][key] = Q K^T  (A = Q from LDS, B = K kept in registers), so that
//                         P and dS are directly the B operands of dV^T = dO^T P and dK^T = Q^T dS.
// LDS tiles are [rows][head_dim] with 16-byte chunks XOR-swizzled by ATile::swz(row); the same image
// serves ds_read_b128 (K-contiguous fragments) and ds_read_b64_tr_b16 / ds_read_b32 (transposed
// fragments).
#include "tiles.cuh"

namespace {
using namespace dltile;

// v_exp_f32 without libm's denormal-range fix-up (arguments here are <= 0 and results below 2^-126 may flush);
// exp2(-inf) = 0 as the online softmax needs
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnP {
  const char *Q, *K, *V, *O, *dO;
  char *Out, *dQ, *dK, *dV;
  float *LSE, *Delta, *raw;
  int64_t q_ps, q_hs, q_rs, k_ps, k_hs, k_rs, v_ps, v_hs, v_rs, o_ps, o_hs, o_rs, o_ss;
  int64_t do_ps, do_hs, do_rs, do_ss, dq_ps, dq_hs, dq_rs, dk_ps, dk_hs, dk_rs, dv_ps, dv_hs, dv_rs;
  int P, H, S, shift, Lq, Lk;
  float scale;
  int algo;
  // key multiplicities (round 5): the keys [tail_start, Lk) each stand for w identical keys of the full attention — their
  // scores get log(w) added (tail_bias = log(w) / scale, added to the UNSCALED score so that everything downstream — the
  // running maximum, exp2(s * c - m * c), LSE = m * scale + log(l), the backward's exp2(s * c - lse2) — stays as it is).
  // tail_start == Lk: no such keys.  Tiles in front of tail_start skip the addition through a wave-uniform branch.
  int tail_start;
  float tail_bias;
};

// scores of a 64-key tile in the transposed layout (lane (il, g) holds keys kt * 16 + 4 g + r of query il)
template <int QT, int NKT>
__device__ __forceinline__ void add_tail_bias_t(f32x4 (&s)[QT][NKT], int k0, int g, int tail_start, float tb) {
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (k0 + kt * 16 + 4 * g + r >= tail_start) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt][r] += tb;
      }
}

__device__ __attribute__((aligned(16))) const uint32_t attn_zero_page[4] = {0u, 0u, 0u, 0u};

// LDS-DMA of `total_rows` (a multiple of 64) rows of head_dim elements into an ATile image: source-side XOR
// swizzle, rows >= valid_rows read a zero page.  NT threads; completes at the next __syncthreads().
template <typename T, int HD, int NT>
__device__ __forceinline__ void dma_rows(char* lds, const T* base, int64_t row_stride, int valid_rows, int total_rows) {
  using TL = ATile<T, HD>;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int nchunks = total_rows * TL::CPR;
  const char* zero = reinterpret_cast<const char*>(attn_zero_page);
  for (int c0 = 0; c0 < nchunks; c0 += NT) {
    const int c = c0 + tid;
    const int row = c / TL::CPR, ch = (c % TL::CPR) ^ TL::swz(row);
    const char* src = (c < nchunks && row < valid_rows) ? reinterpret_cast<const char*>(base + (int64_t)row * row_stride + ch * TL::EPC) : zero;
    const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((c0 + wave * 64) * 16));
    if (c0 + wave * 64 < nchunks)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
  }
}
// sum_d a[d] * b[d] over the 8 bf16 of one fragment
__device__ __forceinline__ float dot8_bf16(u32x4 a, u32x4 b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += bf16lo(a[i]) * bf16lo(b[i]) + bf16hi(a[i]) * bf16hi(b[i]);
  return s;
}

// =================================== forward ===================================================
// KVB keys per streamed tile: 64, or 128 for long key sequences at head_dim 64 in bf16 (half the barriers per key).
// OCC = waves per SIMD the kernel is built for: the QT = 1 form fits 128 VGPRs, so FOUR workgroups share a CU — for shapes
// with few query rows per key (the north-star cross-attention shape: 64 queries x 512 keys per head) every workgroup is a
// short chain of dependent HBM round trips, and what hides them is more workgroups in flight, not wider tiles.
template <typename T, int HD, int QT, int KVB = 64, int OCC = 2>
__global__ __launch_bounds__(ATT_THREADS, OCC) void attn_fwd_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int NKT = KVB / 16, NKP = KVB / KF;
  constexpr int QB = 4 * QT * 16;
  constexpr int BUF = 2 * KVB * TL::RB;                 // one (K tile, V tile) pair
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 15, g = lane >> 4;
  const int bps = (p.Lq + QB - 1) / QB;
  const int seg = blockIdx.x / bps, qb = blockIdx.x % bps;
  const int h = blockIdx.y, pr = blockIdx.z;
  const int qprob = seg == 0 ? pr : (pr + p.shift) % p.P;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pr * p.k_ps + (int64_t)h * p.k_hs;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pr * p.v_ps + (int64_t)h * p.v_hs;
  T* Ob = reinterpret_cast<T*>(p.Out) + (int64_t)seg * p.o_ss + (int64_t)pr * p.o_ps + (int64_t)h * p.o_hs;

  const int qw0 = qb * QB + wave * QT * 16;
  u32x4 qf[QT][NKF];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) qf[qt][kf] = frag_global<T>(Qb + (int64_t)q * p.q_rs, q < p.Lq, kf, g);
  }

  f32x4 o[NDT][QT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) o[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }
  const float c = p.scale * LOG2E;

  // key tiles stream through two LDS buffers by LDS-DMA: one barrier per tile, no register staging
  const int nt = (p.Lk + KVB - 1) / KVB;
  auto stage = [&](int t, int buf) {
    char* b = smem + buf * BUF;
    dma_rows<T, HD, ATT_THREADS>(b, Kb + (int64_t)t * KVB * p.k_rs, p.k_rs, p.Lk - t * KVB, KVB);
    dma_rows<T, HD, ATT_THREADS>(b + KVB * TL::RB, Vb + (int64_t)t * KVB * p.v_rs, p.v_rs, p.Lk - t * KVB, KVB);
  };
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * KVB;
    __syncthreads();                                    // tile t has landed; everyone is done with the other buffer
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    const char* Ks = smem + (t & 1) * BUF;
    const char* Vs = Ks + KVB * TL::RB;
    // ---- S^T = K Q^T ----
    f32x4 s[QT][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const u32x4 ka = frag_kc<T, HD>(Ks, kt * 16, kf, il, g);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt] = Mma<T>::mma(ka, qf[qt][kf], s[qt][kt]);
      }
    }
    if (k0 + KVB > p.tail_start) add_tail_bias_t<QT, NKT>(s, k0, g, p.tail_start, p.tail_bias);
    // ---- optional raw logits (segment 0 only) + key masking ----
    const bool tail = (k0 + KVB > p.Lk);
    if (p.raw && seg == 0) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int q = qw0 + qt * 16 + il;
        if (q < p.Lq) {
          float* rrow = p.raw + (((int64_t)pr * p.H + h) * p.Lq + q) * p.Lk;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            const int key = k0 + kt * 16 + 4 * g;
            if (key + 4 <= p.Lk && (p.Lk & 3) == 0)
              *reinterpret_cast<f32x4*>(rrow + key) = s[qt][kt] * p.scale;
            else
              for (int r = 0; r < 4; ++r)
                if (key + r < p.Lk) rrow[key + r] = s[qt][kt][r] * p.scale;
          }
        }
      }
    }
    if (tail) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k0 + kt * 16 + 4 * g + r >= p.Lk) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[qt][kt][r] = -INFINITY;
          }
    }
    // ---- online softmax (per q = il; replicated over g) ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][kt][r]);
      mx = group4_max(mx);
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = fast_exp2((m_run[qt] - m_new) * c);
      const float mc = m_new * c;
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(s[qt][kt][r] * c - mc);
          s[qt][kt][r] = e;
          rs += e;
        }
      l_run[qt] = l_run[qt] * alpha + rs;   // per-lane partial (own keys); reduced at the end
      m_run[qt] = m_new;
#pragma unroll
      for (int d = 0; d < NDT; ++d) o[d][qt] *= alpha;
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
      u32x4 pb[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) pb[qt] = frag_from_acc<T>(&s[qt][kp * CT]);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const u32x4 va = frag_tr<T, HD>(Vs, kp * KF, d * 16, il, g);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[d][qt] = Mma<T>::mma(va, pb[qt], o[d][qt]);
      }
    }
    // (no barrier here: the one at the top of the next iteration is what separates this tile's reads from the DMA that
    //  refills its buffer two iterations later)
  }
  // ---- epilogue ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
    const float l = group4_sum(l_run[qt]);
    const float inv = 1.0f / l;
    if (q < p.Lq) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) store4_fam<4, T>(Ob + (int64_t)q * p.o_rs + d * 16 + 4 * g, o[d][qt] * inv);
      if (p.LSE && g == 0)
        p.LSE[(((int64_t)seg * p.P + pr) * p.H + h) * p.Lq + q] = m_run[qt] * p.scale + logf(l);
    }
  }
}

// =================================== forward, K/V resident ======================================
// Short-key problems (Lk <= LKMAX = 256: PMMA paired and self attention): the whole K and V of one (problem,
// head) are brought into LDS ONCE by LDS-DMA (64 KB at head_dim 64, 128 KB at 128; rows past Lk come from a zero
// page), after which every wave walks its query tiles of BOTH segments over all key tiles with no workgroup
// barrier and no further K/V traffic; the next unit's Q fragments are requested before the current unit's
// math.  Same transposed-score math as attn_fwd_kernel.
template <typename T, int HD, int QT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_res_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int KVB = 64, NKT = KVB / 16, NKP = KVB / KF;
  constexpr int LKMAX = 256, NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) char smem[2 * LKMAX * TL::RB];
  char* Ks = smem;
  char* Vs = smem + LKMAX * TL::RB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, pr = blockIdx.y;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pr * p.k_ps + (int64_t)h * p.k_hs;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pr * p.v_ps + (int64_t)h * p.v_hs;
  const int nt = (p.Lk + KVB - 1) / KVB;
  {
    const int nchunks = nt * KVB * TL::CPR;          // whole key tiles; rows >= Lk are zero
    const char* zero = reinterpret_cast<const char*>(attn_zero_page);
    for (int c0 = 0; c0 < nchunks; c0 += NT) {
      const int c = c0 + tid;
      const int row = c / TL::CPR, ch = (c % TL::CPR) ^ TL::swz(row);
      const bool ok = c < nchunks && row < p.Lk;
      const char* ksrc = ok ? reinterpret_cast<const char*>(Kb + (int64_t)row * p.k_rs + ch * TL::EPC) : zero;
      const char* vsrc = ok ? reinterpret_cast<const char*>(Vb + (int64_t)row * p.v_rs + ch * TL::EPC) : zero;
      const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((c0 + wave * 64) * 16));
      if (c0 + wave * 64 < nchunks) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ksrc,
                                         (__attribute__((address_space(3))) void*)(Ks + off), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vsrc,
                                         (__attribute__((address_space(3))) void*)(Vs + off), 16, 0, 0);
      }
    }
  }
  const int nqt = (p.Lq + 16 * QT - 1) / (16 * QT);   // query units per segment
  const int nunits = p.S * nqt;
  const float c = p.scale * LOG2E;
  auto load_q = [&](int unit, u32x4 (&qf)[QT][NKF]) {
    const int seg = unit / nqt, q0 = (unit % nqt) * 16 * QT;
    const int qprob = seg == 0 ? pr : (pr + p.shift) % p.P;
    const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = q0 + qt * 16 + il;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) qf[qt][kf] = frag_global<T>(Qb + (int64_t)q * p.q_rs, q < p.Lq, kf, g);
    }
  };
  u32x4 qf[QT][NKF], qn[QT][NKF];
  if (wave < nunits) load_q(wave, qf);
  __syncthreads();                                    // fence + barrier: every wave's DMA has landed

  for (int unit = wave; unit < nunits; unit += NW) {
    if (unit + NW < nunits) load_q(unit + NW, qn);
    const int seg = unit / nqt, qw0 = (unit % nqt) * 16 * QT;
    f32x4 o[NDT][QT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) o[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    for (int t = 0; t < nt; ++t) {
      const int k0 = t * KVB;
      const char* Kt = Ks + k0 * TL::RB;
      const char* Vt = Vs + k0 * TL::RB;
      f32x4 s[QT][NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 ka = frag_kc<T, HD>(Kt, kt * 16, kf, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[qt][kt] = Mma<T>::mma(ka, qf[qt][kf], s[qt][kt]);
        }
      }
      if (p.raw && seg == 0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int q = qw0 + qt * 16 + il;
          if (q < p.Lq) {
            float* rrow = p.raw + (((int64_t)pr * p.H + h) * p.Lq + q) * p.Lk;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
              const int key = k0 + kt * 16 + 4 * g;
              if (key + 4 <= p.Lk && (p.Lk & 3) == 0)
                *reinterpret_cast<f32x4*>(rrow + key) = s[qt][kt] * p.scale;
              else
                for (int r = 0; r < 4; ++r)
                  if (key + r < p.Lk) rrow[key + r] = s[qt][kt][r] * p.scale;
            }
          }
        }
      }
      if (k0 + KVB > p.Lk) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + 4 * g + r >= p.Lk) {
#pragma unroll
              for (int qt = 0; qt < QT; ++qt) s[qt][kt][r] = -INFINITY;
            }
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][kt][r]);
        mx = group4_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = fast_exp2((m_run[qt] - m_new) * c);
        const float mc = m_new * c;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(s[qt][kt][r] * c - mc);
            s[qt][kt][r] = e;
            rs += e;
          }
        l_run[qt] = l_run[qt] * alpha + rs;
        m_run[qt] = m_new;
#pragma unroll
        for (int d = 0; d < NDT; ++d) o[d][qt] *= alpha;
      }
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        u32x4 pb[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) pb[qt] = frag_from_acc<T>(&s[qt][kp * CT]);
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 va = frag_tr<T, HD>(Vt, kp * KF, d * 16, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) o[d][qt] = Mma<T>::mma(va, pb[qt], o[d][qt]);
        }
      }
    }
    T* Ob = reinterpret_cast<T*>(p.Out) + (int64_t)seg * p.o_ss + (int64_t)pr * p.o_ps + (int64_t)h * p.o_hs;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qw0 + qt * 16 + il;
      const float l = group4_sum(l_run[qt]);
      const float inv = 1.0f / l;
      if (q < p.Lq) {
#pragma unroll
        for (int d = 0; d < NDT; ++d) store4_fam<4, T>(Ob + (int64_t)q * p.o_rs + d * 16 + 4 * g, o[d][qt] * inv);
        if (p.LSE && g == 0)
          p.LSE[(((int64_t)seg * p.P + pr) * p.H + h) * p.Lq + q] = m_run[qt] * p.scale + logf(l);
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) qf[qt][kf] = qn[qt][kf];
  }
}

// =================================== backward: Delta ===========================================
// Delta(seg,p,h,q) = sum_d dO*O ; 16 lanes per row
template <typename T, int HD>
__global__ void attn_delta_kernel(const AttnP p) {
  const int64_t rows = (int64_t)p.S * p.P * p.H * p.Lq;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (row >= rows) return;
  int64_t t = row;
  const int q = (int)(t % p.Lq); t /= p.Lq;
  const int h = (int)(t % p.H); t /= p.H;
  const int pr = (int)(t % p.P); const int seg = (int)(t / p.P);
  const T* o = reinterpret_cast<const T*>(p.O) + (int64_t)seg * p.o_ss + (int64_t)pr * p.o_ps +
               (int64_t)h * p.o_hs + (int64_t)q * p.o_rs;
  const T* d = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pr * p.do_ps +
               (int64_t)h * p.do_hs + (int64_t)q * p.do_rs;
  float s = 0.f;
#pragma unroll
  for (int c = sub * 4; c < HD; c += 64) {
    const f32x4 a = load4<T>(o + c), b = load4<T>(d + c);
    s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
  if (sub == 0) p.Delta[row] = s;
}

// =================================== backward: dQ ==============================================
// work item: (q tensor of problem pq, head, q block); loops over the attentions that used it.
// RES (bf16, Lk <= 256): the whole K and V of the attention are brought into LDS once per segment by LDS-DMA and
// the key-tile loop runs without workgroup barriers; Delta = sum_d dO * O is computed here from the dO fragments
// the kernel holds anyway (and written out for the dK/dV kernel), so no separate Delta launch is needed.
template <typename T, int HD, int QT, bool RES>
__global__ __launch_bounds__(ATT_THREADS, (HD == 64 ? 2 : 1)) void attn_bwd_dq_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int KVB = 64, NKT = KVB / 16, NKP = KVB / KF;
  constexpr int QB = 4 * QT * 16;
  constexpr int LROWS = RES ? 256 : KVB;
  __shared__ __attribute__((aligned(16))) char smem[2 * LROWS * TL::RB];
  char* Ks = smem;
  char* Vs = smem + LROWS * TL::RB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x, h = blockIdx.y, pq = blockIdx.z;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)pq * p.q_ps + (int64_t)h * p.q_hs;
  const int qw0 = qb * QB + wave * QT * 16;
  const float c = p.scale * LOG2E;

  u32x4 qf[QT][NKF];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) qf[qt][kf] = frag_global<T>(Qb + (int64_t)q * p.q_rs, q < p.Lq, kf, g);
  }
  f32x4 dq[NDT][QT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) dq[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int seg = 0; seg < p.S; ++seg) {
    const int pa = seg == 0 ? pq : (pq + p.P - p.shift) % p.P;   // attention whose segment `seg` used Q(pq)
    const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pa * p.k_ps + (int64_t)h * p.k_hs;
    const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pa * p.v_ps + (int64_t)h * p.v_hs;
    const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps +
                   (int64_t)h * p.do_hs;
    const int64_t statbase = (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
    const int nt = (p.Lk + KVB - 1) / KVB;
    if constexpr (RES) {
      dma_rows<T, HD, ATT_THREADS>(Ks, Kb, p.k_rs, p.Lk, nt * KVB);
      dma_rows<T, HD, ATT_THREADS>(Vs, Vb, p.v_rs, p.Lk, nt * KVB);
    }
    u32x4 dof[QT][NKF];
    float lse2[QT], delta[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qw0 + qt * 16 + il;
      const bool ok = q < p.Lq;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) dof[qt][kf] = frag_global<T>(dOb + (int64_t)q * p.do_rs, ok, kf, g);
      lse2[qt] = ok ? p.LSE[statbase + q] * LOG2E : INFINITY;
      if constexpr (sizeof(T) == 2) {
        // Delta = sum_d dO * O from the dO fragments this kernel holds anyway (written out for the dK/dV kernel that
        // follows on the stream): no separate Delta launch in the bf16 pipelines
        const T* Ob = reinterpret_cast<const T*>(p.O) + (int64_t)seg * p.o_ss + (int64_t)pa * p.o_ps + (int64_t)h * p.o_hs;
        float part = 0.f;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) part += dot8_bf16(dof[qt][kf], frag_global<T>(Ob + (int64_t)q * p.o_rs, ok, kf, g));
        delta[qt] = group4_sum(part);
        if (ok && g == 0) p.Delta[statbase + q] = delta[qt];
      } else {
        delta[qt] = ok ? p.Delta[statbase + q] : 0.f;
      }
    }
    Stager<T, HD, KVB> sk, sv;
    if constexpr (RES) {
      __syncthreads();                                  // fence + barrier: the K/V image has landed
    } else {
      sk.load(Kb, p.k_rs, 0, p.Lk);
      sv.load(Vb, p.v_rs, 0, p.Lk);
    }
    for (int t = 0; t < nt; ++t) {
      const int k0 = t * KVB;
      if constexpr (!RES) {
        sk.store(Ks);
        sv.store(Vs);
        __syncthreads();
        if (t + 1 < nt) {
          sk.load(Kb, p.k_rs, k0 + KVB, p.Lk);
          sv.load(Vb, p.v_rs, k0 + KVB, p.Lk);
        }
      }
      const char* Kt = RES ? Ks + k0 * TL::RB : Ks;
      const char* Vt = RES ? Vs + k0 * TL::RB : Vs;
      f32x4 s[QT][NKT], dp[QT][NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 ka = frag_kc<T, HD>(Kt, kt * 16, kf, il, g);
          const u32x4 va = frag_kc<T, HD>(Vt, kt * 16, kf, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            s[qt][kt] = Mma<T>::mma(ka, qf[qt][kf], s[qt][kt]);
            dp[qt][kt] = Mma<T>::mma(va, dof[qt][kf], dp[qt][kt]);
          }
        }
      }
      if constexpr (!RES) { if (k0 + KVB > p.tail_start) add_tail_bias_t<QT, NKT>(s, k0, g, p.tail_start, p.tail_bias); }
      // dS^T = P^T o (dP^T - delta) ; keys beyond Lk give zero K rows, so they add nothing to dQ
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp2(s[qt][kt][r] * c - lse2[qt]);
            s[qt][kt][r] = pv * (dp[qt][kt][r] - delta[qt]);
          }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        u32x4 db[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) db[qt] = frag_from_acc<T>(&s[qt][kp * CT]);
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 kta = frag_tr<T, HD>(Kt, kp * KF, d * 16, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) dq[d][qt] = Mma<T>::mma(kta, db[qt], dq[d][qt]);
        }
      }
      if constexpr (!RES) __syncthreads();
    }
    if constexpr (RES) { if (seg + 1 < p.S) __syncthreads(); }   // before the next segment's image overwrites this one
  }
  T* dQb = reinterpret_cast<T*>(p.dQ) + (int64_t)pq * p.dq_ps + (int64_t)h * p.dq_hs;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
    if (q < p.Lq) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
        store4_fam<4, T>(dQb + (int64_t)q * p.dq_rs + d * 16 + 4 * g, dq[d][qt] * p.scale);
    }
  }
}

// =================================== backward: dQ, LDS-DMA ring ================================
// head_dim 128 in bf16 (PMMA self attention, PGCA): the register-staged form above holds 378 registers, i.e. ONE wave per
// SIMD, and half of that wave's cycles are waits on the staged K/V tiles that nobody covers (SQ counters).
// Here the K/V tiles arrive by LDS-DMA into a two-stage ring (no staging registers, one barrier per tile) and the scores of
// a 64-key tile are formed in two 32-key halves (half the score accumulators), which fits 256 registers: two workgroups
// per CU.  Same arithmetic in the same order per output element as attn_bwd_dq_kernel.
__device__ __forceinline__ void att_dma16(const char* gsrc, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void att_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void att_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <typename T, int HD, int QT, bool TAILK = false>      // TAILK: key multiplicities (its own instantiation: the plain form keeps its registers)
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_ring_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int KVB = 64, NKP = KVB / KF;
  constexpr int QB = 4 * QT * 16;
  constexpr int TILE = KVB * TL::RB, STAGE = 2 * TILE;
  constexpr int NCH = KVB * TL::CPR / ATT_THREADS;          // DMA instructions per thread per tensor per tile
  static_assert(sizeof(T) == 2 && KVB * TL::CPR % ATT_THREADS == 0, "bf16 tiles, whole DMA instructions");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x, h = blockIdx.y, pq = blockIdx.z;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)pq * p.q_ps + (int64_t)h * p.q_hs;
  const int qw0 = qb * QB + wave * QT * 16;
  const float c = p.scale * LOG2E;

  u32x4 qf[QT][NKF];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) qf[qt][kf] = frag_global<T>(Qb + (int64_t)q * p.q_rs, q < p.Lq, kf, g);
  }
  f32x4 dq[NDT][QT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) dq[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int seg = 0; seg < p.S; ++seg) {
    const int pa = seg == 0 ? pq : (pq + p.P - p.shift) % p.P;
    const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pa * p.k_ps + (int64_t)h * p.k_hs;
    const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pa * p.v_ps + (int64_t)h * p.v_hs;
    const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps + (int64_t)h * p.do_hs;
    const int64_t statbase = (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
    const int nt = (p.Lk + KVB - 1) / KVB;
    auto issue = [&](int t, int buf) {
      const char* zero = reinterpret_cast<const char*>(attn_zero_page);
      const uint32_t sb = smem_lds + (uint32_t)buf * STAGE;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = tid + i * ATT_THREADS;
        const int row = cc / TL::CPR, ch = (cc % TL::CPR) ^ TL::swz(row);
        const int key = t * KVB + row;
        const char* ks = key < p.Lk ? reinterpret_cast<const char*>(Kb + (int64_t)key * p.k_rs + ch * TL::EPC) : zero;
        const char* vs = key < p.Lk ? reinterpret_cast<const char*>(Vb + (int64_t)key * p.v_rs + ch * TL::EPC) : zero;
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * 64 + i * ATT_THREADS) * 16));
        att_dma16(ks, sb + off);
        att_dma16(vs, sb + TILE + off);
      }
    };
    if (seg > 0) att_barrier();                        // every wave is done with the previous segment's last tiles
    issue(0, 0);
    u32x4 dof[QT][NKF];
    float lse2[QT], delta[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qw0 + qt * 16 + il;
      const bool ok = q < p.Lq;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) dof[qt][kf] = frag_global<T>(dOb + (int64_t)q * p.do_rs, ok, kf, g);
      lse2[qt] = ok ? p.LSE[statbase + q] * LOG2E : INFINITY;
      const T* Ob = reinterpret_cast<const T*>(p.O) + (int64_t)seg * p.o_ss + (int64_t)pa * p.o_ps + (int64_t)h * p.o_hs;
      float part = 0.f;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) part += dot8_bf16(dof[qt][kf], frag_global<T>(Ob + (int64_t)q * p.o_rs, ok, kf, g));
      delta[qt] = group4_sum(part);
      if (ok && g == 0) p.Delta[statbase + q] = delta[qt];
    }
    for (int t = 0; t < nt; ++t) {
      att_wait_all();                                   // own pieces of tile t have landed
      att_barrier();                                    // everyone's have, and everyone is done with tile t - 1
      if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
      const char* Kt = smem + (t & 1) * STAGE;
      const char* Vt = Kt + TILE;
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        f32x4 s[QT][CT], dp[QT][CT];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
          const int kt = kp * CT + ci;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) { s[qt][ci] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][ci] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int kf = 0; kf < NKF; ++kf) {
            const u32x4 ka = frag_kc<T, HD>(Kt, kt * 16, kf, il, g);
            const u32x4 va = frag_kc<T, HD>(Vt, kt * 16, kf, il, g);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
              s[qt][ci] = Mma<T>::mma(ka, qf[qt][kf], s[qt][ci]);
              dp[qt][ci] = Mma<T>::mma(va, dof[qt][kf], dp[qt][ci]);
            }
          }
        }
        if constexpr (TAILK) {
          if (t * KVB + (kp + 1) * KF > p.tail_start) add_tail_bias_t<QT, CT>(s, t * KVB + kp * KF, g, p.tail_start, p.tail_bias);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int ci = 0; ci < CT; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = fast_exp2(s[qt][ci][r] * c - lse2[qt]);
              s[qt][ci][r] = pv * (dp[qt][ci][r] - delta[qt]);
            }
        u32x4 db[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) db[qt] = frag_from_acc<T>(&s[qt][0]);
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 kta = frag_tr<T, HD>(Kt, kp * KF, d * 16, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) dq[d][qt] = Mma<T>::mma(kta, db[qt], dq[d][qt]);
        }
      }
    }
  }
  T* dQb = reinterpret_cast<T*>(p.dQ) + (int64_t)pq * p.dq_ps + (int64_t)h * p.dq_hs;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
    if (q < p.Lq) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
        store4_fam<4, T>(dQb + (int64_t)q * p.dq_rs + d * 16 + 4 * g, dq[d][qt] * p.scale);
    }
  }
}

// =================================== backward: dK, dV ==========================================
// work item: (attention problem pa, head, kv block of 4 waves x KT x 16 keys)
// RES (bf16, Lq <= 256): Q, dO, LSE and Delta of one segment are LDS-resident (LDS-DMA, one barrier per segment).
template <typename T, int HD, int KT, bool RES>
__global__ __launch_bounds__(ATT_THREADS, (HD == 64 ? 2 : 1)) void attn_bwd_dkv_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int QSB = 64;                // q rows staged per step
  constexpr int NQT = QSB / 16;          // q tiles per staged block
  constexpr int NQP = QSB / KF;          // contraction fragments per staged block
  constexpr int KVB = 4 * KT * 16;
  constexpr int LROWS = RES ? 256 : QSB;
  __shared__ __attribute__((aligned(16))) char smem[2 * LROWS * TL::RB + 2 * LROWS * sizeof(float)];
  char* Qs = smem;
  char* dOs = smem + LROWS * TL::RB;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * LROWS * TL::RB);
  float* del_s = lse_s + LROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 15, g = lane >> 4;
  const int kb = blockIdx.x, h = blockIdx.y, pa = blockIdx.z;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pa * p.k_ps + (int64_t)h * p.k_hs;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pa * p.v_ps + (int64_t)h * p.v_hs;
  const int kw0 = kb * KVB + wave * KT * 16;
  const float c = p.scale * LOG2E;

  u32x4 kfr[KT][NKF], vfr[KT][NKF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + il;
    const bool ok = key < p.Lk;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      kfr[kt][kf] = frag_global<T>(Kb + (int64_t)key * p.k_rs, ok, kf, g);
      vfr[kt][kf] = frag_global<T>(Vb + (int64_t)key * p.v_rs, ok, kf, g);
    }
  }
  f32x4 dk[NDT][KT], dv[NDT][KT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { dk[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // key multiplicities: this lane's keys (key = kw0 + kt * 16 + il); the whole wave skips the addition in front of tail_start
  const bool has_tail = !RES && kw0 + KT * 16 > p.tail_start;
  float kbl[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) kbl[kt] = (kw0 + kt * 16 + il >= p.tail_start) ? p.tail_bias : 0.f;

  Stager<T, HD, QSB> sq, sdo;
  const int nqb = (p.Lq + QSB - 1) / QSB;
  for (int seg = 0; seg < p.S; ++seg) {
    const int qprob = seg == 0 ? pa : (pa + p.shift) % p.P;
    const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs;
    const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps +
                   (int64_t)h * p.do_hs;
    const int64_t statbase = (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
    if constexpr (RES) {
      if (seg > 0) __syncthreads();                     // everyone is done with the previous segment's image
      dma_rows<T, HD, ATT_THREADS>(Qs, Qb, p.q_rs, p.Lq, nqb * QSB);
      dma_rows<T, HD, ATT_THREADS>(dOs, dOb, p.do_rs, p.Lq, nqb * QSB);
      for (int q = threadIdx.x; q < nqb * QSB; q += ATT_THREADS) {
        lse_s[q] = q < p.Lq ? p.LSE[statbase + q] * LOG2E : INFINITY;
        del_s[q] = q < p.Lq ? p.Delta[statbase + q] : 0.f;
      }
      __syncthreads();
    } else {
      sq.load(Qb, p.q_rs, 0, p.Lq);
      sdo.load(dOb, p.do_rs, 0, p.Lq);
    }
    for (int qb = 0; qb < nqb; ++qb) {
      const int q0 = qb * QSB;
      if constexpr (!RES) {
        sq.store(Qs);
        sdo.store(dOs);
        if (threadIdx.x < QSB) {
          const int q = q0 + threadIdx.x;
          lse_s[threadIdx.x] = q < p.Lq ? p.LSE[statbase + q] * LOG2E : INFINITY;
          del_s[threadIdx.x] = q < p.Lq ? p.Delta[statbase + q] : 0.f;
        }
        __syncthreads();
        if (qb + 1 < nqb) {
          sq.load(Qb, p.q_rs, q0 + QSB, p.Lq);
          sdo.load(dOb, p.do_rs, q0 + QSB, p.Lq);
        }
      }
      const char* Qt = RES ? Qs + q0 * TL::RB : Qs;
      const char* dOt = RES ? dOs + q0 * TL::RB : dOs;
      const float* lse_t = RES ? lse_s + q0 : lse_s;
      const float* del_t = RES ? del_s + q0 : del_s;
      // S[q][key] = Q K^T and dP[q][key] = dO V^T for the staged 64 rows; lane: q = 4g+r, key = il
      f32x4 s[KT][NQT], dp[KT][NQT];
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 qa = frag_kc<T, HD>(Qt, qt * 16, kf, il, g);
          const u32x4 da = frag_kc<T, HD>(dOt, qt * 16, kf, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            s[kt][qt] = Mma<T>::mma(qa, kfr[kt][kf], s[kt][qt]);
            dp[kt][qt] = Mma<T>::mma(da, vfr[kt][kf], dp[kt][qt]);
          }
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_t + qt * 16 + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_t + qt * 16 + 4 * g);
        if (has_tail) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) s[kt][qt] += kbl[kt];
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp2(s[kt][qt][r] * c - l4[r]);
            s[kt][qt][r] = pv;                              // P
            dp[kt][qt][r] = pv * (dp[kt][qt][r] - d4[r]);  // dS
          }
      }
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
      for (int qp = 0; qp < NQP; ++qp) {
        u32x4 pb[KT], sb[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          pb[kt] = frag_from_acc<T>(&s[kt][qp * CT]);
          sb[kt] = frag_from_acc<T>(&dp[kt][qp * CT]);
        }
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 dota = frag_tr<T, HD>(dOt, qp * KF, d * 16, il, g);
          const u32x4 qta = frag_tr<T, HD>(Qt, qp * KF, d * 16, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dv[d][kt] = Mma<T>::mma(dota, pb[kt], dv[d][kt]);
            dk[d][kt] = Mma<T>::mma(qta, sb[kt], dk[d][kt]);
          }
        }
      }
      if constexpr (!RES) __syncthreads();
    }
  }
  T* dKb = reinterpret_cast<T*>(p.dK) + (int64_t)pa * p.dk_ps + (int64_t)h * p.dk_hs;
  T* dVb = reinterpret_cast<T*>(p.dV) + (int64_t)pa * p.dv_ps + (int64_t)h * p.dv_hs;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + il;
    if (key < p.Lk) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        store4_fam<4, T>(dKb + (int64_t)key * p.dk_rs + d * 16 + 4 * g, dk[d][kt] * p.scale);
        store4_fam<4, T>(dVb + (int64_t)key * p.dv_rs + d * 16 + 4 * g, dv[d][kt]);
      }
    }
  }
}

// =================================== backward: dK, dV, LDS-DMA ring ============================
// Same idea as attn_bwd_dq_ring_kernel for the key side: the 64-row Q / dO blocks arrive by LDS-DMA into a two-stage ring
// (their LSE / Delta rows through registers one block ahead), and the scores of a block are formed in two 32-row halves.
template <typename T, int HD, int KT, bool TAILK = false>
__global__ __launch_bounds__(ATT_THREADS, 1) void attn_bwd_dkv_ring_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int QSB = 64, NQP = QSB / KF;
  constexpr int KVB = 4 * KT * 16;
  constexpr int TILE = QSB * TL::RB, STAGE = 2 * TILE;
  constexpr int NCH = QSB * TL::CPR / ATT_THREADS;
  static_assert(sizeof(T) == 2 && QSB * TL::CPR % ATT_THREADS == 0, "bf16 tiles, whole DMA instructions");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 2 * 2 * QSB * sizeof(float)];
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  float* stat_s = reinterpret_cast<float*>(smem + 2 * STAGE);        // [stage][lse | delta][QSB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const int kb = blockIdx.x, h = blockIdx.y, pa = blockIdx.z;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pa * p.k_ps + (int64_t)h * p.k_hs;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pa * p.v_ps + (int64_t)h * p.v_hs;
  const int kw0 = kb * KVB + wave * KT * 16;
  const float c = p.scale * LOG2E;

  u32x4 kfr[KT][NKF], vfr[KT][NKF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + il;
    const bool ok = key < p.Lk;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      kfr[kt][kf] = frag_global<T>(Kb + (int64_t)key * p.k_rs, ok, kf, g);
      vfr[kt][kf] = frag_global<T>(Vb + (int64_t)key * p.v_rs, ok, kf, g);
    }
  }
  f32x4 dk[NDT][KT], dv[NDT][KT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { dk[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // key multiplicities: this lane's keys (key = kw0 + kt * 16 + il); the whole wave skips the addition in front of tail_start
  const bool has_tail = TAILK && kw0 + KT * 16 > p.tail_start;
  float kbl[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) kbl[kt] = (TAILK && kw0 + kt * 16 + il >= p.tail_start) ? p.tail_bias : 0.f;

  const int nqb = (p.Lq + QSB - 1) / QSB;
  for (int seg = 0; seg < p.S; ++seg) {
    const int qprob = seg == 0 ? pa : (pa + p.shift) % p.P;
    const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs;
    const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps + (int64_t)h * p.do_hs;
    const int64_t statbase = (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
    // block qb -> stage qb & 1: Q / dO rows by DMA, LSE / Delta rows by the first 64 threads
    auto issue = [&](int qb, int buf) {
      const char* zero = reinterpret_cast<const char*>(attn_zero_page);
      const uint32_t sb = smem_lds + (uint32_t)buf * STAGE;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = tid + i * ATT_THREADS;
        const int row = cc / TL::CPR, ch = (cc % TL::CPR) ^ TL::swz(row);
        const int q = qb * QSB + row;
        const char* qs_ = q < p.Lq ? reinterpret_cast<const char*>(Qb + (int64_t)q * p.q_rs + ch * TL::EPC) : zero;
        const char* ds_ = q < p.Lq ? reinterpret_cast<const char*>(dOb + (int64_t)q * p.do_rs + ch * TL::EPC) : zero;
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((wave * 64 + i * ATT_THREADS) * 16));
        att_dma16(qs_, sb + off);
        att_dma16(ds_, sb + TILE + off);
      }
      if (tid < QSB) {
        const int q = qb * QSB + tid;
        stat_s[(buf * 2 + 0) * QSB + tid] = q < p.Lq ? p.LSE[statbase + q] * LOG2E : INFINITY;
        stat_s[(buf * 2 + 1) * QSB + tid] = q < p.Lq ? p.Delta[statbase + q] : 0.f;
      }
    };
    if (seg > 0) att_barrier();                        // every wave is done with the previous segment's last blocks
    issue(0, 0);
    for (int qb = 0; qb < nqb; ++qb) {
      att_wait_all();
      att_barrier();
      if (qb + 1 < nqb) issue(qb + 1, (qb + 1) & 1);
      const char* Qt = smem + (qb & 1) * STAGE;
      const char* dOt = Qt + TILE;
      const float* lse_t = stat_s + ((qb & 1) * 2 + 0) * QSB;
      const float* del_t = stat_s + ((qb & 1) * 2 + 1) * QSB;
#pragma unroll
      for (int qp = 0; qp < NQP; ++qp) {
        f32x4 s[KT][CT], dp[KT][CT];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
          const int qt = qp * CT + ci;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) { s[kt][ci] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][ci] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int kf = 0; kf < NKF; ++kf) {
            const u32x4 qa = frag_kc<T, HD>(Qt, qt * 16, kf, il, g);
            const u32x4 da = frag_kc<T, HD>(dOt, qt * 16, kf, il, g);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              s[kt][ci] = Mma<T>::mma(qa, kfr[kt][kf], s[kt][ci]);
              dp[kt][ci] = Mma<T>::mma(da, vfr[kt][kf], dp[kt][ci]);
            }
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_t + qt * 16 + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_t + qt * 16 + 4 * g);
          if (has_tail) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt][ci] += kbl[kt];
          }
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = fast_exp2(s[kt][ci][r] * c - l4[r]);
              s[kt][ci][r] = pv;
              dp[kt][ci][r] = pv * (dp[kt][ci][r] - d4[r]);
            }
        }
        u32x4 pb[KT], sb[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          pb[kt] = frag_from_acc<T>(&s[kt][0]);
          sb[kt] = frag_from_acc<T>(&dp[kt][0]);
        }
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x4 dota = frag_tr<T, HD>(dOt, qp * KF, d * 16, il, g);
          const u32x4 qta = frag_tr<T, HD>(Qt, qp * KF, d * 16, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dv[d][kt] = Mma<T>::mma(dota, pb[kt], dv[d][kt]);
            dk[d][kt] = Mma<T>::mma(qta, sb[kt], dk[d][kt]);
          }
        }
      }
    }
  }
  T* dKb = reinterpret_cast<T*>(p.dK) + (int64_t)pa * p.dk_ps + (int64_t)h * p.dk_hs;
  T* dVb = reinterpret_cast<T*>(p.dV) + (int64_t)pa * p.dv_ps + (int64_t)h * p.dv_hs;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + il;
    if (key < p.Lk) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        store4_fam<4, T>(dKb + (int64_t)key * p.dk_rs + d * 16 + 4 * g, dk[d][kt] * p.scale);
        store4_fam<4, T>(dVb + (int64_t)key * p.dv_rs + d * 16 + 4 * g, dv[d][kt]);
      }
    }
  }
}

// =================================== backward in ONE pass (head_dim 64, bf16, Lk <= 256) =============================
// dQ, dK and dV of an attention from one evaluation of P and dS (the two-kernel form above evaluates the scores, the exp
// and dP twice — once per orientation — and runs 7 tile products where 5 are needed).
//   workgroup = 8 waves; wave w owns keys 32w .. 32w+31 of the attention's (<= 256) keys: K / V fragments and the dK / dV
//   accumulators in registers.
//   per block of 32 query rows:   S = Q K^T, dP = dO V^T  ->  P, dS   (lane: q = 4g + r, key = il: the dK / dV operand form)
//                                 dV^T += dO^T P,  dK^T += Q^T dS
//                                 dS -> LDS as dS^T[key][q] (the packed accumulator fragments are 8-byte rows of it)
//                                 barrier
//                                 dQ^T[d][q] = sum over ALL 256 keys K^T[d][key] dS^T[key][q]: the 32 x 64 block is 8 output
//                                 tiles, one per wave (K^T fragments of its 16 d-columns in registers, dS^T fragments by the
//                                 transposing LDS read) -> final for this (attention, segment): no cross-wave reduction
//   The dS image is double buffered: one barrier per block.
// Paired attention (two segments, partner(partner(p)) = p): ONE workgroup handles the pair {a, b = partner(a)} — passes
// (Q(a), KV(a)), (Q(b), KV(a)), (Q(b), KV(b)), (Q(a), KV(b)) — so both shares of every dQ row come from the same lane of
// the same workgroup, passes apart: the first share is stored (bf16), the second pass adds to it in fp32 and stores the sum.
// dK / dV of an attention accumulate over its two passes in registers.
// Data movement: the register budget (256 per wave at two waves per SIMD) leaves ONE workgroup per CU, so nothing else
// covers a load phase: a first version that loaded a pass's whole Q / dO images up front spent 40 % of its time in those
// prologues (every CU of the chip asking HBM for ~130 KB at the same moment, then none for 30 us).  Here the 32-row Q / dO /
// O tiles and their LSE rows stream through a FOUR-stage LDS ring by LDS-DMA, issued three blocks ahead and counted with
// vmcnt (in-order), so HBM sees an even demand; Delta = sum_d dO * O of a tile is formed from the ring by all threads one
// block before it is used; the partner's K / V images are fetched a whole pass early into their own LDS regions.
__device__ __forceinline__ int ds_swz(int key) { return (((key >> 2) & 1) << 1) | ((key >> 3) & 1); }
template <typename T, int HD>
__global__ __launch_bounds__(512, 1) void attn_bwd_fused_kernel(const AttnP p) {
  static_assert(sizeof(T) == 2 && HD == 64, "bf16, head_dim 64");
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, KT = 2, NQT = 2, QBLK = 32, NT = 512, NST = 4;
  constexpr int TILE = QBLK * TL::RB;                     // 4 KB: one 32-row tile
  constexpr int STAGE = 3 * TILE;                         // Q, dO, O
  constexpr int IMG = 256 * TL::RB;                       // K / V image
  constexpr int DSB = 256 * 64;                           // one dS^T image: 256 keys x 32 q
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 2 * IMG + 2 * DSB + NST * (64 + QBLK) * (int)sizeof(float)];
  char* ring = smem;
  char* Ks = smem + NST * STAGE;
  char* Vs = Ks + IMG;
  char* dSs = Vs + IMG;
  float* lse_s = reinterpret_cast<float*>(dSs + 2 * DSB);  // [NST][64] (raw natural-log LSE; one 64-lane dword DMA per stage, lanes 32-63 repeat)
  float* del_s = lse_s + NST * 64;                          // [NST][32]
  const int tid = threadIdx.x, lane = tid & 63, il = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything keyed on the wave stays on the scalar unit
                                                               // (as a vector value, the Q / dO choice below became per-lane loads
                                                               //  of the kernel arguments from memory, with a full vmcnt(0) wait)
  const int h = blockIdx.y;
  const int a0 = blockIdx.z;                               // S == 2: the pair {a0, a0 + shift}; S == 1: the attention
  const int npass = p.S == 2 ? 4 : 1;
  const int kw0 = wave * KT * 16;
  const int qt_o = wave & 1, dt_o = wave >> 1;            // this wave's dQ output tile of a block
  const float c = p.scale * LOG2E;
  const int nblk = (p.Lq + QBLK - 1) / QBLK;
  const int G = npass * nblk;                              // blocks of the whole workgroup, in order
  const uint32_t smem_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address (for M0)
  const char* zero = reinterpret_cast<const char*>(attn_zero_page);

  // ---- ring DMA, one block per call, in block order: wave w moves rows 8 (w & 3) .. + 7 of the Q (w < 4) or dO (w >= 4) tile,
  // waves 0-3 also the same rows of the O tile, wave 4 the 32 LSE values: two instructions for waves 0-4, one for waves 5-7.
  // The cursor (pass, block, stage) and the pass's base pointers are wave-uniform and advance incrementally (no division,
  // no 64-bit products per block) -------------------------------------------------------------------------------------------
  const int nd = wave <= 4 ? 2 : 1;
  int pf_pass = 0, pf_blk = 0, pf_stage = 0;
  const T* pfA = nullptr;
  const T* pfO = nullptr;
  const float* pfL = nullptr;
  int64_t pfA_rs = 0;
  auto pf_setup = [&]() {
    const int pa = pf_pass < 2 ? a0 : a0 + p.shift;
    const int seg = pf_pass & 1;
    const int qprob = seg == 0 ? pa : (pa + p.shift) % p.P;
    if (wave < 4) { pfA = reinterpret_cast<const T*>(p.Q) + ((int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs); pfA_rs = p.q_rs; }
    else { pfA = reinterpret_cast<const T*>(p.dO) + ((int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps + (int64_t)h * p.do_hs); pfA_rs = p.do_rs; }
    pfO = reinterpret_cast<const T*>(p.O) + ((int64_t)seg * p.o_ss + (int64_t)pa * p.o_ps + (int64_t)h * p.o_hs);
    pfL = p.LSE + (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
  };
  pf_setup();
  auto issue_next = [&]() {
    const uint32_t st = smem_base + (uint32_t)(pf_stage * STAGE);
    const int row = 8 * (wave & 3) + (lane >> 3), q = pf_blk * QBLK + row;
    const int ch = (lane & 7) ^ TL::swz(row);
    const bool ok = q < p.Lq;
    lds_dma16(ok ? reinterpret_cast<const char*>(pfA + (int64_t)q * pfA_rs + ch * TL::EPC) : zero,
              st + (uint32_t)((wave < 4 ? 0 : TILE) + (wave & 3) * 1024));
    if (wave < 4) {
      lds_dma16(ok ? reinterpret_cast<const char*>(pfO + (int64_t)q * p.o_rs + ch * TL::EPC) : zero, st + (uint32_t)(2 * TILE + wave * 1024));
    } else if (wave == 4) {
      // rows beyond Lq read the last valid row's LSE: their Q / dO rows are zero, so any finite value gives P dS = 0 contributions
      const int ql = pf_blk * QBLK + (lane & 31);
      lds_dma4(pfL + (ql < p.Lq ? ql : p.Lq - 1), smem_base + (uint32_t)((char*)lse_s - smem) + (uint32_t)(pf_stage * 256));
    }
    pf_stage = (pf_stage + 1) & (NST - 1);
    if (++pf_blk == nblk) {
      pf_blk = 0;
      if (++pf_pass < npass) pf_setup();
    }
  };
  // K and V images of attention pa (rows >= Lk: zero rows — the dQ product sums over all 256): 8 instructions per wave
  auto issue_kv = [&](int pa) {
    const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)pa * p.k_ps + (int64_t)h * p.k_hs;
    const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)pa * p.v_ps + (int64_t)h * p.v_hs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (i * 8 + wave) * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ TL::swz(row);
      const bool ok = row < p.Lk;
      lds_dma16(ok ? reinterpret_cast<const char*>(Kb + (int64_t)row * p.k_rs + ch * TL::EPC) : zero,
                smem_base + (uint32_t)(Ks - smem) + (uint32_t)((i * 8 + wave) * 1024));
      lds_dma16(ok ? reinterpret_cast<const char*>(Vb + (int64_t)row * p.v_rs + ch * TL::EPC) : zero,
                smem_base + (uint32_t)(Vs - smem) + (uint32_t)((i * 8 + wave) * 1024));
    }
  };
  // Delta of the ring block in stage sg from its dO and O tiles: 16 lanes per row, 4 elements each
  auto form_delta = [&](int sg) {
    const char* st = ring + sg * STAGE;
    const int row = tid >> 4, off = TL::off(row, 4 * (tid & 15));
    const u32x2 a = *reinterpret_cast<const u32x2*>(st + TILE + off), b = *reinterpret_cast<const u32x2*>(st + 2 * TILE + off);
    float part = bf16lo(a[0]) * bf16lo(b[0]) + bf16hi(a[0]) * bf16hi(b[0]) + bf16lo(a[1]) * bf16lo(b[1]) + bf16hi(a[1]) * bf16hi(b[1]);
    part = row16_sum(part);
    if ((tid & 15) == 0) del_s[sg * QBLK + row] = part;
  };
  // ---- prologue: K / V images of the first attention, ring blocks 0..2, Delta of blocks 0 and 1 ---------------------------
  issue_kv(a0);
  for (int t = 0; t < 3 && t < G; ++t) issue_next();
  vm_wait<0>();
  __syncthreads();
  form_delta(0);
  if (G > 1) form_delta(1);
  __syncthreads();
  u32x4 kfr[KT][NKF], vfr[KT][NKF], ktr[8];
  f32x4 dk[NDT][KT], dv[NDT][KT];
  int gb = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int pa = pass < 2 ? a0 : a0 + p.shift;          // the attention (its K, V, dO, O, LSE)
    const int seg = pass & 1;
    const int qprob = seg == 0 ? pa : (pa + p.shift) % p.P;
    const bool new_kv = (pass & 1) == 0;
    if (new_kv) {
      // (the images landed at least one barrier ago: prologue, or issued a whole pass earlier)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          kfr[kt][kf] = frag_kc<T, HD>(Ks, kw0 + kt * 16, kf, il, g);
          vfr[kt][kf] = frag_kc<T, HD>(Vs, kw0 + kt * 16, kf, il, g);
        }
#pragma unroll
      for (int cki = 0; cki < 8; ++cki) ktr[cki] = frag_tr<T, HD>(Ks, cki * 32, dt_o * 16, il, g);
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { dk[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    T* dQb = reinterpret_cast<T*>(p.dQ) + (int64_t)qprob * p.dq_ps + (int64_t)h * p.dq_hs;
    const bool add_share = pass >= 2;                     // this Q's other share was stored two / three passes ago
    for (int blk = 0; blk < nblk; ++blk, ++gb) {
      const int q0 = blk * QBLK;
      const int sg = gb & (NST - 1);
      const char* stg = ring + sg * STAGE;
      const char* Qt = stg;
      const char* dOt = stg + TILE;
      const float* lse_t = lse_s + sg * 64;
      const float* del_t = del_s + sg * QBLK;
      char* dSb = dSs + (gb & 1) * DSB;
      // this lane's 4 dQ outputs of the block; in the adding passes the stored first share is fetched now, a block's work ahead
      // of its use.  (As assembly: a load the compiler sees makes it wait vmcnt(0) — the ring DMA in flight included — before
      // the register is rewritten and again before its use.  Only in the adding passes: a load in the storing passes would
      // leave the row's line in this CU's L1 and the adding pass would read that stale copy.)
      const int qo = q0 + qt_o * 16 + il;
      T* dst = dQb + (int64_t)(qo < p.Lq ? qo : p.Lq - 1) * p.dq_rs + dt_o * 16 + 4 * g;
      u32x2 prevw = {0u, 0u};
      if (add_share) asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(prevw) : "v"(dst) : "memory");
      // the partner's K / V images, a whole pass before they are used (the regions were last read at the start of pass 0)
      const bool kv_now = npass == 4 && pass == 1 && blk == 0;
      if (kv_now) issue_kv(a0 + p.shift);
      // ring: block gb + 3 (its stage was last read in block gb - 1, which every wave has left)
      if (gb + 3 < G) issue_next();
      // S = Q K^T, dP = dO V^T, P, dS one 16-row q tile at a time; the packed P / dS fragments (contraction over the block's
      // 32 rows: words 0-1 <- q tile 0, words 2-3 <- q tile 1) are filled as each tile finishes
      u32x4 pb[KT], sb[KT];
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        f32x4 s[KT], dp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { s[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 qa = frag_kc<T, HD>(Qt, qt * 16, kf, il, g);
          const u32x4 da = frag_kc<T, HD>(dOt, qt * 16, kf, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            s[kt] = Mma<T>::mma(qa, kfr[kt][kf], s[kt]);
            dp[kt] = Mma<T>::mma(da, vfr[kt][kf], dp[kt]);
          }
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_t + qt * 16 + 4 * g) * LOG2E;
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_t + qt * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp2(s[kt][r] * c - l4[r]);
            s[kt][r] = pv;                              // P
            dp[kt][r] = pv * (dp[kt][r] - d4[r]);       // dS
          }
          pb[kt][2 * qt] = pack_bf16x2(s[kt][0], s[kt][1]);
          pb[kt][2 * qt + 1] = pack_bf16x2(s[kt][2], s[kt][3]);
          sb[kt][2 * qt] = pack_bf16x2(dp[kt][0], dp[kt][1]);
          sb[kt][2 * qt + 1] = pack_bf16x2(dp[kt][2], dp[kt][3]);
        }
      }
      // dS^T[key][q] image: row = key (64 bytes = 32 q = four 16-byte chunks: chunk 2 qt + (g >> 1) holds rows 4g .. 4g+3 of q
      // tile qt); the chunk index is XORed with ds_swz(key) = bit 2 of the key -> bit 1, bit 3 -> bit 0.  Keys il, il + 4, il + 8,
      // il + 12 of one 8-byte write (and rows r, r + 4 of one transposing read below) start on the same bank; the four of a
      // write get the four chunk positions, the two of a read different 32-byte halves.  (Rounds 4-5 toggled the half with bit 2
      // only: keys il and il + 8 collided — 25 % of the kernel's LDS-active cycles were bank conflicts, r4_attn_paired_sq_counters.)
      // words 0-1 of the packed fragment are rows 4g .. 4g+3 of q tile 0 at this lane's key, words 2-3 the same of q tile 1
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int key = kw0 + kt * 16 + il, sw = ds_swz(key);
        *reinterpret_cast<u32x2*>(dSb + key * 64 + (((0 + (g >> 1)) ^ sw) << 4) + 8 * (g & 1)) = u32x2{sb[kt][0], sb[kt][1]};
        *reinterpret_cast<u32x2*>(dSb + key * 64 + (((2 + (g >> 1)) ^ sw) << 4) + 8 * (g & 1)) = u32x2{sb[kt][2], sb[kt][3]};
      }
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const u32x4 dota = frag_tr<T, HD>(dOt, 0, d * 16, il, g);
        const u32x4 qta = frag_tr<T, HD>(Qt, 0, d * 16, il, g);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dv[d][kt] = Mma<T>::mma(dota, pb[kt], dv[d][kt]);
          dk[d][kt] = Mma<T>::mma(qta, sb[kt], dk[d][kt]);
        }
      }
      // everything this wave issued before the newest ring block (and the K / V images when they were issued in this
      // block) has landed: in particular ring block gb + 2, which the barrier publishes to the other waves
      if (gb + 3 < G) {
        if (nd == 2) vm_wait<2>(); else vm_wait<1>();
      } else {
        vm_wait<0>();
      }
      __syncthreads();
      if (gb + 2 < G) form_delta((gb + 2) & (NST - 1));     // visible after the NEXT block's barrier, used the block after
      // dQ^T tile (d columns dt_o, q tile qt_o) over all 256 keys
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cki = 0; cki < 8; ++cki) {
        const int r0 = cki * 32 + 4 * g + (il >> 2), sw = ds_swz(r0);        // (row r0 + 16 has the same swizzle)
        const uint32_t off = (uint32_t)(r0 * 64 + (((2 * qt_o + ((il & 3) >> 1)) ^ sw) << 4) + (il & 1) * 8);
        const u32x2 lo = lds_read_tr16(dSb, off);
        const u32x2 hi = lds_read_tr16(dSb, off + 16 * 64);
        dq = Mma<T>::mma(ktr[cki], u32x4{lo[0], lo[1], hi[0], hi[1]}, dq);
      }
      // the first share must be in its registers now.  LDS-DMA and register loads retire out of order with respect to each
      // other, so a counted wait cannot single the load out (measured: vmcnt(nd) let stale registers through): everything
      // this wave has in flight is waited for — the newest ring block was issued a whole block ago and has normally landed
      if (add_share) asm volatile("s_waitcnt vmcnt(0)" : "+v"(prevw) :: "memory");
      if (qo < p.Lq) {
        f32x4 v = dq * p.scale;
        if (add_share) {
          v = v + f32x4{bf16lo(prevw[0]), bf16hi(prevw[0]), bf16lo(prevw[1]), bf16hi(prevw[1])};
          store4_fam<4, T>(dst, v);
        } else if (npass == 1) {
          store4_fam<4, T>(dst, v);
        } else {
          store4<T>(dst, v);                               // read back by this lane in a later pass: keep it cacheable
        }
      }
    }
    if (!new_kv || npass == 1) {
      T* dKb = reinterpret_cast<T*>(p.dK) + (int64_t)pa * p.dk_ps + (int64_t)h * p.dk_hs;
      T* dVb = reinterpret_cast<T*>(p.dV) + (int64_t)pa * p.dv_ps + (int64_t)h * p.dv_hs;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int key = kw0 + kt * 16 + il;
        if (key < p.Lk) {
#pragma unroll
          for (int d = 0; d < NDT; ++d) {
            store4_fam<4, T>(dKb + (int64_t)key * p.dk_rs + d * 16 + 4 * g, dk[d][kt] * p.scale);
            store4_fam<4, T>(dVb + (int64_t)key * p.dv_rs + d * 16 + 4 * g, dv[d][kt]);
          }
        }
      }
    }
  }
}

template <typename T> constexpr int fwd_qt() { return sizeof(T) == 2 ? 2 : 1; }

int check_common(const char* who, int dtype, int head_dim, int nseg, int P, int H, int Lq, int Lk,
                 const int64_t* strides, int nstrides) {
  DL_CHECK_ARG(dtype == DL_F32 || dtype == DL_BF16, DL_ERR_ARG, "%s: bad dtype", who);
  DL_CHECK_ARG(head_dim == 64 || head_dim == 128, DL_ERR_UNSUPPORTED, "%s: head_dim %d not in {64,128}", who, head_dim);
  DL_CHECK_ARG(nseg == 1 || nseg == 2, DL_ERR_ARG, "%s: n_segments must be 1 or 2", who);
  DL_CHECK_ARG(P > 0 && H > 0 && Lq > 0 && Lk > 0 && P <= 65535 && H <= 65535, DL_ERR_SHAPE, "%s: bad sizes", who);
  const int epc = 16 / (int)dl_dtype_size(dtype);
  for (int i = 0; i < nstrides; ++i)
    DL_CHECK_ARG(strides[i] % epc == 0, DL_ERR_ALIGN, "%s: stride #%d (%ld) not a multiple of %d elements", who,
                 i, (long)strides[i], epc);
  return DL_OK;
}

template <typename T, int HD>
int launch_fwd(const AttnP& p, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    // K/V-resident form for short key sequences (algo = DL_ATTN_ALGO_STREAM keeps a call on the streaming form)
    // (measured: at head_dim 128 the 128 KB image leaves one workgroup per CU and ties with the streaming form)
    if (HD == 64 && p.Lk <= 256 && p.algo != DL_ATTN_ALGO_STREAM && p.tail_start >= p.Lk) {
      constexpr int NW = HD == 64 ? 4 : 8;            // 64 KB -> two workgroups per CU; 128 KB -> one of 8 waves
      hipLaunchKernelGGL((attn_fwd_res_kernel<T, HD, 2, NW>), dim3((uint32_t)p.H, (uint32_t)p.P), dim3(64 * NW), 0, s, p);
      return DL_OK;
    }
  }
  constexpr int QT = fwd_qt<T>();
  constexpr int QB = 4 * QT * 16;
  dim3 grid((uint32_t)(p.S * ((p.Lq + QB - 1) / QB)), (uint32_t)p.H, (uint32_t)p.P);
  if constexpr (sizeof(T) == 2 && HD == 64) {
    if (!p.raw && p.Lq <= 64) {                        // one 16-row query tile per wave, four workgroups per CU
      const dim3 g1((uint32_t)(p.S * ((p.Lq + 63) / 64)), (uint32_t)p.H, (uint32_t)p.P);
      hipLaunchKernelGGL((attn_fwd_kernel<T, HD, 1, 64, 4>), g1, dim3(ATT_THREADS), 0, s, p);
      return DL_OK;
    }
    if (!p.raw && p.Lk >= 1024) {                      // long key sequences: 128-key tiles
      hipLaunchKernelGGL((attn_fwd_kernel<T, HD, QT, 128>), grid, dim3(ATT_THREADS), 0, s, p);
      return DL_OK;
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD, QT>), grid, dim3(ATT_THREADS), 0, s, p);
  return DL_OK;
}

template <typename T, int HD>
int launch_bwd(const AttnP& p, hipStream_t s) {
  constexpr int QT = fwd_qt<T>();
  constexpr int QB = 4 * QT * 16;
  constexpr int KT = sizeof(T) == 2 ? 2 : 1;
  constexpr int KVB = 4 * KT * 16;
  const dim3 gq((uint32_t)((p.Lq + QB - 1) / QB), (uint32_t)p.H, (uint32_t)p.P);
  const dim3 gk((uint32_t)((p.Lk + KVB - 1) / KVB), (uint32_t)p.H, (uint32_t)p.P);
  if constexpr (sizeof(T) == 2 && HD == 64) {
    // LDS-resident forms (64 KB images, two workgroups per CU); Delta comes out of the dQ kernel
    // one pass for dQ, dK and dV; a paired attention's two problems {a, partner(a)} share a workgroup.  One 8-wave workgroup
    // per CU: taken when there is at least one workgroup for every CU (measured, paired, Lq = Lk = 256: 256 pairs 361 vs 408 us,
    // 64 pairs 84 vs 108, 32 pairs — half the CUs idle for four serial passes — 73 vs 56), or when asked for
    const int64_t nwg = (int64_t)p.H * (p.S == 2 ? p.shift : p.P);
    if (p.Lk <= 256 && (p.S == 1 || 2 * p.shift == p.P) && p.tail_start >= p.Lk &&
        (p.algo == DL_ATTN_ALGO_ONE_PASS || (p.algo == DL_ATTN_ALGO_AUTO && nwg >= 256))) {
      const dim3 gf(1u, (uint32_t)p.H, (uint32_t)(p.S == 2 ? p.shift : p.P));
      hipLaunchKernelGGL((attn_bwd_fused_kernel<T, HD>), gf, dim3(512), 0, s, p);
      return DL_OK;
    }
    if (p.Lk <= 256 && p.Lq <= 256 && p.algo != DL_ATTN_ALGO_STREAM && p.tail_start >= p.Lk) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, QT, true>), gq, dim3(ATT_THREADS), 0, s, p);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, KT, true>), gk, dim3(ATT_THREADS), 0, s, p);
      return DL_OK;
    }
  }
  if constexpr (sizeof(T) != 2) {                       // fp32 pipelines: Delta by its own pass (bf16: inside the dQ kernel)
    const int64_t rows = (int64_t)p.S * p.P * p.H * p.Lq;
    hipLaunchKernelGGL((attn_delta_kernel<T, HD>), dim3((uint32_t)((rows + 15) / 16)), dim3(256), 0, s, p);
  }
  if constexpr (sizeof(T) == 2 && HD == 128) {
    if (p.tail_start < p.Lk) {              // key multiplicities (PGCA on the distinct drug rows)
      hipLaunchKernelGGL((attn_bwd_dq_ring_kernel<T, HD, QT, true>), gq, dim3(ATT_THREADS), 0, s, p);
      hipLaunchKernelGGL((attn_bwd_dkv_ring_kernel<T, HD, KT, true>), gk, dim3(ATT_THREADS), 0, s, p);
      return DL_OK;
    }
    if (dl_study_env("DL_ATTN_BWD_RING", 3) & 1) hipLaunchKernelGGL((attn_bwd_dq_ring_kernel<T, HD, QT>), gq, dim3(ATT_THREADS), 0, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, QT, false>), gq, dim3(ATT_THREADS), 0, s, p);
    if (dl_study_env("DL_ATTN_BWD_RING", 3) & 2) hipLaunchKernelGGL((attn_bwd_dkv_ring_kernel<T, HD, KT>), gk, dim3(ATT_THREADS), 0, s, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, KT, false>), gk, dim3(ATT_THREADS), 0, s, p);
    return DL_OK;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, QT, false>), gq, dim3(ATT_THREADS), 0, s, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, KT, false>), gk, dim3(ATT_THREADS), 0, s, p);
  return DL_OK;
}

}  // namespace

extern "C" int dl_attn_fwd(const dl_attn_fwd_args* a, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(a && a->Q && a->K && a->V && a->O, DL_ERR_ARG, "dl_attn_fwd: null pointer");
  const int64_t st[] = {a->q_ps, a->q_hs, a->q_rs, a->k_ps, a->k_hs, a->k_rs, a->v_ps, a->v_hs,
                        a->v_rs, a->o_ps, a->o_hs, a->o_rs, a->o_ss};
  int rc = check_common("dl_attn_fwd", a->dtype, a->head_dim, a->n_segments, a->n_problems, a->n_heads,
                        a->Lq, a->Lk, st, 13);
  if (rc != DL_OK) return rc;
  DL_CHECK_ARG(a->n_segments == 1 || (a->partner_shift >= 0 && a->partner_shift < a->n_problems),
               DL_ERR_ARG, "dl_attn_fwd: bad partner_shift");
  AttnP p = {};
  p.Q = (const char*)a->Q; p.K = (const char*)a->K; p.V = (const char*)a->V; p.Out = (char*)a->O;
  p.LSE = a->LSE; p.raw = a->raw_logits;
  p.q_ps = a->q_ps; p.q_hs = a->q_hs; p.q_rs = a->q_rs; p.k_ps = a->k_ps; p.k_hs = a->k_hs; p.k_rs = a->k_rs;
  p.v_ps = a->v_ps; p.v_hs = a->v_hs; p.v_rs = a->v_rs; p.o_ps = a->o_ps; p.o_hs = a->o_hs; p.o_rs = a->o_rs;
  p.o_ss = a->o_ss;
  p.P = a->n_problems; p.H = a->n_heads; p.S = a->n_segments; p.shift = a->partner_shift;
  p.Lq = a->Lq; p.Lk = a->Lk; p.scale = a->scale; p.algo = a->algo;
  DL_CHECK_ARG(a->key_tail_rows >= 0 && a->key_tail_rows <= a->Lk && (a->key_tail_rows == 0 || (a->key_tail_weight >= 1.f && a->scale > 0.f)),
               DL_ERR_ARG, "dl_attn_fwd: key_tail_rows in [0, Lk], key_tail_weight >= 1");
  DL_CHECK_ARG(a->key_tail_rows == 0 || a->n_segments == 1, DL_ERR_UNSUPPORTED, "dl_attn_fwd: key multiplicities with one segment only");
  p.tail_start = a->Lk - a->key_tail_rows;
  p.tail_bias = a->key_tail_rows ? logf(a->key_tail_weight) / a->scale : 0.f;
  dl_prof_before(1, s);
  if (a->dtype == DL_BF16) rc = a->head_dim == 64 ? launch_fwd<bf16_t, 64>(p, s) : launch_fwd<bf16_t, 128>(p, s);
  else rc = a->head_dim == 64 ? launch_fwd<float, 64>(p, s) : launch_fwd<float, 128>(p, s);
  DL_CHECK_LAUNCH("dl_attn_fwd");
  const double nq = (double)a->n_segments * a->n_problems * a->n_heads * a->Lq;
  const int es = (int)dl_dtype_size(a->dtype);
  dl_prof_after(1, s, 4.0 * nq * a->Lk * a->head_dim,
                (2.0 * nq + 2.0 * a->n_problems * a->n_heads * a->Lk) * a->head_dim * es);
  return rc;
}

extern "C" int dl_attn_bwd(const dl_attn_bwd_args* a, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(a && a->Q && a->K && a->V && a->O && a->dO && a->LSE && a->Delta && a->dQ && a->dK && a->dV,
               DL_ERR_ARG, "dl_attn_bwd: null pointer");
  const int64_t st[] = {a->q_ps, a->q_hs, a->q_rs, a->k_ps, a->k_hs, a->k_rs, a->v_ps, a->v_hs, a->v_rs,
                        a->o_ps, a->o_hs, a->o_rs, a->o_ss, a->do_ps, a->do_hs, a->do_rs, a->do_ss,
                        a->dq_ps, a->dq_hs, a->dq_rs, a->dk_ps, a->dk_hs, a->dk_rs, a->dv_ps, a->dv_hs, a->dv_rs};
  int rc = check_common("dl_attn_bwd", a->dtype, a->head_dim, a->n_segments, a->n_problems, a->n_heads,
                        a->Lq, a->Lk, st, 26);
  if (rc != DL_OK) return rc;
  DL_CHECK_ARG(a->n_segments == 1 || (a->partner_shift >= 0 && a->partner_shift < a->n_problems),
               DL_ERR_ARG, "dl_attn_bwd: bad partner_shift");
  AttnP p = {};
  p.Q = (const char*)a->Q; p.K = (const char*)a->K; p.V = (const char*)a->V; p.O = (const char*)a->O;
  p.dO = (const char*)a->dO; p.LSE = const_cast<float*>(a->LSE); p.Delta = a->Delta;
  p.dQ = (char*)a->dQ; p.dK = (char*)a->dK; p.dV = (char*)a->dV;
  p.q_ps = a->q_ps; p.q_hs = a->q_hs; p.q_rs = a->q_rs; p.k_ps = a->k_ps; p.k_hs = a->k_hs; p.k_rs = a->k_rs;
  p.v_ps = a->v_ps; p.v_hs = a->v_hs; p.v_rs = a->v_rs; p.o_ps = a->o_ps; p.o_hs = a->o_hs; p.o_rs = a->o_rs;
  p.o_ss = a->o_ss;
  p.do_ps = a->do_ps; p.do_hs = a->do_hs; p.do_rs = a->do_rs; p.do_ss = a->do_ss;
  p.dq_ps = a->dq_ps; p.dq_hs = a->dq_hs; p.dq_rs = a->dq_rs; p.dk_ps = a->dk_ps; p.dk_hs = a->dk_hs;
  p.dk_rs = a->dk_rs; p.dv_ps = a->dv_ps; p.dv_hs = a->dv_hs; p.dv_rs = a->dv_rs;
  p.P = a->n_problems; p.H = a->n_heads; p.S = a->n_segments; p.shift = a->partner_shift;
  p.Lq = a->Lq; p.Lk = a->Lk; p.scale = a->scale; p.algo = a->algo;
  DL_CHECK_ARG(a->key_tail_rows >= 0 && a->key_tail_rows <= a->Lk && (a->key_tail_rows == 0 || (a->key_tail_weight >= 1.f && a->scale > 0.f)),
               DL_ERR_ARG, "dl_attn_bwd: key_tail_rows in [0, Lk], key_tail_weight >= 1");
  DL_CHECK_ARG(a->key_tail_rows == 0 || a->n_segments == 1, DL_ERR_UNSUPPORTED, "dl_attn_bwd: key multiplicities with one segment only");
  p.tail_start = a->Lk - a->key_tail_rows;
  p.tail_bias = a->key_tail_rows ? logf(a->key_tail_weight) / a->scale : 0.f;
  dl_prof_before(2, s);
  if (a->dtype == DL_BF16) rc = a->head_dim == 64 ? launch_bwd<bf16_t, 64>(p, s) : launch_bwd<bf16_t, 128>(p, s);
  else rc = a->head_dim == 64 ? launch_bwd<float, 64>(p, s) : launch_bwd<float, 128>(p, s);
  DL_CHECK_LAUNCH("dl_attn_bwd");
  const double nq = (double)a->n_segments * a->n_problems * a->n_heads * a->Lq;
  const int es = (int)dl_dtype_size(a->dtype);
  // (algorithmic flops of the backward: the FIVE products S, dP, dV, dK, dQ — 10 nq Lk hd.  Through round 4 this said 14: the
  //  seven products the two-kernel form executes, i.e. its recomputation counted as work)
  dl_prof_after(2, s, 10.0 * nq * a->Lk * a->head_dim,
                (4.0 * nq + 4.0 * a->n_problems * a->n_heads * a->Lk) * a->head_dim * es);
  return rc;
}

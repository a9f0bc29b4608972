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
// LDS tiles are [rows][head_dim] with 16-byte chunks XOR-swizzled by (row & 7); the same image
// serves ds_read_b128 (K-contiguous fragments) and ds_read_b64_tr_b16 / ds_read_b32 (transposed
// fragments).
#include "tiles.cuh"

namespace {
using namespace dltile;

struct AttnP {
  const char *Q, *K, *V, *O, *dO;
  char *Out, *dQ, *dK, *dV;
  float *LSE, *Delta, *raw;
  int64_t q_ps, q_hs, q_rs, k_ps, k_hs, k_rs, v_ps, v_hs, v_rs, o_ps, o_hs, o_rs, o_ss;
  int64_t do_ps, do_hs, do_rs, do_ss, dq_ps, dq_hs, dq_rs, dk_ps, dk_hs, dk_rs, dv_ps, dv_hs, dv_rs;
  int P, H, S, shift, Lq, Lk;
  float scale;
};

// =================================== forward ===================================================
template <typename T, int HD, int QT>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int KVB = 64, NKT = KVB / 16, NKP = KVB / KF;
  constexpr int QB = 4 * QT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * KVB * TL::RB];
  char* Ks = smem;
  char* Vs = smem + KVB * TL::RB;

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

  Stager<T, HD, KVB> sk, sv;
  const int nt = (p.Lk + KVB - 1) / KVB;
  sk.load(Kb, p.k_rs, 0, p.Lk);
  sv.load(Vb, p.v_rs, 0, p.Lk);
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * KVB;
    sk.store(Ks);
    sv.store(Vs);
    __syncthreads();
    if (t + 1 < nt) {
      sk.load(Kb, p.k_rs, k0 + KVB, p.Lk);
      sv.load(Vb, p.v_rs, k0 + KVB, p.Lk);
    }
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
      const float alpha = exp2f((m_run[qt] - m_new) * c);
      const float mc = m_new * c;
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2f(s[qt][kt][r] * c - mc);
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
    __syncthreads();
  }
  // ---- epilogue ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
    const float l = group4_sum(l_run[qt]);
    const float inv = 1.0f / l;
    if (q < p.Lq) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) store4<T>(Ob + (int64_t)q * p.o_rs + d * 16 + 4 * g, o[d][qt] * inv);
      if (p.LSE && g == 0)
        p.LSE[(((int64_t)seg * p.P + pr) * p.H + h) * p.Lq + q] = m_run[qt] * p.scale + logf(l);
    }
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
template <typename T, int HD, int QT>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int KVB = 64, NKT = KVB / 16, NKP = KVB / KF;
  constexpr int QB = 4 * QT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * KVB * TL::RB];
  char* Ks = smem;
  char* Vs = smem + KVB * TL::RB;
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
    u32x4 dof[QT][NKF];
    float lse2[QT], delta[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qw0 + qt * 16 + il;
      const bool ok = q < p.Lq;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) dof[qt][kf] = frag_global<T>(dOb + (int64_t)q * p.do_rs, ok, kf, g);
      lse2[qt] = ok ? p.LSE[statbase + q] * LOG2E : INFINITY;
      delta[qt] = ok ? p.Delta[statbase + q] : 0.f;
    }
    Stager<T, HD, KVB> sk, sv;
    const int nt = (p.Lk + KVB - 1) / KVB;
    sk.load(Kb, p.k_rs, 0, p.Lk);
    sv.load(Vb, p.v_rs, 0, p.Lk);
    for (int t = 0; t < nt; ++t) {
      const int k0 = t * KVB;
      sk.store(Ks);
      sv.store(Vs);
      __syncthreads();
      if (t + 1 < nt) {
        sk.load(Kb, p.k_rs, k0 + KVB, p.Lk);
        sv.load(Vb, p.v_rs, k0 + KVB, p.Lk);
      }
      f32x4 s[QT][NKT], dp[QT][NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 ka = frag_kc<T, HD>(Ks, kt * 16, kf, il, g);
          const u32x4 va = frag_kc<T, HD>(Vs, kt * 16, kf, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            s[qt][kt] = Mma<T>::mma(ka, qf[qt][kf], s[qt][kt]);
            dp[qt][kt] = Mma<T>::mma(va, dof[qt][kf], dp[qt][kt]);
          }
        }
      }
      // dS^T = P^T o (dP^T - delta) ; keys beyond Lk give zero K rows, so they add nothing to dQ
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = exp2f(s[qt][kt][r] * c - lse2[qt]);
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
          const u32x4 kta = frag_tr<T, HD>(Ks, kp * KF, d * 16, il, g);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) dq[d][qt] = Mma<T>::mma(kta, db[qt], dq[d][qt]);
        }
      }
      __syncthreads();
    }
  }
  T* dQb = reinterpret_cast<T*>(p.dQ) + (int64_t)pq * p.dq_ps + (int64_t)h * p.dq_hs;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 16 + il;
    if (q < p.Lq) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
        store4<T>(dQb + (int64_t)q * p.dq_rs + d * 16 + 4 * g, dq[d][qt] * p.scale);
    }
  }
}

// =================================== backward: dK, dV ==========================================
// work item: (attention problem pa, head, kv block of 4 waves x KT x 16 keys)
template <typename T, int HD, int KT>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(const AttnP p) {
  using TL = ATile<T, HD>;
  constexpr int KF = Mma<T>::KF, NKF = HD / KF, NDT = HD / 16, CT = KF / 16;
  constexpr int QSB = 64;                // q rows staged per step
  constexpr int NQT = QSB / 16;          // q tiles per staged block
  constexpr int NQP = QSB / KF;          // contraction fragments per staged block
  constexpr int KVB = 4 * KT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * QSB * TL::RB + 2 * QSB * sizeof(float)];
  char* Qs = smem;
  char* dOs = smem + QSB * TL::RB;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QSB * TL::RB);
  float* del_s = lse_s + QSB;
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

  Stager<T, HD, QSB> sq, sdo;
  const int nqb = (p.Lq + QSB - 1) / QSB;
  for (int seg = 0; seg < p.S; ++seg) {
    const int qprob = seg == 0 ? pa : (pa + p.shift) % p.P;
    const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)qprob * p.q_ps + (int64_t)h * p.q_hs;
    const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)seg * p.do_ss + (int64_t)pa * p.do_ps +
                   (int64_t)h * p.do_hs;
    const int64_t statbase = (((int64_t)seg * p.P + pa) * p.H + h) * p.Lq;
    sq.load(Qb, p.q_rs, 0, p.Lq);
    sdo.load(dOb, p.do_rs, 0, p.Lq);
    for (int qb = 0; qb < nqb; ++qb) {
      const int q0 = qb * QSB;
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
      // S[q][key] = Q K^T and dP[q][key] = dO V^T for the staged 64 rows; lane: q = 4g+r, key = il
      f32x4 s[KT][NQT], dp[KT][NQT];
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          const u32x4 qa = frag_kc<T, HD>(Qs, qt * 16, kf, il, g);
          const u32x4 da = frag_kc<T, HD>(dOs, qt * 16, kf, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            s[kt][qt] = Mma<T>::mma(qa, kfr[kt][kf], s[kt][qt]);
            dp[kt][qt] = Mma<T>::mma(da, vfr[kt][kf], dp[kt][qt]);
          }
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qt * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = exp2f(s[kt][qt][r] * c - l4[r]);
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
          const u32x4 dota = frag_tr<T, HD>(dOs, qp * KF, d * 16, il, g);
          const u32x4 qta = frag_tr<T, HD>(Qs, qp * KF, d * 16, il, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dv[d][kt] = Mma<T>::mma(dota, pb[kt], dv[d][kt]);
            dk[d][kt] = Mma<T>::mma(qta, sb[kt], dk[d][kt]);
          }
        }
      }
      __syncthreads();
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
        store4<T>(dKb + (int64_t)key * p.dk_rs + d * 16 + 4 * g, dk[d][kt] * p.scale);
        store4<T>(dVb + (int64_t)key * p.dv_rs + d * 16 + 4 * g, dv[d][kt]);
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
  constexpr int QT = fwd_qt<T>();
  constexpr int QB = 4 * QT * 16;
  dim3 grid((uint32_t)(p.S * ((p.Lq + QB - 1) / QB)), (uint32_t)p.H, (uint32_t)p.P);
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD, QT>), grid, dim3(ATT_THREADS), 0, s, p);
  return DL_OK;
}

template <typename T, int HD>
int launch_bwd(const AttnP& p, hipStream_t s) {
  {
    const int64_t rows = (int64_t)p.S * p.P * p.H * p.Lq;
    hipLaunchKernelGGL((attn_delta_kernel<T, HD>), dim3((uint32_t)((rows + 15) / 16)), dim3(256), 0, s, p);
  }
  {
    constexpr int QT = fwd_qt<T>();
    constexpr int QB = 4 * QT * 16;
    dim3 grid((uint32_t)((p.Lq + QB - 1) / QB), (uint32_t)p.H, (uint32_t)p.P);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, QT>), grid, dim3(ATT_THREADS), 0, s, p);
  }
  {
    constexpr int KT = (sizeof(T) == 2 && HD == 64) ? 2 : 1;
    constexpr int KVB = 4 * KT * 16;
    dim3 grid((uint32_t)((p.Lk + KVB - 1) / KVB), (uint32_t)p.H, (uint32_t)p.P);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, KT>), grid, dim3(ATT_THREADS), 0, s, p);
  }
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
  p.Lq = a->Lq; p.Lk = a->Lk; p.scale = a->scale;
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
  p.Lq = a->Lq; p.Lk = a->Lk; p.scale = a->scale;
  dl_prof_before(2, s);
  if (a->dtype == DL_BF16) rc = a->head_dim == 64 ? launch_bwd<bf16_t, 64>(p, s) : launch_bwd<bf16_t, 128>(p, s);
  else rc = a->head_dim == 64 ? launch_bwd<float, 64>(p, s) : launch_bwd<float, 128>(p, s);
  DL_CHECK_LAUNCH("dl_attn_bwd");
  const double nq = (double)a->n_segments * a->n_problems * a->n_heads * a->Lq;
  const int es = (int)dl_dtype_size(a->dtype);
  dl_prof_after(2, s, 14.0 * nq * a->Lk * a->head_dim,
                (4.0 * nq + 4.0 * a->n_problems * a->n_heads * a->Lk) * a->head_dim * es);
  return rc;
}

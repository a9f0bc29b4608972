// norm.hip — LayerNorm forward / backward (HBM-bound: one pass over x, 16-byte/8-byte vector loads,
// one wave per row, statistics in fp32).
#include "common.cuh"

namespace {
constexpr int LN_MAXV = 8;            // up to 8 x (64 lanes x 4 elems) = 2048 columns
constexpr int LN_BWD_ROWS = 32;       // rows per workgroup in backward (8 per wave, two in flight)

// One wave per row, LN_FWD_RPW rows per wave with all of their loads issued first (the one-row form was bound by
// one HBM latency per row: 3.5 TB/s).
constexpr int LN_FWD_RPW = 2;
template <typename T, int NV_>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     int64_t ldy, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t M, int D,
                                                     float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * LN_FWD_RPW;
  if (row0 >= M) return;
  const int nv = D >> 2;  // vec4 chunks per row
  f32x4 v[LN_FWD_RPW][NV_];
#pragma unroll
  for (int r = 0; r < LN_FWD_RPW; ++r)
#pragma unroll
    for (int j = 0; j < NV_; ++j) {
      const int c = lane + 64 * j;
      v[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nv && row0 + r < M) v[r][j] = load4<T>(x + (row0 + r) * ldx + c * 4);
    }
  f32x4 gm[NV_], bt[NV_];
#pragma unroll
  for (int j = 0; j < NV_; ++j) {
    const int c = lane + 64 * j;
    gm[j] = bt[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nv) { gm[j] = *reinterpret_cast<const f32x4*>(gamma + c * 4); bt[j] = *reinterpret_cast<const f32x4*>(beta + c * 4); }
  }
#pragma unroll
  for (int r = 0; r < LN_FWD_RPW; ++r) {
    const int64_t row = row0 + r;
    if (row >= M) break;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV_; ++j) s += (v[r][j][0] + v[r][j][1]) + (v[r][j][2] + v[r][j][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV_; ++j) {
      const int c = lane + 64 * j;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[r][j][e] - mean; q += d * d; }
      }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < NV_; ++j) {
      const int c = lane + 64 * j;
      if (c < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[r][j][e] - mean) * rstd * gm[j][e] + bt[j][e];
        store4<T>(y + row * ldy + c * 4, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// partial layout: [nblocks][2][D]  (0: dgamma, 1: dbeta)
template <typename T, int NV_>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int64_t lddy, int64_t dy_share,
                                                     const T* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma,
                                                     const T* __restrict__ dres, int64_t lddres,
                                                     T* __restrict__ dx, int64_t lddx,
                                                     float* __restrict__ partial, int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) char ln_smem[];  // [4 waves][2][D] floats
  float* sm = reinterpret_cast<float*>(ln_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 2;
  f32x4 gm[NV_], dg[NV_], db[NV_];
#pragma unroll
  for (int j = 0; j < NV_; ++j) {
    const int c = lane + 64 * j;
    dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gm[j] = (c < nv) ? *reinterpret_cast<const f32x4*>(gamma + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t r0 = (int64_t)blockIdx.x * LN_BWD_ROWS;
  // two rows per iteration: both rows' loads are issued before either row's reductions, so one HBM latency is
  // paid per PAIR of rows (the kernel was latency-bound at one row per wave at a time)
  for (int rr = wave; rr < LN_BWD_ROWS; rr += 8) {
    const int64_t rowA = r0 + rr, rowB = r0 + rr + 4;
    if (rowA >= M) break;
    const bool hasB = rowB < M;
    // dy_share consecutive rows of x take the same dy row (the gradient of a mean over tokens, never materialised)
    const int64_t dyA = dy_share == 1 ? rowA : rowA / dy_share, dyB = dy_share == 1 ? rowB : rowB / dy_share;
    f32x4 xa[NV_], da[NV_], xb[NV_], db_[NV_], ra[NV_], rb[NV_];
#pragma unroll
    for (int j = 0; j < NV_; ++j) {
      const int c = lane + 64 * j;
      ra[j] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[j] = ra[j];
      if (c < nv) {
        xa[j] = load4<T>(x + rowA * ldx + c * 4);
        da[j] = load4<T>(dy + dyA * lddy + c * 4);
        if (dres) ra[j] = load4<T>(dres + rowA * lddres + c * 4);
        if (hasB) {
          xb[j] = load4<T>(x + rowB * ldx + c * 4);
          db_[j] = load4<T>(dy + dyB * lddy + c * 4);
          if (dres) rb[j] = load4<T>(dres + rowB * lddres + c * 4);
        }
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && !hasB) break;
      const int64_t row = half == 0 ? rowA : rowB;
      const float mu = mean[row], rs = rstd[row];
      f32x4 xh[NV_], g[NV_];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV_; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
          const f32x4 xv = half == 0 ? xa[j] : xb[j];
          const f32x4 dv = half == 0 ? da[j] : db_[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[j][e] = (xv[e] - mu) * rs;
            g[j][e] = dv[e] * gm[j][e];
            c1 += g[j][e];
            c2 += g[j][e] * xh[j][e];
            dg[j][e] += dv[e] * xh[j][e];
            db[j][e] += dv[e];
          }
        }
      }
      c1 = wave_sum(c1) / (float)D;
      c2 = wave_sum(c2) / (float)D;
#pragma unroll
      for (int j = 0; j < NV_; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (g[j][e] - c1 - xh[j][e] * c2);
          o += half == 0 ? ra[j] : rb[j];
          store4<T>(dx + row * lddx + c * 4, o);
        }
      }
    }
  }
  // cross-wave reduction of dgamma/dbeta through LDS
#pragma unroll
  for (int j = 0; j < NV_; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      *reinterpret_cast<f32x4*>(sm + (wave * 2 + 0) * D + c * 4) = dg[j];
      *reinterpret_cast<f32x4*>(sm + (wave * 2 + 1) * D + c * 4) = db[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, col = i % D;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += sm[(w * 2 + which) * D + col];
    partial[((int64_t)blockIdx.x * 2 + which) * D + col] = s;
  }
}

// ---- bf16 rows of 256 / 512 columns (every LayerNorm of the path) with 16-byte accesses --------------------------------
// A lane owns 8 consecutive columns (one 16-byte load / store per tensor and row): 32 lanes per 256-column row (two
// rows per wave pass), 64 per 512-column row, four passes in flight per wave.  The 8-byte one-wave-per-row kernels above
// reached 2.7 TB/s (backward) / 3.9-4.9 TB/s (forward) — 8-byte accesses run at 0.54-0.70 of the 16-byte rate.
template <int LPR> __device__ __forceinline__ float seg_sum(float v) {      // sum over the LPR (32 / 64) lanes of a row
  v = row16_sum(v);
  const float lo = lane_f32(v, 0) + lane_f32(v, 16), hi = lane_f32(v, 32) + lane_f32(v, 48);
  if constexpr (LPR == 64) return lo + hi;
  else return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ void unpack8(u32x4 w, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = bf16lo(w[e]); f[2 * e + 1] = bf16hi(w[e]);
#ifdef DL_UNPACK_NOP   // (round 6 hazard experiment, tools/_run7.sh: a wait state between the unpacking shift and whatever consumes it)
    asm volatile("s_nop 1" : "+v"(f[2 * e]));
#endif
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}
constexpr int LN16_PASSES = 4;

template <int D>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t ldy,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t M, float eps) {
  constexpr int LPR = D / 8, RPP = 64 / LPR, P = LN16_PASSES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, cl = lane % LPR;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (RPP * P) + sub;
  u32x4 w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int64_t row = row0 + q * RPP;
    w[q] = u32x4{0u, 0u, 0u, 0u};
    if (row < M) w[q] = *reinterpret_cast<const u32x4*>(x + row * ldx + cl * 8);
  }
  float gm[8], bt[8];
  { const f32x4 a = *reinterpret_cast<const f32x4*>(gamma + cl * 8), b = *reinterpret_cast<const f32x4*>(gamma + cl * 8 + 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(beta + cl * 8), d = *reinterpret_cast<const f32x4*>(beta + cl * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gm[e] = a[e]; gm[4 + e] = b[e]; bt[e] = c[e]; bt[4 + e] = d[e]; } }
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int64_t row = row0 + q * RPP;
    float v[8];
    unpack8(w[q], v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    const float mean = seg_sum<LPR>(s) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; sq += d * d; }
    const float rstd = rsqrtf(seg_sum<LPR>(sq) * (1.0f / D) + eps);
    if (row < M) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
      store16_fam<1>(y + row * ldy + cl * 8, pack8(o));
      if (cl == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
    }
  }
}

// partial layout as ln_bwd_kernel: [M / LN_BWD_ROWS workgroups][2][D]
template <int D>
__global__ __launch_bounds__(256) void ln_bwd16_kernel(const bf16_t* __restrict__ dy, int64_t lddy, int64_t dy_share,
                                                       const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const bf16_t* __restrict__ dres, int64_t lddres, bf16_t* __restrict__ dx,
                                                       int64_t lddx, float* __restrict__ partial, int64_t M) {
  constexpr int LPR = D / 8, RPP = 64 / LPR, P = LN16_PASSES, NIT = LN_BWD_ROWS / (4 * RPP * P);
  static_assert(NIT >= 1 && NIT * 4 * RPP * P == LN_BWD_ROWS, "a workgroup covers LN_BWD_ROWS rows");
  __shared__ __attribute__((aligned(16))) float sm[4 * RPP * 2 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, cl = lane % LPR;
  float gm[8], dg[8], db[8];
  { const f32x4 a = *reinterpret_cast<const f32x4*>(gamma + cl * 8), b = *reinterpret_cast<const f32x4*>(gamma + cl * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gm[e] = a[e]; gm[4 + e] = b[e]; } }
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
#pragma unroll 1
  for (int it = 0; it < NIT; ++it) {
    const int64_t row0 = (int64_t)blockIdx.x * LN_BWD_ROWS + (it * 4 + wave) * (RPP * P) + sub;
    u32x4 wx[P], wd[P], wr[P];
    float mu[P], rs[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int64_t row = row0 + q * RPP;
      wx[q] = wd[q] = wr[q] = u32x4{0u, 0u, 0u, 0u};
      mu[q] = 0.f; rs[q] = 0.f;
      if (row < M) {
        const int64_t dr = dy_share == 1 ? row : row / dy_share;
        wx[q] = (DL_NT_MASK & 16) ? load16_nt(x + row * ldx + cl * 8) : *reinterpret_cast<const u32x4*>(x + row * ldx + cl * 8);   // (saved activation: last use)
        wd[q] = *reinterpret_cast<const u32x4*>(dy + dr * lddy + cl * 8);
        if (dres) wr[q] = *reinterpret_cast<const u32x4*>(dres + row * lddres + cl * 8);
        mu[q] = mean[row]; rs[q] = rstd[row];
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int64_t row = row0 + q * RPP;
      float xv[8], dv[8], g[8];
      unpack8(wx[q], xv);
      unpack8(wd[q], dv);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[e] = (xv[e] - mu[q]) * rs[q];                       // xhat (rows past the end: x = dy = 0, rs = 0 -> no contribution)
        g[e] = dv[e] * gm[e];
        c1 += g[e];
        c2 += g[e] * xv[e];
        dg[e] += dv[e] * xv[e];
        db[e] += dv[e];
      }
      c1 = seg_sum<LPR>(c1) * (1.0f / D);
      c2 = seg_sum<LPR>(c2) * (1.0f / D);
      if (row < M) {
        float rv[8], o[8];
        unpack8(wr[q], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs[q] * (g[e] - c1 - xv[e] * c2) + rv[e];
        store16_fam<1>(dx + row * lddx + cl * 8, pack8(o));
      }
    }
  }
  // dgamma / dbeta: the wave's row groups and the four waves through LDS, then one partial row per workgroup
  float* mine = sm + (size_t)(wave * RPP + sub) * 2 * D;
  *reinterpret_cast<f32x4*>(mine + cl * 8) = f32x4{dg[0], dg[1], dg[2], dg[3]};
  *reinterpret_cast<f32x4*>(mine + cl * 8 + 4) = f32x4{dg[4], dg[5], dg[6], dg[7]};
  *reinterpret_cast<f32x4*>(mine + D + cl * 8) = f32x4{db[0], db[1], db[2], db[3]};
  *reinterpret_cast<f32x4*>(mine + D + cl * 8 + 4) = f32x4{db[4], db[5], db[6], db[7]};
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4 * RPP; ++w) s += sm[(size_t)w * 2 * D + i];
    partial[(int64_t)blockIdx.x * 2 * D + i] = s;
  }
}

}  // namespace

extern "C" int dl_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta,
                                void* y, int64_t ldy, float* mean, float* rstd, int64_t M, int64_t D,
                                float eps, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && y && gamma && beta, DL_ERR_ARG, "dl_layernorm_fwd: null pointer");
  DL_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, DL_ERR_SHAPE,
               "dl_layernorm_fwd: D=%ld must be a multiple of 4 and <= %d", (long)D, 256 * LN_MAXV);
  DL_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, DL_ERR_ALIGN, "dl_layernorm_fwd: ld must be multiple of 4");
  const uint32_t blocks = (uint32_t)((M + 4 * LN_FWD_RPW - 1) / (4 * LN_FWD_RPW));
  dl_prof_before(3, s);
  if (dtype == DL_BF16 && (D == 256 || D == 512) && ldx % 8 == 0 && ldy % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
      (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
    if (D == 256) {
      constexpr int RPB = 4 * 2 * LN16_PASSES;            // rows per workgroup: 4 waves x 2 rows x passes
      const uint32_t nb = (uint32_t)((M + RPB - 1) / RPB);
      hipLaunchKernelGGL((ln_fwd16_kernel<256>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, M, eps);
    } else {
      constexpr int RPB = 4 * 1 * LN16_PASSES;
      const uint32_t nb = (uint32_t)((M + RPB - 1) / RPB);
      hipLaunchKernelGGL((ln_fwd16_kernel<512>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, M, eps);
    }
    DL_CHECK_LAUNCH("dl_layernorm_fwd");
    dl_prof_after(3, s, 8.0 * M * D, 2.0 * M * D * dl_dtype_size(dtype));
    return DL_OK;
  }
#define LN_FWD(TT, NVV) hipLaunchKernelGGL((ln_fwd_kernel<TT, NVV>), dim3(blocks), dim3(256), 0, s, (const TT*)x, ldx, \
                                          gamma, beta, (TT*)y, ldy, mean, rstd, M, (int)D, eps)
  const int nvg = (int)((D + 255) / 256);
  if (dtype == DL_BF16) { if (nvg <= 1) LN_FWD(bf16_t, 1); else if (nvg <= 2) LN_FWD(bf16_t, 2); else if (nvg <= 4) LN_FWD(bf16_t, 4); else LN_FWD(bf16_t, 8); }
  else { if (nvg <= 1) LN_FWD(float, 1); else if (nvg <= 2) LN_FWD(float, 2); else if (nvg <= 4) LN_FWD(float, 4); else LN_FWD(float, 8); }
#undef LN_FWD
  DL_CHECK_LAUNCH("dl_layernorm_fwd");
  dl_prof_after(3, s, 8.0 * M * D, 2.0 * M * D * dl_dtype_size(dtype));
  return DL_OK;
}

extern "C" size_t dl_layernorm_bwd_workspace_bytes(int64_t M, int64_t D) {
  const int64_t nb = (M + LN_BWD_ROWS - 1) / LN_BWD_ROWS;
  return (size_t)nb * 2 * (size_t)D * sizeof(float);
}

extern "C" int dl_layernorm_bwd(const void* dy, int64_t lddy, int64_t dy_share, const void* x, int64_t ldx,
                                const float* mean, const float* rstd, const float* gamma,
                                const void* dres, int64_t lddres, void* dx, int64_t lddx, float* dgamma,
                                float* dbeta, int32_t accumulate, int64_t M, int64_t D, int32_t dtype,
                                void* workspace, size_t workspace_bytes, dl_reduce_item* deferred, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  if (deferred) deferred->kind = DL_REDUCE_NONE;
  DL_CHECK_ARG(dy && x && mean && rstd && gamma && dx, DL_ERR_ARG, "dl_layernorm_bwd: null pointer");
  DL_CHECK_ARG(dy_share >= 1, DL_ERR_ARG, "dl_layernorm_bwd: dy_share=%ld must be >= 1", (long)dy_share);
  DL_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, DL_ERR_SHAPE,
               "dl_layernorm_bwd: D=%ld must be a multiple of 4 and <= %d", (long)D, 256 * LN_MAXV);
  DL_CHECK_ARG(workspace && workspace_bytes >= dl_layernorm_bwd_workspace_bytes(M, D),
               DL_ERR_WORKSPACE, "dl_layernorm_bwd: workspace too small");
  const int nb = (int)((M + LN_BWD_ROWS - 1) / LN_BWD_ROWS);
  const size_t smem = 4 * 2 * (size_t)D * sizeof(float);
#define LN_BWD(TT, NVV) hipLaunchKernelGGL((ln_bwd_kernel<TT, NVV>), dim3(nb), dim3(256), smem, s, (const TT*)dy, lddy, dy_share, \
                                          (const TT*)x, ldx, mean, rstd, gamma, (const TT*)dres, lddres, (TT*)dx, lddx, \
                                          (float*)workspace, M, (int)D)
  const int nvg = (int)((D + 255) / 256);
  const bool v16 = dtype == DL_BF16 && (D == 256 || D == 512) && lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 &&
                   (!dres || lddres % 8 == 0) && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)gamma) & 15) == 0;
  if (v16 && D == 256)
    hipLaunchKernelGGL((ln_bwd16_kernel<256>), dim3(nb), dim3(256), 0, s, (const bf16_t*)dy, lddy, dy_share, (const bf16_t*)x, ldx, mean, rstd,
                       gamma, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, (float*)workspace, M);
  else if (v16)
    hipLaunchKernelGGL((ln_bwd16_kernel<512>), dim3(nb), dim3(256), 0, s, (const bf16_t*)dy, lddy, dy_share, (const bf16_t*)x, ldx, mean, rstd,
                       gamma, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, (float*)workspace, M);
  else
  if (dtype == DL_BF16) { if (nvg <= 1) LN_BWD(bf16_t, 1); else if (nvg <= 2) LN_BWD(bf16_t, 2); else if (nvg <= 4) LN_BWD(bf16_t, 4); else LN_BWD(bf16_t, 8); }
  else { if (nvg <= 1) LN_BWD(float, 1); else if (nvg <= 2) LN_BWD(float, 2); else if (nvg <= 4) LN_BWD(float, 4); else LN_BWD(float, 8); }
#undef LN_BWD
  DL_CHECK_LAUNCH("dl_layernorm_bwd");
  if (dgamma && dbeta == dgamma + D && deferred) {
    dl_reduce_item& it = *deferred;
    it.kind = DL_REDUCE_PARTIALS; it.out_dtype = DL_F32; it.src = (const float*)workspace; it.out = dgamma;
    it.mn = 2 * D; it.ldc = 0; it.N = (int32_t)(2 * D); it.splits = nb; it.accumulate = accumulate; it.M = 0;
    it.cs_slabs = nullptr; it.cs_out = nullptr;
    return DL_OK;
  }
  if (dgamma && dbeta == dgamma + D) {
    // adjacent outputs ([2][D]): the partials' [2][D] rows reduce in one launch
    hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((2 * D + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s,
                       (const float*)workspace, nb, (int64_t)(2 * D), (int)(2 * D), dgamma, accumulate);
  } else {
    if (dgamma)
      hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((D + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s,
                         (const float*)workspace, nb, (int64_t)(2 * D), (int)D, dgamma, accumulate);
    if (dbeta)
      hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((D + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s,
                         (const float*)workspace + D, nb, (int64_t)(2 * D), (int)D, dbeta, accumulate);
  }
  DL_CHECK_LAUNCH("dl_layernorm_bwd(final)");
  return DL_OK;
}

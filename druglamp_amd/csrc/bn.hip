// bn.hip — channel-last BatchNorm pieces for the ProteinCNN path (HBM-bound column statistics +
// elementwise apply), with a row-validity window so that zero-padding halos stay zero.
#include "common.cuh"

namespace {
static inline int bn_rows_per_block(int64_t R, int64_t C) {
  const int64_t colgroups = (C + 255) / 256;
  int64_t chunks = (1024 + colgroups - 1) / colgroups;
  int64_t rows = (R + chunks - 1) / chunks;
  if (rows < 32) rows = 32;
  return (int)((rows + 3) / 4 * 4);
}

// 32-bit on purpose: a 64-bit modulo costs ~100 instructions and this runs once per row per thread (R < 2^31 is checked)
__device__ __forceinline__ bool row_valid(int64_t r, int64_t win, int64_t halo, int64_t valid) {
  if (win == 0) return true;
  const uint32_t q = (uint32_t)r % (uint32_t)win;
  return q >= (uint32_t)halo && q < (uint32_t)(halo + valid);
}

// Row rule of every kernel below: rw == nullptr -> the window rule above (weight 1 for valid rows); else rw[r] is the row's
// weight: < 0 halo row (excluded, written as zeros), 0 context row (computed, not part of the statistics), m >= 1 a row that
// stands for m identical rows (the ProteinCNN compact layout, druglamp_amd/protein_plan.py).  -1 = excluded.
__device__ __forceinline__ float row_weight(const float* __restrict__ rw, int64_t r, int64_t win, int64_t halo, int64_t valid) {
  if (rw) return rw[r];
  return row_valid(r, win, halo, valid) ? 1.f : -1.f;
}

// grid (ceil(C/256), chunks); lane -> 4 columns; MODE 0: (w y, w y^2); MODE 1: (dz, dz*yhat) over the rows with w >= 0
// prelu_g / prelu_b (MODE 1 only, round 5): gamma / beta of a BatchNorm whose output went through a ReLU (Linear -> BN -> ReLU of
// the SimSiam MLPs): the incoming gradient counts only where z = (y - mean) rstd gamma + beta > 0.
template <typename T, int MODE>
__global__ void bn_partial_kernel(const T* __restrict__ a, const T* __restrict__ y, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, int64_t R, int C, int64_t win, int64_t halo,
                                  int64_t valid, const float* __restrict__ rw, float* __restrict__ partial, int rows_per_block,
                                  const float* __restrict__ prelu_g = nullptr, const float* __restrict__ prelu_b = nullptr) {
  __shared__ f32x4 red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(R, r0 + rows_per_block);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f}, pg = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) { mu = *reinterpret_cast<const f32x4*>(mean + c); rs = *reinterpret_cast<const f32x4*>(rstd + c); }
    const bool prelu = MODE == 1 && prelu_g != nullptr;
    if (prelu) { pg = *reinterpret_cast<const f32x4*>(prelu_g + c); pb = *reinterpret_cast<const f32x4*>(prelu_b + c); }
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      const float wr = row_weight(rw, r, win, halo, valid);
      if (MODE == 0 ? wr <= 0.f : wr < 0.f) continue;
      f32x4 v = load4<T>(a + r * C + c);
      if (MODE == 0) { const f32x4 t = v * wr; s0 += t; s1 += t * v; }      // (wr = 1: the plain sums, bit for bit)
      else {
        const f32x4 yv = load4<T>(y + r * C + c);
        const f32x4 yh = (yv - mu) * rs;
        if (prelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (yh[e] * pg[e] + pb[e] > 0.f) ? v[e] : 0.f;
        }
        s0 += v; s1 += v * yh;
      }
    }
  }
  red[0][wave][lane] = s0; red[1][wave][lane] = s1;
  __syncthreads();
  if (wave == 0 && c < C) {
    s0 = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    s1 = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    float* d0 = partial + ((int64_t)blockIdx.y * 2 + 0) * C + c;
    float* d1 = partial + ((int64_t)blockIdx.y * 2 + 1) * C + c;
    *reinterpret_cast<f32x4*>(d0) = s0; *reinterpret_cast<f32x4*>(d1) = s1;
  }
}

// bf16, C in {64, 128, 256}: 16-byte loads, every lane busy (256 / (C/8) rows per pass), four passes in flight.
// Same partial layout as above.  grid (1, chunks).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_wide_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int64_t R, int C, int64_t win, int64_t halo, int64_t valid,
                                                              const float* __restrict__ rw, float* __restrict__ partial, int rows_per_block) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int cpr = C >> 3, rpp = 256 / cpr;                  // chunks per row, rows per pass
  const int ch = tid % cpr, rsub = tid / cpr, c = ch * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(R, r0 + rows_per_block);
  float s0[8], s1[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = MODE == 1 ? mean[c + e] : 0.f; rs[e] = MODE == 1 ? rstd[c + e] : 1.f; }
  auto unpack = [](u32x4 w, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(w[i]); f[2 * i + 1] = bf16hi(w[i]); }
  };
  for (int64_t r = r0 + rsub; r < r1; r += 4 * rpp) {
    u32x4 va[4], vy[4];
    bool ok[4];
    float wr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = r + (int64_t)u * rpp;
      wr[u] = rr < r1 ? row_weight(rw, rr, win, halo, valid) : -1.f;
      ok[u] = MODE == 0 ? wr[u] > 0.f : wr[u] >= 0.f;
      va[u] = u32x4{0u, 0u, 0u, 0u}; vy[u] = va[u];
      if (ok[u]) {
        va[u] = *reinterpret_cast<const u32x4*>(a + rr * C + c);
        if (MODE == 1) vy[u] = *reinterpret_cast<const u32x4*>(y + rr * C + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      float v[8], w[8];
      unpack(va[u], v);
      if (MODE == 1) unpack(vy[u], w);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (MODE == 0) { const float t = v[e] * wr[u]; s0[e] += t; s1[e] += t * v[e]; }     // (wr = 1: the plain sums, bit for bit)
        else { s0[e] += v[e]; s1[e] += v[e] * ((w[e] - mu[e]) * rs[e]); }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
  __syncthreads();
  // thread t < 2*C sums column (t % C) of array (t / C) over the rpp row groups
  for (int t = tid; t < 2 * C; t += 256) {
    const int which = t / C, col = t % C, cch = col >> 3, e = col & 7;
    float acc = 0.f;
    for (int g = 0; g < rpp; ++g) acc += red[which][g * cpr + cch][e];
    partial[((int64_t)blockIdx.y * 2 + which) * C + col] = acc;
  }
}

template <typename T, bool RELU = false>
__global__ void bn_apply_fwd_kernel(const T* __restrict__ y, T* __restrict__ z, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int64_t R, int C, int64_t win, int64_t halo,
                                    int64_t valid, const float* __restrict__ rw) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;          // R * C / 4 < 2^32 is checked by the host
  const uint32_t c4 = (uint32_t)C >> 2;
  if (i >= (uint32_t)R * c4) return;
  const int64_t r = i / c4;
  const int c = (int)(i - (uint32_t)r * c4) * 4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (row_weight(rw, r, win, halo, valid) >= 0.f) {
    const f32x4 v = load4<T>(y + r * C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
    o = (v - mu) * rs * g + b;
    if constexpr (RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
  }
  store4<T>(z + r * C + c, o);
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums, float inv_n, int relu_mask, T* __restrict__ dy,
                                    int64_t R, int C, int64_t win, int64_t halo, int64_t valid, const float* __restrict__ rw,
                                    const float* __restrict__ prelu_b = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;          // R * C / 4 < 2^32 is checked by the host
  const uint32_t c4 = (uint32_t)C >> 2;
  if (i >= (uint32_t)R * c4) return;
  const int64_t r = i / c4;
  const int c = (int)(i - (uint32_t)r * c4) * 4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  const float wr = row_weight(rw, r, win, halo, valid);
  if (wr >= 0.f) {
    f32x4 d = load4<T>(dz + r * C + c);
    const f32x4 yv = load4<T>(y + r * C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c), s1 = *reinterpret_cast<const f32x4*>(sums + C + c);
    const f32x4 yh = (yv - mu) * rs;
    if (prelu_b) {                                   // BatchNorm -> ReLU: the incoming gradient counts where the ReLU was open
      const f32x4 pb = *reinterpret_cast<const f32x4*>(prelu_b + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = (yh[e] * g[e] + pb[e] > 0.f) ? d[e] : 0.f;
    }
    // a row that stands for wr identical rows takes the mean terms wr times (its dz is the sum over those rows)
    if (rw) o = g * rs * (d - (s0 * inv_n + yh * (s1 * inv_n)) * wr);
    else o = g * rs * (d - s0 * inv_n - yh * (s1 * inv_n));
    if (relu_mask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? o[e] : 0.f;
    }
  }
  store4<T>(dy + r * C + c, o);
}

// ---- wide forms of the two apply passes (bf16, C in {64, 128, 256}) -------------------------------------------------
// A thread keeps ONE 8-column chunk (16 bytes) for the whole launch, so the per-column parameters are loaded and folded
// once (z = y * a + b with a = rstd * gamma, b = beta - mean * a); it walks rows with a grid stride, BN_WIDE_U rows in
// flight per iteration.  The one-quad-per-thread forms above spend more instructions on index arithmetic and the halo
// test (a 32-bit modulo per 8 bytes) than on the data: 3.2 / 4.0 TB/s; these: see tools/ln_bench.py.
constexpr int BN_WIDE_U = 4;
__device__ __forceinline__ void unpack8(u32x4 w, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(w[i]); f[2 * i + 1] = bf16hi(w[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}

template <int C>
__global__ __launch_bounds__(256) void bn_apply_fwd_wide_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int R, int win, int halo, int valid, const float* __restrict__ rw) {
  constexpr int TPR = C / 8, RPB = 256 / TPR;              // threads per row, rows per block and pass
  const int c = (threadIdx.x % TPR) * 8, rsub = threadIdx.x / TPR;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = rstd[c + e] * gamma[c + e]; b[e] = beta[c + e] - mean[c + e] * a[e]; }
  const int stride = gridDim.x * RPB;
  for (int r0 = blockIdx.x * RPB + rsub; r0 < R; r0 += stride * BN_WIDE_U) {
    u32x4 w[BN_WIDE_U];
    bool ok[BN_WIDE_U];
#pragma unroll
    for (int u = 0; u < BN_WIDE_U; ++u) {
      const int r = r0 + u * stride;
      ok[u] = r < R && row_weight(rw, r, win, halo, valid) >= 0.f;
      w[u] = u32x4{0u, 0u, 0u, 0u};
      if (ok[u]) w[u] = *reinterpret_cast<const u32x4*>(y + (int64_t)r * C + c);
    }
#pragma unroll
    for (int u = 0; u < BN_WIDE_U; ++u) {
      const int r = r0 + u * stride;
      if (r >= R) break;
      u32x4 o = {0u, 0u, 0u, 0u};                         // halo rows are written as zeros (the next conv's padding)
      if (ok[u]) {
        float f[8];
        unpack8(w[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], a[e], b[e]);
        o = pack8(f);
      }
      store16_fam<2>(z + (int64_t)r * C + c, o);
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void bn_bwd_apply_wide_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sums,
                                                                float inv_n, int relu_mask, bf16_t* __restrict__ dy, int R, int win,
                                                                int halo, int valid, const float* __restrict__ rw) {
  constexpr int TPR = C / 8, RPB = 256 / TPR;
  const int c = (threadIdx.x % TPR) * 8, rsub = threadIdx.x / TPR;
  // dy = g*rs*(d - s0/n - yhat*s1/n), yhat = (y - mu)*rs   ->   dy = k0*d + k1*y + k2 per column
  float mu[8], rs[8], gr[8], m0[8], m1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[c + e]; rs[e] = rstd[c + e]; gr[e] = gamma[c + e] * rs[e];
    m0[e] = sums[c + e] * inv_n; m1[e] = sums[C + c + e] * inv_n;
  }
  const int stride = gridDim.x * RPB;
  for (int r0 = blockIdx.x * RPB + rsub; r0 < R; r0 += stride * BN_WIDE_U) {
    u32x4 wd[BN_WIDE_U], wy[BN_WIDE_U];
    bool ok[BN_WIDE_U];
    float wr[BN_WIDE_U];
#pragma unroll
    for (int u = 0; u < BN_WIDE_U; ++u) {
      const int r = r0 + u * stride;
      wr[u] = r < R ? row_weight(rw, r, win, halo, valid) : -1.f;
      ok[u] = wr[u] >= 0.f;
      wd[u] = u32x4{0u, 0u, 0u, 0u}; wy[u] = wd[u];
      if (ok[u]) {
        wd[u] = *reinterpret_cast<const u32x4*>(dz + (int64_t)r * C + c);
        wy[u] = *reinterpret_cast<const u32x4*>(y + (int64_t)r * C + c);
      }
    }
#pragma unroll
    for (int u = 0; u < BN_WIDE_U; ++u) {
      const int r = r0 + u * stride;
      if (r >= R) break;
      u32x4 o = {0u, 0u, 0u, 0u};
      if (ok[u]) {
        float d[8], yv[8];
        unpack8(wd[u], d);
        unpack8(wy[u], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yh = (yv[e] - mu[e]) * rs[e];                  // same operation order as bn_bwd_apply_kernel
          float v = rw ? gr[e] * (d[e] - (m0[e] + yh * m1[e]) * wr[u]) : gr[e] * (d[e] - m0[e] - yh * m1[e]);
          if (relu_mask) v = yv[e] > 0.f ? v : 0.f;
          d[e] = v;
        }
        o = pack8(d);
      }
      store16_fam<2>(dy + (int64_t)r * C + c, o);
    }
  }
}

static inline uint32_t bn_wide_blocks(int64_t R, int C) {
  const int rpb = 256 / (C / 8);
  int64_t b = (R + (int64_t)rpb * BN_WIDE_U - 1) / ((int64_t)rpb * BN_WIDE_U);
  return (uint32_t)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
}  // namespace

extern "C" size_t dl_bn_workspace_bytes(int64_t R, int64_t C) {
  const int rpb = bn_rows_per_block(R, C);
  const int64_t chunks = (R + rpb - 1) / rpb;
  return (size_t)chunks * 2 * (size_t)C * sizeof(float);
}

template <int MODE>
static int bn_reduce(const char* who, const void* a, const void* y, const float* mean, const float* rstd, int64_t R,
                     int64_t C, int64_t win, int64_t halo, int64_t valid, const float* rw, int32_t dtype, float* sums, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  DL_CHECK_ARG(a && sums && R > 0 && C > 0 && C % 4 == 0, DL_ERR_ARG, "%s: bad args", who);
  DL_CHECK_ARG(R < (1ll << 31), DL_ERR_SHAPE, "%s: R must fit 31 bits", who);
  DL_CHECK_ARG(ws && ws_bytes >= dl_bn_workspace_bytes(R, C), DL_ERR_WORKSPACE, "%s: workspace too small", who);
  const int rpb = bn_rows_per_block(R, C);
  const int chunks = (int)((R + rpb - 1) / rpb);
  dim3 grid((uint32_t)((C + 255) / 256), (uint32_t)chunks);
  if (dtype == DL_BF16 && (C == 64 || C == 128 || C == 256) && ((uintptr_t)a & 15) == 0 && (!y || ((uintptr_t)y & 15) == 0))
    hipLaunchKernelGGL((bn_partial_wide_kernel<MODE>), dim3(1, (uint32_t)chunks), dim3(256), 0, s, (const bf16_t*)a,
                       (const bf16_t*)y, mean, rstd, R, (int)C, win, halo, valid, rw, (float*)ws, rpb);
  else if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_partial_kernel<bf16_t, MODE>), grid, dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)y,
                       mean, rstd, R, (int)C, win, halo, valid, rw, (float*)ws, rpb);
  else
    hipLaunchKernelGGL((bn_partial_kernel<float, MODE>), grid, dim3(256), 0, s, (const float*)a, (const float*)y, mean,
                       rstd, R, (int)C, win, halo, valid, rw, (float*)ws, rpb);
  hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((2 * C + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s, (const float*)ws,
                     chunks, (int64_t)(2 * C), (int)(2 * C), sums, 0);
  DL_CHECK_LAUNCH(who);
  return DL_OK;
}

extern "C" int dl_bn_stats(const void* y, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid, int32_t dtype,
                           float* sums, void* workspace, size_t workspace_bytes, dl_stream stream) {
  return bn_reduce<0>("dl_bn_stats", y, nullptr, nullptr, nullptr, R, C, win, halo, valid, nullptr, dtype, sums, workspace,
                      workspace_bytes, (hipStream_t)stream);
}

extern "C" int dl_bn_stats_rw(const void* y, int64_t R, int64_t C, const float* row_w, int32_t dtype, float* sums,
                              void* workspace, size_t workspace_bytes, dl_stream stream) {
  DL_CHECK_ARG(row_w, DL_ERR_ARG, "dl_bn_stats_rw: null row weights");
  return bn_reduce<0>("dl_bn_stats_rw", y, nullptr, nullptr, nullptr, R, C, 0, 0, 0, row_w, dtype, sums, workspace,
                      workspace_bytes, (hipStream_t)stream);
}

extern "C" int dl_bn_bwd_reduce(const void* dz, const void* y, const float* mean, const float* rstd, int64_t R, int64_t C,
                                int64_t win, int64_t halo, int64_t valid, int32_t dtype, float* sums, void* workspace,
                                size_t workspace_bytes, dl_stream stream) {
  DL_CHECK_ARG(y && mean && rstd, DL_ERR_ARG, "dl_bn_bwd_reduce: null pointer");
  return bn_reduce<1>("dl_bn_bwd_reduce", dz, y, mean, rstd, R, C, win, halo, valid, nullptr, dtype, sums, workspace,
                      workspace_bytes, (hipStream_t)stream);
}

extern "C" int dl_bn_bwd_reduce_rw(const void* dz, const void* y, const float* mean, const float* rstd, int64_t R, int64_t C,
                                   const float* row_w, int32_t dtype, float* sums, void* workspace, size_t workspace_bytes,
                                   dl_stream stream) {
  DL_CHECK_ARG(y && mean && rstd && row_w, DL_ERR_ARG, "dl_bn_bwd_reduce_rw: null pointer");
  return bn_reduce<1>("dl_bn_bwd_reduce_rw", dz, y, mean, rstd, R, C, 0, 0, 0, row_w, dtype, sums, workspace,
                      workspace_bytes, (hipStream_t)stream);
}

static int bn_apply_fwd_run(const void* y, void* z, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid,
                            const float* rw, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(R < (1ll << 31) && R * (C / 4) < (1ll << 32), DL_ERR_SHAPE, "dl_bn_apply_fwd: R * C / 4 must fit 32 bits");
  DL_CHECK_ARG(y && z && mean && rstd && gamma && beta && R > 0 && C > 0 && C % 4 == 0, DL_ERR_ARG,
               "dl_bn_apply_fwd: bad args");
  const int64_t n = R * (C / 4);
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  const bool wide = dtype == DL_BF16 && (C == 64 || C == 128 || C == 256) && (((uintptr_t)y | (uintptr_t)z) & 15) == 0;
  if (wide) {
    const uint32_t wb = bn_wide_blocks(R, (int)C);
    if (C == 64) hipLaunchKernelGGL((bn_apply_fwd_wide_kernel<64>), dim3(wb), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)z, mean, rstd, gamma, beta, (int)R, (int)win, (int)halo, (int)valid, rw);
    else if (C == 128) hipLaunchKernelGGL((bn_apply_fwd_wide_kernel<128>), dim3(wb), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)z, mean, rstd, gamma, beta, (int)R, (int)win, (int)halo, (int)valid, rw);
    else hipLaunchKernelGGL((bn_apply_fwd_wide_kernel<256>), dim3(wb), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)z, mean, rstd, gamma, beta, (int)R, (int)win, (int)halo, (int)valid, rw);
  } else if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_apply_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)z, mean,
                       rstd, gamma, beta, R, (int)C, win, halo, valid, rw);
  else
    hipLaunchKernelGGL((bn_apply_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)y, (float*)z, mean,
                       rstd, gamma, beta, R, (int)C, win, halo, valid, rw);
  DL_CHECK_LAUNCH("dl_bn_apply_fwd");
  return DL_OK;
}

extern "C" int dl_bn_apply_fwd(const void* y, void* z, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid,
                               int32_t dtype, dl_stream stream) {
  return bn_apply_fwd_run(y, z, mean, rstd, gamma, beta, R, C, win, halo, valid, nullptr, dtype, stream);
}

extern "C" int dl_bn_apply_fwd_rw(const void* y, void* z, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, int64_t R, int64_t C, const float* row_w, int32_t dtype, dl_stream stream) {
  DL_CHECK_ARG(row_w, DL_ERR_ARG, "dl_bn_apply_fwd_rw: null row weights");
  return bn_apply_fwd_run(y, z, mean, rstd, gamma, beta, R, C, 0, 0, 0, row_w, dtype, stream);
}

static int bn_bwd_apply_run(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma,
                            const float* sums, float inv_n, int32_t relu_mask, void* dy, int64_t R, int64_t C,
                            int64_t win, int64_t halo, int64_t valid, const float* rw, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(R < (1ll << 31) && R * (C / 4) < (1ll << 32), DL_ERR_SHAPE, "dl_bn_bwd_apply: R * C / 4 must fit 32 bits");
  DL_CHECK_ARG(dz && y && mean && rstd && gamma && sums && dy && R > 0 && C > 0 && C % 4 == 0, DL_ERR_ARG,
               "dl_bn_bwd_apply: bad args");
  const int64_t n = R * (C / 4);
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  const bool wide = dtype == DL_BF16 && (C == 64 || C == 128 || C == 256) && (((uintptr_t)y | (uintptr_t)dz | (uintptr_t)dy) & 15) == 0;
  if (wide) {
    const uint32_t wb = bn_wide_blocks(R, (int)C);
    if (C == 64) hipLaunchKernelGGL((bn_bwd_apply_wide_kernel<64>), dim3(wb), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, mean, rstd, gamma, sums, inv_n, relu_mask, (bf16_t*)dy, (int)R, (int)win, (int)halo, (int)valid, rw);
    else if (C == 128) hipLaunchKernelGGL((bn_bwd_apply_wide_kernel<128>), dim3(wb), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, mean, rstd, gamma, sums, inv_n, relu_mask, (bf16_t*)dy, (int)R, (int)win, (int)halo, (int)valid, rw);
    else hipLaunchKernelGGL((bn_bwd_apply_wide_kernel<256>), dim3(wb), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, mean, rstd, gamma, sums, inv_n, relu_mask, (bf16_t*)dy, (int)R, (int)win, (int)halo, (int)valid, rw);
  } else if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y,
                       mean, rstd, gamma, sums, inv_n, relu_mask, (bf16_t*)dy, R, (int)C, win, halo, valid, rw);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)dz, (const float*)y,
                       mean, rstd, gamma, sums, inv_n, relu_mask, (float*)dy, R, (int)C, win, halo, valid, rw);
  DL_CHECK_LAUNCH("dl_bn_bwd_apply");
  return DL_OK;
}

extern "C" int dl_bn_bwd_apply(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma,
                               const float* sums, float inv_n, int32_t relu_mask, void* dy, int64_t R, int64_t C,
                               int64_t win, int64_t halo, int64_t valid, int32_t dtype, dl_stream stream) {
  return bn_bwd_apply_run(dz, y, mean, rstd, gamma, sums, inv_n, relu_mask, dy, R, C, win, halo, valid, nullptr, dtype, stream);
}

extern "C" int dl_bn_bwd_apply_rw(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma,
                                  const float* sums, float inv_n, int32_t relu_mask, void* dy, int64_t R, int64_t C,
                                  const float* row_w, int32_t dtype, dl_stream stream) {
  DL_CHECK_ARG(row_w, DL_ERR_ARG, "dl_bn_bwd_apply_rw: null row weights");
  return bn_bwd_apply_run(dz, y, mean, rstd, gamma, sums, inv_n, relu_mask, dy, R, C, 0, 0, 0, row_w, dtype, stream);
}

// ---- weighted tail rows (MolecularGCN's compact padding form) ------------------------------------------------------
// Inside every window of `win` rows the rows [lead, win) stand for w identical rows each.  dl_bn_bwd_apply computed their
// input gradient with the mean terms once; they count w times: dy[r] -= (w - 1) gamma rstd (S1 / n + xhat[r] S2 / n).
namespace {
template <typename T>
__global__ void bn_tail_fix_kernel(T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ sums,
                                   float inv_n, float wm1, int64_t nwin, int win, int lead, int C) {
  const int tail = win - lead;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (window, tail row, 4 columns)
  const int c4 = C / 4;
  if (i >= nwin * tail * c4) return;
  const int c = (int)(i % c4) * 4;
  const int64_t t = i / c4;
  const int64_t r = (t / tail) * win + lead + (t % tail);
  f32x4 d = load4<T>(dy + r * C + c);
  const f32x4 x = load4<T>(y + r * C + c);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float rs = rstd[c + e];
    const float xhat = (x[e] - mean[c + e]) * rs;
    d[e] -= wm1 * gamma[c + e] * rs * (sums[c + e] * inv_n + xhat * sums[C + c + e] * inv_n);
  }
  store4<T>(dy + r * C + c, d);
}
}  // namespace

extern "C" int dl_bn_tail_fix(void* dy, const void* y, const float* mean, const float* rstd, const float* gamma,
                              const float* sums, float inv_n, int32_t w, int64_t R, int64_t C, int64_t win, int64_t lead,
                              int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dy && y && mean && rstd && gamma && sums && R > 0 && C > 0 && C % 4 == 0 && win > 0 && lead >= 0 && lead < win &&
                   R % win == 0 && w >= 1, DL_ERR_ARG, "dl_bn_tail_fix: bad args");
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_bn_tail_fix: bad dtype");
  const int64_t n = (R / win) * (win - lead) * (C / 4);
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_tail_fix_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (bf16_t*)dy, (const bf16_t*)y, mean, rstd, gamma,
                       sums, inv_n, (float)(w - 1), R / win, (int)win, (int)lead, (int)C);
  else
    hipLaunchKernelGGL((bn_tail_fix_kernel<float>), dim3(blocks), dim3(256), 0, s, (float*)dy, (const float*)y, mean, rstd, gamma,
                       sums, inv_n, (float)(w - 1), R / win, (int)win, (int)lead, (int)C);
  DL_CHECK_LAUNCH("dl_bn_tail_fix");
  return DL_OK;
}

// ---- finalize: sums -> mean / biased var / rstd, and the running-statistics update, in one tiny launch ---------
namespace {
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float inv_n, float unbias, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ var, float* __restrict__ rstd,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = sums[c] * inv_n;
  const float v = fmaxf(sums[C + c] * inv_n - m * m, 0.f);
  mean[c] = m;
  var[c] = v;
  rstd[c] = rsqrtf(v + eps);
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (v * unbias);
}
}  // namespace

extern "C" int dl_bn_finalize(const float* sums, int64_t n, float eps, float momentum, float* mean, float* var,
                              float* rstd, float* running_mean, float* running_var, int64_t C, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(sums && mean && var && rstd && n > 0 && C > 0, DL_ERR_ARG, "dl_bn_finalize: bad args");
  const float unbias = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((uint32_t)((C + 127) / 128)), dim3(128), 0, s, sums, (float)(1.0 / (double)n),
                     unbias, eps, momentum, mean, var, rstd, running_mean, running_var, (int)C);
  DL_CHECK_LAUNCH("dl_bn_finalize");
  return DL_OK;
}

// ---- batch statistics in TWO launches instead of three: the second stage of the column reduction and the finalize step in
// one kernel.  Workgroup b owns channels 8b .. 8b+7: columns c (sum) and C + c (sum of squares) of the partials, reduced with
// EXACTLY the arithmetic of dl_reduce_partials_kernel (row-lane k sums chunks k, k + 64, ... four at a time, then the fixed
// LDS tree), so sums / mean / var / rstd are bit-identical to dl_bn_stats + dl_bn_finalize. ---------------------------------
namespace {
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float inv_n,
                                                                   float unbias, float eps, float momentum, float* __restrict__ sums,
                                                                   float* __restrict__ mean, float* __restrict__ var,
                                                                   float* __restrict__ rstd, float* __restrict__ rmean,
                                                                   float* __restrict__ rvar) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, k = threadIdx.x >> 4;
  const int ch = blockIdx.x * 8 + (cl & 7);
  const int c = (cl < 8 ? 0 : C) + ch;                    // column of the [chunks][2C] partials
  const int64_t stride = 2 * (int64_t)C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ch < C) {
    int z = k;
    for (; z + 192 < chunks; z += 256) {
      a0 += partial[(int64_t)z * stride + c];
      a1 += partial[(int64_t)(z + 64) * stride + c];
      a2 += partial[(int64_t)(z + 128) * stride + c];
      a3 += partial[(int64_t)(z + 192) * stride + c];
    }
    for (; z < chunks; z += 64) a0 += partial[(int64_t)z * stride + c];
  }
  red[k][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int half = 32; half > 0; half >>= 1) {
    if (k < half) red[k][cl] += red[k + half][cl];
    __syncthreads();
  }
  if (k == 0 && ch < C) {
    if (sums) sums[c] = red[0][cl];
    if (cl < 8) {
      const float m = red[0][cl] * inv_n;
      const float v = fmaxf(red[0][cl + 8] * inv_n - m * m, 0.f);
      mean[ch] = m;
      var[ch] = v;
      rstd[ch] = rsqrtf(v + eps);
      if (rmean) rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * m;
      if (rvar) rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (v * unbias);
    }
  }
}
}  // namespace

extern "C" int dl_bn_stats_finalize(const void* y, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid, const float* row_w,
                                    int32_t dtype, int64_t n, float eps, float momentum, float* sums, float* mean, float* var,
                                    float* rstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(y && mean && var && rstd && R > 0 && C > 0 && C % 4 == 0 && n > 0, DL_ERR_ARG, "dl_bn_stats_finalize: bad args");
  DL_CHECK_ARG(R < (1ll << 31), DL_ERR_SHAPE, "dl_bn_stats_finalize: R must fit 31 bits");
  DL_CHECK_ARG(ws && ws_bytes >= dl_bn_workspace_bytes(R, C), DL_ERR_WORKSPACE, "dl_bn_stats_finalize: workspace too small");
  const int rpb = bn_rows_per_block(R, C);
  const int chunks = (int)((R + rpb - 1) / rpb);
  dim3 grid((uint32_t)((C + 255) / 256), (uint32_t)chunks);
  if (row_w) { win = 0; halo = 0; valid = 0; }
  if (dtype == DL_BF16 && (C == 64 || C == 128 || C == 256) && ((uintptr_t)y & 15) == 0)
    hipLaunchKernelGGL((bn_partial_wide_kernel<0>), dim3(1, (uint32_t)chunks), dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, R, (int)C, win, halo, valid, row_w, (float*)ws, rpb);
  else if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 0>), grid, dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, R, (int)C, win, halo, valid, row_w, (float*)ws, rpb);
  else if (dtype == DL_F32)
    hipLaunchKernelGGL((bn_partial_kernel<float, 0>), grid, dim3(256), 0, s, (const float*)y, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, R, (int)C, win, halo, valid, row_w, (float*)ws, rpb);
  else { dl_set_error("dl_bn_stats_finalize: bad dtype"); return DL_ERR_ARG; }
  const float unbias = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((uint32_t)((C + 7) / 8)), dim3(1024), 0, s, (const float*)ws, chunks, (int)C,
                     (float)(1.0 / (double)n), unbias, eps, momentum, sums, mean, var, rstd, running_mean, running_var);
  DL_CHECK_LAUNCH("dl_bn_stats_finalize");
  return DL_OK;
}

// ---- BatchNorm followed by ReLU (round 5: the Linear -> BatchNorm1d -> ReLU stages of the SimSiam projector / predictor MLPs,
// model/self_supervised_learning.py:126-166; through round 4 the ReLU and its backward were torch launches over the 131072 x 512
// activations: 0.77 ms of an SSL-epoch step at batch 256).  z = max(0, BN(y)); the backward passes recompute the ReLU's open set
// from y (z > 0 <=> (y - mean) rstd gamma + beta > 0), so nothing extra is saved. ----------------------------------------------
extern "C" int dl_bn_apply_relu_fwd(const void* y, void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    int64_t R, int64_t C, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(R < (1ll << 31) && R * (C / 4) < (1ll << 32), DL_ERR_SHAPE, "dl_bn_apply_relu_fwd: R * C / 4 must fit 32 bits");
  DL_CHECK_ARG(y && z && mean && rstd && gamma && beta && R > 0 && C > 0 && C % 4 == 0, DL_ERR_ARG, "dl_bn_apply_relu_fwd: bad args");
  const uint32_t blocks = (uint32_t)((R * (C / 4) + 255) / 256);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_apply_fwd_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)y, (bf16_t*)z, mean, rstd, gamma,
                       beta, R, (int)C, (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr);
  else if (dtype == DL_F32)
    hipLaunchKernelGGL((bn_apply_fwd_kernel<float, true>), dim3(blocks), dim3(256), 0, s, (const float*)y, (float*)z, mean, rstd, gamma,
                       beta, R, (int)C, (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr);
  else { dl_set_error("dl_bn_apply_relu_fwd: bad dtype"); return DL_ERR_ARG; }
  DL_CHECK_LAUNCH("dl_bn_apply_relu_fwd");
  return DL_OK;
}

extern "C" int dl_bn_relu_bwd(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              float inv_n, void* dy, float* sums, int64_t R, int64_t C, int32_t dtype, void* ws, size_t ws_bytes,
                              dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dz && y && mean && rstd && gamma && beta && dy && sums && R > 0 && C > 0 && C % 4 == 0, DL_ERR_ARG, "dl_bn_relu_bwd: bad args");
  DL_CHECK_ARG(R < (1ll << 31) && R * (C / 4) < (1ll << 32), DL_ERR_SHAPE, "dl_bn_relu_bwd: R * C / 4 must fit 32 bits");
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_bn_relu_bwd: bad dtype");
  DL_CHECK_ARG(ws && ws_bytes >= dl_bn_workspace_bytes(R, C), DL_ERR_WORKSPACE, "dl_bn_relu_bwd: workspace too small");
  const int rpb = bn_rows_per_block(R, C);
  const int chunks = (int)((R + rpb - 1) / rpb);
  dim3 grid((uint32_t)((C + 255) / 256), (uint32_t)chunks);
  const uint32_t blocks = (uint32_t)((R * (C / 4) + 255) / 256);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, mean, rstd, R, (int)C,
                       (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr, (float*)ws, rpb, gamma, beta);
  else
    hipLaunchKernelGGL((bn_partial_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)dz, (const float*)y, mean, rstd, R, (int)C,
                       (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr, (float*)ws, rpb, gamma, beta);
  hipLaunchKernelGGL(dl_reduce_partials_kernel, dim3((uint32_t)((2 * C + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS)), dim3(1024), 0, s, (const float*)ws,
                     chunks, (int64_t)(2 * C), (int)(2 * C), sums, 0);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, mean, rstd, gamma,
                       (const float*)sums, inv_n, 0, (bf16_t*)dy, R, (int)C, (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr, beta);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)dz, (const float*)y, mean, rstd, gamma,
                       (const float*)sums, inv_n, 0, (float*)dy, R, (int)C, (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr, beta);
  DL_CHECK_LAUNCH("dl_bn_relu_bwd");
  return DL_OK;
}

// losses.hip — loss kernels of the SSL / cross-modality heads (fp32 throughout).
//   cos_rowloss   : SimSiam 2 - 2 cos(x, y) per row                (self_supervised_learning.py:184-187)
//   ntxent_stream : NT-Xent over P = [q; k] with a streaming log-sum-exp; the (2n)^2 logit matrix of
//                   self_supervised_learning.py:168-182 is never materialised (fp32 MFMA tiles).
//   triplet_sigcos: margin triplet loss with distance 1 - sigmoid(cos) over a label matrix
//                   (cross_modality.py:15-47, utils.py:571-574).
#include "tiles.cuh"

namespace {
using namespace dltile;

// ------------------------------------------------------------------------------------------
// cosine row loss.  F.normalize semantics: x / max(||x||, 1e-12).
// ------------------------------------------------------------------------------------------
constexpr float NORM_EPS = 1e-12f;

__global__ void cos_rowloss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                       float* __restrict__ row_loss, int64_t n_rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  float dot = 0.f, nx = 0.f, ny = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * D + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * D + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dot += a[e] * b[e]; nx += a[e] * a[e]; ny += b[e] * b[e]; }
  }
  dot = wave_sum(dot); nx = wave_sum(nx); ny = wave_sum(ny);
  if (lane == 0) {
    const float dx = fmaxf(sqrtf(nx), NORM_EPS), dy = fmaxf(sqrtf(ny), NORM_EPS);
    row_loss[row] = 2.0f - 2.0f * dot / (dx * dy);
  }
}

__global__ void vec_sum_kernel(const float* __restrict__ v, int64_t n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    *out = t * scale;
  }
}

// dx = gscale * d/dx (2 - 2 cos)
__global__ void cos_rowloss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, float gscale,
                                       float* __restrict__ dxo, int64_t n_rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  float dot = 0.f, nx = 0.f, ny = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * D + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * D + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dot += a[e] * b[e]; nx += a[e] * a[e]; ny += b[e] * b[e]; }
  }
  dot = wave_sum(dot); nx = wave_sum(nx); ny = wave_sum(ny);
  const float nxs = sqrtf(nx);
  const float dx = fmaxf(nxs, NORM_EPS), dy = fmaxf(sqrtf(ny), NORM_EPS);
  // cos = dot / (dx*dy); d cos / d x = y/(dx*dy) - dot * x / (dx^3 * dy)   (when ||x|| > eps)
  const float a1 = -2.0f * gscale / (dx * dy);
  const float a2 = (nxs > NORM_EPS) ? 2.0f * gscale * dot / (dx * dx * dx * dy) : 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * D + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * D + c);
    *reinterpret_cast<f32x4*>(dxo + row * D + c) = b * a1 + a * a2;
  }
}

// ------------------------------------------------------------------------------------------
// Cross entropy over rows with few classes (round 5): the masked-LM heads of the SSL epochs — F.cross_entropy(logits (N, 27),
// labels, ignore_index = 0), N = batch x 2304 tokens (model/self_supervised_learning.py:93-99).  torch's path is log_softmax +
// nll_loss_forward_reduce, a ONE-workgroup reduction over the 590 k rows of a batch of 256: 0.53 ms forward + 0.36 ms backward
// per head.  Here: one thread per row (a row of a few dozen logits), a fixed-order two-stage sum.
// logits [N][ld] in T (fp32 or bf16; the GEMM in front pads 27 columns to 32), loss and statistics in fp32.
// ------------------------------------------------------------------------------------------
constexpr int CE_MAXC = 4096;
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                           int64_t N, int C, int64_t ignore, float* __restrict__ lse,
                                                           float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (r < N) {
    const T* row = logits + r * ld;            // (a row is <= 256 bytes: the second pass hits L1)
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, to_f32(row[c]));
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += __expf(to_f32(row[c]) - m);
    const float l = m + __logf(sum);
    lse[r] = l;
    const int64_t y = labels[r];
    if (y != ignore && y >= 0 && y < C) { loss = l - to_f32(row[y]); cnt = 1.f; }
    else if (y != ignore) { loss = __builtin_nanf(""); cnt = 1.f; }     // a label outside [0, C): torch asserts on the device; here the loss is NaN
  }
  loss = wave_sum(loss); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
// out[0] = mean loss over the counted rows, out[1] = their number.  No counted row: NaN (0 / 0), as F.cross_entropy(reduction
// = 'mean') gives when every label is ignore_index; a label outside [0, C) that is not ignore_index makes the loss NaN (torch
// raises a device-side assert there: both are loud, neither skips the row silently — ADVICE r5)
__global__ __launch_bounds__(256) void ce_rows_final_kernel(const float* __restrict__ partial, int64_t nblocks, float* __restrict__ out) {
  __shared__ float red[2][4];
  float a = 0.f, b = 0.f;
  for (int64_t i = threadIdx.x; i < nblocks; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), n = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    out[0] = n > 0.f ? s / n : __builtin_nanf("");
    out[1] = n;
  }
}
// dlogits[r][c] = g / count * (softmax(r)[c] - [c == label]) for counted rows, 0 elsewhere (ignored rows, padding columns c >= C)
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                           int64_t N, int C, int64_t ignore, const float* __restrict__ lse,
                                                           const float* __restrict__ stat, const float* __restrict__ gout,
                                                           T* __restrict__ dlogits, int64_t ldd, int Cp) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const int64_t y = labels[r];
  const bool on = y != ignore && y >= 0 && y < C && stat[1] > 0.f;
  const float g = on ? gout[0] / stat[1] : 0.f;
  const float l = lse[r];
  const T* row = logits + r * ld;
  T* drow = dlogits + r * ldd;
  for (int c = 0; c < Cp; ++c) {
    float d = 0.f;
    if (on && c < C) d = g * (__expf(to_f32(row[c]) - l) - (c == (int)y ? 1.f : 0.f));
    drow[c] = from_f32<T>(d);
  }
}

// Fast forms for the model's shape (bf16 rows of exactly 32 columns = 64 bytes, C <= 32): the row travels as four 16-byte
// accesses per thread instead of 3 x C two-byte loads (the generic kernels above ran at 0.6 TB/s on the 590 k x 32 logits).
__device__ __forceinline__ void ce_load32(const bf16_t* row, float (&v)[32]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4 w = reinterpret_cast<const u32x4*>(row)[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[q * 8 + 2 * i] = bf16lo(w[i]); v[q * 8 + 2 * i + 1] = bf16hi(w[i]); }
  }
}
__global__ __launch_bounds__(256) void ce_rows32_fwd_kernel(const bf16_t* __restrict__ logits, const int64_t* __restrict__ labels, int64_t N,
                                                             int C, int64_t ignore, float* __restrict__ lse, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (r < N) {
    float v[32];
    ce_load32(logits + r * 32, v);
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 32; ++c) m = fmaxf(m, c < C ? v[c] : -INFINITY);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) sum += c < C ? __expf(v[c] - m) : 0.f;
    const float l = m + __logf(sum);
    lse[r] = l;
    const int64_t y = labels[r];
    if (y != ignore && y >= 0 && y < C) {
      float ly = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) ly = c == (int)y ? v[c] : ly;
      loss = l - ly; cnt = 1.f;
    } else if (y != ignore) { loss = __builtin_nanf(""); cnt = 1.f; }
  }
  loss = wave_sum(loss); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
__global__ __launch_bounds__(256) void ce_rows32_bwd_kernel(const bf16_t* __restrict__ logits, const int64_t* __restrict__ labels, int64_t N,
                                                             int C, int64_t ignore, const float* __restrict__ lse, const float* __restrict__ stat,
                                                             const float* __restrict__ gout, bf16_t* __restrict__ dlogits) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const int64_t y = labels[r];
  const bool on = y != ignore && y >= 0 && y < C && stat[1] > 0.f;
  u32x4 o[4] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
  if (on) {
    const float g = gout[0] / stat[1], l = lse[r];
    float v[32];
    ce_load32(logits + r * 32, v);
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = c < C ? g * (__expf(v[c] - l) - (c == (int)y ? 1.f : 0.f)) : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[q][i] = pack_bf16x2(v[q * 8 + 2 * i], v[q * 8 + 2 * i + 1]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) reinterpret_cast<u32x4*>(dlogits + r * 32)[q] = o[q];
}

// ------------------------------------------------------------------------------------------
// triplet loss with D = 1 - sigmoid(cos).  nn.CosineSimilarity eps = 1e-8 (clamps each norm).
// ------------------------------------------------------------------------------------------
constexpr float COS_EPS = 1e-8f;

// row norms (clamped at eps): one wave per row
__global__ void row_norm_kernel(const float* __restrict__ X, int n, int dim, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) { const float a = X[(int64_t)i * dim + c]; s += a * a; }
  s = wave_sum(s);
  if (lane == 0) out[i] = fmaxf(sqrtf(s), COS_EPS);
}

// block = 4 waves, wave handles one anchor i and loops over j
__global__ void sigcos_dist_kernel(const float* __restrict__ P, const float* __restrict__ Dm,
                                   const float* __restrict__ pn, const float* __restrict__ dn, int n_p, int n_d,
                                   int dim, float* __restrict__ dist, float* __restrict__ selfd,
                                   float* __restrict__ cosout) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_p) return;
  const float np = pn[i];
  float np2 = 0.f;
  for (int c = lane; c < dim; c += 64) { const float a = P[(int64_t)i * dim + c]; np2 += a * a; }
  np2 = wave_sum(np2);
  for (int j = 0; j < n_d; ++j) {
    float dot = 0.f;
    for (int c = lane; c < dim; c += 64) dot += P[(int64_t)i * dim + c] * Dm[(int64_t)j * dim + c];
    dot = wave_sum(dot);
    if (lane == 0) {
      const float cs = dot / (np * dn[j]);
      cosout[(int64_t)i * n_d + j] = cs;
      dist[(int64_t)i * n_d + j] = 1.0f - 1.0f / (1.0f + __expf(-cs));
    }
  }
  if (lane == 0) {  // anchor-as-positive distance
    const float cs = np2 / (np * np);
    selfd[i] = 1.0f - 1.0f / (1.0f + __expf(-cs));
  }
}

// one block per anchor i.  partial[i] = sum of hinges, cnt[i] = number of triplets.
// When `gcoef` != nullptr also writes dL/dDist[i][j] (unnormalised: +count for positives, -count for negatives).
__global__ void triplet_reduce_kernel(const float* __restrict__ dist, const float* __restrict__ selfd,
                                      const int8_t* __restrict__ gt, int n_p, int n_d, float margin,
                                      float* __restrict__ partial, float* __restrict__ cnt,
                                      float* __restrict__ gcoef) {
  extern __shared__ int lists[];  // pos list then neg list (n_d ints each)
  __shared__ int npos_s, nneg_s;
  __shared__ float red[8];
  const int i = blockIdx.x;
  int* pos = lists; int* neg = lists + n_d;
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int j = 0; j < n_d; ++j) {
      const int8_t v = gt[(int64_t)i * n_d + j];
      if (v == 1) pos[a++] = j; else if (v == 0) neg[b++] = j;
    }
    npos_s = a; nneg_s = b;
  }
  __syncthreads();
  const int np_ = npos_s, nn_ = nneg_s;
  const float* drow = dist + (int64_t)i * n_d;
  float s = 0.f;
  if (gcoef) for (int j = threadIdx.x; j < n_d; j += blockDim.x) gcoef[(int64_t)i * n_d + j] = 0.f;
  __syncthreads();
  float count = 0.f;
  if (np_ > 0 && nn_ > 0) {
    count = (float)np_ * (float)nn_;
    for (int t = threadIdx.x; t < np_ * nn_; t += blockDim.x) {
      const int pj = pos[t / nn_], nj = neg[t % nn_];
      const float hv = drow[pj] - drow[nj] + margin;
      if (hv > 0.f) {
        s += hv;
        if (gcoef) { atomicAdd(&gcoef[(int64_t)i * n_d + pj], 1.0f); atomicAdd(&gcoef[(int64_t)i * n_d + nj], -1.0f); }
      }
    }
  } else if (nn_ > 0) {
    count = (float)nn_;
    const float dself = selfd[i];
    for (int t = threadIdx.x; t < nn_; t += blockDim.x) {
      const int nj = neg[t];
      const float hv = dself - drow[nj] + margin;
      if (hv > 0.f) { s += hv; if (gcoef) atomicAdd(&gcoef[(int64_t)i * n_d + nj], -1.0f); }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    partial[i] = t; cnt[i] = count;
  }
}

__global__ void triplet_final_kernel(const float* __restrict__ partial, const float* __restrict__ cnt, int n_p,
                                     float* __restrict__ loss, float* __restrict__ n_tri) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f, c = 0.f;
    for (int i = 0; i < n_p; ++i) { s += partial[i]; c += cnt[i]; }
    if (c == 0.f) c = 1.f;
    *loss = s / c; *n_tri = c;
  }
}

// dp_i = sum_j G_ij * (d_j/(|p_i||d_j|) - cos_ij p_i/|p_i|^2); G_ij = -sig'(cos) * coef_ij * gscale / n_tri
__global__ void triplet_bwd_p_kernel(const float* __restrict__ P, const float* __restrict__ Dm,
                                     const float* __restrict__ pn, const float* __restrict__ dn,
                                     const float* __restrict__ cosm, const float* __restrict__ gcoef, int n_p,
                                     int n_d, int dim, float gscale, const float* __restrict__ n_tri,
                                     float* __restrict__ dp) {
  const int i = blockIdx.x;
  const float gs = gscale / n_tri[0];
  const float np = pn[i];
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float acc = 0.f;
    const float pc = P[(int64_t)i * dim + c];
    for (int j = 0; j < n_d; ++j) {
      const float gc = gcoef[(int64_t)i * n_d + j];
      if (gc == 0.f) continue;
      const float cs = cosm[(int64_t)i * n_d + j];
      const float sg = 1.0f / (1.0f + __expf(-cs));
      const float G = -sg * (1.0f - sg) * gc * gs;
      acc += G * (Dm[(int64_t)j * dim + c] / (np * dn[j]) - cs * pc / (np * np));
    }
    dp[(int64_t)i * dim + c] = acc;
  }
}
__global__ void triplet_bwd_d_kernel(const float* __restrict__ P, const float* __restrict__ Dm,
                                     const float* __restrict__ pn, const float* __restrict__ dn,
                                     const float* __restrict__ cosm, const float* __restrict__ gcoef, int n_p,
                                     int n_d, int dim, float gscale, const float* __restrict__ n_tri,
                                     float* __restrict__ dd) {
  const int j = blockIdx.x;
  const float gs = gscale / n_tri[0];
  const float nd = dn[j];
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float acc = 0.f;
    const float dc = Dm[(int64_t)j * dim + c];
    for (int i = 0; i < n_p; ++i) {
      const float gc = gcoef[(int64_t)i * n_d + j];
      if (gc == 0.f) continue;
      const float cs = cosm[(int64_t)i * n_d + j];
      const float sg = 1.0f / (1.0f + __expf(-cs));
      const float G = -sg * (1.0f - sg) * gc * gs;
      acc += G * (P[(int64_t)i * dim + c] / (pn[i] * nd) - cs * dc / (nd * nd));
    }
    dd[(int64_t)j * dim + c] = acc;
  }
}

struct TripletBuf {
  float *dist, *selfd, *partial, *cnt, *cosm, *gcoef, *pn, *dn;
  TripletBuf(float* base, int64_t n_p, int64_t n_d) {
    dist = base; selfd = dist + n_p * n_d; partial = selfd + n_p; cnt = partial + n_p;
    cosm = cnt + n_p; gcoef = cosm + n_p * n_d; pn = gcoef + n_p * n_d; dn = pn + n_p;
  }
};
}  // namespace

extern "C" int dl_cos_rowloss_fwd(const float* x, const float* y, float* row_loss, float* loss_sum, int64_t n_rows,
                                  int64_t D, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && y && row_loss && n_rows > 0 && D > 0 && D % 4 == 0, DL_ERR_ARG, "dl_cos_rowloss_fwd: bad args");
  hipLaunchKernelGGL(cos_rowloss_fwd_kernel, dim3((uint32_t)((n_rows + 3) / 4)), dim3(256), 0, s, x, y, row_loss,
                     n_rows, (int)D);
  DL_CHECK_LAUNCH("dl_cos_rowloss_fwd");
  if (loss_sum) {
    hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_loss, n_rows, 1.0f, loss_sum);
    DL_CHECK_LAUNCH("dl_cos_rowloss_fwd(sum)");
  }
  return DL_OK;
}

extern "C" int dl_cos_rowloss_bwd(const float* x, const float* y, float grad_scale, float* dx, int64_t n_rows,
                                  int64_t D, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && y && dx && n_rows > 0 && D > 0 && D % 4 == 0, DL_ERR_ARG, "dl_cos_rowloss_bwd: bad args");
  hipLaunchKernelGGL(cos_rowloss_bwd_kernel, dim3((uint32_t)((n_rows + 3) / 4)), dim3(256), 0, s, x, y, grad_scale,
                     dx, n_rows, (int)D);
  DL_CHECK_LAUNCH("dl_cos_rowloss_bwd");
  return DL_OK;
}

extern "C" size_t dl_triplet_sigcos_buffer_floats(int64_t n_p, int64_t n_d) {
  return (size_t)(3 * n_p * n_d + 4 * n_p + n_d);
}

// `dist` must hold dl_triplet_sigcos_buffer_floats(n_p, n_d) floats: the distance matrix first, then
// scratch (anchor-as-positive distances, per-anchor partials, cosines, gradient coefficients, norms)
// that the backward call reuses.
extern "C" int dl_triplet_sigcos_fwd(const float* p_lats, const float* d_lats, const int8_t* gt, int64_t n_p,
                                     int64_t n_d, int64_t dim, float margin, float* dist, float* loss, float* n_tri,
                                     dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(p_lats && d_lats && gt && dist && loss && n_tri && n_p > 0 && n_d > 0 && dim > 0, DL_ERR_ARG,
               "dl_triplet_sigcos_fwd: bad args");
  DL_CHECK_ARG(n_d <= 8192, DL_ERR_SHAPE, "dl_triplet_sigcos_fwd: n_d too large for the LDS index lists");
  TripletBuf b(dist, n_p, n_d);
  hipLaunchKernelGGL(row_norm_kernel, dim3((uint32_t)((n_p + 3) / 4)), dim3(256), 0, s, p_lats, (int)n_p, (int)dim, b.pn);
  hipLaunchKernelGGL(row_norm_kernel, dim3((uint32_t)((n_d + 3) / 4)), dim3(256), 0, s, d_lats, (int)n_d, (int)dim, b.dn);
  hipLaunchKernelGGL(sigcos_dist_kernel, dim3((uint32_t)((n_p + 3) / 4)), dim3(256), 0, s, p_lats, d_lats,
                     (const float*)b.pn, (const float*)b.dn, (int)n_p, (int)n_d, (int)dim, b.dist, b.selfd, b.cosm);
  DL_CHECK_LAUNCH("dl_triplet_sigcos_fwd(dist)");
  hipLaunchKernelGGL(triplet_reduce_kernel, dim3((uint32_t)n_p), dim3(256), (size_t)n_d * 2 * sizeof(int), s,
                     (const float*)b.dist, (const float*)b.selfd, gt, (int)n_p, (int)n_d, margin, b.partial, b.cnt,
                     (float*)nullptr);
  DL_CHECK_LAUNCH("dl_triplet_sigcos_fwd(reduce)");
  hipLaunchKernelGGL(triplet_final_kernel, dim3(1), dim3(64), 0, s, (const float*)b.partial, (const float*)b.cnt,
                     (int)n_p, loss, n_tri);
  DL_CHECK_LAUNCH("dl_triplet_sigcos_fwd(final)");
  return DL_OK;
}

extern "C" int dl_triplet_sigcos_bwd(const float* p_lats, const float* d_lats, const int8_t* gt, const float* dist,
                                     int64_t n_p, int64_t n_d, int64_t dim, float margin, const float* n_tri,
                                     float grad_out, float* dp, float* dd, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(p_lats && d_lats && gt && dist && n_tri && dp && dd && n_p > 0 && n_d > 0 && dim > 0, DL_ERR_ARG,
               "dl_triplet_sigcos_bwd: bad args");
  TripletBuf b(const_cast<float*>(dist), n_p, n_d);
  hipLaunchKernelGGL(triplet_reduce_kernel, dim3((uint32_t)n_p), dim3(256), (size_t)n_d * 2 * sizeof(int), s,
                     (const float*)b.dist, (const float*)b.selfd, gt, (int)n_p, (int)n_d, margin, b.partial, b.cnt,
                     b.gcoef);
  DL_CHECK_LAUNCH("dl_triplet_sigcos_bwd(coef)");
  hipLaunchKernelGGL(triplet_bwd_p_kernel, dim3((uint32_t)n_p), dim3(256), 0, s, p_lats, d_lats, (const float*)b.pn,
                     (const float*)b.dn, (const float*)b.cosm, (const float*)b.gcoef, (int)n_p, (int)n_d, (int)dim,
                     grad_out, n_tri, dp);
  hipLaunchKernelGGL(triplet_bwd_d_kernel, dim3((uint32_t)n_d), dim3(256), 0, s, p_lats, d_lats, (const float*)b.pn,
                     (const float*)b.dn, (const float*)b.cosm, (const float*)b.gcoef, (int)n_p, (int)n_d, (int)dim,
                     grad_out, n_tri, dd);
  DL_CHECK_LAUNCH("dl_triplet_sigcos_bwd");
  return DL_OK;
}

extern "C" size_t dl_ce_rows_workspace_bytes(int64_t N) { return (size_t)((N + 255) / 256) * 2 * sizeof(float); }

extern "C" int dl_ce_rows_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t N, int32_t C, int64_t ignore_index,
                              int32_t dtype, float* lse, float* out2, void* workspace, size_t workspace_bytes, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(logits && labels && lse && out2 && N > 0 && C > 0 && C <= CE_MAXC && ld >= C, DL_ERR_ARG,
               "dl_ce_rows_fwd: bad args (1 <= C <= %d, ld >= C)", CE_MAXC);
  DL_CHECK_ARG(dtype == DL_F32 || dtype == DL_BF16, DL_ERR_ARG, "dl_ce_rows_fwd: bad dtype");
  DL_CHECK_ARG(workspace && workspace_bytes >= dl_ce_rows_workspace_bytes(N), DL_ERR_WORKSPACE, "dl_ce_rows_fwd: workspace too small");
  const int64_t nb = (N + 255) / 256;
  if (dtype == DL_BF16 && ld == 32 && C <= 32 && ((uintptr_t)logits & 15) == 0)
    hipLaunchKernelGGL(ce_rows32_fwd_kernel, dim3((uint32_t)nb), dim3(256), 0, s, (const bf16_t*)logits, labels, N, (int)C, ignore_index, lse,
                       (float*)workspace);
  else if (dtype == DL_BF16)
    hipLaunchKernelGGL((ce_rows_fwd_kernel<bf16_t>), dim3((uint32_t)nb), dim3(256), 0, s, (const bf16_t*)logits, ld, labels, N, (int)C,
                       ignore_index, lse, (float*)workspace);
  else
    hipLaunchKernelGGL((ce_rows_fwd_kernel<float>), dim3((uint32_t)nb), dim3(256), 0, s, (const float*)logits, ld, labels, N, (int)C,
                       ignore_index, lse, (float*)workspace);
  hipLaunchKernelGGL(ce_rows_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nb, out2);
  DL_CHECK_LAUNCH("dl_ce_rows_fwd");
  return DL_OK;
}

extern "C" int dl_ce_rows_bwd(const void* logits, int64_t ld, const int64_t* labels, int64_t N, int32_t C, int64_t ignore_index,
                              int32_t dtype, const float* lse, const float* out2, const float* grad_out, void* dlogits, int64_t ldd,
                              int32_t Cp, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(logits && labels && lse && out2 && grad_out && dlogits && N > 0 && C > 0 && C <= CE_MAXC && ld >= C && Cp >= C && ldd >= Cp,
               DL_ERR_ARG, "dl_ce_rows_bwd: bad args");
  DL_CHECK_ARG(dtype == DL_F32 || dtype == DL_BF16, DL_ERR_ARG, "dl_ce_rows_bwd: bad dtype");
  const int64_t nb = (N + 255) / 256;
  if (dtype == DL_BF16 && ld == 32 && ldd == 32 && Cp == 32 && C <= 32 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0)
    hipLaunchKernelGGL(ce_rows32_bwd_kernel, dim3((uint32_t)nb), dim3(256), 0, s, (const bf16_t*)logits, labels, N, (int)C, ignore_index, lse,
                       out2, grad_out, (bf16_t*)dlogits);
  else if (dtype == DL_BF16)
    hipLaunchKernelGGL((ce_rows_bwd_kernel<bf16_t>), dim3((uint32_t)nb), dim3(256), 0, s, (const bf16_t*)logits, ld, labels, N, (int)C,
                       ignore_index, lse, out2, grad_out, (bf16_t*)dlogits, ldd, (int)Cp);
  else
    hipLaunchKernelGGL((ce_rows_bwd_kernel<float>), dim3((uint32_t)nb), dim3(256), 0, s, (const float*)logits, ld, labels, N, (int)C,
                       ignore_index, lse, out2, grad_out, (float*)dlogits, ldd, (int)Cp);
  DL_CHECK_LAUNCH("dl_ce_rows_bwd");
  return DL_OK;
}

// elementwise.hip — HBM-bound helpers on the path: MHLA token gate, positional add + dropout,
// dropout re-application, casts, positional-table gradient, fused AdamW.
#include <math.h>
#include "common.cuh"

namespace {

// ---------------- MHLA token gate -------------------------------------------------------------
// gate[b][j][l] = softmax_l(logits[b][l][j]); one wave per (b, j).
template <typename T>
__global__ void gate_softmax_kernel(const T* __restrict__ logits, float* __restrict__ gate, int L, int H) {
  const int bj = blockIdx.x, b = bj / H, j = bj % H, lane = threadIdx.x;
  const T* src = logits + (int64_t)b * L * H + j;
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) mx = fmaxf(mx, to_f32(src[(int64_t)l * H]));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int l = lane; l < L; l += 64) sum += __expf(to_f32(src[(int64_t)l * H]) - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float* dst = gate + (int64_t)bj * L;
  for (int l = lane; l < L; l += 64) dst[l] = __expf(to_f32(src[(int64_t)l * H]) - mx) * inv;
}

// out[b][f] = v[b][f] * (gate_flat[b][f / hd] + res)
template <typename T>
__global__ void gate_apply_kernel(const T* __restrict__ v, const float* __restrict__ gate, T* __restrict__ out,
                                  int64_t n4, int64_t per_sample, int hd, int64_t gate_per_sample, float res) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int64_t e = i * 4;
  const int64_t b = e / per_sample, f = e % per_sample;
  const float gv = gate[b * gate_per_sample + f / hd] + res;
  f32x4 x = load4<T>(v + e);
  x *= gv;
  store4<T>(out + e, x);
}

// one wave per (b, j): dv = dout * (g + res); dgate = sum_c dout*v; dlogits = g * (dgate - sum_l g*dgate)
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ v, const float* __restrict__ gate,
                                T* __restrict__ dv, T* __restrict__ dlogits, int L, int H, int hd, float res) {
  extern __shared__ float dg_s[];
  const int bj = blockIdx.x, b = bj / H, j = bj % H, lane = threadIdx.x;
  const int64_t base = ((int64_t)b * H + j) * L * hd;  // flat offset of this (b, j) slab of L*hd elements
  const float* g = gate + (int64_t)bj * L;
  float dot = 0.f;
  for (int l = lane; l < L; l += 64) {
    const float gl = g[l];
    float acc = 0.f;
    for (int c = 0; c < hd; c += 4) {
      const int64_t e = base + (int64_t)l * hd + c;
      const f32x4 d4 = load4<T>(dout + e), v4 = load4<T>(v + e);
      acc += d4[0] * v4[0] + d4[1] * v4[1] + d4[2] * v4[2] + d4[3] * v4[3];
      store4<T>(dv + e, d4 * (gl + res));
    }
    dg_s[l] = acc;
    dot += gl * acc;
  }
  dot = wave_sum(dot);
  __syncthreads();
  T* dst = dlogits + (int64_t)b * L * H + j;
  for (int l = lane; l < L; l += 64) dst[(int64_t)l * H] = from_f32<T>(g[l] * (dg_s[l] - dot));
}

// Same contract, coalesced: the (b, j) slab is L rows of hd elements laid out contiguously, so 64 lanes x 4 elements
// cover 256 / hd whole rows per pass (LPR = hd / 4 lanes per row, a power of two) instead of one lane walking a whole
// row with 64-byte-strided 8-byte accesses; the per-row dot product is a shuffle reduction over LPR lanes.
template <typename T>
__global__ void gate_bwd_rows_kernel(const T* __restrict__ dout, const T* __restrict__ v, const float* __restrict__ gate,
                                     T* __restrict__ dv, T* __restrict__ dlogits, int L, int H, int hd, float res) {
  extern __shared__ float dg_s[];
  const int bj = blockIdx.x, b = bj / H, j = bj % H, lane = threadIdx.x;
  const int64_t base = ((int64_t)b * H + j) * L * hd;
  const float* g = gate + (int64_t)bj * L;
  const int lpr = hd >> 2, rpp = 64 / lpr;               // lanes per row, rows per pass
  const int sub = lane % lpr, rsub = lane / lpr;
  float dot = 0.f;
  for (int l0 = 0; l0 < L; l0 += rpp) {
    const int l = l0 + rsub;
    float acc = 0.f, gl = 0.f;
    if (l < L) {
      gl = g[l];
      const int64_t e = base + (int64_t)l * hd + sub * 4;
      const f32x4 d4 = load4<T>(dout + e), v4 = load4<T>(v + e);
      acc = d4[0] * v4[0] + d4[1] * v4[1] + d4[2] * v4[2] + d4[3] * v4[3];
      store4<T>(dv + e, d4 * (gl + res));
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (l < L && sub == 0) { dg_s[l] = acc; dot += gl * acc; }
  }
  dot = wave_sum(dot);
  __syncthreads();
  T* dst = dlogits + (int64_t)b * L * H + j;
  for (int l = lane; l < L; l += 64) dst[(int64_t)l * H] = from_f32<T>(g[l] * (dg_s[l] - dot));
}

// ---------------- positional add + dropout / dropout apply -------------------------------------
template <typename T>
__global__ void add_rowmod_dropout_kernel(const T* __restrict__ x, const T* __restrict__ pe, T* __restrict__ y,
                                          int64_t M, int D, int64_t L, uint32_t thr16, float inv_keep,
                                          uint64_t seed, const uint64_t* __restrict__ seed_off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = M * (D / 4);
  if (i >= n4) return;
  const int64_t row = i / (D / 4);
  const int col = (int)(i % (D / 4)) * 4;
  f32x4 v = load4<T>(x + row * D + col);
  if (pe) v += load4<T>(pe + (row % L) * D + col);
  if (thr16) v = dl_dropout4(v, dl_eff_seed(seed, seed_off), (uint64_t)row, (uint64_t)col, (uint64_t)D, thr16, inv_keep);
  store4<T>(y + row * D + col, v);
}

template <typename T>
__global__ void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int D, int64_t ldx,
                                     int64_t ldy, uint32_t thr16, float inv_keep, uint64_t seed,
                                     const uint64_t* __restrict__ seed_off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = rows * (D / 4);
  if (i >= n4) return;
  const int64_t row = i / (D / 4);
  const int col = (int)(i % (D / 4)) * 4;
  f32x4 v = load4<T>(x + row * ldx + col);
  v = dl_dropout4(v, dl_eff_seed(seed, seed_off), (uint64_t)row, (uint64_t)col, (uint64_t)D, thr16, inv_keep);
  store4<T>(y + row * ldy + col, v);
}

// dx = dy * gelu'(pre)
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 d = load4<T>(dy + i * 4);
  const f32x4 q = load4<T>(pre + i * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] *= gelu_erf_grad(q[e]);  // standalone kernel is HBM-bound: exact form
  store4<T>(dx + i * 4, d);
}

// ---------------- MHLA backward: dpre = gelu'(pre) o (dlogits W2)  (round 5) ----------------------------------------------
// The gradient of lin2 (d_diff -> H = 8 gate logits, model/PMMA/encoder.py:127-140) with respect to its input: a product
// with an inner dimension of EIGHT, followed by the derivative of the GELU in front of it.  Through round 4 the 8 columns
// were zero-padded to one 64-deep k-step (two torch launches) so that dl_gemm could take it with the gelu' epilogue
// (77 us + 46 us of padding per MHLA block at batch 256); as an elementwise kernel it is one pass over `pre`:
// a thread keeps its 8 columns of W2 (H x 8 values) in registers for the whole launch and walks rows with a grid stride.
template <typename T, int H>
__global__ __launch_bounds__(256) void gate_dpre_kernel(const T* __restrict__ dl, const T* __restrict__ w2, const T* __restrict__ pre,
                                                         T* __restrict__ out, int64_t M, int dd) {
  const int cpr = dd >> 3;                                   // 8-column chunks per row
  const int rpb = 256 / cpr > 0 ? 256 / cpr : 1;             // rows per block and pass (cpr <= 256 is checked by the host)
  const int ch = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
  if (rsub >= rpb) return;
  float w[H][8];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const f32x4 a = load4<T>(w2 + (int64_t)h * dd + ch * 8), b = load4<T>(w2 + (int64_t)h * dd + ch * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { w[h][e] = a[e]; w[h][4 + e] = b[e]; }
  }
  for (int64_t r = (int64_t)blockIdx.x * rpb + rsub; r < M; r += (int64_t)gridDim.x * rpb) {
    float d[H];
    if constexpr (H % 4 == 0) {
#pragma unroll
      for (int h = 0; h < H; h += 4) {
        const f32x4 v = load4<T>(dl + r * H + h);
        d[h] = v[0]; d[h + 1] = v[1]; d[h + 2] = v[2]; d[h + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) d[h] = to_f32(dl[r * H + h]);
    }
    const f32x4 p0 = load4<T>(pre + r * dd + ch * 8), p1 = load4<T>(pre + r * dd + ch * 8 + 4);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] = fmaf(d[h], w[h][e], a0[e]); a1[e] = fmaf(d[h], w[h][4 + e], a1[e]); }
    store4<T>(out + r * dd + ch * 8, a0 * gelu_grad4<T>(p0));
    store4<T>(out + r * dd + ch * 8 + 4, a1 * gelu_grad4<T>(p1));
  }
}

// ---------------- LLM feature ingest: fill bit + site pooling in ONE pass -------------------------
// x [B][S][F]; fill[b][s] = (sum_f x[b][s][f] == 0); pooled[b][j][f] = mean_c xcat[b][c*n_site + j][f] for
// c < site_len, xcat = [x | fill]; pooled is Fp = ceil8(F + 1) wide, zero beyond column F.
// One wave per (b, j); lane -> 8-element chunks.
template <typename T, typename TO>
__global__ void fill_pool_kernel(const T* __restrict__ x, T* __restrict__ fill, TO* __restrict__ pooled, int64_t BJ,
                                 int n_site, int site_len, int F, int Fp) {
  const int lane = threadIdx.x & 63;
  const int64_t bj = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (bj >= BJ) return;
  const int64_t b = bj / n_site;
  const int j = (int)(bj % n_site);
  const int S = n_site * site_len;
  const int nch = F >> 3;                 // F % 8 == 0
  float acc[2][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
  float fillsum = 0.f;
  for (int c = 0; c < site_len; ++c) {
    const int srow = c * n_site + j;
    const T* row = x + ((int64_t)b * S + srow) * F;
    float rs = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ch = lane + 64 * q;
      if (ch < nch) {
        const f32x4 v0 = load4<T>(row + ch * 8), v1 = load4<T>(row + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[q][e] += v0[e]; acc[q][4 + e] += v1[e]; rs += v0[e] + v1[e]; }
      }
    }
    rs = wave_sum(rs);
    const float fb = (rs == 0.f) ? 1.f : 0.f;
    fillsum += fb;
    if (lane == 0) fill[(int64_t)b * S + srow] = from_f32<T>(fb);
  }
  const float inv = 1.0f / (float)site_len;
  TO* out = pooled + bj * Fp;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ch = lane + 64 * q;
    if (ch < nch) {
      store4<TO>(out + ch * 8, f32x4{acc[q][0] * inv, acc[q][1] * inv, acc[q][2] * inv, acc[q][3] * inv});
      store4<TO>(out + ch * 8 + 4, f32x4{acc[q][4] * inv, acc[q][5] * inv, acc[q][6] * inv, acc[q][7] * inv});
    }
  }
  if (lane == 0) {
    out[F] = from_f32<TO>(fillsum * inv);
    for (int e = F + 1; e < Fp; ++e) out[e] = from_f32<TO>(0.f);
  }
}

// ---------------- batch assembly from a device-resident embedding store ----------------------------------
// out[b][s][:] = store[offsets[b] + (s mod len_b)][:] for s < reps_b * len_b, zero afterwards, where
//   repeat = 1 : reps_b = S / len_b   (reference repeat_pad, utils.py:314-324; len_b > S gives an all-zero block)
//   repeat = 0 : reps_b = 1           (reference tail_pad,   utils.py:304-312; at most S rows are taken)
// One 16-byte chunk per thread; 32-bit arithmetic inside a sample.
template <typename T>
__global__ __launch_bounds__(256) void gather_pad_kernel(const T* __restrict__ store, const int64_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ lengths, T* __restrict__ out, int S,
                                                          int F, int repeat) {
  const int b = blockIdx.y;
  const int cpr = F / (16 / (int)sizeof(T));
  const int len = lengths[b];
  const int64_t off = offsets[b];
  const int filled = len <= 0 ? 0 : (repeat ? (S / len) * len : (len < S ? len : S));
  const int per_sample = S * cpr;
  for (int ci = blockIdx.x * 256 + threadIdx.x; ci < per_sample; ci += gridDim.x * 256) {
    const int srow = ci / cpr, ch = ci - srow * cpr;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (srow < filled) {
      const int r = repeat ? srow % len : srow;
      v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(store + (off + r) * F) + ch * 16);
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out + ((int64_t)b * S + srow) * F) + ch * 16) = v;
  }
}

// ---------------- ProteinCNN head: embedding gather + fill-bit column + halo rows, in one pass ---------------
// out[b][halo + l][0..D-1] = weight[ids[b][l]][:], out[b][halo + l][D] = fill[b][l]; halo rows are zero.
// (basic_model.py:168-171: embedding, cat with the fill bit; the zero halo is this library's conv padding.)
template <typename T>
__global__ __launch_bounds__(256) void embed_pad_kernel(const int64_t* __restrict__ ids, const T* __restrict__ weight,
                                                         const T* __restrict__ fill, T* __restrict__ out, int64_t B, int L,
                                                         int V, int D, int halo) {
  extern __shared__ __attribute__((aligned(16))) char ep_smem[];
  T* tab = reinterpret_cast<T*>(ep_smem);                    // [V][C]: rows padded to C = D + 1 -> one 16-byte read per chunk
  const int C = D + 1, cpr = C / 8, LP = L + 2 * halo;
  // weight arrives padded to [V][C] (last column unused): the table is copied with 16-byte loads
  for (int i = threadIdx.x; i < V * C * (int)sizeof(T) / 16; i += 256)
    reinterpret_cast<u32x4*>(ep_smem)[i] = reinterpret_cast<const u32x4*>(weight)[i];
  __syncthreads();
  // grid (chunks of one sample / 256, B): 32-bit index arithmetic only (three 64-bit divisions per chunk were the
  // whole cost of the first version of this kernel)
  const int b = blockIdx.y;
  const int per_sample = LP * cpr;
  for (int ci = blockIdx.x * 256 + threadIdx.x; ci < per_sample; ci += gridDim.x * 256) {
    const int lp = ci / cpr, ch = ci - lp * cpr, l = lp - halo;
    const int64_t row = (int64_t)b * LP + lp;
    T v[8];
    if (l < 0 || l >= L) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    } else {
      int64_t id = ids[(int64_t)b * L + l];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      const T* src = tab + id * C + ch * 8;
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e];
      }
      if (ch == cpr - 1) v[7] = fill[(int64_t)b * L + l];  // the last column is the fill bit
    }
    T* dst = out + row * C + ch * 8;
    if constexpr (sizeof(T) == 2) {
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pk[e] = (uint32_t)__builtin_bit_cast(uint16_t, v[2 * e]) | ((uint32_t)__builtin_bit_cast(uint16_t, v[2 * e + 1]) << 16);
      *reinterpret_cast<u32x4*>(dst) = pk;              // one 16-byte store per chunk
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = v[e];
    }
  }
}

// ---------------- ProteinCNN tail: the reference's (B,C,L).view(B,L,C) + site pooling, in one pass -----------
// z is the channel-last conv output with `halo` zero rows around every sample: z[b][halo + l][c].  The reference
// holds the same values channel-first (mem[b][c*L + l]), REINTERPRETS that buffer as (B, L, C) and then averages
// the S = site_len chunks of n_site rows: pooled[b][o] = 1/S sum_s mem[b][s*n_site*C + o], o = j*C + c'.
// With L = S*n_site and o = n_site*q + r (q < C, r < n_site) this is
//   pooled[b][n_site*q + r] = 1/S sum_s z[b][halo + n_site*((s*C + q) % S) + r][(s*C + q) / S],
// so a workgroup that loads the S strips of RR rows {n_site*k + r0 .. + RR} (k < S) has every contribution to
// its C*RR outputs.  Backward is the same index map read the other way (each z element feeds one output).
constexpr int SP_RR = 32;
constexpr int SP_PITCH = 34;      // elements per LDS image row of the forward kernel (68 bytes)
template <typename T>
__global__ __launch_bounds__(256) void cnn_sitepool_fwd_kernel(const T* __restrict__ z, T* __restrict__ out, int L, int C,
                                                                int halo, int S, const int32_t* __restrict__ row_of) {
  // LDS image T[v = c * S + k][i] (pitch SP_PITCH elements): the row index v is exactly the order in which the
  // S contributions of output column q are consumed (v = s * C + q), so the gather below reads contiguous rows;
  // the scattered 2-byte writes of the load phase land on 8 different banks per 16 lanes with this pitch.
  extern __shared__ __attribute__((aligned(16))) char sp_smem[];
  T* img = reinterpret_cast<T*>(sp_smem);                   // [C * S][SP_PITCH]
  const int n_site = L / S;
  const int b = blockIdx.y, r0 = blockIdx.x * SP_RR, tid = threadIdx.x;
  const int LP = L + 2 * halo;
  const T* zb = z + (int64_t)b * LP * C;
  const int cpr = C / 8;                                     // 16-byte chunks per row
  for (int c = tid; c < S * SP_RR * cpr; c += 256) {
    const int row = c / cpr, ch = c % cpr, k = row / SP_RR, i = row % SP_RR;
    const int l = n_site * k + r0 + i;
    T v[8];
    if (r0 + i < n_site) {
      // row_of (round 4: ProteinCNN on distinct rows): position (b, l) is represented by compact row row_of[b * L + l] of z
      const T* src = row_of ? z + (int64_t)row_of[(int64_t)b * L + l] * C : zb + (int64_t)(halo + l) * C;
      *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(src + ch * 8);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) img[((ch * 8 + e) * S + k) * SP_PITCH + i] = v[e];
  }
  __syncthreads();
  const float inv = 1.0f / (float)S;
  // thread -> (q, half of the RR rows): 16 consecutive outputs
  for (int w = tid; w < C * 2; w += 256) {
    const int q = w >> 1, i0 = (w & 1) * (SP_RR / 2);
    float acc[SP_RR / 2];
#pragma unroll
    for (int i = 0; i < SP_RR / 2; ++i) acc[i] = 0.f;
    for (int s = 0; s < S; ++s) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(img + (int64_t)(s * C + q) * SP_PITCH + i0);
#pragma unroll
      for (int i = 0; i < SP_RR / 4; ++i) {
        const uint32_t u = src[i];
        acc[2 * i] += bf16lo(u);
        acc[2 * i + 1] += bf16hi(u);
      }
    }
    T* dst = out + (int64_t)b * L / S * C + (int64_t)n_site * q + r0 + i0;
#pragma unroll
    for (int i = 0; i < SP_RR / 2; ++i)
      if (r0 + i0 + i < n_site) dst[i] = from_f32<T>(acc[i] * inv);
  }
}

// Two (or S) equally shaped row streams [S][R][row] <-> one [R][S * row] buffer (encoder.py:50, `cat((prot, mol), -1)`
// before the self-attention layers, and its gradient): 16-byte chunks, both sides contiguous per (row, stream) segment.
__global__ __launch_bounds__(256) void interleave_streams_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                                                  int64_t R, int cpr, int S, int inverse, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % cpr);
    const int64_t rs = idx / cpr;
    const int s = (int)(rs % S);
    const int64_t r = rs / S;
    const int64_t split = ((int64_t)s * R + r) * cpr + j;       // position in the [S][R][row] layout
    if (inverse) dst[split] = src[idx]; else dst[idx] = src[split];
  }
}

// ahat[b][i][j] = din[i] * adj[b][j][i] * dout[j], dout[j] = clamp(sum_i adj[b][j][i], 1)^-1/2 (out-degree of source j),
// din[i] = clamp(sum_j adj[b][j][i], 1)^-1/2 (in-degree of destination i): dgl GraphConv norm='both' on a dense batched
// graph (basic_model.py:591-617 via dgl 1.0.2 GraphConv), one workgroup per sample, the fp32 adjacency tile in LDS.
template <typename TO>
__global__ __launch_bounds__(256) void norm_adj_kernel(const float* __restrict__ adj, TO* __restrict__ ahat, int n) {
  extern __shared__ __attribute__((aligned(16))) char na_smem[];
  float* a = reinterpret_cast<float*>(na_smem);               // [n][n + 1]
  float* dout = a + (size_t)n * (n + 1);                      // [n]
  float* din = dout + n;                                      // [n]
  const int b = blockIdx.x, tid = threadIdx.x, P = n + 1;
  const float* src = adj + (int64_t)b * n * n;
  for (int e = tid; e < n * n; e += 256) a[(e / n) * P + e % n] = src[e];
  __syncthreads();
  for (int r = tid; r < 2 * n; r += 256) {
    float sum = 0.f;
    if (r < n) { for (int i = 0; i < n; ++i) sum += a[r * P + i]; dout[r] = rsqrtf(fmaxf(sum, 1.f)); }
    else { const int c = r - n; for (int j = 0; j < n; ++j) sum += a[j * P + c]; din[c] = rsqrtf(fmaxf(sum, 1.f)); }
  }
  __syncthreads();
  TO* dst = ahat + (int64_t)b * n * n;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    dst[e] = from_f32<TO>(a[j * P + i] * din[i] * dout[j]);
  }
}

// the same for graphs whose fp32 tile does not fit LDS (n > 190): degrees first (only they live in LDS), the adjacency
// is read from global memory / L2 twice
template <typename TO>
__global__ __launch_bounds__(256) void norm_adj_any_kernel(const float* __restrict__ adj, TO* __restrict__ ahat, int n) {
  extern __shared__ __attribute__((aligned(16))) char na_smem[];
  float* dout = reinterpret_cast<float*>(na_smem);            // [n]
  float* din = dout + n;                                      // [n]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* a = adj + (int64_t)b * n * n;
  for (int r = tid; r < 2 * n; r += 256) {
    float sum = 0.f;
    if (r < n) { for (int i = 0; i < n; ++i) sum += a[(int64_t)r * n + i]; dout[r] = rsqrtf(fmaxf(sum, 1.f)); }
    else { const int c = r - n; for (int j = 0; j < n; ++j) sum += a[(int64_t)j * n + c]; din[c] = rsqrtf(fmaxf(sum, 1.f)); }
  }
  __syncthreads();
  TO* dst = ahat + (int64_t)b * n * n;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    dst[e] = from_f32<TO>(a[(int64_t)j * n + i] * din[i] * dout[j]);
  }
}

// dst[r] = [a[r] | b[r]] for two row-major buffers of ca / cb 16-byte chunks per row (inverse: split dst back into a, b)
__global__ __launch_bounds__(256) void concat2_kernel(u32x4* __restrict__ a, u32x4* __restrict__ b, u32x4* __restrict__ dst,
                                                       int ca, int cb, int inverse, int64_t total) {
  const int cw = ca + cb;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % cw);
    const int64_t r = idx / cw;
    u32x4* side = j < ca ? a + r * ca + j : b + r * cb + (j - ca);
    if (inverse) *side = dst[idx]; else dst[idx] = *side;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cnn_sitepool_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dz, int L, int C,
                                                                int halo, int S) {
  extern __shared__ __attribute__((aligned(16))) char sp_smem[];
  // [C][RR]; row q is rotated by 2 * (q >> 3): the 16 channel chunks a wave reads at once have q = (72 ch + const)
  // mod C, i.e. q >> 3 takes 16 distinct values, and without the rotation all of them sat on the same bank
  // (SQ_LDS_BANK_CONFLICT = 93 % of this kernel's LDS cycles)
  static_assert(SP_RR == 32, "rotation below assumes 32 site rows per workgroup");
  float* g = reinterpret_cast<float*>(sp_smem);
  const int n_site = L / S;
  const int b = blockIdx.y, r0 = blockIdx.x * SP_RR, tid = threadIdx.x;
  const int LP = L + 2 * halo;
  const float inv = 1.0f / (float)S;
  const T* db = dout + (int64_t)b * n_site * C;
  for (int e = tid; e < C * SP_RR; e += 256) {
    const int q = e / SP_RR, i = e % SP_RR;
    g[q * SP_RR + ((i + 2 * (q >> 3)) & 31)] = (r0 + i < n_site) ? to_f32(db[(int64_t)n_site * q + r0 + i]) * inv : 0.f;
  }
  __syncthreads();
  T* zb = dz + (int64_t)b * LP * C;
  const int cpr = C / 8;
  for (int c = tid; c < S * SP_RR * cpr; c += 256) {
    const int row = c / cpr, ch = c % cpr, k = row / SP_RR, i = row % SP_RR;
    if (r0 + i >= n_site) continue;
    const int l = n_site * k + r0 + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int q = ((ch * 8 + e) * S + k) % C;
      v[e] = g[q * SP_RR + ((i + 2 * (q >> 3)) & 31)];
    }
    T* dst = zb + (int64_t)(halo + l) * C + ch * 8;
    store4<T>(dst, f32x4{v[0], v[1], v[2], v[3]});
    store4<T>(dst + 4, f32x4{v[4], v[5], v[6], v[7]});
  }
  if (blockIdx.x == 0) {                                     // halo rows of this sample
    for (int c = tid; c < 2 * halo * cpr; c += 256) {
      const int hr = c / cpr, ch = c % cpr;
      const int row = hr < halo ? hr : L + hr;               // top rows 0..halo-1, bottom rows halo+L..
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(zb + (int64_t)row * C + ch * 8)) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}

// site_len 1 (the masked-LM pass of the SSL epochs: the view reinterpretation alone, no pooling): pooled[b][L*q + r] =
// z[b][halo + r][q] is a plain matrix transpose per sample, and so is its gradient.  64 x 64 tiles through LDS, 16-byte
// accesses on both sides (the general kernels above write 2-byte elements in 32-byte runs there: 1.6 TB/s).
//   dst[b][j][i] = src[b][i][j],  i < rows, j < cols; rows % 64 == cols % 64 == 0; row strides in elements.
constexpr int TV_PITCH = 66;      // elements per LDS row (33 dwords: the column gather of 8 x 8 lanes lands on distinct banks)
__global__ __launch_bounds__(256) void transpose_view_kernel(const bf16_t* __restrict__ src, int64_t src_bs, int64_t src_rs,
                                                              bf16_t* __restrict__ dst, int64_t dst_bs, int64_t dst_rs,
                                                              bf16_t* __restrict__ halo_base, int64_t halo_bs, int halo_rows, int64_t body_rows,
                                                              int row_chunks) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64 * TV_PITCH];
  const int b = blockIdx.z, i0 = blockIdx.x * 64, j0 = blockIdx.y * 64, tid = threadIdx.x;
  const bf16_t* sb = src + (int64_t)b * src_bs + (int64_t)i0 * src_rs + j0;
#pragma unroll
  for (int c = tid; c < 512; c += 256) {
    const int i = c >> 3, jc = c & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(sb + (int64_t)i * src_rs + jc * 8);
    uint32_t* t = reinterpret_cast<uint32_t*>(tile + i * TV_PITCH + jc * 8);      // (4-byte aligned: the pitch is even)
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  }
  __syncthreads();
  bf16_t* db = dst + (int64_t)b * dst_bs + (int64_t)j0 * dst_rs + i0;
#pragma unroll
  for (int c = tid; c < 512; c += 256) {
    const int j = c >> 3, ic = c & 7;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = (uint32_t)tile[(ic * 8 + 2 * e) * TV_PITCH + j] | ((uint32_t)tile[(ic * 8 + 2 * e + 1) * TV_PITCH + j] << 16);
    *reinterpret_cast<u32x4*>(db + (int64_t)j * dst_rs + ic * 8) = u32x4{w[0], w[1], w[2], w[3]};
  }
  if (halo_base && blockIdx.x == 0 && blockIdx.y == 0) {      // backward: the zero rows around this sample's gradient
    bf16_t* hb = halo_base + (int64_t)b * halo_bs;
    for (int c = tid; c < 2 * halo_rows * row_chunks; c += 256) {
      const int hr = c / row_chunks, ch = c % row_chunks;
      const int64_t row = hr < halo_rows ? hr : body_rows + hr;
      *reinterpret_cast<u32x4*>(hb + (row * row_chunks + ch) * 8) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 4 <= n) {
    store4<TD>(dst + i4, load4<TS>(src + i4));
  } else {
    for (int64_t i = i4; i < n; ++i) dst[i] = from_f32<TD>(to_f32(src[i]));
  }
}

// ---------------- weight preparation: many fp32 master tensors -> compute-dtype (transposed) copies ------
// One launch refreshes every low-precision / transposed weight image of the model after an optimiser step.
// block_map[b] = (item, tile): a 64x64 tile of item.src [rows][cols]; plain items land at
// dst[(row0 + r) * ld + c], transposed items at dst[c * ld + row0 + r] (row0 lets several parameters share one
// concatenated image, e.g. q|k|v).
template <typename TD>
__global__ __launch_bounds__(256) void weight_prep_kernel(const dl_wprep_item* __restrict__ items,
                                                          const int32_t* __restrict__ block_map) {
  __shared__ float tile[64][65];
  const dl_wprep_item it = items[block_map[2 * blockIdx.x]];
  const int tcols = (it.cols + 63) / 64;
  const int t = block_map[2 * blockIdx.x + 1];
  const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
  const int tid = threadIdx.x, tr = tid >> 4, tc4 = (tid & 15) * 4;
  TD* dst = reinterpret_cast<TD*>(it.dst);
  const bool vec = (it.cols & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + tr + 16 * i, c = c0 + tc4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < it.rows) {
      if (vec && c + 4 <= it.cols) v = *reinterpret_cast<const f32x4*>(it.src + (int64_t)r * it.cols + c);
      else for (int e = 0; e < 4; ++e) if (c + e < it.cols) v[e] = it.src[(int64_t)r * it.cols + c + e];
    }
    if (it.transpose == 2) {                        // strided scatter (conv layouts: tiny tensors, scalar stores)
      if (r < it.rows)
        for (int e = 0; e < 4; ++e)
          if (c + e < it.cols) dst[(int64_t)it.row0 + (int64_t)r * it.ld + (int64_t)(c + e) * it.cs] = from_f32<TD>(v[e]);
    } else if (!it.transpose) {
      if (r < it.rows) {
        TD* d = dst + (int64_t)(it.row0 + r) * it.ld + c;
        if (c + 4 <= it.cols && (it.ld & 3) == 0 && (((uintptr_t)dst) & 15) == 0) store4<TD>(d, v);
        else for (int e = 0; e < 4; ++e) if (c + e < it.cols) d[e] = from_f32<TD>(v[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[tr + 16 * i][tc4 + e] = v[e];
    }
  }
  if (it.transpose != 1) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + tr + 16 * i, r = r0 + tc4;      // output row = source column
    if (c < it.cols) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tile[tc4 + e][tr + 16 * i];
      TD* d = dst + (int64_t)c * it.ld + it.row0 + r;
      if (r + 4 <= it.rows && (it.ld & 3) == 0 && (it.row0 & 3) == 0 && (((uintptr_t)dst) & 15) == 0) store4<TD>(d, v);
      else for (int e = 0; e < 4; ++e) if (r + e < it.rows) d[e] = from_f32<TD>(v[e]);
    }
  }
}

// out[l][d] (+)= sum_b x[(b*L + l)][d].  One 64-lane column group per workgroup; its four waves take every fourth
// batch element with four loads in flight each, then reduce through LDS (fixed order).
template <typename T>
__global__ __launch_bounds__(256) void rowmod_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M, int D,
                                                          int64_t L, int accumulate) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t n4 = L * (D / 4);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  const int64_t l = i / (D / 4);
  const int col = (int)(i % (D / 4)) * 4;
  if (i < n4) {
    const int64_t nb = M / L;
    int64_t b = wave;
    for (; b + 12 < nb; b += 16) {
      s0 += load4<T>(x + ((b) * L + l) * D + col);
      s1 += load4<T>(x + ((b + 4) * L + l) * D + col);
      s2 += load4<T>(x + ((b + 8) * L + l) * D + col);
      s3 += load4<T>(x + ((b + 12) * L + l) * D + col);
    }
    for (; b < nb; b += 4) s0 += load4<T>(x + (b * L + l) * D + col);
  }
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && i < n4) {
    f32x4 s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float* dst = out + l * D + col;
    if (accumulate) s += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = s;
  }
}

// ---------------- AdamW --------------------------------------------------------------------------
template <typename TL>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, float gscale, TL* __restrict__ lowp) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int cnt = (int)min((int64_t)4, n - i4);
  float pp[4], gg[4], mm[4], vv[4];
  if (cnt == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + i4), b = *reinterpret_cast<const f32x4*>(g + i4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(m + i4), d = *reinterpret_cast<const f32x4*>(v + i4);
    for (int j = 0; j < 4; ++j) { pp[j] = a[j]; gg[j] = b[j]; mm[j] = c[j]; vv[j] = d[j]; }
  } else {
    for (int j = 0; j < cnt; ++j) { pp[j] = p[i4 + j]; gg[j] = g[i4 + j]; mm[j] = m[i4 + j]; vv[j] = v[i4 + j]; }
  }
  for (int j = 0; j < cnt; ++j) {
    const float grad = gg[j] * gscale;
    float x = pp[j] * (1.0f - lr * wd);                    // decoupled weight decay
    mm[j] = b1 * mm[j] + (1.0f - b1) * grad;               // torch: lerp / mul-add
    vv[j] = b2 * vv[j] + (1.0f - b2) * grad * grad;
    const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
    x -= (lr / bc1) * (mm[j] / denom);
    pp[j] = x;
  }
  if (cnt == 4) {
    *reinterpret_cast<f32x4*>(p + i4) = f32x4{pp[0], pp[1], pp[2], pp[3]};
    *reinterpret_cast<f32x4*>(m + i4) = f32x4{mm[0], mm[1], mm[2], mm[3]};
    *reinterpret_cast<f32x4*>(v + i4) = f32x4{vv[0], vv[1], vv[2], vv[3]};
    if (lowp) store4<TL>(lowp + i4, f32x4{pp[0], pp[1], pp[2], pp[3]});
  } else {
    for (int j = 0; j < cnt; ++j) {
      p[i4 + j] = pp[j]; m[i4 + j] = mm[j]; v[i4 + j] = vv[j];
      if (lowp) lowp[i4 + j] = from_f32<TL>(pp[j]);
    }
  }
}

inline uint32_t nblk(int64_t n, int t = 256) { return (uint32_t)((n + t - 1) / t); }
}  // namespace

extern "C" int dl_token_gate_fwd(const void* v, const void* logits, void* out, float* gate_out, int64_t B,
                                 int64_t L, int64_t D, int32_t H, int32_t add_residual, int32_t dtype,
                                 dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(v && logits && out && gate_out, DL_ERR_ARG, "dl_token_gate_fwd: null pointer");
  DL_CHECK_ARG(B > 0 && L > 0 && H > 0 && D % H == 0 && (D / H) % 4 == 0, DL_ERR_SHAPE,
               "dl_token_gate_fwd: need D %% H == 0 and (D/H) %% 4 == 0");
  const int hd = (int)(D / H);
  const int64_t n4 = B * L * D / 4;
  if (dtype == DL_BF16) {
    hipLaunchKernelGGL((gate_softmax_kernel<bf16_t>), dim3((uint32_t)(B * H)), dim3(64), 0, s,
                       (const bf16_t*)logits, gate_out, (int)L, (int)H);
    hipLaunchKernelGGL((gate_apply_kernel<bf16_t>), dim3(nblk(n4)), dim3(256), 0, s, (const bf16_t*)v, gate_out,
                       (bf16_t*)out, n4, L * D, hd, (int64_t)H * L, add_residual ? 1.0f : 0.0f);
  } else {
    hipLaunchKernelGGL((gate_softmax_kernel<float>), dim3((uint32_t)(B * H)), dim3(64), 0, s,
                       (const float*)logits, gate_out, (int)L, (int)H);
    hipLaunchKernelGGL((gate_apply_kernel<float>), dim3(nblk(n4)), dim3(256), 0, s, (const float*)v, gate_out,
                       (float*)out, n4, L * D, hd, (int64_t)H * L, add_residual ? 1.0f : 0.0f);
  }
  DL_CHECK_LAUNCH("dl_token_gate_fwd");
  return DL_OK;
}

extern "C" int dl_token_gate_bwd(const void* dout, const void* v, const float* gate, void* dv, void* dlogits,
                                 int64_t B, int64_t L, int64_t D, int32_t H, int32_t add_residual, int32_t dtype,
                                 dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dout && v && gate && dv && dlogits, DL_ERR_ARG, "dl_token_gate_bwd: null pointer");
  DL_CHECK_ARG(B > 0 && L > 0 && H > 0 && D % H == 0 && (D / H) % 4 == 0 && L <= 12288, DL_ERR_SHAPE,
               "dl_token_gate_bwd: bad shape");
  const int hd = (int)(D / H);
  const float res = add_residual ? 1.0f : 0.0f;
  const int lpr = hd / 4;
  const bool rows_form = lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
  if (dtype == DL_BF16 && rows_form)
    hipLaunchKernelGGL((gate_bwd_rows_kernel<bf16_t>), dim3((uint32_t)(B * H)), dim3(64), (size_t)L * 4, s,
                       (const bf16_t*)dout, (const bf16_t*)v, gate, (bf16_t*)dv, (bf16_t*)dlogits, (int)L, (int)H,
                       hd, res);
  else if (dtype == DL_F32 && rows_form)
    hipLaunchKernelGGL((gate_bwd_rows_kernel<float>), dim3((uint32_t)(B * H)), dim3(64), (size_t)L * 4, s,
                       (const float*)dout, (const float*)v, gate, (float*)dv, (float*)dlogits, (int)L, (int)H, hd, res);
  else if (dtype == DL_BF16)
    hipLaunchKernelGGL((gate_bwd_kernel<bf16_t>), dim3((uint32_t)(B * H)), dim3(64), (size_t)L * 4, s,
                       (const bf16_t*)dout, (const bf16_t*)v, gate, (bf16_t*)dv, (bf16_t*)dlogits, (int)L, (int)H,
                       hd, res);
  else
    hipLaunchKernelGGL((gate_bwd_kernel<float>), dim3((uint32_t)(B * H)), dim3(64), (size_t)L * 4, s,
                       (const float*)dout, (const float*)v, gate, (float*)dv, (float*)dlogits, (int)L, (int)H, hd,
                       res);
  DL_CHECK_LAUNCH("dl_token_gate_bwd");
  return DL_OK;
}

extern "C" int dl_add_rowmod_dropout(const void* x, const void* pe, void* y, int64_t M, int64_t D, int64_t L,
                                     float p, uint64_t seed, const uint64_t* seed_offset, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && y && M > 0 && D > 0 && D % 4 == 0 && L > 0, DL_ERR_ARG, "dl_add_rowmod_dropout: bad args");
  DL_CHECK_ARG(p >= 0.f && p < 1.f, DL_ERR_ARG, "dl_add_rowmod_dropout: bad p");
  const uint32_t thr = p > 0.f ? dl_dropout_thr16(p) : 0u;
  const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const int64_t n4 = M * D / 4;
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((add_rowmod_dropout_kernel<bf16_t>), dim3(nblk(n4)), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)pe, (bf16_t*)y, M, (int)D, L, thr, inv, seed, seed_offset);
  else
    hipLaunchKernelGGL((add_rowmod_dropout_kernel<float>), dim3(nblk(n4)), dim3(256), 0, s, (const float*)x,
                       (const float*)pe, (float*)y, M, (int)D, L, thr, inv, seed, seed_offset);
  DL_CHECK_LAUNCH("dl_add_rowmod_dropout");
  return DL_OK;
}

extern "C" int dl_dropout_apply(const void* x, void* y, int64_t n_rows, int64_t D, int64_t ldx, int64_t ldy,
                                float p, uint64_t seed, const uint64_t* seed_offset, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && y && n_rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, DL_ERR_ARG,
               "dl_dropout_apply: bad args");
  DL_CHECK_ARG(p > 0.f && p < 1.f, DL_ERR_ARG, "dl_dropout_apply: p must be in (0,1)");
  const int64_t n4 = n_rows * D / 4;
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((dropout_apply_kernel<bf16_t>), dim3(nblk(n4)), dim3(256), 0, s, (const bf16_t*)x,
                       (bf16_t*)y, n_rows, (int)D, ldx, ldy, dl_dropout_thr16(p), 1.0f / (1.0f - p), seed, seed_offset);
  else
    hipLaunchKernelGGL((dropout_apply_kernel<float>), dim3(nblk(n4)), dim3(256), 0, s, (const float*)x, (float*)y,
                       n_rows, (int)D, ldx, ldy, dl_dropout_thr16(p), 1.0f / (1.0f - p), seed, seed_offset);
  DL_CHECK_LAUNCH("dl_dropout_apply");
  return DL_OK;
}

extern "C" int dl_fill_pool(const void* x, void* fill, void* pooled, int64_t B, int64_t S, int64_t F, int32_t site_len,
                            int32_t in_dtype, int32_t out_dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && fill && pooled && B > 0 && S > 0 && F > 0 && site_len > 0, DL_ERR_ARG, "dl_fill_pool: bad args");
  DL_CHECK_ARG(S % site_len == 0 && F % 8 == 0 && F <= 1024, DL_ERR_SHAPE,
               "dl_fill_pool: need S %% site_len == 0, F %% 8 == 0, F <= 1024");
  const int n_site = (int)(S / site_len);
  const int Fp = (int)((F + 1 + 7) / 8 * 8);
  const int64_t BJ = B * n_site;
  const uint32_t blocks = (uint32_t)((BJ + 3) / 4);
  if (in_dtype == DL_BF16 && out_dtype == DL_BF16)
    hipLaunchKernelGGL((fill_pool_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)fill,
                       (bf16_t*)pooled, BJ, n_site, site_len, (int)F, Fp);
  else if (in_dtype == DL_F32 && out_dtype == DL_F32)
    hipLaunchKernelGGL((fill_pool_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)fill,
                       (float*)pooled, BJ, n_site, site_len, (int)F, Fp);
  else if (in_dtype == DL_F32 && out_dtype == DL_BF16)
    hipLaunchKernelGGL((fill_pool_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)fill,
                       (bf16_t*)pooled, BJ, n_site, site_len, (int)F, Fp);
  else if (in_dtype == DL_BF16 && out_dtype == DL_F32)
    hipLaunchKernelGGL((fill_pool_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)fill,
                       (float*)pooled, BJ, n_site, site_len, (int)F, Fp);
  else { dl_set_error("dl_fill_pool: bad dtypes"); return DL_ERR_ARG; }
  DL_CHECK_LAUNCH("dl_fill_pool");
  return DL_OK;
}

extern "C" int dl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dy && pre && dx && n > 0 && n % 4 == 0, DL_ERR_ARG, "dl_gelu_bwd: bad args (n %% 4 == 0 required)");
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t>), dim3(nblk(n / 4)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)pre,
                       (bf16_t*)dx, n / 4);
  else
    hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3(nblk(n / 4)), dim3(256), 0, s, (const float*)dy, (const float*)pre,
                       (float*)dx, n / 4);
  DL_CHECK_LAUNCH("dl_gelu_bwd");
  return DL_OK;
}

extern "C" int dl_gate_dpre(const void* dlogits, const void* w2, const void* pre, void* dpre, int64_t M, int64_t dd, int32_t H,
                            int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dlogits && w2 && pre && dpre && M > 0 && dd > 0, DL_ERR_ARG, "dl_gate_dpre: bad args");
  DL_CHECK_ARG(H == 8 && dd % 8 == 0 && dd <= 2048, DL_ERR_UNSUPPORTED, "dl_gate_dpre: H = 8 heads, d_diff a multiple of 8 up to 2048");
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_gate_dpre: bad dtype");
  DL_CHECK_ARG((((uintptr_t)dlogits | (uintptr_t)w2 | (uintptr_t)pre | (uintptr_t)dpre) & 15) == 0, DL_ERR_ALIGN, "dl_gate_dpre: 16-byte alignment");
  const int cpr = (int)(dd / 8), rpb = 256 / cpr;
  int64_t blocks = (M + (int64_t)rpb * 8 - 1) / ((int64_t)rpb * 8);       // ~8 rows per thread: the weight registers are loaded once
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((gate_dpre_kernel<bf16_t, 8>), dim3((uint32_t)blocks), dim3(256), 0, s, (const bf16_t*)dlogits, (const bf16_t*)w2,
                       (const bf16_t*)pre, (bf16_t*)dpre, M, (int)dd);
  else
    hipLaunchKernelGGL((gate_dpre_kernel<float, 8>), dim3((uint32_t)blocks), dim3(256), 0, s, (const float*)dlogits, (const float*)w2,
                       (const float*)pre, (float*)dpre, M, (int)dd);
  DL_CHECK_LAUNCH("dl_gate_dpre");
  return DL_OK;
}

extern "C" int dl_cast(const void* src, int32_t sdt, void* dst, int32_t ddt, int64_t n, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(src && dst && n > 0, DL_ERR_ARG, "dl_cast: bad args");
  const uint32_t blocks = nblk((n + 3) / 4);
  if (sdt == DL_F32 && ddt == DL_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (sdt == DL_BF16 && ddt == DL_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (sdt == DL_F32 && ddt == DL_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (sdt == DL_BF16 && ddt == DL_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  else { dl_set_error("dl_cast: bad dtypes"); return DL_ERR_ARG; }
  DL_CHECK_LAUNCH("dl_cast");
  return DL_OK;
}

extern "C" int dl_interleave_streams(const void* src, void* dst, int64_t R, int64_t row_bytes, int32_t S, int32_t inverse,
                                     dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(src && dst && R > 0 && S > 0 && row_bytes > 0, DL_ERR_ARG, "dl_interleave_streams: bad args");
  DL_CHECK_ARG(row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, DL_ERR_ALIGN,
               "dl_interleave_streams: rows must be whole 16-byte chunks and 16-byte aligned");
  const int cpr = (int)(row_bytes / 16);
  const int64_t total = R * S * cpr;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(interleave_streams_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, R, cpr,
                     (int)S, (int)inverse, total);
  DL_CHECK_LAUNCH("dl_interleave_streams");
  return DL_OK;
}

extern "C" int dl_norm_adjacency(const float* adj, void* ahat, int64_t B, int32_t n, int32_t out_dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(adj && ahat && B > 0 && n > 0, DL_ERR_ARG, "dl_norm_adjacency: bad args");
  const size_t lds = ((size_t)n * (n + 1) + 2 * (size_t)n) * sizeof(float);
  DL_CHECK_ARG(n <= 4096, DL_ERR_SHAPE, "dl_norm_adjacency: n=%d (at most 4096 nodes per graph)", n);
  if (lds > 150 * 1024) {                                     // n > 190: the tile does not fit LDS
    const size_t dl = 2 * (size_t)n * sizeof(float);
    if (out_dtype == DL_BF16) hipLaunchKernelGGL((norm_adj_any_kernel<bf16_t>), dim3((uint32_t)B), dim3(256), dl, s, adj, (bf16_t*)ahat, (int)n);
    else if (out_dtype == DL_F32) hipLaunchKernelGGL((norm_adj_any_kernel<float>), dim3((uint32_t)B), dim3(256), dl, s, adj, (float*)ahat, (int)n);
    else { dl_set_error("dl_norm_adjacency: bad out_dtype"); return DL_ERR_ARG; }
    DL_CHECK_LAUNCH("dl_norm_adjacency");
    return DL_OK;
  }
  if (out_dtype == DL_BF16) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)norm_adj_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((norm_adj_kernel<bf16_t>), dim3((uint32_t)B), dim3(256), lds, s, adj, (bf16_t*)ahat, (int)n);
  } else if (out_dtype == DL_F32) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)norm_adj_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((norm_adj_kernel<float>), dim3((uint32_t)B), dim3(256), lds, s, adj, (float*)ahat, (int)n);
  }
  else { dl_set_error("dl_norm_adjacency: bad out_dtype"); return DL_ERR_ARG; }
  DL_CHECK_LAUNCH("dl_norm_adjacency");
  return DL_OK;
}

extern "C" int dl_concat2(void* a, void* b, void* cat, int64_t R, int64_t a_row_bytes, int64_t b_row_bytes, int32_t inverse,
                          dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(a && b && cat && R > 0 && a_row_bytes > 0 && b_row_bytes > 0, DL_ERR_ARG, "dl_concat2: bad args");
  DL_CHECK_ARG(a_row_bytes % 16 == 0 && b_row_bytes % 16 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)cat) & 15) == 0,
               DL_ERR_ALIGN, "dl_concat2: rows must be whole 16-byte chunks and 16-byte aligned");
  const int ca = (int)(a_row_bytes / 16), cb = (int)(b_row_bytes / 16);
  const int64_t total = R * (ca + cb);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(concat2_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, (u32x4*)a, (u32x4*)b, (u32x4*)cat, ca, cb,
                     (int)inverse, total);
  DL_CHECK_LAUNCH("dl_concat2");
  return DL_OK;
}

extern "C" int dl_gather_pad(const void* store, const int64_t* offsets, const int32_t* lengths, void* out, int64_t B,
                             int64_t S, int64_t F, int32_t repeat, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(store && offsets && lengths && out && B > 0 && S > 0 && F > 0, DL_ERR_ARG, "dl_gather_pad: bad args");
  const int es = (int)dl_dtype_size(dtype);
  DL_CHECK_ARG((F * es) % 16 == 0 && ((uintptr_t)store & 15) == 0 && ((uintptr_t)out & 15) == 0, DL_ERR_ALIGN,
               "dl_gather_pad: rows must be whole 16-byte chunks and 16-byte aligned");
  DL_CHECK_ARG(B <= 65535 && S * (F * es / 16) < (1ll << 31), DL_ERR_SHAPE, "dl_gather_pad: sample too large");
  const int64_t per_sample = S * (F * es / 16);
  int64_t bx = (per_sample + 255) / 256;
  if (bx > 64) bx = 64;
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((gather_pad_kernel<bf16_t>), dim3((uint32_t)bx, (uint32_t)B), dim3(256), 0, s, (const bf16_t*)store, offsets,
                       lengths, (bf16_t*)out, (int)S, (int)F, repeat);
  else
    hipLaunchKernelGGL((gather_pad_kernel<float>), dim3((uint32_t)bx, (uint32_t)B), dim3(256), 0, s, (const float*)store, offsets,
                       lengths, (float*)out, (int)S, (int)F, repeat);
  DL_CHECK_LAUNCH("dl_gather_pad");
  return DL_OK;
}

extern "C" int dl_embed_pad(const int64_t* ids, const void* weight, const void* fill, void* out, int64_t B, int64_t L,
                            int32_t V, int32_t D, int32_t halo, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(ids && weight && fill && out && B > 0 && L > 0 && V > 0 && D > 0 && halo >= 0, DL_ERR_ARG, "dl_embed_pad: bad args");
  DL_CHECK_ARG((D + 1) % 8 == 0 && (size_t)V * D * dl_dtype_size(dtype) <= 64 * 1024, DL_ERR_SHAPE,
               "dl_embed_pad: needs (D + 1) %% 8 == 0 and a table that fits 64 KB of LDS");
  DL_CHECK_ARG(B <= 65535, DL_ERR_SHAPE, "dl_embed_pad: B must fit a grid dimension");
  const int64_t per_sample = (L + 2 * halo) * ((D + 1) / 8);
  const int64_t bx = (per_sample + 255) / 256;
  const dim3 blocks((uint32_t)(bx < 16 ? bx : 16), (uint32_t)B);   // each workgroup builds the LDS table once: keep them few
  const size_t lds = (size_t)V * (D + 1) * dl_dtype_size(dtype);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((embed_pad_kernel<bf16_t>), dim3(blocks), dim3(256), lds, s, ids, (const bf16_t*)weight, (const bf16_t*)fill,
                       (bf16_t*)out, B, (int)L, V, D, halo);
  else
    hipLaunchKernelGGL((embed_pad_kernel<float>), dim3(blocks), dim3(256), lds, s, ids, (const float*)weight, (const float*)fill,
                       (float*)out, B, (int)L, V, D, halo);
  DL_CHECK_LAUNCH("dl_embed_pad");
  return DL_OK;
}

static int sitepool_check(const char* who, const void* a, const void* b, int64_t B, int64_t L, int64_t C, int32_t halo,
                          int32_t site_len, int32_t dtype) {
  DL_CHECK_ARG(a && b && B > 0 && L > 0 && C > 0 && halo >= 0 && site_len > 0, DL_ERR_ARG, "%s: bad args", who);
  DL_CHECK_ARG(L % site_len == 0 && C % 8 == 0 && dtype == DL_BF16, DL_ERR_SHAPE,
               "%s: needs L %% site_len == 0, C %% 8 == 0, bf16 (the fp32 pipelines take the torch formulation)", who);
  DL_CHECK_ARG((int64_t)site_len * C * SP_PITCH * 2 <= 160 * 1024 && B <= 65535, DL_ERR_SHAPE, "%s: strip does not fit LDS", who);
  return DL_OK;
}
extern "C" int dl_cnn_sitepool_fwd(const void* z, void* pooled, int64_t B, int64_t L, int64_t C, int32_t halo,
                                   int32_t site_len, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = sitepool_check("dl_cnn_sitepool_fwd", z, pooled, B, L, C, halo, site_len, dtype);
  if (rc != DL_OK) return rc;
  const int n_site = (int)(L / site_len);
#ifndef DL_NO_TRANSPOSE_VIEW      /* (variant builds of tools/: same-box A/B against the general kernel) */
  if (site_len == 1 && L % 64 == 0 && C % 64 == 0 && ((uintptr_t)z & 15) == 0 && ((uintptr_t)pooled & 15) == 0) {
    // the view reinterpretation alone: pooled[b] as [C][L] is the transpose of z[b]'s L body rows
    hipLaunchKernelGGL(transpose_view_kernel, dim3((uint32_t)(L / 64), (uint32_t)(C / 64), (uint32_t)B), dim3(256), 0, s,
                       (const bf16_t*)z + (int64_t)halo * C, (int64_t)(L + 2 * halo) * C, (int64_t)C, (bf16_t*)pooled, (int64_t)L * C, (int64_t)L,
                       (bf16_t*)nullptr, (int64_t)0, 0, (int64_t)0, 0);
    DL_CHECK_LAUNCH("dl_cnn_sitepool_fwd");
    return DL_OK;
  }
#endif
  const size_t lds = (size_t)site_len * C * SP_PITCH * 2;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)cnn_sitepool_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((cnn_sitepool_fwd_kernel<bf16_t>), dim3((uint32_t)((n_site + SP_RR - 1) / SP_RR), (uint32_t)B), dim3(256), lds, s,
                     (const bf16_t*)z, (bf16_t*)pooled, (int)L, (int)C, (int)halo, (int)site_len, (const int32_t*)nullptr);
  DL_CHECK_LAUNCH("dl_cnn_sitepool_fwd");
  return DL_OK;
}
extern "C" int dl_cnn_sitepool_rows_fwd(const void* z, const int32_t* row_of, void* pooled, int64_t B, int64_t L, int64_t C,
                                        int32_t site_len, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = sitepool_check("dl_cnn_sitepool_rows_fwd", z, pooled, B, L, C, 0, site_len, dtype);
  if (rc != DL_OK) return rc;
  DL_CHECK_ARG(row_of && B * L < (1ll << 31), DL_ERR_ARG, "dl_cnn_sitepool_rows_fwd: row map");
  const int n_site = (int)(L / site_len);
  const size_t lds = (size_t)site_len * C * SP_PITCH * 2;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)cnn_sitepool_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((cnn_sitepool_fwd_kernel<bf16_t>), dim3((uint32_t)((n_site + SP_RR - 1) / SP_RR), (uint32_t)B), dim3(256), lds, s,
                     (const bf16_t*)z, (bf16_t*)pooled, (int)L, (int)C, 0, (int)site_len, row_of);
  DL_CHECK_LAUNCH("dl_cnn_sitepool_rows_fwd");
  return DL_OK;
}

// Backward through the row map: dz[r][c] = 1/S sum_{k < count} dpooled[b][n_site * ((c S + l_k / n_site) % C) + l_k % n_site] over the
// positions l_k = pos0 + k * stride the compact row stands for (rep = (first = b L + pos0, stride, count); count 0: a zero
// row).  One workgroup per (sample, slice of its compact rows): the sample's dpooled (n_site x C bf16 = 64 KB) sits in LDS,
// every thread owns one channel pair of one row and walks the row's positions.  Rows of sample b are [row_lo[b], row_lo[b+1]).
namespace {
constexpr int SPB_CHUNK = 512;   // compact rows whose (first, stride, count) triples are staged in LDS at a time
constexpr int SPB_DEEP = 48;     // a row standing for more positions than this is walked by all four waves
// CT / NT: compile-time C and n_site (0 = run-time values): the model's shape (128 channels, 256 sites) turns every index
// division of the walk into a shift — the kernel is bound by exactly that integer work
template <int CT, int NT>
__global__ __launch_bounds__(256) void cnn_sitepool_rows_bwd_kernel(const bf16_t* __restrict__ dpooled, const int32_t* __restrict__ rep,
                                                                     const int32_t* __restrict__ row_of, bf16_t* __restrict__ dz,
                                                                     int L, int C_rt, int S, int R, int slices) {
  const int C = CT ? CT : C_rt;
  extern __shared__ __attribute__((aligned(16))) char spb_smem[];
  // LDS image g[q][r] of the sample's pooled gradient, row pitch n_site + 2 elements (an ODD number of dwords): the 64 lanes
  // of a wave read 64 different q (stride 2 S) at one r — with the dense pitch of n_site = 256 elements every lane hit the
  // same bank
  bf16_t* g = reinterpret_cast<bf16_t*>(spb_smem);
  const int n_site = NT ? NT : L / S, pitch = n_site + 2;
  const int b = blockIdx.y, tid = threadIdx.x;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(dpooled + (int64_t)b * n_site * C);
    uint32_t* dst = reinterpret_cast<uint32_t*>(spb_smem);
    const int cpr = n_site / 8, total = cpr * C;                    // 16-byte chunks per row (n_site % 8 == 0: checked by the caller)
    for (int i0 = tid; i0 < total; i0 += 256 * 8) {                 // eight 16-byte loads in flight per thread, then the LDS writes
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 256 < total) v[u] = src[i0 + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < total) {
          const int q = i / cpr, d = q * (cpr * 4 + 1) + (i - q * cpr) * 4;
          dst[d] = v[u][0]; dst[d + 1] = v[u][1]; dst[d + 2] = v[u][2]; dst[d + 3] = v[u][3];
        }
      }
    }
  }
  __syncthreads();
  // the sample's compact rows: from the row of its position 0 to the row of the next sample's position 0 (rows are laid out
  // sample by sample, segment A of a sample first); the rows in between that belong to halos have count 0
  const int lo = row_of[(int64_t)b * L] - 4 < 0 ? 0 : row_of[(int64_t)b * L] - 4;
  // (the last sample ends 4 halo rows after the row of its last position; the bucket's padding rows behind it — up to 2047,
  //  several samples' worth — are zeroed by all workgroups together instead of landing on the last sample's workgroups)
  const int last_end = min(R, row_of[(int64_t)gridDim.y * L - 1] + 5);
  const int hi = (b + 1 < (int)gridDim.y) ? row_of[(int64_t)(b + 1) * L] - 4 : last_end;
  {
    const int64_t n16 = (int64_t)(R - last_end) * C / 8, nwg = (int64_t)gridDim.x * gridDim.y, me = (int64_t)b * gridDim.x + blockIdx.x;
    const int64_t share = (n16 + nwg - 1) / nwg, z0 = me * share, z1 = z0 + share < n16 ? z0 + share : n16;
    u32x4* zdst = reinterpret_cast<u32x4*>(dz + (int64_t)last_end * C);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int64_t i = z0 + tid; i < z1; i += 256) zdst[i] = zero;
  }
  const int per = (hi - lo + slices - 1) / slices;
  const int r_begin = lo + blockIdx.x * per, r_end = min(hi, r_begin + per);
  const float inv = 1.0f / (float)S, inv_site = 1.0f / (float)n_site;
  const int cp = C / 2;                                             // channel pairs per row
  __shared__ int32_t srep[SPB_CHUNK * 3];                           // the chunk's (first, stride, count) triples
  __shared__ int32_t deep[SPB_CHUNK];                               // rows of the chunk with a long position list
  __shared__ int32_t n_deep;
  float* part = reinterpret_cast<float*>(spb_smem + (size_t)pitch * C * 2);   // [4][C] partial sums of a deep row
  // sum of the pooled gradient over positions first + k * stride, k = k0, k0 + kstep, ... < count, for channels c and c + 1
  auto walk = [&](int first, int stride, int count, int c, int k0, int kstep, float& a0, float& a1) {
    const int q0 = (c * S) % C, q1 = ((c + 1) * S) % C;             // + kk (< S <= C): one conditional subtraction below
#pragma unroll 4
    for (int k = k0; k < count; k += kstep) {
      const int l = first + k * stride;
      const int kk = NT ? l / NT : (int)(((float)l + 0.5f) * inv_site);   // l / n_site (the float form is exact for l < 2^22)
      const int rr = l - kk * n_site;
      int qa = q0 + kk, qb = q1 + kk;
      qa -= qa >= C ? C : 0;
      qb -= qb >= C ? C : 0;
      a0 += (float)g[qa * pitch + rr];
      a1 += (float)g[qb * pitch + rr];
    }
  };
  for (int chunk = r_begin; chunk < r_end; chunk += SPB_CHUNK) {
    const int n = min(SPB_CHUNK, r_end - chunk);
    if (tid == 0) n_deep = 0;
    for (int i = tid; i < 3 * n; i += 256) srep[i] = rep[(int64_t)3 * chunk + i];
    __syncthreads();
    for (int i = tid; i < n; i += 256)
      if (srep[3 * i + 2] > SPB_DEEP) deep[atomicAdd(&n_deep, 1)] = i;
    for (int w = tid; w < n * cp; w += 256) {
      const int rl = w / cp, c = (w - rl * cp) * 2;
      const int count = srep[3 * rl + 2];
      if (count > SPB_DEEP) continue;                               // the whole workgroup walks those below
      float a0 = 0.f, a1 = 0.f;
      walk(srep[3 * rl] - b * L, srep[3 * rl + 1], count, c, 0, 1, a0, a1);
      *reinterpret_cast<uint32_t*>(dz + (int64_t)(chunk + rl) * C + c) = pack_bf16x2(a0 * inv, a1 * inv);
    }
    __syncthreads();
    // deep rows (the tail representative stands for up to a whole period of positions): four waves split the position list
    for (int d = 0; d < n_deep; ++d) {
      const int rl = deep[d];
      const int first = srep[3 * rl] - b * L, stride = srep[3 * rl + 1], count = srep[3 * rl + 2];
      const int wv = tid >> 6;
      for (int pr = tid & 63; pr < cp; pr += 64) {
        float a0 = 0.f, a1 = 0.f;
        walk(first, stride, count, 2 * pr, wv, 4, a0, a1);
        part[wv * C + 2 * pr] = a0;
        part[wv * C + 2 * pr + 1] = a1;
      }
      __syncthreads();
      for (int pr = tid; pr < cp; pr += 256) {
        const int c = 2 * pr;
        // (fixed order of the four partial sums: repeatable)
        const float a0 = (part[c] + part[C + c]) + (part[2 * C + c] + part[3 * C + c]);
        const float a1 = (part[c + 1] + part[C + c + 1]) + (part[2 * C + c + 1] + part[3 * C + c + 1]);
        *reinterpret_cast<uint32_t*>(dz + (int64_t)(chunk + rl) * C + c) = pack_bf16x2(a0 * inv, a1 * inv);
      }
      __syncthreads();
    }
  }
}
}  // namespace
extern "C" int dl_cnn_sitepool_rows_bwd(const void* dpooled, const int32_t* rep, const int32_t* row_of, void* dz, int64_t B, int64_t L,
                                        int64_t C, int64_t R, int32_t site_len, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(dpooled && rep && row_of && dz && B > 0 && L > 0 && C > 0 && R > 0 && site_len > 0, DL_ERR_ARG, "dl_cnn_sitepool_rows_bwd: bad args");
  DL_CHECK_ARG(L % site_len == 0 && C % 8 == 0 && dtype == DL_BF16 && (L / site_len + 2) * C * 2 + C * 16 + 16 * 1024 <= 160 * 1024 && B <= 65535 && R < (1ll << 31) &&
                   B * L < (1ll << 31) && (L / site_len) % 8 == 0 && site_len <= C, DL_ERR_SHAPE, "dl_cnn_sitepool_rows_bwd: needs bf16, L %% site_len == 0, C %% 8 == 0, one sample's pooled gradient in LDS");
  const size_t lds = (size_t)(L / site_len + 2) * C * 2 + (size_t)C * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)cnn_sitepool_rows_bwd_kernel<128, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    (void)hipFuncSetAttribute((const void*)cnn_sitepool_rows_bwd_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    attr = true;
  }
  const int slices = B >= 128 ? 4 : (B >= 32 ? 8 : 16);          // workgroups per sample: balance (samples differ 10x in rows) and fill the chip at small batches
  if (C == 128 && L / site_len == 256)
    hipLaunchKernelGGL((cnn_sitepool_rows_bwd_kernel<128, 256>), dim3((uint32_t)slices, (uint32_t)B), dim3(256), lds, s, (const bf16_t*)dpooled, rep,
                       row_of, (bf16_t*)dz, (int)L, (int)C, (int)site_len, (int)R, slices);
  else
    hipLaunchKernelGGL((cnn_sitepool_rows_bwd_kernel<0, 0>), dim3((uint32_t)slices, (uint32_t)B), dim3(256), lds, s, (const bf16_t*)dpooled, rep,
                       row_of, (bf16_t*)dz, (int)L, (int)C, (int)site_len, (int)R, slices);
  DL_CHECK_LAUNCH("dl_cnn_sitepool_rows_bwd");
  return DL_OK;
}
extern "C" int dl_cnn_sitepool_bwd(const void* dpooled, void* dz, int64_t B, int64_t L, int64_t C, int32_t halo,
                                   int32_t site_len, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = sitepool_check("dl_cnn_sitepool_bwd", dpooled, dz, B, L, C, halo, site_len, dtype);
  if (rc != DL_OK) return rc;
  const int n_site = (int)(L / site_len);
#ifndef DL_NO_TRANSPOSE_VIEW
  if (site_len == 1 && L % 64 == 0 && C % 64 == 0 && ((uintptr_t)dz & 15) == 0 && ((uintptr_t)dpooled & 15) == 0) {
    hipLaunchKernelGGL(transpose_view_kernel, dim3((uint32_t)(C / 64), (uint32_t)(L / 64), (uint32_t)B), dim3(256), 0, s,
                       (const bf16_t*)dpooled, (int64_t)L * C, (int64_t)L, (bf16_t*)dz + (int64_t)halo * C, (int64_t)(L + 2 * halo) * C, (int64_t)C,
                       halo > 0 ? (bf16_t*)dz : (bf16_t*)nullptr, (int64_t)(L + 2 * halo) * C, (int)halo, (int64_t)L, (int)(C / 8));
    DL_CHECK_LAUNCH("dl_cnn_sitepool_bwd");
    return DL_OK;
  }
#endif
  const size_t lds = (size_t)C * SP_RR * sizeof(float);
  hipLaunchKernelGGL((cnn_sitepool_bwd_kernel<bf16_t>), dim3((uint32_t)((n_site + SP_RR - 1) / SP_RR), (uint32_t)B), dim3(256), lds, s,
                     (const bf16_t*)dpooled, (bf16_t*)dz, (int)L, (int)C, (int)halo, (int)site_len);
  DL_CHECK_LAUNCH("dl_cnn_sitepool_bwd");
  return DL_OK;
}

extern "C" int dl_weight_prep(const dl_wprep_item* items_dev, const int32_t* block_map_dev, int32_t n_blocks,
                              int32_t out_dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(items_dev && block_map_dev && n_blocks > 0, DL_ERR_ARG, "dl_weight_prep: bad args");
  if (out_dtype == DL_BF16)
    hipLaunchKernelGGL((weight_prep_kernel<bf16_t>), dim3((uint32_t)n_blocks), dim3(256), 0, s, items_dev, block_map_dev);
  else if (out_dtype == DL_F32)
    hipLaunchKernelGGL((weight_prep_kernel<float>), dim3((uint32_t)n_blocks), dim3(256), 0, s, items_dev, block_map_dev);
  else { dl_set_error("dl_weight_prep: bad out_dtype"); return DL_ERR_ARG; }
  DL_CHECK_LAUNCH("dl_weight_prep");
  return DL_OK;
}

extern "C" int dl_rowmod_sum(const void* x, float* out, int64_t M, int64_t D, int64_t L, int32_t accumulate,
                             int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && out && M > 0 && D > 0 && D % 4 == 0 && L > 0 && M % L == 0, DL_ERR_ARG, "dl_rowmod_sum: bad args");
  const int64_t n4 = L * D / 4;
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((rowmod_sum_kernel<bf16_t>), dim3(nblk(n4, 64)), dim3(256), 0, s, (const bf16_t*)x, out, M,
                       (int)D, L, accumulate);
  else
    hipLaunchKernelGGL((rowmod_sum_kernel<float>), dim3(nblk(n4, 64)), dim3(256), 0, s, (const float*)x, out, M,
                       (int)D, L, accumulate);
  DL_CHECK_LAUNCH("dl_rowmod_sum");
  return DL_OK;
}

extern "C" int dl_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             float grad_scale, void* param_lowp, int32_t lowp_dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, DL_ERR_ARG, "dl_adamw_step: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const uint32_t blocks = nblk((n + 3) / 4);
  if (param_lowp && lowp_dtype == DL_BF16)
    hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, (bf16_t*)param_lowp);
  else
    hipLaunchKernelGGL((adamw_kernel<float>), dim3(blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, (float*)param_lowp);
  DL_CHECK_LAUNCH("dl_adamw_step");
  return DL_OK;
}

// compact.hip — ProteinCNN on distinct rows (round 4): the row-table kernels around the unchanged conv GEMMs, and the
// device-side padding guards.  Host side of the tables: druglamp_amd/protein_plan.py.
//
//   dl_embed_rows        compact ProteinCNN input: out[r] = [embedding(ids[src[r]]) | fill[src[r]]], zero rows where src < 0;
//                        the same launch verifies, per sample, that ids and fill bits really are periodic with the period the
//                        tables were built for (and constant behind the last whole period) — else flag bit DL_FLAG_PROT_PERIOD
//   dl_rows_gather       out[i] = src[index[i]] (zeros where index < 0): compact output rows -> all positions
//   dl_rows_sum_strided  out[r] = sum_{k < count[r]} x[first[r] + k * stride[r]]: the gather's backward, fixed summation order
//   dl_rows_equal_check  rows row0.. of every sample equal row row0 of sample 0 (bitwise) — else flag bit `code`
//                        (the drug branch's "identical padding rows" contract)
//   dl_protein_plan_build  the row tables themselves (src, w, rep, row_of, period) from the batch's residue counts, on the
//                        device (round 5): the host hands over B integers instead of building and copying ~6 MB of tables
//                        per batch (protein_plan.ProteinPlan stays as the host statement of the same tables; tests compare)
#include "common.cuh"

namespace {
__device__ __forceinline__ uint32_t bits_of(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t bits_of(bf16_t v) { return (uint32_t)__builtin_bit_cast(uint16_t, v); }

// ---- compact input -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ ids, const T* __restrict__ weight,
                                                          const T* __restrict__ fill, const int32_t* __restrict__ src,
                                                          T* __restrict__ out, int R, int V, int D, uint32_t gather_blocks,
                                                          const int32_t* __restrict__ period, int L, uint32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char er_smem[];
  const int tid = threadIdx.x;
  if (blockIdx.x >= gather_blocks) {
    // guard: sample b = blockIdx.x - gather_blocks.  sym(t) = (id, fill bit); the tables assume sym(t) == sym(t + P) while
    // t + P < E = (L / P) * P, and one constant symbol on [E, L).
    const int b = (int)(blockIdx.x - gather_blocks);
    const int P = period[b];
    const int64_t* idb = ids + (int64_t)b * L;
    const T* fb = fill + (int64_t)b * L;
    // P == 0: the tables keep every position of this sample (plain layout, protein_plan._segments: fewer than two whole
    // periods fit) — nothing is assumed about it, nothing to check
    bool bad = P < 0 || P > L;
    if (!bad && P > 0) {
      const int E = (L / P) * P;
      for (int t = tid; t < L; t += 256) {
        if (t + P < E) bad |= idb[t] != idb[t + P] || bits_of(fb[t]) != bits_of(fb[t + P]);
        if (t >= E) bad |= idb[t] != idb[L - 1] || bits_of(fb[t]) != bits_of(fb[L - 1]);
      }
    }
    if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicOr(flags, (uint32_t)DL_FLAG_PROT_PERIOD);
    return;
  }
  T* tab = reinterpret_cast<T*>(er_smem);                    // [V][C], rows padded to C = D + 1 (weight arrives that way)
  const int C = D + 1, cpr = C / 8;
  for (int i = tid; i < V * C * (int)sizeof(T) / 16; i += 256)
    reinterpret_cast<u32x4*>(er_smem)[i] = reinterpret_cast<const u32x4*>(weight)[i];
  __syncthreads();
  const int total = R * cpr;
  for (int ci = blockIdx.x * 256 + tid; ci < total; ci += gather_blocks * 256) {
    const int r = ci / cpr, ch = ci - r * cpr;
    const int sidx = src[r];
    T v[8];
    if (sidx < 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    } else {
      int64_t id = ids[sidx];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      const T* s = tab + id * C + ch * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s[e];
      if (ch == cpr - 1) v[7] = fill[sidx];                  // the last column is the fill bit
    }
    T* dst = out + (int64_t)r * C + ch * 8;
    if constexpr (sizeof(T) == 2) {
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pk[e] = (uint32_t)__builtin_bit_cast(uint16_t, v[2 * e]) | ((uint32_t)__builtin_bit_cast(uint16_t, v[2 * e + 1]) << 16);
      *reinterpret_cast<u32x4*>(dst) = pk;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = v[e];
    }
  }
}

// ---- row gather (16-byte chunks) ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_gather_kernel(const u32x4* __restrict__ src, const int32_t* __restrict__ index,
                                                           u32x4* __restrict__ out, int64_t total, int cpr) {
  for (int64_t ci = (int64_t)blockIdx.x * 256 + threadIdx.x; ci < total; ci += (int64_t)gridDim.x * 256) {
    const int64_t i = ci / cpr;
    const int ch = (int)(ci - i * cpr);
    const int r = index[i];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= 0) v = src[(int64_t)r * cpr + ch];
    out[ci] = v;
  }
}

// ---- strided row sums: one wave per output row; lane -> (8-column chunk ch = lane % CPR, k-group g = lane / CPR) -----------
// bf16, C in {64, 128, 256, 512}: CPR = C / 8 chunks per row, G = 64 / CPR k-groups, each group four rows in flight.
template <int CPR>
__global__ __launch_bounds__(256) void rows_sum_wide_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ rep,
                                                             bf16_t* __restrict__ out, int R) {
  constexpr int G = 64 / CPR, C = CPR * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = lane % CPR, g = lane / CPR;
  const int nw = gridDim.x * 4;
  for (int r = blockIdx.x * 4 + wave; r < R; r += nw) {
    const int first = rep[3 * r], stride = rep[3 * r + 1], count = rep[3 * r + 2];
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int k0 = g; k0 < count; k0 += 4 * G) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * G;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (k < count) v[u] = *reinterpret_cast<const u32x4*>(x + ((int64_t)first + (int64_t)k * stride) * C + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[2 * i] += bf16lo(v[u][i]); acc[2 * i + 1] += bf16hi(v[u][i]); }
    }
    // fixed-order reduction over the G k-groups (lanes ch, ch + CPR, ...)
#pragma unroll
    for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_down(acc[e], off, 64);
    if (g == 0) {
      const u32x4 o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
      *reinterpret_cast<u32x4*>(out + (int64_t)r * C + ch * 8) = o;
    }
  }
}

// general form (fp32, other widths): one thread per (row, 4 columns)
template <typename T>
__global__ void rows_sum_kernel(const T* __restrict__ x, const int32_t* __restrict__ rep, T* __restrict__ out, int64_t R, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  if (i >= R * c4) return;
  const int64_t r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  const int first = rep[3 * r], stride = rep[3 * r + 1], count = rep[3 * r + 2];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < count; ++k) s += load4<T>(x + ((int64_t)first + (int64_t)k * stride) * C + c);
  store4<T>(out + r * C + c, s);
}

// ---- "identical padding rows" guard ---------------------------------------------------------------------------------------
// x [B][N][row]: every row r >= row0 of every sample must equal row row0 of sample 0 bit for bit.  W = uint32_t (any row of
// 4-byte multiples) or u32x4 (16-byte multiples, aligned).
__device__ __forceinline__ bool differs(uint32_t a, uint32_t b) { return a != b; }
__device__ __forceinline__ bool differs(u32x4 a, u32x4 b) { return a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]; }
template <typename W>
__global__ __launch_bounds__(256) void rows_equal_check_kernel(const W* __restrict__ x, int64_t B, int N, int cpr, int row0,
                                                                uint32_t code, uint32_t* __restrict__ flags) {
  const int tail = N - row0;
  const int64_t total = B * tail * cpr;
  bool bad = false;
  for (int64_t ci = (int64_t)blockIdx.x * 256 + threadIdx.x; ci < total; ci += (int64_t)gridDim.x * 256) {
    const int ch = (int)(ci % cpr);
    const int64_t rr = ci / cpr;
    const int64_t b = rr / tail;
    const int r = row0 + (int)(rr - b * tail);
    bad |= differs(x[(b * N + r) * cpr + ch], x[(int64_t)row0 * cpr + ch]);
  }
  if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) atomicOr(flags, code);
}
// ---- row tables of the compact ProteinCNN layout, built on the device ------------------------------------------------------
// One sample with Lr residues in a sequence of S positions (protein_plan.py states the same rules on the host):
//   P = Lr + 2, E = (S / P) * P.  mode 0 (plain): every position its own row, S + 8 rows.
//   mode 1: segments A = [0, P + 14], B = [E - 15, E + 15], C = [S - 15, S - 1]   (P + 85 rows with 4 halo rows per side each)
//   mode 2: segments A, BC = [E - 15, S - 1]                                     (P + S - E + 46 rows)
constexpr int PL_HALO = 4, PL_RFL = 7, PL_RFR = 8;
struct PlanSample { int P, E, mode, rows; };
__device__ __forceinline__ PlanSample plan_sample(int Lr, int S) {
  PlanSample p;
  p.P = Lr + 2;
  const int reps = p.P > 0 ? S / p.P : 0;
  p.E = reps * p.P;
  const bool plain = reps < 2 || p.P + 2 * PL_RFL + 2 * PL_RFR + 40 >= S || p.P + PL_RFL - 1 + PL_RFR >= p.E - PL_RFR - PL_RFL;
  p.mode = plain ? 0 : (S - p.E > 2 * (PL_RFL + PL_RFR) + 2 ? 1 : 2);
  p.rows = plain ? S + 2 * PL_HALO : (p.mode == 1 ? p.P + 85 : p.P + S - p.E + 46);
  return p;
}

__device__ __forceinline__ int block_sum_256(int v, int* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void plan_build_kernel(const int32_t* __restrict__ lengths, int B, int S, int R,
                                                          int32_t* __restrict__ src, float* __restrict__ w, int32_t* __restrict__ rep,
                                                          int32_t* __restrict__ row_of, int32_t* __restrict__ period,
                                                          uint32_t* __restrict__ flags) {
  __shared__ int red[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  // rows in front of this sample (blocks >= B: all rows) — B is a few hundred, every block sums for itself
  int part = 0;
  const int upto = b < B ? b : B;
  for (int i = tid; i < upto; i += 256) part += plan_sample(lengths[i], S).rows;
  const int off = block_sum_256(part, red);
  if (b >= B) {                                                // bucket padding behind the last sample
    if (off > R) {
      if (tid == 0 && b == B && flags) atomicOr(flags, (uint32_t)DL_FLAG_PLAN_ROWS);
      return;
    }
    const int nb = gridDim.x - B;
    for (int r = off + (b - B) * 256 + tid; r < R; r += nb * 256) {
      src[r] = -1; w[r] = -1.f; rep[3 * r] = 0; rep[3 * r + 1] = 1; rep[3 * r + 2] = 0;
    }
    return;
  }
  const PlanSample p = plan_sample(lengths[b], S);
  if (off + p.rows > R) return;                                // (the padding blocks raise the flag)
  const int P = p.P, E = p.E, base = b * S;
  if (tid == 0) period[b] = p.mode == 0 ? 0 : P;
  // segment table: first row, first position, number of positions, representative range
  const int nA = P + PL_RFL + PL_RFR;                          // positions 0 .. P + 14
  const int oB = nA + 2 * PL_HALO, oC = oB + 31 + 2 * PL_HALO;
  for (int j = tid; j < p.rows; j += 256) {
    int pos = -1, count = 0, stride = 1;
    if (p.mode == 0) {
      if (j >= PL_HALO && j < PL_HALO + S) { pos = j - PL_HALO; count = 1; }
    } else if (j < oB) {                                        // A
      if (j >= PL_HALO && j < PL_HALO + nA) {
        pos = j - PL_HALO;
        if (pos <= P + PL_RFL - 1) {
          count = 1;
          if (pos >= PL_RFL) { const int k = (E - PL_RFR - 1 - pos) / P + 1; count = k > 1 ? k : 1; stride = P; }
        }
      }
    } else if (p.mode == 2) {                                   // B and C merged: [E - 15, S - 1], representatives E - 8 ..
      const int n = S - E + 15, q = j - oB - PL_HALO;
      if (q >= 0 && q < n) { pos = E - 15 + q; count = pos >= E - PL_RFR ? 1 : 0; }
    } else if (j < oC) {                                        // B: representatives E - 8 .. E + 7, E + 7 = the deep tail
      const int q = j - oB - PL_HALO;
      if (q >= 0 && q < 31) {
        pos = E - 15 + q;
        count = (pos >= E - PL_RFR && pos <= E + PL_RFL) ? 1 : 0;
        if (pos == E + PL_RFL) { const int c = S - PL_RFR - (E + PL_RFL); count = c > 1 ? c : 1; }
      }
    } else {                                                    // C: representatives S - 8 .. S - 1
      const int q = j - oC - PL_HALO;
      if (q >= 0 && q < 15) { pos = S - 15 + q; count = pos >= S - PL_RFR ? 1 : 0; }
    }
    const int r = off + j;
    src[r] = pos < 0 ? -1 : base + pos;
    w[r] = pos < 0 ? -1.f : (float)count;
    rep[3 * r] = base + (pos < 0 ? 0 : pos);
    rep[3 * r + 1] = stride;
    rep[3 * r + 2] = count;
  }
  for (int t = tid; t < S; t += 256) {
    int j;
    if (p.mode == 0 || t <= P + PL_RFL - 1) j = PL_HALO + t;
    else if (t <= E - PL_RFR - 1) j = PL_HALO + PL_RFL + (t - PL_RFL) % P;           // a copy of representative 7 + (t - 7) mod P
    else if (p.mode == 2 || t <= E + PL_RFL) j = oB + PL_HALO + (t - (E - 15));
    else if (t <= S - PL_RFR - 1) j = oB + PL_HALO + 22;                               // the deep tail: representative E + 7
    else j = oC + PL_HALO + (t - (S - 15));
    row_of[base + t] = off + j;
  }
}
}  // namespace

extern "C" int dl_embed_rows(const int64_t* ids, const void* weight, const void* fill, const int32_t* src, void* out, int64_t R,
                             int32_t V, int32_t D, const int32_t* period, int64_t B, int64_t L, uint32_t* flags, int32_t dtype,
                             dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(ids && weight && fill && src && out && R > 0 && V > 0 && D > 0, DL_ERR_ARG, "dl_embed_rows: bad args");
  DL_CHECK_ARG((D + 1) % 8 == 0 && (size_t)V * (D + 1) * dl_dtype_size(dtype) <= 64 * 1024, DL_ERR_SHAPE,
               "dl_embed_rows: needs (D + 1) %% 8 == 0 and a table that fits 64 KB of LDS");
  DL_CHECK_ARG(R * ((D + 1) / 8) < (1ll << 31) && B * L < (1ll << 31), DL_ERR_SHAPE, "dl_embed_rows: index range");
  DL_CHECK_ARG(!period || (flags && B > 0 && L > 0), DL_ERR_ARG, "dl_embed_rows: the periodicity guard needs flags, B, L");
  const int64_t chunks = R * ((D + 1) / 8);
  int64_t gb = (chunks + 256 * 8 - 1) / (256 * 8);             // ~8 chunks per thread: the LDS table is built once per workgroup
  if (gb < 1) gb = 1;
  if (gb > 2048) gb = 2048;
  const uint32_t nblocks = (uint32_t)gb + (period ? (uint32_t)B : 0u);
  const size_t lds = (size_t)V * (D + 1) * dl_dtype_size(dtype);
  if (dtype == DL_BF16)
    hipLaunchKernelGGL((embed_rows_kernel<bf16_t>), dim3(nblocks), dim3(256), lds, s, ids, (const bf16_t*)weight, (const bf16_t*)fill, src,
                       (bf16_t*)out, (int)R, V, D, (uint32_t)gb, period, (int)L, flags);
  else
    hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(nblocks), dim3(256), lds, s, ids, (const float*)weight, (const float*)fill, src,
                       (float*)out, (int)R, V, D, (uint32_t)gb, period, (int)L, flags);
  DL_CHECK_LAUNCH("dl_embed_rows");
  return DL_OK;
}

extern "C" int dl_rows_gather(const void* src, const int32_t* index, void* out, int64_t N, int64_t row_bytes, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(src && index && out && N > 0 && row_bytes > 0 && row_bytes % 16 == 0, DL_ERR_ARG, "dl_rows_gather: bad args (rows of 16-byte multiples)");
  DL_CHECK_ARG((((uintptr_t)src | (uintptr_t)out) & 15) == 0, DL_ERR_ALIGN, "dl_rows_gather: 16-byte alignment");
  const int cpr = (int)(row_bytes / 16);
  const int64_t total = N * cpr;
  int64_t blocks = (total + 256 * 4 - 1) / (256 * 4);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4*)src, index, (u32x4*)out, total, cpr);
  DL_CHECK_LAUNCH("dl_rows_gather");
  return DL_OK;
}

extern "C" int dl_rows_sum_strided(const void* x, const int32_t* rep, void* out, int64_t R, int64_t C, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && rep && out && R > 0 && C > 0 && C % 4 == 0 && R < (1ll << 31), DL_ERR_ARG, "dl_rows_sum_strided: bad args");
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_rows_sum_strided: bad dtype");
  const bool wide = dtype == DL_BF16 && (C == 64 || C == 128 || C == 256 || C == 512) && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  if (wide) {
    int64_t blocks = (R + 3) / 4;
    if (blocks > 4096) blocks = 4096;
#define DL_RS(CPR) hipLaunchKernelGGL((rows_sum_wide_kernel<CPR>), dim3((uint32_t)blocks), dim3(256), 0, s, (const bf16_t*)x, rep, (bf16_t*)out, (int)R)
    if (C == 64) DL_RS(8); else if (C == 128) DL_RS(16); else if (C == 256) DL_RS(32); else DL_RS(64);
#undef DL_RS
  } else {
    const int64_t n = R * (C / 4);
    if (dtype == DL_BF16)
      hipLaunchKernelGGL((rows_sum_kernel<bf16_t>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, rep, (bf16_t*)out, R, (int)C);
    else
      hipLaunchKernelGGL((rows_sum_kernel<float>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const float*)x, rep, (float*)out, R, (int)C);
  }
  DL_CHECK_LAUNCH("dl_rows_sum_strided");
  return DL_OK;
}

extern "C" int dl_rows_equal_check(const void* x, int64_t B, int64_t N, int64_t row_bytes, int64_t row0, uint32_t code,
                                   uint32_t* flags, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(x && flags && B > 0 && N > 0 && row0 >= 0 && row0 < N && row_bytes > 0 && row_bytes % 4 == 0 && code != 0, DL_ERR_ARG,
               "dl_rows_equal_check: bad args (rows of 4-byte multiples)");
  DL_CHECK_ARG(((uintptr_t)x & 3) == 0 && N < (1ll << 31), DL_ERR_ALIGN, "dl_rows_equal_check: 4-byte alignment");
  const bool wide = row_bytes % 16 == 0 && ((uintptr_t)x & 15) == 0;
  const int cpr = (int)(row_bytes / (wide ? 16 : 4));
  const int64_t total = B * (N - row0) * cpr;
  int64_t blocks = (total + 256 * 8 - 1) / (256 * 8);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (wide) hipLaunchKernelGGL((rows_equal_check_kernel<u32x4>), dim3((uint32_t)blocks), dim3(256), 0, s, (const u32x4*)x, B, (int)N, cpr, (int)row0, code, flags);
  else hipLaunchKernelGGL((rows_equal_check_kernel<uint32_t>), dim3((uint32_t)blocks), dim3(256), 0, s, (const uint32_t*)x, B, (int)N, cpr, (int)row0, code, flags);
  DL_CHECK_LAUNCH("dl_rows_equal_check");
  return DL_OK;
}

extern "C" int dl_protein_plan_build(const int32_t* lengths, int64_t B, int64_t S, int64_t R, int32_t* src, float* w, int32_t* rep,
                                     int32_t* row_of, int32_t* period, uint32_t* flags, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(lengths && src && w && rep && row_of && period && B > 0 && S > 0 && R > 0, DL_ERR_ARG, "dl_protein_plan_build: bad args");
  DL_CHECK_ARG(B * S < (1ll << 31) && R < (1ll << 30) && B < (1 << 20), DL_ERR_SHAPE, "dl_protein_plan_build: index range");
  int64_t pad_blocks = R / (256 * 16) + 1;
  if (pad_blocks > 64) pad_blocks = 64;
  hipLaunchKernelGGL(plan_build_kernel, dim3((uint32_t)(B + pad_blocks)), dim3(256), 0, s, lengths, (int)B, (int)S, (int)R, src, w, rep,
                     row_of, period, flags);
  DL_CHECK_LAUNCH("dl_protein_plan_build");
  return DL_OK;
}

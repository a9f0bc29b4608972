// graph.hip — neighbourhood aggregation of MolecularGCN on a batch of dense molecular graphs
// (reference: dgl GraphConv with norm='both', model/basic_model.py:545-638 — `update_all(copy_u, sum)` over the batched
//  DGL graph; DGL is absent here, parity of the GCN is UNPINNED).
//
//   out[b][i][:] = sum_j A'[b][i][j] * feat[b][j][:]   for the first n nodes (A' = ahat or ahat^T: forward / gradient)
//   out[b][i][:] = feat[b][i][:]                       for nodes n <= i < N (virtual padding nodes: self loop only)
//
// One workgroup per molecule: ahat (n <= 128 real atoms) and the n x C feature block sit in LDS as swizzled 128-wide tile
// images; each of the four waves owns 32 output rows and all C columns (8 accumulator tiles), fragments come from the
// LDS transposing read (features, and the adjacency in the gradient form) or two 8-byte reads (adjacency, forward form).
// Rows of virtual nodes are copied with 16-byte accesses.  (The first version gathered fragments element by element:
// 48 us per launch at any batch size, 0.3 ms per step; this one is bound by the launch and the 64 KB tile fill.)
#include "tiles.cuh"

namespace {
using namespace dltile;

// Both operands live in LDS as ATile<T, 128> images (128-element rows, 16-byte chunks XOR-swizzled — the layout of the
// attention kernels), zero outside the n x n / n x C blocks:
//   Fs[k][c]  = feat[b][k][c]      read as the FIRST MFMA operand F^T[c][k] through the transposing fragment read
//   As[r][q]  = ahat[b][r][q]      second operand A'^T[k][i]: row i, slots k (two 8-byte reads, CTILE slot map) for the
//                                  forward form, the transposing read for the gradient form (A' = ahat^T)
// so that D[c][i] = sum_k F^T[c][k] A'[i][k] leaves every lane with 4 consecutive columns c of one output row i.
template <typename T, int C>
__global__ __launch_bounds__(256) void graph_aggregate_kernel(const T* __restrict__ ahat, const T* __restrict__ feat, T* __restrict__ out,
                                                               int n, int N, int transpose) {
  static_assert(C == 128, "tile images are 128 elements wide");
  using TL = ATile<T, 128>;
  constexpr int NP = 128, EPC = TL::EPC, KF = Mma<T>::KF;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* As = smem_raw;                                      // [NP] rows of TL::RB bytes
  char* Fs = smem_raw + NP * TL::RB;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const T* A = ahat + (int64_t)b * n * n;
  const T* F = feat + (int64_t)b * N * C;
  T* O = out + (int64_t)b * N * C;
  const bool avec = (n % EPC) == 0 && ((uintptr_t)A & 15) == 0;
  for (int c = tid; c < NP * TL::CPR; c += 256) {           // 16-byte chunks
    const int row = c / TL::CPR, ch = c % TL::CPR, col = ch * EPC;
    u32x4 wa = {0u, 0u, 0u, 0u}, wf = {0u, 0u, 0u, 0u};
    if (row < n) {
      wf = *reinterpret_cast<const u32x4*>(F + (int64_t)row * C + col);
      if (avec) {
        if (col < n) wa = *reinterpret_cast<const u32x4*>(A + (int64_t)row * n + col);
      } else {
        T tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = (col + e < n) ? A[(int64_t)row * n + col + e] : from_f32<T>(0.f);
        wa = *reinterpret_cast<const u32x4*>(tmp);
      }
    }
    lds_write16(As, row * TL::RB + ((ch ^ TL::swz(row)) << 4), wa);
    lds_write16(Fs, row * TL::RB + ((ch ^ TL::swz(row)) << 4), wf);
  }
  // virtual padding nodes: identity
  for (int64_t e = tid; e < (int64_t)(N - n) * (C / EPC); e += 256) {
    const int64_t off = (int64_t)n * C + e * EPC;
    *reinterpret_cast<u32x4*>(O + off) = *reinterpret_cast<const u32x4*>(F + off);
  }
  __syncthreads();
  const int nk = (n + KF - 1) / KF;
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {                          // two 16-row output tiles per wave
    const int i0 = wave * 32 + it * 16;
    if (i0 >= n) break;
    f32x4 acc[C / 16];
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kf = 0; kf < nk; ++kf) {
      const int kb = kf * KF;
      u32x4 af;
      if (transpose) {
        af = frag_tr<T, 128>(As, kb, i0, il, g);            // A'[i][k] = ahat[k][i]
      } else if constexpr (sizeof(T) == 2) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(As + TL::off(i0 + il, kb + 4 * g));
        const u32x2 hi = *reinterpret_cast<const u32x2*>(As + TL::off(i0 + il, kb + 16 + 4 * g));
        af = u32x4{lo[0], lo[1], hi[0], hi[1]};
      } else {
        af = lds_read16(As, TL::off(i0 + il, kb + 4 * g));  // f32: CTILE and CONTIG slot maps coincide
      }
#pragma unroll
      for (int ct = 0; ct < C / 16; ++ct) {
        const u32x4 ff = frag_tr<T, 128>(Fs, kb, ct * 16, il, g);
        acc[ct] = Mma<T>::mma(ff, af, acc[ct]);             // lane (il, g): acc[r] = out[i0 + il][ct*16 + 4g + r]
      }
    }
    const int i = i0 + il;
    if (i < n) {
#pragma unroll
      for (int ct = 0; ct < C / 16; ++ct) store4<T>(O + (int64_t)i * C + ct * 16 + 4 * g, acc[ct]);
    }
  }
}

// Graphs with more than 128 real atoms (rare: drug-like molecules seldom pass 100 heavy atoms; MAX_NODES is 512): the same
// sums on the vector ALU straight from global memory / L2 — a workgroup takes 32 output rows, a thread 4 rows x 4
// columns, fp32 accumulation.  9-66 MFLOP per molecule: this path is bound by its launch, not by arithmetic; it exists so
// that no molecule size leaves the HIP path (round 2 handed these to torch.bmm).
template <typename T, int C>
__global__ __launch_bounds__(256) void graph_aggregate_any_kernel(const T* __restrict__ ahat, const T* __restrict__ feat, T* __restrict__ out,
                                                                   int n, int N, int transpose) {
  const int b = blockIdx.y, tid = threadIdx.x, c4 = (tid & 31) * 4, rl = tid >> 5;
  const T* A = ahat + (int64_t)b * n * n;
  const T* F = feat + (int64_t)b * N * C;
  T* O = out + (int64_t)b * N * C;
  const int r0 = blockIdx.x * 32;
  if (r0 >= n) {                                            // workgroups past the real atoms copy the virtual nodes' rows
    const int64_t first = (int64_t)n * C, total = (int64_t)(N - n) * C;
    const int64_t nb = gridDim.x - (n + 31) / 32, me = blockIdx.x - (n + 31) / 32;
    for (int64_t e = (me * 256 + tid) * 4; e < total; e += nb * 256 * 4) store4<T>(O + first + e, load4<T>(F + first + e));
    return;
  }
  f32x4 acc[4];
  int rows[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] = f32x4{0.f, 0.f, 0.f, 0.f}; rows[q] = r0 + rl + 8 * q; }
  for (int k = 0; k < n; ++k) {
    const f32x4 f = load4<T>(F + (int64_t)k * C + c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = rows[q] < n ? rows[q] : n - 1;
      const float a = to_f32(transpose ? A[(int64_t)k * n + i] : A[(int64_t)i * n + k]);
      acc[q] += f * a;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (rows[q] < n) store4<T>(O + (int64_t)rows[q] * C + c4, acc[q]);
}

}  // namespace

extern "C" int dl_graph_aggregate(const void* ahat, const void* feat, void* out, int64_t B, int32_t n, int32_t N, int32_t C,
                                  int32_t transpose, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(ahat && feat && out && B > 0 && n > 0 && N >= n, DL_ERR_ARG, "dl_graph_aggregate: bad args");
  DL_CHECK_ARG(C == 128, DL_ERR_UNSUPPORTED, "dl_graph_aggregate: feature width %d (only 128, the model's n_hidden)", C);
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_graph_aggregate: bad dtype");
  DL_CHECK_ARG((((uintptr_t)feat | (uintptr_t)out) & 15) == 0, DL_ERR_ALIGN, "dl_graph_aggregate: feat / out must be 16-byte aligned");
  DL_CHECK_ARG(B <= 0x7fffffff && (n <= 128 || B <= 65535), DL_ERR_SHAPE, "dl_graph_aggregate: batch too large");
  if (n > 128) {
    // one workgroup per 32 real rows + (if there are virtual nodes) a few that copy their rows through
    const uint32_t gx = (uint32_t)((n + 31) / 32) + (N > n ? 4u : 0u);
    if (dtype == DL_BF16)
      hipLaunchKernelGGL((graph_aggregate_any_kernel<bf16_t, 128>), dim3(gx, (uint32_t)B), dim3(256), 0, s, (const bf16_t*)ahat,
                         (const bf16_t*)feat, (bf16_t*)out, (int)n, (int)N, (int)transpose);
    else
      hipLaunchKernelGGL((graph_aggregate_any_kernel<float, 128>), dim3(gx, (uint32_t)B), dim3(256), 0, s, (const float*)ahat,
                         (const float*)feat, (float*)out, (int)n, (int)N, (int)transpose);
    DL_CHECK_LAUNCH("dl_graph_aggregate");
    return DL_OK;
  }
  if (dtype == DL_BF16) {
    const size_t lds = (size_t)2 * 128 * 128 * 2;
    (void)hipFuncSetAttribute((const void*)graph_aggregate_kernel<bf16_t, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((graph_aggregate_kernel<bf16_t, 128>), dim3((uint32_t)B), dim3(256), lds, s, (const bf16_t*)ahat,
                       (const bf16_t*)feat, (bf16_t*)out, (int)n, (int)N, (int)transpose);
  } else {
    const size_t lds = (size_t)2 * 128 * 128 * 4;
    (void)hipFuncSetAttribute((const void*)graph_aggregate_kernel<float, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((graph_aggregate_kernel<float, 128>), dim3((uint32_t)B), dim3(256), lds, s, (const float*)ahat,
                       (const float*)feat, (float*)out, (int)n, (int)N, (int)transpose);
  }
  DL_CHECK_LAUNCH("dl_graph_aggregate");
  return DL_OK;
}

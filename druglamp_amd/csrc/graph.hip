// graph.hip — neighbourhood aggregation of MolecularGCN on a batch of dense molecular graphs
// (reference: dgl GraphConv with norm='both', model/basic_model.py:545-638 — `update_all(copy_u, sum)` over the batched
//  DGL graph; DGL is absent here, parity of the GCN is UNPINNED).
//
//   out[b][i][:] = sum_j A'[b][i][j] * feat[b][j][:]   for the first n nodes (A' = ahat or ahat^T: forward / gradient)
//   out[b][i][:] = feat[b][i][:]                       for nodes n <= i < N (virtual padding nodes: self loop only)
//
// One workgroup per molecule: ahat (n <= 128 real atoms, <= 32 KB in bf16) and the n x C feature block sit in LDS; each of
// the four waves owns 32 output rows and walks the C columns in 16-wide MFMA tiles.  Fragments are gathered element by
// element from LDS (the problem is 4 MFLOP per molecule — launch-bound, not worth a transposing tile layout); rows of
// virtual nodes are copied with 16-byte accesses.
#include "common.cuh"

namespace {

template <typename T> struct GFrag;
template <> struct GFrag<bf16_t> {
  static constexpr int KF = 32, SL = 8;
  __device__ static __forceinline__ u32x4 pack(const float (&v)[8]) {
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  }
};
template <> struct GFrag<float> {
  static constexpr int KF = 16, SL = 4;
  __device__ static __forceinline__ u32x4 pack(const float (&v)[8]) {
    return __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});
  }
};

template <typename T, int C>
__global__ __launch_bounds__(256) void graph_aggregate_kernel(const T* __restrict__ ahat, const T* __restrict__ feat, T* __restrict__ out,
                                                               int n, int N, int transpose) {
  constexpr int NP = 128;                                  // padded node count of the LDS tiles
  constexpr int PA = NP + 4 / (int)sizeof(T);               // scalar accesses only: a 65 / 129-dword pitch keeps column walks conflict-free
  constexpr int PF = C + 16 / (int)sizeof(T);               // rows are written with 16-byte stores: pitch a multiple of 16 bytes
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);                  // [NP][PA], zero outside n x n
  T* Fs = As + NP * PA;                                    // [NP][PF], zero rows >= n
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 15, g = lane >> 4;
  const T* A = ahat + (int64_t)b * n * n;
  const T* F = feat + (int64_t)b * N * C;
  T* O = out + (int64_t)b * N * C;
  for (int e = tid; e < NP * NP; e += 256) {
    const int i = e / NP, j = e % NP;
    As[i * PA + j] = (i < n && j < n) ? A[(int64_t)i * n + j] : from_f32<T>(0.f);
  }
  constexpr int EPC = 16 / (int)sizeof(T);
  for (int e = tid; e < NP * (C / EPC); e += 256) {
    const int j = e / (C / EPC), c = (e % (C / EPC)) * EPC;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (j < n) w = *reinterpret_cast<const u32x4*>(F + (int64_t)j * C + c);
    *reinterpret_cast<u32x4*>(Fs + j * PF + c) = w;
  }
  // virtual padding nodes: identity
  for (int64_t e = tid; e < (int64_t)(N - n) * (C / EPC); e += 256) {
    const int64_t off = (int64_t)n * C + e * EPC;
    *reinterpret_cast<u32x4*>(O + off) = *reinterpret_cast<const u32x4*>(F + off);
  }
  __syncthreads();
  constexpr int KF = GFrag<T>::KF, SL = GFrag<T>::SL;
  const int nk = (n + KF - 1) / KF;
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {                         // two 16-row tiles per wave
    const int i0 = wave * 32 + it * 16;
    if (i0 >= n) break;
#pragma unroll 1
    for (int c0 = 0; c0 < C; c0 += 16) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int kf = 0; kf < nk; ++kf) {
        const int k0 = kf * KF + g * SL;                   // CONTIG slot map: slot s of group g <-> k0 + s
        float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SL; ++s) {
          av[s] = to_f32(transpose ? As[(k0 + s) * PA + i0 + il] : As[(i0 + il) * PA + k0 + s]);
          bv[s] = to_f32(Fs[(k0 + s) * PF + c0 + il]);
        }
        // D[i][c] = sum_k A'[i][k] feat[k][c]: first operand rows = c (feat^T), second operand columns = i  ->  D^T[c][i];
        // written the other way round so that a lane ends up with 4 CONSECUTIVE columns c of one row i:
        acc = Mma<T>::mma(GFrag<T>::pack(bv), GFrag<T>::pack(av), acc);       // lane (il, g): acc[r] = out[i0 + il][c0 + 4g + r]
      }
      const int i = i0 + il;
      if (i < n) store4<T>(O + (int64_t)i * C + c0 + 4 * g, acc);
    }
  }
}

}  // namespace

extern "C" int dl_graph_aggregate(const void* ahat, const void* feat, void* out, int64_t B, int32_t n, int32_t N, int32_t C,
                                  int32_t transpose, int32_t dtype, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(ahat && feat && out && B > 0 && n > 0 && N >= n, DL_ERR_ARG, "dl_graph_aggregate: bad args");
  DL_CHECK_ARG(n <= 128, DL_ERR_UNSUPPORTED, "dl_graph_aggregate: at most 128 real nodes per graph (got %d)", n);
  DL_CHECK_ARG(C == 128, DL_ERR_UNSUPPORTED, "dl_graph_aggregate: feature width %d (only 128, the model's n_hidden)", C);
  DL_CHECK_ARG(dtype == DL_BF16 || dtype == DL_F32, DL_ERR_ARG, "dl_graph_aggregate: bad dtype");
  DL_CHECK_ARG((((uintptr_t)feat | (uintptr_t)out) & 15) == 0, DL_ERR_ALIGN, "dl_graph_aggregate: feat / out must be 16-byte aligned");
  DL_CHECK_ARG(B <= 0x7fffffff, DL_ERR_SHAPE, "dl_graph_aggregate: batch too large");
  if (dtype == DL_BF16) {
    const size_t lds = (size_t)128 * (128 + 2) * 2 + (size_t)128 * (128 + 8) * 2;
    (void)hipFuncSetAttribute((const void*)graph_aggregate_kernel<bf16_t, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((graph_aggregate_kernel<bf16_t, 128>), dim3((uint32_t)B), dim3(256), lds, s, (const bf16_t*)ahat,
                       (const bf16_t*)feat, (bf16_t*)out, (int)n, (int)N, (int)transpose);
  } else {
    const size_t lds = (size_t)128 * (128 + 1) * 4 + (size_t)128 * (128 + 4) * 4;
    (void)hipFuncSetAttribute((const void*)graph_aggregate_kernel<float, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((graph_aggregate_kernel<float, 128>), dim3((uint32_t)B), dim3(256), lds, s, (const float*)ahat,
                       (const float*)feat, (float*)out, (int)n, (int)N, (int)transpose);
  }
  DL_CHECK_LAUNCH("dl_graph_aggregate");
  return DL_OK;
}

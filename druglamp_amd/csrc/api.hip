// api.hip — error reporting, version, per-kernel-family timing hooks.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "common.cuh"

namespace {
thread_local char g_err[512] = "";

struct ProfRec { hipEvent_t a, b; double flops, bytes; int tag; };
constexpr int NTAG = 8;           // sub-family tags of a family (dl_gemm_args.prof_tag; 0 = untagged)
struct ProfFamily {
  int period = 0;                 // 0 = off; N >= 1: HIP events around every N-th launch of the family
  int64_t seen = 0;               // launches since dl_prof_enable (timed or not), with their algorithmic work
  double all_flops = 0, all_bytes = 0;
  int64_t tag_seen[NTAG] = {};    // the same per sub-family tag
  double tag_flops[NTAG] = {}, tag_bytes[NTAG] = {};
  std::vector<ProfRec> recs;
  hipEvent_t pending = nullptr;
};
constexpr int NFAM = 8;
ProfFamily g_prof[NFAM];
std::mutex g_prof_mu;
}  // namespace

extern "C" void dl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dl_last_error(void) { return g_err; }
extern "C" int dl_version(void) { return 100; }

void dl_prof_before(int family, hipStream_t s) {
  if (family < 0 || family >= NFAM || !g_prof[family].period) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  // 1 launch in `period`, chosen by a multiplicative hash of the launch index: a plain "every period-th" would keep
  // hitting the same launch slots of the step whenever period divides the launches per step
  if (f.period > 1 && (((uint32_t)f.seen * 2654435761u) >> 12) % (uint32_t)f.period != 0) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  f.pending = e;
}

void dl_prof_after(int family, hipStream_t s, double flops, double bytes, int tag) {
  if (family < 0 || family >= NFAM || !g_prof[family].period) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  if (tag < 0 || tag >= NTAG) tag = 0;
  ++f.seen;
  f.all_flops += flops;
  f.all_bytes += bytes;
  ++f.tag_seen[tag];
  f.tag_flops[tag] += flops;
  f.tag_bytes[tag] += bytes;
  if (!f.pending) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  f.recs.push_back(ProfRec{f.pending, e, flops, bytes, tag});
  f.pending = nullptr;
}

extern "C" int dl_prof_enable(int32_t family, int32_t on) {
  DL_CHECK_ARG(family >= 0 && family < NFAM, DL_ERR_ARG, "dl_prof_enable: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  for (auto& r : f.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  f.recs.clear();
  if (f.pending) { (void)hipEventDestroy(f.pending); f.pending = nullptr; }
  f.period = on > 0 ? on : 0;
  f.seen = 0;
  f.all_flops = f.all_bytes = 0;
  for (int t = 0; t < NTAG; ++t) { f.tag_seen[t] = 0; f.tag_flops[t] = f.tag_bytes[t] = 0; }
  return DL_OK;
}

extern "C" int dl_prof_totals(int32_t family, int64_t* launches, double* total_flops, double* total_bytes) {
  DL_CHECK_ARG(family >= 0 && family < NFAM, DL_ERR_ARG, "dl_prof_totals: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  if (launches) *launches = f.seen;
  if (total_flops) *total_flops = f.all_flops;
  if (total_bytes) *total_bytes = f.all_bytes;
  return DL_OK;
}

extern "C" int dl_prof_collect(int32_t family, int64_t* launches, double* total_ms, double* total_flops,
                               double* total_bytes) {
  DL_CHECK_ARG(family >= 0 && family < NFAM, DL_ERR_ARG, "dl_prof_collect: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  double ms = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (auto& r : f.recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  return DL_OK;
}

extern "C" int dl_prof_collect_tag(int32_t family, int32_t tag, int64_t* timed_launches, double* timed_ms, double* timed_flops,
                                   double* timed_bytes, int64_t* all_launches, double* all_flops, double* all_bytes) {
  DL_CHECK_ARG(family >= 0 && family < NFAM && tag >= 0 && tag < NTAG, DL_ERR_ARG, "dl_prof_collect_tag: bad family %d / tag %d", family, tag);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  double ms = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (auto& r : f.recs) {
    if (r.tag != tag) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  if (timed_launches) *timed_launches = n;
  if (timed_ms) *timed_ms = ms;
  if (timed_flops) *timed_flops = fl;
  if (timed_bytes) *timed_bytes = by;
  if (all_launches) *all_launches = f.tag_seen[tag];
  if (all_flops) *all_flops = f.tag_flops[tag];
  if (all_bytes) *all_bytes = f.tag_bytes[tag];
  return DL_OK;
}

__global__ void dl_reduce_partials_kernel(const float* __restrict__ partial, int chunks, int64_t stride, int ncols,
                                          float* __restrict__ out, int accumulate) {
  // 1024 threads = 16 columns x 64 row-lanes (DL_REDUCE_COLS columns per workgroup -> ncols / 16 workgroups, four
  // times the parallelism of the old 64-column form whose 8..16 workgroups took 12 us on 2048-chunk LayerNorm
  // partials).  Row-lane k sums chunks k, k + 64, ... with four loads in flight, then a fixed-order LDS tree.
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, k = threadIdx.x >> 4;
  const int c = blockIdx.x * DL_REDUCE_COLS + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < ncols) {
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
  if (k == 0 && c < ncols) out[c] = accumulate ? out[c] + red[0][cl] : red[0][cl];
}

// ---- several pending second-stage reductions in one launch (dl_reduce_batch) ---------------------------------------
namespace {
struct ReduceBatch {
  dl_reduce_item items[DL_REDUCE_BATCH_MAX];
  uint32_t start[DL_REDUCE_BATCH_MAX + 1];      // first workgroup of every item; start[n] = grid size
  int32_t n;
};

// The per-element arithmetic is that of splitk_reduce_kernel (gemm.hip) and dl_reduce_partials_kernel above, so that
// a batched reduction is bit-identical to the immediate one; only the workgroup shape differs for the split-K items
// (1024 threads here).
template <typename TO>
__device__ __forceinline__ void splitk_item(const dl_reduce_item& it, uint32_t blk) {
  const int64_t mn = it.mn;
  const uint32_t main_blocks = (uint32_t)((mn / 4 + 1023) / 1024);
  const int splits = it.splits;
  if (blk >= main_blocks) {
    const int m = (int)(blk - main_blocks) * 1024 + (int)threadIdx.x;
    if (m < it.M) {
      const float* cs = it.cs_slabs;
      float a0 = 0.f, a1 = 0.f;
      int z = 0;
      for (; z + 1 < splits; z += 2) { a0 += cs[(int64_t)z * it.M + m]; a1 += cs[(int64_t)(z + 1) * it.M + m]; }
      if (z < splits) a0 += cs[(int64_t)z * it.M + m];
      it.cs_out[m] = a0 + a1;
    }
    return;
  }
  const int64_t i4 = ((int64_t)blk * 1024 + threadIdx.x) * 4;
  if (i4 >= mn) return;
  const float* slabs = it.src;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int z = 0;
  for (; z + 3 < splits; z += 4) {
    s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * mn + i4);
    s1 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 1) * mn + i4);
    s2 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 2) * mn + i4);
    s3 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 3) * mn + i4);
  }
  for (; z < splits; ++z) s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * mn + i4);
  f32x4 s = (s0 + s1) + (s2 + s3);
  const int64_t m = i4 / it.N, n = i4 % it.N;
  TO* dst = reinterpret_cast<TO*>(it.out) + m * it.ldc + n;
  if (it.accumulate) { const f32x4 o = load4<TO>(dst); s += o; }
  store4<TO>(dst, s);
}

__global__ __launch_bounds__(1024) void reduce_batch_kernel(const ReduceBatch b) {
  __shared__ float red[64][17];
  int j = 0;
  while (j + 1 < b.n && blockIdx.x >= b.start[j + 1]) ++j;
  const dl_reduce_item& it = b.items[j];
  const uint32_t blk = blockIdx.x - b.start[j];
  if (it.kind == DL_REDUCE_SPLITK) {
    if (it.out_dtype == DL_F32) splitk_item<float>(it, blk); else splitk_item<bf16_t>(it, blk);
    return;
  }
  const float* partial = it.src;
  const int chunks = it.splits, ncols = it.N;
  const int64_t stride = it.mn;
  float* out = reinterpret_cast<float*>(it.out);
  const int cl = threadIdx.x & 15, k = threadIdx.x >> 4;
  const int c = (int)blk * DL_REDUCE_COLS + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < ncols) {
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
  if (k == 0 && c < ncols) out[c] = it.accumulate ? out[c] + red[0][cl] : red[0][cl];
}
}  // namespace

extern "C" int dl_reduce_batch(const dl_reduce_item* items, int32_t n, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  DL_CHECK_ARG(items && n >= 0 && n <= DL_REDUCE_BATCH_MAX, DL_ERR_ARG, "dl_reduce_batch: n = %d items (at most %d)", n,
               (int)DL_REDUCE_BATCH_MAX);
  ReduceBatch b;
  b.n = 0;
  uint32_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    const dl_reduce_item& it = items[i];
    if (it.kind == DL_REDUCE_NONE) continue;
    DL_CHECK_ARG(it.kind == DL_REDUCE_SPLITK || it.kind == DL_REDUCE_PARTIALS, DL_ERR_ARG, "dl_reduce_batch: item %d has kind %d", i, it.kind);
    DL_CHECK_ARG(it.src && it.out && it.splits >= 1 && it.N >= 1 && it.mn >= 1, DL_ERR_ARG, "dl_reduce_batch: item %d is incomplete", i);
    uint32_t nb;
    if (it.kind == DL_REDUCE_SPLITK) {
      DL_CHECK_ARG(it.mn % 4 == 0 && it.N % 4 == 0 && (it.out_dtype == DL_F32 || it.out_dtype == DL_BF16), DL_ERR_UNSUPPORTED,
                   "dl_reduce_batch: split-K item %d needs N %% 4 == 0 and an f32 / bf16 output", i);
      DL_CHECK_ARG(it.M == 0 || (it.cs_slabs && it.cs_out), DL_ERR_ARG, "dl_reduce_batch: item %d has column sums without buffers", i);
      nb = (uint32_t)((it.mn / 4 + 1023) / 1024) + (uint32_t)((it.M + 1023) / 1024);
    } else {
      nb = (uint32_t)((it.N + DL_REDUCE_COLS - 1) / DL_REDUCE_COLS);
    }
    b.items[b.n] = it;
    b.start[b.n] = blocks;
    blocks += nb;
    ++b.n;
  }
  if (b.n == 0) return DL_OK;
  b.start[b.n] = blocks;
  hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(1024), 0, s, b);
  DL_CHECK_LAUNCH("dl_reduce_batch");
  return DL_OK;
}

// api.hip — error reporting, version, per-kernel-family timing hooks.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "common.cuh"

namespace {
thread_local char g_err[512] = "";

struct ProfRec { hipEvent_t a, b; double flops, bytes; };
struct ProfFamily {
  int period = 0;                 // 0 = off; N >= 1: HIP events around every N-th launch of the family
  int64_t seen = 0;               // launches since dl_prof_enable (timed or not), with their algorithmic work
  double all_flops = 0, all_bytes = 0;
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

void dl_prof_after(int family, hipStream_t s, double flops, double bytes) {
  if (family < 0 || family >= NFAM || !g_prof[family].period) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfFamily& f = g_prof[family];
  ++f.seen;
  f.all_flops += flops;
  f.all_bytes += bytes;
  if (!f.pending) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  f.recs.push_back(ProfRec{f.pending, e, flops, bytes});
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

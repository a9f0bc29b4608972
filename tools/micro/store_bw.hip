// Store-rate microbenchmark for the GEMM epilogue: G workgroups of 512 threads each write `tiles` output tiles of
// 256 rows x 512 bytes (a 256x256 bf16 tile of an N = 2048 output: row pitch 4096 bytes), 16 bytes per lane, 8 lanes
// per 128-byte line — the store pattern of gemm_big_kernel's epilogue — with `busy` dependent FMAs per tile between the
// store phases (a stand-in for the main loop: lets the stores of one tile drain while the workgroup "computes").
// Question: is a CU's own store rate (~20 GB/s?) or the chip's HBM write rate what an epilogue phase waits for?
// build: hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip ; run: ./store_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void store_tiles(char* __restrict__ out, int tiles, int tiles_per_row, int64_t pitch, int busy, int do_store, int wait_ack,
                                                   float* __restrict__ sink) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float f = (float)tid;
  for (int t = 0; t < tiles; ++t) {
    const int64_t tile = (int64_t)blockIdx.x + (int64_t)t * gridDim.x;
    const int64_t m0 = (tile / tiles_per_row) * 256, n0b = (tile % tiles_per_row) * 512;
    for (int i = 0; i < busy; ++i) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
    const uint32_t w = __builtin_bit_cast(uint32_t, f);
    const u32x4 v = {w, w + 1u, w + 2u, w + 3u};
    // wave (wm = wave / 4, wn = wave % 4) owns rows wm*128 .. +127, bytes wn*128 .. +127 of the tile
    const int wm = wave >> 2, wn = wave & 3;
if (do_store)
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = wm * 128 + it * 8 + (lane >> 3);
      *reinterpret_cast<u32x4*>(out + (m0 + row) * pitch + n0b + wn * 128 + (lane & 7) * 16) = v;
    }
    if (wait_ack) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // what the GEMM's next tile does before its first k-step
  }
  if (f == 12345.f) sink[0] = f;
}

int main() {
  const int64_t M = 65536, pitch = 4096;
  char* out; float* sink;
  hipMalloc(&out, M * pitch); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int total_tiles = (int)(M / 256) * 8;           // 2048 tiles of 128 KB = 268 MB
  printf("%6s %6s %8s %10s %10s %10s %12s\n", "G", "tiles", "busy", "us no st", "us", "TB/s", "GB/s per WG");
  for (int wait_ack : {0, 1})
  for (int busy : {0, 500}) {
    for (int G : {16, 64, 256}) {
      const int tiles = total_tiles / G;
      float best[2] = {1e9f, 1e9f};
      for (int st = 0; st < 2; ++st)
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(store_tiles, dim3(G), dim3(512), 0, 0, out, tiles, 8, pitch, busy, st, wait_ack, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best[st]) best[st] = ms;
        }
      const double bytes = (double)G * tiles * 131072.0;
      printf("ack %d %6d %6d %8d %10.1f %10.1f %10.2f %12.1f  exposed %.2f us/tile\n", wait_ack, G, tiles, busy, best[0] * 1e3, best[1] * 1e3, bytes / best[1] / 1e9, bytes / best[1] / 1e6 / G, (best[1] - best[0]) * 1e3 / tiles);
    }
  }
  return 0;
}

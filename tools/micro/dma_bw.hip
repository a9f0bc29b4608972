// L2 -> LDS feed-rate microbenchmark for the GEMM operand stream (no MFMA): every workgroup streams
// [rows][row_bytes] tiles of a panel it shares with `share` neighbouring workgroups, through LDS-DMA or through
// registers, with `depth` tiles in flight.  Prints aggregate TB/s.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_bw dma_bw.hip ; run: ./dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int TILE_BYTES, int DEPTH, bool DMA>
__global__ __launch_bounds__(NT) void feed(const char* __restrict__ base, int64_t panel_bytes, int share, int steps,
                                           uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PER = TILE_BYTES / 16 / NT;      // 16-byte chunks per thread per tile
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* panel = base + (int64_t)(blockIdx.x / share) * panel_bytes;
  u32x4 acc = {0u, 0u, 0u, 0u};
  auto issue = [&](int step, int buf) {
    const char* src = panel + ((int64_t)step * TILE_BYTES) % panel_bytes;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      if constexpr (DMA) {
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(buf * TILE_BYTES + (wave * 64 + i * NT) * 16));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 16),
                                         (__attribute__((address_space(3))) void*)(smem + off), 16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + c * 16);
        *reinterpret_cast<u32x4*>(smem + buf * TILE_BYTES + c * 16) = v;
      }
    }
  };
  for (int s = 0; s < DEPTH - 1; ++s) issue(s, s % DEPTH);
  for (int s = 0; s < steps; ++s) {
    if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER) : "memory");
    asm volatile("s_barrier" ::: "memory");
    issue(s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
    // touch the landed tile lightly so that the reads are not optimised away
    acc ^= *reinterpret_cast<const u32x4*>(smem + (s % DEPTH) * TILE_BYTES + tid * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

template <int NT, int TILE_BYTES, int DEPTH, bool DMA>
void run(const char* name, const char* buf, int64_t panel_bytes, int share, int blocks, int steps, uint32_t* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t lds = (size_t)DEPTH * TILE_BYTES;
  hipFuncSetAttribute((const void*)feed<NT, TILE_BYTES, DEPTH, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((feed<NT, TILE_BYTES, DEPTH, DMA>), dim3(blocks), dim3(NT), lds, 0, buf, panel_bytes, share, steps, sink);
  hipEventRecord(a);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((feed<NT, TILE_BYTES, DEPTH, DMA>), dim3(blocks), dim3(NT), lds, 0, buf, panel_bytes, share, steps, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double bytes = (double)blocks * steps * TILE_BYTES;
  printf("%-34s share %3d blocks %4d: %7.1f us  %6.2f TB/s  (%5.1f B/clk/CU @2.4GHz)  err=%s\n", name, share, blocks, ms * 1e3,
         bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e0 / 1e0 * 1e-0, hipGetErrorString(hipGetLastError()));
}

int main() {
  const int64_t total = 5ll << 30;
  char* buf; uint32_t* sink;
  hipMalloc(&buf, total); hipMemset(buf, 1, total); hipMalloc(&sink, 64);
  const int steps = 256;
  for (int share : {1, 8, 32, 256}) {
    // panel per group of `share` workgroups: steps * tile bytes (streamed once) -> HBM-fed unless shared
    {
      const int64_t pb = (int64_t)steps * 65536;
      run<512, 65536, 2, true>("dma  512thr 64KB x2", buf, pb, share, 256, steps, sink);
      run<512, 65536, 2, false>("regs 512thr 64KB x2", buf, pb, share, 256, steps, sink);
    }
    {
      const int64_t pb = (int64_t)steps * 32768;
      run<512, 32768, 4, true>("dma  512thr 32KB x4", buf, pb, share, 256, steps, sink);
      run<256, 32768, 2, true>("dma  256thr 32KB x2 (2/CU)", buf, pb, share, 512, steps, sink);
      run<256, 32768, 2, false>("regs 256thr 32KB x2 (2/CU)", buf, pb, share, 512, steps, sink);
    }
  }
  // tiny panel: everything L2-resident
  run<512, 65536, 2, true>("dma  512thr 64KB x2 L2-resident", buf, 65536 * 4, 256, 256, steps, sink);
  run<512, 65536, 2, false>("regs 512thr 64KB x2 L2-resident", buf, 65536 * 4, 256, 256, steps, sink);
  run<512, 32768, 4, true>("dma  512thr 32KB x4 L2-resident", buf, 65536 * 4, 256, 256, steps, sink);
  return 0;
}

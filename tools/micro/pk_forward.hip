// pk_forward.hip — reproducer attempt for the round-3 LayerNorm-backward fault: in the SLP-vectorised build of
// csrc/norm.hip the sequence
//      v_and_b32      vHI, 0xffff0000, vW          (bf16 -> fp32, odd element)
//      ...
//      v_lshlrev_b32  vLO, 16, vW                  (even element)
//      v_pk_add_f32   v[LO:HI], v[LO:HI], v[c:c+1] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]      <- reads vLO the instruction after it is written
// returned a STALE vLO in lanes 48-63 in a few launches per hundred, only while other kernels shared the chip (round 6: the same
// from a second stream of the SAME process, so not queue time-slicing; tools/contention_repeat.py CR_LOAD=thread).  This program runs
// exactly that instruction sequence (explicit registers, inline assembly) in a loop, with NOPS wait states between the shift and the
// packed op, against plain scalar arithmetic, while a second stream runs a load kernel; it prints the number of wrong results.
//   hipcc --offload-arch=gfx950 -O3 -o pk_forward pk_forward.hip && ./pk_forward
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int NOPS>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ in, unsigned long long* bad, unsigned* first_lane, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t w = in[tid];
  const float c0 = 0.37f + (tid & 7), c1 = -1.25f - (tid & 3);
  const float m0 = 1.5f, m1 = -0.75f;
  unsigned long long nbad = 0;
  for (int i = 0; i < iters; ++i) {
    w = (w & 0x80ff80ffu) | 0x3f003f00u;              // both bf16 halves finite and normal (|value| in [0.5, 2)): no NaN payload questions
    // reference: scalar ops, values passed through an asm barrier so that nothing is folded / vectorised
    float lo = __builtin_bit_cast(float, w << 16), hi = __builtin_bit_cast(float, w & 0xffff0000u);
    asm volatile("" : "+v"(lo), "+v"(hi));
    // op_sel:[0,1] on src1 = (c1 for the low lane? no: op_sel picks which half of the 64-bit source feeds the LOW result: [0,1] -> src0.lo, src1.hi
    //  and op_sel_hi defaults [1,1] -> src0.hi, src1.hi) ; neg_lo / neg_hi [0,1] negate src1 -> r.lo = lo - c1, r.hi = hi - c1
    const float r0 = (lo - c1) * m1, r1 = (hi - c1) * m0;                 // then v_pk_mul with op_sel:[1,0]: lo result = src0.hi * src1.lo ...
    float o0, o1;
    if constexpr (NOPS == 0) {
      asm volatile(
          "v_mov_b32 v212, %[c0]\n\tv_mov_b32 v213, %[c1]\n\tv_mov_b32 v214, %[m0]\n\tv_mov_b32 v215, %[m1]\n\t"
          "v_and_b32 v211, 0xffff0000, %[w]\n\t"
          "v_add_f32 v216, 0, v212\n\t"
          "v_add_f32 v217, v216, v213\n\t"
          "v_lshlrev_b32 v210, 16, %[w]\n\t"
          "v_pk_add_f32 v[210:211], v[210:211], v[212:213] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_lshlrev_b32 v218, 16, %[w]\n\t"
          "v_pk_mul_f32 v[210:211], v[214:215], v[210:211] op_sel:[1,0]\n\t"
          "v_mov_b32 %[o0], v210\n\tv_mov_b32 %[o1], v211\n\t"
          : [o0] "=&v"(o0), [o1] "=&v"(o1)
          : [w] "v"(w), [c0] "v"(c0), [c1] "v"(c1), [m0] "v"(m0), [m1] "v"(m1)
          : "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218");
    } else {
      asm volatile(
          "v_mov_b32 v212, %[c0]\n\tv_mov_b32 v213, %[c1]\n\tv_mov_b32 v214, %[m0]\n\tv_mov_b32 v215, %[m1]\n\t"
          "v_and_b32 v211, 0xffff0000, %[w]\n\t"
          "v_add_f32 v216, 0, v212\n\t"
          "v_add_f32 v217, v216, v213\n\t"
          "v_lshlrev_b32 v210, 16, %[w]\n\t"
          "s_nop 1\n\t"
          "v_pk_add_f32 v[210:211], v[210:211], v[212:213] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_lshlrev_b32 v218, 16, %[w]\n\t"
          "v_pk_mul_f32 v[210:211], v[214:215], v[210:211] op_sel:[1,0]\n\t"
          "v_mov_b32 %[o0], v210\n\tv_mov_b32 %[o1], v211\n\t"
          : [o0] "=&v"(o0), [o1] "=&v"(o1)
          : [w] "v"(w), [c0] "v"(c0), [c1] "v"(c1), [m0] "v"(m0), [m1] "v"(m1)
          : "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218");
    }
    // v_pk_mul op_sel:[1,0]: low result = src0.hi (m1) * src1.lo ; op_sel_hi default [1,1]: high result = src0.hi (m1) * src1.hi
    const float e0 = m1 * (lo - c1), e1 = m1 * (hi - c1);
    (void)r0; (void)r1;
    if (__builtin_bit_cast(uint32_t, o0) != __builtin_bit_cast(uint32_t, e0) || __builtin_bit_cast(uint32_t, o1) != __builtin_bit_cast(uint32_t, e1)) {
      if (nbad == 0) atomicMin(first_lane, (unsigned)(threadIdx.x & 63));
      ++nbad;
    }
    w = w * 1664525u + 1013904223u + (uint32_t)i;
  }
  if (nbad) atomicAdd(bad, nbad);
}

// the load: FMA chains + streaming traffic on a second stream
__global__ void load_kernel(float* buf, size_t n, int rounds) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float a = 1.0f + (i & 15), b = 0.999f;
  for (int r = 0; r < rounds; ++r) {
    for (int k = 0; k < 64; ++k) a = a * b + 0.5f;
    buf[(i + (size_t)r * 7919) % n] = a;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NOPS>
unsigned long long run(bool with_load, int launches, unsigned* first_lane_out) {
  const int nthreads = 256 * 2048;
  std::vector<uint32_t> h(nthreads);
  for (int i = 0; i < nthreads; ++i) h[i] = 0x3f800000u ^ (uint32_t)(i * 2654435761u);
  uint32_t* din; unsigned long long* dbad; unsigned* dfirst; float* lbuf;
  const size_t ln = (size_t)64 << 20;
  CK(hipMalloc(&din, nthreads * 4)); CK(hipMalloc(&dbad, 8)); CK(hipMalloc(&dfirst, 4)); CK(hipMalloc(&lbuf, ln * 4));
  CK(hipMemcpy(din, h.data(), nthreads * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dbad, 0, 8)); CK(hipMemset(dfirst, 0xff, 4));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int l = 0; l < launches; ++l) {
    if (with_load) hipLaunchKernelGGL(load_kernel, dim3(4096), dim3(256), 0, s2, lbuf, ln, 24);
    hipLaunchKernelGGL(probe<NOPS>, dim3(nthreads / 256), dim3(256), 0, s1, din, dbad, dfirst, 256);
  }
  CK(hipDeviceSynchronize());
  unsigned long long bad; CK(hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(first_lane_out, dfirst, 4, hipMemcpyDeviceToHost));
  CK(hipFree(din)); CK(hipFree(dbad)); CK(hipFree(dfirst)); CK(hipFree(lbuf));
  return bad;
}

int main() {
  unsigned fl;
  const int launches = 400;     // 400 x 524288 threads x 256 iterations = 5.4e10 evaluations of the sequence per configuration
  for (int load = 0; load <= 1; ++load) {
    unsigned long long b0 = run<0>(load, launches, &fl);
    printf("shift -> v_pk directly behind it, %s: %llu wrong results (first wrong lane %d)\n", load ? "with a load kernel on a second stream" : "idle GPU", b0, b0 ? (int)fl : -1);
    unsigned long long b1 = run<1>(load, launches, &fl);
    printf("shift -> s_nop 1 -> v_pk,          %s: %llu wrong results (first wrong lane %d)\n", load ? "with a load kernel on a second stream" : "idle GPU", b1, b1 ? (int)fl : -1);
  }
  return 0;
}

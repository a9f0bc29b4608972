// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (MXFP8, e4m3) operand / result layout on gfx950.
// A lane l supplies row i = l & 31 and 32 fp8 bytes, B lane l column j = l & 31 and 32 bytes; byte b of A lane (i, h = l >> 5)
// multiplies byte b of B lane (j, h); D[i][j] sits in lane (j + 32*h'), register r with i = (r & 3) + 8 * (r >> 2) + 4 * h'.
// MEASURED (hypothesis 2 below): the E8M0 scale byte (opsel 0) of lane-half h does NOT scale that lane's 32 bytes — it
// scales bytes 16h .. 16h+15 of BOTH lane halves of the row, i.e. MX block 0 = bytes 0..15 of lanes (i, 0) and (i, 1),
// block 1 = bytes 16..31 of both (the lanes hold k = 16h + b and 32 + 16h + (b - 16)).  csrc/attention_fp8.hip lays its
// scale blocks out accordingly.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f8_probe.hip -o tools/micro/mfma_f8_probe && tools/micro/mfma_f8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A8, const unsigned char* B8, const unsigned char* sA, const unsigned char* sB, float* D) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = *reinterpret_cast<const int*>(A8 + i * 64 + kb * 32 + w * 4);
    b[w] = *reinterpret_cast<const int*>(B8 + i * 64 + kb * 32 + w * 4);     // B8 stored [j][k]
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, (int)sA[i * 2 + kb], 0, (int)sB[i * 2 + kb]);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}

static float e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int main() {
  unsigned char hA[32 * 64], hB[32 * 64], hsA[64], hsB[64];
  srand(7);
  for (int t = 0; t < 32 * 64; ++t) {
    do { hA[t] = rand() & 0xff; } while ((hA[t] & 0x7f) >= 0x78);       // keep finite, |x| < 256
    do { hB[t] = rand() & 0xff; } while ((hB[t] & 0x7f) >= 0x78);
  }
  for (int t = 0; t < 64; ++t) { hsA[t] = 120 + rand() % 12; hsB[t] = 122 + rand() % 8; }
  unsigned char *dA, *dB, *dsA, *dsB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsA, hsA, 64, hipMemcpyHostToDevice); hipMemcpy(dsB, hsB, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, dD);
  float hD[32 * 32];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  // hypotheses about which of the two lanes' scales applies to byte b of lane-half kb:
  //   H1: the lane's own scale for all 32 bytes;  H2: bytes 0..15 take lane-half 0's scale, bytes 16..31 lane-half 1's;
  //   H3: no scale at all;  H4: transposed result (D[j][i]) with H1
  int ok_any = 0;
  for (int hyp = 1; hyp <= 4; ++hyp) {
    double worst = 0, scale = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int kb = 0; kb < 2; ++kb)
          for (int b = 0; b < 32; ++b) {
            const int kk = kb * 32 + b;
            const int sb = hyp == 2 ? b / 16 : kb;
            double sa = ldexp(1.0, hsA[i * 2 + sb] - 127), sbb = ldexp(1.0, hsB[j * 2 + sb] - 127);
            if (hyp == 3) sa = sbb = 1.0;
            ref += (double)e4m3(hA[i * 64 + kk]) * sa * (double)e4m3(hB[j * 64 + kk]) * sbb;
          }
        const float got = hyp == 4 ? hD[j * 32 + i] : hD[i * 32 + j];
        worst = fmax(worst, fabs(ref - got));
        scale = fmax(scale, fabs(ref));
      }
    printf("hypothesis %d: max |D - ref| = %.3g (max |ref| = %.3g) -> %s\n", hyp, worst, scale, worst <= 1e-4 * scale ? "HOLDS" : "no");
    if (worst <= 1e-4 * scale) ok_any = hyp;
  }
  printf("first rows of D: %g %g %g %g\n", hD[0], hD[1], hD[32], hD[33]);
  return ok_any == 2 ? 0 : 1;
}

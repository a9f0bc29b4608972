// Standalone probe: dumps the lane<->element maps of v_mfma_f32_16x16x4_f32 on the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, f32x4* c) {
  int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  c[l] = acc;
}
int main() {
  float ha[64], hb[64]; f32x4 hc[64];
  float *da, *db; f32x4* dc;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024);
  // A: lane id + 1 ; B: one-hot at lane t  -> D[i][j_t] = A[i][k_t]
  for (int t = 0; t < 64; t += 5) {
    for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (l == t) ? 1.f : 0.f; }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc); hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    printf("B one-hot at lane %d: nonzero D (lane,reg)=A-lane+1:", t);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l][r] != 0.f) printf(" (%d,%d)=%g", l, r, hc[l][r]);
    printf("\n");
  }
  for (int t = 0; t < 64; t += 7) {
    for (int l = 0; l < 64; ++l) { hb[l] = (float)(l + 1); ha[l] = (l == t) ? 1.f : 0.f; }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc); hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    printf("A one-hot at lane %d: nonzero D (lane,reg)=B-lane+1:", t);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l][r] != 0.f) printf(" (%d,%d)=%g", l, r, hc[l][r]);
    printf("\n");
  }
  return 0;
}

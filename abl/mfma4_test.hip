// probe: operand / result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // A[lane] = 100 + lane, B[lane] = 1 + lane/1000
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(100.f + l, 1.f + l * 0.001f, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) out[l * 4 + i] = c[i];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // hypothesis: lane l = 4b + j, VGPR i: D = A[4b+i] * B[4b+j]
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
    const int b = l / 4, j = l % 4;
    const float e = (100.f + 4 * b + i) * (1.f + (4 * b + j) * 0.001f);
    if (fabsf(h[l * 4 + i] - e) > 1e-3f) bad++;
  }
  printf("hypothesis lane=4b+j, vgpr=i: mismatches %d\n", bad);
  for (int l = 0; l < 8; l++) printf("lane %d: %.3f %.3f %.3f %.3f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}

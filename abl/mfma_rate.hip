// probe: issue rate of f32 MFMA shapes on gfx950 (one wave, independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void k(long long* out, float* sink, int iters) {
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f32x16 d0 = {0}, d1 = {0};
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
    } else if (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
    } else if (KIND == 2) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
    } else {   // dependent chain of 4x4x1 on ONE accumulator
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  sink[threadIdx.x] = s[0] + s[1] + s[2] + s[3] + d0[0] + d1[5];
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
}
int main() {
  long long* d; float* s; hipMalloc(&d, 64); hipMalloc(&s, 1024);
  const int iters = 100000;
  const char* nm[4] = {"4x4x1_16B (8 indep)", "16x16x4 (8 indep)", "32x32x2 (2 indep)", "4x4x1 dependent chain"};
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, iters);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, iters);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, s, iters);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, s, iters);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; i++) printf("%-26s %.2f ticks of s_memtime per MFMA\n", nm[i], (double)h[i] / (8.0 * iters));
  // wall-clock calibration of the tick
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, iters * 10); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("16x16x4: %.3f ms for %d MFMAs -> %.2f ns per MFMA; %lld ticks -> tick = %.3f ns\n", ms, iters * 80, ms * 1e6 / (iters * 80.0), h[1], ms * 1e6 / h[1]);
  return 0;
}

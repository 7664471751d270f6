// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) against double sin/cos on [-3, 3]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float r = x[i] * 0.15915494309189535f; s[i] = __builtin_amdgcn_sinf(r); c[i] = __builtin_amdgcn_cosf(r); }
}
int main() {
  const int n = 1 << 22;
  float *hx = new float[n], *hs = new float[n], *hc = new float[n];
  for (int i = 0; i < n; i++) hx[i] = -3.0f + 6.0f * (float)i / (n - 1);
  float *dx, *ds, *dc;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
  hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, es1 = 0;
  for (int i = 0; i < n; i++) {
    double a = fabs((double)hs[i] - sin((double)hx[i])), b = fabs((double)hc[i] - cos((double)hx[i]));
    if (a > es) es = a;
    if (b > ec) ec = b;
    if (fabs(hx[i]) <= 1.0 && a > es1) es1 = a;
  }
  printf("v_sin_f32 max abs err on [-3,3]: %.3g (on [-1,1]: %.3g); v_cos_f32: %.3g\n", es, es1, ec);
  return 0;
}

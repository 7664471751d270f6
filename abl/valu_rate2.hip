#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(long long* out, float* sink, int iters) {
  float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f + threadIdx.x * 1e-6f;
  float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP8(asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x0) : "v"(b));) }
    else if (KIND == 1) { REP8(asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(x0) : "v"(b));) }
    else if (KIND == 2) { REP8(asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));) }
    else if (KIND == 3) { REP8(asm volatile("v_add_f32_e64 %0, %4, %0\n v_add_f32_e64 %1, %4, %1\n v_add_f32_e64 %2, %4, %2\n v_add_f32_e64 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));) }
    else if (KIND == 4) { REP8(asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(x0) : "v"(b));) }
    else if (KIND == 5) { REP8(asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x0) : "v"(b));) }
    else if (KIND == 6) { REP8(asm volatile("s_nop 0");) }
    else if (KIND == 7) { REP8(asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x1) : "v"(b));) }
    else if (KIND == 8) { REP8(asm volatile("v_nop");) }
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
}
int main() {
  long long* d; float* s; (void)hipMalloc(&d, 256); (void)hipMalloc(&s, 1024);
  const int iters = 20000;
  const char* nm[9] = {"dep v_add_f32_e32 (4 B)", "dep v_add_f32_e64 (8 B)", "4 indep chains e32", "4 indep chains e64", "dep v_fmac_f32_e32 (4 B)", "dep v_fma_f32 (8 B)", "s_nop 0", "indep v_mov_b32_e32", "v_nop"};
  const int per[9] = {8, 8, 32, 32, 8, 8, 8, 8, 8};
#define L(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, d, s, iters)
  for (int rep = 0; rep < 2; rep++) { L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); (void)hipDeviceSynchronize(); }
  long long h[32]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 9; i++) printf("%-28s %7.2f ticks per instruction\n", nm[i], (double)h[i] / ((double)per[i] * iters));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, iters * 50); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("tick = %.3f ns; dep add e32 = %.3f ns each\n", ms * 1e6 / h[0], ms * 1e6 / (iters * 400.0));
  return 0;
}

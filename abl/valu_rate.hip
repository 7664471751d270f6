// probe: issue cost of VALU instruction patterns for ONE wave on gfx950 — what bounds a single wave's dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(long long* out, float* sink, int iters) {
  float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f + threadIdx.x * 1e-6f, c = 0.25f;
  float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  f32x2 p0 = {a, a + 1}, p1 = {a + 2, a + 3}, pb = {b, b};
  double d0 = a;
  __shared__ float lds[64 * 24];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 q0 = {a, b, c, a};
  unsigned la = (unsigned)(size_t)(&lds[threadIdx.x * 20]) , la2 = la;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));) }                    // dependent fma
    else if (KIND == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(b), "v"(c));) }   // 2 chains (16 instr)
    else if (KIND == 2) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));) }                   // dependent pk_fma
    else if (KIND == 3) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(p0), "+v"(p1) : "v"(pb));) }
    else if (KIND == 4) { REP8(asm volatile("v_sin_f32 %0, %0" : "+v"(x0));) }                                        // dependent sin
    else if (KIND == 5) { REP8(asm volatile("v_sin_f32 %0, %0\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(b), "v"(c));) }  // sin + 2 indep fma (24 instr)
    else if (KIND == 6) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(x0) : "v"(b), "v"(c));) }     // dependent fma + s_nop (16 instr)
    else if (KIND == 7) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s20, 0x3f000000" : "+v"(x0) : "v"(b), "v"(c) : "s20");) }   // + salu
    else if (KIND == 8) { REP8(asm volatile("v_add_f64 %0, %0, %0" : "+v"(d0));) }                                    // dependent f64 add
    else if (KIND == 9) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));) }  // 4 chains (32)
    else if (KIND == 10) { REP8(asm volatile("v_sin_f32 %0, %1\n v_sin_f32 %2, %3" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3));) }   // independent sins (16)
    else if (KIND == 11) { REP8(asm volatile("v_rcp_f32 %0, %0" : "+v"(x0));) }
    else if (KIND == 12) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(b) : "vcc");) }   // cmp + select (16)
    else if (KIND == 13) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x0) : "v"(b) : "vcc", "scc", "s20", "s21");) }  // (24)
    else if (KIND == 14) { REP8(asm volatile("ds_write_b128 %0, %1 offset:0" :: "v"(la), "v"(q0) : "memory");) }
    else if (KIND == 15) { REP8(asm volatile("ds_write_b64 %0, %1 offset:0" :: "v"(la), "v"(p0) : "memory");) }
    else if (KIND == 16) { REP8(asm volatile("ds_write_b32 %0, %1 offset:0" :: "v"(la), "v"(x1) : "memory");) }
    else if (KIND == 17) { REP8(asm volatile("ds_write_b128 %0, %1 offset:0\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "+v"(la2), "+v"(q0), "+v"(x0) : "v"(b), "v"(c) : "memory");) }   // write + 3 fma (32)
    else if (KIND == 18) { REP8(asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"(la), "v"(p0), "v"(p1) : "memory");) }
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = lds[threadIdx.x] + x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y + (float)d0;
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
}
int main() {
  long long* d; float* s; hipMalloc(&d, 256); hipMalloc(&s, 1024);
  const int iters = 20000;
  const char* nm[19] = {"dependent v_fma_f32", "2 independent fma chains", "dependent v_pk_fma_f32", "2 independent pk_fma chains", "dependent v_sin_f32",
                        "sin + 2 independent fma", "dependent fma + s_nop 0", "dependent fma + s_mov", "dependent v_add_f64", "4 independent fma chains",
                        "2 independent v_sin", "dependent v_rcp_f32", "v_cmp + v_cndmask (dep)", "v_cmp + s_and + v_cndmask (dep)", "ds_write_b128", "ds_write_b64", "ds_write_b32", "ds_write_b128 + 3 dep fma", "ds_write2_b64"};
  const int per[19] = {8, 16, 8, 16, 8, 24, 16, 16, 8, 32, 16, 8, 16, 24, 8, 8, 8, 32, 8};
#define L(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, d, s, iters)
  for (int rep = 0; rep < 2; rep++) { L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9); L(10); L(11); L(12); L(13); L(14); L(15); L(16); L(17); L(18); hipDeviceSynchronize(); }
  long long h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 19; i++) printf("%-34s %7.2f ticks per instruction (%d per iteration), %8.1f per group\n", nm[i], (double)h[i] / ((double)per[i] * iters), per[i], (double)h[i] / (8.0 * iters));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, iters * 50); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("dependent fma: %.3f ms for %d instr -> %.3f ns each; %lld ticks -> tick = %.3f ns\n", ms, iters * 400, ms * 1e6 / (iters * 400.0), h[0], ms * 1e6 / h[0]);
  return 0;
}

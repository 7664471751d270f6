// One wave per SIMD on gfx950: what a DEPENDENT instruction costs against an independent one (round 6: what bounds the metric's stepping loop).
//   hipcc --offload-arch=gfx950 -O3 abl/valu_lat.hip -o abl/valu_lat && abl/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(long long* out, float* sink, float a0) {
  float a = a0 + threadIdx.x, b = a0 * 2.f, c = a0 * 3.f, d = a0 * 5.f, x = 0.7f;
  f32x2 P = {a, b}, Q = {c, d}, K = {1.0001f, 0.9999f};
  long long t[16];
  int i = 0;
#define T0 t[i++] = __builtin_readcyclecounter(); asm volatile("" ::: "memory");
  T0 REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(b));)                                  // 0: dependent v_fma
  T0 REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(x));)   // 1: 4 independent chains
  T0 REP64(asm volatile("v_sin_f32 %0, %0" : "+v"(a));)                                                            // 2: dependent v_sin
  T0 REP16(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)   // 3: independent v_sin
  T0 REP64(asm volatile("v_sin_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(x));)             // 4: sin -> fma chain (pairs)
  T0 REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)   // 5: dependent dpp mov (+ s_nop 1)
  T0 REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(P) : "v"(K), "v"(Q));)                                // 6: dependent pk_fma
  T0 REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(P), "+v"(Q) : "v"(K), "v"(K));)   // 7: two alternating pk chains
  T0 REP64(asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(a), "+v"(b) : "v"(x));)       // 8: two alternating chains (128 instr)
  T0 REP64(asm volatile("v_rcp_f32 %0, %0\n s_nop 0\n v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));)                  // 9: rcp -> mul
  T0 REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x) : "vcc");)   // 10: cmp -> cndmask dependent
  T0 REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&P) : "v"(*(double*)&Q));)                          // 11: dependent f64 add
  T0
  if (threadIdx.x == 0) for (int j = 0; j + 1 < i; j++) out[blockIdx.x * 16 + j] = t[j + 1] - t[j];
  sink[threadIdx.x] = a + b + c + d + P.x + P.y + Q.x + Q.y;
}
int main() {
  long long* o; float* s;
  hipMalloc(&o, 16 * 8 * 4); hipMalloc(&s, 64 * 4 * 4);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, 0.001f);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"dependent v_fma_f32 (64)", "4 independent v_fma chains (64)", "dependent v_sin_f32 (64)", "4 independent v_sin chains (64)", "v_sin -> s_nop 0 -> v_fma pairs (64 pairs)",
                      "s_nop 1 + dependent v_mov_dpp (64)", "dependent v_pk_fma_f32 (64)", "2 alternating v_pk_fma chains (64)", "2 alternating v_fma chains (128)", "v_rcp -> s_nop 0 -> v_mul pairs (64)",
                      "v_cmp -> v_cndmask dependent pairs (64)", "dependent v_add_f64 (64)"};
  const int cnt[] = {64, 64, 64, 64, 64, 64, 64, 64, 128, 64, 64, 64};
  for (int j = 0; j < 12; j++) printf("%-46s %6lld cycles = %5.1f per unit\n", nm[j], h[j], (double)h[j] / cnt[j]);
  return 0;
}

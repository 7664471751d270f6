// dependent-issue cost of the instruction forms the stepping loop is made of (one wave on a SIMD, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(long long* out, float* sink, float a0) {
  float a = a0 + threadIdx.x, b = a0 * 2.f, x = 0.7f, y = 1.0001f;
  f32x2 P = {a, b}, K = {1.0001f, 0.9999f}, Q = {b, a}; double dd = a;
  long long t[24];
  int i = 0;
#define T0 t[i++] = __builtin_readcyclecounter(); asm volatile("" ::: "memory");
  T0 REP64(asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));)                                  // 0 (acc chain)
  T0 REP64(asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a) : "v"(y));)                                           // 1
  T0 REP64(asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a) : "v"(x));)                                           // 2
  T0 REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));)                                   // 3 VOP3 fma, dependent through src2
  T0 REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(y), "v"(x));)                                   // 4 VOP3 fma, dependent through src0
  T0 REP64(asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a) : "v"(x));)                                           // 5
  T0 REP64(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));)                                  // 6
  T0 REP64(asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a));)                                                        // 7
  T0 REP64(asm volatile("v_log_f32_e32 %0, %0" : "+v"(a));)                                                        // 8
  T0 REP64(asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a));)                                                        // 9
  T0 REP64(asm volatile("v_sin_f32_e32 %0, %0\n v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));)           // 10 sin -> fmac (no nop)
  T0 REP64(asm volatile("v_sin_f32_e32 %0, %0\n v_mul_f32_e32 %0, %1, %0" : "+v"(a) : "v"(y));)   // 11 sin -> mul
  T0 REP64(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)   // 12 dependent add_dpp
  T0 REP64(asm volatile("v_mul_f32_e32 %0, %1, %0\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(y));)   // 13 mul -> dpp pairs
  T0 REP64(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(P) : "v"(K));)                                            // 14
  T0 REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(P) : "v"(K), "v"(Q));)                                // 15 pk_fma through src2
  T0 REP64(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(dd), "+v"(a));)             // 16 cvt pair
  T0 REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 0\n s_cbranch_vccz 0" : : "v"(a), "v"(x) : "vcc");)      // 17 cmp -> branch (not taken / taken to next)
  T0 REP64(asm volatile("ds_write_b32 %0, %1" : : "v"(threadIdx.x * 4), "v"(a) : "memory");)                       // 18 lds writes
  T0 REP16(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(threadIdx.x * 4) : "memory");) // 19 lds read + wait (16)
  T0 REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20" : "+v"(a) : : "s20");)                   // 20 readfirstlane -> mov
  T0
  if (threadIdx.x == 0) for (int j = 0; j + 1 < i; j++) out[j] = t[j + 1] - t[j];
  sink[threadIdx.x] = a + b + P.x + P.y + Q.x + Q.y;
}
int main() {
  long long* o; float* s;
  (void)hipMalloc(&o, 32 * 8); (void)hipMalloc(&s, 64 * 4 * 4);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, 0.001f);
  (void)hipDeviceSynchronize();
  long long h[32]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"v_fmac_f32_e32 chain", "v_mul_f32 chain", "v_add_f32 chain", "v_fma_f32 (VOP3) chain via src2", "v_fma_f32 (VOP3) chain via src0", "v_max_f32 chain", "v_med3_f32 chain",
                      "v_exp_f32 chain", "v_log_f32 chain", "v_rcp_f32 chain", "v_sin -> v_fmac pairs", "v_sin -> v_mul pairs", "s_nop 1 + v_add_f32_dpp chain",
                      "v_mul -> s_nop 1 -> v_mov_dpp pairs", "v_pk_mul_f32 chain", "v_pk_fma_f32 chain via src2", "v_cvt_f64_f32 -> v_cvt_f32_f64 pairs", "v_cmp -> s_nop -> s_cbranch_vccz", "ds_write_b32 (independent)",
                      "ds_read_b32 + s_waitcnt (16)", "v_readfirstlane -> v_mov pairs"};
  const int cnt[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 64};
  for (int j = 0; j < 21; j++) printf("%-46s %6lld cycles = %5.1f per unit\n", nm[j], h[j], (double)h[j] / cnt[j]);
  return 0;
}

// lde_device.h — device-side building blocks shared by the gfx950 kernels of liblde.
//
// Everything here is per-lane register code: Tsit5 5(4) with the free 4th-order interpolant,
// classical RK4, the Hairer–Nørsett–Wanner initial step and the PI step controller, i.e. the
// pieces of OrdinaryDiffEq 6.27.1 [REF Manifest.toml:979] that run under
//     solve(ens_prob, solver, EnsembleThreads(); saveat = t, kwargs...)   [REF src/models/GOKU.jl:121]
// State and slopes are f32, time / dt are f64 (the reference's `t` is a Float64 range while the
// state is Float32 [REF examples/pendulum_friction-less/model_train.jl:44, :181]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lde.h"
#include "lde_types.h"

namespace lde {

// ---- fast scalar math (gfx950 hardware transcendentals; ≈1 ulp) ----------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// x^y for x > 0 via v_log_f32 / v_exp_f32 (the step controller only needs ~1e-6 relative accuracy;
// OrdinaryDiffEq itself uses a low-precision `fastpow` here)
__device__ __forceinline__ float fast_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

// sin and cos with ≈1.3e-7 absolute error: k = rint(x/π), r = x − kπ (two-term Cody–Waite with FMA),
// odd/even minimax polynomials on [−π/2, π/2], sign (−1)^k. ~25 VALU instructions for the pair,
// branch-free (ocml's sinf carries a Payne–Hanek slow path that costs ≈150 instructions of code per call).
__device__ __forceinline__ void fast_sincos(float x, float& s, float& c) {
  const float k = rintf(x * 0.3183098861837907f);
  float r = fmaf(-k, 3.1415927410125732f, x);
  r = fmaf(-k, -8.742277657347586e-08f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, 2.6348157007305417e-06f, -0.00019822761532850564f);
  ps = fmaf(ps, r2, 0.008333242498338223f);
  ps = fmaf(ps, r2, -0.1666666567325592f);
  const float sv = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.629789719321707e-07f, 2.477459747751709e-05f);
  pc = fmaf(pc, r2, -0.0013888651737943292f);
  pc = fmaf(pc, r2, 0.0416666604578495f);
  pc = fmaf(pc, r2, -0.5f);
  const float cv = fmaf(r2, pc, 1.0f);
  const int sign = ((int)k) << 31;
  s = __int_as_float(__float_as_int(sv) ^ sign);
  c = __int_as_float(__float_as_int(cv) ^ sign);
}
__device__ __forceinline__ float fast_sin(float x) {
  const float k = rintf(x * 0.3183098861837907f);
  float r = fmaf(-k, 3.1415927410125732f, x);
  r = fmaf(-k, -8.742277657347586e-08f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, 2.6348157007305417e-06f, -0.00019822761532850564f);
  ps = fmaf(ps, r2, 0.008333242498338223f);
  ps = fmaf(ps, r2, -0.1666666567325592f);
  const float sv = fmaf(r * r2, ps, r);
  return __int_as_float(__float_as_int(sv) ^ (((int)k) << 31));
}
// The hardware's v_sin_f32 / v_cos_f32 on x/2π — one multiply and one quarter-rate instruction each; measured max absolute
// error 2.7e-7 on [−3, 3], 1.4e-7 on [−1, 1] (abl/vsin_test.hip). Used by the analytic GOKU kernels (lde_pendulum.hip), whose
// duration at small batches is one wave's dependent-instruction chain: the parity margins are unchanged against the
// polynomials (abl/sin_margin.py: 2.4e-6 vs 2.3e-6 from the oracle at 1e-6/1e-6, gate 1e-5) and a Tsit5 step is 11 % shorter.
// NOT used where the sine rides along a network (physics + MLP): there the adaptive adjoint's longest trajectory took 9 %
// more steps with the slightly noisier sine (c3: 5.65 → 6.01 ms per step), more than the instruction saving is worth.
// -DLDE_HW_SIN=0 (diagnostic) makes these the polynomials too.
#ifndef LDE_HW_SIN
#define LDE_HW_SIN 1
#endif
// v_sin_f32 / v_cos_f32 are only defined for |x/2π| ≤ 256 (outside they return 0 / 1 — a pendulum that has rotated past
// ≈ 1608 rad would silently lose its restoring force). The argument is therefore reduced — not per sine (v_fract_f32 would
// put one more instruction on every stage's dependent chain AND lose the absolute precision of small negative angles:
// fract(−ε) = 1 − ε rounds at 6e-8 turns), but ONCE PER STEP ATTEMPT: the caller anchors the whole turns of the step's start
// angle, noff = −rint(x₀/2π), and every stage evaluates sin(2π·fma(x, 1/2π, noff)) — the multiply that was there anyway,
// exact for |x| < π (noff = 0), and within a step the angle moves by |h·ω| ≪ 256 turns.
constexpr float INV_2PI = 0.15915494309189535f;
__device__ __forceinline__ float turn_anchor(float x0) { return LDE_HW_SIN ? -rintf(x0 * INV_2PI) : 0.f; }
__device__ __forceinline__ float hw_sin(float x, float noff = 0.f) {
  return LDE_HW_SIN ? __builtin_amdgcn_sinf(fmaf(x, INV_2PI, noff)) : fast_sin(x);
}
__device__ __forceinline__ void hw_sincos(float x, float& s, float& c, float noff = 0.f) {
  if (LDE_HW_SIN) {
    const float r = fmaf(x, INV_2PI, noff);
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
  } else
    fast_sincos(x, s, c);
}

// ---- Tsit5 tableau (Tsitouras 2011), f32 copies of the f64 constants ------------------------
namespace ts5 {
constexpr float A[7][6] = {
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.161f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {(float)-0.008480655492356989, (float)0.335480655492357, 0.f, 0.f, 0.f, 0.f},
    {(float)2.8971530571054935, (float)-6.359448489975075, (float)4.3622954328695815, 0.f, 0.f, 0.f},
    {(float)5.325864828439257, (float)-11.748883564062828, (float)7.4955393428898365, (float)-0.09249506636175525, 0.f, 0.f},
    {(float)5.86145544294642, (float)-12.92096931784711, (float)8.159367898576159, (float)-0.071584973281401,
     (float)-0.028269050394068383, 0.f},
    {(float)0.09646076681806523, 0.01f, (float)0.4798896504144996, (float)1.379008574103742, (float)-3.290069515436081,
     (float)2.324710524099774}};
constexpr float BT[7] = {(float)-0.00178001105222577714, (float)-0.0008164344596567469, (float)0.007880878010261995,
                         (float)-0.1447110071732629,     (float)0.5823571654525552,     (float)-0.45808210592918697,
                         (float)0.015151515151515152};
constexpr double C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
constexpr float R1[3] = {(float)-2.763706197274826, (float)2.9132554618219126, (float)-1.0530884977290216};
constexpr float R[6][3] = {{(float)0.13169999999999998, -0.2234f, 0.1017f},
                           {(float)3.9302962368947516, (float)-5.941033872131505, (float)2.490627285651253},
                           {(float)-12.411077166933676, (float)30.33818863028232, (float)-16.548102889244902},
                           {(float)37.50931341651104, (float)-88.1789048947664, (float)47.37952196281928},
                           {(float)-27.896526289197286, (float)65.09189467479366, (float)-34.87065786149661},
                           {1.5f, -4.0f, 2.5f}};
}  // namespace ts5

// weights of the continuous extension: u(t+Θh) = u + h Σ b_i(Θ) k_i
__device__ __forceinline__ void tsit5_interp_weights(float th, float (&bw)[7]) {
  bw[0] = th * (1.0f + th * (ts5::R1[0] + th * (ts5::R1[1] + th * ts5::R1[2])));
  const float th2 = th * th;
#pragma unroll
  for (int i = 0; i < 6; i++) bw[i + 1] = th2 * (ts5::R[i][0] + th * (ts5::R[i][1] + th * ts5::R[i][2]));
}

// Dense output in Horner form: u(t+Θh) = u + h·(Θ·k₁ + Θ²·(P₂ + Θ(P₃ + Θ·P₄))) with the Θ-independent
// P_m = Σ_i r_{i,m} k_i computed once per accepted step (5 FMA per component and save point afterwards).
template <int N>
__device__ __forceinline__ void tsit5_dense_coeffs(const float (&k)[7][N], float (&P)[3][N]) {
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int i = 0; i < N; i++) {
      float acc = ts5::R1[m] * k[0][i];
#pragma unroll
      for (int s = 0; s < 6; s++) acc += ts5::R[s][m] * k[s + 1][i];
      P[m][i] = acc;
    }
}
template <int N>
__device__ __forceinline__ float tsit5_dense_eval(float th, float h, float y, float k1, float P2, float P3, float P4) {
  return y + (h * th) * (k1 + th * (P2 + th * (P3 + th * P4)));
}

// One Tsit5 attempt on an N-vector held in registers. k[0] = f(y) on entry.
// Leaves k[1..6], yn; returns the RMS error estimate (0 when !adaptive).
// MSQ: return the MEAN SQUARE of the scaled error (EEst²) instead of EEst — for callers that run the controller on log₂ EEst.
// ADAPT: 1 / 0 = the error estimate is / is not needed, decided at compile time; −1 = o.adaptive decides at run time.
template <int N, class F, bool MSQ = false, int ADAPT = -1>
__device__ __forceinline__ float tsit5_attempt(F& f, float h, const float (&y)[N], float (&k)[7][N], float (&yn)[N],
                                               const KOpts& o) {
  float tmp[N];
#pragma unroll
  for (int s = 1; s < 6; s++) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      float acc = ts5::A[s][0] * k[0][i];
#pragma unroll
      for (int j = 1; j < s; j++) acc += ts5::A[s][j] * k[j][i];
      tmp[i] = y[i] + h * acc;
    }
    f(tmp, k[s]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    float acc = ts5::A[6][0] * k[0][i];
#pragma unroll
    for (int j = 1; j < 6; j++) acc += ts5::A[6][j] * k[j][i];
    yn[i] = y[i] + h * acc;
  }
  f(yn, k[6]);
  if (ADAPT == 0 || (ADAPT < 0 && !o.adaptive)) return 0.f;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < N; i++) {
    float e = ts5::BT[0] * k[0][i];
#pragma unroll
    for (int j = 1; j < 7; j++) e += ts5::BT[j] * k[j][i];
    e *= h;
    const float sk = o.abstol + fmaxf(fabsf(y[i]), fabsf(yn[i])) * o.reltol;
    const float r = e * fast_rcp(sk);
    s2 += r * r;
  }
  return MSQ ? s2 * (1.0f / N) : sqrtf(s2 * (1.0f / N));
}

// classical RK4 on registers; k[0] = f(y) on entry, leaves k[4] = f(yn).
template <int N, class F>
__device__ __forceinline__ void rk4_step(F& f, float h, const float (&y)[N], float (&k)[7][N], float (&yn)[N]) {
  float tmp[N];
  const float hh = 0.5f * h;
#pragma unroll
  for (int i = 0; i < N; i++) tmp[i] = y[i] + hh * k[0][i];
  f(tmp, k[1]);
#pragma unroll
  for (int i = 0; i < N; i++) tmp[i] = y[i] + hh * k[1][i];
  f(tmp, k[2]);
#pragma unroll
  for (int i = 0; i < N; i++) tmp[i] = y[i] + h * k[2][i];
  f(tmp, k[3]);
  const float h6 = h * (1.0f / 6.0f);
#pragma unroll
  for (int i = 0; i < N; i++) yn[i] = y[i] + h6 * (k[0][i] + 2.0f * (k[1][i] + k[2][i]) + k[3][i]);
  f(yn, k[4]);
}

// PI controller (OrdinaryDiffEq PIController, Tsit5 defaults β₁=7/50, β₂=2/25).
__device__ __forceinline__ float pi_q(float EEst, float qold, const KOpts& o, float& q11) {
  if (EEst == 0.f) {
    q11 = 0.f;
    return o.q_lo;
  }
  q11 = fast_pow(EEst, o.beta1);
  float q = q11 * fast_pow(qold, -o.beta2);
  return fmaxf(o.q_lo, fminf(o.q_hi, q * o.inv_gamma));
}

// Hairer–Nørsett–Wanner initial step for a 5th-order method; f0 = f(y0). sign = ±1.
template <int N, class F>
__device__ __forceinline__ double init_dt(F& f, const float (&y0)[N], const float (&f0)[N], float sign, double dtmax,
                                          const KOpts& o) {
  float s0 = 0.f, s1 = 0.f;
  float sk[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    sk[i] = fast_rcp(o.abstol + fabsf(y0[i]) * o.reltol);  // 1/scale
    const float a = y0[i] * sk[i], b = f0[i] * sk[i];
    s0 += a * a;
    s1 += b * b;
  }
  const float d0 = sqrtf(s0 * (1.0f / N)), d1 = sqrtf(s1 * (1.0f / N));
  double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
  if (dt0 > dtmax) dt0 = dtmax;
  const float h = sign * (float)dt0;
  float tmp[N], f1[N];
#pragma unroll
  for (int i = 0; i < N; i++) tmp[i] = y0[i] + h * f0[i];
  f(tmp, f1);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const float a = (f1[i] - f0[i]) * sk[i];
    s2 += a * a;
  }
  const float d2 = sqrtf(s2 * (1.0f / N)) * fast_rcp((float)dt0);
  const float dm = fmaxf(d1, d2);
  // 10^(-(2+log10 dm)/5) = 10^-0.4 · dm^-0.2
  const double dt1 = (dm <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm, -0.2f));
  double dt = fmin(100.0 * dt0, dt1);
  return dt > dtmax ? dtmax : dt;
}

template <int N>
__device__ __forceinline__ bool all_finite(const float (&y)[N]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && isfinite(y[i]);
  return ok;
}

}  // namespace lde

// lde_pendulum.hip — GOKU path, analytic right-hand sides, one lane per trajectory.
//
// Replaces the per-trajectory ensemble solve of
//     diffeq_layer(::Decoder{<:GOKU}, (ẑ₀, θ̂), t)          [REF src/models/GOKU.jl:98-130]
// (prob_func/remake: column i ↦ trajectory i [REF :111]; output_func: NaN block on failure [REF :114];
//  result laid out [D × B × T] after permutedims [REF :125]) and its reverse-mode pullback.
//
// Design (gfx950): every trajectory is an independent IVP with its own adaptive step sequence, so a
// lane owns a trajectory and keeps the whole solver state in VGPRs — state (2 f32), the seven Tsit5
// slopes, t/dt in f64. There is no LDS traffic in the step loop and no inter-lane communication;
// the only global traffic is the algorithmic one: 12 B in, 8·T B out per trajectory, with lane ↔ batch
// index so that every load/store of a wave covers 64 consecutive float2 (512 B).
// Wave-level divergence comes only from differing step counts (11–17 at default tolerances).
#include <cstdlib>

#include "lde_device.h"

namespace lde {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the closed menu of 2-state / 1-parameter physics RHS -------------------------------------
// KIND 0: du = [y, -(G/L) sin x]                    [REF examples/pendulum_friction-less/pendulum.jl:19-26]
// KIND 1: du = [y, -(G/L) sin x - (b/m) y], b/m=0.7  [REF pendulum.jl:65-74]
template <int KIND>
struct PendFwd {
  float ngl;  // -G/L
  __device__ __forceinline__ explicit PendFwd(float L) : ngl(-10.0f / L) {}  // one IEEE division per trajectory
  __device__ __forceinline__ void operator()(const float (&y)[2], float (&dy)[2]) const {
    dy[0] = y[1];
    float acc = ngl * hw_sin(y[0]);
    if (KIND == 1) acc -= 0.7f * y[1];
    dy[1] = acc;
  }
};

// augmented reverse-time system on [z₀ z₁ λ₀ λ₁ g]:  ż=f, λ̇=-(∂f/∂z)ᵀλ, ġ=-(∂f/∂L)ᵀλ
template <int KIND>
struct PendBwd {
  float ngl, gl2;  // -G/L, G/L²
  __device__ __forceinline__ explicit PendBwd(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)) {}
  __device__ __forceinline__ void operator()(const float (&y)[5], float (&dy)[5]) const {
    float s, c;
    hw_sincos(y[0], s, c);
    dy[0] = y[1];
    float acc = ngl * s;
    if (KIND == 1) acc -= 0.7f * y[1];
    dy[1] = acc;
    dy[2] = -(ngl * c * y[3]);
    float v1 = y[2];
    if (KIND == 1) v1 -= 0.7f * y[3];
    dy[3] = -v1;
    dy[4] = -(gl2 * s * y[3]);
  }
};

constexpr int TS_LDS_MAX = 6000;   // doubles of the save-time grid kept in LDS (48 KB)

// ---- forward ------------------------------------------------------------------------------------
template <int KIND, int SOLVER, bool TS_LDS>
__global__ void __launch_bounds__(256) k_pend_forward(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                      const double* __restrict__ ts_g, KOpts o,
                                                      float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                      int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                      int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];
  const int T = o.T, B = o.B;
  // the save-time grid is staged in LDS when it fits (T ≤ 6000, TS_LDS); longer grids are read from L2
  if (TS_LDS)
    for (int i = threadIdx.x; i < T; i += blockDim.x) s_lds[i] = ts_g[i];
  __syncthreads();
  auto s_ts = [&](int i) -> double { return TS_LDS ? s_lds[i] : ts_g[i]; };   // compile-time choice
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;

  const float2 zi = z0[b];
  PendFwd<KIND> f(theta[b]);
  float y[2] = {zi.x, zi.y};
  float k[7][2], yn[2];
  z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
  int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;

  if (T > 1) {
    double t = s_ts(0);
    const double tend = s_ts(T - 1), dtmax = tend - t;
    f(y, k[0]);
    nfe = 1;
    double dt;
    if (o.adaptive) {
      if (o.dt_fixed > 0) dt = fmin(o.dt_fixed, dtmax);
      else {
        dt = init_dt<2>(f, y, k[0], 1.0f, dtmax, o);
        nfe++;
      }
    } else
      dt = o.dt_fixed;
    float qold = 1e-4f;
    long long iters = 0;
    int j = 1;
    // next save time kept in a register, the one after it prefetched: no LDS round trip on the save loop's exit test
    double tj = s_ts(1), tjn = s_ts(min(2, T - 1));
    while (t < tend) {
      if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
      double dtp = dt;
      bool last = false;
      if (t + dt >= tend - 1e-12 * fabs(tend)) { dt = tend - t; last = true; }
      const float h = (float)dt;
      float EEst = 0.f;
      if (SOLVER == LDE_SOLVER_TSIT5) {
        EEst = tsit5_attempt<2>(f, h, y, k, yn, o);
        nfe += 6;
      } else {
        rk4_step<2>(f, h, y, k, yn);
        nfe += 4;
      }
      if (!all_finite<2>(yn) || !(EEst == EEst)) {
        if (o.adaptive && dt > o.dtmin) { nrej++; dt = dt * (double)o.qmin; continue; }
        ret = LDE_RET_NONFINITE;
        break;
      }
      if (o.adaptive) {
        float q11;
        const float q = pi_q(EEst, qold, o, q11);
        if (EEst > 1.0f) {
          nrej++;
          dt = dt * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
          if (dt < o.dtmin) { ret = LDE_RET_DTMIN; break; }
          continue;
        }
        qold = fmaxf(EEst, 1e-4f);
        dtp = dt * (double)fast_rcp(q);
        if (dtp > dtmax) dtp = dtmax;
      }
      nacc++;
      const double tnew = last ? tend : t + dt;
      if (j < T && tj <= tnew) {  // at least one save time in (t, tnew]
        float P[3][2];
        if (SOLVER == LDE_SOLVER_TSIT5) tsit5_dense_coeffs<2>(k, P);
        const float rh = fast_rcp(h);
        do {
          float2 out;
          if (tj >= tnew || (j == T - 1 && last)) {
            out = make_float2(yn[0], yn[1]);
          } else {
            const float th = (float)(tj - t) * rh;
            if (SOLVER == LDE_SOLVER_TSIT5) {
              out.x = tsit5_dense_eval<2>(th, h, y[0], k[0][0], P[0][0], P[1][0], P[2][0]);
              out.y = tsit5_dense_eval<2>(th, h, y[1], k[0][1], P[0][1], P[1][1], P[2][1]);
            } else {  // cubic Hermite between (y,k1) and (yn,f(yn))
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              out.x = h00 * y[0] + (h10 * h) * k[0][0] + h01 * yn[0] + (h11 * h) * k[4][0];
              out.y = h00 * y[1] + (h10 * h) * k[0][1] + h01 * yn[1] + (h11 * h) * k[4][1];
            }
          }
          z_out[(size_t)j * B + b] = out;
          j++;
          tj = tjn;
          tjn = s_ts(min(j + 1, T - 1));
        } while (j < T && tj <= tnew);
      }
      y[0] = yn[0];
      y[1] = yn[1];
      constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // FSAL slope
      k[0][0] = k[FS][0];
      k[0][1] = k[FS][1];
      t = tnew;
      dt = o.adaptive ? dtp : o.dt_fixed;
    }
  }
  if (ret != LDE_RET_SUCCESS) {  // failed solve ⇒ NaN block, never an error [REF GOKU.jl:114]
    const float qn = __int_as_float(0x7fc00000);
    for (int j = 0; j < T; j++) z_out[(size_t)j * B + b] = make_float2(qn, qn);
  }
  if (retcode) retcode[b] = ret;
  st_ret[b] = ret;
  st_nfe[b] = nfe;
  st_nacc[b] = nacc;
  st_nrej[b] = nrej;
}

// ---- forward, small batches: stepping and dense output on different waves ---------------------------------------------
// At B ≤ 16384 (one 64-trajectory workgroup per CU) the launch is a handful of waves per CU and its duration is one wave's dependent-instruction chain. In
// k_pend_forward a third of that chain is the dense output: ≈ 75 wave-iterations (the per-step maximum over 64 lanes of
// the saves inside the step) of interpolation, f64 save-time compares and stores — 9.4 of 28.4 µs at T = 50
// (abl/pend_T.py). Here the stepping wave only RECORDS each accepted step (t, h, y, k₁…k₇) in LDS, five 16-byte words per step,
// and the workgroup's sixteen waves then evaluate the saves from the records, each a contiguous slice of the save grid for
// the same 64 trajectories (one barrier pair per ≤ 20 accepted steps; a trajectory that needs more steps than the record
// area holds simply goes through another round). Same formulas as k_pend_forward; the two compilations contract
// multiply-adds differently, so they agree to the solver's tolerance, not bit for bit (tests/test_gpu_pendulum.py).
// Measured (B = 256, T = 50): 28.4 → 22.7 µs.
constexpr int WS_CAP = 20;       // accepted steps recorded between two save phases
constexpr int WS_WAVES = 16;     // waves per workgroup: one steps, all sixteen evaluate saves
constexpr int WS_THREADS = 64 * WS_WAVES;
constexpr int WS_RW = 20;        // floats per record = five 16-byte words: {h, y₀, y₁, –} {k₁ k₂} {k₃ k₄} {k₅ k₆} {k₇, t (f64)};
                                 // a lane stride of 80 B spreads the 16 lanes of a b128 access over all 64 banks

template <int KIND, int SOLVER>
__global__ void __launch_bounds__(WS_THREADS) k_pend_forward_ws(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                         const double* __restrict__ ts_g, KOpts o,
                                                         float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                         int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                         int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];
  __shared__ int s_cnt[64];
  __shared__ int s_done;
  const int T = o.T, B = o.B, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < T; i += WS_THREADS) s_lds[i] = ts_g[i];
  float* rec = reinterpret_cast<float*>(s_lds + ((T + 1) & ~1));          // [(WS_CAP + 1)][64][WS_RW]
  auto rec_at = [&](int n) -> float* { return rec + (size_t)(n * 64 + lane) * WS_RW; };
  auto rec_t = [&](int n) -> double { return *reinterpret_cast<const double*>(rec_at(n) + 18); };   // start time of record n
  __syncthreads();
  auto s_ts = [&](int i) -> double { return s_lds[i]; };
  const int b = blockIdx.x * 64 + lane;
  const bool valid = b < B;
  constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // FSAL slope

  // --- the stepping wave's state (wave 0) ---
  float y[2] = {0.f, 0.f}, k[7][2], yn[2], kf[2] = {0.f, 0.f};   // kf: f(y) at the current state (the first-same-as-last slope)
  PendFwd<KIND> f(valid && w == 0 ? theta[b] : 1.0f);
  int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;
  double t = 0.0, dt = 0.0, tend = 0.0, dtmax = 0.0;
  float lqold = -13.287712379549449f;   // log₂ of qold = 1e-4: the PI controller runs on log₂ EEst here (see below)
  long long iters = 0;
  bool active = false;
#pragma unroll
  for (int s = 0; s < 7; s++) k[s][0] = k[s][1] = 0.f;
  if (w == 0 && valid) {
    const float2 zi = z0[b];
    y[0] = zi.x;
    y[1] = zi.y;
    z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
    if (T > 1) {
      t = s_ts(0);
      tend = s_ts(T - 1);
      dtmax = tend - t;
      f(y, kf);
      nfe = 1;
      if (o.adaptive) {
        if (o.dt_fixed > 0) dt = fmin(o.dt_fixed, dtmax);
        else {
          dt = init_dt<2>(f, y, kf, 1.0f, dtmax, o);
          nfe++;
        }
      } else
        dt = o.dt_fixed;
      active = t < tend;
    }
  }
  // --- the save waves' state: wave w owns the saves j ∈ [jq, jend) of its 64 trajectories ---
  const int chunk = (T - 1 + WS_WAVES - 1) / WS_WAVES;
  int jq = 1 + w * chunk;
  const int jend = min(T, 1 + (w + 1) * chunk);

  for (;;) {
    if (w == 0) {
      int n = 0;
      for (;;) {
        const bool go = active && n < WS_CAP;
        if (!__any(go)) break;
        if (go) do {
          if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; active = false; break; }
          double dtp = dt;
          bool last = false;
          if (t + dt >= tend - 1e-12 * fabs(tend)) { dt = tend - t; last = true; }
          const float h = (float)dt;
          k[0][0] = kf[0];
          k[0][1] = kf[1];
          float EEst = 0.f;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            EEst = tsit5_attempt<2, PendFwd<KIND>, true>(f, h, y, k, yn, o);   // EEst²
            nfe += 6;
          } else {
            rk4_step<2>(f, h, y, k, yn);
            nfe += 4;
          }
          if (!all_finite<2>(yn) || !(EEst == EEst)) {
            if (o.adaptive && dt > o.dtmin) { nrej++; dt = dt * (double)o.qmin; break; }
            ret = LDE_RET_NONFINITE;
            active = false;
            break;
          }
          if (o.adaptive) {
            // PI controller on l = log₂ EEst = ½ log₂ EEst²: q = EEst^β₁ · qold^(−β₂) = 2^(β₁ l − β₂ l_old) — no square root
            // and one v_log / v_exp pair per step instead of two (EEst = 0 ⇒ l = −∞ ⇒ q = q_lo, as pi_q has it)
            const float l = 0.5f * __builtin_amdgcn_logf(EEst);
            if (EEst > 1.0f) {
              nrej++;
              const float q11 = __builtin_amdgcn_exp2f(o.beta1 * l);
              dt = dt * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
              if (dt < o.dtmin) { ret = LDE_RET_DTMIN; active = false; }
              break;
            }
            const float q = fmaxf(o.q_lo, fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l - o.beta2 * lqold) * o.inv_gamma));
            lqold = fmaxf(l, -13.287712379549449f);
            dtp = dt * (double)fast_rcp(q);
            if (dtp > dtmax) dtp = dtmax;
          }
          nacc++;
          {   // record the accepted step
            f32x4* r = reinterpret_cast<f32x4*>(rec_at(n));
            r[0] = f32x4{h, y[0], y[1], 0.f};
            r[1] = f32x4{k[0][0], k[0][1], k[1][0], k[1][1]};
            r[2] = f32x4{k[2][0], k[2][1], k[3][0], k[3][1]};
            r[3] = f32x4{k[4][0], k[4][1], k[5][0], k[5][1]};
            const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
            r[4] = f32x4{k[6][0], k[6][1], __uint_as_float((unsigned)tb), __uint_as_float((unsigned)(tb >> 32))};
            n++;
          }
          y[0] = yn[0];
          y[1] = yn[1];
          kf[0] = k[FS][0];
          kf[1] = k[FS][1];
          t = last ? tend : t + dt;
          dt = o.adaptive ? dtp : o.dt_fixed;
          if (!(t < tend)) active = false;
        } while (0);
      }
      // sentinel: where the trajectory stands now (end time and end state of its last record)
      *reinterpret_cast<double*>(rec_at(n) + 18) = t;
      rec_at(n)[1] = y[0];
      rec_at(n)[2] = y[1];
      s_cnt[lane] = n;
      const bool any_active = __any(active);
      if (lane == 0) s_done = any_active ? 0 : 1;
    }
    __syncthreads();
    {   // save phase, all waves
      const int cnt = s_cnt[lane];
      int n2 = 0, pn = -1;
      if (jq < jend && cnt > 1) {   // first record whose end time reaches this wave's first save time (bisection: ≤ 5 LDS reads)
        const double tj0 = s_ts(jq);
        int lo = 0, hi = cnt;       // invariant: end(lo − 1) < tj0; answer in [lo, hi]
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rec_t(mid + 1) < tj0) lo = mid + 1;
          else hi = mid;
        }
        n2 = lo;
      }
      float h = 0.f, rh = 0.f, ys[2] = {0.f, 0.f}, k0[2] = {0.f, 0.f}, kE[2] = {0.f, 0.f}, P[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      double tn = 0.0;
      while (jq < jend) {
        const double tj = s_ts(jq);
        while (n2 < cnt && rec_t(n2 + 1) < tj) n2++;
        if (n2 == cnt) break;   // beyond what has been integrated so far
        const double t1 = rec_t(n2 + 1);
        float2 out;
        if (tj >= t1) {   // the save time is the step's end: the next record's start state
          const float* r1 = rec_at(n2 + 1);
          out = make_float2(r1[1], r1[2]);
        } else {
          if (pn != n2) {
            const f32x4* r = reinterpret_cast<const f32x4*>(rec_at(n2));
            const f32x4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
            h = q0[0];
            rh = fast_rcp(h);
            ys[0] = q0[1];
            ys[1] = q0[2];
            tn = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(q4[3]) << 32) | __float_as_uint(q4[2])));
            k0[0] = q1[0];
            k0[1] = q1[1];
            if (SOLVER == LDE_SOLVER_TSIT5) {
              const float kk[7][2] = {{q1[0], q1[1]}, {q1[2], q1[3]}, {q2[0], q2[1]}, {q2[2], q2[3]}, {q3[0], q3[1]}, {q3[2], q3[3]}, {q4[0], q4[1]}};
              tsit5_dense_coeffs<2>(kk, P);
            } else {
              kE[0] = q3[0];   // k₅ = f(yₙ₊₁), the slope at the end of the step
              kE[1] = q3[1];
            }
            pn = n2;
          }
          const float th = (float)(tj - tn) * rh;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            out.x = tsit5_dense_eval<2>(th, h, ys[0], k0[0], P[0][0], P[1][0], P[2][0]);
            out.y = tsit5_dense_eval<2>(th, h, ys[1], k0[1], P[0][1], P[1][1], P[2][1]);
          } else {  // cubic Hermite between (y,k1) and (yn,f(yn))
            const float* r1 = rec_at(n2 + 1);
            const float y1a = r1[1], y1b = r1[2];
            const float om = 1.0f - th;
            const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
            const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
            out.x = h00 * ys[0] + (h10 * h) * k0[0] + h01 * y1a + (h11 * h) * kE[0];
            out.y = h00 * ys[1] + (h10 * h) * k0[1] + h01 * y1b + (h11 * h) * kE[1];
          }
        }
        if (valid) z_out[(size_t)jq * B + b] = out;
        jq++;
      }
    }
    const int done = s_done;
    __syncthreads();   // the records are overwritten in the next round
    if (done) break;
  }
  if (w != 0 || !valid) return;
  if (ret != LDE_RET_SUCCESS) {  // failed solve ⇒ NaN block, never an error [REF GOKU.jl:114]
    const float qn = __int_as_float(0x7fc00000);
    for (int j = 0; j < T; j++) z_out[(size_t)j * B + b] = make_float2(qn, qn);
  }
  if (retcode) retcode[b] = ret;
  st_ret[b] = ret;
  st_nfe[b] = nfe;
  st_nacc[b] = nacc;
  st_nrej[b] = nrej;
}

// ---- adjoint --------------------------------------------------------------------------------------
// Reverse-time integration of [z, λ, g_L] from t_T to t_1 with a forced stop at every save time:
// λ += Δ_j there, and (checkpointed mode) z is reset to the saved ẑ(t_j).
template <int KIND, int SOLVER, bool TS_LDS>
__global__ void __launch_bounds__(256) k_pend_adjoint(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                      const double* __restrict__ ts_g, KOpts o,
                                                      const float2* __restrict__ dz_out, float2* __restrict__ dz0,
                                                      float* __restrict__ dtheta, int32_t* __restrict__ st_nfe,
                                                      int32_t* __restrict__ st_nacc, int32_t* __restrict__ st_nrej,
                                                      int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];
  const int T = o.T, B = o.B;
  if (TS_LDS)
    for (int i = threadIdx.x; i < T; i += blockDim.x) s_lds[i] = ts_g[i];
  __syncthreads();
  auto s_ts = [&](int i) -> double { return TS_LDS ? s_lds[i] : ts_g[i]; };   // compile-time choice
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;

  PendBwd<KIND> f(theta[b]);
  const float2 zT = z_out[(size_t)(T - 1) * B + b];
  const float2 dT = dz_out[(size_t)(T - 1) * B + b];
  float y[5] = {zT.x, zT.y, dT.x, dT.y, 0.f};
  float k[7][5], yn[5];
  int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;
  bool bad = !(isfinite(zT.x) && isfinite(zT.y));  // failed forward trajectory: NaN block is a constant ⇒ zero gradient

  if (T > 1 && !bad) {
    double t = s_ts(T - 1);
    const double dtmax = fabs(t - s_ts(0));
    int j = T - 2;
    // prefetch the jump data of the next stop
    float2 zc = z_out[(size_t)j * B + b], dc = dz_out[(size_t)j * B + b];
    f(y, k[0]);
    nfe = 1;
    double dt;
    if (o.adaptive) {
      if (o.dt_fixed > 0) dt = fmin(o.dt_fixed, dtmax);
      else {
        dt = init_dt<5>(f, y, k[0], -1.0f, dtmax, o);
        nfe++;
      }
    } else
      dt = o.dt_fixed;
    float qold = 1e-4f;
    long long iters = 0;
    while (j >= 0) {
      if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
      const double tstop = s_ts(j);
      const double dist = t - tstop;
      double hmag = dt;
      bool hit = false;
      if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = true; }
      const float h = -(float)hmag;
      float EEst = 0.f;
      if (SOLVER == LDE_SOLVER_TSIT5) {
        EEst = tsit5_attempt<5>(f, h, y, k, yn, o);
        nfe += 6;
      } else {
        rk4_step<5>(f, h, y, k, yn);
        nfe += 4;
      }
      if (!all_finite<5>(yn) || !(EEst == EEst)) {
        if (o.adaptive && hmag > o.dtmin) { nrej++; dt = hmag * (double)o.qmin; continue; }
        ret = LDE_RET_NONFINITE;
        break;
      }
      double dtp = dt;
      if (o.adaptive) {
        float q11;
        const float q = pi_q(EEst, qold, o, q11);
        if (EEst > 1.0f) {
          nrej++;
          dt = hmag * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
          if (dt < o.dtmin) { ret = LDE_RET_DTMIN; break; }
          continue;
        }
        qold = fmaxf(EEst, 1e-4f);
        dtp = hmag * (double)fast_rcp(q);
        if (dtp > dtmax) dtp = dtmax;
      }
      nacc++;
#pragma unroll
      for (int i = 0; i < 5; i++) y[i] = yn[i];
      constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;
      if (hit) {
        t = tstop;
        y[2] += dc.x;
        y[3] += dc.y;
        if (o.checkpoint) {
          y[0] = zc.x;
          y[1] = zc.y;
        }
        j--;
        if (j >= 0) {
          zc = z_out[(size_t)j * B + b];
          dc = dz_out[(size_t)j * B + b];
          f(y, k[0]);  // the jump invalidates the FSAL slope
          nfe++;
        }
      } else {
        t -= hmag;
#pragma unroll
        for (int i = 0; i < 5; i++) k[0][i] = k[FS][i];
      }
      dt = o.adaptive ? dtp : o.dt_fixed;
    }
  }
  if (ret != LDE_RET_SUCCESS) bad = true;
  dz0[b] = bad ? make_float2(0.f, 0.f) : make_float2(y[2], y[3]);
  dtheta[b] = bad ? 0.f : y[4];
  st_nfe[b] = nfe;
  st_nacc[b] = nacc;
  st_nrej[b] = nrej;
  st_ret[b] = bad ? (ret ? ret : LDE_RET_NONFINITE) : 0;
}


// ---- parallel-in-time checkpointed adjoint (LDE_SENSE_PARALLEL_CHECKPOINTED) ---------------------------------------
// With z reset to the saved ẑ(t_j) at every save time, the T−1 save intervals are independent, and λ (and g) enter
// the reverse-time system linearly. Phase 1 gives every (trajectory b, interval j) pair its own lane, which integrates
// [z | λᵃ λᵇ | gᵃ gᵇ] (λᵃ(t_{j+1}) = e₁, λᵇ(t_{j+1}) = e₂) from t_{j+1} down to t_j: the 2×2 transition matrix M_j and
// the row n_j of the parameter-gradient functional. Phase 2 composes them with a 49-term scan per trajectory:
//   g += n_j·λ ;  λ ← M_j λ + Δ_j.
// The sequential chain of ≥49 Tsit5 steps per trajectory (k_pend_adjoint) becomes ≈1–2 steps + 49·8 FMAs, and a batch
// of 256 trajectories fills 196 wavefronts instead of 4.
template <int KIND>
struct PendBasis {
  float ngl, gl2;
  __device__ __forceinline__ explicit PendBasis(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)) {}
  // y = [z0 z1 | la0 la1 lb0 lb1 | ga gb]
  __device__ __forceinline__ void operator()(const float (&y)[8], float (&dy)[8]) const {
    float s, c;
    hw_sincos(y[0], s, c);
    dy[0] = y[1];
    float acc = ngl * s;
    if (KIND == 1) acc -= 0.7f * y[1];
    dy[1] = acc;
    const float nc = ngl * c, gs = gl2 * s;
    dy[2] = -(nc * y[3]);
    float va = y[2];
    if (KIND == 1) va -= 0.7f * y[3];
    dy[3] = -va;
    dy[4] = -(nc * y[5]);
    float vb = y[4];
    if (KIND == 1) vb -= 0.7f * y[5];
    dy[5] = -vb;
    dy[6] = -(gs * y[3]);
    dy[7] = -(gs * y[5]);
  }
};

// integrate one save interval [t0, t1] backwards for the basis state; returns retcode, leaves the operator in y[2..7]
template <int KIND, int SOLVER>
__device__ __forceinline__ int pend_interval_operator(float2 zc, float L, double t0, double t1, const KOpts& o, float (&y)[8],
                                                      int& nacc, int& nrej) {
  PendBasis<KIND> f(L);
  y[0] = zc.x; y[1] = zc.y; y[2] = 1.f; y[3] = 0.f; y[4] = 0.f; y[5] = 1.f; y[6] = 0.f; y[7] = 0.f;
  float k[7][8], yn[8];
  int ret = LDE_RET_SUCCESS;
  nacc = 0;
  nrej = 0;
  if (!(isfinite(zc.x) && isfinite(zc.y))) return LDE_RET_NONFINITE;
  const double len = t1 - t0;
  double t = t1, dt = o.adaptive ? len : o.dt_fixed;   // first attempt: the whole interval
  const double dtmax = len;
  float qold = 1e-4f;
  long long iters = 0;
  f(y, k[0]);
  for (;;) {
    if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
    const double dist = t - t0;
    double hmag = dt;
    bool hit = false;
    if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = true; }
    const float h = -(float)hmag;
    float EEst = 0.f;
    if (SOLVER == LDE_SOLVER_TSIT5) EEst = tsit5_attempt<8>(f, h, y, k, yn, o);
    else rk4_step<8>(f, h, y, k, yn);
    if (!all_finite<8>(yn) || !(EEst == EEst)) {
      if (o.adaptive && hmag > o.dtmin) { nrej++; dt = hmag * (double)o.qmin; continue; }
      ret = LDE_RET_NONFINITE;
      break;
    }
    double dtp = dt;
    if (o.adaptive && (EEst > 1.0f || !hit)) {   // an accepted step that ends the interval needs no next step size
      float q11;
      const float q = pi_q(EEst, qold, o, q11);
      if (EEst > 1.0f) {
        nrej++;
        dt = hmag * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
        if (dt < o.dtmin) { ret = LDE_RET_DTMIN; break; }
        continue;
      }
      qold = fmaxf(EEst, 1e-4f);
      dtp = hmag * (double)fast_rcp(q);
      if (dtp > dtmax) dtp = dtmax;
    }
    nacc++;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = yn[i];
    if (hit) break;
    t -= hmag;
    constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;
#pragma unroll
    for (int i = 0; i < 8; i++) k[0][i] = k[FS][i];
    dt = o.adaptive ? dtp : o.dt_fixed;
  }
  return ret;
}

// two-kernel form (large batches: every access coalesced over the batch index)
template <int KIND, int SOLVER>
__global__ void __launch_bounds__(256) k_pend_adjoint_par1(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                           const double* __restrict__ ts_g, KOpts o,
                                                           float* __restrict__ ops, int32_t* __restrict__ info) {
  const int T = o.T, B = o.B;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (T - 1) * B) return;
  const int j = gid / B, b = gid - j * B;   // lanes of a wave share j and cover consecutive b (coalesced)
  float y[8];
  int nacc, nrej;
  const int ret = pend_interval_operator<KIND, SOLVER>(z_out[(size_t)(j + 1) * B + b], theta[b], ts_g[j], ts_g[j + 1], o, y,
                                                       nacc, nrej);
  // operator of interval j for trajectory b: [la0 la1 lb0 lb1 ga gb], plane-major so that phase 2 reads coalesced
  const size_t plane = (size_t)(T - 1) * B, at = (size_t)j * B + b;
#pragma unroll
  for (int i = 0; i < 6; i++) ops[(size_t)i * plane + at] = y[2 + i];
  info[at] = (ret << 24) | (min(nrej, 4095) << 12) | min(nacc, 4095);
}

// fused form (small/medium batches): one workgroup per trajectory, one lane per save interval; the interval operators
// never leave registers — they are composed by an order-preserving tree reduction over the wave (affine maps
// (λ,g) ↦ (Mλ+Δ, g+n·λ+γ) compose associatively), then across waves through LDS.
struct AffOp {
  float m00, m01, m10, m11;   // M (λ' = Mλ + d)
  float d0, d1;
  float n0, n1, gam;          // g' = g + n·λ + γ
};
// apply `a` first, then `b`
__device__ __forceinline__ AffOp aff_compose(const AffOp& a, const AffOp& b) {
  AffOp r;
  r.m00 = b.m00 * a.m00 + b.m01 * a.m10;
  r.m01 = b.m00 * a.m01 + b.m01 * a.m11;
  r.m10 = b.m10 * a.m00 + b.m11 * a.m10;
  r.m11 = b.m10 * a.m01 + b.m11 * a.m11;
  r.d0 = b.m00 * a.d0 + b.m01 * a.d1 + b.d0;
  r.d1 = b.m10 * a.d0 + b.m11 * a.d1 + b.d1;
  r.n0 = a.n0 + b.n0 * a.m00 + b.n1 * a.m10;
  r.n1 = a.n1 + b.n0 * a.m01 + b.n1 * a.m11;
  r.gam = a.gam + b.gam + b.n0 * a.d0 + b.n1 * a.d1;
  return r;
}
__device__ __forceinline__ AffOp aff_shfl_down(const AffOp& a, int off) {
  AffOp r;
  r.m00 = __shfl_down(a.m00, off); r.m01 = __shfl_down(a.m01, off); r.m10 = __shfl_down(a.m10, off);
  r.m11 = __shfl_down(a.m11, off); r.d0 = __shfl_down(a.d0, off); r.d1 = __shfl_down(a.d1, off);
  r.n0 = __shfl_down(a.n0, off); r.n1 = __shfl_down(a.n1, off); r.gam = __shfl_down(a.gam, off);
  return r;
}

template <int KIND, int SOLVER>
__global__ void __launch_bounds__(1024) k_pend_adjoint_fused(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                             const double* __restrict__ ts_g, KOpts o,
                                                             const float2* __restrict__ dz_out, float2* __restrict__ dz0,
                                                             float* __restrict__ dtheta, int32_t* __restrict__ st_nfe,
                                                             int32_t* __restrict__ st_nacc, int32_t* __restrict__ st_nrej,
                                                             int32_t* __restrict__ st_ret) {
  __shared__ AffOp s_op[16];
  __shared__ int s_stat[16][3];
  const int T = o.T, B = o.B;
  // XCD-aware trajectory ↔ workgroup map: workgroups are dealt round-robin to the 8 XCDs, and a 128-B line of ẑ/Δẑ holds
  // 16 neighbouring trajectories, so give each XCD a contiguous range of b (otherwise every XCD's L2 fetches every line:
  // 8× the algorithmic read traffic, measured with FETCH_SIZE). Speed only — any placement is correct.
  const int chunk = gridDim.x >> 3;
  const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (b >= B) return;
  const int l = threadIdx.x;              // l-th interval counted from the end: j = T-2-l (applied first ⇒ lowest lane)
  const int j = T - 2 - l;
  const int lane = l & 63, wave = l >> 6, nwave = (blockDim.x + 63) >> 6;
  AffOp op = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // identity for lanes beyond the first interval
  int nacc = 0, nrej = 0, ret = 0;
  if (j >= 0) {
    float y[8];
    ret = pend_interval_operator<KIND, SOLVER>(z_out[(size_t)(j + 1) * B + b], theta[b], ts_g[j], ts_g[j + 1], o, y, nacc, nrej);
    const float2 d = dz_out[(size_t)j * B + b];
    op = AffOp{y[2], y[4], y[3], y[5], d.x, d.y, y[6], y[7], 0.f};   // λ' = λ₀·(la) + λ₁·(lb) + Δ_j ; g' = g + ga λ₀ + gb λ₁
  }
  // order-preserving tree reduction inside the wave: lane ℓ ← (ℓ's block first, then the block of ℓ+off)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const AffOp hi = aff_shfl_down(op, off);
    const int a2 = __shfl_down(nacc, off), r2 = __shfl_down(nrej, off), e2 = __shfl_down(ret, off);
    if ((lane & (2 * off - 1)) == 0) {
      op = aff_compose(op, hi);
      nacc += a2;
      nrej += r2;
      ret = ret ? ret : e2;
    }
  }
  if (lane == 0) {
    s_op[wave] = op;
    s_stat[wave][0] = nacc;
    s_stat[wave][1] = nrej;
    s_stat[wave][2] = ret;
  }
  __syncthreads();
  if (l == 0) {
    for (int w = 1; w < nwave; w++) {
      op = aff_compose(op, s_op[w]);
      nacc += s_stat[w][0];
      nrej += s_stat[w][1];
      ret = ret ? ret : s_stat[w][2];
    }
    const float2 zT = z_out[(size_t)(T - 1) * B + b];
    const float2 dT = dz_out[(size_t)(T - 1) * B + b];
    if (!(isfinite(zT.x) && isfinite(zT.y))) ret = ret ? ret : LDE_RET_NONFINITE;
    const float l0 = op.m00 * dT.x + op.m01 * dT.y + op.d0;
    const float l1 = op.m10 * dT.x + op.m11 * dT.y + op.d1;
    const float g = op.n0 * dT.x + op.n1 * dT.y + op.gam;
    dz0[b] = ret ? make_float2(0.f, 0.f) : make_float2(l0, l1);
    dtheta[b] = ret ? 0.f : g;
    st_nacc[b] = nacc;
    st_nrej[b] = nrej;
    st_nfe[b] = (T - 1) + 6 * (nacc + nrej);
    st_ret[b] = ret;
  }
}

__global__ void __launch_bounds__(256) k_pend_adjoint_par2(const float2* __restrict__ z_out, const float2* __restrict__ dz_out,
                                                           const float* __restrict__ ops, const int32_t* __restrict__ info,
                                                           int T, int B, float2* __restrict__ dz0, float* __restrict__ dtheta,
                                                           int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                           int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float2 zT = z_out[(size_t)(T - 1) * B + b];
  const float2 dT = dz_out[(size_t)(T - 1) * B + b];
  float l0 = dT.x, l1 = dT.y, g = 0.f;
  int ret = (isfinite(zT.x) && isfinite(zT.y)) ? 0 : LDE_RET_NONFINITE;
  int nacc = 0, nrej = 0;
  const size_t plane = (size_t)(T - 1) * B;
  for (int j = T - 2; j >= 0; j--) {
    const size_t at = (size_t)j * B + b;
    const float la0 = ops[at], la1 = ops[plane + at], lb0 = ops[2 * plane + at], lb1 = ops[3 * plane + at];
    const float ga = ops[4 * plane + at], gb = ops[5 * plane + at];
    const float2 d = dz_out[at];
    const int inf = info[at];
    g += ga * l0 + gb * l1;
    const float n0 = la0 * l0 + lb0 * l1 + d.x;
    const float n1 = la1 * l0 + lb1 * l1 + d.y;
    l0 = n0;
    l1 = n1;
    nacc += inf & 4095;
    nrej += (inf >> 12) & 4095;
    if (!ret) ret = inf >> 24;
  }
  dz0[b] = ret ? make_float2(0.f, 0.f) : make_float2(l0, l1);
  dtheta[b] = ret ? 0.f : g;
  st_nacc[b] = nacc;
  st_nrej[b] = nrej;
  st_nfe[b] = (T - 1) + 6 * (nacc + nrej);
  st_ret[b] = ret;
}

// ---- host-side launchers (called from lde_api.cpp) -------------------------------------------------
static inline int pick_block(int B) { return B <= 4096 ? 64 : 256; }

int launch_pend_forward(int kind, int solver, const float* z0, const float* theta, const double* ts_dev, const KOpts& o,
                        float* z_out, int32_t* retcode, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                        hipStream_t stream) {
  const int block = pick_block(o.B), grid = (o.B + block - 1) / block;
  const size_t shm = o.T <= TS_LDS_MAX ? (size_t)o.T * sizeof(double) : 0;
  // small batches: stepping and dense output on different waves of a 64-trajectory workgroup (k_pend_forward_ws)
  static const bool ws_on = [] { const char* e = getenv("LDE_PEND_WS"); return !e || atoi(e) != 0; }();
  static const int ws_max_b = [] { const char* e = getenv("LDE_PEND_WS_MAX_B"); return e ? atoi(e) : 16384; }();   // measured (abl/pend_B.py): 21.6 vs 31.2 µs at 16384, 36.8 vs 32.6 µs at 32768
  if (ws_on && shm && o.T > 2 && o.B <= ws_max_b) {
    const size_t lds = (size_t)((o.T + 1) & ~1) * sizeof(double) + (size_t)(WS_CAP + 1) * 64 * WS_RW * sizeof(float);
    const int g64 = (o.B + 63) / 64;
#define LDE_LAUNCH_WS(K, S)                                                                                            \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)k_pend_forward_ws<K, S>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                              160 * 1024 - 512) != hipSuccess)                                                         \
        return LDE_ERR_HIP;                                                                                            \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_pend_forward_ws<K, S>), dim3(g64), dim3(WS_THREADS), lds, stream, (const float2*)z0, theta, ts_dev, \
                       o, (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                              \
  } while (0)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_WS(0, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH_WS(0, LDE_SOLVER_RK4);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_WS(1, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH_WS(1, LDE_SOLVER_RK4);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH_WS
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
#define LDE_LAUNCH(K, S)                                                                                              \
  do {                                                                                                                \
    if (shm)                                                                                                          \
      hipLaunchKernelGGL((k_pend_forward<K, S, true>), dim3(grid), dim3(block), shm, stream, (const float2*)z0, theta,  \
                         ts_dev, o, (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                   \
    else                                                                                                              \
      hipLaunchKernelGGL((k_pend_forward<K, S, false>), dim3(grid), dim3(block), 0, stream, (const float2*)z0, theta,   \
                         ts_dev, o, (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                   \
  } while (0)
  if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
  else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

int launch_pend_adjoint(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev,
                        const KOpts& o, const float* dz_out, float* dz0, float* dtheta, int32_t* nfe, int32_t* nacc,
                        int32_t* nrej, int32_t* ret, hipStream_t stream) {
  const int block = pick_block(o.B), grid = (o.B + block - 1) / block;
  const size_t shm = o.T <= TS_LDS_MAX ? (size_t)o.T * sizeof(double) : 0;
#define LDE_LAUNCH(K, S)                                                                                          \
  do {                                                                                                              \
    if (shm)                                                                                                        \
      hipLaunchKernelGGL((k_pend_adjoint<K, S, true>), dim3(grid), dim3(block), shm, stream, (const float2*)z_out,    \
                         theta, ts_dev, o, (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret);      \
    else                                                                                                            \
      hipLaunchKernelGGL((k_pend_adjoint<K, S, false>), dim3(grid), dim3(block), 0, stream, (const float2*)z_out,     \
                         theta, ts_dev, o, (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret);      \
  } while (0)
  if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
  else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

int launch_pend_adjoint_par(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev,
                            const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* ops, int32_t* info,
                            int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret, hipStream_t stream) {
  static const int fused_max_b = [] { const char* e = getenv("LDE_FUSED_MAX_B"); return e ? atoi(e) : 32768; }();
  if (o.T > 1 && o.T - 1 <= 1024 && o.B <= fused_max_b) {   // fused: one workgroup per trajectory, one lane per interval
    const int block = ((o.T - 1 + 63) / 64) * 64;
#define LDE_LAUNCH(K, S)                                                                                                  \
  hipLaunchKernelGGL((k_pend_adjoint_fused<K, S>), dim3(((o.B + 7) / 8) * 8), dim3(block), 0, stream, (const float2*)z_out, theta, ts_dev, o, \
                     (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  if (o.T > 1) {
    const long long n = (long long)(o.T - 1) * o.B;
    const int block = n <= 65536 ? 64 : 256;
    const int grid = (int)((n + block - 1) / block);
#define LDE_LAUNCH(K, S)                                                                                                 \
  hipLaunchKernelGGL((k_pend_adjoint_par1<K, S>), dim3(grid), dim3(block), 0, stream, (const float2*)z_out, theta, ts_dev, o, \
                     ops, info)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
    if (hipGetLastError() != hipSuccess) return LDE_ERR_HIP;
  }
  const int block2 = pick_block(o.B), grid2 = (o.B + block2 - 1) / block2;
  hipLaunchKernelGGL(k_pend_adjoint_par2, dim3(grid2), dim3(block2), 0, stream, (const float2*)z_out, (const float2*)dz_out, ops,
                     info, o.T, o.B, (float2*)dz0, dtheta, nfe, nacc, nrej, ret);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

}  // namespace lde

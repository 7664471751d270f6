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
#include "lde_host.h"

namespace lde {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- the closed menu of 2-state / 1-parameter physics RHS -------------------------------------
// KIND 0: du = [y, -(G/L) sin x]                    [REF examples/pendulum_friction-less/pendulum.jl:19-26]
// KIND 1: du = [y, -(G/L) sin x - (b/m) y], b/m=0.7  [REF pendulum.jl:65-74]
template <int KIND>
struct PendFwd {
  float ngl;   // -G/L
  float noff;  // −(whole turns of the step's start angle): see turn_anchor (lde_device.h)
  __device__ __forceinline__ explicit PendFwd(float L) : ngl(-10.0f / L), noff(0.f) {}  // one IEEE division per trajectory
  __device__ __forceinline__ void anchor(float x0) { noff = turn_anchor(x0); }
  __device__ __forceinline__ f32x2 ev(f32x2 y) const {   // the same right-hand side on a register pair (v_pk_* arithmetic around it)
    float acc = ngl * hw_sin(y.x, noff);
    if (KIND == 1) acc -= 0.7f * y.y;
    return f32x2{y.y, acc};
  }
  __device__ __forceinline__ void operator()(const float (&y)[2], float (&dy)[2]) const {
    dy[0] = y[1];
    float acc = ngl * hw_sin(y[0], noff);
    if (KIND == 1) acc -= 0.7f * y[1];
    dy[1] = acc;
  }
};

// augmented reverse-time system on [z₀ z₁ λ₀ λ₁ g]:  ż=f, λ̇=-(∂f/∂z)ᵀλ, ġ=-(∂f/∂L)ᵀλ
template <int KIND>
struct PendBwd {
  float ngl, gl2;  // -G/L, G/L²
  float noff;
  __device__ __forceinline__ explicit PendBwd(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)), noff(0.f) {}
  __device__ __forceinline__ void anchor(float x0) { noff = turn_anchor(x0); }
  __device__ __forceinline__ void operator()(const float (&y)[5], float (&dy)[5]) const {
    float s, c;
    hw_sincos(y[0], s, c, noff);
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

// -DLDE_PEND_PROF=1 (diagnostic builds, abl/pend_prof.py): lane 0 of wave 0 of workgroup 0 of k_pend_forward_ws stamps the
// 100 MHz wall clock and the shader cycle counter at its phase boundaries; read back with lde_debug_pend_prof.
#ifndef LDE_PEND_PROF
#define LDE_PEND_PROF 0
#endif
#ifndef LDE_PEND_ABL
#define LDE_PEND_ABL 0   // diagnostic ablations of the stepping loop (abl/pend_prof.py): 1 no record writes, 2 no controller, 3 idle helpers, 4 fused adjoint without the interval integration, 5 k_pend_forward_sh without helpers
#endif
#if LDE_PEND_PROF
static __device__ long long g_pprof[32];
#define PPROF(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_pprof[2 * (i)] = wall_clock64(); g_pprof[2 * (i) + 1] = __builtin_readcyclecounter(); } } while (0)
#define PPROF_VAL(i, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pprof[i] = (v); } while (0)
#else
#define PPROF(i) do { } while (0)
#define PPROF_VAL(i, v) do { } while (0)
#endif

// One Tsit5 attempt on the 2-vector state held as ONE register pair: every stage sum is a chain of v_pk_fma_f32 with a
// scalar coefficient (no per-component instructions, no operand shuffles) — the stepping wave's instruction count is the
// metric kernel's duration. Same tableau, same order of operations per component as tsit5_attempt<2>; k[0] = f(y) on entry;
// leaves k[1..6], yn; returns the mean square of the scaled error (EEst²; 0 when !ADAPT).
// max(|a|, |b|) as the one instruction it is (fmaxf(fabsf(a), fabsf(b)) compiles to a canonicalising v_max |a|, |a| in front of it when a
// is a loop-carried value; the result is the same number).
__device__ __forceinline__ float max_abs(float a, float b) {
  float r;
  asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// max(a, b) of two numbers that are not NaN as one instruction (fmaxf canonicalises a loop-carried operand first: v_max a, a)
__device__ __forceinline__ float max_f(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <class F, bool ADAPT>
__device__ __forceinline__ float tsit5_attempt_pair(F& f, float h, f32x2 y, f32x2 (&k)[7], f32x2& yn, const KOpts& o) {
  // Column-oriented: as soon as a slope exists it is added to the sums of ALL later stages — independent instructions that fill the
  // wait states behind a packed FMA and behind v_sin, where the row-oriented form (finish one stage's sum, then evaluate) is a chain
  // of dependent v_pk_fma_f32 with an s_nop between every two. Every sum still receives its terms in the order j = 0, 1, …: the
  // same arithmetic, bit for bit.
  f32x2 acc[7];
#pragma unroll
  for (int s = 1; s <= 6; s++) acc[s] = k[0] * ts5::A[s][0];
#pragma unroll
  for (int j = 1; j < 6; j++) {
    k[j] = f.ev(y + acc[j] * h);
#pragma unroll
    for (int s = j + 1; s <= 6; s++) {
      acc[s] += k[j] * ts5::A[s][j];
      asm volatile("" : "+v"(acc[s]));   // keep the column order (the scheduler otherwise re-serialises the later stages' sums into chains)
    }
  }
  yn = y + acc[6] * h;
  k[6] = f.ev(yn);
  if (!ADAPT) return 0.f;
  f32x2 e = k[0] * ts5::BT[0];
#pragma unroll
  for (int j = 1; j < 7; j++) e += k[j] * ts5::BT[j];
  e *= h;
  const f32x2 sk = f32x2{max_abs(y.x, yn.x), max_abs(y.y, yn.y)} * o.reltol + o.abstol;
  const f32x2 r = e * f32x2{fast_rcp(sk.x), fast_rcp(sk.y)};
  const f32x2 r2 = r * r;
  return (r2.x + r2.y) * 0.5f;
}

// which kernel the last launch_pend_* call of this thread ran (lde_last_kernel: bench.py checks its committed profile summaries against it)
static thread_local const char* g_pend_last[2] = {"", ""};
const char* pend_last_kernel(int which) { return g_pend_last[which & 1]; }

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
    f.anchor(y[0]);
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
      f.anchor(y[0]);
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
      if (o.rec.n && nacc < o.rec.cap) {   // LDE_SENSE_DISCRETE / step tracing: the accepted step's start time, size and start state
        o.rec.t[(size_t)nacc * B + b] = t;
        o.rec.dt[(size_t)nacc * B + b] = dt;
        reinterpret_cast<float2*>(o.rec.y)[(size_t)nacc * B + b] = make_float2(y[0], y[1]);
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
  if (o.rec.n) o.rec.n[b] = ret == LDE_RET_SUCCESS ? nacc : 0;
}

// ---- forward, small batches: stepping and dense output on different waves, pipelined through LDS ------------------------
// At B ≤ 16384 (one 64-trajectory workgroup per CU) the launch is a handful of waves per CU and its duration is one wave's
// dependent-instruction chain. In k_pend_forward a third of that chain is the dense output: ≈ 75 wave-iterations (the
// per-step maximum over 64 lanes of the saves inside the step) of interpolation, f64 save-time compares and stores — 9.4 of
// 28.4 µs at T = 50 (abl/pend_T.py). Here wave 0 (the stepper) does nothing but step, and what it leaves behind per
// accepted step is THREE floats in LDS — the step size and the state after the step — followed by the trajectory's step
// count (LDS executes one wave's instructions in order, so a reader that sees the count sees the records; no barrier, no
// fence on the stepping chain). One wave's LDS writes cost ≈ 5 cycles per float (abl/valu_rate.hip): the full record of
// round 1 — h, y, k₁…k₇, t: 25 floats — was 216 of the 1130 cycles of a step (abl/pend_prof.py with -DLDE_PEND_ABL=1).
// The helper waves poll the counts and evaluate the saves, helper i of the nh helpers that do not share the stepper's
// SIMD a contiguous slice of the save grid for the same 64 trajectories, each lane as soon as ITS trajectory has passed the save
// time: the lane re-evaluates the six stages of the ONE step its save time falls in from (yₙ, hₙ) — same code, same
// inputs as the stepper's — and interpolates; start times are rebuilt by the same f64 additions the stepper makes. A
// helper serves 4–5 neighbouring save times, so it repeats ≈ 2 of the ≈ 19 steps, off the critical path. Only the saves inside the
// last step remain after the stepper is done (round 1: 2.2 µs of save phase + 0.7 µs of barriers behind the stepping loop).
// The record area holds WS_CAP steps; a trajectory that needs more goes through another round (one barrier pair per round).
// Same formulas as k_pend_forward; the two compilations contract multiply-adds differently, so they agree to the solver's
// tolerance, not bit for bit (tests/test_gpu_pendulum.py).
constexpr int WS_CAP = 96;       // accepted steps recorded per round
constexpr int WS_WAVES = 16;     // waves per workgroup: wave 0 steps, waves 1…15 evaluate saves
constexpr int WS_HELPERS = WS_WAVES - 1;
constexpr int WS_THREADS = 64 * WS_WAVES;
constexpr int WS_RW = 4;         // floats per record: {h, y₀, y₁ after the step, –}: 16 B per lane, conflict-free b128 / b64 accesses

template <int KIND, int SOLVER, bool ADAPT, bool REC>   // REC: the instantiation that writes step records (LDE_SENSE_DISCRETE, "step_trace")
__global__ void __launch_bounds__(WS_THREADS) k_pend_forward_ws(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                         const double* __restrict__ ts_g, KOpts o,
                                                         float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                         int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                         int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];
  __shared__ int s_cnt[64];      // records published per trajectory in this round
  __shared__ int s_fin;          // 0: the stepper is stepping; 1: its round is over, another follows; 2: all trajectories done
  __shared__ int s_fail;         // some trajectory of the workgroup failed (NaN block to be written after the helpers' stores)
  __shared__ int s_simd[WS_WAVES];   // the SIMD each wave landed on
  __shared__ __attribute__((aligned(16))) float s_base[64 * 4];   // where every trajectory stands at the start of the round: {t (f64), y₀, y₁}
  const int T = o.T, B = o.B, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  PPROF(0);
#if LDE_PEND_PROF
  if (blockIdx.x == 0 && tid == 0) { g_pprof[27] = 0; g_pprof[29] = 0; g_pprof[24] = 0; }
#endif
  const int b = blockIdx.x * 64 + lane;
  const bool valid = b < B;
  // inputs are requested before the save grid is staged: one memory latency instead of two
  float2 zi = make_float2(0.f, 0.f);
  float Lb = 1.0f;
  if (valid) {
    zi = z0[b];
    Lb = theta[b];
  }
  for (int i = tid; i < T; i += WS_THREADS) s_lds[i] = ts_g[i];
  float* rec = reinterpret_cast<float*>(s_lds + ((T + 1) & ~1));          // [WS_CAP][64][WS_RW]
  auto rec_at = [&](int n) -> float* { return rec + (size_t)(n * 64 + lane) * WS_RW; };
  if (tid < 64) s_cnt[tid] = 0;
  if (tid == 0) { s_fin = 0; s_fail = 0; }
  // HW_REG_HW_ID[5:4] = SIMD_ID (gfx9): a helper that shares the stepper's SIMD would take issue slots from the one wave whose
  // instruction chain IS the kernel's duration, so those waves stay idle
  if (lane == 0) s_simd[w] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  __syncthreads();
  PPROF(1);
  auto s_ts = [&](int i) -> double { return s_lds[i]; };
  constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // FSAL slope
  constexpr int NS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // RHS evaluations per attempt
  PendFwd<KIND> f(Lb);
  const double tend = T > 1 ? s_ts(T - 1) : 0.0;

  if (w == 0) {
    // ================= the stepper =================
    // t is f64 (the reference's time is Float64 on Float32 state), the step size proposal dt is f32: a step is taken with
    // h = (float)dt anyway, and only ONE f64 addition per accepted step (t += h) and one per attempt (tend − t) remain.
    f32x2 y = {zi.x, zi.y}, k[7], yn = {0.f, 0.f}, kf = {0.f, 0.f};   // register pairs; kf: f(y) at the current state (first-same-as-last slope)
    int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;   // nfe: the evaluations before the first step; the rest is NS × attempts
    double t = 0.0;
    float dt = 0.f, dtmax = 0.f;
    constexpr float LQ_MIN = -13.287712379549449f;   // log₂ of qoldinit = 1e-4
    float lqold = LQ_MIN;                            // the PI controller runs on log₂ EEst here (see below)
    const int maxit = o.maxiters > 0x7fffffffLL ? 0x7fffffff : (int)o.maxiters;
    const float dtmin = (float)o.dtmin;
    int iters = 0;
    bool active = false;
#pragma unroll
    for (int s = 0; s < 7; s++) k[s] = f32x2{0.f, 0.f};
    if (valid) {
      z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
      if (T > 1) {
        t = s_ts(0);
        const double dtmax_d = tend - t;
        dtmax = (float)dtmax_d;
        f.anchor(y.x);
        kf = f.ev(y);
        nfe = 1;
        if (ADAPT) {
          if (o.dt_fixed > 0) dt = (float)fmin(o.dt_fixed, dtmax_d);
          else {
            const float ya[2] = {y.x, y.y}, fa[2] = {kf.x, kf.y};
            dt = (float)init_dt<2>(f, ya, fa, 1.0f, dtmax_d, o);
            nfe++;
          }
        } else
          dt = (float)o.dt_fixed;
        active = t < tend && maxit > 0;
        if (t < tend && !active) ret = LDE_RET_MAXITERS;
      }
    }
    bool first_round = true;
    for (;;) {   // rounds
      int n = 0;
      float* rp = rec_at(0);
      if (!first_round) {   // (the first round's base is written by the helpers themselves: they hold z₀ and ts[0])
        *reinterpret_cast<double*>(&s_base[lane * 4]) = t;
        *reinterpret_cast<float2*>(&s_base[lane * 4 + 2]) = make_float2(y.x, y.y);
        __syncthreads();   // B: the counts are reset, the base is in place
      }
      first_round = false;
      const double t_round = t;   // (REC) where this round's first record starts
      const f32x2 y_round = y;
      int wave_iters = 0;
      PPROF(2);
      // The stepping loop is the launch's critical path. What one wave pays for on gfx950 (abl/valu_rate.hip): 4 cycles per
      // VALU instruction whether dependent or not, ≈ 8 per transcendental, ≈ 55 per TAKEN branch, ≈ 18 extra whenever a
      // lane mask goes VALU → SALU → VALU back to back, ≈ 5 per float written to LDS. So the loop is straight-line code with
      // ONE predicated block, and one float decides acceptance: mq = EEst² (+ NaN if the new state is not finite, + ∞ for
      // a lane that is not stepping: `pen`); accepted ⇔ mq ≤ 1. Everything rare (rejections, non-finite attempts, step-size
      // underflow) is out of line.
      float pen = (active && iters < maxit) ? 0.f : __builtin_inff();
      for (;;) {
        if (!__any(pen == 0.f)) break;
        wave_iters++;
        const float rem = (float)(tend - t);
        const bool last = dt >= rem * 0.99999988f;   // the step would reach (or pass) the end: land on it exactly
        const float h = last ? rem : dt;             // (the helpers recognise the last step by h == (float)(tend − t))
        k[0] = kf;
        f.anchor(y.x);
        float msq = 0.f;   // mean square of the scaled error estimate (EEst²)
        if (SOLVER == LDE_SOLVER_TSIT5) msq = tsit5_attempt_pair<PendFwd<KIND>, ADAPT>(f, h, y, k, yn, o);
        else {
          const float ya[2] = {y.x, y.y};
          float ka[7][2], yna[2];
          ka[0][0] = kf.x;
          ka[0][1] = kf.y;
          rk4_step<2>(f, h, ya, ka, yna);
#pragma unroll
          for (int s = 1; s <= 4; s++) k[s] = f32x2{ka[s][0], ka[s][1]};
          yn = f32x2{yna[0], yna[1]};
        }
        const float mq = fmaf(0.f, fabsf(yn.x) + fabsf(yn.y), msq + pen);   // ∞·0 = NaN: a non-finite state never passes
        const bool ok = mq <= 1.0f;
        // PI controller on l = log₂ EEst = ½ log₂ EEst²: q = EEst^β₁ · qold^(−β₂) = 2^(β₁ l − β₂ l_old) — no square root and
        // one v_log / v_exp pair per step; accepted: clamp(q/γ, 1/qmax, 1/qmin). EEst = 0 ⇒ l = −∞ ⇒ q = 1/qmax, as pi_q has it.
        float dtn = (float)o.dt_fixed, l = 0.f;
        if (ADAPT) {
#if LDE_PEND_ABL == 2
          dtn = dt;
#else
          l = 0.5f * __builtin_amdgcn_logf(msq);
          const float q = fmaxf(o.q_lo, fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l - o.beta2 * lqold) * o.inv_gamma));
          dtn = fminf(h * fast_rcp(q), dtmax);
#endif
        }
        if (__builtin_expect(__any(!ok && pen == 0.f), 0)) {   // rare: a rejected or non-finite attempt somewhere in the wave
          if (!ok && pen == 0.f) {
            const bool fin = (fabsf(yn.x) + fabsf(yn.y)) < __builtin_inff();   // false for NaN too
            nrej++;
            iters++;
            if (!ADAPT) { ret = LDE_RET_NONFINITE; active = false; nrej--; }
            else if (!fin) {
              if (h > dtmin) dt = h * o.qmin;
              else { ret = LDE_RET_NONFINITE; active = false; nrej--; }
            } else {   // rejected: dt / min(1/qmin, EEst^β₁/γ); a NaN estimate gives 1/qmin (v_min returns the other operand)
              dt = h * fast_rcp(fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l) * o.inv_gamma));
              if (dt < dtmin) { ret = LDE_RET_DTMIN; active = false; }
            }
            if (!active || iters >= maxit) pen = __builtin_inff();
          }
        }
        if (ok) {   // the accepted step: leave {h, yₙ₊₁} behind, publish, advance
#if LDE_PEND_ABL != 1
          *reinterpret_cast<f32x4*>(rp) = f32x4{h, yn.x, yn.y, 0.f};
#endif
          rp += 64 * WS_RW;
          n++;
          asm volatile("" ::: "memory");                                      // the count is published AFTER the record
          __hip_atomic_store(&s_cnt[lane], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a plain LDS store: ds_write_b32)
          nacc++;
          iters++;
          y = yn;
          kf = k[FS];
          t = last ? tend : t + (double)h;
          dt = dtn;
          lqold = fmaxf(l, LQ_MIN);
          active = !last;
          if (last || n >= WS_CAP || iters >= maxit) pen = __builtin_inff();
        }
      }
      if (active && iters >= maxit) { ret = LDE_RET_MAXITERS; active = false; }
      PPROF(3);
      PPROF_VAL(30, wave_iters);
      const bool more = __any(active);
      if (__any(ret != LDE_RET_SUCCESS) && lane == 0) __hip_atomic_store(&s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_fin, more ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (REC) {
        // The step records of this round (LDE_SENSE_DISCRETE / "step_trace"), written by the stepping wave while the helpers finish their
        // dense output: every lane walks ITS records in the ring — start state = the step before's end, start time = the stepper's own
        // f64 sum replayed — and the stores of a step are coalesced over the 64 trajectories.
        const int i0 = nacc - n;
        double tr = t_round;
        f32x2 yr = y_round;
        const float* rq = rec_at(0);
        for (int i = 0; __any(i < n); i++, rq += 64 * WS_RW) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(rq);
          if (valid && i < n && i0 + i < o.rec.cap) {
            o.rec.t[(size_t)(i0 + i) * B + b] = tr;
            o.rec.dt[(size_t)(i0 + i) * B + b] = (double)q[0];
            reinterpret_cast<float2*>(o.rec.y)[(size_t)(i0 + i) * B + b] = make_float2(yr.x, yr.y);
          }
          tr += (double)q[0];
          yr = f32x2{q[1], q[2]};
        }
      }
      if (!more) break;
      __syncthreads();   // A: the helpers have consumed this round's records
      s_cnt[lane] = 0;
      if (lane == 0) s_fin = 0;
    }
    PPROF(4);
    if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) __syncthreads();   // F: every helper store has been issued and waited for
    if (valid) {
      if (ret != LDE_RET_SUCCESS) {  // failed solve ⇒ NaN block, never an error [REF GOKU.jl:114]
        const float qn = __int_as_float(0x7fc00000);
        for (int j = 0; j < T; j++) z_out[(size_t)j * B + b] = make_float2(qn, qn);
      }
      if (retcode) retcode[b] = ret;
      st_ret[b] = ret;
      st_nfe[b] = nfe + NS * (nacc + nrej);
      st_nacc[b] = nacc;
      st_nrej[b] = nrej;
      if (REC) o.rec.n[b] = ret == LDE_RET_SUCCESS ? nacc : 0;
    }
    PPROF(7);
    return;
  }

  // ================= the helpers: dense output for save times rank+1, rank+1+nh, … of the workgroup's 64 trajectories =================
  int nh = 0, rank = 0;   // helpers that do not share the stepper's SIMD, and this wave's rank among them
  {
    const int s0 = s_simd[0];
    int ne = 0;
    for (int i = 1; i < WS_WAVES; i++) {
      const bool e = s_simd[i] != s0;
      if (i == w) rank = ne;
      ne += e ? 1 : 0;
    }
    nh = ne;
    if (nh == 0) { nh = WS_HELPERS; rank = w - 1; }          // (every wave on one SIMD: cannot happen with 16 waves; all help)
    else if (s_simd[w] == s0) rank = -1;
  }
  constexpr int T_none = 0x7fffffff;
#if LDE_PEND_ABL == 3
  rank = -1;
#endif
  // helper `rank` serves a CONTIGUOUS slice of the save grid: neighbouring save times mostly fall into the same step (≈ 2.6 per
  // step at the default tolerance), whose slopes are then rebuilt once — interleaved slices rebuilt one step per save time and
  // kept the helpers' SIMDs busy for 6 µs after the stepper had finished
  const int chunk = (T - 1 + nh - 1) / nh;
  int jq = rank < 0 ? T_none : 1 + rank * chunk;   // per lane from here on: a lane serves a save time as soon as ITS trajectory has passed it
  const int jend = rank < 0 ? 0 : min(T, 1 + (rank + 1) * chunk);
  // A wave with nothing (left) to serve leaves the kernel: s_barrier only counts the workgroup's surviving waves, so the
  // round / failure barriers below stay matched among those that remain, and nobody polls next to the stepper.
  // (Its stores are waited for first: a failed trajectory's NaN block is written by the stepper later.)
  if (rank < 0 || jq >= jend) return;
#if LDE_PEND_PROF
  if (blockIdx.x == 0 && threadIdx.x == 64) g_pprof[28] = nh;
#endif
  bool first_round = true;
  for (;;) {   // rounds
    // where the walk over this round's records starts: record n2 begins at time tn in state ys
    int n2 = 0, pn = -1;
    double tn = T > 1 ? s_ts(0) : 0.0;
    f32x2 ys = {zi.x, zi.y};
    if (!first_round) {
      __syncthreads();   // B: the stepper has reset the counts and written the base
      tn = *reinterpret_cast<const double*>(&s_base[lane * 4]);
      const float2 yb = *reinterpret_cast<const float2*>(&s_base[lane * 4 + 2]);
      ys = f32x2{yb.x, yb.y};
    }
    first_round = false;
    // the step the walk stands in: size, end time and end state (valid while n2 < cnt)
    float h = 0.f, rh = 0.f;
    double t1 = 0.0;
    f32x2 ye = {0.f, 0.f}, k0 = {0.f, 0.f}, kE = {0.f, 0.f}, P2 = {0.f, 0.f}, P3 = {0.f, 0.f}, P4 = {0.f, 0.f};
    int ld = -1;   // record whose {h, t1, ye} are loaded
    int fin;
    for (;;) {
#if LDE_PEND_PROF
      if (blockIdx.x == 0 && lane == 0 && jend == T) g_pprof[24]++;
#endif
      fin = __hip_atomic_load(&s_fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // read BEFORE the count: if the round is over, the count is final
      const int cnt = __hip_atomic_load(&s_cnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      while (jq < jend) {
        const double tj = s_ts(jq);
        bool have = false;
        while (n2 < cnt) {
          if (ld != n2) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(rec_at(n2));
            h = q[0];
            ye = f32x2{q[1], q[2]};
            t1 = (h == (float)(tend - tn)) ? tend : tn + (double)h;   // exactly the stepper's arithmetic
            ld = n2;
          }
          if (t1 >= tj) { have = true; break; }
          tn = t1;       // the save time lies beyond this step: move on
          ys = ye;
          n2++;
        }
        if (!have) break;   // beyond what has been integrated so far
        float2 out;
        if (tj >= t1) out = make_float2(ye.x, ye.y);   // the save time is the step's end
        else {
          if (pn != n2) {   // rebuild the step's slopes: the stepper's code on the stepper's inputs (yₙ, hₙ)
            f.anchor(ys.x);
            k0 = f.ev(ys);
            rh = fast_rcp(h);
            if (SOLVER == LDE_SOLVER_TSIT5) {
              f32x2 kk[7], ynr;
              kk[0] = k0;
              (void)tsit5_attempt_pair<PendFwd<KIND>, false>(f, h, ys, kk, ynr, o);
              P2 = kk[0] * ts5::R1[0];
              P3 = kk[0] * ts5::R1[1];
              P4 = kk[0] * ts5::R1[2];
#pragma unroll
              for (int s = 0; s < 6; s++) {
                P2 += kk[s + 1] * ts5::R[s][0];
                P3 += kk[s + 1] * ts5::R[s][1];
                P4 += kk[s + 1] * ts5::R[s][2];
              }
            } else
              kE = f.ev(ye);   // f(yₙ₊₁), the slope at the end of the step
            pn = n2;
          }
          const float th = (float)(tj - tn) * rh;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            out.x = tsit5_dense_eval<2>(th, h, ys.x, k0.x, P2.x, P3.x, P4.x);
            out.y = tsit5_dense_eval<2>(th, h, ys.y, k0.y, P2.y, P3.y, P4.y);
          } else {  // cubic Hermite between (y,k1) and (yn,f(yn))
            const float om = 1.0f - th;
            const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
            const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
            out.x = h00 * ys.x + (h10 * h) * k0.x + h01 * ye.x + (h11 * h) * kE.x;
            out.y = h00 * ys.y + (h10 * h) * k0.y + h01 * ye.y + (h11 * h) * kE.y;
          }
        }
        if (valid) z_out[(size_t)jq * B + b] = out;
#if LDE_PEND_PROF
        if (blockIdx.x == 0 && lane == 0 && jend == T && jend - jq <= 8) g_pprof[16 + (jend - jq - 1)] = wall_clock64();
#endif
        jq++;
      }
      if (fin || !__any(jq < jend)) break;
      // Twelve helpers share three SIMDs: a helper whose slice the stepper is still several steps away from must not spin
      // there (s_sleep's unit is ≈ 0.14 µs on gfx950, far longer than 64 core cycles).
      const bool near = jq < jend && (float)(s_ts(min(jq, T - 1)) - tn) < 3.0f * h;
      if (__any(near)) __builtin_amdgcn_s_sleep(1);
      else __builtin_amdgcn_s_sleep(4);
    }
    if (fin == 2 || !__any(jq < jend)) break;
    __syncthreads();   // A
  }
#if LDE_PEND_PROF
  if (blockIdx.x == 0 && lane == 0) {
    atomicMax(reinterpret_cast<unsigned long long*>(&g_pprof[29]), (unsigned long long)wall_clock64());
    if (rank == 0) g_pprof[26] = wall_clock64();
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- forward, the smallest batches: lanes = save times ---------------------------------------------------------------------
// At B ≤ 1024 the chip has more SIMDs than there are trajectories, so a wave is spent on TPW trajectories only (TPW = 1:
// one trajectory per wave): ALL 64/TPW lanes of a trajectory's group step it — redundantly, which costs nothing in a SIMT
// machine — and each lane OWNS save times (lane s of the group: j = 1+s, 1+s+64/TPW, …). After an accepted step every lane
// looks whether its save time fell into the step and, if so, interpolates straight from the slopes in its registers and
// stores. No LDS, no records, no helper waves, no barrier, no tail after the last step; with TPW = 1 the accept decision is
// wave-uniform and the wave runs its own trajectory's step count, not the maximum over 64. The dense output costs the step
// ≈ 45 instructions (the Θ-independent coefficients once, one evaluation per lane) instead of the wave-sequential save
// loop of k_pend_forward (≈ 75 iterations at T = 50). Same formulas as the other two forward kernels.
// ONE = true (T − 1 ≤ 64/TPW: a lane serves exactly one save time, fetched before the first step): the dense output is a single
// predicated block and the stepping loop contains no load at all. With the next save time's load in it the compiler must put
// `s_waitcnt vmcnt(0)` in front of the block, and vmcnt counts the previous step's ẑ STORES too — the stepping chain then waits
// for write acknowledgements it has no use for.
template <int KIND, int SOLVER, bool ADAPT, int TPW, bool ONE = false, int RING = 0, bool REC = false>   // REC (with RING: a lane per trajectory): step records at accept
__global__ void __launch_bounds__(64) k_pend_forward_tl(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                        const double* __restrict__ ts_g, KOpts o,
                                                        float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                        int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                        int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  constexpr int LPT = 64 / TPW;   // lanes per trajectory
  static_assert(RING == 0 || (TPW == 64 && !ONE && RING <= 32 && (RING & (RING - 1)) == 0), "the row ring is for one trajectory per lane");
  static_assert(!REC || RING > 0, "step records: the lane-per-trajectory form only");
  // RING > 0 (large batches, a lane owns a trajectory): the lanes of a wave pass a given save time at different iterations, so a
  // direct store is 64 lanes × 8 bytes into up to 64 different rows of ẑ, and the 512 contiguous bytes a wave owns in row j arrive
  // piecemeal over many iterations — with ≈ 25 KB of such half-written rows per resident wave and thousands of waves the L2s
  // cannot hold them until they are whole (measured at B = 2²⁰, T = 50: 1128 MB written for 419 MB of ẑ). Here a lane parks its
  // values in its own column of an LDS ring of RING rows, and the wave writes row jc — all 64 lanes, one full 512-byte store — as
  // soon as its slowest trajectory has passed it. A lane that runs more than RING rows ahead stores directly (bit clear in `inring`).
  __shared__ float2 s_ring[RING > 0 ? RING * 64 : 1];
  unsigned inring = 0;
  int jc = 1;                     // wave-uniform: the next row to leave the ring
  const int T = o.T, B = o.B, lane = threadIdx.x, slot = lane % LPT;
  // XCD-aware trajectory ↔ workgroup map (as in k_pend_adjoint_fused): workgroups are dealt round-robin to the 8 XCDs and a 128-byte
  // line of ẑ holds 16 neighbouring trajectories' values of one save time, each written by another wave — with neighbouring
  // trajectories on ONE XCD their partial writes meet in one L2 instead of reaching HBM as eight partial lines. Traffic only.
  const int chunk = gridDim.x >> 3;   // the grid is a multiple of 8 workgroups
  const int g = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const int b = g * TPW + lane / LPT;
  const bool valid = b < B;
  const int bc = valid ? b : B - 1;   // an out-of-range group computes on a copy of the last trajectory and stores nothing
  const double t_first = o.t_first, t_last = o.t_last;   // ts[0], ts[T−1] arrive with the kernel arguments: no load before the first step
  const float2 zi = z0[bc];
  PendFwd<KIND> f(theta[bc]);
  constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // FSAL slope
  constexpr int NS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // RHS evaluations per attempt
  const double dinf = __longlong_as_double(0x7ff0000000000000LL);
  // (RING) the save grid lives in LDS, one copy per wave: a global load in the dense-output loop is waited for with vmcnt, which
  // counts the ring's row stores too — the loop would stall on write acknowledgements. A wave's LDS operations execute in order:
  // no barrier between the staging and the first read.
  extern __shared__ __attribute__((aligned(16))) double s_tsl[];
  if (RING > 0)
    for (int i = lane; i < T; i += 64) s_tsl[i] = ts_g[i];
  auto ts_at = [&](int i) -> double { return RING > 0 ? s_tsl[i] : ts_g[i]; };
  int j = 1 + slot;                                                   // the save time this lane serves next,
  double tj = j < T ? ts_at(j) : dinf;                                // its value, and the one after it (prefetched:
  double tjn = j + LPT < T ? ts_at(j + LPT) : dinf;                   // a load on the stepping chain would cost a memory latency)
  float2* const dst1 = z_out + (size_t)(j < T ? j : 0) * B + (valid ? b : 0);
  f32x2 y = {zi.x, zi.y}, k[7], yn = {0.f, 0.f}, kf = {0.f, 0.f};
  int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;
  double t = 0.0, tend = 0.0;
  float dt = 0.f, dtmax = 0.f;
  constexpr float LQ_MIN = -13.287712379549449f;   // log₂ of qoldinit = 1e-4
  float lqold = LQ_MIN;
  const int maxit = o.maxiters > 0x7fffffffLL ? 0x7fffffff : (int)o.maxiters;
  const float dtmin = (float)o.dtmin;
  int iters = 0;
  bool active = false;
#pragma unroll
  for (int s = 0; s < 7; s++) k[s] = f32x2{0.f, 0.f};
  if (valid && slot == 0) z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
  if (T > 1) {
    t = t_first;
    tend = t_last;
    const double dtmax_d = tend - t;
    dtmax = (float)dtmax_d;
    f.anchor(y.x);
    kf = f.ev(y);
    nfe = 1;
    if (ADAPT) {
      if (o.dt_fixed > 0) dt = (float)fmin(o.dt_fixed, dtmax_d);
      else {
        const float ya[2] = {y.x, y.y}, fa[2] = {kf.x, kf.y};
        dt = (float)init_dt<2>(f, ya, fa, 1.0f, dtmax_d, o);
        nfe++;
      }
    } else
      dt = (float)o.dt_fixed;
    active = t < tend && maxit > 0;
    if (t < tend && !active) ret = LDE_RET_MAXITERS;
  }
  // the stepping loop of k_pend_forward_ws (see there): one float decides acceptance, rare cases out of line. With one trajectory per
  // wave (TPW = 1) every lane carries the same solve, so each decision is taken as a vote — the lane's own answer, but a scalar branch:
  // the loop's variables are then updated in place instead of through exec-masked copies (as in k_pend_forward_sh).
  constexpr bool UNI = TPW == 1 && RING == 0;
  auto vote = [](bool c) -> bool { return UNI ? (bool)__any(c) : c; };
  float pen = (active && iters < maxit) ? 0.f : __builtin_inff();
  for (;;) {
    if (!__any(pen == 0.f)) break;
    const float rem = (float)(tend - t);
    const bool last = vote(dt >= rem * 0.99999988f);
    // (RING) a trajectory that has run to within lb_hold rows of the ring's end sits this iteration out: its attempt is computed and
    // dropped like a stopped lane's, nothing of its state changes, and the rows it would have had to store piecemeal wait for the wave
    const float penh = (RING > 0 && j >= jc + RING - o.lb_hold && o.lb_hold > 0) ? __builtin_inff() : pen;
    const float h = last ? rem : dt;
    k[0] = kf;
    f.anchor(y.x);
    float msq = 0.f;
    if (SOLVER == LDE_SOLVER_TSIT5) msq = tsit5_attempt_pair<PendFwd<KIND>, ADAPT>(f, h, y, k, yn, o);
    else {
      const float ya[2] = {y.x, y.y};
      float ka[7][2], yna[2];
      ka[0][0] = kf.x;
      ka[0][1] = kf.y;
      rk4_step<2>(f, h, ya, ka, yna);
#pragma unroll
      for (int s = 1; s <= 4; s++) k[s] = f32x2{ka[s][0], ka[s][1]};
      yn = f32x2{yna[0], yna[1]};
    }
    const float mq = fmaf(0.f, fabsf(yn.x) + fabsf(yn.y), msq + penh);   // ∞·0 = NaN: a non-finite state never passes
    const bool ok = vote(mq <= 1.0f);
    float dtn = (float)o.dt_fixed, l = 0.f;
    if (ADAPT) {
      l = 0.5f * __builtin_amdgcn_logf(msq);
      const float q = fmaxf(o.q_lo, fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l - o.beta2 * lqold) * o.inv_gamma));
      dtn = fminf(h * fast_rcp(q), dtmax);
    }
    if (__builtin_expect(__any(!ok && penh == 0.f), 0)) {   // rare: a rejected or non-finite attempt
      if (vote(!ok && penh == 0.f)) {
        const bool fin = vote((fabsf(yn.x) + fabsf(yn.y)) < __builtin_inff());
        nrej++;
        iters++;
        if (!ADAPT) { ret = LDE_RET_NONFINITE; active = false; nrej--; }
        else if (!fin) {
          if (vote(h > dtmin)) dt = h * o.qmin;
          else { ret = LDE_RET_NONFINITE; active = false; nrej--; }
        } else {
          dt = h * fast_rcp(fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l) * o.inv_gamma));
          if (vote(dt < dtmin)) { ret = LDE_RET_DTMIN; active = false; }
        }
        if (!active || iters >= maxit) pen = __builtin_inff();
      }
    }
    if (ok) {
      const double tn = last ? tend : t + (double)h;
      if (REC && valid && nacc < o.rec.cap) {   // LDE_SENSE_DISCRETE / step tracing: the accepted step's start time, size and start state
        o.rec.t[(size_t)nacc * B + b] = t;
        o.rec.dt[(size_t)nacc * B + b] = (double)h;
        reinterpret_cast<float2*>(o.rec.y)[(size_t)nacc * B + b] = make_float2(y.x, y.y);
      }
      if (__any(tj <= tn)) {   // dense output: some lane's save time lies in (t, tn]
        f32x2 P2, P3, P4;
        if (SOLVER == LDE_SOLVER_TSIT5) {
          P2 = k[0] * ts5::R1[0];
          P3 = k[0] * ts5::R1[1];
          P4 = k[0] * ts5::R1[2];
#pragma unroll
          for (int s = 0; s < 6; s++) {
            P2 += k[s + 1] * ts5::R[s][0];
            P3 += k[s + 1] * ts5::R[s][1];
            P4 += k[s + 1] * ts5::R[s][2];
          }
        }
        const float rh = fast_rcp(h);
        for (bool more = tj <= tn; more; more = !ONE && tj <= tn) {   // (one trip when the group has a lane per save time)
          float2 out;
          if (tj >= tn) out = make_float2(yn.x, yn.y);
          else {
            const float th = (float)(tj - t) * rh;
            if (SOLVER == LDE_SOLVER_TSIT5) {
              out.x = tsit5_dense_eval<2>(th, h, y.x, k[0].x, P2.x, P3.x, P4.x);
              out.y = tsit5_dense_eval<2>(th, h, y.y, k[0].y, P2.y, P3.y, P4.y);
            } else {  // cubic Hermite between (y,k1) and (yn,f(yn))
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              out.x = h00 * y.x + (h10 * h) * k[0].x + h01 * yn.x + (h11 * h) * k[4].x;
              out.y = h00 * y.y + (h10 * h) * k[0].y + h01 * yn.y + (h11 * h) * k[4].y;
            }
          }
          if (RING > 0 && j < jc + RING) {
            s_ring[(j & (RING - 1)) * 64 + lane] = out;
            inring |= 1u << (j & (RING - 1));
          } else if (valid)
            *(ONE ? dst1 : z_out + (size_t)j * B + b) = out;   // (ONE: the lane's only store, address formed before the loop)
          j += LPT;
          tj = tjn;
          tjn = (!ONE && j + LPT < T) ? ts_at(j + LPT) : dinf;
        }
      }
      nacc++;
      iters++;
      y = yn;
      kf = k[FS];
      t = tn;
      dt = dtn;
      lqold = fmaxf(l, LQ_MIN);
      active = !last;
      if (last || iters >= maxit) pen = __builtin_inff();
    }
    if (RING > 0) {   // rows every trajectory of the wave has passed (a stopped one — done, failed, out of range — holds nothing back)
      const int jeff = pen == 0.f ? j : T;
      while (jc < T && !__any(jeff <= jc)) {
        const int r = jc & (RING - 1);
        if (valid && ((inring >> r) & 1u)) z_out[(size_t)jc * B + b] = s_ring[r * 64 + lane];
        inring &= ~(1u << r);
        jc++;
      }
    }
  }
  if (active && iters >= maxit) { ret = LDE_RET_MAXITERS; active = false; }
  if (!valid) return;
  if (ret != LDE_RET_SUCCESS) {  // failed solve ⇒ NaN block, never an error [REF GOKU.jl:114]
    const float qn = __int_as_float(0x7fc00000);
    for (int jj = slot; jj < T; jj += LPT) z_out[(size_t)jj * B + b] = make_float2(qn, qn);
  }
  if (slot == 0) {
    if (retcode) retcode[b] = ret;
    st_ret[b] = ret;
    st_nfe[b] = nfe + NS * (nacc + nrej);
    st_nacc[b] = nacc;
    st_nrej[b] = nrej;
    if (REC) o.rec.n[b] = ret == LDE_RET_SUCCESS ? nacc : 0;
  }
}

// ---- forward, B ≤ one workgroup per CU: a stepping wave and three dense-output waves per TRAJECTORY (round 3) -------------------
// In k_pend_forward_tl the wave that steps also evaluates the saves: ≈ 45 instructions + 21 constants per step of ≈ 165, on the one
// instruction chain whose length IS the launch. Here a workgroup is ONE trajectory on four waves, one per SIMD: wave 0 steps and does
// nothing else — per accepted step it leaves {h, yₙ₊₁} in LDS (one 16-byte write per lane into the lane's own copy, then the step
// count; LDS executes a wave's operations in order, so a reader that sees the count sees the record: no barrier, no fence) — and
// waves 1…3 walk the records: each rebuilds the slopes of every third step from (yₙ, hₙ) — the stepper's code on the stepper's
// inputs, so the same bits — and its lanes, which own the save times as in the tl kernel (lane s: j = 1+s, 65+s, …), interpolate
// and store what falls into that step. A helper has three step times for one step's slopes + dense output, so the stepper never
// waits; what follows the last step is one helper's last evaluation. Records live in a ring of SH_CAP steps (a solve that needs
// more goes through another round: one barrier pair); a failed solve's NaN block is written after the helpers' stores have landed.
constexpr int SH_CAP = 48;       // accepted steps recorded per round (48 KB of LDS: 16 B × 64 lane copies per step)
#ifndef LDE_SH_NH
#define LDE_SH_NH 3
#endif
constexpr int SH_NH = LDE_SH_NH;   // helper waves (measured at B = 256 with 3 / 5 / 7: 20.2 / 20.0 / 20.0–20.2 M trajectories/s — the tail behind the
                                  // stepper's last step is not the helpers' throughput)
#ifndef LDE_PEND_SH_SLEEP
#define LDE_PEND_SH_SLEEP 0
#endif

template <int KIND, int SOLVER, bool ADAPT, bool REC>   // REC: the instantiation that writes step records (LDE_SENSE_DISCRETE, "step_trace")
__global__ void __launch_bounds__(64 * (1 + SH_NH)) k_pend_forward_sh(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                         const double* __restrict__ ts_g, KOpts o,
                                                         float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                         int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                         int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  __shared__ __attribute__((aligned(16))) float s_rec[SH_CAP * 64 * 4];
  __shared__ __attribute__((aligned(16))) float s_klast[64 * 16];   // the last step's seven slopes (a copy per lane)
  __shared__ int s_cnt[64];
  __shared__ int s_fin;    // 0: stepping; 1: the round is over, another follows; 2: done
  __shared__ int s_fail;
  const int T = o.T, B = o.B, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = gridDim.x >> 3;   // XCD-aware trajectory ↔ workgroup map (as k_pend_forward_tl); the grid is a multiple of 8
  const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const bool valid = b < B;
  const int bc = valid ? b : B - 1;
  const double t_first = o.t_first, tend = o.t_last;
  const float2 zi = z0[bc];
  PendFwd<KIND> f(theta[bc]);
  if (tid < 64) s_cnt[tid] = 0;
  if (tid == 0) { s_fin = 0; s_fail = 0; }
  __syncthreads();
  constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // FSAL slope
  constexpr int NS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;  // RHS evaluations per attempt
  auto rec_at = [&](int n) -> float* { return s_rec + (size_t)(n * 64 + lane) * 4; };

  if (w == 0) {
    // ================= the stepper: k_pend_forward_tl's loop without the dense output =================
    PPROF(8);
    f32x2 y = {zi.x, zi.y}, k[7], yn = {0.f, 0.f}, kf = {0.f, 0.f};
    int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;
    double t = t_first;
    float dt = 0.f;
    constexpr float LQ_MIN = -13.287712379549449f;   // log₂ of qoldinit = 1e-4
    float lqold = LQ_MIN;
    const int maxit = o.maxiters > 0x7fffffffLL ? 0x7fffffff : (int)o.maxiters;
    const float dtmin = (float)o.dtmin;
#pragma unroll
    for (int s = 0; s < 7; s++) k[s] = f32x2{0.f, 0.f};
    if (valid && lane == 0) z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
    const double dtmax_d = tend - t;
    const float dtmax = (float)dtmax_d;
    f.anchor(y.x);
    kf = f.ev(y);
    nfe = 1;
    if (ADAPT) {
      if (o.dt_fixed > 0) dt = (float)fmin(o.dt_fixed, dtmax_d);
      else {
        const float ya[2] = {y.x, y.y}, fa[2] = {kf.x, kf.y};
        dt = (float)init_dt<2>(f, ya, fa, 1.0f, dtmax_d, o);
        nfe++;
      }
    } else
      dt = (float)o.dt_fixed;
    bool active = __any(t < tend) && maxit > 0;   // (votes: scalar from here on)
    if (__any(t < tend) && !active) ret = LDE_RET_MAXITERS;
    PPROF(9);
    for (;;) {   // rounds
      int n = 0;
      float* rp = rec_at(0);
      const double t_round = t;   // (REC) where this round's first record starts
      const f32x2 y_round = y;
      // `go` (scalar): this round goes on. (k_pend_forward_tl carries it as a float penalty added to the error norm, because its lanes stop
      // one by one; here the wave stops as one, and inside the loop the penalty would be the constant 0: msq + 0 is msq, bit for bit.)
      // (attempts so far = nacc + nrej; `lim` = accepted steps this round may still record: the ring's capacity or what maxiters leaves,
      //  one less per rejected attempt — so an accepted step costs one scalar add and one compare)
      int lim = min(SH_CAP, maxit - (nacc + nrej));
      bool go = active && lim > 0;
      while (go) {
        const float rem = (float)(tend - t);
        const bool last = __any(dt >= rem * 0.99999988f);   // (every lane of this wave carries the same solve: a vote is the lane's own answer, as a scalar branch)
        const float h = last ? rem : dt;
        k[0] = kf;
        f.anchor(y.x);
        float msq = 0.f;
        if (SOLVER == LDE_SOLVER_TSIT5) msq = tsit5_attempt_pair<PendFwd<KIND>, ADAPT>(f, h, y, k, yn, o);
        else {
          const float ya[2] = {y.x, y.y};
          float ka[7][2], yna[2];
          ka[0][0] = kf.x;
          ka[0][1] = kf.y;
          rk4_step<2>(f, h, ya, ka, yna);
#pragma unroll
          for (int s = 1; s <= 4; s++) k[s] = f32x2{ka[s][0], ka[s][1]};
          yn = f32x2{yna[0], yna[1]};
        }
        const float mq = fmaf(0.f, fabsf(yn.x) + fabsf(yn.y), msq);   // ∞·0 = NaN: a non-finite state never passes
        const bool ok = __any(mq <= 1.0f);
        float dtn = (float)o.dt_fixed, l = 0.f;
        if (ADAPT) {
          l = 0.5f * __builtin_amdgcn_logf(msq);
          const float q = fmaxf(o.q_lo, fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l - o.beta2 * lqold) * o.inv_gamma));
          dtn = fminf(h * fast_rcp(q), dtmax);
        }
        if (__builtin_expect(!ok, 0)) {                        // rare: a rejected or non-finite attempt (scalar branches throughout: the
                                                               // loop's variables are updated in place, no exec-masked copies at the loop edge)
          const bool fin = __any((fabsf(yn.x) + fabsf(yn.y)) < __builtin_inff());
          nrej++;
          lim--;
          if (!ADAPT) { ret = LDE_RET_NONFINITE; active = false; nrej--; }
          else if (!fin) {
            if (__any(h > dtmin)) dt = h * o.qmin;
            else { ret = LDE_RET_NONFINITE; active = false; nrej--; }
          } else {
            dt = h * fast_rcp(fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l) * o.inv_gamma));
            if (__any(dt < dtmin)) { ret = LDE_RET_DTMIN; active = false; }
          }
          if (!active || n >= lim) go = false;
          continue;
        }
        {   // the accepted step: leave {h, yₙ₊₁} behind, publish, advance
          if (last) {   // once per solve: the last step's slopes travel too — its helper would otherwise start rebuilding them only now, a
                        // whole step's worth of instructions behind the end of the solve (measured: 1.3 µs of the launch)
            float* kl = s_klast + lane * 4;   // [quarter][lane][4]: every 16-byte write conflict-free
            *reinterpret_cast<f32x4*>(kl) = f32x4{k[0].x, k[0].y, k[1].x, k[1].y};
            *reinterpret_cast<f32x4*>(kl + 256) = f32x4{k[2].x, k[2].y, k[3].x, k[3].y};
            *reinterpret_cast<f32x4*>(kl + 512) = f32x4{k[4].x, k[4].y, k[5].x, k[5].y};
            *reinterpret_cast<f32x4*>(kl + 768) = f32x4{k[6].x, k[6].y, 0.f, 0.f};
          }
          *reinterpret_cast<f32x2*>(rp) = yn;   // record = {yₙ₊₁ (the pair as it sits in its registers: one 8-byte write), h}
          rp[2] = h;
          rp += 64 * 4;
          n++;
          asm volatile("" ::: "memory");                                      // the count is published AFTER the record
          __hip_atomic_store(&s_cnt[lane], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a plain LDS store: ds_write_b32)
          y = yn;
          kf = k[FS];
          t += (double)h;   // (the last step ends the loop: its t is not read again)
          dt = dtn;
          lqold = max_f(l, LQ_MIN);
          active = !last;
          if (last || n >= lim) go = false;
        }
      }
      nacc += n;
      PPROF(10);
      PPROF_VAL(30, nacc + nrej);
      if (active && nacc + nrej >= maxit) { ret = LDE_RET_MAXITERS; active = false; }
      if (ret != LDE_RET_SUCCESS && lane == 0) __hip_atomic_store(&s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_fin, active ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (REC) {
        // The step records (LDE_SENSE_DISCRETE / "step_trace") of this round, written by the stepping wave while the helpers finish their
        // dense output (the stepper is done ≈ 2 µs before them): lane i = the round's step i, read back from the ring — start state = the
        // step before's end, start time = the stepper's own running sum (replayed in its order: the same doubles).
        const int i0 = nacc - n;
        const bool mine = lane < n;
        const f32x4 q = *reinterpret_cast<const f32x4*>(s_rec + (size_t)((mine ? lane : 0) * 64 + lane) * 4);
        const f32x4 qp = *reinterpret_cast<const f32x4*>(s_rec + (size_t)((mine && lane > 0 ? lane - 1 : 0) * 64 + lane) * 4);
        const f32x2 ys = lane == 0 ? y_round : f32x2{qp[0], qp[1]};
        double tacc = t_round, ti = t_round;
        const int hbits = __float_as_int(q[2]);
        for (int i = 0; i < n; i++) {
          if (lane == i) ti = tacc;
          tacc += (double)__int_as_float(__builtin_amdgcn_readlane(hbits, i));
        }
        if (valid && mine && i0 + lane < o.rec.cap) {
          o.rec.t[(size_t)(i0 + lane) * B + b] = ti;
          o.rec.dt[(size_t)(i0 + lane) * B + b] = (double)q[2];
          reinterpret_cast<float2*>(o.rec.y)[(size_t)(i0 + lane) * B + b] = make_float2(ys.x, ys.y);
        }
      }
      if (!active) break;
      __syncthreads();   // A: the helpers have consumed this round's records
      s_cnt[lane] = 0;
      if (lane == 0) s_fin = 0;
      __syncthreads();   // B: counts reset
    }
    if (ret != LDE_RET_SUCCESS) {  // failed solve ⇒ NaN block, never an error [REF GOKU.jl:114] — after every helper store has landed
      __syncthreads();   // F
      if (valid) {
        const float qn = __int_as_float(0x7fc00000);
        for (int j = lane; j < T; j += 64) z_out[(size_t)j * B + b] = make_float2(qn, qn);
      }
    }
    if (valid && lane == 0) {
      if (retcode) retcode[b] = ret;
      st_ret[b] = ret;
      st_nfe[b] = nfe + NS * (nacc + nrej);
      st_nacc[b] = nacc;
      st_nrej[b] = nrej;
      if (REC) o.rec.n[b] = ret == LDE_RET_SUCCESS ? nacc : 0;
    }
    PPROF(11);
    return;
  }

  // ================= the helpers: wave 1 + hid rebuilds every SH_NH-th step; lanes = save times =================
#if LDE_PEND_ABL == 5   // diagnostic: the stepper alone (no dense output at all — results are wrong)
  return;
#endif
  const int hid = w - 1;
  const double dinf = __longlong_as_double(0x7ff0000000000000LL);
  int jq = 1 + lane;                                   // the save time this lane looks for next
  double tj = jq < T ? ts_g[jq] : dinf;
  double tn = t_first;                                 // the walk over the records: record n2 starts at time tn in state ys
  f32x2 ys = {zi.x, zi.y};
  int nrec = 0;                                        // records walked so far over all rounds (whose turn a step is)
  for (;;) {   // rounds
    int n2 = 0, fin;
    for (;;) {
      fin = __hip_atomic_load(&s_fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // read BEFORE the count: if the round is over, the count is final
      const int cnt = __hip_atomic_load(&s_cnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      while (n2 < cnt) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(rec_at(n2));
        const float h = q[2];
        const f32x2 ye = {q[0], q[1]};
        const double t1 = (h == (float)(tend - tn)) ? tend : tn + (double)h;   // exactly the stepper's arithmetic
        const bool mine = (nrec % SH_NH) == hid;
        if (__any(tj <= t1)) {
          f32x2 k0 = {0.f, 0.f}, kE = {0.f, 0.f}, P2 = {0.f, 0.f}, P3 = {0.f, 0.f}, P4 = {0.f, 0.f};
          float rh = 0.f;
          if (mine && __any(tj < t1)) {   // rebuild the step's slopes: the stepper's code on the stepper's inputs (yₙ, hₙ)
            rh = fast_rcp(h);
            if (SOLVER == LDE_SOLVER_TSIT5) {
              f32x2 kk[7], ynr;
              if (t1 == tend) {   // the step that reaches the end: its slopes came with the record
                const float* kl = s_klast + lane * 4;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(kl), a1 = *reinterpret_cast<const f32x4*>(kl + 256);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(kl + 512), a3 = *reinterpret_cast<const f32x4*>(kl + 768);
                kk[0] = f32x2{a0[0], a0[1]}; kk[1] = f32x2{a0[2], a0[3]}; kk[2] = f32x2{a1[0], a1[1]}; kk[3] = f32x2{a1[2], a1[3]};
                kk[4] = f32x2{a2[0], a2[1]}; kk[5] = f32x2{a2[2], a2[3]}; kk[6] = f32x2{a3[0], a3[1]};
              } else {
                f.anchor(ys.x);
                kk[0] = f.ev(ys);
                (void)tsit5_attempt_pair<PendFwd<KIND>, false>(f, h, ys, kk, ynr, o);
              }
              k0 = kk[0];
              P2 = kk[0] * ts5::R1[0];
              P3 = kk[0] * ts5::R1[1];
              P4 = kk[0] * ts5::R1[2];
#pragma unroll
              for (int s = 0; s < 6; s++) {
                P2 += kk[s + 1] * ts5::R[s][0];
                P3 += kk[s + 1] * ts5::R[s][1];
                P4 += kk[s + 1] * ts5::R[s][2];
              }
            } else {
              f.anchor(ys.x);
              k0 = f.ev(ys);
              kE = f.ev(ye);   // f(yₙ₊₁), the slope at the end of the step
            }
          }
          while (tj <= t1) {   // this lane's save times inside the step (whoever serves them, the lane moves past them)
            if (mine) {
              float2 out;
              if (tj >= t1) out = make_float2(ye.x, ye.y);   // the save time is the step's end
              else {
                const float th = (float)(tj - tn) * rh;
                if (SOLVER == LDE_SOLVER_TSIT5) {
                  out.x = tsit5_dense_eval<2>(th, h, ys.x, k0.x, P2.x, P3.x, P4.x);
                  out.y = tsit5_dense_eval<2>(th, h, ys.y, k0.y, P2.y, P3.y, P4.y);
                } else {  // cubic Hermite between (y,k1) and (yn,f(yn))
                  const float om = 1.0f - th;
                  const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
                  const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
                  out.x = h00 * ys.x + (h10 * h) * k0.x + h01 * ye.x + (h11 * h) * kE.x;
                  out.y = h00 * ys.y + (h10 * h) * k0.y + h01 * ye.y + (h11 * h) * kE.y;
                }
              }
              if (valid) z_out[(size_t)jq * B + b] = out;
            }
            jq += 64;
            tj = jq < T ? ts_g[jq] : dinf;
          }
        }
        tn = t1;
        ys = ye;
        n2++;
        nrec++;
      }
      if (fin) break;   // (read before the count: everything of this round has been walked)
#if LDE_PEND_SH_SLEEP
      __builtin_amdgcn_s_sleep(1);   // (≈ 0.14 µs on gfx950; the helpers have a SIMD each, so they may as well watch the count)
#endif
    }
    if (fin == 2) break;
    __syncthreads();   // A
    __syncthreads();   // B
  }
#if LDE_PEND_PROF
  if (blockIdx.x == 0 && lane == 0 && hid < 3) { g_pprof[2 * (12 + hid)] = wall_clock64(); g_pprof[2 * (12 + hid) + 1] = __builtin_readcyclecounter(); }   // helper hid has stored its last save
#endif
  if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // F: every helper store has been issued and waited for
  }
}

#include "lde_pend_lp.h"

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
    f.anchor(y[0]);
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
      f.anchor(y[0]);
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
          f.anchor(y[0]);
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
  float noff;
  __device__ __forceinline__ explicit PendBasis(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)), noff(0.f) {}
  __device__ __forceinline__ void anchor(float x0) { noff = turn_anchor(x0); }
  // y = [z0 z1 | la0 la1 lb0 lb1 | ga gb]
  __device__ __forceinline__ void operator()(const float (&y)[8], float (&dy)[8]) const {
    float s, c;
    hw_sincos(y[0], s, c, noff);
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

// The same system on register pairs: p0 = (z0, z1), p1 = (λᵃ₀, λᵇ₀), p2 = (λᵃ₁, λᵇ₁), p3 = (gᵃ, gᵇ) — the two basis solutions share
// every coefficient, so their halves of the right-hand side are three packed operations.
template <int KIND>
struct PendBasisPair {
  float ngl, gl2, noff;
  __device__ __forceinline__ explicit PendBasisPair(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)), noff(0.f) {}
  __device__ __forceinline__ void anchor(float x0) { noff = turn_anchor(x0); }
  __device__ __forceinline__ void ev(const f32x2 (&y)[4], f32x2 (&dy)[4]) const {
    float s, c;
    hw_sincos(y[0].x, s, c, noff);
    float acc = ngl * s;
    if (KIND == 1) acc -= 0.7f * y[0].y;
    dy[0] = f32x2{y[0].y, acc};
    const float nc = ngl * c, gs = gl2 * s;
    dy[1] = y[2] * (-nc);
    f32x2 v = y[1];
    if (KIND == 1) v -= y[2] * 0.7f;
    dy[2] = -v;
    dy[3] = y[2] * (-gs);
  }
};

// one Tsit5 attempt on four register pairs (all eight entries count); k[0] = f(y) on entry; returns the RMS error estimate
template <class F>
__device__ __forceinline__ float tsit5_attempt_pair4(F& f, float h, const f32x2 (&y)[4], f32x2 (&k)[7][4], f32x2 (&yn)[4], const KOpts& o) {
  f32x2 tmp[4];
#pragma unroll
  for (int s = 1; s < 6; s++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f32x2 acc = k[0][i] * ts5::A[s][0];
#pragma unroll
      for (int j = 1; j < s; j++) acc += k[j][i] * ts5::A[s][j];
      tmp[i] = y[i] + acc * h;
    }
    f.ev(tmp, k[s]);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f32x2 acc = k[0][i] * ts5::A[6][0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc += k[j][i] * ts5::A[6][j];
    yn[i] = y[i] + acc * h;
  }
  f.ev(yn, k[6]);
  if (!o.adaptive) return 0.f;
  f32x2 s2 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f32x2 e = k[0][i] * ts5::BT[0];
#pragma unroll
    for (int j = 1; j < 7; j++) e += k[j][i] * ts5::BT[j];
    e *= h;
    const f32x2 sk = f32x2{fmaxf(fabsf(y[i].x), fabsf(yn[i].x)), fmaxf(fabsf(y[i].y), fabsf(yn[i].y))} * o.reltol + o.abstol;
    const f32x2 r = e * f32x2{fast_rcp(sk.x), fast_rcp(sk.y)};
    s2 += r * r;
  }
  return sqrtf((s2.x + s2.y) * 0.125f);
}

// integrate one save interval [t0, t1] backwards for the basis state; returns retcode, leaves the operator in y[2..7]
// Tsit5: on register pairs (PendBasisPair / tsit5_attempt_pair4) — same control flow as the generic form below, half the
// vector instructions of the stage sums and of the right-hand side.
template <int KIND>
__device__ __forceinline__ int pend_interval_operator_pairs(float2 zc, float L, double t0, double t1, const KOpts& o, float (&yo)[8],
                                                            int& nacc, int& nrej) {
  PendBasisPair<KIND> f(L);
  f32x2 y[4] = {f32x2{zc.x, zc.y}, f32x2{1.f, 0.f}, f32x2{0.f, 1.f}, f32x2{0.f, 0.f}};
  f32x2 k[7][4], yn[4];
  auto out = [&]() {   // back to [z0 z1 | la0 la1 lb0 lb1 | ga gb]
    yo[0] = y[0].x; yo[1] = y[0].y; yo[2] = y[1].x; yo[3] = y[2].x; yo[4] = y[1].y; yo[5] = y[2].y; yo[6] = y[3].x; yo[7] = y[3].y;
  };
  int ret = LDE_RET_SUCCESS;
  nacc = 0;
  nrej = 0;
  out();
  if (!(isfinite(zc.x) && isfinite(zc.y))) return LDE_RET_NONFINITE;
  const double len = t1 - t0;
  double t = t1, dt = o.adaptive ? len : o.dt_fixed;   // first attempt: the whole interval
  const double dtmax = len;
  float qold = 1e-4f;
  long long iters = 0;
  f.anchor(y[0].x);
  f.ev(y, k[0]);
  for (;;) {
    if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
    f.anchor(y[0].x);
    const double dist = t - t0;
    double hmag = dt;
    bool hit = false;
    if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = true; }
    const float h = -(float)hmag;
    const float EEst = tsit5_attempt_pair4(f, h, y, k, yn, o);
    float fin = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) fin += fabsf(yn[i].x) + fabsf(yn[i].y);   // one non-finite entry makes the sum non-finite
    if (!(fin < __builtin_inff()) || !(EEst == EEst)) {
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
    for (int i = 0; i < 4; i++) y[i] = yn[i];
    if (hit) break;
    t -= hmag;
#pragma unroll
    for (int i = 0; i < 4; i++) k[0][i] = k[6][i];
    dt = o.adaptive ? dtp : o.dt_fixed;
  }
  out();
  return ret;
}

template <int KIND, int SOLVER>
__device__ __forceinline__ int pend_interval_operator(float2 zc, float L, double t0, double t1, const KOpts& o, float (&y)[8],
                                                      int& nacc, int& nrej) {
  if (SOLVER == LDE_SOLVER_TSIT5) return pend_interval_operator_pairs<KIND>(zc, L, t0, t1, o, y, nacc, nrej);
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
  f.anchor(y[0]);
  f(y, k[0]);
  for (;;) {
    if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
    f.anchor(y[0]);
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
  // on register pairs: the rows of a's matrix and (n₀, n₁) are pairs, b's entries scalar multipliers — 13 instructions instead of 27
  const f32x2 ar0 = {a.m00, a.m01}, ar1 = {a.m10, a.m11}, an = {a.n0, a.n1};
  const f32x2 r0 = ar0 * b.m00 + ar1 * b.m01;
  const f32x2 r1 = ar0 * b.m10 + ar1 * b.m11;
  const f32x2 rn = an + ar0 * b.n0 + ar1 * b.n1;
  AffOp r;
  r.m00 = r0.x; r.m01 = r0.y; r.m10 = r1.x; r.m11 = r1.y;
  r.d0 = b.m00 * a.d0 + b.m01 * a.d1 + b.d0;
  r.d1 = b.m10 * a.d0 + b.m11 * a.d1 + b.d1;
  r.n0 = rn.x; r.n1 = rn.y;
  r.gam = a.gam + b.gam + b.n0 * a.d0 + b.n1 * a.d1;
  return r;
}
// The value of lane ℓ+OFF for the lanes the tree uses at that level (ℓ a multiple of 2·OFF), without an LDS round trip per
// float (`__shfl_down` = address arithmetic + ds_bpermute_b32: twelve of them per level were a third of the kernel):
//   OFF ≤ 8: the partner sits in the same row of 16 lanes — DPP row_shl:OFF, one VALU move;
//   OFF = 16: lanes 0 and 32 read lanes 16 and 48 — ds_swizzle (xor 16), no address register;
//   OFF = 32: only lane 0 reads lane 32 — v_readlane, the value arrives in a scalar register.
// Other lanes receive unspecified values; they do not use them.
template <int OFF>
__device__ __forceinline__ int lane_down(int v) {
  if (OFF < 16) return __builtin_amdgcn_update_dpp(v, v, 0x100 + OFF, 0xf, 0xf, false);   // row_shl:OFF
  if (OFF == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401f);                            // and 0x1f, or 0, xor 0x10
  return __builtin_amdgcn_readlane(v, 32);
}
template <int OFF>
__device__ __forceinline__ float lane_down(float v) { return __builtin_bit_cast(float, lane_down<OFF>(__builtin_bit_cast(int, v))); }
template <int OFF>
__device__ __forceinline__ AffOp aff_down(const AffOp& a) {
  AffOp r;
  r.m00 = lane_down<OFF>(a.m00); r.m01 = lane_down<OFF>(a.m01); r.m10 = lane_down<OFF>(a.m10);
  r.m11 = lane_down<OFF>(a.m11); r.d0 = lane_down<OFF>(a.d0); r.d1 = lane_down<OFF>(a.d1);
  r.n0 = lane_down<OFF>(a.n0); r.n1 = lane_down<OFF>(a.n1); r.gam = lane_down<OFF>(a.gam);
  return r;
}
// one level of the order-preserving tree: lane ℓ ← (ℓ's block first, then the block of ℓ+OFF); `cnt` = nacc | nrej << 16
template <int OFF>
__device__ __forceinline__ void aff_tree_level(AffOp& op, int& cnt, int& ret, int lane) {
  const AffOp hi = aff_down<OFF>(op);
  const int c2 = lane_down<OFF>(cnt), e2 = lane_down<OFF>(ret);
  // Unconditional on every lane: a lane that is not a multiple of 2·OFF computes something nobody reads — the lanes the next level reads
  // (multiples of 2·OFF and their partners at +2·OFF, themselves multiples of 2·OFF) were all updated from valid partners — and the
  // eleven per-value selects of the predicated form were a quarter of the level's instructions.
  (void)lane;
  op = aff_compose(op, hi);
  cnt += c2;
  ret = ret ? ret : e2;
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
  // Every global load of the kernel is issued HERE, back to back — the interval's ẑ, its cotangent, its two save times, and the
  // trajectory's last save point, which lane 0 needs after the reduction. As the source reads naturally (ẑ → finite? → times →
  // integrate → Δẑ → reduce → ẑ_T, Δẑ_T) the compiler kept four dependent memory round trips on a 5 µs kernel; the empty asm pins
  // the values to this point, so there is one.
  const int jc = j >= 0 ? j : 0;
  float2 zc = z_out[(size_t)(jc + 1) * B + b];
  float2 dj = dz_out[(size_t)jc * B + b];
  double t0 = ts_g[jc], t1 = ts_g[jc + 1];
  float2 zT = z_out[(size_t)(T - 1) * B + b];
  float2 dT = dz_out[(size_t)(T - 1) * B + b];
  const float Lb = theta[b];
  asm volatile("" : "+v"(zc.x), "+v"(zc.y), "+v"(dj.x), "+v"(dj.y), "+v"(t0), "+v"(t1), "+v"(zT.x), "+v"(zT.y), "+v"(dT.x), "+v"(dT.y));
  if (j >= 0) {
    float y[8];
#if LDE_PEND_ABL == 4   // diagnostic: what the kernel costs WITHOUT the interval integration (loads, tree, launch) — DESIGN §4.2b
    y[2] = 1.f + zc.x * 1e-9f; y[3] = 0.f; y[4] = 0.f; y[5] = 1.f; y[6] = (float)(t1 - t0) * Lb * 1e-9f; y[7] = 0.f;
    nacc = 1;
#else
    ret = pend_interval_operator<KIND, SOLVER>(zc, Lb, t0, t1, o, y, nacc, nrej);
#endif
    op = AffOp{y[2], y[4], y[3], y[5], dj.x, dj.y, y[6], y[7], 0.f};   // λ' = λ₀·(la) + λ₁·(lb) + Δ_j ; g' = g + ga λ₀ + gb λ₁
  }
  // order-preserving tree reduction inside the wave (a save interval takes a handful of attempts: the two counts share a word)
  // (per-lane counts saturate at 1023 / 511 so that 64 of them cannot carry into the neighbouring field)
  int cnt = min(nacc, 1023) | (min(nrej, 511) << 16);
  aff_tree_level<1>(op, cnt, ret, lane);
  aff_tree_level<2>(op, cnt, ret, lane);
  aff_tree_level<4>(op, cnt, ret, lane);
  aff_tree_level<8>(op, cnt, ret, lane);
  aff_tree_level<16>(op, cnt, ret, lane);
  aff_tree_level<32>(op, cnt, ret, lane);
  nacc = cnt & 0xffff;
  nrej = (cnt >> 16) & 0x7fff;
  if (nwave > 1) {   // T − 1 > 64: the waves' operators meet in LDS (a single wave needs neither the copy nor the barrier)
    if (lane == 0) {
      s_op[wave] = op;
      s_stat[wave][0] = nacc;
      s_stat[wave][1] = nrej;
      s_stat[wave][2] = ret;
    }
    __syncthreads();
  }
  if (l == 0) {
    for (int w = 1; w < nwave; w++) {
      op = aff_compose(op, s_op[w]);
      nacc += s_stat[w][0];
      nrej += s_stat[w][1];
      ret = ret ? ret : s_stat[w][2];
    }
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
                        hipStream_t stream, const PendTune& tn) {
  const int block = pick_block(o.B), grid = (o.B + block - 1) / block;
  const size_t shm = o.T <= TS_LDS_MAX ? (size_t)o.T * sizeof(double) : 0;
  // which mapping serves this solve: csrc/lde_host.h (pend_forward_mapping — pure host logic, tested on the CPU: tests/test_sanitizers.py)
  const bool recording = o.rec.n != nullptr;   // k_pend_forward_lp / _sh, k_pend_forward_ws, k_pend_forward and the 16-row ring form write step records
  const lde_host::PendFwdMap map = lde_host::pend_forward_mapping(kind, solver, o.adaptive != 0, recording, o.B, o.T, tn, TS_LDS_MAX);
  const bool lp_shape = map == lde_host::PEND_FWD_LP4 || map == lde_host::PEND_FWD_LP3;
  if (lp_shape || map == lde_host::PEND_FWD_SH) {
    const bool ad = o.adaptive != 0;
    const int g8 = ((o.B + 7) / 8) * 8;
#define LDE_LAUNCH_SH(K, S, A)                                                                                          \
  do {                                                                                                                  \
    if (o.rec.n)                                                                                                        \
      hipLaunchKernelGGL((k_pend_forward_sh<K, S, A, true>), dim3(g8), dim3(64 * (1 + SH_NH)), 0, stream, (const float2*)z0, theta, ts_dev, o,  \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
    else                                                                                                                \
      hipLaunchKernelGGL((k_pend_forward_sh<K, S, A, false>), dim3(g8), dim3(64 * (1 + SH_NH)), 0, stream, (const float2*)z0, theta, ts_dev, o, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
  } while (0)
    g_pend_last[0] = lp_shape ? "k_pend_forward_lp" : "k_pend_forward_sh";
    if (lp_shape) {
      // the metric's shape: the stepping wave's lanes in pairs, the step as a Nyström scheme (lde_pend_lp.h; option "pend_lp" = 0: k_pend_forward_sh);
      // four dense-output waves while at most two workgroups share a CU, three beyond
#define LDE_LAUNCH_LP(R, NH)                                                                                                              \
  hipLaunchKernelGGL((k_pend_forward_lp<R, NH>), dim3(g8), dim3(64 * (1 + NH)), 0, stream, (const float2*)z0, theta, ts_dev, o, (float2*)z_out, \
                     retcode, nfe, nacc, nrej, ret)
      if (map == lde_host::PEND_FWD_LP4) {
        if (o.rec.n) LDE_LAUNCH_LP(true, 4);
        else LDE_LAUNCH_LP(false, 4);
      } else {
        if (o.rec.n) LDE_LAUNCH_LP(true, 3);
        else LDE_LAUNCH_LP(false, 3);
      }
#undef LDE_LAUNCH_LP
    } else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_SH(0, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_SH(0, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH_SH(0, LDE_SOLVER_RK4, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_SH(1, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_SH(1, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH_SH(1, LDE_SOLVER_RK4, false);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH_SH
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  if (map == lde_host::PEND_FWD_TL) {
    g_pend_last[0] = "k_pend_forward_tl";
    const bool ad = o.adaptive != 0;
    const bool few = o.T - 1 <= 64;   // a lane serves exactly one save time: the variant without a load in the stepping loop
#define LDE_LAUNCH_TL(K, S, A)                                                                                          \
  do {                                                                                                                  \
    if (few)                                                                                                            \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 1, true>), dim3(((o.B + 7) / 8) * 8), dim3(64), 0, stream, (const float2*)z0, theta, ts_dev, o, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
    else                                                                                                                \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 1>), dim3(((o.B + 7) / 8) * 8), dim3(64), 0, stream, (const float2*)z0, theta, ts_dev, o, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
  } while (0)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_TL(0, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_TL(0, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH_TL(0, LDE_SOLVER_RK4, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_TL(1, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_TL(1, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH_TL(1, LDE_SOLVER_RK4, false);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH_TL
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  if (map == lde_host::PEND_FWD_WS) {   // (writes step records too)
    g_pend_last[0] = "k_pend_forward_ws";
    const size_t lds = (size_t)((o.T + 1) & ~1) * sizeof(double) + (size_t)WS_CAP * 64 * WS_RW * sizeof(float);
    const int g64 = (o.B + 63) / 64;
#define LDE_LAUNCH_WS(K, S, A)                                                                                         \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)k_pend_forward_ws<K, S, A, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024 - 2048) != hipSuccess ||                                                       \
          hipFuncSetAttribute((const void*)k_pend_forward_ws<K, S, A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024 - 2048) != hipSuccess)                                                         \
        return LDE_ERR_HIP;                                                                                            \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    if (recording)                                                                                                     \
      hipLaunchKernelGGL((k_pend_forward_ws<K, S, A, true>), dim3(g64), dim3(WS_THREADS), lds, stream, (const float2*)z0, theta, ts_dev, \
                         o, (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                            \
    else                                                                                                               \
      hipLaunchKernelGGL((k_pend_forward_ws<K, S, A, false>), dim3(g64), dim3(WS_THREADS), lds, stream, (const float2*)z0, theta, ts_dev, \
                         o, (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                            \
  } while (0)
    const bool ad = o.adaptive != 0;
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_WS(0, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_WS(0, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH_WS(0, LDE_SOLVER_RK4, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_WS(1, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_WS(1, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH_WS(1, LDE_SOLVER_RK4, false);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH_WS
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  // large batches (B ≥ 2¹⁷ — option "pend_lb_min_b"; T ≤ 2048): the lanes-as-save-times kernel with 64 trajectories per wave, the row ring
  // and the save grid in LDS (k_pend_forward_tl<…, 64, false, RING>). k_pend_forward's per-lane stores reach HBM as partial lines —
  // 1128 MB written for 419 MB of ẑ at B = 2²⁰ — and the launch is bound by that traffic: 303 µs. With 16 ring rows and an 8-row hold
  // 444 … 574 MB are written and the launch takes 206 µs (2.1 TB/s of ẑ; abl/pend_LB.py, abl/pend_LB_pmc.sh); 8 rows / 3 the same within the
  // run-to-run spread, 32 rows leave 2.5 waves per SIMD (16 KB of LDS per wave) and take 258 µs. What remains is instruction issue:
  // ≈ 8.2 k instructions per wave (5.1 k VALU), of which the wave-sequential dense-output loop is ≈ 75 × 45 and the row flush 49 × 28.
  // options "pend_lb" = rows of the ring (8 / 16 / 32; 0: off), "pend_lb_hold" = the hold margin.
  const int lb_ring = tn.lb_ring;   // rows of the ring; 0: off
  if (map == lde_host::PEND_FWD_RING) {   // (the save grid in LDS beside the ring: T ≤ 2048; a recording forward: the 16-row ring)
    g_pend_last[0] = "k_pend_forward_tl";
    // a lane sits out while j ≥ jc + RING − hold; the slowest lane has j = jc, so hold ≤ RING − 1 keeps it (and with it jc) moving —
    // hold ≥ RING would hold EVERY lane on every iteration and the solve loop would never end. Default: half the ring.
    const int lb_hold_env = tn.lb_hold;
    const int ring_rows = recording ? 16 : (lb_ring >= 32 ? 32 : (lb_ring >= 16 ? 16 : 8));
    KOpts oh = o;
    oh.lb_hold = std::max(0, std::min(lb_hold_env >= 0 ? lb_hold_env : ring_rows / 2, ring_rows - 1));
    const bool ad = o.adaptive != 0;
    const int g8 = (((o.B + 63) / 64 + 7) / 8) * 8;
#define LDE_LAUNCH_LB(K, S, A)                                                                                          \
  do {                                                                                                                  \
    if (recording)                                                                                                      \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 64, false, 16, true>), dim3(g8), dim3(64), (size_t)o.T * sizeof(double), stream, (const float2*)z0, theta, ts_dev, oh, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
    else if (lb_ring >= 32)                                                                                             \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 64, false, 32>), dim3(g8), dim3(64), (size_t)o.T * sizeof(double), stream, (const float2*)z0, theta, ts_dev, oh, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
    else if (lb_ring >= 16)                                                                                             \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 64, false, 16>), dim3(g8), dim3(64), (size_t)o.T * sizeof(double), stream, (const float2*)z0, theta, ts_dev, oh, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
    else                                                                                                                \
      hipLaunchKernelGGL((k_pend_forward_tl<K, S, A, 64, false, 8>), dim3(g8), dim3(64), (size_t)o.T * sizeof(double), stream, (const float2*)z0, theta, ts_dev, oh, \
                         (float2*)z_out, retcode, nfe, nacc, nrej, ret);                                                \
  } while (0)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_LB(0, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_LB(0, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH_LB(0, LDE_SOLVER_RK4, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5 && ad) LDE_LAUNCH_LB(1, LDE_SOLVER_TSIT5, true);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH_LB(1, LDE_SOLVER_TSIT5, false);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH_LB(1, LDE_SOLVER_RK4, false);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH_LB
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  g_pend_last[0] = "k_pend_forward<";
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
  g_pend_last[1] = "k_pend_adjoint<";
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

// ---- discrete (exact) sensitivity: LDE_SENSE_DISCRETE ---------------------------------------------------------------------------
// What the reference's GOKU default ForwardDiffSensitivity() [REF examples/pendulum_friction-less/pendulum.jl:11] differentiates
// [REF src/models/GOKU.jl:107, :121]: the discrete solve on its accepted step sequence, step sizes constant. A lane owns a trajectory and
// sweeps ITS recorded steps (t_n, dt_n, y_n) from the last to the first: the six (four) stage points and their sin / cos are rebuilt from
// y_n — the forward kernel's arithmetic on the forward kernel's inputs — and the cotangent is pulled through the dense output of every save
// time inside the step, the FSAL slope, the solution weights and the stage sums.
// No controller, no error norm, no forced stops at the save times: (2S + 1) sine/cosine pairs per accepted step. Traffic = the record
// (24 B per step) + Δẑ; every access coalesced over the batch index.
#ifndef LDE_DISC_PF
#define LDE_DISC_PF 4   // B = 2²⁰: 0.43 → 0.37–0.39 ms (2: 0.40; 6, 8 and workgroups of 64 / 128 lanes: the same within the run-to-run spread; abl/lb_ab.sh)
#endif
template <int KIND, int SOLVER>
__global__ void __launch_bounds__(256) k_pend_adjoint_disc(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                           const double* __restrict__ ts_g, KOpts o,
                                                           const float2* __restrict__ dz_out, float2* __restrict__ dz0,
                                                           float* __restrict__ dtheta, int32_t* __restrict__ st_nfe,
                                                           int32_t* __restrict__ st_nacc, int32_t* __restrict__ st_nrej,
                                                           int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];
  const int T = o.T, B = o.B;
  const bool ts_lds = T <= TS_LDS_MAX;
  if (ts_lds)
    for (int i = threadIdx.x; i < T; i += blockDim.x) s_lds[i] = ts_g[i];
  __syncthreads();
  auto s_ts = [&](int i) -> double { return ts_lds ? s_lds[i] : ts_g[i]; };
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  constexpr int S = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  // a_ij of the stages and, in row S, the solution weights
  constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                              {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
  auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
  const StepRec& R = o.rec;
  const float L = theta[b];
  const float ngl = -10.0f / L, gl2 = 10.0f / (L * L);
  const float2 y0 = z_out[b];
  const int ns = R.n[b];
  int ret = LDE_RET_SUCCESS;
  if (!isfinite(y0.x) || !isfinite(y0.y)) ret = LDE_RET_NONFINITE;      // a failed forward trajectory: zero gradient [REF GOKU.jl:114]
  else if (T > 1 && (ns < 1 || ns > R.cap)) ret = LDE_RET_MAXITERS;      // no usable record: NaN gradient (never a truncated sweep)
  float yb[2] = {0.f, 0.f}, carry[2] = {0.f, 0.f}, gth = 0.f;
  float sn[S + 1], cs[S + 1];
  sn[0] = cs[0] = 0.f;
  if (ret == LDE_RET_SUCCESS && T > 1) {
    const double tend = s_ts(T - 1);
    int j = T - 1;
    double tnext = tend;
#if LDE_DISC_PF
    // Every load of the sweep is requested a step (the record) resp. LDE_DISC_PF save times (Δẑ) before its use: the walk is one dependent
    // chain per lane, and a load at the head of each link leaves only the other waves of the SIMD to cover its latency.
    float2 dq[LDE_DISC_PF];
#pragma unroll
    for (int u = 0; u < LDE_DISC_PF; u++) dq[u] = dz_out[(size_t)(j - u > 0 ? j - u : 0) * B + b];
    double t_pf = R.t[(size_t)(ns - 1) * B + b], dt_pf = R.dt[(size_t)(ns - 1) * B + b];
    float2 y_pf = reinterpret_cast<const float2*>(R.y)[(size_t)(ns - 1) * B + b];
#endif
    for (int s = ns - 1; s >= 0; s--) {
#if LDE_DISC_PF
      const double t = t_pf, dt = dt_pf;
      const float2 yv = y_pf;
      {
        const int sp = s > 0 ? s - 1 : 0;
        t_pf = R.t[(size_t)sp * B + b];
        dt_pf = R.dt[(size_t)sp * B + b];
        y_pf = reinterpret_cast<const float2*>(R.y)[(size_t)sp * B + b];
      }
#else
      const double t = R.t[(size_t)s * B + b], dt = R.dt[(size_t)s * B + b];
      const float2 yv = reinterpret_cast<const float2*>(R.y)[(size_t)s * B + b];
#endif
      const float h = (float)dt;
      const bool last = s == ns - 1;
      const double tnew = tnext;
      tnext = t;
      // ---- the stage points again: g_0 = y_n, g_i = y_n + h Σ a_iq k_q, g_S = y_{n+1}
      const float noff = turn_anchor(yv.x);
      float gx[S + 1], gy[S + 1], kx[S], ky[S];
      gx[0] = yv.x;
      gy[0] = yv.y;
#pragma unroll
      for (int i = 0; i <= S; i++) {
        if (i > 0) {
          if (SOLVER == LDE_SOLVER_RK4 && i == S) {   // the forward kernel's own form of the RK4 update
            const float h6 = h * (1.0f / 6.0f);
            gx[i] = yv.x + h6 * (kx[0] + 2.0f * (kx[1] + kx[2]) + kx[3]);
            gy[i] = yv.y + h6 * (ky[0] + 2.0f * (ky[1] + ky[2]) + ky[3]);
          } else {
            float ax = A(i, 0) * kx[0], ay = A(i, 0) * ky[0];
#pragma unroll
            for (int q = 1; q < i; q++) {
              ax += A(i, q) * kx[q];
              ay += A(i, q) * ky[q];
            }
            gx[i] = yv.x + h * ax;
            gy[i] = yv.y + h * ay;
          }
        }
        hw_sincos(gx[i], sn[i], cs[i], noff);
        if (i < S) {
          kx[i] = gy[i];
          float acc = ngl * sn[i];
          if (KIND == 1) acc -= 0.7f * gy[i];
          ky[i] = acc;
        }
      }
      // ---- cotangents of the slopes: the FSAL slope carries the next step's k̄₁
      float kbx[S + 1], kby[S + 1];
#pragma unroll
      for (int i = 0; i < S; i++) kbx[i] = kby[i] = 0.f;
      kbx[S] = carry[0];
      kby[S] = carry[1];
      float ynx = 0.f, yny = 0.f;   // cotangent reaching y_n directly
      // the save times inside the step (t, tnew]: moments of Θ for the interpolant's polynomial weights
      float c1x = 0.f, c1y = 0.f, c2x = 0.f, c2y = 0.f, c3x = 0.f, c3y = 0.f, c4x = 0.f, c4y = 0.f;
      const float rh = fast_rcp(h);
      while (j >= 1) {
        const double tj = s_ts(j);
        if (!(tj > t)) break;
#if LDE_DISC_PF
        const float2 dj = dq[0];
#pragma unroll
        for (int u = 0; u + 1 < LDE_DISC_PF; u++) dq[u] = dq[u + 1];
        dq[LDE_DISC_PF - 1] = dz_out[(size_t)(j - LDE_DISC_PF > 0 ? j - LDE_DISC_PF : 0) * B + b];
#else
        const float2 dj = dz_out[(size_t)j * B + b];
#endif
        if (tj >= tnew || (j == T - 1 && last)) {
          yb[0] += dj.x;
          yb[1] += dj.y;
        } else {
          const float th = (float)(tj - t) * rh;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            ynx += dj.x;
            yny += dj.y;
            const float t2 = th * th, t3 = t2 * th, t4 = t2 * t2;
            c1x += th * dj.x; c1y += th * dj.y;
            c2x += t2 * dj.x; c2y += t2 * dj.y;
            c3x += t3 * dj.x; c3y += t3 * dj.y;
            c4x += t4 * dj.x; c4y += t4 * dj.y;
          } else {   // cubic Hermite on (y_n, k₁, y_{n+1}, f(y_{n+1}))
            const float om = 1.0f - th;
            const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
            const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
            ynx += h00 * dj.x; yny += h00 * dj.y;
            kbx[0] += (h10 * h) * dj.x; kby[0] += (h10 * h) * dj.y;
            yb[0] += h01 * dj.x; yb[1] += h01 * dj.y;
            kbx[S] += (h11 * h) * dj.x; kby[S] += (h11 * h) * dj.y;
          }
        }
        j--;
      }
      if (SOLVER == LDE_SOLVER_TSIT5) {   // k̄_i += h Σ_j b_i(Θ_j) Δ_j, b_1 = Θ + r₁₂Θ² + r₁₃Θ³ + r₁₄Θ⁴, b_i = r_i2Θ² + r_i3Θ³ + r_i4Θ⁴
        kbx[0] += h * (c1x + ts5::R1[0] * c2x + ts5::R1[1] * c3x + ts5::R1[2] * c4x);
        kby[0] += h * (c1y + ts5::R1[0] * c2y + ts5::R1[1] * c3y + ts5::R1[2] * c4y);
#pragma unroll
        for (int i = 0; i < 6; i++) {
          kbx[i + 1] += h * (ts5::R[i][0] * c2x + ts5::R[i][1] * c3x + ts5::R[i][2] * c4x);
          kby[i + 1] += h * (ts5::R[i][0] * c2y + ts5::R[i][1] * c3y + ts5::R[i][2] * c4y);
        }
      }
      // Jᵀk̄ at stage point i: J = [[0, 1], [ngl cos x, −b/m]], ∂f₂/∂L = gl2 sin x
      auto vjp = [&](int i, float kx_, float ky_, float& vx, float& vy) {
        vx = ngl * cs[i] * ky_;
        vy = kx_;
        if (KIND == 1) vy -= 0.7f * ky_;
        gth += gl2 * sn[i] * ky_;
      };
      {   // the FSAL evaluation at y_{n+1}
        float vx, vy;
        vjp(S, kbx[S], kby[S], vx, vy);
        yb[0] += vx;
        yb[1] += vy;
      }
#pragma unroll
      for (int i = 0; i < S; i++) {   // y_{n+1} = y_n + h Σ b_i k_i
        kbx[i] += (h * A(S, i)) * yb[0];
        kby[i] += (h * A(S, i)) * yb[1];
      }
      ynx += yb[0];
      yny += yb[1];
#pragma unroll
      for (int i = S - 1; i >= 1; i--) {
        float vx, vy;
        vjp(i, kbx[i], kby[i], vx, vy);
        ynx += vx;
        yny += vy;
#pragma unroll
        for (int q = 0; q < i; q++) {
          if (A(i, q) != 0.f) {
            kbx[q] += (h * A(i, q)) * vx;
            kby[q] += (h * A(i, q)) * vy;
          }
        }
      }
      carry[0] = kbx[0];
      carry[1] = kby[0];
      yb[0] = ynx;
      yb[1] = yny;
    }
    {   // k₁ of the first step = f(y_0): sn[0], cs[0] of the last iteration are those of y_0
      float vx = ngl * cs[0] * carry[1], vy = carry[0];
      if (KIND == 1) vy -= 0.7f * carry[1];
      gth += gl2 * sn[0] * carry[1];
      yb[0] += vx;
      yb[1] += vy;
    }
  }
  if (ret == LDE_RET_SUCCESS) {   // save time 0 is ẑ₀ itself
    const float2 d0 = dz_out[b];
    yb[0] += d0.x;
    yb[1] += d0.y;
  }
  const float qn = __int_as_float(0x7fc00000);
  const bool nanout = ret == LDE_RET_MAXITERS;
  dz0[b] = ret == LDE_RET_SUCCESS ? make_float2(yb[0], yb[1]) : (nanout ? make_float2(qn, qn) : make_float2(0.f, 0.f));
  dtheta[b] = ret == LDE_RET_SUCCESS ? gth : (nanout ? qn : 0.f);
  st_ret[b] = ret;
  const int nst = ret == LDE_RET_SUCCESS && T > 1 ? ns : 0;
  st_nfe[b] = nst * (2 * S + 1) + (nst ? 1 : 0);
  st_nacc[b] = nst;
  st_nrej[b] = 0;
}

// LDE_SENSE_DISCRETE at small batches: the steps of a trajectory side by side. With the record every accepted step is a known map
// y_n → y_{n+1} (and → the saves inside it), so its Jacobian needs nothing from its neighbours: a wave owns a trajectory, lanes (s, τ)
// carry the tangent τ ∈ {∂/∂y_x, ∂/∂y_y, ∂/∂L} through step s — the forward kernel's stage points rebuilt from y_n, then
//   G'_i = e_τ + h Σ a_iq K'_q,   K'_i = J_f(g_i) G'_i + ∂f/∂L(g_i)        (i = 0 … S; K'_S: the FSAL slope at y_{n+1})
// — and contract it with the cotangents of the save times inside the step (the moment sums of the sequential kernel above):
//   c_τ = Σ_j Δ_j · ∂ẑ_j/∂τ,   Y'_τ = ∂y_{n+1}/∂τ.
// (each of the step's three lanes takes a third of the save times; the partial sums meet in LDS).
// What is sequential is then an affine map per step: (ȳ, dθ) ← ((ȳ + e_s)·Y'_x + c_x, (ȳ + e_s)·Y'_y + c_y, dθ + (ȳ + e_s)·Y'_θ + c_θ),
// e_s = the cotangents of the saves ON the step's end, from the last step to the first — and the maps are COMPOSED (a suffix scan over the
// steps, see "the sweep" below) rather than applied one after the other: a chain of 21 dependent applications costs ≈ 90 cycles per step on a
// lone wave however the value travels between lanes. 21 steps per round (63 lanes); longer records take further rounds from the top. Same derivative as k_pend_adjoint_disc (forward
// instead of reverse accumulation inside a step: rounding differs, tests/test_gpu_discrete.py compares both with the oracle);
// same failure semantics and statistics.
constexpr int DTP_STEPS = 21;
__device__ __forceinline__ float dtp_from_lane(float v, int src) {   // lane ℓ ← v of lane src(ℓ) (ds_bpermute_b32: through the LDS crossbar, no LDS memory)
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}
template <int CTRL>
__device__ __forceinline__ float dtp_row_dpp(float v, float edge) {   // DPP row_shl:n (CTRL = 0x100 + n): lane ℓ ← v of lane ℓ + n of its row of 16, `edge` beyond the row
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dtp_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int KIND, int SOLVER>
__global__ void __launch_bounds__(64) k_pend_adjoint_disc_tp(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                             const double* __restrict__ ts_g, KOpts o,
                                                             const float2* __restrict__ dz_out, float2* __restrict__ dz0,
                                                             float* __restrict__ dtheta, int32_t* __restrict__ st_nfe,
                                                             int32_t* __restrict__ st_nacc, int32_t* __restrict__ st_nrej,
                                                             int32_t* __restrict__ st_ret) {
  extern __shared__ __attribute__((aligned(16))) double s_lds[];   // the save grid [T] | this trajectory's Δẑ [T] | partial sums [66][12]
  const int T = o.T, B = o.B, lane = threadIdx.x;
  // trajectory ↔ workgroup as in k_pend_forward_sh (the grid is a multiple of 8): the workgroup lands on the XCD whose L2 holds the
  // record and the ẑ the forward launch wrote for this trajectory
  const int b = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (b >= B) return;
  double* s_t = s_lds;
  float2* s_d = reinterpret_cast<float2*>(s_lds + T);
  float* s_red = reinterpret_cast<float*>(s_lds + 2 * T);   // [66][12]: where the three lanes of a step add up their shares of its save times (lane 63 reads rows 63 … 65)
  constexpr int S = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                              {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
  auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
  const StepRec& R = o.rec;
  const int sl = (lane * 43) >> 7, tau = lane - 3 * sl;   // lane = 3·(step of the round) + tangent
  PPROF(0);
  // ---- everything whose address is known now: the save grid, Δẑ, the record's first round, the step count
  const bool j0ok = lane < T;
  const double tsv = j0ok ? ts_g[lane] : 0.0;
  const float2 dv = j0ok ? dz_out[(size_t)lane * B + b] : make_float2(0.f, 0.f);
  const int spec_ok = sl < DTP_STEPS && sl < R.cap;
  double t0r = 0.0, dt0r = 0.0, tn0r = 0.0;
  float2 y0r = make_float2(0.f, 0.f);
  if (spec_ok) {
    t0r = R.t[(size_t)sl * B + b];
    dt0r = R.dt[(size_t)sl * B + b];
    y0r = reinterpret_cast<const float2*>(R.y)[(size_t)sl * B + b];
    if (sl + 1 < R.cap) tn0r = R.t[(size_t)(sl + 1) * B + b];
  }
  const int ns = __builtin_amdgcn_readfirstlane(R.n[b]);
  const float L = theta[b];
  const float2 y0 = z_out[b];
  if (j0ok) {
    s_t[lane] = tsv;
    s_d[lane] = dv;
  }
  for (int j = lane + 64; j < T; j += 64) {
    s_t[j] = ts_g[j];
    s_d[j] = dz_out[(size_t)j * B + b];
  }
  __syncthreads();
  PPROF(1);
  const float ngl = -10.0f / L, gl2 = 10.0f / (L * L);
  int ret = LDE_RET_SUCCESS;
  if (!isfinite(y0.x) || !isfinite(y0.y)) ret = LDE_RET_NONFINITE;      // a failed forward trajectory: zero gradient [REF GOKU.jl:114]
  else if (T > 1 && (ns < 1 || ns > R.cap)) ret = LDE_RET_MAXITERS;      // no usable record: NaN gradient (never a truncated sweep)
  float ax = 0.f, ay = 0.f, gth = 0.f;   // wave-uniform: the cotangent below the steps done so far, dθ
  if (ret == LDE_RET_SUCCESS && T > 1) {
    const double tbeg = s_t[0], tend = s_t[T - 1];
    const double jscale = (double)(T - 1) / (tend - tbeg);   // where a uniform grid would have a save time (a first guess, checked)
    const float e0x = tau == 0 ? 1.f : 0.f, e0y = tau == 1 ? 1.f : 0.f, pth = tau == 2 ? gl2 : 0.f;
    for (int r = (ns - 1) / DTP_STEPS; r >= 0; r--) {
      const int s = r * DTP_STEPS + sl;
      const bool act = sl < DTP_STEPS && s < ns;
      double t = t0r, dt = dt0r, tnew = tn0r;
      float2 yv = y0r;
      if (r > 0 && act) {
        t = R.t[(size_t)s * B + b];
        dt = R.dt[(size_t)s * B + b];
        yv = reinterpret_cast<const float2*>(R.y)[(size_t)s * B + b];
        if (s + 1 < ns) tnew = R.t[(size_t)(s + 1) * B + b];
      }
      const bool last = s == ns - 1;
      if (last) tnew = tend;
      if (!act) {
        yv = make_float2(0.f, 0.f);
        dt = 0.0;
      }
      const float h = (float)dt;
      // ---- the stage points again (the forward kernel's arithmetic on its inputs) and tangent τ through them
      const float noff = turn_anchor(yv.x);
      float gx[S + 1], gy[S + 1], kx[S], ky[S], sn, cs;
      float Kx[S + 1], Ky[S + 1], Yx = 0.f, Yy = 0.f;
      gx[0] = yv.x;
      gy[0] = yv.y;
#pragma unroll
      for (int i = 0; i <= S; i++) {
        float Gx = e0x, Gy = e0y;
        if (i > 0) {
          if (SOLVER == LDE_SOLVER_RK4 && i == S) {   // the forward kernel's own form of the RK4 update
            const float h6 = h * (1.0f / 6.0f);
            gx[i] = yv.x + h6 * (kx[0] + 2.0f * (kx[1] + kx[2]) + kx[3]);
            gy[i] = yv.y + h6 * (ky[0] + 2.0f * (ky[1] + ky[2]) + ky[3]);
            Gx = e0x + h6 * (Kx[0] + 2.0f * (Kx[1] + Kx[2]) + Kx[3]);
            Gy = e0y + h6 * (Ky[0] + 2.0f * (Ky[1] + Ky[2]) + Ky[3]);
          } else {
            float ax_ = A(i, 0) * kx[0], ay_ = A(i, 0) * ky[0], tx = A(i, 0) * Kx[0], ty = A(i, 0) * Ky[0];
#pragma unroll
            for (int q = 1; q < i; q++) {
              ax_ += A(i, q) * kx[q];
              ay_ += A(i, q) * ky[q];
              if (A(i, q) != 0.f) {
                tx += A(i, q) * Kx[q];
                ty += A(i, q) * Ky[q];
              }
            }
            gx[i] = yv.x + h * ax_;
            gy[i] = yv.y + h * ay_;
            Gx = e0x + h * tx;
            Gy = e0y + h * ty;
          }
        }
        hw_sincos(gx[i], sn, cs, noff);
        if (i < S) {
          kx[i] = gy[i];
          float acc = ngl * sn;
          if (KIND == 1) acc -= 0.7f * gy[i];
          ky[i] = acc;
        }
        Kx[i] = Gy;                              // J = [[0, 1], [ngl cos x, −b/m]], ∂f₂/∂L = gl2 sin x
        float kt = (ngl * cs) * Gx + pth * sn;
        if (KIND == 1) kt -= 0.7f * Gy;
        Ky[i] = kt;
        if (i == S) {
          Yx = Gx;
          Yy = Gy;
        }
      }
      PPROF(2);
      // ---- the save times in (t, tnew]: on the step's end → e_s; inside → the interpolant's weights
      // (One wave per SIMD: nothing hides a dependent instruction's latency but the wave's own independent instructions. So the four save
      //  times of a round trip are four independent, branch-free bodies — a save time outside the step enters with weight zero — on
      //  register pairs (x, y); only the ten running sums chain from body to body.)
      f32x2 ev = {0.f, 0.f}, ynv = {0.f, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f}, c4 = {0.f, 0.f};   // Tsit5: moments of Θ; RK4: the four Hermite sums
      if (act) {
        // the last save time ≤ tnew: where a uniform grid has it (exact there: the index in f64 with a margin far below a grid spacing),
        // checked against the grid itself in the round trip that also fetches the first four save times below it
        int j = (int)((tnew - tbeg) * jscale + 1e-6);
        j = j < 0 ? 0 : (j > T - 1 ? T - 1 : j);
        const float rh = fast_rcp(h);
        double tq[4];
        f32x2 dq[4];
        auto fetch = [&]() {
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int jj = j - 3 * u > 0 ? j - 3 * u : 0;
            tq[u] = s_t[jj];
            dq[u] = *reinterpret_cast<const f32x2*>(&s_d[jj]);
          }
        };
        {
          const double ttop = s_t[j], tup = s_t[j < T - 1 ? j + 1 : j];
          const bool ragged = !(ttop <= tnew && (j == T - 1 || tup > tnew));
          j -= tau;   // the step's three lanes share its save times: lane τ takes the τ-th, (τ + 3)-th, … from the last one down
          fetch();
          if (__builtin_expect(ragged, 0)) {   // (a ragged grid: walk to it)
            j += tau;
            while (j < T - 1 && s_t[j + 1] <= tnew) j++;
            while (j > 0 && s_t[j] > tnew) j--;
            j -= tau;
            fetch();
          }
        }
        for (;;) {
          bool in = false;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double tj = tq[u];
            in = j - 3 * u >= 1 && tj > t;                   // (the grid ascends: once false, false for every save time below)
            const bool end = tj >= tnew || (j - 3 * u == T - 1 && last);
            const f32x2 zero = {0.f, 0.f};
            const f32x2 de = in && end ? dq[u] : zero, di = in && !end ? dq[u] : zero;
            const float th = (float)(tj - t) * rh;
            ev += de;
            if (SOLVER == LDE_SOLVER_TSIT5) {
              const float t2 = th * th, t3 = t2 * th, t4 = t2 * t2;
              ynv += di;
              c1 += th * di;
              c2 += t2 * di;
              c3 += t3 * di;
              c4 += t4 * di;
            } else {   // cubic Hermite on (y_n, k₁, y_{n+1}, f(y_{n+1}))
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              ynv += h00 * di;
              c1 += (h10 * h) * di;
              c2 += h01 * di;
              c3 += (h11 * h) * di;
            }
          }
          if (!in) break;
          j -= 12;
          fetch();
        }
      }
      {   // the three lanes' partial sums meet in LDS (a wave's LDS operations execute in order) and are added in the order τ = 0, 1, 2
        float* rw = s_red + lane * 12;
        *reinterpret_cast<f32x4*>(rw) = f32x4{ev.x, ev.y, ynv.x, ynv.y};
        *reinterpret_cast<f32x4*>(rw + 4) = f32x4{c1.x, c1.y, c2.x, c2.y};
        *reinterpret_cast<f32x4*>(rw + 8) = f32x4{c3.x, c3.y, c4.x, c4.y};
        __syncthreads();
        const float* rr = s_red + (lane - tau) * 12;
        f32x4 q0 = *reinterpret_cast<const f32x4*>(rr), q1 = *reinterpret_cast<const f32x4*>(rr + 4), q2 = *reinterpret_cast<const f32x4*>(rr + 8);
#pragma unroll
        for (int g = 1; g < 3; g++) {
          q0 += *reinterpret_cast<const f32x4*>(rr + 12 * g);
          q1 += *reinterpret_cast<const f32x4*>(rr + 12 * g + 4);
          q2 += *reinterpret_cast<const f32x4*>(rr + 12 * g + 8);
        }
        __syncthreads();
        ev = f32x2{q0[0], q0[1]}; ynv = f32x2{q0[2], q0[3]};
        c1 = f32x2{q1[0], q1[1]}; c2 = f32x2{q1[2], q1[3]};
        c3 = f32x2{q2[0], q2[1]}; c4 = f32x2{q2[2], q2[3]};
      }
      const float ex = ev.x, ey = ev.y, ynx = ynv.x, yny = ynv.y;
      const float c1x = c1.x, c1y = c1.y, c2x = c2.x, c2y = c2.y, c3x = c3.x, c3y = c3.y, c4x = c4.x, c4y = c4.y;
      PPROF(3);
      float c = ynx * e0x + yny * e0y;
      if (SOLVER == LDE_SOLVER_TSIT5) {   // k̄_i = h Σ_j b_i(Θ_j) Δ_j, b_1 = Θ + r₁₂Θ² + r₁₃Θ³ + r₁₄Θ⁴, b_i = r_i2Θ² + r_i3Θ³ + r_i4Θ⁴
        const float b0x = h * (c1x + ts5::R1[0] * c2x + ts5::R1[1] * c3x + ts5::R1[2] * c4x);
        const float b0y = h * (c1y + ts5::R1[0] * c2y + ts5::R1[1] * c3y + ts5::R1[2] * c4y);
        c += b0x * Kx[0] + b0y * Ky[0];
#pragma unroll
        for (int i = 0; i < 6; i++) {
          const float bx = h * (ts5::R[i][0] * c2x + ts5::R[i][1] * c3x + ts5::R[i][2] * c4x);
          const float by = h * (ts5::R[i][0] * c2y + ts5::R[i][1] * c3y + ts5::R[i][2] * c4y);
          c += bx * Kx[i + 1] + by * Ky[i + 1];
        }
      } else {
        c += c1x * Kx[0] + c1y * Ky[0] + c2x * Yx + c2y * Yy + c3x * Kx[S] + c3y * Ky[S];
      }
      PPROF(4);
      // ---- the sweep. Lane p ≤ 20 gathers step p's affine map ȳ → ȳ·A_p + d_p (from lanes 3p + τ); lanes ≥ 21 are sources (A = 0, d = the
      // cotangent that enters the round). A chain of 21 dependent map applications costs ≈ 90 cycles per step here whichever way the value
      // travels (v_readlane → scalar → VALU, DPP wave_shl, DPP row_shl: measured), so the maps are composed instead: a suffix scan
      // within each row of 16 lanes — four levels, partner = lane + 1, 2, 4, 8 by DPP row_shl, eighteen independent multiply-adds per level,
      // a partner outside the row = the identity — leaves in lane p the composite of steps p … 15 (first row) and of steps p … 20 on the
      // source (second row, a constant). Lane 16's constant enters the first row's composites; what entered step p (for dθ) is what
      // left step p + 1. Steps beyond the record's end carry h = 0 and no save times: the identity map.
      const float dfl = __builtin_fmaf(ex, Yx, __builtin_fmaf(ey, Yy, c));   // (ȳ + e)·Y' + c = ȳ·Y' + (e·Y' + c)
      const int src = lane < DTP_STEPS ? 3 * lane : 63;
      float a00 = dtp_from_lane(Yx, src), a10 = dtp_from_lane(Yy, src), d0 = dtp_from_lane(dfl, src);               // new ȳ_x = ȳ_x a00 + ȳ_y a10 + d0
      float a01 = dtp_from_lane(Yx, src + 1), a11 = dtp_from_lane(Yy, src + 1), d1 = dtp_from_lane(dfl, src + 1);   // new ȳ_y = ȳ_x a01 + ȳ_y a11 + d1
      const float mtx = dtp_from_lane(Yx, src + 2), mty = dtp_from_lane(Yy, src + 2), mtd = dtp_from_lane(dfl, src + 2);   // dθ += ȳ_x mtx + ȳ_y mty + mtd
      if (lane >= DTP_STEPS) {
        a00 = a01 = a10 = a11 = 0.f;
        d0 = ax;
        d1 = ay;
      }
      PPROF(6);
#define LDE_DTP_LEVEL(CTRL)                                                                                     \
  {                                                                                                             \
    const float h00 = dtp_row_dpp<CTRL>(a00, 1.f), h01 = dtp_row_dpp<CTRL>(a01, 0.f);                           \
    const float h10 = dtp_row_dpp<CTRL>(a10, 0.f), h11 = dtp_row_dpp<CTRL>(a11, 1.f);                           \
    const float g0 = dtp_row_dpp<CTRL>(d0, 0.f), g1 = dtp_row_dpp<CTRL>(d1, 0.f);                               \
    const float n00 = __builtin_fmaf(h00, a00, h01 * a10), n01 = __builtin_fmaf(h00, a01, h01 * a11);           \
    const float n10 = __builtin_fmaf(h10, a00, h11 * a10), n11 = __builtin_fmaf(h10, a01, h11 * a11);           \
    d0 = __builtin_fmaf(g0, a00, __builtin_fmaf(g1, a10, d0));                                                  \
    d1 = __builtin_fmaf(g0, a01, __builtin_fmaf(g1, a11, d1));                                                  \
    a00 = n00; a01 = n01; a10 = n10; a11 = n11;                                                                 \
  }
      LDE_DTP_LEVEL(0x101)
      LDE_DTP_LEVEL(0x102)
      LDE_DTP_LEVEL(0x104)
      LDE_DTP_LEVEL(0x108)
#undef LDE_DTP_LEVEL
      PPROF(7);
      {
        const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0), 16));   // what leaves step 16 = what enters step 15
        const float hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1), 16));
        const float ox = __builtin_fmaf(hx, a00, __builtin_fmaf(hy, a10, d0));                 // what leaves step p (second row: A = 0, the constant itself)
        const float oy = __builtin_fmaf(hx, a01, __builtin_fmaf(hy, a11, d1));
        const float qx = dtp_row_dpp<0x101>(ox, hx), qy = dtp_row_dpp<0x101>(oy, hy);           // what entered step p
        float gp = lane < DTP_STEPS ? __builtin_fmaf(qx, mtx, __builtin_fmaf(qy, mty, mtd)) : 0.f;
        gp = dtp_dpp_add<0x111>(gp);   // row_shr:1, 2, 4, 8: inclusive sums within the rows of 16 lanes
        gp = dtp_dpp_add<0x112>(gp);
        gp = dtp_dpp_add<0x114>(gp);
        gp = dtp_dpp_add<0x118>(gp);
        gth += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gp), 15)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gp), 31));
        ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ox), 0));
        ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oy), 0));
      }
    }
  }
  PPROF(5);
  if (lane == 0) {
    const float2 d0 = s_d[0];   // save time 0 is ẑ₀ itself
    const float qn = __int_as_float(0x7fc00000);
    const bool ok = ret == LDE_RET_SUCCESS, nanout = ret == LDE_RET_MAXITERS;
    dz0[b] = ok ? make_float2(ax + d0.x, ay + d0.y) : (nanout ? make_float2(qn, qn) : make_float2(0.f, 0.f));
    dtheta[b] = ok ? gth : (nanout ? qn : 0.f);
    st_ret[b] = ret;
    const int nst = ok && T > 1 ? ns : 0;
    st_nfe[b] = nst * (2 * S + 1) + (nst ? 1 : 0);
    st_nacc[b] = nst;
    st_nrej[b] = 0;
  }
}

int launch_pend_adjoint_disc(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev, const KOpts& o,
                             const float* dz_out, float* dz0, float* dtheta, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                             hipStream_t stream, const PendTune& tn) {
  if (!o.rec.n || !o.rec.y) return LDE_ERR_INVALID_ARG;
  const int block = pick_block(o.B), grid = (o.B + block - 1) / block;
  const size_t shm = o.T <= TS_LDS_MAX ? (size_t)o.T * sizeof(double) : 0;
  // the steps side by side (a wave per trajectory) while the chip has waves to spare: option "pend_disc_tp_max_b"
  const bool tp = o.T > 1 && o.T <= 3840 && o.B <= tn.disc_tp_max_b;   // (16·T + 3 168 bytes of LDS: within the 64 KB a launch gets without asking)
  g_pend_last[1] = tp ? "k_pend_adjoint_disc_tp" : "k_pend_adjoint_disc<";
#define LDE_LAUNCH(K, S)                                                                                                         \
  do {                                                                                                                           \
    if (tp)                                                                                                                      \
      hipLaunchKernelGGL((k_pend_adjoint_disc_tp<K, S>), dim3(((o.B + 7) / 8) * 8), dim3(64), (size_t)o.T * 16 + 66 * 12 * sizeof(float), stream, (const float2*)z_out, theta, ts_dev, o, \
                         (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret);                                     \
    else                                                                                                                         \
      hipLaunchKernelGGL((k_pend_adjoint_disc<K, S>), dim3(grid), dim3(block), shm, stream, (const float2*)z_out, theta, ts_dev, o, \
                         (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret);                                     \
  } while (0)
  if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
  else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
  else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

// ---- time-parallel adjoint's large-batch form: the same interval-by-interval control, streamed per trajectory -------------------
// Beyond a few 10⁴ trajectories there is no parallelism left to win from giving every save interval its own lane, and the
// two-kernel form above pays for it: six operator planes + an info word per (trajectory, interval) written and read back —
// 2.7 KB per trajectory against 412 B of algorithmic traffic, 344 GB/s algorithmic at B = 2²⁰ — and integrates two basis
// vectors of λ where one actual λ is wanted. Here a lane owns a trajectory and walks its T−1 intervals from the last to the
// first, each exactly as pend_interval_operator treats it (z reset to the saved ẑ(t_{j+1}), first attempt = the whole
// interval, adaptive inside, no step size carried across the save time), but on the 5-state [z | λ | g] itself: the only
// traffic is the algorithmic one (Δẑ, and ẑ as checkpoints), every access coalesced over the batch index, the next
// interval's ẑ / Δẑ requested before the current one is integrated. Arithmetic on register pairs (z, λ, (g,·)).
template <int KIND>
struct PendBwdPair {
  float ngl, gl2, noff;
  __device__ __forceinline__ explicit PendBwdPair(float L) : ngl(-10.0f / L), gl2(10.0f / (L * L)), noff(0.f) {}
  __device__ __forceinline__ void anchor(float x0) { noff = turn_anchor(x0); }
  // ż = f(z),  λ̇ = −(∂f/∂z)ᵀλ,  ġ = −(∂f/∂L)ᵀλ
  __device__ __forceinline__ void ev(const f32x2 (&y)[3], f32x2 (&dy)[3]) const {
    float s, c;
    hw_sincos(y[0].x, s, c, noff);
    float acc = ngl * s, v1 = y[1].x;
    if (KIND == 1) {
      acc -= 0.7f * y[0].y;
      v1 -= 0.7f * y[1].y;
    }
    dy[0] = f32x2{y[0].y, acc};
    dy[1] = f32x2{-(ngl * c * y[1].y), -v1};
    dy[2] = f32x2{-(gl2 * s * y[1].y), 0.f};
  }
};

// one Tsit5 attempt on three register pairs; k[0] = f(y) on entry; returns the mean square of the scaled error over the 5 entries
template <class F, bool ADAPT>
__device__ __forceinline__ float tsit5_attempt_pair3(F& f, float h, const f32x2 (&y)[3], f32x2 (&k)[7][3], f32x2 (&yn)[3], const KOpts& o) {
  f32x2 tmp[3];
#pragma unroll
  for (int s = 1; s < 6; s++) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      f32x2 acc = k[0][i] * ts5::A[s][0];
#pragma unroll
      for (int j = 1; j < s; j++) acc += k[j][i] * ts5::A[s][j];
      tmp[i] = y[i] + acc * h;
    }
    f.ev(tmp, k[s]);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    f32x2 acc = k[0][i] * ts5::A[6][0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc += k[j][i] * ts5::A[6][j];
    yn[i] = y[i] + acc * h;
  }
  f.ev(yn, k[6]);
  if (!ADAPT) return 0.f;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    f32x2 e = k[0][i] * ts5::BT[0];
#pragma unroll
    for (int j = 1; j < 7; j++) e += k[j][i] * ts5::BT[j];
    e *= h;
    const f32x2 sk = f32x2{fmaxf(fabsf(y[i].x), fabsf(yn[i].x)), fmaxf(fabsf(y[i].y), fabsf(yn[i].y))} * o.reltol + o.abstol;
    const f32x2 r = e * f32x2{fast_rcp(sk.x), fast_rcp(sk.y)};
    s2 += r.x * r.x;
    if (i < 2) s2 += r.y * r.y;   // (the second half of the g pair is padding)
  }
  return s2 * 0.2f;
}

template <int KIND, int SOLVER>
__global__ void __launch_bounds__(256) k_pend_adjoint_stream(const float2* __restrict__ z_out, const float* __restrict__ theta,
                                                             const double* __restrict__ ts_g, KOpts o,
                                                             const float2* __restrict__ dz_out, float2* __restrict__ dz0,
                                                             float* __restrict__ dtheta, int32_t* __restrict__ st_nfe,
                                                             int32_t* __restrict__ st_nacc, int32_t* __restrict__ st_nrej,
                                                             int32_t* __restrict__ st_ret) {
  const int T = o.T, B = o.B;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  PendBwdPair<KIND> f(theta[b]);
  float2 zc = z_out[(size_t)(T - 1) * B + b];
  const float2 dT = dz_out[(size_t)(T - 1) * B + b];
  f32x2 lam = {dT.x, dT.y};
  float g = 0.f;
  int ret = (isfinite(zc.x) && isfinite(zc.y)) ? LDE_RET_SUCCESS : LDE_RET_NONFINITE;   // failed forward trajectory: zero gradient
  int nacc = 0, nrej = 0;
  float2 zn = zc, dn = dT;
  if (T > 1) {
    zn = z_out[(size_t)(T - 2) * B + b];
    dn = dz_out[(size_t)(T - 2) * B + b];
  }
  const float dtmin = (float)o.dtmin;
  for (int j = T - 2; j >= 0 && ret == LDE_RET_SUCCESS; j--) {
    const float2 zj = zn, dj = dn;   // ẑ(t_j) (the next interval's start state) and the jump Δ_j
    if (j > 0) {
      zn = z_out[(size_t)(j - 1) * B + b];
      dn = dz_out[(size_t)(j - 1) * B + b];
    }
    const double t0 = ts_g[j], t1 = ts_g[j + 1];
    // integrate [z | λ | g] from t1 down to t0; first attempt: the whole interval
    f32x2 y[3] = {f32x2{zc.x, zc.y}, lam, f32x2{g, 0.f}}, yn[3], k[7][3];
    const float len = (float)(t1 - t0);
    float left = len, dt = o.adaptive ? len : fminf((float)o.dt_fixed, len);
    float lqold = -13.287712379549449f;
    long long iters = 0;
    f.anchor(y[0].x);
    f.ev(y, k[0]);
    for (;;) {
      if (iters++ >= o.maxiters) { ret = LDE_RET_MAXITERS; break; }
      const bool hit = dt >= left * 0.99999988f;
      const float hmag = hit ? left : dt;
      f.anchor(y[0].x);
      float msq = 0.f;
      if (SOLVER == LDE_SOLVER_TSIT5) {
        if (o.adaptive) msq = tsit5_attempt_pair3<PendBwdPair<KIND>, true>(f, -hmag, y, k, yn, o);
        else (void)tsit5_attempt_pair3<PendBwdPair<KIND>, false>(f, -hmag, y, k, yn, o);
      } else {   // classical RK4 on the pairs
        f32x2 tmp[3];
        const float h = -hmag, hh = 0.5f * h;
#pragma unroll
        for (int i = 0; i < 3; i++) tmp[i] = y[i] + k[0][i] * hh;
        f.ev(tmp, k[1]);
#pragma unroll
        for (int i = 0; i < 3; i++) tmp[i] = y[i] + k[1][i] * hh;
        f.ev(tmp, k[2]);
#pragma unroll
        for (int i = 0; i < 3; i++) tmp[i] = y[i] + k[2][i] * h;
        f.ev(tmp, k[3]);
#pragma unroll
        for (int i = 0; i < 3; i++) yn[i] = y[i] + (k[0][i] + (k[1][i] + k[2][i]) * 2.0f + k[3][i]) * (h * (1.0f / 6.0f));
        f.ev(yn, k[4]);
      }
      const float size = fabsf(yn[0].x) + fabsf(yn[0].y) + fabsf(yn[1].x) + fabsf(yn[1].y) + fabsf(yn[2].x);
      const bool fin = size < __builtin_inff() && msq == msq;
      if (!fin) {
        if (o.adaptive && hmag > dtmin) { nrej++; dt = hmag * o.qmin; continue; }
        ret = LDE_RET_NONFINITE;
        break;
      }
      if (o.adaptive && (msq > 1.0f || !hit)) {   // an accepted step that ends the interval needs no next step size
        const float l = 0.5f * __builtin_amdgcn_logf(msq);
        if (msq > 1.0f) {
          nrej++;
          dt = hmag * fast_rcp(fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l) * o.inv_gamma));
          if (dt < dtmin) { ret = LDE_RET_DTMIN; break; }
          continue;
        }
        const float q = fmaxf(o.q_lo, fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * l - o.beta2 * lqold) * o.inv_gamma));
        lqold = fmaxf(l, -13.287712379549449f);
        dt = fminf(hmag * fast_rcp(q), len);
      }
      nacc++;
#pragma unroll
      for (int i = 0; i < 3; i++) y[i] = yn[i];
      if (hit) break;
      left -= hmag;
      constexpr int FS = (SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4;
#pragma unroll
      for (int i = 0; i < 3; i++) k[0][i] = k[FS][i];
      if (!o.adaptive) dt = (float)o.dt_fixed;
    }
    // the jump at t_j; z restarts from the checkpoint
    lam = f32x2{y[1].x + dj.x, y[1].y + dj.y};
    g = y[2].x;
    zc = zj;
  }
  dz0[b] = ret ? make_float2(0.f, 0.f) : make_float2(lam.x, lam.y);
  dtheta[b] = ret ? 0.f : g;
  st_nacc[b] = nacc;
  st_nrej[b] = nrej;
  st_nfe[b] = (T - 1) + ((SOLVER == LDE_SOLVER_TSIT5) ? 6 : 4) * (nacc + nrej);
  st_ret[b] = ret;
}

// does the time-parallel adjoint need the operator planes in HBM for this shape? (only the two-kernel form does)
// ONE decision, ONE caching policy (the switches are read once per process) for both the buffer reservation and the launcher: two
// readers with different caching could disagree after an environment change and send a NULL `ops` into the two-kernel form.
enum { PEND_ADJ_FUSED = 0, PEND_ADJ_STREAM = 1, PEND_ADJ_TWO_KERNEL = 2 };
static int pend_adjoint_form(int B, int T) {
  constexpr int fused_max_b = 24576;   // measured (abl/adj_B.py): fused 9.5 µs vs stream 39 µs at 4096, 48 vs 41 µs at 32768
  constexpr bool stream_on = true;
  if (T > 1 && T - 1 <= 1024 && B <= fused_max_b) return PEND_ADJ_FUSED;
  return stream_on ? PEND_ADJ_STREAM : PEND_ADJ_TWO_KERNEL;
}
bool pend_adjoint_needs_ops(int B, int T) { return pend_adjoint_form(B, T) == PEND_ADJ_TWO_KERNEL; }

int launch_pend_adjoint_par(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev,
                            const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* ops, int32_t* info,
                            int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret, hipStream_t stream) {
  const int form = pend_adjoint_form(o.B, o.T);
  g_pend_last[1] = form == PEND_ADJ_FUSED ? "k_pend_adjoint_fused" : (form == PEND_ADJ_STREAM ? "k_pend_adjoint_stream" : "k_pend_adjoint_par");
  if (form == PEND_ADJ_FUSED) {   // fused: one workgroup per trajectory, one lane per interval
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
  if (form == PEND_ADJ_STREAM) {   // large batches: one lane per trajectory, interval by interval (k_pend_adjoint_stream)
    const int block = 256, grid = (o.B + block - 1) / block;
#define LDE_LAUNCH(K, S)                                                                                                  \
  hipLaunchKernelGGL((k_pend_adjoint_stream<K, S>), dim3(grid), dim3(block), 0, stream, (const float2*)z_out, theta, ts_dev, o, \
                     (const float2*)dz_out, (float2*)dz0, dtheta, nfe, nacc, nrej, ret)
    if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(0, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_RK4) LDE_LAUNCH(0, LDE_SOLVER_RK4);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_TSIT5) LDE_LAUNCH(1, LDE_SOLVER_TSIT5);
    else if (kind == LDE_RHS_PENDULUM_FRICTION && solver == LDE_SOLVER_RK4) LDE_LAUNCH(1, LDE_SOLVER_RK4);
    else return LDE_ERR_UNSUPPORTED;
#undef LDE_LAUNCH
    return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
  }
  if (!ops || !info) return LDE_ERR_INVALID_ARG;   // the two-kernel form keeps its operator planes in HBM
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

#if LDE_PEND_PROF
extern "C" int lde_debug_pend_prof(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lde::g_pprof), sizeof(long long) * 32) == hipSuccess ? 0 : -4;
}
#endif

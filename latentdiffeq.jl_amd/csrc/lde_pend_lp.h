// lde_pend_lp.h — k_pend_forward_lp: the metric's forward solve (frictionless pendulum, Tsit5, adaptive; a trajectory per workgroup, B ≤ 1 024:
// the launch code's thresholds) with the stepping wave's 64 redundant lanes put to work (round 6; included by lde_pendulum.hip behind k_pend_forward_sh, whose protocol it keeps).
//
// k_pend_forward_sh's stepping wave carries the SAME solve in all 64 lanes and its duration is one dependent-instruction chain: six stage
// evaluations one behind the other, each (stage sum → angle in turns → v_sin_f32 → ·(−g/L) → into the later stages' sums), ≈ 757 cycles per
// accepted step. Two observations remove half of that chain:
//
//  1. For f = (v, s(x)), s = −(g/L)·sin x, Tsit5 is a Nyström scheme: the stage ANGLES need no stage velocities,
//         x_i = x + c_i·h·v + h²·Σ_{l ≤ i−2} Ā_il s_l,   Ā = A·A,
//     so s_i depends on s_1 … s_{i−2} only: the six evaluations of a step are two interleaved chains (2 → 4 → 6) and (3 → 5 → 7) of depth
//     THREE. (The register-pair form hides this: a v_pk_fma_f32 on (x, v) waits for s_{i−1} although only its v half needs it.)
//  2. The two chains run in the two lanes of a lane pair under ONE instruction stream: even lanes evaluate stages 3, 5 and the new state's
//     slope 7 (= the next step's first: FSAL), odd lanes 2, 4, 6, with lane-dependent tableau coefficients in registers; a level is
//     (sum → v_sin_f32 → quad_perm swap), three levels per step. Even lanes also carry x, odd lanes v: y_new, the error estimate's two
//     components and their scaled squares are one instruction each for both components, and the two squares meet by one DPP add.
//
// Per accepted step ≈ 85 instructions with 7 quarter-rate ones on a chain of ≈ 30 (k_pend_forward_sh: 128 / 9 / ≈ 56). The arithmetic is the
// same method with the same coefficients — derived here in double precision from the tableau (lp::TB; abl/rkn_coeffs.py checks the
// identities in float64) — in another order of operations: results agree with k_pend_forward_sh like two correct f32 solves
// (tests/test_gpu_pendulum.py: both held to the oracle by the same gates; option "pend_lp" = 0 selects the older kernel).
//
// The record a step leaves for the dense-output waves is {σ₂ … σ₇, y_new, h} (σ_i = sin x_i): the helpers no longer re-run the stages —
// the interpolant's polynomials follow from the sines directly, P_m = (h·(−g/L)·Σ_l RA_ml σ_l, (−g/L)·Σ_j r_jm σ_j), RA = rᵀA — so the
// last step's slopes need no second hand-over (k_pend_forward_sh's s_klast) and a helper's work per step is ≈ 60 shallow instructions.
// [REF src/models/GOKU.jl:98-130: the solve this replaces; examples/pendulum_friction-less/pendulum.jl:19-26: the right-hand side]
#pragma once

namespace lp {
// Tsit5 in double precision, 1-based (the f32 tables of lde_device.h are these constants rounded)
struct Tab {
  double C[8];        // c_i = Σ_j a_ij
  double Ab[8][8];    // Ā = A·A; row 7 = B̄ (x_new)
  double Et[8];       // Ẽ_l = Σ_j b̃_j a_jl (a_7l = b_l): err_x = h² Σ_l Ẽ_l s_l
  double RA[3][8];    // RA_ml = Σ_j r_jm a_jl: the dense output's Θ², Θ³, Θ⁴ polynomials, x component
  double RR[8][3];    // r_jm
  double A[8][8], BT[8];
};
constexpr Tab make_tab() {
  Tab t{};
  constexpr double a[8][8] = {
      {0, 0, 0, 0, 0, 0, 0, 0},
      {0, 0, 0, 0, 0, 0, 0, 0},
      {0, 0.161, 0, 0, 0, 0, 0, 0},
      {0, -0.008480655492356989, 0.335480655492357, 0, 0, 0, 0, 0},
      {0, 2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0, 0},
      {0, 5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0, 0},
      {0, 5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0, 0},
      {0, 0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0}};
  constexpr double bt[8] = {0, -0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                            0.5823571654525552, -0.45808210592918697, 0.015151515151515152};
  constexpr double rr[8][3] = {{0, 0, 0},
                               {-2.763706197274826, 2.9132554618219126, -1.0530884977290216},
                               {0.13169999999999998, -0.2234, 0.1017},
                               {3.9302962368947516, -5.941033872131505, 2.490627285651253},
                               {-12.411077166933676, 30.33818863028232, -16.548102889244902},
                               {37.50931341651104, -88.1789048947664, 47.37952196281928},
                               {-27.896526289197286, 65.09189467479366, -34.87065786149661},
                               {1.5, -4.0, 2.5}};
  for (int i = 0; i < 8; i++) {
    t.BT[i] = bt[i];
    double c = 0;
    for (int j = 0; j < 8; j++) { t.A[i][j] = a[i][j]; c += a[i][j]; }
    t.C[i] = c;
    for (int m = 0; m < 3; m++) t.RR[i][m] = rr[i][m];
  }
  for (int i = 0; i < 8; i++)
    for (int l = 0; l < 8; l++) {
      double s = 0;
      for (int j = 0; j < 8; j++) s += a[i][j] * a[j][l];
      t.Ab[i][l] = s;
    }
  for (int l = 0; l < 8; l++) {
    double s = 0;
    for (int j = 0; j < 8; j++) s += bt[j] * a[j][l];
    t.Et[l] = s;
    for (int m = 0; m < 3; m++) {
      double q = 0;
      for (int j = 0; j < 8; j++) q += rr[j][m] * a[j][l];
      t.RA[m][l] = q;
    }
  }
  return t;
}
constexpr Tab TB = make_tab();
// the structure the kernel relies on: x_i needs s_1 … s_{i−2} only
static_assert(TB.Ab[2][1] == 0 && TB.Ab[3][2] == 0 && TB.Ab[4][3] == 0 && TB.Ab[5][4] == 0 && TB.Ab[6][5] == 0 && TB.Ab[7][6] == 0, "Nyström structure");
static_assert(TB.C[6] > 0.999999999 && TB.C[6] < 1.000000001 && TB.C[7] > 0.999999999 && TB.C[7] < 1.000000001, "c₆ = c₇ = 1");

constexpr int COPIES = 16;         // copies of a step's record (lane ℓ uses copy ℓ >> 2: every access a 16-byte chunk of its own)
constexpr int RECF = 12;           // floats per copy: {σ₃, σ₅, σ₇, x_new | σ₂, σ₄, σ₆, v_new | h, h, ·, ·}
constexpr int STEPF = COPIES * RECF;

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {   // the value of the lane CTRL maps this lane to (quad_perm patterns: every lane has a source)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr float TWO_PI = 6.283185307179586f;
#ifndef LDE_LP_SIN
#define LDE_LP_SIN 0   // diagnostic (abl/lp_accuracy.py): 1 = the polynomial sine on 2π·angle instead of v_sin_f32
#endif
__device__ __forceinline__ float sin_turns(float t) { return LDE_LP_SIN ? fast_sin(t * TWO_PI) : __builtin_amdgcn_sinf(t); }
constexpr int SWAP = 0xB1;   // quad_perm [1,0,3,2]: the partner lane
constexpr int EVEN = 0xA0;   // quad_perm [0,0,2,2]: the pair's even lane
constexpr int ODD = 0xF5;    // quad_perm [1,1,3,3]: the pair's odd lane
}  // namespace lp

#if LDE_PEND_PROF   // stamps by the stepping wave's first lane, whichever wave of the workgroup that is
#define LPPROF(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { g_pprof[2 * (i)] = wall_clock64(); g_pprof[2 * (i) + 1] = __builtin_readcyclecounter(); } } while (0)
#define LPPROF_VAL(i, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_pprof[i] = (v); } while (0)
#else
#define LPPROF(i) do { } while (0)
#define LPPROF_VAL(i, v) do { } while (0)
#endif
// Measured at B = 256 on one box (abl/ab_libs.sh, three alternations; M trajectories/s at K = 200 | K = 20): 3 helpers 22.85 | 20.75 (round 6's
// first form); 4: 23.61 | 21.52; 5: 23.57 | 21.44; 6: 23.5 | 21.4; 7: 23.1 | 21.0; 9: 22.5 | 20.6; 11: 22.0; 15: 21.6 — a helper's own record costs
// ≈ 1 350 cycles and three of them had ≈ 1 750 per record between them: they ran behind, and what they were behind by was the launch's tail
// (1.3–1.6 → 0.6–1.1 µs behind the stepper's last step). With five waves on four SIMDs one SIMD holds two: the stepping wave as wave 2
// measured +0.4 % over wave 0 (23.72 | 21.64) and as the LAST wave −7 % (it starts last). Also measured, slower, not kept: the helper
// walking up to its next record before that is published (−1.4 %), the helpers writing the step records instead of the stepper's
// replay (−3 %: three more global stores on the last record's path).
// Beyond one workgroup per CU (B > 256; the launch code's thresholds, abl/lp_midB.py): the same kernel serves B ≤ 1 024 — up to four
// workgroups share a CU, their stepping waves still mostly a SIMD apart — with THREE dense-output waves from B > 512 on (forward µs per
// launch with 4 | 3 helpers: B = 512 7.7 | 8.2, 768 9.6 | 9.5, 1 024 13.2 | 10.5, 2 048 23.0 | 18.2; k_pend_forward_ws: 15.4–16.1 throughout).
template <int NH>
struct LpShape {
  static constexpr int LP_NH = NH;              // dense-output waves
  static constexpr int LP_SW = NH >= 4 ? 2 : 0;   // which wave of the workgroup steps
};
template <bool REC, int NH = 4>   // REC: the instantiation that writes step records (LDE_SENSE_DISCRETE, "step_trace"); NH: dense-output waves
__global__ void __launch_bounds__(64 * (1 + NH)) k_pend_forward_lp(const float2* __restrict__ z0, const float* __restrict__ theta,
                                                         const double* __restrict__ ts_g, KOpts o,
                                                         float2* __restrict__ z_out, int32_t* __restrict__ retcode,
                                                         int32_t* __restrict__ st_nfe, int32_t* __restrict__ st_nacc,
                                                         int32_t* __restrict__ st_nrej, int32_t* __restrict__ st_ret) {
  using namespace lp;
  constexpr int LP_NH = LpShape<NH>::LP_NH, LP_SW = LpShape<NH>::LP_SW;
  __shared__ __attribute__((aligned(16))) float s_rec[SH_CAP * STEPF];
  __shared__ int s_cnt[64];
  __shared__ int s_fin;    // 0: stepping; 1: the round is over, another follows; 2: done
  __shared__ int s_fail;
  const int T = o.T, B = o.B, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = gridDim.x >> 3;   // XCD-aware trajectory ↔ workgroup map (as k_pend_forward_sh); the grid is a multiple of 8
  const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const bool valid = b < B;
  const int bc = valid ? b : B - 1;
  const double t_first = o.t_first, tend = o.t_last;
  const float2 zi = z0[bc];
  PendFwd<0> f(theta[bc]);
  const float ngl = f.ngl;
  if (tid < 64) s_cnt[tid] = 0;
  if (tid == 0) { s_fin = 0; s_fail = 0; }
  __syncthreads();
  constexpr int NS = 6;  // RHS evaluations per attempt
  float* const myrec = s_rec + (lane >> 2) * RECF;   // this lane's copy of record 0

  if (w == LP_SW) {
    // ================= the stepper =================
    LPPROF(8);
    const bool ev = (lane & 1) == 0;
    auto sel = [&](double a, double c) -> float { return ev ? (float)a : (float)c; };
    // lane-dependent coefficients: even lanes stages 3, 5, new (7); odd lanes stages 2, 4, 6. "own" = this lane's sine of the level
    // before, "par" = the partner lane's.
    // level sums: in1 (stage 3 | 2) alone, P = {in2, in3} (stages 5 | 4, the new state's slope 7 | stage 6) as a register pair
    const float K11 = sel(TB.Ab[3][1], 0.0);
    const float GK11 = K11 * (ngl * INV_2PI);
    const f32x2 C12 = {sel(TB.C[3], TB.C[2]), sel(TB.C[5], TB.C[4])};
    const f32x2 KP1 = {sel(TB.Ab[5][1], TB.Ab[4][1]), sel(TB.Ab[7][1], TB.Ab[6][1])};      // · σ₁
    const f32x2 KPo1 = {sel(TB.Ab[5][3], TB.Ab[4][2]), sel(TB.Ab[7][3], TB.Ab[6][2])};     // · own sine of level 1
    const f32x2 KPp1 = {sel(TB.Ab[5][2], 0.0), sel(TB.Ab[7][2], TB.Ab[6][3])};             // · the partner's
    const float K3o2 = sel(TB.Ab[7][5], TB.Ab[6][4]), K3p2 = sel(TB.Ab[7][4], 0.0);
    // Q = {new state's sum, error estimate's sum}: even lanes (x_new: B̄, err_x: Ẽ), odd lanes (v_new: b, err_v: b̃)
    const f32x2 Q1 = {sel(TB.Ab[7][1], TB.A[7][1]), sel(TB.Et[1], TB.BT[1])};
    const f32x2 Qo1 = {sel(TB.Ab[7][3], TB.A[7][2]), sel(TB.Et[3], TB.BT[2])}, Qp1 = {sel(TB.Ab[7][2], TB.A[7][3]), sel(TB.Et[2], TB.BT[3])};
    const f32x2 Qo2 = {sel(TB.Ab[7][5], TB.A[7][4]), sel(TB.Et[5], TB.BT[4])}, Qp2 = {sel(TB.Ab[7][4], TB.A[7][5]), sel(TB.Et[4], TB.BT[5])};
    const f32x2 Qo3 = {sel(0.0, TB.A[7][6]), sel(0.0, TB.BT[6])};
    const float Ep3 = sel(TB.Et[6], TB.BT[7]);
    const float evm = ev ? 1.f : 0.f;
    const float glt = ngl * INV_2PI;   // −g/L per turn: the stepper's state is in TURNS — ξ = x/2π (even lanes), ω = v/2π (odd lanes)
    // the controller works on 1/q = γ·2^(β₂·l_old − β₁·l), l = log₂ EEst = ½·log₂ m2 − ½, clamped to [qmin, qmax]: in terms of lg = log₂ m2,
    // 1/q = 2^(cb·lg_old − ca·lg + c0) with ca = β₁/2, cb = β₂/2, c0 = log₂ γ + (β₁ − β₂)/2 — one fma, one v_exp_f32, one v_med3_f32 (no reciprocal)
    const float ca = 0.5f * o.beta1, cb = 0.5f * o.beta2, c0 = 0.5f * (o.beta1 - o.beta2) - log2f(o.inv_gamma);
    const float iq_lo = o.qmin, iq_hi = 1.0f / o.q_lo;
    float abst = o.abstol * INV_2PI, relt = o.reltol;   // (the scaled error is a ratio: abstol in the state's unit)
    asm volatile("" : "+v"(abst), "+v"(relt));   // (kept in registers: as kernel-argument scalars two of them in one v_fma need a copy per step)

    int ret = LDE_RET_SUCCESS, nfe = 0, nacc = 0, nrej = 0;
    double t = t_first;
    float dt = 0.f;
    constexpr float LQ_MIN = -13.287712379549449f;   // log₂ of qoldinit = 1e-4
    constexpr float LG_MIN = 2.0f * LQ_MIN + 1.0f;   // the same bound on lg = log₂ m2 = 2·l + 1
    float lgold = LG_MIN;
    const int maxit = o.maxiters > 0x7fffffffLL ? 0x7fffffff : (int)o.maxiters;
    const float dtmin = (float)o.dtmin;
    if (valid && lane == 0) z_out[b] = zi;  // ts[0] is saved as ẑ₀ itself
    const double dtmax_d = tend - t;
    const float dtmax = (float)dtmax_d;
    float s1 = hw_sin(zi.x, turn_anchor(zi.x));   // σ₁ = sin x₀ (uniform)
    nfe = 1;
    if (o.dt_fixed > 0) dt = (float)fmin(o.dt_fixed, dtmax_d);
    else {
      f.anchor(zi.x);
      const float ya[2] = {zi.x, zi.y}, fa[2] = {zi.y, ngl * s1};
      dt = (float)init_dt<2>(f, ya, fa, 1.0f, dtmax_d, o);
      nfe++;
    }
    float yc = (ev ? zi.x : zi.y) * INV_2PI;   // even lanes ξ = x/2π, odd lanes ω = v/2π
    bool active = __any(t < tend) && maxit > 0;   // (votes: scalar from here on)
    if (__any(t < tend) && !active) ret = LDE_RET_MAXITERS;
    LPPROF(9);
    // the step about to be attempted: its size h and whether it reaches the end (decided when the step BEFORE it was accepted — with the
    // f64 time arithmetic done while that step's stages were in flight — so that neither sits between two steps' dependent chains)
    float rem = (float)(tend - t);
    bool last = __any(dt >= rem * 0.99999988f);
    float h = last ? rem : dt;
    for (;;) {   // rounds
      int n = 0;
      float* rp = myrec + (lane & 1) * 4;
      const double t_round = t;   // (REC) where this round's first record starts
      const float xr = dpp<EVEN>(yc), vr = dpp<ODD>(yc);
      int lim = min(SH_CAP, maxit - (nacc + nrej));
      bool go = active && lim > 0;
      while (go) {
        float xa = dpp<EVEN>(yc);
        const float wa = dpp<ODD>(yc);
        asm volatile("" : "+v"(xa));   // (ROCm 7.2's DPP combiner folds the lane move into v_rndne_f32_dpp AND drops it for the subtraction's other operand, which then reads a stale register)
        const float xi0 = xa - rintf(xa);                         // the start angle with its whole turns removed (v_sin_f32's domain is ±256 turns)
        const float hw = h * wa, hg = h * glt, hhg = hg * h;
        const float scale = ev ? hhg : hg;
        // angle_i = ξ₀ + c_i·h·ω + h²·(−g/2πL)·Σ_l Ā_il σ_l  (turns)
        const f32x2 A12 = C12 * f32x2{hw, hw} + f32x2{xi0, xi0};
        f32x2 P = KP1 * f32x2{s1, s1}, Q = Q1 * f32x2{s1, s1};
        // level 1: stages 3 | 2
        const float g1 = sin_turns(fmaf(h, fmaf(h, GK11 * s1, C12.x * wa), xi0));   // (Horner in h: two instructions behind h instead of three)
        const float p1 = dpp<SWAP>(g1);
        P = KPo1 * f32x2{g1, g1} + P;
        P = KPp1 * f32x2{p1, p1} + P;
        // level 2: stages 5 | 4
        const float g2 = sin_turns(fmaf(hhg, P.x, A12.y));
        const float p2 = dpp<SWAP>(g2);
        float in3 = fmaf(K3o2, g2, P.y);
        in3 = fmaf(K3p2, p2, in3);
        // level 3: the new state's slope (7) | stage 6
        const float g3 = sin_turns(fmaf(hhg, in3, xi0 + hw));
        const float p3 = dpp<SWAP>(g3);
        Q = Qo1 * f32x2{g1, g1} + Q;
        Q = Qp1 * f32x2{p1, p1} + Q;
        Q = Qo2 * f32x2{g2, g2} + Q;
        Q = Qp2 * f32x2{p2, p2} + Q;
        Q = Qo3 * f32x2{g3, g3} + Q;
        // new state and error estimate, a component per lane: ξ_new = ξ + h·ω + h²(−g/2πL)·Σ B̄σ | ω_new = ω + h(−g/2πL)·Σ bσ
        const float newc = fmaf(scale, Q.x, fmaf(evm, hw, yc));
        const float rs = fast_rcp(fmaf(max_abs(yc, newc), relt, abst));
        const float r = (scale * fmaf(Ep3, p3, Q.y)) * rs;
        // = 2·EEst², THE SAME BITS in both lanes: the square is rounded before the sum (contracted into the sum — fma(r, r, partner's r²) —
        // the two lanes' m2 differ in the last place, then their step sizes, and x and v drift apart: seen as a phase error growing with t)
        float r2 = r * r;
        asm volatile("" : "+v"(r2));   // (opaque: neither -ffp-contract nor __fmul_rn keeps the compiler from fusing the square into the sum)
        const float m2 = r2 + dpp<SWAP>(r2);
        const float mq = fmaf(0.f, newc, m2);                   // ∞·0 = NaN: a non-finite state never passes
        const bool ok = !__any(!(mq <= 2.0f));
        // PI controller on 1/q = γ·2^(β₂·log₂ EEst_old − β₁·log₂ EEst), log₂ EEst = ½·log₂ m2 − ½  (no reciprocal on the chain)
        const float lg = __builtin_amdgcn_logf(m2);
        const float iq = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(fmaf(-ca, lg, fmaf(cb, lgold, c0))), iq_lo, iq_hi);
        const float dtn = h * iq;   // (≤ dtmax where it matters: the step is clipped to what remains of the interval below)
        // (for the step after this one, if this one is accepted: f64 time arithmetic away from the step's end)
        const double t_n = t + (double)h;
        const float rem_n = (float)(tend - t_n), remc_n = rem_n * 0.99999988f;
        asm volatile("" :: "v"(dtn), "v"(remc_n));             // (the controller and the time arithmetic issue BEFORE the branch on `ok`: a scalar branch on a
                                                               //  fresh vector compare costs ≈ 48 cycles — abl/valu_lat2.hip — which now pass under them)
        if (__builtin_expect(!ok, 0)) {                        // rare: a rejected or non-finite attempt
          const bool fin = !__any(!(fabsf(newc) < __builtin_inff()));
          nrej++;
          lim--;
          if (!fin) {
            if (__any(h > dtmin)) dt = h * o.qmin;
            else { ret = LDE_RET_NONFINITE; active = false; nrej--; }
          } else {
            dt = h * fast_rcp(fminf(o.q_hi, __builtin_amdgcn_exp2f(o.beta1 * fmaf(0.5f, lg, -0.5f)) * o.inv_gamma));
            if (__any(dt < dtmin)) { ret = LDE_RET_DTMIN; active = false; }
          }
          last = __any(dt >= rem * 0.99999988f);   // (t has not moved)
          h = last ? rem : dt;
          if (!active || n >= lim) go = false;
          continue;
        }
        {   // the accepted step: leave {σ's, y_new, h} behind, publish, advance
          *reinterpret_cast<f32x4*>(rp) = f32x4{g1, g2, g3, newc};
          myrec[n * STEPF + 8 + (lane & 1)] = h;
          rp += STEPF;
          n++;
          asm volatile("" ::: "memory");                                      // the count is published AFTER the record
          __hip_atomic_store(&s_cnt[lane], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a plain LDS store: ds_write_b32)
          yc = newc;
          s1 = dpp<EVEN>(g3);   // FSAL: σ₇ is the next step's σ₁
          t = t_n;
          rem = rem_n;
          dt = dtn;
          lgold = max_f(lg, LG_MIN);
          active = !last;
          if (last || n >= lim) go = false;
          h = dtn >= remc_n ? rem_n : dtn;   // (a per-lane select — every lane carries the same numbers — not a vote: no scalar round trip in front of the next step)
          last = __any(dtn >= remc_n);
        }
      }
      nacc += n;
      LPPROF(10);
      LPPROF_VAL(30, nacc + nrej);
      if (active && nacc + nrej >= maxit) { ret = LDE_RET_MAXITERS; active = false; }
      if (ret != LDE_RET_SUCCESS && lane == 0) __hip_atomic_store(&s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_fin, active ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (REC) {
        // The step records (LDE_SENSE_DISCRETE / "step_trace") of this round, written by the stepping wave while the helpers finish their
        // dense output: lane i = the round's step i, read back from the ring — start state = the step before's end, start time = the
        // stepper's own running sum (replayed in its order: the same doubles).
        const int i0 = nacc - n;
        const bool mine = lane < n;
        const float* ri = s_rec + (size_t)(mine ? lane : 0) * STEPF;
        const float* rq = s_rec + (size_t)(mine && lane > 0 ? lane - 1 : 0) * STEPF;
        const float hi = ri[8];
        const float2 ys = lane == 0 ? (i0 == 0 ? zi : make_float2(xr * TWO_PI, vr * TWO_PI)) : make_float2(rq[3] * TWO_PI, rq[7] * TWO_PI);   // (turns → radians, as the helpers convert)
        double tacc = t_round, ti = t_round;
        const int hbits = __float_as_int(hi);
        for (int i = 0; i < n; i++) {
          if (lane == i) ti = tacc;
          tacc += (double)__int_as_float(__builtin_amdgcn_readlane(hbits, i));
        }
        if (valid && mine && i0 + lane < o.rec.cap) {
          o.rec.t[(size_t)(i0 + lane) * B + b] = ti;
          o.rec.dt[(size_t)(i0 + lane) * B + b] = (double)hi;
          reinterpret_cast<float2*>(o.rec.y)[(size_t)(i0 + lane) * B + b] = ys;
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
    LPPROF(11);
    return;
  }

  // ================= the helpers: wave 1 + hid serves every LP_NH-th step; lanes = save times =================
  // A helper touches ONLY its own records: of the records between two of its own it reads the step sizes (the time) and, from the one
  // before its own, the end state and the last sine (its record's start state and first sine). (k_pend_forward_sh's helpers walk every
  // record — two LDS round trips and the f64 time arithmetic per record and helper, ≈ 550 cycles against a stepper's ≈ 600 per step: with
  // a faster stepper they fell behind, and what they were behind by was the launch's tail.)
  const int hid = w < LP_SW ? w : w - 1;
#if LDE_PEND_PROF
  long long pq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long pq_in = __builtin_readcyclecounter();
#endif
  const double dinf = __longlong_as_double(0x7ff0000000000000LL);
  int jq = 1 + lane;                                   // the save time this lane looks for next
  double tj = jq < T ? ts_g[jq] : dinf;
  double tn = t_first;                                 // record n2 of this round starts at time tn in state ys with first sine sg1
  f32x2 ys = {zi.x, zi.y};
  float sg1 = hw_sin(zi.x, turn_anchor(zi.x));         // the stepper's σ₁ of the first step (the same instructions on the same input)
  int nrec0 = 0;                                       // records of the rounds before this one (whose turn a record is)
  auto advance = [&](int from, int to) {               // the time at the start of record `to`, given tn at the start of record `from` (the stepper's sums)
    for (int q = from; q < to; q++) {
      const float hq = myrec[(size_t)q * STEPF + 8];
      tn = (hq == (float)(tend - tn)) ? tend : tn + (double)hq;
    }
    if (to > from) {
      const float* rl = myrec + (size_t)(to - 1) * STEPF;
      ys = f32x2{rl[3], rl[7]} * TWO_PI;               // the stepper's state is in turns
      sg1 = rl[2];                                     // FSAL
    }
  };
  for (;;) {   // rounds
    int n2 = 0, fin;
    for (;;) {
      const int m = n2 + (hid + LP_NH - (nrec0 + n2) % LP_NH) % LP_NH;   // my next record of this round
      fin = __hip_atomic_load(&s_fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // read BEFORE the count: if the round is over, the count is final
      const int cnt = __hip_atomic_load(&s_cnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      if (m < cnt) {
#if LDE_PEND_PROF
        const long long pc0 = __builtin_readcyclecounter();
#endif
        advance(n2, m);
#if LDE_PEND_PROF
        asm volatile("" : "+v"(ys), "+v"(sg1));
        const long long pc1 = __builtin_readcyclecounter();
#endif
        const float* rc = myrec + (size_t)m * STEPF;
        const f32x4 qe = *reinterpret_cast<const f32x4*>(rc), qo = *reinterpret_cast<const f32x4*>(rc + 4);
        const float h = rc[8];
        const f32x2 ye = f32x2{qe[3], qo[3]} * TWO_PI;
        const double t1 = (h == (float)(tend - tn)) ? tend : tn + (double)h;   // exactly the stepper's arithmetic
        while (tj <= tn) {   // save times of the records in between: the other helpers'
          jq += 64;
          tj = jq < T ? ts_g[jq] : dinf;
        }
#if LDE_PEND_PROF
        const long long pc2 = __builtin_readcyclecounter();
        long long pc3 = pc2;
#endif
        if (__any(tj <= t1)) {
          f32x2 k0 = {0.f, 0.f}, P2 = {0.f, 0.f}, P3 = {0.f, 0.f}, P4 = {0.f, 0.f};
          float rh = 0.f;
          if (__any(tj < t1)) {   // the interpolant's polynomials from the step's seven sines
            rh = fast_rcp(h);
            const float sg[8] = {0.f, sg1, qo[0], qe[0], qo[1], qe[1], qo[2], qe[2]};   // σ₁ … σ₇
            const float hg = h * ngl;
            float px[3], pv[3];
#pragma unroll
            for (int mm = 0; mm < 3; mm++) {
              float ax = (float)TB.RA[mm][1] * sg[1], av = (float)TB.RR[1][mm] * sg[1];
#pragma unroll
              for (int l = 2; l <= 6; l++) ax = fmaf((float)TB.RA[mm][l], sg[l], ax);
#pragma unroll
              for (int j = 2; j <= 7; j++) av = fmaf((float)TB.RR[j][mm], sg[j], av);
              px[mm] = hg * ax;
              pv[mm] = ngl * av;
            }
            k0 = f32x2{ys.y, ngl * sg1};
            P2 = f32x2{px[0], pv[0]};
            P3 = f32x2{px[1], pv[1]};
            P4 = f32x2{px[2], pv[2]};
          }
#if LDE_PEND_PROF
          asm volatile("" : "+v"(P2), "+v"(P3), "+v"(P4));
          pc3 = __builtin_readcyclecounter();
#endif
          while (tj <= t1) {   // this lane's save times inside the step
            float2 out;
            if (tj >= t1) out = make_float2(ye.x, ye.y);   // the save time is the step's end
            else {
              const float th = (float)(tj - tn) * rh;
              out.x = tsit5_dense_eval<2>(th, h, ys.x, k0.x, P2.x, P3.x, P4.x);
              out.y = tsit5_dense_eval<2>(th, h, ys.y, k0.y, P2.y, P3.y, P4.y);
            }
            if (valid) z_out[(size_t)jq * B + b] = out;
            jq += 64;
            tj = jq < T ? ts_g[jq] : dinf;
          }
        }
        tn = t1;
        ys = ye;
        sg1 = qe[2];   // FSAL
        n2 = m + 1;
#if LDE_PEND_PROF
        { const long long pc4 = __builtin_readcyclecounter(); pq[0] += pc4 - pc0; pq[1] += 1; pq[7] += pc1 - pc0; pq[8] += pc2 - pc1; pq[9] += pc3 - pc2; pq[10] += pc4 - pc3; }
#endif
        continue;
      }
#if LDE_PEND_PROF
      pq[4] += 1;   // polls
#endif
      if (fin) {   // the round is over and holds no further record of mine: what is left only moves the time and the state on
        advance(n2, cnt);
        n2 = cnt;
        break;
      }
    }
    nrec0 += n2;
    if (fin == 2) break;
    __syncthreads();   // A
    __syncthreads();   // B
  }
#if LDE_PEND_PROF
  if (blockIdx.x == 0 && hid == 0 && lane == 0) {   // helper 0: entry → done
    for (int i = 0; i < 5; i++) g_pprof[i] = pq[i];
    for (int i = 7; i < 11; i++) g_pprof[i] = pq[i];
    g_pprof[6] = __builtin_readcyclecounter() - pq_in;
  }
  if (blockIdx.x == 0 && lane == 0 && hid < 3) { g_pprof[2 * (12 + hid)] = wall_clock64(); g_pprof[2 * (12 + hid) + 1] = __builtin_readcyclecounter(); }   // helper hid has stored its last save
#endif
  if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // F: every helper store has been issued and waited for
  }
}

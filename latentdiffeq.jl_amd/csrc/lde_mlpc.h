// lde_mlpc.h — k_mlpb's design (lde_mlpb.h) for networks of at most 128 hidden units on states of up to 32 rows, with TWO trajectories
// per workgroup (round 4; included by lde_mlp.hip after lde_mlpb.h).
//
// BASELINE.json configs[3] (LatentODE 32-128-128-32 relu, Tsit5, ONE coupled solve, 512 trajectories per GPU) is two trajectories per CU.
// k_mlpw gives each its own two-wave workgroup with its own 320 weight registers per lane, stages (a_l, δ_l) through HBM (359 MB per
// launch) and spends 11 k cycles per evaluation in exchange latency. Here the CU's two trajectories A and B share ONE workgroup of four
// waves (one per SIMD, 512 registers per lane) and ONE register copy of the weights:
//   * lane (r, c) of the 16 × 16 lane grid owns the block W₂[8r … 8r+7][9c … 9c+8] (72 VGPRs; column H₁ carries b₂ against a
//     constant 1, as in k_mlpb) and every product is a v_pk_fma_f32 whose two halves are the two trajectories: weight on both
//     halves (op_sel), (x_A, x_B) pairs out of LDS, where every hidden vector is stored [unit][trajectory]. W₂·h₁: 72 v_pk_fma_f32 and a
//     4-step DPP butterfly per row and trajectory; W₂ᵀ·δ₂: 72 more on the same registers, partial sums across the 16 row groups
//     through LDS — per PAIR of trajectories what k_mlpb pays per trajectory, and the barriers, LDS round trips and the step control
//     (one coupled step size, or a fixed one: both trajectories walk the same step sequence) are paid once for the two;
//   * thin products: lane (u, t) = (tid & 127, tid >> 7) does unit u of trajectory t (rows of W₁ | columns of W₃ by unit in LDS, the
//     state broadcast); H → D′ as in k_mlpw / k_mlpb with lanes = (K-segment, output), both trajectories per lane;
//   * the weight gradient stays on the CU: ring of the evaluations' vectors by stage, folded at accept time by v_mfma_f32_16x16x4_f32
//     with K = four (stage, trajectory) pairs per instruction into 26 accumulator tiles per wave (gW₂ᵀ 9 × 8 incl. the gb₂ row, gW₁
//     8 × 2, gW₃ᵀ 8 × 2) — the two trajectories add into the SAME tiles; one slab row per workgroup, k_sum_rows adds them.
// Limits: exactly three Dense layers D′ → H₁ → H₂ → D′ with D′ ≤ 32, H ≤ 128, P = 0, no analytic part, coupled or fixed-step control.

struct CDims {
  int o_wb, o_w3b, o_w13, o_b1, o_b3, o_n1, total;   // float offsets in the packed array
};

#ifndef LDE_MLPC_PF
#define LDE_MLPC_PF 1   // c4 discrete: 0.4272 → 0.4230 ms per step (abl/ab_mlp.sh)
#endif
namespace mlpc {
constexpr int A0 = 256 - 4 * 26;     // the compiler's AGPRs: a[0 : A0); the 26 tiles of a wave sit behind them (hidden: lde_mlpb.h)

constexpr int UT = 256;
constexpr int DP = 32, G1 = DP / 4, SEG = 2, GS = 16;   // state lanes, float4 groups of a state half, narrow products: K-segments and float4 groups per lane
constexpr int RB = 8, CB = 9;          // block: 8 rows × 9 columns
constexpr int HU = 144;                // units of a hidden vector in LDS (16 lanes × 9 columns; unit H₁ ≤ 128 is the constant 1)
constexpr int HV = 2 * HU;             // floats: [unit][trajectory]
constexpr int XS = 128;                // [trajectory][z (32) | λ (32)]
constexpr int SLOT = XS + 4 * HV;      // ring slot: xs | h₁ | g₂ → δ₂ | h₂ | δ₁
constexpr int W13S = 2 * DP + 4;       // floats of a unit's row in the thin-layer weight table (conflict-free ds_read_b128)
constexpr int NTL = 26;                // weight-gradient tiles of a wave: 18 of gW₂ᵀ, 4 of gW₁, 4 of gW₃ᵀ
}  // namespace mlpc

// one-time packing (set_weights): everything in the order the kernel's lanes read it
static __global__ void k_build_cpack(const float* __restrict__ Wflat, MlpDims dm, CDims cd, float* __restrict__ wp) {
  using namespace mlpc;
  const int H1 = dm.sizes[1], H2 = dm.sizes[2], Dp = dm.Dp;
  const float *W1 = Wflat + dm.w_off[0], *W2 = Wflat + dm.w_off[1], *W3 = Wflat + dm.w_off[2];   // column-major [out×in]
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cd.total; e += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (e < cd.o_w3b) {            // wb[i·CB + j][tid] = W₂(8r + i, 9c + j); column H₁ carries b₂
      const int q = e / UT, tid = e % UT, j = q % CB, i = q / CB;
      const int row = RB * (tid >> 4) + i, col = CB * (tid & 15) + j;
      if (row < H2) {
        if (col < H1) v = W2[row + H2 * col];
        else if (col == H1) v = Wflat[dm.b_off[1] + row];
      }
    } else if (e < cd.o_w13) {     // w3b[2i + h][tid] = W₃(2c + h, 8r + i): the lane's block of the output layer (f = W₃h₂ from the block product's h₂)
      const int q = (e - cd.o_w3b) / UT, tid = (e - cd.o_w3b) % UT, hh = q & 1, i = q >> 1;
      const int d = 2 * (tid & 15) + hh, uu = RB * (tid >> 4) + i;
      if (d < Dp && uu < H2) v = W3[d + Dp * uu];
    } else if (e < cd.o_b1) {      // w13[u][0 … 32) = W₁(u, ·), w13[u][32 … 64) = W₃(·, u)
      const int r = e - cd.o_w13, u = r / W13S, k = r % W13S;
      if (k < DP) { if (u < H1 && k < Dp) v = W1[u + H1 * k]; }
      else if (k < 2 * DP) { const int d = k - DP; if (u < H2 && d < Dp) v = W3[d + Dp * u]; }
    } else if (e < cd.o_b3) {
      const int u = e - cd.o_b1;
      if (u < H1) v = Wflat[dm.b_off[0] + u];
    } else if (e < cd.o_n1) {
      const int d = (e - cd.o_b3) % DP;
      if (d < Dp) v = Wflat[dm.b_off[2] + d];
    } else {                       // narrow slice of W₁ᵀ: [g][lane][4]; lane = seg·DP + d, k = (seg·GS + g)·4 + c:  vz_d = Σ_k W₁(k, d) δ₁_k
      const int r = e - cd.o_n1;
      const int c = r & 3, ln = (r >> 2) & 63, g = r >> 8;
      const int seg = ln / DP, d = ln % DP, k = (seg * GS + g) * 4 + c;
      if (d < Dp && k < H1) v = W1[k + H1 * d];
    }
    wp[e] = v;
  }
}

// DISC (with ADJ): LDE_SENSE_DISCRETE — the reverse sweep over the forward solve's step record instead of a reverse-time solve (the block
// behind the evaluation lambdas; lde_mlpd.h has the algorithm). The template's LAST bool stays ADJ (check_agprs.py reads it).
template <int SOLVER, int ACT, bool DISC, bool ADJ>
__global__ void __launch_bounds__(256, 1) k_mlpc(MlpDims dm, CDims cd, KOpts o, VArgs a) {
  using namespace mlpc;
  static_assert(ADJ || !DISC, "the discrete sweep is an adjoint");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;                // weighted stages of a step = ring slots the fold reads
  // ring slots. Continuous adjoint: the NST weighted stages + one scratch slot (initial-step probes, the FSAL stage). Discrete sweep: three
  // rotating slots for first stage points (= FSAL points of the step before) and two banks of NST − 1 stage slots — the step being reversed
  // and the step whose slopes are being rebuilt beside it (paired evaluations: see the sweep)
  constexpr int NSL = DISC ? 2 * NST + 1 : (ADJ ? NST + 1 : 1);
  constexpr bool SPEC = ADJ && SOLVER == LDE_SOLVER_TSIT5;                // an attempt never evaluates its first stage (k_mlpb: see there)
  constexpr int act = ACT;
  const int T = o.T, B = o.B, D = dm.D, Dp = dm.Dp, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = tid & 127, ut = tid >> 7;   // thin products: unit and trajectory of this lane
  const int br = tid >> 4, bc = tid & 15;   // block coordinates
  const int H1 = dm.sizes[1], H2 = dm.sizes[2];
  const int bt0 = 2 * blockIdx.x;           // the workgroup's trajectories: bt0 and bt0 + 1 (the second one may not exist: an odd batch)
  const bool two = bt0 + 1 < B;
  // ---- LDS: save times | ring | partial sums of W₂ᵀδ₂ | narrow slices | thin-layer weights by unit | cotangents
  double* s_ts = reinterpret_cast<double*>(smem);
  float* ring = reinterpret_cast<float*>(smem + (((size_t)T * 8 + 15) & ~size_t(15)));
  float* part = ring + NSL * SLOT;
  float* fpart = part + (ADJ ? 16 * HV : 0);   // f = W₃h₂: the 16 row groups' partial sums, [r][c][output 2c + h][trajectory]
  float* w13 = fpart + 16 * 16 * 4;
  f32x4* n1 = reinterpret_cast<f32x4*>(w13 + 128 * W13S);
  float* s_np = reinterpret_cast<float*>(n1 + (ADJ ? GS * 64 : 0));           // adjoint: the waves' partial outputs of the K-split narrow product, [wave][D′][trajectory]
  float* s_cot = s_np + (ADJ ? 4 * DP * 2 : 0);                               // adjoint: the two trajectories' cotangents (and saved states) by save time: [t][T][Dp] (+ the same for ẑ)
  for (int i = tid; i < T; i += UT) s_ts[i] = a.ts[i];
  for (int i = tid; i < NSL * SLOT + (ADJ ? 16 * HV : 0); i += UT) ring[i] = 0.f;
  {
    if (ADJ) {
      const f32x4* g1 = reinterpret_cast<const f32x4*>(a.wpack + cd.o_n1);
      for (int i = tid; i < GS * 64; i += UT) n1[i] = g1[i];
    }
    const f32x4* g13 = reinterpret_cast<const f32x4*>(a.wpack + cd.o_w13);
    for (int i = tid; i < 128 * W13S / 4; i += UT) reinterpret_cast<f32x4*>(w13)[i] = g13[i];
    if (ADJ && a.cot_lds) {
      const int per = T * Dp;
      for (int i = tid; i < 2 * per; i += UT) {
        const int t = i / per, q = i - t * per;
        const int bb = bt0 + t < B ? bt0 + t : bt0;
        const size_t g = (size_t)Dp * ((size_t)bb + (size_t)B * (q / Dp)) + (q % Dp);
        s_cot[i] = bt0 + t < B ? a.dz_out[g] : 0.f;
        if (!DISC && o.checkpoint) s_cot[2 * per + i] = a.z_out[g];   // (the discrete sweep takes its states from the step record)
      }
    }
  }
  __syncthreads();
  if (tid < NSL) {   // unit H₁ of every slot's h₁: the constant 1 that carries b₂ (a lane rewrites it per evaluation when H₁ < 128)
    float* h1v = ring + tid * SLOT + XS;
    h1v[2 * H1] = 1.f;
    h1v[2 * H1 + 1] = 1.f;
  }
  // ---- registers: the lane's block of W₂
  float wb[RB][CB];
  {
    const float* wp = a.wpack + cd.o_wb + tid;
#pragma unroll
    for (int i = 0; i < RB; i++)
#pragma unroll
      for (int j = 0; j < CB; j++) wb[i][j] = wp[(i * CB + j) * UT];
  }
  float w3b[RB][2];            // the lane's block of W₃: outputs 2c, 2c + 1 × its 8 units
  {
    const float* wp = a.wpack + cd.o_w3b + tid;
#pragma unroll
    for (int i = 0; i < RB; i++) { w3b[i][0] = wp[(2 * i) * UT]; w3b[i][1] = wp[(2 * i + 1) * UT]; }
  }
  const f32x4* const my13 = reinterpret_cast<const f32x4*>(w13 + u * W13S);   // this lane's row of W₁ | column of W₃ (LDS)
  const float b1 = a.wpack[cd.o_b1 + u], b3 = a.wpack[cd.o_b3 + (lane % DP)];
  // the weight gradient: accumulator-resident 16×16 tiles (C/D layout of v_mfma_f32_16x16x4_f32: column = lane & 15, row = 4·(lane >> 4) + reg)
  // — in AGPRs the compiler does not know about (a[A0 + 4n : A0 + 4n + 3], inline asm with literal numbers; see lde_mlpb.h)
  float gb1 = 0.f, gb3 = 0.f;
  if (ADJ) {
    asm volatile("" ::: "a255");   // the kernel owns all 512 registers of a lane
    mlpb::static_for<0, NTL>([&](auto nc) { mlpb::areg_zero<A0 + 4 * decltype(nc)::value>(); });
  }
  __syncthreads();

  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = fabs(tend - t0);
  unsigned gen = 0;

  // ---- state: lane i < 32 holds z_i, lane 32 + i holds λ_i (adjoint), of trajectory A in [0] and B in [1]; the other lanes stay 0
  const bool is_z = lane < Dp, is_l = ADJ && lane >= DP && lane < DP + Dp;
  const bool counted = is_z || is_l;
  const int row = is_l ? lane - DP : lane;
  float y[2] = {0.f, 0.f}, yn[2] = {0.f, 0.f}, tmp[2] = {0.f, 0.f}, scr[2] = {0.f, 0.f}, k[7][2];
#pragma unroll
  for (int s = 0; s < 7; s++) k[s][0] = k[s][1] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    if (bt0 + t < B) {
      if (!ADJ) {
        if (lane < D) y[t] = a.z0[(size_t)(bt0 + t) * D + lane];
      } else if (DISC) {
        if (is_z) y[t] = a.z_out[(size_t)(bt0 + t) * Dp + row];   // ẑ₀ (save time 0): only the failure check reads it here
      } else if (counted) {
        const size_t srcg = (size_t)Dp * ((size_t)(bt0 + t) + (size_t)B * (T - 1)) + row;
        y[t] = is_z ? a.z_out[srcg] : a.dz_out[srcg];
      }
    }
  }
  double t = ADJ ? tend : t0, dt = 0.0, tnew = 0.0;
  float h = 0.f, qold = 1e-4f, wq = 0.f, d1n = 0.f;
  int status, j = ADJ ? T - 2 : 1, last = 0, hit = 0, nfe = 0, nacc = 0, nrej = 0;
  long long iters = 0;
  {
    // a failed forward trajectory is a constant NaN block ⇒ zero gradient  [REF GOKU.jl:114] (one control for both trajectories: either one failing ends both)
    const bool bad = ADJ && __any(is_z && (!isfinite(y[0]) || !isfinite(y[1])));
    status = bad ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1);
    if (ADJ && bad) y[0] = y[1] = 0.f;
  }
  if (!ADJ && wv == 0 && lane < Dp) {   // save time 0 = ẑ₀ itself (augmented rows 0)
    a.z_out[(size_t)bt0 * Dp + lane] = y[0];
    if (two) a.z_out[(size_t)(bt0 + 1) * Dp + lane] = y[1];
  }

  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : (ADJ ? 3 : 4);
  const float dirn = ADJ ? -1.f : 1.f;
  // the error norm: coupled control over the whole batch; fixed steps need none. (A per-trajectory adaptive solve does not run here.)
  const float nnorm = (float)(ADJ ? 2 * Dp : Dp) * (coupled ? (float)(a.Bnorm > 0 ? a.Bnorm : B) : (two ? 2.f : 1.f));

  auto begin_step = [&]() -> bool {
    if (status == 0 && iters++ >= o.maxiters) status = 1 + LDE_RET_MAXITERS;
    if (status == 0) {
      if (!ADJ) {
        double d = dt;
        last = 0;
        if (t + d >= tend - 1e-12 * fabs(tend)) { d = tend - t; last = 1; }
        tnew = last ? tend : t + d;
        h = (float)d;
        wq = (float)d;
        dt = d;
      } else {
        const double dist = t - s_ts[j];
        double hmag = dt;
        hit = 0;
        if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = 1; }
        tnew = hmag;
        h = -(float)hmag;
        wq = (float)hmag;
      }
    } else {
      h = 0.f;
      wq = 0.f;
      hit = 0;
    }
    return status == 0;
  };
  // H₁ → D′ (vz = W₁ᵀδ₁) with lanes = (K-segment, output), both trajectories per lane: the register slice against the LDS vector `vec` ([unit][trajectory]);
  // every lane with lane % 32 == d gets output d of A in .x and of B in .y
  auto narrow = [&](const float* vec) -> f32x2 {
    const f32x4* hv = reinterpret_cast<const f32x4*>(vec) + (lane / DP) * GS * 2;
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < GS; g++) {
      const f32x4 wq4 = n1[g * 64 + lane], x01 = hv[2 * g], x23 = hv[2 * g + 1];
      p0 += f32x2{wq4.x, wq4.x} * x01.lo;
      p1 += f32x2{wq4.y, wq4.y} * x01.hi;
      p0 += f32x2{wq4.z, wq4.z} * x23.lo;
      p1 += f32x2{wq4.w, wq4.w} * x23.hi;
    }
    const f32x2 p = p0 + p1;
    return f32x2{swap_sum<32>(p.x), swap_sum<32>(p.y)};
  };
  // the same product split over the four waves along K (as k_mlpb's KSPLIT): this wave's four of the sixteen read groups — 12 LDS reads in
  // flight together instead of 48 that the register pressure serialises — and the partial outputs meet in s_np behind a barrier
  auto narrow_part = [&](const float* vec) -> f32x2 {
    constexpr int GP = GS / 4;
    const f32x4* hv = reinterpret_cast<const f32x4*>(vec) + (lane / DP) * GS * 2;
    f32x4 wq4[GP], x01[GP], x23[GP];
#pragma unroll
    for (int q = 0; q < GP; q++) {
      const int g = wv * GP + q;
      wq4[q] = n1[g * 64 + lane];
      x01[q] = hv[2 * g];
      x23[q] = hv[2 * g + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < GP; q++) {
      p0 += f32x2{wq4[q].x, wq4[q].x} * x01[q].lo;
      p1 += f32x2{wq4[q].y, wq4[q].y} * x01[q].hi;
      p0 += f32x2{wq4[q].z, wq4[q].z} * x23[q].lo;
      p1 += f32x2{wq4[q].w, wq4[q].w} * x23[q].hi;
    }
    const f32x2 p = p0 + p1;
    return f32x2{swap_sum<32>(p.x), swap_sum<32>(p.y)};
  };

  // one evaluation of the (augmented) right-hand side for both trajectories: src → dst; its vectors stay in ring slot `slot`
  // vj (wave-uniform; only the discrete sweep passes false): false = the forward half alone — f, h₁, h₂ — for the evaluations that rebuild slopes.
  // vslot ≥ 0 (the discrete sweep's PAIRED evaluation): the vector-Jacobian half belongs to ANOTHER point — the one whose forward half left
  // z, h₁, h₂ in ring slot `vslot` earlier — while the forward half evaluates `src`'s z lanes into `slot`: two independent evaluations in one
  // evaluation's phases. (vslot < 0: both halves at the same point, the activations taken from registers — the continuous adjoint's form.)
  auto eval = [&](const float (&src)[2], int slot, float (&dst)[2], bool vj = true, int vslot = -1) {
    PROF_T2(e0);
    const bool paired = DISC && vslot >= 0;
    float* xs = ring + slot * SLOT;
    float* xv = paired ? ring + vslot * SLOT : xs;   // the slot of the vector-Jacobian half: its λ, W₃ᵀλ → δ₂, δ₁ (and, paired, its saved h₁, h₂)
    float *h1v = xs + XS, *h2v = h1v + 2 * HV;
    float *d2v = xv + XS + HV, *d1v = xv + XS + 3 * HV;
    if (paired) {   // z lanes feed the forward half's slot, λ lanes the other one's
      (lane < DP ? xs : xv)[lane] = src[0];
      (lane < DP ? xs : xv)[64 + lane] = src[1];
    } else {
      xs[lane] = src[0];          // (every wave stores the same values)
      xs[64 + lane] = src[1];
    }
    asm volatile("" ::: "memory");   // same wave, in-order LDS: the broadcast reads below see the write (no barrier needed)
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xs + 64 * ut);
    const f32x4* x4v = reinterpret_cast<const f32x4*>(xv + 64 * ut);
    float h1;
    {
      // (the sixteen reads of a thin product are issued before its first multiplication waits for one — the compiler's own order was eight,
      //  then two at a time: five LDS round trips per product)
      f32x4 xv[G1], wv4[G1];
#pragma unroll
      for (int g = 0; g < G1; g++) { xv[g] = x4[g]; wv4[g] = my13[g]; }
      __builtin_amdgcn_sched_barrier(0);
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G1; g++) {
        c01 += wv4[g].lo * xv[g].lo;
        c23 += wv4[g].hi * xv[g].hi;
      }
      const float a1 = b1 + ((c01.x + c01.y) + (c23.x + c23.y));
      h1 = u == H1 ? 1.f : act_fn(act, a1);   // unit H₁ (when < 128): the constant that carries b₂ (rows beyond: zero weights and bias ⇒ act(0) = 0)
    }
    const float h1b = paired ? xv[XS + 2 * u + ut] : h1;   // h₁ of the vector-Jacobian half's point (read before this lane's own word is rewritten: X and Y may be one slot)
    h1v[2 * u + ut] = h1;
    if (ADJ && vj) {
      f32x4 xv[G1], wv4[G1];
#pragma unroll
      for (int g = 0; g < G1; g++) { xv[g] = x4v[G1 + g]; wv4[g] = my13[G1 + g]; }   // λ
      __builtin_amdgcn_sched_barrier(0);
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G1; g++) {
        c01 += wv4[g].lo * xv[g].lo;
        c23 += wv4[g].hi * xv[g].hi;
      }
      d2v[2 * u + ut] = (c01.x + c01.y) + (c23.x + c23.y);   // (W₃ᵀλ)_u; becomes δ₂ below
    }
    __syncthreads();
    PROF_T2(e1);
    // ---- the block products: every v_pk_fma_f32 is (trajectory A, trajectory B) against the weight on both halves
    f32x2 h2[RB];
    {
      const f32x2* hc = reinterpret_cast<const f32x2*>(h1v) + CB * bc;
      f32x2 hp[CB];
#pragma unroll
      for (int jj = 0; jj < CB; jj++) hp[jj] = hc[jj];
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = f32x2{wb[i][0], wb[i][0]} * hp[0];
#pragma unroll
      for (int jj = 1; jj < CB; jj++)
#pragma unroll
        for (int i = 0; i < RB; i++) h2[i] += f32x2{wb[i][jj], wb[i][jj]} * hp[jj];
      // the 16 lanes of a row add their partial sums: four butterfly steps, all 16 values per step
#pragma unroll
      for (int i = 0; i < RB; i++) { h2[i].x = dpp_xadd<0xB1>(h2[i].x); h2[i].y = dpp_xadd<0xB1>(h2[i].y); }
#pragma unroll
      for (int i = 0; i < RB; i++) { h2[i].x = dpp_xadd<0x4E>(h2[i].x); h2[i].y = dpp_xadd<0x4E>(h2[i].y); }
#pragma unroll
      for (int i = 0; i < RB; i++) { h2[i].x = dpp_xadd<0x141>(h2[i].x); h2[i].y = dpp_xadd<0x141>(h2[i].y); }
#pragma unroll
      for (int i = 0; i < RB; i++) {
        h2[i].x = act_fn(act, dpp_xadd<0x140>(h2[i].x));
        h2[i].y = act_fn(act, dpp_xadd<0x140>(h2[i].y));
      }
    }
    {   // f = W₃h₂, this row group's share: outputs 2c, 2c + 1 over the lane's 8 units; the 16 shares meet behind the barrier
      f32x2 f0 = f32x2{w3b[0][0], w3b[0][0]} * h2[0], f1 = f32x2{w3b[0][1], w3b[0][1]} * h2[0];
#pragma unroll
      for (int i = 1; i < RB; i++) {
        f0 += f32x2{w3b[i][0], w3b[i][0]} * h2[i];
        f1 += f32x2{w3b[i][1], w3b[i][1]} * h2[i];
      }
      reinterpret_cast<f32x4*>(fpart)[br * 16 + bc] = f32x4{f0.x, f0.y, f1.x, f1.y};
    }
    if (ADJ && vj) {
      f32x2 d2[RB];
      const f32x2* dv = reinterpret_cast<const f32x2*>(d2v) + RB * br;
      const f32x2* hb = reinterpret_cast<const f32x2*>(xv + XS + 2 * HV) + RB * br;   // (paired) h₂ of the other point, as its forward half left it
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const f32x2 g2 = dv[i];
        const f32x2 hh = paired ? hb[i] : h2[i];
        d2[i] = f32x2{g2.x * act_grad(act, hh.x), g2.y * act_grad(act, hh.y)};
      }
      if (bc == 0) {   // one lane of the row leaves h₂ and δ₂ for the thin products and the fold
        f32x2* hw = reinterpret_cast<f32x2*>(h2v) + RB * br;
        f32x2* dw = reinterpret_cast<f32x2*>(d2v) + RB * br;
#pragma unroll
        for (int i = 0; i < RB; i++) { hw[i] = h2[i]; dw[i] = d2[i]; }
      }
      f32x2 gp[CB];
#pragma unroll
      for (int jj = 0; jj < CB; jj++) gp[jj] = f32x2{wb[0][jj], wb[0][jj]} * d2[0];
#pragma unroll
      for (int i = 1; i < RB; i++)
#pragma unroll
        for (int jj = 0; jj < CB; jj++) gp[jj] += f32x2{wb[i][jj], wb[i][jj]} * d2[i];
      f32x2* pp = reinterpret_cast<f32x2*>(part) + br * HU + CB * bc;
#pragma unroll
      for (int jj = 0; jj < CB; jj++) pp[jj] = gp[jj];
    } else if (DISC && bc == 0) {   // the forward half alone (the sweep's prologue): h₂ stays in the slot for the paired evaluation that follows
      f32x2* hw = reinterpret_cast<f32x2*>(h2v) + RB * br;
#pragma unroll
      for (int i = 0; i < RB; i++) hw[i] = h2[i];
    }
    __syncthreads();
    PROF_T2(e2);
    {
      f32x2 f = {b3, b3};
      const f32x2* fq = reinterpret_cast<const f32x2*>(fpart) + (lane & 31);   // output d = lane: [r][d >> 1][d & 1] pairs
#pragma unroll
      for (int r = 0; r < 16; r++) f += fq[r * 32];
      dst[0] = is_z ? f.x : 0.f;
      dst[1] = is_z ? f.y : 0.f;
    }
    PROF_ADD2(3, e0, e1);
    PROF_ADD2(4, e1, e2);
    if (ADJ && vj) {
      float g1 = 0.f;
      {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; r++) p[r] = part[(r * HU + u) * 2 + ut];
#pragma unroll
        for (int r = 0; r < 16; r++) g1 += p[r];
      }
      const float d1 = u < H1 ? g1 * act_grad(act, h1b) : 0.f;
      d1v[2 * u + ut] = d1;
      __syncthreads();
      const f32x2 pvz = narrow_part(d1v);
      if (lane < DP) reinterpret_cast<f32x2*>(s_np)[wv * DP + lane] = pvz;
      __syncthreads();
      const f32x2* qz = reinterpret_cast<const f32x2*>(s_np) + (lane % DP);
      const f32x2 vz = (qz[0] + qz[DP]) + (qz[2 * DP] + qz[3 * DP]);   // every lane with lane % 32 == d holds vz_d of A and B (the same order in every wave)
      if (is_l) { dst[0] = -vz.x; dst[1] = -vz.y; }
      PROF_T2(e3);
      PROF_ADD2(5, e2, e3);
    }
    PROF_T2(e5);
    PROF_ADD2(1, e0, e5);
    PROF_ADD2(20, e5 - 1, e5);
  };

  // the accepted step's share of the quadrature gW = Σ_s |h| b_s (∂f/∂W)ᵀλ of BOTH trajectories, from the ring. K slot q = 4g + (lane >> 4)
  // ↔ (stage e, trajectory) = (q >> 1, q & 1). Tiles of wave w (operand addresses = a per-lane base + compile-time offsets):
  //   n = 2·ti + m, ti < 9, m < 2 : gW₂ᵀ tile (ti, tj = 4m + w)     n = 18 + 2m + tk : gW₁ tile (4m + w, tk)     n = 22 + 2m + tk : gW₃ᵀ tile (4m + w, tk)
  int vbank = 3, fsx = 0;   // (discrete sweep) first slot of the bank of the step being reversed (stage point i ≥ 1 at vbank + i − 1), and the FSAL point's slot
  auto fslot = [&](int e) -> int { return DISC ? (e == 0 ? fsx : vbank + (NST - e) - 1) : e; };   // ring slot of the fold's evaluation e
  auto fold = [&](int nvalid) {   // nvalid (wave-uniform): ring slots [0, nvalid) count
    PROF_T2(f0);
    const int l15 = lane & 15, e4 = lane >> 4;
#pragma unroll
    for (int g = 0; g < (2 * NST + 3) / 4; g++) {
      const int q = 4 * g + e4, e = q >> 1, tt = q & 1;
      float bs;
      if (DISC) bs = 1.f;   // (the scale h·b_i is inside k̄)
      else if (SOLVER == LDE_SOLVER_TSIT5) bs = e == 0 ? ts5::A[6][0] : e == 1 ? ts5::A[6][1] : e == 2 ? ts5::A[6][2] : e == 3 ? ts5::A[6][3] : e == 4 ? ts5::A[6][4] : ts5::A[6][5];
      else bs = (e == 0 || e == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
      const bool ev = e < nvalid;
      const float wsc = DISC ? 1.f : wq * bs;
      const float* sl = ring + fslot(ev ? e : 0) * SLOT;
      const float* pv = sl + XS + 2 * l15 + tt;           // + 32·tile: element (16·tile + l15) of trajectory tt of the slot's first vector
      const float* pw = pv + 32 * wv;                     // … of the tiles 4m + w
      float bm[2];
      const float am = ev ? 1.f : 0.f, wm = ev ? wsc : 0.f;   // (reads unconditional — the slot index is clamped — and masked by a factor)
#pragma unroll
      for (int m = 0; m < 2; m++) bm[m] = pw[HV + 128 * m] * wm;                 // δ₂ tiles 4m + w, scaled
      // every operand of the group is read before the first MFMA (the asm statements keep their order: a read between them would be
      // waited for in front of the next one)
      float av[9], a1[2], a3[2], zv[2], lv[2];
#pragma unroll
      for (int ti = 0; ti < 9; ti++) av[ti] = pv[32 * ti] * am;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        a1[m] = pw[3 * HV + 128 * m] * am;
        a3[m] = pw[2 * HV + 128 * m] * am;
      }
#pragma unroll
      for (int tk = 0; tk < 2; tk++) {   // (unconditional reads, clamped rows: a read under a lane condition is waited for where it stands)
        const int rz = 16 * tk + l15 < Dp ? 16 * tk + l15 : 0;
        zv[tk] = sl[64 * tt + rz];
        lv[tk] = sl[64 * tt + DP + rz];
      }
      float bz[2], bl[2];
#pragma unroll
      for (int tk = 0; tk < 2; tk++) {
        const bool in = ev && 16 * tk + l15 < Dp;
        bz[tk] = in ? zv[tk] * wsc : 0.f;         // z rows of the tile, scaled
        bl[tk] = in ? lv[tk] * wsc : 0.f;         // λ rows, scaled
      }
      __builtin_amdgcn_sched_barrier(0);
      mlpb::static_for<0, 9>([&](auto tic) {   // gW₂ᵀ[i][o] += Σ h₁[i] · (w δ₂)[o]   (row H₁: gb₂)
        constexpr int ti = decltype(tic)::value;
        mlpb::areg_mfma<A0 + 4 * (2 * ti + 0)>(av[ti], bm[0]);
        mlpb::areg_mfma<A0 + 4 * (2 * ti + 1)>(av[ti], bm[1]);
      });
      mlpb::static_for<0, 2>([&](auto mc) {    // gW₁[u][k] += Σ (w δ₁)[u] · z[k];  gW₃ᵀ[u][d] += Σ h₂[u] · (w λ)[d]
        constexpr int m = decltype(mc)::value;
        mlpb::areg_mfma<A0 + 4 * (18 + 2 * m + 0)>(a1[m], bz[0]);
        mlpb::areg_mfma<A0 + 4 * (18 + 2 * m + 1)>(a1[m], bz[1]);
        mlpb::areg_mfma<A0 + 4 * (22 + 2 * m + 0)>(a3[m], bl[0]);
        mlpb::areg_mfma<A0 + 4 * (22 + 2 * m + 1)>(a3[m], bl[1]);
      });
    }
    // thin biases: gb₁[u] += Σ w (δ₁_A + δ₁_B)[u] (lanes tid < 128), gb₃[d] += Σ w (λ_A + λ_B)[d] (the λ lanes); gb₂ is row H₁ of the gW₂ᵀ tiles
#pragma unroll
    for (int e = 0; e < NST; e++) {
      const float bs = SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[6][e] : ((e == 0 || e == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f));
      const float wb_ = DISC ? (e < nvalid ? 1.f : 0.f) : wq * bs;
      const float* sl = ring + fslot(e) * SLOT;
      const f32x2 dd = reinterpret_cast<const f32x2*>(sl + XS + 3 * HV)[u];
      gb1 += wb_ * (dd.x + dd.y);
      gb3 += wb_ * (sl[lane] + sl[64 + lane]);
    }
    __syncthreads();   // the next attempt overwrites the ring: every wave has read it
    PROF_T2(f1);
    PROF_ADD2(6, f0, f1);
  };

  // this wave's sum of the attempt's scaled squared errors over both trajectories (every wave holds the whole state)
  auto err_sum = [&]() -> float {
    float r2 = 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      if (o.adaptive && counted && (tt == 0 || two)) {
        float er = ts5::BT[0] * k[0][tt];
#pragma unroll
        for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * k[jj][tt];
        er *= h;
        const float sk = o.abstol + fmaxf(fabsf(y[tt]), fabsf(yn[tt])) * o.reltol;
        const float r = er * fast_rcp(sk);
        r2 += r * r;
      }
      if (!isfinite(yn[tt])) r2 = __int_as_float(0x7fc00000);   // a non-finite state poisons the sum
    }
    return wave_sum64(r2);
  };
  // the adjoint's state behind save time j: λ += Δ_j, z reset to the saved ẑ(t_j) (checkpointing)
  auto jumped = [&](float (&v)[2]) {
    if (counted) {
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        if (a.cot_lds) {
          if (is_l) v[tt] += s_cot[(tt * T + j) * Dp + row];
          else if (o.checkpoint) v[tt] = s_cot[((2 + tt) * T + j) * Dp + row];
        } else if (bt0 + tt < B) {
          const size_t srcg = (size_t)Dp * ((size_t)(bt0 + tt) + (size_t)B * j) + row;
          if (is_l) v[tt] += a.dz_out[srcg];
          else if (o.checkpoint) v[tt] = a.z_out[srcg];
        }
      }
    }
  };
  if constexpr (DISC) {
    // ---- LDE_SENSE_DISCRETE: the recorded steps (t_n, dt_n, y_n), last to first. z lanes carry the stage points and slopes (y = y_n,
    //      yn = y_{n+1}, k[i] = k_{i+1}); λ lanes carry the cotangents (y = ȳ_{n+1}, yn = ȳ_n, k[i] = k̄_{i+1}, scr = the k̄₁ that travels on)
    constexpr int S = NST;
    constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                                {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
    auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
    const StepRec R = o.rec;
    const int seq = coupled ? 0 : bt0;   // (both trajectories walk ONE sequence: a coupled solve's, or the same fixed steps)
    const int ns = R.n[seq];
    if (status == 0 && (ns < 1 || ns > R.cap)) status = 1 + LDE_RET_MAXITERS;   // no usable record: NaN gradient, never a truncated sweep
    y[0] = y[1] = 0.f;                  // (ẑ₀ has served the failure check; the last evaluation reloads it)
    // stage point i of the step from y_n and the slopes (z lanes; i == S: y_{n+1})
    auto point = [&](int i, float hh, int tt) -> float {
      float zv = y[tt];
      if (SOLVER == LDE_SOLVER_RK4 && i == S) zv = y[tt] + (hh * (1.0f / 6.0f)) * (k[0][tt] + 2.0f * (k[1][tt] + k[2][tt]) + k[3][tt]);
      else {
#define DSTAGE(S_)                                                                   \
  case S_: {                                                                         \
    float accv = A(S_, 0) * k[0][tt];                                                \
    _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += A(S_, jj) * k[jj][tt]; \
    zv = y[tt] + hh * accv;                                                          \
  } break;
        switch (i) {
          DSTAGE(1) DSTAGE(2) DSTAGE(3) DSTAGE(4)
          default: break;
        }
        if constexpr (S == 6) {
          switch (i) {
            DSTAGE(5) DSTAGE(6)
            default: break;
          }
        }
#undef DSTAGE
      }
      return zv;
    };
    if (status == 0) {
      // PAIRED evaluations: while step m is reversed, the slopes of step m − 1 are rebuilt — the two chains are independent, so the forward
      // half of one evaluation's phases rebuilds a slope of step m − 1 and its vector-Jacobian half pulls a k̄ of step m through a stage point
      // whose z, h₁, h₂ the forward half of an EARLIER evaluation left in the ring: S evaluation-times per step instead of 2S.
      // Ring slots: three rotating ones — sx: the FSAL point y_{m+1} of the step being reversed (= the first stage point of step m + 1),
      // sy: the first stage point of step m (the next sx), sz: where the forward halves put the first stage point of step m − 1 — and two
      // banks of S − 1 slots (stage i ≥ 2 at bank + i − 2): vbank of step m, fbank of step m − 1.
      int sx = 0, sy = 1, sz = 2, fbank = 3;
      vbank = S + 2;
      j = T - 1;
      double tnext = tend;
      {   // prologue: the slopes of the last step (forward halves alone: first stage point → sy, the others → vbank), then f at its end
          // state y_ns — the FSAL point of that step — into sx
        const float hz = (float)R.dt[(size_t)(ns - 1) * R.nseq + seq];
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
          if (is_z) y[tt] = (tt == 0 || two) ? R.y[((size_t)(ns - 1) * B + (bt0 + tt)) * Dp + row] : 0.f;
#pragma unroll 1
        for (int i = 0; i <= S; i++) {
          float src[2], dst[2];
#pragma unroll
          for (int tt = 0; tt < 2; tt++) src[tt] = is_z ? point(i, hz, tt) : 0.f;
          eval(src, i == 0 ? sy : (i == S ? sx : vbank + i - 1), dst, false);
#pragma unroll
          for (int q = 0; q < S; q++)
            if (q == i && is_z) { k[q][0] = dst[0]; k[q][1] = dst[1]; }
        }
        nfe += S + 1;
      }
#if LDE_MLPC_PF   // the record of the step after next is requested while this step's evaluations run (a load at the head of a step is covered by nothing)
      double t_pf = R.t[(size_t)(ns - 1) * R.nseq + seq], dt_pf = R.dt[(size_t)(ns - 1) * R.nseq + seq];
      double dtz_pf = ns > 1 ? R.dt[(size_t)(ns - 2) * R.nseq + seq] : 0.0;
      float y_pf[2];
#pragma unroll
      for (int tt = 0; tt < 2; tt++) y_pf[tt] = (is_z && ns > 1 && (tt == 0 || two)) ? R.y[((size_t)(ns - 2) * B + (bt0 + tt)) * Dp + row] : 0.f;
#endif
#pragma unroll 1
      for (int sidx = ns - 1; sidx >= 0; sidx--) {
#if LDE_MLPC_PF
        const double ts_n = t_pf, dts = dt_pf;
        const float hh = (float)dts;                                                                   // the step being reversed (λ lanes)
        const float hz = sidx > 0 ? (float)dtz_pf : 0.f;                                               // the step being rebuilt (z lanes)
        const float y_rec[2] = {y_pf[0], y_pf[1]};
        {
          const size_t sp = sidx > 0 ? sidx - 1 : 0, sq = sidx > 1 ? sidx - 2 : 0;
          t_pf = R.t[sp * R.nseq + seq];
          dt_pf = dtz_pf;
          dtz_pf = R.dt[sq * R.nseq + seq];
#pragma unroll
          for (int tt = 0; tt < 2; tt++)
            if (is_z && (tt == 0 || two)) y_pf[tt] = R.y[(sq * B + (bt0 + tt)) * Dp + row];
        }
#else
        const double ts_n = R.t[(size_t)sidx * R.nseq + seq], dts = R.dt[(size_t)sidx * R.nseq + seq];
        const float hh = (float)dts;                                                                   // the step being reversed (λ lanes)
        const float hz = sidx > 0 ? (float)R.dt[(size_t)(sidx - 1) * R.nseq + seq] : 0.f;             // the step being rebuilt (z lanes)
#endif
        const bool lastst = sidx == ns - 1;
        const double tnw = tnext;
        tnext = ts_n;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          if (is_z) {
#if LDE_MLPC_PF
            if (sidx > 0) y[tt] = y_rec[tt];
#else
            if (sidx > 0) y[tt] = (tt == 0 || two) ? R.y[((size_t)(sidx - 1) * B + (bt0 + tt)) * Dp + row] : 0.f;   // (the first step rebuilds nothing: a dummy forward half)
#endif
          } else {
            yn[tt] = 0.f;
#pragma unroll
            for (int q = 0; q < S; q++) k[q][tt] = 0.f;
            k[S][tt] = is_l ? scr[tt] : 0.f;
          }
        }
        // the save times inside the step (t_n, t_{n+1}]
        while (j >= 1 && sgpr_d(s_ts[j]) > ts_n) {   // (wave-uniform: kept scalar)
          const double tj = sgpr_d(s_ts[j]);
          const bool at_end = tj >= tnw || (j == T - 1 && lastst);
          const float th = at_end ? 2.0f : (float)(tj - ts_n) * fast_rcp(hh);
          if (is_l) {
            float bw[7];
            if (SOLVER == LDE_SOLVER_TSIT5) tsit5_interp_weights(th, bw);
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
              float dj = 0.f;
              if (a.cot_lds) dj = s_cot[(tt * T + j) * Dp + row];
              else if (bt0 + tt < B) dj = a.dz_out[(size_t)Dp * ((size_t)(bt0 + tt) + (size_t)B * j) + row];
              if (at_end) y[tt] += dj;
              else if (SOLVER == LDE_SOLVER_TSIT5) {
                yn[tt] += dj;
#pragma unroll
                for (int q = 0; q < 7; q++) k[q][tt] += (hh * bw[q]) * dj;
              } else {
                const float om = 1.0f - th;
                const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
                const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
                yn[tt] += h00 * dj;
                k[0][tt] += (h10 * hh) * dj;
                y[tt] += h01 * dj;
                k[S][tt] += (h11 * hh) * dj;
              }
            }
          }
          j--;
        }
        // S paired evaluations: vector-Jacobian half at y_{n+1} (slot sx), then at g_S … g_2 (vbank); forward half at the stage points 1 … S
        // of step n − 1 (→ sz, fbank)
#pragma unroll 1
        for (int kk = 0; kk < S; kk++) {
          const int i = S - kk;   // the stage whose k̄ is pulled back (i == S: the FSAL slope at y_{n+1})
          float src[2], dst[2];
#pragma unroll
          for (int tt = 0; tt < 2; tt++) {
            float kb = 0.f;
#pragma unroll
            for (int q = 0; q <= S; q++) kb = q == i ? k[q][tt] : kb;
            src[tt] = is_z ? point(kk, hz, tt) : kb;
          }
          eval(src, kk == 0 ? sz : fbank + kk - 1, dst, true, kk == 0 ? sx : vbank + i - 1);
          if (is_z) {
#pragma unroll
            for (int q = 0; q < S; q++)
              if (q == kk) { k[q][0] = dst[0]; k[q][1] = dst[1]; }
          }
          if (is_l) {
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
              const float v = -dst[tt];
              if (i == S) {
                y[tt] += v;
#pragma unroll
                for (int q = 0; q < S; q++) k[q][tt] += (hh * A(S, q)) * y[tt];
                yn[tt] += y[tt];
              } else {
                yn[tt] += v;
#pragma unroll
                for (int q = 0; q < S - 1; q++) {
                  float aq = 0.f;
#pragma unroll
                  for (int ii = 1; ii < S; ii++) aq = ii == i ? A(ii, q) : aq;
                  if (q < i) k[q][tt] += (hh * aq) * v;
                }
              }
            }
          }
        }
        fsx = sx;
        fold(S);
        if (is_l) {
#pragma unroll
          for (int tt = 0; tt < 2; tt++) {
            scr[tt] = k[0][tt];
            y[tt] = yn[tt];
          }
        }
        {   // the rebuilt step becomes the one to reverse: its first stage point is the next FSAL point
          const int t3 = sx;
          sx = sy; sy = sz; sz = t3;
          const int tb = vbank;
          vbank = fbank; fbank = tb;
        }
        nfe += sidx > 0 ? 2 * S : S;   // evaluations as the other discrete kernels count them: a paired evaluation = the vector-Jacobian half of this
                                       // step's stage + the forward half of a stage of step sidx − 1
        nacc++;
      }
      {   // k_1 of the first step = f(y_0): its forward half ran as the first stage point of step 0 (slot sx by now)
        float src[2], dst[2];
#pragma unroll
        for (int tt = 0; tt < 2; tt++) src[tt] = is_z ? y[tt] : (is_l ? scr[tt] : 0.f);
        eval(src, sz, dst, true, sx);
        fsx = sx;
        fold(1);
        if (is_l) { y[0] -= dst[0]; y[1] -= dst[1]; }
        nfe++;
      }
      status = 1;
    }
    if (is_l) {   // save time 0 is ẑ₀ itself
#pragma unroll
      for (int tt = 0; tt < 2; tt++)
        if (bt0 + tt < B) y[tt] += a.dz_out[(size_t)(bt0 + tt) * Dp + row];
    }
  }
  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  int phase = (ADJ && !auto_dt) ? PH_STAGE : PH_K0, s = 0;
  bool running = !DISC && T > 1 && status == 0;
  if (ADJ && running && !auto_dt) {
    dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    running = begin_step();
  }
  // the step control is wave-uniform (every lane computes the same values): it is kept in scalar registers across the evaluations —
  // made scalar wherever it has been through vector arithmetic (the step-end block, the two probes of the initial step size), not once
  // per evaluation (21 v_readfirstlane)
  auto scalarise = [&]() {
    s = __builtin_amdgcn_readfirstlane(s);
    phase = __builtin_amdgcn_readfirstlane(phase);
    status = __builtin_amdgcn_readfirstlane(status);
    j = __builtin_amdgcn_readfirstlane(j); last = __builtin_amdgcn_readfirstlane(last); hit = __builtin_amdgcn_readfirstlane(hit);
    nfe = __builtin_amdgcn_readfirstlane(nfe); nacc = __builtin_amdgcn_readfirstlane(nacc); nrej = __builtin_amdgcn_readfirstlane(nrej);
    iters = sgpr_ll(iters);
    t = sgpr_d(t); dt = sgpr_d(dt); tnew = sgpr_d(tnew);
    h = sgpr_f(h); qold = sgpr_f(qold); wq = sgpr_f(wq); d1n = sgpr_f(d1n);
  };
  // Dense output of the attempted step at the save times it covers, written BEFORE the step is known to be accepted — between the error
  // sum's publication and its collection, where the workgroup would wait for the grid anyway. A rejected attempt's values are overwritten:
  // every save time is covered again by a later accepted step (j advances only on acceptance), or the solve fails and the NaN block follows.
  auto dense_output = [&](int jj) -> int {   // returns the index of the first save time behind the step
    while (jj < T && s_ts[jj] <= tnew) {
      const double tj = s_ts[jj];
      const float th = (tj >= tnew || (jj == T - 1 && last)) ? 2.0f : (float)(tj - t) * fast_rcp(wq);
      float ov[2];
      if (th > 1.5f) { ov[0] = yn[0]; ov[1] = yn[1]; }
      else if (SOLVER == LDE_SOLVER_TSIT5) {
        float bw[7];
        tsit5_interp_weights(th, bw);
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          float acc = bw[0] * k[0][tt];
#pragma unroll
          for (int q = 1; q < 7; q++) acc += bw[q] * k[q][tt];
          ov[tt] = y[tt] + wq * acc;
        }
      } else {
        const float om = 1.0f - th;
        const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
        const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
#pragma unroll
        for (int tt = 0; tt < 2; tt++) ov[tt] = h00 * y[tt] + (h10 * wq) * k[0][tt] + h01 * yn[tt] + (h11 * wq) * k[4][tt];
      }
      if (wv == 0 && lane < Dp) {
        a.z_out[(size_t)Dp * ((size_t)bt0 + (size_t)B * jj) + lane] = ov[0];
        if (two) a.z_out[(size_t)Dp * ((size_t)(bt0 + 1) + (size_t)B * jj) + lane] = ov[1];
      }
      jj++;
    }
    return jj;
  };
  while (__builtin_amdgcn_readfirstlane((int)running)) {
    // inner loop: the evaluations of one unit of work (the two probes of the initial step size, or the stages of one step attempt) — the
    // weight-gradient tiles are touched only outside it, in the step-end block below
    bool step_end = false;
    float s2 = 0.f, s2b = 0.f;
    int jd = j;
    scalarise();
    PROF_T(a0);
    do {
      PROF_T2(l0);
#if LDE_PROF >= 2
      struct ProfEnd { long long t0; __device__ ~ProfEnd() { PROF_T(t1); PROF_ADD(11, t0, t1); } } prof_end{l0};
#endif
      float src[2];
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        float sv = phase == PH_INIT1 ? tmp[tt] : y[tt];
        if (phase == PH_STAGE) {
          if (SOLVER == LDE_SOLVER_TSIT5) {
            if (s > 0) {
#define CSTAGE(S_)                                                                       \
  case S_: {                                                                             \
    float accv = ts5::A[S_][0] * k[0][tt];                                               \
    _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += ts5::A[S_][jj] * k[jj][tt]; \
    sv = y[tt] + h * accv;                                                               \
  } break;
              switch (s) {
                CSTAGE(1) CSTAGE(2) CSTAGE(3) CSTAGE(4) CSTAGE(5) CSTAGE(6)
                default: break;
              }
#undef CSTAGE
              if (s == 6) yn[tt] = sv;
            }
          } else if (ADJ || s < 4) {
            if (s > 0) {
              const float cs = (s == 3 ? 1.0f : 0.5f) * h;
              sv = y[tt] + cs * (s == 1 ? k[0][tt] : (s == 2 ? k[1][tt] : k[2][tt]));
            }
          } else {
            const float h6 = h * (1.0f / 6.0f);
            yn[tt] = y[tt] + h6 * (k[0][tt] + 2.0f * (k[1][tt] + k[2][tt]) + k[3][tt]);
            sv = yn[tt];
          }
          if (SPEC && s == 7) sv = tmp[tt];   // the speculative first evaluation of the next attempt: the state behind the jump
        }
        src[tt] = sv;
      }
      float dst[2];
      eval(src, ADJ ? ((phase == PH_STAGE && s < NST) ? s : NST) : 0, dst);
      {
        const int ks = phase == PH_K0 ? 0 : (phase == PH_INIT1 ? 1 : s);
#pragma unroll
        for (int q = 0; q < 7; q++)
          if (q == ks) { k[q][0] = dst[0]; k[q][1] = dst[1]; }
        if (SPEC && ks == 7) { scr[0] = dst[0]; scr[1] = dst[1]; }
      }
      if (status == 0) nfe++;

      if (phase == PH_K0 && !(ADJ || auto_dt)) {
        dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
        phase = PH_STAGE;
        s = 1;
        running = begin_step();
        scalarise();
      } else if (phase == PH_K0) {
        // Hairer–Nørsett–Wanner, part 1
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          const float sk = fast_rcp(o.abstol + fabsf(y[tt]) * o.reltol);
          scr[tt] = sk;
          const float a0 = y[tt] * sk, a1v = k[0][tt] * sk;
          if (counted && (tt == 0 || two)) { p0 += a0 * a0; p1 += a1v * a1v; }   // (an odd batch's missing trajectory is not part of the norm)
        }
        float v0 = wave_sum64(p0), v1 = wave_sum64(p1);
        if (coupled) {
          if (status != 0) v0 = v1 = 0.f;
          w_grid_sum<true>(a.gs, gen, a.epoch, v0, v1);
        }
        const float d0 = sqrtf(v0 / nnorm);
        d1n = sqrtf(v1 / nnorm);
        double dt0 = (d0 < 1e-5f || d1n < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1n));
        if (dt0 > dtmax) dt0 = dtmax;
        dt = dt0;
        h = status == 0 ? dirn * (float)dt0 : 0.f;
        tmp[0] = y[0] + h * k[0][0];
        tmp[1] = y[1] + h * k[0][1];
        phase = PH_INIT1;
        scalarise();
      } else if (phase == PH_INIT1) {
        float pd = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          const float dd = (k[1][tt] - k[0][tt]) * scr[tt];
          if (counted && (tt == 0 || two)) pd += dd * dd;
        }
        float w0 = wave_sum64(pd), w1 = 0.f;
        if (coupled) {
          if (status != 0) w0 = 0.f;
          w_grid_sum<false>(a.gs, gen, a.epoch, w0, w1);
        }
        const double dt0 = dt;
        const float d2 = sqrtf(w0 / nnorm) * fast_rcp((float)dt0);
        const float dm_ = fmaxf(d1n, d2);
        const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
        const double dn = fmin(100.0 * dt0, dt1);
        dt = dn > dtmax ? dtmax : dn;
        phase = PH_STAGE;
        s = ADJ ? 0 : 1;
        running = begin_step();
        scalarise();
      } else if (s < LAST_STAGE) {
        s++;
      } else if (SPEC && s == LAST_STAGE) {   // the attempt's slopes are complete: its error sum leaves for the grid, and — behind a save time —
                                              // the next attempt's first evaluation runs while the other workgroups' sums arrive
        s2 = err_sum();
        if (coupled) {
          if (status != 0) s2 = 0.f;
          w_grid_publish<false>(a.gs, gen, a.epoch, s2, 0.f);
        }
        if (hit) {
          tmp[0] = yn[0];
          tmp[1] = yn[1];
          jumped(tmp);
        }
        if (hit && j > 0 && status == 0) s = 7;
        else step_end = true;
      } else
        step_end = true;
    } while (!__builtin_amdgcn_readfirstlane((int)step_end));
    PROF_T(g0);
    PROF_ADD(10, a0, g0);   // (diagnostic builds) the attempt's evaluations with their stage arithmetic
    // ---- the end of a step attempt: error norm, controller, accept / reject
    if (ADJ && SOLVER == LDE_SOLVER_RK4) {
      const float h6 = h * (1.0f / 6.0f);
#pragma unroll
      for (int tt = 0; tt < 2; tt++) yn[tt] = y[tt] + h6 * (k[0][tt] + 2.0f * (k[1][tt] + k[2][tt]) + k[3][tt]);
    }
    if (SPEC) {
      if (coupled) w_grid_collect<false>(a.gs, gen, a.epoch, s2, s2b);
    } else {
      s2 = err_sum();
      if (coupled) {
        if (status != 0) s2 = 0.f;
        w_grid_publish<false>(a.gs, gen, a.epoch, s2, s2b);
      }
      if (!ADJ && status == 0) jd = dense_output(j);
      if (coupled) w_grid_collect<false>(a.gs, gen, a.epoch, s2, s2b);
    }
    PROF_T(g1);
    PROF_ADD(12, g0, g1);   // error sum + the grid-wide round trip
    bool accepted = false;
    double hrec = 0.0;   // the attempted step as f64 (the controller overwrites dt below)
    if (status == 0) {
      const float EEst = o.adaptive ? sqrtf(s2 / nnorm) : (s2 == s2 ? 0.f : s2);
      const double hmag = ADJ ? tnew : dt;
      hrec = hmag;
      if (!(EEst == EEst)) {
        if (o.adaptive && hmag > o.dtmin) { nrej++; dt = hmag * (double)o.qmin; }
        else status = 1 + LDE_RET_NONFINITE;
      } else if (o.adaptive) {
        float q11;
        const float q = pi_q(EEst, qold, o, q11);
        if (EEst > 1.0f) {
          nrej++;
          const double nd = hmag * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
          dt = nd;
          if (nd < o.dtmin) status = 1 + LDE_RET_DTMIN;
        } else {
          qold = fmaxf(EEst, 1e-4f);
          double dtp = hmag * (double)fast_rcp(q);
          if (dtp > dtmax) dtp = dtmax;
          dt = dtp;
          accepted = true;
        }
      } else {
        dt = o.dt_fixed;
        accepted = true;
      }
      if (accepted) nacc++;
    }
    if (accepted && o.rec.n && nacc <= o.rec.cap) {   // the step record (forward: start time, size, start state) / the reverse-time trace (size)
      if (tid < 2 && (coupled ? (blockIdx.x == 0 && tid == 0) : bt0 + tid < B)) {
        const size_t ri = (size_t)(nacc - 1) * o.rec.nseq + (coupled ? 0 : bt0 + tid);
        if (!ADJ) o.rec.t[ri] = t;
        o.rec.dt[ri] = hrec;
      }
      if (!ADJ && wv == 0 && lane < Dp) {
        o.rec.y[((size_t)(nacc - 1) * B + bt0) * Dp + lane] = y[0];
        if (two) o.rec.y[((size_t)(nacc - 1) * B + bt0 + 1) * Dp + lane] = y[1];
      }
    }
    if (!ADJ) {
      if (accepted) j = jd;   // (the step's dense output was written ahead: dense_output)
      if (accepted) {
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          y[tt] = yn[tt];
          k[0][tt] = k[LAST_STAGE][tt];
        }
        t = tnew;
        if (last) status = 1;
      }
      s = 1;
      running = begin_step();
    } else {
      if (__builtin_amdgcn_readfirstlane((int)accepted)) {   // (workgroup-uniform: every wave takes bitwise the same decisions — a scalar branch)
        fold(NST);
        if (SPEC) {
#pragma unroll
          for (int tt = 0; tt < 2; tt++) {
            y[tt] = hit ? tmp[tt] : yn[tt];
            k[0][tt] = hit ? scr[tt] : k[6][tt];
          }
          // that evaluation's vectors (the spare ring slot: the seventh stage's, or the speculative one's) are the coming attempt's first stage's
          const f32x4* s4 = reinterpret_cast<const f32x4*>(ring + NST * SLOT);
          f32x4* d4 = reinterpret_cast<f32x4*>(ring);
          for (int i = tid; i < SLOT / 4; i += UT) d4[i] = s4[i];
        } else {
          y[0] = yn[0];
          y[1] = yn[1];
          if (hit) jumped(y);
        }
        if (hit) {
          t = s_ts[j];
          j--;
          if (j < 0) status = 1;
        } else
          t -= tnew;
      }
      s = SPEC ? 1 : 0;   // (SPEC: k₁ and its ring slot are in place — accepted: from above; rejected: the attempt's own)
      running = begin_step();
    }
    PROF_T(g2);
    PROF_ADD(13, g1, g2);   // controller, record, dense output / fold, begin_step
    PROF_ADD(21, g2 - 1, g2);   // attempts
  }

  // ---- results
  const int st = status;
  if (!ADJ) {
    if (st > 1) {
      const float qn = __int_as_float(0x7fc00000);
      for (int tt = 0; tt < (two ? 2 : 1); tt++)
        for (int e = tid; e < Dp * T; e += UT) a.z_out[(size_t)Dp * ((size_t)(bt0 + tt) + (size_t)B * (e / Dp)) + (e % Dp)] = qn;
    }
    if (tid < (two ? 2 : 1)) {
      const int ret = st > 1 ? st - 1 : 0;
      if (a.retcode) a.retcode[bt0 + tid] = ret;
      a.st_ret[bt0 + tid] = ret;
    }
  } else {
    if (wv == 0 && lane >= DP && lane < DP + D) {
      const float fv = (DISC && st == 1 + LDE_RET_MAXITERS) ? __int_as_float(0x7fc00000) : 0.f;   // (no usable step record: NaN, not zeros)
      a.dz0[(size_t)bt0 * D + (lane - DP)] = st > 1 ? fv : y[0];
      if (two) a.dz0[(size_t)(bt0 + 1) * D + (lane - DP)] = st > 1 ? fv : y[1];
    }
    if (tid < (two ? 2 : 1)) a.st_ret[bt0 + tid] = st > 1 ? st - 1 : 0;
    // the workgroup's row of the [workgroups × row stride] slab, flat destructure order (vec(W) column-major [out×in], then b): every
    // entry is owned by exactly one lane (a failed solve contributes zeros, as its dẑ₀ does)
    float* out = a.stage + (size_t)blockIdx.x * a.cap;
    const bool keep = st <= 1;
    const int l15 = lane & 15, e4 = lane >> 4;
    asm volatile("s_nop 15\n\ts_nop 15");   // the last fold's MFMAs → the v_accvgpr_read of their tiles (nothing pads hidden registers)
    auto put2 = [&](const f32x4 tv, int ti, int m) {   // gW₂ᵀ tile (ti, 4m + w): rows = h₁ index (row H₁: gb₂), columns = δ₂ index
      const int oo = 16 * (4 * m + wv) + l15;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = 16 * ti + 4 * e4 + r;
        if (oo < H2) {
          if (i < H1) out[dm.w_off[1] + oo + H2 * i] = keep ? tv[r] : 0.f;
          else if (i == H1) out[dm.b_off[1] + oo] = keep ? tv[r] : 0.f;
        }
      }
    };
    mlpb::static_for<0, 9>([&](auto tic) {
      constexpr int ti = decltype(tic)::value;
      put2(mlpb::areg_read<A0 + 4 * (2 * ti + 0)>(), ti, 0);
      put2(mlpb::areg_read<A0 + 4 * (2 * ti + 1)>(), ti, 1);
    });
    auto put13 = [&](const f32x4 t1, const f32x4 t3, int m, int tk) {
      const int kk = 16 * tk + l15;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int uu = 16 * (4 * m + wv) + 4 * e4 + r;
        if (kk < Dp) {
          if (uu < H1) out[dm.w_off[0] + uu + H1 * kk] = keep ? t1[r] : 0.f;
          if (uu < H2) out[dm.w_off[2] + kk + Dp * uu] = keep ? t3[r] : 0.f;
        }
      }
    };
    mlpb::static_for<0, 2>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      put13(mlpb::areg_read<A0 + 4 * (18 + 2 * m + 0)>(), mlpb::areg_read<A0 + 4 * (22 + 2 * m + 0)>(), m, 0);
      put13(mlpb::areg_read<A0 + 4 * (18 + 2 * m + 1)>(), mlpb::areg_read<A0 + 4 * (22 + 2 * m + 1)>(), m, 1);
    });
    if (tid < H1) out[dm.b_off[0] + tid] = keep ? gb1 : 0.f;
    if (wv == 0 && is_l) out[dm.b_off[2] + row] = keep ? gb3 : 0.f;
  }
  if (tid < (two ? 2 : 1)) {
    const bool rep = !coupled || (blockIdx.x == 0 && tid == 0);   // coupled: one step sequence for the whole batch, reported once
    a.st_nfe[bt0 + tid] = rep ? nfe : 0;
    a.st_nacc[bt0 + tid] = rep ? nacc : 0;
    a.st_nrej[bt0 + tid] = rep ? nrej : 0;
    if (!DISC && o.rec.n && rep) o.rec.n[coupled ? 0 : bt0 + tid] = st > 1 ? 0 : nacc;
  }
}

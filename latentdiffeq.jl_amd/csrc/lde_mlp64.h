// lde_mlp64.h — small networks on small states, per-trajectory control: ONE WAVE per trajectory, everything in registers
// (included by lde_mlp.hip after lde_mlpv.h).
//
// BASELINE.json configs[2] — GOKU pendulum + a 2-64-64-2 MLP, Tsit5, B = 1024 with adjoint — is 1024 independent adaptive
// solves of a 2-dimensional state through a network of 4.4 k weights. In k_mlpv (one trajectory per workgroup, activations and
// state in LDS) every product is a chain of ≈ 100-cycle LDS round trips: ≈ 1 300 cycles per layer for 64×64 work, 5.8 µs per
// evaluation of the adjoint. Here weights, state and control stay in the register file:
//   * lane j owns hidden unit j of both hidden layers (widths ≤ 64); its rows of W₁, W₂, its COLUMNS of W₂, W₃ and its biases
//     are loaded once into VGPRs (≈ 140 of them);
//   * a 64×64 product: the 64 activations go through a 256-byte LDS vector once (one ds_write per lane, one wave per workgroup:
//     in-order LDS, no barrier) and come back as 16 broadcast ds_read_b128 feeding 32 v_pk_fma_f32 on the lane's register row
//     held as pairs — and Wᵀδ is the same loop over the lane's column. (First version: 64 × (v_readlane_b32 → SGPR, v_fmac):
//     one scalar register, two wait states and one accumulator per element — ≈ 770 cycles per product against ≈ 350.)
//   * the D' ≤ 4 outputs are wave sums (four DPP row shifts + four readlanes each);
//   * the state [z; λ; g], the seven slopes and the whole step control are wave-uniform values that every lane carries
//     redundantly (a SIMT machine does that for free), so an evaluation touches no memory besides those LDS vectors;
//   * the weight gradient gW_l = Σ_evaluations w_e δ_l a_lᵀ (a quadrature over the accepted steps, w_e = b_s·|h|) never leaves
//     the wave either (round 3; rounds 1–2 staged every evaluation's (a_l, δ_l) panels to HBM for a second kernel, k_mlp_dw:
//     587 MB written and read back per c3 launch against 0.84 MB of algorithmic traffic, 20 % of the step). The LDS vectors
//     through which h₁ and δ₂ reach every lane are a RING indexed by the stage, so at accept time the attempt's six (h₁, δ₂)
//     pairs are still there: they are folded into four AGPR-resident 32×32 tiles of gW₂ᵀ with v_mfma_f32_32x32x2_f32 (K = two
//     evaluations per instruction, the δ operand scaled by its quadrature weight: 12 MFMAs per accepted step), while the thin
//     layers' gradients (gW₁ = Σ w δ₁zᵀ, gW₃ = Σ w λh₂ᵀ, the biases) are a handful of FMAs per evaluation on lane-owned
//     registers — first into per-attempt sums with the weights b_s, scaled by |h| and added at accept (a rejected attempt's sums
//     are dropped). A wave walks the trajectories b, b + grid, … with the same accumulators and writes ONE row of a [waves × nW]
//     slab in flat destructure order at the end; k_sum_rows adds the rows in a fixed order (bit-reproducible).
// Same algorithm and control arithmetic as k_mlp_adjoint / k_mlpv (HNW initial step, PI controller carried across the save
// times, k₁ re-evaluated every attempt, quadrature weights b_s·|h| of the accepted steps): agreement to solver tolerance. Limits: exactly three Dense layers D' → H₁ → H₂ → D' with H ≤ 64 and D' ∈ {2, 4}-padded, P ≤ 1, per-trajectory
// control (coupled control needs the grid-wide sum: k_mlpv). Anything else runs k_mlpv / the tile kernels.

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef LDE_M64_PF
#define LDE_M64_PF 4   // c3 discrete: 0.2306 → 0.2263 ms per step (abl/ab_mlp.sh, three alternations)
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {   // v + (v of the lane CTRL points at, 0 outside the row)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Σ over the 64 lanes, returned wave-uniform (inclusive prefix within each row of 16, then the four row totals)
__device__ __forceinline__ float wave_sum64(float v) {
  v = dpp_add<0x111>(v);   // row_shr:1
  v = dpp_add<0x112>(v);   // row_shr:2
  v = dpp_add<0x114>(v);   // row_shr:4
  v = dpp_add<0x118>(v);   // row_shr:8
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float lane_bcast(float v, int k) {   // k compile-time after unrolling
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

template <int DP>
struct Net64 {
  float w1[DP], b1;        // row j of W₁ [H₁×D'], bias
  f32x2 w2r[32]; float b2; // row j of W₂ [H₂×H₁], as register pairs (v_pk_fma_f32)
  f32x2 w2c[32];           // column j of W₂: W₂[i][j], i < H₂
  float* hx;               // LDS: two 64-float vectors through which h₁ / δ₂ reach every lane (broadcast ds_read_b128); one pair per stage (a ring)
  float w3c[DP];           // column j of W₃ [D'×H₂]: W₃[d][j]
  float b3[DP];            // uniform
  float ngl, gl2;          // pendulum: −G/L, G/L²
  int act, has_pend;
  float h1, h2;            // post-activation hidden units of the last evaluation (lane j)
};

// f(z) (+ pendulum); leaves h1, h2 in the net
template <int DP>
__device__ __forceinline__ void net64_rhs(Net64<DP>& n, const float (&z)[DP], float (&f)[DP]) {
  float a1 = n.b1;
#pragma unroll
  for (int k = 0; k < DP; k++) a1 += n.w1[k] * z[k];
  n.h1 = act_fn(n.act, a1);
  // 64×64 product: h₁ goes through LDS once and comes back as 16 broadcast 16-byte reads feeding 32 packed FMAs on the lane's
  // register row. (The v_readlane form — SGPR broadcast, one v_fmac per element — is 64 × (readlane, 2 wait states, fmac) on
  // ONE accumulator: ≈ 770 cycles per product against ≈ 350 here.) One wave per workgroup: in-order LDS, no barrier.
  n.hx[threadIdx.x & 63] = n.h1;
  asm volatile("" ::: "memory");
  float a2 = n.b2;
  {
    const f32x4* hv = reinterpret_cast<const f32x4*>(n.hx);
    f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
    // (eight broadcast reads in flight per batch: left alone the compiler rotates three buffers — three reads in flight, sixteen times)
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += 8) {
      f32x4 xv[8];
#pragma unroll
      for (int g = 0; g < 8; g++) xv[g] = hv[g0 + g];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 8; g++) {
        c01 += n.w2r[2 * (g0 + g)] * xv[g].lo;
        c23 += n.w2r[2 * (g0 + g) + 1] * xv[g].hi;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    a2 += (c01.x + c01.y) + (c23.x + c23.y);
  }
  n.h2 = act_fn(n.act, a2);
#pragma unroll
  for (int d = 0; d < DP; d++) f[d] = n.b3[d] + wave_sum64(n.w3c[d] * n.h2);
  if (n.has_pend) {
    f[0] += z[1];
    f[1] += n.ngl * fast_sin(z[0]);
  }
}

// after net64_rhs on the same z: vz = (∂f/∂z)ᵀλ, vth = (∂f/∂L)ᵀλ; d1, d2 = the hidden layers' δ (lane j)
template <int DP>
__device__ __forceinline__ void net64_vjp(const Net64<DP>& n, const float (&z)[DP], const float (&lam)[DP], float (&vz)[DP], float& vth,
                                          float& d1, float& d2) {
  float s2 = 0.f;
#pragma unroll
  for (int d = 0; d < DP; d++) s2 += n.w3c[d] * lam[d];
  d2 = s2 * act_grad(n.act, n.h2);
  n.hx[64 + (threadIdx.x & 63)] = d2;
  asm volatile("" ::: "memory");
  float s1;
  {
    const f32x4* hv = reinterpret_cast<const f32x4*>(n.hx + 64);
    f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
    // (eight broadcast reads in flight per batch: left alone the compiler rotates three buffers — three reads in flight, sixteen times)
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += 8) {
      f32x4 xv[8];
#pragma unroll
      for (int g = 0; g < 8; g++) xv[g] = hv[g0 + g];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 8; g++) {
        c01 += n.w2c[2 * (g0 + g)] * xv[g].lo;
        c23 += n.w2c[2 * (g0 + g) + 1] * xv[g].hi;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s1 = (c01.x + c01.y) + (c23.x + c23.y);
  }
  d1 = s1 * act_grad(n.act, n.h1);
#pragma unroll
  for (int k = 0; k < DP; k++) vz[k] = wave_sum64(n.w1[k] * d1);
  vth = 0.f;
  if (n.has_pend) {
    float sn, cs;
    fast_sincos(z[0], sn, cs);
    vz[0] += n.ngl * cs * lam[1];
    vz[1] += lam[0];
    vth = n.gl2 * sn * lam[1];
  }
}

// weights into registers (flat destructure order: vec(W) column-major [out×in], then b)
template <int DP, bool ADJ>
__device__ __forceinline__ void net64_load(Net64<DP>& n, const MlpDims& dm, const VArgs& a, int lane, float* s_hx) {
  const int Dp = dm.Dp, H1 = dm.sizes[1], H2 = dm.sizes[2];
  const float* W = a.Wflat;
  const float *W1 = W + dm.w_off[0], *W2 = W + dm.w_off[1], *W3 = W + dm.w_off[2];
  // Every load UNCONDITIONAL, from a clamped (always valid) index, and the zero padding applied as a MULTIPLICATION by a 0/1 mask:
  // written as `cond ? W[i] : 0` the compiler predicates each load on its lane condition — ≈ 270 exec-masked blocks, each with
  // `s_waitcnt vmcnt(0)` before the next: 270 dependent L2 round trips before the first step (≈ half of the forward kernel's time
  // at B = 1024); a select after an unconditional load is folded back into the same thing. (w is a finite weight, so w·0 = ±0.)
  const int l1 = min(lane, H1 - 1), l2 = min(lane, H2 - 1);
  const float m1 = lane < H1 ? 1.f : 0.f, m2 = lane < H2 ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < DP; k++) n.w1[k] = W1[l1 + H1 * min(k, Dp - 1)] * (k < Dp ? m1 : 0.f);
  n.b1 = W[dm.b_off[0] + l1] * m1;
#pragma unroll
  for (int k = 0; k < 64; k++) n.w2r[k >> 1][k & 1] = W2[l2 + H2 * min(k, H1 - 1)] * (k < H1 ? m2 : 0.f);
  n.b2 = W[dm.b_off[1] + l2] * m2;
#pragma unroll
  for (int i = 0; i < 64; i++) n.w2c[i >> 1][i & 1] = ADJ ? W2[min(i, H2 - 1) + H2 * l1] * (i < H2 ? m1 : 0.f) : 0.f;
  n.hx = s_hx + 6 * 128;
#pragma unroll
  for (int d = 0; d < DP; d++) {
    n.w3c[d] = W3[min(d, Dp - 1) + Dp * l2] * (d < Dp ? m2 : 0.f);
    n.b3[d] = W[dm.b_off[2] + min(d, Dp - 1)] * (d < Dp ? 1.f : 0.f);
  }
  n.act = dm.act;
  n.has_pend = dm.has_pend;
  n.h1 = n.h2 = 0.f;
}

// the weight gradient of one wave: accepted sums, and the sums of the attempt in flight (weights b_s, not yet scaled by |h|)
template <int DP>
struct Grad64 {
  f32x16 w2[2][2];                                   // gW₂ᵀ tiles [input tile of h₁][output tile of δ₂], C/D layout of v_mfma_f32_32x32x2_f32
  float w1[DP], b1, b2, w3[DP], b3[DP];              // lane j: gW₁[j][k], gb₁[j], gb₂[j], gW₃[d][j]; gb₃[d] (uniform)
  float pw1[DP], pb1, pb2, pw3[DP], pb3[DP];         // the attempt's
};

// the workgroup's row of the [workgroups × row stride] slab, flat destructure order (vec(W) column-major [out×in], then b):
// every wave lays its sums out in LDS, the four copies are added in wave order (fixed: bit-reproducible) and ONE row leaves the CU —
// a quarter of the slab bytes of a row per wave (c3: 18.6 → 4.6 MB written, and as much less read back by k_sum_rows)
template <int DP>
__device__ __forceinline__ void grad64_store_row(const Grad64<DP>& g, const MlpDims& dm, const VArgs& a, int wv, int nwv) {
  extern __shared__ __attribute__((aligned(16))) float s_red[];
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, H1 = dm.sizes[1], H2 = dm.sizes[2];
  float* row = s_red + (size_t)wv * a.cap;
  const int Dpv = dm.Dp;
  for (int e = lane; e < a.cap; e += 64) row[e] = 0.f;   // (entries no lane owns: the padding of the row stride)
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int i = 32 * ti + ((r & 3) | (half << 2) | ((r >> 2) << 3)), oo = 32 * tj + l31;
        if (i < H1 && oo < H2) row[dm.w_off[1] + oo + H2 * i] = g.w2[ti][tj][r];
      }
#pragma unroll
  for (int d = 0; d < DP; d++) {
    if (d < Dpv && lane < H1) row[dm.w_off[0] + lane + H1 * d] = g.w1[d];
    if (d < Dpv && lane < H2) row[dm.w_off[2] + d + Dpv * lane] = g.w3[d];
    if (lane == d && d < Dpv) row[dm.b_off[2] + d] = g.b3[d];
  }
  if (lane < H1) row[dm.b_off[0] + lane] = g.b1;
  if (lane < H2) row[dm.b_off[1] + lane] = g.b2;
  __syncthreads();
  float* out = a.stage + (size_t)blockIdx.x * a.cap;
  for (int e = threadIdx.x; e < a.cap; e += blockDim.x) {
    float t = s_red[e];
    for (int w = 1; w < nwv; w++) t += s_red[(size_t)w * a.cap + e];
    out[e] = t;
  }
}

template <int SOLVER, int DP, bool ADJ>
__device__ __forceinline__ void mlp64_body(const MlpDims& dm, const KOpts& o, const VArgs& a) {
  const int T = o.T, B = o.B, D = dm.D, Dp = dm.Dp, NP = dm.P, lane = threadIdx.x & 63;
  const int wv = ADJ ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0, nwv = ADJ ? (int)(blockDim.x >> 6) : 1;   // the adjoint: NWV waves (trajectories) per workgroup
  constexpr int NS = ADJ ? 2 * DP + 1 : DP;      // [z | λ | g] (g stays 0 without a parameter)
  const int H1 = dm.sizes[1], H2 = dm.sizes[2];
  __shared__ __attribute__((aligned(16))) float s_hx_all[(ADJ ? 4 : 1) * 7 * 128];   // per wave a ring: stage s → h₁ at [128 s, +64), δ₂ at [128 s + 64, +64)
  float* const s_hx = s_hx_all + wv * 7 * 128;
  Net64<DP> n;
  net64_load<DP, ADJ>(n, dm, a, lane, s_hx);
  const double* ts = a.ts;
  const double t0 = ts[0], tend = ts[T - 1], dtmax = fabs(tend - t0);
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;   // weighted evaluations of a step (Tsit5: b₇ = 0)
  auto bq = [](int s_) -> float {                           // quadrature weight of stage s_ (without |h|)
    if (SOLVER == LDE_SOLVER_TSIT5) return ts5::A[6][s_ < 6 ? s_ : 0];
    return (s_ == 0 || s_ == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
  };
  Grad64<DP> g;
  const int half = lane >> 5, l31 = lane & 31;
  float bsel[NST / 2];                                      // the lane's half of the MFMA's K pair: stage 2m + half
  if (ADJ) {
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int tj = 0; tj < 2; tj++)
#pragma unroll
        for (int r = 0; r < 16; r++) g.w2[ti][tj][r] = 0.f;
#pragma unroll
    for (int d = 0; d < DP; d++) g.w1[d] = g.w3[d] = g.b3[d] = g.pw1[d] = g.pw3[d] = g.pb3[d] = 0.f;
    g.b1 = g.b2 = g.pb1 = g.pb2 = 0.f;
#pragma unroll
    for (int m = 0; m < NST / 2; m++) bsel[m] = half ? bq(2 * m + 1) : bq(2 * m);
  }

  for (int b = blockIdx.x * nwv + wv; b < B; b += gridDim.x * nwv) {   // the adjoint walks several trajectories with one set of gradient sums
  {
    float L = 1.f;
    if (dm.has_pend) L = a.theta[(size_t)b * NP];
    n.ngl = -10.0f / L;
    n.gl2 = 10.0f / (L * L);
  }

  // state: y = [z (DP) | λ (DP) | g]
  float y[NS], yn[NS], tmp[NS], k[7][NS], scr[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    y[i] = yn[i] = tmp[i] = scr[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 7; s++) k[s][i] = 0.f;
  }
  bool bad = false;
  if (!ADJ) {
#pragma unroll
    for (int r = 0; r < DP; r++) y[r] = r < D ? a.z0[(size_t)b * D + r] : 0.f;
  } else {
#pragma unroll
    for (int r = 0; r < DP; r++)
      if (r < Dp) {
        const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * (T - 1)) + r;
        y[r] = a.z_out[srcg];
        y[DP + r] = a.dz_out[srcg];
        bad = bad || !isfinite(y[r]);
      }
  }
  double t = ADJ ? tend : t0, dt = 0.0, tnew = 0.0;
  float h = 0.f, qold = 1e-4f, wq = 0.f, d1n = 0.f;
  int status = bad ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1), j = ADJ ? T - 2 : 1, last = 0, hit = 0, nfe = 0, nacc = 0, nrej = 0;
  long long iters = 0;
  if (ADJ && status > 1) {
#pragma unroll
    for (int i = 0; i < NS; i++) y[i] = 0.f;
  }
  if (!ADJ && lane < Dp) a.z_out[(size_t)b * Dp + lane] = lane < D ? a.z0[(size_t)b * D + lane] : 0.f;   // save time 0 = ẑ₀ itself

  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : (ADJ ? 3 : 4);
  const float dirn = ADJ ? -1.f : 1.f;
  const float nnorm = (float)(ADJ ? 2 * Dp + NP : Dp);
  auto counted = [&](int i) -> bool {   // does state entry i count in the norms?
    if (!ADJ) return i < Dp;
    return (i < DP && i < Dp) || (i >= DP && i < 2 * DP && i - DP < Dp) || (i == 2 * DP && NP > 0);
  };

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
        const double dist = t - ts[j];
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

  // one evaluation of the (augmented) right-hand side at `src`; `slot`: the ring slot its h₁ / δ₂ vectors go through; `bs` ≠ 0: the
  // evaluation is a weighted stage of the attempt — its thin-layer gradient terms are added to the attempt's sums
  // FSAL in the reverse solve (Tsit5): inside a save interval f(yₙ₊₁) — the accepted step's seventh evaluation — is the next attempt's k₁:
  // its slope moves to k[0], its (h₁, δ₂) from ring slot 6 to slot 0, and its thin-layer terms (δ₁, h₂, δ₂ of the lane; z and λ are the
  // state itself) open the next attempt's sums with the weight b₁ — the values a re-evaluation would produce. (Behind a save time the
  // state jumps, and a rejected attempt re-evaluates as before: its pending sums are dropped with it.)
  bool fsal = false;
  float last_d1 = 0.f, last_h2 = 0.f, last_d2 = 0.f, f_d1 = 0.f, f_h2 = 0.f, f_d2 = 0.f;
  auto eval = [&](const float (&src)[NS], float (&dst)[NS], int slot, float bs) {
    float z[DP], f[DP];
    n.hx = s_hx + slot * 128;
#pragma unroll
    for (int r = 0; r < DP; r++) z[r] = src[r];
    net64_rhs<DP>(n, z, f);
#pragma unroll
    for (int r = 0; r < DP; r++) dst[r] = f[r];
    if (ADJ) {
      float lam[DP], vz[DP], vth, d1, d2;
#pragma unroll
      for (int r = 0; r < DP; r++) lam[r] = src[DP + r];
      net64_vjp<DP>(n, z, lam, vz, vth, d1, d2);
#pragma unroll
      for (int r = 0; r < DP; r++) dst[DP + r] = -vz[r];
      dst[ADJ ? 2 * DP : 0] = -vth;
      last_d1 = d1;
      last_h2 = n.h2;
      last_d2 = d2;
      if (bs != 0.f) {   // (a₀, δ₁) = (z, δ₁), (a₂, δ₃) = (h₂, λ) and the three bias terms; (a₁, δ₂) = (h₁, δ₂) waits in the ring
        const float bd1 = bs * d1, bh2 = bs * n.h2;
#pragma unroll
        for (int r = 0; r < DP; r++) {
          g.pw1[r] += bd1 * z[r];
          g.pw3[r] += bh2 * lam[r];
          g.pb3[r] += bs * lam[r];
        }
        g.pb1 += bd1;
        g.pb2 += bs * d2;
      }
    }
  };

  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  int phase = (ADJ && !auto_dt) ? PH_STAGE : PH_K0, s = 0;
  bool running = T > 1 && status == 0;
  if (ADJ && running && !auto_dt) {
    dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    running = begin_step();
  }
  // One iteration of the loop = one whole ATTEMPT (all stages, unrolled) once the initial step size is known; the two evaluations
  // that find it (PH_K0, PH_INIT1) keep the one-evaluation-per-iteration form. In every unrolled copy the stage number `s` is a
  // compile-time constant: the stage sum is straight-line code and the slope lands in k[s] directly — as one evaluation per
  // iteration with a run-time `s` every evaluation paid a jump table for the stage sum and 7·NS conditional moves to store its
  // slope "at a static register index".
  while (running) {
    if (phase == PH_STAGE) {
#pragma unroll
      for (int s = ADJ ? 0 : 1; s <= LAST_STAGE; s++) {
        if (ADJ && SOLVER == LDE_SOLVER_TSIT5 && s == 0 && fsal) {   // k₁ = the previous step's f(yₙ₊₁): nothing to evaluate
          const float bs0 = bq(0), bd1 = bs0 * f_d1, bh2 = bs0 * f_h2;
#pragma unroll
          for (int r = 0; r < DP; r++) {
            g.pw1[r] = bd1 * y[r];
            g.pw3[r] = bh2 * y[DP + r];
            g.pb3[r] = bs0 * y[DP + r];
          }
          g.pb1 = bd1;
          g.pb2 = bs0 * f_d2;
          continue;
        }
        float src[NS];
        bool any_w = false;
#pragma unroll
        for (int i = 0; i < NS; i++) src[i] = y[i];
        if (phase == PH_STAGE) {
          if (SOLVER == LDE_SOLVER_TSIT5) {
            if (s > 0) {
#define M64STAGE(S_)                                                                  \
      case S_:                                                                            \
        _Pragma("unroll") for (int i = 0; i < NS; i++) {                                  \
          float accv = ts5::A[S_][0] * k[0][i];                                           \
          _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += ts5::A[S_][jj] * k[jj][i]; \
          src[i] = y[i] + h * accv;                                                       \
        }                                                                                 \
        break;
              switch (s) {
                M64STAGE(1) M64STAGE(2) M64STAGE(3) M64STAGE(4) M64STAGE(5) M64STAGE(6)
                default: break;
              }
#undef M64STAGE
              if (s == 6) {
#pragma unroll
                for (int i = 0; i < NS; i++) yn[i] = src[i];
              }
            }
            any_w = ADJ && s < 6;
          } else if (ADJ || s < 4) {
            if (s > 0) {
              const float cs = (s == 3 ? 1.0f : 0.5f) * h;
#pragma unroll
              for (int i = 0; i < NS; i++) src[i] = y[i] + cs * (s == 1 ? k[0][i] : (s == 2 ? k[1][i] : k[2][i]));
            }
            any_w = ADJ;
          } else {
            const float h6 = h * (1.0f / 6.0f);
#pragma unroll
            for (int i = 0; i < NS; i++) {
              yn[i] = y[i] + h6 * (k[0][i] + 2.0f * (k[1][i] + k[2][i]) + k[3][i]);
              src[i] = yn[i];
            }
          }
          if (ADJ && s == 0) {   // a new attempt: its thin-layer sums start from zero
#pragma unroll
            for (int d = 0; d < DP; d++) g.pw1[d] = g.pw3[d] = g.pb3[d] = 0.f;
            g.pb1 = g.pb2 = 0.f;
          }
        }
        float dst[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) dst[i] = 0.f;
        eval(src, dst, s, any_w ? bq(s) : 0.f);
        {   // store the slope where the phase wants it (static register indices)
          const int ks = phase == PH_K0 ? 0 : (phase == PH_INIT1 ? 1 : s);
#pragma unroll
          for (int q = 0; q < 7; q++)
            if (q == ks) {
#pragma unroll
              for (int i = 0; i < NS; i++) k[q][i] = dst[i];
            }
        }
        if (status == 0) nfe++;

      }
      if (ADJ && SOLVER == LDE_SOLVER_RK4) {
        const float h6 = h * (1.0f / 6.0f);
#pragma unroll
        for (int i = 0; i < NS; i++) yn[i] = y[i] + h6 * (k[0][i] + 2.0f * (k[1][i] + k[2][i]) + k[3][i]);
      }
      float s2 = 0.f;
      bool fin = true;
#pragma unroll
      for (int i = 0; i < NS; i++) {
        fin = fin && isfinite(yn[i]);
        if (o.adaptive && counted(i)) {
          float er = ts5::BT[0] * k[0][i];
#pragma unroll
          for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * k[jj][i];
          er *= h;
          const float sk = o.abstol + fmaxf(fabsf(y[i]), fabsf(yn[i])) * o.reltol;
          const float r = er * fast_rcp(sk);
          s2 += r * r;
        }
      }
      if (!fin) s2 = __int_as_float(0x7fc00000);
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
        const size_t ri = (size_t)(nacc - 1) * B + b;
        if (lane == 0) {
          if (!ADJ) o.rec.t[ri] = t;
          o.rec.dt[ri] = hrec;
        }
        if (!ADJ) {
          float yv = 0.f;
#pragma unroll
          for (int r = 0; r < DP; r++) yv = lane == r ? y[r] : yv;
          if (lane < Dp) o.rec.y[ri * Dp + lane] = yv;
        }
      }
      if (!ADJ) {
        while (accepted && j < T && ts[j] <= tnew) {   // dense output at every save time inside the accepted step
          const double tj = ts[j];
          const float th = (tj >= tnew || (j == T - 1 && last)) ? 2.0f : (float)(tj - t) * fast_rcp(wq);
          float outv = 0.f;
#pragma unroll
          for (int r = 0; r < DP; r++) {
            float ov;
            if (th > 1.5f) ov = yn[r];
            else if (SOLVER == LDE_SOLVER_TSIT5) {
              float bw[7];
              tsit5_interp_weights(th, bw);
              float acc = bw[0] * k[0][r];
#pragma unroll
              for (int q = 1; q < 7; q++) acc += bw[q] * k[q][r];
              ov = y[r] + wq * acc;
            } else {
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              ov = h00 * y[r] + (h10 * wq) * k[0][r] + h01 * yn[r] + (h11 * wq) * k[4][r];
            }
            outv = lane == r ? ov : outv;
          }
          if (lane < Dp) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * j) + lane] = outv;
          j++;
        }
        if (accepted) {
#pragma unroll
          for (int i = 0; i < NS; i++) {
            y[i] = yn[i];
            k[0][i] = k[LAST_STAGE][i];
          }
          t = tnew;
          if (last) status = 1;
        }
        s = 1;
        running = begin_step();
      } else {
        if (accepted) {   // the accepted step's share of the quadrature gW = Σ |h| b_s (∂f/∂W)ᵀλ
#pragma unroll
          for (int d = 0; d < DP; d++) {
            g.w1[d] += wq * g.pw1[d];
            g.w3[d] += wq * g.pw3[d];
            g.b3[d] += wq * g.pb3[d];
          }
          g.b1 += wq * g.pb1;
          g.b2 += wq * g.pb2;
          // gW₂ᵀ[i][o] += Σ_s (h₁)_s[i] · (|h| b_s δ₂)_s[o]: the stages' vectors are still in the ring; one MFMA takes two stages
          // (A: lane → h₁ of stage 2m + half, unit 32 ti + (lane & 31); B: the same lane map on the scaled δ₂)
#pragma unroll
          for (int m = 0; m < NST / 2; m++) {
            const float* rs = s_hx + (2 * m + half) * 128 + l31;
            const float wsc = wq * bsel[m];
            const float a0 = rs[0], a1 = rs[32], b0 = rs[64] * wsc, b1 = rs[96] * wsc;
            g.w2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, g.w2[0][0], 0, 0, 0);
            g.w2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, g.w2[0][1], 0, 0, 0);
            g.w2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, g.w2[1][0], 0, 0, 0);
            g.w2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, g.w2[1][1], 0, 0, 0);
          }
        }
        if (accepted) {
#pragma unroll
          for (int i = 0; i < NS; i++) y[i] = yn[i];
          if (hit) {
#pragma unroll
            for (int r = 0; r < DP; r++)
              if (r < Dp) {
                const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * j) + r;
                y[DP + r] += a.dz_out[srcg];
                if (o.checkpoint) y[r] = a.z_out[srcg];
              }
            t = ts[j];
            j--;
            if (j < 0) status = 1;
          } else
            t -= tnew;
        }
        fsal = false;
        if (SOLVER == LDE_SOLVER_TSIT5 && accepted && !hit && status == 0) {
          fsal = true;
#pragma unroll
          for (int i = 0; i < NS; i++) k[0][i] = k[6][i];
          s_hx[lane] = s_hx[6 * 128 + lane];             // (this wave's ring: LDS operations of one wave execute in order)
          s_hx[64 + lane] = s_hx[6 * 128 + 64 + lane];
          f_d1 = last_d1;
          f_h2 = last_h2;
          f_d2 = last_d2;
        }
        s = 0;
        running = begin_step();
      }
    } else {
      float src[NS];
#pragma unroll
      for (int i = 0; i < NS; i++) src[i] = phase == PH_INIT1 ? tmp[i] : y[i];
      float dst[NS];
#pragma unroll
      for (int i = 0; i < NS; i++) dst[i] = 0.f;
      eval(src, dst, 6, 0.f);
      {   // store the slope where the phase wants it (static register indices)
        const int ks = phase == PH_K0 ? 0 : (phase == PH_INIT1 ? 1 : s);
#pragma unroll
        for (int q = 0; q < 7; q++)
          if (q == ks) {
#pragma unroll
            for (int i = 0; i < NS; i++) k[q][i] = dst[i];
          }
      }
      if (status == 0) nfe++;

      if (phase == PH_K0 && !(ADJ || auto_dt)) {
        dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
        phase = PH_STAGE;
        s = 1;
        running = begin_step();
      } else if (phase == PH_K0) {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int i = 0; i < NS; i++) {
          const float sk = fast_rcp(o.abstol + fabsf(y[i]) * o.reltol);
          scr[i] = sk;
          const float a0 = y[i] * sk, a1 = k[0][i] * sk;
          if (counted(i)) { v0 += a0 * a0; v1 += a1 * a1; }
        }
        const float d0 = sqrtf(v0 / nnorm);
        d1n = sqrtf(v1 / nnorm);
        double dt0 = (d0 < 1e-5f || d1n < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1n));
        if (dt0 > dtmax) dt0 = dtmax;
        dt = dt0;
        h = status == 0 ? dirn * (float)dt0 : 0.f;
#pragma unroll
        for (int i = 0; i < NS; i++) tmp[i] = y[i] + h * k[0][i];
        phase = PH_INIT1;
      } else if (phase == PH_INIT1) {
        float w0 = 0.f;
#pragma unroll
        for (int i = 0; i < NS; i++) {
          const float dd = (k[1][i] - k[0][i]) * scr[i];
          if (counted(i)) w0 += dd * dd;
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
      }
    }
  }

  const int st = status;
  if (!ADJ) {
    if (st > 1) {
      const float qn = __int_as_float(0x7fc00000);
      for (int e = lane; e < Dp * T; e += 64) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * (e / Dp)) + (e % Dp)] = qn;
    }
    if (lane == 0) {
      const int ret = st > 1 ? st - 1 : 0;
      if (a.retcode) a.retcode[b] = ret;
      a.st_ret[b] = ret;
    }
  } else {
    float g0 = 0.f;
#pragma unroll
    for (int r = 0; r < DP; r++) g0 = lane == r ? y[DP + r] : g0;
    if (lane < D) a.dz0[(size_t)b * D + lane] = st > 1 ? 0.f : g0;
    if (lane == 0 && NP > 0) a.dtheta[(size_t)b * NP] = st > 1 ? 0.f : y[ADJ ? 2 * DP : 0];
    if (lane == 0) a.st_ret[b] = st > 1 ? st - 1 : 0;
  }
  if (lane == 0) {
    a.st_nfe[b] = nfe;
    a.st_nacc[b] = nacc;
    a.st_nrej[b] = nrej;
    if (o.rec.n) o.rec.n[b] = st > 1 ? 0 : nacc;
  }
  }   // trajectories of this wave

  if (ADJ) grad64_store_row<DP>(g, dm, a, wv, nwv);
}

// LDE_SENSE_DISCRETE on the same register layout (lde_mlpd.h has the algorithm; [REF examples/pendulum_friction-less/pendulum.jl:11],
// [REF src/models/GOKU.jl:107, :121]): a wave sweeps its trajectory's recorded steps (t_n, dt_n, y_n) from the last to the first — pass 1
// rebuilds the slopes k_1 … k_S with forward evaluations, the save times inside the step put their cotangents on y_n / the slopes /
// y_{n+1}, pass 2 pulls k̄ through f at y_{n+1} and at g_S … g_2 (the vector-Jacobian half alone: the hidden units of every stage point
// are lane-owned registers that pass 1 keeps, so no point is evaluated twice; (h₁, δ₂) through ring slot S − i). The weight gradient is folded per step exactly as in the continuous adjoint, at weight 1 (the scale h·b_i is
// inside k̄). S forward + S vector-Jacobian halves per accepted FORWARD step — c3: ≈ 160 half-evaluations per trajectory against ≈ 480
// fused ones of the reverse-time solve.
template <int SOLVER, int DP>
__device__ __forceinline__ void mlp64_disc_body(const MlpDims& dm, const KOpts& o, const VArgs& a) {
  const int T = o.T, B = o.B, D = dm.D, Dp = dm.Dp, NP = dm.P, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = (int)(blockDim.x >> 6);
  __shared__ __attribute__((aligned(16))) float s_hx_all[4 * 7 * 128];
  float* const s_hx = s_hx_all + wv * 7 * 128;
  Net64<DP> n;
  net64_load<DP, true>(n, dm, a, lane, s_hx);
  constexpr int S = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                              {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
  auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
  const double* ts = a.ts;
  const double tend = ts[T - 1];
  const StepRec R = o.rec;
  Grad64<DP> g;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int r = 0; r < 16; r++) g.w2[ti][tj][r] = 0.f;
#pragma unroll
  for (int d = 0; d < DP; d++) g.w1[d] = g.w3[d] = g.b3[d] = g.pw1[d] = g.pw3[d] = g.pb3[d] = 0.f;
  g.b1 = g.b2 = g.pb1 = g.pb2 = 0.f;

  // the vector-Jacobian half at (z, k̄) with the hidden units (h1, h2) its forward evaluation produced (lane-owned registers: pass 1 keeps
  // them per stage, so nothing is evaluated twice): vz = Jᵀk̄, the thin layers' gradient terms at weight 1, δ₂ into ring slot `slot` — whose h₁
  // half pass 1 filled (put_h1: the FSAL point's, which belongs to the step before, is written here)
  auto vjp_at = [&](const float (&z)[DP], const float (&kb)[DP], int slot, float h1, float h2, bool put_h1, float (&vz)[DP], float& gth) {
    float vth, d1, d2;
    n.hx = s_hx + slot * 128;
    n.h1 = h1;
    n.h2 = h2;
    if (put_h1) n.hx[lane] = h1;
    net64_vjp<DP>(n, z, kb, vz, vth, d1, d2);
    gth += vth;
#pragma unroll
    for (int r = 0; r < DP; r++) {
      g.w1[r] += d1 * z[r];
      g.w3[r] += n.h2 * kb[r];
      g.b3[r] += kb[r];
    }
    g.b1 += d1;
    g.b2 += d2;
  };
  // gW₂ᵀ += Σ over ring slots [0, nslots) of h₁ ⊗ δ₂ (one MFMA takes two slots)
  auto fold = [&](int nslots) {
#pragma unroll
    for (int m = 0; m < S / 2; m++) {
      const float* rs = s_hx + (2 * m + half) * 128 + l31;
      const float wsc = 2 * m + half < nslots ? 1.f : 0.f;
      const float a0 = rs[0], a1 = rs[32], b0 = rs[64] * wsc, b1 = rs[96] * wsc;
      g.w2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, g.w2[0][0], 0, 0, 0);
      g.w2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, g.w2[0][1], 0, 0, 0);
      g.w2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, g.w2[1][0], 0, 0, 0);
      g.w2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, g.w2[1][1], 0, 0, 0);
    }
  };
  for (int e = lane; e < 7 * 128; e += 64) s_hx[e] = 0.f;   // (slots a fold reads at weight 0 must hold finite numbers)

  for (int b = blockIdx.x * nwv + wv; b < B; b += gridDim.x * nwv) {
    {
      float L = 1.f;
      if (dm.has_pend) L = a.theta[(size_t)b * NP];
      n.ngl = -10.0f / L;
      n.gl2 = 10.0f / (L * L);
    }
    float y0[DP], yb[DP], carry[DP], gth = 0.f;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < DP; r++) {
      y0[r] = r < Dp ? a.z_out[(size_t)b * Dp + r] : 0.f;   // save time 0 is ẑ₀ itself
      bad = bad || !isfinite(y0[r]);
      yb[r] = carry[r] = 0.f;
    }
    const int ns = R.n[b];
    int status = bad ? 1 + LDE_RET_NONFINITE : ((T > 1 && (ns < 1 || ns > R.cap)) ? 1 + LDE_RET_MAXITERS : 1);
    int nfe = 0;
    if (status == 1 && T > 1) {
      int j = T - 1;
      double tnext = tend;
      float h1f = 0.f, h2f = 0.f;   // hidden units at the FSAL point y_{n+1} of the step being reversed (= the first stage point of step n + 1)
#if LDE_M64_PF
      // every load of the sweep is requested a link ahead of its use — the next step's record while this step's evaluations run, the Δẑ of
      // the next LDE_M64_PF save times while this one is applied: with one wave per SIMD nothing else covers a load at the head of a link
      double t_pf = R.t[(size_t)(ns - 1) * B + b], dt_pf = R.dt[(size_t)(ns - 1) * B + b];
      float yN_pf[DP], dq[LDE_M64_PF][DP];
#pragma unroll
      for (int r = 0; r < DP; r++) yN_pf[r] = r < Dp ? R.y[((size_t)(ns - 1) * B + b) * Dp + r] : 0.f;
#pragma unroll
      for (int u = 0; u < LDE_M64_PF; u++)
#pragma unroll
        for (int r = 0; r < DP; r++) dq[u][r] = r < Dp ? a.dz_out[(size_t)Dp * ((size_t)b + (size_t)B * (j - u > 0 ? j - u : 0)) + r] : 0.f;
      double tq[LDE_M64_PF];   // … and the save times themselves (the grid is in global memory)
#pragma unroll
      for (int u = 0; u < LDE_M64_PF; u++) tq[u] = ts[j - u > 0 ? j - u : 0];
#endif
#pragma unroll 1
      for (int s = ns - 1; s >= 0; s--) {
#if LDE_M64_PF
        const double t = t_pf, dt = dt_pf;
#else
        const double t = R.t[(size_t)s * B + b], dt = R.dt[(size_t)s * B + b];
#endif
        const float h = (float)dt;
        const bool last = s == ns - 1;
        const double tnew = tnext;
        tnext = t;
        float yN[DP], K[S][DP];
#if LDE_M64_PF
        {
          const size_t sp = s > 0 ? s - 1 : 0;
#pragma unroll
          for (int r = 0; r < DP; r++) yN[r] = yN_pf[r];
          t_pf = R.t[sp * B + b];
          dt_pf = R.dt[sp * B + b];
#pragma unroll
          for (int r = 0; r < DP; r++) yN_pf[r] = r < Dp ? R.y[(sp * B + b) * Dp + r] : 0.f;
        }
#else
#pragma unroll
        for (int r = 0; r < DP; r++) yN[r] = r < Dp ? R.y[((size_t)s * B + b) * Dp + r] : 0.f;
#endif
        auto point = [&](int i, float (&zp)[DP]) {   // g_i (i < S) or y_{n+1} (i == S) from y_n and the slopes
#pragma unroll
          for (int r = 0; r < DP; r++) {
            float zv = yN[r];
            if (i > 0) {
              if (SOLVER == LDE_SOLVER_RK4 && i == S) zv = yN[r] + (h * (1.0f / 6.0f)) * (K[0][r] + 2.0f * (K[1][r] + K[2][r]) + K[3][r]);
              else {
                float acc = A(i, 0) * K[0][r];
#pragma unroll
                for (int q = 1; q < i; q++) acc += A(i, q) * K[q][r];
                zv = yN[r] + h * acc;
              }
            }
            zp[r] = zv;
          }
        };
        // ---- pass 1: the slopes — forward evaluations; stage point i ≥ 1 leaves its h₁ in the ring slot its pullback will use (S − i) and
        //      its hidden units in this lane's registers; the first one's (needed when step n − 1 is reversed) go through the scratch slot
        float h1s[S], h2s[S];
#pragma unroll
        for (int i = 0; i < S; i++) {
          float zp[DP];
          point(i, zp);
          n.hx = s_hx + (i == 0 ? 6 : S - i) * 128;
          net64_rhs<DP>(n, zp, K[i]);
          h1s[i] = n.h1;
          h2s[i] = n.h2;
        }
        if (last) {   // the end state of the solve: the FSAL point of the last step
          float zp[DP], fe[DP];
          point(S, zp);
          n.hx = s_hx + 6 * 128;
          net64_rhs<DP>(n, zp, fe);
          h1f = n.h1;
          h2f = n.h2;
          nfe++;
        }
        // ---- the save times inside the step
        float KB[S + 1][DP], ybn[DP];
#pragma unroll
        for (int r = 0; r < DP; r++) {
#pragma unroll
          for (int i = 0; i < S; i++) KB[i][r] = 0.f;
          KB[S][r] = carry[r];
          ybn[r] = 0.f;
        }
        const float rh = fast_rcp(h);
#if LDE_M64_PF
        while (j >= 1 && tq[0] > t) {
          const double tj = tq[0];
#pragma unroll
          for (int u = 0; u + 1 < LDE_M64_PF; u++) tq[u] = tq[u + 1];
          tq[LDE_M64_PF - 1] = ts[j - LDE_M64_PF > 0 ? j - LDE_M64_PF : 0];
#else
        while (j >= 1 && ts[j] > t) {
          const double tj = ts[j];
#endif
          float dj[DP];
#if LDE_M64_PF
#pragma unroll
          for (int r = 0; r < DP; r++) {
            dj[r] = dq[0][r];
#pragma unroll
            for (int u = 0; u + 1 < LDE_M64_PF; u++) dq[u][r] = dq[u + 1][r];
            dq[LDE_M64_PF - 1][r] = r < Dp ? a.dz_out[(size_t)Dp * ((size_t)b + (size_t)B * (j - LDE_M64_PF > 0 ? j - LDE_M64_PF : 0)) + r] : 0.f;
          }
#else
#pragma unroll
          for (int r = 0; r < DP; r++) dj[r] = r < Dp ? a.dz_out[(size_t)Dp * ((size_t)b + (size_t)B * j) + r] : 0.f;
#endif
          if (tj >= tnew || (j == T - 1 && last)) {
#pragma unroll
            for (int r = 0; r < DP; r++) yb[r] += dj[r];
          } else {
            const float th = (float)(tj - t) * rh;
            if (SOLVER == LDE_SOLVER_TSIT5) {
              float bw[7];
              tsit5_interp_weights(th, bw);
#pragma unroll
              for (int r = 0; r < DP; r++) {
                ybn[r] += dj[r];
#pragma unroll
                for (int q = 0; q < 7; q++) KB[q][r] += (h * bw[q]) * dj[r];
              }
            } else {
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
#pragma unroll
              for (int r = 0; r < DP; r++) {
                ybn[r] += h00 * dj[r];
                KB[0][r] += (h10 * h) * dj[r];
                yb[r] += h01 * dj[r];
                KB[S][r] += (h11 * h) * dj[r];
              }
            }
          }
          j--;
        }
        // ---- pass 2: Jᵀk̄ at y_{n+1}, then at g_S … g_2
#pragma unroll
        for (int i = S; i >= 1; i--) {
          float zp[DP], vz[DP];
          point(i, zp);
          vjp_at(zp, KB[i], S - i, i == S ? h1f : h1s[i < S ? i : 0], i == S ? h2f : h2s[i < S ? i : 0], i == S, vz, gth);
          if (i == S) {
#pragma unroll
            for (int r = 0; r < DP; r++) {
              yb[r] += vz[r];
#pragma unroll
              for (int q = 0; q < S; q++) KB[q][r] += (h * A(S, q)) * yb[r];
              ybn[r] += yb[r];
            }
          } else {
#pragma unroll
            for (int r = 0; r < DP; r++) {
              ybn[r] += vz[r];
#pragma unroll
              for (int q = 0; q < i; q++)
                if (A(i, q) != 0.f) KB[q][r] += (h * A(i, q)) * vz[r];
            }
          }
        }
        fold(S);
#pragma unroll
        for (int r = 0; r < DP; r++) {
          carry[r] = KB[0][r];
          yb[r] = ybn[r];
        }
        h1f = h1s[0];   // this step's first stage point is the FSAL point of the step before it
        h2f = h2s[0];
        nfe += 2 * S;
      }
      {   // k_1 of the first step = f(y_0): the first stage point of step 0, evaluated in the last iteration's pass 1
        float vz[DP];
        vjp_at(y0, carry, 0, h1f, h2f, true, vz, gth);
        fold(1);
#pragma unroll
        for (int r = 0; r < DP; r++) yb[r] += vz[r];
        nfe++;
      }
    }
    const float qn = __int_as_float(0x7fc00000);
    float g0 = 0.f;
#pragma unroll
    for (int r = 0; r < DP; r++) g0 = lane == r ? yb[r] : g0;
    if (lane < D) a.dz0[(size_t)b * D + lane] = status == 1 ? g0 + a.dz_out[(size_t)b * Dp + lane] : (status == 1 + LDE_RET_MAXITERS ? qn : 0.f);
    if (lane == 0 && NP > 0) a.dtheta[(size_t)b * NP] = status == 1 ? gth : (status == 1 + LDE_RET_MAXITERS ? qn : 0.f);
    if (lane == 0) {
      a.st_ret[b] = status > 1 ? status - 1 : 0;
      a.st_nfe[b] = nfe;
      a.st_nacc[b] = status == 1 && T > 1 ? ns : 0;
      a.st_nrej[b] = 0;
    }
  }
  grad64_store_row<DP>(g, dm, a, wv, nwv);
}

template <int SOLVER, int DP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mlp64(MlpDims dm, KOpts o, VArgs a) {
  mlp64_body<SOLVER, DP, false>(dm, o, a);
}
// the adjoint keeps 64 accumulator registers of gW₂ᵀ on top of the weights: one wave per SIMD, 512 registers; four waves (one per SIMD of
// a CU, each on its own trajectories) form a workgroup so that their gradient sums meet in LDS before they leave the CU
template <int SOLVER, int DP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp64_adj(MlpDims dm, KOpts o, VArgs a) {
  mlp64_body<SOLVER, DP, true>(dm, o, a);
}

template <int SOLVER, int DP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp64_disc(MlpDims dm, KOpts o, VArgs a) {
  mlp64_disc_body<SOLVER, DP>(dm, o, a);
}

// dW[e] += Σ_s rows[s][e], s in index order (bit-reproducible): the rows the adjoint's waves left. A workgroup owns 64 consecutive
// entries; its sixteen waves each add a contiguous block of rows (eight loads in flight), the sixteen partial sums meet in LDS and
// are added in wave order.
// assign != 0: dW[e] = the sum (option "adjoint_overwrite": the caller's zero fill disappears).
static __global__ void __launch_bounds__(1024) k_sum_rows(const float* __restrict__ rows, int nrows, int stride, int n, float* __restrict__ dW,
                                                          int assign = 0) {
  __shared__ float part[16][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6, e = blockIdx.x * 64 + c;
  const int per = (nrows + 15) / 16, r0 = rg * per, r1 = min(nrows, r0 + per);
  float s = 0.f;
  if (e < n) {
    const float* p = rows + e;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = p[(size_t)(r + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; r < r1; r++) s += p[(size_t)r * stride];
  }
  part[rg][c] = s;
  __syncthreads();
  if (rg == 0 && e < n) {
    float t = part[0][c];
#pragma unroll
    for (int u = 1; u < 16; u++) t += part[u][c];
    dW[e] = assign ? t : dW[e] + t;
  }
}

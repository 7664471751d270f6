// lde_mlp.hip — right-hand sides with a Dense/relu MLP: the LatentODE path and "physics + MLP".
//
// Replaces what runs under
//     nODE = NeuralODE(dudt, (t[1], t[end]), solver; saveat = t, kwargs...); ẑ = Array(nODE(ẑ₀))
//                                                                   [REF src/models/LatentODE.jl:70-72]
// with dudt = Chain(Dense(D', H, relu), Dense(H, H, relu), Dense(H, D'))     [REF examples/pendulum_friction-less/nODE.jl:12-14]
// (DiffEqFlux 1.52.0 / OrdinaryDiffEq 6.27.1 / SciMLSensitivity 7.10.0, un-vendored) and its pullback.
//
// Design (gfx950)
//  * A workgroup (256 threads = 4 waves, one per SIMD) owns a tile of NB = 16 trajectories (columns) for the
//    WHOLE solve: state, the seven Tsit5 slopes and the hidden activations of the tile live in LDS as
//    [row][18] f32 panels (stride 18 ⇒ the MFMA operand reads below are bank-conflict-free or 2-way).
//  * Every Dense layer is Y[out×16] = W[out×in]·X[in×16] on the f32 matrix cores:
//    v_mfma_f32_16x16x4_f32, A = a 16×4 weight fragment, B = a 4×16 slab of the activation panel. Weights are
//    re-laid out ONCE per lde_set_weights into MFMA fragment order (one coalesced 256-B load per fragment, for W
//    and for Wᵀ), so the inner loop is {1 global/L2 load, 1 ds_read_b32, 1 MFMA}; f32 MFMA runs at the f32 vector
//    rate (64 FLOP/clk/SIMD), which an L2-resident weight stream can feed.
//  * Per-trajectory step control (GOKU semantics) costs nothing extra: all 16 columns run the same stage of
//    their own step (own t, dt, accept/reject) in lock-step; a finished column idles with h = 0.
//  * Coupled control (NeuralODE semantics: one dt, RMS norm over all D'·B entries) needs one grid-wide sum per
//    step: each workgroup publishes its partial, a monotonic-counter barrier (agent-scope release/acquire)
//    follows, and every workgroup adds the partials in the same order ⇒ bitwise identical decisions everywhere.
//  * Adjoint: reverse-time Tsit5/RK4 on [z; λ; g_θ] with the MLP re-evaluated at every stage (relu masks are
//    recomputed, not stored), vector-Jacobian products through Wᵀ fragments, and the weight gradient
//    gW += (w·δ_l)·act_{l-1}ᵀ accumulated in MFMA accumulators across the stages of a step (K = the 16 columns
//    of the tile), committed to the workgroup's slab only when the step is accepted.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "lde_device.h"

namespace lde {

constexpr int NB = 16;        // trajectories (columns) per workgroup
constexpr int LDP = 18;       // LDS panel stride (floats)
constexpr int NTHREADS = 256;
constexpr int MAXL = LDE_MAX_LAYERS;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Static description of the RHS handed to the kernels by value.
struct MlpDims {
  int nL;                 // Dense layers
  int sizes[MAXL + 1];    // [in, h1, ..., out]
  int act;                // hidden activation
  int D, Dp, P;           // state_dim, D + augment_dim, param_dim
  int has_pend;           // PENDULUM_PLUS_MLP
  int frag_off[MAXL];     // float offset of layer l's W fragments   (RT_l × KS_l × 64)
  int fragT_off[MAXL];    // float offset of layer l's Wᵀ fragments  (RTt_l × KSt_l × 64)
  int w_off[MAXL];        // offset of vec(W_l) in the flat (destructure-order) weight vector
  int b_off[MAXL];        // offset of b_l
  int hmax;               // widest hidden panel
  int coupled;            // LDE_BATCH_COUPLED
  int solver;
  int nW;
  int bias_lin[MAXL];     // offset of layer l's bias gradient in the compact [Σ out] LDS vector
  int nbias;
  int tile_off[MAXL + 1]; // first weight-gradient tile (16×16 of Wᵀ) of layer l in the global tile enumeration
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- one-time weight re-layout: flat destructure order → MFMA fragment order (W and Wᵀ) ---------------
// fragment (rt, ks) of a matrix M[R×K]: lane l holds M[rt*16 + (l&15)][ks*4 + (l>>4)] (0 outside).
__global__ void k_build_frags(const float* __restrict__ Wflat, MlpDims dm, float* __restrict__ frag,
                              float* __restrict__ fragT) {
  const int l = blockIdx.y;
  const int in = dm.sizes[l], out = dm.sizes[l + 1];
  const float* W = Wflat + dm.w_off[l];  // column-major [out×in]: W(o,i) at o + out*i
  {
    const int RT = cdiv(out, 16), KS = cdiv(in, 4), n = RT * KS * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
      const int lane = e & 63, f = e >> 6, rt = f / KS, ks = f % KS;
      const int o = rt * 16 + (lane & 15), i = ks * 4 + (lane >> 4);
      frag[dm.frag_off[l] + e] = (o < out && i < in) ? W[o + (size_t)out * i] : 0.f;
    }
  }
  {
    const int RT = cdiv(in, 16), KS = cdiv(out, 4), n = RT * KS * 64;  // Wᵀ[in×out]
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
      const int lane = e & 63, f = e >> 6, rt = f / KS, ks = f % KS;
      const int i = rt * 16 + (lane & 15), o = ks * 4 + (lane >> 4);
      fragT[dm.fragT_off[l] + e] = (o < out && i < in) ? W[o + (size_t)out * i] : 0.f;
    }
  }
}

__device__ __forceinline__ float act_fn(int kind, float x) { return kind == LDE_ACT_TANH ? tanhf(x) : fmaxf(x, 0.f); }
__device__ __forceinline__ float act_grad(int kind, float a) { return kind == LDE_ACT_TANH ? 1.f - a * a : (a > 0.f ? 1.f : 0.f); }

// Y[R×16] (+)= M[R×K]·X[K×16] for one workgroup; M given as fragments. EPI(row, col, acc) stores.
// Row tiles are dealt to the 4 waves; two tiles are kept in flight per wave to cover the 40-cycle
// dependent-accumulator latency of v_mfma_f32_16x16x4_f32.
template <class Epi>
__device__ __forceinline__ void panel_gemm(const float* __restrict__ frag, int R, int K, const float* X, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int RT = cdiv(R, 16), KS = cdiv(K, 4);
  const float* xb = X + (lane >> 4) * LDP + (lane & 15);
  for (int rt = wave; rt < RT; rt += 8) {
    const int rt2 = rt + 4;
    const bool two = rt2 < RT;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* f0 = frag + (size_t)rt * KS * 64 + lane;
    const float* f1 = frag + (size_t)(two ? rt2 : rt) * KS * 64 + lane;
    // K-steps in chunks of 8: all 24 operand loads of a chunk are issued before its 16 MFMAs, so the L2 latency of
    // the weight-fragment stream is paid once per chunk instead of once per K-step.
    for (int ks0 = 0; ks0 < KS; ks0 += 8) {
      float a0[8], a1[8], bb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int ks = min(ks0 + u, KS - 1);
        a0[u] = f0[ks * 64];
        a1[u] = f1[ks * 64];
        bb[u] = xb[ks * 4 * LDP];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (ks0 + u < KS) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bb[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bb[u], acc1, 0, 0, 0);
        }
      }
    }
    const int col = lane & 15, rbase = (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = rt * 16 + rbase + r;
      if (row < R) epi(row, col, acc0[r]);
    }
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = rt2 * 16 + rbase + r;
        if (row < R) epi(row, col, acc1[r]);
      }
    }
  }
}

// ---- grid-wide deterministic sum (coupled mode) ----------------------------------------------------
struct GridSync {
  unsigned* counter;   // monotonic arrival counter, zeroed before the launch
  float* slots;        // [2][nWG][4] partials (ping-pong by generation parity)
  int* abort_flag;     // set when a spin times out
  int nwg;
};

// every thread of the workgroup calls this; v[0..3] of thread 0 are the workgroup's partials; returns the totals.
__device__ __forceinline__ void grid_sum4(const GridSync& gs, unsigned& gen, float (&v)[4], float* s_bcast) {
  if (gs.nwg == 1) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) s_bcast[i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = s_bcast[i];
    __syncthreads();
    return;
  }
  gen++;
  float* slots = gs.slots + (size_t)(gen & 1) * gs.nwg * 4;
  if (threadIdx.x == 0) {
    float* mine = slots + (size_t)blockIdx.x * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) mine[i] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(gs.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * (unsigned)gs.nwg;
    long long spins = 0;
    while (__hip_atomic_load(gs.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000000LL) {  // ≈ seconds: a peer is not resident — give up instead of hanging the GPU
        __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < gs.nwg; w++) {  // fixed order ⇒ identical totals in every workgroup
#pragma unroll
      for (int i = 0; i < 4; i++) t[i] += slots[(size_t)w * 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s_bcast[i] = t[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = s_bcast[i];
  __syncthreads();
}

// ---- per-column control block kept in LDS --------------------------------------------------------------
struct Ctl {
  double t[NB], dt[NB], tnew[NB];
  float h[NB], qold[NB], eest[NB], ngl[NB], gl2[NB], wq[NB], th[NB];
  int status[NB];   // 0 = integrating, 1 = finished, ≥2 = failed with retcode status-1
  int j[NB], accepted[NB], last[NB], hit[NB], nfe[NB], nacc[NB], nrej[NB], iters[NB], savej[NB];
  int any_active, any_save, all_accepted;
  float bcast[4];
};

struct Panels {       // LDS carve-up
  float* y;           // [NSr][LDP]   NSr = rows of the integrated state, padded to a multiple of 4
  float* yn;
  float* tmp;
  float* kbase;       // k[s] = kbase + s*pstride   (computed, not an array of pointers: no scratch)
  float* hidbase;     // hidden activations (post-activation) of layers 0..nL-2: hid[l] = hidbase + l*hstride
  float* delbase;     // backprop panels (adjoint only)
  float* scr;         // [NSr][LDP] scratch (error terms)
  int pstride, hstride;
  __device__ __forceinline__ float* k(int s) const { return kbase + s * pstride; }
  __device__ __forceinline__ float* hid(int l) const { return hidbase + l * hstride; }
  __device__ __forceinline__ float* del(int i) const { return delbase + i * hstride; }
};

// f(z) for the tile: dst[0:Dp] = MLP(src[0:Dp]) (+ pendulum); hidden activations are left in P.hid[*].
__device__ __forceinline__ void eval_rhs(const MlpDims& dm, const float* __restrict__ frag,
                                         const float* __restrict__ Wflat, const Panels& P, const Ctl* c,
                                         const float* src, float* dst) {
  const float* X = src;
  for (int l = 0; l < dm.nL; l++) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const bool lastl = l == dm.nL - 1;
    float* Y = lastl ? dst : P.hid(l);
    const float* bias = Wflat + dm.b_off[l];
    const int actk = dm.act;
    panel_gemm(frag + dm.frag_off[l], out, in, X, [&](int row, int col, float v) {
      v += bias[row];
      if (!lastl) v = act_fn(actk, v);
      Y[row * LDP + col] = v;
    });
    __syncthreads();
    X = Y;
  }
  if (dm.has_pend) {
    if (threadIdx.x < NB) {
      const int col = threadIdx.x;
      const float x = src[0 * LDP + col], yv = src[1 * LDP + col];
      dst[0 * LDP + col] += yv;
      dst[1 * LDP + col] += c->ngl[col] * fast_sin(x);
    }
    __syncthreads();
  }
}

// ================================================ forward ==================================================
struct FwdArgs {
  const float* z0;
  const float* theta;
  const double* ts;
  const float* frag;
  const float* Wflat;
  float* z_out;
  int32_t* retcode;
  int32_t *st_nfe, *st_nacc, *st_nrej, *st_ret;
  GridSync gs;
};

__global__ void __launch_bounds__(NTHREADS) k_mlp_forward(MlpDims dm, KOpts o, FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = o.T, B = o.B, Dp = dm.Dp, D = dm.D;
  // ---- carve LDS -------------------------------------------------------------------------------------
  Ctl* c = reinterpret_cast<Ctl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(Ctl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  Panels P;
  const int NS = Dp;
  const int NSr = (NS + 3) & ~3;   // panels are padded to a multiple of 4 rows (K-steps read whole groups of 4)
  float* p = base;
  P.y = p; p += NSr * LDP;
  P.yn = p; p += NSr * LDP;
  P.tmp = p; p += NSr * LDP;
  P.pstride = NSr * LDP;
  P.hstride = dm.hmax * LDP;
  P.kbase = p; p += 7 * P.pstride;
  P.scr = p; p += NSr * LDP;
  P.hidbase = p; p += (dm.nL > 1 ? dm.nL - 1 : 0) * P.hstride;
  P.delbase = nullptr;
  const int nfloat = (int)(p - base);
  for (int i = threadIdx.x; i < nfloat; i += NTHREADS) base[i] = 0.f;   // pad rows/cols must be finite (0·x)
  for (int i = threadIdx.x; i < T; i += NTHREADS) s_ts[i] = a.ts[i];
  __syncthreads();

  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * NB;
  const int nel = NS * NB;                      // elements of a state panel handled cooperatively
  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = tend - t0;
  unsigned gen = 0;

  // ---- load the tile: column-major z0 [D×B]; augmented rows stay 0 -----------------------------------
  for (int e = tid; e < NB * D; e += NTHREADS) {
    const int col = e / D, row = e % D;
    if (b0 + col < B) P.y[row * LDP + col] = a.z0[(size_t)(b0 + col) * D + row];
  }
  if (tid < NB) {
    const int col = tid;
    const bool valid = b0 + col < B;
    c->t[col] = t0;
    c->dt[col] = 0.0;
    c->qold[col] = 1e-4f;
    c->status[col] = (valid && T > 1) ? 0 : 1;
    c->j[col] = 1;
    c->nfe[col] = c->nacc[col] = c->nrej[col] = c->iters[col] = 0;
    c->h[col] = 0.f;
    float L = 1.f;
    if (dm.has_pend && valid) L = a.theta[(size_t)(b0 + col) * dm.P];
    c->ngl[col] = -10.0f / L;
  }
  __syncthreads();
  // save time 0 = ẑ₀ itself (augmented rows 0)
  for (int e = tid; e < NB * Dp; e += NTHREADS) {
    const int col = e / Dp, row = e % Dp;
    if (b0 + col < B) a.z_out[(size_t)(b0 + col) * Dp + row] = P.y[row * LDP + col];
  }

  if (T > 1) {
    eval_rhs(dm, a.frag, a.Wflat, P, c, P.y, P.k(0));
    if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;

    // ---- initial step size ---------------------------------------------------------------------------
    if (o.adaptive && !(o.dt_fixed > 0)) {
      // d0, d1 (per column, or over the whole batch when coupled)
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB;
        const float yv = P.y[row * LDP + col];
        const float sk = fast_rcp(o.abstol + fabsf(yv) * o.reltol);
        P.scr[row * LDP + col] = sk;
        const float a0 = yv * sk, a1 = P.k(0)[row * LDP + col] * sk;
        P.tmp[row * LDP + col] = a0 * a0;
        P.yn[row * LDP + col] = a1 * a1;
      }
      __syncthreads();
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (tid < NB) {
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < NS; r++) { s0 += P.tmp[r * LDP + tid]; s1 += P.yn[r * LDP + tid]; }
        c->eest[tid] = s0;   // reuse as scratch: Σ (y/sk)²
        c->wq[tid] = s1;     //                   Σ (f0/sk)²
      }
      __syncthreads();
      if (coupled) {
        if (tid == 0) {
          for (int col = 0; col < NB; col++)
            if (b0 + col < B) { v[0] += c->eest[col]; v[1] += c->wq[col]; }
        }
        grid_sum4(a.gs, gen, v, c->bcast);
      }
      if (tid < NB) {
        const float n = coupled ? (float)NS * (float)B : (float)NS;
        const float d0 = sqrtf((coupled ? v[0] : c->eest[tid]) / n), d1 = sqrtf((coupled ? v[1] : c->wq[tid]) / n);
        double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
        if (dt0 > dtmax) dt0 = dtmax;
        c->dt[tid] = dt0;
        c->h[tid] = (float)dt0;
        c->th[tid] = d1;
      }
      __syncthreads();
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB;
        P.tmp[row * LDP + col] = P.y[row * LDP + col] + c->h[col] * P.k(0)[row * LDP + col];
      }
      __syncthreads();
      eval_rhs(dm, a.frag, a.Wflat, P, c, P.tmp, P.k(1));
      if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB;
        const float d = (P.k(1)[row * LDP + col] - P.k(0)[row * LDP + col]) * P.scr[row * LDP + col];
        P.yn[row * LDP + col] = d * d;
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.yn[r * LDP + tid];
        c->eest[tid] = s2;
      }
      __syncthreads();
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (coupled) {
        if (tid == 0)
          for (int col = 0; col < NB; col++)
            if (b0 + col < B) w[0] += c->eest[col];
        grid_sum4(a.gs, gen, w, c->bcast);
      }
      if (tid < NB) {
        const float n = coupled ? (float)NS * (float)B : (float)NS;
        const double dt0 = c->dt[tid];
        const float d2 = sqrtf((coupled ? w[0] : c->eest[tid]) / n) * fast_rcp((float)dt0);
        const float dm_ = fmaxf(c->th[tid], d2);
        const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
        double dt = fmin(100.0 * dt0, dt1);
        c->dt[tid] = dt > dtmax ? dtmax : dt;
      }
      __syncthreads();
    } else if (tid < NB) {
      c->dt[tid] = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    }
    __syncthreads();

    // ---- main loop -------------------------------------------------------------------------------------
    for (;;) {
      if (tid < NB) {
        const int col = tid;
        if (c->status[col] == 0) {
          if (c->iters[col]++ >= o.maxiters) c->status[col] = 1 + LDE_RET_MAXITERS;
        }
        if (c->status[col] == 0) {
          double dt = c->dt[col];
          const double t = c->t[col];
          int last = 0;
          if (t + dt >= tend - 1e-12 * fabs(tend)) { dt = tend - t; last = 1; }
          c->last[col] = last;
          c->tnew[col] = last ? tend : t + dt;
          c->h[col] = (float)dt;
          c->wq[col] = (float)dt;   // step actually attempted (f32)
          c->dt[col] = dt;
        } else
          c->h[col] = 0.f;
      }
      __syncthreads();
      if (tid == 0) {
        int any = 0;
        for (int col = 0; col < NB; col++) any |= (c->status[col] == 0);
        c->any_active = any;
      }
      __syncthreads();
      int any_active = c->any_active;
      if (coupled && a.gs.nwg > 1) {  // all workgroups make identical decisions, but leave together
        // (identical control arithmetic ⇒ any_active agrees everywhere; nothing to exchange)
      }
      if (!any_active) break;

      if (dm.solver == LDE_SOLVER_TSIT5) {
        for (int s = 1; s <= 6; s++) {
          float* dst = s < 6 ? P.tmp : P.yn;
          for (int e = tid; e < nel; e += NTHREADS) {
            const int row = e / NB, col = e % NB, idx = row * LDP + col;
            float acc = ts5::A[s][0] * P.k(0)[idx];
            for (int jj = 1; jj < s; jj++) acc += ts5::A[s][jj] * P.k(jj)[idx];
            dst[idx] = P.y[idx] + c->h[col] * acc;
          }
          __syncthreads();
          eval_rhs(dm, a.frag, a.Wflat, P, c, dst, P.k(s));
        }
        if (tid < NB && c->status[tid] == 0) c->nfe[tid] += 6;
      } else {  // classical RK4; k[4] = f(yn) doubles as the next k1
        for (int s = 1; s <= 3; s++) {
          const float cs = s == 3 ? 1.0f : 0.5f;
          for (int e = tid; e < nel; e += NTHREADS) {
            const int row = e / NB, col = e % NB, idx = row * LDP + col;
            P.tmp[idx] = P.y[idx] + (cs * c->h[col]) * P.k(s - 1)[idx];
          }
          __syncthreads();
          eval_rhs(dm, a.frag, a.Wflat, P, c, P.tmp, P.k(s));
        }
        for (int e = tid; e < nel; e += NTHREADS) {
          const int row = e / NB, col = e % NB, idx = row * LDP + col;
          P.yn[idx] = P.y[idx] + (c->h[col] * (1.0f / 6.0f)) * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        }
        __syncthreads();
        eval_rhs(dm, a.frag, a.Wflat, P, c, P.yn, P.k(4));
        if (tid < NB && c->status[tid] == 0) c->nfe[tid] += 4;
      }

      // ---- error estimate -----------------------------------------------------------------------------
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB, idx = row * LDP + col;
        float r2 = 0.f;
        const float yv = P.y[idx], ynv = P.yn[idx];
        if (o.adaptive) {
          float er = ts5::BT[0] * P.k(0)[idx];
#pragma unroll
          for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * P.k(jj)[idx];
          er *= c->h[col];
          const float sk = o.abstol + fmaxf(fabsf(yv), fabsf(ynv)) * o.reltol;
          const float r = er * fast_rcp(sk);
          r2 = r * r;
        }
        // non-finite state poisons the column's sum
        P.scr[idx] = isfinite(ynv) ? r2 : __int_as_float(0x7fc00000);
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.scr[r * LDP + tid];
        c->eest[tid] = s2;
      }
      __syncthreads();
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (coupled) {
        if (tid == 0)
          for (int col = 0; col < NB; col++)
            if (b0 + col < B) v[0] += c->eest[col];
        grid_sum4(a.gs, gen, v, c->bcast);
      }

      // ---- accept / reject, next dt (one lane per column) ----------------------------------------------
      if (tid < NB && c->status[tid] == 0) {
        const int col = tid;
        const float n = coupled ? (float)NS * (float)B : (float)NS;
        const float s2 = coupled ? v[0] : c->eest[col];
        const float EEst = o.adaptive ? sqrtf(s2 / n) : (s2 == s2 ? 0.f : s2);
        const double dt = c->dt[col];
        int accepted = 0;
        if (!(EEst == EEst)) {  // non-finite step
          if (o.adaptive && dt > o.dtmin) { c->nrej[col]++; c->dt[col] = dt * (double)o.qmin; }
          else c->status[col] = 1 + LDE_RET_NONFINITE;
        } else if (o.adaptive) {
          float q11;
          const float q = pi_q(EEst, c->qold[col], o, q11);
          if (EEst > 1.0f) {
            c->nrej[col]++;
            const double nd = dt * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
            c->dt[col] = nd;
            if (nd < o.dtmin) c->status[col] = 1 + LDE_RET_DTMIN;
          } else {
            c->qold[col] = fmaxf(EEst, 1e-4f);
            double dtp = dt * (double)fast_rcp(q);
            if (dtp > dtmax) dtp = dtmax;
            c->dt[col] = dtp;
            accepted = 1;
          }
        } else {
          c->dt[col] = o.dt_fixed;
          accepted = 1;
        }
        c->accepted[col] = accepted;
        if (accepted) c->nacc[col]++;
      } else if (tid < NB)
        c->accepted[tid] = 0;
      __syncthreads();

      // ---- dense output at every save time inside the accepted step ------------------------------------
      for (;;) {
        if (tid < NB) {
          const int col = tid;
          int sv = 0;
          if (c->accepted[col] && c->j[col] < T && s_ts[c->j[col]] <= c->tnew[col]) {
            const double tj = s_ts[c->j[col]];
            const int jj = c->j[col];
            c->savej[col] = jj;
            c->th[col] = (tj >= c->tnew[col] || (jj == T - 1 && c->last[col])) ? 2.0f
                                                                               : (float)(tj - c->t[col]) * fast_rcp(c->wq[col]);
            c->j[col] = jj + 1;
            sv = 1;
          }
          c->hit[col] = sv;
        }
        __syncthreads();
        if (tid == 0) {
          int any = 0;
          for (int col = 0; col < NB; col++) any |= c->hit[col];
          c->any_save = any;
        }
        __syncthreads();
        if (!c->any_save) break;
        for (int e = tid; e < NB * Dp; e += NTHREADS) {
          const int col = e / Dp, row = e % Dp, idx = row * LDP + col;
          if (!c->hit[col]) continue;
          const float th = c->th[col], h = c->wq[col];
          float out;
          if (th > 1.5f) out = P.yn[idx];
          else if (dm.solver == LDE_SOLVER_TSIT5) {
            float bw[7];
            tsit5_interp_weights(th, bw);
            float acc = bw[0] * P.k(0)[idx];
#pragma unroll
            for (int s = 1; s < 7; s++) acc += bw[s] * P.k(s)[idx];
            out = P.y[idx] + h * acc;
          } else {
            const float om = 1.0f - th;
            const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
            const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
            out = h00 * P.y[idx] + (h10 * h) * P.k(0)[idx] + h01 * P.yn[idx] + (h11 * h) * P.k(4)[idx];
          }
          a.z_out[(size_t)Dp * ((size_t)(b0 + col) + (size_t)B * c->savej[col]) + row] = out;
        }
        __syncthreads();
      }

      // ---- advance accepted columns ----------------------------------------------------------------------
      {
        const int fs = dm.solver == LDE_SOLVER_TSIT5 ? 6 : 4;
        for (int e = tid; e < nel; e += NTHREADS) {
          const int row = e / NB, col = e % NB, idx = row * LDP + col;
          if (c->accepted[col]) {
            P.y[idx] = P.yn[idx];
            P.k(0)[idx] = P.k(fs)[idx];
          }
        }
      }
      if (tid < NB && c->accepted[tid]) {
        c->t[tid] = c->tnew[tid];
        if (c->last[tid]) c->status[tid] = 1;
      }
      __syncthreads();
    }
  }

  // ---- epilogue: NaN blocks for failed columns, statistics -------------------------------------------------
  for (int e = tid; e < NB * Dp; e += NTHREADS) {
    const int col = e / Dp, row = e % Dp;
    if (b0 + col < B && c->status[col] > 1) {
      const float qn = __int_as_float(0x7fc00000);
      for (int jj = 0; jj < T; jj++) a.z_out[(size_t)Dp * ((size_t)(b0 + col) + (size_t)B * jj) + row] = qn;
    }
  }
  if (tid < NB && b0 + tid < B) {
    const int col = tid, b = b0 + col;
    const int ret = c->status[col] > 1 ? c->status[col] - 1 : 0;
    if (a.retcode) a.retcode[b] = ret;
    a.st_ret[b] = ret;
    const bool rep = !coupled || b == 0;   // coupled: one step sequence for the whole batch, reported once
    a.st_nfe[b] = rep ? c->nfe[col] : 0;
    a.st_nacc[b] = rep ? c->nacc[col] : 0;
    a.st_nrej[b] = rep ? c->nrej[col] : 0;
  }
}


// ================================================ adjoint ==================================================
struct BwdArgs {
  const float* z_out;
  const float* dz_out;
  const float* theta;
  const double* ts;
  const float* frag;
  const float* fragT;
  const float* Wflat;
  float* dz0;
  float* dtheta;
  float* slab;        // [nWG][nW] this launch's per-workgroup weight-gradient slabs
  int32_t *st_nfe, *st_nacc, *st_nrej, *st_ret;
  GridSync gs;
};

// f, −(∂f/∂z)ᵀλ, −(∂f/∂θ)ᵀλ for the tile, and the weighted outer products of this stage.
//   src/dst rows: [0,Dp) z | [Dp,2Dp) λ | [2Dp,2Dp+P) g.   wst[col] = quadrature weight of this stage (0 ⇒ none)
template <int NDW>
__device__ __forceinline__ void eval_bwd(const MlpDims& dm, const BwdArgs& a, const Panels& P, const Ctl* c,
                                         const float* src, float* dst, const float* wst, bool any_w,
                                         f32x4 (&acc)[NDW], float* bstep, const int* tile_off) {
  const int Dp = dm.Dp, nL = dm.nL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 1. forward through the MLP (relu masks are recomputed here, not stored by the forward solve)
  eval_rhs(dm, a.frag, a.Wflat, P, c, src, dst);
  // 2. back-propagate λ; δ_L = λ_stage
  const float* dl = src + Dp * LDP;
  for (int l = nL - 1; l >= 0; l--) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const float* al = l == 0 ? src : P.hid(l - 1);   // input activation of layer l
    if (any_w) {
      // gWᵀ tile [i][o] += Σ_n a_l[i][n] · (w_n δ[o][n]); tiles of layer l owned by this wave
      const int IT = cdiv(in, 16);
      const int t0 = tile_off[l], t1 = tile_off[l + 1];
      float wl[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) wl[s4] = wst[s4 * 4 + (lane >> 4)];
#pragma unroll
      for (int m = 0; m < NDW; m++) {
        const int t = wave + 4 * m;
        if (t >= t0 && t < t1) {
          const int tt = t - t0, ot = tt / IT, it = tt % IT;
          const float* ap = al + (it * 16 + (lane & 15)) * LDP + (lane >> 4);
          const float* bp = dl + (ot * 16 + (lane & 15)) * LDP + (lane >> 4);
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s4 * 4], bp[s4 * 4] * wl[s4], acc[m], 0, 0, 0);
        }
      }
      for (int row = threadIdx.x; row < out; row += NTHREADS) {
        float sacc = 0.f;
#pragma unroll
        for (int n = 0; n < NB; n++) sacc += wst[n] * dl[row * LDP + n];
        bstep[dm.bias_lin[l] + row] += sacc;   // each (layer,row) is owned by exactly one thread
      }
    }
    // δ_in = W_lᵀ δ  (⊙ act'(a_l) for hidden layers); layer 0 gives (∂f/∂z)ᵀλ
    if (l > 0) {
      float* dn = P.del((nL - 1 - l) & 1);
      const int actk = dm.act;
      panel_gemm(a.fragT + dm.fragT_off[l], in, out, dl, [&](int row, int col, float v) {
        dn[row * LDP + col] = v * act_grad(actk, al[row * LDP + col]);
      });
      __syncthreads();
      dl = dn;
    } else {
      float* dlam = dst + Dp * LDP;
      panel_gemm(a.fragT + dm.fragT_off[0], in, out, dl, [&](int row, int col, float v) { dlam[row * LDP + col] = -v; });
      __syncthreads();
    }
  }
  // 3. known-physics part: J = [[0,1],[ngl·cos x, 0]], ∂f₂/∂L = gl2·sin x
  if (dm.has_pend) {
    if (threadIdx.x < NB) {
      const int col = threadIdx.x;
      float sn, cs;
      fast_sincos(src[0 * LDP + col], sn, cs);
      const float l0 = src[(Dp + 0) * LDP + col], l1 = src[(Dp + 1) * LDP + col];
      // eval_rhs already added the pendulum to f (rows 0,1)
      dst[(Dp + 0) * LDP + col] -= c->ngl[col] * cs * l1;
      dst[(Dp + 1) * LDP + col] -= l0;
      dst[(2 * Dp) * LDP + col] = -(c->gl2[col] * sn * l1);
    }
    __syncthreads();
  }
}

// Reverse-time solve of [z; λ; g_θ] for one tile, with forced stops + jumps at the save times.
template <int NDW>
__global__ void __launch_bounds__(NTHREADS) k_mlp_adjoint(MlpDims dm, KOpts o, BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = o.T, B = o.B, Dp = dm.Dp, D = dm.D, NP = dm.P;
  Ctl* c = reinterpret_cast<Ctl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(Ctl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  Panels P;
  const int NS = 2 * Dp + NP;
  const int NSr = (NS + 3) & ~3;
  float* p = base;
  P.pstride = NSr * LDP;
  P.hstride = dm.hmax * LDP;
  P.y = p; p += P.pstride;
  P.yn = p; p += P.pstride;
  P.tmp = p; p += P.pstride;
  P.kbase = p; p += 7 * P.pstride;
  P.scr = p; p += P.pstride;
  P.hidbase = p; p += (dm.nL > 1 ? dm.nL - 1 : 0) * P.hstride;
  P.delbase = p; p += 2 * P.hstride;
  float* bstep = p; p += (dm.nbias + 3) & ~3;
  float* wst = p; p += NB;
  const int nfloat = (int)(p - base);
  for (int i = threadIdx.x; i < nfloat; i += NTHREADS) base[i] = 0.f;
  for (int i = threadIdx.x; i < T; i += NTHREADS) s_ts[i] = a.ts[i];
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * NB;
  const int nel = NS * NB;
  const bool coupled = dm.coupled != 0;
  const double tT = s_ts[T - 1], dtmax = fabs(tT - s_ts[0]);
  unsigned gen = 0;
  float* slab = a.slab + (size_t)blockIdx.x * dm.nW;
  for (int i = tid; i < dm.nW; i += NTHREADS) slab[i] = 0.f;

  f32x4 acc[NDW];
#pragma unroll
  for (int m = 0; m < NDW; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // commit the step's outer products (held as Wᵀ tiles [i][o] in MFMA accumulators) and bias sums to the slab
  auto commit = [&]() {
    for (int l = 0; l < dm.nL; l++) {
      const int in = dm.sizes[l], out = dm.sizes[l + 1], IT = cdiv(in, 16);
      const int t0 = dm.tile_off[l], t1 = dm.tile_off[l + 1];
      float* gW = slab + dm.w_off[l];
#pragma unroll
      for (int m = 0; m < NDW; m++) {
        const int t = wave + 4 * m;
        if (t >= t0 && t < t1) {
          const int tt = t - t0, ot = tt / IT, it = tt % IT;
          const int oc = ot * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int ir = it * 16 + (lane >> 4) * 4 + r;
            if (oc < out && ir < in) gW[oc + (size_t)out * ir] += acc[m][r];
          }
          acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      float* gb = slab + dm.b_off[l];
      for (int row = tid; row < out; row += NTHREADS) {
        gb[row] += bstep[dm.bias_lin[l] + row];
        bstep[dm.bias_lin[l] + row] = 0.f;
      }
    }
  };
  auto discard = [&]() {
#pragma unroll
    for (int m = 0; m < NDW; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < dm.nbias; i += NTHREADS) bstep[i] = 0.f;
  };

  // ---- load the terminal condition: z = ẑ(t_T), λ = Δ_T, g = 0 ------------------------------------------
  for (int e = tid; e < NB * Dp; e += NTHREADS) {
    const int col = e / Dp, row = e % Dp;
    if (b0 + col < B) {
      const size_t src = (size_t)Dp * ((size_t)(b0 + col) + (size_t)B * (T - 1)) + row;
      P.y[row * LDP + col] = a.z_out[src];
      P.y[(Dp + row) * LDP + col] = a.dz_out[src];
    }
  }
  __syncthreads();
  if (tid < NB) {
    const int col = tid;
    const bool valid = b0 + col < B;
    bool bad = false;
    for (int r = 0; r < Dp; r++) bad = bad || !isfinite(P.y[r * LDP + col]);
    c->t[col] = tT;
    c->dt[col] = 0.0;
    c->qold[col] = 1e-4f;
    // a failed forward trajectory is a constant NaN block ⇒ zero gradient  [REF GOKU.jl:114]
    c->status[col] = !valid ? 1 : (bad ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1));
    c->j[col] = T - 2;
    c->nfe[col] = c->nacc[col] = c->nrej[col] = c->iters[col] = 0;
    c->h[col] = 0.f;
    float L = 1.f;
    if (dm.has_pend && valid) L = a.theta[(size_t)(b0 + col) * NP];
    c->ngl[col] = -10.0f / L;
    c->gl2[col] = 10.0f / (L * L);
    wst[col] = 0.f;
  }
  __syncthreads();
  if (tid < NB && c->status[tid] > 1) {   // neutralise the NaN column so that it cannot trip the tile-wide logic
    for (int r = 0; r < NS; r++) P.y[r * LDP + tid] = 0.f;
  }
  __syncthreads();

  if (T > 1) {
    // ---- initial step size (Hairer) on the augmented state, direction −1 ------------------------------------
    if (o.adaptive && !(o.dt_fixed > 0)) {
      eval_bwd<NDW>(dm, a, P, c, P.y, P.k(0), wst, false, acc, bstep, dm.tile_off);
      if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB, idx = row * LDP + col;
        const float yv = P.y[idx];
        const float sk = fast_rcp(o.abstol + fabsf(yv) * o.reltol);
        P.scr[idx] = sk;
        const float a0 = yv * sk, a1 = P.k(0)[idx] * sk;
        P.tmp[idx] = a0 * a0;
        P.yn[idx] = a1 * a1;
      }
      __syncthreads();
      if (tid < NB) {
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < NS; r++) { s0 += P.tmp[r * LDP + tid]; s1 += P.yn[r * LDP + tid]; }
        c->eest[tid] = s0;
        c->wq[tid] = s1;
      }
      __syncthreads();
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (coupled) {
        if (tid == 0)
          for (int col = 0; col < NB; col++)
            if (c->status[col] == 0) { v[0] += c->eest[col]; v[1] += c->wq[col]; }
        grid_sum4(a.gs, gen, v, c->bcast);
      }
      if (tid < NB) {
        const float n = coupled ? (float)NS * (float)B : (float)NS;
        const float d0 = sqrtf((coupled ? v[0] : c->eest[tid]) / n), d1 = sqrtf((coupled ? v[1] : c->wq[tid]) / n);
        double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
        if (dt0 > dtmax) dt0 = dtmax;
        c->dt[tid] = dt0;
        c->h[tid] = c->status[tid] == 0 ? -(float)dt0 : 0.f;
        c->th[tid] = d1;
      }
      __syncthreads();
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB, idx = row * LDP + col;
        P.tmp[idx] = P.y[idx] + c->h[col] * P.k(0)[idx];
      }
      __syncthreads();
      eval_bwd<NDW>(dm, a, P, c, P.tmp, P.k(1), wst, false, acc, bstep, dm.tile_off);
      if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB, idx = row * LDP + col;
        const float dd = (P.k(1)[idx] - P.k(0)[idx]) * P.scr[idx];
        P.yn[idx] = dd * dd;
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.yn[r * LDP + tid];
        c->eest[tid] = s2;
      }
      __syncthreads();
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (coupled) {
        if (tid == 0)
          for (int col = 0; col < NB; col++)
            if (c->status[col] == 0) w[0] += c->eest[col];
        grid_sum4(a.gs, gen, w, c->bcast);
      }
      if (tid < NB) {
        const float n = coupled ? (float)NS * (float)B : (float)NS;
        const double dt0 = c->dt[tid];
        const float d2 = sqrtf((coupled ? w[0] : c->eest[tid]) / n) * fast_rcp((float)dt0);
        const float dm_ = fmaxf(c->th[tid], d2);
        const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
        double dt = fmin(100.0 * dt0, dt1);
        c->dt[tid] = dt > dtmax ? dtmax : dt;
      }
      __syncthreads();
    } else if (tid < NB) {
      c->dt[tid] = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    }
    __syncthreads();

    // ---- main loop ---------------------------------------------------------------------------------------------
    bool replay = false;
    for (;;) {
      if (!replay) {
        if (tid < NB) {
          const int col = tid;
          if (c->status[col] == 0 && c->iters[col]++ >= o.maxiters) c->status[col] = 1 + LDE_RET_MAXITERS;
          if (c->status[col] == 0) {
            const double tstop = s_ts[c->j[col]];
            const double dist = c->t[col] - tstop;
            double hmag = c->dt[col];
            int hit = 0;
            if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = 1; }
            c->hit[col] = hit;
            c->tnew[col] = hmag;            // step magnitude actually attempted
            c->h[col] = -(float)hmag;
            c->wq[col] = (float)hmag;       // quadrature weight scale |h|
          } else {
            c->h[col] = 0.f;
            c->wq[col] = 0.f;
            c->hit[col] = 0;
          }
        }
        __syncthreads();
        if (tid == 0) {
          int any = 0;
          for (int col = 0; col < NB; col++) any |= (c->status[col] == 0);
          c->any_active = any;
        }
        __syncthreads();
        if (!c->any_active) break;
      }

      // ---- stages (k₁ is evaluated fresh: it carries this step's quadrature weight) ---------------------------
      if (dm.solver == LDE_SOLVER_TSIT5) {
        for (int s = 0; s <= 6; s++) {
          const float* src = P.y;
          if (s > 0) {
            float* dst = s < 6 ? P.tmp : P.yn;
            for (int e = tid; e < nel; e += NTHREADS) {
              const int row = e / NB, col = e % NB, idx = row * LDP + col;
              float accv = ts5::A[s][0] * P.k(0)[idx];
              for (int jj = 1; jj < s; jj++) accv += ts5::A[s][jj] * P.k(jj)[idx];
              dst[idx] = P.y[idx] + c->h[col] * accv;
            }
            src = dst;
          }
          const float bs = s < 6 ? ts5::A[6][s] : 0.f;
          if (tid < NB) wst[tid] = (s < 6 && (!replay || c->accepted[tid])) ? c->wq[tid] * bs : 0.f;
          __syncthreads();
          eval_bwd<NDW>(dm, a, P, c, src, P.k(s), wst, s < 6, acc, bstep, dm.tile_off);
        }
        if (!replay && tid < NB && c->status[tid] == 0) c->nfe[tid] += 7;
      } else {
        for (int s = 0; s <= 3; s++) {
          const float* src = P.y;
          if (s > 0) {
            const float cs = s == 3 ? 1.0f : 0.5f;
            for (int e = tid; e < nel; e += NTHREADS) {
              const int row = e / NB, col = e % NB, idx = row * LDP + col;
              P.tmp[idx] = P.y[idx] + (cs * c->h[col]) * P.k(s - 1)[idx];
            }
            src = P.tmp;
          }
          const float bs = (s == 0 || s == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
          if (tid < NB) wst[tid] = (!replay || c->accepted[tid]) ? c->wq[tid] * bs : 0.f;
          __syncthreads();
          eval_bwd<NDW>(dm, a, P, c, src, P.k(s), wst, true, acc, bstep, dm.tile_off);
        }
        for (int e = tid; e < nel; e += NTHREADS) {
          const int row = e / NB, col = e % NB, idx = row * LDP + col;
          P.yn[idx] = P.y[idx] + (c->h[col] * (1.0f / 6.0f)) * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        }
        __syncthreads();
        if (!replay && tid < NB && c->status[tid] == 0) c->nfe[tid] += 4;
      }

      if (!replay) {
        // ---- error estimate + control ----------------------------------------------------------------------------
        for (int e = tid; e < nel; e += NTHREADS) {
          const int row = e / NB, col = e % NB, idx = row * LDP + col;
          float r2 = 0.f;
          const float yv = P.y[idx], ynv = P.yn[idx];
          if (o.adaptive) {
            float er = ts5::BT[0] * P.k(0)[idx];
#pragma unroll
            for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * P.k(jj)[idx];
            er *= c->h[col];
            const float sk = o.abstol + fmaxf(fabsf(yv), fabsf(ynv)) * o.reltol;
            const float r = er * fast_rcp(sk);
            r2 = r * r;
          }
          P.scr[idx] = isfinite(ynv) ? r2 : __int_as_float(0x7fc00000);
        }
        __syncthreads();
        if (tid < NB) {
          float s2 = 0.f;
          for (int r = 0; r < NS; r++) s2 += P.scr[r * LDP + tid];
          c->eest[tid] = s2;
        }
        __syncthreads();
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (coupled) {
          if (tid == 0)
            for (int col = 0; col < NB; col++)
              if (c->status[col] == 0) v[0] += c->eest[col];
          grid_sum4(a.gs, gen, v, c->bcast);
        }
        if (tid < NB && c->status[tid] == 0) {
          const int col = tid;
          const float n = coupled ? (float)NS * (float)B : (float)NS;
          const float s2 = coupled ? v[0] : c->eest[col];
          const float EEst = o.adaptive ? sqrtf(s2 / n) : (s2 == s2 ? 0.f : s2);
          const double hmag = c->tnew[col];
          int accepted = 0;
          if (!(EEst == EEst)) {
            if (o.adaptive && hmag > o.dtmin) { c->nrej[col]++; c->dt[col] = hmag * (double)o.qmin; }
            else c->status[col] = 1 + LDE_RET_NONFINITE;
          } else if (o.adaptive) {
            float q11;
            const float q = pi_q(EEst, c->qold[col], o, q11);
            if (EEst > 1.0f) {
              c->nrej[col]++;
              const double nd = hmag * (double)fast_rcp(fminf(o.q_hi, q11 * o.inv_gamma));
              c->dt[col] = nd;
              if (nd < o.dtmin) c->status[col] = 1 + LDE_RET_DTMIN;
            } else {
              c->qold[col] = fmaxf(EEst, 1e-4f);
              double dtp = hmag * (double)fast_rcp(q);
              if (dtp > dtmax) dtp = dtmax;
              c->dt[col] = dtp;
              accepted = 1;
            }
          } else {
            c->dt[col] = o.dt_fixed;
            accepted = 1;
          }
          c->accepted[col] = accepted;
          if (accepted) c->nacc[col]++;
        } else if (tid < NB)
          c->accepted[tid] = 0;
        __syncthreads();
        if (tid == 0) {
          int all = 1, any = 0;
          for (int col = 0; col < NB; col++) {
            if (c->wq[col] != 0.f) { all &= c->accepted[col]; any |= c->accepted[col]; }
          }
          c->all_accepted = all;
          c->any_save = any;
        }
        __syncthreads();
        if (c->all_accepted) {
          if (o.adaptive) commit();            // fixed step: everything is accepted, commit once at the end
        } else {
          discard();
          if (c->any_save) { replay = true; __syncthreads(); continue; }   // redo the stages for the accepted columns only
        }
      } else {
        commit();
        replay = false;
      }
      __syncthreads();

      // ---- advance accepted columns; jump at a save time --------------------------------------------------------
      for (int e = tid; e < nel; e += NTHREADS) {
        const int row = e / NB, col = e % NB, idx = row * LDP + col;
        if (c->accepted[col]) P.y[idx] = P.yn[idx];
      }
      __syncthreads();
      for (int e = tid; e < NB * Dp; e += NTHREADS) {
        const int col = e / Dp, row = e % Dp;
        if (c->accepted[col] && c->hit[col]) {
          const size_t src = (size_t)Dp * ((size_t)(b0 + col) + (size_t)B * c->j[col]) + row;
          P.y[(Dp + row) * LDP + col] += a.dz_out[src];
          if (o.checkpoint) P.y[row * LDP + col] = a.z_out[src];
        }
      }
      __syncthreads();
      if (tid < NB && c->accepted[tid]) {
        const int col = tid;
        if (c->hit[col]) {
          c->t[col] = s_ts[c->j[col]];
          c->j[col]--;
          if (c->j[col] < 0) c->status[col] = 1;
        } else
          c->t[col] -= c->tnew[col];
      }
      __syncthreads();
    }
  }
  if (!o.adaptive) commit();
  __syncthreads();

  // ---- results ----------------------------------------------------------------------------------------------------
  for (int e = tid; e < NB * D; e += NTHREADS) {
    const int col = e / D, row = e % D;
    if (b0 + col < B) a.dz0[(size_t)(b0 + col) * D + row] = c->status[col] > 1 ? 0.f : P.y[(Dp + row) * LDP + col];
  }
  if (NP) {
    for (int e = tid; e < NB * NP; e += NTHREADS) {
      const int col = e / NP, row = e % NP;
      if (b0 + col < B) a.dtheta[(size_t)(b0 + col) * NP + row] = c->status[col] > 1 ? 0.f : P.y[(2 * Dp + row) * LDP + col];
    }
  }
  if (tid < NB && b0 + tid < B) {
    const int col = tid, b = b0 + col;
    a.st_ret[b] = c->status[col] > 1 ? c->status[col] - 1 : 0;
    const bool rep = !coupled || b == 0;
    a.st_nfe[b] = rep ? c->nfe[col] : 0;
    a.st_nacc[b] = rep ? c->nacc[col] : 0;
    a.st_nrej[b] = rep ? c->nrej[col] : 0;
  }
}

// dW[i] += Σ_wg slab[wg][i]   (fixed order ⇒ deterministic)
__global__ void k_reduce_slabs(const float* __restrict__ slab, int nwg, int nW, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nW) return;
  float s = 0.f;
  for (int w = 0; w < nwg; w++) s += slab[(size_t)w * nW + i];
  dW[i] += s;
}

// ================================================ host side =================================================
struct MlpPlan {
  MlpDims dm;
  float* frag = nullptr;
  float* fragT = nullptr;
  size_t nfrag = 0, nfragT = 0;
  unsigned* counter = nullptr;
  float* slots = nullptr;
  int* abort_flag = nullptr;
  int cap_wg = 0;
  float* slab = nullptr;       // [nWG][nW] per-workgroup weight-gradient slabs (adjoint)
  size_t slab_cap = 0;
};

void mlp_plan_destroy(MlpPlan* p);

int mlp_plan_create(const lde_problem_desc& d, MlpPlan** out, std::string& err) {
  MlpPlan* p = new MlpPlan();
  MlpDims& dm = p->dm;
  dm.nL = d.n_layers;
  for (int i = 0; i <= d.n_layers; i++) dm.sizes[i] = d.layer_sizes[i];
  dm.act = d.activation;
  dm.D = d.state_dim;
  dm.Dp = d.state_dim + d.augment_dim;
  dm.P = d.param_dim;
  dm.has_pend = d.rhs_kind == LDE_RHS_PENDULUM_PLUS_MLP;
  dm.coupled = d.batching == LDE_BATCH_COUPLED;
  dm.solver = d.solver;
  int off = 0, offT = 0, woff = 0, hmax = 1;
  for (int l = 0; l < dm.nL; l++) {
    const int in = dm.sizes[l], o = dm.sizes[l + 1];
    dm.frag_off[l] = off;
    off += cdiv(o, 16) * cdiv(in, 4) * 64;
    dm.fragT_off[l] = offT;
    offT += cdiv(in, 16) * cdiv(o, 4) * 64;
    dm.w_off[l] = woff;
    woff += o * in;
    dm.b_off[l] = woff;
    woff += o;
    if (l + 1 < dm.nL && o > hmax) hmax = o;
  }
  dm.hmax = (hmax + 15) & ~15;   // padded to whole 16-row tiles
  dm.nW = woff;
  int blin = 0, toff = 0;
  for (int l = 0; l < dm.nL; l++) {
    dm.bias_lin[l] = blin;
    blin += dm.sizes[l + 1];
    dm.tile_off[l] = toff;
    toff += cdiv(dm.sizes[l + 1], 16) * cdiv(dm.sizes[l], 16);
  }
  dm.tile_off[dm.nL] = toff;
  dm.nbias = blin;
  p->nfrag = off;
  p->nfragT = offT;
  if (dm.Dp > 256 || hmax > 1024) {
    err = "MLP RHS: state_dim+augment_dim ≤ 256 and hidden width ≤ 1024 supported";
    delete p;
    return LDE_ERR_UNSUPPORTED;
  }
  if (hipMalloc(&p->frag, p->nfrag * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->fragT, p->nfragT * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->counter, 64) != hipSuccess || hipMalloc(&p->abort_flag, 64) != hipSuccess) {
    err = "MLP plan: hipMalloc failed";
    mlp_plan_destroy(p);
    return LDE_ERR_ALLOC;
  }
  (void)hipMemset(p->abort_flag, 0, 64);
  *out = p;
  return LDE_OK;
}

void mlp_plan_destroy(MlpPlan* p) {
  if (!p) return;
  if (p->frag) (void)hipFree(p->frag);
  if (p->fragT) (void)hipFree(p->fragT);
  if (p->counter) (void)hipFree(p->counter);
  if (p->abort_flag) (void)hipFree(p->abort_flag);
  if (p->slots) (void)hipFree(p->slots);
  if (p->slab) (void)hipFree(p->slab);
  delete p;
}

int mlp_reserve(MlpPlan* p, int B, int T, std::string& err) {
  const int nwg = cdiv(B, NB);
  if (nwg > p->cap_wg) {
    if (p->slots) (void)hipFree(p->slots);
    p->slots = nullptr;
    if (hipMalloc(&p->slots, (size_t)2 * nwg * 4 * sizeof(float)) != hipSuccess) {
      err = "MLP plan: hipMalloc(slots) failed";
      return LDE_ERR_ALLOC;
    }
    p->cap_wg = nwg;
  }
  const size_t need = (size_t)nwg * (size_t)p->dm.nW;
  if (need > p->slab_cap) {
    if (p->slab) (void)hipFree(p->slab);
    p->slab = nullptr;
    if (hipMalloc(&p->slab, need * sizeof(float)) != hipSuccess) {
      err = "MLP plan: hipMalloc(slab) failed";
      return LDE_ERR_ALLOC;
    }
    p->slab_cap = need;
  }
  (void)T;
  return LDE_OK;
}

int mlp_set_weights(MlpPlan* p, const float* W_dev, hipStream_t stream, std::string& err) {
  hipLaunchKernelGGL(k_build_frags, dim3(64, p->dm.nL), dim3(256), 0, stream, W_dev, p->dm, p->frag, p->fragT);
  if (hipGetLastError() != hipSuccess) {
    err = "k_build_frags launch failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

static size_t fwd_lds_bytes(const MlpDims& dm, int T) {
  size_t b = (sizeof(Ctl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  b += (size_t)(11 * ((dm.Dp + 3) & ~3) + (dm.nL > 1 ? dm.nL - 1 : 0) * dm.hmax) * LDP * sizeof(float);
  return b;
}

int mlp_forward(MlpPlan* p, const float* W_dev, const float* z0, const float* theta, const double* ts_dev,
                const KOpts& o, float* z_out, int32_t* retcode, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                hipStream_t stream, std::string& err) {
  const MlpDims& dm = p->dm;
  const int nwg = cdiv(o.B, NB);
  const bool sync = dm.coupled && o.adaptive && nwg > 1;
  if (sync && nwg > 256) {
    err = "coupled adaptive solve: batch per GPU limited to 4096 trajectories (one resident workgroup per CU)";
    return LDE_ERR_UNSUPPORTED;
  }
  const size_t lds = fwd_lds_bytes(dm, o.T);
  if (lds > 160 * 1024) {
    err = "MLP forward: tile state does not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_mlp_forward, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_forward) failed";
      return LDE_ERR_HIP;
    }
    attr_set = true;
  }
  FwdArgs a;
  a.z0 = z0; a.theta = theta; a.ts = ts_dev; a.frag = p->frag; a.Wflat = W_dev; a.z_out = z_out; a.retcode = retcode;
  a.st_nfe = nfe; a.st_nacc = nacc; a.st_nrej = nrej; a.st_ret = ret;
  a.gs.counter = p->counter; a.gs.slots = p->slots; a.gs.abort_flag = p->abort_flag; a.gs.nwg = sync ? nwg : 1;
  if (sync && hipMemsetAsync(p->counter, 0, sizeof(unsigned), stream) != hipSuccess) {
    err = "hipMemsetAsync(counter) failed";
    return LDE_ERR_HIP;
  }
  hipLaunchKernelGGL(k_mlp_forward, dim3(nwg), dim3(NTHREADS), lds, stream, dm, o, a);
  if (hipGetLastError() != hipSuccess) {
    err = "k_mlp_forward launch failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

static size_t bwd_lds_bytes(const MlpDims& dm, int T) {
  size_t b = (sizeof(Ctl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  const int NSr = (2 * dm.Dp + dm.P + 3) & ~3;
  b += (size_t)(11 * NSr + ((dm.nL > 1 ? dm.nL - 1 : 0) + 2) * dm.hmax) * LDP * sizeof(float);
  b += (size_t)(((dm.nbias + 3) & ~3) + NB) * sizeof(float);
  return b;
}

template <int NDW>
static int launch_adjoint(MlpPlan* p, const KOpts& o, const BwdArgs& a, int nwg, size_t lds, hipStream_t stream,
                          std::string& err) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_mlp_adjoint<NDW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_adjoint) failed";
      return LDE_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((k_mlp_adjoint<NDW>), dim3(nwg), dim3(NTHREADS), lds, stream, p->dm, o, a);
  return LDE_OK;
}

int mlp_adjoint(MlpPlan* p, const float* W_dev, const float* z_out, const float* theta, const double* ts_dev,
                const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* dW, int32_t* nfe, int32_t* nacc,
                int32_t* nrej, int32_t* ret, hipStream_t stream, std::string& err) {
  const MlpDims& dm = p->dm;
  const int nwg = cdiv(o.B, NB);
  const bool sync = dm.coupled && o.adaptive && nwg > 1;
  if (sync && nwg > 256) {
    err = "coupled adaptive solve: batch per GPU limited to 4096 trajectories (one resident workgroup per CU)";
    return LDE_ERR_UNSUPPORTED;
  }
  const size_t lds = bwd_lds_bytes(dm, o.T);
  if (lds > 160 * 1024) {
    err = "MLP adjoint: tile state does not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  BwdArgs a;
  a.z_out = z_out; a.dz_out = dz_out; a.theta = theta; a.ts = ts_dev; a.frag = p->frag; a.fragT = p->fragT; a.Wflat = W_dev;
  a.dz0 = dz0; a.dtheta = dtheta; a.slab = p->slab;
  a.st_nfe = nfe; a.st_nacc = nacc; a.st_nrej = nrej; a.st_ret = ret;
  a.gs.counter = p->counter; a.gs.slots = p->slots; a.gs.abort_flag = p->abort_flag; a.gs.nwg = sync ? nwg : 1;
  if (sync && hipMemsetAsync(p->counter, 0, sizeof(unsigned), stream) != hipSuccess) {
    err = "hipMemsetAsync(counter) failed";
    return LDE_ERR_HIP;
  }
  const int per_wave = cdiv(dm.tile_off[dm.nL], 4);   // weight-gradient tiles held in each wave's accumulators
  int rc;
  if (per_wave <= 8) rc = launch_adjoint<8>(p, o, a, nwg, lds, stream, err);
  else if (per_wave <= 24) rc = launch_adjoint<24>(p, o, a, nwg, lds, stream, err);
  else if (per_wave <= 52) rc = launch_adjoint<52>(p, o, a, nwg, lds, stream, err);
  else {
    err = "MLP adjoint: more than 208 16x16 weight-gradient tiles (hidden width too large for the register-resident accumulators)";
    return LDE_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (hipGetLastError() != hipSuccess) {
    err = "k_mlp_adjoint launch failed";
    return LDE_ERR_HIP;
  }
  hipLaunchKernelGGL(k_reduce_slabs, dim3(cdiv(dm.nW, 256)), dim3(256), 0, stream, p->slab, nwg, dm.nW, dW);
  if (hipGetLastError() != hipSuccess) {
    err = "k_reduce_slabs launch failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

}  // namespace lde

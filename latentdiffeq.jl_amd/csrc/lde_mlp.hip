// lde_mlp.hip — right-hand sides with a Dense/relu MLP: the LatentODE path and "physics + MLP".
//
// Replaces what runs under
//     nODE = NeuralODE(dudt, (t[1], t[end]), solver; saveat = t, kwargs...); ẑ = Array(nODE(ẑ₀))
//                                                                   [REF src/models/LatentODE.jl:70-72]
// with dudt = Chain(Dense(D', H, relu), Dense(H, H, relu), Dense(H, D'))     [REF examples/pendulum_friction-less/nODE.jl:12-14]
// (DiffEqFlux 1.52.0 / OrdinaryDiffEq 6.27.1 / SciMLSensitivity 7.10.0, un-vendored) and its pullback.
//
// Design (gfx950)
//  * A workgroup of 8 waves (two per SIMD: one wave's LDS / barrier waits are covered by the other; 4 waves only when
//    the adjoint needs more than 8 weight-gradient tiles per wave) owns a tile of NB = 16 trajectories (columns) for the
//    WHOLE solve. State, the seven Tsit5 slopes and the hidden activations of the tile live in LDS as TRANSPOSED
//    panels Xt[col][row] (row contiguous) with a stride ≡ 8 (mod 32) floats, which makes the 16-byte MFMA operand
//    reads below bank-conflict-free.
//  * Every Dense layer is Y[out×16] = W[out×in]·X[in×16] on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32,
//    64 FLOP/clk/SIMD — gfx950 has no xf32, and bf16 would break the 1e-4 tolerance). The K index is permuted so that
//    one ds_read_b128 per lane feeds FOUR consecutive MFMAs on each side: in K-group kg, MFMA s contracts
//    k = 16·kg + 4·(lane>>4) + s. Weights are re-laid out ONCE per lde_set_weights into that fragment order
//    (for W and for Wᵀ), and copied into LDS at kernel start as far as they fit (the rest streams from L2).
//    Two accumulators per wave are always in flight (two row tiles, or the even/odd K-groups of one), narrow
//    layers (fewer than 3 row tiles) are split along K across the waves and reduced through LDS. Bias + activation
//    are fused into the epilogue, which writes the 4 consecutive rows a lane owns with one ds_write_b128.
//  * Per-trajectory step control (GOKU semantics): all 16 columns run the same stage of their own step
//    (own t, dt, accept/reject) in lock-step; a finished column idles with h = 0.
//  * Coupled control (NeuralODE semantics: one dt, RMS norm over all D'·B entries): one grid-wide sum per step —
//    each workgroup publishes its partial, a monotonic-counter barrier (agent-scope release/acquire) follows, and
//    every workgroup adds the partials in the same order ⇒ bitwise identical decisions everywhere.
//  * Adjoint: reverse-time Tsit5/RK4 on [z; λ; g_θ] with the MLP re-evaluated at every stage (relu masks are
//    recomputed, not stored) and vector-Jacobian products through Wᵀ fragments. The weight gradient is a quadrature
//    over the accepted steps, gW_l = Σ_stages Σ_n w_n δ_l[:,n] a_l[:,n]ᵀ, and does not feed back into the solve, so
//    the solve kernel only STAGES the (a_l, δ_l) panels of every weighted stage in HBM (a few coalesced 16-byte
//    stores per lane) with the per-column quadrature weights next to them (weights of rejected columns are zeroed
//    afterwards, slots of wholly rejected attempts are reused). A second kernel, k_mlp_dw, then forms the gradient as
//    ONE large-K product per 32×32 tile of W_lᵀ over all staged columns, spread over the whole chip (tile × K-split ×
//    layer jobs) on v_mfma_f32_32x32x2_f32, and k_reduce_tiles adds the partial slabs in a fixed order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include "lde_device.h"
#include "lde_mfma.h"

namespace lde {

__device__ __forceinline__ float act_fn(int kind, float x) { return kind == LDE_ACT_TANH ? tanhf(x) : fmaxf(x, 0.f); }
__device__ __forceinline__ float act_grad(int kind, float a) { return kind == LDE_ACT_TANH ? 1.f - a * a : (a > 0.f ? 1.f : 0.f); }

// One wave, one or two 16-row tiles, K-groups [k0,k1): software-pipelined through a PF-deep register ring (operands of
// group k+PF are requested before the MFMAs of group k issue; indices past the end are clamped, not branched, so the
// loop body stays straight-line and the s_waitcnt counters exact). Even/odd groups go to two accumulators.
__device__ __forceinline__ void mfma4(const f32x4& a, const f32x4& b, f32x4& acc) {
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) acc = mfma16(a[s4], b[s4], acc);
}

template <int PF>
__device__ __forceinline__ void mac1(const f32x4* A, const float* xp, int k0, int k1, f32x4& acc0, f32x4& acc1) {
  if (k0 >= k1) return;
  f32x4 ra[PF], rb[PF];
  const int kl = k1 - 1;
#pragma unroll
  for (int i = 0; i < PF; i++) {
    const int k = min(k0 + i, kl);
    ra[i] = A[k * 64];
    rb[i] = *reinterpret_cast<const f32x4*>(xp + k * 16);
  }
  int kg = k0;
  for (; kg + PF <= k1; kg += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const f32x4 ca = ra[i], cb = rb[i];
      const int k = min(kg + i + PF, kl);
      ra[i] = A[k * 64];
      rb[i] = *reinterpret_cast<const f32x4*>(xp + k * 16);
      mfma4(ca, cb, (i & 1) ? acc1 : acc0);
    }
  }
  const int r = k1 - kg;
#pragma unroll
  for (int i = 0; i < PF - 1; i++)
    if (i < r) mfma4(ra[i], rb[i], (i & 1) ? acc1 : acc0);
}

template <int PF>
__device__ __forceinline__ void mac2(const f32x4* A0, const f32x4* A1, const float* xp, int k1, f32x4& acc0, f32x4& acc1) {
  f32x4 r0[PF], r1[PF], rb[PF];
  const int kl = k1 - 1;
#pragma unroll
  for (int i = 0; i < PF; i++) {
    const int k = min(i, kl);
    r0[i] = A0[k * 64];
    r1[i] = A1[k * 64];
    rb[i] = *reinterpret_cast<const f32x4*>(xp + k * 16);
  }
  int kg = 0;
  for (; kg + PF <= k1; kg += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const f32x4 c0 = r0[i], c1 = r1[i], cb = rb[i];
      const int k = min(kg + i + PF, kl);
      r0[i] = A0[k * 64];
      r1[i] = A1[k * 64];
      rb[i] = *reinterpret_cast<const f32x4*>(xp + k * 16);
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        acc0 = mfma16(c0[s4], cb[s4], acc0);
        acc1 = mfma16(c1[s4], cb[s4], acc1);
      }
    }
  }
  const int r = k1 - kg;
#pragma unroll
  for (int i = 0; i < PF - 1; i++)
    if (i < r) {
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        acc0 = mfma16(r0[i][s4], rb[i][s4], acc0);
        acc1 = mfma16(r1[i][s4], rb[i][s4], acc1);
      }
    }
}

// Y[R×16] = M[R×K]·X[K×16] for one workgroup. M as K4 fragments (LDS copy, or global/L2 when GLB), X a transposed panel
// (Xt[col*ldx + row], rows [0,K) starting at the pointer), EPI(row0, col, acc4) gets the 4 consecutive rows a lane owns.
// `red` = (NT/64)·1 KiB LDS scratch for the split-K reduction.
//   more row tiles than waves : a wave takes tiles rt, rt + NW (two accumulators = the two tiles), then leftovers singly
//   3 ≤ tiles ≤ NW            : one tile per wave (two accumulators = even/odd K-groups)
//   1–2 tiles                 : K split over NW/tiles waves per tile, partial sums reduced through LDS
template <int NT, bool GLB, class Epi>
__device__ __forceinline__ void panel_gemm(const float* frag, int R, int K, const float* Xt, int ldx, float* red, Epi epi) {
  if (LDE_ABL == 1) return;   // diagnostic build: no GEMM at all
  constexpr int NW = NT / 64;
  constexpr int PF1 = GLB ? 6 : 2, PF2 = GLB ? 4 : 2;   // L2 latency ≈ 5 K-groups of MFMA time; LDS ≈ 1
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform ⇒ scalar loop control below
  const int RT = cdiv(R, 16), KG = cdiv(K, 16);
  const f32x4* A = reinterpret_cast<const f32x4*>(frag) + lane;
  const float* xp = Xt + (lane & 15) * ldx + 4 * (lane >> 4);
  const int col = lane & 15, rsub = 4 * (lane >> 4);
  if (RT >= 3 || 2 * RT > NW) {   // (splitting K for 3–4 row tiles was measured slower: the reduction's extra barrier)
    for (int rt = wave; rt < RT; rt += 2 * NW) {
      const int rt2 = rt + NW;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (rt2 < RT) {
        mac2<PF2>(A + (size_t)rt * KG * 64, A + (size_t)rt2 * KG * 64, xp, KG, acc0, acc1);
        if (LDE_ABL != 3) epi(rt * 16 + rsub, col, acc0);
        if (LDE_ABL != 3) epi(rt2 * 16 + rsub, col, acc1);
      } else {
        mac1<PF1>(A + (size_t)rt * KG * 64, xp, 0, KG, acc0, acc1);
        if (LDE_ABL != 3) epi(rt * 16 + rsub, col, acc0 + acc1);
      }
    }
  } else {
    const int nparts = NW / RT;
    const int rt = wave % RT, part = wave / RT;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (part < nparts) {
      const int per = cdiv(KG, nparts), k0 = part * per, k1 = min(KG, k0 + per);
      mac1<PF1>(A + (size_t)rt * KG * 64, xp, k0, k1, acc0, acc1);
    }
    f32x4* rp = reinterpret_cast<f32x4*>(red);
    rp[wave * 64 + lane] = acc0 + acc1;
    __syncthreads();
    if (part == 0) {
      f32x4 t = rp[rt * 64 + lane];
      for (int pp = 1; pp < nparts; pp++) t += rp[(rt + pp * RT) * 64 + lane];
      if (LDE_ABL != 3) epi(rt * 16 + rsub, col, t);
    }
  }
}

// A-fragments either from the LDS cache (pointer formed from the LDS base ⇒ ds_read_b128) or from global/L2
// (kernel-argument pointer ⇒ global_load_dwordx4) — never through a generic pointer.
template <int NT, class Epi>
__device__ __forceinline__ void layer_gemm(const float* lds_base, int ofs, const float* gfrag, int R, int K, const float* Xt,
                                           int ldx, float* red, Epi epi) {
  if (ofs >= 0) panel_gemm<NT, false>(lds_base + ofs, R, K, Xt, ldx, red, epi);
  else panel_gemm<NT, true>(gfrag, R, K, Xt, ldx, red, epi);
}

// ---- grid-wide deterministic sum (coupled mode) ----------------------------------------------------
constexpr int LDE_MAX_PEERS = 8;   // ranks of a device-side cross-rank sum: one node
struct GridSync {
  unsigned* counter;   // monotonic arrival counter, zeroed before the launch
  float* slots;        // [2][nWG][4] partials (ping-pong by generation parity)
  int* abort_flag;     // set when a spin times out
  int nwg;
  // LDE_BATCH_COUPLED_GLOBAL (the batch is sharded over ranks, the norm is over ALL ranks' columns): k_mlpw relays every device-wide
  // sum through the host, which adds the other ranks' (lde_set_global_sum_hook). Mailbox words = {value bits, tag << 32} in pinned,
  // coherent host memory: req[0..1] values, req[2] = {count, tag} written last; rep[0..1] the host's answer; dev_rep[0..1] the same
  // words republished in device memory by workgroup 0 for the other workgroups. nullptr: off.
  unsigned long long* host_req;
  unsigned long long* host_rep;
  unsigned long long* dev_rep;
  // … or, without the host (lde_set_global_sum_peers): every rank's mailbox is mapped on every device — peer[r] = rank r's, [2 parities][ranks][2
  // words] — and workgroup 0 writes this rank's words into slot `rank` of all of them, then adds the `nranks` words of its own in rank order
  // (the same bits on every rank). nranks = 0: off.
  unsigned long long* peer[LDE_MAX_PEERS];
  int rank, nranks;
  unsigned xlaunch;    // the peer exchange's OWN launch counter (16 bits): +1 per launch that exchanges, never skipped, never reset — the mailbox
                       // words' tag and parity come from it, not from the grid words' epoch (which skips 0 at its wrap and restarts when the
                       // words are reallocated: ranks whose local batch grows at different calls would fall out of step)
  int xspin_k;         // cross-rank wait: give up (poison the sums) after xspin_k·1024 polls of ≈ 1 µs (option "peer_spin_k", default 8192 ≈ 10 s)
  __host__ __device__ bool cross() const { return host_req != nullptr || nranks > 0; }   // the sums leave the device
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
    bool aborted = __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // sticky
    while (!aborted && __hip_atomic_load(gs.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000000LL) {  // ≈ seconds: a peer is not resident — give up instead of hanging the GPU
        __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        aborted = true;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < gs.nwg; w++) {  // fixed order ⇒ identical totals in every workgroup
#pragma unroll
      for (int i = 0; i < 4; i++) t[i] += slots[(size_t)w * 4 + i];
    }
    // a timed-out barrier poisons the sums: every step is then "non-finite", dt shrinks to dtmin and the solve ends with
    // retcode != 0 and NaN blocks instead of silently wrong numbers (later calls return at once: the flag is sticky)
#pragma unroll
    for (int i = 0; i < 4; i++) s_bcast[i] = aborted ? __int_as_float(0x7fc00000) : t[i];
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
  int wofs[MAXL];    // ≥0: float offset (from the LDS base) of the cached copy of layer l's W fragments; <0: not cached
  int wTofs[MAXL];   // same for Wᵀ  (offsets, not pointers: a pointer read back from memory is "flat" to the compiler and
                     //  every flat load drains vmcnt AND lgkmcnt — measured: ~50 % of all wave cycles parked)
};

struct Panels {       // LDS carve-up; every panel is transposed: element (row, col) at col*ld + row
  float* y;
  float* yn;
  float* tmp;
  float* kbase;       // k[s] = kbase + s*pstride
  float* hidbase;     // hidden activations (post-activation) of layers 0..nL-2
  float* delbase;     // backprop panels (adjoint only)
  float* scr;         // scratch (error terms, 1/scale)
  float* red;         // (NT / 64)*256 floats: split-K reduction scratch
  float* biasc;       // LDS copy of all biases (compact, bias_lin order)
  int pstride, hstride;
  int lds, ldh;       // strides of state / hidden panels
  const float* lbase; // LDS base (float view of the dynamic shared array)
  const float* gfrag; // global fragment arrays
  const float* gfragT;
  __device__ __forceinline__ float* k(int s) const { return kbase + s * pstride; }
  const int* hoff;    // dm.h_off
  __device__ __forceinline__ float* hid(int l) const { return hidbase + hoff[l]; }
  __device__ __forceinline__ float* del(int i) const { return delbase + i * hstride; }
};

// copy weight fragments into the LDS cache while they fit; record where each layer's fragments live
template <int NT>
__device__ __forceinline__ float* cache_frags(const MlpDims& dm, const float* gfrag, const int* off, const int* cnt,
                                              int* table, const float* lds_base, float* cache, float* cache_end) {
  for (int l = 0; l < dm.nL; l++) {
    const float* src = gfrag + off[l];
    const int n = cnt[l];
    const bool fits = cache + n <= cache_end;
    if (fits) {
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
      f32x4* d4 = reinterpret_cast<f32x4*>(cache);
      for (int i = threadIdx.x; i < n / 4; i += NT) d4[i] = s4[i];
    }
    if (threadIdx.x == 0) table[l] = fits ? (int)(cache - lds_base) : -1;
    if (fits) cache += n;
  }
  return cache;
}

// f(z) for the tile: dst rows [0,Dp) = MLP(src rows [0,Dp)) (+ pendulum); hidden activations are left in P.hid(*).
// (A variant with the layer loop unrolled at compile time was measured: −8 % time for 3× code and build time — dropped.)
template <int NT>
__device__ __forceinline__ void eval_rhs(const MlpDims& dm, const Panels& P, const Ctl* c, const float* src, float* dst) {
  const float* X = src;
  int ldx = P.lds;
  const int nL = dm.nL;
  for (int l = 0; l < nL; l++) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const bool lastl = l == nL - 1;
    float* Y = lastl ? dst : P.hid(l);
    const int ldy = lastl ? P.lds : dm.ld_hl[l];
    const float* bias = P.biasc + dm.bias_lin[l];
    const int actk = dm.act;
    PROF_T(pt0);
    layer_gemm<NT>(P.lbase, c->wofs[l], P.gfrag + dm.frag_off[l], out, in, X, ldx, P.red, [&](int row0, int col, f32x4 v) {
      f32x4 r;
      if (row0 + 3 < out) {   // whole group of 4 rows inside the layer: vector bias load, no per-row selects
        r = v + *reinterpret_cast<const f32x4*>(bias + row0);
        if (!lastl) {
#pragma unroll
          for (int q = 0; q < 4; q++) r[q] = act_fn(actk, r[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float x = row0 + q < out ? v[q] + bias[row0 + q] : 0.f;
          if (!lastl) x = act_fn(actk, x);
          r[q] = x;
        }
      }
      if (lastl && row0 + 3 >= out) {  // never write the rows that follow the output block (λ rows in the adjoint state)
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (row0 + q < out) Y[col * ldy + row0 + q] = r[q];
      } else
        *reinterpret_cast<f32x4*>(Y + col * ldy + row0) = r;
    });
    PROF_T(pt1);
    __syncthreads();
    PROF_T(pt2);
    PROF_ADD(2 + 2 * l, pt0, pt1);
    PROF_ADD(3 + 2 * l, pt1, pt2);
    X = Y;
    ldx = ldy;
  }
  if (dm.has_pend) {
    if (threadIdx.x < NB) {
      const int col = threadIdx.x;
      const float x = src[col * P.lds + 0], yv = src[col * P.lds + 1];
      dst[col * P.lds + 0] += yv;
      dst[col * P.lds + 1] += c->ngl[col] * fast_sin(x);
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
  int lds_bytes;      // dynamic LDS actually requested (the weight cache takes what the panels leave)
};

template <int NT>
__device__ __forceinline__ void load_biases(const MlpDims& dm, const float* Wflat, float* biasc) {
  for (int l = 0; l < dm.nL; l++)
    for (int i = threadIdx.x; i < dm.sizes[l + 1]; i += NT) biasc[dm.bias_lin[l] + i] = Wflat[dm.b_off[l] + i];
}

template <int SOLVER, int NT>
__global__ void __launch_bounds__(NT) k_mlp_forward(MlpDims dm, KOpts o, FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = o.T, B = o.B, Dp = dm.Dp, D = dm.D;
  // ---- carve LDS -------------------------------------------------------------------------------------
  Ctl* c = reinterpret_cast<Ctl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(Ctl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  Panels P;
  const int NS = Dp;
  P.lds = dm.ld_sf;
  P.ldh = dm.ld_h;
  P.pstride = NB * P.lds;
  P.hstride = NB * P.ldh;
  float* p = base;
  P.y = p; p += P.pstride;
  P.yn = p; p += P.pstride;
  P.tmp = p; p += P.pstride;
  P.kbase = p; p += 7 * P.pstride;
  P.scr = p; p += P.pstride;
  P.hidbase = p; p += dm.h_total;
  P.hoff = dm.h_off;
  P.delbase = nullptr;
  P.red = p; p += (NT / 64) * 256;
  P.biasc = p; p += (dm.nbias + 3) & ~3;
  const int nfloat = (int)(p - base);
  for (int i = threadIdx.x; i < nfloat; i += NT) base[i] = 0.f;   // pad rows must be finite (0·x)
  for (int i = threadIdx.x; i < T; i += NT) s_ts[i] = a.ts[i];
  __syncthreads();
  P.lbase = reinterpret_cast<const float*>(smem);
  P.gfrag = a.frag;
  P.gfragT = nullptr;
  load_biases<NT>(dm, a.Wflat, P.biasc);
  cache_frags<NT>(dm, a.frag, dm.frag_off, dm.frag_n, c->wofs, P.lbase, p, reinterpret_cast<float*>(smem + a.lds_bytes));
  __syncthreads();

  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * NB;
  const int nel = NS * NB;                      // elements of a state panel handled cooperatively
  const int lds = P.lds;
  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = tend - t0;
  unsigned gen = 0;
// all NS×16 elements of a state panel: 32 row-lanes × NT/32 columns per pass (shifts only; conflict-free b32 accesses)
#define FOR_ELEMS(idx, colv)                            \
  for (int colv = tid >> 5; colv < NB; colv += NT / 32) \
    for (int r_ = tid & 31, idx = colv * lds + r_; r_ < NS; r_ += 32, idx += 32)

  // ---- load the tile: column-major z0 [D×B]; augmented rows stay 0 -----------------------------------
  for (int e = tid; e < NB * D; e += NT) {
    const int col = e / D, row = e % D;
    if (b0 + col < B) P.y[col * lds + row] = a.z0[(size_t)(b0 + col) * D + row];
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
  for (int e = tid; e < NB * Dp; e += NT) {
    const int col = e / Dp, row = e % Dp;
    if (b0 + col < B) a.z_out[(size_t)(b0 + col) * Dp + row] = P.y[col * lds + row];
  }

  // ---- the solve as a phase machine: the MLP evaluation below is the ONLY instantiation of eval_rhs in this kernel
  //      (five inlined copies of the GEMM code made the hot loop several times larger than the instruction cache)
  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;   // Tsit5: k2..k7 ; RK4: k2,k3,k4 and f(y_new)

  // start of a step: iteration guard, clip the step to t_end, per-column h; false when every column has finished
  auto begin_step = [&]() -> bool {
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
    return c->any_active != 0;   // coupled: identical control arithmetic ⇒ every workgroup leaves together
  };

  int phase = PH_K0, s = 0;
  bool running = T > 1;
  while (running) {
    PROF_T(pl0);
    // ---- input of this evaluation ---------------------------------------------------------------------------
    const float* src = P.y;
    if (phase == PH_INIT1) src = P.tmp;
    if (phase == PH_STAGE) {
      if (SOLVER == LDE_SOLVER_TSIT5) {
        float* dstp = s < 6 ? P.tmp : P.yn;
        FOR_ELEMS(idx, ecol) {
          float acc = ts5::A[s][0] * P.k(0)[idx];
          for (int jj = 1; jj < s; jj++) acc += ts5::A[s][jj] * P.k(jj)[idx];
          dstp[idx] = P.y[idx] + c->h[ecol] * acc;
        }
        src = dstp;
      } else if (s < 4) {
        const float cs = s == 3 ? 1.0f : 0.5f;
        FOR_ELEMS(idx, ecol) {
          P.tmp[idx] = P.y[idx] + (cs * c->h[ecol]) * P.k(s - 1)[idx];
        }
        src = P.tmp;
      } else {
        FOR_ELEMS(idx, ecol) {
          P.yn[idx] = P.y[idx] + (c->h[ecol] * (1.0f / 6.0f)) * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        }
        src = P.yn;
      }
      __syncthreads();
    }
    float* dst = phase == PH_K0 ? P.k(0) : (phase == PH_INIT1 ? P.k(1) : P.k(s));

    PROF_T(pl1);
    PROF_ADD(0, pl0, pl1);
    eval_rhs<NT>(dm, P, c, src, dst);
    if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;
    PROF_T(pl2);
    PROF_ADD(40, pl1, pl2);
#if LDE_PROF
    struct ProfEnd { long long t0; __device__ ~ProfEnd() { PROF_T(t1); PROF_ADD(1, t0, t1); } } prof_end{pl2};
#endif

    // ---- what follows the evaluation ---------------------------------------------------------------------------
    if (phase == PH_K0) {
      if (o.adaptive && !(o.dt_fixed > 0)) {
        // Hairer–Nørsett–Wanner, part 1: d0, d1, trial Euler step
        FOR_ELEMS(idx, ecol) {
          const float yv = P.y[idx];
          const float sk = fast_rcp(o.abstol + fabsf(yv) * o.reltol);
          P.scr[idx] = sk;
          const float a0 = yv * sk, a1 = P.k(0)[idx] * sk;
          P.tmp[idx] = a0 * a0;
          P.yn[idx] = a1 * a1;
        }
        __syncthreads();
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < NB) {
          float s0 = 0.f, s1 = 0.f;
          for (int r = 0; r < NS; r++) { s0 += P.tmp[tid * lds + r]; s1 += P.yn[tid * lds + r]; }
          c->eest[tid] = s0;   // Σ (y/sk)²
          c->wq[tid] = s1;     // Σ (f0/sk)²
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
        FOR_ELEMS(idx, ecol) {
          P.tmp[idx] = P.y[idx] + c->h[ecol] * P.k(0)[idx];
        }
        __syncthreads();
        phase = PH_INIT1;
      } else {
        if (tid < NB) c->dt[tid] = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
        __syncthreads();
        phase = PH_STAGE;
        s = 1;
        running = begin_step();
      }
    } else if (phase == PH_INIT1) {
      // part 2: d2 and the initial step
      FOR_ELEMS(idx, ecol) {
        const float d = (P.k(1)[idx] - P.k(0)[idx]) * P.scr[idx];
        P.yn[idx] = d * d;
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.yn[tid * lds + r];
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
      phase = PH_STAGE;
      s = 1;
      running = begin_step();
    } else if (s < LAST_STAGE) {
      s++;
    } else {
      // ---- end of a step attempt: error estimate ------------------------------------------------------------
      FOR_ELEMS(idx, ecol) {
        float r2 = 0.f;
        const float yv = P.y[idx], ynv = P.yn[idx];
        if (o.adaptive) {
          float er = ts5::BT[0] * P.k(0)[idx];
#pragma unroll
          for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * P.k(jj)[idx];
          er *= c->h[ecol];
          const float sk = o.abstol + fmaxf(fabsf(yv), fabsf(ynv)) * o.reltol;
          const float r = er * fast_rcp(sk);
          r2 = r * r;
        }
        P.scr[idx] = isfinite(ynv) ? r2 : __int_as_float(0x7fc00000);   // non-finite state poisons the column's sum
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.scr[tid * lds + r];
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
        if (accepted) {
          if (o.rec.n && c->nacc[col] < o.rec.cap && (!coupled || b0 + col == 0)) {   // the step record: start time and size (a coupled solve has ONE sequence)
            const size_t ri = (size_t)c->nacc[col] * o.rec.nseq + (coupled ? 0 : b0 + col);
            o.rec.t[ri] = c->t[col];
            o.rec.dt[ri] = dt;
          }
          c->nacc[col]++;
        }
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
        for (int e = tid; e < NB * Dp; e += NT) {
          const int col = e / Dp, row = e % Dp, idx = col * lds + row;
          if (!c->hit[col]) continue;
          const float th = c->th[col], h = c->wq[col];
          float out;
          if (th > 1.5f) out = P.yn[idx];
          else if (SOLVER == LDE_SOLVER_TSIT5) {
            float bw[7];
            tsit5_interp_weights(th, bw);
            float acc = bw[0] * P.k(0)[idx];
#pragma unroll
            for (int q = 1; q < 7; q++) acc += bw[q] * P.k(q)[idx];
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

      // ---- advance accepted columns (FSAL: the last slope becomes k1) --------------------------------------
      FOR_ELEMS(idx, ecol) {
        if (c->accepted[ecol]) {
          if (o.rec.n && c->nacc[ecol] <= o.rec.cap && b0 + ecol < B)   // … and the state the step started from
            o.rec.y[((size_t)(c->nacc[ecol] - 1) * B + (b0 + ecol)) * Dp + (idx - ecol * lds)] = P.y[idx];
          P.y[idx] = P.yn[idx];
          P.k(0)[idx] = P.k(LAST_STAGE)[idx];
        }
      }
      if (tid < NB && c->accepted[tid]) {
        c->t[tid] = c->tnew[tid];
        if (c->last[tid]) c->status[tid] = 1;
      }
      __syncthreads();
      s = 1;
      running = begin_step();
    }
  }

  // ---- epilogue: NaN blocks for failed columns, statistics -------------------------------------------------
  for (int e = tid; e < NB * Dp; e += NT) {
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
    if (o.rec.n && rep) o.rec.n[coupled ? 0 : b] = ret == 0 ? c->nacc[col] : 0;
  }
#undef FOR_ELEMS
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
  float* slab;        // [nWG][slab_n] private slabs, written only by a workgroup whose staging area overflowed
  float* stage;       // [nWG][cap][blk_floats] staged (a_l, δ_l) panels of every weighted stage evaluation
  float* wts;         // [nWG][cap][16] quadrature weight of each staged column (0 ⇒ rejected / idle)
  int32_t* nslots;    // [nWG] staged evaluations that count
  int32_t* nflush;    // [nWG] times the workgroup had to fold its staging area into its private slab
  int cap;            // staging slots per workgroup
  int32_t* ovf;       // set to 1 by the 4-columns-per-wave kernel when a wave ran out of staging slots
  int fallback;       // k_mlp_adjoint launched behind that kernel: run only if *ovf != 0 (then redo everything)
  int32_t *st_nfe, *st_nacc, *st_nrej, *st_ret;
  GridSync gs;
  int lds_bytes;
};

// f, −(∂f/∂z)ᵀλ, −(∂f/∂θ)ᵀλ for the tile; when `blk` is given, the (a_l, δ_l) panels of this evaluation are staged there.
//   src/dst rows: [0,Dp) z | [DpA,DpA+Dp) λ | [2DpA,2DpA+P) g.
template <int NT>
__device__ __forceinline__ void eval_bwd(const MlpDims& dm, const Panels& P, const Ctl* c, const float* src, float* dst,
                                         float* blk) {
  static_assert(NT == 512, "the staging copy maps one half-wave to each of the 16 columns");
  const int DpA = dm.DpA, nL = dm.nL;
  // 1. forward through the MLP (relu masks are recomputed here, not stored by the forward solve)
  eval_rhs<NT>(dm, P, c, src, dst);
  // 2. back-propagate λ; δ_L = λ_stage
  const float* dl = src + DpA;
  int ldd = P.lds;
  for (int l = nL - 1; l >= 0; l--) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const float* al = l == 0 ? src : P.hid(l - 1);   // input activation of layer l
    const int lda = l == 0 ? P.lds : dm.ld_hl[l - 1];
    PROF_T(ps0);
    if (blk && LDE_ABL != 4) {
      // stage a_l rows [0,in32) and δ_l rows [0,out32) of the 16 columns: one half-wave per column, 512 contiguous bytes
      // per store instruction. Pad rows are whatever follows in the panel (finite): they only reach pad rows of gWᵀ tiles.
      const int in32 = pad32(in), out32 = pad32(out);
      float* ga = blk + dm.blk_off[l];
      float* gd = ga + NB * in32;
      const int col = threadIdx.x >> 5, l31 = threadIdx.x & 31;
      for (int r4 = l31; 4 * r4 < in32; r4 += 32)
        *reinterpret_cast<f32x4*>(ga + col * in32 + 4 * r4) = *reinterpret_cast<const f32x4*>(al + col * lda + 4 * r4);
      for (int r4 = l31; 4 * r4 < out32; r4 += 32)
        *reinterpret_cast<f32x4*>(gd + col * out32 + 4 * r4) = *reinterpret_cast<const f32x4*>(dl + col * ldd + 4 * r4);
    }
    PROF_T(ps1);
    PROF_ADD(30, ps0, ps1);
    // δ_in = W_lᵀ δ  (⊙ act'(a_l) for hidden layers); layer 0 gives (∂f/∂z)ᵀλ
    if (l > 0) {
      float* dn = P.del((nL - 1 - l) & 1);
      const int actk = dm.act, ldh = dm.ld_hl[l - 1];   // δ_{l-1} shares the geometry of the activation it masks
      layer_gemm<NT>(P.lbase, c->wTofs[l], P.gfragT + dm.fragT_off[l], in, out, dl, ldd, P.red, [&](int row0, int col, f32x4 v) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(al + col * ldh + row0);
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; q++) r[q] = row0 + q < in ? v[q] * act_grad(actk, av[q]) : 0.f;
        *reinterpret_cast<f32x4*>(dn + col * ldh + row0) = r;
      });
      PROF_T(ps2);
      __syncthreads();
      PROF_T(ps3);
      PROF_ADD(16 + 2 * l, ps1, ps2);
      PROF_ADD(17 + 2 * l, ps2, ps3);
      dl = dn;
      ldd = ldh;
    } else {
      float* dlam = dst + DpA;
      const int lds = P.lds;
      layer_gemm<NT>(P.lbase, c->wTofs[0], P.gfragT + dm.fragT_off[0], in, out, dl, ldd, P.red, [&](int row0, int col, f32x4 v) {
        if (row0 + 3 < in) {
          *reinterpret_cast<f32x4*>(dlam + col * lds + row0) = -v;
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (row0 + q < in) dlam[col * lds + row0 + q] = -v[q];
        }
      });
      PROF_T(ps2);
      __syncthreads();
      PROF_T(ps3);
      PROF_ADD(16, ps1, ps2);
      PROF_ADD(17, ps2, ps3);
    }
  }
  // 3. known-physics part: J = [[0,1],[ngl·cos x, 0]], ∂f₂/∂L = gl2·sin x
  if (dm.has_pend) {
    if (threadIdx.x < NB) {
      const int col = threadIdx.x, o_ = col * P.lds;
      float sn, cs;
      fast_sincos(src[o_ + 0], sn, cs);
      const float l0 = src[o_ + DpA + 0], l1 = src[o_ + DpA + 1];
      // eval_rhs already added the pendulum to f (rows 0,1)
      dst[o_ + DpA + 0] -= c->ngl[col] * cs * l1;
      dst[o_ + DpA + 1] -= l0;
      dst[o_ + 2 * DpA] = -(c->gl2[col] * sn * l1);
    }
    __syncthreads();
  }
}

// Slow path, taken only when a workgroup runs out of staging slots: gW contributions of its slots [0, ns) are added to
// its private slab (same fragment order as k_mlp_dw writes). One 32×32 tile per wave at a time, operands straight from
// L2 with agent-scope loads (the blocks were written by this workgroup's own stores a moment ago).
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NT>
__device__ __forceinline__ void flush_stage(const MlpDims& dm, const float* stage, const float* wts, int ns, float* slab, bool first) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  if (first) {
    for (int i = tid; i < dm.slab_n; i += NT) slab[i] = 0.f;
    __syncthreads();
  }
  for (int l = 0; l < dm.nL; l++) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1], in32 = pad32(in), out32 = pad32(out);
    const int IT = in32 / 32, OT = out32 / 32;
    for (int tl = wave; tl < IT * OT; tl += NT / 64) {
      const int ot = tl / IT, it = tl - ot * IT;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
      for (int e = 0; e < ns; e++) {
        const float* pa = stage + (size_t)e * dm.blk_floats + dm.blk_off[l];
        const float* pd = pa + NB * in32;
        const float* wv = wts + (size_t)e * NB;
        for (int s8 = 0; s8 < 8; s8++) {
          const int n = 2 * s8 + half;
          const float w = ld_agent(wv + n);
          const float av = ld_agent(pa + n * in32 + it * 32 + l31);
          const float dv = ld_agent(pd + n * out32 + ot * 32 + l31);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w != 0.f ? dv * w : 0.f, acc, 0, 0, 0);
        }
      }
      float* g = slab + ((size_t)(dm.tile_off[l] + tl) * 64 + lane) * 16;
#pragma unroll
      for (int r = 0; r < 16; r++) g[r] += acc[r];
    }
    for (int row = tid; row < out; row += NT) {
      float sacc = 0.f;
      for (int e = 0; e < ns; e++) {
        const float* pd = stage + (size_t)e * dm.blk_floats + dm.blk_off[l] + NB * in32;
        for (int n = 0; n < NB; n++) {
          const float w = ld_agent(wts + (size_t)e * NB + n);
          const float dv = ld_agent(pd + n * out32 + row);
          sacc += w != 0.f ? dv * w : 0.f;
        }
      }
      slab[(size_t)dm.tile_off[dm.nL] * 1024 + dm.bias_lin[l] + row] += sacc;
    }
  }
}

// Reverse-time solve of [z; λ; g_θ] for one tile, with forced stops + jumps at the save times.
template <int SOLVER, int NT>
__global__ void __launch_bounds__(NT) k_mlp_adjoint(MlpDims dm, KOpts o, BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.fallback && __hip_atomic_load(a.ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // uniform
  const int T = o.T, B = o.B, Dp = dm.Dp, DpA = dm.DpA, D = dm.D, NP = dm.P;
  Ctl* c = reinterpret_cast<Ctl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(Ctl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  Panels P;
  const int NS = 2 * DpA + NP;        // panel rows in use (pad rows between the blocks stay 0)
  const int NREAL = 2 * Dp + NP;      // entries that count in the error norm
  P.lds = dm.ld_sb;
  P.ldh = dm.ld_h;
  P.pstride = NB * P.lds;
  P.hstride = NB * P.ldh;
  float* p = base;
  P.y = p; p += P.pstride;
  P.yn = p; p += P.pstride;
  P.tmp = p; p += P.pstride;
  P.kbase = p; p += 7 * P.pstride;
  P.scr = p; p += P.pstride;
  P.hidbase = p; p += dm.h_total;
  P.hoff = dm.h_off;
  P.delbase = p; p += 2 * P.hstride;
  P.red = p; p += (NT / 64) * 256;
  P.biasc = p; p += (dm.nbias + 3) & ~3;
  float* wst = p; p += NB;
  const int nfloat = (int)(p - base);
  for (int i = threadIdx.x; i < nfloat; i += NT) base[i] = 0.f;
  for (int i = threadIdx.x; i < T; i += NT) s_ts[i] = a.ts[i];
  __syncthreads();
  P.lbase = reinterpret_cast<const float*>(smem);
  P.gfrag = a.frag;
  P.gfragT = a.fragT;
  load_biases<NT>(dm, a.Wflat, P.biasc);
  {
    float* cend = reinterpret_cast<float*>(smem + a.lds_bytes);
    float* cp = cache_frags<NT>(dm, a.fragT, dm.fragT_off, dm.fragT_n, c->wTofs, P.lbase, p, cend);
    cache_frags<NT>(dm, a.frag, dm.frag_off, dm.frag_n, c->wofs, P.lbase, cp, cend);
  }
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * NB;
  const int nel = NS * NB;
  const int lds = P.lds;
  const bool coupled = dm.coupled != 0;
  const double tT = s_ts[T - 1], dtmax = fabs(tT - s_ts[0]);
  unsigned gen = 0;
  // staging area of this workgroup: slot e holds one weighted stage evaluation (block of panels + 16 column weights)
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;   // weighted stages per step attempt
  float* const my_stage = a.stage + (size_t)blockIdx.x * a.cap * dm.blk_floats;
  float* const my_wts = a.wts + (size_t)blockIdx.x * a.cap * NB;
  int slot_base = 0;      // slots [0, slot_base) hold accepted work; the running attempt writes slot_base + stage
  int nflush = 0;
// all NS×16 elements of a state panel: 32 row-lanes × NT/32 columns per pass (shifts only; conflict-free b32 accesses)
#define FOR_ELEMS(idx, colv)                            \
  for (int colv = tid >> 5; colv < NB; colv += NT / 32) \
    for (int r_ = tid & 31, idx = colv * lds + r_; r_ < NS; r_ += 32, idx += 32)


  // ---- load the terminal condition: z = ẑ(t_T), λ = Δ_T, g = 0 ------------------------------------------
  for (int e = tid; e < NB * Dp; e += NT) {
    const int col = e / Dp, row = e % Dp;
    if (b0 + col < B) {
      const size_t src = (size_t)Dp * ((size_t)(b0 + col) + (size_t)B * (T - 1)) + row;
      P.y[col * lds + row] = a.z_out[src];
      P.y[col * lds + DpA + row] = a.dz_out[src];
    }
  }
  __syncthreads();
  if (tid < NB) {
    const int col = tid;
    const bool valid = b0 + col < B;
    bool bad = false;
    for (int r = 0; r < Dp; r++) bad = bad || !isfinite(P.y[col * lds + r]);
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
    for (int r = 0; r < NS; r++) P.y[tid * lds + r] = 0.f;
  }
  __syncthreads();

  // ---- the reverse-time solve as a phase machine: ONE instantiation of eval_bwd in this kernel -------------------
  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 3;   // Tsit5: k1..k7 evaluated (s = 0..6); RK4: k1..k4 (s = 0..3)

  // start of a step attempt: iteration guard, clip to the next save time; false when every column has finished
  auto begin_step = [&]() -> bool {
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
    return c->any_active != 0;
  };

  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  int phase = auto_dt ? PH_K0 : PH_STAGE, s = 0;
  bool running = T > 1;
  if (running && !auto_dt) {
    if (tid < NB) c->dt[tid] = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    __syncthreads();
    running = begin_step();
  }
  while (running) {
    PROF_T(pl0);
    // ---- input of this evaluation + this stage's quadrature weights ------------------------------------------------
    const float* src = P.y;
    bool any_w = false;
    if (phase == PH_INIT1) src = P.tmp;
    if (phase == PH_STAGE) {
      if (s == 0 && slot_base + NST > a.cap) {
        // staging area full (far more steps than the sizing heuristic expected): fold it into the private slab
        __syncthreads();
        flush_stage<NT>(dm, my_stage, my_wts, slot_base, a.slab + (size_t)blockIdx.x * dm.slab_n, nflush == 0);
        slot_base = 0;
        nflush++;
        __syncthreads();
      }
      float bs;
      if (SOLVER == LDE_SOLVER_TSIT5) {
        if (s > 0) {
          float* dstp = s < 6 ? P.tmp : P.yn;
          FOR_ELEMS(idx, ecol) {
            float accv = ts5::A[s][0] * P.k(0)[idx];
            for (int jj = 1; jj < s; jj++) accv += ts5::A[s][jj] * P.k(jj)[idx];
            dstp[idx] = P.y[idx] + c->h[ecol] * accv;
          }
          src = dstp;
        }
        bs = s < 6 ? ts5::A[6][s] : 0.f;   // k₁ is evaluated fresh each attempt: it carries this step's weight b₁|h|
        any_w = s < 6;
      } else {
        if (s > 0) {
          const float cs = s == 3 ? 1.0f : 0.5f;
          FOR_ELEMS(idx, ecol) {
            P.tmp[idx] = P.y[idx] + (cs * c->h[ecol]) * P.k(s - 1)[idx];
          }
          src = P.tmp;
        }
        bs = (s == 0 || s == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
        any_w = true;
      }
      if (any_w && tid < NB) my_wts[(size_t)(slot_base + s) * NB + tid] = c->wq[tid] * bs;   // optimistic: zeroed on rejection
      __syncthreads();
    }
    float* dst = phase == PH_K0 ? P.k(0) : (phase == PH_INIT1 ? P.k(1) : P.k(s));

    PROF_T(pl1);
    PROF_ADD(0, pl0, pl1);
    eval_bwd<NT>(dm, P, c, src, dst, any_w ? my_stage + (size_t)(slot_base + s) * dm.blk_floats : nullptr);
    if (tid < NB && c->status[tid] == 0) c->nfe[tid]++;
    PROF_T(pl2);
    PROF_ADD(40, pl1, pl2);
#if LDE_PROF
    struct ProfEnd { long long t0; __device__ ~ProfEnd() { PROF_T(t1); PROF_ADD(1, t0, t1); } } prof_end{pl2};
#endif

    // ---- what follows the evaluation -----------------------------------------------------------------------------
    if (phase == PH_K0) {
      // Hairer–Nørsett–Wanner on the augmented state, direction −1: part 1
      FOR_ELEMS(idx, ecol) {
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
        for (int r = 0; r < NS; r++) { s0 += P.tmp[tid * lds + r]; s1 += P.yn[tid * lds + r]; }
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
        const float n = coupled ? (float)NREAL * (float)B : (float)NREAL;
        const float d0 = sqrtf((coupled ? v[0] : c->eest[tid]) / n), d1 = sqrtf((coupled ? v[1] : c->wq[tid]) / n);
        double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
        if (dt0 > dtmax) dt0 = dtmax;
        c->dt[tid] = dt0;
        c->h[tid] = c->status[tid] == 0 ? -(float)dt0 : 0.f;
        c->th[tid] = d1;
      }
      __syncthreads();
      FOR_ELEMS(idx, ecol) {
        P.tmp[idx] = P.y[idx] + c->h[ecol] * P.k(0)[idx];
      }
      __syncthreads();
      phase = PH_INIT1;
    } else if (phase == PH_INIT1) {
      FOR_ELEMS(idx, ecol) {
        const float dd = (P.k(1)[idx] - P.k(0)[idx]) * P.scr[idx];
        P.yn[idx] = dd * dd;
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.yn[tid * lds + r];
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
        const float n = coupled ? (float)NREAL * (float)B : (float)NREAL;
        const double dt0 = c->dt[tid];
        const float d2 = sqrtf((coupled ? w[0] : c->eest[tid]) / n) * fast_rcp((float)dt0);
        const float dm_ = fmaxf(c->th[tid], d2);
        const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
        double dt = fmin(100.0 * dt0, dt1);
        c->dt[tid] = dt > dtmax ? dtmax : dt;
      }
      __syncthreads();
      phase = PH_STAGE;
      s = 0;
      running = begin_step();
    } else if (s < LAST_STAGE) {
      s++;
    } else {
      // ---- all stages of this attempt are done ---------------------------------------------------------------------
      if (SOLVER == LDE_SOLVER_RK4) {
        FOR_ELEMS(idx, ecol) {
          P.yn[idx] = P.y[idx] + (c->h[ecol] * (1.0f / 6.0f)) * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        }
        __syncthreads();
      }
      // ---- error estimate + control ----------------------------------------------------------------------------
      FOR_ELEMS(idx, ecol) {
        float r2 = 0.f;
        const float yv = P.y[idx], ynv = P.yn[idx];
        if (o.adaptive) {
          float er = ts5::BT[0] * P.k(0)[idx];
#pragma unroll
          for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * P.k(jj)[idx];
          er *= c->h[ecol];
          const float sk = o.abstol + fmaxf(fabsf(yv), fabsf(ynv)) * o.reltol;
          const float r = er * fast_rcp(sk);
          r2 = r * r;
        }
        P.scr[idx] = isfinite(ynv) ? r2 : __int_as_float(0x7fc00000);
      }
      __syncthreads();
      if (tid < NB) {
        float s2 = 0.f;
        for (int r = 0; r < NS; r++) s2 += P.scr[tid * lds + r];
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
        const float n = coupled ? (float)NREAL * (float)B : (float)NREAL;
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
        int any = 0;
        for (int col = 0; col < NB; col++) any |= c->accepted[col];
        c->any_save = any;
      }
      __syncthreads();
      // the attempt's staged evaluations: rejected columns do not contribute; if nothing was accepted the slots are reused
      if (tid < NB && !c->accepted[tid] && c->wq[tid] != 0.f) {
#pragma unroll
        for (int st = 0; st < NST; st++) my_wts[(size_t)(slot_base + st) * NB + tid] = 0.f;
      }
      if (c->any_save) slot_base += NST;

      // ---- advance accepted columns; jump at a save time --------------------------------------------------------
      FOR_ELEMS(idx, ecol) {
        if (c->accepted[ecol]) P.y[idx] = P.yn[idx];
      }
      __syncthreads();
      for (int e = tid; e < NB * Dp; e += NT) {
        const int col = e / Dp, row = e % Dp;
        if (c->accepted[col] && c->hit[col]) {
          const size_t srcg = (size_t)Dp * ((size_t)(b0 + col) + (size_t)B * c->j[col]) + row;
          P.y[col * lds + DpA + row] += a.dz_out[srcg];
          if (o.checkpoint) P.y[col * lds + row] = a.z_out[srcg];
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
      s = 0;
      running = begin_step();
    }
  }

  if (tid == 0) {
    a.nslots[blockIdx.x] = slot_base;
    a.nflush[blockIdx.x] = nflush;
  }

  // ---- results ----------------------------------------------------------------------------------------------------
  for (int e = tid; e < NB * D; e += NT) {
    const int col = e / D, row = e % D;
    if (b0 + col < B) a.dz0[(size_t)(b0 + col) * D + row] = c->status[col] > 1 ? 0.f : P.y[col * lds + DpA + row];
  }
  if (NP) {
    for (int e = tid; e < NB * NP; e += NT) {
      const int col = e / NP, row = e % NP;
      if (b0 + col < B) a.dtheta[(size_t)(b0 + col) * NP + row] = c->status[col] > 1 ? 0.f : P.y[col * lds + 2 * DpA + row];
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
#undef FOR_ELEMS
}

#include "lde_mlpd.h"
#include "lde_mlp4.h"
#include "lde_mlpv.h"
#include "lde_mlp64.h"
#include "lde_mlpw.h"
#include "lde_mlpb.h"
#include "lde_mlpc.h"

// ================================================ host side =================================================
struct MlpPlan {
  MlpDims dm;
  VecDims vd;                  // small-batch kernels (lde_mlpv.h): geometry and the swizzled weight copies
  bool vec_ok = false;
  float* vecw = nullptr;
  WDims wd;                    // W-waves-per-trajectory register kernels (lde_mlpw.h)
  bool w_ok = false;
  float* wpack = nullptr;
  BDims bd;                    // block-layout register kernels (lde_mlpb.h)
  bool b_ok = false;
  float* bpack = nullptr;
  CDims cd;                    // two trajectories per workgroup, networks up to 128 wide (lde_mlpc.h)
  bool c_ok = false;
  float* cpack = nullptr;
  unsigned epoch = 0;          // launch counter of k_mlpw's tagged grid-sum words
  float* wslots = nullptr;     // k_mlpw's own [2][nWG][4] words ({value, tag} pairs): never shared with the float partials of grid_sum4
  int wslots_cap = 0;
  float* frag = nullptr;
  float* fragT = nullptr;
  size_t nfrag = 0, nfragT = 0;
  unsigned* counter = nullptr;
  float* slots = nullptr;
  int* abort_flag = nullptr;
  int cap_wg = 0;
  // adjoint workspace (allocated by the first lde_adjoint / lde_reserve, not by forward-only use)
  float* slab = nullptr;       // [nWG·(1+KS)+1][slab_n]: private (overflow) slabs, k_mlp_dw's (tile, part) slabs, their sum
  size_t slab_cap = 0;
  float* stage = nullptr;      // [nWG][cap][blk_floats]
  size_t stage_cap = 0;        // floats
  float* wts = nullptr;        // [nWG][cap][16]
  size_t wts_cap = 0;
  int32_t* nslots = nullptr;   // [2][nWG]: staged slots, flush counts
  int nslots_cap = 0;
  int adj_cap = 0;             // staging slots per workgroup the workspace is currently laid out for
  int adj_ks = 1;              // K-split of k_mlp_dw
  // feedback: the largest per-workgroup overflow count of the previous adjoint call, copied to pinned memory behind the
  // kernels; the next reservation doubles the staging area when it was non-zero (no host synchronisation anywhere)
  int32_t* fb_dev = nullptr;
  int32_t* fb_host = nullptr;
  hipEvent_t fb_ev = nullptr;
  bool fb_pending = false;
  int cap_scale = 1;
  // k_mlp64's adjoint (lde_mlp64.h) accumulates the weight gradient in registers: one row of nW floats per wave, summed by k_sum_rows
  float* rows = nullptr;       // [waves][rows_stride]
  size_t rows_cap = 0;
  int rows_stride = 0;
  // LDE_BATCH_COUPLED_GLOBAL: the cross-rank exchange of the step-control sums (lde_set_global_sum_hook)
  bool global_mode = false;
  lde_sum_hook sum_hook = nullptr;
  void* sum_user = nullptr;
  int64_t global_batch = 0;
  unsigned long long* mbox = nullptr;       // pinned, coherent host memory: [0..2] request words, [8..9] reply words
  unsigned long long* mbox_dev = nullptr;   // device memory: the reply republished for the other workgroups
  hipEvent_t mbox_ev = nullptr;
  // … or device to device (lde_set_global_sum_peers): every rank's mailbox as mapped on THIS device
  int peer_n = 0, peer_rank = 0;
  unsigned peer_launch = 0;    // launches that exchanged through the peer mailboxes (GridSync::xlaunch)
  unsigned long long* peer_box[LDE_MAX_PEERS] = {};
  // lde_set_phase_timing: HIP events around the adjoint's solve kernel and its weight-gradient tail (bench.py's per-kernel roofline)
  bool phase_on = false;
  hipEvent_t ph_ev[3] = {nullptr, nullptr, nullptr};
  bool disc = false;           // LDE_SENSE_DISCRETE: lde_adjoint sweeps the forward solve's step record (lde_mlpd.h)
  MlpTune tune;                // kernel-family switches and thresholds (lde_set_option)
  int last_family = -1;        // the kernel family of the last lde_adjoint: 0 tiles (gW in the tail: k_mlp_dw), 1 k_mlp64, 2 k_mlpb, 3 k_mlpc (gW folded
                               // in the solve kernel), 4 k_mlpw, 5 k_mlpv, 6 k_mlp4 (staged, tail)
};

void mlp_plan_destroy(MlpPlan* p);
static bool mlp64_applicable(const MlpPlan* p, int B);
static int mlp64_adj_waves(int B);
static bool b_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive);
static bool c_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive);
enum { DISC_TILES = 0, DISC_64 = 1, DISC_B = 2, DISC_C = 3 };
static int disc_family(const MlpPlan* p, int B, int T);

int mlp_plan_create(const lde_problem_desc& d, MlpPlan** out, std::string& err) {
  MlpPlan* p = new MlpPlan();
  MlpDims& dm = p->dm;
  dm.nL = d.n_layers;
  for (int i = 0; i <= d.n_layers; i++) dm.sizes[i] = d.layer_sizes[i];
  dm.act = d.activation;
  dm.D = d.state_dim;
  dm.Dp = d.state_dim + d.augment_dim;
  dm.DpA = (dm.Dp + 3) & ~3;
  dm.P = d.param_dim;
  dm.has_pend = d.rhs_kind == LDE_RHS_PENDULUM_PLUS_MLP;
  dm.coupled = d.batching == LDE_BATCH_COUPLED || d.batching == LDE_BATCH_COUPLED_GLOBAL;
  p->global_mode = d.batching == LDE_BATCH_COUPLED_GLOBAL;
  p->disc = d.sensealg == LDE_SENSE_DISCRETE;
  dm.solver = d.solver;
  int hmax = 16;
  for (int l = 0; l + 1 < dm.nL; l++) hmax = std::max(hmax, dm.sizes[l + 1]);
  fill_layer_offsets(dm, &p->nfrag, &p->nfragT);
  dm.hmax = (hmax + 31) & ~31;   // whole 32-row weight-gradient tiles (and 16-row K-groups)
  dm.ld_h = panel_stride(dm.hmax);
  dm.h_total = 0;
  for (int l = 0; l < MAXL; l++) { dm.ld_hl[l] = dm.ld_h; dm.h_off[l] = 0; }
  for (int l = 0; l + 1 < dm.nL; l++) {
    dm.ld_hl[l] = panel_stride((dm.sizes[l + 1] + 31) & ~31);
    dm.h_off[l] = dm.h_total;
    dm.h_total += NB * dm.ld_hl[l];
  }
  const int Dp16 = (dm.Dp + 15) & ~15, Dp32 = (dm.Dp + 31) & ~31;
  dm.ld_sf = panel_stride(Dp16);
  // adjoint state rows: z [0,Dp) | λ [DpA,DpA+Dp) | g [2DpA, 2DpA+P); operand reads reach DpA + Dp32 rows (32-row tiles)
  int rows_b = 2 * dm.DpA + dm.P;
  if (dm.DpA + Dp32 > rows_b) rows_b = dm.DpA + Dp32;
  dm.ld_sb = panel_stride((rows_b + 15) & ~15);
  if (dm.Dp > 256 || hmax > 1024) {
    err = "MLP RHS: state_dim+augment_dim ≤ 256 and hidden width ≤ 1024 supported";
    delete p;
    return LDE_ERR_UNSUPPORTED;
  }
  {   // one-trajectory-per-workgroup kernels: every width ≤ 256
    VecDims& vd = p->vd;
    int maxw = dm.Dp;
    for (int l = 0; l <= dm.nL; l++) maxw = std::max(maxw, dm.sizes[l]);
    p->vec_ok = maxw <= 256;
    if (p->vec_ok) {
      int nt = 64;
      while (nt < maxw) nt *= 2;
      // Register-resident hidden layer (lde_mlpv.h: vec_matvec_reg): three Dense layers with a hidden×hidden product of at
      // most 128×128 — the workgroup takes S·r lanes so that a lane's share of a row is VREG_K groups
      vd.reg_l = -1;
      int reg_r = 0, reg_s = 0;
      {
        const int hm = dm.nL == 3 ? std::max(dm.sizes[1], dm.sizes[2]) : 0;
        if (dm.nL == 3 && hm <= 128 && hm > 16) {
          reg_r = hm <= 64 ? 64 : 128;
          reg_s = reg_r / 64;                 // 64 → one wave holds the rows whole; 128 → two lane groups split K
          const int want = reg_r * reg_s;     // 64 or 256 lanes
          if (nt <= want) {
            nt = want;
            vd.reg_l = 1;
          }
        }
      }
      vd.NT = nt;
      auto geom = [&](int rows, int K, int* rp, int* lg, int* k4, int* S) {
        int r = 4, g = 2;
        while (r < rows) { r *= 2; g++; }
        const int kg = (K + 3) / 4;           // K-groups of 4
        int sp = nt / r;
        sp = sp > 8 ? 8 : sp;
        while (sp > 1 && (kg + sp - 1) / sp < 4) sp /= 2;   // a split lane's share is at least one round of 4 groups
        sp = sp < 1 ? 1 : sp;
        *S = sp;
        *lg = g;
        *rp = sp > 1 ? r : ((rows + 3) & ~3);   // no split: lane = row, no padding to a power of two
        const int per = (kg + sp - 1) / sp;
        *k4 = per < 4 ? per : ((per + 3) & ~3);   // groups per lane: 1–3, or a multiple of 4 (zero-padded) for the pipelined loop
      };
      int off = 0;
      for (int l = 0; l < dm.nL; l++) {
        geom(dm.sizes[l + 1], dm.sizes[l], &vd.rpf[l], &vd.lgf[l], &vd.k4f[l], &vd.sf[l]);
        if (l == vd.reg_l) { vd.rpf[l] = reg_r; vd.lgf[l] = reg_r == 64 ? 6 : 7; vd.k4f[l] = VREG_K; vd.sf[l] = reg_s; }
        vd.off_f[l] = off;
        off += vd.sf[l] * vd.k4f[l] * vd.rpf[l];
      }
      for (int l = 0; l < dm.nL; l++) {
        geom(dm.sizes[l], dm.sizes[l + 1], &vd.rpb[l], &vd.lgb[l], &vd.k4b[l], &vd.sb[l]);
        if (l == vd.reg_l) { vd.rpb[l] = reg_r; vd.lgb[l] = reg_r == 64 ? 6 : 7; vd.k4b[l] = VREG_K; vd.sb[l] = reg_s; }
        vd.off_b[l] = off;
        off += vd.sb[l] * vd.k4b[l] * vd.rpb[l];
      }
      vd.total4 = off;
      // every vector a product reads as x is followed by zeros up to the product's padded K (4·S·cnt floats): the padded weight
      // groups are zero, but 0·NaN is NaN, so the tail must never be another vector's (possibly non-finite) data
      vd.htotal = 0;
      vd.maxw4 = (maxw + 3) & ~3;
      for (int l = 0; l < dm.nL; l++) vd.maxw4 = std::max(vd.maxw4, 4 * vd.sb[l] * vd.k4b[l]);
      for (int l = 0; l < MAXL; l++) vd.hoff[l] = 0;
      for (int l = 0; l + 1 < dm.nL; l++) {
        vd.hoff[l] = vd.htotal;
        vd.htotal += std::max((dm.sizes[l + 1] + 3) & ~3, 4 * vd.sf[l + 1] * vd.k4f[l + 1]);
      }
      const int x0 = 4 * vd.sf[0] * vd.k4f[0], xt = dm.DpA + 4 * vd.sb[dm.nL - 1] * vd.k4b[dm.nL - 1];
      vd.nsp_f = std::max((dm.Dp + 3) & ~3, x0);
      vd.nsp_b = std::max(std::max((2 * dm.DpA + dm.P + 3) & ~3, x0), xt);
      if (hipMalloc(&p->vecw, (size_t)vd.total4 * 4 * sizeof(float)) != hipSuccess) {
        err = "MLP plan: hipMalloc failed";
        mlp_plan_destroy(p);
        return LDE_ERR_ALLOC;
      }
    }
  }
  {   // W waves per trajectory, weights in registers (lde_mlpw.h): three layers, no analytic part, H ≤ 200, D′ ≤ 32
    const int hm = dm.nL == 3 ? std::max(dm.sizes[1], dm.sizes[2]) : 0;
    p->w_ok = dm.nL == 3 && !dm.has_pend && dm.P == 0 && dm.Dp <= 32 && hm >= 1 && hm <= 200;
    if (p->w_ok) {
      WDims& wd = p->wd;
      wd.DP = dm.Dp <= 8 ? 8 : 32;
      wd.HP = hm <= 128 ? 128 : 200;
      wd.W = hm <= 128 ? 2 : 4;
      wd.UT = 64 * wd.W;
      wd.SEG = 64 / wd.DP;
      wd.GS = (wd.HP / wd.SEG + 3) / 4;   // = the kernel's compile-time GS
      wd.HX = (std::max(wd.UT, wd.SEG * wd.GS * 4) + 3) & ~3;
      wd.o_w1r = 0;
      wd.o_w2r = wd.o_w1r + wd.DP * wd.UT;
      wd.o_w2c = wd.o_w2r + wd.HP * wd.UT;
      wd.o_w3c = wd.o_w2c + wd.HP * wd.UT;
      wd.o_b1 = wd.o_w3c + wd.DP * wd.UT;
      wd.o_b2 = wd.o_b1 + wd.UT;
      wd.o_b3 = wd.o_b2 + wd.UT;
      wd.o_n3 = wd.o_b3 + 64;
      wd.o_n1 = wd.o_n3 + wd.GS * 64 * 4;
      wd.total = wd.o_n1 + wd.GS * 64 * 4;
      if (hipMalloc(&p->wpack, (size_t)wd.total * sizeof(float)) != hipSuccess) {
        err = "MLP plan: hipMalloc failed";
        mlp_plan_destroy(p);
        return LDE_ERR_ALLOC;
      }
    }
  }
  {   // four waves per trajectory, W₂ as 2-D register blocks, the weight gradient on the CU (lde_mlpb.h): three layers, H ≤ 200, D′ ≤ 16
    const int hm = dm.nL == 3 ? std::max(dm.sizes[1], dm.sizes[2]) : 0;
    p->b_ok = dm.nL == 3 && !dm.has_pend && dm.P == 0 && dm.Dp <= 16 && hm >= 1 && hm <= 200;
    if (p->b_ok) {
      BDims& bd = p->bd;
      bd.DP = dm.Dp <= 8 ? 8 : 16;
      bd.SEG = 64 / bd.DP;
      bd.GS = (200 / bd.SEG + 3) / 4;   // = the kernel's compile-time GS
      bd.o_wb = 0;
      bd.o_w13 = bd.o_wb + mlpb::RB * mlpb::CB * mlpb::UT;
      bd.o_b1 = bd.o_w13 + mlpb::HV * (2 * bd.DP + 4);
      bd.o_b3 = bd.o_b1 + mlpb::UT;
      bd.o_n3 = bd.o_b3 + 64;
      bd.o_n1 = bd.o_n3 + bd.GS * 64 * 4;
      bd.total = bd.o_n1 + bd.GS * 64 * 4;
      if (hipMalloc(&p->bpack, (size_t)bd.total * sizeof(float)) != hipSuccess) {
        err = "MLP plan: hipMalloc failed";
        mlp_plan_destroy(p);
        return LDE_ERR_ALLOC;
      }
    }
  }
  {   // two trajectories per workgroup on one register copy of the weights (lde_mlpc.h): three layers, H ≤ 128, D′ ≤ 32, coupled control
    const int hm = dm.nL == 3 ? std::max(dm.sizes[1], dm.sizes[2]) : 0;
    p->c_ok = dm.nL == 3 && !dm.has_pend && dm.P == 0 && dm.Dp <= 32 && hm >= 1 && hm <= 128 && dm.coupled;
    if (p->c_ok) {
      CDims& cd = p->cd;
      cd.o_wb = 0;
      cd.o_w3b = cd.o_wb + mlpc::RB * mlpc::CB * mlpc::UT;
      cd.o_w13 = cd.o_w3b + mlpc::RB * 2 * mlpc::UT;
      cd.o_b1 = cd.o_w13 + 128 * mlpc::W13S;
      cd.o_b3 = cd.o_b1 + 128;
      cd.o_n1 = cd.o_b3 + 64;
      cd.total = cd.o_n1 + mlpc::GS * 64 * 4;
      if (hipMalloc(&p->cpack, (size_t)cd.total * sizeof(float)) != hipSuccess) {
        err = "MLP plan: hipMalloc failed";
        mlp_plan_destroy(p);
        return LDE_ERR_ALLOC;
      }
    }
  }
  if (hipMalloc(&p->frag, p->nfrag * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->fragT, p->nfragT * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->counter, 64) != hipSuccess || hipMalloc(&p->abort_flag, 64) != hipSuccess ||
      hipMalloc(&p->fb_dev, 64) != hipSuccess || hipHostMalloc((void**)&p->fb_host, 64, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&p->fb_ev, hipEventDisableTiming) != hipSuccess) {
    err = "MLP plan: hipMalloc failed";
    mlp_plan_destroy(p);
    return LDE_ERR_ALLOC;
  }
  (void)hipMemset(p->abort_flag, 0, 64);
  (void)hipMemset(p->fb_dev, 0, 64);
  p->fb_host[0] = 0;
  *out = p;
  return LDE_OK;
}

void mlp_plan_destroy(MlpPlan* p) {
  if (!p) return;
  if (p->frag) (void)hipFree(p->frag);
  if (p->fragT) (void)hipFree(p->fragT);
  if (p->vecw) (void)hipFree(p->vecw);
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->bpack) (void)hipFree(p->bpack);
  if (p->cpack) (void)hipFree(p->cpack);
  if (p->wslots) (void)hipFree(p->wslots);
  if (p->counter) (void)hipFree(p->counter);
  if (p->abort_flag) (void)hipFree(p->abort_flag);
  if (p->slots) (void)hipFree(p->slots);
  if (p->slab) (void)hipFree(p->slab);
  if (p->stage) (void)hipFree(p->stage);
  if (p->wts) (void)hipFree(p->wts);
  if (p->nslots) (void)hipFree(p->nslots);
  if (p->rows) (void)hipFree(p->rows);
  if (p->mbox) (void)hipHostFree(p->mbox);
  if (p->mbox_dev) (void)hipFree(p->mbox_dev);
  if (p->mbox_ev) (void)hipEventDestroy(p->mbox_ev);
  for (int i = 0; i < 3; i++)
    if (p->ph_ev[i]) (void)hipEventDestroy(p->ph_ev[i]);
  if (p->fb_dev) (void)hipFree(p->fb_dev);
  if (p->fb_host) (void)hipHostFree(p->fb_host);
  if (p->fb_ev) (void)hipEventDestroy(p->fb_ev);
  delete p;
}

int mlp_reserve(MlpPlan* p, int B, int T, std::string& err) {
  const int nwg = B + 1;   // grid-sum slots: one per workgroup; the small-batch kernels run one trajectory per workgroup
  if (nwg > p->cap_wg) {
    if (p->slots) (void)hipFree(p->slots);
    p->slots = nullptr;
    if (hipMalloc(&p->slots, (size_t)2 * nwg * 4 * sizeof(float)) != hipSuccess) {
      err = "MLP plan: hipMalloc(slots) failed";
      return LDE_ERR_ALLOC;
    }
    p->cap_wg = nwg;
  }
  (void)T;
  return LDE_OK;
}

// Workspace of the adjoint for batches up to B with T save points. `steps_hint` > 0: step attempts known in advance
// (fixed step size). Staging slots per workgroup: stages × (3 step attempts per save interval + 32) — doubled whenever
// the previous call reported an overflow — bounded by a memory budget
// (24 GiB of the 288); a workgroup that needs more folds its slots into its private slab
// (slow but correct). option "mlp_stage_slots" forces the slot count (tests).
int mlp_reserve_adjoint(MlpPlan* p, int B, int T, int64_t steps_hint, std::string& err) {
  const MlpDims& dm = p->dm;
  if (mlp64_applicable(p, B)) {   // no staging area: a slab row per wave is the kernel's only workspace (continuous and discrete adjoint alike)
    p->rows_stride = (dm.nW + 63) & ~63;
    if (!grow(&p->rows, &p->rows_cap, (size_t)mlp64_adj_waves(B) * p->rows_stride)) {
      err = "MLP plan: hipMalloc of the weight-gradient rows failed";
      return LDE_ERR_ALLOC;
    }
    return LDE_OK;
  }
  const int dfam = p->disc ? disc_family(p, B, T) : -1;
  if (p->disc ? (dfam == DISC_B || dfam == DISC_C)
              : (b_applicable(p, B, T, true, dm.coupled != 0) || c_applicable(p, B, T, true, dm.coupled != 0))) {   // no staging area either: one slab row per workgroup
    p->rows_stride = (dm.nW + 63) & ~63;
    if (!grow(&p->rows, &p->rows_cap, (size_t)B * p->rows_stride)) {
      err = "MLP plan: hipMalloc of the weight-gradient rows failed";
      return LDE_ERR_ALLOC;
    }
    return LDE_OK;
  }
  const int nwg = cdiv(B, NB);
  const int nst = dm.solver == LDE_SOLVER_RK4 ? 4 : 6;
  constexpr long budget_mb = 24576L;
  const int slots_force = p->tune.stage_slots;
  if (p->fb_pending) {
    if (hipEventQuery(p->fb_ev) == hipSuccess) {
      p->fb_pending = false;
      if (p->fb_host[0] > 0 && p->cap_scale < 64) p->cap_scale *= 2;
    } else
      (void)hipGetLastError();   // hipErrorNotReady is not an error of the caller's
  }
  int64_t want = (int64_t)nst * (steps_hint > 0 ? steps_hint + 2 : p->cap_scale * (3 * (int64_t)(T > 1 ? T - 1 : 1) + 32));
  const int64_t fit = ((int64_t)budget_mb << 20) / ((int64_t)(nwg + 1) * dm.blk_floats * (int64_t)sizeof(float));
  if (want > fit) want = fit;
  if (slots_force > 0) want = slots_force;
  if (want < nst) want = nst;
  if (want > (1 << 24)) want = 1 << 24;
  int ks = cdiv(768, nwg * dw_jobs(dm, dw_pick_ndw(dm)));
  ks = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
  size_t nsl = (size_t)p->nslots_cap;
  // the staging area is the large allocation: if the device cannot give it (another allocator holds the memory, a smaller
  // device), halve the slot count down to one step's worth — a workgroup that runs out of slots folds them into its private
  // slab inside the solve kernel (exact), so fewer slots cost time, never correctness
  int cap = (int)want;
  const size_t stage_before = p->stage_cap;
  for (;;) {
    if (grow(&p->stage, &p->stage_cap, (size_t)(nwg + 1) * cap * dm.blk_floats)) break;
    (void)hipGetLastError();
    if (cap <= nst || slots_force > 0 || steps_hint > 0) { cap = 0; break; }
    cap = cap / 2 < nst ? nst : cap / 2;
  }
  if (cap == 0 || !grow(&p->wts, &p->wts_cap, (size_t)(nwg + 1) * cap * NB) ||
      !grow(&p->slab, &p->slab_cap, ((size_t)(nwg + 1) * (1 + ks) + 1) * dm.slab_n) || !grow(&p->nslots, &nsl, (size_t)2 * (nwg + 1))) {
    err = "MLP plan: hipMalloc of the adjoint workspace failed";
    return LDE_ERR_ALLOC;
  }
  // a fresh staging area holds arbitrary bits; the one-trajectory-per-workgroup kernel fills a tile's slots column by column,
  // and a column that never reaches a slot carries weight 0 there — its a-panel must still be finite (0·NaN) for k_mlp_dw
  if (p->stage_cap != stage_before && hipMemset(p->stage, 0, p->stage_cap * sizeof(float)) != hipSuccess) {
    err = "MLP plan: hipMemset(stage) failed";
    return LDE_ERR_HIP;
  }
  p->nslots_cap = (int)nsl;
  p->adj_cap = cap;
  p->adj_ks = ks;
  return LDE_OK;
}

int mlp_set_weights(MlpPlan* p, const float* W_dev, hipStream_t stream, std::string& err) {
  hipLaunchKernelGGL(k_build_frags, dim3(64, p->dm.nL), dim3(256), 0, stream, W_dev, p->dm, p->frag, p->fragT, (float*)nullptr, (__bf16*)nullptr, (__bf16*)nullptr);
  if (p->vec_ok) hipLaunchKernelGGL(k_build_vec, dim3(64, p->dm.nL), dim3(256), 0, stream, W_dev, p->dm, p->vd, p->vecw);
  if (p->w_ok) hipLaunchKernelGGL(k_build_wpack, dim3(128), dim3(256), 0, stream, W_dev, p->dm, p->wd, p->wpack);
  if (p->b_ok) hipLaunchKernelGGL(k_build_bpack, dim3(128), dim3(256), 0, stream, W_dev, p->dm, p->bd, p->bpack);
  if (p->c_ok) hipLaunchKernelGGL(k_build_cpack, dim3(128), dim3(256), 0, stream, W_dev, p->dm, p->cd, p->cpack);
  if (hipGetLastError() != hipSuccess) {
    err = "k_build_frags launch failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}


#if LDE_PROF
static void prof_reset() {
  long long z[64] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
}
static void prof_dump(const char* what, hipStream_t stream) {
  static int calls = 0;
  (void)hipStreamSynchronize(stream);
  long long v[64];
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
  if (++calls % 20 != 0) return;   // a few samples are enough
  fprintf(stderr, "[prof %s] cycles(100MHz ticks):", what);
  for (int i = 0; i < 64; i++)
    if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
  fprintf(stderr, "\n");
}
#endif

static size_t fwd_lds_fixed(const MlpDims& dm, int T, int nt) {
  const int NW = nt / 64;
  size_t b = (sizeof(Ctl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  b += (size_t)(11 * NB * dm.ld_sf + dm.h_total + NW * 256 + ((dm.nbias + 3) & ~3)) * sizeof(float);
  return b;
}

static size_t bwd_lds_fixed(const MlpDims& dm, int T, int nt) {
  const int NW = nt / 64;
  size_t b = (sizeof(Ctl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  b += (size_t)(11 * NB * dm.ld_sb + dm.h_total + 2 * NB * dm.ld_h + NW * 256 + ((dm.nbias + 3) & ~3) + NB) * sizeof(float);
  return b;
}

// the weight cache takes whatever the panels leave, up to what the fragments need
static size_t with_cache(size_t fixed, size_t want_floats) {
  size_t total = fixed + want_floats * sizeof(float);
  if (total > LDS_MAX) total = LDS_MAX;
  return total & ~size_t(15);
}

// Kernels that run the grid-wide sum (coupled adaptive control) need every workgroup resident at once: they are launched
// COOPERATIVELY — the runtime checks that the grid fits the device and keeps other streams' work from taking its CUs —
// instead of assuming residency (the bounded spin of grid_sum4 stays as the last line of defence).
template <class... A>
static int launch_maybe_coop(bool coop, const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::string& err,
                             const char* what, A&... args) {
  constexpr bool coop_on = true;
  void* argv[] = {(void*)&args...};
  hipError_t rc;
  if (coop && coop_on) {
    rc = hipLaunchCooperativeKernel(fn, grid, block, argv, (unsigned)lds, stream);
    if (rc == hipErrorCooperativeLaunchTooLarge) {
      (void)hipGetLastError();
      err = std::string(what) + ": the coupled adaptive solve needs all its workgroups resident at once and this grid does not fit the device";
      return LDE_ERR_UNSUPPORTED;
    }
  } else
    rc = hipLaunchKernel(fn, grid, block, argv, lds, stream);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    err = std::string(what) + " launch failed: " + hipGetErrorString(rc);
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

// ---- one wave per trajectory, everything in registers (lde_mlp64.h): small networks on small states, per-trajectory control
static bool mlp64_applicable(const MlpPlan* p, int B) {
  const MlpDims& dm = p->dm;
  if (!p->tune.mlp64) return false;
  const int maxb = 65536;   // measured (c3 shape): 0.28 + 3.9 ms vs 0.61 + 5.8 for the tile kernels at B = 4096, 0.77 + 10.4 vs 2.2 + 12.0 at 16384
  return dm.nL == 3 && dm.sizes[1] <= 64 && dm.sizes[2] <= 64 && dm.Dp <= 4 && dm.P <= 1 && !dm.coupled && B <= maxb;
}
// waves of the adjoint launch: one per SIMD (the kernel takes more than 256 registers); a wave walks trajectories b, b + waves, …
// with ONE set of gradient sums, so the slab the final sum reads has `waves` rows whatever the batch
// workgroups of the adjoint launch: four waves each (one per SIMD of a CU — the kernel takes more than 256 registers), ≤ 256 of them; a
// wave walks trajectories b, b + 4·workgroups, … with ONE set of gradient sums, the four waves' sums meet in LDS, and the slab the final
// sum reads has one row per workgroup whatever the batch
constexpr int MLP64_NWV = 4;
static int mlp64_adj_waves(int B) {   // = workgroups = slab rows
  constexpr int maxw = 256;
  const int need = (B + MLP64_NWV - 1) / MLP64_NWV;
  return need < maxw ? need : maxw;
}
template <bool ADJ, bool DISC = false>
static int launch_mlp64(const MlpDims& dm, const KOpts& o, const VArgs& a, hipStream_t stream, std::string& err) {
  const bool rk4 = dm.solver == LDE_SOLVER_RK4, d2 = dm.Dp <= 2;
  if (ADJ) {
    const dim3 grid(mlp64_adj_waves(o.B));
    const size_t lds = (size_t)MLP64_NWV * a.cap * sizeof(float);   // the four waves' rows (c3: 4 × 17.9 KB)
    const void* fn = DISC ? (rk4 ? (d2 ? (const void*)k_mlp64_disc<LDE_SOLVER_RK4, 2> : (const void*)k_mlp64_disc<LDE_SOLVER_RK4, 4>)
                                 : (d2 ? (const void*)k_mlp64_disc<LDE_SOLVER_TSIT5, 2> : (const void*)k_mlp64_disc<LDE_SOLVER_TSIT5, 4>))
                          : (rk4 ? (d2 ? (const void*)k_mlp64_adj<LDE_SOLVER_RK4, 2> : (const void*)k_mlp64_adj<LDE_SOLVER_RK4, 4>)
                                 : (d2 ? (const void*)k_mlp64_adj<LDE_SOLVER_TSIT5, 2> : (const void*)k_mlp64_adj<LDE_SOLVER_TSIT5, 4>));
    static bool attr_set[2][2] = {{false, false}, {false, false}};
    if (!attr_set[rk4][d2]) {
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, fn);
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX - (int)fa.sharedSizeBytes) != hipSuccess) {
        (void)hipGetLastError();
        err = "hipFuncSetAttribute(k_mlp64_adj) failed";
        return LDE_ERR_HIP;
      }
      attr_set[rk4][d2] = true;
    }
    MlpDims dmv = dm;
    KOpts ov = o;
    VArgs av = a;
    void* argv[] = {(void*)&dmv, (void*)&ov, (void*)&av};
    (void)hipLaunchKernel(fn, grid, dim3(64 * MLP64_NWV), argv, lds, stream);
  } else if (rk4 && d2) hipLaunchKernelGGL((k_mlp64<LDE_SOLVER_RK4, 2>), dim3(o.B), dim3(64), 0, stream, dm, o, a);
  else if (rk4) hipLaunchKernelGGL((k_mlp64<LDE_SOLVER_RK4, 4>), dim3(o.B), dim3(64), 0, stream, dm, o, a);
  else if (d2) hipLaunchKernelGGL((k_mlp64<LDE_SOLVER_TSIT5, 2>), dim3(o.B), dim3(64), 0, stream, dm, o, a);
  else hipLaunchKernelGGL((k_mlp64<LDE_SOLVER_TSIT5, 4>), dim3(o.B), dim3(64), 0, stream, dm, o, a);
  if (hipGetLastError() != hipSuccess) {
    err = "k_mlp64 launch failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

// ---- the one-trajectory-per-workgroup kernels (lde_mlpv.h): applicability, LDS budget, launch -------------------------------------
static size_t vec_lds_fixed(const MlpDims& dm, const VecDims& vd, int T, bool adj) {
  const int nsp = adj ? vd.nsp_b : vd.nsp_f;
  size_t b = (sizeof(VCtl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  b += (size_t)(11 * nsp + vd.htotal + MAXL * vd.maxw4 + vd.NT + ((dm.nbias + 3) & ~3) + 2 * MAXL * (sizeof(VLayer) / 4)) * sizeof(float);
  return (b + 15) & ~size_t(15);
}
// Which batches run there: see the measurement below. Option "mlpv" = 0 switches the kernels off.
static bool vec_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive, size_t* lds, std::string& why) {
  if (!p->vec_ok || !p->tune.mlpv) return false;
  // measured (MI355X, c2 / c3 / c4 shapes, abl/ + profiles/): the one-trajectory workgroups win while the chip has a SIMD per
  // wave (B·NT/64 ≤ 1024: c2 0.88 + 1.97 ms vs 1.77 + 3.44 at B = 256, c3 0.36 + 4.28 vs 0.63 + 5.0 at 1024, c4 0.46 + 3.77 vs
  // 0.45 + 4.3 at 512) and lose beyond (c2 at B = 1024: 1.94 + 4.5 vs 1.78 + 3.9) — the tiles then have enough columns
  const int maxb = (p->vd.reg_l >= 0 ? 2048 : 1024) * 64 / p->vd.NT;   // register-resident layer: two waves per SIMD still win
  if (B > maxb) return false;
  const size_t fixed = vec_lds_fixed(p->dm, p->vd, T, adj);
  if (fixed > LDS_MAX / 2) return false;
  // LDS per workgroup: everything when a CU gets one workgroup, a share otherwise (coupled adaptive control needs all B resident)
  const int per_cu = cdiv(B, 256);
  size_t budget = LDS_MAX / (size_t)per_cu;
  if (per_cu > 1) budget -= 256;
  if (budget < fixed) {
    if (coupled_adaptive) return false;
    budget = fixed;
  }
  const size_t want = fixed + (size_t)p->vd.total4 * 16;
  *lds = (std::min(want, budget)) & ~size_t(15);
  (void)why;
  return true;
}

template <int SOLVER, bool ADJ>
static int launch_vec(const MlpPlan* p, const KOpts& o, VArgs& a, size_t lds, bool coop, hipStream_t stream, std::string& err) {
  MlpDims dmv = p->dm;
  VecDims vdv = p->vd;
  KOpts ov = o;
  constexpr int KB = ADJ ? VREG_K : 0;
  const bool reg = p->vd.reg_l >= 0;
  const void* fn = reg ? (p->vd.NT == 64 ? (const void*)k_mlpv<SOLVER, 64, ADJ, VREG_K, KB> : (const void*)k_mlpv<SOLVER, 256, ADJ, VREG_K, KB>)
                       : (p->vd.NT == 64 ? (const void*)k_mlpv<SOLVER, 64, ADJ> : p->vd.NT == 128 ? (const void*)k_mlpv<SOLVER, 128, ADJ> : (const void*)k_mlpv<SOLVER, 256, ADJ>);
  static bool attr_set[5] = {false, false, false, false, false};
  const int ki = reg ? (p->vd.NT == 64 ? 3 : 4) : (p->vd.NT == 64 ? 0 : (p->vd.NT == 128 ? 1 : 2));
  if (!attr_set[ki]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlpv) failed";
      return LDE_ERR_HIP;
    }
    attr_set[ki] = true;
  }
  a.lds_bytes = (int)lds;
#if LDE_PROF
  prof_reset();
#endif
  const int rcl = launch_maybe_coop(coop, fn, dim3(o.B), dim3(p->vd.NT), lds, stream, err, "k_mlpv", dmv, vdv, ov, a);
#if LDE_PROF
  prof_dump(ADJ ? "vec adjoint" : "vec forward", stream);
#endif
  return rcl;
}

// ---- W waves per trajectory, weights and state in registers (lde_mlpw.h)
static size_t w_lds_base(const WDims& wd, int T, bool adj) {   // save times, state copies, exchange buffers, narrow slices
  return (((size_t)T * 8 + 15) & ~size_t(15)) + (size_t)(wd.W * 64 + 2 * wd.HX) * 4 + (size_t)wd.GS * 64 * 16 * (adj ? 2 : 1) + 16;
}
static bool w_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive) {
  if (!p->w_ok || !p->tune.mlpw) return false;
  if (w_lds_base(p->wd, T, adj) > LDS_MAX * p->wd.W / 4) return false;   // 4/W workgroups share a CU's LDS (one wave per SIMD)
  // one wave per SIMD (the weights take most of the 512 registers): 1024 waves are resident at once; an uncoupled solve may
  // queue a second round, a coupled one needs every trajectory resident
  const int maxw = coupled_adaptive ? 1024 : 2048;
  return (long long)B * p->wd.W <= maxw;
}
// tagged grid-sum words of w_grid_sum: an own buffer (zeroed: no tag is 0), a fresh epoch per launch, cleared when the epoch wraps
static int grid_words_prepare(MlpPlan* p, int B, VArgs& a, hipStream_t stream, std::string& err) {
  const size_t bytes = (size_t)2 * (B + 1) * 4 * sizeof(float);
  if (B + 1 > p->wslots_cap) {
    if (p->wslots) (void)hipFree(p->wslots);
    p->wslots = nullptr;
    p->wslots_cap = 0;
    if (hipMalloc(&p->wslots, bytes) != hipSuccess || hipMemsetAsync(p->wslots, 0, bytes, stream) != hipSuccess) {
      (void)hipGetLastError();
      err = "MLP plan: hipMalloc of the grid-sum words failed";
      return LDE_ERR_ALLOC;
    }
    p->wslots_cap = B + 1;
    p->epoch = 0;
  }
  p->epoch = (p->epoch + 1) & 0xffffu;
  if (p->epoch == 0) {
    if (hipMemsetAsync(p->wslots, 0, (size_t)2 * p->wslots_cap * 4 * sizeof(float), stream) != hipSuccess) {
      err = "hipMemsetAsync(grid-sum words) failed";
      return LDE_ERR_HIP;
    }
    p->epoch = 1;
  }
  a.epoch = p->epoch;
  a.gs.slots = p->wslots;
  return LDE_OK;
}
template <int SOLVER, bool ADJ>
static int launch_w(MlpPlan* p, const KOpts& o, VArgs& a, bool coop, hipStream_t stream, std::string& err) {
  MlpDims dmv = p->dm;
  WDims wdv = p->wd;
  KOpts ov = o;
  const bool d8 = wdv.DP == 8, w2 = wdv.W == 2;
  const void* fn = w2 ? (d8 ? (const void*)k_mlpw<SOLVER, 8, 128, 2, ADJ> : (const void*)k_mlpw<SOLVER, 32, 128, 2, ADJ>)
                      : (d8 ? (const void*)k_mlpw<SOLVER, 8, 200, 4, ADJ> : (const void*)k_mlpw<SOLVER, 32, 200, 4, ADJ>);
  size_t lds = w_lds_base(wdv, o.T, ADJ);
  const size_t cot = ADJ ? (size_t)o.T * dmv.Dp * 4 * (o.checkpoint ? 2 : 1) : 0;
  a.cot_lds = ADJ && cot <= 40 * 1024 && lds + cot <= LDS_MAX * wdv.W / 4;   // the trajectory's dẑ (and saved ẑ) by save time: no global load inside the solve
  if (a.cot_lds) lds += cot;
  if (lds > LDS_MAX) {
    err = "k_mlpw: the save-time grid does not fit LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  {   // dynamic LDS beyond the default limit needs the attribute, once per instantiation
    static bool attr_set[2][2] = {{false, false}, {false, false}};
    if (!attr_set[d8][w2]) {
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, fn);
      const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX - (int)fa.sharedSizeBytes);
      if (ea != hipSuccess) {
        (void)hipGetLastError();
        err = std::string("hipFuncSetAttribute(k_mlpw) failed: ") + hipGetErrorString(ea) + " d8=" + std::to_string(d8) + " w2=" + std::to_string(w2) +
              " adj=" + std::to_string(ADJ) + " static=" + std::to_string(fa.sharedSizeBytes) + " regs=" + std::to_string(fa.numRegs);
        return LDE_ERR_HIP;
      }
      attr_set[d8][w2] = true;
    }
  }
  a.wpack = p->wpack;
  const bool relay = a.gs.cross();   // LDE_BATCH_COUPLED_GLOBAL: sums leave the device; a plain launch (two ranks' cooperative
                                                 // launches on one device could be serialised by the runtime — each would wait for the other's sums)
  if (coop || relay) {
    const int rcw = grid_words_prepare(p, o.B, a, stream, err);
    if (rcw) return rcw;
  }
#if LDE_PROF
  prof_reset();
#endif
  const int rcl = launch_maybe_coop(coop && !relay, fn, dim3(o.B), dim3(wdv.UT), lds, stream, err, "k_mlpw", dmv, wdv, ov, a);
#if LDE_PROF
  prof_dump(ADJ ? "w adjoint" : "w forward", stream);
#endif
  return rcl;
}

// ---- four waves per trajectory, W₂ as register blocks, the weight gradient folded on the CU (lde_mlpb.h)
static size_t b_lds_base(const BDims& bd, int T, bool adj, int nst) {   // save times, ring (+ partial sums), narrow slices
  const int nsl = adj ? nst + 1 : 1;
  return (((size_t)T * 8 + 15) & ~size_t(15)) + (size_t)(nsl * mlpb::SLOT + (adj ? 16 * mlpb::HV : 0)) * 4 +
         (size_t)bd.GS * 64 * 16 * (adj ? 2 : 1) + (size_t)mlpb::HV * (2 * bd.DP + 4) * 4 + 16 + ((adj && bd.DP == 16) ? 2 * mlpb::W * 16 * 4 : 0);
}
static bool b_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive) {
  // option "mlpb": 0 = off (k_mlpw: the parity reference of this kernel), 2 = also the networks of at most 128 units that k_mlpw's
  // two-wave form serves by default; "mlpw" = 0 switches BOTH register families off (the tests' "tiles" / "k_mlpv" legs)
  const int mode = p->tune.mlpb;
  if (!p->b_ok || mode == 0 || !p->tune.mlpw) return false;
  const int hm = std::max(p->dm.sizes[1], p->dm.sizes[2]);
  if (hm <= 128 && mode != 2) return false;
  if (b_lds_base(p->bd, T, adj, p->dm.solver == LDE_SOLVER_RK4 ? 4 : 6) > LDS_MAX) return false;
  // one workgroup per CU (512 registers per lane): 256 trajectories are resident at once; an uncoupled solve may queue a second
  // round, a coupled adaptive one needs every trajectory resident
  return B <= (coupled_adaptive ? 256 : 512);
}
template <int SOLVER, bool ADJ, bool DISC = false>
static int launch_b(MlpPlan* p, const KOpts& o, VArgs& a, bool coop, hipStream_t stream, std::string& err) {
  MlpDims dmv = p->dm;
  BDims bdv = p->bd;
  KOpts ov = o;
  const bool d8 = bdv.DP == 8;
  const bool tanh_ = dmv.act == LDE_ACT_TANH;
  const size_t cot = ADJ ? (size_t)o.T * dmv.Dp * 4 * (o.checkpoint ? 2 : 1) : 0;
  const void* fn = tanh_ ? (d8 ? (const void*)k_mlpb<SOLVER, 8, LDE_ACT_TANH, DISC, ADJ> : (const void*)k_mlpb<SOLVER, 16, LDE_ACT_TANH, DISC, ADJ>)
                         : (d8 ? (const void*)k_mlpb<SOLVER, 8, LDE_ACT_RELU, DISC, ADJ> : (const void*)k_mlpb<SOLVER, 16, LDE_ACT_RELU, DISC, ADJ>);
  size_t lds = b_lds_base(bdv, o.T, ADJ, SOLVER == LDE_SOLVER_RK4 ? 4 : 6);
  a.cot_lds = ADJ && cot <= 40 * 1024 && lds + cot <= LDS_MAX;   // the trajectory's dẑ (and saved ẑ) by save time: no global load inside the solve
  if (a.cot_lds) lds += cot;
  {   // dynamic LDS beyond the default limit needs the attribute, once per instantiation
    static bool attr_set[2][2] = {};
    if (!attr_set[d8][tanh_]) {
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, fn);
      const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX - (int)fa.sharedSizeBytes);
      if (ea != hipSuccess) {
        (void)hipGetLastError();
        err = std::string("hipFuncSetAttribute(k_mlpb) failed: ") + hipGetErrorString(ea);
        return LDE_ERR_HIP;
      }
      attr_set[d8][tanh_] = true;
    }
  }
  a.wpack = p->bpack;
  const bool relay = a.gs.cross();   // LDE_BATCH_COUPLED_GLOBAL: sums leave the device; a plain launch (see launch_w)
  if (coop || relay) {
    const int rcw = grid_words_prepare(p, o.B, a, stream, err);
    if (rcw) return rcw;
  }
#if LDE_PROF
  prof_reset();
#endif
  const int rcl = launch_maybe_coop(coop && !relay, fn, dim3(o.B), dim3(mlpb::UT), lds, stream, err, "k_mlpb", dmv, bdv, ov, a);
#if LDE_PROF
  prof_dump(ADJ ? "b adjoint" : "b forward", stream);
#endif
  return rcl;
}

// ---- two trajectories per workgroup, networks up to 128 wide (lde_mlpc.h)
static size_t c_lds_base(int T, bool adj, int nst, bool disc = false) {
  const int nsl = disc ? 2 * nst + 1 : (adj ? nst + 1 : 1);   // (the discrete sweep: three rotating first-stage slots + two banks of nst − 1)
  return (((size_t)T * 8 + 15) & ~size_t(15)) + (size_t)(nsl * mlpc::SLOT + (adj ? 16 * mlpc::HV : 0) + 16 * 16 * 4) * 4 +
         (size_t)128 * mlpc::W13S * 4 + (adj ? (size_t)mlpc::GS * 64 * 16 + 4 * mlpc::DP * 2 * 4 : 0) + 16;
}
static bool c_applicable(const MlpPlan* p, int B, int T, bool adj, bool coupled_adaptive) {
  if (!p->c_ok || p->tune.mlpb == 0 || !p->tune.mlpw) return false;   // (the switches of k_mlpb: 0 = k_mlpw instead — the parity reference)
  if (c_lds_base(T, adj, p->dm.solver == LDE_SOLVER_RK4 ? 4 : 6) > LDS_MAX) return false;
  // one workgroup (two trajectories) per CU: 512 trajectories are resident at once, which a coupled adaptive solve needs
  return B <= (coupled_adaptive ? 512 : 1024);
}
// LDE_SENSE_DISCRETE: the kernel family that sweeps the step record. The register kernels take the shapes they serve in the continuous
// adjoint — without that path's residency limit (no grid-wide sum here: workgroups may queue) — up to what one slab row per workgroup
// costs in memory; everything else runs on the tiles (lde_mlpd.h).
static int disc_family(const MlpPlan* p, int B, int T) {
  const MlpDims& dm = p->dm;
  if (mlp64_applicable(p, B)) return DISC_64;
  const int nst = dm.solver == LDE_SOLVER_RK4 ? 4 : 6;
  if (b_applicable(p, std::min(B, 256), T, true, false) && B <= 1024 && b_lds_base(p->bd, T, true, nst) + (size_t)T * dm.Dp * 4 <= LDS_MAX) return DISC_B;
  if (c_applicable(p, std::min(B, 512), T, true, false) && B <= 2048 && c_lds_base(T, true, nst, true) <= LDS_MAX) return DISC_C;
  return DISC_TILES;
}

template <int SOLVER, bool ADJ, bool DISC = false>
static int launch_c(MlpPlan* p, const KOpts& o, VArgs& a, bool coop, hipStream_t stream, std::string& err) {
  MlpDims dmv = p->dm;
  CDims cdv = p->cd;
  KOpts ov = o;
  const bool tanh_ = dmv.act == LDE_ACT_TANH;
  const void* fn = tanh_ ? (const void*)k_mlpc<SOLVER, LDE_ACT_TANH, DISC, ADJ> : (const void*)k_mlpc<SOLVER, LDE_ACT_RELU, DISC, ADJ>;
  size_t lds = c_lds_base(o.T, ADJ, SOLVER == LDE_SOLVER_RK4 ? 4 : 6, DISC);
  if (lds > LDS_MAX) {
    err = "k_mlpc: the save-time grid does not fit LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  const size_t cot = ADJ ? (size_t)2 * o.T * dmv.Dp * 4 * ((o.checkpoint && !DISC) ? 2 : 1) : 0;
  a.cot_lds = ADJ && lds + cot <= LDS_MAX;   // the two trajectories' dẑ (and saved ẑ) by save time: no global load inside the solve
  if (a.cot_lds) lds += cot;
  {   // dynamic LDS beyond the default limit needs the attribute, once per instantiation
    static bool attr_set[2] = {false, false};
    if (!attr_set[tanh_]) {
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, fn);
      const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX - (int)fa.sharedSizeBytes);
      if (ea != hipSuccess) {
        (void)hipGetLastError();
        err = std::string("hipFuncSetAttribute(k_mlpc) failed: ") + hipGetErrorString(ea);
        return LDE_ERR_HIP;
      }
      attr_set[tanh_] = true;
    }
  }
  a.wpack = p->cpack;
  const int nwg = (o.B + 1) / 2;
  a.gs.nwg = a.gs.nwg > 1 ? nwg : a.gs.nwg;   // (grid-wide sums: one word pair per workgroup)
  const bool relay = a.gs.cross();   // LDE_BATCH_COUPLED_GLOBAL: sums leave the device; a plain launch (see launch_w)
  if (coop || relay) {
    const int rcw = grid_words_prepare(p, o.B, a, stream, err);
    if (rcw) return rcw;
  }
#if LDE_PROF
  prof_reset();
#endif
  const int rcl = launch_maybe_coop(coop && !relay, fn, dim3(nwg), dim3(mlpc::UT), lds, stream, err, "k_mlpc", dmv, cdv, ov, a);
#if LDE_PROF
  prof_dump(ADJ ? "c adjoint" : "c forward", stream);
#endif
  return rcl;
}

int mlp_last_family(const MlpPlan* p) { return p->last_family; }
MlpTune* mlp_tune(MlpPlan* p) { return &p->tune; }

int mlp_set_phase_timing(MlpPlan* p, int on) {
  if (on && !p->ph_ev[0])
    for (int i = 0; i < 3; i++)
      if (hipEventCreate(&p->ph_ev[i]) != hipSuccess) return LDE_ERR_HIP;
  p->phase_on = on != 0;
  return LDE_OK;
}
// ms of the last lde_adjoint's phases: [0] the solve kernel(s), [1] what follows it (weight-gradient product, slab / row sums)
int mlp_get_phase_ms(MlpPlan* p, float* out) {
  if (!p->phase_on || !p->ph_ev[0]) return LDE_ERR_INVALID_ARG;
  if (hipEventSynchronize(p->ph_ev[2]) != hipSuccess) return LDE_ERR_HIP;
  if (hipEventElapsedTime(&out[0], p->ph_ev[0], p->ph_ev[1]) != hipSuccess || hipEventElapsedTime(&out[1], p->ph_ev[1], p->ph_ev[2]) != hipSuccess)
    return LDE_ERR_HIP;
  return LDE_OK;
}
static void phase_mark(MlpPlan* p, int i, hipStream_t stream) {
  if (p->phase_on) (void)hipEventRecord(p->ph_ev[i], stream);
}

int mlp_set_sum_hook(MlpPlan* p, lde_sum_hook hook, void* user, int64_t global_batch, std::string& err) {
  if (!p->mbox) {
    if (hipHostMalloc((void**)&p->mbox, 16 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipMalloc(&p->mbox_dev, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreateWithFlags(&p->mbox_ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      err = "lde_set_global_sum_hook: allocating the mailbox failed";
      return LDE_ERR_ALLOC;
    }
  }
  p->sum_hook = hook;
  p->sum_user = user;
  p->global_batch = global_batch;
  return LDE_OK;
}

size_t mlp_sum_mailbox_bytes(int nranks) { return (size_t)4 * (size_t)std::max(nranks, 1) * 2 * sizeof(unsigned long long); }   // (launch parity, sum parity) × ranks × 2 words
int mlp_set_sum_peers(MlpPlan* p, int rank, int nranks, void* const* boxes, int64_t global_batch, std::string& err) {
  if (nranks == 0) {   // off
    p->peer_n = 0;
    return LDE_OK;
  }
  if (nranks < 1 || nranks > LDE_MAX_PEERS || rank < 0 || rank >= nranks || !boxes || global_batch < 1) {
    err = "lde_set_global_sum_peers: 1 … 8 ranks, 0 ≤ rank < nranks, a mailbox pointer per rank, global_batch ≥ 1";
    return LDE_ERR_INVALID_ARG;
  }
  for (int r = 0; r < nranks; r++)
    if (!boxes[r] || (((uintptr_t)boxes[r]) & 7) != 0) {
      err = "lde_set_global_sum_peers: NULL or unaligned mailbox pointer";
      return LDE_ERR_INVALID_ARG;
    }
  if (!p->mbox_dev && hipMalloc(&p->mbox_dev, 16 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    err = "lde_set_global_sum_peers: allocating the republish words failed";
    return LDE_ERR_ALLOC;
  }
  p->sum_hook = nullptr;   // one exchange path at a time
  p->peer_launch = 0;      // a fresh set of (zeroed) mailboxes: every rank starts its count with them
  p->peer_n = nranks;
  p->peer_rank = rank;
  for (int r = 0; r < nranks; r++) p->peer_box[r] = (unsigned long long*)boxes[r];
  p->global_batch = global_batch;
  return LDE_OK;
}

// LDE_BATCH_COUPLED_GLOBAL with a hook: arm the mailbox before the launch …
static int global_arm(MlpPlan* p, VArgs& a, hipStream_t stream, std::string& err) {
  a.gs.host_req = nullptr;
  a.gs.host_rep = nullptr;
  a.gs.dev_rep = nullptr;
  a.gs.nranks = 0;
  a.gs.rank = 0;
  a.gs.xlaunch = 0;
  a.gs.xspin_k = p->tune.peer_spin_k > 0 ? p->tune.peer_spin_k : 8192;
  a.Bnorm = 0;
  if (p->global_mode && p->peer_n > 0) {   // device to device: nothing for the host to serve, the call stays asynchronous
    if (hipMemsetAsync(p->mbox_dev, 0, 16 * sizeof(unsigned long long), stream) != hipSuccess) {
      err = "hipMemsetAsync(mailbox) failed";
      return LDE_ERR_HIP;
    }
    a.gs.nranks = p->peer_n;
    a.gs.rank = p->peer_rank;
    p->peer_launch = (p->peer_launch + 1) & 0xffffu;
    a.gs.xlaunch = p->peer_launch;
    for (int r = 0; r < p->peer_n; r++) a.gs.peer[r] = p->peer_box[r];
    a.gs.dev_rep = p->mbox_dev;
    a.Bnorm = p->global_batch;
    return LDE_OK;
  }
  if (!p->global_mode || !p->sum_hook) return LDE_OK;
  for (int i = 0; i < 16; i++) p->mbox[i] = 0;
  if (hipMemsetAsync(p->mbox_dev, 0, 16 * sizeof(unsigned long long), stream) != hipSuccess) {
    err = "hipMemsetAsync(mailbox) failed";
    return LDE_ERR_HIP;
  }
  unsigned long long* dptr = nullptr;
  if (hipHostGetDevicePointer((void**)&dptr, p->mbox, 0) != hipSuccess) {
    err = "hipHostGetDevicePointer(mailbox) failed";
    return LDE_ERR_HIP;
  }
  a.gs.host_req = dptr;
  a.gs.host_rep = dptr + 8;
  a.gs.dev_rep = p->mbox_dev;
  a.Bnorm = p->global_batch;
  return LDE_OK;
}
// … and serve it until the kernel has finished: every request {count, tag} is answered with the sums over all ranks
static int global_serve(MlpPlan* p, hipStream_t stream, std::string& err) {
  if (!p->global_mode || !p->sum_hook) return LDE_OK;
  if (hipEventRecord(p->mbox_ev, stream) != hipSuccess) {
    err = "hipEventRecord(mailbox) failed";
    return LDE_ERR_HIP;
  }
  volatile unsigned long long* mb = p->mbox;
  unsigned last_tag = 0;
  int rc = LDE_OK;
  for (;;) {
    const unsigned long long c = __atomic_load_n(&mb[2], __ATOMIC_ACQUIRE);
    const unsigned tag = (unsigned)(c >> 32);
    if (tag != 0 && tag != last_tag) {
      const int n = (int)(c & 0xffffffffu) == 2 ? 2 : 1;
      const unsigned long long w0 = __atomic_load_n(&mb[0], __ATOMIC_RELAXED), w1 = __atomic_load_n(&mb[1], __ATOMIC_RELAXED);
      if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) {
        auto f = [](unsigned long long w) { unsigned u = (unsigned)w; float v; std::memcpy(&v, &u, 4); return (double)v; };
        double vals[2] = {f(w0), f(w1)};
        if (rc != LDE_OK) {   // after a failure the hook is still called, with NaN: peers that are still in the exchange receive NaN sums and
                              // end their solves with retcode != 0 too, instead of waiting for a rank that has stopped answering
          vals[0] = vals[1] = std::nan("");
          (void)p->sum_hook(p->sum_user, vals, n);
        } else if (p->sum_hook(p->sum_user, vals, n) != 0) {
          err = "LDE_BATCH_COUPLED_GLOBAL: the sum hook reported an error";
          rc = LDE_ERR_INVALID_ARG;
        }
        if (rc != LDE_OK) vals[0] = vals[1] = std::nan("");   // poisons the solve: it ends with retcode != 0 on this rank
        auto g = [&](double v) { float x = (float)v; unsigned u; std::memcpy(&u, &x, 4); return ((unsigned long long)tag << 32) | u; };
        __atomic_store_n(&mb[9], g(n == 2 ? vals[1] : 0.0), __ATOMIC_RELAXED);
        __atomic_store_n(&mb[8], g(vals[0]), __ATOMIC_RELEASE);
        last_tag = tag;
        continue;
      }
    }
    const hipError_t q = hipEventQuery(p->mbox_ev);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) {
      err = std::string("LDE_BATCH_COUPLED_GLOBAL: ") + hipGetErrorString(q);
      return LDE_ERR_HIP;
    }
    (void)hipGetLastError();
  }
  return rc;
}

// Control words and staging weights a call starts from zero: ONE launch for all of them. As hipMemsetAsync calls they were up to eight
// dependent ≈ 5 µs fill kernels in front of a coupled solve (rocprofv3: 10.7 fills per c4 step, 55 µs of a 2 ms step).
struct ZeroRegions {
  uint32_t* p[6];
  unsigned long long n[6];   // 32-bit words
};
__global__ void __launch_bounds__(256) k_zero_regions(ZeroRegions z) {
  for (int r = 0; r < 6; r++) {
    uint32_t* q = z.p[r];
    const unsigned long long n = z.n[r];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) q[i] = 0u;
  }
}
static bool zero_regions(hipStream_t stream, std::initializer_list<std::pair<void*, size_t>> regs) {
  ZeroRegions z{};
  int r = 0;
  unsigned long long most = 0;
  for (const auto& pr : regs) {
    if (!pr.first || pr.second == 0) continue;
    if (r == 6 || (pr.second & 3) || ((uintptr_t)pr.first & 3)) return false;
    z.p[r] = (uint32_t*)pr.first;
    z.n[r] = pr.second / 4;
    most = std::max(most, z.n[r]);
    r++;
  }
  if (r == 0) return true;
  const int grid = (int)std::min<unsigned long long>(1024, (most + 255) / 256);
  hipLaunchKernelGGL(k_zero_regions, dim3(grid), dim3(256), 0, stream, z);
  return hipGetLastError() == hipSuccess;
}

int mlp_forward(MlpPlan* p, const float* W_dev, const float* z0, const float* theta, const double* ts_dev,
                const KOpts& o, float* z_out, int32_t* retcode, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                hipStream_t stream, std::string& err) {
  const MlpDims& dm = p->dm;
  const bool recording = o.rec.n != nullptr;   // the step record is written by k_mlp64 and by k_mlp_forward (the tiles)
  if (mlp64_applicable(p, o.B)) {   // small network, small state: one wave per trajectory, registers only (lde_mlp64.h)
    VArgs va{};
    va.z0 = z0; va.theta = theta; va.ts = ts_dev; va.Wflat = W_dev; va.z_out = z_out; va.retcode = retcode;
    va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
    return launch_mlp64<false>(dm, o, va, stream, err);
  }
  {   // small batches: one trajectory per workgroup, lanes = hidden units (lde_mlpv.h)
    size_t ldsv = 0;
    const bool ca = dm.coupled && o.adaptive && o.B > 1;
    const bool use_b = b_applicable(p, o.B, o.T, false, ca);
    const bool use_c = !use_b && c_applicable(p, o.B, o.T, false, ca);
    const bool use_w = use_b || use_c || (!recording && w_applicable(p, o.B, o.T, false, ca));   // (k_mlpw / k_mlpv write no step record)
    if (p->global_mode && !use_w) {
      err = "LDE_BATCH_COUPLED_GLOBAL: this shape / batch is not served by the register kernels (three Dense layers, 2·D' ≤ 64, H ≤ 200, B·W ≤ 1024 waves)";
      return LDE_ERR_UNSUPPORTED;
    }
    if (use_w || (!recording && vec_applicable(p, o.B, o.T, false, ca, &ldsv, err))) {
      VArgs va{};
      va.z0 = z0; va.theta = theta; va.ts = ts_dev; va.vecw = p->vecw; va.Wflat = W_dev; va.z_out = z_out; va.retcode = retcode;
      va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
      va.gs.counter = p->counter; va.gs.slots = p->slots; va.gs.abort_flag = p->abort_flag; va.gs.nwg = ca ? o.B : 1;
      if (ca && !zero_regions(stream, {{p->counter, sizeof(unsigned)}, {p->abort_flag, sizeof(int)}})) {
        err = "k_zero_regions(counter) failed";
        return LDE_ERR_HIP;
      }
      if (use_w) {
        const int rca = global_arm(p, va, stream, err);
        if (rca) return rca;
        const int rcw = use_b ? (dm.solver == LDE_SOLVER_RK4 ? launch_b<LDE_SOLVER_RK4, false>(p, o, va, ca, stream, err)
                                                             : launch_b<LDE_SOLVER_TSIT5, false>(p, o, va, ca, stream, err))
                        : use_c ? (dm.solver == LDE_SOLVER_RK4 ? launch_c<LDE_SOLVER_RK4, false>(p, o, va, ca, stream, err)
                                                               : launch_c<LDE_SOLVER_TSIT5, false>(p, o, va, ca, stream, err))
                              : (dm.solver == LDE_SOLVER_RK4 ? launch_w<LDE_SOLVER_RK4, false>(p, o, va, ca, stream, err)
                                                             : launch_w<LDE_SOLVER_TSIT5, false>(p, o, va, ca, stream, err));
        return rcw ? rcw : global_serve(p, stream, err);
      }
      return dm.solver == LDE_SOLVER_RK4 ? launch_vec<LDE_SOLVER_RK4, false>(p, o, va, ldsv, ca, stream, err)
                                         : launch_vec<LDE_SOLVER_TSIT5, false>(p, o, va, ldsv, ca, stream, err);
    }
  }
  const int nwg = cdiv(o.B, NB);
  const bool sync = dm.coupled && o.adaptive && nwg > 1;
  if (sync && nwg > 256) {
    err = "coupled adaptive solve: batch per GPU limited to 4096 trajectories (one resident workgroup per CU)";
    return LDE_ERR_UNSUPPORTED;
  }
  constexpr int NTF = 512;   // forward: 8 waves (2 per SIMD) — the second wave hides the first one's LDS/barrier waits
  const size_t fixed = fwd_lds_fixed(dm, o.T, NTF);
  if (fixed > LDS_MAX) {
    err = "MLP forward: tile state does not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  const bool rk4 = dm.solver == LDE_SOLVER_RK4;
  const void* kfn = rk4 ? (const void*)k_mlp_forward<LDE_SOLVER_RK4, NTF> : (const void*)k_mlp_forward<LDE_SOLVER_TSIT5, NTF>;
  static bool attr_set[2] = {false, false};
  const int ki = rk4 ? 1 : 0;
  if (!attr_set[ki]) {
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_forward) failed";
      return LDE_ERR_HIP;
    }
    attr_set[ki] = true;
  }
  FwdArgs a;
  a.z0 = z0; a.theta = theta; a.ts = ts_dev; a.frag = p->frag; a.Wflat = W_dev; a.z_out = z_out; a.retcode = retcode;
  a.st_nfe = nfe; a.st_nacc = nacc; a.st_nrej = nrej; a.st_ret = ret;
  a.gs.counter = p->counter; a.gs.slots = p->slots; a.gs.abort_flag = p->abort_flag; a.gs.nwg = sync ? nwg : 1;
  const size_t lds = with_cache(fixed, p->nfrag);
  a.lds_bytes = (int)lds;
  if (sync && !zero_regions(stream, {{p->counter, sizeof(unsigned)}, {p->abort_flag, sizeof(int)}})) {
    err = "k_zero_regions(counter) failed";
    return LDE_ERR_HIP;
  }
#if LDE_PROF
  prof_reset();
#endif
  MlpDims dmv = dm;
  KOpts ov = o;
  const int rcl = launch_maybe_coop(sync, kfn, dim3(nwg), dim3(NTF), lds, stream, err, "k_mlp_forward", dmv, ov, a);
#if LDE_PROF
  prof_dump("forward", stream);
#endif
  return rcl;
}

template <int SOLVER>
static int launch_adjoint(MlpPlan* p, const KOpts& o, const BwdArgs& a, int nwg, size_t lds, hipStream_t stream,
                          std::string& err, bool coop) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_mlp_adjoint<SOLVER, 512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_adjoint) failed";
      return LDE_ERR_HIP;
    }
    attr_set = true;
  }
  MlpDims dmv = p->dm;
  KOpts ov = o;
  BwdArgs av = a;
  return launch_maybe_coop(coop, (const void*)k_mlp_adjoint<SOLVER, 512>, dim3(nwg), dim3(512), lds, stream, err, "k_mlp_adjoint", dmv, ov, av);
}

// ---- the 4-columns-per-wave adjoint (lde_mlp4.h): applicability, LDS layout, launch --------------------------------
static bool mlp4_layout(const MlpTune& tune, const MlpDims& dm, int T, int B, bool coupled_adaptive, Mlp4Dims* md, size_t* lds, int* nblocks) {
  if (!tune.mlp4 || dm.Dp > 64 || dm.P > 1) return false;
  int maxw = 0;
  for (int l = 0; l <= dm.nL; l++) maxw = std::max(maxw, dm.sizes[l]);
  // measured (MI355X): one wave has ONE SIMD's matrix pipe and v_mfma_f32_4x4x1 costs 11 cycles per 256 MACs (the
  // 16x16x4 form: 8), so the 4-column kernel only wins while the layers are small enough for the 16-column kernel's
  // fixed ≈ 2 000 cycles per layer to dominate: config 3 (64 wide) 7.3 → 5.3 ms, config 4 (128 wide) 4.4 → 5.5 ms.
  const int maxw_lim = tune.mlp4_maxw;
  if (maxw > maxw_lim || maxw > 256) return false;
  int off = 0;
  for (int l = 0; l < dm.nL; l++) {
    int v = (dm.sizes[l] + 3) & ~3;
    if (((v >> 2) & 1) == 0) v += 4;   // (ldw/4) odd ⇒ the row-segment reads of 16 consecutive rows hit distinct banks
    md->ldw[l] = v;
    md->wl_off[l] = off;
    off += dm.sizes[l + 1] * v;
  }
  for (int l = 0; l < dm.nL; l++) {
    md->bl_off[l] = off;
    off += (dm.sizes[l + 1] + 63) & ~63;
  }
  md->w_total = off;
  md->ldx = ((maxw + 63) & ~63) + 8;   // +8: the 4 columns of a panel start 8 banks apart (broadcast b128 reads, b128 writes: conflict-free)
  const int nwaves = cdiv(B, 4);
  int wpb = cdiv(nwaves, 256);
  wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
  md->wpb = wpb;
  *nblocks = cdiv(nwaves, wpb);
  if (coupled_adaptive && *nblocks > 256) return false;   // grid-wide sums need every workgroup resident
  *lds = ((size_t)md->w_total + (size_t)wpb * (dm.nL + 1) * 4 * md->ldx + (size_t)wpb * 4 + 8) * sizeof(float) + (size_t)T * sizeof(double) + 64 + 8 * 264 * sizeof(float);   // tail slack: operand prefetch reads ≤ 8 rows past the end
  return *lds <= LDS_MAX;
}

template <int SOLVER, int NTH>
static int launch_mlp4(const MlpDims& dm, const Mlp4Dims& md, const KOpts& o, const BwdArgs& a, int nblocks, size_t lds,
                       hipStream_t stream, std::string& err, bool coop) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_mlp4_adjoint<SOLVER, NTH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp4_adjoint) failed";
      return LDE_ERR_HIP;
    }
    attr_set = true;
  }
  MlpDims dmv = dm;
  Mlp4Dims mdv = md;
  KOpts ov = o;
  BwdArgs av = a;
  return launch_maybe_coop(coop, (const void*)k_mlp4_adjoint<SOLVER, NTH>, dim3(nblocks), dim3(64 * md.wpb), lds, stream, err,
                           "k_mlp4_adjoint", dmv, mdv, ov, av);
}

// LDE_SENSE_DISCRETE: the reverse sweep over the forward solve's step record (lde_mlpd.h), then the weight gradient from the staged panels
static int mlp_adjoint_disc(MlpPlan* p, const float* W_dev, const float* z_out, const float* theta, const double* ts_dev,
                            const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* dW, int32_t* nfe, int32_t* nacc,
                            int32_t* nrej, int32_t* ret, hipStream_t stream, std::string& err) {
  const MlpDims& dm = p->dm;
  if (!o.rec.n || !o.rec.y) {
    err = "MLP adjoint (LDE_SENSE_DISCRETE): no step record";
    return LDE_ERR_INVALID_ARG;
  }
  const int fam = disc_family(p, o.B, o.T);
  p->last_family = fam == DISC_B ? 2 : (fam == DISC_C ? 3 : 0);
  if (fam == DISC_B || fam == DISC_C) {   // W₂ as register blocks, the weight gradient folded on the CU: two launches, no staging area, no grid-wide sum
    const int nrows = fam == DISC_B ? o.B : (o.B + 1) / 2;
    if (!p->rows || p->rows_cap < (size_t)o.B * p->rows_stride || p->rows_stride < dm.nW) {
      err = "MLP adjoint: workspace not reserved";
      return LDE_ERR_INVALID_ARG;
    }
    VArgs va{};
    va.theta = theta; va.ts = ts_dev; va.Wflat = W_dev; va.z_out = const_cast<float*>(z_out); va.dz_out = dz_out;
    va.dz0 = dz0; va.dtheta = dtheta; va.stage = p->rows; va.cap = p->rows_stride;
    va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
    va.gs.counter = p->counter; va.gs.slots = p->slots; va.gs.abort_flag = p->abort_flag; va.gs.nwg = 1;
    va.gs.host_req = nullptr; va.gs.host_rep = nullptr; va.gs.dev_rep = nullptr; va.gs.nranks = 0; va.gs.rank = 0; va.gs.xlaunch = 0; va.gs.xspin_k = 8192; va.Bnorm = 0;
    phase_mark(p, 0, stream);
    const bool rk4 = dm.solver == LDE_SOLVER_RK4;
    const int rcb = fam == DISC_B ? (rk4 ? launch_b<LDE_SOLVER_RK4, true, true>(p, o, va, false, stream, err)
                                         : launch_b<LDE_SOLVER_TSIT5, true, true>(p, o, va, false, stream, err))
                                  : (rk4 ? launch_c<LDE_SOLVER_RK4, true, true>(p, o, va, false, stream, err)
                                         : launch_c<LDE_SOLVER_TSIT5, true, true>(p, o, va, false, stream, err));
    if (rcb) return rcb;
    phase_mark(p, 1, stream);
    if (dW) {
      hipLaunchKernelGGL(k_sum_rows, dim3(cdiv(dm.nW, 64)), dim3(1024), 0, stream, p->rows, nrows, p->rows_stride, dm.nW, dW, o.dw_overwrite);
      if (hipGetLastError() != hipSuccess) {
        err = "k_sum_rows launch failed";
        return LDE_ERR_HIP;
      }
    }
    phase_mark(p, 2, stream);
    return LDE_OK;
  }
  const int nwg = cdiv(o.B, NB);
  const size_t fixed = disc_lds_fixed(dm, o.T, 512);
  if (fixed > LDS_MAX) {
    err = "MLP adjoint (LDE_SENSE_DISCRETE): tile state does not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  if (!p->stage || p->adj_cap < 1) {
    err = "MLP adjoint: workspace not reserved";
    return LDE_ERR_INVALID_ARG;
  }
  BwdArgs a{};
  a.z_out = z_out; a.dz_out = dz_out; a.theta = theta; a.ts = ts_dev; a.frag = p->frag; a.fragT = p->fragT; a.Wflat = W_dev;
  a.dz0 = dz0; a.dtheta = dtheta; a.slab = p->slab;
  a.stage = p->stage; a.wts = p->wts; a.nslots = p->nslots; a.nflush = p->nslots + (nwg + 1); a.cap = p->adj_cap;
  a.ovf = p->fb_dev + 1; a.fallback = 0;
  a.st_nfe = nfe; a.st_nacc = nacc; a.st_nrej = nrej; a.st_ret = ret;
  a.gs.counter = p->counter; a.gs.slots = p->slots; a.gs.abort_flag = p->abort_flag; a.gs.nwg = 1;
  const size_t lds = with_cache(fixed, p->nfrag + p->nfragT);
  a.lds_bytes = (int)lds;
  if (!zero_regions(stream, {{p->fb_dev, 2 * sizeof(int32_t)}, {p->nslots, (size_t)2 * (nwg + 1) * sizeof(int32_t)}})) {
    err = "k_zero_regions failed";
    return LDE_ERR_HIP;
  }
  const bool rk4 = dm.solver == LDE_SOLVER_RK4;
  const void* fn = rk4 ? (const void*)k_mlp_adjoint_disc<LDE_SOLVER_RK4, 512> : (const void*)k_mlp_adjoint_disc<LDE_SOLVER_TSIT5, 512>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[rk4]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_adjoint_disc) failed";
      return LDE_ERR_HIP;
    }
    attr_set[rk4] = true;
  }
  MlpDims dmv = dm;
  KOpts ov = o;
  phase_mark(p, 0, stream);
  int rc = launch_maybe_coop(false, fn, dim3(nwg), dim3(512), lds, stream, err, "k_mlp_adjoint_disc", dmv, ov, a);
  if (rc) return rc;
  phase_mark(p, 1, stream);
  DwArgs da;
  da.stage = p->stage; da.wts = p->wts; da.nslots = p->nslots; da.slab = p->slab + (size_t)(nwg + 1) * dm.slab_n; da.cap = p->adj_cap; da.total = 0;
  if (dW) {
    rc = launch_weight_gradient(dm, da, nwg, p->adj_ks, p->slab, p->nslots + (nwg + 1), nwg, dW, p->fb_dev, stream, err, o.dw_overwrite != 0);
    if (rc) return rc;
  }
  phase_mark(p, 2, stream);
  return LDE_OK;
}

int mlp_adjoint(MlpPlan* p, const float* W_dev, const float* z_out, const float* theta, const double* ts_dev,
                const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* dW, int32_t* nfe, int32_t* nacc,
                int32_t* nrej, int32_t* ret, hipStream_t stream, std::string& err) {
  const MlpDims& dm = p->dm;
  if (p->disc && !mlp64_applicable(p, o.B))
    return mlp_adjoint_disc(p, W_dev, z_out, theta, ts_dev, o, dz_out, dz0, dtheta, dW, nfe, nacc, nrej, ret, stream, err);
  if (mlp64_applicable(p, o.B)) {   // one wave per trajectory, registers only, the weight gradient included (lde_mlp64.h): two launches
    p->last_family = 1;
    const int waves = mlp64_adj_waves(o.B);
    if (!p->rows || p->rows_cap < (size_t)waves * p->rows_stride || p->rows_stride < dm.nW) {
      err = "MLP adjoint: workspace not reserved";
      return LDE_ERR_INVALID_ARG;
    }
    VArgs va{};
    va.theta = theta; va.ts = ts_dev; va.Wflat = W_dev; va.z_out = const_cast<float*>(z_out); va.dz_out = dz_out;
    va.dz0 = dz0; va.dtheta = dtheta; va.stage = p->rows; va.cap = p->rows_stride;
    va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
    phase_mark(p, 0, stream);
    if (p->disc && (!o.rec.n || !o.rec.y)) {
      err = "MLP adjoint (LDE_SENSE_DISCRETE): no step record";
      return LDE_ERR_INVALID_ARG;
    }
    const int rc6 = p->disc ? launch_mlp64<true, true>(dm, o, va, stream, err) : launch_mlp64<true>(dm, o, va, stream, err);
    if (rc6) return rc6;
    phase_mark(p, 1, stream);
    if (dW) {
      hipLaunchKernelGGL(k_sum_rows, dim3(cdiv(dm.nW, 64)), dim3(1024), 0, stream, p->rows, waves, p->rows_stride, dm.nW, dW, o.dw_overwrite);
      if (hipGetLastError() != hipSuccess) {
        err = "k_sum_rows launch failed";
        return LDE_ERR_HIP;
      }
    }
    phase_mark(p, 2, stream);
    return LDE_OK;
  }
  {   // W₂ as register blocks, the weight gradient folded on the CU (lde_mlpb.h): two launches, no staging area
    const bool ca = dm.coupled && o.adaptive && o.B > 1;
    const bool use_b = b_applicable(p, o.B, o.T, true, dm.coupled != 0);
    const bool use_c = !use_b && c_applicable(p, o.B, o.T, true, dm.coupled != 0);
    if (use_b || use_c) {
      p->last_family = use_b ? 2 : 3;
      const int nrows = use_b ? o.B : (o.B + 1) / 2;
      if (!p->rows || p->rows_cap < (size_t)o.B * p->rows_stride || p->rows_stride < dm.nW) {
        err = "MLP adjoint: workspace not reserved";
        return LDE_ERR_INVALID_ARG;
      }
      VArgs va{};
      va.theta = theta; va.ts = ts_dev; va.Wflat = W_dev; va.z_out = const_cast<float*>(z_out); va.dz_out = dz_out;
      va.dz0 = dz0; va.dtheta = dtheta; va.stage = p->rows; va.cap = p->rows_stride;
      va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
      va.gs.counter = p->counter; va.gs.slots = p->slots; va.gs.abort_flag = p->abort_flag; va.gs.nwg = ca ? o.B : 1;
      if (ca && !zero_regions(stream, {{p->counter, sizeof(unsigned)}, {p->abort_flag, sizeof(int)}})) {
        err = "k_zero_regions(counter) failed";
        return LDE_ERR_HIP;
      }
      const int rca = global_arm(p, va, stream, err);
      if (rca) return rca;
      phase_mark(p, 0, stream);
      int rcb = use_b ? (dm.solver == LDE_SOLVER_RK4 ? launch_b<LDE_SOLVER_RK4, true>(p, o, va, ca, stream, err)
                                                     : launch_b<LDE_SOLVER_TSIT5, true>(p, o, va, ca, stream, err))
                      : (dm.solver == LDE_SOLVER_RK4 ? launch_c<LDE_SOLVER_RK4, true>(p, o, va, ca, stream, err)
                                                     : launch_c<LDE_SOLVER_TSIT5, true>(p, o, va, ca, stream, err));
      if (!rcb) rcb = global_serve(p, stream, err);
      if (rcb) return rcb;
      phase_mark(p, 1, stream);
      if (dW) {
        hipLaunchKernelGGL(k_sum_rows, dim3(cdiv(dm.nW, 64)), dim3(1024), 0, stream, p->rows, nrows, p->rows_stride, dm.nW, dW, o.dw_overwrite);
        if (hipGetLastError() != hipSuccess) {
          err = "k_sum_rows launch failed";
          return LDE_ERR_HIP;
        }
      }
      phase_mark(p, 2, stream);
      return LDE_OK;
    }
  }
  const int nwg = cdiv(o.B, NB);
  const bool sync = dm.coupled && o.adaptive && nwg > 1;
  if (sync && nwg > 256) {
    err = "coupled adaptive solve: batch per GPU limited to 4096 trajectories (one resident workgroup per CU)";
    return LDE_ERR_UNSUPPORTED;
  }
  const size_t fixed = bwd_lds_fixed(dm, o.T, 512);
  if (fixed > LDS_MAX) {
    err = "MLP adjoint: tile state does not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  if (!p->stage || p->adj_cap < 1) {
    err = "MLP adjoint: workspace not reserved";
    return LDE_ERR_INVALID_ARG;
  }
  const int ks = p->adj_ks;
  BwdArgs a;
  a.z_out = z_out; a.dz_out = dz_out; a.theta = theta; a.ts = ts_dev; a.frag = p->frag; a.fragT = p->fragT; a.Wflat = W_dev;
  a.dz0 = dz0; a.dtheta = dtheta; a.slab = p->slab;
  a.stage = p->stage; a.wts = p->wts; a.nslots = p->nslots; a.nflush = p->nslots + (nwg + 1); a.cap = p->adj_cap;
  a.ovf = p->fb_dev + 1; a.fallback = 0;
  a.st_nfe = nfe; a.st_nacc = nacc; a.st_nrej = nrej; a.st_ret = ret;
  a.gs.counter = p->counter; a.gs.slots = p->slots; a.gs.abort_flag = p->abort_flag; a.gs.nwg = sync ? nwg : 1;
  const size_t lds = with_cache(fixed, p->nfrag + p->nfragT);
  a.lds_bytes = (int)lds;
  bool ctl_zeroed = false;   // counter + abort flag of the grid-wide sums: zeroed together with whatever else the first solve kernel needs
#if LDE_PROF
  prof_reset();
#endif
  int ntile_dw = nwg;
  bool vec_done = false;
  p->last_family = 0;
  if (!vec_done) {   // small batches: one trajectory per workgroup, lanes = hidden units (lde_mlpv.h)
    size_t ldsv = 0;
    const bool ca = dm.coupled && o.adaptive && o.B > 1;
    const bool use_w = w_applicable(p, o.B, o.T, true, ca);
    if (p->global_mode && !use_w) {
      err = "LDE_BATCH_COUPLED_GLOBAL: this shape / batch is not served by the register kernels (three Dense layers, 2·D' ≤ 64, H ≤ 200, B·W ≤ 1024 waves)";
      return LDE_ERR_UNSUPPORTED;
    }
    if (use_w || vec_applicable(p, o.B, o.T, true, ca, &ldsv, err)) {
      p->last_family = use_w ? 4 : 5;
      if (!zero_regions(stream, {{p->fb_dev, 2 * sizeof(int32_t)}, {p->nslots, (size_t)2 * (nwg + 1) * sizeof(int32_t)},
                                 {p->wts, (size_t)nwg * p->adj_cap * NB * sizeof(float)},
                                 {(ca || sync) ? p->counter : nullptr, sizeof(unsigned)}, {(ca || sync) ? p->abort_flag : nullptr, sizeof(int)}})) {
        err = "k_zero_regions(staging weights) failed";
        return LDE_ERR_HIP;
      }
      ctl_zeroed = true;
      VArgs va{};
      va.theta = theta; va.ts = ts_dev; va.vecw = p->vecw; va.Wflat = W_dev; va.z_out = const_cast<float*>(z_out); va.dz_out = dz_out;
      va.dz0 = dz0; va.dtheta = dtheta; va.stage = p->stage; va.wts = p->wts; va.nslots = p->nslots; va.cap = p->adj_cap; va.ovf = p->fb_dev + 1;
      va.st_nfe = nfe; va.st_nacc = nacc; va.st_nrej = nrej; va.st_ret = ret;
      va.gs.counter = p->counter; va.gs.slots = p->slots; va.gs.abort_flag = p->abort_flag; va.gs.nwg = ca ? o.B : 1;
      if (use_w) {
        const int rca = global_arm(p, va, stream, err);
        if (rca) return rca;
      }
      phase_mark(p, 0, stream);
      int rcv = use_w ? (dm.solver == LDE_SOLVER_RK4 ? launch_w<LDE_SOLVER_RK4, true>(p, o, va, ca, stream, err)
                                                     : launch_w<LDE_SOLVER_TSIT5, true>(p, o, va, ca, stream, err))
                      : (dm.solver == LDE_SOLVER_RK4 ? launch_vec<LDE_SOLVER_RK4, true>(p, o, va, ldsv, ca, stream, err)
                                                     : launch_vec<LDE_SOLVER_TSIT5, true>(p, o, va, ldsv, ca, stream, err));
      if (!rcv && use_w) rcv = global_serve(p, stream, err);
      if (rcv) return rcv;
      phase_mark(p, 1, stream);
      // a trajectory that ran out of staging slots sets *ovf: k_mlp_adjoint (which folds its slots into a private slab) then
      // redoes the whole call; otherwise it returns at once. The decision is taken on the device.
      vec_done = true;
      a.fallback = 1;
      if (sync && hipMemsetAsync(p->counter, 0, sizeof(unsigned), stream) != hipSuccess) {
        err = "hipMemsetAsync(counter) failed";
        return LDE_ERR_HIP;
      }
    }
  }
  if (!vec_done) phase_mark(p, 0, stream);
  // networks whose weights fit LDS once: four columns per wave, no barriers (lde_mlp4.h)
  if (!vec_done) {
    Mlp4Dims md;
    size_t lds4 = 0;
    int nblocks = 0;
    if (mlp4_layout(p->tune, dm, o.T, o.B, dm.coupled && o.adaptive, &md, &lds4, &nblocks)) {
      const bool sync4 = dm.coupled && o.adaptive && nblocks > 1;
      p->last_family = 6;
      const int ntile4 = cdiv(nblocks * md.wpb, 4);   // ≤ nwg + 1: the workspace is sized for that
      if (!zero_regions(stream, {{p->fb_dev, 2 * sizeof(int32_t)}, {p->nslots, (size_t)2 * (nwg + 1) * sizeof(int32_t)},
                                 {p->wts, (size_t)ntile4 * p->adj_cap * NB * sizeof(float)},
                                 {(sync4 || sync) ? p->counter : nullptr, sizeof(unsigned)}, {(sync4 || sync) ? p->abort_flag : nullptr, sizeof(int)}})) {
        err = "k_zero_regions(staging weights) failed";
        return LDE_ERR_HIP;
      }
      ctl_zeroed = true;
      a.gs.nwg = sync4 ? nblocks : 1;
      int hmaxw = 0;
      for (int l = 1; l < dm.nL; l++) hmaxw = std::max(hmaxw, dm.sizes[l]);
      const int nth = hmaxw <= 64 ? 1 : (hmaxw <= 128 ? 2 : 4);
      const bool rk4 = dm.solver == LDE_SOLVER_RK4;
      int rc4;
      if (nth == 1) rc4 = rk4 ? launch_mlp4<LDE_SOLVER_RK4, 1>(dm, md, o, a, nblocks, lds4, stream, err, sync4) : launch_mlp4<LDE_SOLVER_TSIT5, 1>(dm, md, o, a, nblocks, lds4, stream, err, sync4);
      else if (nth == 2) rc4 = rk4 ? launch_mlp4<LDE_SOLVER_RK4, 2>(dm, md, o, a, nblocks, lds4, stream, err, sync4) : launch_mlp4<LDE_SOLVER_TSIT5, 2>(dm, md, o, a, nblocks, lds4, stream, err, sync4);
      else rc4 = rk4 ? launch_mlp4<LDE_SOLVER_RK4, 4>(dm, md, o, a, nblocks, lds4, stream, err, sync4) : launch_mlp4<LDE_SOLVER_TSIT5, 4>(dm, md, o, a, nblocks, lds4, stream, err, sync4);
      if (rc4) return rc4;
      // A wave that ran out of staging slots sets *ovf: k_mlp_adjoint (which can fold its slots into a private slab) then
      // redoes the whole call; otherwise it returns at once. The host never waits: the decision is taken on the device.
      ntile_dw = cdiv(nblocks * md.wpb, 4);
      a.fallback = 1;
      a.gs.nwg = sync ? nwg : 1;
      if (sync && hipMemsetAsync(p->counter, 0, sizeof(unsigned), stream) != hipSuccess) {
        err = "hipMemsetAsync(counter) failed";
        return LDE_ERR_HIP;
      }
    }
  }
  if (sync && !ctl_zeroed && !zero_regions(stream, {{p->counter, sizeof(unsigned)}, {p->abort_flag, sizeof(int)}})) {
    err = "k_zero_regions(counter) failed";
    return LDE_ERR_HIP;
  }
  int rc = dm.solver == LDE_SOLVER_RK4 ? launch_adjoint<LDE_SOLVER_RK4>(p, o, a, nwg, lds, stream, err, sync)
                                       : launch_adjoint<LDE_SOLVER_TSIT5>(p, o, a, nwg, lds, stream, err, sync);
#if LDE_PROF
  prof_dump("adjoint", stream);
#endif
  if (rc) return rc;
  // the weight gradient from the staged panels
  DwArgs da;
  da.stage = p->stage; da.wts = p->wts; da.nslots = p->nslots; da.slab = p->slab + (size_t)(nwg + 1) * dm.slab_n; da.cap = p->adj_cap; da.total = 0;
  if (!vec_done) phase_mark(p, 1, stream);   // (the tile kernels: the solve is the launch above; the small-batch kernels marked theirs already)
  rc = launch_weight_gradient(dm, da, ntile_dw, ks, p->slab, p->nslots + (nwg + 1), nwg, dW, p->fb_dev, stream, err, o.dw_overwrite != 0);
  if (rc) return rc;
  phase_mark(p, 2, stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {   // not inside a hipGraph capture
    if (hipMemcpyAsync(p->fb_host, p->fb_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipEventRecord(p->fb_ev, stream) == hipSuccess)
      p->fb_pending = true;
  }
  return LDE_OK;
}

}  // namespace lde

// lde_rnn.hip — the recurrent pattern extractor (scope row f-2).
//
// Replaces what runs under
//     pe_out = apply_pattern_extractor(encoder, fe_out)      [REF src/models/GOKU.jl:32-51], [REF src/models/LatentODE.jl:24-33]
// i.e. stacks Chain(RNN(32,16,relu), RNN(16,16,relu)) / Chain(LSTM(32,16), LSTM(16,16)) [REF src/models/GOKU.jl:229-238]
// run over the T frames of fe_out [in×B×T] (forward or reversed), keeping the last output, and their pullback.
// Cells: Flux 0.13.6 RNNCell / LSTMCell (un-vendored [REF Manifest.toml:452]); equations and weight order in include/lde.h.
//
// Design (gfx950). The cells are tiny (16 hidden units, K = in + h = 48) and strictly sequential in time: this is
// latency-bound scalar work, not matrix-core work (north star: "MFMA only if latent_dim×hidden is large enough to fill
// a tile"). Hp = pow2(G·h) ≤ 64 lanes per trajectory (all lanes of a trajectory sit in one wave, so a step needs no
// barrier); lane (trajectory, row r) computes one gate pre-activation as the dot product of the LDS-resident row r of
// [Wi | Wh] with the trajectory's [x_t ; h_{t-1}] vector (16-byte LDS reads, broadcast within the trajectory), the first h
// lanes then combine the gates of their unit. The pullback recomputes the sweep with per-step records in HBM, walks it
// backwards (δ per gate, [d_in; dh_prev] = [Wi|Wh]ᵀδ from a transposed LDS copy so that it reads 16-byte rows too), and
// STAGES the (a, δ) = ([x_t;h_{t-1}], gate deltas) panels of every (step, layer) in the block layout of lde_mfma.h, so
// that the weight gradient — the only matrix-shaped part: K' = B·T columns — is formed by the shared large-K MFMA kernel
// k_mlp_dw, once per cell.
// What sets the speed: trajectories per workgroup chosen at launch (a small batch runs one wave per
// CU instead of sixteen waves queueing on one LDS), compile-time instantiations for the reference's default stacks (no
// per-layer kernel-argument loads, run-time loops or branches: they were ≈ 60 % of a step), and branch-free prefetch of the
// next frame / record (a load under a branch costs a vmcnt(0), which on gfx950 also drains the staging stores).
// (Measured alternative: 2 or 4 trajectories per wave sharing every weight read — a quarter of the LDS traffic, but fewer
// waves to hide the LDS latency: slower.)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lde_mfma.h"

namespace lde {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int RNN_ML = LDE_RNN_MAX_LAYERS;

struct RnnDims {
  int cell, nL, reverse, G;
  int sizes[RNN_ML + 1];
  int Hp;               // lanes per trajectory
  int K[RNN_ML];        // in_l + h_l
  int ldk[RNN_ML];      // row stride of [Wi | Wh] in LDS: ≥ pad4(K), (ldk/4) odd
  int w_off[RNN_ML];    // LDS float offsets: rows [G·h][ldk]
  int b_off[RNN_ML];    // bias [G·h]
  int s_off[RNN_ML];    // state0: h0 [h] (LSTM: then c0 [h])
  int f_off[RNN_ML];    // offset of the cell in the flat weight vector
  int wt;               // 1: LDS also holds the transposed copies [K][ldr] (the pullback's Wᵀδ then reads 16-byte rows too)
  int ldr[RNN_ML];      // their row stride: ≥ pad4(G·h), (ldr/4) odd
  int wt_off[RNN_ML];
  int lds_w;            // floats of the weight area
  int vmax;             // floats of one trajectory's [x; h] vector (pad4(max K) + 4)
  int rmax;             // floats of one trajectory's δ vector (pad4(max G·h))
  int hmax;
  int recw;             // floats of one (step, layer, trajectory) record: gates G·hmax | c hmax | h hmax
};

__device__ __forceinline__ float sigm(float x) { return fast_rcp(1.0f + __expf(-x)); }
// tanh x = 1 − 2/(1 + e^{2x}): v_exp_f32 + v_rcp_f32, ≈ 2e-7 absolute (saturates cleanly: e^{2x} → ∞ ⇒ 1, → 0 ⇒ −1)
// (Fused multiply-adds of the cell formulas are spelled out, here and below: left to the compiler, what is fused depends on the code
// around the expression — one body got v_fma(−t, t, 1), another a packed multiply and a subtraction — and the instantiations of the
// recurrent kernels promise each other the same bits.)
__device__ __forceinline__ float fast_tanh(float x) { return __builtin_fmaf(-2.0f, fast_rcp(1.0f + __expf(2.0f * x)), 1.0f); }

struct RnnArgs {
  const float* x;       // [in × B × T]
  const float* Wflat;
  float* y;             // forward: [hL × B]
  float* rec;           // training: [T][L][B][recw]
  const float* dy;      // backward
  float* dx;            // [in × B × T] or NULL
  float* stage[RNN_ML]; // per layer: [ntile][T][blk_l]
  int blk[RNN_ML];      // floats of one staged block of layer l: 16·(pad32(K) + pad32(G·h))
  float* wts;           // [ntile][T][16]
  float* g0;            // [B][Σ state0 sizes]: per-trajectory gradient of the initial states
  int g0w;              // floats per trajectory in g0
  int T, B, mode;       // mode 0: forward only (writes y); 1: backward (recompute + BPTT)
  int tpw;              // trajectories per workgroup: 1, 2, 4, 8 or 16 (a staging tile of 16 spans 16/tpw workgroups)
  int ldy, lddy;        // floats between the rows of y / dy (≥ hL: the stack writes, or reads, a column block of a wider [B × ·] array —
                        // the θ branch's vcat(pe_forward, pe_backward) [REF src/models/GOKU.jl:47] without a concatenation launch)
  const float* dy2;     // backward: the output gradient is dy + dy2 (nullptr: absent; same row stride)
};
__device__ __forceinline__ float rnn_dy(const RnnArgs& a, long long b, int u) {
  const float g = a.dy[(size_t)a.lddy * b + u];
  return a.dy2 ? g + a.dy2[(size_t)a.lddy * b + u] : g;
}

// weights → LDS: rows of [Wi | Wh] (row r = g·h + u), zero padded; biases; state0
__device__ __forceinline__ void rnn_load_weights(const RnnDims& rd, const float* Wflat, float* lw, int nthr, bool transposed) {
  const int tid = threadIdx.x;
  for (int i = tid; i < rd.lds_w; i += nthr) lw[i] = 0.f;
  __syncthreads();
  for (int l = 0; l < rd.nL; l++) {
    const int in = rd.sizes[l], h = rd.sizes[l + 1], R = rd.G * h, ldk = rd.ldk[l];
    const float* Wi = Wflat + rd.f_off[l];
    const float* Wh = Wi + (size_t)R * in;
    const float* b = Wh + (size_t)R * h;
    const float* s0 = b + R;
    for (int e = tid; e < R * in; e += nthr) { const int k = e / R, r = e - k * R; lw[rd.w_off[l] + r * ldk + k] = Wi[e]; }
    for (int e = tid; e < R * h; e += nthr) { const int k = e / R, r = e - k * R; lw[rd.w_off[l] + r * ldk + in + k] = Wh[e]; }
    for (int e = tid; e < R; e += nthr) lw[rd.b_off[l] + e] = b[e];
    if (transposed && rd.wt)   // vec([Wi | Wh]) is column-major: column k of the flat order is row k of the transposed copy
      for (int e = tid; e < R * (in + h); e += nthr) { const int k = e / R, r = e - k * R; lw[rd.wt_off[l] + k * rd.ldr[l] + r] = Wi[e]; }
    const int ns = rd.cell == LDE_CELL_LSTM ? 2 * h : h;
    for (int e = tid; e < ns; e += nthr) lw[rd.s_off[l] + e] = s0[e];
  }
  __syncthreads();
}

constexpr int rnn_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
__host__ __device__ constexpr int rnn_ldk(int K) { int v = (K + 3) & ~3; return ((v >> 2) & 1) ? v : v + 4; }

// One workgroup = tpw trajectories × Hp lanes. mode 0: forward sweep → y. mode 1: sweep with records, then BPTT.
// CELL_ ≥ 0 instantiates the kernel for one stack shape (cell kind, IN0_ → H_ → … → H_, L_ cells) — the reference's default
// pattern extractors [REF src/models/GOKU.jl:229-238] —: every loop bound and LDS stride is then a compile-time constant, the
// layer loops unroll, and the per-layer kernel-argument reads, the short runtime loops and the vmcnt(0) waits they force
// (the prefetched frame / record behind a variable number of staging stores) disappear. CELL_ = −1: any shape, at run time.
// REGW (specialised shapes launched with ONE wave per workgroup — every small batch): each lane copies its rows of [Wi | Wh]
// (and, for the pullback, of the transposed copy) from LDS into registers once; a dot product is then K/4 broadcast reads of
// [x; h] feeding v_pk_fma_f32 on register pairs — half the LDS instructions and half the FMA issue slots of the LDS-row form
// (measured per (step, layer) on the LSTM stack: forward product 610 → ≈ 330 cycles, Wᵀδ 940 → ≈ 450).
template <int CELL_, int IN0_, int H_, int L_, int MODE_, bool REGW>
__device__ __forceinline__ void rnn_body(const RnnDims& rd, const RnnArgs& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  constexpr bool SP = CELL_ >= 0;
  static_assert(!REGW || SP, "register-resident weights need a compile-time shape");
  constexpr int UL = SP ? L_ : 1, UK = SP ? 16 : 4;   // unroll factors: layer loops, dot-product loops
  const int cellk = SP ? CELL_ : rd.cell;
  const bool lstm = cellk == LDE_CELL_LSTM;
  const int G = lstm ? 4 : 1, L = SP ? L_ : rd.nL;
  const int Hp = SP ? rnn_pow2((CELL_ == LDE_CELL_LSTM ? 4 : 1) * H_) : rd.Hp, hmaxv = SP ? H_ : rd.hmax;
  auto size_of = [&](int l) { return SP ? (l == 0 ? IN0_ : H_) : rd.sizes[l]; };
  auto ldk_of = [&](int l) { return SP ? rnn_ldk(size_of(l) + size_of(l + 1)) : rd.ldk[l]; };
  const int tpw = a.tpw, nthr = tpw * Hp, tid = threadIdx.x, tr = tid / Hp, u = tid - tr * Hp;
  const int T = a.T, B = a.B;
  const int mode = SP ? MODE_ : a.mode;   // compile-time in the instantiated kernels: the staging stores are then unconditional
                                          // and the waits for the prefetched frames count them instead of draining them
  float* lw = rsm;
  float* vbuf = lw + rd.lds_w + tr * rd.vmax;                     // this trajectory's [x; h_prev]
  float* dbuf = lw + rd.lds_w + tpw * rd.vmax + tr * rd.rmax;      // its gate deltas
  float* hst = lw + rd.lds_w + tpw * (rd.vmax + rd.rmax) + tr * (2 * L * hmaxv);   // h[l][hmax], then c[l][hmax]
  float* cst = hst + L * hmaxv;
  float* dhs = lw + rd.lds_w + tpw * (rd.vmax + rd.rmax + 2 * L * hmaxv) + tr * (2 * L * hmaxv);   // dh, dc
  float* dcs = dhs + L * hmaxv;
  // mode 0: forward sweep → y. 1: sweep with records and staged a-panels, then back-propagation through time. 2 (training forward): the
  // sweep of mode 1, then y — the records and panels stay in the handle's workspace. 3: the back-propagation of mode 1 from what a mode-2
  // call left there (the pullback then does not repeat the sweep: 121 → ≈ 75 µs for the LSTM stack at B = 256, T = 50).
  const bool keep = mode == 1 || mode == 2, bptt = mode == 1 || mode == 3;
  rnn_load_weights(rd, a.Wflat, lw, nthr, bptt);
  // REGW: the lane's weight rows, as float4 groups (zero padded like the LDS rows)
  constexpr int GC = SP ? (CELL_ == LDE_CELL_LSTM ? 4 : 1) : 1;
  constexpr int HPC = SP ? rnn_pow2(GC * (H_ > 0 ? H_ : 1)) : 1;                 // lanes per trajectory
  constexpr int KF4 = REGW ? (IN0_ + H_ + 3) / 4 : 1;                             // float4 groups of the widest forward row
  constexpr int RB4 = REGW ? (GC * H_ + 3) / 4 : 1;                               // float4 groups of a transposed row (R = G·h)
  constexpr int NKI = REGW ? (IN0_ + H_ + HPC - 1) / HPC : 1;                      // outputs of Wᵀδ per lane
  f32x4 wrow[REGW ? L_ : 1][KF4];
  constexpr bool BP = MODE_ == 1 || MODE_ == 3;
  f32x4 wcol[(REGW && BP) ? L_ : 1][(REGW && BP) ? NKI : 1][(REGW && BP) ? RB4 : 1];
  if (REGW) {
#pragma unroll
    for (int l = 0; l < (REGW ? L_ : 0); l++) {
      const int K = size_of(l) + size_of(l + 1), ldk = ldk_of(l), Rl = GC * size_of(l + 1);
      const float* wr = lw + rd.w_off[l] + (u < Rl ? u : 0) * ldk;
#pragma unroll
      for (int k4 = 0; k4 < KF4; k4++)
        wrow[l][k4] = (u < Rl && 4 * k4 < K) ? *reinterpret_cast<const f32x4*>(wr + 4 * k4) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (BP) {
        const int ldr = rnn_ldk(Rl);
#pragma unroll
        for (int q = 0; q < ((REGW && BP) ? NKI : 0); q++) {
          const int k = u + q * HPC;
          const float* wk = lw + rd.wt_off[l] + (k < K ? k : 0) * ldr;
#pragma unroll
          for (int r4 = 0; r4 < RB4; r4++)
            wcol[l][q][r4] = (k < K && 4 * r4 < Rl) ? *reinterpret_cast<const f32x4*>(wk + 4 * r4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  }
  const long long b = (long long)bx * tpw + tr;
  const bool valid = b < B;
  const size_t tile = (size_t)(b >> 4);   // staging tile (16 trajectories = one column slot of the weight-gradient kernel)
  const int row = (int)(b & 15);
  const int in0 = size_of(0);
  const size_t bc = (size_t)(valid ? b : B - 1);   // (a trajectory past B computes on a copy of trajectory B−1 and stores nothing)

  // ---- forward sweep (state0 → … → last frame) ------------------------------------------------------------------------
  if (mode != 3) {
#pragma unroll UL
  for (int l = 0; l < L; l++) {
    const int h = size_of(l + 1);
    if (u < h) {
      hst[l * hmaxv + u] = lw[rd.s_off[l] + u];
      cst[l * hmaxv + u] = lstm ? lw[rd.s_off[l] + h + u] : 0.f;
    }
  }
  // frame s+1 is fetched while frame s is processed (a global round trip per time step would otherwise be exposed T times)
  constexpr int XQ = 4;                       // prefetched inputs per lane (in ≤ XQ·Hp; wider inputs fall back to direct loads)
  float xq[XQ];
  // (branch-free: always-in-bounds addresses and no masking — a trajectory past B computes on a copy of trajectory B−1 and
  // stores nothing. A load under a branch, or a select right behind it, would turn the wait for it into a vmcnt(0) at the
  // point of issue, i.e. no prefetch at all.)
  auto fetch_x = [&](int s) {
    const int t = rd.reverse ? T - 1 - s : s;
#pragma unroll
    for (int q = 0; q < XQ; q++) {
      const int k = u + q * Hp;
      if (SP && k >= in0) { xq[q] = 0.f; continue; }
      xq[q] = a.x[(size_t)in0 * (bc + (size_t)B * t) + (k < in0 ? k : in0 - 1)];
    }
  };
  fetch_x(0);
  for (int s = 0; s < T; s++) {
    const int t = rd.reverse ? T - 1 - s : s;
  #pragma unroll UL
  for (int l = 0; l < L; l++) {
      const int in = size_of(l), h = size_of(l + 1), K = in + h, ldk = ldk_of(l);
      PROF_T(p0);
      // assemble [input ; h_prev] (the tail up to pad4(K) stays zero: vbuf was zero-filled, entries beyond K never written)
      if (l == 0) {
#pragma unroll
        for (int q = 0; q < XQ; q++)
          if (u + q * Hp < in) vbuf[u + q * Hp] = xq[q];
        for (int k = u + XQ * Hp; k < in; k += Hp) vbuf[k] = a.x[(size_t)in0 * (bc + (size_t)B * t) + k];
        fetch_x(s + 1 < T ? s + 1 : s);
      } else {
        for (int k = u; k < in; k += Hp) vbuf[k] = hst[(l - 1) * hmaxv + k];
      }
      for (int k = u; k < h; k += Hp) vbuf[in + k] = hst[l * hmaxv + k];
      for (int k = K + u; k < ((K + 3) & ~3); k += Hp) vbuf[k] = 0.f;
      PROF_T(p1);
      PROF_ADD(0, p0, p1);
      if (keep) {   // the layer's input vector is the a-panel of the weight gradient: staged here, while it is in LDS
        float* ga = a.stage[l] + (tile * T + s) * a.blk[l] + row * pad32(K);
        for (int k = u; k < pad32(K); k += Hp) ga[k] = (valid && k < K) ? vbuf[k] : 0.f;
      }
      // one lane per gate ROW (Hp ≥ G·h lanes per trajectory): z_r = b_r + [Wi|Wh]_r · [x; h]; the unit lanes then pick
      // their G pre-activations up from LDS. (One lane per UNIT doing G rows was 4× the dependent work per lane.)
      PROF_T(p2);
      PROF_ADD(1, p1, p2);
      const int Rl = G * h;
      if (REGW) {
        if (u < Rl) {
          f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
          for (int k4 = 0; k4 < KF4; k4++) {
            if (4 * k4 < K) {   // (compile-time after unrolling: layer 1's rows are shorter)
              const f32x4 xv = *reinterpret_cast<const f32x4*>(vbuf + 4 * k4);
              c01 += wrow[REGW ? l : 0][k4].lo * xv.lo;
              c23 += wrow[REGW ? l : 0][k4].hi * xv.hi;
            }
          }
          dbuf[u] = lw[rd.b_off[l] + u] + ((c01.x + c01.y) + (c23.x + c23.y));
        }
      } else if (u < Rl) {
        const int K4 = (K + 3) >> 2;
        const float* wr = lw + rd.w_off[l] + u * ldk;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll UK
        for (int k4 = 0; k4 < K4; k4++)
          acc += *reinterpret_cast<const f32x4*>(wr + 4 * k4) * *reinterpret_cast<const f32x4*>(vbuf + 4 * k4);
        dbuf[u] = lw[rd.b_off[l] + u] + ((acc[0] + acc[1]) + (acc[2] + acc[3]));
      }
      PROF_T(p3);
      PROF_ADD(2, p2, p3);
      if (u < h) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; g++)
          if (g < G) z[g] = dbuf[g * h + u];
        float hn, cn = 0.f;
        if (lstm) {
          const float ig = sigm(z[0]), fg = sigm(z[1]), gg = fast_tanh(z[2]), og = sigm(z[3]);
          z[0] = ig; z[1] = fg; z[2] = gg; z[3] = og;
          cn = __builtin_fmaf(fg, cst[l * hmaxv + u], ig * gg);   // (spelled out: left to the compiler, which product is fused depends on the surrounding code — rnn_body2 must give the same bits)
          hn = og * fast_tanh(cn);
        } else {
          hn = cellk == LDE_CELL_RNN_TANH ? fast_tanh(z[0]) : fmaxf(z[0], 0.f);
          z[0] = hn;
        }
        if (keep && valid) {
          float* r = a.rec + (((size_t)s * L + l) * B + (size_t)b) * rd.recw;
#pragma unroll
          for (int g = 0; g < 4; g++)
            if (g < G) r[g * h + u] = z[g];
          r[G * h + u] = cn;
          r[G * h + h + u] = hn;
        }
        hst[l * hmaxv + u] = hn;   // every lane of the trajectory has copied h_prev into vbuf already (same wave, in order)
        cst[l * hmaxv + u] = cn;
      }
      PROF_T(p4);
      PROF_ADD(3, p3, p4);
    }
  }
  if (mode == 0 || mode == 2) {
    const int hL = size_of(L);
    if (valid && u < hL) a.y[(size_t)a.ldy * b + u] = hst[(L - 1) * hmaxv + u];
    return;
  }
  }   // (mode 3 starts here)
  __syncthreads();   // the records are read back below by other lanes of the trajectory: stores drained (vmcnt(0)) first

  // ---- back-propagation through time -----------------------------------------------------------------------------------
#pragma unroll UL
  for (int l = 0; l < L; l++) {
    const int h = size_of(l + 1);
    if (u < h) {
      dhs[l * hmaxv + u] = (l == L - 1 && valid) ? rnn_dy(a, b, u) : 0.f;
      dcs[l * hmaxv + u] = 0.f;
    }
  }
  // records of the next (step, layer) are fetched while the current one is processed
  float rq[6];   // gates (≤ 4), c_new, c_prev
  auto fetch_rec = [&](int s, int l) {   // branch-free like fetch_x; c_prev of step 0 (the trainable c0) is patched in at the use
    const int h = size_of(l + 1), R = G * h;
    const int uc = u < h ? u : h - 1;
    const float* r = a.rec + (((size_t)s * L + l) * B + bc) * rd.recw;
#pragma unroll
    for (int g = 0; g < 4; g++) rq[g] = g < G ? r[g * h + uc] : 0.f;
    rq[4] = 0.f;
    rq[5] = 0.f;
    if (lstm) {
      rq[4] = r[R + uc];
      rq[5] = a.rec[(((size_t)(s > 0 ? s - 1 : 0) * L + l) * B + bc) * rd.recw + R + uc];
    }
  };
  fetch_rec(T - 1, L - 1);
  for (int s = T - 1; s >= 0; s--) {
    const int t = rd.reverse ? T - 1 - s : s;
    if (u == 0) a.wts[(tile * T + s) * NB + row] = valid ? 1.f : 0.f;
#pragma unroll UL
    for (int l = L - 1; l >= 0; l--) {
      const int in = size_of(l), h = size_of(l + 1), K = in + h, R = G * h, ldk = ldk_of(l);
      const int K32 = pad32(K), R32 = pad32(R);
      PROF_T(q0);
      float cur[6];
#pragma unroll
      for (int q = 0; q < 6; q++) cur[q] = rq[q];
      if (l > 0) fetch_rec(s, l - 1);
      else fetch_rec(s > 0 ? s - 1 : 0, L - 1);
      PROF_T(q1);
      PROF_ADD(10, q0, q1);
      // gate deltas of this lane's unit
      if (u < h) {
        const float dh = dhs[l * hmaxv + u];
        if (lstm) {
          const float ig = cur[0], fg = cur[1], gg = cur[2], og = cur[3], cn = cur[4];
          const float cp = s > 0 ? cur[5] : lw[rd.s_off[l] + h + u];
          const float tc = fast_tanh(cn);
          const float dct = __builtin_fmaf(dh * og, __builtin_fmaf(-tc, tc, 1.f), dcs[l * hmaxv + u]);
          dbuf[u] = dct * gg * ig * (1.f - ig);
          dbuf[h + u] = dct * cp * fg * (1.f - fg);
          dbuf[2 * h + u] = dct * ig * __builtin_fmaf(-gg, gg, 1.f);
          dbuf[3 * h + u] = dh * tc * og * (1.f - og);
          dcs[l * hmaxv + u] = dct * fg;
        } else {
          const float av = cur[0];
          dbuf[u] = dh * (cellk == LDE_CELL_RNN_TANH ? __builtin_fmaf(-av, av, 1.f) : (av > 0.f ? 1.f : 0.f));
        }
      }
      PROF_T(q2);
      PROF_ADD(11, q1, q2);
      {   // the δ-panel next to the a-panel the forward sweep staged
        float* gd = a.stage[l] + (tile * T + s) * a.blk[l] + NB * K32 + row * R32;
        for (int k = u; k < R32; k += Hp) gd[k] = (valid && k < R) ? dbuf[k] : 0.f;
      }
      PROF_T(q3);
      PROF_ADD(12, q2, q3);
      // [d_in ; dh_prev] = [Wi | Wh]ᵀ δ: lane u owns the outputs k = u, u+Hp, …; four rows of δ per step, independent sums
      const bool wt = SP || rd.wt;
      if (REGW) {
#pragma unroll
        for (int q = 0; q < ((REGW && BP) ? NKI : 0); q++) {
          const int k = u + q * HPC;
          if (k < K) {
            f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
            for (int r4 = 0; r4 < RB4; r4++) {
              const f32x4 dq = *reinterpret_cast<const f32x4*>(dbuf + 4 * r4);
              c01 += wcol[(REGW && BP) ? l : 0][q][r4].lo * dq.lo;
              c23 += wcol[(REGW && BP) ? l : 0][q][r4].hi * dq.hi;
            }
            const float acc = (c01.x + c01.y) + (c23.x + c23.y);
            if (k < in) {
              if (l > 0) dhs[(l - 1) * hmaxv + k] += acc;
              else if (a.dx && valid) a.dx[(size_t)in0 * ((size_t)b + (size_t)B * t) + k] = acc;
            } else
              dhs[l * hmaxv + (k - in)] = acc;
          }
        }
      } else
      for (int k = u; k < K; k += Hp) {
        const float* wc = lw + rd.w_off[l] + k;
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        const int R4 = R >> 2;
        if (wt) {   // row k of the transposed copy · δ, 16 bytes at a time (same shape of loop as the forward dot product)
          const float* wk = lw + rd.wt_off[l] + k * (SP ? rnn_ldk(R) : rd.ldr[l]);
#pragma unroll UK
          for (int r4 = 0; r4 < R4; r4++)
            acc4 += *reinterpret_cast<const f32x4*>(wk + 4 * r4) * *reinterpret_cast<const f32x4*>(dbuf + 4 * r4);
        } else {
#pragma unroll 4
          for (int r4 = 0; r4 < R4; r4++) {
            const f32x4 dq = *reinterpret_cast<const f32x4*>(dbuf + 4 * r4);
            const float* w = wc + (4 * r4) * ldk;
            acc4[0] += w[0] * dq[0];
            acc4[1] += w[ldk] * dq[1];
            acc4[2] += w[2 * ldk] * dq[2];
            acc4[3] += w[3 * ldk] * dq[3];
          }
        }
        float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        for (int rr = 4 * R4; rr < R; rr++) acc += wc[rr * ldk] * dbuf[rr];
        if (k < in) {
          if (l > 0) dhs[(l - 1) * hmaxv + k] += acc;
          else if (a.dx && valid) a.dx[(size_t)in0 * ((size_t)b + (size_t)B * t) + k] = acc;
        } else
          dhs[l * hmaxv + (k - in)] = acc;
      }
      PROF_T(q4);
      PROF_ADD(13, q3, q4);
    }
  }
  // what is left flows into the trainable initial states
  if (valid) {
    int off = 0;
  #pragma unroll UL
  for (int l = 0; l < L; l++) {
      const int h = size_of(l + 1);
      if (u < h) {
        a.g0[(size_t)b * a.g0w + off + u] = dhs[l * hmaxv + u];
        if (lstm) a.g0[(size_t)b * a.g0w + off + h + u] = dcs[l * hmaxv + u];
      }
      off += lstm ? 2 * h : h;
    }
  }
}

// ---- the default stacks (32 → 16 → 16) as a TWO-WAVE PIPELINE: one wave per cell (round 3) -------------------------------------------------
// In rnn_body ONE wave walks time and, inside a step, the two cells: 100 sequential cell evaluations per sweep, and a lone wave issues
// an instruction every ≈ 8 cycles whatever its dependences — half of what its SIMD could take. Here a workgroup is two waves (the CU
// places them on different SIMDs): wave 0 owns cell 1, wave 1 owns cell 2. Forward: wave 0 publishes h¹ₛ into an LDS ring of PIPE_R
// steps and then its step count (a wave's LDS operations execute in order: a reader that sees the count sees the vector; no barrier, no
// fence); wave 1 runs cell 2 on step s while wave 0 is on step s+1…. Backward the other way round: wave 1 publishes ∂L/∂h¹ₛ, wave 0
// adds it to what its own step s+1 left. A producer PIPE_R steps ahead waits for the consumer's count. Records, staged panels, state
// gradients: each wave its cell's, same addresses and same arithmetic as rnn_body ⇒ the same bits (tests/test_gpu_rnn.py).
constexpr int PIPE_R = 8;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap / v_permlane16_swap with both operands the same register: .x = the even rows (of 32 / 16 lanes) repeated, .y = the odd
// rows repeated — a lane of row 0 reads its partner in row 1 in one VALU instruction instead of an LDS round trip (gfx950)
__device__ __forceinline__ void rows32(float v, float& even, float& odd) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  even = __uint_as_float(r.x);
  odd = __uint_as_float(r.y);
}
__device__ __forceinline__ void rows16(float v, float& even, float& odd) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  even = __uint_as_float(r.x);
  odd = __uint_as_float(r.y);
}

template <int CELL_, int MODE_, int LY>
__device__ __forceinline__ void rnn_pipe_wave(const RnnDims& rd, const RnnArgs& a, const unsigned bx, float* lw, float* xring, float* gring,
                                              int* cnt) {
  constexpr int IN0 = 32, H = 16, L = 2;
  constexpr bool lstm = CELL_ == LDE_CELL_LSTM;
  constexpr int G = lstm ? 4 : 1, Hp = rnn_pow2(G * H), TPW = 64 / Hp;
  constexpr int in = LY == 0 ? IN0 : H, h = H, K = in + h, ldk = rnn_ldk(K), Rl = G * h, ldr = rnn_ldk(Rl);
  constexpr int KF4 = (K + 3) / 4, RB4 = (Rl + 3) / 4, NKI = (K + Hp - 1) / Hp;
  constexpr bool keep = MODE_ == 1 || MODE_ == 2, bptt = MODE_ == 1 || MODE_ == 3;
  static_assert(Rl == Hp && in % 4 == 0 && K % 4 == 0, "one lane per gate row; whole float4 groups");
  const int lane = threadIdx.x & 63, tr = lane / Hp, u = lane - tr * Hp;
  const int T = a.T, B = a.B;
  // What is on the recurrence of a cell goes through LDS ONCE per step (the vector every lane's dot product reads); the rest stays in
  // registers: the lane's weight rows and bias, the unit lanes' c (forward) and ∂L/∂h, ∂L/∂c (backward); the LSTM's four pre-activations
  // reach their unit lane by lane-row swaps. (rnn_body: [x; h] assembled from two LDS buffers, pre-activations and states through LDS —
  // three to four dependent LDS round trips of ≈ 100+ cycles per step.)
  const int PER = rd.vmax + rd.rmax + 4 * H;              // per (cell, trajectory): the [x; h] vector and the gate-delta vector
  float* base = lw + rd.lds_w + (LY * TPW + tr) * PER;
  float* vbuf = base;
  float* dbuf = vbuf + rd.vmax;
  f32x4 wrow[KF4];
  f32x4 wcol[bptt ? NKI : 1][bptt ? RB4 : 1];
  {
    const float* wr = lw + rd.w_off[LY] + u * ldk;
#pragma unroll
    for (int k4 = 0; k4 < KF4; k4++) wrow[k4] = *reinterpret_cast<const f32x4*>(wr + 4 * k4);
    if (bptt) {
#pragma unroll
      for (int q = 0; q < (bptt ? NKI : 0); q++) {
        const int k = u + q * Hp;
        const float* wk = lw + rd.wt_off[LY] + (k < K ? k : 0) * ldr;
#pragma unroll
        for (int r4 = 0; r4 < RB4; r4++)
          wcol[bptt ? q : 0][bptt ? r4 : 0] = k < K ? *reinterpret_cast<const f32x4*>(wk + 4 * r4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  const float bias = lw[rd.b_off[LY] + u];
  const float h0 = u < h ? lw[rd.s_off[LY] + u] : 0.f;
  const float c0 = (lstm && u < h) ? lw[rd.s_off[LY] + h + u] : 0.f;
  const long long b = (long long)bx * TPW + tr;
  const bool valid = b < B;
  const size_t tile = (size_t)(b >> 4);
  const int row = (int)(b & 15);
  const size_t bc = (size_t)(valid ? b : B - 1);
  auto ld_cnt = [&](int i) { return __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto st_cnt = [&](int i, int v) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(&cnt[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  if (MODE_ != 3) {
    // ---- forward sweep ----
    float creg = c0, hn = h0;
    if (u < h) vbuf[in + u] = h0;
    constexpr int XQ = (IN0 + Hp - 1) / Hp;      // inputs per lane of the first cell
    float xq[XQ];
    const ptrdiff_t xstep = (ptrdiff_t)IN0 * B * (rd.reverse ? -1 : 1);
    const float* xp = a.x + (size_t)IN0 * (bc + (size_t)B * (rd.reverse ? T - 1 : 0));   // the frame last asked for
    auto fetch_x = [&]() {   // (branch-free, always in bounds: see rnn_body)
#pragma unroll
      for (int q = 0; q < XQ; q++) {
        const int k = u + q * Hp;
        xq[q] = k < IN0 ? xp[k] : 0.f;
      }
    };
    auto put_x = [&]() {
#pragma unroll
      for (int q = 0; q < XQ; q++)
        if (u + q * Hp < in) vbuf[u + q * Hp] = xq[q];
    };
    if (LY == 0) {   // frame 0 into the vector, frame 1 into registers; in the loop frame s+1 is written once the reads of step s are issued
      fetch_x();
      put_x();
      if (1 < T) xp += xstep;
      fetch_x();
    }
    float* ga = keep ? a.stage[LY] + (tile * T) * a.blk[LY] + row * pad32(K) : nullptr;
    float* rp = keep ? a.rec + ((size_t)LY * B + bc) * rd.recw : nullptr;
    const size_t rstep = (size_t)L * B * rd.recw;
    int slot = 0;                                  // s mod PIPE_R
    int avail = 0, taken = 0;                      // the other wave's counts as last seen
    for (int s = 0; s < T; s++) {
      const float* xs = LY == 0 ? vbuf : xring + (slot * TPW + tr) * H;   // the cell's input: cell 2 reads it where cell 1 published it
      if (LY == 1) {
        while (avail <= s) avail = ld_cnt(0);
        asm volatile("" ::: "memory");
      }
      if (keep) {   // the a-panel row of the weight gradient
        for (int k = u; k < pad32(K); k += Hp) {
          const float* src = (LY == 1 && k < in) ? xs + k : vbuf + (k < K ? k : 0);
          const float v = *src;
          ga[k] = (valid && k < K) ? v : 0.f;
        }
        ga += a.blk[LY];
      }
      float z;
      {
        f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
        for (int k4 = 0; k4 < KF4; k4++) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>((4 * k4 < in ? xs : vbuf) + 4 * k4);
          c01 += wrow[k4].lo * xv.lo;
          c23 += wrow[k4].hi * xv.hi;
        }
        z = bias + ((c01.x + c01.y) + (c23.x + c23.y));
      }
      if (LY == 1) st_cnt(1, s + 1);   // taken: this wave's LDS reads above execute before this write
      if (LY == 0) {
        put_x();                       // (the reads of step s are issued: in order)
        if (s + 2 < T) xp += xstep;
        fetch_x();
      }
      float zz[4] = {z, 0.f, 0.f, 0.f};
      if (lstm) {   // unit lane u < 16 needs rows u, 16+u, 32+u, 48+u
        float lo, hi;
        rows32(z, lo, hi);
        rows16(lo, zz[0], zz[1]);
        rows16(hi, zz[2], zz[3]);
      }
      if (u < h) {
        float cn = 0.f;
        if (lstm) {
          const float ig = sigm(zz[0]), fg = sigm(zz[1]), gg = fast_tanh(zz[2]), og = sigm(zz[3]);
          zz[0] = ig; zz[1] = fg; zz[2] = gg; zz[3] = og;
          cn = __builtin_fmaf(fg, creg, ig * gg);
          hn = og * fast_tanh(cn);
        } else {
          hn = CELL_ == LDE_CELL_RNN_TANH ? fast_tanh(zz[0]) : fmaxf(zz[0], 0.f);
          zz[0] = hn;
        }
        creg = cn;
        vbuf[in + u] = hn;   // the next step's h_prev
        if (keep && valid) {
          float* r = rp;
#pragma unroll
          for (int g = 0; g < 4; g++)
            if (g < G) r[g * h + u] = zz[g];
          r[G * h + u] = cn;
          r[G * h + h + u] = hn;
        }
        if (keep) rp += rstep;
      }
      if (LY == 0) {   // hand h of this step to the cell above: its slot is free once the step PIPE_R back has been taken
        while (s >= PIPE_R && taken <= s - PIPE_R) taken = ld_cnt(1);
        if (u < h) xring[(slot * TPW + tr) * H + u] = hn;
        st_cnt(0, s + 1);
      }
      slot = (slot + 1) & (PIPE_R - 1);
    }
    if (MODE_ == 0 || MODE_ == 2) {
      if (LY == L - 1 && valid && u < h) a.y[(size_t)a.ldy * b + u] = hn;
      return;
    }
    __syncthreads();   // the records are read back below: stores drained first (both waves of the workgroup arrive: mode 1 only)
  }

  // ---- back-propagation through time ----
  float dhr = (u < h && LY == L - 1 && valid) ? rnn_dy(a, b, u) : 0.f, dcr = 0.f;
  float rq[6];
  const size_t rstepb = (size_t)L * B * rd.recw;
  const int ucb = u < h ? u : h - 1;
  const float* rpb = a.rec + (((size_t)(T - 1) * L + LY) * B + bc) * rd.recw + ucb;   // this lane's record entries of the step being fetched
  auto fetch_rec = [&](bool first) {   // (first: the step has no predecessor; c_prev is then the trainable c0, patched in at the use)
#pragma unroll
    for (int g = 0; g < 4; g++) rq[g] = g < G ? rpb[g * h] : 0.f;
    rq[4] = 0.f;
    rq[5] = 0.f;
    if (lstm) {
      rq[4] = rpb[Rl];
      rq[5] = (first ? rpb : rpb - rstepb)[Rl];
    }
  };
  fetch_rec(T == 1);
  float* gd = a.stage[LY] + (tile * T + (T - 1)) * a.blk[LY] + NB * pad32(K) + row * pad32(Rl);
  float* wp = a.wts + (tile * T + (T - 1)) * NB + row;
  const ptrdiff_t dxstep = (ptrdiff_t)IN0 * B * (rd.reverse ? 1 : -1);
  float* dxp = a.dx ? a.dx + (size_t)IN0 * (bc + (size_t)B * (rd.reverse ? 0 : T - 1)) : nullptr;
  int slot = 0;
  int avail = 0, taken = 0;
  float gin = 0.f;          // cell 1: what the cell above sends back for the coming step, picked up early when it is already there
  bool have = false;
  for (int s = T - 1; s >= 0; s--) {
    const int idx = T - 1 - s;
    if (LY == 0 && u == 0) *wp = valid ? 1.f : 0.f;
    wp -= NB;
    float cur[6];
#pragma unroll
    for (int q = 0; q < 6; q++) cur[q] = rq[q];
    if (s > 0) rpb -= rstepb;
    fetch_rec(s <= 1);
    if (LY == 0) {   // ∂L/∂h of this step: what this cell's own step s+1 left plus what the cell above sends
      if (!have) {
        while (avail <= idx) avail = ld_cnt(2);
        asm volatile("" ::: "memory");
        gin = u < h ? gring[(slot * TPW + tr) * H + u] : 0.f;
        st_cnt(3, idx + 1);
      }
      dhr += gin;
    }
    if (u < h) {
      const float dhv = dhr;
      if (lstm) {
        const float ig = cur[0], fg = cur[1], gg = cur[2], og = cur[3], cn = cur[4];
        const float cp = s > 0 ? cur[5] : c0;
        const float tc = fast_tanh(cn);
        const float dct = __builtin_fmaf(dhv * og, __builtin_fmaf(-tc, tc, 1.f), dcr);
        dbuf[u] = dct * gg * ig * (1.f - ig);
        dbuf[h + u] = dct * cp * fg * (1.f - fg);
        dbuf[2 * h + u] = dct * ig * __builtin_fmaf(-gg, gg, 1.f);
        dbuf[3 * h + u] = dhv * tc * og * (1.f - og);
        dcr = dct * fg;
      } else {
        const float av = cur[0];
        dbuf[u] = dhv * (CELL_ == LDE_CELL_RNN_TANH ? __builtin_fmaf(-av, av, 1.f) : (av > 0.f ? 1.f : 0.f));
      }
    }
    {
      for (int k = u; k < pad32(Rl); k += Hp) gd[k] = (valid && k < Rl) ? dbuf[k] : 0.f;
      gd -= a.blk[LY];
    }
    if (LY == 0) {   // the next step's share from above, if it is there already: its LDS latency then hides behind the products below
      have = false;
      if (s > 0 && avail > idx + 1) {
        const int nslot = (slot + 1) & (PIPE_R - 1);
        gin = u < h ? gring[(nslot * TPW + tr) * H + u] : 0.f;
        st_cnt(3, idx + 2);
        have = true;
      }
    }
    if (LY == 1) {   // the ring slot of this step's ∂/∂h (to the cell below) is free once the step PIPE_R back has been taken
      while (idx >= PIPE_R && taken <= idx - PIPE_R) taken = ld_cnt(3);
    }
    float accl = 0.f;   // the last of the lane's outputs
#pragma unroll
    for (int q = 0; q < (bptt ? NKI : 0); q++) {
      const int k = u + q * Hp;
      float acc = 0.f;
      if (k < K) {
        f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
        for (int r4 = 0; r4 < RB4; r4++) {
          const f32x4 dq = *reinterpret_cast<const f32x4*>(dbuf + 4 * r4);
          c01 += wcol[bptt ? q : 0][bptt ? r4 : 0].lo * dq.lo;
          c23 += wcol[bptt ? q : 0][bptt ? r4 : 0].hi * dq.hi;
        }
        acc = (c01.x + c01.y) + (c23.x + c23.y);
        if (k < in) {
          if (LY > 0) gring[(slot * TPW + tr) * H + k] = acc;
          else if (dxp && valid) dxp[k] = acc;
        }
      }
      accl = acc;
    }
    // ∂L/∂h_prev (outputs in … K−1) to the unit lanes: the LSTM's sit one lane row (cell 2) or two (cell 1) up; the plain cells' are the
    // lane's own last output
    if (lstm) {
      float ev, od;
      if (LY == 0) rows32(accl, ev, od);
      else rows16(accl, ev, od);
      dhr = od;
    } else
      dhr = accl;
    if (LY == 1) st_cnt(2, idx + 1);
    if (LY == 0 && dxp) dxp += dxstep;
    slot = (slot + 1) & (PIPE_R - 1);
  }
  if (valid && u < h) {
    const int off = LY == 0 ? 0 : (lstm ? 2 * h : h);
    a.g0[(size_t)b * a.g0w + off + u] = dhr;
    if (lstm) a.g0[(size_t)b * a.g0w + off + h + u] = dcr;
  }
}

// LDS floats of the pipeline form beyond the weight area: per (cell, trajectory) buffers, the two rings, the four counters
__host__ __device__ inline int rnn_pipe_extra_floats(const RnnDims& rd, int tpw) {
  return 2 * tpw * (rd.vmax + rd.rmax + 4 * 16) + 2 * PIPE_R * tpw * 16 + 16;
}
template <int CELL_, int MODE_>
__device__ __forceinline__ void rnn_body2(const RnnDims& rd, const RnnArgs& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  constexpr int Hp = rnn_pow2((CELL_ == LDE_CELL_LSTM ? 4 : 1) * 16), TPW = 64 / Hp;
  float* lw = rsm;
  float* xring = lw + rd.lds_w + 2 * TPW * (rd.vmax + rd.rmax + 4 * 16);
  float* gring = xring + PIPE_R * TPW * 16;
  int* cnt = reinterpret_cast<int*>(gring + PIPE_R * TPW * 16);
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  rnn_load_weights(rd, a.Wflat, lw, 128, MODE_ == 1 || MODE_ == 3);   // (ends with a barrier: the counters are zero for both waves)
  for (int i = threadIdx.x; i < 2 * TPW * (rd.vmax + rd.rmax + 4 * 16); i += 128) lw[rd.lds_w + i] = 0.f;
  __syncthreads();
  if (threadIdx.x < 64) rnn_pipe_wave<CELL_, MODE_, 0>(rd, a, bx, lw, xring, gring, cnt);
  else rnn_pipe_wave<CELL_, MODE_, 1>(rd, a, bx, lw, xring, gring, cnt);
}
template <int CELL_, int MODE_>
__global__ void __launch_bounds__(128) k_rnn2(RnnDims rd, RnnArgs a) {
  rnn_body2<CELL_, MODE_>(rd, a, blockIdx.x);
}

template <int CELL_, int IN0_, int H_, int L_, int MODE_, bool REGW = false>
__global__ void __launch_bounds__(REGW ? 64 : 1024) k_rnn(RnnDims rd, RnnArgs a) {
  rnn_body<CELL_, IN0_, H_, L_, MODE_, REGW>(rd, a, blockIdx.x);
}
// Several stacks of the default shape (32 → 16 → 16, one wave per workgroup) in ONE launch — the three pattern extractors of the GOKU
// encoder [REF src/models/GOKU.jl:32-51] run on the same frames; on three streams inside a captured step their kernels started 15 and
// 60 µs apart and their weight-gradient tails queued behind each other (kernel trace, profiles/r3_goku_step_mixed_*). The cell kind is
// a block-uniform switch; same code on the same data per stack.
constexpr int RNN_GROUP_MAX = 3;
template <int MODE_, bool PIPE = false>
__global__ void __launch_bounds__(PIPE ? 128 : 64) k_rnn_group(GroupTable<RnnDims, RnnArgs, RNN_GROUP_MAX> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  const unsigned bx = blockIdx.x - g.start[j];
  const int cell = g.dims[j].cell;
  if (PIPE) {
    if (cell == LDE_CELL_LSTM) rnn_body2<LDE_CELL_LSTM, MODE_>(g.dims[j], g.args[j], bx);
    else if (cell == LDE_CELL_RNN_RELU) rnn_body2<LDE_CELL_RNN_RELU, MODE_>(g.dims[j], g.args[j], bx);
    else rnn_body2<LDE_CELL_RNN_TANH, MODE_>(g.dims[j], g.args[j], bx);
  } else {
    if (cell == LDE_CELL_LSTM) rnn_body<LDE_CELL_LSTM, 32, 16, 2, MODE_, true>(g.dims[j], g.args[j], bx);
    else if (cell == LDE_CELL_RNN_RELU) rnn_body<LDE_CELL_RNN_RELU, 32, 16, 2, MODE_, true>(g.dims[j], g.args[j], bx);
    else rnn_body<LDE_CELL_RNN_TANH, 32, 16, 2, MODE_, true>(g.dims[j], g.args[j], bx);
  }
}

// dW[state0 slots] += Σ_b g0[b][·]: one wave per state entry, lane j adds trajectories j, j+64, … in order, then a fixed
// butterfly over the lanes (deterministic)
__device__ __forceinline__ void rnn_state0_body(const float* __restrict__ g0, int B, int g0w, const RnnDims& rd, float* __restrict__ dW,
                                                int assign, const int i) {
  int off = 0, l = 0;
  for (; l < rd.nL; l++) {
    const int ns = (rd.cell == LDE_CELL_LSTM ? 2 : 1) * rd.sizes[l + 1];
    if (i < off + ns) break;
    off += ns;
  }
  const int in = rd.sizes[l], h = rd.sizes[l + 1], R = rd.G * h;
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += g0[(size_t)b * g0w + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) {
    float* o = dW + rd.f_off[l] + (size_t)R * in + (size_t)R * h + R + (i - off);
    *o = assign ? s : *o + s;
  }
}
__global__ void __launch_bounds__(64) k_rnn_state0(const float* __restrict__ g0, int B, int g0w, RnnDims rd, float* __restrict__ dW,
                                                    int assign) {
  rnn_state0_body(g0, B, g0w, rd, dW, assign, (int)blockIdx.x);
}
struct State0Args { const float* g0; int B, g0w; float* dW; int assign; };
__global__ void __launch_bounds__(64) k_rnn_state0_group(GroupTable<RnnDims, State0Args, RNN_GROUP_MAX> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  const State0Args& a = g.args[j];
  rnn_state0_body(a.g0, a.B, a.g0w, g.dims[j], a.dW, a.assign, (int)blockIdx.x - g.start[j]);
}
// the slab sums of all (stack, cell) weight gradients AND the initial-state sums in one launch (they depend on different kernels of the
// pullback, both already behind them in the stream): workgroups [0, u.start[u.n]) reduce tiles, the rest — their first wave — sum a state entry
struct ReduceState0Tables {
  GroupTable<MlpDims, ReduceArgs, GROUP_MAX_DW> u;
  GroupTable<RnnDims, State0Args, RNN_GROUP_MAX> s;
};
static_assert(sizeof(ReduceState0Tables) <= 4096, "the two tables must fit the kernel-argument segment");
__global__ void __launch_bounds__(256) k_rnn_reduce_state0_group(ReduceState0Tables t) {
  const int nred = t.u.start[t.u.n];
  if ((int)blockIdx.x < nred) {
    const int j = group_find(t.u.start, t.u.n, blockIdx.x);
    const ReduceArgs& a = t.u.args[j];
    reduce_tiles_body(a.priv, a.nflush, a.nwg, a.slab, a.nslab, t.u.dims[j], a.dW, a.feedback, a.assign, (int)blockIdx.x - t.u.start[j]);
  } else if (threadIdx.x < 64) {
    const int b = (int)blockIdx.x - nred;
    const int j = group_find(t.s.start, t.s.n, b);
    const State0Args& a = t.s.args[j];
    rnn_state0_body(a.g0, a.B, a.g0w, t.s.dims[j], a.dW, a.assign, b - t.s.start[j]);
  }
}

}  // namespace lde

// ================================================ C ABI ======================================================
using namespace lde;

struct lde_rnn {
  lde_rnn_desc d;
  RnnDims rd;
  MlpDims dmw[RNN_ML];     // one-layer descriptions [K_l → G·h_l] for the weight-gradient kernel
  int64_t nW = 0;
  float* W_dev = nullptr;
  bool have_W = false;
  size_t lds = 0;
  int g0w = 0;
  DwSync dws;              // weight-gradient kernels on the dw stream (lde_set_dw_stream)
  int staged_T = 0, staged_B = 0;   // set by lde_rnn_backward_dx: the panels lde_rnn_backward_dw consumes
  int kept_T = 0, kept_B = 0; const float* kept_x = nullptr;   // set by lde_rnn_forward_train: the records and a-panels its sweep left for the pullback
  // workspace
  float* rec = nullptr; size_t rec_cap = 0;
  float* stage[RNN_ML] = {nullptr, nullptr, nullptr, nullptr}; size_t stage_cap[RNN_ML] = {0, 0, 0, 0};
  float* wts = nullptr; size_t wts_cap = 0;
  float* g0 = nullptr; size_t g0_cap = 0;
  float* slab = nullptr; size_t slab_cap = 0; size_t slab_layer = 0;
  int32_t* ints = nullptr; size_t ints_cap = 0;
  bool accumulate = true;   // pullback: dW += gradient (default) or dW = gradient
  int opt_generic = 0, opt_regw = 1, opt_pipe = 1;   // lde_rnn_set_option: "generic", "regw", "pipe" (kernel-choice knobs of the parity tests)
  int opt_async_dw = 1;                              // "async_dw": with a dw stream set, a grouped pullback's weight-gradient products go there (0: stay on the caller's stream)
  void (*kernel[4][3])(lde::RnnDims, lde::RnnArgs) = {};   // the k_rnn instantiations for this stack: [mode][any workgroup size, one wave per workgroup, one wave per cell]
  int io_ldy = 0, io_lddy = 0;       // set around a call by the *_ld group entry points (0: rows are hL apart)
  const float* io_dy2 = nullptr;
  std::string err;
};

static int rnn_desc_ok(const lde_rnn_desc* d) {
  if (!d || d->abi_version != LDE_ABI_VERSION || d->n_layers < 1 || d->n_layers > LDE_RNN_MAX_LAYERS) return 0;
  if (d->cell < 0 || d->cell > LDE_CELL_LSTM) return 0;
  for (int l = 0; l <= d->n_layers; l++)
    if (d->sizes[l] < 1) return 0;
  return 1;
}

// lde_chain.hip (lde_refresh_weights): the handle's own weight buffer, which that call fills
bool rnn_refresh_target(lde_rnn* r, float** W_dev, int64_t* nW) {
  if (!r || !r->W_dev) return false;
  *W_dev = r->W_dev;
  *nW = r->nW;
  r->have_W = true;
  r->kept_T = r->kept_B = 0;
  return true;
}

typedef void (*rnn_kernel_t)(RnnDims, RnnArgs);
template <int CELL_, bool ONE>
static rnn_kernel_t rnn_pick_mode(int mode) {
  switch (mode) {
    case 0: return k_rnn<CELL_, 32, 16, 2, 0, ONE>;
    case 1: return k_rnn<CELL_, 32, 16, 2, 1, ONE>;
    case 2: return k_rnn<CELL_, 32, 16, 2, 2, ONE>;
    default: return k_rnn<CELL_, 32, 16, 2, 3, ONE>;
  }
}

template <int CELL_>
static rnn_kernel_t rnn_pick_pipe(int mode) {
  switch (mode) {
    case 0: return k_rnn2<CELL_, 0>;
    case 1: return k_rnn2<CELL_, 1>;
    case 2: return k_rnn2<CELL_, 2>;
    default: return k_rnn2<CELL_, 3>;
  }
}

extern "C" {

int64_t lde_rnn_num_weights(const lde_rnn_desc* d) {
  if (!rnn_desc_ok(d)) return -1;
  const int64_t G = d->cell == LDE_CELL_LSTM ? 4 : 1, S = d->cell == LDE_CELL_LSTM ? 2 : 1;
  int64_t n = 0;
  for (int l = 0; l < d->n_layers; l++) {
    const int64_t in = d->sizes[l], h = d->sizes[l + 1];
    n += G * h * in + G * h * h + G * h + S * h;
  }
  return n;
}

void lde_rnn_destroy(lde_rnn* r) {
  if (!r) return;
  if (r->W_dev) (void)hipFree(r->W_dev);
  if (r->rec) (void)hipFree(r->rec);
  for (int l = 0; l < RNN_ML; l++)
    if (r->stage[l]) (void)hipFree(r->stage[l]);
  if (r->wts) (void)hipFree(r->wts);
  if (r->g0) (void)hipFree(r->g0);
  if (r->slab) (void)hipFree(r->slab);
  if (r->ints) (void)hipFree(r->ints);
  dw_sync_destroy(r->dws);
  delete r;
}

int lde_rnn_create(const lde_rnn_desc* d, lde_rnn** out) {
  if (!out) return LDE_ERR_INVALID_ARG;
  *out = nullptr;
  if (!rnn_desc_ok(d)) return LDE_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return LDE_ERR_NO_DEVICE;   // no CPU fallback
  lde_rnn* r = new lde_rnn();
  *out = r;
  r->d = *d;
  RnnDims& rd = r->rd;
  std::memset(&rd, 0, sizeof(rd));
  rd.cell = d->cell; rd.nL = d->n_layers; rd.reverse = d->reverse ? 1 : 0; rd.G = d->cell == LDE_CELL_LSTM ? 4 : 1;
  int hmax = 0, kmax = 0, rmax = 0;
  for (int l = 0; l <= d->n_layers; l++) rd.sizes[l] = d->sizes[l];
  for (int l = 1; l <= d->n_layers; l++) hmax = std::max(hmax, d->sizes[l]);
  if (hmax > 64 || d->sizes[0] > 256) {
    r->err = "recurrent stack: hidden width ≤ 64 and input width ≤ 256 supported";
    return LDE_ERR_UNSUPPORTED;
  }
  int Hp = 1;   // lanes per trajectory: one per gate row of the widest cell (≤ 64 ⇒ a trajectory never leaves its wave)
  while (Hp < rd.G * hmax && Hp < 64) Hp <<= 1;
  if (Hp < rd.G * hmax) {
    r->err = "recurrent stack: G·h ≤ 64 gate rows per cell supported (LSTM: h ≤ 16, RNN: h ≤ 64)";
    return LDE_ERR_UNSUPPORTED;
  }
  rd.Hp = Hp;
  rd.hmax = hmax;
  int off = 0, foff = 0;
  r->g0w = 0;
  for (int l = 0; l < rd.nL; l++) {
    const int in = rd.sizes[l], h = rd.sizes[l + 1], R = rd.G * h, K = in + h;
    rd.K[l] = K;
    int v = (K + 3) & ~3;
    if (((v >> 2) & 1) == 0) v += 4;
    rd.ldk[l] = v;
    rd.w_off[l] = off; off += R * v;
    rd.b_off[l] = off; off += (R + 3) & ~3;
    rd.s_off[l] = off; off += (2 * h + 3) & ~3;
    rd.f_off[l] = foff;
    foff += R * in + R * h + R + (d->cell == LDE_CELL_LSTM ? 2 : 1) * h;
    kmax = std::max(kmax, K);
    rmax = std::max(rmax, R);
    r->g0w += (d->cell == LDE_CELL_LSTM ? 2 : 1) * h;
    // one Dense-like layer [K → R] for the weight-gradient kernel: vec([Wi|Wh]) column-major then b = the cell's flat order
    MlpDims& dm = r->dmw[l];
    std::memset(&dm, 0, sizeof(dm));
    dm.nL = 1; dm.sizes[0] = K; dm.sizes[1] = R;
    size_t nf, nft;
    fill_layer_offsets(dm, &nf, &nft);
  }
  {   // transposed copies for the pullback, when they fit beside everything else
    int offT = off;
    for (int l = 0; l < rd.nL; l++) {
      rd.ldr[l] = rnn_ldk(rd.G * rd.sizes[l + 1]);
      rd.wt_off[l] = offT;
      offT += rd.K[l] * rd.ldr[l];
    }
    const size_t per_traj = (size_t)((kmax + 3) & ~3) + 4 + ((rmax + 3) & ~3) + 4 * rd.nL * hmax;
    rd.wt = ((size_t)offT + 16 * per_traj) * sizeof(float) <= LDS_MAX ? 1 : 0;
    if (rd.wt) off = offT;
  }
  rd.lds_w = off;
  rd.vmax = ((kmax + 3) & ~3) + 4;
  rd.rmax = (rmax + 3) & ~3;
  rd.recw = (rd.G + 2) * hmax;
  r->nW = foff;
  r->lds = ((size_t)rd.lds_w + 16 * ((size_t)rd.vmax + rd.rmax + 4 * rd.nL * hmax)) * sizeof(float);
  if (r->lds > LDS_MAX) {
    r->err = "recurrent stack: weights do not fit the 160 KiB LDS";
    return LDE_ERR_UNSUPPORTED;
  }
  if (hipMalloc(&r->W_dev, (size_t)r->nW * sizeof(float)) != hipSuccess) {
    r->err = "recurrent stack: hipMalloc failed";
    return LDE_ERR_ALLOC;
  }
  return LDE_OK;
}

int lde_rnn_set_weights(lde_rnn* r, const float* flat_host, int64_t n) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (n != r->nW || !flat_host) {
    r->err = "lde_rnn_set_weights: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  if (hipMemcpy(r->W_dev, flat_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    r->err = "lde_rnn_set_weights: hipMemcpy failed";
    return LDE_ERR_HIP;
  }
  r->have_W = true;
  r->kept_T = r->kept_B = 0;
  return LDE_OK;
}

int lde_rnn_set_weights_device(lde_rnn* r, const float* flat_dev, int64_t n, void* stream) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (n != r->nW || !flat_dev) {
    r->err = "lde_rnn_set_weights_device: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  if (hipMemcpyAsync(r->W_dev, flat_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    r->err = "lde_rnn_set_weights_device: hipMemcpyAsync failed";
    return LDE_ERR_HIP;
  }
  r->have_W = true;
  r->kept_T = r->kept_B = 0;   // records of a sweep with the previous weights are not this network's
  return LDE_OK;
}

int lde_rnn_reserve(lde_rnn* r, int B, int T) {
  if (!r || !r->W_dev || B < 1 || T < 1) return LDE_ERR_INVALID_ARG;
  const RnnDims& rd = r->rd;
  const size_t ntile = (size_t)cdiv(B, 16);
  size_t slab_need = 0;
  bool ok = grow(&r->rec, &r->rec_cap, (size_t)T * rd.nL * B * rd.recw) && grow(&r->wts, &r->wts_cap, ntile * T * NB) &&
            grow(&r->g0, &r->g0_cap, (size_t)B * r->g0w);
  if (ok && !r->ints) {   // two zero words for the slab reduction (no private slabs; feedback sink)
    ok = hipMalloc(&r->ints, 64) == hipSuccess && hipMemset(r->ints, 0, 64) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
    r->ints_cap = 16;
  }
  for (int l = 0; l < rd.nL && ok; l++) {
    ok = grow(&r->stage[l], &r->stage_cap[l], ntile * T * r->dmw[l].blk_floats);
    slab_need = std::max(slab_need, (ntile * 8 + 1) * (size_t)r->dmw[l].slab_n);
  }
  r->slab_layer = slab_need;   // one region per layer: a grouped tail runs the layers' weight-gradient products in one launch
  ok = ok && grow(&r->slab, &r->slab_cap, slab_need * rd.nL);
  if (!ok) {
    r->err = "recurrent stack: hipMalloc of the workspace failed";
    return LDE_ERR_ALLOC;
  }
  return LDE_OK;
}


// the instantiation for this stack: the reference's default pattern extractors (32 → 16 → 16) have their own, any other shape
// runs the run-time-shaped kernel
static rnn_kernel_t rnn_pick(const RnnDims& rd, int mode, bool one_wave = false, bool generic_only = false) {
  if (!generic_only && rd.wt && rd.nL == 2 && rd.sizes[0] == 32 && rd.sizes[1] == 16 && rd.sizes[2] == 16) {
    if (one_wave) {
      if (rd.cell == LDE_CELL_LSTM) return rnn_pick_mode<LDE_CELL_LSTM, true>(mode);
      if (rd.cell == LDE_CELL_RNN_RELU) return rnn_pick_mode<LDE_CELL_RNN_RELU, true>(mode);
      if (rd.cell == LDE_CELL_RNN_TANH) return rnn_pick_mode<LDE_CELL_RNN_TANH, true>(mode);
    } else {
      if (rd.cell == LDE_CELL_LSTM) return rnn_pick_mode<LDE_CELL_LSTM, false>(mode);
      if (rd.cell == LDE_CELL_RNN_RELU) return rnn_pick_mode<LDE_CELL_RNN_RELU, false>(mode);
      if (rd.cell == LDE_CELL_RNN_TANH) return rnn_pick_mode<LDE_CELL_RNN_TANH, false>(mode);
    }
  }
  return k_rnn<-1, 0, 0, 0, 0>;
}

// ---- grouped calls (lde_rnn_group_*): as in lde_chain.hip — while a recorder is installed the launch sites record; the group entry point
// issues each stage once for all stacks where they ask for the same kernel family, one by one otherwise.
struct RnnRecMain { rnn_kernel_t fn; bool groupable, pipe; int mode; RnnDims rd; RnnArgs a; unsigned grid, block; size_t lds; };
struct RnnRecDw { int ndw; MlpDims dm; DwArgs da; int gx, gy, gz; size_t lds; MlpDims rdm; ReduceArgs ra; unsigned rgrid; };
struct RnnRecS0 { RnnDims rd; State0Args a; unsigned grid; };
struct RnnGroupRec {
  int n = 0;
  bool main_set[RNN_GROUP_MAX] = {};
  RnnRecMain main[RNN_GROUP_MAX];
  int ndw = 0;
  RnnRecDw dw[GROUP_MAX_DW];
  int ns0 = 0;
  RnnRecS0 s0[RNN_GROUP_MAX];
  lde_rnn* hs[RNN_GROUP_MAX] = {};   // the stacks of the call (their weight-gradient events, when the group's products go to the dw stream)
  int nhs = 0;
};
static thread_local RnnGroupRec* t_rrec = nullptr;
static_assert(sizeof(GroupTable<RnnDims, RnnArgs, RNN_GROUP_MAX>) <= 4096, "a group's argument table must fit the kernel-argument segment");

static int rnn_launch(lde_rnn* r, const RnnArgs& a, int B, hipStream_t stream) {
  const int m = a.mode & 3;
  // Trajectories per workgroup. The sweep is sequential in time and every trajectory re-reads the cell's weights from LDS
  // at every step, so a small batch is spread over as many CUs as it has waves (one wave per workgroup: the LDS of a CU then
  // serves one wave instead of sixteen); only a batch that would exceed ~4 workgroups per CU packs more trajectories
  // behind one LDS copy of the weights.
  int tpw = std::max(1, 64 / r->rd.Hp);
  while (tpw < 16 && cdiv(B, tpw) > 1024) tpw *= 2;
  const bool one_wave = tpw * r->rd.Hp == 64 && r->opt_regw != 0;   // one wave per workgroup: the register-resident-weights instantiation (option "regw")
  // … and, for the default shape, one wave per CELL (rnn_body2): option "pipe" = 0 keeps the single wave
  const RnnDims& rd0 = r->rd;
  const bool def_shape = rd0.wt && rd0.nL == 2 && rd0.sizes[0] == 32 && rd0.sizes[1] == 16 && rd0.sizes[2] == 16;
  const bool generic_only = r->opt_generic != 0;
  const bool pipe = one_wave && def_shape && !generic_only && !LDE_PROF && r->opt_pipe != 0;
  const int one = pipe ? 2 : (one_wave ? 1 : 0);
  if (!r->kernel[m][one]) {
    if (pipe)
      r->kernel[m][one] = rd0.cell == LDE_CELL_LSTM ? rnn_pick_pipe<LDE_CELL_LSTM>(m)
                          : rd0.cell == LDE_CELL_RNN_RELU ? rnn_pick_pipe<LDE_CELL_RNN_RELU>(m) : rnn_pick_pipe<LDE_CELL_RNN_TANH>(m);
    else
      r->kernel[m][one] = rnn_pick(r->rd, m, one != 0, generic_only);
    if (hipFuncSetAttribute((const void*)r->kernel[m][one], hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      r->kernel[m][one] = nullptr;
      r->err = "hipFuncSetAttribute(k_rnn) failed";
      return LDE_ERR_HIP;
    }
  }
  RnnArgs aa = a;
  aa.tpw = tpw;
  const size_t lds = pipe ? ((size_t)r->rd.lds_w + rnn_pipe_extra_floats(r->rd, tpw)) * sizeof(float)
                          : ((size_t)r->rd.lds_w + tpw * ((size_t)r->rd.vmax + r->rd.rmax + 4 * r->rd.nL * r->rd.hmax)) * sizeof(float);
  const unsigned block = pipe ? 128u : (unsigned)(tpw * r->rd.Hp);
  // whole staging tiles are covered (rows past B write zero panels and zero column weights)
#if LDE_PROF
  { long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
#endif
  if (t_rrec) {
    RnnRecMain& q = t_rrec->main[t_rrec->n];
    q.fn = r->kernel[m][one];
    const RnnDims& rd = r->rd;
    q.groupable = one && rd.wt && rd.nL == 2 && rd.sizes[0] == 32 && rd.sizes[1] == 16 && rd.sizes[2] == 16 &&
                  q.fn != (rnn_kernel_t)k_rnn<-1, 0, 0, 0, 0>;
    q.pipe = pipe;
    q.mode = m; q.rd = rd; q.a = aa; q.grid = (unsigned)(cdiv(B, 16) * (16 / tpw)); q.block = block; q.lds = lds;
    t_rrec->main_set[t_rrec->n] = true;
    return LDE_OK;
  }
  hipLaunchKernelGGL(r->kernel[m][one], dim3(cdiv(B, 16) * (16 / tpw)), dim3(block), lds, stream, r->rd, aa);
  if (hipGetLastError() != hipSuccess) {
    r->err = "k_rnn launch failed";
    return LDE_ERR_HIP;
  }
#if LDE_PROF
  {
    static int calls = 0;
    (void)hipStreamSynchronize(stream);
    long long v[64];
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
    if (++calls % 20 == 0) {
      fprintf(stderr, "[prof rnn mode %d cell %d] cycles:", m, r->rd.cell);
      for (int i = 0; i < 64; i++)
        if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return LDE_OK;
}

int lde_rnn_forward(lde_rnn* r, const float* x, int T, int B, float* y, void* stream_) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (!x || !y || T < 1 || B < 1) {
    r->err = "lde_rnn_forward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (!r->have_W) {
    r->err = "lde_rnn_forward: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  RnnArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.Wflat = r->W_dev; a.y = y; a.T = T; a.B = B; a.mode = 0;
  a.ldy = r->io_ldy ? r->io_ldy : r->rd.sizes[r->rd.nL];
  return rnn_launch(r, a, B, (hipStream_t)stream_);
}

// Training variant of the forward call: the sweep also leaves its per-step records and the weight gradient's a-panels in the handle's
// workspace, and the next lde_rnn_backward[_dx] on the SAME (x, T, B) runs the back-propagation alone instead of repeating the sweep
// (what lde_chain_forward_save is to the chains; here the handle keeps the buffers — they are its staging area already).
int lde_rnn_forward_train(lde_rnn* r, const float* x, int T, int B, float* y, void* stream_) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (!x || !y || T < 1 || B < 1) {
    r->err = "lde_rnn_forward_train: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (!r->have_W) {
    r->err = "lde_rnn_forward_train: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  r->kept_T = r->kept_B = 0;
  r->kept_x = nullptr;
  r->staged_T = r->staged_B = 0;
  int rc = lde_rnn_reserve(r, B, T);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (!dw_sync_begin(r->dws, stream)) {   // the workspace is about to be rewritten
    r->err = "lde_rnn_forward_train: waiting for the previous weight gradient failed";
    return LDE_ERR_HIP;
  }
  const RnnDims& rd = r->rd;
  RnnArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.Wflat = r->W_dev; a.y = y; a.rec = r->rec; a.wts = r->wts; a.g0 = r->g0; a.g0w = r->g0w;
  a.ldy = r->io_ldy ? r->io_ldy : r->rd.sizes[r->rd.nL];
  a.T = T; a.B = B; a.mode = 2;
  for (int l = 0; l < rd.nL; l++) { a.stage[l] = r->stage[l]; a.blk[l] = r->dmw[l].blk_floats; }
  rc = rnn_launch(r, a, B, stream);
  if (rc) return rc;
  r->kept_T = T;
  r->kept_B = B;
  r->kept_x = x;
  return LDE_OK;
}

// The pullback in two halves (include/lde.h): the sweep that produces dx and stages the panels, and the weight-gradient tail.
int lde_rnn_backward_dx(lde_rnn* r, const float* x, const float* dy, int T, int B, float* dx, void* stream_) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (!x || !dy || T < 1 || B < 1) {
    r->err = "lde_rnn_backward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (!r->have_W) {
    r->err = "lde_rnn_backward: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  r->staged_T = r->staged_B = 0;
  const bool kept = r->kept_T == T && r->kept_B == B && r->kept_x == x;   // lde_rnn_forward_train left the records and a-panels of this very call
  r->kept_T = r->kept_B = 0;
  r->kept_x = nullptr;
  int rc = lde_rnn_reserve(r, B, T);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (!kept && !dw_sync_begin(r->dws, stream)) {   // the workspace is about to be rewritten
    r->err = "lde_rnn_backward: waiting for the previous weight gradient failed";
    return LDE_ERR_HIP;
  }
  const RnnDims& rd = r->rd;
  RnnArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.Wflat = r->W_dev; a.rec = r->rec; a.dy = dy; a.dx = dx; a.wts = r->wts; a.g0 = r->g0; a.g0w = r->g0w;
  a.lddy = r->io_lddy ? r->io_lddy : r->rd.sizes[r->rd.nL];
  a.dy2 = r->io_dy2;
  a.T = T; a.B = B; a.mode = kept ? 3 : 1;
  for (int l = 0; l < rd.nL; l++) { a.stage[l] = r->stage[l]; a.blk[l] = r->dmw[l].blk_floats; }
  rc = rnn_launch(r, a, B, stream);
  if (rc) return rc;
  r->staged_T = T;
  r->staged_B = B;
  return LDE_OK;
}

int lde_rnn_backward_dw(lde_rnn* r, float* dW, void* stream_) {
  if (!r || !r->W_dev) return LDE_ERR_INVALID_ARG;
  if (!dW) {
    r->err = "lde_rnn_backward_dw: NULL pointer";
    return LDE_ERR_INVALID_ARG;
  }
  if (r->staged_T < 1) {
    r->err = "lde_rnn_backward_dw: no staged pullback (call lde_rnn_backward_dx first; its panels are consumed once)";
    return LDE_ERR_INVALID_ARG;
  }
  const int T = r->staged_T, B = r->staged_B;
  r->staged_T = r->staged_B = 0;
  hipStream_t stream = (hipStream_t)stream_;
  const RnnDims& rd = r->rd;
  const int ntile = cdiv(B, 16);
  bool sw_ok = true;
  hipStream_t wst = t_rrec ? stream : dw_sync_switch(r->dws, stream, &sw_ok);   // the weight-gradient kernels: on the dw stream when one is set (a grouped call's stay on the caller's)
  if (!sw_ok) {
    r->err = "lde_rnn_backward: switching to the weight-gradient stream failed";
    return LDE_ERR_HIP;
  }
  for (int l = 0; l < rd.nL; l++) {
    DwArgs da;
    da.stage = r->stage[l]; da.wts = r->wts; da.nslots = nullptr; da.slab = r->slab + (size_t)l * r->slab_layer; da.cap = T; da.total = (long long)ntile * T;   // every tile staged exactly T slots
    int ks = cdiv(512, ntile * dw_jobs(r->dmw[l], dw_pick_ndw(r->dmw[l])));
    ks = ks < 1 ? 1 : (ks > 8 ? 8 : ks);
    if (t_rrec && t_rrec->ndw < GROUP_MAX_DW) {
      RnnRecDw& q = t_rrec->dw[t_rrec->ndw++];
      q.ndw = dw_pick_ndw(r->dmw[l]); q.dm = r->dmw[l]; q.da = da; q.gx = ntile; q.gy = ks; q.gz = dw_jobs(r->dmw[l], q.ndw);
      q.lds = dw_lds_floats(r->dmw[l], q.ndw) * sizeof(float);
      q.rdm = r->dmw[l];
      q.ra = ReduceArgs{nullptr, r->ints, 0, da.slab, ntile * ks, dW + rd.f_off[l], r->ints + 2, r->accumulate ? 0 : 1};
      q.rgrid = (unsigned)cdiv(r->dmw[l].slab_n, 1024);
      continue;
    }
    int rc = launch_weight_gradient(r->dmw[l], da, ntile, ks, nullptr, r->ints, 0, dW + rd.f_off[l], r->ints + 2, wst, r->err, !r->accumulate);
    if (rc) return rc;
  }
  if (t_rrec) {
    RnnRecS0& q = t_rrec->s0[t_rrec->ns0++];
    q.rd = rd; q.a = State0Args{r->g0, B, r->g0w, dW, r->accumulate ? 0 : 1}; q.grid = (unsigned)r->g0w;
    return LDE_OK;
  }
  hipLaunchKernelGGL(k_rnn_state0, dim3(r->g0w), dim3(64), 0, wst, r->g0, B, r->g0w, rd, dW, r->accumulate ? 0 : 1);
  if (hipGetLastError() != hipSuccess || !dw_sync_end(r->dws, wst, stream)) {
    r->err = "recurrent stack: gradient kernels failed to launch";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

int lde_rnn_backward(lde_rnn* r, const float* x, const float* dy, int T, int B, float* dx, float* dW, void* stream_) {
  if (r && r->W_dev && !dW) {
    r->err = "lde_rnn_backward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  const int rc = lde_rnn_backward_dx(r, x, dy, T, B, dx, stream_);
  return rc ? rc : lde_rnn_backward_dw(r, dW, stream_);
}

static int rnn_group_flush(RnnGroupRec& g, hipStream_t stream) {
  const int n = g.n;
  {   // the sweeps
    bool any = false, same = n >= 2;
    for (int j = 0; j < n; j++) { any = any || g.main_set[j]; same = same && g.main_set[j] && g.main[j].groupable && g.main[j].mode == g.main[0].mode && g.main[j].pipe == g.main[0].pipe; }
    if (any && same) {
      GroupTable<RnnDims, RnnArgs, RNN_GROUP_MAX> t{};
      t.n = n;
      size_t lds = 0;
      for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.main[j].grid; t.dims[j] = g.main[j].rd; t.args[j] = g.main[j].a; lds = std::max(lds, g.main[j].lds); }
      static bool attr[8] = {};
      const int m = g.main[0].mode;
      const bool pipe = g.main[0].pipe;
      const void* fn = pipe ? (m == 0 ? (const void*)k_rnn_group<0, true> : m == 1 ? (const void*)k_rnn_group<1, true> : m == 2 ? (const void*)k_rnn_group<2, true> : (const void*)k_rnn_group<3, true>)
                            : (m == 0 ? (const void*)k_rnn_group<0> : m == 1 ? (const void*)k_rnn_group<1> : m == 2 ? (const void*)k_rnn_group<2> : (const void*)k_rnn_group<3>);
      if (!attr[m + (pipe ? 4 : 0)]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) return LDE_ERR_HIP;
        attr[m + (pipe ? 4 : 0)] = true;
      }
      void* argv[] = {(void*)&t};
      (void)hipLaunchKernel(fn, dim3(t.start[n]), dim3(pipe ? 128 : 64), argv, lds, stream);
    } else if (any) {
      for (int j = 0; j < n; j++)
        if (g.main_set[j]) hipLaunchKernelGGL(g.main[j].fn, dim3(g.main[j].grid), dim3(g.main[j].block), g.main[j].lds, stream, g.main[j].rd, g.main[j].a);
    }
  }
  // With a weight-gradient stream set (lde_set_dw_stream) the products, their slab sums and the initial-state sums of the whole group go
  // there — a branch beside whatever the caller enqueues next (the feature extractor's pullback needs only the sweeps' input gradients);
  // every stack's event is recorded behind them (its next call waits for it before its workspace is rewritten).
  hipStream_t origin = stream;
  bool forked = false;
  if ((g.ndw > 0 || g.ns0 > 0) && g.nhs > 0 && g.hs[0]->opt_async_dw) {
    bool ok = true;
    hipStream_t wst = dw_sync_switch(g.hs[0]->dws, stream, &ok);
    if (!ok) return LDE_ERR_HIP;
    forked = wst != stream;
    stream = wst;
  }
  if (g.ndw > 0) {   // the weight-gradient products of every (stack, cell), then their slab sums
    bool same = g.ndw >= 2;
    for (int j = 0; j < g.ndw; j++) same = same && g.dw[j].ndw == 1;
    static bool attr[3] = {false, false, false};
    if (same) {
      GroupTable<MlpDims, DwArgs, GROUP_MAX_DW> t{};
      GroupTable<MlpDims, ReduceArgs, GROUP_MAX_DW> u{};
      t.n = u.n = g.ndw;
      size_t lds = 0;
      for (int j = 0; j < g.ndw; j++) {
        const RnnRecDw& q = g.dw[j];
        t.start[j + 1] = t.start[j] + q.gx * q.gy * q.gz; t.gx[j] = q.gx; t.gy[j] = q.gy; t.dims[j] = q.dm; t.args[j] = q.da; lds = std::max(lds, q.lds);
        u.start[j + 1] = u.start[j] + (int)q.rgrid; u.dims[j] = q.rdm; u.args[j] = q.ra;
      }
      if (!attr[0]) {
        if (hipFuncSetAttribute((const void*)k_mlp_dw_group<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) return LDE_ERR_HIP;
        attr[0] = true;
      }
      void* argv[] = {(void*)&t};
      (void)hipLaunchKernel((const void*)k_mlp_dw_group<1, false>, dim3(t.start[g.ndw]), dim3(512), argv, lds, stream);
      if (g.ns0 >= 2) {   // … and the initial-state sums ride on the slab sums' launch
        ReduceState0Tables rs{};
        rs.u = u;
        rs.s.n = g.ns0;
        for (int j = 0; j < g.ns0; j++) { rs.s.start[j + 1] = rs.s.start[j] + (int)g.s0[j].grid; rs.s.dims[j] = g.s0[j].rd; rs.s.args[j] = g.s0[j].a; }
        void* argr[] = {(void*)&rs};
        (void)hipLaunchKernel((const void*)k_rnn_reduce_state0_group, dim3(u.start[g.ndw] + rs.s.start[g.ns0]), dim3(256), argr, 0, stream);
        g.ns0 = 0;
      } else {
        void* argu[] = {(void*)&u};
        (void)hipLaunchKernel((const void*)k_reduce_tiles_group, dim3(u.start[g.ndw]), dim3(256), argu, 0, stream);
      }
    } else {
      std::string err;
      for (int j = 0; j < g.ndw; j++) {
        const RnnRecDw& q = g.dw[j];
        const int rc = launch_weight_gradient(q.dm, q.da, q.gx, q.gy, nullptr, q.ra.nflush, 0, q.ra.dW, q.ra.feedback, stream, err, q.ra.assign != 0);
        if (rc) return rc;
      }
    }
  }
  if (g.ns0 > 0) {
    if (g.ns0 >= 2) {
      GroupTable<RnnDims, State0Args, RNN_GROUP_MAX> t{};
      t.n = g.ns0;
      for (int j = 0; j < g.ns0; j++) { t.start[j + 1] = t.start[j] + (int)g.s0[j].grid; t.dims[j] = g.s0[j].rd; t.args[j] = g.s0[j].a; }
      void* argv[] = {(void*)&t};
      (void)hipLaunchKernel((const void*)k_rnn_state0_group, dim3(t.start[g.ns0]), dim3(64), argv, 0, stream);
    } else {
      const RnnRecS0& q = g.s0[0];
      hipLaunchKernelGGL(k_rnn_state0, dim3(q.grid), dim3(64), 0, stream, q.a.g0, q.a.B, q.a.g0w, q.rd, q.a.dW, q.a.assign);
    }
  }
#if LDE_DW_DEBUG
  { hipError_t e = hipPeekAtLastError(); if (e != hipSuccess) fprintf(stderr, "[dw rnn flush] before the events: %s\n", hipGetErrorString(e)); }
#endif
  if (forked)
    for (int j = 0; j < g.nhs; j++)
      if (!dw_sync_ensure(g.hs[j]->dws) || !dw_sync_end(g.hs[j]->dws, stream, origin)) {
#if LDE_DW_DEBUG
        fprintf(stderr, "[dw rnn flush] dw_sync_end %d failed: %s\n", j, hipGetErrorString(hipPeekAtLastError()));
#endif
        return LDE_ERR_HIP;
      }
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}
static bool rnn_group_ok(int n) {
#if LDE_PROF
  return false;
#else
  return n >= 2 && n <= RNN_GROUP_MAX;   // (with a weight-gradient stream set, a group's jobs stay on the caller's stream)
#endif
}
static bool rnn_group_fits(int n, lde_rnn* const* rs) {   // every (stack, cell) weight-gradient job must fit one table; a handle's workspace serves one call at a time
  int jobs = 0;
  for (int i = 0; i < n; i++) {
    jobs += rs[i]->rd.nL;
    for (int j = i + 1; j < n; j++)
      if (rs[i] == rs[j]) return false;
  }
  return jobs <= GROUP_MAX_DW;
}
int lde_rnn_group_forward(int n, lde_rnn* const* rs, const float* const* xs, int T, int B, float* const* ys, void* stream) {
  if (n < 1 || !rs || !xs || !ys) return LDE_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++)
    if (!rs[i]) return LDE_ERR_INVALID_ARG;
  if (!rnn_group_ok(n) || !rnn_group_fits(n, rs)) {
    for (int i = 0; i < n; i++) {
      const int rc = lde_rnn_forward(rs[i], xs[i], T, B, ys[i], stream);
      if (rc) return rc;
    }
    return LDE_OK;
  }
  RnnGroupRec g;
  t_rrec = &g;
  for (int i = 0; i < n; i++) {
    g.n = i;
    const int rc = lde_rnn_forward(rs[i], xs[i], T, B, ys[i], stream);
    if (rc) { t_rrec = nullptr; return rc; }
  }
  g.n = n;
  t_rrec = nullptr;
  const int rc = rnn_group_flush(g, (hipStream_t)stream);
  if (rc) rs[0]->err = "lde_rnn_group_forward: launch failed";
  return rc;
}
int lde_rnn_group_forward_train(int n, lde_rnn* const* rs, const float* const* xs, int T, int B, float* const* ys, void* stream) {
  if (n < 1 || !rs || !xs || !ys) return LDE_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++)
    if (!rs[i]) return LDE_ERR_INVALID_ARG;
  if (!rnn_group_ok(n) || !rnn_group_fits(n, rs)) {
    for (int i = 0; i < n; i++) {
      const int rc = lde_rnn_forward_train(rs[i], xs[i], T, B, ys[i], stream);
      if (rc) return rc;
    }
    return LDE_OK;
  }
  RnnGroupRec g;
  t_rrec = &g;
  for (int i = 0; i < n; i++) {
    g.n = i;
    const int rc = lde_rnn_forward_train(rs[i], xs[i], T, B, ys[i], stream);
    if (rc) { t_rrec = nullptr; return rc; }
  }
  g.n = n;
  t_rrec = nullptr;
  const int rc = rnn_group_flush(g, (hipStream_t)stream);
  if (rc) rs[0]->err = "lde_rnn_group_forward_train: launch failed";
  return rc;
}
int lde_rnn_group_backward(int n, lde_rnn* const* rs, const float* const* xs, const float* const* dys, int T, int B, float* const* dxs,
                           float* const* dWs, void* stream) {
  if (n < 1 || !rs || !xs || !dys || !dWs) return LDE_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++)
    if (!rs[i] || !dWs[i]) return LDE_ERR_INVALID_ARG;
  if (!rnn_group_ok(n) || !rnn_group_fits(n, rs)) {
    for (int i = 0; i < n; i++) {
      const int rc = lde_rnn_backward(rs[i], xs[i], dys[i], T, B, dxs ? dxs[i] : nullptr, dWs[i], stream);
      if (rc) return rc;
    }
    return LDE_OK;
  }
  RnnGroupRec g;
  t_rrec = &g;
  g.nhs = n;
  for (int i = 0; i < n; i++) g.hs[i] = rs[i];
  for (int i = 0; i < n; i++) {
    g.n = i;
    int rc = lde_rnn_backward_dx(rs[i], xs[i], dys[i], T, B, dxs ? dxs[i] : nullptr, stream);
    if (!rc) rc = lde_rnn_backward_dw(rs[i], dWs[i], stream);
    if (rc) { t_rrec = nullptr; return rc; }
  }
  g.n = n;
  t_rrec = nullptr;
  const int rc = rnn_group_flush(g, (hipStream_t)stream);
  if (rc) rs[0]->err = "lde_rnn_group_backward: launch failed";
  return rc;
}

// The grouped calls with the stacks' outputs (and output gradients) as column blocks of wider [B × ld] arrays, and a second source for
// each output gradient: what the GOKU encoder needs to run pattern extractor → vcat → latent_in and back without a concatenation, two
// strided copies and two additions as launches of their own (include/lde.h).
static int rnn_ld_ok(int n, lde_rnn* const* rs, const int* lds, const char* what) {
  for (int i = 0; i < n; i++) {
    if (!rs[i]) return LDE_ERR_INVALID_ARG;
    if (lds && lds[i] < rs[i]->rd.sizes[rs[i]->rd.nL]) {
      rs[i]->err = std::string(what) + ": a row stride below the stack's output width";
      return LDE_ERR_INVALID_ARG;
    }
  }
  return LDE_OK;
}
int lde_rnn_group_forward_ld(int n, lde_rnn* const* rs, const float* const* xs, int T, int B, float* const* ys, const int* ldys, int train,
                             void* stream) {
  if (n < 1 || !rs || !xs || !ys) return LDE_ERR_INVALID_ARG;
  int rc = rnn_ld_ok(n, rs, ldys, "lde_rnn_group_forward_ld");
  if (rc) return rc;
  for (int i = 0; i < n; i++) rs[i]->io_ldy = ldys ? ldys[i] : 0;
  rc = train ? lde_rnn_group_forward_train(n, rs, xs, T, B, ys, stream) : lde_rnn_group_forward(n, rs, xs, T, B, ys, stream);
  for (int i = 0; i < n; i++) rs[i]->io_ldy = 0;
  return rc;
}
int lde_rnn_group_backward_ld(int n, lde_rnn* const* rs, const float* const* xs, const float* const* dys, const float* const* dys2,
                              const int* lddys, int T, int B, float* const* dxs, float* const* dWs, void* stream) {
  if (n < 1 || !rs || !xs || !dys || !dWs) return LDE_ERR_INVALID_ARG;
  int rc = rnn_ld_ok(n, rs, lddys, "lde_rnn_group_backward_ld");
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    rs[i]->io_lddy = lddys ? lddys[i] : 0;
    rs[i]->io_dy2 = dys2 ? dys2[i] : nullptr;
  }
  rc = lde_rnn_group_backward(n, rs, xs, dys, T, B, dxs, dWs, stream);
  for (int i = 0; i < n; i++) {
    rs[i]->io_lddy = 0;
    rs[i]->io_dy2 = nullptr;
  }
  return rc;
}

int lde_rnn_set_accumulate(lde_rnn* r, int on) {
  if (!r) return LDE_ERR_INVALID_ARG;
  r->accumulate = on != 0;
  return LDE_OK;
}

int lde_rnn_set_option(lde_rnn* r, const char* key, double value) {
  if (!r || !key) return LDE_ERR_INVALID_ARG;
  int* slot = !std::strcmp(key, "generic") ? &r->opt_generic : !std::strcmp(key, "regw") ? &r->opt_regw : !std::strcmp(key, "pipe") ? &r->opt_pipe
              : !std::strcmp(key, "async_dw") ? &r->opt_async_dw : nullptr;
  if (!slot || !(value >= 0)) {
    r->err = std::string("lde_rnn_set_option: unknown key or negative value: ") + key;
    return LDE_ERR_INVALID_ARG;
  }
  *slot = (int)value;
  for (auto& km : r->kernel)      // the instantiations are cached per (mode, form): picked afresh under the new options
    for (auto& k : km) k = nullptr;
  return LDE_OK;
}

const char* lde_rnn_last_error(const lde_rnn* r) { return r ? r->err.c_str() : "null handle"; }

}  // extern "C"

// lde_chain.hip — the dense chains either side of the solve (scope row f-1).
//
// Replaces what runs under
//     l̂ = apply_latent_out(decoder, l̃)       [REF src/models/GOKU.jl:83-91], [REF src/models/LatentODE.jl:53]
//     x̂ = apply_reconstructor(decoder, ẑ)     [REF src/models/GOKU.jl:148],   [REF src/models/LatentODE.jl:80]
// i.e. Flux `Chain`s of `Dense(in,out,act)` / `SkipConnection(Dense,+)` applied column-wise (Flux 0.13.6, un-vendored),
// and their pullbacks. The reconstructor is the first genuinely matrix-core-heavy operator next to the solve:
// 2·(D·200 + 2·200² + 200·784) ≈ 474 kflop per (trajectory, save time), N = B·T columns.
//
// Design (gfx950)
//  * Forward: a workgroup of 8 waves owns NC = 16·CG columns (CG = 4 when the panels fit the 160 KB LDS) and runs the
//    whole chain on them: activations live in LDS as transposed panels (stride ≡ 8 mod 32 floats, as in lde_mlp.hip),
//    weights stream from L2 in MFMA-fragment order through a 4-deep register ring, every A fragment is used for CG
//    column groups × 2 row tiles (32 `v_mfma_f32_16x16x4_f32` per 2 fragment loads). Bias, activation and the skip
//    addition are fused into the epilogue; the last layer's epilogue stores x̂ straight to HBM (16 B per lane).
//  * Backward, kernel 1 (per column tile): recompute the hidden activations, stage every layer's input panel a_l in
//    HBM; δ_L = dy ⊙ act'(y) is formed from the caller's y, staged, and read back as the B operand of W_Lᵀ·δ_L (the
//    widest layer never touches LDS); then δ flows down the chain through Wᵀ fragments, each δ_l staged as it appears.
//    The gradient wrt a layer's output is kept in its own panel so that skip connections add it back.
//  * Backward, kernel 2: the weight gradient is the same large-K product over staged (a_l, δ_l) panels as in the MLP
//    adjoint — `k_mlp_dw` + `k_reduce_tiles` from lde_mfma.h, with every 16-column group as one slot of weight 1.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lde_mfma.h"

namespace lde {

struct ChainDims {
  MlpDims dm;          // sizes + offsets (fill_layer_offsets)
  int act[MAXL];
  int skip[MAXL];
  int ld0;             // stride of the input panel (≥ pad32(in)); 0 when the first layer reads x straight from HBM (gx)
  int f_off[MAXL];     // skip layers: offset (in a staged block) of the extra panel holding act(W·x+b) BEFORE the skip addition
  int sv_pre[MAXL];    // saved-activation buffer: hidden layer l's output [h_l × N] starts at N·sv_pre[l] floats, its pre-skip copy (skip
                       // layers) right after it
  int sv_total;        // floats per column of that buffer
  int gx;              // wide input (in % 16 == 0): x[in×N] itself is the B operand of layer 0 — no input panel, wider tiles
  int ldh;             // stride of hidden / gradient panels
};

// Activations on the hardware's v_exp_f32 / v_log_f32 / v_rcp_f32 (a few instructions each, ≈ 1e-7 ABSOLUTE error). The libm forms
// (tanhf, log1pf, expf: hundreds of instructions each) were inlined into every unrolled epilogue copy — 4 values × column groups × 2 row
// tiles × every chain_mac variant: 45–90 k instructions per chain kernel, several times the instruction cache, for 700 MFMAs.
__device__ __forceinline__ float cact(int kind, float x) {
  switch (kind) {
    case LDE_CACT_RELU: return fmaxf(x, 0.f);
    case LDE_CACT_TANH: {   // |x| < 1/4: the odd series (next term 8e-9 relative); beyond: (1 − e)/(1 + e), e = exp(−2|x|) ≤ 0.61 — no cancellation
      const float t = fabsf(x), t2 = t * t, e = __expf(-2.0f * t);
      const float big = (1.0f - e) * fast_rcp(1.0f + e);
      const float sm = t * (1.0f + t2 * (-0.33333334f + t2 * (0.13333334f + t2 * (-0.053968254f + t2 * 0.021869488f))));
      return copysignf(t < 0.25f ? sm : big, x);
    }
    case LDE_CACT_SIGMOID: return fast_rcp(1.0f + __expf(-x));   // v_exp_f32 + v_rcp_f32: ≈ 1e-7 absolute on (0,1)
    case LDE_CACT_SOFTPLUS: return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x)));
    default: return x;
  }
}
// The same on four values with ONE wave-uniform branch on the kind. Written per value inside the epilogue's unrolled loop, the switch
// was if-converted: every value evaluated EVERY activation (three v_exp, a v_log, three v_rcp, the tanh series …) and selected — ≈ 900
// cycles per epilogue call, measured with in-kernel stamps: the epilogues, not the K loops, were most of a chain kernel.
__device__ __forceinline__ f32x4 cact4(int kind, f32x4 x) {
  const int k = __builtin_amdgcn_readfirstlane(kind);
  f32x4 r;
  if (k == LDE_CACT_RELU) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = fmaxf(x[q], 0.f);
  } else if (k == LDE_CACT_SIGMOID) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = cact(LDE_CACT_SIGMOID, x[q]);
  } else if (k == LDE_CACT_TANH) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = cact(LDE_CACT_TANH, x[q]);
  } else if (k == LDE_CACT_SOFTPLUS) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = cact(LDE_CACT_SOFTPLUS, x[q]);
  } else
    r = x;
  return r;
}
__device__ __forceinline__ f32x4 cact_grad_out4(int kind, f32x4 f) {
  const int k = __builtin_amdgcn_readfirstlane(kind);
  f32x4 r;
  if (k == LDE_CACT_RELU) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = f[q] > 0.f ? 1.f : 0.f;
  } else if (k == LDE_CACT_SIGMOID) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = f[q] * (1.f - f[q]);
  } else if (k == LDE_CACT_TANH) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = 1.f - f[q] * f[q];
  } else if (k == LDE_CACT_SOFTPLUS) {
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = 1.f - __expf(-f[q]);
  } else
    r = f32x4{1.f, 1.f, 1.f, 1.f};
  return r;
}
// derivative of the activation expressed through its OUTPUT f = act(pre)
__device__ __forceinline__ float cact_grad_out(int kind, float f) {
  switch (kind) {
    case LDE_CACT_RELU: return f > 0.f ? 1.f : 0.f;
    case LDE_CACT_TANH: return 1.f - f * f;
    case LDE_CACT_SIGMOID: return f * (1.f - f);
    case LDE_CACT_SOFTPLUS: return 1.f - __expf(-f);
    default: return 1.f;
  }
}

// One wave: NT_ (1 or 2) 16-row tiles × NCG column groups, all K-groups, software-pipelined: A fragments (global/L2) run
// PFA K-groups ahead in a register ring, B operands PFB ahead; pre() is called before the K loop (its loads overlap the
// MFMAs), epi() after. Per K-group NT_ fragment loads + NCG operand loads feed 4·NT_·NCG MFMAs.
template <int NT_, int NCG, bool BGLB, bool BF, class Pre, class Epi>
__device__ __forceinline__ void chain_mac(const f32x4* A0, const f32x4* A1, const float* bp, int cgstride, int KG, int row_a,
                                          int row_b, int cg0, int col, Pre pre, Epi epi) {
  constexpr int PFA = 4, PFB = BGLB ? 4 : 2;
  const int kl = KG - 1;
  f32x4 acc0[NCG], acc1[NCG];
  decltype(pre(0, 0, 0)) p0[NCG], p1[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++) {
    acc0[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    p0[cg] = pre(row_a, cg0 + cg, col);
    if (NT_ == 2) p1[cg] = pre(row_b >= 0 ? row_b : row_a, cg0 + cg, col);
  }
  f32x4 ra0[PFA], ra1[PFA], rb[PFB][NCG];
#pragma unroll
  for (int i = 0; i < PFA; i++) {
    const int k = min(i, kl);
    ra0[i] = A0[k * 64];
    if (NT_ == 2) ra1[i] = A1[k * 64];
  }
#pragma unroll
  for (int i = 0; i < PFB; i++) {
    const int k = min(i, kl);
#pragma unroll
    for (int cg = 0; cg < NCG; cg++) rb[i][cg] = *reinterpret_cast<const f32x4*>(bp + (cg0 + cg) * cgstride + k * 16);
  }
  for (int kg = 0; kg < KG; kg += PFA) {
#pragma unroll
    for (int i = 0; i < PFA; i++) {
      if (kg + i < KG) {
        const f32x4 c0 = ra0[i];
        f32x4 c1 = c0;
        if (NT_ == 2) c1 = ra1[i];
        f32x4 cb[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; cg++) cb[cg] = rb[i % PFB][cg];
        const int ka = min(kg + i + PFA, kl), kb = min(kg + i + PFB, kl);
        ra0[i] = A0[ka * 64];
        if (NT_ == 2) ra1[i] = A1[ka * 64];
#pragma unroll
        for (int cg = 0; cg < NCG; cg++) rb[i % PFB][cg] = *reinterpret_cast<const f32x4*>(bp + (cg0 + cg) * cgstride + kb * 16);
        if (BF) {   // bf16 operands: one 16x16x16 MFMA per (row tile, column group) and K-group
          const s16x4 a0 = cvt_bf16x4(c0), a1 = cvt_bf16x4(c1);
#pragma unroll
          for (int cg = 0; cg < NCG; cg++) {
            const s16x4 b4 = cvt_bf16x4(cb[cg]);
            acc0[cg] = mfma16_bf(a0, b4, acc0[cg]);
            if (NT_ == 2) acc1[cg] = mfma16_bf(a1, b4, acc1[cg]);
          }
        } else {
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) {   // K-step outer: 2·NCG independent accumulators between two uses of the same one
#pragma unroll
            for (int cg = 0; cg < NCG; cg++) {
              acc0[cg] = mfma16(c0[s4], cb[cg][s4], acc0[cg]);
              if (NT_ == 2) acc1[cg] = mfma16(c1[s4], cb[cg][s4], acc1[cg]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int cg = 0; cg < NCG; cg++) {
    epi(row_a, cg0 + cg, col, acc0[cg], p0[cg]);
    if (NT_ == 2 && row_b >= 0) epi(row_b, cg0 + cg, col, acc1[cg], p1[cg]);   // row_b < 0: the second tile was a stand-in
  }
}

// Y[R × 16·CG] = M[R×K] · B[K × 16·CG] for one workgroup of 8 waves. M as K4 fragments in global memory (L2-resident),
// B transposed: element (k, column c of group cg) at Bp[cg*cgstride + c*ldb + k], in LDS or (BGLB) in global memory.
// Row tiles are dealt 16 at a time (a wave takes tiles rt and rt+8 together); 9–15 left-over tiles make one more such
// pass, exactly 8 a single-tile pass, and fewer than 8 are dealt as (tile, column group) units so that the last pass
// still uses every wave (49 tiles of the 784-row layer: 3 passes + 1/8 instead of 4).
template <int CG, bool BGLB, bool BF, class Pre, class Epi>
__device__ __forceinline__ void chain_gemm(const float* __restrict__ gfrag, int R, int K, const float* Bp, int ldb,
                                           int cgstride, Pre pre, Epi epi) {
  constexpr int NW = 8;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int RT = cdiv(R, 16), KG = cdiv(K, 16);
  const f32x4* A = reinterpret_cast<const f32x4*>(gfrag) + lane;
  const float* bp = Bp + (lane & 15) * ldb + 4 * (lane >> 4);
  const int col = lane & 15, rsub = 4 * (lane >> 4);
  int base = 0;
  for (; base + 2 * NW <= RT; base += 2 * NW) {
    const int rt = base + wave, rt2 = rt + NW;
    chain_mac<2, CG, BGLB, BF>(A + (size_t)rt * KG * 64, A + (size_t)rt2 * KG * 64, bp, cgstride, KG, rt * 16 + rsub,
                           rt2 * 16 + rsub, 0, col, pre, epi);
  }
  if (RT - base > NW) {   // 9–15 tiles left: one more double pass; a wave without a second tile repeats its first
    const int rt = base + wave, rt2 = rt + NW;
    const bool two = rt2 < RT;
    chain_mac<2, CG, BGLB, BF>(A + (size_t)rt * KG * 64, A + (size_t)(two ? rt2 : rt) * KG * 64, bp, cgstride, KG,
                           rt * 16 + rsub, two ? rt2 * 16 + rsub : -1, 0, col, pre, epi);
    return;
  }
  if (RT - base == NW) {
    const int rt = base + wave;
    chain_mac<1, CG, BGLB, BF>(A + (size_t)rt * KG * 64, nullptr, bp, cgstride, KG, rt * 16 + rsub, 0, 0, col, pre, epi);
    return;
  }
  const int units = (RT - base) * CG;
  for (int u = wave; u < units; u += NW) {
    const int rt = base + u / CG, cg = u % CG;
    chain_mac<1, 1, BGLB, BF>(A + (size_t)rt * KG * 64, nullptr, bp, cgstride, KG, rt * 16 + rsub, 0, cg, col, pre, epi);
  }
}

struct NoPre {};

// copy rows [0, rows32) of the 16 columns of column group cg from an LDS panel to a staged block panel [col][rows32]
__device__ __forceinline__ void stage_panel(const float* panel, int ld, int rows32, float* dst) {
  const int col = threadIdx.x >> 5, l31 = threadIdx.x & 31;   // one half-wave per column (512 threads = 16 columns)
  for (int r4 = l31; 4 * r4 < rows32; r4 += 32)
    *reinterpret_cast<f32x4*>(dst + col * rows32 + 4 * r4) = *reinterpret_cast<const f32x4*>(panel + col * ld + 4 * r4);
}

struct ChainFwdArgs {
  const float* x;
  float* y;
  const float* frag;
  const float* Wflat;
  long long N;
  float* saved;        // training: hidden activations go here as well (lde_chain_forward_save)
  const float* mse_t;  // lde_chain_forward_save_mse: Σ (y − mse_t)² over this tile's own columns goes to mse_part[tile] from the last
  float* mse_part;     // layer's epilogue, while y is in registers (the loss's forward pass over x and x̂ as a launch of its own: gone)
};
// workgroup sum of one value per thread in a fixed order (wave butterfly, then the eight waves by index); valid in thread 0
// (cws: eight floats of the kernel's DYNAMIC LDS that no wave still reads — a static array would come off the dynamic maximum the
// kernels ask for)
__device__ __forceinline__ float chain_wg_sum(float v, float* cws) {
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) cws[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < 8; w++) s += cws[w];
  return s;
}

template <int CG>
__device__ __forceinline__ void chain_load_tile(const ChainDims& cd, const float* x, long long n0, long long N, float* X0,
                                                float* biasc, const float* Wflat, int nfloat_zero, float* zero_base) {
  constexpr int NC = 16 * CG;
  const int tid = threadIdx.x;
  for (int i = tid; i < nfloat_zero; i += 512) zero_base[i] = 0.f;   // pad rows / pad columns must be finite
  __syncthreads();
  const MlpDims& dm = cd.dm;
  for (int l = 0; l < dm.nL; l++)
    for (int i = tid; i < dm.sizes[l + 1]; i += 512) biasc[dm.bias_lin[l] + i] = Wflat[dm.b_off[l] + i];
  const int in0 = dm.sizes[0];
  if (!cd.gx) {
    for (int e = tid; e < NC * in0; e += 512) {
      const int c = e / in0, r = e - c * in0;
      if (n0 + c < N) X0[c * cd.ld0 + r] = x[(size_t)(n0 + c) * in0 + r];
    }
  }
  __syncthreads();
}

// first column of a workgroup's tile. With gx the ragged last tile is shifted back to end at N (its first `dup` columns
// repeat the previous tile's: same values are stored twice, and they get weight 0 in the weight gradient), so that
// every column a lane reads from x exists.
__device__ __forceinline__ long long chain_tile_start(const ChainDims& cd, int NC, long long N, int* dup, unsigned bx) {
  const long long n = (long long)bx * NC;
  long long n0 = n;
  if (cd.gx && n + NC > N) n0 = N - NC;
  *dup = (int)(n - n0);
  return n0;
}

// one hidden (non-last) layer: Y = [Xin +] act(W·Xin + b) into an LDS panel
// fstage (pullback only, skip layers only): the activation before the skip addition goes to the staged block as well —
// the derivative is taken from it. (Recovering it as h − x loses it when it is tiny next to x: relu′ flips from 1 to 0 for
// about one unit in 10⁷, which a test with 3·10⁵ columns caught as a 3 % error in one column's gradient.)
struct SaveTo { float* base; long long n0, N; };   // base == nullptr: nothing is saved

template <int CG, bool BF, bool BG = false>
__device__ __forceinline__ void chain_hidden_layer(const ChainDims& cd, int l, const float* frag, const float* biasc,
                                                   const float* Xin, int ldx, float* Y, float* fstage = nullptr,
                                                   SaveTo sv = SaveTo{nullptr, 0, 0}) {
  const MlpDims& dm = cd.dm;
  const int in = dm.sizes[l], out = dm.sizes[l + 1], actk = cd.act[l], skip = BG ? 0 : cd.skip[l], ldh = cd.ldh;
  const float* bias = biasc + dm.bias_lin[l];
  const int out32 = pad32(out);
  chain_gemm<CG, BG, BF>(frag + dm.frag_off[l], out, in, Xin, ldx, 16 * ldx, [](int, int, int) { return NoPre{}; },
                        [&](int row0, int cg, int col, f32x4 v, NoPre) {
                          const int c = cg * 16 + col;
                          f32x4 r = v;
#pragma unroll
                          for (int q = 0; q < 4; q++) r[q] += bias[min(row0 + q, out - 1)];
                          r = cact4(actk, r);
#pragma unroll
                          for (int q = 0; q < 4; q++) r[q] = row0 + q < out ? r[q] : 0.f;
                          if (skip && fstage && row0 < out32)
                            *reinterpret_cast<f32x4*>(fstage + (size_t)cg * dm.blk_floats + col * out32 + row0) = r;
                          const long long nsv = sv.n0 + c;
                          const bool dosv = sv.base && nsv < sv.N && row0 < out;
                          float* svp = sv.base + (size_t)sv.N * cd.sv_pre[l] + (size_t)nsv * out + row0;
                          auto put = [&](float* p, const f32x4& v4) {
                            if ((out & 3) == 0) *reinterpret_cast<f32x4*>(p) = v4;
                            else {
#pragma unroll
                              for (int q = 0; q < 4; q++)
                                if (row0 + q < out) p[q] = v4[q];
                            }
                          };
                          if (dosv && cd.skip[l]) put(svp + (size_t)sv.N * out, r);   // before the skip addition
                          if (skip) r += *reinterpret_cast<const f32x4*>(Xin + c * ldx + row0);   // in == out; pad rows are 0
                          if (dosv) put(svp, r);
                          *reinterpret_cast<f32x4*>(Y + c * ldh + row0) = r;
                        });
}

constexpr int F32_OCC = 1;   // waves per SIMD of the f32 chain kernels' launch bounds
template <int CG, bool BF>
__device__ __forceinline__ void chain_forward_body(const ChainDims& cd, const ChainFwdArgs& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float csm[];
  constexpr int NC = 16 * CG;
  const MlpDims& dm = cd.dm;
  const int nL = dm.nL, ldh = cd.ldh;
  float* X0 = csm;
  float* H0 = X0 + NC * cd.ld0;
  float* H1 = H0 + NC * ldh;
  float* biasc = H1 + NC * ldh;
  int dup;
  const long long n0 = chain_tile_start(cd, NC, a.N, &dup, bx);
  PROF_T(pc0);
  chain_load_tile<CG>(cd, a.x, n0, a.N, X0, biasc, a.Wflat, NC * cd.ld0 + 2 * NC * ldh, csm);
  PROF_T(pc1);
  PROF_ADD(0, pc0, pc1);
  const float* Xin = X0;
  int ldx = cd.ld0;
  const float* xg = a.x + (size_t)n0 * dm.sizes[0];   // gx: column c of the tile at xg + c·in
  for (int l = 0; l + 1 < nL; l++) {
    float* Y = (l & 1) ? H1 : H0;
    PROF_T(pl0);
    const SaveTo sv{a.saved, n0, a.N};
    if (l == 0 && cd.gx) chain_hidden_layer<CG, BF, true>(cd, 0, a.frag, biasc, xg, dm.sizes[0], Y, nullptr, sv);
    else chain_hidden_layer<CG, BF>(cd, l, a.frag, biasc, Xin, ldx, Y, nullptr, sv);
    PROF_T(pl1);
    __syncthreads();
    PROF_T(pl2);
    PROF_ADD(2 + 2 * l, pl0, pl1);
    PROF_ADD(3 + 2 * l, pl1, pl2);
    Xin = Y;
    ldx = ldh;
  }
  PROF_T(pz0);
  {  // last layer: straight to HBM
    const int l = nL - 1, in = dm.sizes[l], out = dm.sizes[l + 1], actk = cd.act[l];
    const float* bias = biasc + dm.bias_lin[l];
    const bool vec = (out & 3) == 0;
    float msum = 0.f;
    auto epi_last = [&](int row0, int cg, int col, f32x4 v, NoPre) {
                            const long long n = n0 + cg * 16 + col;
                            if (n >= a.N || row0 >= out) return;
                            f32x4 r = v;
#pragma unroll
                            for (int q = 0; q < 4; q++) r[q] += bias[min(row0 + q, out - 1)];
                            r = cact4(actk, r);
                            if (a.mse_t && cg * 16 + col >= dup) {   // (a shifted last tile repeats `dup` columns of its neighbour: theirs)
                              const float* tp = a.mse_t + (size_t)n * out + row0;
                              if (vec) {   // (one 16-byte load, like the store below)
                                const f32x4 t4 = *reinterpret_cast<const f32x4*>(tp);
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                  const float d = r[q] - t4[q];
                                  msum = __builtin_fmaf(d, d, msum);
                                }
                              } else {
#pragma unroll
                                for (int q = 0; q < 4; q++)
                                  if (row0 + q < out) {
                                    const float d = r[q] - tp[q];
                                    msum = __builtin_fmaf(d, d, msum);
                                  }
                              }
                            }
                            float* yp = a.y + (size_t)n * out + row0;
                            if (vec) *reinterpret_cast<f32x4*>(yp) = r;
                            else {
#pragma unroll
                              for (int q = 0; q < 4; q++)
                                if (row0 + q < out) yp[q] = r[q];
                            }
                          };
    auto nopre = [](int, int, int) { return NoPre{}; };
    if (nL == 1 && cd.gx) chain_gemm<CG, true, BF>(a.frag + dm.frag_off[l], out, in, xg, in, 16 * in, nopre, epi_last);
    else chain_gemm<CG, false, BF>(a.frag + dm.frag_off[l], out, in, Xin, ldx, 16 * ldx, nopre, epi_last);
    if (a.mse_t) {
      const float t = chain_wg_sum(msum, csm);   // (the panels are free behind the barrier inside)
      if (threadIdx.x == 0) a.mse_part[bx] = t;
    }
  }
  PROF_T(pz1);
  PROF_ADD(2 + 2 * (nL - 1), pz0, pz1);
  PROF_ADD(40, pc0, pz1);
}
template <int CG, bool BF>
__global__ void __launch_bounds__(512, (CG <= 2 ? F32_OCC : 1)) k_chain_forward(ChainDims cd, ChainFwdArgs a) {
  chain_forward_body<CG, BF>(cd, a, blockIdx.x);
}
template <int CG, bool BF>
__global__ void __launch_bounds__(512, (CG <= 2 ? F32_OCC : 1)) k_chain_forward_group(GroupTable<ChainDims, ChainFwdArgs> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  chain_forward_body<CG, BF>(g.dims[j], g.args[j], blockIdx.x - g.start[j]);
}

struct ChainBwdArgs {
  const float* x;
  const float* y;
  const float* dy;
  float* dx;
  const float* frag;
  const float* fragT;
  const float* Wflat;
  float* stage;        // [slots][blk_floats]
  float* wts;          // [slots][16]
  long long N;
  const float* saved;  // hidden activations written by lde_chain_forward_save (nullptr: recompute them)
  const float* dy2;    // lde_chain_backward_saved_sum: the output gradient is (dy + dy2) + dy3 (nullptr: absent) — the sum a caller whose
  const float* dy3;    // output feeds several consumers would otherwise form with launches of its own
  const float* mse_t;  // lde_chain_backward_saved_mse: dy is not read; the first source of the output gradient is 2·(g·scale)·(y − mse_t),
  const float* mse_g;  // the pullback of base + scale·Σ(y − t)² [REF model_train.jl:225-238] — formed here instead of by a launch that
  float mse_scale;     // reads x and x̂ and writes ∂L/∂x̂ (the largest array of a GOKU step) for this kernel to read back
};

struct PrePair { f32x4 h, a; };

template <int CG, bool BF>
__device__ __forceinline__ void chain_backward_body(const ChainDims& cd, const ChainBwdArgs& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float csm[];
  constexpr int NC = 16 * CG;
  const MlpDims& dm = cd.dm;
  const int nL = dm.nL, ldh = cd.ldh, tid = threadIdx.x;
  float* X0 = csm;
  float* P0 = X0 + NC * cd.ld0;
  float* P1 = P0 + NC * ldh;
  float* G = P1 + NC * ldh;
  float* biasc = G + NC * ldh;
  int dup;
  const long long n0 = chain_tile_start(cd, NC, a.N, &dup, bx);
  const size_t slot0 = (size_t)bx * CG;
  float* const blk0 = a.stage + slot0 * dm.blk_floats;   // the CG staged blocks of this tile are contiguous
  chain_load_tile<CG>(cd, a.x, n0, a.N, X0, biasc, a.Wflat, NC * cd.ld0 + 3 * NC * ldh, csm);
  const float* xg = a.x + (size_t)n0 * dm.sizes[0];

  // ---- 1. recompute the hidden activations; stage every layer's input panel ---------------------------------------
  if (cd.gx) {   // a_0 straight from x (every column of a shifted tile exists)
    const int in0 = dm.sizes[0], in32 = pad32(in0), col = tid >> 5, l31 = tid & 31;
#pragma unroll
    for (int cg = 0; cg < CG; cg++) {
      float* dst = blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[0] + col * in32;
      const float* src = xg + (size_t)(cg * 16 + col) * in0;
      for (int r4 = l31; 4 * r4 < in32; r4 += 32)
        *reinterpret_cast<f32x4*>(dst + 4 * r4) = 4 * r4 < in0 ? *reinterpret_cast<const f32x4*>(src + 4 * r4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
#pragma unroll
    for (int cg = 0; cg < CG; cg++)
      stage_panel(X0 + cg * 16 * cd.ld0, cd.ld0, pad32(dm.sizes[0]), blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[0]);
  }
  if (a.saved) {   // the forward call kept the hidden activations: copy them into the staged panels, no recomputation
    const int col = tid >> 5, l31 = tid & 31;
    for (int l = 0; l + 1 < nL; l++) {
      const int h = dm.sizes[l + 1], h32 = pad32(h);
      const bool vec = (h & 3) == 0;
#pragma unroll
      for (int cg = 0; cg < CG; cg++) {
        const long long n = n0 + cg * 16 + col;
        const float* sh = a.saved + (size_t)a.N * cd.sv_pre[l] + (size_t)n * h;
        float* dst = blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[l + 1] + col * h32;
        float* dstf = blk0 + (size_t)cg * dm.blk_floats + cd.f_off[l] + col * h32;
        for (int r4 = l31; 4 * r4 < h32; r4 += 32) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f}, vf = v;
          if (n < a.N && 4 * r4 < h) {
            if (vec) {
              v = *reinterpret_cast<const f32x4*>(sh + 4 * r4);
              if (cd.skip[l]) vf = *reinterpret_cast<const f32x4*>(sh + (size_t)a.N * h + 4 * r4);
            } else {
#pragma unroll
              for (int q = 0; q < 4; q++)
                if (4 * r4 + q < h) {
                  v[q] = sh[4 * r4 + q];
                  if (cd.skip[l]) vf[q] = sh[(size_t)a.N * h + 4 * r4 + q];
                }
            }
          }
          *reinterpret_cast<f32x4*>(dst + 4 * r4) = v;
          if (cd.skip[l]) *reinterpret_cast<f32x4*>(dstf + 4 * r4) = vf;
        }
      }
    }
  } else {
    const float* Xin = X0;
    int ldx = cd.ld0;
    for (int l = 0; l + 1 < nL; l++) {
      float* Y = (l & 1) ? P1 : P0;
      if (l == 0 && cd.gx) chain_hidden_layer<CG, BF, true>(cd, 0, a.frag, biasc, xg, dm.sizes[0], Y);
      else chain_hidden_layer<CG, BF>(cd, l, a.frag, biasc, Xin, ldx, Y, cd.skip[l] ? blk0 + cd.f_off[l] : nullptr);
      __syncthreads();
#pragma unroll
      for (int cg = 0; cg < CG; cg++)
        stage_panel(Y + cg * 16 * ldh, ldh, pad32(dm.sizes[l + 1]), blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[l + 1]);
      Xin = Y;
      ldx = ldh;
    }
  }
  // ---- 2. δ_L = dy ⊙ act'(y) from the caller's arrays, staged as layer L's δ-panel; column weights ------------------
  const int L1 = nL - 1;
  {
    const int out = dm.sizes[nL], out32 = pad32(out), actk = cd.act[L1];
    const bool vec = (out & 3) == 0;
    const int col = tid >> 5, l31 = tid & 31;
#pragma unroll
    for (int cg = 0; cg < CG; cg++) {
      float* dst = blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[L1] + NB * pad32(dm.sizes[L1]) + col * out32;
      const long long n = n0 + cg * 16 + col;
      const float* dyp = a.dy + (size_t)n * out;
      const float* yp = a.y + (size_t)n * out;
      for (int r4 = l31; 4 * r4 < out32; r4 += 32) {
        const int r = 4 * r4;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (n < a.N && r < out) {
          if (vec) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(yp + r);
            f32x4 g;
            if (a.mse_t) {
              const float k2 = 2.0f * (a.mse_g[0] * a.mse_scale);
              const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.mse_t + (size_t)n * out + r);
#pragma unroll
              for (int q = 0; q < 4; q++) g[q] = __fmul_rn(k2, f[q] - t4[q]);   // (rounded on its own: the sum with a further cotangent must not fuse it)
            } else
              g = *reinterpret_cast<const f32x4*>(dyp + r);
            if (a.dy2) g += *reinterpret_cast<const f32x4*>(a.dy2 + (size_t)n * out + r);
            if (a.dy3) g += *reinterpret_cast<const f32x4*>(a.dy3 + (size_t)n * out + r);
            d = cact_grad_out4(actk, f);
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] *= g[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (r + q < out) {
                float g = a.mse_t ? __fmul_rn(2.0f * (a.mse_g[0] * a.mse_scale), yp[r + q] - a.mse_t[(size_t)n * out + r + q]) : dyp[r + q];
                if (a.dy2) g += a.dy2[(size_t)n * out + r + q];
                if (a.dy3) g += a.dy3[(size_t)n * out + r + q];
                d[q] = g * cact_grad_out(actk, yp[r + q]);
              }
          }
        }
        *reinterpret_cast<f32x4*>(dst + r) = d;
      }
      if (tid < 16) a.wts[(slot0 + cg) * NB + tid] = (n0 + cg * 16 + tid < a.N && cg * 16 + tid >= dup) ? 1.f : 0.f;   // repeated columns of a shifted tile: weight 0
    }
  }
  __syncthreads();   // the staged panels are read back below (s_waitcnt vmcnt(0) + barrier ⇒ they are in L2)

  // ---- 3. δ down the chain ------------------------------------------------------------------------------------------
  const float* Dcur = nullptr;   // LDS panel holding δ_l for l < L1
  for (int l = L1; l >= 0; l--) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const int skl = cd.skip[l];
    const float* fragT = a.fragT + dm.fragT_off[l];
    const float* Bglb = blk0 + dm.blk_off[l] + NB * pad32(in);   // δ_l panel of column group 0 (only used for l == L1)
    if (l > 0) {
      float* Dn = (l & 1) ? P1 : P0;
      const int actp = cd.act[l - 1], skp = cd.skip[l - 1];
      const int in32 = pad32(in), inp32 = pad32(dm.sizes[l - 1]);
      // activation output of layer l-1: its staged output a_l, or — for a skip layer — the extra panel written before the
      // skip addition (rows padded to 32 like every staged panel; in == out for a skip layer, so the strides agree)
      const float* hblk = skp ? blk0 + cd.f_off[l - 1] : blk0 + dm.blk_off[l];
      auto pre = [&](int row0, int cg, int col) {
        PrePair p;
        p.h = *reinterpret_cast<const f32x4*>(hblk + (size_t)cg * dm.blk_floats + col * in32 + row0);
        p.a = f32x4{0.f, 0.f, 0.f, 0.f};
        return p;
      };
      auto epi = [&](int row0, int cg, int col, f32x4 v, PrePair p) {
        const int c = cg * 16 + col;
        f32x4 g = v;
        if (skl) g += *reinterpret_cast<const f32x4*>(G + c * ldh + row0);
        if (skp) *reinterpret_cast<f32x4*>(G + c * ldh + row0) = g;   // layer l-1 adds it back to what flows through it
        f32x4 d = cact_grad_out4(actp, p.h - p.a);
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = row0 + q < in ? g[q] * d[q] : 0.f;
        *reinterpret_cast<f32x4*>(Dn + c * ldh + row0) = d;
      };
      if (l == L1) chain_gemm<CG, true, BF>(fragT, in, out, Bglb, pad32(out), dm.blk_floats, pre, epi);
      else chain_gemm<CG, false, BF>(fragT, in, out, Dcur, ldh, 16 * ldh, pre, epi);
      __syncthreads();
#pragma unroll
      for (int cg = 0; cg < CG; cg++)
        stage_panel(Dn + cg * 16 * ldh, ldh, in32, blk0 + (size_t)cg * dm.blk_floats + dm.blk_off[l - 1] + NB * inp32);
      Dcur = Dn;
    } else if (a.dx) {
      auto pre = [](int, int, int) { return NoPre{}; };
      auto epi = [&](int row0, int cg, int col, f32x4 v, NoPre) {
        const int c = cg * 16 + col;
        const long long n = n0 + c;
        f32x4 g = v;
        if (skl) g += *reinterpret_cast<const f32x4*>(G + c * ldh + row0);
        if (n < a.N) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (row0 + q < in) a.dx[(size_t)n * in + row0 + q] = g[q];
        }
      };
      if (l == L1) chain_gemm<CG, true, BF>(fragT, in, out, Bglb, pad32(out), dm.blk_floats, pre, epi);
      else chain_gemm<CG, false, BF>(fragT, in, out, Dcur, ldh, 16 * ldh, pre, epi);
    }
  }
}
template <int CG, bool BF>
__global__ void __launch_bounds__(512, (CG <= 2 ? F32_OCC : 1)) k_chain_backward(ChainDims cd, ChainBwdArgs a) {
  chain_backward_body<CG, BF>(cd, a, blockIdx.x);
}
template <int CG, bool BF>
__global__ void __launch_bounds__(512, (CG <= 2 ? F32_OCC : 1)) k_chain_backward_group(GroupTable<ChainDims, ChainBwdArgs> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  chain_backward_body<CG, BF>(g.dims[j], g.args[j], blockIdx.x - g.start[j]);
}

#include "lde_chain_bf16.h"

}  // namespace lde

// ================================================ C ABI ======================================================
using namespace lde;

struct lde_chain {
  bool accumulate = true;   // pullback: dW += gradient (default) or dW = gradient
  bool bf16 = false;        // lde_chain_set_dtype: the native bf16 path (lde_chain_bf16.h): bf16 weight fragments, panels, saved activations, δ
  bool frag_stale = false;   // lde_refresh_weights rebuilt only the bf16 fragments (bf16 mode): the f32 ones are rebuilt from W_dev if the mode goes back
  __bf16* fragb = nullptr;  // bf16 K = 32 fragment copies of W / Wᵀ, rebuilt with the f32 ones
  __bf16* fragTb = nullptr;
  BfDims bd, bdx;           // bf16 layouts (bdx: the panel-free wide-input layout)
  int bcg_fwd = 0, bcg_bwd = 0, bcgx_fwd = 0, bcgx_bwd = 0;
  __bf16* dstage = nullptr; size_t dstage_cap = 0;       // δ_l matrices of the bf16 pullback
  int64_t delta_N = -1;     // lde_chain_forward_save_mse_delta left δ_L′ of this many columns in dstage (−1: none)
  const float* delta_saved = nullptr;   // … for the forward call whose saved-activation buffer this is: the token the pullback must present
  __bf16* svscratch = nullptr; size_t svscratch_cap = 0;  // lde_chain_backward (no saved buffer) in bf16 mode: its own forward pass saves here
  lde_chain_desc d;
  ChainDims cd;
  int64_t nW = 0;
  size_t nfrag = 0, nfragT = 0;
  float* W_dev = nullptr;
  float* frag = nullptr;
  float* fragT = nullptr;
  MlpDims* dm_dev = nullptr;   // device copy of cd.dm (lde_refresh_weights' job table points at it)
  bool have_W = false;
  int cg_fwd = 0, cg_bwd = 0;
  size_t lds_fwd = 0, lds_bwd = 0;
  ChainDims cdx;               // the panel-free layout for wide inputs (gx), when applicable
  int cgx_fwd = 0, cgx_bwd = 0;
  int opt_group = 1;        // lde_chain_set_option "group": this chain may take part in a merged (one launch per stage) grouped call
  int opt_async_dw = 1;     // lde_chain_set_option "async_dw": this chain's weight-gradient kernels go to the dw stream when one is set (lde_set_dw_stream)
  size_t ldsx_fwd = 0, ldsx_bwd = 0;
  // backward workspace
  float* stage = nullptr; size_t stage_cap = 0;
  float* wts = nullptr; size_t wts_cap = 0;
  float* slab = nullptr; size_t slab_cap = 0;
  int32_t* ints = nullptr; size_t ints_cap = 0;   // zero words for the slab reduction: [0] "no private slabs", [2..3] feedback sink
  DwSync dws;                  // weight-gradient kernels on the dw stream (lde_set_dw_stream)
  std::string err;
};

static int chain_desc_ok(const lde_chain_desc* d) {
  if (!d || d->abi_version != LDE_ABI_VERSION || d->n_layers < 1 || d->n_layers > LDE_CHAIN_MAX_LAYERS) return 0;
  for (int l = 0; l <= d->n_layers; l++)
    if (d->sizes[l] < 1) return 0;
  for (int l = 0; l < d->n_layers; l++) {
    if (d->activation[l] < 0 || d->activation[l] > LDE_CACT_SOFTPLUS) return 0;
    if (d->skip[l] && d->sizes[l] != d->sizes[l + 1]) return 0;
  }
  return 1;
}

static size_t chain_lds(const ChainDims& cd, int cg, int npanels) {
  const int NC = 16 * cg;
  return ((size_t)NC * cd.ld0 + (size_t)npanels * NC * cd.ldh + ((cd.dm.nbias + 3) & ~3)) * sizeof(float);
}

// native bf16 path: LDS bytes of the forward (two bf16 panels + input panel + biases) and pullback (two bf16 panels + the f32
// skip-gradient panel) kernels
static size_t chain_lds_b(const ChainDims& cd, const BfDims& bd, int cg, bool bwd) {
  const size_t NC = 16 * cg;
  if (bwd) return NC * bd.ldb * 2 * 2 + NC * bd.ldg * 4;
  return (2 + (bd.fpanel ? 1 : 0)) * NC * bd.ldb * 2 + (size_t)((cd.dm.nbias + 3) & ~3) * 4   // the input panel shares the second panel's space
         + (cd.gx ? chain_xs_bytes(cg) : 0);                                 // wide input: the first layer's chunk buffers
}

// lde_loss.hip: out[0] = (base ? base[0] : 0) + scale·Σ scratch[0..g) in index order (k_loss_final)
int loss_finalize(const float* scratch, int g, float scale, const float* base, float* out, hipStream_t stream);
// lde_rnn.hip: where lde_refresh_weights copies a recurrent stack's flat weights to (marks the handle as holding weights)
bool rnn_refresh_target(lde_rnn* r, float** W_dev, int64_t* nW);

extern "C" {

int64_t lde_chain_num_weights(const lde_chain_desc* d) {
  if (!chain_desc_ok(d)) return -1;
  int64_t n = 0;
  for (int l = 0; l < d->n_layers; l++) n += (int64_t)d->sizes[l + 1] * d->sizes[l] + d->sizes[l + 1];
  return n;
}

void lde_chain_destroy(lde_chain* c) {
  if (!c) return;
  if (c->W_dev) (void)hipFree(c->W_dev);
  if (c->frag) (void)hipFree(c->frag);
  if (c->fragT) (void)hipFree(c->fragT);
  if (c->fragb) (void)hipFree(c->fragb);
  if (c->fragTb) (void)hipFree(c->fragTb);
  if (c->dstage) (void)hipFree(c->dstage);
  if (c->svscratch) (void)hipFree(c->svscratch);
  if (c->dm_dev) (void)hipFree(c->dm_dev);
  if (c->stage) (void)hipFree(c->stage);
  if (c->wts) (void)hipFree(c->wts);
  if (c->slab) (void)hipFree(c->slab);
  if (c->ints) (void)hipFree(c->ints);
  dw_sync_destroy(c->dws);
  delete c;
}

int lde_chain_create(const lde_chain_desc* d, lde_chain** out) {
  if (!out) return LDE_ERR_INVALID_ARG;
  *out = nullptr;
  if (!chain_desc_ok(d)) return LDE_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return LDE_ERR_NO_DEVICE;   // no CPU fallback
  lde_chain* c = new lde_chain();
  c->d = *d;
  ChainDims& cd = c->cd;
  std::memset(&cd, 0, sizeof(cd));
  MlpDims& dm = cd.dm;
  dm.nL = d->n_layers;
  int hmax = 16;
  for (int l = 0; l <= d->n_layers; l++) {
    dm.sizes[l] = d->sizes[l];
    if (l > 0 && l < d->n_layers) hmax = std::max(hmax, d->sizes[l]);
  }
  for (int l = 0; l < d->n_layers; l++) {
    cd.act[l] = d->activation[l];
    cd.skip[l] = d->skip[l] ? 1 : 0;
  }
  fill_layer_offsets(dm, &c->nfrag, &c->nfragT);
  c->nW = dm.nW;
  for (int l = 0; l < d->n_layers; l++) {   // extra staged panel per skip layer (after the (a, δ) panels k_mlp_dw reads)
    cd.f_off[l] = 0;
    if (cd.skip[l]) {
      cd.f_off[l] = dm.blk_floats;
      dm.blk_floats += NB * pad32(dm.sizes[l + 1]);
    }
  }
  cd.sv_total = 0;
  for (int l = 0; l + 1 < d->n_layers; l++) {
    cd.sv_pre[l] = cd.sv_total;   // widths rounded up to 4 keep every region 16-byte aligned (N·sv_pre floats from the base)
    cd.sv_total += ((dm.sizes[l + 1] + 3) & ~3) * (cd.skip[l] ? 2 : 1);
  }
  cd.ld0 = panel_stride(pad32(dm.sizes[0]));
  cd.ldh = panel_stride(pad32(hmax));
  *out = c;   // from here on errors carry a message
  if (cd.skip[d->n_layers - 1] && d->n_layers > 1) {
    c->err = "a skip connection around the last layer of a chain is not supported";
    return LDE_ERR_UNSUPPORTED;
  }
  if (cd.skip[d->n_layers - 1]) {   // single skip layer: would need the gradient panel at the output width
    c->err = "a chain made of one skip layer is not supported";
    return LDE_ERR_UNSUPPORTED;
  }
  for (int l = 0; l <= d->n_layers; l++)
    if (d->sizes[l] > 1024) {
      c->err = "chain layer widths up to 1024 are supported";
      return LDE_ERR_UNSUPPORTED;
    }
  // Column groups per workgroup: the widest tile that still lets TWO workgroups share a CU (≤ 80 KB of LDS each) — one
  // tile's barrier / prologue latencies are then covered by the other's MFMAs. Measured on the reconstructor
  // (N = 12800): forward 64 columns 110 µs, 32 columns 107 µs; backward 32 columns 331 µs, 16 columns 313 µs.
  // Falls back to the widest tile that fits at all.
  auto pick = [&](const ChainDims& q, int* cgf, size_t* ldf, int* cgb, size_t* ldb) {
    *cgf = *cgb = 0;
    for (size_t lim : {LDS_MAX / 2, LDS_MAX}) {
      for (int cg : {4, 2, 1}) {
        if (!*cgf && chain_lds(q, cg, 2) <= lim) { *cgf = cg; *ldf = chain_lds(q, cg, 2); }
        if (cg <= 2 && !*cgb && chain_lds(q, cg, 3) <= lim) { *cgb = cg; *ldb = chain_lds(q, cg, 3); }
      }
    }
  };
  pick(cd, &c->cg_fwd, &c->lds_fwd, &c->cg_bwd, &c->lds_bwd);
  // wide inputs (an image encoder's first layer): x itself is the B operand of layer 0, no input panel
  c->cdx = cd;
  if (dm.sizes[0] % 16 == 0 && dm.sizes[0] >= 128 && !cd.skip[0]) {   // (lde_chain_set_option "gx" = 0 takes the layout away again: tests)
    c->cdx.gx = 1;
    c->cdx.ld0 = 0;
    pick(c->cdx, &c->cgx_fwd, &c->ldsx_fwd, &c->cgx_bwd, &c->ldsx_bwd);
  }
  if (!c->cg_fwd || !c->cg_bwd) {
    if (c->cgx_fwd && c->cgx_bwd) {   // only the panel-free layout fits: it needs N ≥ one tile (checked per call)
      c->cg_fwd = c->cg_bwd = 0;
    } else {
      c->err = "chain: activation panels do not fit the 160 KiB LDS";
      return LDE_ERR_UNSUPPORTED;
    }
  }
  {   // native bf16 layouts (lde_chain_bf16.h)
    BfDims& bd = c->bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.ldb = panel_stride_b(pad32(std::max(hmax, dm.sizes[0])));   // the input panel shares the second hidden panel's space
    bd.ld0 = bd.ldb;
    bd.ldg = cd.ldh;
    int off = 0;
    for (int l = 0; l < d->n_layers; l++) {
      bd.dl_w[l] = l == d->n_layers - 1 ? pad32(dm.sizes[l + 1]) : (dm.sizes[l + 1] + 7) & ~7;   // the last layer's rows are read back as a B operand: whole K-groups
      bd.dl_off[l] = off;
      off += bd.dl_w[l];
    }
    bd.dl_total = off;
    for (int l = 0; l + 1 < d->n_layers; l++) bd.fpanel = bd.fpanel || cd.skip[l];
    c->bdx = bd;
    c->bdx.ldb = panel_stride_b(pad32(hmax));   // wide input read in place: the panels only hold hidden vectors
    c->bdx.ld0 = 0;
    // 32-column tiles (CG = 2): ≤ 128 registers and ≤ 80 KB of LDS, so two workgroups share a CU (lde_chain_bf16.h: BF_PFA / BF_OCC);
    // (a 64-column forward instantiation exists; it is only picked when nothing narrower fits)
    auto pickb = [&](const ChainDims& q, const BfDims& b, int* cgf, int* cgb) {
      *cgf = *cgb = 0;
      for (size_t lim : {LDS_MAX / 2, LDS_MAX}) {
        for (int cg : {2, 1}) {
          if (!*cgf && chain_lds_b(q, b, cg, false) <= lim) *cgf = cg;
          if (!*cgb && chain_lds_b(q, b, cg, true) <= lim) *cgb = cg;
        }
      }
    };
    pickb(cd, bd, &c->bcg_fwd, &c->bcg_bwd);
    if (c->cdx.gx) pickb(c->cdx, c->bdx, &c->bcgx_fwd, &c->bcgx_bwd);
  }
  size_t nfb = 0, nfTb = 0;
  for (int l = 0; l < d->n_layers; l++) { nfb += bf_frag_elems(dm, l, false); nfTb += bf_frag_elems(dm, l, true); }
  if (hipMalloc(&c->W_dev, (size_t)c->nW * sizeof(float)) != hipSuccess ||
      hipMalloc(&c->frag, c->nfrag * sizeof(float)) != hipSuccess ||
      hipMalloc(&c->fragT, c->nfragT * sizeof(float)) != hipSuccess || hipMalloc(&c->dm_dev, sizeof(MlpDims)) != hipSuccess ||
      hipMalloc(&c->fragb, nfb * sizeof(__bf16)) != hipSuccess || hipMalloc(&c->fragTb, nfTb * sizeof(__bf16)) != hipSuccess) {
    c->err = "chain: hipMalloc failed";
    return LDE_ERR_ALLOC;
  }
  if (hipMemcpy(c->dm_dev, &c->cd.dm, sizeof(MlpDims), hipMemcpyHostToDevice) != hipSuccess) {
    c->err = "chain: hipMemcpy of the dimensions failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

static int chain_frags(lde_chain* c, const float* src, hipStream_t stream) {
  // src == W_dev: fragments only; otherwise the kernel also copies src into W_dev
  hipLaunchKernelGGL(k_build_frags, dim3(64, c->cd.dm.nL), dim3(256), 0, stream, src, c->cd.dm, c->frag, c->fragT,
                     src == c->W_dev ? (float*)nullptr : c->W_dev, c->fragb, c->fragTb);
  if (hipGetLastError() != hipSuccess) {
    c->err = "k_build_frags launch failed";
    return LDE_ERR_HIP;
  }
  c->have_W = true;
  c->frag_stale = false;
  return LDE_OK;
}

int lde_chain_set_weights(lde_chain* c, const float* flat_host, int64_t n) {
  if (!c || !c->W_dev) return LDE_ERR_INVALID_ARG;
  if (n != c->nW || !flat_host) {
    c->err = "lde_chain_set_weights: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  if (hipMemcpy(c->W_dev, flat_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    c->err = "lde_chain_set_weights: hipMemcpy failed";
    return LDE_ERR_HIP;
  }
  return chain_frags(c, c->W_dev, nullptr);
}

int lde_chain_set_weights_device(lde_chain* c, const float* flat_dev, int64_t n, void* stream) {
  if (!c || !c->W_dev) return LDE_ERR_INVALID_ARG;
  if (n != c->nW || !flat_dev) {
    c->err = "lde_chain_set_weights_device: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  return chain_frags(c, flat_dev, (hipStream_t)stream);
}

// One launch for every module of a model (see include/lde.h). The job table lives in device memory and is uploaded only
// when it differs from the previous call's (a training loop passes the same handles and pointers every step).
int lde_refresh_weights(int n, const int* kinds, void* const* handles, const float* const* flat_dev, void* stream) {
  static std::mutex mu;
  static std::vector<RefreshJob> last;
  static RefreshJob* jobs_dev = nullptr;
  static size_t jobs_cap = 0;
  if (n < 0 || (n > 0 && (!kinds || !handles || !flat_dev))) return LDE_ERR_INVALID_ARG;
  if (n == 0) return LDE_OK;
  std::vector<RefreshJob> jobs;
  for (int m = 0; m < n; m++) {
    if (!handles[m] || !flat_dev[m]) return LDE_ERR_INVALID_ARG;
    if (kinds[m] == LDE_MODULE_CHAIN) {
      lde_chain* c = (lde_chain*)handles[m];
      if (!c->W_dev || !c->dm_dev) return LDE_ERR_INVALID_ARG;
      for (int l = 0; l < c->cd.dm.nL; l++) {
        RefreshJob q;
        std::memset(&q, 0, sizeof(q));   // (the table is compared byte-wise with the previous call's: padding too)
        q.src = flat_dev[m]; q.keep = flat_dev[m] == c->W_dev ? nullptr : c->W_dev;
        q.frag = c->bf16 ? nullptr : c->frag; q.fragT = c->bf16 ? nullptr : c->fragT;   // (a bf16 chain reads bf16 fragments only: half the hand-over's work)
        q.g = frag_geom(c->cd.dm, l); q.layer = l; q.n = 0; q.fragb = c->fragb; q.fragTb = c->fragTb;
        jobs.push_back(q);
      }
    } else if (kinds[m] == LDE_MODULE_RNN) {
      float* dst = nullptr;
      int64_t nw = 0;
      if (!rnn_refresh_target((lde_rnn*)handles[m], &dst, &nw)) return LDE_ERR_INVALID_ARG;
      if (dst != flat_dev[m]) {
        RefreshJob q;
        std::memset(&q, 0, sizeof(q));
        q.src = flat_dev[m]; q.keep = dst; q.layer = -1; q.n = (int)nw;
        jobs.push_back(q);
      }
    } else
      return LDE_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(mu);
  hipStream_t st = (hipStream_t)stream;
  if (jobs.size() != last.size() || std::memcmp(jobs.data(), last.data(), jobs.size() * sizeof(RefreshJob)) != 0) {
    // rare path: an earlier launch may still be reading the table
    if (hipDeviceSynchronize() != hipSuccess) return LDE_ERR_HIP;
    if (jobs.size() > jobs_cap) {
      if (jobs_dev) (void)hipFree(jobs_dev);
      jobs_dev = nullptr;
      jobs_cap = 0;
      if (hipMalloc(&jobs_dev, jobs.size() * sizeof(RefreshJob)) != hipSuccess) return LDE_ERR_ALLOC;
      jobs_cap = jobs.size();
    }
    if (hipMemcpy(jobs_dev, jobs.data(), jobs.size() * sizeof(RefreshJob), hipMemcpyHostToDevice) != hipSuccess) return LDE_ERR_HIP;
    last = jobs;
  }
  if (!jobs.empty()) {
    hipLaunchKernelGGL(k_refresh_many, dim3(64, (unsigned)jobs.size()), dim3(256), 0, st, jobs_dev);
    if (hipGetLastError() != hipSuccess) return LDE_ERR_HIP;
  }
  for (int m = 0; m < n; m++)
    if (kinds[m] == LDE_MODULE_CHAIN) {
      lde_chain* c = (lde_chain*)handles[m];
      c->have_W = true;
      c->frag_stale = c->bf16;
    }
  return LDE_OK;
}

// which layout a call uses: the panel-free one (gx) when the input is wide, x is 16-byte aligned and N fills a tile
struct ChainPick { const ChainDims* cd; int cg; size_t lds; };
static bool chain_pick(const lde_chain* c, const float* x, int64_t N, bool bwd, ChainPick* p) {
  const int cgx = bwd ? c->cgx_bwd : c->cgx_fwd, cg = bwd ? c->cg_bwd : c->cg_fwd;
  // Mid-size batches (a training step's N = B·T ≈ 3 200 columns): the widest tile leaves most CUs without a workgroup —
  // 50–100 tiles on 256 CUs — and a tile's time barely depends on its width (the weight fragments stream through the
  // workgroup either way). Narrow the tile until the grid has ≈ 200 workgroups (measured, GOKU training step at B = 64:
  // 1.70 → 1.32 ms and 2.17 → 1.84 ms in two back-to-back pairs; at B = 256 the grids are full and nothing changes).
  auto narrow = [&](const ChainDims& q, int cg0, size_t lds0) {
    int g = cg0;
    while (g > 1 && (N + 16 * g - 1) / (16 * g) < 192) g /= 2;
    return ChainPick{&q, g, g == cg0 ? lds0 : chain_lds(q, g, bwd ? 3 : 2)};
  };
  if (cgx && N >= 16 * cgx && (((uintptr_t)x) & 15) == 0) {
    *p = narrow(c->cdx, cgx, bwd ? c->ldsx_bwd : c->ldsx_fwd);
    return true;
  }
  if (!cg) return false;
  *p = narrow(c->cd, cg, bwd ? c->lds_bwd : c->lds_fwd);
  return true;
}

// virtual tiling of the slot range for k_mlp_dw: (virtual tiles × jobs) ≈ one workgroup per CU — the kernel's register
// footprint allows one resident workgroup per CU, so 256 equal shares beat 384 (a second, half-empty round)
static void chain_dw_split(const lde_chain* c, int cg_bwd, int64_t N, int* nvt, int* cap, int64_t* total) {
  const int64_t tiles = (N + 16 * cg_bwd - 1) / (16 * cg_bwd);
  *total = tiles * cg_bwd;
  int v = 256 / dw_jobs(c->cd.dm, dw_pick_ndw(c->cd.dm));
  if (v < 1) v = 1;
  if (*total < v) v = (int)*total;
  *nvt = v;
  *cap = (int)((*total + v - 1) / v);
  if (*cap < 1) *cap = 1;
}

int lde_chain_reserve(lde_chain* c, int64_t N) {
  if (!c || !c->W_dev || N < 1) return LDE_ERR_INVALID_ARG;
  int nvt, cap;
  int64_t total;
  chain_dw_split(c, 2, N, &nvt, &cap, &total);   // 2 column groups per tile rounds the slot count up the most
  const MlpDims& dm = c->cd.dm;
  if (!grow(&c->stage, &c->stage_cap, (size_t)total * dm.blk_floats) || !grow(&c->wts, &c->wts_cap, (size_t)total * NB) ||
      !grow(&c->slab, &c->slab_cap, ((size_t)nvt + 1) * dm.slab_n)) {
    c->err = "chain: hipMalloc of the backward workspace failed";
    return LDE_ERR_ALLOC;
  }
  if (!c->ints) {   // two words the slab reduction reads as "no private slabs" / writes its feedback to: zero, once
    if (hipMalloc(&c->ints, 64) != hipSuccess || hipMemset(c->ints, 0, 64) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
      c->err = "chain: hipMalloc of the backward workspace failed";
      return LDE_ERR_ALLOC;
    }
    c->ints_cap = 16;
  }
  return LDE_OK;
}

// ---- the native bf16 path (lde_chain_bf16.h) ------------------------------------------------------------------------------------
struct ChainPickB { const ChainDims* cd; const BfDims* bd; int cg; size_t lds; };
static bool chain_pick_b(const lde_chain* c, const float* x, int64_t N, bool bwd, ChainPickB* p) {
  const int cgx = bwd ? c->bcgx_bwd : c->bcgx_fwd, cg = bwd ? c->bcg_bwd : c->bcg_fwd;
  auto narrow = [&](const ChainDims& q, const BfDims& b, int cg0) {
    int g = cg0;
    while (g > 1 && (N + 16 * g - 1) / (16 * g) < 192) g /= 2;
    return ChainPickB{&q, &b, g, chain_lds_b(q, b, g, bwd)};
  };
  if (c->cdx.gx && cgx && N >= 16 * cgx && (((uintptr_t)x) & 15) == 0) {
    *p = narrow(c->cdx, c->bdx, cgx);
    return true;
  }
  if (!cg) return false;
  *p = narrow(c->cd, c->bd, cg);
  return true;
}

// ---- grouped calls (lde_chain_group_*): while a recorder is installed the launch sites below RECORD what they would launch; the group
// entry point then issues each stage once for all the modules (k_*_group, lde_mfma.h) when they ask for the same kernel instance, and
// one by one otherwise. Same arguments, same code per module: the results are those of the separate calls, bit for bit.
struct RecMain {   // stage 0: the forward kernel, or the pullback's first kernel
  int kind;        // 0 k_chain_forward<·,false>, 1 k_chain_forward_b, 2 k_chain_backward<·,false>, 3 k_chain_backward_b, 4: other (launched, not recorded)
  int cg;
  ChainDims cd;
  BfDims bd;
  ChainFwdArgs f32;
  ChainFwdArgsB fb;
  ChainBwdArgs b32;
  ChainBwdArgsB bb;
  unsigned grid;
  size_t lds;
};
struct RecDw {     // stage 1: the weight-gradient product
  int kind;        // 0 k_mlp_dw<ndw,false>, 1 k_chain_dw_b<ndw>
  int ndw;
  MlpDims dm;
  DwArgs d32;
  ChainDims cd;
  BfDims bd;
  DwArgsB db;
  int gx, gy, gz;  // f32: (tiles, ks, jobs); bf16: (parts, jobs, 1)
  size_t lds;
};
struct RecRed {    // stage 2: the fixed-order slab sums
  MlpDims dm;
  ReduceArgs a;
  unsigned grid;
};
struct GroupRec {
  int n = 0;
  bool main_set[GROUP_MAX] = {}, dw_set[GROUP_MAX] = {}, red_set[GROUP_MAX] = {};
  RecMain main[GROUP_MAX];
  RecDw dw[GROUP_MAX];
  RecRed red[GROUP_MAX];
};
struct FwdMse { const float* t; float* part; unsigned tiles; bool delta; float scale; };   // delta: also leave δ_L′ for the pullback (bf16 mode)
static thread_local FwdMse t_fwd_mse = {nullptr, nullptr, 0, false, 0.f};                // lde_chain_forward_save_mse (tiles: the forward launch's grid, set by the launch site)
struct MseSrc { const float* t; const float* g; float scale; };
static thread_local MseSrc t_mse = {nullptr, nullptr, 0.f};
static thread_local const float* t_delta_g = nullptr;   // lde_chain_backward_saved_delta: δ_L′ is staged; the cotangent g multiplies dx / dW at the end              // lde_chain_backward_saved_mse
static thread_local const float* t_dy_more[2] = {nullptr, nullptr};   // lde_chain_backward_saved_sum: further sources of the output gradient
static thread_local GroupRec* t_rec = nullptr;
// (kernel arguments: 4 KB on this runtime)
static_assert(sizeof(GroupTable<ChainBDims, ChainBwdArgsB>) <= 4096 && sizeof(GroupTable<ChainDims, ChainBwdArgs>) <= 4096 &&
              sizeof(GroupTable<ChainBDims, DwArgsB>) <= 4096 && sizeof(GroupTable<MlpDims, DwArgs, GROUP_MAX_DW>) <= 4096 &&
              sizeof(GroupTable<MlpDims, ReduceArgs, GROUP_MAX_DW>) <= 4096, "a group's argument table must fit the kernel-argument segment");

static bool set_max_lds_(const void* fn, bool* done) {
  if (*done) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) return false;
  *done = true;
  return true;
}
static void launch_main_single(const RecMain& r, hipStream_t stream) {
  ChainDims cdv = r.cd;
  BfDims bdv = r.bd;
  if (r.kind == 0) {
    ChainFwdArgs a = r.f32;
    void* argv[] = {(void*)&cdv, (void*)&a};
    const void* fn = r.cg == 4 ? (const void*)k_chain_forward<4, false> : r.cg == 2 ? (const void*)k_chain_forward<2, false> : (const void*)k_chain_forward<1, false>;
    (void)hipLaunchKernel(fn, dim3(r.grid), dim3(512), argv, r.lds, stream);
  } else if (r.kind == 1) {
    ChainFwdArgsB a = r.fb;
    void* argv[] = {(void*)&cdv, (void*)&bdv, (void*)&a};
    const void* fn = r.cg == 4 ? (const void*)k_chain_forward_b<4> : r.cg == 2 ? (const void*)k_chain_forward_b<2> : (const void*)k_chain_forward_b<1>;
    (void)hipLaunchKernel(fn, dim3(r.grid), dim3(512), argv, r.lds, stream);
  } else if (r.kind == 2) {
    ChainBwdArgs a = r.b32;
    void* argv[] = {(void*)&cdv, (void*)&a};
    const void* fn = r.cg == 2 ? (const void*)k_chain_backward<2, false> : (const void*)k_chain_backward<1, false>;
    (void)hipLaunchKernel(fn, dim3(r.grid), dim3(512), argv, r.lds, stream);
  } else {
    ChainBwdArgsB a = r.bb;
    void* argv[] = {(void*)&cdv, (void*)&bdv, (void*)&a};
    const void* fn = r.cg == 4 ? (const void*)k_chain_backward_b<4> : r.cg == 2 ? (const void*)k_chain_backward_b<2> : (const void*)k_chain_backward_b<1>;
    (void)hipLaunchKernel(fn, dim3(r.grid), dim3(512), argv, r.lds, stream);
  }
}
static void launch_dw_single(const RecDw& r, hipStream_t stream) {
  if (r.kind == 0) {
    static bool attr[3] = {false, false, false};
    (void)set_max_lds_((const void*)k_mlp_dw<1, false>, &attr[0]);
    (void)set_max_lds_((const void*)k_mlp_dw<2, false>, &attr[1]);
    (void)set_max_lds_((const void*)k_mlp_dw<4, false>, &attr[2]);
    const dim3 grid(r.gx, r.gy, r.gz);
    if (r.ndw == 1) hipLaunchKernelGGL((k_mlp_dw<1, false>), grid, dim3(512), r.lds, stream, r.dm, r.d32);
    else if (r.ndw == 2) hipLaunchKernelGGL((k_mlp_dw<2, false>), grid, dim3(512), r.lds, stream, r.dm, r.d32);
    else hipLaunchKernelGGL((k_mlp_dw<4, false>), grid, dim3(512), r.lds, stream, r.dm, r.d32);
  } else {
    const dim3 grid(r.gx, r.gy);
    if (r.ndw == 1) hipLaunchKernelGGL(k_chain_dw_b<1>, grid, dim3(512), r.lds, stream, r.cd, r.bd, r.db);
    else if (r.ndw == 2) hipLaunchKernelGGL(k_chain_dw_b<2>, grid, dim3(512), r.lds, stream, r.cd, r.bd, r.db);
    else hipLaunchKernelGGL(k_chain_dw_b<4>, grid, dim3(512), r.lds, stream, r.cd, r.bd, r.db);
  }
}
static void launch_red_single(const RecRed& r, hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_tiles, dim3(r.grid), dim3(256), 0, stream, r.a.priv, r.a.nflush, r.a.nwg, r.a.slab, r.a.nslab, r.dm, r.a.dW, r.a.feedback,
                     r.a.assign);
}
// the three stages of a recorded group: ONE launch per stage when every module asks for the same small-tile kernel instance
static int group_flush(GroupRec& g, hipStream_t stream) {
  const int n = g.n;
  {   // stage 0
    bool same = n >= 2;
    for (int j = 0; j < n; j++) same = same && g.main_set[j] && g.main[j].kind == g.main[0].kind && g.main[j].cg == 1;
    if (same) {
      size_t lds = 0;
      for (int j = 0; j < n; j++) lds = std::max(lds, g.main[j].lds);
      const int kind = g.main[0].kind;
      static bool attr[4] = {false, false, false, false};
      if (kind == 0 || kind == 2) {
        if (kind == 0) {
          GroupTable<ChainDims, ChainFwdArgs> t{};
          t.n = n;
          for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.main[j].grid; t.dims[j] = g.main[j].cd; t.args[j] = g.main[j].f32; }
          if (!set_max_lds_((const void*)k_chain_forward_group<1, false>, &attr[0])) return LDE_ERR_HIP;
          void* argv[] = {(void*)&t};
          (void)hipLaunchKernel((const void*)k_chain_forward_group<1, false>, dim3(t.start[n]), dim3(512), argv, lds, stream);
        } else {
          GroupTable<ChainDims, ChainBwdArgs> t{};
          t.n = n;
          for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.main[j].grid; t.dims[j] = g.main[j].cd; t.args[j] = g.main[j].b32; }
          if (!set_max_lds_((const void*)k_chain_backward_group<1, false>, &attr[2])) return LDE_ERR_HIP;
          void* argv[] = {(void*)&t};
          (void)hipLaunchKernel((const void*)k_chain_backward_group<1, false>, dim3(t.start[n]), dim3(512), argv, lds, stream);
        }
      } else if (kind == 1) {
        GroupTable<ChainBDims, ChainFwdArgsB> t{};
        t.n = n;
        for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.main[j].grid; t.dims[j].cd = g.main[j].cd; t.dims[j].bd = g.main[j].bd; t.args[j] = g.main[j].fb; }
        if (!set_max_lds_((const void*)k_chain_forward_b_group<1>, &attr[1])) return LDE_ERR_HIP;
        void* argv[] = {(void*)&t};
        (void)hipLaunchKernel((const void*)k_chain_forward_b_group<1>, dim3(t.start[n]), dim3(512), argv, lds, stream);
      } else {
        GroupTable<ChainBDims, ChainBwdArgsB> t{};
        t.n = n;
        for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.main[j].grid; t.dims[j].cd = g.main[j].cd; t.dims[j].bd = g.main[j].bd; t.args[j] = g.main[j].bb; }
        if (!set_max_lds_((const void*)k_chain_backward_b_group<1>, &attr[3])) return LDE_ERR_HIP;
        void* argv[] = {(void*)&t};
        (void)hipLaunchKernel((const void*)k_chain_backward_b_group<1>, dim3(t.start[n]), dim3(512), argv, lds, stream);
      }
    } else {
      for (int j = 0; j < n; j++)
        if (g.main_set[j]) launch_main_single(g.main[j], stream);
    }
  }
  {   // stage 1
    bool any = false, same = n >= 2;
    for (int j = 0; j < n; j++) { any = any || g.dw_set[j]; same = same && g.dw_set[j] && g.dw[j].kind == g.dw[0].kind && g.dw[j].ndw == 1; }
    if (any && same) {
      size_t lds = 0;
      for (int j = 0; j < n; j++) lds = std::max(lds, g.dw[j].lds);
      static bool attr[2] = {false, false};
      if (g.dw[0].kind == 0) {
        GroupTable<MlpDims, DwArgs, GROUP_MAX_DW> t{};
        t.n = n;
        for (int j = 0; j < n; j++) {
          t.start[j + 1] = t.start[j] + g.dw[j].gx * g.dw[j].gy * g.dw[j].gz;
          t.gx[j] = g.dw[j].gx; t.gy[j] = g.dw[j].gy; t.dims[j] = g.dw[j].dm; t.args[j] = g.dw[j].d32;
        }
        if (!set_max_lds_((const void*)k_mlp_dw_group<1, false>, &attr[0])) return LDE_ERR_HIP;
        void* argv[] = {(void*)&t};
        (void)hipLaunchKernel((const void*)k_mlp_dw_group<1, false>, dim3(t.start[n]), dim3(512), argv, lds, stream);
      } else {
        GroupTable<ChainBDims, DwArgsB> t{};
        t.n = n;
        for (int j = 0; j < n; j++) {
          t.start[j + 1] = t.start[j] + g.dw[j].gx * g.dw[j].gy;
          t.gx[j] = g.dw[j].gx; t.gy[j] = g.dw[j].gy; t.dims[j].cd = g.dw[j].cd; t.dims[j].bd = g.dw[j].bd; t.args[j] = g.dw[j].db;
        }
        if (!set_max_lds_((const void*)k_chain_dw_b_group<1>, &attr[1])) return LDE_ERR_HIP;
        void* argv[] = {(void*)&t};
        (void)hipLaunchKernel((const void*)k_chain_dw_b_group<1>, dim3(t.start[n]), dim3(512), argv, lds, stream);
      }
    } else if (any) {
      for (int j = 0; j < n; j++)
        if (g.dw_set[j]) launch_dw_single(g.dw[j], stream);
    }
  }
  {   // stage 2
    bool any = false, all = n >= 2;
    for (int j = 0; j < n; j++) { any = any || g.red_set[j]; all = all && g.red_set[j]; }
    if (any && all) {
      GroupTable<MlpDims, ReduceArgs, GROUP_MAX_DW> t{};
      t.n = n;
      for (int j = 0; j < n; j++) { t.start[j + 1] = t.start[j] + (int)g.red[j].grid; t.dims[j] = g.red[j].dm; t.args[j] = g.red[j].a; }
      void* argv[] = {(void*)&t};
      (void)hipLaunchKernel((const void*)k_reduce_tiles_group, dim3(t.start[n]), dim3(256), argv, 0, stream);
    } else if (any) {
      for (int j = 0; j < n; j++)
        if (g.red_set[j]) launch_red_single(g.red[j], stream);
    }
  }
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

static int chain_forward_b(lde_chain* c, const float* x, int64_t N, float* y, __bf16* saved, hipStream_t stream) {
  ChainPickB pk;
  if (!chain_pick_b(c, x, N, false, &pk)) {
    c->err = "lde_chain_forward (bf16): no tile layout fits LDS for this input";
    return LDE_ERR_UNSUPPORTED;
  }
  ChainFwdArgsB a{x, y, c->fragb, c->W_dev, (long long)N, saved, t_fwd_mse.t, t_fwd_mse.part, nullptr, 0.f, 0};
  if (t_fwd_mse.delta) {   // δ_L′ into the last layer's δ matrix of the pullback's workspace (the allocation the pullback would make: it finds it there)
    const MlpDims& dmf = c->cd.dm;
    if (!grow(&c->dstage, &c->dstage_cap, (size_t)N * c->bd.dl_total + (size_t)64 * c->bd.dl_w[dmf.nL - 1] + 64)) {
      c->err = "chain: hipMalloc of the bf16 backward workspace failed";
      return LDE_ERR_ALLOC;
    }
    ChainPickB pkb;   // (the δ matrices' layout is the pullback's choice)
    if (!chain_pick_b(c, x, N, true, &pkb)) {
      c->err = "lde_chain_forward_save_mse_delta: no pullback tile layout fits LDS for this input";
      return LDE_ERR_UNSUPPORTED;
    }
    a.dL = c->dstage + (size_t)N * pkb.bd->dl_off[dmf.nL - 1];
    a.dk2 = 2.0f * t_fwd_mse.scale;
    a.dlw = pkb.bd->dl_w[dmf.nL - 1];
  }
  const int NC = 16 * pk.cg;
  const dim3 grid((unsigned)((N + NC - 1) / NC));
  t_fwd_mse.tiles = grid.x;
  static bool attr[5] = {false, false, false, false, false};
  const void* fn = pk.cg == 4 ? (const void*)k_chain_forward_b<4> : pk.cg == 2 ? (const void*)k_chain_forward_b<2> : (const void*)k_chain_forward_b<1>;
  if (!attr[pk.cg]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      c->err = "hipFuncSetAttribute(k_chain_forward_b) failed";
      return LDE_ERR_HIP;
    }
    attr[pk.cg] = true;
  }
  ChainDims cdv = *pk.cd;
  BfDims bdv = *pk.bd;
  void* argv[] = {(void*)&cdv, (void*)&bdv, (void*)&a};
#if LDE_PROF
  { long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
#endif
  if (t_rec) {
    RecMain& r = t_rec->main[t_rec->n];
    r.kind = 1; r.cg = pk.cg; r.cd = cdv; r.bd = bdv; r.fb = a; r.grid = grid.x; r.lds = pk.lds;
    t_rec->main_set[t_rec->n] = true;
    (void)argv;
    return LDE_OK;
  }
  (void)hipLaunchKernel(fn, grid, dim3(512), argv, pk.lds, stream);
  if (hipGetLastError() != hipSuccess) {
    c->err = "k_chain_forward_b launch failed";
    return LDE_ERR_HIP;
  }
#if LDE_PROF
  {
    static int calls = 0;
    (void)hipStreamSynchronize(stream);
    long long v[64];
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
    if (N > 1000 && ++calls % 20 == 0) {
      fprintf(stderr, "[prof chain fwd bf16 cg=%d] cycles:", pk.cg);
      for (int i = 0; i < 64; i++)
        if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return LDE_OK;
}

static int chain_backward_b(lde_chain* c, const float* x, const float* y, const float* dy, const __bf16* saved, int64_t N, float* dx,
                            float* dW, hipStream_t stream) {
  const MlpDims& dm = c->cd.dm;
  const int ndw = dw_pick_ndw(dm), jobs = dw_jobs(dm, ndw);
  const int64_t nchunks = (N + DWB_NK - 1) / DWB_NK;
  int parts = 256 / jobs;
  parts = parts < 1 ? 1 : parts;
  if (parts > nchunks) parts = (int)nchunks;
  // (+ one tile of rows of the last layer's matrix: its read-back as a B operand covers the ragged tile's columns beyond N — their
  //  results are dropped, but the loads must stay inside the allocation)
  if (!grow(&c->dstage, &c->dstage_cap, (size_t)N * c->bd.dl_total + (size_t)64 * c->bd.dl_w[dm.nL - 1] + 64) || !grow(&c->slab, &c->slab_cap, ((size_t)parts + 1) * dm.slab_n)) {
    c->err = "chain: hipMalloc of the bf16 backward workspace failed";
    return LDE_ERR_ALLOC;
  }
  if (!c->ints) {
    if (hipMalloc(&c->ints, 64) != hipSuccess || hipMemset(c->ints, 0, 64) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
      c->err = "chain: hipMalloc of the backward workspace failed";
      return LDE_ERR_ALLOC;
    }
    c->ints_cap = 16;
  }
  if (!saved) {   // the caller kept nothing: the pullback's own forward pass fills a scratch copy of the saved matrices
    const size_t need = (size_t)c->cd.sv_total * N + 8;
    if (!grow(&c->svscratch, &c->svscratch_cap, need)) {
      c->err = "chain: hipMalloc of the saved-activation scratch failed";
      return LDE_ERR_ALLOC;
    }
    const int rcf = chain_forward_b(c, x, N, nullptr, c->svscratch, stream);
    if (rcf) return rcf;
    saved = c->svscratch;
  }
  ChainPickB pk;
  if (!chain_pick_b(c, x, N, true, &pk)) {
    c->err = "lde_chain_backward (bf16): no tile layout fits LDS for this input";
    return LDE_ERR_UNSUPPORTED;
  }
  ChainBwdArgsB a{x, y, dy, dx, c->fragTb, c->W_dev, c->dstage, (long long)N, saved, t_dy_more[0], t_dy_more[1], t_mse.t, t_mse.g, t_mse.scale,
                  t_delta_g ? 1 : 0, t_delta_g};
  if (!t_delta_g) { c->delta_N = -1; c->delta_saved = nullptr; }   // (this pullback writes its own δ_L over whatever the forward pass staged)
  const int NC = 16 * pk.cg;
  {
    const dim3 grid((unsigned)((N + NC - 1) / NC));
    static bool attr[5] = {false, false, false, false, false};
    const void* fn = pk.cg == 4 ? (const void*)k_chain_backward_b<4> : pk.cg == 2 ? (const void*)k_chain_backward_b<2> : (const void*)k_chain_backward_b<1>;
    if (!attr[pk.cg]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
        c->err = "hipFuncSetAttribute(k_chain_backward_b) failed";
        return LDE_ERR_HIP;
      }
      attr[pk.cg] = true;
    }
    ChainDims cdv = *pk.cd;
    BfDims bdv = *pk.bd;
    void* argv[] = {(void*)&cdv, (void*)&bdv, (void*)&a};
    if (t_rec) {
      RecMain& r = t_rec->main[t_rec->n];
      r.kind = 3; r.cg = pk.cg; r.cd = cdv; r.bd = bdv; r.bb = a; r.grid = grid.x; r.lds = pk.lds;
      t_rec->main_set[t_rec->n] = true;
    } else {
      (void)hipLaunchKernel(fn, grid, dim3(512), argv, pk.lds, stream);
      if (hipGetLastError() != hipSuccess) {
        c->err = "k_chain_backward_b launch failed";
        return LDE_ERR_HIP;
      }
    }
  }
  // weight gradient: [n][feature] matrices through the transposing LDS reads, then the fixed-order slab reduction
  bool sw_ok = true;
  hipStream_t wst = (t_rec || !c->opt_async_dw) ? stream : dw_sync_switch(c->dws, stream, &sw_ok);   // (recorded for a grouped launch: on the caller's stream)
  if (!sw_ok) {
    c->err = "lde_chain_backward: switching to the weight-gradient stream failed";
    return LDE_ERR_HIP;
  }
  {
    const size_t dlds = dw_b_lds_bytes(dm, ndw);
    if (dlds > LDS_MAX) {
      c->err = "layer too wide for the bf16 weight-gradient kernel's LDS images";
      return LDE_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)k_chain_dw_b<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_chain_dw_b<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_chain_dw_b<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
        c->err = "hipFuncSetAttribute(k_chain_dw_b) failed";
        return LDE_ERR_HIP;
      }
      attr_set = true;
    }
    DwArgsB da{x, saved, c->dstage, c->slab, (long long)N, t_delta_g};
    const dim3 grid(parts, jobs);
#if LDE_PROF
    { (void)hipStreamSynchronize(wst); long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
#endif
    ChainDims cdv = c->cd;
    BfDims bdv = c->bd;
    if (t_rec) {
      RecDw& r = t_rec->dw[t_rec->n];
      r.kind = 1; r.ndw = ndw; r.cd = cdv; r.bd = bdv; r.db = da; r.gx = parts; r.gy = jobs; r.gz = 1; r.lds = dlds;
      t_rec->dw_set[t_rec->n] = true;
      RecRed& q = t_rec->red[t_rec->n];
      q.dm = dm; q.a = ReduceArgs{nullptr, c->ints, 0, c->slab, parts, dW, c->ints + 2, c->accumulate ? 0 : 1}; q.grid = (unsigned)cdiv(dm.slab_n, 1024);
      t_rec->red_set[t_rec->n] = true;
      return LDE_OK;
    }
    if (ndw == 1) hipLaunchKernelGGL(k_chain_dw_b<1>, grid, dim3(512), dlds, wst, cdv, bdv, da);
    else if (ndw == 2) hipLaunchKernelGGL(k_chain_dw_b<2>, grid, dim3(512), dlds, wst, cdv, bdv, da);
    else hipLaunchKernelGGL(k_chain_dw_b<4>, grid, dim3(512), dlds, wst, cdv, bdv, da);
    hipLaunchKernelGGL(k_reduce_tiles, dim3(cdiv(dm.slab_n, 1024)), dim3(256), 0, wst, (const float*)nullptr, c->ints, 0, c->slab, parts, dm, dW,
                       c->ints + 2, c->accumulate ? 0 : 1);
    if (hipGetLastError() != hipSuccess) {
      c->err = "bf16 weight-gradient kernels failed to launch";
      return LDE_ERR_HIP;
    }
#if LDE_PROF
    {
      static int calls = 0;
      (void)hipStreamSynchronize(wst);
      long long v[64];
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
      if (N > 1000 && ++calls % 20 == 0) {
        fprintf(stderr, "[prof chain dw bf16 parts=%d jobs=%d ndw=%d] cycles:", parts, jobs, ndw);
        for (int i = 0; i < 64; i++)
          if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
        fprintf(stderr, "\n");
      }
    }
#endif
  }
  if (!dw_sync_end(c->dws, wst, stream)) {
    c->err = "lde_chain_backward: hipEventRecord failed";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

static int chain_forward_impl(lde_chain* c, const float* x, int64_t N, float* y, float* saved, void* stream_) {
  if (!c || !c->W_dev) return LDE_ERR_INVALID_ARG;
  if (!x || (!y && !t_fwd_mse.delta) || N < 1) {   // (y may be NULL only where the forward pass leaves δ_L′ instead: lde_chain_forward_save_mse_delta)
    c->err = "lde_chain_forward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (!c->have_W) {
    c->err = "lde_chain_forward: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  if (((uintptr_t)y & 15) != 0) {
    c->err = "lde_chain_forward: y must be 16-byte aligned";
    return LDE_ERR_INVALID_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  if (c->bf16) return chain_forward_b(c, x, N, y, reinterpret_cast<__bf16*>(saved), stream);
  ChainFwdArgs a{x, y, c->frag, c->W_dev, (long long)N, saved, t_fwd_mse.t, t_fwd_mse.part};
  ChainPick pk;
  if (!chain_pick(c, x, N, false, &pk)) {
    c->err = "lde_chain_forward: the only layout whose panels fit LDS reads x in place and needs N ≥ one tile and a 16-byte aligned x";
    return LDE_ERR_UNSUPPORTED;
  }
  const int NC = 16 * pk.cg;
  const dim3 grid((unsigned)((N + NC - 1) / NC));
  t_fwd_mse.tiles = grid.x;
  static bool attr[2][5] = {{false, false, false, false, false}, {false, false, false, false, false}};
  const int bf = c->bf16 ? 1 : 0;
  const void* fn = bf ? (pk.cg == 4 ? (const void*)k_chain_forward<4, true> : pk.cg == 2 ? (const void*)k_chain_forward<2, true> : (const void*)k_chain_forward<1, true>)
                      : (pk.cg == 4 ? (const void*)k_chain_forward<4, false> : pk.cg == 2 ? (const void*)k_chain_forward<2, false> : (const void*)k_chain_forward<1, false>);
  if (!attr[bf][pk.cg]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      c->err = "hipFuncSetAttribute(k_chain_forward) failed";
      return LDE_ERR_HIP;
    }
    attr[bf][pk.cg] = true;
  }
#if LDE_PROF
  { long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
#endif
  if (t_rec && !bf) {
    RecMain& r = t_rec->main[t_rec->n];
    r.kind = 0; r.cg = pk.cg; r.cd = *pk.cd; r.f32 = a; r.grid = grid.x; r.lds = pk.lds;
    t_rec->main_set[t_rec->n] = true;
    return LDE_OK;
  }
  {
    ChainDims cdv = *pk.cd;
    void* argv[] = {(void*)&cdv, (void*)&a};
    (void)hipLaunchKernel(fn, grid, dim3(512), argv, pk.lds, stream);
  }
  if (hipGetLastError() != hipSuccess) {
    c->err = "k_chain_forward launch failed";
    return LDE_ERR_HIP;
  }
#if LDE_PROF
  {
    static int calls = 0;
    (void)hipStreamSynchronize(stream);
    long long v[64];
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
    if (N > 1000 && ++calls % 20 == 0) {
      fprintf(stderr, "[prof chain fwd] cycles:");
      for (int i = 0; i < 64; i++)
        if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return LDE_OK;
}

static int chain_backward_impl(lde_chain* c, const float* x, const float* y, const float* dy, const float* saved, int64_t N,
                               float* dx, float* dW, void* stream_) {
  if (!c || !c->W_dev) return LDE_ERR_INVALID_ARG;
  if (!x || !y || !dy || !dW || N < 1) {
    c->err = "lde_chain_backward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (!c->have_W) {
    c->err = "lde_chain_backward: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  if ((((uintptr_t)y | (uintptr_t)dy) & 15) != 0) {
    c->err = "lde_chain_backward: y and dy must be 16-byte aligned";
    return LDE_ERR_INVALID_ARG;
  }
  if (!dw_sync_begin(c->dws, (hipStream_t)stream_)) {   // the workspace is about to be rewritten
    c->err = "lde_chain_backward: waiting for the previous weight gradient failed";
    return LDE_ERR_HIP;
  }
  if (c->bf16) return chain_backward_b(c, x, y, dy, reinterpret_cast<const __bf16*>(saved), N, dx, dW, (hipStream_t)stream_);
  int rc = lde_chain_reserve(c, N);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const MlpDims& dm = c->cd.dm;
  ChainPick pk;
  if (!chain_pick(c, x, N, true, &pk)) {
    c->err = "lde_chain_backward: the only layout whose panels fit LDS reads x in place and needs N ≥ one tile and a 16-byte aligned x";
    return LDE_ERR_UNSUPPORTED;
  }
  int nvt, cap;
  int64_t total;
  chain_dw_split(c, pk.cg, N, &nvt, &cap, &total);
  ChainBwdArgs a{x, y, dy, dx, c->frag, c->fragT, c->W_dev, c->stage, c->wts, (long long)N, saved, t_dy_more[0], t_dy_more[1], t_mse.t, t_mse.g, t_mse.scale};
  const int NC = 16 * pk.cg;
  const dim3 grid((unsigned)((N + NC - 1) / NC));
  static bool attr[2][3] = {{false, false, false}, {false, false, false}};
  const int bf = c->bf16 ? 1 : 0;
  const void* fn = bf ? (pk.cg == 2 ? (const void*)k_chain_backward<2, true> : (const void*)k_chain_backward<1, true>)
                      : (pk.cg == 2 ? (const void*)k_chain_backward<2, false> : (const void*)k_chain_backward<1, false>);
  if (!attr[bf][pk.cg]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      c->err = "hipFuncSetAttribute(k_chain_backward) failed";
      return LDE_ERR_HIP;
    }
    attr[bf][pk.cg] = true;
  }
  const bool rec32 = t_rec && !bf;
  if (rec32) {
    RecMain& r = t_rec->main[t_rec->n];
    r.kind = 2; r.cg = pk.cg; r.cd = *pk.cd; r.b32 = a; r.grid = grid.x; r.lds = pk.lds;
    t_rec->main_set[t_rec->n] = true;
  } else {
    ChainDims cdv = *pk.cd;
    void* argv[] = {(void*)&cdv, (void*)&a};
    (void)hipLaunchKernel(fn, grid, dim3(512), argv, pk.lds, stream);
  }
  if (hipGetLastError() != hipSuccess) {
    c->err = "k_chain_backward launch failed";
    return LDE_ERR_HIP;
  }
  // weight gradient: large-K product over the staged panels (lde_mfma.h)
  DwArgs da;
  da.stage = c->stage; da.wts = c->wts; da.nslots = nullptr; da.slab = c->slab; da.cap = cap; da.total = total;   // tiles filled in order
#if LDE_PROF
  { (void)hipStreamSynchronize(stream); long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
#endif
  bool sw_ok = true;
  hipStream_t wst = (t_rec || !c->opt_async_dw) ? stream : dw_sync_switch(c->dws, stream, &sw_ok);   // (recorded for a grouped launch: on the caller's stream)
  if (!sw_ok) {
    c->err = "lde_chain_backward: switching to the weight-gradient stream failed";
    return LDE_ERR_HIP;
  }
  if (rec32) {
    const int ndw = dw_pick_ndw(dm);
    RecDw& r = t_rec->dw[t_rec->n];
    r.kind = 0; r.ndw = ndw; r.dm = dm; r.d32 = da; r.gx = nvt; r.gy = 1; r.gz = dw_jobs(dm, ndw); r.lds = dw_lds_floats(dm, ndw) * sizeof(float);
    t_rec->dw_set[t_rec->n] = true;
    RecRed& q = t_rec->red[t_rec->n];
    q.dm = dm; q.a = ReduceArgs{nullptr, c->ints, 0, da.slab, nvt, dW, c->ints + 2, c->accumulate ? 0 : 1}; q.grid = (unsigned)cdiv(dm.slab_n, 1024);
    t_rec->red_set[t_rec->n] = true;
    return LDE_OK;
  }
  rc = launch_weight_gradient(dm, da, nvt, 1, nullptr, c->ints, 0, dW, c->ints + 2, wst, c->err, !c->accumulate, c->bf16);
  if (rc == LDE_OK && !dw_sync_end(c->dws, wst, stream)) {
    c->err = "lde_chain_backward: hipEventRecord failed";
    return LDE_ERR_HIP;
  }
  if (rc) return rc;
#if LDE_PROF
  {
    static int calls = 0;
    (void)hipStreamSynchronize(stream);
    long long v[64];
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prof), sizeof(v));
    if (N > 1000 && ++calls % 20 == 0) {
      fprintf(stderr, "[prof chain dw] nvt=%d cap=%d jobs=%d ndw=%d cycles:", nvt, cap, dw_jobs(dm, dw_pick_ndw(dm)), dw_pick_ndw(dm));
      for (int i = 0; i < 64; i++)
        if (v[i]) fprintf(stderr, " %d:%lld", i, v[i]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return LDE_OK;
}

int lde_chain_set_dtype(lde_chain* c, int dtype) {
  if (!c || (dtype != LDE_DTYPE_F32 && dtype != LDE_DTYPE_BF16)) return LDE_ERR_INVALID_ARG;
  c->bf16 = dtype == LDE_DTYPE_BF16;
  if (!c->bf16 && c->frag_stale && c->have_W) {   // rare: back to f32 after hand-overs that skipped the f32 fragments
    const int rc = chain_frags(c, c->W_dev, nullptr);
    if (rc || hipStreamSynchronize(nullptr) != hipSuccess) return rc ? rc : LDE_ERR_HIP;
  }
  c->frag_stale = false;
  return LDE_OK;
}
int lde_chain_forward(lde_chain* c, const float* x, int64_t N, float* y, void* stream) {
  return chain_forward_impl(c, x, N, y, nullptr, stream);
}
int lde_chain_backward(lde_chain* c, const float* x, const float* y, const float* dy, int64_t N, float* dx, float* dW, void* stream) {
  return chain_backward_impl(c, x, y, dy, nullptr, N, dx, dW, stream);
}
int64_t lde_chain_saved_floats(const lde_chain* c, int64_t N) {
  if (!c || N < 1) return -1;
  const int64_t n = (int64_t)c->cd.sv_total * N;
  return n > 0 ? n : 4;   // a chain without hidden layers saves nothing; keep the buffer non-empty
}
int lde_chain_forward_save(lde_chain* c, const float* x, int64_t N, float* y, float* saved, void* stream) {
  if (c && (!saved || (((uintptr_t)saved) & 15) != 0)) {
    c->err = "lde_chain_forward_save: saved must be a 16-byte aligned device buffer of lde_chain_saved_floats(c, N) floats";
    return LDE_ERR_INVALID_ARG;
  }
  return chain_forward_impl(c, x, N, y, saved, stream);
}
int lde_chain_backward_saved(lde_chain* c, const float* x, const float* y, const float* dy, const float* saved, int64_t N,
                             float* dx, float* dW, void* stream) {
  if (c && (!saved || (((uintptr_t)saved) & 15) != 0)) {
    c->err = "lde_chain_backward_saved: saved must be the buffer lde_chain_forward_save filled";
    return LDE_ERR_INVALID_ARG;
  }
  return chain_backward_impl(c, x, y, dy, saved, N, dx, dW, stream);
}

int64_t lde_chain_mse_scratch_floats(const lde_chain* c, int64_t N) { return (!c || N < 1) ? -1 : (N + 15) / 16 + 1; }
int lde_chain_forward_save_mse(lde_chain* c, const float* x, int64_t N, float* y, float* saved, const float* target, float scale, const float* base,
                               float* out, float* scratch, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (!target || !out || !scratch || (((uintptr_t)target) & 15) != 0) {
    c->err = "lde_chain_forward_save_mse: NULL pointer or unaligned target";
    return LDE_ERR_INVALID_ARG;
  }
  if (t_rec) {
    c->err = "lde_chain_forward_save_mse: not inside a grouped call";
    return LDE_ERR_UNSUPPORTED;
  }
  t_fwd_mse = FwdMse{target, scratch, 0};
  const int rc = saved ? lde_chain_forward_save(c, x, N, y, saved, stream) : lde_chain_forward(c, x, N, y, stream);
  const unsigned tiles = t_fwd_mse.tiles;
  t_fwd_mse = FwdMse{nullptr, nullptr, 0};
  if (rc) return rc;
  return loss_finalize(scratch, (int)tiles, scale, base, out, (hipStream_t)stream);   // out = base + scale·Σ partials, in tile order
}
int lde_chain_backward_saved_sum(lde_chain* c, const float* x, const float* y, int n_dy, const float* const* dys, const float* saved, int64_t N,
                                 float* dx, float* dW, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (n_dy < 1 || n_dy > 3 || !dys) {
    c->err = "lde_chain_backward_saved_sum: 1 to 3 output-gradient arrays";
    return LDE_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_dy; i++)
    if (!dys[i] || (((uintptr_t)dys[i]) & 15) != 0) {
      c->err = "lde_chain_backward_saved_sum: NULL or unaligned output-gradient array";
      return LDE_ERR_INVALID_ARG;
    }
  t_dy_more[0] = n_dy > 1 ? dys[1] : nullptr;
  t_dy_more[1] = n_dy > 2 ? dys[2] : nullptr;
  const int rc = saved ? lde_chain_backward_saved(c, x, y, dys[0], saved, N, dx, dW, stream) : lde_chain_backward(c, x, y, dys[0], N, dx, dW, stream);
  t_dy_more[0] = t_dy_more[1] = nullptr;
  return rc;
}

int lde_chain_backward_saved_mse(lde_chain* c, const float* x, const float* y, const float* target, const float* g_dev, float scale,
                                 const float* dy_more, const float* saved, int64_t N, float* dx, float* dW, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (!target || !g_dev || (((uintptr_t)target) & 15) != 0 || (dy_more && (((uintptr_t)dy_more) & 15) != 0)) {
    c->err = "lde_chain_backward_saved_mse: NULL or unaligned target / cotangent";
    return LDE_ERR_INVALID_ARG;
  }
  t_mse = MseSrc{target, g_dev, scale};
  t_dy_more[0] = dy_more;
  const int rc = saved ? lde_chain_backward_saved(c, x, y, target, saved, N, dx, dW, stream) : lde_chain_backward(c, x, y, target, N, dx, dW, stream);
  t_mse = MseSrc{nullptr, nullptr, 0.f};
  t_dy_more[0] = nullptr;
  return rc;
}

// The reconstructor under the loss, bf16 mode, without the x̂ round trip: the forward launch's last epilogue — where x̂ and the target are in
// registers for the squares anyway — also leaves δ_L′ = 2·scale·(x̂ − target)·act′(x̂) as the pullback's bf16 δ matrix, and stores x̂ only when
// the caller wants it (y may be NULL); the pullback then starts from that matrix (no pass over x̂ / target, 120 MB of a GOKU step's HBM traffic)
// and multiplies dx and dW by the loss's cotangent g at the end (δ is linear in g; with g = 1 — the loss IS the objective — the bits are
// those of lde_chain_backward_saved_mse). LDE_ERR_UNSUPPORTED outside the bf16 mode or for an output width that is not a multiple of 8.
int lde_chain_forward_save_mse_delta(lde_chain* c, const float* x, int64_t N, float* y, float* saved, const float* target, float scale, const float* base,
                                     float* out, float* scratch, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (!target || !out || !scratch || !saved || (((uintptr_t)target) & 15) != 0) {
    c->err = "lde_chain_forward_save_mse_delta: NULL pointer or unaligned target";
    return LDE_ERR_INVALID_ARG;
  }
  if (!c->bf16 || (c->cd.dm.sizes[c->cd.dm.nL] & 7) != 0 || t_rec) {
    c->err = "lde_chain_forward_save_mse_delta: bf16 mode, an output width that is a multiple of 8, not inside a grouped call";
    return LDE_ERR_UNSUPPORTED;
  }
  t_fwd_mse = FwdMse{target, scratch, 0, true, scale};
  const int rc = lde_chain_forward_save(c, x, N, y, saved, stream);
  const unsigned tiles = t_fwd_mse.tiles;
  t_fwd_mse = FwdMse{nullptr, nullptr, 0, false, 0.f};
  if (rc) return rc;
  c->delta_N = N;
  c->delta_saved = saved;
  return loss_finalize(scratch, (int)tiles, scale, base, out, (hipStream_t)stream);
}
int lde_chain_delta_is_staged(const lde_chain* c, const float* saved, int64_t N) {
  return c && c->bf16 && c->dstage && c->delta_N == N && N >= 0 && c->delta_saved == saved && saved != nullptr;
}
int lde_chain_backward_saved_delta(lde_chain* c, const float* x, const float* g_dev, const float* saved, int64_t N, float* dx, float* dW, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (!g_dev || !saved) {
    c->err = "lde_chain_backward_saved_delta: NULL cotangent / saved activations";
    return LDE_ERR_INVALID_ARG;
  }
  if (!lde_chain_delta_is_staged(c, saved, N)) {
    // (another forward of this chain has re-staged δ_L′ since — or none ran: the δ in the workspace is not this call's; using it would give
    //  silently wrong gradients)
    c->err = "lde_chain_backward_saved_delta: the staged δ_L is not the one of the forward call these saved activations belong to "
             "(lde_chain_forward_save_mse_delta must be the chain's last such call; lde_chain_delta_is_staged tells)";
    return LDE_ERR_INVALID_ARG;
  }
  t_delta_g = g_dev;
  const int rc = lde_chain_backward_saved(c, x, x, x, saved, N, dx, dW, stream);   // (y and dy are not read: δ_L′ is staged)
  t_delta_g = nullptr;
  c->delta_N = -1;          // consumed: a second pullback from the same staging must not pass silently either
  c->delta_saved = nullptr;
  return rc;
}

// ---- grouped calls: n independent chains, each stage of the call ONE launch when the chains ask for the same small-tile kernel ----------
static bool group_ok(int n, lde_chain* const* cs) {
#if LDE_PROF
  return false;
#else
  for (int i = 0; i < n; i++)
    if (!cs[i] || !cs[i]->opt_group) return false;   // (lde_chain_set_option "group" = 0: the chains of the call run one after the other)
  return n >= 2 && n <= GROUP_MAX;   // (with a weight-gradient stream set, a group's jobs stay on the caller's stream: they are small; the stream is for the large single chains)
#endif
}
static bool group_distinct(int n, lde_chain* const* cs) {   // a handle's workspace serves one call at a time: the same chain twice runs one after the other
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (cs[i] == cs[j]) return false;
  return true;
}
int lde_chain_group_forward_save(int n, lde_chain* const* cs, const float* const* xs, const int64_t* Ns, float* const* ys, float* const* saveds,
                                 void* stream) {
  if (n < 1 || !cs || !xs || !Ns || !ys) return LDE_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++)
    if (!cs[i]) return LDE_ERR_INVALID_ARG;
  if (!group_ok(n, cs) || !group_distinct(n, cs)) {
    for (int i = 0; i < n; i++) {
      const int rc = chain_forward_impl(cs[i], xs[i], Ns[i], ys[i], saveds ? saveds[i] : nullptr, stream);
      if (rc) return rc;
    }
    return LDE_OK;
  }
  GroupRec g;
  t_rec = &g;
  for (int i = 0; i < n; i++) {
    g.n = i;
    const int rc = chain_forward_impl(cs[i], xs[i], Ns[i], ys[i], saveds ? saveds[i] : nullptr, stream);
    if (rc) { t_rec = nullptr; return rc; }
  }
  g.n = n;
  t_rec = nullptr;
  const int rc = group_flush(g, (hipStream_t)stream);
  if (rc) cs[0]->err = "lde_chain_group_forward_save: launch failed";
  return rc;
}
int lde_chain_group_backward_saved(int n, lde_chain* const* cs, const float* const* xs, const float* const* ys, const float* const* dys,
                                   const float* const* saveds, const int64_t* Ns, float* const* dxs, float* const* dWs, void* stream) {
  if (n < 1 || !cs || !xs || !ys || !dys || !Ns || !dWs) return LDE_ERR_INVALID_ARG;
  bool grp = group_ok(n, cs);
  for (int i = 0; i < n; i++) {
    if (!cs[i]) return LDE_ERR_INVALID_ARG;
    if (cs[i]->bf16 && !(saveds && saveds[i])) grp = false;   // (a bf16 pullback without saved activations starts with a forward launch of its own)
  }
  grp = grp && group_distinct(n, cs);
  if (!grp) {
    for (int i = 0; i < n; i++) {
      const int rc = chain_backward_impl(cs[i], xs[i], ys[i], dys[i], saveds ? saveds[i] : nullptr, Ns[i], dxs ? dxs[i] : nullptr, dWs[i], stream);
      if (rc) return rc;
    }
    return LDE_OK;
  }
  GroupRec g;
  t_rec = &g;
  for (int i = 0; i < n; i++) {
    g.n = i;
    const int rc = chain_backward_impl(cs[i], xs[i], ys[i], dys[i], saveds ? saveds[i] : nullptr, Ns[i], dxs ? dxs[i] : nullptr, dWs[i], stream);
    if (rc) { t_rec = nullptr; return rc; }
  }
  g.n = n;
  t_rec = nullptr;
  const int rc = group_flush(g, (hipStream_t)stream);
  if (rc) cs[0]->err = "lde_chain_group_backward_saved: launch failed";
  return rc;
}

int lde_chain_set_accumulate(lde_chain* c, int on) {
  if (!c) return LDE_ERR_INVALID_ARG;
  c->accumulate = on != 0;
  return LDE_OK;
}

int lde_chain_set_option(lde_chain* c, const char* key, double value) {
  if (!c || !key) return LDE_ERR_INVALID_ARG;
  if (!std::strcmp(key, "group")) {
    c->opt_group = value != 0;
    return LDE_OK;
  }
  if (!std::strcmp(key, "async_dw")) {   // 0: this chain's weight-gradient kernels stay on the caller's stream even when a dw stream is set
    c->opt_async_dw = value != 0;
    return LDE_OK;
  }
  if (!std::strcmp(key, "gx")) {   // 0: never the panel-free layout of a wide first layer (the two layouts are each other's parity reference)
    if (value != 0) return LDE_OK;
    if (!c->cg_fwd || !c->cg_bwd) {
      c->err = "lde_chain_set_option(gx = 0): only the panel-free layout fits this chain";
      return LDE_ERR_UNSUPPORTED;
    }
    c->cdx.gx = 0;
    c->cgx_fwd = c->cgx_bwd = c->bcgx_fwd = c->bcgx_bwd = 0;
    return LDE_OK;
  }
  c->err = std::string("lde_chain_set_option: unknown key: ") + key;
  return LDE_ERR_INVALID_ARG;
}

const char* lde_chain_last_error(const lde_chain* c) { return c ? c->err.c_str() : "null handle"; }

}  // extern "C"

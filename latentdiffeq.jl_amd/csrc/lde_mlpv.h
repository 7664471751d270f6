// lde_mlpv.h — MLP right-hand sides at SMALL batches: one trajectory per workgroup, lanes = hidden units (included by lde_mlp.hip).
//
// The 16-column MFMA tiles of lde_mlp.hip need B/16 ≥ 256 tiles to fill the chip; at the BASELINE batch sizes (c2: B = 256,
// c3: 1024, c4: 512 per GPU) only 16–64 CUs own a tile and the time of a step is ONE tile's latency (≈ 10–17 µs per RHS
// evaluation of the adjoint: LDS panel round trips, split-K reductions, barriers per layer). "MFMA only if latent_dim×hidden
// is large enough to fill a tile": below that, the right mapping is the other one — a workgroup owns ONE trajectory, lane i
// owns hidden unit i, and a Dense layer is a matrix–vector product on the vector ALU:
//     y_i = Σ_k W[i,k] x_k :  lane i walks k four at a time — one 16-byte load of W[i, k..k+3] (weights pre-swizzled so
//     that consecutive lanes read consecutive 16-byte words: coalesced from L2, conflict-free from LDS) and one 16-byte
//     LDS broadcast read of x_k..k+3 per 4 FMAs.
// The transposed products of the pullback (δ_in = Wᵀδ) use a second swizzled copy with the roles of rows and columns
// exchanged, so both directions have lanes = output units. Narrow layers (out ≤ NT/2) split K over S ≤ 8 lane groups and
// add the partial sums through LDS. Swizzled weights are cached in LDS as far as the budget allows (the rest streams from
// L2 — every workgroup reads the same few hundred KB). State vectors [z; λ; g] are a few dozen floats in LDS; step control
// is one lane. B trajectories = B workgroups: 256–1024 of them put every CU to work, and the time of a step is one
// trajectory's matrix–vector chain (≈ 1–3 µs per evaluation).
// The weight gradient is staged exactly as the tile kernels stage it — trajectory b writes column b mod 16 of tile b/16's
// slots, its quadrature weights at accept time — and formed by k_mlp_dw afterwards (lde_mfma.h).
// Same algorithms, same control arithmetic as k_mlp_forward / k_mlp_adjoint; agreement to solver tolerance, not bitwise.
// Limits: layer widths and D' ≤ 256; otherwise (and for large batches) the tile kernels run.

struct VecDims {
  int NT;                                                      // threads per workgroup: 64, 128 or 256
  int rpf[MAXL], lgf[MAXL], k4f[MAXL], sf[MAXL], off_f[MAXL];  // W·x of layer l: rows = out padded to rpf, K-groups (of 4) PER LANE (multiple of 4; the array holds S·k4 groups, zero-padded), split, float4 offset
  int rpb[MAXL], lgb[MAXL], k4b[MAXL], sb[MAXL], off_b[MAXL];  // Wᵀ·δ of layer l: rows = in, K = out
  int total4;                                                  // float4 entries of the swizzled array
  int hoff[MAXL], htotal;                                      // hidden vectors (post-activation), each padded to a multiple of 4
  int maxw4;                                                   // floats of one δ buffer (widest vector / widest padded K)
  int nsp_f, nsp_b;                                            // floats of one state vector in the forward / adjoint kernel (incl. the zero tail a product may read)
  int reg_l;                                                   // layer whose two products keep their weights in REGISTERS (−1: none), VREG_K groups per lane
};
constexpr int VREG_K = 16;   // K-groups (of 4) per lane of a register-resident product: 64 VGPRs per direction

struct VArgs {
  const float* z0;        // forward: [D×B]
  const float* theta;
  const double* ts;
  const float* vecw;      // swizzled weights
  const float* wpack;     // k_mlpw's packed weights (lde_mlpw.h)
  unsigned epoch;         // k_mlpw: launch epoch (16 bits) of the tagged grid-sum words
  int cot_lds;            // k_mlpw adjoint: the trajectory's cotangents fit LDS
  long long Bnorm;        // k_mlpw, coupled control: the batch size of the error norm when it spans several ranks (0: this launch's B)
  const float* Wflat;     // flat weights (biases)
  float* z_out;           // forward: written; adjoint: the saved ẑ (read)
  int32_t* retcode;
  const float* dz_out;
  float* dz0;
  float* dtheta;
  float* stage;
  float* wts;
  int32_t* nslots;
  int cap;
  int32_t* ovf;
  int32_t *st_nfe, *st_nacc, *st_nrej, *st_ret;
  GridSync gs;
  int lds_bytes;
};

struct VCtl {
  double t, dt, tnew;
  float h, qold, wq, th, d1, ngl, gl2, sum;
  int status, j, accepted, last, hit, nfe, nacc, nrej, savej, any;
  long long iters;
  float bcast[4];
};

// one-time re-layout: flat destructure order → the two swizzled copies
static __global__ void k_build_vec(const float* __restrict__ Wflat, MlpDims dm, VecDims vd, float* __restrict__ vecw) {
  const int l = blockIdx.y;
  const int in = dm.sizes[l], out = dm.sizes[l + 1];
  const float* W = Wflat + dm.w_off[l];   // column-major [out×in]: W(o,i) at o + out*i
  {
    const int rp = vd.rpf[l], n = vd.sf[l] * vd.k4f[l] * rp * 4;
    float* dst = vecw + (size_t)vd.off_f[l] * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
      const int c4 = e & 3, i = (e >> 2) % rp, k = 4 * ((e >> 2) / rp) + c4;
      dst[e] = (i < out && k < in) ? W[i + (size_t)out * k] : 0.f;
    }
  }
  {
    const int rp = vd.rpb[l], n = vd.sb[l] * vd.k4b[l] * rp * 4;
    float* dst = vecw + (size_t)vd.off_b[l] * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
      const int c4 = e & 3, i = (e >> 2) % rp, k = 4 * ((e >> 2) / rp) + c4;
      dst[e] = (i < in && k < out) ? W[k + (size_t)out * i] : 0.f;
    }
  }
}

// y_i = Σ_k M[i,k] x_k for the rows the lanes own; the result is valid in lanes tid < rows (row i = tid).
// Wq either an LDS pointer (formed from the LDS base) or a kernel-argument pointer — never generic.
// S = 1: lane tid owns row tid (rp = the row count padded to 4). S > 1 (narrow layers): rp a power of two, lane (s, i) =
// (tid >> lg, tid & (rp−1)) adds the K-groups [s·cnt, (s+1)·cnt) of row i and the partial sums meet in LDS.
// cnt is a multiple of 4 and the swizzled array is zero-padded to S·cnt groups (x is read up to 4·S·cnt floats: every x
// buffer is followed by finite LDS data), so the loop has no clamps, no masks and no multiplications: four 16-byte weight
// loads + four broadcast x loads in flight while the previous four groups are multiplied.
template <int NT>
__device__ __forceinline__ float vec_matvec(const f32x4* __restrict__ Wq, int rp, int lg, int cnt, int S, const float* x, float* red) {
  const int tid = threadIdx.x;
  const int i = S > 1 ? (tid & (rp - 1)) : tid, s = S > 1 ? (tid >> lg) : 0;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  if (s < S && i < rp) {
    const f32x4* w = Wq + (unsigned)(s * cnt) * (unsigned)rp + i;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x) + s * cnt;
    const int st = rp;
    if (cnt < 4) {   // (uniform) a very short contraction (K ≤ 12 per lane): no pipeline to fill
      for (int it = 0; it < cnt; it++) {
        const f32x4 wv = w[it * st], xx = xv[it];
        acc0 += wv[0] * xx[0] + wv[1] * xx[1] + wv[2] * xx[2] + wv[3] * xx[3];
      }
    } else {
    f32x4 w0 = w[0], w1 = w[st], w2 = w[2 * st], w3 = w[3 * st];
    f32x4 x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
    for (int it = 4; it < cnt; it += 4) {
      w += 4 * st;
      xv += 4;
      const f32x4 n0 = w[0], n1 = w[st], n2 = w[2 * st], n3 = w[3 * st];
      const f32x4 y0 = xv[0], y1 = xv[1], y2 = xv[2], y3 = xv[3];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc0 += w0[e] * x0[e];
        acc1 += w1[e] * x1[e];
        acc2 += w2[e] * x2[e];
        acc3 += w3[e] * x3[e];
      }
      w0 = n0; w1 = n1; w2 = n2; w3 = n3;
      x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      acc0 += w0[e] * x0[e];
      acc1 += w1[e] * x1[e];
      acc2 += w2[e] * x2[e];
      acc3 += w3[e] * x3[e];
    }
    }
  }
  float acc = (acc0 + acc1) + (acc2 + acc3);
  if (S > 1) {   // uniform
    if (s < S) red[s * rp + i] = acc;
    __syncthreads();
    if (s == 0) {
      acc = red[i];
      for (int q = 1; q < S; q++) acc += red[q * rp + i];
    }
  }
  return acc;
}

// The same product with the lane's weights already in registers (KR groups, zero-padded): no weight traffic at all — the
// evaluation's only loads are the KR broadcast reads of x. This is what the square hidden layer of c4 (128×128) and of the
// smaller three-layer networks runs on: a workgroup of S·rp lanes holds the whole matrix once per direction (KR·4 = 64 VGPRs).
template <int NT, int KR>
__device__ __forceinline__ float vec_matvec_reg(const f32x4 (&w)[KR], int rp, int lg, int S, const float* x, float* red) {
  const int tid = threadIdx.x;
  const int i = S > 1 ? (tid & (rp - 1)) : tid, s = S > 1 ? (tid >> lg) : 0;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  {
    const f32x4* xv = reinterpret_cast<const f32x4*>(x) + (s < S ? s : 0) * KR;
#pragma unroll
    for (int g = 0; g < KR; g += 4) {
      const f32x4 x0 = xv[g], x1 = xv[g + 1], x2 = xv[g + 2], x3 = xv[g + 3];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc0 += w[g][e] * x0[e];
        acc1 += w[g + 1][e] * x1[e];
        acc2 += w[g + 2][e] * x2[e];
        acc3 += w[g + 3][e] * x3[e];
      }
    }
  }
  float acc = (acc0 + acc1) + (acc2 + acc3);
  if (S > 1) {   // uniform
    if (s < S) red[s * rp + i] = acc;
    __syncthreads();
    if (s == 0) {
      acc = red[i];
      for (int q = 1; q < S; q++) acc += red[q * rp + i];
    }
  }
  return acc;
}
template <int KF, int KB>
struct VRegs {
  f32x4 wf[KF > 0 ? KF : 1], wb[KB > 0 ? KB : 1];
  int l;   // the layer they belong to
};

// Per-layer parameters of the two products, kept as one 64-byte LDS record each: a dynamically indexed kernel-argument array
// costs a dependent scalar load (+ a full s_waitcnt) per field and layer — measured: most of an evaluation's time for the small
// networks — where the record is four broadcast ds_read_b128 in flight together.
struct VLayer {
  int rp, lg, k4n, S;       // geometry of the product (vec_matvec); k4n = K-groups PER LANE (a multiple of 4)
  int goff4, lofs;          // float4 offset of the swizzled weights in the global array; float offset of their LDS copy or −1
  int rows, K;              // rows produced, length contracted
  int bias, yofs;           // W·x: offset of the bias in the compact vector; offset of the output vector in P.hid (−1: the caller's dst)
  int blk_off, in32, out32; // Wᵀ·δ: where this layer's (a, δ) panels sit in a staged block
  int aofs;                 // Wᵀ·δ: offset of the layer's input activation in P.hid (−1: the evaluation's input vector)
  int pad0, pad1;
};
__device__ __forceinline__ VLayer vlayer_load(const VLayer* t) {
  const int4* q = reinterpret_cast<const int4*>(t);
  const int4 a = q[0], b = q[1], c = q[2], d = q[3];
  VLayer r;
#define RFL(x) __builtin_amdgcn_readfirstlane(x)
  r.rp = RFL(a.x); r.lg = RFL(a.y); r.k4n = RFL(a.z); r.S = RFL(a.w);
  r.goff4 = RFL(b.x); r.lofs = RFL(b.y); r.rows = RFL(b.z); r.K = RFL(b.w);
  r.bias = RFL(c.x); r.yofs = RFL(c.y); r.blk_off = RFL(c.z); r.in32 = RFL(c.w);
  r.out32 = RFL(d.x); r.aofs = RFL(d.y); r.pad0 = 0; r.pad1 = 0;
#undef RFL
  return r;
}

template <int NT>
__device__ __forceinline__ float vec_layer(const float* lds_base, const float* gw, const VLayer& q, const float* x, float* red) {
  if (q.lofs >= 0) return vec_matvec<NT>(reinterpret_cast<const f32x4*>(lds_base + q.lofs), q.rp, q.lg, q.k4n, q.S, x, red);
  return vec_matvec<NT>(reinterpret_cast<const f32x4*>(gw) + q.goff4, q.rp, q.lg, q.k4n, q.S, x, red);
}

struct VPanels {
  float *y, *yn, *tmp, *kbase, *scr, *hid, *del, *red, *biasc;
  int nsp;                      // floats of one state vector
  const float* lbase;
  const float* gw;
  const VLayer* tf;             // per-layer records of W·x and of Wᵀ·δ
  const VLayer* tb;
  int nL, DpA, act, has_pend, maxw4;
  __device__ __forceinline__ float* k(int s) const { return kbase + s * nsp; }
};

// f(z): dst rows [0,Dp) = MLP(src rows [0,Dp)) (+ pendulum); post-activation hidden vectors stay in P.hid
template <int NT, int KF, int KB>
__device__ __forceinline__ void vec_eval_rhs(const VPanels& P, const VRegs<KF, KB>& R, const VCtl* c, const float* src, float* dst) {
  const int tid = threadIdx.x, nL = P.nL;
  const float* x = src;
  for (int l = 0; l < nL; l++) {
    PROF_T(v0);
    const VLayer q = vlayer_load(P.tf + l);
    const bool lastl = q.yofs < 0;
    float* Y = lastl ? dst : P.hid + q.yofs;
    PROF_T(v1);
    float acc;
    if constexpr (KF > 0) {
      if (l == R.l) acc = vec_matvec_reg<NT, KF>(R.wf, q.rp, q.lg, q.S, x, P.red);
      else acc = vec_layer<NT>(P.lbase, P.gw, q, x, P.red);
    } else {
      acc = vec_layer<NT>(P.lbase, P.gw, q, x, P.red);
    }
    PROF_T(v2);
    if (tid < q.rows) {
      float v = acc + P.biasc[q.bias + tid];
      if (!lastl) v = act_fn(P.act, v);
      Y[tid] = v;
    }
    __syncthreads();
    PROF_T(v3);
    PROF_ADD(2, v0, v1);
    PROF_ADD(3 + l, v1, v2);
    PROF_ADD(10, v2, v3);
    x = Y;
  }
  if (P.has_pend) {
    if (tid == 0) {
      dst[0] += src[1];
      dst[1] += c->ngl * fast_sin(src[0]);
    }
    __syncthreads();
  }
}

// f, −(∂f/∂z)ᵀλ, −(∂f/∂θ)ᵀλ; rows: [0,Dp) z | [DpA,DpA+Dp) λ | [2DpA,2DpA+P) g. With `blk`, column `n` of the staged block gets (a_l, δ_l).
template <int NT, int KF, int KB>
__device__ __forceinline__ void vec_eval_bwd(const VPanels& P, const VRegs<KF, KB>& R, const VCtl* c, const float* src, float* dst, float* blk, int n) {
  const int tid = threadIdx.x, nL = P.nL, DpA = P.DpA;
  PROF_T(r0);
  vec_eval_rhs<NT, KF, KB>(P, R, c, src, dst);
  PROF_T(r1);
  const float* dl = src + DpA;
  for (int l = nL - 1; l >= 0; l--) {
    const VLayer q = vlayer_load(P.tb + l);      // rows = in, K = out
    const float* al = q.aofs < 0 ? src : P.hid + q.aofs;
    PROF_T(b0);
    float acc;
    if constexpr (KB > 0) {
      if (l == R.l) acc = vec_matvec_reg<NT, KB>(R.wb, q.rp, q.lg, q.S, dl, P.red);
      else acc = vec_layer<NT>(P.lbase, P.gw, q, dl, P.red);
    } else {
      acc = vec_layer<NT>(P.lbase, P.gw, q, dl, P.red);
    }
    PROF_T(b1);
    PROF_ADD(13 + l, b0, b1);
    if (l > 0) {
      float* dn = P.del + (l - 1) * P.maxw4;     // δ_{l−1}: every layer keeps its own buffer (they are staged together below)
      if (tid < q.rows) dn[tid] = acc * act_grad(P.act, al[tid]);
      __syncthreads();
      dl = dn;
    } else {
      if (tid < q.rows) dst[DpA + tid] = -acc;
      __syncthreads();
    }
  }
  // The (a_l, δ_l) panels go to HBM in ONE pass after the last product: on gfx9 stores and loads share the in-order vmcnt
  // counter, so a weight load issued behind a layer's staging stores waited for their write acknowledgements (measured:
  // ≈ 7 000 of an evaluation's 22 000 cycles on c2 / c4) — now only the next evaluation's first load does.
  PROF_T(sg0);
  PROF_ADD(23, r0, r1);
  PROF_ADD(24, r1, sg0);
  if (blk) {
    for (int l = nL - 1; l >= 0; l--) {
      const VLayer q = vlayer_load(P.tb + l);
      const float* al = q.aofs < 0 ? src : P.hid + q.aofs;
      const float* dv = l == nL - 1 ? src + DpA : P.del + l * P.maxw4;
      float* ga = blk + q.blk_off + n * q.in32;
      float* gd = blk + q.blk_off + NB * q.in32 + n * q.out32;
      for (int r = tid; r < q.in32; r += NT) ga[r] = r < q.rows ? al[r] : 0.f;
      for (int r = tid; r < q.out32; r += NT) gd[r] = r < q.K ? dv[r] : 0.f;
    }
  }
  PROF_T(sg1);
  PROF_ADD(22, sg0, sg1);
  if (P.has_pend) {
    if (tid == 0) {
      float sn, cs;
      fast_sincos(src[0], sn, cs);
      const float l0 = src[DpA + 0], l1 = src[DpA + 1];
      dst[DpA + 0] -= c->ngl * cs * l1;
      dst[DpA + 1] -= l0;
      dst[2 * DpA] = -(c->gl2 * sn * l1);
    }
    __syncthreads();
  }
}

// Σ v[0..n) in a fixed order (strided partial sums in wave 0, butterfly); the result is left in c->sum for every thread
template <int NT>
__device__ __forceinline__ float vec_sum(const float* v, int n, VCtl* c) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    float p = 0.f;
    for (int r = tid; r < n; r += 64) p += v[r];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off);
    if (tid == 0) c->sum = p;
  }
  __syncthreads();
  return c->sum;
}

// grid-wide deterministic sum for many small workgroups: arrival as in grid_sum4, then ALL threads add the partials in an
// order that depends on the indices only (thread t takes w ≡ t mod NT ascending; butterfly per wave; waves in order)
template <int NT>
__device__ __forceinline__ void vec_grid_sum(const GridSync& gs, unsigned& gen, float (&v)[4], VCtl* c, float* red) {
  const int tid = threadIdx.x;
  if (gs.nwg == 1) {
    if (tid == 0)
      for (int i = 0; i < 4; i++) c->bcast[i] = v[i];
    __syncthreads();
    for (int i = 0; i < 4; i++) v[i] = c->bcast[i];
    __syncthreads();
    return;
  }
  gen++;
  PROF_T(gs0);
  float* slots = gs.slots + (size_t)(gen & 1) * gs.nwg * 4;
  if (tid == 0) {
    float* mine = slots + (size_t)blockIdx.x * 4;
    for (int i = 0; i < 4; i++) mine[i] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(gs.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * (unsigned)gs.nwg;
    long long spins = 0;
    bool aborted = __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // sticky
    while (!aborted && __hip_atomic_load(gs.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000LL) {   // a peer is not resident — give up instead of hanging the GPU (the launch is cooperative: cannot happen)
        __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        aborted = true;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    c->any = aborted ? 1 : 0;
  }
  __syncthreads();
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  for (int w = tid; w < gs.nwg; w += NT)
    for (int i = 0; i < 4; i++) p[i] += __hip_atomic_load(slots + (size_t)w * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    for (int i = 0; i < 4; i++) p[i] += __shfl_xor(p[i], off);
  if ((tid & 63) == 0)
    for (int i = 0; i < 4; i++) red[(tid >> 6) * 4 + i] = p[i];
  __syncthreads();
  const bool aborted = c->any != 0;
  for (int i = 0; i < 4; i++) {
    float t = red[i];
    for (int w = 1; w < NT / 64; w++) t += red[w * 4 + i];
    v[i] = aborted ? __int_as_float(0x7fc00000) : t;   // a timed-out barrier poisons the sums: the solve ends with retcode != 0
  }
  __syncthreads();
  PROF_T(gs1);
  PROF_ADD(12, gs0, gs1);
  PROF_ADD(21, gs1 - 1, gs1);
}

template <int SOLVER, int NT, bool ADJ, int KF = 0, int KB = 0>
__global__ void __launch_bounds__(NT, (KF > 0 ? 2 : 1)) k_mlpv(MlpDims dm, VecDims vd, KOpts o, VArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = o.T, B = o.B, Dp = dm.Dp, DpA = dm.DpA, D = dm.D, NP = dm.P;
  const int tid = threadIdx.x, b = blockIdx.x;
  VCtl* c = reinterpret_cast<VCtl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(VCtl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  const int NS = ADJ ? 2 * DpA + NP : Dp;         // state rows in use
  const int NREAL = ADJ ? 2 * Dp + NP : Dp;       // entries that count in the error norm
  VPanels P;
  P.nsp = ADJ ? vd.nsp_b : vd.nsp_f;
  float* p = base;
  P.y = p; p += P.nsp;
  P.yn = p; p += P.nsp;
  P.tmp = p; p += P.nsp;
  P.kbase = p; p += 7 * P.nsp;
  P.scr = p; p += P.nsp;
  P.hid = p; p += vd.htotal;
  P.del = p; p += MAXL * vd.maxw4;       // δ of every hidden layer (staged together after the last product)
  P.red = p; p += NT;                  // split-K partial sums (S·rp ≤ NT) / the grid sum's per-wave partials
  P.biasc = p; p += (dm.nbias + 3) & ~3;
  VLayer* tf = reinterpret_cast<VLayer*>(p); p += MAXL * (sizeof(VLayer) / 4);
  VLayer* tb = reinterpret_cast<VLayer*>(p); p += MAXL * (sizeof(VLayer) / 4);
  P.tf = tf; P.tb = tb;
  P.nL = dm.nL; P.DpA = dm.DpA; P.act = dm.act; P.has_pend = dm.has_pend; P.maxw4 = vd.maxw4;
  const int nfloat = (int)(p - base);
  for (int i = tid; i < nfloat; i += NT) base[i] = 0.f;   // pad entries must be 0 (they are read as x_k)
  for (int i = tid; i < T; i += NT) s_ts[i] = a.ts[i];
  __syncthreads();
  P.lbase = reinterpret_cast<const float*>(smem);
  P.gw = a.vecw;
  for (int l = 0; l < dm.nL; l++)
    for (int i = tid; i < dm.sizes[l + 1]; i += NT) P.biasc[dm.bias_lin[l] + i] = a.Wflat[dm.b_off[l] + i];
  {   // LDS cache of the swizzled weights while they fit (evaluation order), and the per-layer records
    float* cache = p;
    float* cend = reinterpret_cast<float*>(smem + a.lds_bytes);
    for (int pass = 0; pass < (ADJ ? 2 : 1); pass++)
      for (int l = 0; l < dm.nL; l++) {
        const int n = 4 * (pass ? vd.sb[l] * vd.k4b[l] * vd.rpb[l] : vd.sf[l] * vd.k4f[l] * vd.rpf[l]);
        const float* srcw = a.vecw + (size_t)(pass ? vd.off_b[l] : vd.off_f[l]) * 4;
        const bool in_regs = l == vd.reg_l && (pass ? KB > 0 : KF > 0);
        const bool fits = !in_regs && cache + n <= cend;
        if (fits) {
          const f32x4* s4 = reinterpret_cast<const f32x4*>(srcw);
          f32x4* d4 = reinterpret_cast<f32x4*>(cache);
          for (int i = tid; i < n / 4; i += NT) d4[i] = s4[i];
        }
        if (tid == 0) {
          VLayer q;
          const int in = dm.sizes[l], out = dm.sizes[l + 1];
          q.rp = pass ? vd.rpb[l] : vd.rpf[l];
          q.lg = pass ? vd.lgb[l] : vd.lgf[l];
          q.k4n = pass ? vd.k4b[l] : vd.k4f[l];
          q.S = pass ? vd.sb[l] : vd.sf[l];
          q.goff4 = pass ? vd.off_b[l] : vd.off_f[l];
          q.lofs = fits ? (int)(cache - P.lbase) : -1;
          q.rows = pass ? in : out;
          q.K = pass ? out : in;
          q.bias = dm.bias_lin[l];
          q.yofs = l == dm.nL - 1 ? -1 : vd.hoff[l];
          q.blk_off = dm.blk_off[l];
          q.in32 = pad32(in);
          q.out32 = pad32(out);
          q.aofs = l == 0 ? -1 : vd.hoff[l - 1];
          q.pad0 = q.pad1 = 0;
          (pass ? tb : tf)[l] = q;
        }
        if (fits) cache += n;
      }
  }
  __syncthreads();
  VRegs<KF, KB> R;
  R.l = (KF > 0 || KB > 0) ? vd.reg_l : -1;
  if (KF > 0 || KB > 0) {   // the lane's share of the register-resident layer, once (geometry: rp·S = NT lanes, VREG_K groups each)
    const int l = vd.reg_l;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(a.vecw);
    if (KF > 0) {
      const int rp = vd.rpf[l], S = vd.sf[l], i = S > 1 ? (tid & (rp - 1)) : tid, sg = S > 1 ? (tid >> vd.lgf[l]) : 0;
      const bool on = sg < S && i < rp;
#pragma unroll
      for (int g = 0; g < (KF > 0 ? KF : 1); g++) R.wf[g] = on ? g4[vd.off_f[l] + (sg * KF + g) * rp + i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (KB > 0) {
      const int rp = vd.rpb[l], S = vd.sb[l], i = S > 1 ? (tid & (rp - 1)) : tid, sg = S > 1 ? (tid >> vd.lgb[l]) : 0;
      const bool on = sg < S && i < rp;
#pragma unroll
      for (int g = 0; g < (KB > 0 ? KB : 1); g++) R.wb[g] = on ? g4[vd.off_b[l] + (sg * KB + g) * rp + i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = fabs(tend - t0);
  unsigned gen = 0;
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;   // weighted stages per step attempt (adjoint)
  const int tile = b >> 4, ncol = b & 15;
  float* const my_stage = ADJ ? a.stage + (size_t)tile * a.cap * dm.blk_floats : nullptr;
  float* const my_wts = ADJ ? a.wts + (size_t)tile * a.cap * NB : nullptr;
  int slot_base = 0;
  bool overflow = false;
  const int blk_floats = dm.blk_floats;
#define VFOR(idx) for (int idx = tid; idx < NS; idx += NT)

  // ---- initial / terminal condition -------------------------------------------------------------------------------------
  if (!ADJ) {
    for (int r = tid; r < D; r += NT) P.y[r] = a.z0[(size_t)b * D + r];
  } else {
    for (int r = tid; r < Dp; r += NT) {
      const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * (T - 1)) + r;
      P.y[r] = a.z_out[srcg];
      P.y[DpA + r] = a.dz_out[srcg];
    }
  }
  __syncthreads();
  // Step control lives in REGISTERS, computed redundantly by every lane (a SIMT machine does that for free): a control block
  // in LDS, written by one lane and read back by the others, made every decision a chain of ≈ 100-cycle LDS round trips —
  // measured ≈ 14 000 cycles per step, more than the seven evaluations' arithmetic. Only sums cross lanes (vec_sum).
  double t = ADJ ? tend : t0, dt = 0.0, tnew = 0.0;
  float h = 0.f, qold = 1e-4f, wq = 0.f, d1 = 0.f;
  int status, j = ADJ ? T - 2 : 1, last = 0, hit = 0, nfe = 0, nacc = 0, nrej = 0;
  long long iters = 0;
  {
    bool bad = false;
    if (ADJ)
      for (int r = 0; r < Dp; r++) bad = bad || !isfinite(P.y[r]);
    // a failed forward trajectory is a constant NaN block ⇒ zero gradient  [REF GOKU.jl:114]
    status = bad ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1);
    float L = 1.f;
    if (dm.has_pend) L = a.theta[(size_t)b * NP];
    if (tid == 0) {
      c->ngl = -10.0f / L;
      c->gl2 = 10.0f / (L * L);
    }
  }
  __syncthreads();
  if (ADJ && status > 1) {   // neutralise the NaN column
    VFOR(r) P.y[r] = 0.f;
    __syncthreads();
  }
  if (!ADJ)
    for (int r = tid; r < Dp; r += NT) a.z_out[(size_t)b * Dp + r] = P.y[r];   // save time 0 = ẑ₀ itself (augmented rows 0)

  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : (ADJ ? 3 : 4);
  const float dirn = ADJ ? -1.f : 1.f;
  const float nnorm = coupled ? (float)NREAL * (float)B : (float)NREAL;

  // start of a step attempt: false when the trajectory has finished
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
        tnew = hmag;              // step magnitude actually attempted
        h = -(float)hmag;
        wq = (float)hmag;         // quadrature weight scale |h|
      }
    } else {
      h = 0.f;
      wq = 0.f;
      hit = 0;
    }
    return status == 0;
  };

  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  int phase = (ADJ && !auto_dt) ? PH_STAGE : PH_K0, s = 0;
  bool running = T > 1 && status == 0;
  if (ADJ && running && !auto_dt) {
    dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    running = begin_step();
  }
  while (running) {
    PROF_T(pl0);
    // ---- input of this evaluation ----------------------------------------------------------------------------------------
    const float* src = P.y;
    bool any_w = false;
    if (phase == PH_INIT1) src = P.tmp;
    if (phase == PH_STAGE) {
      if (SOLVER == LDE_SOLVER_TSIT5) {
        if (s > 0) {
          float* dstp = s < 6 ? P.tmp : P.yn;
          // the tableau row as immediates (a dynamically indexed constant table is a dependent load per coefficient)
#define VSTAGE(S_)                                                             \
  case S_:                                                                     \
    VFOR(idx) {                                                                \
      float accv = ts5::A[S_][0] * P.k(0)[idx];                                \
      _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += ts5::A[S_][jj] * P.k(jj)[idx]; \
      dstp[idx] = P.y[idx] + h * accv;                                         \
    }                                                                          \
    break;
          switch (s) {
            VSTAGE(1) VSTAGE(2) VSTAGE(3) VSTAGE(4) VSTAGE(5) VSTAGE(6)
            default: break;
          }
#undef VSTAGE
          src = dstp;
        }
        any_w = ADJ && s < 6;
      } else if (ADJ) {
        if (s > 0) {
          const float cs = (s == 3 ? 1.0f : 0.5f) * h;
          VFOR(idx) P.tmp[idx] = P.y[idx] + cs * P.k(s - 1)[idx];
          src = P.tmp;
        }
        any_w = true;
      } else if (s < 4) {
        const float cs = (s == 3 ? 1.0f : 0.5f) * h;
        VFOR(idx) P.tmp[idx] = P.y[idx] + cs * P.k(s - 1)[idx];
        src = P.tmp;
      } else {
        const float h6 = h * (1.0f / 6.0f);
        VFOR(idx) P.yn[idx] = P.y[idx] + h6 * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        src = P.yn;
      }
      if (ADJ && s == 0 && slot_base + NST > a.cap) overflow = true;   // out of staging slots: the tile kernel redoes this call
      __syncthreads();
    }
    float* dst = phase == PH_K0 ? P.k(0) : (phase == PH_INIT1 ? P.k(1) : P.k(s));
    PROF_T(pe0);
    PROF_ADD(0, pl0, pe0);
    if (ADJ)
      vec_eval_bwd<NT, KF, KB>(P, R, c, src, dst, (any_w && !overflow) ? my_stage + (size_t)(slot_base + s) * blk_floats : nullptr, ncol);
    else
      vec_eval_rhs<NT, KF, KB>(P, R, c, src, dst);
    if (status == 0) nfe++;
    PROF_T(pe1);
    PROF_ADD(1, pe0, pe1);
    PROF_ADD(20, pe1 - 1, pe1);   // evaluations counted
#if LDE_PROF
    struct ProfEnd { long long t0; __device__ ~ProfEnd() { PROF_T(t1); PROF_ADD(11, t0, t1); } } prof_end{pe1};
#endif

    // ---- what follows the evaluation ---------------------------------------------------------------------------------
    if (phase == PH_K0 && !(ADJ || auto_dt)) {   // forward with a user step size: no initial-step probe
      dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
      phase = PH_STAGE;
      s = 1;
      running = begin_step();
    } else if (phase == PH_K0) {
      // Hairer–Nørsett–Wanner, part 1: d0, d1, trial Euler step
      VFOR(idx) {
        const float yv = P.y[idx];
        const float sk = fast_rcp(o.abstol + fabsf(yv) * o.reltol);
        P.scr[idx] = sk;
        const float a0 = yv * sk, a1 = P.k(0)[idx] * sk;
        P.tmp[idx] = a0 * a0;
        P.yn[idx] = a1 * a1;
      }
      __syncthreads();
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      v[0] = vec_sum<NT>(P.tmp, NS, c);
      __syncthreads();
      v[1] = vec_sum<NT>(P.yn, NS, c);
      __syncthreads();
      if (coupled) {
        if (status != 0) v[0] = v[1] = 0.f;
        vec_grid_sum<NT>(a.gs, gen, v, c, P.red);
      }
      {
        const float d0 = sqrtf(v[0] / nnorm);
        d1 = sqrtf(v[1] / nnorm);
        double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
        if (dt0 > dtmax) dt0 = dtmax;
        dt = dt0;
        h = status == 0 ? dirn * (float)dt0 : 0.f;
      }
      VFOR(idx) P.tmp[idx] = P.y[idx] + h * P.k(0)[idx];
      __syncthreads();
      phase = PH_INIT1;
    } else if (phase == PH_INIT1) {
      VFOR(idx) {
        const float dd = (P.k(1)[idx] - P.k(0)[idx]) * P.scr[idx];
        P.yn[idx] = dd * dd;
      }
      __syncthreads();
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      w[0] = vec_sum<NT>(P.yn, NS, c);
      __syncthreads();
      if (coupled) {
        if (status != 0) w[0] = 0.f;
        vec_grid_sum<NT>(a.gs, gen, w, c, P.red);
      }
      {
        const double dt0 = dt;
        const float d2 = sqrtf(w[0] / nnorm) * fast_rcp((float)dt0);
        const float dm_ = fmaxf(d1, d2);
        const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
        const double dn = fmin(100.0 * dt0, dt1);
        dt = dn > dtmax ? dtmax : dn;
      }
      phase = PH_STAGE;
      s = ADJ ? 0 : 1;
      running = begin_step();
    } else if (s < LAST_STAGE) {
      s++;
    } else {
      // ---- all stages of this attempt are done: error estimate, control -----------------------------------------------
      if (ADJ && SOLVER == LDE_SOLVER_RK4) {
        const float h6 = h * (1.0f / 6.0f);
        VFOR(idx) P.yn[idx] = P.y[idx] + h6 * (P.k(0)[idx] + 2.0f * (P.k(1)[idx] + P.k(2)[idx]) + P.k(3)[idx]);
        __syncthreads();
      }
      VFOR(idx) {
        float r2 = 0.f;
        const float yv = P.y[idx], ynv = P.yn[idx];
        if (o.adaptive) {
          float er = ts5::BT[0] * P.k(0)[idx];
#pragma unroll
          for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * P.k(jj)[idx];
          er *= h;
          const float sk = o.abstol + fmaxf(fabsf(yv), fabsf(ynv)) * o.reltol;
          const float r = er * fast_rcp(sk);
          r2 = r * r;
        }
        P.scr[idx] = isfinite(ynv) ? r2 : __int_as_float(0x7fc00000);   // a non-finite state poisons the sum
      }
      __syncthreads();
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      v[0] = vec_sum<NT>(P.scr, NS, c);
      if (coupled) {
        __syncthreads();
        if (status != 0) v[0] = 0.f;
        vec_grid_sum<NT>(a.gs, gen, v, c, P.red);
      }
      bool accepted = false;
      if (status == 0) {
        const float s2 = v[0];
        const float EEst = o.adaptive ? sqrtf(s2 / nnorm) : (s2 == s2 ? 0.f : s2);
        const double hmag = ADJ ? tnew : dt;
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
      if (!ADJ) {
        // ---- dense output at every save time inside the accepted step --------------------------------------------------
        while (accepted && j < T && s_ts[j] <= tnew) {
          const double tj = s_ts[j];
          const float th = (tj >= tnew || (j == T - 1 && last)) ? 2.0f : (float)(tj - t) * fast_rcp(wq);
          for (int row = tid; row < Dp; row += NT) {
            float out;
            if (th > 1.5f) out = P.yn[row];
            else if (SOLVER == LDE_SOLVER_TSIT5) {
              float bw[7];
              tsit5_interp_weights(th, bw);
              float acc = bw[0] * P.k(0)[row];
#pragma unroll
              for (int q = 1; q < 7; q++) acc += bw[q] * P.k(q)[row];
              out = P.y[row] + wq * acc;
            } else {
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              out = h00 * P.y[row] + (h10 * wq) * P.k(0)[row] + h01 * P.yn[row] + (h11 * wq) * P.k(4)[row];
            }
            a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * j) + row] = out;
          }
          j++;
        }
        __syncthreads();
        if (accepted) {   // advance (FSAL: the last slope becomes k1)
          VFOR(idx) {
            P.y[idx] = P.yn[idx];
            P.k(0)[idx] = P.k(LAST_STAGE)[idx];
          }
          t = tnew;
          if (last) status = 1;
        }
        __syncthreads();
        s = 1;
        running = begin_step();
      } else {
        // ---- the attempt's staged evaluations count only if it was accepted: their quadrature weights are written now
        if (accepted && !overflow) {
          if (tid < NST) {
            float bs;
            if (SOLVER == LDE_SOLVER_TSIT5) bs = ts5::A[6][tid];
            else bs = (tid == 0 || tid == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
            my_wts[(size_t)(slot_base + tid) * NB + ncol] = wq * bs;
          }
          slot_base += NST;
        }
        __syncthreads();
        if (accepted) {   // advance; jump at a save time
          VFOR(idx) P.y[idx] = P.yn[idx];
          __syncthreads();
          if (hit) {
            for (int row = tid; row < Dp; row += NT) {
              const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * j) + row;
              P.y[DpA + row] += a.dz_out[srcg];
              if (o.checkpoint) P.y[row] = a.z_out[srcg];
            }
            t = s_ts[j];
            j--;
            if (j < 0) status = 1;
          } else
            t -= tnew;
        }
        __syncthreads();
        s = 0;
        running = begin_step();
      }
    }
  }

  // ---- results --------------------------------------------------------------------------------------------------------
  const int st = status;
  if (!ADJ) {
    if (st > 1) {
      const float qn = __int_as_float(0x7fc00000);
      for (int e = tid; e < Dp * T; e += NT) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * (e / Dp)) + (e % Dp)] = qn;
    }
    if (tid == 0) {
      const int ret = st > 1 ? st - 1 : 0;
      if (a.retcode) a.retcode[b] = ret;
      a.st_ret[b] = ret;
    }
  } else {
    for (int row = tid; row < D; row += NT) a.dz0[(size_t)b * D + row] = st > 1 ? 0.f : P.y[DpA + row];
    for (int row = tid; row < NP; row += NT) a.dtheta[(size_t)b * NP + row] = st > 1 ? 0.f : P.y[2 * DpA + row];
    if (tid == 0) {
      a.st_ret[b] = st > 1 ? st - 1 : 0;
      atomicMax(&a.nslots[tile], slot_base);
      if (overflow) __hip_atomic_store(a.ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    const bool rep = !coupled || b == 0;   // coupled: one step sequence for the whole batch, reported once
    a.st_nfe[b] = rep ? nfe : 0;
    a.st_nacc[b] = rep ? nacc : 0;
    a.st_nrej[b] = rep ? nrej : 0;
  }
#undef VFOR
}

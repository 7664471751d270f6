// lde_mlp4.h — the MLP adjoint for networks whose weights fit LDS ONCE: four trajectories per WAVE, no barriers.
// (included by lde_mlp.hip inside namespace lde: uses BwdArgs, GridSync, grid_sum4, KOpts, act_fn/act_grad.)
//
// Why: in the 16-columns-per-workgroup kernel (k_mlp_adjoint) every layer costs ≈ 2 000 cycles of fixed latency — operand
// round trips through LDS, split-K reduction, epilogue, barrier — regardless of its size, and a small
// batch leaves most of the chip idle. Here a wave owns 4 columns for the whole solve and never synchronises with
// another wave (per-trajectory / fixed-step control):
//  * `v_mfma_f32_4x4x1_16B_f32` computes 16 independent 4×4 outer products per instruction. Block b ↔ rows 4b..4b+3
//    of a 64-row tile, the 4 columns of every block are the wave's 4 trajectories: D[b][i][j] += W[64rt+4b+i][k]·X[k][j]
//    (lane 4b+j, register i). A vector of R rows therefore lives in ⌈R/64⌉ float4 registers per lane, and every
//    elementwise operation of the integrator (stage sums, error terms, jumps) is plain register arithmetic.
//  * The B operand X[k][j] must be the same in all 16 blocks: the vector is written once to a wave-private 4-column
//    panel in LDS (one ds_write_b128 per tile) and read back broadcast, one ds_read_b128 per 4 values of k.
//  * ONE row-major copy of W_l in LDS (row stride/4 odd ⇒ conflict-free) serves both products: W·x reads a row
//    segment (ds_read_b128), Wᵀ·δ reads a column segment (4 × ds_read_b32).
//  * The weight gradient is staged exactly as in k_mlp_adjoint (same block layout, 16 columns = 4 waves per slot)
//    and formed by k_mlp_dw afterwards.
// Limits: D' ≤ 64, hidden width ≤ 64·NTH (NTH ≤ 4), weights + biases ≤ LDS. Anything else runs k_mlp_adjoint.

struct Mlp4Dims {
  int wl_off[MAXL];   // float offset (in the LDS weight area) of layer l's row-major copy W_l[o][ldw]
  int ldw[MAXL];      // row stride: ≥ pad4(in), (ldw/4) odd
  int bl_off[MAXL];   // float offset of layer l's bias copy (padded with zeros to a multiple of 64 rows)
  int w_total;        // floats of the weight area
  int ldx;            // stride of one column of a wave's operand panel
  int wpb;            // waves per block
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

struct St4 {           // [z; λ; g] of the wave's 4 columns: lane 4b+j holds rows 4b..4b+3 of column j; g lives in lanes b == 0
  f32x4 z, lam;
  float g;
};
__device__ __forceinline__ St4 st4_zero() { return St4{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, 0.f}; }
__device__ __forceinline__ St4 st4_axpy(const St4& y, float h, const St4& k) { return St4{y.z + h * k.z, y.lam + h * k.lam, y.g + h * k.g}; }

// Σ over the 16 lanes of a column (lanes with equal lane & 3)
__device__ __forceinline__ float col_sum(float v) {
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// Σ over the wave's 4 columns of a per-column value (already equal in the 16 lanes of each column)
__device__ __forceinline__ float wave_cols_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

// NT 64-row tiles of W·X at once (they share the X operand): wr[t] = &W[row_t][0] of this lane's row in tile t,
// xj = this lane's column of the operand panel. A lone wave has nobody to hide LDS latency behind, so the operands of
// the next U K-quads are requested before the MFMAs of the current U issue: two register sets used alternately (no
// copies), pointers advanced by constants (no per-load address arithmetic). The last request reads up to U quads
// past the end of the row / panel — inside the workgroup's LDS allocation, never multiplied.
template <int NT>
__device__ __forceinline__ void g4_fwd(const float* const (&wr)[NT], const float* xj, int K4, f32x4 (&out)[NT]) {
  constexpr int U = NT == 1 ? 4 : 2;
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xq = xj;
  const float* wq[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) wq[t] = wr[t];
  f32x4 xa[U], wa[U][NT], xb[U], wb[U][NT];
  auto load = [&](f32x4 (&x)[U], f32x4 (&w)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = *reinterpret_cast<const f32x4*>(xq + 4 * u);
#pragma unroll
      for (int t = 0; t < NT; t++) w[u][t] = *reinterpret_cast<const f32x4*>(wq[t] + 4 * u);
    }
    xq += 4 * U;
#pragma unroll
    for (int t = 0; t < NT; t++) wq[t] += 4 * U;
  };
  auto mac = [&](const f32x4 (&x)[U], const f32x4 (&w)[U][NT], int nvalid) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (u < nvalid) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int t = 0; t < NT; t++) acc[t][q & 1] = mfma4(w[u][t][q], x[u][q], acc[t][q & 1]);
        }
      }
    }
  };
  load(xa, wa);
  for (int k0 = 0;;) {
    load(xb, wb);
    mac(xa, wa, K4 - k0);
    k0 += U;
    if (k0 >= K4) break;
    load(xa, wa);
    mac(xb, wb, K4 - k0);
    k0 += U;
    if (k0 >= K4) break;
  }
#pragma unroll
  for (int t = 0; t < NT; t++) out[t] = acc[t][0] + acc[t][1];
}
// NT 64-row tiles of Wᵀ·δ: wcol[t] = &W[0][col_t] of this lane's input index in tile t (column segments: 4 ds_read_b32)
template <int NT>
__device__ __forceinline__ void g4_bwd(const float* const (&wcol)[NT], int ldw, const float* dj, int K4, f32x4 (&out)[NT]) {
  constexpr int U = NT == 1 ? 4 : 2;
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* dq = dj;
  const float* wq[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) wq[t] = wcol[t];
  const int step = 4 * U * ldw;
  f32x4 da[U], wa[U][NT], db[U], wb[U][NT];
  auto load = [&](f32x4 (&dd)[U], f32x4 (&w)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      dd[u] = *reinterpret_cast<const f32x4*>(dq + 4 * u);
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const float* r = wq[t] + (4 * u) * ldw;
        w[u][t] = f32x4{r[0], r[ldw], r[2 * ldw], r[3 * ldw]};
      }
    }
    dq += 4 * U;
#pragma unroll
    for (int t = 0; t < NT; t++) wq[t] += step;
  };
  auto mac = [&](const f32x4 (&dd)[U], const f32x4 (&w)[U][NT], int nvalid) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (u < nvalid) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int t = 0; t < NT; t++) acc[t][q & 1] = mfma4(w[u][t][q], dd[u][q], acc[t][q & 1]);
        }
      }
    }
  };
  load(da, wa);
  for (int k0 = 0;;) {
    load(db, wb);
    mac(da, wa, K4 - k0);
    k0 += U;
    if (k0 >= K4) break;
    load(da, wa);
    mac(db, wb, K4 - k0);
    k0 += U;
    if (k0 >= K4) break;
  }
#pragma unroll
  for (int t = 0; t < NT; t++) out[t] = acc[t][0] + acc[t][1];
}

// f, −(∂f/∂z)ᵀλ, −(∂f/∂θ)ᵀλ for the wave's 4 columns; when `blk` is given the (a_l, δ_l) panels are staged there
// (columns 4·wq + j of the 16-column block). xp = the wave's nL+1 operand panels: panel l holds the INPUT of layer l
// (panel 0 = z, panel l = activation of layer l-1) — it is the B operand of the forward product and is read back in
// the backward sweep for the activation derivative and the staging copy; panel nL carries δ. The layer loops are
// runtime loops (a fully unrolled variant with the activations in registers was 14 k instructions: the instruction
// cache, not the matrix pipe, set its speed).
template <int NTH>
__device__ __forceinline__ void eval4(const MlpDims& dm, const Mlp4Dims& md, const float* lw, float* xp, const St4& src,
                                      St4& dst, float ngl, float gl2, float* blk, int wq) {
  const int lane = threadIdx.x & 63, b = lane >> 2, j = lane & 3, nL = dm.nL, actk = dm.act;
  const int pstride = 4 * md.ldx;
  float* xj = xp + j * md.ldx;           // this lane's column in panel 0
  float* dj = xj + nL * pstride;         // … in the δ panel
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- 1. forward (relu masks are recomputed here, not stored by the forward solve)
  *reinterpret_cast<f32x4*>(xj + 4 * b) = src.z;
  for (int l = 0; l < nL; l++) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const float* Wl = lw + md.wl_off[l];
    const float* bias = lw + md.bl_off[l];
    const float* xin = xj + l * pstride;
    const int ldw = md.ldw[l], K4 = (in + 3) >> 2;
    const bool last = l == nL - 1;
    f32x4 accv[NTH];
#pragma unroll
    for (int rt = 0; rt < NTH; rt++) accv[rt] = zero4;
    if (NTH > 1 && out > 64) {   // hidden-width output: all NTH tiles together (tiles past `out` repeat its last row; masked below)
      const float* wr[NTH];
#pragma unroll
      for (int rt = 0; rt < NTH; rt++) wr[rt] = Wl + (size_t)min(64 * rt + lane, out - 1) * ldw;
      g4_fwd<NTH>(wr, xin, K4, accv);
    } else {
      const float* wr[1] = {Wl + (size_t)min(lane, out - 1) * ldw};
      f32x4 o1[1];
      g4_fwd<1>(wr, xin, K4, o1);
      accv[0] = o1[0];
    }
#pragma unroll
    for (int rt = 0; rt < NTH; rt++) {
      if (64 * rt < out) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + 64 * rt + 4 * b);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const bool ok = 64 * rt + 4 * b + i < out;
          float x = accv[rt][i] + bq[i];
          if (!last) x = act_fn(actk, x);
          v[i] = ok ? x : 0.f;
        }
        if (!last) *reinterpret_cast<f32x4*>(xj + (l + 1) * pstride + 64 * rt + 4 * b) = v;
        else if (rt == 0) dst.z = v;
      }
    }
  }
  // ---- 2. back-propagate λ; δ_L = λ_stage
  f32x4 d[NTH];
#pragma unroll
  for (int t = 0; t < NTH; t++) d[t] = t == 0 ? src.lam : zero4;
  for (int l = nL - 1; l >= 0; l--) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const float* ain = xj + l * pstride;   // input activation of layer l (own rows: ain[64t + 4b + i])
    if (blk && LDE_ABL != 4) {
      const int in32 = pad32(in), out32 = pad32(out);
      float* ga = blk + dm.blk_off[l] + (4 * wq + j) * in32;
      float* gd = blk + dm.blk_off[l] + NB * in32 + (4 * wq + j) * out32;
#pragma unroll
      for (int t = 0; t < NTH; t++) {
        const int r0 = 64 * t + 4 * b;
        if (r0 < in32) *reinterpret_cast<f32x4*>(ga + r0) = *reinterpret_cast<const f32x4*>(ain + r0);
        if (r0 < out32) *reinterpret_cast<f32x4*>(gd + r0) = d[t];
      }
    }
#pragma unroll
    for (int t = 0; t < NTH; t++)
      if (64 * t < out) *reinterpret_cast<f32x4*>(dj + 64 * t + 4 * b) = d[t];
    const float* Wl = lw + md.wl_off[l];
    const int ldw = md.ldw[l], K4 = (out + 3) >> 2;
    f32x4 Gv[NTH];
#pragma unroll
    for (int rt = 0; rt < NTH; rt++) Gv[rt] = zero4;
    if (NTH > 1 && in > 64) {
      const float* wcl[NTH];
#pragma unroll
      for (int rt = 0; rt < NTH; rt++) wcl[rt] = Wl + min(64 * rt + lane, in - 1);
      g4_bwd<NTH>(wcl, ldw, dj, K4, Gv);
    } else {
      const float* wcl[1] = {Wl + min(lane, in - 1)};
      f32x4 o1[1];
      g4_bwd<1>(wcl, ldw, dj, K4, o1);
      Gv[0] = o1[0];
    }
#pragma unroll
    for (int rt = 0; rt < NTH; rt++) {
      f32x4 v = zero4;
      if (64 * rt < in) {
        const f32x4 hv = l > 0 ? *reinterpret_cast<const f32x4*>(ain + 64 * rt + 4 * b) : zero4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const bool ok = 64 * rt + 4 * b + i < in;
          const float x = l > 0 ? Gv[rt][i] * act_grad(actk, hv[i]) : -Gv[rt][i];
          v[i] = ok ? x : 0.f;
        }
      }
      d[rt] = v;
    }
  }
  dst.lam = d[0];
  dst.g = 0.f;
  // ---- 3. known-physics part: J = [[0,1],[ngl·cos x, 0]], ∂f₂/∂L = gl2·sin x   (rows 0,1 live in the lanes b == 0)
  if (dm.has_pend && b == 0) {
    float sn, cs;
    fast_sincos(src.z[0], sn, cs);
    const float l0 = src.lam[0], l1 = src.lam[1];
    dst.z[0] += src.z[1];
    dst.z[1] += ngl * sn;
    dst.lam[0] -= ngl * cs * l1;
    dst.lam[1] -= l0;
    dst.g = -(gl2 * sn * l1);
  }
}

// Reverse-time solve of [z; λ; g_θ]: 4 columns per wave, forced stops + jumps at the save times.
template <int SOLVER, int NTH>
__global__ void __launch_bounds__(256) k_mlp4_adjoint(MlpDims dm, Mlp4Dims md, KOpts o, BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  const int T = o.T, B = o.B, Dp = dm.Dp, D = dm.D, NP = dm.P, nL = dm.nL;
  float* lw = reinterpret_cast<float*>(smem4);
  float* panels = lw + md.w_total;
  const int pwave = (nL + 1) * 4 * md.ldx;   // floats of one wave's operand panels
  double* s_ts = reinterpret_cast<double*>(panels + md.wpb * pwave);
  float* s_red = reinterpret_cast<float*>(s_ts + T);   // [wpb][4] workgroup reduction + [4] broadcast
  const int tid = threadIdx.x, lane = tid & 63, b = lane >> 2, j = lane & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthr = md.wpb * 64;
  // ---- weights: one row-major padded copy per layer, biases padded with zeros
  for (int i = tid; i < md.w_total; i += nthr) lw[i] = 0.f;
  __syncthreads();
  for (int l = 0; l < nL; l++) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1], ldw = md.ldw[l];
    const float* W = a.Wflat + dm.w_off[l];
    for (int e = tid; e < in * out; e += nthr) {   // flat order: o fastest (coalesced global reads)
      const int i = e / out, oo = e - i * out;
      lw[md.wl_off[l] + oo * ldw + i] = W[e];
    }
    for (int e = tid; e < out; e += nthr) lw[md.bl_off[l] + e] = a.Wflat[dm.b_off[l] + e];
  }
  for (int i = tid; i < T; i += nthr) s_ts[i] = a.ts[i];
  for (int i = tid; i < md.wpb * pwave; i += nthr) panels[i] = 0.f;
  __syncthreads();

  float* xp = panels + wave * pwave;
  const long long gw = (long long)blockIdx.x * md.wpb + wave;   // global wave = group of 4 columns
  const long long colg = 4 * gw + j;
  const bool valid = colg < B;
  const long long tile = gw >> 2;                                // 16-column staging slot row
  const int wq = (int)(gw & 3);
  const bool coupled = dm.coupled != 0;
  const double tT = s_ts[T - 1], dtmax = fabs(tT - s_ts[0]);
  unsigned gen = 0;
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 3;
  float* const my_stage = a.stage + (size_t)tile * a.cap * dm.blk_floats;
  float* const my_wts = a.wts + (size_t)tile * a.cap * NB + 4 * wq + j;
  int slot_base = 0;
  const bool rowok[4] = {4 * b + 0 < Dp, 4 * b + 1 < Dp, 4 * b + 2 < Dp, 4 * b + 3 < Dp};
  const float NREAL = (float)(2 * Dp + NP);

  // ---- terminal condition: z = ẑ(t_T), λ = Δ_T, g = 0
  auto load_rows = [&](const float* base, int jt) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const float* p = base + (size_t)Dp * ((size_t)colg + (size_t)B * jt) + 4 * b;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (rowok[i]) v[i] = p[i];
    }
    return v;
  };
  St4 y = st4_zero(), yn = st4_zero(), tmp = st4_zero(), k[7];
#pragma unroll
  for (int s = 0; s < 7; s++) k[s] = st4_zero();
  y.z = load_rows(a.z_out, T - 1);
  y.lam = load_rows(a.dz_out, T - 1);
  float bad = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) bad += isfinite(y.z[i]) ? 0.f : 1.f;
  bad = col_sum(bad);
  double t = tT, dt = 0.0, tnew = 0.0;
  float h = 0.f, qold = 1e-4f, wqd = 0.f, th1 = 0.f;
  // a failed forward trajectory is a constant NaN block ⇒ zero gradient  [REF GOKU.jl:114]
  // coupled control: padding columns shadow the shared step sequence (they are excluded from every sum and store)
  int status = !valid ? (coupled ? 0 : 1) : (bad > 0.f ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1));
  int jsave = T - 2, hit = 0, accepted = 0, nfe = 0, nacc = 0, nrej = 0;
  long long iters = 0;
  float L = 1.f;
  if (dm.has_pend && valid) L = a.theta[(size_t)colg * NP];
  const float ngl = -10.0f / L, gl2 = 10.0f / (L * L);
  if (status > 1) y = st4_zero();   // neutralise the NaN column
  bool overflow = false;

  // Σ over the columns that count, of up to 2 per-column values: per trajectory = the column's own value; coupled =
  // grid-wide total (workgroup partial through LDS, then the monotonic-counter barrier of lde_mlp.hip)
  auto norm2 = [&](float& v0, float& v1) {
    v0 = col_sum(v0);
    v1 = col_sum(v1);
    if (!coupled) return;
    const bool cnt = valid && status == 0;
    float w0 = wave_cols_sum(cnt ? v0 : 0.f), w1 = wave_cols_sum(cnt ? v1 : 0.f);
    if (lane == 0) { s_red[wave * 4 + 0] = w0; s_red[wave * 4 + 1] = w1; }
    __syncthreads();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid == 0)
      for (int w = 0; w < md.wpb; w++) { v[0] += s_red[w * 4 + 0]; v[1] += s_red[w * 4 + 1]; }
    grid_sum4(a.gs, gen, v, s_red + md.wpb * 4);
    v0 = v[0];
    v1 = v[1];
  };
  auto sumsq = [&](const St4& q) {   // Σ over this lane's real entries
    float s = q.g * q.g;
#pragma unroll
    for (int i = 0; i < 4; i++) s += rowok[i] ? q.z[i] * q.z[i] + q.lam[i] * q.lam[i] : 0.f;
    return s;
  };

  // start of a step attempt: iteration guard, clip to the next save time; false when every column of the wave (coupled:
  // of the grid — all columns share one step sequence there) has finished
  auto begin_step = [&]() -> bool {
    if (status == 0 && iters++ >= o.maxiters) status = 1 + LDE_RET_MAXITERS;
    if (status == 0 && slot_base + NST > a.cap) {   // staging area exhausted: flag it — k_mlp_adjoint redoes the call (and the host doubles the area for the next one)
      status = 1 + LDE_RET_MAXITERS;
      overflow = true;
    }
    if (status == 0) {
      const double tstop = s_ts[jsave];
      const double dist = t - tstop;
      double hmag = dt;
      hit = 0;
      if (hmag >= dist * (1.0 - 1e-12)) { hmag = dist; hit = 1; }
      tnew = hmag;
      h = -(float)hmag;
      wqd = (float)hmag;
    } else {
      h = 0.f;
      wqd = 0.f;
      hit = 0;
    }
    return __any(status == 0) != 0;
  };

  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  int phase = auto_dt ? PH_K0 : PH_STAGE, s = 0;
  bool running = T > 1;
  if (running && !auto_dt) {
    dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    running = begin_step();
  }
  while (running) {
    St4 src = y;
    bool any_w = false;
    if (phase == PH_INIT1) src = tmp;
    if (phase == PH_STAGE) {
      float bs;
      if (SOLVER == LDE_SOLVER_TSIT5) {
        if (s > 0) {
          St4 acc = st4_zero();
#pragma unroll
          for (int jj = 0; jj < 6; jj++)
            if (jj < s) acc = st4_axpy(acc, ts5::A[s][jj], k[jj]);
          src = st4_axpy(y, h, acc);
          if (s == 6) yn = src;
        }
        bs = s < 6 ? ts5::A[6][s] : 0.f;
        any_w = s < 6;
      } else {
        if (s > 0) {   // k[s-1] with a static register index
          St4 kp = k[0];
          if (s == 2) kp = k[1];
          if (s == 3) kp = k[2];
          src = st4_axpy(y, (s == 3 ? 1.0f : 0.5f) * h, kp);
        }
        bs = (s == 0 || s == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
        any_w = true;
      }
      if (any_w && b == 0) my_wts[(size_t)(slot_base + s) * NB] = (valid && status == 0) ? wqd * bs : 0.f;   // optimistic: zeroed on rejection
    }
    St4 dst;
    eval4<NTH>(dm, md, lw, xp, src, dst, ngl, gl2, any_w ? my_stage + (size_t)(slot_base + s) * dm.blk_floats : nullptr, wq);
    if (status == 0) nfe++;
    if (phase == PH_K0) k[0] = dst;
    else if (phase == PH_INIT1) k[1] = dst;
    else {
#pragma unroll
      for (int q = 0; q < 7; q++)
        if (q == s) k[q] = dst;
    }

    if (phase == PH_K0) {
      // Hairer–Nørsett–Wanner on the augmented state, direction −1: part 1
      St4 sk;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        sk.z[i] = fast_rcp(o.abstol + fabsf(y.z[i]) * o.reltol);
        sk.lam[i] = fast_rcp(o.abstol + fabsf(y.lam[i]) * o.reltol);
        if (rowok[i]) {
          const float a0 = y.z[i] * sk.z[i], a1 = k[0].z[i] * sk.z[i], c0 = y.lam[i] * sk.lam[i], c1 = k[0].lam[i] * sk.lam[i];
          s0 += a0 * a0 + c0 * c0;
          s1 += a1 * a1 + c1 * c1;
        }
      }
      sk.g = fast_rcp(o.abstol + fabsf(y.g) * o.reltol);
      if (NP && b == 0) { s0 += (y.g * sk.g) * (y.g * sk.g); s1 += (k[0].g * sk.g) * (k[0].g * sk.g); }
      norm2(s0, s1);
      const float n = coupled ? NREAL * (float)B : NREAL;
      const float d0 = sqrtf(s0 / n), d1 = sqrtf(s1 / n);
      double dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6 : 0.01 * (double)(d0 * fast_rcp(d1));
      if (dt0 > dtmax) dt0 = dtmax;
      dt = dt0;
      h = status == 0 ? -(float)dt0 : 0.f;
      th1 = d1;
      tmp = st4_axpy(y, h, k[0]);
      yn = sk;   // 1/scale, needed by part 2
      phase = PH_INIT1;
    } else if (phase == PH_INIT1) {
      float s2 = 0.f, dummy = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (rowok[i]) {
          const float dz = (k[1].z[i] - k[0].z[i]) * yn.z[i], dl = (k[1].lam[i] - k[0].lam[i]) * yn.lam[i];
          s2 += dz * dz + dl * dl;
        }
      if (NP && b == 0) { const float dg = (k[1].g - k[0].g) * yn.g; s2 += dg * dg; }
      norm2(s2, dummy);
      const float n = coupled ? NREAL * (float)B : NREAL;
      const double dt0 = dt;
      const float d2 = sqrtf(s2 / n) * fast_rcp((float)dt0);
      const float dm_ = fmaxf(th1, d2);
      const double dt1 = (dm_ <= 1e-15f) ? fmax(1e-6, dt0 * 1e-3) : (double)(0.39810717055349726f * fast_pow(dm_, -0.2f));
      double dtn = fmin(100.0 * dt0, dt1);
      dt = dtn > dtmax ? dtmax : dtn;
      phase = PH_STAGE;
      s = 0;
      running = begin_step();
    } else if (s < LAST_STAGE) {
      s++;
    } else {
      // ---- all stages of this attempt are done ---------------------------------------------------------------------
      if (SOLVER == LDE_SOLVER_RK4) {
        St4 acc = st4_axpy(k[0], 2.0f, k[1]);
        acc = st4_axpy(acc, 2.0f, k[2]);
        acc = st4_axpy(acc, 1.0f, k[3]);
        yn = st4_axpy(y, h * (1.0f / 6.0f), acc);
      }
      float e2 = 0.f, nf = 0.f;
      {
        St4 er = st4_zero();
        if (o.adaptive) {
#pragma unroll
          for (int jj = 0; jj < 7; jj++) er = st4_axpy(er, ts5::BT[jj], k[jj]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (rowok[i]) {
            const float rz = er.z[i] * h * fast_rcp(o.abstol + fmaxf(fabsf(y.z[i]), fabsf(yn.z[i])) * o.reltol);
            const float rl = er.lam[i] * h * fast_rcp(o.abstol + fmaxf(fabsf(y.lam[i]), fabsf(yn.lam[i])) * o.reltol);
            e2 += rz * rz + rl * rl;
            nf += (isfinite(yn.z[i]) && isfinite(yn.lam[i])) ? 0.f : 1.f;
          }
        if (NP && b == 0) {
          const float rg = er.g * h * fast_rcp(o.abstol + fmaxf(fabsf(y.g), fabsf(yn.g)) * o.reltol);
          e2 += rg * rg;
          nf += isfinite(yn.g) ? 0.f : 1.f;
        }
      }
      norm2(e2, nf);
      accepted = 0;
      if (status == 0) {
        const float n = coupled ? NREAL * (float)B : NREAL;
        const bool nonfinite = nf > 0.f || !(e2 == e2);
        const float EEst = o.adaptive ? sqrtf(e2 / n) : 0.f;
        const double hmag = tnew;
        if (nonfinite) {
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
            accepted = 1;
          }
        } else {
          dt = o.dt_fixed;
          accepted = 1;
        }
        if (accepted) nacc++;
      }
      // the attempt's staged evaluations: rejected columns do not contribute; if nothing was accepted the slots are reused
      if (b == 0 && !accepted && wqd != 0.f) {
#pragma unroll
        for (int st = 0; st < NST; st++) my_wts[(size_t)(slot_base + st) * NB] = 0.f;
      }
      if (__any(accepted && valid)) slot_base += NST;
      // ---- advance accepted columns; jump at a save time
      if (accepted) {
        y = yn;
        if (hit) {
          const f32x4 jl = load_rows(a.dz_out, jsave);
          y.lam += jl;
          if (o.checkpoint) {
            const f32x4 jz = load_rows(a.z_out, jsave);
            if (valid) y.z = jz;
          }
          t = s_ts[jsave];
          jsave--;
          if (jsave < 0) status = 1;
        } else
          t -= tnew;
      }
      s = 0;
      running = begin_step();
    }
  }

  // ---- results
  if (lane < 4) {
    atomicMax(&a.nslots[tile], slot_base);
    if (__any(overflow)) __hip_atomic_store(a.ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (valid) {
    const bool failed = status > 1;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (4 * b + i < D) a.dz0[(size_t)colg * D + 4 * b + i] = failed ? 0.f : y.lam[i];
    if (NP && b == 0) a.dtheta[(size_t)colg * NP] = failed ? 0.f : y.g;
    if (b == 0) {
      a.st_ret[colg] = failed ? status - 1 : 0;
      const bool rep = !coupled || colg == 0;
      a.st_nfe[colg] = rep ? nfe : 0;
      a.st_nacc[colg] = rep ? nacc : 0;
      a.st_nrej[colg] = rep ? nrej : 0;
    }
  }
}

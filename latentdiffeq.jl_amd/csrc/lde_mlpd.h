// lde_mlpd.h — LDE_SENSE_DISCRETE for MLP right-hand sides on the 16-column MFMA tiles (included by lde_mlp.hip behind k_mlp_adjoint:
// it uses that kernel's panels, eval_rhs / eval_bwd, the staging area and k_mlp_dw).
//
// What the reference's ForwardDiffSensitivity() differentiates [REF examples/pendulum_friction-less/pendulum.jl:11],
// [REF src/models/GOKU.jl:107, :121]: the discrete solve on its accepted step sequence, step sizes constant — here in reverse mode.
// lde_forward left (t_n, dt_n, y_n) per accepted step in the step record
// (lde_device.h: StepRec); a workgroup owns 16 trajectories for the whole sweep and walks THEIR steps from the last to the first:
//   pass 1  k_i = f(g_i), g_i = y_n + h Σ a_iq k_q, i = 1..S        (eval_rhs: forward only — the stage points are what is needed)
//   saves   every save time inside the step puts its cotangent Δ_j on y_n, on the slopes (h·b_i(Θ_j)) or on y_{n+1}
//   pass 2  Jᵀk̄ at y_{n+1} (the FSAL slope: it carries the next step's k̄₁), then at g_S … g_2, each with its (a_l, δ_l) panels staged
//           for k_mlp_dw at weight 1 (the scale h·b_i is inside k̄)       (eval_bwd with λ := k̄_i: the same fused evaluation the continuous
//           adjoint runs, 2S of them per accepted step instead of 6–7 per ATTEMPT of a reverse-time solve with forced stops)
// No controller, no error norm, no grid-wide sum: a coupled solve (one step sequence for the batch) needs no cooperative launch here.
// Per-trajectory control: the 16 columns sweep their own sequences in lock-step; a column that has reached its first step runs one
// pseudo-step (h = 0) that pulls the last k̄₁ through f(y_0), then idles with zero cotangents.

template <int SOLVER, int NT>
__global__ void __launch_bounds__(NT) k_mlp_adjoint_disc(MlpDims dm, KOpts o, BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int S = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                              {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
  auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
  const int T = o.T, B = o.B, Dp = dm.Dp, DpA = dm.DpA, D = dm.D, NP = dm.P;
  const StepRec R = o.rec;
  Ctl* c = reinterpret_cast<Ctl*>(smem);
  double* s_ts = reinterpret_cast<double*>(smem + ((sizeof(Ctl) + 15) & ~size_t(15)));
  float* base = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_ts) + (((size_t)T * 8 + 15) & ~size_t(15)));
  Panels P;
  P.lds = dm.ld_sb;
  P.ldh = dm.ld_h;
  P.pstride = NB * P.lds;
  P.hstride = NB * P.ldh;
  const int ldc = dm.ld_sf, cstride = NB * ldc;   // compact panels: the z rows only
  float* p = base;
  float* src = p; p += P.pstride;                 // work panels in the adjoint layout z | λ | g (what eval_bwd reads and writes)
  float* dst = p; p += P.pstride;
  float* yN = p; p += cstride;                    // y_n
  float* yN1 = p; p += cstride;                   // y_{n+1}
  float* Kc = p; p += S * cstride;                // k_1 … k_S
  float* KB = p; p += (S + 1) * cstride;          // k̄_1 … k̄_{S+1}; KB[S] carries the next step's k̄₁ into this one
  float* YB = p; p += cstride;                    // cotangent of y_{n+1}
  float* YBN = p; p += cstride;                   // cotangent reaching y_n
  float* gth = p; p += NB;                        // ∂L/∂θ per column
  P.y = P.yn = P.tmp = P.kbase = P.scr = nullptr;
  P.hidbase = p; p += dm.h_total;
  P.hoff = dm.h_off;
  P.delbase = p; p += 2 * P.hstride;
  P.red = p; p += (NT / 64) * 256;
  P.biasc = p; p += (dm.nbias + 3) & ~3;
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

  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * NB;
  const int lds = P.lds;
  const bool coupled = dm.coupled != 0;
  const double tend = s_ts[T - 1];
  float* const my_stage = a.stage + (size_t)blockIdx.x * a.cap * dm.blk_floats;
  float* const my_wts = a.wts + (size_t)blockIdx.x * a.cap * NB;
  int slot_base = 0, nflush = 0;
// the Dp×16 elements of a compact panel; `e_` → thread is the same map in every phase, so a thread re-reads only what it wrote
#define FOR_C(colv, rowv) for (int e_ = tid, colv = e_ / Dp, rowv = e_ - colv * Dp; e_ < NB * Dp; e_ += NT, colv = e_ / Dp, rowv = e_ - colv * Dp)

  // ---- per-column control: Ctl's fields, re-used — nacc = the step the column is at (−1: the pseudo-step, −2: done), iters = its step
  //      count, j = the last save time not yet served, last = 1 the step that ends the solve / 2 the pseudo-step
  if (tid < NB) {
    const int col = tid, b = b0 + col;
    const bool valid = b < B;
    const int seq = coupled ? 0 : (valid ? b : 0);
    const int ns = valid ? R.n[seq] : 0;
    bool bad = false;
    if (valid)
      for (int r = 0; r < Dp; r++) bad = bad || !isfinite(a.z_out[(size_t)b * Dp + r]);
    int st = 1;
    if (valid) {
      if (bad) st = 1 + LDE_RET_NONFINITE;                              // a failed forward trajectory: zero gradient [REF GOKU.jl:114]
      else if (T > 1 && (ns < 1 || ns > R.cap)) st = 1 + LDE_RET_MAXITERS;   // no usable record: NaN gradient, never a truncated sweep
      else if (T > 1) st = 0;
    }
    c->status[col] = st;
    c->iters[col] = ns;
    c->nacc[col] = ns - 1;
    c->j[col] = T - 1;
    c->nfe[col] = 0;
    c->h[col] = 0.f;
    float L = 1.f;
    if (dm.has_pend && valid) L = a.theta[(size_t)b * NP];
    c->ngl[col] = -10.0f / L;
    c->gl2[col] = 10.0f / (L * L);
  }
  __syncthreads();

  for (;;) {
    // ---- the step each column is at ----------------------------------------------------------------------------------------------
    if (tid < NB) {
      const int col = tid, b = b0 + col;
      const int seq = coupled ? 0 : (b < B ? b : 0);
      if (c->status[col] == 0) {
        const int cur = c->nacc[col], ns = c->iters[col];
        if (cur >= 0) {
          c->t[col] = R.t[(size_t)cur * R.nseq + seq];
          c->dt[col] = R.dt[(size_t)cur * R.nseq + seq];
          c->h[col] = (float)c->dt[col];
          c->last[col] = cur == ns - 1;
          c->tnew[col] = cur == ns - 1 ? tend : R.t[(size_t)(cur + 1) * R.nseq + seq];
        } else {   // the pseudo-step at y_0
          c->h[col] = 0.f;
          c->last[col] = 2;
        }
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
    if (!c->any_active) break;
    if (slot_base + S > a.cap) {   // staging area full: fold it into the private slab (exact, slow — the sizing makes it rare)
      flush_stage<NT>(dm, my_stage, my_wts, slot_base, a.slab + (size_t)blockIdx.x * dm.slab_n, nflush == 0);
      slot_base = 0;
      nflush++;
      __syncthreads();
    }
    FOR_C(col, row) {
      const int b = b0 + col, ci = col * ldc + row;
      if (c->status[col] == 0) {
        const int cur = c->nacc[col];
        yN[ci] = cur >= 0 ? R.y[((size_t)cur * B + b) * Dp + row] : a.z_out[(size_t)b * Dp + row];
      }
#pragma unroll
      for (int i = 0; i < S; i++) KB[i * cstride + ci] = 0.f;
      YBN[ci] = 0.f;
      src[col * lds + row] = yN[ci];
    }
    __syncthreads();
    // ---- pass 1: the slopes k_1 … k_S (forward evaluations only) -----------------------------------------------------------------------
#pragma unroll 1
    for (int i = 0; i < S; i++) {
      eval_rhs<NT>(dm, P, c, src, dst);
      FOR_C(col, row) {
        const int ci = col * ldc + row;
        const float h = c->h[col];
        Kc[i * cstride + ci] = dst[col * lds + row];
        if (i + 1 < S) {   // the next stage point
          float acc = 0.f;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            acc = ts5::A[i + 1][0] * Kc[ci];
            for (int q = 1; q <= i; q++) acc += ts5::A[i + 1][q] * Kc[q * cstride + ci];
          } else
            acc = (i + 1 == 3 ? 1.0f : 0.5f) * Kc[i * cstride + ci];
          src[col * lds + row] = yN[ci] + h * acc;
        } else {           // y_{n+1}, in the forward kernel's own form
          float yv;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            float acc2 = ts5::A[6][0] * Kc[ci];
#pragma unroll
            for (int q = 1; q < 6; q++) acc2 += ts5::A[6][q] * Kc[q * cstride + ci];
            yv = yN[ci] + h * acc2;
          } else
            yv = yN[ci] + (h * (1.0f / 6.0f)) * (Kc[ci] + 2.0f * (Kc[cstride + ci] + Kc[2 * cstride + ci]) + Kc[3 * cstride + ci]);
          yN1[ci] = yv;
        }
      }
      __syncthreads();
    }
    // ---- the save times inside the step (t, tnew]: their cotangents reach y_n, the slopes and y_{n+1} ---------------------------------------
    for (;;) {
      if (tid < NB) {
        const int col = tid;
        int sv = 0;
        if (c->status[col] == 0 && c->last[col] != 2 && c->j[col] >= 1 && s_ts[c->j[col]] > c->t[col]) {
          const int jj = c->j[col];
          const double tj = s_ts[jj];
          c->savej[col] = jj;
          c->th[col] = (tj >= c->tnew[col] || (jj == T - 1 && c->last[col] == 1)) ? 2.0f : (float)(tj - c->t[col]) * fast_rcp(c->h[col]);
          c->j[col] = jj - 1;
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
      FOR_C(col, row) {
        if (!c->hit[col]) continue;
        const int ci = col * ldc + row;
        const float th = c->th[col], h = c->h[col];
        const float dj = a.dz_out[(size_t)Dp * ((size_t)(b0 + col) + (size_t)B * c->savej[col]) + row];
        if (th > 1.5f) YB[ci] += dj;
        else if (SOLVER == LDE_SOLVER_TSIT5) {
          float bw[7];
          tsit5_interp_weights(th, bw);
          YBN[ci] += dj;
#pragma unroll
          for (int q = 0; q < 7; q++) KB[q * cstride + ci] += (h * bw[q]) * dj;
        } else {   // cubic Hermite on (y_n, k_1, y_{n+1}, f(y_{n+1}))
          const float om = 1.0f - th;
          const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
          const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
          YBN[ci] += h00 * dj;
          KB[ci] += (h10 * h) * dj;
          YB[ci] += h01 * dj;
          KB[S * cstride + ci] += (h11 * h) * dj;
        }
      }
      __syncthreads();
    }
    // ---- pass 2: Jᵀk̄ at y_{n+1}, then at g_S … g_2 ------------------------------------------------------------------------------------
#pragma unroll 1
    for (int i = S; i >= 1; i--) {
      FOR_C(col, row) {
        const int ci = col * ldc + row;
        float zv;
        if (i == S) zv = yN1[ci];
        else {
          const float h = c->h[col];
          float acc = 0.f;
          if (SOLVER == LDE_SOLVER_TSIT5) {
            acc = ts5::A[i][0] * Kc[ci];
            for (int q = 1; q < i; q++) acc += ts5::A[i][q] * Kc[q * cstride + ci];
          } else
            acc = (i == 3 ? 1.0f : 0.5f) * Kc[(i - 1) * cstride + ci];
          zv = yN[ci] + h * acc;
        }
        src[col * lds + row] = zv;
        src[col * lds + DpA + row] = KB[i * cstride + ci];
      }
      if (tid < NB) my_wts[(size_t)(slot_base + (S - i)) * NB + tid] = (b0 + tid < B && c->status[tid] == 0) ? 1.f : 0.f;
      __syncthreads();
      eval_bwd<NT>(dm, P, c, src, dst, my_stage + (size_t)(slot_base + (S - i)) * dm.blk_floats);
      // dst: −Jᵀk̄ in the λ rows, −(∂f/∂θ)ᵀk̄ in the g row
      if (NP && tid < NB && c->status[tid] == 0) gth[tid] -= dst[tid * lds + 2 * DpA];
      FOR_C(col, row) {
        const int ci = col * ldc + row;
        const float h = c->h[col];
        const float v = -dst[col * lds + DpA + row];
        if (i == S) {
          const float yb = YB[ci] + v;
          YB[ci] = yb;
#pragma unroll
          for (int q = 0; q < S; q++) KB[q * cstride + ci] += (h * A(S, q)) * yb;   // y_{n+1} = y_n + h Σ b_q k_q
          YBN[ci] += yb;
        } else {
          YBN[ci] += v;
          for (int q = 0; q < i; q++) {
            const float aq = A(i, q);
            if (aq != 0.f) KB[q * cstride + ci] += (h * aq) * v;
          }
        }
      }
      __syncthreads();
    }
    slot_base += S;
    // ---- the step is done: k̄_1 travels on, ȳ_n becomes ȳ_{n+1} ------------------------------------------------------------------------------
    FOR_C(col, row) {
      const int ci = col * ldc + row;
      const bool act = c->status[col] == 0, pseudo = c->last[col] == 2;
      if (act) {
        KB[S * cstride + ci] = pseudo ? 0.f : KB[ci];
        YB[ci] = YBN[ci];
      }
    }
    __syncthreads();
    if (tid < NB && c->status[tid] == 0) {
      const int col = tid;
      c->nfe[col] += 2 * S;
      if (c->last[col] == 2) c->status[col] = 1;
      else c->nacc[col]--;
    }
    __syncthreads();
  }

  if (tid == 0) {
    a.nslots[blockIdx.x] = slot_base;
    a.nflush[blockIdx.x] = nflush;
  }
  // ---- results: save time 0 is ẑ₀ itself ------------------------------------------------------------------------------------------------------
  const float qn = __int_as_float(0x7fc00000);
  for (int e = tid; e < NB * D; e += NT) {
    const int col = e / D, row = e % D, b = b0 + col;
    if (b < B) {
      const int st = c->status[col];
      a.dz0[(size_t)b * D + row] = st == 1 ? YB[col * ldc + row] + a.dz_out[(size_t)b * Dp + row] : (st == 1 + LDE_RET_MAXITERS ? qn : 0.f);
    }
  }
  if (NP) {
    for (int e = tid; e < NB * NP; e += NT) {
      const int col = e / NP, row = e % NP, b = b0 + col;
      if (b < B) {
        const int st = c->status[col];
        a.dtheta[(size_t)b * NP + row] = st == 1 ? gth[col] : (st == 1 + LDE_RET_MAXITERS ? qn : 0.f);
      }
    }
  }
  if (tid < NB && b0 + tid < B) {
    const int col = tid, b = b0 + col;
    a.st_ret[b] = c->status[col] > 1 ? c->status[col] - 1 : 0;
    const bool rep = !coupled || b == 0;
    const int ns = c->status[col] == 1 && T > 1 ? c->iters[col] : 0;
    a.st_nfe[b] = rep ? c->nfe[col] : 0;
    a.st_nacc[b] = rep ? ns : 0;
    a.st_nrej[b] = 0;
  }
#undef FOR_C
}

static size_t disc_lds_fixed(const MlpDims& dm, int T, int nt) {
  const int NW = nt / 64, S1 = 7;
  size_t b = (sizeof(Ctl) + 15) & ~size_t(15);
  b += ((size_t)T * 8 + 15) & ~size_t(15);
  b += (size_t)(2 * NB * dm.ld_sb + (2 + 6 + S1 + 2) * NB * dm.ld_sf + NB + dm.h_total + 2 * NB * dm.ld_h + NW * 256 + ((dm.nbias + 3) & ~3)) * sizeof(float);
  return b;
}

// lde_mlpw.h — three-layer networks wider than one wave at SMALL batches: W waves per trajectory, weights AND solver state in
// registers (included by lde_mlp.hip after lde_mlp64.h).
//
// BASELINE.json configs[1] (LatentODE 8-200-200-8, RK4, B = 256) and configs[3] (32-128-128-32 relu, Tsit5, coupled control,
// B = 512 per GPU) are a few hundred sequential solves of ≈ 200–350 right-hand-side evaluations each. In k_mlpv (lanes =
// hidden units, vectors and state in LDS, weights in LDS / streamed from L2) an evaluation of the adjoint costs ≈ 16–22 k
// cycles: every layer is a chain of LDS round trips (layer record, x reads, weight reads, split-K partial sums, epilogue,
// barrier), the stage sums and the staging of (a_l, δ_l) walk LDS again, and the hidden×hidden weights of c2 do not fit LDS at
// all. Here the register file does the work — a CU has 512 KB of it against 160 KB of LDS:
//   * a workgroup of W = 2 or 4 waves owns one trajectory; lane `lane` of wave w owns hidden unit u = 64·w + lane of BOTH
//     hidden layers and keeps, in VGPRs (AGPRs beyond 256), its row of W₁, its row AND its column of W₂ and its column of W₃ —
//     c4: 320 registers, c2: 416 — loaded once from a pre-packed copy (k_build_wpack: every load coalesced);
//   * the solver state [z | λ] lives in lanes 0 … 2·DP−1 of EVERY wave (each wave carries the whole state redundantly, so
//     stage sums, error norms and the step controller need no communication: a stage sum is seven FMAs, a norm is one DPP
//     wave sum, and every wave takes bitwise the same decisions);
//   * a hidden vector crosses the waves through a 1 KB LDS buffer: one ds_write per lane, one barrier, then H/4 broadcast
//     ds_read_b128 feed H FMAs against the lane's register row — no weight traffic of any kind during the solve;
//   * the two narrow products (H → D′: f = W₃h₂ and vz = W₁ᵀδ₁) run with lanes = (K-segment, output) on slices kept in LDS
//     (16 KB each for c4), partial sums meet by xor-shuffles, every wave computes all D′ outputs for its own copy of the state;
//   * the (a_l, δ_l) panels of the weight gradient are stored straight from the registers that hold them.
// Same algorithm and control arithmetic as k_mlp_adjoint / k_mlpv / k_mlp64 (HNW initial step, PI controller carried across the
// save times, quadrature weights at accept, k_mlp_dw forms dW; coupled control sums over the grid in a fixed order):
// agreement to solver tolerance. Limits: exactly three Dense layers D′ → H₁ → H₂ → D′ with 2·D′ ≤ 64, H ≤ 200, no analytic
// part (P = 0), B·W ≤ 1024 waves (coupled) so that all of them are resident. Anything else: k_mlp64 / k_mlpv / the tiles.

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WDims {
  int DP, HP, W;          // lanes per state half (8 or 32), padded hidden width (floats per register row), waves per trajectory
  int UT;                 // 64·W
  int SEG, GS;            // narrow products: K-segments (64/DP), float4 groups per lane
  int HX;                 // floats of one exchange buffer (≥ UT and ≥ SEG·GS·4)
  // float offsets in the packed array
  int o_w1r, o_w2r, o_w2c, o_w3c, o_b1, o_b2, o_b3, o_n3, o_n1, total;
};

// one-time packing (set_weights): everything in the order the kernel's lanes read it
static __global__ void k_build_wpack(const float* __restrict__ Wflat, MlpDims dm, WDims wd, float* __restrict__ wp) {
  const int H1 = dm.sizes[1], H2 = dm.sizes[2], Dp = dm.Dp, UT = wd.UT;
  const float *W1 = Wflat + dm.w_off[0], *W2 = Wflat + dm.w_off[1], *W3 = Wflat + dm.w_off[2];   // column-major [out×in]
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < wd.total; e += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (e < wd.o_w2r) {            // w1r[k][u] = W₁(u, k)
      const int k = (e - wd.o_w1r) / UT, u = (e - wd.o_w1r) % UT;
      if (u < H1 && k < Dp) v = W1[u + H1 * k];
    } else if (e < wd.o_w2c) {     // w2r[k][u] = W₂(u, k)
      const int k = (e - wd.o_w2r) / UT, u = (e - wd.o_w2r) % UT;
      if (u < H2 && k < H1) v = W2[u + H2 * k];
    } else if (e < wd.o_w3c) {     // w2c[i][u] = W₂(i, u)
      const int i = (e - wd.o_w2c) / UT, u = (e - wd.o_w2c) % UT;
      if (u < H1 && i < H2) v = W2[i + H2 * u];
    } else if (e < wd.o_b1) {      // w3c[d][u] = W₃(d, u)
      const int d = (e - wd.o_w3c) / UT, u = (e - wd.o_w3c) % UT;
      if (u < H2 && d < Dp) v = W3[d + Dp * u];
    } else if (e < wd.o_b2) {
      const int u = e - wd.o_b1;
      if (u < H1) v = Wflat[dm.b_off[0] + u];
    } else if (e < wd.o_b3) {
      const int u = e - wd.o_b2;
      if (u < H2) v = Wflat[dm.b_off[1] + u];
    } else if (e < wd.o_n3) {
      const int d = (e - wd.o_b3) % wd.DP;
      if (d < Dp) v = Wflat[dm.b_off[2] + d];
    } else {                       // narrow slices: [g][lane][4]; lane = seg·DP + d, k = (seg·GS + g)·4 + c
      const bool n1 = e >= wd.o_n1;
      const int r = e - (n1 ? wd.o_n1 : wd.o_n3);
      const int c = r & 3, ln = (r >> 2) & 63, g = r >> 8;
      const int seg = ln / wd.DP, d = ln % wd.DP, k = (seg * wd.GS + g) * 4 + c;
      if (!n1) { if (d < Dp && k < H2) v = W3[d + Dp * k]; }      // f_d  = Σ_k W₃(d, k) h₂_k
      else     { if (d < Dp && k < H1) v = W1[k + H1 * d]; }      // vz_d = Σ_k W₁(k, d) δ₁_k
    }
    wp[e] = v;
  }
}

// Grid-wide sum of (v0[, v1]) for workgroups whose waves all hold the same values — ONE device-scope round trip instead of
// grid_sum4's three (publish, arrive on a counter, read back; ≈ 8 µs on MI355X's eight L2s): every workgroup publishes
// {value, tag} as ONE 8-byte word per value, tag = (launch epoch, generation), and every WAVE polls all the words until each
// carries the current tag, then adds the values in index order (lane w mod 64 ascending, DPP wave sum) — identical bits in every
// wave of every workgroup, no counter, no LDS reduction. Two buffers by generation parity: a workgroup publishes generation
// g + 2 only after it has read all of g + 1, which every workgroup wrote after it finished reading g. Stale words of earlier
// launches carry another epoch (the words live in a buffer of their own, zeroed when allocated and when the 16-bit epoch wraps).
// the cross-rank stage of a sum (LDE_BATCH_COUPLED_GLOBAL): every wave holds the same device-wide (v0, v1); workgroup 0 hands them to
// the host, which adds the other ranks' values through the caller's hook, and republishes the answer in device memory
template <bool TWO>
__device__ __forceinline__ void w_host_sum(const GridSync& gs, unsigned tag, float& v0, float& v1) {
  if (!gs.cross()) return;
  bool aborted = false;
  if (gs.nranks > 0) {
    // device to device: this rank's words go into slot `rank` of EVERY rank's mailbox (its own included), then the nranks words of its own
    // mailbox are awaited and added in rank order. Word sets by (launch parity, sum parity): a rank writes sum g + 2 only after it has read all
    // of g + 1, which every rank wrote after reading g; the first sums of the NEXT launch use the other launch parity — the parity of
    // gs.xlaunch, the exchange's own launch counter, which advances by exactly one per exchanging launch on every rank (the grid words'
    // epoch skips 0 when it wraps and restarts when the words are reallocated). Every rank must make the same sequence of calls on its
    // handle (the tag = (exchange launch count, sum count) is compared across ranks). System scope: the words cross xGMI
    // (fine-grained, peer-mapped memory).
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      // the words' tag: (the exchange's own launch count, this launch's sum count) — the same on every rank by the contract above
      const unsigned xt = (gs.xlaunch << 16) | (tag & 0xffffu);
      const size_t par = (size_t)(((gs.xlaunch & 1u) << 1) | (xt & 1u)) * gs.nranks * 2;   // four word sets: (launch parity, sum parity)
      const unsigned long long w0 = ((unsigned long long)xt << 32) | __float_as_uint(v0), w1 = ((unsigned long long)xt << 32) | __float_as_uint(TWO ? v1 : 0.f);
      for (int r = 0; r < gs.nranks; r++) {
        unsigned long long* box = gs.peer[r] + par + (size_t)gs.rank * 2;
        __hip_atomic_store(box + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(box + 0, w0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      float t0 = 0.f, t1 = 0.f;
      bool bad = false;
      const unsigned long long* mine = gs.peer[gs.rank] + par;
      const long long spin_max = (long long)gs.xspin_k * 1024;
      for (int r = 0; r < gs.nranks && !bad; r++) {
        unsigned long long q0 = 0, q1 = 0;
        long long spins = 0;
        for (;;) {
          q0 = __hip_atomic_load(mine + (size_t)r * 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          q1 = __hip_atomic_load(mine + (size_t)r * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((unsigned)(q0 >> 32) == xt && (unsigned)(q1 >> 32) == xt) break;
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 1023) == 0 && (spins >= spin_max || __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a peer rank is gone: poison the sums instead of hanging
            bad = true;
            break;
          }
        }
        t0 += __uint_as_float((unsigned)q0);
        t1 += __uint_as_float((unsigned)q1);
      }
      const unsigned nanb = 0x7fc00000u;
      __hip_atomic_store(gs.dev_rep + 1, ((unsigned long long)tag << 32) | (bad ? nanb : __float_as_uint(t1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gs.dev_rep + 0, ((unsigned long long)tag << 32) | (bad ? nanb : __float_as_uint(t0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(gs.host_req + 0, ((unsigned long long)tag << 32) | __float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(gs.host_req + 1, ((unsigned long long)tag << 32) | __float_as_uint(TWO ? v1 : 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(gs.host_req + 2, ((unsigned long long)tag << 32) | (TWO ? 2u : 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long r0 = 0, r1 = 0;
    long long spins = 0;
    for (;;) {
      r0 = __hip_atomic_load(gs.host_rep + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      r1 = __hip_atomic_load(gs.host_rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned)(r0 >> 32) == tag && (unsigned)(r1 >> 32) == tag) break;
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 1023) == 0 && (spins > 8000000LL || __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        // the host does not answer (a peer rank is gone, the hook failed): poison the sums instead of hanging
        __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r0 = ((unsigned long long)tag << 32) | 0x7fc00000u;
        r1 = r0;
        break;
      }
    }
    __hip_atomic_store(gs.dev_rep + 1, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gs.dev_rep + 0, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long q0 = 0, q1 = 0;
  long long spins = 0;
  for (;;) {
    q0 = __hip_atomic_load(gs.dev_rep + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q1 = __hip_atomic_load(gs.dev_rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(q0 >> 32) == tag && (unsigned)(q1 >> 32) == tag) break;
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 4095) == 0 && (spins > 40000000LL || __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      aborted = true;
      break;
    }
  }
  aborted = __syncthreads_or(aborted ? 1 : 0) != 0;
  v0 = aborted ? __int_as_float(0x7fc00000) : __uint_as_float((unsigned)q0);
  if (TWO) v1 = aborted ? __int_as_float(0x7fc00000) : __uint_as_float((unsigned)q1);
}

// The two halves of a sum, for a caller with work to do between them (k_mlpb / k_mlpc: the next attempt's first evaluation runs while
// the other workgroups' words arrive): publish this workgroup's words — collect all of them. w_grid_sum = one behind the other.
template <bool TWO>
__device__ __forceinline__ void w_grid_publish(const GridSync& gs, unsigned& gen, unsigned epoch, float v0, float v1) {
  if (gs.nwg == 1 && !gs.cross()) return;
  gen++;
  if (gs.nwg == 1) return;
  const unsigned tag = (epoch << 16) + gen;
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(gs.slots) + (size_t)(gen & 1) * gs.nwg * 2;
  if (threadIdx.x == 0) {
    __hip_atomic_store(slots + (size_t)blockIdx.x * 2, ((unsigned long long)tag << 32) | __float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (TWO)
      __hip_atomic_store(slots + (size_t)blockIdx.x * 2 + 1, ((unsigned long long)tag << 32) | __float_as_uint(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool TWO>
__device__ __forceinline__ void w_grid_collect(const GridSync& gs, unsigned gen, unsigned epoch, float& v0, float& v1) {   // `gen` as w_grid_publish left it
  if (gs.nwg == 1 && !gs.cross()) return;
  PROF_T(g0);
  const unsigned tag = (epoch << 16) + gen;
  if (gs.nwg == 1) {
    w_host_sum<TWO>(gs, tag, v0, v1);
    return;
  }
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(gs.slots) + (size_t)(gen & 1) * gs.nwg * 2;
  // Every wave of the workgroup polls ITS share of the words — word w belongs to thread w mod blockDim — so that a lane has one word and the
  // sum is ONE round trip to the memory side (agent-scope loads do not stop at this XCD's L2). (Round 4 had one wave poll all words, a lane's
  // four one after the other: 3.2 µs per sum with 256 workgroups against 1.8 µs with 64 — abl: -DLDE_PROF=1, slot 12; every wave polling ALL
  // words, before that, was four times the traffic in front of the words still on their way.) Added up in a fixed order — a wave's words by
  // its fixed tree, then the waves' sums in wave order: the same bits in every workgroup.
  __shared__ float s_gpart[2][8][2];   // by generation parity: the sum after the next one rewrites a word, and this workgroup's barrier of the next sum lies between
  float p0 = 0.f, p1 = 0.f;
  bool aborted = false;
  // (up to 64 workgroups one wave has a word per lane already: the other waves go straight to the barrier, as in round 4)
  const int gpoll = gs.nwg <= 64 ? 64 : (int)blockDim.x;
  const int gwv = threadIdx.x >> 6, gnw = (gpoll + 63) >> 6;
  for (int w = threadIdx.x < gpoll ? (int)threadIdx.x : gs.nwg; w < gs.nwg; w += gpoll) {
    unsigned long long q0, q1 = 0;
    long long spins = 0;
    for (;;) {
      q0 = __hip_atomic_load(slots + (size_t)w * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (TWO) q1 = __hip_atomic_load(slots + (size_t)w * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(q0 >> 32) == tag && (!TWO || (unsigned)(q1 >> 32) == tag)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 4095) == 0 &&
          (spins > 20000000LL || __hip_atomic_load(gs.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        // a peer is not resident (the launch is cooperative: cannot happen) — give up instead of hanging
        __hip_atomic_store(gs.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        aborted = true;
        break;
      }
    }
    p0 += __uint_as_float((unsigned)q0);
    p1 += __uint_as_float((unsigned)q1);
  }
  p0 = wave_sum64(p0);
  if (TWO) p1 = wave_sum64(p1);
  if ((threadIdx.x & 63) == 0 && gwv < gnw) { s_gpart[gen & 1][gwv][0] = p0; s_gpart[gen & 1][gwv][1] = p1; }
  aborted = __syncthreads_or(aborted ? 1 : 0) != 0;   // the waves of a workgroup must take the same decision
  float t0 = s_gpart[gen & 1][0][0], t1 = s_gpart[gen & 1][0][1];
  for (int i = 1; i < gnw; i++) { t0 += s_gpart[gen & 1][i][0]; t1 += s_gpart[gen & 1][i][1]; }
  v0 = aborted ? __int_as_float(0x7fc00000) : t0;   // a timed-out barrier poisons the sums: retcode != 0
  if (TWO) v1 = aborted ? __int_as_float(0x7fc00000) : t1;
  w_host_sum<TWO>(gs, tag, v0, v1);
  PROF_T(g1);
  PROF_ADD(12, g0, g1);
  PROF_ADD(21, g1 - 1, g1);
}
template <bool TWO>
__device__ __forceinline__ void w_grid_sum(const GridSync& gs, unsigned& gen, unsigned epoch, float& v0, float& v1) {
  w_grid_publish<TWO>(gs, gen, epoch, v0, v1);
  w_grid_collect<TWO>(gs, gen, epoch, v0, v1);
}

template <int SOLVER, int DP, int HP, int W, bool ADJ>
__global__ void __launch_bounds__(64 * W, 1) k_mlpw(MlpDims dm, WDims wd, KOpts o, VArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int UT = 64 * W, SEG = 64 / DP, G1 = DP / 4, GH = HP / 4, GS = (HP / SEG + 3) / 4;   // GS: host = wd.GS
  static_assert(HP % 4 == 0 && HP <= UT && 2 * DP <= 64, "k_mlpw geometry");
  const int T = o.T, B = o.B, D = dm.D, Dp = dm.Dp, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
  const int u = tid;                      // the hidden unit this lane owns
  const int H1 = dm.sizes[1], H2 = dm.sizes[2], HX = wd.HX;
  // ---- LDS: save times | per-wave state copies | two exchange buffers | narrow slices | flag
  double* s_ts = reinterpret_cast<double*>(smem);
  float* fbase = reinterpret_cast<float*>(smem + (((size_t)T * 8 + 15) & ~size_t(15)));
  float* xs = fbase + wv * 64;            // this wave's copy of the evaluation's input
  float* hxA = fbase + W * 64;
  float* hxB = hxA + HX;
  f32x4* n3 = reinterpret_cast<f32x4*>(hxB + HX);
  f32x4* n1 = n3 + GS * 64;
  float* s_cot = reinterpret_cast<float*>(n1 + (ADJ ? GS * 64 : 0));   // adjoint: the trajectory's cotangents (and saved states) by save time
  for (int i = tid; i < T; i += UT) s_ts[i] = a.ts[i];
  for (int i = tid; i < W * 64 + 2 * HX; i += UT) fbase[i] = 0.f;
  {
    const f32x4* g3 = reinterpret_cast<const f32x4*>(a.wpack + wd.o_n3);
    for (int i = tid; i < GS * 64; i += UT) n3[i] = g3[i];
    if (ADJ) {
      const f32x4* g1 = reinterpret_cast<const f32x4*>(a.wpack + wd.o_n1);
      for (int i = tid; i < GS * 64; i += UT) n1[i] = g1[i];
    }
    if (ADJ && a.cot_lds) {   // [T][Dp] dẑ (+ [T][Dp] ẑ when the adjoint restarts from the saved states): no global load inside the solve
      for (int i = tid; i < T * Dp; i += UT) {
        const size_t g = (size_t)Dp * ((size_t)b + (size_t)B * (i / Dp)) + (i % Dp);
        s_cot[i] = a.dz_out[g];
        if (o.checkpoint) s_cot[T * Dp + i] = a.z_out[g];
      }
    }
  }
  // ---- registers: the lane's rows / columns
  // (register PAIRS: a product is v_pk_fma_f32 on (w_k, w_k+1)·(x_k, x_k+1) — half the issue slots of scalar FMAs)
  f32x2 w1r[DP / 2], w2r[HP / 2], w2c[ADJ ? HP / 2 : 1], w3c[ADJ ? DP / 2 : 1];
  {
    const float* wp = a.wpack;
#pragma unroll
    for (int k = 0; k < DP / 2; k++) w1r[k] = f32x2{wp[wd.o_w1r + (2 * k) * UT + u], wp[wd.o_w1r + (2 * k + 1) * UT + u]};
#pragma unroll
    for (int k = 0; k < HP / 2; k++) w2r[k] = f32x2{wp[wd.o_w2r + (2 * k) * UT + u], wp[wd.o_w2r + (2 * k + 1) * UT + u]};
    if (ADJ) {
#pragma unroll
      for (int i = 0; i < HP / 2; i++) w2c[i] = f32x2{wp[wd.o_w2c + (2 * i) * UT + u], wp[wd.o_w2c + (2 * i + 1) * UT + u]};
#pragma unroll
      for (int d = 0; d < DP / 2; d++) w3c[d] = f32x2{wp[wd.o_w3c + (2 * d) * UT + u], wp[wd.o_w3c + (2 * d + 1) * UT + u]};
    }
  }
  const float b1 = a.wpack[wd.o_b1 + u], b2 = a.wpack[wd.o_b2 + u], b3 = a.wpack[wd.o_b3 + (lane % DP)];
  const int act = dm.act;
  __syncthreads();

  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = fabs(tend - t0);
  unsigned gen = 0;
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;
  const int tile = b >> 4, ncol = b & 15;
  float* const my_stage = ADJ ? a.stage + (size_t)tile * a.cap * dm.blk_floats : nullptr;
  float* const my_wts = ADJ ? a.wts + (size_t)tile * a.cap * NB : nullptr;
  int slot_base = 0;
  bool overflow = false;
  const int blk_floats = dm.blk_floats;
  const int in32_0 = pad32(Dp), h1_32 = pad32(H1), h2_32 = pad32(H2);
  const int boff0 = dm.blk_off[0], boff1 = dm.blk_off[1], boff2 = dm.blk_off[2];

  // ---- state: lane i < DP holds z_i, lane DP + i holds λ_i (adjoint); the other lanes stay 0
  const bool is_z = lane < Dp, is_l = ADJ && lane >= DP && lane < DP + Dp;
  const bool counted = is_z || is_l;
  const int row = is_l ? lane - DP : lane;
  float y = 0.f, yn = 0.f, tmp = 0.f, scr = 0.f, k[7];
#pragma unroll
  for (int s = 0; s < 7; s++) k[s] = 0.f;
  if (!ADJ) {
    if (lane < D) y = a.z0[(size_t)b * D + lane];
  } else if (counted) {
    const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * (T - 1)) + row;
    y = is_z ? a.z_out[srcg] : a.dz_out[srcg];
  }
  double t = ADJ ? tend : t0, dt = 0.0, tnew = 0.0;
  float h = 0.f, qold = 1e-4f, wq = 0.f, d1n = 0.f;
  int status, j = ADJ ? T - 2 : 1, last = 0, hit = 0, nfe = 0, nacc = 0, nrej = 0;
  long long iters = 0;
  {
    // a failed forward trajectory is a constant NaN block ⇒ zero gradient  [REF GOKU.jl:114]
    const bool bad = ADJ && __any(is_z && !isfinite(y));
    status = bad ? 1 + LDE_RET_NONFINITE : (T > 1 ? 0 : 1);
    if (ADJ && bad) y = 0.f;
  }
  if (!ADJ && wv == 0 && lane < Dp) a.z_out[(size_t)b * Dp + lane] = y;   // save time 0 = ẑ₀ itself (augmented rows 0)

  enum { PH_K0 = 0, PH_INIT1 = 1, PH_STAGE = 2 };
  constexpr int LAST_STAGE = SOLVER == LDE_SOLVER_TSIT5 ? 6 : (ADJ ? 3 : 4);
  const float dirn = ADJ ? -1.f : 1.f;
  const float nnorm = (float)(ADJ ? 2 * Dp : Dp) * (coupled ? (float)(a.Bnorm > 0 ? a.Bnorm : B) : 1.f);   // Bnorm: the batch over ALL ranks (LDE_BATCH_COUPLED_GLOBAL)

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
  // (coupled control: the status is a function of the shared sums, so every workgroup leaves the loop at the same step)
  auto xor_segs = [&](float p) -> float {
#pragma unroll
    for (int m = DP; m < 64; m <<= 1) p += __shfl_xor(p, m);
    return p;
  };

  // one evaluation of the (augmented) right-hand side: src → dst, staged into `blk` when given
  auto eval = [&](float src, float* blk) -> float {
    PROF_T(e0);
    xs[lane] = src;
    asm volatile("" ::: "memory");   // same wave, in-order LDS: the broadcast reads below see the write (no barrier needed)
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
    float a1 = b1;
    {
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G1; g++) {
        const f32x4 xv = x4[g];
        c01 += w1r[2 * g] * xv.lo;
        c23 += w1r[2 * g + 1] * xv.hi;
      }
      a1 += (c01.x + c01.y) + (c23.x + c23.y);
    }
    const float h1 = act_fn(act, a1);
    hxA[u] = h1;
    __syncthreads();
    float a2 = b2;
    {
      const f32x4* hv = reinterpret_cast<const f32x4*>(hxA);
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < GH; g++) {
        const f32x4 xv = hv[g];
        c01 += w2r[2 * g] * xv.lo;
        c23 += w2r[2 * g + 1] * xv.hi;
      }
      a2 += (c01.x + c01.y) + (c23.x + c23.y);
    }
    const float h2 = act_fn(act, a2);
    hxB[u] = h2;
    __syncthreads();
    PROF_T(e1);
    float f;
    {
      const f32x4* hv = reinterpret_cast<const f32x4*>(hxB) + (lane / DP) * GS;
      f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < GS; g++) {
        const f32x4 wq4 = n3[g * 64 + lane], xv = hv[g];
        p01 += wq4.lo * xv.lo;
        p23 += wq4.hi * xv.hi;
      }
      f = xor_segs((p01.x + p01.y) + (p23.x + p23.y)) + b3;
    }
    float dst = is_z ? f : 0.f;
    PROF_T(e2);
    PROF_ADD(3, e0, e1);
    PROF_ADD(4, e1, e2);
    if (ADJ) {
      float g2;
      {
        f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G1; g++) {
          const f32x4 xv = x4[DP / 4 + g];   // λ
          c01 += w3c[2 * g] * xv.lo;
          c23 += w3c[2 * g + 1] * xv.hi;
        }
        g2 = (c01.x + c01.y) + (c23.x + c23.y);
      }
      const float d2 = g2 * act_grad(act, h2);
      hxA[u] = d2;
      __syncthreads();
      float d1;
      {
        const f32x4* hv = reinterpret_cast<const f32x4*>(hxA);
        f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < GH; g++) {
          const f32x4 xv = hv[g];
          c01 += w2c[2 * g] * xv.lo;
          c23 += w2c[2 * g + 1] * xv.hi;
        }
        d1 = ((c01.x + c01.y) + (c23.x + c23.y)) * act_grad(act, h1);
      }
      hxB[u] = d1;
      __syncthreads();
      {
        const f32x4* hv = reinterpret_cast<const f32x4*>(hxB) + (lane / DP) * GS;
        f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < GS; g++) {
          const f32x4 wq4 = n1[g * 64 + lane], xv = hv[g];
          p01 += wq4.lo * xv.lo;
          p23 += wq4.hi * xv.hi;
        }
        const float vz = xor_segs((p01.x + p01.y) + (p23.x + p23.y));     // every lane with lane % DP == d holds vz_d
        if (is_l) dst = -vz;
      }
      PROF_T(e3);
      PROF_ADD(5, e2, e3);
      if (blk) {   // (a_l, δ_l) of the three layers, column ncol of the tile's slot; rows beyond the layer are zeros
        if (wv == 0 && lane < in32_0) {
          blk[boff0 + ncol * in32_0 + lane] = lane < Dp ? src : 0.f;                                 // a₀ = z
          blk[boff2 + NB * h2_32 + ncol * in32_0 + lane] = lane < Dp ? xs[DP + lane] : 0.f;          // δ₃ = λ
        }
        if (u < h1_32) {
          blk[boff0 + NB * in32_0 + ncol * h1_32 + u] = d1;                                          // δ₁ (0 beyond H₁: zero weights)
          blk[boff1 + ncol * h1_32 + u] = u < H1 ? h1 : 0.f;                                         // a₁ = h₁
        }
        if (u < h2_32) {
          blk[boff1 + NB * h1_32 + ncol * h2_32 + u] = d2;                                           // δ₂
          blk[boff2 + ncol * h2_32 + u] = u < H2 ? h2 : 0.f;                                         // a₂ = h₂
        }
      }
      PROF_T(e4);
      PROF_ADD(6, e3, e4);
    }
    PROF_T(e5);
    PROF_ADD(1, e0, e5);
    PROF_ADD(20, e5 - 1, e5);
    return dst;
  };

  const bool auto_dt = o.adaptive && !(o.dt_fixed > 0);
  int phase = (ADJ && !auto_dt) ? PH_STAGE : PH_K0, s = 0;
  bool running = T > 1 && status == 0;
  if (ADJ && running && !auto_dt) {
    dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
    running = begin_step();
  }
  while (running) {
    PROF_T(l0);
#if LDE_PROF
    struct ProfEnd { long long t0; __device__ ~ProfEnd() { PROF_T(t1); PROF_ADD(11, t0, t1); } } prof_end{l0};
#endif
    float src = phase == PH_INIT1 ? tmp : y;
    bool any_w = false;
    if (phase == PH_STAGE) {
      if (SOLVER == LDE_SOLVER_TSIT5) {
        if (s > 0) {
#define WSTAGE(S_)                                                                 \
  case S_: {                                                                       \
    float accv = ts5::A[S_][0] * k[0];                                             \
    _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += ts5::A[S_][jj] * k[jj]; \
    src = y + h * accv;                                                            \
  } break;
          switch (s) {
            WSTAGE(1) WSTAGE(2) WSTAGE(3) WSTAGE(4) WSTAGE(5) WSTAGE(6)
            default: break;
          }
#undef WSTAGE
          if (s == 6) yn = src;
        }
        any_w = ADJ && s < 6;
      } else if (ADJ || s < 4) {
        if (s > 0) {
          const float cs = (s == 3 ? 1.0f : 0.5f) * h;
          src = y + cs * (s == 1 ? k[0] : (s == 2 ? k[1] : k[2]));
        }
        any_w = ADJ;
      } else {
        const float h6 = h * (1.0f / 6.0f);
        yn = y + h6 * (k[0] + 2.0f * (k[1] + k[2]) + k[3]);
        src = yn;
      }
      if (ADJ && s == 0 && slot_base + NST > a.cap) overflow = true;
    }
    const float dst = eval(src, (any_w && !overflow) ? my_stage + (size_t)(slot_base + s) * blk_floats : nullptr);
    {
      const int ks = phase == PH_K0 ? 0 : (phase == PH_INIT1 ? 1 : s);
#pragma unroll
      for (int q = 0; q < 7; q++)
        if (q == ks) k[q] = dst;
    }
    if (status == 0) nfe++;

    if (phase == PH_K0 && !(ADJ || auto_dt)) {
      dt = o.adaptive ? fmin(o.dt_fixed, dtmax) : o.dt_fixed;
      phase = PH_STAGE;
      s = 1;
      running = begin_step();
    } else if (phase == PH_K0) {
      // Hairer–Nørsett–Wanner, part 1
      const float sk = fast_rcp(o.abstol + fabsf(y) * o.reltol);
      scr = sk;
      const float a0 = y * sk, a1v = k[0] * sk;
      float v0 = wave_sum64(counted ? a0 * a0 : 0.f), v1 = wave_sum64(counted ? a1v * a1v : 0.f);
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
      tmp = y + h * k[0];
      phase = PH_INIT1;
    } else if (phase == PH_INIT1) {
      const float dd = (k[1] - k[0]) * scr;
      float w0 = wave_sum64(counted ? dd * dd : 0.f), w1 = 0.f;
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
    } else if (s < LAST_STAGE) {
      s++;
    } else {
      if (ADJ && SOLVER == LDE_SOLVER_RK4) {
        const float h6 = h * (1.0f / 6.0f);
        yn = y + h6 * (k[0] + 2.0f * (k[1] + k[2]) + k[3]);
      }
      float r2 = 0.f;
      if (o.adaptive && counted) {
        float er = ts5::BT[0] * k[0];
#pragma unroll
        for (int jj = 1; jj < 7; jj++) er += ts5::BT[jj] * k[jj];
        er *= h;
        const float sk = o.abstol + fmaxf(fabsf(y), fabsf(yn)) * o.reltol;
        const float r = er * fast_rcp(sk);
        r2 = r * r;
      }
      if (!isfinite(yn)) r2 = __int_as_float(0x7fc00000);   // a non-finite state poisons the sum
      float s2 = wave_sum64(r2), s2b = 0.f;
      if (coupled) {
        if (status != 0) s2 = 0.f;
        w_grid_sum<false>(a.gs, gen, a.epoch, s2, s2b);
      }
      bool accepted = false;
      if (status == 0) {
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
        while (accepted && j < T && s_ts[j] <= tnew) {   // dense output at every save time inside the accepted step
          const double tj = s_ts[j];
          const float th = (tj >= tnew || (j == T - 1 && last)) ? 2.0f : (float)(tj - t) * fast_rcp(wq);
          float ov;
          if (th > 1.5f) ov = yn;
          else if (SOLVER == LDE_SOLVER_TSIT5) {
            float bw[7];
            tsit5_interp_weights(th, bw);
            float acc = bw[0] * k[0];
#pragma unroll
            for (int q = 1; q < 7; q++) acc += bw[q] * k[q];
            ov = y + wq * acc;
          } else {
            const float om = 1.0f - th;
            const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
            const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
            ov = h00 * y + (h10 * wq) * k[0] + h01 * yn + (h11 * wq) * k[4];
          }
          if (wv == 0 && lane < Dp) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * j) + lane] = ov;
          j++;
        }
        if (accepted) {
          y = yn;
          k[0] = k[LAST_STAGE];
          t = tnew;
          if (last) status = 1;
        }
        s = 1;
        running = begin_step();
      } else {
        if (accepted && !overflow) {
          if (tid < NST) {
            float bs;
            if (SOLVER == LDE_SOLVER_TSIT5) bs = ts5::A[6][tid];
            else bs = (tid == 0 || tid == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
            my_wts[(size_t)(slot_base + tid) * NB + ncol] = wq * bs;
          }
          slot_base += NST;
        }
        if (accepted) {
          y = yn;
          if (hit) {
            if (counted) {
              if (a.cot_lds) {
                if (is_l) y += s_cot[j * Dp + row];
                else if (o.checkpoint) y = s_cot[(T + j) * Dp + row];
              } else {
                const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * j) + row;
                if (is_l) y += a.dz_out[srcg];
                else if (o.checkpoint) y = a.z_out[srcg];
              }
            }
            t = s_ts[j];
            j--;
            if (j < 0) status = 1;
          } else
            t -= tnew;
        }
        s = 0;
        running = begin_step();
      }
    }
  }

  // ---- results
  const int st = status;
  if (!ADJ) {
    if (st > 1) {
      const float qn = __int_as_float(0x7fc00000);
      for (int e = tid; e < Dp * T; e += UT) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * (e / Dp)) + (e % Dp)] = qn;
    }
    if (tid == 0) {
      const int ret = st > 1 ? st - 1 : 0;
      if (a.retcode) a.retcode[b] = ret;
      a.st_ret[b] = ret;
    }
  } else {
    if (wv == 0 && lane >= DP && lane < DP + D) a.dz0[(size_t)b * D + (lane - DP)] = st > 1 ? 0.f : y;
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
}

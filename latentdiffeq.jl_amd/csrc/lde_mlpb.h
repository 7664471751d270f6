// lde_mlpb.h — three-layer networks wider than one wave at SMALL batches, second form (round 4): the hidden×hidden matrix as 2-D
// register BLOCKS, one copy serving both W₂·h and W₂ᵀ·δ, and the weight gradient kept on the CU (included by lde_mlp.hip after
// lde_mlpw.h).
//
// k_mlpw (lde_mlpw.h) gives every lane a ROW and a COLUMN of W₂: 2·H weight registers per lane (c2: 416, half of them AGPRs the
// VALU reaches only through v_accvgpr_read), every lane needs the WHOLE input vector of each product (H/4 broadcast ds_read_b128
// per lane and product: 800 LDS-array cycles per product and CU), and nothing is left for the weight gradient, whose (a_l, δ_l)
// panels it stages through HBM for k_mlp_dw (c2: 203 MB per launch against 0.84 MB of algorithmic traffic). Here:
//   * the 256 lanes of the workgroup (one trajectory, one wave per SIMD) form a 16 × 16 grid: lane (r, c) — r = tid >> 4 a DPP row,
//     c = tid & 15 the lane in it — owns the BLOCK W₂[13r … 13r+12][13c … 13c+12] as 13 rows × (6 register pairs + a single
//     register) (RB = CB = 13, CBP = 6: 169 VGPRs, all arch VGPRs). W₂·h₁: the lane needs only h₁[13c … 13c+12] (13 floats from
//     LDS), 78 v_pk_fma_f32 + 13 v_fma_f32, and the row sums meet inside the DPP row (quad_perm ×2, row_half_mirror, row_mirror:
//     4 steps per row, every lane of the row ends with all 13 sums, bitwise equal). W₂ᵀ·δ₂ runs on the SAME registers — 78
//     v_pk_fma_f32 + 13 v_fma_f32 with δ₂[i] on both halves — and its 13 partial sums per lane cross the 16 row groups through LDS
//     (then lane u adds its unit's 16 partials). The bias b₂ rides as column H₁ of the block against a constant 1 in h₁[H₁]
//     (16·13 = 208 block columns for H₁ ≤ 200 units: the padding), so it costs nothing — and the same 1 makes row H₁ of the
//     weight-gradient tiles below the gradient of b₂;
//   * the thin products stay as in k_mlpw: lane u owns unit u's row of W₁ and column of W₃ (a₁ and W₃ᵀλ: D′ FMAs each on the
//     broadcast state), and H → D′ (f = W₃h₂, vz = W₁ᵀδ₁) with lanes = (K-segment, output) on slices kept in LDS;
//   * THE WEIGHT GRADIENT NEVER LEAVES THE CU: the vectors an evaluation passes through LDS anyway (z|λ, h₁, δ₂, h₂, δ₁) are a RING
//     indexed by the stage; at accept time the step's 4 (RK4) or 6 (Tsit5) evaluations are folded by v_mfma_f32_16x16x4_f32 — K =
//     four evaluations per instruction, the δ-side operand scaled by its quadrature weight b_s·|h| — into accumulator-resident
//     16 × 16 tiles: 13 × 13 of gW₂ᵀ, 13 of gW₁, 13 of gW₃ᵀ, dealt round-robin to the four waves (50 tiles = 200 AGPRs per lane); a
//     rejected attempt folds nothing. Biases of the thin layers: one FMA per evaluation and lane at fold time. At the end every
//     workgroup writes ONE row of a [workgroups × nW] slab in flat destructure order and k_sum_rows adds the rows in index order
//     (bit-reproducible). No staging area, no k_mlp_dw, no k_reduce_tiles, no memsets: lde_adjoint is two launches.
// Same algorithm and control arithmetic as k_mlpw (state in lanes 0 … 2·DP−1 of every wave, HNW initial step, PI controller
// carried across the save times, one-round-trip grid sum for coupled control). Limits: exactly three Dense layers D′ → H₁ → H₂ →
// D′ with D′ ≤ 16, 128 < max(H₁, H₂) ≤ 200 (narrower networks: k_mlpw's two-wave form), P = 0, no analytic part.

struct BDims {
  int DP;                 // lanes per state half: 8 or 16
  int SEG, GS;            // narrow products: K-segments (64/DP), float4 groups per lane
  // float offsets in the packed array
  int o_wb, o_w13, o_b1, o_b3, o_n3, o_n1, total;
};

namespace mlpb {
constexpr int W = 4, UT = 256;        // waves, lanes of a workgroup
constexpr int RB = 13, CB = 13, CBP = 6; // block: 13 rows × 13 columns = 6 column pairs + a single column
constexpr int HV = 224;               // floats of a hidden vector in LDS (16·13 = 208 block columns / rows, padded for the narrow products' float4 reads)
constexpr int NT = 13;                // 16-wide tiles along a hidden index (unit H₁ = 200, the constant 1, included)
constexpr int XS = 32;                // floats of the [z | λ] part of a ring slot
constexpr int SLOT = XS + 4 * HV;     // ring slot: xs | h₁ | g₂ → δ₂ | h₂ | δ₁
constexpr int NTL = 51;               // weight-gradient tile slots of a wave: 39 + 4 of gW₂ᵀ, 4 of gW₁, 4 of gW₃ᵀ
// The weight-gradient tiles live in AGPRs the COMPILER DOES NOT KNOW ABOUT (round 4): an empty asm clobbering a255 makes the kernel
// allocate all 512 registers of a lane, and tile n of a wave is a[A0 + 4n : A0 + 4n + 3], touched only by inline asm with literal register
// numbers (v_accvgpr_write at the start, the fold's v_mfma_f32_16x16x4_f32 with the tile as C and D, v_accvgpr_read at the end). As
// ordinary variables the loop-carried tiles were kept in TWO places by the register allocator (8 AGPRs per tile, `v_accvgpr_mov` blocks
// at the loop edges), so that only 28–40 of the 51 fit, the rest lived in LDS — and 11 of those were spilled to scratch around the
// evaluation loop. NOTHING IN LLVM RESERVES THESE REGISTERS (`amdgpu_num_vgpr` is a budget that is split evenly between VGPRs and AGPRs
// once a kernel uses AGPRs, not a cap on one half: a variant of this kernel with more live values was seen to use a0–a91): what the
// compiler needs for itself — copies of VGPR values, ≤ 40 AGPRs in every instantiation — happens to sit below A0 because it allocates
// from a0 upwards. EVERY BUILD IS THEREFORE CHECKED (check_agprs.py, called by build.py and by tests/test_abi.py): no instruction of
// these kernels may write an AGPR ≥ A0 except the asm's own three forms; a violation fails the build.
constexpr int NTH = 51;               // tiles per wave in hidden AGPRs (all of them)
constexpr int A0 = 256 - 4 * NTH;     // the compiler's AGPRs: a[0 : A0)
static_assert(NTH == NTL, "every tile slot of a wave is a hidden register tile (there is no LDS tile path any more)");
// tile at a[R : R + 3] += av ⊗ bv (K = the four lane groups). The two wait states in front of the MFMA are the VALU → MFMA-operand distance
// (the compiler pads nothing inside an asm string); an MFMA that takes the previous one's D whole as its C needs none.
template <int R>
__device__ __forceinline__ void areg_mfma(float av, float bv) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(av), "v"(bv), "n"(R), "n"(R + 3));
}
template <int R>
__device__ __forceinline__ void areg_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\tv_accvgpr_write_b32 a[%3], 0"
               :: "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
}
template <int R>
__device__ __forceinline__ f32x4 areg_read() {   // (the caller has put the MFMA → reader distance in front of the first read)
  f32x4 r;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
  return r;
}
template <int N> __device__ __forceinline__ void tile_mfma(float av, float bv) { areg_mfma<A0 + 4 * N>(av, bv); }
template <int N> __device__ __forceinline__ void tile_zero() { areg_zero<A0 + 4 * N>(); }
template <int N> __device__ __forceinline__ f32x4 tile_read() { return areg_read<A0 + 4 * N>(); }
// f(integral_constant<int, I>) for I = B … E − 1: the tile number must be a constant EXPRESSION where the asm is written (an unrolled loop
// variable is not one: the switch over 51 cases stayed a run-time branch tree)
template <int B_, int E_, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B_ < E_) {
    f(std::integral_constant<int, B_>{});
    static_for<B_ + 1, E_>(f);
  }
}
}  // namespace mlpb

// one-time packing (set_weights): everything in the order the kernel's lanes read it
static __global__ void k_build_bpack(const float* __restrict__ Wflat, MlpDims dm, BDims bd, float* __restrict__ wp) {
  using namespace mlpb;
  const int H1 = dm.sizes[1], H2 = dm.sizes[2], Dp = dm.Dp;
  const float *W1 = Wflat + dm.w_off[0], *W2 = Wflat + dm.w_off[1], *W3 = Wflat + dm.w_off[2];   // column-major [out×in]
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < bd.total; e += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (e < bd.o_w13) {            // wb[i·CB + j][tid] = W₂(13r + i, 13c + j); column H₁ carries b₂
      const int q = e / UT, tid = e % UT, j = q % CB, i = q / CB;
      const int row = RB * (tid >> 4) + i, col = CB * (tid & 15) + j;
      if (row < H2) {
        if (col < H1) v = W2[row + H2 * col];
        else if (col == H1) v = Wflat[dm.b_off[1] + row];
      }
    } else if (e < bd.o_b1) {      // w13[u][0 … DP) = W₁(u, ·), w13[u][DP … 2·DP) = W₃(·, u); rows of 2·DP + 4 floats (conflict-free ds_read_b128)
      const int r = e - bd.o_w13, u = r / (2 * bd.DP + 4), k = r % (2 * bd.DP + 4);
      if (k < bd.DP) { if (u < H1 && k < Dp) v = W1[u + H1 * k]; }
      else if (k < 2 * bd.DP) { const int d = k - bd.DP; if (u < H2 && d < Dp) v = W3[d + Dp * u]; }
    } else if (e < bd.o_b3) {
      const int u = e - bd.o_b1;
      if (u < H1) v = Wflat[dm.b_off[0] + u];
    } else if (e < bd.o_n3) {
      const int d = (e - bd.o_b3) % bd.DP;
      if (d < Dp) v = Wflat[dm.b_off[2] + d];
    } else {                       // narrow slices: [g][lane][4]; lane = seg·DP + d, k = (seg·GS + g)·4 + c
      const bool n1 = e >= bd.o_n1;
      const int r = e - (n1 ? bd.o_n1 : bd.o_n3);
      const int c = r & 3, ln = (r >> 2) & 63, g = r >> 8;
      const int seg = ln / bd.DP, d = ln % bd.DP, k = (seg * bd.GS + g) * 4 + c;
      if (!n1) { if (d < Dp && k < H2) v = W3[d + Dp * k]; }      // f_d  = Σ_k W₃(d, k) h₂_k
      else     { if (d < Dp && k < H1) v = W1[k + H1 * d]; }      // vz_d = Σ_k W₁(k, d) δ₁_k
    }
    wp[e] = v;
  }
}

// v + (v of the lane 16 / 32 further, modulo 32 / 64): v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows of one
// register with the even rows of the other; with both registers holding v, one ends up with the even rows repeated, the other with
// the odd rows. (Inline asm: ROCm 7.2's __builtin_amdgcn_permlane*_swap returns its FIRST result in both elements — probed.)
template <int W_>
__device__ __forceinline__ float swap_sum(float v) {
  float a = v, b = v;
  if (W_ == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {   // v + (v of the lane CTRL maps this lane to): the butterfly steps inside a row of 16
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (lane i ↔ 7 − i), row_mirror (lane i ↔ 15 − i) = the sum over the 16
//  lanes of a DPP row in every lane of the row; the two operands of every add are exchanged between the partner lanes and f32
//  addition commutes, so all 16 results are bitwise equal)

__device__ __forceinline__ float sgpr_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ long long sgpr_ll(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double sgpr_d(double v) { return __builtin_bit_cast(double, sgpr_ll(__builtin_bit_cast(long long, v))); }

// DISC (with ADJ): LDE_SENSE_DISCRETE — the reverse sweep over the forward solve's step record instead of a reverse-time solve (the block
// behind the evaluation lambdas; lde_mlpd.h has the algorithm). The template's LAST bool stays ADJ (check_agprs.py reads it).
template <int SOLVER, int DP, int ACT, bool DISC, bool ADJ>
__global__ void __launch_bounds__(256, 1) k_mlpb(MlpDims dm, BDims bd, KOpts o, VArgs a) {
  using namespace mlpb;
  static_assert(ADJ || !DISC, "the discrete sweep is an adjoint");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SEG = 64 / DP, G1 = DP / 4, GS = (200 / SEG + 3) / 4;   // GS: host = bd.GS
  constexpr int NST = SOLVER == LDE_SOLVER_TSIT5 ? 6 : 4;                // weighted stages of a step = ring slots the fold reads
  constexpr int NSL = ADJ ? NST + 1 : 1;                                  // + one scratch slot (initial-step probes, the FSAL stage)
  // SPEC (adjoint, Tsit5): an attempt never evaluates its first stage. f(yₙ₊₁) — the step's seventh evaluation — IS the next attempt's k₁
  // when the step is accepted inside a save interval, k₁ of a rejected attempt is still valid, and behind a save time (where λ jumps) the
  // first evaluation at the jumped state runs BEFORE the decision is known, while the grid-wide error sum is under way (coupled control):
  // its vectors sit in the spare ring slot and move to slot 0 at accept. Same evaluations on the same inputs — the same bits, fewer of them.
  constexpr bool SPEC = ADJ && SOLVER == LDE_SOLVER_TSIT5;
  constexpr int W13S = 2 * DP + 4;
  static_assert(DP == 8 || DP == 16, "k_mlpb geometry");
  const int T = o.T, B = o.B, D = dm.D, Dp = dm.Dp, tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = tid;                      // the hidden unit this lane owns in the thin products
  const int br = tid >> 4, bc = tid & 15; // block coordinates
  const int H1 = dm.sizes[1], H2 = dm.sizes[2];
  // ---- LDS: save times | ring | partial sums of W₂ᵀδ₂ | narrow slices | thin-layer weights | K-split partials | cotangents
  double* s_ts = reinterpret_cast<double*>(smem);
  float* ring = reinterpret_cast<float*>(smem + (((size_t)T * 8 + 15) & ~size_t(15)));
  float* part = ring + NSL * SLOT;
  f32x4* n3 = reinterpret_cast<f32x4*>(part + (ADJ ? 16 * HV : 0));
  f32x4* n1 = n3 + GS * 64;
  float* w13 = reinterpret_cast<float*>(n1 + (ADJ ? GS * 64 : 0));     // rows of W₁ | columns of W₃ by unit: [HV][2·DP + 4]
  // KSPLIT (adjoint, D′ = 16): the two H → D′ products are split over the four waves along K — a quarter of the 14 read groups each,
  // which this kernel's register pressure serialises into dependent LDS round trips and which cannot be batched at this width (112
  // registers) — and the partial results meet in s_np behind a barrier (the first product's behind the one that is there anyway)
  constexpr bool KSPLIT = ADJ && DP == 16;
  float* s_np = w13 + HV * W13S;                                       // [2 products][4 waves][D′] partial outputs (KSPLIT)
  float* s_cot = s_np + (KSPLIT ? 2 * W * DP : 0);                     // adjoint: the trajectory's cotangents (and saved states) by save time
  for (int i = tid; i < T; i += UT) s_ts[i] = a.ts[i];
  for (int i = tid; i < NSL * SLOT + (ADJ ? 16 * HV : 0); i += UT) ring[i] = 0.f;
  {
    const f32x4* g3 = reinterpret_cast<const f32x4*>(a.wpack + bd.o_n3);
    for (int i = tid; i < GS * 64; i += UT) n3[i] = g3[i];
    if (ADJ) {
      const f32x4* g1 = reinterpret_cast<const f32x4*>(a.wpack + bd.o_n1);
      for (int i = tid; i < GS * 64; i += UT) n1[i] = g1[i];
    }
    const f32x4* g13 = reinterpret_cast<const f32x4*>(a.wpack + bd.o_w13);
    for (int i = tid; i < HV * W13S / 4; i += UT) reinterpret_cast<f32x4*>(w13)[i] = g13[i];
    if (ADJ && a.cot_lds) {   // [T][Dp] dẑ (+ [T][Dp] ẑ when the adjoint restarts from the saved states): no global load inside the solve
      for (int i = tid; i < T * Dp; i += UT) {
        const size_t g = (size_t)Dp * ((size_t)b + (size_t)B * (i / Dp)) + (i % Dp);
        s_cot[i] = a.dz_out[g];
        if (o.checkpoint) s_cot[T * Dp + i] = a.z_out[g];
      }
    }
  }
  // ---- registers: the lane's block of W₂ (pairs along the columns), its row of W₁ and column of W₃
  f32x2 wb[RB][CBP];
  float ws[RB];
  {
    const float* wp = a.wpack + bd.o_wb + tid;
#pragma unroll
    for (int i = 0; i < RB; i++) {
#pragma unroll
      for (int jp = 0; jp < CBP; jp++) wb[i][jp] = f32x2{wp[(i * CB + 2 * jp) * UT], wp[(i * CB + 2 * jp + 1) * UT]};
      ws[i] = wp[(i * CB + CB - 1) * UT];
    }
  }
  const f32x4* const my13 = reinterpret_cast<const f32x4*>(w13 + (u < HV ? u : 0) * W13S);   // this lane's row of W₁ | column of W₃ (LDS)
  const float b1 = a.wpack[bd.o_b1 + u], b3 = a.wpack[bd.o_b3 + (lane % DP)];
  constexpr int act = ACT;   // (compile-time: a run-time activation is a branch per call, and the block product applies it 13 times per lane)
  // the weight gradient: accumulator-resident 16×16 tiles (C/D layout of v_mfma_f32_16x16x4_f32: column = lane & 15, row = 4·(lane >> 4) + reg)
  float gb1 = 0.f, gb3 = 0.f;
  if (ADJ) {
    asm volatile("" ::: "a255");   // the kernel owns all 512 registers of a lane
    static_for<0, NTL>([&](auto nc) { tile_zero<decltype(nc)::value>(); });
  }
  __syncthreads();

  const bool coupled = dm.coupled != 0;
  const double t0 = s_ts[0], tend = s_ts[T - 1], dtmax = fabs(tend - t0);
  unsigned gen = 0;

  // ---- state: lane i < DP holds z_i, lane DP + i holds λ_i (adjoint); the other lanes stay 0
  const bool is_z = lane < Dp, is_l = ADJ && lane >= DP && lane < DP + Dp;
  const bool counted = is_z || is_l;
  const int row = is_l ? lane - DP : lane;
  float y = 0.f, yn = 0.f, tmp = 0.f, scr = 0.f, k[7];
#pragma unroll
  for (int s = 0; s < 7; s++) k[s] = 0.f;
  if (!ADJ) {
    if (lane < D) y = a.z0[(size_t)b * D + lane];
  } else if (DISC) {
    if (is_z) y = a.z_out[(size_t)b * Dp + row];   // ẑ₀ (save time 0): only the failure check reads it here
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
  // Σ over the lanes that differ in the K-segment bits (lane bits ≥ log₂ DP), in every lane — on the VALU (a DPP rotate inside the
  // row of 16, v_permlane16_swap / v_permlane32_swap across rows), not three dependent ds_bpermute round trips
  auto xor_segs = [&](float p) -> float {
    if (DP <= 8) p = dpp_xadd<0x128>(p);   // row_ror:8
    return swap_sum<32>(swap_sum<16>(p));
  };
  // H → D′ with lanes = (K-segment, output): the slice `ns` against the LDS vector `vec`; every lane with lane % DP == d gets output d
  auto narrow = [&](const f32x4* ns, const float* vec) -> float {
    const f32x4* hv = reinterpret_cast<const f32x4*>(vec) + (lane / DP) * GS;
    f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
    if (DP <= 8 && !DISC) {   // (not in the discrete sweep: its three inlined evaluations leave the allocator no room for 56 registers of reads —
                              //  the compiler's copies then climb into the hidden tiles and check_agprs.py fails the build)
      // every read issued before the first product waits for one: left to itself under this kernel's register pressure the compiler reuses
      // ONE pair of buffers — read, wait, multiply: GS dependent LDS round trips (c2: 0.918 -> 0.901 ms). Not at D′ = 16: twice the
      // reads, and the 112 registers they need at once are spilled (latentode_ref: 3.43 -> 3.51 ms with it).
      f32x4 wq4[GS], xv[GS];
#pragma unroll
      for (int g = 0; g < GS; g++) {
        wq4[g] = ns[g * 64 + lane];
        xv[g] = hv[g];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GS; g++) {
        p01 += wq4[g].lo * xv[g].lo;
        p23 += wq4[g].hi * xv[g].hi;
      }
    } else {
#pragma unroll
      for (int g = 0; g < GS; g++) {
        const f32x4 wq4 = ns[g * 64 + lane], xv = hv[g];
        p01 += wq4.lo * xv.lo;
        p23 += wq4.hi * xv.hi;
      }
    }
    return xor_segs((p01.x + p01.y) + (p23.x + p23.y));
  };
  // this wave's quarter of the K range of the same product (KSPLIT): groups wv·GP … wv·GP + GP − 1, all reads in flight together
  auto narrow_part = [&](const f32x4* ns, const float* vec) -> float {
    constexpr int GP = (GS + W - 1) / W;
    const f32x4* hv = reinterpret_cast<const f32x4*>(vec) + (lane / DP) * GS;
    f32x4 wq4[GP], xv[GP];
#pragma unroll
    for (int q = 0; q < GP; q++) {
      const int g = wv * GP + q, gc = g < GS ? g : GS - 1;
      wq4[q] = ns[gc * 64 + lane];
      xv[q] = hv[gc];
      if (g >= GS) xv[q] = f32x4{0.f, 0.f, 0.f, 0.f};   // (wave-uniform: a group beyond the range contributes zeros)
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < GP; q++) {
      p01 += wq4[q].lo * xv[q].lo;
      p23 += wq4[q].hi * xv[q].hi;
    }
    return xor_segs((p01.x + p01.y) + (p23.x + p23.y));
  };

  // one evaluation of the (augmented) right-hand side: src → dst; its vectors stay in ring slot `slot`
  // vj (wave-uniform; only the discrete sweep passes false): false = the forward half alone — f, h₁, h₂ — for the evaluations that rebuild slopes
  // fwd (wave-uniform; only the discrete sweep passes false): false = the vector-Jacobian half alone, at the point whose forward half left
  // z, h₁, h₂ in ring slot `slot` earlier (the sweep's pass 1): no W₁z, no W₂h₁, no f
  auto eval = [&](float src, int slot, bool vj = true, bool fwd = true) -> float {
    PROF_T2(e0);
    float* xs = ring + slot * SLOT;
    float *h1v = xs + XS, *d2v = h1v + HV, *h2v = d2v + HV, *d1v = h2v + HV;
    if (lane < XS) xs[lane] = src;   // (every wave stores the same values)
    asm volatile("" ::: "memory");   // same wave, in-order LDS: the broadcast reads below see the write (no barrier needed)
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
    float h1;
    if (!DISC || fwd) {
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G1; g++) {
        const f32x4 xv = x4[g], wv4 = my13[g];
        c01 += wv4.lo * xv.lo;
        c23 += wv4.hi * xv.hi;
      }
      const float a1 = b1 + ((c01.x + c01.y) + (c23.x + c23.y));
      h1 = u == H1 ? 1.f : act_fn(act, a1);   // unit H₁: the constant that carries b₂ (rows beyond: zero weights and bias ⇒ act(0) = 0)
      if (u < HV) h1v[u] = h1;
    } else
      h1 = u < HV ? h1v[u] : 0.f;
    if (ADJ && vj) {
      f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G1; g++) {
        const f32x4 xv = x4[DP / 4 + g], wv4 = my13[DP / 4 + g];   // λ
        c01 += wv4.lo * xv.lo;
        c23 += wv4.hi * xv.hi;
      }
      if (u < HV) d2v[u] = (c01.x + c01.y) + (c23.x + c23.y);   // (W₃ᵀλ)_u; becomes δ₂ below
    }
    __syncthreads();
    PROF_T2(e1);
    // ---- the block products (independent chains side by side: two groups of rows, each as 7 / 6 accumulator pairs + singles)
    float h2[RB];
    if (DISC && !fwd) {   // h₂ as pass 1 left it (a row's 16 lanes read the same 13 words)
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = h2v[RB * br + i];
    } else {
      const float* hc = h1v + CB * bc;
      f32x2 hp[CBP];
#pragma unroll
      for (int jp = 0; jp < CBP; jp++) hp[jp] = f32x2{hc[2 * jp], hc[2 * jp + 1]};
      const float hs = hc[CB - 1];
#pragma unroll
      for (int i0 = 0; i0 < RB; i0 += 7) {
        f32x2 acc[7];
        float as[7];
#pragma unroll
        for (int ii = 0; ii < 7; ii++)
          if (i0 + ii < RB) { acc[ii] = wb[i0 + ii][0] * hp[0]; as[ii] = ws[i0 + ii] * hs; }
#pragma unroll
        for (int jp = 1; jp < CBP; jp++)
#pragma unroll
          for (int ii = 0; ii < 7; ii++)
            if (i0 + ii < RB) acc[ii] += wb[i0 + ii][jp] * hp[jp];
#pragma unroll
        for (int ii = 0; ii < 7; ii++)
          if (i0 + ii < RB) h2[i0 + ii] = (acc[ii].x + acc[ii].y) + as[ii];
      }
      // the 16 lanes of a row add their 13 partial sums: four butterfly steps, all 13 values per step
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = dpp_xadd<0xB1>(h2[i]);    // quad_perm [1,0,3,2]
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = dpp_xadd<0x4E>(h2[i]);    // quad_perm [2,3,0,1]
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = dpp_xadd<0x141>(h2[i]);   // row_half_mirror
#pragma unroll
      for (int i = 0; i < RB; i++) h2[i] = act_fn(act, dpp_xadd<0x140>(h2[i]));   // row_mirror; every lane of the row holds the 13 sums, bitwise equal
    }
    if (ADJ && vj) {
      float d2[RB];
#pragma unroll
      for (int i = 0; i < RB; i++) d2[i] = d2v[RB * br + i] * act_grad(act, h2[i]);
      if (bc == 0) {   // one lane of the row leaves h₂ and δ₂ for the thin products and the fold
#pragma unroll
        for (int i = 0; i < RB; i++) {
          if (!DISC || fwd) h2v[RB * br + i] = h2[i];
          d2v[RB * br + i] = d2[i];
        }
      }
      f32x2 gp[CBP];
      float gs = 0.f;
#pragma unroll
      for (int jp = 0; jp < CBP; jp++) gp[jp] = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const f32x2 dd = {d2[i], d2[i]};
#pragma unroll
        for (int jp = 0; jp < CBP; jp++) gp[jp] += wb[i][jp] * dd;
        gs += ws[i] * d2[i];
      }
      float* pp = part + br * HV + CB * bc;
#pragma unroll
      for (int jp = 0; jp < CBP; jp++) { pp[2 * jp] = gp[jp].x; pp[2 * jp + 1] = gp[jp].y; }
      pp[CB - 1] = gs;
    } else if (bc == 0) {
#pragma unroll
      for (int i = 0; i < RB; i++) h2v[RB * br + i] = h2[i];
    }
    __syncthreads();
    PROF_T2(e2);
    float f = 0.f;
    if (!DISC || fwd) {
      if (KSPLIT) {
        const float pf = narrow_part(n3, h2v);
        if (lane < DP) s_np[wv * DP + lane] = pf;
      } else
        f = narrow(n3, h2v) + b3;
    }
    float dst = is_z ? f : 0.f;
    PROF_ADD2(3, e0, e1);
    PROF_ADD2(4, e1, e2);
    if (ADJ && !vj && KSPLIT) {   // the forward half alone, K split over the waves: the partial outputs meet behind a barrier of their own
      __syncthreads();
      const int dd = lane % DP;
      f = b3 + ((s_np[dd] + s_np[DP + dd]) + (s_np[2 * DP + dd] + s_np[3 * DP + dd]));
      dst = is_z ? f : 0.f;
    }
    if (ADJ && vj) {
      float g1 = 0.f;
      if (u < HV) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; r++) p[r] = part[r * HV + u];
#pragma unroll
        for (int r = 0; r < 16; r++) g1 += p[r];
      }
      const float d1 = u < H1 ? g1 * act_grad(act, h1) : 0.f;
      if (u < HV) d1v[u] = d1;
      __syncthreads();
      float vz;
      if (KSPLIT) {
        const int dd = lane % DP;
        f = b3 + ((s_np[dd] + s_np[DP + dd]) + (s_np[2 * DP + dd] + s_np[3 * DP + dd]));   // the same order in every wave: the same bits
        dst = is_z && (!DISC || fwd) ? f : 0.f;   // (the vector-Jacobian half alone: no f)
        const float pv = narrow_part(n1, d1v);
        if (lane < DP) s_np[(W + wv) * DP + lane] = pv;
        __syncthreads();
        vz = (s_np[W * DP + dd] + s_np[(W + 1) * DP + dd]) + (s_np[(W + 2) * DP + dd] + s_np[(W + 3) * DP + dd]);
      } else
        vz = narrow(n1, d1v);     // every lane with lane % DP == d holds vz_d
      if (is_l) dst = -vz;
      PROF_T2(e3);
      PROF_ADD2(5, e2, e3);
    }
    PROF_T2(e5);
    PROF_ADD2(1, e0, e5);
    PROF_ADD2(20, e5 - 1, e5);
    return dst;
  };

  // the accepted step's share of the quadrature gW = Σ_s |h| b_s (∂f/∂W)ᵀλ, from the vectors its evaluations left in the ring.
  // Tile slots of wave w (every operand address = a per-lane base + a compile-time offset):
  //   n = 3·ti + m, ti < 13, m < 3 : gW₂ᵀ tile (ti, tj = 4m + w)          n = 39 + q : gW₂ᵀ tile (ti = 4q + w, tj = 12)
  //   n = 43 + q : gW₁ tile 4q + w (A = δ₁, B = z)                        n = 47 + q : gW₃ᵀ tile 4q + w (A = h₂, B = λ)
  // (q = 3 is a real tile for wave 0 only; the other waves' slot accumulates finite junk that is never written out)
  auto fold = [&](int nvalid) {   // nvalid (wave-uniform): ring slots [0, nvalid) count
    PROF_T2(f0);
    const int l15 = lane & 15, e4 = lane >> 4;
#pragma unroll
    for (int g = 0; g < (NST + 3) / 4; g++) {
      const int e = 4 * g + e4;
      float bs;
      if (DISC) bs = 1.f;   // (the scale h·b_i is inside k̄)
      else if (SOLVER == LDE_SOLVER_TSIT5) bs = e == 0 ? ts5::A[6][0] : e == 1 ? ts5::A[6][1] : e == 2 ? ts5::A[6][2] : e == 3 ? ts5::A[6][3] : e == 4 ? ts5::A[6][4] : e == 5 ? ts5::A[6][5] : 0.f;
      else bs = (e == 0 || e == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
      const bool ev = e < nvalid;              // (Tsit5's second group has two evaluations: the other two K slots are zeros)
      const float wsc = ev ? (DISC ? 1.f : wq * bs) : 0.f, am = ev ? 1.f : 0.f;
      const float* sl = ring + (ev ? e : 0) * SLOT + l15;
      const float* pa = sl + XS + 16 * wv;     // A operands of the slots whose tile index is 4q + w
      float bm[3];
#pragma unroll
      for (int m = 0; m < 3; m++) bm[m] = sl[XS + HV + 64 * m + 16 * wv] * wsc;
      // every operand of the group is read before the first MFMA (one LDS round trip, not one per tile row: the asm statements keep their
      // order, and a read placed between them would be waited for in front of the next one)
      float av[NT], a39[4], a43[4], a47[4];
#pragma unroll
      for (int ti = 0; ti < NT; ti++) av[ti] = sl[XS + 16 * ti] * am;
      const float b12 = sl[XS + HV + 16 * 12] * wsc;
      const float z0v = sl[0], l0v = sl[DP];   // (unconditional reads: a read under a lane condition is waited for where it stands)
      const float bz = l15 < Dp ? z0v * wsc : 0.f, bl = l15 < Dp ? l0v * wsc : 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        a39[q] = pa[64 * q] * am;
        a43[q] = pa[3 * HV + 64 * q] * am;
        a47[q] = pa[2 * HV + 64 * q] * am;
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NT>([&](auto tic) {        // gW₂ᵀ[i][o] += Σ_e h₁_e[i] · (w_e δ₂_e)[o]
        constexpr int ti = decltype(tic)::value;
        tile_mfma<3 * ti + 0>(av[ti], bm[0]);
        tile_mfma<3 * ti + 1>(av[ti], bm[1]);
        tile_mfma<3 * ti + 2>(av[ti], bm[2]);
      });
      static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; tile_mfma<39 + q>(a39[q], b12); });
      // gW₁[u][k] += Σ_e (w_e δ₁_e)[u] · z_e[k];  gW₃ᵀ[u][d] += Σ_e h₂_e[u] · (w_e λ_e)[d]
      static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; tile_mfma<43 + q>(a43[q], bz); });
      static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; tile_mfma<47 + q>(a47[q], bl); });
    }
    // thin biases: gb₁[u] += Σ_e w_e δ₁_e[u] (lane u), gb₃[d] += Σ_e w_e λ_e[d] (the λ lanes); gb₂ is row H₁ of the gW₂ᵀ tiles
#pragma unroll
    for (int e = 0; e < NST; e++) {
      const float bs = SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[6][e] : ((e == 0 || e == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f));
      const float wb_ = DISC ? (e < nvalid ? 1.f : 0.f) : wq * bs;
      const float* sl = ring + e * SLOT;
      if (u < HV) gb1 += wb_ * sl[XS + 3 * HV + u];
      if (lane < XS) gb3 += wb_ * sl[lane];
    }
    __syncthreads();   // the next attempt overwrites the ring: every wave has read it
    PROF_T2(f1);
    PROF_ADD2(6, f0, f1);
  };

  // this wave's sum of the attempt's scaled squared errors (every wave holds the whole state: the same number in all of them)
  auto err_sum = [&]() -> float {
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
    return wave_sum64(r2);
  };
  // the adjoint's state behind save time j: λ += Δ_j, z reset to the saved ẑ(t_j) (checkpointing)
  auto jumped = [&](float v) -> float {
    if (counted) {
      if (a.cot_lds) {
        if (is_l) v += s_cot[j * Dp + row];
        else if (o.checkpoint) v = s_cot[(T + j) * Dp + row];
      } else {
        const size_t srcg = (size_t)Dp * ((size_t)b + (size_t)B * j) + row;
        if (is_l) v += a.dz_out[srcg];
        else if (o.checkpoint) v = a.z_out[srcg];
      }
    }
    return v;
  };
  if constexpr (DISC) {
    // ---- LDE_SENSE_DISCRETE: the recorded steps (t_n, dt_n, y_n), last to first. z lanes carry the stage points and slopes (y = y_n,
    //      yn = y_{n+1}, k[i] = k_{i+1}); λ lanes carry the cotangents (y = ȳ_{n+1}, yn = ȳ_n, k[i] = k̄_{i+1}, scr = the k̄₁ that travels on)
    constexpr int S = NST;
    constexpr float RK[5][4] = {{0.f, 0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f, 0.f}, {0.f, 0.5f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f},
                                {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f}};
    auto A = [&](int i, int q) -> float { return SOLVER == LDE_SOLVER_TSIT5 ? ts5::A[i][q] : RK[i][q]; };
    const StepRec R = o.rec;
    const int seq = coupled ? 0 : b;
    const int ns = R.n[seq];
    if (status == 0 && (ns < 1 || ns > R.cap)) status = 1 + LDE_RET_MAXITERS;   // no usable record: NaN gradient, never a truncated sweep
    y = 0.f;                            // (ẑ₀ has served the failure check; the last evaluation reloads it)
    // stage point i of the step from y_n and the slopes (z lanes; i == S: y_{n+1})
    auto point = [&](int i, float hh) -> float {
      float zv = y;
      if (SOLVER == LDE_SOLVER_RK4 && i == S) zv = y + (hh * (1.0f / 6.0f)) * (k[0] + 2.0f * (k[1] + k[2]) + k[3]);
      else {
#define DSTAGE(S_)                                                               \
  case S_: {                                                                     \
    float accv = A(S_, 0) * k[0];                                                \
    _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += A(S_, jj) * k[jj]; \
    zv = y + hh * accv;                                                          \
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
      j = T - 1;
      double tnext = tend;
      // The next step's record is requested while this step's evaluations run (a load at the head of a step is covered by nothing): Tsit5
      // −1.6 % on the reference NODE's sweep; the RK4 instantiation measured +1 % with it (abl/ab_mlp.sh) and loads at the head.
      constexpr bool PFR = SOLVER == LDE_SOLVER_TSIT5;
      double t_pf = 0.0, dt_pf = 0.0;
      float y_pf = 0.f;
      if (PFR) {
        t_pf = R.t[(size_t)(ns - 1) * R.nseq + seq];
        dt_pf = R.dt[(size_t)(ns - 1) * R.nseq + seq];
        if (is_z) y_pf = R.y[((size_t)(ns - 1) * B + b) * Dp + row];
      }
#pragma unroll 1
      for (int sidx = ns - 1; sidx >= 0; sidx--) {
        double ts_n, dts;
        float y_rec = 0.f;
        if (PFR) {
          ts_n = t_pf;
          dts = dt_pf;
          y_rec = y_pf;
          const size_t sp = sidx > 0 ? sidx - 1 : 0;
          t_pf = R.t[sp * R.nseq + seq];
          dt_pf = R.dt[sp * R.nseq + seq];
          if (is_z) y_pf = R.y[(sp * B + b) * Dp + row];
        } else {
          ts_n = R.t[(size_t)sidx * R.nseq + seq];
          dts = R.dt[(size_t)sidx * R.nseq + seq];
          if (is_z) y_rec = R.y[((size_t)sidx * B + b) * Dp + row];
        }
        const float hh = (float)dts;
        const bool lastst = sidx == ns - 1;
        const double tnw = tnext;
        tnext = ts_n;
        if (is_z) y = y_rec;
        else {
          yn = 0.f;
#pragma unroll
          for (int q = 0; q < S; q++) k[q] = 0.f;
          k[S] = is_l ? scr : 0.f;
        }
        // pass 1: the slopes k_1 … k_S — forward halves; stage i ≥ 1 leaves z, h₁, h₂ in ITS ring slot (S − i) for pass 2's
        // vector-Jacobian half, the first stage point goes through the scratch slot (its pullback is the next step's FSAL point)
#pragma unroll 1
        for (int i = 0; i < S; i++) {
          const float dstv = eval(is_z ? point(i, hh) : 0.f, i == 0 ? NST : S - i, false);
#pragma unroll
          for (int q = 0; q < S; q++)
            if (q == i && is_z) k[q] = dstv;
        }
        if (is_z) yn = point(S, hh);
        // the save times inside the step (t_n, t_{n+1}]
        while (j >= 1 && sgpr_d(s_ts[j]) > ts_n) {   // (wave-uniform: kept scalar)
          const double tj = sgpr_d(s_ts[j]);
          const bool at_end = tj >= tnw || (j == T - 1 && lastst);
          const float th = at_end ? 2.0f : (float)(tj - ts_n) * fast_rcp(hh);
          if (is_l) {
            const float dj = a.cot_lds ? s_cot[j * Dp + row] : a.dz_out[(size_t)Dp * ((size_t)b + (size_t)B * j) + row];
            if (at_end) y += dj;
            else if (SOLVER == LDE_SOLVER_TSIT5) {
              float bw[7];
              tsit5_interp_weights(th, bw);
              yn += dj;
#pragma unroll
              for (int q = 0; q < 7; q++) k[q] += (hh * bw[q]) * dj;
            } else {
              const float om = 1.0f - th;
              const float h00 = (1.0f + 2.0f * th) * om * om, h10 = th * om * om;
              const float h01 = th * th * (3.0f - 2.0f * th), h11 = th * th * (th - 1.0f);
              yn += h00 * dj;
              k[0] += (h10 * hh) * dj;
              y += h01 * dj;
              k[S] += (h11 * hh) * dj;
            }
          }
          j--;
        }
        // pass 2: Jᵀk̄ at y_{n+1} (a fused evaluation: nothing has been evaluated there), then at g_S … g_2 (the vector-Jacobian half
        // alone, on the activations pass 1 left); their vectors stay in ring slots 0 … S − 1 for the fold
#pragma unroll 1
        for (int i = S; i >= 1; i--) {
          float kb = 0.f;
#pragma unroll
          for (int q = 0; q <= S; q++) kb = q == i ? k[q] : kb;
          const float dstv = eval(is_z ? (i == S ? yn : point(i, hh)) : kb, S - i, true, i == S);
          if (is_l) {
            const float v = -dstv;
            if (i == S) {
              y += v;
#pragma unroll
              for (int q = 0; q < S; q++) k[q] += (hh * A(S, q)) * y;
              yn += y;
            } else {
              yn += v;
#pragma unroll
              for (int q = 0; q < S - 1; q++) {
                float aq = 0.f;
#pragma unroll
                for (int ii = 1; ii < S; ii++) aq = ii == i ? A(ii, q) : aq;
                if (q < i) k[q] += (hh * aq) * v;
              }
            }
          }
        }
        fold(S);
        if (is_l) {
          scr = k[0];
          y = yn;
        }
        nfe += 2 * S;
        nacc++;
      }
      {   // k_1 of the first step = f(y_0)
        const float dstv = eval(is_z ? a.z_out[(size_t)b * Dp + row] : (is_l ? scr : 0.f), 0, true);
        fold(1);
        if (is_l) y -= dstv;
        nfe++;
      }
      status = 1;
    }
    if (is_l) y += a.dz_out[(size_t)b * Dp + row];   // save time 0 is ẑ₀ itself
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
      if (wv == 0 && lane < Dp) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * jj) + lane] = ov;
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
      float src = phase == PH_INIT1 ? tmp : y;
      if (phase == PH_STAGE) {
        if (SOLVER == LDE_SOLVER_TSIT5) {
          if (s > 0) {
  #define BSTAGE(S_)                                                                 \
    case S_: {                                                                       \
      float accv = ts5::A[S_][0] * k[0];                                             \
      _Pragma("unroll") for (int jj = 1; jj < S_; jj++) accv += ts5::A[S_][jj] * k[jj]; \
      src = y + h * accv;                                                            \
    } break;
            switch (s) {
              BSTAGE(1) BSTAGE(2) BSTAGE(3) BSTAGE(4) BSTAGE(5) BSTAGE(6)
              default: break;
            }
  #undef BSTAGE
            if (s == 6) yn = src;
          }
        } else if (ADJ || s < 4) {
          if (s > 0) {
            const float cs = (s == 3 ? 1.0f : 0.5f) * h;
            src = y + cs * (s == 1 ? k[0] : (s == 2 ? k[1] : k[2]));
          }
        } else {
          const float h6 = h * (1.0f / 6.0f);
          yn = y + h6 * (k[0] + 2.0f * (k[1] + k[2]) + k[3]);
          src = yn;
        }
        if (SPEC && s == 7) src = tmp;   // the speculative first evaluation of the next attempt: the state behind the jump
      }
      const float dst = eval(src, ADJ ? ((phase == PH_STAGE && s < NST) ? s : NST) : 0);
      {
        const int ks = phase == PH_K0 ? 0 : (phase == PH_INIT1 ? 1 : s);
  #pragma unroll
        for (int q = 0; q < 7; q++)
          if (q == ks) k[q] = dst;
        if (SPEC && ks == 7) scr = dst;
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
        scalarise();
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
        if (hit) tmp = jumped(yn);
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
      yn = y + h6 * (k[0] + 2.0f * (k[1] + k[2]) + k[3]);
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
      if (tid == 0 && (!coupled || b == 0)) {
        const size_t ri = (size_t)(nacc - 1) * o.rec.nseq + (coupled ? 0 : b);
        if (!ADJ) o.rec.t[ri] = t;
        o.rec.dt[ri] = hrec;
      }
      if (!ADJ && wv == 0 && lane < Dp) o.rec.y[((size_t)(nacc - 1) * B + b) * Dp + lane] = y;
    }
    if (!ADJ) {
      if (accepted) j = jd;   // (the step's dense output was written ahead: dense_output)
      if (accepted) {
        y = yn;
        k[0] = k[LAST_STAGE];
        t = tnew;
        if (last) status = 1;
      }
      s = 1;
      running = begin_step();
    } else {
      if (__builtin_amdgcn_readfirstlane((int)accepted)) {   // (workgroup-uniform: every wave takes bitwise the same decisions — say so to the compiler: a scalar branch)
        fold(NST);
        if (SPEC) {
          y = hit ? tmp : yn;
          k[0] = hit ? scr : k[6];
          // that evaluation's vectors (the spare ring slot: the seventh stage's, or the speculative one's) are the coming attempt's first stage's
          const f32x4* s4 = reinterpret_cast<const f32x4*>(ring + NST * SLOT);
          f32x4* d4 = reinterpret_cast<f32x4*>(ring);
          for (int i = tid; i < SLOT / 4; i += UT) d4[i] = s4[i];
        } else {
          y = yn;
          if (hit) y = jumped(y);
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
      for (int e = tid; e < Dp * T; e += UT) a.z_out[(size_t)Dp * ((size_t)b + (size_t)B * (e / Dp)) + (e % Dp)] = qn;
    }
    if (tid == 0) {
      const int ret = st > 1 ? st - 1 : 0;
      if (a.retcode) a.retcode[b] = ret;
      a.st_ret[b] = ret;
    }
  } else {
    if (wv == 0 && lane >= DP && lane < DP + D)   // (no usable step record: NaN, not zeros)
      a.dz0[(size_t)b * D + (lane - DP)] = st > 1 ? ((DISC && st == 1 + LDE_RET_MAXITERS) ? __int_as_float(0x7fc00000) : 0.f) : y;
    if (tid == 0) a.st_ret[b] = st > 1 ? st - 1 : 0;
    // the workgroup's row of the [workgroups × row stride] slab, flat destructure order (vec(W) column-major [out×in], then b): every
    // entry is owned by exactly one lane (a failed trajectory contributes zeros: nothing was folded after the failure … and what was
    // folded before it is dropped, as its dẑ₀ is)
    float* out = a.stage + (size_t)b * a.cap;
    const bool keep = st <= 1;
    asm volatile("s_nop 15\n\ts_nop 15");   // the last fold's MFMAs → the v_accvgpr_read of their tiles (nothing pads hidden registers)
    const int l15 = lane & 15, e4 = lane >> 4;
    auto put2 = [&](const f32x4 tv, int ti, int tj) {   // gW₂ᵀ tile (ti, tj): rows = h₁ index (row H₁: gb₂), columns = δ₂ index
      const int oo = 16 * tj + l15;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = 16 * ti + 4 * e4 + r;
        if (ti < NT && oo < H2) {
          if (i < H1) out[dm.w_off[1] + oo + H2 * i] = keep ? tv[r] : 0.f;
          else if (i == H1) out[dm.b_off[1] + oo] = keep ? tv[r] : 0.f;
        }
      }
    };
    static_for<0, NT>([&](auto tic) {
      constexpr int ti = decltype(tic)::value;
      put2(tile_read<3 * ti + 0>(), ti, 0 + wv);
      put2(tile_read<3 * ti + 1>(), ti, 4 + wv);
      put2(tile_read<3 * ti + 2>(), ti, 8 + wv);
    });
    static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; put2(tile_read<39 + q>(), 4 * q + wv, 12); });
    static_for<0, 4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const f32x4 t1 = tile_read<43 + q>(), t3 = tile_read<47 + q>();
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int uu = 16 * (4 * q + wv) + 4 * e4 + r;
        if (l15 < Dp) {
          if (uu < H1) out[dm.w_off[0] + uu + H1 * l15] = keep ? t1[r] : 0.f;
          if (uu < H2) out[dm.w_off[2] + l15 + Dp * uu] = keep ? t3[r] : 0.f;
        }
      }
    });
    if (u < H1) out[dm.b_off[0] + u] = keep ? gb1 : 0.f;
    if (wv == 0 && is_l) out[dm.b_off[2] + row] = keep ? gb3 : 0.f;
  }
  if (tid == 0) {
    const bool rep = !coupled || b == 0;   // coupled: one step sequence for the whole batch, reported once
    a.st_nfe[b] = rep ? nfe : 0;
    a.st_nacc[b] = rep ? nacc : 0;
    a.st_nrej[b] = rep ? nrej : 0;
    if (!DISC && o.rec.n && rep) o.rec.n[coupled ? 0 : b] = st > 1 ? 0 : nacc;
  }
}

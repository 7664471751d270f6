// lde_chain_bf16.h — the dense chains in NATIVE bf16 (round 3; included by lde_chain.hip).
//
// BASELINE.json configs[4] runs the encoder / decoder chains "in bf16". Round 2's mode kept every array in f32 and rounded the two
// operands of each product right before a CDNA3-shaped MFMA (v_mfma_f32_16x16x16_bf16_1k, one v_cvt_pk per operand and K-group in
// the K loop): 1.3–1.5× over f32 where the matrix cores offer 16×. Here the mode is what gfx950 is built for:
//   * weights: bf16 copies of W and Wᵀ beside the f32 masters, in the K = 32 fragment order of v_mfma_f32_16x16x32_bf16 (a lane's
//     eight consecutive-k values are ONE 16-byte load), rebuilt by the same launch that rebuilds the f32 fragments
//     (lde_chain_set_weights[_device], lde_refresh_weights);
//   * activations and δ: bf16 in LDS as transposed panels Xt[col][feature] whose byte stride ≡ 32 (mod 128) — the B operand of a
//     K-group is one conflict-free ds_read_b128 (half the LDS bytes and a quarter of the read instructions of the f32 path per flop);
//   * one v_mfma_f32_16x16x32_bf16 per (row tile, column group, K-group of 32), f32 accumulation; no conversion in the K loop (the
//     only f32 → bf16 conversions are the epilogues', and the first layer's read of the caller's f32 input);
//   * saved hidden activations (training variant) and the δ matrices the weight gradient reads are bf16 in HBM, all in ONE layout:
//     [n][feature] (a column's features contiguous) — what the epilogues produce with 8- and 16-byte stores;
//   * weight gradient: gW_lᵀ[i][o] = Σ_n a_l[i,n] δ_l[o,n] needs both operands n-contiguous per feature, i.e. the TRANSPOSE of how
//     everything is stored. k_chain_dw_b copies [n][feature] chunks into LDS as they are (wide loads) and reads the fragments
//     with ds_read_b64_tr_b16, gfx950's transposing LDS read (a 16-lane group reads a 4 × 16 block and every lane receives one
//     column of it): two of them per operand feed one v_mfma_f32_32x32x16_bf16. a_l comes straight from the saved activations
//     (a_0 from the caller's x, converted on the way in): the pullback kernel stages no a-panels at all.
// What the mode computes (the restatement in tests/test_gpu_chain_bf16.py): every product on bf16 operands with f32
// accumulation; hidden activations, the pre-skip activation and δ are STORED rounded to bf16 (round-to-nearest-even); biases,
// the skip-gradient panel, x̂, dx and dW are f32.

// Occupancy beats prefetch depth here (measured, reconstructor at N = 12 800): with the A ring two K-groups deep the 32-column kernels
// take ≤ 128 registers, TWO workgroups share a CU (16 waves: one's barrier / L2 waits are the other's MFMAs) and all 400 tiles are
// resident at once — forward 57 → 38 µs, pullback kernel 72 → 57 µs; eight K-groups deep (200+ registers, one workgroup per CU) the
// 64-column forward took 45 µs, and four deep at 128 registers spills (pullback 118 → 134 µs).
constexpr int BF_PFA = 2;   // A-fragment ring depth of the 16- / 32-column kernels (K-groups in flight)
constexpr int BF_OCC = 4;   // __launch_bounds__' second argument = waves per SIMD: 4 = two 512-thread workgroups per CU
// (the wide-input first layer's operand is staged through LDS by all waves — chain_gemm_b_gx; the same staging for the pullback's first
//  product was measured slower — 58 → 69 µs with 64-wide chunks, 73 µs with 128-wide ones — and is gone: abl/HISTORY.md §4.5)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct BfDims {
  int ldb;            // stride (elements) of the bf16 activation / δ panels: ≡ 16 (mod 64), ≥ pad32(widest hidden width)
  int ld0;            // stride of the bf16 input panel (0 with gx)
  int ldg;            // stride (floats) of the f32 skip-gradient panel
  int dl_off[MAXL];   // δ-stage: δ_l as a matrix [N][dl_w[l]] bf16 starts at N·dl_off[l] elements
  int dl_w[MAXL];     // its row width: out_l rounded up to 8 (16-byte rows; the pad entries are written as zeros)
  int dl_total;
  int fpanel;         // forward: a third panel for the activation before a skip addition (chains with skip layers; used when saving)
};

// smallest stride ≥ rows with stride ≡ 16 (mod 64) elements = 32 (mod 128) bytes: the ds_read_b128 operand pattern "16 columns ×
// 4 lane groups of 16 B" is then conflict-free in every 16-lane service group (the f32 panels' rule, in bytes)
__host__ __device__ inline int panel_stride_b(int rows) {
  int v = ((rows + 63) / 64) * 64 + 16;
  if (v - 64 >= rows) v -= 64;
  return v;
}

__device__ __forceinline__ bf16x4 to_bf4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 from_bf4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ f32x4 mfma32_b(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// Y[R × 16·CG] = M[R×K] · B[K × 16·CG] for one workgroup of 8 waves; M as bf16 K = 32 fragments in global memory (L2-resident).
// Row tiles are dealt 16 per PASS: wave w takes tiles 16p + w and 16p + w + 8 (a wave without a tile skips the pass). Per K-group
// 2 fragment loads + CG operand reads feed 2·CG MFMAs.
// BSRC 0 / 2: the B operand is a bf16 image Xt[col][k] — an LDS panel (0) or a [n][feature] matrix in global memory (2: same addressing,
//         a deeper prefetch ring): column group cg at + cg·cgstride elements, K-group k at + 32·k.
// BSRC 1: the B operand is the caller's f32 x[n][in] (first layer of a wide-input chain): eight floats are loaded and rounded per
//         K-group — the one conversion of the forward pass; a chunk beyond the row's end (K not a multiple of 32) is read from the
//         row's last chunk instead (finite values against zero weights).
// What orders the memory operations (measured with in-kernel stamps: a 13-tile layer with ONE K-group took as long as one with
// seven, ≈ 19 k cycles — the time was not in the K loop):
//   * vmcnt counts loads and stores together, in order: a fragment load issued AFTER an epilogue's global stores cannot be waited for
//     without waiting for those stores' acknowledgements. So the NEXT pass's first fragments (and its pre() loads) are issued BEFORE
//     the current pass's epilogue, and the caller's own deferred stores (`hook`: the previous layer's coalesced copy of a finished
//     panel to global memory) run right after the first pass's loads are in flight — every load is older than every store it meets;
//   * A fragments run PFA = 8 K-groups ahead (the whole K of a 200-wide layer in flight at once: L2 latency ≈ 500–700 cycles under
//     load against 2·CG × 16 cycles of MFMA per K-group).
template <int CG, int BSRC, class Pre, class Epi, class Hook>
__device__ __forceinline__ void chain_gemm_b(const __bf16* __restrict__ gfrag, int R, int K, const void* Bp, int ldb, long cgstride,
                                             Pre pre, Epi epi, Hook hook) {
  constexpr int NW = 8, PFA = CG >= 4 ? 8 : BF_PFA, PFB0 = BSRC == 0 ? 2 : 4, PFB = PFB0 < PFA ? PFB0 : PFA;
  static_assert(PFA % PFB == 0, "the B ring is indexed by i % PFB inside a PFA-unrolled block");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int RT = cdiv(R, 16), KG = cdiv(K, 32), kl = KG - 1;
  const bf16x8* A = reinterpret_cast<const bf16x8*>(gfrag) + lane;
  const int lg8 = 8 * (lane >> 4);
  const void* bpv = BSRC == 1 ? (const void*)(reinterpret_cast<const float*>(Bp) + (long)(lane & 15) * ldb + lg8)
                              : (const void*)(reinterpret_cast<const __bf16*>(Bp) + (long)(lane & 15) * ldb + lg8);
  const int col = lane & 15, rsub = 4 * (lane >> 4);
  const int npass = cdiv(RT, 2 * NW);
  auto loadB = [&](int cg, int k) -> bf16x8 {
    if (BSRC != 1) return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(bpv) + (long)cg * cgstride + k * 32);
    int off = k * 32;
    if (off + lg8 + 8 > K) off = K - 8 - lg8;
    const float* p = reinterpret_cast<const float*>(bpv) + (long)cg * cgstride + off;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    const bf16x4 bl = to_bf4(lo), bh = to_bf4(hi);
    return bf16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
  };
  typedef decltype(pre(0, 0, 0)) PreT;
  bf16x8 ra0[PFA], ra1[PFA];
  PreT p0[CG], p1[CG], q0[CG], q1[CG];
  // the loads of pass p that do not depend on the B operand: the first PFA fragments of both tiles and the epilogue's pre() values
  auto prologue = [&](int p, PreT (&d0)[CG], PreT (&d1)[CG]) {
    const int rt = 2 * NW * p + wave, rt2 = rt + NW;
    if (rt >= RT) return;
    const bf16x8* A0 = A + (size_t)rt * KG * 64;
    const bf16x8* A1 = A + (size_t)(rt2 < RT ? rt2 : rt) * KG * 64;
#pragma unroll
    for (int i = 0; i < PFA; i++) {
      const int k = min(i, kl);
      ra0[i] = A0[k * 64];
      ra1[i] = A1[k * 64];
    }
#pragma unroll
    for (int cg = 0; cg < CG; cg++) {
      d0[cg] = pre(rt * 16 + rsub, cg, col);
      d1[cg] = pre((rt2 < RT ? rt2 : rt) * 16 + rsub, cg, col);
    }
  };
  PROF_T(g0);
  prologue(0, p0, p1);
  hook();
  PROF_T(g1);
  PROF_ADD(20, g0, g1);
  for (int p = 0; p < npass; p++) {
    const int rt = 2 * NW * p + wave, rt2 = rt + NW;
    const bool va = rt < RT, vb = rt2 < RT;   // wave-uniform
    PROF_T(g2);
    f32x4 acc0[CG], acc1[CG];
#pragma unroll
    for (int cg = 0; cg < CG; cg++) acc0[cg] = acc1[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (va) {
      const bf16x8* A0 = A + (size_t)rt * KG * 64;
      const bf16x8* A1 = A + (size_t)(vb ? rt2 : rt) * KG * 64;
      bf16x8 rb[PFB][CG];
#pragma unroll
      for (int i = 0; i < PFB; i++) {
        const int k = min(i, kl);
#pragma unroll
        for (int cg = 0; cg < CG; cg++) rb[i][cg] = loadB(cg, k);
      }
      for (int kg = 0; kg < KG; kg += PFA) {
#pragma unroll
        for (int i = 0; i < PFA; i++) {
          if (kg + i < KG) {
            const bf16x8 c0 = ra0[i], c1 = ra1[i];
            bf16x8 cb[CG];
#pragma unroll
            for (int cg = 0; cg < CG; cg++) cb[cg] = rb[i % PFB][cg];
            if (kg + i + PFA < KG) {   // (K wider than the ring: refill in the loop; a 200-wide layer never gets here)
              ra0[i] = A0[(kg + i + PFA) * 64];
              ra1[i] = A1[(kg + i + PFA) * 64];
            }
            const int kb = min(kg + i + PFB, kl);
#pragma unroll
            for (int cg = 0; cg < CG; cg++) rb[i % PFB][cg] = loadB(cg, kb);
#pragma unroll
            for (int cg = 0; cg < CG; cg++) {
              acc0[cg] = mfma32_b(c0, cb[cg], acc0[cg]);
              if (vb) acc1[cg] = mfma32_b(c1, cb[cg], acc1[cg]);
            }
          }
        }
      }
    }
#if LDE_PROF
    asm volatile("" : "+v"(acc0[0][0]));   // (the stamp below waits for the accumulators)
    asm volatile("s_nop 0" ::: "memory");
#endif
    PROF_T(g3);
    if (p + 1 < npass) prologue(p + 1, q0, q1);   // before this pass's stores
    if (va) {
#pragma unroll
      for (int cg = 0; cg < CG; cg++) {
        epi(rt * 16 + rsub, cg, col, acc0[cg], p0[cg]);
        if (vb) epi(rt2 * 16 + rsub, cg, col, acc1[cg], p1[cg]);
      }
    }
#pragma unroll
    for (int cg = 0; cg < CG; cg++) {
      p0[cg] = q0[cg];
      p1[cg] = q1[cg];
    }
    PROF_T(g4);
    PROF_ADD(21, g2, g3);
    PROF_ADD(22, g3, g4);
    PROF_ADD(23, g4 - 1, g4);
  }
}
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// First layer of a wide-input chain (the encoder's 784-pixel frames): Y[R × 16·CG] = M[R×K] · bf16(x) with the tile's columns of the
// caller's f32 x[n][K] staged through LDS in K-chunks of XKC by ALL waves — each float is read from global memory once per workgroup
// (16-byte loads, a column's chunk is 512 contiguous bytes), rounded once, and every wave's MFMAs read it from LDS. (BSRC 1 of
// chain_gemm_b: every wave loads and rounds the whole operand itself — eight times the L1/L2 requests and the conversions, a global
// round trip per K-group in each wave's K loop: the feature extractor's forward took 55 µs where the reconstructor's, which writes as
// many bytes as this one reads, took 37.) Two chunk buffers; one barrier per chunk: the next chunk's global loads are issued before the
// current chunk's MFMAs and written to the other buffer behind them. A chunk past the row's end reads the row's last floats instead
// (finite values against zero weights, as in BSRC 1). Same products on the same rounded operands in the same order: the same bits.
constexpr int XKC = 128;        // chunk of the forward's first layer / of the pullback's first product (whose LDS is fuller); the LDS row
                                              // stride is the chunk + 16 elements (≡ 16 mod 64 like every bf16 panel)
__host__ __device__ inline size_t chain_xs_bytes(int cg, int kc = XKC) { return (size_t)2 * 16 * cg * (kc + 16) * 2; }
// SrcT float: the caller's f32 x[n][ldx] (rounded here); SrcT __bf16: a [n][ldx] bf16 matrix as it stands — the pullback's δ_L, which
// the same workgroup has just written (BSRC 2 of chain_gemm_b read it back per wave the same way).
template <int CG, int XKC, class SrcT, class Pre, class Epi, class Hook>
__device__ __forceinline__ void chain_gemm_b_gx(const __bf16* __restrict__ gfrag, int R, int K, const SrcT* __restrict__ xg, int ldx, __bf16* XS,
                                                Pre pre, Epi epi, Hook hook) {
  constexpr bool F32 = sizeof(SrcT) == 4;
  constexpr int XLD = XKC + 16;
  constexpr int NW = 8, NC = 16 * CG, KGC = XKC / 32, PFA = KGC, NIT = (NC * (XKC / 8) + 511) / 512;   // the A ring holds one chunk: its refills are consumed a whole chunk later, behind the staging wait
  const int lane = threadIdx.x & 63, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int RT = cdiv(R, 16), KG = cdiv(K, 32), kl = KG - 1, NCH = cdiv(K, XKC);
  const bf16x8* A = reinterpret_cast<const bf16x8*>(gfrag) + lane;
  const int col = lane & 15, rsub = 4 * (lane >> 4), lg8 = 8 * (lane >> 4);
  const int npass = cdiv(RT, 2 * NW);
  f32x4 st[NIT][F32 ? 2 : 1];   // (bf16 source: one 16-byte load, kept as it is)
  const int klast = (ldx < ((K + 7) & ~7) ? ldx : ((K + 7) & ~7)) - 8;   // the last whole 8-group inside the row
  auto gload = [&](int ch) {   // this thread's eight entries of chunk ch (column e / 16, entries 8·(e % 16) … of the chunk)
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int e = tid + 512 * it;
      if (NC * (XKC / 8) % 512 != 0 && e >= NC * (XKC / 8)) continue;
      const int c = e / (XKC / 8), sg = e % (XKC / 8);
      int k = ch * XKC + 8 * sg;
      if (k > klast) k = klast;
      const SrcT* p = xg + (size_t)c * ldx + k;
      st[it][0] = *reinterpret_cast<const f32x4*>(p);
      if (F32) st[it][F32 ? 1 : 0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + 4);
    }
  };
  auto sput = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int e = tid + 512 * it;
      if (NC * (XKC / 8) % 512 != 0 && e >= NC * (XKC / 8)) continue;
      const int c = e / (XKC / 8), sg = e % (XKC / 8);
      __bf16* dst = XS + buf * NC * XLD + c * XLD + 8 * sg;
      if (F32) {
        const bf16x4 bl = to_bf4(st[it][0]), bh = to_bf4(st[it][F32 ? 1 : 0]);
        *reinterpret_cast<bf16x8*>(dst) = bf16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
      } else
        *reinterpret_cast<f32x4*>(dst) = st[it][0];
    }
  };
  typedef decltype(pre(0, 0, 0)) PreT;
  bool hooked = false;
  for (int p = 0; p < npass; p++) {
    const int rt = 2 * NW * p + wave, rt2 = rt + NW;
    const bool va = rt < RT, vb = rt2 < RT;   // wave-uniform
    const bf16x8* A0 = A + (size_t)(va ? rt : 0) * KG * 64;
    const bf16x8* A1 = A + (size_t)(vb ? rt2 : (va ? rt : 0)) * KG * 64;
    bf16x8 ra0[PFA], ra1[PFA];
#pragma unroll
    for (int i = 0; i < PFA; i++) {
      ra0[i] = A0[min(i, kl) * 64];
      ra1[i] = A1[min(i, kl) * 64];
    }
    PreT p0[CG], p1[CG];
#pragma unroll
    for (int cg = 0; cg < CG; cg++) {
      p0[cg] = pre((va ? rt : 0) * 16 + rsub, cg, col);
      p1[cg] = pre((vb ? rt2 : (va ? rt : 0)) * 16 + rsub, cg, col);
    }
    gload(0);
    if (!hooked) { hook(); hooked = true; }
    if (p > 0) __syncthreads();   // the previous pass's last chunk is still being read
    sput(0);
    __syncthreads();
    f32x4 acc0[CG], acc1[CG];
#pragma unroll
    for (int cg = 0; cg < CG; cg++) acc0[cg] = acc1[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < NCH; ch++) {
      if (ch + 1 < NCH) gload(ch + 1);
      const __bf16* xb = XS + (ch & 1) * NC * XLD + col * XLD + lg8;
#pragma unroll
      for (int kgl = 0; kgl < KGC; kgl++) {
        const int kg = ch * KGC + kgl;
        if (kg < KG && va) {
          const bf16x8 c0 = ra0[kgl % PFA], c1 = ra1[kgl % PFA];
          if (kg + PFA < KG) {
            ra0[kgl % PFA] = A0[(kg + PFA) * 64];
            ra1[kgl % PFA] = A1[(kg + PFA) * 64];
          }
#pragma unroll
          for (int cg = 0; cg < CG; cg++) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(xb + cg * 16 * XLD + kgl * 32);
            acc0[cg] = mfma32_b(c0, b, acc0[cg]);
            if (vb) acc1[cg] = mfma32_b(c1, b, acc1[cg]);
          }
        }
      }
      if (ch + 1 < NCH) sput((ch + 1) & 1);
      __syncthreads();
    }
    if (va) {
#pragma unroll
      for (int cg = 0; cg < CG; cg++) {
        epi(rt * 16 + rsub, cg, col, acc0[cg], p0[cg]);
        if (vb) epi(rt2 * 16 + rsub, cg, col, acc1[cg], p1[cg]);
      }
    }
  }
}

struct ChainFwdArgsB {
  const float* x;
  float* y;
  const __bf16* fragb;
  const float* Wflat;
  long long N;
  __bf16* saved;       // training: hidden activations [n][h] bf16 (the regions of ChainDims::sv_pre, in elements)
  const float* mse_t;  // as in ChainFwdArgs: Σ (y − mse_t)² of the tile's own columns → mse_part[tile]
  float* mse_part;
  // lde_chain_forward_save_mse_delta: the last epilogue also leaves δ_L′ = dk2·(y − mse_t)·act′(y), dk2 = 2·scale, as the last layer's bf16
  // δ matrix [n][dl_w] for the pullback (which multiplies dx and dW by the loss's cotangent g at the end): the pullback then reads neither
  // y nor the target, and y itself is only stored when the caller wants it (y == nullptr)
  __bf16* dL;
  float dk2;
  int dlw;             // row width of that matrix (BfDims::dl_w of the last layer: the output width padded to whole K-groups; the pad is written as zeros)
};

// zero the panels, copy the biases, and (no gx) the tile's input columns rounded to bf16
template <int CG>
__device__ __forceinline__ void chain_load_tile_b(const ChainDims& cd, const BfDims& bd, const float* x, long long n0, long long N,
                                                  __bf16* X0, float* biasc, const float* Wflat, int nzero16, void* zero_base) {
  constexpr int NC = 16 * CG;
  const int tid = threadIdx.x;
  for (int i = tid; i < nzero16; i += 512) reinterpret_cast<f32x4*>(zero_base)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // pad rows / columns: finite
  __syncthreads();
  const MlpDims& dm = cd.dm;
  for (int l = 0; l < dm.nL; l++)
    for (int i = tid; i < dm.sizes[l + 1]; i += 512) biasc[dm.bias_lin[l] + i] = Wflat[dm.b_off[l] + i];
  const int in0 = dm.sizes[0];
  if (!cd.gx) {
    for (int e = tid; e < NC * in0; e += 512) {
      const int c = e / in0, r = e - c * in0;
      if (n0 + c < N) X0[c * bd.ld0 + r] = (__bf16)x[(size_t)(n0 + c) * in0 + r];
    }
  }
  __syncthreads();
}

// copy the first `rows` features of this tile's columns from a bf16 LDS panel to a [n][rows] matrix in global memory: 16-byte pieces,
// consecutive lanes on consecutive pieces of a column (fully coalesced); rows % 8 != 0: element-wise
template <int CG>
__device__ __forceinline__ void copy_panel_out_b(const __bf16* panel, int ld, int rows, __bf16* dst, long long n0, long long N) {
  constexpr int NC = 16 * CG;
  if ((rows & 7) == 0) {
    const int chunks = rows / 8;
    for (int e = threadIdx.x; e < NC * chunks; e += 512) {
      const int c = e / chunks, ch = e - c * chunks;
      if (n0 + c < N) *reinterpret_cast<bf16x8*>(dst + (size_t)(n0 + c) * rows + 8 * ch) = *reinterpret_cast<const bf16x8*>(panel + c * ld + 8 * ch);
    }
  } else {
    for (int e = threadIdx.x; e < NC * rows; e += 512) {
      const int c = e / rows, r = e - c * rows;
      if (n0 + c < N) dst[(size_t)(n0 + c) * rows + r] = panel[c * ld + r];
    }
  }
}

// one hidden layer: Y = [Xin +] act(W·Xin + b) into a bf16 LDS panel; F (skip layers, training): the activation before the addition.
// Nothing goes to global memory here: the saved-activation matrices are written from the panels, coalesced, by the NEXT product's hook.
template <int CG, int BSRC, class Hook>
__device__ __forceinline__ void chain_hidden_layer_b(const ChainDims& cd, const BfDims& bd, int l, const __bf16* fragb, const float* biasc,
                                                     const void* Xin, int ldx, __bf16* Y, __bf16* F, Hook hook, __bf16* XS = nullptr) {
  const MlpDims& dm = cd.dm;
  const int in = dm.sizes[l], out = dm.sizes[l + 1], actk = cd.act[l], skip = BSRC == 1 ? 0 : cd.skip[l], ldh = bd.ldb;
  const float* bias = biasc + dm.bias_lin[l];
  auto epi =             [&](int row0, int cg, int col, f32x4 v, NoPre) {
                           const int c = cg * 16 + col;
                           const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + min(row0, ((out + 3) & ~3) - 4));   // (bias blocks are padded to 4; rows beyond `out` are masked below)
                           f32x4 r = cact4(actk, v + b4);
#pragma unroll
                           for (int q = 0; q < 4; q++) r[q] = row0 + q < out ? r[q] : 0.f;
                           if (skip) {
                             if (F) *reinterpret_cast<bf16x4*>(F + c * ldh + row0) = to_bf4(r);   // before the skip addition
                             r += from_bf4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(Xin) + c * ldx + row0));
                           }
                           *reinterpret_cast<bf16x4*>(Y + c * ldh + row0) = to_bf4(r);
                         };
  if (BSRC == 1 && XS)
    chain_gemm_b_gx<CG, XKC>(fragb + bf_frag_off(dm, l, false), out, in, reinterpret_cast<const float*>(Xin), in, XS, [](int, int, int) { return NoPre{}; }, epi, hook);
  else chain_gemm_b<CG, BSRC>(fragb + bf_frag_off(dm, l, false), out, in, Xin, ldx, 16L * ldx, [](int, int, int) { return NoPre{}; }, epi, hook);
}

template <int CG>
__device__ __forceinline__ void chain_forward_b_body(const ChainDims& cd, const BfDims& bd, const ChainFwdArgsB& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float csm[];
  constexpr int NC = 16 * CG;
  const MlpDims& dm = cd.dm;
  const int nL = dm.nL, ldh = bd.ldb;
  // H0 | H1 | [F: pre-skip activations, training with skip layers only] | biases; the input panel lives in H1's space (layer 0 reads it
  // and writes H0; H1 is first written by layer 1). Its pad rows are zero like every panel's.
  __bf16* H0 = reinterpret_cast<__bf16*>(csm);
  __bf16* H1 = H0 + NC * ldh;
  __bf16* X0 = H1;
  bool any_skip = false;
  for (int l = 0; l + 1 < nL; l++) any_skip = any_skip || cd.skip[l];
  __bf16* F = (a.saved && any_skip && bd.fpanel) ? H1 + NC * ldh : nullptr;
  float* biasc = reinterpret_cast<float*>(H1 + NC * ldh + (bd.fpanel ? NC * ldh : 0));
  // wide input: the chunk buffers of the first layer's staged operand (chain_gemm_b_gx) behind the biases; K ≥ 8 and 16-byte rows
  __bf16* XS = (cd.gx && dm.sizes[0] >= 8 && dm.sizes[0] % 4 == 0) ? reinterpret_cast<__bf16*>(biasc + ((dm.nbias + 3) & ~3)) : nullptr;
  int dup;
  const long long n0 = chain_tile_start(cd, NC, a.N, &dup, bx);
  PROF_T(pc0);
  chain_load_tile_b<CG>(cd, bd, a.x, n0, a.N, X0, biasc, a.Wflat, (2 * NC * ldh) / 8, csm);
  PROF_T(pc1);
  PROF_ADD(0, pc0, pc1);
  const void* Xin = X0;
  int ldx = bd.ld0;
  const float* xg = a.x + (size_t)n0 * dm.sizes[0];   // gx: column c of the tile at xg + c·in
  // layer l's output panel (and F) go to the saved matrices from the hook of the NEXT product, behind that product's first loads
  auto save_prev = [&](int l) {   // l: the layer whose panels are complete (−1: none)
    if (l < 0 || !a.saved) return;
    const int h = dm.sizes[l + 1];
    const __bf16* Yl = (l & 1) ? H1 : H0;
    __bf16* base = a.saved + (size_t)a.N * cd.sv_pre[l];
    copy_panel_out_b<CG>(Yl, ldh, h, base, n0, a.N);
    if (cd.skip[l]) {
      copy_panel_out_b<CG>(F, ldh, h, base + (size_t)a.N * h, n0, a.N);
      if (l + 2 < nL && cd.skip[l + 1]) __syncthreads();   // the next layer's epilogues write F again
    }
  };
  for (int l = 0; l + 1 < nL; l++) {
    __bf16* Y = (l & 1) ? H1 : H0;
    PROF_T(pl0);
    auto hook = [&]() { save_prev(l - 1); };
    if (l == 0 && cd.gx) chain_hidden_layer_b<CG, 1>(cd, bd, 0, a.fragb, biasc, xg, dm.sizes[0], Y, F, hook, XS);
    else chain_hidden_layer_b<CG, 0>(cd, bd, l, a.fragb, biasc, Xin, ldx, Y, F, hook);
    PROF_T(pl1);
    __syncthreads();
    PROF_T(pl2);
    PROF_ADD(2 + 2 * l, pl0, pl1);
    PROF_ADD(3 + 2 * l, pl1, pl2);
    Xin = Y;
    ldx = ldh;
  }
  PROF_T(pz0);
  if (a.y || a.dL) {  // last layer: f32 straight to HBM (neither: the pullback's own forward pass, which only wants the saved matrices)
    const int l = nL - 1, in = dm.sizes[l], out = dm.sizes[l + 1], actk = cd.act[l];
    const float* bias = biasc + dm.bias_lin[l];
    const bool vec = (out & 3) == 0;
    float msum = 0.f;
    auto epi_last = [&](int row0, int cg, int col, f32x4 v, NoPre) {
      const long long n = n0 + cg * 16 + col;
      if (n >= a.N || row0 >= out) return;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + row0);
      const f32x4 r = cact4(actk, v + b4);
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
      if (a.dL) {   // (out % 8 == 0 and a 16-byte-aligned target: checked by the host) — the same arithmetic, in the same order, as the pullback's own δ_L with g = 1
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.mse_t + (size_t)n * out + row0);
        const f32x4 a4 = cact_grad_out4(actk, r);
        f32x4 d;
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = __fmul_rn(a.dk2, r[q] - t4[q]) * a4[q];
        __bf16* dp = a.dL + (size_t)n * a.dlw;
        *reinterpret_cast<bf16x4*>(dp + row0) = to_bf4(d);
        if (row0 + 4 >= out)   // the lane of the row's last four features also zeroes the pad up to the matrix's row width
          for (int r = out; r < a.dlw; r += 4) *reinterpret_cast<bf16x4*>(dp + r) = to_bf4(f32x4{0.f, 0.f, 0.f, 0.f});
      }
      if (!a.y) return;
      float* yp = a.y + (size_t)n * out + row0;
      if (vec) *reinterpret_cast<f32x4*>(yp) = r;
      else {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (row0 + q < out) yp[q] = r[q];
      }
    };
    auto nopre = [](int, int, int) { return NoPre{}; };
    auto hook = [&]() { save_prev(nL - 2); };
    if (nL == 1 && cd.gx) chain_gemm_b<CG, 1>(a.fragb + bf_frag_off(dm, l, false), out, in, xg, in, 16L * in, nopre, epi_last, hook);
    else chain_gemm_b<CG, 0>(a.fragb + bf_frag_off(dm, l, false), out, in, Xin, ldx, 16L * ldx, nopre, epi_last, hook);
    if (a.mse_t) {
      const float t = chain_wg_sum(msum, csm);   // (the panels are free behind the barrier inside)
      if (threadIdx.x == 0) a.mse_part[bx] = t;
    }
  } else
    save_prev(nL - 2);
  PROF_T(pz1);
  PROF_ADD(2 + 2 * (nL - 1), pz0, pz1);
  PROF_ADD(40, pc0, pz1);
}
struct ChainBDims { ChainDims cd; BfDims bd; };
template <int CG>
__global__ void __launch_bounds__(512, (CG <= 2 ? BF_OCC : 1)) k_chain_forward_b(ChainDims cd, BfDims bd, ChainFwdArgsB a) {
  chain_forward_b_body<CG>(cd, bd, a, blockIdx.x);
}
template <int CG>
__global__ void __launch_bounds__(512, (CG <= 2 ? BF_OCC : 1)) k_chain_forward_b_group(GroupTable<ChainBDims, ChainFwdArgsB> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  chain_forward_b_body<CG>(g.dims[j].cd, g.dims[j].bd, g.args[j], blockIdx.x - g.start[j]);
}

struct ChainBwdArgsB {
  const float* x;
  const float* y;
  const float* dy;
  float* dx;
  const __bf16* fragTb;
  const float* Wflat;
  __bf16* dstage;       // δ_l matrices [n][dl_w[l]] bf16 (BfDims::dl_off), read by k_chain_dw_b
  long long N;
  const __bf16* saved;  // hidden activations written by k_chain_forward_b
  const float* dy2;     // the output gradient is (dy + dy2) + dy3 (nullptr: absent), as in ChainBwdArgs
  const float* dy3;
  const float* mse_t;   // as in ChainBwdArgs: the first source is 2·(g·scale)·(y − mse_t) instead of dy
  const float* mse_g;
  float mse_scale;
  int dl_ready;         // δ_L′ (without the loss's cotangent g) is already in dstage (ChainFwdArgsB::dL): no first pass over y / target / dy …
  const float* gs;      // … and dx is multiplied by g = gs[0] here (dW: in k_chain_dw_b); nullptr: 1
};

struct PreH { f32x4 h; };

// copy a bf16 LDS panel (columns of this tile, `rows` features each, zero beyond) to the [n][w] matrix in global memory: 16-byte
// chunks, consecutive lanes on consecutive chunks of a column
template <int CG>
__device__ __forceinline__ void stage_delta_b(const __bf16* panel, int ld, int rows, __bf16* dst, int w, long long n0, long long N) {
  constexpr int NC = 16 * CG;
  const int chunks = w / 8;
  for (int e = threadIdx.x; e < NC * chunks; e += 512) {
    const int c = e / chunks, ch = e - c * chunks;
    if (n0 + c >= N) continue;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(panel + c * ld + 8 * ch);
    if (8 * ch + 8 > rows) {
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (8 * ch + q >= rows) v[q] = (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(dst + (size_t)(n0 + c) * w + 8 * ch) = v;
  }
}

template <int CG>
__device__ __forceinline__ void chain_backward_b_body(const ChainDims& cd, const BfDims& bd, const ChainBwdArgsB& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) float csm[];
  constexpr int NC = 16 * CG;
  const MlpDims& dm = cd.dm;
  const int nL = dm.nL, ldh = bd.ldb, ldg = bd.ldg, tid = threadIdx.x;
  __bf16* P0 = reinterpret_cast<__bf16*>(csm);
  __bf16* P1 = P0 + NC * ldh;
  float* G = reinterpret_cast<float*>(P1 + NC * ldh);
  int dup;
  const long long n0 = chain_tile_start(cd, NC, a.N, &dup, bx);
  for (int i = tid; i < (2 * NC * ldh) / 8 + (NC * ldg) / 4; i += 512) reinterpret_cast<f32x4*>(csm)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- δ_L = dy ⊙ act'(y) from the caller's arrays → the δ-stage matrix of the last layer (bf16) -----------------------
  const int L1 = nL - 1;
  const float gsv = a.gs ? a.gs[0] : 1.0f;
  if (!a.dl_ready) {
    const int out = dm.sizes[nL], w = bd.dl_w[L1], actk = cd.act[L1], chunks = w / 8;
    __bf16* dL = a.dstage + (size_t)a.N * bd.dl_off[L1];
    const bool vec = (out & 3) == 0;
    for (int e = tid; e < NC * chunks; e += 512) {
      const int c = e / chunks, ch = e - c * chunks;
      const long long n = n0 + c;
      if (n >= a.N) continue;
      const float* dyp = a.dy + (size_t)n * out + 8 * ch;
      const float* yp = a.y + (size_t)n * out + 8 * ch;
      bf16x8 d;
      if (vec && 8 * ch + 8 <= out) {
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(yp), f1 = *reinterpret_cast<const f32x4*>(yp + 4);
        f32x4 g0, g1;
        if (a.mse_t) {
          const float k2 = 2.0f * (a.mse_g[0] * a.mse_scale);
          const float* tp = a.mse_t + (size_t)n * out + 8 * ch;
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            g0[q] = __fmul_rn(k2, f0[q] - t0[q]);
            g1[q] = __fmul_rn(k2, f1[q] - t1[q]);
          }
        } else {
          g0 = *reinterpret_cast<const f32x4*>(dyp);
          g1 = *reinterpret_cast<const f32x4*>(dyp + 4);
        }
        if (a.dy2) {
          const float* q2 = a.dy2 + (size_t)n * out + 8 * ch;
          g0 += *reinterpret_cast<const f32x4*>(q2);
          g1 += *reinterpret_cast<const f32x4*>(q2 + 4);
        }
        if (a.dy3) {
          const float* q3 = a.dy3 + (size_t)n * out + 8 * ch;
          g0 += *reinterpret_cast<const f32x4*>(q3);
          g1 += *reinterpret_cast<const f32x4*>(q3 + 4);
        }
        const f32x4 a0 = cact_grad_out4(actk, f0), a1 = cact_grad_out4(actk, f1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          d[q] = (__bf16)(g0[q] * a0[q]);
          d[4 + q] = (__bf16)(g1[q] * a1[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          float g = 0.f;
          if (8 * ch + q < out) {
            g = a.mse_t ? __fmul_rn(2.0f * (a.mse_g[0] * a.mse_scale), yp[q] - a.mse_t[(size_t)n * out + 8 * ch + q]) : dyp[q];
            if (a.dy2) g += a.dy2[(size_t)n * out + 8 * ch + q];
            if (a.dy3) g += a.dy3[(size_t)n * out + 8 * ch + q];
            g *= cact_grad_out(actk, yp[q]);
          }
          d[q] = (__bf16)g;
        }
      }
      *reinterpret_cast<bf16x8*>(dL + (size_t)n * w + 8 * ch) = d;
    }
  }
  __syncthreads();   // s_waitcnt vmcnt(0) + barrier: the rows are in L2 for the read-back below; the LDS panels are zero

  // ---- δ down the chain ------------------------------------------------------------------------------------------------
  const __bf16* Dcur = nullptr;   // LDS panel holding δ_l for l < L1
  int pend = -1;                  // layer whose δ panel still has to be copied to its matrix
  auto stage_prev = [&]() {
    if (pend < 0) return;
    stage_delta_b<CG>(Dcur, ldh, dm.sizes[pend + 1], a.dstage + (size_t)a.N * bd.dl_off[pend], bd.dl_w[pend], n0, a.N);
    pend = -1;
  };
  for (int l = L1; l >= 0; l--) {
    const int in = dm.sizes[l], out = dm.sizes[l + 1];
    const int skl = cd.skip[l];
    const __bf16* fragT = a.fragTb + bf_frag_off(dm, l, true);
    const __bf16* Bglb = a.dstage + (size_t)a.N * bd.dl_off[l] + (size_t)n0 * bd.dl_w[l];   // δ_l rows of this tile (only used for l == L1)
    if (l > 0) {
      __bf16* Dn = (l & 1) ? P1 : P0;
      const int actp = cd.act[l - 1], skp = cd.skip[l - 1];
      // activation output of layer l-1 (before its skip addition): from the saved matrices, 8 bytes per (column, 4 features)
      const __bf16* hsv = a.saved + (size_t)a.N * cd.sv_pre[l - 1] + (skp ? (size_t)a.N * in : 0);
      const bool hvec = (in & 3) == 0;
      auto pre = [&](int row0, int cg, int col) {
        PreH p;
        p.h = f32x4{0.f, 0.f, 0.f, 0.f};
        const long long n = n0 + cg * 16 + col;
        if (n < a.N && row0 < in) {
          const __bf16* hp = hsv + (size_t)n * in + row0;
          if (hvec) p.h = from_bf4(*reinterpret_cast<const bf16x4*>(hp));
          else {
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (row0 + q < in) p.h[q] = (float)hp[q];
          }
        }
        return p;
      };
      auto epi = [&](int row0, int cg, int col, f32x4 v, PreH p) {
        const int c = cg * 16 + col;
        f32x4 g = v;
        if (skl) g += *reinterpret_cast<const f32x4*>(G + c * ldg + row0);
        if (skp) *reinterpret_cast<f32x4*>(G + c * ldg + row0) = g;   // layer l-1 adds it back to what flows through it
        f32x4 d = cact_grad_out4(actp, p.h);
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = row0 + q < in ? g[q] * d[q] : 0.f;
        *reinterpret_cast<bf16x4*>(Dn + c * ldh + row0) = to_bf4(d);
      };
      if (l == L1) chain_gemm_b<CG, 2>(fragT, in, out, Bglb, bd.dl_w[l], 16L * bd.dl_w[l], pre, epi, stage_prev);
      else chain_gemm_b<CG, 0>(fragT, in, out, Dcur, ldh, 16L * ldh, pre, epi, stage_prev);
      __syncthreads();
      Dcur = Dn;
      pend = l - 1;   // δ_{l-1} is complete in Dn: it goes to its matrix from the next product's hook (behind that product's first loads)
    } else if (a.dx) {
      auto pre = [](int, int, int) { return NoPre{}; };
      auto epi = [&](int row0, int cg, int col, f32x4 v, NoPre) {
        const int c = cg * 16 + col;
        const long long n = n0 + c;
        f32x4 g = v;
        if (skl) g += *reinterpret_cast<const f32x4*>(G + c * ldg + row0);
        if (n < a.N) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (row0 + q < in) a.dx[(size_t)n * in + row0 + q] = a.gs ? g[q] * gsv : g[q];
        }
      };
      if (l == L1) chain_gemm_b<CG, 2>(fragT, in, out, Bglb, bd.dl_w[l], 16L * bd.dl_w[l], pre, epi, stage_prev);
      else chain_gemm_b<CG, 0>(fragT, in, out, Dcur, ldh, 16L * ldh, pre, epi, stage_prev);
    }
  }
  stage_prev();   // (no product followed the last δ: the input gradient was not asked for)
}
template <int CG>
__global__ void __launch_bounds__(512, (CG <= 2 ? BF_OCC : 1)) k_chain_backward_b(ChainDims cd, BfDims bd, ChainBwdArgsB a) {
  chain_backward_b_body<CG>(cd, bd, a, blockIdx.x);
}
template <int CG>
__global__ void __launch_bounds__(512, (CG <= 2 ? BF_OCC : 1)) k_chain_backward_b_group(GroupTable<ChainBDims, ChainBwdArgsB> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  chain_backward_b_body<CG>(g.dims[j].cd, g.dims[j].bd, g.args[j], blockIdx.x - g.start[j]);
}

// ---- weight gradient: gW_lᵀ[i][o] = Σ_n a_l[i,n] δ_l[o,n] over [n][feature] matrices, transposing LDS reads ---------------------------
// grid (K-split parts, jobs); a job = a block of ≤ 8·NDW 32×32 tiles of one layer's gWᵀ (dw_decode, as k_mlp_dw). The workgroup
// walks its share of the n range in chunks of NK = 32 rows: the chunk's rows of a_l (features of the job's input tiles) and of δ_l
// (features of its output tiles) are copied into LDS as they are — [n][feature], row stride ≡ 64 (mod 128) bytes, which makes the four
// 64-byte row pieces a 32-lane half of ds_read_b64_tr_b16 touches fall on disjoint banks — and every wave feeds its tiles: per
// 16 rows of n, two transposing reads per operand (rows 8h + 4r .. + 3, r = 0, 1: the lane receives its feature's values at those
// four n) make the eight-k operand of one v_mfma_f32_32x32x16_bf16. Bias gradients: column sums of the δ image, by the jobs that
// hold the first input tile. Output: the (part) slab in accumulator-fragment order — k_reduce_tiles adds the parts in a fixed order.
struct DwArgsB {
  const float* x;          // a_0 [N][in_0] f32 (rounded to bf16 on the way into LDS)
  const __bf16* saved;     // a_l = hidden l-1 [N][h] (ChainDims::sv_pre)
  const __bf16* dstage;    // δ_l [N][dl_w[l]]
  float* slab;             // [parts][slab_n]
  long long N;
  const float* gs;         // the δ matrices lack the loss's cotangent g (ChainBwdArgsB::dl_ready): everything written is multiplied by gs[0]; nullptr: 1
};

__host__ __device__ inline int tr_stride_bytes(int feats) {   // row stride of an image of `feats` bf16 features: smallest ≥ 2·feats that is ≡ 64 (mod 128)
  return ((2 * feats + 63) / 128) * 128 + 64;
}
constexpr int DWB_NK = 64;   // rows of n per chunk of k_chain_dw_b: a chunk's MFMA work (4 K-steps) against one global round trip
// LDS bytes of k_chain_dw_b: the widest job's two images
inline size_t dw_b_lds_bytes(const MlpDims& dm, int ndw) {
  size_t mx = 0;
  for (int z = 0, n = dw_jobs(dm, ndw); z < n; z++) {
    const DwJob j = dw_decode(dm, z, 8 * ndw);
    mx = std::max(mx, (size_t)DWB_NK * (tr_stride_bytes(32 * (j.i1 - j.i0)) + tr_stride_bytes(32 * (j.o1 - j.o0))));
  }
  return mx;
}

template <int DW_NDW>
__device__ __forceinline__ void chain_dw_b_body(const ChainDims& cd, const BfDims& bd, const DwArgsB& a, const int part, const int KS, const int jobz) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  constexpr int NK = DWB_NK;
  const MlpDims& dm = cd.dm;
  const DwJob jb = dw_decode(dm, jobz, 8 * DW_NDW);
  const int l = jb.l, in = dm.sizes[l], out = dm.sizes[l + 1];
  const int IT = cdiv(in, 32), nit = jb.i1 - jb.i0, ntile = (jb.o1 - jb.o0) * nit;
  const int na = 32 * nit, nd = 32 * (jb.o1 - jb.o0), ra0 = 32 * jb.i0, rd0 = 32 * jb.o0;
  const int sa = tr_stride_bytes(na), sd = tr_stride_bytes(nd);
  unsigned char* ia = reinterpret_cast<unsigned char*>(dsm);
  unsigned char* id = ia + NK * sa;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // operand addressing of the transposing read: 16-lane group g = lane >> 4 → (k half h = g >> 1, feature half g & 1); lane 4q + p of
  // the group supplies the address of row q, features 4p .. 4p + 3 of the block
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lrow = 8 * (g >> 1) + q4, lfeat = 16 * (g & 1) + 4 * p4;
  f32x16 acc[DW_NDW];
  int aoff[DW_NDW], doff[DW_NDW];
#pragma unroll
  for (int m = 0; m < DW_NDW; m++) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
    const int t = wave + 8 * m;
    const int otl = t / nit, itl = t - otl * nit;
    aoff[m] = t < ntile ? lrow * sa + 2 * (itl * 32 + lfeat) : -1;
    doff[m] = lrow * sd + 2 * (otl * 32 + lfeat);
  }
  float bsum[2] = {0.f, 0.f};
  // the operands' matrices in global memory: a_l rows of width aw, δ_l rows of width dw
  const bool a_f32 = l == 0;
  const int aw = in;
  const __bf16* ab = a_f32 ? nullptr : a.saved + (size_t)a.N * cd.sv_pre[l - 1];
  const __bf16* db = a.dstage + (size_t)a.N * bd.dl_off[l];
  const int dw = bd.dl_w[l];
  const bool avec = (aw & 7) == 0;
  const long long nchunks = (a.N + NK - 1) / NK;
  // The next chunk's first PQ pieces per thread and image travel into registers while the current chunk is multiplied (a piece = 16
  // bytes = 8 features of one n). Every load is UNCONDITIONAL, from a clamped (always valid) address — a load under a lane condition is
  // waited for where the branches merge, i.e. right where it was issued (in-kernel stamps: 2 760 cycles per chunk "issuing" the next
  // chunk's loads) — and the zero padding (rows beyond N, features beyond the layer) is applied when the piece is stored to LDS. An f32
  // source (x) needs two 16-byte loads per piece and is rounded at store time; a bf16 source issues the second load at the same address.
  constexpr int PQ = 4;
  const int apr = na / 8, dpr = nd / 8, npa = NK * apr, npd = NK * dpr;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 qlo[PQ], qhi[PQ], qdl[PQ];
  const char* abase = a_f32 ? reinterpret_cast<const char*>(a.x) : reinterpret_cast<const char*>(ab);
  const int aes = a_f32 ? 4 : 2;                                   // bytes per element of the a source
  const int flast = in >= 8 ? in - 8 : 0, dflast = dw - 8;
  // per-thread piece geometry, fixed over the chunks (two integer divisions per piece: ≈ 1 300 cycles per chunk when they sat in the loop)
  int pa_r[PQ], pa_c8[PQ], pd_r[PQ], pd_c8[PQ];
  size_t pa_off[PQ], pd_off[PQ];
#pragma unroll
  for (int i = 0; i < PQ; i++) {
    const int e = tid + 512 * i;
    const int ea = min(e, npa - 1), ed = min(e, npd - 1);
    pa_r[i] = ea / apr;
    pa_c8[i] = ea - pa_r[i] * apr;
    pd_r[i] = ed / dpr;
    pd_c8[i] = ed - pd_r[i] * dpr;
    pa_off[i] = (size_t)min(ra0 + 8 * pa_c8[i], flast) * aes;
    pd_off[i] = (size_t)min(rd0 + 8 * pd_c8[i], dflast) * 2;
  }
  const size_t arow_b = (size_t)aw * aes, drow_b = (size_t)dw * 2;
  auto fetch = [&](long long chn) {
    const long long nb = chn * NK;
#pragma unroll
    for (int i = 0; i < PQ; i++) {
      // (a width that is not a multiple of 8 takes the element-wise path below: its prefetch loads are aimed at valid δ bytes and unused)
      const char* p = avec ? abase + (size_t)min(nb + pa_r[i], a.N - 1) * arow_b + pa_off[i] : reinterpret_cast<const char*>(db);
      const char* p2 = p + ((a_f32 && avec) ? 16 : 0);
      asm volatile("" : "+v"(p2));   // (opaque: with a visible "p2 == p" the compiler turns the second load into a conditional COPY of the first — and waits for it here)
      qlo[i] = *reinterpret_cast<const f32x4*>(p);
      qhi[i] = *reinterpret_cast<const f32x4*>(p2);
      qdl[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(db) + (size_t)min(nb + pd_r[i], a.N - 1) * drow_b + pd_off[i]);
    }
  };
  // the bias gradient for free: the first pad feature of the a image (feature index `in`, present when in is not a multiple of 32) is set
  // to 1 for every n < N — row `in` of gWᵀ's last input tile then accumulates Σ_n δ_l[o,n]
  const bool ones_row = (in & 31) != 0;
  const bool do_bias_sum = !ones_row && jb.i0 == 0;                // explicit column sums only when there is no pad row
  const bool do_bias_row = ones_row && jb.i1 == IT;
  if (part < nchunks) fetch(part);
  for (long long ch = part; ch < nchunks; ch += KS) {
    const long long nb = ch * NK;
    PROF_T(d0);
    // ---- the chunk into LDS: 16-byte pieces; rows beyond N and features beyond the layer are zeros
#pragma unroll
    for (int i = 0; i < PQ; i++) {
      const int e = tid + 512 * i;
      if (avec && e < npa) {
        const int r = pa_r[i], c8 = pa_c8[i], f = ra0 + 8 * c8;
        const bool ok = nb + r < a.N && f < in;
        f32x4 v = qlo[i];
        if (a_f32) {
          const bf16x4 lo = to_bf4(qlo[i]), hi = to_bf4(qhi[i]);
          v = __builtin_bit_cast(f32x4, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        }
        v = ok ? v : z4;
        if (ones_row && f == (in & ~7) && nb + r < a.N) {   // (in % 8 == 0 here: the piece [in, in + 8) is all padding)
          bf16x8 o = __builtin_bit_cast(bf16x8, z4);
          o[0] = (__bf16)1.0f;
          v = __builtin_bit_cast(f32x4, o);
        }
        *reinterpret_cast<f32x4*>(ia + r * sa + 16 * c8) = v;
      }
      if (e < npd) {
        const int r = pd_r[i], c8 = pd_c8[i];
        const bool ok = nb + r < a.N && rd0 + 8 * c8 < dw;
        *reinterpret_cast<f32x4*>(id + r * sd + 16 * c8) = ok ? qdl[i] : z4;
      }
    }
    for (int e = tid + (avec ? 512 * PQ : 0); e < npa; e += 512) {   // wider jobs' remaining pieces, and every piece of a layer whose width is not a multiple of 8
      const int r = e / apr, c8 = e - r * apr;
      const long long n = nb + r;
      const int f = ra0 + 8 * c8;
      bf16x8 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      if (n < a.N && f < in) {
        if (a_f32) {
          const float* xp = a.x + (size_t)n * aw + f;
          if (avec) {
            const bf16x4 lo = to_bf4(*reinterpret_cast<const f32x4*>(xp)), hi = to_bf4(*reinterpret_cast<const f32x4*>(xp + 4));
            v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
              if (f + q < in) v[q] = (__bf16)xp[q];
          }
        } else {
          const __bf16* hp = ab + (size_t)n * aw + f;
          if (avec) v = *reinterpret_cast<const bf16x8*>(hp);
          else {
#pragma unroll
            for (int q = 0; q < 8; q++)
              if (f + q < in) v[q] = hp[q];
          }
        }
      }
      if (ones_row && n < a.N && f <= in && in < f + 8) v[in - f] = (__bf16)1.0f;
      *reinterpret_cast<bf16x8*>(ia + r * sa + 16 * c8) = v;
    }
    for (int e = tid + 512 * PQ; e < npd; e += 512) {
      const int r = e / dpr, c8 = e - r * dpr;
      const long long n = nb + r;
      const int f = rd0 + 8 * c8;
      f32x4 v = z4;
      if (n < a.N && f < dw) v = *reinterpret_cast<const f32x4*>(db + (size_t)n * dw + f);
      *reinterpret_cast<f32x4*>(id + r * sd + 16 * c8) = v;
    }
    PROF_T(d1);
    __syncthreads();
    PROF_T(d2);
    if (ch + KS < nchunks) fetch(ch + KS);
    PROF_T(d3);
#pragma unroll
    for (int m = 0; m < DW_NDW; m++) {
      if (aoff[m] >= 0) {   // wave-uniform: EXEC stays all ones for the transposing reads
#pragma unroll
        for (int ks = 0; ks < NK / 16; ks++) {
          typedef short s16x4_ __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
          const unsigned char* pa = ia + aoff[m] + ks * 16 * sa;
          const unsigned char* pd = id + doff[m] + ks * 16 * sd;
          const s16x4_ a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa));
          const s16x4_ a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * sa));
          const s16x4_ d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pd));
          const s16x4_ d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pd + 4 * sd));
          const bf16x8 av = __builtin_bit_cast(bf16x8, (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
          const bf16x8 dv = __builtin_bit_cast(bf16x8, (s16x8){d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]});
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dv, acc[m], 0, 0, 0);
        }
      }
    }
#if LDE_PROF
    asm volatile("" : "+v"(acc[0][0]));
#endif
    PROF_T(d4);
    if (do_bias_sum) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int row = tid + 512 * q;
        if (row < nd) {
          float sacc = 0.f;
#pragma unroll 8
          for (int n = 0; n < NK; n++) sacc += (float)*reinterpret_cast<const __bf16*>(id + n * sd + 2 * row);
          bsum[q] += sacc;
        }
      }
    }
    PROF_T(d5);
    __syncthreads();
    PROF_T(d6);
    PROF_ADD(50, d0, d1);   // registers → LDS (waits for the prefetched loads)
    PROF_ADD(51, d1, d2);   // barrier
    PROF_ADD(52, d2, d3);   // next chunk's loads issued
    PROF_ADD(53, d3, d4);   // transposing reads + MFMA
    PROF_ADD(54, d4, d5);   // bias sums
    PROF_ADD(55, d5, d6);   // barrier
    PROF_ADD(56, d6 - 1, d6);
  }
  float* slab = a.slab + (size_t)part * dm.slab_n;
  const float gsv = a.gs ? a.gs[0] : 1.0f;
#pragma unroll
  for (int m = 0; m < DW_NDW; m++) {
    const int t = wave + 8 * m;
    if (t < ntile) {
      const int otl = t / nit, itl = t - otl * nit;
      const int tl = (jb.o0 + otl) * IT + jb.i0 + itl;   // the layer's tile enumeration: output tile major
      f32x4* g4 = reinterpret_cast<f32x4*>(slab + ((size_t)(dm.tile_off[l] + tl) * 64 + lane) * 16);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 v;
        v[0] = acc[m][4 * q + 0]; v[1] = acc[m][4 * q + 1]; v[2] = acc[m][4 * q + 2]; v[3] = acc[m][4 * q + 3];
        g4[q] = a.gs ? v * gsv : v;
      }
      if (do_bias_row && jb.i0 + itl == IT - 1) {   // row `in` of the last input tile = the bias gradient of this tile's 32 outputs
        const int r = in & 31, reg = (r & 3) + 4 * (r >> 3), o = (jb.o0 + otl) * 32 + (lane & 31);
        float bv = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) bv = rr == reg ? acc[m][rr] : bv;
        if ((lane >> 5) == ((r >> 2) & 1) && o < out) slab[(size_t)dm.tile_off[dm.nL] * 1024 + dm.bias_lin[l] + o] = a.gs ? bv * gsv : bv;
      }
    }
  }
  if (do_bias_sum) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int row = tid + 512 * q;
      if (row < nd && rd0 + row < out) slab[(size_t)dm.tile_off[dm.nL] * 1024 + dm.bias_lin[l] + rd0 + row] = a.gs ? bsum[q] * gsv : bsum[q];
    }
  }
}
template <int DW_NDW>
__global__ void __launch_bounds__(512) k_chain_dw_b(ChainDims cd, BfDims bd, DwArgsB a) {
  chain_dw_b_body<DW_NDW>(cd, bd, a, blockIdx.x, gridDim.x, blockIdx.y);
}
template <int DW_NDW>
__global__ void __launch_bounds__(512) k_chain_dw_b_group(GroupTable<ChainBDims, DwArgsB> g) {
  const int j = group_find(g.start, g.n, blockIdx.x), r = blockIdx.x - g.start[j];
  chain_dw_b_body<DW_NDW>(g.dims[j].cd, g.dims[j].bd, g.args[j], r % g.gx[j], g.gx[j], r / g.gx[j]);
}

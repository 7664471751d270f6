// lde_mfma.h — pieces shared by the MLP right-hand-side kernels (lde_mlp.hip) and the dense-chain kernels (lde_chain.hip):
// the layer description, the one-time weight re-layout into MFMA fragment order, and the large-K weight-gradient
// kernel + slab reduction that consume staged (a_l, δ_l) panels.
#ifndef LDE_MFMA_H
#define LDE_MFMA_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "lde_device.h"

namespace lde {

constexpr int NB = 16;        // trajectories (columns) per workgroup
constexpr int MAXL = LDE_MAX_LAYERS;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef LDE_ABL
#define LDE_ABL 0
#endif
// diagnostic builds only (LDE_ABL != 0): 1 = no GEMM, 2 = operands loaded but no MFMA, 3 = no epilogue,
// 4 = no weight-gradient products (and no bias sums), 5 = no bias sums
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  if (LDE_ABL == 2) { c[0] += a + b; return c; }
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16 operand mode of the dense chains (BASELINE.json configs[4]: "mixed fp32 solve / bf16 encoder-decoder"): storage, biases,
// activations and accumulation stay f32; only the two MFMA operands are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) right before the multiply. A lane's float4 of a K-group — k = 16·kg + 4·(lane>>4) + 0..3 — is exactly the
// operand of ONE v_mfma_f32_16x16x16_bf16, which replaces the four v_mfma_f32_16x16x4_f32 of the f32 path.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef int i32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 cvt_bf16x4(f32x4 v) {
  const bf16x2_ lo = __builtin_convertvector((f32x2_){v[0], v[1]}, bf16x2_), hi = __builtin_convertvector((f32x2_){v[2], v[3]}, bf16x2_);
  const i32x2_ r = {__builtin_bit_cast(int, lo), __builtin_bit_cast(int, hi)};
  return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ f32x4 mfma16_bf(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// LDE_PROF builds (diagnostic): thread 0 of workgroup 0 accumulates s_memtime cycles per phase into g_prof[].
//   0 stage combination   1 control / error norm / save   2+2l GEMM of layer l (to its last MFMA/epilogue)   3+2l barrier after it
//   in the adjoint the backward layers follow at 2+2(nL+l'), and 30 = panel staging stores
#ifndef LDE_PROF
#define LDE_PROF 0
#endif
#if LDE_PROF
static __device__ long long g_prof[64];
#define PROF_T(var) const long long var = (long long)__builtin_readcyclecounter()
#define PROF_ADD(slot, t0, t1) do { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_fetch_add((unsigned long long*)&g_prof[slot], (unsigned long long)((t1) - (t0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)   /* no return value: nothing waits for it */
#else
#define PROF_T(var) do {} while (0)
#define PROF_ADD(slot, t0, t1) do {} while (0)
#endif
// -DLDE_PROF=2: also the stamps inside an evaluation (every stamp waits for the wave's outstanding LDS / scalar loads, so the phases of an
// evaluation are priced at the cost of stretching it; level 1 keeps to a handful of stamps per step attempt)
#if LDE_PROF >= 2
#define PROF_T2(var) PROF_T(var)
#define PROF_ADD2(slot, t0, t1) PROF_ADD(slot, t0, t1)
#else
#define PROF_T2(var) do {} while (0)
#define PROF_ADD2(slot, t0, t1) do {} while (0)
#endif

// Static description of the RHS handed to the kernels by value.
struct MlpDims {
  int nL;                 // Dense layers
  int sizes[MAXL + 1];    // [in, h1, ..., out]
  int act;                // hidden activation
  int D, Dp, P;           // state_dim, D + augment_dim, param_dim
  int DpA;                // Dp rounded up to 4: in the adjoint state λ rows start at DpA, g rows at 2·DpA
  int has_pend;           // PENDULUM_PLUS_MLP
  int frag_off[MAXL];     // float offset of layer l's W fragments   (RT_l × KG_l × 256)
  int fragT_off[MAXL];    // float offset of layer l's Wᵀ fragments
  int frag_n[MAXL], fragT_n[MAXL];
  int w_off[MAXL];        // offset of vec(W_l) in the flat (destructure-order) weight vector
  int b_off[MAXL];        // offset of b_l
  int hmax;               // rows of the widest hidden panel (padded to 32)
  int ld_h;               // stride of the widest hidden panel (the two back-propagation panels use it)
  int ld_hl[MAXL];        // stride of hidden panel l (activation of layer l), sized for ITS width
  int h_off[MAXL];        // float offset of hidden panel l from the first one
  int h_total;            // floats of all hidden panels
  int ld_sf, ld_sb;       // stride of state panels in the forward / adjoint kernel
  int coupled;            // LDE_BATCH_COUPLED
  int solver;
  int nW;
  int bias_lin[MAXL];     // offset of layer l's bias (gradient) in the compact [Σ out] vector
  int nbias;
  int tile_off[MAXL + 1]; // first weight-gradient tile (32×32 of Wᵀ) of layer l in the global tile enumeration
  int slab_n;             // floats of one workgroup's slab: ntiles·1024 (tiles in accumulator-fragment order) + nbias
  int blk_off[MAXL];      // staged block of one evaluation: layer l's a-panel [16][in32] at blk_off[l], its δ-panel [16][out32] right after
  int blk_floats;         // floats of one staged block
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// smallest stride ≥ rows with stride ≡ 8 (mod 32): (stride/4) ≡ 2 (mod 8) makes the ds_read_b128 operand pattern
// "16 columns × 4 lane-groups of 16 B" conflict-free in every 16-lane service group.
__host__ __device__ inline int panel_stride(int rows) {
  int v = ((rows + 31) / 32) * 32 + 8;
  if (v - 32 >= rows) v -= 32;
  return v;
}

// ---- one-time weight re-layout: flat destructure order → MFMA fragment order (W and Wᵀ) ---------------
// fragment (rt, kg) of a matrix M[R×K]: lane l holds the float4 M[rt*16 + (l&15)][kg*16 + 4*(l>>4) + 0..3] (0 outside).
// (`keep`, optional: the flat vector is also copied there — the handle's own copy, read for the biases later — which
//  saves a separate device-to-device copy per lde_*_set_weights_device call: one launch instead of two per training step
//  and module.)
// bf16 copies (round 3, the chains' bf16 mode): fragment (rt, kg) of M[R×K] in the K = 32 layout of v_mfma_f32_16x16x32_bf16 — lane l
// holds the EIGHT consecutive-k values M[rt*16 + (l&15)][kg*32 + 8*(l>>4) + 0..7] as one 16-byte word (0 outside). Offsets of layer l
// in the two bf16 arrays (elements): bf_frag_off(dm, l, false / true).
__host__ __device__ inline size_t bf_frag_elems(const MlpDims& dm, int l, bool T) {
  const int R = T ? dm.sizes[l] : dm.sizes[l + 1], K = T ? dm.sizes[l + 1] : dm.sizes[l];
  return (size_t)((R + 15) / 16) * ((K + 31) / 32) * 512;
}
__host__ __device__ inline size_t bf_frag_off(const MlpDims& dm, int l, bool T) {
  size_t o = 0;
  for (int m = 0; m < l; m++) o += bf_frag_elems(dm, m, T);
  return o;
}
// what the re-layout of ONE layer needs, as plain numbers: k_refresh_many reads them from its job table in one load (through a device
// copy of MlpDims every workgroup walked a chain of dependent scalar loads before its first useful instruction)
struct FragGeom {                       // (no padding bytes: lde_refresh_weights compares its job table byte-wise with the previous call's)
  long long bf_off, bfT_off;
  int in, out, w_off, lo, hi;           // the layer's W at Wflat + w_off; its slice [lo, hi) of the flat vector (weights and, behind them, biases)
  int frag_off, frag_nf, fragT_off, fragT_nf;   // f32 fragments (256 floats each)
  int bf_nf, bfT_nf;                    // bf16 fragments (512 elements each)
  int pad_;
};
static_assert(sizeof(FragGeom) == 2 * 8 + 12 * 4, "FragGeom must not contain padding");
__host__ __device__ inline FragGeom frag_geom(const MlpDims& dm, int l) {
  FragGeom g;
  g.pad_ = 0;
  g.in = dm.sizes[l]; g.out = dm.sizes[l + 1]; g.w_off = dm.w_off[l];
  g.lo = dm.w_off[l]; g.hi = l + 1 < dm.nL ? dm.w_off[l + 1] : dm.nW;
  g.frag_off = dm.frag_off[l]; g.frag_nf = dm.frag_n[l] >> 8; g.fragT_off = dm.fragT_off[l]; g.fragT_nf = dm.fragT_n[l] >> 8;
  g.bf_nf = (int)(bf_frag_elems(dm, l, false) >> 9); g.bfT_nf = (int)(bf_frag_elems(dm, l, true) >> 9);
  g.bf_off = (long long)bf_frag_off(dm, l, false); g.bfT_off = (long long)bf_frag_off(dm, l, true);
  return g;
}

// dst[e] = src[e] for e = first, first + stride, … < hi — four loads in flight per thread (as a plain loop every iteration is a dependent
// global round trip: ten of them per thread for the encoder's 784 × 200 layer)
__device__ inline void copy_strided(float* __restrict__ dst, const float* __restrict__ src, int first, int hi, int stride) {
  int e = first;
  for (; e + 3 * stride < hi; e += 4 * stride) {
    const float v0 = src[e], v1 = src[e + stride], v2 = src[e + 2 * stride], v3 = src[e + 3 * stride];
    dst[e] = v0; dst[e + stride] = v1; dst[e + 2 * stride] = v2; dst[e + 3 * stride] = v3;
  }
  for (; e < hi; e += stride) dst[e] = src[e];
}

// One WAVE per fragment (launches use 256 threads per block): the fragment's tile coordinates are wave-uniform — one scalar division per
// 512 (256) elements instead of two vector divisions per element, which was what these launches spent their time on (k_refresh_many:
// 15 µs per training step) — and a lane writes its 16 bytes of the fragment with one store.
__device__ inline void build_frags_layer_bf(const float* __restrict__ Wflat, const FragGeom& g, __bf16* __restrict__ fragb,
                                            __bf16* __restrict__ fragTb, int bx, int nbx) {
  typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
  const int in = g.in, out = g.out;
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(bx * wpb + (int)(threadIdx.x >> 6)), nw = nbx * wpb;
  const int r16 = lane & 15, k8 = 8 * (lane >> 4);
  const float* W = Wflat + g.w_off;  // column-major [out×in]: W(o,i) at o + out*i
  {
    const int KG = (in + 31) / 32, nf = g.bf_nf;
    __bf16* dst = fragb + g.bf_off;
    for (int f = wave; f < nf; f += nw) {
      const int rt = f / KG, kg = f - rt * KG;
      const int o = rt * 16 + r16, i0 = kg * 32 + k8;
      bfx8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (__bf16)((o < out && i0 + j < in) ? W[o + (size_t)out * (i0 + j)] : 0.f);
      *reinterpret_cast<bfx8*>(dst + ((size_t)f << 9) + 8 * lane) = v;
    }
  }
  {
    const int KG = (out + 31) / 32, nf = g.bfT_nf;   // Wᵀ[in×out]
    __bf16* dst = fragTb + g.bfT_off;
    for (int f = wave; f < nf; f += nw) {
      const int rt = f / KG, kg = f - rt * KG;
      const int i = rt * 16 + r16, o0 = kg * 32 + k8;
      bfx8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (__bf16)((o0 + j < out && i < in) ? W[o0 + j + (size_t)out * i] : 0.f);
      *reinterpret_cast<bfx8*>(dst + ((size_t)f << 9) + 8 * lane) = v;
    }
  }
}

__device__ inline void build_frags_layer(const float* __restrict__ Wflat, const FragGeom& g, float* __restrict__ frag,
                                         float* __restrict__ fragT, float* __restrict__ keep, int bx, int nbx) {
  const int in = g.in, out = g.out;
  const int stride = nbx * blockDim.x, first = bx * blockDim.x + threadIdx.x;
  if (keep) copy_strided(keep, Wflat, g.lo + first, g.hi, stride);   // this layer's slice [w_off[l], w_off[l+1]) of the flat vector
  const float* W = Wflat + g.w_off;  // column-major [out×in]: W(o,i) at o + out*i
  if (!frag) return;                     // (lde_refresh_weights for a chain in bf16 mode: its f32 fragments are not read)
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(bx * wpb + (int)(threadIdx.x >> 6)), nw = nbx * wpb;
  const int r16 = lane & 15, k4 = 4 * (lane >> 4);
  {
    const int KG = cdiv(in, 16), nf = g.frag_nf;
    float* dst = frag + g.frag_off;
    for (int f = wave; f < nf; f += nw) {
      const int rt = f / KG, kg = f - rt * KG;
      const int o = rt * 16 + r16, i0 = kg * 16 + k4;
      f32x4 v;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) v[s4] = (o < out && i0 + s4 < in) ? W[o + (size_t)out * (i0 + s4)] : 0.f;
      *reinterpret_cast<f32x4*>(dst + ((size_t)f << 8) + 4 * lane) = v;
    }
  }
  {
    const int KG = cdiv(out, 16), nf = g.fragT_nf;  // Wᵀ[in×out]
    float* dst = fragT + g.fragT_off;
    for (int f = wave; f < nf; f += nw) {
      const int rt = f / KG, kg = f - rt * KG;
      const int i = rt * 16 + r16, o0 = kg * 16 + k4;
      f32x4 v;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) v[s4] = (o0 + s4 < out && i < in) ? W[o0 + s4 + (size_t)out * i] : 0.f;
      *reinterpret_cast<f32x4*>(dst + ((size_t)f << 8) + 4 * lane) = v;
    }
  }
}

static __global__ void k_build_frags(const float* __restrict__ Wflat, MlpDims dm, float* __restrict__ frag,
                              float* __restrict__ fragT, float* __restrict__ keep, __bf16* __restrict__ fragb = nullptr,
                              __bf16* __restrict__ fragTb = nullptr) {
  const FragGeom g = frag_geom(dm, blockIdx.y);
  if (fragb) build_frags_layer_bf(Wflat, g, fragb, fragTb, blockIdx.x, gridDim.x);   // (reads Wflat: before `keep` could alias it)
  build_frags_layer(Wflat, g, frag, fragT, keep, blockIdx.x, gridDim.x);
}

// The same re-layout for MANY modules in one launch (lde_refresh_weights: once per optimiser step instead of once per
// module and step): blockIdx.y walks a device-resident job table — one job per (chain, layer), or a plain copy of a flat
// vector into a handle's own buffer (layer < 0: the recurrent stacks read their weights in destructure order).
struct RefreshJob {
  const float* src;
  float* keep;
  float* frag;
  float* fragT;
  FragGeom g;          // the layer's geometry (layer ≥ 0)
  int layer;
  int n;
  __bf16* fragb;       // bf16 K = 32 fragment copies (chains; nullptr: none)
  __bf16* fragTb;
};
static __global__ void k_refresh_many(const RefreshJob* __restrict__ jobs) {
  const RefreshJob j = jobs[blockIdx.y];
  if (j.layer < 0) {
    copy_strided(j.keep, j.src, blockIdx.x * blockDim.x + threadIdx.x, j.n, gridDim.x * blockDim.x);
    return;
  }
  if (j.fragb) build_frags_layer_bf(j.src, j.g, j.fragb, j.fragTb, blockIdx.x, gridDim.x);
  build_frags_layer(j.src, j.g, j.frag, j.fragT, j.keep, blockIdx.x, gridDim.x);
}

__host__ __device__ inline int pad32(int v) { return (v + 31) & ~31; }

// Offsets every kernel derives from the layer sizes: fragment arrays, flat (destructure-order) weight vector, compact
// bias vector, 32×32 weight-gradient tile enumeration, staged-block layout. Returns {W-fragment floats, Wᵀ-fragment floats}.
inline void fill_layer_offsets(MlpDims& dm, size_t* nfrag, size_t* nfragT) {
  int off = 0, offT = 0, woff = 0, blin = 0, toff = 0, boff = 0;
  for (int l = 0; l < dm.nL; l++) {
    const int in = dm.sizes[l], o = dm.sizes[l + 1];
    dm.frag_off[l] = off;
    dm.frag_n[l] = cdiv(o, 16) * cdiv(in, 16) * 256;
    off += dm.frag_n[l];
    dm.fragT_off[l] = offT;
    dm.fragT_n[l] = cdiv(in, 16) * cdiv(o, 16) * 256;
    offT += dm.fragT_n[l];
    dm.w_off[l] = woff;
    woff += o * in;
    dm.b_off[l] = woff;
    woff += o;
    dm.bias_lin[l] = blin;
    blin += (o + 3) & ~3;   // 16-byte aligned bias blocks (vector loads in the epilogue)
    dm.tile_off[l] = toff;
    toff += cdiv(o, 32) * cdiv(in, 32);
    dm.blk_off[l] = boff;
    boff += NB * (pad32(in) + pad32(o));
  }
  dm.nW = woff;
  dm.tile_off[dm.nL] = toff;
  dm.nbias = blin;
  dm.slab_n = toff * 1024 + ((blin + 3) & ~3);
  dm.blk_floats = boff;
  *nfrag = off;
  *nfragT = offT;
}

static constexpr size_t LDS_MAX = 160 * 1024;

// ---- the weight gradient as one large-K product over everything the solve staged ---------------------------------------
// grid (solve tile, K-split part, job). A job is a block of 32×32 tiles of ONE layer's gWᵀ[in×out]: a range of output
// tiles [o0,o1) × a range of input tiles [i0,i1), at most 8·NDW tiles (NDW per wave). The workgroup walks the tile's
// staged slots part, part+KS, …: copies the rows of that layer's a-panel and (weight-scaled) δ-panel it needs into LDS
// and every wave adds 8 MFMAs (v_mfma_f32_32x32x2_f32, K = the 16 columns of the slot) to each of its tiles.
// ≈ 110 VGPRs and a few tens of KB of LDS ⇒ two workgroups per CU: one loads while the other multiplies.
// Result: the job's part of the (tile, part) slab, tiles in accumulator-fragment order (+ bias sums).
struct DwArgs {
  const float* stage;
  const float* wts;
  const int32_t* nslots;   // staged slots per tile, or nullptr: tiles are filled in order, `total` slots in all (cap per tile)
  float* slab;          // [nWG·KS][slab_n]
  int cap;
  long long total;
};

// accumulator tiles per wave: 4 (32 tiles per job) in general; 1 or 2 when every layer is that small — fewer registers,
// more workgroups per CU, which is what a kernel that mostly streams staged panels needs (config 3: 8 tiles in all)
inline int dw_pick_ndw(const MlpDims& dm) {
  int mx = 0;
  for (int l = 0; l < dm.nL; l++) mx = std::max(mx, cdiv(dm.sizes[l], 32) * cdiv(dm.sizes[l + 1], 32));
  return mx <= 8 ? 1 : (mx <= 16 ? 2 : 4);
}

struct DwJob { int l, o0, o1, i0, i1; };

// A layer's IT × OT tile grid is cut into jobs of ≤ cap tiles along its LONGER side, each job spanning the whole shorter
// side: a job loads (rows of a) + (rows of δ) per slot, and that sum is smallest this way (the encoder's 784 → 200
// layer: 25 × 7 tiles; 7 jobs of one output tile each would load the 800-row a-panel seven times — 53 KB per job and
// slot — where jobs of 4 input tiles × 7 output tiles load 22 KB).
__host__ __device__ inline int dw_layer_jobs(int IT, int OT, int cap) {
  if (IT <= OT || OT > cap) return IT <= cap ? cdiv(OT, cap / IT) : OT * cdiv(IT, cap);
  return cdiv(IT, cap / OT);
}
__host__ __device__ inline int dw_jobs(const MlpDims& dm, int ndw) {
  int n = 0;
  for (int l = 0; l < dm.nL; l++) n += dw_layer_jobs(cdiv(dm.sizes[l], 32), cdiv(dm.sizes[l + 1], 32), 8 * ndw);
  return n;
}
__host__ __device__ inline DwJob dw_decode(const MlpDims& dm, int z, int DW_CAP) {
  DwJob j{0, 0, 0, 0, 0};
  for (int l = 0; l < dm.nL; l++) {
    const int IT = cdiv(dm.sizes[l], 32), OT = cdiv(dm.sizes[l + 1], 32), nj = dw_layer_jobs(IT, OT, DW_CAP);
    if (z < nj) {
      j.l = l;
      if (IT > OT && OT <= DW_CAP) {          // cut along the input tiles
        const int ci = DW_CAP / OT;
        j.o0 = 0;
        j.o1 = OT;
        j.i0 = z * ci;
        j.i1 = min(IT, j.i0 + ci);
      } else if (IT <= DW_CAP) {              // cut along the output tiles
        const int ro = DW_CAP / IT;
        j.o0 = z * ro;
        j.o1 = min(OT, j.o0 + ro);
        j.i0 = 0;
        j.i1 = IT;
      } else {                                // both sides longer than a job: one output tile × chunks of input tiles
        const int nic = cdiv(IT, DW_CAP);
        j.o0 = z / nic;
        j.o1 = j.o0 + 1;
        j.i0 = (z % nic) * DW_CAP;
        j.i1 = min(IT, j.i0 + DW_CAP);
      }
      return j;
    }
    z -= nj;
  }
  return j;
}
// LDS floats a job of this layer set needs at most
inline size_t dw_lds_floats(const MlpDims& dm, int ndw) {
  size_t mx = 0;
  for (int z = 0, n = dw_jobs(dm, ndw); z < n; z++) {
    const DwJob j = dw_decode(dm, z, 8 * ndw);
    mx = std::max(mx, (size_t)NB * (((32 * (j.i1 - j.i0)) | 32) + ((32 * (j.o1 - j.o0)) | 32)));
  }
  return mx;
}

// BF: bf16 operands (chains in bf16 mode): the lane's eight (a, δ) pairs of a slot — columns 2·s8 + half — feed two
// v_mfma_f32_32x32x8_bf16 (s8 = 0..3 and 4..7: the same column set on both operands, so the contraction is the same sum).
template <int DW_NDW, bool BF>
static __device__ __forceinline__ void mlp_dw_body(const MlpDims& dm, const DwArgs& a, const int tile, const int part, const int KS, const int jobz) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const DwJob jb = dw_decode(dm, jobz, 8 * DW_NDW);
  const int l = jb.l, in = dm.sizes[l], out = dm.sizes[l + 1], in32 = pad32(in), out32 = pad32(out);
  const int IT = in32 / 32, nit = jb.i1 - jb.i0, ntile = (jb.o1 - jb.o0) * nit;
  const int na = 32 * nit, nd = 32 * (jb.o1 - jb.o0), ra0 = 32 * jb.i0, rd0 = 32 * jb.o0;
  // LDS strides: an odd multiple of 32 floats puts the two half-waves of a ds_read_b32 on disjoint bank halves
  const int lsa = na | 32, lsd = nd | 32;
  float* pa = dsm;
  float* pd = pa + NB * lsa;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[DW_NDW];
  int aoff[DW_NDW], doff[DW_NDW];
#pragma unroll
  for (int m = 0; m < DW_NDW; m++) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
    const int t = wave + 8 * m;
    const int otl = t / nit, itl = t - otl * nit;
    aoff[m] = t < ntile ? half * lsa + itl * 32 + l31 : -1;
    doff[m] = half * lsd + otl * 32 + l31;
  }
  const bool do_bias = jb.i0 == 0;   // the jobs that hold the first input tile of their output rows also sum the bias gradient
  float bsum[2] = {0.f, 0.f};
  int ns;
  if (a.nslots) ns = a.nslots[tile];
  else {
    const long long left = a.total - (long long)tile * a.cap;
    ns = left <= 0 ? 0 : (left < a.cap ? (int)left : a.cap);
  }
  const int col = tid >> 5;
  // Slot e+KS's rows are fetched into registers while slot e is multiplied (up to PQ quads of each panel per lane;
  // wider jobs fetch the rest at store time).
  constexpr int PQ = 2;
  f32x4 qa[PQ], qd[PQ];
  float qw = 0.f;
  auto fetch = [&](int e) {
    const float* blk = a.stage + ((size_t)tile * a.cap + e) * dm.blk_floats + dm.blk_off[l];
    qw = a.wts[((size_t)tile * a.cap + e) * NB + col];
#pragma unroll
    for (int q = 0; q < PQ; q++) {
      const int r4 = l31 + 32 * q;
      if (4 * r4 < na) qa[q] = *reinterpret_cast<const f32x4*>(blk + col * in32 + ra0 + 4 * r4);
      if (4 * r4 < nd) qd[q] = *reinterpret_cast<const f32x4*>(blk + NB * in32 + col * out32 + rd0 + 4 * r4);
    }
  };
  auto scale = [&](f32x4 dv, float w) {
    // a column that carries no weight may hold anything (a diverged trajectory's NaN): 0·NaN must not reach the sum
#pragma unroll
    for (int q = 0; q < 4; q++) dv[q] = w != 0.f ? dv[q] * w : 0.f;
    return dv;
  };
  PROF_T(dw0);
  if (part < ns) fetch(part);
  for (int e = part; e < ns; e += KS) {
    const float* blk = a.stage + ((size_t)tile * a.cap + e) * dm.blk_floats + dm.blk_off[l];
    const float w = qw;
#pragma unroll
    for (int q = 0; q < PQ; q++) {
      const int r4 = l31 + 32 * q;
      if (4 * r4 < na) *reinterpret_cast<f32x4*>(pa + col * lsa + 4 * r4) = qa[q];
      if (4 * r4 < nd) *reinterpret_cast<f32x4*>(pd + col * lsd + 4 * r4) = scale(qd[q], w);
    }
    for (int r4 = l31 + 32 * PQ; 4 * r4 < na; r4 += 32)
      *reinterpret_cast<f32x4*>(pa + col * lsa + 4 * r4) = *reinterpret_cast<const f32x4*>(blk + col * in32 + ra0 + 4 * r4);
    for (int r4 = l31 + 32 * PQ; 4 * r4 < nd; r4 += 32)
      *reinterpret_cast<f32x4*>(pd + col * lsd + 4 * r4) =
          scale(*reinterpret_cast<const f32x4*>(blk + NB * in32 + col * out32 + rd0 + 4 * r4), w);
    __syncthreads();
    if (e + KS < ns) fetch(e + KS);
#pragma unroll
    for (int m = 0; m < DW_NDW; m++) {
      if (aoff[m] >= 0) {   // (a branch-free variant — absent tiles multiplied and dropped — was measured: no faster, and slower on small layers)
        const float* ap = pa + aoff[m];
        const float* bp = pd + doff[m];
        float av[8], bv[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
          av[s8] = ap[s8 * 2 * lsa];
          bv[s8] = bp[s8 * 2 * lsd];
        }
        if (BF) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(cvt_bf16x4(f32x4{av[0], av[1], av[2], av[3]}), cvt_bf16x4(f32x4{bv[0], bv[1], bv[2], bv[3]}), acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(cvt_bf16x4(f32x4{av[4], av[5], av[6], av[7]}), cvt_bf16x4(f32x4{bv[4], bv[5], bv[6], bv[7]}), acc[m], 0, 0, 0);
        } else {
#pragma unroll
          for (int s8 = 0; s8 < 8; s8++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s8], bv[s8], acc[m], 0, 0, 0);
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int row = tid + 512 * q;
        if (row < nd) {
          float sacc = 0.f;
#pragma unroll
          for (int n = 0; n < NB; n++) sacc += pd[n * lsd + row];
          bsum[q] += sacc;
        }
      }
    }
    __syncthreads();
  }
  PROF_T(dw1);
  PROF_ADD(50, dw0, dw1);
  PROF_ADD(51, dw1 - (long long)ns, dw1);   // slots
  PROF_ADD(52, dw1 - 1, dw1);               // workgroups counted
  float* slab = a.slab + ((size_t)tile * KS + part) * dm.slab_n;
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
        g4[q] = v;
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int row = tid + 512 * q;
      if (row < nd && rd0 + row < out) slab[(size_t)dm.tile_off[dm.nL] * 1024 + dm.bias_lin[l] + rd0 + row] = bsum[q];
    }
  }
}
template <int DW_NDW, bool BF = false>
static __global__ void __launch_bounds__(512) k_mlp_dw(MlpDims dm, DwArgs a) {
  mlp_dw_body<DW_NDW, BF>(dm, a, blockIdx.x, blockIdx.y, gridDim.y, blockIdx.z);
}

// ---- grouped launches (lde_chain_group_*): several small independent modules in ONE launch of each kernel ------------------------------
// A step of the GOKU model runs six chains on B columns only (latent_in's four heads, latent_out's two): 6 forward and 18 pullback
// launches of ≈ 5–7 µs each, one behind the other — a quarter of the captured training step. Every kernel of that family exists as a
// body taking its block coordinates as arguments; the grouped form carries up to GROUP_MAX argument sets in the kernel arguments and a
// workgroup finds its module from the prefix sums of the modules' grids. Same code on the same data per module: bit-equal results.
constexpr int GROUP_MAX = 4;      // chains per group
constexpr int GROUP_MAX_DW = 6;   // weight-gradient jobs per group (a recurrent stack brings one per cell: three stacks of two)
template <class Dims, class Args, int CAP = GROUP_MAX>
struct GroupTable {
  int n;
  int start[CAP + 1];   // first flat block of module j (start[n] = the grid)
  int gx[CAP], gy[CAP]; // the module's own grid extents (x, y) where a kernel decodes more than one coordinate
  Dims dims[CAP];
  Args args[CAP];
};
__device__ __forceinline__ int group_find(const int* start, int n, int b) {
  int j = 0;
  while (j + 1 < n && b >= start[j + 1]) j++;
  return j;
}
template <int DW_NDW, bool BF>
static __global__ void __launch_bounds__(512) k_mlp_dw_group(GroupTable<MlpDims, DwArgs, GROUP_MAX_DW> g) {
  const int j = group_find(g.start, g.n, blockIdx.x), r = blockIdx.x - g.start[j];
  const int per = g.gx[j] * g.gy[j];
  mlp_dw_body<DW_NDW, BF>(g.dims[j], g.args[j], r % g.gx[j], (r / g.gx[j]) % g.gy[j], g.gy[j], r / per);
}

// dW[flat] (+)= Σ_w slab[w][fragment position of flat] (+ the private slabs of workgroups that overflowed their staging area):
// ONE launch after k_mlp_dw (round 1 summed the slabs in one launch and gathered in a second). A workgroup owns one 32×32
// tile of the slab (or 1024 floats of its bias tail): every lane sums ITS 16 bytes over the partial slabs, in slab order —
// a fixed order, so the result is bit-reproducible — with sixteen fully coalesced loads in flight, and only then maps its
// four accumulator-fragment positions back to flat (destructure-order) weight indices: the scatter happens once, on
// nW floats, not once per slab.
static __device__ __forceinline__ void reduce_tiles_body(const float* __restrict__ priv, const int32_t* __restrict__ nflush, int nwg,
                                                         const float* __restrict__ slab, int nslab, const MlpDims& dm, float* __restrict__ dW,
                                                         int32_t* __restrict__ feedback, int assign, const int bx) {
  const int tid = threadIdx.x;
  if (bx == 0 && tid == 0) {   // tell the host (asynchronously) whether any workgroup ran out of staging slots
    int mx = 0;
    for (int w = 0; w < nwg; w++) mx = max(mx, nflush[w]);
    feedback[0] = max(mx, feedback[1]);   // feedback[1]: set by a kernel that ran out of staging slots without a private slab
  }
  const size_t base = (size_t)bx * 1024 + 4 * tid;
  if (base >= (size_t)dm.slab_n) return;
  const float* p = slab + base;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 16 <= nslab; w += 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(w + u) * dm.slab_n);
#pragma unroll
    for (int u = 0; u < 16; u++) s += v[u];
  }
  for (; w + 4 <= nslab; w += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(w + u) * dm.slab_n);
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u];
  }
  for (; w < nslab; w++) s += *reinterpret_cast<const f32x4*>(p + (size_t)w * dm.slab_n);
  if (priv)
    for (int g = 0; g < nwg; g++)
      if (nflush[g]) s += *reinterpret_cast<const f32x4*>(priv + (size_t)g * dm.slab_n + base);
  const int ntl = dm.tile_off[dm.nL];
  if (bx < ntl) {
    const int t = bx;
    int l = 0;
    while (l + 1 < dm.nL && t >= dm.tile_off[l + 1]) l++;
    const int in = dm.sizes[l], out = dm.sizes[l + 1], nit = cdiv(in, 32);
    const int otl = (t - dm.tile_off[l]) / nit, itl = (t - dm.tile_off[l]) - otl * nit;
    const int lane64 = tid >> 2, col = lane64 & 31, h = lane64 >> 5;   // C/D layout of v_mfma_f32_32x32x2_f32, tile holds Wᵀ
    const int o = otl * 32 + col;
    if (o < out) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * (tid & 3) + q, row = (r & 3) | (h << 2) | ((r >> 2) << 3);
        const int i = itl * 32 + row;
        if (i < in) {
          float* d = dW + dm.w_off[l] + (size_t)i * out + o;   // vec(W) column-major [out×in]
          *d = assign ? s[q] : *d + s[q];
        }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int bp = (int)(base - (size_t)ntl * 1024) + q;   // position in the compact bias vector (blocks padded to 4)
      if (bp >= dm.nbias) break;
      int l = 0;
      while (l + 1 < dm.nL && bp >= dm.bias_lin[l + 1]) l++;
      const int j = bp - dm.bias_lin[l];
      if (j < dm.sizes[l + 1]) {
        float* d = dW + dm.b_off[l] + j;
        *d = assign ? s[q] : *d + s[q];
      }
    }
  }
}
static __global__ void k_reduce_tiles(const float* __restrict__ priv, const int32_t* __restrict__ nflush, int nwg,
                                      const float* __restrict__ slab, int nslab, MlpDims dm, float* __restrict__ dW,
                                      int32_t* __restrict__ feedback, int assign) {
  reduce_tiles_body(priv, nflush, nwg, slab, nslab, dm, dW, feedback, assign, (int)blockIdx.x);
}
struct ReduceArgs {
  const float* priv;
  const int32_t* nflush;
  int nwg;
  const float* slab;
  int nslab;
  float* dW;
  int32_t* feedback;
  int assign;
};
static __global__ void k_reduce_tiles_group(GroupTable<MlpDims, ReduceArgs, GROUP_MAX_DW> g) {
  const int j = group_find(g.start, g.n, blockIdx.x);
  const ReduceArgs& a = g.args[j];
  reduce_tiles_body(a.priv, a.nflush, a.nwg, a.slab, a.nslab, g.dims[j], a.dW, a.feedback, a.assign, (int)blockIdx.x - g.start[j]);
}

// ---- the weight-gradient stream (lde_set_dw_stream, include/lde.h): defined in lde_api.hip, shared by the chain and recurrent
// pullbacks. A pullback's dW kernels depend on its staged panels only, and nothing but the optimiser depends on them: on a
// stream of their own they leave the critical path of the backward pass (dx of this module → the next module's pullback).
hipStream_t dw_stream_get();
// After the pullback kernel was enqueued on `stream`: returns the stream its weight-gradient kernels go to — the dw stream,
// made to wait for everything enqueued on `stream` so far, or `stream` itself when none is set. `prev_done`: this handle's
// event after its previous weight-gradient launch (the workspace is per handle); the caller waited on it already.
struct DwSync {
  hipEvent_t staged = nullptr;   // the pullback kernel's panels are complete (recorded on the caller's stream)
  hipEvent_t done = nullptr;     // this handle's weight-gradient kernels are complete (recorded on the dw stream)
  bool pending = false;
  bool done_captured = false;    // `done` was recorded while the dw stream was part of a stream capture (a graph node, not a live event)
};
inline bool dw_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
inline void dw_sync_destroy(DwSync& s) {
  if (s.staged) (void)hipEventDestroy(s.staged);
  if (s.done) (void)hipEventDestroy(s.done);
  s.staged = s.done = nullptr;
}
// before the pullback kernel overwrites the handle's workspace: wait for the handle's previous weight-gradient kernels
#if LDE_DW_DEBUG
inline void dw_dbg(const char* what, const DwSync* s, hipStream_t a, hipStream_t b) {
  hipStreamCaptureStatus ca = hipStreamCaptureStatusNone, cb = hipStreamCaptureStatusNone;
  if (a) (void)hipStreamIsCapturing(a, &ca);
  if (b) (void)hipStreamIsCapturing(b, &cb);
  fprintf(stderr, "[dw %s] sync=%p stream=%p(cap %d) dws=%p(cap %d) pending=%d\n", what, (const void*)s, (void*)a, (int)ca, (void*)b, (int)cb, s ? (int)s->pending : -1);
}
#else
#define dw_dbg(...) do {} while (0)
#endif
inline bool dw_sync_begin(DwSync& s, hipStream_t stream) {
  dw_dbg("begin", &s, stream, dw_stream_get());
  if (!s.pending) return true;
  s.pending = false;
  // The wait is an edge between two pieces of work of the same kind: live stream ↔ live event, or both inside ONE capture. Across the
  // boundary there is nothing to wait for — a capture starts behind a device synchronisation (the eager weight-gradient kernels are long
  // done; waiting on their event from a capturing stream while the dw stream itself is being captured is refused by the runtime), and a
  // captured graph joins the dw stream back into its origin before it ends (lde_join_dw), so whatever follows a replay is ordered behind it.
  if (dw_is_capturing(stream) != s.done_captured) return true;
  return hipStreamWaitEvent(stream, s.done, 0) == hipSuccess;
}
inline hipStream_t dw_sync_switch(DwSync& s, hipStream_t stream, bool* ok) {
  *ok = true;
  hipStream_t dws = dw_stream_get();
  dw_dbg("switch", &s, stream, dws);
  if (!dws || dws == stream) return stream;
  if (!s.staged && (hipEventCreateWithFlags(&s.staged, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)) {
    *ok = false;
    return stream;
  }
  if (hipEventRecord(s.staged, stream) != hipSuccess || hipStreamWaitEvent(dws, s.staged, 0) != hipSuccess) {
    *ok = false;
    return stream;
  }
  return dws;
}
inline bool dw_sync_ensure(DwSync& s) {   // the two events exist (dw_sync_switch creates them for the handle it is called with)
  if (s.staged) return true;
  return hipEventCreateWithFlags(&s.staged, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
}
inline bool dw_sync_end(DwSync& s, hipStream_t used, hipStream_t stream) {
  dw_dbg("end", &s, stream, used);
  if (used == stream) return true;
  s.pending = true;
  s.done_captured = dw_is_capturing(used);
  return hipEventRecord(s.done, used) == hipSuccess;
}

template <class T>
static bool grow(T** ptr, size_t* cap, size_t need) {
  if (need <= *cap) return true;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr;
  *cap = 0;
  if (hipMalloc(ptr, need * sizeof(T)) != hipSuccess) return false;
  *cap = need;
  return true;
}

// dW += Σ over the staged slots: k_mlp_dw over (ntile × ks × jobs), then the slab reduction (k_reduce_tiles).
//   slabs: [ntile·ks][slab_n] partial slabs
static int launch_weight_gradient(const MlpDims& dm, const DwArgs& da, int ntile, int ks, const float* priv,
                                  const int32_t* nflush, int npriv, float* dW, int32_t* feedback, hipStream_t stream,
                                  std::string& err, bool assign = false, bool bf16 = false) {
  const int ndw = dw_pick_ndw(dm);
  const size_t dlds = dw_lds_floats(dm, ndw) * sizeof(float);
  if (dlds > LDS_MAX) {
    err = "layer too wide for the weight-gradient kernel's LDS panels";
    return LDE_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_mlp_dw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_dw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_dw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_dw<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_dw<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_dw<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) {
      err = "hipFuncSetAttribute(k_mlp_dw) failed";
      return LDE_ERR_HIP;
    }
    attr_set = true;
  }
  const dim3 grid(ntile, ks, dw_jobs(dm, ndw));
  if (bf16) {
    if (ndw == 1) hipLaunchKernelGGL((k_mlp_dw<1, true>), grid, dim3(512), dlds, stream, dm, da);
    else if (ndw == 2) hipLaunchKernelGGL((k_mlp_dw<2, true>), grid, dim3(512), dlds, stream, dm, da);
    else hipLaunchKernelGGL((k_mlp_dw<4, true>), grid, dim3(512), dlds, stream, dm, da);
  } else if (ndw == 1) hipLaunchKernelGGL(k_mlp_dw<1>, grid, dim3(512), dlds, stream, dm, da);
  else if (ndw == 2) hipLaunchKernelGGL(k_mlp_dw<2>, grid, dim3(512), dlds, stream, dm, da);
  else hipLaunchKernelGGL(k_mlp_dw<4>, grid, dim3(512), dlds, stream, dm, da);
  hipLaunchKernelGGL(k_reduce_tiles, dim3(cdiv(dm.slab_n, 1024)), dim3(256), 0, stream, priv, nflush, npriv, da.slab, ntile * ks, dm,
                     dW, feedback, assign ? 1 : 0);
  if (hipGetLastError() != hipSuccess) {
    err = "weight-gradient kernels failed to launch";
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

}  // namespace lde
#endif  // LDE_MFMA_H

// lde_loss.hip — the variational sample and the loss terms of the training step (scope row f-3).
//
// Replaces what the example script runs as broadcast expressions between the encoder and the decoder and around the model's
// output:  sample(μ, logσ²)  [REF src/models/GOKU.jl:155-163],  vector_kl  [REF src/utils/utils.jl:15-49],  and
// reconstruction_loss = sum(mean((x − x̂)², dims=(2,3)))  [REF examples/pendulum_friction-less/model_train.jl:225-238].
//
// Design (gfx950). Pure HBM-bound streaming: one pass over the operands per call, 16-byte accesses, no LDS staging of data.
// The two reductions are bit-reproducible: a fixed grid of workgroups (a function of n only) each sums a contiguous,
// 16-byte-aligned slice in a fixed order (per-lane strided running sums → wave DPP tree → one LDS exchange per workgroup),
// writes its partial to `scratch`, and a one-wave kernel adds the partials by index. No atomics anywhere.
// Algorithmic bytes: sample 16 B/element (3 reads + 1 write), its pullback 16 B, kl 8 B, its pullback 16 B, mse 8 B, its
// pullback 12 B.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lde.h"

namespace lde {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int LOSS_WG = 256;                    // threads per workgroup
constexpr int LOSS_CHUNK = 4 * LOSS_WG * 8;     // floats per workgroup pass: 8 × 16-byte loads per lane

static inline int loss_grid(int64_t n) {   // a function of n only ⇒ the summation order is too
  const int64_t g = (n + LOSS_CHUNK - 1) / LOSS_CHUNK;
  return (int)(g < 1 ? 1 : (g > LDE_LOSS_SCRATCH_FLOATS ? LDE_LOSS_SCRATCH_FLOATS : g));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// workgroup sum of one value per thread, fixed order; result valid in thread 0
__device__ __forceinline__ float wg_sum(float v) {
  __shared__ float part[LOSS_WG / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < LOSS_WG / 64; w++) s += part[w];
  return s;
}

__device__ __forceinline__ float kl_term(float m, float lv) { return 0.5f * (__expf(lv) + m * m - lv - 1.0f); }

// TERM 0: kl(a = μ, b = logσ²); TERM 1: (a − b)²; TERM 2: kl(a, b) and, in the same pass, the sample l = a + c·exp(b/2) → l
template <int TERM>
__device__ __forceinline__ void loss_partial_body(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float* __restrict__ scratch, const float* __restrict__ c, float* __restrict__ l, float fscale,
                                                  float* fout, const float* fbase, const unsigned bx, const unsigned nblk) {   // (fout may BE fbase: k_sample_kl_pair's second part adds to the first's result)
  // workgroup w owns the slice [w·per, (w+1)·per) with per a multiple of 4 floats
  const int64_t nwg = nblk;
  int64_t per = (n + nwg - 1) / nwg;
  per = (per + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)bx * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const bool al = ((((uintptr_t)a) | ((uintptr_t)b) | (TERM == 2 ? ((uintptr_t)c) | ((uintptr_t)l) : 0)) & 15) == 0;
  int64_t i = lo + 4 * (int64_t)threadIdx.x;
  if (al) {
    for (; i + 4 <= hi; i += 4 * LOSS_WG) {
      const f4 va = *reinterpret_cast<const f4*>(a + i), vb = *reinterpret_cast<const f4*>(b + i);
      if (TERM == 0 || TERM == 2) {
        s0 += kl_term(va[0], vb[0]); s1 += kl_term(va[1], vb[1]); s2 += kl_term(va[2], vb[2]); s3 += kl_term(va[3], vb[3]);
        if (TERM == 2) {
          const f4 vc = *reinterpret_cast<const f4*>(c + i);
          f4 r;
#pragma unroll
          for (int q = 0; q < 4; q++) r[q] = va[q] + vc[q] * __expf(0.5f * vb[q]);   // = loss_map1<0>
          *reinterpret_cast<f4*>(l + i) = r;
        }
      } else {
        const f4 d = va - vb;
        s0 += d[0] * d[0]; s1 += d[1] * d[1]; s2 += d[2] * d[2]; s3 += d[3] * d[3];
      }
    }
  }
  for (; i < hi; i += 4 * LOSS_WG)   // unaligned operands, and the ragged end of the last slice
    for (int q = 0; q < 4 && i + q < hi; q++) {
      const float x = a[i + q], y = b[i + q];
      s0 += TERM != 1 ? kl_term(x, y) : (x - y) * (x - y);
      if (TERM == 2) l[i + q] = x + c[i + q] * __expf(0.5f * y);
    }
  const float s = wg_sum((s0 + s1) + (s2 + s3));
  if (threadIdx.x == 0) {
    scratch[bx] = s;
    if (fout) {   // a one-workgroup sum (n ≤ 8192: the KL terms of a training step) is its own last kernel: what k_loss_final computes from
                  // one partial, with the same separately rounded product and addition — one launch instead of two
#pragma clang fp contract(off)
      const float term = fscale * s;
      fout[0] = fbase ? fbase[0] + term : term;
    }
  }
}
template <int TERM>
__global__ void __launch_bounds__(LOSS_WG) k_loss_partial(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                          float* __restrict__ scratch, const float* __restrict__ c = nullptr,
                                                          float* __restrict__ l = nullptr, float fscale = 0.f, float* __restrict__ fout = nullptr,
                                                          const float* __restrict__ fbase = nullptr) {
  loss_partial_body<TERM>(a, b, n, scratch, c, l, fscale, fout, fbase, blockIdx.x, gridDim.x);
}

// out = scale·Σ partials (+ base[0]: a running total of loss terms — the elementwise additions of the loss expression folded in)
__global__ void __launch_bounds__(64) k_loss_final(const float* __restrict__ scratch, int nwg, float scale, float* __restrict__ out,
                                                   const float* __restrict__ base = nullptr) {
#pragma clang fp contract(off)
  float s = 0.f;
  for (int i = threadIdx.x; i < nwg; i += 64) s += scratch[i];
  s = wave_sum(s);
  // (no contraction into an fma: base + the separately rounded term, the bits of the separate entry point followed by an addition)
  if (threadIdx.x == 0) out[0] = base ? base[0] + scale * s : scale * s;
}

// OP 0: l = μ + ε·exp(lv/2)                           (a = μ, b = lv, c = ε → o0)
// OP 1: dlv = dl·ε·exp(lv/2)/2                        (a = lv, b = ε, c = dl → o0)
// OP 2: dμ = k·μ, dlv = k·(exp(lv) − 1)/2, k = g·scale (a = μ, b = lv → o0, o1)
// OP 3: dx̂ = 2k·(x̂ − x)                              (a = x, b = x̂ → o0)
// OP 4: OP 1 and OP 2 in one pass, summed with the cotangent of the sample: dμ = dl + k·μ,
//       dlv = dl·ε·exp(lv/2)/2 + k·(exp(lv) − 1)/2    (a = μ, b = lv, c = ε, d = dl → o0, o1)
template <int OP>
__device__ __forceinline__ void loss_map1(float a, float b, float c, float k, float& o0, float& o1) {
  if (OP == 0) o0 = a + c * __expf(0.5f * b);
  else if (OP == 1) o0 = 0.5f * c * b * __expf(0.5f * a);
  else if (OP == 2) { o0 = k * a; o1 = 0.5f * k * (__expf(b) - 1.0f); }
  else o0 = 2.0f * k * (b - a);
}

__device__ __forceinline__ void sample_kl_bwd_body(const float* __restrict__ mu, const float* __restrict__ lv,
                                                   const float* __restrict__ eps, const float* __restrict__ dl,
                                                   const float* __restrict__ g, float scale, int64_t n,
                                                   float* __restrict__ dmu, float* __restrict__ dlv, const unsigned bx, const unsigned nblk) {
  const float k = g[0] * scale;
  const bool al = ((((uintptr_t)mu) | ((uintptr_t)lv) | ((uintptr_t)eps) | ((uintptr_t)dl) | ((uintptr_t)dmu) | ((uintptr_t)dlv)) & 15) == 0;
  const int64_t stride = 4 * (int64_t)nblk * LOSS_WG;
  int64_t i = 4 * ((int64_t)bx * LOSS_WG + threadIdx.x);
  auto one = [&](float m, float v, float e, float d, float& om, float& ov) {
#pragma clang fp contract(off)   // the two parts rounded as the separate kernels round them, then plainly added
    float t0, t1, u0, u1 = 0.f;
    loss_map1<2>(m, v, 0.f, k, t0, t1);    // the KL term's part
    loss_map1<1>(v, e, d, k, u0, u1);      // the sample's part of dlv
    om = d + t0;              // plain additions of the two separately rounded cotangents (what autograd's accumulation does)
    ov = u0 + t1;
  };
  if (al) {
    for (; i + 4 <= n; i += stride) {
      const f4 vm = *reinterpret_cast<const f4*>(mu + i), vv = *reinterpret_cast<const f4*>(lv + i);
      const f4 ve = *reinterpret_cast<const f4*>(eps + i), vd = *reinterpret_cast<const f4*>(dl + i);
      f4 r0, r1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float t0, t1;
        one(vm[q], vv[q], ve[q], vd[q], t0, t1);
        r0[q] = t0;
        r1[q] = t1;
      }
      *reinterpret_cast<f4*>(dmu + i) = r0;
      *reinterpret_cast<f4*>(dlv + i) = r1;
    }
  }
  for (; i < n; i += stride)
    for (int q = 0; q < 4 && i + q < n; q++) one(mu[i + q], lv[i + q], eps[i + q], dl[i + q], dmu[i + q], dlv[i + q]);
}
__global__ void __launch_bounds__(LOSS_WG) k_sample_kl_bwd(const float* __restrict__ mu, const float* __restrict__ lv,
                                                           const float* __restrict__ eps, const float* __restrict__ dl,
                                                           const float* __restrict__ g, float scale, int64_t n,
                                                           float* __restrict__ dmu, float* __restrict__ dlv) {
  sample_kl_bwd_body(mu, lv, eps, dl, g, scale, n, dmu, dlv, blockIdx.x, gridDim.x);
}
// two parts (the GOKU tuple (z₀, θ) [REF src/models/GOKU.jl:155-163]) in one launch: workgroups [0, nblk_a) the first, the rest the second
struct SampleBwdPart { const float* mu; const float* lv; const float* eps; const float* dl; float scale; long long n; float* dmu; float* dlv; };
__global__ void __launch_bounds__(LOSS_WG) k_sample_kl_bwd_pair(SampleBwdPart pa, SampleBwdPart pb, const float* __restrict__ g, unsigned nblk_a) {
  if (blockIdx.x < nblk_a) sample_kl_bwd_body(pa.mu, pa.lv, pa.eps, pa.dl, g, pa.scale, pa.n, pa.dmu, pa.dlv, blockIdx.x, nblk_a);
  else sample_kl_bwd_body(pb.mu, pb.lv, pb.eps, pb.dl, g, pb.scale, pb.n, pb.dmu, pb.dlv, blockIdx.x - nblk_a, gridDim.x - nblk_a);
}

template <int OP>
__global__ void __launch_bounds__(LOSS_WG) k_loss_map(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, const float* __restrict__ g, float scale,
                                                      int64_t n, float* __restrict__ o0, float* __restrict__ o1) {
  const float k = g ? g[0] * scale : scale;
  const bool three = OP == 0 || OP == 1, two_out = OP == 2;
  const bool al = ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)(three ? c : a)) | ((uintptr_t)o0) | ((uintptr_t)(two_out ? o1 : o0))) & 15) == 0;
  const int64_t stride = 4 * (int64_t)gridDim.x * LOSS_WG;
  int64_t i = 4 * ((int64_t)blockIdx.x * LOSS_WG + threadIdx.x);
  if (al) {
    for (; i + 4 <= n; i += stride) {
      const f4 va = *reinterpret_cast<const f4*>(a + i), vb = *reinterpret_cast<const f4*>(b + i);
      f4 vc = {0.f, 0.f, 0.f, 0.f}, r0, r1;
      if (three) vc = *reinterpret_cast<const f4*>(c + i);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float t0, t1 = 0.f;
        loss_map1<OP>(va[q], vb[q], vc[q], k, t0, t1);
        r0[q] = t0;
        r1[q] = t1;
      }
      *reinterpret_cast<f4*>(o0 + i) = r0;
      if (two_out) *reinterpret_cast<f4*>(o1 + i) = r1;
    }
  }
  for (; i < n; i += stride)
    for (int q = 0; q < 4 && i + q < n; q++) {
      float r0, r1 = 0.f;
      loss_map1<OP>(a[i + q], b[i + q], three ? c[i + q] : 0.f, k, r0, r1);
      o0[i + q] = r0;
      if (two_out) o1[i + q] = r1;
    }
}

static inline int map_grid(int64_t n) {
  const int64_t g = (n + 4 * LOSS_WG - 1) / (4 * LOSS_WG);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

template <int TERM>
static int loss_reduce(const float* a, const float* b, int64_t n, float scale, float* out, float* scratch, void* stream_,
                       const float* base = nullptr, const float* c = nullptr, float* l = nullptr) {
  if (!out || !scratch || n < 0 || (n > 0 && (!a || !b))) return LDE_ERR_INVALID_ARG;
  if (TERM == 2 && n > 0 && (!c || !l)) return LDE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = n == 0 ? 0 : loss_grid(n);   // an empty sum is 0
  if (g == 1) hipLaunchKernelGGL(k_loss_partial<TERM>, dim3(1), dim3(LOSS_WG), 0, stream, a, b, n, scratch, c, l, scale, out, base);
  else {
    if (g) hipLaunchKernelGGL(k_loss_partial<TERM>, dim3(g), dim3(LOSS_WG), 0, stream, a, b, n, scratch, c, l, 0.f, (float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, stream, scratch, g, scale, out, base);
  }
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

template <int OP>
static int loss_map(const float* a, const float* b, const float* c, const float* g, float scale, int64_t n, float* o0, float* o1,
                    void* stream_) {
  if (n < 0) return LDE_ERR_INVALID_ARG;
  if (n == 0) return LDE_OK;
  hipLaunchKernelGGL(k_loss_map<OP>, dim3(map_grid(n)), dim3(LOSS_WG), 0, (hipStream_t)stream_, a, b, c, g, scale, n, o0, o1);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

// ε ~ N(0, 1) for the variational sample (round 3). Philox4x32-10 [Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
// 1, 2, 3", SC'11] — the counter-based generator torch and Julia's Random123 use — keyed by the caller's 64-bit seed; block i of four
// outputs has the counter (i, call, offset + *epoch): `epoch` is a DEVICE counter (the optimiser's step count), so a captured training
// step draws fresh noise at every replay without the generator bookkeeping a framework puts in front of a replay (three launches and
// a host-to-device copy per step for two torch.randn calls). The four words of a block make two Box–Muller pairs.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ void randn_block(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long offset,
                                            unsigned call, const long long* __restrict__ epoch, unsigned* __restrict__ raw, const long long blk) {
  if (4 * blk >= n) return;
  const unsigned long long off = offset + (epoch ? (unsigned long long)*epoch : 0ull);
  unsigned w[4];
  philox4x32_10((unsigned)blk, call, (unsigned)off, (unsigned)(off >> 32), (unsigned)seed, (unsigned)(seed >> 32), w);
  float z[4];
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {
    const float u1 = ((float)(w[2 * pr] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1): 24 bits, never 0
    const float u2 = ((float)(w[2 * pr + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    z[2 * pr] = rad * cs;
    z[2 * pr + 1] = rad * sn;
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (4 * blk + q < n) {
      out[4 * blk + q] = z[q];
      if (raw) raw[4 * blk + q] = w[q];
    }
}
__global__ void __launch_bounds__(256) k_randn(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long offset,
                                                unsigned call, const long long* __restrict__ epoch, unsigned* __restrict__ raw) {
  randn_block(out, n, seed, offset, call, epoch, raw, (long long)blockIdx.x * 256 + threadIdx.x);
}

// The variational sample of a GOKU step in ONE launch: ε of both latent parts, l̃ = μ + ε·exp(logσ²/2) and the running total
// base + scale_a·Σ kl_a + scale_b·Σ kl_b — what lde_randn, lde_sample_kl_forward, lde_randn, lde_sample_kl_forward compute one after the
// other (four launches of ≈ 5 µs on 16 × B numbers), by the same device code in the same order: the same bits. One workgroup; parts of at
// most LOSS_CHUNK entries (a part that size is one workgroup in the separate calls too).
struct SampleFwdPart { const float* mu; const float* lv; float* eps; float* l; float scale; long long n; unsigned long long offset; unsigned call; };
constexpr int PAIR_WG = 1024;   // ε (ten Philox rounds and a Box–Muller pair per four numbers) by sixteen waves; the sums by the first four, as in the separate kernels
__global__ void __launch_bounds__(PAIR_WG) k_sample_kl_pair(SampleFwdPart pa, SampleFwdPart pb, unsigned long long seed, const long long* __restrict__ epoch,
                                                            const float* __restrict__ base, float* __restrict__ out, float* __restrict__ scratch) {
  for (long long blk = threadIdx.x; 4 * blk < pa.n; blk += PAIR_WG) randn_block(pa.eps, pa.n, seed, pa.offset, pa.call, epoch, nullptr, blk);
  for (long long blk = threadIdx.x; 4 * blk < pb.n; blk += PAIR_WG) randn_block(pb.eps, pb.n, seed, pb.offset, pb.call, epoch, nullptr, blk);
  __syncthreads();   // (the workgroup's own global stores, read back below: drained and visible)
  const bool summing = threadIdx.x < LOSS_WG;   // wave-uniform; the other waves keep the barrier count of loss_partial_body (one, in wg_sum)
  if (summing) loss_partial_body<2>(pa.mu, pa.lv, pa.n, scratch, pa.eps, pa.l, pa.scale, out, base, 0u, 1u);
  else __syncthreads();
  __syncthreads();   // wg_sum's staging is reused; thread 0 wrote out[0] and reads it as the second part's base
  if (summing) loss_partial_body<2>(pb.mu, pb.lv, pb.n, scratch + 1, pb.eps, pb.l, pb.scale, out, out, 0u, 1u);
  else __syncthreads();
}

}  // namespace lde

using namespace lde;

int loss_finalize(const float* scratch, int g, float scale, const float* base, float* out, hipStream_t stream) {
  if (!scratch || !out || g < 1) return LDE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, stream, scratch, g, scale, out, base);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

extern "C" {

int lde_sample_forward(const float* mu, const float* logvar, const float* eps, int64_t n, float* l, void* stream) {
  if (n > 0 && (!mu || !logvar || !eps || !l)) return LDE_ERR_INVALID_ARG;
  return loss_map<0>(mu, logvar, eps, nullptr, 1.0f, n, l, nullptr, stream);
}

int lde_sample_backward(const float* logvar, const float* eps, const float* dl, int64_t n, float* dlogvar, void* stream) {
  if (n > 0 && (!logvar || !eps || !dl || !dlogvar)) return LDE_ERR_INVALID_ARG;
  return loss_map<1>(logvar, eps, dl, nullptr, 1.0f, n, dlogvar, nullptr, stream);
}

int lde_kl_forward(const float* mu, const float* logvar, int64_t n, float scale, float* out, float* scratch, void* stream) {
  return loss_reduce<0>(mu, logvar, n, scale, out, scratch, stream);
}

int lde_kl_backward(const float* mu, const float* logvar, int64_t n, float scale, const float* dout, float* dmu, float* dlogvar,
                    void* stream) {
  if (n > 0 && (!mu || !logvar || !dout || !dmu || !dlogvar)) return LDE_ERR_INVALID_ARG;
  return loss_map<2>(mu, logvar, nullptr, dout, scale, n, dmu, dlogvar, stream);
}

int lde_mse_forward(const float* x, const float* xhat, int64_t n, float scale, float* out, float* scratch, void* stream) {
  return loss_reduce<1>(x, xhat, n, scale, out, scratch, stream);
}

int lde_mse_forward_add(const float* x, const float* xhat, int64_t n, float scale, const float* base, float* out, float* scratch,
                        void* stream) {
  return loss_reduce<1>(x, xhat, n, scale, out, scratch, stream, base);
}

int lde_sample_kl_forward(const float* mu, const float* logvar, const float* eps, int64_t n, float scale, const float* base,
                          float* l, float* out, float* scratch, void* stream) {
  return loss_reduce<2>(mu, logvar, n, scale, out, scratch, stream, base, eps, l);
}

int lde_sample_kl_backward(const float* mu, const float* logvar, const float* eps, const float* dl, const float* dout, float scale,
                           int64_t n, float* dmu, float* dlogvar, void* stream) {
  if (n < 0 || (n > 0 && (!mu || !logvar || !eps || !dl || !dout || !dmu || !dlogvar))) return LDE_ERR_INVALID_ARG;
  if (n == 0) return LDE_OK;
  hipLaunchKernelGGL(k_sample_kl_bwd, dim3(map_grid(n)), dim3(LOSS_WG), 0, (hipStream_t)stream, mu, logvar, eps, dl, dout, scale, n,
                     dmu, dlogvar);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

int lde_mse_backward(const float* x, const float* xhat, int64_t n, float scale, const float* dout, float* dxhat, void* stream) {
  if (n > 0 && (!x || !xhat || !dout || !dxhat)) return LDE_ERR_INVALID_ARG;
  return loss_map<3>(x, xhat, nullptr, dout, scale, n, dxhat, nullptr, stream);
}

int lde_sample_kl_pair_forward(const float* mu_a, const float* logvar_a, int64_t n_a, float scale_a, const float* mu_b, const float* logvar_b,
                               int64_t n_b, float scale_b, const float* base, uint64_t seed, uint64_t offset_a, uint64_t offset_b, uint32_t call_a,
                               uint32_t call_b, const int64_t* epoch_dev, float* eps_a, float* eps_b, float* l_a, float* l_b, float* out,
                               float* scratch, void* stream) {
  if (n_a < 1 || n_b < 1 || n_a > LOSS_CHUNK || n_b > LOSS_CHUNK) return LDE_ERR_UNSUPPORTED;   // the caller makes the separate calls
  if (!mu_a || !logvar_a || !mu_b || !logvar_b || !eps_a || !eps_b || !l_a || !l_b || !out || !scratch) return LDE_ERR_INVALID_ARG;
  const SampleFwdPart pa{mu_a, logvar_a, eps_a, l_a, scale_a, (long long)n_a, (unsigned long long)offset_a, (unsigned)call_a};
  const SampleFwdPart pb{mu_b, logvar_b, eps_b, l_b, scale_b, (long long)n_b, (unsigned long long)offset_b, (unsigned)call_b};
  hipLaunchKernelGGL(k_sample_kl_pair, dim3(1), dim3(PAIR_WG), 0, (hipStream_t)stream, pa, pb, (unsigned long long)seed, (const long long*)epoch_dev, base,
                     out, scratch);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}
int lde_sample_kl_pair_backward(const float* mu_a, const float* logvar_a, const float* eps_a, const float* dl_a, int64_t n_a, float scale_a,
                                const float* mu_b, const float* logvar_b, const float* eps_b, const float* dl_b, int64_t n_b, float scale_b,
                                const float* dout, float* dmu_a, float* dlogvar_a, float* dmu_b, float* dlogvar_b, void* stream) {
  if (n_a < 1 || n_b < 1) return LDE_ERR_UNSUPPORTED;
  if (!mu_a || !logvar_a || !eps_a || !dl_a || !mu_b || !logvar_b || !eps_b || !dl_b || !dout || !dmu_a || !dlogvar_a || !dmu_b || !dlogvar_b)
    return LDE_ERR_INVALID_ARG;
  const SampleBwdPart pa{mu_a, logvar_a, eps_a, dl_a, scale_a, (long long)n_a, dmu_a, dlogvar_a};
  const SampleBwdPart pb{mu_b, logvar_b, eps_b, dl_b, scale_b, (long long)n_b, dmu_b, dlogvar_b};
  const unsigned ga = (unsigned)map_grid(n_a), gb = (unsigned)map_grid(n_b);   // each part's grid as in lde_sample_kl_backward
  hipLaunchKernelGGL(k_sample_kl_bwd_pair, dim3(ga + gb), dim3(LOSS_WG), 0, (hipStream_t)stream, pa, pb, dout, ga);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

int lde_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, uint32_t call, const int64_t* epoch_dev, uint32_t* raw_words, void* stream) {
  if (n < 0 || (n > 0 && !out)) return LDE_ERR_INVALID_ARG;
  if (n == 0) return LDE_OK;
  const long long blocks = (n + 3) / 4;
  if (blocks > 0xffffffffll) return LDE_ERR_INVALID_ARG;   // the block index is one 32-bit counter word
  hipLaunchKernelGGL(k_randn, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)n, (unsigned long long)seed,
                     (unsigned long long)offset, (unsigned)call, (const long long*)epoch_dev, raw_words);
  return hipGetLastError() == hipSuccess ? LDE_OK : LDE_ERR_HIP;
}

}  // extern "C"

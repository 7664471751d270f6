// lde_optim.hip — the parameter update of the training step (scope row f-3).
//
// Replaces `Flux.Optimise.update!(opt, ps, grads)` with `opt = ADAMW(η, (β₁, β₂), decay)`
// [REF examples/pendulum_friction-less/model_train.jl:138, :190-192]; in the pinned Flux 0.13.6 [REF Manifest.toml:452] that is
// `Optimiser(ADAM(η, β), WeightDecay(decay))`, applied array by array as broadcast expressions:
//     m ← β₁·m + (1 − β₁)·g;   v ← β₂·v + (1 − β₂)·g²;   Δ = m/(1 − β₁ᵗ) / (√(v/(1 − β₂ᵗ)) + ε) · η;   Δ ← Δ + decay·x;   x ← x − Δ
// (the decay is NOT scaled by η). Here: ONE launch for all parameter arrays of a model. The array table travels in the kernel
// arguments (the gradient pointers change every step, so a device-resident table would need an upload per step); a
// workgroup owns 2048 consecutive elements of one array; 16-byte accesses; 28 algorithmic bytes per element (read x, g, m,
// v; write x, m, v) — HBM-bound in principle, launch-bound at a model's size (1.3 MB of parameters).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lde.h"

namespace lde {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int OPT_WG = 256;
constexpr int OPT_PER_WG = 8 * OPT_WG;   // elements per workgroup: two 16-byte accesses per lane and array
constexpr int OPT_MAXT = 48;             // arrays per launch (kernel-argument table: 48 × 40 B)

struct OptTable {
  lde_adam_tensor t[OPT_MAXT];
  int blk0[OPT_MAXT + 1];   // first workgroup of array i
  int n;
};

struct OptCoef { float b1, b2, omb1, omb2, inv_bc1, inv_bc2, eps, lr, decay; };

__device__ __forceinline__ void adam1(float& x, float g, float& m, float& v, const OptCoef& c) {
  m = c.b1 * m + c.omb1 * g;
  v = c.b2 * v + c.omb2 * g * g;
  const float d = (m * c.inv_bc1) / (__builtin_sqrtf(v * c.inv_bc2) + c.eps) * c.lr + c.decay * x;
  x -= d;
}

// `step_dev` != nullptr: the step count t lives in device memory (a captured hipGraph replays the SAME kernel arguments every step, so
// the bias corrections 1/(1 − βᵗ) cannot travel in them): every thread derives them from t = *step_dev + 1, and the count is advanced
// by the update kernel itself (`bump`, the step's last launch) — by the last workgroup to have READ it: each workgroup counts itself
// in behind a barrier, i.e. after all of its waves hold t. (Until round 3 a one-thread kernel in front of the update did it: one more
// launch per training step.)
__device__ unsigned g_adam_seen = 0;
__global__ void k_adam_tick(int64_t* step_dev) { *step_dev += 1; }   // (a step over empty arrays only)
__global__ void __launch_bounds__(OPT_WG) k_adamw_flux(OptTable tab, OptCoef c, int64_t* __restrict__ step_dev, int bump) {
  if (step_dev) {
    const int64_t ti = *step_dev + 1;
    const double t = (double)ti;
    c.inv_bc1 = (float)(1.0 / (1.0 - pow((double)c.b1, t)));
    c.inv_bc2 = (float)(1.0 / (1.0 - pow((double)c.b2, t)));
    if (bump) {
      __syncthreads();
      if (threadIdx.x == 0 && atomicAdd(&g_adam_seen, 1u) == gridDim.x - 1) {
        *step_dev = ti;
        g_adam_seen = 0;
      }
    }
  }
  int i = 0;
  while (i + 1 < tab.n && (int)blockIdx.x >= tab.blk0[i + 1]) i++;
  const lde_adam_tensor t = tab.t[i];
  const int64_t lo = (int64_t)((int)blockIdx.x - tab.blk0[i]) * OPT_PER_WG;
  const int64_t hi = lo + OPT_PER_WG < t.n ? lo + OPT_PER_WG : t.n;
  const bool al = ((((uintptr_t)t.p) | ((uintptr_t)t.g) | ((uintptr_t)t.m) | ((uintptr_t)t.v)) & 15) == 0;
  int64_t e = lo + 4 * (int64_t)threadIdx.x;
  if (al) {
    for (; e + 4 <= hi; e += 4 * OPT_WG) {
      f4 x = *reinterpret_cast<f4*>(t.p + e), m = *reinterpret_cast<f4*>(t.m + e), v = *reinterpret_cast<f4*>(t.v + e);
      const f4 g = *reinterpret_cast<const f4*>(t.g + e);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float xq = x[q], mq = m[q], vq = v[q];
        adam1(xq, g[q], mq, vq, c);
        x[q] = xq; m[q] = mq; v[q] = vq;
      }
      *reinterpret_cast<f4*>(t.p + e) = x;
      *reinterpret_cast<f4*>(t.m + e) = m;
      *reinterpret_cast<f4*>(t.v + e) = v;
    }
  }
  for (; e < hi; e += 4 * OPT_WG)   // unaligned arrays, and the ragged end
    for (int q = 0; q < 4 && e + q < hi; q++) adam1(t.p[e + q], t.g[e + q], t.m[e + q], t.v[e + q], c);
}

}  // namespace lde

using namespace lde;

static int adamw_impl(int n, const lde_adam_tensor* t, float lr, float beta1, float beta2, float eps, float decay, int64_t step,
                      int64_t* step_dev, void* stream) {
  if (n < 0 || (n > 0 && !t) || (!step_dev && step < 1) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return LDE_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++)
    if (t[i].n < 0 || (t[i].n > 0 && (!t[i].p || !t[i].g || !t[i].m || !t[i].v))) return LDE_ERR_INVALID_ARG;
  OptCoef c;
  c.b1 = beta1; c.b2 = beta2; c.omb1 = 1.0f - beta1; c.omb2 = 1.0f - beta2; c.eps = eps; c.lr = lr; c.decay = decay;
  c.inv_bc1 = (float)(1.0 / (1.0 - __builtin_pow((double)beta1, (double)(step_dev ? 1 : step))));   // Flux carries β₁ᵗ, β₂ᵗ as a running product
  c.inv_bc2 = (float)(1.0 / (1.0 - __builtin_pow((double)beta2, (double)(step_dev ? 1 : step))));
  bool launched = false;
  for (int first = 0; first < n;) {
    OptTable tab;
    tab.n = 0;
    int blk = 0;
    while (first < n && tab.n < OPT_MAXT) {
      if (t[first].n > 0) {
        const int64_t nb = (t[first].n + OPT_PER_WG - 1) / OPT_PER_WG;
        if (nb > 0x3fffffff - blk) break;   // next launch
        tab.t[tab.n] = t[first];
        tab.blk0[tab.n] = blk;
        blk += (int)nb;
        tab.n++;
      }
      first++;
    }
    if (tab.n == 0) {
      if (first < n) return LDE_ERR_INVALID_ARG;   // a single array beyond 2³⁰ workgroups
      break;
    }
    launched = true;
    tab.blk0[tab.n] = blk;
    bool more = false;   // another launch follows? (only the step's last one advances the count)
    for (int j = first; j < n; j++) more = more || t[j].n > 0;
    hipLaunchKernelGGL(k_adamw_flux, dim3(blk), dim3(OPT_WG), 0, (hipStream_t)stream, tab, c, step_dev, (step_dev && !more) ? 1 : 0);
    if (hipGetLastError() != hipSuccess) return LDE_ERR_HIP;
  }
  if (step_dev && !launched) {
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    if (hipGetLastError() != hipSuccess) return LDE_ERR_HIP;
  }
  return LDE_OK;
}

extern "C" int lde_adamw_flux_step(int n, const lde_adam_tensor* t, float lr, float beta1, float beta2, float eps, float decay,
                                   int64_t step, void* stream) {
  return adamw_impl(n, t, lr, beta1, beta2, eps, decay, step, nullptr, stream);
}
extern "C" int lde_adamw_flux_step_dev(int n, const lde_adam_tensor* t, float lr, float beta1, float beta2, float eps, float decay,
                                       int64_t* step_dev, void* stream) {
  if (!step_dev) return LDE_ERR_INVALID_ARG;
  return adamw_impl(n, t, lr, beta1, beta2, eps, decay, 0, step_dev, stream);
}

/*
 * lde_chain_oracle.c — CPU restatement of the dense chains either side of the solve (scope row f-1).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * reported baseline. The product (liblde.so) never links, loads or calls anything in here.
 *
 * PARITY UNPINNED against reference-produced vectors: the reference is Julia (Flux 0.13.6 `Chain`/`Dense`/
 * `SkipConnection`, un-vendored [REF Manifest.toml:452]) and cannot run in this image; its test suite holds no
 * vectors for this path [REF test/runtests.jl:4-6]. Pinned instead against an independent implementation of the same
 * published layer definitions: torch.nn.functional on CPU (tests/test_oracle_chain.py, fixtures in tests/golden/).
 *
 * What it follows
 *   apply_latent_out(decoder, l̃):  ẑ₀ = lo_z₀(z̃₀), θ̂ = lo_θ(θ̃)                 [REF src/models/GOKU.jl:83-91]
 *   apply_reconstructor(decoder, ẑ) = decoder.reconstructor(ẑ)                  [REF src/models/GOKU.jl:148]
 *   default layers: Chain(Dense(·,200,relu), SkipConnection(Dense(200,200,relu),+) ×2, Dense(200,784,σ)),
 *                   Chain(Dense(16,200,relu), Dense(200,D | P, identity | softplus))  [REF src/models/GOKU.jl:252-269]
 *   Dense(in,out,act)(x) = act.(W*x .+ b);  SkipConnection(l,+)(x) = l(x) + x;  σ(x) = 1/(1+e^-x);
 *   softplus(x) = log(1+e^x) (evaluated as max(x,0) + log1p(exp(-|x|)), NNlib's stable form).
 *   Flat weights = Flux.destructure order: per Dense vec(W) column-major [out×in], then b.
 *
 * Memory convention (same as the C ABI): x is [in × N] column-major, i.e. column n contiguous at x + n*in.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#include "../include/lde.h"

#ifdef ORACLE_F64
typedef double real;
#define r_exp exp
#define r_log1p log1p
#define r_tanh tanh
#define r_fabs fabs
#else
typedef float real;
#define r_exp expf
#define r_log1p log1pf
#define r_tanh tanhf
#define r_fabs fabsf
#endif

static real cact(int kind, real x) {
  switch (kind) {
    case LDE_CACT_RELU: return x > 0 ? x : (real)0;
    case LDE_CACT_TANH: return r_tanh(x);
    case LDE_CACT_SIGMOID: return (real)1 / ((real)1 + r_exp(-x));
    case LDE_CACT_SOFTPLUS: return (x > 0 ? x : (real)0) + r_log1p(r_exp(-r_fabs(x)));
    default: return x;
  }
}

/* derivative of the activation, from the pre-activation value */
static real cact_grad(int kind, real pre) {
  switch (kind) {
    case LDE_CACT_RELU: return pre > 0 ? (real)1 : (real)0;
    case LDE_CACT_TANH: { const real t = r_tanh(pre); return (real)1 - t * t; }
    case LDE_CACT_SIGMOID: { const real s = (real)1 / ((real)1 + r_exp(-pre)); return s * ((real)1 - s); }
    case LDE_CACT_SOFTPLUS: return (real)1 / ((real)1 + r_exp(-pre));
    default: return (real)1;
  }
}

static int chain_ok(const lde_chain_desc* d) {
  if (!d || d->n_layers < 1 || d->n_layers > LDE_CHAIN_MAX_LAYERS) return 0;
  for (int l = 0; l <= d->n_layers; l++)
    if (d->sizes[l] < 1) return 0;
  for (int l = 0; l < d->n_layers; l++) {
    if (d->activation[l] < 0 || d->activation[l] > LDE_CACT_SOFTPLUS) return 0;
    if (d->skip[l] && d->sizes[l] != d->sizes[l + 1]) return 0;
  }
  return 1;
}

int64_t oracle_chain_num_weights(const lde_chain_desc* d) {
  if (!chain_ok(d)) return -1;
  int64_t n = 0;
  for (int l = 0; l < d->n_layers; l++) n += (int64_t)d->sizes[l + 1] * d->sizes[l] + d->sizes[l + 1];
  return n;
}

/* forward of one column; act[l] (size sizes[l+1]) receives the layer OUTPUT, pre[l] the pre-activation */
static void col_forward(const lde_chain_desc* d, const real* W, const real* x, real** pre, real** out) {
  const real* in = x;
  const real* w = W;
  for (int l = 0; l < d->n_layers; l++) {
    const int ni = d->sizes[l], no = d->sizes[l + 1];
    const real* b = w + (size_t)no * ni;
    for (int o = 0; o < no; o++) {
      real s = b[o];
      for (int i = 0; i < ni; i++) s += w[o + (size_t)no * i] * in[i];
      pre[l][o] = s;
      real a = cact(d->activation[l], s);
      if (d->skip[l]) a += in[o];
      out[l][o] = a;
    }
    in = out[l];
    w = b + no;
  }
}

int oracle_chain_forward(const lde_chain_desc* d, const real* W, const real* x, int64_t N, real* y, int nthreads) {
  if (!chain_ok(d) || !W || !x || !y || N < 0) return LDE_ERR_INVALID_ARG;
  const int L = d->n_layers, ni0 = d->sizes[0], noL = d->sizes[L];
  int maxw = 1;
  for (int l = 1; l <= L; l++)
    if (d->sizes[l] > maxw) maxw = d->sizes[l];
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
  {
    real* buf = (real*)malloc(sizeof(real) * (size_t)maxw * 2 * L);
    real *pre[LDE_CHAIN_MAX_LAYERS], *out[LDE_CHAIN_MAX_LAYERS];
    for (int l = 0; l < L; l++) {
      pre[l] = buf + (size_t)maxw * (2 * l);
      out[l] = buf + (size_t)maxw * (2 * l + 1);
    }
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; n++) {
      col_forward(d, W, x + (size_t)n * ni0, pre, out);
      memcpy(y + (size_t)n * noL, out[L - 1], sizeof(real) * (size_t)noL);
    }
    free(buf);
  }
  return LDE_OK;
}

/* dx (may be NULL) is written; dW (n_weights) is accumulated (+=), like the C ABI. */
int oracle_chain_backward(const lde_chain_desc* d, const real* W, const real* x, const real* dy, int64_t N, real* dx,
                          real* dW, int nthreads) {
  if (!chain_ok(d) || !W || !x || !dy || !dW || N < 0) return LDE_ERR_INVALID_ARG;
  const int L = d->n_layers, ni0 = d->sizes[0], noL = d->sizes[L];
  const int64_t nW = oracle_chain_num_weights(d);
  int maxw = ni0;
  for (int l = 1; l <= L; l++)
    if (d->sizes[l] > maxw) maxw = d->sizes[l];
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  const real* wl[LDE_CHAIN_MAX_LAYERS];
  int64_t woff[LDE_CHAIN_MAX_LAYERS];
  {
    int64_t off = 0;
    for (int l = 0; l < L; l++) {
      wl[l] = W + off;
      woff[l] = off;
      off += (int64_t)d->sizes[l + 1] * d->sizes[l] + d->sizes[l + 1];
    }
  }
  real* partial = (real*)calloc((size_t)nthreads * (size_t)nW, sizeof(real));
  if (!partial) return LDE_ERR_ALLOC;
#pragma omp parallel num_threads(nthreads)
  {
    const int tid = omp_get_thread_num();
    real* gw = partial + (size_t)tid * (size_t)nW;
    real* buf = (real*)malloc(sizeof(real) * (size_t)maxw * (2 * L + 2));
    real *pre[LDE_CHAIN_MAX_LAYERS], *out[LDE_CHAIN_MAX_LAYERS];
    for (int l = 0; l < L; l++) {
      pre[l] = buf + (size_t)maxw * (2 * l);
      out[l] = buf + (size_t)maxw * (2 * l + 1);
    }
    real* g = buf + (size_t)maxw * (2 * L);       /* gradient wrt the current layer's output */
    real* gin = buf + (size_t)maxw * (2 * L + 1); /* gradient wrt its input */
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; n++) {
      const real* xc = x + (size_t)n * ni0;
      col_forward(d, W, xc, pre, out);
      memcpy(g, dy + (size_t)n * noL, sizeof(real) * (size_t)noL);
      for (int l = L - 1; l >= 0; l--) {
        const int ni = d->sizes[l], no = d->sizes[l + 1];
        const real* in = l == 0 ? xc : out[l - 1];
        const real* w = wl[l];
        real* gwl = gw + woff[l];
        for (int i = 0; i < ni; i++) gin[i] = d->skip[l] ? g[i] : (real)0;
        for (int o = 0; o < no; o++) {
          const real dl = g[o] * cact_grad(d->activation[l], pre[l][o]);
          gwl[(size_t)no * ni + o] += dl;
          for (int i = 0; i < ni; i++) {
            gwl[o + (size_t)no * i] += dl * in[i];
            gin[i] += w[o + (size_t)no * i] * dl;
          }
        }
        real* t = g; g = gin; gin = t;
      }
      if (dx) memcpy(dx + (size_t)n * ni0, g, sizeof(real) * (size_t)ni0);
    }
    free(buf);
  }
  for (int t = 0; t < nthreads; t++)
    for (int64_t i = 0; i < nW; i++) dW[i] += partial[(size_t)t * (size_t)nW + i];
  free(partial);
  return LDE_OK;
}

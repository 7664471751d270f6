/*
 * lde_loss_oracle.c — CPU restatement of the variational sample and the loss terms (scope row f-3).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as the checker. The product (liblde.so) never links, loads or calls it.
 *
 * PARITY UNPINNED against reference-produced vectors (the reference is Julia and cannot run in this image; its tests hold
 * no vectors for these functions [REF test/runtests.jl:4-6]). Pinned against closed forms worked out by hand and against
 * torch autograd on CPU (tests/test_oracle_loss.py).
 *
 * What it follows
 *   sample(μ, logσ²) = μ + ε·exp(logσ²/2), ε = randn                     [REF src/models/GOKU.jl:155-163]
 *   kl(μ, logσ²) = (exp(logσ²) + μ² − logσ² − 1)/2, vector_kl = Σ kl / B  [REF src/utils/utils.jl:15-49]
 *   reconstruction_loss = sum(mean((x − x̂)², dims=(2,3)))               [REF examples/pendulum_friction-less/model_train.jl:225-238]
 * Sums are accumulated in double whatever the storage precision (plain left-to-right loops).
 */
#include <math.h>
#include <stdint.h>

#ifdef ORACLE_F64
typedef double real;
#else
typedef float real;
#endif

void oracle_sample_forward(const real* mu, const real* logvar, const real* eps, int64_t n, real* l) {
  for (int64_t i = 0; i < n; i++) l[i] = (real)((double)mu[i] + (double)eps[i] * exp(0.5 * (double)logvar[i]));
}

/* dμ = dl; dlogvar = dl·ε·exp(logσ²/2)/2 */
void oracle_sample_backward(const real* logvar, const real* eps, const real* dl, int64_t n, real* dmu, real* dlogvar) {
  for (int64_t i = 0; i < n; i++) {
    dmu[i] = dl[i];
    dlogvar[i] = (real)(0.5 * (double)dl[i] * (double)eps[i] * exp(0.5 * (double)logvar[i]));
  }
}

double oracle_kl_forward(const real* mu, const real* logvar, int64_t n, double scale) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) s += 0.5 * (exp((double)logvar[i]) + (double)mu[i] * (double)mu[i] - (double)logvar[i] - 1.0);
  return scale * s;
}

void oracle_kl_backward(const real* mu, const real* logvar, int64_t n, double scale, double g, real* dmu, real* dlogvar) {
  for (int64_t i = 0; i < n; i++) {
    dmu[i] = (real)(g * scale * (double)mu[i]);
    dlogvar[i] = (real)(g * scale * 0.5 * (exp((double)logvar[i]) - 1.0));
  }
}

double oracle_mse_forward(const real* x, const real* xhat, int64_t n, double scale) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) {
    const double d = (double)x[i] - (double)xhat[i];
    s += d * d;
  }
  return scale * s;
}

void oracle_mse_backward(const real* x, const real* xhat, int64_t n, double scale, double g, real* dxhat) {
  for (int64_t i = 0; i < n; i++) dxhat[i] = (real)(2.0 * g * scale * ((double)xhat[i] - (double)x[i]));
}

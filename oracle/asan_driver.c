/* asan_driver.c — the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (test infrastructure; `make -C oracle asan`,
 * run by tests/test_sanitizers.py). GPU sanitizers are not available on this pool; SURVEY.md §5 "race detection / sanitizers" applies to
 * what runs on the CPU: the checker itself (everything the parity tests trust) and the host logic of the C ABI (tests/host_logic_driver.cpp).
 * Every family of entry points is driven at small, ragged and degenerate shapes: analytic / MLP / physics + MLP right-hand sides,
 * per-trajectory and coupled control, Tsit5 and fixed-step RK4, failing solves (NaN blocks), step records recorded, prescribed and
 * overflowing, the discrete sweep, the solve on dual numbers, single save time, single trajectory. Exit code 0 and a clean stderr = pass. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/lde.h"

#ifdef ORACLE_F64
typedef double real;
#else
typedef float real;
#endif
int oracle_forward(const lde_problem_desc* d, const real* W, const real* z0, const real* theta, const double* ts, int T, int B, real* z_out,
                   int32_t* retcode, int64_t* stats, double* dt_trace, int* n_trace, int max_trace, int nthreads);
int oracle_forward_steps(const lde_problem_desc* d, const real* W, const real* z0, const real* theta, const double* ts, int T, int B,
                         real* z_out, int32_t* retcode, int64_t* stats, double* rec_t, double* rec_dt, int32_t* rec_n, int rec_cap, int presc,
                         int nthreads);
int oracle_adjoint(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T, int B,
                   const real* dz_out, real* dz0, real* dtheta, real* dW, int64_t* stats, int nthreads);
int oracle_adjoint_discrete(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T, int B,
                            const real* dz_out, const double* rec_t, const double* rec_dt, const int32_t* rec_n, int rec_cap, real* dz0,
                            real* dtheta, real* dW, int64_t* stats, int nthreads);
int oracle_forward_dual(const lde_problem_desc* d, const real* z0, const real* theta, const double* ts, int T, int B, int dual_norm,
                        real* z_out, real* J_out, const real* dz_out, real* dz0, real* dtheta, int32_t* retcode, int64_t* stats,
                        double* rec_t, double* rec_dt, int32_t* rec_n, int rec_cap, int presc, int nthreads);
int64_t oracle_num_weights(const lde_problem_desc* d);

static unsigned long long rs = 88172645463325252ULL;
static double rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }

static lde_problem_desc desc(int rhs, int D, int P, int aug, int nl, const int* sizes, int solver, int batching, int sense) {
  lde_problem_desc d;
  memset(&d, 0, sizeof(d));
  d.abi_version = LDE_ABI_VERSION; d.rhs_kind = rhs; d.state_dim = D; d.param_dim = P; d.augment_dim = aug; d.n_layers = nl;
  for (int i = 0; i <= nl; i++) d.layer_sizes[i] = sizes[i];
  d.solver = solver; d.batching = batching; d.sensealg = sense; d.adaptive = solver == LDE_SOLVER_TSIT5; d.maxiters = 100000;
  d.dt = solver == LDE_SOLVER_RK4 ? 0.013 : 0.0; d.abstol = 1e-6; d.reltol = 1e-3; d.qmin = 0.2; d.qmax = 10; d.gamma = 0.9;
  d.beta1 = 0.14; d.beta2 = 0.08;
  return d;
}

static int run_case(lde_problem_desc d, int B, int T, int cap, long long maxiters) {
  const int D = d.state_dim, Dp = D + d.augment_dim, P = d.param_dim;
  const int64_t nW = oracle_num_weights(&d);
  d.maxiters = maxiters;
  /* exact-size heap buffers: an access one element past any of them is a report */
  real* W = nW ? (real*)malloc((size_t)nW * sizeof(real)) : NULL;
  real* z0 = (real*)malloc((size_t)B * D * sizeof(real));
  real* th = P ? (real*)malloc((size_t)B * P * sizeof(real)) : NULL;
  double* ts = (double*)malloc((size_t)T * sizeof(double));
  real* z = (real*)malloc((size_t)T * B * Dp * sizeof(real));
  real* dz = (real*)malloc((size_t)T * B * Dp * sizeof(real));
  real* g0 = (real*)malloc((size_t)B * D * sizeof(real));
  real* gth = (real*)malloc((size_t)B * (P ? P : 1) * sizeof(real));
  real* gW = (real*)calloc((size_t)(nW ? nW : 1), sizeof(real));
  int32_t* ret = (int32_t*)malloc((size_t)B * sizeof(int32_t));
  const int nseq = d.batching == LDE_BATCH_PER_TRAJECTORY ? B : 1;
  double* rt = (double*)calloc((size_t)nseq * cap, sizeof(double));
  double* rdt = (double*)calloc((size_t)nseq * cap, sizeof(double));
  int32_t* rn = (int32_t*)calloc((size_t)nseq, sizeof(int32_t));
  int64_t st[5];
  for (int64_t i = 0; i < nW; i++) W[i] = (real)(0.3 * (rnd() - 0.5));
  for (int i = 0; i < B * D; i++) z0[i] = (real)(rnd() - 0.5);
  for (int i = 0; i < B * P; i++) th[i] = (real)(1.0 + rnd());
  for (int j = 0; j < T; j++) ts[j] = 0.05 * j + (j ? 0.01 * rnd() : 0.0);
  for (int i = 0; i < T * B * Dp; i++) dz[i] = (real)((rnd() - 0.5) / (B * T));
  int rc = 0;
  rc |= oracle_forward(&d, W, z0, th, ts, T, B, z, ret, st, NULL, NULL, 0, 2);
  if (d.sensealg != LDE_SENSE_DISCRETE) rc |= oracle_adjoint(&d, W, z, th, ts, T, B, dz, g0, gth, gW, st, 2);   /* (the discrete sweep needs a record: below) */
  rc |= oracle_forward_steps(&d, W, z0, th, ts, T, B, z, ret, st, rt, rdt, rn, cap, 0, 2);
  int overflow = 0;
  for (int q = 0; q < nseq; q++) overflow |= rn[q] > cap;
  if (!overflow && T > 1) {
    rc |= oracle_forward_steps(&d, W, z0, th, ts, T, B, z, ret, st, rt, rdt, rn, cap, 1, 2);          /* the same steps, prescribed */
    rc |= oracle_adjoint_discrete(&d, W, z, th, ts, T, B, dz, rt, rdt, rn, cap, g0, gth, gW, st, 2);
  }
  if (d.rhs_kind <= LDE_RHS_PENDULUM_FRICTION && d.batching == LDE_BATCH_PER_TRAJECTORY) {
    real* J = (real*)malloc((size_t)T * B * 2 * 3 * sizeof(real));
    rc |= oracle_forward_dual(&d, z0, th, ts, T, B, 1, z, J, dz, g0, gth, ret, st, rt, rdt, rn, cap, 0, 2);
    rc |= oracle_forward_dual(&d, z0, th, ts, T, B, 0, z, NULL, NULL, NULL, NULL, ret, st, NULL, NULL, NULL, 0, 0, 1);
    free(J);
  }
  free(W); free(z0); free(th); free(ts); free(z); free(dz); free(g0); free(gth); free(gW); free(ret); free(rt); free(rdt); free(rn);
  if (rc) fprintf(stderr, "case rhs %d solver %d batching %d sense %d B %d T %d: rc %d\n", d.rhs_kind, d.solver, d.batching, d.sensealg, B, T, rc);
  return rc;
}

int main(void) {
  int rc = 0, n = 0;
  const int none[1] = {0}, c3[4] = {2, 16, 16, 2}, nd[4] = {5, 12, 12, 5}, aug[4] = {6, 8, 8, 6}, five[6] = {3, 7, 5, 6, 4, 3};
  const int shapes[][2] = {{1, 1}, {1, 2}, {3, 2}, {7, 13}, {33, 50}};
  for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); s++) {
    const int B = shapes[s][0], T = shapes[s][1];
    for (int solver = 0; solver < 2; solver++) {
      for (int kind = 0; kind < 2; kind++) {
        rc |= run_case(desc(kind, 2, 1, 0, 0, none, solver, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_DISCRETE), B, T, 256, 100000); n++;
        rc |= run_case(desc(kind, 2, 1, 0, 0, none, solver, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_PARALLEL_CHECKPOINTED), B, T, 256, 100000); n++;
      }
      rc |= run_case(desc(LDE_RHS_PENDULUM_PLUS_MLP, 2, 1, 0, 3, c3, solver, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_BACKSOLVE_CHECKPOINTED), B, T, 256, 100000); n++;
      rc |= run_case(desc(LDE_RHS_MLP, 5, 0, 0, 3, nd, solver, LDE_BATCH_COUPLED, LDE_SENSE_BACKSOLVE_CHECKPOINTED), B, T, 256, 100000); n++;
      rc |= run_case(desc(LDE_RHS_MLP, 4, 0, 2, 3, aug, solver, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_BACKSOLVE), B, T, 256, 100000); n++;
      rc |= run_case(desc(LDE_RHS_MLP, 3, 0, 0, 5, five, solver, LDE_BATCH_COUPLED, LDE_SENSE_DISCRETE), B, T, 256, 100000); n++;
    }
  }
  /* failing solves (maxiters exhausted: NaN blocks), records too small for the solve (overflow: counts run on past the capacity) */
  rc |= run_case(desc(0, 2, 1, 0, 0, none, 0, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_DISCRETE), 9, 50, 256, 3); n++;
  rc |= run_case(desc(0, 2, 1, 0, 0, none, 0, LDE_BATCH_PER_TRAJECTORY, LDE_SENSE_DISCRETE), 9, 50, 4, 100000); n++;
  rc |= run_case(desc(LDE_RHS_MLP, 5, 0, 0, 3, nd, 0, LDE_BATCH_COUPLED, LDE_SENSE_DISCRETE), 6, 50, 3, 100000); n++;
  printf("oracle under ASan + UBSan: %d cases, rc = %d\n", n, rc);
  return rc != 0;
}

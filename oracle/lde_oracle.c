/*
 * lde_oracle.c — CPU restatement of the latent-ODE forward solve + adjoint.
 *
 * *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (latentdiffeq.jl_amd → liblde.so, HIP) never calls into it.
 *
 * *** PARITY UNPINNED against the reference's own tests ***
 * The reference ships an empty test suite [REF test/runtests.jl:4-6], no golden vectors, and its
 * arithmetic lives in un-vendored Julia packages (OrdinaryDiffEq 6.27.1 [REF Manifest.toml:979],
 * SciMLSensitivity 7.10.0 [REF Manifest.toml:1200], DiffEqFlux 1.52.0 [REF Manifest.toml:304]);
 * there is no Julia in the build image. This file restates the *published* algorithms those
 * packages implement (Tsitouras 2011 5(4) pair with the free 4th-order interpolant; Hairer–Nørsett–
 * Wanner initial step; PI step controller; classical RK4; continuous adjoint with jumps at the
 * observation times) and is pinned instead by independent known-answer tests in tests/:
 * order conditions of the tableau, the exact pendulum (Jacobi elliptic functions), scipy DOP853,
 * scipy expm for linear RHS, float64 finite-difference gradients, torch autograd through RK4.
 *
 * What it follows in the reference (semantics, shapes, ordering):
 *   - diffeq_layer, GOKU:      per-trajectory ensemble, column i ↦ trajectory i, saveat=t,
 *                              NaN block on failure, [D×B×T] output   [REF src/models/GOKU.jl:98-130]
 *   - diffeq_layer, LatentODE: one ODE on the [D'×B] matrix state, optional zero augmentation
 *                                                                     [REF src/models/LatentODE.jl:61-78]
 *   - pendulum RHS / friction  [REF examples/pendulum_friction-less/pendulum.jl:19-26, :65-74]
 *   - NODE RHS MLP, Dense/relu, destructure order  [REF examples/pendulum_friction-less/nODE.jl:12-14]
 *
 * Build: see oracle/Makefile (compiled twice: REAL=float → liblde_oracle_f32.so,
 *                                             REAL=double → liblde_oracle_f64.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/lde.h"

#ifdef ORACLE_F64
typedef double real;
#define r_sin sin
#define r_cos cos
#define r_fabs fabs
#define r_sqrt sqrt
#define r_pow pow
#define r_tanh tanh
#define r_fmax fmax
#define r_fmin fmin
#else
typedef float real;
#define r_sin sinf
#define r_cos cosf
#define r_fabs fabsf
#define r_sqrt sqrtf
#define r_pow powf
#define r_tanh tanhf
#define r_fmax fmaxf
#define r_fmin fminf
#endif

/* ------------------------------------------------------------------------------------------ */
/* Tsit5 tableau (Tsitouras 2011) and its 4th-order continuous extension.                      */
static const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double TS_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {0.161, 0, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
static const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                -0.1447110071732629,     0.5823571654525552,     -0.45808210592918697,
                                0.015151515151515152};
/* b_1(Θ) = Θ(1 + Θ(r12 + Θ(r13 + Θ r14))) ; b_i(Θ) = Θ²(ri2 + Θ(ri3 + Θ ri4)), i=2..7 */
static const double TS_R1[3] = {-2.763706197274826, 2.9132554618219126, -1.0530884977290216};
static const double TS_R[6][3] = {{0.13169999999999998, -0.2234, 0.1017},
                                  {3.9302962368947516, -5.941033872131505, 2.490627285651253},
                                  {-12.411077166933676, 30.33818863028232, -16.548102889244902},
                                  {37.50931341651104, -88.1789048947664, 47.37952196281928},
                                  {-27.896526289197286, 65.09189467479366, -34.87065786149661},
                                  {1.5, -4.0, 2.5}};

static void tsit5_interp_weights(double th, double bw[7]) {
  bw[0] = th * (1.0 + th * (TS_R1[0] + th * (TS_R1[1] + th * TS_R1[2])));
  for (int i = 0; i < 6; i++) bw[i + 1] = th * th * (TS_R[i][0] + th * (TS_R[i][1] + th * TS_R[i][2]));
}

/* exported so the tests can check the order conditions of the constants compiled in here.
 * out: c[7], a[7*6], btilde[7], r1[3], r[6*3]  (= 7+42+7+3+18 = 77 doubles) */
void oracle_tsit5_tableau(double* out) {
  int k = 0;
  for (int i = 0; i < 7; i++) out[k++] = TS_C[i];
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 6; j++) out[k++] = TS_A[i][j];
  for (int i = 0; i < 7; i++) out[k++] = TS_BT[i];
  for (int i = 0; i < 3; i++) out[k++] = TS_R1[i];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 3; j++) out[k++] = TS_R[i][j];
}

int oracle_real_size(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------------------------------ */
/* RHS for ONE column (one trajectory).                                                        */
typedef struct {
  const lde_problem_desc* d;
  int Dp, P, nL;
  const real* W;          /* flat weights, destructure order */
  int64_t woff[LDE_MAX_LAYERS], boff[LDE_MAX_LAYERS];
  real* act[LDE_MAX_LAYERS + 1]; /* act[0] = input copy, act[l+1] = output of layer l (post-activation) */
  real* del[2];           /* backprop scratch */
  real* dW_step;          /* [nW] contributions of the current step attempt (NULL if none) */
  int64_t nW;
  double* margin;         /* (optional) running minimum of |pre-activation| / (Σ|w·x| + |b|) over the hidden units evaluated: how close this
                             column's solve came to a relu kink (tests: a gradient that differs between two f32 implementations by more than
                             round-off must have passed within round-off of one) */
} colrhs;

static int64_t num_weights(const lde_problem_desc* d) {
  int64_t n = 0;
  for (int l = 0; l < d->n_layers; l++) n += (int64_t)d->layer_sizes[l + 1] * d->layer_sizes[l] + d->layer_sizes[l + 1];
  return n;
}
int64_t oracle_num_weights(const lde_problem_desc* d) { return num_weights(d); }

static int has_mlp(const lde_problem_desc* d) { return d->rhs_kind == LDE_RHS_MLP || d->rhs_kind == LDE_RHS_PENDULUM_PLUS_MLP; }
static int has_pend(const lde_problem_desc* d) { return d->rhs_kind != LDE_RHS_MLP; }

static int colrhs_init(colrhs* c, const lde_problem_desc* d, const real* W) {
  memset(c, 0, sizeof(*c));
  c->d = d;
  c->Dp = d->state_dim + d->augment_dim;
  c->P = d->param_dim;
  c->nL = has_mlp(d) ? d->n_layers : 0;
  c->W = W;
  c->nW = has_mlp(d) ? num_weights(d) : 0;
  int64_t off = 0;
  int maxw = c->Dp;
  for (int l = 0; l < c->nL; l++) {
    c->woff[l] = off;
    off += (int64_t)d->layer_sizes[l + 1] * d->layer_sizes[l];
    c->boff[l] = off;
    off += d->layer_sizes[l + 1];
    if (d->layer_sizes[l + 1] > maxw) maxw = d->layer_sizes[l + 1];
    if (d->layer_sizes[l] > maxw) maxw = d->layer_sizes[l];
  }
  for (int l = 0; l <= c->nL; l++) c->act[l] = (real*)calloc((size_t)maxw, sizeof(real));
  c->del[0] = (real*)calloc((size_t)maxw, sizeof(real));
  c->del[1] = (real*)calloc((size_t)maxw, sizeof(real));
  if (c->nW) c->dW_step = (real*)calloc((size_t)c->nW, sizeof(real));
  return 0;
}
static void colrhs_free(colrhs* c) {
  for (int l = 0; l <= LDE_MAX_LAYERS; l++) free(c->act[l]);
  free(c->del[0]);
  free(c->del[1]);
  free(c->dW_step);
}

static real act_fn(int kind, real x) { return kind == LDE_ACT_TANH ? r_tanh(x) : (x > 0 ? x : (real)0); }
/* derivative expressed through the post-activation value a */
static real act_grad(int kind, real a) { return kind == LDE_ACT_TANH ? (real)1 - a * a : (a > 0 ? (real)1 : (real)0); }

static void mlp_forward(colrhs* c, const real* z) {
  const lde_problem_desc* d = c->d;
  for (int i = 0; i < d->layer_sizes[0]; i++) c->act[0][i] = z[i];
  for (int l = 0; l < c->nL; l++) {
    int in = d->layer_sizes[l], out = d->layer_sizes[l + 1];
    const real* Wl = c->W + c->woff[l]; /* col-major [out×in]: W(o,i) at o + out*i */
    const real* bl = c->W + c->boff[l];
    real* y = c->act[l + 1];
    for (int o = 0; o < out; o++) y[o] = bl[o];
    for (int i = 0; i < in; i++) {
      real xi = c->act[l][i];
      const real* col = Wl + (int64_t)out * i;
      for (int o = 0; o < out; o++) y[o] += col[o] * xi;
    }
    if (c->margin && l < c->nL - 1)
      for (int o = 0; o < out; o++) {
        double sa = fabs((double)bl[o]);
        for (int i = 0; i < in; i++) sa += fabs((double)Wl[o + (int64_t)out * i] * (double)c->act[l][i]);
        const double m = sa > 0 ? fabs((double)y[o]) / sa : 1.0;
        if (m < *c->margin) *c->margin = m;
      }
    if (l < c->nL - 1)
      for (int o = 0; o < out; o++) y[o] = act_fn(d->activation, y[o]);
  }
}

/* f(z, θ) for one column */
static void rhs_col(colrhs* c, const real* z, const real* th, real* out) {
  const lde_problem_desc* d = c->d;
  for (int i = 0; i < c->Dp; i++) out[i] = 0;
  if (c->nL) {
    mlp_forward(c, z);
    for (int i = 0; i < c->Dp; i++) out[i] = c->act[c->nL][i];
  }
  if (has_pend(d)) {
    const real G = (real)10;
    real L = th[0];
    out[0] += z[1];
    real acc = (-G / L) * r_sin(z[0]);
    if (d->rhs_kind == LDE_RHS_PENDULUM_FRICTION) acc -= ((real)0.7 / (real)1) * z[1];
    out[1] += acc;
  }
}

/* f, (∂f/∂z)ᵀλ, (∂f/∂θ)ᵀλ for one column, and dW_step += wq·(∂f/∂W)ᵀλ  (wq==0 → skipped) */
static void rhs_vjp_col(colrhs* c, const real* z, const real* th, const real* lam, real* f, real* vz, real* vth,
                        real wq) {
  const lde_problem_desc* d = c->d;
  for (int i = 0; i < c->Dp; i++) {
    f[i] = 0;
    vz[i] = 0;
  }
  for (int p = 0; p < c->P; p++) vth[p] = 0;
  if (c->nL) {
    mlp_forward(c, z);
    for (int i = 0; i < c->Dp; i++) f[i] = c->act[c->nL][i];
    real* dl = c->del[0];
    real* dn = c->del[1];
    for (int i = 0; i < c->Dp; i++) dl[i] = lam[i];
    for (int l = c->nL - 1; l >= 0; l--) {
      int in = d->layer_sizes[l], out = d->layer_sizes[l + 1];
      const real* Wl = c->W + c->woff[l];
      if (l < c->nL - 1)
        for (int o = 0; o < out; o++) dl[o] *= act_grad(d->activation, c->act[l + 1][o]);
      if (wq != 0 && c->dW_step) {
        real* gW = c->dW_step + c->woff[l];
        real* gb = c->dW_step + c->boff[l];
        for (int i = 0; i < in; i++) {
          real xi = wq * c->act[l][i];
          real* col = gW + (int64_t)out * i;
          for (int o = 0; o < out; o++) col[o] += dl[o] * xi;
        }
        for (int o = 0; o < out; o++) gb[o] += wq * dl[o];
      }
      for (int i = 0; i < in; i++) {
        const real* col = Wl + (int64_t)out * i;
        real s = 0;
        for (int o = 0; o < out; o++) s += col[o] * dl[o];
        dn[i] = s;
      }
      real* t = dl;
      dl = dn;
      dn = t;
    }
    for (int i = 0; i < c->Dp; i++) vz[i] = dl[i];
  }
  if (has_pend(d)) {
    const real G = (real)10;
    real L = th[0];
    real s = r_sin(z[0]), co = r_cos(z[0]);
    f[0] += z[1];
    real acc = (-G / L) * s;
    if (d->rhs_kind == LDE_RHS_PENDULUM_FRICTION) acc -= (real)0.7 * z[1];
    f[1] += acc;
    /* J = [[0,1],[-(G/L)cos x, -b/m]] ;  Jᵀλ = [-(G/L)cos x·λ₂, λ₁ - (b/m)λ₂] */
    vz[0] += (-G / L) * co * lam[1];
    vz[1] += lam[0];
    if (d->rhs_kind == LDE_RHS_PENDULUM_FRICTION) vz[1] -= (real)0.7 * lam[1];
    /* ∂f₂/∂L = (G/L²) sin x */
    vth[0] += (G / (L * L)) * s * lam[1];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Generic explicit RK drivers on an n-vector state.                                           */
typedef void (*ode_fn)(void* ctx, double t, const real* y, real* dy, real wq);
typedef void (*hook_fn)(void* ctx);

typedef struct {
  int adaptive, solver;
  double dt_fixed, abstol, reltol, dtmin, qmin, qmax, gamma, beta1, beta2;
  int64_t maxiters;
} sopts;

typedef struct {
  int64_t nfe, nacc, nrej;
  int retcode;
} sstat;

static void opts_from_desc(const lde_problem_desc* d, sopts* o, double t0, double t1) {
  o->adaptive = d->adaptive;
  o->solver = d->solver;
  o->dt_fixed = d->dt;
  o->abstol = d->abstol;
  o->reltol = d->reltol;
  o->dtmin = d->dtmin > 0 ? d->dtmin : 1e-12 * fabs(t1 - t0);
  o->qmin = d->qmin;
  o->qmax = d->qmax;
  o->gamma = d->gamma;
  o->beta1 = d->beta1;
  o->beta2 = d->beta2;
  o->maxiters = d->maxiters;
}

/* Coupled control with the batch sharded over ranks (LDE_BATCH_COUPLED_GLOBAL, include/lde.h): the checker's stand-in for the cross-rank
 * exchange. With a hook set, every sum of the step control of a COUPLED solve — the Hairer initial-step sums, each attempt's Σr², and
 * a non-finite flag — is passed through it (vals ← Σ over ranks, in place) and the norm's element count is scaled by g_nscale =
 * global batch / local batch. Per process (one solve at a time), set from Python with oracle_set_sum_hook. */
typedef int (*oracle_sum_hook)(void* user, double* vals, int n);
static oracle_sum_hook g_hook = 0;
static void* g_hook_user = 0;
static double g_nscale = 1.0;
static int g_hook_on = 0;   /* armed only inside a coupled solve */
void oracle_set_sum_hook(oracle_sum_hook hook, void* user, double nscale) {
  g_hook = hook;
  g_hook_user = user;
  g_nscale = hook ? nscale : 1.0;
}
static void global_sums(double* v, int n) {
  if (g_hook_on && g_hook) g_hook(g_hook_user, v, n);
}
static double global_n(int64_t n) { return (g_hook_on && g_hook) ? (double)n * g_nscale : (double)n; }

static int all_finite(const real* y, int64_t n) {
  for (int64_t i = 0; i < n; i++)
    if (!isfinite((double)y[i])) return 0;
  return 1;
}

/* Hairer–Nørsett–Wanner initial step (order 5); f0 = f(t0,y0) given. sign = +1/-1 direction. */
static double init_dt(ode_fn fn, void* ctx, int64_t n, double t0, const real* y0, const real* f0, double sign,
                      double dtmax, const sopts* o, real* tmp, real* f1, sstat* st) {
  real at = (real)o->abstol, rt = (real)o->reltol;
  double s0 = 0, s1 = 0;
  for (int64_t i = 0; i < n; i++) {
    real sk = at + r_fabs(y0[i]) * rt;
    real a = y0[i] / sk, b = f0[i] / sk;
    s0 += (double)(a * a);
    s1 += (double)(b * b);
  }
  { double v[2] = {s0, s1}; global_sums(v, 2); s0 = v[0]; s1 = v[1]; }
  double d0 = sqrt(s0 / global_n(n)), d1 = sqrt(s1 / global_n(n));
  double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  if (dt0 > dtmax) dt0 = dtmax;
  real h = (real)(sign * dt0);
  for (int64_t i = 0; i < n; i++) tmp[i] = y0[i] + h * f0[i];
  fn(ctx, t0 + sign * dt0, tmp, f1, 0);
  st->nfe++;
  double s2 = 0;
  for (int64_t i = 0; i < n; i++) {
    real sk = at + r_fabs(y0[i]) * rt;
    real a = (f1[i] - f0[i]) / sk;
    s2 += (double)(a * a);
  }
  global_sums(&s2, 1);
  double d2 = sqrt(s2 / global_n(n)) / dt0;
  double dm = d1 > d2 ? d1 : d2;
  double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / 5.0);
  double dt = fmin(100.0 * dt0, dt1);
  if (dt > dtmax) dt = dtmax;
  return dt;
}

typedef struct {
  int64_t n;
  real *k[7], *ynew, *tmp;
} rkwork;
static void rkwork_init(rkwork* w, int64_t n) {
  w->n = n;
  for (int i = 0; i < 7; i++) w->k[i] = (real*)calloc((size_t)n, sizeof(real));
  w->ynew = (real*)calloc((size_t)n, sizeof(real));
  w->tmp = (real*)calloc((size_t)n, sizeof(real));
}
static void rkwork_free(rkwork* w) {
  for (int i = 0; i < 7; i++) free(w->k[i]);
  free(w->ynew);
  free(w->tmp);
}

/* One Tsit5 attempt from (t,y) with signed step dt. k[0] must hold f(t,y) unless eval_k1.
 * wscale: quadrature weight scale (|dt| for the backward pass, 0 for forward); stage i gets wscale*b_i. */
static real tsit5_attempt(ode_fn fn, void* ctx, rkwork* w, double t, double dt, const real* y, int eval_k1,
                          double wscale, const sopts* o, sstat* st) {
  int64_t n = w->n;
  real h = (real)dt;
  if (eval_k1) {
    fn(ctx, t, y, w->k[0], (real)(wscale * TS_A[6][0]));
    st->nfe++;
  }
  for (int s = 1; s < 6; s++) {
    for (int64_t i = 0; i < n; i++) {
      real acc = (real)TS_A[s][0] * w->k[0][i];
      for (int j = 1; j < s; j++) acc += (real)TS_A[s][j] * w->k[j][i];
      w->tmp[i] = y[i] + h * acc;
    }
    fn(ctx, t + TS_C[s] * dt, w->tmp, w->k[s], (real)(wscale * TS_A[6][s]));
    st->nfe++;
  }
  for (int64_t i = 0; i < n; i++) {
    real acc = (real)TS_A[6][0] * w->k[0][i];
    for (int j = 1; j < 6; j++) acc += (real)TS_A[6][j] * w->k[j][i];
    w->ynew[i] = y[i] + h * acc;
  }
  fn(ctx, t + dt, w->ynew, w->k[6], 0);
  st->nfe++;
  if (!o->adaptive) return 0;
  real at = (real)o->abstol, rt = (real)o->reltol;
  double s2 = 0;
  for (int64_t i = 0; i < n; i++) {
    real e = (real)TS_BT[0] * w->k[0][i];
    for (int j = 1; j < 7; j++) e += (real)TS_BT[j] * w->k[j][i];
    e *= h;
    real sk = at + r_fmax(r_fabs(y[i]), r_fabs(w->ynew[i])) * rt;
    real r = e / sk;
    s2 += (double)(r * r);
  }
  if (g_hook_on && g_hook) {   /* a non-finite state on ANY rank must reject the attempt on every rank: it poisons the shared sum */
    if (!all_finite(w->ynew, n)) s2 = NAN;
    global_sums(&s2, 1);
  }
  return (real)sqrt(s2 / global_n(n));
}

/* classical RK4 step; k[0] = f(t,y) must be valid unless eval_k1. Leaves k[4] = f(t+dt, ynew) if want_fnew. */
static void rk4_step(ode_fn fn, void* ctx, rkwork* w, double t, double dt, const real* y, int eval_k1, double wscale,
                     int want_fnew, sstat* st) {
  int64_t n = w->n;
  real h = (real)dt, hh = (real)(0.5 * dt);
  if (eval_k1) {
    fn(ctx, t, y, w->k[0], (real)(wscale / 6.0));
    st->nfe++;
  }
  for (int64_t i = 0; i < n; i++) w->tmp[i] = y[i] + hh * w->k[0][i];
  fn(ctx, t + 0.5 * dt, w->tmp, w->k[1], (real)(wscale / 3.0));
  for (int64_t i = 0; i < n; i++) w->tmp[i] = y[i] + hh * w->k[1][i];
  fn(ctx, t + 0.5 * dt, w->tmp, w->k[2], (real)(wscale / 3.0));
  for (int64_t i = 0; i < n; i++) w->tmp[i] = y[i] + h * w->k[2][i];
  fn(ctx, t + dt, w->tmp, w->k[3], (real)(wscale / 6.0));
  st->nfe += 3;
  real h6 = (real)(dt / 6.0);
  for (int64_t i = 0; i < n; i++)
    w->ynew[i] = y[i] + h6 * (w->k[0][i] + (real)2 * (w->k[1][i] + w->k[2][i]) + w->k[3][i]);
  if (want_fnew) {
    fn(ctx, t + dt, w->ynew, w->k[4], 0);
    st->nfe++;
  }
}

/* PI controller (OrdinaryDiffEq PIController): returns q; *q11_out for the reject branch */
static real pi_q(real EEst, real qold, const sopts* o, real* q11_out) {
  real q;
  if (EEst == 0) {
    q = (real)(1.0 / o->qmax);
    *q11_out = 0;
  } else {
    real q11 = r_pow(EEst, (real)o->beta1);
    q = q11 / r_pow(qold, (real)o->beta2);
    *q11_out = q11;
    q = r_fmax((real)(1.0 / o->qmax), r_fmin((real)(1.0 / o->qmin), q / (real)o->gamma));
  }
  return q;
}

/* The accepted steps of ONE solve (one per trajectory, or one for a coupled solve): start time and step size of step n. Written by an
 * adaptive solve (presc = 0) or READ by a solve that is to take exactly these steps (presc = 1: no error control, every step is
 * accepted) — the latter is how the tests give the checker the step sequence a kernel recorded, so that the two are compared on the
 * SAME discrete solve instead of through two step-size controllers that wander apart (tests/test_gpu_discrete.py). */
typedef struct {
  double* t;     /* [cap] (forward only; NULL in a reverse-time record) */
  double* dt;    /* [cap] step size (forward) / magnitude (reverse) as f64; the state advances by (real)dt */
  int32_t* n;    /* steps recorded / prescribed */
  int cap;
  int presc;
} steprec;

/* Forward solve with saveat: out[j*n + i] = y_i(ts[j]). dt_trace (optional): accepted dt's. */
static void solve_forward(ode_fn fn, void* ctx, int64_t n, const real* y0, const double* ts, int T, const sopts* o,
                          real* out, sstat* st, double* dt_trace, int* n_trace, int max_trace, steprec* rec) {
  rkwork w;
  rkwork_init(&w, n);
  real* y = (real*)malloc((size_t)n * sizeof(real));
  memcpy(y, y0, (size_t)n * sizeof(real));
  memcpy(out, y0, (size_t)n * sizeof(real));
  st->retcode = LDE_RET_SUCCESS;
  int ntr = 0;
  if (T > 1) {
    double t = ts[0], tend = ts[T - 1], dtmax = tend - t;
    int j = 1;
    fn(ctx, t, y, w.k[0], 0);
    st->nfe++;
    double dt;
    if (o->adaptive)
      dt = o->dt_fixed > 0 ? fmin(o->dt_fixed, dtmax) : init_dt(fn, ctx, n, t, y, w.k[0], 1.0, dtmax, o, w.tmp, w.k[1], st);
    else
      dt = o->dt_fixed;
    real qold = (real)1e-4;
    int64_t iters = 0;
    const int presc = rec && rec->presc;
    int adaptive = o->adaptive && !presc;
    while (t < tend) {
      if (iters++ >= o->maxiters) { st->retcode = LDE_RET_MAXITERS; break; }
      double dtp = dt; /* controller proposal */
      int last = 0;
      if (presc) {   /* step ntr of the prescribed sequence: its recorded start time and size; the last one ends at tend */
        if (ntr >= *rec->n) { st->retcode = LDE_RET_MAXITERS; break; }
        t = rec->t[ntr];
        dt = rec->dt[ntr];
        last = ntr == *rec->n - 1;
      } else if (t + dt >= tend - 1e-12 * fabs(tend)) { dt = tend - t; last = 1; }
      real EEst = 0;
      if (o->solver == LDE_SOLVER_TSIT5)
        EEst = tsit5_attempt(fn, ctx, &w, t, dt, y, 0, 0.0, o, st);
      else
        rk4_step(fn, ctx, &w, t, dt, y, 0, 0.0, 1, st);
      if (!all_finite(w.ynew, n) || !(EEst == EEst)) {
        if (adaptive && dt > o->dtmin) { /* treat as a rejected step with maximal shrink */
          st->nrej++;
          dt = dt * o->qmin;
          continue;
        }
        st->retcode = LDE_RET_NONFINITE;
        break;
      }
      if (adaptive) {
        real q11, q = pi_q(EEst, qold, o, &q11);
        if (EEst > (real)1) {
          st->nrej++;
          dt = dt / (double)r_fmin((real)(1.0 / o->qmin), q11 / (real)o->gamma);
          if (dt < o->dtmin) { st->retcode = LDE_RET_DTMIN; break; }
          continue;
        }
        qold = r_fmax(EEst, (real)1e-4);
        dtp = dt / (double)q;
        if (dtp > dtmax) dtp = dtmax;
      }
      st->nacc++;
      if (dt_trace && ntr < max_trace) dt_trace[ntr] = dt;
      if (rec && !presc && ntr < rec->cap) { rec->t[ntr] = t; rec->dt[ntr] = dt; }
      ntr++;
      double tnew = last ? tend : (presc ? rec->t[ntr] : t + dt);
      while (j < T && ts[j] <= tnew) {
        double th = (ts[j] - t) / dt;
        real* o_j = out + (int64_t)j * n;
        if (th >= 1.0 || (j == T - 1 && last)) {
          memcpy(o_j, w.ynew, (size_t)n * sizeof(real));
        } else if (o->solver == LDE_SOLVER_TSIT5) {
          double bw[7];
          tsit5_interp_weights(th, bw);
          real h = (real)dt;
          for (int64_t i = 0; i < n; i++) {
            real acc = (real)bw[0] * w.k[0][i];
            for (int s = 1; s < 7; s++) acc += (real)bw[s] * w.k[s][i];
            o_j[i] = y[i] + h * acc;
          }
        } else { /* cubic Hermite between (y,k1) and (ynew,f(ynew)) */
          double h00 = (1 + 2 * th) * (1 - th) * (1 - th), h10 = th * (1 - th) * (1 - th);
          double h01 = th * th * (3 - 2 * th), h11 = th * th * (th - 1);
          for (int64_t i = 0; i < n; i++)
            o_j[i] = (real)h00 * y[i] + (real)(h10 * dt) * w.k[0][i] + (real)h01 * w.ynew[i] + (real)(h11 * dt) * w.k[4][i];
        }
        j++;
      }
      memcpy(y, w.ynew, (size_t)n * sizeof(real));
      real* fs = (o->solver == LDE_SOLVER_TSIT5) ? w.k[6] : w.k[4];
      real* t0 = w.k[0];
      if (o->solver == LDE_SOLVER_TSIT5) { w.k[0] = fs; w.k[6] = t0; } else { w.k[0] = fs; w.k[4] = t0; }
      t = tnew;
      dt = o->adaptive ? dtp : o->dt_fixed;
    }
    if (rec && !presc) *rec->n = ntr;   /* (> cap: the record is incomplete — the caller checks) */
  }
  if (st->retcode != LDE_RET_SUCCESS) {
    for (int64_t i = 0; i < (int64_t)T * n; i++) out[i] = (real)NAN;
  }
  if (n_trace) *n_trace = ntr;
  free(y);
  rkwork_free(&w);
}

/* Backward solve from ts[T-1] to ts[0] with a forced stop + jump at every save time.
 * y holds the state at ts[T-1] (first jump already applied). jump(ctx, j, y) is called on arrival at ts[j].
 * begin/commit/discard bracket the dW quadrature of one step attempt. */
typedef void (*jump_fn)(void* ctx, int j, real* y);
static void solve_backward(ode_fn fn, void* ctx, int64_t n, real* y, const double* ts, int T, const sopts* o,
                           jump_fn jump, hook_fn begin, hook_fn commit, sstat* st, steprec* rec) {
  rkwork w;
  rkwork_init(&w, n);
  st->retcode = LDE_RET_SUCCESS;
  if (T > 1) {
    double t = ts[T - 1], t0 = ts[0], dtmax = fabs(t - t0);
    double dt; /* magnitude of the proposed step */
    if (o->adaptive) {
      if (o->dt_fixed > 0) dt = fmin(o->dt_fixed, dtmax);
      else {
        fn(ctx, t, y, w.k[0], 0);
        st->nfe++;
        dt = init_dt(fn, ctx, n, t, y, w.k[0], -1.0, dtmax, o, w.tmp, w.k[1], st);
      }
    } else
      dt = o->dt_fixed;
    real qold = (real)1e-4;
    int64_t iters = 0;
    int j = T - 2;
    const int presc = rec && rec->presc;
    const int adaptive = o->adaptive && !presc;
    int ntr = 0;
    while (j >= 0) {
      if (iters++ >= o->maxiters) { st->retcode = LDE_RET_MAXITERS; break; }
      double tstop = ts[j];
      double dist = t - tstop;
      if (presc) {   /* the recorded magnitude of accepted step ntr (already clipped to its save time where it hit one) */
        if (ntr >= *rec->n) { st->retcode = LDE_RET_MAXITERS; break; }
        dt = rec->dt[ntr];
      }
      double h = dt;
      int hit = 0;
      if (h >= dist * (1.0 - 1e-12)) { h = dist; hit = 1; }
      if (begin) begin(ctx);
      real EEst = 0;
      if (o->solver == LDE_SOLVER_TSIT5)
        EEst = tsit5_attempt(fn, ctx, &w, t, -h, y, 1, h, o, st);
      else
        rk4_step(fn, ctx, &w, t, -h, y, 1, h, 0, st);
      if (!all_finite(w.ynew, n) || !(EEst == EEst)) {
        if (adaptive && h > o->dtmin) { st->nrej++; dt = h * o->qmin; continue; }
        st->retcode = LDE_RET_NONFINITE;
        break;
      }
      double dtp = dt;
      if (adaptive) {
        real q11, q = pi_q(EEst, qold, o, &q11);
        if (EEst > (real)1) {
          st->nrej++;
          dt = h / (double)r_fmin((real)(1.0 / o->qmin), q11 / (real)o->gamma);
          if (dt < o->dtmin) { st->retcode = LDE_RET_DTMIN; break; }
          continue;
        }
        qold = r_fmax(EEst, (real)1e-4);
        dtp = h / (double)q;
        if (dtp > dtmax) dtp = dtmax;
      }
      st->nacc++;
      if (rec && !presc && ntr < rec->cap) rec->dt[ntr] = h;
      ntr++;
      if (commit) commit(ctx);
      memcpy(y, w.ynew, (size_t)n * sizeof(real));
      if (hit) {
        t = tstop;
        jump(ctx, j, y);
        j--;
      } else
        t -= h;
      dt = o->adaptive ? dtp : o->dt_fixed;
    }
    if (rec && !presc) *rec->n = ntr;
  }
  rkwork_free(&w);
}

/* ------------------------------------------------------------------------------------------ */
/* Problem-level contexts: a block of `ncol` columns integrated as ONE ODE state                */
/* (ncol = 1 → per-trajectory / EnsembleProblem semantics; ncol = B → coupled / NeuralODE).    */
typedef struct {
  colrhs c;
  int ncol, Dp, P;
  const real* theta;   /* [P × ncol] for this block */
  /* backward only */
  const real* zsave;   /* z_out base for this block: element (d,b,j) at d + Dp*(b + Bstride*j) */
  const real* dzout;
  int64_t Bstride;     /* full batch size B (stride between save times) */
  int checkpoint;
  double* dW_acc;      /* [nW] committed quadrature (double accumulation) */
  colrhs* cs;          /* coupled mode with nth > 1: one RHS scratch per OpenMP thread (columns of a stage evaluation are
                          independent — the reference gets the same from OpenBLAS threads under its per-stage sgemms) */
  int nth;
  double* col_margin;  /* (optional) [ncol] relu-kink margins of the block's columns (discrete sweep) */
} blockctx;

static void block_threads_init(blockctx* b, const lde_problem_desc* d, const real* W, int nthreads) {
  b->nth = 1;
  b->cs = NULL;
#ifdef _OPENMP
  if (nthreads > 1) {
    b->nth = nthreads;
    b->cs = (colrhs*)calloc((size_t)nthreads, sizeof(colrhs));
    for (int t = 0; t < nthreads; t++) colrhs_init(&b->cs[t], d, W);
  }
#else
  (void)d; (void)W; (void)nthreads;
#endif
}
static void block_threads_free(blockctx* b) {
  if (b->cs) {
    for (int t = 0; t < b->nth; t++) colrhs_free(&b->cs[t]);
    free(b->cs);
  }
}

static void fwd_fn(void* vctx, double t, const real* y, real* dy, real wq) {
  (void)t; (void)wq;
  blockctx* b = (blockctx*)vctx;
#ifdef _OPENMP
  if (b->nth > 1) {
#pragma omp parallel for num_threads(b->nth) schedule(static)
    for (int c = 0; c < b->ncol; c++)
      rhs_col(&b->cs[omp_get_thread_num()], y + (int64_t)c * b->Dp, b->theta + (int64_t)c * b->P, dy + (int64_t)c * b->Dp);
    return;
  }
#endif
  for (int c = 0; c < b->ncol; c++) rhs_col(&b->c, y + (int64_t)c * b->Dp, b->theta + (int64_t)c * b->P, dy + (int64_t)c * b->Dp);
}

/* backward state layout: [ Z (Dp·ncol) | Λ (Dp·ncol) | Gθ (P·ncol) ] */
static void bwd_fn(void* vctx, double t, const real* y, real* dy, real wq) {
  (void)t;
  blockctx* b = (blockctx*)vctx;
  int64_t nz = (int64_t)b->Dp * b->ncol;
#ifdef _OPENMP
  if (b->nth > 1) {
#pragma omp parallel for num_threads(b->nth) schedule(static)
    for (int c = 0; c < b->ncol; c++) {
      real vz[1024], vth[16];
      const real* z = y + (int64_t)c * b->Dp;
      const real* lam = y + nz + (int64_t)c * b->Dp;
      real* fz = dy + (int64_t)c * b->Dp;
      real* dl = dy + nz + (int64_t)c * b->Dp;
      real* dg = dy + 2 * nz + (int64_t)c * b->P;
      colrhs* cc = &b->cs[omp_get_thread_num()];
      cc->margin = b->col_margin ? b->col_margin + c : NULL;
      rhs_vjp_col(cc, z, b->theta + (int64_t)c * b->P, lam, fz, vz, vth, wq);
      cc->margin = NULL;
      for (int i = 0; i < b->Dp; i++) dl[i] = -vz[i];
      for (int p = 0; p < b->P; p++) dg[p] = -vth[p];
    }
    return;
  }
#endif
  real vz[1024]; /* Dp <= 1024 (check_desc) */
  real vth[16];
  for (int c = 0; c < b->ncol; c++) {
    const real* z = y + (int64_t)c * b->Dp;
    const real* lam = y + nz + (int64_t)c * b->Dp;
    real* fz = dy + (int64_t)c * b->Dp;
    real* dl = dy + nz + (int64_t)c * b->Dp;
    real* dg = dy + 2 * nz + (int64_t)c * b->P;
    b->c.margin = b->col_margin ? b->col_margin + c : NULL;
    rhs_vjp_col(&b->c, z, b->theta + (int64_t)c * b->P, lam, fz, vz, vth, wq);
    b->c.margin = NULL;
    for (int i = 0; i < b->Dp; i++) dl[i] = -vz[i];
    for (int p = 0; p < b->P; p++) dg[p] = -vth[p];
  }
}

static void bwd_jump(void* vctx, int j, real* y) {
  blockctx* b = (blockctx*)vctx;
  int64_t nz = (int64_t)b->Dp * b->ncol;
  for (int c = 0; c < b->ncol; c++)
    for (int i = 0; i < b->Dp; i++) {
      int64_t src = i + (int64_t)b->Dp * (c + b->Bstride * j);
      y[nz + (int64_t)c * b->Dp + i] += b->dzout[src];
      if (b->checkpoint) y[(int64_t)c * b->Dp + i] = b->zsave[src];
    }
}
static void bwd_begin(void* vctx) {
  blockctx* b = (blockctx*)vctx;
  if (b->c.dW_step) memset(b->c.dW_step, 0, (size_t)b->c.nW * sizeof(real));
  if (b->cs)
    for (int t = 0; t < b->nth; t++)
      if (b->cs[t].dW_step) memset(b->cs[t].dW_step, 0, (size_t)b->cs[t].nW * sizeof(real));
}
static void bwd_commit(void* vctx) {
  blockctx* b = (blockctx*)vctx;
  if (b->c.dW_step && b->dW_acc)
    for (int64_t i = 0; i < b->c.nW; i++) b->dW_acc[i] += (double)b->c.dW_step[i];
  if (b->cs && b->dW_acc)
    for (int t = 0; t < b->nth; t++)   /* fixed thread order: deterministic for a given thread count */
      if (b->cs[t].dW_step)
        for (int64_t i = 0; i < b->cs[t].nW; i++) b->dW_acc[i] += (double)b->cs[t].dW_step[i];
}

static int check_desc(const lde_problem_desc* d) {
  if (!d || d->abi_version != LDE_ABI_VERSION) return LDE_ERR_INVALID_ARG;
  if (d->state_dim < 1 || d->param_dim < 0 || d->param_dim > 16 || d->augment_dim < 0) return LDE_ERR_INVALID_ARG;
  if (d->state_dim + d->augment_dim > 1024) return LDE_ERR_INVALID_ARG;
  if (has_pend(d) && (d->state_dim != 2 || d->param_dim != 1 || d->augment_dim != 0)) return LDE_ERR_INVALID_ARG;
  if (has_mlp(d)) {
    if (d->n_layers < 1 || d->n_layers > LDE_MAX_LAYERS) return LDE_ERR_INVALID_ARG;
    int Dp = d->state_dim + d->augment_dim;
    if (d->layer_sizes[0] != Dp || d->layer_sizes[d->n_layers] != Dp) return LDE_ERR_INVALID_ARG;
  }
  if (d->solver == LDE_SOLVER_RK4 && d->adaptive) return LDE_ERR_UNSUPPORTED;
  if (!d->adaptive && !(d->dt > 0)) return LDE_ERR_INVALID_ARG;
  return LDE_OK;
}

/* out layouts as in include/lde.h. stats: [nfe, naccept, nreject, nfailed, max_steps].
 * dt_trace/n_trace: accepted step sizes of trajectory 0 (or of the coupled solve). */
static int forward_impl(const lde_problem_desc* d, const real* W, const real* z0, const real* theta, const double* ts, int T,
                        int B, real* z_out, int32_t* retcode, int64_t* stats, double* dt_trace, int* n_trace, int max_trace,
                        double* rec_t, double* rec_dt, int32_t* rec_n, int rec_cap, int presc, int nthreads) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (T < 1 || B < 1) return LDE_ERR_INVALID_ARG;
  const int D = d->state_dim, Dp = D + d->augment_dim, P = d->param_dim;
  sopts o;
  opts_from_desc(d, &o, ts[0], ts[T - 1]);
  int64_t nfe = 0, nacc = 0, nrej = 0, nfail = 0, maxsteps = 0;
  if (n_trace) *n_trace = 0;
  if (d->batching != LDE_BATCH_PER_TRAJECTORY) {
    g_hook_on = d->batching == LDE_BATCH_COUPLED_GLOBAL;
    blockctx b;
    memset(&b, 0, sizeof(b));
    colrhs_init(&b.c, d, W);
    b.ncol = B; b.Dp = Dp; b.P = P; b.theta = theta;
    block_threads_init(&b, d, W, nthreads);
    int64_t n = (int64_t)Dp * B;
    real* y0 = (real*)calloc((size_t)n, sizeof(real));
    for (int c = 0; c < B; c++)
      for (int i = 0; i < D; i++) y0[(int64_t)c * Dp + i] = z0[(int64_t)c * D + i];
    sstat st = {0, 0, 0, 0};
    /* out as [T][Dp*B] is exactly the [Dp × B × T] column-major layout */
    steprec rec = {rec_t, rec_dt, rec_n, rec_cap, presc};
    solve_forward(fwd_fn, &b, n, y0, ts, T, &o, z_out, &st, dt_trace, n_trace, max_trace, rec_dt ? &rec : NULL);
    for (int c = 0; c < B; c++)
      if (retcode) retcode[c] = st.retcode;
    nfe = st.nfe; nacc = st.nacc; nrej = st.nrej; nfail = st.retcode ? B : 0; maxsteps = st.nacc + st.nrej;
    free(y0);
    block_threads_free(&b);
    colrhs_free(&b.c);
    g_hook_on = 0;
  } else {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : nfe, nacc, nrej, nfail) reduction(max : maxsteps)
    {
      blockctx b;
      memset(&b, 0, sizeof(b));
      colrhs_init(&b.c, d, W);
      b.ncol = 1; b.Dp = Dp; b.P = P;
      real* y0 = (real*)calloc((size_t)Dp, sizeof(real));
      real* out = (real*)calloc((size_t)Dp * T, sizeof(real));
#pragma omp for schedule(static)
      for (int c = 0; c < B; c++) {
        for (int i = 0; i < Dp; i++) y0[i] = i < D ? z0[(int64_t)c * D + i] : (real)0;
        b.theta = theta ? theta + (int64_t)c * P : NULL;
        sstat st = {0, 0, 0, 0};
        steprec rec = {rec_t ? rec_t + (int64_t)c * rec_cap : NULL, rec_dt ? rec_dt + (int64_t)c * rec_cap : NULL, rec_n ? rec_n + c : NULL,
                       rec_cap, presc};
        solve_forward(fwd_fn, &b, Dp, y0, ts, T, &o, out, &st, c == 0 ? dt_trace : NULL, c == 0 ? n_trace : NULL, max_trace,
                      rec_dt ? &rec : NULL);
        for (int j = 0; j < T; j++)
          for (int i = 0; i < Dp; i++) z_out[i + (int64_t)Dp * (c + (int64_t)B * j)] = out[(int64_t)j * Dp + i];
        if (retcode) retcode[c] = st.retcode;
        nfe += st.nfe; nacc += st.nacc; nrej += st.nrej; nfail += st.retcode ? 1 : 0;
        if (st.nacc + st.nrej > maxsteps) maxsteps = st.nacc + st.nrej;
      }
      free(y0);
      free(out);
      colrhs_free(&b.c);
    }
  }
  if (stats) { stats[0] = nfe; stats[1] = nacc; stats[2] = nrej; stats[3] = nfail; stats[4] = maxsteps; }
  return LDE_OK;
}

int oracle_forward(const lde_problem_desc* d, const real* W, const real* z0, const real* theta, const double* ts, int T,
                   int B, real* z_out, int32_t* retcode, int64_t* stats, double* dt_trace, int* n_trace, int max_trace,
                   int nthreads) {
  return forward_impl(d, W, z0, theta, ts, T, B, z_out, retcode, stats, dt_trace, n_trace, max_trace, NULL, NULL, NULL, 0, 0, nthreads);
}
/* The same solve with its accepted steps recorded (presc = 0) or prescribed (presc = 1): rec_t, rec_dt [nseq][rec_cap], rec_n [nseq],
 * nseq = B (per-trajectory control) or 1 (coupled). */
int oracle_forward_steps(const lde_problem_desc* d, const real* W, const real* z0, const real* theta, const double* ts, int T,
                         int B, real* z_out, int32_t* retcode, int64_t* stats, double* rec_t, double* rec_dt, int32_t* rec_n,
                         int rec_cap, int presc, int nthreads) {
  if (!rec_t || !rec_dt || !rec_n || rec_cap < 1) return LDE_ERR_INVALID_ARG;
  return forward_impl(d, W, z0, theta, ts, T, B, z_out, retcode, stats, NULL, NULL, 0, rec_t, rec_dt, rec_n, rec_cap, presc, nthreads);
}

/* ---- parallel-in-time checkpointed adjoint (LDE_SENSE_PARALLEL_CHECKPOINTED), analytic RHS ------------------ */
/* interval state: [ z (Dp) | Λ (Dp basis vectors × Dp) | G (Dp basis vectors × P) ] */
static void basis_fn(void* vctx, double t, const real* y, real* dy, real wq) {
  (void)t; (void)wq;
  blockctx* b = (blockctx*)vctx;
  const int Dp = b->Dp, P = b->P;
  real f[64], vz[64], vth[16];
  for (int v = 0; v < Dp; v++) {
    rhs_vjp_col(&b->c, y, b->theta, y + Dp + (int64_t)v * Dp, f, vz, vth, 0);
    for (int i = 0; i < Dp; i++) dy[Dp + (int64_t)v * Dp + i] = -vz[i];
    for (int p = 0; p < P; p++) dy[Dp + (int64_t)Dp * Dp + (int64_t)v * P + p] = -vth[p];
  }
  for (int i = 0; i < Dp; i++) dy[i] = f[i];
}
static void no_jump(void* ctx, int j, real* y) { (void)ctx; (void)j; (void)y; }

static int adjoint_parallel(const lde_problem_desc* d, const real* z_out, const real* theta, const double* ts, int T, int B,
                            const real* dz_out, real* dz0, real* dtheta, int64_t* stats, int nthreads) {
  const int Dp = d->state_dim, P = d->param_dim;
  if (has_mlp(d) || d->batching != LDE_BATCH_PER_TRAJECTORY || Dp > 8) return LDE_ERR_UNSUPPORTED;
  int64_t nfe = 0, nacc = 0, nrej = 0, nfail = 0, maxsteps = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : nfe, nacc, nrej, nfail) reduction(max : maxsteps)
  {
    blockctx b;
    memset(&b, 0, sizeof(b));
    colrhs_init(&b.c, d, NULL);
    b.ncol = 1; b.Dp = Dp; b.P = P;
    const int n = Dp + Dp * Dp + Dp * P;
    real* y = (real*)calloc((size_t)n, sizeof(real));
    real* M = (real*)calloc((size_t)(T > 1 ? T - 1 : 1) * (Dp * Dp + Dp * P), sizeof(real));
#pragma omp for schedule(static)
    for (int c = 0; c < B; c++) {
      b.theta = theta + (int64_t)c * P;
      int bad = 0;
      int64_t steps = 0;
      for (int j = T - 2; j >= 0 && !bad; j--) {  /* independent intervals (a GPU runs them concurrently) */
        for (int i = 0; i < n; i++) y[i] = 0;
        for (int i = 0; i < Dp; i++) {
          y[i] = z_out[i + (int64_t)Dp * (c + (int64_t)B * (j + 1))];
          y[Dp + i * Dp + i] = 1;
          if (!isfinite((double)y[i])) bad = 1;
        }
        if (bad) break;
        double tsj[2] = {ts[j], ts[j + 1]};
        sopts o;
        opts_from_desc(d, &o, ts[0], ts[T - 1]);
        if (o.adaptive) o.dt_fixed = ts[j + 1] - ts[j];   /* first attempt = the whole interval */
        sstat st = {0, 0, 0, 0};
        solve_backward(basis_fn, &b, n, y, tsj, 2, &o, no_jump, NULL, NULL, &st, NULL);
        if (st.retcode) bad = 1;
        memcpy(M + (int64_t)j * (Dp * Dp + Dp * P), y + Dp, (size_t)(Dp * Dp + Dp * P) * sizeof(real));
        nfe += st.nfe; nacc += st.nacc; nrej += st.nrej; steps += st.nacc + st.nrej;
      }
      if (T > 0) {
        real lam[8], nl[8], g[16];
        for (int i = 0; i < Dp; i++) {
          lam[i] = dz_out[i + (int64_t)Dp * (c + (int64_t)B * (T - 1))];
          if (!isfinite((double)z_out[i + (int64_t)Dp * (c + (int64_t)B * (T - 1))])) bad = 1;
        }
        for (int p = 0; p < P; p++) g[p] = 0;
        for (int j = T - 2; j >= 0 && !bad; j--) {  /* the scan: λ_j = M_j λ_{j+1} + Δ_j ; g += n_j · λ_{j+1} */
          const real* Mj = M + (int64_t)j * (Dp * Dp + Dp * P);
          for (int i = 0; i < Dp; i++) {
            real s = 0;
            for (int v = 0; v < Dp; v++) s += Mj[v * Dp + i] * lam[v];
            nl[i] = s + dz_out[i + (int64_t)Dp * (c + (int64_t)B * j)];
          }
          for (int p = 0; p < P; p++) {
            real s = g[p];
            for (int v = 0; v < Dp; v++) s += Mj[Dp * Dp + v * P + p] * lam[v];
            g[p] = s;
          }
          for (int i = 0; i < Dp; i++) lam[i] = nl[i];
        }
        for (int i = 0; i < Dp; i++) dz0[(int64_t)c * Dp + i] = bad ? 0 : lam[i];
        for (int p = 0; p < P; p++) dtheta[(int64_t)c * P + p] = bad ? 0 : g[p];
      }
      nfail += bad;
      if (steps > maxsteps) maxsteps = steps;
    }
    free(y);
    free(M);
    colrhs_free(&b.c);
  }
  if (stats) { stats[0] = nfe; stats[1] = nacc; stats[2] = nrej; stats[3] = nfail; stats[4] = maxsteps; }
  return LDE_OK;
}

static int adjoint_discrete(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T, int B,
                            const real* dz_out, const double* rec_t, const double* rec_dt, const int32_t* rec_n, int rec_cap, real* dz0,
                            real* dtheta, real* dW, int64_t* stats, double* margins, int nthreads);

/* dW is ACCUMULATED (+=), as in lde_adjoint. rec_*: the reverse-time solve's accepted step magnitudes, recorded or prescribed. */
static int adjoint_impl(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts,
                        int T, int B, const real* dz_out, real* dz0, real* dtheta, real* dW, int64_t* stats, double* rec_dt,
                        int32_t* rec_n, int rec_cap, int presc, double* margins, int nthreads) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (T < 1 || B < 1) return LDE_ERR_INVALID_ARG;
  if (margins)
    for (int c = 0; c < B; c++) margins[c] = 1.0;
  if (d->sensealg == LDE_SENSE_DISCRETE) return LDE_ERR_INVALID_ARG;   /* needs the forward record: oracle_adjoint_discrete */
  if (d->sensealg == LDE_SENSE_PARALLEL_CHECKPOINTED && !has_mlp(d) && d->batching == LDE_BATCH_PER_TRAJECTORY && !rec_dt)
    return adjoint_parallel(d, z_out, theta, ts, T, B, dz_out, dz0, dtheta, stats, nthreads);
  const int D = d->state_dim, Dp = D + d->augment_dim, P = d->param_dim;
  const int64_t nW = has_mlp(d) ? num_weights(d) : 0;
  sopts o;
  opts_from_desc(d, &o, ts[0], ts[T - 1]);
  int64_t nfe = 0, nacc = 0, nrej = 0, nfail = 0, maxsteps = 0;
  double* dW_tot = nW ? (double*)calloc((size_t)nW, sizeof(double)) : NULL;
  const int ckpt = d->sensealg != LDE_SENSE_BACKSOLVE;
  if (d->batching != LDE_BATCH_PER_TRAJECTORY) {
    g_hook_on = d->batching == LDE_BATCH_COUPLED_GLOBAL;
    blockctx b;
    memset(&b, 0, sizeof(b));
    colrhs_init(&b.c, d, W);
    b.ncol = B; b.Dp = Dp; b.P = P; b.theta = theta;
    b.zsave = z_out; b.dzout = dz_out; b.Bstride = B; b.checkpoint = ckpt; b.dW_acc = dW_tot;
    b.col_margin = margins;
    block_threads_init(&b, d, W, nthreads);
    int64_t nz = (int64_t)Dp * B, n = 2 * nz + (int64_t)P * B;
    real* y = (real*)calloc((size_t)n, sizeof(real));
    int bad = 0;
    for (int64_t i = 0; i < nz; i++) {
      y[i] = z_out[i + nz * (T - 1)];
      y[nz + i] = dz_out[i + nz * (T - 1)];
      if (!isfinite((double)y[i])) bad = 1;
    }
    if (g_hook_on && g_hook) {   /* one shard's failed forward solve fails the shared solve on every rank */
      double v = bad;
      global_sums(&v, 1);
      bad = v != 0;
    }
    sstat st = {0, 0, 0, 0};
    steprec rec = {NULL, rec_dt, rec_n, rec_cap, presc};
    if (!bad) solve_backward(bwd_fn, &b, n, y, ts, T, &o, bwd_jump, bwd_begin, bwd_commit, &st, rec_dt ? &rec : NULL);
    for (int c = 0; c < B; c++) {
      for (int i = 0; i < D; i++) dz0[(int64_t)c * D + i] = bad ? 0 : y[nz + (int64_t)c * Dp + i];
      for (int p = 0; p < P; p++) dtheta[(int64_t)c * P + p] = bad ? 0 : y[2 * nz + (int64_t)c * P + p];
    }
    nfe = st.nfe; nacc = st.nacc; nrej = st.nrej; nfail = (bad || st.retcode) ? B : 0; maxsteps = st.nacc + st.nrej;
    free(y);
    block_threads_free(&b);
    colrhs_free(&b.c);
    g_hook_on = 0;
  } else {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : nfe, nacc, nrej, nfail) reduction(max : maxsteps)
    {
      blockctx b;
      memset(&b, 0, sizeof(b));
      colrhs_init(&b.c, d, W);
      b.ncol = 1; b.Dp = Dp; b.P = P; b.Bstride = B; b.checkpoint = ckpt;
      b.dW_acc = nW ? (double*)calloc((size_t)nW, sizeof(double)) : NULL;
      int n = 2 * Dp + P;
      real* y = (real*)calloc((size_t)n, sizeof(real));
#pragma omp for schedule(static)
      for (int c = 0; c < B; c++) {
        b.theta = theta ? theta + (int64_t)c * P : NULL;
        b.zsave = z_out + (int64_t)Dp * c;
        b.dzout = dz_out + (int64_t)Dp * c;
        b.col_margin = margins ? margins + c : NULL;
        int bad = 0;
        for (int i = 0; i < Dp; i++) {
          int64_t src = i + (int64_t)Dp * ((int64_t)B * (T - 1));
          y[i] = b.zsave[src];
          y[Dp + i] = b.dzout[src];
          if (!isfinite((double)y[i])) bad = 1;
        }
        for (int p = 0; p < P; p++) y[2 * Dp + p] = 0;
        sstat st = {0, 0, 0, 0};
        steprec rec = {NULL, rec_dt ? rec_dt + (int64_t)c * rec_cap : NULL, rec_n ? rec_n + c : NULL, rec_cap, presc};
        if (rec_n && !presc) rec_n[c] = 0;
        if (!bad) solve_backward(bwd_fn, &b, n, y, ts, T, &o, bwd_jump, bwd_begin, bwd_commit, &st, rec_dt ? &rec : NULL);
        if (st.retcode) bad = 1;
        for (int i = 0; i < D; i++) dz0[(int64_t)c * D + i] = bad ? 0 : y[Dp + i];
        for (int p = 0; p < P; p++) dtheta[(int64_t)c * P + p] = bad ? 0 : y[2 * Dp + p];
        nfe += st.nfe; nacc += st.nacc; nrej += st.nrej; nfail += bad;
        if (st.nacc + st.nrej > maxsteps) maxsteps = st.nacc + st.nrej;
      }
      if (nW) {
#pragma omp critical
        for (int64_t i = 0; i < nW; i++) dW_tot[i] += b.dW_acc[i];
        free(b.dW_acc);
      }
      free(y);
      colrhs_free(&b.c);
    }
  }
  if (nW && dW)
    for (int64_t i = 0; i < nW; i++) dW[i] += (real)dW_tot[i];
  free(dW_tot);
  if (stats) { stats[0] = nfe; stats[1] = nacc; stats[2] = nrej; stats[3] = nfail; stats[4] = maxsteps; }
  return LDE_OK;
}

int oracle_adjoint(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts,
                   int T, int B, const real* dz_out, real* dz0, real* dtheta, real* dW, int64_t* stats, int nthreads) {
  return adjoint_impl(d, W, z_out, theta, ts, T, B, dz_out, dz0, dtheta, dW, stats, NULL, NULL, 0, 0, NULL, nthreads);
}
/* The continuous adjoint (sensealg 0 / 1) with the reverse-time solve's accepted step magnitudes recorded (presc = 0) or prescribed
 * (presc = 1): rec_dt [nseq][rec_cap], rec_n [nseq], nseq = B or 1 as in oracle_forward_steps. */
int oracle_adjoint_steps(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts,
                         int T, int B, const real* dz_out, real* dz0, real* dtheta, real* dW, int64_t* stats, double* rec_dt,
                         int32_t* rec_n, int rec_cap, int presc, int nthreads) {
  if (!rec_dt || !rec_n || rec_cap < 1) return LDE_ERR_INVALID_ARG;
  return adjoint_impl(d, W, z_out, theta, ts, T, B, dz_out, dz0, dtheta, dW, stats, rec_dt, rec_n, rec_cap, presc, NULL, nthreads);
}
/* … with the relu-kink margins of the reverse-time solve (see oracle_adjoint_discrete_margins): margins[B]. */
int oracle_adjoint_steps_margins(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts,
                                 int T, int B, const real* dz_out, real* dz0, real* dtheta, real* dW, int64_t* stats, double* rec_dt,
                                 int32_t* rec_n, int rec_cap, int presc, double* margins, int nthreads) {
  if (!rec_dt || !rec_n || rec_cap < 1) return LDE_ERR_INVALID_ARG;
  return adjoint_impl(d, W, z_out, theta, ts, T, B, dz_out, dz0, dtheta, dW, stats, rec_dt, rec_n, rec_cap, presc, margins, nthreads);
}

/* ---- discrete (exact) sensitivity: LDE_SENSE_DISCRETE ----------------------------------------------------------------------------
 * What `ForwardDiffSensitivity()` — the GOKU default [REF examples/pendulum_friction-less/pendulum.jl:11], splatted into solve at
 * [REF src/models/GOKU.jl:107, :121] — differentiates: the discrete solve itself, on its accepted step sequence, the step sizes being
 * plain Float64 numbers (not duals), i.e. constants of the differentiation. SciMLSensitivity 7.10.0 [REF Manifest.toml:1200] gets it
 * by pushing dual numbers through the stepper; the same derivative is restated here in reverse mode (one sweep instead of D + P):
 *   step n:  g_i = y_n + h Σ_{j<i} a_ij k_j,  k_i = f(g_i)  (i = 1..S);  y_{n+1} = y_n + h Σ b_i k_i;  k_{S+1} = f(y_{n+1})  (FSAL: the next k_1)
 *   saveat:  ẑ(t_j) = y_n + h Σ_{i≤S+1} w_i(Θ_j) k_i  (Tsit5's free interpolant; RK4: the cubic Hermite on (y_n, k_1, y_{n+1}, k_{S+1}))
 *   reverse: k̄_{S+1} = (next step's k̄_1) + h w_{S+1}(Θ_j) Δ_j;  ȳ_{n+1} += J(y_{n+1})ᵀ k̄_{S+1};  k̄_i += h b_i ȳ_{n+1} + h w_i(Θ_j) Δ_j;
 *            ȳ_n = ȳ_{n+1} + Σ_j Δ_j;  for i = S..2: ḡ = J(g_i)ᵀ k̄_i, ȳ_n += ḡ, k̄_j += h a_ij ḡ (j < i);  k̄_1 travels to step n − 1
 *            (at n = 0: ȳ_0 += J(y_0)ᵀ k̄_1).  θ̄ and W̄ collect (∂f/∂θ)ᵀ k̄_i, (∂f/∂W)ᵀ k̄_i of every evaluation.
 * One block of `ncol` columns sharing ONE step sequence (ncol = 1: a GOKU trajectory; ncol = B: the coupled NeuralODE solve). */
typedef struct {
  int S;                 /* stages before the FSAL evaluation: 6 (Tsit5) / 4 (RK4) */
  double A[7][7];        /* a_ij, row S = the solution weights b */
} dtab;
static void dtab_init(dtab* tb, int solver) {
  memset(tb, 0, sizeof(*tb));
  if (solver == LDE_SOLVER_TSIT5) {
    tb->S = 6;
    for (int i = 0; i < 7; i++)
      for (int j = 0; j < 6; j++) tb->A[i][j] = TS_A[i][j];
  } else {
    tb->S = 4;
    tb->A[1][0] = 0.5; tb->A[2][1] = 0.5; tb->A[3][2] = 1.0;
    tb->A[4][0] = 1.0 / 6.0; tb->A[4][1] = 1.0 / 3.0; tb->A[4][2] = 1.0 / 3.0; tb->A[4][3] = 1.0 / 6.0;
  }
}

/* f for the block's columns */
static void blk_f(blockctx* b, const real* y, real* dy) {
  if (!b->col_margin) { fwd_fn(b, 0.0, y, dy, 0); return; }
  for (int c = 0; c < b->ncol; c++) {
    b->c.margin = b->col_margin + c;
    rhs_col(&b->c, y + (int64_t)c * b->Dp, b->theta + (int64_t)c * b->P, dy + (int64_t)c * b->Dp);
  }
  b->c.margin = NULL;
}
/* vz += J(y)ᵀ kb; gth += (∂f/∂θ)ᵀ kb; dW_acc += (∂f/∂W)ᵀ kb — skipped altogether when kb is identically zero.
 * A coupled block with nth > 1 spreads its columns over OpenMP threads (as the continuous adjoint's bwd_fn does — the role OpenBLAS
 * threads play under the reference's per-stage sgemms): per-thread scratch, the threads' weight-gradient sums added in thread order. */
static void blk_vjp(blockctx* b, const real* y, const real* kb, real* vz_add, real* gth, sstat* st) {
  int64_t n = (int64_t)b->Dp * b->ncol;
  int any = 0;
  for (int64_t i = 0; i < n && !any; i++) any = kb[i] != 0;
  if (!any) return;
#ifdef _OPENMP
  if (b->nth > 1 && !b->col_margin) {
    for (int t = 0; t < b->nth; t++)
      if (b->cs[t].dW_step) memset(b->cs[t].dW_step, 0, (size_t)b->cs[t].nW * sizeof(real));
#pragma omp parallel for num_threads(b->nth) schedule(static)
    for (int c = 0; c < b->ncol; c++) {
      real f[1024], vz[1024], vth[16];
      rhs_vjp_col(&b->cs[omp_get_thread_num()], y + (int64_t)c * b->Dp, b->theta + (int64_t)c * b->P, kb + (int64_t)c * b->Dp, f, vz, vth, (real)1);
      for (int i = 0; i < b->Dp; i++) vz_add[(int64_t)c * b->Dp + i] += vz[i];
      for (int p = 0; p < b->P; p++) gth[(int64_t)c * b->P + p] += vth[p];
    }
    if (b->dW_acc)
      for (int t = 0; t < b->nth; t++)
        if (b->cs[t].dW_step)
          for (int64_t i = 0; i < b->cs[t].nW; i++) b->dW_acc[i] += (double)b->cs[t].dW_step[i];
    st->nfe++;
    return;
  }
#endif
  real f[1024], vz[1024], vth[16];
  for (int c = 0; c < b->ncol; c++) {
    if (b->c.dW_step) memset(b->c.dW_step, 0, (size_t)b->c.nW * sizeof(real));
    b->c.margin = b->col_margin ? b->col_margin + c : NULL;
    rhs_vjp_col(&b->c, y + (int64_t)c * b->Dp, b->theta + (int64_t)c * b->P, kb + (int64_t)c * b->Dp, f, vz, vth, (real)1);
    b->c.margin = NULL;
    for (int i = 0; i < b->Dp; i++) vz_add[(int64_t)c * b->Dp + i] += vz[i];
    for (int p = 0; p < b->P; p++) gth[(int64_t)c * b->P + p] += vth[p];
    if (b->c.dW_step && b->dW_acc)
      for (int64_t i = 0; i < b->c.nW; i++) b->dW_acc[i] += (double)b->c.dW_step[i];
  }
  st->nfe++;
}

/* returns 0, or a retcode when the record is unusable (then the caller reports zeros, as for a failed trajectory) */
static int discrete_block(blockctx* b, const lde_problem_desc* d, const real* y0, const double* ts, int T, const double* rt,
                          const double* rdt, int ns, real* ybar /* out: ∂L/∂y_0 [n] */, real* gth /* out [P·ncol] */, sstat* st) {
  const int64_t n = (int64_t)b->Dp * b->ncol;
  dtab tb;
  dtab_init(&tb, d->solver);
  const int S = tb.S;
  const double tend = ts[T - 1];
  real* Y = (real*)malloc((size_t)(ns + 1) * n * sizeof(real));
  int* jlo = (int*)malloc((size_t)(ns + 1) * sizeof(int));   /* save times of step s: jlo[s] ≤ j < jlo[s+1] */
  real *k[8], *g[8], *kb[8];
  for (int i = 0; i < 8; i++) {
    k[i] = (real*)calloc((size_t)n, sizeof(real));
    g[i] = (real*)calloc((size_t)n, sizeof(real));
    kb[i] = (real*)calloc((size_t)n, sizeof(real));
  }
  real* yb_n = (real*)calloc((size_t)n, sizeof(real));
  real* gbar = (real*)calloc((size_t)n, sizeof(real));
  memcpy(Y, y0, (size_t)n * sizeof(real));
  /* pass 1: the states y_n of the recorded steps (the forward solve's own arithmetic) and which save times each step serves */
  int j = 1, rcode = 0;
  for (int s = 0; s < ns; s++) {
    const real h = (real)rdt[s];
    const real* y = Y + (int64_t)s * n;
    real* yn = Y + (int64_t)(s + 1) * n;
    blk_f(b, y, k[0]);
    for (int i = 1; i < S; i++) {
      for (int64_t e = 0; e < n; e++) {
        real acc = (real)tb.A[i][0] * k[0][e];
        for (int q = 1; q < i; q++) acc += (real)tb.A[i][q] * k[q][e];
        g[i][e] = y[e] + h * acc;
      }
      blk_f(b, g[i], k[i]);
    }
    if (d->solver == LDE_SOLVER_TSIT5) {
      for (int64_t e = 0; e < n; e++) {
        real acc = (real)tb.A[S][0] * k[0][e];
        for (int q = 1; q < S; q++) acc += (real)tb.A[S][q] * k[q][e];
        yn[e] = y[e] + h * acc;
      }
    } else {
      const real h6 = (real)(rdt[s] / 6.0);
      for (int64_t e = 0; e < n; e++) yn[e] = y[e] + h6 * (k[0][e] + (real)2 * (k[1][e] + k[2][e]) + k[3][e]);
    }
    st->nfe += S;
    if (!all_finite(yn, n)) { rcode = LDE_RET_NONFINITE; break; }
    const double tnew = s == ns - 1 ? tend : rt[s + 1];
    jlo[s] = j;
    while (j < T && ts[j] <= tnew) j++;
  }
  jlo[ns] = j;
  if (!rcode && j < T) rcode = LDE_RET_MAXITERS;   /* the record does not reach the last save time */
  for (int64_t e = 0; e < n; e++) ybar[e] = 0;
  for (int64_t e = 0; e < (int64_t)b->P * b->ncol; e++) gth[e] = 0;
  real* carry = kb[7];   /* k̄_1 of the step behind (in time: ahead of) the current one */
  /* pass 2 */
  for (int s = ns - 1; s >= 0 && !rcode; s--) {
    const double t = rt[s], dt = rdt[s];
    const real h = (real)dt;
    const int last = s == ns - 1;
    const real* y = Y + (int64_t)s * n;
    const real* yn = Y + (int64_t)(s + 1) * n;
    /* the stage points again (k_i are needed only to form them) */
    blk_f(b, y, k[0]);
    for (int i = 1; i < S; i++) {
      for (int64_t e = 0; e < n; e++) {
        real acc = (real)tb.A[i][0] * k[0][e];
        for (int q = 1; q < i; q++) acc += (real)tb.A[i][q] * k[q][e];
        g[i][e] = y[e] + h * acc;
      }
      if (i < S - 1) blk_f(b, g[i], k[i]);   /* (k_S is not needed to form a stage point) */
    }
    st->nfe += S - 1;
    for (int i = 0; i < S; i++) memset(kb[i], 0, (size_t)n * sizeof(real));
    memcpy(kb[S], carry, (size_t)n * sizeof(real));
    memset(yb_n, 0, (size_t)n * sizeof(real));
    /* the save times inside the step */
    for (int jj = jlo[s]; jj < jlo[s + 1]; jj++) {
      const double th = (ts[jj] - t) / dt;
      const int at_end = th >= 1.0 || (jj == T - 1 && last);
      for (int c = 0; c < b->ncol; c++)
        for (int i = 0; i < b->Dp; i++) {
          const int64_t e = (int64_t)c * b->Dp + i;
          const real D_ = b->dzout[i + (int64_t)b->Dp * (c + b->Bstride * jj)];
          if (at_end) ybar[e] += D_;
          else if (d->solver == LDE_SOLVER_TSIT5) {
            double bw[7];
            tsit5_interp_weights(th, bw);
            yb_n[e] += D_;
            for (int q = 0; q < 7; q++) kb[q][e] += (h * (real)bw[q]) * D_;
          } else {
            const double h00 = (1 + 2 * th) * (1 - th) * (1 - th), h10 = th * (1 - th) * (1 - th);
            const double h01 = th * th * (3 - 2 * th), h11 = th * th * (th - 1);
            yb_n[e] += (real)h00 * D_;
            kb[0][e] += (real)(h10 * dt) * D_;
            ybar[e] += (real)h01 * D_;
            kb[4][e] += (real)(h11 * dt) * D_;
          }
        }
    }
    /* the FSAL evaluation at y_{n+1} */
    blk_vjp(b, yn, kb[S], ybar, gth, st);
    /* y_{n+1} = y_n + h Σ b_i k_i */
    for (int i = 0; i < S; i++)
      for (int64_t e = 0; e < n; e++) kb[i][e] += (h * (real)tb.A[S][i]) * ybar[e];
    for (int64_t e = 0; e < n; e++) yb_n[e] += ybar[e];
    /* stages S..2 */
    for (int i = S - 1; i >= 1; i--) {
      memset(gbar, 0, (size_t)n * sizeof(real));
      blk_vjp(b, g[i], kb[i], gbar, gth, st);
      for (int64_t e = 0; e < n; e++) yb_n[e] += gbar[e];
      for (int q = 0; q < i; q++)
        if (tb.A[i][q] != 0)
          for (int64_t e = 0; e < n; e++) kb[q][e] += (h * (real)tb.A[i][q]) * gbar[e];
    }
    memcpy(carry, kb[0], (size_t)n * sizeof(real));
    memcpy(ybar, yb_n, (size_t)n * sizeof(real));
    st->nacc++;
  }
  if (!rcode) blk_vjp(b, Y, carry, ybar, gth, st);   /* k_1 of the first step = f(y_0) */
  for (int i = 0; i < 8; i++) { free(k[i]); free(g[i]); free(kb[i]); }
  free(yb_n); free(gbar); free(Y); free(jlo);
  return rcode;
}

static int adjoint_discrete(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T, int B,
                            const real* dz_out, const double* rec_t, const double* rec_dt, const int32_t* rec_n, int rec_cap, real* dz0,
                            real* dtheta, real* dW, int64_t* stats, double* margins, int nthreads) {
  int rc = check_desc(d);
  if (margins)
    for (int c = 0; c < B; c++) margins[c] = 1.0;
  if (rc) return rc;
  if (T < 1 || B < 1 || !rec_t || !rec_dt || !rec_n) return LDE_ERR_INVALID_ARG;
  const int D = d->state_dim, Dp = D + d->augment_dim, P = d->param_dim;
  const int64_t nW = has_mlp(d) ? num_weights(d) : 0;
  int64_t nfe = 0, nacc = 0, nfail = 0, maxsteps = 0;
  double* dW_tot = nW ? (double*)calloc((size_t)nW, sizeof(double)) : NULL;
  const int coupled = d->batching != LDE_BATCH_PER_TRAJECTORY;
  const int nblk = coupled ? 1 : B, ncol = coupled ? B : 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel if (!coupled) reduction(+ : nfe, nacc, nfail) reduction(max : maxsteps)   /* a coupled solve is ONE block: its columns are threaded instead */
  {
    blockctx b;
    memset(&b, 0, sizeof(b));
    colrhs_init(&b.c, d, W);
    b.ncol = ncol; b.Dp = Dp; b.P = P; b.Bstride = B;
    b.dW_acc = nW ? (double*)calloc((size_t)nW, sizeof(double)) : NULL;
    if (coupled) block_threads_init(&b, d, W, nthreads);   /* (one block: the parallel region above has one thread at work; the columns get the rest) */
    const int64_t n = (int64_t)Dp * ncol;
    real* y0 = (real*)calloc((size_t)n, sizeof(real));
    real* ybar = (real*)calloc((size_t)n, sizeof(real));
    real* gth = (real*)calloc((size_t)(P * ncol + 1), sizeof(real));
#pragma omp for schedule(static)
    for (int blk = 0; blk < nblk; blk++) {
      const int c0 = coupled ? 0 : blk;
      b.theta = theta ? theta + (int64_t)c0 * P : NULL;
      b.dzout = dz_out + (int64_t)Dp * c0;
      b.col_margin = margins ? margins + c0 : NULL;
      int bad = 0;
      for (int c = 0; c < ncol; c++)
        for (int i = 0; i < Dp; i++) {
          y0[(int64_t)c * Dp + i] = z_out[i + (int64_t)Dp * (c0 + c)];   /* save time 0 is ẑ₀ itself (augmented rows 0) */
          if (!isfinite((double)y0[(int64_t)c * Dp + i])) bad = 1;
        }
      const int ns = rec_n[blk];
      if (ns > rec_cap || (T > 1 && ns < 1)) bad = 1;
      sstat st = {0, 0, 0, 0};
      if (!bad && T > 1) bad = discrete_block(&b, d, y0, ts, T, rec_t + (int64_t)blk * rec_cap, rec_dt + (int64_t)blk * rec_cap, ns, ybar, gth, &st) != 0;
      else if (!bad) {   /* T == 1: the output is ẑ₀ */
        for (int64_t e = 0; e < n; e++) ybar[e] = 0;
        for (int e = 0; e < P * ncol; e++) gth[e] = 0;
      }
      if (!bad && T >= 1)   /* save time 0: ẑ(t_1) = ẑ₀ */
        for (int c = 0; c < ncol; c++)
          for (int i = 0; i < Dp; i++) ybar[(int64_t)c * Dp + i] += dz_out[i + (int64_t)Dp * (c0 + c)];
      for (int c = 0; c < ncol; c++) {
        for (int i = 0; i < D; i++) dz0[(int64_t)(c0 + c) * D + i] = bad ? 0 : ybar[(int64_t)c * Dp + i];
        for (int p = 0; p < P; p++) dtheta[(int64_t)(c0 + c) * P + p] = bad ? 0 : gth[(int64_t)c * P + p];
      }
      nfe += st.nfe; nacc += st.nacc; nfail += bad ? ncol : 0;
      if (st.nacc > maxsteps) maxsteps = st.nacc;
    }
    if (nW) {
#pragma omp critical
      for (int64_t i = 0; i < nW; i++) dW_tot[i] += b.dW_acc[i];
      free(b.dW_acc);
    }
    free(y0); free(ybar); free(gth);
    block_threads_free(&b);
    colrhs_free(&b.c);
  }
  if (nW && dW)
    for (int64_t i = 0; i < nW; i++) dW[i] += (real)dW_tot[i];
  free(dW_tot);
  if (stats) { stats[0] = nfe; stats[1] = nacc; stats[2] = 0; stats[3] = nfail; stats[4] = maxsteps; }
  return LDE_OK;
}
int oracle_adjoint_discrete(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T, int B,
                            const real* dz_out, const double* rec_t, const double* rec_dt, const int32_t* rec_n, int rec_cap, real* dz0,
                            real* dtheta, real* dW, int64_t* stats, int nthreads) {
  return adjoint_discrete(d, W, z_out, theta, ts, T, B, dz_out, rec_t, rec_dt, rec_n, rec_cap, dz0, dtheta, dW, stats, NULL, nthreads);
}
/* … and how close each trajectory's discrete solve came to a relu kink: margins[B] = min over its hidden-unit evaluations of
 * |pre-activation| / (Σ|w·x| + |b|) (1 for tanh networks' purposes too: the number is what it is, the tests use it for relu). */
int oracle_adjoint_discrete_margins(const lde_problem_desc* d, const real* W, const real* z_out, const real* theta, const double* ts, int T,
                                    int B, const real* dz_out, const double* rec_t, const double* rec_dt, const int32_t* rec_n, int rec_cap,
                                    real* dz0, real* dtheta, real* dW, int64_t* stats, double* margins, int nthreads) {
  return adjoint_discrete(d, W, z_out, theta, ts, T, B, dz_out, rec_t, rec_dt, rec_n, rec_cap, dz0, dtheta, dW, stats, margins, nthreads);
}

/* RHS and VJP of a single column, exported for unit tests of the RHS menu. */
/* ---- ForwardDiffSensitivity as the reference EXECUTES it: the solve on dual numbers ---------------------------------------------------
 * `Pendulum()` carries sensealg = ForwardDiffSensitivity() [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into solve() at
 * [REF src/models/GOKU.jl:107, :121]. SciMLSensitivity 7.10.0 [REF Manifest.toml:1200] then seeds (u0, p) with D + P dual partials and runs
 * the SAME OrdinaryDiffEq solve on Dual numbers: every state component carries its value and its D + P partials through the
 * stages, the solution weights, the FSAL slope and the saveat interpolant; the pullback is Σ_j J_jᵀ Δ_j with J_j = ∂ẑ(t_j)/∂(u0, p) read off
 * the partials. Two things follow from running the solver itself on duals (DiffEqBase 6.104.3 [REF Manifest.toml:292]):
 *   - the step sizes are plain Float64s: the derivative is that of the DISCRETE solve on its accepted steps (what LDE_SENSE_DISCRETE
 *     computes in reverse mode: oracle_adjoint_discrete);
 *   - the error norm sees the partials: ODE_DEFAULT_NORM(u::Dual) = sqrt(value² + Σ partials²), so a component's scale is
 *     abstol + reltol·max(‖u_i‖, ‖u_new,i‖) with these dual norms, its scaled error the dual norm of ũ_i over that scale, and EEst the RMS over
 *     the n components. The accepted step sequence of a TRAINING solve therefore differs from that of the same solve run without AD
 *     (dual_norm = 1 here; dual_norm = 0 controls the steps by the values alone: the primal sequence, on which this routine's Jacobians
 *     equal oracle_adjoint_discrete's pullback to round-off — tests/test_oracle_dual.py).
 * Analytic right-hand sides, per-trajectory control (the GOKU path): state (x, v), partials with respect to (x0, v0, L).
 * Also bench.py's cpu_baseline for the GOKU workloads: this IS the algorithm the reference's CPU path runs (kind "port"). */
#define DN 2
#define DPAR 3
typedef struct { real v[DN]; real p[DN][DPAR]; } dstate;

static real (*volatile dual_sin)(real) = r_sin;   /* (through a pointer: the compiler would merge sin and cos into ONE sincos call, whose sine is not the bits of sin) */
static void dual_rhs(int kind, real L, const dstate* y, dstate* dy) {
  const real ngl = -(real)10 / L, s = dual_sin(y->v[0]), c = r_cos(y->v[0]);   /* (rhs_col's arithmetic on the values: the same bits) */
  const int fr = kind == LDE_RHS_PENDULUM_FRICTION;
  dy->v[0] = y->v[1];
  real acc = ngl * s;
  if (fr) acc -= ((real)0.7 / (real)1) * y->v[1];
  dy->v[1] = acc;
  for (int q = 0; q < DPAR; q++) {
    dy->p[0][q] = y->p[1][q];
    real a = ngl * c * y->p[0][q];
    if (fr) a -= (real)0.7 * y->p[1][q];
    dy->p[1][q] = a;
  }
  dy->p[1][2] += ((real)10 / (L * L)) * s;   /* ∂f/∂L = (G/L²) sin x */
}
static void dual_axpy(dstate* out, const dstate* y, real h, const dstate* acc) {
  for (int i = 0; i < DN; i++) {
    out->v[i] = y->v[i] + h * acc->v[i];
    for (int q = 0; q < DPAR; q++) out->p[i][q] = y->p[i][q] + h * acc->p[i][q];
  }
}
static void dual_lincomb(dstate* acc, const double* coef, const dstate* k, int n) {   /* acc = Σ_{j<n} coef[j] k[j], terms in the order j = 0, 1, … */
  for (int i = 0; i < DN; i++) {
    real a = (real)coef[0] * k[0].v[i];
    for (int j = 1; j < n; j++) a += (real)coef[j] * k[j].v[i];
    acc->v[i] = a;
    for (int q = 0; q < DPAR; q++) {
      real b = (real)coef[0] * k[0].p[i][q];
      for (int j = 1; j < n; j++) b += (real)coef[j] * k[j].p[i][q];
      acc->p[i][q] = b;
    }
  }
}
static real dual_abs(const dstate* y, int i, int dual_norm) {   /* ODE_DEFAULT_NORM of one (dual) component */
  if (!dual_norm) return r_fabs(y->v[i]);
  real s2 = y->v[i] * y->v[i];
  for (int q = 0; q < DPAR; q++) s2 += y->p[i][q] * y->p[i][q];
  return r_sqrt(s2);
}
static int dual_finite(const dstate* y) {
  for (int i = 0; i < DN; i++) {
    if (!isfinite((double)y->v[i])) return 0;
    for (int q = 0; q < DPAR; q++) if (!isfinite((double)y->p[i][q])) return 0;
  }
  return 1;
}
/* RMS over the n components of (‖e_i‖ / sk_i), sk_i = abstol + reltol·sc_i */
static double dual_rms(const dstate* e, const real* sk, int dual_norm) {
  double s2 = 0;
  for (int i = 0; i < DN; i++) {
    real r = dual_abs(e, i, dual_norm) / sk[i];
    s2 += (double)(r * r);
  }
  return sqrt(s2 / DN);
}

/* one trajectory: z [T][2], J [T][2][3]; returns the retcode */
static int dual_solve(int kind, real L, const real* z0, const double* ts, int T, const sopts* o, int dual_norm, real* z, real* J,
                      sstat* st, steprec* rec) {
  dstate y, yn, tmp, acc, k[7];
  memset(&y, 0, sizeof(y));
  y.v[0] = z0[0]; y.v[1] = z0[1];
  y.p[0][0] = 1; y.p[1][1] = 1;      /* seeds: ∂y0/∂x0, ∂y0/∂v0; ∂y0/∂L = 0 */
  int ret = LDE_RET_SUCCESS, ntr = 0;
#define DUAL_SAVE(j, ys) do { for (int i_ = 0; i_ < DN; i_++) { z[(j) * DN + i_] = (ys).v[i_]; for (int q_ = 0; q_ < DPAR; q_++) J[((j) * DN + i_) * DPAR + q_] = (ys).p[i_][q_]; } } while (0)
  DUAL_SAVE(0, y);
  if (T > 1) {
    double t = ts[0], tend = ts[T - 1], dtmax = tend - t, dt;
    const real at = (real)o->abstol, rt = (real)o->reltol;
    int j = 1;
    dual_rhs(kind, L, &y, &k[0]);
    st->nfe++;
    if (!o->adaptive) dt = o->dt_fixed;
    else if (o->dt_fixed > 0) dt = fmin(o->dt_fixed, dtmax);
    else {   /* Hairer–Nørsett–Wanner on duals: every norm the dual norm (as init_dt above on values) */
      real sk[DN];
      for (int i = 0; i < DN; i++) sk[i] = at + dual_abs(&y, i, dual_norm) * rt;
      const double d0 = dual_rms(&y, sk, dual_norm), d1 = dual_rms(&k[0], sk, dual_norm);
      double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
      if (dt0 > dtmax) dt0 = dtmax;
      dual_axpy(&tmp, &y, (real)dt0, &k[0]);
      dual_rhs(kind, L, &tmp, &k[1]);
      st->nfe++;
      for (int i = 0; i < DN; i++) {
        acc.v[i] = k[1].v[i] - k[0].v[i];
        for (int q = 0; q < DPAR; q++) acc.p[i][q] = k[1].p[i][q] - k[0].p[i][q];
      }
      const double d2 = dual_rms(&acc, sk, dual_norm) / dt0, dm = d1 > d2 ? d1 : d2;
      const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / 5.0);
      dt = fmin(100.0 * dt0, dt1);
      if (dt > dtmax) dt = dtmax;
    }
    real qold = (real)1e-4;
    int64_t iters = 0;
    const int presc = rec && rec->presc;
    const int adaptive = o->adaptive && !presc;
    while (t < tend) {
      if (iters++ >= o->maxiters) { ret = LDE_RET_MAXITERS; break; }
      double dtp = dt;
      int last = 0;
      if (presc) {
        if (ntr >= *rec->n) { ret = LDE_RET_MAXITERS; break; }
        t = rec->t[ntr]; dt = rec->dt[ntr]; last = ntr == *rec->n - 1;
      } else if (t + dt >= tend - 1e-12 * fabs(tend)) { dt = tend - t; last = 1; }
      const real h = (real)dt;
      const int S = o->solver == LDE_SOLVER_TSIT5 ? 6 : 4;   /* index of the FSAL slope */
      if (o->solver == LDE_SOLVER_TSIT5) {
        for (int s = 1; s < 6; s++) {
          dual_lincomb(&acc, TS_A[s], k, s);
          dual_axpy(&tmp, &y, h, &acc);
          dual_rhs(kind, L, &tmp, &k[s]);
        }
        dual_lincomb(&acc, TS_A[6], k, 6);
        dual_axpy(&yn, &y, h, &acc);
        dual_rhs(kind, L, &yn, &k[6]);
        st->nfe += 6;
      } else {
        const real hh = (real)(0.5 * dt);   /* (rk4_step's own arithmetic: y + (h/2)·k) */
        dual_axpy(&tmp, &y, hh, &k[0]); dual_rhs(kind, L, &tmp, &k[1]);
        dual_axpy(&tmp, &y, hh, &k[1]); dual_rhs(kind, L, &tmp, &k[2]);
        dual_axpy(&tmp, &y, h, &k[2]); dual_rhs(kind, L, &tmp, &k[3]);
        for (int i = 0; i < DN; i++) {   /* the oracle's own order: y + (h/6)·(k1 + 2(k2 + k3) + k4) */
          const real h6 = (real)(dt / 6.0);
          yn.v[i] = y.v[i] + h6 * (k[0].v[i] + (real)2 * (k[1].v[i] + k[2].v[i]) + k[3].v[i]);
          for (int q = 0; q < DPAR; q++) yn.p[i][q] = y.p[i][q] + h6 * (k[0].p[i][q] + (real)2 * (k[1].p[i][q] + k[2].p[i][q]) + k[3].p[i][q]);
        }
        dual_rhs(kind, L, &yn, &k[4]);
        st->nfe += 4;
      }
      real EEst = 0;
      if (o->adaptive && o->solver == LDE_SOLVER_TSIT5) {
        dual_lincomb(&acc, TS_BT, k, 7);
        real sk[DN];
        for (int i = 0; i < DN; i++) {
          sk[i] = at + r_fmax(dual_abs(&y, i, dual_norm), dual_abs(&yn, i, dual_norm)) * rt;
          acc.v[i] *= h;
          for (int q = 0; q < DPAR; q++) acc.p[i][q] *= h;
        }
        EEst = (real)dual_rms(&acc, sk, dual_norm);
      }
      if (!dual_finite(&yn) || !(EEst == EEst)) {
        if (adaptive && dt > o->dtmin) { st->nrej++; dt = dt * o->qmin; continue; }
        ret = LDE_RET_NONFINITE;
        break;
      }
      if (adaptive) {
        real q11, q = pi_q(EEst, qold, o, &q11);
        if (EEst > (real)1) {
          st->nrej++;
          dt = dt / (double)r_fmin((real)(1.0 / o->qmin), q11 / (real)o->gamma);
          if (dt < o->dtmin) { ret = LDE_RET_DTMIN; break; }
          continue;
        }
        qold = r_fmax(EEst, (real)1e-4);
        dtp = dt / (double)q;
        if (dtp > dtmax) dtp = dtmax;
      }
      st->nacc++;
      if (rec && !presc && ntr < rec->cap) { rec->t[ntr] = t; rec->dt[ntr] = dt; }
      ntr++;
      const double tnew = last ? tend : (presc ? rec->t[ntr] : t + dt);
      while (j < T && ts[j] <= tnew) {
        const double th = (ts[j] - t) / dt;
        if (th >= 1.0 || (j == T - 1 && last)) DUAL_SAVE(j, yn);
        else if (o->solver == LDE_SOLVER_TSIT5) {
          double bw[7];
          tsit5_interp_weights(th, bw);
          dual_lincomb(&acc, bw, k, 7);
          dual_axpy(&tmp, &y, h, &acc);
          DUAL_SAVE(j, tmp);
        } else {   /* cubic Hermite between (y, k1) and (y_new, f(y_new)) */
          const double h00 = (1 + 2 * th) * (1 - th) * (1 - th), h10 = th * (1 - th) * (1 - th), h01 = th * th * (3 - 2 * th), h11 = th * th * (th - 1);
          for (int i = 0; i < DN; i++) {
            tmp.v[i] = (real)h00 * y.v[i] + (real)(h10 * dt) * k[0].v[i] + (real)h01 * yn.v[i] + (real)(h11 * dt) * k[4].v[i];
            for (int q = 0; q < DPAR; q++)
              tmp.p[i][q] = (real)h00 * y.p[i][q] + (real)(h10 * dt) * k[0].p[i][q] + (real)h01 * yn.p[i][q] + (real)(h11 * dt) * k[4].p[i][q];
          }
          DUAL_SAVE(j, tmp);
        }
        j++;
      }
      y = yn;
      k[0] = k[S];
      t = tnew;
      dt = o->adaptive ? dtp : o->dt_fixed;
    }
    if (rec && !presc) *rec->n = ntr;
  }
#undef DUAL_SAVE
  st->retcode = ret;
  return ret;
}

/* z_out [T][B][2], J_out [T][B][2][3] (NULL: not kept); with dz_out [T][B][2] given: dz0 [B][2], dtheta [B][1] = Σ_j J_jᵀ Δ_j (the pullback the reference
 * forms from the stored partials). A failed trajectory: NaN block, zero gradient [REF src/models/GOKU.jl:114]. */
int oracle_forward_dual(const lde_problem_desc* d, const real* z0, const real* theta, const double* ts, int T, int B, int dual_norm,
                        real* z_out, real* J_out, const real* dz_out, real* dz0, real* dtheta, int32_t* retcode, int64_t* stats,
                        double* rec_t, double* rec_dt, int32_t* rec_n, int rec_cap, int presc, int nthreads) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (T < 1 || B < 1 || !z0 || !theta || !z_out) return LDE_ERR_INVALID_ARG;
  if (has_mlp(d) || d->batching != LDE_BATCH_PER_TRAJECTORY) return LDE_ERR_UNSUPPORTED;
  sopts o;
  opts_from_desc(d, &o, ts[0], ts[T - 1]);
  int64_t nfe = 0, nacc = 0, nrej = 0, nfail = 0, maxsteps = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : nfe, nacc, nrej, nfail) reduction(max : maxsteps)
  {
    real* z = (real*)calloc((size_t)T * DN, sizeof(real));
    real* J = (real*)calloc((size_t)T * DN * DPAR, sizeof(real));
#pragma omp for schedule(static)
    for (int c = 0; c < B; c++) {
      sstat st = {0, 0, 0, 0};
      steprec rec = {rec_t ? rec_t + (int64_t)c * rec_cap : NULL, rec_dt ? rec_dt + (int64_t)c * rec_cap : NULL, rec_n ? rec_n + c : NULL, rec_cap, presc};
      const int ret = dual_solve(d->rhs_kind, theta[c], z0 + (int64_t)c * DN, ts, T, &o, dual_norm, z, J, &st, rec_dt ? &rec : NULL);
      real g[DPAR] = {0, 0, 0};
      for (int j = 0; j < T; j++) {
        for (int i = 0; i < DN; i++) {
          const int64_t e = i + (int64_t)DN * (c + (int64_t)B * j);
          z_out[e] = ret ? (real)NAN : z[j * DN + i];
          for (int q = 0; q < DPAR; q++) {
            const real Jv = ret ? (real)0 : J[(j * DN + i) * DPAR + q];
            if (J_out) J_out[e * DPAR + q] = Jv;
            if (dz_out) g[q] += Jv * dz_out[e];
          }
        }
      }
      if (dz_out && dz0) { dz0[(int64_t)c * DN] = g[0]; dz0[(int64_t)c * DN + 1] = g[1]; }
      if (dz_out && dtheta) dtheta[c] = g[2];
      if (retcode) retcode[c] = ret;
      nfe += st.nfe; nacc += st.nacc; nrej += st.nrej; nfail += ret ? 1 : 0;
      if (st.nacc + st.nrej > maxsteps) maxsteps = st.nacc + st.nrej;
    }
    free(z);
    free(J);
  }
  if (stats) { stats[0] = nfe; stats[1] = nacc; stats[2] = nrej; stats[3] = nfail; stats[4] = maxsteps; }
  return LDE_OK;
}
#undef DN
#undef DPAR

int oracle_rhs(const lde_problem_desc* d, const real* W, const real* z, const real* theta, real* out) {
  colrhs c;
  colrhs_init(&c, d, W);
  rhs_col(&c, z, theta, out);
  colrhs_free(&c);
  return 0;
}
int oracle_rhs_vjp(const lde_problem_desc* d, const real* W, const real* z, const real* theta, const real* lam, real* f,
                   real* vz, real* vth, real* dW) {
  colrhs c;
  colrhs_init(&c, d, W);
  if (c.dW_step) memset(c.dW_step, 0, (size_t)c.nW * sizeof(real));
  rhs_vjp_col(&c, z, theta, lam, f, vz, vth, (real)1);
  if (dW && c.dW_step) memcpy(dW, c.dW_step, (size_t)c.nW * sizeof(real));
  colrhs_free(&c);
  return 0;
}

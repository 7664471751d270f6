/*
 * lde_rnn_oracle.c — CPU restatement of the recurrent pattern extractor (scope row f-2).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * reported baseline. The product (liblde.so) never links, loads or calls anything in here.
 *
 * PARITY UNPINNED against reference-produced vectors (Julia / Flux 0.13.6 cannot run in this image and the reference's
 * test suite holds no vectors [REF test/runtests.jl:4-6]). Pinned instead against an independent implementation of the
 * same published cell definitions: torch autograd on CPU, float64 (tests/test_oracle_rnn.py).
 *
 * What it follows
 *   apply_pattern_extractor: `[pe(x) for x in fe_out_rev][end]`, `Flux.reset!`      [REF src/models/GOKU.jl:32-51]
 *   stacks: Chain(RNN(32,16,relu), RNN(16,16,relu)); Chain(LSTM(32,16), LSTM(16,16)) ×2  [REF src/models/GOKU.jl:229-238]
 *   Flux RNNCell / LSTMCell equations and parameter order (Wi, Wh, b, state0): see include/lde.h.
 * Memory: x is [in × B × T] column-major: frame t, trajectory b at x + in*(b + B*t).
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
#define r_tanh tanh
#else
typedef float real;
#define r_exp expf
#define r_tanh tanhf
#endif

static real sigm(real x) { return (real)1 / ((real)1 + r_exp(-x)); }

static int rnn_ok(const lde_rnn_desc* d) {
  if (!d || d->n_layers < 1 || d->n_layers > LDE_RNN_MAX_LAYERS || d->cell < 0 || d->cell > LDE_CELL_LSTM) return 0;
  for (int l = 0; l <= d->n_layers; l++)
    if (d->sizes[l] < 1) return 0;
  return 1;
}
static int gates(const lde_rnn_desc* d) { return d->cell == LDE_CELL_LSTM ? 4 : 1; }
static int nstate(const lde_rnn_desc* d) { return d->cell == LDE_CELL_LSTM ? 2 : 1; }

int64_t oracle_rnn_num_weights(const lde_rnn_desc* d) {
  if (!rnn_ok(d)) return -1;
  int64_t n = 0;
  for (int l = 0; l < d->n_layers; l++) {
    const int64_t in = d->sizes[l], h = d->sizes[l + 1], G = gates(d);
    n += G * h * in + G * h * h + G * h + nstate(d) * h;
  }
  return n;
}

typedef struct { const real *Wi, *Wh, *b, *h0, *c0; int64_t off; int in, h; } cellw;

static void split(const lde_rnn_desc* d, const real* W, cellw* cw) {
  int64_t off = 0;
  for (int l = 0; l < d->n_layers; l++) {
    const int in = d->sizes[l], h = d->sizes[l + 1], G = gates(d);
    cw[l].off = off; cw[l].in = in; cw[l].h = h;
    cw[l].Wi = W + off; off += (int64_t)G * h * in;
    cw[l].Wh = W + off; off += (int64_t)G * h * h;
    cw[l].b = W + off; off += (int64_t)G * h;
    cw[l].h0 = W + off; off += h;
    cw[l].c0 = d->cell == LDE_CELL_LSTM ? W + off : NULL;
    if (d->cell == LDE_CELL_LSTM) off += h;
  }
}

/* per-trajectory record of one (step, layer): gate activations (G*h), c (h, LSTM), h (h) */
#define HMAXO 256
static void cell_step(const lde_rnn_desc* d, const cellw* c, const real* xin, const real* hprev, const real* cprev, real* act,
                      real* cnew, real* hnew) {
  const int in = c->in, h = c->h, G = gates(d), R = G * h;
  for (int r = 0; r < R; r++) {
    real s = c->b[r];
    for (int k = 0; k < in; k++) s += c->Wi[r + (size_t)R * k] * xin[k];
    for (int k = 0; k < h; k++) s += c->Wh[r + (size_t)R * k] * hprev[k];
    act[r] = s;
  }
  if (d->cell == LDE_CELL_LSTM) {
    for (int u = 0; u < h; u++) {
      const real ig = sigm(act[u]), fg = sigm(act[h + u]), gg = r_tanh(act[2 * h + u]), og = sigm(act[3 * h + u]);
      act[u] = ig; act[h + u] = fg; act[2 * h + u] = gg; act[3 * h + u] = og;
      cnew[u] = fg * cprev[u] + ig * gg;
      hnew[u] = og * r_tanh(cnew[u]);
    }
  } else {
    for (int u = 0; u < h; u++) {
      const real v = d->cell == LDE_CELL_RNN_TANH ? r_tanh(act[u]) : (act[u] > 0 ? act[u] : (real)0);
      act[u] = v;
      hnew[u] = v;
    }
  }
}

/* forward sweep of one trajectory; rec (optional) receives per (step s, layer l): [G*h act | h c | h h] at stride recw */
static void traj_forward(const lde_rnn_desc* d, const cellw* cw, const real* x, int T, int B, int b, real* rec, int recw,
                         real* ytop) {
  const int L = d->n_layers;
  real hs[LDE_RNN_MAX_LAYERS][HMAXO], cs[LDE_RNN_MAX_LAYERS][HMAXO], act[4 * HMAXO], hn[HMAXO], cn[HMAXO];
  for (int l = 0; l < L; l++)
    for (int u = 0; u < cw[l].h; u++) { hs[l][u] = cw[l].h0[u]; cs[l][u] = cw[l].c0 ? cw[l].c0[u] : (real)0; }
  for (int s = 0; s < T; s++) {
    const int t = d->reverse ? T - 1 - s : s;
    const real* xin = x + (size_t)d->sizes[0] * ((size_t)b + (size_t)B * t);
    for (int l = 0; l < L; l++) {
      const int h = cw[l].h, G = gates(d);
      cell_step(d, &cw[l], xin, hs[l], cs[l], act, cn, hn);
      if (rec) {
        real* r = rec + ((size_t)s * L + l) * recw;
        memcpy(r, act, sizeof(real) * (size_t)G * h);
        memcpy(r + G * h, cn, sizeof(real) * (size_t)h);
        memcpy(r + G * h + h, hn, sizeof(real) * (size_t)h);
      }
      memcpy(hs[l], hn, sizeof(real) * (size_t)h);
      if (d->cell == LDE_CELL_LSTM) memcpy(cs[l], cn, sizeof(real) * (size_t)h);
      xin = hs[l];
    }
  }
  memcpy(ytop, hs[L - 1], sizeof(real) * (size_t)cw[L - 1].h);
}

int oracle_rnn_forward(const lde_rnn_desc* d, const real* W, const real* x, int T, int B, real* y, int nthreads) {
  if (!rnn_ok(d) || !W || !x || !y || T < 1 || B < 1) return LDE_ERR_INVALID_ARG;
  for (int l = 1; l <= d->n_layers; l++)
    if (d->sizes[l] > HMAXO) return LDE_ERR_UNSUPPORTED;
  cellw cw[LDE_RNN_MAX_LAYERS];
  split(d, W, cw);
  const int hL = d->sizes[d->n_layers];
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; b++) traj_forward(d, cw, x, T, B, b, NULL, 0, y + (size_t)hL * b);
  return LDE_OK;
}

int oracle_rnn_backward(const lde_rnn_desc* d, const real* W, const real* x, const real* dy, int T, int B, real* dx, real* dW,
                        int nthreads) {
  if (!rnn_ok(d) || !W || !x || !dy || !dW || T < 1 || B < 1) return LDE_ERR_INVALID_ARG;
  for (int l = 1; l <= d->n_layers; l++)
    if (d->sizes[l] > HMAXO) return LDE_ERR_UNSUPPORTED;
  cellw cw[LDE_RNN_MAX_LAYERS];
  split(d, W, cw);
  const int L = d->n_layers, G = gates(d), lstm = d->cell == LDE_CELL_LSTM;
  const int64_t nW = oracle_rnn_num_weights(d);
  int hmax = 0;
  for (int l = 1; l <= L; l++)
    if (d->sizes[l] > hmax) hmax = d->sizes[l];
  const int recw = (G + 2) * hmax;
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  real* partial = (real*)calloc((size_t)nthreads * (size_t)nW, sizeof(real));
  if (!partial) return LDE_ERR_ALLOC;
#pragma omp parallel num_threads(nthreads)
  {
    real* gw = partial + (size_t)omp_get_thread_num() * (size_t)nW;
    real* rec = (real*)malloc(sizeof(real) * (size_t)T * L * recw);
    real ytop[HMAXO];
    real dh[LDE_RNN_MAX_LAYERS][HMAXO], dc[LDE_RNN_MAX_LAYERS][HMAXO], del[4 * HMAXO], din[HMAXO > 256 ? HMAXO : 256];
#pragma omp for schedule(static)
    for (int b = 0; b < B; b++) {
      traj_forward(d, cw, x, T, B, b, rec, recw, ytop);
      for (int l = 0; l < L; l++)
        for (int u = 0; u < cw[l].h; u++) { dh[l][u] = 0; dc[l][u] = 0; }
      for (int u = 0; u < cw[L - 1].h; u++) dh[L - 1][u] = dy[(size_t)cw[L - 1].h * b + u];
      for (int s = T - 1; s >= 0; s--) {
        const int t = d->reverse ? T - 1 - s : s;
        for (int l = L - 1; l >= 0; l--) {
          const int in = cw[l].in, h = cw[l].h, R = G * h;
          const real* r = rec + ((size_t)s * L + l) * recw;
          const real* act = r;
          const real* cnew = r + R;
          const real* hprev = s > 0 ? rec + ((size_t)(s - 1) * L + l) * recw + R + h : cw[l].h0;
          const real* cprev = lstm ? (s > 0 ? rec + ((size_t)(s - 1) * L + l) * recw + R : cw[l].c0) : NULL;
          const real* xin = l > 0 ? rec + ((size_t)s * L + (l - 1)) * recw + G * cw[l - 1].h + cw[l - 1].h
                                  : x + (size_t)in * ((size_t)b + (size_t)B * t);
          if (lstm) {
            for (int u = 0; u < h; u++) {
              const real ig = act[u], fg = act[h + u], gg = act[2 * h + u], og = act[3 * h + u];
              const real tc = r_tanh(cnew[u]);
              const real dct = dc[l][u] + dh[l][u] * og * ((real)1 - tc * tc);
              del[u] = dct * gg * ig * ((real)1 - ig);
              del[h + u] = dct * cprev[u] * fg * ((real)1 - fg);
              del[2 * h + u] = dct * ig * ((real)1 - gg * gg);
              del[3 * h + u] = dh[l][u] * tc * og * ((real)1 - og);
              dc[l][u] = dct * fg;
            }
          } else {
            for (int u = 0; u < h; u++) {
              const real a = act[u];
              del[u] = dh[l][u] * (d->cell == LDE_CELL_RNN_TANH ? (real)1 - a * a : (a > 0 ? (real)1 : (real)0));
            }
          }
          real* gWi = gw + cw[l].off;
          real* gWh = gWi + (size_t)R * in;
          real* gb = gWh + (size_t)R * h;
          for (int k = 0; k < in; k++) din[k] = 0;
          for (int u = 0; u < h; u++) dh[l][u] = 0;
          for (int rr = 0; rr < R; rr++) {
            const real dl = del[rr];
            gb[rr] += dl;
            for (int k = 0; k < in; k++) {
              gWi[rr + (size_t)R * k] += dl * xin[k];
              din[k] += cw[l].Wi[rr + (size_t)R * k] * dl;
            }
            for (int k = 0; k < h; k++) {
              gWh[rr + (size_t)R * k] += dl * hprev[k];
              dh[l][k] += cw[l].Wh[rr + (size_t)R * k] * dl;
            }
          }
          if (l > 0) {
            for (int k = 0; k < in; k++) dh[l - 1][k] += din[k];
          } else if (dx) {
            for (int k = 0; k < in; k++) dx[(size_t)in * ((size_t)b + (size_t)B * t) + k] = din[k];
          }
        }
      }
      for (int l = 0; l < L; l++) {   /* what is left flows into the trainable initial state */
        const int in = cw[l].in, h = cw[l].h, R = G * h;
        real* g0 = gw + cw[l].off + (size_t)R * in + (size_t)R * h + R;
        for (int u = 0; u < h; u++) g0[u] += dh[l][u];
        if (lstm)
          for (int u = 0; u < h; u++) g0[h + u] += dc[l][u];
      }
    }
    free(rec);
  }
  for (int t = 0; t < nthreads; t++)
    for (int64_t i = 0; i < nW; i++) dW[i] += partial[(size_t)t * (size_t)nW + i];
  free(partial);
  return LDE_OK;
}

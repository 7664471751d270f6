// lde_host.h — the host-side LOGIC of the C ABI that touches no device: validation of a problem description, the flat weight count,
// the layout arithmetic of a step record, the option block handed to the kernels, the step count of a fixed-step solve, the checks on a
// save-time grid. lde_api.hip is these functions plus HIP calls; kept apart so that an ordinary host compiler can build them under
// AddressSanitizer + UndefinedBehaviorSanitizer (tests/host_logic_driver.cpp, tests/test_sanitizers.py: GPU sanitizers are not available
// on this pool) and drive them with hostile inputs: a C ABI's arguments come from another language's runtime.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <string>

#include "lde_types.h"

namespace lde_host {
using lde::KOpts;
using lde::StepRec;

static bool has_mlp(const lde_problem_desc& d) {
  return d.rhs_kind == LDE_RHS_MLP || d.rhs_kind == LDE_RHS_PENDULUM_PLUS_MLP;
}
static bool has_pend(const lde_problem_desc& d) { return d.rhs_kind != LDE_RHS_MLP; }

static int validate(const lde_problem_desc* d, std::string* why) {
  auto bad = [&](const char* m) {
    if (why) *why = m;
    return (int)LDE_ERR_INVALID_ARG;
  };
  if (!d) return bad("desc is NULL");
  if (d->abi_version != LDE_ABI_VERSION) return bad("abi_version mismatch");
  if (d->rhs_kind < 0 || d->rhs_kind > LDE_RHS_PENDULUM_PLUS_MLP) return bad("unknown rhs_kind");
  if (d->state_dim < 1 || d->param_dim < 0 || d->augment_dim < 0) return bad("bad dims");
  if (has_pend(*d) && (d->state_dim != 2 || d->param_dim != 1 || d->augment_dim != 0))
    return bad("pendulum RHS needs state_dim=2, param_dim=1, augment_dim=0");
  if (d->rhs_kind == LDE_RHS_MLP && d->param_dim != 0) return bad("MLP RHS takes no per-trajectory parameters");
  if (has_mlp(*d)) {
    if (d->n_layers < 1 || d->n_layers > LDE_MAX_LAYERS) return bad("n_layers out of range");
    const int Dp = d->state_dim + d->augment_dim;
    if (d->layer_sizes[0] != Dp || d->layer_sizes[d->n_layers] != Dp) return bad("MLP in/out must equal D+augment_dim");
    for (int l = 0; l <= d->n_layers; l++)
      if (d->layer_sizes[l] < 1) return bad("layer size < 1");
    if (d->activation != LDE_ACT_RELU && d->activation != LDE_ACT_TANH) return bad("unknown activation");
  }
  if (d->solver != LDE_SOLVER_TSIT5 && d->solver != LDE_SOLVER_RK4) return bad("unknown solver");
  if (d->batching != LDE_BATCH_PER_TRAJECTORY && d->batching != LDE_BATCH_COUPLED && d->batching != LDE_BATCH_COUPLED_GLOBAL)
    return bad("unknown batching");
  if (d->batching == LDE_BATCH_COUPLED_GLOBAL && !has_mlp(*d)) return bad("LDE_BATCH_COUPLED_GLOBAL needs an MLP right-hand side");
  if (d->sensealg < LDE_SENSE_BACKSOLVE_CHECKPOINTED || d->sensealg > LDE_SENSE_DISCRETE) return bad("unknown sensealg");
  // (LDE_SENSE_DISCRETE with LDE_BATCH_COUPLED_GLOBAL: every rank records the common step sequence and ITS columns' states; the sweep
  //  has no step control, hence no sum to exchange)

  if (d->solver == LDE_SOLVER_RK4 && d->adaptive) {
    if (why) *why = "RK4 is fixed-step only here: pass adaptive=0, dt=h";
    return LDE_ERR_UNSUPPORTED;
  }
  if (!d->adaptive && !(d->dt > 0)) return bad("adaptive=0 needs dt>0");
  if (d->adaptive && (!(d->abstol > 0) || !(d->reltol > 0))) return bad("tolerances must be > 0");
  if (d->maxiters < 1) return bad("maxiters < 1");
  if (!(d->qmin > 0) || !(d->qmax > 0) || !(d->gamma > 0)) return bad("controller constants must be > 0");
  return LDE_OK;
}


static int rec_nseq(const lde_problem_desc& d, int B) { return d.batching == LDE_BATCH_PER_TRAJECTORY ? B : 1; }
// accepted steps a record holds per sequence: the "record_capacity" option, else max(64, 4T) (forward) / max(256, 16T) (reverse-time trace), never more than maxiters
static int rec_capacity(const lde_problem_desc& d, int opt_record_capacity, int T, int which) {
  if (opt_record_capacity > 0) return opt_record_capacity;
  const int64_t c = which == 0 ? std::max<int64_t>(64, 4 * (int64_t)T) : std::max<int64_t>(256, 16 * (int64_t)T);
  return (int)std::min<int64_t>(c, std::max<int64_t>(1, d.maxiters));
}
static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
// layout: n [nseq] | t [cap][nseq] | dt [cap][nseq] | y [cap][B][D'] (forward records only)
static size_t rec_bytes(const lde_problem_desc& d, int B, int cap, bool with_y) {
  const size_t nseq = (size_t)rec_nseq(d, B), Dp = (size_t)(d.state_dim + d.augment_dim);
  return align256(nseq * 4) + 2 * align256((size_t)cap * nseq * 8) + (with_y ? align256((size_t)cap * B * Dp * 4) : 0);
}
static StepRec rec_view(const lde_problem_desc& d, void* base, int B, int cap, bool with_y) {
  StepRec r{};
  const size_t nseq = (size_t)rec_nseq(d, B);
  unsigned char* p = (unsigned char*)base;
  r.n = (int32_t*)p; p += align256(nseq * 4);
  r.t = (double*)p; p += align256((size_t)cap * nseq * 8);
  r.dt = (double*)p; p += align256((size_t)cap * nseq * 8);
  r.y = with_y ? (float*)p : nullptr;
  r.cap = cap;
  r.nseq = (int)nseq;
  return r;
}

// Number of floats in the flat weight vector implied by desc (0 for analytic right-hand sides; n_layers clamped to the struct's capacity)
static int64_t num_weights(const lde_problem_desc* d) {
  if (!d || !has_mlp(*d)) return 0;
  int64_t n = 0;
  for (int l = 0; l < d->n_layers && l < LDE_MAX_LAYERS; l++)
    n += (int64_t)d->layer_sizes[l + 1] * d->layer_sizes[l] + d->layer_sizes[l + 1];
  return n;
}

// ts must be finite and strictly increasing
static bool grid_ok(const double* ts, int T) {
  for (int j = 0; j < T; j++)
    if (!std::isfinite(ts[j]) || (j && !(ts[j] > ts[j - 1]))) return false;
  return true;
}

// a fixed-step solve's number of step attempts over the grid (0: adaptive — unknown here)
static int64_t fixed_step_count(const lde_problem_desc& d, const double* ts, int T) {
  if (d.adaptive || !(d.dt > 0)) return 0;
  int64_t steps = 0;
  for (int j = 0; j + 1 < T; j++) {
    const double n = std::ceil((ts[j + 1] - ts[j]) / d.dt * (1.0 - 1e-12));
    steps += n < 1 ? 1 : (n > 1e9 ? (int64_t)1e9 : (int64_t)n);
    if (steps > d.maxiters) return d.maxiters;
  }
  return steps > d.maxiters ? d.maxiters : steps;
}

// Which forward mapping of the analytic right-hand sides serves a solve (csrc/lde_pendulum.hip's launch code switches on this; DESIGN.md §4.1;
// the thresholds are measurements: abl/lp_midB.py, abl/pend_B.py, abl/pend_LB.py). `ts_lds_max`: the longest save grid the kernels stage in LDS.
enum PendFwdMap {
  PEND_FWD_LP4 = 0,   // k_pend_forward_lp<REC, 4>: a trajectory per workgroup, lane pairs / Nyström form, four dense-output waves
  PEND_FWD_LP3,       // … three (more than two workgroups per CU)
  PEND_FWD_SH,        // k_pend_forward_sh: a trajectory per workgroup, every other solve (friction, RK4, fixed steps; option "pend_lp" = 0)
  PEND_FWD_TL,        // k_pend_forward_tl<…, 1>: lanes = save times (writes no step record)
  PEND_FWD_WS,        // k_pend_forward_ws: a stepping wave + dense-output waves per 64 trajectories
  PEND_FWD_RING,      // k_pend_forward_tl<…, 64, RING>: a lane per trajectory, ẑ rows through an LDS ring (large batches)
  PEND_FWD_LANE       // k_pend_forward: a lane per trajectory, direct stores
};
static PendFwdMap pend_forward_mapping(int kind, int solver, bool adaptive, bool recording, int B, int T, const lde::PendTune& tn, int ts_lds_max) {
  const bool lp_shape = kind == LDE_RHS_PENDULUM && solver == LDE_SOLVER_TSIT5 && adaptive && tn.lp;
  // option "pend_sh_max_b" ≥ 0: ONE threshold for both mappings with a trajectory per workgroup (what the tests force a mapping with); −1: the measured ones
  const int sh_max_b = tn.sh_max_b >= 0 ? tn.sh_max_b : lp_shape ? 1024 : recording ? 768 : 256;
  if (T > 1 && B <= sh_max_b) return lp_shape ? (B <= 512 ? PEND_FWD_LP4 : PEND_FWD_LP3) : PEND_FWD_SH;
  if (!recording && T > 1 && B <= tn.tl_max_b) return PEND_FWD_TL;
  if (tn.ws != 0 && T <= ts_lds_max && T > 2 && B <= 16384) return PEND_FWD_WS;   // (21.6 against 31.2 µs at 16 384, 36.8 against 32.6 at 32 768)
  if (tn.lb_ring > 0 && T > 1 && T <= 2048 && B >= tn.lb_min_b) return PEND_FWD_RING;
  return PEND_FWD_LANE;
}

static KOpts make_opts(const lde_problem_desc& d, const double* ts, int T, int B) {
  KOpts o;
  o.abstol = (float)d.abstol;
  o.reltol = (float)d.reltol;
  o.beta1 = (float)d.beta1;
  o.beta2 = (float)d.beta2;
  o.inv_gamma = (float)(1.0 / d.gamma);
  o.q_lo = (float)(1.0 / d.qmax);
  o.q_hi = (float)(1.0 / d.qmin);
  o.qmin = (float)d.qmin;
  o.dtmin = d.dtmin > 0 ? d.dtmin : 1e-12 * std::fabs(ts[T - 1] - ts[0]);
  o.dt_fixed = d.dt;
  o.maxiters = d.maxiters;
  o.adaptive = d.adaptive;
  o.checkpoint = d.sensealg != LDE_SENSE_BACKSOLVE;
  o.T = T;
  o.B = B;
  o.lb_hold = 0;
  o.dw_overwrite = 0;
  o.t_first = ts[0];
  o.t_last = ts[T - 1];
  o.rec = StepRec{};
  return o;
}

}  // namespace lde_host

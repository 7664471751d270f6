// lde_api.hip — host side of the C ABI declared in include/lde.h.
//
// Owns what the reference rebuilds on every call (remake / EnsembleProblem / NeuralODE +
// Flux.destructure [REF src/models/GOKU.jl:111-118], [REF src/models/LatentODE.jl:70-71]):
// the validated problem description, the device copy of the RHS-MLP weights, the device copy of the
// save-time grid and the per-trajectory statistics workspace. No torch types, no allocation on the
// hot path once lde_reserve() (or a first call of the same size) has run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lde_device.h"
#include "lde_host.h"

namespace lde {
int launch_pend_forward(int kind, int solver, const float* z0, const float* theta, const double* ts_dev, const KOpts& o,
                        float* z_out, int32_t* retcode, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                        hipStream_t stream, const PendTune& tn);
int launch_pend_adjoint(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev,
                        const KOpts& o, const float* dz_out, float* dz0, float* dtheta, int32_t* nfe, int32_t* nacc,
                        int32_t* nrej, int32_t* ret, hipStream_t stream);
int launch_pend_adjoint_par(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev,
                            const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* ops, int32_t* info,
                            int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret, hipStream_t stream);
bool pend_adjoint_needs_ops(int B, int T);
const char* pend_last_kernel(int which);
int launch_pend_adjoint_disc(int kind, int solver, const float* z_out, const float* theta, const double* ts_dev, const KOpts& o,
                             const float* dz_out, float* dz0, float* dtheta, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                             hipStream_t stream, const PendTune& tn);
struct MlpPlan;
int mlp_plan_create(const lde_problem_desc& d, MlpPlan** out, std::string& err);
void mlp_plan_destroy(MlpPlan* p);
int mlp_reserve(MlpPlan* p, int B, int T, std::string& err);
int mlp_set_sum_hook(MlpPlan* p, lde_sum_hook hook, void* user, int64_t global_batch, std::string& err);
size_t mlp_sum_mailbox_bytes(int nranks);
int mlp_set_sum_peers(MlpPlan* p, int rank, int nranks, void* const* boxes, int64_t global_batch, std::string& err);
int mlp_set_phase_timing(MlpPlan* p, int on);
int mlp_last_family(const MlpPlan* p);
MlpTune* mlp_tune(MlpPlan* p);
int mlp_get_phase_ms(MlpPlan* p, float* out);
int mlp_reserve_adjoint(MlpPlan* p, int B, int T, int64_t steps_hint, std::string& err);
int mlp_set_weights(MlpPlan* p, const float* W_dev, hipStream_t stream, std::string& err);
int mlp_forward(MlpPlan* p, const float* W_dev, const float* z0, const float* theta, const double* ts_dev,
                const KOpts& o, float* z_out, int32_t* retcode, int32_t* nfe, int32_t* nacc, int32_t* nrej, int32_t* ret,
                hipStream_t stream, std::string& err);
int mlp_adjoint(MlpPlan* p, const float* W_dev, const float* z_out, const float* theta, const double* ts_dev,
                const KOpts& o, const float* dz_out, float* dz0, float* dtheta, float* dW, int32_t* nfe, int32_t* nacc,
                int32_t* nrej, int32_t* ret, hipStream_t stream, std::string& err);
}  // namespace lde

static constexpr int TS_RING = 8;

struct lde_handle {
  lde_problem_desc d;
  int device = 0;
  int64_t nW = 0;
  float* W_dev = nullptr;
  bool have_W = false;
  lde::MlpPlan* mlp = nullptr;
  // save-time grid cache
  std::vector<double> ts_host;
  double* ts_dev = nullptr;
  int ts_cap = 0;
  double* ts_pinned[TS_RING] = {};
  hipEvent_t ts_ev[TS_RING] = {};
  int ts_pin_cap = 0;
  int ring = 0;
  // per-trajectory statistics, [0]=forward, [1]=adjoint: nfe, nacc, nrej, ret
  int32_t* st[2][4] = {};
  int cap_B = 0;
  int last_B[2] = {0, 0};
  // parallel-in-time adjoint: per-(interval, trajectory) transition operators
  float* par_ops = nullptr;
  int32_t* par_info = nullptr;
  size_t par_cap = 0;
  // step records (include/lde.h: lde_set_step_record): [0] the forward solve's (LDE_SENSE_DISCRETE / step tracing), [1] the continuous
  // adjoint's reverse-time steps (step tracing). own: the handle's buffer; user: the caller's (forward record only)
  void* rec_own[2] = {nullptr, nullptr};
  size_t rec_own_bytes[2] = {0, 0};
  void* rec_user = nullptr;
  size_t rec_user_bytes = 0;
  lde::StepRec rec_last[2] = {};     // the views the last forward / adjoint were given (lde_get_step_record reads them back)
  int rec_B = 0, rec_T = 0;          // shape of the forward that wrote rec_last[0]
  int opt_record_capacity = 0;       // 0: automatic
  int opt_step_trace = 0;
  int opt_adjoint_overwrite = 0;
  lde::PendTune pend_tune;           // kernel-choice knobs of the analytic right-hand sides (MLP ones live in the plan)
  const char* last_kernel[2] = {"", ""};   // lde_last_kernel: the solve kernel the last lde_forward / lde_adjoint launched
  std::string err = "";
};

using namespace lde_host;   // validate, has_mlp, has_pend, rec_nseq, rec_capacity, rec_bytes, rec_view, align256, make_opts, num_weights, grid_ok, fixed_step_count
static int rec_capacity(const lde_handle* h, int T, int which);

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return LDE_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

// the weight-gradient stream of the chain / recurrent pullbacks (lde_mfma.h: dw_stream_get); one per process
static std::atomic<hipStream_t> g_dw_stream{nullptr};
static hipEvent_t g_dw_join = nullptr;
namespace lde {
hipStream_t dw_stream_get() { return g_dw_stream.load(std::memory_order_acquire); }
}

extern "C" {

int lde_abi_version(void) { return LDE_ABI_VERSION; }

int lde_set_dw_stream(void* stream) {
  g_dw_stream.store((hipStream_t)stream, std::memory_order_release);
  return LDE_OK;
}

int lde_join_dw(void* stream) {
  hipStream_t dws = g_dw_stream.load(std::memory_order_acquire);
#if LDE_DW_DEBUG
  {
    hipStreamCaptureStatus ca = hipStreamCaptureStatusNone, cb = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing((hipStream_t)stream, &ca);
    if (dws) (void)hipStreamIsCapturing(dws, &cb);
    fprintf(stderr, "[dw join] stream=%p(cap %d) dws=%p(cap %d)\n", stream, (int)ca, (void*)dws, (int)cb);
  }
#endif
  if (!dws || dws == (hipStream_t)stream) return LDE_OK;
  if (!g_dw_join && hipEventCreateWithFlags(&g_dw_join, hipEventDisableTiming) != hipSuccess) return LDE_ERR_HIP;
  if (hipEventRecord(g_dw_join, dws) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, g_dw_join, 0) != hipSuccess) return LDE_ERR_HIP;
  return LDE_OK;
}

int lde_problem_desc_default(lde_problem_desc* d) {
  if (!d) return LDE_ERR_INVALID_ARG;
  std::memset(d, 0, sizeof(*d));
  d->abi_version = LDE_ABI_VERSION;
  d->rhs_kind = LDE_RHS_PENDULUM;
  d->state_dim = 2;
  d->param_dim = 1;
  d->solver = LDE_SOLVER_TSIT5;
  d->batching = LDE_BATCH_PER_TRAJECTORY;
  // `Pendulum()` carries ForwardDiffSensitivity() [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into solve() at
  // [REF src/models/GOKU.jl:107, :121]: the exact derivative of the discrete solve — LDE_SENSE_DISCRETE. (A NODE binding sets
  // LDE_SENSE_BACKSOLVE_CHECKPOINTED itself: DiffEqFlux's InterpolatingAdjoint [REF src/models/LatentODE.jl:67-70].)
  d->sensealg = LDE_SENSE_DISCRETE;
  d->activation = LDE_ACT_RELU;
  d->adaptive = 1;
  d->maxiters = 100000;
  d->abstol = 1e-6;
  d->reltol = 1e-3;
  d->qmin = 0.2;
  d->qmax = 10.0;
  d->gamma = 0.9;
  d->beta1 = 7.0 / 50.0;
  d->beta2 = 2.0 / 25.0;
  return LDE_OK;
}

int64_t lde_num_weights(const lde_problem_desc* d) { return num_weights(d); }

int lde_create(const lde_problem_desc* desc, lde_handle** out) {
  if (!out) return LDE_ERR_INVALID_ARG;
  *out = nullptr;
  int rc = validate(desc, nullptr);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return LDE_ERR_NO_DEVICE;
  lde_handle* h = new (std::nothrow) lde_handle();
  if (!h) return LDE_ERR_ALLOC;
  h->d = *desc;
  // the time-parallel adjoint exists for analytic right-hand sides with per-trajectory control; elsewhere the same
  // checkpointed adjoint runs sequentially (documented in include/lde.h)
  if (h->d.sensealg == LDE_SENSE_PARALLEL_CHECKPOINTED && (has_mlp(h->d) || h->d.batching != LDE_BATCH_PER_TRAJECTORY))
    h->d.sensealg = LDE_SENSE_BACKSOLVE_CHECKPOINTED;
  if (hipGetDevice(&h->device) != hipSuccess) {
    delete h;
    return LDE_ERR_NO_DEVICE;
  }
  h->nW = lde_num_weights(desc);
  if (h->nW) {
    if (hipMalloc(&h->W_dev, (size_t)h->nW * sizeof(float)) != hipSuccess) {
      delete h;
      return LDE_ERR_ALLOC;
    }
    rc = lde::mlp_plan_create(h->d, &h->mlp, h->err);
    if (rc) {
      (void)hipFree(h->W_dev);
      delete h;
      return rc;
    }
  }
  for (int i = 0; i < TS_RING; i++)
    if (hipEventCreateWithFlags(&h->ts_ev[i], hipEventDisableTiming) != hipSuccess) {
      lde_destroy(h);
      return LDE_ERR_HIP;
    }
  *out = h;
  return LDE_OK;
}

void lde_destroy(lde_handle* h) {
  if (!h) return;
  if (h->mlp) lde::mlp_plan_destroy(h->mlp);
  if (h->W_dev) (void)hipFree(h->W_dev);
  if (h->ts_dev) (void)hipFree(h->ts_dev);
  if (h->par_ops) (void)hipFree(h->par_ops);
  if (h->par_info) (void)hipFree(h->par_info);
  for (int i = 0; i < 2; i++)
    if (h->rec_own[i]) (void)hipFree(h->rec_own[i]);
  for (int i = 0; i < TS_RING; i++) {
    if (h->ts_pinned[i]) (void)hipHostFree(h->ts_pinned[i]);
    if (h->ts_ev[i]) (void)hipEventDestroy(h->ts_ev[i]);
  }
  for (int w = 0; w < 2; w++)
    for (int i = 0; i < 4; i++)
      if (h->st[w][i]) (void)hipFree(h->st[w][i]);
  delete h;
}

int lde_set_weights(lde_handle* h, const float* flat_host, int64_t n) {
  if (!h) return LDE_ERR_INVALID_ARG;
  if (n != h->nW || (n && !flat_host)) {
    h->err = "lde_set_weights: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  if (!n) return LDE_OK;
  HIP_TRY(h, hipMemcpy(h->W_dev, flat_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  h->have_W = true;
  return lde::mlp_set_weights(h->mlp, h->W_dev, nullptr, h->err);
}

int lde_set_weights_device(lde_handle* h, const float* flat_dev, int64_t n, void* stream) {
  if (!h) return LDE_ERR_INVALID_ARG;
  if (n != h->nW || (n && !flat_dev)) {
    h->err = "lde_set_weights_device: wrong weight count";
    return LDE_ERR_INVALID_ARG;
  }
  if (!n) return LDE_OK;
  HIP_TRY(h, hipMemcpyAsync(h->W_dev, flat_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  h->have_W = true;
  return lde::mlp_set_weights(h->mlp, h->W_dev, (hipStream_t)stream, h->err);
}

}  // extern "C"

// ---- step records --------------------------------------------------------------------------------------------------------------
static int rec_capacity(const lde_handle* h, int T, int which) { return lde_host::rec_capacity(h->d, h->opt_record_capacity, T, which); }
// the record a call of shape (B, T) uses: the caller's buffer if one was handed over (forward records), else the handle's own, grown here
static int rec_prepare(lde_handle* h, int which, int B, int T, lde::StepRec* out) {
  const int cap = rec_capacity(h, T, which);
  const bool with_y = which == 0;
  const size_t need = rec_bytes(h->d, B, cap, with_y);
  void* base = nullptr;
  if (which == 0 && h->rec_user) {
    if (h->rec_user_bytes < need) {
      h->err = "step record: the caller's buffer is smaller than lde_step_record_bytes(h, B, T)";
      return LDE_ERR_INVALID_ARG;
    }
    base = h->rec_user;
  } else {
    if (h->rec_own_bytes[which] < need) {
      if (h->rec_own[which]) (void)hipFree(h->rec_own[which]);
      h->rec_own[which] = nullptr;
      h->rec_own_bytes[which] = 0;
      // whatever the last call left in the old buffer is gone: a pullback (or lde_get_step_record) that would still read it through
      // rec_last must be refused, not handed freed memory (e.g. "record_capacity" raised between lde_forward and lde_adjoint)
      if (!(which == 0 && h->rec_user && h->rec_last[0].n && (void*)h->rec_last[0].n == h->rec_user)) h->rec_last[which] = lde::StepRec{};
      HIP_TRY(h, hipMalloc(&h->rec_own[which], need));
      h->rec_own_bytes[which] = need;
    }
    base = h->rec_own[which];
  }
  *out = rec_view(h->d, base, B, cap, with_y);
  return LDE_OK;
}

// adjoint_ws: also size the workspace only lde_adjoint needs (the MLP adjoint's staging area is large)
static int reserve_impl(lde_handle* h, int B, int T, bool adjoint_ws, int64_t steps_hint) {
  if (!h || B < 1 || T < 1) return LDE_ERR_INVALID_ARG;
  if (B > h->cap_B) {
    for (int w = 0; w < 2; w++)
      for (int i = 0; i < 4; i++) {
        if (h->st[w][i]) (void)hipFree(h->st[w][i]);
        h->st[w][i] = nullptr;
        HIP_TRY(h, hipMalloc(&h->st[w][i], (size_t)B * sizeof(int32_t)));
      }
    h->cap_B = B;
  }
  if (T > h->ts_cap) {
    if (h->ts_dev) (void)hipFree(h->ts_dev);
    h->ts_dev = nullptr;
    HIP_TRY(h, hipMalloc(&h->ts_dev, (size_t)T * sizeof(double)));
    h->ts_cap = T;
    h->ts_host.clear();
  }
  if (T > h->ts_pin_cap) {
    for (int i = 0; i < TS_RING; i++) {
      if (h->ts_pinned[i]) (void)hipHostFree(h->ts_pinned[i]);
      h->ts_pinned[i] = nullptr;
      HIP_TRY(h, hipHostMalloc((void**)&h->ts_pinned[i], (size_t)T * sizeof(double), hipHostMallocDefault));
    }
    h->ts_pin_cap = T;
  }
  if (!h->mlp && h->d.sensealg == LDE_SENSE_PARALLEL_CHECKPOINTED && lde::pend_adjoint_needs_ops(B, T)) {
    const size_t need = (size_t)(T > 1 ? T - 1 : 1) * (size_t)B;
    if (need > h->par_cap) {
      if (h->par_ops) (void)hipFree(h->par_ops);
      if (h->par_info) (void)hipFree(h->par_info);
      h->par_ops = nullptr;
      h->par_info = nullptr;
      HIP_TRY(h, hipMalloc(&h->par_ops, need * 6 * sizeof(float)));
      HIP_TRY(h, hipMalloc(&h->par_info, need * sizeof(int32_t)));
      h->par_cap = need;
    }
  }
  if (!h->rec_user && (h->d.sensealg == LDE_SENSE_DISCRETE || h->opt_step_trace)) {   // (lde_reserve pre-sizes the handle's own records too)
    lde::StepRec tmp;
    int rc = rec_prepare(h, 0, B, T, &tmp);
    if (!rc && h->opt_step_trace && adjoint_ws && h->d.sensealg != LDE_SENSE_DISCRETE) rc = rec_prepare(h, 1, B, T, &tmp);
    if (rc) return rc;
  }
  if (h->mlp) {
    const int rc = lde::mlp_reserve(h->mlp, B, T, h->err);
    if (rc || !adjoint_ws) return rc;
    return lde::mlp_reserve_adjoint(h->mlp, B, T, steps_hint, h->err);
  }
  return LDE_OK;
}

extern "C" int lde_reserve(lde_handle* h, int B, int T) { return reserve_impl(h, B, T, true, 0); }

// Make the device copy of the save-time grid current (no-op when `ts` is unchanged since the last call).
static int stage_ts(lde_handle* h, const double* ts, int T, hipStream_t stream) {
  if (!grid_ok(ts, T)) {
    h->err = "ts must be finite and strictly increasing";
    return LDE_ERR_INVALID_ARG;
  }
  if ((int)h->ts_host.size() == T && std::memcmp(h->ts_host.data(), ts, (size_t)T * sizeof(double)) == 0) return LDE_OK;
  const int slot = h->ring;
  h->ring = (h->ring + 1) % TS_RING;
  HIP_TRY(h, hipEventSynchronize(h->ts_ev[slot]));  // slot's previous copy has drained
  std::memcpy(h->ts_pinned[slot], ts, (size_t)T * sizeof(double));
  HIP_TRY(h, hipMemcpyAsync(h->ts_dev, h->ts_pinned[slot], (size_t)T * sizeof(double), hipMemcpyHostToDevice, stream));
  HIP_TRY(h, hipEventRecord(h->ts_ev[slot], stream));
  h->ts_host.assign(ts, ts + T);
  return LDE_OK;
}

extern "C" {

int lde_forward(lde_handle* h, const float* z0, const float* theta, const double* ts, int T, int B, float* z_out,
                int32_t* retcode, void* stream_) {
  if (!h) return LDE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (!z0 || !ts || !z_out || T < 1 || B < 1 || (h->d.param_dim > 0 && !theta)) {
    h->err = "lde_forward: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (h->nW && !h->have_W) {
    h->err = "lde_forward: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  int rc = reserve_impl(h, B, T, false, 0);
  if (rc) return rc;
  rc = stage_ts(h, ts, T, stream);
  if (rc) return rc;
  lde::KOpts o = make_opts(h->d, ts, T, B);
  h->rec_last[0] = lde::StepRec{};
  if (h->d.sensealg == LDE_SENSE_DISCRETE || h->opt_step_trace) {
    rc = rec_prepare(h, 0, B, T, &o.rec);
    if (rc) return rc;
    h->rec_last[0] = o.rec;
    h->rec_B = B;
    h->rec_T = T;
  }
  int32_t** st = h->st[0];
  h->last_B[0] = B;
  if (h->mlp)
    return lde::mlp_forward(h->mlp, h->W_dev, z0, theta, h->ts_dev, o, z_out, retcode, st[0], st[1], st[2], st[3], stream, h->err);
  rc = lde::launch_pend_forward(h->d.rhs_kind, h->d.solver, z0, theta, h->ts_dev, o, z_out, retcode, st[0], st[1], st[2],
                                st[3], stream, h->pend_tune);
  h->last_kernel[0] = lde::pend_last_kernel(0);
  if (rc) h->err = "lde_forward: kernel launch failed";
  return rc;
}

int lde_adjoint(lde_handle* h, const float* z_out, const float* theta, const double* ts, int T, int B,
                const float* dz_out, float* dz0, float* dtheta, float* dW, void* stream_) {
  if (!h) return LDE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (!z_out || !ts || !dz_out || !dz0 || T < 1 || B < 1 || (h->d.param_dim > 0 && (!theta || !dtheta)) ||
      (h->nW && !dW)) {
    h->err = "lde_adjoint: NULL pointer or empty batch";
    return LDE_ERR_INVALID_ARG;
  }
  if (h->nW && !h->have_W) {
    h->err = "lde_adjoint: weights not set";
    return LDE_ERR_NO_WEIGHTS;
  }
  const int64_t steps_hint = h->mlp ? fixed_step_count(h->d, ts, T) : 0;   // fixed step size: the number of step attempts is known here
  int rc = reserve_impl(h, B, T, true, steps_hint);
  if (rc) return rc;
  rc = stage_ts(h, ts, T, stream);
  if (rc) return rc;
  lde::KOpts o = make_opts(h->d, ts, T, B);
  o.dw_overwrite = h->opt_adjoint_overwrite;
  int32_t** st = h->st[1];
  h->last_B[1] = B;
  h->rec_last[1] = lde::StepRec{};
  if (h->d.sensealg == LDE_SENSE_DISCRETE) {
    // the record of the forward solve this call differentiates: the caller's buffer (lde_set_step_record) or the handle's own
    lde::StepRec r;
    if (h->rec_user) {
      rc = rec_prepare(h, 0, B, T, &r);
      if (rc) return rc;
    } else {
      if (!h->rec_last[0].n || h->rec_B != B || h->rec_T != T) {
        h->err = "lde_adjoint (LDE_SENSE_DISCRETE): no step record of an lde_forward with this (B, T) on this handle";
        return LDE_ERR_INVALID_ARG;
      }
      r = h->rec_last[0];
    }
    o.rec = r;
    if (h->mlp)
      return lde::mlp_adjoint(h->mlp, h->W_dev, z_out, theta, h->ts_dev, o, dz_out, dz0, dtheta, dW, st[0], st[1], st[2], st[3], stream,
                              h->err);
    rc = lde::launch_pend_adjoint_disc(h->d.rhs_kind, h->d.solver, z_out, theta, h->ts_dev, o, dz_out, dz0, dtheta, st[0], st[1], st[2],
                                       st[3], stream, h->pend_tune);
    h->last_kernel[1] = lde::pend_last_kernel(1);
    if (rc) h->err = "lde_adjoint: kernel launch failed";
    return rc;
  }
  if (h->opt_step_trace) {
    rc = rec_prepare(h, 1, B, T, &o.rec);
    if (rc) return rc;
    h->rec_last[1] = o.rec;
    HIP_TRY(h, hipMemsetAsync(o.rec.n, 0, (size_t)o.rec.nseq * 4, stream));
  }
  if (h->mlp)
    return lde::mlp_adjoint(h->mlp, h->W_dev, z_out, theta, h->ts_dev, o, dz_out, dz0, dtheta, dW, st[0], st[1], st[2],
                            st[3], stream, h->err);
  if (h->d.sensealg == LDE_SENSE_PARALLEL_CHECKPOINTED)
    rc = lde::launch_pend_adjoint_par(h->d.rhs_kind, h->d.solver, z_out, theta, h->ts_dev, o, dz_out, dz0, dtheta, h->par_ops,
                                      h->par_info, st[0], st[1], st[2], st[3], stream);
  else
    rc = lde::launch_pend_adjoint(h->d.rhs_kind, h->d.solver, z_out, theta, h->ts_dev, o, dz_out, dz0, dtheta, st[0], st[1],
                                  st[2], st[3], stream);
  h->last_kernel[1] = lde::pend_last_kernel(1);
  if (rc) h->err = "lde_adjoint: kernel launch failed";
  return rc;
}

int lde_get_stats(lde_handle* h, int which, lde_stats* out, void* stream_) {
  if (!h || !out || which < 0 || which > 1) return LDE_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  const int B = h->last_B[which];
  if (B < 1) return LDE_OK;
  HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream_));
  std::vector<int32_t> buf((size_t)B * 4, 0);
  for (int i = 0; i < 4; i++)
    HIP_TRY(h, hipMemcpy(buf.data() + (size_t)i * B, h->st[which][i], (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int b = 0; b < B; b++) {
    out->nfe += buf[b];
    out->naccept += buf[(size_t)B + b];
    out->nreject += buf[(size_t)2 * B + b];
    const int64_t steps = (int64_t)buf[(size_t)B + b] + buf[(size_t)2 * B + b];
    if (steps > out->max_steps) out->max_steps = steps;
    if (buf[(size_t)3 * B + b]) out->nfailed++;
  }
  return LDE_OK;
}

int lde_set_phase_timing(lde_handle* h, int on) {
  if (!h || !h->mlp) return LDE_ERR_INVALID_ARG;
  return lde::mlp_set_phase_timing(h->mlp, on);
}
int lde_get_phase_ms(lde_handle* h, float* ms2) {
  if (!h || !h->mlp || !ms2) return LDE_ERR_INVALID_ARG;
  return lde::mlp_get_phase_ms(h->mlp, ms2);
}

int lde_set_global_sum_hook(lde_handle* h, lde_sum_hook hook, void* user, int64_t global_batch) {
  if (!h) return LDE_ERR_INVALID_ARG;
  if (h->d.batching != LDE_BATCH_COUPLED_GLOBAL || !h->mlp) {
    h->err = "lde_set_global_sum_hook: the handle was not created with LDE_BATCH_COUPLED_GLOBAL";
    return LDE_ERR_INVALID_ARG;
  }
  if (hook && global_batch < 1) {
    h->err = "lde_set_global_sum_hook: global_batch < 1";
    return LDE_ERR_INVALID_ARG;
  }
  return lde::mlp_set_sum_hook(h->mlp, hook, user, hook ? global_batch : 0, h->err);
}

int64_t lde_global_sum_mailbox_bytes(int nranks) { return nranks < 1 || nranks > 8 ? 0 : (int64_t)lde::mlp_sum_mailbox_bytes(nranks); }
int lde_set_global_sum_peers(lde_handle* h, int rank, int nranks, void* const* mailboxes, int64_t global_batch) {
  if (!h) return LDE_ERR_INVALID_ARG;
  if (h->d.batching != LDE_BATCH_COUPLED_GLOBAL || !h->mlp) {
    h->err = "lde_set_global_sum_peers: the handle was not created with LDE_BATCH_COUPLED_GLOBAL";
    return LDE_ERR_INVALID_ARG;
  }
  return lde::mlp_set_sum_peers(h->mlp, rank, nranks, mailboxes, global_batch, h->err);
}

int64_t lde_step_record_bytes(const lde_handle* h, int B, int T) {
  if (!h || B < 1 || T < 1) return 0;
  return (int64_t)rec_bytes(h->d, B, rec_capacity(h, T, 0), true);
}

int lde_set_step_record(lde_handle* h, void* rec_dev, int64_t bytes) {
  if (!h || (rec_dev && bytes < 1) || ((uintptr_t)rec_dev & 255)) {
    if (h) h->err = "lde_set_step_record: the buffer must be 256-byte aligned device memory";
    return LDE_ERR_INVALID_ARG;
  }
  h->rec_user = rec_dev;
  h->rec_user_bytes = rec_dev ? (size_t)bytes : 0;
  return LDE_OK;
}

int lde_step_record_capacity(const lde_handle* h, int T) { return (!h || T < 1) ? 0 : rec_capacity(h, T, 0); }

int lde_step_record_status(lde_handle* h, const void* rec_dev, int B, int T, int32_t* max_steps, int32_t* capacity, void* stream_) {
  if (!h || !max_steps || B < 1 || T < 1) return LDE_ERR_INVALID_ARG;
  lde::StepRec r;
  if (rec_dev) {
    r = rec_view(h->d, const_cast<void*>(rec_dev), B, rec_capacity(h, T, 0), true);
  } else {
    r = h->rec_last[0];
    if (!r.n || h->rec_B != B || h->rec_T != T) {
      h->err = "lde_step_record_status: no step record of an lde_forward with this (B, T) on this handle";
      return LDE_ERR_INVALID_ARG;
    }
  }
  HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream_));
  std::vector<int32_t> n((size_t)r.nseq);
  HIP_TRY(h, hipMemcpy(n.data(), r.n, n.size() * 4, hipMemcpyDeviceToHost));
  int32_t m = 0;
  for (int32_t v : n) m = std::max(m, v);
  *max_steps = m;
  if (capacity) *capacity = r.cap;
  return LDE_OK;
}

int lde_get_step_record(lde_handle* h, int which, double* t_host, double* dt_host, int32_t* n_host, int nseq, int cap, void* stream_) {
  if (!h || which < 0 || which > 1 || !dt_host || !n_host || nseq < 1 || cap < 1) return LDE_ERR_INVALID_ARG;
  const lde::StepRec& r = h->rec_last[which];
  if (!r.n) {
    h->err = "lde_get_step_record: the last call made no record (LDE_SENSE_DISCRETE or option \"step_trace\")";
    return LDE_ERR_INVALID_ARG;
  }
  if (nseq != r.nseq) {
    h->err = "lde_get_step_record: nseq does not match the record";
    return LDE_ERR_INVALID_ARG;
  }
  HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream_));
  HIP_TRY(h, hipMemcpy(n_host, r.n, (size_t)nseq * 4, hipMemcpyDeviceToHost));
  std::vector<double> buf((size_t)r.cap * nseq);
  for (int a = 0; a < 2; a++) {
    double* dst = a == 0 ? t_host : dt_host;
    const double* src = a == 0 ? r.t : r.dt;
    if (!dst || (a == 0 && which == 1)) continue;
    HIP_TRY(h, hipMemcpy(buf.data(), src, buf.size() * 8, hipMemcpyDeviceToHost));
    for (int q = 0; q < nseq; q++) {
      const int n = std::min(std::min(n_host[q], r.cap), cap);
      for (int i = 0; i < cap; i++) dst[(size_t)q * cap + i] = i < n ? buf[(size_t)i * nseq + q] : 0.0;
    }
  }
  return LDE_OK;
}

static int* option_slot(lde_handle* h, const char* key) {
  if (!std::strcmp(key, "record_capacity")) return &h->opt_record_capacity;
  if (!std::strcmp(key, "step_trace")) return &h->opt_step_trace;
  if (!std::strcmp(key, "adjoint_overwrite")) return &h->opt_adjoint_overwrite;
  // kernel-choice knobs (include/lde.h): what the tests force a family / a threshold with
  lde::PendTune& pt = h->pend_tune;
  if (!std::strcmp(key, "pend_ws")) return &pt.ws;
  if (!std::strcmp(key, "pend_tl_max_b")) return &pt.tl_max_b;
  if (!std::strcmp(key, "pend_sh_max_b")) return &pt.sh_max_b;
  if (!std::strcmp(key, "pend_lp")) return &pt.lp;
  if (!std::strcmp(key, "pend_lb")) return &pt.lb_ring;
  if (!std::strcmp(key, "pend_lb_min_b")) return &pt.lb_min_b;
  if (!std::strcmp(key, "pend_lb_hold")) return &pt.lb_hold;
  if (!std::strcmp(key, "pend_disc_tp_max_b")) return &pt.disc_tp_max_b;
  if (h->mlp) {
    lde::MlpTune& mt = *lde::mlp_tune(h->mlp);
    if (!std::strcmp(key, "mlp64")) return &mt.mlp64;
    if (!std::strcmp(key, "mlpv")) return &mt.mlpv;
    if (!std::strcmp(key, "mlpw")) return &mt.mlpw;
    if (!std::strcmp(key, "mlpb")) return &mt.mlpb;
    if (!std::strcmp(key, "mlp4")) return &mt.mlp4;
    if (!std::strcmp(key, "mlp4_maxw")) return &mt.mlp4_maxw;
    if (!std::strcmp(key, "mlp_stage_slots")) return &mt.stage_slots;
    if (!std::strcmp(key, "peer_spin_k")) return &mt.peer_spin_k;
  }
  return nullptr;
}
int lde_set_option(lde_handle* h, const char* key, double value) {
  if (!h || !key) return LDE_ERR_INVALID_ARG;
  int* slot = option_slot(h, key);
  if (!slot || !(value >= ((std::strcmp(key, "pend_lb_hold") && std::strcmp(key, "pend_sh_max_b")) ? 0 : -1)) || value > 2e9) {
    h->err = std::string("lde_set_option: unknown key or value out of range: ") + key;
    return LDE_ERR_INVALID_ARG;
  }
  *slot = (int)value;
  return LDE_OK;
}
int lde_get_option(const lde_handle* h, const char* key, double* value) {
  if (!h || !key || !value) return LDE_ERR_INVALID_ARG;
  if (!std::strcmp(key, "adjoint_family")) {   // read-only: the kernel family the last lde_adjoint ran (MLP right-hand sides; −1: none)
    *value = h->mlp ? (double)lde::mlp_last_family(h->mlp) : -1.0;
    return LDE_OK;
  }
  int* slot = option_slot(const_cast<lde_handle*>(h), key);
  if (!slot) return LDE_ERR_INVALID_ARG;
  *value = (double)*slot;
  return LDE_OK;
}

const char* lde_last_error(const lde_handle* h) { return h ? h->err.c_str() : "NULL handle"; }

const char* lde_last_kernel(const lde_handle* h, int which) {
  if (!h || which < 0 || which > 1) return "";
  if (h->mlp) {   // the MLP families by the adjoint's last choice ("adjoint_family"); the forward solve runs the same family's forward instantiation
    static const char* fam[] = {"k_mlp_", "k_mlp64", "k_mlpb", "k_mlpc", "k_mlpw", "k_mlpv", "k_mlp4"};
    const int f = lde::mlp_last_family(h->mlp);
    return f >= 0 && f < 7 ? fam[f] : "";
  }
  return h->last_kernel[which];
}

}  // extern "C"

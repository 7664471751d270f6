/*
 * lde.h — C ABI of the MI355X-native latent-ODE solve + adjoint ("liblde").
 *
 * This is the drop-in boundary for ONE hot path of gabrevaya/LatentDiffEq.jl:
 * what runs *under*
 *
 *     ẑ = diffeq_layer(decoder, l̂, t)            [REF src/models/LatentDiffEqModel.jl:107]
 *
 * i.e. the bodies of
 *     diffeq_layer(::Decoder{<:GOKU},    (ẑ₀, θ̂), t)   [REF src/models/GOKU.jl:98-130]
 *     diffeq_layer(::Decoder{LatentODE},  ẑ₀,      t)   [REF src/models/LatentODE.jl:61-78]
 * and their reverse-mode pullbacks (in the reference these are delegated to the un-vendored
 * OrdinaryDiffEq 6.27.1 / SciMLSensitivity 7.10.0 / DiffEqFlux 1.52.0, pinned in Manifest.toml).
 *
 * The reference has no FFI today (pure Julia multiple dispatch); the entry points below are what a
 * `ccall` binding inside `diffeq_layer` + its `ChainRulesCore.rrule` would bind.
 *
 * Conventions
 *  - plain C types only; every array is a raw pointer + sizes. No torch / HIP types in signatures:
 *    a stream is passed as `void*` (a hipStream_t; NULL = the default stream).
 *  - `z0`, `theta`, `z_out`, `dz_out`, `dz0`, `dtheta`, `dW`, `retcode` are DEVICE pointers
 *    (caller-owned). `ts` and flat weights given to lde_set_weights are HOST pointers.
 *  - array layouts are the reference's (Julia column-major):
 *        z0     [D  × B]      element (d,b)   at d + D*b            [REF GOKU.jl:111  ẑ₀[:,i]]
 *        theta  [P  × B]      element (p,b)   at p + P*b            [REF GOKU.jl:111  θ̂[:,i]]
 *        z_out  [D' × B × T]  element (d,b,j) at d + D'*(b + B*j)   [REF GOKU.jl:125  permutedims(ẑ,[1,3,2])]
 *    with D' = D + augment_dim                                      [REF LatentODE.jl:71]
 *  - flat weight layout = Flux.destructure order: per Dense layer vec(W) (column-major [out×in])
 *    followed by b                                                  [REF nODE.jl:12-14]
 *  - all calls return 0 on success or a negative lde_status; they never abort/throw.
 *    A trajectory whose solve fails (maxiters, dt<dtmin, non-finite) gets retcode!=0 and its
 *    [D'×T] block filled with NaN — the call itself still returns 0  [REF GOKU.jl:114].
 *  - a handle is not thread-safe; calls are asynchronous on the given stream.
 */
#ifndef LDE_H
#define LDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDE_ABI_VERSION 1
#define LDE_MAX_LAYERS 6          /* Dense layers in the RHS MLP */

/* ---- enums (kept as plain ints inside the struct for a stable ABI) ---------------------- */

/* Right-hand sides: the closed menu that replaces "any Julia function". */
enum lde_rhs_kind {
  LDE_RHS_PENDULUM          = 0,  /* du = [y, -G/L sin x], G=10, L=p[1]    [REF examples/pendulum_friction-less/pendulum.jl:19-26] */
  LDE_RHS_PENDULUM_FRICTION = 1,  /* ... - (b/m) y, b=0.7, m=1             [REF pendulum.jl:65-74] */
  LDE_RHS_MLP               = 2,  /* Chain(Dense(relu)...Dense)            [REF examples/pendulum_friction-less/nODE.jl:12-14] */
  LDE_RHS_PENDULUM_PLUS_MLP = 3   /* pendulum(z,L) + MLP(z): BASELINE.json configs[2] */
};

enum lde_solver {
  LDE_SOLVER_TSIT5 = 0,           /* Tsit5()  [REF pendulum.jl:11], [REF nODE.jl:15] */
  LDE_SOLVER_RK4   = 1            /* RK4(), fixed step only (adaptive=0, dt=h) — BASELINE.json configs[1] */
};

enum lde_batching {
  LDE_BATCH_PER_TRAJECTORY = 0,   /* B independent solves, own dt each: EnsembleProblem  [REF GOKU.jl:111-121] */
  LDE_BATCH_COUPLED        = 1,   /* one solve on the [D'×B] matrix state, shared dt, RMS norm over D'·B: NeuralODE [REF LatentODE.jl:70-72] */
  LDE_BATCH_COUPLED_GLOBAL = 2    /* the same ONE solve with its batch SHARDED over ranks (SURVEY.md §8e option (ii)): this handle integrates its
                                     rank's columns, and every sum of the step control — the two Hairer initial-step norms and each attempt's
                                     Σr² — is the sum over ALL ranks' columns, exchanged through lde_set_global_sum_hook. Results equal the
                                     unsharded LDE_BATCH_COUPLED solve up to summation order. A parity mode: lde_forward / lde_adjoint are
                                     SYNCHRONOUS in it (the host services the exchange while the kernel runs). Three-layer MLP right-hand
                                     sides served by the register kernels (2·D' ≤ 64, H ≤ 200); otherwise LDE_ERR_UNSUPPORTED. */
};

enum lde_sensealg {
  LDE_SENSE_BACKSOLVE_CHECKPOINTED = 0, /* reverse-time adjoint, z re-integrated backwards and reset to the saved ẑ(t_j) at every save time */
  LDE_SENSE_BACKSOLVE              = 1, /* same without the reset (BacksolveAdjoint, hinted at [REF nODE.jl:17]) */
  LDE_SENSE_DISCRETE               = 3, /* discrete (exact) sensitivity: the derivative of the DISCRETE solve on its accepted step sequence, step sizes held
                                           constant — what the reference's GOKU default ForwardDiffSensitivity() delivers
                                           [REF examples/pendulum_friction-less/pendulum.jl:11], [REF src/models/GOKU.jl:107, :121] (there by dual numbers
                                           through the stepper, here by reverse mode: same derivative). lde_forward records (t_n, dt_n, y_n) per accepted
                                           step (lde_set_step_record / the handle's own buffer); lde_adjoint sweeps those steps in reverse: the stages are
                                           rebuilt from y_n, the cotangent is pulled through the stage sums, through the dense-output weights b_i(Θ_j) of
                                           every save time inside the step and through the FSAL slope; no controller, no forced stops, no error norm.
                                           lde_adjoint must follow the lde_forward that made the record (same B, T, ts). One deviation from upstream, stated:
                                           under dual numbers OrdinaryDiffEq's error norm also counts the partials, so the reference's accepted step
                                           sequence under ForwardDiffSensitivity differs from its primal solve's; here the primal sequence is used.
                                           With LDE_BATCH_COUPLED_GLOBAL every rank records the common step sequence and its own columns' states; the
                                           sweep has no step control, so the sharded pullback exchanges nothing. */
  LDE_SENSE_PARALLEL_CHECKPOINTED  = 2  /* checkpointed adjoint, parallel in time: with z reset at every save time the T-1 save
                                           intervals are independent and λ enters linearly, so each (trajectory, interval) pair
                                           integrates the interval's transition operator (λ_j = M_j λ_{j+1}, g += n_j·λ_{j+1}) on
                                           its own lane and a short scan composes them. Same continuous adjoint as mode 0, agreeing
                                           to solver tolerance. Implemented for analytic right-hand sides with per-trajectory batching;
                                           for MLP right-hand sides / coupled batching mode 0 runs instead. */
};

enum lde_activation { LDE_ACT_RELU = 0, LDE_ACT_TANH = 1 };

enum lde_status {
  LDE_OK               =  0,
  LDE_ERR_INVALID_ARG  = -1,
  LDE_ERR_UNSUPPORTED  = -2,
  LDE_ERR_NO_DEVICE    = -3,
  LDE_ERR_HIP          = -4,
  LDE_ERR_NO_WEIGHTS   = -5,
  LDE_ERR_ALLOC        = -6
};

/* per-trajectory return codes written to retcode[B] */
enum lde_retcode {
  LDE_RET_SUCCESS   = 0,
  LDE_RET_MAXITERS  = 1,
  LDE_RET_DTMIN     = 2,
  LDE_RET_NONFINITE = 3
};

/* ---- problem description: the `diffeq` plug-in struct, flattened ------------------------- */
/* mirrors Pendulum{prob,solver,sensealg,kwargs}   [REF pendulum.jl:4-46]
 *     and NODE{dudt,solver,neural_model,latent_dim_in,latent_dim_out,augment_dim,kwargs} [REF nODE.jl:3-32];
 * the option block mirrors the `kwargs...` splat into solve()  [REF GOKU.jl:121], [REF LatentODE.jl:70]. */
typedef struct lde_problem_desc {
  int32_t abi_version;            /* = LDE_ABI_VERSION */
  int32_t rhs_kind;               /* lde_rhs_kind */
  int32_t state_dim;              /* D  (latent_dim_in)  */
  int32_t param_dim;              /* P  (per-trajectory ODE parameters θ̂; 1 for the pendulum, 0 for NODE) */
  int32_t augment_dim;            /* extra zero-initialised state rows (AugmentedNDELayer) */
  int32_t n_layers;               /* Dense layers of the RHS MLP (0 if none) */
  int32_t layer_sizes[LDE_MAX_LAYERS + 1]; /* [in, h1, ..., out]; in = out = D + augment_dim */
  int32_t activation;             /* hidden-layer activation (last layer is linear) */
  int32_t solver;                 /* lde_solver */
  int32_t batching;               /* lde_batching */
  int32_t sensealg;               /* lde_sensealg */
  int32_t adaptive;               /* 1 = error-controlled steps, 0 = fixed dt */
  int64_t maxiters;               /* default 100000 */
  double  dt;                     /* fixed step size (adaptive=0); 0 = automatic initial dt (adaptive=1) */
  double  abstol;                 /* default 1e-6 (OrdinaryDiffEq default) */
  double  reltol;                 /* default 1e-3 */
  double  dtmin;                  /* default 0 → 1e-12·|tspan| is used */
  double  qmin, qmax, gamma;      /* PI controller: 0.2, 10, 0.9 */
  double  beta1, beta2;           /* 7/50, 2/25 (Tsit5) */
} lde_problem_desc;

typedef struct lde_stats {
  int64_t nfe;                    /* RHS evaluations, summed over trajectories (coupled: of the matrix RHS) */
  int64_t naccept;
  int64_t nreject;
  int64_t nfailed;                /* trajectories with retcode != 0 */
  int64_t max_steps;              /* max over trajectories of accepted+rejected steps */
} lde_stats;

typedef struct lde_handle lde_handle;

/* ---- entry points ------------------------------------------------------------------------ */

/* library ABI version (= LDE_ABI_VERSION it was built with). */
int lde_abi_version(void);

/* What this binary was built and validated with: the hipcc version line and the verdict of the build's register check (the weight-gradient
 * tiles of k_mlpb / k_mlpc live in accumulator registers the compiler is not told about; every build disassembles the object and refuses to
 * link if the compiler's own code touches them — latentdiffeq.jl_amd/check_agprs.py). A static string; never NULL. */
const char* lde_build_info(void);

/* Fill `desc` with the defaults `Pendulum()` would carry: Tsit5, per-trajectory, abstol 1e-6,
 * reltol 1e-3, maxiters 1e5, PI controller constants, and sensealg = LDE_SENSE_DISCRETE — `Pendulum()`'s ForwardDiffSensitivity()
 * [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into solve() at [REF src/models/GOKU.jl:107, :121]: the exact derivative
 * of the discrete solve. A binding of `NODE` sets LDE_SENSE_BACKSOLVE_CHECKPOINTED itself (DiffEqFlux's InterpolatingAdjoint
 * [REF src/models/LatentODE.jl:67-70]); the time-parallel continuous adjoint (LDE_SENSE_PARALLEL_CHECKPOINTED) stays selectable. */
int lde_problem_desc_default(lde_problem_desc* desc);

/* Number of floats in the flat weight vector implied by desc (0 for analytic RHS). */
int64_t lde_num_weights(const lde_problem_desc* desc);

/* Create / destroy a solver handle for one `diffeq` struct on the current HIP device.
 * Replaces the per-call `remake`/`EnsembleProblem`/`NeuralODE(...)` construction
 * [REF GOKU.jl:111-118], [REF LatentODE.jl:70-71]. */
int  lde_create(const lde_problem_desc* desc, lde_handle** out);
void lde_destroy(lde_handle* h);

/* Upload the RHS-MLP weights (HOST pointer, Flux.destructure order, n = lde_num_weights).
 * Replaces `p, re = Flux.destructure(dudt)` on every call [REF LatentODE.jl:70 → DiffEqFlux NeuralODE]. */
int lde_set_weights(lde_handle* h, const float* flat_host, int64_t n);
/* Same from a DEVICE pointer (async copy on `stream`). */
int lde_set_weights_device(lde_handle* h, const float* flat_dev, int64_t n, void* stream);

/* Pre-size the handle's workspace for batches up to B and T save points (so that later
 * lde_forward/lde_adjoint calls allocate nothing and can be captured in a hipGraph). */
int lde_reserve(lde_handle* h, int B, int T);

/* Forward solve: the body of diffeq_layer            [REF GOKU.jl:98-130], [REF LatentODE.jl:61-78].
 *   z0    [D×B]  device, theta [P×B] device (NULL if P==0), ts[T] HOST, strictly increasing (f64),
 *   z_out [D'×B×T] device (written), retcode[B] device int32 (written; may be NULL). */
int lde_forward(lde_handle* h, const float* z0, const float* theta,
                const double* ts, int T, int B,
                float* z_out, int32_t* retcode, void* stream);

/* Reverse-mode pullback of lde_forward (continuous adjoint with jumps at the save times).
 *   z_out  [D'×B×T] device: the forward result (saved for backward),
 *   dz_out [D'×B×T] device: incoming cotangent ∂L/∂ẑ,
 *   dz0    [D×B]    device (written),
 *   dtheta [P×B]    device (written; NULL if P==0),
 *   dW     [n_weights] device, ACCUMULATED (+=) — caller zeroes it (NULL if no MLP). */
int lde_adjoint(lde_handle* h, const float* z_out, const float* theta,
                const double* ts, int T, int B,
                const float* dz_out,
                float* dz0, float* dtheta, float* dW, void* stream);

/* Solver statistics of the most recent lde_forward (which=0) or lde_adjoint (which=1) on this
 * handle. Synchronises `stream`. */
int lde_get_stats(lde_handle* h, int which, lde_stats* out, void* stream);

/* Measurement aid (MLP right-hand sides): with timing on, every lde_adjoint records HIP events — on the caller's stream — around its
 * solve kernel and around what follows it (the weight-gradient product where there is one, the fixed-order sums). lde_get_phase_ms
 * waits for the last call and returns ms2[0] = the solve kernel, ms2[1] = the tail. bench.py prices its per-kernel roofline with
 * it; off by default (two event records per call). No counterpart in the reference (it has no profiling hooks). */
int lde_set_phase_timing(lde_handle* h, int on);
int lde_get_phase_ms(lde_handle* h, float* ms2);

/* LDE_BATCH_COUPLED_GLOBAL: the exchange of the step-control sums between the ranks that share ONE coupled solve
 * [REF src/models/LatentODE.jl:70-72: the reference's NeuralODE norm runs over the whole [D'×B] state — here B is spread over ranks].
 * `hook(user, vals, n)` must replace vals[0..n) (n = 1 or 2, f64) by their sums over all ranks and return 0; every rank calls it
 * the same number of times in the same order (the decisions that follow are functions of the sums, hence identical everywhere). It
 * is called on the thread that called lde_forward / lde_adjoint, while the solve's kernel waits for the answer — use a HOST
 * collective (MPI, gloo, a pipe): the device is occupied — and nothing in the hook may synchronise the device (hipFree, a null-stream
 * copy, a finaliser that releases device memory): it would wait for the kernel that waits for the hook. `global_batch` = Σ over ranks of
 * their B (the norm's divisor). The values are float32 sums widened to f64 (the device forms and consumes them in f32). If the hook
 * returns non-zero the solve is poisoned with NaN sums and ends with retcode != 0 / rc -1; requests that still arrive are passed to
 * the hook with NaN payloads (its result ignored) so that peers still in the exchange fail too — but a failed solve issues few
 * requests, so give the host collective a finite time-out: a peer then fails by it instead of waiting for ever.
 * hook == NULL clears it (the mode then equals LDE_BATCH_COUPLED on this rank's columns). */
typedef int (*lde_sum_hook)(void* user, double* vals, int n);
int lde_set_global_sum_hook(lde_handle* h, lde_sum_hook hook, void* user, int64_t global_batch);

/* The same exchange WITHOUT the host (round 5): every rank owns a mailbox of lde_global_sum_mailbox_bytes(nranks) bytes of device memory,
 * zero-initialised once, that every other rank's device can write (one process per GPU: fine-grained memory — hipExtMallocWithFlags(…,
 * hipDeviceMallocFinegrained) — shared by hipIpcGetMemHandle / hipIpcOpenMemHandle with peer access enabled; ranks that share a device
 * pass plain device pointers). `mailboxes[r]` = rank r's mailbox AS MAPPED ON THIS DEVICE, the same order on every rank, `rank` = this
 * handle's index in it. Workgroup 0 of the solve's kernel writes this rank's (float32) sums into slot `rank` of every mailbox, waits for the
 * nranks words of its own and adds them in rank order — the same bits on every rank — so lde_forward / lde_adjoint stay asynchronous and
 * a sum costs one xGMI round trip instead of a host collective. Every rank must make the same sequence of calls on its handle (the words are
 * tagged with the count of exchanging launches since lde_set_global_sum_peers — a counter of its own that never skips or restarts — and the
 * launch's sum count, and those are compared across ranks); a rank that stops answering poisons the others' sums after a bounded spin
 * (option "peer_spin_k"; retcode != 0, never a hang). nranks ≤ 8 (one node); nranks == 0 switches the path off;
 * setting peers clears a hook. (Ranks that SHARE a device — the tests' construction — must issue their calls on streams that cannot share a
 * hardware queue, e.g. streams of different priority: the kernels wait for each other, so they have to be in flight at once.)
 * [REF src/models/LatentODE.jl:70-72] as above. Exercised in this repository by two handles on ONE device and by two PROCESSES sharing
 * one device through HIP IPC handles (tests/test_gpu_coupled_global.py, tests/test_gpu_ipc_mailboxes.py) — the mapping across DEVICES has
 * not run on hardware here (one-GPU boxes). */
int64_t lde_global_sum_mailbox_bytes(int nranks);
int lde_set_global_sum_peers(lde_handle* h, int rank, int nranks, void* const* mailboxes, int64_t global_batch);

/* ---- step records: LDE_SENSE_DISCRETE's hand-over between lde_forward and lde_adjoint, and the parity tests' view of a solve's steps ----
 * A record holds, per step sequence (one per trajectory with LDE_BATCH_PER_TRAJECTORY, one for a coupled solve) and accepted step n,
 * the step's start time and size (f64) and — forward records — the state y_n [D'×B] it started from. With sensealg = LDE_SENSE_DISCRETE
 * lde_forward writes one and lde_adjoint reads it. By default it lives in the handle: ONE outstanding forward per handle. A caller with
 * several forwards in flight before their pullbacks (an AD tape) owns the records instead: lde_step_record_bytes(h, B, T) device bytes
 * each, handed over with lde_set_step_record before lde_forward AND before the matching lde_adjoint (NULL: back to the handle's own).
 * Capacity = option "record_capacity" accepted steps per sequence (default max(64, 4·T)); a solve that needs more leaves the record
 * incomplete and the discrete adjoint then returns NaN gradients with retcode LDE_RET_MAXITERS in its statistics — never a silently
 * truncated sweep. The reference has no counterpart (dual numbers carry the derivative through the solve [REF src/models/GOKU.jl:121]). */
int64_t lde_step_record_bytes(const lde_handle* h, int B, int T);
int lde_set_step_record(lde_handle* h, void* rec_dev, int64_t bytes);
/* Did the record hold the solve? lde_step_record_capacity(h, T) = the accepted steps per sequence a record made now would hold;
 * lde_step_record_status waits for `stream` and returns, for the record of a finished lde_forward of shape (B, T) — `rec_dev`, the caller's
 * buffer, or NULL = the handle's last one — the largest accepted-step count of its sequences (*max_steps; counts run on past the capacity)
 * and the capacity the record was made with (*capacity, may be NULL). max_steps > capacity ⇒ the pullback would return NaN gradients: raise
 * "record_capacity" (to ≥ max_steps), repeat lde_forward (deterministic: the same steps), then lde_adjoint — what a host's pullback does
 * before it hands NaNs to an optimiser (latentdiffeq.jl_amd/api.py: _SolveFn.backward; julia/LdeNative.jl: the rrule). The reference has no
 * counterpart: ForwardDiffSensitivity differentiates any solve up to maxiters [REF src/models/GOKU.jl:121]. */
int lde_step_record_capacity(const lde_handle* h, int T);
int lde_step_record_status(lde_handle* h, const void* rec_dev, int B, int T, int32_t* max_steps, int32_t* capacity, void* stream);
/* Host copy of the step sequences of the last call: which = 0 the forward record (needs LDE_SENSE_DISCRETE or option "step_trace"),
 * which = 1 the reverse-time solve of the continuous adjoint (option "step_trace"; t_host is not written: NULL). Arrays [nseq][cap]
 * row-major, n_host[nseq] (a count > cap: truncated); nseq = B or 1. Synchronises `stream`. What tests/test_gpu_discrete.py hands to the
 * checker so that kernel and oracle are compared on the SAME discrete solve. */
int lde_get_step_record(lde_handle* h, int which, double* t_host, double* dt_host, int32_t* n_host, int nseq, int cap, void* stream);

/* Options that are not part of the reference's `diffeq` struct (a library must not be steered by environment variables):
 *   "record_capacity"  accepted steps a step record holds per sequence (0: automatic)
 *   "step_trace"       1: lde_forward records its steps whatever the sensealg, lde_adjoint (continuous) records its reverse-time steps
 *   "adjoint_overwrite" 1: lde_adjoint WRITES dW (every entry exactly once) instead of accumulating — the caller's zero fill disappears
 * and the kernel-choice knobs the parity tests force a kernel family / a threshold with (defaults = the measured choices; a production
 * host never sets them): "pend_ws", "pend_tl_max_b", "pend_sh_max_b" (−1, the default: the measured thresholds of the two mappings with a
 * trajectory per workgroup; ≥ 0: one threshold for both), "pend_lp", "pend_lb", "pend_lb_min_b", "pend_lb_hold", "pend_disc_tp_max_b" (analytic
 * right-hand sides: which of the forward / pullback mappings serves a batch), "mlp64", "mlpv", "mlpw", "mlpb" (0 off, 2 also ≤ 128-wide networks), "mlp4",
 * "mlp4_maxw", "mlp_stage_slots" (MLP right-hand sides), "peer_spin_k" (lde_set_global_sum_peers: the cross-rank wait poisons the sums after
 * this many × 1024 polls of ≈ 1 µs; 0 = 8192 ≈ 10 s — raise it when ranks may enter a solve seconds apart, e.g. a first call's module load;
 * all ranks should be warmed up before the first exchanging call). The library reads NO environment variable for any of this.
 * lde_get_option also answers the read-only "adjoint_family": the kernel family the last lde_adjoint ran on an MLP right-hand side
 * (0 tiles, 1 k_mlp64, 2 k_mlpb, 3 k_mlpc, 4 k_mlpw, 5 k_mlpv, 6 k_mlp4; −1 none) — what bench.py prices its roofline with.
 * Unknown key: LDE_ERR_INVALID_ARG. */
int lde_set_option(lde_handle* h, const char* key, double value);
int lde_get_option(const lde_handle* h, const char* key, double* value);

/* Human-readable text for the last error on this handle (never NULL). */
const char* lde_last_error(const lde_handle* h);

/* The name (prefix) of the solve kernel the last lde_forward (which = 0) / lde_adjoint (which = 1) on this handle launched, e.g.
 * "k_pend_forward_lp", "k_pend_adjoint_disc_tp", "k_mlpc" — "" before the first call. A measurement aid like lde_get_phase_ms: bench.py attaches
 * HBM counter bytes from a committed rocprofv3 summary to its roofline line only when that summary's kernel IS the one the run launched
 * (a stale summary must not survive a kernel change). Static strings; never NULL. */
const char* lde_last_kernel(const lde_handle* h, int which);


/* ======================================================================================================
 * Dense chains either side of the solve — scope row f-1: what runs under
 *
 *     l̂ = apply_latent_out(decoder, l̃)         [REF src/models/GOKU.jl:83-91], [REF src/models/LatentODE.jl:53]
 *     x̂ = apply_reconstructor(decoder, ẑ)       [REF src/models/GOKU.jl:148],   [REF src/models/LatentODE.jl:80]
 *
 * i.e. a Flux `Chain` of `Dense(in, out, act)` layers, each optionally wrapped in `SkipConnection(·, +)`, applied
 * column-wise: the default layers are  lo_z₀ = Chain(Dense(16,200,relu), Dense(200,D)),
 * lo_θ = Chain(Dense(16,200,relu), Dense(200,P,softplus)),
 * reconstructor = Chain(Dense(D,200,relu), Skip(Dense(200,200,relu)), Skip(Dense(200,200,relu)), Dense(200,784,σ))
 *                                                                  [REF src/models/GOKU.jl:252-269]
 * A Flux Dense on a [D×B×T] array acts on the first dimension, so the reconstructor consumes lde_forward's z_out
 * in place as x [D' × N], N = B·T, and produces x̂ [784 × B × T] in the reference's layout.
 *  - x, y, dy, dx are DEVICE pointers, column-major [rows × N] (rows fastest).
 *  - flat weights: Flux.destructure order of the Chain — per Dense vec(W) (column-major [out×in]) then b.
 *  - lde_chain_backward recomputes the hidden activations from x (nothing is kept from the forward call); it takes
 *    the forward OUTPUT y so that the last (widest) layer is not recomputed.
 */
#define LDE_CHAIN_MAX_LAYERS 6

enum lde_chain_activation {
  LDE_CACT_IDENTITY = 0, LDE_CACT_RELU = 1, LDE_CACT_TANH = 2, LDE_CACT_SIGMOID = 3, LDE_CACT_SOFTPLUS = 4
};

typedef struct lde_chain_desc {
  int32_t abi_version;                           /* = LDE_ABI_VERSION */
  int32_t n_layers;                              /* Dense layers, 1..LDE_CHAIN_MAX_LAYERS */
  int32_t sizes[LDE_CHAIN_MAX_LAYERS + 1];       /* [in, h1, ..., out] */
  int32_t activation[LDE_CHAIN_MAX_LAYERS];      /* lde_chain_activation of each layer */
  int32_t skip[LDE_CHAIN_MAX_LAYERS];            /* 1: SkipConnection(Dense, +): y = x + act(Wx+b); needs in == out; not on the last layer */
} lde_chain_desc;

typedef struct lde_chain lde_chain;

/* Several INDEPENDENT chains in one call: the (μ, log σ²) heads of apply_latent_in [REF src/models/GOKU.jl:61-72] and the two chains of
 * apply_latent_out [REF src/models/GOKU.jl:83-91] act on B columns only — six forward and eighteen pullback launches of a few
 * microseconds each in a training step. With n ≤ 4 chains in the same dtype mode every stage of the call (forward; pullback, weight-
 * gradient product, fixed-order sums) is ONE launch; anything else runs the chains one after the other. Per chain the same kernels on
 * the same arguments as lde_chain_forward_save / lde_chain_backward_saved: results equal bit for bit. saveds / dxs may be NULL or hold
 * NULL entries with the single calls' meaning. */
int  lde_chain_group_forward_save(int n, lde_chain* const* chains, const float* const* xs, const int64_t* Ns, float* const* ys,
                                  float* const* saveds, void* stream);
int  lde_chain_group_backward_saved(int n, lde_chain* const* chains, const float* const* xs, const float* const* ys, const float* const* dys,
                                    const float* const* saveds, const int64_t* Ns, float* const* dxs, float* const* dWs, void* stream);

int64_t lde_chain_num_weights(const lde_chain_desc* desc);
int  lde_chain_create(const lde_chain_desc* desc, lde_chain** out);
void lde_chain_destroy(lde_chain* c);
int  lde_chain_set_weights(lde_chain* c, const float* flat_host, int64_t n);
int  lde_chain_set_weights_device(lde_chain* c, const float* flat_dev, int64_t n, void* stream);
/* Pre-size the workspace for up to N columns (the backward pass stages its activations / deltas in HBM). */
int  lde_chain_reserve(lde_chain* c, int64_t N);
/* y[out×N] = chain(x[in×N]). */
int  lde_chain_forward(lde_chain* c, const float* x, int64_t N, float* y, void* stream);
/* Pullback: dx[in×N] (written; may be NULL when the input needs no gradient), dW[n_weights] ACCUMULATED (+=). */
int  lde_chain_backward(lde_chain* c, const float* x, const float* y, const float* dy, int64_t N,
                        float* dx, float* dW, void* stream);
/* Training variant: the forward call also writes the hidden activations into a caller-owned buffer `saved`
 * (lde_chain_saved_floats(c, N) floats, device), and lde_chain_backward_saved reads them instead of recomputing the
 * hidden layers. Same results as lde_chain_backward; the handle keeps no state between the two calls. */
int64_t lde_chain_saved_floats(const lde_chain* c, int64_t N);
int  lde_chain_forward_save(lde_chain* c, const float* x, int64_t N, float* y, float* saved, void* stream);
int  lde_chain_backward_saved(lde_chain* c, const float* x, const float* y, const float* dy, const float* saved, int64_t N,
                              float* dx, float* dW, void* stream);
/* The pullback of a chain whose OUTPUT feeds several consumers (the feature extractor's frames go to the three pattern extractors
 * [REF src/models/GOKU.jl:32-51]): the output gradient is (dys[0] + dys[1]) + dys[2], formed where the kernel reads it — the sums a caller
 * would otherwise launch (two elementwise kernels over [out×N] in a GOKU step). n_dy = 1..3; 16-byte aligned arrays; saved may be NULL
 * (then as lde_chain_backward). With n_dy = 1 the call IS lde_chain_backward_saved. */
int  lde_chain_backward_saved_sum(lde_chain* c, const float* x, const float* y, int n_dy, const float* const* dys, const float* saved,
                                  int64_t N, float* dx, float* dW, void* stream);
/* The forward call of such a chain with the loss value from the same launch: out[0] = (base ? base[0] : 0) + scale·Σ (y − target)², the
 * squares summed per column tile in the last layer's epilogue while y is in registers and the tile sums added in tile order by a one-wave
 * kernel — lde_chain_forward[_save] followed by lde_mse_forward[_add] without the latter's pass over x and x̂ (the sum's order differs from
 * lde_mse_forward's: equal to rounding, bit-reproducible). scratch: lde_chain_mse_scratch_floats(c, N) floats. saved may be NULL. */
int64_t lde_chain_mse_scratch_floats(const lde_chain* c, int64_t N);
int  lde_chain_forward_save_mse(lde_chain* c, const float* x, int64_t N, float* y, float* saved, const float* target, float scale,
                                const float* base, float* out, float* scratch, void* stream);
/* The pullback of a chain whose output y goes into base + scale·Σ (y − target)² (the reconstructor under reconstruction_loss
 * [REF examples/pendulum_friction-less/model_train.jl:225-238]): the output gradient is 2·(g·scale)·(y − target), g = *g_dev the scalar
 * loss's cotangent, formed where the kernel reads it — what lde_mse_backward writes as an [out×N] array for lde_chain_backward_saved to read
 * back (the largest array of a GOKU step: one launch and 3 × 4·out·N bytes less). dy_more (or NULL): a further cotangent of y, added.
 * Same values as the two calls, bit for bit. */
int  lde_chain_backward_saved_mse(lde_chain* c, const float* x, const float* y, const float* target, const float* g_dev, float scale,
                                  const float* dy_more, const float* saved, int64_t N, float* dx, float* dW, void* stream);
/* The same pair without the x̂ round trip (LDE_DTYPE_BF16 chains; LDE_ERR_UNSUPPORTED otherwise, or when the output width is not a
 * multiple of 8): the forward launch's last epilogue — where x̂ and the target are in registers for the squares anyway — also leaves
 * δ_L′ = 2·scale·(y − target)·act′(y) as the pullback's bf16 δ matrix and stores y only when asked (y may be NULL); the pullback starts
 * from that matrix — it reads neither y nor the target: of a GOKU step's reconstructor [REF src/models/GOKU.jl:252-269] 40 MB written and
 * 80 MB read less — and multiplies dx and dW by g = *g_dev at the end (δ is linear in g; with g = 1, the loss being the objective
 * [REF model_train.jl:225-238], the results are those of lde_chain_backward_saved_mse bit for bit). The δ matrix lives in the chain's
 * workspace: the pullback must follow the forward call of the same N with no other pullback of this chain in between. */
int  lde_chain_forward_save_mse_delta(lde_chain* c, const float* x, int64_t N, float* y, float* saved, const float* target, float scale,
                                      const float* base, float* out, float* scratch, void* stream);
int  lde_chain_backward_saved_delta(lde_chain* c, const float* x, const float* g_dev, const float* saved, int64_t N, float* dx, float* dW,
                                    void* stream);
/* 1 when the δ matrix in the chain's workspace is the one lde_chain_forward_save_mse_delta staged for THIS forward call — identified by
 * its `saved` buffer and N — and no later forward / pullback of the chain has overwritten or consumed it; 0 otherwise. A caller that may
 * have several forwards of one chain in flight (an AD tape) asks before lde_chain_backward_saved_delta — which refuses with
 * LDE_ERR_INVALID_ARG rather than use another call's δ — and takes lde_chain_backward_saved_mse instead (same values; needs y). */
int  lde_chain_delta_is_staged(const lde_chain* c, const float* saved, int64_t N);
/* How the pullbacks deliver the weight gradient: on = 1 (default) dW += gradient, like lde_adjoint; on = 0: dW = gradient — every
 * entry of dW is written exactly once, so a caller that wants the plain gradient needs no zero fill (one launch less). */
int  lde_chain_set_accumulate(lde_chain* c, int on);
/* Arithmetic of the chain's matrix products — BASELINE.json configs[4]: "mixed fp32 solve / bf16 encoder-decoder". LDE_DTYPE_F32
 * (default): exact f32 products (v_mfma_f32_16x16x4_f32). LDE_DTYPE_BF16: both operands of every product (forward, input gradient,
 * weight gradient) are rounded to bfloat16 (round-to-nearest-even) and multiplied on the bf16 matrix cores with f32 accumulation;
 * weights, biases, activations, saved activations and gradients stay f32 in memory (f32 master weights), so a caller switches modes
 * without re-uploading anything. The solve (lde_forward / lde_adjoint) is f32 in either mode. */
enum lde_dtype { LDE_DTYPE_F32 = 0, LDE_DTYPE_BF16 = 1 };
int  lde_chain_set_dtype(lde_chain* c, int dtype);
/* Kernel-choice knobs of the parity tests (a production host never sets them; the library reads no environment variable):
 * "gx" = 0: never the panel-free layout of a wide first layer (LDE_ERR_UNSUPPORTED when it is the only one that fits);
 * "group" = 0: this chain does not take part in merged grouped calls (its stages then run as launches of their own).
 * A scheduling choice, not a parity knob: "async_dw" = 0: with a weight-gradient stream set (lde_set_dw_stream) this chain's weight-gradient
 * kernels stay on the caller's stream all the same — the stream pays where something runs beside them (the reconstructor's, beside the
 * rest of the pullback) and costs a fork and a join where nothing does (the feature extractor's: the last kernel of the pullback). */
int  lde_chain_set_option(lde_chain* c, const char* key, double value);
const char* lde_chain_last_error(const lde_chain* c);

/* ======================================================================================================
 * Recurrent pattern extractor — scope row f-2: what runs under
 *
 *     pe_out = apply_pattern_extractor(encoder, fe_out)     [REF src/models/GOKU.jl:32-51], [REF src/models/LatentODE.jl:24-33]
 *
 * i.e. a stack `Chain(RNN(in,h,relu), RNN(h,h,relu))` or `Chain(LSTM(in,h), LSTM(h,h))` [REF src/models/GOKU.jl:229-238]
 * applied to the T time frames of fe_out [in×B×T] in forward or REVERSED order, keeping the LAST output
 * (`[pe(x) for x in fe_out_rev][end]`), with the hidden state starting from the cell's `state0` on every call
 * (`Flux.reset!`). Cells follow Flux 0.13.6 (un-vendored [REF Manifest.toml:452]):
 *   RNNCell : h' = act.(Wi*x .+ Wh*h .+ b)
 *   LSTMCell: g = Wi*x .+ Wh*h .+ b;  input, forget, cell, output = σ(g[1:o]), σ(g[o+1:2o]), tanh(g[2o+1:3o]), σ(g[3o+1:4o]);
 *             c' = forget.*c .+ input.*cell;  h' = output.*tanh.(c')
 * Flat weights = Flux.destructure order per cell: vec(Wi) [G·h × in], vec(Wh) [G·h × h] (column-major), b [G·h],
 * state0 (h0 [h]; LSTM: then c0 [h]) — the initial state is a trainable parameter in this Flux version.
 *  - x [in×B×T], y [h_last×B], dy, dx are DEVICE pointers in the reference's column-major layout.
 *  - lde_rnn_backward recomputes the forward sweep (nothing is kept from lde_rnn_forward).
 */
#define LDE_RNN_MAX_LAYERS 4

enum lde_cell_kind { LDE_CELL_RNN_RELU = 0, LDE_CELL_RNN_TANH = 1, LDE_CELL_LSTM = 2 };

typedef struct lde_rnn_desc {
  int32_t abi_version;                       /* = LDE_ABI_VERSION */
  int32_t cell;                              /* lde_cell_kind */
  int32_t n_layers;                          /* stacked cells, 1..LDE_RNN_MAX_LAYERS */
  int32_t sizes[LDE_RNN_MAX_LAYERS + 1];     /* [in, h1, ..., hL]; gate rows G·h ≤ 64 (LSTM: h ≤ 16, RNN: h ≤ 64), in ≤ 256 */
  int32_t reverse;                           /* 1: feed the frames T, T-1, ..., 1 (reverse(fe_out)) */
} lde_rnn_desc;

typedef struct lde_rnn lde_rnn;

int64_t lde_rnn_num_weights(const lde_rnn_desc* desc);
int  lde_rnn_create(const lde_rnn_desc* desc, lde_rnn** out);
void lde_rnn_destroy(lde_rnn* r);
int  lde_rnn_set_weights(lde_rnn* r, const float* flat_host, int64_t n);
int  lde_rnn_set_weights_device(lde_rnn* r, const float* flat_dev, int64_t n, void* stream);
int  lde_rnn_reserve(lde_rnn* r, int B, int T);
/* y[hL×B] = output of the top cell after the last frame. */
int  lde_rnn_forward(lde_rnn* r, const float* x, int T, int B, float* y, void* stream);
/* Back-propagation through time from dy[hL×B]: dx[in×B×T] (written; may be NULL), dW[n_weights] ACCUMULATED (+=). */
int  lde_rnn_backward(lde_rnn* r, const float* x, const float* dy, int T, int B, float* dx, float* dW, void* stream);
/* The same pullback in two calls, for a caller that overlaps several stacks (the GOKU pattern extractor is three independent
 * stacks on the same frames [REF src/models/GOKU.jl:32-51]): _dx runs the sweep — dx written, the weight gradient's panels staged
 * in the handle — and _dw turns the staged panels into dW (+= or =, as lde_rnn_set_accumulate says) exactly once. Issue every
 * stack's _dx (each on its stream) before the first _dw and the long sweeps start together. lde_rnn_backward = _dx then _dw. */
int  lde_rnn_backward_dx(lde_rnn* r, const float* x, const float* dy, int T, int B, float* dx, void* stream);
int  lde_rnn_backward_dw(lde_rnn* r, float* dW, void* stream);
/* Training variant of lde_rnn_forward: the sweep also leaves its per-step records and the weight gradient's input panels in the handle's
 * workspace; the next lde_rnn_backward[_dx] with the SAME (x, T, B) back-propagates from them instead of repeating the sweep. Same
 * results as the plain pair. Anything that changes the weights in between drops the records (the pullback then sweeps again). */
int  lde_rnn_forward_train(lde_rnn* r, const float* x, int T, int B, float* y, void* stream);
/* Several stacks on the same frames in one call (the GOKU encoder's three pattern extractors [REF src/models/GOKU.jl:32-51]): with
 * n ≤ 3 stacks of the default shape every stage — the sweep, the weight-gradient products of all (stack, cell) pairs, their fixed-order
 * sums, the initial-state sums — is ONE launch; anything else runs the stacks one after the other. Per stack the same kernels on the
 * same arguments as lde_rnn_forward / lde_rnn_backward: results equal bit for bit. dxs may be NULL or hold NULL entries. */
int  lde_rnn_group_forward(int n, lde_rnn* const* stacks, const float* const* xs, int T, int B, float* const* ys, void* stream);
int  lde_rnn_group_forward_train(int n, lde_rnn* const* stacks, const float* const* xs, int T, int B, float* const* ys, void* stream);
int  lde_rnn_group_backward(int n, lde_rnn* const* stacks, const float* const* xs, const float* const* dys, int T, int B,
                            float* const* dxs, float* const* dWs, void* stream);
/* The grouped calls with each stack's output (ys[i]) and output gradient (dys[i]) given as a COLUMN BLOCK of a wider array: row b of
 * stack i starts at ys[i] + b·ldys[i] (ldys[i] ≥ the stack's output width; NULL: dense rows) — the θ branch's vcat(pe_forward,
 * pe_backward) [REF src/models/GOKU.jl:47] is then written in place by the two stacks, and its gradient read in place. dys2 (NULL, or
 * NULL entries: absent): a second source of the output gradient with the same row stride, the gradient used is dys[i] + dys2[i] — the
 * (μ, log σ²) heads of apply_latent_in [REF src/models/GOKU.jl:61-72] both read a stack's output. train != 0: the sweep keeps its records
 * (lde_rnn_forward_train). Otherwise as lde_rnn_group_forward[_train] / lde_rnn_group_backward, bit for bit. */
int  lde_rnn_group_forward_ld(int n, lde_rnn* const* stacks, const float* const* xs, int T, int B, float* const* ys, const int* ldys,
                              int train, void* stream);
int  lde_rnn_group_backward_ld(int n, lde_rnn* const* stacks, const float* const* xs, const float* const* dys, const float* const* dys2,
                               const int* lddys, int T, int B, float* const* dxs, float* const* dWs, void* stream);
int  lde_rnn_set_accumulate(lde_rnn* r, int on);   /* as lde_chain_set_accumulate */
/* Kernel-choice knobs of the parity tests: "generic" = 1 the run-time-shaped kernel for every stack, "regw" = 0 weights in LDS instead
 * of registers, "pipe" = 0 one wave per stack instead of one per cell. Same results to round-off; the library reads no environment variable. */
int  lde_rnn_set_option(lde_rnn* r, const char* key, double value);
const char* lde_rnn_last_error(const lde_rnn* r);

/* After an optimiser step: hand the new flat weights of MANY modules to their handles in ONE launch — what n calls of
 * lde_chain_set_weights_device / lde_rnn_set_weights_device do, one launch each (the reference has no counterpart: Flux layers
 * read their arrays in place, `Flux.update!` [REF examples/pendulum_friction-less/model_train.jl:190-192]; here a chain keeps
 * MFMA-fragment-ordered copies of W and Wᵀ that must follow the parameters). kinds[m] ∈ lde_module_kind, handles[m] the
 * lde_chain* / lde_rnn*, flat_dev[m] its weights in Flux.destructure order (device). The handles may then be used with those
 * weights until they change again. */
enum lde_module_kind { LDE_MODULE_CHAIN = 0, LDE_MODULE_RNN = 1 };
int lde_refresh_weights(int n, const int* kinds, void* const* handles, const float* const* flat_dev, void* stream);

/* ====================================================================================================================
 * The variational sample and the loss terms (scope row f-3): what the example script computes between encoder and
 * decoder and around the model's output. Stateless elementwise / reduction kernels; all pointers are device memory.
 *
 *   sample:  l̃ = μ + ε·exp(logσ²/2), ε ~ N(0,1) supplied by the caller   [REF src/models/GOKU.jl:155-163],
 *                                                                        [REF src/models/LatentODE.jl:82-89]
 *   kl:      out = scale · Σ_i (exp(logσ²_i) + μ_i² − logσ²_i − 1)/2      [REF src/utils/utils.jl:15-49]  (scale = 1/B)
 *   mse:     out = scale · Σ_i (x_i − x̂_i)²                              [REF examples/pendulum_friction-less/
 *            (sum(mean(·, dims=(2,3))) ⇒ scale = 1/(B·T))                       model_train.jl:225-238]
 *
 * Reductions run in a fixed order (per-workgroup partial sums in `scratch`, combined by index): results are
 * bit-reproducible. `scratch` must hold LDE_LOSS_SCRATCH_FLOATS floats and must not be shared by concurrent calls.
 * The pullbacks take the cotangent of the scalar output as a DEVICE pointer (no host synchronisation).
 * ==================================================================================================================== */
#define LDE_LOSS_SCRATCH_FLOATS 1024

/* out[0..n) ~ N(0, 1): the ε of `sample` (the reference draws it with `randn` [REF src/models/GOKU.jl:155-163]). Philox4x32-10 keyed by
 * `seed`; block i of four consecutive outputs is Box–Muller of the words of counter (i, call, offset + *epoch_dev) — u = (w >> 8 + ½)/2²⁴,
 * (√(−2 ln u₁)·cos 2πu₂, √(−2 ln u₁)·sin 2πu₂) per pair of words. `epoch_dev` (NULL: 0) is a device counter read by the kernel: inside
 * a captured training step it is the optimiser's step count, so every replay draws fresh noise with no host-side generator state.
 * `call` separates the draws of one step, `offset` the steps of an eager loop. raw_words (NULL, or n words): the Philox output itself
 * (what the known-answer vectors of the generator are stated in; tests). */
int lde_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, uint32_t call, const int64_t* epoch_dev, uint32_t* raw_words, void* stream);
/* The variational sample of the GOKU tuple (z₀, θ) [REF src/models/GOKU.jl:155-163] in one launch each way: forward = lde_randn(eps_a),
 * lde_sample_kl_forward(part a, base), lde_randn(eps_b), lde_sample_kl_forward(part b, base = the first total) — the same device code in
 * the same order, bit for bit; ε is written to eps_a / eps_b for the pullback. Parts of 1 … 8192 entries (LDE_ERR_UNSUPPORTED otherwise:
 * make the separate calls). scratch: 2 floats. backward = the two lde_sample_kl_backward calls. */
int lde_sample_kl_pair_forward(const float* mu_a, const float* logvar_a, int64_t n_a, float scale_a, const float* mu_b, const float* logvar_b,
                               int64_t n_b, float scale_b, const float* base, uint64_t seed, uint64_t offset_a, uint64_t offset_b, uint32_t call_a,
                               uint32_t call_b, const int64_t* epoch_dev, float* eps_a, float* eps_b, float* l_a, float* l_b, float* out,
                               float* scratch, void* stream);
int lde_sample_kl_pair_backward(const float* mu_a, const float* logvar_a, const float* eps_a, const float* dl_a, int64_t n_a, float scale_a,
                                const float* mu_b, const float* logvar_b, const float* eps_b, const float* dl_b, int64_t n_b, float scale_b,
                                const float* dout, float* dmu_a, float* dlogvar_a, float* dmu_b, float* dlogvar_b, void* stream);
int lde_sample_forward(const float* mu, const float* logvar, const float* eps, int64_t n, float* l, void* stream);
/* dμ = dl (not written: it is the input); dlogvar_i = dl_i · ε_i · exp(logσ²_i/2) / 2 */
int lde_sample_backward(const float* logvar, const float* eps, const float* dl, int64_t n, float* dlogvar, void* stream);
int lde_kl_forward(const float* mu, const float* logvar, int64_t n, float scale, float* out, float* scratch, void* stream);
/* dμ_i = g·scale·μ_i,  dlogvar_i = g·scale·(exp(logσ²_i) − 1)/2,  g = *dout */
int lde_kl_backward(const float* mu, const float* logvar, int64_t n, float scale, const float* dout, float* dmu,
                    float* dlogvar, void* stream);
int lde_mse_forward(const float* x, const float* xhat, int64_t n, float scale, float* out, float* scratch, void* stream);
/* dx̂_i = g·scale·2·(x̂_i − x_i),  g = *dout */
int lde_mse_backward(const float* x, const float* xhat, int64_t n, float scale, const float* dout, float* dxhat, void* stream);

/* The same terms as `loss_batch` composes them  [REF examples/pendulum_friction-less/model_train.jl:225-238]:
 * `reconstruction_loss + β·kl_loss` on a model output that was sampled from (μ, logσ²)  [REF src/models/LatentDiffEqModel.jl:31].
 * The sample and its KL term read the same (μ, logσ²): one pass computes both, one pass back delivers the SUM of both cotangents
 * (the reference's Zygote adds them as separate broadcasts), and the scalar additions of the loss expression ride on `base`:
 *   sample_kl forward:  l_i = μ_i + ε_i·exp(logσ²_i/2);   out = (base ? *base : 0) + scale·Σ_i kl_i          (scale = β/B)
 *   sample_kl backward: dμ_i = dl_i + g·scale·μ_i;   dlogσ²_i = dl_i·ε_i·exp(logσ²_i/2)/2 + g·scale·(exp(logσ²_i) − 1)/2,  g = *dout
 *   mse_forward_add:    out = (base ? *base : 0) + scale·Σ_i (x_i − x̂_i)²       (pullback: lde_mse_backward; d base = dout)
 * Same per-element formulas and the same fixed-order reductions as the separate entry points above. */
int lde_sample_kl_forward(const float* mu, const float* logvar, const float* eps, int64_t n, float scale, const float* base,
                          float* l, float* out, float* scratch, void* stream);
int lde_sample_kl_backward(const float* mu, const float* logvar, const float* eps, const float* dl, const float* dout, float scale,
                           int64_t n, float* dmu, float* dlogvar, void* stream);
int lde_mse_forward_add(const float* x, const float* xhat, int64_t n, float scale, const float* base, float* out, float* scratch,
                        void* stream);

/* Where the chain / recurrent pullbacks enqueue their WEIGHT-GRADIENT kernels. Default (NULL): on the caller's stream, after
 * the pullback kernel — `dW` is complete in stream order like every other output. With a stream set, lde_chain_backward[_saved]
 * and lde_rnn_backward enqueue only the input-gradient kernel on the caller's stream and the kernels that produce `dW` on this
 * one (ordered after it by an event): the next module's pullback does not wait for them — in the reference's Zygote pullback
 * they are independent closures too [REF examples/pendulum_friction-less/model_train.jl:186-189]. `dW` is then complete only
 * after lde_join_dw(s), which makes stream `s` wait (on the device; the host does not block) for every weight-gradient kernel
 * enqueued so far: call it before the optimiser / all-reduce reads the gradients. A handle's next pullback waits for its own
 * previous weight-gradient kernels by itself (its workspace is reused). Process-wide; lde_adjoint's dW is not affected. */
int lde_set_dw_stream(void* stream);
int lde_join_dw(void* stream);

/* The parameter update of the training step: `Flux.Optimise.update!(opt, ps, gs)` with `opt = ADAMW(η, (β₁, β₂), decay)`
 * [REF examples/pendulum_friction-less/model_train.jl:138, :190-192] — in the pinned Flux 0.13.6 [REF Manifest.toml:452]
 * `Optimiser(ADAM(η, β), WeightDecay(decay))`: per array  m ← β₁m + (1−β₁)g;  v ← β₂v + (1−β₂)g²;
 * Δ = m/(1−β₁ᵗ) / (√(v/(1−β₂ᵗ)) + ε)·η + decay·x;  x ← x − Δ  (the decay is not scaled by η). ONE launch for all n arrays
 * (`t`: a HOST array of device pointers; m and v are the caller's state arrays, zero before the first step); `step` = t ≥ 1. */
typedef struct lde_adam_tensor {
  float* p;          /* parameters, updated in place */
  const float* g;    /* their gradient */
  float* m;          /* first moment  (state) */
  float* v;          /* second moment (state) */
  int64_t n;         /* floats */
} lde_adam_tensor;
int lde_adamw_flux_step(int n, const lde_adam_tensor* t, float lr, float beta1, float beta2, float eps, float decay, int64_t step,
                        void* stream);
/* The same update with the step count t in DEVICE memory (*step_dev is advanced by one, then used): what a training step captured in a
 * hipGraph needs — a graph replays the same kernel arguments every time, so the bias corrections cannot travel in them
 * [REF examples/pendulum_friction-less/model_train.jl:186-204: the loop the graph replaces]. */
int lde_adamw_flux_step_dev(int n, const lde_adam_tensor* t, float lr, float beta1, float beta2, float eps, float decay,
                            int64_t* step_dev, void* stream);

/* ====================================================================================================================
 * The one collective of the path. The reference's only parallelism is `EnsembleThreads()` over the
 * trajectories of one batch  [REF src/models/GOKU.jl:121]; here the batch shards by trajectory over one process per GPU
 * with NO collective inside lde_forward / lde_adjoint, and the gradients of the parameters every rank shares — the
 * RHS-MLP `dW` of lde_adjoint, the chains' and recurrent stacks' `dW` — are summed once per optimiser step
 * [REF examples/pendulum_friction-less/model_train.jl:186-204 is the step]. These entry points give a `ccall` host that
 * exchange through the same boundary: an in-place f32 sum over RCCL (xGMI inside a node).
 *
 *   rank 0:      lde_comm_unique_id(id)            → hand the 128 bytes to every rank by any host-side means
 *                                                     (MPI.bcast, a file, Distributed.jl's remotecall — not the library's concern)
 *   every rank:  hipSetDevice(local_rank); lde_comm_init(&c, nranks, rank, id)        (collective)
 *   every step:  lde_comm_allreduce_f32(c, flat_gradient, n, stream)                  (asynchronous on `stream`)
 *
 * The loss is a mean over the GLOBAL batch [REF model_train.jl:232]: fold 1/B_global into the cotangent and the plain sum is
 * exact. librccl is bound at run time (a copy already in the process is preferred; LDE_RCCL_PATH overrides); without it these
 * calls return LDE_ERR_UNSUPPORTED and everything else in the library still works.
 * ==================================================================================================================== */
#define LDE_COMM_ID_BYTES 128

typedef struct lde_comm lde_comm;

int  lde_comm_unique_id(char* id /* [LDE_COMM_ID_BYTES], host, written */);
int  lde_comm_init(lde_comm** out, int nranks, int rank, const char* id /* [LDE_COMM_ID_BYTES] */);
/* buf[n] (device) ← Σ over ranks of buf[n], in place, asynchronous on `stream`; every rank passes the same n. */
int  lde_comm_allreduce_f32(lde_comm* c, float* buf, int64_t n, void* stream);
int  lde_comm_nranks(const lde_comm* c);
int  lde_comm_rank(const lde_comm* c);
void lde_comm_destroy(lde_comm* c);
/* c == NULL: the last error of lde_comm_unique_id / lde_comm_init in this process. */
const char* lde_comm_last_error(const lde_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* LDE_H */

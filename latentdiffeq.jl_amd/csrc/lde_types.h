// lde_types.h — the plain-C++ structs the host side of the C ABI and the kernels share: step records, kernel-choice knobs, the option
// block handed to every kernel. No HIP in here: csrc/lde_host.h (the C ABI's argument / workspace / record logic) includes this file and
// is compiled by an ordinary host compiler too — under AddressSanitizer + UBSan in tests/host_logic_driver.cpp.
#pragma once
#include <stdint.h>

#include "../../include/lde.h"

namespace lde {

// A step record in device memory (include/lde.h: lde_set_step_record): the accepted steps of each of `nseq` step sequences (one per
// trajectory, or one for a coupled solve) — written by a forward solve (LDE_SENSE_DISCRETE: start time, step size and start state of
// every accepted step, what the discrete adjoint sweeps in reverse), read by that adjoint; the continuous adjoint writes the magnitudes of
// its reverse-time steps into one when step tracing is on (y == nullptr). n == nullptr: no record.
struct StepRec {
  int32_t* n;    // [nseq] accepted steps; a count > cap means the record is incomplete
  double* t;     // [cap][nseq] start time of step n
  double* dt;    // [cap][nseq] its size (the state advances by (float)dt)
  float* y;      // [cap][B][D'] state at the start of step n
  int cap, nseq;
};

// Kernel-choice knobs of a handle (lde_set_option; tests force a family / a threshold through them — formerly LDE_* environment variables,
// which a library behind a `ccall` host must not read). Defaults = the measured choices.
struct PendTune {
  int ws = 1;                 // "pend_ws": k_pend_forward_ws for B ≤ 16384
  int tl_max_b = 2048;        // "pend_tl_max_b": k_pend_forward_tl up to this batch (solves that write no step record; 13.5 against k_pend_forward_ws's 15.9 µs at 2 048, 20.9 against 15.8 at 4 096)
  int sh_max_b = -1;          // "pend_sh_max_b": k_pend_forward_sh / k_pend_forward_lp (a trajectory per workgroup) up to this batch; −1: the measured thresholds (lp 1 024; sh 768 with a step record, 256 without)
  int lp = 1;                 // "pend_lp": frictionless Tsit5 adaptive solves of that shape run k_pend_forward_lp (lane pairs, Nyström form); 0: k_pend_forward_sh
  int lb_ring = 16;           // "pend_lb": rows of the large-batch row ring (8 / 16 / 32; 0: off)
  int lb_min_b = 1 << 17;     // "pend_lb_min_b": the large-batch form from this batch on
  int lb_hold = -1;           // "pend_lb_hold": its hold margin (−1: half the ring)
  int disc_tp_max_b = 16384;  // "pend_disc_tp_max_b": LDE_SENSE_DISCRETE pullback with a wave per trajectory (k_pend_adjoint_disc_tp) up to this batch
};
struct MlpTune {
  int mlp64 = 1, mlpv = 1, mlpw = 1, mlp4 = 1;   // "mlp64", "mlpv", "mlpw", "mlp4": 0 switches the family off
  int mlpb = 1;               // "mlpb": 0 off (k_mlpw instead: the parity reference), 2 also the networks of ≤ 128 units
  int mlp4_maxw = 64;         // "mlp4_maxw": widest layer k_mlp4_adjoint takes
  int stage_slots = 0;        // "mlp_stage_slots": staging slots per workgroup (0: automatic)
  int peer_spin_k = 0;        // "peer_spin_k": lde_set_global_sum_peers' cross-rank wait gives up after this many × 1024 polls (0: 8192 ≈ 10 s)
};

// Options handed to every kernel by value (mirrors the `kwargs...` splat into solve()).
struct KOpts {
  float abstol, reltol;
  float beta1, beta2;
  float inv_gamma;   // 1/γ
  float q_lo;        // 1/qmax : lower clamp of q
  float q_hi;        // 1/qmin : upper clamp of q
  float qmin;
  double dtmin;
  double dt_fixed;   // fixed step (adaptive=0) or user initial dt (adaptive=1, >0)
  long long maxiters;
  int adaptive;
  int checkpoint;    // adjoint: reset z to the saved ẑ(t_j) at every save time
  int T, B;
  double t_first, t_last;   // ts[0], ts[T−1] (the host has the grid): a kernel need not load them before its first step
  int lb_hold;              // large-batch forward: a lane this close to the end of the row ring waits for its wave (0: never)
  StepRec rec;              // forward: the record to write; discrete adjoint: the record to read; continuous adjoint: the trace to write
  int dw_overwrite;         // adjoint: dW is written, not accumulated (option "adjoint_overwrite")
};


}  // namespace lde

// host_logic_driver.cpp — the C ABI's host-side logic (latentdiffeq.jl_amd/csrc/lde_host.h: what lde_api.hip does BEFORE it touches the
// device) under AddressSanitizer + UndefinedBehaviorSanitizer, driven with well-formed and hostile inputs. Built and run by
// tests/test_sanitizers.py (g++; no HIP, no GPU). A C ABI's arguments come from another language's runtime: every field of a problem
// description is an int or a double somebody else filled in.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../latentdiffeq.jl_amd/csrc/lde_host.h"

using namespace lde_host;

static unsigned long long rs = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static int hostile_int() {
  static const int v[] = {0, 1, -1, 2, 6, 7, 8, 64, 255, 256, 1024, 1025, 1 << 20, std::numeric_limits<int>::max(), std::numeric_limits<int>::min(), -7};
  return (rnd() & 3) ? v[rnd() % (sizeof(v) / sizeof(v[0]))] : (int)rnd();
}
static double hostile_double() {
  static const double v[] = {0.0, -0.0, 1e-6, 1e-3, 1.0, -1.0, 1e300, -1e300, 1e-320, std::numeric_limits<double>::infinity(),
                             -std::numeric_limits<double>::infinity(), std::numeric_limits<double>::quiet_NaN(), 0.05, 0.2, 10.0, 0.9};
  return v[rnd() % (sizeof(v) / sizeof(v[0]))];
}

static lde_problem_desc good() {
  lde_problem_desc d;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = LDE_ABI_VERSION; d.rhs_kind = LDE_RHS_PENDULUM; d.state_dim = 2; d.param_dim = 1; d.solver = LDE_SOLVER_TSIT5;
  d.sensealg = LDE_SENSE_DISCRETE; d.adaptive = 1; d.maxiters = 100000; d.abstol = 1e-6; d.reltol = 1e-3; d.qmin = 0.2; d.qmax = 10; d.gamma = 0.9;
  d.beta1 = 0.14; d.beta2 = 0.08;
  return d;
}

int main() {
  int n_ok = 0, n_bad = 0;
  std::string why;
  // 1. well-formed descriptions of every family validate; their derived quantities are what include/lde.h says
  {
    lde_problem_desc d = good();
    assert(validate(&d, &why) == LDE_OK && num_weights(&d) == 0);
    d.rhs_kind = LDE_RHS_MLP; d.state_dim = 16; d.param_dim = 0; d.n_layers = 3; d.batching = LDE_BATCH_COUPLED;
    const int s[4] = {16, 200, 200, 16};
    for (int i = 0; i < 4; i++) d.layer_sizes[i] = s[i];
    assert(validate(&d, &why) == LDE_OK && num_weights(&d) == 46816);            // the reference's default NODE [REF nODE.jl:11-14]
    assert(rec_nseq(d, 64) == 1 && rec_capacity(d, 0, 50, 0) == 200 && rec_capacity(d, 0, 50, 1) == 800 && rec_capacity(d, 7, 50, 0) == 7);
    d.maxiters = 5;
    assert(rec_capacity(d, 0, 50, 0) == 5);
    d.maxiters = 100000; d.solver = LDE_SOLVER_RK4;
    assert(validate(&d, &why) == LDE_ERR_UNSUPPORTED);                           // adaptive RK4: declared out of scope
    d.adaptive = 0; d.dt = 0.05;
    assert(validate(&d, &why) == LDE_OK);
    std::vector<double> ts(50);
    for (int j = 0; j < 50; j++) ts[j] = 0.05 * j;
    assert(grid_ok(ts.data(), 50) && fixed_step_count(d, ts.data(), 50) == 49);
    d.dt = 1e-30;
    assert(fixed_step_count(d, ts.data(), 50) == d.maxiters);                    // capped, no overflow
    KOpts o = make_opts(d, ts.data(), 50, 64);
    assert(o.T == 50 && o.B == 64 && o.t_first == 0.0 && o.t_last == ts[49] && o.checkpoint == 1 && o.dtmin > 0);
    // the record layout: consecutive, aligned, inside rec_bytes
    d = good();
    for (int B : {1, 3, 64, 257}) for (int cap : {1, 7, 64, 200}) {
      const size_t bytes = rec_bytes(d, B, cap, true);
      std::vector<unsigned char> buf(bytes + 256);
      unsigned char* base = (unsigned char*)(((uintptr_t)buf.data() + 255) & ~(uintptr_t)255);
      StepRec r = rec_view(d, base, B, cap, true);
      assert((unsigned char*)r.n == base && ((uintptr_t)r.t & 255) == 0 && ((uintptr_t)r.dt & 255) == 0 && ((uintptr_t)r.y & 255) == 0);
      assert((unsigned char*)(r.y + (size_t)cap * B * 2) <= base + bytes && r.cap == cap && r.nseq == B);
      // touch the last element of every array: an overrun is an ASan report
      r.n[B - 1] = 1; r.t[(size_t)cap * B - 1] = 1.0; r.dt[(size_t)cap * B - 1] = 1.0; r.y[(size_t)cap * B * 2 - 1] = 1.f;
    }
  }
  // 2. hostile descriptions: never a crash, never an out-of-bounds index, always a status
  for (int it = 0; it < 200000; it++) {
    lde_problem_desc d = good();
    const int nmut = 1 + (int)(rnd() % 5);
    for (int m = 0; m < nmut; m++) {
      switch (rnd() % 22) {
        case 0: d.abi_version = hostile_int(); break;
        case 1: d.rhs_kind = hostile_int(); break;
        case 2: d.state_dim = hostile_int(); break;
        case 3: d.param_dim = hostile_int(); break;
        case 4: d.augment_dim = hostile_int(); break;
        case 5: d.n_layers = hostile_int(); break;
        case 6: d.layer_sizes[rnd() % (LDE_MAX_LAYERS + 1)] = hostile_int(); break;
        case 7: d.activation = hostile_int(); break;
        case 8: d.solver = hostile_int(); break;
        case 9: d.batching = hostile_int(); break;
        case 10: d.sensealg = hostile_int(); break;
        case 11: d.adaptive = hostile_int(); break;
        case 12: d.maxiters = (int64_t)rnd() * ((rnd() & 1) ? 1 : -1); break;
        case 13: d.dt = hostile_double(); break;
        case 14: d.abstol = hostile_double(); break;
        case 15: d.reltol = hostile_double(); break;
        case 16: d.dtmin = hostile_double(); break;
        case 17: d.qmin = hostile_double(); break;
        case 18: d.qmax = hostile_double(); break;
        case 19: d.gamma = hostile_double(); break;
        case 20: d.beta1 = hostile_double(); break;
        default: d.beta2 = hostile_double(); break;
      }
    }
    const int rc = validate(&d, &why);
    (void)num_weights(&d);                                       // (defined for ANY description: n_layers is clamped to the struct's capacity)
    if (rc == LDE_OK) {
      n_ok++;
      const int T = 1 + (int)(rnd() % 300), B = 1 + (int)(rnd() % 5000);
      std::vector<double> ts(T);
      double t = hostile_double();
      if (!std::isfinite(t) || std::fabs(t) > 1e6) t = 0;   // (a grid that stays strictly increasing in f64)
      for (int j = 0; j < T; j++) { ts[j] = t; t += 1e-3 + (double)(rnd() % 1000) * 1e-4; }
      assert(grid_ok(ts.data(), T));
      KOpts o = make_opts(d, ts.data(), T, B);
      assert(o.T == T && o.B == B);
      (void)fixed_step_count(d, ts.data(), T);
      const int cap = rec_capacity(d, (int)(rnd() % 3) ? 0 : 1 + (int)(rnd() % 4096), T, (int)(rnd() & 1));
      assert(cap >= 1);
      assert(rec_bytes(d, B, cap, true) >= (size_t)rec_nseq(d, B) * 4);
    } else {
      n_bad++;
      assert(rc == LDE_ERR_INVALID_ARG || rc == LDE_ERR_UNSUPPORTED);
    }
  }
  assert(validate(nullptr, &why) == LDE_ERR_INVALID_ARG && num_weights(nullptr) == 0);
  // 3. save-time grids
  {
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double a[3] = {0, 1, 1}, b[3] = {0, nan, 2}, c[2] = {0, inf}, e[1] = {5};
    assert(!grid_ok(a, 3) && !grid_ok(b, 3) && !grid_ok(c, 2) && grid_ok(e, 1));
  }
  // 4. which forward mapping serves a solve of the analytic right-hand sides (csrc/lde_pendulum.hip's launch code switches on this)
  {
    const lde::PendTune def;   // the measured thresholds
    const int P = LDE_RHS_PENDULUM, F = LDE_RHS_PENDULUM_FRICTION, TS = LDE_SOLVER_TSIT5, RK = LDE_SOLVER_RK4, LMAX = 6000;
    auto m = [&](int kind, int solver, bool ad, bool rec, int B, int T, const lde::PendTune& tn) { return pend_forward_mapping(kind, solver, ad, rec, B, T, tn, LMAX); };
    // the metric's shape (frictionless, Tsit5, adaptive): lane pairs up to 1 024, four dense-output waves up to 512 — recording or not
    for (int rec = 0; rec < 2; rec++) {
      assert(m(P, TS, true, rec, 1, 50, def) == PEND_FWD_LP4 && m(P, TS, true, rec, 256, 50, def) == PEND_FWD_LP4 && m(P, TS, true, rec, 512, 50, def) == PEND_FWD_LP4);
      assert(m(P, TS, true, rec, 513, 50, def) == PEND_FWD_LP3 && m(P, TS, true, rec, 1024, 50, def) == PEND_FWD_LP3);
    }
    assert(m(P, TS, true, false, 1025, 50, def) == PEND_FWD_TL && m(P, TS, true, false, 2048, 50, def) == PEND_FWD_TL && m(P, TS, true, false, 2049, 50, def) == PEND_FWD_WS);
    assert(m(P, TS, true, true, 1025, 50, def) == PEND_FWD_WS && m(P, TS, true, true, 16384, 50, def) == PEND_FWD_WS);
    // every other solve of a trajectory per workgroup: B ≤ 768 when it writes step records, ≤ 256 when it does not
    assert(m(F, TS, true, true, 768, 50, def) == PEND_FWD_SH && m(F, TS, true, true, 769, 50, def) == PEND_FWD_WS);
    assert(m(F, TS, true, false, 256, 50, def) == PEND_FWD_SH && m(F, TS, true, false, 257, 50, def) == PEND_FWD_TL);
    assert(m(P, RK, false, false, 256, 50, def) == PEND_FWD_SH && m(P, TS, false, true, 700, 50, def) == PEND_FWD_SH);
    // beyond: a lane per trajectory, through the row ring from 2^17 on (save grids that fit beside it)
    assert(m(P, TS, true, false, 16385, 50, def) == PEND_FWD_LANE && m(P, TS, true, true, (1 << 17) - 1, 50, def) == PEND_FWD_LANE);
    assert(m(P, TS, true, false, 1 << 17, 50, def) == PEND_FWD_RING && m(P, TS, true, true, 1 << 20, 50, def) == PEND_FWD_RING && m(P, TS, true, false, 1 << 20, 2049, def) == PEND_FWD_LANE);
    // degenerate grids: T = 1 solves nothing (a lane per trajectory writes ẑ₀); T = 2 has no interior for k_pend_forward_ws
    assert(m(P, TS, true, false, 256, 1, def) == PEND_FWD_LANE && m(P, TS, true, true, 5000, 2, def) == PEND_FWD_LANE && m(P, TS, true, true, 5000, 6001, def) == PEND_FWD_LANE);
    // the options the tests force a mapping with: "pend_sh_max_b" ≥ 0 is ONE threshold for lp and sh; "pend_lp" = 0 sends the metric's shape to sh
    lde::PendTune t = def;
    t.sh_max_b = 0;
    assert(m(P, TS, true, false, 1, 50, t) == PEND_FWD_TL && m(P, TS, true, true, 1, 50, t) == PEND_FWD_WS);
    t.sh_max_b = 1 << 20;
    assert(m(P, TS, true, true, 5000, 50, t) == PEND_FWD_LP3 && m(F, TS, true, false, 5000, 50, t) == PEND_FWD_SH);
    t.lp = 0;
    assert(m(P, TS, true, true, 100, 50, t) == PEND_FWD_SH);
    t = def; t.ws = 0; t.tl_max_b = 0; t.sh_max_b = 0; t.lb_min_b = 0;
    assert(m(P, TS, true, false, 1000, 50, t) == PEND_FWD_RING);
    t.lb_ring = 0;
    assert(m(P, TS, true, false, 1000, 50, t) == PEND_FWD_LANE);
    // … and for ANY arguments one of the seven
    for (int it = 0; it < 20000; it++) {
      lde::PendTune h;
      h.ws = (int)(rnd() % 3); h.tl_max_b = (int)(rnd() % 5000); h.sh_max_b = (int)(rnd() % 3000) - 1; h.lp = (int)(rnd() & 1);
      h.lb_ring = (int)(rnd() % 40); h.lb_min_b = (int)(rnd() % 300000);
      const PendFwdMap r = m((int)(rnd() % 2), (int)(rnd() % 2), rnd() & 1, rnd() & 1, 1 + (int)(rnd() % (1 << 21)), 1 + (int)(rnd() % 7000), h);
      assert(r >= PEND_FWD_LP4 && r <= PEND_FWD_LANE);
    }
  }
  std::printf("host logic under ASan + UBSan: %d accepted, %d refused hostile descriptions; forward mappings as measured\n", n_ok, n_bad);
  return 0;
}

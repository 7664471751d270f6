"""Where the discrete pullback of the metric configuration (k_pend_adjoint_disc_tp, B = 256: a wave per trajectory) spends its time —
cycle stamps of workgroup 0 from a -DLDE_PEND_PROF=1 build.

    python abl/disc_tp_prof.py            (on the GPU box; uses / builds abl/liblde_pprof.so)

Stamps: 0 entry, 1 loads landed (save grid and Δẑ in LDS, the record's first round in registers), 2 stage points and the tangent through
them done, 3 save times contracted (the three lanes' shares added up), 4 c_τ formed, 6 the step maps gathered (lane p = step p), 7 the four
scan levels done, 5 sweep done. The compiler moves arithmetic across the stamps (most of c_τ lands behind stamp 4), so the split between
neighbouring phases is approximate; the sum is not. With several rounds (records longer than 21 steps) 2–7 are the last round's.

Round 5, B = 256 (µs, shader clock 2.3 GHz): loads 0.62, tangent 0.49, save times 0.85, c_τ + gather 0.6, scan 0.23, rest 0.17: 2.96 from entry
to the sweep's end, 4.5 per launch back to back. What was tried on the way (each measured with this script):
  the sweep as a chain of 21 map applications — value passed by v_readlane → scalar → VALU: 92 cycles per step; by DPP wave_shl:1: 100; by DPP
  row_shl:1 within rows: 110 (one wave per SIMD: every dependent instruction's latency is exposed) — as a suffix scan of the maps: 530 cycles;
  the save times of a step walked by every one of its three lanes: 2700 cycles (the longest step of the trajectory decides: 3 round trips
  of 4) — shared between the three lanes and added up in LDS: 2000."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PLIB = os.path.join(ROOT, "abl", "liblde_pprof.so")
if "LDE_LIB_PATH" not in os.environ:
    if not os.path.exists(PLIB):
        import latentdiffeq_amd as l
        l.build_lib(extra_flags=["-DLDE_PEND_PROF=1"], out=PLIB)
    os.environ["LDE_LIB_PATH"] = PLIB
    sys.exit(subprocess.run([sys.executable] + sys.argv).returncode)     # (a child: _lib reads LDE_LIB_PATH at import)

import torch                                           # noqa: E402
from latentdiffeq_amd import _lib as L                 # noqa: E402
from latentdiffeq_amd import synthetic as S            # noqa: E402

lib = L.load()
B, T = 256, 50
d = L.ProblemDesc()
lib.lde_problem_desc_default(C.byref(d))
d.sensealg = L.SENSE_DISCRETE
h = C.c_void_p()
L.check(lib.lde_create(C.byref(d), C.byref(h)), None, "create")
z0, th = S.pendulum_inputs(B)
ts = S.time_grid(T)
z0d, thd = torch.from_numpy(z0).cuda(), torch.from_numpy(th).cuda()
zo = torch.empty(T, B, 2, device="cuda")
dz = torch.randn(T, B, 2, device="cuda")
g0, gth = torch.empty(B, 2, device="cuda"), torch.empty(B, 1, device="cuda")
ret = torch.empty(B, dtype=torch.int32, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
L.check(lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zo), p(ret), C.c_void_p()), h, "fwd")
rows = []
for it in range(40):
    L.check(lib.lde_adjoint(h, p(zo), p(thd), tsp, T, B, p(dz), p(g0), p(gth), None, C.c_void_p()), h, "adj")
    torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    assert lib.lde_debug_pend_prof(out) == 0
    rows.append(np.array(out[:], dtype=np.int64))
v = np.array(rows[10:]).astype(np.float64)
wall = lambda i: v[:, 2 * i]
cyc = lambda i: v[:, 2 * i + 1]
names = ["loads (grid, cotangents, record) + LDS fill", "stage points + tangent through the step", "save times of the step contracted",
         "c_tau from the slopes' tangents", "sweep over the steps (gather, suffix scan, dθ)"]
print(f"k_pend_adjoint_disc_tp, B = {B}, workgroup 0")
for i, nm in enumerate(names):
    print(f"  {nm:48s} {(cyc(i + 1) - cyc(i)).mean():8.0f} cycles {(wall(i + 1) - wall(i)).mean() * 10 / 1e3:6.2f} us")
print(f"    of the sweep: gather {(cyc(6) - cyc(4)).mean():.0f}, four scan levels {(cyc(7) - cyc(6)).mean():.0f}, rest {(cyc(5) - cyc(7)).mean():.0f} cycles")
print(f"  entry → sweep done                               {(cyc(5) - cyc(0)).mean():8.0f} cycles {(wall(5) - wall(0)).mean() * 10 / 1e3:6.2f} us")
for name, fn in (("lde_adjoint", lambda: lib.lde_adjoint(h, p(zo), p(thd), tsp, T, B, p(dz), p(g0), p(gth), None, C.c_void_p())),
                 ("lde_forward (recording)", lambda: lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zo), p(ret), C.c_void_p()))):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 500
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"  {name} back to back (prof build): {ev[0].elapsed_time(ev[1]) / n * 1e3:.2f} us per launch")

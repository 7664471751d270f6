"""Where the metric's forward kernel (k_pend_forward_sh, B = 256: one trajectory per workgroup) spends its time — cycle stamps of the
stepping wave and of the three dense-output waves from a -DLDE_PEND_PROF=1 build, beside the launch floor of a kernel that does nothing.

    python abl/metric_floor.py            (on the GPU box; builds abl/liblde_pprof.so on first use: ≈ 3 min)

Stamps (workgroup 0; shader clock = __builtin_readcyclecounter, wall = 100 MHz): 8 stepper entry, 9 stepping loop starts (inputs loaded,
k₁ = f(y₀) and the Hairer initial step — two evaluations, two pow — done), 10 loop ended, 11 stepper done; 12–14 dense-output wave i has
stored its last save. DESIGN.md §9 prices the metric's floor with these numbers."""
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
h = C.c_void_p()
L.check(lib.lde_create(C.byref(d), C.byref(h)), None, "create")
z0, th = S.pendulum_inputs(B)
ts = S.time_grid(T)
z0d, thd = torch.from_numpy(z0).cuda(), torch.from_numpy(th).cuda()
zo = torch.empty(T, B, 2, device="cuda")
ret = torch.empty(B, dtype=torch.int32, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
rows = []
for it in range(40):
    L.check(lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zo), p(ret), C.c_void_p()), h, "fwd")
    torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    assert lib.lde_debug_pend_prof(out) == 0
    rows.append(np.array(out[:], dtype=np.int64))
v = np.array(rows[10:]).astype(np.float64)
wall = lambda i: v[:, 2 * i]
cyc = lambda i: v[:, 2 * i + 1]
steps = v[:, 30].mean()
seg = [("stepper: loads, k1 = f(y0), Hairer initial step", 8, 9), ("stepper: the stepping loop", 9, 10), ("stepper: publish, statistics, exit", 10, 11)]
lib.lde_last_kernel.restype = C.c_char_p
print(f"{lib.lde_last_kernel(h, 0).decode()}, B = {B}, workgroup 0: {steps:.1f} step attempts")
for name, a, b_ in seg:
    dc, dw = (cyc(b_) - cyc(a)).mean(), (wall(b_) - wall(a)).mean() * 10.0
    extra = f"   = {dc / steps:.0f} cycles = {dw / steps:.0f} ns per step" if a == 9 else ""
    print(f"  {name:52s} {dc:8.0f} cycles {dw / 1e3:6.2f} us{extra}")
for i in range(3):
    print(f"  dense-output wave {i} stores its last save {((wall(12 + i) - wall(8)).mean()) / 100:6.2f} us after the stepper's entry "
          f"({((wall(12 + i) - wall(10)).mean()) / 100:5.2f} us after the loop's end)")
if v[:, 1].mean() > 0:   # k_pend_forward_lp's helper statistics (slots 0..6)
    print(f"  helper 0: own records {v[:, 0].mean() / v[:, 1].mean():6.0f} cycles each ({v[:, 1].mean():.1f}), idle polls {v[:, 4].mean():.0f}, entry to done {v[:, 6].mean():.0f} cycles")
    n_ = v[:, 1].mean()
    print(f"            per own record: walk to it {v[:, 7].mean() / n_:.0f}, read + times + skip {v[:, 8].mean() / n_:.0f}, polynomials {v[:, 9].mean() / n_:.0f}, evaluate + store {v[:, 10].mean() / n_:.0f} cycles")
print(f"  shader clock during the loop: {((cyc(10) - cyc(9)) / ((wall(10) - wall(9)) * 10e-9)).mean() / 1e9:.2f} GHz")
# the kernel as the host sees it (HIP events, back to back) and an empty kernel's launch floor
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 500
ev[0].record()
for _ in range(n):
    lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zo), p(ret), C.c_void_p())
ev[1].record()
torch.cuda.synchronize()
print(f"  lde_forward back to back (prof build): {ev[0].elapsed_time(ev[1]) / n * 1e3:.2f} us per launch")

"""Phase profile of k_pend_forward_ws (metric config) from a -DLDE_PEND_PROF=1 build:
    python -c "import latentdiffeq_amd as l; l.build_lib(extra_flags=['-DLDE_PEND_PROF=1'], out='abl/liblde_pprof.so')"
    LDE_LIB_PATH=$PWD/abl/liblde_pprof.so python abl/pend_prof.py
Stamps (stepper wave): 0 entry, 1 after the save-grid fill + barrier, 2 stepping loop start, 3 loop end, 4 end of round
published, 7 kernel end. Wall clock is 100 MHz; cycles = s_memtime."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentdiffeq_amd import _lib as L          # noqa: E402
from latentdiffeq_amd import synthetic as S     # noqa: E402

lib = L.load()
B, T = int(os.environ.get("B", 256)), 50
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
for it in range(30):
    L.check(lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zo), p(ret), C.c_void_p()), h, "fwd")
    torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    assert lib.lde_debug_pend_prof(out) == 0
    v = np.array(out[:], dtype=np.int64)
    rows.append(v)
v = np.array(rows[10:])
idx = [0, 1, 2, 3, 4, 7]
wall = np.stack([v[:, 2 * i] for i in idx], axis=1).astype(np.float64)
cyc = np.stack([v[:, 2 * i + 1] for i in idx], axis=1).astype(np.float64)
names = ["input loads + grid fill + barrier", "init_dt", "stepping loop", "publish end of round", "wait for helpers + epilogue"]
iters = v[:, 30].mean()
print(f"helpers nh={v[:,28].mean():.1f}"); print(f"B={B} wave-iterations of the stepping loop (workgroup 0, last round): {iters:.1f}")
for i, n in enumerate(names):
    dw = (wall[:, i + 1] - wall[:, i]).mean() * 10.0          # ns
    dc = (cyc[:, i + 1] - cyc[:, i]).mean()
    extra = f"  = {dw / iters:.0f} ns, {dc / iters:.0f} cycles per step" if n == "stepping loop" else ""
    print(f"{n:34s} {dw / 1000:7.2f} us {dc:9.0f} cycles{extra}")
print(f"{'total (stepper wave)':34s} {(wall[:, -1] - wall[:, 0]).mean() / 100:7.2f} us")
t0 = v[:, 0].astype(np.float64)
print(f"helper rank 0 exits at {((v[:, 26] - t0).mean()) / 100:7.2f} us, last working helper at {((v[:, 29] - t0).mean()) / 100:7.2f} us, last idle wave at {((v[:, 27] - t0).mean()) / 100:7.2f} us after kernel entry")
print("last helper (lane 0): polls", v[:, 24].mean(), " its saves stored at (us after entry, last save first):", [round(float(((v[:, 16 + i] - t0).mean()) / 100), 2) for i in range(5)])

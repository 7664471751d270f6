"""k_pend_forward_lp against k_pend_forward_sh (option "pend_lp" = 0) and the f64 oracle over random problems: batches 1 … 1 024 (both helper
counts), save grids uniform / random / dense / sparse, tolerances 1e-3 … 1e-8, amplitudes up to near the separatrix, time spans, both sensealgs
(recording or not). Two correct f32 solves of one method: the gates are the ones tests/test_gpu_pendulum.py holds every mapping to —
99 % of trajectories within 10·reltol-scaled bounds of the f64 solve, failures / NaN blocks identical, records consistent (t[k+1] = t[k] + dt[k])."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
o64 = O.Oracle("f64")
o32 = O.Oracle("f32")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = dict(z=0.0, zs=0.0, dsteps=0, g=0.0)
bad = 0
for it in range(N):
    B = int(rng.choice([1, 3, 8, 63, 64, 200, 256, 257, 400, 512, 513, 700, 1024]))
    T = int(rng.choice([2, 3, 17, 50, 65, 66, 130, 400]))
    tol = 10.0 ** rng.uniform(-8, -3)
    rt = tol if rng.random() < 0.5 else 10.0 ** rng.uniform(-8, -3)
    span = float(rng.choice([0.3, 2.45, 6.0, 15.0]))
    gk = rng.integers(0, 3)
    ts = np.linspace(0.0, span, T) if gk == 0 else np.concatenate([[0.0], np.sort(rng.uniform(0, span, T - 1))]) if gk == 1 else \
        np.concatenate([[0.0], np.sort(np.concatenate([rng.uniform(0.4 * span, 0.41 * span, (T - 1) // 2), rng.uniform(0, span, T - 1 - (T - 1) // 2)]))])
    ts = np.maximum.accumulate(ts + np.arange(T) * 1e-9)
    amp = float(rng.choice([0.5, 1.5, 2.9]))
    z0 = np.stack([rng.uniform(-amp, amp, B), rng.uniform(-1.5, 1.5, B)], 1).astype(np.float32)
    Lp = rng.uniform(0.5, 2.5, (B, 1)).astype(np.float32)
    sense = LL.SENSE_DISCRETE if rng.random() < 0.6 else LL.SENSE_PARALLEL_CHECKPOINTED
    kw = dict(abstol=tol, reltol=rt, sensealg=sense)
    d = make_desc(**kw)
    od = copy_desc_to_oracle(d)
    res = {}
    for name, opts in (("lp", {"pend_sh_max_b": 1 << 20}), ("sh", {"pend_sh_max_b": 1 << 20, "pend_lp": 0})):
        nat = Native(make_desc(**kw))
        for k, v in opts.items(): nat.set_option(k, v)
        nat.set_option("record_capacity", 4096)
        z, ret, st = nat.forward(z0, Lp, ts)
        assert nat.lib.lde_last_kernel(nat.h, 0).decode() == ("k_pend_forward_lp" if name == "lp" else "k_pend_forward_sh")
        rec = nat.step_record(0, B) if sense == LL.SENSE_DISCRETE else None
        g = None
        if sense == LL.SENSE_DISCRETE and (ret == 0).all():
            dz = (rng.standard_normal((T, B, 2)) / (B * T)).astype(np.float32)
            rng_state = None
            g = nat.adjoint(z, Lp, ts, dz)[:2]
            r = o64.adjoint_discrete(od, o64.forward_steps(od, z0, Lp, ts, rec=rec)[0], Lp, ts, dz, rec)[:2]
            ge = max(np.abs(g[0] - r[0]).max() / max(np.abs(r[0]).max(), 1e-30), np.abs(g[1] - r[1]).max() / max(np.abs(r[1]).max(), 1e-30))
            worst["g"] = max(worst["g"], ge)
            if ge > 2e-4:
                r32 = o32.adjoint_discrete(od, o32.forward_steps(od, z0, Lp, ts, rec=rec)[0], Lp, ts, dz, rec)[:2]
                rel = lambda a, b: max(np.abs(a[0] - b[0]).max() / max(np.abs(b[0]).max(), 1e-30), np.abs(a[1] - b[1]).max() / max(np.abs(b[1]).max(), 1e-30))
                g32, o3264 = rel(g, r32), rel(r32, r)
                ib = int(np.argmax(np.abs(g[0] - r[0]).max(axis=1)))
                print("GRAD", it, name, B, T, f"{tol:.1e} {rt:.1e}", span, f"gpu-o64 {ge:.2e} gpu-o32 {g32:.2e} o32-o64 {o3264:.2e}; worst b {ib} z0 {z0[ib]} L {Lp[ib]} n {int(rec['n'][ib])} |g| {np.abs(r[0][ib]).max():.2e} max|g| {np.abs(r[0]).max():.2e}")
                if g32 > 10 * o3264 + 1e-4: bad += 1
        res[name] = (z, ret, st, rec)
    zt, rt64, _ = o64.forward(od, z0, Lp, ts)
    zl, rl, sl, recl = res["lp"]; zs, rs, ss, _ = res["sh"]
    ok = (rl == 0) & (rs == 0) & (rt64 == 0)
    if not np.array_equal(rl != 0, rs != 0): print("RET differs", it, B, T, (rl != 0).sum(), (rs != 0).sum())
    if ok.any():
        el = np.abs(zl[:, ok] - zt[:, ok]).max(axis=(0, 2)); es = np.abs(zs[:, ok] - zt[:, ok]).max(axis=(0, 2))
        # lp no farther from f64 truth than 2× sh's distance + a floor (two f32 solves with free-running controllers)
        lim = 2.0 * np.quantile(es, 0.99) + 2e-5 * max(1.0, span)
        q = np.quantile(el, 0.99)
        worst["z"] = max(worst["z"], q / lim)
        if q > lim: bad += 1; print("Z", it, B, T, tol, rt, span, q, lim)
        worst["dsteps"] = max(worst["dsteps"], abs(sl["naccept"] - ss["naccept"]) / max(1, ss["naccept"]))
    if recl is not None:
        n = recl["n"]
        for b in range(min(B, 64)):
            k = int(n[b])
            if k > 1 and not np.array_equal(recl["t"][b, 1:k], recl["t"][b, :k - 1] + recl["dt"][b, :k - 1]):
                bad += 1; print("REC inconsistent", it, b); break
    print(f"{it:3d} B={B:5d} T={T:4d} tol={tol:.1e}/{rt:.1e} span={span:5.2f} grid={gk} {'disc' if sense == LL.SENSE_DISCRETE else 'cont'} fails {int((rl != 0).sum())} steps lp/sh {sl['naccept']}/{ss['naccept']}", flush=True)
print("worst", worst, "bad", bad)

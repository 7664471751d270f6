"""float32 emulation (numpy; fma = one rounding) of k_pend_forward_lp's step against the register-pair form and float64: where does
round-off per step come from?"""
import numpy as np
import sys
sys.path.insert(0, "/root/repo/abl")
from rkn_coeffs import A, BT, C, Abar, Et
f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
INV = f32(0.15915494309189535); TWO_PI = f32(6.283185307179586)
def vsin_turns(t): return f32(np.sin(2 * np.pi * np.float64(t)))       # exact sine of the f32 argument, rounded

def step_lp(xi, om, s1, h, glt, turns=True):
    """xi, om: state (turns); returns new state, sines"""
    xi0 = f32(xi - np.rint(xi)); hw = f32(h * om); hg = f32(h * glt); hhg = f32(hg * h)
    sg = [None, s1]
    def inner(i, upto):
        acc = f32(f32(Abar[i, 1]) * sg[1])
        for l in range(2, upto + 1): acc = fma(f32(Abar[i, l]), sg[l], acc)
        return acc
    for i in range(2, 8):
        base = fma(f32(C[i]), hw, xi0) if i < 6 else f32(xi0 + hw)
        ang = fma(hhg, inner(i, i - 2), base) if i > 2 else base
        sg.append(vsin_turns(ang))
    nx = inner(7, 5)
    xn = fma(hhg, nx, fma(f32(1), hw, xi))
    nv = f32(f32(A[7, 1]) * sg[1])
    for l in range(2, 7): nv = fma(f32(A[7, l]), sg[l], nv)
    on = fma(hg, nv, om)
    return xn, on, sg[7]

def step_pair(x, v, k1, h, ngl):
    """the register-pair form in f32 (x, v radians); k = (v_j, s_j)"""
    k = [None, (v, k1)]
    for i in range(2, 8):
        ax = f32(f32(A[i, 1]) * k[1][0]); av = f32(f32(A[i, 1]) * k[1][1])
        for j in range(2, i): ax = fma(f32(A[i, j]), k[j][0], ax); av = fma(f32(A[i, j]), k[j][1], av)
        xi = fma(ax, h, x); vi = fma(av, h, v)
        if i == 7: xn, vn = xi, vi
        t = fma(xi, INV, f32(-np.rint(f32(xi * INV))))
        k.append((vi, f32(ngl * vsin_turns(t))))
    return xn, vn, k[7][1]

def step64(x, v, h, gl):
    f = lambda y: np.array([y[1], -gl * np.sin(y[0])])
    y = np.array([x, v]); k = [None, f(y)]
    for i in range(2, 7): k.append(f(y + h * sum(A[i, j] * k[j] for j in range(1, i))))
    return y + h * sum(A[7, j] * k[j] for j in range(1, 7))

rng = np.random.default_rng(0)
for h in (0.012, 0.05, 0.19):
    worst = [0, 0]
    for tr in range(20):
        x0, v0, L = rng.uniform(-0.5, 0.5), rng.uniform(-1, 1), rng.uniform(1, 2)
        gl = 10 / L; ngl = f32(-10.0) / f32(L); glt = f32(ngl * INV)
        n = int(2.45 / h)
        y = np.array([x0, v0])
        x, v = f32(x0), f32(v0); k1 = f32(ngl * vsin_turns(f32(x * INV)))
        xi, om = f32(x * INV), f32(v * INV); s1 = vsin_turns(f32(x * INV))
        hh = f32(h)
        for _ in range(n):
            y = step64(y[0], y[1], np.float64(hh), -np.float64(ngl))
            x, v, k1 = step_pair(x, v, k1, hh, ngl)
            xi, om, s1 = step_lp(xi, om, s1, hh, glt)
        worst[0] = max(worst[0], abs(x - y[0]), abs(v - y[1]))
        worst[1] = max(worst[1], abs(f32(xi * TWO_PI) - y[0]), abs(f32(om * TWO_PI) - y[1]))
    print(f"h = {h}: {n} steps: pair form |err| {worst[0]:.2e}   lp form |err| {worst[1]:.2e}")

# ---- the dense output from the sines (helpers of k_pend_forward_lp) against the pair form, in f32, one step
from rkn_coeffs import RR, RA
def dense_lp(x, v, sg, h, ngl, th):
    hg = f32(h * ngl)
    px, pv = [], []
    for m in range(3):
        ax = f32(f32(RA[m, 1]) * sg[1]); av = f32(f32(RR[1, m]) * sg[1])
        for l in range(2, 7): ax = fma(f32(RA[m, l]), sg[l], ax)
        for j in range(2, 8): av = fma(f32(RR[j, m]), sg[j], av)
        px.append(f32(hg * ax)); pv.append(f32(ngl * av))
    def ev(y, k1, P2, P3, P4):
        return f32(y + f32(h * th) * f32(k1 + th * f32(P2 + th * f32(P3 + th * P4))))
    return ev(x, v, px[0], px[1], px[2]), ev(v, f32(ngl * sg[1]), pv[0], pv[1], pv[2])
def dense64(x, v, h, gl, th):
    f = lambda y: np.array([y[1], -gl * np.sin(y[0])])
    y = np.array([x, v]); k = [None, f(y)]
    for i in range(2, 7): k.append(f(y + h * sum(A[i, j] * k[j] for j in range(1, i))))
    yn = y + h * sum(A[7, j] * k[j] for j in range(1, 7)); k.append(f(yn))
    b = [0, th * (1 + th * (RR[1, 0] + th * (RR[1, 1] + th * RR[1, 2])))] + [th * th * (RR[j, 0] + th * (RR[j, 1] + th * RR[j, 2])) for j in range(2, 8)]
    return y + h * sum(b[j] * k[j] for j in range(1, 8))
for h in (0.012, 0.05, 0.19):
    w = 0
    for tr in range(200):
        x0, v0, L, th = rng.uniform(-0.5, 0.5), rng.uniform(-1, 1), rng.uniform(1, 2), rng.uniform(0.05, 0.95)
        ngl = f32(-10.0) / f32(L); glt = f32(ngl * INV)
        x, v = f32(x0), f32(v0)
        xi, om = f32(x * INV), f32(v * INV); s1 = vsin_turns(f32(x * INV))
        # sines of the step (lp form)
        xi0 = f32(xi - np.rint(xi)); hw = f32(f32(h) * om); hg_t = f32(f32(h) * glt); hhg = f32(hg_t * f32(h))
        sg = [None, s1]
        for i in range(2, 8):
            acc = f32(0)
            if i > 2:
                acc = f32(f32(Abar[i, 1]) * sg[1])
                for l in range(2, i - 1): acc = fma(f32(Abar[i, l]), sg[l], acc)
            base = fma(f32(C[i]), hw, xi0) if i < 6 else f32(xi0 + hw)
            sg.append(vsin_turns(fma(hhg, acc, base)))
        dx, dv = dense_lp(x, v, sg, f32(h), ngl, f32(th))
        t = dense64(np.float64(x), np.float64(v), np.float64(f32(h)), -np.float64(ngl), np.float64(f32(th)))
        w = max(w, abs(dx - t[0]), abs(dv - t[1]))
    print(f"dense output from the sines, h = {h}: max |err| vs f64 {w:.2e}")

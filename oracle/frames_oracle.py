"""CPU oracle for the frame rasteriser of scope row f-4 (test infrastructure: imported by tests only).

The reference draws a frame with Luxor [REF examples/pendulum_friction-less/create_data.jl:90-104]: on a black 28×28 canvas (origin
at the centre, y down) a white disc of radius r at the bob, one at the pivot (0, −8.5), a white rod of thickness w between them, and a
black disc of radius r/2 on the pivot. With the example's numbers (r = 1.75, w = 3.75 [REF model_train.jl / create_data.jl:17-18:
high_dim_args = (19, 1.75, 3.75)]) the rod's half thickness 1.875 exceeds r, so — taking the rod with round caps, as the product's
rasteriser does — both discs lie inside it and the white set is exactly the CAPSULE {p : dist(p, segment pivot–bob) ≤ w/2} minus the
inner disc. (Luxor's rod is the glyph "|" drawn at font size 8 [REF create_data.jl:84-87]; its exact outline needs Cairo, which is
why parity with the reference's pixels is out of reach; the geometry above is what is pinned.)

Exact coverage of pixel [i, i+1) × [j, j+1) = ∫ over x of the length of the shape's vertical section clipped to the pixel. The section
of a capsule or a disc at abscissa x is ONE interval whose ends are closed-form (a capsule is convex): the lower envelope of the two cap
circles and the two side lines. The x-integral is a composite Simpson rule on 2 049 points per pixel column (the integrand is
continuous, piecewise smooth; error ≲ 1e-6 of a pixel), in float64 — three orders of magnitude finer than the product's 4×4 supersampling.
"""
import numpy as np


def _capsule_section(x, ax, ay, bx, by, r):
    """[lo, hi] of {y : dist((x, y), segment a–b) ≤ r} for an array of abscissae (NaN where empty)."""
    lo = np.full_like(x, np.inf)
    hi = np.full_like(x, -np.inf)
    for cx, cy in ((ax, ay), (bx, by)):                      # the two cap discs
        d2 = r * r - (x - cx) ** 2
        ok = d2 >= 0
        s = np.sqrt(np.where(ok, d2, 0.0))
        lo = np.where(ok, np.minimum(lo, cy - s), lo)
        hi = np.where(ok, np.maximum(hi, cy + s), hi)
    dx, dy = bx - ax, by - ay
    L = np.hypot(dx, dy)
    if L > 0:                                               # the rectangle between the caps: |n·(p − a)| ≤ r, 0 ≤ t·(p − a) ≤ L
        tx, ty, nx, ny = dx / L, dy / L, -dy / L, dx / L
        # along the vertical line p = (x, y): n·(p−a) = nx(x−ax) + ny(y−ay), t·(p−a) = tx(x−ax) + ty(y−ay): linear in y → an interval
        def lin_interval(c0, c1, lo_v, hi_v):               # {y : lo_v ≤ c0 + c1·y ≤ hi_v}
            with np.errstate(divide="ignore", invalid="ignore"):
                y1, y2 = (lo_v - c0) / c1, (hi_v - c0) / c1
            a_, b_ = np.minimum(y1, y2), np.maximum(y1, y2)
            if c1 == 0:
                inside = (c0 >= lo_v) & (c0 <= hi_v)
                return np.where(inside, -np.inf, np.inf), np.where(inside, np.inf, -np.inf)
            return a_, b_
        n0, n1 = nx * (x - ax) - ny * ay, ny
        t0, t1 = tx * (x - ax) - ty * ay, ty
        la, ha = lin_interval(n0, n1, -r, r)
        lb, hb = lin_interval(t0, t1, 0.0, L)
        rl, rh = np.maximum(la, lb), np.minimum(ha, hb)
        ok = rl <= rh
        lo = np.where(ok, np.minimum(lo, rl), lo)
        hi = np.where(ok, np.maximum(hi, rh), hi)
    return lo, hi


def exact_frame(theta, pendulumlength=19.0, radius=1.75, rodthickness=3.75, w=28, h=28, n=2049):
    """Exact coverage image [h, w] (float64) for one pendulum angle; valid when rodthickness/2 ≥ radius (the example's numbers)."""
    assert rodthickness / 2 >= radius, "the closed form assumes the discs lie inside the rod's caps"
    ox, oy = 0.0, -8.5
    px, py = ox + pendulumlength * np.cos(np.pi / 2 + theta), oy + pendulumlength * np.sin(np.pi / 2 + theta)
    img = np.zeros((h, w))
    u = np.linspace(0.0, 1.0, n)
    wts = np.ones(n)
    wts[1:-1:2], wts[2:-1:2] = 4.0, 2.0
    wts *= 1.0 / (3.0 * (n - 1))                               # Simpson on [0, 1]
    for i in range(w):
        x = i - w / 2 + u
        lo, hi = _capsule_section(x, ox, oy, px, py, rodthickness / 2)
        d2 = (radius / 2) ** 2 - (x - ox) ** 2
        s = np.sqrt(np.maximum(d2, 0.0))
        ilo, ihi = np.where(d2 >= 0, oy - s, np.inf), np.where(d2 >= 0, oy + s, -np.inf)
        for j in range(h):
            y0, y1 = j - h / 2, j - h / 2 + 1
            white = np.maximum(np.minimum(hi, y1) - np.maximum(lo, y0), 0.0)
            black = np.maximum(np.minimum(ihi, y1) - np.maximum(ilo, y0), 0.0)
            img[j, i] = float(np.dot(wts, white - black))
    return img


def exact_area(pendulumlength=19.0, radius=1.75, rodthickness=3.75):
    """Area of the white set in closed form: capsule minus the inner disc (all of it inside the canvas for |θ| ≤ π/2)."""
    r = rodthickness / 2
    return np.pi * r * r + 2 * r * pendulumlength - np.pi * (radius / 2) ** 2

"""The frame rasteriser of scope row f-4 (latentdiffeq_amd.data.create_frames) against an exact-coverage oracle
(oracle/frames_oracle.py: closed-form vertical sections of the drawing's geometry [REF examples/pendulum_friction-less/
create_data.jl:90-104], integrated in float64). The rasteriser is a torch expression — no HIP kernel — so this runs on the CPU."""
import numpy as np
import pytest
import torch

from latentdiffeq_amd.data import create_frames
from oracle.frames_oracle import exact_area, exact_frame


def test_oracle_total_area_is_the_closed_form():
    for th in (0.0, 0.4, -0.65):                       # |θ| ≤ 0.69: the whole drawing lies on the canvas (the data's swings stay below)
        img = exact_frame(th)
        assert abs(img.sum() - exact_area()) <= 2e-5 * exact_area(), (th, img.sum(), exact_area())
        assert img.min() >= -1e-9 and img.max() <= 1 + 1e-9


@pytest.mark.parametrize("theta", [0.0, 0.3, -0.52, 1.0, -1.2])
def test_create_frames_against_exact_coverage(theta):
    ex = exact_frame(theta)
    for ss, tol_max, tol_mean in ((4, 0.15, 8e-3), (16, 0.035, 1e-3)):   # measured: 0.125 / 6.6e-3 at θ = 0 (axis-aligned edges, the worst case), 0.008 / 1.5e-4
        f = create_frames(torch.tensor([theta]), ss=ss)[0].double().numpy()          # [h, w]
        err = np.abs(f - ex)
        # ss×ss point samples of a shape with a smooth boundary: a boundary pixel's coverage is off by O(1/ss) at worst, the
        # image mean by O(1/ss²); interior and exterior pixels are exact
        assert err.max() <= tol_max and err.mean() <= tol_mean, (ss, err.max(), err.mean())
        assert abs(f.sum() - ex.sum()) <= (0.02 if ss == 4 else 0.004) * exact_area()     # (beyond |θ| = 0.69 part of the bob leaves the canvas)
        inside, outside = ex > 1 - 1e-9, ex < 1e-9
        assert np.all(f[inside] == 1.0) and np.all(f[outside] == 0.0)

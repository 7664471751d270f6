"""Tsit5 on the frictionless pendulum as a Nyström scheme (round 6, k_pend_forward_lp): with f = (v, s(x)), s = −(g/L) sin x, the stage
angles need no stage velocities — x_i = x + c_i h v + h² Σ_l Ā_il s_l, Ā = A·A — and s_i depends only on s_1 … s_{i−2}: two interleaved
chains (2,4,6) and (3,5,7) of depth three instead of one chain of six. Derives the coefficient tables the kernel carries (lde_pend_lp.h
recomputes them as constexpr doubles; this script prints them and checks one step against the plain Tsit5 step in float64)."""
import numpy as np

A = np.zeros((8, 8))
A[2, 1] = 0.161
A[3, 1:3] = [-0.008480655492356989, 0.335480655492357]
A[4, 1:4] = [2.8971530571054935, -6.359448489975075, 4.3622954328695815]
A[5, 1:5] = [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525]
A[6, 1:6] = [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383]
A[7, 1:7] = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]
BT = np.array([0, -0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
               -0.45808210592918697, 0.015151515151515152])
C = A.sum(axis=1)
Abar = A @ A            # Abar[i, l] = Σ_j a_ij a_jl  (row 7 = B̄)
Et = BT @ A             # Ẽ_l = Σ_j b̃_j a_jl  (j = 1..7, a_7l = b_l)
R1 = np.array([-2.763706197274826, 2.9132554618219126, -1.0530884977290216])
R = np.array([[0.13169999999999998, -0.2234, 0.1017], [3.9302962368947516, -5.941033872131505, 2.490627285651253],
              [-12.411077166933676, 30.33818863028232, -16.548102889244902], [37.50931341651104, -88.1789048947664, 47.37952196281928],
              [-27.896526289197286, 65.09189467479366, -34.87065786149661], [1.5, -4.0, 2.5]])
RR = np.zeros((8, 3)); RR[1] = R1; RR[2:] = R      # r_{j,m}: dense-output polynomial coefficients of k_j (Θ², Θ³, Θ⁴)
RS = RR.sum(axis=0)                                 # Σ_j r_jm
RA = RR.T @ A                                       # RA[m, l] = Σ_j r_jm a_jl


def step_plain(x, v, h, gl):
    f = lambda y: np.array([y[1], -gl * np.sin(y[0])])
    y = np.array([x, v]); k = [None] * 8
    k[1] = f(y)
    for i in range(2, 8):
        k[i] = f(y + h * sum(A[i, j] * k[j] for j in range(1, i)))
    yn = y + h * sum(A[7, j] * k[j] for j in range(1, 7))
    err = h * sum(BT[j] * k[j] for j in range(1, 8))
    return yn, err, k


def step_rkn(x, v, h, gl):
    s = np.zeros(8); hv, hh = h * v, h * h
    s[1] = -gl * np.sin(x)
    for i in range(2, 8):
        xi = x + C[i] * hv + hh * sum(Abar[i, l] * s[l] for l in range(1, i - 1))
        s[i] = -gl * np.sin(xi)
    xn = x + hv + hh * sum(Abar[7, l] * s[l] for l in range(1, 6))
    vn = v + h * sum(A[7, l] * s[l] for l in range(1, 7))
    ex = hh * sum(Et[l] * s[l] for l in range(1, 7))
    ev = h * sum(BT[l] * s[l] for l in range(1, 8))
    return np.array([xn, vn]), np.array([ex, ev]), s


if __name__ == "__main__":
    assert abs(C[7] - 1) < 1e-15 and abs(C[6] - 1) < 1e-15 and abs(BT.sum()) < 1e-16
    for i in range(2, 8):
        assert np.all(Abar[i, max(i - 1, 1):] == 0), i          # x_i needs s_1 … s_{i−2} only
    rng = np.random.default_rng(0)
    worst = 0
    for _ in range(200):
        x, v, h, gl = rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.01, 0.4), 10 / rng.uniform(1, 2)
        y1, e1, k = step_plain(x, v, h, gl)
        y2, e2, s = step_rkn(x, v, h, gl)
        worst = max(worst, np.abs(y1 - y2).max(), np.abs(e1 - e2).max(), max(abs(k[j][1] - s[j]) for j in range(1, 8)))
        # dense-output polynomials from the s_j alone
        for m in range(3):
            Px = sum(RR[j, m] * k[j][0] for j in range(1, 8)); Pv = sum(RR[j, m] * k[j][1] for j in range(1, 8))
            Px2 = RS[m] * v + h * sum(RA[m, l] * s[l] for l in range(1, 7)); Pv2 = sum(RR[j, m] * s[j] for j in range(1, 8))
            worst = max(worst, abs(Px - Px2), abs(Pv - Pv2))
    print("max |plain − Nyström| over 200 random steps:", worst)
    assert worst < 2e-11   # (float64 round-off through coefficients of size 90)
    np.set_printoptions(precision=17, linewidth=200)
    print("C", C[1:]); print("Abar"); print(Abar[1:, 1:7]); print("Et", Et[1:7]); print("RS", RS); print("RA"); print(RA[:, 1:7])

"""Forward pendulum kernel time vs number of save points (same adaptive steps): what do the saves cost?"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latentdiffeq_amd as la
from oracle import oracle as O
B = int(os.environ.get("PB", 256))
z0, L = O.pendulum_inputs(B)
dec = la.Decoder(la.GOKU_basic(), (None, la.Pendulum(), None))
z0t = torch.tensor(z0.T.copy(), device="cuda"); tht = torch.tensor(L.T.copy(), device="cuda")
for T in (2, 3, 6, 11, 26, 50, 99):
    ts = np.linspace(0.0, 2.45, T)
    with torch.no_grad():
        for _ in range(5): la.diffeq_layer(dec, (z0t, tht), ts)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): la.diffeq_layer(dec, (z0t, tht), ts)
        e1.record(); torch.cuda.synchronize()
    print("T=%3d  %.1f us per forward call (incl. host)" % (T, e0.elapsed_time(e1) * 10))

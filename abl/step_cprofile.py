import sys, os, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latentdiffeq_amd as M
from latentdiffeq_amd import train as TR
torch.manual_seed(0)
B, T, NI = 256, 50, 784
mt, diffeq = M.GOKU_basic(), M.Pendulum()
enc, dec = TR.default_layers(mt, NI, diffeq, device="cuda")
with torch.no_grad():
    dec[0][1]._dense[-1].bias.fill_(1.0)
model = TR.LatentDiffEqModel(mt, enc, dec)
opt = TR.FluxADAMW(model.parameters(), lr=1e-3, decay=1e-10)
x = torch.rand(T, B, NI, device="cuda").permute(2, 1, 0); ts = np.arange(T) * 0.05
def step():
    opt.zero_grad(set_to_none=True)
    loss = TR.loss_batch(model, x, ts, 1e-3, True)
    loss.backward(); opt.step(); model.refresh_weights()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45); st.sort_stats("cumulative").print_stats(45)

"""Whole goku_step captured in a HIP graph (torch.cuda.CUDAGraph): forward + pullback + AdamW replayed as one launch."""
import sys, os, time
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
params = model.parameters()
opt = torch.optim.AdamW(params, lr=1e-3, capturable=True)
x = torch.rand(NI, B, T, device="cuda"); ts = np.arange(T) * 0.05
loss_out = torch.zeros((), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    loss = TR.loss_batch(model, x, ts, 1e-3, True)
    loss.backward(); opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    l = step()
    loss_out.copy_(l.detach())
torch.cuda.synchronize()
before = float(loss_out)
for _ in range(10): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("graph replay per step %.3f ms; loss %.5f -> %.5f" % ((t1 - t0) / 100 * 1e3, before, float(loss_out)))

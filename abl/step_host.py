"""How long does the HOST take to enqueue one goku_step (no synchronisation), vs the synchronised step time?"""
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
opt = TR.FluxADAMW(params, lr=1e-3, decay=1e-10)
x = torch.rand(NI, B, T, device="cuda"); ts = np.arange(T) * 0.05
def step():
    opt.zero_grad(set_to_none=True)
    loss = TR.loss_batch(model, x, ts, 1e-3, True)
    loss.backward(); opt.step(); model.refresh_weights()
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per step %.3f ms; total per step %.3f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
# the same with an EMPTY queue (so that the host can never be throttled by the device): n steps after a synchronise
for n in (1, 2, 4):
    best = (1e9, 0)
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        best = min(best, ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    print("n=%d: host %.3f ms/step, host+drain %.3f ms/step" % (n, best[0], best[1]))

"""Whole goku_step (encoder → sample/KL → decoder → loss → pullback → Flux-flavour ADAMW → weight hand-over) captured in ONE HIP graph
(torch.cuda.CUDAGraph) and replayed, against the eager step: per-step time, and the loss after k steps bit for bit.
    python abl/step_graph.py [f32|mixed] [B]          (LDE_BRANCH_STREAMS=0/1 picks the encode() variant that is captured)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latentdiffeq_amd as M
from latentdiffeq_amd import _lib as L
from latentdiffeq_amd.chain import decode, default_decoder_layers
from latentdiffeq_amd.loss import reconstruction_loss, sample_with_kl
from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode
from latentdiffeq_amd.train import FluxADAMW

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T, NI = 50, 784
dev = torch.device("cuda", 0)


def build():
    torch.manual_seed(100)
    mt, diffeq = M.GOKU_basic(), M.Pendulum()
    enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
    dec = M.Decoder(mt, default_decoder_layers(mt, NI, diffeq, device=dev))
    with torch.no_grad():
        dec.latent_out[1]._dense[-1].bias.fill_(1.0)
    mods = [enc.feature_extractor, *enc.pattern_extractor, *enc.latent_in, *dec.latent_out, dec.reconstructor]
    if dtype == "mixed":
        for m in (enc.feature_extractor, *enc.latent_in, *dec.latent_out, dec.reconstructor):
            m.set_dtype("bf16")
    params = [p for m in mods for p in m.parameters()]
    opt = FluxADAMW(params, lr=1e-3, decay=1e-10, capturable=True)
    return enc, dec, mods, opt


torch.manual_seed(1000)
x = torch.rand(T, B, NI, device=dev).permute(2, 1, 0)
eps_seed = torch.zeros((), device=dev)
ts = np.arange(T) * 0.05


def make_step(enc, dec, mods, opt):
    def step():
        opt.zero_grad(set_to_none=True)
        mu, logvar = encode(enc, x)
        l_tilde, bkl = sample_with_kl(mu, logvar, 1e-3, B)
        x_hat, _, _ = decode(dec, l_tilde, ts)
        loss = reconstruction_loss(x, x_hat, B, plus=bkl)
        loss.backward()
        opt.step()
        L.refresh_weights(mods)
        return loss
    return step


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# eager
step_e = make_step(*build())
torch.manual_seed(7)
for _ in range(10):
    le = step_e()
t_e = timed(step_e, 50)
# graph
enc, dec, mods, opt = build()
step_g = make_step(enc, dec, mods, opt)
torch.manual_seed(7)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step_g()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
loss_out = torch.zeros((), device=dev)
with torch.cuda.graph(g):
    l = step_g()
    loss_out.copy_(l.detach())
torch.cuda.synchronize()
for _ in range(6):
    g.replay()
t_g = timed(g.replay, 50)
print(f"{dtype} B={B} branch_streams={os.environ.get('LDE_BRANCH_STREAMS', '1')}: eager {t_e:.3f} ms/step, graph replay {t_g:.3f} ms/step; "
      f"loss eager(60 steps) {float(le):.6f} graph {float(loss_out):.6f}")

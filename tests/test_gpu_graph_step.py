"""GPU: the whole GOKU training step captured in ONE hipGraph (train.GraphedStep: encoder → sample/KL → decoder → loss → pullback →
Flux-flavour ADAMW with its step count in device memory → weight hand-over) replays to the SAME numbers as the eager step: with ε
drawn once and shared, the loss of every step and every parameter after k steps are equal bit for bit
[REF examples/pendulum_friction-less/model_train.jl:186-204: the loop body]. Runs in a subprocess: the encoder's side streams must be
off (recurrent._BRANCH_STREAMS = False) for the capture."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["LDE_ROOT"])
import numpy as np, torch
import latentdiffeq_amd as M
from latentdiffeq_amd import _lib as L
from latentdiffeq_amd import recurrent as _rec
_rec._BRANCH_STREAMS = False          # the captured step runs on one stream (module attribute: the package reads no environment variable)
from latentdiffeq_amd.chain import decode, default_decoder_layers
from latentdiffeq_amd.loss import reconstruction_loss, sample_with_kl
from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode
from latentdiffeq_amd.train import FluxADAMW, GraphedStep
dtype = sys.argv[1]
split = len(sys.argv) > 2 and sys.argv[2] == "split"
discrete = len(sys.argv) > 2 and sys.argv[2] == "discrete"      # the solve's pullback as LDE_SENSE_DISCRETE: a step record per autograd node
if split:   # several GPUs in miniature: a ONE-rank RCCL group, the gradient all-reduce forced on (dist.FORCE_ALLREDUCE)
    import torch.distributed as dist
    from latentdiffeq_amd import dist as _ldist
    _ldist.FORCE_ALLREDUCE = True
    from latentdiffeq_amd.dist import FlatGradAllReduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
B, T, NI = 64, 20, 784
dev = torch.device("cuda", 0)
def build():
    torch.manual_seed(100)
    mt, diffeq = M.GOKU_basic(), (M.Pendulum(sensealg=M.DiscreteSensitivity()) if discrete else M.Pendulum())
    enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
    dec = M.Decoder(mt, default_decoder_layers(mt, NI, diffeq, device=dev))
    with torch.no_grad():
        dec.latent_out[1]._dense[-1].bias.fill_(1.0)
    mods = [enc.feature_extractor, *enc.pattern_extractor, *enc.latent_in, *dec.latent_out, dec.reconstructor]
    if dtype == "mixed":
        for m in (enc.feature_extractor, *enc.latent_in, *dec.latent_out, dec.reconstructor):
            m.set_dtype("bf16")
    params = [p for m in mods for p in m.parameters()]
    return enc, dec, mods, params, FluxADAMW(params, lr=1e-3, decay=1e-10, capturable=True)
torch.manual_seed(5)
xs = [torch.rand(T, B, NI, device=dev).permute(2, 1, 0) for _ in range(3)]        # three different minibatches, cycled
eps = (torch.randn(16, B, device=dev), torch.randn(16, B, device=dev))            # ε of the two latent parts, drawn once
ts = np.arange(T) * 0.05
def make(enc, dec, mods, params, opt, x):
    sync = FlatGradAllReduce(params) if split else (lambda: None)
    def a():
        opt.zero_grad(set_to_none=True)
        mu, logvar = encode(enc, x)
        l_tilde, bkl = sample_with_kl(mu, logvar, 1e-3, B, eps=eps)
        x_hat, _, _ = decode(dec, l_tilde, ts)
        loss = reconstruction_loss(x, x_hat, B, plus=bkl)
        loss.backward()
        return loss
    def b():
        opt.step()
        L.refresh_weights(mods)
    def step():
        loss = a()
        sync()
        b()
        return loss
    step.a, step.b, step.sync = a, b, sync
    return step
K, W = 7, 3
# eager: W warm-up steps on minibatch 0 (what GraphedStep's warm-up does), then K steps cycling the minibatches
e = build(); xe = xs[0].clone(); se = make(*e, xe)
for _ in range(W): se()
le = []
for k in range(K):
    xe.copy_(xs[k % 3]); le.append(float(se()))
# graph: W eager warm-up steps, the capture (which records and runs nothing), then K replays, each copying its minibatch into the
# captured input tensor
g = build(); xg = xs[0].clone(); sg = make(*g, xg)
# (split: graph (zero_grad … backward) · eager all-reduce · graph (update, hand-over) — what a step with several GPUs replays)
gs = GraphedStep(sg.a, static_inputs=[xg], warmup=W, between=sg.sync, fn2=sg.b) if split else GraphedStep(sg, static_inputs=[xg], warmup=W)
lg = [float(gs.replay(xs[k % 3])) for k in range(K)]
torch.cuda.synchronize()
assert le == lg, (le, lg)
for a, b in zip(e[3], g[3]):
    assert torch.equal(a, b)
assert int(g[4]._step_dev) == W + K
if split:
    dist.destroy_process_group()
print("ok", dtype, le[-1])
'''


@pytest.mark.parametrize("dtype", ["f32", "mixed"])
def test_graph_replay_equals_eager_step(tmp_path, dtype):
    f = tmp_path / "graph_step.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, LDE_ROOT=ROOT)
    r = subprocess.run([sys.executable, str(f), dtype], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_graph_replay_with_the_discrete_sensitivity_equals_eager(tmp_path):
    """The same with `Pendulum(sensealg=DiscreteSensitivity())` — the reference's GOKU default gradient definition: the step record of the
    captured solve lives in the graph's memory pool, forward writes it and the pullback sweeps it on every replay."""
    f = tmp_path / "graph_step_discrete.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, LDE_ROOT=ROOT)
    r = subprocess.run([sys.executable, str(f), "f32", "discrete"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_split_graph_step_with_a_process_group_equals_eager(tmp_path):
    """The several-GPU form of the captured step — graph (zero_grad … backward) · eager gradient all-reduce over RCCL · graph (update,
    weight hand-over) — with a ONE-rank group on this box's GPU: the same losses and parameters as the eager loop with the same
    all-reduce, bit for bit."""
    f = tmp_path / "graph_step_split.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, LDE_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(f), "mixed", "split"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


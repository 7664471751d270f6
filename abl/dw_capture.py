"""Which pullback leaves the weight-gradient stream unjoined under stream capture? (set_async_weight_gradients + torch.cuda.graph)
    python abl/dw_capture.py chain|rnn|enc|step"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import latentdiffeq_amd as M
from latentdiffeq_amd import _lib as L
from latentdiffeq_amd.chain import Chain, Dense, SkipConnection
from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode

what = sys.argv[1] if len(sys.argv) > 1 else "chain"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
L.set_async_weight_gradients(True, dev)
B, T, NI = 64, 50, 784
if what == "chain":
    m = Chain(Dense(2, 64, "relu"), SkipConnection(Dense(64, 64, "relu")), Dense(64, 48, "sigmoid")).to(dev)
    x = torch.randn(2, B, T, device=dev, requires_grad=True)
    def step():
        y = m(x)
        (y * y).sum().backward()
        L.join_weight_gradients()
elif what in ("rnn", "enc"):
    mt = M.GOKU_basic()
    enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
    from latentdiffeq_amd import recurrent as _rec
    _rec._BRANCH_STREAMS = False
    if what == "rnn":
        _rec._ENCODER_FUSED = False
    x = torch.rand(T, B, NI, device=dev).permute(2, 1, 0)
    def step():
        mu, lv = encode(enc, x)
        s = sum((a * a).sum() for a in mu) + sum((a * a).sum() for a in lv)
        s.backward()
        L.join_weight_gradients()
elif what in ("fe", "pe", "li"):
    mt = M.GOKU_basic()
    enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
    from latentdiffeq_amd import recurrent as _rec
    _rec._BRANCH_STREAMS = False
    fe, pes, lis = enc.feature_extractor, enc.pattern_extractor, enc.latent_in
    x = torch.rand(T, B, NI, device=dev).permute(2, 1, 0)
    with torch.no_grad():
        yfe = fe(x)
        ype = pes[0](yfe)
    yfe = yfe.detach().requires_grad_(True)
    ype = ype.detach().requires_grad_(True)
    def step():
        if what == "fe":
            y = fe(x)
            (y * y).sum().backward()
        elif what == "pe":
            s = sum((pe(yfe) ** 2).sum() for pe in pes)
            s.backward()
        else:
            s = sum((li(ype) ** 2).sum() for li in lis[:1])
            s.backward()
        L.join_weight_gradients()
for _ in range(3):
    step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g, stream=s):
        step()
    g.replay()
    torch.cuda.synchronize()
    print(what, "captured and replayed")
except Exception as e:
    print(what, "FAILED:", str(e).splitlines()[0])

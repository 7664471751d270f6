"""Which part of the branch-stream encoder breaks hipGraph capture? python abl/graph_bisect.py <variant>
   variants: enc_fwd | enc_fwd_bwd | chain_side_fwd | chain_side_fwd_bwd | rnn_side_fwd_bwd"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latentdiffeq_amd as M
from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode
v = sys.argv[1]
dev = torch.device("cuda", 0)
B, T, NI = 256, 50, 784
torch.manual_seed(1)
mt = M.GOKU_basic()
enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
x = torch.rand(T, B, NI, device=dev).permute(2, 1, 0)
params = [p for m in (enc.feature_extractor, *enc.pattern_extractor, *enc.latent_in) for p in m.parameters()]


def fn():
    if v.startswith("enc"):
        mu, lv = encode(enc, x)
        out = sum(t.sum() for t in (*mu, *lv))
    else:
        fe = enc.feature_extractor(x)
        main = torch.cuda.current_stream()
        st = side
        st.wait_stream(main)
        with torch.cuda.stream(st):
            if v.startswith("chain"):
                y = enc.latent_in[0](fe[:16, :, 0].contiguous())
            else:
                y = enc.pattern_extractor[0](fe)
        main.wait_stream(st)
        out = y.sum()
    if v.endswith("bwd"):
        for p in params:
            p.grad = None
        out.backward()
    return out


side = torch.cuda.Stream()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o = fn()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(v, "captured and replayed:", float(o))

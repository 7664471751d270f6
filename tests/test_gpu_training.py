"""GPU: the gradients are good enough to LEARN with — a miniature of what the reference's training loop does with this
layer [REF examples/pendulum_friction-less/model_train.jl:195-201]: recover the pendulum length and initial state of
every trajectory from its observed latent path by gradient descent through diffeq_layer, and fit a NODE's weights
(which the reference never trains) to a linear vector field."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_recover_pendulum_length_and_initial_state_by_gradient_descent():
    import torch
    import latentdiffeq_amd as la
    torch.manual_seed(0)
    B, T = 64, 50
    z0_true, L_true = O.pendulum_inputs(B, seed=3)
    ts = O.time_grid(T)
    dec = la.Decoder(la.GOKU_basic(), (None, la.Pendulum(abstol=1e-6, reltol=1e-5), None))
    dev = "cuda"
    with torch.no_grad():
        target = la.diffeq_layer(dec, (torch.tensor(z0_true.T.copy(), device=dev), torch.tensor(L_true.T.copy(), device=dev)), ts)
    # start from wrong guesses; θ̂ goes through softplus like the reference's latent_out head [REF GOKU.jl:252-256]
    z0 = torch.tensor((z0_true * 0.5).T.copy(), device=dev, requires_grad=True)
    raw = torch.full((1, B), 1.0, device=dev, requires_grad=True)
    opt = torch.optim.Adam([z0, raw], lr=0.05)
    losses = []
    for it in range(300):
        opt.zero_grad()
        zhat = la.diffeq_layer(dec, (z0, torch.nn.functional.softplus(raw) + 0.5), ts)
        loss = ((zhat - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    L_fit = (torch.nn.functional.softplus(raw) + 0.5).detach().cpu().numpy().T
    assert losses[-1] < 1e-3 * losses[0], (losses[0], losses[-1])
    assert np.median(np.abs(L_fit - L_true)) < 0.02 and np.abs(z0.detach().cpu().numpy().T - z0_true).max() < 0.05


def test_node_weights_learn_a_rotation_field():
    import torch
    import latentdiffeq_amd as la
    torch.manual_seed(1)
    D, B, T, w = 2, 48, 25, 1.5
    ts = O.time_grid(T, 0.1)
    rng = np.random.default_rng(0)
    z0 = rng.standard_normal((B, D)).astype(np.float32)
    c, s = np.cos(w * ts)[:, None], np.sin(w * ts)[:, None]
    target = np.stack([c * z0[None, :, 0] - s * z0[None, :, 1], s * z0[None, :, 0] + c * z0[None, :, 1]], axis=2)  # (T,B,2)
    node = la.NODE(D, hidden_dim=32, device="cuda", activation="tanh", abstol=1e-5, reltol=1e-4)
    dec = la.Decoder(la.LatentODE(), (None, node, None))
    tgt = torch.tensor(target, device="cuda", dtype=torch.float32).permute(2, 1, 0)
    z0t = torch.tensor(z0.T.copy(), device="cuda")
    opt = torch.optim.Adam(node.dudt.parameters(), lr=0.01)
    first = last = None
    for it in range(250):
        opt.zero_grad()
        loss = ((la.diffeq_layer(dec, z0t, ts) - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    assert last < 0.05 * first, (first, last)


def test_whole_goku_model_trains_on_synthetic_frames():
    """The full model on this library — encoder (dense chain + RNN / LSTM stacks + latent_in) → sample → latent_out →
    pendulum solve → reconstructor — through torch autograd with AdamW: the loss must fall on a small synthetic batch
    (frames rendered from true pendulum trajectories, 8×8 pixels)."""
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd.chain import decode, default_decoder_layers
    from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode, sample
    torch.manual_seed(0)
    B, T, side = 32, 20, 8
    NI = side * side
    dev = "cuda"
    # ground truth: pendulum angle → a bright blob at the bob position  [REF examples/pendulum_friction-less/create_data.jl:90-111] (simplified)
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    ztrue, _, _ = O.Oracle("f64").forward(O.make_desc(abstol=1e-9, reltol=1e-9), z0, L, ts)      # (T, B, 2)
    ang = torch.tensor(ztrue[:, :, 0].T.copy(), dtype=torch.float32)                              # (B, T)
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, side), torch.linspace(-1, 1, side), indexing="ij")
    bx, by = 0.7 * torch.sin(ang), 0.7 * torch.cos(ang)
    frames = torch.exp(-(((gx[None, None] - bx[..., None, None]) ** 2 + (gy[None, None] - by[..., None, None]) ** 2) / 0.08))
    x = frames.reshape(B, T, NI).permute(2, 0, 1).contiguous().to(dev)                            # [NI, B, T]
    mt = M.GOKU_basic()
    diffeq = M.Pendulum()
    enc = Encoder(mt, default_encoder_layers(mt, NI, hidden_dim_resnet=64, device=dev))
    dec = M.Decoder(mt, default_decoder_layers(mt, NI, diffeq, hidden_dim_resnet=64, latent_to_diffeq_dim=64, device=dev))
    with torch.no_grad():
        dec.latent_out[1]._dense[-1].bias.fill_(1.0)
    mods = [enc.feature_extractor, *enc.pattern_extractor, *enc.latent_in, *dec.latent_out, dec.reconstructor]
    params = [p for m in mods for p in m.parameters()]
    opt = torch.optim.AdamW(params, lr=2e-3, weight_decay=1e-10)
    losses = []
    for it in range(60):
        opt.zero_grad(set_to_none=True)
        mu, logvar = encode(enc, x)
        x_hat, z_hat, _ = decode(dec, sample(mu, logvar), ts)
        rec = ((x_hat - x) ** 2).mean()
        kl = sum((-0.5 * (1 + s - m ** 2 - torch.exp(s))).mean() for m, s in zip(mu, logvar))
        loss = rec + 1e-4 * kl
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        opt.step()
        losses.append(float(rec.detach()))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])


def _pendulum_frames(B, T, side, seed=1):
    import torch
    z0, L = O.pendulum_inputs(B, seed=seed)
    ts = O.time_grid(T)
    ztrue, _, _ = O.Oracle("f64").forward(O.make_desc(abstol=1e-9, reltol=1e-9), z0, L, ts)
    ang = torch.tensor(ztrue[:, :, 0].T.copy(), dtype=torch.float32)
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, side), torch.linspace(-1, 1, side), indexing="ij")
    bx, by = 0.7 * torch.sin(ang), 0.7 * torch.cos(ang)
    fr = torch.exp(-(((gx[None, None] - bx[..., None, None]) ** 2 + (gy[None, None] - by[..., None, None]) ** 2) / 0.08))
    return fr.reshape(B, T, side * side).permute(2, 0, 1).contiguous()          # [pixels, B, T]


@pytest.mark.parametrize("kind", ["goku", "latentode"])
def test_train_harness_runs_both_model_types(kind):
    """train() — cyclical β, random time windows, AdamW, validation loss after every minibatch, best weights — on the GOKU
    model and on the LatentODE model (whose NODE weights are trained here, unlike in the reference: SURVEY.md B2)."""
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd import train as TR
    torch.manual_seed(3)
    side, full, seq = 6, 24, 12
    data = _pendulum_frames(40, full, side).to("cuda")
    train_set, val_set = data[:, :32], data[:, 32:, :seq]
    if kind == "goku":
        mt, diffeq = M.GOKU_basic(), M.Pendulum()
    else:
        mt, diffeq = M.LatentODE(), M.NODE(4, hidden_dim=32, device="cuda", batching="per_trajectory")
    enc, dec = TR.default_layers(mt, side * side, diffeq, device="cuda", hidden_dim_resnet=48, latent_to_diffeq_dim=48)
    if kind == "goku":
        with torch.no_grad():
            dec[0][1]._dense[-1].bias.fill_(1.0)
    model = TR.LatentDiffEqModel(mt, enc, dec)
    loader = [train_set[:, :16], train_set[:, 16:]]
    hist, best = TR.train(model, loader, val_set, dt=0.05, epochs=12, seq_len=seq, full_seq_len=full, lr=2e-3, end_beta=1e-3,
                          n_cycle=2, ratio=0.5, rng=np.random.default_rng(0))
    losses = np.array([h[1] for h in hist])
    assert len(hist) == 24 and np.isfinite(losses).all() and best is not None and len(best) == len(model.parameters())
    assert losses[-4:].mean() < 0.8 * losses[:4].mean(), (losses[:4], losses[-4:])
    if kind == "latentode":
        assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in diffeq.dudt.parameters())


def test_generate_dataset_matches_the_float64_pendulum():
    """generate_dataset: latent trajectories from lde_forward (tight tolerance) against the float64 oracle; frames rendered from them."""
    import torch
    from latentdiffeq_amd.data import create_frames, generate_dataset
    latent, u0s, ps, frames = generate_dataset(n_traj=12, seed=4)
    T = 100
    assert latent.shape == (2, T, 12) and u0s.shape == (2, 12) and ps.shape == (1, 12) and frames.shape == (28, 28, T, 12)
    z64, _, _ = O.Oracle("f64").forward(O.make_desc(abstol=1e-11, reltol=1e-11), u0s.cpu().numpy().T, ps.cpu().numpy().T, 0.05 * np.arange(T))
    assert np.abs(latent.cpu().numpy().transpose(1, 2, 0) - z64).max() <= 2e-5
    assert float(frames.min()) >= 0 and float(frames.max()) <= 1
    again = create_frames(latent[0, 7, :3])
    assert torch.allclose(again, frames[:, :, 7, :3].permute(2, 0, 1))

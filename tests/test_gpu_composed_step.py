"""GPU: one whole GOKU training step — encode → sample → latent_out → pendulum solve → reconstructor → loss, and its pullback —
against the COMPOSITION of the CPU oracles (float64), module by module:

    x ─ feature extractor (oracle chain) ─ three recurrent stacks (oracle rnn) ─ four latent_in heads (oracle chain)
      ─ sample + β·KL (oracle loss) ─ latent_out (oracle chain ×2) ─ pendulum solve (oracle forward / adjoint)
      ─ reconstructor (oracle chain) ─ reconstruction loss (oracle loss)

    [REF src/models/LatentDiffEqModel.jl:63-75, :101-113], [REF src/models/GOKU.jl:19-72, :83-91, :98-130, :148, :155-163],
    [REF examples/pendulum_friction-less/model_train.jl:225-238]

tests/test_gpu_mixed_step.py compares the mixed-precision step with its own f32 twin; this test is what anchors that twin: the
f32 step of the product (every kernel of rows a, f-1, f-2, f-3 in one autograd graph, default GOKU layers, 784-pixel frames) equals
an independent restatement of the same composition — the loss to 1e-5, every module's parameter gradient to 2e-3 in relative L2
(float32 through ≈ 20 layers and a 49-interval adjoint against float64; relu units within round-off of zero may switch), the
latent trajectories to 2e-5. ε is handed to both sides (the in-kernel Philox stream has its own known-answer tests).
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _chain_desc(ch):
    return O.make_chain_desc(list(ch.sizes), list(ch.acts), list(ch.skips))


def _rnn_desc(rec):
    return O.make_rnn_desc(rec.code, list(rec.sizes), reverse=rec.reverse)    # (lde_cell_kind: the same codes on both sides)


def _w(m):
    return m.flat_weights().detach().cpu().numpy().astype(np.float64)


def test_f32_goku_step_equals_the_composed_oracles(o64):
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd.chain import decode, default_decoder_layers
    from latentdiffeq_amd.loss import reconstruction_loss, sample_with_kl
    from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode
    B, T, NI, beta = 64, 50, 784, 1e-3
    torch.manual_seed(100)
    dev = torch.device("cuda", 0)
    mt = M.GOKU_basic()
    diffeq = M.Pendulum(abstol=1e-6, reltol=1e-6)
    enc = Encoder(mt, default_encoder_layers(mt, NI, device=dev))
    dec = M.Decoder(mt, default_decoder_layers(mt, NI, diffeq, device=dev))
    with torch.no_grad():
        dec.latent_out[1]._dense[-1].bias.fill_(1.0)          # pendulum lengths inside the data range
    fe, (pe_z0, pe_f, pe_b), (li_mu_z0, li_ls_z0, li_mu_th, li_ls_th) = enc.feature_extractor, enc.pattern_extractor, enc.latent_in
    (lo_z0, lo_th), rec = dec.latent_out, dec.reconstructor
    mods = [fe, pe_z0, pe_f, pe_b, li_mu_z0, li_ls_z0, li_mu_th, li_ls_th, lo_z0, lo_th, rec]
    names = ["feature_extractor", "pe_z0", "pe_th_fwd", "pe_th_bwd", "li_mu_z0", "li_ls_z0", "li_mu_th", "li_ls_th", "lo_z0", "lo_th", "reconstructor"]
    rng = np.random.default_rng(7)
    x_tbp = rng.uniform(0, 1, (T, B, NI)).astype(np.float32)                 # memory (T, B, pixels) == the reference's [pixels × B × T]
    eps_z0, eps_th = rng.standard_normal((B, 16)).astype(np.float32), rng.standard_normal((B, 16)).astype(np.float32)
    ts = np.arange(T) * 0.05

    # ---- the product: one autograd graph ------------------------------------------------------------------------------------------
    x = torch.from_numpy(x_tbp).to(dev).permute(2, 1, 0)
    mu, logvar = encode(enc, x)
    eps = (torch.from_numpy(eps_z0).to(dev).t(), torch.from_numpy(eps_th).to(dev).t())
    l_tilde, bkl = sample_with_kl(mu, logvar, beta, B, eps=eps)
    x_hat, z_hat, l_hat = decode(dec, l_tilde, ts)
    loss = reconstruction_loss(x, x_hat, B, plus=bkl)
    loss.backward()
    torch.cuda.synchronize()
    g_gpu = [m.flat_weights().grad.detach().cpu().numpy().astype(np.float64) for m in mods]
    z_gpu = z_hat.detach().permute(2, 1, 0).cpu().numpy()                     # (T, B, 2)

    # ---- the composition of the oracles (float64) ------------------------------------------------------------------------------------
    W = {n: _w(m) for n, m in zip(names, mods)}
    d_fe, d_rec = _chain_desc(fe), _chain_desc(rec)
    d_li = [_chain_desc(m) for m in (li_mu_z0, li_ls_z0, li_mu_th, li_ls_th)]
    d_lo = [_chain_desc(lo_z0), _chain_desc(lo_th)]
    d_pe = [_rnn_desc(m) for m in (pe_z0, pe_f, pe_b)]
    X = x_tbp.reshape(T * B, NI).astype(np.float64)
    fe_out = o64.chain_forward(d_fe, W["feature_extractor"], X)              # (T·B, 32)
    fe3 = fe_out.reshape(T, B, -1)
    y_z0 = o64.rnn_forward(d_pe[0], W["pe_z0"], fe3)
    y_f = o64.rnn_forward(d_pe[1], W["pe_th_fwd"], fe3)
    y_b = o64.rnn_forward(d_pe[2], W["pe_th_bwd"], fe3)
    y_th = np.concatenate([y_f, y_b], axis=1)                                # vcat(pe_forward, pe_backward)  [REF GOKU.jl:47]
    mu_z0 = o64.chain_forward(d_li[0], W["li_mu_z0"], y_z0)
    ls_z0 = o64.chain_forward(d_li[1], W["li_ls_z0"], y_z0)
    mu_th = o64.chain_forward(d_li[2], W["li_mu_th"], y_th)
    ls_th = o64.chain_forward(d_li[3], W["li_ls_th"], y_th)
    l_z0 = o64.sample_forward(mu_z0, ls_z0, eps_z0)
    l_th = o64.sample_forward(mu_th, ls_th, eps_th)
    kl = o64.kl_forward(mu_z0, ls_z0, beta / B) + o64.kl_forward(mu_th, ls_th, beta / B)
    z0h = o64.chain_forward(d_lo[0], W["lo_z0"], l_z0)                       # (B, 2)
    thh = o64.chain_forward(d_lo[1], W["lo_th"], l_th)                       # (B, 1)
    d_solve = O.make_desc(abstol=1e-6, reltol=1e-6)
    z, ret, _ = o64.forward(d_solve, z0h, thh, ts)
    assert (ret == 0).all()
    xh = o64.chain_forward(d_rec, W["reconstructor"], z.reshape(T * B, 2))
    scale = 1.0 / (B * T)
    loss_o = o64.mse_forward(X, xh, scale) + kl
    # pullback
    dxh = o64.mse_backward(X, xh, scale)
    dz, gW_rec = o64.chain_backward(d_rec, W["reconstructor"], z.reshape(T * B, 2), dxh)
    dz0h, dthh, _, _ = o64.adjoint(d_solve, z, thh, ts, dz.reshape(T, B, 2))
    dl_z0, gW_lo_z0 = o64.chain_backward(d_lo[0], W["lo_z0"], l_z0, dz0h)
    dl_th, gW_lo_th = o64.chain_backward(d_lo[1], W["lo_th"], l_th, dthh)

    def head_back(mu_, ls_, eps_, dl_):
        dmu_s, dls_s = o64.sample_backward(ls_, eps_, dl_)
        dmu_k, dls_k = o64.kl_backward(mu_, ls_, beta / B)
        return dmu_s + dmu_k, dls_s + dls_k
    dmu_z0, dls_z0 = head_back(mu_z0, ls_z0, eps_z0, dl_z0)
    dmu_th, dls_th = head_back(mu_th, ls_th, eps_th, dl_th)
    dy1, gW_li0 = o64.chain_backward(d_li[0], W["li_mu_z0"], y_z0, dmu_z0)
    dy2, gW_li1 = o64.chain_backward(d_li[1], W["li_ls_z0"], y_z0, dls_z0)
    dy3, gW_li2 = o64.chain_backward(d_li[2], W["li_mu_th"], y_th, dmu_th)
    dy4, gW_li3 = o64.chain_backward(d_li[3], W["li_ls_th"], y_th, dls_th)
    dy_z0, dy_th = dy1 + dy2, dy3 + dy4
    h = y_f.shape[1]
    dfe_a, gW_pe0 = o64.rnn_backward(d_pe[0], W["pe_z0"], fe3, dy_z0)
    dfe_b, gW_pe1 = o64.rnn_backward(d_pe[1], W["pe_th_fwd"], fe3, np.ascontiguousarray(dy_th[:, :h]))
    dfe_c, gW_pe2 = o64.rnn_backward(d_pe[2], W["pe_th_bwd"], fe3, np.ascontiguousarray(dy_th[:, h:]))
    dfe = (dfe_a + dfe_b + dfe_c).reshape(T * B, -1)
    _, gW_fe = o64.chain_backward(d_fe, W["feature_extractor"], X, dfe, need_dx=False)
    g_orc = [gW_fe, gW_pe0, gW_pe1, gW_pe2, gW_li0, gW_li1, gW_li2, gW_li3, gW_lo_z0, gW_lo_th, gW_rec]

    # ---- the comparison --------------------------------------------------------------------------------------------------------------
    assert abs(float(loss.detach()) - loss_o) <= 1e-5 * abs(loss_o), (float(loss.detach()), loss_o)
    assert np.abs(z_gpu - z).max() <= 2e-5, np.abs(z_gpu - z).max()
    for n, a, b in zip(names, g_gpu, g_orc):
        assert a.shape == b.shape, n
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
        assert np.isfinite(a).all() and rel <= 2e-3, (n, rel)

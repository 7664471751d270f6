"""CPU: the loss-term oracle (oracle/lde_loss_oracle.c, scope row f-3) against closed forms worked out by hand from the
reference's definitions and against torch autograd on CPU (an independent implementation of the same formulas)
[REF src/utils/utils.jl:15-49], [REF src/models/GOKU.jl:155-163], [REF examples/pendulum_friction-less/model_train.jl:225-238]."""
import numpy as np
import pytest
import torch

from oracle import oracle as O


@pytest.fixture(scope="module")
def o64():
    return O.Oracle("f64")


@pytest.fixture(scope="module")
def o32():
    return O.Oracle("f32")


def test_known_answers(o64):
    mu = np.array([[0.5, -1.0], [0.0, 2.0]])
    ls = np.array([[0.0, 0.3], [-0.2, 0.1]])
    # kl(0.5, 0) = (1 + 0.25 − 0 − 1)/2 = 0.125;  kl(0, −0.2) = (e^−0.2 + 0.2 − 1)/2
    e = np.array([[0.125, (np.exp(0.3) + 1 - 0.3 - 1) / 2], [(np.exp(-0.2) + 0.2 - 1) / 2, (np.exp(0.1) + 4 - 0.1 - 1) / 2]])
    assert np.isclose(o64.kl_forward(mu, ls, 1 / 2), e.sum() / 2, rtol=1e-14)             # vector_kl: / batch size (columns)
    x = np.arange(24, dtype=np.float64).reshape(3, 2, 4)                                  # [pixels, B, T]
    want = ((0.5 * x) ** 2).mean(axis=(1, 2)).sum()                                       # sum(mean(·, dims=(2,3)))
    assert np.isclose(o64.mse_forward(x, 0.5 * x, 1 / 8), want, rtol=1e-14)
    eps = np.array([[1.0, -2.0], [0.5, 0.0]])
    assert np.allclose(o64.sample_forward(mu, ls, eps), mu + eps * np.exp(ls / 2), rtol=1e-15)
    assert np.array_equal(o64.sample_forward(mu, ls, np.zeros_like(mu)), mu)               # ε = 0 ⇒ l̃ = μ


@pytest.mark.parametrize("n,B", [(1, 1), (16 * 7, 7), (16 * 256, 256), (10007, 13)])
def test_oracle_agrees_with_torch_autograd(o64, o32, n, B):
    rng = np.random.default_rng(n)
    mu, ls, eps, ct = (rng.standard_normal(n) for _ in range(4))
    ls *= 0.5
    tm, tl = torch.tensor(mu, requires_grad=True), torch.tensor(ls, requires_grad=True)
    te, tc = torch.tensor(eps), torch.tensor(ct)
    l = tm + te * torch.exp(tl / 2)
    (l * tc).sum().backward()
    assert np.allclose(o64.sample_forward(mu, ls, eps), l.detach().numpy(), rtol=1e-13, atol=1e-14)
    dmu, dlv = o64.sample_backward(ls, eps, ct)
    assert np.allclose(dmu, tm.grad.numpy(), rtol=1e-13) and np.allclose(dlv, tl.grad.numpy(), rtol=1e-12, atol=1e-14)
    tm.grad = tl.grad = None
    k = ((torch.exp(tl) + tm ** 2 - tl - 1) / 2).sum() / B
    (1.7 * k).backward()
    assert np.isclose(o64.kl_forward(mu, ls, 1 / B), float(k), rtol=1e-12)
    dmu, dlv = o64.kl_backward(mu, ls, 1 / B, 1.7)
    assert np.allclose(dmu, tm.grad.numpy(), rtol=1e-12, atol=1e-15) and np.allclose(dlv, tl.grad.numpy(), rtol=1e-12, atol=1e-15)
    tx, th = torch.tensor(mu), torch.tensor(eps, requires_grad=True)
    m = ((tx - th) ** 2).sum() / B
    (0.3 * m).backward()
    assert np.isclose(o64.mse_forward(mu, eps, 1 / B), float(m), rtol=1e-12)
    assert np.allclose(o64.mse_backward(mu, eps, 1 / B, 0.3), th.grad.numpy(), rtol=1e-12, atol=1e-15)
    # the f32 build: storage precision only (sums in double)
    assert np.isclose(o32.kl_forward(mu, ls, 1 / B), float(k), rtol=2e-6)
    assert np.isclose(o32.mse_forward(mu, eps, 1 / B), float(m), rtol=2e-6)


def test_empty(o64):
    z = np.zeros(0)
    assert o64.kl_forward(z, z, 1.0) == 0.0 and o64.mse_forward(z, z, 1.0) == 0.0

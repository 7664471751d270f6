"""Synthetic pendulum videos (scope row f-4): the data the reference's example trains on.

    reference                                                                      here
    -----------------------------------------------------------------------------  -------------------------------
    generate_dataset(; diffeq = Pendulum(), tspan, dt, u₀_range, p₀_range, n_traj, seed, high_dim_args)
                                   [REF examples/pendulum_friction-less/create_data.jl:13-57]   generate_dataset
    maketrajectories / frame / create_frames (Luxor)   [REF create_data.jl:66-117]               create_frames

The latent trajectories are solved by lde_forward (the product's own solver, on the GPU). The frames are an
approximation of the Luxor drawing, not a pixel copy: Cairo's anti-aliasing and the "|" text glyph of `drawrod`
[REF create_data.jl:84-87] cannot be reproduced without Cairo, so a frame here is the supersampled coverage of the same
geometry — bob disc and pivot disc of radius 1.75, a rod 3.75 thick between them, the pivot's black inner disc — on the
same 28×28 canvas (origin at the centre, y down, pivot at (0, −8.5), pendulum length 19 px). Julia's `Random.seed!`
stream is not reproducible either: numpy's `default_rng(seed)` draws the initial states and lengths.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch


def create_frames(theta: torch.Tensor, pendulumlength: float = 19.0, radius: float = 1.75, rodthickness: float = 3.75,
                  w: int = 28, h: int = 28, ss: int = 4) -> torch.Tensor:
    """theta [...]: pendulum angles → frames [..., h, w] in [0, 1] (supersampled ss×ss coverage)."""
    dev = theta.device
    sub = (torch.arange(ss, device=dev, dtype=torch.float32) + 0.5) / ss
    xs = (torch.arange(w, device=dev, dtype=torch.float32)[:, None] + sub[None, :]).reshape(-1) - w / 2      # [w·ss]
    ys = (torch.arange(h, device=dev, dtype=torch.float32)[:, None] + sub[None, :]).reshape(-1) - h / 2      # [h·ss]
    Y, X = torch.meshgrid(ys, xs, indexing="ij")                                                                # [h·ss, w·ss]
    ox, oy = 0.0, -8.5                                                                                          # offset = Point(0, −8.5)
    th = theta.to(torch.float32)[..., None, None]
    px = ox + pendulumlength * torch.cos(math.pi / 2 + th)                                                      # angle1 = π/2 + θ
    py = oy + pendulumlength * torch.sin(math.pi / 2 + th)
    bob = ((X - px) ** 2 + (Y - py) ** 2) <= radius ** 2
    piv = ((X - ox) ** 2 + (Y - oy) ** 2) <= radius ** 2
    dx, dy = px - ox, py - oy
    tt = (((X - ox) * dx + (Y - oy) * dy) / (dx * dx + dy * dy)).clamp(0, 1)
    rod = ((X - (ox + tt * dx)) ** 2 + (Y - (oy + tt * dy)) ** 2) <= (rodthickness / 2) ** 2
    inner = ((X - ox) ** 2 + (Y - oy) ** 2) <= (radius / 2) ** 2
    img = ((bob | piv | rod) & ~inner).to(torch.float32)
    lead = img.shape[:-2]
    return img.reshape(*lead, h, ss, w, ss).mean(dim=(-3, -1))


def generate_dataset(diffeq=None, tspan: Tuple[float, float] = (0.0, 4.95), dt: float = 0.05,
                     u0_range=((-math.pi / 6, math.pi / 6), (-math.pi / 3, math.pi / 3)), p0_range: Tuple[float, float] = (1.0, 2.0),
                     n_traj: int = 450, seed: int = 1, high_dim_args=(19.0, 1.75, 3.75), device: Optional[str] = None):
    """(latent_data [2, T, n], u0s [2, n], ps [1, n], high_dim_data [28, 28, T, n])  [REF create_data.jl:31-57]."""
    from .api import Decoder, GOKU_basic, Pendulum, diffeq_layer
    device = device or "cuda"
    diffeq = diffeq or Pendulum(abstol=1e-8, reltol=1e-8)
    rng = np.random.default_rng(seed)
    ps = rng.uniform(p0_range[0], p0_range[1], (1, n_traj)).astype(np.float32)
    u0s = np.stack([rng.uniform(lo, hi, n_traj) for lo, hi in u0_range]).astype(np.float32)
    T = int(round((tspan[1] - tspan[0]) / dt)) + 1
    ts = tspan[0] + dt * np.arange(T)
    dec = Decoder(GOKU_basic(), (None, diffeq, None))
    with torch.no_grad():
        z = diffeq_layer(dec, (torch.from_numpy(u0s).to(device), torch.from_numpy(ps).to(device)), ts)      # [2, n, T]
    latent = z.permute(0, 2, 1).contiguous()                                                                    # [2, T, n]
    frames = create_frames(latent[0].t(), *high_dim_args)                                                        # [n, T, 28, 28]
    return latent, torch.from_numpy(u0s).to(device), torch.from_numpy(ps).to(device), frames.permute(2, 3, 1, 0).contiguous()


# ---- on-disk container and the example script's loader (host code, no kernels) ------------------------------------------------
DATA_KEYS = ("latent_data", "u0s", "ps", "high_dim_data")


def save_dataset(path: str, latent_data, u0s, ps, high_dim_data) -> str:
    """`@save data_path data` with `data = (latent_data, u0s, ps, high_dim_data)`  [REF examples/pendulum_friction-less/
    model_train.jl:86-91]. A path ending in `.bson` writes the reference's container — a BSON document {data: <the 4-tuple in the
    reference's nesting, lowered as BSON.jl lowers it>} (`bson.py`; unpinned against BSON.jl's own bytes: none exist here) — any other
    path one `.npz` with the tuple's names. Dense arrays in the reference's shapes either way: latent_data [2, T, n], u0s [2, n],
    ps [1, n], high_dim_data [28, 28, T, n] (float32)."""
    arrs = {k: np.ascontiguousarray((v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)), dtype=np.float32)
            for k, v in zip(DATA_KEYS, (latent_data, u0s, ps, high_dim_data))}
    if path.endswith(".bson"):
        from . import bson
        return bson.dump_dataset(path, *(arrs[k] for k in DATA_KEYS))
    if not path.endswith(".npz"):
        path += ".npz"
    np.savez(path, **arrs)
    return path


def load_dataset(path: str):
    """`load(data_path, :data)` → (latent_data, u0s, ps, high_dim_data) as numpy arrays  [REF model_train.jl:95]."""
    if path.endswith(".bson"):
        from . import bson
        return bson.load_dataset(path)
    with np.load(path) as f:
        missing = [k for k in DATA_KEYS if k not in f]
        if missing:
            raise KeyError(f"{path}: not a dataset container (missing {missing})")
        return tuple(np.array(f[k]) for k in DATA_KEYS)


def save_weights(path: str, weights) -> str:
    """`@save "…/best_model_weights.bson" weights` with `weights = Flux.params(model)` [REF model_train.jl:213-215]: the list of
    parameter arrays (here: each module's flat `Flux.destructure` vector), as a BSON vector of Float32 arrays."""
    from . import bson
    return bson.save(path, weights=[np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w),
                                                         dtype=np.float32) for w in weights])


def load_weights(path: str):
    from . import bson
    return bson.load(path)["weights"]


def prepare_training_data(data, at: float = 0.9):
    """What the example script does with the loaded tuple  [REF model_train.jl:96-121]: vectorise the frames to
    [input_dim, full_seq_len, observations], split the OBSERVATIONS 90 / 10 in order (`splitobs(·, 0.9)`: the first
    90 % (rounded to nearest, as MLUtils' splitobs does) train, the rest validation — no shuffle), and bring the validation set to [input_dim, n_val, T].
    Returns dict(train_set [input_dim, T, n_train], val_set [input_dim, n_val, T], train/val latent and params, input_dim,
    full_seq_len)."""
    latent, _u0s, ps, high = data
    h, w, T, n = high.shape
    train_data = high.reshape(h * w, T, n, order="F").astype(np.float32)           # Julia's reshape(train_data, :, T, n) is column-major
    n_train = int(round(at * n))                                                   # MLUtils.splitobs: round(Int, at·n)
    sp = lambda a: (a[..., :n_train], a[..., n_train:])
    tr, va = sp(train_data)
    trl, val = sp(np.asarray(latent))
    trp, vap = sp(np.asarray(ps))
    return dict(train_set=tr, val_set=np.transpose(va, (0, 2, 1)), train_set_latent=trl, val_set_latent=val,
                train_set_params=trp, val_set_params=vap, input_dim=h * w, full_seq_len=T)


def data_loader(train_set, batch_size: int, rng: Optional[np.random.Generator] = None, device: Optional[str] = None):
    """`DataLoader(train_set, batchsize, shuffle=true, partial=false)`  [REF model_train.jl:117]: one pass over a random
    permutation of the observations in full minibatches [input_dim, full_seq_len, batch] → yielded as [input_dim, batch,
    full_seq_len] tensors (the script's `permutedims(x, [1,3,2])` inside the loop [REF model_train.jl:179])."""
    rng = rng or np.random.default_rng()
    n = train_set.shape[-1]
    perm = rng.permutation(n)
    for i in range(0, n - batch_size + 1, batch_size):
        xb = np.transpose(train_set[..., perm[i:i + batch_size]], (0, 2, 1))
        t = torch.from_numpy(np.ascontiguousarray(xb))
        yield t.to(device) if device else t

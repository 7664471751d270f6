"""latentdiffeq.jl_amd — MI355X-native latent-ODE forward solve + adjoint behind LatentDiffEq.jl's
`diffeq_layer` API (and nothing else of the reference). Import as `latentdiffeq_amd`.

Layout:
  csrc/      HIP kernels (gfx950) + the C ABI of include/lde.h  → liblde.so
  _lib.py    ctypes binding of liblde.so (no fallback: raises if the library is missing)
  api.py     host-side mirror of the reference interface: Pendulum / Pendulum_friction / NODE,
             GOKU_basic / LatentODE, Decoder, diffeq_layer, transform_after_diffeq
  chain.py   Dense / SkipConnection / Chain, apply_latent_out, apply_reconstructor (the dense chains either side of the solve)
  recurrent.py  RNN / LSTM / Recurrent, Encoder, apply_feature_extractor / _pattern_extractor / _latent_in, sample
  loss.py    sample, vector_kl, reconstruction_loss (lde_sample_* / lde_kl_* / lde_mse_*)
  train.py   LatentDiffEqModel, default_layers, loss_batch, frange_cycle_linear, time_loader, train (host only)
  data.py    generate_dataset, create_frames: synthetic pendulum videos (approximation of the Luxor drawing)
  dist.py    one-process-per-GPU batch sharding + the single gradient all-reduce (RCCL / gloo)
"""
from .build import build_lib  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch or the built library
    import importlib

    if name.startswith("_"):   # `from . import _lib` inside a submodule: a plain submodule import, not an API name
        try:
            return importlib.import_module(f"{__name__}.{name}")
        except ModuleNotFoundError:
            raise AttributeError(name) from None
    for mod in ("api", "dist", "chain", "recurrent", "loss", "train", "data"):
        try:
            m = importlib.import_module(f"{__name__}.{mod}")
        except ModuleNotFoundError:
            continue
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)

"""Synthetic inputs of the hot path: shared by bench.py, the tests and golden generation.
Pure numpy; array *memory* is the reference's column-major layout: z0 (B,D) == [D×B], ẑ (T,B,D') == [D'×B×T]."""
import numpy as np

def pendulum_inputs(B: int, seed: int = 1, dtype=np.float32):
    """θ₀~U(±π/6), ω₀~U(±π/3), L~U(1,2)  [REF examples/pendulum_friction-less/create_data.jl:19-22]."""
    rng = np.random.default_rng(seed)
    z0 = np.stack([rng.uniform(-np.pi / 6, np.pi / 6, B), rng.uniform(-np.pi / 3, np.pi / 3, B)], axis=1)
    L = rng.uniform(1.0, 2.0, (B, 1))
    return z0.astype(dtype), L.astype(dtype)


def time_grid(T: int = 50, dt: float = 0.05, t0: float = 0.0):
    """t = range(0, step=0.05, length=50), Float64  [REF examples/pendulum_friction-less/model_train.jl:40,44,181]."""
    return t0 + dt * np.arange(T, dtype=np.float64)


def cotangent(T: int, B: int, Dp: int, seed: int = 2, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((T, B, Dp)) / (B * T)).astype(dtype)


def mlp_weights(layers, seed: int = 3, scale: float = 1.0, dtype=np.float32):
    """Flat Flux.destructure-order weights, U(±1/√fan_in)·scale, biases U(±1/√fan_in)·scale."""
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(len(layers) - 1):
        fan_in, out = layers[i], layers[i + 1]
        bound = scale / np.sqrt(fan_in)
        Wl = rng.uniform(-bound, bound, (out, fan_in))
        bl = rng.uniform(-bound, bound, out)
        parts.append(Wl.flatten(order="F"))  # vec(W), column-major [out×in]
        parts.append(bl)
    return np.concatenate(parts).astype(dtype)

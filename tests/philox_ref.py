"""Restatement (numpy) of what lde_randn computes (include/lde.h): Philox4x32-10 [Salmon et al., SC'11] and the Box–Muller map of its
words. Test infrastructure only. Pinned by the generator's published known-answer vectors (tests/test_philox.py)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays (or scalars) of counter words and key words → four arrays of output words (uint32 values held in uint64)."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3)]
    k = [np.asarray(v, dtype=np.uint64) & MASK for v in (k0, k1)]
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + np.uint64(W0)) & MASK, (k[1] + np.uint64(W1)) & MASK]
    return c


def words(n, seed, offset, call, epoch=0):
    """The n raw words lde_randn(…, raw_words) writes: block i = philox(counter (i, call, offset + epoch), key seed)."""
    nb = (n + 3) // 4
    off = (int(offset) + int(epoch)) & 0xFFFFFFFFFFFFFFFF
    w = philox4x32_10(np.arange(nb, dtype=np.uint64), call, off & 0xFFFFFFFF, off >> 32, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(w, axis=1).reshape(-1)[:n].astype(np.uint32)


def normals(w):
    """Box–Muller of consecutive word pairs (f64): u = ((w >> 8) + ½)/2²⁴; (√(−2 ln u₁) cos 2πu₂, √(−2 ln u₁) sin 2πu₂)."""
    w = np.asarray(w, dtype=np.uint64)
    pad = (-len(w)) % 2
    w2 = np.concatenate([w, np.zeros(pad, np.uint64)]).reshape(-1, 2)
    u = ((w2 >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0
    r = np.sqrt(-2.0 * np.log(u[:, 0]))
    z = np.stack([r * np.cos(2 * np.pi * u[:, 1]), r * np.sin(2 * np.pi * u[:, 1])], axis=1).reshape(-1)
    return z[:len(w)]

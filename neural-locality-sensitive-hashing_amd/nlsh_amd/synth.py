"""Seeded synthetic workloads: no datasets or checkpoints exist offline.

Everything here is numpy (`default_rng`, PCG64: stable across numpy versions) so the
golden-vector generator, the oracle, the tests and bench.py all see identical bytes.
Shapes follow BASELINE.json `configs`: SIFT-like (128-d, clustered non-negative integers
stored as fp32), GloVe-like (100-d, cosine), Deep-like (96-d, unit norm).
"""
import numpy as np

SEED_DATA = 1234
SEED_QUERY = 4321
SEED_WEIGHTS = 0
SEED_CENTRES = 99


def _centres(n_clusters, d, lo, hi, seed=SEED_CENTRES):
    return np.random.default_rng(seed).uniform(lo, hi, size=(n_clusters, d)).astype(np.float32)


def sift_like(n, d=128, seed=SEED_DATA, n_clusters=1000, sigma=24.0, chunk=1 << 16):
    """Clustered integer-valued fp32 rows in [0, 218] (SIFT descriptors are uint8-ish)."""
    cen = _centres(n_clusters, d, 0.0, 128.0)
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        which = rng.integers(0, n_clusters, size=e - s)
        x = cen[which] + rng.standard_normal((e - s, d), dtype=np.float32) * np.float32(sigma)
        out[s:e] = np.clip(np.rint(x), 0, 218)
    return out


def sift_manifold(n, d=128, seed=SEED_DATA, latent_dim=6, n_clusters=32, spread=0.5, chunk=1 << 16):
    """SIFT-like integers in [0, 218] with LOW INTRINSIC DIMENSION (a random smooth map of a
    `latent_dim`-d Gaussian mixture), so that nearest neighbours are meaningful.

    `sift_like` (isotropic 128-d noise around cluster centres) has concentrated distances: the 10-NN
    of a point are arbitrary members of its 1000-point cluster, so recall@10 of ANY space partition is
    ~ candidates/1000.  Real SIFT descriptors have intrinsic dimension ~10; this generator is what the
    learned-hash bench workload uses, `sift_like` stays for the golden fixtures.
    """
    g = np.random.default_rng(SEED_CENTRES + 7)
    A1 = g.standard_normal((latent_dim, 64)).astype(np.float32)
    b1 = g.uniform(-1, 1, size=64).astype(np.float32)
    A2 = (g.standard_normal((64, d)) / 8.0).astype(np.float32)
    cen = g.standard_normal((n_clusters, latent_dim)).astype(np.float32)

    def fmap(z):
        return np.maximum(z @ A1 + b1, 0) @ A2

    cal = fmap(cen[g.integers(0, n_clusters, 8192)] + spread * g.standard_normal((8192, latent_dim)).astype(np.float32))
    mu, sd = np.float32(cal.mean()), np.float32(cal.std())
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        z = cen[rng.integers(0, n_clusters, size=e - s)] + np.float32(spread) * rng.standard_normal((e - s, latent_dim), dtype=np.float32)
        out[s:e] = np.clip(np.rint(64.0 + 32.0 * (fmap(z) - mu) / sd), 0, 218)
    return out


def glove_manifold(n, d=100, seed=SEED_DATA, latent_dim=8, n_clusters=64, spread=0.6, chunk=1 << 16):
    """Real-valued embedding-like rows with low intrinsic dimension (cosine workloads): a random smooth
    map of a `latent_dim`-d Gaussian mixture, centred, plus 5 % isotropic noise.  Same motivation as
    `sift_manifold`: `glove_like` is isotropic, so its cosine k-NN are arbitrary."""
    g = np.random.default_rng(SEED_CENTRES + 11)
    A1 = g.standard_normal((latent_dim, 96)).astype(np.float32)
    b1 = g.uniform(-1, 1, size=96).astype(np.float32)
    A2 = (g.standard_normal((96, d)) / 9.8).astype(np.float32)
    cen = g.standard_normal((n_clusters, latent_dim)).astype(np.float32)

    def fmap(z):
        return np.tanh(z @ A1 + b1) @ A2

    cal = fmap(cen[g.integers(0, n_clusters, 8192)] + spread * g.standard_normal((8192, latent_dim)).astype(np.float32))
    mu = cal.mean(axis=0).astype(np.float32)
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        z = cen[rng.integers(0, n_clusters, size=e - s)] + np.float32(spread) * rng.standard_normal((e - s, latent_dim), dtype=np.float32)
        out[s:e] = fmap(z) - mu + np.float32(0.05) * rng.standard_normal((e - s, d), dtype=np.float32)
    return out


def glove_like(n, d=100, seed=SEED_DATA, chunk=1 << 16):
    """N(0,1) rows with a fixed per-dimension scale in U(0.3, 1.0) (cosine workloads)."""
    scale = np.random.default_rng(SEED_CENTRES + 1).uniform(0.3, 1.0, size=d).astype(np.float32)
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = rng.standard_normal((e - s, d), dtype=np.float32) * scale
    return out


def deep_like(n, d=96, seed=SEED_DATA, n_clusters=1000, sigma=0.35, chunk=1 << 16):
    """Gaussian mixture projected to the unit sphere (Deep1B descriptors are L2-normalised)."""
    cen = _centres(n_clusters, d, -1.0, 1.0, seed=SEED_CENTRES + 2)
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        which = rng.integers(0, n_clusters, size=e - s)
        x = cen[which] + rng.standard_normal((e - s, d), dtype=np.float32) * np.float32(sigma)
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        out[s:e] = x
    return out


def standardise(x, mean=None, std=None):
    """Per-dimension standardisation, what `SIFT(unit_norm=True)` does (reference nlsh/data.py:125-129)."""
    if mean is None:
        mean = x.mean(axis=0, dtype=np.float64).astype(np.float32)
        std = x.std(axis=0, dtype=np.float64).astype(np.float32)
    std = np.where(std == 0, np.float32(1), std).astype(np.float32)
    return ((x - mean) / std).astype(np.float32), mean, std


def make_weights(dims, seed=SEED_WEIGHTS, bias=True, gain=1.0):
    """`nn.Linear`-default-shaped init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)), from numpy.

    dims = [d, h1, ..., H].  Returns ([W_l of shape (dims[l+1], dims[l])], [b_l or None]).
    """
    rng = np.random.default_rng(seed)
    Ws, bs = [], []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = gain / np.sqrt(fan_in)
        Ws.append(rng.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32))
        bs.append(rng.uniform(-bound, bound, size=(fan_out,)).astype(np.float32) if bias else None)
    return Ws, bs


def brute_force_topk_np(queries, corpus, k, metric="l2", chunk=256):
    """Exact top-k ground truth in float64 (small cases; tests only)."""
    q64 = queries.astype(np.float64)
    c64 = corpus.astype(np.float64)
    out = np.empty((len(queries), k), dtype=np.int64)
    if metric == "cosine":
        c64 = c64 / np.maximum(np.linalg.norm(c64, axis=1, keepdims=True), 1e-12)
    for s in range(0, len(queries), chunk):
        qq = q64[s:s + chunk]
        if metric == "l2":
            dist = (qq * qq).sum(1)[:, None] - 2.0 * qq @ c64.T + (c64 * c64).sum(1)[None, :]
        else:
            qq = qq / np.maximum(np.linalg.norm(qq, axis=1, keepdims=True), 1e-12)
            dist = 1.0 - qq @ c64.T
        # stable order: (distance, row id)
        idx = np.lexsort((np.broadcast_to(np.arange(dist.shape[1]), dist.shape), dist), axis=1)[:, :k]
        out[s:s + chunk] = idx
    return out

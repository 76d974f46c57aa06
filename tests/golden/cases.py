"""Seeded input construction shared by make_golden.py (reference side) and the tests (our side).

Only inputs are built here (numpy, `nlsh_amd.synth`): nothing from the reference is imported.
"""
import numpy as np

from nlsh_amd import synth

G2_CASES = [
    # name, d, hidden, H, tanh, two_layer, data
    ("sift_256_256_16", 128, (256, 256), 16, False, False, "sift_std"),
    ("glove_256_256_24_tanh", 100, (256, 256), 24, True, False, "glove"),
    ("deep_256_256_32", 96, (256, 256), 32, False, False, "deep"),
    ("sift_64_64_12", 128, (64, 64), 12, False, False, "sift_std"),
    ("two_layer_8", 128, (256, 256), 8, False, True, "sift_std"),
    ("glove25_96_8", 25, (96,), 8, False, False, "glove"),
    # bias-free encoders (reference encoders.py:10,31 `with_bias=False`; the output layer keeps its bias, hashings.py:19)
    ("two_layer_16_nobias", 128, (256, 256), 16, False, True, "sift_std"),
    ("glove_64_64_12_nobias", 100, (64, 64), 12, False, False, "glove"),
]


def g2_with_bias(name):
    return not name.endswith("_nobias")


def g2_weights(case):
    """Seeded weights of a G2 case; encoder biases are None for the bias-free cases."""
    name, d, hidden, H = case[0], case[1], case[2], case[3]
    i = [c[0] for c in G2_CASES].index(name)
    Ws, bs = synth.make_weights([d] + list(hidden) + [H], seed=100 + i)
    if not g2_with_bias(name):
        bs = [None] * (len(bs) - 1) + [bs[-1]]
    return Ws, bs


def g2_inputs(kind, d, n=96):
    if kind == "sift_std":
        x, _, _ = synth.standardise(synth.sift_like(n, d, seed=11))
    elif kind == "glove":
        x = synth.glove_like(n, d, seed=12)
    else:
        x = synth.deep_like(n, d, seed=13)
    return x


def g5_data(metric, d, N, Q, seed):
    if metric == "l2":
        corpus = synth.sift_like(N, d, seed=seed, n_clusters=20)
        queries = synth.sift_like(Q, d, seed=seed + 1, n_clusters=20)
    else:
        corpus = synth.glove_like(N, d, seed=seed)
        queries = synth.glove_like(Q, d, seed=seed + 1)
    # engineered exact ties: duplicate rows (same vector -> same bucket -> same distance)
    corpus[N // 2: N // 2 + 40] = corpus[:40]
    return corpus, queries


def g5_inputs(meta):
    corpus, queries = g5_data(meta["metric"], meta["d"], meta["N"], meta["Q"], meta["seed"])
    Ws, bs = synth.make_weights([meta["d"], 64, 64, meta["H"]], seed=meta["seed"] + 2)
    return corpus, queries, Ws, bs


G7 = dict(N=10000, Q=100, d=128, H=8, k=10, hidden=(256, 256))


def g7_inputs():
    corpus, mean, std = synth.standardise(synth.sift_like(G7["N"], G7["d"], seed=synth.SEED_DATA))
    queries, _, _ = synth.standardise(synth.sift_like(G7["Q"], G7["d"], seed=synth.SEED_QUERY), mean, std)
    Ws, bs = synth.make_weights([G7["d"]] + list(G7["hidden"]) + [G7["H"]], seed=synth.SEED_WEIGHTS)
    return corpus, queries, Ws, bs


def assert_topk_equivalent(got_ids, ref_ids, cand_rows, cand_dist, k, tol):
    """Top-k id lists agree modulo ties/near-ties (torch.topk tie order is unspecified, F11).

    Both lists must have length k, contain only candidates, and their distance profiles must
    match within `tol` (relative to max(1,|d|)); ids may differ only among candidates whose
    distance is within `tol` of the k-th distance.
    """
    assert len(got_ids) == k and len(ref_ids) == k
    dist_of = {}
    for r, dd in zip(cand_rows.tolist(), cand_dist.tolist()):
        dist_of[int(r)] = float(dd)
    assert all(i in dist_of for i in got_ids), "returned a non-candidate"
    assert len(set(got_ids)) == k, "duplicate ids"
    dg = np.array([dist_of[i] for i in got_ids])
    dr = np.array([dist_of[i] for i in ref_ids])
    slack = tol * np.maximum(1.0, np.abs(dr))
    assert np.all(np.abs(np.sort(dg) - np.sort(dr)) <= slack), "distance profile differs"
    kth = np.sort(dr)[-1]
    for i in set(got_ids) ^ set(ref_ids):
        assert abs(dist_of[i] - kth) <= tol * max(1.0, abs(kth)), "id differs away from the k-th tie"
    # ascending order (within tolerance)
    assert np.all(np.diff(dg) >= -slack[:-1])

"""Shared helpers for the GPU parity tests (product objects built from seeded numpy weights)."""
import importlib.util
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
_spec = importlib.util.spec_from_file_location("golden_cases", os.path.join(G, "cases.py"))
cases = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cases)


def make_hashing(d, hidden, H, Ws, bs, tanh=False, two_layer=False, compat=True, seed=0):
    from nlsh_amd.encoders import MultiLayerRelu, TwoLayer256Relu
    from nlsh_amd.hashings import MultivariateBernoulli
    enc = TwoLayer256Relu(d) if two_layer else MultiLayerRelu(d, list(hidden))
    hashing = MultivariateBernoulli(enc, H, None, tanh_output=tanh, compat=compat, seed=seed)
    lin = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    assert len(lin) == len(Ws)
    with torch.no_grad():
        for m, W, b in zip(lin, Ws, bs):
            m.weight.copy_(torch.from_numpy(W))
            if b is not None:
                m.bias.copy_(torch.from_numpy(b))
    hashing.train_mode(False)
    return hashing


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def check_topk_against_candidates(idx_row, dist_row, cand_rows, cand_d64, k, rtol=2e-5):
    """One query: returned (ids, distances) vs exact fp64 distances of ITS candidate set.

    - ids are distinct candidates, distances ascending;
    - each returned distance matches the fp64 distance of that id;
    - nothing outside the result beats the k-th result by more than the tolerance (ties / fp32
      near-ties at the boundary may resolve either way: SURVEY F11).
    """
    n = min(k, len(cand_rows))
    ids = [int(i) for i in idx_row[:n]]
    assert all(int(i) == -1 for i in idx_row[n:]), "padding must be -1"
    assert all(np.isinf(v) for v in dist_row[n:]), "padding must be +inf"
    d_of = {int(r): float(v) for r, v in zip(cand_rows, cand_d64)}
    assert len(set(ids)) == n and all(i in d_of for i in ids)
    got = np.array([d_of[i] for i in ids])
    tol = rtol * np.maximum(1.0, np.abs(got))
    assert np.all(np.abs(np.asarray(dist_row[:n], dtype=np.float64) - got) <= tol), "distance value off"
    assert np.all(np.diff(np.asarray(dist_row[:n])) >= 0), "distances not ascending"
    if n:
        kth = got.max()
        rest = np.array([v for r, v in d_of.items() if r not in set(ids)])
        if len(rest):
            assert rest.min() >= kth - rtol * max(1.0, abs(kth)), "a closer candidate was missed"

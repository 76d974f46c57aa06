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
    with_bias = bs[0] is not None          # encoders.py:10,31: bias-free encoder layers (the output layer keeps its bias)
    enc = TwoLayer256Relu(d, with_bias=with_bias) if two_layer else MultiLayerRelu(d, list(hidden), with_bias=with_bias)
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


def check_scan_properties(ix, qg, cg, keys, nkeys, dist, idx, nc, k, metric, rtol=2e-5):
    """Size-independent properties of one scan result on a full-size index (the oracle cannot brute-force these
    sizes): candidate counts vs an independent torch recomputation, membership of every returned id in one of the
    query's probed buckets, ascending order, no duplicates, distances vs stock torch ops (nlsh/data.py:99-109,191-201)."""
    d = cg.shape[1]
    pos = torch.searchsorted(ix.uniq_keys, keys.clamp(min=int(ix.uniq_keys.min()), max=int(ix.uniq_keys.max())))
    pos = pos.clamp(max=ix.n_buckets - 1)
    hit = ix.uniq_keys[pos] == keys
    sizes = (ix.offsets[1:] - ix.offsets[:-1])[pos] * hit
    valid = torch.arange(keys.shape[1], device=keys.device)[None, :] < nkeys[:, None]
    assert torch.equal((sizes * valid).sum(1).int(), nc), "n_candidates != sum of probed bucket sizes"
    ok = idx >= 0
    assert torch.equal(ok.sum(1).int(), nc.clamp(max=k)), "a query returned fewer ids than min(k, C_q)"
    ck = ix.corpus_keys[idx.clamp(min=0).long()]
    member = ((ck[:, :, None] == keys[:, None, :]) & valid[:, None, :]).any(-1)
    assert bool((member | ~ok).all()), "a returned id is not in a probed bucket"
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()), "distances not ascending"
    srt = torch.sort(idx, dim=1).values
    assert bool(((srt[:, 1:] != srt[:, :-1]) | (srt[:, 1:] < 0)).all()), "duplicate ids"
    qq = qg[:, None, :].expand(-1, k, -1).reshape(-1, d)
    cc = cg[idx.clamp(min=0).long().reshape(-1)]
    if metric == "l2":
        ref = torch.nn.functional.pairwise_distance(qq, cc).reshape(-1, k)
    else:
        ref = (1 - torch.nn.functional.cosine_similarity(qq, cc, dim=-1)).reshape(-1, k)
    assert bool(((dist - ref).abs() <= rtol * ref.abs().clamp(min=1.0))[ok].all()), "distance value off"


def assert_lists_differ_only_at_ties(ids_a, ids_b, query, corpus, metric, rtol=2e-5):
    """Two id lists for one query (ours / the oracle's or the reference's): either identical, or of the same length
    with the same fp64 distance profile -- i.e. they differ only in WHICH of several (near-)equidistant candidates
    they name (torch.topk's tie order is unspecified; fp32 summation order moves near-ties)."""
    ids_a, ids_b = [int(i) for i in ids_a], [int(i) for i in ids_b]
    if ids_a == ids_b:
        return True
    assert len(ids_a) == len(ids_b), (ids_a, ids_b)
    q = np.asarray(query, dtype=np.float64)

    def d64(ids):
        c = np.asarray(corpus[ids], dtype=np.float64)
        if metric == "l2":
            return np.sqrt((((q - c) + 1e-6) ** 2).sum(1))
        return 1.0 - (c @ q) / np.maximum(np.linalg.norm(c, axis=1) * np.linalg.norm(q), 1e-8)
    da, db = d64(ids_a), d64(ids_b)
    assert np.all(np.abs(da - db) <= rtol * np.maximum(1.0, np.abs(db))), "id lists differ beyond a distance tie"
    return False


def l2_forms_differ_only_at_ties(qg, cg, d_exact, i_exact, d_folded, i_folded, rtol=1e-4):
    """Full-size comparison of the two L2 forms on the device (BASELINE.json: "L2 within 1e-4"): returns the number of queries whose id
    lists differ at all, after asserting that (a) every returned distance of the folded form is within rtol * max(1, d) of the exact
    form's at the same rank, (b) where the lists differ, the fp64 distances (nlsh/data.py:201's formula) of the two lists agree rank
    by rank to the same tolerance -- i.e. the lists name different members of a tie at that tolerance, nothing else -- and (c) the
    candidate sets were the same (same padding)."""
    assert torch.equal(i_exact < 0, i_folded < 0)
    ok = i_exact >= 0
    assert bool(((d_exact - d_folded).abs() <= rtol * d_exact.abs().clamp(min=1.0))[ok].all()), "folded distance outside the tolerance"
    diff = (i_exact != i_folded).any(1)
    n_diff = int(diff.sum().item())
    if n_diff:
        rows = torch.nonzero(diff)[:, 0]
        q64 = qg[rows].double()[:, None, :]

        def d64(ids):
            c = cg[ids[rows].clamp(min=0).long()].double()
            return (((q64 - c) + 1e-6) ** 2).sum(-1).sqrt()
        da, db = d64(i_exact), d64(i_folded)
        m = ok[rows]
        assert bool(((da - db).abs() <= rtol * da.clamp(min=1.0))[m].all()), "id lists differ beyond a distance tie at the stated tolerance"
    return n_diff

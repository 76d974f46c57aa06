#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference hot path on CPU.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  Only the small `.npz` / `.json` outputs next to this file are committed; they are
data (inputs + expected outputs), never reference source.

Shims: stub `siren` / `h5py` / `dotenv` modules (absent here, never
called on the path), `.cuda()` -> identity (no GPU here), pyximport build dir in /tmp.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd"))
sys.path.insert(0, "/root/reference")

import pyximport  # noqa: E402

pyximport.install(setup_args={"include_dirs": np.get_include()}, build_dir="/tmp/nlsh_oracle_pyxbld")
for _name, _attrs in (("siren", {"SIREN": object}), ("h5py", {"File": None}),
                      ("hnswlib", {"Index": object}),      # nlsh/trainers/__init__.py:10 pulls it; never called here
                      ("dotenv", {"load_dotenv": lambda *a, **k: None})):
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        _m.__dict__.update(_attrs)
        sys.modules[_name] = _m
os.environ.setdefault("NLSH_MODEL_SAVE_DIR", "/tmp")
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

from encoders import MultiLayerRelu, TwoLayer256Relu  # noqa: E402  (reference encoders.py)
from nlsh.hashings import MultivariateBernoulli  # noqa: E402
from nlsh.indexer import Indexer, build_index  # noqa: E402
from nlsh.data import SIFT, Glove  # noqa: E402
from nlsh.metrics import calculate_recall  # noqa: E402
from nlsh.utils import hash_codes  # noqa: E402
import eval as ref_eval  # noqa: E402  (only for _binarr_to_int, eval.py:49-53)

from nlsh_amd import synth  # noqa: E402  (our seeded generators: shared with the tests)
sys.path.insert(0, HERE)
import cases  # noqa: E402  (tests/golden/cases.py: seeded inputs shared with the tests)

torch.set_num_threads(1)


def _sets_to_lists(sets):
    """Key sets in the reference's own iteration order (`list(qi)`, nlsh/indexer.py:67)."""
    return [[int(k) for k in list(s)] for s in sets]


def set_weights(hashing, Ws, bs):
    """Load numpy weights into the reference modules (Linear layers in forward order)."""
    linears = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    assert len(linears) == len(Ws)
    with torch.no_grad():
        for lin, W, b in zip(linears, Ws, bs):
            assert tuple(lin.weight.shape) == W.shape
            lin.weight.copy_(torch.from_numpy(W))
            if b is not None:
                lin.bias.copy_(torch.from_numpy(b))


def make_hashing(d, hidden, H, seed, tanh=False, two_layer=False, with_bias=True):
    enc = TwoLayer256Relu(d, with_bias=with_bias) if two_layer else MultiLayerRelu(d, list(hidden), with_bias=with_bias)
    hashing = MultivariateBernoulli(enc, H, None, tanh_output=tanh)
    Ws, bs = synth.make_weights([d] + list(hidden) + [H], seed=seed)
    if not with_bias:       # encoders.py:10,31: the encoder's Linear layers have no bias; the output layer keeps its own
        bs = [None] * (len(bs) - 1) + [bs[-1]]
    set_weights(hashing, Ws, bs)
    hashing.train_mode(False)
    return hashing


# ---------------------------------------------------------------- G1: hash_codes / bit packing
def g1():
    out = {"hand": [], "random": []}
    hand = [
        ([[[0, 1, 1], [1, 0, 1]], [[1, 1, 1], [1, 1, 1]]]),
        np.ones((1, 1, 16), dtype=int).tolist(),
        (np.eye(16, dtype=int)[:1][None]).tolist(),                     # MSB only of 16
        (np.eye(24, dtype=int)[:1][None]).tolist(),                     # MSB only of 24 -> wraps to 0
        (np.eye(32, dtype=int)[-1:][None]).tolist(),                    # LSB only of 32
        np.zeros((2, 3, 8), dtype=int).tolist(),
    ]
    for codes in hand:
        arr = np.asarray(codes, dtype=np.int32)
        ref = hash_codes(arr)
        full = [[int(ref_eval._binarr_to_int([int(b) for b in row])) for row in rows] for rows in arr]
        out["hand"].append({"codes": arr.tolist(), "ref_int16_sets": [sorted(int(k) for k in s) for s in ref],
                            "full_keys": full})
    rng = np.random.default_rng(7)
    for H in (8, 12, 16, 24, 32):
        arr = rng.integers(0, 2, size=(16, 5, H)).astype(np.int32)
        ref = hash_codes(arr)
        full = [[int(ref_eval._binarr_to_int([int(b) for b in row])) for row in rows] for rows in arr]
        out["random"].append({"H": H, "codes": arr.tolist(),
                              "ref_int16_sets": [sorted(int(k) for k in s) for s in ref],
                              "ref_int16_iter": _sets_to_lists(ref),
                              "full_keys": full})
    empty = hash_codes(np.zeros((0, 1, 8), dtype=np.int32))
    out["empty_len"] = len(empty)
    with open(os.path.join(HERE, "g1_hash_codes.json"), "w") as f:
        json.dump(out, f)


# ---------------------------------------------------------------- G2: hasher forward
G2_CASES = cases.G2_CASES
g2_inputs = cases.g2_inputs


def g2():
    arrays = {}
    for i, (name, d, hidden, H, tanh, two_layer, kind) in enumerate(G2_CASES):
        hashing = make_hashing(d, hidden, H, seed=100 + i, tanh=tanh, two_layer=two_layer, with_bias=cases.g2_with_bias(name))
        if not cases.g2_with_bias(name):
            assert all(m.bias is None for m in hashing._encoder.modules() if isinstance(m, torch.nn.Linear))
        x = g2_inputs(kind, d)
        xt = torch.from_numpy(x)
        with torch.no_grad():
            probs = hashing.predict(xt)                                  # nlsh/hashings.py:39-40
            z = hashing._hasher.output_layer(hashing._hasher._encoder(xt))
            keys = hashing.hash(xt, 1)                                   # hard hash, hashings.py:66-92
        p = probs.numpy()
        pb = p / 2.0 + 0.5 if tanh else p
        bits = (pb > 0.5).astype(np.int32)
        full = [int(ref_eval._binarr_to_int([int(b) for b in row])) for row in bits]
        arrays[name + "/probs"] = p.astype(np.float32)
        arrays[name + "/z"] = z.numpy().astype(np.float32)
        arrays[name + "/bits"] = bits.astype(np.uint8)
        arrays[name + "/key_ref_int16"] = np.array([list(s)[0] for s in keys], dtype=np.int32)
        arrays[name + "/key_full"] = np.array(full, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "g2_hasher.npz"), **arrays)


# ---------------------------------------------------------------- G3: Indexer.hash batching rule (F6)
def g3():
    out = []
    hashing = make_hashing(128, (64, 64), 12, seed=300)
    corpus, _, _ = synth.standardise(synth.sift_like(64, 128, seed=31))
    indexer = Indexer(hashing, torch.from_numpy(corpus), SIFT.distance)
    for Q, bs, ht in ((10, 4, 5), (8, 4, 5), (3, 4, 5), (9, 3, 4)):
        x, _, _ = synth.standardise(synth.sift_like(Q, 128, seed=32))
        torch.manual_seed(0)
        with torch.no_grad():
            sets = indexer.hash(torch.from_numpy(x), batch_size=bs, hash_times=ht)
            hard = indexer.hash(torch.from_numpy(x), batch_size=bs, hash_times=1)
        out.append({"Q": Q, "batch_size": bs, "hash_times": ht,
                    "sizes": [len(s) for s in sets],
                    "hard_keys": [int(list(s)[0]) for s in hard],
                    "hard_in_set": [int(list(h)[0]) in s for h, s in zip(hard, sets)]})
    with open(os.path.join(HERE, "g3_batching.json"), "w") as f:
        json.dump(out, f)


# ---------------------------------------------------------------- G4: build_index
def g4():
    out = {}
    # the reference's own unit-test vector (nlsh/tests/test_indexer.py:7-19)
    idx = build_index([set([1, 2]), set([2, 3, 4]), set([1, 5])], cuda=False)
    out["ref_test"] = {"input": [[1, 2], [2, 3, 4], [1, 5]],
                       "expected": {str(k): v.tolist() for k, v in idx.items()}}
    rng = np.random.default_rng(41)
    keys = rng.integers(-300, 300, size=5000).astype(np.int32)
    keys[:50] = 32767
    keys[50:60] = -32768
    idx = build_index([{int(k)} for k in keys], cuda=False)
    arrays = {"keys": keys}
    uk = np.array(sorted(idx.keys()), dtype=np.int32)
    arrays["uniq_keys"] = uk
    arrays["rows_concat"] = np.concatenate([idx[int(k)].numpy() for k in uk]).astype(np.int32)
    arrays["sizes"] = np.array([len(idx[int(k)]) for k in uk], dtype=np.int32)
    with open(os.path.join(HERE, "g4_build_index.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "g4_build_index.npz"), **arrays)


# ---------------------------------------------------------------- G5/G6: Indexer.query on injected keys
def g5_case(name, metric, d, N, Q, H, k, seed, arrays, meta):
    corpus, queries = cases.g5_data(metric, d, N, Q, seed)
    dist_fn = SIFT.distance if metric == "l2" else Glove.distance     # nlsh/data.py:191-201 | 99-109
    hashing = make_hashing(d, (64, 64), H, seed=seed + 2, tanh=(metric == "cosine"))
    ct, qt = torch.from_numpy(corpus), torch.from_numpy(queries)
    with torch.no_grad():
        indexer = Indexer(hashing, ct, dist_fn)                          # builds index via real hashing
        corpus_keys = np.array([list(s)[0] for s in indexer.hash(ct, hash_times=1)], dtype=np.int32)
    present = sorted(indexer.index2row.keys())
    rng = np.random.default_rng(seed + 3)
    injected = []
    for q in range(Q):
        r = q % 10
        if r == 0:
            ks = {1000000 + q}                                           # unknown key -> C_q = 0
        elif r == 1:
            # smallest bucket only -> likely C_q < k (F7)
            smallest = min(present, key=lambda kk: len(indexer.index2row[kk]))
            ks = {int(smallest)}
        elif r == 2:
            two = sorted(present, key=lambda kk: len(indexer.index2row[kk]))[:2]
            ks = {int(t) for t in two} | {999999}
        else:
            n_keys = int(rng.integers(1, 7))
            ks = {int(present[i]) for i in rng.integers(0, len(present), size=n_keys)}
        injected.append(ks)
    indexer.hash = lambda query_vectors, batch_size=4096, hash_times=1: injected
    with torch.no_grad():
        ids, ncand = indexer.query(qt, k=k, hash_times=10)               # nlsh/indexer.py:56-96
    # per-query candidate rows (concat order) + reference distances for them
    cand_rows, cand_dist, cand_off = [], [], [0]
    for q in range(Q):
        rows = [indexer.index2row[kk] for kk in list(injected[q]) if kk in indexer.index2row]
        rows = torch.cat(rows) if rows else torch.LongTensor([])
        if len(rows):
            with torch.no_grad():
                dd = dist_fn(qt[q], ct[rows])
        else:
            dd = torch.zeros(0)
        cand_rows.append(rows.numpy().astype(np.int32))
        cand_dist.append(dd.numpy().astype(np.float32))
        cand_off.append(cand_off[-1] + len(rows))
    gt = synth.brute_force_topk_np(queries, corpus, k, metric=metric)
    recalls = calculate_recall(list(gt), ids)                            # nlsh/metrics.py:10-25
    arrays[name + "/corpus_keys"] = corpus_keys
    arrays[name + "/cand_rows"] = np.concatenate(cand_rows) if cand_rows else np.zeros(0, np.int32)
    arrays[name + "/cand_dist"] = np.concatenate(cand_dist) if cand_dist else np.zeros(0, np.float32)
    arrays[name + "/cand_off"] = np.array(cand_off, dtype=np.int64)
    arrays[name + "/ground_truth"] = gt.astype(np.int32)
    arrays[name + "/recalls"] = np.array(recalls, dtype=np.float64)
    arrays[name + "/ncand"] = np.array(ncand, dtype=np.int64)
    meta[name] = {"metric": metric, "d": d, "N": N, "Q": Q, "H": H, "k": k, "seed": seed,
                  "injected_iter": _sets_to_lists(injected), "result_ids": [[int(i) for i in r] for r in ids],
                  "mean_recall": float(np.mean(recalls))}


def g5():
    arrays, meta = {}, {}
    g5_case("l2_small", "l2", 128, 3000, 60, 8, 10, 500, arrays, meta)
    g5_case("cos_small", "cosine", 100, 2500, 50, 8, 10, 600, arrays, meta)
    g5_case("l2_k3", "l2", 128, 1500, 30, 6, 3, 700, arrays, meta)
    np.savez_compressed(os.path.join(HERE, "g5_query.npz"), **arrays)
    with open(os.path.join(HERE, "g5_query.json"), "w") as f:
        json.dump(meta, f)


# ---------------------------------------------------------------- G7: end-to-end (real hashing, hard keys)
def g7():
    """Full Indexer build + query with hash_times=1 (deterministic) on SIFT-small-like shape."""
    N, Q, d, H, k = (cases.G7[x] for x in ("N", "Q", "d", "H", "k"))
    corpus, queries, Ws, bs = cases.g7_inputs()
    hashing = MultivariateBernoulli(MultiLayerRelu(d, list(cases.G7["hidden"])), H, None)
    set_weights(hashing, Ws, bs)
    hashing.train_mode(False)
    ct, qt = torch.from_numpy(corpus), torch.from_numpy(queries)
    with torch.no_grad():
        indexer = Indexer(hashing, ct, SIFT.distance)
        ids, ncand = indexer.query(qt, k=k, hash_times=1)
        qkeys = [int(list(s)[0]) for s in indexer.hash(qt, hash_times=1)]
        z = hashing._hasher.output_layer(hashing._hasher._encoder(qt)).numpy()
    gt = synth.brute_force_topk_np(queries, corpus, k)
    rec = calculate_recall(list(gt), ids, np.mean)
    sizes = {int(kk): int(len(v)) for kk, v in indexer.index2row.items()}
    meta = {"N": N, "Q": Q, "d": d, "H": H, "k": k, "result_ids": [[int(i) for i in r] for r in ids],
            "ncand": [int(c) for c in ncand], "qkeys": qkeys, "bucket_sizes": sizes, "mean_recall": float(rec)}
    with open(os.path.join(HERE, "g7_sift_small.json"), "w") as f:
        json.dump(meta, f)
    np.savez_compressed(os.path.join(HERE, "g7_sift_small.npz"), query_z=z.astype(np.float32),
                        ground_truth=gt.astype(np.int32))


def g8():
    """Training-side formulas the minimal trainer restates: triplet loss (nlsh/trainers/triplet.py:16-26) on the
    Bernoulli-code L2 distance (nlsh/learning/distances.py:245-254), values + gradients on seeded probabilities."""
    from nlsh.trainers.triplet import triplet_loss
    from nlsh.learning.distances import MVBernoulliL2
    rng = np.random.default_rng(808)
    arrays = {}
    for name, (n, H, margin) in {"a": (64, 16, 0.1), "b": (33, 24, 1.0), "c": (8, 8, 0.0)}.items():
        pa, pp, pn = (rng.uniform(0.01, 0.99, size=(n, H)).astype(np.float32) for _ in range(3))
        pp[:3] = pa[:3]                                    # zero positive distance rows
        ta, tp, tn = (torch.from_numpy(x).requires_grad_(True) for x in (pa, pp, pn))
        dist_fn = MVBernoulliL2().rowwise
        loss = triplet_loss(ta, tp, tn, dist_fn, margin=margin)
        loss.backward()
        arrays.update({f"{name}_anchor": pa, f"{name}_pos": pp, f"{name}_neg": pn, f"{name}_margin": np.float32(margin),
                       f"{name}_d_pos": dist_fn(ta, tp).detach().numpy(), f"{name}_loss": loss.detach().numpy(),
                       f"{name}_grad_anchor": ta.grad.numpy(), f"{name}_grad_neg": tn.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "g8_triplet.npz"), **arrays)


if __name__ == "__main__":
    g1(); g2(); g3(); g4(); g5(); g7(); g8()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith((".json", ".npz")):
            print(f"{fn}: {os.path.getsize(os.path.join(HERE, fn))} bytes")

#!/usr/bin/env python3
"""Cross-time the CPU oracle against the UNMODIFIED reference on the same cores: shows that the
oracle bench.py reports as `cpu_baseline` (kind "port") is a fair -- in fact stronger -- stand-in for the reference's
own CPU path.  Runs ONLY in the build container (needs /root/reference); prints one JSON line that DESIGN.md quotes.

    python tests/golden/cross_time_reference.py [--n 1000000 --q 10000]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd"))
sys.path.insert(0, "/root/reference")

import pyximport  # noqa: E402

pyximport.install(setup_args={"include_dirs": np.get_include()}, build_dir="/tmp/nlsh_oracle_pyxbld")
for _name, _attrs in (("siren", {"SIREN": object}), ("h5py", {"File": None})):
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        _m.__dict__.update(_attrs)
        sys.modules[_name] = _m
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

from encoders import MultiLayerRelu  # noqa: E402  (reference)
from nlsh.hashings import MultivariateBernoulli  # noqa: E402
from nlsh.indexer import Indexer  # noqa: E402
from nlsh.data import SIFT  # noqa: E402
from nlsh.learning.distances import MVBernoulliL2  # noqa: E402

from nlsh_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--q", type=int, default=10_000)
    args = ap.parse_args()
    N, Q, d, H, k, P = args.n, args.q, 128, 16, 10, 10
    corpus, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
    queries, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
    arrs = np.load(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
    Ws, bs = [arrs[f"W{i}"] for i in range(3)], [arrs[f"b{i}"] for i in range(3)]
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, MVBernoulliL2())
    linears = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for lin, W, b in zip(linears, Ws, bs):
            lin.weight.copy_(torch.from_numpy(W)); lin.bias.copy_(torch.from_numpy(b))
    hashing.train_mode(False)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    indexer = Indexer(hashing, torch.from_numpy(corpus), SIFT.distance)
    t_build = time.perf_counter() - t0
    torch.manual_seed(0)
    t0 = time.perf_counter()
    ids, ncand = indexer.query(torch.from_numpy(queries), k=k, hash_times=P)
    t_ref = time.perf_counter() - t0

    ox = oracle.OracleIndexer(Ws, bs, corpus)
    t0 = time.perf_counter()
    keys, nk = ox.hash_arrays(queries, hash_times=P)
    od, oi, nc = oracle.query_batch(ox.corpus, ox.perm, ox.uniq_keys, ox.offsets, queries, keys, nk, k)
    t_or = time.perf_counter() - t0
    print(json.dumps({"workload": f"SIFT1M-shaped N={N} Q={Q} H={H} k={k} hash_times={P}, learned hash, this container",
                      "reference": {"index_build_s": t_build, "query_s": t_ref, "queries_per_s": Q / t_ref, "torch_threads": threads,
                                    "mean_candidates": float(np.mean(ncand))},
                      "oracle": {"query_s": t_or, "queries_per_s": Q / t_or, "omp_threads": oracle.num_threads(),
                                 "mean_candidates": float(nc.mean())}}))


if __name__ == "__main__":
    main()

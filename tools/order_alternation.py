#!/usr/bin/env python3
"""Does walking the cells' schedule order in alternating directions on consecutive batches let the Infinity Cache serve part of a batch?
(r06 experiment; `Indexer.alternate_order`).  Sequential device-resident steps over 4 rotating batches, scan kernel by HIP events:
same order every batch | alternating | alternating behind a prefix of the largest cells | same order with the cache flushed between
batches (a 512-MiB write: what a batch costs when NOTHING of the previous one is left on the die).

    python tools/order_alternation.py glove|clusters|sift1m
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import Glove, SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "glove"
Q, B, steps = 10_000, 4, 40
ck = lambda n: os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", n)   # noqa: E731
if wl == "glove":
    corpus_h = synth.glove_manifold(1_183_514, 100, seed=synth.SEED_DATA)
    batches = [synth.glove_manifold(Q, 100, seed=synth.SEED_QUERY + 17 * i) for i in range(B)]
    Ws, bs = io.load_hasher_weights(ck("glove_manifold_h24.npz"))
    dist, compat = Glove.distance, False
else:
    gen = synth.sift_like if wl == "clusters" else synth.sift_manifold
    corpus_h, mean, std = synth.standardise(gen(1_000_000, 128, seed=synth.SEED_DATA))
    batches = [synth.standardise(gen(Q, 128, seed=synth.SEED_QUERY + 17 * i), mean, std)[0] for i in range(B)]
    Ws, bs = io.load_hasher_weights(ck("sift1m_clusters_h16.npz" if wl == "clusters" else "sift1m_manifold_h16.npz"))
    dist, compat = SIFT.distance, True
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=compat), torch.from_numpy(corpus_h).cuda(), dist, compat=compat)
qb = [torch.from_numpy(b).cuda() for b in batches]
ix.query_tensors(qb[0], k=10, hash_times=10, seed=1)
flush = torch.empty((1 << 27,), dtype=torch.float32, device="cuda")     # 512 MiB
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for a, b in ev:
    a.record(); b.record()


def run(alternate, keep=0, flush_between=False):
    Indexer.alternate_order, Indexer.alternate_keep = alternate, keep
    for i in range(4):
        ix.query_tensors(qb[i % B], k=10, hash_times=10, seed=10 + i, check=False)
    torch.cuda.synchronize()
    for i in range(steps):
        if flush_between:
            flush.fill_(float(i))
        ix.query_tensors(qb[i % B], k=10, hash_times=10, seed=100 + i, check=False, events=ev[i])
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return float(np.mean(ms)), float(np.min(ms))


out = {"workload": wl}
for rep in range(2):
    out[f"same_order_{rep}"] = run(False)
    out[f"alternating_{rep}"] = run(True)
    out[f"alternating_keep_256_{rep}"] = run(True, 256)
    out[f"alternating_keep_2048_{rep}"] = run(True, 2048)
out["same_order_cache_flushed_between_batches"] = run(False, flush_between=True)
Indexer.alternate_order = False
print(json.dumps(out))

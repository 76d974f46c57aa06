#!/usr/bin/env python3
"""What ONE hipGraphLaunch of a whole query batch costs the host, next to the eager enqueue of the same five launches (r06; VERDICT r05
item 5).  A small shard (1/8 of the headline corpus) so that the device is never the longer side: the step is captured once
(torch.cuda.CUDAGraph around `Indexer.query_tensors`: one `nlsh_query_batch` call = encode + lookup, bscan, bscatter, scan, bmerge) and
replayed; host time per enqueue = wall time of a burst of enqueues with NO synchronisation inside, the queue drained before it.
Also timed: `hipGraphExecKernelNodeSetParams`-free replay only -- a per-batch graph would add one such call per node whose arguments
change (the encode's batch pointer / seed, the scan's query pointer), each a runtime call of the same order as a launch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

N, d, Q = 125_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance)
q = torch.from_numpy(synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)[0]).cuda()
ix.query_tensors(q, k=10, hash_times=10, seed=1)
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ix.query_tensors(q, k=10, hash_times=10, seed=7, check=False)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    ix.query_tensors(q, k=10, hash_times=10, seed=7, check=False)
torch.cuda.synchronize()


def burst(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * host / n, 1e3 * (time.perf_counter() - t0) / n


out = {"shard_rows": N, "Q": Q}
for n in (8, 32):
    with torch.cuda.stream(side):
        out[f"eager_host_ms_per_batch_burst{n}"], out[f"eager_wall_ms_per_batch_burst{n}"] = burst(lambda: ix.query_tensors(q, k=10, hash_times=10, seed=7, check=False), n)
    out[f"graph_host_ms_per_batch_burst{n}"], out[f"graph_wall_ms_per_batch_burst{n}"] = burst(g.replay, n)
print(json.dumps(out))

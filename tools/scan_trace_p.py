#!/usr/bin/env python3
"""Per-wave phase totals of the PERSISTENT tiled scan (diagnostic; library built with EXTRA=-DNLSH_SCAN_TRACE)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import _capi, io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

N, d, Q = 1_000_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
queries_h, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
hashing = io.hashing_from_weights(Ws, bs, compat=True)
indexer = Indexer(hashing, torch.from_numpy(corpus_h).cuda(), SIFT.distance, algo="tiled")
queries = torch.from_numpy(queries_h).cuda()
for i in range(3):
    indexer.query_tensors(queries, k=10, hash_times=10, seed=7)
torch.cuda.synchronize()
L = _capi.lib()
buf = np.zeros((1 << 14, 8), dtype=np.float32)
rc = L.nlsh_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size))
assert rc == 0, rc
buf = buf[buf[:, 5] > 0]
names = ["total", "stage(barriers+lds write)", "compute", "epilogue(select)", "switch", "tasks", "lists"]
print(f"waves with work: {len(buf)}  (ticks = 10 ns)")
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {buf[:, i].mean():10.1f}  min {buf[:, i].min():9.1f}  max {buf[:, i].max():9.1f}  share {buf[:, i].sum() / buf[:, 0].sum():.3f}")
start = buf[:, 7]
print("  epilogue ticks per list:", buf[:, 3].sum() / buf[:, 6].sum())

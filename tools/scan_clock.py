#!/usr/bin/env python3
"""Shader clock the tiled scan kernel holds (diagnostic).  Needs a build with -DNLSH_SCAN_TRACE -DNLSH_SCAN_TRACE_CLOCK
(NLSH_HIP_LIB=...): every workgroup leaves the s_memtime (core cycles) and s_memrealtime (100 MHz) length of the same
interval; clock = ratio x 100 MHz, median over workgroups longer than 20 us, after ~2 s of back-to-back launches."""
import ctypes
import os
import sys
import time

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
indexer = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance, algo="tiled")
queries = torch.from_numpy(queries_h).cuda()
keys, nkeys = indexer.hash_device(queries, hash_times=10, seed=7)
indexer.scan_tensors(queries, keys, nkeys, k=10)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(200):
        indexer.scan_tensors(queries, keys, nkeys, k=10, check=False)
    torch.cuda.synchronize()
n_tasks = int(indexer.last_status.cpu()[0])
buf = np.zeros((min(n_tasks, 1 << 16), 8), dtype=np.float32)
assert _capi.lib().nlsh_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size)) == 0
b = buf[buf[:, 1] > 2000]
clk = b[:, 4] / b[:, 1] * 0.1
print(f"workgroups {len(b)} (>20 us of {len(buf)}): clock median {np.median(clk):.3f} GHz, p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}")

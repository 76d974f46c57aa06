#!/usr/bin/env python3
"""Phase breakdown of the tiled scan's workgroups (diagnostic).  Needs the library built with -DNLSH_SCAN_TRACE
(NLSH_HIP_LIB=path/to/that/build).  Runs the headline workload and prints, per task shape, where wave 0 of a
workgroup spends its time (100 MHz wall-clock ticks -> us): prologue (launch -> first stage issued), barrier-1 wait
(the slowest wave's previous k-block), stage (own loads + LDS write + barrier 2), compute, epilogue (selection)."""
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

import argparse  # noqa: E402
from nlsh_amd.data import Glove  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sift1m", choices=["sift1m", "clusters", "glove"])
ap.add_argument("--window", type=int, default=None, help="row window of the small-bucket packing (default: the facade's choice)")
args = ap.parse_args()
Q = 10_000
if args.workload == "glove":
    N, d = 1_183_514, 100
    corpus_h, queries_h = synth.glove_manifold(N, d, seed=synth.SEED_DATA), synth.glove_manifold(Q, d, seed=synth.SEED_QUERY)
    ck, dist_fn, compat = "glove_manifold_h24.npz", Glove.distance, False
else:
    N, d = 1_000_000, 128
    gen = synth.sift_manifold if args.workload == "sift1m" else synth.sift_like
    corpus_h, mean, std = synth.standardise(gen(N, d, seed=synth.SEED_DATA))
    queries_h, _, _ = synth.standardise(gen(Q, d, seed=synth.SEED_QUERY), mean, std)
    ck, dist_fn, compat = ("sift1m_manifold_h16.npz" if args.workload == "sift1m" else "sift1m_clusters_h16.npz"), SIFT.distance, True
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", ck))
hashing = io.hashing_from_weights(Ws, bs, compat=compat)
indexer = Indexer(hashing, torch.from_numpy(corpus_h).cuda(), dist_fn, compat=compat, algo="tiled", window_rows=args.window)
queries = torch.from_numpy(queries_h).cuda()
for i in range(3):
    indexer.query_tensors(queries, k=10, hash_times=10, seed=7)
torch.cuda.synchronize()
n_tasks = int(indexer.last_status.cpu()[0])
L = _capi.lib()
buf = np.zeros((min(n_tasks, 1 << 16), 8), dtype=np.float32)
rc = L.nlsh_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size))
assert rc == 0, rc
np.save(os.path.join(ROOT, 'gpurun_out', f'scan_trace_{args.workload}_{args.window}.npy'), buf)
buf = buf[buf[:, 0] > 0]
start = buf[:, 7].astype(np.int64)
nq_nrows = buf[:, 6].astype(np.int64)
buf[:, 6], buf[:, 7] = nq_nrows // 1000, nq_nrows % 1000
tot = buf[:, 0].sum()
print(f"tasks {n_tasks} traced {len(buf)}; summed workgroup time {tot / 100:.0f} us = {tot / 100 / 1792:.1f} us x 1792 slots")
print(f"{'queries':>8s} {'rows':>8s} {'tasks':>6s} {'time%':>6s} {'mean us':>8s} | {'prolog':>7s} {'bar1':>7s} {'stage':>7s} {'compute':>7s} {'select':>7s} {'other':>7s}")
for qlo, qhi in ((1, 1), (2, 4), (5, 8), (9, 12), (13, 15), (16, 16)):
    for rlo, rhi in ((1, 64), (65, 128), (129, 255), (256, 256)):
        m = (buf[:, 6] >= qlo) & (buf[:, 6] <= qhi) & (buf[:, 7] >= rlo) & (buf[:, 7] <= rhi)
        if not m.any():
            continue
        b = buf[m]
        mean = b[:, 0].mean()
        parts = [b[:, 1].mean(), b[:, 2].mean(), b[:, 4].mean(), b[:, 3].mean(), b[:, 5].mean()]
        other = mean - sum(parts)
        print(f"{qlo:3d}-{qhi:<4d} {rlo:3d}-{rhi:<4d} {m.sum():6d} {100 * b[:, 0].sum() / tot:6.1f} {mean / 100:8.1f} | " +
              " ".join(f"{v / 100:7.1f}" for v in parts + [other]))
b = buf
parts = [b[:, 1].sum(), b[:, 2].sum(), b[:, 4].sum(), b[:, 3].sum(), b[:, 5].sum()]
print("all tasks, share of summed time: prologue %.3f  barrier-1 %.3f  stage %.3f  compute %.3f  select %.3f  other %.3f" %
      tuple([v / tot for v in parts] + [1 - sum(parts) / tot]))
# concurrency over time: workgroups in flight per 10 us slice (start stamps are device-wide 100 MHz ticks, 24 bits kept)
start = (start - start.min()) % (1 << 24)
end = start + buf[:, 0].astype(np.int64)
span = int(end.max())
busy = np.zeros(span // 1000 + 1)
for s0, e0 in zip(start, end):
    busy[s0 // 1000:e0 // 1000 + 1] += 1
print(f"kernel span {span / 100:.1f} us (first entry -> last exit); workgroups in flight per 10 us slice (1792 = 7 per CU):")
print(" ".join(str(int(v)) for v in busy))

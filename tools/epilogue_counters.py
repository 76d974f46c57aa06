#!/usr/bin/env python3
"""What reaches the top-k selection of the tiled scan (diagnostic; prices a seeded per-query bound).
Needs a library built with `-DNLSH_SCAN_TRACE -DNLSH_SCAN_TRACE_EPILOGUE` (NLSH_HIP_LIB=.../libnlsh_hip_trace.so): every
(task, query) list adds to six counters in the last trace slot just before `select_k_smallest`:
lists, lists that saw a published bound, lists with 0 / fewer than k survivors of the bound, candidates, survivors."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import _capi, io, synth  # noqa: E402
from nlsh_amd.data import Glove, SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

SLOTS = 1 << 16


def counters():
    buf = np.zeros((SLOTS, 8), dtype=np.float32)
    rc = _capi.lib().nlsh_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size))
    assert rc == 0, rc
    return buf[SLOTS - 1].astype(np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sift1m", choices=["sift1m", "clusters", "glove"])
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
    ix = Indexer(hashing, torch.from_numpy(corpus_h).cuda(), dist_fn, compat=compat, algo="tiled")
    q = torch.from_numpy(queries_h).cuda()
    keys, nkeys = ix.hash_device(q, hash_times=10, seed=7)
    ix.scan_tensors(q, keys, nkeys, k=10)
    torch.cuda.synchronize()
    c0 = counters()
    n = 5
    for _ in range(n):
        ix.scan_tensors(q, keys, nkeys, k=10, check=False)
    torch.cuda.synchronize()
    c = (counters() - c0) / n
    lists, seen, none_live, few_live, cand, live = c[:6]
    print(json.dumps({"workload": args.workload, "tasks": int(ix.last_status.cpu()[0]), "lists_per_launch": lists,
                      "saw_a_published_bound": seen / lists, "zero_survivors": none_live / lists,
                      "fewer_than_k_survivors": few_live / lists, "full_bisection": 1 - (none_live + few_live) / lists,
                      "candidates_per_list": cand / lists, "survivors_per_list": live / lists}))


if __name__ == "__main__":
    main()

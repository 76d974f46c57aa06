#!/usr/bin/env python3
"""Per-rank step cost of the sharded headline workload WITHOUT the collective: builds rank `--rank`'s shard of
`--world` (contiguous rows, global ids, whole-corpus schedule statistics) on one GPU and times the local step
(encode_hash -> plan -> scan -> merge).  Shows the fixed per-step cost that bounds strong scaling; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/shard_step_profile.py --world 8 --rank 0 --steps 50
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--q", type=int, default=10_000)
    ap.add_argument("--shard", default="buckets", choices=["buckets", "rows"])
    ap.add_argument("--pipeline", action="store_true", help="three-stage pipeline over three streams (nlsh_amd/pipeline.py)")
    ap.add_argument("--split-front", default="auto", choices=["auto", "on", "off"], help="with --pipeline: the PLAN phase on a stream of its own (four stages); auto = the pipeline's own choice")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--graph-events", action="store_true", help="graph slots: pass scan events anyway (the batch is then launched eagerly on the slot's stream)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], help="with --pipeline: graph slots (one captured hipGraph per slot on its own stream) or the staged streams of r03-r05")
    args = ap.parse_args()
    from nlsh_amd import io, synth
    from nlsh_amd.data import SIFT
    from nlsh_amd.distributed import plan_bucket_shards, shard_range
    from nlsh_amd.indexer import Indexer

    dev = torch.device("cuda", 0)
    N, d, Q = 1_000_000, 128, args.q
    corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
    queries_h, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
    Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
    hashing = io.hashing_from_weights(Ws, bs, compat=True)
    queries = torch.from_numpy(queries_h).to(dev)
    full = torch.from_numpy(corpus_h).to(dev)
    keys, _ = hashing.hash_device(full, n=1)
    owner, stats = plan_bucket_shards(keys.view(-1), args.world)     # what ShardedIndexer derives from the all-gathered keys
    if args.shard == "rows":
        lo, hi = shard_range(N, args.rank, args.world)
        indexer = Indexer(hashing, full[lo:hi], SIFT.distance, compat=True, id_base=lo, schedule_stats=stats)
    else:
        sel = torch.nonzero(owner == args.rank).view(-1)
        lo, hi = 0, int(sel.numel())
        indexer = Indexer(hashing, full[sel], SIFT.distance, compat=True, row_ids=sel.int(), schedule_stats=stats)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record(); b.record()
    indexer.query_tensors(queries, k=10, hash_times=10, seed=1, want_keys=True, check=True)
    for i in range(3):
        indexer.query_tensors(queries, k=10, hash_times=10, seed=2 + i, want_keys=True, check=False)
    torch.cuda.synchronize()
    pipe = None
    if args.pipeline:
        from nlsh_amd.pipeline import QueryPipeline
        pipe = QueryPipeline(indexer, queries, k=10, hash_times=10, depth=args.depth, want_keys=True, split_front={"auto": None, "on": True, "off": False}[args.split_front],
                             graph={"auto": None, "on": True, "off": False}[args.graph])
        for i in range(3):
            pipe.submit(queries, seed=50 + i)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if pipe is not None:
            pipe.submit(queries, seed=100 + i, events=None if (pipe.graph and not args.graph_events) else ev[i])   # graph slots launch a batch eagerly when its scan is to be bracketed by events
        else:
            indexer.query_tensors(queries, k=10, hash_times=10, seed=100 + i, want_keys=True, check=False, events=ev[i])
    enqueue_ms = 1e3 * (time.perf_counter() - t0) / args.steps      # host time to enqueue a step (the GPU runs behind)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"world": args.world, "rank": args.rank, "shard": args.shard, "pipeline": bool(args.pipeline), "split_front": (pipe.split_front if pipe is not None else None), "graph_slots": (pipe.graph if pipe is not None else None), "depth": args.depth, "rows": hi - lo, "algo": indexer.last_algo,
                      "local_step_ms": 1e3 * el / args.steps, "host_enqueue_ms": enqueue_ms, "scan_ms": None if (pipe is not None and pipe.graph and not args.graph_events) else float(np.mean([a.elapsed_time(b) for a, b in ev]))}))


if __name__ == "__main__":
    main()

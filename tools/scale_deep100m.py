#!/usr/bin/env python3
"""Scale run for BASELINE.json configs[4] on ONE GPU: Deep100M-shaped corpus (N x 96-d fp32, unit norm,
HBM-resident: 38.4 GB + the bucket-sorted copy), 32-bit learned hash (full-width keys), 100k queries.

    python tools/scale_deep100m.py [--n 100000000 --q 100000 --hash-size 32]

The corpus is generated ON THE DEVICE in chunks (a host copy would be 38 GB); the hash is trained on a
1M-row sample with the minimal triplet trainer; recall is measured on a query sample against a chunked
brute force.  Prints one JSON line with build time, queries/s, the scan kernel's algorithmic GB/s and
size-independent property checks (sorted, members of the probed buckets, candidate counts).
Not the headline bench (bench.py); the 8-GPU form shards these rows with nlsh_amd.distributed.
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


def deep_manifold_device(n, d, seed, dev, latent_dim=10, n_clusters=256, spread=0.5, chunk=1 << 22, out=None):
    """Unit-norm rows on a low-dimensional manifold, generated on the device (torch Philox generator)."""
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    A1 = torch.randn((latent_dim, 96), generator=g, device=dev)
    b1 = torch.rand((96,), generator=g, device=dev) * 2 - 1
    A2 = torch.randn((96, d), generator=g, device=dev) / 9.8
    cen = torch.randn((n_clusters, latent_dim), generator=g, device=dev)
    g.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=dev) if out is None else out
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        which = torch.randint(0, n_clusters, (e - s,), generator=g, device=dev)
        z = cen[which] + spread * torch.randn((e - s, latent_dim), generator=g, device=dev)
        x = torch.tanh(z @ A1 + b1) @ A2 + 0.02 * torch.randn((e - s, d), generator=g, device=dev)
        out[s:e] = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--q", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=96)
    ap.add_argument("--hash-size", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--hash-times", type=int, default=10)
    ap.add_argument("--train-rows", type=int, default=1_000_000)
    ap.add_argument("--train-steps", type=int, default=5000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--recall-queries", type=int, default=1000)
    args = ap.parse_args()
    from nlsh_amd import training
    from nlsh_amd.data import SIFT
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.metrics import calculate_recall

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    N, Q, d, H, k, P = args.n, args.q, args.dim, args.hash_size, args.k, args.hash_times
    t0 = time.time()
    corpus = deep_manifold_device(N, d, 1234, dev)
    queries = deep_manifold_device(Q, d, 4321, dev)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    print(f"[deep] generated {N} x {d} on device in {gen_s:.1f}s ({corpus.numel() * 4 / 1e9:.1f} GB)", flush=True)

    # learned hash on a sample (unit-norm rows: L2 ranking == cosine ranking; Deep1B is searched with L2)
    sample = corpus[:: max(1, N // args.train_rows)][: args.train_rows].contiguous()
    knn = training.self_knn(sample, 10)
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, None, compat=H <= 16)
    t0 = time.time()
    training.fit_triplet(hashing, sample, knn, n_steps=args.train_steps, margin=1.0, log=lambda s: None)
    train_s = time.time() - t0
    del knn, sample

    torch.cuda.synchronize()
    t0 = time.time()
    indexer = Indexer(hashing, corpus, SIFT.distance, compat=H <= 16)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    stats = indexer.bucket_stats()
    print(f"[deep] index built in {build_s:.2f}s: {stats}", flush=True)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record(); b.record()
    out = indexer.query_tensors(queries, k=k, hash_times=P, seed=1, check=True)   # sizes the task table
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = indexer.query_tensors(queries, k=k, hash_times=P, seed=1, check=False, events=ev[i])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert int(indexer.last_status.cpu()[1]) == 0
    dist, idx, nc, _ = out
    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    sum_c = int(nc.long().sum().item())

    # properties (size independent)
    ok = idx >= 0
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()), "not ascending"
    keys, nkeys = indexer.hash_device(queries, hash_times=P, seed=1)
    valid = torch.arange(keys.shape[1], device=dev)[None, :] < nkeys[:, None]
    ck = indexer.corpus_keys[idx.clamp(min=0).long()]
    member = ((ck[:, :, None] == keys[:, None, :]) & valid[:, None, :]).any(-1)
    assert bool((member | ~ok).all()), "a result is not in a probed bucket"
    sel = slice(0, 2048)
    ref = torch.nn.functional.pairwise_distance(queries[sel, None, :].expand(-1, k, -1).reshape(-1, d),
                                                corpus[idx[sel].clamp(min=0).long().reshape(-1)]).reshape(-1, k)
    assert bool(((dist[sel] - ref).abs() <= 2e-5 * ref.clamp(min=1.0))[ok[sel]].all()), "distance mismatch"

    # recall on a sample (chunked brute force over the 100M rows)
    R = min(args.recall_queries, Q)
    qs = queries[:R]
    best_d = torch.full((R, k), float("inf"), device=dev)
    best_i = torch.full((R, k), -1, dtype=torch.int64, device=dev)
    qq = (qs * qs).sum(1)[:, None]
    for s in range(0, N, 1 << 22):
        c = corpus[s:s + (1 << 22)]
        dd = qq - 2.0 * (qs @ c.T) + (c * c).sum(1)[None, :]
        td, ti = dd.topk(k, dim=1, largest=False)
        cat_d, cat_i = torch.cat([best_d, td], 1), torch.cat([best_i, ti + s], 1)
        o = cat_d.topk(k, dim=1, largest=False).indices
        best_d, best_i = cat_d.gather(1, o), cat_i.gather(1, o)
    recall = float(np.mean(calculate_recall(list(best_i.cpu().numpy()), [r[r >= 0].tolist() for r in idx[:R].cpu().numpy()])))

    algo_bytes = 4.0 * d * sum_c
    print(json.dumps({
        "workload": f"configs[4] on ONE GPU: Deep100M-shaped, N={N} d={d} Q={Q} H={H} k={k} hash_times={P}",
        "corpus_gb": N * d * 4 / 1e9, "generate_s": gen_s, "train_s": train_s, "index_build_s": build_s,
        "n_buckets": stats["n_indexes"], "bucket_mean": stats["mean"], "bucket_max": stats["max"],
        "queries_per_s": Q * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
        "scan_kernel": {0: "query-major", 1: "bucket-major", 2: "bucket-major LDS-tiled"}[indexer.last_algo],
        "scan_ms": scan_ms, "mean_candidates_per_query": sum_c / Q, "algorithmic_GBps": algo_bytes / (scan_ms * 1e-3) / 1e9,
        "recall_at_10_on_sample": recall, "recall_sample": R, "properties": "ascending, members of probed buckets, distances vs torch: ok",
        "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9}), flush=True)


if __name__ == "__main__":
    main()

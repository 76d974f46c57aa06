#!/usr/bin/env python3
"""The reference's offline evaluation flow (eval.py:103-197) on the HIP path: load a hasher, hash the
corpus, build the index, then for n_samples = 1..N probe keys per query report the mean number of
candidates and recall@K -- the recall-vs-candidates trade-off curve eval.py prints (eval.py:196).

    python tools/eval_curve.py --model checkpoints/sift1m_manifold_h16.npz --data synth:sift1m [--max-samples 32]
    python tools/eval_curve.py --model run_cpu.pt --base base.fvecs --query query.fvecs --gt gt.ivecs --metric l2

Differences from eval.py, on purpose: keys are full width (eval.py's `_binarr_to_int`, eval.py:49-53),
every query is multi-probed (no trailing-batch rule), `<K` candidates return all of them (eval.py:185-186).
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
    ap.add_argument("--model", required=True, help=".npz | TorchScript _cpu.pt | state dict")
    ap.add_argument("--data", default=None, help="synth:sift1m | synth:glove (seeded generators of bench.py)")
    ap.add_argument("--base"); ap.add_argument("--query"); ap.add_argument("--gt")
    ap.add_argument("--metric", default=None, choices=["l2", "cosine"])
    ap.add_argument("-k", type=int, default=10)
    ap.add_argument("--max-samples", type=int, default=32)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--q", type=int, default=10000)
    ap.add_argument("--tanh", action="store_true")
    ap.add_argument("--seed", type=int, default=1, help="Philox seed of the probes: one stream, so the probe sets are nested in n_samples")
    args = ap.parse_args()
    from nlsh_amd import io as nio, synth
    from nlsh_amd.data import Glove, SIFT, brute_force_topk
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.metrics import calculate_recall

    model = args.model if os.path.exists(args.model) else os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", args.model)
    Ws, bs = nio.load_hasher_weights(model)
    if args.data == "synth:sift1m":
        n = args.n or 1_000_000
        corpus, mean, std = synth.standardise(synth.sift_manifold(n, 128))
        queries, _, _ = synth.standardise(synth.sift_manifold(args.q, 128, seed=synth.SEED_QUERY), mean, std)
        metric = "l2"
    elif args.data == "synth:glove":
        corpus, queries, metric = synth.glove_manifold(args.n or 1_183_514, 100), synth.glove_manifold(args.q, 100, seed=synth.SEED_QUERY), "cosine"
    else:
        rd = lambda p: nio.read_bvecs(p) if p.endswith(".bvecs") else nio.read_fvecs(p)  # noqa: E731
        corpus, queries, metric = rd(args.base), rd(args.query), args.metric or "l2"
    metric = args.metric or metric
    cg, qg = torch.from_numpy(corpus).cuda(), torch.from_numpy(queries).cuda()
    gt = nio.read_ivecs(args.gt)[:, :args.k] if args.gt else brute_force_topk(qg, cg, args.k, metric).cpu().numpy()

    hashing = nio.hashing_from_weights(Ws, bs, tanh_output=args.tanh, compat=False)
    t0 = time.time()
    indexer = Indexer(hashing, cg, SIFT.distance if metric == "l2" else Glove.distance, compat=False)
    torch.cuda.synchronize()
    print(f"# index: {indexer.bucket_stats()} built in {time.time() - t0:.3f}s", flush=True)
    print("n_samples avg_n_candidates recall qps")
    rows = []
    for n_samples in range(1, min(args.max_samples, 100) + 1):   # eval.py:148 range(1, 101)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist, idx, nc, _ = indexer.query_tensors(qg, k=args.k, hash_times=n_samples, seed=args.seed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ids = [r[r >= 0].tolist() for r in idx.cpu().numpy()]
        rec = float(np.mean(calculate_recall(list(gt), ids)))
        rows.append({"n_samples": n_samples, "avg_n_candidates": float(nc.float().mean()), "recall": rec, "qps": len(ids) / dt})
        print(n_samples, f"{rows[-1]['avg_n_candidates']:.1f}", f"{rec:.4f}", f"{rows[-1]['qps']:.0f}", flush=True)
    print(json.dumps({"model": args.model, "metric": metric, "k": args.k, "curve": rows}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Train a learned hash for a synthetic bench workload on the GPU box and save a portable checkpoint.

    python tools/train_hash.py --out gpurun_out/sift1m_like_h16.npz [--steps 6000 --balance 0.0 ...]
    python tools/train_hash.py --dataset /data/sift-128-euclidean.hdf5 --metric l2 --unit-norm --out sift1m_h16.npz   # real files

Uses nlsh_amd.training (stock autograd, reference's triplet recipe) and validates through the HIP
`Indexer` exactly like nlsh/trainers/base.py:80-108.  The checkpoint is our own artefact (weights
trained on seeded synthetic data), committed under neural-locality-sensitive-hashing_amd/checkpoints/.
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
    ap.add_argument("--workload", default="sift1m", choices=["sift1m", "sift1m_iso", "glove1m"])
    ap.add_argument("--dataset", default=None, help="REAL data: ann-benchmarks HDF5 (needs h5py) or a TEXMEX directory, read by nlsh_amd.data.SIFT / Glove; "
                                                     "the file's `train_knn` is used when present (precompute.py:91-97), else the self-kNN is computed here")
    ap.add_argument("--metric", default=None, choices=["l2", "cosine"], help="--dataset: which dataset class / distance (default l2)")
    ap.add_argument("--unit-norm", action="store_true", help="--dataset: standardise like SIFT(unit_norm=True) (nlsh/data.py:125-129)")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--q", type=int, default=10_000)
    ap.add_argument("--hash-size", type=int, default=16)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--margin", type=float, default=0.1)
    ap.add_argument("--positive-k", type=int, default=10)
    ap.add_argument("--balance", type=float, default=0.0)
    ap.add_argument("--neg-band", type=int, nargs=2, default=None, help="negatives from kNN ranks [lo, hi)")
    ap.add_argument("--knn-k", type=int, default=10)
    ap.add_argument("--tanh", action="store_true")
    ap.add_argument("--every", type=int, default=1000)
    ap.add_argument("--out", default="gpurun_out/hash.npz")
    args = ap.parse_args()

    from nlsh_amd import synth, training
    from nlsh_amd.data import Glove, SIFT, brute_force_topk
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli

    torch.manual_seed(0)
    knn_file = gt_file = None
    if args.dataset:
        metric = args.metric or "l2"
        ds = (SIFT if metric == "l2" else Glove)(args.dataset, unit_norm=args.unit_norm)
        ds.load()
        corpus, queries = np.ascontiguousarray(ds.training, np.float32), np.ascontiguousarray(ds.testing[: args.q], np.float32)
        d, dist_fn = corpus.shape[1], (SIFT.distance if metric == "l2" else Glove.distance)
        gt_file = np.asarray(ds.ground_truth)[: args.q, :10]
        try:
            knn_file = np.asarray(ds.training_self_knn)
        except AttributeError:
            knn_file = None
    elif args.workload in ("sift1m", "sift1m_iso"):
        d, metric, dist_fn = 128, "l2", SIFT.distance
        gen = synth.sift_manifold if args.workload == "sift1m" else synth.sift_like
        corpus, mean, std = synth.standardise(gen(args.n, d, seed=synth.SEED_DATA))
        queries, _, _ = synth.standardise(gen(args.q, d, seed=synth.SEED_QUERY), mean, std)
    else:
        d, metric, dist_fn = 100, "cosine", Glove.distance
        corpus = synth.glove_manifold(args.n, d, seed=synth.SEED_DATA)
        queries = synth.glove_manifold(args.q, d, seed=synth.SEED_QUERY)
    cg, qg = torch.from_numpy(corpus).cuda(), torch.from_numpy(queries).cuda()
    t0 = time.time()
    need_k = max(args.positive_k, args.knn_k, args.neg_band[1] if args.neg_band else 0)
    if knn_file is not None and knn_file.shape[1] >= need_k:
        knn = torch.from_numpy(knn_file[:, :need_k].astype(np.int64)).cuda()
    else:
        knn = training.self_knn(cg, need_k, metric=metric)
    gt = gt_file if gt_file is not None and gt_file.shape[1] >= 10 else brute_force_topk(qg, cg, 10, metric).cpu().numpy()
    torch.cuda.synchronize()
    print(f"[train] self-kNN + ground truth: {time.time() - t0:.1f}s", flush=True)
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), args.hash_size, None, tanh_output=args.tanh,
                                    compat=args.hash_size <= 16)
    validate = training.make_validator(hashing, cg, qg, gt, dist_fn)
    t0 = time.time()
    hist = training.fit_triplet(hashing, cg, knn, n_steps=args.steps, batch_size=args.batch_size, learning_rate=args.lr,
                                margin=args.margin, positive_k=args.positive_k, balance_weight=args.balance,
                                negative_band=tuple(args.neg_band) if args.neg_band else None,
                                validate=validate, test_every_updates=args.every)
    print(f"[train] {args.steps} steps in {time.time() - t0:.1f}s", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    meta = dict(vars(args), history=hist, dims=hashing.dims())
    np.savez_compressed(args.out, meta=json.dumps(meta), **training.export_weights(hashing))
    print(json.dumps(hist[-1] if hist else {}))


if __name__ == "__main__":
    main()

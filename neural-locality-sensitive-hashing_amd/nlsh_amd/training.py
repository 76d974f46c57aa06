"""Minimal trainer to obtain a *learned* hash -- stock PyTorch-ROCm autograd.

Not part of the query-time hot path and not accelerated: it exists because the headline metric is
quoted on a "16-bit learned hash" and no checkpoint exists offline.  It restates the reference's
triplet recipe: Adam(amsgrad) loop with periodic validation through `Indexer` (the live caller of the
hot path, nlsh/trainers/base.py:36-115), triplet loss on the L2 distance between Bernoulli code
vectors (nlsh/trainers/triplet.py:16-26, nlsh/learning/distances.py:245-254), anchors with a random
k-NN positive and a random negative (triplet.py:101-131), brute-force self-kNN of the training set
(precompute.py:57-67).  Validation = the HIP path (`Indexer.query`).
"""
import time
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .indexer import Indexer
from .metrics import calculate_recall


def self_knn(x: torch.Tensor, k: int, chunk: int = 4096, metric: str = "l2") -> torch.Tensor:
    """Row ids [n, k] of each row's k nearest OTHER rows (exact, chunked mm + topk on the device)."""
    n = x.shape[0]
    xn = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12) if metric == "cosine" else x
    sq = (xn * xn).sum(1)
    out = torch.empty((n, k), dtype=torch.int64, device=x.device)
    for s in range(0, n, chunk):
        q = xn[s:s + chunk]
        dist = sq[None, :] - 2.0 * (q @ xn.T) if metric == "l2" else -(q @ xn.T)
        dist[torch.arange(q.shape[0], device=x.device), torch.arange(s, s + q.shape[0], device=x.device)] = float("inf")
        out[s:s + chunk] = dist.topk(k, dim=1, largest=False).indices
    return out


def code_l2_rowwise(p, q):
    """MVBernoulliL2.rowwise (distances.py:247-254)."""
    return F.pairwise_distance(p, q)


def triplet_loss(anchor, pos, neg, distance_func=code_l2_rowwise, margin=0.1):
    return torch.clamp(distance_func(anchor, pos) - distance_func(anchor, neg) + margin, min=0).mean()


def triplet_batches(n: int, knn: torch.Tensor, positive_k: int, batch_size: int, generator: torch.Generator,
                    negative_band=None):
    """One epoch of (anchor, positive, negative) row-id batches: shuffled anchors, a random one of the
    first `positive_k` neighbours, a uniformly random negative (triplet.py:101-131, method "random").
    negative_band=(lo, hi): negatives are the anchor's neighbours of rank lo..hi-1 instead (a cheap
    stand-in for the reference's unimplemented "hard"/"semi-hard" modes, triplet.py:10-13)."""
    dev = knn.device
    anchors = torch.randperm(n, generator=generator, device=dev)
    cols = torch.randint(0, positive_k, (n,), generator=generator, device=dev)
    if negative_band is None:
        negs = torch.randint(0, n, (n,), generator=generator, device=dev)
    else:
        ncols = torch.randint(negative_band[0], negative_band[1], (n,), generator=generator, device=dev)
    for s in range(0, n - batch_size + 1, batch_size):
        a = anchors[s:s + batch_size]
        neg = negs[s:s + batch_size] if negative_band is None else knn[a, ncols[s:s + batch_size]]
        yield a, knn[a, cols[s:s + batch_size]], neg


def fit_triplet(hashing, train_vectors: torch.Tensor, knn: torch.Tensor, n_steps: int = 3000, batch_size: int = 1024,
                learning_rate: float = 3e-4, margin: float = 0.1, positive_k: int = 10, balance_weight: float = 0.0,
                negative_band=None, seed: int = 0, validate: Optional[Callable[[int], Dict]] = None, test_every_updates: int = 1000,
                log: Callable[[str], None] = print) -> List[Dict]:
    """Adam(amsgrad) triplet training (base.py:58-79).  `balance_weight` > 0 adds a bit-balance
    term (mean probability of every bit -> 0.5), an extension that is OFF by default."""
    gen = torch.Generator(device=train_vectors.device)
    gen.manual_seed(seed)
    opt = torch.optim.Adam(list(hashing.parameters()), lr=learning_rate, amsgrad=True)
    history, step = [], 0
    n = train_vectors.shape[0]
    while step < n_steps:
        for a, p, ng in triplet_batches(n, knn, positive_k, batch_size, gen, negative_band):
            hashing.train_mode(True)
            opt.zero_grad()
            pa, pp, pn = (hashing.predict(train_vectors[i]) for i in (a, p, ng))
            loss = triplet_loss(pa, pp, pn, margin=margin)
            if balance_weight > 0:
                loss = loss + balance_weight * ((pa.mean(0) - 0.5) ** 2).sum()
            loss.backward()
            opt.step()
            step += 1
            if validate is not None and (step % test_every_updates == 0 or step == n_steps):
                hashing.train_mode(False)
                rec = dict(step=step, loss=float(loss.detach()), **validate(step))
                history.append(rec)
                log(f"[train] {rec}")
            if step >= n_steps:
                break
    hashing.train_mode(False)
    return history


def make_validator(hashing, corpus_gpu, queries_gpu, ground_truth, distance_func, k=10, hash_times=10):
    """The reference's validation block (base.py:80-108): build an `Indexer`, time `query`,
    report n_indexes / std_index_rows / recall / query_size / qps -- through the HIP path."""
    def validate(step):
        indexer = Indexer(hashing, corpus_gpu, distance_func)
        torch.cuda.synchronize()
        t1 = time.time()
        recalls, n_candidates = indexer.query(queries_gpu, k=k, hash_times=hash_times)
        t2 = time.time()
        stats = indexer.bucket_stats()
        return {"test/n_indexes": stats["n_indexes"], "test/std_index_rows": stats["std_index_rows"],
                "test/recall": float(calculate_recall(list(ground_truth[:, :k]), recalls, np.mean)),
                "test/query_size": float(np.mean(n_candidates)), "test/qps": queries_gpu.shape[0] / (t2 - t1)}
    return validate


def export_weights(hashing) -> Dict[str, np.ndarray]:
    """Plain arrays (W0, b0, W1, ...) of the folded Linear stack: a portable checkpoint."""
    out = {}
    for i, (w, b) in enumerate(hashing.linear_stack()):
        out[f"W{i}"] = w.detach().cpu().numpy().astype(np.float32)
        if b is not None:
            out[f"b{i}"] = b.detach().cpu().numpy().astype(np.float32)
    return out


def load_weights(hashing, arrays) -> None:
    """Inverse of `export_weights` for encoders without BatchNorm (Linear layers in forward order)."""
    linears = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for i, lin in enumerate(linears):
            lin.weight.copy_(torch.as_tensor(arrays[f"W{i}"]))
            if lin.bias is not None and f"b{i}" in arrays:
                lin.bias.copy_(torch.as_tensor(arrays[f"b{i}"]))

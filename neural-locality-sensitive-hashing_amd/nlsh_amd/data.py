"""Point-space distances with the reference's signatures (nlsh/data.py:99-109, 191-201) plus
synthetic stand-ins for the HDF5-backed dataset classes (no data files exist offline).

`SIFT.distance` / `Glove.distance` carry a `metric` tag; `Indexer` uses it to select the fused
gfx950 scan kernel.  Their torch bodies only serve callers that invoke them directly.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import synth


def _tag(metric):
    def deco(fn):
        fn.nlsh_metric = metric
        return fn
    return deco


@_tag("l2")
def l2_distance(v1, v2):
    """(d), (n, d) -> (n): ||v1 - v2 + 1e-6||_2, i.e. F.pairwise_distance (nlsh/data.py:201)."""
    return F.pairwise_distance(v1, v2)


@_tag("cosine")
def cosine_distance(v1, v2):
    """(d), (n, d) -> (n): 1 - cos(v1, v2) (nlsh/data.py:109)."""
    return 1 - F.cosine_similarity(v1, v2, dim=-1)


def metric_of(distance_func):
    """'l2' | 'cosine' | None for a caller-supplied distance callable."""
    fn = getattr(distance_func, "__func__", distance_func)
    return getattr(fn, "nlsh_metric", None)


class _SyntheticSet:
    """Same attribute surface the trainers/eval read: training, testing, ground_truth, distance, load()."""
    metric = "l2"

    def __init__(self, n_train, n_test, dim, k=100, seed=synth.SEED_DATA):
        self._n_train, self._n_test, self._dim, self._k, self._seed = n_train, n_test, dim, k, seed
        self.training = self.testing = self.ground_truth = None

    def _generate(self, n, seed):
        raise NotImplementedError

    def load(self, ground_truth=True):
        self.training = self._generate(self._n_train, self._seed)
        self.testing = self._generate(self._n_test, self._seed + 1)
        if ground_truth:
            self.ground_truth = brute_force_topk(self.testing, self.training, self._k, self.metric).cpu().numpy()
        self.prepared = True

    @property
    def dim(self):
        return self._dim


class SIFT(_SyntheticSet):
    metric = "l2"
    distance = staticmethod(l2_distance)

    def __init__(self, path=None, unit_norm=False, n_train=10000, n_test=100, dim=128, **kw):
        super().__init__(n_train, n_test, dim, **kw)
        self._unit_norm = unit_norm
        self._stats = None

    def _generate(self, n, seed):
        x = synth.sift_like(n, self._dim, seed=seed)
        if self._unit_norm:  # reference: per-dimension standardisation (nlsh/data.py:125-129)
            if self._stats is None:
                x, mean, std = synth.standardise(x)
                self._stats = (mean, std)
            else:
                x, _, _ = synth.standardise(x, *self._stats)
        return x


class Glove(_SyntheticSet):
    metric = "cosine"
    distance = staticmethod(cosine_distance)

    def __init__(self, path=None, n_train=10000, n_test=100, dim=100, **kw):
        super().__init__(n_train, n_test, dim, **kw)

    def _generate(self, n, seed):
        return synth.glove_like(n, self._dim, seed=seed)


def brute_force_topk(queries, corpus, k, metric="l2", chunk=1024, device=None):
    """Exact ground truth (row ids [Q,k], ascending distance) with stock torch ops, chunked.

    Measurement harness only (the reference's precompute.py:57-67 does the same with mm+topk).
    """
    q = torch.as_tensor(queries)
    c = torch.as_tensor(corpus)
    if device is not None:
        q, c = q.to(device), c.to(device)
    if metric == "cosine":
        c = c / c.norm(dim=1, keepdim=True).clamp_min(1e-12)
        q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    c_sq = (c * c).sum(1)
    out = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
    for s in range(0, q.shape[0], chunk):
        qq = q[s:s + chunk]
        if metric == "l2":
            dist = c_sq[None, :] - 2.0 * (qq @ c.T) + (qq * qq).sum(1)[:, None]
        else:
            dist = 1.0 - qq @ c.T
        out[s:s + chunk] = dist.topk(k, dim=1, largest=False).indices
    return out

"""Point-space distances with the reference's signatures (nlsh/data.py:99-109, 191-201) and its dataset classes
(nlsh/data.py:14-46,112-167): `SIFT(path, unit_norm)` / `Glove(path, unit_norm, unit_ball)` READ THE FILE they are
given -- an ann-benchmarks HDF5 (`train` / `test` / `neighbors` [/ `train_knn`], needs h5py) or a TEXMEX directory
(`*_base.fvecs`, `*_query.fvecs`, `*_groundtruth.ivecs` [, `*_train_knn.ivecs`]) -- and raise if they cannot.  The
seeded synthetic stand-ins the offline benchmarks use are separate, explicitly named classes (`SyntheticSIFT`,
`SyntheticGlove`): nothing silently substitutes generated data for a dataset path.

`SIFT.distance` / `Glove.distance` carry a `metric` tag; `Indexer` uses it to select the fused
gfx950 scan kernel.  Their torch bodies only serve callers that invoke them directly.
"""
import glob
import os

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


def norm_to_unit_sphere(arr):
    """nlsh/data.py:9-10."""
    return arr / np.linalg.norm(arr, axis=1)[:, np.newaxis]


def _read_dataset(path):
    """path -> dict(training, testing, ground_truth[, training_self_knn]) as numpy arrays."""
    from . import io
    if path is None:
        raise ValueError("a dataset path is required (use SyntheticSIFT / SyntheticGlove for the seeded stand-ins)")
    path = os.fspath(path)
    if os.path.isdir(path):
        def one(pattern, reader, required=True):
            hits = sorted(glob.glob(os.path.join(path, pattern)))
            if not hits:
                if required:
                    raise FileNotFoundError(f"{path}: no file matches {pattern}")
                return None
            return reader(hits[0])
        base = lambda p_: io.read_bvecs(p_) if p_.endswith(".bvecs") else io.read_fvecs(p_)  # noqa: E731
        out = {"training": one("*base.[fb]vecs", base), "testing": one("*query.[fb]vecs", base),
               "ground_truth": one("*groundtruth.ivecs", io.read_ivecs)}
        knn = one("*train_knn.ivecs", io.read_ivecs, required=False)
        if knn is not None:
            out["training_self_knn"] = knn
        return out
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return io.load_hdf5(path, with_train_knn=True)          # ImportError if h5py is missing: no silent substitute


class _Dataset:
    """Attribute surface the trainers / eval read (nlsh/data.py:48-96,142-188): prepared, dim, training, testing,
    ground_truth, training_self_knn, distance; `load()` fills them."""
    metric = "l2"

    def __init__(self):
        self._prepared = False
        self._training = self._testing = self._ground_truth = self._training_self_knn = None

    def _check_prepared(self):
        if not self._prepared:
            raise ValueError(f"{self.__class__.__name__} is not prepared. call `load` beforehand.")

    prepared = property(lambda self: self._prepared)

    @property
    def dim(self):
        self._check_prepared()
        return self._training.shape[1]

    @property
    def training(self):
        self._check_prepared()
        return self._training

    @property
    def testing(self):
        self._check_prepared()
        return self._testing

    @property
    def ground_truth(self):
        self._check_prepared()
        return self._ground_truth

    @property
    def training_self_knn(self):
        self._check_prepared()
        if self._training_self_knn is None:   # the reference raises AttributeError here (nlsh/data.py:36-41): `precompute.py` not run
            raise AttributeError("no train_knn in the dataset: run the self-kNN precompute (nlsh_amd.training.self_knn)")
        return self._training_self_knn

    def _standardise(self):                   # nlsh/data.py:27-31,125-129 (numpy default float64 accumulation, cast back)
        mean, std = self._training.mean(0), self._training.std(0)
        self._training = ((self._training - mean) / std).astype(np.float32)
        self._testing = ((self._testing - mean) / std).astype(np.float32)


class _FileDataset(_Dataset):
    def __init__(self, path, unit_norm=False, unit_ball=False):
        super().__init__()
        if path is None:
            raise ValueError("a dataset path is required (use SyntheticSIFT / SyntheticGlove for the seeded stand-ins)")
        self._path, self._unit_norm, self._unit_ball = path, unit_norm, unit_ball

    def load(self):
        arrays = _read_dataset(self._path)
        self._training = np.asarray(arrays["training"], dtype=np.float32)
        self._testing = np.asarray(arrays["testing"], dtype=np.float32)
        if self._unit_norm:
            self._standardise()
        if self._unit_ball:                   # nlsh/data.py:33-35 (Glove only)
            self._training, self._testing = norm_to_unit_sphere(self._training), norm_to_unit_sphere(self._testing)
        self._ground_truth = np.asarray(arrays["ground_truth"])
        self._training_self_knn = arrays.get("training_self_knn")
        self._prepared = True


class SIFT(_FileDataset):
    """nlsh/data.py:112-201."""
    metric = "l2"
    distance = staticmethod(l2_distance)

    def __init__(self, path, unit_norm=False):
        super().__init__(path, unit_norm=unit_norm)


class Glove(_FileDataset):
    """nlsh/data.py:14-109."""
    metric = "cosine"
    distance = staticmethod(cosine_distance)

    def __init__(self, path, unit_norm=False, unit_ball=False):
        super().__init__(path, unit_norm=unit_norm, unit_ball=unit_ball)


class _SyntheticSet(_Dataset):
    """Seeded generated stand-in with the same attribute surface (no dataset files exist offline)."""

    def __init__(self, n_train, n_test, dim, k=100, seed=synth.SEED_DATA, unit_norm=False, with_train_knn=False):
        super().__init__()
        self._n_train, self._n_test, self._d, self._k, self._seed = n_train, n_test, dim, k, seed
        self._unit_norm, self._with_train_knn = unit_norm, with_train_knn

    def _generate(self, n, seed):
        raise NotImplementedError

    def load(self, ground_truth=True):
        self._training = self._generate(self._n_train, self._seed)
        self._testing = self._generate(self._n_test, self._seed + 1)
        if self._unit_norm:
            self._standardise()
        if ground_truth:
            self._ground_truth = brute_force_topk(self._testing, self._training, self._k, self.metric).cpu().numpy()
        if self._with_train_knn:
            knn = brute_force_topk(self._training, self._training, self._k + 1, self.metric).cpu().numpy()
            self._training_self_knn = knn[:, 1:]
        self._prepared = True


class SyntheticSIFT(_SyntheticSet):
    metric = "l2"
    distance = staticmethod(l2_distance)

    def __init__(self, n_train=10000, n_test=100, dim=128, manifold=False, **kw):
        super().__init__(n_train, n_test, dim, **kw)
        self._gen = synth.sift_manifold if manifold else synth.sift_like

    def _generate(self, n, seed):
        return self._gen(n, self._d, seed=seed)


class SyntheticGlove(_SyntheticSet):
    metric = "cosine"
    distance = staticmethod(cosine_distance)

    def __init__(self, n_train=10000, n_test=100, dim=100, manifold=False, **kw):
        super().__init__(n_train, n_test, dim, **kw)
        self._gen = synth.glove_manifold if manifold else synth.glove_like

    def _generate(self, n, seed):
        return self._gen(n, self._d, seed=seed)


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

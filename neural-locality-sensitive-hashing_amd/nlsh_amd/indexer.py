"""`build_index` / `Indexer` with the reference's surface (nlsh/indexer.py:6-96) on gfx950.

Index build: hash the corpus on the device (`encode_hash`), stable radix sort of (key, row)
-> CSR (`perm`, `uniq_keys`, `offsets`) and a bucket-contiguous copy of the corpus.
Query: ONE `encode_hash` launch for the whole batch (multi-probe keys stay on the device) and
ONE `nlsh_scan_topk` call (plan -> scan -> merge kernels) instead of the reference's per-query
Python loop with ~(P+4) launches and a device sync per query.

`query()` returns the reference's `(List[List[int]], List[int])`; `query_tensors()` is the same
computation with device-resident results (what bench.py times).  compat=True (default) keeps the
reference quirks: int16 key wrap (F2), single-probe trailing partial batch (F6), `<k` fallback =
last key's bucket rows (F7).  No CPU fallback exists anywhere in this module.
"""
import gc
import threading
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch

from . import _capi
from .data import metric_of
from .hashings import host_key_set, keys_to_sets

HASH_BATCH = 4096  # nlsh/indexer.py:40 default batch_size


class _CollectorPause:
    """The cyclic collector paused for a region, re-entrant and shared by every thread of the process (the collector's switch is one
    process-wide flag, so the pause has to be one process-wide object): the FIRST region to open records the state the application
    left the collector in and disables it, regions opened meanwhile -- nested in the same thread (`query` -> `_plain_lists`) or by
    `query()` calls of other threads -- only count, and the LAST one to close restores what the first one found.  A caller that had
    the collector disabled gets it back disabled; a caller that enables it while a region is open keeps it enabled (the exit only
    ever ENABLES, and only when the pause itself disabled).  What the gc module cannot show is another thread calling
    `gc.disable()` while a region is open (a no-op on an already disabled collector): that thread finds the collector enabled again
    once the last region closes -- INTEGRATION.md says so.

    `application_state()` is the collector state the application chose: what the first open region found, `gc.isenabled()` when none
    is open.  `Indexer.promote_results` asks it (inside a paused `query()` `gc.isenabled()` is always False)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._depth = 0
        self._found_enabled = False

    def __enter__(self):
        with self._lock:
            if self._depth == 0:
                self._found_enabled = gc.isenabled()
                if self._found_enabled:
                    gc.disable()
            self._depth += 1
        return self

    def __exit__(self, *exc):
        with self._lock:
            self._depth -= 1
            if self._depth == 0 and self._found_enabled:
                self._found_enabled = False
                gc.enable()
        return False

    def application_state(self):
        with self._lock:
            return self._found_enabled if self._depth else gc.isenabled()


_collector_pause = _CollectorPause()


def _load_fastlists():
    """csrc/fastlists.c (built by the Makefile next to the HIP library): the result lists in one tight C loop.  Host-side and
    optional -- `ndarray.tolist()` builds the identical lists -- unlike the HIP library, which has no substitute."""
    import importlib.machinery
    import importlib.util
    import os
    # only a build for THIS interpreter's ABI (the Makefile names the file with `python3-config --extension-suffix`)
    path = os.path.join(os.path.dirname(_capi.LIB_PATH), "_nlsh_fastlists" + importlib.machinery.EXTENSION_SUFFIXES[0])
    if not os.path.exists(path):
        return None
    try:
        spec = importlib.util.spec_from_file_location("_nlsh_fastlists", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.rows_to_lists
    except (ImportError, OSError):
        return None


_rows_to_lists = _load_fastlists()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def build_csr_device(keys: torch.Tensor):
    """int32 keys [N] (device) -> (perm int32 [N], uniq_keys int32 [nb], offsets int32 [nb+1])."""
    L = _capi.lib()
    keys = keys.contiguous()
    n = keys.shape[0]
    dev = keys.device
    perm = torch.empty((n,), dtype=torch.int32, device=dev)
    uniq = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    offs = torch.empty((n + 1,), dtype=torch.int32, device=dev)
    nb = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = L.nlsh_build_csr_workspace(n)
    if ws_bytes == 0:
        _capi.check(_capi.E_HIP)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    _capi.check(L.nlsh_build_csr(_capi.ptr(keys), n, _capi.ptr(perm), _capi.ptr(uniq), _capi.ptr(offs), _capi.ptr(nb),
                                 _capi.ptr(ws), ws_bytes, _stream(dev)))
    n_buckets = int(nb.item())  # index build may synchronise; the query path never does
    return perm, uniq[:n_buckets], offs[:n_buckets + 1]


def build_index(indexes, cuda=True) -> Dict[int, torch.Tensor]:
    """nlsh/indexer.py:6-24: {key: LongTensor of the rows whose key set contains it, ascending}.

    cuda=True sorts the (key, row) pairs with the device radix sort; cuda=False is the host-only
    variant the reference's own unit test uses (nlsh/tests/test_indexer.py) and needs no GPU.
    """
    rows, keys = [], []
    for row, key_set in enumerate(indexes):
        for key in key_set:
            rows.append(row)
            keys.append(int(key))
    if not keys:
        return {}
    keys_np = np.asarray(keys, dtype=np.int64)
    rows_t = torch.as_tensor(np.asarray(rows, dtype=np.int64))
    if cuda:
        if keys_np.min() < -(1 << 31) or keys_np.max() >= (1 << 32):
            raise ValueError("bucket keys must fit 32 bits")
        k32 = torch.as_tensor(keys_np.astype(np.uint32).view(np.int32) if keys_np.max() >= (1 << 31)
                              else keys_np.astype(np.int32)).cuda()
        perm, uniq, offs = build_csr_device(k32)
        grouped = rows_t.cuda()[perm.long()]
        sizes = torch.diff(offs).cpu().tolist()
        # report keys as the caller passed them (first pair of each bucket)
        first = perm[offs[:-1].long()].cpu().numpy()
        names = keys_np[first].tolist()
    else:
        order = torch.sort(torch.as_tensor(keys_np), stable=True)
        uniq, counts = torch.unique_consecutive(order.values, return_counts=True)
        grouped = rows_t[order.indices]
        sizes = counts.tolist()
        names = uniq.tolist()
    return {int(k): chunk for k, chunk in zip(names, torch.split(grouped, sizes))}


class Indexer:

    def __init__(self, hashing, candidate_vectors_gpu, distance_func, compat=True, metric: Optional[str] = None,
                 seg_rows: int = 0, id_base: int = 0, algo: Optional[str] = None,
                 row_ids: Optional[torch.Tensor] = None, schedule_stats: Optional[Tuple[float, float]] = None,
                 corpus_keys: Optional[torch.Tensor] = None, l2_form: str = "exact", window_rows: Optional[int] = None,
                 row_align: int = 4):
        self._hashing = hashing
        self._candidate_vectors_gpu = candidate_vectors_gpu
        self._distance_func = distance_func
        self.compat = compat
        self.metric = metric or metric_of(distance_func)
        self.seg_rows = seg_rows
        self.algo = algo            # None = choose per batch; "query" | "bucket" | "tiled" force a schedule
        # (size-biased bucket size, row count) of the WHOLE corpus when this index holds a shard of it (the sharded
        # builds compute them from the all-gathered keys): the schedule is chosen from whole-corpus statistics, so every
        # shard count runs the same arithmetic and results stay bit-identical
        self.schedule_stats = schedule_stats
        self.id_base = int(id_base)
        # global row id of every local row (int32 [N], device) when the shard is not a contiguous range
        # (bucket-sharded corpus); default: id_base + local row
        self.row_ids = row_ids
        # bucket key of every row (int32 [N], device) when the caller already has them (keys recorded by another run of
        # the same hash: parity on an identical index); default: hash the corpus here (indexer.py:36-38)
        self._given_keys = corpus_keys
        # "exact" (default): sqrt(sum(((q - c) + 1e-6)^2)) in F.pairwise_distance's operation order, bit-identical to the oracle.
        # "folded" (opt-in, NLSH_METRIC_L2_EPS_FOLDED): sqrt(sum(((q + 1e-6) - c)^2)) -- 2 instead of 3 vector operations per element in
        # the LDS-tiled schedule, one rounding per element away from the reference's order (|d - d_exact| <= 1e-4 * max(1, d))
        if l2_form not in ("exact", "folded"):
            raise ValueError("l2_form must be 'exact' or 'folded'")
        self.l2_form = l2_form
        # Small-bucket packing of the tiled schedule (include/nlsh_hip.h, nlsh_build_cells): consecutive buckets of <= window_rows rows
        # share one row window and its tasks.  None = choose per batch (`choose_window`), 0 = off (one task list per bucket, r03),
        # 64 / 128 / 256 = forced.  Never changes a result bit, only the number of tasks.
        if window_rows is not None and not 0 <= int(window_rows) <= 256:
            raise ValueError("window_rows must be None (auto) or in [0, 256]")
        self.window_rows = None if window_rows is None else int(window_rows)
        # Row stride of the bucket-sorted corpus copy, in floats: ceil(d / row_align) * row_align.  4 (default) packs rows at 16-byte
        # granularity; 32 starts every row on a 128-byte line of the L2 at the price of padding (100-d: 400 -> 512 bytes per row).
        # Never changes a result (padding columns are never read: the kernels walk d, not the stride).
        if row_align % 4 or row_align < 4:
            raise ValueError("row_align must be a multiple of 4 floats")
        self.row_align = int(row_align)
        self._cells = {}            # window_rows -> (cell_of, cell_offsets, cell_order, n_cells), built on first use
        self._index2row = None
        self._perm_host = None
        self._e_sb = None
        self._pin = None            # pinned host staging of query()'s single device->host copy
        self._ws = {}               # scan workspace per stream (concurrent query batches on different HIP streams)
        self._max_tasks = {}
        self._build_index()

    # ------------------------------------------------------------------ build
    def _build_index(self):
        corpus = self._candidate_vectors_gpu
        if corpus.device.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "Indexer needs a device-resident corpus; there is no CPU path")
        L = _capi.lib()
        N, d = corpus.shape
        if corpus.dtype != torch.float32 or corpus.stride(1) != 1:
            corpus = corpus.float().contiguous()
        if self._given_keys is not None:
            keys = self._given_keys
            if keys.shape != (N,) or keys.dtype != torch.int32 or keys.device != corpus.device:
                raise ValueError("corpus_keys must be int32 [N] on the corpus device")
            keys = keys.contiguous()
        else:
            keys, _ = self.hash_device(corpus, hash_times=1)       # indexer.py:36-38: hash(candidates, hash_times=1), in 4096-row batches when the module's mode makes batches matter
        self.corpus_keys = keys.view(-1)
        self.perm, self.uniq_keys, self.offsets = build_csr_device(self.corpus_keys)
        self.n_buckets = int(self.uniq_keys.shape[0])
        self.dim = d
        align = max(4, int(self.row_align))       # floats; 4 = 16-byte rows (the default), 32 = every row starts on a 128-byte line
        self.row_stride = (d + align - 1) // align * align
        dev = corpus.device
        self.corpus_sorted = torch.empty((N, self.row_stride), dtype=torch.float32, device=dev)
        self.gid = torch.empty((N,), dtype=torch.int32, device=dev)
        self.inv_norm = torch.empty((N,), dtype=torch.float32, device=dev) if self.metric == "cosine" else None
        _capi.check(L.nlsh_gather_rows(_capi.ptr(corpus), corpus.stride(0) if N else d, d, _capi.ptr(self.perm), N,
                                       _capi.ptr(self.corpus_sorted), self.row_stride, _capi.ptr(self.inv_norm),
                                       _capi.ptr(self.gid), self.id_base, _stream(dev)))
        if self.row_ids is not None:
            if self.row_ids.shape[0] != N or self.row_ids.dtype != torch.int32:
                raise ValueError("row_ids must be int32 [N]")
            self.gid = self.row_ids[self.perm.long()].contiguous()
        # schedule order of the buckets (largest first) for the bucket-major scans: static per index
        self.bucket_order = torch.empty((max(self.n_buckets, 1),), dtype=torch.int32, device=dev)
        if self.n_buckets:
            ob = L.nlsh_bucket_order_workspace(self.n_buckets)
            if ob == 0:
                _capi.check(_capi.E_HIP)
            ows = torch.empty((ob,), dtype=torch.uint8, device=dev)
            _capi.check(L.nlsh_bucket_order(_capi.ptr(self.offsets), self.n_buckets, _capi.ptr(self.bucket_order), _capi.ptr(ows),
                                            ob, _stream(dev)))
        self._uniq_host = self.uniq_keys.cpu().numpy()
        self._offs_host = self.offsets.cpu().numpy().astype(np.int64)
        self.bucket_sizes = np.diff(self._offs_host)

    @property
    def index2row(self) -> Dict[int, torch.Tensor]:
        """Reference attribute (indexer.py:38; read by trainers/base.py:87-89): materialised lazily
        as views of one int64 copy of `perm`."""
        if self._index2row is None:
            rows = self.gid.long()
            names = self._uniq_host.astype(np.int64)
            if self._hashing.key_mode == _capi.KEY_FULL:
                names = names & 0xFFFFFFFF
            self._index2row = {int(k): v for k, v in zip(names.tolist(), torch.split(rows, self.bucket_sizes.tolist()))}
        return self._index2row

    def bucket_stats(self):
        s = self.bucket_sizes
        if len(s) == 0:
            return {"n_indexes": 0, "mean": 0.0, "median": 0.0, "max": 0, "std_index_rows": 0.0}
        return {"n_indexes": int(len(s)), "mean": float(s.mean()), "median": float(np.median(s)), "max": int(s.max()),
                "std_index_rows": float(s.std())}

    # ------------------------------------------------------------------ hashing
    def _n_multi_rows(self, n, batch_size=HASH_BATCH):
        # indexer.py:43-53: only full batches get hash_times; the trailing partial batch is n=1 (F6)
        return (n // batch_size) * batch_size if self.compat else n

    def hash_device(self, query_vectors, batch_size=HASH_BATCH, hash_times=1, seed=None, out=None):
        needs_train = getattr(self._hashing, "_needs_train_forward", None)
        if needs_train is not None and needs_train() and query_vectors.shape[0] > batch_size:
            # BatchNorm encoder in TRAIN mode: the reference feeds the module `batch_size`-row batches (nlsh/indexer.py:40-53), so the
            # batch statistics -- and the running-statistics updates -- are per batch, the trailing partial batch on its own
            # statistics with n = 1 (F6).  One forward over all rows would use other statistics and hold every activation at once.
            n, parts = query_vectors.shape[0], []
            full = (n // batch_size) * batch_size
            for lo in range(0, full, batch_size):
                parts.append(self._hashing.hash_device(query_vectors[lo:lo + batch_size], n=hash_times, seed=seed))
            if full < n:
                tail_n = 1 if self.compat else hash_times
                tk, tn = self._hashing.hash_device(query_vectors[full:], n=tail_n, seed=seed)
                if tk.shape[1] < hash_times:
                    tk = torch.nn.functional.pad(tk, (0, hash_times - tk.shape[1]))
                parts.append((tk, tn))
            keys, nkeys = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            if out is not None:
                out[0].copy_(keys)
                out[1].copy_(nkeys)
                keys, nkeys = out
            return keys, nkeys
        return self._hashing.hash_device(query_vectors, n=hash_times,
                                         n_multi_rows=self._n_multi_rows(query_vectors.shape[0], batch_size), seed=seed, out=out)

    def hash(self, query_vectors, batch_size=HASH_BATCH, hash_times=1) -> List[Set[int]]:
        keys, nkeys = self.hash_device(query_vectors, batch_size, hash_times)
        return keys_to_sets(keys, nkeys, self._hashing.key_mode)

    # ------------------------------------------------------------------ query
    def _size_biased_bucket(self):
        if self._e_sb is None:                                       # static per index; every batch asks for it
            s = self.bucket_sizes.astype(np.float64)
            n = max(float(s.sum()), 1.0)
            self._e_sb = float((s * s).sum() / n) if len(s) else 0.0  # expected size of the bucket a point lands in
        return self._e_sb

    def choose_algo(self, Q, P):
        """The LDS-tiled bucket-major schedule from 64 (query, probe) pairs per batch on, the query-major stream below that.

        r04, with small buckets sharing 64-row windows (`choose_window`), tools/scan_bench.py --queries Q --algo ..., scan kernel /
        whole device step in ms on one box (profiles/r04_algo_by_batch_size.txt):
                            Q = 1          16           200          1000         2500         5000         10000
            GloVe   tiled   .011 / .113   .013 / .082   .017 / .082  .038 / .101  .068 / .135  .095 / .168  .123 / .22
                    query   .043 / .093   .044 / .104   .050 / .135  .057 / .143  .071 / .158  .114 / .203  .237 (r03)
            SIFT1M  tiled   .014 / .074   .014 / .074   .020 / .074  .051 / .104  .075 / .131  .163 / .222  .24
                    query   .031 / .067   .046 / .080   .054 / .092  .069 / .108  .146 / .189  .846 / .924  1.65 (r01)
        (the SURVEY-generator SIFT1M index behaves like GloVe; the wave-level bucket-major schedule, algo="bucket", lost every cell of
        this table by 2-4x and is only reachable by name).  r03's rule sent balanced hashes with tiny buckets and small batches to the
        query-major stream because a task per (bucket, query group) cost ~9 us of latency per pair; windows removed that.  Only a
        batch of a handful of queries still prefers the query-major schedule's shorter PLAN phase (one kernel instead of four).
        The rule depends on the batch shape alone, so every shard count picks the same arithmetic (`schedule_stats` is kept for the
        sharded builds' interface and no longer enters)."""
        if self.algo is not None:
            return {"query": _capi.SCAN_QUERY_MAJOR, "bucket": _capi.SCAN_BUCKET_MAJOR, "tiled": _capi.SCAN_BUCKET_TILED}[self.algo]
        return _capi.SCAN_BUCKET_TILED if Q * P >= 64 else _capi.SCAN_QUERY_MAJOR

    def cells(self, window_rows):
        """(cell_of, cell_offsets, cell_order, n_cells) of this index for one window size: `nlsh_build_cells`, once per (index, size)."""
        got = self._cells.get(window_rows)
        if got is None:
            L = _capi.lib()
            dev, nb = self.offsets.device, self.n_buckets
            cell_of = torch.empty((max(nb, 1),), dtype=torch.int32, device=dev)
            cell_offsets = torch.empty((nb + 1,), dtype=torch.int32, device=dev)
            cell_order = torch.empty((max(nb, 1),), dtype=torch.int32, device=dev)
            n_cells = torch.zeros((1,), dtype=torch.int32, device=dev)
            wb = L.nlsh_build_cells_workspace(nb)
            if wb == 0 and nb:
                _capi.check(_capi.E_HIP)
            ws = torch.empty((max(wb, 1),), dtype=torch.uint8, device=dev)
            _capi.check(L.nlsh_build_cells(_capi.ptr(self.offsets), nb, window_rows, _capi.ptr(cell_of), _capi.ptr(cell_offsets),
                                           _capi.ptr(cell_order), _capi.ptr(n_cells), _capi.ptr(ws), wb, _stream(dev)))
            nc = int(n_cells.item())    # like the bucket count of the index build: once per index, not on the query path
            got = self._cells[window_rows] = (cell_of, cell_offsets[:nc + 1], cell_order[:max(nc, 1)], nc)
        return got

    # EXPERIMENT (r06, off by default): consecutive batches walk the schedule order of the cells in OPPOSITE directions behind a common
    # prefix of the `alternate_keep` largest cells.  The order changes speed, never results.  Idea: a batch of a balanced hash touches
    # most of the corpus (GloVe-shaped: 0.36 of 0.47 GB) in the SAME order every time, so what the 256-MiB Infinity Cache holds at the
    # end of batch i -- the rows of the schedule's tail -- is what batch i+1 needs last; walked backwards it needs them first.
    alternate_order = False
    alternate_keep = 0

    def _order_for_this_batch(self, window, cell_order, nc):
        if not self.alternate_order or nc < 2:
            return cell_order
        self._order_phase = getattr(self, "_order_phase", 0) ^ 1
        if not self._order_phase:
            return cell_order
        rev = self.__dict__.setdefault("_cell_order_rev", {}).get((window, self.alternate_keep))
        if rev is None:
            keep = min(max(int(self.alternate_keep), 0), nc)
            rev = torch.cat([cell_order[:keep], cell_order[keep:nc].flip(0)]).contiguous()
            self._cell_order_rev[(window, self.alternate_keep)] = rev
        return rev

    def choose_window(self, Q, P, algo):
        """Row window of the small-bucket packing for one batch shape (tiled schedule only): 64 rows unless forced.
        Measured (r04, tools/scan_bench.py --window 0,64,128,256 --rounds 4, same process and keys, scan kernel ms, one box):
            window            0       64      128     256
            SIFT1M headline   0.2427  0.2400  0.2434  0.2522     (skewed 16-bit hash, 45 k pairs on 5.1 k probed buckets)
            SIFT1M clusters   0.1613  0.1359  0.1377  0.1575     (SURVEY 8(d) generator, 49.7 k pairs on 18.1 k buckets)
            GloVe-1.2M        0.1588  0.1231  0.1244  0.1497     (24-bit cosine, 64.7 k pairs on 36.6 k buckets)
        A 64-row window adds no arithmetic (a lone small bucket already pays a whole 64-row tile and the fat single-tile stages)
        and only merges tasks; wider windows score every row of the window for every query of the task, which the balanced
        workloads earn back in task count up to 128 rows and the skewed headline does not.  One value keeps every shard count on
        the same arithmetic-free choice (results never depend on it)."""
        if algo != _capi.SCAN_BUCKET_TILED or self.n_buckets == 0:
            return 0
        if self.window_rows is not None:
            return self.window_rows
        return 64

    def _tkey(self, algo, Q, P):
        """Key of a batch shape's task table in `_max_tasks`: (schedule, Q, P, row window of the small-bucket packing)."""
        return (algo, Q, P, self.choose_window(Q, P, algo))

    def _estimate_tasks(self, Q, P, seg, algo):
        biased = self._size_biased_bucket()
        if algo == _capi.SCAN_BUCKET_TILED:   # tasks = (256-row segment, <= 16 queries)
            est = Q * min(P, 4) * (1.0 / 16 + biased / 256 / 16) + self.n_buckets
        elif algo == _capi.SCAN_BUCKET_MAJOR:
            est = Q * min(P, 4) * (1.0 / 4 + biased / seg / 4) + self.n_buckets
        else:
            est = Q * (1.0 + min(P, 4) * biased / seg)
        return int(min(max(1.5 * est + 1024, Q + 1024), 2 ** 31 - 8))

    def scan_tensors(self, query_vectors, keys, nkeys, k=10, want_keys=False, check=True, events=None, algo=None):
        """Scan stage on a device key table -> (dist [Q,k], idx [Q,k], ncand [Q], keys64 | None)."""
        if self.metric not in ("l2", "cosine"):
            raise NotImplementedError("fused scan needs metric 'l2' or 'cosine' (use SIFT.distance / Glove.distance)")
        L = _capi.lib()
        q = query_vectors
        if q.device.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "queries must be device-resident; there is no CPU path")
        if q.dtype != torch.float32 or q.stride(1) != 1:
            q = q.float().contiguous()
        Q, d = q.shape
        if d != self.dim:
            raise ValueError(f"query dim {d} != corpus dim {self.dim}")
        dev = q.device
        if keys.shape[1] > _capi.MAX_PROBES:
            return self._scan_sliced(q, keys, nkeys, k, want_keys, check)
        keys = keys.contiguous()
        nkeys = nkeys.contiguous()
        P = keys.shape[1]
        seg = self.seg_rows or 512
        out_dist = torch.empty((Q, k), dtype=torch.float32, device=dev)
        # ids, candidate counts and the two status words share ONE int32 buffer: `query()` brings all three to the host
        # with a single copy (and a single synchronisation) instead of four
        pack = torch.empty((Q * k + Q + 2,), dtype=torch.int32, device=dev)
        out_idx, ncand, status = pack[:Q * k].view(Q, k), pack[Q * k:Q * k + Q], pack[Q * k + Q:]
        out_keys = torch.empty((Q, k), dtype=torch.int64, device=dev) if want_keys else None
        if algo is None:
            algo = self.choose_algo(Q, P)
        # the task table is sized per (schedule, batch shape): a larger batch after a smaller one re-estimates instead of
        # reusing a table that `check=False` callers would silently overflow
        window = self.choose_window(Q, P, algo)
        tkey = self._tkey(algo, Q, P)
        if tkey not in self._max_tasks:
            grown = [v for (a_, q_, p_, w_), v in self._max_tasks.items() if a_ == algo and q_ >= Q and p_ >= P and w_ == window]
            self._max_tasks[tkey] = min(grown) if grown else self._estimate_tasks(Q, P, seg, algo)
        while True:
            max_tasks = self._max_tasks[tkey]
            ws_bytes = L.nlsh_scan_workspace(Q, P, k, max_tasks, self.n_buckets, d)
            stream = _stream(dev)
            # one workspace per stream and schedule family: the bucket-major PLAN phase keeps its per-bucket counters at the
            # head of the workspace ZERO between calls (include/nlsh_hip.h), so nothing else may write there
            wkey = (stream, algo != _capi.SCAN_QUERY_MAJOR)
            ws = self._ws.get(wkey)
            if ws is None or ws.numel() < ws_bytes or ws.device != dev:
                ws = self._ws[wkey] = torch.zeros((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
            try:
                self._scan_launch(q, keys, nkeys, k, algo, max_tasks, out_dist, out_idx, out_keys, ncand, status, ws,
                                  _capi.PHASE_ALL, events, window)
            except _capi.NlshHipError:
                self._ws.pop(wkey, None)    # a call that failed part-way may have left the counters at the head non-zero
                raise
            if not check or Q == 0:
                break
            needed, overflow = status.cpu().tolist()
            if not overflow:
                self._trim_task_table(tkey, needed, max_tasks)
                break
            self._grow_task_table(tkey, needed, overflow)           # segment table too small: grow and repeat
        self.last_status = status
        self.last_algo, self.last_window = algo, window
        self._last_pack, self._last_tkey, self._last_max_tasks = pack, tkey, max_tasks
        return out_dist, out_idx, ncand, out_keys

    def _trim_task_table(self, tkey, needed, max_tasks):
        """The one-shot tiled kernel launches a workgroup per table slot: a table sized for the bucket-per-task estimate and then
        run with shared windows (a fifth of the tasks) is trimmed once the batch shape's real need is known.  Called wherever the
        host learns `needed` anyway -- the checked scan, and `query()`'s own copy of the status words (the default
        `query()` path never trimmed and kept launching mostly empty workgroups)."""
        if 3 * (needed + 1024) < max_tasks:
            self._max_tasks[tkey] = int(needed * 1.25) + 1024

    def _grow_task_table(self, tkey, needed, flag=1):
        """status[1] == 1: the task table was too small; grow it, the caller repeats the call.
        status[1] == 2: the PLAN phase found the per-bucket counters at the head of the workspace non-zero on entry
        (include/nlsh_hip.h, workspace contract): the batch got no tasks; drop the workspaces and fail loudly."""
        if flag == 3:
            self._ws.clear()
            raise _capi.NlshHipError(_capi.E_INVALID, "scan_topk(tiled): the cells hold a shared window wider than one 256-row segment "
                                                      "(cells must come from nlsh_build_cells with window_rows <= 256)")
        if flag == 2:
            self._ws.clear()
            raise _capi.NlshHipError(_capi.E_WORKSPACE, "scan_topk(bucket-major): the pair counters at the head of the workspace "
                                                        "were not zero on entry (workspace contract, include/nlsh_hip.h)")
        self._max_tasks[tkey] = int(needed * 1.25) + 1024

    def _scan_args(self, Q, d, keys, nkeys, k, algo, max_tasks, out_dist, out_idx, out_keys, ncand, status, ws, window=None):
        """The arguments of `nlsh_scan_topk_cells_phase` that do not change between batches of one shape, as plain ints:
        (everything before `queries`, everything between `q_stride`/`Q` and the events).  Callers that launch many
        batches (nlsh_amd/pipeline.py) build them once per buffer set; a call is then one ctypes transition."""
        a = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        metric = (_capi.METRIC_L2_EPS_FOLDED if self.l2_form == "folded" else _capi.METRIC_L2_EPS) if self.metric == "l2" else _capi.METRIC_COSINE
        if window is None:
            window = self.choose_window(Q, keys.shape[1], algo)
        if window and algo == _capi.SCAN_BUCKET_TILED:
            cell_of, cell_offsets, cell_order, nc = self.cells(window)
            sched = (a(self._order_for_this_batch(window, cell_order, nc)), self.n_buckets, a(cell_of), a(cell_offsets), nc)
        else:
            sched = (a(self.bucket_order), self.n_buckets, None, None, 0)
        pre = (a(self.corpus_sorted), self.row_stride, d, a(self.gid), a(self.uniq_keys), a(self.offsets), *sched, a(self.inv_norm))
        post = (Q, a(keys), a(nkeys), keys.shape[1], k, metric, algo, self.seg_rows or 512, a(out_dist), a(out_idx), a(out_keys),
                a(ncand), a(status), a(ws), ws.numel(), max_tasks)
        return pre, post

    def _scan_launch(self, q, keys, nkeys, k, algo, max_tasks, out_dist, out_idx, out_keys, ncand, status, ws, phases, events=None,
                     window=None):
        """One `nlsh_scan_topk_cells_phase` call on the current stream with caller-owned buffers (any subset of the phases)."""
        Q, d = q.shape
        pre, post = self._scan_args(Q, d, keys, nkeys, k, algo, max_tasks, out_dist, out_idx, out_keys, ncand, status, ws, window)
        _capi.check(_capi.lib().nlsh_scan_topk_cells_phase(
            *pre, q.data_ptr(), q.stride(0) if Q else d, *post,
            events[0].cuda_event if events else None, events[1].cuda_event if events else None, _stream(q.device), phases))

    def _scan_sliced(self, q, keys, nkeys, k, want_keys, check):
        """hash_times > 64 (eval.py:148 sweeps n_samples up to 100): the key table is scanned in column slices of
        64.  A row's keys are distinct (encode_hash de-duplicates across all probes) and buckets are disjoint (F9),
        so the slices see disjoint candidates: merging their top-k lists with the (distance, id) comparator and
        adding their candidate counts is exact."""
        from .distributed import merge_topk_device
        parts, counts = [], []
        for c0 in range(0, keys.shape[1], _capi.MAX_PROBES):
            kc = keys[:, c0:c0 + _capi.MAX_PROBES].contiguous()
            nk = (nkeys - c0).clamp(0, kc.shape[1]).to(torch.int32)
            _, _, nc, k64 = self.scan_tensors(q, kc, nk, k=k, want_keys=True, check=check)
            parts.append(k64)
            counts.append(nc)
        packed = torch.cat([torch.stack(parts), torch.stack(counts).long()[:, :, None]], dim=2)
        dist, idx, ncand = merge_topk_device(packed, k)
        keys64 = None
        if want_keys:   # the 64-bit sort keys of the merged lists: monotone(dist) << 32 | id, ~0 for padding
            bits = dist.view(torch.int32).long() & 0xFFFFFFFF
            mono = torch.where(bits >= (1 << 31), (~bits) & 0xFFFFFFFF, bits | (1 << 31))
            keys64 = torch.where(idx < 0, torch.full_like(mono, -1), (mono << 32) | (idx.long() & 0xFFFFFFFF))
        return dist, idx, ncand, keys64

    def query_tensors(self, query_vectors, k=10, hash_times=10, seed=None, want_keys=False, check=True, events=None):
        """Device-resident form of `query`: hashing + scan, nothing copied to the host.
        events = (begin, end) torch.cuda.Event pair (already recorded once) bracketing the scan kernel."""
        return self._batch_tensors(query_vectors, k, hash_times, seed, want_keys=want_keys, check=check, events=events)[:4]

    def _fuses(self, q, hash_times, algo):
        """One `nlsh_query_batch` call (ABI v4: five launches, the bucket lookup in the encode's epilogue) serves a batch when the
        fused kernel may hash it at all (not a BatchNorm encoder in train mode), the key table fits one scan call and the schedule
        is a bucket-major one (the query-major stream's PLAN phase is a single kernel of its own)."""
        needs_train = getattr(self._hashing, "_needs_train_forward", None)
        return (self.metric in ("l2", "cosine") and 1 <= hash_times <= _capi.MAX_PROBES and algo != _capi.SCAN_QUERY_MAJOR
                and self.n_buckets > 0 and q.shape[0] > 0 and not (needs_train is not None and needs_train())
                and hasattr(self._hashing, "encode_args"))

    def _batch_tensors(self, query_vectors, k, hash_times, seed, want_keys=False, check=True, events=None, algo=None, row0=0,
                       n_multi=None, out=None):
        """hash + scan of one batch (or one row range of a batch: `row0`, `n_multi` = rows of the RANGE that are multi-probe, `out` =
        the range's slice of the batch's key table) -> (dist, idx, ncand, keys64 | None, keys, nkeys).  One C-ABI call where the
        fused form applies (`_fuses`), `hash_device` + `scan_tensors` otherwise; same results either way, bit for bit."""
        if hash_times < 1:
            raise ValueError(f"`n` should be positive integer, but got {hash_times}")
        q = self._as_queries(query_vectors)
        Q, d = q.shape
        if algo is None:
            algo = self.choose_algo(Q, hash_times)
        if n_multi is None:
            n_multi = self._n_multi_rows(Q)
        if not self._fuses(q, hash_times, algo):
            if row0 or out is not None:
                keys, nkeys = self._hashing.hash_device(q, n=hash_times, n_multi_rows=n_multi, seed=seed, row0=row0, out=out)
            else:
                keys, nkeys = self.hash_device(q, hash_times=hash_times, seed=seed)
            return self.scan_tensors(q, keys, nkeys, k=k, want_keys=want_keys, check=check, events=events, algo=algo) + (keys, nkeys)
        if d != self.dim:
            raise ValueError(f"query dim {d} != corpus dim {self.dim}")
        import ctypes
        L, h, dev, P = _capi.lib(), self._hashing, q.device, hash_times
        if out is not None:
            keys, nkeys = out
        else:
            keys = torch.empty((Q, P), dtype=torch.int32, device=dev)
            nkeys = torch.empty((Q,), dtype=torch.int32, device=dev)
        out_dist = torch.empty((Q, k), dtype=torch.float32, device=dev)
        pack = torch.empty((Q * k + Q + 2,), dtype=torch.int32, device=dev)    # ids, counts, status: ONE buffer, one copy for `query()`
        out_idx, ncand, status = pack[:Q * k].view(Q, k), pack[Q * k:Q * k + Q], pack[Q * k + Q:]
        out_keys = torch.empty((Q, k), dtype=torch.int64, device=dev) if want_keys else None
        window = self.choose_window(Q, P, algo)
        tkey = self._tkey(algo, Q, P)
        if tkey not in self._max_tasks:
            grown = [v for (a_, q_, p_, w_), v in self._max_tasks.items() if a_ == algo and q_ >= Q and p_ >= P and w_ == window]
            self._max_tasks[tkey] = min(grown) if grown else self._estimate_tasks(Q, P, self.seg_rows or 512, algo)
        if seed is None:
            seed = h.next_seed()
        (n_layers, dims_arr, packed_ptr, act, key_mode, n_probes), _ = h.encode_args(P, keys, nkeys)
        stream = _stream(dev)
        lookup_done = 0
        while True:
            max_tasks = self._max_tasks[tkey]
            ws_bytes = L.nlsh_scan_workspace(Q, P, k, max_tasks, self.n_buckets, d)
            wkey = (stream, True)
            ws = self._ws.get(wkey)
            if ws is None or ws.numel() < ws_bytes or ws.device != dev:
                ws = self._ws[wkey] = torch.zeros((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
            pre, post = self._scan_args(Q, d, keys, nkeys, k, algo, max_tasks, out_dist, out_idx, out_keys, ncand, status, ws, window)
            (corpus, row_stride, d_, gid, uniq, offsets, order, n_buckets, cell_of, cell_offsets, n_cells, inv_norm) = pre
            (Q_, qkeys_p, nkeys_p, P_, k_, metric, algo_, seg, od, oi, ok, nc, st, wsp, wsb, mt) = post
            desc = _capi.StepDesc(
                n_layers=n_layers, act=act, key_mode=key_mode, n_probes=n_probes, dims=ctypes.cast(dims_arr, ctypes.c_void_p), packed=packed_ptr,
                n_multi_rows=int(n_multi), corpus_sorted=corpus, row_stride=row_stride, gid=gid, uniq_keys=uniq, offsets=offsets,
                bucket_order=order, cell_of=cell_of, cell_offsets=cell_offsets, inv_norm=inv_norm, d=d_, n_buckets=n_buckets, n_cells=n_cells,
                k=k_, metric=metric, algo=algo_, seg_rows=seg, hold_done=0, Q=Q_, qkeys=qkeys_p, nkeys=nkeys_p, out_dist=od, out_idx=oi,
                out_keys=ok, out_ncand=nc, status=st, workspace=wsp, workspace_bytes=wsb, max_tasks=mt, front=None, plan=None, mid=None, tail=None)
            try:
                _capi.check(L.nlsh_query_batch(ctypes.byref(desc), ctypes.sizeof(desc), q.data_ptr(), q.stride(0), seed, int(row0), lookup_done,
                                               events[0].cuda_event if events else None, events[1].cuda_event if events else None, stream))
            except _capi.NlshHipError:
                self._ws.pop(wkey, None)    # a call that failed part-way may have left the counters at the head non-zero
                raise
            if not check:
                break
            needed, overflow = status.cpu().tolist()
            if not overflow:
                self._trim_task_table(tkey, needed, max_tasks)
                break
            self._grow_task_table(tkey, needed, overflow)           # task table too small: grow and repeat the scan part on the same keys
            lookup_done = 1
        self.last_status = status
        self.last_algo, self.last_window = algo, window
        self._last_pack, self._last_tkey, self._last_max_tasks = pack, tkey, max_tasks
        return out_dist, out_idx, ncand, out_keys, keys, nkeys

    def _rows_of_key(self, key):
        """Ascending global row ids of one bucket (host list) or [] for an unknown key.  Called once per query with fewer than k
        candidates (the reference's F7 rule) -- ~20 times per 10^4-query batch on the headline workload, so it is a dict lookup and a
        slice of host copies made once per index, not numpy searches (those cost 4 us per call: 0.07 ms of a 0.88-ms `query()`)."""
        key = int(key)
        if self._hashing.key_mode == _capi.KEY_FULL and key >= (1 << 31):
            key -= 1 << 32
        directory = self.__dict__.get("_bucket_dir")
        if directory is None:   # signed int32 key -> (first sorted row, end): one pass over the CSR, once per index
            directory = self._bucket_dir = dict(zip(self._uniq_host.tolist(), zip(self._offs_host[:-1].tolist(), self._offs_host[1:].tolist())))
        span = directory.get(key)
        if span is None:
            return []
        if self._perm_host is None:   # one D2H of the permutation, then every fallback list is a host slice
            self._perm_host = self.gid.cpu().numpy().astype(np.int64)
        return self._perm_host[span[0]:span[1]].tolist()

    # `query()` launches the same row ranges of the same batch shape call after call, and everything it needs on the device is consumed
    # before it returns (the results leave as Python lists).  So the range's buffers -- its slice of the batch's key table, distances,
    # the id / count / status block -- and the descriptor of its `nlsh_query_batch` call are built ONCE per (batch shape, range, k,
    # schedule) and kept: a later call checks that the task table, the workspace and the hasher's weights are still the ones the
    # descriptor names and makes its one C call.  r06: the Python around that call (allocations, the descriptor's forty fields, the
    # weight walk, argument tuples) was 0.08 ms per range = 0.2 ms of a 1.1-ms call (tools/query_host_profile.py).  `query_tensors`
    # hands its tensors to the caller and keeps allocating fresh ones.  Like the scan workspace (one per stream), the kept buffers
    # assume one `query()` at a time per indexer and stream.
    def _range_plan(self, q, keys, nkeys, k, lo, hi, algo, hash_times, n_multi):
        import ctypes
        Q_all, P = keys.shape
        dev, m = q.device, hi - lo
        stream = _stream(dev)
        ckey = (stream, Q_all, lo, hi, P, k, algo, self.l2_form, keys.data_ptr(), q.shape[1])
        plans = self.__dict__.setdefault("_range_plans", {})
        plan = plans.get(ckey)
        tkey = self._tkey(algo, m, P)
        window = self.choose_window(m, P, algo)
        if tkey not in self._max_tasks:
            grown = [v for (a_, q_, p_, w_), v in self._max_tasks.items() if a_ == algo and q_ >= m and p_ >= P and w_ == window]
            self._max_tasks[tkey] = min(grown) if grown else self._estimate_tasks(m, P, self.seg_rows or 512, algo)
        max_tasks = self._max_tasks[tkey]
        wkey = (stream, True)
        ws = self._ws.get(wkey)
        sig = self._hashing._weights_signature()
        if plan is not None and plan["max_tasks"] == max_tasks and plan["ws"] is ws and plan["window"] == window:
            if plan["sig"] != sig:        # a training step between two calls: the descriptor gets the new blob
                plan["packed"] = self._hashing.packed_weights()
                plan["desc"].packed = plan["packed"].data_ptr()
                plan["sig"] = sig
            return plan
        L, d = _capi.lib(), q.shape[1]
        ws_bytes = L.nlsh_scan_workspace(m, P, k, max_tasks, self.n_buckets, d)
        if ws is None or ws.numel() < ws_bytes or ws.device != dev:
            ws = self._ws[wkey] = torch.zeros((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        rk, rn = keys[lo:hi], nkeys[lo:hi]
        out_dist = torch.empty((m, k), dtype=torch.float32, device=dev)
        pack = torch.empty((m * k + m + 2,), dtype=torch.int32, device=dev)
        out_idx, ncand, status = pack[:m * k].view(m, k), pack[m * k:m * k + m], pack[m * k + m:]
        (n_layers, dims_arr, packed_ptr, act, key_mode, n_probes), _ = self._hashing.encode_args(P, rk, rn)
        pre, post = self._scan_args(m, d, rk, rn, k, algo, max_tasks, out_dist, out_idx, None, ncand, status, ws, window)
        (corpus, row_stride, d_, gid, uniq, offsets, order, n_buckets, cell_of, cell_offsets, n_cells, inv_norm) = pre
        (Q_, qkeys_p, nkeys_p, P_, k_, metric, algo_, seg, od, oi, ok, nc, st, wsp, wsb, mt) = post
        desc = _capi.StepDesc(
            n_layers=n_layers, act=act, key_mode=key_mode, n_probes=n_probes, dims=ctypes.cast(dims_arr, ctypes.c_void_p), packed=packed_ptr,
            n_multi_rows=int(min(max(n_multi - lo, 0), m)), corpus_sorted=corpus, row_stride=row_stride, gid=gid, uniq_keys=uniq, offsets=offsets,
            bucket_order=order, cell_of=cell_of, cell_offsets=cell_offsets, inv_norm=inv_norm, d=d_, n_buckets=n_buckets, n_cells=n_cells,
            k=k_, metric=metric, algo=algo_, seg_rows=seg, hold_done=0, Q=Q_, qkeys=qkeys_p, nkeys=nkeys_p, out_dist=od, out_idx=oi,
            out_keys=ok, out_ncand=nc, status=st, workspace=wsp, workspace_bytes=wsb, max_tasks=mt, front=None, plan=None, mid=None, tail=None)
        plan = plans[ckey] = dict(desc=desc, ref=ctypes.byref(desc), size=ctypes.sizeof(desc), dims_arr=dims_arr, packed=self._hashing.packed_weights(), sig=sig,
                                  keys=rk, nkeys=rn, out_dist=out_dist, pack=pack, status=status, ws=ws, wkey=wkey, max_tasks=max_tasks, tkey=tkey,
                                  window=window, stream=stream, order_rotates=bool(self.alternate_order))
        if len(plans) > 64:           # batch shapes come and go (a caller sweeping Q): keep the dictionary bounded
            for old_key in list(plans)[:-32]:
                del plans[old_key]
        return plan

    def _range_tensors(self, q, keys, nkeys, k, lo, hi, algo, fused):
        """One row range of a `query()` batch on the stream: `fused` = (hash_times, seed, rows of the batch that are multi-probe) --
        the range is hashed AND scanned by one `nlsh_query_batch` call into its slice of the batch's key table; None: the table was
        filled by a whole-batch `hash_device` and the range is only scanned."""
        if fused is None:
            self.scan_tensors(q[lo:hi], keys[lo:hi], nkeys[lo:hi], k=k, check=False, algo=algo)
            return
        hash_times, seed, n_multi = fused
        if self.alternate_order:      # (experiment switch: the schedule order changes from call to call, so nothing is kept)
            self._batch_tensors(q[lo:hi], k, hash_times, seed, check=False, algo=algo, row0=lo, n_multi=min(max(n_multi - lo, 0), hi - lo),
                                out=(keys[lo:hi], nkeys[lo:hi]))
            return
        plan = self._range_plan(q, keys, nkeys, k, lo, hi, algo, hash_times, n_multi)
        qr = q[lo:hi]
        try:
            _capi.check(_capi.lib().nlsh_query_batch(plan["ref"], plan["size"], qr.data_ptr(), qr.stride(0), seed, lo, 0, None, None, plan["stream"]))
        except _capi.NlshHipError:
            self._ws.pop(plan["wkey"], None)    # a call that failed part-way may have left the counters at the head non-zero
            self._range_plans.clear()
            raise
        self.last_status = plan["status"]
        self.last_algo, self.last_window = algo, plan["window"]
        self._last_pack, self._last_tkey, self._last_max_tasks = plan["pack"], plan["tkey"], plan["max_tasks"]

    def _host_results(self, q, keys, nkeys, k, fused=None):
        """Scan + device->host copies of (ids, candidate counts, status) and of the key table into pinned buffers + ONE
        stream synchronisation: the only sync of a `query()` call (the key table rides along because the F7 rule needs
        the key sets of the few queries with < k candidates: 400 KB more on the wire is cheaper than a second round of
        device indexing + copies + syncs after the first).  Repeats the scan if the task table overflowed."""
        Q, P = keys.shape
        algo = self.choose_algo(Q, P)
        while True:
            self._range_tensors(q, keys, nkeys, k, 0, Q, algo, fused)
            pack, tkey = self._last_pack, self._last_tkey
            n, nk = pack.numel(), Q * P + Q
            pin = self._pin
            if pin is None or pin.numel() < n + nk:
                pin = self._pin = torch.empty((max(n + nk, 1 << 16),), dtype=torch.int32, pin_memory=True)
            pin[:n].copy_(pack, non_blocking=True)
            if self.compat:
                pin[n:n + Q * P].view(Q, P).copy_(keys, non_blocking=True)
                pin[n + Q * P:n + nk].copy_(nkeys, non_blocking=True)
            self._release_held()                                    # the device is busy now: free what an earlier call left with us
            torch.cuda.current_stream(q.device).synchronize()
            host = pin.numpy()
            needed, overflow = int(host[n - 2]), int(host[n - 1])
            if not overflow or Q == 0:
                self._trim_task_table(tkey, needed, self._last_max_tasks)
                return (host[:Q * k].reshape(Q, k), host[Q * k:Q * k + Q], host[n:n + Q * P].reshape(Q, P), host[n + Q * P:n + nk])
            self._grow_task_table(tkey, needed, overflow)

    # `query()` on a large batch scans it in `query_chunks` row ranges on the stream and converts range c to Python lists while
    # the device scans range c+1: the conversion (0.45 ms per 10^4 queries) is as long as the scan, and a single range leaves
    # the host idle during the scan and the device idle during the conversion.  Results do not depend on the split (the hash
    # runs once over the whole batch; every query's candidates and top-k are its own; the schedule is the whole batch's).
    # None = automatic: 2 ranges, or 1 when `defer_result_release` is on -- the previous call's lists are then freed under the scan,
    # the host is the longer side of the call whatever the split, and a second range only adds its launches (r03, 10^4 queries:
    # 10.4 M queries/s with one range, 9.4 M with two, 7.6 M with three; with the release in the caller's loop 5.6 / 5.9 / 6.4 M).
    query_chunks = None
    _CHUNK_MIN_ROWS = 2048

    def _n_chunks(self):
        if self.query_chunks is not None:
            return int(self.query_chunks)
        return 1 if self.defer_result_release else 2

    def _chunked_results(self, q, keys, nkeys, k, n_chunks, fused=None):
        """Generator over row ranges of the batch: (lo, hi, ids [hi-lo, k], counts [hi-lo], keys, nkeys) as host arrays, each
        yielded as soon as ITS scan and copies are done (one event per range; later ranges keep the device busy meanwhile)."""
        Q, P = keys.shape
        dev = q.device
        algo = self.choose_algo(Q, P)
        stream = torch.cuda.current_stream(dev)
        bounds = [(Q * c // n_chunks, Q * (c + 1) // n_chunks) for c in range(n_chunks)]
        per = max(hi - lo for lo, hi in bounds)
        words = per * k + per + 2 + per * P + per
        if self._pin is None or self._pin.numel() < n_chunks * words:
            self._pin = torch.empty((max(n_chunks * words, 1 << 16),), dtype=torch.int32, pin_memory=True)
        pin = self._pin
        inflight = []

        def launch(c, lo, hi):
            self._range_tensors(q, keys, nkeys, k, lo, hi, algo, fused)
            pack, tkey = self._last_pack, self._last_tkey
            base, n, m = c * words, pack.numel(), hi - lo
            pin[base:base + n].copy_(pack, non_blocking=True)
            if self.compat:
                pin[base + n:base + n + m * P].view(m, P).copy_(keys[lo:hi], non_blocking=True)
                pin[base + n + m * P:base + n + m * P + m].copy_(nkeys[lo:hi], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            return ev, pack, tkey, self._last_max_tasks

        for c, (lo, hi) in enumerate(bounds):
            inflight.append(launch(c, lo, hi))
        self._release_held()                                        # the device is busy now: free what an earlier call left with us
        host = pin.numpy()
        for c, (lo, hi) in enumerate(bounds):
            m = hi - lo
            while True:
                ev, pack, tkey, max_tasks = inflight[c]
                ev.synchronize()
                base, n = c * words, pack.numel()
                needed, overflow = int(host[base + n - 2]), int(host[base + n - 1])
                if not overflow or m == 0:
                    self._trim_task_table(tkey, needed, max_tasks)
                    break
                self._grow_task_table(tkey, needed, overflow)       # task table too small for this range: grow, repeat it
                inflight[c] = launch(c, lo, hi)
            a = host[base:base + n + m * P + m]
            yield (lo, hi, a[:m * k].reshape(m, k), a[m * k:m * k + m], a[n:n + m * P].reshape(m, P), a[n + m * P:n + m * P + m])

    # The fresh result lists (10^4 per batch) land in the collector's youngest generation; the first container allocation after
    # the conversion then runs a generation-0 collection that walks all of them: 0.25-0.32 ms of a 1.55 ms call on the bench box
    # (tools/query_host_profile.py, QGC=1).  Instead: (1) a generation-0 collection BEFORE the conversion, while the young
    # generation holds only what the application made since the last call (microseconds; its cyclic garbage is reclaimed
    # here, as it would have been); (2) the conversion with the collector paused; (3) gc.freeze() + gc.unfreeze(), which moves
    # every tracked object to the oldest generation in O(1) and zeroes the young counters, so the results are first walked by
    # the next FULL collection, like any long-lived object.  Because the counters restart from zero at every call, a tight
    # query loop would never reach the collector's thresholds for the older generations on its own: every
    # `_FULL_COLLECT_EVERY`-th promotion is followed by an explicit full collection (tens of ms in a torch process, so rare).
    # It rewrites the host application's collector generations, so it is OPT-IN (`Indexer.promote_results = True`; bench.py never
    # turns it on): by default the conversion only pauses the collector for its own duration.  Even when on it is skipped when the application holds
    # frozen objects of its own (gc.get_freeze_count() > 0) or has the collector disabled.
    promote_results = False
    # A caller that loops `ids, counts = indexer.query(batch)` frees the previous call's 10^4 lists when it rebinds the names: ~0.3 ms
    # of host time per 10^4 queries that sits between two calls, while the device idles.  OPT-IN (`Indexer.defer_result_release = True`):
    # the indexer keeps a reference to its last TWO results and drops the older one right after the NEXT call has queued its device
    # work -- the deallocation then runs under the scan instead of in front of it.  Results are unchanged and stay the caller's to keep;
    # the cost is that up to two extra result sets (a few MB of Python objects each) stay alive per indexer.  Off by default for the
    # same reason as `promote_results`: a library call should not change object lifetimes unasked.  bench.py's headline `value` is
    # measured with all of these off (r04); `protocol_qps_opt_in` is the same region with this switch and `untracked_results` on.
    defer_result_release = False
    # OPT-IN: hand the inner result lists out untracked by the cyclic collector (csrc/fastlists.c).  A row of ints cannot be part of a
    # cycle and 10^4 tracked young lists per call make the collector's next young pass cost ~0.26 ms -- but CPython never untracks
    # lists, and a caller that later stores something that references the row INTO the row would leak that cycle.  Off by default
    # like the two switches above (r03 untracked whenever the helper was built).
    untracked_results = False
    _FULL_COLLECT_EVERY = 2048
    _promotions = 0

    @classmethod
    def _plain_lists(cls, idx_h, nc_h):
        """Host arrays -> (list of id rows, list of counts): one C-level conversion each for the whole batch."""
        # the state the APPLICATION left the collector in -- not `gc.isenabled()`, which is False whenever `query()` holds its pause
        promote = _collector_pause.application_state() and cls.promote_results and len(nc_h) >= 512 and gc.get_freeze_count() == 0
        if promote:
            gc.collect(0)
        with _collector_pause:   # 10^4 fresh lists would trigger a dozen collections over the whole heap: a third of the conversion
            if _rows_to_lists is not None and idx_h.dtype == np.int32 and idx_h.ndim == 2 and idx_h.flags.c_contiguous:
                out = _rows_to_lists(idx_h, idx_h.shape[0], idx_h.shape[1], bool(cls.untracked_results)), nc_h.tolist()
            else:
                out = idx_h.tolist(), nc_h.tolist()
            if promote:
                gc.freeze()
                gc.unfreeze()
                Indexer._promotions += 1
                if Indexer._promotions % cls._FULL_COLLECT_EVERY == 0:
                    gc.collect()
            return out

    def _to_lists(self, key_sets, idx_h, nc_h, k):
        """Host arrays -> the reference's (List[List[int]], List[int]) (indexer.py:88-95)."""
        results, counts = self._plain_lists(idx_h, nc_h)
        for qi in np.nonzero(nc_h < k)[0].tolist():   # only the short queries need the reference's special case
            if self.compat:
                # indexer.py:89-93 (F7): topk raises -> rows of the LAST key of the set iteration
                order = list(key_sets[qi])
                results[qi] = self._rows_of_key(order[-1]) if order else []
            else:
                results[qi] = [int(v) for v in idx_h[qi] if v >= 0]
        return results, counts

    def _release_held(self):
        """Drop all but the newest result kept by `defer_result_release` (called once the current call's device work is queued)."""
        held = self.__dict__.get("_held")
        if held:
            del held[:-1]

    def _keep(self, result):
        if self.defer_result_release:
            held = self.__dict__.setdefault("_held", [])
            held.append(result)
            # never more than the newest two, whichever path produced them: the generic-metric and sliced (hash_times > 64) paths do
            # not pass through `_release_held`, and without this the list grew by one full result set per call
            del held[:-2]
        elif self.__dict__.get("_held"):
            self._held = []
        return result

    # The cyclic collector is paused for the length of a `query()` call, not only while a row range is converted: with tracked rows
    # (the default) every resume between two ranges is followed by a young-generation pass over the lists made so far, and every tenth
    # young pass by an older-generation one over all the tracked lists alive -- the collector's work per call grew with the number of row
    # ranges (r04, tools/query_modes.py, same box: 1 range 1.47 ms per call, 2 ranges 1.93, 4 ranges 2.66), which is what kept the split
    # from paying.  Paused across the call the fresh lists are walked once, after the call, whatever the split.  No generation is
    # rewritten and nothing is frozen (unlike `promote_results`): the collector simply does not run inside the call, as it already did
    # not inside the conversions.  False restores r04's behaviour (tools/query_modes.py times both).
    # The pause is `_collector_pause`: one re-entrant, lock-protected region counter for the whole process, so concurrent `query()` calls
    # of two threads and the nested pause of `_plain_lists` leave the collector as the FIRST of them found it (r06).
    pause_collector_for_call = True

    def query(self, query_vectors, k=10, hash_times=10, seed=None) -> Tuple[List[List[int]], List[int]]:
        """nlsh/indexer.py:56-96.  `seed` (not in the reference): the Philox seed of the multi-probe draws; None takes the next one
        from the hasher's call counter, like every other hashing call."""
        if not self.pause_collector_for_call:
            return self._keep(self._query(query_vectors, k, hash_times, seed))
        with _collector_pause:
            return self._keep(self._query(query_vectors, k, hash_times, seed))

    def _query(self, query_vectors, k, hash_times, seed):
        if self.metric not in ("l2", "cosine"):
            return self._query_generic(query_vectors, k, hash_times)
        q = self._as_queries(query_vectors)
        Q = q.shape[0]
        if hash_times < 1:
            raise ValueError(f"`n` should be positive integer, but got {hash_times}")
        fused = None
        if self._fuses(q, hash_times, self.choose_algo(Q, hash_times)):
            # every row range is hashed and scanned by ONE call (five launches); the ranges share the batch's key table, seed and the
            # Philox counters of its rows, so the split changes no key
            tab = self.__dict__.setdefault("_key_tables", {})
            tkey_ = (q.device, Q, hash_times)
            if tkey_ not in tab:          # the batch's key table is consumed inside the call (pinned copy for the F7 rule): kept per batch shape
                if len(tab) > 8:
                    tab.clear()
                tab[tkey_] = (torch.empty((Q, hash_times), dtype=torch.int32, device=q.device), torch.empty((Q,), dtype=torch.int32, device=q.device))
            keys, nkeys = tab[tkey_]
            fused = (hash_times, self._hashing.next_seed() if seed is None else seed, self._n_multi_rows(Q))
        else:
            keys, nkeys = self.hash_device(q, hash_times=hash_times, seed=seed)
        if keys.shape[1] > _capi.MAX_PROBES:
            _, idx, ncand, _ = self.scan_tensors(q, keys, nkeys, k=k)
            idx_h, nc_h = idx.cpu().numpy(), ncand.cpu().numpy()
            keys_h, nkeys_h = (keys.cpu().numpy(), nkeys.cpu().numpy()) if self.compat else (None, None)
        elif self._n_chunks() > 1 and q.shape[0] >= self._n_chunks() * self._CHUNK_MIN_ROWS:
            results, counts = [], []
            for lo, hi, idx_h, nc_h, keys_h, nkeys_h in self._chunked_results(q, keys, nkeys, k, self._n_chunks(), fused):
                key_sets = {}
                if self.compat:
                    for qi in np.nonzero(nc_h < k)[0].tolist():
                        key_sets[qi] = host_key_set(keys_h[qi], int(nkeys_h[qi]), self._hashing.key_mode)
                r, c = self._to_lists(key_sets, idx_h, nc_h, k)
                results += r
                counts += c
            return results, counts
        else:
            idx_h, nc_h, keys_h, nkeys_h = self._host_results(q, keys, nkeys, k, fused)
        key_sets = {}
        if self.compat:  # F7 needs the key SET (Python iteration order) of the queries with < k candidates only
            for qi in np.nonzero(nc_h < k)[0].tolist():
                key_sets[qi] = host_key_set(keys_h[qi], int(nkeys_h[qi]), self._hashing.key_mode)
        return self._to_lists(key_sets, idx_h, nc_h, k)

    def _as_queries(self, query_vectors):
        q = query_vectors
        if q.device.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "queries must be device-resident; there is no CPU path")
        if q.dtype != torch.float32 or q.stride(1) != 1:
            q = q.float().contiguous()
        return q

    def query_with_keys(self, query_vectors, key_lists: Sequence[Sequence[int]], k=10):
        """Scan stage on caller-supplied key lists (each in the iteration order the caller saw):
        parity on identical candidate sets, independent of hashing."""
        # a query's keys are a set (nlsh/utils.pyx:27-31): a repeated key probes its bucket once; first-occurrence order kept
        key_lists = [list(dict.fromkeys(int(key) for key in ks)) for ks in key_lists]
        Q = len(key_lists)
        P = max([len(ks) for ks in key_lists] + [1])
        tab = np.zeros((Q, P), dtype=np.int64)
        cnt = np.zeros((Q,), dtype=np.int32)
        for i, ks in enumerate(key_lists):
            cnt[i] = len(ks)
            tab[i, :len(ks)] = list(ks)
        tab = np.where(tab >= (1 << 31), tab - (1 << 32), tab).astype(np.int32)
        dev = query_vectors.device
        keys = torch.as_tensor(tab).to(dev)
        nkeys = torch.as_tensor(cnt).to(dev)
        dist, idx, ncand, _ = self.scan_tensors(query_vectors, keys, nkeys, k=k)
        res, nc = self._to_lists(key_lists, idx.cpu().numpy(), ncand.cpu().numpy(), k)
        return res, nc, dist, idx

    def _global_ids(self, local_rows):
        return self.row_ids[local_rows].long() if self.row_ids is not None else local_rows + self.id_base

    def _query_generic(self, query_vectors, k, hash_times):
        """Arbitrary distance callable: the reference's gather -> callable -> topk per query
        (indexer.py:62-95) on the CSR index, with stock device ops.  Not accelerated."""
        key_sets = self.hash(query_vectors, hash_times=hash_times)
        results, n_candidates = [], []
        corpus = self._candidate_vectors_gpu
        for qi, ks in enumerate(key_sets):
            chunks, last = [], None
            for key in list(ks):
                kk = int(key) - (1 << 32) if int(key) >= (1 << 31) else int(key)
                i = int(np.searchsorted(self._uniq_host, kk))
                last = None                                          # indexer.py:68,92: the last key's rows, [] if it has no bucket
                if i < len(self._uniq_host) and int(self._uniq_host[i]) == kk:
                    last = self.perm[int(self._offs_host[i]):int(self._offs_host[i + 1])].long()
                    chunks.append(last)
            rows = torch.cat(chunks) if chunks else torch.zeros((0,), dtype=torch.int64, device=corpus.device)
            n_candidates.append(int(rows.numel()))
            if rows.numel() >= k:
                dist = self._distance_func(query_vectors[qi], corpus[rows])
                results.append(self._global_ids(rows[dist.topk(k, largest=False)[1]]).tolist())
            elif self.compat:
                results.append([] if last is None else self._global_ids(last).tolist())
            else:
                results.append(self._global_ids(rows).tolist())
        return results, n_candidates

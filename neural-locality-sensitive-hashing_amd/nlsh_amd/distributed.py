"""Corpus sharding across the GPUs of one node; the reference has no
distributed code at all, this is the MI355X-native addition BASELINE.json's north_star asks for.

One process per GPU.  Two partitions of the corpus, same query-time protocol:

* shard="buckets" (default; north_star: "shard the corpus buckets across the 8 GPUs"): every bucket lives
  WHOLE on one rank.  Build: each rank encodes its contiguous row range, the int32 keys are all-gathered
  (4 B/row), every rank derives the same bucket -> owner table (buckets ordered by size, dealt in snake
  order, so row counts and the size-squared scan work are both balanced), and ONE all-to-all moves each row
  (+ its global id) to the owner of its bucket.  A rank then scans 1/G of the (bucket, query-group) tasks at
  full bucket size, so the bucket-major scan keeps its row reuse as G grows.
* shard="rows": rank r keeps its contiguous row range; every bucket is split ~evenly over the ranks.  No
  build-time exchange, perfectly balanced under any skew, but each rank still runs every task on 1/G of the
  rows (measured: the tiled scan only drops 0.42 -> 0.17 ms from G=1 to G=8).

Query time (both): every rank hashes the full, replicated query batch (multi-probe keys are identical on
every rank: the Philox stream is keyed by (seed, global query row, probe) and hard bits are deterministic),
scans its own shard, and the only exchange step is ONE all-gather (RCCL over xGMI when the backend is
"nccl") of the per-rank `[Q, k]` 64-bit (distance,id) keys + `[Q]` candidate counts, followed by the same
(distance, id) merge the single-GPU path uses -> results identical to one GPU.
"""
import os
import warnings
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _capi


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced row range [lo, hi) of rank `rank` (first n_rows % world ranks get +1)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_topk_device(packed_all: torch.Tensor, k: int):
    """packed_all int64 [G, Q, k+1]: per shard and query the k uint64 sort keys (bit patterns) followed by
    the shard's candidate count -> (dist, idx, ncand) via nlsh_merge_topk."""
    if packed_all.device.type != "cuda":
        raise _capi.NlshHipError(_capi.E_INVALID, "merge_topk needs device tensors; there is no CPU path")
    G, Q, stride = packed_all.shape
    assert stride == k + 1
    dev = packed_all.device
    out_dist = torch.empty((Q, k), dtype=torch.float32, device=dev)
    out_idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    out_nc = torch.empty((Q,), dtype=torch.int32, device=dev)
    _capi.check(_capi.lib().nlsh_merge_topk(_capi.ptr(packed_all.contiguous()), stride, G, Q, k, None,
                                            _capi.ptr(out_dist), _capi.ptr(out_idx), _capi.ptr(out_nc),
                                            torch.cuda.current_stream(dev).cuda_stream))
    return out_dist, out_idx, out_nc


def gather_and_merge(local_keys: torch.Tensor, local_ncand: torch.Tensor, k: int, group=None,
                     merge_fn: Callable = merge_topk_device):
    """The exchange step: ONE all-gather of per-rank rows [Q, k+1] (k top-k keys + candidate count),
    then merge.  `merge_fn` is the device merge in production; tests inject a checker to exercise the
    collective + layout on the gloo backend without a GPU.
    """
    world = dist.get_world_size(group)
    Q = local_keys.shape[0]
    packed = torch.cat([local_keys, local_ncand.to(local_keys.dtype)[:, None]], dim=1).contiguous()
    if _staged(packed, group):          # gloo rehearsal on device tensors: the collective runs on host copies
        host = torch.empty((world * Q, k + 1), dtype=packed.dtype)
        dist.all_gather_into_tensor(host, packed.cpu(), group=group)
        return merge_fn(host.to(packed.device).view(world, Q, k + 1), k)
    # concatenation form ([world*Q, ...]): accepted by both the nccl (RCCL) and the gloo backend
    packed_all = torch.empty((world * Q, k + 1), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(packed_all, packed, group=group)
    return merge_fn(packed_all.view(world, Q, k + 1), k)


class TopkExchange:
    """`gather_and_merge` for a caller that runs it every batch (nlsh_amd/pipeline.py's tail stage): the packed send row,
    the gathered table and the merged outputs are allocated once per distinct key table (one per pipeline slot) instead
    of per call, so the host side of the exchange is two small copies, one collective and one ctypes transition.
    The returned tensors are overwritten when the same key table comes round again."""

    def __init__(self, k: int, group=None):
        self.k, self.group, self.world = k, group, dist.get_world_size(group)
        self._bufs = {}

    def __call__(self, keys64: torch.Tensor, ncand: torch.Tensor):
        if keys64.device.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "merge_topk needs device tensors; there is no CPU path")
        k, world = self.k, self.world
        Q, dev = keys64.shape[0], keys64.device
        b = self._bufs.get(keys64.data_ptr())
        if b is None:
            b = self._bufs[keys64.data_ptr()] = (
                torch.empty((Q, k + 1), dtype=torch.int64, device=dev), torch.empty((world * Q, k + 1), dtype=torch.int64, device=dev),
                torch.empty((Q, k), dtype=torch.float32, device=dev), torch.empty((Q, k), dtype=torch.int32, device=dev),
                torch.empty((Q,), dtype=torch.int32, device=dev))
        packed, packed_all, out_dist, out_idx, out_nc = b
        packed[:, :k].copy_(keys64)
        packed[:, k].copy_(ncand)
        if _staged(packed, self.group):     # gloo rehearsal on device tensors
            host = torch.empty(packed_all.shape, dtype=packed_all.dtype)
            dist.all_gather_into_tensor(host, packed.cpu(), group=self.group)
            packed_all.copy_(host)
        else:
            dist.all_gather_into_tensor(packed_all, packed, group=self.group)
        _capi.check(_capi.lib().nlsh_merge_topk(packed_all.data_ptr(), k + 1, world, Q, k, None, out_dist.data_ptr(), out_idx.data_ptr(),
                                                out_nc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return out_dist, out_idx, out_nc


# ----------------------------------------------------------------------------- bucket partition
def assign_buckets(counts: torch.Tensor, world: int) -> torch.Tensor:
    """Owner rank of every bucket, given the bucket sizes in ascending-key order (int64 [nb]).

    Buckets are ordered by (size descending, key ascending) and dealt in snake order 0..G-1,G-1..0: every
    rank gets one bucket of each size class per round, which balances rows (HBM) and sum of size^2 (the scan
    work: a bucket is probed by a number of queries roughly proportional to its size).  Deterministic, so
    every rank computes the same table without communication."""
    nb = counts.shape[0]
    order = torch.argsort(-counts.long(), stable=True)            # ties keep ascending-key order
    pos = torch.arange(nb, device=counts.device)
    rnd, r = pos // world, pos % world
    owner_sorted = torch.where(rnd % 2 == 0, r, world - 1 - r)
    owner = torch.empty((nb,), dtype=torch.int64, device=counts.device)
    owner[order] = owner_sorted
    return owner


def corpus_statistics(counts: torch.Tensor) -> Tuple[float, float]:
    """(size-biased mean bucket size, rows) of the whole corpus: what `Indexer.choose_algo` decides on."""
    c = counts.double()
    n = float(c.sum().item())
    return (float((c * c).sum().item() / n) if n else 0.0), n


def plan_bucket_shards(keys_all: torch.Tensor, world: int):
    """All corpus keys (int32 [N]) -> (owner rank of every ROW int64 [N], (size-biased bucket, N))."""
    uniq, inverse, counts = torch.unique(keys_all, return_inverse=True, return_counts=True)
    return assign_buckets(counts, world)[inverse], corpus_statistics(counts)


def _exchange_mode(group) -> str:
    """"alltoall" | "allgather": how the bucket partition moves rows, decided BEFORE any data collective and agreed
    by all ranks (a rank that fell back on its own would enter a different collective sequence than its peers and
    hang them).  NLSH_SHARD_EXCHANGE=allgather forces the gather form; a backend whose `all_to_all_single` with uneven
    splits fails a zero-byte-safe probe also selects it.  Failures inside the data collectives (out of memory, a lost
    peer) are NOT caught: they propagate."""
    want = 0 if os.environ.get("NLSH_SHARD_EXCHANGE", "alltoall") == "allgather" else 1
    if want:
        try:                                                        # capability probe: one element to the next rank only
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            send_counts = [1 if r == (rank + 1) % world else 0 for r in range(world)]
            recv_counts = [1 if r == (rank - 1) % world else 0 for r in range(world)]
            out = torch.empty((1,), dtype=torch.int32, device=dev)
            dist.all_to_all_single(out, torch.full((1,), rank, dtype=torch.int32, device=dev), recv_counts, send_counts, group=group)
            want = 1 if int(out.item()) == (rank - 1) % world else 0
        except (RuntimeError, NotImplementedError) as e:
            warnings.warn(f"all_to_all_single with uneven splits unavailable on this backend ({e}); using the all-gather form")
            want = 0
    flag = torch.tensor([want], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device())
                        if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)       # every rank takes the same branch
    return "alltoall" if int(flag.item()) else "allgather"


def _staged(t: torch.Tensor, group) -> bool:
    # gloo rehearsals on device tensors go through the host; RCCL ("nccl") works on device memory directly
    return t.device.type == "cuda" and dist.get_backend(group) == "gloo"


def _all_gather_rows(t: torch.Tensor, group) -> torch.Tensor:
    """All-gather of per-rank [n_r, ...] tensors with different n_r -> [sum n_r, ...] in rank order."""
    world = dist.get_world_size(group)
    dev = t.device
    work = t.cpu() if _staged(t, group) else t
    n = torch.tensor([work.shape[0]], dtype=torch.int64, device=work.device)
    sizes = torch.empty((world,), dtype=torch.int64, device=work.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = sizes.cpu().tolist()
    n_max = max(sizes)
    padded = torch.zeros((n_max,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    padded[:work.shape[0]] = work
    out = torch.empty((world * n_max,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = torch.cat([out[r * n_max:r * n_max + sizes[r]] for r in range(world)])
    return out.to(dev)


def _all_to_all_rows(send: torch.Tensor, send_counts, recv_counts, group) -> torch.Tensor:
    dev = send.device
    work = send.cpu() if _staged(send, group) else send.contiguous()
    out = torch.empty((int(sum(recv_counts)),) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    dist.all_to_all_single(out, work, list(recv_counts), list(send_counts), group=group)
    return out.to(dev)


def exchange_rows_by_bucket(local_rows: torch.Tensor, local_keys: torch.Tensor, id_base: int, group=None):
    """Build-time exchange of the bucket partition.  In: this rank's contiguous row range `[n_r, d]`, its
    bucket keys (int32 [n_r]) and the global id of its first row.  Out: (rows this rank OWNS [m_r, d], their
    global ids int32 [m_r] ascending within a bucket, (size-biased bucket, N) of the whole corpus, the bucket keys of
    ALL corpus rows in global row order -- the all-gather every rank already paid for).
    Collectives: one all-gather of the keys, one of the [G] send counts, two all-to-alls (rows, ids)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    keys_all = _all_gather_rows(local_keys.view(-1), group)
    lo = _all_gather_rows(torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_keys.device), group)
    start = int(lo[:rank].sum().item())
    owner_all, stats = plan_bucket_shards(keys_all, world)
    n_local = local_rows.shape[0]
    ids_local = torch.arange(n_local, device=local_rows.device, dtype=torch.int32) + int(id_base)
    if _exchange_mode(group) == "alltoall":
        dest = owner_all[start:start + n_local]
        order = torch.argsort(dest, stable=True)                   # keeps ascending row order per destination
        send_counts = torch.bincount(dest, minlength=world)
        counts_all = _all_gather_rows(send_counts.view(1, world), group)  # [G, G]: row r = what rank r sends
        recv_counts = counts_all[:, rank].cpu().tolist()
        send_counts = send_counts.cpu().tolist()
        rows = _all_to_all_rows(local_rows[order], send_counts, recv_counts, group)
        ids = _all_to_all_rows(ids_local[order], send_counts, recv_counts, group)
        return rows, ids, stats, keys_all
    # fallback (NLSH_SHARD_EXCHANGE=allgather): every rank gathers all rows and keeps the ones it owns -- G times the
    # traffic of the all-to-all, the same rows in the same (ascending global id) order
    mine = torch.nonzero(owner_all == rank).view(-1)
    rows = _all_gather_rows(local_rows, group)[mine]
    ids = _all_gather_rows(ids_local, group)[mine]
    return rows, ids, stats, keys_all


def global_statistics(local_keys: torch.Tensor, group=None):
    """Whole-corpus schedule statistics for the row partition (one all-gather of the keys) + the gathered keys."""
    keys_all = _all_gather_rows(local_keys.view(-1), group)
    return corpus_statistics(torch.unique(keys_all, return_counts=True)[1]), keys_all


class ShardedIndexer:
    """`Indexer` over this rank's part of the corpus + the all-gather/merge exchange step.

    `local_corpus_gpu` is this rank's contiguous row range of the corpus and `id_base` the global id of its
    first row.  shard="buckets" re-partitions it at build time (see module docstring); shard="rows" keeps it."""

    def __init__(self, hashing, local_corpus_gpu, distance_func, id_base: int, group=None, shard: str = "buckets", **kw):
        from .indexer import Indexer
        if shard not in ("buckets", "rows"):
            raise ValueError("shard must be 'buckets' or 'rows'")
        self.group = group
        self.shard = shard
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._directory = None      # host CSR of ALL buckets (global row ids), built lazily from keys_all for the F7 rule
        if world == 1:
            self.local = Indexer(hashing, local_corpus_gpu, distance_func, id_base=id_base, **kw)
            self.keys_all = None
            return
        keys, _ = hashing.hash_device(local_corpus_gpu, n=1)       # same launch the index build uses (indexer.py:36-38)
        if shard == "buckets":
            rows, ids, stats, self.keys_all = exchange_rows_by_bucket(local_corpus_gpu, keys.view(-1), id_base, group)
            self.local = Indexer(hashing, rows, distance_func, row_ids=ids, schedule_stats=stats, **kw)
        else:
            stats, self.keys_all = global_statistics(keys.view(-1), group)
            self.local = Indexer(hashing, local_corpus_gpu, distance_func, id_base=id_base, schedule_stats=stats, **kw)

    def rows_of_key(self, key):
        """Ascending GLOBAL row ids of one bucket of the WHOLE corpus ([] for an unknown key), answered locally on every
        rank: the build-time all-gather left every rank the bucket key of every corpus row (4 B per row), so the F7 rule
        of the rare short queries (nlsh/indexer.py:91-93: rows of the last key) needs no collective at query time."""
        import numpy as np
        if self.keys_all is None:
            return self.local._rows_of_key(key)
        if self._directory is None:
            kh = self.keys_all.cpu().numpy()
            order = np.argsort(kh, kind="stable")
            uniq, first = np.unique(kh[order], return_index=True)
            self._directory = (uniq, np.append(first, len(kh)), order)
        uniq, offs, order = self._directory
        key = int(key)
        if self.local._hashing.key_mode == _capi.KEY_FULL and key >= (1 << 31):
            key -= 1 << 32
        i = int(np.searchsorted(uniq, key))
        if i >= len(uniq) or int(uniq[i]) != key:
            return []
        return order[offs[i]:offs[i + 1]].tolist()

    def query_tensors(self, query_vectors, k=10, hash_times=10, seed=0, check=True, events=None):
        """`seed` must be the same on every rank (default 0; pass a per-step value to vary probes)."""
        _, _, ncand, keys64 = self.local.query_tensors(query_vectors, k=k, hash_times=hash_times, seed=seed,
                                                       want_keys=True, check=check, events=events)
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return merge_topk_device(torch.cat([keys64, ncand.long()[:, None]], dim=1)[None], k)
        return gather_and_merge(keys64, ncand, k, self.group)

    def query(self, query_vectors, k=10, hash_times=10, seed=None, own_slice=False):
        """`Indexer.query` (nlsh/indexer.py:56-96) over the sharded corpus: the reference's `(List[List[int]],
        List[int])`, identical to the single-GPU answer.  hash -> local scan -> all-gather + merge -> one
        device->host copy.  The multi-probe seed comes from the hasher's call counter (identical on all ranks as long
        as they make the same calls) unless given.

        own_slice=False: every rank returns the lists of ALL queries.  own_slice=True: rank r returns the lists of ITS
        contiguous slice `shard_range(Q, r, G)` of the batch only -- every rank still scans all queries over its shard
        (the merged device tensors are complete everywhere), but the host-side work of the reference's return type
        (device->host copy + 10^5 Python ints per 10^4 queries) is divided over the ranks like the scan is."""
        import numpy as np
        local = self.local
        if seed is None:
            seed = local._hashing.next_seed()
        keys, nkeys = local.hash_device(query_vectors, hash_times=hash_times, seed=seed)
        _, _, ncand, keys64 = local.scan_tensors(query_vectors, keys, nkeys, k=k, want_keys=True)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if world > 1 else 0
        if world == 1:
            _, idx, nc = merge_topk_device(torch.cat([keys64, ncand.long()[:, None]], dim=1)[None], k)
        else:
            _, idx, nc = gather_and_merge(keys64, ncand, k, self.group)
        Q = idx.shape[0]
        lo, hi = shard_range(Q, rank, world) if own_slice else (0, Q)
        local._release_held()      # the device (and the collective) are busy: free what an earlier call left with us (opt-in, Indexer.defer_result_release)
        idx_h, nc_h = idx[lo:hi].cpu().numpy(), nc[lo:hi].cpu().numpy()
        results, counts = local._plain_lists(idx_h, nc_h)                        # short lists are replaced below
        # queries with fewer than k candidates (rare)
        short = np.nonzero(nc_h < k)[0]
        if short.size and local.compat:
            # F7 (indexer.py:91-93): rows of the LAST key of the set iteration -- a bucket of the WHOLE corpus, looked up in
            # the replicated bucket directory (no collective, whichever rank owns the bucket)
            from .hashings import host_key_set
            sel = torch.as_tensor(short + lo, device=keys.device)
            keys_h, nkeys_h = keys[sel].cpu().numpy(), nkeys[sel].cpu().numpy()
            for j, row, cnt in zip(short.tolist(), keys_h, nkeys_h):
                ks = host_key_set(row, int(cnt), local._hashing.key_mode)
                results[j] = self.rows_of_key(list(ks)[-1]) if ks else []
        elif short.size:
            for j in short.tolist():
                results[j] = [int(v) for v in idx_h[j] if v >= 0]
        return local._keep((results, counts))

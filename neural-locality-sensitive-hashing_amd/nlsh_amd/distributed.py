"""Corpus sharding across the GPUs of one node; the reference has no
distributed code at all, this is the MI355X-native addition BASELINE.json's north_star asks for.

One process per GPU.  Rank r owns a contiguous row range of the corpus (so every bucket is split
~evenly over ranks and each rank scans C_q/G candidates per query), builds its own CSR with GLOBAL
row ids (`id_base`), and answers the full (replicated) query batch over its shard.  The only
exchange step is ONE all-gather (RCCL over xGMI when the backend is "nccl") of the per-rank
`[Q, k]` 64-bit (distance,id) keys + `[Q]` candidate counts, followed by the same
(distance, id) merge the single-GPU path uses -> results identical to one GPU.
Multi-probe keys are identical on every rank because the Philox stream is keyed by
(seed, global query row, probe) and the hard bits are deterministic.
"""
from typing import Callable, Tuple

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
    # concatenation form ([world*Q, ...]): accepted by both the nccl (RCCL) and the gloo backend
    packed_all = torch.empty((world * Q, k + 1), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(packed_all, packed, group=group)
    return merge_fn(packed_all.view(world, Q, k + 1), k)


class ShardedIndexer:
    """`Indexer` over this rank's corpus shard + the all-gather/merge exchange step."""

    def __init__(self, hashing, local_corpus_gpu, distance_func, id_base: int, group=None, **kw):
        from .indexer import Indexer
        self.group = group
        kw.setdefault("stats_scale", dist.get_world_size(group) if dist.is_initialized() else 1)
        self.local = Indexer(hashing, local_corpus_gpu, distance_func, id_base=id_base, **kw)

    def query_tensors(self, query_vectors, k=10, hash_times=10, seed=0, check=True, events=None):
        """`seed` must be the same on every rank (default 0; pass a per-step value to vary probes)."""
        _, _, ncand, keys64 = self.local.query_tensors(query_vectors, k=k, hash_times=hash_times, seed=seed,
                                                       want_keys=True, check=check, events=events)
        return gather_and_merge(keys64, ncand, k, self.group)

"""N>1 path on CPU: world_size-2 gloo run of the exchange step (shard ranges, all-gather layout,
merge-of-shards == single-shard result).  The local scan and the merge are played by the oracle
here (no GPU); on the GPU box the same functions run with the HIP kernels (test_gpu_parity)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mono(dist32):
    u = dist32.astype(np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000)


def make_keys(dist32, idx32):
    key = (_mono(dist32) << np.uint64(32)) | idx32.astype(np.uint32).astype(np.uint64)
    key = np.where(idx32 < 0, np.uint64(0xFFFFFFFFFFFFFFFF), key)
    return key.view(np.int64)


def merge_numpy(packed_all, k):
    G, Q, _ = packed_all.shape
    keys_all, nc_all = packed_all[:, :, :k].contiguous(), packed_all[:, :, k]
    flat = keys_all.numpy().view(np.uint64).transpose(1, 0, 2).reshape(Q, G * k)
    best = np.sort(flat, axis=1)[:, :k]
    idx = np.where(best == np.uint64(0xFFFFFFFFFFFFFFFF), -1, (best & np.uint64(0xFFFFFFFF)).astype(np.int64)).astype(np.int32)
    return best, idx, nc_all.sum(0)


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nlsh_amd import synth
    from nlsh_amd.distributed import gather_and_merge, shard_range
    from oracle import oracle
    N, Q, d, H, k = 3001, 40, 128, 6, 10
    corpus = synth.sift_like(N, d, seed=1)
    queries = synth.sift_like(Q, d, seed=2)
    Ws, bs = synth.make_weights([d, 64, H], seed=3)
    lo, hi = shard_range(N, rank, world)
    ox = oracle.OracleIndexer(Ws, bs, corpus[lo:hi])
    keys, nk = ox.hash_arrays(queries, hash_times=4)
    od, oi, nc = oracle.query_batch(ox.corpus, ox.perm, ox.uniq_keys, ox.offsets, queries, keys, nk, k)
    oi = np.where(oi >= 0, oi + lo, -1).astype(np.int32)                 # global row ids (id_base = lo)
    local_keys = torch.from_numpy(make_keys(od, oi))
    best, idx, nc_sum = gather_and_merge(local_keys, torch.from_numpy(nc.astype(np.int32)), k, merge_fn=merge_numpy)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), idx=idx, nc=nc_sum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_shard_range_partitions_rows():
    from nlsh_amd.distributed import shard_range
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_merge_equals_single_shard(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "merged.npz")
    from nlsh_amd import synth
    from oracle import oracle
    N, Q, d, H, k = 3001, 40, 128, 6, 10
    corpus, queries = synth.sift_like(N, d, seed=1), synth.sift_like(Q, d, seed=2)
    Ws, bs = synth.make_weights([d, 64, H], seed=3)
    ox = oracle.OracleIndexer(Ws, bs, corpus)
    keys, nk = ox.hash_arrays(queries, hash_times=4)
    od, oi, nc = oracle.query_batch(ox.corpus, ox.perm, ox.uniq_keys, ox.offsets, queries, keys, nk, k)
    assert np.array_equal(got["nc"], nc)
    assert np.array_equal(got["idx"], oi)          # same (distance, id) comparator -> identical lists


# ----------------------------------------------------------------------------- bucket partition (build-time exchange)
def _exchange_worker(rank, world, port, out_dir, mode="alltoall"):
    for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NLSH_SHARD_EXCHANGE"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nlsh_amd.distributed import exchange_rows_by_bucket, shard_range
    keys_all, rows_all = _exchange_case()
    lo, hi = shard_range(len(keys_all), rank, world)
    rows, ids, stats, gathered = exchange_rows_by_bucket(torch.from_numpy(rows_all[lo:hi]), torch.from_numpy(keys_all[lo:hi]), lo)
    assert np.array_equal(gathered.numpy(), keys_all)                                 # every rank holds the key of every corpus row
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=rows.numpy(), ids=ids.numpy(), stats=np.asarray(stats))
    dist.barrier()
    dist.destroy_process_group()


def _exchange_case():
    rng = np.random.default_rng(11)
    N = 5003
    keys = np.minimum(rng.geometric(0.02, size=N), 400).astype(np.int32) - 200       # skewed sizes, negative keys too
    rows = np.stack([np.arange(N, dtype=np.float32), keys.astype(np.float32), rng.standard_normal(N).astype(np.float32)], 1)
    return keys, rows


@pytest.mark.parametrize("world,mode", [(2, "alltoall"), (3, "alltoall"), (2, "allgather"), (8, "alltoall"), (8, "allgather")])
def test_gloo_bucket_exchange_moves_every_bucket_whole(tmp_path, world, mode):
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    from nlsh_amd.distributed import assign_buckets, corpus_statistics
    keys_all, rows_all = _exchange_case()
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    uniq, counts = np.unique(keys_all, return_counts=True)
    owner = assign_buckets(torch.from_numpy(counts), world).numpy()
    want_stats = corpus_statistics(torch.from_numpy(counts))
    all_ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(all_ids), np.arange(len(keys_all)))                 # every row exactly once
    for r, p in enumerate(parts):
        ids = p["ids"]
        assert np.array_equal(p["rows"], rows_all[ids])                               # rows travel with their ids
        assert np.all(np.diff(ids) > 0)                                               # ascending global id (stable)
        assert np.all(owner[np.searchsorted(uniq, keys_all[ids])] == r)               # only buckets this rank owns
        assert tuple(p["stats"]) == want_stats                                        # identical schedule statistics
    sizes = np.array([len(p["ids"]) for p in parts])
    assert sizes.max() <= 1.1 * sizes.mean()                                          # balanced rows

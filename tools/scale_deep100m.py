#!/usr/bin/env python3
"""Scale run for BASELINE.json configs[4]: Deep100M-shaped corpus (N x 96-d fp32, unit norm, HBM-resident: 38.4 GB +
the bucket-sorted copy), 32-bit learned hash (full-width keys), 100k queries -- on ONE GPU, or sharded over the ranks
of a torchrun launch (each rank generates its own row range on its device, the bucket partition's all-to-all moves the
rows to their owners, batches go through the three-stage pipeline with the all-gather in the tail stage).

    python tools/scale_deep100m.py [--rows 100000000 --queries 100000 --hash-size 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/scale_deep100m.py

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


CHUNK = 1 << 22


def _manifold_params(dev, d, latent_dim=10, n_clusters=256):
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    A1 = torch.randn((latent_dim, 96), generator=g, device=dev)
    b1 = torch.rand((96,), generator=g, device=dev) * 2 - 1
    A2 = torch.randn((96, d), generator=g, device=dev) / 9.8
    cen = torch.randn((n_clusters, latent_dim), generator=g, device=dev)
    return A1, b1, A2, cen


def deep_chunk(params, chunk_index, rows, d, seed, dev, spread=0.5):
    """Rows [chunk_index*CHUNK, +rows) of the corpus: seeded per chunk, so any rank can (re)generate any range."""
    A1, b1, A2, cen = params
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 1000003 + chunk_index)
    which = torch.randint(0, cen.shape[0], (rows,), generator=g, device=dev)
    z = cen[which] + spread * torch.randn((rows, cen.shape[1]), generator=g, device=dev)
    x = torch.tanh(z @ A1 + b1) @ A2 + 0.02 * torch.randn((rows, d), generator=g, device=dev)
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)


def deep_manifold_device(lo, hi, d, seed, dev, params):
    """Unit-norm rows [lo, hi) on a low-dimensional manifold, generated on the device chunk by chunk."""
    out = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    for ci in range(lo // CHUNK, (hi + CHUNK - 1) // CHUNK):
        c0, c1 = ci * CHUNK, (ci + 1) * CHUNK
        a, b = max(lo, c0), min(hi, c1)
        full = deep_chunk(params, ci, CHUNK, d, seed, dev)      # the whole chunk, so the stream does not depend on (lo, hi)
        out[a - lo:b - lo] = full[a - c0:b - c0]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", "--n", dest="n", type=int, default=100_000_000)
    ap.add_argument("--queries", "--q", dest="q", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=96)
    ap.add_argument("--hash-size", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--hash-times", type=int, default=10)
    ap.add_argument("--train-rows", type=int, default=1_000_000)
    ap.add_argument("--train-steps", type=int, default=5000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--shard", default="buckets", choices=["buckets", "rows"])
    ap.add_argument("--save-hash", default="", help="write the trained hash as a portable .npz checkpoint (rank 0)")
    ap.add_argument("--load-hash", default="", help="skip training: load a checkpoint written by --save-hash")
    ap.add_argument("--pipeline", default="on", choices=["on", "off"],
                    help="on: batches through the three-stage pipeline (what a serving loop runs; queries/s is quoted on it).  off: every kernel of a "
                         "step back to back on one stream -- the scan kernel alone on the chip, which is the region a roofline figure and a "
                         "rocprofv3 --pmc pass should be taken in")
    ap.add_argument("--l2-form", default="exact", choices=["exact", "folded"])
    ap.add_argument("--window", type=int, default=None, help="row window of the tiled schedule's small-bucket packing (default: the facade's)")
    ap.add_argument("--pmc-summary", default="", help="tools/pmc_summary.py output of a rocprofv3 --pmc pass over THIS command (same arguments, --pipeline off): "
                                                      "fills roofline.traffic / valu counters of the line (bytes = FETCH_SIZE KB x 1024 x 2, MI355X_MICROARCH.md)")
    args = ap.parse_args()
    import torch.distributed as dist
    from nlsh_amd import training
    from nlsh_amd.data import SIFT
    from nlsh_amd.distributed import ShardedIndexer, gather_and_merge, shard_range
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    from nlsh_amd.metrics import calculate_recall
    from nlsh_amd.pipeline import QueryPipeline

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = 0 if os.environ.get("NLSH_BENCH_SAME_DEVICE") else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("NLSH_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    torch.manual_seed(0)
    N, Q, d, H, k, P = args.n, args.q, args.dim, args.hash_size, args.k, args.hash_times
    params = _manifold_params(dev, d)
    lo, hi = shard_range(N, rank, world)
    t0 = time.time()
    corpus = deep_manifold_device(lo, hi, d, 1234, dev, params)          # this rank's row range only
    queries = deep_manifold_device(0, Q, d, 4321, dev, params)           # replicated
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    if rank == 0:
        print(f"[deep] rank 0 generated rows [{lo}, {hi}) x {d} on device in {gen_s:.1f}s ({corpus.numel() * 4 / 1e9:.1f} GB)", flush=True)

    # learned hash on a sample of rank 0's rows, broadcast (unit-norm rows: L2 ranking == cosine ranking; Deep1B uses L2)
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, None, compat=H <= 16)
    t0 = time.time()
    if args.load_hash:
        from nlsh_amd import io
        Ws_, bs_ = io.load_hasher_weights(args.load_hash)
        training.load_weights(hashing, {**{f"W{i}": w for i, w in enumerate(Ws_)}, **{f"b{i}": b for i, b in enumerate(bs_)}})
    elif rank == 0:
        sample = corpus[:: max(1, (hi - lo) // args.train_rows)][: args.train_rows].contiguous()
        knn = training.self_knn(sample, 10)
        training.fit_triplet(hashing, sample, knn, n_steps=args.train_steps, margin=1.0, log=lambda s: None)
        del knn, sample
    if world > 1:
        for p_ in hashing.parameters():
            dist.broadcast(p_.data, src=0)
    hashing.train_mode(False)
    train_s = time.time() - t0
    if args.save_hash and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(args.save_hash)), exist_ok=True)
        np.savez_compressed(args.save_hash, meta=json.dumps(dict(vars(args), generator="tools/scale_deep100m.py deep_manifold_device seed 1234")),
                            **training.export_weights(hashing))

    torch.cuda.synchronize()
    t0 = time.time()
    sharded = ShardedIndexer(hashing, corpus, SIFT.distance, id_base=lo, shard=args.shard, compat=H <= 16, l2_form=args.l2_form,
                             window_rows=args.window)
    indexer = sharded.local
    if indexer._candidate_vectors_gpu is not corpus:
        del corpus                                                       # the bucket partition owns its own copy of the rows
    torch.cuda.synchronize()
    build_s = time.time() - t0
    stats = indexer.bucket_stats()
    if rank == 0:
        print(f"[deep] rank 0 index built in {build_s:.2f}s ({indexer._candidate_vectors_gpu.shape[0]} rows): {stats}", flush=True)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record(); b.record()
    pipe = None
    if args.pipeline == "on":
        pipe = QueryPipeline(indexer, queries, k=k, hash_times=P, depth=3, want_keys=True,
                             exchange=(lambda k64, nc: gather_and_merge(k64, nc, k)) if world > 1 else None)

    def step(i, events=None):
        if pipe is not None:
            return pipe.submit(queries, seed=1, events=events)
        d_, i_, n_, k64 = indexer.query_tensors(queries, k=k, hash_times=P, seed=1, want_keys=world > 1, check=False, events=events)
        return gather_and_merge(k64, n_, k) if world > 1 else (d_, i_, n_)

    def drain():
        if pipe is not None:
            pipe.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if pipe is None:
        indexer.query_tensors(queries, k=k, hash_times=P, seed=1, check=True)   # sizes the task table (the pipeline's constructor does it too)
    step(-1)
    drain()
    t0 = time.perf_counter()
    graph_slots = pipe is not None and getattr(pipe, "graph", False)
    for i in range(args.steps):
        out = step(i, events=None if graph_slots else ev[i])    # graph slots replay a batch only without scan events, and overlap whole batches
    drain()
    elapsed = time.perf_counter() - t0
    if graph_slots:    # the scan kernel alone on the chip: a sequential region of its own (same batch, same seed), outside the timed one
        for i in range(args.steps):
            indexer.query_tensors(queries, k=k, hash_times=P, seed=1, want_keys=world > 1, check=False, events=ev[i])
        drain()
    if pipe is not None:
        assert not pipe.overflowed()
    else:
        assert int(indexer.last_status.cpu()[1]) == 0, "task table overflow inside the timed region"
    algo = pipe.algo if pipe is not None else int(indexer.last_algo)
    dist_, idx, nc = out[:3]
    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    sum_c_local = int(indexer.query_tensors(queries, k=k, hash_times=P, seed=1, check=False)[2].long().sum().item())
    sum_c = int(nc.long().sum().item())
    # rows of the buckets the batch probes at all, each once (bench.py: the bucket-major schedules' algorithmic bytes)
    kk, nn = indexer.hash_device(queries, hash_times=P, seed=1)
    pos = torch.searchsorted(indexer.uniq_keys, kk.clamp(min=int(indexer.uniq_keys[0]), max=int(indexer.uniq_keys[-1]))).clamp(max=indexer.n_buckets - 1)
    hit = (indexer.uniq_keys[pos] == kk) & (torch.arange(kk.shape[1], device=dev)[None, :] < nn[:, None])
    unique_bytes = 4.0 * d * float((indexer.offsets[1:] - indexer.offsets[:-1]).long()[torch.unique(pos[hit])].sum().item())
    del kk, nn, pos, hit
    traffic, valu_insts, salu_insts = None, None, None
    if args.pmc_summary:
        pm = json.load(open(args.pmc_summary))
        kern_name = {0: "scan_kernel", 1: "bscan2_kernel", 2: "bscan3_kernel"}[algo]
        name = next(kn for kn in pm if kern_name in kn)
        traffic, valu_insts, salu_insts = pm[name]["FETCH_SIZE"] * 1024 * 2, pm[name].get("SQ_INSTS_VALU"), pm[name].get("SQ_INSTS_SALU")

    if rank == 0:
        # properties (size independent): ascending, distances of the returned ids (rows regenerated from their ids)
        ok = idx >= 0
        assert bool((dist_[:, 1:] >= dist_[:, :-1]).all()), "not ascending"
        sel = idx[:256].clamp(min=0).long().reshape(-1)
        rows = torch.empty((sel.numel(), d), dtype=torch.float32, device=dev)
        for ci in torch.unique(sel // CHUNK).tolist():
            m = (sel // CHUNK) == ci
            rows[m] = deep_chunk(params, ci, CHUNK, d, 1234, dev)[sel[m] - ci * CHUNK]
        ref = torch.nn.functional.pairwise_distance(queries[:256, None, :].expand(-1, k, -1).reshape(-1, d), rows).reshape(-1, k)
        assert bool(((dist_[:256] - ref).abs() <= 2e-5 * ref.clamp(min=1.0))[ok[:256]].all()), "distance mismatch"
        # recall on a sample (chunked brute force over all N rows, regenerated chunk by chunk)
        R = min(args.recall_queries, Q)
        recall = None
        if R > 0:
            qs = queries[:R]
            best_d = torch.full((R, k), float("inf"), device=dev)
            best_i = torch.full((R, k), -1, dtype=torch.int64, device=dev)
            qq = (qs * qs).sum(1)[:, None]
            for ci in range((N + CHUNK - 1) // CHUNK):
                c = deep_chunk(params, ci, CHUNK, d, 1234, dev)[: min(CHUNK, N - ci * CHUNK)]
                dd = qq - 2.0 * (qs @ c.T) + (c * c).sum(1)[None, :]
                td, ti = dd.topk(k, dim=1, largest=False)
                cat_d, cat_i = torch.cat([best_d, td], 1), torch.cat([best_i, ti + ci * CHUNK], 1)
                o = cat_d.topk(k, dim=1, largest=False).indices
                best_d, best_i = cat_d.gather(1, o), cat_i.gather(1, o)
            recall = float(np.mean(calculate_recall(list(best_i.cpu().numpy()), [r[r >= 0].tolist() for r in idx[:R].cpu().numpy()])))
        print(json.dumps({
            "workload": f"configs[4]: Deep100M-shaped, N={N} d={d} Q={Q} H={H} k={k} hash_times={P}, {world} rank(s), corpus {args.shard} sharded",
            "corpus_gb": N * d * 4 / 1e9, "generate_s": gen_s, "train_s": train_s, "index_build_s": build_s,
            "rank0_rows": int(indexer._candidate_vectors_gpu.shape[0]), "rank0_buckets": stats["n_indexes"], "bucket_mean": stats["mean"],
            "bucket_max": stats["max"], "queries_per_s": Q * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
            "step_driver": ("graph slots: every batch one captured hipGraph on its slot's own stream" if graph_slots else "three-stage pipeline") if pipe is not None else "sequential: every kernel of a step back to back on one stream",
            "scan_kernel": {0: "query-major", 1: "bucket-major", 2: "bucket-major LDS-tiled"}[algo],
            "rank0_scan_ms": scan_ms, "mean_candidates_per_query": sum_c / Q,
            "rank0_algorithmic_GBps": 4.0 * d * sum_c_local / (scan_ms * 1e-3) / 1e9,
            # same accounting as bench.py: the bucket-major schedules share a fetched row between the queries of a group, so the roof is the
            # larger of pair flops / fp32 vector peak (3 flop per element: (q-c), +eps, fma; 157.3 TF) and distinct-candidate-row bytes /
            # HBM peak; the query-major schedule re-reads rows per query (HBM on 4 d sum C_q, 8 TB/s)
            "roofline": ({"bound": "hbm", "achieved": 4.0 * d * sum_c_local / (scan_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                          "frac": 4.0 * d * sum_c_local / (scan_ms * 1e-3) / 1e9 / 8000.0} if algo == 0 else
                         ({"bound": "valu", "achieved": (3.0 if args.l2_form == "exact" else 3.0) * d * sum_c_local / (scan_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                           "frac": 3.0 * d * sum_c_local / (scan_ms * 1e-3) / 1e12 / 157.3}
                          if 1.5 * 3.0 * d * sum_c_local / 1e12 / 157.3 >= unique_bytes / 1e9 / 8000.0 else
                          {"bound": "hbm", "achieved": unique_bytes / (scan_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": unique_bytes / (scan_ms * 1e-3) / 1e9 / 8000.0})) | {
                "kernel": {0: "scan_kernel", 1: "bscan2_kernel", 2: "bscan3_kernel"}[algo], "avg_launch_ms": scan_ms,
                "sum_candidates_per_launch": sum_c_local, "algorithmic_bytes_per_launch": 4.0 * d * sum_c_local,
                "distinct_candidate_row_bytes_per_launch": unique_bytes, "valu_frac": 3.0 * d * sum_c_local / (scan_ms * 1e-3) / 1e12 / 157.3,
                "hbm_frac_of_distinct_candidate_rows": unique_bytes / (scan_ms * 1e-3) / 1e9 / 8000.0,
                "traffic": traffic, "traffic_source": (f"rocprofv3 --pmc pass over this command ({os.path.basename(args.pmc_summary)}): FETCH_SIZE KB x 1024 x 2" if traffic else None),
                "valu_wave_instructions_per_launch": valu_insts, "salu_wave_instructions_per_launch": salu_insts,
                "valu_issue_frac": (valu_insts * 2.0 / (1024 * scan_ms * 1e-3 * 2.4e9)) if valu_insts else None,
                "l2_form": args.l2_form, "window_rows": int(indexer.last_window),
                "note": ("scan kernel timed with HIP events in a SEQUENTIAL region: every kernel of a step back to back on one stream, the scan alone on the chip"
                         if pipe is None else
                         ("queries/s through graph slots; the scan kernel timed with HIP events in a sequential region of its own after the timed one (two batches' scans may share the chip inside it)"
                          if graph_slots else
                          "scan kernel timed with HIP events inside the three-stage pipeline (shares the chip with the neighbouring batches' small kernels)"))},
            "recall_at_10_on_sample": recall, "recall_sample": R, "properties": "ascending, distances vs torch on regenerated rows: ok",
            "rank0_peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: queries/sec + recall@10 of the nlsh query-time hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over the whole query batch (BASELINE.json configs[1]:
SIFT1M-shaped, 1M x 128-d corpus, 10k queries, 16-bit hash, k=10, hash_times=10):
encode_hash (MLP on fp32 MFMA + bits + multi-probe keys) -> plan -> scan_topk -> merge, all
device-resident (inputs in HBM before the timed region, results left in HBM).  At N=1 every kernel of a step runs
back to back on one stream (each kernel alone on the chip: undisturbed roofline timings).  At N>1 the K timed steps run
as a three-stage pipeline over three HIP streams (nlsh_amd/pipeline.py): encode + plan of batch i+1 and merge +
all-gather of batch i-1 beside the scan of batch i -- every kernel of every step runs inside the timed region, scan
kernels never overlap each other, results are bit-identical to sequential calls (`--pipeline on|off` forces either).
N>1: corpus buckets sharded over the ranks (whole buckets per rank, one build-time all-to-all; `--shard rows` keeps
contiguous row ranges instead), every rank answers all queries over its shard, one all-gather (RCCL) of
the per-rank top-k + merge per step ("strong" scaling: total work fixed).

Prints ONE JSON line (rank 0).  `roofline` is the scan kernel's algorithmic bytes (4*d*sum C_q,
SURVEY.md §8(d)) over its HIP-event-measured duration; `cpu_baseline` is the CPU oracle timed on
this box's host cores on a bounded sample of the same queries and candidate sets.
There is no dataset or reference checkpoint offline: data is seeded synthetic (`synth.sift_manifold`)
and the hash is the one our minimal trainer learned on it (checkpoints/, see config.hash).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (155 TF measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sift1m", choices=["sift1m", "glove"],
                    help="sift1m = BASELINE.json configs[1] (headline); glove = configs[2] (1,183,514 x 100-d, cosine, 24-bit)")
    ap.add_argument("--n", type=int, default=int(os.environ.get("NLSH_BENCH_N", 0)))
    ap.add_argument("--q", type=int, default=int(os.environ.get("NLSH_BENCH_Q", 10_000)))
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--hash-size", type=int, default=0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--hash-times", type=int, default=10)
    ap.add_argument("--seg-rows", type=int, default=0)
    ap.add_argument("--shard", default="buckets", choices=["buckets", "rows"],
                    help="N>1 partition of the corpus: whole buckets per rank (default) or contiguous row ranges")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="on: three-stage pipeline over three HIP streams, encode+plan of batch i+1 and merge(+all-gather) of "
                         "batch i-1 on high-priority streams beside the scan of batch i.  off: every kernel of a step back "
                         "to back on one stream (each kernel alone on the chip).  auto: on for N>1 (hides the collective and "
                         "the per-batch fixed kernels), off for N=1 (worth +2..10 %% there, but the scan kernel's roofline "
                         "timing is then taken while it shares the chip: --also-other reports the other mode too)")
    ap.add_argument("--also-other", action="store_true", help="additionally time the K steps in the other mode (reported as `other_mode`)")
    ap.add_argument("--algo", default=None, choices=["query", "bucket", "tiled"], help="force a scan schedule (default: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--random-init", action="store_true", help="ignore the learned-hash checkpoint")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus={args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if os.environ.get("NLSH_BENCH_SAME_DEVICE"):  # rehearsal of the N>1 path on a one-GPU box (gloo, all ranks on cuda:0)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("NLSH_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from nlsh_amd import _capi, synth
    from nlsh_amd.data import Glove, SIFT, brute_force_topk
    from nlsh_amd.distributed import ShardedIndexer, TopkExchange, gather_and_merge, shard_range
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.metrics import calculate_recall
    _capi.lib()  # fail loudly if the HIP library is missing

    wl = {"sift1m": dict(N=1_000_000, d=128, H=16, metric="l2", ckpt="sift1m_manifold_h16.npz", cfg="configs[1]: SIFT1M-shaped "
                         "(synthetic SIFT-like integers on a 6-d latent manifold, synth.sift_manifold, standardised)"),
          "glove": dict(N=1_183_514, d=100, H=24, metric="cosine", ckpt="glove_manifold_h24.npz", cfg="configs[2]: GloVe-1.2M-shaped "
                        "(synthetic 100-d embeddings on an 8-d latent manifold, synth.glove_manifold, cosine)")}[args.workload]
    N, d, H = args.n or wl["N"], args.dim or wl["d"], args.hash_size or wl["H"]
    args.hash_size_eff = H
    Q, k, P, metric = args.q, args.k, args.hash_times, wl["metric"]
    t_setup = time.time()
    if args.workload == "sift1m":
        corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
        queries_h, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
    else:
        corpus_h = synth.glove_manifold(N, d, seed=synth.SEED_DATA)
        queries_h = synth.glove_manifold(Q, d, seed=synth.SEED_QUERY)
    ckpt = os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", wl["ckpt"])
    if d == wl["d"] and H == wl["H"] and os.path.exists(ckpt) and not args.random_init:
        arrs = np.load(ckpt)
        Ws, bs = [arrs[f"W{i}"] for i in range(3)], [arrs[f"b{i}"] for i in range(3)]
        hash_desc = (f"learned: triplet loss (margin 1.0, random negatives, 5000 Adam steps, tools/train_hash.py) on this "
                     f"synthetic corpus, {d}->256->256->{H}, checkpoints/{wl['ckpt']}")
    else:
        Ws, bs = synth.make_weights([d, 256, 256, H], seed=synth.SEED_WEIGHTS)
        hash_desc = f"seeded nn.Linear-default init {d}->256->256->{H} (random-init hash)"
    # keys wrap to int16 like the reference (nlsh/utils.pyx:7-15) up to 16 bits; wider hashes use the full code
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, None, compat=H <= 16)
    lin = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for m, W, b in zip(lin, Ws, bs):
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
    hashing.train_mode(False)

    lo, hi = shard_range(N, rank, world)
    shard = torch.from_numpy(corpus_h[lo:hi]).to(dev)
    queries = torch.from_numpy(queries_h).to(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    distance = SIFT.distance if metric == "l2" else Glove.distance
    if world > 1:   # build-time exchange (all-gather of keys + all-to-all of rows for --shard buckets) is inside build_s
        indexer = ShardedIndexer(hashing, shard, distance, id_base=lo, shard=args.shard, compat=H <= 16,
                                 seg_rows=args.seg_rows, algo=args.algo).local
    else:
        indexer = Indexer(hashing, shard, distance, compat=H <= 16, seg_rows=args.seg_rows, algo=args.algo)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    stats = indexer.bucket_stats()

    steps, warmup = args.steps, args.warmup
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    ev_x = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev + ev_x:  # instantiate the hipEvent handles
        a.record(); b.record()

    def step(i, check=False, events=None):
        seed = 1000 + i  # identical on every rank -> identical multi-probe keys
        dist_, idx_, nc_, k64 = indexer.query_tensors(queries, k=k, hash_times=P, seed=seed, want_keys=world > 1,
                                                       check=check, events=events)
        if world > 1:
            if events is not None:
                ev_x[i][0].record()
            dist_, idx_, nc_ = gather_and_merge(k64, nc_, k)
            if events is not None:
                ev_x[i][1].record()
        return dist_, idx_, nc_

    step(-1, check=True)  # sizes the segment table (may retry once); untimed
    pipe, cur = None, [None]

    xchg = TopkExchange(k) if world > 1 else None

    def exchange(k64, nc):   # tail stage of the pipeline on a sharded index: all-gather of per-shard top-k + merge
        i = cur[0]
        if i is not None:
            ev_x[i][0].record()
        out_ = xchg(k64, nc)
        if i is not None:
            ev_x[i][1].record()
        return out_

    use_pipeline = args.pipeline == "on" or (args.pipeline == "auto" and world > 1)
    if use_pipeline:
        from nlsh_amd.pipeline import QueryPipeline
        pipe = QueryPipeline(indexer, queries, k=k, hash_times=P, depth=3, exchange=exchange if world > 1 else None)

    def run_step(i, events=None):
        if pipe is None:
            return step(i, events=events)
        cur[0] = i if events is not None else None
        return pipe.submit(queries, seed=1000 + i, events=events)[:3]

    for i in range(warmup):
        run_step(-2 - i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = run_step(i, events=ev[i])
    if pipe is not None:
        pipe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    last_status = pipe.last_slot.status if pipe is not None else indexer.last_status
    n_tasks, overflow = (int(v) for v in last_status.cpu())
    assert overflow == 0 and not (pipe is not None and pipe.overflowed()), "segment table overflow inside the timed region"
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    other_mode = None
    if args.also_other and world == 1:
        from nlsh_amd.pipeline import QueryPipeline
        alt = QueryPipeline(indexer, queries, k=k, hash_times=P, depth=3) if pipe is None else None
        for i in range(warmup):
            alt.submit(queries, seed=i) if alt is not None else step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            alt.submit(queries, seed=1000 + i) if alt is not None else step(i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        other_mode = {"mode": "pipeline" if alt is not None else "sequential", "value": Q * steps / el2, "unit": "queries/s",
                      "ms_per_step": 1e3 * el2 / steps}

    scan_ms = [a.elapsed_time(b) for a, b in ev]
    scan_avg_ms = float(np.mean(scan_ms))
    dist_, idx_, nc_ = out
    local_nc = indexer.query_tensors(queries, k=k, hash_times=P, seed=1000 + steps - 1)[2]
    sum_c_local = int(local_nc.long().sum().item())
    algo_bytes = 4.0 * d * sum_c_local
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9
    # the API-level call (reference return type: Python lists; includes D2H + F7 handling)
    api_qps = None
    if world == 1:
        for _ in range(3):                              # warm-ups (first use of the host-side torch ops), SURVEY.md §8(d)
            indexer.query(queries, k=k, hash_times=P)
        times = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids_api, nc_api = indexer.query(queries, k=k, hash_times=P)
            times.append(time.perf_counter() - t0)
        api_qps = Q / float(np.median(times))           # wall time of the reference-typed call incl. hashing, D2H, list building

    # encoder (MFMA) utilisation on this rank's corpus rows, and the box's measured HBM copy rate next to the spec peak
    enc = None
    if world == 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hashing.hash_device(shard, n=1)
        e0.record()
        for _ in range(5):
            hashing.hash_device(shard, n=1)
        e1.record()
        torch.cuda.synchronize()
        enc_ms = e0.elapsed_time(e1) / 5
        flops = 2.0 * (d * 256 + 256 * 256 + 256 * H) * shard.shape[0]
        enc = {"kernel": "encode_hash_kernel (fp32 MFMA 32x32x2, fused bits + keys)", "rows": int(shard.shape[0]), "ms": enc_ms,
               "tflops": flops / (enc_ms * 1e-3) / 1e12, "peak_tflops": MFMA_F32_PEAK_TFLOPS,
               "mfma_util": flops / (enc_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}
        src = torch.empty((1 << 28,), dtype=torch.float32, device=dev)      # 1 GiB read + 1 GiB written per copy
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 2.0 * src.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    # HBM traffic per launch: PMC numbers cannot be collected from inside the process; they come from
    # the committed rocprofv3 --pmc passes of this round (profiles/traffic_r01.json) when the workload matches.
    traffic, valu_insts = None, None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
        w = tr["workload"]
        if (w["N"], w["d"], w["Q"], w["H"], w["hash_times"]) == (N, d, Q, H, P) and world == 1 and "learned" in hash_desc \
                and args.workload == "sift1m":
            traffic = tr["traffic_bytes_per_launch"].get(str(indexer.last_algo))
            valu_insts = tr.get("valu_wave_instructions_per_launch", {}).get(str(indexer.last_algo))
    except (OSError, KeyError, ValueError):
        pass

    result = None
    if rank == 0:
        gt = brute_force_topk(queries, torch.from_numpy(corpus_h).to(dev), k, metric).cpu().numpy()
        idx_h = idx_.cpu().numpy()
        recall = float(np.mean(calculate_recall(list(gt), [r[r >= 0].tolist() for r in idx_h])))
        mean_c = float(nc_.float().mean().item())
        value = Q * steps / elapsed
        result = {
            "metric": "queries/sec + recall@10, SIFT1M 128-d 16-bit hash, 1/2/4/8 GPU" if args.workload == "sift1m" else
                      "queries/sec + recall@10 (GloVe-1.2M 100-d cosine 24-bit: BASELINE.json configs[2], not the headline)",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "recall_at_10": recall,
            "config": {"workload": f"{wl['cfg']}, N={N} d={d} Q={Q} H={H} k={k} hash_times={P}",
                       "hash": hash_desc,
                       "parallelism": (f"corpus {args.shard} sharded x{world} ({indexer._candidate_vectors_gpu.shape[0]} rows on rank 0), "
                                       f"all-gather of per-shard top-k + merge: {float(np.mean([a.elapsed_time(b) for a, b in ev_x])):.4f} ms/step"
                                       ) if world > 1 else "single GPU",
                       "n_buckets": stats["n_indexes"], "bucket_mean": stats["mean"], "bucket_median": stats["median"],
                       "bucket_max": stats["max"], "mean_candidates_per_query": mean_c,
                       "index_build_s": build_s, "api_list_qps": api_qps},
            "roofline": {"bound": "hbm", "kernel": {0: "scan_kernel (query-major)", 1: "bscan2_kernel (bucket-major, 8 queries in registers)", 2: "bscan3_kernel (bucket-major, LDS-tiled)"}[indexer.last_algo] + " " + metric, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "note": "achieved = ALGORITHMIC bytes (4*d*sum C_q) / kernel time; the bucket-major schedules fetch each row once per query GROUP, so frac > 1 means HBM traffic (see traffic, bytes/launch from PMC) is far below the algorithmic bytes and the kernel is fp32-VALU-bound",
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_avg_ms,
                         "sum_candidates_per_launch": sum_c_local, "tasks_per_launch": n_tasks},
        }
        result["config"]["step_driver"] = ("three-stage pipeline over three HIP streams (nlsh_amd/pipeline.py)" if pipe is not None
                                           else "sequential: every kernel of a step back to back on one stream")
        if other_mode is not None:
            result["other_mode"] = other_mode
        if valu_insts:   # the tiled kernel is fp32-VALU-bound: its issue-rate utilisation beside the HBM figure the contract asks for
            result["roofline"]["valu_wave_instructions_per_launch"] = valu_insts
            result["roofline"]["valu_issue_frac"] = valu_insts * 2.0 / (1024 * scan_avg_ms * 1e-3 * 2.4e9)
        if enc is not None:
            result["encoder"] = enc
            result["roofline"]["hbm_copy_measured_GBps"] = copy_gbps
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, corpus_h, queries_h, Ws, bs, indexer, hashing, queries, steps, metric)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, corpus_h, queries_h, Ws, bs, indexer, hashing, queries, steps, metric):
    """CPU oracle (oracle/: C + OpenMP scan, BLAS forward) on a bounded sample of the same workload:
    same queries, same multi-probe keys (so identical candidate sets), same k."""
    from oracle import oracle
    k, P = args.k, args.hash_times
    keys, nkeys = indexer.hash_device(queries, hash_times=P, seed=1000 + steps - 1)
    kh, nh = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
    ck = indexer.corpus_keys.cpu().numpy().astype(np.int64)
    if args.hash_size_eff > 16:  # full-width keys travel as int32 bit patterns
        ck, kh = ck & 0xFFFFFFFF, kh & 0xFFFFFFFF
    perm, uniq, offs = oracle.build_csr(ck)
    threads = oracle.num_threads()

    def run(sample):
        t0 = time.perf_counter()
        z = oracle.mlp_forward_blas(queries_h[:sample], Ws, bs)
        _, p01 = oracle.head_probs(z)
        oracle.row_keys(p01, P, "ref_int16" if args.hash_size_eff <= 16 else "full", seed=1000 + steps - 1, n_multi_rows=(sample // 4096) * 4096)
        t1 = time.perf_counter()
        oracle.query_batch(corpus_h, perm, uniq, offs, queries_h[:sample], kh[:sample], nh[:sample], k, metric)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    Q = len(queries_h)
    probe = min(Q, 256)
    th, ts = run(probe)
    per_q = (th + ts) / probe
    sample = int(min(Q, max(probe, args.cpu_seconds / max(per_q, 1e-9))))
    passes, th, ts = 0, 0.0, 0.0
    while passes == 0 or (th + ts < args.cpu_seconds and passes < 64):     # bounded: ~cpu_seconds of CPU work
        a, b = run(sample)
        th, ts, passes = th + a, ts + b, passes + 1
    out = {"value": sample * passes / (th + ts), "unit": "queries/s", "cores": threads, "kind": "port",
           "sample": f"first {sample} of {Q} queries x {passes} passes, same keys/candidate sets as the GPU run; "
                     f"hash {th:.3f}s (numpy BLAS) + scan {ts:.3f}s (C, OpenMP x{threads})",
           "host_cpu_count": os.cpu_count(), "host_cpu_model": _cpu_model()}
    # the same restatement on ONE thread (the reference's per-query loop is single-threaded Python over torch ops)
    oracle.set_num_threads(1)
    try:
        one = max(16, min(sample, int(3.0 / max(per_q * threads * 0.5, 1e-9))))
        t0 = time.perf_counter()
        oracle.query_batch(corpus_h, perm, uniq, offs, queries_h[:one], kh[:one], nh[:one], k, metric)
        out["scan_only_1_thread_qps"] = one / (time.perf_counter() - t0)
    finally:
        oracle.set_num_threads(threads)
    return out


if __name__ == "__main__":
    main()

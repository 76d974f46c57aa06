#!/usr/bin/env python3
"""Kernel-iteration harness for the candidate scan: builds the headline index once (SIFT1M-shaped, learned 16-bit hash),
then times the scan phase alone (HIP events around the scan kernel, as bench.py does) and the whole device-resident
step, and checks the result against the query-major schedule (ids equal up to fp32 near-ties, candidate counts exact).

    python tools/scan_bench.py [--algo tiled] [--iters 30] [--workload sift1m|clusters|glove] [--window 0,64,128,256]
    NLSH_HIP_LIB=/path/to/other/build.so python tools/scan_bench.py --tag other      # A/B against another build
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

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import Glove, SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="tiled")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--workload", default="sift1m")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--tag", default="")
    ap.add_argument("--window", default="", help="comma-separated row windows of the small-bucket packing to time one after the other on the same index and keys (0 = one task list per bucket; empty = the facade's choice); one JSON line each")
    ap.add_argument("--rounds", type=int, default=1, help="repeat the --window list this many times, interleaved (A B C A B C ...), and end with a summary line: per window the median and the minimum of the rounds' mean scan times -- same-process A/B that clock drift cannot order")
    ap.add_argument("--metric", default="", choices=["", "l2", "cosine"], help="override the workload's metric (sift1m with cosine = the same buckets and candidates through the cosine bodies)")
    ap.add_argument("--l2-form", default="exact", choices=["exact", "folded"], help="folded: the opt-in 2-op L2 block (NLSH_METRIC_L2_EPS_FOLDED)")
    ap.add_argument("--stress", type=int, default=0, help="repeat the scan N more times and count results that differ from the first one in any bit (the tiled schedule's results do not depend on timing: any difference is a race)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--row-align", type=int, default=4, help="row stride of the sorted corpus copy = ceil(d / A) * A floats (32: every row starts on a 128-byte line)")
    ap.add_argument("--zeros", action="store_true", help="zero the grouped corpus and the queries after the index and the keys exist: same tasks and instruction stream, operands that toggle nothing (DVFS probe; pair with -DNLSH_ABLATE=5, ties change the selection)")
    ap.add_argument("--riffle", default="", choices=["", "half", "third", "twothirds", "rev", "rand"], help="experiment: re-deal the static size order of the cells (proportional merge of its heavy prefix holding this share of the rows with the rest; rev / rand as sensitivity checks)")
    ap.add_argument("--order", default="", choices=["", "pairs", "work", "density"], help="experiment: schedule order of the buckets recomputed on the host from THIS batch's keys (pairs: by (query, probe) pairs hitting the bucket; work: pairs x rows; density: full 16-query groups first, then by pairs), in place of the static size order")
    ap.add_argument("--tight", type=float, default=0.0, help="task table (= grid of the one-shot scan kernel) set to TIGHT x the tasks the batch needs (experiment; 0: the facade's estimate)")
    args = ap.parse_args()
    Q = args.queries
    if args.workload in ("sift1m", "clusters"):
        N, d = args.rows or 1_000_000, 128
        gen = synth.sift_manifold if args.workload == "sift1m" else synth.sift_like   # clusters: SURVEY 8(d)'s own generator
        corpus_h, mean, std = synth.standardise(gen(N, d, seed=synth.SEED_DATA))
        queries_h, _, _ = synth.standardise(gen(Q, d, seed=synth.SEED_QUERY), mean, std)
        ck, dist_fn, compat = ("sift1m_manifold_h16.npz" if args.workload == "sift1m" else "sift1m_clusters_h16.npz"), SIFT.distance, True
    else:
        N, d = args.rows or 1_183_514, 100
        corpus_h, queries_h = synth.glove_manifold(N, d, seed=synth.SEED_DATA), synth.glove_manifold(Q, d, seed=synth.SEED_QUERY)
        ck, dist_fn, compat = "glove_manifold_h24.npz", Glove.distance, False
    Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", ck))
    hashing = io.hashing_from_weights(Ws, bs, compat=compat)
    cg, qg = torch.from_numpy(corpus_h).cuda(), torch.from_numpy(queries_h).cuda()
    if args.metric:
        dist_fn = SIFT.distance if args.metric == "l2" else Glove.distance
    ix = Indexer(hashing, cg, dist_fn, compat=compat, algo=args.algo, l2_form=args.l2_form, row_align=args.row_align)
    keys, nkeys = ix.hash_device(qg, hash_times=10, seed=7)
    if args.order:
        uk = ix.uniq_keys.cpu().numpy().astype(np.int64)
        kh, nh = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
        valid = np.arange(kh.shape[1])[None, :] < nh[:, None]
        flat = kh[valid]
        pos = np.searchsorted(uk, flat)
        pos[pos >= len(uk)] = 0
        hit = uk[pos] == flat
        m = np.bincount(pos[hit], minlength=len(uk)).astype(np.int64)
        size = ix.bucket_sizes.astype(np.int64)
        key = {"pairs": m * 100000 + size, "work": m * size, "density": np.minimum(m, 16) * 10**9 + m * 4096 + size}[args.order]
        order = np.argsort(-key, kind="stable").astype(np.int32)
        ix.bucket_order = torch.from_numpy(order).cuda()
    if args.riffle:
        # experiment: the static size order re-dealt so that heavy (many rows: compute-leaning) and light (few rows: memory-leaning) cells
        # are in flight together over the whole launch instead of one after the other
        ix.scan_tensors(qg, keys, nkeys, k=10)
        w = ix.last_window
        if w:
            cell_of, cell_offsets, cell_order, nc = ix._cells[w]
            co = cell_offsets.cpu().numpy().astype(np.int64)
            sizes_by_id, order = co[1:] - co[:-1], cell_order.cpu().numpy()
        else:
            sizes_by_id, order = ix.bucket_sizes.astype(np.int64), ix.bucket_order.cpu().numpy()
        sz = sizes_by_id[order]
        n = len(order)
        if args.riffle == "rev":
            new = order[::-1].copy()
        elif args.riffle == "rand":
            new = order[np.random.default_rng(1).permutation(n)]
        else:
            frac = {"half": 0.5, "third": 1.0 / 3, "twothirds": 2.0 / 3}[args.riffle]
            cut = int(np.searchsorted(np.cumsum(sz), frac * sz.sum()))
            H, Lt = order[:cut], order[cut:]
            # proportional merge: position i of the merged list comes from H when its share of H consumed lags
            pos = np.concatenate([(np.arange(len(H)) + 0.5) / max(len(H), 1), (np.arange(len(Lt)) + 0.5) / max(len(Lt), 1)])
            new = np.concatenate([H, Lt])[np.argsort(pos, kind="stable")]
        new_t = torch.from_numpy(np.ascontiguousarray(new).astype(np.int32)).cuda()
        if w:
            ix._cells[w] = (cell_of, cell_offsets, new_t, nc)
        else:
            ix.bucket_order = new_t
    windows = ([None] if not args.window else [int(w) for w in args.window.split(",")]) * max(1, args.rounds)
    ref_out, first, by_window = None, None, {}
    # rows of the buckets this batch probes at all (each counted once): what a schedule that fetches every needed row exactly once reads
    uk_ = ix.uniq_keys.cpu().numpy().astype(np.int64)
    kh_, nh_ = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
    flat_ = kh_[np.arange(kh_.shape[1])[None, :] < nh_[:, None]]
    pos_ = np.searchsorted(uk_, flat_); pos_[pos_ >= len(uk_)] = 0
    probed_ = np.unique(pos_[uk_[pos_] == flat_])
    unique_rows, n_pairs = int(ix.bucket_sizes[probed_].sum()), int((uk_[pos_] == flat_).sum())
    for window in windows:
        ix.window_rows = window
        ix.scan_tensors(qg, keys, nkeys, k=10)                      # sizes the task table
        if args.tight:
            ix._max_tasks[ix._last_tkey] = int(args.tight * int(ix.last_status.cpu()[0])) + 1
        if args.zeros:
            ix.corpus_sorted.zero_(); qg.zero_()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
        for a, b in evs:
            a.record(); b.record()
        for _ in range(3):
            ix.scan_tensors(qg, keys, nkeys, k=10, check=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.iters):
            out = ix.scan_tensors(qg, keys, nkeys, k=10, check=False, events=evs[i])
        torch.cuda.synchronize()
        scan_call_ms = 1e3 * (time.perf_counter() - t0) / args.iters
        kern = np.array([a.elapsed_time(b) for a, b in evs])
        t0 = time.perf_counter()
        for i in range(args.iters):
            ix.query_tensors(qg, k=10, hash_times=10, seed=7, check=False)
        torch.cuda.synchronize()
        step_ms = 1e3 * (time.perf_counter() - t0) / args.iters
        dist, idx, nc, _ = out
        rec = {"tag": args.tag, "workload": args.workload, "l2_form": args.l2_form, "lib": os.path.basename(os.environ.get("NLSH_HIP_LIB", "default")), "algo": args.algo,
               "scan_kernel_ms": float(kern.mean()), "scan_kernel_ms_min": float(kern.min()), "scan_phases_ms": scan_call_ms,
               "step_ms": step_ms, "tasks": int(ix.last_status.cpu()[0]), "max_tasks": ix._last_max_tasks, "sum_candidates": int(nc.long().sum())}
        if os.environ.get("SCAN_BENCH_GROUPS"):   # pairs by the size of the query group they sit in (host recomputation from the keys)
            uk = ix.uniq_keys.cpu().numpy().astype(np.int64)
            kh, nh = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
            flat = kh[np.arange(kh.shape[1])[None, :] < nh[:, None]]
            pos = np.searchsorted(uk, flat); pos[pos >= len(uk)] = 0
            m = np.bincount(pos[uk[pos] == flat], minlength=len(uk)).astype(np.int64)
            size = ix.bucket_sizes.astype(np.int64)
            rem = m % 16
            tot = float((m * size).sum())
            rec["pairs_share_by_group_size"] = {"16": float(((m - rem) * size).sum()) / tot, **{f"<={n}": float((rem * size)[(rem > 0) & (rem <= n)].sum()) / tot for n in (4, 8, 12, 15)}}
            segs = (size + 255) // 256
            rec["tasks_by_group_size"] = {"16": int(((m // 16) * segs).sum()), **{f"<={n}": int(segs[(rem > 0) & (rem <= n)].sum()) for n in (4, 8, 12, 15)}}
        if args.stress:
            bad = 0
            for i in range(args.stress):
                o = ix.scan_tensors(qg, keys, nkeys, k=10, check=False)
                bad += int(not (torch.equal(o[0], dist) and torch.equal(o[1], idx) and torch.equal(o[2], nc)))
            rec["stress_runs"], rec["stress_mismatches"] = args.stress, bad
        if not args.no_check:
            if ref_out is None:
                ref = Indexer(hashing, cg, dist_fn, compat=compat, algo="query")
                ref_out = ref.scan_tensors(qg, keys, nkeys, k=10)[:3]
                del ref
            d0, i0, n0 = ref_out
            rec["ncand_equal"] = bool(torch.equal(n0, nc))
            rec["ids_equal_frac"] = float((i0 == idx).all(1).float().mean())
            both = (i0 >= 0) & (idx >= 0)
            rec["max_abs_dist_diff"] = float((d0 - dist).abs()[both].max())
        rec["window_rows"], rec["workload"], rec["row_stride"] = ix.last_window, args.workload, ix.row_stride
        rec["pairs"], rec["probed_buckets"], rec["unique_probed_rows"], rec["unique_probed_bytes"] = n_pairs, int(len(probed_)), unique_rows, unique_rows * 4 * ix.dim
        by_window.setdefault(ix.last_window, []).append(rec["scan_kernel_ms"])
        if window:
            rec["n_cells"], rec["n_buckets"] = ix.cells(window)[3], ix.n_buckets
        if first is None:
            first = (dist, idx, nc)
        else:   # every window size must give the first one's bits
            rec["bits_equal_first_window"] = bool(torch.equal(dist.view(torch.int32), first[0].view(torch.int32)) and torch.equal(idx, first[1]) and torch.equal(nc, first[2]))
        print(json.dumps(rec), flush=True)
    if args.rounds > 1:
        print(json.dumps({"summary": args.workload, "rounds": args.rounds, "scan_kernel_ms_median_by_window": {str(w): float(np.median(v)) for w, v in by_window.items()},
                          "scan_kernel_ms_min_by_window": {str(w): float(np.min(v)) for w, v in by_window.items()}}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the tiled scan's fetched bytes go, by cause, from the task table the PLAN phase leaves in the workspace:

  probed      rows of the buckets the batch probes at all, each once          (what any row-sharing schedule must read)
  window      rows of the tasks' row ranges, each once                       (= probed + rows of UNPROBED buckets inside shared windows)
  staged      sum over tasks of the rows each one stages                     (= window + re-staging of a range by its 2nd, 3rd ... query group)
  cross_xcd   rows staged by tasks of the same range that land on DIFFERENT XCDs under the kernel's block -> task map
              (a second XCD's L2 cannot hit the first one's lines: these re-reads go to the fabric)
  lines       128-byte lines the tasks' row ranges touch, summed per XCD     (row ranges are not line-aligned: d = 100 rows are 400 bytes)

All in bytes of corpus rows (row_stride * 4 per row).  Compare with the PMC figure of the same launch (profiles/traffic_r05.json).

    python tools/traffic_tally.py --workload glove [--window 64]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import _capi, io, synth  # noqa: E402
from nlsh_amd.data import Glove, SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402


def xcd_of_task(t, xc=16):
    """Inverse of bscan3_kernel's block -> task map: t = ((j / XC) * 8 + (b & 7)) * XC + j % XC with j = b >> 3; workgroup b runs on XCD b & 7."""
    return (t // xc) % 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="glove", choices=["sift1m", "clusters", "glove"])
    ap.add_argument("--window", type=int, default=None)
    args = ap.parse_args()
    Q = 10_000
    if args.workload == "glove":
        N, d = 1_183_514, 100
        corpus_h, queries_h = synth.glove_manifold(N, d, seed=synth.SEED_DATA), synth.glove_manifold(Q, d, seed=synth.SEED_QUERY)
        ck, dist_fn, compat = "glove_manifold_h24.npz", Glove.distance, False
    else:
        N, d = 1_000_000, 128
        gen = synth.sift_manifold if args.workload == "sift1m" else synth.sift_like
        corpus_h, mean, std = synth.standardise(gen(N, d, seed=synth.SEED_DATA))
        queries_h, _, _ = synth.standardise(gen(Q, d, seed=synth.SEED_QUERY), mean, std)
        ck, dist_fn, compat = ("sift1m_manifold_h16.npz" if args.workload == "sift1m" else "sift1m_clusters_h16.npz"), SIFT.distance, True
    Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", ck))
    hashing = io.hashing_from_weights(Ws, bs, compat=compat)
    ix = Indexer(hashing, torch.from_numpy(corpus_h).cuda(), dist_fn, compat=compat, algo="tiled", window_rows=args.window)
    q = torch.from_numpy(queries_h).cuda()
    keys, nkeys = ix.hash_device(q, hash_times=10, seed=1000)
    ix.scan_tensors(q, keys, nkeys, k=10)
    ix.scan_tensors(q, keys, nkeys, k=10)          # second call: the trimmed task table
    torch.cuda.synchronize()
    n_tasks, max_tasks = int(ix.last_status.cpu()[0]), ix._last_max_tasks
    off, off_q = ctypes.c_size_t(), ctypes.c_size_t()
    _capi.check(_capi.lib().nlsh_scan_workspace_layout(Q, keys.shape[1], 10, max_tasks, ix.n_buckets, d, _capi.SCAN_BUCKET_TILED, ctypes.byref(off), ctypes.byref(off_q), None))
    ws = next(w for (s_, bm), w in ix._ws.items() if bm)
    tasks = ws[off.value:off.value + 16 * n_tasks].view(torch.int32).view(n_tasks, 4).cpu().numpy().astype(np.int64)
    row0, nrows = tasks[:, 2], tasks[:, 3]
    # rows of a task that ANY of its queries owns: the hull [min lo, max hi) of the slots' row ranges, and their exact union
    qr = ws[off_q.value:off_q.value + 128 * n_tasks].view(torch.int32).view(n_tasks, 16, 2).cpu().numpy().astype(np.int64)
    rng = qr[:, :, 1]
    lo_s, hi_s = rng & 0xFFFF, rng >> 16
    live = np.arange(16)[None, :] < tasks[:, 1:2]
    hull_rows = int((np.where(live, hi_s, 0).max(1) - np.where(live, lo_s, 1 << 20).min(1)).sum())
    union_rows = 0
    for t in np.nonzero(nrows <= 256)[0]:
        m = np.zeros(257, np.int32)
        np.add.at(m, lo_s[t][live[t]], 1)
        np.add.at(m, hi_s[t][live[t]], -1)
        union_rows += int((np.cumsum(m)[:256] > 0).sum())
    rb = ix.row_stride * 4
    # rows of the probed buckets, each once
    uk = ix.uniq_keys.cpu().numpy().astype(np.int64)
    kh, nh = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
    flat = kh[np.arange(kh.shape[1])[None, :] < nh[:, None]]
    pos = np.searchsorted(uk, flat)
    pos[pos >= len(uk)] = 0
    probed = np.unique(pos[uk[pos] == flat])
    probed_rows = int(ix.bucket_sizes[probed].sum())
    # distinct row ranges (a range = one segment of a cell; its query groups are consecutive tasks)
    rng_key = row0 * 1024 + nrows
    uniq_rng, first, inv = np.unique(rng_key, return_index=True, return_inverse=True)
    window_rows = int(nrows[first].sum())
    staged_rows = int(nrows.sum())
    xcd = xcd_of_task(np.arange(n_tasks))
    pair = np.unique(inv * 8 + xcd)                                  # (range, XCD) pairs that occur
    per_xcd_rows = int(nrows[first][pair // 8].sum())               # rows each XCD's L2 has to fetch at least once
    lo, hi = row0[first] * rb, (row0[first] + nrows[first]) * rb
    lines_once = int(((hi + 127) // 128 - lo // 128).sum()) * 128
    lines_per_xcd = int((((hi + 127) // 128 - lo // 128))[pair // 8].sum()) * 128
    out = {"workload": args.workload, "window_rows": ix.last_window, "tasks": n_tasks, "row_ranges": int(len(uniq_rng)), "row_bytes": rb,
           "probed_bytes": probed_rows * rb, "window_bytes": window_rows * rb, "staged_bytes": staged_rows * rb,
           "per_xcd_bytes": per_xcd_rows * rb, "hull_of_slot_ranges_bytes": hull_rows * rb, "union_of_slot_ranges_bytes": union_rows * rb, "lines_once_bytes": lines_once, "lines_per_xcd_bytes": lines_per_xcd,
           "unprobed_rows_in_windows": (window_rows - probed_rows) * rb, "second_and_later_query_groups": (staged_rows - window_rows) * rb,
           "ranges_on_more_than_one_xcd": (per_xcd_rows - window_rows) * rb, "line_granularity": lines_once - window_rows * rb}
    out["ratios_to_probed"] = {k: round(out[k] / out["probed_bytes"], 3) for k in ("window_bytes", "staged_bytes", "hull_of_slot_ranges_bytes", "union_of_slot_ranges_bytes", "per_xcd_bytes", "lines_once_bytes", "lines_per_xcd_bytes")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Same-box comparison of the host-side modes of `Indexer.query()` on the headline workload (ms per call over a loop that keeps
only the latest result, like bench.py's protocol region): numpy `tolist()` vs csrc/fastlists.c, the opt-ins
(`untracked_results`, `promote_results`, `defer_result_release`) on / off, 1 to 4 row ranges."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import indexer as ixmod, io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

N, d, Q = 1_000_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance)
qb = [torch.from_numpy(synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY + 17 * i), mean, std)[0]).cuda() for i in range(4)]
fast = ixmod._rows_to_lists


def run(builder, promote, defer, chunks, untrack=False, n=60, pause=True):
    ixmod._rows_to_lists = builder
    Indexer.promote_results, Indexer.defer_result_release, Indexer.query_chunks, Indexer.untracked_results = promote, defer, chunks, untrack
    Indexer.pause_collector_for_call = pause
    keep = None
    for i in range(8):
        keep = ix.query(qb[i % 4], 10, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        keep = ix.query(qb[i % 4], 10, 10)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


rows = []
for rep in range(3):
    for name, builder, promote, defer, chunks, untrack, pause in (
            ("tolist, r04 defaults (2 ranges, collector paused per range)", None, False, False, 2, False, False),
            ("fastlists, r04 defaults (tracked rows, 2 ranges, paused per range)", fast, False, False, 2, False, False),
            ("fastlists, 1 range, paused per range", fast, False, False, 1, False, False),
            ("fastlists, 4 ranges, paused per range", fast, False, False, 4, False, False),
            ("fastlists, 1 range, collector paused for the call", fast, False, False, 1, False, True),
            ("fastlists, 2 ranges, collector paused for the call", fast, False, False, 2, False, True),
            ("fastlists, 3 ranges, collector paused for the call", fast, False, False, 3, False, True),
            ("fastlists, 4 ranges, collector paused for the call", fast, False, False, 4, False, True),
            ("fastlists, r05 defaults (automatic ranges, paused for the call)", fast, False, False, None, False, True),
            ("fastlists + untracked rows", fast, False, False, None, True, True),
            ("fastlists + defer", fast, False, True, None, False, True),
            ("fastlists + untracked + defer (bench.py's opt-in region)", fast, False, True, None, True, True),
            ("fastlists + untracked + promote + defer", fast, True, True, None, True, True)):
        if builder is None or fast is not None:
            rows.append((name, round(run(builder, promote, defer, chunks, untrack, pause=pause), 4)))
Indexer.promote_results, Indexer.defer_result_release, Indexer.query_chunks, Indexer.untracked_results = False, False, None, False
Indexer.pause_collector_for_call = True
ixmod._rows_to_lists = fast
print(json.dumps(rows))

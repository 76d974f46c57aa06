#!/usr/bin/env python3
"""Diagnostic: 40 batches of the headline workload through the three-stream pipeline (run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/pipeline_timeline.py`), then
`python3 tools/pipeline_timeline.py --parse DIR` prints, for the last batches, when each kernel of a batch ran relative to
the scan kernels on the mid stream: how much of a step the mid stream waits for the next batch's PLAN."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "nlsh::" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("nlsh::")[1].split("(")[0].split("<")[0], int(r["Grid_Size_X"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    scans = [r for r in rows if r[2] == "bscan3_kernel"]
    last = scans[-12:]
    t0 = last[0][0]
    print("scan launches (us from the first shown): start end dur gap_to_previous_end")
    prev = None
    for s in last:
        print(f"  {(s[0] - t0) / 1e3:9.1f} {(s[1] - t0) / 1e3:9.1f} {(s[1] - s[0]) / 1e3:7.1f} {'' if prev is None else f'{(s[0] - prev) / 1e3:7.1f}'}")
        prev = s[1]
    print("all kernels in the window of the last 4 scans: name start end dur stream")
    w0 = scans[-4][0] - 100_000
    for r in rows:
        if r[0] >= w0:
            print(f"  {r[2]:16s} {(r[0] - t0) / 1e3:9.1f} {(r[1] - t0) / 1e3:9.1f} {(r[1] - r[0]) / 1e3:7.1f}  {r[4]}")
    sys.exit(0)

import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402
from nlsh_amd.pipeline import QueryPipeline  # noqa: E402

N, d, Q = 1_000_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance)
qb = [torch.from_numpy(synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY + 17 * i), mean, std)[0]).cuda() for i in range(4)]
pipe = QueryPipeline(ix, qb[0], k=10, hash_times=10, depth=int(os.environ.get("PIPE_DEPTH", "3")))
for i in range(40):
    pipe.submit(qb[i % 4], seed=1000 + i)
pipe.synchronize()

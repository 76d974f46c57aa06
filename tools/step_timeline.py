#!/usr/bin/env python3
"""Where the device-resident step's time goes, kernel by kernel (headline workload, sequential step on one stream).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/step_timeline.py      # run under the tracer
    python3 tools/step_timeline.py --parse DIR                                                  # per-kernel offsets / gaps
    python3 tools/step_timeline.py --graph                                                      # eager vs hipGraph replay of one step

--parse prints, as medians over the last steps: each kernel's start relative to the step's first kernel, its duration and
the idle gap in front of it (previous kernel's end -> this kernel's start), plus the step period.  --graph captures ONE
`query_tensors` step (fixed batch, fixed probe seed: kernel arguments are baked into a captured graph) in a
torch.cuda.CUDAGraph and times its replay against the eager launches: what launch gaps cost, as an upper bound of what a
graph of the step could save."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "nlsh::" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("nlsh::")[1].split("(")[0].split("<")[0]))
    rows.sort()
    # a step starts at its encode launch (r06: the query batch's form has a kernel name of its own, encode_hash_het_kernel)
    first = rows[0][2] if not any(r[2].startswith("encode_hash") for r in rows) else None
    starts = [i for i, r in enumerate(rows) if (r[2] == first if first else r[2].startswith("encode_hash"))]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-20:]
    names = [r[2] for r in steps[-1]]
    steps = [s for s in steps if [r[2] for r in s] == names]
    print(f"{len(steps)} steps; per kernel: start_us dur_us gap_before_us")
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731  (median: one step that waits for the profiler's buffer flush must not be averaged in)
    for j, nme in enumerate(names):
        st = med([s[j][0] - s[0][0] for s in steps]) / 1e3
        du = med([s[j][1] - s[j][0] for s in steps]) / 1e3
        gp = 0.0 if j == 0 else med([s[j][0] - s[j - 1][1] for s in steps]) / 1e3
        print(f"  {nme:22s} {st:8.1f} {du:8.1f} {gp:7.1f}")
    per = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps[:-1], steps[1:])]
    busy = med([sum(r[1] - r[0] for r in s) for s in steps]) / 1e3
    print(f"step period (median) {med(per) if per else 0.0:.1f} us; kernels busy {busy:.1f} us")
    sys.exit(0)

import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

wl = os.environ.get("STEP_WORKLOAD", "sift1m")     # sift1m | clusters | glove (the environment, not argv: the tracer passes argv through)
Q = 10_000
ck = lambda n: os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", n)   # noqa: E731
if wl == "glove":
    from nlsh_amd.data import Glove
    corpus_h = synth.glove_manifold(1_183_514, 100, seed=synth.SEED_DATA)
    Ws, bs = io.load_hasher_weights(ck("glove_manifold_h24.npz"))
    ix = Indexer(io.hashing_from_weights(Ws, bs, compat=False), torch.from_numpy(corpus_h).cuda(), Glove.distance, compat=False)
    qb = [torch.from_numpy(synth.glove_manifold(Q, 100, seed=synth.SEED_QUERY + 17 * i)).cuda() for i in range(4)]
else:
    gen = synth.sift_like if wl == "clusters" else synth.sift_manifold
    corpus_h, mean, std = synth.standardise(gen(1_000_000, 128, seed=synth.SEED_DATA))
    Ws, bs = io.load_hasher_weights(ck("sift1m_clusters_h16.npz" if wl == "clusters" else "sift1m_manifold_h16.npz"))
    ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance)
    qb = [torch.from_numpy(synth.standardise(gen(Q, 128, seed=synth.SEED_QUERY + 17 * i), mean, std)[0]).cuda() for i in range(4)]
ix.query_tensors(qb[0], k=10, hash_times=10, seed=1)       # sizes the task table
torch.cuda.synchronize()


def eager(n):
    t0 = time.perf_counter()
    for i in range(n):
        out = ix.query_tensors(qb[i % 4], k=10, hash_times=10, seed=1000 + i, check=False)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


eager(5)
ms, _ = eager(40)
print(f"eager sequential step: {ms:.4f} ms", flush=True)
if "--graph" in sys.argv:
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ix.query_tensors(qb[0], k=10, hash_times=10, seed=7, check=False)      # workspace of the capture stream exists before capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = ix.query_tensors(qb[0], k=10, hash_times=10, seed=7, check=False)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    gms = 1e3 * (time.perf_counter() - t0) / 40
    ref = ix.query_tensors(qb[0], k=10, hash_times=10, seed=7, check=False)
    torch.cuda.synchronize()
    print(f"hipGraph replay of the same step: {gms:.4f} ms; results equal: {torch.equal(out[1], ref[1]) and torch.equal(out[0], ref[0])}", flush=True)

#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a bench.py run -> the scan kernel's launches grouped into RUNS of back-to-back launches.

bench.py launches the same full-batch scan kernel in several regions: the protocol regions (one launch per `Indexer.query()` call: the
device idles ~0.5 ms between two launches while the host builds Python lists, and the kernel runs at a higher clock), the sequential
device-resident region (steps back to back on one stream: the launches `roofline.avg_launch_ms` times with HIP events), the pipelined
region (scans 13 us apart, sharing the chip with the neighbouring batches' small kernels) and the untimed recomputation of the
candidate counts (one launch per host synchronisation).  `--stats` averages over the kernel NAME, i.e. over all of them.  A run = a
maximal sequence of launches whose starts are less than 0.6 ms apart; isolated launches are pooled.

    python tools/kernel_trace_regions.py DIR [kernel-substring=bscan3_kernel<0]
"""
import csv
import glob
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "bscan3_kernel<0"
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])))
rows.sort()
grid = max(r[2] for r in rows)
rows = [r for r in rows if r[2] == grid]
runs, cur = [], [rows[0]]
for prev, r in zip(rows[:-1], rows[1:]):
    if r[0] - prev[0] < 600_000:
        cur.append(r)
    else:
        runs.append(cur)
        cur = [r]
runs.append(cur)
iso = [(r[1] - r[0]) / 1e3 for run in runs if len(run) < 3 for r in run]
print(f"{needle}, grid {grid}: {len(rows)} launches")
if iso:
    print(f"  isolated launches (protocol regions, candidate-count recomputation): {len(iso)}, avg {sum(iso) / len(iso):.2f} us, min {min(iso):.2f}, max {max(iso):.2f}")
for run in runs:
    if len(run) >= 3:
        d = [(r[1] - r[0]) / 1e3 for r in run]
        period = (run[-1][0] - run[0][0]) / 1e3 / (len(run) - 1)
        print(f"  run of {len(run)} back-to-back launches, start-to-start {period:.1f} us: avg {sum(d) / len(d):.2f} us, min {min(d):.2f}, max {max(d):.2f}; last 20: avg {sum(d[-20:]) / len(d[-20:]):.2f} us")

#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) call count and average / min / max duration (us).
`--stats` averages over a kernel NAME; `Indexer.query()` scans a batch in two row ranges, so the same scan kernel runs with
half-batch grids (protocol region of bench.py) and full-batch grids (device-resident regions, the ones `roofline` times):
    python tools/kernel_trace_by_grid.py DIR [name-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict

root, needles = sys.argv[1], sys.argv[2:]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if needles and not any(n in name for n in needles):
            continue
        acc[(name, int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("kernel,grid_threads,workgroup_threads,calls,avg_us,min_us,max_us")
for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f'"{name}",{grid},{wg},{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f}')

#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch): python tools/pmc_summary.py DIR [substr]"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if needle in name:
            per[(name, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (name, _), cs in per.items():
        for c, v in cs.items():
            acc[name][c].append(v)
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, indent=1))

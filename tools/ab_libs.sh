#!/bin/bash
# A/B of library variants on the scan kernel: tools/scan_bench.py per (workload, variant), one JSON line each (same box, back to back,
# two rounds).  usage: bash tools/ab_libs.sh OUTFILE "variant1 variant2 ..." "workload1 ..." [scan_bench args]   ("" = shipped library)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; VARS=$2; WLS=$3; shift 3
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do for w in $WLS; do for v in shipped $VARS; do
  lib=$R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip_$v.so; [ $v = shipped ] && lib=$R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip.so
  NLSH_HIP_LIB=$lib timeout -k 10 200 python3 $R/tools/scan_bench.py --workload $w --iters 40 --tag $v "$@" 2>/dev/null | tail -1 >> $OUT
done; done; done
python3 - $OUT <<'PY'
import json,sys,collections
acc=collections.defaultdict(list)
for l in open(sys.argv[1]):
    try: j=json.loads(l)
    except Exception: continue
    if "scan_kernel_ms" in j: acc[(j.get("workload"),j.get("tag"))].append((j["scan_kernel_ms"],j["scan_kernel_ms_min"],j["step_ms"]))
for k,v in sorted(acc.items()): print(k, "  ".join(f"scan {a:.4f} (min {m:.4f}) step {b:.4f}" for a,m,b in v))
PY

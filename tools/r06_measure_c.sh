# r06: whole GPU suite on the current library, the default bench line (with the `workloads` object) timed by the wall clock, the two
# side workloads as lines of their own (the 5 % agreement check), and the counters this box's rocprofv3 offers
set -e
R=$PWD; O=$R/gpurun_out/r06c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
t0=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "default bench.py wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
timeout -k 10 300 python bench.py --workload glove --no-cpu-baseline > $O/bench_glove.json 2>/dev/null
timeout -k 10 300 python bench.py --data clusters --no-cpu-baseline > $O/bench_clusters.json 2>/dev/null
(cd /tmp && TMPDIR=/tmp timeout -k 10 120 rocprofv3 --list-avail > $O/rocprof_list_avail.txt 2>&1) || true
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06c/bench.json").read().strip().splitlines()[-1])
print("headline", b["value"], b["ms_per_step"], b["device_resident_ms_per_step"], b["device_resident_pipelined_qps"], b["roofline"]["avg_launch_ms"], b["roofline"]["frac"], b["recall_at_10"])
for t in ("glove","clusters"):
    w=b["workloads"][t]; j=json.loads(open(f"gpurun_out/r06c/bench_{t}.json").read().strip().splitlines()[-1])
    print(t, "in-line", w["scan_ms"], w["device_resident_ms_per_step"], w["roofline"]["frac"], w["recall_at_10"], w["mean_candidates_per_query"], "| own line", j["roofline"]["avg_launch_ms"], j["device_resident_ms_per_step"], j["roofline"]["frac"], j["recall_at_10"])
PY

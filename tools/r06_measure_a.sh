set -e
R=$PWD; O=$R/gpurun_out/r06a; mkdir -p $O
rm -rf /tmp/kts && (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -- python3 $R/tools/step_timeline.py > /dev/null 2> /tmp/kts.err)
python3 $R/tools/step_timeline.py --parse /tmp/kts > $O/step_timeline.txt
cat $O/step_timeline.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python bench.py --workload glove --no-cpu-baseline > $O/bench_glove.json 2>/dev/null
timeout -k 10 300 python bench.py --data clusters --no-cpu-baseline > $O/bench_clusters.json 2>/dev/null
for w in 1 8; do timeout -k 10 200 python tools/shard_step_profile.py --world $w --rank 0 --steps 50 --pipeline; done 2>/dev/null > $O/shard_step_profile_pipelined.jsonl
cat $O/shard_step_profile_pipelined.jsonl
python - <<'PY'
import json
for f in ("bench","bench_glove","bench_clusters"):
    j=json.loads(open(f"gpurun_out/r06a/{f}.json").read().strip().splitlines()[-1])
    print(f, j["value"], j.get("device_resident_ms_per_step"), j.get("device_resident_pipelined_qps"), j["roofline"].get("avg_launch_ms"), j["roofline"].get("frac"))
PY

set -e
R=$PWD; O=$R/gpurun_out/r06k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_deep100m.py > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for v in "" nofuse; do
  lib=$R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip${v:+_$v}.so
  for wl in sift1m glove; do
    rm -rf /tmp/kts && (cd /tmp && STEP_WORKLOAD=$wl NLSH_HIP_LIB=$lib TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -- python3 $R/tools/step_timeline.py > /dev/null 2> /tmp/kts.err)
    echo "== variant ${v:-shipped} $wl" >> $O/step_timeline.txt
    python3 $R/tools/step_timeline.py --parse /tmp/kts >> $O/step_timeline.txt
  done
done
cat $O/step_timeline.txt

# r06: A/B of the query-batch encoder forms inside the device step (kernel trace of tools/step_timeline.py per library variant)
set -e
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }; }
[ -n "$SKIP_TESTS" ] || tail -2 $O/tests.txt
for v in "" $VARIANTS; do
  lib=$R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip${v:+_$v}.so
  rm -rf /tmp/kts && (cd /tmp && NLSH_HIP_LIB=$lib TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -- python3 $R/tools/step_timeline.py > /dev/null 2> /tmp/kts.err)
  echo "== variant ${v:-shipped}" >> $O/step_timeline.txt
  python3 $R/tools/step_timeline.py --parse /tmp/kts >> $O/step_timeline.txt
done
cat $O/step_timeline.txt

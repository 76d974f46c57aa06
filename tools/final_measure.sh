set -e
R=$PWD
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > gpurun_out/final/tests.txt
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --query-chunks 1 > $R/gpurun_out/final/bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $R/gpurun_out/final/kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --no-cpu-baseline > /tmp/kt2.out 2> /tmp/kt2.err
python3 $R/tools/kernel_trace_by_grid.py /tmp/kt2 bscan3 encode_hash bmerge bplan bscan_kernel bscatter bcount > $R/gpurun_out/final/kernel_trace_by_grid_default_command.csv
rm -rf /tmp/pm && rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --query-chunks 1 > /tmp/pm.out 2> /tmp/pm.err
python3 $R/tools/pmc_summary.py /tmp/pm > $R/gpurun_out/final/pmc.json
cd $R
python bench.py --workload glove --no-cpu-baseline --pipeline on > gpurun_out/final/bench_glove.json 2>/dev/null
cat gpurun_out/final/tests.txt; cat gpurun_out/final/bench.json

# End-of-round measurement set (run on the GPU box from the repo root through gpurun; results under gpurun_out/final/, the ones
# to be judged are copied to profiles/ afterwards).  Two calls (the Deep100M part needs the box to itself for a few minutes):
#   bash tools/final_measure.sh lines    # tests, bench lines, shard-step profiles, host-enqueue probe, order / cache-flush experiment
#   bash tools/final_measure.sh counters # kernel statistics, step timelines, counters of configs[1] (both generators, both L2 forms) and configs[2]
#   bash tools/final_measure.sh deep     # configs[4] on one GPU: pipelined line, sequential line, counters of the sequential command
#   (main = lines + counters in one call: longer than one gpurun call may last)
# Needs lib/libnlsh_hip_trace.so for the clock pass:
#   make -C neural-locality-sensitive-hashing_amd/csrc VARIANT=trace EXTRA="-DNLSH_SCAN_TRACE -DNLSH_SCAN_TRACE_CLOCK"
set -e
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
PMC="FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU"
pmc_bench() {   # $1 = tag, rest = bench.py flags: counters of the bench command in their own pass (no tracing domains beside --pmc)
  tag=$1; shift
  rm -rf /tmp/pm_$tag
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $PMC --output-format csv -d /tmp/pm_$tag -- python3 $R/bench.py --no-cpu-baseline --no-side-workloads --steps 3 --warmup 1 --query-chunks 1 "$@" > $O/pmc_${tag}_line.json 2> /tmp/pm_$tag.err)
  python3 $R/tools/pmc_summary.py /tmp/pm_$tag > $O/pmc_$tag.json
}
if [ "${1:-main}" = "main" ] || [ "$1" = "lines" ]; then
  python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > $O/tests.txt
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --l2-form folded --no-cpu-baseline > $O/bench_folded.json 2>/dev/null
  python bench.py --data clusters --no-cpu-baseline > $O/bench_clusters.json 2>/dev/null
  python bench.py --workload glove --no-cpu-baseline > $O/bench_glove.json 2>/dev/null
  python bench.py --workload glove --no-cpu-baseline --algo query > $O/bench_glove_query_major.json 2>/dev/null
  for w in 1 2 4 8; do python tools/shard_step_profile.py --world $w --rank 0 --steps 50; done 2>/dev/null > $O/shard_step_profile_sequential.jsonl
  for g in on off; do for w in 1 2 4 8; do python tools/shard_step_profile.py --world $w --rank 0 --steps 100 --pipeline --graph $g; done; done 2>/dev/null > $O/shard_step_profile_pipelined.jsonl
  python tools/graph_enqueue_probe.py 2>/dev/null > $O/graph_enqueue_probe.json
  for w in glove clusters sift1m; do python tools/order_alternation.py $w 2>/dev/null; done > $O/order_alternation_and_cache_flush.jsonl
  cat $O/tests.txt; cat $O/bench.json
fi
if [ "${1:-main}" = "main" ] || [ "$1" = "counters" ]; then
  CLK=""
  if [ -f $R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip_trace.so ]; then
    NLSH_HIP_LIB=$R/neural-locality-sensitive-hashing_amd/lib/libnlsh_hip_trace.so python tools/scan_clock.py > $O/scan_clock.txt 2>/dev/null || true
    CLK=$(grep -oE 'clock median [0-9.]+' $O/scan_clock.txt | head -1 | awk '{print $3}')
  fi
  # per-kernel statistics of the bench command (one row range per query() call: every scan launch has the full-batch grid)
  rm -rf /tmp/kt && (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --no-side-workloads --query-chunks 1 > $O/bench_under_rocprof.json 2> /tmp/kt.err)
  cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
  python3 $R/tools/kernel_trace_regions.py /tmp/kt > $O/kernel_trace_regions.txt
  : > $O/step_timeline.txt
  for wl in sift1m clusters glove; do
    rm -rf /tmp/kts && (cd /tmp && STEP_WORKLOAD=$wl TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -- python3 $R/tools/step_timeline.py > /dev/null 2> /tmp/kts.err)
    echo "== $wl (sequential device step, one nlsh_query_batch call per batch)" >> $O/step_timeline.txt
    python3 $R/tools/step_timeline.py --parse /tmp/kts >> $O/step_timeline.txt 2>/dev/null || true
  done
  for wl in glove clusters; do
    flags="--workload glove"; [ $wl = clusters ] && flags="--data clusters"
    rm -rf /tmp/kt_$wl && (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$wl -- python3 $R/bench.py --no-cpu-baseline --query-chunks 1 $flags > $O/bench_${wl}_under_rocprof.json 2> /tmp/kt_$wl.err)
    cp $(find /tmp/kt_$wl -name '*kernel_stats.csv' | head -1) $O/${wl}_kernel_stats.csv
  done
  pmc_bench sift1m
  pmc_bench sift1m_folded --l2-form folded
  pmc_bench clusters --data clusters
  pmc_bench glove --workload glove
  # what the L2's fabric-side read requests of the GloVe launch are made of: all of them, those routed to local DRAM (as opposed to
  # GMI / IO -- NOT "missed the Infinity Cache": no counter of this rocprofv3 sits behind the MALL), the 32-byte ones
  rm -rf /tmp/pm_ea && (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/pm_ea -- python3 $R/bench.py --no-cpu-baseline --no-side-workloads --steps 3 --warmup 1 --query-chunks 1 --workload glove > /dev/null 2> /tmp/pm_ea.err) || true
  python3 $R/tools/pmc_summary.py /tmp/pm_ea bscan3 > $O/pmc_glove_ea_requests.json || true
  python3 tools/make_traffic.py --entry $O/pmc_sift1m.json:$O/pmc_sift1m_line.json${CLK:+:$CLK} --entry $O/pmc_sift1m_folded.json:$O/pmc_sift1m_folded_line.json \
      --entry $O/pmc_clusters.json:$O/pmc_clusters_line.json --entry $O/pmc_glove.json:$O/pmc_glove_line.json \
      $( [ -f $O/order_alternation_and_cache_flush.jsonl ] && echo --origin glove:manifold:exact:$O/pmc_glove_ea_requests.json:$O/order_alternation_and_cache_flush.jsonl ) > $O/traffic.json
  cat $O/step_timeline.txt
fi
if [ "$1" = "deep" ]; then
  CK=$R/neural-locality-sensitive-hashing_amd/checkpoints/deep100m_manifold_h32.npz
  python tools/scale_deep100m.py --load-hash $CK > $O/deep100m_pipelined.log 2>&1
  rm -rf /tmp/pm_deep
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $PMC --output-format csv -d /tmp/pm_deep -- python3 $R/tools/scale_deep100m.py --load-hash $CK --pipeline off --steps 2 --recall-queries 0 > /tmp/pm_deep.out 2> /tmp/pm_deep.err)
  python3 tools/pmc_summary.py /tmp/pm_deep bscan > $O/pmc_deep100m.json
  python tools/scale_deep100m.py --load-hash $CK --pipeline off --pmc-summary $O/pmc_deep100m.json > $O/deep100m_sequential.log 2>&1
  tail -1 $O/deep100m_pipelined.log; tail -1 $O/deep100m_sequential.log
fi

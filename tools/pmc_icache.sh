#!/bin/bash
# Two rocprofv3 --pmc passes over tools/scan_bench.py: instruction-cache and scalar-data-cache counters of the tiled scan kernel.
# bash tools/pmc_icache.sh [scan_bench args, e.g. --workload glove --window 128]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_ic -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/pmc_ic.log 2>&1 || tail -5 /tmp/pmc_ic.log
python3 $R/tools/pmc_summary.py /tmp/pmc_ic bscan3 | tr -d '\n' | sed 's/  */ /g'; echo
rm -rf /tmp/pmc_dc
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQC_ICACHE_BUSY_CYCLES --output-format csv -d /tmp/pmc_dc -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/pmc_dc.log 2>&1 || tail -5 /tmp/pmc_dc.log
python3 $R/tools/pmc_summary.py /tmp/pmc_dc bscan3 | tr -d '\n' | sed 's/  */ /g'; echo
rm -rf /tmp/pmc_w
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_w -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/pmc_w.log 2>&1 || tail -5 /tmp/pmc_w.log
python3 $R/tools/pmc_summary.py /tmp/pmc_w bscan3 | tr -d '\n' | sed 's/  */ /g'; echo

#!/bin/bash
# One rocprofv3 --pmc pass (instruction counts) over tools/scan_bench.py; prints per-dispatch means for kernels matching $1.
R=${GRAFT_REPO_ROOT:-$PWD}
K=${1:-bscan3}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_c -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/pmc_c.log 2>&1 || tail -5 /tmp/pmc_c.log
python3 $R/tools/pmc_summary.py /tmp/pmc_c $K | tr -d '\n' | sed 's/  */ /g'; echo

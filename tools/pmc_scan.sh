#!/bin/bash
# Three rocprofv3 --pmc passes (8 SQ counters each) over tools/scan_bench.py; prints the per-dispatch means of the
# kernels whose name contains $1 (default bscan3).  Run on the GPU box from the repo root: bash tools/pmc_scan.sh [substr] [scan_bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
K=${1:-bscan3}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH" \
           "SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/pmc_$i.log 2>&1 || tail -5 /tmp/pmc_$i.log
  python3 $R/tools/pmc_summary.py /tmp/pmc_$i $K
done

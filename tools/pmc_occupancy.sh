#!/bin/bash
# One rocprofv3 --pmc pass: resident waves (SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES) and vector-memory queue depth of the scan kernel.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_o
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_o -- python3 $R/tools/scan_bench.py --no-check --iters 5 > /tmp/pmc_o.log 2>&1 || tail -5 /tmp/pmc_o.log
python3 $R/tools/pmc_summary.py /tmp/pmc_o bscan3

#!/bin/bash
# One rocprofv3 --pmc pass per row window over tools/scan_bench.py: HBM traffic (FETCH_SIZE) and instruction counts of the tiled scan
# kernel.  bash tools/pmc_window.sh WORKLOAD "0 64 128 256" [kernel substr]
R=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-glove}; WINS=${2:-"0 128"}; K=${3:-bscan3}
cd /tmp && export TMPDIR=/tmp
for w in $WINS; do
  rm -rf /tmp/pmc_w$w
  rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_w$w -- python3 $R/tools/scan_bench.py --workload $WL --window $w --no-check --iters 5 > /tmp/pmc_w$w.log 2>&1 || tail -5 /tmp/pmc_w$w.log
  echo "window $w"; python3 $R/tools/pmc_summary.py /tmp/pmc_w$w $K | tr -d '\n' | sed 's/  */ /g'; echo
done

#!/bin/bash
# r06: per-class issue / stall counters of the headline's tiled scan launch, five rocprofv3 --pmc passes over
# tools/scan_bench.py (each pass its own process; no tracing domain beside --pmc), then ONE thread-trace attempt (--att).
# Run on the GPU box from the repo root: bash tools/r06_scan_stalls.sh [scan_bench args]; results in gpurun_out/r06e/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rm -rf /tmp/st_$i
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d /tmp/st_$i -- python3 $R/tools/scan_bench.py --no-check --iters 5 "$@" > /tmp/st_$i.log 2>&1 || tail -5 /tmp/st_$i.log
  python3 $R/tools/pmc_summary.py /tmp/st_$i bscan3 > $O/stall_pass_$i.json
  echo "pass $i done" >&2
done
grep -h '"scan_ms"\|scan_ms' /tmp/st_1.log | tail -2 > $O/scan_bench_line_under_pmc.txt
rm -rf /tmp/att1
(timeout -k 10 200 rocprofv3 --att --kernel-include-regex bscan3 -d /tmp/att1 -- python3 $R/tools/scan_bench.py --no-check --iters 2 "$@" > $O/att_attempt.log 2>&1; echo "att exit code $?" >> $O/att_attempt.log) || true
ls -R /tmp/att1 2>/dev/null | head -30 >> $O/att_attempt.log
tail -15 $O/att_attempt.log >&2

#!/bin/bash
# Same-box A/B of the shipped library against a variant build, interleaved (shipped, variant, shipped, ...):
#   make -C neural-locality-sensitive-hashing_amd/csrc VARIANT=name EXTRA="-D..."      # builds lib/libnlsh_hip_name.so
#   gpurun -- 'bash tools/ab_variant.sh name [workload ...]'                            # workloads of tools/scan_bench.py; default sift1m
# Every line: tag, mean scan ms, fastest launch, device step, ids equal to the query-major schedule, candidate counts equal.
set -e
V=$1; shift || true
WL=${*:-sift1m}
mkdir -p gpurun_out/ab
L=neural-locality-sensitive-hashing_amd/lib
: > gpurun_out/ab/$V.txt
for r in 1 2 3; do
 for W in $WL; do
  for v in "" _$V; do
      NLSH_HIP_LIB=$PWD/$L/libnlsh_hip$v.so timeout -k 10 200 python tools/scan_bench.py --workload $W --iters 40 --tag "r$r$v-$W" >> gpurun_out/ab/$V.txt 2>gpurun_out/ab/err.txt
      tail -1 gpurun_out/ab/$V.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], round(r['scan_kernel_ms'],4), round(r['scan_kernel_ms_min'],4), round(r.get('step_ms',0),4), r.get('ids_equal_frac', r.get('ids_eq')), r.get('ncand_equal', r.get('nc_eq')))"
  done
 done
done

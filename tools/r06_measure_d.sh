# r06: graph slots against staged slots (pipelined local step of 1 / 2 / 4 / 8 bucket shards emulated on one GPU), after the pipeline tests
set -e
R=$PWD; O=$R/gpurun_out/r06d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_facade.py tests/test_gpu_distributed.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for g in on off; do for w in 1 2 4 8; do timeout -k 10 200 python tools/shard_step_profile.py --world $w --rank 0 --steps 100 --pipeline --graph $g; done; done 2>/dev/null > $O/shard_step_profile_graph_vs_staged.jsonl
cat $O/shard_step_profile_graph_vs_staged.jsonl

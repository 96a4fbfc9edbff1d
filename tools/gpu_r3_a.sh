#!/bin/bash
# round 3, first GPU call: the GPU tests, the headline, config 5 at its own size (10 000 spp, once) and its 1/8 shard
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03a_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r03a_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03a_gpu_tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err || { tail -20 gpurun_out/r03a_bench.err; exit 1; }
cat gpurun_out/r03a_bench.json
echo "config 5 full"
timeout -k 10 400 python bench.py --workload book2_final --steps 1 --warmup 0 --cpu-seconds 10 > gpurun_out/r03a_config5_full.json 2> gpurun_out/r03a_config5_full.err || { tail -20 gpurun_out/r03a_config5_full.err; exit 1; }
cat gpurun_out/r03a_config5_full.json
timeout -k 10 200 python tools/config5_shard.py 10000 8 > gpurun_out/r03a_config5_shard.json 2>&1
cat gpurun_out/r03a_config5_shard.json

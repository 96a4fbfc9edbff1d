#!/bin/bash
# the round's evidence in ONE call, all from the same (final) library: GPU tests, counter summaries + bench lines of all four workloads,
# config 5 at its own size (10 000 spp, once) and its 1/8 shard, the differential fuzz campaign
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r03_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03_final_gpu_tests.log
bash tools/gpu_r3_profiles.sh
bash tools/gpu_r3_lines.sh > /dev/null
echo "lines done"
timeout -k 10 500 python bench.py --workload book2_final --steps 1 --warmup 0 --cpu-seconds 10 > gpurun_out/r03_config5_full.json 2> gpurun_out/r03_config5_full.err || { tail -20 gpurun_out/r03_config5_full.err; exit 1; }
cut -c1-250 gpurun_out/r03_config5_full.json
timeout -k 10 200 python tools/config5_shard.py 10000 8 2>/dev/null > gpurun_out/r03_config5_shard.json
cat gpurun_out/r03_config5_shard.json
timeout -k 10 200 python tools/queue_perf.py 64 > gpurun_out/r03_traversal_modes_perf.txt 2>/dev/null
cat gpurun_out/r03_traversal_modes_perf.txt
timeout -k 10 600 python tools/fuzz_campaign.py --seeds 20000 --first 30000 --variants > gpurun_out/fuzz_r03v.txt 2>&1 || { tail -5 gpurun_out/fuzz_r03v.txt; exit 1; }
tail -1 gpurun_out/fuzz_r03v.txt

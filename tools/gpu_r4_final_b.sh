#!/bin/bash
# round-4 evidence, part B (one gpurun call): config 5 at its own size (10 000 spp, once) and its 1/8 shard, shard timings, the scaling run's shapes
# through bench.py's own N > 1 path (bare invocation, gloo, every rank on this GPU: structure of the line, not a scaling number), fuzz campaigns
set -e
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --workload book2_final --steps 1 --warmup 0 --cpu-seconds 10 > gpurun_out/r04_config5_full.json 2> gpurun_out/r04_config5_full.err || { tail -20 gpurun_out/r04_config5_full.err; exit 1; }
cut -c1-250 gpurun_out/r04_config5_full.json
timeout -k 10 200 python tools/config5_shard.py 10000 8 2>/dev/null > gpurun_out/r04_config5_shard.json
cat gpurun_out/r04_config5_shard.json
python tools/shard_perf.py > gpurun_out/r04_shard_perf.txt 2>/dev/null
cat gpurun_out/r04_shard_perf.txt
for n in 2 4; do
  python bench.py --gpus $n --backend gloo --same-device --steps 3 --warmup 1 > gpurun_out/r04_rehearsal_n$n.json 2> gpurun_out/r04_rehearsal_n$n.err || { tail -20 gpurun_out/r04_rehearsal_n$n.err; exit 1; }
  cut -c1-160 gpurun_out/r04_rehearsal_n$n.json
done
timeout -k 10 400 python tools/fuzz_campaign.py --seeds 6000 --first 200000 --force-variant 6 > gpurun_out/r04_fuzz_variant6.txt 2>&1 || { tail -5 gpurun_out/r04_fuzz_variant6.txt; exit 1; }
tail -1 gpurun_out/r04_fuzz_variant6.txt
timeout -k 10 700 python tools/fuzz_campaign.py --seeds 20000 --first 300000 --variants > gpurun_out/r04_fuzz.txt 2>&1 || { tail -5 gpurun_out/r04_fuzz.txt; exit 1; }
tail -1 gpurun_out/r04_fuzz.txt

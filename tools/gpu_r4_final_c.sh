#!/bin/bash
# round-4 evidence, part C (one gpurun call, final tree): the whole GPU suite, the tolerance-mode measurement + counters, the bench lines of configs[2..4],
# and the fuzz campaign forced onto kernel variant 6 over sphere worlds of the reference's feature set
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r04_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_final_gpu_tests.log
bash tools/gpu_r4_final_a2.sh
timeout -k 10 400 python tools/fuzz_campaign.py --seeds 20000 --first 200000 --force-variant 6 > gpurun_out/r04_fuzz_variant6.txt 2>&1 || { tail -5 gpurun_out/r04_fuzz_variant6.txt; exit 1; }
tail -1 gpurun_out/r04_fuzz_variant6.txt

#!/bin/bash
# round 4, after the split of rt_device.hip and the pruning: the whole GPU suite, then the fuzz campaign forced onto kernel variant 6 (tolerance mode)
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_b_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r04_b_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r04_b_gpu_tests.log
timeout -k 10 500 python tools/fuzz_campaign.py --seeds 6000 --first 100000 --force-variant 6 > gpurun_out/r04_fuzz_variant6.txt 2>&1 || { tail -5 gpurun_out/r04_fuzz_variant6.txt; exit 1; }
tail -2 gpurun_out/r04_fuzz_variant6.txt

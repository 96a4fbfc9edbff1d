#!/bin/bash
# after a change of the default kernel: the whole GPU suite, a fuzz campaign over every variant, and the bench line
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_validate_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r04_validate_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_validate_gpu_tests.log
timeout -k 10 400 python tools/fuzz_campaign.py --seeds 30000 --first 600000 --variants > gpurun_out/r04_validate_fuzz.txt 2>&1 || { tail -5 gpurun_out/r04_validate_fuzz.txt; exit 1; }
tail -1 gpurun_out/r04_validate_fuzz.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_validate_bench.json 2> gpurun_out/r04_validate_bench.err || { tail -5 gpurun_out/r04_validate_bench.err; exit 1; }
cut -c1-300 gpurun_out/r04_validate_bench.json

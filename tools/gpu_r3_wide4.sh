#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_traversal_queue.py tests/test_multi_gpu_c.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_wide4_tests.log 2>&1 || { tail -30 gpurun_out/r03_wide4_tests.log; exit 1; }
tail -2 gpurun_out/r03_wide4_tests.log
timeout -k 10 200 python tools/queue_perf.py 64 > gpurun_out/r03_traversal_modes_perf.txt 2>&1
cat gpurun_out/r03_traversal_modes_perf.txt

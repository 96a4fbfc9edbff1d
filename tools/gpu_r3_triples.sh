#!/bin/bash
# experiment: (min, max, min) triples in global memory for the BIG kernels (-DRT_BIG_TRIPLES) — parity, then A/B on the Book-2 final scene
set -e
mkdir -p gpurun_out
L=$PWD/ray-tracing-v06_amd/csrc/build/librt06_triples.so
RT06_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_cornell.py tests/test_gpu_baseline_configs.py tests/test_multi_gpu_c.py -m gpu -x -q > gpurun_out/r03_triples_tests.log 2>&1 || { tail -30 gpurun_out/r03_triples_tests.log; exit 1; }
tail -2 gpurun_out/r03_triples_tests.log
AB_WORKLOAD=book2_final timeout -k 10 300 python tools/ab_kernels.py ray-tracing-v06_amd/csrc/librt06.so $L 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_ab_big_triples.txt
cat gpurun_out/r03_ab_big_triples.txt

#!/bin/bash
# round-4 evidence, part A (one gpurun call): the whole GPU suite, the counter summaries + bench lines of all four workloads (default kernel) and of
# the headline under kernel variant 6, the tolerance-mode measurement, bench lines of configs[2..4]
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r04_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_final_gpu_tests.log
bash tools/gpu_r4_profiles.sh
bash tools/gpu_r4_tol.sh > gpurun_out/r04_tol.log 2>&1 || { tail -20 gpurun_out/r04_tol.log; exit 1; }
tail -8 gpurun_out/r04_tol.log | cut -c1-300
python bench.py --workload book2_moving --steps 5 --warmup 1 > gpurun_out/r04_line_book2_moving.json 2> gpurun_out/r04_line_book2_moving.err
python bench.py --workload cornell_box --spp 1000 --steps 5 --warmup 1 > gpurun_out/r04_line_cornell_box.json 2> gpurun_out/r04_line_cornell_box.err
python bench.py --workload cornell_box --steps 2 --warmup 1 --cpu-seconds 5 > gpurun_out/r04_line_cornell_box_5000spp.json 2> gpurun_out/r04_line_cornell_box_5000spp.err
python bench.py --workload book2_final --spp 64 --steps 5 --warmup 1 > gpurun_out/r04_line_book2_final.json 2> gpurun_out/r04_line_book2_final.err
cat gpurun_out/r04_line_*.json | cut -c1-200

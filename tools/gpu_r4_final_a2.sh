#!/bin/bash
# round-4 evidence, part A continued (the first call stopped in the tolerance script's counter loop, which then still asked the refused Cornell box for
# variant 6): the tolerance-mode measurement on the final library and the bench lines of configs[2..4]
set -e
mkdir -p gpurun_out
bash tools/gpu_r4_tol.sh > gpurun_out/r04_tol.log 2>&1 || { tail -20 gpurun_out/r04_tol.log; exit 1; }
tail -6 gpurun_out/r04_tol.log | cut -c1-300
python bench.py --workload book2_moving --steps 5 --warmup 1 > gpurun_out/r04_line_book2_moving.json 2> gpurun_out/r04_line_book2_moving.err
python bench.py --workload cornell_box --spp 1000 --steps 5 --warmup 1 > gpurun_out/r04_line_cornell_box.json 2> gpurun_out/r04_line_cornell_box.err
python bench.py --workload cornell_box --steps 2 --warmup 1 --cpu-seconds 5 > gpurun_out/r04_line_cornell_box_5000spp.json 2> gpurun_out/r04_line_cornell_box_5000spp.err
python bench.py --workload book2_final --spp 64 --steps 5 --warmup 1 > gpurun_out/r04_line_book2_final.json 2> gpurun_out/r04_line_book2_final.err
cat gpurun_out/r04_line_*.json | cut -c1-200

#!/bin/bash
# bench lines of BASELINE configs[2..4] at the sizes their r03 counter summaries were taken at (profiles/r03_<workload>_pmc_summary.csv)
set -e
python bench.py --workload book2_moving --steps 5 --warmup 1 > gpurun_out/r03_line_book2_moving.json 2> gpurun_out/r03_line_book2_moving.err
python bench.py --workload cornell_box --spp 1000 --steps 5 --warmup 1 > gpurun_out/r03_line_cornell_box.json 2> gpurun_out/r03_line_cornell_box.err
python bench.py --workload cornell_box --steps 2 --warmup 1 --cpu-seconds 5 > gpurun_out/r03_line_cornell_box_5000spp.json 2> gpurun_out/r03_line_cornell_box_5000spp.err
python bench.py --workload book2_final --spp 64 --steps 5 --warmup 1 > gpurun_out/r03_line_book2_final.json 2> gpurun_out/r03_line_book2_final.err
cat gpurun_out/r03_line_*.json | cut -c1-300

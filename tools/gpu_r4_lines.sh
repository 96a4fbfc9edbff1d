#!/bin/bash
# round 4: the bench lines that are priced with committed counter summaries, re-taken once the summaries of the FINAL library are in the tree
# (configs[2..4] at the sizes of their summaries, the Cornell box and config 5 at their own sizes, the N > 1 rehearsal lines)
set -e
mkdir -p gpurun_out
python bench.py --workload book2_moving --steps 5 --warmup 1 > gpurun_out/r04_line_book2_moving.json 2> gpurun_out/r04_line_book2_moving.err
python bench.py --workload cornell_box --spp 1000 --steps 5 --warmup 1 > gpurun_out/r04_line_cornell_box.json 2> gpurun_out/r04_line_cornell_box.err
python bench.py --workload cornell_box --steps 2 --warmup 1 --cpu-seconds 5 > gpurun_out/r04_line_cornell_box_5000spp.json 2> gpurun_out/r04_line_cornell_box_5000spp.err
python bench.py --workload book2_final --spp 64 --steps 5 --warmup 1 > gpurun_out/r04_line_book2_final.json 2> gpurun_out/r04_line_book2_final.err
timeout -k 10 500 python bench.py --workload book2_final --steps 1 --warmup 0 --cpu-seconds 10 > gpurun_out/r04_config5_full.json 2> gpurun_out/r04_config5_full.err
for n in 2 4; do
  python bench.py --gpus $n --backend gloo --same-device --steps 3 --warmup 1 2> gpurun_out/r04_rehearsal_n$n.err | grep '^{' > gpurun_out/r04_rehearsal_n$n.json
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_headline_again.json 2> gpurun_out/r04_headline_again.err
cat gpurun_out/r04_line_*.json gpurun_out/r04_config5_full.json gpurun_out/r04_rehearsal_n*.json gpurun_out/r04_headline_again.json | cut -c1-160

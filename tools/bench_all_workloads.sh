#!/bin/bash
# the headline (default) and BASELINE configs[2..4] through the same harness; one JSON line each
python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-parity
python bench.py --workload book2_moving --steps 2 --warmup 1 --cpu-seconds 4
python bench.py --workload cornell_box --steps 2 --warmup 1 --cpu-seconds 4
python bench.py --workload book2_final --spp 200 --steps 2 --warmup 1 --cpu-seconds 4

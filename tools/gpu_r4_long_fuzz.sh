#!/bin/bash
# the differential fuzz campaign over kernel variants 0-5 for as long as one gpurun call lasts (~107 worlds per second): FIRST = first seed
set -e
FIRST=${1:-400000}
mkdir -p gpurun_out
timeout -k 10 1120 python tools/fuzz_campaign.py --seeds 110000 --first $FIRST --variants > gpurun_out/r04_fuzz_long_$FIRST.txt 2>&1 || { tail -3 gpurun_out/r04_fuzz_long_$FIRST.txt; exit 1; }
tail -1 gpurun_out/r04_fuzz_long_$FIRST.txt

#!/bin/bash
# round-3 evidence in one call: headline profile + bench line, then the other BASELINE configs (profiles at one-pass sizes + bench lines)
set -e
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1 || { tail -20 gpurun_out/r03_profile_round.log; exit 1; }
tail -2 gpurun_out/r03_profile_round.log | cut -c1-400
WL_W=800 WL_H=800 WL_SPP=1000 WL_DEPTH=50 bash tools/profile_workload.sh r03 book2_moving > gpurun_out/r03_profile_b2m.log 2>&1 || { tail -20 gpurun_out/r03_profile_b2m.log; exit 1; }
echo "book2_moving done"
WL_W=600 WL_H=600 WL_SPP=1000 WL_DEPTH=50 bash tools/profile_workload.sh r03 cornell_box > gpurun_out/r03_profile_cb.log 2>&1 || { tail -20 gpurun_out/r03_profile_cb.log; exit 1; }
echo "cornell_box done"
WL_W=3840 WL_H=2160 WL_SPP=64 WL_DEPTH=50 bash tools/profile_workload.sh r03 book2_final > gpurun_out/r03_profile_b2f.log 2>&1 || { tail -20 gpurun_out/r03_profile_b2f.log; exit 1; }
echo "book2_final done"

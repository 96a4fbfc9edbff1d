#!/bin/bash
# round-4 evidence in one call, all from the same library: headline profile + bench line (default kernel, then kernel variant 6 = tolerance mode),
# then the other BASELINE configs (profiles at one-pass sizes + bench lines)
set -e
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1 || { tail -20 gpurun_out/r04_profile_round.log; exit 1; }
tail -1 gpurun_out/r04_profile_round.log | cut -c1-300
bash tools/profile_round.sh r04 6 > gpurun_out/r04_profile_round_v6.log 2>&1 || { tail -20 gpurun_out/r04_profile_round_v6.log; exit 1; }
tail -1 gpurun_out/r04_profile_round_v6.log | cut -c1-300
WL_W=800 WL_H=800 WL_SPP=1000 WL_DEPTH=50 bash tools/profile_workload.sh r04 book2_moving > gpurun_out/r04_profile_b2m.log 2>&1 || { tail -20 gpurun_out/r04_profile_b2m.log; exit 1; }
echo "book2_moving done"
WL_W=600 WL_H=600 WL_SPP=1000 WL_DEPTH=50 bash tools/profile_workload.sh r04 cornell_box > gpurun_out/r04_profile_cb.log 2>&1 || { tail -20 gpurun_out/r04_profile_cb.log; exit 1; }
echo "cornell_box done"
WL_W=3840 WL_H=2160 WL_SPP=64 WL_DEPTH=50 bash tools/profile_workload.sh r04 book2_final > gpurun_out/r04_profile_b2f.log 2>&1 || { tail -20 gpurun_out/r04_profile_b2f.log; exit 1; }
echo "book2_final done"

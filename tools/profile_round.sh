#!/bin/bash
# One gpurun call that produces everything tools/summarize_profiles.py copies into profiles/:
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh'   then   python tools/summarize_profiles.py r01
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r1 gpurun_out/pmc_fetch_r1 gpurun_out/pmc_write_r1 gpurun_out/pmcA gpurun_out/pmcB
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/prof_r1.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_r1 -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_r1 -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_write.log 2>&1
echo "hbm counters done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcA -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcB -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmcB.log 2>&1
echo "sq counters done"
cat gpurun_out/bench_r1.json

#!/bin/bash
# rocprofv3 summaries of another bench.py workload (the global-memory path on the Book-2 final scene by default):
#   gpurun --timeout 1100 -- 'bash tools/profile_workload.sh r02 book2_final'   then   python tools/summarize_profiles.py r02 book2_final
# Same rules as tools/profile_round.sh: counters in their own passes (--pmc with --kernel-trace only), HBM counters one per pass;
# plus the vector-L1 / L2 counters that say what the global-memory form of the streaming kernel waits for.
set -e
TAG=${1:-r03}
WL=${2:-book2_final}
export TMPDIR=/tmp
D=gpurun_out/prof_${TAG}_$WL
rm -rf $D && mkdir -p $D
python -c "import __graft_entry__ as G; print(G.load_package().capi.library_hash())" > $D/csrc_sha256.txt   # the hash embedded in the loaded BINARY
# profiled at an spp that fits ONE pass (bench.py scales the per-launch counters to other sample counts): WL_W WL_H WL_SPP WL_DEPTH
W=${WL_W:-3840}; H=${WL_H:-2160}; SPP=${WL_SPP:-64}; DEPTH=${WL_DEPTH:-50}
echo "$WL $W $H $SPP $DEPTH" > $D/config.txt
ARGS="--workload $WL --width $W --height $H --spp $SPP --depth $DEPTH"
BENCH="python3 bench.py $ARGS --steps 2 --warmup 1 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $BENCH > $D/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- $BENCH > $D/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- $BENCH > $D/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $D/pmcA -- $BENCH > $D/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $D/pmcB -- $BENCH > $D/pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $D/pmcC -- $BENCH > $D/pmcC.log 2>&1
echo "counters done"
python tools/summarize_profiles.py $TAG $WL > /dev/null     # on the box: profiles/${TAG}_${WL}_pmc_summary.csv of THIS library, for the line below
python bench.py $ARGS --steps 5 --warmup 1 > $D/bench.json 2> $D/bench.err
cat $D/bench.json

#!/bin/bash
# One gpurun call that produces everything tools/summarize_profiles.py copies into profiles/ (TAG = round, default r02):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03'   then   python tools/summarize_profiles.py r03
# Counters are collected in their own passes (rocprofv3 --pmc with --kernel-trace only), HBM counters one per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  The bench line is taken LAST, on the same box, after the counter
# summary of this very library has been written (so its roofline is priced with counters that are not stale).
set -e
TAG=${1:-r03}
VARIANT=${2:-0}      # kernel variant of the headline command: 0 (default), or 6 = tolerance mode -> profiles/${TAG}_bench_v6_*
export TMPDIR=/tmp
if [ "$VARIANT" = "0" ]; then NAME=bench; D=gpurun_out/prof_$TAG; VARG=""; VCFG=""; else NAME=bench_v$VARIANT; D=gpurun_out/prof_${TAG}_$NAME; VARG="--variant $VARIANT"; VCFG=" variant=$VARIANT"; fi
rm -rf $D && mkdir -p $D
python -c "import __graft_entry__ as G; print(G.load_package().capi.library_hash())" > $D/csrc_sha256.txt   # the hash embedded in the loaded BINARY
echo "book1_final 1200 800 500 50$VCFG" > $D/config.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 bench.py $VARG --steps 3 --warmup 1 --cpu-seconds 0 > $D/stats.log 2>&1
echo "stats done"
BENCH="python3 bench.py $VARG --steps 2 --warmup 1 --cpu-seconds 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- $BENCH > $D/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- $BENCH > $D/pmc_write.log 2>&1
echo "hbm counters done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $D/pmcA -- $BENCH > $D/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $D/pmcB -- $BENCH > $D/pmcB.log 2>&1
echo "sq counters done"
python tools/summarize_profiles.py $TAG $NAME > /dev/null     # on the box: profiles/${TAG}_bench_pmc_summary.csv of THIS library, for the line below
python bench.py $VARG --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
cat $D/bench.json

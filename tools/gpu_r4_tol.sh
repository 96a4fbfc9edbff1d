#!/bin/bash
# tolerance-mode box test (kernel variant 6): timing + full-frame deltas against the oracle (the two workloads of the reference's feature set at their own
# sizes; the other two are refused since the second acceptance run, which is recorded in the tool's output too), and the vector-instruction counters of
# variants 0 / 6 on those two workloads (one --pmc pass each, kernel trace only)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/tol
timeout -k 10 900 python tools/tolerance_mode.py > gpurun_out/tol/tolerance_mode.jsonl 2> gpurun_out/tol/tolerance_mode.err || { tail -20 gpurun_out/tol/tolerance_mode.err; exit 1; }
cat gpurun_out/tol/tolerance_mode.jsonl
: > gpurun_out/tol/counters.txt
for wl in "book1_final" "book2_moving"; do
  name=${wl%% *}
  for v in 0 6; do
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/tol/pmc_${name}_v$v -- python3 bench.py --workload $wl --variant $v --steps 2 --warmup 1 --cpu-seconds 0 --no-parity > gpurun_out/tol/pmc_${name}_v$v.log 2>&1
    python - <<PY >> gpurun_out/tol/counters.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/tol/pmc_${name}_v$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel_stream" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in acc.items()}
print("$wl variant $v: SQ_INSTS_VALU %.4g  lanes_active %.3f  SQ_INSTS_SALU %.4g  SQ_WAVE_CYCLES %.4g  SQ_WAIT_INST_ANY %.4g" % (m["SQ_INSTS_VALU"], m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_INSTS_VALU"]), m["SQ_INSTS_SALU"], m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"]), flush=True)
PY
  done
done
cat gpurun_out/tol/counters.txt

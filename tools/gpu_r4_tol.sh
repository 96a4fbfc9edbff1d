#!/bin/bash
# tolerance-mode box test (kernel variants 6 / 7): timing + full-frame deltas against the oracle on all four workloads, and the
# vector-instruction counters of variants 0 / 6 / 7 on config 2 (one --pmc pass each, kernel trace only)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/tol
timeout -k 10 900 python tools/tolerance_mode.py > gpurun_out/tol/tolerance_mode.jsonl 2> gpurun_out/tol/tolerance_mode.err || { tail -20 gpurun_out/tol/tolerance_mode.err; exit 1; }
cat gpurun_out/tol/tolerance_mode.jsonl
for v in 0 6 7; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/tol/pmc_v$v -- python3 bench.py --variant $v --steps 2 --warmup 1 --cpu-seconds 0 --no-parity > gpurun_out/tol/pmc_v$v.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/tol/pmc_v$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel_stream" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("variant $v", {k: sum(v) / len(v) for k, v in acc.items()}, flush=True)
PY
done

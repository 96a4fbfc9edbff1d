#!/bin/bash
# rocprofv3 counter passes of the ray-exchange kernel (variant 5) and, on the same box, of the streaming kernel (variant 3):
#   gpurun --timeout 900 -- 'bash tools/profile_xchg.sh r03'     -> gpurun_out/prof_r03_xchg/{v5,v3}_*/
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
D=gpurun_out/prof_${TAG}_xchg
rm -rf $D && mkdir -p $D
python -c "import __graft_entry__ as G; print(G.load_package().capi.library_hash())" > $D/csrc_sha256.txt
echo "RT06_XCHG=${RT06_XCHG:-default}" > $D/settings.txt
for V in 5 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/v${V}_stats -- python3 tools/xchg_prof.py $V 3 > $D/v${V}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $D/v${V}_pmcA -- python3 tools/xchg_prof.py $V 2 > $D/v${V}_pmcA.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $D/v${V}_pmcB -- python3 tools/xchg_prof.py $V 2 > $D/v${V}_pmcB.log 2>&1
  echo "variant $V done"
done
python3 - <<'PY'
import collections, csv, glob, os
D = sorted(glob.glob("gpurun_out/prof_*_xchg"))[-1]
for V in (5, 3):
    agg = collections.defaultdict(list)
    for d in ("pmcA", "pmcB"):
        for f in glob.glob(f"{D}/v{V}_{d}/*/*_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if "render_kernel" in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    v = {k: sum(x) / len(x) for k, x in agg.items()}
    if not v:
        continue
    print(f"variant {V}: " + ", ".join(f"{k} {x:.4g}" for k, x in sorted(v.items())))
    print(f"   lanes active {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_INSTS_VALU']):.3f}; scalar per vector {v['SQ_INSTS_SALU'] / v['SQ_INSTS_VALU']:.3f}; "
          f"LDS insts per vector {v['SQ_INSTS_LDS'] / v['SQ_INSTS_VALU']:.3f}; wait_any/wave_cycles {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}; "
          f"wait_inst_any/wave_cycles {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}; active_inst_any/wave_cycles {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
PY

#!/bin/bash
# tools/gpu/ledger.sh <tag>: the block ledger's raw material (tools/exp_ledger.py, tools/ledger_fit.py) in one gpurun call:
#   counts.jsonl   block executions / lanes of every launch of the job list, from the -DRT_LEDGER=1 build
#   cycles.jsonl   the same for the headline frame with shader-clock cycles per kind of block (-DRT_LEDGER=2)
#   pmc.csv        SQ_INSTS_VALU ... per dispatch of the PRODUCT library over the same job list (rocprofv3 --pmc)
#   ta.txt         TA-side counters of the headline frame, two per pass
set -u
TAG=${1:-ledger}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
RT_LIB_PATH=$R/tools/exp/librt_ledger.so timeout -k 10 300 python3 tools/exp_ledger.py > $OUT/counts.jsonl 2> $OUT/counts.err || { tail -5 $OUT/counts.err; exit 1; }
echo "counts: $(wc -l < $OUT/counts.jsonl) launches"
RT_LIB_PATH=$R/tools/exp/librt_ledger2.so RT_LEDGER_JOBS=1 timeout -k 10 120 python3 tools/exp_ledger.py > $OUT/cycles.jsonl 2> $OUT/cycles.err || { tail -5 $OUT/cycles.err; exit 1; }
timeout -k 10 120 python3 tools/exp_ledger.py > $OUT/product_plain.jsonl 2> $OUT/product_plain.err || { tail -5 $OUT/product_plain.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
unset RT_LIB_PATH
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_raw -- python3 $R/tools/exp_ledger.py > $OUT/product.jsonl 2> $OUT/product.err || { tail -5 $OUT/product.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/pmc_raw/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_path_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
with open(out + "/pmc.csv", "w") as fh:
    fh.write("dispatch_id,counter,value\n")
    for d, c, v in rows:
        fh.write(f"{d},{c},{v:.0f}\n")
print("pmc rows", len(rows))
PY
rm -rf $OUT/pmc_raw
cd $R
RT_EXP_REPS=0 bash tools/gpu/pmc.sh $TAG/ta "" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" > $OUT/ta.log 2>&1
tail -12 $OUT/ta.log

#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per argument (a quoted counter group) on bench.py, one frame.
# Usage: tools/pmc_gpu.sh <tag> "<counters A>" "<counters B>" ...   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_${TAG}.txt
i=0
for C in "$@"; do
  i=$((i+1))
  D=$OUT/pmcraw_${TAG}_$i
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py --no-cpu-baseline --no-bvh-compare --steps 1 --warmup 0 ${BENCH_ARGS:-} > $D.json 2> $D.err || echo "pass $i failed: $C" >> $OUT/pmc_${TAG}.txt
  python3 - "$D" >> $OUT/pmc_${TAG}.txt <<'PY'
import csv,glob,sys,collections
agg=collections.OrderedDict()
for f in glob.glob(sys.argv[1]+"/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rt_path_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
for k,v in agg.items(): print(f"{k} {v:.6g}")
PY
  rm -rf $D
done
cat $OUT/pmc_${TAG}.txt

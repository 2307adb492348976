#!/bin/bash
# One rocprofv3 --pmc pass per counter group on ONE frame of config #3 (tools/exp_kernels.py, 1 rep), counters summed per kernel name.
# Usage: tools/gpu/pmc.sh <tag> "<env assignments, comma separated or empty>" "<counters A>" "<counters B>" ...   -> gpurun_out/<tag>/pmc.txt
set -u
TAG=$1; ENVS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=',' read -ra KV <<< "$ENVS"; for kv in "${KV[@]}"; do [ -n "$kv" ] && export "$kv"; done
export RT_EXP_RANKS="" RT_EXP_REPS=${RT_EXP_REPS:-0} RT_EXP_ROUNDS=1
: > $OUT/pmc.txt
i=0
for C in "$@"; do
  i=$((i+1))
  D=$OUT/raw_$i
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/exp_kernels.py "arm:" > $D.log 2>&1 || echo "pass $i failed: $C" >> $OUT/pmc.txt
  python3 - "$D" >> $OUT/pmc.txt <<'PY'
import csv,glob,sys,collections
agg=collections.OrderedDict(); n=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not k.startswith("rt_"): continue
        key=(k,r["Counter_Name"])
        agg[key]=agg.get(key,0.0)+float(r["Counter_Value"]); n[key]+=1
for (k,c),v in agg.items(): print(f"{k:48s} {c:40s} {v:.6g}  launches={n[(k,c)]}")
PY
  rm -rf $D
done
cat $OUT/pmc.txt

#!/bin/bash
# per-kernel times of the wavefront pipeline (rocprofv3 kernel trace): tools/gpu/wf_prof.sh <tag> [env assignments...]
out=gpurun_out/${1:-r03b}; shift; rm -rf $out/prof; mkdir -p $out
for kv in "$@"; do export "$kv"; done
export RT_PIPELINE=${RT_PIPELINE:-wf}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RT_EXP_RANKS="" RT_EXP_REPS=2 RT_EXP_ROUNDS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python3 $R/tools/exp_kernels.py "arm:" > $R/$out/prof.log 2>&1
tail -3 $R/$out/prof.log
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/$out/kernel_stats.csv; cat $R/$out/kernel_stats.csv | cut -c1-200
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$R/$out/prof/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(fs[0])))
# per launch sequence of the LAST frame: print name + duration in order
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=[i for i,r in enumerate(rows) if "camera" in r["Kernel_Name"]]
if last:
    seq=rows[last[-1]:]
    t0=int(seq[0]["Start_Timestamp"])
    for r in seq[:40]:
        n=r["Kernel_Name"][:60]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        print(f"{(s-t0)/1e3:10.1f} us  +{(e-s)/1e3:9.1f} us  {n}  vgpr={r.get('VGPR_Count')} scratch={r.get('Scratch_Size')}")
PY

#!/bin/bash
# tools/gpu/refresh_tables.sh <tag>: the numbers DESIGN.md / README.md quote, in one gpurun call -> gpurun_out/<tag>/
#   bench.json  configs.md  boundary.md  ranks{2,4,8}.txt  overlap.txt
set -u
TAG=${1:-tables}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 500 python tools/run_configs.py > $OUT/configs.md 2> $OUT/configs.err || { tail -5 $OUT/configs.err; exit 1; }
timeout -k 10 200 python tools/boundary.py > $OUT/boundary.md 2> $OUT/boundary.err || { tail -5 $OUT/boundary.err; exit 1; }
for w in 2 4 8; do timeout -k 10 200 python tools/exp_ranks.py $w > $OUT/ranks$w.txt 2>&1 || exit 1; done
timeout -k 10 200 python tools/exp_overlap.py > $OUT/overlap.txt 2>&1 || exit 1
grep -h "^rank\|^world" $OUT/ranks*.txt $OUT/overlap.txt

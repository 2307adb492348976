#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>_*/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-bvh-compare ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- $BENCH --steps ${STEPS:-9} --warmup 1 > $OUT/prof_${TAG}_stats.json 2> $OUT/prof_${TAG}_stats.err || echo "stats pass failed"
[ -n "${ONLY_STATS:-}" ] && exit 0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_pmc_$N -- $BENCH --steps 1 --warmup 0 > $OUT/prof_${TAG}_pmc_$N.json 2> $OUT/prof_${TAG}_pmc_$N.err || echo "pmc pass $N failed"
done
rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1 || true
find $OUT -name "*.csv" | head -50

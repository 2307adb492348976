#!/bin/bash
# tools/gpu/r04_final.sh <tag>: the round's committed measurements in one gpurun call: boundary table (tools/boundary.py), all
# five BASELINE configs with the CPU beside them (tools/run_configs.py), a bench line, rocprofv3 stats + PMC passes.
tag=${1:-r04}; out=gpurun_out/$tag; mkdir -p $out
make -C examples > /dev/null 2>&1
timeout -k 10 300 python tools/boundary.py 12 > $out/boundary.md 2> $out/boundary.err || { tail -5 $out/boundary.err; exit 1; }
cat $out/boundary.md
timeout -k 10 500 python tools/run_configs.py > $out/configs.md 2> $out/configs.err || { tail -5 $out/configs.err; exit 1; }
cat $out/configs.md
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
STEPS=9 bash tools/profile_gpu.sh $tag > $out/profile.log 2>&1
tail -2 $out/profile.log

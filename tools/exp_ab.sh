#!/bin/bash
# A/B two builds of librt_hip.so on the same GPU box, interleaved: tools/exp_ab.sh <base.so> [config]
# (build the baseline from a commit with `git archive <rev> raytracing_c_amd/csrc include | tar -x -C /tmp/base`).
base=$1; cfg=${2:-helmet}
for i in 1 2 3; do
  for lib in "$base" raytracing_c_amd/librt_hip.so; do
    echo "== $lib"
    RT_LIB_PATH=$(realpath $lib) RT_EXP=${RT_EXP:-auto} timeout -k 10 200 python tools/exp_runtime.py $cfg 2>&1 | grep -v "^$" || exit 1
  done
done

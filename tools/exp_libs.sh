#!/bin/bash
# A/B of differently built librt_hip.so files on one box: tools/exp_libs.sh <log> <name>...   (tools/exp/librt_<name>.so;
# "hip" = the in-tree build).  Two rounds, arms interleaved, one process per arm and round.
log=$1; shift
: > "$log"
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = hip ]; then lib=raytracing_c_amd/librt_hip.so; else lib=tools/exp/librt_$v.so; fi
    RT_LIB_PATH=$lib RT_EXP_ROUNDS=1 timeout -k 10 300 python tools/exp_kernels.py "$v:" 2>/dev/null | grep round >> "$log" || exit 1
  done
done
cat "$log"

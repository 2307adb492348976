#!/bin/bash
# tools/gpu/ab_libs.sh <tag> <lib name>...  (+ env RT_PIPELINE etc. exported by the caller): interleaved A/B of library builds
out=gpurun_out/$1; shift; mkdir -p $out
RT_EXP_RANKS="${RT_EXP_RANKS-}" RT_EXP_REPS=${RT_EXP_REPS:-3} bash tools/exp_libs.sh $out/ab.log "$@"

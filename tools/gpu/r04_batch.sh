#!/bin/bash
# tools/gpu/r04_batch.sh <tag>: one gpurun call = the -m gpu suite (all of it, no -x), the contract A/B (librt_hip_v1.so vs the
# product), a bench line, and the rocprofv3 stats + PMC passes.  Every step logs under gpurun_out/<tag>/; a failing step does
# not stop the later ones unless it timed out (then nothing more touches the GPU).
tag=${1:-r04}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 ${PYTEST_TIMEOUT:-700} python -m pytest -q -m gpu tests > $out/pytest.log 2>&1; rc=$?
tail -15 $out/pytest.log
[ $rc -ge 124 ] && exit $rc
if [ -z "${SKIP_AB:-}" ]; then
  mkdir -p tools/exp && cp raytracing_c_amd/librt_hip_v1.so tools/exp/librt_v1.so
  RT_EXP_RANKS="" RT_EXP_REPS=3 bash tools/exp_libs.sh $out/ab.log v1 hip || exit 1
fi
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
if [ -z "${SKIP_PROF:-}" ]; then
  STEPS=9 bash tools/profile_gpu.sh $tag > $out/profile.log 2>&1
  tail -3 $out/profile.log
fi
exit $rc

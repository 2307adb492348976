#!/bin/bash
# tools/gpu/tests_and_ab.sh <tag> <lib name>...: the -m gpu suite on the tree's build, then an interleaved A/B of library builds
# (tools/exp/librt_<name>.so; "hip" = the tree's), three repetitions per arm and round, the headline frame only.
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout -k 10 ${PYTEST_TIMEOUT:-700} python -m pytest -q -m gpu tests > $out/pytest.log 2>&1; rc=$?
  tail -12 $out/pytest.log
  [ $rc -ge 124 ] && exit $rc
fi
RT_EXP_RANKS="" RT_EXP_REPS=${RT_EXP_REPS:-3} bash tools/exp_libs.sh $out/ab.log "$@"

#!/bin/bash
# tools/gpu/pytest_gpu.sh <tag> [pytest args...]: the -m gpu tests on the box, log under gpurun_out/<tag>/
out=gpurun_out/$1; shift; mkdir -p $out
timeout -k 10 ${PYTEST_TIMEOUT:-1000} python -m pytest -x -q -m gpu "$@" > $out/pytest.log 2>&1
rc=$?
tail -25 $out/pytest.log
exit $rc

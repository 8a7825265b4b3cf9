#!/bin/bash
# Closing run of round 4, part A (GPU box): the full -m gpu suite once, smoke(), the bench lines with live counters
# (c4 default, c2, c3, c5) and the rocprofv3 --kernel-trace --stats summaries of c4 / c5.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r04}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/${TAG}_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
TAG=$TAG CONFIGS="c4 c2 c3 c5" STATS_CONFIGS="c4 c5" bash tools/round_profiles.sh

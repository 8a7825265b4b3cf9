#!/bin/bash
# Closing run of a round, part A (GPU box): the full -m gpu suite, smoke(), the bench lines with live counters and the
# rocprofv3 kernel-trace summaries (tools/round_profiles.sh).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
TAG=$TAG bash tools/round_profiles.sh

#!/bin/bash
# Development check of a pair-kernel change (GPU box): parity suites, tie sweep, yeast, a short c4 line.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/check_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/check_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/tie_sweep.py $1 2>&1 | grep distinct
timeout -k 10 100 python tools/yeast_time.py $1 2>&1 | tail -2
timeout -k 10 100 python tools/quick_time.py c4 2>&1 | tail -2

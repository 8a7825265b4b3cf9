#!/bin/bash
# Development check for the half-wave kernels of 11 .. 15 words per lane (GPU box): parity on long tied columns, tie sweeps.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/check_long.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/check_long.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_gpu.py 3000 341 big 2>&1 | grep "done"
for nf in 30000 20000; do echo "== N_FEAT=$nf N_SAMP=128"; N_FEAT=$nf N_SAMP=128 timeout -k 10 200 python tools/tie_sweep.py 2>&1 | grep distinct; done

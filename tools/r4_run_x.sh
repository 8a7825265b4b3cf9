set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4x_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4x_parity.log
[ $rc -eq 0 ] || exit $rc
for p in "" "solo=0"; do echo "== $p"; timeout -k 10 120 python tools/tie_sweep.py "$p" 2>&1 | grep distinct; done > gpurun_out/r4x_tie.log 2>&1
cat gpurun_out/r4x_tie.log
timeout -k 10 100 python tools/yeast_time.py 2>&1 | tail -2

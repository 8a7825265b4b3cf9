set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r4s_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4s_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/k0_time.py > gpurun_out/r4s_k0.log 2>&1; cat gpurun_out/r4s_k0.log

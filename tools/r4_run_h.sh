set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4h_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4h_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/k0_time.py > gpurun_out/r4h_k0.log 2>&1; echo "k0 rc=$?"; cat gpurun_out/r4h_k0.log
echo done

set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tie or tied or joint or heavy" > gpurun_out/r4m_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4m_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/tie_sweep.py $1 > gpurun_out/r4m_tie.log 2>&1; echo "tie rc=$?"; cat gpurun_out/r4m_tie.log
echo done

set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r4o_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4o_parity.log
[ $rc -eq 0 ] || exit $rc
for p in "" "solo=0" "tgmax=1200" "tgmax=1200,solo=0"; do echo "== $p"; timeout -k 10 120 python tools/tie_sweep.py "$p" 2>&1 | grep distinct; done > gpurun_out/r4o_tie_plans.log 2>&1
cat gpurun_out/r4o_tie_plans.log

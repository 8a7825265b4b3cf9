set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r4g_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4g_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/step_stats.py > gpurun_out/r4g_step_stats.md 2> gpurun_out/r4g_step_stats.err; echo "step stats rc=$?"
CONFIG=c5 N_SAMP=512 TAG=r4g_stall_c5 timeout -k 10 500 bash tools/pmc_stall.sh > gpurun_out/r4g_stall_c5.json 2> gpurun_out/r4g_stall_c5.err; echo "stall rc=$?"
echo done

set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/pipe_probe.py 2>&1 | grep -v "amdgpu\|K1 plan"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3_gpu.log
python tools/host_path_time.py 2>&1 | grep "icikt_pairs"

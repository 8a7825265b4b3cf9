set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3_gpu.log

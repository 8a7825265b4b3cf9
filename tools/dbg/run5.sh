set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/n_sweep.py 20000 30000 36000 50000 65535 > gpurun_out/s3_n_sweep.log 2>&1; cat gpurun_out/s3_n_sweep.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3_gpu.log

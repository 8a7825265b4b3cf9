set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s3_gpu.log
python tools/tie_sweep.py > gpurun_out/s3_tie_sweep.log 2>&1 && cat gpurun_out/s3_tie_sweep.log
python tools/yeast_time.py > gpurun_out/s3_yeast.log 2>&1 && tail -2 gpurun_out/s3_yeast.log
python tools/n_sweep.py 10000 12000 14272 16000 18336 > gpurun_out/s3_n_sweep.log 2>&1 && cat gpurun_out/s3_n_sweep.log

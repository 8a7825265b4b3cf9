set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3_gpu.log
python tools/n_sweep.py 18336 20000 26000 30000 31000 > gpurun_out/s3_n_sweep.log 2>&1; grep -v amdgpu gpurun_out/s3_n_sweep.log | cut -c1-100
N_FEAT=30000 N_SAMP=128 python tools/tie_sweep.py 2>&1 | grep -v amdgpu
N_FEAT=30000 N_SAMP=128 python tools/tie_sweep.py half=0 2>&1 | grep -v amdgpu

set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PYTEST_ADDOPTS="--capture=sys" timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3_gpu.log
python tools/host_path_time.py 2>&1 | grep -v "amdgpu\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl"
python bench.py --pmc off --cpu-sample 0 > gpurun_out/s3_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s3_bench.json')); print('c4', d['value'], d['pcie_inclusive'], d.get('e2e_ici_kendalltau_ms'))"
python bench.py --config c3 --pmc off --cpu-sample 0 > gpurun_out/s3_bench3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s3_bench3.json')); print('c3', d['value'], d['pcie_inclusive'], d.get('e2e_ici_kendalltau_ms'))"

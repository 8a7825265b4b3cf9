set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r4i_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4i_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/tie_sweep.py > gpurun_out/r4i_tie.log 2>&1; echo "tie rc=$?"; cat gpurun_out/r4i_tie.log
timeout -k 10 100 python bench.py --config c2 --steps 20 --warmup 5 > gpurun_out/r4i_c2.json 2> gpurun_out/r4i_c2.err; echo "c2 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r4i_c2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
timeout -k 10 100 python bench.py --config c4 --steps 10 --warmup 3 > gpurun_out/r4i_c4.json 2> gpurun_out/r4i_c4.err; echo "c4 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r4i_c4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
echo done

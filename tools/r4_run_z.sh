cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 100 python tools/yeast_time.py 2>&1 | tail -3
timeout -k 10 100 python tools/yeast_time.py "tgmax=128" 2>&1 | tail -2
timeout -k 10 120 python tools/tie_sweep.py 2>&1 | grep distinct
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2

#!/bin/bash
# Randomised sweeps on the build with task segments (GPU box): HIP path vs the C oracle; split = 1 / 2 / 4 forced at random.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "## $*"; timeout -k 10 ${T:-700} "$@" 2>&1 | grep -v amdgpu | grep "done\|FAIL\|failures\|cases ok" | tail -3; }
run python tools/fuzz_gpu.py 40000 351 r4
run python tools/fuzz_gpu.py 10000 352 mid
run python tools/fuzz_gpu.py 6000 353 big
run python tools/fuzz_matrix.py 3000 451
run python tools/fuzz_matrix.py 1000 551 multi2
run python tools/fuzz_pipe.py 400 652
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29577 tools/fuzz_dist.py 400 752

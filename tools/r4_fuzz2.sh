#!/bin/bash
# Randomised sweeps on the build with step records / count mode / SOLO steps (GPU box): HIP path vs the C oracle.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "## $*"; timeout -k 10 ${T:-500} "$@" 2>&1 | grep -v amdgpu | grep "done\|FAIL\|failures\|cases ok" | tail -3; }
run python tools/fuzz_gpu.py 30000 321 r4
run python tools/fuzz_gpu.py 6000 322 mid
run python tools/fuzz_gpu.py 4000 323 big
run python tools/fuzz_matrix.py 3000 421
run python tools/fuzz_matrix.py 1000 521 multi2
run python tools/fuzz_pipe.py 600 622
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29577 tools/fuzz_dist.py 500 722

#!/bin/bash
# Randomised sweeps on the final build of round 4 (GPU box): HIP path vs the C oracle.  Prints one summary line per sweep.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "## $*"; timeout -k 10 ${T:-400} "$@" 2>&1 | grep -v amdgpu | grep "done\|FAIL\|failures\|cases ok" | tail -3; }
run python tools/fuzz_gpu.py 24000 311 r4
run python tools/fuzz_gpu.py 6000 312 mid
run python tools/fuzz_gpu.py 6000 313 big
run python tools/fuzz_gpu.py 12000 314
run python tools/fuzz_matrix.py 5000 411
run python tools/fuzz_matrix.py 1500 511 multi2
run python tools/fuzz_wide.py 80 412
run python tools/fuzz_pipe.py 1200 612
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29577 tools/fuzz_dist.py 800 712

#!/bin/bash
# Randomised sweeps on the last build of round 4 (GPU box): HIP path vs the C oracle, after count mode in the whole-wave kernels.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "## $*"; timeout -k 10 ${T:-500} "$@" 2>&1 | grep -v amdgpu | grep "done\|FAIL\|failures\|cases ok" | tail -3; }
run python tools/fuzz_gpu.py 1200 971 big
run python tools/fuzz_gpu.py 800 972 mid
run python tools/fuzz_gpu.py 14000 973 r4
run python tools/fuzz_gpu.py 50 974 fam
run python tools/fuzz_matrix.py 1500 975
run python tools/fuzz_pipe.py 300 976
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29577 tools/fuzz_dist.py 300 977

#!/bin/bash
# A longer randomised sweep on the final build (GPU box): HIP path vs the C oracle, other seeds than tools/r4_fuzz2.sh.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "## $*"; timeout -k 10 ${T:-700} "$@" 2>&1 | grep -v amdgpu | grep "done\|FAIL\|failures\|cases ok" | tail -3; }
run python tools/fuzz_gpu.py 60000 331 r4
run python tools/fuzz_gpu.py 15000 332 mid
run python tools/fuzz_gpu.py 10000 333 big
run python tools/fuzz_matrix.py 4000 431

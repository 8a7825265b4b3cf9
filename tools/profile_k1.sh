#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for bench.py's workload.
# Outputs under gpurun_out/prof_*; summaries are copied into profiles/ by tools/summarize_profile.py.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-5}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps $STEPS --warmup 2 --cpu-sample 0 > gpurun_out/prof_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_sq -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_sq.log 2>&1 || exit 1
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq -name "*.csv" | head -40

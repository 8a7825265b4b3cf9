#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for bench.py's workload.
# Outputs under gpurun_out/prof_*; summaries are copied into profiles/ by tools/summarize_profile.py.
# Few counters per pass (a request the hardware cannot schedule aborts rocprofv3), each pass under its own timeout.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-5}
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps $STEPS --warmup 2 --cpu-sample 0 --no-extras > gpurun_out/prof_stats.log 2>&1 || exit 1
pass() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; exit 1; }; }
pass prof_fetch FETCH_SIZE
pass prof_write WRITE_SIZE
pass prof_sq SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass prof_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD
pass prof_ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass prof_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass prof_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find gpurun_out/prof_* -name "*.csv" | head -40

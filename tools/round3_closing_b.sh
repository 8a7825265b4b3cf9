#!/bin/bash
# Closing run of a round, part B (GPU box): sweeps, host path, in-library multi-rank line, step kinds on tied data.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
mkdir -p gpurun_out
python tools/tie_sweep.py > gpurun_out/${TAG}_tie_sweep.log 2>&1; echo "tie rc=$?"
python tools/yeast_time.py > gpurun_out/${TAG}_yeast.log 2>&1; echo "yeast rc=$?"
python tools/n_sweep.py 500 2000 5000 10000 12000 14272 16000 18336 20000 30000 36000 50000 60000 65535 > gpurun_out/${TAG}_n_sweep.log 2>&1; echo "n rc=$?"
python tools/na_sweep.py > gpurun_out/${TAG}_na_sweep.log 2>&1; echo "na rc=$?"
python tools/host_path_time.py > gpurun_out/${TAG}_host_path.log 2>&1; echo "host rc=$?"
python bench.py --launcher inlib > gpurun_out/${TAG}_bench_inlib1.json 2> gpurun_out/${TAG}_bench_inlib1.err; echo "inlib rc=$?"
python tools/step_stats.py > gpurun_out/${TAG}_step_stats.md 2>&1; echo "stats rc=$?"
bash tools/pmc_tie.sh 0 5000 1000 200 50 10 > gpurun_out/${TAG}_pmc_tie.log 2>&1; echo "pmc tie rc=$?"
python tools/tie_kinds.py gpurun_out/${TAG}_step_stats.md gpurun_out/${TAG}_pmc_tie.log > gpurun_out/${TAG}_tie_kinds.md 2>&1; echo "kinds rc=$?"; cat gpurun_out/${TAG}_tie_kinds.md

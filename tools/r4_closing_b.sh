#!/bin/bash
# Closing run of round 4, part B (GPU box): sweeps, host path, in-library line, K0 alone, step kinds and counters on tied data.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r04}
mkdir -p gpurun_out
python tools/tie_sweep.py > gpurun_out/${TAG}_tie_sweep.log 2>&1; echo "tie rc=$?"
N_FEAT=30000 N_SAMP=128 python tools/tie_sweep.py > gpurun_out/${TAG}_tie_sweep_30000.log 2>&1; echo "tie30000 rc=$?"
python tools/yeast_time.py > gpurun_out/${TAG}_yeast.log 2>&1; echo "yeast rc=$?"
python tools/k0_time.py > gpurun_out/${TAG}_k0.log 2>&1; echo "k0 rc=$?"
python tools/n_sweep.py 500 2000 5000 10000 12000 14272 16000 18336 20000 30000 36000 50000 60000 65535 > gpurun_out/${TAG}_n_sweep.log 2>&1; echo "n rc=$?"
python tools/na_sweep.py > gpurun_out/${TAG}_na_sweep.log 2>&1; echo "na rc=$?"
python tools/host_path_time.py > gpurun_out/${TAG}_host_path.log 2>&1; echo "host rc=$?"
python tools/ab_pipe_k0.py > gpurun_out/${TAG}_ab_pipe_k0.log 2>&1; echo "ab pipe rc=$?"
python bench.py --launcher inlib > gpurun_out/${TAG}_bench_inlib1.json 2> gpurun_out/${TAG}_bench_inlib1.err; echo "inlib rc=$?"
python tools/balance_probe.py 4 > gpurun_out/${TAG}_balance_probe.log 2>&1; echo "balance rc=$?"
bash tools/pmc_tie.sh 0 5000 1000 200 50 10 > gpurun_out/${TAG}_pmc_tie.log 2>&1; echo "pmc tie rc=$?"
CONFIG=c5 N_SAMP=512 TAG=${TAG}_stall_c5 bash tools/pmc_stall.sh > gpurun_out/${TAG}_stall_c5.json 2> gpurun_out/${TAG}_stall_c5.err; echo "stall rc=$?"

cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for nf in 50000 30000 20000; do echo "== N_FEAT=$nf N_SAMP=128"; N_FEAT=$nf N_SAMP=128 timeout -k 10 200 python tools/tie_sweep.py 2>&1 | grep distinct; done > gpurun_out/r4t_tie_long.log 2>&1
cat gpurun_out/r4t_tie_long.log

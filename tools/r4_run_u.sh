cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for p in "waves=12" "waves=8"; do echo "== N_FEAT=30000 N_SAMP=128 $p"; N_FEAT=30000 N_SAMP=128 timeout -k 10 200 python tools/tie_sweep.py "$p,verbose=1" 2>&1 | grep "distinct\|K1 plan" | sort -u; done > gpurun_out/r4u_tie_long.log 2>&1
cat gpurun_out/r4u_tie_long.log

cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for p in "list=32" "list=8" "tgmax=128"; do echo "== $p"; timeout -k 10 120 python tools/tie_sweep.py "$p" 2>&1 | grep distinct; done > gpurun_out/r4n_tie_plans.log 2>&1
cat gpurun_out/r4n_tie_plans.log

cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for p in "waves=20" "waves=20,solo=0" "waves=16" "waves=16,solo=0"; do echo "== $p"; timeout -k 10 120 python tools/tie_sweep.py "$p,verbose=1" 2>&1 | grep "distinct\|K1 plan" | sort -u | grep -v "^\[icikt\] K1 plan.*0 tie-group" ; done > gpurun_out/r4p_tie_plans.log 2>&1
cat gpurun_out/r4p_tie_plans.log

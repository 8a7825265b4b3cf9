cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for p in "verbose=1" "solo=0" "tgmax=128" "tgmax=-1" "list=0" ; do echo "== $p"; timeout -k 10 120 python tools/yeast_time.py "$p" 2>&1 | grep "K1\|K1 plan" | sort -u | tail -4; done > gpurun_out/r4q_yeast_plans.log 2>&1
cat gpurun_out/r4q_yeast_plans.log

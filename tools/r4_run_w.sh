cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for lib in "" tools/libicikt_cnt32.so; do
  echo "== lib=${lib:-product}"
  ICIKT_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python tools/tie_sweep.py 2>&1 | grep distinct
done > gpurun_out/r4w_cnt32.log 2>&1
cat gpurun_out/r4w_cnt32.log

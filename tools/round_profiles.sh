#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the round's bench lines with live counters (bench.py takes them itself: child
# rocprofv3 --pmc runs) and the rocprofv3 --kernel-trace --stats summaries of the same commands.  Everything lands in
# gpurun_out/$TAG_*; copy what is to be judged into profiles/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
mkdir -p gpurun_out
for cfg in ${CONFIGS:-c4 c3 c5}; do
  python3 bench.py --config $cfg --save-pmc > gpurun_out/${TAG}_bench_$cfg.json 2> gpurun_out/${TAG}_bench_$cfg.err || { echo "bench $cfg failed"; tail -5 gpurun_out/${TAG}_bench_$cfg.err; exit 1; }
  echo "bench $cfg done"
done
cp profiles/k1_pmc_records.json gpurun_out/${TAG}_k1_pmc_records.json
for cfg in ${STATS_CONFIGS:-c4 c5}; do
  steps=5; [ $cfg = c5 ] && steps=2
  rm -rf gpurun_out/${TAG}_stats_$cfg
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/${TAG}_stats_$cfg" -- python3 "$OLDPWD/bench.py" --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --no-extras --pmc off > "$OLDPWD/gpurun_out/${TAG}_stats_$cfg.log" 2>&1) || { echo "stats $cfg failed"; tail -5 gpurun_out/${TAG}_stats_$cfg.log; exit 1; }
  f=$(find gpurun_out/${TAG}_stats_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_$cfg.csv
  echo "stats $cfg done"
done
ls gpurun_out | grep "^${TAG}_" | head -30

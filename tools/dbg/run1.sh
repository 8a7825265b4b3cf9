set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/tie_sweep.py > gpurun_out/s3_tie_sweep.log 2>&1 && echo tie ok
python tools/yeast_time.py > gpurun_out/s3_yeast.log 2>&1 && echo yeast ok
python tools/n_sweep.py 10000 20000 30000 36000 50000 65535 > gpurun_out/s3_n_sweep.log 2>&1 && echo nsweep ok
python tools/step_stats.py > gpurun_out/s3_step_stats.md 2>&1 && echo stats ok
python bench.py --config c5 --steps 3 --warmup 1 --pmc off --no-extras > gpurun_out/s3_bench_c5.json 2> gpurun_out/s3_bench_c5.err && echo c5 ok

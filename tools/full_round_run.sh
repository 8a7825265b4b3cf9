set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/full_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"
python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "c3 rc=$?"
python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "c5 rc=$?"
python bench.py --launcher inlib > gpurun_out/bench_inlib1.json 2> gpurun_out/bench_inlib1.err; echo "inlib rc=$?"
bash tools/profile_k1.sh > gpurun_out/profile.log 2>&1; echo "profile rc=$?"
python tools/tie_sweep.py > gpurun_out/tie_sweep.log 2>&1
python tools/yeast_time.py > gpurun_out/yeast.log 2>&1
python tools/n_sweep.py 500 2000 5000 10000 12000 13000 14272 16000 20000 30000 50000 65535 > gpurun_out/n_sweep.log 2>&1
python tools/na_sweep.py > gpurun_out/na_sweep.log 2>&1
python tools/host_path_time.py > gpurun_out/host_path.log 2>&1
echo done

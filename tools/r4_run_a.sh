# round 4, first GPU call: the whole -m gpu suite once on the new host path, then the bench lines (c4 default, c2).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r4a_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4a_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r4a_bench_c4.json 2> gpurun_out/r4a_bench_c4.err; echo "c4 rc=$?"
timeout -k 10 300 python bench.py --config c2 > gpurun_out/r4a_bench_c2.json 2> gpurun_out/r4a_bench_c2.err; echo "c2 rc=$?"
echo done

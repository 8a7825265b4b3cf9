#!/bin/bash
# L2 / fabric counters of K1 on the full c5 matrix for two launch plans (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export N_FEAT=50000 N_SAMP=${N_SAMP:-2048} REPS=1 MAX_PAIRS=${MAX_PAIRS:-1000000000}
pass() { name=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- python3 tools/run_k1_once.py > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; exit 1; }; }
[ $# -gt 0 ] || set -- pend=g pend=l
for P in "$@"; do
  export PLAN="$P"; tag=$(echo $P | tr '=,' '__')
  pass c5_${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass c5_${tag}_fetch FETCH_SIZE
  pass c5_${tag}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  pass c5_${tag}_sq SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/c5_*")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

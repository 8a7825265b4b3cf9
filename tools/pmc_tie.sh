#!/bin/bash
# SQ counters of K1 on tied data (run on the GPU box): LEVELS distinct values per column
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
pass() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- python3 tools/run_tie_once.py > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; exit 1; }; }
for L in "$@"; do
  export LEVELS=$L
  pass tie${L}_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES
  pass tie${L}_b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/tie*_[ab]")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

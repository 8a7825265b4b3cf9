#!/bin/bash
# Pipeline counter passes of K1 on a long-column workload (run on the GPU box): N_FEAT x N_SAMP, MAX_PAIRS pairs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export N_FEAT=${N_FEAT:-50000} N_SAMP=${N_SAMP:-512} MAX_PAIRS=${MAX_PAIRS:-60000} REPS=2
pass() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- python3 tools/run_k1_once.py > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; exit 1; }; }
pass pl_a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES
pass pl_b TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass pl_c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pl_[abc]")):
    fs = sorted(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]
    for f in fs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.environ.get("N_FEAT"), d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

#!/bin/bash
# SQ counter passes of K1 on a long-column workload (run on the GPU box): N_FEAT x N_SAMP, MAX_PAIRS pairs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export N_FEAT=${N_FEAT:-50000} N_SAMP=${N_SAMP:-512} MAX_PAIRS=${MAX_PAIRS:-60000} REPS=2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pl_stats -- python3 tools/run_k1_once.py > gpurun_out/pl_stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pl_a -- python3 tools/run_k1_once.py > gpurun_out/pl_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pl_b -- python3 tools/run_k1_once.py > gpurun_out/pl_b.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pl_c -- python3 tools/run_k1_once.py > gpurun_out/pl_c.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pl_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
for d in sorted(glob.glob("gpurun_out/pl_[abc]")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

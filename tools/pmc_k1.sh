#!/bin/bash
# SQ counter passes for K1 (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for NP in ${NPS:-2}; do
  export PLAN="np=$NP,wpb=${WPB:-4}"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_a_np$NP -- python3 tools/run_k1_once.py > gpurun_out/pmc_a_np$NP.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_b_np$NP -- python3 tools/run_k1_once.py > gpurun_out/pmc_b_np$NP.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d gpurun_out/pmc_c_np$NP -- python3 tools/run_k1_once.py > gpurun_out/pmc_c_np$NP.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_*_np*")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

#!/bin/bash
# Where the waves of the pair kernel wait: SQ wait / active counters, LDS queue and conflict counters, TA / TCP stall
# counters of K1 for one bench workload (CONFIG=c4|c5, N_SAMP to shorten), each group in a pass of its own (run on the GPU box).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export CONFIG=${CONFIG:-c5} REPS=1
TAG=${TAG:-stall_$CONFIG}
pass() { name=$1; shift; timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$TAG/$name -- python3 tools/run_k1_once.py > gpurun_out/$TAG/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$TAG/$name.log; }; }
mkdir -p gpurun_out/$TAG
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS
pass c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
pass d SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass e SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
pass f SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# (TA_* / TCP_* stall counters are NOT collected: on this pool rocprofv3 reports "incomplete dispatches" for them and never
#  returns -- three such passes cost 15 GPU-minutes before they were killed)
pass j GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_LEVEL_WAVES
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys, json
tag = sys.argv[1]
res = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k1_pairs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
for f in sorted(glob.glob(f"gpurun_out/{tag}/*.log")):
    for line in open(f):
        if line.startswith("{"):
            res.setdefault("k1_ms", []).append(json.loads(line)["k1_ms_per_launch"])
print(json.dumps(res, indent=1))
PY

#!/bin/bash
# Memory-pipeline counter passes for K1 (run on the GPU box): is the gather path (TA / TCP / L2) the limiter?
# Few counters per pass (a request the hardware cannot schedule aborts rocprofv3), each pass under its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
pass() { name=$1; shift; timeout -k 5 90 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- python3 tools/run_k1_once.py > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; exit 1; }; }
pass pm_a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass pm_b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass pm_c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass pm_d TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass pm_e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pm_[abcde]")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

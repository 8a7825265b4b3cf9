#!/usr/bin/env python3
"""bench.py -- all-pairs ICI-Kendall-tau throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]
    python bench.py --gpus N --launcher inlib     one process, N GPUs behind ONE C call (icikt_pairs_multi_f64)

`python bench.py --gpus N` runs AS TYPED for N = 1, 2, 4, 8: with N > 1 and no WORLD_SIZE in the environment this
process -- before it imports torch or touches a GPU -- starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process (never an
exec), relays its one JSON line and exits with its status.  Started under torch.distributed.run directly (WORLD_SIZE set,
as the driver does for N > 1) it is a rank.  The timed region is the same at every N, so the N = 1 line equals BENCH.

Default workload = BASELINE.md c4, the configuration the metric is quoted on: synthetic 10 000 features x
1 024 samples, numpy default_rng(4).standard_normal, per column the 1 000 smallest values missing
(10 % left-censored; tie group < 1024 so the reference's int32 tie sums do not wrap),
perspective = "global": 523 776 column pairs, each a pair of length-10 000 vectors.
--config c3: 10 000 x 256, 500 smallest missing (32 640 pairs).  --config c5: 50 000 x 2 048, 1 000 smallest
missing, 2 096 128 pairs in BOTH perspectives ("local" is the epilogue over the same pair counts), plus the
include_only subset (first 64 names: 128 992 pairs) and pairwise_completeness on that subset, timed beside it.
--config c2: the yeast matrix of the reference (data/yeast_missing.rda as the fixture tests/golden/yeast_missing.npz:
6 887 x 96, zeros -> missing), perspective = "global", all 4 560 pairs; the CPU leg runs ALL of them (no sampling).

One "step" = one full pass of the hot path over the matrix, input already resident in HBM:
  K0 per-column pre-pass (N > 1: each rank sorts S/N columns, then one RCCL all-gather of the prepared state;
  all ranks together fall back to sorting all columns if that cannot be set up) -> K1 pair kernel over this rank's
  contiguous block of the combn-ordered pair list (the reference's `core` chunks, R/kendalltau.R:250-255) -> K2
  epilogue -> (N > 1) RCCL gather of the P/N x 4 results to rank 0.
value = pairs of the whole job / wall time (max over ranks); total work is fixed, so scaling = strong.
Beside it, timed the same way (K steps, mean): `pcie_inclusive` -- the host-buffer entry icikt_pairs_f64, i.e. the metric
as SURVEY.md section 8(d) words it (H2D of the matrix, kernels, D2H of the results) -- and `full_matrix_wall_ms` -- the
icikt_matrix_f64 call that returns the five S x S matrices of ici_kendalltau().
Exit status is non-zero when the results disagree with the oracle sample (a wrong line is not a bench line).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
# (torch is imported inside main(): the self-launching parent must not load it)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ATOL = 1e-10

CONFIGS = {
    "c2": dict(fixture="yeast_missing.npz", n_feat=6887, n_samp=96, n_na=0, seed=0, steps=50, warmup=10, cpu_sample=4560, pre_warm=30),
    "c3": dict(n_feat=10000, n_samp=256, n_na=500, seed=3, steps=20, warmup=5, cpu_sample=6000, pre_warm=8),
    "c4": dict(n_feat=10000, n_samp=1024, n_na=1000, seed=4, steps=20, warmup=5, cpu_sample=12000, pre_warm=5),
    "c5": dict(n_feat=50000, n_samp=2048, n_na=1000, seed=5, steps=3, warmup=1, cpu_sample=1200),
    # not a BASELINE configuration: c4's shape with TIED data (values rounded to ~1 000 distinct levels per column, central tie
    # groups of ~23 rows), the regime the package's count data live in -- every step of the pair kernel is a tie step
    "c4t": dict(n_feat=10000, n_samp=1024, n_na=1000, seed=4, levels=1000, steps=10, warmup=3, cpu_sample=6000, pre_warm=5),
}


def make_matrix(n_feat: int, n_samp: int, n_na: int, seed: int, levels: int = 0) -> np.ndarray:
    """BASELINE.md synthetic config: column-major float64, per column the n_na smallest -> NaN.  levels > 0: the values
    rounded to about that many distinct levels per column (tools/tie_sweep.py's tied data)."""
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n_feat, n_samp)))
    if levels:
        X = np.asfortranarray(np.round(X * (levels / 6.0)))
    if n_na:
        idx = np.argpartition(X, n_na, axis=0)[:n_na]
        np.put_along_axis(X, idx, np.nan, axis=0)
    return X


def workload_matrix(config: str, cfg: dict) -> np.ndarray:
    """The matrix of a configuration as the pair entries take it (column-major float64, NaN = missing).  c2 is the
    reference's own yeast data (a committed fixture decoded from data/yeast_missing.rda by tests/golden/make_golden.py),
    zeros -> missing as global_na = c(NA, Inf, 0) makes them (R/utils.R:1-23)."""
    if CONFIGS[config].get("fixture"):
        if any(cfg[k] != CONFIGS[config][k] for k in ("n_feat", "n_samp", "n_na", "seed")):
            raise SystemExit(f"--config {config} is a fixed data set: --n-feat / --n-samp / --n-na / --seed do not apply")
        z = np.load(os.path.join(ROOT, "tests", "golden", CONFIGS[config]["fixture"]))
        X = np.asfortranarray(z["X"].astype(np.float64))
        X[(X == 0) | ~np.isfinite(X)] = np.nan
        return X
    return make_matrix(cfg["n_feat"], cfg["n_samp"], cfg["n_na"], cfg["seed"], cfg.get("levels", 0))


# ---- self-launch: `python bench.py --gpus N` as typed ----------------------------------------------------------------
def should_spawn(gpus: int, launcher: str, env) -> bool:
    """True when this process is the PARENT of a torch.distributed run: no WORLD_SIZE yet, and either several GPUs are
    asked for or the one-rank rehearsal of the distributed sequence (ICIKT_BENCH_FORCE_DIST=1)."""
    if launcher != "torchrun" or env.get("WORLD_SIZE"):
        return False
    return gpus > 1 or env.get("ICIKT_BENCH_FORCE_DIST") == "1"


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def child_command(argv, gpus: int, port: int):
    """The command line of the child: one rank per GPU of ONE node over torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.join(ROOT, "bench.py"), *argv]


def spawn_ranks(argv, gpus: int) -> int:
    """Parent: start the ranks as a child process, relay the JSON line (everything else the ranks print goes to stderr)
    and return the child's exit status.  Nothing here imports torch or touches a GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = child_command(argv, gpus, free_port())
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        try:
            d = json.loads(line)
            d["launched_by"] = {"parent": "bench.py (self-launch)", "cmd": " ".join(cmd[1:9]) + " bench.py ...",
                                "child_wall_s": time.perf_counter() - t0, "child_rc": rc}
            line = json.dumps(d)
        except Exception:
            pass
        print(line, flush=True)
    elif rc == 0:
        rc = 3   # no line is not a result
    return rc


def host_cores() -> int:
    """Cores this process may actually use: the cgroup CPU quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    env = os.environ.get("ICIKT_CPU_BASELINE_CORES")
    return int(env) if env else n


def cpu_baseline(X: np.ndarray, P_total: int, sample_pairs: int, perspective="global", seed: int = 0):
    """The CPU restatement (oracle/, kind = "port") on this box's host cores over a bounded random
    sample of the same pair list; one C thread per core (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    O.lib()
    n, S = X.shape
    rng = np.random.default_rng(seed)
    iu, ju = np.triu_indices(S, k=1)
    # (every pair, in order, when the sample covers the list: c2 runs all 4 560 pairs of the yeast matrix)
    sel = np.arange(P_total) if sample_pairs >= P_total else rng.choice(P_total, size=sample_pairs, replace=False)
    pi, pj = iu[sel].astype(np.int32), ju[sel].astype(np.int32)
    cores = host_cores()
    chunks = np.array_split(np.arange(len(sel)), cores * 4)

    def work(ix):
        return O.ici_pairs(X, pi[ix], pj[ix], perspective, want_counts=True)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    out = np.concatenate([r[0] for r in res])
    cnt = np.concatenate([r[1] for r in res])
    return {"value": len(sel) / dt, "unit": "column-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{'ALL ' if len(sel) == P_total else ''}{len(sel)} {'' if len(sel) == P_total else 'random '}pairs of the same {n}x{S} matrix, {dt:.2f} s wall on {cores} threads "
                      f"({dt * cores / len(sel) * 1e3:.2f} core-ms per pair)",
            "note": "CPU restatement in C (oracle/), NOT the reference's Rcpp path (R is absent from the image); it "
                    "copies less than Rcpp sugar does, so per pair it is about 3x faster than the reference's own "
                    "single-core README figures interpolated to this length (README.md:180-185: 352.7 us at n = 1 000, "
                    "18 326 us at n = 40 000)"}, sel, out, cnt


def cpu_single_core_us():
    """Single-core per-pair latency of the CPU restatement at the README's two lengths (README.md:166-185:
    microbenchmark of ici_kt(x, y, "global"), times = 5, x = rnorm(n), y = rnorm(n)); median of 5."""
    from oracle import oracle as O
    out = {}
    for n in (1000, 40000):
        rng = np.random.default_rng(n)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        O.ici_kt(x, y, "global")
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            O.ici_kt(x, y, "global")
            ts.append(time.perf_counter() - t0)
        out[f"n{n}"] = float(np.median(ts) * 1e6)
    out["reference_readme_us"] = {"n1000": 352.7, "n40000": 18326.0}
    return out


def check_against_oracle(got_out, got_cnt, ref_out, ref_cnt):
    """Counts bit-exact, doubles within 1e-10 (BASELINE.json north_star)."""
    chk = {"pairs_checked_against_oracle": int(ref_out.shape[0]),
           "max_abs_diff": float(np.nanmax(np.abs(got_out - ref_out))),
           "nan_rows": int(np.isnan(got_out[:, 0]).sum())}
    if got_cnt is not None:
        chk["counts_bit_exact"] = bool(np.array_equal(got_cnt, ref_cnt[:, :got_cnt.shape[1]]))
    chk["ok"] = bool(chk["max_abs_diff"] <= ATOL and chk["nan_rows"] == 0 and chk.get("counts_bit_exact", True))
    return chk


PMC_RECORDS = os.path.join(ROOT, "profiles", "k1_pmc_records.json")
PMC_PASSES = (  # one rocprofv3 --pmc run each: 8 SQ slots (+ GRBM, its own block); FETCH_SIZE and WRITE_SIZE cannot share a pass
    ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE",
            "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("l2", ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"]),
)


def kernel_source_hash() -> str:
    """What a PMC record belongs to: the kernels, the structures they share with the host and the launch plan."""
    import hashlib
    h = hashlib.sha256()
    for f in ("icikt_kernels.hip", "icikt_device.h", "icikt_capi.cpp"):
        with open(os.path.join(ROOT, "icikendalltau_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_pass(config: str, counters, outdir: str, timeout_s: int):
    """One `rocprofv3 --pmc <counters> -- python3 tools/run_k1_once.py` child (the profiler starts the program itself:
    no shell, no env wrapper); returns ({counter: mean per K1 dispatch}, K1 ms per launch in that process)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    shutil.rmtree(outdir, ignore_errors=True)
    env = dict(os.environ, CONFIG=config, REPS="2", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "N_FEAT", "N_SAMP", "N_NA", "SEED", "PLAN", "MAX_PAIRS"):
        env.pop(k, None)
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", outdir, "--",
           sys.executable, os.path.join(ROOT, "tools", "run_k1_once.py")]
    res = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout_s)
    if res.returncode != 0:
        raise RuntimeError(f"rocprofv3 pass failed (rc {res.returncode}): {res.stderr[-300:]}")
    info = {}
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and "k1_ms_per_launch" in ln:
            info = json.loads(ln)
    agg = collections.defaultdict(list)
    names = collections.Counter()
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k1_pairs" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    names[r["Kernel_Name"]] += 1
    if not agg:
        raise RuntimeError("no k1_pairs dispatch in the counter output")
    info["kernel_names"] = dict(names)
    return {c: sum(v) / len(v) for c, v in agg.items()}, info   # one row per dispatch and counter: the mean per dispatch


def k1_instance_of(kernel_name: str):
    """'k1_pairsILi<NP>ELi<HI>E' (the mangled substring tools/valu_mix.py looks for) of a dispatch row's kernel name,
    demangled ('k1_pairs<2, 5>') or not; None if it is not one."""
    import re
    m = re.search(r"k1_pairs<\s*(\d+)\s*,\s*(\d+)\s*>", kernel_name) or re.search(r"k1_pairsILi(\d+)ELi(\d+)E", kernel_name)
    return f"k1_pairsILi{m.group(1)}ELi{m.group(2)}E" if m else None


def static_valu_mix(kernel: str):
    """Full-rate / half-rate vector instructions in the hot loop of pair kernel instance `kernel` (as the DISPATCH ROWS
    of the counter pass name it: no table of "the plan's choice" to go stale), from the compiler's assembly of THIS
    source (tools/valu_mix.py; ~7 s of hipcc)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    return valu_mix.hot_loop_mix(kernel)


def collect_pmc(config: str, timeout_s: int = 240):
    """The counters of the dominant kernel for this workload, taken NOW, on this box, with this build: three child
    runs under rocprofv3 (never combined with a trace).  Returns a record or raises."""
    import tempfile
    rec = {"config": config, "src_hash": kernel_source_hash(), "taken": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    base = tempfile.mkdtemp(prefix="icikt_pmc_")
    for name, counters in PMC_PASSES:
        means, info = _pmc_pass(config, counters, os.path.join(base, name), timeout_s)
        rec.update(means)
        if name == "sq":
            rec["k1_ms_in_pmc_pass"] = info.get("k1_ms_per_launch")
            rec["pairs_per_launch"] = info.get("pairs_per_launch")
            # the pair-kernel instance that RAN, from the dispatch rows of the counter output
            inst = {k1_instance_of(k): v for k, v in (info.get("kernel_names") or {}).items()}
            inst.pop(None, None)
            if inst:
                rec["k1_kernel"] = max(inst, key=inst.get)
                if len(inst) > 1:
                    rec["k1_kernel_note"] = f"several instances were dispatched: {inst}"
    try:
        if not rec.get("k1_kernel"):
            raise RuntimeError("the counter rows name no k1_pairs instance")
        rec["valu_mix_hot_loop"] = static_valu_mix(rec["k1_kernel"])
        assert rec["valu_mix_hot_loop"]["kernel"] == rec["k1_kernel"]
    except Exception as e:  # noqa: BLE001
        rec["valu_mix_note"] = f"static mix unavailable ({e!r})"[:200]
    return rec


def load_pmc_record(config: str, pairs_per_launch: int):
    """The committed record of this workload -- only if it was taken on THIS source (kernels + plan)."""
    try:
        with open(PMC_RECORDS) as f:
            recs = json.load(f)
    except Exception:
        return None, "no profiles/k1_pmc_records.json"
    h = kernel_source_hash()
    for r in recs:
        if r.get("config") == config and r.get("pairs_per_launch") == pairs_per_launch:
            if r.get("src_hash") == h:
                return r, None
            return None, f"the committed record of {config} was taken on source {r.get('src_hash')}, this build is {h}: refused"
    return None, f"no committed record for {config}"


def save_pmc_record(rec):
    try:
        with open(PMC_RECORDS) as f:
            recs = json.load(f)
    except Exception:
        recs = []
    recs = [r for r in recs if not (r.get("config") == rec["config"] and r.get("pairs_per_launch") == rec.get("pairs_per_launch"))]
    recs.append(rec)
    os.makedirs(os.path.dirname(PMC_RECORDS), exist_ok=True)
    with open(PMC_RECORDS, "w") as f:
        json.dump(sorted(recs, key=lambda r: r["config"]), f, indent=1)


N_SIMD = 1024           # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)


def build_roofline(pmc, pmc_source, pmc_note, k1_avg_s, P_local, n):
    """What bounds the pair kernel is vector-instruction issue: achieved = VALU wave-instructions
    per second in the timed run, peak = SIMDs x clock / issue cycles.  The HBM figures are kept beside it: the notional
    one (the reference's data flow priced against HBM: bytes that are never moved, can exceed 1) and the measured one."""
    alg_bytes = P_local * (16 * n + 32)
    roof = {"bound": "valu_issue", "kernel": "k1_pairs", "avg_launch_ms": k1_avg_s * 1e3, "unit": "G wave-instr/s",
            "achieved": None, "peak": None, "frac": None, "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "hbm_notional_GBs": alg_bytes / k1_avg_s / 1e9 if k1_avg_s > 0 else None,
            "hbm_notional_frac": alg_bytes / k1_avg_s / 1e9 / HBM_PEAK_GBS if k1_avg_s > 0 else None,
            "hbm_notional_note": "pairs x (16 n + 32) B (SURVEY.md 8(d): two f64 columns in, four f64 out per pair, as "
                                 "ici_split hands them to ici_kt) / launch time / 8 TB/s. Above 1 = bytes that are never "
                                 "moved: columns are sorted once and reused S-1 times. Not a utilisation.",
            "pmc_source": pmc_source}
    if pmc_note:
        roof["pmc_note"] = pmc_note
    if not pmc or not pmc.get("SQ_INSTS_VALU") or k1_avg_s <= 0:
        return roof
    insts = pmc["SQ_INSTS_VALU"]
    roof["valu_insts_per_launch"] = insts
    roof["achieved"] = insts / k1_avg_s / 1e9
    # Issue cost of a vector wave-instruction on a SIMD (tools/ubench/valu_rate.hip, valu_mix.hip; profiles/r03_valu_mix.log):
    # ONE wave issues at most one every 4 cycles; a SIMD retires a HALF-RATE form (DPP, VOPC / carry, v_bcnt, v_perm, every
    # three-operand integer form) every 4 cycles whatever the waves, and FULL-RATE forms (add, shift, logic, mov) of two
    # DIFFERENT waves every 2.  The floor of an instruction mix with a share h of half-rate forms is therefore
    # 4 h + 2 (1 - h) cycles per instruction; h comes from the compiler's assembly of the hot loop.
    mix = pmc.get("valu_mix_hot_loop") or {}
    nF, nH = mix.get("full", 0), mix.get("half", 0) + 2 * mix.get("permlane", 0)
    h = nH / (nF + nH) if nF + nH else 1.0
    cyc = 4.0 * h + 2.0 * (1.0 - h)
    gui = pmc.get("GRBM_GUI_ACTIVE")
    if gui:
        cycles = gui / 8.0                   # rocprofv3 sums the counter over the 8 XCDs: shader cycles of one K1 launch
        # The cycle count of a launch is a property of the kernel and its input; the TIME it takes depends on the clock the
        # chip holds, and that is lower under the profiler (MI355X_MICROARCH.md, DVFS (2)).  The timed run's clock is
        # therefore cycles / its launch time; peak and achieved are both at that clock, and frac is a ratio of counters
        # (and the static mix) alone: no timer enters it.
        clock = cycles / k1_avg_s
        roof["clock_GHz"] = clock / 1e9
        roof["clock_source"] = ("GRBM_GUI_ACTIVE / 8 (shader cycles of one K1 launch, SQ counter pass) / the un-profiled launch time "
                                "of the timed region; in the profiled pass itself: %.3f GHz over %.3f ms"
                                % (cycles / ((pmc.get("k1_ms_in_pmc_pass") or float("nan")) * 1e-3) / 1e9, pmc.get("k1_ms_in_pmc_pass") or float("nan")))
        roof["half_rate_share_hot_loop"] = h
        roof["issue_cycles_per_instruction_floor"] = cyc
        roof["peak"] = N_SIMD * clock / cyc / 1e9
        roof["frac"] = roof["achieved"] / roof["peak"]                      # == insts x floor cycles / (SIMDs x launch cycles)
        roof["frac_kind"] = ("MIX-RELATIVE: against the issue floor of THIS kernel's hot-loop instruction mix (h from the "
                             "compiler's assembly of the instance the dispatch rows name, %s); the two fixed-denominator "
                             "fractions beside it need no model: frac_of_2_cycle_slots (the guide's full-rate issue, a true "
                             "upper bound) and frac_of_4_cycle_slots (every instruction in a DPP-stream slot; can exceed 1)"
                             % (pmc.get("k1_kernel") or "?"))
        roof["frac_of_4_cycle_slots"] = insts * 4.0 / (N_SIMD * cycles)     # every instruction in a 4-cycle slot: > 1 = some full-rate forms did overlap
        roof["frac_of_2_cycle_slots"] = insts * 2.0 / (N_SIMD * cycles)     # against the SIMD's best case (full-rate forms only)
        roof["issue_note"] = ("peak = SIMDs x clock / (4 h + 2 (1 - h)) with h the share of half-rate forms in the hot loop "
                              "(tools/valu_mix.py on this source); see DESIGN.md section 4 and profiles/r03_valu_mix.log")
    busy = pmc.get("SQ_BUSY_CU_CYCLES")
    if busy and pmc.get("SQ_LDS_IDX_ACTIVE"):
        roof["lds_active_frac"] = pmc["SQ_LDS_IDX_ACTIVE"] / busy
        if pmc.get("SQ_LDS_BANK_CONFLICT") is not None:
            roof["lds_bank_conflict_frac"] = pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]
    if pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
        # MI355X_MICROARCH.md, HBM: both in KiB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        roof["traffic"] = traffic
        roof["traffic_uncorrected"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        roof["hbm_measured_GBs"] = traffic / k1_avg_s / 1e9
        roof["hbm_measured_frac"] = roof["hbm_measured_GBs"] / HBM_PEAK_GBS
    if pmc.get("TCC_REQ_sum"):
        # The scattered `rec` gathers of the pair kernel: one L2 request per lane and gather when the gathered block does not
        # sit in the CU's L1 (long columns).  MI355X_MICROARCH.md, "Indexed rows": rows served from an XCD's L2 arrive at
        # 16.8-18.8 TB/s chip-wide; at 64 B per request that is ~2.8e11 requests/s.  Long columns ran AT that rate with one
        # pair per gather, which is why they now run two pairs per gather.
        req = pmc["TCC_REQ_sum"]
        roof["l2_requests_per_launch"] = req
        roof["l2_requests_per_s"] = req / k1_avg_s
        roof["l2_GBs_at_64B_per_request"] = req * 64.0 / k1_avg_s / 1e9
        roof["l2_frac_of_guide_gather_rate"] = req * 64.0 / k1_avg_s / 1e9 / 17800.0
        if pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None and (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]) > 0:
            roof["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
        if pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            roof["l1_hit_rate"] = 1.0 - req / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] if pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] > req else 0.0
    roof["pmc"] = {k: pmc[k] for k in sorted(pmc) if k[:3] in ("SQ_", "GRB", "FET", "WRI", "TCC", "TCP") or
                   k in ("src_hash", "taken", "k1_ms_in_pmc_pass", "valu_mix_hot_loop", "valu_mix_note", "k1_kernel", "k1_kernel_note")}
    if pmc.get("k1_kernel"):
        roof["kernel"] = pmc["k1_kernel"]
    return roof


def workload_text(config, cfg, P_total, both):
    n, S = cfg["n_feat"], cfg["n_samp"]
    if CONFIGS[config].get("fixture"):
        return (f"{config}: yeast_missing (the reference's data/yeast_missing.rda), {n} features x {S} samples, zeros -> missing, "
                f"perspective=global, {P_total} column pairs")
    tied = f"values rounded to ~{cfg['levels']} distinct levels per column (tied data), " if cfg.get("levels") else ""
    return (f"{config}: {n} features x {S} samples, {tied}{cfg['n_na']} smallest per column missing, "
            f"perspective={'global + local' if both else 'global'}, {P_total} column pairs")


def spread(rows, keys):
    """{key: {"max": .., "min": ..}} over the ranks' rows (a list of dicts)."""
    return {k: {"max": max(r[k] for r in rows), "min": min(r[k] for r in rows)} for k in keys}


def run_inlib(args, cfg):
    """N GPUs behind one C call from one process: what an R caller's n_gpu reaches (icikt_pairs_multi_f64).  The
    boundary hands over HOST buffers, so this rate is PCIe-inclusive by construction."""
    from icikendalltau_amd import _lib
    n, S = cfg["n_feat"], cfg["n_samp"]
    X = workload_matrix(args.config, cfg)
    P_total = S * (S - 1) // 2
    # (RCCL prints a version banner on stdout when its first communicator is made: keep stdout to the one JSON line)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        m = _lib.MultiContext(n_gpu=args.gpus, exchange=os.environ.get("ICIKT_BENCH_EXCHANGE", "auto"))
        m.pairs(X, perspective="global", want_counts=False)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    out = None
    for _ in range(args.warmup):
        out, cnt, _r = m.pairs(X, perspective="global", want_counts=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, _c, _r = m.pairs(X, perspective="global", want_counts=False)
    elapsed = time.perf_counter() - t0
    m.pairs(X, perspective="global", want_counts=False, flags=_lib.FLAG_TIMING)
    phases = m.phase_ms()
    ranks = m.rank_phase_ms()[:max(1, m.ranks_used)]
    line = {
        "metric": "column-pairs/s (+ full-matrix wall time) at 10k feat x 1k samp",
        "value": P_total * args.steps / elapsed, "unit": "column-pairs/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int32 rank/popcount counting + f64 epilogue",
        "data": "synthetic" if not CONFIGS[args.config].get("fixture") else "reference fixture (yeast_missing)", "launcher": "inlib",
        "config": {"workload": workload_text(args.config, cfg, P_total, False) + ", HOST buffers in and out "
                               "(icikt_pairs_multi_f64: PCIe-inclusive)",
                   "pairs": P_total, "n_feat": n, "n_samp": S, "exchange": "rccl" if m.uses_rccl else "copy"},
        "rccl_ranks": m.comm_ranks,          # ncclCommCount of the handle's communicator (0: device copies, no communicator)
        "ranks_used": m.ranks_used,
        "phase_ms_synced": phases,
        "rank_phase_ms": spread(ranks, _lib.MULTI_PHASES + ("wait",)),
    }
    if args.cpu_sample > 0:
        _b, sel, ref_out, ref_cnt = cpu_baseline(X, P_total, min(args.cpu_sample, 2000))
        line["check"] = check_against_oracle(out[sel], cnt[sel] if cnt is not None else None, ref_out, ref_cnt)
    m.close()
    print(json.dumps(line), flush=True)
    return 0 if line.get("check", {}).get("ok", True) else 1


PHASES = ("k0", "exchange", "k1", "gather")   # per step and rank: pre-pass of the rank's columns | all-gather + rebuild | pair kernel + epilogue | result gather


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c4")
    ap.add_argument("--launcher", choices=("torchrun", "inlib"), default="torchrun",
                    help="torchrun: one process per GPU over torch.distributed (self-launched when WORLD_SIZE is unset); "
                         "inlib: one process, N GPUs behind one C call")
    ap.add_argument("--n-feat", type=int, default=None)
    ap.add_argument("--n-samp", type=int, default=None)
    ap.add_argument("--n-na", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cpu-sample", type=int, default=None, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive / end-to-end legs")
    ap.add_argument("--pmc", choices=("live", "record", "off"), default="live",
                    help="roofline counters of the pair kernel: taken now by child rocprofv3 --pmc runs (N = 1; falls back "
                         "to the committed record of the same source), from the committed record only, or not at all")
    ap.add_argument("--save-pmc", action="store_true", help="write the live counters to profiles/k1_pmc_records.json")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # `python bench.py --gpus N` as typed: this process is the parent of the ranks (no torch, no GPU in it)
    if should_spawn(args.gpus, args.launcher, os.environ):
        return spawn_ranks(argv, args.gpus)

    import torch
    cfg = dict(CONFIGS[args.config])
    for k in ("n_feat", "n_samp", "n_na", "seed", "steps", "warmup", "cpu_sample"):
        v = getattr(args, k)
        if v is not None:
            cfg[k] = v
    args.steps, args.warmup, args.cpu_sample = cfg["steps"], cfg["warmup"], cfg["cpu_sample"]
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    if args.launcher == "inlib":
        return run_inlib(args, cfg)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world   # started under torch.distributed.run with another rank count: the launcher's count holds
    # ICIKT_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: every rank uses the visible device
    # and the collectives go through host memory.  The driver's N > 1 runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("ICIKT_BENCH_BACKEND", "nccl")
    via_host = backend != "nccl"
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # ICIKT_BENCH_FORCE_DIST=1: a process group of ONE rank runs the whole N > 1 sequence (sharded pre-pass, in-place
    # all-gather of the library's device arrays, gather) over the real nccl = RCCL backend on a one-GPU box
    distributed = world > 1 or os.environ.get("ICIKT_BENCH_FORCE_DIST") == "1"
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from icikendalltau_amd import _lib, sharding
    ctx = _lib.Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    n, S = cfg["n_feat"], cfg["n_samp"]
    X = workload_matrix(args.config, cfg)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)  # (S, n) row-major == n x S column-major
    P_total = S * (S - 1) // 2
    begin, end, n_each = sharding.pair_block(P_total, rank, world)  # ceiling(n_todo / ncore), R/kendalltau.R:250
    P_local = end - begin
    out_local = torch.full((n_each, 4), float("nan"), dtype=torch.float64, device=dev)
    out_second = torch.empty((n_each, 4), dtype=torch.float64, device=dev) if args.config == "c5" else None
    flags = _lib.FLAG_TIMING
    gathered = None
    both = args.config == "c5"   # BASELINE config 5: perspective = "local" vs "global"

    # N > 1: each rank sorts only its S/N columns (K0); order, the bitsets and stats are all-gathered (RCCL over
    # xGMI) and the rest of the prepared state is rebuilt locally, instead of every rank repeating the whole
    # pre-pass; all ranks together fall back to the replicated pre-pass if that cannot be set up.
    sp = None
    prep_mode = "single"
    if distributed:
        prep_mode = "replicated"
        if os.environ.get("ICIKT_BENCH_PREP", "sharded") == "sharded":
            sp = sharding.ShardedPrepass(ctx, dist, dev, via_host)
            if not sp.setup(S, lambda c0, c1, alloc, fl: ctx.prepare_cols_dev(dX.data_ptr(), n, S, n, c0, c1, alloc, fl),
                            sync=torch.cuda.synchronize):
                sp = None
            else:
                prep_mode = sp.mode
    ctx.set_pairs_combn(S, begin, end)

    # per-rank phase times: events on the stream everything of a step is launched on (torch's current stream: the
    # library runs on it -- set_stream above -- and the collectives synchronise with it), five per step
    ev_pool = []

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def step(timed=False):
        nonlocal gathered
        evs = [mark()] if timed else None
        if sp is None:
            ctx.prepare_dev(dX.data_ptr(), n, S, n, flags)
            if timed:
                evs += [mark(), mark()]          # k0 | (no exchange)
        else:
            sp.run(flags, on_phase=(lambda _name: evs.append(mark())) if timed else None)   # "k0", "exchange"
        if P_local > 0:
            ctx.run_dev(_lib.PERSPECTIVE["global"], _lib.ALTERNATIVE["two.sided"], False, flags, out_local.data_ptr())
            if both:  # the second perspective is the epilogue over the same pair counts
                ctx.run_dev(_lib.PERSPECTIVE["local"], _lib.ALTERNATIVE["two.sided"], False,
                            flags | _lib.FLAG_REUSE_COUNTS, out_second.data_ptr())
        if timed:
            evs.append(mark())                   # k1 (+ epilogue)
        if distributed:
            # every rank's P/N x 4 results to rank 0 (RCCL over xGMI)
            gathered = sharding.gather_blocks(dist, out_local, n_each, dev, via_host, to_all=False)
            if both:
                sharding.gather_blocks(dist, out_second, n_each, dev, via_host, to_all=False)
        if timed:
            evs.append(mark())                   # gather
            ev_pool.append(evs)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: the first launches after start-up run at lower clocks (c4: 11.0, 9.9, 9.4, 9.2, 9.1 ... ms); a few
    # untimed passes before the caller's own warm-up steps keep that out of short timed runs
    for _ in range(cfg.get("pre_warm", 0)):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if via_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # this rank's phases, ms per step (the events are complete: fence() synchronised)
    my_phase = [sum(evs[i].elapsed_time(evs[i + 1]) for evs in ev_pool) / max(1, len(ev_pool)) for i in range(4)]
    rank_rows = [dict(zip(PHASES, my_phase))]
    if distributed:
        pt = torch.tensor(my_phase, dtype=torch.float64, device="cpu" if via_host else dev)
        parts = [torch.empty_like(pt) for _ in range(world)]
        dist.all_gather(parts, pt)
        rank_rows = [dict(zip(PHASES, [float(v) for v in q.cpu()])) for q in parts]

    k_ms = {name: ctx.kernel_ms(k) for name, k in (("prepare", _lib.K_PREPARE), ("pairs", _lib.K_PAIRS),
                                                    ("epilogue", _lib.K_EPILOGUE))}
    status = 0
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P_total * args.steps / elapsed
        # roofline of the dominant kernel (K1): algorithmic bytes per launch = pairs in the launch x (16 n + 32)
        # (two float64 columns in, four float64 out per pair: SURVEY.md section 8(d)) / avg launch duration
        # (the pair kernel runs as one launch per pre-pass group when it overlaps the pre-pass: the launches of a
        #  step together process P_local pairs, so the step's K1 time is what the bytes are divided by)
        k1_ms, k1_n = k_ms["pairs"]
        k1_avg_s = (k1_ms / args.steps) / 1e3
        # counters of the dominant kernel: measured with the run (child rocprofv3 passes, after the timed region), or the
        # committed record IF it was taken on this source; otherwise null + a note, never a stale constant
        pmc, pmc_source, pmc_note = None, "none", None
        std_workload = all(cfg[k] == CONFIGS[args.config][k] for k in ("n_feat", "n_samp", "n_na", "seed")) and world == 1
        if args.pmc != "off" and std_workload:
            if args.pmc == "live":
                try:
                    pmc = collect_pmc(args.config)
                    pmc_source = "live: rocprofv3 --pmc child runs of tools/run_k1_once.py on this box, this build"
                    if args.save_pmc:
                        save_pmc_record(pmc)
                except Exception as e:  # noqa: BLE001
                    pmc_note = f"live collection failed ({e!r})"[:400]
            if pmc is None:
                pmc, why = load_pmc_record(args.config, P_local)
                if pmc is not None:
                    pmc_source = f"record: profiles/k1_pmc_records.json taken {pmc.get('taken')} on source {pmc.get('src_hash')}"
                else:
                    pmc_note = "; ".join(x for x in (pmc_note, why) if x)
        elif args.pmc != "off":
            pmc_note = "counters are kept for the standard single-GPU workloads only"
        roof = build_roofline(pmc, pmc_source, pmc_note, k1_avg_s, P_local, n)
        roof["kernel_launches_per_step"] = k1_n / args.steps
        if cfg.get("levels"):
            roof["tied_data_note"] = ("tied data: every step of the pair kernel is a tie step (MIXED / SOLO / GROUP, DESIGN.md section 2), "
                                      "not the hot loop whose instruction mix `frac` is priced against; frac_of_4_cycle_slots and "
                                      "frac_of_2_cycle_slots need no mix")
        line = {
            "metric": "column-pairs/s (+ full-matrix wall time) at 10k feat x 1k samp",
            "value": value, "unit": "column-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 rank/popcount counting + f64 epilogue",
            "data": "synthetic" if not CONFIGS[args.config].get("fixture") else "reference fixture (yeast_missing)",
            "config": {"workload": workload_text(args.config, cfg, P_total, both), "pairs": P_total, "n_feat": n, "n_samp": S,
                       "sharding": f"combn-order blocks over {world} rank(s)", "pre_pass": prep_mode},
            "value_definition": "matrix resident in HBM when the timed region starts; K0 + (N > 1: all-gather + rebuild) + K1 + K2 "
                                "+ (N > 1: gather to rank 0); barrier + synchronize on both sides, max over ranks",
            # the communicator as the backend reports it (nccl = RCCL); null when the run had no process group
            "rccl_ranks": (dist.get_world_size() if (distributed and backend == "nccl") else None),
            "backend": (backend if distributed else None),
            # per rank and step, from events on the launch stream: max and min over the ranks
            "rank_phase_ms": spread(rank_rows, PHASES),
            "roofline": roof,
            # accumulated event time of each kernel id / steps (the pre-pass id covers K0 and, at N > 1, the two
            # rebuild launches of a step, not the all-gather between them)
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in k_ms.items()},
        }
        if world == 1 and args.cpu_sample > 0:
            base, sel, ref_out, ref_cnt = cpu_baseline(X, P_total, args.cpu_sample)
            line["cpu_baseline"] = base
            # the sampled pairs once more through the counts-carrying entry: integers bit-exact, doubles <= 1e-10;
            # and the timed run's own output on the same pairs
            iu, ju = np.triu_indices(S, k=1)
            cnt_t = torch.zeros((len(sel), len(_lib.CNT_FIELDS)), dtype=torch.int64, device=dev)
            out_t = torch.empty((len(sel), 4), dtype=torch.float64, device=dev)
            ctx.set_pairs(iu[sel].astype(np.int32), ju[sel].astype(np.int32))
            ctx.run_dev(_lib.PERSPECTIVE["global"], 0, False, 0, out_t.data_ptr(), cnt_t.data_ptr())
            torch.cuda.synchronize()
            line["check"] = check_against_oracle(out_local[:P_local].cpu().numpy()[sel], cnt_t.cpu().numpy(), ref_out, ref_cnt)
            line["check"]["sampled_rerun_equal"] = bool(np.array_equal(out_t.cpu().numpy(), out_local[:P_local].cpu().numpy()[sel]))
            line["check"]["ok"] = line["check"]["ok"] and line["check"]["sampled_rerun_equal"]
            if both:
                from oracle import oracle as O
                k = min(200, len(sel))
                ref_l, _c, _r = O.ici_pairs(X, iu[sel[:k]], ju[sel[:k]], "local", want_counts=False)
                d = float(np.nanmax(np.abs(out_second[:P_local].cpu().numpy()[sel[:k]] - ref_l)))
                line["check"]["local_max_abs_diff"] = d
                line["check"]["ok"] = line["check"]["ok"] and d <= ATOL
            ctx.set_pairs_combn(S, begin, end)
        if distributed:
            # the assembled result: rank blocks concatenated in combn order (no NaN may be left in real pairs)
            full = sharding.assemble(gathered, P_total, n_each).cpu().numpy()
            chk = {"assembled_pairs": int(full.shape[0]), "nan_rows": int(np.isnan(full[:, 0]).sum())}
            chk["ok"] = full.shape[0] == P_total and chk["nan_rows"] == 0
            if args.cpu_sample > 0:
                from oracle import oracle as O
                rng = np.random.default_rng(1)
                sel = rng.choice(P_total, size=min(2000, P_total), replace=False)
                iu, ju = np.triu_indices(S, k=1)
                ref, _c, _r = O.ici_pairs(X, iu[sel], ju[sel], "global", want_counts=False)
                chk.update(pairs_checked_against_oracle=int(len(sel)), max_abs_diff=float(np.nanmax(np.abs(full[sel] - ref))))
                chk["ok"] = chk["ok"] and chk["max_abs_diff"] <= ATOL
            if "check" in line:   # (one rank over a process group: both checks hold)
                chk["ok"] = chk["ok"] and line["check"]["ok"]
                line["check"].update(chk)
            else:
                line["check"] = chk
        if not distributed and not args.no_extras:
            extras(line, args, cfg, X, ctx, dev)
        status = 0 if line.get("check", {}).get("ok", True) else 1
        print(json.dumps(line), flush=True)
    if distributed:
        st = torch.tensor([status], dtype=torch.int32, device="cpu" if via_host else dev)
        dist.broadcast(st, src=0)
        status = int(st.item())
        dist.barrier()
        dist.destroy_process_group()
    return status


def extras(line, args, cfg, X, ctx, dev):
    """N = 1 only, after the resident timed region: the figures SURVEY.md section 8(d) asks for beside the metric, each
    over K steps (mean, not a best-of)."""
    import torch
    from icikendalltau_amd import _lib, api
    n, S = X.shape
    P_total = S * (S - 1) // 2
    ctx.use_own_stream()
    L = _lib.lib()
    # (1) PCIe-inclusive -- the metric as section 8(d) words it: the host-buffer entry the R glue binds (pageable host
    #     matrix in: H2D in column chunks overlapped with K0 and the pair kernel; D2H of the results into pageable host
    #     arrays), `steps` calls after `warmup`, wall clock around the loop, mean
    out4 = np.empty((P_total, 4))
    rsn = np.zeros(P_total, np.int32)

    def host_call():
        rc = L.icikt_pairs_f64(ctx._h, X.ctypes.data, n, S, n, None, None, 0, 1, 0, 0, 0, out4.ctypes.data, None, rsn.ctypes.data)
        assert rc == 0, L.icikt_last_error(ctx._h)

    for _ in range(max(1, args.warmup)):
        host_call()
    ts = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        host_call()          # (synchronous: the results are in the caller's arrays when it returns)
        ts.append(time.perf_counter() - t0)
    mean = sum(ts) / len(ts)
    line["pcie_inclusive"] = {"value": P_total / mean, "unit": "column-pairs/s", "ms_per_step": mean * 1e3, "steps": len(ts),
                              "warmup": max(1, args.warmup), "ms_min": min(ts) * 1e3, "ms_max": max(ts) * 1e3,
                              "entry": "icikt_pairs_f64 (pageable host matrix in, host results out; one perspective)",
                              "h2d_bytes": int(X.nbytes), "d2h_bytes": int(out4.nbytes + rsn.nbytes)}
    # (2) the five S x S matrices of ici_kendalltau() behind ONE library call (icikt_matrix_f64: exclusion rule in the
    #     pre-pass, pair kernels, scale_and_reshape on the device, one D2H of 5 S^2 doubles + keep): mean over the steps
    reps = min(args.steps, 5)
    ctx.matrix(X, (float("nan"),))
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.matrix(X, (float("nan"),))
    line["full_matrix_wall_ms"] = (time.perf_counter() - t0) / reps * 1e3
    line["full_matrix_note"] = f"icikt_matrix_f64, host buffers in and out, mean of {reps} calls; d2h {5 * S * S * 8 + n * S} bytes"
    # (3) the whole ici_kendalltau() equivalent of the Python front-end: argument checks, the call above, data frames
    if args.config != "c5":
        names = [f"s{i}" for i in range(S)]
        eng = api.HipEngine(device=dev.index)
        ts, rt = [], []
        for _ in range(3):   # the first call also creates the engine's context and workspaces
            t0 = time.perf_counter()
            res = api.ici_kendalltau(X, global_na=(float("nan"),), perspective="global", colnames=names, engine=eng)
            ts.append(time.perf_counter() - t0)
            rt.append(res["run_time"])
        line["e2e_ici_kendalltau_ms"] = min(ts) * 1e3
        line["e2e_first_call_ms"] = ts[0] * 1e3
        line["e2e_run_time_field_ms"] = min(rt) * 1e3          # the run_time field: the library call alone
        line["e2e_vs_pcie_inclusive"] = line["e2e_ici_kendalltau_ms"] / line["pcie_inclusive"]["ms_per_step"]
    else:
        # (4) BASELINE config 5's subset legs: include_only = the first 64 names (pairs with s1 OR s2 among them) in
        #     both perspectives, and pairwise_completeness (self pairs included) on the same subset
        names = [f"s{i}" for i in range(S)]
        pi, pj, _core = api.setup_comparisons(names, include_only=names[:64], diag_good=True, ncore=1)
        dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
        sub = torch.empty((len(pi), 4), dtype=torch.float64, device=dev)
        ctx.prepare_dev(dX.data_ptr(), n, S, n, 0)
        ctx.set_pairs(pi, pj)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.run_dev(_lib.PERSPECTIVE["global"], 0, False, 0, sub.data_ptr())
        ctx.run_dev(_lib.PERSPECTIVE["local"], 0, False, _lib.FLAG_REUSE_COUNTS, sub.data_ptr())
        ctx.sync()
        t_sub = time.perf_counter() - t0
        del dX
        pc_i, pc_j, _c = api.setup_comparisons(names, include_only=names[:64], diag_good=False, ncore=1)
        ctx.missingness(X, pc_i, pc_j)
        t0 = time.perf_counter()
        miss = ctx.missingness(X, pc_i, pc_j)
        t_pc = time.perf_counter() - t0
        line["include_only_subset"] = {"pairs": int(len(pi)), "both_perspectives_ms": t_sub * 1e3,
                                       "pairs_per_s": len(pi) / t_sub,
                                       "pairwise_completeness_pairs": int(len(pc_i)), "pairwise_completeness_ms": t_pc * 1e3,
                                       "pairwise_completeness_note": "host-buffer entry icikt_missingness_f64: H2D of the matrix "
                                                                     "in chunks, a mask-only pre-pass per chunk (no sort), popcounts",
                                       "missingness_checksum": int(miss.sum())}
    # (5) single-core latency of the CPU restatement beside the reference's README table
    line["cpu_single_core_us"] = cpu_single_core_us()


if __name__ == "__main__":
    sys.exit(main())

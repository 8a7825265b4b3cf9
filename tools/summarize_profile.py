#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (written by tools/profile_k1.sh on the GPU box) into profiles/.

    python tools/summarize_profile.py r01 [--n-feat 10000 --n-samp 1024]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc.md (per-kernel counter means) and profiles/k1_hbm_traffic.json (read by bench.py).
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request, so
read bytes = 2 x FETCH_SIZE x 1024 (uncalibrated for this kernel's 2-byte-per-lane loads; the
uncorrected figure is kept beside it).
"""
import argparse, collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--n-feat", type=int, default=10000)
ap.add_argument("--n-samp", type=int, default=1024)
ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out"))
a = ap.parse_args()
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

stats = sorted(glob.glob(os.path.join(a.src, "prof_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:  # gpurun merges every call's files into the same directories: the newest one is this run's
    shutil.copy(stats[-1], os.path.join(out, f"{a.tag}_kernel_stats.csv"))

means = collections.defaultdict(dict)
for d in ("prof_fetch", "prof_write", "prof_sq", "prof_sq2", "prof_ta", "prof_tcp", "prof_tcc"):
    for f in sorted(glob.glob(os.path.join(a.src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
        for (k, c), v in agg.items():
            means[k][c] = sum(v) / len(v)
            means[k]["_meta"] = meta[k]

with open(os.path.join(out, f"{a.tag}_pmc.md"), "w") as f:
    f.write(f"# rocprofv3 --pmc means per dispatch ({a.tag}; bench.py c4 workload, {a.n_feat} x {a.n_samp})\n\n")
    for k, cs in sorted(means.items()):
        if "icikt" not in k:
            continue
        g = cs.get("_meta")
        f.write(f"## {k}\n\ngrid={g[0]} wg={g[1]} lds={g[2]} vgpr={g[3]} sgpr={g[4]}\n\n| counter | mean per dispatch |\n|---|---|\n")
        for c, v in sorted(cs.items()):
            if c != "_meta":
                f.write(f"| {c} | {v:.6g} |\n")
        f.write("\n")

k1 = next((v for k, v in means.items() if "k1_pairs" in k), None)
if k1 and "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
    P = a.n_samp * (a.n_samp - 1) // 2
    js = {"tag": a.tag, "n_feat": a.n_feat, "n_samp": a.n_samp, "pairs_per_launch": P,
          "FETCH_SIZE_KiB": k1["FETCH_SIZE"], "WRITE_SIZE_KiB": k1["WRITE_SIZE"],
          "hbm_bytes_per_launch": (2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024,
          "hbm_bytes_per_launch_uncorrected": (k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024,
          "source": f"profiles/{a.tag}_pmc.md (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes)",
          "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); separate --pmc passes"}
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE"):
        if c in k1:
            js[c] = k1[c]
    json.dump(js, open(os.path.join(out, "k1_hbm_traffic.json"), "w"), indent=1)
    print(js)
print("wrote", sorted(os.listdir(out)))

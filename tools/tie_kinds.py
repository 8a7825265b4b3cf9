"""Where the pair kernel's instructions go on tied data: per tie-sweep matrix (10 000 x 256,
columns rounded to ~L distinct values, 5 % missing) the steps a task takes by KIND and their rows (diagnostic build,
tools/step_stats.py) beside the vector / scalar / LDS instructions of a K1 launch (rocprofv3 --pmc, tools/pmc_tie.sh),
per task, per step and per row of the streamed column (a row of a task = one row of two pairs).

    python tools/tie_kinds.py gpurun_out/<tag>_step_stats.md gpurun_out/<tag>_pmc_tie.log
"""
import ast, re, sys
import numpy as np

md, pmc = sys.argv[1], sys.argv[2]
cases, cur = {}, None
for ln in open(md):
    m = re.match(r"## ~(\w+) distinct values: (\d+) x (\d+), (\d+) pairs", ln)
    if m:
        cur = 0 if m.group(1) == "continuous" else int(m.group(1))
        cases[cur] = {"pairs": int(m.group(4)), "kinds": {}}
        continue
    if ln.startswith("## "):
        cur = None
    m = re.match(r"\| (\w+) \| (\d+) \| ([\d.]+) \|", ln)
    if m and cur is not None:
        cases[cur]["kinds"][m.group(1)] = (int(m.group(2)), float(m.group(3)))
counters = {}
for ln in open(pmc):
    m = re.match(r"gpurun_out/tie(\d+)_a (\{.*\})", ln.strip())
    if m:
        counters[int(m.group(1))] = {k: float(v) for k, v in ast.literal_eval(m.group(2)).items()}
levels = [L for L in sorted(cases) if L in counters]
kinds = ["hot", "mixed", "group"]
def steps(L, kind):
    k = cases[L]["kinds"]
    if kind == "hot":
        return k.get("hot_loop", (0, 0))[0] + k.get("hot_in_main", (0, 0))[0]
    return k.get(kind, (0, 0))[0]
print("| distinct values | steps per task: hot / MIXED / GROUP | rows per MIXED / GROUP step | VALU per task | per step | per row | SALU per row | LDS per row | VALU per row vs continuous |")
print("|---|---|---|---|---|---|---|---|---|")
base = None
for L in levels:
    k = cases[L]["kinds"]
    tasks = k.get("setup", (2, 0))[0] / 2.0            # the set-up mark is stamped twice per task
    st = [steps(L, kd) / tasks for kd in kinds]
    rows = sum(k.get(kd, (0, 0))[0] * k.get(kd, (0, 0))[1] for kd in ("hot_loop", "hot_in_main", "mixed", "group", "tail")) / tasks
    v, sa, ld = (counters[L][c] / tasks for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
    if base is None:
        base = v / rows
    print(f"| {L or 'continuous'} | {st[0]:.0f} / {st[1]:.0f} / {st[2]:.0f} | {k.get('mixed',(0,0))[1]:.1f} / {k.get('group',(0,0))[1]:.1f} | "
          f"{v:.0f} | {v / max(sum(st), 1):.0f} | {v / rows:.2f} | {sa / rows:.2f} | {ld / rows:.2f} | {v / rows / base:.2f} |")

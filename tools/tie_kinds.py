"""Instructions per step KIND of the pair kernel on tied data: fits, over the tie-sweep matrices,
    SQ_INSTS_x (per K1 launch, rocprofv3 --pmc: tools/pmc_tie.sh)  =  sum over kinds of steps_kind x c_kind + tasks x c_task
by least squares, with the steps per kind taken from the diagnostic build (tools/step_stats.py).  Six matrices
(continuous, ~5000, 1000, 200, 50, 10 distinct values per column), four unknowns (hot, MIXED, GROUP step; per task).

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
A = np.array([[steps(L, k) for k in kinds] + [cases[L]["kinds"].get("setup", (0, 0))[0]] for L in levels], dtype=float)
print("| distinct values | steps per task: hot / MIXED / GROUP | rows per MIXED / GROUP step | VALU per launch | SALU per launch |")
print("|---|---|---|---|---|")
for i, L in enumerate(levels):
    t = A[i, 3] or 1
    k = cases[L]["kinds"]
    print(f"| {L or 'continuous'} | {A[i,0]/t:.1f} / {A[i,1]/t:.1f} / {A[i,2]/t:.1f} | {k.get('mixed',(0,0))[1]:.1f} / {k.get('group',(0,0))[1]:.1f} | "
          f"{counters[L]['SQ_INSTS_VALU']:.3e} | {counters[L]['SQ_INSTS_SALU']:.3e} |")
print()
print("| instructions per step (least squares over the matrices) | hot | MIXED | GROUP | per task (set-up, tail, reductions) |")
print("|---|---|---|---|---|")
for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    y = np.array([counters[L][name] for L in levels])
    c, *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"| {name} | {c[0]:.0f} | {c[1]:.0f} | {c[2]:.0f} | {c[3]:.0f} |")

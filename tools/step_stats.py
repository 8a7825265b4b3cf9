"""Where the pair kernel's time goes, by step kind: builds a diagnostic variant of the library
(-DICIKT_STEP_STATS: s_memtime stamps around every step; the product build has none), runs the tie-sweep matrices
(10 000 x 256, columns rounded to ~L distinct values, 5 % missing) and the yeast matrix on it and prints, per kind,
steps, rows per step, wave cycles per step and per row, and the share of all wave cycles.

    python tools/step_stats.py build          (here: compiles tools/libicikt_stats.so, which travels with gpurun)
    python tools/step_stats.py [plan]         (GPU box)
"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "tools", "libicikt_stats.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from icikendalltau_amd import _lib
    print(_lib.build(extra_flags=("-DICIKT_STEP_STATS",), out=VARIANT))
    sys.exit(0)
if os.environ.get("ICIKT_LIB") != VARIANT:
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, ICIKT_LIB=VARIANT)).returncode)
import numpy as np, torch
from icikendalltau_amd import _lib


def report(tag, ctx, X, plan):
    n, S = X.shape
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    P = S * (S - 1) // 2
    ctx.debug_set_plan(plan)
    ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    ctx.prepare_dev(dX.data_ptr(), n, S, n, 0)
    ctx.run_dev(1, 0, False, 0, out.data_ptr()); ctx.sync()
    ctx.step_stats(reset=True)
    ctx.reset_timers()
    ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
    ms = ctx.kernel_ms(_lib.K_PAIRS)[0]
    st = ctx.step_stats(reset=True)
    tot = sum(v[2] for v in st.values()) or 1
    print(f"## {tag}: {n} x {S}, {P} pairs, K1 {ms:.3f} ms (diagnostic build: stamps cost ~10 %)")
    print("| kind | steps | rows / step | wave cycles / step | / row | share of wave cycles |")
    print("|---|---|---|---|---|---|")
    for k, (s, r, c) in st.items():
        if s:
            print(f"| {k} | {s} | {r / s:.1f} | {c / s:.0f} | {(c / r) if r else float('nan'):.1f} | {100.0 * c / tot:.1f} % |")
    print(flush=True)


plan = sys.argv[1] if len(sys.argv) > 1 else ""
ctx = _lib.Context(0)
n, S = 10000, 256
rng = np.random.default_rng(3)
base = rng.standard_normal((n, S))
for levels in (0, 5000, 1000, 200, 50, 10):
    X = base.copy() if levels == 0 else np.round(base * (levels / 6.0))
    X[rng.random(X.shape) < 0.05] = np.nan
    report(f"~{levels or 'continuous'} distinct values", ctx, np.asfortranarray(X), plan)
z = np.load(os.path.join(ROOT, "tests", "golden", "yeast_missing.npz"))
Y = np.asfortranarray(z[z.files[0]].astype(np.float64))
Y[Y == 0] = np.nan
report("yeast (c2)", ctx, Y, plan)

"""Step kinds of the pair kernel for a data-set shape on the diagnostic build (development aid): argv n_feat n_samp levels."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "tools", "libicikt_stats.so")
if os.environ.get("ICIKT_LIB") != VARIANT:
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, ICIKT_LIB=VARIANT)).returncode)
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix
n, S, levels = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X = make_matrix(n, S, n // 10, 11, levels)
ctx = _lib.Context(0)
if len(sys.argv) > 4: ctx.debug_set_plan(sys.argv[4])
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0)
ctx.step_stats(reset=True)
ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
print(f"{n} x {S} levels={levels}: K1 {ctx.kernel_ms(_lib.K_PAIRS)[0]:.3f} ms (diagnostic build)")
st = ctx.step_stats()
tot = sum(v[2] for v in st.values()) or 1
for k, (steps, rows, cyc) in st.items():
    if steps: print(f"  {k:28s} steps {steps:9d} rows/step {rows / steps:6.1f} cycles/step {cyc / steps:9.0f} share {100 * cyc / tot:5.1f} %")

"""Step kinds of the pair kernel on count-like data (tools/counts_time.py's generator) on the diagnostic build: argv n_feat n_samp [plan]."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "tools", "libicikt_stats.so")
if os.environ.get("ICIKT_LIB") != VARIANT:
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, ICIKT_LIB=VARIANT)).returncode)
import numpy as np, torch
from icikendalltau_amd import _lib
n, S = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(21)
mu = np.exp(rng.normal(2.0, 2.0, size=n))[:, None] * np.exp(rng.normal(0.0, 0.3, size=S))[None, :]
X = rng.negative_binomial(2.0, 2.0 / (2.0 + mu)).astype(np.float64); X[X == 0] = np.nan
X = np.asfortranarray(X)
ctx = _lib.Context(0)
if len(sys.argv) > 3: ctx.debug_set_plan(sys.argv[3])
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0)
ctx.step_stats(reset=True)
ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
print(f"counts {n} x {S}: K1 {ctx.kernel_ms(_lib.K_PAIRS)[0]:.3f} ms (diagnostic build)")
st = ctx.step_stats()
tot = sum(v[2] for v in st.values()) or 1
for k, (steps, rows, cyc) in st.items():
    if steps: print(f"  {k:28s} steps {steps:9d} rows/step {rows / steps:6.1f} cycles/step {cyc / steps:9.0f} share {100 * cyc / tot:5.1f} %")

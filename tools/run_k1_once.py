"""One prepare + a few K1 launches of a bench.py workload, for rocprofv3 counter passes (bench.py spawns it under
`rocprofv3 --pmc ...`, tools/pmc_*.sh do too).  CONFIG = c2 | c3 | c4 | c5 selects bench.py's generator parameters;
N_FEAT / N_SAMP / N_NA / SEED / MAX_PAIRS / PLAN / REPS override.  Prints one JSON line: the K1 time per launch as
HIP events see it IN THIS (profiled) process -- the time the counters of the same pass belong to."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import CONFIGS, make_matrix, workload_matrix
cfg = dict(CONFIGS[os.environ.get("CONFIG", "c4")])
n = int(os.environ.get("N_FEAT", cfg["n_feat"]))
S = int(os.environ.get("N_SAMP", cfg["n_samp"]))
na = int(os.environ.get("N_NA", cfg["n_na"] if "N_FEAT" not in os.environ else n // 10))
seed = int(os.environ.get("SEED", cfg["seed"]))
if CONFIGS[os.environ.get("CONFIG", "c4")].get("fixture"):
    X = workload_matrix(os.environ["CONFIG"], cfg)       # c2: the yeast matrix (no generator parameters)
    n, S = X.shape
else:
    X = make_matrix(n, S, na, seed, int(os.environ.get("LEVELS", cfg.get("levels", 0))))
ctx = _lib.Context(0)
ctx.debug_set_plan(os.environ.get("PLAN", ""))  # e.g. PLAN="np=1,wpb=4"
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = min(S * (S - 1) // 2, int(os.environ.get("MAX_PAIRS", "1000000000")))
ctx.set_pairs_combn(S, 0, P)
out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
ctx.prepare_dev(dX.data_ptr(), n, S, n, 0); ctx.sync()
reps = int(os.environ.get("REPS", "2"))
ctx.run_dev(1, 0, False, 0, out.data_ptr()); ctx.sync()      # warm
ctx.reset_timers()
for _ in range(reps):
    ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
ms, k = ctx.kernel_ms(_lib.K_PAIRS)
print(json.dumps({"k1_ms_per_launch": ms / max(k, 1), "launches": int(k) + 1, "pairs_per_launch": int(P), "n_feat": n, "n_samp": S}))

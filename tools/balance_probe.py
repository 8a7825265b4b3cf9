"""Equal-count against cost-weighted pair blocks of the in-library multi-device driver (development aid, one-GPU box:
rank threads on device 0 with device copies in place of the collectives -- the ranks SHARE the GPU, so their phase times
are not those of separate devices; what the run shows is the blocks and each rank's share of the pair-kernel work).
A matrix whose second half is tied: icikt_multi_rank_phase_ms per rank, ICIKT_FLAG_TIMING (phases end with a stream
synchronisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icikendalltau_amd import _lib
n, S, G = 10000, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(5)
X = rng.standard_normal((n, S))
X[:, S // 2:] = np.round(X[:, S // 2:] * 170)      # ~1 000 distinct values: the most expensive columns to stream
X[rng.random(X.shape) < 0.05] = np.nan
X = np.asfortranarray(X)
m = _lib.MultiContext([0] * G)
one = _lib.Context(0)
# the cost of each block measured one at a time on the whole GPU: the pair kernel alone over the block's pairs
import torch
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
P = S * (S - 1) // 2
for name, fl in (("equal count", 0), ("equal cost", _lib.FLAG_BALANCE_COST)):
    m.pairs(X, perspective="global", want_counts=False, flags=fl)
    m.pairs(X, perspective="global", want_counts=False, flags=fl | _lib.FLAG_TIMING)
    b = m.block_bounds()
    ks = []
    out = torch.empty((max(y - x for x, y in zip(b, b[1:])), 4), dtype=torch.float64, device="cuda")
    one.prepare_dev(dX.data_ptr(), n, S, n, 0)
    for x, y in zip(b, b[1:]):
        one.set_pairs_combn(S, x, y)
        one.run_dev(1, 0, False, 0, out.data_ptr()); one.sync()
        one.reset_timers()
        one.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); one.sync()
        ks.append(one.kernel_ms(_lib.K_PAIRS)[0])
    print(f"{name}: blocks {[y - x for x, y in zip(b, b[1:])]} pairs; pair kernel of each block ALONE on the GPU: "
          + " ".join(f"{k:.2f}" for k in ks) + f" ms (max / mean {max(ks) / (sum(ks) / len(ks)):.2f}); "
          f"rank threads sharing the GPU, pairs phase: " + " ".join(f"{r['pairs']:.2f}" for r in m.rank_phase_ms()), flush=True)

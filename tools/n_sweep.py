"""K1 throughput against column length for the default plan and for forced plans (development aid): one pair per wave;
two pairs on the whole wave (half=0: also where the half-wave kernels are the default) with the singleton region in the
half layout (the default) and with the previous whole-wave loop (hyb=0); workgroups of 1 / 3 waves instead of 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icikendalltau_amd import _lib
from bench import make_matrix

ns = [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000, 16000, 17000, 25000, 26000, 40000, 65535]
variants = [dict(), dict(np="1"), dict(half="0", np="2"), dict(half="0", np="2", hyb="0"), dict(half="0", np="2", wpb="1"), dict(half="0", np="2", wpb="3")]
if os.environ.get("VARIANTS"):
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in os.environ["VARIANTS"].split(";")]
ctx = _lib.Context(0)
for n in ns:
    S = 512
    X = make_matrix(n, S, max(1, n // 50), 5)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    P = min(S * (S - 1) // 2, max(20000, int(6e8 / n)))
    ctx.set_pairs_combn(S, 0, P)
    out = torch.empty((P, 4), dtype=torch.float64, device="cuda")
    ctx.prepare_dev(dX.data_ptr(), n, S, n, _lib.FLAG_TIMING); ctx.sync()
    line = [f"n={n:6d} K0/col {ctx.kernel_ms(_lib.K_PREPARE)[0] / S * 1e3:6.1f} us"]
    import time
    t_w = time.time()
    while time.time() - t_w < 1.0:  # warm-up (clocks, first-touch): the first variant used to read 5-15 % low
        ctx.run_dev(1, 0, False, 0, out.data_ptr()); ctx.sync()
    ref = None
    for v in variants:
        ctx.debug_set_plan(v)
        ts = []
        for _ in range(3):
            ctx.reset_timers(); ctx.run_dev(1, 0, False, _lib.FLAG_TIMING, out.data_ptr()); ctx.sync()
            ts.append(ctx.kernel_ms(_lib.K_PAIRS)[0])
        o = out.cpu().numpy()
        if ref is None: ref = o.copy()
        tag = "default" if not v else ",".join(f"{k}={x}" for k, x in v.items())
        line.append(f"{tag} {P / (min(ts) / 1e3):.3e}{'' if np.array_equal(o, ref, equal_nan=True) else ' DIFF'}")
    print("  ".join(line), flush=True)

"""A/B of builds (ICIKT_LIB=<other .so>) on several quick workloads: c4 / c3 K1, yeast, the tie sweep."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "tools", "exp_libA.so")]
for which in [None] + libs + [None] + libs:
    env = dict(os.environ)
    env.pop("ICIKT_LIB", None)
    if which:
        env["ICIKT_LIB"] = which
    print("==", which or "current", flush=True)
    for tool, args in (("quick_time.py", ["c4"]), ("quick_time.py", ["c3"]), ("yeast_time.py", []), ("tie_sweep.py", [])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            if "amdgpu" not in ln:
                print("  ", ln, flush=True)

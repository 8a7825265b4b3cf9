"""A/B of builds of the library on K0 (c4, c3, a 50 000-row matrix and the yeast shape): ICIKT_LIB=<other .so>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "tools", "exp_libA.so")]
for which in [None] + libs + [None] + libs:
    env = dict(os.environ)
    env.pop("ICIKT_LIB", None)
    if which:
        env["ICIKT_LIB"] = which
    print("==", which or "current", flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k0_time.py")], env=env)

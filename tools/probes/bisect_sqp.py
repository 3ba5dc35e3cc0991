"""Diagnostic: SQP-mode persistent throughput (ERK) of a given library build.  usage: bisect_sqp.py [lib ...]"""
import json, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] == "all"):
    libs = sys.argv[1:] if sys.argv[1] != "all" else []
    for l in libs:
        subprocess.run([sys.executable, __file__, l])
    sys.exit(0)
lib = sys.argv[1] if len(sys.argv) > 1 else "cur"
sys.path.insert(0, ROOT)
if lib != "cur":
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", f"libihm2mpc_{lib}.so")
import bench
out = [lib]
for sqp in ("FIXED_STEP", "MERIT_BACKTRACKING"):
    r = bench.rti_throughput(model="fkin6", B=1024, sqp=sqp, persistent=True, steps=300, warmup=20)
    out.append((sqp, round(r["solves_per_s"])))
print(out, flush=True)

"""Diagnostic: launches per step at B = 8192 and B = 1024 (k_qp_wave) for a given build.  usage: b8192.py [lib]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = sys.argv[1] if len(sys.argv) > 1 else "cur"
if lib != "cur":
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", f"libihm2mpc_{lib}.so")
import bench
out = []
for kw in (dict(model="fkin6", B=8192), dict(model="fkin6", B=8192), dict(model="fkin6", B=1024), dict(model="fkin6", B=8192, integrator="IRK")):
    r = bench.rti_throughput(**kw)
    out.append((round(r["solves_per_s"]), round(r["qp_ms"], 3)))
print(lib, out, flush=True)

"""Diagnostic: persistent loop, soft track rows + collocation intervals (k_steps<8,3,1,1,0,1,0>) for a given build.  usage: soft_irk.py [lib]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = sys.argv[1] if len(sys.argv) > 1 else "cur"
if lib != "cur":
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", f"libihm2mpc_{lib}.so")
import bench
for rep in range(3):
    r = bench.rti_throughput(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=200, warmup=20, integrator="IRK")
    print(lib, round(r["solves_per_s"]), flush=True)

"""Diagnostic: the soft / track-row workloads for a given build.  usage: soft_all.py [lib]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = sys.argv[1] if len(sys.argv) > 1 else "cur"
if lib != "cur":
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", f"libihm2mpc_{lib}.so")
import bench
out = []
for kw in (dict(model="fkin6", B=1024, track_rows="soft"), dict(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=200, warmup=20),
           dict(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=200, warmup=20, integrator="IRK"),
           dict(model="fdyn6u", B=8192, terminal_bounds="stage", track_rows="soft", recover=True),
           dict(model="fdyn6u", B=8192, terminal_bounds="stage", track_rows="soft", recover=True, integrator="IRK")):
    r = bench.rti_throughput(**kw)
    out.append(round(r["solves_per_s"]))
print(lib, out, flush=True)

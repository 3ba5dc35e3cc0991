"""A/B of the headline and of the live options between two builds of the library on ONE box, alternating:
    python tools/probes/ab_headline.py tools/probes/bin/libihm2mpc_<commit>.so
Entry points an older build does not export are dropped from the binding table for that run (diagnostic only)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
other = os.path.abspath(sys.argv[1])
CHILD = r'''
import sys, os, json, ctypes
sys.path.insert(0, %r)
from ihm2_amd import _lib
lib = sys.argv[1]
if lib != "product":
    _lib.LIB_PATH = lib
    probe = ctypes.CDLL(lib)
    for name in list(_lib.SYMBOLS):
        if not hasattr(probe, name): del _lib.SYMBOLS[name]
import bench
sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
''' % ROOT
def run(lib, *args):
    out = subprocess.run([sys.executable, "-c", CHILD, lib, *args], capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    if not line:
        return {"error": out.stderr[-400:]}
    d = json.loads(line[-1])
    return {"value": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4)}
for rep in range(3):
    for name, lib in (("product", "product"), ("other", other)):
        print(rep, name, "headline20", run(lib, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"), flush=True)
for rep in range(2):
    for name, lib in (("product", "product"), ("other", other)):
        print(rep, name, "headline500", run(lib, "--no-cpu-baseline", "--no-extras"), flush=True)
        print(rep, name, "live20", run(lib, "--live-options", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"), flush=True)

"""A/B of the live options (SQP x 2, merit line search, IRK) between the product library and another build, alternating on one box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
other = os.path.abspath(sys.argv[1])
def run(lib, *args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--live-options", "--no-cpu-baseline", *args] + (["--lib", lib] if lib else [])
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    if not line: return {"error": out.stderr[-300:]}
    d = json.loads(line[-1]); return {"value": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4)}
for rep in range(3):
    for name, lib in (("product", None), ("other", other)):
        print(rep, name, "live500", run(lib), "live20", run(lib, "--steps", "20", "--warmup", "5"), flush=True)

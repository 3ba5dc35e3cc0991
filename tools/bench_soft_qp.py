#!/usr/bin/env python3
"""QP cost of the soft / track-row constraint tables against the all-hard table (B = 1024 fkin6, launches per step): the figures item 4 of
the round-2 review asks for.  usage: tools/bench_soft_qp.py [--lib path]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if "--lib" in sys.argv:
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import bench
for kw in (dict(), dict(track_rows="soft"), dict(track_rows="soft", persistent=True, steps=100)):
    r = bench.rti_throughput(model="fkin6", B=1024, steps=kw.pop("steps", 30), warmup=5, **kw)
    print(json.dumps({k: r[k] for k in ("track_rows", "persistent", "solves_per_s", "linearize_ms", "qp_ms", "qp_iter_mean", "ok_fraction")}), flush=True)
r = bench.rti_throughput(model="fdyn6u", B=8192, steps=10, warmup=3, terminal_bounds="stage", track_rows="soft", recover=True)
print(json.dumps({k: r[k] for k in ("model", "solves_per_s", "linearize_ms", "qp_ms", "qp_iter_mean", "ok_fraction")}), flush=True)

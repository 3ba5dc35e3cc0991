"""Oracle-only sweep of an ADAPTIVE fraction-to-boundary rule for the interior-point step (NOTES R4: a fixed 0.9999 saves 5 % of the
iterations but loses marginal instances).  Rule: f = max(0.995, 1 - c * mu) with mu the complementarity of the iterate the step leaves
from, so that only the last iterations of a QP step closer to the boundary.  A patched copy of oracle/ihm2_oracle_qp.c (one line) is
compiled into /tmp; the repository's oracle is not touched.  Closed loop of the bench workload, B instances x STEPS control steps.
    python tools/probes/step_fraction_rule.py [B] [STEPS]"""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
OR = os.path.join(ROOT, "oracle")
src = open(os.path.join(OR, "ihm2_oracle_qp.c")).read()
old = "alpha = fmin(1.0, ORC_IPM_STEP_FRACTION * amax); alpha_d = fmin(1.0, ORC_IPM_STEP_FRACTION * amax_d);"
assert src.count(old) == 1
new = "{ double f_ = orc_frac_(mu); alpha = fmin(1.0, f_ * amax); alpha_d = fmin(1.0, f_ * amax_d); }"
helper = """
#include <stdlib.h>
static double orc_frac_(double mu) {
    const char *m = getenv("ORC_FRAC_MIN"), *c = getenv("ORC_FRAC_C"), *x = getenv("ORC_FRAC_MAX");
    double fmin_ = m ? atof(m) : 0.995, cc = c ? atof(c) : 0.0, fmax_ = x ? atof(x) : 0.99999;
    double f = 1.0 - cc * mu; if (cc == 0.0) f = fmin_;
    if (f < fmin_) f = fmin_; if (f > fmax_) f = fmax_; return f; }
"""
i = src.index("#include")
tmp = "/tmp/orc_variant"; os.makedirs(tmp, exist_ok=True)
open(os.path.join(tmp, "ihm2_oracle_qp.c"), "w").write(src[:i] + helper + src[i:].replace(old, new))
files = [os.path.join(tmp, "ihm2_oracle_qp.c")] + [os.path.join(OR, f) for f in
         ("ihm2_oracle_model.c", "ihm2_oracle_rti.c", "ihm2_oracle_track.c", "ihm2_oracle_dyn10.c")]
so = os.path.join(tmp, "liborc_variant.so")
subprocess.check_call(["cc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-std=gnu11", "-I", OR, "-shared", "-o", so] + files + ["-lm"])

from oracle import oracle as orc
orc._LIB_PATH = so                     # the wrapper loads the variant; make still keeps the repository's own library fresh
from conftest import make_ocp, sample_x0
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
ocp = make_ocp()
P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
x0_all = sample_x0(track, B)

def run(env):
    for k in ("ORC_FRAC_MIN", "ORC_FRAC_C", "ORC_FRAC_MAX"): os.environ.pop(k, None)
    os.environ.update(env)
    x, u = orc.stanley_guess(P, track.s_ref, track.kappa_ref, x0_all.copy(), 40, 25)
    pi = lam = None; xcur = x0_all.copy(); its = []; sts = []; hist = []
    for _ in range(STEPS):
        xcur = P.sim_step(xcur, u[:, 0].copy(), 0, 25)
        yref, yref_e = orc.prepare_step(40, xcur, 40.0, x, u)
        out = P.rti_step(x, u, xcur, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out.get("pi"), out.get("lam")
        its.append(out["qp_iter"].copy()); sts.append(out["status"].copy()); hist.append(u[:, 0].copy())
    return np.array(its), np.array(sts), np.array(hist)

base = None
for label, env in (("fixed 0.995 (shipped)", {}), ("fixed 0.999", {"ORC_FRAC_MIN": "0.999"}), ("fixed 0.9999", {"ORC_FRAC_MIN": "0.9999"}),
                   ("max(0.995, 1 - mu)", {"ORC_FRAC_C": "1"}), ("max(0.995, 1 - 0.1 mu)", {"ORC_FRAC_C": "0.1"}),
                   ("max(0.995, 1 - 0.01 mu)", {"ORC_FRAC_C": "0.01"}), ("max(0.995, 1 - 0.001 mu)", {"ORC_FRAC_C": "0.001"}),
                   ("min(0.9999, max(0.995, 1 - 0.01 mu))", {"ORC_FRAC_C": "0.01", "ORC_FRAC_MAX": "0.9999"}),
                   ("min(0.999, max(0.995, 1 - 0.01 mu))", {"ORC_FRAC_C": "0.01", "ORC_FRAC_MAX": "0.999"})):
    it, st, h = run(env)
    if base is None: base = h
    tot = it.sum(axis=0)
    print(f"{label:38s}: iterations/solve {it.mean():5.2f}  slowest instance over the window {tot.max():4d} (mean {tot.mean():6.1f})  "
          f"failed solves {int((st != 0).sum()):4d} of {st.size}  max |u0 - shipped| {np.abs(h - base).max():.2e}", flush=True)

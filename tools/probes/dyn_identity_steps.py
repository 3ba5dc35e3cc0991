"""Six closed-loop steps, dynamic OCP + kinematic plant: persistent loop against launches per step, differences per component."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["IHM2MPC_BLOCK_QP"] = "0"
from conftest import make_ocp, sample_x0
from ihm2_amd.track import track_table
from ihm2_amd.solver import BatchedOcpSolver
track = track_table("fsds_competition_1")
B, M_sim, steps, plant = 70, 30, 6, 0
x0 = sample_x0(track, B, seed=91)
res = []
for persistent in (False, True):
    s = BatchedOcpSolver(make_ocp(model="fdyn6u"), B, track.s_ref, track.kappa_ref)
    s.set_lap_wrap(True); s.set_x0(x0); s.init_guess(); s.step(40.0, model=plant, M_sim=M_sim)
    if persistent:
        h = s.run_steps(40.0, steps, model=plant, M_sim=M_sim, u0_hist=True, x0_hist=True, status_hist=True, qp_iter_hist=True)
    else:
        h = dict(u0=[], x0=[], status=[], qp_iter=[])
        for _ in range(steps):
            s.step(40.0, model=plant, M_sim=M_sim)
            h["u0"].append(s.get_u0()); h["x0"].append(s.get_x0()); h["status"].append(s.get_status()); h["qp_iter"].append(s.get_qp_iter())
        h = {k: np.array(v) for k, v in h.items()}
    res.append((h, s.get_x(), s.get_u(), s.get_qp_residuals()))
    s.free()
(ha, xa, ua, qa), (hb, xb, ub, qb) = res
same = np.all(ha["qp_iter"] == hb["qp_iter"], axis=0) & np.all(np.isin(ha["status"], (0, 2)), axis=0)
print("instances with equal iteration counts and good statuses:", int(same.sum()), "of", B)
du = np.abs(ha["u0"] - hb["u0"])[:, same]
print("u0 history max abs diff per step [u_T, u_delta]:", du.max(axis=1))
dx = np.abs(xa - xb)[same]
print("final iterate max abs diff per state component:", dx.max(axis=(0, 1)))
i = np.unravel_index(np.argmax(dx[..., 6]), dx[..., 6].shape); bi = np.flatnonzero(same)[i[0]]
print("worst T: instance", bi, "stage", i[1], xa[bi, i[1], 6], xb[bi, i[1], 6], "iters", ha["qp_iter"][:, bi], hb["qp_iter"][:, bi], "status", ha["status"][:, bi], "qp_res", qa[bi], qb[bi])

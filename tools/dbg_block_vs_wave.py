import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import make_ocp, sample_x0
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
def run(mode, B, seed, steps=60, n_max=2.0):
    os.environ["IHM2MPC_BLOCK_QP"] = mode
    from ihm2_amd.solver import BatchedOcpSolver
    s = BatchedOcpSolver(make_ocp(n_max=n_max), B, track.s_ref, track.kappa_ref)
    x0 = sample_x0(track, B, seed=seed)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    st, it, u = [], [], []
    for _ in range(steps):
        s.step(40.0, model=-1, M_sim=30)
        st.append(s.get_status().copy()); it.append(s.get_qp_iter().copy()); u.append(s.get_u0().copy())
    s.free()
    return np.array(st), np.array(it), np.array(u)
for B, seed, n_max in ((1, 1, 2.0), (2, 2, 2.0), (7, 3, 0.9), (33, 4, 2.0), (130, 5, 0.9), (256, 6, 2.0)):
    a = run("1", B, seed, n_max=n_max); b = run("0", B, seed, n_max=n_max)
    ok = (a[0] == 0) & (b[0] == 0)
    print(f"B {B:3d} n_max {n_max}: status equal {np.mean(a[0] == b[0]):.4f}  iterations equal {np.mean(a[1] == b[1]):.4f}  status0 {np.mean(a[0] == 0):.3f}/{np.mean(b[0] == 0):.3f}  "
          f"max |du0| over common solved {np.max(np.abs(a[2][ok] - b[2][ok]) / np.maximum(1, np.abs(b[2][ok]))):.2e}  mean its {a[1].mean():.2f}/{b[1].mean():.2f}", flush=True)

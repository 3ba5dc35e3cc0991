"""One forced rank over torch.distributed (nccl): where do the gathered statuses go wrong?  (round 4, fallback carrier)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
if len(sys.argv) > 1 and sys.argv[1] == "native_first":
    # the product's communicator first: librccl of /opt/rocm/lib is mapped before torch is imported (torch then binds ITS c10d to that one)
    from ihm2_amd import _lib
    import ctypes as C
    ident = (C.c_uint8 * 128)()
    print("native unique id rc", _lib.load().ihm2mpc_comm_unique_id(ident))
    print([l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l][:1])
import torch
import torch.distributed as dist

print(sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l)))
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
a = torch.arange(140, dtype=torch.float64, device="cuda").reshape(70, 2) + 0.5
b = torch.arange(70, dtype=torch.int32, device="cuda")
for name, t in (("f64", a), ("i32", b)):
    out = [torch.empty_like(t)]
    dist.all_gather(out, t)
    print(name, "all_gather equal:", bool((out[0] == t).all().item()))

from ihm2_amd import dist as D
from ihm2_amd.solver import BatchedOcpSolver
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd.track import track_table

track = track_table("fsds_competition_1")
B = 70
s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
s.set_x0(sample_x0(track, B, seed=4)); s.init_guess(); s.prepare_step(40.0); st = s.solve()
u0_dev = torch.empty((B, 2), dtype=torch.float64, device="cuda")
st_dev = torch.full((B,), -7, dtype=torch.int32, device="cuda")
print("ptrs", hex(u0_dev.data_ptr()), hex(st_dev.data_ptr()))
s.get_u0_device(u0_dev.data_ptr()); s.get_status_device(st_dev.data_ptr()); s.synchronize()
torch.cuda.synchronize()
print("u0 copy equal:", np.array_equal(u0_dev.cpu().numpy(), s.get_u0()), "status copy equal:", np.array_equal(st_dev.cpu().numpy(), st), st_dev[:6].cpu().numpy(), st[:6])
u0_all = D.all_gather_blocks(u0_dev, B); st_all = D.all_gather_blocks(st_dev, B)
print("gathered u0 equal:", np.array_equal(u0_all.cpu().numpy(), s.get_u0()), "gathered status equal:", np.array_equal(st_all.cpu().numpy(), st))
dist.destroy_process_group()

# the flow of tests/test_native_comm.py::test_fallback_carrier_gathers_the_same_results
os.environ["IHM2_FORCE_DIST"] = "1"; os.environ["MASTER_PORT"] = "29643"
def broken(*a, **k):
    raise OSError("address already in use")
D.NativeComm = broken
ctx = D.RankContext(carrier="auto", backend="nccl")
ctx.attach(s, B)
print("carrier", ctx.carrier, ctx.fallback)
for variant in ("plain", "after barrier", "after max/sum"):
    if variant == "after barrier":
        ctx.barrier(s)
    if variant == "after max/sum":
        print(ctx.max(3.25), ctx.sum(2.0))
    u0, status = ctx.gather_results(s)
    print(variant, "u0 equal:", np.array_equal(u0, s.get_u0()), "status equal:", np.array_equal(status, st), status[:4])
ctx.close()

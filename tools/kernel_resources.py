#!/usr/bin/env python3
"""Register / scratch / occupancy listing of every kernel of libihm2mpc.so (hipcc -Rpass-analysis=kernel-resource-usage on each source
with the Makefile's flags; the QP source once per instantiation set).  usage: tools/kernel_resources.py > profiles/rN/kernel_resources.txt"""
import os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ihm2_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
QP = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
jobs = [(f, []) for f in ("kernels_misc.hip", "kernels_linearize.hip", "kernels_cart.hip", "kernels_dyn10.hip", "kernels_sqp.hip", "kernels_irk.hip")]
jobs += [("kernels_qp.hip", ["-DQP_SET=0"] + QP), ("kernels_qp.hip", ["-DQP_SET=1"] + QP), ("kernels_qp.hip", ["-DQP_SET=2"] + QP)]
for f, extra in jobs:
    out = subprocess.run(BASE + extra + [f], cwd=SRC, capture_output=True, text=True).stderr
    recs, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m: continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"mangled": v}; recs.append(cur)
        elif cur is not None:
            cur.setdefault(k.split(" ")[0], v)
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in recs), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(recs, names):
        if "k_" not in n: continue          # kernels only (device functions called from them are listed by the compiler as well)
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
        print(f"{f:22s} {n:40s} VGPRs {r.get('VGPRs', '?'):>3s}  AGPRs {r.get('AGPRs', '?'):>3s}  scratch {r.get('ScratchSize', '?'):>5s} B/lane  occupancy {r.get('Occupancy', '?')} waves/SIMD")

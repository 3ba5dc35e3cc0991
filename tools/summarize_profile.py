"""Condenses a rocprofv3 run (gpurun_out/<dir>/{trace,pmc_fetch,pmc_write}) into profiles/<round>/<tag>_*.csv|json.

    python tools/summarize_profile.py gpurun_out/prof_v3 profiles/r1 v3 [steps_per_launch]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled when the
per-launch HBM traffic is formed: traffic = 2 * FETCH_SIZE + WRITE_SIZE.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
steps_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else None      # of the persistent loop k_steps (bench.py --steps = --warmup)
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
out = {"source": src, "kernels": {}}
for row in csv.DictReader(open(stats)):
    name = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    out["kernels"].setdefault(name, {}).update(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3,
                                                 min_us=float(row["MinNs"]) / 1e3, max_us=float(row["MaxNs"]) / 1e3,
                                                 pct=float(row["Percentage"]))
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
        meta[name] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                          lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]))
    for name, v in agg.items():
        k = out["kernels"].setdefault(name, {})
        k[counter + "_KiB_per_launch"] = sum(v) / len(v)
        k.update(meta[name])
for name, k in out["kernels"].items():
    if "FETCH_SIZE_KiB_per_launch" in k and "WRITE_SIZE_KiB_per_launch" in k:
        k["hbm_traffic_bytes_per_launch"] = (2 * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024
for name, k in out["kernels"].items():
    if name.startswith("k_steps") and steps_per_launch:
        k["steps_per_launch"] = steps_per_launch
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1, sort_keys=True)
for name, k in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("pct", 0)):
    print(f"{name:28s} avg {k.get('avg_us', 0):10.1f} us  {k.get('pct', 0):6.2f} %  "
          f"traffic {k.get('hbm_traffic_bytes_per_launch', 0) / 1e6:9.1f} MB  lds {k.get('lds', '-')}  vgpr {k.get('vgpr', '-')}+{k.get('agpr', '-')}")

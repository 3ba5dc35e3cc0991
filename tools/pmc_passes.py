"""Runs rocprofv3 --pmc passes of a bench.py command and condenses them into one JSON (mean counter value per kernel launch).

    python tools/pmc_passes.py gpurun_out/pmc_x profiles/r2/x_counters.json -- --steps 30 --warmup 10 --no-cpu-baseline [--per-step-launches]

Counters are collected in their own runs (no trace flags beside --pmc).  Run on the GPU box.
"""
import collections, csv, glob, json, os, subprocess, sys

PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM",
    "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU",
    "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM",
]
ICACHE = ["SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"]
if os.environ.get("PMC_SET") == "icache":        # instruction-cache pass only
    PASSES = ICACHE
elif os.environ.get("PMC_SET") == "all":
    PASSES = PASSES + ICACHE
out_dir, out_json = sys.argv[1], sys.argv[2]
cmd = sys.argv[sys.argv.index("--") + 1:]
res = collections.defaultdict(dict)
for i, p in enumerate(PASSES):
    d = os.path.join(out_dir, f"pass{i}")
    subprocess.run(["rocprofv3", "--pmc"] + p.split() + ["--output-format", "csv", "-d", d, "--", "python3", "bench.py"] + cmd,
                   stdout=open(os.path.join(out_dir, f"pass{i}.log"), "w") if os.makedirs(out_dir, exist_ok=True) is None else None, stderr=subprocess.STDOUT)
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            res[name]["_launches"] = 0
        for (name, c), v in agg.items():
            res[name][c] = sum(v) / len(v)
            res[name]["_launches"] = len(v)
os.makedirs(os.path.dirname(out_json), exist_ok=True)
json.dump({"command": "bench.py " + " ".join(cmd), "passes": PASSES, "kernels": res}, open(out_json, "w"), indent=1, sort_keys=True)
for name, k in res.items():
    if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"] > 0:
        print(name, "VALU-active/wave-cycles %.3f" % (k.get("SQ_ACTIVE_INST_VALU", 0) / k["SQ_WAVE_CYCLES"]), {c: round(v) for c, v in k.items()})

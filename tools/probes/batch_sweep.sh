set -e
for B in 128 256 512 1024; do
IHM2MPC_BLOCK_QP=0 python bench.py --no-extras --no-cpu-baseline --per-step-launches --batch $B > gpurun_out/bs_$B.log 2>&1
IHM2MPC_BLOCK_QP=0 python bench.py --no-extras --no-cpu-baseline --batch $B --steps 20 --warmup 5 > gpurun_out/bp_$B.log 2>&1
done
python - <<'PY'
import json
for B in (128,256,512,1024):
    d=json.loads(open(f"gpurun_out/bs_{B}.log").read().strip().splitlines()[-1]); p=json.loads(open(f"gpurun_out/bp_{B}.log").read().strip().splitlines()[-1])
    print(B, "per-step qp_ms", round(d["roofline"]["qp_ms"],4), "lin_ms", round(d["roofline"]["linearize_ms"],4), "n_ipm", d["roofline"]["n_ipm_mean"], "| persistent ms/step", round(p["ms_per_step"],4), "solves/s", round(p["value"]))
PY

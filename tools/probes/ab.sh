set -e
for i in 1 2 3; do
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --lib tools/probes/libihm2mpc_base.so > gpurun_out/ab_base_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_new_$i.log 2>&1
done
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 --lib tools/probes/libihm2mpc_base.so > gpurun_out/ab_base_500.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 > gpurun_out/ab_new_500.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.log")):
    d=json.loads(open(f).read().strip().splitlines()[-1]);print(f,round(d["value"]),d["ms_per_step"])
PY

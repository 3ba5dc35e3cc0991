set -e
V=$1
for i in 1 2 3; do
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab2_cur_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --lib $V > gpurun_out/ab2_var_$i.log 2>&1
done
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 > gpurun_out/ab2_cur_500.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 --lib $V > gpurun_out/ab2_var_500.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab2_*.log")):
    d=json.loads(open(f).read().strip().splitlines()[-1]);print(f,round(d["value"]),d["ms_per_step"])
PY

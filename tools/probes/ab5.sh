set -e
V=$1
for i in 1 2 3 4 5; do
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab5_cur_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --lib $V > gpurun_out/ab5_var_$i.log 2>&1
done
for i in 1 2; do
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 > gpurun_out/ab5_cur_500_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 500 --warmup 50 --lib $V > gpurun_out/ab5_var_500_$i.log 2>&1
done
python - <<'PY'
import json,glob,statistics
for tag in ("cur","var"):
    v20=[json.loads(open(f).read().strip().splitlines()[-1])["value"] for f in sorted(glob.glob(f"gpurun_out/ab5_{tag}_[1-5].log"))]
    v500=[json.loads(open(f).read().strip().splitlines()[-1])["value"] for f in sorted(glob.glob(f"gpurun_out/ab5_{tag}_500_*.log"))]
    print(tag, "20 steps:", [round(x) for x in v20], "mean", round(statistics.mean(v20)), "| 500:", [round(x) for x in v500])
PY

set -e
for L in "$@"; do
  if [ "$L" = cur ]; then A=""; else A="--lib tools/probes/libihm2mpc_$L.so"; fi
  python bench.py --live-options --no-cpu-baseline $A > gpurun_out/bis_$L.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/bis_$L.log").read().strip().splitlines()[-1]); print("$L", round(d["value"]), d["ms_per_step"])
PY
done

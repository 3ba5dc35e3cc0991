#!/bin/bash
# the other bench lines and the configuration list of a round, ON THE GPU BOX from the repo root:  tools/refresh_config_lines.sh r3
set -u
R=${1:-r3}
S=gpurun_out/lines_$R
mkdir -p $S
python3 bench.py --config 2 --no-cpu-baseline > $S/bench_line_config2.json 2> $S/config2.err
python3 bench.py --config 3 --no-cpu-baseline > $S/bench_line_config3.json 2> $S/config3.err
python3 bench.py --config 4 --no-cpu-baseline > $S/bench_line_config4.json 2> $S/config4.err
python3 bench.py --live-options --no-cpu-baseline > $S/bench_line_live_options.json 2> $S/live.err
python3 bench.py --gpus 2 --device 0 --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $S/bench_line_2ranks_gloo_one_gpu.json 2> $S/2ranks.err
python3 tools/measure_configs.py > $S/configs_measured.jsonl 2> $S/measure.err
wc -c $S/*.json $S/*.jsonl

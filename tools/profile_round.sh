#!/bin/bash
# rocprofv3 evidence of a round, run ON THE GPU BOX from the repo root (the profiler gets `python3 bench.py ...` itself after `--`):
#   tools/profile_round.sh r2
# trace pass (kernel durations), two PMC passes for the HBM traffic (FETCH_SIZE, WRITE_SIZE; each in its own run), three PMC
# passes of instruction / wait counters for the persistent loop and for the per-step kernels.  Summaries land in
# gpurun_out/prof_<round>/summary/ (copy them to profiles/<round>/ -- gpurun_out is scratch).
set -u
R=${1:-r3}
D=gpurun_out/prof_$R
S=$D/summary
mkdir -p $S
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 20 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 bench.py $ARGS > $D/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 bench.py $ARGS > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 bench.py $ARGS > $D/write.log 2>&1
python3 tools/summarize_profile.py $D $S bench20 20 > $S/bench20_table.txt 2>&1
python3 tools/pmc_passes.py $D/counters $S/bench20_counters.json -- $ARGS > $S/bench20_counters.txt 2>&1
python3 tools/pmc_passes.py $D/counters_ps $S/perstep_counters.json -- --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-step-launches > $S/perstep_counters.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $S/bench_line_20.json 2> $S/bench_line_20.err
python3 bench.py > $S/bench_line_default.json 2> $S/bench_line_default.err
tail -n 12 $S/bench20_table.txt

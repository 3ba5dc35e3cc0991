"""Closed-loop steps with the live solver options of python/main.py:230-237 ("SQP", max_iter 2, "MERIT_BACKTRACKING"), for profiling:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sqp -- python3 tools/run_sqp_mode.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import measure_configs as m  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    integ = sys.argv[2] if len(sys.argv) > 2 else "ERK"          # "IRK": the live options as a whole (python/main.py:227-238)
    print(json.dumps(m.rti_throughput(model="fkin6", B=B, sqp="MERIT_BACKTRACKING", steps=40, integrator=integ)))

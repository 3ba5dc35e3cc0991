"""The command lines of bench.py the driver and a user run, on the GPU box: the default line's contract keys, and the N > 1 path with the ranks
bench.py starts ITSELF (`--gpus 2` without a launcher) -- rehearsed on the one device of the box over gloo (`--device 0 --dist-backend gloo`),
which is the same code path as one rank per GPU over RCCL except for the carrier of the three exchanges."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE JSON line, from rank 0
    assert r.stdout.strip().splitlines() == lines, r.stdout[-2000:]      # and nothing else on stdout (RCCL's banner, gloo's notes go to stderr)
    return json.loads(lines[0])


def test_default_line_carries_the_contract_keys():
    d = _run("--steps", "6", "--warmup", "2", "--batch", "128", "--cpu-budget", "1", "--no-extras")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["roofline"]["bound"] and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["unit"] == "TFLOP/s"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert abs(d["value"] - 128 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert d["max_rel_dev_vs_oracle"] < 1e-7 and d["oracle_check"]["status_equal"]             # tolerance 1e-7 relative (north star: 1e-5)


def test_bench_starts_its_own_ranks_and_gathers_in_global_order():
    d = _run("--gpus", "2", "--device", "0", "--dist-backend", "gloo", "--steps", "4", "--warmup", "1", "--batch", "96", "--no-cpu-baseline", "--no-extras")
    col = d["config"]["collective"]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 192 and "gloo" in col["carrier"]
    # who answered is part of the line: two ranks on ONE device, accepted only as the declared rehearsal it is
    assert col["world"] == 2 and len(col["devices"]) == 2 and col["distinct_devices"] == 1 and "rehearsal" in col
    assert sum(d["status_counts"].values()) == 192
    # one global batch, uneven blocks (strong scaling)
    d = _run("--gpus", "2", "--device", "0", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1", "--global-batch", "101", "--no-cpu-baseline", "--no-extras")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and sum(d["status_counts"].values()) == 101


def test_default_carrier_refused_by_rccl_goes_on_over_gloo_and_says_so():
    """Two self-started ranks on ONE device with the default carrier: the id channel and RCCL's bootstrap run between two real processes, then
    `ncclCommInitRank` refuses the duplicate device on both ranks -- the failure path of the default carrier for real -- and the exchanges go over
    gloo on host buffers with the reason in the line (dist.py::RankContext.attach)."""
    d = _run("--gpus", "2", "--device", "0", "--steps", "4", "--warmup", "1", "--batch", "96", "--no-cpu-baseline", "--no-extras")
    col = d["config"]["collective"]
    assert d["n_gpus"] == 2 and col["world"] == 2 and "gloo" in col["carrier"] and "rehearsal" in col
    assert "ncclCommInitRank" in col["fallback_from_rccl_carrier"]
    assert sum(d["status_counts"].values()) == 192


def test_sharded_config3_and_config4_lines():
    d = _run("--config", "3", "--gpus", "2", "--device", "0", "--dist-backend", "gloo", "--batch", "256", "--steps", "3", "--warmup", "2")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and d["success_fraction"] > 0.8
    assert sum(d["status_counts_last_step"].values()) == 512
    d = _run("--config", "4", "--gpus", "2", "--device", "0", "--dist-backend", "gloo", "--batch", "48", "--steps", "25")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 96 and d["cars_alive"] + d["cars_finished"] + d["cars_failed"] == 96
    assert d["cars_alive"] + d["cars_finished"] >= 90


def test_a_rank_without_a_device_fails_the_run():
    """`--gpus 2` on a one-GPU box: rank 1 cannot open device 1 -- the run fails instead of reporting a silent one-GPU result."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    hip.hipGetDeviceCount(ctypes.byref(n))
    if n.value >= 2:
        pytest.skip("more than one device visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_the_launch_line_of_the_driver_prints_one_line_and_nothing_else():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` as the driver
    starts it, both ranks on the one device of the box (`--device 0`, a declared rehearsal): the default carrier finds RCCL refusing the duplicate
    device and takes gloo; stdout is the line alone.  Without `--device` rank 1 has no device and the run fails without a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    tail = ["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "96", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(launch + ["--master-port", "29541", os.path.join(ROOT, "bench.py")] + tail + ["--device", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 192 and d["config"]["collective"]["world"] == 2
    r = subprocess.run(launch + ["--master-port", "29543", os.path.join(ROOT, "bench.py")] + tail, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())
    assert "out of range" in r.stderr

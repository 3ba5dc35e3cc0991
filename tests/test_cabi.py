"""The C-ABI shared library loads and exports every symbol include/ihm2mpc.h declares (no compute:
runs without a GPU), and the product never routes through the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ihm2mpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ihm2mpc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    for must in ("ihm2mpc_create", "ihm2mpc_free", "ihm2mpc_solve", "ihm2mpc_set_x0", "ihm2mpc_get_u0",
                 "ihm2mpc_get_status", "ihm2mpc_set_tracks", "ihm2mpc_set_weights", "ihm2mpc_prepare_step"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from ihm2_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/ihm2mpc.h but not exported"
    assert set(_lib.SYMBOLS) == set(_declared_symbols())      # the ctypes table covers the header, no extras


def test_no_gpu_means_loud_failure_not_fallback():
    import shutil

    from ihm2_amd import _lib

    lib = _lib.load()
    if shutil.which("rocminfo") and os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    cfg = _lib.Config(batch=1, N=10, M=25, model=0, ntracks=1, nknots=10, device=0, nlp_solver_type=0,
                      nlp_solver_max_iter=1, ipm_iter_max=10, dt=0.05, cost_scale_stage=0.05, ipm_tol=1e-6,
                      ipm_mu0=0.03, ipm_tau0=0.1, nlp_tol=1e-6)
    h = ctypes.c_void_p()
    assert lib.ihm2mpc_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"hip" in lib.ihm2mpc_last_error().lower()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "ihm2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_lean_trig_functions_of_the_models_match_libm(tmp_path):
    """The models' sincos / tanh (csrc/model.hpp: fast_sincos, tanh_e) restated in C with the same constants and operation order:
    within 2.5e-16 absolute of libm over +-64 rad / +-200 (tools/probes/check_fast_trig.c)."""
    import subprocess

    src = os.path.join(ROOT, "tools", "probes", "check_fast_trig.c")
    exe = str(tmp_path / "check_fast_trig")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    # the constants in the header are the ones the C check uses
    hdr = open(os.path.join(ROOT, "ihm2_amd", "csrc", "model.hpp")).read()
    for c in ("6.36619772367581382433e-01", "1.57079632679489655800e+00", "6.12323399573676603587e-17", "-1.49738490485916983693e-33",
              "1.58969099521155010221e-10", "-1.13596475577881948265e-11"):
        assert c in hdr and c in open(src).read()


def test_oracle_and_library_share_the_interior_point_constants():
    """The checker and the product define the algorithm's constants independently (the product never includes anything under oracle/): same values."""
    import re

    def define(path, name):
        m = re.search(r"#define\s+%s\s+([0-9.eE+-]+)" % name, open(os.path.join(ROOT, path)).read())
        assert m, (path, name)
        return float(m.group(1))

    assert define("include/ihm2mpc.h", "IHM2MPC_IPM_STEP_FRACTION") == define("oracle/ihm2_oracle.h", "ORC_IPM_STEP_FRACTION")
    assert define("include/ihm2mpc.h", "IHM2MPC_IRK_NEWTON_ITER") == define("oracle/ihm2_oracle.h", "ORC_IRK_NEWTON_ITER")

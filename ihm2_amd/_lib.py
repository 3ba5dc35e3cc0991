"""ctypes binding of ``libihm2mpc.so`` (the C ABI declared in ``include/ihm2mpc.h``).

No PyTorch, no fallback: if the HIP library is missing or fails to load, importing the solver raises.
``cffi`` is not installed in the target image, hence ``ctypes`` (stdlib).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libihm2mpc.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class Config(C.Structure):
    """``ihm2mpc_config`` (include/ihm2mpc.h)."""

    _fields_ = [
        ("batch", C.c_int32), ("N", C.c_int32), ("M", C.c_int32), ("model", C.c_int32),
        ("ntracks", C.c_int32), ("nknots", C.c_int32), ("device", C.c_int32),
        ("nlp_solver_type", C.c_int32), ("nlp_solver_max_iter", C.c_int32), ("ipm_iter_max", C.c_int32),
        ("dt", C.c_double), ("cost_scale_stage", C.c_double), ("ipm_tol", C.c_double),
        ("ipm_mu0", C.c_double), ("ipm_tau0", C.c_double), ("nlp_tol", C.c_double),
        ("integrator_type", C.c_int32), ("sim_integrator_type", C.c_int32),
    ]


# every symbol include/ihm2mpc.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "ihm2mpc_last_error": (C.c_char_p, []),
    "ihm2mpc_version": (C.c_char_p, []),
    "ihm2mpc_group_create": (C.c_int, [C.POINTER(_H), C.c_int32, C.POINTER(_H)]),
    "ihm2mpc_group_allgather_results": (C.c_int, [_H, c_double_p, c_int32_p]),
    "ihm2mpc_group_free": (C.c_int, [_H]),
    "ihm2mpc_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "ihm2mpc_comm_init": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), c_int32_p]),
    "ihm2mpc_comm_allgather_results": (C.c_int, [_H, c_double_p, c_int32_p]),
    "ihm2mpc_comm_allreduce_max": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_comm_info": (C.c_int, [_H, c_int32_p, C.POINTER(C.c_int64)]),
    "ihm2mpc_comm_free": (C.c_int, [_H]),
    "ihm2mpc_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "ihm2mpc_free": (C.c_int, [_H]),
    "ihm2mpc_synchronize": (C.c_int, [_H]),
    "ihm2mpc_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "ihm2mpc_set_tracks": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_build_tracks": (C.c_int, [_H, C.c_int32, c_int32_p, c_double_p, c_double_p]),
    "ihm2mpc_fit_tracks": (C.c_int, [_H, C.c_int32, c_int32_p, c_double_p, C.c_double, c_double_p, c_double_p]),
    "ihm2mpc_get_tracks": (C.c_int, [_H] + [c_double_p] * 5),
    "ihm2mpc_set_track_id": (C.c_int, [_H, c_int32_p]),
    "ihm2mpc_set_weights": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_set_bounds": (C.c_int, [_H] + [c_double_p] * 8),
    "ihm2mpc_set_soft": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_set_path_constraints": (C.c_int, [_H, C.c_int32, C.c_double, C.c_double, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_set_alat_constraint": (C.c_int, [_H, C.c_int32, C.c_double, C.c_double, c_double_p, c_double_p]),
    "ihm2mpc_set_alat_multipliers": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_get_alat_multipliers": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_set_x0": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_x": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_u": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_yref": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_yref_e": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_multipliers": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_set_stage": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_char_p, c_double_p, C.c_int32]),
    "ihm2mpc_get_stage": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_char_p, c_double_p, C.c_int32]),
    "ihm2mpc_init_guess": (C.c_int, [_H, C.c_double]),
    "ihm2mpc_reinit_failed": (C.c_int, [_H, C.c_double]),
    "ihm2mpc_prepare_step": (C.c_int, [_H, C.c_double]),
    "ihm2mpc_solve": (C.c_int, [_H, C.c_int32]),
    "ihm2mpc_compute_control": (C.c_int, [_H, c_double_p, C.c_double, c_double_p, c_int32_p]),
    "ihm2mpc_reserve_history": (C.c_int, [_H, C.c_int32]),
    "ihm2mpc_run_steps": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_double, c_double_p, c_double_p, c_int32_p, c_int32_p]),
    "ihm2mpc_set_sqp_options": (C.c_int, [_H, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, c_double_p]),
    "ihm2mpc_get_sqp_stats": (C.c_int, [_H, c_int32_p, c_double_p]),
    "ihm2mpc_linearize": (C.c_int, [_H]),
    "ihm2mpc_get_linearization": (C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_get_x": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_u": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_u0": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_status": (C.c_int, [_H, c_int32_p]),
    "ihm2mpc_get_qp_iter": (C.c_int, [_H, c_int32_p]),
    "ihm2mpc_sim_step_dyn10": (C.c_int, [_H, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_get_residuals": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_qp_residuals": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_multipliers": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_get_slacks": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_slacks": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_get_timings": (C.c_int, [_H, c_double_p, C.c_int32]),
    "ihm2mpc_set_x0_device": (C.c_int, [_H, C.c_void_p]),
    "ihm2mpc_get_u0_device": (C.c_int, [_H, C.c_void_p]),
    "ihm2mpc_get_x_device": (C.c_int, [_H, C.c_void_p]),
    "ihm2mpc_get_u_device": (C.c_int, [_H, C.c_void_p]),
    "ihm2mpc_get_status_device": (C.c_int, [_H, C.c_void_p]),
    "ihm2mpc_sim_step": (C.c_int, [_H, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_sim_advance": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "ihm2mpc_step": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double]),
    "ihm2mpc_set_active": (C.c_int, [_H, c_int32_p]),
    "ihm2mpc_set_lap_wrap": (C.c_int, [_H, C.c_int32]),
    "ihm2mpc_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "ihm2mpc_host_free": (C.c_int, [C.c_void_p]),
    "ihm2mpc_get_u0_async": (C.c_int, [_H, c_double_p]),
    "ihm2mpc_set_track_geometry": (C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_sim_step_cart": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, c_double_p, c_double_p, c_double_p]),
    "ihm2mpc_project": (C.c_int, [_H, c_double_p, c_double_p, C.c_double, c_double_p]),
    "ihm2mpc_set_cart_state": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_get_cart_state": (C.c_int, [_H, c_double_p, c_double_p]),
    "ihm2mpc_sim_advance_cart": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double]),
    "ihm2mpc_get_x0": (C.c_int, [_H, c_double_p]),
}

_lib = None


class Ihm2mpcError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Fails loudly: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Ihm2mpcError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ihm2_amd/csrc`; the solver has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the library does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise Ihm2mpcError(load().ihm2mpc_last_error().decode())

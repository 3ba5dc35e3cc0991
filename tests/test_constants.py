"""Golden constants: the product's restated car parameters against the values dumped from the
reference's python/constants.py (tests/golden/constants.json, made by tests/golden/make_constants.py)."""
import json
import os

import pytest

from ihm2_amd import constants as K

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "constants.json")))["values"]


@pytest.mark.parametrize("name", sorted(GOLD))
def test_constant_matches_reference(name):
    assert hasattr(K, name), f"constant {name} missing from ihm2_amd.constants"
    assert getattr(K, name) == pytest.approx(GOLD[name], rel=1e-15, abs=0.0)


def test_survey_spot_values():
    # SURVEY.md section 8c, captured by import
    assert K.static_weight == pytest.approx(564.075, rel=1e-15)
    assert K.Ba == pytest.approx(14.550048996411983, rel=1e-15)
    assert K.Ea == pytest.approx(0.5464720000000001, rel=1e-15)
    assert K.rear_weight_distribution == 0.5


def test_car_parameter_records_carry_the_constants():
    """The ``new_python`` records (new_python/controller.py:11-131) filled from the constants this file pins."""
    import numpy as np

    from ihm2_amd import constants as K
    from ihm2_amd.car_params import CarState, default_car_params

    p = default_car_params()
    assert (p.mass, p.yaw_inertia, p.geometry.wheelbase, p.geometry.cog_to_rear_axle) == (K.m, K.I_z, K.wheelbase, K.l_R)
    assert p.tire_params.lateral_pacejka_coefficients.cornering_stiffness == K.BCDa and p.tire_params.longitudinal_pacejka_coefficients.B == K.Bs
    assert p.tire_params.radius == K.R_w and p.drivetrain_params.C_r2 == K.C_r2 and p.actuator_params.steering_time_constant == K.t_delta
    st = CarState.from_array(np.arange(15.0))
    assert st.T == 10 + 11 + 12 + 13 and np.array_equal(st.to_array(), np.arange(15.0)) and st.v == np.hypot(3.0, 4.0)
    x = st.to_frenet_dyn10(1.0, 0.2, 0.05)
    assert x.shape == (15,) and x[6] == st.omega_FL and x[7] == st.omega_FR and x[10] == st.tau_FL and x[14] == st.delta

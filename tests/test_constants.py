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

"""The reference's input-reader tests on their own data files
(tests/sipnet/test_sipnet_infrastructure/testClimInput.c, testParamInput.c; files under
tests/golden/sipnet_infra)."""
import os

import pytest

import sipnet_amd as sa
from sipnet_amd import _lib
from sipnet_amd.config import param_index as pi
from tests import helpers

D = os.path.join(helpers.GOLDEN, "sipnet_infra")
DERIVED = {"psnTMax": 9, "coarseRootAllocation": 47}


@pytest.mark.parametrize("name", ["standard.clim", "with_loc.clim"])
def test_climate_files_with_and_without_location_column(name):
    assert sa.read_clim(os.path.join(D, name)).n_steps == 10


@pytest.mark.parametrize("name", ["multi_loc.clim", "missing_one.clim"])
def test_climate_files_the_reference_rejects(name):
    with pytest.raises(sa.SipnetError) as e:
        sa.read_clim(os.path.join(D, name))
    assert e.value.code == _lib.ERR_INPUT_FILE


@pytest.mark.parametrize("root", ["standard", "with_est"])
def test_parameter_files_read_the_expected_values(root):
    """`name value [estimation columns ignored]`; the .exp file lists name: %.2f of what was read"""
    got, _ = sa.read_params(os.path.join(D, root + ".param"), sa.flags_from())
    n = 0
    for line in open(os.path.join(D, root + ".exp")):
        name, val = line.split(":")
        idx = DERIVED.get(name.strip(), pi(name.strip()))
        assert idx >= 0, name
        assert "%.2f" % got[idx] == val.strip(), name
        n += 1
    assert n == 60


def test_spatially_varying_marker_is_refused():
    """spatial_val.param uses the obsolete `*` value: EXIT_CODE_BAD_PARAMETER_VALUE"""
    with pytest.raises(sa.SipnetError) as e:
        sa.read_params(os.path.join(D, "spatial_val.param"), sa.flags_from())
    assert e.value.code == _lib.ERR_BAD_PARAMETER

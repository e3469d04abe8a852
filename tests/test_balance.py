"""Mass balance on the reference's own balance case (tests/sipnet/test_modeling/testBalance.c with
balance.clim / balance.param / events_leaf.in): |delta C| and |delta N| of every step stay
below 1e-8 (the reference's TEST_EPS) -- oracle on the CPU; on the GPU the strict all-flags
kernel reproduces the oracle's records for the same three configurations."""
import os

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd.config import param_index as pi
from tests import helpers

D = os.path.join(helpers.GOLDEN, "balance")
CONFIGS = {
    "ncycle_computed_leaf": (dict(litterPool=1, nitrogenCycle=1, gdd=0, waterHResp=1, anaerobic=1), False),
    "ncycle_leaf_events": (dict(litterPool=1, nitrogenCycle=1, gdd=0, waterHResp=1, anaerobic=1), True),
    "no_litter_pool": (dict(litterPool=0, nitrogenCycle=0, gdd=0, waterHResp=1, anaerobic=1), False),
}


def load(name):
    kw, leaf_events = CONFIGS[name]
    flags = sa.flags_from(**kw)
    params, _ = sa.read_params(os.path.join(D, "balance.param"), flags)
    clim = sa.read_clim(os.path.join(D, "balance.clim"), gdd=flags[1])
    events = []
    if leaf_events:                       # testBalanceLeafEvents: day-of-year phenology switched off
        params[pi("leafOnDay")] = 0
        params[pi("leafOffDay")] = 0
        events = sa.read_events(os.path.join(D, "events_leaf.in"), flags, params)
    return flags, params, clim, events


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_conserves_carbon_and_nitrogen(oracle, name):
    flags, params, clim, events = load(name)
    st, rec, diag = oracle.run_member(flags, params, clim, events)
    assert st == 0 and np.isfinite(rec).all()
    assert diag.max_abs_dC < 1e-8 and diag.max_abs_dN < 1e-8 and diag.n_balance_warn == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_gpu_strict_kernel_on_the_balance_case(oracle, name):
    flags, params, clim, events = load(name)
    st, want, _ = oracle.run_member(flags, params, clim, events)
    assert st == 0
    b = sa.Batch(flags, 1, 1, sa.F64, fast_math=False)
    b.set_events(0, events)
    b.set_climate(0, clim)
    b.set_params(0, params)
    b.setup()
    _, rec = b.run(full=True)
    got = rec.cpu().numpy()[:, :36, 0]
    b.close()
    scale = np.maximum(np.abs(want).max(0), 1e-9)
    assert (np.abs(got - want) / scale).max() < 1e-11

"""Pins the CPU oracle (oracle/sipnet_oracle.c) -- and with it the product's host
parsers and `.out` formatter -- against the reference's own golden vectors:

  * tests/smoke/{niwot,russell_1,russell_2,russell_3}: committed sipnet.out / events.out
    (+ russell_4, skipped upstream: expected outputs from the reference binary built here)
    of the reference repository, byte for byte;
  * full-precision per-step records produced by running the REAL reference step loop
    (oracle/_ref, tools/make_golden.py) -- exact equality in every captured column;
  * a 16-member synthetic ensemble and an all-optional-flags run, same source.

CPU only.  The oracle is test infrastructure; nothing here touches the GPU path.
"""
import os

import numpy as np
import pytest

import sipnet_amd as sa
from tests import helpers


@pytest.mark.parametrize("case_name", helpers.SMOKE_CASES)
def test_oracle_reproduces_reference_smoke_goldens(case_name, oracle, tmp_path):
    case = helpers.load_smoke_case(case_name, str(tmp_path))
    ev_out = str(tmp_path / "events.out")
    st, rec, diag = oracle.run_member(case["flags"], case["params"], case["clim"],
                                      case["events"], events_out=ev_out)
    assert st == 0
    txt = helpers.out_text(case["clim"], rec, header=bool(case["cfg"]["printHeader"]))
    assert txt == case["golden_out"], "sipnet.out differs from the reference's committed golden"
    # events.out: the reference prints a header line when PRINT_HEADER is on (events.c:371-378)
    if not case["flags"][0]:       # EVENTS = 0 (russell_4): no events.out at all
        assert not os.path.exists(ev_out)
    else:
        got = open(ev_out, "rb").read()
        gold = case["golden_events"]
        if case["cfg"]["printHeader"] and gold:
            gold = gold.split(b"\n", 1)[1]
        assert got == gold, "events.out differs from the reference's committed golden"
    # the reference's mass-balance check (balance.c:122-169, EPS 1e-8) never fires
    assert diag.n_balance_warn == 0
    assert diag.max_abs_dC < 1e-8 and diag.max_abs_dN < 1e-8


@pytest.mark.parametrize("case_name", helpers.SMOKE_CASES)
def test_oracle_equals_reference_step_loop_exactly(case_name, oracle, tmp_path):
    ref = np.load(os.path.join(helpers.GOLDEN, "ref_smoke_records.npz"))
    case = helpers.load_smoke_case(case_name, str(tmp_path))
    st, rec, _ = oracle.run_member(case["flags"], case["params"], case["clim"], case["events"])
    assert st == 0
    idx = ref[f"{case_name}_idx"]
    assert np.array_equal(rec[idx], ref[f"{case_name}_rec"])
    assert np.array_equal(rec[-1], ref[f"{case_name}_final"])


def _synth_clim(tmp_path):
    p = str(tmp_path / "hh.clim")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "synth", "halfhourly.clim.gz"), p)
    return sa.read_clim(p)


def test_oracle_equals_reference_on_synthetic_special_members(oracle, tmp_path):
    clim = _synth_clim(tmp_path)
    members = np.load(os.path.join(helpers.GOLDEN, "synth", "members_raw.npy"))
    ref = np.load(os.path.join(helpers.GOLDEN, "synth", "ref_synth.npz"))
    planes, final, status = oracle.run_block(sa.flags_from(), members, clim)
    assert (status == 0).all()
    idx = ref["idx"]
    assert np.array_equal(planes[0][idx].T, ref["nee"])
    assert np.array_equal(planes[1][idx].T, ref["gpp"])
    assert np.array_equal(planes[2][idx].T, ref["et"])
    assert np.array_equal(final, ref["final"])
    assert np.array_equal(planes[0].sum(0), ref["sum_nee"]) or \
        np.allclose(planes[0].sum(0), ref["sum_nee"], rtol=0, atol=1e-9)
    # the fixture really exercises the rare branches
    assert final[8, 14] == 0.0 and final[9, 14] == 0.0          # two dead members
    assert ref["snow"].max() > 1.0                              # a snow pack builds up
    assert ref["leaf"][6].max() > 2 * ref["leaf"][6].min()      # deciduous leaf flush / fall


def test_oracle_equals_reference_with_every_optional_flag_on(oracle, tmp_path):
    ref = np.load(os.path.join(helpers.GOLDEN, "synth", "ref_allflags.npz"))
    flags = [int(x) for x in ref["flags"]]
    base = helpers.load_smoke_case("russell_2", str(tmp_path))
    params, _ = sa.read_params(os.path.join(helpers.GOLDEN, "synth", "allflags.param"), flags)
    st, rec, _ = oracle.run_member(flags, params, base["clim"], base["events"])
    assert st == 0
    assert np.array_equal(rec[ref["idx"]], ref["rec"])


def test_oracle_status_codes(oracle, tmp_path):
    """Conditions on which the reference exits become status codes."""
    case = helpers.load_smoke_case("niwot", str(tmp_path))
    clim = case["clim"].slice(0, 50)
    bad = case["params"].copy()
    bad[sa.config.param_index("leafAllocation")] = 0.7
    bad[sa.config.param_index("woodAllocation")] = 0.7   # allocations sum > 1: sipnet.c:1117-1122
    st, _, _ = oracle.run_member(case["flags"], bad, clim)
    assert st == 3
    c2 = sa.ClimTable(clim.data.copy(), clim.year, clim.day)
    c2.data[10, 0] = 0.0                                  # non-positive step length: events.c:460-465
    st, _, _ = oracle.run_member(case["flags"], case["params"], c2)
    assert st == 3


def test_oracle_diagnostics_equal_the_reference_binarys_own_warning_lines(oracle, tmp_path):
    """tests/golden/synth/ref_warning_counts.json holds what the REAL reference printed (not quiet,
    tools/make_golden.py: warning_counts): per member the number of ensureNonNegative() warnings
    (sipnet.c:1346-1356) and of checkBalance() failures (balance.c:149-163) -- on the 16 special
    members over the synthetic year, and on a 1e11-gC stand whose carbon total has an ulp above the
    balance threshold.  The oracle's sipo_diag counters, which the GPU counters are held against
    (tests/test_gpu_full.py), must be those numbers."""
    import json
    from sipnet_amd import synth
    from sipnet_amd.config import param_index as pi
    gold = json.load(open(os.path.join(helpers.GOLDEN, "synth", "ref_warning_counts.json")))
    flags = sa.flags_from()
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "synth", "halfhourly.clim.gz"), str(tmp_path / "hh.clim"))
    clim = sa.read_clim(str(tmp_path / "hh.clim"))
    members = np.load(os.path.join(helpers.GOLDEN, "synth", "members_raw.npy"))
    diags = [oracle.run_member(flags, members[m], clim, want_rec=False)[2] for m in range(members.shape[0])]
    assert [d.n_clamp_warn for d in diags] == gold["special16"]["n_clamp_warn"]
    assert [d.n_balance_warn for d in diags] == gold["special16"]["n_balance_warn"]
    assert sum(gold["special16"]["n_clamp_warn"]) > 10000        # the reference did warn (member 9's snow pack)
    # the heavy stand: 20 days from day 150 of the same generator
    base, _ = sa.read_params(os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param"), flags)
    heavy = synth.perturbed_params(base, 8)
    heavy[:, pi("plantWoodInit")] = 1e11
    raw = synth.half_hourly_year_raw(150 * 48 + 48 * 20)
    raw = {k: v[150 * 48:] for k, v in raw.items()}
    synth.write_clim(str(tmp_path / "d20.clim"), synth.round_like_file(raw))
    clim20 = sa.read_clim(str(tmp_path / "d20.clim"))
    dh = [oracle.run_member(flags, heavy[m], clim20, want_rec=False)[2] for m in range(8)]
    g = gold["wood1e11_20days_from_day150"]
    assert [d.n_balance_warn for d in dh] == g["n_balance_warn"] and min(g["n_balance_warn"]) > 900
    assert [d.n_clamp_warn for d in dh] == g["n_clamp_warn"]

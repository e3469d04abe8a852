"""SIPNET's text formats at the boundary (host_io.cpp), mirroring the reference's
tests/sipnet/test_sipnet_infrastructure/{testClimInput,testParamInput,testOutputHeader}.c
and test_events_infrastructure/."""
import os

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import _lib
from tests import helpers

ROW12 = "2020 10 3.00 0.125 12.5 8.0 20.0 1.5 900.0 300.0 800.0 2.5\n"
ROW14 = "0 2020 10 3.00 0.125 12.5 8.0 20.0 1.5 900.0 300.0 800.0 2.5 0.4\n"


def test_clim_12_columns_and_conversions(tmp_path):
    p = tmp_path / "a.clim"
    p.write_text(ROW12 + "2020 10 6.00 -10800 -3.0 8.0 0.0 0.0 0.0004 300.0 800.0 0.0\n")
    c = sa.read_clim(p)
    assert c.n_steps == 2 and c.year[0] == 2020 and c.day[1] == 10
    r = c.data[0]
    assert r[0] == 0.125 and r[1] == 12.5 and r[2] == 8.0
    assert r[3] == 20.0 * (1.0 / 0.125)           # PAR per day, sipnet.c:216
    assert r[4] == 1.5 * 0.1                       # mm -> cm
    assert r[5] == 900.0 * 0.001 and r[6] == 300.0 * 0.001 and r[7] == 800.0 * 0.001
    assert r[8] == 2.5 and r[9] == 12.5 * 0.125 and r[10] == 3.0
    r = c.data[1]
    assert r[0] == -10800 / -86400.0               # negative length = seconds, sipnet.c:209-211
    assert r[5] == 1e-6 and r[8] == 1e-6           # vpd / wspd clamps to TINY
    assert r[9] == 0.0                             # negative GDD contribution clamps to 0
    assert sa.read_clim(p, gdd=0).data[0, 9] == 0.0


def test_clim_legacy_14_columns(tmp_path):
    p = tmp_path / "b.clim"
    p.write_text(ROW14 * 3)
    c = sa.read_clim(p)
    assert c.n_steps == 3 and c.data[2, 1] == 12.5


def test_clim_errors(tmp_path):
    p = tmp_path / "c.clim"
    p.write_text(ROW14 + ROW14.replace("0 2020", "1 2020", 1))
    with pytest.raises(sa.SipnetError) as e:      # multiple locations: sipnet.c:258-264
        sa.read_clim(p)
    assert e.value.code == _lib.ERR_INPUT_FILE
    p.write_text(ROW12.strip() + " 7\n")          # 13 columns
    with pytest.raises(sa.SipnetError) as e:
        sa.read_clim(p)
    assert e.value.code == _lib.ERR_INPUT_FILE
    p.write_text("")
    with pytest.raises(sa.SipnetError) as e:
        sa.read_clim(p)
    assert e.value.code == _lib.ERR_INPUT_FILE
    with pytest.raises(sa.SipnetError) as e:
        sa.read_clim(tmp_path / "missing.clim")
    assert e.value.code == _lib.ERR_FILE_OPEN
    p.write_text(ROW12 + "2020 11 3.00 0.125 12.5 8.0\n")    # truncated record
    with pytest.raises(sa.SipnetError):
        sa.read_clim(p)


def test_niwot_clim_matches_reference_parse(tmp_path):
    """The reference's legacy-format smoke forcing parses to the same numbers the
    reference holds in memory (captured through the harness into the fixture)."""
    case = helpers.load_smoke_case("niwot", str(tmp_path))
    c = case["clim"]
    assert c.n_steps == 5237
    assert (c.year[0], c.day[0], c.year[-1], c.day[-1]) == (1998, 305, 2005, 365)
    assert abs(c.data[:, 0].min() - 0.292) < 1e-12


BASE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "sipnet_amd", "data", "base_forest.param")


def test_param_file_rules(tmp_path):
    fl = sa.flags_from()
    p, seen = sa.read_params(BASE, fl)
    assert seen.sum() == 58 and p[sa.config.param_index("aMax")] == 8.3
    txt = open(BASE).read()
    f = tmp_path / "p.param"
    f.write_text(txt + "AMAX 9.9 extra columns are ignored 1 2 3\n")          # duplicate (case-insens.)
    with pytest.raises(sa.SipnetError) as e:
        sa.read_params(f, fl)
    assert e.value.code == _lib.ERR_INPUT_FILE
    f.write_text(txt.replace("aMax ", "! aMax "))                            # required missing
    with pytest.raises(sa.SipnetError) as e:
        sa.read_params(f, fl)
    assert e.value.code == _lib.ERR_INPUT_FILE and "aMax" in str(e.value)
    f.write_text(txt + "someFutureParam 3.0\n! comment\n\n")                 # unknown names ignored
    p2, _ = sa.read_params(f, fl)
    assert np.array_equal(p, p2)
    # a parameter only required under a flag: nitrogen cycle needs leafCN & co
    with pytest.raises(sa.SipnetError):
        sa.read_params(BASE, sa.flags_from(litterPool=1, anaerobic=1, nitrogenCycle=1))
    import re
    f.write_text(re.sub(r"(?m)^cFracLeaf .*$", "cFracLeaf 0", txt))          # divisor clamp
    p3, _ = sa.read_params(f, fl)
    assert p3[sa.config.param_index("cFracLeaf")] == 1e-6
    f.write_text(txt + "leafOnDay *\n")
    with pytest.raises(sa.SipnetError):
        sa.read_params(f, sa.flags_from(gdd=0))


def test_events_file(tmp_path):
    fl = sa.flags_from()
    f = tmp_path / "events.in"
    f.write_text("2016 92 irrig 2.8 1\n2016 96 harv 0.1 0.2 0.3 0.4\n2016 96 till 0.5\n"
                 "2016 100 fert 1 2 3\n2016 120 plant 1 2 3 4\n")
    ev = sa.read_events(f, fl)
    assert [e.type for e in ev] == [2, 1, 4, 0, 3]
    assert list(ev[1].p) == [0.1, 0.2, 0.3, 0.4] and ev[0].p[1] == 1.0
    assert sa.read_events(tmp_path / "none.in", fl) == []                    # no file = no events
    f.write_text("")
    assert sa.read_events(f, fl) == []
    f.write_text("2016 96 irrig 1 0\n2016 92 irrig 1 0\n")                   # out of order
    with pytest.raises(sa.SipnetError) as e:
        sa.read_events(f, fl)
    assert e.value.code == _lib.ERR_INPUT_FILE
    f.write_text("2016 96 frobnicate 1\n")
    with pytest.raises(sa.SipnetError) as e:
        sa.read_events(f, fl)
    assert e.value.code == _lib.ERR_UNKNOWN_EVENT
    f.write_text("2016 96 harv 0.9 0.0 0.2 0.0\n")                           # fractions > 1
    with pytest.raises(sa.SipnetError) as e:
        sa.read_events(f, fl)
    assert e.value.code == _lib.ERR_BAD_PARAMETER
    f.write_text("2016 96 leafon\n")                                         # with gdd phenology on
    with pytest.raises(sa.SipnetError) as e:
        sa.read_events(f, fl, np.zeros(80))
    assert e.value.code == _lib.ERR_BAD_PARAMETER
    assert len(sa.read_events(f, sa.flags_from(gdd=0), np.zeros(80))) == 1
    f.write_text("2016 96 leafon 3\n")                                       # takes no parameters
    with pytest.raises(sa.SipnetError):
        sa.read_events(f, sa.flags_from(gdd=0), np.zeros(80))


def test_out_header_and_row_format():
    h = sa.format_out_header()
    assert h.startswith("year day  time plantWoodC plantLeafC woodCreation     soil ")
    assert h.endswith("nUptake      ch4  nppStorage\n") and h.count("\n") == 1
    rec = np.arange(36, dtype=float) / 7.0
    row = sa.format_out_row(1998, 305, 7.0, rec)
    cols = row.split()
    assert len(cols) == 35 and cols[:3] == ["1998", "305", "7.00"]
    assert cols[3] == "%.2f" % (rec[14] + rec[26])        # plantWoodC column is wood + delta
    assert cols[23] == "%.8f" % rec[2]                    # evapotranspiration %18.8f
    assert len(row) == len(sa.format_out_row(2001, 1, 0.0, np.zeros(36)))


def test_config_file(tmp_path):
    for case in helpers.SMOKE_CASES:
        cfg = sa.read_config(os.path.join(helpers.smoke_dir(case), "sipnet.in"))
        assert cfg["dumpConfig"] == 1
    cfg = sa.read_config(os.path.join(helpers.smoke_dir("russell_3"), "sipnet.in"))
    assert (cfg["growthResp"], cfg["leafWater"], cfg["litterPool"], cfg["waterHResp"]) == (1, 1, 1, 0)
    f = tmp_path / "sipnet.in"
    f.write_text("RUNTYPE = montecarlo\n")
    with pytest.raises(ValueError):
        sa.read_config(f)
    f.write_text("file-name: abc ! trailing comment\nLITTER_POOL\t1\nOUT_FILE = none\nLOCATION=0\n")
    cfg = sa.read_config(f)
    assert cfg["filePrefix"] == "abc" and cfg["litterPool"] == 1

"""The reference's event-file parser tests and its events.out format test, on their own data files
(tests/sipnet/test_events_infrastructure/testEventInfra*.c, testEventOutputFile.c;
tests/sipnet/test_bugfixes/testEventFileOrderChecks.c; files under tests/golden/events_infra)."""
import ctypes as C
import os

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import _lib
from sipnet_amd.config import param_index as pi
from tests import helpers

D = os.path.join(helpers.GOLDEN, "events_infra")


def test_parser_accepts_the_good_files():
    fl = sa.flags_from()
    ev = sa.read_events(os.path.join(D, "infra_events_simple.in"), fl)
    assert [(e.year, e.day, e.type) for e in ev] == [(2022, 40, 2), (2022, 40, 0), (2022, 45, 4),
                                                     (2022, 46, 3), (2022, 250, 1)]
    assert list(ev[1].p[:3]) == [15.0, 5.0, 10.0] and ev[2].p[0] == 0.1 and list(ev[4].p) == [0.4, 0.1, 0.2, 0.3]
    assert sa.read_events(os.path.join(D, "infra_events_empty_file.in"), fl) == []
    # day numbers restart with the year: not an ordering violation (testEventFileOrderChecks.c)
    assert len(sa.read_events(os.path.join(D, "infra_events_year_boundary.in"), fl)) == 4


@pytest.mark.parametrize("name,code", [("infra_events_unknown.in", _lib.ERR_UNKNOWN_EVENT),
                                       ("infra_events_date_ooo.in", _lib.ERR_INPUT_FILE),
                                       ("infra_events_bad_first.in", _lib.ERR_INPUT_FILE),
                                       ("infra_events_bad_till.in", _lib.ERR_INPUT_FILE),
                                       ("infra_events_bad_order.in", _lib.ERR_INPUT_FILE)])
def test_parser_rejects_the_bad_files_with_the_reference_exit_code(name, code):
    with pytest.raises(sa.SipnetError) as e:
        sa.read_events(os.path.join(D, name), sa.flags_from())
    assert e.value.code == code


def test_events_out_text_matches_the_reference_goldens(oracle, tmp_path):
    """testEventOutputFile.c: pools 1/2/3/4/5, six records of half a day, first without the litter
    pool and no header, then (pools carried over) with it and the header"""
    L = oracle.lib
    L.sipo_probe_events_series.restype = C.c_int
    year = np.array([2023, 2023, 2023, 2024, 2024, 2024], dtype=np.int32)
    day = np.array([65, 70, 200, 65, 70, 200], dtype=np.int32)
    length = np.full(6, 0.5)
    p = np.zeros(80)
    p[pi("immedEvapFrac")] = 0.5
    names = ("plantWoodC", "plantLeafC", "soilC", "soilWater", "litterC", "snow", "coarseRootC", "fineRootC")
    envi = np.zeros(13)
    for k, v in dict(litterC=1, plantLeafC=2, plantWoodC=3, fineRootC=4, coarseRootC=5).items():
        envi[names.index(k)] = v
    for stem, litter, header in (("events_output_no_header", 0, 0), ("events_output_header", 1, 1)):
        fl = sa.flags_from(litterPool=litter)
        ev = sa.read_events(os.path.join(D, stem + ".in"), fl)
        n, arr = oracle._events(ev)
        out = tmp_path / (stem + ".out")
        rc = L.sipo_probe_events_series((C.c_int * 12)(*fl), p.ctypes.data_as(C.c_void_p),
                                        envi.ctypes.data_as(C.c_void_p), 6, year.ctypes.data_as(C.c_void_p),
                                        day.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p),
                                        n, arr, str(out).encode(), header)
        assert rc == 0
        assert out.read_text() == open(os.path.join(D, stem + ".out")).read()

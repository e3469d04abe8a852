"""Known-answer vectors of the reference's white-box unit tests, replayed against the
oracle's single-function probes (tolerance 1e-6 = tests/utils/tUtils.h:57-59).

Sources: tests/sipnet/test_modeling/testDependencyFunctions.c:109-190,
testSoilMoisture.c:48-161; runmean semantics from src/sipnet/runmean.c:61-116.
"""
import ctypes as C

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd.config import param_index as pi

TOL = 1e-6


def _p(**kw):
    p = np.zeros(80)
    for k, v in kw.items():
        assert pi(k) >= 0, k
        p[pi(k)] = v
    return p


def _fl(**kw):
    return (C.c_int * 12)(*sa.flags_from(**kw))


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture()
def dep_params():
    return _p(soilWHC=10.0, soilRespMoistEffect=2.0, kCN=10.0, soilRespQ10=3.0, fAnoxia=0.75,
              anaerobicDecompRate=0.5, anaerobicTransExp=2.0)


def test_resp_moist_effect_branches(oracle, dep_params):
    L, p = oracle.lib, dep_params
    d = C.c_double
    f = _fl(litterPool=1)
    assert abs(L.sipo_resp_moist_effect(f, _dp(p), d(20.0), d(5.0), d(10.0)) - 0.25) < TOL
    assert abs(L.sipo_resp_moist_effect(_fl(litterPool=1, waterHResp=0), _dp(p), d(20.0), d(5.0), d(10.0)) - 1) < TOL
    assert abs(L.sipo_resp_moist_effect(f, _dp(p), d(-10.0), d(5.0), d(10.0)) - 1) < TOL
    assert abs(L.sipo_resp_moist_effect(f, _dp(p), d(20.0), d(12.0), d(10.0)) - 1.0) < TOL
    assert abs(L.sipo_resp_moist_effect(f, _dp(p), d(20.0), d(-1.0), d(10.0)) - 0.0) < TOL
    fa = _fl(litterPool=1, anaerobic=1)
    assert abs(L.sipo_resp_moist_effect(fa, _dp(p), d(20.0), d(5.0), d(10.0)) - 2.0 / 3.0) < TOL
    p2 = p.copy()
    p2[pi("fAnoxia")] = 0.4
    assert abs(L.sipo_resp_moist_effect(fa, _dp(p2), d(20.0), d(5.0), d(10.0)) - 5.5 / 6.0) < TOL


def test_volatilization_methane_temp_cn(oracle, dep_params):
    L, p = oracle.lib, dep_params
    d = C.c_double
    p2 = p.copy()
    p2[pi("fAnoxia")] = 0.4
    assert abs(L.sipo_volatilization_moist_effect(_dp(p), d(5.0), d(10.0)) - 0.05) < TOL
    assert abs(L.sipo_volatilization_moist_effect(_dp(p2), d(5.0), d(10.0)) - 26.0 / 45.0) < TOL
    assert abs(L.sipo_methane_moist_effect(_dp(p), d(5.0), d(10.0)) - 0.0) < TOL
    assert abs(L.sipo_methane_moist_effect(_dp(p2), d(5.0), d(10.0)) - 1.0 / 36.0) < TOL
    assert abs(L.sipo_temp_effect(_dp(p), d(20.0)) - 9.0) < TOL
    fn = _fl(litterPool=1, anaerobic=1, nitrogenCycle=1)
    assert abs(L.sipo_cn_effect(fn, d(10.0), d(15.0), d(2.0)) - 10.0 / 17.5) < TOL
    assert abs(L.sipo_cn_effect(fn, d(10.0), d(7.5), d(1.5)) - 10.0 / 15.0) < TOL
    assert abs(L.sipo_cn_effect(fn, d(5.0), d(7.5), d(1.5)) - 0.5) < TOL
    assert abs(L.sipo_cn_effect(_fl(), d(10.0), d(15.0), d(2.0)) - 1.0) < TOL


def test_clipped_water_frac(oracle):
    L, d = oracle.lib, C.c_double
    for water, exp in [(5.0, 0.5), (10.0, 1.0), (15.0, 1.0), (0.0, 0.0), (-1.0, 0.0)]:
        assert abs(L.sipo_clipped_water_frac(d(water), d(10.0)) - exp) < TOL


@pytest.fixture()
def water_params():
    return _p(soilWHC=10.0, fastFlowFrac=0.0, waterDrainFrac=1.0, waterRemoveFrac=0.5,
              frozenSoilThreshold=-30.0, frozenSoilEff=1.0, wueConst=10.0, rdConst=100.0,
              rSoilConst1=3.5, rSoilConst2=10.0)


def test_drainage_with_water_drain_frac(oracle, water_params):
    """testSoilMoisture.c:66-121: snow suppresses evaporation; drainage of the 2 cm excess."""
    L, d = oracle.lib, C.c_double
    length, water = 0.125, 12.0
    excess = (water - 10.0) / length
    out = np.zeros(3)

    def drainage(frac, w=water):
        p = water_params.copy()
        p[pi("waterDrainFrac")] = frac
        L.sipo_soil_water_fluxes(_fl(flooding=1), _dp(p), d(length), d(0.5), d(2.0), d(1.0), d(w),
                                 d(0.0), d(0.0), d(0.0), _dp(out))
        return out[2]

    assert abs(drainage(2.0) - excess * length * 2.0) < TOL
    assert abs(drainage(0.5) - excess * length * 0.5) < TOL
    assert abs(drainage(0.0) - 0.0) < TOL
    assert abs(drainage(1.0, w=10.0) - 0.0) < TOL
    assert abs(drainage(20.0) - excess) < TOL


def test_moisture_flooded_soil(oracle, water_params):
    """testSoilMoisture.c:137-161."""
    L, d = oracle.lib, C.c_double
    a, b, c = np.zeros(2), np.zeros(2), np.zeros(2)
    L.sipo_moisture(_dp(water_params), d(20.0), d(50.0), d(1.0), d(10.0), _dp(a))
    L.sipo_moisture(_dp(water_params), d(20.0), d(50.0), d(1.0), d(20.0), _dp(b))
    L.sipo_moisture(_dp(water_params), d(20.0), d(50.0), d(1.0), d(2.0), _dp(c))
    assert abs(a[0] - b[0]) < TOL and abs(a[1] - b[1]) < TOL
    assert abs(c[0] - 1.0) < TOL and abs(a[0] - 50.0 / 10.0 * 1000 * (44 / 12) / 10000) < TOL


def test_light_effect_limits(oracle):
    L, d = oracle.lib, C.c_double
    p = _p(attenuation=0.5, halfSatPar=17.0)
    assert L.sipo_light_eff(_dp(p), d(0.0), d(30.0)) == 0.0
    assert L.sipo_light_eff(_dp(p), d(4.0), d(0.0)) == 0.0
    e1 = L.sipo_light_eff(_dp(p), d(4.0), d(30.0))
    e2 = L.sipo_light_eff(_dp(p), d(4.0), d(60.0))
    assert 0 < e1 < e2 < 1
    # 7-point Simpson of 1 - 2^(-par*exp(-k*lai*x)/h) over x in [0,1]
    x = np.linspace(0, 1, 7)
    y = 1 - 2.0 ** (-30.0 * np.exp(-0.5 * 4.0 * x) / 17.0)
    w = np.array([1, 4, 2, 4, 2, 4, 1]) / 18.0
    assert abs(e1 - float((w * y).sum())) < 1e-12


def test_running_mean_ring_semantics(oracle):
    """runmean.c:61-116: weighted eviction, partial eviction, reset and overflow."""
    L = oracle.lib
    err = C.c_int()

    def mean(vals, wts):
        v, w = np.array(vals, dtype=float), np.array(wts, dtype=float)
        m = L.sipo_ring_probe(len(v), _dp(v), _dp(w), C.byref(err))
        return m, err.value

    # one day of value 10 displaces one fifth of the initial zero mean
    assert abs(mean([10.0], [1.0])[0] - 2.0) < 1e-12
    # five days fully replace it; a sixth evicts the oldest
    assert abs(mean([1, 2, 3, 4, 5], [1] * 5)[0] - 3.0) < 1e-12
    assert abs(mean([1, 2, 3, 4, 5, 6], [1] * 6)[0] - 4.0) < 1e-12
    # partial eviction: weights 2 + 2 + 2 keeps half of the first entry
    assert abs(mean([1, 2, 3], [2, 2, 2])[0] - (1 * 1 + 2 * 2 + 3 * 2) / 5.0) < 1e-12
    # weight >= total replaces everything
    assert abs(mean([1, 2, 9], [1, 1, 7])[0] - 9.0) < 1e-12
    # non-positive weight is an input error, ring untouched
    m, e = mean([4.0, 5.0], [1.0, 0.0])
    assert e == -1 and abs(m - 0.8) < 1e-12
    # 251 entries inside 5 days overflow the 250 slots (sipnet.c:1562-1569 exits 7)
    m, e = mean([1.0] * 260, [0.01] * 260)
    assert e == -2
    # half-hourly forcing (240 entries per 5 days) is the densest legal one
    m, e = mean([1.0] * 1000, [1.0 / 48.0] * 1000)
    assert e == 0 and abs(m - 1.0) < 1e-9

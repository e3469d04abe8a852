"""Synthetic forcing and perturbed-parameter ensembles.

Deterministic generators shared by bench.py, the parity tests and the fixture
script, so that CPU checkers and the GPU engine consume identical inputs.
No reference code involved: this is new workload definition.
"""
import numpy as np

from ._lib import NCLIM, NPARAMS
from .io import ClimTable

SEED_FORCING = 20260821
SEED_PARAMS = 1234


def half_hourly_year(n_steps=17520, site=0, year=2021, seed=SEED_FORCING, gdd=1):
    """Synthetic half-hourly forcing (length = 1800 s) as a converted ClimTable.

    Raw columns are generated in `.clim` units and then converted exactly as
    readClimData does (sipnet.c:201-238), so `write_clim` + the reference binary see
    the same numbers."""
    raw = half_hourly_year_raw(n_steps, site, year, seed)
    return convert_raw(raw, gdd)


def half_hourly_year_raw(n_steps=17520, site=0, year=2021, seed=SEED_FORCING):
    rng = np.random.default_rng(seed + site)
    t = np.arange(n_steps)
    doy = 1 + (t // 48)
    hour = (t % 48) * 0.5
    yr = year + (doy - 1) // 365
    doy = 1 + (doy - 1) % 365
    phase = 0.15 * site  # latitude-like shift between sites
    tair = (5.0 + 12.0 * np.sin(2 * np.pi * (doy - 110) / 365.0 + phase)
            + 6.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0) + rng.normal(0, 1.5, n_steps))
    daily = tair.reshape(-1, 48).mean(1) if n_steps % 48 == 0 else None
    if daily is not None:
        lag = np.concatenate([np.repeat(daily[0], 3), daily[:-3]]) if len(daily) > 3 else daily
        tsoil = np.repeat(np.maximum(0.6 * lag, -2.0), 48)
    else:
        tsoil = np.maximum(0.6 * tair, -2.0)
    decl = 23.45 * np.pi / 180 * np.sin(2 * np.pi * (doy - 81) / 365.0)
    lat = 40.0 * np.pi / 180
    elev = (np.sin(lat) * np.sin(decl)
            + np.cos(lat) * np.cos(decl) * np.cos(2 * np.pi * (hour - 12.0) / 24.0))
    cloud = rng.uniform(0.4, 1.0, n_steps)
    par = np.maximum(0.0, elev) * 0.9 * cloud            # Einstein m-2 per 30 min
    wet = rng.random(n_steps) < 0.06
    precip = np.where(wet, rng.exponential(1.2, n_steps), 0.0)   # mm per step
    rh = rng.uniform(0.4, 0.95, n_steps)
    es = 611.0 * np.exp(17.27 * tair / (tair + 237.3))
    vpd = np.maximum(10.0, es * (1 - rh))
    es_s = 611.0 * np.exp(17.27 * tsoil / (tsoil + 237.3))
    vpd_soil = np.maximum(10.0, es_s * (1 - rh))
    vpress = es * rh
    wspd = np.maximum(0.1, rng.lognormal(0.5, 0.5, n_steps))
    length = np.full(n_steps, -1800.0)
    return dict(year=yr.astype(np.int32), day=doy.astype(np.int32), time=hour, length=length,
                tair=tair, tsoil=tsoil, par=par, precip=precip, vpd=vpd, vpdSoil=vpd_soil,
                vPress=vpress, wspd=wspd)


def round_like_file(raw, fmt_digits=4):
    """Round the raw columns to what `write_clim` prints, so in-memory and file paths agree."""
    out = dict(raw)
    for k in ("tair", "tsoil", "par", "precip", "vpd", "vpdSoil", "vPress", "wspd"):
        out[k] = np.round(raw[k], fmt_digits)
    out["time"] = np.round(raw["time"], 2)
    return out


def convert_raw(raw, gdd=1):
    """readClimData's conversions (sipnet.c:209-238) on raw `.clim` columns."""
    TINY = 0.000001
    length = np.where(raw["length"] < 0, raw["length"] / -86400.0, raw["length"])
    n = len(length)
    d = np.zeros((n, NCLIM))
    d[:, 0] = length
    d[:, 1] = raw["tair"]
    d[:, 2] = raw["tsoil"]
    d[:, 3] = raw["par"] * (1.0 / length)
    d[:, 4] = raw["precip"] * 0.1
    d[:, 5] = np.maximum(raw["vpd"] * 0.001, TINY)
    d[:, 6] = raw["vpdSoil"] * 0.001
    d[:, 7] = raw["vPress"] * 0.001
    d[:, 8] = np.maximum(raw["wspd"], TINY)
    d[:, 9] = np.maximum(raw["tair"] * length, 0.0) if gdd else 0.0
    d[:, 10] = raw["time"]
    return ClimTable(d, raw["year"], raw["day"])


def write_clim(path, raw):
    """Write raw columns as a 12-column `.clim` file (docs/user-guide/model-inputs.md)."""
    with open(path, "w") as fh:
        for i in range(len(raw["year"])):
            fh.write("%d %d %.2f %.1f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n" % (
                raw["year"][i], raw["day"][i], raw["time"][i], raw["length"][i], raw["tair"][i],
                raw["tsoil"][i], raw["par"][i], raw["precip"][i], raw["vpd"][i],
                raw["vpdSoil"][i], raw["vPress"][i], raw["wspd"][i]))


# Parameters perturbed for ensembles: name -> (min, max, sigma), the `changeable`
# rows of the legacy columns of tests/smoke/niwot/sipnet.param (data, not code).
PERTURB = {
    "soilWFracInit": (0.0, 1.0, 0.1), "aMax": (0.0, 34.0, 0.2), "psnTMin": (-8.0, 8.0, 0.5),
    "psnTOpt": (5.0, 30.0, 0.5), "dVpdSlope": (0.01, 0.25, 0.005),
    "halfSatPar": (4.0, 27.0, 5.0), "baseVegResp": (0.0006, 0.06, 0.00002),
    "baseFolRespFrac": (0.05, 0.3, 0.005), "baseFineRootResp": (0.003, 0.6, 0.001),
    "baseCoarseRootResp": (0.003, 0.6, 0.001), "vegRespQ10": (1.4, 2.6, 0.05),
    "fineRootQ10": (1.4, 5.0, 0.05), "coarseRootQ10": (1.4, 5.0, 0.05),
    "frozenSoilThreshold": (-5.0, 5.0, 0.5), "woodTurnoverRate": (0.001, 1.0, 0.001),
    "leafTurnoverRate": (0.001, 1.0, 0.03), "fineRootTurnoverRate": (0.001, 1.0, 0.001),
    "coarseRootTurnoverRate": (0.001, 1.0, 0.001), "wueConst": (0.01, 109.0, 0.5),
    "soilWHC": (0.1, 36.0, 1.0),
}


def perturbed_params(base, n_members, seed=SEED_PARAMS, scale=1.0, names=None):
    """[n_members][80] raw parameter vectors: clip(v + sigma*N(0,1), min, max) on the
    changeable parameters; member 0 is the unperturbed base."""
    from .config import param_index
    rng = np.random.default_rng(seed)
    base = np.asarray(base, dtype=np.float64)
    out = np.tile(base, (n_members, 1))
    for name, (lo, hi, sigma) in PERTURB.items():
        if names is not None and name not in names:
            continue
        k = param_index(name)
        z = rng.standard_normal(n_members)
        v = np.clip(base[k] + scale * sigma * z, lo, hi)
        v[0] = base[k]
        out[:, k] = v
    return out

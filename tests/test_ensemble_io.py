"""Ensemble output block (F4): one NetCDF-3 file for all members, read back unchanged."""
import numpy as np

import sipnet_amd as sa
from sipnet_amd import ensemble_io as eio


def test_netcdf_block_round_trip(tmp_path):
    T, M = 50, 7
    rng = np.random.default_rng(0)
    data = np.zeros((T, 11))
    data[:, 0] = 1.0 / 48
    data[:, 10] = (np.arange(T) % 48) * 0.5
    clim = sa.ClimTable(data, np.full(T, 2021), 1 + np.arange(T) // 48)
    rec = rng.normal(size=(T, 44, M))
    planes = np.stack([rec[:, 0], rec[:, 1], rec[:, 2]])
    p = tmp_path / "ens.nc"
    eio.write_ensemble_netcdf(p, clim, planes=planes, rec=rec, member_ids=np.arange(100, 100 + M),
                              attrs={"site": "synthetic"})
    got = eio.read_ensemble_netcdf(p)
    assert got["nee"].shape == (T, M)
    for name, (idx, _) in eio.OUT_COLUMNS.items():
        want = rec[:, idx[0], :] + rec[:, idx[1], :] if isinstance(idx, tuple) else rec[:, idx, :]
        np.testing.assert_array_equal(got[name], want)
    np.testing.assert_array_equal(got["year"], clim.year)
    np.testing.assert_array_equal(got["hour"], data[:, 10])
    np.testing.assert_array_equal(got["member"], np.arange(100, 100 + M))
    # planes only, single precision
    eio.write_ensemble_netcdf(p, clim, planes=planes.astype(np.float32), dtype="f4")
    got = eio.read_ensemble_netcdf(p)
    assert set(eio.PLANE_NAMES) <= set(got) and got["gpp"].dtype == np.float32
    np.testing.assert_array_equal(got["gpp"], planes[1].astype(np.float32))


def test_column_table_matches_the_out_header():
    """names and order of the `.out` header (sipnet.c:434-452) map onto record columns"""
    hdr = sa.format_out_header().split()
    names = [h for h in hdr if h not in ("year", "day", "time")]
    assert set(names) == set(eio.OUT_COLUMNS)


def test_block_equals_the_text_writer(tmp_path):
    """the same numbers the `.out` text shows, at full precision: compare after formatting"""
    T, M = 12, 3
    rng = np.random.default_rng(1)
    data = np.zeros((T, 11)); data[:, 0] = 0.125; data[:, 10] = (np.arange(T) % 8) * 3.0
    clim = sa.ClimTable(data, np.full(T, 2020), 1 + np.arange(T) // 8)
    rec = np.abs(rng.normal(size=(T, 44, M))) * 10
    p = tmp_path / "e.nc"
    eio.write_ensemble_netcdf(p, clim, rec=rec)
    got = eio.read_ensemble_netcdf(p)
    hdr = [h for h in sa.format_out_header().split()]
    for m in range(M):
        for t in (0, T - 1):
            row = sa.format_out_row(2020, int(clim.day[t]), data[t, 10], rec[t, :, m]).split()
            for name, text in zip(hdr[3:], row[3:]):
                assert abs(float(text) - got[name][t, m]) <= 0.51 * 10 ** -(len(text.split(".")[1]) if "." in text else 0)

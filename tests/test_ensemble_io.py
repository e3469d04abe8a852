"""Ensemble output block (F4): one NetCDF-3 file for all members, written behind the C boundary
(csrc/ensemble_io.cpp), read back unchanged -- by this package's reader AND by scipy's independent one."""
import threading

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import ensemble_io as eio


def _clim(T, per_day=48):
    data = np.zeros((T, 11))
    data[:, 0] = 1.0 / per_day
    data[:, 10] = (np.arange(T) % per_day) * (24.0 / per_day)
    return sa.ClimTable(data, np.full(T, 2021), 1 + np.arange(T) // per_day)


def test_netcdf_block_round_trip(tmp_path):
    T, M = 50, 7
    rng = np.random.default_rng(0)
    clim = _clim(T)
    rec = rng.normal(size=(T, 44, M))
    planes = np.stack([rec[:, 0], rec[:, 1], rec[:, 2]])
    p = tmp_path / "ens.nc"
    eio.write_ensemble_netcdf(p, clim, rec=rec, member_ids=np.arange(100, 100 + M), attrs={"site": "synthetic"})
    got = eio.read_ensemble_netcdf(p)
    assert got["nee"].shape == (T, M)
    for name, (idx, _) in eio.OUT_COLUMNS.items():
        want = rec[:, idx[0], :] + rec[:, idx[1], :] if isinstance(idx, tuple) else rec[:, idx, :]
        np.testing.assert_array_equal(got[name], want)
    np.testing.assert_array_equal(got["year"], clim.year)
    np.testing.assert_array_equal(got["hour"], clim.data[:, 10])
    np.testing.assert_array_equal(got["length"], clim.data[:, 0])
    np.testing.assert_array_equal(got["member"], np.arange(100, 100 + M))
    _, dims, gatts, units = eio.open_ensemble_netcdf(p)
    assert dims == {"time": T, "member": M} and gatts["site"] == "synthetic" and gatts["model_version"] == "2.1.0"
    assert units["nee"] == "g C m-2 step-1" and units["soilWater"] == "cm"
    # planes only, single precision, with a padded leading dimension on the caller's side
    eio.write_ensemble_netcdf(p, clim, planes=planes.astype(np.float32), dtype="f4")
    got = eio.read_ensemble_netcdf(p)
    assert set(eio.PLANE_NAMES) <= set(got) and got["gpp"].dtype == np.float32
    np.testing.assert_array_equal(got["gpp"], planes[1].astype(np.float32))
    # a selection of columns
    eio.write_ensemble_netcdf(p, clim, rec=rec, columns=["soilWater", "plantWoodC"])
    got = eio.read_ensemble_netcdf(p)
    assert [k for k in got if k not in ("year", "day", "hour", "length", "member")] == ["soilWater", "plantWoodC"]
    np.testing.assert_array_equal(got["plantWoodC"], rec[:, 14] + rec[:, 26])


def test_an_independent_reader_agrees(tmp_path):
    """scipy's NetCDF-3 reader knows nothing of this writer: header layout, padding, offsets, byte order"""
    netcdf_file = pytest.importorskip("scipy.io").netcdf_file
    T, M = 37, 5   # odd sizes: 4-byte padding of the int coordinate variables and of names
    rng = np.random.default_rng(3)
    clim = _clim(T)
    rec = rng.normal(size=(T, 44, M))
    for dtype in ("f8", "f4"):
        p = tmp_path / f"e_{dtype}.nc"
        eio.write_ensemble_netcdf(p, clim, rec=rec, dtype=dtype, attrs={"a": "b=c", "note": "x y"})
        mine = eio.read_ensemble_netcdf(p)
        with netcdf_file(str(p), "r", mmap=False) as f:
            assert f.version_byte == 2 and f.dimensions == {"time": T, "member": M}
            assert f.a == b"b=c" and f.note == b"x y"
            assert list(f.variables)[:5] == ["year", "day", "hour", "length", "member"]
            for k, v in f.variables.items():
                np.testing.assert_array_equal(np.array(v[:]), mine[k])
            assert f.variables["gpp"].units == b"g C m-2 step-1"
            assert f.variables["gpp"].dimensions == ("time", "member")


def test_piecewise_puts_from_threads_and_cdf5(tmp_path):
    """member ranges x step ranges in any order, concurrently (what device shards do); the 64-bit-data format"""
    T, M = 64, 40
    rng = np.random.default_rng(5)
    clim = _clim(T)
    a, b = rng.normal(size=(T, M)), rng.normal(size=(T, M)).astype(np.float32)
    L = sa.lib()
    for storage, dt in ((0, np.float64), (1, np.float32), (2, np.float64), (3, np.float32)):
        p = tmp_path / f"p{storage}.nc"
        f = eio.EnsembleFile.__new__(eio.EnsembleFile)
        import ctypes as C
        f._h = C.c_void_p()
        f.names = ["alpha", "nee"]
        arr = (C.c_char_p * 2)(b"alpha", b"nee")
        from sipnet_amd._lib import check
        check(L.sipnet_io_ensemble_create(str(p).encode(), T, M, clim.year.ctypes.data, clim.day.ctypes.data,
                                          clim.data.ctypes.data, None, 2, C.cast(arr, C.c_void_p), None, storage, None,
                                          C.byref(f._h)), "create")
        jobs = [("alpha", a, t0, m0) for t0 in (0, 32) for m0 in (0, 13, 26)] + [("nee", b, t0, m0) for t0 in (32, 0) for m0 in (26, 0, 13)]
        def work(j):
            name, src, t0, m0 = j
            m1 = min(M, m0 + 13) if m0 < 26 else M
            f.put(name, src[t0:t0 + 32, m0:m1], step0=t0, member0=m0)
        th = [threading.Thread(target=work, args=(j,)) for j in jobs]
        [t.start() for t in th]
        [t.join() for t in th]
        f.close()
        assert open(p, "rb").read(4) == b"CDF" + bytes([5 if storage & 2 else 2])
        got = eio.read_ensemble_netcdf(p)
        assert got["alpha"].dtype == dt
        np.testing.assert_array_equal(got["alpha"], a.astype(dt))
        np.testing.assert_array_equal(got["nee"], b.astype(dt))
        _, _, _, units = eio.open_ensemble_netcdf(p)
        assert units["alpha"] == "" and units["nee"] == "g C m-2 step-1"


def test_bad_arguments_are_refused(tmp_path):
    clim = _clim(4)
    with pytest.raises(sa.SipnetError):
        eio.write_ensemble_netcdf(tmp_path / "x.nc", clim, rec=np.zeros((4, 44, 2)), columns=["nope"])
    with pytest.raises(sa.SipnetError):
        eio.write_ensemble_netcdf(tmp_path / "no_such_dir" / "x.nc", clim, planes=np.zeros((3, 4, 2)))
    with pytest.raises(sa.SipnetError):
        eio.EnsembleFile(tmp_path / "y.nc", clim, 2, ["nee", "nee"])


def test_column_table_matches_the_out_header():
    """names and order of the `.out` header (sipnet.c:434-452) map onto record columns"""
    hdr = sa.format_out_header().split()
    names = [h for h in hdr if h not in ("year", "day", "time")]
    assert names == list(eio.OUT_COLUMNS)


def test_block_equals_the_text_writer(tmp_path):
    """the same numbers the `.out` text shows, at full precision: compare after formatting"""
    T, M = 12, 3
    rng = np.random.default_rng(1)
    clim = _clim(T, per_day=8)
    rec = np.abs(rng.normal(size=(T, 44, M))) * 10
    p = tmp_path / "e.nc"
    eio.write_ensemble_netcdf(p, clim, rec=rec)
    got = eio.read_ensemble_netcdf(p)
    hdr = [h for h in sa.format_out_header().split()]
    for m in range(M):
        for t in (0, T - 1):
            row = sa.format_out_row(2021, int(clim.day[t]), clim.data[t, 10], rec[t, :, m]).split()
            for name, text in zip(hdr[3:], row[3:]):
                assert abs(float(text) - got[name][t, m]) <= 0.51 * 10 ** -(len(text.split(".")[1]) if "." in text else 0)

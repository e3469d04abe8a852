"""Ensemble output block: every member's outputs in ONE self-describing file.

The reference writes one `<prefix>.out` text file per process (sipnet.c:434-473), so a
10 k-member PEcAn ensemble means 10 k directories of text that `model2netcdf.SIPNET` parses
again.  The block is written behind the C boundary (`sipnet_io_write_ensemble_block`,
`sipnet_io_ensemble_create / _put / _close`, csrc/ensemble_io.cpp: NetCDF-3 classic by hand,
CDF-2 or CDF-5) and by `sipnet --ensemble-out`; this module binds the writer and READS the
format -- a small parser of its own (classic header + big-endian arrays through numpy.memmap),
so that blocks beyond CDF-2's 4 GiB per variable (CDF-5) can be read too.
"""
import ctypes as C
import struct

import numpy as np

from ._lib import NCLIM, NREC, check, lib

PLANE_NAMES = ("nee", "gpp", "evapotranspiration")


def out_columns():
    """`.out` column -> (record index or (index, index), units), order of outputHeader (sipnet.c:434-444);
    the table lives in the C library (sipnet_io_out_column)"""
    L = lib()
    out = {}
    for k in range(L.sipnet_io_out_column_count()):
        name, units = C.c_char_p(), C.c_char_p()
        r0, r1 = C.c_int32(), C.c_int32()
        check(L.sipnet_io_out_column(k, C.byref(name), C.byref(r0), C.byref(r1), C.byref(units)), "out_column")
        out[name.value.decode()] = ((r0.value, r1.value) if r1.value >= 0 else r0.value, units.value.decode())
    return out


OUT_COLUMNS = out_columns()


def _attrs_text(attrs):
    return None if not attrs else "\n".join(f"{k}={v}" for k, v in attrs.items()).encode()


def write_ensemble_netcdf(path, clim, planes=None, rec=None, columns=None, member_ids=None,
                          attrs=None, dtype="f8"):
    """planes[3][T][M] (NEE, GPP, ET) or rec[T][44][M] (full records; `columns` selects names
    from OUT_COLUMNS, default all) -> NetCDF-3 file, through sipnet_io_write_ensemble_block.
    Arrays may be numpy or torch (they are brought to the host)."""
    to_np = lambda x: x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    T = clim.n_steps
    pl = rc = None
    if rec is not None:
        rc = np.ascontiguousarray(to_np(rec), dtype=np.float64)
        assert rc.shape[0] == T and rc.shape[1] == NREC, rc.shape
        M = rc.shape[2]
    else:
        assert planes is not None, "nothing to write"
        pl = np.ascontiguousarray(to_np(planes), dtype=np.float64)
        assert pl.shape[0] == 3 and pl.shape[1] == T
        M = pl.shape[2]
    ids = None if member_ids is None else np.ascontiguousarray(member_ids, dtype=np.int32)
    cols = None if not columns else ",".join(columns).encode()
    check(lib().sipnet_io_write_ensemble_block(
        str(path).encode(), T, M, clim.year.ctypes.data, clim.day.ctypes.data, clim.data.ctypes.data,
        None if ids is None else ids.ctypes.data, None if pl is None else pl.ctypes.data,
        None if rc is None else rc.ctypes.data, M, cols, 1 if dtype == "f4" else 0, _attrs_text(attrs)),
        "write_ensemble_block")


class EnsembleFile:
    """sipnet_io_ensemble_create / _put / _close: a block filled piece by piece (any step range x member
    range of any variable, in any order)."""

    def __init__(self, path, clim, n_members, names, units=None, member_ids=None, attrs=None, dtype="f8"):
        L = lib()
        self._h = C.c_void_p()
        self.names = list(names)
        arr = (C.c_char_p * len(self.names))(*[n.encode() for n in self.names])
        uarr = None if units is None else (C.c_char_p * len(self.names))(*[None if u is None else u.encode() for u in units])
        ids = None if member_ids is None else np.ascontiguousarray(member_ids, dtype=np.int32)
        check(L.sipnet_io_ensemble_create(str(path).encode(), clim.n_steps, n_members, clim.year.ctypes.data,
                                          clim.day.ctypes.data, clim.data.ctypes.data,
                                          None if ids is None else ids.ctypes.data, len(self.names),
                                          C.cast(arr, C.c_void_p), None if uarr is None else C.cast(uarr, C.c_void_p),
                                          1 if dtype == "f4" else 0, _attrs_text(attrs), C.byref(self._h)),
              "ensemble_create")

    def put(self, name, data, step0=0, member0=0):
        a = np.asarray(data)
        a = np.ascontiguousarray(a, dtype=np.float32 if a.dtype == np.float32 else np.float64)
        check(lib().sipnet_io_ensemble_put(self._h, self.names.index(name), step0, a.shape[0], member0, a.shape[1],
                                           a.ctypes.data, a.shape[1], 1 if a.dtype == np.float32 else 0), "ensemble_put")

    def close(self):
        if self._h:
            check(lib().sipnet_io_ensemble_close(self._h), "ensemble_close")
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- reader: NetCDF classic (CDF-1 / CDF-2 / CDF-5), fixed-size variables ------------------------------------------
_NC_TYPES = {1: ">i1", 2: "S1", 3: ">i2", 4: ">i4", 5: ">f4", 6: ">f8"}


class _Cursor:
    def __init__(self, buf, wide):
        self.b, self.p, self.wide = buf, 0, wide

    def u32(self):
        v = struct.unpack_from(">I", self.b, self.p)[0]
        self.p += 4
        return v

    def u64(self):
        v = struct.unpack_from(">Q", self.b, self.p)[0]
        self.p += 8
        return v

    def count(self):
        return self.u64() if self.wide else self.u32()

    def name(self):
        n = self.count()
        s = bytes(self.b[self.p:self.p + n]).decode()
        self.p += (n + 3) & ~3
        return s

    def attrs(self):
        tag, n = self.u32(), self.count()
        out = {}
        assert tag in (0, 12), tag
        for _ in range(n):
            key = self.name()
            typ, cnt = self.u32(), self.count()
            dt = np.dtype(_NC_TYPES[typ])
            raw = bytes(self.b[self.p:self.p + cnt * dt.itemsize])
            self.p += (cnt * dt.itemsize + 3) & ~3
            out[key] = raw.decode() if typ == 2 else np.frombuffer(raw, dtype=dt).astype(dt.newbyteorder("="))
        return out


def open_ensemble_netcdf(path):
    """-> (variables: name -> read-only big-endian memmap of the variable, dims, global attributes, units)"""
    with open(path, "rb") as f:
        head = f.read(4)
        assert head[:3] == b"CDF" and head[3] in (1, 2, 5), "not a NetCDF classic file"
    version = head[3]
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    c = _Cursor(mm, wide=version == 5)
    c.p = 4
    numrecs = c.count()
    assert numrecs == 0, "record variables are not used by this format"
    tag, n = c.u32(), c.count()
    assert tag in (0, 10)
    dims = []
    for _ in range(n):
        nm = c.name()
        dims.append((nm, c.count()))
    gatts = c.attrs()
    tag, n = c.u32(), c.count()
    assert tag in (0, 11)
    variables, units = {}, {}
    for _ in range(n):
        nm = c.name()
        rank = c.count()
        shape = tuple(dims[c.count()][1] for _ in range(rank))
        va = c.attrs()
        typ = c.u32()
        c.count()                                   # vsize
        begin = c.u64() if version != 1 else c.u32()
        dt = np.dtype(_NC_TYPES[typ])
        variables[nm] = np.memmap(path, dtype=dt, mode="r", offset=begin, shape=shape)
        units[nm] = va.get("units", "")
    return variables, dict(dims), gatts, units


def read_ensemble_netcdf(path):
    """-> dict name -> array in native byte order (copied out of the file)"""
    variables, _, _, _ = open_ensemble_netcdf(path)
    return {k: np.array(v).astype(v.dtype.newbyteorder("=")) for k, v in variables.items()}

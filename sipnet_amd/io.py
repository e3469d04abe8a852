"""SIPNET text formats through the C-ABI host functions (host_io.cpp)."""
import ctypes as C

import numpy as np

from ._lib import NCLIM, NPARAMS, NREC, Event, check, lib


class ClimTable:
    """Converted climate of one site: data[n][11] = length tair tsoil par precip vpd
    vpdSoil vPress wspd gdd time (units as after readClimData, sipnet.c:201-238)."""

    def __init__(self, data, year, day):
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.year = np.ascontiguousarray(year, dtype=np.int32)
        self.day = np.ascontiguousarray(day, dtype=np.int32)
        assert self.data.shape == (len(self.year), NCLIM)

    @property
    def n_steps(self):
        return len(self.year)

    def slice(self, a, b):
        return ClimTable(self.data[a:b], self.year[a:b], self.day[a:b])


def read_clim(path, gdd=1):
    L = lib()
    h = C.c_void_p()
    check(L.sipnet_io_read_clim(str(path).encode(), int(gdd), C.byref(h)), "read_clim")
    try:
        n = L.sipnet_clim_nsteps(h)
        data = np.ctypeslib.as_array(L.sipnet_clim_data(h), shape=(n, NCLIM)).copy()
        year = np.ctypeslib.as_array(L.sipnet_clim_year(h), shape=(n,)).copy()
        day = np.ctypeslib.as_array(L.sipnet_clim_day(h), shape=(n,)).copy()
    finally:
        L.sipnet_clim_free(h)
    return ClimTable(data, year, day)


def read_params(path, flags):
    """-> (raw[80] float64, is_read[80] int32)"""
    L = lib()
    fl = (C.c_int32 * 12)(*flags)
    out = np.zeros(NPARAMS)
    seen = np.zeros(NPARAMS, dtype=np.int32)
    check(L.sipnet_io_read_params(str(path).encode(), fl, out.ctypes.data, seen.ctypes.data),
          "read_params")
    return out, seen


def read_events(path, flags, params=None):
    """-> list of Event"""
    L = lib()
    fl = (C.c_int32 * 12)(*flags)
    ptr = C.POINTER(Event)()
    n = C.c_int32(0)
    pp = None if params is None else np.ascontiguousarray(params, dtype=np.float64).ctypes.data
    check(L.sipnet_io_read_events(str(path).encode(), fl, pp, C.byref(ptr), C.byref(n)),
          "read_events")
    evs = []
    for i in range(n.value):
        e = Event()
        C.memmove(C.byref(e), C.byref(ptr[i]), C.sizeof(Event))
        evs.append(e)
    if n.value:
        L.sipnet_io_free(ptr)
    return evs


def format_out_header():
    buf = C.create_string_buffer(1024)
    n = lib().sipnet_io_format_out_header(buf, 1024)
    return buf.raw[:n].decode()


def format_out_row(year, day, time, rec):
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    assert rec.ndim == 1 and rec.shape[0] >= 36    # the 36 output columns (+ optional event log)
    buf = C.create_string_buffer(1024)
    n = lib().sipnet_io_format_out_row(buf, 1024, int(year), int(day), float(time),
                                       rec.ctypes.data, 1)
    return buf.raw[:n].decode()


def write_out(path, clim, rec, print_header=False):
    rec = np.asarray(rec, dtype=np.float64)
    if rec.shape[1] < NREC:   # records without the event-log columns
        rec = np.concatenate([rec, np.zeros((rec.shape[0], NREC - rec.shape[1]))], axis=1)
    rec = np.ascontiguousarray(rec)
    assert rec.shape == (clim.n_steps, NREC)
    check(lib().sipnet_io_write_out(str(path).encode(), int(print_header), clim.n_steps,
                                    clim.year.ctypes.data, clim.day.ctypes.data,
                                    clim.data.ctypes.data, rec.ctypes.data), "write_out")


def write_debug_logs(prefix, clim, rec, dbg, print_header=False):
    """`<prefix>_envi.log`, `_fluxes.log`, `_trackers.log` of one member (debug_log.c:181-312)
    from its records rec[n_steps][NREC] and debug plane dbg[n_steps][NDBG]."""
    from ._lib import NDBG
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    dbg = np.ascontiguousarray(dbg, dtype=np.float64)
    assert rec.shape == (clim.n_steps, NREC) and dbg.shape == (clim.n_steps, NDBG)
    check(lib().sipnet_io_write_debug_logs(str(prefix).encode(), int(print_header), clim.n_steps,
                                           clim.year.ctypes.data, clim.day.ctypes.data,
                                           clim.data.ctypes.data, rec.ctypes.data,
                                           dbg.ctypes.data), "write_debug_logs")


def write_events_out(path, flags, raw_params, clim, events, rec, init_pools, print_header=False):
    """`events.out` of one member regenerated from its full records (events.c:369-418)."""
    import ctypes as C
    from ._lib import Event
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    assert rec.shape == (clim.n_steps, NREC)
    raw = np.ascontiguousarray(raw_params, dtype=np.float64)
    pools = np.ascontiguousarray(init_pools, dtype=np.float64)
    fl = (C.c_int32 * 12)(*flags)
    n = len(events)
    arr = (Event * max(n, 1))(*events)
    check(lib().sipnet_io_write_events_out(str(path).encode(), int(print_header), fl, raw.ctypes.data,
                                           clim.n_steps, clim.year.ctypes.data, clim.day.ctypes.data,
                                           clim.data.ctypes.data, n, arr, rec.ctypes.data,
                                           pools.ctypes.data), "write_events_out")


def read_restart(path):
    """Parse a `SIPNET_RESTART` checkpoint (restart.c:590-756) -> _lib.Restart."""
    from ._lib import Restart
    r = Restart()
    check(lib().sipnet_io_read_restart(str(path).encode(), C.byref(r)), "read_restart")
    return r


def write_restart(path, restart):
    """Write a checkpoint in the reference's text layout (restart.c:787-828)."""
    check(lib().sipnet_io_write_restart(str(path).encode(), C.byref(restart)), "write_restart")


def check_restart(restart, flags, clim):
    """Load-time checks of restartLoadCheckpoint (restart.c:968-996) against the segment
    about to run; returns the warning bits, raises SipnetError(9) on a mismatch."""
    fl = (C.c_int32 * 12)(*flags)
    warn = C.c_int32(0)
    has = 1 if clim is not None and clim.n_steps > 0 else 0
    y0 = int(clim.year[0]) if has else 0
    d0 = int(clim.day[0]) if has else 0
    t0 = float(clim.data[0, 10]) if has else 0.0
    l0 = float(clim.data[0, 0]) if has else 0.0
    check(lib().sipnet_restart_check(C.byref(restart), fl, has, y0, d0, t0, l0, C.byref(warn)),
          "check_restart")
    return warn.value

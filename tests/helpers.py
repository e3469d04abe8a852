"""Test-side access to the checkers under oracle/ and to the golden fixtures.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
oracle/ -- the product (sipnet_amd/) never does.
"""
import ctypes as C
import gzip
import os
import shutil
import subprocess
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "liboracle.so")
REF_SO = os.path.join(REPO, "oracle", "_ref", "libsipnet_ref.so")
NREC = 36


class OEvent(C.Structure):
    _fields_ = [("type", C.c_int), ("year", C.c_int), ("day", C.c_int), ("pad", C.c_int),
                ("p", C.c_double * 4)]


class Diag(C.Structure):
    _fields_ = [("n_clamp_warn", C.c_long), ("n_balance_warn", C.c_long),
                ("max_abs_dC", C.c_double), ("max_abs_dN", C.c_double),
                ("died_at_step", C.c_int)]


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        L = self.lib
        L.sipo_run_member.restype = C.c_int
        L.sipo_run_block.restype = C.c_int
        L.sipo_time_members.restype = C.c_double
        for n in ("sipo_clipped_water_frac", "sipo_resp_moist_effect", "sipo_temp_effect",
                  "sipo_cn_effect", "sipo_anaerobic_index", "sipo_methane_moist_effect",
                  "sipo_volatilization_moist_effect", "sipo_light_eff", "sipo_ring_probe"):
            getattr(L, n).restype = C.c_double

    @staticmethod
    def _events(events):
        n = len(events) if events else 0
        arr = (OEvent * max(n, 1))()
        for i in range(n):
            e = events[i]
            arr[i].type, arr[i].year, arr[i].day = e.type, e.year, e.day
            for k in range(4):
                arr[i].p[k] = e.p[k]
        return n, arr

    def run_member(self, flags, raw, clim, events=None, events_out=None, want_rec=True):
        """-> (status, rec[n][36], diag)"""
        fl = (C.c_int * 12)(*flags)
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        n = clim.n_steps
        rec = np.zeros((n, NREC)) if want_rec else None
        nev, evarr = self._events(events)
        diag = Diag()
        st = self.lib.sipo_run_member(
            fl, raw.ctypes.data_as(C.c_void_p), n, clim.data.ctypes.data_as(C.c_void_p),
            clim.year.ctypes.data_as(C.c_void_p), clim.day.ctypes.data_as(C.c_void_p), nev, evarr,
            rec.ctypes.data_as(C.c_void_p) if want_rec else None, None, None, None,
            events_out.encode() if events_out else None, C.byref(diag))
        return st, rec, diag

    def run_member_debug(self, flags, raw, clim, events=None):
        """-> (status, rec[n][36], dbg[n][72]): the record plus what --debug-log prints"""
        fl = (C.c_int * 12)(*flags)
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        n = clim.n_steps
        rec = np.zeros((n, NREC))
        dbg = np.zeros((n, 72))
        nev, evarr = self._events(events)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.sipo_run_member_debug.restype = C.c_int
        st = self.lib.sipo_run_member_debug(fl, vp(raw), n, vp(clim.data), vp(clim.year),
                                            vp(clim.day), nev, evarr, vp(rec), vp(dbg))
        return st, rec, dbg

    def run_block(self, flags, raw, clim, events=None, m0=0, m1=None, want_final=True):
        """raw[n_members][80] -> planes[3][n_steps][n_members], final[n_members][36], status"""
        fl = (C.c_int * 12)(*flags)
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        M = raw.shape[0]
        if m1 is None:
            m1 = M
        n = clim.n_steps
        planes = np.zeros((3, n, M))
        final = np.zeros((M, NREC))
        status = np.zeros(M, dtype=np.int32)
        nev, evarr = self._events(events)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.sipo_run_block(fl, vp(raw), m0, m1, M, n, vp(clim.data), vp(clim.year),
                                vp(clim.day), nev, evarr, vp(planes[0]), vp(planes[1]),
                                vp(planes[2]), vp(final), vp(status))
        return planes, final, status

    def time_members(self, flags, raw, clim):
        fl = (C.c_int * 12)(*flags)
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        sink = C.c_double()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        return self.lib.sipo_time_members(fl, vp(raw), raw.shape[0], clim.n_steps, vp(clim.data),
                                          vp(clim.year), vp(clim.day), C.byref(sink))


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
    return Oracle(ORACLE_SO)


# ---- golden smoke cases (inputs + expected outputs of the reference's own tests) ----
SMOKE_CASES = ["niwot", "russell_1", "russell_2", "russell_3", "russell_4"]


def smoke_dir(case):
    return os.path.join(GOLDEN, "smoke", case)


def gunzip_to(src_gz, dst):
    with gzip.open(src_gz, "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)


def load_smoke_case(case, tmpdir=None):
    """-> dict(cfg, flags, clim, params, events, golden_out(bytes), golden_events(bytes))"""
    import sipnet_amd as sa
    d = smoke_dir(case)
    cfg = sa.read_config(os.path.join(d, "sipnet.in"))
    flags = [cfg[n] for n in sa.FLAG_NAMES]
    tmp = tmpdir or tempfile.mkdtemp(prefix="sipnet_golden_")
    clim_name = "sipnet.clim.gz"
    clim_src = os.path.join(d, clim_name)
    if not os.path.exists(clim_src):  # russell cases share one forcing file
        clim_src = os.path.join(GOLDEN, "smoke", "russell_1", clim_name)
    clim_path = os.path.join(tmp, f"{case}.clim")
    gunzip_to(clim_src, clim_path)
    clim = sa.read_clim(clim_path, gdd=flags[1])
    params, _ = sa.read_params(os.path.join(d, "sipnet.param"), flags)
    events = sa.read_events(os.path.join(d, "events.in"), flags, params) if flags[0] else []
    with gzip.open(os.path.join(d, "sipnet.out.gz"), "rb") as fh:
        golden_out = fh.read()
    with open(os.path.join(d, "events.out"), "rb") as fh:
        golden_events = fh.read()
    return dict(cfg=cfg, flags=flags, clim=clim, params=params, events=events,
                golden_out=golden_out, golden_events=golden_events, clim_path=clim_path)


def out_text(clim, rec, header):
    """Format records as `.out` text through the product's formatter."""
    import sipnet_amd as sa
    parts = [sa.format_out_header()] if header else []
    for t in range(clim.n_steps):
        parts.append(sa.format_out_row(clim.year[t], clim.day[t], clim.data[t, 10], rec[t]))
    return "".join(parts).encode()

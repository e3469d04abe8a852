"""Node: the multi-GPU host object of the C-ABI (sipnet_node_*, csrc/node.cpp) from Python.

One process, one shard (batch + stream + RCCL rank + host thread) per listed device; members or whole
sites sharded; the collectives are issued by the C library.  bench.py's N-rank runs use one process per
GPU and torch.distributed instead -- this wrapper exists for tests and tools that drive the C host.
"""
import ctypes as C

import numpy as np

from ._lib import ALL_SITES, F64, NPARAMS, SHARD_MEMBERS, SHARD_SITES, Event, check, lib


class Node:
    def __init__(self, flags, n_sites, n_members, precision=F64, devices=(0,), shard=SHARD_MEMBERS, fast_math=None,
                 kernel=None, kernel_options=0):
        self.L = lib()
        self.n_sites, self.n_members, self.precision = int(n_sites), int(n_members), precision
        self.devices = list(devices)
        h = C.c_void_p()
        fl = (C.c_int32 * 12)(*flags)
        dv = (C.c_int32 * len(self.devices))(*self.devices)
        check(self.L.sipnet_node_create_sharded(fl, self.n_sites, self.n_members, precision, dv, len(self.devices),
                                                int(shard), C.byref(h)), "node_create_sharded")
        self.h = h
        self.n = self.L.sipnet_node_n_devices(h)
        self.shard = shard
        self.ld = int(self.L.sipnet_node_ld(h))
        self.n_run = 0
        if fast_math is not None and precision == F64:
            check(self.L.sipnet_node_set_math(h, 1 if fast_math else 0), "node_set_math")
        if kernel is not None:
            check(self.L.sipnet_node_set_kernel(h, int(kernel), int(kernel_options)), "node_set_kernel")

    def close(self):
        if getattr(self, "h", None):
            self.L.sipnet_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- layout -------------------------------------------------------------------
    def member_range(self, k):
        a, c = C.c_int32(), C.c_int32()
        check(self.L.sipnet_node_member_range(self.h, k, C.byref(a), C.byref(c)), "member_range")
        return a.value, c.value

    def site_range(self, k):
        a, c = C.c_int32(), C.c_int32()
        check(self.L.sipnet_node_site_range(self.h, k, C.byref(a), C.byref(c)), "site_range")
        return a.value, c.value

    def collective_library(self):
        return self.L.sipnet_node_collective_library(self.h).decode()

    def kernel_name(self, k=0):
        return self.L.sipnet_batch_last_kernel_name(self.L.sipnet_node_batch(self.h, k)).decode()

    # -- inputs -------------------------------------------------------------------
    def set_climate(self, site, clim):
        check(self.L.sipnet_node_set_climate(self.h, site, clim.n_steps, clim.data.ctypes.data, clim.year.ctypes.data,
                                             clim.day.ctypes.data), "node_set_climate")

    def set_events(self, site, events):
        n = len(events)
        arr = (Event * max(n, 1))(*events)
        check(self.L.sipnet_node_set_events(self.h, site, n, arr), "node_set_events")

    def set_params(self, site, raw, first_member=0):
        """site=None: the same members at every site (SIPNET_ALL_SITES)"""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        assert raw.ndim == 2 and raw.shape[1] == NPARAMS
        check(self.L.sipnet_node_set_params(self.h, ALL_SITES if site is None else site, first_member, raw.shape[0],
                                            raw.ctypes.data), "node_set_params")

    def setup(self):
        check(self.L.sipnet_node_setup(self.h), "node_setup")

    # -- run + collectives ----------------------------------------------------------
    def run(self, step0, n_steps):
        check(self.L.sipnet_node_run(self.h, step0, n_steps), "node_run")
        self.n_run = n_steps

    def forecast(self, step0, n_steps):
        check(self.L.sipnet_node_forecast(self.h, step0, n_steps), "node_forecast")
        self.n_run = n_steps

    def sync(self):
        check(self.L.sipnet_node_sync(self.h), "node_sync")

    def status(self):
        st = np.zeros((self.n_sites, self.n_members), dtype=np.int32)
        check(self.L.sipnet_node_get_status(self.h, st.ctypes.data), "node_get_status")
        return st

    def gather_stats(self):
        """-> the whole ensemble's statistics [3][n_run][n_sites][2] on the host"""
        tot = np.zeros((3, self.n_run, self.n_sites, 2))
        check(self.L.sipnet_node_gather_stats(self.h, tot.ctypes.data), "node_gather_stats")
        return tot

    def _to_host(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype=dtype)
        check(self.L.sipnet_dev_to_host(out.ctypes.data, ptr, out.nbytes, None), "dev_to_host")
        return out

    def _elem(self):
        return np.float64 if self.precision == F64 else np.float32

    def planes(self, k):
        """shard k's planes [3][n_run][ld] on the host (after sync).  After run_gathering the device block lies
        segment by segment ([3][len_j][ld] each, include/sipnet_amd.h): reassembled here."""
        nseg = self.L.sipnet_node_n_segments(self.h)
        if nseg == 0:
            return self._to_host(self.L.sipnet_node_planes(self.h, k), (3, self.n_run, self.ld), self._elem())
        flat = self._to_host(self.L.sipnet_node_planes(self.h, k), (3 * self.n_run * self.ld,), self._elem())
        out = np.empty((3, self.n_run, self.ld), dtype=flat.dtype)
        first0 = None
        for j in range(nseg):
            first, length = C.c_int32(0), C.c_int32(0)
            self.L.sipnet_node_gathered_segment(self.h, k, j, C.byref(first), C.byref(length))
            first0 = first.value if first0 is None else first0
            a, n = first.value - first0, length.value
            out[:, a:a + n] = flat[3 * a * self.ld:3 * (a + n) * self.ld].reshape(3, n, self.ld)
        return out

    def gather_planes(self):
        check(self.L.sipnet_node_gather_planes(self.h), "node_gather_planes")

    def gathered_planes(self, k):
        return self._to_host(self.L.sipnet_node_gathered_planes(self.h, k), (self.n, 3, self.n_run, self.ld), self._elem())

    def run_gathering(self, step0, n_steps, n_segments):
        """the run in n_segments launches, segment j's member-resolved planes all-gathered on the shards' second
        streams under the kernel of segment j + 1 (no statistics)"""
        check(self.L.sipnet_node_run_gathering(self.h, step0, n_steps, n_segments), "node_run_gathering")
        self.n_run = n_steps

    def gathered_segment(self, k, j):
        """-> (first record, segment j of every shard as device k holds it: [n][3][len][ld]) on the host (after sync)"""
        first, length = C.c_int32(0), C.c_int32(0)
        ptr = self.L.sipnet_node_gathered_segment(self.h, k, j, C.byref(first), C.byref(length))
        if not ptr:
            raise ValueError("no such segment")
        return first.value, self._to_host(ptr, (self.n, 3, length.value, self.ld), self._elem())

    def gathered_member_planes(self, k):
        """what device k holds after run_gathering, as [3][n_run][n_sites][n_members] on the host"""
        self.sync()
        out = np.zeros((3, self.n_run, self.n_sites, self.n_members), dtype=self._elem())
        t = 0
        for j in range(self.L.sipnet_node_n_segments(self.h)):
            _, g = self.gathered_segment(k, j)
            for q in range(self.n):
                m0, mc = self.member_range(q)
                s0, sc = self.site_range(q)
                out[:, t:t + g.shape[2], s0:s0 + sc, m0:m0 + mc] = g[q][:, :, :sc * mc].reshape(3, g.shape[2], sc, mc)
            t += g.shape[2]
        return out

    def run_gathering_reduced(self, step0, n_steps, n_segments, form, sum_steps=0):
        """sipnet_node_run_gathering_reduced: the member-resolved exchange in a form that fits under the kernel -- form
        "f32" (the planes as floats) or "sums" (every member's sums over groups of sum_steps steps, doubles)"""
        f = {"f32": 1, "sums": 2}[form]
        check(self.L.sipnet_node_run_gathering_reduced(self.h, step0, n_steps, n_segments, f, int(sum_steps)),
              "node_run_gathering_reduced")
        self.n_run = n_steps

    def gathered_reduced_member_rows(self, k):
        """what device k holds after run_gathering_reduced, as [3][rows][n_sites][n_members] on the host (rows = steps, or
        groups of steps)"""
        self.sync()
        parts = []
        for j in range(self.L.sipnet_node_n_segments(self.h)):
            first, rows, eb = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            ptr = self.L.sipnet_node_gathered_reduced(self.h, k, j, C.byref(first), C.byref(rows), C.byref(eb))
            if not ptr:
                raise ValueError("no reduced segment")
            g = self._to_host(ptr, (self.n, 3, rows.value, self.ld), np.float32 if eb.value == 4 else np.float64)
            out = np.zeros((3, rows.value, self.n_sites, self.n_members), dtype=g.dtype)
            for q in range(self.n):
                m0, mc = self.member_range(q)
                s0, sc = self.site_range(q)
                out[:, :, s0:s0 + sc, m0:m0 + mc] = g[q][:, :, :sc * mc].reshape(3, rows.value, sc, mc)
            parts.append(out)
        return np.concatenate(parts, axis=1)

    def gathered_stats(self, k):
        mx = self.n_sites if self.shard == SHARD_MEMBERS else max(self.site_range(j)[1] for j in range(self.n))
        return self._to_host(self.L.sipnet_node_gathered_stats(self.h, k), (self.n, 3, self.n_run, mx, 2), np.float64)

    def member_planes(self):
        """the planes of the last run as [3][n_run][n_sites][n_members], assembled from the shards on the host"""
        self.sync()
        out = np.zeros((3, self.n_run, self.n_sites, self.n_members), dtype=self._elem())
        for k in range(self.n):
            p = self.planes(k)
            m0, mc = self.member_range(k)
            s0, sc = self.site_range(k)
            out[:, :, s0:s0 + sc, m0:m0 + mc] = p[:, :, :sc * mc].reshape(3, self.n_run, sc, mc)
        return out

    # -- per-shard state (through the shard's batch) ------------------------------------
    def shard_state(self, k):
        from ._lib import NSTATE
        m0, mc = self.member_range(k)
        s0, sc = self.site_range(k)
        st = np.zeros((sc * mc, NSTATE))
        check(self.L.sipnet_batch_get_state(self.L.sipnet_node_batch(self.h, k), st.ctypes.data,
                                            self.L.sipnet_node_stream(self.h, k)), "get_state")
        return st

    def shard_rings(self, k):
        from ._lib import RING_SLOTS
        m0, mc = self.member_range(k)
        s0, sc = self.site_range(k)
        r = np.zeros((sc * mc, RING_SLOTS))
        check(self.L.sipnet_batch_get_rings(self.L.sipnet_node_batch(self.h, k), r.ctypes.data,
                                            self.L.sipnet_node_stream(self.h, k)), "get_rings")
        return r

    # -- particle filter ----------------------------------------------------------------
    def pf_connect(self, with_params=True):
        check(self.L.sipnet_node_pf_connect(self.h, int(with_params)), "node_pf_connect")

    def pf_arm(self, obs, sigma):
        """before forecast(): the shards' forecast launches leave their log-weight blocks themselves (sipnet_node_pf_arm)"""
        check(self.L.sipnet_node_pf_arm(self.h, float(obs), float(sigma)), "node_pf_arm")

    def pf_analysis(self, variable, obs, sigma, u0):
        check(self.L.sipnet_node_pf_analysis(self.h, int(variable), float(obs), float(sigma), float(u0)), "node_pf_analysis")

    def pf_check(self):
        n = C.c_int32()
        check(self.L.sipnet_node_pf_check(self.h, C.byref(n)), "node_pf_check")
        return n.value

    def pf_info(self, k):
        """sipnet_batch_pf_info of shard k's batch"""
        from ._lib import PfInfo
        d = PfInfo()
        check(self.L.sipnet_batch_pf_info(self.L.sipnet_node_batch(self.h, k), C.byref(d), self.L.sipnet_node_stream(self.h, k)),
              "pf_info")
        return {f: getattr(d, f) for f, _ in PfInfo._fields_}

    def pf_ancestors(self, k):
        return self._to_host(self.L.sipnet_node_pf_ancestors(self.h, k), (self.member_range(k)[1],), np.int32)

"""Batch: an ensemble x site batch resident in HBM (sipnet_batch of the C-ABI).

Mirrors the reference call sequence of frontend.c:212-250 --
initModel / initEvents / setupModel / runModelOutput / cleanupModel -- for many
members and sites at once.  Outputs live in torch CUDA tensors that the caller
(or this class) allocates; the C-ABI only ever sees their raw device pointers.
"""
import ctypes as C

import numpy as np

from ._lib import (F32_MIXED, F64, KERNEL_AUTO, NPARAMS, NREC, NSTATE, RING_SLOTS, Event, LaunchInfo,
                   check, lib)


class Batch:
    def __init__(self, flags, n_sites, n_members, precision=F64, device=0, fast_math=None,
                 kernel=KERNEL_AUTO, kernel_options=0):
        """fast_math (fp64 batches): True = throughput kernels, False / None = strict reference
        order (the library default; no environment variable changes it).  kernel /
        kernel_options: sipnet_batch_set_kernel (KERNEL_* / KOPT_* of _lib)."""
        import torch  # device memory + streams only
        self._torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("sipnet_amd.Batch needs a HIP device (no CPU path exists)")
        self.L = lib()
        self.flags = list(flags)
        self.n_sites, self.n_members = int(n_sites), int(n_members)
        self.ncol = self.n_sites * self.n_members
        self.precision = precision
        self.device = torch.device("cuda", device)
        self.out_dtype = torch.float64 if precision == F64 else torch.float32
        h = C.c_void_p()
        fl = (C.c_int32 * 12)(*self.flags)
        check(self.L.sipnet_batch_create(fl, self.n_sites, self.n_members, precision,
                                         device, C.byref(h)), "batch_create")
        self.h = h
        self.n_steps = 0
        if fast_math is not None and precision == F64:
            self.set_math(fast_math)
        if kernel != KERNEL_AUTO or kernel_options:
            self.set_kernel(kernel, kernel_options)

    def set_math(self, fast):
        """sipnet_batch_set_math: arithmetic policy of an fp64 batch (strict order / throughput)"""
        check(self.L.sipnet_batch_set_math(self.h, 1 if fast else 0), "set_math")

    def set_kernel(self, kernel=KERNEL_AUTO, options=0):
        """sipnet_batch_set_kernel: force one step kernel (KERNEL_ONE_WAVE / COOP_LDS / COOP_HBM /
        STRICT) or go back to the shape-based choice (KERNEL_AUTO)."""
        check(self.L.sipnet_batch_set_kernel(self.h, int(kernel), int(options)), "set_kernel")

    def enable_diagnostics(self, on=True):
        """count the reference's per-step clamp / mass-balance warnings per member"""
        check(self.L.sipnet_batch_enable_diagnostics(self.h, int(on)), "enable_diagnostics")

    def get_diagnostics(self):
        """-> dict(n_clamp_warn[ncol], n_balance_warn[ncol], max_abs_dC[ncol], max_abs_dN[ncol])"""
        c = np.zeros(self.ncol, dtype=np.int64)
        w = np.zeros(self.ncol, dtype=np.int64)
        dc, dn = np.zeros(self.ncol), np.zeros(self.ncol)
        check(self.L.sipnet_batch_get_diagnostics(self.h, c.ctypes.data, w.ctypes.data, dc.ctypes.data,
                                                  dn.ctypes.data, self._stream()), "get_diagnostics")
        return dict(n_clamp_warn=c, n_balance_warn=w, max_abs_dC=dc, max_abs_dN=dn)

    def last_launch(self):
        """What the last run() launched: dict(kernel, grid, block_threads, waves_per_simd,
        lds_bytes, num_cus, plan_threads, plan_build_ms, plan_upload_ms)."""
        li = LaunchInfo()
        check(self.L.sipnet_batch_last_launch(self.h, C.byref(li)), "last_launch")
        return {n: (getattr(li, n).decode() if n == "kernel" else getattr(li, n))
                for n, _ in LaunchInfo._fields_}

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.sipnet_batch_destroy(self.h)
            self.h = None
            # the batch-less sipnet_pf_* entry points (dist.pf_systematic_ancestors, pf_exchange_plan) keep device
            # scratch per host thread; nothing else frees it
            self.L.sipnet_pf_release_scratch()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    # -- inputs -----------------------------------------------------------------
    def set_climate(self, site, clim):
        check(self.L.sipnet_batch_set_climate(self.h, site, clim.n_steps, clim.data.ctypes.data,
                                              clim.year.ctypes.data, clim.day.ctypes.data),
              "set_climate")
        self.n_steps = int(self.L.sipnet_batch_nsteps(self.h))     # the longest site's (sites may differ in length)

    def set_climates(self, clims, first_site=0):
        """sipnet_batch_set_climate_sites: the forcings of sites first_site.. in one call (copied on the plan threads)"""
        n = len(clims)
        ns = (C.c_int32 * n)(*[c.n_steps for c in clims])
        cp = (C.c_void_p * n)(*[c.data.ctypes.data for c in clims])
        yp = (C.c_void_p * n)(*[c.year.ctypes.data for c in clims])
        dp = (C.c_void_p * n)(*[c.day.ctypes.data for c in clims])
        check(self.L.sipnet_batch_set_climate_sites(self.h, first_site, n, ns, cp, yp, dp), "set_climate_sites")
        self.n_steps = int(self.L.sipnet_batch_nsteps(self.h))

    def site_n_steps(self, site):
        """the number of records of this site's forcing (self.n_steps is the longest site's)"""
        return int(self.L.sipnet_batch_site_nsteps(self.h, site))

    def set_events(self, site, events):
        n = len(events)
        arr = (Event * max(n, 1))(*events)
        check(self.L.sipnet_batch_set_events(self.h, site, n, arr), "set_events")

    def set_params(self, site, raw, first_member=0):
        """site: a site index, or None / ALL_SITES = the same members at every site (one upload)"""
        if site is None:
            site = -1   # SIPNET_ALL_SITES
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        if raw.ndim == 1:
            raw = np.broadcast_to(raw, (self.n_members, NPARAMS)).copy()
        assert raw.shape[1] == NPARAMS
        check(self.L.sipnet_batch_set_params(self.h, site, first_member, raw.shape[0],
                                             raw.ctypes.data), "set_params")

    def setup(self):
        """== setupModel() for every member (sipnet.c:1858-1951)."""
        check(self.L.sipnet_batch_setup(self.h, self._stream()), "setup")

    # -- run --------------------------------------------------------------------
    def alloc_outputs(self, n_steps, full=False):
        t = self._torch
        planes = t.empty((3, n_steps, self.ncol), dtype=self.out_dtype, device=self.device)
        rec = (t.empty((n_steps, NREC, self.ncol), dtype=t.float64, device=self.device)
               if full else None)
        return planes, rec

    def run(self, step0=0, n_steps=None, planes=None, rec=None, want_planes=True, full=False):
        """== the time loop of runModelOutput() (sipnet.c:1969-1982).

        Returns (planes, rec): planes[3][n_steps][ncol] = NEE, GPP, ET per step;
        rec[n_steps][36][ncol] full records when `full`."""
        if n_steps is None:
            n_steps = self.n_steps - step0
        if planes is None and want_planes:
            planes, _ = self.alloc_outputs(n_steps, False)
        if rec is None and full:
            rec = self._torch.empty((n_steps, NREC, self.ncol), dtype=self._torch.float64,
                                    device=self.device)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        nee = planes[0] if planes is not None else None
        gpp = planes[1] if planes is not None else None
        et = planes[2] if planes is not None else None
        check(self.L.sipnet_batch_run(self.h, step0, n_steps, ptr(nee), ptr(gpp), ptr(et),
                                      ptr(rec), self.ncol, self._stream()), "run")
        return planes, rec

    def run_debug(self, step0=0, n_steps=None):
        """The same advance with the reference's `--debug-log` content (debug_log.c:285-312).

        Returns (rec[n_steps][NREC][ncol], dbg[n_steps][NDBG][ncol]) device tensors."""
        from ._lib import NDBG
        t = self._torch
        if n_steps is None:
            n_steps = self.n_steps - step0
        rec = t.empty((n_steps, NREC, self.ncol), dtype=t.float64, device=self.device)
        dbg = t.empty((n_steps, NDBG, self.ncol), dtype=t.float64, device=self.device)
        check(self.L.sipnet_batch_run_debug(self.h, step0, n_steps, C.c_void_p(rec.data_ptr()),
                                            C.c_void_p(dbg.data_ptr()), self.ncol,
                                            self._stream()), "run_debug")
        return rec, dbg

    def run_stats(self, step0=0, n_steps=None, planes=None, stats=None):
        """run() + the ensemble statistics of the three planes from the same launch
        (sipnet_batch_run_stats): stats[3][n_steps][n_sites][2] = sum, sum of squares over each
        site's members.  Returns (planes, stats)."""
        t = self._torch
        if n_steps is None:
            n_steps = self.n_steps - step0
        if planes is None:
            planes, _ = self.alloc_outputs(n_steps, False)
        if stats is None:
            stats = t.empty((3, n_steps, self.n_sites, 2), dtype=t.float64, device=self.device)
        assert stats.is_contiguous() and stats.dtype == t.float64
        check(self.L.sipnet_batch_run_stats(self.h, step0, n_steps, C.c_void_p(planes[0].data_ptr()),
                                            C.c_void_p(planes[1].data_ptr()), C.c_void_p(planes[2].data_ptr()),
                                            self.ncol, C.c_void_p(stats.data_ptr()), self._stream()), "run_stats")
        return planes, stats

    def reduce_plane(self, plane, stats=None):
        """Per (step, site) ensemble sum and sum of squares of one output plane
        [n_steps][ncol] -> stats[n_steps][n_sites][2] (float64, on device)."""
        t = self._torch
        n_steps = plane.shape[0]
        if stats is None:
            stats = t.empty((n_steps, self.n_sites, 2), dtype=t.float64, device=self.device)
        check(self.L.sipnet_batch_reduce_plane(self.h, C.c_void_p(plane.data_ptr()),
                                               int(plane.dtype == t.float32), n_steps, self.ncol,
                                               C.c_void_p(stats.data_ptr()), self._stream()),
              "reduce_plane")
        return stats

    def time_next_launch(self):
        """bracket the next run()'s step kernel with the timing events whatever its length (launches of 512 steps and
        more always are): last_kernel_ms() after it"""
        check(self.L.sipnet_batch_time_next_launch(self.h), "time_next_launch")

    def last_kernel_ms(self):
        return self.L.sipnet_batch_last_kernel_ms(self.h)

    # -- state ------------------------------------------------------------------
    def get_state(self):
        st = np.zeros((self.ncol, NSTATE))
        check(self.L.sipnet_batch_get_state(self.h, st.ctypes.data, self._stream()), "get_state")
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.float64)
        assert st.shape == (self.ncol, NSTATE)
        check(self.L.sipnet_batch_set_state(self.h, st.ctypes.data, self._stream()), "set_state")

    def get_status(self):
        s = np.zeros(self.ncol, dtype=np.int32)
        check(self.L.sipnet_batch_get_status(self.h, s.ctypes.data, self._stream()), "get_status")
        return s

    def get_ring(self, col):
        v = np.zeros(RING_SLOTS)
        check(self.L.sipnet_batch_get_ring(self.h, col, v.ctypes.data, self._stream()), "get_ring")
        return v

    def get_rings(self):
        r = np.zeros((self.ncol, RING_SLOTS))
        check(self.L.sipnet_batch_get_rings(self.h, r.ctypes.data, self._stream()), "get_rings")
        return r

    def set_rings(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        assert r.shape == (self.ncol, RING_SLOTS)
        check(self.L.sipnet_batch_set_rings(self.h, r.ctypes.data, self._stream()), "set_rings")

    def checkpoint(self):
        """Complete per-member carried state: (state[ncol][32], rings[ncol][250])."""
        return self.get_state(), self.get_rings()

    def restore(self, ckpt):
        self.set_state(ckpt[0])
        self.set_rings(ckpt[1])

    # -- restart checkpoints (sipnet.c:1963-1989, restart.c) ---------------------
    def set_resume(self, site, restart):
        """Site-uniform part of a checkpoint (gdd, lastYear, d_till_mod, ring layout); call
        before setup().  restart=None clears it."""
        ptr = C.byref(restart) if restart is not None else None
        check(self.L.sipnet_batch_set_resume(self.h, site, ptr), "set_resume")

    def import_restart(self, site, restarts, first_member=0):
        """After setup(): overwrite the carried state of len(restarts) members of `site`."""
        from ._lib import Restart
        arr = (Restart * len(restarts))(*restarts)
        check(self.L.sipnet_batch_import_restart(self.h, site, first_member, len(restarts), arr,
                                                 self._stream()), "import_restart")

    def run_sums(self, step0, n_steps, sum_steps, out=None):
        """sipnet_batch_run_sums: every member's sums over groups of sum_steps steps, summed inside the step kernel's
        launch -> f64 device tensor [3][groups][ncol] (NEE, GPP, ET)"""
        t = self._torch
        groups = (n_steps + sum_steps - 1) // sum_steps
        if out is None:
            out = t.empty((3, groups, self.ncol), dtype=t.float64, device=self.device)
        assert out.dtype == t.float64 and out.is_contiguous() and out.shape == (3, groups, self.ncol)
        check(self.L.sipnet_batch_run_sums(self.h, int(step0), int(n_steps), int(sum_steps), C.c_void_p(out[0].data_ptr()),
                                           C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), self.ncol,
                                           self._stream()), "run_sums")
        return out

    def sums_in_kernel(self):
        return bool(self.L.sipnet_batch_sums_in_kernel(self.h))

    def export_restart(self, site, member, n_steps_done, last_rec=None, prev_pools=None):
        """Checkpoint of one member after n_steps_done records (restartWriteCheckpoint)."""
        from ._lib import Restart
        r = Restart()
        lr = pp = None
        if last_rec is not None:
            lr = np.ascontiguousarray(last_rec, dtype=np.float64)
            assert lr.shape == (NREC,)
        if prev_pools is not None:
            pp = np.ascontiguousarray(prev_pools, dtype=np.float64)
            assert pp.shape[0] >= 13
        check(self.L.sipnet_batch_export_restart(
            self.h, site, member, int(n_steps_done), lr.ctypes.data if lr is not None else None,
            pp.ctypes.data if pp is not None else None, C.byref(r), self._stream()), "export_restart")
        return r

    # -- particle-filter analysis step (pf.hip; BASELINE config C5) ---------------
    def pf_log_weights(self, plane, obs, sigma, out=None):
        """Gaussian log-likelihood of an observed flux sum: plane[T][ncol] (NEE, GPP or ET
        plane of the forecast) -> logw[ncol] f64 on the device."""
        t = self._torch
        if out is None:
            out = t.empty(self.ncol, dtype=t.float64, device=self.device)
        check(self.L.sipnet_batch_pf_log_weights(
            self.h, C.c_void_p(plane.data_ptr()), int(plane.dtype == t.float32), plane.shape[0],
            plane.shape[1], float(obs), float(sigma), C.c_void_p(out.data_ptr()), self._stream()),
            "pf_log_weights")
        return out

    def pf_arm(self, obs, sigma):
        """sipnet_batch_pf_arm: the next run() (one-wave kernel, planes only) also leaves the log-weights that
        pf_analysis_local(planes[0], obs, sigma, ...) would otherwise compute in a pass of its own"""
        t = self._torch
        if getattr(self, "_pf_buf", None) is None:
            self._pf_buf = (t.empty(self.ncol, dtype=t.float64, device=self.device),
                            t.empty(self.ncol, dtype=t.int32, device=self.device))
        check(self.L.sipnet_batch_pf_arm(self.h, float(obs), float(sigma), C.c_void_p(self._pf_buf[0].data_ptr())), "pf_arm")

    def pf_arm_block(self, obs, sigma, block):
        """the same for a connected filter: the next run() leaves the log-weights in `block` (this rank's slice of the
        all-gather's buffer, pf_local_weights' target), which then only adds the block maxima"""
        check(self.L.sipnet_batch_pf_arm(self.h, float(obs), float(sigma), C.c_void_p(block.data_ptr())), "pf_arm")

    def pf_analysis_local(self, plane, obs, sigma, u0, with_params=False, total_out=None):
        """log-weights -> systematic resampling -> resample, all particles in this batch, ONE library call
        (sipnet_batch_pf_analysis).  Returns (ancestors int32 [ncol], logw f64 [ncol]) on the device."""
        t = self._torch
        if getattr(self, "_pf_buf", None) is None:
            self._pf_buf = (t.empty(self.ncol, dtype=t.float64, device=self.device),
                            t.empty(self.ncol, dtype=t.int32, device=self.device))
        logw, anc = self._pf_buf
        check(self.L.sipnet_batch_pf_analysis(
            self.h, C.c_void_p(plane.data_ptr()), int(plane.dtype == t.float32), plane.shape[0], plane.shape[1],
            float(obs), float(sigma), float(u0), int(with_params), C.c_void_p(logw.data_ptr()),
            C.c_void_p(anc.data_ptr()), C.c_void_p(total_out.data_ptr()) if total_out is not None else None,
            self._stream()), "pf_analysis")
        return anc, logw

    # -- the filter across ranks by peer reads (sipnet_batch_pf_publish / _connect / _resample_peers) ----------
    def pf_publish(self, with_params=True):
        """-> bytes of this batch's sipnet_pf_peer descriptor (exchange them, then pf_connect)"""
        from ._lib import PfPeer
        d = PfPeer()
        check(self.L.sipnet_batch_pf_publish(self.h, int(with_params), C.byref(d)), "pf_publish")
        return bytes(d)

    def pf_connect(self, descriptors, rank):
        """descriptors: every rank's pf_publish() bytes in rank order"""
        from ._lib import PfPeer
        arr = (PfPeer * len(descriptors))(*[PfPeer.from_buffer_copy(x) for x in descriptors])
        check(self.L.sipnet_batch_pf_connect(self.h, len(descriptors), int(rank), arr), "pf_connect")
        self._pf_gathered = None

    def pf_block_len(self):
        return int(self.L.sipnet_batch_pf_block_len(self.h))

    def pf_info(self):
        """sipnet_batch_pf_info: what the last analysis did (one launch or several, its grid, the resident-workgroup
        budget) and how many of this rank's particles have crossed ranks since pf_connect (synchronises the stream)"""
        from ._lib import PfInfo
        d = PfInfo()
        check(self.L.sipnet_batch_pf_info(self.h, C.byref(d), self._stream()), "pf_info")
        return {k: getattr(d, k) for k, _ in PfInfo._fields_}

    def debug_set_num_cus(self, n):
        """test hook: pretend the device has n compute units (32 = one partition of a CPX-mode MI355X)"""
        check(self.L.sipnet_debug_set_num_cus(self.h, int(n)), "debug_set_num_cus")

    def debug_pf_barrier(self, spin_budget=0, absent_workgroup=-1):
        """test hook: the poll budget of the one-launch analysis' barriers, and a workgroup of the NEXT such launch that
        leaves without arriving (the "grid not co-resident" path)"""
        check(self.L.sipnet_debug_pf_barrier(self.h, int(spin_budget), int(absent_workgroup)), "debug_pf_barrier")

    def set_device_share(self, n_filters):
        """sipnet_batch_set_device_share: how many filters analyse on this device at the same time"""
        check(self.L.sipnet_batch_set_device_share(self.h, int(n_filters)), "set_device_share")

    def pf_local_weights(self, plane, obs, sigma, block):
        """this rank's block [nmax log-weights | block maxima] of the all-gather, into `block` (f64 device
        tensor of pf_block_len() entries, e.g. this rank's slice of the gathered buffer)"""
        t = self._torch
        assert block.dtype == t.float64 and block.is_contiguous() and block.numel() == self.pf_block_len()
        check(self.L.sipnet_batch_pf_local_weights(
            self.h, C.c_void_p(plane.data_ptr()), int(plane.dtype == t.float32), plane.shape[0], plane.shape[1],
            float(obs), float(sigma), C.c_void_p(block.data_ptr()), self._stream()), "pf_local_weights")
        return block

    def pf_resample_peers(self, gathered, u0, ancestors=None, total_out=None):
        """gathered[world][pf_block_len()]: every rank's block.  Resamples this rank's particles from wherever
        their ancestors live (peer reads); returns the ancestors' slots (int32 [ncol])."""
        t = self._torch
        assert gathered.dtype == t.float64 and gathered.is_contiguous()
        if ancestors is None:
            ancestors = t.empty(self.ncol, dtype=t.int32, device=self.device)
        check(self.L.sipnet_batch_pf_resample_peers(
            self.h, C.c_void_p(gathered.data_ptr()), float(u0), C.c_void_p(ancestors.data_ptr()),
            C.c_void_p(total_out.data_ptr()) if total_out is not None else None, self._stream()), "pf_resample_peers")
        return ancestors

    def pack_members(self, cols, with_params=False):
        """cols: int32 device tensor of local column indices -> packed block [words][n] of 8-byte words
        (state rows, ring rows -- floats for an fp32-mixed batch --, parameter rows)"""
        t = self._torch
        words = self.L.sipnet_batch_member_words(self.h, int(with_params))
        cols = cols.to(device=self.device, dtype=t.int32).contiguous()
        buf = t.empty((words, cols.numel()), dtype=t.float64, device=self.device)
        if cols.numel():
            check(self.L.sipnet_batch_pack_members(self.h, C.c_void_p(cols.data_ptr()), cols.numel(),
                                                   int(with_params), C.c_void_p(buf.data_ptr()),
                                                   self._stream()), "pack_members")
        return buf

    def resample(self, src, recv=None, block_cols=(), with_params=False):
        """Column j becomes its ancestor src[j]: < ncol = own old column, ncol + k = received
        column k of `recv` (concatenated packed blocks with block_cols[s] columns each)."""
        t = self._torch
        src = src.to(device=self.device, dtype=t.int32).contiguous()
        assert src.numel() == self.ncol
        nb = len(block_cols)
        bc = (C.c_int64 * max(nb, 1))(*[int(x) for x in block_cols])
        rp = C.c_void_p(recv.data_ptr()) if recv is not None and recv.numel() else None
        check(self.L.sipnet_batch_resample(self.h, C.c_void_p(src.data_ptr()), rp, nb, bc,
                                           int(with_params), self._stream()), "resample")

    def site_series(self, site):
        g = np.zeros(self.n_steps)
        d = np.zeros(self.n_steps)
        check(self.L.sipnet_batch_get_site_series(self.h, site, g.ctypes.data, d.ctypes.data),
              "site_series")
        return g, d

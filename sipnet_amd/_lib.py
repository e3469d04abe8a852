"""ctypes binding of libsipnet_amd.so (include/sipnet_amd.h).

The library is the product; there is no Python or CPU fallback for the compute
path.  Import fails loudly when the shared object has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C sipnet_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsipnet_amd.so")

NPARAMS, NFLAGS, NCLIM, NREC, NSTATE, RING_SLOTS = 80, 12, 11, 44, 32, 250
NDBG = 72  # --debug-log plane: 56 fluxes + tracker fields (include/sipnet_amd.h)
NREC_OUT = 36  # the output columns proper; 36..43 are the event log

OK = 0
ERR_BAD_PARAMETER = 3
ERR_UNKNOWN_EVENT = 4
ERR_INPUT_FILE = 5
ERR_FILE_OPEN = 6
ERR_INTERNAL = 7
ERR_BAD_CLI = 8
ERR_RESTART = 9
ERR_NO_DEVICE = 100
ERR_BAD_ARGUMENT = 101

F64, F32_MIXED = 0, 1
# enum sipnet_kernel / sipnet_kernel_option
KERNEL_AUTO, KERNEL_ONE_WAVE, KERNEL_COOP_LDS, KERNEL_COOP_HBM, KERNEL_STRICT, KERNEL_COOP_PAIR, KERNEL_COOP_QUAD = range(7)
KERNEL_COOP_NCYCLE = 7
KERNEL_COOP_NCYCLE_PAIR = 8
KOPT_ONE_WAVE_PER_SIMD, KOPT_RUNTIME_FLAGS, KOPT_FULL_STATE, KOPT_NO_REGULAR_TILES = 1, 2, 4, 8
KOPT_STATS_IN_KERNEL = 16
KOPT_BOUNDED_WAITS = 32
KOPT_WAIT_SELFTEST = 64
KOPT_DEVICE_PLAN = 256   # device plan for every site that can have one, whatever its step lengths (tests)
KOPT_HOST_PLAN = 128     # every site plan on host threads (default: eligible sites on the device, csrc/plan_device.h)
KOPT_PF_MULTI_LAUNCH = 512   # particle filter: the analysis as separate launches, never the one-launch kernel
KOPT_PF_MOVE_PARAMS = 1024   # particle filter across ranks: parameter rows travel with the particles (no replicated bank)
SHARD_MEMBERS, SHARD_SITES = 0, 1
ALL_SITES = -1


class Event(C.Structure):
    """struct sipnet_event"""
    _fields_ = [("type", C.c_int32), ("year", C.c_int32), ("day", C.c_int32),
                ("pad", C.c_int32), ("p", C.c_double * 4)]


RT_NAMES = ("gpp", "rtot", "ra", "rh", "rRoot", "rSoil", "rAboveground", "npp", "nee",
            "woodCreation", "gdd", "evapotranspiration", "soilWetnessFrac", "yearlyGpp",
            "yearlyRtot", "yearlyRa", "yearlyRh", "yearlyNpp", "yearlyNee", "yearlyLitter",
            "totGpp", "totRtot", "totRa", "totRh", "totNpp", "totNee", "methane", "n2o",
            "nLeaching", "nFixation", "nUptake", "meanNPP")   # enum sipnet_restart_tracker


class Restart(C.Structure):
    """struct sipnet_restart: one member's `SIPNET_RESTART` checkpoint"""
    _fields_ = [("model_version", C.c_char * 32), ("build_info", C.c_char * 96),
                ("checkpoint_utc_epoch", C.c_int64), ("processed_steps", C.c_int64),
                ("flags", C.c_int32 * 12),
                ("boundary_year", C.c_int32), ("boundary_day", C.c_int32),
                ("boundary_time", C.c_double), ("boundary_length", C.c_double),
                ("envi", C.c_double * 13), ("trackers", C.c_double * 32),
                ("trackers_last_year", C.c_int32),
                ("did_leaf_growth", C.c_int32), ("did_leaf_fall", C.c_int32),
                ("phenology_last_year", C.c_int32), ("is_alive", C.c_int32),
                ("mean_length", C.c_int32),
                ("d_till_mod", C.c_double), ("harvest_frac_removed", C.c_double),
                ("harvest_frac_transferred", C.c_double), ("mean_tot_weight", C.c_double),
                ("mean_start", C.c_int32), ("mean_last", C.c_int32), ("mean_sum", C.c_double),
                ("mean_values", C.c_double * 250), ("mean_weights", C.c_double * 250)]

    def tracker(self, name):
        return self.trackers[RT_NAMES.index(name)]

    def live_slots(self):
        """ring slots from the oldest to the newest entry"""
        out, i = [], self.mean_start
        while True:
            out.append(i)
            if i == self.mean_last:
                return out
            i = (i + 1) % 250


class LaunchInfo(C.Structure):
    """struct sipnet_launch_info"""
    _fields_ = [("kernel", C.c_char * 96), ("grid", C.c_int32), ("block_threads", C.c_int32),
                ("waves_per_simd", C.c_int32), ("lds_bytes", C.c_int32), ("num_cus", C.c_int32),
                ("plan_threads", C.c_int32), ("plan_build_ms", C.c_double),
                ("plan_upload_ms", C.c_double), ("plan_device_sites", C.c_int32), ("reserved", C.c_int32)]


class PfPeer(C.Structure):
    """struct sipnet_pf_peer: where a rank keeps its particles' checkpoint matrices (plain bytes: ranks
    exchange it with all_gather_object / any byte channel)"""
    _fields_ = [("process_id", C.c_int64), ("device", C.c_int32), ("n_particles", C.c_int32),
                ("precision", C.c_int32), ("with_params", C.c_int32), ("ipc_valid", C.c_int32),
                ("generic_exponents", C.c_int32), ("params_by_index", C.c_int32), ("reserved", C.c_int32),
                ("address", C.c_uint64 * 8), ("ipc", (C.c_ubyte * 64) * 8)]


class PfInfo(C.Structure):
    """struct sipnet_pf_info: what the last particle-filter analysis did, what the exchange has moved"""
    _fields_ = [("fused", C.c_int32), ("grid", C.c_int32), ("budget", C.c_int32), ("world", C.c_int32),
                ("n_slots", C.c_int64), ("cycles", C.c_int64), ("crossing", C.c_int64),
                ("params_by_index", C.c_int32), ("device_share", C.c_int32)]


PF_VOID_TOTAL = -(1 << 63)


RESTART_WARN_BOUNDARY_NOT_MIDNIGHT, RESTART_WARN_BUILD_INFO, RESTART_WARN_TIME_GAP = 1, 2, 4

# name -> (restype, argtypes); every symbol declared in include/sipnet_amd.h
_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_DP = C.POINTER(C.c_double)
SIGNATURES = {
    "sipnet_version": (C.c_char_p, []),
    "sipnet_last_error": (C.c_char_p, []),
    "sipnet_device_count": (C.c_int, []),
    "sipnet_batch_create": (C.c_int, [_I32P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "sipnet_batch_destroy": (None, [_P]),
    "sipnet_batch_set_climate": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sipnet_batch_set_events": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "sipnet_batch_set_params": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sipnet_batch_setup": (C.c_int, [_P, _P]),
    "sipnet_batch_set_math": (C.c_int, [_P, C.c_int32]),
    "sipnet_batch_set_kernel": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sipnet_kernel_choice": (C.c_int32, [_I32P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sipnet_batch_last_launch": (C.c_int, [_P, _P]),
    "sipnet_batch_last_kernel_name": (C.c_char_p, [_P]),
    "sipnet_batch_enable_diagnostics": (C.c_int, [_P, C.c_int32]),
    "sipnet_batch_get_diagnostics": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "sipnet_batch_run": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64, _P]),
    "sipnet_batch_run_debug": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    "sipnet_batch_reduce_plane": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P]),
    "sipnet_batch_run_stats": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P, _P]),
    "sipnet_batch_get_state": (C.c_int, [_P, _P, _P]),
    "sipnet_batch_set_state": (C.c_int, [_P, _P, _P]),
    "sipnet_batch_get_ring": (C.c_int, [_P, C.c_int64, _P, _P]),
    "sipnet_batch_get_rings": (C.c_int, [_P, _P, _P]),
    "sipnet_batch_set_rings": (C.c_int, [_P, _P, _P]),
    "sipnet_batch_get_status": (C.c_int, [_P, _P, _P]),
    "sipnet_batch_set_resume": (C.c_int, [_P, C.c_int32, _P]),
    "sipnet_batch_import_restart": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "sipnet_batch_export_restart": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "sipnet_batch_pf_log_weights": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_double,
                                              C.c_double, _P, _P]),
    "sipnet_pf_systematic_ancestors": (C.c_int, [_P, C.c_int64, C.c_double, _P, _P, _P]),
    "sipnet_pf_systematic_ancestors_async": (C.c_int, [_P, C.c_int64, C.c_double, _P, _P, _P, _P]),
    "sipnet_pf_exchange_plan": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "sipnet_pf_release_scratch": (None, []),
    "sipnet_pf_member_words": (C.c_int32, [C.c_int32]),
    "sipnet_batch_member_words": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sipnet_batch_pf_analysis": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double,
                                           C.c_int32, _P, _P, _P, _P]),
    "sipnet_batch_pf_publish": (C.c_int, [_P, C.c_int32, _P]),
    "sipnet_batch_pf_connect": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "sipnet_batch_pf_block_len": (C.c_int64, [_P]),
    "sipnet_batch_pf_local_weights": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "sipnet_batch_pf_resample_peers": (C.c_int, [_P, _P, C.c_double, _P, _P, _P]),
    "sipnet_batch_pack_members": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    "sipnet_batch_resample": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, _P]),
    "sipnet_node_create": (C.c_int, [_I32P, C.c_int32, C.c_int32, C.c_int32, _I32P, C.c_int32, C.POINTER(_P)]),
    "sipnet_node_create_sharded": (C.c_int, [_I32P, C.c_int32, C.c_int32, C.c_int32, _I32P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "sipnet_node_shard_mode": (C.c_int32, [_P]),
    "sipnet_node_site_range": (C.c_int, [_P, C.c_int32, _I32P, _I32P]),
    "sipnet_node_stream": (_P, [_P, C.c_int32]),
    "sipnet_node_forecast": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sipnet_node_pf_arm": (C.c_int, [_P, C.c_double, C.c_double]),
    "sipnet_node_get_status": (C.c_int, [_P, _P]),
    "sipnet_node_pf_connect": (C.c_int, [_P, C.c_int32]),
    "sipnet_node_pf_analysis": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "sipnet_node_pf_check": (C.c_int, [_P, _I32P]),
    "sipnet_node_pf_ancestors": (_P, [_P, C.c_int32]),
    "sipnet_node_pf_block_len": (C.c_int64, [_P]),
    "sipnet_node_destroy": (None, [_P]),
    "sipnet_node_n_devices": (C.c_int32, [_P]),
    "sipnet_node_batch": (_P, [_P, C.c_int32]),
    "sipnet_node_run_gathering_reduced": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sipnet_node_gathered_reduced": (_P, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sipnet_node_reduced_in_kernel": (C.c_int32, [_P]),
    "sipnet_node_member_range": (C.c_int, [_P, C.c_int32, _I32P, _I32P]),
    "sipnet_node_collective_library": (C.c_char_p, [_P]),
    "sipnet_comm_unique_id": (C.c_int, [_P]),
    "sipnet_comm_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "sipnet_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "sipnet_comm_world": (C.c_int32, [_P]),
    "sipnet_comm_destroy": (None, [_P]),
    "sipnet_node_set_climate": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sipnet_node_set_events": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "sipnet_node_set_params": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sipnet_node_set_math": (C.c_int, [_P, C.c_int32]),
    "sipnet_node_set_kernel": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sipnet_node_setup": (C.c_int, [_P]),
    "sipnet_node_run": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sipnet_node_sync": (C.c_int, [_P]),
    "sipnet_node_ld": (C.c_int64, [_P]),
    "sipnet_node_planes": (_P, [_P, C.c_int32]),
    "sipnet_node_stats": (_P, [_P, C.c_int32]),
    "sipnet_node_gather_stats": (C.c_int, [_P, _P]),
    "sipnet_node_gathered_stats": (_P, [_P, C.c_int32]),
    "sipnet_node_gather_planes": (C.c_int, [_P]),
    "sipnet_node_gathered_planes": (_P, [_P, C.c_int32]),
    "sipnet_node_run_gathering": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "sipnet_node_n_segments": (C.c_int32, [_P]),
    "sipnet_node_gathered_segment": (_P, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sipnet_batch_ncol": (C.c_int64, [_P]),
    "sipnet_batch_nsteps": (C.c_int32, [_P]),
    "sipnet_batch_site_nsteps": (C.c_int32, [_P, C.c_int32]),
    "sipnet_batch_get_site_series": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sipnet_batch_last_kernel_ms": (C.c_double, [_P]),
    "sipnet_batch_time_next_launch": (C.c_int, [_P]),
    "sipnet_batch_pf_arm": (C.c_int, [_P, C.c_double, C.c_double, _P]),
    "sipnet_dev_alloc": (_P, [C.c_size_t]),
    "sipnet_dev_free": (None, [_P]),
    "sipnet_dev_to_host": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "sipnet_dev_to_host_2d": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t, _P]),
    "sipnet_dev_to_dev_2d": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t, _P]),
    "sipnet_batch_set_climate_sites": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "sipnet_debug_plan_compare": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "sipnet_batch_run_sums": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    "sipnet_batch_sums_in_kernel": (C.c_int32, [_P]),
    "sipnet_debug_set_num_cus": (C.c_int, [_P, C.c_int32]),
    "sipnet_debug_pf_barrier": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sipnet_batch_set_device_share": (C.c_int, [_P, C.c_int32]),
    "sipnet_batch_pf_info": (C.c_int, [_P, _P, _P]),
    "sipnet_stream_sync": (C.c_int, [_P]),
    "sipnet_stream_create": (_P, [C.c_int32]),
    "sipnet_stream_destroy": (None, [_P]),
    "sipnet_io_read_clim": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_P)]),
    "sipnet_clim_nsteps": (C.c_int32, [_P]),
    "sipnet_clim_data": (_DP, [_P]),
    "sipnet_clim_year": (_I32P, [_P]),
    "sipnet_clim_day": (_I32P, [_P]),
    "sipnet_clim_free": (None, [_P]),
    "sipnet_io_read_params": (C.c_int, [C.c_char_p, _I32P, _P, _P]),
    "sipnet_param_name": (C.c_char_p, [C.c_int32]),
    "sipnet_param_index": (C.c_int32, [C.c_char_p]),
    "sipnet_io_read_events": (C.c_int, [C.c_char_p, _I32P, _P, C.POINTER(C.POINTER(Event)), _I32P]),
    "sipnet_io_free": (None, [_P]),
    "sipnet_io_format_out_header": (C.c_int, [C.c_char_p, C.c_size_t]),
    "sipnet_io_format_out_row": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, C.c_double, _P, C.c_int64]),
    "sipnet_io_write_out": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "sipnet_io_write_debug_logs": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "sipnet_io_write_events_out": (C.c_int, [C.c_char_p, C.c_int32, _I32P, _P, C.c_int32, _P, _P, _P,
                                             C.c_int32, _P, _P, _P]),
    "sipnet_io_ensemble_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P,
                                            C.c_int32, C.c_char_p, C.POINTER(_P)]),
    "sipnet_io_ensemble_put": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64,
                                         C.c_int32]),
    "sipnet_io_ensemble_close": (C.c_int, [_P]),
    "sipnet_io_out_column_count": (C.c_int32, []),
    "sipnet_io_out_column_index": (C.c_int32, [C.c_char_p]),
    "sipnet_io_out_column": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), _I32P, _I32P, C.POINTER(C.c_char_p)]),
    "sipnet_io_write_ensemble_block": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int64,
                                                 C.c_char_p, C.c_int32, C.c_char_p]),
    "sipnet_io_read_restart": (C.c_int, [C.c_char_p, _P]),
    "sipnet_io_write_restart": (C.c_int, [C.c_char_p, _P]),
    "sipnet_restart_check": (C.c_int, [_P, _I32P, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                       C.c_double, _I32P]),
    "sipnet_restart_check_boundary_for_write": (C.c_int, [_P, _I32P]),
}

_lib = None
_DEV_LIBRARY = False


def use_library(path):
    """Development hook (tools/variant_bench.py): load another build of the C-ABI library
    instead of the in-tree product.  Must be called before the first lib(); never read from
    the environment."""
    global LIB_PATH, _DEV_LIBRARY
    if _lib is not None:
        raise RuntimeError("sipnet_amd: the library is already loaded")
    LIB_PATH = os.path.abspath(path)
    _DEV_LIBRARY = True


def lib():
    """Load (once) and return the C-ABI library."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "sipnet_amd has no CPU fallback.")
        # Load order matters inside a PyTorch process: torch bundles its own HIP runtime
        # (SONAME libamdhip64.so.7, requested by torch as "libamdhip64.so").  If this
        # library pulled in /opt/rocm's copy first, torch would load a second runtime and
        # one of the two would see no devices.  Importing torch first makes our NEEDED
        # entry resolve to the runtime torch already loaded, so streams and device
        # pointers are shared.  (The C++ CLI has no torch and uses /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if _DEV_LIBRARY and not hasattr(_lib, name):
                continue   # (an older / experimental build under build/variants: tools only, never the product)
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _strip_comments(text):
    """C / C++ source without its comments and with runs of white space collapsed (string and character literals kept as they
    are): what the compiler sees of it, near enough -- a comment edit must not make a committed counter profile look stale"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                   # a literal: copy up to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the device sources the step kernels are built from -- comments and white space
    left out --: stamped into profiles/pmc_traffic.json by the profiling target and checked by bench.py, so that a counter
    collected on an older kernel is not quoted under a fresh kernel time"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    # (the step kernels and what they include; not the engine / filter / node sources around them)
    for name in ("step_kernel.hip", "step_fast.hip", "fast_body.inc", "step_coop.hip", "coop_mailboxes.inc", "coop_stats.inc", "coop_wave_carbon.inc",
                 "coop_wave_factor.inc", "coop_wave_light.inc", "coop_wave_soil.inc", "coop_wave_water.inc", "step_kernel.h",
                 "fast_math.h", "coop_probes.h", "plan.h"):
        h.update(name.encode())
        h.update(_strip_comments(open(os.path.join(csrc, name), "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


class SipnetError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().sipnet_last_error().decode(errors="replace")
        super().__init__(f"{where}: status {code}: {msg}")


def check(code, where=""):
    if code != OK:
        raise SipnetError(code, where)

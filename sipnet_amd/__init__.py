"""sipnet_amd -- MI355X-native SIPNET flux-integration engine.

Host-side mirror of the reference's C interface for the hot path
(/root/reference/src/sipnet/sipnet.h:26-64) over the C-ABI in
include/sipnet_amd.h.  PyTorch is used only for device memory, streams and
torch.distributed; all model arithmetic runs in hand-written HIP kernels
(sipnet_amd/csrc/step_kernel.hip).
"""
from ._lib import (KERNEL_AUTO, KERNEL_COOP_HBM, KERNEL_COOP_LDS, KERNEL_COOP_PAIR, KERNEL_COOP_QUAD, KERNEL_COOP_NCYCLE, KERNEL_COOP_NCYCLE_PAIR, KERNEL_ONE_WAVE, KERNEL_STRICT,
                   KOPT_FULL_STATE, KOPT_NO_REGULAR_TILES, KOPT_ONE_WAVE_PER_SIMD, KOPT_RUNTIME_FLAGS,
                   KOPT_STATS_IN_KERNEL, KOPT_BOUNDED_WAITS, KOPT_WAIT_SELFTEST, KOPT_HOST_PLAN, KOPT_DEVICE_PLAN,
                   KOPT_PF_MULTI_LAUNCH, KOPT_PF_MOVE_PARAMS, PF_VOID_TOTAL)
from ._lib import (F32_MIXED, F64, NCLIM, NFLAGS, NPARAMS, NREC, NSTATE, RING_SLOTS,
                   Event, Restart, SipnetError, lib)
from .config import (DEFAULT_FLAGS, FLAG_NAMES, PARAM_NAMES, flags_from, read_config)
from .io import (ClimTable, format_out_header, format_out_row, read_clim, read_events,
                 read_params, read_restart, check_restart, write_debug_logs, write_events_out, write_out,
                 write_restart)
from .batch import Batch

__all__ = [
    "Batch", "ClimTable", "Event", "Restart", "SipnetError", "read_restart", "write_restart",
    "check_restart", "lib", "read_clim", "read_params",
    "read_events", "write_out", "write_events_out", "write_debug_logs", "format_out_header", "format_out_row", "read_config",
    "flags_from", "FLAG_NAMES", "DEFAULT_FLAGS", "PARAM_NAMES", "F64", "F32_MIXED",
    "KERNEL_AUTO", "KERNEL_ONE_WAVE", "KERNEL_COOP_LDS", "KERNEL_COOP_HBM", "KERNEL_COOP_PAIR", "KERNEL_COOP_QUAD", "KERNEL_COOP_NCYCLE", "KERNEL_COOP_NCYCLE_PAIR", "KERNEL_STRICT",
    "KOPT_ONE_WAVE_PER_SIMD", "KOPT_RUNTIME_FLAGS", "KOPT_FULL_STATE", "KOPT_NO_REGULAR_TILES", "KOPT_STATS_IN_KERNEL", "KOPT_BOUNDED_WAITS", "KOPT_WAIT_SELFTEST", "KOPT_HOST_PLAN", "KOPT_DEVICE_PLAN", "KOPT_PF_MULTI_LAUNCH", "KOPT_PF_MOVE_PARAMS", "PF_VOID_TOTAL",
    "NPARAMS", "NFLAGS", "NCLIM", "NREC", "NSTATE", "RING_SLOTS",
]

"""The drop-in boundary: libsipnet_amd.so loads, exports every symbol that
include/sipnet_amd.h declares, and refuses compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import sipnet_amd as sa
from sipnet_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "sipnet_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sipnet_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = sa.lib()
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the Python binding knows the signature of each one
    unbound = [n for n in names if n not in _lib.SIGNATURES]
    assert not unbound, unbound


def test_version_and_param_table():
    L = sa.lib()
    assert b"sipnet_amd" in L.sipnet_version()
    names = sa.config._load_names()
    assert len(names) == 80 and names[0] == "plantWoodInit" and names[79] == "soilCSaturation"
    assert names[9] == "" and names[47] == ""          # psnTMax, coarseRootAllocation are derived
    assert sa.config.param_index("MINERALNINIT") == 60  # names are case-insensitive
    assert sa.config.param_index("nope") == -1


def test_param_def_order_matches_oracle_record():
    """include/sipnet_params.def is the single source of the vector layout."""
    lines = [l for l in open(os.path.join(REPO, "include", "sipnet_params.def"))
             if l.startswith("SIPNET_PARAM(")]
    idx = [int(re.match(r"SIPNET_PARAM\(\s*(\d+)", l).group(1)) for l in lines]
    assert idx == list(range(80))


@pytest.mark.skipif(sa.lib().sipnet_device_count() > 0, reason="a GPU is present")
def test_compute_fails_loudly_without_a_gpu():
    L = sa.lib()
    h = C.c_void_p()
    fl = (C.c_int32 * 12)(*sa.flags_from())
    rc = L.sipnet_batch_create(fl, 1, 64, sa.F64, 0, C.byref(h))
    assert rc == _lib.ERR_NO_DEVICE
    assert b"no usable HIP device" in L.sipnet_last_error()
    with pytest.raises(RuntimeError):
        sa.Batch(sa.flags_from(), 1, 64)


def test_the_communicator_refuses_bad_arguments_and_a_missing_device():
    """sipnet_comm_*: argument checks answer before RCCL or a device is touched; on a box without a GPU a well-formed call fails
    loudly too (no RCCL without a device, or no device: SIPNET_ERR_NO_DEVICE either way)"""
    L = sa.lib()
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert L.sipnet_comm_create(None, 1, 0, 0, C.byref(h)) == _lib.ERR_BAD_ARGUMENT
    assert L.sipnet_comm_create(ident, 0, 0, 0, C.byref(h)) == _lib.ERR_BAD_ARGUMENT
    assert L.sipnet_comm_create(ident, 2, 2, 0, C.byref(h)) == _lib.ERR_BAD_ARGUMENT
    assert L.sipnet_comm_all_gather(None, None, None, 8, None) == _lib.ERR_BAD_ARGUMENT
    assert L.sipnet_comm_world(None) == 0
    L.sipnet_comm_destroy(None)
    if L.sipnet_device_count() == 0:
        assert L.sipnet_comm_create(ident, 1, 0, 0, C.byref(h)) == _lib.ERR_NO_DEVICE


def test_flag_coupling_rules_are_enforced():
    """context.c:195-223 -> SIPNET_ERR_BAD_PARAMETER (the reference exits with code 3)."""
    L = sa.lib()
    h = C.c_void_p()
    for bad in (dict(soilPhenol=1, gdd=1), dict(nitrogenCycle=1), dict(anaerobic=1, waterHResp=0),
                dict(carbonSaturation=1)):
        fl = (C.c_int32 * 12)(*sa.flags_from(**bad))
        assert L.sipnet_batch_create(fl, 1, 1, sa.F64, 0, C.byref(h)) == _lib.ERR_BAD_PARAMETER


def test_python_constants_mirror_the_header_enums():
    """the ctypes mirror's KERNEL_* / KOPT_* / MATH_* numbers are the header's (enum values are
    part of the C-ABI: a caller passes plain ints)"""
    import re
    import sipnet_amd as sa
    text = open(os.path.join(REPO, "include", "sipnet_amd.h")).read()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(SIPNET_(?:KERNEL|KOPT|MATH)_[A-Z_0-9]+)\s*=\s*(\d+)", text)}
    assert len([k for k in enums if k.startswith("SIPNET_KERNEL_")]) == 9
    for name, value in enums.items():
        py = name[len("SIPNET_"):]
        if py.startswith("MATH_"):
            continue                      # exposed as Batch(fast_math=...) / set_math(bool)
        assert getattr(sa, py) == value, (name, value, getattr(sa, py, None))


def test_every_bench_workload_has_traffic_evidence_for_the_kernel_auto_picks():
    """profiles/pmc_traffic.json (rocprofv3 PMC, tools/distill_profile.py) feeds bench.py's
    `roofline.traffic`: every workload of bench.WORKLOADS has an entry, and the kernel it was collected
    on is the one SIPNET_KERNEL_AUTO picks for that shape on a 256-CU device (sipnet_kernel_choice, the
    function sipnet_batch_run uses) -- a stale entry would silently print `traffic: null`."""
    import json
    import sipnet_amd as sa
    from bench import WORKLOADS
    L = sa.lib()
    traffic = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    family = {sa.KERNEL_ONE_WAVE: "stepFastKernel<", sa.KERNEL_COOP_LDS: "stepCoopKernel<", sa.KERNEL_COOP_PAIR: "stepCoopPairKernel<",
              sa.KERNEL_COOP_QUAD: "stepCoopQuadKernel<", sa.KERNEL_COOP_NCYCLE: "stepCoopNKernel<", sa.KERNEL_COOP_NCYCLE_PAIR: "stepCoopNPairKernel<",
              sa.KERNEL_STRICT: "stepKernel<"}
    for name, wl in WORKLOADS.items():
        flags = (C.c_int32 * 12)(*sa.flags_from(**wl.get("flags", {})))
        prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
        k = L.sipnet_kernel_choice(flags, wl["sites"], wl["members"], prec, 1, 0, 256)
        assert k in family, (name, k)
        ent = traffic.get(name)
        assert ent and ent.get("hbm_bytes_per_launch", 0) > 0, f"no HBM-traffic evidence for workload {name}"
        fam = family[k]
        fl = wl.get("flags", {})
        if fl and not fl.get("nitrogenCycle") and k in (sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_PAIR):
            fam = fam.replace("stepCoop", "stepCoopX")        # the optional-physics instantiation of that layout
        assert ent.get("kernel", "").startswith(fam), (name, ent.get("kernel"), fam)
        # ... and the ALU cross-check (`roofline.valu_busy_frac`): vector-ALU-active cycles of the same profile,
        # a plausible share of the chip's SIMD-cycles over the profiled launch
        assert ent.get("valu_active_cycles_per_launch", 0) > 0, f"no SQ_ACTIVE_INST_VALU evidence for workload {name}"
        busy = ent["valu_active_cycles_per_launch"] / (1024 * ent["profiled_kernel_avg_ns"] * 2.4)
        assert 0.005 < busy < 1.0, (name, busy)
        if k == sa.KERNEL_COOP_LDS and fam == "stepCoopKernel<":       # the ring-in-LDS instantiation: third template argument true
            assert ent["kernel"].split(",")[2].strip() == "true", ent["kernel"]


def test_auto_kernel_policy_by_shape_and_flags():
    """sipnet_kernel_choice (host-only; the function sipnet_batch_run uses): layouts by chunks per CU, the flags
    that are data (events, gdd, soil_phenol, water_hresp) keep the default-physics kernels, the nitrogen-cycle
    sets have their own one- and two-chunk kernels, every other optional flag takes the optional-physics
    instantiations of the one- and two-chunk layouts, strict math the strict kernel"""
    import sipnet_amd as sa
    L = sa.lib()

    def choice(members, sites=1, prec=sa.F64, math=1, full=0, cus=256, **fl):
        flags = (C.c_int32 * 12)(*sa.flags_from(**fl))
        return L.sipnet_kernel_choice(flags, sites, members, prec, math, full, cus)
    assert choice(1024) == choice(16384) == sa.KERNEL_COOP_LDS                 # <= 1 chunk per CU
    assert choice(16385) == choice(32768) == sa.KERNEL_COOP_PAIR               # <= 2
    assert choice(32769) == choice(65536) == sa.KERNEL_COOP_QUAD               # <= 4, lean
    assert choice(65536, full=1) == choice(65537) == sa.KERNEL_ONE_WAVE
    assert choice(1024, sites=32) == sa.KERNEL_COOP_PAIR                       # C4: 512 chunks
    assert choice(10240, math=0) == sa.KERNEL_STRICT
    assert choice(10240, math=0, prec=sa.F32_MIXED) == sa.KERNEL_COOP_LDS      # fp32-mixed is always fast
    for data_flags in (dict(events=0), dict(gdd=0), dict(gdd=0, soilPhenol=1), dict(waterHResp=0),
                       dict(events=0, gdd=0, soilPhenol=1)):                   # (the last: russell_4)
        assert choice(10240, **data_flags) == sa.KERNEL_COOP_LDS, data_flags
        assert choice(1024, sites=32, **data_flags) == sa.KERNEL_COOP_PAIR, data_flags
    ncyc = dict(litterPool=1, anaerobic=1, nitrogenCycle=1)
    assert choice(10240, **ncyc) == choice(10240, gdd=0, soilPhenol=1, **ncyc) == sa.KERNEL_COOP_NCYCLE
    assert choice(1024, sites=32, **ncyc) == sa.KERNEL_COOP_NCYCLE_PAIR
    assert choice(32769, **ncyc) == sa.KERNEL_ONE_WAVE
    assert choice(1024, sites=32, full=2, **ncyc) == sa.KERNEL_COOP_NCYCLE_PAIR   # (2: diagnostics counters -- round 6: one slot of the mass-total rows per chunk)
    assert choice(10240, full=2, **ncyc) == sa.KERNEL_COOP_NCYCLE           # (round 5: the soil wave runs the balance check)
    assert choice(10240, full=1, **ncyc) == sa.KERNEL_COOP_NCYCLE and choice(1024, sites=32, full=1, **ncyc) == sa.KERNEL_COOP_NCYCLE_PAIR
    # every other optional flag: the optional-physics instantiations of the one- and two-chunk layouts (lean or full
    # state), the one-wave kernel beyond two chunks per CU; with the nitrogen cycle on top: its kernels
    for other in (dict(growthResp=1), dict(leafWater=1), dict(litterPool=1), dict(flooding=1),
                  dict(litterPool=1, carbonSaturation=1), dict(anaerobic=1),
                  dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0)):                  # (the last: russell_3)
        assert choice(10240, **other) == sa.KERNEL_COOP_LDS, other
        assert choice(1024, sites=32, **other) == sa.KERNEL_COOP_PAIR, other
        assert choice(65536, **other) == sa.KERNEL_ONE_WAVE and choice(10240, full=1, **other) == sa.KERNEL_COOP_LDS, other
        # (round 5: four chunks per CU in an fp32-mixed batch -- stepCoopXQuadKernel; the fp64 build would spill)
        assert choice(65536, prec=sa.F32_MIXED, **other) == sa.KERNEL_COOP_QUAD, other
        assert choice(65536, prec=sa.F32_MIXED, full=1, **other) == choice(65537, prec=sa.F32_MIXED, **other) == sa.KERNEL_ONE_WAVE
    everything = dict(carbonSaturation=1, flooding=1, growthResp=1, leafWater=1, **ncyc)
    assert choice(10240, **everything) == sa.KERNEL_COOP_NCYCLE and choice(1024, sites=32, **everything) == sa.KERNEL_COOP_NCYCLE_PAIR
    assert choice(10240, full=1, **everything) == sa.KERNEL_COOP_NCYCLE          # (round 5: stepCoopNXFullKernel)
    assert choice(1024, sites=32, full=1, **everything) == sa.KERNEL_COOP_NCYCLE_PAIR
    assert choice(10240, full=2, **everything) == sa.KERNEL_COOP_NCYCLE          # diagnostics counters with the nitrogen cycle
    assert choice(1024, sites=32, full=2, **everything) == sa.KERNEL_COOP_NCYCLE_PAIR    # (round 6)
    assert choice(0) == -1 and choice(64, cus=0) == -1

/*
 * sipnet_amd.h -- C-ABI of the MI355X-native SIPNET flux-integration engine.
 *
 * The reference (PecanProject/sipnet) has no plugin / FFI layer.  Its de-facto
 * C boundary for the hot path is the five functions of
 * /root/reference/src/sipnet/sipnet.h:26-64, called from frontend.c:212-250:
 *
 *     initModel(&mp, paramFile, climFile)        sipnet.h:36  (sipnet.c:2001)
 *     setupModel()                               sipnet.h:45  (sipnet.c:1858)
 *     runModelOutput(out, dbg, items, header)    sipnet.h:53  (sipnet.c:1954)
 *     setupOutputItems(items)                    sipnet.h:59  (sipnet.c:1993)
 *     cleanupModel()                             sipnet.h:64  (sipnet.c:2012)
 *
 * Those operate on process globals, one member of one site at a time.  This
 * header is the batched, re-entrant replacement a reference-side binding would
 * call instead: an opaque handle holding an ensemble x site batch in HBM, plain
 * pointers and sizes only, int status returns (0 = ok, otherwise the reference
 * exit code of common/exitCodes.h:16-27 that the same condition produces in
 * the reference), no globals.  Mapping:
 *
 *     initModel        -> sipnet_io_read_params + sipnet_io_read_clim
 *                         + sipnet_batch_create/_set_params/_set_climate
 *     initEvents       -> sipnet_io_read_events + sipnet_batch_set_events
 *                         (events.c:427-433)
 *     setupModel       -> sipnet_batch_setup
 *     runModelOutput   -> sipnet_batch_run (+ sipnet_io_write_out_rows for text)
 *     setupOutputItems -> the NEE/GPP planes of sipnet_batch_run and
 *                         sipnet_io_write_single_outputs
 *     cleanupModel     -> sipnet_batch_destroy
 *
 * All compute entry points enqueue work on the HIP stream passed in and need a
 * gfx950 device; they return SIPNET_ERR_NO_DEVICE (never a CPU fallback) when
 * none is usable.  The sipnet_io_* functions are host-only.
 */
#ifndef SIPNET_AMD_H
#define SIPNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIPNET_NPARAMS 80 /* include/sipnet_params.def */
#define SIPNET_NFLAGS 12
#define SIPNET_NCLIM 11   /* converted climate record, see sipnet_io_read_clim */
#define SIPNET_NREC 44    /* full per-step record: 36 output columns + 8 event-log columns */
#define SIPNET_NDBG 72    /* per-step debug plane (--debug-log): 56 fluxes + 14 tracker fields */
#define SIPNET_NSTATE 32  /* per-member carried state (doubles), see below */
#define SIPNET_RING_SLOTS 250 /* MEAN_NPP_MAX_ENTRIES, sipnet.c:39-40 */

/* model flags, order of struct Context (common/context.h:46-57) */
enum sipnet_flag {
  SIPNET_F_EVENTS = 0,
  SIPNET_F_GDD,
  SIPNET_F_GROWTH_RESP,
  SIPNET_F_LEAF_WATER,
  SIPNET_F_LITTER_POOL,
  SIPNET_F_SNOW,
  SIPNET_F_SOIL_PHENOL,
  SIPNET_F_WATER_HRESP,
  SIPNET_F_NITROGEN_CYCLE,
  SIPNET_F_ANAEROBIC,
  SIPNET_F_FLOODING,
  SIPNET_F_CARBON_SATURATION
};

/* status codes = reference exit codes (common/exitCodes.h:16-27) + own */
enum sipnet_status {
  SIPNET_OK = 0,
  SIPNET_ERR_FAILURE = 1,
  SIPNET_ERR_BAD_PARAMETER = 3,    /* EXIT_CODE_BAD_PARAMETER_VALUE */
  SIPNET_ERR_UNKNOWN_EVENT = 4,    /* EXIT_CODE_UNKNOWN_EVENT_TYPE_OR_PARAM */
  SIPNET_ERR_INPUT_FILE = 5,       /* EXIT_CODE_INPUT_FILE_ERROR */
  SIPNET_ERR_FILE_OPEN = 6,        /* EXIT_CODE_FILE_OPEN_OR_READ_ERROR */
  SIPNET_ERR_INTERNAL = 7,         /* EXIT_CODE_INTERNAL_ERROR */
  SIPNET_ERR_BAD_CLI = 8,          /* EXIT_CODE_BAD_CLI_ARGUMENT */
  SIPNET_ERR_RESTART = 9,          /* EXIT_CODE_BAD_RESTART_PARAMETER */
  SIPNET_ERR_NO_DEVICE = 100,      /* no usable gfx950 device / HIP failure */
  SIPNET_ERR_BAD_ARGUMENT = 101
};

/* arithmetic of the step kernel */
enum sipnet_precision {
  SIPNET_F64 = 0,      /* everything in fp64 (parity configuration) */
  SIPNET_F32_MIXED = 1 /* flux arithmetic fp32, pools and accumulators fp64 */
};

/* event types, order of enum EventType (sipnet/events.h:17-27) */
enum sipnet_event_type {
  SIPNET_EV_FERT = 0,
  SIPNET_EV_HARVEST,
  SIPNET_EV_IRRIG,
  SIPNET_EV_PLANT,
  SIPNET_EV_TILL,
  SIPNET_EV_LEAFON,
  SIPNET_EV_LEAFOFF
};

/* One agronomic event (events.in line).  p[] per type (events.h:29-97):
 *   harvest: fracRemovedAbove fracRemovedBelow fracTransferredAbove fracTransferredBelow
 *   irrig  : amountAdded method(0 canopy, 1 soil)
 *   fert   : orgN orgC minN        plant: leafC woodC fineRootC coarseRootC
 *   till   : tillageEffect         leafon / leafoff: none */
typedef struct sipnet_event {
  int32_t type, year, day, pad;
  double p[4];
} sipnet_event;

/* Full per-step record (SIPNET_NREC doubles), the superset of the `.out` row
 * (sipnet.c:453-473):
 *  0 nee 1 gpp 2 evapotranspiration 3 totNee(cumNEE) 4 npp 5 rAboveground
 *  6 rSoil 7 rRoot 8 ra 9 rh 10 rtot 11 woodCreation 12 soilWetnessFrac
 * 13 transpiration(flux) 14 plantWoodC 15 plantLeafC 16 soilC 17 soilWater
 * 18 litterC 19 snow 20 coarseRootC 21 fineRootC 22 minN 23 soilOrgN
 * 24 litterN 25 plantStorageN 26 plantCAccountingDelta 27 n2o 28 nLeaching
 * 29 nFixation 30 nUptake 31 methane 32 meanNPP 33 gdd 34 d_till_mod 35 totGpp
 * event log (amounts of the computed events of this step, for `events.out`):
 * 36 leafOnCreation*len 37 leafOnCreationFromWood*len 38 computed leaf-off litter*len
 * 39 eventLeafOnCreation*len 40 eventLeafOnCreationFromWood*len
 * 41 totalWoodC and 42 totalRootC at the moment of plant death 43 died-this-step flag
 *
 * Per-member carried state (SIPNET_NSTATE doubles), what the reference's restart
 * schema (restart.c:216-296) persists for one member minus site-uniform items:
 *  0..12 the 13 pools in `Envi` order (state.h:416-463)
 * 13 ring sum (meanNPP.sum) 14..19 tot{Gpp,Rtot,Ra,Rh,Npp,Nee}
 * 20..26 yearly{Gpp,Rtot,Ra,Rh,Npp,Nee,Litter} 27 phenology bits
 * (1 didLeafGrowth | 2 didLeafFall) 28 ring_valid_from (first step whose ring
 * insert is live; earlier slots count as zero) 29 status
 * 30 died_at_step (-1 = never) 31 clamp-warning count */

typedef struct sipnet_batch sipnet_batch;

/* One member's restart checkpoint: the content of a `SIPNET_RESTART` file
 * (sipnet/restart.c:152-305 lists the keys; docs/developer-guide/restart-checkpoint.md).
 * trackers[] holds the 32 double-valued `trackers.*` keys in file order
 * (restart.c:236-268; the integer trackers.lastYear is a separate member). */
enum sipnet_restart_tracker {
  SIPNET_RT_GPP = 0, SIPNET_RT_RTOT, SIPNET_RT_RA, SIPNET_RT_RH, SIPNET_RT_RROOT,
  SIPNET_RT_RSOIL, SIPNET_RT_RABOVEGROUND, SIPNET_RT_NPP, SIPNET_RT_NEE,
  SIPNET_RT_WOODCREATION, SIPNET_RT_GDD, SIPNET_RT_ET, SIPNET_RT_SOILWETNESSFRAC,
  SIPNET_RT_YEARLYGPP, SIPNET_RT_YEARLYRTOT, SIPNET_RT_YEARLYRA, SIPNET_RT_YEARLYRH,
  SIPNET_RT_YEARLYNPP, SIPNET_RT_YEARLYNEE, SIPNET_RT_YEARLYLITTER,
  SIPNET_RT_TOTGPP, SIPNET_RT_TOTRTOT, SIPNET_RT_TOTRA, SIPNET_RT_TOTRH,
  SIPNET_RT_TOTNPP, SIPNET_RT_TOTNEE, SIPNET_RT_METHANE, SIPNET_RT_N2O,
  SIPNET_RT_NLEACHING, SIPNET_RT_NFIXATION, SIPNET_RT_NUPTAKE, SIPNET_RT_MEANNPP,
  SIPNET_RT_COUNT
};
typedef struct sipnet_restart {
  char model_version[32];          /* meta_info.model_version, must equal "2.1.0" */
  char build_info[96];             /* meta_info.build_info (informational) */
  int64_t checkpoint_utc_epoch;    /* meta_info.checkpoint_utc_epoch */
  int64_t processed_steps;         /* meta_info.processed_steps */
  int32_t flags[SIPNET_NFLAGS];    /* flags.* in enum sipnet_flag order */
  int32_t boundary_year, boundary_day; /* last processed climate record */
  double boundary_time, boundary_length;
  double envi[13];                 /* envi.* in `Envi` order (= state rows 0..12) */
  double trackers[SIPNET_RT_COUNT];
  int32_t trackers_last_year;      /* trackers.lastYear */
  int32_t did_leaf_growth, did_leaf_fall, phenology_last_year; /* phenology.* */
  int32_t is_alive;                /* survival.isAlive */
  int32_t mean_length;             /* mean.npp.length, must be SIPNET_RING_SLOTS */
  double d_till_mod, harvest_frac_removed, harvest_frac_transferred; /* event_trackers.* */
  double mean_tot_weight;          /* mean.npp.totWeight (5 days) */
  int32_t mean_start, mean_last;   /* ring cursors */
  double mean_sum;                 /* mean.npp.sum */
  double mean_values[SIPNET_RING_SLOTS];
  double mean_weights[SIPNET_RING_SLOTS];
} sipnet_restart;

/* ------------------------------------------------------------------ library */
const char *sipnet_version(void);      /* "sipnet_amd x.y (reference 2.1.0)" */
const char *sipnet_last_error(void);   /* thread-local message of last failure */
int sipnet_device_count(void);         /* 0 when no HIP device is visible */

/* ------------------------------------------------------------- batch engine */
/* Create an ensemble x site batch on HIP device `device`.
 * Columns are site-major: col = site * n_members + member. */
int sipnet_batch_create(const int32_t flags[SIPNET_NFLAGS], int32_t n_sites,
                        int32_t n_members, int32_t precision, int32_t device,
                        sipnet_batch **out);
void sipnet_batch_destroy(sipnet_batch *b);

/* Forcing of one site: clim[n_steps][SIPNET_NCLIM] already converted exactly
 * as readClimData does (sipnet.c:201-238):
 *   length(d) tair tsoil par(/d) precip(cm) vpd(kPa) vpdSoil vPress wspd gdd time(h)
 * Sites of a batch may differ in n_steps: a launch advances every site to the end of ITS records (sipnet_batch_nsteps
 * = the longest site's; plane / record rows past a site's last record are left untouched, its statistics there are
 * zero with sipnet_batch_run_stats on a cooperative kernel, undefined otherwise).  The arrays are copied (into a pinned
 * block the batch keeps; from there the forcing leaves for the device asynchronously when the site's plan may be built
 * there, SIPNET_KOPT_HOST_PLAN); the site plan (member-independent schedule: running-mean ring weights, GDD sums, year
 * roll-overs, event matching, tillage decay) is built by the next sipnet_batch_setup. */
int sipnet_batch_set_climate(sipnet_batch *b, int32_t site, int32_t n_steps,
                             const double *clim, const int32_t *year,
                             const int32_t *day);
/* The same for sites first_site .. first_site + count - 1 in one call: n_steps[k], clim[k], year[k], day[k] are site
 * first_site + k's.  The copies run on the batch's plan threads (32 sites x 17 520 records: 0.7 ms instead of 2.4). */
int sipnet_batch_set_climate_sites(sipnet_batch *b, int32_t first_site, int32_t count, const int32_t *n_steps,
                                   const double *const *clim, const int32_t *const *year,
                                   const int32_t *const *day);
/* Events of one site in file order; call before sipnet_batch_set_climate or
 * re-call set_climate afterwards (the plan is rebuilt in either order). */
int sipnet_batch_set_events(sipnet_batch *b, int32_t site, int32_t n_events,
                            const sipnet_event *events);
/* Raw parameters of `count` members of `site`, starting at first_member:
 * raw[count][SIPNET_NPARAMS], include/sipnet_params.def order, file units.
 * site = SIPNET_ALL_SITES: the same members at every site of the batch (the usual ensemble: one
 * parameter draw, many sites) -- ONE upload and one conversion launch instead of one per site. */
#define SIPNET_ALL_SITES (-1)
int sipnet_batch_set_params(sipnet_batch *b, int32_t site, int32_t first_member,
                            int32_t count, const double *raw);

/* Arithmetic policy of an fp64 batch (an fp32-mixed batch is always SIPNET_MATH_FAST):
 *   SIPNET_MATH_STRICT  operation order, true divisions and pow / exp calls as the reference
 *                       writes them (differences are OCML-vs-glibc rounding, <= 1.1e-14);
 *                       one-wavefront kernel, all flags, full records
 *   SIPNET_MATH_FAST    the throughput kernels: site-only sub-expressions from the host plan,
 *                       reciprocals instead of divisions, a degree-9 polynomial exp2 (3.7e-14
 *                       relative); <= 2e-14 gC m-2 per step on NEE, GPP and ET against the reference
 *                       over a year of the benchmark ensemble's first 1 024 members (measured 1.9e-14,
 *                       pinned by tests/test_gpu_configs.py::test_c2_...; the bar is 1e-6)
 * A new fp64 batch is STRICT (no environment variable changes that).  May be changed between
 * runs. */
enum sipnet_math { SIPNET_MATH_STRICT = 0, SIPNET_MATH_FAST = 1 };
int sipnet_batch_set_math(sipnet_batch *b, int32_t policy);

/* Which step kernel sipnet_batch_run launches.  AUTO (the default) picks by batch shape (sipnet_kernel_choice
 * answers without a device).  With SIPNET_MATH_STRICT, or for the debug plane: the strict-order kernel.  With
 * SIPNET_MATH_FAST, by 64-member chunks per compute unit (CU):
 *   - flag sets WITHOUT the nitrogen cycle -- the default physics and every combination of growth respiration,
 *     leaf water, flooding, litter pool, carbon saturation, anaerobic + methane (run-time flags of the
 *     optional-physics instantiations) --: up to one chunk per CU one four-wave workgroup per chunk with the
 *     running-mean ring in LDS; up to two, one eight-wave workgroup per TWO chunks (rings in HBM); up to four,
 *     default physics and lean state only, one twelve-wave workgroup per FOUR chunks; beyond that the
 *     one-wavefront kernel;
 *   - flag sets WITH the nitrogen cycle (alone = litter pool + anaerobic + nitrogen cycle, or with any of the other
 *     options): their own cooperative kernels (a soil wave next to light, water, carbon) up to two chunks per CU;
 *     records / SIPNET_KOPT_FULL_STATE there for the nitrogen-cycle set itself and for the sets with further options;
 *     the one-wavefront kernel beyond two chunks per CU;
 *   - records, SIPNET_KOPT_FULL_STATE and the diagnostics counters take the "Full" instantiations of the same
 *     kernels (not the four-chunk layout; the counters with the nitrogen cycle on the one- and -- round 6 -- two-chunk layouts).
 * "Default physics" means the model, not the flag values: events, gdd, soil_phenol and water_hresp may have any
 * legal value (no events; leaf-on by growing degree days, soil temperature or day of year -- russell_4's set; no
 * moisture effect on heterotrophic respiration): they change what the site plan puts into the step records, not the
 * kernels.  The other values of the enum force one kernel (tests and measurements compare every instantiation with
 * the oracle this way); a forced kernel that cannot run the batch (a throughput kernel under SIPNET_MATH_STRICT,
 * the four-chunk layout with optional physics or full state, a nitrogen-cycle kernel without the nitrogen cycle or
 * the other way round) makes sipnet_batch_run return SIPNET_ERR_BAD_ARGUMENT.  Nothing in the launch path reads the
 * environment.
 * What AUTO hands to the one-wavefront kernel although a cooperative layout exists for a neighbouring shape, why, and what
 * it costs (MI355X; profiles/r05_flag_sets.md, r05_bench_all.jsonl, tools/big_batch_time.py):
 *   shape (SIPNET_MATH_FAST)                                   why no cooperative kernel                          cost
 *   > 4 chunks per CU, any flag set                            the one-wave kernel IS the faster one there        none (131 072 fp32 members x 1 year:
 *                                                              (two waves per SIMD hide each other's latencies)   19.3 ms against 20.0 four-chunk cooperative)
 *   2 < chunks per CU <= 4, optional physics, fp64             the four-chunk build needs 198 VGPRs of the        x 2.0 - 2.2 (c3's shape: 21 - 23 ms against
 *                                                              layout's 168 (three waves per SIMD): it would      10.2 - 10.7 ms of the fp32-mixed build, which
 *                                                              spill in the carbon wave's loop                    exists: use SIPNET_F32_MIXED there)
 *   > 2 chunks per CU, nitrogen cycle (any precision)          a chunk's soil-wave mailboxes + four private       x 2.2 (c3's shape, fp32-mixed: 21.0 ms against
 *                                                              record tiles are ~60 KB of LDS: four chunks do     9.6 ms for the default physics)
 *                                                              not fit a CU's 160 KB
 *   > 2 chunks per CU, records / SIPNET_KOPT_FULL_STATE /      no full-state build of the four-chunk layout:      x 1.5 (c10k's members x 4, with the record:
 *   diagnostics                                                the record columns and accumulators spill under    one-wave Full build; the record's 44 stores
 *                                                              its register budget (fp64 284 B, fp32 136 B per    per member-step dominate either way)
 *                                                              lane)
 * sipnet_kernel_choice answers for any shape without a device; sipnet_batch_last_launch names what a launch took. */
enum sipnet_kernel {
  SIPNET_KERNEL_AUTO = 0,
  SIPNET_KERNEL_ONE_WAVE = 1, /* stepFastKernel: one wavefront per 64 members */
  SIPNET_KERNEL_COOP_LDS = 2, /* stepCoopKernel, ring in LDS (one workgroup per CU) */
  SIPNET_KERNEL_COOP_HBM = 3, /* stepCoopKernel, ring in HBM */
  SIPNET_KERNEL_STRICT = 4,   /* stepKernel (with SIPNET_MATH_FAST: its fast-math variant) */
  SIPNET_KERNEL_COOP_PAIR = 5, /* stepCoopPairKernel: two chunks per workgroup, ring in HBM */
  SIPNET_KERNEL_COOP_QUAD = 6, /* stepCoopQuadKernel: four chunks per twelve-wave workgroup */
  SIPNET_KERNEL_COOP_NCYCLE = 7, /* stepCoopNKernel: the nitrogen-cycle flag set (litter pool + anaerobic +
                                   nitrogen cycle), four wavefronts per chunk, soil and nitrogen on a
                                   wavefront of their own */
  SIPNET_KERNEL_COOP_NCYCLE_PAIR = 8 /* stepCoopNPairKernel: the same, two chunks per eight-wave workgroup
                                   (AUTO's choice between one and two chunks per CU) */
};
enum sipnet_kernel_option {
  SIPNET_KOPT_ONE_WAVE_PER_SIMD = 1, /* one-wave kernel: never the 256-VGPR (two waves/SIMD) build */
  SIPNET_KOPT_RUNTIME_FLAGS = 2,     /* one-wave kernel: always the run-time-flag instantiation */
  SIPNET_KOPT_NO_REGULAR_TILES = 8,  /* cooperative kernel: always the general per-step path, never the
                                        record-free path of regular 16-step tiles (A/B measurement,
                                        tests of the two paths against each other) */
  SIPNET_KOPT_STATS_IN_KERNEL = 16,  /* sipnet_batch_run_stats: sum the plane tiles inside the step kernel on
                                        EVERY cooperative layout (default: only where a workgroup has a
                                        compute unit to itself and its fourth wavefront does the summing;
                                        on the two- / four-chunk layouts three reduction passes are cheaper) */
  SIPNET_KOPT_BOUNDED_WAITS = 32,    /* cooperative kernels, lean launches: the build whose hand-over waits have a budget of
                                        polls (the product's spin without an exit: a protocol bug would hang the GPU).  A wait
                                        that exhausts it ends the launch; sipnet_batch_run then synchronises the stream and
                                        answers SIPNET_ERR_INTERNAL naming the wait and the step.  ~10 % slower: for fuzz
                                        campaigns and tests, never chosen by SIPNET_KERNEL_AUTO */
  SIPNET_KOPT_WAIT_SELFTEST = 64,    /* with SIPNET_KOPT_BOUNDED_WAITS only: the light wavefront stops posting after 100 steps --
                                        the test of the error path itself (the launch must end with SIPNET_ERR_INTERNAL) */
  SIPNET_KOPT_HOST_PLAN = 128,       /* build every site's per-step records on host threads (plan.cpp).  Without it a site
                                        whose steps are all at least 0.0202 days long and come in long runs of equal length
                                        has them built on the DEVICE from its climate (csrc/plan_device.h: 63 MB instead of
                                        143 MB over PCIe at 32 sites x 17 520 records; fresh or resumed, with or without
                                        events) -- when the batch is idle at the hand-over; while its previous launch still
                                        runs (a caller pipelining forcings) the host's idle cores build them.  The records are
                                        the same bytes either way */
  SIPNET_KOPT_DEVICE_PLAN = 256,     /* build a site's records on the device whenever it CAN (steps >= 0.0202 d, no site-fatal
                                        condition): also while the batch is busy, also when its step lengths do not come in
                                        long runs -- the default leaves such a forcing (half-daily niwot) to the host, whose
                                        cores walk the ring's schedule ~8 x faster than the one lane that has to on the
                                        device (tests use this) */
  SIPNET_KOPT_PF_MULTI_LAUNCH = 512, /* particle filter: the analysis as separate launches (log-weights | fixed-point weights |
                                        prefix sum | ancestors), never the one-launch kernel whose workgroups spin at barriers
                                        in device memory.  Chosen without being asked when fewer than 8 such workgroups could
                                        be resident (sipnet_batch_set_device_share, a sliver of a partitioned device) */
  SIPNET_KOPT_PF_MOVE_PARAMS = 1024, /* particle filter across ranks, particles carrying their parameters: move the 640 bytes
                                        of converted parameter rows with every resampled particle (round 5) instead of
                                        replicating all ranks' parameters once at sipnet_batch_pf_connect and moving a 4-byte
                                        index (the default: 80 x 8 bytes x all ranks' particles of HBM per rank) */
  SIPNET_KOPT_FULL_STATE = 4         /* throughput kernels: advance EVERY accumulator of the restart
                                        schema (trackers.tot*, trackers.yearly*); without it only
                                        totNee / totGpp advance on the throughput path.  Implied by a
                                        full record (d_rec) and by enabled diagnostics */
};
int sipnet_batch_set_kernel(sipnet_batch *b, int32_t kernel, int32_t options);
/* What SIPNET_KERNEL_AUTO picks for a batch shape on a device with num_cus compute units (MI355X: 256):
 * host-only, no device needed (tools and tests ask it; sipnet_batch_run uses the same function).
 * math = enum sipnet_math (ignored for SIPNET_F32_MIXED); want_full = 0 lean, 1 records /
 * SIPNET_KOPT_FULL_STATE, 2 the diagnostics counters as well.  Returns an enum sipnet_kernel, -1 on a bad argument. */
int32_t sipnet_kernel_choice(const int32_t flags[SIPNET_NFLAGS], int32_t n_sites, int32_t n_members,
                             int32_t precision, int32_t math, int32_t want_full, int32_t num_cus);

/* Per-member initialisation == setupModel() (sipnet.c:1858-1951): parameter
 * unit conversion, derived parameters, initial pools, trackers, phenology state
 * from the first climate record, ring reset.  Members whose allocation
 * parameters are invalid get status SIPNET_ERR_BAD_PARAMETER and are skipped
 * by run (the reference exits, sipnet.c:1117-1122). */
int sipnet_batch_setup(sipnet_batch *b, void *hip_stream);

/* Advance every member n_steps steps starting at climate record step0
 * (== the while loop of runModelOutput, sipnet.c:1969-1982).  State stays in
 * HBM between calls, so a run may be split at any step boundary.
 *
 * d_nee/d_gpp/d_et: DEVICE pointers (or NULL) to planes [n_steps][ld] written
 * as plane[t * ld + col]; element type double for SIPNET_F64, float for
 * SIPNET_F32_MIXED; ld >= n_sites*n_members.  Plane entries of members whose status is
 * non-zero (skipped members) are undefined.
 * d_rec: DEVICE pointer (or NULL) to [n_steps][SIPNET_NREC][ld] doubles
 * (full record, for `.out` text and checkpoints); written by whichever kernel the policy picks
 * (SIPNET_MATH_FAST: the throughput kernels' Full instantiations). */
int sipnet_batch_run(sipnet_batch *b, int32_t step0, int32_t n_steps,
                     void *d_nee, void *d_gpp, void *d_et, double *d_rec,
                     int64_t ld, void *hip_stream);

/* Per-member counts of the reference's per-step self-checks (diagnostics only, no effect on
 * state): n_clamp_warn = ensureNonNegative() warnings, a pool clamped to 0 from |v| > 1e-8
 * (sipnet.c:1346-1356); n_balance_warn = checkBalance() warnings, |carbon or nitrogen mass
 * balance residual of a step| >= 1e-8 (balance.c:122-169); max_abs_dC / max_abs_dN = the largest
 * residual seen.  Enabled per batch (a [4][ncol] device block; every step kernel then runs its
 * counting variant), zeroed by sipnet_batch_setup, accumulated over launches, read back into HOST
 * arrays of ncol entries (any may be NULL). */
int sipnet_batch_enable_diagnostics(sipnet_batch *b, int32_t on);
int sipnet_batch_get_diagnostics(sipnet_batch *b, int64_t *n_clamp_warn, int64_t *n_balance_warn,
                                 double *max_abs_dC, double *max_abs_dN, void *hip_stream);

/* Every member's SUMS over groups of sum_steps consecutive steps instead of the steps themselves (what a consumer of the
 * reference's per-step output rows, sipnet.c:453-473, aggregates anyway: sum_steps = 48 gives the daily NEE / GPP / ET of a
 * half-hourly forcing): d_*_sums[ceil(n_steps / sum_steps)][ld] doubles (DEVICE; any may be NULL), groups counted from
 * step0, the last as long as the run leaves it, each value added in step order by the wavefront that computes it -- inside
 * the step kernel's own launch (1 / sum_steps of the planes' HBM writes, no second pass).  The sums are doubles whatever
 * the batch's arithmetic: an fp32-mixed batch widens each float value and adds it in double, so the result is bit for bit its
 * float planes added up that way.  For batches sipnet_batch_sums_in_kernel answers 1 for: SIPNET_MATH_FAST (fp64 or
 * fp32-mixed), any flag set and shape, no diagnostics / full state -- the cooperative kernels' Sums instantiations: fp64 up to two 64-member chunks per compute unit stepCoopSumsKernel,
 * stepCoopPairSumsKernel, their optional-physics relatives stepCoopXSumsKernel / XPairSums and the nitrogen-cycle layouts'
 * stepCoopNSumsKernel / NPairSums, where the soil wave forms NEE and sums it; fp32-mixed batches on every layout and fp64 on
 * the four-chunk layout stepCoopSumsAtKernel<arithmetic, plain exponents, layout, optional physics> (step_coop_sums.hip);
 * batches on the one-wavefront kernel (more than four chunks per compute unit, or forced) stepFastSumsKernel
 * (step_fast_sums.hip).  The cooperative kernels' sums are bit for bit the same batch's planes added up in step order, and
 * the state they leave is the plain launch's; the one-wavefront sums build equals its plain build to the arithmetic's last
 * bits only (both are compiled with -ffp-contract=fast, and without the plane stores a few products on rare paths fuse
 * differently: one flux in ~10^6 differs in its last bit).  Cost against the planes' launch (MI355X,
 * profiles/r06_sums_time.txt): 10 240 fp64 members +0.08 ms of 8.31, c4's shape -0.10 of 9.68, 65 536 fp32-mixed members
 * (c3) +0.16 of 9.20.  A batch under SIPNET_MATH_STRICT, with diagnostics or SIPNET_KOPT_FULL_STATE gets
 * SIPNET_ERR_BAD_ARGUMENT and sums its planes (sipnet_node_run_gathering_reduced does either by itself).  A
 * site that ends inside a group leaves the group's sum over its own records. */
int sipnet_batch_run_sums(sipnet_batch *b, int32_t step0, int32_t n_steps, int32_t sum_steps, double *d_nee_sums,
                          double *d_gpp_sums, double *d_et_sums, int64_t ld, void *hip_stream);
int32_t sipnet_batch_sums_in_kernel(const sipnet_batch *b);

/* The same advance with the reference's `--debug-log` content (outputDebugState,
 * debug_log.c:285-312, called after every updateState, sipnet.c:1974): besides the full
 * record d_rec, d_dbg[n_steps][SIPNET_NDBG][ld] (DEVICE, doubles) receives per step
 *   0..55  the 56 `Fluxes` fields in the order of the fluxes log (debug_log.c:70-125)
 *   56..62 trackers.yearlyGpp, yearlyRtot, yearlyRa, yearlyRh, yearlyNpp, yearlyNee, yearlyLitter
 *   63..66 trackers.totRtot, totRa, totRh, totNpp
 *   67..69 phenologyTrackers.didLeafGrowth, didLeafFall, plantSurvivalTracker.isAlive
 * Always runs the strict-order kernel (no fast-math substitutions of the flux expressions
 * unless the batch is fast-math). */
int sipnet_batch_run_debug(sipnet_batch *b, int32_t step0, int32_t n_steps, double *d_rec,
                           double *d_dbg, int64_t ld, void *hip_stream);

/* Ensemble statistics of an output plane: for every step and site, the sum and
 * sum of squares over the site's members (wavefront-shuffle reduction):
 *   d_stats[(t * n_sites + site) * 2 + {0,1}]   (double)
 * `elem_is_f32` selects the plane's element type. */
int sipnet_batch_reduce_plane(sipnet_batch *b, const void *d_plane,
                              int32_t elem_is_f32, int32_t n_steps, int64_t ld,
                              double *d_stats, void *hip_stream);

/* sipnet_batch_run with the ensemble statistics of the three planes produced by the same
 * launch: d_stats[((v * n_steps + t) * n_sites + site) * 2 + {0,1}] (DEVICE, doubles; v = 0 NEE,
 * 1 GPP, 2 ET) = sum / sum of squares over the site's members of step step0 + t -- the block the
 * ranks of a multi-GPU run all-gather.  All three planes are required.  On the
 * one-chunk-per-compute-unit cooperative kernel (batches of up to 64 x #CUs members: c10k, c2 and
 * its stacked form) the workgroup's fourth wavefront sums the tiles the carbon and water wavefronts
 * have just stored (from L2: the planes are not read from HBM again) and a small second kernel adds
 * the chunks of a site up; on the other kernels the launch is followed by three
 * sipnet_batch_reduce_plane passes.  Both give the same sums up to the order of the additions. */
int sipnet_batch_run_stats(sipnet_batch *b, int32_t step0, int32_t n_steps, void *d_nee,
                           void *d_gpp, void *d_et, int64_t ld, double *d_stats,
                           void *hip_stream);

/* Copy per-member state to / from HOST memory: state[ncol][SIPNET_NSTATE]. */
int sipnet_batch_get_state(sipnet_batch *b, double *state, void *hip_stream);
int sipnet_batch_set_state(sipnet_batch *b, const double *state, void *hip_stream);
/* Ring contents of one column to HOST: values[SIPNET_RING_SLOTS]. */
int sipnet_batch_get_ring(sipnet_batch *b, int64_t col, double *values,
                          void *hip_stream);
/* Whole ring block to / from HOST: rings[ncol][SIPNET_RING_SLOTS].  Together with the state
 * vector this is the complete per-member checkpoint (what restart.c:216-296 persists). */
int sipnet_batch_get_rings(sipnet_batch *b, double *rings, void *hip_stream);
int sipnet_batch_set_rings(sipnet_batch *b, const double *rings, void *hip_stream);
/* Per-member status to HOST: status[ncol] (enum sipnet_status). */
int sipnet_batch_get_status(sipnet_batch *b, int32_t *status, void *hip_stream);

/* ---- restart checkpoints (runModelOutput's hooks, sipnet.c:1965-1989) ----
 * Resume sequence == the reference's (restart-checkpoint.md "Runtime sequence"):
 *   set_climate/events/params of the NEW segment -> set_resume -> setup ->
 *   import_restart (overwrites the members' state) -> run.
 *
 * set_resume: the site-uniform part of a checkpoint that the site plan owns --
 * trackers.gdd, trackers.lastYear, phenology.lastYear, event_trackers.d_till_mod and the
 * ring layout (weights, cursors).  r == NULL clears it.  Call before setup. */
int sipnet_batch_set_resume(sipnet_batch *b, int32_t site, const sipnet_restart *r);
/* import_restart: after setup, overwrite the carried state and ring values of `count`
 * members of `site` with r[count].  A member whose ring layout differs from the site's
 * (a member that died and was re-planted in the earlier segment) is re-laid onto the
 * site layout when its older entries are all zero, else SIPNET_ERR_RESTART.
 * survival.isAlive is implied by the pools (sipnet.c:1530-1544); a checkpoint that
 * contradicts them is rejected. */
int sipnet_batch_import_restart(sipnet_batch *b, int32_t site, int32_t first_member,
                                int32_t count, const sipnet_restart *r, void *hip_stream);
/* export_restart: checkpoint of one member after the first n_steps_done records
 * (== restartWriteCheckpoint, restart.c:932-996; boundary = record n_steps_done-1).
 * last_rec[SIPNET_NREC] (HOST, may be NULL) is the member's full record of that last
 * step and supplies the per-step `trackers.*` values, which a resume overwrites
 * before use; prev_pools[13] (HOST, may be NULL) are the pools before that step and
 * are only needed for harvest fractions when a harvest falls on the last record.
 * meta_info: model_version "2.1.0", build_info = sipnet_version() sanitised,
 * epoch = now, processed_steps = n_steps_done (+ the count resumed from). */
int sipnet_batch_export_restart(sipnet_batch *b, int32_t site, int32_t member,
                                int32_t n_steps_done, const double *last_rec,
                                const double *prev_pools, sipnet_restart *out,
                                void *hip_stream);

/* ---- particle-filter analysis step (BASELINE config C5) -------------------
 * The reference has no particle filter: PEcAn runs one process per particle and moves
 * restart files between cycles.  Here a cycle is: sipnet_batch_run (forecast) ->
 * pf_log_weights -> [all-gather of log-weights across GPUs] -> pf_systematic_ancestors over
 * the GLOBAL particle set (computed redundantly and bit-identically on every rank) ->
 * pack_members for the columns other ranks need -> [all-to-all] -> resample.  What moves per
 * particle is its checkpoint: the carried state vector and ring (+ converted parameters
 * when particles carry their own parameters).  One site per batch.
 *
 * logw[col] = -0.5 * ((sum_t plane[t][col] - obs) / sigma)^2 (Gaussian likelihood of an
 * observed flux sum, e.g. daily NEE); -inf for members whose status is non-zero. */
int sipnet_batch_pf_log_weights(sipnet_batch *b, const void *d_plane, int32_t elem_is_f32,
                                int32_t n_steps, int64_t ld, double obs, double sigma,
                                double *d_logw, void *hip_stream);
/* Systematic resampling: w_i = llrint(exp(logw_i - max logw) * 2^30) (integer weights make
 * the prefix sum exact and the result identical on every rank), S = sum w,
 * ancestor[j] = first i with cdf[i] > min(((j + u0) * S) / n, S - 1), 0 <= u0 < 1.
 * Ancestors are non-decreasing.  d_logw[n], d_ancestors[n], optional d_fixed_weights[n]
 * (the integer weights, for checking) are DEVICE pointers; n <= 4 194 304. */
int sipnet_pf_systematic_ancestors(const double *d_logw, int64_t n, double u0,
                                   int32_t *d_ancestors, int64_t *d_fixed_weights,
                                   void *hip_stream);
/* The same without the host round trip: nothing is synchronised; the total integer weight S
 * goes to d_total (DEVICE int64, may be NULL) and the caller tests S > 0 ("a particle
 * survived") at its next natural synchronisation point.  With S = 0 the ancestors are all 0. */
int sipnet_pf_systematic_ancestors_async(const double *d_logw, int64_t n, double u0,
                                         int32_t *d_ancestors, int64_t *d_fixed_weights,
                                         int64_t *d_total, void *hip_stream);
/* Who sends what, from the GLOBAL ancestor vector d_ancestors[world * n_local] (DEVICE; identical
 * on every rank, non-decreasing): for every destination rank d the local columns it needs from
 * this rank, each once, concatenated in rank order into d_send_cols (DEVICE, capacity
 * world * n_local; send_counts[d] entries for rank d, 0 for d == rank); d_src[n_local] (DEVICE):
 * where this rank's new column j comes from -- < n_local: its own old column, n_local + k: the
 * k-th received column, received blocks concatenated in source-rank order with recv_counts[s]
 * columns each.  send_counts / recv_counts are HOST arrays of `world` entries (the split sizes
 * of the all-to-all; filling them is the plan's one host synchronisation).  world <= 64.
 * An ancestor outside [0, world * n_local) or a decreasing pair is detected on the device and answered
 * with SIPNET_ERR_BAD_ARGUMENT (no index of such a vector is ever used). */
int sipnet_pf_exchange_plan(const int32_t *d_ancestors, int64_t n_local, int32_t world, int32_t rank,
                            int32_t *d_send_cols, int32_t *d_src, int64_t *send_counts,
                            int64_t *recv_counts, void *hip_stream);
/* The resampling and the exchange plan keep device scratch per host thread between calls; this frees
 * the calling thread's (call it before the thread ends or the device is reset; nothing else does). */
void sipnet_pf_release_scratch(void);
/* DEPRECATED -- SIPNET_F64 batches only: 8-byte words per particle in a packed block, SIPNET_NSTATE +
 * SIPNET_RING_SLOTS (+ SIPNET_NPARAMS).  A SIPNET_F32_MIXED batch packs fewer (its ring travels in fp32): size
 * blocks, all-to-all splits and offsets with sipnet_batch_member_words(b, ...) of the batch that packs / resamples;
 * this one stays exported for callers built against the fp64-only layout ... */
int32_t sipnet_pf_member_words(int32_t with_params);
/* ... and of batch b: a SIPNET_F32_MIXED batch keeps its running-mean ring in fp32 (the values are NPP
 * rates, fp32 numbers there), on the device and in a packed block, where the SIPNET_RING_SLOTS rows of
 * n floats take SIPNET_RING_SLOTS / 2 rows of words -- a third less to move per resampled particle.
 * (sipnet_batch_get_ring(s) / set_rings and the restart records stay in doubles.) */
int32_t sipnet_batch_member_words(const sipnet_batch *b, int32_t with_params);
/* Pack columns d_cols[n] (DEVICE, local column indices) into d_buf laid out
 * [sipnet_batch_member_words][n]: state rows, ring rows, then parameter rows. */
int sipnet_batch_pack_members(sipnet_batch *b, const int32_t *d_cols, int64_t n,
                              int32_t with_params, double *d_buf, void *hip_stream);
/* Replace every column j by its ancestor d_src[j] (DEVICE, [ncol]): an index < ncol is one
 * of this batch's own (old) columns; ncol + k is received column k, where d_recv is the
 * concatenation of n_blocks packed blocks with block_cols[s] (HOST) columns each.
 * Gathers into spare buffers and swaps them in. */
int sipnet_batch_resample(sipnet_batch *b, const int32_t *d_src, const double *d_recv,
                          int32_t n_blocks, const int64_t *block_cols, int32_t with_params,
                          void *hip_stream);

/* The whole analysis step of a filter whose particles all live in this batch (one rank), in one call:
 * sipnet_batch_pf_log_weights -> sipnet_pf_systematic_ancestors(_async when d_total is given: no host
 * round trip, see there) -> sipnet_batch_resample with the ancestors.  d_logw, d_ancestors: DEVICE [ncol],
 * filled.  (A filter spread over ranks calls the three steps itself, with its all-gather and all-to-all
 * in between; this entry exists because the host-side cost of six calls was the analysis step's
 * duration at C5's shape.) */
int sipnet_batch_pf_analysis(sipnet_batch *b, const void *d_plane, int32_t elem_is_f32, int32_t n_steps,
                             int64_t ld, double obs, double sigma, double u0, int32_t with_params,
                             double *d_logw, int32_t *d_ancestors, int64_t *d_total, void *hip_stream);

/* ---- the filter across ranks WITHOUT an all-to-all: peer reads over xGMI -----------------------------
 * After systematic resampling the ancestors a rank needs from another rank are few (the two ends of its
 * range) and known on the device only; RCCL's send / receive sizes are host arguments, so an all-to-all
 * costs a device -> host round trip per cycle (sipnet_pf_exchange_plan), or padded slots.  On one node
 * every GPU can instead LOAD from its peers' HBM: each rank publishes its checkpoint matrices once
 * (state, ring, converted parameters, and their spares: a resampling gathers into the spare and swaps, all
 * ranks in lockstep), maps the others' -- the addresses themselves inside one process
 * (hipDeviceEnablePeerAccess; the node object below), hipIpcMemHandle mappings between processes (one
 * process per GPU under torch.distributed) -- and a cycle's exchange is then
 *     sipnet_batch_pf_local_weights  -> my block [nmax log-weights | their 256-wide block maxima]
 *     ONE all-gather of the blocks   (RCCL; the caller's, or sipnet_node_pf_analysis's)
 *     sipnet_batch_pf_resample_peers -> weights of all slots, exact prefix sum, the ancestors of MY particles,
 *                                       one gather kernel that reads every ancestor where it lives
 * with no host synchronisation and no second collective.  The all-gather is also the only ordering the
 * scheme needs: a rank's block leaves after its forecast, so whoever holds the gathered blocks may read
 * any rank's current buffers; and a rank overwrites a buffer its peers read in cycle c only after the
 * all-gather of cycle c + 1, which needs every peer's block, which that peer enqueued after its gather.
 * Ranks may hold different numbers of particles (nmax = the largest; the slots past a rank's own particles
 * weigh nothing); world <= 16; one site per batch.  Results are bit-identical to sipnet_batch_pf_analysis
 * over the concatenated particle set. */
typedef struct sipnet_pf_peer {
  int64_t process_id;        /* getpid() of the publishing process */
  int32_t device, n_particles, precision, with_params;
  int32_t ipc_valid;         /* 0: hipIpcGetMemHandle failed (peers in the same process do not need it) */
  int32_t generic_exponents; /* some particle's parameters need the general-exponent kernel variant */
  int32_t params_by_index;   /* with_params: every rank will hold ALL ranks' parameters and particles carry an index (default);
                                0 with SIPNET_KOPT_PF_MOVE_PARAMS: the rows travel with every resampled particle */
  int32_t reserved;
  uint64_t address[8];       /* state, spare state, ring, spare ring, parameters, spare parameters, parameter index, its spare */
  unsigned char ipc[8][64];  /* hipIpcMemHandle_t of the same eight allocations */
} sipnet_pf_peer;
/* Allocate the spare buffers and describe this batch's matrices; with_params: particles carry their
 * (converted) parameters.  Exchange the descriptors by any means (they are plain bytes), then ... */
int sipnet_batch_pf_publish(sipnet_batch *b, int32_t with_params, sipnet_pf_peer *out);
/* ... connect: peers[world] in rank order, peers[rank] this batch's own.  Once connected, resample only
 * through sipnet_batch_pf_resample_peers (or sipnet_batch_resample on EVERY rank in the same cycle).
 * with_params (and no SIPNET_KOPT_PF_MOVE_PARAMS): connect copies every rank's converted parameters into a bank on THIS
 * rank, [SIPNET_NPARAMS][world x nmax] doubles (0.67 GB at 8 x 131 072 particles), once; from then on a particle carries
 * its column in that bank (4 bytes) instead of its 640 bytes of rows, across ranks as inside one batch, and every
 * forecast reads its parameters from local HBM.  Parameters are constants of a particle: sipnet_batch_set_params on a
 * connected batch (or a sipnet_batch_resample that moves rows) makes the next sipnet_batch_pf_resample_peers fail with
 * SIPNET_ERR_BAD_ARGUMENT until every rank has published and connected again. */
int sipnet_batch_pf_connect(sipnet_batch *b, int32_t world, int32_t rank, const sipnet_pf_peer *peers);
/* How many filters may run their analysis on this batch's DEVICE at the same time (default 1; a node sets its shards per
 * device).  The one-launch analysis keeps every workgroup resident and spinning at barriers in device memory, so its grid
 * is sized to the device's capacity (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units) / n_filters, at most
 * 512; below 8 workgroups the analysis runs as separate launches instead.  A grid that is not co-resident after all (a
 * share that was promised and not kept) does not hang: a barrier that waits longer than ~0.3 s gives up, the launch's
 * total weight reads INT64_MIN and the next synchronising call (sipnet_batch_pf_analysis without d_total,
 * sipnet_node_pf_check) answers SIPNET_ERR_INTERNAL; later launches are not affected. */
int sipnet_batch_set_device_share(sipnet_batch *b, int32_t n_filters);
#define SIPNET_PF_VOID_TOTAL INT64_MIN   /* *d_total of an analysis whose grid barrier gave up */
/* What the last analysis of this batch did and what its exchange has moved (synchronises hip_stream for `crossing`). */
typedef struct sipnet_pf_info {
  int32_t fused;           /* 1: the one-launch kernel (pfFusedKernel), 0: separate launches */
  int32_t grid;            /* its workgroups (0 when not fused) */
  int32_t budget;          /* workgroups the device share allowed */
  int32_t world;           /* ranks of the connection (1: not connected) */
  int64_t n_slots;         /* weight slots the last cross-rank analysis went over (world x nmax) */
  int64_t cycles;          /* sipnet_batch_pf_resample_peers calls since sipnet_batch_pf_connect */
  int64_t crossing;        /* particles of THIS rank copied from ANOTHER rank's slot in those cycles: each is one checkpoint
                              (state 256 B + ring 1 000 / 2 000 B + 4 B of index, or + 640 B of rows) read over xGMI */
  int32_t params_by_index; /* 1: all ranks' parameters are replicated here, particles carry an index */
  int32_t device_share;    /* sipnet_batch_set_device_share */
} sipnet_pf_info;
int sipnet_batch_pf_info(sipnet_batch *b, sipnet_pf_info *out, void *hip_stream);
/* doubles per rank in the gathered buffer: nmax + ceil(nmax / 256) */
int64_t sipnet_batch_pf_block_len(const sipnet_batch *b);
/* sipnet_batch_pf_log_weights into this rank's block d_block[sipnet_batch_pf_block_len] (DEVICE) */
int sipnet_batch_pf_local_weights(sipnet_batch *b, const void *d_plane, int32_t elem_is_f32,
                                  int32_t n_steps, int64_t ld, double obs, double sigma,
                                  double *d_block, void *hip_stream);
/* d_gathered[world][block_len] (DEVICE): every rank's block.  d_ancestors[ncol] (DEVICE) receives, for
 * each of this rank's particles, the slot (rank * nmax + particle) it was copied from; d_total (DEVICE
 * int64, may be NULL) the total integer weight -- 0 = no particle survived, every ancestor is slot 0 --
 * for the caller's check at its next synchronisation point.  Nothing is synchronised.  A batch that was
 * never connected is a filter of one rank. */
int sipnet_batch_pf_resample_peers(sipnet_batch *b, const double *d_gathered, double u0,
                                   int32_t *d_ancestors, int64_t *d_total, void *hip_stream);

/* ---- one node, several GPUs (north star: "the ensemble axis shards across the 8 GPUs of one node
 * with a single RCCL all-gather over xGMI of the NEE/GPP/ET output block", from the C host) --------
 * A sipnet_node is the multi-GPU host object of ONE process: one sipnet_batch, one HIP stream and
 * one RCCL rank (ncclCommInitAll) per listed device; the n_members members of every site are
 * sharded contiguously (device k holds members [first_k, first_k + count_k), sizes differ by at
 * most one).  It replaces, for an ensemble, the reference's one-process-per-member host
 * (frontend.c:212-250): set_* = initModel / initEvents, setup = setupModel, run = runModelOutput's
 * loop on every device at once (statistics included: sipnet_batch_run_stats), gather_* = the
 * collective.  RCCL (librccl.so.1) is loaded on first use; without it create fails with
 * SIPNET_ERR_NO_DEVICE -- there is no fallback.  A node with ONE device is valid (and still goes
 * through RCCL).  Calls on one node must come from one thread at a time. */
typedef struct sipnet_node sipnet_node;
/* How a node cuts the batch (SURVEY 8(e) "Partitioning"): MEMBERS -- every shard holds a contiguous range of
 * every site's members and every site's forcing (one-site ensembles, the particle filter); SITES -- every
 * shard holds whole sites with all their members, so a site's forcing, events and plan exist on ONE device
 * (BASELINE config 4: 256 sites x 1 024 members = 32 sites per GPU). */
enum sipnet_shard_mode { SIPNET_SHARD_MEMBERS = 0, SIPNET_SHARD_SITES = 1 };
int sipnet_node_create(const int32_t flags[SIPNET_NFLAGS], int32_t n_sites, int32_t n_members,
                       int32_t precision, const int32_t *devices, int32_t n_devices,
                       sipnet_node **out);
/* The same with the cut chosen (sipnet_node_create = SIPNET_SHARD_MEMBERS).  Here a device MAY be listed more
 * than once: several shards then share it -- the rehearsal of an N-shard run on a smaller machine.  RCCL
 * refuses two ranks on one device, so the all-gathers of such a node are device-to-device copies ordered by
 * HIP events between the shards' streams (sipnet_node_collective_library says so); sharding, uploads,
 * kernels, peer tables and gathered layouts are the same code as with one device per shard.
 * SIPNET_SHARD_SITES: shard k owns sites [k S / N, (k + 1) S / N) (sipnet_node_site_range) with all n_members
 * members; set_climate / set_events / set_params of a site reach its owner only; ld = (most sites of a shard)
 * x n_members; sipnet_node_gather_stats's host_total lists the sites in global order. */
int sipnet_node_create_sharded(const int32_t flags[SIPNET_NFLAGS], int32_t n_sites, int32_t n_members,
                               int32_t precision, const int32_t *devices, int32_t n_devices,
                               int32_t shard_mode, sipnet_node **out);
void sipnet_node_destroy(sipnet_node *nd);
int32_t sipnet_node_n_devices(const sipnet_node *nd);
int32_t sipnet_node_shard_mode(const sipnet_node *nd);
int sipnet_node_site_range(const sipnet_node *nd, int32_t k, int32_t *first, int32_t *count);
void *sipnet_node_stream(sipnet_node *nd, int32_t k);          /* shard k's HIP stream */
sipnet_batch *sipnet_node_batch(sipnet_node *nd, int32_t k);   /* device k's batch (its own members only) */
int sipnet_node_member_range(const sipnet_node *nd, int32_t k, int32_t *first, int32_t *count);
const char *sipnet_node_collective_library(const sipnet_node *nd); /* "librccl.so.1 (RCCL 22204)" */
/* inputs, as for a batch; parameters are dealt to the devices that own the members */
int sipnet_node_set_climate(sipnet_node *nd, int32_t site, int32_t n_steps, const double *clim,
                            const int32_t *year, const int32_t *day);
int sipnet_node_set_events(sipnet_node *nd, int32_t site, int32_t n_events, const sipnet_event *events);
int sipnet_node_set_params(sipnet_node *nd, int32_t site, int32_t first_member, int32_t count,
                           const double *raw);
int sipnet_node_set_math(sipnet_node *nd, int32_t policy);
int sipnet_node_set_kernel(sipnet_node *nd, int32_t kernel, int32_t options);
int sipnet_node_setup(sipnet_node *nd);
/* Every device advances its members n_steps steps from record step0 on its own stream (one host
 * thread per device enqueues) into node-owned planes [3][n_steps][ld] and a statistics block
 * [3][n_steps][n_sites][2]; ld = sipnet_node_ld = n_sites * (largest shard, rounded up to even);
 * columns past a device's own members are zero.  Returns once everything is enqueued. */
int sipnet_node_run(sipnet_node *nd, int32_t step0, int32_t n_steps);
/* the same without the statistics (planes only: the forecast of a particle-filter cycle) */
int sipnet_node_forecast(sipnet_node *nd, int32_t step0, int32_t n_steps);
/* before a forecast whose NEE plane sipnet_node_pf_analysis(nd, 0, obs, sigma, ...) will weigh: every shard's forecast
 * launch then leaves its log-weights in its slice of the all-gather's buffer itself (sipnet_batch_pf_arm).  A hint. */
int sipnet_node_pf_arm(sipnet_node *nd, double obs, double sigma);
int sipnet_node_sync(sipnet_node *nd);
/* every member's status, status[n_sites][n_members] (HOST), after synchronising every shard's stream */
int sipnet_node_get_status(sipnet_node *nd, int32_t *status);
int64_t sipnet_node_ld(const sipnet_node *nd);
void *sipnet_node_planes(sipnet_node *nd, int32_t k);     /* DEVICE k: [3][n_steps][ld], double / float */
double *sipnet_node_stats(sipnet_node *nd, int32_t k);    /* DEVICE k: its members' sums */
/* ONE ncclAllGather (grouped over the devices): afterwards every device holds every device's
 * statistics block, sipnet_node_gathered_stats(nd, k)[n_devices][3][n_steps][n_sites][2]; host_total
 * (HOST, may be NULL) receives their sum over the devices = the whole ensemble's sum / sum of squares
 * per (variable, step, site).  0.84 MB per device and half-hourly year. */
int sipnet_node_gather_stats(sipnet_node *nd, double *host_total);
double *sipnet_node_gathered_stats(sipnet_node *nd, int32_t k);
/* The north star's exchange as written -- ONE ncclAllGather of the member-resolved output block:
 * afterwards sipnet_node_gathered_planes(nd, k) on every device is [n_devices][3][n_steps][ld].
 * (4.3 GB per device at 10 240 members x 17 520 steps: see DESIGN.md section 5 for what it costs.) */
int sipnet_node_gather_planes(sipnet_node *nd);
void *sipnet_node_gathered_planes(sipnet_node *nd, int32_t k);
/* The same exchange overlapped with the computation (SURVEY 8(e) "gather per chunk ... on a side stream"): the run
 * [step0, step0 + n_steps) is cut into n_segments launches (at whole 16-step tiles where the segments are long
 * enough) and segment j's planes travel -- ONE all-gather per segment on each shard's second stream -- under the
 * step kernel of segment j + 1.  No statistics.  Returns once everything is enqueued; sipnet_node_sync waits for
 * the gathers too.  Afterwards, on device k, sipnet_node_gathered_segment(nd, k, j, &first, &len) is segment j of
 * every shard, [n_devices][3][len][ld] (first = its first record, len = its length), and the shard's own planes
 * (sipnet_node_planes) hold the segments one after the other, [3][len_j][ld] each -- NOT one [3][n_steps][ld]
 * block: until the next sipnet_node_run / _forecast, sipnet_node_gather_stats, sipnet_node_gather_planes and
 * sipnet_node_pf_analysis return SIPNET_ERR_BAD_ARGUMENT, and a reader of sipnet_node_planes must walk the
 * segments (sipnet_node_n_segments, sipnet_node_gathered_segment's first / len).  A shard whose task fails (a stale
 * plan, a launch error) makes every shard give up BEFORE the segment's collective is enqueued: the call returns that
 * shard's error and the node stays usable. */
int sipnet_node_run_gathering(sipnet_node *nd, int32_t step0, int32_t n_steps, int32_t n_segments);
/* The member-resolved exchange in a form that fits under the kernel (the raw fp64 planes of a year are 4.3 GB per rank at
 * 10 240 members, ~100 ms of link time against 8 ms of compute): segment j's planes are REDUCED on the shard's second stream
 * while segment j + 1 computes, and the reduced block is what the all-gather moves --
 *   SIPNET_GATHER_F32   the same [3][steps][ld] as floats (an fp64 node; half the bytes)
 *   SIPNET_GATHER_SUMS  every member's sums over groups of sum_steps consecutive steps, in step order, as doubles:
 *                       [3][groups][ld] (sum_steps = 48: the daily NEE / GPP / ET of a half-hourly year, 1 / 48 of the bytes --
 *                       90 MB per rank at 10 240 members; what a consumer of the reference's per-step rows, sipnet.c:453-473,
 *                       aggregates anyway).  Groups count from step0; the last may be shorter; segments hold whole groups.
 * Afterwards sipnet_node_gathered_reduced(nd, k, j, &first_row, &n_rows, &elem_bytes) on device k is segment j of every
 * shard, [n_devices][3][n_rows][ld] (rows = steps or groups; first_row = its first), column layout as the planes'.  The
 * shards' own planes are left segment by segment as by sipnet_node_run_gathering (same restrictions until the next run) --
 * except where SIPNET_GATHER_SUMS sums inside the step kernel's launch (sipnet_node_reduced_in_kernel): no planes then. */
enum sipnet_gather_form { SIPNET_GATHER_F32 = 1, SIPNET_GATHER_SUMS = 2 };
int sipnet_node_run_gathering_reduced(sipnet_node *nd, int32_t step0, int32_t n_steps, int32_t n_segments, int32_t form,
                                      int32_t sum_steps);
void *sipnet_node_gathered_reduced(sipnet_node *nd, int32_t k, int32_t segment, int32_t *first_row, int32_t *n_rows,
                                   int32_t *elem_bytes);
/* 1: the last SIPNET_GATHER_SUMS run summed inside the step kernels' own launches (sipnet_batch_run_sums: every shard's
 * batch had such a kernel) -- the shards' planes were NOT written; 0: the planes were, and a pass on the second stream
 * summed them */
int32_t sipnet_node_reduced_in_kernel(const sipnet_node *nd);
int32_t sipnet_node_n_segments(const sipnet_node *nd);   /* of the last sipnet_node_run_gathering; 0 after a plain run */
void *sipnet_node_gathered_segment(sipnet_node *nd, int32_t k, int32_t segment, int32_t *first_step, int32_t *n_steps);
/* Column layout of a plane row: shard k's member m of its local site s sits at s * count_k + m (count_k =
 * the shard's own member count: the site stride is NOT the common maximum); columns from n_sites_k * count_k
 * up to ld are zero.  Any run length may be gathered (the planes of a run are [3][n_steps][ld] at the start
 * of the buffer). */

/* The particle filter over a node's shards (BASELINE config 5; SURVEY 8(e) "PF extra exchange"): one site,
 * SIPNET_SHARD_MEMBERS, at most 16 devices.  pf_connect publishes and maps every shard's checkpoint
 * matrices once (sipnet_batch_pf_publish / _connect).  A cycle is then
 *     sipnet_node_setup / sipnet_node_forecast(step0, n_steps)   each shard's particles forward
 *     sipnet_node_pf_analysis(variable, obs, sigma, u0)          likelihood of the observed sum of plane
 *         `variable` (0 NEE, 1 GPP, 2 ET) over the forecast -> ONE all-gather of the log-weight blocks ->
 *         every shard resamples its own particles, reading each ancestor where it lives (peer HBM)
 * with nothing synchronised: every shard's host thread only enqueues.  The total weight of each cycle is kept
 * on the devices (a ring of 64 cycles); sipnet_node_pf_check synchronises and answers SIPNET_ERR_BAD_PARAMETER
 * if a cycle since the last check ended with every particle at zero weight (its resampling then copied
 * particle 0 everywhere), SIPNET_ERR_INTERNAL if the shards disagree.  sipnet_node_pf_ancestors(nd, k):
 * DEVICE k, the slots (shard * nmax + particle, nmax = the largest shard) its particles were copied from in
 * the last cycle.  with_params: pf_connect also copies every shard's converted parameters onto every shard once (a
 * particle that crosses shards then brings a 4-byte column number: sipnet_batch_pf_connect; SIPNET_KOPT_PF_MOVE_PARAMS
 * through sipnet_node_set_kernel keeps the rows travelling).  Shards that share a device analyse at the same time: the
 * node tells each batch its share of the device's resident workgroups (sipnet_batch_set_device_share), and a cycle whose
 * analysis kernel nevertheless gave up at its barrier makes sipnet_node_pf_check answer SIPNET_ERR_INTERNAL. */
int sipnet_node_pf_connect(sipnet_node *nd, int32_t with_params);
int sipnet_node_pf_analysis(sipnet_node *nd, int32_t variable, double obs, double sigma, double u0);
int sipnet_node_pf_check(sipnet_node *nd, int32_t *n_cycles_checked);

/* ---- ranks that are PROCESSES (one per GPU; torch.distributed, MPI): a RCCL communicator of the engine's own ------------
 * The node object above is one process driving several devices.  Where every rank is a process of its own -- bench.py's
 * layout, the driver's `torch.distributed.run --nproc-per-node N` -- the hot path's one real collective, the particle
 * filter's all-gather of log-weight blocks, is enqueued on the CALLER'S stream through this communicator:
 * no hop onto a library-owned stream and back (torch.distributed's process group runs its collectives on an internal stream:
 * two cross-stream event waits, ~10 us on a 140 us cycle, profiles/r06_c5_cycle_timeline.txt).  sipnet_comm_unique_id on
 * rank 0 (ncclGetUniqueId; 128 bytes), the bytes carried to the other ranks by whatever launched them (a broadcast of the
 * launcher's), sipnet_comm_create on every rank (ncclCommInitRank on `device`; collective: all ranks call it), then
 * sipnet_comm_all_gather(c, send, recv, bytes_per_rank, stream): recv = [world][bytes_per_rank] on the device, send = this
 * rank's block -- may be its own slice of recv (in place).  The same librccl the process already holds is used (PyTorch's,
 * when loaded there).  Errors: SIPNET_ERR_NO_DEVICE with RCCL's message. */
typedef struct sipnet_comm sipnet_comm;
int sipnet_comm_unique_id(uint8_t id[128]);
int sipnet_comm_create(const uint8_t id[128], int32_t world, int32_t rank, int32_t device, sipnet_comm **out);
int sipnet_comm_all_gather(sipnet_comm *c, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *hip_stream);
int32_t sipnet_comm_world(const sipnet_comm *c);
void sipnet_comm_destroy(sipnet_comm *c);
int32_t *sipnet_node_pf_ancestors(sipnet_node *nd, int32_t k);
int64_t sipnet_node_pf_block_len(const sipnet_node *nd);

int64_t sipnet_batch_ncol(const sipnet_batch *b);
int32_t sipnet_batch_nsteps(const sipnet_batch *b);                     /* the longest site's record count */
int32_t sipnet_batch_site_nsteps(const sipnet_batch *b, int32_t site);  /* this site's (0: no forcing set yet) */
/* Site-uniform trajectory computed by the plan: gdd[t] (trackers.gdd after
 * step t) and d_till_mod[t] (eventTrackers.d_till_mod used in step t). */
int sipnet_batch_get_site_series(sipnet_batch *b, int32_t site, double *gdd,
                                 double *d_till_mod);
/* Last launch's kernel time in ms measured with HIP events on the launch stream (for bench.py's roofline
 * line); <0 if the last launch was not timed.  Launches of 512 steps and more are always timed; a shorter one
 * (a particle filter's forecast, where the two event records would cost 6 % of the cycle) only when
 * sipnet_batch_time_next_launch was called before it. */
double sipnet_batch_last_kernel_ms(sipnet_batch *b);
int sipnet_batch_time_next_launch(sipnet_batch *b);
/* Particle filter: tell the NEXT sipnet_batch_run that an analysis with this observation and sigma follows it.  When
 * that run takes the one-wave kernel's lean build (a filter's many particles, no record) it also leaves the log-weights
 * of the NEE it sums over its steps in d_logw[ncol] -- the first phase of sipnet_batch_pf_analysis, which then skips
 * its pass over the plane and one grid barrier (11 us of a 150 us cycle at 131 072 particles x 48 steps) when it is
 * called with the same plane, number of steps, observation, sigma and d_logw.  Any other launch, or an analysis with
 * other arguments, computes them as before: the call is a hint, the results are bit-identical either way. */
int sipnet_batch_pf_arm(sipnet_batch *b, double obs, double sigma, double *d_logw);
/* What the last sipnet_batch_run actually launched: the step kernel's instantiation as
 * rocprofv3 names it (e.g. "stepCoopKernel<double, true, true>"), its launch shape, and the
 * host-side costs of the last sipnet_batch_setup that rebuilt the site plans. */
typedef struct sipnet_launch_info {
  char kernel[96];
  int32_t grid, block_threads;  /* workgroups, threads per workgroup */
  int32_t waves_per_simd;       /* resident wavefronts per SIMD the register budget allows */
  int32_t lds_bytes;            /* static LDS per workgroup */
  int32_t num_cus;              /* compute units of the device */
  int32_t plan_threads;         /* host threads that built the site plans */
  double plan_build_ms;         /* host: building all site plans */
  double plan_upload_ms;        /* host -> device copy of the plans */
  int32_t plan_device_sites;    /* sites whose records the device built itself (SIPNET_KOPT_HOST_PLAN: 0) */
  int32_t reserved;
} sipnet_launch_info;
int sipnet_batch_last_launch(sipnet_batch *b, sipnet_launch_info *out);
const char *sipnet_batch_last_kernel_name(sipnet_batch *b); /* "" before the first run */

/* Test hook: the per-step records and ring evictions the device built for `site` against the host builder's, byte by
 * byte (ignore_log2: leaving out the log2(vpd) field, which the device path fills only once a member with dVpdExp != 2
 * exists).  device_info[8]: run descriptors, evictions written, status (0 ok; 3: the eviction list ran out of room), the step of a non-zero status, 10-ns ticks of
 * the ring walk and of the GDD walk, two spare. */
int sipnet_debug_plan_compare(sipnet_batch *b, int32_t site, int32_t ignore_log2, int64_t *n_records_differing,
                              int64_t *n_ops_differing, int32_t *first_step, int32_t *first_offset,
                              int32_t *device_info);
/* Test hooks.  sipnet_debug_set_num_cus: pretend the device has this many compute units (the kernel choice of
 * SIPNET_KERNEL_AUTO and the particle filter's resident-grid budget follow it: 32 = one partition of a CPX-mode MI355X).
 * sipnet_debug_pf_barrier: polls a barrier of the one-launch analysis waits before it gives up (0: the default, ~0.3 s), and
 * a workgroup of the NEXT such launch that leaves without arriving (-1: none) -- the test of the "grid not co-resident"
 * path without having to produce one. */
int sipnet_debug_set_num_cus(sipnet_batch *b, int32_t num_cus);
int sipnet_debug_pf_barrier(sipnet_batch *b, int32_t spin_budget, int32_t absent_workgroup);

/* Device buffer helpers for callers without their own allocator (the CLI). */
void *sipnet_dev_alloc(size_t bytes);
void sipnet_dev_free(void *p);
int sipnet_dev_to_host(void *host, const void *dev, size_t bytes, void *hip_stream);
/* `rows` pieces of width_bytes, dev_pitch apart on the device, host_pitch apart on the host: one column of a
 * record block [n_steps][SIPNET_NREC][ld] as a dense [n_steps][n_members] array, for the ensemble output block */
int sipnet_dev_to_host_2d(void *host, size_t host_pitch, const void *dev, size_t dev_pitch,
                          size_t width_bytes, size_t rows, void *hip_stream);
/* the same on the device (asynchronous on the stream): a strided column gathered into a dense device array first --
 * one dense copy to the host then moves it 20 x faster than 17 520 row pieces over PCIe */
int sipnet_dev_to_dev_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch,
                         size_t width_bytes, size_t rows, void *hip_stream);
int sipnet_stream_sync(void *hip_stream);
/* a HIP stream of the caller's own on `device` (non-blocking with respect to the null stream): what a host that
 * pipelines forcings over two batches gives each of them (NULL on failure) */
void *sipnet_stream_create(int32_t device);
void sipnet_stream_destroy(void *hip_stream);

/* --------------------------------------------------------- host I/O (no GPU) */
typedef struct sipnet_clim_table sipnet_clim_table;
/* Parse a `<prefix>.clim` file: 12-column or legacy 14-column format,
 * unit conversions and clamps of readClimData (sipnet.c:128-277).
 * gdd_flag = ctx.gdd. */
int sipnet_io_read_clim(const char *path, int32_t gdd_flag,
                        sipnet_clim_table **out);
int32_t sipnet_clim_nsteps(const sipnet_clim_table *t);
const double *sipnet_clim_data(const sipnet_clim_table *t);  /* [n][SIPNET_NCLIM] */
const int32_t *sipnet_clim_year(const sipnet_clim_table *t);
const int32_t *sipnet_clim_day(const sipnet_clim_table *t);
void sipnet_clim_free(sipnet_clim_table *t);

/* Parse a `<prefix>.param` file (modelParams.c:136-230, sipnet.c:290-427):
 * `name value [ignored...]`, `!` comments, case-insensitive names, unknown names
 * ignored, duplicates / missing required -> SIPNET_ERR_INPUT_FILE; divisor
 * clamps applied.  out[SIPNET_NPARAMS]; is_read[SIPNET_NPARAMS] may be NULL. */
int sipnet_io_read_params(const char *path, const int32_t flags[SIPNET_NFLAGS],
                          double *out, int32_t *is_read);
const char *sipnet_param_name(int32_t index); /* file name, "" for derived */
int32_t sipnet_param_index(const char *name); /* case-insensitive, -1 unknown */

/* Parse an `events.in` file (events.c:263-367).  A missing or empty file yields
 * zero events.  *out is malloc'ed (free with sipnet_io_free). */
int sipnet_io_read_events(const char *path, const int32_t flags[SIPNET_NFLAGS],
                          const double *params, sipnet_event **out,
                          int32_t *n_events);
void sipnet_io_free(void *p);

/* `.out` text (sipnet.c:434-473): header line and one row per step from a full
 * record.  rec points at record t of one member with element stride
 * rec_stride (1 for a packed [NREC] record, ld for a device-layout plane).
 * Both append to `buf` (size cap) and return the number of bytes written or
 * a negative value if the buffer is too small. */
int sipnet_io_format_out_header(char *buf, size_t cap);
int sipnet_io_format_out_row(char *buf, size_t cap, int32_t year, int32_t day,
                             double time, const double *rec, int64_t rec_stride);
/* Write a whole `.out` file for one member from host records
 * rec[n_steps][SIPNET_NREC]. */
int sipnet_io_write_out(const char *path, int32_t print_header, int32_t n_steps,
                        const int32_t *year, const int32_t *day,
                        const double *clim, const double *rec);
/* Write the `events.out` file of one member (events.c:369-418 format): the input events
 * with the pool deltas they caused, the computed leaf-on / leaf-off events and plant death,
 * regenerated on the host from the member's full records rec[n_steps][SIPNET_NREC], its raw
 * parameters and its pools before the first step init_pools[13] (`Envi` order). */
int sipnet_io_write_events_out(const char *path, int32_t print_header,
                               const int32_t flags[SIPNET_NFLAGS], const double *raw_params,
                               int32_t n_steps, const int32_t *year, const int32_t *day,
                               const double *clim, int32_t n_events,
                               const sipnet_event *events, const double *rec,
                               const double *init_pools);

/* Write the three `--debug-log` files of one member, <prefix>_envi.log, <prefix>_fluxes.log
 * and <prefix>_trackers.log (debug_log.c:181-312: header `year day time <names>` when
 * print_header is set (sipnet.c:1959-1961), rows
 * `%4d %3d %5.2f` then ` %.15g` per double / ` %d` per int), from host records
 * rec[n_steps][SIPNET_NREC] and dbg[n_steps][SIPNET_NDBG].  Returns 6 when a file cannot be
 * opened, 3 when the prefix is too long (debug_log.c:169-178). */
int sipnet_io_write_debug_logs(const char *prefix, int32_t print_header, int32_t n_steps,
                               const int32_t *year, const int32_t *day, const double *clim,
                               const double *rec, const double *dbg);

/* ---- ensemble output block ----------------------------------------------
 * The reference writes one `<prefix>.out` text file per process (outputHeader / outputState,
 * sipnet.c:434-473; single-variable files outputItems.c:126-150): a 10 240-member year is
 * 10 240 files and 20 GB of text that PEcAn's model2netcdf parses again.  This is the bulk path:
 * every member's outputs in ONE self-describing NetCDF-3 file, written here without a NetCDF
 * library (classic format, CDF-2 / 64-bit offsets; CDF-5 when a variable exceeds 4 GiB).
 *   dimensions   time = n_steps, member = n_members
 *   coordinates  year(time) i4, day(time) i4 [day of year], hour(time) f8 [clim column 10],
 *                length(time) f8 [days, clim column 0], member(member) i4 [member_ids or 0..M-1]
 *   data         <name>(time, member) f8 (or f4 with store_f32) + attribute `units`
 * Data variables are fixed-size and contiguous: sipnet_io_ensemble_put may fill any (step range x
 * member range) of any variable, in any order, from several threads at once (pwrite) -- device
 * shards stream their own member ranges one variable at a time; what is never `put` reads as 0.
 * attrs: global attributes as "key=value" lines (may be NULL); units[v] may be NULL (the `.out`
 * column table's units for a name it knows, else none).  store_f32: SIPNET_NC_* bits. */
enum sipnet_nc_storage {
  SIPNET_NC_F64 = 0,        /* data variables as doubles */
  SIPNET_NC_F32 = 1,        /* ... as floats (half the file; fp32-mixed batches lose nothing) */
  SIPNET_NC_FORCE_CDF5 = 2  /* the 64-bit-data format even when CDF-2 would do */
};
typedef struct sipnet_ensemble_file sipnet_ensemble_file;
int sipnet_io_ensemble_create(const char *path, int32_t n_steps, int32_t n_members,
                              const int32_t *year, const int32_t *day, const double *clim,
                              const int32_t *member_ids, int32_t n_vars,
                              const char *const *names, const char *const *units,
                              int32_t store_f32, const char *attrs, sipnet_ensemble_file **out);
/* data: HOST rows [n_steps][ld] of doubles (floats with data_is_f32), row t = step step0 + t,
 * element m = member member0 + m. */
int sipnet_io_ensemble_put(sipnet_ensemble_file *f, int32_t var, int32_t step0, int32_t n_steps,
                           int32_t member0, int32_t n_members, const void *data, int64_t ld,
                           int32_t data_is_f32);
int sipnet_io_ensemble_close(sipnet_ensemble_file *f);
/* The `.out` columns (order of outputHeader, sipnet.c:434-444) as record columns: value =
 * rec[rec0] (+ rec[rec1] when rec1 >= 0: plantWoodC is printed with the accounting delta,
 * state.c:17-19).  index = -1 for an unknown name. */
int32_t sipnet_io_out_column_count(void);
int32_t sipnet_io_out_column_index(const char *name);
int sipnet_io_out_column(int32_t k, const char **name, int32_t *rec0, int32_t *rec1,
                         const char **units);
/* One call for host-resident results: planes[3][n_steps][ld] (NEE, GPP, ET -> nee, gpp,
 * evapotranspiration) or, when rec != NULL, the columns named in the comma-separated list
 * `columns` (NULL / "" = all 32) from rec[n_steps][SIPNET_NREC][ld]. */
int sipnet_io_write_ensemble_block(const char *path, int32_t n_steps, int32_t n_members,
                                   const int32_t *year, const int32_t *day, const double *clim,
                                   const int32_t *member_ids, const double *planes,
                                   const double *rec, int64_t ld, const char *columns,
                                   int32_t store_f32, const char *attrs);

/* `SIPNET_RESTART` checkpoint text (restart.c): read follows readRestartState
 * (restart.c:590-756: magic line, `<key> <value>` lines, strict number parsing, duplicate /
 * unknown / missing keys, schema_layout sizes, complete ring arrays, lines after
 * `end_restart` ignored); write follows writeRestartState (restart.c:787-828: key order,
 * %.17g, blank line between groups).  Failures return SIPNET_ERR_RESTART (9). */
int sipnet_io_read_restart(const char *path, sipnet_restart *out);
int sipnet_io_write_restart(const char *path, const sipnet_restart *in);
/* The load-time checks of restartLoadCheckpoint (restart.c:968-996) against the run that
 * is about to resume: positive boundary length, identical model flags, model version,
 * first climate record strictly after the boundary, ring cursors in range.
 * *warnings gets SIPNET_RESTART_WARN_* bits for the conditions the reference only warns
 * about.  has_climate = 0 -> SIPNET_ERR_INPUT_FILE like the reference. */
enum sipnet_restart_warning {
  SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT = 1, /* restart.c:380-388 */
  SIPNET_RESTART_WARN_BUILD_INFO = 2,            /* restart.c:862-865 */
  SIPNET_RESTART_WARN_TIME_GAP = 4               /* restart.c:899-909 */
};
int sipnet_restart_check(const sipnet_restart *r, const int32_t flags[SIPNET_NFLAGS],
                         int32_t has_climate, int32_t year0, int32_t day0, double time0,
                         double length0, int32_t *warnings);
/* restartWriteCheckpoint's own check (restart.c:344-366): 0, SIPNET_ERR_RESTART for a
 * non-positive boundary length; *warnings gets SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT. */
int sipnet_restart_check_boundary_for_write(const sipnet_restart *r, int32_t *warnings);

#ifdef __cplusplus
}
#endif
#endif /* SIPNET_AMD_H */

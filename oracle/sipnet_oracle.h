/*
 * sipnet_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, fp64, one member at a time, re-entrant, no
 * globals) of the SIPNET per-timestep state update that the HIP kernels in
 * sipnet_amd/csrc implement.  It exists to CHECK the GPU path; it is never
 * linked into, imported by or executed from the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this
 * restatement against (a) the reference's committed smoke goldens
 * tests/smoke/{niwot,russell_1,russell_2,russell_3}/sipnet.out + events.out
 * (byte-identical through the product's formatter), (b) full-precision
 * per-step records produced by the real reference step loop
 * (oracle/_ref/libsipnet_ref.so, built by oracle/Makefile from the sources
 * under /root/reference) and committed under tests/golden/, and (c) the
 * known-answer vectors of the reference's unit tests.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/).
 */
#ifndef SIPNET_ORACLE_H
#define SIPNET_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define SIPO_NPARAMS 80
#define SIPO_NFLAGS 12
#define SIPO_NCLIM 11 /* length tair tsoil par precip vpd vpdSoil vPress wspd gdd time */
#define SIPO_NREC 36  /* per-step capture record, layout below */
#define SIPO_RING_SLOTS 250 /* MEAN_NPP_MAX_ENTRIES, sipnet/sipnet.c:39-40 */

/* flag indices (common/context.h:46-57) */
enum {
  SIPO_F_EVENTS = 0,
  SIPO_F_GDD,
  SIPO_F_GROWTH_RESP,
  SIPO_F_LEAF_WATER,
  SIPO_F_LITTER_POOL,
  SIPO_F_SNOW,
  SIPO_F_SOIL_PHENOL,
  SIPO_F_WATER_HRESP,
  SIPO_F_NITROGEN_CYCLE,
  SIPO_F_ANAEROBIC,
  SIPO_F_FLOODING,
  SIPO_F_CARBON_SATURATION
};

/* event types (sipnet/events.h:17-27 order) */
enum {
  SIPO_EV_FERT = 0,
  SIPO_EV_HARVEST,
  SIPO_EV_IRRIG,
  SIPO_EV_PLANT,
  SIPO_EV_TILL,
  SIPO_EV_LEAFON,
  SIPO_EV_LEAFOFF
};

typedef struct {
  int type, year, day, pad;
  double p[4]; /* harvest: fracRA fracRB fracTA fracTB; irrig: amount method;
                  fert: orgN orgC minN; plant: leafC woodC fineRootC coarseRootC;
                  till: effect */
} sipo_event;

/* status codes returned by sipo_run_member (mirror common/exitCodes.h) */
enum {
  SIPO_OK = 0,
  SIPO_ERR_BAD_PARAM = 3,   /* allocation params / non-positive step length */
  SIPO_ERR_INPUT_FILE = 5,  /* event without a matching climate record */
  SIPO_ERR_INTERNAL = 7     /* running-mean ring overflow */
};

/* Per-step record (identical to oracle/ref_harness.c captureRecord()):
 *  0 nee  1 gpp  2 evapotranspiration  3 totNee  4 npp  5 rAboveground
 *  6 rSoil 7 rRoot 8 ra 9 rh 10 rtot 11 woodCreation 12 soilWetnessFrac
 * 13 transpiration(flux) 14..26 envi: plantWoodC plantLeafC soilC soilWater
 *    litterC snow coarseRootC fineRootC minN soilOrgN litterN plantStorageN
 *    plantCAccountingDelta
 * 27 n2o 28 nLeaching 29 nFixation 30 nUptake 31 methane 32 meanNPP 33 gdd
 * 34 d_till_mod 35 totGpp */

typedef struct {
  long n_clamp_warn;    /* ensureNonNegative warnings (sipnet.c:1348) */
  long n_balance_warn;  /* checkBalance warnings (balance.c:146-160) */
  double max_abs_dC, max_abs_dN; /* largest raw mass-balance residual */
  int died_at_step;     /* first step with an alive->dead transition, or -1 */
} sipo_diag;

/*
 * Run one member over n_steps climate records.  Restates runModelOutput()
 * (sipnet/sipnet.c:1954-1990) without text output.
 *
 *  flags       [12]
 *  raw_params  [80] in include/sipnet_params.def order, pre-setup units
 *  clim        [n_steps][11] ALREADY converted as readClimData does
 *              (sipnet/sipnet.c:201-238)
 *  year, day   [n_steps]
 *  events      [n_events] in file order (may be NULL)
 *  rec         NULL or [n_steps][SIPO_NREC]
 *  nee,gpp,et  NULL or [n_steps]
 *  events_out  NULL or path: writes the events.out text (events.c:381-407)
 *  diag        NULL or diagnostics
 */
int sipo_run_member(const int *flags, const double *raw_params, int n_steps,
                    const double *clim, const int *year, const int *day,
                    int n_events, const sipo_event *events, double *rec,
                    double *nee, double *gpp, double *et,
                    const char *events_out, sipo_diag *diag);

/* Same run, additionally filling dbg[n_steps][SIPO_NDBG] with what the reference's
 * --debug-log prints beyond the record (debug_log.c:285-312): 0..55 the Fluxes fields in the
 * fluxes log's order, 56..62 yearly trackers, 63..66 totRtot/totRa/totRh/totNpp,
 * 67..69 didLeafGrowth / didLeafFall / isAlive, 70 trackers.lastYear, 71 phenology lastYear. */
#define SIPO_NDBG 72
int sipo_run_member_debug(const int *flags, const double *raw_params, int n_steps,
                          const double *clim, const int *year, const int *day,
                          int n_events, const sipo_event *events, double *rec,
                          double *dbg);

/* Time n_members member runs back to back (no capture); returns seconds. */
double sipo_time_members(const int *flags, const double *raw_params,
                         int n_members, int n_steps, const double *clim,
                         const int *year, const int *day, double *sink);

/* Run a contiguous block of members [m0, m1) capturing nee/gpp/et as
 * [var][step][member] planes with member stride n_members_total (the product's
 * output layout), plus final state records.  Used by parity tests. */
int sipo_run_block(const int *flags, const double *raw_params, int m0, int m1,
                   int n_members_total, int n_steps, const double *clim,
                   const int *year, const int *day, int n_events,
                   const sipo_event *events, double *nee, double *gpp,
                   double *et, double *final_rec, int *status);

/* ---- single-function probes for the reference's known-answer unit tests ---- */
/* depeffects.c:11-96 */
double sipo_clipped_water_frac(double water, double whc);
double sipo_resp_moist_effect(const int *flags, const double *params,
                              double tsoil, double water, double whc);
double sipo_temp_effect(const double *params, double tsoil);
double sipo_cn_effect(const int *flags, double kCN, double poolC, double poolN);
double sipo_anaerobic_index(const double *params, double water, double whc);
double sipo_methane_moist_effect(const double *params, double water, double whc);
double sipo_volatilization_moist_effect(const double *params, double water,
                                        double whc);
/* sipnet.c:963-1031; out[3] = fastFlow, evaporation, drainage */
void sipo_soil_water_fluxes(const int *flags, const double *params,
                            double length, double vpdSoil, double wspd,
                            double snow, double water, double netRain,
                            double snowMelt, double trans, double *out);
/* sipnet.c:656-699; out[2] = transpiration, dWater */
void sipo_moisture(const double *params, double tsoil, double potGrossPsn,
                   double vpd, double soilWater, double *out);
/* sipnet.c:517-570 */
double sipo_light_eff(const double *params, double lai, double par);
/* runmean.c:61-116 on a fresh ring (initMean 0, totWeight 5, 250 slots):
 * push n (value, weight) pairs, return mean; err receives last status */
/* stage probes on an arbitrary state (the reference's unit-test pattern); Rates order =
 * sipnet/state.h:469-645 */
int sipo_probe_events(const int *flags, const double *params, double *envi, double length,
                      int year, int day, int n_events, const sipo_event *events,
                      double *d_till_mod, double *rates_out);
int sipo_probe_events_series(const int *flags, const double *params, double *envi, int n_rec,
                             const int *year, const int *day, const double *length, int n_events,
                             const sipo_event *events, const char *out_path, int print_header);
int sipo_probe_fluxes(const int *flags, const double *params, const double *envi,
                      const double *clim, int year, int day, double mean_npp, double d_till_mod,
                      double gdd_so_far, int did_leaf_growth, int did_leaf_fall,
                      double *rates_out);
int sipo_probe_pools(const int *flags, const double *params, double *envi, const double *rates,
                     double length, int was_alive, int *alive_out);
int sipo_probe_nitrogen(const int *flags, const double *params, double *envi, double *rates,
                        double length, double tsoil, int stages);
int sipo_num_rates(void);
double sipo_ring_probe(int n, const double *values, const double *weights,
                       int *err);

#ifdef __cplusplus
}
#endif
#endif

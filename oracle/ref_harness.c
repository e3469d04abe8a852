/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * An in-process ensemble driver around the REAL reference step loop.  It is
 * compiled together with the reference sources where they lie under
 * /root/reference (see oracle/Makefile); no reference source text is copied
 * into this repository.  Like the reference's own white-box unit tests
 * (tests/sipnet/test_modeling/testFluxCalculations.c:2-3) it #includes
 * sipnet/sipnet.c so that the file-static mean-NPP tracker is reachable.
 *
 * What it adds on top of the reference:
 *   - flags are set programmatically (ctx.* fields, context.h:42-60);
 *   - the global `params` (state.h:66-408) is overwritten from a caller array
 *     before every setupModel() call, because setupModel() rescales params in
 *     place (sipnet.c:1873-1902) and is therefore not re-entrant;
 *   - per-step tracker/pool values are captured at full double precision
 *     instead of being printed at %8.3f.
 *
 * Uses: (1) generating tests/golden fixtures in the build container,
 * (2) pinning oracle/sipnet_oracle.c, (3) bench.py's cpu_baseline leg with
 * kind "reference" (the built .so travels to the GPU box, the sources do not).
 */
#include "sipnet/sipnet.c"

#include <time.h>

/* Number of doubles captured per step by ref_run_member(); layout documented
 * in oracle/sipnet_oracle.h (SIPO_NREC) -- both checkers use the same record. */
#define REF_NREC 36

static ModelParams *g_modelParams = NULL;
static int g_nSteps = 0;
static double g_baseParams[NUM_PARAMS];

int ref_num_params(void) { return (int)NUM_PARAMS; }
int ref_rec_len(void) { return REF_NREC; }
int ref_num_steps(void) { return g_nSteps; }

/* flags[12] in Context order (context.h:46-57):
 * events gdd growthResp leafWater litterPool snow soilPhenol waterHResp
 * nitrogenCycle anaerobic flooding carbonSaturation */
int ref_init(const int *flags, const char *paramFile, const char *climFile,
             const char *eventsInFile, const char *eventsOutFile) {
  initContext();
  ctx.events = flags[0];
  ctx.gdd = flags[1];
  ctx.growthResp = flags[2];
  ctx.leafWater = flags[3];
  ctx.litterPool = flags[4];
  ctx.snow = flags[5];
  ctx.soilPhenol = flags[6];
  ctx.waterHResp = flags[7];
  ctx.nitrogenCycle = flags[8];
  ctx.anaerobic = flags[9];
  ctx.flooding = flags[10];
  ctx.carbonSaturation = flags[11];
  ctx.quiet = 1;
  validateContext();

  memset(&params, 0, sizeof(params));
  initModel(&g_modelParams, paramFile, climFile);
  memcpy(g_baseParams, &params, sizeof(params));

  g_nSteps = 0;
  for (ClimateNode *c = firstClimate; c != NULL; c = c->nextClim) {
    g_nSteps++;
  }

  if (ctx.events) {
    initEvents(eventsInFile ? eventsInFile : "", eventsOutFile ? eventsOutFile
                                                               : "/dev/null",
               0);
  }
  return g_nSteps;
}

void ref_get_base_params(double *out) {
  memcpy(out, g_baseParams, sizeof(g_baseParams));
}

/* Converted climate as the reference holds it after readClimData()
 * (sipnet.c:201-238): [n_steps][11] = length tair tsoil par precip vpd vpdSoil
 * vPress wspd gdd time; plus year/day. */
void ref_get_climate(double *clim, int *year, int *day) {
  int t = 0;
  for (ClimateNode *c = firstClimate; c != NULL; c = c->nextClim, t++) {
    double *r = clim + 11 * t;
    r[0] = c->length;
    r[1] = c->tair;
    r[2] = c->tsoil;
    r[3] = c->par;
    r[4] = c->precip;
    r[5] = c->vpd;
    r[6] = c->vpdSoil;
    r[7] = c->vPress;
    r[8] = c->wspd;
    r[9] = c->gdd;
    r[10] = c->time;
    year[t] = c->year;
    day[t] = c->day;
  }
}

static void captureRecord(double *r) {
  r[0] = trackers.nee;
  r[1] = trackers.gpp;
  r[2] = trackers.evapotranspiration;
  r[3] = trackers.totNee;
  r[4] = trackers.npp;
  r[5] = trackers.rAboveground;
  r[6] = trackers.rSoil;
  r[7] = trackers.rRoot;
  r[8] = trackers.ra;
  r[9] = trackers.rh;
  r[10] = trackers.rtot;
  r[11] = trackers.woodCreation;
  r[12] = trackers.soilWetnessFrac;
  r[13] = fluxes.transpiration;
  r[14] = envi.plantWoodC;
  r[15] = envi.plantLeafC;
  r[16] = envi.soilC;
  r[17] = envi.soilWater;
  r[18] = envi.litterC;
  r[19] = envi.snow;
  r[20] = envi.coarseRootC;
  r[21] = envi.fineRootC;
  r[22] = envi.minN;
  r[23] = envi.soilOrgN;
  r[24] = envi.litterN;
  r[25] = envi.plantStorageN;
  r[26] = envi.plantCAccountingDelta;
  r[27] = trackers.n2o;
  r[28] = trackers.nLeaching;
  r[29] = trackers.nFixation;
  r[30] = trackers.nUptake;
  r[31] = trackers.methane;
  r[32] = trackers.meanNPP;
  r[33] = trackers.gdd;
  r[34] = eventTrackers.d_till_mod;
  r[35] = trackers.totGpp;
}

/* Let the reference print its own [WARNING] lines (ensureNonNegative sipnet.c:1346-1356, checkBalance
 * balance.c:149-163): tools/make_golden.py counts them per member to pin the oracle's diagnostics. */
void ref_set_quiet(int quiet) { ctx.quiet = quiet; }

/* Run one member over the whole climate file.
 *   raw_params : NUM_PARAMS doubles in `Params` struct order, PRE-setupModel units
 *   rec        : NULL or [n_steps][REF_NREC]
 *   nee/gpp/et : NULL or [n_steps] (cheap capture for baselines)
 * Mirrors runModelOutput() sipnet.c:1954-1990 without the text output. */
int ref_run_member(const double *raw_params, double *rec, double *nee,
                   double *gpp, double *et) {
  memcpy(&params, raw_params, sizeof(params));
  setupModel();
  if (ctx.events) {
    setupEvents();
  }
  int t = 0;
  while (climate != NULL) {
    updateState();
    if (rec) {
      captureRecord(rec + (size_t)REF_NREC * t);
    }
    if (nee) {
      nee[t] = trackers.nee;
    }
    if (gpp) {
      gpp[t] = trackers.gpp;
    }
    if (et) {
      et[t] = trackers.evapotranspiration;
    }
    climate = climate->nextClim;
    t++;
  }
  return t;
}

/* Time `n_members` back-to-back member runs (params given as
 * [n_members][NUM_PARAMS]); returns seconds; sink defeats dead-code removal. */
double ref_time_members(const double *raw_params, int n_members,
                        double *sink) {
  struct timespec t0, t1;
  double acc = 0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int m = 0; m < n_members; m++) {
    ref_run_member(raw_params + (size_t)m * NUM_PARAMS, NULL, NULL, NULL, NULL);
    acc += trackers.totNee;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (sink) {
    *sink = acc;
  }
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

void ref_cleanup(void) {
  cleanupModel();
  if (g_modelParams) {
    deleteModelParams(g_modelParams);
    g_modelParams = NULL;
  }
}

/*
 * sipnet_oracle.c -- TEST INFRASTRUCTURE ONLY (see sipnet_oracle.h).
 *
 * CPU restatement of the SIPNET per-timestep update.  One `Member` object
 * carries everything the reference keeps in process globals
 * (sipnet/state.c:8-15, sipnet/sipnet.c:110, sipnet/events.c:33-37,807).
 * Arithmetic keeps the reference's operation order so that results agree to
 * the last bit with an -O2, non-FMA build of the reference; this file is
 * therefore compiled with -ffp-contract=off.
 *
 * Citations are file:line under /root/reference/src/.
 */
#define _POSIX_C_SOURCE 200809L
#include "sipnet_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- parameter indices, from the shared data-format table ---- */
enum {
#define SIPNET_PARAM(idx, field, fname, rule) SP_##field = idx,
#include "../include/sipnet_params.def"
#undef SIPNET_PARAM
  SP_COUNT
};
typedef char sipo_check_nparams[(SP_COUNT == SIPO_NPARAMS) ? 1 : -1];

/* ---- constants: sipnet/sipnet.c:33-49, common/util.h:14, sipnet/balance.h:6,
 *      sipnet/events.h:56-58 ---- */
#define TINY 0.000001
#define EPS 1e-8
#define C_WEIGHT 12.0
#define TEN_9 1000000000.0
#define SEC_PER_DAY 86400.0
#define MEAN_NPP_DAYS 5
#define LAMBDA 2501000.
#define LAMBDA_S 2835000.
#define RHO 1.3
#define CP 1005.
#define GAMMA 66.
#define E_STAR_SNOW 0.6
#define TILLAGE_THRESHOLD 0.01
#define TILLAGE_DECAY_FACTOR (1 / 30.0)

/* ---- climate record view ---- */
typedef struct {
  int year, day;
  double length, tair, tsoil, par, precip, vpd, vpdSoil, vPress, wspd, gdd, time;
} Clim;

static Clim climAt(const double *clim, const int *year, const int *day, int t) {
  const double *r = clim + (size_t)SIPO_NCLIM * t;
  Clim c;
  c.year = year[t];
  c.day = day[t];
  c.length = r[0];
  c.tair = r[1];
  c.tsoil = r[2];
  c.par = r[3];
  c.precip = r[4];
  c.vpd = r[5];
  c.vpdSoil = r[6];
  c.vPress = r[7];
  c.wspd = r[8];
  c.gdd = r[9];
  c.time = r[10];
  return c;
}

/* ---- pools: sipnet/state.h:416-463 ---- */
typedef struct {
  double plantWoodC, plantLeafC, soilC, soilWater, litterC, snow, coarseRootC,
      fineRootC, minN, soilOrgN, litterN, plantStorageN, plantCAccountingDelta;
} Pools;

/* ---- per-step rates: sipnet/state.h:469-645 ---- */
typedef struct {
  double photosynthesis, leafLitter, woodLitter, rVeg, rSoil, rain,
      transpiration, drainage, litterToSoil, rLitter, snowFall, snowMelt,
      sublimation, immedEvap, fastFlow, evaporation, fineRootLoss,
      coarseRootLoss, fineRootCreation, coarseRootCreation, rCoarseRoot,
      rFineRoot, leafCreation, woodCreation, leafOnCreation,
      leafOnCreationFromWood, nVolatilization, nLeaching, nOrgSoil, nOrgLitter,
      nMin, nFixation, nUptake, leafOffNResorption, reductionNResorption,
      eventLeafC, eventWoodC, eventFineRootC, eventCoarseRootC, eventEvap,
      eventSoilWater, eventSoilC, eventLitterC, eventMinN, eventSoilOrgN,
      eventLitterN, eventInputC, eventOutputC, eventInputN, eventOutputN,
      eventLeafOnCreation, eventLeafOnCreationFromWood, eventLeafOffLitter,
      eventLeafOffNResorption, soilMethane, litterMethane;
} Rates;

/* ---- trackers: sipnet/state.h:650-726 ---- */
typedef struct {
  double gpp, rtot, ra, rh, rRoot, rSoil, rAboveground, npp, nee, woodCreation,
      gdd, evapotranspiration, soilWetnessFrac;
  double yearlyGpp, yearlyRtot, yearlyRa, yearlyRh, yearlyNpp, yearlyNee,
      yearlyLitter;
  double totGpp, totRtot, totRa, totRh, totNpp, totNee;
  int lastYear;
  double methane, n2o, nLeaching, nFixation, nUptake, meanNPP;
} Track;

/* ---- running weighted mean: sipnet/runmean.h:8-22 ---- */
typedef struct {
  double values[SIPO_RING_SLOTS], weights[SIPO_RING_SLOTS];
  int length, start, last;
  double totWeight, sum;
} Ring;

typedef struct {
  const int *flag;
  double p[SIPO_NPARAMS]; /* converted parameters */
  Pools e;
  Rates f;
  Track tr;
  int didLeafGrowth, didLeafFall, phenLastYear; /* state.h:731-747 */
  int isAlive;                                  /* state.h:750-756 */
  double d_till_mod, harvestFracRemoved, harvestFracTransferred; /* events.h:213-221 */
  Ring ring;
  /* mass-balance diagnostics: balance.h:8-30 */
  double preC, preN, postC, postN, finC, finN;
  /* events */
  int n_events, ev_next;
  const sipo_event *events;
  FILE *evout;
  sipo_diag diag;
  int status;
} Member;

#define FLAG(M, name) ((M)->flag[SIPO_F_##name])
#define P(M, name) ((M)->p[SP_##name])

static double unitClip(double x) { return fmin(fmax(x, 0.0), 1.0); } /* util.h:38 */
static double calcRatio(double num, double den) {                    /* util.c:72-75 */
  const double d = den < TINY ? TINY : den;
  return num / d;
}
static double totalWoodC(const Member *M) { /* state.c:17-19 */
  return M->e.plantWoodC + M->e.plantCAccountingDelta;
}

/* ------------------------------------------------------------------ ring */
/* runmean.c:44-52 */
static void ringReset(Ring *r, double initMean) {
  r->start = r->last = 0;
  r->values[0] = initMean;
  r->weights[0] = r->totWeight;
  r->sum = initMean * r->totWeight;
}
/* runmean.c:61-116 */
static int ringAdd(Ring *r, double value, double weight) {
  if (weight <= 0) {
    return -1;
  }
  if (weight >= r->totWeight) {
    ringReset(r, value);
    return 0;
  }
  double left = weight;
  int i = r->start;
  while (left > 0) {
    if (r->weights[i] > left) {
      r->weights[i] -= left;
      r->sum -= left * r->values[i];
      left = 0;
    } else {
      r->sum -= r->weights[i] * r->values[i];
      left -= r->weights[i];
      i = (i + 1) % r->length;
    }
  }
  r->start = i;
  i = (r->last + 1) % r->length;
  if (i == r->start) {
    r->weights[i] += weight;
    r->sum += weight * r->values[i];
    return -2;
  }
  r->last = i;
  r->values[i] = value;
  r->weights[i] = weight;
  r->sum += value * weight;
  return 0;
}
static double ringMean(const Ring *r) { return r->sum / r->totWeight; } /* runmean.c:119-121 */

/* ------------------------------------------------------- dependency effects */
/* depeffects.c:11-13 */
static double clippedWaterFrac(double water, double whc) {
  return unitClip(water / whc);
}
/* depeffects.c:15-21 */
static double anaerobicIndex(const double *p, double water, double whc) {
  double f_whc = clippedWaterFrac(water, whc);
  double f_a = p[SP_fAnoxia];
  return unitClip((f_whc - f_a) / (1 - f_a));
}
/* depeffects.c:23-62 */
static double respMoistEffect(const int *flag, const double *p, double tsoil,
                              double water, double whc) {
  if (!flag[SIPO_F_WATER_HRESP] || tsoil < 0) {
    return 1.0;
  }
  double f_whc = clippedWaterFrac(water, whc);
  if (!flag[SIPO_F_ANAEROBIC]) {
    return pow(f_whc, p[SP_soilRespMoistEffect]);
  }
  double D_aer = unitClip(f_whc / p[SP_fAnoxia]);
  double A = anaerobicIndex(p, water, whc);
  return (1 - A) * D_aer + p[SP_anaerobicDecompRate] * A;
}
/* depeffects.c:64-69 */
static double methaneMoistEffect(const double *p, double water, double whc) {
  double A = anaerobicIndex(p, water, whc);
  return pow(A, p[SP_anaerobicTransExp]);
}
/* depeffects.c:71-74 */
static double tempEffect(const double *p, double tsoil) {
  return pow(p[SP_soilRespQ10], tsoil / 10);
}
/* depeffects.c:78-87 */
static double cnEffect(const int *flag, double kCN, double poolC, double poolN) {
  if (!flag[SIPO_F_NITROGEN_CYCLE]) {
    return 1.0;
  }
  double cn = calcRatio(poolC, poolN);
  return kCN / (kCN + cn);
}
/* depeffects.c:89-96 */
static double volatilizationMoistEffect(const double *p, double water,
                                        double whc) {
  double A = anaerobicIndex(p, water, whc);
  return 0.05 + 3.8 * A * (1 - A);
}

/* --------------------------------------------------------------- nitrogen */
/* nitrogen.c:84-86 */
static double leafOnNFromC(const Member *M, double leafOnC) {
  return fmax(0.0, leafOnC / P(M, leafCN) - leafOnC / P(M, woodCN));
}
/* nitrogen.c:89-104 */
static double plantNDemandFlux(const Member *M) {
  if (!FLAG(M, NITROGEN_CYCLE)) {
    return 0.0;
  }
  const Rates *f = &M->f;
  double d = f->woodCreation / P(M, woodCN) + f->leafCreation / P(M, leafCN) +
             f->fineRootCreation / P(M, fineRootCN) +
             f->coarseRootCreation / P(M, woodCN);
  return fmax(0.0, d);
}
/* nitrogen.c:122-124 */
static double minNNonUptakeFluxes(const Member *M) {
  return M->f.nMin - M->f.nVolatilization - M->f.nLeaching;
}
/* nitrogen.c:127-134 */
static double unclaimedStorageN(const Member *M, double len) {
  double leafOnC = M->f.leafOnCreation + M->f.eventLeafOnCreation;
  double leafOnN = leafOnNFromC(M, leafOnC);
  double u = M->e.plantStorageN - leafOnN * len;
  return fmax(0.0, u);
}
/* nitrogen.c:137-152 */
static double nFixationFrac(const Member *M) {
  double inhibition;
  double denom = P(M, halfNFixationMax) + M->e.minN;
  if (denom < TINY) {
    inhibition = 1;
  } else {
    inhibition = P(M, halfNFixationMax) / denom;
  }
  return P(M, nFixationFracMax) * inhibition;
}
/* nitrogen.c:155-168 */
static void nFixationAndUptake(Member *M, double len) {
  double demand = plantNDemandFlux(M);
  double storage = unclaimedStorageN(M, len) / len;
  double rem = fmax(0.0, demand - storage);
  double frac = nFixationFrac(M);
  M->f.nFixation = frac * rem;
  M->f.nUptake = (1 - frac) * rem;
}
/* nitrogen.c:170-196 */
static void nResorptionFluxes(Member *M) {
  Rates *f = &M->f;
  if (f->woodCreation + f->leafCreation + f->fineRootCreation +
          f->coarseRootCreation <
      0.0) {
    f->reductionNResorption -=
        (f->leafCreation / P(M, leafCN) + f->woodCreation / P(M, woodCN) +
         f->coarseRootCreation / P(M, woodCN) +
         f->fineRootCreation / P(M, fineRootCN));
  }
  double nResorp = P(M, leafNResorptionFrac) * f->leafLitter / P(M, leafCN);
  f->leafOffNResorption += nResorp;
}
/* nitrogen.c:15-26 */
static void nVolatilizationFlux(Member *M, const Clim *c) {
  double d_temp = tempEffect(M->p, c->tsoil);
  double d_water = volatilizationMoistEffect(M->p, M->e.soilWater, P(M, soilWHC));
  M->f.nVolatilization = P(M, nVolatilizationFrac) * M->e.minN * d_temp * d_water;
}
/* nitrogen.c:31-41 */
static void nLeachingFlux(Member *M) {
  Rates *f = &M->f;
  double phi;
  if ((f->drainage / P(M, soilWHC)) < 1) {
    phi = f->drainage / P(M, soilWHC);
  } else {
    phi = 1;
  }
  f->nLeaching = M->e.minN * phi * P(M, nLeachingFrac);
}
/* nitrogen.c:45-82 */
static void nPoolFluxes(Member *M) {
  Rates *f = &M->f;
  const Pools *e = &M->e;
  double litterCN = calcRatio(e->litterC, e->litterN);
  double soilCN = calcRatio(e->soilC, e->soilOrgN);
  double litterMin = f->rLitter / litterCN;
  double soilMin = f->rSoil / soilCN;
  double soilNInputs = f->litterToSoil / litterCN +
                       f->fineRootLoss / P(M, fineRootCN) +
                       f->coarseRootLoss / P(M, woodCN);
  double sat = FLAG(M, CARBON_SATURATION)
                   ? unitClip(e->soilC / P(M, soilCSaturation))
                   : 0.0;
  f->nOrgLitter = f->leafLitter / P(M, leafCN) - f->leafOffNResorption +
                  f->woodLitter / P(M, woodCN) - litterMin -
                  f->litterToSoil / litterCN + (soilNInputs * sat);
  f->nOrgSoil = soilNInputs * (1 - sat) - soilMin;
  f->nMin = litterMin + soilMin;
}
/* nitrogen.c:199-207 */
static void nitrogenFluxes(Member *M, const Clim *c) {
  nResorptionFluxes(M);
  nVolatilizationFlux(M, c);
  nLeachingFlux(M);
  nPoolFluxes(M);
  nFixationAndUptake(M, c->length);
}
/* nitrogen.c:210-239 */
static void updateNitrogenPools(Member *M, double len) {
  Rates *f = &M->f;
  double demand = plantNDemandFlux(M);
  double storageDemand = demand - f->nUptake - f->nFixation;
  double leafOnN = leafOnNFromC(M, f->leafOnCreation);
  M->e.plantStorageN += (f->leafOffNResorption + f->reductionNResorption -
                         storageDemand - leafOnN) *
                        len;
  double nonUptake = minNNonUptakeFluxes(M);
  M->e.minN += (nonUptake - f->nUptake) * len;
  M->e.soilOrgN += f->nOrgSoil * len;
  M->e.litterN += f->nOrgLitter * len;
}

/* ------------------------------------------------------------- limitations */
/* limitations.c:13-64 (log lines dropped) */
static void leafOnLimitation(const Member *M, double len, double *leafOnFlux) {
  double cDemand = *leafOnFlux * len;
  if (cDemand < TINY) {
    return;
  }
  double availableC =
      (M->e.plantWoodC + M->e.coarseRootC) * P(M, leafOnReallocFrac);
  double cLimiter = availableC / cDemand;
  double nLimiter = 1.0;
  if (FLAG(M, NITROGEN_CYCLE)) {
    double nDemand = leafOnNFromC(M, cDemand);
    double availableN = M->e.plantStorageN;
    if (nDemand > TINY) {
      nLimiter = availableN / nDemand;
    }
  }
  double lim = unitClip(fmin(cLimiter, nLimiter));
  if (lim < 1) {
    *leafOnFlux *= lim;
  }
}
/* limitations.c:146-185 */
static void negativeCreationCheck(Member *M, double len) {
  Rates *f = &M->f;
  const Pools *e = &M->e;
  double turnover = e->plantLeafC * P(M, leafTurnoverRate);
  double leafDeficit = e->plantLeafC / len + f->leafCreation - turnover;
  if (leafDeficit < 0) {
    f->woodCreation += leafDeficit;
    f->leafCreation -= leafDeficit;
  }
  double fineDef = e->fineRootC / len + f->fineRootCreation - f->fineRootLoss;
  double coarseDef =
      e->coarseRootC / len + f->coarseRootCreation - f->coarseRootLoss;
  if ((fineDef < 0.0) != (coarseDef < 0.0)) {
    if (fineDef < 0.0) {
      f->coarseRootCreation += fineDef;
      f->fineRootCreation -= fineDef;
    }
    if (coarseDef < 0.0) {
      f->fineRootCreation += coarseDef;
      f->coarseRootCreation -= coarseDef;
    }
  }
}
/* limitations.c:119-129 */
static void mineralNLimitation(Member *M, double len) {
  Rates *f = &M->f;
  double pool = M->e.minN + (f->nMin + f->eventMinN) * len;
  double loss = (f->nLeaching + f->nVolatilization) * len;
  if (loss > TINY && loss > pool) {
    double red = pool / loss;
    f->nLeaching *= red;
    f->nVolatilization *= red;
  }
}
/* limitations.c:69-114 */
static void nitrogenLimitation(Member *M, double len) {
  Rates *f = &M->f;
  double uptakeDemand = f->nUptake * len;
  double nonUptakeDelta = minNNonUptakeFluxes(M) * len;
  double availableMinN = M->e.minN + nonUptakeDelta;
  if (uptakeDemand > TINY && uptakeDemand > availableMinN) {
    double unclaimed = unclaimedStorageN(M, len);
    double demand = plantNDemandFlux(M) * len;
    double uptakeFrac = 1 - nFixationFrac(M);
    double red = (availableMinN / uptakeFrac + unclaimed) / demand;
    f->woodCreation *= red;
    f->leafCreation *= red;
    f->fineRootCreation *= red;
    f->coarseRootCreation *= red;
    nFixationAndUptake(M, len);
  }
}

/* ------------------------------------------------------------------ events */
static const char *evName(int type) { /* events.c:186-208 */
  switch (type) {
    case SIPO_EV_IRRIG: return "irrig";
    case SIPO_EV_PLANT: return "plant";
    case SIPO_EV_HARVEST: return "harv";
    case SIPO_EV_FERT: return "fert";
    case SIPO_EV_TILL: return "till";
    case SIPO_EV_LEAFON: return "leafon";
    case SIPO_EV_LEAFOFF: return "leafoff";
    default: return "plantdeath";
  }
}
/* events.c:381-407: "%4d  %3d  %-7s  " then name=%-.2f pairs */
static void evWrite(Member *M, int year, int day, const char *type, int n,
                    const char **names, const double *vals) {
  if (!M->evout) {
    return;
  }
  fprintf(M->evout, "%4d  %3d  %-7s  ", year, day, type);
  for (int i = 0; i < n - 1; i++) {
    fprintf(M->evout, "%s=%-.2f,", names[i], vals[i]);
  }
  fprintf(M->evout, "%s=%-.2f\n", names[n - 1], vals[n - 1]);
}

/* events.c:449-742 */
static void processEvents(Member *M, const Clim *c) {
  const double len = c->length;
  if (len <= 0) {
    M->status = SIPO_ERR_BAD_PARAM;
    return;
  }
  M->harvestFracRemoved = 0;
  M->harvestFracTransferred = 0;
  Rates *f = &M->f;
  const Pools *e = &M->e;
  while (M->ev_next < M->n_events && M->events[M->ev_next].year <= c->year &&
         M->events[M->ev_next].day <= c->day) {
    const sipo_event *ev = &M->events[M->ev_next];
    if (ev->year < c->year || ev->day < c->day) {
      M->status = SIPO_ERR_INPUT_FILE;
      return;
    }
    switch (ev->type) {
      case SIPO_EV_IRRIG: {
        const double amount = ev->p[0];
        double soilAmount, evapAmount;
        if ((int)ev->p[1] == 0) { /* CANOPY */
          evapAmount = P(M, immedEvapFrac) * amount;
          soilAmount = amount - evapAmount;
        } else {
          evapAmount = 0.0;
          soilAmount = amount;
        }
        f->eventEvap += evapAmount / len;
        f->eventSoilWater += soilAmount / len;
        const char *nm[] = {"eventSoilWater", "eventEvap"};
        double v[] = {soilAmount, evapAmount};
        evWrite(M, ev->year, ev->day, evName(ev->type), 2, nm, v);
      } break;
      case SIPO_EV_PLANT: {
        const double leafC = ev->p[0], woodC = ev->p[1], fineC = ev->p[2],
                     coarseC = ev->p[3];
        f->eventLeafC += leafC / len;
        f->eventWoodC += woodC / len;
        f->eventFineRootC += fineC / len;
        f->eventCoarseRootC += coarseC / len;
        const double inputC = leafC + woodC + fineC + coarseC;
        double inputN = 0.0;
        f->eventInputC += inputC / len;
        if (FLAG(M, NITROGEN_CYCLE)) {
          inputN = leafC / P(M, leafCN) + woodC / P(M, woodCN) +
                   fineC / P(M, fineRootCN) + coarseC / P(M, woodCN);
          f->eventInputN += inputN / len;
        }
        const char *nm[] = {"eventLeafC",       "eventWoodC",  "eventFineRootC",
                            "eventCoarseRootC", "eventInputC", "eventInputN"};
        double v[] = {leafC, woodC, fineC, coarseC, inputC, inputN};
        evWrite(M, ev->year, ev->day, evName(ev->type), 6, nm, v);
      } break;
      case SIPO_EV_HARVEST: {
        const double fracRA = ev->p[0], fracRB = ev->p[1], fracTA = ev->p[2],
                     fracTB = ev->p[3];
        const double woodC = e->plantWoodC + e->plantCAccountingDelta;
        double above = woodC + e->plantLeafC;
        double below = e->fineRootC + e->coarseRootC;
        double total = above + below;
        if (total > TINY) {
          double removed = fracRA * above + fracRB * below;
          double moved = fracTA * above + fracTB * below;
          M->harvestFracRemoved += removed / total;
          M->harvestFracTransferred += moved / total;
        }
        double litterAdd = fracTA * (e->plantLeafC + woodC);
        double soilAdd = fracTB * (e->fineRootC + e->coarseRootC);
        const double leafDelta = -e->plantLeafC * (fracRA + fracTA);
        const double woodDelta = -woodC * (fracRA + fracTA);
        const double fineDelta = -e->fineRootC * (fracRB + fracTB);
        const double coarseDelta = -e->coarseRootC * (fracRB + fracTB);
        if (!FLAG(M, LITTER_POOL)) {
          soilAdd += litterAdd;
          litterAdd = 0.0;
        }
        f->eventLitterC += litterAdd / len;
        f->eventSoilC += soilAdd / len;
        f->eventLeafC += leafDelta / len;
        f->eventWoodC += woodDelta / len;
        f->eventFineRootC += fineDelta / len;
        f->eventCoarseRootC += coarseDelta / len;
        double litterNAdd = 0.0, soilNAdd = 0.0;
        if (FLAG(M, NITROGEN_CYCLE)) {
          const double totAbove = (e->plantLeafC / P(M, leafCN)) +
                                  (e->plantWoodC / P(M, woodCN));
          const double totBelow = (e->fineRootC / P(M, fineRootCN)) +
                                  (e->coarseRootC / P(M, woodCN));
          litterNAdd = fracTA * totAbove;
          soilNAdd = fracTB * totBelow;
          f->eventSoilOrgN += soilNAdd / len;
          f->eventLitterN += litterNAdd / len;
        }
        const double outputC = ((woodC + e->plantLeafC) * fracRA +
                                (e->fineRootC + e->coarseRootC) * fracRB);
        double outputN = 0.0;
        f->eventOutputC += outputC / len;
        if (FLAG(M, NITROGEN_CYCLE)) {
          outputN = (e->plantWoodC / P(M, woodCN) + e->plantLeafC / P(M, leafCN)) *
                        fracRA +
                    (e->fineRootC / P(M, fineRootCN) +
                     e->coarseRootC / P(M, woodCN)) *
                        fracRB;
          f->eventOutputN += outputN / len;
        }
        const char *nm[] = {"eventSoilC",     "eventLitterC",     "eventLeafC",
                            "eventWoodC",     "eventFineRootC",   "eventCoarseRootC",
                            "eventSoilOrgN",  "eventLitterN",     "eventOutputC",
                            "eventOutputN"};
        double v[] = {soilAdd,   litterAdd,   leafDelta, woodDelta,  fineDelta,
                      coarseDelta, soilNAdd,  litterNAdd, outputC,   outputN};
        evWrite(M, ev->year, ev->day, evName(ev->type), 10, nm, v);
      } break;
      case SIPO_EV_TILL: {
        M->d_till_mod += ev->p[0];
        const char *nm[] = {"eventTrackers.d_till_mod"};
        double v[] = {ev->p[0]};
        evWrite(M, ev->year, ev->day, evName(ev->type), 1, nm, v);
      } break;
      case SIPO_EV_FERT: {
        const double orgC = ev->p[1];
        double orgN = 0.0, minN = 0.0;
        if (FLAG(M, NITROGEN_CYCLE)) {
          orgN = ev->p[0];
          minN = ev->p[2];
        }
        if (FLAG(M, LITTER_POOL)) {
          f->eventLitterC += orgC / len;
        } else {
          f->eventSoilC += orgC / len;
        }
        if (FLAG(M, NITROGEN_CYCLE)) {
          f->eventLitterN += orgN / len;
          f->eventMinN += minN / len;
        }
        f->eventInputC += orgC / len;
        if (FLAG(M, NITROGEN_CYCLE)) {
          f->eventInputN += (orgN + minN) / len;
        }
        const char *nm[] = {"eventLitterC", "eventSoilC",  "eventMinN",
                            "eventLitterN", "eventInputC", "eventInputN"};
        double v[] = {FLAG(M, LITTER_POOL) ? orgC : 0.0,
                      FLAG(M, LITTER_POOL) ? 0.0 : orgC,
                      minN,
                      orgN,
                      orgC,
                      (orgN + minN)};
        evWrite(M, ev->year, ev->day, evName(ev->type), 6, nm, v);
      } break;
      case SIPO_EV_LEAFON: {
        double leafOnFlux = P(M, leafGrowth) / len;
        leafOnLimitation(M, len, &leafOnFlux);
        f->eventLeafOnCreation += leafOnFlux;
        double src = e->plantWoodC + e->coarseRootC;
        if (src > TINY) {
          f->eventLeafOnCreationFromWood += leafOnFlux * e->plantWoodC / src;
        }
      } break;
      case SIPO_EV_LEAFOFF: {
        double leafOff = e->plantLeafC * P(M, fracLeafFall);
        f->eventLeafOffLitter += leafOff / len;
        double litterNAdd = 0.0, resorb = 0.0;
        if (FLAG(M, NITROGEN_CYCLE)) {
          double leafN = leafOff / P(M, leafCN);
          resorb = leafN * P(M, leafNResorptionFrac);
          litterNAdd = leafN - resorb;
          f->eventLeafOffNResorption += resorb / len;
          f->eventLitterN += litterNAdd / len;
        }
        const char *nm[] = {"eventLeafOffLitter", "eventLeafOffNResorption",
                            "eventLitterN"};
        double v[] = {leafOff, resorb, litterNAdd};
        evWrite(M, ev->year, ev->day, evName(ev->type), 3, nm, v);
      } break;
      default:
        M->status = SIPO_ERR_INPUT_FILE;
        return;
    }
    M->ev_next++;
  }
}

/* events.c:744-790 */
static void updatePoolsForEvents(Member *M, double len) {
  const Rates *f = &M->f;
  Pools *e = &M->e;
  e->plantWoodC += f->eventWoodC * len;
  e->plantLeafC += f->eventLeafC * len;
  e->soilC += f->eventSoilC * len;
  if (FLAG(M, LITTER_POOL)) {
    e->litterC += f->eventLitterC * len;
  }
  e->plantWoodC -= f->eventLeafOnCreationFromWood * len;
  double fromRoot = f->eventLeafOnCreation - f->eventLeafOnCreationFromWood;
  e->coarseRootC -= fromRoot * len;
  e->plantLeafC += (f->eventLeafOnCreation - f->eventLeafOffLitter) * len;
  if (FLAG(M, LITTER_POOL)) {
    e->litterC += f->eventLeafOffLitter * len;
  } else {
    e->soilC += f->eventLeafOffLitter * len;
  }
  e->coarseRootC += f->eventCoarseRootC * len;
  e->fineRootC += f->eventFineRootC * len;
  e->soilWater += f->eventSoilWater * len;
  if (FLAG(M, NITROGEN_CYCLE)) {
    e->minN += f->eventMinN * len;
    e->soilOrgN += f->eventSoilOrgN * len;
    e->litterN += f->eventLitterN * len;
    double leafOnN = leafOnNFromC(M, f->eventLeafOnCreation);
    e->plantStorageN += (f->eventLeafOffNResorption - leafOnN) * len;
  }
}

/* ---------------------------------------------------------------- physics */
/* sipnet.c:517-570 */
static double lightEff(const double *p, double lai, double par) {
  enum { NUM_LAYERS = 6 };
  if (!(lai > 0 && par > 0)) {
    return 0;
  }
  double cum = 0.0, curr = 0.0;
  int coeff = 1;
  for (int layer = 0; layer <= NUM_LAYERS;) {
    double cumLai = lai * ((double)layer / NUM_LAYERS);
    double intensity = par * exp(-1.0 * p[SP_attenuation] * cumLai);
    curr = (1 - pow(2, (-1.0 * intensity / p[SP_halfSatPar])));
    cum += coeff * curr;
    layer++;
    coeff = 2 * (1 + layer % 2);
  }
  cum -= curr;
  return cum / (3.0 * NUM_LAYERS);
}
/* sipnet.c:590-641 */
static void potPsn(const double *p, double lai, double tair, double vpd,
                   double par, double *potGrossPsn, double *baseFolResp) {
  double respPerGram = p[SP_baseFolRespFrac] * p[SP_aMax];
  double grossAMax = p[SP_aMax] * p[SP_aMaxFrac] + respPerGram;
  double dTemp = (p[SP_psnTMax] - tair) * (tair - p[SP_psnTMin]) /
                 pow((p[SP_psnTMax] - p[SP_psnTMin]) / 2.0, 2);
  dTemp = fmax(dTemp, 0.0);
  double dVpd = 1.0 - p[SP_dVpdSlope] * pow(vpd, p[SP_dVpdExp]);
  dVpd = fmax(dVpd, 0.0);
  double dLight = lightEff(p, lai, par);
  double conversion = C_WEIGHT * (1.0 / TEN_9) *
                      (p[SP_leafCSpWt] / p[SP_cFracLeaf]) * lai * SEC_PER_DAY;
  *potGrossPsn = grossAMax * dTemp * dVpd * dLight * conversion;
  *baseFolResp = respPerGram * conversion;
}
/* sipnet.c:656-699 */
static void moisture(const double *p, double tsoil, double potGrossPsn,
                     double vpd, double soilWater, double *trans,
                     double *dWater) {
  if (potGrossPsn < TINY) {
    *trans = 0.0;
    *dWater = 1;
    return;
  }
  double wue = p[SP_wueConst] / vpd;
  double potTrans = potGrossPsn / wue * 1000.0 * (44.0 / 12.0) * (1.0 / 10000.0);
  double removable = fmin(soilWater, p[SP_soilWHC]) * p[SP_waterRemoveFrac];
  if (tsoil < p[SP_frozenSoilThreshold]) {
    removable *= p[SP_frozenSoilEff];
  }
  *trans = fmin(removable, potTrans);
  *dWater = *trans / potTrans;
}
/* sipnet.c:705-731 */
static int pastLeafGrowth(const Member *M, const Clim *c) {
  if (FLAG(M, GDD)) {
    double cumGdd = c->gdd;
    if (c->year == M->tr.lastYear) {
      cumGdd += M->tr.gdd;
    }
    return (cumGdd >= P(M, gddLeafOn));
  }
  if (FLAG(M, SOIL_PHENOL)) {
    return (c->tsoil >= P(M, soilTempLeafOn));
  }
  if (P(M, leafOnDay) > 0) {
    double now = (double)c->day + c->time / 24.0;
    return (now >= P(M, leafOnDay));
  }
  return 0;
}
/* sipnet.c:733-742 */
static int pastLeafFall(const Member *M, const Clim *c) {
  if (P(M, leafOffDay) > 0) {
    return ((c->day + c->time / 24.0) >= P(M, leafOffDay));
  }
  return 0;
}
/* sipnet.c:848-882 */
static void precipFluxes(Member *M, const Clim *c, double lai) {
  Rates *f = &M->f;
  if (c->tair <= 0) {
    f->snowFall = c->precip / c->length;
    f->rain = 0;
  } else {
    f->snowFall = 0;
    f->rain = c->precip / c->length;
  }
  if (FLAG(M, LEAF_WATER)) {
    double maxLeafPool = lai * P(M, leafPoolDepth);
    f->immedEvap = f->rain * P(M, immedEvapFrac);
    if (f->immedEvap > maxLeafPool) {
      f->immedEvap = maxLeafPool;
    }
  } else {
    f->immedEvap = f->rain * P(M, immedEvapFrac);
  }
}
/* sipnet.c:888-946 */
static void snowPack(Member *M, const Clim *c) {
  static const double CONVERSION = (RHO * CP) / GAMMA * (1. / LAMBDA_S) *
                                   1000. * 1000. * (1. / 10000) * SEC_PER_DAY;
  Rates *f = &M->f;
  if (M->e.snow <= 0) {
    f->snowMelt = 0;
    f->sublimation = 0;
    return;
  }
  double rd = P(M, rdConst) / c->wspd;
  f->sublimation = CONVERSION * (E_STAR_SNOW - c->vPress) / rd;
  double remaining = M->e.snow + (f->snowFall * c->length);
  if (f->sublimation < 0) {
    f->sublimation = 0;
  }
  if (remaining - (f->sublimation * c->length) < 0) {
    f->sublimation = remaining / c->length;
    remaining = 0;
  } else {
    remaining -= (f->sublimation * c->length);
  }
  if (c->tair <= 0) {
    f->snowMelt = 0;
  } else {
    f->snowMelt = P(M, snowMelt) * c->tair;
    if (remaining - (f->snowMelt * c->length) < 0) {
      f->snowMelt = remaining / c->length;
    }
  }
}
/* sipnet.c:963-1031 */
static void soilWaterFluxes(const int *flag, const double *p, double length,
                            double vpdSoil, double wspd, double snow,
                            double water, double netRain, double snowMelt,
                            double trans, double *fastFlow, double *evaporation,
                            double *drainage) {
  static const double CONVERSION = (RHO * CP) / GAMMA * (1. / LAMBDA) * 1000. *
                                   1000. * (1. / 10000) * SEC_PER_DAY;
  double netIn = netRain + snowMelt;
  *fastFlow = netIn * p[SP_fastFlowFrac];
  netIn -= *fastFlow;
  double remaining = water + netIn * length - trans * length;
  if (snow > 0) {
    *evaporation = 0;
  } else {
    double waterFrac = clippedWaterFrac(water, p[SP_soilWHC]);
    double rd = p[SP_rdConst] / wspd;
    double rsoil = exp(p[SP_rSoilConst1] - p[SP_rSoilConst2] * waterFrac);
    *evaporation = CONVERSION * vpdSoil / (rd + rsoil);
    if (*evaporation < 0) {
      *evaporation = 0;
    }
    if (remaining - (*evaporation * length) < TINY) {
      *evaporation = (remaining - TINY) / length;
      remaining = 0;
    } else {
      remaining -= (*evaporation * length);
    }
  }
  if (remaining > p[SP_soilWHC]) {
    double excess = remaining - p[SP_soilWHC];
    if (flag[SIPO_F_FLOODING]) {
      *drainage = fmin(excess * p[SP_waterDrainFrac], excess / length);
    } else {
      *drainage = excess / length;
    }
  } else {
    *drainage = 0;
  }
}

/* sipnet.c:1256-1336 */
static void calculateFluxes(Member *M, const Clim *c) {
  Rates *f = &M->f;
  Pools *e = &M->e;
  const double *p = M->p;
  double baseFolResp, potGrossPsn, dWater;
  double lai = e->plantLeafC / p[SP_leafCSpWt];

  potPsn(p, lai, c->tair, c->vpd, c->par, &potGrossPsn, &baseFolResp);
  moisture(p, c->tsoil, potGrossPsn, c->vpd, e->soilWater, &f->transpiration,
           &dWater);
  precipFluxes(M, c, lai);
  double netRain = f->rain - f->immedEvap;
  snowPack(M, c);
  soilWaterFluxes(M->flag, p, c->length, c->vpdSoil, c->wspd, e->snow,
                  e->soilWater, netRain, f->snowMelt, f->transpiration,
                  &f->fastFlow, &f->evaporation, &f->drainage);
  f->photosynthesis = potGrossPsn * dWater; /* sipnet.c:1034-1037 */

  /* vegetation respiration, sipnet.c:1051-1103,1289-1295 */
  {
    double folResp =
        baseFolResp * pow(p[SP_vegRespQ10], (c->tair - p[SP_psnTOpt]) / 10.0);
    if (c->tsoil < p[SP_frozenSoilThreshold]) {
      folResp *= p[SP_frozenSoilFolREff];
    }
    double woodResp = p[SP_baseVegResp] * totalWoodC(M) *
                      pow(p[SP_vegRespQ10], c->tair / 10.0);
    if (FLAG(M, GROWTH_RESP)) {
      double growthResp = p[SP_growthRespFrac] * ringMean(&M->ring);
      if (growthResp < 0) {
        growthResp = 0;
      }
      f->rVeg = folResp + woodResp + growthResp;
    } else {
      f->rVeg = folResp + woodResp;
    }
  }

  /* wood & leaf creation / litter, sipnet.c:756-782 */
  {
    f->woodLitter += totalWoodC(M) * p[SP_woodTurnoverRate];
    double leafLitter = e->plantLeafC * p[SP_leafTurnoverRate];
    f->leafLitter += leafLitter;
    double npp = ringMean(&M->ring);
    double leafCreation = npp * p[SP_leafAllocation];
    double woodCreation = npp * p[SP_woodAllocation];
    f->leafCreation += leafCreation;
    f->woodCreation += woodCreation;
  }

  /* phenology transitions, sipnet.c:800-842 */
  {
    if (c->year > M->phenLastYear) {
      M->didLeafGrowth = 0;
      M->didLeafFall = 0;
      M->phenLastYear = c->year;
    }
    if (!M->didLeafGrowth && pastLeafGrowth(M, c)) {
      double leafOn = p[SP_leafGrowth] / c->length;
      leafOnLimitation(M, c->length, &leafOn);
      f->leafOnCreation += leafOn;
      double src = e->plantWoodC + e->coarseRootC;
      if (src > TINY) {
        f->leafOnCreationFromWood += leafOn * e->plantWoodC / src;
      }
      M->didLeafGrowth = 1;
    }
    if (!M->didLeafFall && pastLeafFall(M, c)) {
      double len = c->length;
      double leafOff = (e->plantLeafC * p[SP_fracLeafFall]) / len;
      f->leafLitter += leafOff;
      M->didLeafFall = 1;
      if (leafOff > TINY && FLAG(M, EVENTS)) {
        const char *nm[] = {"leafLitter"};
        double v[] = {leafOff * len};
        evWrite(M, c->year, c->day, "leafoff", 1, nm, v);
      }
    }
  }

  /* litter pool, sipnet.c:1150-1171 */
  if (FLAG(M, LITTER_POOL)) {
    double te = tempEffect(p, c->tsoil);
    double me = respMoistEffect(M->flag, p, c->tsoil, e->soilWater, p[SP_soilWHC]);
    double till = 1 + M->d_till_mod; /* depeffects.c:76 */
    double cn = cnEffect(M->flag, p[SP_kCN], e->litterC, e->litterN);
    double breakdown =
        e->litterC * p[SP_litterBreakdownRate] * te * me * till * cn;
    f->rLitter = breakdown * p[SP_fracLitterRespired];
    f->litterToSoil = breakdown * (1.0 - p[SP_fracLitterRespired]);
  } else {
    f->rLitter = 0;
    f->litterToSoil = 0;
  }

  /* roots, sipnet.c:1176-1196, 1073-1077 */
  {
    f->coarseRootLoss += p[SP_coarseRootTurnoverRate] * e->coarseRootC;
    f->fineRootLoss += p[SP_fineRootTurnoverRate] * e->fineRootC;
    double npp = ringMean(&M->ring);
    double coarseCreation = p[SP_coarseRootAllocation] * npp;
    double fineCreation = p[SP_fineRootAllocation] * npp;
    f->coarseRootCreation += coarseCreation;
    f->fineRootCreation += fineCreation;
    f->rCoarseRoot = p[SP_baseCoarseRootResp] * e->coarseRootC *
                     pow(p[SP_coarseRootQ10], c->tsoil / 10.0);
    f->rFineRoot = p[SP_baseFineRootResp] * e->fineRootC *
                   pow(p[SP_fineRootQ10], c->tsoil / 10.0);
  }

  /* soil respiration, sipnet.c:1132-1148 */
  {
    double me = respMoistEffect(M->flag, p, c->tsoil, e->soilWater, p[SP_soilWHC]);
    double te = tempEffect(p, c->tsoil);
    double till = 1 + M->d_till_mod;
    double cn = cnEffect(M->flag, p[SP_kCN], e->soilC, e->soilOrgN);
    f->rSoil = e->soilC * p[SP_baseSoilResp] * me * te * till * cn;
  }

  /* methane, sipnet.c:1201-1214 */
  if (FLAG(M, ANAEROBIC)) {
    double te = tempEffect(p, c->tsoil);
    double me = methaneMoistEffect(p, e->soilWater, p[SP_soilWHC]);
    f->soilMethane = p[SP_soilMethaneRate] * e->soilC * te * me;
    if (FLAG(M, LITTER_POOL)) {
      f->litterMethane = p[SP_litterMethaneRate] * e->litterC * te * me;
    } else {
      f->litterMethane = 0.0;
    }
  }

  negativeCreationCheck(M, c->length);

  if (FLAG(M, NITROGEN_CYCLE)) {
    nitrogenFluxes(M, c);
    mineralNLimitation(M, c->length); /* limitations.c:132-139 */
    nitrogenLimitation(M, c->length);
  }

  /* delayed leaf-on event lines, sipnet.c:1230-1247 */
  if (FLAG(M, EVENTS)) {
    const double len = c->length;
    if (f->leafOnCreation > TINY) {
      const char *nm[] = {"leafOnCreation", "leafOnCreationFromWood"};
      double v[] = {f->leafOnCreation * len, f->leafOnCreationFromWood * len};
      evWrite(M, c->year, c->day, "leafon", 2, nm, v);
    }
    if (f->eventLeafOnCreation > TINY) {
      const char *nm[] = {"eventLeafOnCreation", "eventLeafOnCreationFromWood"};
      double v[] = {f->eventLeafOnCreation * len,
                    f->eventLeafOnCreationFromWood * len};
      evWrite(M, c->year, c->day, "leafon", 2, nm, v);
    }
  }
}

/* sipnet.c:1530-1536 */
static int hasSufficientBiomass(const Member *M) {
  double wood = totalWoodC(M);
  double root = M->e.fineRootC + M->e.coarseRootC;
  return M->e.plantWoodC > TINY && wood > TINY && root > TINY;
}

/* balance.c:13-30 */
static void massTotals(const Member *M, double *carbon, double *nitrogen) {
  const Pools *e = &M->e;
  *carbon = (e->plantWoodC + e->plantCAccountingDelta) + e->plantLeafC +
            e->fineRootC + e->coarseRootC + e->soilC;
  if (FLAG(M, LITTER_POOL)) {
    *carbon += e->litterC;
  }
  if (FLAG(M, NITROGEN_CYCLE)) {
    *nitrogen = e->plantWoodC / P(M, woodCN) + e->plantLeafC / P(M, leafCN) +
                e->fineRootC / P(M, fineRootCN) + e->coarseRootC / P(M, woodCN) +
                e->soilOrgN + e->litterN + e->minN + e->plantStorageN;
  } else {
    *nitrogen = 0.0;
  }
}

/* sipnet.c:1346-1356 */
static void clampStock(Member *M, double *v, double minVal) {
  if (*v < minVal) {
    if (fabs(*v) > EPS) {
      M->diag.n_clamp_warn++;
    }
    *v = 0.;
  }
}

/* sipnet.c:1769-1806 */
static void updatePoolsAndBalance(Member *M, const Clim *c, int t) {
  Rates *f = &M->f;
  Pools *e = &M->e;
  const double len = c->length;

  massTotals(M, &M->preC, &M->preN);
  updatePoolsForEvents(M, len);

  /* main pools, sipnet.c:1579-1626 */
  {
    double r_a = f->rVeg + f->rFineRoot + f->rCoarseRoot;
    double alloc = f->leafCreation + f->woodCreation + f->fineRootCreation +
                   f->coarseRootCreation;
    e->plantCAccountingDelta += ((f->photosynthesis - r_a) - alloc) * len;
    e->plantWoodC +=
        (f->woodCreation - f->woodLitter - f->leafOnCreationFromWood) * len;
    e->plantLeafC +=
        (f->leafCreation + f->leafOnCreation - f->leafLitter) * len;
    e->soilWater += (f->rain + f->snowMelt - f->immedEvap - f->fastFlow -
                     f->evaporation - f->transpiration - f->drainage) *
                    len;
    e->snow += (f->snowFall - f->snowMelt - f->sublimation) * len;
  }
  /* soil pools, sipnet.c:1634-1680 */
  {
    if (FLAG(M, LITTER_POOL)) {
      double soilInputs = f->coarseRootLoss + f->fineRootLoss + f->litterToSoil;
      double sat = FLAG(M, CARBON_SATURATION)
                       ? unitClip(e->soilC / P(M, soilCSaturation))
                       : 0.0;
      e->litterC += (f->woodLitter + f->leafLitter + (soilInputs * sat) -
                     f->litterToSoil - f->rLitter - f->litterMethane) *
                    len;
      e->soilC += (soilInputs * (1 - sat) - f->rSoil - f->soilMethane) * len;
    } else {
      e->soilC += (f->coarseRootLoss + f->fineRootLoss + f->woodLitter +
                   f->leafLitter - f->rSoil - f->soilMethane) *
                  len;
    }
    double fromRoot = f->leafOnCreation - f->leafOnCreationFromWood;
    e->coarseRootC +=
        (f->coarseRootCreation - f->coarseRootLoss - fromRoot) * len;
    e->fineRootC += (f->fineRootCreation - f->fineRootLoss) * len;
  }
  if (FLAG(M, NITROGEN_CYCLE)) {
    updateNitrogenPools(M, len);
  }
  massTotals(M, &M->postC, &M->postN);

  /* mortality, sipnet.c:1688-1767 */
  if (!M->isAlive) {
    if (hasSufficientBiomass(M)) {
      M->isAlive = 1;
    }
  } else if (!hasSufficientBiomass(M)) {
    M->isAlive = 0;
    if (M->diag.died_at_step < 0) {
      M->diag.died_at_step = t;
    }
    double wood = totalWoodC(M);
    double root = e->fineRootC + e->coarseRootC;
    e->soilC += root;
    if (FLAG(M, LITTER_POOL)) {
      e->litterC += e->plantWoodC + e->plantLeafC + e->plantCAccountingDelta;
    } else {
      e->soilC += e->plantWoodC + e->plantLeafC + e->plantCAccountingDelta;
    }
    if (FLAG(M, NITROGEN_CYCLE)) {
      e->soilOrgN +=
          e->fineRootC / P(M, fineRootCN) + e->coarseRootC / P(M, woodCN);
      e->litterN += e->plantWoodC / P(M, woodCN) +
                    e->plantLeafC / P(M, leafCN) + e->plantStorageN;
    }
    e->plantWoodC = 0.0;
    e->plantLeafC = 0.0;
    e->coarseRootC = 0.0;
    e->fineRootC = 0.0;
    e->plantCAccountingDelta = 0.0;
    if (FLAG(M, NITROGEN_CYCLE)) {
      e->plantStorageN = 0.0;
    }
    ringReset(&M->ring, 0.0);
    if (FLAG(M, EVENTS)) {
      const char *nm[] = {"harvestFracRemoved", "harvestFracTransferred",
                          "totalWoodC", "totalRootC"};
      double v[] = {M->harvestFracRemoved, M->harvestFracTransferred, wood, root};
      evWrite(M, c->year, c->day, "plantdeath", 4, nm, v);
    }
  }

  /* non-negative stocks, sipnet.c:1368-1397 */
  clampStock(M, &e->plantWoodC, 0);
  clampStock(M, &e->plantLeafC, 0);
  if (FLAG(M, LITTER_POOL)) {
    clampStock(M, &e->litterC, 0);
  }
  clampStock(M, &e->soilC, 0);
  clampStock(M, &e->coarseRootC, 0);
  clampStock(M, &e->fineRootC, 0);
  clampStock(M, &e->soilWater, 0);
  clampStock(M, &e->snow, TINY);
  clampStock(M, &e->minN, 0);
  clampStock(M, &e->soilOrgN, 0);
  clampStock(M, &e->litterN, 0);
  clampStock(M, &e->plantStorageN, 0);

  /* balance diagnostics, balance.c:40-169 (no effect on state) */
  {
    massTotals(M, &M->finC, &M->finN);
    double clampedC = M->finC - M->postC;
    if (clampedC < EPS) {
      clampedC = 0;
    }
    double clampedN = M->finN - M->postN;
    if (clampedN < EPS) {
      clampedN = 0;
    }
    double inC = f->photosynthesis + f->eventInputC;
    double outC = f->rVeg + f->rFineRoot + f->rCoarseRoot + f->rSoil +
                  f->soilMethane + f->eventOutputC;
    if (FLAG(M, LITTER_POOL)) {
      outC += f->rLitter + f->litterMethane;
    }
    inC *= len;
    outC *= len;
    double inN = 0, outN = 0;
    if (FLAG(M, NITROGEN_CYCLE)) {
      inN = f->nFixation + f->eventInputN;
      outN = f->nLeaching + f->nVolatilization + f->eventOutputN;
      inN *= len;
      outN *= len;
    }
    inC += clampedC;
    if (FLAG(M, NITROGEN_CYCLE)) {
      inN += clampedN;
    }
    double dC = (M->finC - M->preC) - (inC - outC);
    double dN = (M->finN - M->preN) + (outN - inN);
    if (fabs(dC) > M->diag.max_abs_dC) {
      M->diag.max_abs_dC = fabs(dC);
    }
    if (fabs(dN) > M->diag.max_abs_dN) {
      M->diag.max_abs_dN = fabs(dN);
    }
    if (!(fabs(dC) < EPS)) {
      M->diag.n_balance_warn++;
    }
    if (!(fabs(dN) < EPS)) {
      M->diag.n_balance_warn++;
    }
  }
}

/* sipnet.c:1420-1496 */
static void updateTrackers(Member *M, const Clim *c, double oldSoilWater) {
  Track *tr = &M->tr;
  const Rates *f = &M->f;
  const double len = c->length;
  if (c->year != tr->lastYear) {
    tr->yearlyGpp = 0.0;
    tr->yearlyRtot = 0.0;
    tr->yearlyRa = 0.0;
    tr->yearlyRh = 0.0;
    tr->yearlyNpp = 0.0;
    tr->yearlyNee = 0.0;
    tr->gdd = 0.0;
    tr->lastYear = c->year;
  }
  tr->gpp = f->photosynthesis * len;
  tr->rh = (f->rLitter + f->rSoil) * len;
  tr->rAboveground = (f->rVeg) * len;
  tr->rRoot = (f->rCoarseRoot + f->rFineRoot) * len;
  tr->rSoil = tr->rRoot + tr->rh;
  tr->ra = tr->rRoot + tr->rAboveground;
  tr->rtot = tr->ra + tr->rh;
  tr->npp = tr->gpp - tr->ra;
  tr->nee = -1.0 * (tr->npp - tr->rh);
  tr->yearlyGpp += tr->gpp;
  tr->yearlyRa += tr->ra;
  tr->yearlyRh += tr->rh;
  tr->yearlyRtot += tr->rtot;
  tr->yearlyNpp += tr->npp;
  tr->yearlyNee += tr->nee;
  tr->totGpp += tr->gpp;
  tr->totRa += tr->ra;
  tr->totRh += tr->rh;
  tr->totRtot += tr->rtot;
  tr->totNpp += tr->npp;
  tr->totNee += tr->nee;
  tr->woodCreation = f->woodCreation * len;
  tr->methane = (f->soilMethane + f->litterMethane) * len;
  tr->evapotranspiration = (f->transpiration + f->immedEvap + f->evaporation +
                            f->sublimation + f->eventEvap) *
                           len;
  tr->soilWetnessFrac =
      (oldSoilWater + M->e.soilWater) / (2.0 * P(M, soilWHC));
  tr->yearlyLitter += f->leafLitter + f->eventLeafOffLitter;
  if (FLAG(M, GDD)) {
    tr->gdd += c->gdd;
  } else {
    tr->gdd = 0.0;
  }
  tr->meanNPP = ringMean(&M->ring);
  if (FLAG(M, NITROGEN_CYCLE)) {
    tr->n2o = f->nVolatilization * len;
    tr->nLeaching = f->nLeaching * len;
    tr->nFixation = f->nFixation * len;
    tr->nUptake = f->nUptake * len;
  }
}

/* sipnet.c:1818-1855 */
static void step(Member *M, const Clim *c, int t) {
  double oldSoilWater = M->e.soilWater;
  memset(&M->f, 0, sizeof(M->f));               /* sipnet.c:1222 */
  M->isAlive = hasSufficientBiomass(M) ? 1 : 0; /* sipnet.c:1538-1544 */

  /* called unconditionally (sipnet.c:1836); with events off the list is
     empty but the step-length check and harvest-tracker reset still run */
  processEvents(M, c);
  if (M->status) {
    return;
  }
  calculateFluxes(M, c);
  updatePoolsAndBalance(M, c, t);
  updateTrackers(M, c, oldSoilWater);

  /* sipnet.c:1546-1570 */
  if (M->isAlive) {
    double npp = M->f.photosynthesis - M->f.rVeg - M->f.rCoarseRoot -
                 M->f.rFineRoot;
    int err = ringAdd(&M->ring, npp, c->length);
    if (err != 0) {
      M->status = SIPO_ERR_INTERNAL;
      return;
    }
  }
  /* events.c:811-822 */
  if (M->d_till_mod > 0) {
    M->d_till_mod *= exp(-c->length * TILLAGE_DECAY_FACTOR);
    if (M->d_till_mod < TILLAGE_THRESHOLD) {
      M->d_till_mod = 0.0;
    }
  }
}

/* sipnet.c:1858-1951 (+ :1111-1123, :1406-1413, :1501-1527) */
static int setupMember(Member *M, const int *flags, const double *raw,
                       const Clim *first) {
  memset(M, 0, sizeof(*M));
  M->flag = flags;
  memcpy(M->p, raw, sizeof(M->p));
  double *p = M->p;

  p[SP_coarseRootAllocation] =
      1 - p[SP_leafAllocation] - p[SP_woodAllocation] - p[SP_fineRootAllocation];
  if ((p[SP_leafAllocation] >= 1.0) || (p[SP_woodAllocation] >= 1.0) ||
      (p[SP_fineRootAllocation] >= 1.0) || (p[SP_coarseRootAllocation] < 0)) {
    return SIPO_ERR_BAD_PARAM;
  }
  p[SP_baseVegResp] /= 365.0;
  p[SP_litterBreakdownRate] /= 365.0;
  p[SP_baseSoilResp] /= 365.0;
  p[SP_woodTurnoverRate] /= 365.0;
  p[SP_leafTurnoverRate] /= 365.0;
  p[SP_psnTMax] = p[SP_psnTOpt] + (p[SP_psnTOpt] - p[SP_psnTMin]);

  Pools *e = &M->e;
  e->plantWoodC =
      (1 - p[SP_coarseRootFrac] - p[SP_fineRootFrac]) * p[SP_plantWoodInit];
  e->plantCAccountingDelta = 0.0;
  e->plantLeafC = p[SP_laiInit] * p[SP_leafCSpWt];
  e->litterC = FLAG(M, LITTER_POOL) ? p[SP_litterInit] : 0.0;
  e->soilC = p[SP_soilInit];

  p[SP_fineRootTurnoverRate] /= 365.0;
  p[SP_coarseRootTurnoverRate] /= 365.0;
  p[SP_baseCoarseRootResp] /= 365.0;
  p[SP_baseFineRootResp] /= 365.0;

  if (p[SP_fAnoxia] <= 0.0) {
    p[SP_fAnoxia] = TINY;
  } else if (p[SP_fAnoxia] >= 1.0) {
    p[SP_fAnoxia] = 1.0 - TINY;
  }
  if (p[SP_anaerobicDecompRate] <= 0.0) {
    p[SP_anaerobicDecompRate] = TINY;
  } else if (p[SP_anaerobicDecompRate] > 1.0) {
    p[SP_anaerobicDecompRate] = 1.0;
  }

  e->coarseRootC = p[SP_coarseRootFrac] * p[SP_plantWoodInit];
  e->fineRootC = p[SP_fineRootFrac] * p[SP_plantWoodInit];
  e->soilWater = p[SP_soilWFracInit] * p[SP_soilWHC];
  if (e->soilWater < 0) {
    e->soilWater = 0;
  }
  e->snow = p[SP_snowInit];
  if (FLAG(M, NITROGEN_CYCLE)) {
    e->minN = p[SP_minNInit];
    e->soilOrgN = p[SP_soilOrgNInit];
    e->litterN = p[SP_litterOrgNInit];
    e->plantStorageN = p[SP_plantStorageNInit];
  } else {
    e->minN = 0.0;
    e->soilOrgN = 0.0;
    e->litterN = 0.0;
    /* plantStorageN keeps its zero-initialised value (sipnet.c:1936-1940) */
  }

  /* trackers, sipnet.c:1406-1413 */
  M->tr.soilWetnessFrac = e->soilWater / p[SP_soilWHC];
  M->tr.lastYear = -1;
  /* phenology, sipnet.c:1501-1527 */
  M->didLeafGrowth = pastLeafGrowth(M, first);
  M->didLeafFall = pastLeafFall(M, first);
  if (M->didLeafFall && !M->didLeafGrowth) {
    M->didLeafGrowth = 1;
  }
  M->phenLastYear = first->year;
  M->d_till_mod = 0.0;
  /* ring: newMeanTracker(0, 5, 250) sipnet.c:2008 + reset :1948 */
  M->ring.length = SIPO_RING_SLOTS;
  M->ring.totWeight = MEAN_NPP_DAYS;
  ringReset(&M->ring, 0);
  M->diag.died_at_step = -1;
  return SIPO_OK;
}

static void capture(const Member *M, double *r) {
  const Track *tr = &M->tr;
  const Pools *e = &M->e;
  r[0] = tr->nee;
  r[1] = tr->gpp;
  r[2] = tr->evapotranspiration;
  r[3] = tr->totNee;
  r[4] = tr->npp;
  r[5] = tr->rAboveground;
  r[6] = tr->rSoil;
  r[7] = tr->rRoot;
  r[8] = tr->ra;
  r[9] = tr->rh;
  r[10] = tr->rtot;
  r[11] = tr->woodCreation;
  r[12] = tr->soilWetnessFrac;
  r[13] = M->f.transpiration;
  r[14] = e->plantWoodC;
  r[15] = e->plantLeafC;
  r[16] = e->soilC;
  r[17] = e->soilWater;
  r[18] = e->litterC;
  r[19] = e->snow;
  r[20] = e->coarseRootC;
  r[21] = e->fineRootC;
  r[22] = e->minN;
  r[23] = e->soilOrgN;
  r[24] = e->litterN;
  r[25] = e->plantStorageN;
  r[26] = e->plantCAccountingDelta;
  r[27] = tr->n2o;
  r[28] = tr->nLeaching;
  r[29] = tr->nFixation;
  r[30] = tr->nUptake;
  r[31] = tr->methane;
  r[32] = tr->meanNPP;
  r[33] = tr->gdd;
  r[34] = M->d_till_mod;
  r[35] = tr->totGpp;
}

static int runMember(Member *M, const int *flags, const double *raw_params,
                     int n_steps, const double *clim, const int *year,
                     const int *day, int n_events, const sipo_event *events,
                     double *rec, double *nee, double *gpp, double *et,
                     size_t out_stride, FILE *evout, double *dbg) {
  if (n_steps <= 0) {
    return SIPO_OK;
  }
  Clim c0 = climAt(clim, year, day, 0);
  int st = setupMember(M, flags, raw_params, &c0);
  if (st) {
    return st;
  }
  M->n_events = flags[SIPO_F_EVENTS] ? n_events : 0;
  M->events = events;
  M->ev_next = 0;
  M->evout = evout;
  for (int t = 0; t < n_steps; t++) {
    Clim c = climAt(clim, year, day, t);
    step(M, &c, t);
    if (M->status) {
      return M->status;
    }
    if (rec) {
      capture(M, rec + (size_t)SIPO_NREC * t);
    }
    if (dbg) { /* what outputDebugState() prints beyond the record: debug_log.c:285-312 */
      double *g = dbg + (size_t)SIPO_NDBG * t;
      memcpy(g, &M->f, 56 * sizeof(double)); /* Rates is declared in the log's field order */
      g[56] = M->tr.yearlyGpp;
      g[57] = M->tr.yearlyRtot;
      g[58] = M->tr.yearlyRa;
      g[59] = M->tr.yearlyRh;
      g[60] = M->tr.yearlyNpp;
      g[61] = M->tr.yearlyNee;
      g[62] = M->tr.yearlyLitter;
      g[63] = M->tr.totRtot;
      g[64] = M->tr.totRa;
      g[65] = M->tr.totRh;
      g[66] = M->tr.totNpp;
      g[67] = M->didLeafGrowth;
      g[68] = M->didLeafFall;
      g[69] = M->isAlive;
      g[70] = M->tr.lastYear;
      g[71] = M->phenLastYear;
    }
    if (nee) {
      nee[out_stride * t] = M->tr.nee;
    }
    if (gpp) {
      gpp[out_stride * t] = M->tr.gpp;
    }
    if (et) {
      et[out_stride * t] = M->tr.evapotranspiration;
    }
  }
  return SIPO_OK;
}

int sipo_run_member(const int *flags, const double *raw_params, int n_steps,
                    const double *clim, const int *year, const int *day,
                    int n_events, const sipo_event *events, double *rec,
                    double *nee, double *gpp, double *et,
                    const char *events_out, sipo_diag *diag) {
  Member *M = (Member *)malloc(sizeof(Member));
  FILE *evout = NULL;
  if (events_out && flags[SIPO_F_EVENTS]) {
    evout = fopen(events_out, "w");
  }
  int st = runMember(M, flags, raw_params, n_steps, clim, year, day, n_events,
                     events, rec, nee, gpp, et, 1, evout, NULL);
  if (evout) {
    fclose(evout);
  }
  if (diag) {
    *diag = M->diag;
  }
  free(M);
  return st;
}

int sipo_run_member_debug(const int *flags, const double *raw_params, int n_steps,
                          const double *clim, const int *year, const int *day,
                          int n_events, const sipo_event *events, double *rec,
                          double *dbg) {
  Member *M = (Member *)malloc(sizeof(Member));
  int st = runMember(M, flags, raw_params, n_steps, clim, year, day, n_events,
                     events, rec, NULL, NULL, NULL, 1, NULL, dbg);
  free(M);
  return st;
}

int sipo_run_block(const int *flags, const double *raw_params, int m0, int m1,
                   int n_members_total, int n_steps, const double *clim,
                   const int *year, const int *day, int n_events,
                   const sipo_event *events, double *nee, double *gpp,
                   double *et, double *final_rec, int *status) {
  Member *M = (Member *)malloc(sizeof(Member));
  int worst = 0;
  for (int m = m0; m < m1; m++) {
    int st = runMember(M, flags, raw_params + (size_t)SIPO_NPARAMS * m, n_steps,
                       clim, year, day, n_events, events, NULL,
                       nee ? nee + m : NULL, gpp ? gpp + m : NULL,
                       et ? et + m : NULL, (size_t)n_members_total, NULL, NULL);
    if (status) {
      status[m] = st;
    }
    if (st > worst) {
      worst = st;
    }
    if (final_rec) {
      capture(M, final_rec + (size_t)SIPO_NREC * m);
    }
  }
  free(M);
  return worst;
}

double sipo_time_members(const int *flags, const double *raw_params,
                         int n_members, int n_steps, const double *clim,
                         const int *year, const int *day, double *sink) {
  Member *M = (Member *)malloc(sizeof(Member));
  struct timespec t0, t1;
  double acc = 0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int m = 0; m < n_members; m++) {
    runMember(M, flags, raw_params + (size_t)SIPO_NPARAMS * m, n_steps, clim,
              year, day, 0, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL);
    acc += M->tr.totNee;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (sink) {
    *sink = acc;
  }
  free(M);
  return (double)(t1.tv_sec - t0.tv_sec) +
         1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* --------------------------------------------------------------- probes */
double sipo_clipped_water_frac(double water, double whc) {
  return clippedWaterFrac(water, whc);
}
double sipo_resp_moist_effect(const int *flags, const double *params,
                              double tsoil, double water, double whc) {
  return respMoistEffect(flags, params, tsoil, water, whc);
}
double sipo_temp_effect(const double *params, double tsoil) {
  return tempEffect(params, tsoil);
}
double sipo_cn_effect(const int *flags, double kCN, double poolC, double poolN) {
  return cnEffect(flags, kCN, poolC, poolN);
}
double sipo_anaerobic_index(const double *params, double water, double whc) {
  return anaerobicIndex(params, water, whc);
}
double sipo_methane_moist_effect(const double *params, double water,
                                 double whc) {
  return methaneMoistEffect(params, water, whc);
}
double sipo_volatilization_moist_effect(const double *params, double water,
                                        double whc) {
  return volatilizationMoistEffect(params, water, whc);
}
void sipo_soil_water_fluxes(const int *flags, const double *params,
                            double length, double vpdSoil, double wspd,
                            double snow, double water, double netRain,
                            double snowMelt, double trans, double *out) {
  soilWaterFluxes(flags, params, length, vpdSoil, wspd, snow, water, netRain,
                  snowMelt, trans, &out[0], &out[1], &out[2]);
}
void sipo_moisture(const double *params, double tsoil, double potGrossPsn,
                   double vpd, double soilWater, double *out) {
  moisture(params, tsoil, potGrossPsn, vpd, soilWater, &out[0], &out[1]);
}
double sipo_light_eff(const double *params, double lai, double par) {
  return lightEff(params, lai, par);
}
double sipo_ring_probe(int n, const double *values, const double *weights,
                       int *err) {
  Ring r;
  r.length = SIPO_RING_SLOTS;
  r.totWeight = MEAN_NPP_DAYS;
  ringReset(&r, 0);
  int e = 0;
  for (int i = 0; i < n; i++) {
    e = ringAdd(&r, values[i], weights[i]);
  }
  if (err) {
    *err = e;
  }
  return ringMean(&r);
}

/* ---- probes on an arbitrary state: the pattern of the reference's own unit tests, which set
 * the globals `envi` / `params` / `ctx` / `climate` directly and call one stage of the step
 * (tests/sipnet/test_events_types/ *.c -> procEvents(); test_modeling/testMethane.c,
 * testCarbonSaturation.c, testFluxCalculations.c -> calculateFluxes() pieces). ---------------- */
static void probeMember(Member *M, const int *flags, const double *params, const double *envi) {
  memset(M, 0, sizeof(*M));
  M->flag = flags;
  memcpy(M->p, params, sizeof(M->p));
  memcpy(&M->e, envi, sizeof(M->e));
  ringReset(&M->ring, 0.0);
  M->ring.length = SIPO_RING_SLOTS;
  M->ring.totWeight = MEAN_NPP_DAYS;
  M->isAlive = hasSufficientBiomass(M) ? 1 : 0;
}

/* processEvents() + updatePoolsForEvents() (events.c:449-790): envi[13] in/out, d_till_mod in/out;
 * rates_out (may be NULL) receives the Rates struct as doubles. */
int sipo_probe_events(const int *flags, const double *params, double *envi, double length,
                      int year, int day, int n_events, const sipo_event *events,
                      double *d_till_mod, double *rates_out) {
  Member M;
  probeMember(&M, flags, params, envi);
  M.d_till_mod = d_till_mod ? *d_till_mod : 0.0;
  M.n_events = n_events;
  M.events = events;
  Clim c;
  memset(&c, 0, sizeof(c));
  c.year = year;
  c.day = day;
  c.length = length;
  processEvents(&M, &c);
  if (M.status) return M.status;
  updatePoolsForEvents(&M, length);
  memcpy(envi, &M.e, sizeof(M.e));
  if (d_till_mod) *d_till_mod = M.d_till_mod;
  if (rates_out) memcpy(rates_out, &M.f, sizeof(M.f));
  return 0;
}

/* The same over a series of records, pools carried from one to the next, with the events.out
 * text the reference writes while processing (events.c:369-418): the pattern of
 * test_events_infrastructure/testEventOutputFile.c. */
int sipo_probe_events_series(const int *flags, const double *params, double *envi, int n_rec,
                             const int *year, const int *day, const double *length, int n_events,
                             const sipo_event *events, const char *out_path, int print_header) {
  Member M;
  probeMember(&M, flags, params, envi);
  M.n_events = n_events;
  M.events = events;
  FILE *out = fopen(out_path, "w");
  if (!out) return SIPO_ERR_INPUT_FILE;
  if (print_header) {
    fprintf(out, "%4s  %3s  %-7s  %s", "year", "day", "type",
            "param_name=delta[,param_name=delta,...]\n");
  }
  M.evout = out;
  for (int i = 0; i < n_rec && !M.status; i++) {
    Clim c;
    memset(&c, 0, sizeof(c));
    c.year = year[i];
    c.day = day[i];
    c.length = length[i];
    memset(&M.f, 0, sizeof(M.f));
    processEvents(&M, &c);
    if (!M.status) updatePoolsForEvents(&M, c.length);
  }
  fclose(out);
  memcpy(envi, &M.e, sizeof(M.e));
  return M.status;
}

/* calculateFluxes() (sipnet.c:1256-1336) on the given pools and climate record clim[11] (layout
 * of sipo_run_member); mean_npp seeds the running mean, gdd_so_far / last_year the trackers used
 * by the phenology tests.  rates_out receives the Rates struct as doubles. */
int sipo_probe_fluxes(const int *flags, const double *params, const double *envi,
                      const double *clim, int year, int day, double mean_npp, double d_till_mod,
                      double gdd_so_far, int did_leaf_growth, int did_leaf_fall,
                      double *rates_out) {
  Member M;
  probeMember(&M, flags, params, envi);
  ringReset(&M.ring, mean_npp);
  M.d_till_mod = d_till_mod;
  M.tr.gdd = gdd_so_far;
  M.tr.lastYear = year;
  M.phenLastYear = year;
  M.didLeafGrowth = did_leaf_growth;
  M.didLeafFall = did_leaf_fall;
  Clim c = climAt(clim, &year, &day, 0);
  calculateFluxes(&M, &c);
  if (M.status) return M.status;
  memcpy(rates_out, &M.f, sizeof(M.f));
  return 0;
}

/* updatePoolsAndBalance() (sipnet.c:1769-1806: event, main, soil and N pool updates, mortality,
 * non-negativity clamps) with the given per-step rates: envi[13] in/out. */
int sipo_probe_pools(const int *flags, const double *params, double *envi, const double *rates,
                     double length, int was_alive, int *alive_out) {
  Member M;
  probeMember(&M, flags, params, envi);
  if (was_alive >= 0) M.isAlive = was_alive; /* the tracker of the step before, state.h:750-756 */
  memcpy(&M.f, rates, sizeof(M.f));
  M.diag.died_at_step = -1;
  Clim c;
  memset(&c, 0, sizeof(c));
  c.year = 2024;
  c.day = 70;
  c.length = length;
  updatePoolsAndBalance(&M, &c, 0);
  if (M.status) return M.status;
  memcpy(envi, &M.e, sizeof(M.e));
  if (alive_out) *alive_out = M.isAlive;
  return 0;
}

/* The stages of the nitrogen cycle one by one, on prescribed rates (the pattern of
 * test_modeling/testNitrogenCycle.c): stage bits 1 resorption (nitrogen.c:170-196), 2 volatilisation
 * (:15-26), 4 leaching (:31-41), 8 organic-pool fluxes (:45-82), 16 fixation + uptake (:155-168),
 * 32 mineral-N limitation (limitations.c:119-129), 64 nitrogen limitation (limitations.c:69-114),
 * 128 updateNitrogenPools (nitrogen.c:210-239).  envi[13] and rates[] in/out. */
int sipo_probe_nitrogen(const int *flags, const double *params, double *envi, double *rates,
                        double length, double tsoil, int stages) {
  Member M;
  probeMember(&M, flags, params, envi);
  memcpy(&M.f, rates, sizeof(M.f));
  Clim c;
  memset(&c, 0, sizeof(c));
  c.year = 2024;
  c.day = 70;
  c.length = length;
  c.tsoil = tsoil;
  if (stages & 1) nResorptionFluxes(&M);
  if (stages & 2) nVolatilizationFlux(&M, &c);
  if (stages & 4) nLeachingFlux(&M);
  if (stages & 8) nPoolFluxes(&M);
  if (stages & 16) nFixationAndUptake(&M, length);
  if (stages & 32) mineralNLimitation(&M, length);
  if (stages & 64) nitrogenLimitation(&M, length);
  if (stages & 128) updateNitrogenPools(&M, length);
  memcpy(envi, &M.e, sizeof(M.e));
  memcpy(rates, &M.f, sizeof(M.f));
  return 0;
}

int sipo_num_rates(void) { return (int)(sizeof(Rates) / sizeof(double)); }

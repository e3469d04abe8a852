// plan.cpp -- host-side construction of the per-site plan (see plan.h).
#include "plan.h"

#include <cstring>

#include <cmath>
#include <cstdio>

namespace sipnet {

namespace {
constexpr double kTiny = 0.000001;            // common/util.h:14
constexpr double kMeanNppDays = 5.0;          // sipnet.c:39
constexpr double kTillThreshold = 0.01;       // events.h:56
constexpr double kTillDecay = 1 / 30.0;       // events.h:58
constexpr double kEStarSnow = 0.6;             // sipnet.c:890-891
}  // namespace

void RingSched::reset(int32_t step) {
  start = last = 0;
  w[0] = kMeanNppDays;
  insStep[0] = step;
}

int32_t RingSched::advance(int32_t step, double weight, std::vector<RingOp>* ops,
                           bool* overflow) {
  if (!(weight > 0)) {
    return 0;  // unreachable in a valid run (the plan carries a site-fatal status)
  }
  if (weight >= kMeanNppDays) {  // runmean.c:67-69
    reset(step);
    return -1;
  }
  double left = weight;
  int i = start;
  while (left > 0) {  // runmean.c:76-86
    RingOp op;
    op.slot = i;
    op.insStep = insStep[i];
    if (w[i] > left) {
      w[i] -= left;
      op.w = left;
      left = 0;
    } else {
      op.w = w[i];
      left -= w[i];
      i = (i + 1) % SIPNET_RING_SLOTS;
    }
    if (ops) ops->push_back(op);
  }
  start = i;
  i = (last + 1) % SIPNET_RING_SLOTS;
  if (i == start) {  // runmean.c:93-95
    if (overflow) *overflow = true;
    return i;
  }
  last = i;
  w[i] = weight;
  insStep[i] = step;
  return i;
}

namespace {
// the fast record of one step from its StepRec and its ring evictions (slots of the NEXT step and
// the tile summary are patched in by the caller once they are known)
// phenMode: what the leaf-on test compares with its threshold (sipnet.c:705-731) -- 0 the year-to-date GDD,
// 1 the soil temperature (soil-phenology flag), 2 the day of year (both flags off)
// moistHResp: the water_hresp flag; off = heterotrophic respiration never sees the soil moisture (depeffects.c:23-57),
// which is what the kernels do on a step with frozen soil: the same bit says both
void fillFastRec(const StepRec& s, const RingOp* ops, bool tsoilSame, int phenMode, bool moistHResp, FastRec& f, int& slot0,
                 int& slot1) {
  f = FastRec{};
  f.len = s.length;
  f.invLen = s.invLen;
  f.tair = s.tair;
  f.tsoil = s.tsoil;
  f.negPar = -s.par;
  f.vpd = s.vpd;
  f.rainRate = s.rainRate;
  f.sublW = s.sublNum * s.wspd;
  f.evapNum = s.evapNum;
  f.invWspd = s.invWspd;
  f.tair10 = s.tair10;
  f.tsoil10 = s.tsoil10;
  f.cumGdd = phenMode == 0 ? s.cumGdd : phenMode == 1 ? s.tsoil : s.dayTime;
  f.dayTime = s.dayTime;
  f.log2vpd = s.log2vpd;
  f.tillP1 = 1.0 + s.dTill;
  f.gddAfter = s.gddAfter;
  f.tillAfter = s.tillAfter;
  // inline evictions; a missing one is a no-op on a valid slot (w = 0)
  const int safeSlot = s.ringInsSlot >= 0 ? s.ringInsSlot : 0;
  slot0 = s.ringOpCount > 0 ? ops[0].slot : safeSlot;
  f.ins0 = s.ringOpCount > 0 ? ops[0].insStep : -1;
  f.w0 = s.ringOpCount > 0 ? ops[0].w : 0.0;
  slot1 = s.ringOpCount > 1 ? ops[1].slot : slot0;
  f.ins1 = s.ringOpCount > 1 ? ops[1].insStep : f.ins0;
  f.w1 = s.ringOpCount > 1 ? ops[1].w : 0.0;
  const int bits = (s.bits & STEP_PHEN_NEW_YEAR ? FAST_PHEN_NEW_YEAR : 0) |
                   (s.bits & STEP_TRACK_NEW_YEAR ? FAST_TRACK_NEW_YEAR : 0) |
                   (s.tair > 0 ? FAST_TAIR_POS : 0) | (s.par > 0 ? FAST_PAR_POS : 0) |
                   ((s.tsoil < 0 || !moistHResp) ? FAST_TSOIL_NEG : 0) | (f.w1 != 0.0 ? FAST_HAS_W1 : 0) |
                   (s.dTill != 0.0 ? FAST_HAS_TILL : 0) | (tsoilSame ? FAST_TSOIL_SAME : 0) |
                   (s.ringOpCount == 1 && s.ringInsSlot >= 0 ? FAST_RING_REGULAR : 0);
  f.bitsOps = bits | (s.ringOpCount << 16);
  f.insSlot = s.ringInsSlot;
  f.evCount = s.evCount;
  f.opFirst = s.ringOpFirst;
  f.evFirst = s.evFirst;
  f.year = s.year;
  f.day = s.day;
}

// the narrow fields of an fp32-mixed batch's records (plan.h): float in the low word, quiet-NaN tag in the high one
double narrowSlot(double v) {
  const float f = (float)v;   // round to nearest even, as v_cvt_f32_f64 does
  uint32_t lo;
  std::memcpy(&lo, &f, sizeof lo);
  const uint64_t bits = 0x7FF8000000000000ull | lo;
  double d;
  std::memcpy(&d, &bits, sizeof d);
  return d;
}
void narrowFastRec(FastRec& f) {
  double* const slots[] = {&f.tair,  &f.tsoil,   &f.negPar,  &f.vpd,    &f.tillP1,  &f.rainRate,
                           &f.sublW, &f.evapNum, &f.invWspd, &f.tair10, &f.tsoil10, &f.log2vpd};
  for (double* p : slots) *p = narrowSlot(*p);
}

// summary of the tile [b, e) (see FastRec::tileBits); slot0 / slot1 hold the tile's eviction slots
void summariseTile(FastRec* out, int b, int e, const int* slot0, const int* slot1) {
  bool regular = true;
  int32_t dayMask = 0;
  const int nOps0 = out[b].bitsOps >> 16;
  if (nOps0 < 1 || nOps0 > 2) regular = false;
  auto next = [](int s) { return s + 1 == SIPNET_RING_SLOTS ? 0 : s + 1; };
  for (int t = b; t < e; t++) {
    const FastRec& f = out[t];
    const int bits = f.bitsOps & 0xffff;
    if (bits & FAST_PAR_POS) dayMask |= 1 << (t - b);
    if ((bits & (FAST_PHEN_NEW_YEAR | FAST_TRACK_NEW_YEAR)) || f.evCount != 0 || f.insSlot < 0 ||
        (f.bitsOps >> 16) != nOps0 || f.len != out[b].len || f.invLen != out[b].invLen ||
        f.w0 != out[b].w0 || f.w1 != out[b].w1)
      regular = false;
    if (t > b && (slot0[t - b] != next(slot0[t - b - 1]) || slot1[t - b] != next(slot1[t - b - 1]) ||
                  f.insSlot != next(out[t - 1].insSlot)))
      regular = false;
  }
  // what the regular path compares with the members' phenology thresholds: the LARGEST leaf-on variable
  // (year-to-date GDD; soil temperature or day of year with the GDD flag off) and day of year of the tile (inside a tile without a year roll-over both normally only grow,
  // but a forcing file with misordered or duplicated records may step back: the maximum, not the
  // last record's value, proves that no switch can fire anywhere in the tile)
  double maxGdd = out[b].cumGdd, maxDay = out[b].dayTime;
  for (int t = b + 1; t < e; t++) {
    if (out[t].cumGdd > maxGdd) maxGdd = out[t].cumGdd;
    if (out[t].dayTime > maxDay) maxDay = out[t].dayTime;
  }
  for (int t = b; t < e; t++) {
    out[t].tileBits = (regular ? FAST_TILE_REGULAR : 0) | (dayMask << 16);
    out[t].tilePad = 0;
    out[t].tileEndCumGdd = maxGdd;
    out[t].tileEndDayTime = maxDay;
  }
}
}  // namespace

SitePlan buildSitePlan(const int32_t* flags, int32_t n_steps, const double* clim,
                       const int32_t* year, const int32_t* day, int32_t n_events,
                       const sipnet_event* events, const PlanCarry* init, PlanCarry* fin,
                       bool wantSteps, StepRec* stepsOut, FastRec* fastOut, bool narrowFast) {
  SitePlan plan;
  if (wantSteps) plan.steps.resize(n_steps);
  plan.gddAfter.resize(n_steps);
  plan.dTill.resize(n_steps);
  plan.ringOps.reserve((size_t)n_steps * 2 + 8);
  double prevTsoil10 = 0.0;
  int tileSlot0[kFastTile], tileSlot1[kFastTile];
  const bool useEvents = flags[SIPNET_F_EVENTS] != 0;
  if (!useEvents) {
    n_events = 0;
  }

  const double convS = planConvS(), convE = planConvE();

  RingSched ring;               // fresh: one zero entry, insStep -1
  int trackLastYear = -1;       // trackers.lastYear, sipnet.c:1412
  double trackGdd = 0.0;        // trackers.gdd
  int phenLastYear = n_steps > 0 ? year[0] : 0;  // sipnet.c:1524
  double dTill = 0.0;           // events.c:809
  int evNext = 0;
  if (init && init->set) {
    // resumed segment: the checkpoint overwrites what setupModel() initialised
    // (sipnet.c:1963-1967).  Pre-loaded ring entries are live for every member, so they
    // carry insert step 0 (>= any member's ring_valid_from of 0).
    ring = init->ring;
    for (int i = 0; i < SIPNET_RING_SLOTS; i++) ring.insStep[i] = 0;
    trackLastYear = init->trackLastYear;
    trackGdd = init->gdd;
    phenLastYear = init->phenLastYear;
    dTill = init->dTill;
  }

  // frontend.c:216-223
  if (n_events > 0 && n_steps > 0) {
    const bool before = events[0].year != year[0] ? events[0].year < year[0]
                                                  : events[0].day < day[0];
    if (before) {
      plan.status = SIPNET_ERR_INPUT_FILE;
      plan.message = "First event occurs before the start of the climate file";
    }
  }

  for (int t = 0; t < n_steps; t++) {
    const double* r = clim + (size_t)SIPNET_NCLIM * t;
    StepRec s = StepRec{};
    s.length = r[0];
    s.tair = r[1];
    s.tsoil = r[2];
    s.par = r[3];
    s.precip = r[4];
    s.vpd = r[5];
    s.vpdSoil = r[6];
    s.vPress = r[7];
    s.wspd = r[8];
    const double gdd = r[9];
    s.dayTime = (double)day[t] + r[10] / 24.0;
    s.year = year[t];
    s.day = day[t];

    if (!(s.length > 0) && plan.status == SIPNET_OK) {  // events.c:460-465
      plan.status = SIPNET_ERR_BAD_PARAMETER;
      char buf[160];
      snprintf(buf, sizeof buf,
               "climate length (%f) on year %d day %d is non-positive", s.length,
               year[t], day[t]);
      plan.message = buf;
    }

    // phenology new-year reset, sipnet.c:811-815
    if (year[t] > phenLastYear) {
      s.bits |= STEP_PHEN_NEW_YEAR;
      phenLastYear = year[t];
    }
    // GDD seen by pastLeafGrowth() in this step, sipnet.c:706-716 (trackers are
    // still those of the previous step at that point)
    double cum = gdd;
    if (year[t] == trackLastYear) {
      cum += trackGdd;
    }
    s.cumGdd = cum;
    // updateTrackers(), sipnet.c:1421-1431, :1480-1484
    if (year[t] != trackLastYear) {
      s.bits |= STEP_TRACK_NEW_YEAR;
      trackGdd = 0.0;
      trackLastYear = year[t];
    }
    if (flags[SIPNET_F_GDD]) {
      trackGdd += gdd;
    } else {
      trackGdd = 0.0;
    }
    s.gddAfter = trackGdd;

    // events falling on this record, events.c:470-482
    s.evFirst = (int32_t)plan.events.size();
    while (evNext < n_events && events[evNext].year <= year[t] &&
           events[evNext].day <= day[t]) {
      const sipnet_event& ev = events[evNext];
      if ((ev.year < year[t] || ev.day < day[t]) && plan.status == SIPNET_OK) {
        plan.status = SIPNET_ERR_INPUT_FILE;
        char buf[160];
        snprintf(buf, sizeof buf,
                 "Agronomic event found for year: %d day: %d that does not have "
                 "a corresponding record in the climate file",
                 ev.year, ev.day);
        plan.message = buf;
      }
      if (ev.type == SIPNET_EV_TILL) {
        dTill += ev.p[0];  // events.c:629-639
      }
      if (ev.type == SIPNET_EV_IRRIG && (int)ev.p[1] != 0 && (int)ev.p[1] != 1 &&
          plan.status == SIPNET_OK) {
        plan.status = SIPNET_ERR_UNKNOWN_EVENT;  // events.c:497-500
        plan.message = "Unknown irrigation method type";
      }
      EvRec e;
      e.type = ev.type;
      e.pad = 0;
      for (int k = 0; k < 4; k++) e.p[k] = ev.p[k];
      plan.events.push_back(e);
      evNext++;
    }
    s.evCount = (int32_t)plan.events.size() - s.evFirst;
    s.dTill = dTill;
    // events.c:811-822
    if (dTill > 0) {
      dTill *= std::exp(-s.length * kTillDecay);
      if (dTill < kTillThreshold) {
        dTill = 0.0;
      }
    }
    s.tillAfter = dTill;

    // running-mean ring schedule, runmean.c:61-116
    s.ringOpFirst = (int32_t)plan.ringOps.size();
    bool overflow = false;
    s.ringInsSlot = ring.advance(t, s.length, &plan.ringOps, &overflow);
    if (overflow && plan.status == SIPNET_OK) {  // sipnet.c:1562-1569
      plan.status = SIPNET_ERR_INTERNAL;
      plan.message = "running-mean NPP ring overflow (more than 250 steps in 5 days)";
    }
    s.ringOpCount = (int32_t)plan.ringOps.size() - s.ringOpFirst;

    // member-independent sub-expressions for the fast-math variants
    s.invLen = 1.0 / s.length;
    s.tair10 = s.tair / 10.0;
    s.tsoil10 = s.tsoil / 10.0;
    s.log2vpd = std::log2(s.vpd > 0 ? s.vpd : kTiny);
    s.rainRate = s.precip / s.length;
    s.invWspd = 1.0 / s.wspd;
    s.sublNum = convS * (kEStarSnow - s.vPress);
    s.evapNum = convE * s.vpdSoil;

    plan.gddAfter[t] = s.gddAfter;
    plan.dTill[t] = s.dTill;
    if (t == 0) {
      plan.startCumGdd = s.cumGdd;
      plan.startTsoil = s.tsoil;
      plan.startDayTime = s.dayTime;
    }
    if (wantSteps) plan.steps[t] = s;
    if (stepsOut) stepsOut[t] = s;
    if (fastOut) {
      int s0, s1;
      fillFastRec(s, plan.ringOps.data() + s.ringOpFirst, t > 0 && prevTsoil10 == s.tsoil10,
                  flags[SIPNET_F_GDD] ? 0 : flags[SIPNET_F_SOIL_PHENOL] ? 1 : 2, flags[SIPNET_F_WATER_HRESP] != 0, fastOut[t],
                  s0, s1);
      // eviction slots: this step's, and (in the record of the step before) the next step's
      fastOut[t].slots = s0 | (s1 << 8) | (s0 << 16) | (s1 << 24);
      if (t > 0) fastOut[t - 1].slots = (fastOut[t - 1].slots & 0xffff) | (s0 << 16) | (s1 << 24);
      tileSlot0[t % kFastTile] = s0;
      tileSlot1[t % kFastTile] = s1;
      if (t % kFastTile == kFastTile - 1 || t == n_steps - 1) {
        summariseTile(fastOut, t - t % kFastTile, t + 1, tileSlot0, tileSlot1);
        // (after the summary, which compares the tile's records as doubles; while the tile is still in the cache)
        if (narrowFast)
          for (int k = t - t % kFastTile; k <= t; k++) narrowFastRec(fastOut[k]);
      }
    }
    prevTsoil10 = s.tsoil10;
  }
  if (fin) {
    fin->set = true;
    fin->gdd = trackGdd;
    fin->trackLastYear = trackLastYear;
    fin->phenLastYear = phenLastYear;
    fin->dTill = dTill;
    fin->ring = ring;
  }
  return plan;
}

PlanLight buildSitePlanLight(const int32_t* flags, int32_t n_steps, const double* clim, const int32_t* year, const int32_t* day,
                             int32_t n_events, const sipnet_event* events, const PlanCarry* init, double minLen, int32_t minRun,
                             double* gddAfter, int32_t* evFirst, int32_t* evCount, double* dTillOut, double* tillAfter) {
  PlanLight plan;
  if (!flags[SIPNET_F_EVENTS]) n_events = 0;
  plan.hasEvents = n_events > 0 || (init && init->set && init->dTill != 0.0);
  int trackLastYear = -1;       // trackers.lastYear, sipnet.c:1412
  double trackGdd = 0.0;        // trackers.gdd
  double dTill = 0.0;           // events.c:809
  int evNext = 0;
  if (init && init->set) {      // sipnet.c:1963-1967
    trackLastYear = init->trackLastYear;
    trackGdd = init->gdd;
    dTill = init->dTill;
  }
  if (n_events > 0 && n_steps > 0) {   // frontend.c:216-223
    const bool before = events[0].year != year[0] ? events[0].year < year[0] : events[0].day < day[0];
    if (before) {
      plan.status = SIPNET_ERR_INPUT_FILE;
      plan.message = "First event occurs before the start of the climate file";
    }
  }
  // runs of equal step lengths (plan_device.h: of a long run one lane walks the first 5 days / length + a few steps)
  int32_t runLen = 0;
  double runL = 0.0;
  auto closeRun = [&]() {
    const int64_t head = (int64_t)(kMeanNppDays / runL) + 6;
    plan.walked += (runLen >= minRun + head) ? head : runLen;
  };
  for (int t = 0; t < n_steps; t++) {
    const double* r = clim + (size_t)SIPNET_NCLIM * t;
    const double length = r[0], gdd = r[9];
    if (!(length >= minLen)) plan.lengthsOk = false;
    if (!(length > 0) && plan.status == SIPNET_OK) plan.status = SIPNET_ERR_BAD_PARAMETER;   // (worded by buildSitePlan: such a site is the host's)
    if (t > 0 && length == runL) {
      runLen++;
    } else {
      if (t > 0 && plan.lengthsOk) closeRun();
      runL = length;
      runLen = 1;
    }
    // sipnet.c:706-716, :1421-1431, :1480-1484
    double cum = gdd;
    if (year[t] == trackLastYear) cum += trackGdd;
    if (year[t] != trackLastYear) {
      trackGdd = 0.0;
      trackLastYear = year[t];
    }
    if (flags[SIPNET_F_GDD]) trackGdd += gdd;
    else trackGdd = 0.0;
    gddAfter[t] = trackGdd;
    if (t == 0) {
      plan.startCumGdd = cum;
      plan.startTsoil = r[2];
      plan.startDayTime = (double)day[0] + r[10] / 24.0;
    }
    if (!plan.hasEvents) continue;
    // events falling on this record, events.c:470-482
    const int32_t first = (int32_t)plan.events.size();
    while (evNext < n_events && events[evNext].year <= year[t] && events[evNext].day <= day[t]) {
      const sipnet_event& ev = events[evNext];
      if ((ev.year < year[t] || ev.day < day[t]) && plan.status == SIPNET_OK) {
        plan.status = SIPNET_ERR_INPUT_FILE;
        plan.message = "Agronomic event without a corresponding climate record";
      }
      if (ev.type == SIPNET_EV_TILL) dTill += ev.p[0];  // events.c:629-639
      if (ev.type == SIPNET_EV_IRRIG && (int)ev.p[1] != 0 && (int)ev.p[1] != 1 && plan.status == SIPNET_OK) {
        plan.status = SIPNET_ERR_UNKNOWN_EVENT;  // events.c:497-500
        plan.message = "Unknown irrigation method type";
      }
      EvRec e;
      e.type = ev.type;
      e.pad = 0;
      for (int k = 0; k < 4; k++) e.p[k] = ev.p[k];
      plan.events.push_back(e);
      evNext++;
    }
    evFirst[t] = first;
    evCount[t] = (int32_t)plan.events.size() - first;
    dTillOut[t] = dTill;
    if (dTill > 0) {   // events.c:811-822
      dTill *= std::exp(-length * kTillDecay);
      if (dTill < kTillThreshold) dTill = 0.0;
    }
    tillAfter[t] = dTill;
  }
  if (n_steps > 0 && plan.lengthsOk) closeRun();
  return plan;
}

}  // namespace sipnet

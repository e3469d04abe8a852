// plan.h -- site plan: everything about a run that is the same for every member
// of a site, worked out once on the host and streamed to the kernel as
// wave-uniform (scalar) data.
//
// The reference recomputes these per member because it runs one member per
// process: year roll-overs (sipnet.c:1421-1431, :811-815), year-to-date GDD
// (sipnet.c:706-716, :1480-1484), the weights and cursors of the running-mean
// ring (runmean.c:61-116 -- they depend only on the step lengths), which
// events fall on which climate record (events.c:470-482) and the tillage
// modifier and its decay (events.c:629-639, :811-822).  None of it depends on
// member state, so a batch does it once per site.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace sipnet {

// One record per (site, step); 32 x 8 B = 256 B so a step's record is two
// s_load_dwordx16 from the scalar cache.
struct StepRec {
  // converted climate, sipnet.c:201-238
  double length, tair, tsoil, par, precip, vpd, vpdSoil, vPress, wspd;  // 0..8
  double cumGdd;    //  9 year-to-date GDD compared with gddLeafOn in this step
  double dayTime;   // 10 day + time/24 (sipnet.c:722, :736)
  double dTill;     // 11 eventTrackers.d_till_mod in effect during this step
  double gddAfter;  // 12 trackers.gdd after this step (record column 33)
  // member-independent sub-expressions for the fast-math kernel variants
  double invLen;    // 13 1/length
  double tair10;    // 14 tair/10
  double tsoil10;   // 15 tsoil/10
  double log2vpd;   // 16 log2(vpd)
  double rainRate;  // 17 precip/length
  double invWspd;   // 18 1/wspd
  double sublNum;   // 19 CONV_S * (E_STAR_SNOW - vPress), sipnet.c:911
  double evapNum;   // 20 CONV * vpdSoil, sipnet.c:1000
  double tillAfter; // 21 d_till_mod after this step's decay (record column 34)
  double spare22, spare23, spare24, spare25;
  int32_t bits;          // 1 = phenology new year, 2 = tracker new year
  int32_t ringInsSlot;   // slot this step's NPP goes to; -1: ring reset to value
  int32_t ringOpFirst;   // index into RingOp array (global across sites)
  int32_t ringOpCount;
  int32_t evFirst;       // index into EvRec array (global across sites)
  int32_t evCount;
  int32_t year, day;
  int32_t pad[4];
};
static_assert(sizeof(StepRec) == 256, "StepRec must stay 256 bytes");

enum : int32_t { STEP_PHEN_NEW_YEAR = 1, STEP_TRACK_NEW_YEAR = 2 };

// One eviction of the running-mean ring: sum -= w * value[slot]
// (runmean.c:76-86; w is `weightLeft` for a partial or weights[i] for a full one).
struct RingOp {
  double w;
  int32_t slot;
  int32_t insStep;  // step that wrote the slot, -1 for the initial zero entry
};
static_assert(sizeof(RingOp) == 16, "RingOp layout");

struct EvRec {
  int32_t type, pad;
  double p[4];
};
static_assert(sizeof(EvRec) == 40, "EvRec layout");

// Compact record of the fast-math kernels (step_fast.hip): 256 B, staged through LDS in
// tiles of kFastTile steps.  The first 144 bytes are everything a normal step consumes (nine
// broadcast ds_read_b128), pre-combined so that no division by a site quantity is left in the
// time loop; the rest is read only when a flag bit says so.  It carries the step's ring
// evictions and the NEXT step's eviction slots, so that the ring values of step t+1 are
// requested during step t.
// NARROW fields (marked [n]): climate values the step only uses at the precision of its flux arithmetic.  In the
// records of an fp32-mixed batch (buildSitePlan(..., narrowFast = true)) such a slot holds the correctly rounded
// FLOAT in its low four bytes and a quiet-NaN tag in the high four; the fp32 kernels take the low word (fast_math.h,
// recR) instead of converting a wave-uniform double in every wavefront on every step.  fp64 batches: plain doubles.
struct FastRec {
  // ---- hot: bytes 0..143 ----   [n] tair tsoil | negPar vpd tillP1 rainRate | sublW evapNum invWspd tair10 | tsoil10; rare: log2vpd
  double len, invLen, tair, tsoil;           //  0.. 3
  double negPar, vpd, tillP1, rainRate;      //  4.. 7  -par, vpd, 1 + d_till_mod, precip/len
  double sublW, evapNum, invWspd, tair10;    //  8..11  CONV_S*(0.6-vPress)*wspd, CONV*vpdSoil, 1/wspd
  double tsoil10, cumGdd, dayTime, w0;       // 12..15  cumGdd: the leaf-on variable (year-to-date GDD; soil temperature
                                             //         or day of year by flag); w0: weight of the first ring eviction
  int32_t bitsOps;   // FAST_* flag bits | (number of ring evictions << 16)
  int32_t slots;     // slot0 | slot1<<8 | pfSlot0<<16 | pfSlot1<<24 (slots < 250)
  int32_t insSlot;   // slot of this step's insert, -1 = reset ring to the new value
  int32_t evCount;   // events on this record
  // ---- rare: read only under a flag ----
  double w1;         // weight of the second eviction (FAST_HAS_W1)
  double spareD;
  double log2vpd;    // for members whose dVpdExp is not 2
  double gddAfter, tillAfter;
  int32_t ins0, ins1;  // steps that wrote slot0 / slot1 (dead-member epochs)
  int32_t opFirst;     // site-local RingOp index of this step's eviction list (nOps > 2); + siteBase[3 * site]
  int32_t evFirst;     // site-local EvRec index; + siteBase[3 * site + 1]
  int32_t year, day;
  // ---- the 16-step tile this record belongs to (steps [16k, 16k+16) of the site) ----
  // tileBits: FAST_TILE_REGULAR when every step of the tile has the same length, the same one or
  // two ring evictions (weights w0, w1 constant; with half-hourly steps the steady state is TWO:
  // a 3e-15 residue of the oldest entry and all but that of the next one, because 240 x (1/48)
  // is not exactly 5 in floating point) and a plain insert, all slots advancing by one per step,
  // no events and no year roll-over: a wavefront then needs NO per-step record at all (constants
  // hoisted, slots counted) and reads this block once per tile.  Bits 16..31: the tile's
  // FAST_PAR_POS flags, bit 16+k for step 16k' + k.
  int32_t tileBits;                       // offset 208
  int32_t tilePad;
  double tileEndCumGdd, tileEndDayTime;   // 216, 224: the largest cumGdd / dayTime of the tile's steps
  int32_t pad[6];
};
static_assert(sizeof(FastRec) == 256, "FastRec must stay 256 bytes");
static_assert(offsetof(FastRec, tileBits) == 208 && offsetof(FastRec, tileEndCumGdd) == 216, "FastRec tile block");
enum : int32_t { FAST_TILE_REGULAR = 1 };
constexpr int kFastTile = 16;  // steps per LDS tile (4 KB)
enum : int32_t {
  FAST_PHEN_NEW_YEAR = 1, FAST_TRACK_NEW_YEAR = 2,
  FAST_TAIR_POS = 4,     // tair > 0
  FAST_PAR_POS = 8,      // par > 0
  FAST_TSOIL_NEG = 16,   // tsoil < 0 -- or the water_hresp flag is off: no moisture effect on heterotrophic respiration
  FAST_HAS_W1 = 32,      // a second eviction with non-zero weight
  FAST_HAS_TILL = 64,    // tillage modifier in effect
  FAST_TSOIL_SAME = 128, // tsoil identical to the previous record: Q10 factors can be reused
  FAST_RING_REGULAR = 256  // exactly one eviction and a plain insert (no ring reset)
};

// Weights and cursors of the running-mean ring as the reference holds them when a member
// inserts on every step (runmean.c:44-52, :61-116); they depend on the step lengths only.
struct RingSched {
  double w[SIPNET_RING_SLOTS];
  int32_t insStep[SIPNET_RING_SLOTS];  // step that wrote the slot (RingOp.insStep)
  int32_t start = 0, last = 0;
  RingSched() {
    for (int i = 0; i < SIPNET_RING_SLOTS; i++) {
      w[i] = 0.0;
      insStep[i] = -1;
    }
    reset(-1);
  }
  // resetMeanTracker(), runmean.c:44-52: one entry carrying the whole window
  void reset(int32_t step);
  // addValueToMeanTracker(), runmean.c:61-116, for a step of length `weight`: appends the
  // evictions to *ops (may be null), returns the insert slot (-1 = ring reset to the new
  // value); *overflow is set when the ring is full (return code -2 of the reference).
  int32_t advance(int32_t step, double weight, std::vector<RingOp>* ops, bool* overflow);
};

// Site-uniform state a plan starts from / ends with: what a restart checkpoint carries for
// the quantities the plan owns (restart.c:246, :262, :278, :294 and the mean.npp.* layout).
struct PlanCarry {
  bool set = false;
  double gdd = 0.0;          // trackers.gdd
  int32_t trackLastYear = -1;  // trackers.lastYear (sipnet.c:1412)
  int32_t phenLastYear = 0;  // phenologyTrackers.lastYear (sipnet.c:1524)
  double dTill = 0.0;        // eventTrackers.d_till_mod (events.c:809)
  RingSched ring;
};

// CONV_S / CONV of the snow sublimation and soil evaporation numerators (sipnet.c:44-49, :890-891, :968-969), as the
// plan folds them into its records (plan.cpp; plan_device.hip takes the same two doubles as kernel arguments)
inline double planConvS() {
  constexpr double kLambdaS = 2835000., kRho = 1.3, kCp = 1005., kGamma = 66., kSecPerDay = 86400.0;
  return (kRho * kCp) / kGamma * (1. / kLambdaS) * 1000. * 1000. * (1. / 10000) * kSecPerDay;
}
inline double planConvE() {
  constexpr double kLambda = 2501000., kRho = 1.3, kCp = 1005., kGamma = 66., kSecPerDay = 86400.0;
  return (kRho * kCp) / kGamma * (1. / kLambda) * 1000. * 1000. * (1. / 10000) * kSecPerDay;
}

struct SitePlan {
  std::vector<StepRec> steps;   // only when the caller asked for them (strict-order kernel, checkpoints)
  std::vector<double> gddAfter, dTill;  // per step: trackers.gdd after it, d_till_mod during it
  double startCumGdd = 0.0, startTsoil = 0.0, startDayTime = 0.0;  // of the first record (setupModel)
  std::vector<RingOp> ringOps;  // StepRec.ringOpFirst / FastRec.opFirst are local to this vector
  std::vector<EvRec> events;    // StepRec.evFirst / FastRec.evFirst are local to this vector
  int status = SIPNET_OK;       // site-fatal condition found while planning
  std::string message;
};

// Build the plan of one site in ONE pass over its climate.  clim[n_steps][SIPNET_NCLIM] converted
// climate.  The per-step records go where the caller wants them: stepsOut[n_steps] (StepRec, the
// strict-order kernel) and / or fastOut[n_steps] (FastRec with tile summaries, the throughput
// kernels) -- neither needs to be initialised, either may be null; wantSteps additionally keeps a
// copy of the StepRecs in the returned plan.  Ring-op and event indices in the records are local to
// the site (the kernels add the site's base, KernelArgs::siteBase).
SitePlan buildSitePlan(const int32_t* flags, int32_t n_steps, const double* clim,
                       const int32_t* year, const int32_t* day, int32_t n_events,
                       const sipnet_event* events, const PlanCarry* init = nullptr,
                       PlanCarry* fin = nullptr, bool wantSteps = true, StepRec* stepsOut = nullptr,
                       FastRec* fastOut = nullptr, bool narrowFast = false);


// The host's share of a DEVICE-built plan (plan_device.h): what of buildSitePlan's loop is sequential and cheap on a core --
// the year-to-date GDD chain, the events falling on each record with the tillage modifier's decay, the site-fatal
// conditions -- plus the facts that decide who builds the site (every step long enough that the ring cannot overflow, how
// many steps one lane would have to walk).  No ring schedule, no per-step records.  Same statements as buildSitePlan's,
// same order; tests/test_gpu_plan_device.py holds the two together byte by byte.
struct PlanLight {
  int status = SIPNET_OK;
  std::string message;
  std::vector<EvRec> events;               // as SitePlan::events
  bool hasEvents = false;                  // the four per-step arrays below were written
  double startCumGdd = 0.0, startTsoil = 0.0, startDayTime = 0.0;
  bool lengthsOk = true;                   // every step >= minLen (false for a NaN)
  int64_t walked = 0;                      // steps outside the part of a run of equal lengths that one descriptor covers
};
// gddAfter[n]: trackers.gdd after each record (always written).  evFirst / evCount / dTill / tillAfter [n]: written when the
// site has events (may be null when n_events is 0 or the events flag is off).
PlanLight buildSitePlanLight(const int32_t* flags, int32_t n_steps, const double* clim, const int32_t* year, const int32_t* day,
                             int32_t n_events, const sipnet_event* events, const PlanCarry* init, double minLen, int32_t minRun,
                             double* gddAfter, int32_t* evFirst, int32_t* evCount, double* dTill, double* tillAfter);

}  // namespace sipnet

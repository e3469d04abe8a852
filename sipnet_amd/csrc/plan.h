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
// tiles of kFastTile steps.  Holds only what that kernel consumes, pre-combined so that no
// division by a site quantity is left in the time loop, with the step's first two ring
// evictions inline and the NEXT step's eviction slots (so the ring values of step t+1 are
// requested at the top of step t and are in registers long before they are needed).
struct FastRec {
  double len, invLen, tair, tsoil;        //  0.. 3
  double negPar, vpd, log2vpd, vpd2;      //  4.. 7  -par, vpd, log2(vpd), vpd*vpd
  double rainRate, sublW, evapNum, invWspd;  //  8..11  precip/len, CONV_S*(0.6-vPress)*wspd, CONV*vpdSoil, 1/wspd
  double tair10, tsoil10, cumGdd, dayTime;   // 12..15
  double tillP1, w0, w1, gddAfter;        // 16..19  1+d_till_mod; weights of the inline ring evictions
  double tillAfter, spare0, spare1, spare2;  // 20..23
  int32_t bits;      // FAST_* below
  int32_t insSlot;   // slot of this step's insert, -1 = reset ring to the new value
  int32_t nOps;      // total evictions of this step (first two inline, rest via opFirst)
  int32_t slot0, slot1, ins0, ins1;  // inline evictions (slot1==slot0, w1==0 when nOps<2)
  int32_t opFirst;   // global RingOp index of this step's eviction list
  int32_t evFirst, evCount;
  int32_t pfSlot0, pfSlot1;  // slots evicted by step t+1 (prefetch targets)
  int32_t year, day, pad0, pad1;
};
static_assert(sizeof(FastRec) == 256, "FastRec must stay 256 bytes");
constexpr int kFastTile = 16;  // steps per LDS tile (4 KB)
enum : int32_t {
  FAST_PHEN_NEW_YEAR = 1, FAST_TRACK_NEW_YEAR = 2,
  FAST_TAIR_POS = 4,    // tair > 0
  FAST_PAR_POS = 8,     // par > 0
  FAST_TSOIL_NEG = 16,  // tsoil < 0
  FAST_PF_STALE = 32    // a prefetched slot of THIS step was written by step t-1: reload
};

struct SitePlan {
  std::vector<StepRec> steps;
  std::vector<RingOp> ringOps;  // StepRec.ringOpFirst is local to this vector
  std::vector<EvRec> events;    // StepRec.evFirst is local to this vector
  int status = SIPNET_OK;       // site-fatal condition found while planning
  std::string message;
};

// Build the plan of one site.  clim[n_steps][SIPNET_NCLIM] converted climate.
// Derive the fast records of one site from its plan (op/event indices stay site-local).
std::vector<FastRec> buildFastRecs(const SitePlan& plan);

SitePlan buildSitePlan(const int32_t* flags, int32_t n_steps, const double* clim,
                       const int32_t* year, const int32_t* day, int32_t n_events,
                       const sipnet_event* events);

}  // namespace sipnet

// plan_device.h -- the site plan built ON THE DEVICE from the uploaded climate (plan_device.hip).
//
// plan.cpp builds the 256-byte per-step records on host threads and sends them over PCIe (143 MB at 32 sites x
// 17 520 steps; 4.3 ms of a 20 ms hand-over of one forcing).  Here the same records are produced from the site's raw
// climate (63 MB over the wire instead) -- fresh or resumed from a checkpoint (the ring's live entries are handed over as
// the walk's initial queue), with or without agronomic events (the host's pass matches them to records and runs the
// tillage modifier's decay: 24 bytes a step for such a site):
//   planPrepKernel    step lengths as a compact array, per 256 steps "a length changes here" and the largest year
//                                                                                               (one thread per step)
//   planSeqKernel     the part that IS sequential in floating point, one wavefront per site: the running-mean ring's
//                     eviction schedule (runmean.c:61-116 over step lengths only) as a two-pointer walk; only the FRONT
//                     entry of the ring is ever partly evicted, so the state is (front entry, its remaining weight) and
//                     the other entries' weights are the lengths of their insert steps.  Inside a run of equal step
//                     lengths the walk reaches a fixed point (same remaining weight, front advancing by one): the rest
//                     of the run is described by ONE descriptor and filled in parallel (planRunsKernel) -- a year of
//                     half-hourly records is ~245 walked steps and one descriptor.  One lane walks at ~0.5 us a step (a
//                     lone wavefront issues an instruction every ~6 cycles), a host core at 0.07: forcings without long
//                     runs stay with the host (kDevPlanMaxWalked)
//   planRunsKernel    the steps covered by run descriptors: their eviction slots / insert steps / weights are the
//                     template step's, shifted                                                  (one thread per step)
//   planExpandKernel  the record itself: the member-independent sub-expressions of plan.cpp:319-327 (IEEE divisions,
//                     no contraction), flag bits (the phenology year roll-overs of sipnet.c:811-815 are a prefix
//                     maximum over the years), the next step's eviction slots, the 16-step tile summaries
//                     (cross-lane), the narrow fields of fp32-mixed batches; each record written once.
// The other sequential quantity, the year-to-date GDD sum (sipnet.c:1480-1484: an fp64 add chain restarted at each year
// roll-over), is the HOST's: the plan threads run it in the pass over the step lengths that decides who builds the site
// (~25 us a site; measured on the device: 200 us, the add's dependent-issue latency) and send 8 bytes a step.
// The records are bit-identical to buildSitePlan()'s -- except FastRec::log2vpd, which only members with dVpdExp != 2
// read: OCML's log2 is not glibc's to the last bit, so that field is filled from a HOST-computed array when (and only
// when) such a member exists (engine.hip, fillDeviceLog2).  tests/test_gpu_plan_device.py compares downloaded records
// and eviction lists byte by byte.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace sipnet {

// Sites whose every step is at least this long cannot overflow the 250-slot ring (249 whole entries of at least this
// weight exceed the 5-day window); shorter steps -- the reference stops with an error at 250 entries -- take the host
// path, which reports it.
constexpr double kDevPlanMinLen = 0.0202;
// ... and whose step lengths come in long runs: at most this many steps outside the part of a run that one descriptor covers
// (engine.hip deviceEligible; a year of half-hourly records: ~245)
constexpr int64_t kDevPlanMaxWalked = 1024;

struct DevPlanSite {
  const double* clim;     // device [n][SIPNET_NCLIM]
  const int32_t* year;    // device [n]
  const int32_t* day;     // device [n]
  const double* preW;     // device [SIPNET_RING_SLOTS]: weights of the ring's live entries before the first record, front first
  int32_t n;              // records of the site
  int32_t site;           // its position in the batch (records at fast + site * nT)
  int32_t opBase;         // its first RingOp in the flat array (room for opCap = 2 n + preK + 8: engine.hip devRingOpRoom)
  int32_t preK;           // number of those entries (a fresh ring: one, carrying the whole window -- runmean.c:44-52)
  int32_t preStart;       // the front one's slot
  int32_t preIns;         // their insert step in the eviction records: -1 fresh, 0 resumed (buildSitePlan's init)
  int32_t phenInit;       // phenologyTrackers.lastYear before the first record (sipnet.c:1524; a checkpoint's)
  int32_t trackInit;      // trackers.lastYear before the first record (sipnet.c:1412: -1; a checkpoint's)
  int32_t hasEvents;      // the site's rows of evFirst / evCount / dTill / tillAfter are filled
  int32_t opCap;          // entries of room behind opBase: the walk and planRunsKernel never write past it (status 3)
};

// what the ring walk (or planRunsKernel) leaves per step
struct DevPlanSeq {
  double w0, w1;
  int32_t ins0, ins1, opFirst, nOps;
  int32_t packed;         // slot0 | slot1 << 8 | (insSlot + 1) << 16
  int32_t r0, r1, r2;
  double r3, r4;
};
static_assert(sizeof(DevPlanSeq) == 64, "DevPlanSeq layout");

// steps t0 + 1 .. t0 + count repeat step t0's evictions with every slot and insert step advanced by the distance
struct DevPlanRun {
  int32_t t0, count, nOps, pad;
};
constexpr int kDevPlanMinRun = 48;   // shorter runs are walked

struct DevPlanArgs {
  const DevPlanSite* sites;   // device [nDev]
  int32_t nDev, nT;           // nT: stride of the per-step arrays (the batch's longest site)
  FastRec* fast;              // [n_sites][nT]
  RingOp* ringOps;
  double* lenC;               // [nDev][nT]
  const double* gddAfter;     // [nDev][nT]: trackers.gdd after each record -- the HOST's add chain (see above)
  // sites with events (DevPlanSite::hasEvents), from the host's pass (plan.cpp buildSitePlanLight): the events falling on
  // each record (site-local EvRec index + count), the tillage modifier during and after each record (events.c:811-822)
  const int32_t* evFirst;     // [nDev][nT] each
  const int32_t* evCount;
  const double* dTill;
  const double* tillAfter;
  DevPlanSeq* seq;            // [nDev][nT]
  DevPlanRun* runs;           // [nDev][runCap]
  int32_t runCap;
  int32_t* blockInfo;         // [nDev][nBlk][2]: per 256 steps, "a step length changes in here" and the largest year
  int32_t nBlk;               // ceil(nT / 256)
  int32_t* siteOut;           // [nDev][8]: descriptors written, ring evictions written, status (0 ok, 1 ring overflow,
                              //            2 ring ran empty, 3 eviction list out of room), the step it happened at, ticks (10 ns) of the ring walk, the room its eviction list had
  int32_t flagGdd, phenMode, moistHResp, narrow;
  double convS, convE;
};
inline int32_t devPlanRunCap(int32_t nT) { return nT / kDevPlanMinRun + 2; }
void launchDevicePlan(const DevPlanArgs& a, int32_t maxSteps, hipStream_t stream);
// log2vpd of device-built records from a host-computed array ([nDev][nT], device copy)
void launchDevicePlanLog2(const DevPlanSite* sites, int32_t nDev, int32_t nT, int32_t maxSteps, FastRec* fast, const double* log2vpd,
                          int32_t narrow, hipStream_t stream);

}  // namespace sipnet

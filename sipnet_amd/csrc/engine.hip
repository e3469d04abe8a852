// engine.hip -- the batch object behind include/sipnet_amd.h: HBM layout,
// uploads, launches.  No CPU compute path exists here: every compute entry
// point needs a HIP device and reports SIPNET_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sipnet_amd.h"
#include "batch_impl.h"
#include "plan.h"
#include "plan_pool.h"
#include "step_kernel.h"

namespace sipnet {
thread_local std::string g_lastError;
void setError(const std::string& s) { g_lastError = s; }
}  // namespace sipnet

using namespace sipnet;

static double nowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// -DSIPNET_TRACE_HOST (diagnostic build): where the host side of an upload spends its time / blocks
#ifdef SIPNET_TRACE_HOST
#define TRACE_T(label) fprintf(stderr, "  [host %.3f] %s\n", nowMs(), label)
#else
#define TRACE_T(label)
#endif

// Site plans are independent of each other: they are built by a pool of host threads (one site
// at a time each).  What every launch needs (ring evictions, events, site status, the first
// record's phenology inputs) is uploaded right away; the per-step records -- 256 B per step for
// the strict-order kernel, 256 B per step for the throughput kernels -- are flattened and uploaded
// on the first launch that reads them (ensureStepRecs / ensureFastRecs), so a batch pays for the
// record type it uses only.  bench.py reports the sum as plan_ms.
static int planThreadsFor(int nS) {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (n > 16) n = 16;
  return n > nS ? nS : n;
}
// (a worker that fails -- f returns false, or throws: std::bad_alloc on a huge forcing must not reach
// std::terminate in the caller's process -- stops the others at their next site; *failed says so)
template <class F>
static void forEachSite(int nS, int nThreads, std::atomic<bool>* failed, F f) {
  PlanPool::get().run(nS, nThreads, [&](int s) {
    if (failed->load()) return;
    bool ok = false;
    try {
      ok = f(s);
    } catch (...) {
      ok = false;
    }
    if (!ok) failed->store(true);
  });
}

// flat record buffers are written by the worker threads (first touch in parallel), so they are
// allocated without value-initialisation; the batch keeps them for the next hand-over of a forcing
// device buffer for `count` records; a launch that still reads the previous plan (on any stream)
// must have finished before the first site's records land in it
template <class Rec>
static int reserveRecords(sipnet_batch* b, Rec** d_ptr, size_t* cap, size_t count) {
  if (count > *cap) {
    int rcIdle = waitIdle(b);
    if (rcIdle) return rcIdle;
    if (*d_ptr) HIP_TRY(hipFree(*d_ptr));
    *d_ptr = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(d_ptr, count * sizeof(Rec)));
    *cap = count;
  }
  return SIPNET_OK;
}

// Which record type the next launch will read, as far as it is known at setup time: the
// throughput kernels' FastRec, or the strict-order kernel's StepRec.
static bool wantsFastRecs(const sipnet_batch* b) {
  return b->fastMath && b->kernelPolicy != SIPNET_KERNEL_STRICT;
}

// One pass per site (buildSitePlan) writes the record type the batch is set up for straight
// into the flat upload buffer; the other type is produced by a second pass only if a launch ever
// asks for it (ensureRecords).
// pinned host block of at least `count` records (kept between hand-overs of a forcing: a fresh buffer costs its
// first touch -- 143 MB at c4: 24 ms of page faults, more than building the records -- and its pinning)
template <class Rec>
static int reservePinned(Rec** ptr, size_t* cap, size_t count) {
  if (count <= *cap) return SIPNET_OK;
  if (*ptr) HIP_TRY(hipHostFree(*ptr));
  *ptr = nullptr;
  *cap = 0;
  HIP_TRY(hipHostMalloc((void**)ptr, count * sizeof(Rec), hipHostMallocDefault));
  *cap = count;
  return SIPNET_OK;
}

// uploads travel on the batch's own copy stream (nothing but copies is ever queued on it, so they are not held up
// behind another batch's step kernel in a shared hardware queue); the caller's stream waits for them
static int joinUploads(sipnet_batch* b, hipStream_t stream) {
  HIP_TRY(hipEventRecord(b->evStaged, b->upStream));
  b->staged = true;
  HIP_TRY(hipStreamWaitEvent(stream, b->evStaged, 0));
  return SIPNET_OK;
}

// ---- device-built site plans (plan_device.h) ---------------------------------------------------------------------------
// Who builds the plans of a hand-over?  The device, unless told otherwise (SIPNET_KOPT_HOST_PLAN) -- or unless this batch's
// own last launch is still running: a caller who hands the next forcing over while the previous one computes (one batch
// back to back, or two batches taking turns: bench.py's pipelined leg) has made the GPU the bottleneck, its host cores are
// idle, and the four plan kernels would only queue behind the step kernel (measured: c2x16 pipelined 9.6 -> 10.1 ms per
// forcing with them, against 14.1 -> 10.7 ms for a forcing handed to an idle device).  SIPNET_KOPT_DEVICE_PLAN: always.
static bool mayBuildOnDevice(const sipnet_batch* b) {
  if (!wantsFastRecs(b) || (b->kernelOptions & SIPNET_KOPT_HOST_PLAN)) return false;
  if (b->kernelOptions & SIPNET_KOPT_DEVICE_PLAN) return true;
  return !stillRunning(const_cast<sipnet_batch*>(b));
}
// the site's forcing block -> its device block, asynchronously on the copy stream (behind the plan kernels that may still
// be reading the previous forcing there)
static int sendClimate(sipnet_batch* b, int32_t site) {
  SiteClim& c = b->sc[site];
  const size_t bytes = SiteClim::bytesFor(c.n);
  if (bytes > c.devCap) {
    if (b->planKernelsQueued) HIP_TRY(hipEventSynchronize(b->evPlanDone));
    if (c.dev) HIP_TRY(hipFree(c.dev));
    c.dev = nullptr;
    c.devCap = 0;
    const size_t cap = bytes + bytes / 8;
    HIP_TRY(hipMalloc((void**)&c.dev, cap));
    c.devCap = cap;
  }
  if (b->planKernelsQueued) HIP_TRY(hipStreamWaitEvent(b->upStream, b->evPlanDone, 0));
  HIP_TRY(hipMemcpyAsync(c.dev, c.host, bytes, hipMemcpyHostToDevice, b->upStream));
  if (!c.evCopied) HIP_TRY(hipEventCreateWithFlags(&c.evCopied, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c.evCopied, b->upStream));
  c.copyQueued = true;
  c.onDevice = true;
  return SIPNET_OK;
}
// The plan threads' pass over a site before anybody builds its records (plan.cpp buildSitePlanLight): the GDD chain, the
// events per record and the tillage series, the site-fatal conditions -- and may the DEVICE build the records?  Every step
// (and every whole entry of a resumed ring) at least kDevPlanMinLen long, so that the ring cannot overflow; no site-fatal
// condition (the host path words the reference's message); step lengths in long runs: one lane walks the ring's schedule
// outside such runs at ~0.5 us a step (profiles/r05_plan_device.txt; a host core builds a whole step in 0.07 us), so a
// half-daily forcing like niwot's stays with the host unless SIPNET_KOPT_DEVICE_PLAN asks.
static bool devicePrepass(sipnet_batch* b, int32_t s, PlanLight* out) {
  const SiteClim& c = b->sc[s];
  const size_t nT = (size_t)b->n_steps;
  const bool ev = b->hostEv != nullptr;
  unsigned char* e = ev ? b->hostEv + (size_t)s * nT * 24 : nullptr;
  const PlanCarry* init = b->resume[s].set ? &b->resume[s] : nullptr;
  *out = buildSitePlanLight(b->flags, c.n, c.clim(), c.year(), c.day(), (int32_t)b->events[s].size(), b->events[s].data(), init,
                            kDevPlanMinLen, kDevPlanMinRun, b->hostGdd + (size_t)s * nT, (int32_t*)e, (int32_t*)(e + 4 * nT),
                            (double*)(e + 8 * nT), (double*)(e + 16 * nT));
  if (out->status != SIPNET_OK || !out->lengthsOk) return false;
  if (init) {   // the ring a checkpoint hands over: its whole entries (all but the front one) count like steps
    const RingSched& r = init->ring;
    for (int i = (r.start + 1) % SIPNET_RING_SLOTS; i != (r.last + 1) % SIPNET_RING_SLOTS && r.start != r.last; i = (i + 1) % SIPNET_RING_SLOTS)
      if (!(r.w[i] >= kDevPlanMinLen)) return false;
    if (!(r.w[r.start] > 0)) return false;
    // ... and together they carry the 5-day window: a ring that holds more reaches the reference's "ring full" stop
    // (runmean.c:93-95), one that holds less runs empty -- the host builder reports the first and walks the second as the
    // reference does; the device walk is only handed rings it cannot leave (its status word is a debugging aid)
    double sum = 0.0;
    for (int i = r.start;; i = (i + 1) % SIPNET_RING_SLOTS) {
      sum += r.w[i];
      if (i == r.last) break;
    }
    if (!(std::fabs(sum - 5.0) <= 1e-9)) return false;
  }
  return out->walked <= kDevPlanMaxWalked || (b->kernelOptions & SIPNET_KOPT_DEVICE_PLAN);
}

static int buildAndUpload(sipnet_batch* b, bool fastType, bool first, hipStream_t stream) {
  const double t0 = nowMs();
  const int nS = b->n_sites, nT = b->n_steps;   // nT: the longest site's records = the stride of the record arrays
  const int nThreads = planThreadsFor(nS);
  const size_t nFast = (size_t)nS * nT + kFastTile, nSteps = (size_t)nS * nT;
  TRACE_T("plan: begin");
  // the staging block must be free (the copies of the previous hand-over done: an event behind them); the DEVICE
  // records may still be read by this batch's last launch -- then the sites are built first (host only) and sent
  // once that launch has finished, instead of as they are built
  int rc = waitStaged(b);
  if (rc) return rc;
  rc = fastType ? reserveRecords(b, &b->d_fast, &b->fastCap, nFast) : reserveRecords(b, &b->d_plan, &b->planCap, nSteps);
  if (rc) return rc;
  const bool deferCopies = stillRunning(b);
  // (sites whose records the device builds itself need no staging: plan_device.h)
  const bool devPass = fastType && first && b->nDevSites > 0;
  const bool anyHostSite = !devPass || b->nDevSites < nS;
  if (anyHostSite) rc = fastType ? reservePinned(&b->hostFast, &b->hostFastCap, nFast) : reservePinned(&b->hostSteps, &b->hostStepsCap, nSteps);
  if (rc) return rc;
  TRACE_T("plan: reserved");
  FastRec* const fast = b->hostFast;
  StepRec* const steps = b->hostSteps;
  // every worker sends off the site it has just built while the others go on building: asynchronous copies out
  // of the pinned block on the caller's stream (the setup and step kernels that follow on it are ordered behind
  // them; the host does not wait, and nothing here needs a compute queue)
  std::atomic<int> copyErr{0};
  std::atomic<int64_t> copyUs{0};
  std::atomic<bool> failed{false};
  forEachSite(nS, nThreads, &failed, [&](int s) -> bool {
    const int nTs = b->siteSteps[s];            // this site's own length (its tail of the stride is never read)
    if (devPass && b->devSite[s]) {             // what the host still needs of such a site: setupModel()'s inputs, its events
      PlanLight& l = b->planLight[s];
      SitePlan p;
      p.startCumGdd = l.startCumGdd;
      p.startTsoil = l.startTsoil;
      p.startDayTime = l.startDayTime;
      p.events = std::move(l.events);
      b->plans[s] = std::move(p);
      return true;
    }
    SitePlan p = buildSitePlan(b->flags, nTs, b->sc[s].clim(), b->sc[s].year(), b->sc[s].day(),
                               (int32_t)b->events[s].size(), b->events[s].data(),
                               b->resume[s].set ? &b->resume[s] : nullptr, nullptr, /*wantSteps=*/false,
                               fastType ? nullptr : steps + (size_t)s * nT,
                               fastType ? fast + (size_t)s * nT : nullptr,
                               /*narrowFast=*/b->precision == SIPNET_F32_MIXED);
    if (first) b->plans[s] = std::move(p);
    const double c0 = nowMs();
    const size_t tail = (fastType && s == nS - 1) ? kFastTile : 0;  // tile padding after the last site
    if (tail) memset((void*)(fast + (size_t)nS * nT), 0, tail * sizeof(FastRec));
    if (deferCopies) return true;
    hipError_t e = hipSetDevice(b->device);
    if (e == hipSuccess) {
      e = fastType ? hipMemcpyAsync(b->d_fast + (size_t)s * nT, fast + (size_t)s * nT, ((size_t)nT + tail) * sizeof(FastRec),
                                    hipMemcpyHostToDevice, b->upStream)
                   : hipMemcpyAsync(b->d_plan + (size_t)s * nT, steps + (size_t)s * nT, (size_t)nT * sizeof(StepRec),
                                    hipMemcpyHostToDevice, b->upStream);
    }
    if (e != hipSuccess) {
      int none = 0;
      copyErr.compare_exchange_strong(none, (int)e);   // the FIRST error is the one reported
    }
    copyUs.fetch_add((int64_t)((nowMs() - c0) * 1e3));
    return e == hipSuccess;
  });
  if (copyErr.load() != 0) {
    setError(std::string("sipnet_batch: uploading the site records failed: ") + hipGetErrorString((hipError_t)copyErr.load()));
    return SIPNET_ERR_INTERNAL;
  }
  if (failed.load()) {
    setError("sipnet_batch: building the site plans failed (out of host memory?)");
    return SIPNET_ERR_INTERNAL;
  }
  if (deferCopies && anyHostSite) {   // once this batch's last launch is through with the old records
    rc = waitIdle(b);
    if (rc) return rc;
    if (!devPass) {   // everything in one piece
      if (fastType) HIP_TRY(hipMemcpyAsync(b->d_fast, fast, nFast * sizeof(FastRec), hipMemcpyHostToDevice, b->upStream));
      else HIP_TRY(hipMemcpyAsync(b->d_plan, steps, nSteps * sizeof(StepRec), hipMemcpyHostToDevice, b->upStream));
    } else {
      for (int s2 = 0; s2 < nS; s2++)
        if (!b->devSite[s2])
          HIP_TRY(hipMemcpyAsync(b->d_fast + (size_t)s2 * nT, fast + (size_t)s2 * nT,
                                 ((size_t)nT + (s2 == nS - 1 ? kFastTile : 0)) * sizeof(FastRec), hipMemcpyHostToDevice, b->upStream));
    }
  }
  if (deferCopies && !anyHostSite) {
    // every site is the device's (SIPNET_KOPT_DEVICE_PLAN forced while the previous launch still runs): nothing above has
    // waited for that launch, and the copy stream is about to overwrite what it reads -- the tile padding here, the site
    // bases / status / starts / events in uploadPlan.  A device-side wait: the host does not stop.
    rc = orderBehindBusy(b, b->upStream);
    if (rc) return rc;
  }
  // the tile padding behind the last site, when that one is the device's
  if (devPass && b->devSite[nS - 1]) HIP_TRY(hipMemsetAsync(b->d_fast + (size_t)nS * nT, 0, kFastTile * sizeof(FastRec), b->upStream));
  TRACE_T("plan: sites built, copies enqueued");
  (fastType ? b->fastRecsUploaded : b->stepRecsUploaded) = true;
  // wall time of the whole pass; the workers' share spent enqueueing the copies is reported as the upload part
  // (the copies themselves run on the stream, under the build of the following sites)
  const double wall = nowMs() - t0, copyShare = copyUs.load() * 1e-3 / nThreads;
  b->planBuildMs += wall - (copyShare < wall ? copyShare : wall);
  b->planUploadMs += copyShare < wall ? copyShare : wall;
  rc = joinUploads(b, stream);
  return rc ? rc : markBusy(b, stream);
}

// FastRec::log2vpd of the device-built records -- read only by members whose dVpdExp is not 2 (FastArgs::plainExp): the
// host's log2 (glibc's, as plan.cpp takes it), computed by the plan threads when such a member exists, sent and written
// into the records; the device's own log2 differs from it in the last bit now and then.
static int fillDeviceLog2(sipnet_batch* b, hipStream_t stream) {
  const int nS = b->n_sites, nT = b->n_steps, nDev = b->nDevSites;
  int rc = waitStaged(b);
  if (rc) return rc;
  rc = reservePinned(&b->hostLog2, &b->hostLog2Cap, (size_t)nDev * nT);
  if (rc) return rc;
  std::vector<int> siteOf;
  for (int s = 0; s < nS; s++)
    if (b->devSite[s]) siteOf.push_back(s);
  std::atomic<bool> none{false};
  forEachSite(nDev, planThreadsFor(nDev), &none, [&](int d) -> bool {
    const SiteClim& c = b->sc[siteOf[d]];
    double* out = b->hostLog2 + (size_t)d * nT;
    for (int32_t t = 0; t < c.n; t++) {
      const double vpd = c.clim()[(size_t)SIPNET_NCLIM * t + 5];
      out[t] = std::log2(vpd > 0 ? vpd : 0.000001);   // plan.cpp: log2 of vpd, of TINY (common/util.h:14) when not positive
    }
    return true;
  });
  HIP_TRY(hipMemcpyAsync(b->d_devLog2, b->hostLog2, (size_t)nDev * nT * sizeof(double), hipMemcpyHostToDevice, b->upStream));
  rc = joinUploads(b, stream);
  if (rc) return rc;
  launchDevicePlanLog2(b->devPlan.sites, nDev, nT, b->devPlanMaxSteps, b->d_fast, b->d_devLog2, b->precision == SIPNET_F32_MIXED, stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->evPlanDone, stream));
  b->devLog2Done = true;
  return markBusy(b, stream);
}

// Room for a device-built site's eviction list.  Every eviction either removes a whole entry (at most one per entry that ever
// lived: the n inserted ones + the preK a checkpoint's ring starts with, one for a fresh ring) or ends its step (at most n):
// 2 n + preK, + 8 spare.  The walk and planRunsKernel are also handed the number and stop writing at it (DevPlanSite::opCap).
static int32_t devRingPreK(const sipnet_batch* b, int s) {
  if (!b->resume[s].set) return 1;
  const RingSched& r = b->resume[s].ring;
  return (r.last - r.start + SIPNET_RING_SLOTS) % SIPNET_RING_SLOTS + 1;
}
static size_t devRingOpRoom(const sipnet_batch* b, int s) { return (size_t)2 * b->siteSteps[s] + devRingPreK(b, s) + 8; }

// The device-built sites' records: scratch carved out of one block, the site table sent, the four plan kernels queued on
// the caller's stream behind the climate copies.
static int buildOnDevice(sipnet_batch* b, const std::vector<int32_t>& bases, hipStream_t stream) {
  const int nS = b->n_sites, nT = b->n_steps, nDev = b->nDevSites;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const int32_t runCap = devPlanRunCap(nT), nBlk = (nT + 255) / 256;
  const size_t perStep = (size_t)nDev * nT;
  const size_t offSites = 0, offLen = offSites + align(nDev * sizeof(DevPlanSite)), offGdd = offLen + align(perStep * sizeof(double)),
               offSeq = offGdd + align(perStep * sizeof(double)), offRuns = offSeq + align(perStep * sizeof(DevPlanSeq)),
               offOut = offRuns + align((size_t)nDev * runCap * sizeof(DevPlanRun)), offLog2 = offOut + align((size_t)nDev * 8 * sizeof(int32_t)),
               offBlk = offLog2 + align(perStep * sizeof(double)), offPre = offBlk + align((size_t)nDev * nBlk * 2 * sizeof(int32_t)),
               offEv = offPre + align((size_t)nDev * SIPNET_RING_SLOTS * sizeof(double)),
               total = offEv + (b->hostEv ? align(perStep * 24) : 0);
  // (the events block: evFirst[nDev][nT], evCount[nDev][nT], dTill[nDev][nT], tillAfter[nDev][nT])
  const size_t offEvCount = offEv + perStep * 4, offDTill = offEv + perStep * 8, offTillAfter = offEv + perStep * 16;
  if (total > b->planScratchCap) {
    if (b->planKernelsQueued) HIP_TRY(hipEventSynchronize(b->evPlanDone));
    if (b->d_planScratch) HIP_TRY(hipFree(b->d_planScratch));
    b->d_planScratch = nullptr;
    b->planScratchCap = 0;
    HIP_TRY(hipMalloc((void**)&b->d_planScratch, total));
    b->planScratchCap = total;
  }
  // the site table (pinned staging: the small-array block is free again only after its copies, so a block of its own)
  std::vector<DevPlanSite> tab(nDev);
  std::vector<double> preW((size_t)nDev * SIPNET_RING_SLOTS, 0.0);
  int32_t maxSteps = 0;
  // (the scratch block is rewritten: behind the previous forcing's plan kernels)
  if (b->planKernelsQueued) HIP_TRY(hipStreamWaitEvent(b->upStream, b->evPlanDone, 0));
  for (int s = 0, d = 0; s < nS; s++) {
    if (!b->devSite[s]) continue;
    SiteClim& c = b->sc[s];
    if (!c.onDevice) {
      int rc = sendClimate(b, s);
      if (rc) return rc;
    }
    // the host's GDD chains (uploadPlan): one copy per run of neighbouring device-built sites (32 copies of 140 KB kept the
    // copy stream busy for 0.6 ms; rows are nT apart on both sides)
    if (b->flags[SIPNET_F_GDD] && (s == 0 || !b->devSite[s - 1])) {
      int e = s;
      while (e < nS && b->devSite[e]) e++;
      HIP_TRY(hipMemcpyAsync(b->d_planScratch + offGdd + (size_t)d * nT * sizeof(double), b->hostGdd + (size_t)s * nT,
                             (size_t)(e - s) * nT * sizeof(double), hipMemcpyHostToDevice, b->upStream));
    }
    const bool hasEv = b->planLight[s].hasEvents && b->hostEv;
    if (hasEv) {   // the events on each record and the tillage series (plan.cpp buildSitePlanLight)
      const unsigned char* h = b->hostEv + (size_t)s * nT * 24;
      unsigned char* dv = b->d_planScratch;
      HIP_TRY(hipMemcpyAsync(dv + offEv + (size_t)d * nT * 4, h, (size_t)c.n * 4, hipMemcpyHostToDevice, b->upStream));
      HIP_TRY(hipMemcpyAsync(dv + offEvCount + (size_t)d * nT * 4, h + 4 * (size_t)nT, (size_t)c.n * 4, hipMemcpyHostToDevice, b->upStream));
      HIP_TRY(hipMemcpyAsync(dv + offDTill + (size_t)d * nT * 8, h + 8 * (size_t)nT, (size_t)c.n * 8, hipMemcpyHostToDevice, b->upStream));
      HIP_TRY(hipMemcpyAsync(dv + offTillAfter + (size_t)d * nT * 8, h + 16 * (size_t)nT, (size_t)c.n * 8, hipMemcpyHostToDevice, b->upStream));
    }
    DevPlanSite& e = tab[d];
    e.clim = c.devClim();
    e.year = c.devYear();
    e.day = c.devDay();
    e.preW = (const double*)(b->d_planScratch + offPre) + (size_t)d * SIPNET_RING_SLOTS;
    e.n = c.n;
    e.site = s;
    e.opBase = bases[3 * s];
    // the ring the walk starts from: a fresh one (one entry carrying the 5-day window, runmean.c:44-52), or a checkpoint's
    double* pw = preW.data() + (size_t)d * SIPNET_RING_SLOTS;
    if (b->resume[s].set) {
      const PlanCarry& rc0 = b->resume[s];
      e.preK = devRingPreK(b, s);
      e.preStart = rc0.ring.start;
      e.preIns = 0;
      for (int i = 0; i < e.preK; i++) pw[i] = rc0.ring.w[(rc0.ring.start + i) % SIPNET_RING_SLOTS];
      e.phenInit = rc0.phenLastYear;
      e.trackInit = rc0.trackLastYear;
    } else {
      e.preK = 1;
      e.preStart = 0;
      e.preIns = -1;
      pw[0] = 5.0;                // MEAN_NPP_DAYS, sipnet.c:39
      e.phenInit = c.year()[0];   // sipnet.c:1524
      e.trackInit = -1;           // sipnet.c:1412
    }
    e.hasEvents = hasEv ? 1 : 0;
    e.opCap = (int32_t)devRingOpRoom(b, s);
    d++;
    maxSteps = std::max(maxSteps, c.n);
  }
  // (a pageable source: the runtime stages these few hundred bytes itself before the call returns)
  HIP_TRY(hipMemcpyAsync(b->d_planScratch + offSites, tab.data(), nDev * sizeof(DevPlanSite), hipMemcpyHostToDevice, b->upStream));
  HIP_TRY(hipMemcpyAsync(b->d_planScratch + offPre, preW.data(), preW.size() * sizeof(double), hipMemcpyHostToDevice, b->upStream));
  int rc = joinUploads(b, stream);
  if (rc) return rc;
  DevPlanArgs& a = b->devPlan;
  a.sites = (const DevPlanSite*)(b->d_planScratch + offSites);
  a.nDev = nDev;
  a.nT = nT;
  a.fast = b->d_fast;
  a.ringOps = b->d_ringOps;
  a.lenC = (double*)(b->d_planScratch + offLen);
  a.gddAfter = (const double*)(b->d_planScratch + offGdd);
  a.evFirst = (const int32_t*)(b->d_planScratch + offEv);
  a.evCount = (const int32_t*)(b->d_planScratch + offEvCount);
  a.dTill = (const double*)(b->d_planScratch + offDTill);
  a.tillAfter = (const double*)(b->d_planScratch + offTillAfter);
  a.seq = (DevPlanSeq*)(b->d_planScratch + offSeq);
  a.runs = (DevPlanRun*)(b->d_planScratch + offRuns);
  a.runCap = runCap;
  a.blockInfo = (int32_t*)(b->d_planScratch + offBlk);
  a.nBlk = nBlk;
  a.siteOut = (int32_t*)(b->d_planScratch + offOut);
  a.flagGdd = b->flags[SIPNET_F_GDD] != 0;
  a.phenMode = b->flags[SIPNET_F_GDD] ? 0 : b->flags[SIPNET_F_SOIL_PHENOL] ? 1 : 2;
  a.moistHResp = b->flags[SIPNET_F_WATER_HRESP] != 0;
  a.narrow = b->precision == SIPNET_F32_MIXED;
  a.convS = planConvS();
  a.convE = planConvE();
  b->d_devLog2 = (double*)(b->d_planScratch + offLog2);
  b->devPlanMaxSteps = maxSteps;
  launchDevicePlan(a, maxSteps, stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->evPlanDone, stream));
  b->planKernelsQueued = true;
  b->devLog2Done = false;
  rc = markBusy(b, stream);
  if (rc) return rc;
  return b->genericExponents ? fillDeviceLog2(b, stream) : SIPNET_OK;
}

static int uploadPlan(sipnet_batch* b, hipStream_t stream) {
  // every site needs a forcing; they may differ in length (a launch advances each site to the end of ITS records)
  const int nS = b->n_sites;
  b->siteSteps.assign(nS, 0);
  b->n_steps = 0;
  for (int s = 0; s < nS; s++) {
    if (b->sc[s].n <= 0) {
      setError("sipnet_batch: climate not set for every site");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    b->siteSteps[s] = b->sc[s].n;
    if (b->siteSteps[s] > b->n_steps) b->n_steps = b->siteSteps[s];
  }
  b->plans.clear();
  b->plans.resize(nS);
  b->stepRecsUploaded = false;
  b->fastRecsUploaded = false;
  b->planBuildMs = b->planUploadMs = 0.0;
  b->planThreads = planThreadsFor(nS);
  // which sites' records the device builds from the climate it has been sent (plan_device.h)
  std::fill(b->devSite.begin(), b->devSite.end(), 0);
  b->nDevSites = 0;
  if (mayBuildOnDevice(b)) {
    int rcW = waitStaged(b);   // (the previous hand-over's copies out of the staging blocks)
    if (rcW) return rcW;
    rcW = reservePinned(&b->hostGdd, &b->hostGddCap, (size_t)nS * b->n_steps);
    if (rcW) return rcW;
    bool anyEvents = false;
    for (int s = 0; s < nS; s++)
      anyEvents |= (b->flags[SIPNET_F_EVENTS] && !b->events[s].empty()) || (b->resume[s].set && b->resume[s].dTill != 0.0);
    if (anyEvents) {
      rcW = reservePinned(&b->hostEv, &b->hostEvCap, (size_t)nS * b->n_steps * 24);
      if (rcW) return rcW;
    } else if (b->hostEv) {   // (no site has events this time: the block is not looked at)
      HIP_TRY(hipHostFree(b->hostEv));
      b->hostEv = nullptr;
      b->hostEvCap = 0;
    }
    b->planLight.assign(nS, PlanLight{});
    std::atomic<bool> none{false};
    forEachSite(nS, b->planThreads, &none, [&](int s) -> bool {
      b->devSite[s] = devicePrepass(b, s, &b->planLight[s]) ? 1 : 0;
      return true;
    });
    for (int s = 0; s < nS; s++) b->nDevSites += b->devSite[s];
    // the plan kernels overwrite records this batch's last launch may still be reading on another stream
    if (b->nDevSites) {
      int rcO = orderBehindBusy(b, stream);
      if (rcO) return rcO;
    }
  }
  int rc = buildAndUpload(b, wantsFastRecs(b), /*first=*/true, stream);
  if (rc) return rc;
  const double t0 = nowMs();
  // ring evictions and events of all sites in one array each; the records index them site-locally
  // and the kernels add the site's base
  std::vector<int32_t> bases((size_t)3 * nS);   // per site: ring-op base, event base, number of records
  std::vector<SiteStart> starts(nS);
  size_t nOps = 0, nEv = 0;
  for (int s = 0; s < nS; s++) {
    const SitePlan& p = b->plans[s];
    b->siteStatus[s] = p.status;
    bases[3 * s] = (int32_t)nOps;
    bases[3 * s + 1] = (int32_t)nEv;
    bases[3 * s + 2] = b->siteSteps[s];
    nOps += b->devSite[s] ? devRingOpRoom(b, s) : p.ringOps.size();   // (the device's list: room for the bound)
    nEv += p.events.size();
    starts[s] = SiteStart{p.startCumGdd, p.startTsoil, p.startDayTime};
  }
  const double t1 = nowMs();
  if (nOps + 1 > b->ringOpCap) {
    if (b->d_ringOps) HIP_TRY(hipFree(b->d_ringOps));
    b->d_ringOps = nullptr;
    b->ringOpCap = 0;
    HIP_TRY(hipMalloc(&b->d_ringOps, (nOps + 1) * sizeof(RingOp)));
    b->ringOpCap = nOps + 1;
  }
  if (nEv + 1 > b->evCap) {
    if (b->d_events) HIP_TRY(hipFree(b->d_events));
    b->d_events = nullptr;
    b->evCap = 0;
    HIP_TRY(hipMalloc(&b->d_events, (nEv + 1) * sizeof(EvRec)));
    b->evCap = nEv + 1;
  }
  // the small arrays: flattened into one pinned block and sent on the same stream (buildAndUpload has waited for
  // every launch that might still read the previous plan; an empty list keeps one inert entry)
  // (ring evictions: the HOST-built sites' only -- a device-built site's list is written by its walk, and its room in the flat
  // array, 2 n + preK + 8 entries, is not sent: 18 MB and 0.32 ms of the copy stream at 32 sites x 17 520 records)
  size_t nHostOps = 0;
  for (int s = 0; s < nS; s++)
    if (!b->devSite[s]) nHostOps += b->plans[s].ringOps.size();
  const size_t opsBytes = (nHostOps ? nHostOps : 1) * sizeof(RingOp), evBytes = (nEv ? nEv : 1) * sizeof(EvRec);
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t offEv = align16(opsBytes), offStatus = offEv + align16(evBytes), offStart = offStatus + align16(nS * sizeof(int32_t)),
               offBase = offStart + align16(nS * sizeof(SiteStart)), total = offBase + align16(bases.size() * sizeof(int32_t));
  rc = reservePinned(&b->hostMisc, &b->hostMiscCap, total);
  if (rc) return rc;
  RingOp* hOps = (RingOp*)b->hostMisc;
  EvRec* hEv = (EvRec*)(b->hostMisc + offEv);
  std::vector<size_t> hostOff(nS, 0);
  {
    size_t off = 0;
    for (int s = 0; s < nS; s++) {
      const SitePlan& p = b->plans[s];
      hostOff[s] = off;
      if (!b->devSite[s] && !p.ringOps.empty()) {
        memcpy(hOps + off, p.ringOps.data(), p.ringOps.size() * sizeof(RingOp));
        off += p.ringOps.size();
      }
      if (!p.events.empty()) memcpy(hEv + bases[3 * s + 1], p.events.data(), p.events.size() * sizeof(EvRec));
    }
  }
  if (nEv == 0) hEv[0] = EvRec{0, 0, {0, 0, 0, 0}};
  memcpy(b->hostMisc + offStatus, b->siteStatus.data(), nS * sizeof(int32_t));
  memcpy(b->hostMisc + offStart, starts.data(), nS * sizeof(SiteStart));
  memcpy(b->hostMisc + offBase, bases.data(), bases.size() * sizeof(int32_t));
  if (nOps == 0) {
    hOps[0] = RingOp{0.0, 0, -1};
    HIP_TRY(hipMemcpyAsync(b->d_ringOps, hOps, sizeof(RingOp), hipMemcpyHostToDevice, b->upStream));
  }
  for (int s = 0; s < nS;) {   // runs of neighbouring host-built sites: contiguous here and there
    if (b->devSite[s]) { s++; continue; }
    int e = s;
    size_t cnt = 0;
    while (e < nS && !b->devSite[e]) cnt += b->plans[e++].ringOps.size();
    if (cnt) HIP_TRY(hipMemcpyAsync(b->d_ringOps + bases[3 * s], hOps + hostOff[s], cnt * sizeof(RingOp), hipMemcpyHostToDevice, b->upStream));
    s = e;
  }
  HIP_TRY(hipMemcpyAsync(b->d_events, hEv, evBytes, hipMemcpyHostToDevice, b->upStream));
  HIP_TRY(hipMemcpyAsync(b->d_siteStatus, b->hostMisc + offStatus, nS * sizeof(int32_t), hipMemcpyHostToDevice, b->upStream));
  HIP_TRY(hipMemcpyAsync(b->d_siteStart, b->hostMisc + offStart, nS * sizeof(SiteStart), hipMemcpyHostToDevice, b->upStream));
  HIP_TRY(hipMemcpyAsync(b->d_siteBase, b->hostMisc + offBase, bases.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->upStream));
  rc = joinUploads(b, stream);
  if (rc) return rc;
  rc = markBusy(b, stream);
  if (rc) return rc;
  TRACE_T("plan: small arrays enqueued");
  if (b->nDevSites) {
    rc = buildOnDevice(b, bases, stream);
    if (rc) return rc;
  }
  b->planDirty = false;
  b->exportCacheSite = -1;
  b->planBuildMs += t1 - t0;
  b->planUploadMs += nowMs() - t1;
  return SIPNET_OK;
}

static int ensureStepRecs(sipnet_batch* b, hipStream_t stream) {  // records of the strict-order kernel
  return b->stepRecsUploaded ? SIPNET_OK : buildAndUpload(b, /*fastType=*/false, /*first=*/false, stream);
}
static int ensureFastRecs(sipnet_batch* b, hipStream_t stream) {  // records of the throughput kernels
  return b->fastRecsUploaded ? SIPNET_OK : buildAndUpload(b, /*fastType=*/true, /*first=*/false, stream);
}

// The shape-based kernel choice of SIPNET_KERNEL_AUTO (also exported as sipnet_kernel_choice, so
// that tools and tests can ask without a device).
// Few 64-member chunks per CU: the step is bound by what one wavefront can issue, so three
// wavefronts share each chunk (step_coop.hip) -- with the chunk's ring in LDS when there is
// at most one chunk per CU (c10k 9.0 vs 18.2 ms); up to two per CU as ONE eight-wave workgroup
// per CU carrying two chunks with their rings in HBM, which keeps every carbon wave alone on
// its SIMD (c4 10.6 ms; two three-wave workgroups per CU: 12.4; one-wave kernel: 19.3).
// Up to four per CU: one twelve-wave workgroup per four chunks, every SIMD running the three
// waves of one chunk (c3 13.0 ms; one-wave kernel 15.3); no full-state build of that one (VGPRs).
// Bigger batches fill the SIMDs with the one-wave kernel, two waves per SIMD.
// The nitrogen-cycle flag set has cooperative kernels of its own (lean state; a soil wave S next to
// L, W, C: one chunk per CU in 213 registers / 71 KB of LDS, or two chunks per eight-wave workgroup);
// every other optional flag set takes the one-wave kernel.  Strict arithmetic and the debug plane: the strict-order kernel.  Full records,
// diagnostics and SIPNET_KOPT_FULL_STATE: the "Full" instantiations of the same throughput kernels.
// The running-mean ring on the device, [SIPNET_RING_SLOTS][ncol]: doubles, or -- fp32-mixed batches -- floats:
// the values are NPP rates, which such a batch computes in fp32, so the narrower store loses nothing and
// halves what a resampling moves (a ring value IMPORTED from a checkpoint is rounded to fp32 there).
// Host <-> device copies of ncols columns from col0, host side [slot][ncols] doubles.
static int ringToHost(sipnet_batch* b, int64_t col0, int64_t ncols, double* out) {
  const size_t eb = ringElemBytes(b);
  if (eb == sizeof(double)) {
    HIP_TRY(hipMemcpy2D(out, (size_t)ncols * eb, b->d_ring + col0, (size_t)b->ncol * eb, (size_t)ncols * eb,
                        SIPNET_RING_SLOTS, hipMemcpyDeviceToHost));
    return SIPNET_OK;
  }
  std::vector<float> tmp((size_t)ncols * SIPNET_RING_SLOTS);
  HIP_TRY(hipMemcpy2D(tmp.data(), (size_t)ncols * eb, (const float*)b->d_ring + col0, (size_t)b->ncol * eb,
                      (size_t)ncols * eb, SIPNET_RING_SLOTS, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < tmp.size(); i++) out[i] = (double)tmp[i];
  return SIPNET_OK;
}
static int ringFromHost(sipnet_batch* b, int64_t col0, int64_t ncols, const double* in) {
  const size_t eb = ringElemBytes(b);
  if (eb == sizeof(double)) {
    HIP_TRY(hipMemcpy2D(b->d_ring + col0, (size_t)b->ncol * eb, in, (size_t)ncols * eb, (size_t)ncols * eb,
                        SIPNET_RING_SLOTS, hipMemcpyHostToDevice));
    return SIPNET_OK;
  }
  std::vector<float> tmp((size_t)ncols * SIPNET_RING_SLOTS);
  for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (float)in[i];
  HIP_TRY(hipMemcpy2D((float*)b->d_ring + col0, (size_t)b->ncol * eb, tmp.data(), (size_t)ncols * eb,
                      (size_t)ncols * eb, SIPNET_RING_SLOTS, hipMemcpyHostToDevice));
  return SIPNET_OK;
}

// rows of 8-byte words, pitches in words (sipnet_dev_to_dev_2d)
__global__ __launch_bounds__(256) void copyRows8Kernel(uint64_t* __restrict__ dst, size_t dstPitch, const uint64_t* __restrict__ src,
                                                       size_t srcPitch, size_t width, size_t rows) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  for (size_t r = blockIdx.y; r < rows; r += gridDim.y) dst[r * dstPitch + c] = src[r * srcPitch + c];
}

// wantFull: 0 lean, 1 record / SIPNET_KOPT_FULL_STATE, 2 diagnostics counters as well
static int autoKernel(const int32_t* flags, int32_t n_sites, int32_t n_members, bool fastMath, bool debugPlane,
                      int wantFull, int32_t numCUs, bool f32) {
  const bool defaultFlags = isDefaultFlagSet(flags);
  const int64_t blocks = (int64_t)n_sites * ((n_members + 63) / 64);
  if (!fastMath || debugPlane) return SIPNET_KERNEL_STRICT;
  if (!flags[SIPNET_F_NITROGEN_CYCLE]) {
    // default physics, or -- up to two chunks per CU -- its optional-physics instantiations
    // (growth respiration, leaf water, flooding, litter pool, carbon saturation, anaerobic: run-time flags)
    const bool ext = !defaultFlags;
    if (blocks <= (int64_t)numCUs) return SIPNET_KERNEL_COOP_LDS;
    if (blocks <= 2 * (int64_t)numCUs) return SIPNET_KERNEL_COOP_PAIR;
    // (four chunks per CU with optional physics: the fp32-mixed build only -- the fp64 one would spill, step_coop.hip)
    if ((!ext || f32) && blocks <= 4 * (int64_t)numCUs && !wantFull) return SIPNET_KERNEL_COOP_QUAD;
    return SIPNET_KERNEL_ONE_WAVE;
  }
  // the nitrogen cycle (with litter pool + anaerobic, which it requires), alone or with the other options; full state
  // (record, every accumulator) and the diagnostics counters (wantFull == 2) too -- the plant side's mass totals travel to
  // the soil wave through eleven more mailbox rows: two slots of them on the one-chunk layout, one per chunk on the two-chunk
  // layout (round 6: the carbon wave waits for the soil wave's balance check of the step before; coop_mailboxes.inc)
  if (blocks <= (int64_t)numCUs) return SIPNET_KERNEL_COOP_NCYCLE;
  if (blocks <= 2 * (int64_t)numCUs) return SIPNET_KERNEL_COOP_NCYCLE_PAIR;
  return SIPNET_KERNEL_ONE_WAVE;
}

extern "C" {

int32_t sipnet_kernel_choice(const int32_t* flags, int32_t n_sites, int32_t n_members, int32_t precision,
                             int32_t math, int32_t want_full, int32_t num_cus) {
  if (!flags || n_sites <= 0 || n_members <= 0 || num_cus <= 0) return -1;
  const bool fast = precision == SIPNET_F32_MIXED || math == SIPNET_MATH_FAST;
  return autoKernel(flags, n_sites, n_members, fast, false, want_full, num_cus, precision == SIPNET_F32_MIXED);
}

const char* sipnet_version(void) { return "sipnet_amd 0.1 (reference SIPNET 2.1.0)"; }
const char* sipnet_last_error(void) { return g_lastError.c_str(); }

int sipnet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sipnet_batch_create(const int32_t* flags, int32_t n_sites, int32_t n_members,
                        int32_t precision, int32_t device, sipnet_batch** out) {
  if (!flags || !out || n_sites <= 0 || n_members <= 0 ||
      (precision != SIPNET_F64 && precision != SIPNET_F32_MIXED)) {
    setError("sipnet_batch_create: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  // flag coupling rules, common/context.c:195-223
  if ((flags[SIPNET_F_SOIL_PHENOL] && flags[SIPNET_F_GDD]) ||
      (flags[SIPNET_F_NITROGEN_CYCLE] &&
       !(flags[SIPNET_F_LITTER_POOL] && flags[SIPNET_F_ANAEROBIC])) ||
      (flags[SIPNET_F_ANAEROBIC] && !flags[SIPNET_F_WATER_HRESP]) ||
      (flags[SIPNET_F_CARBON_SATURATION] && !flags[SIPNET_F_LITTER_POOL])) {
    setError("sipnet_batch_create: incompatible model flags (context.c:195-223)");
    return SIPNET_ERR_BAD_PARAMETER;
  }
  if (sipnet_device_count() <= device || device < 0) {
    setError("sipnet_batch_create: no usable HIP device (this engine has no CPU path)");
    return SIPNET_ERR_NO_DEVICE;
  }
  sipnet_batch* b = new sipnet_batch();
  memcpy(b->flags, flags, sizeof(b->flags));
  b->n_sites = n_sites;
  b->n_members = n_members;
  b->precision = precision;
  b->device = device;
  b->ncol = (int64_t)n_sites * n_members;

  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) b->numCUs = prop.multiProcessorCount;
  }
  b->fastMath = (precision == SIPNET_F32_MIXED);
  b->sc.resize(n_sites);
  b->devSite.assign(n_sites, 0);
  b->events.resize(n_sites);
  b->resume.resize(n_sites);
  b->resumeProcessed.assign(n_sites, 0);
  b->siteStatus.assign(n_sites, 0);
  int rc = useDevice(b);
  if (rc) { delete b; return rc; }
  const size_t nc = (size_t)b->ncol;
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMalloc(&b->d_prm, nc * SIPNET_NPARAMS * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&b->d_state, nc * SIPNET_NSTATE * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&b->d_ring, nc * SIPNET_RING_SLOTS * ringElemBytes(b));
  if (e == hipSuccess) e = hipMalloc(&b->d_siteStatus, n_sites * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&b->d_siteStart, n_sites * sizeof(SiteStart));
  if (e == hipSuccess) e = hipMalloc(&b->d_siteBase, (size_t)3 * n_sites * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&b->d_scratchRow, nc * sizeof(double));
  if (e == hipSuccess) e = hipMemset(b->d_prm, 0, nc * SIPNET_NPARAMS * sizeof(double));
  if (e == hipSuccess) e = hipMemset(b->d_state, 0, nc * SIPNET_NSTATE * sizeof(double));
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evBusy, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evStaged, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evOrder, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evPlanDone, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->upStream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    setError(std::string("sipnet_batch_create: ") + hipGetErrorString(e));
    sipnet_batch_destroy(b);
    return SIPNET_ERR_NO_DEVICE;
  }
  *out = b;
  return SIPNET_OK;
}

void sipnet_batch_destroy(sipnet_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  pfRelease(b);
  if (b->d_rawStage) (void)hipFree(b->d_rawStage);
  if (b->hostRaw) (void)hipHostFree(b->hostRaw);
  if (b->hostFast) (void)hipHostFree(b->hostFast);
  if (b->hostSteps) (void)hipHostFree(b->hostSteps);
  if (b->hostMisc) (void)hipHostFree(b->hostMisc);
  if (b->hostLog2) (void)hipHostFree(b->hostLog2);
  if (b->hostGdd) (void)hipHostFree(b->hostGdd);
  if (b->hostEv) (void)hipHostFree(b->hostEv);
  for (SiteClim& c : b->sc) {
    if (c.host) (void)hipHostFree(c.host);
    if (c.dev) (void)hipFree(c.dev);
    if (c.evCopied) (void)hipEventDestroy(c.evCopied);
  }
  if (b->d_planScratch) (void)hipFree(b->d_planScratch);
  if (b->evPlanDone) (void)hipEventDestroy(b->evPlanDone);
  if (b->d_prm) (void)hipFree(b->d_prm);
  if (b->d_state) (void)hipFree(b->d_state);
  if (b->d_ring) (void)hipFree(b->d_ring);
  if (b->d_prm2) (void)hipFree(b->d_prm2);
  if (b->d_prmId) (void)hipFree(b->d_prmId);
  if (b->d_prmId2) (void)hipFree(b->d_prmId2);
  if (b->d_prmBank) (void)hipFree(b->d_prmBank);
  if (b->d_pfCrossing) (void)hipFree(b->d_pfCrossing);
  if (b->d_state2) (void)hipFree(b->d_state2);
  if (b->d_ring2) (void)hipFree(b->d_ring2);
  if (b->d_plan) (void)hipFree(b->d_plan);
  if (b->d_fast) (void)hipFree(b->d_fast);
  if (b->d_scratchRow) (void)hipFree(b->d_scratchRow);
  if (b->d_ringOps) (void)hipFree(b->d_ringOps);
  if (b->d_events) (void)hipFree(b->d_events);
  if (b->d_siteStatus) (void)hipFree(b->d_siteStatus);
  if (b->d_siteStart) (void)hipFree(b->d_siteStart);
  if (b->d_siteBase) (void)hipFree(b->d_siteBase);
  if (b->d_diag) (void)hipFree(b->d_diag);
  if (b->d_pfPreMax) (void)hipFree(b->d_pfPreMax);
  if (b->d_statsPart) (void)hipFree(b->d_statsPart);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->evBusy) (void)hipEventDestroy(b->evBusy);
  if (b->evStaged) (void)hipEventDestroy(b->evStaged);
  if (b->evOrder) (void)hipEventDestroy(b->evOrder);
  if (b->upStream) (void)hipStreamDestroy(b->upStream);
  delete b;
}

// the host side of a hand-over of one site's forcing, in three parts so that the copies of several sites can run on the
// plan threads (sipnet_batch_set_climate_sites): room in the pinned block, the copy, the send-off
static int climateReserve(sipnet_batch* b, int32_t site, int32_t n_steps) {
  SiteClim& c = b->sc[site];
  const size_t bytes = SiteClim::bytesFor(n_steps);
  // the previous forcing's copy out of this block must be through before the host writes it again
  if (c.copyQueued) HIP_TRY(hipEventSynchronize(c.evCopied));
  c.copyQueued = false;
  if (bytes > c.hostCap) {
    if (c.host) HIP_TRY(hipHostFree(c.host));
    c.host = nullptr;
    c.hostCap = 0;
    const size_t cap = bytes + bytes / 8;
    HIP_TRY(hipHostMalloc((void**)&c.host, cap, hipHostMallocDefault));
    c.hostCap = cap;
  }
  c.n = n_steps;
  c.onDevice = false;
  return SIPNET_OK;
}
static void climateCopy(sipnet_batch* b, int32_t site, const double* clim, const int32_t* year, const int32_t* day) {
  SiteClim& c = b->sc[site];
  memcpy(c.host, clim, (size_t)c.n * SIPNET_NCLIM * sizeof(double));
  memcpy((void*)c.year(), year, (size_t)c.n * sizeof(int32_t));
  memcpy((void*)c.day(), day, (size_t)c.n * sizeof(int32_t));
}
static void climateDone(sipnet_batch* b) {
  b->n_steps = 0;   // the longest site set so far (sites may differ in length; the plan is rebuilt anyway)
  for (int s = 0; s < b->n_sites; s++) b->n_steps = std::max<int32_t>(b->n_steps, b->sc[s].n);
  b->planDirty = true;
}

int sipnet_batch_set_climate(sipnet_batch* b, int32_t site, int32_t n_steps,
                             const double* clim, const int32_t* year, const int32_t* day) {
  if (!b || site < 0 || site >= b->n_sites || n_steps <= 0 || !clim || !year || !day) {
    setError("sipnet_batch_set_climate: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  rc = climateReserve(b, site, n_steps);
  if (rc) return rc;
  climateCopy(b, site, clim, year, day);
  climateDone(b);
  // a batch that may build the site's plan on the device sends the forcing off now: the copy runs under the caller's
  // preparation of the next site (63 MB at 32 sites x 17 520 records, against 143 MB of host-built records)
  return mayBuildOnDevice(b) ? sendClimate(b, site) : SIPNET_OK;
}

int sipnet_batch_set_climate_sites(sipnet_batch* b, int32_t first_site, int32_t count, const int32_t* n_steps,
                                   const double* const* clim, const int32_t* const* year, const int32_t* const* day) {
  if (!b || first_site < 0 || count <= 0 || first_site + count > b->n_sites || !n_steps || !clim || !year || !day) {
    setError("sipnet_batch_set_climate_sites: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  for (int32_t k = 0; k < count; k++) {
    if (n_steps[k] <= 0 || !clim[k] || !year[k] || !day[k]) {
      setError("sipnet_batch_set_climate_sites: bad argument");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
  }
  int rc = useDevice(b);
  if (rc) return rc;
  for (int32_t k = 0; k < count; k++) {
    rc = climateReserve(b, first_site + k, n_steps[k]);
    if (rc) return rc;
  }
  // (each thread sends its site off as soon as it is copied, so the DMA of the first sites runs under the copies of the
  // others.  Eight threads: the copies are bound by the host's memory fabric -- 63 MB in 1.6 ms = 39 GB/s of copy on this
  // box, the 36 us DMAs wait for them; sixteen threads were slower, 1.9 ms, and slowed the DMAs to 55 us)
  const bool send = mayBuildOnDevice(b);
  std::atomic<int> firstErr{0};
  std::string errText;
  std::mutex errMu;
  PlanPool::get().run(count, std::min(planThreadsFor(count), 8), [&](int k) {
    climateCopy(b, first_site + k, clim[k], year[k], day[k]);
    if (!send) return;
    int rcS = useDevice(b);
    if (!rcS) rcS = sendClimate(b, first_site + k);
    int none = 0;
    if (rcS && firstErr.compare_exchange_strong(none, rcS)) {
      std::lock_guard<std::mutex> lk(errMu);
      errText = sipnet_last_error();   // (the error text is per thread: carried to the caller's)
    }
  });
  climateDone(b);
  if (firstErr.load()) {
    setError(errText);
    return firstErr.load();
  }
  return SIPNET_OK;
}

int sipnet_batch_set_events(sipnet_batch* b, int32_t site, int32_t n_events,
                            const sipnet_event* events) {
  if (!b || site < 0 || site >= b->n_sites || n_events < 0 || (n_events > 0 && !events)) {
    setError("sipnet_batch_set_events: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  for (int i = 0; i < n_events; i++) {
    if (events[i].type < SIPNET_EV_FERT || events[i].type > SIPNET_EV_LEAFOFF) {
      setError("sipnet_batch_set_events: unknown event type");
      return SIPNET_ERR_UNKNOWN_EVENT;
    }
    // events.c:334-341: records must be in time-ascending order
    if (i > 0 && (events[i].year < events[i - 1].year ||
                  (events[i].year == events[i - 1].year && events[i].day < events[i - 1].day))) {
      setError("sipnet_batch_set_events: event records must be in time-ascending order");
      return SIPNET_ERR_INPUT_FILE;
    }
  }
  b->events[site].assign(events, events + n_events);
  b->planDirty = true;
  return SIPNET_OK;
}

int sipnet_batch_set_params(sipnet_batch* b, int32_t site, int32_t first_member,
                            int32_t count, const double* raw) {
  if (!b || (site != SIPNET_ALL_SITES && (site < 0 || site >= b->n_sites)) || first_member < 0 || count <= 0 ||
      first_member + count > b->n_members || !raw) {
    setError("sipnet_batch_set_params: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int32_t nRep = site == SIPNET_ALL_SITES ? b->n_sites : 1;
  if (site == SIPNET_ALL_SITES) site = 0;
  int rc = useDevice(b);
  if (rc) return rc;
  for (int32_t m = 0; m < count; m++) {
    const double* r = raw + (size_t)m * SIPNET_NPARAMS;
    if (r[SP_dVpdExp] != 2.0 || r[SP_soilRespMoistEffect] != 1.0) b->genericExponents = true;
  }
  const int64_t col0 = (int64_t)site * b->n_members + first_member;
  // no upload of earlier rows may still be reading the staging block (an event is recorded behind the copy)
  rc = waitStaged(b);
  if (rc) return rc;
  const size_t need = b->hostRawUsed + (size_t)count;
  if (need > b->hostRawCap) {
    double* bigger = nullptr;
    const size_t cap = need > 2 * b->hostRawCap ? need : 2 * b->hostRawCap;
    HIP_TRY(hipHostMalloc((void**)&bigger, cap * SIPNET_NPARAMS * sizeof(double), hipHostMallocDefault));
    if (b->hostRawUsed) memcpy(bigger, b->hostRaw, b->hostRawUsed * SIPNET_NPARAMS * sizeof(double));
    if (b->hostRaw) HIP_TRY(hipHostFree(b->hostRaw));
    b->hostRaw = bigger;
    b->hostRawCap = cap;
  }
  memcpy(b->hostRaw + b->hostRawUsed * SIPNET_NPARAMS, raw, (size_t)count * SIPNET_NPARAMS * sizeof(double));
  b->pendingParams.push_back({b->hostRawUsed, col0, count, nRep});
  b->hostRawUsed = need;
  return SIPNET_OK;
}

}  // extern "C"

// The parameter half of setupModel() (sipnet.c:1873-1916) for everything set_params has staged, on the caller's
// stream: ONE upload of the raw rows, one conversion launch per set_params call; from here on the converted block
// is the only copy of the members' parameters on the device.  The caller records the batch busy behind it.
int flushParams(sipnet_batch* b, hipStream_t stream) {
  if (b->pendingParams.empty()) return SIPNET_OK;
  {   // new rows are converted into column order: a resampled index must be resolved first
    int rcM = materializeParams(b, stream);
    if (rcM) return rcM;
    // (... and the copy of this rank's parameters that a connected filter's peers hold is out of date: connect again)
    pfDropBank(b);
  }
  if (b->hostRawUsed > b->rawStageCap) {
    int rcI = waitIdle(b);
    if (rcI) return rcI;
    if (b->d_rawStage) HIP_TRY(hipFree(b->d_rawStage));
    b->d_rawStage = nullptr;
    b->rawStageCap = 0;
    HIP_TRY(hipMalloc(&b->d_rawStage, b->hostRawUsed * SIPNET_NPARAMS * sizeof(double)));
    b->rawStageCap = b->hostRawUsed;
  }
  // The conversion writes d_prm: it must not start while this batch's last launch -- possibly on ANOTHER stream of
  // the caller's (a node shard's, the null stream of pf_publish) -- still reads it.  A device-side wait, no host stall.
  {
    int rcO = orderBehindBusy(b, stream);
    if (rcO) return rcO;
  }
  // (an earlier conversion on `stream` may still read the device block: the copy stream waits for the caller's first)
  HIP_TRY(hipEventRecord(b->evOrder, stream));
  HIP_TRY(hipStreamWaitEvent(b->upStream, b->evOrder, 0));
  HIP_TRY(hipMemcpyAsync(b->d_rawStage, b->hostRaw, b->hostRawUsed * SIPNET_NPARAMS * sizeof(double), hipMemcpyHostToDevice, b->upStream));
  int rcS = joinUploads(b, stream);
  if (rcS) return rcS;
  for (const auto& p : b->pendingParams)
    launchConvertParams(b->d_rawStage + p.row0 * SIPNET_NPARAMS, b->d_prm, b->ncol, p.col0, p.count,
                        b->flags[SIPNET_F_GDD] ? 0 : b->flags[SIPNET_F_SOIL_PHENOL] ? 1 : 2, stream, p.nRep, b->n_members);
  HIP_TRY(hipGetLastError());
  b->pendingParams.clear();
  b->hostRawUsed = 0;
  return markBusy(b, stream);   // (whoever flushes next, on whatever stream, waits for these conversions)
}

extern "C" {

int sipnet_batch_setup(sipnet_batch* b, void* hip_stream) {
  if (!b) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  b->pfPre.valid = false;   // (log-weights a forecast left, an analysis it was armed for: of the state this call replaces)
  b->pfArm.set = false;
  if (b->planDirty) {
    rc = uploadPlan(b, stream);
    if (rc) return rc;
  }
  rc = flushParams(b, stream);
  if (rc) return rc;
  rc = materializeParams(b, stream);   // (setupModel() for every column from the parameters it carries now)
  if (rc) return rc;
  SetupArgs a;
  a.siteStart = b->d_siteStart;
  a.prm = b->d_prm;
  a.state = b->d_state;
  a.ring = b->d_ring;
  a.ringF32 = b->precision == SIPNET_F32_MIXED;
  a.ncol = b->ncol;
  a.n_sites = b->n_sites;
  a.n_members = b->n_members;
  memcpy(a.flags, b->flags, sizeof(a.flags));
  a.siteStatus = b->d_siteStatus;
  launchSetup(a, stream);
  HIP_TRY(hipGetLastError());
  if (b->d_diag) HIP_TRY(hipMemsetAsync(b->d_diag, 0, (size_t)4 * b->ncol * sizeof(double), stream));
  rc = markBusy(b, stream);
  if (rc) return rc;
  b->stepsDone = 0;
  // a site-fatal plan condition is reported like the reference's exit code
  for (int s = 0; s < b->n_sites; s++) {
    if (b->siteStatus[s] != SIPNET_OK) {
      setError("site " + std::to_string(s) + ": " + b->plans[s].message);
      return b->siteStatus[s];
    }
  }
  return SIPNET_OK;
}

static int runImpl(sipnet_batch* b, int32_t step0, int32_t n_steps, void* d_nee, void* d_gpp,
                   void* d_et, double* d_rec, double* d_dbg, int64_t ld, void* hip_stream,
                   double* d_stats = nullptr, int32_t sumEvery = 0);

int sipnet_batch_set_math(sipnet_batch* b, int32_t policy) {
  if (!b || (policy != SIPNET_MATH_STRICT && policy != SIPNET_MATH_FAST)) {
    setError("sipnet_batch_set_math: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->precision == SIPNET_F32_MIXED && policy == SIPNET_MATH_STRICT) {
    setError("sipnet_batch_set_math: an fp32-mixed batch has no strict-order arithmetic");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  b->fastMath = policy == SIPNET_MATH_FAST;
  return SIPNET_OK;
}

int sipnet_batch_set_kernel(sipnet_batch* b, int32_t kernel, int32_t options) {
  if (!b || kernel < SIPNET_KERNEL_AUTO || kernel > SIPNET_KERNEL_COOP_NCYCLE_PAIR ||
      (options & ~(SIPNET_KOPT_ONE_WAVE_PER_SIMD | SIPNET_KOPT_RUNTIME_FLAGS | SIPNET_KOPT_FULL_STATE |
                   SIPNET_KOPT_NO_REGULAR_TILES | SIPNET_KOPT_STATS_IN_KERNEL | SIPNET_KOPT_BOUNDED_WAITS | SIPNET_KOPT_WAIT_SELFTEST |
                   SIPNET_KOPT_HOST_PLAN | SIPNET_KOPT_DEVICE_PLAN | SIPNET_KOPT_PF_MULTI_LAUNCH | SIPNET_KOPT_PF_MOVE_PARAMS))) {
    setError("sipnet_batch_set_kernel: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if ((options ^ b->kernelOptions) & (SIPNET_KOPT_HOST_PLAN | SIPNET_KOPT_DEVICE_PLAN)) b->planDirty = true;   // (who builds the plan has changed)
  b->kernelPolicy = kernel;
  b->kernelOptions = options;
  return SIPNET_OK;
}

int sipnet_batch_set_device_share(sipnet_batch* b, int32_t n_filters) {
  if (!b || n_filters < 1 || n_filters > 1024) {
    setError("sipnet_batch_set_device_share: bad argument (1 <= n_filters <= 1024)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  b->deviceShare = n_filters;
  return SIPNET_OK;
}

int sipnet_debug_set_num_cus(sipnet_batch* b, int32_t num_cus) {
  if (!b || num_cus < 1 || num_cus > 4096) return SIPNET_ERR_BAD_ARGUMENT;
  b->numCUs = num_cus;
  return SIPNET_OK;
}

int sipnet_debug_pf_barrier(sipnet_batch* b, int32_t spin_budget, int32_t absent_workgroup) {
  if (!b || spin_budget < 0) return SIPNET_ERR_BAD_ARGUMENT;
  b->pfSpinBudget = spin_budget;
  b->pfDebugAbsent = absent_workgroup;
  return SIPNET_OK;
}

int sipnet_batch_enable_diagnostics(sipnet_batch* b, int32_t on) {
  if (!b) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  if (on && !b->d_diag) {
    HIP_TRY(hipMalloc(&b->d_diag, (size_t)4 * b->ncol * sizeof(double)));
    HIP_TRY(hipMemset(b->d_diag, 0, (size_t)4 * b->ncol * sizeof(double)));
  } else if (!on && b->d_diag) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(b->d_diag));
    b->d_diag = nullptr;
  }
  return SIPNET_OK;
}

int sipnet_batch_get_diagnostics(sipnet_batch* b, int64_t* n_clamp_warn, int64_t* n_balance_warn,
                                 double* max_abs_dC, double* max_abs_dN, void* hip_stream) {
  if (!b) return SIPNET_ERR_BAD_ARGUMENT;
  if (!b->d_diag) {
    setError("sipnet_batch_get_diagnostics: call sipnet_batch_enable_diagnostics first");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  std::vector<double> tmp((size_t)4 * b->ncol);
  HIP_TRY(hipMemcpy(tmp.data(), b->d_diag, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t c = 0; c < b->ncol; c++) {
    if (n_clamp_warn) n_clamp_warn[c] = (int64_t)tmp[c];
    if (n_balance_warn) n_balance_warn[c] = (int64_t)tmp[(size_t)b->ncol + c];
    if (max_abs_dC) max_abs_dC[c] = tmp[(size_t)2 * b->ncol + c];
    if (max_abs_dN) max_abs_dN[c] = tmp[(size_t)3 * b->ncol + c];
  }
  return SIPNET_OK;
}

int sipnet_batch_run(sipnet_batch* b, int32_t step0, int32_t n_steps, void* d_nee,
                     void* d_gpp, void* d_et, double* d_rec, int64_t ld, void* hip_stream) {
  return runImpl(b, step0, n_steps, d_nee, d_gpp, d_et, d_rec, nullptr, ld, hip_stream);
}

int sipnet_batch_run_stats(sipnet_batch* b, int32_t step0, int32_t n_steps, void* d_nee, void* d_gpp,
                           void* d_et, int64_t ld, double* d_stats, void* hip_stream) {
  if (!d_nee || !d_gpp || !d_et || !d_stats) {
    setError("sipnet_batch_run_stats: needs the three planes and the statistics block");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return runImpl(b, step0, n_steps, d_nee, d_gpp, d_et, nullptr, nullptr, ld, hip_stream, d_stats);
}

int sipnet_batch_run_debug(sipnet_batch* b, int32_t step0, int32_t n_steps, double* d_rec,
                           double* d_dbg, int64_t ld, void* hip_stream) {
  if (!d_rec || !d_dbg) {
    setError("sipnet_batch_run_debug: needs both the record and the debug plane");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return runImpl(b, step0, n_steps, nullptr, nullptr, nullptr, d_rec, d_dbg, ld, hip_stream);
}

// Which cooperative kernel sums a batch's outputs over groups of steps inside its own launch (sipnet_batch_run_sums):
// throughput arithmetic (fp64 or fp32-mixed), any flag set, no record / diagnostics / full state -- AUTO's choice for the shape,
// or a throughput kernel forced.  0: none (the strict-order kernel).
static int sumsKernelFor(const sipnet_batch* b) {
  if (!b->fastMath || b->d_diag || (b->kernelOptions & SIPNET_KOPT_FULL_STATE)) return 0;
  int kernel = b->kernelPolicy;
  if (kernel == SIPNET_KERNEL_AUTO) kernel = autoKernel(b->flags, b->n_sites, b->n_members, true, false, 0, b->numCUs, false);
  if (kernel == SIPNET_KERNEL_ONE_WAVE) return kernel;
  const bool ncyc = b->flags[SIPNET_F_NITROGEN_CYCLE] != 0;
  if (ncyc) return (kernel == SIPNET_KERNEL_COOP_NCYCLE || kernel == SIPNET_KERNEL_COOP_NCYCLE_PAIR) ? kernel : 0;
  return (kernel == SIPNET_KERNEL_COOP_LDS || kernel == SIPNET_KERNEL_COOP_HBM || kernel == SIPNET_KERNEL_COOP_PAIR ||
          kernel == SIPNET_KERNEL_COOP_QUAD) ? kernel : 0;   // (a forced four-chunk layout the batch cannot take: the launch path says so)
}
int32_t sipnet_batch_sums_in_kernel(const sipnet_batch* b) { return b ? (sumsKernelFor(b) != 0) : 0; }

int sipnet_batch_run_sums(sipnet_batch* b, int32_t step0, int32_t n_steps, int32_t sum_steps, double* d_nee_sums, double* d_gpp_sums,
                          double* d_et_sums, int64_t ld, void* hip_stream) {
  if (!b || sum_steps <= 0) {
    setError("sipnet_batch_run_sums: sum_steps must be positive");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (!sumsKernelFor(b)) {
    setError("sipnet_batch_run_sums: no kernel sums this batch's outputs inside its launch (SIPNET_MATH_FAST, no diagnostics / "
             "full state: sipnet_batch_sums_in_kernel); run the planes and sum them");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return runImpl(b, step0, n_steps, d_nee_sums, d_gpp_sums, d_et_sums, nullptr, nullptr, ld, hip_stream, nullptr, sum_steps);
}

static int runImpl(sipnet_batch* b, int32_t step0, int32_t n_steps, void* d_nee, void* d_gpp,
                   void* d_et, double* d_rec, double* d_dbg, int64_t ld, void* hip_stream, double* d_stats, int32_t sumEvery) {
  if (!b || step0 < 0 || n_steps < 0 || step0 + n_steps > b->n_steps) {
    setError("sipnet_batch_run: step range outside the climate record");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if ((d_nee || d_gpp || d_et || d_rec) && ld < b->ncol) {
    setError("sipnet_batch_run: ld smaller than the number of columns");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->planDirty) {
    setError("sipnet_batch_run: call sipnet_batch_setup after changing climate/events");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (n_steps == 0) return SIPNET_OK;
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  rc = flushParams(b, stream);   // (parameters set after the last setup: a particle filter's, a re-draw)
  if (rc) return rc;
  KernelArgs a;
  a.plan = b->d_plan;
  a.ringOps = b->d_ringOps;
  a.events = b->d_events;
  a.siteBase = b->d_siteBase;
  a.prm = b->d_prm;
  a.state = b->d_state;
  a.ring = b->d_ring;
  a.nee = d_nee;
  a.gpp = d_gpp;
  a.et = d_et;
  a.rec = d_rec;
  a.dbg = d_dbg;
  a.diag = b->d_diag;
  a.ncol = b->ncol;
  a.ld = ld;
  a.n_sites = b->n_sites;
  a.n_members = b->n_members;
  a.n_steps_total = b->n_steps;
  a.step0 = step0;
  a.n_steps = n_steps;
  memcpy(a.flags, b->flags, sizeof(a.flags));
  // ---- kernel choice (sipnet_batch_set_kernel); nothing here reads the environment ----------
  const bool defaultFlags = isDefaultFlagSet(b->flags);
  int kernel = b->kernelPolicy;
  const bool wantFull = d_rec || b->d_diag || (b->kernelOptions & SIPNET_KOPT_FULL_STATE);
  if (kernel == SIPNET_KERNEL_AUTO) {
    kernel = autoKernel(b->flags, b->n_sites, b->n_members, b->fastMath, d_dbg != nullptr, b->d_diag ? 2 : wantFull ? 1 : 0, b->numCUs,
                        b->precision == SIPNET_F32_MIXED);
  } else if (kernel != SIPNET_KERNEL_STRICT) {
    if (!b->fastMath) {
      setError("sipnet_batch_run: the throughput kernels need SIPNET_MATH_FAST (sipnet_batch_set_math)");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    if (d_dbg) {
      setError("sipnet_batch_run_debug: the debug plane is written by the strict-order kernel only");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    if (kernel == SIPNET_KERNEL_COOP_NCYCLE || kernel == SIPNET_KERNEL_COOP_NCYCLE_PAIR) {
      if (!b->flags[SIPNET_F_NITROGEN_CYCLE]) {
        setError("sipnet_batch_run: the nitrogen-cycle cooperative kernels run flag sets with the nitrogen cycle on (records, "
                 "SIPNET_KOPT_FULL_STATE and the diagnostics counters included)");
        return SIPNET_ERR_BAD_ARGUMENT;
      }
    } else if (kernel != SIPNET_KERNEL_ONE_WAVE && b->flags[SIPNET_F_NITROGEN_CYCLE]) {
      setError("sipnet_batch_run: a flag set with the nitrogen cycle takes SIPNET_KERNEL_COOP_NCYCLE(_PAIR) or the one-wave kernel");
      return SIPNET_ERR_BAD_ARGUMENT;
    } else if (kernel == SIPNET_KERNEL_COOP_QUAD && !defaultFlags && b->precision != SIPNET_F32_MIXED) {
      setError("sipnet_batch_run: the optional-physics instantiations of the cooperative kernel carry one or two chunks per "
               "workgroup (four in an fp32-mixed batch only: the fp64 build would spill)");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    if (kernel == SIPNET_KERNEL_COOP_QUAD && wantFull) {
      setError("sipnet_batch_run: the four-chunk cooperative kernel has no full-state instantiation "
               "(records, diagnostics, SIPNET_KOPT_FULL_STATE)");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
  }
  // the throughput kernels index the ring [slot][col] with 32-bit element offsets (the strict-order kernel
  // uses 64-bit ones and takes any size)
  if (kernel != SIPNET_KERNEL_STRICT && b->ncol * SIPNET_RING_SLOTS >= (int64_t)1 << 31) {
    setError("sipnet_batch_run: the throughput kernels need n_sites * n_members * 250 < 2^31 (8.5 M columns per "
             "batch); split the ensemble into several batches or use SIPNET_MATH_STRICT");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  rc = kernel != SIPNET_KERNEL_STRICT ? ensureFastRecs(b, stream) : ensureStepRecs(b, stream);
  // (a member with dVpdExp != 2 has appeared since the device built its records: their log2vpd field, plan_device.h)
  if (!rc && kernel != SIPNET_KERNEL_STRICT && b->nDevSites && b->genericExponents && !b->devLog2Done) rc = fillDeviceLog2(b, stream);
  if (rc) return rc;
  // a resampled parameter index (particle filter): the one-wave kernel reads through it, every other kernel gets the
  // parameters back in column order first
  if (b->prmIndexed && kernel != SIPNET_KERNEL_ONE_WAVE) {
    rc = materializeParams(b, stream);
    if (rc) return rc;
  }
  // ensemble statistics with the launch (sipnet_batch_run_stats): a wavefront of the cooperative
  // kernel sums the planes' tiles per chunk while they are still in L2; any other kernel is
  // followed by three streaming reductions over the finished planes
  // (measured, DESIGN.md section 5: on the one-chunk-per-CU layout, whose fourth wavefront does the
  // summing, the launch grows by 3-5 % against 10-15 % for the three passes; on the two- / four-chunk
  // layouts the light wave would do it and its loads cost more than the passes -- SIPNET_KOPT_STATS_IN_KERNEL
  // forces it there for tests and measurements)
  const bool coop = kernel == SIPNET_KERNEL_COOP_LDS || kernel == SIPNET_KERNEL_COOP_PAIR ||
                    (kernel == SIPNET_KERNEL_COOP_QUAD && b->precision == SIPNET_F32_MIXED) ||
                    ((b->kernelOptions & SIPNET_KOPT_STATS_IN_KERNEL) && kernel != SIPNET_KERNEL_STRICT &&
                     kernel != SIPNET_KERNEL_ONE_WAVE && kernel != SIPNET_KERNEL_COOP_NCYCLE &&
                     kernel != SIPNET_KERNEL_COOP_NCYCLE_PAIR);
  const int chunksPerSite = (b->n_members + 63) / 64;
  if (d_stats && coop) {
    const size_t need = (size_t)3 * b->n_sites * chunksPerSite * n_steps * 2;
    if (need > b->statsPartCap) {
      HIP_TRY(hipStreamSynchronize(stream));
      if (b->d_statsPart) HIP_TRY(hipFree(b->d_statsPart));
      b->d_statsPart = nullptr;
      b->statsPartCap = 0;
      HIP_TRY(hipMalloc(&b->d_statsPart, need * sizeof(double)));
      b->statsPartCap = need;
    }
    // sites of different lengths: the rows past a site's last record are never written -- zero sums there
    bool ragged = false;
    for (int s = 0; s < b->n_sites; s++) ragged = ragged || b->siteSteps[s] != b->n_steps;
    if (ragged) HIP_TRY(hipMemsetAsync(b->d_statsPart, 0, need * sizeof(double), stream));
  }
  a.plan = b->d_plan;
  bool boundedWaits = false;
  // the timing events around the step kernel: for a long launch, or when asked for (sipnet_batch_time_next_launch) -- a
  // particle filter's 48-step forecasts run back to back with their analyses, and two event records per cycle cost the
  // device ~10 us of 165 (batch_impl.h markBusy)
  const bool timeIt = n_steps >= 512 || b->timeNext;
  b->timeNext = false;
  if (timeIt) HIP_TRY(hipEventRecord(b->ev0, stream));
  // log-weights a previous forecast left belong to the state BEFORE this launch, and an armed analysis belongs to THIS
  // launch, whichever kernel it takes
  const bool armed = b->pfArm.set;
  b->pfArm.set = false;
  b->pfPre.valid = false;
  if (kernel != SIPNET_KERNEL_STRICT) {
    // throughput path: step_fast.hip / step_coop.hip
    FastArgs f;
    f.fast = b->d_fast;
    f.ringOps = b->d_ringOps;
    f.events = b->d_events;
    f.siteBase = b->d_siteBase;
    // (a particle filter's batch after a resampling: the one-wave kernel reads the parameters through the particles' index --
    // into the batch's own block, or into the bank of all ranks' parameters of a connected filter)
    const bool throughIndex = b->prmIndexed && kernel == SIPNET_KERNEL_ONE_WAVE;
    f.prm = (throughIndex && b->d_prmBank) ? b->d_prmBank : b->d_prm;
    f.prmPitch = (throughIndex && b->d_prmBank) ? b->prmBankPitch : b->ncol;
    f.prmId = throughIndex ? b->d_prmId : nullptr;
    f.sumEvery = sumEvery;
    f.padEnd = 0;
    // a particle filter's forecast (sipnet_batch_pf_arm): the one-wave kernel's lean build leaves the log-weights too
    f.pfLogw = nullptr;
    f.pfBlockMax = nullptr;
    f.pfObs = f.pfInvSigma = 0.0;
    if (armed) {
      const int64_t blocks1 = (int64_t)b->n_sites * ((b->n_members + 63) / 64);
      bool sameLength = true;
      for (int s = 0; s < b->n_sites; s++) sameLength = sameLength && b->siteSteps[s] >= step0 + n_steps;
      if (kernel == SIPNET_KERNEL_ONE_WAVE && !wantFull && d_nee && sameLength) {
        if ((size_t)blocks1 > b->pfPreMaxCap) {
          if (b->d_pfPreMax) HIP_TRY(hipFree(b->d_pfPreMax));
          b->d_pfPreMax = nullptr;
          b->pfPreMaxCap = 0;
          HIP_TRY(hipMalloc((void**)&b->d_pfPreMax, (size_t)blocks1 * sizeof(double)));
          b->pfPreMaxCap = (size_t)blocks1;
        }
        f.pfLogw = b->pfArm.d_logw;
        f.pfBlockMax = b->d_pfPreMax;
        f.pfObs = b->pfArm.obs;
        f.pfInvSigma = 1.0 / b->pfArm.sigma;
        b->pfPre.valid = true;
        b->pfPre.plane = d_nee;
        b->pfPre.nSteps = n_steps;
        b->pfPre.nMax = (int32_t)blocks1;
        b->pfPre.ld = ld;
        b->pfPre.obs = b->pfArm.obs;
        b->pfPre.sigma = b->pfArm.sigma;
        b->pfPre.d_logw = b->pfArm.d_logw;
      }
    }
    f.state = b->d_state;
    f.ring = b->d_ring;
    f.nee = d_nee;
    f.gpp = d_gpp;
    f.et = d_et;
    f.ncol = b->ncol;
    f.ld = ld;
    f.n_sites = b->n_sites;
    f.n_members = b->n_members;
    f.n_steps_total = b->n_steps;
    f.step0 = step0;
    f.n_steps = n_steps;
    f.plainExp = b->genericExponents ? 0 : 1;
    f.rec = d_rec;
    f.diag = b->d_diag;
    f.full = wantFull ? 1 : 0;
    f.options = b->kernelOptions;
    f.scratchRow = b->d_scratchRow;
    memcpy(f.flags, b->flags, sizeof(f.flags));
    f.numCUs = b->numCUs;
    f.statsPart = (d_stats && coop) ? b->d_statsPart : nullptr;
    f.statsChunks = b->n_sites * chunksPerSite;
    const int layout = kernel == SIPNET_KERNEL_COOP_LDS ? COOP_RING_LDS
                       : kernel == SIPNET_KERNEL_COOP_PAIR ? COOP_PAIR
                       : kernel == SIPNET_KERNEL_COOP_QUAD ? COOP_QUAD
                       : kernel == SIPNET_KERNEL_COOP_NCYCLE ? COOP_NCYCLE
                       : kernel == SIPNET_KERNEL_COOP_NCYCLE_PAIR ? COOP_NCYCLE_PAIR : COOP_RING_HBM;
    boundedWaits = (b->kernelOptions & SIPNET_KOPT_BOUNDED_WAITS) && kernel != SIPNET_KERNEL_ONE_WAVE && !wantFull && !sumEvery;
    if (kernel == SIPNET_KERNEL_ONE_WAVE && sumEvery) sums2::launchStepFastSums(f, b->precision, b->kernelOptions, stream, &b->lastLaunch);
    else if (kernel == SIPNET_KERNEL_ONE_WAVE) launchStepFast(f, b->precision, b->kernelOptions, stream, &b->lastLaunch);
    else if (boundedWaits) bounded::launchStepCoop(f, b->precision, layout, stream, &b->lastLaunch);
    else if (sumEvery && (b->precision != SIPNET_F64 || layout == COOP_QUAD)) sums2::launchStepCoopSums(f, b->precision, layout, stream, &b->lastLaunch);
    else launchStepCoop(f, b->precision, layout, stream, &b->lastLaunch);
  } else {
    launchStep(a, b->precision, b->fastMath, stream, &b->lastLaunch);
  }
  HIP_TRY(hipGetLastError());
  if (timeIt) HIP_TRY(hipEventRecord(b->ev1, stream));
  if (d_stats) {
    if (coop) {
      launchFinishStats(b->d_statsPart, n_steps, b->n_sites, chunksPerSite, d_stats, stream);
    } else {
      const bool f32 = b->precision == SIPNET_F32_MIXED;
      const size_t plane = (size_t)n_steps * b->n_sites * 2;
      launchReducePlane(d_nee, f32, n_steps, ld, b->n_sites, b->n_members, d_stats, stream);
      launchReducePlane(d_gpp, f32, n_steps, ld, b->n_sites, b->n_members, d_stats + plane, stream);
      launchReducePlane(d_et, f32, n_steps, ld, b->n_sites, b->n_members, d_stats + 2 * plane, stream);
    }
    HIP_TRY(hipGetLastError());
  }
  b->timed = timeIt;
  b->stepsDone = (b->stepsDone == step0) ? step0 + n_steps : -1;
  rc = markBusy(b, stream);
  if (!rc && n_steps >= 512) rc = recordBusy(b);   // (a long launch: the event now, batch_impl.h markBusy)
  if (rc) return rc;
  if (boundedWaits) {   // the diagnostic build: did a hand-over wait give up?
    unsigned long long stuck[2] = {0, 0};
    if (bounded::readCoopStuck(stuck, stream) != 0) {
      setError("sipnet_batch_run: reading the bounded-wait report failed");
      return SIPNET_ERR_INTERNAL;
    }
    if (stuck[0] != 0) {
      setError("sipnet_batch_run: hand-over wait " + std::to_string((unsigned)((stuck[0] >> 32) & 0x7fffffffu)) + " (step_coop.hip, \"hand-over waits\") of workgroup " +
               std::to_string(stuck[1]) + " gave up at step " + std::to_string((int)(unsigned)(stuck[0] & 0xffffffffu)) + " of " + b->lastLaunch.kernel +
               ": a producer never posted (the launch's results are void)");
      return SIPNET_ERR_INTERNAL;
    }
  }
  return SIPNET_OK;
}

double sipnet_batch_last_kernel_ms(sipnet_batch* b) {
  if (!b || !b->timed) return -1.0;
  if (hipSetDevice(b->device) != hipSuccess) return -1.0;
  if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.0;
  b->lastMs = ms;
  return (double)ms;
}

int sipnet_batch_pf_arm(sipnet_batch* b, double obs, double sigma, double* d_logw) {
  if (!b || !d_logw || !(sigma > 0)) {
    setError("sipnet_batch_pf_arm: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  b->pfArm.set = true;
  b->pfArm.obs = obs;
  b->pfArm.sigma = sigma;
  b->pfArm.d_logw = d_logw;
  return SIPNET_OK;
}

int sipnet_batch_time_next_launch(sipnet_batch* b) {
  if (!b) return SIPNET_ERR_BAD_ARGUMENT;
  b->timeNext = true;
  return SIPNET_OK;
}

int sipnet_batch_last_launch(sipnet_batch* b, sipnet_launch_info* out) {
  if (!b || !out) return SIPNET_ERR_BAD_ARGUMENT;
  memset(out, 0, sizeof(*out));
  snprintf(out->kernel, sizeof out->kernel, "%s", b->lastLaunch.kernel);
  out->grid = b->lastLaunch.grid;
  out->block_threads = b->lastLaunch.block;
  out->waves_per_simd = b->lastLaunch.wavesPerSimd;
  out->lds_bytes = b->lastLaunch.ldsBytes;
  out->num_cus = b->numCUs;
  out->plan_threads = b->planThreads;
  out->plan_build_ms = b->planBuildMs;
  out->plan_upload_ms = b->planUploadMs;
  out->plan_device_sites = b->nDevSites;
  return SIPNET_OK;
}
const char* sipnet_batch_last_kernel_name(sipnet_batch* b) { return b ? b->lastLaunch.kernel : ""; }

int sipnet_batch_reduce_plane(sipnet_batch* b, const void* d_plane, int32_t elem_is_f32,
                              int32_t n_steps, int64_t ld, double* d_stats,
                              void* hip_stream) {
  if (!b || !d_plane || !d_stats || n_steps <= 0 || ld < b->ncol) {
    setError("sipnet_batch_reduce_plane: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  launchReducePlane(d_plane, elem_is_f32 != 0, n_steps, ld, b->n_sites, b->n_members,
                    d_stats, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return SIPNET_OK;
}

// state is exchanged with the host as [ncol][NSTATE]; HBM holds [NSTATE][ncol]
int sipnet_batch_get_state(sipnet_batch* b, double* state, void* hip_stream) {
  if (!b || !state) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipStreamSynchronize(stream));
  std::vector<double> tmp((size_t)b->ncol * SIPNET_NSTATE);
  HIP_TRY(hipMemcpy(tmp.data(), b->d_state, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int k = 0; k < SIPNET_NSTATE; k++)
    for (int64_t c = 0; c < b->ncol; c++)
      state[c * SIPNET_NSTATE + k] = tmp[(size_t)k * b->ncol + c];
  return SIPNET_OK;
}

int sipnet_batch_set_state(sipnet_batch* b, const double* state, void* hip_stream) {
  if (!b || !state) return SIPNET_ERR_BAD_ARGUMENT;
  b->pfPre.valid = false;
  b->pfArm.set = false;
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipStreamSynchronize(stream));
  std::vector<double> tmp((size_t)b->ncol * SIPNET_NSTATE);
  for (int k = 0; k < SIPNET_NSTATE; k++)
    for (int64_t c = 0; c < b->ncol; c++)
      tmp[(size_t)k * b->ncol + c] = state[c * SIPNET_NSTATE + k];
  HIP_TRY(hipMemcpy(b->d_state, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
  b->stepsDone = -1;  // the caller moved the state; only it knows to which record
  return SIPNET_OK;
}

int sipnet_batch_get_ring(sipnet_batch* b, int64_t col, double* values, void* hip_stream) {
  if (!b || !values || col < 0 || col >= b->ncol) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  return ringToHost(b, col, 1, values);
}

int sipnet_batch_get_rings(sipnet_batch* b, double* rings, void* hip_stream) {
  if (!b || !rings) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  std::vector<double> tmp((size_t)b->ncol * SIPNET_RING_SLOTS);
  rc = ringToHost(b, 0, b->ncol, tmp.data());
  if (rc) return rc;
  for (int k = 0; k < SIPNET_RING_SLOTS; k++)
    for (int64_t c = 0; c < b->ncol; c++)
      rings[c * SIPNET_RING_SLOTS + k] = tmp[(size_t)k * b->ncol + c];
  return SIPNET_OK;
}

int sipnet_batch_set_rings(sipnet_batch* b, const double* rings, void* hip_stream) {
  if (!b || !rings) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  std::vector<double> tmp((size_t)b->ncol * SIPNET_RING_SLOTS);
  for (int k = 0; k < SIPNET_RING_SLOTS; k++)
    for (int64_t c = 0; c < b->ncol; c++)
      tmp[(size_t)k * b->ncol + c] = rings[c * SIPNET_RING_SLOTS + k];
  return ringFromHost(b, 0, b->ncol, tmp.data());
}

int sipnet_batch_get_status(sipnet_batch* b, int32_t* status, void* hip_stream) {
  if (!b || !status) return SIPNET_ERR_BAD_ARGUMENT;
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  std::vector<double> tmp((size_t)b->ncol);
  HIP_TRY(hipMemcpy(tmp.data(), b->d_state + (size_t)ST_status * b->ncol,
                    tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t c = 0; c < b->ncol; c++) status[c] = (int32_t)tmp[c];
  return SIPNET_OK;
}

// ---- restart checkpoints ------------------------------------------------------
namespace {
constexpr double kTinyBiomass = 0.000001;  // common/util.h:14
constexpr double kRingWindow = 5.0;        // MEAN_NPP_DAYS, sipnet.c:39

bool sufficientBiomass(const double* envi) {  // hasSufficientBiomass(), sipnet.c:1530-1537
  return envi[0] > kTinyBiomass && envi[0] + envi[12] > kTinyBiomass &&
         envi[7] + envi[6] > kTinyBiomass;
}
int nextSlot(int i) { return (i + 1) % SIPNET_RING_SLOTS; }
int prevSlot(int i) { return (i + SIPNET_RING_SLOTS - 1) % SIPNET_RING_SLOTS; }

bool sameLayout(const sipnet_restart& r, const RingSched& s) {
  if (r.mean_start != s.start || r.mean_last != s.last) return false;
  for (int i = s.start;; i = nextSlot(i)) {
    if (r.mean_weights[i] != s.w[i]) return false;
    if (i == s.last) break;
  }
  return true;
}

// Re-express a member's ring on the site's layout: entries are matched newest first; the
// member's remaining (older) entries must all hold zero, which any layout represents.
bool relayRing(const sipnet_restart& r, const RingSched& s, double* values) {
  for (int i = 0; i < SIPNET_RING_SLOTS; i++) values[i] = 0.0;
  int im = r.mean_last, is = s.last;
  for (;;) {
    bool restZero = true;
    for (int k = r.mean_start;; k = nextSlot(k)) {
      if (r.mean_values[k] != 0.0) restZero = false;
      if (k == im) break;
    }
    if (restZero) return true;
    if (r.mean_weights[im] != s.w[is]) return false;
    values[is] = r.mean_values[im];
    if (im == r.mean_start) return true;
    if (is == s.start) return false;
    im = prevSlot(im);
    is = prevSlot(is);
  }
}
}  // namespace

int sipnet_batch_set_resume(sipnet_batch* b, int32_t site, const sipnet_restart* r) {
  if (!b || site < 0 || site >= b->n_sites) {
    setError("sipnet_batch_set_resume: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  PlanCarry& c = b->resume[site];
  c = PlanCarry{};
  b->resumeProcessed[site] = 0;
  b->planDirty = true;
  if (!r) return SIPNET_OK;
  if (r->mean_length != SIPNET_RING_SLOTS || r->mean_start < 0 ||
      r->mean_start >= SIPNET_RING_SLOTS || r->mean_last < 0 ||
      r->mean_last >= SIPNET_RING_SLOTS) {  // restart.c:727-733, :987-992
    setError("Restart mean-tracker length or cursor out of range");
    return SIPNET_ERR_RESTART;
  }
  c.set = true;
  b->resumeProcessed[site] = r->processed_steps;  // the count runs on, restart.c:912-920
  c.gdd = r->trackers[SIPNET_RT_GDD];
  c.trackLastYear = r->trackers_last_year;
  c.phenLastYear = r->phenology_last_year;
  c.dTill = r->d_till_mod;
  c.ring.start = r->mean_start;
  c.ring.last = r->mean_last;
  for (int i = 0; i < SIPNET_RING_SLOTS; i++) {
    c.ring.w[i] = r->mean_weights[i];
    c.ring.insStep[i] = 0;
  }
  return SIPNET_OK;
}

int sipnet_batch_import_restart(sipnet_batch* b, int32_t site, int32_t first_member,
                                int32_t count, const sipnet_restart* r, void* hip_stream) {
  if (!b || site < 0 || site >= b->n_sites || first_member < 0 || count <= 0 ||
      first_member + count > b->n_members || !r) {
    setError("sipnet_batch_import_restart: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->planDirty || !b->resume[site].set) {
    setError("sipnet_batch_import_restart: call sipnet_batch_set_resume and "
             "sipnet_batch_setup first");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  b->pfPre.valid = false;
  b->pfArm.set = false;
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  const PlanCarry& c = b->resume[site];
  const int64_t col0 = (int64_t)site * b->n_members + first_member;
  const size_t pitchD = (size_t)b->ncol * sizeof(double), pitchH = (size_t)count * sizeof(double);
  // current state block [NSTATE][count]: keeps what setup decided (status, diagnostics)
  std::vector<double> st((size_t)SIPNET_NSTATE * count), ring((size_t)SIPNET_RING_SLOTS * count);
  HIP_TRY(hipMemcpy2D(st.data(), pitchH, b->d_state + col0, pitchD, pitchH, SIPNET_NSTATE,
                      hipMemcpyDeviceToHost));
  double vals[SIPNET_RING_SLOTS];
  for (int32_t m = 0; m < count; m++) {
    const sipnet_restart& k = r[m];
    const std::string who = "member " + std::to_string(first_member + m) + " of site " +
                            std::to_string(site);
    for (int i = 0; i < SIPNET_NFLAGS; i++) {
      if (k.flags[i] != b->flags[i]) {
        setError("Restart context mismatch: model flags must match checkpoint exactly (" + who + ")");
        return SIPNET_ERR_RESTART;
      }
    }
    if (k.trackers[SIPNET_RT_GDD] != c.gdd || k.trackers_last_year != c.trackLastYear ||
        k.phenology_last_year != c.phenLastYear || k.d_till_mod != c.dTill) {
      setError("sipnet_batch_import_restart: " + who + " disagrees with the site's resume "
               "state (gdd, lastYear or d_till_mod); members of a site share one forcing history");
      return SIPNET_ERR_RESTART;
    }
    if ((k.is_alive != 0) != sufficientBiomass(k.envi)) {
      setError("sipnet_batch_import_restart: survival.isAlive of " + who +
               " contradicts its pools (sipnet.c:1530-1544)");
      return SIPNET_ERR_RESTART;
    }
    if (k.mean_length != SIPNET_RING_SLOTS || k.mean_start < 0 ||
        k.mean_start >= SIPNET_RING_SLOTS || k.mean_last < 0 ||
        k.mean_last >= SIPNET_RING_SLOTS) {
      setError("Restart mean-tracker length or cursor out of range (" + who + ")");
      return SIPNET_ERR_RESTART;
    }
    if (sameLayout(k, c.ring)) {
      memcpy(vals, k.mean_values, sizeof vals);
    } else if (!relayRing(k, c.ring, vals)) {
      setError("sipnet_batch_import_restart: running-mean ring layout of " + who +
               " cannot be expressed on the site's layout");
      return SIPNET_ERR_RESTART;
    }
    for (int i = 0; i < SIPNET_RING_SLOTS; i++) ring[(size_t)i * count + m] = vals[i];
    double* s = st.data() + m;
    auto S = [&](int row) -> double& { return s[(size_t)row * count]; };
    for (int i = 0; i < 13; i++) S(i) = k.envi[i];
    S(ST_ringSum) = k.mean_sum;
    S(ST_totGpp) = k.trackers[SIPNET_RT_TOTGPP];
    S(ST_totRtot) = k.trackers[SIPNET_RT_TOTRTOT];
    S(ST_totRa) = k.trackers[SIPNET_RT_TOTRA];
    S(ST_totRh) = k.trackers[SIPNET_RT_TOTRH];
    S(ST_totNpp) = k.trackers[SIPNET_RT_TOTNPP];
    S(ST_totNee) = k.trackers[SIPNET_RT_TOTNEE];
    S(ST_yearlyGpp) = k.trackers[SIPNET_RT_YEARLYGPP];
    S(ST_yearlyRtot) = k.trackers[SIPNET_RT_YEARLYRTOT];
    S(ST_yearlyRa) = k.trackers[SIPNET_RT_YEARLYRA];
    S(ST_yearlyRh) = k.trackers[SIPNET_RT_YEARLYRH];
    S(ST_yearlyNpp) = k.trackers[SIPNET_RT_YEARLYNPP];
    S(ST_yearlyNee) = k.trackers[SIPNET_RT_YEARLYNEE];
    S(ST_yearlyLitter) = k.trackers[SIPNET_RT_YEARLYLITTER];
    S(ST_phenBits) = (double)((k.did_leaf_growth ? 1 : 0) | (k.did_leaf_fall ? 2 : 0));
    S(ST_ringValidFrom) = 0.0;
  }
  HIP_TRY(hipMemcpy2D(b->d_state + col0, pitchD, st.data(), pitchH, pitchH, SIPNET_NSTATE,
                      hipMemcpyHostToDevice));
  return ringFromHost(b, col0, count, ring.data());
}

int sipnet_batch_export_restart(sipnet_batch* b, int32_t site, int32_t member,
                                int32_t n_steps_done, const double* last_rec,
                                const double* prev_pools, sipnet_restart* out,
                                void* hip_stream) {
  if (!b || site < 0 || site >= b->n_sites || member < 0 || member >= b->n_members || !out) {
    setError("sipnet_batch_export_restart: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->planDirty) {
    setError("sipnet_batch_export_restart: no run to take a checkpoint of");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (n_steps_done <= 0 || n_steps_done > b->siteSteps[site]) {  // restart.c:933-937
    setError("Cannot write restart checkpoint: no timestep processed");
    return SIPNET_ERR_RESTART;
  }
  // (a site shorter than the batch's longest stops at its own last record)
  if (b->stepsDone >= 0 && std::min(b->stepsDone, b->siteSteps[site]) != n_steps_done) {
    setError("sipnet_batch_export_restart: the carried state is at record " +
             std::to_string(b->stepsDone) + ", not " + std::to_string(n_steps_done));
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  const int64_t col = (int64_t)site * b->n_members + member;
  const size_t pitchD = (size_t)b->ncol * sizeof(double);
  double st[SIPNET_NSTATE], ring[SIPNET_RING_SLOTS];
  HIP_TRY(hipMemcpy2D(st, sizeof(double), b->d_state + col, pitchD, sizeof(double),
                      SIPNET_NSTATE, hipMemcpyDeviceToHost));
  rc = ringToHost(b, col, 1, ring);
  if (rc) return rc;
  if ((int)st[ST_status] != SIPNET_OK) {
    setError("sipnet_batch_export_restart: member did not run (status " +
             std::to_string((int)st[ST_status]) + ")");
    return (int)st[ST_status];
  }

  // what the plan owns, after n_steps_done records
  const int n = n_steps_done;
  // (cached: a CLI exporting every member of a site asks for the same boundary again and again)
  if (b->exportCacheSite != site || b->exportCacheN != n) {
    b->exportHead = buildSitePlan(
        b->flags, n, b->sc[site].clim(), b->sc[site].year(), b->sc[site].day(),
        (int32_t)b->events[site].size(), b->events[site].data(),
        b->resume[site].set ? &b->resume[site] : nullptr, &b->exportFin);
    b->exportCacheSite = site;
    b->exportCacheN = n;
  }
  const PlanCarry& fin = b->exportFin;
  const SitePlan& head = b->exportHead;
  const double* lastClim = b->sc[site].clim() + (size_t)SIPNET_NCLIM * (n - 1);

  memset(out, 0, sizeof(*out));
  snprintf(out->model_version, sizeof out->model_version, "2.1.0");
  {
    std::string info = sipnet_version();
    for (char& ch : info)
      if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') ch = '_';
    snprintf(out->build_info, sizeof out->build_info, "%s", info.c_str());
  }
  out->checkpoint_utc_epoch = (int64_t)time(nullptr);
  out->processed_steps = b->resumeProcessed[site] + n;
  memcpy(out->flags, b->flags, sizeof out->flags);
  out->boundary_year = b->sc[site].year()[n - 1];
  out->boundary_day = b->sc[site].day()[n - 1];
  out->boundary_time = lastClim[10];
  out->boundary_length = lastClim[0];
  for (int i = 0; i < 13; i++) out->envi[i] = st[i];
  double* T = out->trackers;
  if (last_rec) {  // the per-step trackers of the last record, sipnet.c:1433-1497
    T[SIPNET_RT_GPP] = last_rec[1];
    T[SIPNET_RT_RTOT] = last_rec[10];
    T[SIPNET_RT_RA] = last_rec[8];
    T[SIPNET_RT_RH] = last_rec[9];
    T[SIPNET_RT_RROOT] = last_rec[7];
    T[SIPNET_RT_RSOIL] = last_rec[6];
    T[SIPNET_RT_RABOVEGROUND] = last_rec[5];
    T[SIPNET_RT_NPP] = last_rec[4];
    T[SIPNET_RT_NEE] = last_rec[0];
    T[SIPNET_RT_WOODCREATION] = last_rec[11];
    T[SIPNET_RT_ET] = last_rec[2];
    T[SIPNET_RT_SOILWETNESSFRAC] = last_rec[12];
    T[SIPNET_RT_METHANE] = last_rec[31];
    T[SIPNET_RT_N2O] = last_rec[27];
    T[SIPNET_RT_NLEACHING] = last_rec[28];
    T[SIPNET_RT_NFIXATION] = last_rec[29];
    T[SIPNET_RT_NUPTAKE] = last_rec[30];
    T[SIPNET_RT_MEANNPP] = last_rec[32];
  } else {
    T[SIPNET_RT_MEANNPP] = st[ST_ringSum] / kRingWindow;
  }
  T[SIPNET_RT_GDD] = fin.gdd;
  T[SIPNET_RT_YEARLYGPP] = st[ST_yearlyGpp];
  T[SIPNET_RT_YEARLYRTOT] = st[ST_yearlyRtot];
  T[SIPNET_RT_YEARLYRA] = st[ST_yearlyRa];
  T[SIPNET_RT_YEARLYRH] = st[ST_yearlyRh];
  T[SIPNET_RT_YEARLYNPP] = st[ST_yearlyNpp];
  T[SIPNET_RT_YEARLYNEE] = st[ST_yearlyNee];
  T[SIPNET_RT_YEARLYLITTER] = st[ST_yearlyLitter];
  T[SIPNET_RT_TOTGPP] = st[ST_totGpp];
  T[SIPNET_RT_TOTRTOT] = st[ST_totRtot];
  T[SIPNET_RT_TOTRA] = st[ST_totRa];
  T[SIPNET_RT_TOTRH] = st[ST_totRh];
  T[SIPNET_RT_TOTNPP] = st[ST_totNpp];
  T[SIPNET_RT_TOTNEE] = st[ST_totNee];
  out->trackers_last_year = fin.trackLastYear;
  const int phenBits = (int)st[ST_phenBits];
  out->did_leaf_growth = phenBits & 1;
  out->did_leaf_fall = (phenBits >> 1) & 1;
  out->phenology_last_year = fin.phenLastYear;
  out->is_alive = sufficientBiomass(out->envi) ? 1 : 0;
  out->d_till_mod = fin.dTill;
  // harvest fractions of the last record's events (events.c:467-469, :553-562)
  if (prev_pools && b->flags[SIPNET_F_EVENTS]) {
    const StepRec& ls = head.steps[n - 1];
    const double woodC = prev_pools[0] + prev_pools[12];
    const double above = woodC + prev_pools[1], below = prev_pools[7] + prev_pools[6];
    for (int e = 0; e < ls.evCount; e++) {
      const EvRec& ev = head.events[ls.evFirst + e];
      if (ev.type == SIPNET_EV_HARVEST && above + below > kTinyBiomass) {
        out->harvest_frac_removed += (ev.p[0] * above + ev.p[1] * below) / (above + below);
        out->harvest_frac_transferred += (ev.p[2] * above + ev.p[3] * below) / (above + below);
      }
    }
  }

  // running-mean ring in the reference's own layout
  out->mean_length = SIPNET_RING_SLOTS;
  out->mean_tot_weight = kRingWindow;
  out->mean_sum = st[ST_ringSum];
  const int validFrom = (int)st[ST_ringValidFrom];
  if (validFrom <= 0) {
    // a member that never died holds exactly the plan's ring
    out->mean_start = fin.ring.start;
    out->mean_last = fin.ring.last;
    for (int i = 0; i < SIPNET_RING_SLOTS; i++) {
      out->mean_weights[i] = fin.ring.w[i];
      out->mean_values[i] = ring[i];
    }
  } else {
    // the reference reset this member's ring when it died (sipnet.c:1757) and inserted
    // again from record validFrom on: replay that schedule, take the values by insert step
    RingSched fresh;
    bool overflow = false;
    for (int t = validFrom; t < n; t++)
      fresh.advance(t, b->sc[site].clim()[(size_t)SIPNET_NCLIM * t], nullptr, &overflow);
    out->mean_start = fresh.start;
    out->mean_last = fresh.last;
    for (int i = 0; i < SIPNET_RING_SLOTS; i++) out->mean_weights[i] = fresh.w[i];
    for (int i = fresh.start;; i = nextSlot(i)) {
      const int ins = fresh.insStep[i];
      if (ins >= validFrom) {
        for (int j = 0; j < SIPNET_RING_SLOTS; j++) {
          if (fin.ring.insStep[j] == ins) {
            out->mean_values[i] = ring[j];
            break;
          }
        }
      }
      if (i == fresh.last) break;
    }
  }
  return SIPNET_OK;
}

int64_t sipnet_batch_ncol(const sipnet_batch* b) { return b ? b->ncol : 0; }
int32_t sipnet_batch_nsteps(const sipnet_batch* b) { return b ? b->n_steps : 0; }
int32_t sipnet_batch_site_nsteps(const sipnet_batch* b, int32_t site) {
  return (b && site >= 0 && site < b->n_sites) ? b->sc[site].n : 0;
}

int sipnet_batch_get_site_series(sipnet_batch* b, int32_t site, double* gdd,
                                 double* d_till_mod) {
  if (!b || site < 0 || site >= b->n_sites || b->planDirty ||
      (int)b->plans.size() != b->n_sites)
    return SIPNET_ERR_BAD_ARGUMENT;
  if (b->devSite[site] && b->plans[site].gddAfter.empty()) {   // a device-built site: the series from a host pass of its own
    const SiteClim& c = b->sc[site];
    SitePlan hp = buildSitePlan(b->flags, c.n, c.clim(), c.year(), c.day(), (int32_t)b->events[site].size(), b->events[site].data(),
                                b->resume[site].set ? &b->resume[site] : nullptr, nullptr, /*wantSteps=*/false);
    b->plans[site].gddAfter = std::move(hp.gddAfter);
    b->plans[site].dTill = std::move(hp.dTill);
  }
  const SitePlan& p = b->plans[site];
  for (int t = 0; t < b->siteSteps[site]; t++) {   // (the site's own length: sipnet_batch_nsteps is the longest site's)
    if (gdd) gdd[t] = p.gddAfter[t];
    if (d_till_mod) d_till_mod[t] = p.dTill[t];
  }
  return SIPNET_OK;
}

/* Test hook: the records and ring evictions the DEVICE built for `site` (plan_device.h) against buildSitePlan()'s on the
 * host, byte by byte.  ignore_log2 != 0: FastRec::log2vpd is left out (it is only filled when a member reads it). */
int sipnet_debug_plan_compare(sipnet_batch* b, int32_t site, int32_t ignore_log2, int64_t* n_records_differing,
                              int64_t* n_ops_differing, int32_t* first_step, int32_t* first_offset, int32_t* device_info) {
  if (!b || site < 0 || site >= b->n_sites || b->planDirty || !b->devSite[site]) {
    setError("sipnet_debug_plan_compare: not a device-built site");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  const SiteClim& c = b->sc[site];
  const int n = c.n;
  std::vector<FastRec> host(n), dev(n);
  SitePlan hp = buildSitePlan(b->flags, n, c.clim(), c.year(), c.day(), (int32_t)b->events[site].size(), b->events[site].data(),
                              b->resume[site].set ? &b->resume[site] : nullptr, nullptr, /*wantSteps=*/false, nullptr, host.data(),
                              b->precision == SIPNET_F32_MIXED);
  HIP_TRY(hipMemcpy(dev.data(), b->d_fast + (size_t)site * b->n_steps, (size_t)n * sizeof(FastRec), hipMemcpyDeviceToHost));
  int d = 0;
  for (int s = 0; s < site; s++) d += b->devSite[s];
  int32_t out4[8];
  HIP_TRY(hipMemcpy(out4, b->devPlan.siteOut + 8 * d, sizeof out4, hipMemcpyDeviceToHost));
  if (device_info) memcpy(device_info, out4, sizeof out4);
  int64_t nr = 0, no = 0;
  int32_t fs = -1, fo = -1;
  for (int t = 0; t < n; t++) {
    if (ignore_log2) dev[t].log2vpd = host[t].log2vpd;
    if (memcmp(&host[t], &dev[t], sizeof(FastRec)) != 0) {
      if (fs < 0) {
        fs = t;
        const unsigned char *x = (const unsigned char*)&host[t], *y = (const unsigned char*)&dev[t];
        for (size_t k = 0; k < sizeof(FastRec); k++)
          if (x[k] != y[k]) { fo = (int32_t)k; break; }
      }
      nr++;
    }
  }
  std::vector<RingOp> ops(hp.ringOps.size() + 1);
  std::vector<int32_t> base(3);
  HIP_TRY(hipMemcpy(base.data(), b->d_siteBase + 3 * site, 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (!hp.ringOps.empty())
    HIP_TRY(hipMemcpy(ops.data(), b->d_ringOps + base[0], hp.ringOps.size() * sizeof(RingOp), hipMemcpyDeviceToHost));
  for (size_t k = 0; k < hp.ringOps.size(); k++)
    if (memcmp(&ops[k], &hp.ringOps[k], sizeof(RingOp)) != 0) no++;
  if ((size_t)out4[1] != hp.ringOps.size()) no += 1 + llabs((long long)out4[1] - (long long)hp.ringOps.size());
  // (the event records the light pass matched to the climate records)
  const std::vector<EvRec>& evs = b->plans[site].events;
  if (evs.size() != hp.events.size() || (!evs.empty() && memcmp(evs.data(), hp.events.data(), evs.size() * sizeof(EvRec)) != 0)) no += 1000000;
  if (n_records_differing) *n_records_differing = nr;
  if (n_ops_differing) *n_ops_differing = no;
  if (first_step) *first_step = fs;
  if (first_offset) *first_offset = fo;
  return SIPNET_OK;
}

void* sipnet_dev_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    setError("sipnet_dev_alloc: hipMalloc failed");
    return nullptr;
  }
  return p;
}
void sipnet_dev_free(void* p) {
  if (p) (void)hipFree(p);
}
int sipnet_dev_to_host(void* host, const void* dev, size_t bytes, void* hip_stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return SIPNET_OK;
}
int sipnet_dev_to_host_2d(void* host, size_t host_pitch, const void* dev, size_t dev_pitch, size_t width_bytes, size_t rows,
                          void* hip_stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  HIP_TRY(hipMemcpy2D(host, host_pitch, dev, dev_pitch, width_bytes, rows, hipMemcpyDeviceToHost));
  return SIPNET_OK;
}
int sipnet_dev_to_dev_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows,
                         void* hip_stream) {
  if (!dst || !src || rows == 0 || width_bytes == 0) {
    setError("sipnet_dev_to_dev_2d: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  // a kernel of our own for the case that matters (8-byte elements): the runtime's 2-D copy moved 17 520 rows of 80 KB at
  // 0.5 GB/s (3 s per column of c10k's record), this streams them at the HBM rate
  if (((width_bytes | dst_pitch | src_pitch | (size_t)(uintptr_t)dst | (size_t)(uintptr_t)src) & 7) == 0) {
    const size_t w8 = width_bytes / 8;
    const dim3 grid((unsigned)((w8 + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535));
    hipLaunchKernelGGL(copyRows8Kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, (uint64_t*)dst, dst_pitch / 8,
                       (const uint64_t*)src, src_pitch / 8, w8, rows);
    HIP_TRY(hipGetLastError());
    return SIPNET_OK;
  }
  HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  return SIPNET_OK;
}
int sipnet_stream_sync(void* hip_stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  return SIPNET_OK;
}
void* sipnet_stream_create(int32_t device) {
  hipStream_t s = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    setError("sipnet_stream_create: hipStreamCreate failed");
    return nullptr;
  }
  return (void*)s;
}
void sipnet_stream_destroy(void* hip_stream) {
  if (hip_stream) (void)hipStreamDestroy((hipStream_t)hip_stream);
}

}  // extern "C"

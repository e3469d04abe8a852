// batch_impl.h -- the batch object behind include/sipnet_amd.h (shared by engine.hip and pf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <string>
#include <memory>
#include <vector>

#include "../../include/sipnet_amd.h"
#include "plan.h"
#include "plan_device.h"
#include "step_kernel.h"

namespace sipnet {
void setError(const std::string& s);
}
using namespace sipnet;  // internal header: only engine.hip and pf.hip include it

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      setError(std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return SIPNET_ERR_NO_DEVICE;                                            \
    }                                                                         \
  } while (0)

struct PfScratch;  // pf.hip
struct PfPeers;

// A site's forcing as the caller handed it over: one PINNED host block ([n][NCLIM] doubles, then year[n], day[n]), kept so
// that the plan can be rebuilt in any call order and so that the block can leave for the device by an asynchronous copy
// straight out of sipnet_batch_set_climate (device-built plans, plan_device.h); grow-only.
struct SiteClim {
  unsigned char* host = nullptr;
  unsigned char* dev = nullptr;
  size_t hostCap = 0, devCap = 0;   // bytes
  int32_t n = 0;                    // records
  bool onDevice = false;            // the device block holds this forcing (a copy is at least queued on upStream)
  hipEvent_t evCopied = nullptr;    // behind that copy: the host may write the pinned block again once it has fired
  bool copyQueued = false;
  static size_t bytesFor(int32_t n) { return (size_t)n * (SIPNET_NCLIM * sizeof(double) + 2 * sizeof(int32_t)); }
  const double* clim() const { return (const double*)host; }
  const int32_t* year() const { return (const int32_t*)(host + (size_t)n * SIPNET_NCLIM * sizeof(double)); }
  const int32_t* day() const { return year() + n; }
  const double* devClim() const { return (const double*)dev; }
  const int32_t* devYear() const { return (const int32_t*)(dev + (size_t)n * SIPNET_NCLIM * sizeof(double)); }
  const int32_t* devDay() const { return devYear() + n; }
};

struct sipnet_batch {
  int32_t flags[SIPNET_NFLAGS];
  int32_t n_sites = 0, n_members = 0, precision = 0, device = 0;
  int32_t numCUs = 256;
  int64_t ncol = 0;
  int32_t n_steps = 0;  // records of the longest site (the stride of the per-step records; sites may be shorter: siteSteps)
  std::vector<int32_t> siteSteps;   // per site: its number of climate records
  bool fastMath = false;
  int32_t kernelPolicy = SIPNET_KERNEL_AUTO, kernelOptions = 0;
  LaunchInfo lastLaunch{};
  int32_t planThreads = 0;
  double planBuildMs = 0.0, planUploadMs = 0.0;
  bool genericExponents = false;  // some member has dVpdExp != 2 or soilRespMoistEffect != 1

  // host-side inputs kept so the plan can be rebuilt in any call order
  std::vector<SiteClim> sc;                    // per site: climate [n_steps][NCLIM], year, day (pinned)
  std::vector<std::vector<sipnet_event>> events;
  std::vector<SitePlan> plans;
  std::vector<PlanCarry> resume;  // per site: state the plan starts from (restart)
  std::vector<int64_t> resumeProcessed;  // per site: meta_info.processed_steps resumed from
  std::vector<int32_t> siteStatus;
  bool planDirty = true;
  int32_t stepsDone = 0;       // records the carried state reflects, -1 = unknown
  // device-built site plans (plan_device.h): which sites, the kernels' arguments (scratch carved out of one block)
  std::vector<uint8_t> devSite;
  int32_t nDevSites = 0;
  DevPlanArgs devPlan{};
  unsigned char* d_planScratch = nullptr;
  size_t planScratchCap = 0;
  double* d_devLog2 = nullptr;   // [nDevSites][n_steps] inside d_planScratch
  double* hostLog2 = nullptr;    // pinned staging of the host-computed log2(vpd)
  size_t hostLog2Cap = 0;
  double* hostGdd = nullptr;     // pinned [n_sites][n_steps]: trackers.gdd after every record, the plan threads' chain (engine.hip deviceEligible)
  size_t hostGddCap = 0;
  unsigned char* hostEv = nullptr;   // pinned, only while a site has events: per site evFirst[n_steps] evCount[n_steps] (int32), dTill[n_steps] tillAfter[n_steps]
  size_t hostEvCap = 0;
  std::vector<PlanLight> planLight;  // per site: the plan threads' pass before the records are built (engine.hip devicePrepass)
  bool devLog2Done = false;
  int32_t devPlanMaxSteps = 0;
  hipEvent_t evPlanDone = nullptr;   // behind the plan kernels: what the next forcing's climate copy waits for
  bool planKernelsQueued = false;

  // HBM
  double* d_rawStage = nullptr;  // [rawStageCap][NPARAMS] raw rows of the set_params calls since the last launch
  size_t rawStageCap = 0;
  // set_params is host-only: the raw rows go into PINNED staging and a list of pending conversions; the next
  // stream-taking entry point (setup, run, the particle filter's) uploads and converts them on ITS stream -- no
  // kernel of this batch then runs, and nothing of it blocks, while another batch's step kernel fills the device
  // (a conversion kernel on a stream of its own, and synchronous copies, were measured stuck behind that kernel)
  double* hostRaw = nullptr;     // pinned [hostRawCap][NPARAMS]
  size_t hostRawCap = 0, hostRawUsed = 0;
  struct PendingParams { size_t row0; int64_t col0; int32_t count, nRep; };
  std::vector<PendingParams> pendingParams;
  double* d_prm = nullptr;     // [NPARAMS][ncol] converted parameters: the one copy on the device
  double* d_state = nullptr;   // [NSTATE][ncol]
  double* d_ring = nullptr;    // [RING_SLOTS][ncol] doubles; fp32-mixed batches: floats (ringElemBytes)
  // second copies for particle-filter resampling (gather into the spare, then swap); lazily made
  double* d_prm2 = nullptr;
  // Particle filter (round 5): converted parameters are read-only during a forecast, so a resampling of a filter whose
  // particles carry their parameters need not MOVE 640 bytes per particle -- it moves an index.  prmIndexed: column c's
  // parameters are column d_prmId[c] of d_prm (the parameter BANK: a row set once by set_params, never copied);
  // the one-wave step kernel dereferences the index at launch start (FastArgs::prmId), every other reader of d_prm
  // first gets the bank gathered back into column order (materializeParams).  Not indexed: d_prmId is unused.
  int32_t* d_prmId = nullptr;
  int32_t* d_prmId2 = nullptr;
  bool prmIndexed = false;
  // A filter spread over ranks whose particles carry their parameters (round 6): every rank holds a copy of ALL ranks'
  // converted parameters, [NPARAMS][world * nmax] (slot = rank * nmax + particle; sipnet_batch_pf_connect fills it once
  // from the peers' blocks), and a particle's d_prmId is such a slot for as long as the connection lasts -- a resampling
  // across ranks moves 4 bytes of index per particle instead of 640 bytes of rows, and the forecast reads local HBM.
  // While d_prmBank is set, d_prmId / d_prmId2 are always maintained (peers read them); prmIndexed then only says that
  // d_prm, the column-order copy every kernel but the one-wave kernel reads, is behind the index.
  double* d_prmBank = nullptr;
  int64_t prmBankPitch = 0;
  // how many filters may run their one-launch analysis on this device at the same time (a node's shards on one device):
  // the spinning grid is sized to 1 / deviceShare of what the device holds (pf.hip fusedBudget)
  int32_t deviceShare = 1;
  int32_t pfSpinBudget = 0;      // polls a barrier waits before it declares the launch void (0: SIPNET_PF_SPIN_BUDGET)
  int32_t pfDebugAbsent = -1;    // test hook: this workgroup of the NEXT fused launch leaves without arriving
  struct PfInfo { int32_t fused = 0, grid = 0, budget = 0; int64_t nSlots = 0, cycles = 0; } pfInfo;
  unsigned long long* d_pfCrossing = nullptr;   // particles this rank has copied from ANOTHER rank's slot, all cycles
  double* d_state2 = nullptr;
  double* d_ring2 = nullptr;
  StepRec* d_plan = nullptr;   // [n_sites][n_steps]
  FastRec* d_fast = nullptr;   // [n_sites][n_steps] + kFastTile padding records
  double* d_scratchRow = nullptr;  // [ncol]
  RingOp* d_ringOps = nullptr;
  EvRec* d_events = nullptr;
  int32_t* d_siteStatus = nullptr;
  double* d_statsPart = nullptr;     // [3][chunks][n_steps of the launch][2]: per-chunk plane statistics
  size_t statsPartCap = 0;           //   (sipnet_batch_run_stats on a cooperative kernel), doubles
  double* d_diag = nullptr;          // [4][ncol] per-member diagnostics, allocated on request
  SiteStart* d_siteStart = nullptr;  // [n_sites] what setupModel() reads of a site's first record
  size_t planCap = 0, fastCap = 0, ringOpCap = 0, evCap = 0;
  // host staging of the flat per-step records, kept between hand-overs of a forcing: a fresh buffer
  // costs its first touch (143 MB at c4: 24 ms of page faults, more than building the records)
  // (pinned: the sites' records leave by hipMemcpyAsync on the setup's stream as the worker threads finish them)
  FastRec* hostFast = nullptr;
  StepRec* hostSteps = nullptr;
  size_t hostFastCap = 0, hostStepsCap = 0;
  unsigned char* hostMisc = nullptr;   // pinned staging of the small per-plan arrays (ring evictions, events, site tables)
  size_t hostMiscCap = 0;
  int32_t* d_siteBase = nullptr;  // [n_sites][3]: offset of a site's ring ops / events in the flat arrays, its number of records
  bool stepRecsUploaded = false, fastRecsUploaded = false;  // per-step records: uploaded on first use
  // last boundary a checkpoint was exported at (sipnet_batch_export_restart)
  int32_t exportCacheSite = -1, exportCacheN = -1;
  SitePlan exportHead;
  PlanCarry exportFin;

  // particle filter (pf.hip): resampling scratch owned by the batch, and -- between sipnet_batch_pf_connect and
  // destroy -- the table of the peers' checkpoint matrices a cross-rank resampling reads from
  PfScratch* pfScratch = nullptr;
  PfPeers* pfPeers = nullptr;

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  // sipnet_batch_pf_arm: the next lean one-wave launch also leaves the log-weights of its NEE sum (FastArgs::pfLogw);
  // pfPre describes what such a launch has left, for the analysis that follows it
  struct PfArm { bool set = false; double obs = 0.0, sigma = 0.0; double* d_logw = nullptr; } pfArm;
  struct PfPre { bool valid = false; const void* plane = nullptr; int32_t nSteps = 0, nMax = 0; int64_t ld = 0; double obs = 0.0, sigma = 0.0;
                 double* d_logw = nullptr; } pfPre;
  double* d_pfPreMax = nullptr;   // [workgroups of the one-wave launch]: their columns' largest log-weight
  size_t pfPreMaxCap = 0;
  bool timeNext = false;   // sipnet_batch_time_next_launch: the next step kernel is bracketed by the timing events whatever its length
  double lastMs = -1.0;
  // recorded behind every launch of this batch that reads its inputs (setupModel, a step kernel): what an upload
  // that overwrites those inputs waits for -- this batch's own work only, not the device (another batch's kernel
  // may be running: a caller pipelines forcings over two batches, bench.py's end_to_end.pipelined)
  hipEvent_t evBusy = nullptr;
  bool busy = false;
  hipStream_t busyStream = nullptr;   // where the last launch went; evBusy is recorded there on demand (markBusy / recordBusy below)
  bool busyRecorded = true;
  hipEvent_t evStaged = nullptr;   // behind the last copy out of the pinned staging blocks
  hipEvent_t evOrder = nullptr;    // caller's stream -> copy stream ordering
  bool staged = false;
  // uploads that need a kernel (the parameter conversion) run on a stream of the batch's own: on the null stream
  // they would queue behind whatever another batch is running on a blocking stream
  hipStream_t upStream = nullptr;
};
int flushParams(sipnet_batch* b, hipStream_t stream);   // engine.hip: upload + convert what set_params left pending
int materializeParams(sipnet_batch* b, hipStream_t stream);   // pf.hip: d_prm back into column order (no-op unless prmIndexed)
void pfDropBank(sipnet_batch* b);   // pf.hip: a connected filter's bank of all ranks' parameters is void (new parameters, moved rows)

// "This batch has work in flight on `stream`."  The event itself is recorded only when somebody needs it (a wait from
// another stream, a host-side wait or query): work queued later on the SAME stream is ordered behind it anyway, and an
// event record between two kernels costs the device ~5 us of an otherwise back-to-back dispatch -- two of them per
// 165 us particle-filter cycle (profiles/r05_c5.md: the gaps around the forecast kernel).  Long launches (a setup, a run of
// 512 steps or more) record it at once (recordBusy): there it costs nothing, and a later "is it still running?" gets an
// answer that does not depend on how fast a fresh event completes.
inline int markBusy(sipnet_batch* b, hipStream_t stream) {
  // (launches move to another stream: the old stream's work gets its event now -- the callers have ordered the new
  // stream behind it through orderBehindBusy, and the next event will be the new stream's)
  if (b->busy && !b->busyRecorded && b->busyStream != stream) (void)hipEventRecord(b->evBusy, b->busyStream);
  (void)hipGetLastError();
  b->busyStream = stream;
  b->busy = true;
  b->busyRecorded = false;
#ifdef SIPNET_EAGER_BUSY   // (A/B build: the event after every launch, as before round 5)
  HIP_TRY(hipEventRecord(b->evBusy, stream));
  b->busyRecorded = true;
#endif
  return SIPNET_OK;
}
// the event behind the batch's last launch, recorded if it has not been yet.  A stream the caller has destroyed in the
// meantime cannot be recorded on: whatever ran on it is waited for through the device instead.
inline int recordBusy(sipnet_batch* b) {
  if (!b->busy || b->busyRecorded) return SIPNET_OK;
  if (hipEventRecord(b->evBusy, b->busyStream) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(hipDeviceSynchronize());
    b->busy = false;
  }
  b->busyRecorded = true;
  return SIPNET_OK;
}
inline int waitIdle(sipnet_batch* b) {
  if (b->busy) {
    int rc = recordBusy(b);
    if (rc) return rc;
    if (b->busy) HIP_TRY(hipEventSynchronize(b->evBusy));
  }
  b->busy = false;
  return SIPNET_OK;
}
// is the batch's last launch still running?  An event recorded only NOW (after a short launch, see markBusy) takes the
// device a few microseconds even on an idle stream: it is given 40 before the batch is called busy.  (hipStreamQuery
// instead of an event was tried: on this runtime it returned only when the stream had drained -- 7.5 ms inside
// sipnet_batch_set_climate_sites of the pipelined whole-job leg, tools/pipeline_probe.py.)
inline bool stillRunning(sipnet_batch* b) {
  if (!b->busy) return false;
  const bool fresh = !b->busyRecorded;
  if (recordBusy(b) != SIPNET_OK || !b->busy) return false;
  hipError_t q = hipEventQuery(b->evBusy);
  if (fresh && q == hipErrorNotReady) {
    const auto t0 = std::chrono::steady_clock::now();
    while (q == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(40)) q = hipEventQuery(b->evBusy);
  }
  (void)hipGetLastError();
  if (q == hipSuccess) b->busy = false;   // (nothing of this batch is in flight any more)
  return q == hipErrorNotReady;
}
// `stream` will touch what the batch's last launch reads or writes: a device-side wait, unless that launch went to the
// same stream (then the stream's own order does it)
inline int orderBehindBusy(sipnet_batch* b, hipStream_t stream) {
  if (!b->busy || (!b->busyRecorded && b->busyStream == stream)) return SIPNET_OK;
  int rc = recordBusy(b);
  if (rc) return rc;
  if (b->busy) HIP_TRY(hipStreamWaitEvent(stream, b->evBusy, 0));
  return SIPNET_OK;
}
// the pinned staging blocks (raw parameter rows, site records, the small plan arrays): free for the host to write
// again once the copies out of them have completed -- long before the kernels that read the device side have
inline int markStaged(sipnet_batch* b, hipStream_t stream) {
  HIP_TRY(hipEventRecord(b->evStaged, stream));
  b->staged = true;
  return SIPNET_OK;
}
inline int waitStaged(sipnet_batch* b) {
  if (b->staged) HIP_TRY(hipEventSynchronize(b->evStaged));
  b->staged = false;
  return SIPNET_OK;
}

// bytes of a ring element: fp32-mixed batches keep the running-mean ring in fp32 (its values are NPP rates such
// a batch computes in fp32)
inline size_t ringElemBytes(const sipnet_batch* b) {
  return b->precision == SIPNET_F32_MIXED ? sizeof(float) : sizeof(double);
}

void pfRelease(sipnet_batch* b);   // pf.hip: frees pfScratch / pfPeers (called by sipnet_batch_destroy, device current)

inline int useDevice(const sipnet_batch* b) {
  HIP_TRY(hipSetDevice(b->device));
  return SIPNET_OK;
}


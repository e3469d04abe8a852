// step_kernel.h -- the SIPNET per-timestep state update as a CDNA4 HIP kernel.
//
// One thread = one ensemble member; one 64-thread workgroup (one wavefront) =
// 64 consecutive members of ONE site, so every climate-driven branch is
// wave-uniform and the site plan (plan.h) arrives through scalar loads.  The
// time loop runs inside the kernel with the member's pools, accumulators and
// converted parameters held in registers; per step a thread touches HBM only
// for its running-mean ring slot(s) ([slot][col], coalesced) and its outputs
// ([t][col], coalesced).
//
// The arithmetic follows updateState() of the reference,
// /root/reference/src/sipnet/sipnet.c:1818-1855 (citations below are relative to
// /root/reference/src/).  Two math policies:
//   Strict : operation order, true divisions and pow/exp calls as the reference
//            writes them (differences vs glibc are the <=1-2 ulp of OCML's
//            pow/exp/exp2);
//   Fast   : member-independent sub-expressions come from the site plan,
//            x/length becomes x*invLen, pow(q,T/10) becomes exp2(T/10*log2 q)
//            with log2 q hoisted out of the time loop, the Simpson layers reuse
//            one exp.  Same model, ~1e-15 relative differences.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace sipnet {

enum ParamIndex : int {
#define SIPNET_PARAM(idx, field, fname, rule) SP_##field = idx,
#include "../../include/sipnet_params.def"
#undef SIPNET_PARAM
  SP_COUNT
};
static_assert(SP_COUNT == SIPNET_NPARAMS, "parameter table size");

// state vector rows (include/sipnet_amd.h)
enum StateIndex : int {
  ST_plantWoodC = 0, ST_plantLeafC, ST_soilC, ST_soilWater, ST_litterC, ST_snow,
  ST_coarseRootC, ST_fineRootC, ST_minN, ST_soilOrgN, ST_litterN,
  ST_plantStorageN, ST_plantCAccountingDelta,
  ST_ringSum = 13,
  ST_totGpp = 14, ST_totRtot, ST_totRa, ST_totRh, ST_totNpp, ST_totNee,
  ST_yearlyGpp = 20, ST_yearlyRtot, ST_yearlyRa, ST_yearlyRh, ST_yearlyNpp,
  ST_yearlyNee, ST_yearlyLitter,
  ST_phenBits = 27, ST_ringValidFrom = 28, ST_status = 29, ST_diedAt = 30,
  ST_clampCount = 31
};
static_assert(ST_clampCount + 1 == SIPNET_NSTATE, "state vector size");

struct KernelArgs {
  const StepRec* plan;    // [n_sites][n_steps_total]
  const RingOp* ringOps;  // all sites; StepRec.ringOpFirst is site-local, siteBase[3*site] its base
  const EvRec* events;    // all sites; StepRec.evFirst is site-local, siteBase[3*site+1] its base
  const int32_t* siteBase;  // [n_sites][3]: ring-op base, event base, the site's number of records (<= n_steps_total)
  const double* prm;      // [SIPNET_NPARAMS][ncol] converted parameters
  double* state;          // [SIPNET_NSTATE][ncol]
  double* ring;           // [SIPNET_RING_SLOTS][ncol] of the kernel's real type (floats for fp32-mixed batches)
  void* nee;              // [n_steps][ld] or null
  void* gpp;
  void* et;
  double* rec;            // [n_steps][SIPNET_NREC][ld] or null
  double* dbg;            // [n_steps][SIPNET_NDBG][ld] or null (only with rec)
  double* diag;           // [4][ncol] diagnostics (see FastArgs) or null
  int64_t ncol, ld;
  int32_t n_sites, n_members, n_steps_total, step0, n_steps;
  int32_t flags[SIPNET_NFLAGS];
};

// what setupModel() reads of a site's first climate record (phenology state, sipnet.c:1501-1527)
struct SiteStart {
  double cumGdd, tsoil, dayTime;
};
struct SetupArgs {
  const SiteStart* siteStart;  // [n_sites]
  const double* prm;      // [SIPNET_NPARAMS][ncol] converted parameters (launchConvertParams)
  double* state;          // [SIPNET_NSTATE][ncol]
  double* ring;           // slot 0 row is zeroed
  int32_t ringF32;        // fp32-mixed batches: the ring is [SIPNET_RING_SLOTS][ncol] floats (see sipnet_batch)
  int64_t ncol;
  int32_t n_sites, n_members;
  int32_t flags[SIPNET_NFLAGS];
  const int32_t* siteStatus;  // [n_sites] plan status (site-fatal conditions)
};

// arguments of the throughput kernels (step_fast.hip, step_coop.hip): lean outputs
struct FastArgs {
  const FastRec* fast;    // [n_sites][n_steps_total] (+ kFastTile records of padding)
  const RingOp* ringOps;      // all sites; FastRec.opFirst / evFirst are site-local ...
  const EvRec* events;
  const int32_t* siteBase;    // ... [n_sites][3]: the site's base in ringOps / events, its number of records (<= n_steps_total)
  const double* prm;
  double* state;
  double* ring;
  void* nee;
  void* gpp;
  void* et;
  int64_t ncol, ld;
  int32_t n_sites, n_members, n_steps_total, step0, n_steps;
  // "full" launches (the Full instantiations): every accumulator of the restart schema advances,
  // and optionally the per-step record and the per-member diagnostics are written
  double* rec;       // [n_steps][SIPNET_NREC][ld] or null
  double* diag;      // [4][ncol]: clamp warnings, balance warnings, max|dC|, max|dN|; or null
  int32_t full;      // 1: take the Full instantiation
  int32_t options;   // SIPNET_KOPT_* bits the kernels themselves look at
  void* scratchRow;  // [ncol] doubles: target of the stores of planes the caller left NULL
  int32_t plainExp;  // 1: every member has dVpdExp == 2 and soilRespMoistEffect == 1
  int32_t numCUs;                // compute units of the device (kernel / occupancy choice)
  int32_t flags[SIPNET_NFLAGS];  // model flags; anything but the default set selects the
                                 // run-time-flag instantiation of the one-wave kernel
  // cooperative kernels only: per-chunk ensemble statistics of the three planes, written by the
  // light wave from the planes' freshly stored tiles (still in L2) --
  // statsPart[((plane * statsChunks + site * chunksPerSite + chunk) * n_steps + (t - step0)) * 2 + {0, 1}]
  // = sum / sum of squares over the chunk's members of plane[t]; null: not wanted.  All three
  // planes must be given.  launchFinishStats adds the chunks of a site up.
  double* statsPart;
  int32_t statsChunks;           // n_sites * chunksPerSite
  // (at the END, so that the fields above keep the offsets the cooperative kernels' code was measured with: a field in the
  // middle re-rolled c10k's register allocation and cost it 1 %)
  const int32_t* prmId;   // null, or (one-wave kernel only; a particle filter's batch): column c reads the parameters of column prmId[c]
  // one-wave kernel, lean launches, a particle filter's forecast (sipnet_batch_pf_arm): the launch also leaves every
  // column's log-weight of the NEE it has summed over its steps -- -0.5 ((sum_t nee[t] - pfObs) pfInvSigma)^2 in the
  // arithmetic of pf.hip's logWeightOf, -inf for a member that did not run -- and the maximum of each workgroup's 64
  // columns: the analysis then starts at its second phase (no pass over the plane, one grid barrier less)
  double* pfLogw;         // [ncol] or null
  double* pfBlockMax;     // [workgroups of the launch]
  double pfObs, pfInvSigma;
  int64_t prmPitch;       // one-wave kernel: columns of prm (ncol; a filter's parameter bank shared by all ranks: world * nmax)
  int32_t sumEvery;       // > 0 (the Sums builds of the cooperative and the one-wavefront kernels): nee / gpp / et receive DOUBLE sums over groups of this many steps
  int32_t padEnd;
};
// What a launcher actually put on the stream (sipnet_batch_last_launch): the instantiation's
// name as rocprofv3 prints its template arguments, and the launch shape.
struct LaunchInfo {
  char kernel[96];
  int32_t grid, block;      // workgroups, threads per workgroup
  int32_t wavesPerSimd;     // resident wavefronts per SIMD the instantiation's register budget allows
  int32_t ldsBytes;         // static LDS per workgroup
};
// options: SIPNET_KOPT_* bits of include/sipnet_amd.h
void launchStepFast(const FastArgs& a, int precision, int options, hipStream_t stream, LaunchInfo* info);
// three cooperating wavefronts per 64 members (step_coop.hip); same results as launchStepFast
enum CoopLayout { COOP_RING_LDS = 0, COOP_RING_HBM = 1, COOP_PAIR = 2, COOP_QUAD = 3, COOP_NCYCLE = 4, COOP_NCYCLE_PAIR = 5 };  // step_coop.hip
void launchStepCoop(const FastArgs& a, int precision, int layout, hipStream_t stream, LaunchInfo* info);
namespace bounded {   // step_coop_bounded.hip: the same with bounded hand-over waits (SIPNET_KOPT_BOUNDED_WAITS)
void launchStepCoop(const FastArgs& a, int precision, int layout, hipStream_t stream, LaunchInfo* info);
int readCoopStuck(unsigned long long out[2], hipStream_t stream);
}
namespace sums2 {     // step_coop_sums.hip: the in-launch sums (FastArgs::sumEvery) of fp32-mixed batches, and of fp64 ones on the four-chunk layout
void launchStepCoopSums(const FastArgs& a, int precision, int layout, hipStream_t stream, LaunchInfo* info);
void launchStepFastSums(const FastArgs& a, int precision, int options, hipStream_t stream, LaunchInfo* info);   // step_fast_sums.hip: the one-wavefront kernels'
}
// the flag sets the throughput kernels have compiled in; events, gdd and soil_phenol may take any (legal) value
// in both -- they only change what the plan puts into the records (step_kernel.hip)
bool isPhenologyOrEventsFlag(int flag);
bool isDefaultFlagSet(const int32_t* flags);
bool isNCycleFlagSet(const int32_t* flags);   // defaults + litter pool + anaerobic + nitrogen cycle (step_fast.hip)

// launchers (step_kernel.hip)
void launchSetup(const SetupArgs& a, hipStream_t stream);
// raw parameter rows [count][SIPNET_NPARAMS] (DEVICE staging) -> columns col0.. of prm[NPARAMS][ncol]
// leafOnMode: 0 gdd, 1 soil_phenol, 2 neither (day of year) -- with 1 and 2 the converted gddLeafOn row receives
// the threshold of THAT mode (soilTempLeafOn; leafOnDay, +inf when non-positive: sipnet.c:705-731), which is what
// the throughput kernels compare the plan's leaf-on variable with; the strict kernel reads the original rows
void launchConvertParams(const double* rawRows, double* prm, int64_t ncol, int64_t col0,
                         int32_t count, int32_t leafOnMode, hipStream_t stream, int32_t nRep = 1,
                         int64_t repStride = 0);
// variant: bit0 = fast math, bit1 = generic flags (runtime), else default flags
void launchStep(const KernelArgs& a, int precision, bool fastMath, hipStream_t stream, LaunchInfo* info);
// stats[((plane * n_steps + t) * n_sites + site) * 2 + {0, 1}] = sum over the site's chunks of statsPart
void launchFinishStats(const double* statsPart, int32_t n_steps, int32_t n_sites, int32_t chunksPerSite,
                       double* stats, hipStream_t stream);
void launchReducePlane(const void* plane, bool isF32, int32_t n_steps, int64_t ld,
                       int32_t n_sites, int32_t n_members, double* stats,
                       hipStream_t stream);

}  // namespace sipnet

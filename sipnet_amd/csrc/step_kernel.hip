// step_kernel.hip -- CDNA4 (gfx950) kernels of the SIPNET flux-integration engine.
// See step_kernel.h for the execution model.  Compiled with -ffp-contract=off:
// every fused multiply-add in here is written explicitly.
#include "step_kernel.h"

#include <hip/hip_runtime.h>

#include <cstdio>

namespace sipnet {

namespace {

// ---- constants: sipnet/sipnet.c:33-49, common/util.h:14, sipnet/balance.h:6
constexpr double kTiny = 0.000001;
constexpr double kEps = 1e-8;
constexpr double kCWeight = 12.0, kTen9 = 1000000000.0, kSecPerDay = 86400.0;
constexpr double kMeanNppDays = 5.0;
constexpr double kLambda = 2501000., kLambdaS = 2835000., kRho = 1.3, kCp = 1005.,
                 kGamma = 66., kEStarSnow = 0.6;
constexpr int kNumLayers = 6;  // sipnet.c:524

// ---- math policies ---------------------------------------------------------
template <class R> __device__ __forceinline__ R rmin(R a, R b) { return a < b ? a : b; }
// fmin/fmax semantics of the reference for non-NaN operands
__device__ __forceinline__ double dmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float dmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float dmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double rexp(double x) { return exp(x); }
__device__ __forceinline__ float rexp(float x) { return __expf(x); }
__device__ __forceinline__ double rexp2(double x) { return exp2(x); }
__device__ __forceinline__ float rexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double rlog2(double x) { return log2(x); }
__device__ __forceinline__ float rlog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ double rpow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float rpow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double rabs(double x) { return fabs(x); }
template <class R> __device__ __forceinline__ R unitClip(R x) {  // util.h:38
  return dmin(dmax(x, R(0)), R(1));
}

// ---- configuration of one kernel instantiation ------------------------------
template <class Real, bool Fast, bool Generic, bool FullRec>
struct Cfg {
  using real = Real;
  static constexpr bool fast = Fast;
  static constexpr bool generic = Generic;
  static constexpr bool fullRec = FullRec;
};

// default model flags: common/context.c:35-53
__host__ __device__ __forceinline__ constexpr bool defaultFlag(int f) {
  return f == SIPNET_F_EVENTS || f == SIPNET_F_GDD || f == SIPNET_F_SNOW ||
         f == SIPNET_F_WATER_HRESP;
}

template <class C>
struct Flags {
  bool events, gdd, growthResp, leafWater, litterPool, soilPhenol, waterHResp,
      nitrogenCycle, anaerobic, flooding, carbonSaturation;
  __device__ explicit Flags(const int32_t* f) {
    auto get = [&](int i) { return C::generic ? (f[i] != 0) : defaultFlag(i); };
    events = get(SIPNET_F_EVENTS);
    gdd = get(SIPNET_F_GDD);
    growthResp = get(SIPNET_F_GROWTH_RESP);
    leafWater = get(SIPNET_F_LEAF_WATER);
    litterPool = get(SIPNET_F_LITTER_POOL);
    soilPhenol = get(SIPNET_F_SOIL_PHENOL);
    waterHResp = get(SIPNET_F_WATER_HRESP);
    nitrogenCycle = get(SIPNET_F_NITROGEN_CYCLE);
    anaerobic = get(SIPNET_F_ANAEROBIC);
    flooding = get(SIPNET_F_FLOODING);
    carbonSaturation = get(SIPNET_F_CARBON_SATURATION);
  }
};

// XCD-aware (site, chunk) assignment.  Workgroups are dealt round-robin over
// the 8 XCDs, so blocks b and b+8 share an L2; keep all chunks of a site on
// one XCD group so that the site's plan records are fetched into one L2 only.
// Placement changes speed, never results.
__device__ __forceinline__ void blockToSiteChunk(int b, int n_sites, int chunksPerSite,
                                                 int& site, int& chunk) {
  if ((n_sites & 7) == 0) {
    const int g = b & 7, j = b >> 3;
    site = g + 8 * (j / chunksPerSite);
    chunk = j % chunksPerSite;
  } else {
    site = b / chunksPerSite;
    chunk = b % chunksPerSite;
  }
}

// scalar (wave-uniform) loads of plan data
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// -----------------------------------------------------------------------------
// setupModel(), sipnet.c:1858-1951 (+ :1111-1123, :1406-1413, :1501-1527), in two kernels:
//   convertParamsKernel  the parameter half (unit conversions, derived parameters, clamps), run
//                        when parameters are uploaded: raw rows (AoS, as in the file) -> the
//                        converted SoA block, which is from then on the ONLY copy of a member's
//                        parameters on the device (a particle-filter resampling permutes it);
//   setupKernel          the state half (initial pools, trackers, phenology state from the first
//                        climate record, ring reset), from the converted block.
// One thread per member; both run once, so plain AoS reads are fine.
// -----------------------------------------------------------------------------
// (one thread per member and PARAMETER ROW, blockIdx.y = row: a handful of registers, so that an upload's conversion
// finds room on a device another batch's step kernel is filling -- with one thread per member and the 80 values in
// registers it had to wait for that kernel to end, which serialised a pipeline of forcings over two batches)
__global__ __launch_bounds__(256) void convertParamsKernel(const double* __restrict__ raw,
                                                           double* __restrict__ prm, int64_t ncol,
                                                           int64_t col0, int32_t count, int32_t leafOnMode,
                                                           int32_t nRep, int64_t repStride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (int)blockIdx.y;
  if (i >= count) return;
  const double* __restrict__ r = raw + i * SIPNET_NPARAMS;
  double v = r[k];
  switch (k) {
    // ensureAllocation, sipnet.c:1111-1123 (the validity test itself is in setupKernel)
    case SP_coarseRootAllocation: v = 1 - r[SP_leafAllocation] - r[SP_woodAllocation] - r[SP_fineRootAllocation]; break;
    // per-year -> per-day, sipnet.c:1873-1877, :1898-1902
    case SP_baseVegResp: case SP_litterBreakdownRate: case SP_baseSoilResp: case SP_woodTurnoverRate:
    case SP_leafTurnoverRate: case SP_fineRootTurnoverRate: case SP_coarseRootTurnoverRate:
    case SP_baseCoarseRootResp: case SP_baseFineRootResp: v /= 365.0; break;
    case SP_psnTMax: v = r[SP_psnTOpt] + (r[SP_psnTOpt] - r[SP_psnTMin]); break;
    // sipnet.c:1905-1916
    case SP_fAnoxia: v = v <= 0.0 ? kTiny : (v >= 1.0 ? 1.0 - kTiny : v); break;
    case SP_anaerobicDecompRate: v = v <= 0.0 ? kTiny : (v > 1.0 ? 1.0 : v); break;
    // the throughput kernels' leaf-on threshold (step_kernel.h): unused as a GDD sum in these modes
    case SP_gddLeafOn:
      if (leafOnMode == 1) v = r[SP_soilTempLeafOn];
      if (leafOnMode == 2) v = r[SP_leafOnDay] > 0 ? r[SP_leafOnDay] : 1e300;
      break;
    default: break;
  }
  // (nRep > 1: the same members at nRep sites, repStride columns apart -- SIPNET_ALL_SITES)
  for (int rep = 0; rep < nRep; rep++) prm[(int64_t)k * ncol + col0 + (int64_t)rep * repStride + i] = v;
}

__global__ __launch_bounds__(256) void setupKernel(SetupArgs a) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.ncol) return;
  const int site = (int)(col / a.n_members);
  const SiteStart s0 = a.siteStart[site];
  const double* __restrict__ pp = a.prm + col;
#define P(name) pp[(int64_t)SP_##name * a.ncol]

  int status = a.siteStatus[site];
  // ensureAllocation, sipnet.c:1111-1123
  if ((P(leafAllocation) >= 1.0) || (P(woodAllocation) >= 1.0) ||
      (P(fineRootAllocation) >= 1.0) || (P(coarseRootAllocation) < 0)) {
    if (status == 0) status = SIPNET_ERR_BAD_PARAMETER;
  }

  double st[SIPNET_NSTATE];
#pragma unroll
  for (int k = 0; k < SIPNET_NSTATE; k++) st[k] = 0.0;
  // pools, sipnet.c:1884-1940
  st[ST_plantWoodC] = (1 - P(coarseRootFrac) - P(fineRootFrac)) * P(plantWoodInit);
  st[ST_plantLeafC] = P(laiInit) * P(leafCSpWt);
  st[ST_litterC] = a.flags[SIPNET_F_LITTER_POOL] ? P(litterInit) : 0.0;
  st[ST_soilC] = P(soilInit);
  st[ST_coarseRootC] = P(coarseRootFrac) * P(plantWoodInit);
  st[ST_fineRootC] = P(fineRootFrac) * P(plantWoodInit);
  double sw = P(soilWFracInit) * P(soilWHC);
  if (sw < 0) sw = 0;
  st[ST_soilWater] = sw;
  st[ST_snow] = P(snowInit);
  if (a.flags[SIPNET_F_NITROGEN_CYCLE]) {
    st[ST_minN] = P(minNInit);
    st[ST_soilOrgN] = P(soilOrgNInit);
    st[ST_litterN] = P(litterOrgNInit);
    st[ST_plantStorageN] = P(plantStorageNInit);
  }
  // phenology state from the first record, sipnet.c:1501-1527 with :705-742
  int grew = 0, fell = 0;
  if (a.flags[SIPNET_F_GDD]) {
    grew = (s0.cumGdd >= P(gddLeafOn));
  } else if (a.flags[SIPNET_F_SOIL_PHENOL]) {
    grew = (s0.tsoil >= P(soilTempLeafOn));
  } else if (P(leafOnDay) > 0) {
    grew = (s0.dayTime >= P(leafOnDay));
  }
  if (P(leafOffDay) > 0) {
    fell = (s0.dayTime >= P(leafOffDay));
  }
  if (fell && !grew) grew = 1;
  st[ST_phenBits] = (double)(grew | (fell << 1));
  st[ST_ringValidFrom] = 0.0;
  st[ST_status] = (double)status;
  st[ST_diedAt] = -1.0;
#undef P

#pragma unroll
  for (int k = 0; k < SIPNET_NSTATE; k++) a.state[(int64_t)k * a.ncol + col] = st[k];
  // slot 0 = the initial (mean 0, weight 5) entry
  if (a.ringF32) ((float*)a.ring)[col] = 0.0f;
  else a.ring[col] = 0.0;
}

// -----------------------------------------------------------------------------
// the step kernel
// -----------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(64) void stepKernel(KernelArgs a) {
  using R = typename C::real;
  const Flags<C> F(a.flags);

  const int chunksPerSite = (a.n_members + 63) >> 6;
  int site, chunk;
  blockToSiteChunk((int)blockIdx.x, a.n_sites, chunksPerSite, site, chunk);
  const int m = (chunk << 6) + (int)threadIdx.x;
  if (m >= a.n_members) return;  // no barriers below: a lane may simply leave
  const int64_t col = (int64_t)site * a.n_members + m;
  // ring-eviction and event indices in the site's records are local to the site
  const int opBase = uni(a.siteBase[3 * site]), evBase = uni(a.siteBase[3 * site + 1]);
  const int siteSteps = uni(a.siteBase[3 * site + 2]);   // records of THIS site (sites of a batch may differ in length)
  const int64_t nc = a.ncol;

  double* __restrict__ stp = a.state + col;
  if (stp[(int64_t)ST_status * nc] != 0.0) return;  // member skipped, see setup

  // ---- parameters -> registers (converted by setupKernel) --------------------
  const double* __restrict__ pp = a.prm + col;
#define PRM(name) ((R)pp[(int64_t)SP_##name * nc])
  const R aMax = PRM(aMax), aMaxFrac = PRM(aMaxFrac), baseFolRespFrac = PRM(baseFolRespFrac);
  const R psnTMin = PRM(psnTMin), psnTOpt = PRM(psnTOpt), psnTMax = PRM(psnTMax);
  const R dVpdSlope = PRM(dVpdSlope), dVpdExp = PRM(dVpdExp);
  const R halfSatPar = PRM(halfSatPar), attenuation = PRM(attenuation);
  const R leafOnDay = PRM(leafOnDay), leafOffDay = PRM(leafOffDay), gddLeafOn = PRM(gddLeafOn);
  const R baseVegResp = PRM(baseVegResp), vegRespQ10 = PRM(vegRespQ10);
  const R baseSoilResp = PRM(baseSoilResp), soilRespQ10 = PRM(soilRespQ10);
  const R waterRemoveFrac = PRM(waterRemoveFrac), wueConst = PRM(wueConst);
  const R soilWHC = PRM(soilWHC), leafCSpWt = PRM(leafCSpWt), cFracLeaf = PRM(cFracLeaf);
  const R woodTurnoverRate = PRM(woodTurnoverRate), leafTurnoverRate = PRM(leafTurnoverRate);
  const R frozenSoilEff = PRM(frozenSoilEff), frozenSoilFolREff = PRM(frozenSoilFolREff);
  const R frozenSoilThreshold = PRM(frozenSoilThreshold);
  const R immedEvapFrac = PRM(immedEvapFrac), fastFlowFrac = PRM(fastFlowFrac);
  const R snowMeltP = PRM(snowMelt), rdConst = PRM(rdConst);
  const R rSoilConst1 = PRM(rSoilConst1), rSoilConst2 = PRM(rSoilConst2);
  const R leafAllocation = PRM(leafAllocation), woodAllocation = PRM(woodAllocation);
  const R fineRootAllocation = PRM(fineRootAllocation), coarseRootAllocation = PRM(coarseRootAllocation);
  const R fineRootTurnoverRate = PRM(fineRootTurnoverRate), coarseRootTurnoverRate = PRM(coarseRootTurnoverRate);
  const R baseFineRootResp = PRM(baseFineRootResp), baseCoarseRootResp = PRM(baseCoarseRootResp);
  const R fineRootQ10 = PRM(fineRootQ10), coarseRootQ10 = PRM(coarseRootQ10);
  const R leafGrowth = PRM(leafGrowth), fracLeafFall = PRM(fracLeafFall);
  const R soilRespMoistEffect = PRM(soilRespMoistEffect), leafOnReallocFrac = PRM(leafOnReallocFrac);
  // optional-flag parameters (only read by generic instantiations)
  const R soilTempLeafOn = C::generic ? PRM(soilTempLeafOn) : R(0);
  const R growthRespFrac = C::generic ? PRM(growthRespFrac) : R(0);
  const R leafPoolDepth = C::generic ? PRM(leafPoolDepth) : R(0);
  const R waterDrainFrac = C::generic ? PRM(waterDrainFrac) : R(0);
  const R litterBreakdownRate = C::generic ? PRM(litterBreakdownRate) : R(0);
  const R fracLitterRespired = C::generic ? PRM(fracLitterRespired) : R(0);
  const R nVolatilizationFrac = C::generic ? PRM(nVolatilizationFrac) : R(0);
  const R nLeachingFrac = C::generic ? PRM(nLeachingFrac) : R(0);
  const R leafCN = C::generic ? PRM(leafCN) : R(1);
  const R woodCN = C::generic ? PRM(woodCN) : R(1);
  const R fineRootCN = C::generic ? PRM(fineRootCN) : R(1);
  const R kCN = C::generic ? PRM(kCN) : R(0);
  const R nFixationFracMax = C::generic ? PRM(nFixationFracMax) : R(0);
  const R halfNFixationMax = C::generic ? PRM(halfNFixationMax) : R(0);
  const R leafNResorptionFrac = C::generic ? PRM(leafNResorptionFrac) : R(0);
  const R fAnoxia = C::generic ? PRM(fAnoxia) : R(0.5);
  const R anaerobicDecompRate = C::generic ? PRM(anaerobicDecompRate) : R(0);
  const R anaerobicTransExp = C::generic ? PRM(anaerobicTransExp) : R(0);
  const R soilMethaneRate = C::generic ? PRM(soilMethaneRate) : R(0);
  const R litterMethaneRate = C::generic ? PRM(litterMethaneRate) : R(0);
  const R soilCSaturation = C::generic ? PRM(soilCSaturation) : R(1);
#undef PRM

  // ---- loop invariants (same value every step in the reference too) ---------
  const R respPerGram = baseFolRespFrac * aMax;              // sipnet.c:617
  const R grossAMax = aMax * aMaxFrac + respPerGram;          // sipnet.c:620
  const R dTempDen = C::fast ? ((psnTMax - psnTMin) / R(2)) * ((psnTMax - psnTMin) / R(2))
                             : rpow((psnTMax - psnTMin) / R(2.0), R(2));  // sipnet.c:625
  const R convLeaf = R(kCWeight) * R(1.0 / kTen9) * (leafCSpWt / cFracLeaf);  // sipnet.c:636-637
  const R convSnow = R((kRho * kCp) / kGamma * (1. / kLambdaS) * 1000. * 1000. * (1. / 10000) * kSecPerDay);
  const R convEvap = R((kRho * kCp) / kGamma * (1. / kLambda) * 1000. * 1000. * (1. / 10000) * kSecPerDay);
  // fast-math hoists
  const R lgVegQ10 = C::fast ? rlog2(vegRespQ10) : R(0);
  const R lgSoilQ10 = C::fast ? rlog2(soilRespQ10) : R(0);
  const R lgFineQ10 = C::fast ? rlog2(fineRootQ10) : R(0);
  const R lgCoarseQ10 = C::fast ? rlog2(coarseRootQ10) : R(0);
  const R folQ10Shift = C::fast ? rexp2(-(psnTOpt / R(10)) * lgVegQ10) : R(0);
  const R invHalfSat = C::fast ? R(1) / halfSatPar : R(0);
  const R invLeafCSpWt = C::fast ? R(1) / leafCSpWt : R(0);
  const R invSoilWHC = C::fast ? R(1) / soilWHC : R(0);
  const R attK = C::fast ? -attenuation * R(1.0 / kNumLayers) : R(0);
  const R kExpLog2e = R(1.4426950408889634074);

  // ---- carried state -> registers --------------------------------------------
#define ST(name) stp[(int64_t)ST_##name * nc]
  double plantWoodC = ST(plantWoodC), plantLeafC = ST(plantLeafC), soilC = ST(soilC);
  double soilWater = ST(soilWater), litterC = ST(litterC), snow = ST(snow);
  double coarseRootC = ST(coarseRootC), fineRootC = ST(fineRootC);
  double minN = ST(minN), soilOrgN = ST(soilOrgN), litterN = ST(litterN);
  double plantStorageN = ST(plantStorageN), plantCAccountingDelta = ST(plantCAccountingDelta);
  double ringSum = ST(ringSum);
  double totGpp = ST(totGpp), totRtot = ST(totRtot), totRa = ST(totRa), totRh = ST(totRh);
  double totNpp = ST(totNpp), totNee = ST(totNee);
  double yearlyGpp = ST(yearlyGpp), yearlyRtot = ST(yearlyRtot), yearlyRa = ST(yearlyRa);
  double yearlyRh = ST(yearlyRh), yearlyNpp = ST(yearlyNpp), yearlyNee = ST(yearlyNee);
  double yearlyLitter = ST(yearlyLitter);
  int phenBits = (int)ST(phenBits);
  int ringValidFrom = (int)ST(ringValidFrom);
  int diedAt = (int)ST(diedAt);
  int clampCount = (int)ST(clampCount);
  const int clampCount0 = clampCount;
  int balanceWarn = 0;
  double diagMaxDC = 0.0, diagMaxDN = 0.0;

  const StepRec* __restrict__ plan = a.plan + (int64_t)site * a.n_steps_total;
  R* __restrict__ ringp = (R*)a.ring + col;   // NPP values of type R (fp32-mixed batches keep them as fp32)
  R* __restrict__ oNee = a.nee ? (R*)a.nee + col : nullptr;
  R* __restrict__ oGpp = a.gpp ? (R*)a.gpp + col : nullptr;
  R* __restrict__ oEt = a.et ? (R*)a.et + col : nullptr;

  const int nLocal = a.step0 + a.n_steps <= siteSteps ? a.n_steps : siteSteps - a.step0;   // (<= 0: the site ended earlier)
  for (int tl = 0; tl < nLocal; tl++) {
    const int t = a.step0 + tl;
    const StepRec& s = plan[t];
    const R len = (R)s.length, tair = (R)s.tair, tsoil = (R)s.tsoil, par = (R)s.par;
    const R vpd = (R)s.vpd, vpdSoil = (R)s.vpdSoil, vPress = (R)s.vPress, wspd = (R)s.wspd;
    const R precip = (R)s.precip;
    const R invLen = C::fast ? (R)s.invLen : R(0);
    const int bits = uni(s.bits);

    // issue this step's ring loads early: their values are only needed at the end
    const int nOps = uni(s.ringOpCount);
    const int opFirst = uni(s.ringOpFirst);
    double evictAcc = 0.0;  // not used directly; ops applied in order below
    (void)evictAcc;

    // ---- 0. per-step init: sipnet.c:1821-1828 ---------------------------------
    const double oldSoilWater = soilWater;
    bool alive = (plantWoodC > kTiny) && (plantWoodC + plantCAccountingDelta > kTiny) &&
                 (fineRootC + coarseRootC > kTiny);  // sipnet.c:1530-1544

    // pools seen by the flux arithmetic
    const R eWood = (R)plantWoodC, eLeaf = (R)plantLeafC, eSoilC = (R)soilC;
    const R eWater = (R)soilWater, eLitter = (R)litterC, eSnow = (R)snow;
    const R eCoarse = (R)coarseRootC, eFine = (R)fineRootC, eDelta = (R)plantCAccountingDelta;
    const R eMinN = (R)minN, eSoilOrgN = (R)soilOrgN, eLitterN = (R)litterN, eStorN = (R)plantStorageN;
    const R totalWoodC = eWood + eDelta;  // state.c:17-19

    // ---- 1. events: events.c:449-742 ------------------------------------------
    R evLeafC = 0, evWoodC = 0, evFineRootC = 0, evCoarseRootC = 0, evEvap = 0, evSoilWater = 0;
    R evSoilC = 0, evLitterC = 0, evMinN = 0, evSoilOrgN = 0, evLitterN = 0;
    R evLeafOnCreation = 0, evLeafOnFromWood = 0, evLeafOffLitter = 0, evLeafOffNResorp = 0;
    R evInputC = 0, evOutputC = 0, evInputN = 0, evOutputN = 0;
    const R dTill = (R)s.dTill;

    auto leafOnNFromC = [&](R leafOnC) -> R {  // nitrogen.c:84-86
      return dmax(R(0), leafOnC / leafCN - leafOnC / woodCN);
    };
    auto leafOnLimit = [&](R flux) -> R {  // limitations.c:13-64
      const R cDemand = flux * len;
      if (cDemand < R(kTiny)) return flux;
      const R availableC = (eWood + eCoarse) * leafOnReallocFrac;
      const R cLimiter = availableC / cDemand;
      R nLimiter = 1;
      if (F.nitrogenCycle) {
        const R nDemand = leafOnNFromC(cDemand);
        if (nDemand > R(kTiny)) nLimiter = eStorN / nDemand;
      }
      const R lim = unitClip(dmin(cLimiter, nLimiter));
      return lim < R(1) ? flux * lim : flux;
    };

    if (F.events) {
      const int nEv = uni(s.evCount);
      const int ev0 = uni(s.evFirst);
      for (int k = 0; k < nEv; k++) {
        const EvRec& ev = a.events[evBase + ev0 + k];
        const int type = uni(ev.type);
        const R p0 = (R)ev.p[0], p1 = (R)ev.p[1], p2 = (R)ev.p[2], p3 = (R)ev.p[3];
        if (type == SIPNET_EV_IRRIG) {  // events.c:484-506
          R evapAmount, soilAmount;
          if ((int)ev.p[1] == 0) {
            evapAmount = immedEvapFrac * p0;
            soilAmount = p0 - evapAmount;
          } else {
            evapAmount = 0;
            soilAmount = p0;
          }
          evEvap += evapAmount / len;
          evSoilWater += soilAmount / len;
        } else if (type == SIPNET_EV_PLANT) {  // events.c:507-543
          evLeafC += p0 / len;
          evWoodC += p1 / len;
          evFineRootC += p2 / len;
          evCoarseRootC += p3 / len;
          evInputC += (p0 + p1 + p2 + p3) / len;
          if (F.nitrogenCycle) {
            evInputN += (p0 / leafCN + p1 / woodCN + p2 / fineRootCN + p3 / woodCN) / len;
          }
        } else if (type == SIPNET_EV_HARVEST) {  // events.c:544-628
          const R fracRA = p0, fracRB = p1, fracTA = p2, fracTB = p3;
          const R woodC = eWood + eDelta;
          R litterAdd = fracTA * (eLeaf + woodC);
          R soilAdd = fracTB * (eFine + eCoarse);
          const R leafDelta = -eLeaf * (fracRA + fracTA);
          const R woodDelta = -woodC * (fracRA + fracTA);
          const R fineDelta = -eFine * (fracRB + fracTB);
          const R coarseDelta = -eCoarse * (fracRB + fracTB);
          if (!F.litterPool) {
            soilAdd += litterAdd;
            litterAdd = 0;
          }
          evLitterC += litterAdd / len;
          evSoilC += soilAdd / len;
          evLeafC += leafDelta / len;
          evWoodC += woodDelta / len;
          evFineRootC += fineDelta / len;
          evCoarseRootC += coarseDelta / len;
          if (F.nitrogenCycle) {
            const R totAbove = (eLeaf / leafCN) + (eWood / woodCN);
            const R totBelow = (eFine / fineRootCN) + (eCoarse / woodCN);
            evSoilOrgN += (fracTB * totBelow) / len;
            evLitterN += (fracTA * totAbove) / len;
            evOutputN += ((eWood / woodCN + eLeaf / leafCN) * fracRA +
                          (eFine / fineRootCN + eCoarse / woodCN) * fracRB) / len;
          }
          evOutputC += ((woodC + eLeaf) * fracRA + (eFine + eCoarse) * fracRB) / len;
        } else if (type == SIPNET_EV_FERT) {  // events.c:640-683
          const R orgC = p1;
          if (F.litterPool) {
            evLitterC += orgC / len;
          } else {
            evSoilC += orgC / len;
          }
          if (F.nitrogenCycle) {
            evLitterN += p0 / len;
            evMinN += p2 / len;
            evInputN += (p0 + p2) / len;
          }
          evInputC += orgC / len;
        } else if (type == SIPNET_EV_LEAFON) {  // events.c:684-702
          const R flux = leafOnLimit(leafGrowth / len);
          evLeafOnCreation += flux;
          const R src = eWood + eCoarse;
          if (src > R(kTiny)) evLeafOnFromWood += flux * eWood / src;
        } else if (type == SIPNET_EV_LEAFOFF) {  // events.c:703-726
          const R leafOff = eLeaf * fracLeafFall;
          evLeafOffLitter += leafOff / len;
          if (F.nitrogenCycle) {
            const R leafN = leafOff / leafCN;
            const R resorb = leafN * leafNResorptionFrac;
            evLeafOffNResorp += resorb / len;
            evLitterN += (leafN - resorb) / len;
          }
        }
        // SIPNET_EV_TILL is folded into StepRec.dTill by the plan (events.c:629-639)
      }
    }

    // ---- 2. fluxes: calculateFluxes(), sipnet.c:1256-1336 ---------------------
    const R lai = C::fast ? eLeaf * invLeafCSpWt : eLeaf / leafCSpWt;  // sipnet.c:1274

    // potPsn(), sipnet.c:590-641
    R dTemp = (psnTMax - tair) * (tair - psnTMin) / dTempDen;
    dTemp = dmax(dTemp, R(0));
    R vpdPow;
    if (C::fast) {
      vpdPow = (dVpdExp == R(2)) ? vpd * vpd : rexp2(dVpdExp * (R)s.log2vpd);
    } else {
      vpdPow = rpow(vpd, dVpdExp);
    }
    R dVpd = R(1) - dVpdSlope * vpdPow;
    dVpd = dmax(dVpd, R(0));
    // calcLightEff(), sipnet.c:517-570: Simpson's rule over 7 canopy layers
    R dLight = 0;
    if (lai > R(0) && par > R(0)) {
      R cum = 0, curr = 0;
      if (C::fast) {
        // exp(-k*lai*i/6) = r^i with r = exp(-k*lai/6); 2^(-I/h) through exp2
        const R r1 = rexp2(attK * lai * kExpLog2e);
        const R q = -par * invHalfSat;
        R ri = 1;
#pragma unroll
        for (int layer = 0; layer <= kNumLayers; layer++) {
          curr = R(1) - rexp2(q * ri);
          const int coeff = (layer == 0) ? 1 : 2 * (1 + layer % 2);
          cum += R(coeff) * curr;
          ri *= r1;
        }
      } else {
#pragma unroll
        for (int layer = 0; layer <= kNumLayers; layer++) {
          const R cumLai = lai * (R((double)layer / kNumLayers));
          const R intensity = par * rexp(R(-1.0) * attenuation * cumLai);
          curr = (R(1) - rexp2((R(-1.0) * intensity / halfSatPar)));
          const int coeff = (layer == 0) ? 1 : 2 * (1 + layer % 2);
          cum += R(coeff) * curr;
        }
      }
      cum -= curr;
      dLight = cum / R(3.0 * kNumLayers);
    }
    const R conversion = convLeaf * lai * R(kSecPerDay);
    const R potGrossPsn = grossAMax * dTemp * dVpd * dLight * conversion;
    const R baseFolResp = respPerGram * conversion;

    // moisture(), sipnet.c:656-699
    R transpiration, dWater;
    if (potGrossPsn < R(kTiny)) {
      transpiration = 0;
      dWater = 1;
    } else {
      const R wue = wueConst / vpd;
      const R potTrans = potGrossPsn / wue * R(1000.0) * R(44.0 / 12.0) * R(1.0 / 10000.0);
      R removable = dmin(eWater, soilWHC) * waterRemoveFrac;
      if (tsoil < frozenSoilThreshold) removable *= frozenSoilEff;
      transpiration = dmin(removable, potTrans);
      dWater = transpiration / potTrans;
    }

    // calcPrecip(), sipnet.c:848-882
    R rain, snowFall, immedEvap;
    {
      const R rate = C::fast ? (R)s.rainRate : precip / len;
      if (tair <= R(0)) {
        snowFall = rate;
        rain = 0;
      } else {
        snowFall = 0;
        rain = rate;
      }
      immedEvap = rain * immedEvapFrac;
      if (F.leafWater) {
        const R maxLeafPool = lai * leafPoolDepth;
        if (immedEvap > maxLeafPool) immedEvap = maxLeafPool;
      }
    }
    const R netRain = rain - immedEvap;

    // snowPack(), sipnet.c:888-946
    R snowMelt, sublimation;
    if (eSnow <= R(0)) {
      snowMelt = 0;
      sublimation = 0;
    } else {
      const R rd = rdConst / wspd;
      sublimation = convSnow * (R(kEStarSnow) - vPress) / rd;
      R remaining = eSnow + (snowFall * len);
      if (sublimation < R(0)) sublimation = 0;
      if (remaining - (sublimation * len) < R(0)) {
        sublimation = remaining / len;
        remaining = 0;
      } else {
        remaining -= (sublimation * len);
      }
      if (tair <= R(0)) {
        snowMelt = 0;
      } else {
        snowMelt = snowMeltP * tair;
        if (remaining - (snowMelt * len) < R(0)) snowMelt = remaining / len;
      }
    }

    // calcSoilWaterFluxes(), sipnet.c:963-1031
    R fastFlow, evaporation, drainage;
    {
      R netIn = netRain + snowMelt;
      fastFlow = netIn * fastFlowFrac;
      netIn -= fastFlow;
      R remaining = eWater + netIn * len - transpiration * len;
      if (eSnow > R(0)) {
        evaporation = 0;
      } else {
        const R waterFrac = unitClip(C::fast ? eWater * invSoilWHC : eWater / soilWHC);
        const R rd = rdConst / wspd;
        const R rsoil = C::fast ? rexp2((rSoilConst1 - rSoilConst2 * waterFrac) * kExpLog2e)
                                : rexp(rSoilConst1 - rSoilConst2 * (waterFrac));
        evaporation = convEvap * vpdSoil / (rd + rsoil);
        if (evaporation < R(0)) evaporation = 0;
        if (remaining - (evaporation * len) < R(kTiny)) {
          evaporation = (remaining - R(kTiny)) / len;
          remaining = 0;
        } else {
          remaining -= (evaporation * len);
        }
      }
      if (remaining > soilWHC) {
        const R excess = remaining - soilWHC;
        if (F.flooding) {
          drainage = dmin(excess * waterDrainFrac, excess / len);
        } else {
          drainage = C::fast ? excess * invLen : excess / len;
        }
      } else {
        drainage = 0;
      }
    }
    const R photosynthesis = potGrossPsn * dWater;  // sipnet.c:1034-1037

    // mean of recent NPP, runmean.c:119-121
    const R meanNpp = (R)(ringSum / kMeanNppDays);

    // vegResp()/vegResp2(), sipnet.c:1051-1103
    R rVeg;
    {
      R folResp, woodQ;
      if (C::fast) {
        woodQ = rexp2((R)s.tair10 * lgVegQ10);
        folResp = baseFolResp * (woodQ * folQ10Shift);
      } else {
        folResp = baseFolResp * rpow(vegRespQ10, (tair - psnTOpt) / R(10.0));
        woodQ = rpow(vegRespQ10, tair / R(10.0));
      }
      if (tsoil < frozenSoilThreshold) folResp *= frozenSoilFolREff;
      const R woodResp = baseVegResp * totalWoodC * woodQ;
      rVeg = folResp + woodResp;
      if (F.growthResp) {
        R growthResp = growthRespFrac * meanNpp;
        if (growthResp < R(0)) growthResp = 0;
        rVeg = folResp + woodResp + growthResp;
      }
    }

    // calcWoodAndLeafFluxes(), sipnet.c:756-782
    const R woodLitter = totalWoodC * woodTurnoverRate;
    R leafLitter = eLeaf * leafTurnoverRate;
    R leafCreation = meanNpp * leafAllocation;
    R woodCreation = meanNpp * woodAllocation;

    // calcLeafOnOffFluxes(), sipnet.c:800-842
    R leafOnCreation = 0, leafOnFromWood = 0, leafOffComputed = 0;
    {
      if (bits & STEP_PHEN_NEW_YEAR) phenBits = 0;
      bool pastGrowth;  // pastLeafGrowth(), sipnet.c:705-731
      if (F.gdd) {
        pastGrowth = (s.cumGdd >= (double)gddLeafOn);
      } else if (F.soilPhenol) {
        pastGrowth = (tsoil >= soilTempLeafOn);
      } else {
        pastGrowth = (leafOnDay > R(0)) && (s.dayTime >= (double)leafOnDay);
      }
      if (!(phenBits & 1) && pastGrowth) {
        const R leafOn = leafOnLimit(leafGrowth / len);
        leafOnCreation += leafOn;
        const R src = eWood + eCoarse;
        if (src > R(kTiny)) leafOnFromWood += leafOn * eWood / src;
        phenBits |= 1;
      }
      const bool pastFall = (leafOffDay > R(0)) && (s.dayTime >= (double)leafOffDay);  // sipnet.c:733-742
      if (!(phenBits & 2) && pastFall) {
        leafOffComputed = (eLeaf * fracLeafFall) / len;
        leafLitter += leafOffComputed;
        phenBits |= 2;
      }
    }

    // dependency effects shared by litter and soil respiration, depeffects.c:23-87
    R tempEff, moistEff;
    if (C::fast) {
      tempEff = rexp2((R)s.tsoil10 * lgSoilQ10);
    } else {
      tempEff = rpow(soilRespQ10, tsoil / R(10));
    }
    if (!F.waterHResp || tsoil < R(0)) {
      moistEff = 1;
    } else {
      const R f_whc = unitClip(C::fast ? eWater * invSoilWHC : eWater / soilWHC);
      if (!F.anaerobic) {
        if (C::fast && soilRespMoistEffect == R(1)) {
          moistEff = f_whc;
        } else {
          moistEff = rpow(f_whc, soilRespMoistEffect);
        }
      } else {
        const R D_aer = unitClip(f_whc / fAnoxia);
        const R A = unitClip((f_whc - fAnoxia) / (R(1) - fAnoxia));
        moistEff = (R(1) - A) * D_aer + anaerobicDecompRate * A;
      }
    }
    const R tillEff = R(1) + dTill;  // depeffects.c:76

    // calcLitterFluxes(), sipnet.c:1150-1171
    R rLitter = 0, litterToSoil = 0;
    if (F.litterPool) {
      R cn = 1;
      if (F.nitrogenCycle) {
        const R den = eLitterN < R(kTiny) ? R(kTiny) : eLitterN;  // util.c:72-75
        cn = kCN / (kCN + eLitter / den);
      }
      const R breakdown = eLitter * litterBreakdownRate * tempEff * moistEff * tillEff * cn;
      rLitter = breakdown * fracLitterRespired;
      litterToSoil = breakdown * (R(1.0) - fracLitterRespired);
    }

    // calcRootFluxes(), sipnet.c:1176-1196
    const R coarseRootLoss = coarseRootTurnoverRate * eCoarse;
    const R fineRootLoss = fineRootTurnoverRate * eFine;
    R coarseRootCreation = coarseRootAllocation * meanNpp;
    R fineRootCreation = fineRootAllocation * meanNpp;
    R rCoarseRoot, rFineRoot;
    if (C::fast) {
      rCoarseRoot = baseCoarseRootResp * eCoarse * rexp2((R)s.tsoil10 * lgCoarseQ10);
      rFineRoot = baseFineRootResp * eFine * rexp2((R)s.tsoil10 * lgFineQ10);
    } else {
      rCoarseRoot = baseCoarseRootResp * eCoarse * rpow(coarseRootQ10, tsoil / R(10.0));
      rFineRoot = baseFineRootResp * eFine * rpow(fineRootQ10, tsoil / R(10.0));
    }

    // calcSoilRespiration(), sipnet.c:1132-1148
    R rSoil;
    {
      R cn = 1;
      if (F.nitrogenCycle) {
        const R den = eSoilOrgN < R(kTiny) ? R(kTiny) : eSoilOrgN;
        cn = kCN / (kCN + eSoilC / den);
      }
      rSoil = eSoilC * baseSoilResp * moistEff * tempEff * tillEff * cn;
    }

    // calcMethaneFlux(), sipnet.c:1201-1214
    R soilMethane = 0, litterMethane = 0;
    if (F.anaerobic) {
      const R f_whc = unitClip(eWater / soilWHC);
      const R A = unitClip((f_whc - fAnoxia) / (R(1) - fAnoxia));
      const R mMoist = rpow(A, anaerobicTransExp);
      soilMethane = soilMethaneRate * eSoilC * tempEff * mMoist;
      if (F.litterPool) litterMethane = litterMethaneRate * eLitter * tempEff * mMoist;
    }

    // checkNegativeCreation(), limitations.c:146-182
    {
      const R turnover = eLeaf * leafTurnoverRate;
      const R leafDeficit = (C::fast ? eLeaf * invLen : eLeaf / len) + leafCreation - turnover;
      if (leafDeficit < R(0)) {
        woodCreation += leafDeficit;
        leafCreation -= leafDeficit;
      }
      const R fineDef = (C::fast ? eFine * invLen : eFine / len) + fineRootCreation - fineRootLoss;
      const R coarseDef = (C::fast ? eCoarse * invLen : eCoarse / len) + coarseRootCreation - coarseRootLoss;
      if ((fineDef < R(0)) != (coarseDef < R(0))) {
        if (fineDef < R(0)) {
          coarseRootCreation += fineDef;
          fineRootCreation -= fineDef;
        }
        if (coarseDef < R(0)) {
          fineRootCreation += coarseDef;
          coarseRootCreation -= coarseDef;
        }
      }
    }

    // nitrogen cycle, nitrogen.c:15-207 + limitations.c:69-139
    R nVolatilization = 0, nLeaching = 0, nOrgSoil = 0, nOrgLitter = 0, nMin = 0;
    R nFixation = 0, nUptake = 0, leafOffNResorption = 0, reductionNResorption = 0;
    if (F.nitrogenCycle) {
      auto plantNDemand = [&]() -> R {  // nitrogen.c:89-104
        const R d = woodCreation / woodCN + leafCreation / leafCN +
                    fineRootCreation / fineRootCN + coarseRootCreation / woodCN;
        return dmax(R(0), d);
      };
      auto unclaimedStorage = [&]() -> R {  // nitrogen.c:127-134
        const R leafOnN = leafOnNFromC(leafOnCreation + evLeafOnCreation);
        return dmax(R(0), eStorN - leafOnN * len);
      };
      auto fixationFrac = [&]() -> R {  // nitrogen.c:137-152
        const R denom = halfNFixationMax + eMinN;
        const R inhibition = (denom < R(kTiny)) ? R(1) : halfNFixationMax / denom;
        return nFixationFracMax * inhibition;
      };
      auto fixationAndUptake = [&]() {  // nitrogen.c:155-168
        const R demand = plantNDemand();
        const R storage = unclaimedStorage() / len;
        const R rem = dmax(R(0), demand - storage);
        const R frac = fixationFrac();
        nFixation = frac * rem;
        nUptake = (R(1) - frac) * rem;
      };
      // calcNResorptionFluxes, nitrogen.c:170-196
      if (woodCreation + leafCreation + fineRootCreation + coarseRootCreation < R(0)) {
        reductionNResorption -= (leafCreation / leafCN + woodCreation / woodCN +
                                 coarseRootCreation / woodCN + fineRootCreation / fineRootCN);
      }
      leafOffNResorption += leafNResorptionFrac * leafLitter / leafCN;
      // volatilisation, nitrogen.c:15-26 (+ depeffects.c:89-96)
      {
        const R f_whc = unitClip(eWater / soilWHC);
        const R A = unitClip((f_whc - fAnoxia) / (R(1) - fAnoxia));
        const R d_water = R(0.05) + R(3.8) * A * (R(1) - A);
        nVolatilization = nVolatilizationFrac * eMinN * tempEff * d_water;
      }
      // leaching, nitrogen.c:31-41
      {
        const R ratio = drainage / soilWHC;
        const R phi = (ratio < R(1)) ? ratio : R(1);
        nLeaching = eMinN * phi * nLeachingFrac;
      }
      // pool fluxes, nitrogen.c:45-82
      {
        const R litterCN = eLitter / (eLitterN < R(kTiny) ? R(kTiny) : eLitterN);
        const R soilCN = eSoilC / (eSoilOrgN < R(kTiny) ? R(kTiny) : eSoilOrgN);
        const R litterMin = rLitter / litterCN;
        const R soilMin = rSoil / soilCN;
        const R soilNInputs = litterToSoil / litterCN + fineRootLoss / fineRootCN +
                              coarseRootLoss / woodCN;
        const R sat = F.carbonSaturation ? unitClip(eSoilC / soilCSaturation) : R(0);
        nOrgLitter = leafLitter / leafCN - leafOffNResorption + woodLitter / woodCN -
                     litterMin - litterToSoil / litterCN + (soilNInputs * sat);
        nOrgSoil = soilNInputs * (R(1) - sat) - soilMin;
        nMin = litterMin + soilMin;
      }
      fixationAndUptake();
      // checkMineralNLimitation, limitations.c:119-129
      {
        const R pool = eMinN + (nMin + evMinN) * len;
        const R loss = (nLeaching + nVolatilization) * len;
        if (loss > R(kTiny) && loss > pool) {
          const R red = pool / loss;
          nLeaching *= red;
          nVolatilization *= red;
        }
      }
      // checkNitrogenLimitation, limitations.c:69-114
      {
        const R uptakeDemand = nUptake * len;
        const R nonUptakeDelta = (nMin - nVolatilization - nLeaching) * len;
        const R availableMinN = eMinN + nonUptakeDelta;
        if (uptakeDemand > R(kTiny) && uptakeDemand > availableMinN) {
          const R unclaimed = unclaimedStorage();
          const R demand = plantNDemand() * len;
          const R uptakeFrac = R(1) - fixationFrac();
          const R red = (availableMinN / uptakeFrac + unclaimed) / demand;
          woodCreation *= red;
          leafCreation *= red;
          fineRootCreation *= red;
          coarseRootCreation *= red;
          fixationAndUptake();
        }
      }
    }

    // ---- 3. pools: updatePoolsAndBalance(), sipnet.c:1769-1806 ---------------
    const double dl = (double)len;
    // getMassTotals(), balance.c:13-36, for the per-step balance diagnostics (a.diag)
    auto massC = [&]() -> double {
      double c = (plantWoodC + plantCAccountingDelta) + plantLeafC + fineRootC + coarseRootC + soilC;
      if (F.litterPool) c += litterC;
      return c;
    };
    auto massN = [&]() -> double {
      if (!F.nitrogenCycle) return 0.0;
      return plantWoodC / (double)woodCN + plantLeafC / (double)leafCN + fineRootC / (double)fineRootCN +
             coarseRootC / (double)woodCN + soilOrgN + litterN + minN + plantStorageN;
    };
    double preC = 0.0, preN = 0.0, postC = 0.0, postN = 0.0;
    if (a.diag) {
      preC = massC();
      preN = massN();
    }
    // updatePoolsForEvents(), events.c:744-790
    if (F.events) {
      plantWoodC += (double)(evWoodC * len);
      plantLeafC += (double)(evLeafC * len);
      soilC += (double)(evSoilC * len);
      if (F.litterPool) litterC += (double)(evLitterC * len);
      plantWoodC -= (double)(evLeafOnFromWood * len);
      const R evFromRoot = evLeafOnCreation - evLeafOnFromWood;
      coarseRootC -= (double)(evFromRoot * len);
      plantLeafC += (double)((evLeafOnCreation - evLeafOffLitter) * len);
      if (F.litterPool) {
        litterC += (double)(evLeafOffLitter * len);
      } else {
        soilC += (double)(evLeafOffLitter * len);
      }
      coarseRootC += (double)(evCoarseRootC * len);
      fineRootC += (double)(evFineRootC * len);
      soilWater += (double)(evSoilWater * len);
      if (F.nitrogenCycle) {
        minN += (double)(evMinN * len);
        soilOrgN += (double)(evSoilOrgN * len);
        litterN += (double)(evLitterN * len);
        const R leafOnN = leafOnNFromC(evLeafOnCreation);
        plantStorageN += (double)((evLeafOffNResorp - leafOnN) * len);
      }
    }
    // updateMainPools(), sipnet.c:1579-1626
    {
      const R r_a = rVeg + rFineRoot + rCoarseRoot;
      const R alloc = leafCreation + woodCreation + fineRootCreation + coarseRootCreation;
      plantCAccountingDelta += (double)(((photosynthesis - r_a) - alloc) * len);
      plantWoodC += (double)((woodCreation - woodLitter - leafOnFromWood) * len);
      plantLeafC += (double)((leafCreation + leafOnCreation - leafLitter) * len);
      soilWater += (double)((rain + snowMelt - immedEvap - fastFlow - evaporation -
                             transpiration - drainage) * len);
      snow += (double)((snowFall - snowMelt - sublimation) * len);
    }
    // updatePoolsForSoil(), sipnet.c:1634-1680
    {
      if (F.litterPool) {
        const R soilInputs = coarseRootLoss + fineRootLoss + litterToSoil;
        // envi.soilC at this point already holds this step's event fluxes (events.c:744-790
        // runs first), unlike the start-of-step value the nitrogen fluxes saw
        const R sat = F.carbonSaturation ? unitClip((R)soilC / soilCSaturation) : R(0);
        litterC += (double)((woodLitter + leafLitter + (soilInputs * sat) - litterToSoil -
                             rLitter - litterMethane) * len);
        soilC += (double)((soilInputs * (R(1) - sat) - rSoil - soilMethane) * len);
      } else {
        soilC += (double)((coarseRootLoss + fineRootLoss + woodLitter + leafLitter - rSoil -
                           soilMethane) * len);
      }
      const R fromRoot = leafOnCreation - leafOnFromWood;
      coarseRootC += (double)((coarseRootCreation - coarseRootLoss - fromRoot) * len);
      fineRootC += (double)((fineRootCreation - fineRootLoss) * len);
    }
    // updateNitrogenPools(), nitrogen.c:210-239
    if (F.nitrogenCycle) {
      const R d = woodCreation / woodCN + leafCreation / leafCN +
                  fineRootCreation / fineRootCN + coarseRootCreation / woodCN;
      const R demand = dmax(R(0), d);
      const R storageDemand = demand - nUptake - nFixation;
      const R leafOnN = leafOnNFromC(leafOnCreation);
      plantStorageN += (double)((leafOffNResorption + reductionNResorption - storageDemand -
                                 leafOnN) * len);
      const R nonUptake = nMin - nVolatilization - nLeaching;
      minN += (double)((nonUptake - nUptake) * len);
      soilOrgN += (double)(nOrgSoil * len);
      litterN += (double)(nOrgLitter * len);
    }
    (void)dl;
    if (a.diag) {
      postC = massC();
      postN = massN();
    }

    // checkForMortality(), sipnet.c:1688-1767
    double diedNow = 0.0, deathWood = 0.0, deathRoot = 0.0;
    {
      const bool sufficient = (plantWoodC > kTiny) &&
                              (plantWoodC + plantCAccountingDelta > kTiny) &&
                              (fineRootC + coarseRootC > kTiny);
      if (!alive) {
        if (sufficient) alive = true;
      } else if (!sufficient) {
        alive = false;
        if (diedAt < 0) diedAt = t;
        const double root = fineRootC + coarseRootC;
        diedNow = 1.0;
        deathWood = plantWoodC + plantCAccountingDelta;
        deathRoot = root;
        soilC += root;
        if (F.litterPool) {
          litterC += plantWoodC + plantLeafC + plantCAccountingDelta;
        } else {
          soilC += plantWoodC + plantLeafC + plantCAccountingDelta;
        }
        if (F.nitrogenCycle) {
          soilOrgN += fineRootC / (double)fineRootCN + coarseRootC / (double)woodCN;
          litterN += plantWoodC / (double)woodCN + plantLeafC / (double)leafCN + plantStorageN;
          plantStorageN = 0.0;
        }
        plantWoodC = 0.0;
        plantLeafC = 0.0;
        coarseRootC = 0.0;
        fineRootC = 0.0;
        plantCAccountingDelta = 0.0;
        ringSum = 0.0;  // resetMeanTracker(meanNPP, 0), sipnet.c:1757
      }
    }

    // ensureNonNegativeStocks(), sipnet.c:1368-1397
#define CLAMP(v, minVal)                 \
  if (v < (minVal)) {                    \
    if (fabs(v) > kEps) clampCount++;    \
    v = 0.;                              \
  }
    CLAMP(plantWoodC, 0.0)
    CLAMP(plantLeafC, 0.0)
    if (F.litterPool) {
      CLAMP(litterC, 0.0)
    }
    CLAMP(soilC, 0.0)
    CLAMP(coarseRootC, 0.0)
    CLAMP(fineRootC, 0.0)
    CLAMP(soilWater, 0.0)
    CLAMP(snow, kTiny)
    if (C::generic) {
      CLAMP(minN, 0.0)
      CLAMP(soilOrgN, 0.0)
      CLAMP(litterN, 0.0)
      CLAMP(plantStorageN, 0.0)
    }
#undef CLAMP
    if (a.diag) {  // updateBalanceTrackerPostClamp() + checkBalance(), balance.c:40-169
      const double finC = massC(), finN = massN();
      double clampedC = finC - postC, clampedN = finN - postN;
      if (clampedC < kEps) clampedC = 0.0;
      if (clampedN < kEps) clampedN = 0.0;
      double inC = (double)photosynthesis + (double)evInputC;
      double outC = (double)rVeg + (double)rFineRoot + (double)rCoarseRoot + (double)rSoil +
                    (double)soilMethane + (double)evOutputC;
      if (F.litterPool) outC += (double)rLitter + (double)litterMethane;
      inC *= (double)len;
      outC *= (double)len;
      double inN = 0.0, outN = 0.0;
      if (F.nitrogenCycle) {
        inN = ((double)nFixation + (double)evInputN) * (double)len;
        outN = ((double)nLeaching + (double)nVolatilization + (double)evOutputN) * (double)len;
      }
      inC += clampedC;
      if (F.nitrogenCycle) inN += clampedN;
      const double dC = (finC - preC) - (inC - outC);
      const double dN = (finN - preN) + (outN - inN);
      diagMaxDC = fmax(diagMaxDC, fabs(dC));
      diagMaxDN = fmax(diagMaxDN, fabs(dN));
      if (!(fabs(dC) < kEps)) balanceWarn++;
      if (!(fabs(dN) < kEps)) balanceWarn++;
    }

    // ---- 4. trackers: updateTrackers(), sipnet.c:1420-1496 --------------------
    if (bits & STEP_TRACK_NEW_YEAR) {
      yearlyGpp = yearlyRtot = yearlyRa = yearlyRh = yearlyNpp = yearlyNee = 0.0;
    }
    const R tGpp = photosynthesis * len;
    const R tRh = (rLitter + rSoil) * len;
    const R tRAbove = (rVeg)*len;
    const R tRRoot = (rCoarseRoot + rFineRoot) * len;
    const R tRSoil = tRRoot + tRh;
    const R tRa = tRRoot + tRAbove;
    const R tRtot = tRa + tRh;
    const R tNpp = tGpp - tRa;
    const R tNee = R(-1.0) * (tNpp - tRh);
    yearlyGpp += (double)tGpp;
    yearlyRa += (double)tRa;
    yearlyRh += (double)tRh;
    yearlyRtot += (double)tRtot;
    yearlyNpp += (double)tNpp;
    yearlyNee += (double)tNee;
    totGpp += (double)tGpp;
    totRa += (double)tRa;
    totRh += (double)tRh;
    totRtot += (double)tRtot;
    totNpp += (double)tNpp;
    totNee += (double)tNee;
    const R tEt = (transpiration + immedEvap + evaporation + sublimation + evEvap) * len;
    yearlyLitter += (double)(leafLitter + evLeafOffLitter);

    // ---- outputs --------------------------------------------------------------
    if (oNee) oNee[(int64_t)tl * a.ld] = tNee;
    if (oGpp) oGpp[(int64_t)tl * a.ld] = tGpp;
    if (oEt) oEt[(int64_t)tl * a.ld] = tEt;
    if (C::fullRec && a.rec) {
      double* __restrict__ r = a.rec + (int64_t)tl * SIPNET_NREC * a.ld + col;
      const int64_t L = a.ld;
      r[0 * L] = (double)tNee;
      r[1 * L] = (double)tGpp;
      r[2 * L] = (double)tEt;
      r[3 * L] = totNee;
      r[4 * L] = (double)tNpp;
      r[5 * L] = (double)tRAbove;
      r[6 * L] = (double)tRSoil;
      r[7 * L] = (double)tRRoot;
      r[8 * L] = (double)tRa;
      r[9 * L] = (double)tRh;
      r[10 * L] = (double)tRtot;
      r[11 * L] = (double)(woodCreation * len);
      r[12 * L] = (oldSoilWater + soilWater) / (2.0 * (double)soilWHC);
      r[13 * L] = (double)transpiration;
      r[14 * L] = plantWoodC;
      r[15 * L] = plantLeafC;
      r[16 * L] = soilC;
      r[17 * L] = soilWater;
      r[18 * L] = litterC;
      r[19 * L] = snow;
      r[20 * L] = coarseRootC;
      r[21 * L] = fineRootC;
      r[22 * L] = minN;
      r[23 * L] = soilOrgN;
      r[24 * L] = litterN;
      r[25 * L] = plantStorageN;
      r[26 * L] = plantCAccountingDelta;
      r[27 * L] = F.nitrogenCycle ? (double)(nVolatilization * len) : 0.0;
      r[28 * L] = F.nitrogenCycle ? (double)(nLeaching * len) : 0.0;
      r[29 * L] = F.nitrogenCycle ? (double)(nFixation * len) : 0.0;
      r[30 * L] = F.nitrogenCycle ? (double)(nUptake * len) : 0.0;
      r[31 * L] = (double)((soilMethane + litterMethane) * len);
      r[32 * L] = ringSum / kMeanNppDays;
      r[33 * L] = s.gddAfter;
      r[34 * L] = s.tillAfter;
      r[35 * L] = totGpp;
      // event log for events.out (sipnet.c:1230-1247, :829-841, :1759-1765)
      r[36 * L] = (double)(leafOnCreation * len);
      r[37 * L] = (double)(leafOnFromWood * len);
      r[38 * L] = (double)(leafOffComputed * len);
      r[39 * L] = (double)(evLeafOnCreation * len);
      r[40 * L] = (double)(evLeafOnFromWood * len);
      r[41 * L] = deathWood;
      r[42 * L] = deathRoot;
      r[43 * L] = diedNow;
      // debug plane (--debug-log): the 56 Fluxes fields in the order of the reference's
      // fluxes log (debug_log.c:70-125), then the tracker fields no record column carries
      if (a.dbg) {
        double* __restrict__ g = a.dbg + (int64_t)tl * SIPNET_NDBG * a.ld + col;
        const R v[SIPNET_NDBG] = {
            photosynthesis, leafLitter, woodLitter, rVeg, rSoil, rain, transpiration, drainage,
            litterToSoil, rLitter, snowFall, snowMelt, sublimation, immedEvap, fastFlow,
            evaporation, fineRootLoss, coarseRootLoss, fineRootCreation, coarseRootCreation,
            rCoarseRoot, rFineRoot, leafCreation, woodCreation, leafOnCreation, leafOnFromWood,
            nVolatilization, nLeaching, nOrgSoil, nOrgLitter, nMin, nFixation, nUptake,
            leafOffNResorption, reductionNResorption, evLeafC, evWoodC, evFineRootC,
            evCoarseRootC, evEvap, evSoilWater, evSoilC, evLitterC, evMinN, evSoilOrgN,
            evLitterN, evInputC, evOutputC, evInputN, evOutputN, evLeafOnCreation,
            evLeafOnFromWood, evLeafOffLitter, evLeafOffNResorp, soilMethane, litterMethane};
#pragma unroll
        for (int k = 0; k < 56; k++) g[(int64_t)k * L] = (double)v[k];
        g[56 * L] = yearlyGpp;
        g[57 * L] = yearlyRtot;
        g[58 * L] = yearlyRa;
        g[59 * L] = yearlyRh;
        g[60 * L] = yearlyNpp;
        g[61 * L] = yearlyNee;
        g[62 * L] = yearlyLitter;
        g[63 * L] = totRtot;
        g[64 * L] = totRa;
        g[65 * L] = totRh;
        g[66 * L] = totNpp;
        g[67 * L] = (double)(phenBits & 1);
        g[68 * L] = (double)((phenBits >> 1) & 1);
        g[69 * L] = alive ? 1.0 : 0.0;
        g[70 * L] = 0.0;
        g[71 * L] = 0.0;
      }
    }

    // ---- 5. running mean of NPP: updateMeanTrackers(), sipnet.c:1546-1570 with
    //         runmean.c:61-116 driven by the site plan ---------------------------
    {
      const int insSlot = uni(s.ringInsSlot);
      if (alive) {
        const double npp = (double)(photosynthesis - rVeg - rCoarseRoot - rFineRoot);
        if (insSlot < 0) {
          // weight >= totWeight: the new value replaces everything (runmean.c:67-69)
          ringp[0] = (R)npp;
          ringSum = npp * kMeanNppDays;
        } else {
          for (int k = 0; k < nOps; k++) {
            const RingOp& op = a.ringOps[opBase + opFirst + k];
            const int slot = uni(op.slot);
            const int ins = uni(op.insStep);
            const double v = (ins >= ringValidFrom) ? (double)ringp[(int64_t)slot * nc] : 0.0;
            ringSum -= op.w * v;
          }
          ringp[(int64_t)insSlot * nc] = (R)npp;
          ringSum += npp * dl;
        }
      } else {
        // dead members do not insert (sipnet.c:1547-1552); everything inserted up
        // to now counts as zero from here on (ring epochs)
        ringValidFrom = t + 1;
      }
    }
  }

  // ---- state back to HBM ------------------------------------------------------
  ST(plantWoodC) = plantWoodC;
  ST(plantLeafC) = plantLeafC;
  ST(soilC) = soilC;
  ST(soilWater) = soilWater;
  ST(litterC) = litterC;
  ST(snow) = snow;
  ST(coarseRootC) = coarseRootC;
  ST(fineRootC) = fineRootC;
  ST(minN) = minN;
  ST(soilOrgN) = soilOrgN;
  ST(litterN) = litterN;
  ST(plantStorageN) = plantStorageN;
  ST(plantCAccountingDelta) = plantCAccountingDelta;
  ST(ringSum) = ringSum;
  ST(totGpp) = totGpp;
  ST(totRtot) = totRtot;
  ST(totRa) = totRa;
  ST(totRh) = totRh;
  ST(totNpp) = totNpp;
  ST(totNee) = totNee;
  ST(yearlyGpp) = yearlyGpp;
  ST(yearlyRtot) = yearlyRtot;
  ST(yearlyRa) = yearlyRa;
  ST(yearlyRh) = yearlyRh;
  ST(yearlyNpp) = yearlyNpp;
  ST(yearlyNee) = yearlyNee;
  ST(yearlyLitter) = yearlyLitter;
  ST(phenBits) = (double)phenBits;
  ST(ringValidFrom) = (double)ringValidFrom;
  ST(diedAt) = (double)diedAt;
  ST(clampCount) = (double)clampCount;
  if (a.diag) {
    double* __restrict__ dg = a.diag + col;
    dg[0 * nc] += (double)(clampCount - clampCount0);
    dg[1 * nc] += (double)balanceWarn;
    dg[2 * nc] = fmax(dg[2 * nc], diagMaxDC);
    dg[3 * nc] = fmax(dg[3 * nc], diagMaxDN);
  }
#undef ST
}

// -----------------------------------------------------------------------------
// ensemble statistics of an output plane: per (step, site) sum and sum of
// squares over members.  One wavefront per (step, site); lanes stride over the
// site's members (coalesced), then a DPP/shuffle butterfly folds the 64 lanes.
// HBM-bound streaming read.
// -----------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void reducePlaneKernel(const T* __restrict__ plane,
                                                         int32_t n_steps, int64_t ld,
                                                         int32_t n_sites, int32_t n_members,
                                                         double* __restrict__ stats) {
  const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t nItems = (int64_t)n_steps * n_sites;
  if (wave >= nItems) return;
  const int t = wave / n_sites, site = wave % n_sites;
  const T* __restrict__ row = plane + (int64_t)t * ld + (int64_t)site * n_members;
  // 16-byte loads, four independent accumulator pairs: enough bytes in flight per wave to
  // stream at HBM rate; a scalar tail / fallback covers odd lengths and unaligned rows
  double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, c1 = 0.0, c2 = 0.0, d1 = 0.0, d2 = 0.0;
  constexpr int kPer = 16 / (int)sizeof(T);  // elements per 16-byte load
  int done = 0;
  if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    typedef T vec_t __attribute__((ext_vector_type(kPer)));
    const vec_t* __restrict__ rv = reinterpret_cast<const vec_t*>(row);
    const int nVec = n_members / kPer;
    int i = lane;
    for (; i + 192 < nVec; i += 256) {
      const vec_t v0 = rv[i], v1 = rv[i + 64], v2 = rv[i + 128], v3 = rv[i + 192];
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const double x0 = (double)v0[k], x1 = (double)v1[k], x2 = (double)v2[k], x3 = (double)v3[k];
        a1 += x0; a2 += x0 * x0;
        b1 += x1; b2 += x1 * x1;
        c1 += x2; c2 += x2 * x2;
        d1 += x3; d2 += x3 * x3;
      }
    }
    for (; i < nVec; i += 64) {
      const vec_t v0 = rv[i];
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const double x0 = (double)v0[k];
        a1 += x0; a2 += x0 * x0;
      }
    }
    done = nVec * kPer;
  }
  for (int m = done + lane; m < n_members; m += 64) {
    const double v = (double)row[m];
    a1 += v;
    a2 += v * v;
  }
  double s1 = (a1 + b1) + (c1 + d1), s2 = (a2 + b2) + (c2 + d2);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if (lane == 0) {
    stats[(int64_t)wave * 2 + 0] = s1;
    stats[(int64_t)wave * 2 + 1] = s2;
  }
}

// second stage of the in-kernel ensemble statistics (step_coop.hip, wave L): one thread per
// (plane, step, site) adds up the site's per-chunk partial sums, chunk by chunk in index order
// (deterministic); neighbouring threads read neighbouring steps
__global__ __launch_bounds__(256) void finishStatsKernel(const double* __restrict__ part, int32_t n_steps,
                                                         int32_t n_sites, int32_t chunksPerSite,
                                                         double* __restrict__ stats) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t perPlane = (int64_t)n_steps * n_sites;
  if (i >= 3 * perPlane) return;
  const int p = (int)(i / perPlane);
  const int64_t rem = i - (int64_t)p * perPlane;
  const int site = (int)(rem / n_steps), t = (int)(rem % n_steps);
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* __restrict__ src =
      reinterpret_cast<const d2*>(part) + ((int64_t)p * n_sites + site) * chunksPerSite * n_steps + t;
  double s1 = 0.0, s2 = 0.0;
  int c = 0;
  for (; c + 8 <= chunksPerSite; c += 8) {   // eight loads in flight, added in chunk order (1 024 chunks at C3:
    d2 v[8];                                 // one dependent load at a time took 0.5 ms)
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = src[(int64_t)(c + k) * n_steps];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      s1 += v[k].x;
      s2 += v[k].y;
    }
  }
  for (; c < chunksPerSite; c++) {
    const d2 v = src[(int64_t)c * n_steps];
    s1 += v.x;
    s2 += v.y;
  }
  double* dst = stats + (((int64_t)p * n_steps + t) * n_sites + site) * 2;
  dst[0] = s1;
  dst[1] = s2;
}

template <class C>
void launchOne(const KernelArgs& a, hipStream_t stream) {
  const int chunksPerSite = (a.n_members + 63) / 64;
  const int grid = a.n_sites * chunksPerSite;
  hipLaunchKernelGGL(stepKernel<C>, dim3(grid), dim3(64), 0, stream, a);
}

}  // namespace

void launchFinishStats(const double* statsPart, int32_t n_steps, int32_t n_sites, int32_t chunksPerSite,
                       double* stats, hipStream_t stream) {
  const int64_t n = (int64_t)3 * n_steps * n_sites;
  hipLaunchKernelGGL(finishStatsKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, statsPart,
                     n_steps, n_sites, chunksPerSite, stats);
}
void launchSetup(const SetupArgs& a, hipStream_t stream) {
  const int grid = (int)((a.ncol + 255) / 256);
  hipLaunchKernelGGL(setupKernel, dim3(grid), dim3(256), 0, stream, a);
}
void launchConvertParams(const double* rawRows, double* prm, int64_t ncol, int64_t col0,
                         int32_t count, int32_t leafOnMode, hipStream_t stream, int32_t nRep, int64_t repStride) {
  hipLaunchKernelGGL(convertParamsKernel, dim3((count + 255) / 256, SIPNET_NPARAMS), dim3(256), 0, stream, rawRows,
                     prm, ncol, col0, count, leafOnMode, nRep, repStride);
}

static bool isDefaultFlags(const int32_t* f) {
  for (int i = 0; i < SIPNET_NFLAGS; i++) {
    if (i == SIPNET_F_SNOW) continue;  // snow only gates a parameter's required-ness
    if ((f[i] != 0) != defaultFlag(i)) return false;
  }
  return true;
}

// For the throughput kernels with the default flag set compiled in, four flags are DATA, not code: events
// (no events, no event records), the phenology pair gdd / soil_phenol (the plan puts the leaf-on variable the
// flags ask for into the record, the parameter conversion the matching threshold into the row the kernels read)
// and water_hresp (off = the plan marks every step like one with frozen soil: moisture effect 1).
bool isPhenologyOrEventsFlag(int i) { return i == SIPNET_F_EVENTS || i == SIPNET_F_GDD || i == SIPNET_F_SOIL_PHENOL; }
bool isDefaultFlagSet(const int32_t* f) {
  for (int i = 0; i < SIPNET_NFLAGS; i++) {
    if (i == SIPNET_F_SNOW || i == SIPNET_F_WATER_HRESP || isPhenologyOrEventsFlag(i)) continue;  // (snow only gates a parameter's required-ness)
    if ((f[i] != 0) != defaultFlag(i)) return false;
  }
  return true;
}

void launchStep(const KernelArgs& a, int precision, bool fastMath, hipStream_t stream, LaunchInfo* info) {
  const bool generic = !isDefaultFlags(a.flags);
  const bool full = a.rec != nullptr;
  // full-record launches (CLI text output, checkpoints) always use the generic
  // strict/fast kernels with FullRec; throughput launches use the lean ones.
  if (precision == SIPNET_F64) {
    if (full) {
      if (fastMath) launchOne<Cfg<double, true, true, true>>(a, stream);
      else launchOne<Cfg<double, false, true, true>>(a, stream);
    } else if (generic) {
      if (fastMath) launchOne<Cfg<double, true, true, false>>(a, stream);
      else launchOne<Cfg<double, false, true, false>>(a, stream);
    } else {
      if (fastMath) launchOne<Cfg<double, true, false, false>>(a, stream);
      else launchOne<Cfg<double, false, false, false>>(a, stream);
    }
  } else {
    if (full) launchOne<Cfg<float, true, true, true>>(a, stream);
    else if (generic) launchOne<Cfg<float, true, true, false>>(a, stream);
    else launchOne<Cfg<float, true, false, false>>(a, stream);
  }
  if (info) {
    const bool fm = fastMath || precision != SIPNET_F64;
    snprintf(info->kernel, sizeof info->kernel, "stepKernel<Cfg<%s, %s, %s, %s>>",
             precision == SIPNET_F64 ? "double" : "float", fm ? "true" : "false",
             (full || generic) ? "true" : "false", full ? "true" : "false");
    info->grid = a.n_sites * ((a.n_members + 63) / 64);
    info->block = 64;
    info->wavesPerSimd = 1;
    info->ldsBytes = 0;
  }
}

void launchReducePlane(const void* plane, bool isF32, int32_t n_steps, int64_t ld,
                       int32_t n_sites, int32_t n_members, double* stats,
                       hipStream_t stream) {
  const int64_t waves = (int64_t)n_steps * n_sites;
  const int grid = (int)((waves * 64 + 255) / 256);
  if (isF32) {
    hipLaunchKernelGGL(reducePlaneKernel<float>, dim3(grid), dim3(256), 0, stream,
                       (const float*)plane, n_steps, ld, n_sites, n_members, stats);
  } else {
    hipLaunchKernelGGL(reducePlaneKernel<double>, dim3(grid), dim3(256), 0, stream,
                       (const double*)plane, n_steps, ld, n_sites, n_members, stats);
  }
}

}  // namespace sipnet

// step_fast.hip -- the throughput kernels: fast-math policy; default model flags compiled in
// (Generic = false) or any flag set read at run time (Generic = true: litter pool, nitrogen
// cycle, anaerobic / methane, carbon saturation, flooding, growth respiration, leaf water,
// soil-temperature / calendar phenology, no moisture effect on heterotrophic respiration).
//
// Same model as stepKernel<...> in step_kernel.hip (which stays the strict, all-flags
// reference path on the GPU); this translation unit is where the instruction count of a
// member-step is driven down, because with one wavefront per SIMD the kernel is bound by
// VALU issue (4 cycles per wave64 fp64 instruction), not by HBM:
//
//   * site-uniform data arrives as FastRec tiles staged through LDS (async global->LDS
//     DMA one tile ahead, double-buffered), read back with broadcast ds_read_b128;
//   * the ring value evicted in step t+1 is requested at the top of step t;
//   * no division by a site quantity and none by a loop-invariant member quantity is left in
//     the loop; the two remaining true divisions use v_rcp + Newton steps;
//   * exp2 is a 9th-degree polynomial (<= 3.7e-14 relative; fast_math.h, tools/fit_exp2.py) +
//     v_ldexp; pow(q, T/10) = exp2(T/10 * log2 q) with log2 q hoisted; the seven Simpson layers
//     share one exp;
//   * compiled with -ffp-contract=fast (a*b+c fuses to v_fma_f64).
//
// Reference arithmetic being reproduced: /root/reference/src/sipnet/sipnet.c:1256-1336,
// :1420-1496, :1546-1680, :1688-1767 (citations relative to /root/reference/src/).
#include <hip/hip_runtime.h>

#include <cstdio>

#include "fast_math.h"
#include "step_kernel.h"

namespace sipnet {
namespace {

// Diagnostic build only (-DSIPNET_STAMPS): s_memtime stamps around the segments of a step,
// summed per workgroup 0 into g_stamps.  Never compiled into the shipped library.
#ifdef SIPNET_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(k)                                                                     \
  {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                               \
    unsigned long long now_;                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);                                               \
    stampAcc##k += now_ - lastStamp;                                                 \
    lastStamp = now_;                                                                \
  }
#else
#define STAMP(k)
#endif

}  // namespace

// -------------------------------------------------------------------------------
// A lone wavefront cannot hide the instruction-fetch restart after a taken branch (measured:
// ~31 branch instructions per step cost about a third of the step).  The step is therefore
// written branch-light: per-lane conditions become selects, rare paths (events, phenology
// switches, mortality, irregular ring steps) hide behind ONE wave-uniform test each, and the
// only common-path branches are the day/night test, the snow / bare-soil evaporation split
// and the Q10 reuse test.
//
// Generic = true adds the optional-flag arithmetic of the reference (nitrogen.c:15-239,
// limitations.c:69-139, depeffects.c:23-96, sipnet.c:1084-1103, :1150-1171, :1201-1214,
// :1645-1668) under wave-uniform run-time flags, with the same conventions: reciprocals of the
// per-member C:N ratios hoisted out of the loop, divisions by pools through v_rcp + Newton.
//
// Mode: kFlagsDefault = the reference's default flag set compiled in; kFlagsRuntime = flags
// read from the launch arguments (wave-uniform branches); kFlagsNCycle = the nitrogen-cycle
// configuration (litter pool + anaerobic + nitrogen cycle on top of the defaults -- what
// nitrogen-cycle requires, context.c:203-212) compiled in, so that its flag tests cost no
// branches either.
enum : int { kFlagsDefault = 0, kFlagsRuntime = 1, kFlagsNCycle = 2 };
template <int Mode>
struct FastFlags {
  bool gdd, growthResp, leafWater, litterPool, soilPhenol, waterHResp, nitrogen, anaerobic,
      flooding, carbonSat;
  __device__ explicit FastFlags(const int32_t* f) {
    auto get = [&](int i, bool dflt, bool nset) {  // defaults: context.c:35-53
      return Mode == kFlagsRuntime ? (f[i] != 0) : (Mode == kFlagsNCycle ? nset : dflt);
    };
    gdd = get(SIPNET_F_GDD, true, true);
    growthResp = get(SIPNET_F_GROWTH_RESP, false, false);
    leafWater = get(SIPNET_F_LEAF_WATER, false, false);
    litterPool = get(SIPNET_F_LITTER_POOL, false, true);
    soilPhenol = get(SIPNET_F_SOIL_PHENOL, false, false);
    waterHResp = get(SIPNET_F_WATER_HRESP, true, true);
    nitrogen = get(SIPNET_F_NITROGEN_CYCLE, false, true);
    anaerobic = get(SIPNET_F_ANAEROBIC, false, true);
    flooding = get(SIPNET_F_FLOODING, false, false);
    carbonSat = get(SIPNET_F_CARBON_SATURATION, false, false);
  }
};
bool isNCycleFlagSet(const int32_t* f) {
  for (int i = 0; i < SIPNET_NFLAGS; i++) {
    if (i == SIPNET_F_SNOW || isPhenologyOrEventsFlag(i)) continue;  // (data, not code: see isDefaultFlagSet)
    const bool want = i == SIPNET_F_WATER_HRESP || i == SIPNET_F_LITTER_POOL || i == SIPNET_F_NITROGEN_CYCLE ||
                      i == SIPNET_F_ANAEROBIC;
    if ((f[i] != 0) != want) return false;
  }
  return true;
}

// Occ = wavefronts per SIMD the register budget is cut for: 1 (up to 512 VGPRs: the fp64
// instantiations take 257-400) or 2 (at most 256 VGPRs; the fp64 default-flag kernel then
// spills two registers).  With more chunks than SIMDs two resident waves hide each other's
// issue gaps (fp64, 131 072 members: 77 vs 61 G steps/s); with at most one chunk per SIMD the
// tighter budget only costs (19.6 vs 18.7 ms at 65 536 members), so the launcher picks.  (fp32:
// 191-198 VGPRs, two waves as it is; a 168-VGPR cut for three gains 13 % only beyond 500 000
// members and a 128-VGPR cut for four halves the rate through spills -- neither is built.)
// Full = true: the "complete" variant for callers that want more than the three flux planes --
// every accumulator of the restart schema advances (trackers.tot*, trackers.yearly*, sipnet.c:
// 1420-1496), the 44-column per-step record of the strict kernel can be written (a.rec: `.out` /
// `events.out` text and checkpoints from the throughput path), and the reference's per-step
// diagnostics are counted per member (a.diag: clamp warnings sipnet.c:1346-1356, mass-balance
// warnings balance.c:40-169).  Same flux arithmetic; the lean variant stays the benchmarked one.
template <class R, bool PlainExp, int Mode, int Occ, bool Full>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(Occ)))
void stepFastKernel(FastArgs a) {
  constexpr bool Generic = Mode != kFlagsDefault;
  const FastFlags<Mode> F(a.flags);
  // LDS: two tiles of kFastTile site records (2 x 4 KB).  ONE __shared__ object.
  __shared__ alignas(16) unsigned char lds[2 * kFastTile * sizeof(FastRec)];

  const int chunksPerSite = (a.n_members + 63) >> 6;
  int site, chunk;
  {
    const int b = (int)blockIdx.x;
    if ((a.n_sites & 7) == 0) {  // keep a site's chunks on one XCD group (speed only)
      const int g = b & 7, j = b >> 3;
      site = g + 8 * (j / chunksPerSite);
      chunk = j % chunksPerSite;
    } else {
      site = b / chunksPerSite;
      chunk = b % chunksPerSite;
    }
  }
  const int lane = (int)threadIdx.x;
  int m = (chunk << 6) + lane;
  // lanes past the end of the site keep running on a clamped column (they are needed for the
  // cooperative tile copies) but never store
  const bool live = m < a.n_members;
  if (!live) m = a.n_members - 1;
  const int64_t col = (int64_t)site * a.n_members + m;
  // ring-eviction and event indices in the site's records are local to the site
  const int opBase = uni(a.siteBase[3 * site]), evBase = uni(a.siteBase[3 * site + 1]);
  const int siteSteps = uni(a.siteBase[3 * site + 2]);   // records of THIS site (sites of a batch may differ in length)
  const int64_t nc = a.ncol;

  double* __restrict__ stp = a.state + col;
  const bool skip = stp[(int64_t)ST_status * nc] != 0.0;
  const bool act = live && !skip;

  // ---- per-member constants ---------------------------------------------------
  // (a particle filter's batch keeps its parameters where set_params put them and resamples an index: batch_impl.h)
  // (... or, a filter spread over ranks, in the bank of all ranks' parameters, whose rows are prmPitch columns long)
  const double* __restrict__ pp = a.prm + (a.prmId ? (int64_t)a.prmId[col] : col);
  const int64_t pnc = a.prmPitch;
#define PRM(name) (pp[(int64_t)SP_##name * pnc])
  const double leafCSpWt = PRM(leafCSpWt);
  const double convK = kCWeight * (1.0 / kTen9) * (leafCSpWt / PRM(cFracLeaf)) * kSecPerDay;
  const double respPerGram = PRM(baseFolRespFrac) * PRM(aMax);
  const R K_g = (R)((PRM(aMax) * PRM(aMaxFrac) + respPerGram) * convK);
  const R K_rpg = (R)(respPerGram * convK);
  const R K_invLcsw = (R)(1.0 / leafCSpWt);
  const R K_tmin = (R)PRM(psnTMin), K_tmax = (R)PRM(psnTMax);
  const R K_invDen = (R)(1.0 / (((PRM(psnTMax) - PRM(psnTMin)) / 2.0) * ((PRM(psnTMax) - PRM(psnTMin)) / 2.0)));
  const R K_slope = (R)PRM(dVpdSlope), K_vexp = (R)PRM(dVpdExp);
  const R K_attl = (R)(-PRM(attenuation) * (1.0 / 6.0) * kLog2e);
  const R K_invHalf = (R)(1.0 / PRM(halfSatPar));
  const R K_tr = (R)(1000.0 * (44.0 / 12.0) * (1.0 / 10000.0) / PRM(wueConst));
  const R K_whc = (R)PRM(soilWHC), K_invWhc = (R)(1.0 / PRM(soilWHC));
  const R K_wrf = (R)PRM(waterRemoveFrac);
  const R K_frozThr = (R)PRM(frozenSoilThreshold), K_frozEff = (R)PRM(frozenSoilEff);
  const R K_frozFolEff = (R)PRM(frozenSoilFolREff);
  const R K_immed = (R)PRM(immedEvapFrac), K_ff = (R)PRM(fastFlowFrac);
  const R K_invRd = (R)(1.0 / PRM(rdConst)), K_rd = (R)PRM(rdConst), K_melt = (R)PRM(snowMelt);
  const R K_c1l = (R)(PRM(rSoilConst1) * kLog2e), K_c2l = (R)(PRM(rSoilConst2) * kLog2e);
  const R K_lgVeg = (R)log2(PRM(vegRespQ10)), K_lgSoil = (R)log2(PRM(soilRespQ10));
  const R K_lgFine = (R)log2(PRM(fineRootQ10)), K_lgCoarse = (R)log2(PRM(coarseRootQ10));
  const R K_folShift = (R)exp2(-(PRM(psnTOpt) / 10.0) * log2(PRM(vegRespQ10)));
  const R K_bvr = (R)PRM(baseVegResp), K_bsr = (R)PRM(baseSoilResp);
  const R K_bfr = (R)PRM(baseFineRootResp), K_bcr = (R)PRM(baseCoarseRootResp);
  const R K_wtr = (R)PRM(woodTurnoverRate), K_ltr = (R)PRM(leafTurnoverRate);
  const R K_frt = (R)PRM(fineRootTurnoverRate), K_crt = (R)PRM(coarseRootTurnoverRate);
  const R K_la = (R)PRM(leafAllocation), K_wa = (R)PRM(woodAllocation);
  const R K_fa = (R)PRM(fineRootAllocation), K_ca = (R)PRM(coarseRootAllocation);
  const R K_moistExp = (R)PRM(soilRespMoistEffect);
  // leaf-on test "x >= threshold" (sipnet.c:705-731): x is the year-to-date GDD, the soil
  // temperature or the day of year, by flag; a non-positive leafOnDay never fires
  // (the compiled-in flag sets leave the phenology mode to the launch: the plan puts the matching variable
  // into the record's cumGdd field)
  // (with the gdd flag off convertParamsKernel has put the soil-temperature / day-of-year threshold into this row)
  const double gddLeafOn = PRM(gddLeafOn);
  // Optional-flag parameters (Generic only; dead code otherwise).  A taken branch costs a lone
  // wavefront an instruction-fetch restart, so the small options are not branched around: with
  // the flag off their parameter takes a neutral value (rate 0, cap "infinite") and the same
  // few instructions run to an exactly unchanged result; only the nitrogen cycle and the
  // methane pow() sit behind wave-uniform branches.
  constexpr double kNoCap = 3.0e38;  // finite in fp32 too
  const R G_growthFrac = Generic && F.growthResp ? (R)PRM(growthRespFrac) : R(0);
  const R G_leafPool = Generic ? (R)PRM(leafPoolDepth) : R(0);
  const R G_drainFrac = Generic && F.flooding ? (R)PRM(waterDrainFrac) : R(kNoCap);
  const R G_lbr = Generic && F.litterPool ? (R)PRM(litterBreakdownRate) : R(0);
  const R G_flr = Generic ? (R)PRM(fracLitterRespired) : R(0);
  const R G_nVol = Generic ? (R)PRM(nVolatilizationFrac) : R(0);
  const R G_nLeach = Generic ? (R)PRM(nLeachingFrac) : R(0);
  const R G_iLeafCN = Generic ? (R)(1.0 / PRM(leafCN)) : R(0);
  const R G_iWoodCN = Generic ? (R)(1.0 / PRM(woodCN)) : R(0);
  const R G_iFineCN = Generic ? (R)(1.0 / PRM(fineRootCN)) : R(0);
  const R G_kCN = Generic ? (R)PRM(kCN) : R(0);
  const R G_nFixMax = Generic ? (R)PRM(nFixationFracMax) : R(0);
  const R G_halfNFix = Generic ? (R)PRM(halfNFixationMax) : R(0);
  const R G_resorb = Generic ? (R)PRM(leafNResorptionFrac) : R(0);
  const R G_fAnox = Generic ? (R)PRM(fAnoxia) : R(0);
  const R G_iFAnox = Generic ? (R)(1.0 / PRM(fAnoxia)) : R(0);
  const R G_iOneMinusAnox = Generic ? (R)(1.0 / (1.0 - PRM(fAnoxia))) : R(0);
  const R G_anDecomp = Generic ? (R)PRM(anaerobicDecompRate) : R(0);
  const R G_anExp = Generic ? (R)PRM(anaerobicTransExp) : R(0);
  const R G_soilCH4 = Generic && F.anaerobic ? (R)PRM(soilMethaneRate) : R(0);
  const R G_litCH4 = Generic && F.anaerobic && F.litterPool ? (R)PRM(litterMethaneRate) : R(0);
  const R G_iSoilCSat = Generic ? (R)(1.0 / PRM(soilCSaturation)) : R(0);
  const double leafOffDay = PRM(leafOffDay) > 0 ? PRM(leafOffDay) : 1e300;  // "never" (sipnet.c:735)
  // rarely needed parameters are re-read from HBM inside their (rare) branches
#define PRM_RARE(name) ((R)pp[(int64_t)SP_##name * pnc])

  const Exp2Coef EC = loadExp2Coef();

  // ---- carried state ----------------------------------------------------------
#define ST(name) stp[(int64_t)ST_##name * nc]
  double plantWoodC = ST(plantWoodC), plantLeafC = ST(plantLeafC), soilC = ST(soilC);
  double soilWater = ST(soilWater), snow = ST(snow);
  double coarseRootC = ST(coarseRootC), fineRootC = ST(fineRootC);
  double delta = ST(plantCAccountingDelta);
  double litterC = Generic ? ST(litterC) : 0.0, minN = Generic ? ST(minN) : 0.0;
  double soilOrgN = Generic ? ST(soilOrgN) : 0.0, litterN = Generic ? ST(litterN) : 0.0;
  double storN = Generic ? ST(plantStorageN) : 0.0;
  double ringSum = ST(ringSum), totNee = ST(totNee), totGpp = ST(totGpp);
  [[maybe_unused]] double pfNee = 0.0;
  int phenBits = (int)ST(phenBits);
  int ringValidFrom = (int)ST(ringValidFrom);
  int diedAt = (int)ST(diedAt);
  // Full: the other accumulators of updateTrackers() and the diagnostics counters
  double totRtot = Full ? ST(totRtot) : 0.0, totRa = Full ? ST(totRa) : 0.0;
  double totRh = Full ? ST(totRh) : 0.0, totNpp = Full ? ST(totNpp) : 0.0;
  double yGpp = Full ? ST(yearlyGpp) : 0.0, yRtot = Full ? ST(yearlyRtot) : 0.0;
  double yRa = Full ? ST(yearlyRa) : 0.0, yRh = Full ? ST(yearlyRh) : 0.0;
  double yNpp = Full ? ST(yearlyNpp) : 0.0, yNee = Full ? ST(yearlyNee) : 0.0;
  double yLitter = Full ? ST(yearlyLitter) : 0.0;
  const bool wantDiag = Full && a.diag != nullptr;
  const double K_whc2 = Full ? 2.0 * PRM(soilWHC) : 0.0;  // soilWetnessFrac denominator, sipnet.c:1470
  int clampWarn = 0, balanceWarn = 0;
  double maxDC = 0.0, maxDN = 0.0;
  double* __restrict__ recp = Full && a.rec ? a.rec + col : nullptr;
#ifdef SIPNET_STAMPS
  unsigned long long stampAcc0 = 0, stampAcc1 = 0, stampAcc2 = 0, stampAcc3 = 0, stampAcc4 = 0,
                     stampAcc5 = 0, stampAcc6 = 0, stampAcc7 = 0, lastStamp;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lastStamp)::"memory");
#endif

  const unsigned char* __restrict__ planBytes =
      (const unsigned char*)(a.fast + (int64_t)site * a.n_steps_total);
  // the ring holds NPP values of type R (fp32-mixed batches: fp32 numbers, stored as such)
  R* __restrict__ ringp = (R*)a.ring + col;
  // Output planes: a plane the caller does not want is pointed at a one-row scratch buffer
  // with stride 0, so the stores below need no test.  Lanes past the end of a site work on a
  // copy of the site's last member and store identical values to its addresses; members with
  // a non-zero status compute garbage, so their plane entries are undefined (documented).
  R* __restrict__ oNee = (R*)(a.nee ? a.nee : a.scratchRow) + col;
  R* __restrict__ oGpp = (R*)(a.gpp ? a.gpp : a.scratchRow) + col;
  R* __restrict__ oEt = (R*)(a.et ? a.et : a.scratchRow) + col;
  const int64_t ldNee = a.nee ? a.ld : 0, ldGpp = a.gpp ? a.ld : 0, ldEt = a.et ? a.ld : 0;

  // ---- tile staging: async global -> LDS, 16 B per lane, 4 pieces per 4 KB tile ----
  constexpr int kTileBytes = kFastTile * (int)sizeof(FastRec);
  auto tileFirst = [&](int tile) -> int64_t {
    // records [tile*kFastTile, +kFastTile) clamped to the plan's end (a tail tile re-reads
    // earlier records so that it never runs past the site's plan)
    int64_t first = (int64_t)tile * kFastTile;
    const int64_t lastStart = (int64_t)a.n_steps_total - kFastTile;
    if (first > lastStart) first = lastStart > 0 ? lastStart : 0;
    return first;
  };
  auto stageTile = [&](int tile, int buf) {
    const unsigned char* src = planBytes + tileFirst(tile) * (int64_t)sizeof(FastRec);
#pragma unroll
    for (int k = 0; k < kTileBytes / 1024; k++) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + k * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(lds + buf * kTileBytes + k * 1024), 16, 0, 0);
    }
  };

  const int tBegin = a.step0, tEnd = a.step0 + a.n_steps < siteSteps ? a.step0 + a.n_steps : siteSteps;
  if (tBegin >= tEnd) return;   // this site's forcing ended before the launch's range (wave-uniform; no barrier has been met)
  const uint32_t ncu = (uint32_t)nc;  // ring element offsets fit 32 bits (250 * ncol < 2^31)
  int curTile = tBegin / kFastTile;
  stageTile(curTile, curTile & 1);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  // ring values a step evicts are requested at the END of the previous step (ahead of that
  // step's stores in the memory queue) and consumed a whole step later; the first step's are
  // requested here
  R rv0, rv1;
  {
    const int32_t slots0 = uni(*(const int32_t*)(lds + (curTile & 1) * kTileBytes +
                                                 (int)(tBegin - tileFirst(curTile)) * (int)sizeof(FastRec) + 132));
    rv0 = ringp[(uint32_t)(slots0 & 255) * ncu];
    rv1 = ringp[(uint32_t)((slots0 >> 8) & 255) * ncu];
  }
  // when the slot a step evicts is the very slot the previous step wrote, the value is taken
  // from that step's NPP register instead of memory (wave-uniform flags)
  double lastNpp = 0.0;
  bool useLast0 = false, useLast1 = false;
  // Q10 factors of the soil temperature, reused while tsoil does not change
  R qSoil = 0, qFine = 0, qCoarse = 0;
  bool haveQ = false;

  for (int tileStart = curTile * kFastTile; tileStart < tEnd; tileStart += kFastTile, curTile++) {
    if (tileStart > tBegin) {
      // the tile staged one tile-time ago has long landed; drain before reading it
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
    }
    stageTile(curTile + 1, (curTile + 1) & 1);  // into the buffer the previous tile vacated
    const int tFirst = tileStart > tBegin ? tileStart : tBegin;
    const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
    const unsigned char* recB = lds + (curTile & 1) * kTileBytes +
                                (int)(tFirst - tileFirst(curTile)) * (int)sizeof(FastRec);
  for (int t = tFirst; t < tLast; t++, recB += sizeof(FastRec)) {
    // ---- the hot 144 bytes of the site record: nine broadcast LDS reads --------------
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    const d2* rq = (const d2*)recB;
    const d2 q0 = rq[0], q1 = rq[1], q2 = rq[2], q3 = rq[3], q4 = rq[4], q5 = rq[5];
    const d2 q6 = rq[6], q7 = rq[7];
    const i4 j0 = *(const i4*)(recB + 128);
    const double* rare = (const double*)(recB + 144);   // w1 - log2vpd gddAfter tillAfter
    const int32_t* rareI = (const int32_t*)(recB + 184);  // ins0 ins1 opFirst evFirst

    const R len = (R)q0.x, invLen = (R)q0.y, tair = recR<R>(q1.x), tsoil = recR<R>(q1.y);
    const int bits = uni(j0.x);
    const int slots = uni(j0.y);
    const int insSlot = uni(j0.z);
    const int nEv = uni(j0.w);

    STAMP(0)
    // ---- 0. start of step (sipnet.c:1821-1828) -------------------------------------
    const bool alive0 = (plantWoodC > kTiny) && (plantWoodC + delta > kTiny) &&
                        (fineRootC + coarseRootC > kTiny);
    const R eWood = (R)plantWoodC, eLeaf = (R)plantLeafC, eSoilC = (R)soilC;
    const R eWater = (R)soilWater, eSnow = (R)snow;
    const R eCoarse = (R)coarseRootC, eFine = (R)fineRootC;
    const R totalWoodC = (R)(plantWoodC + delta);
    const R eLitter = (R)litterC, eMinN = (R)minN, eSoilOrgN = (R)soilOrgN;
    const R eLitterN = (R)litterN, eStorN = (R)storN;
    const double oldSoilWater = soilWater;  // soilWetnessFrac, sipnet.c:1470
    // getMassTotals() before the pool updates, balance.c:13-36
    auto massC = [&]() -> double {
      double c = (plantWoodC + delta) + plantLeafC + fineRootC + coarseRootC + soilC;
      if (Generic && F.litterPool) c += litterC;
      return c;
    };
    auto massN = [&]() -> double {
      if (!(Generic && F.nitrogen)) return 0.0;
      return plantWoodC * (double)G_iWoodCN + plantLeafC * (double)G_iLeafCN +
             fineRootC * (double)G_iFineCN + coarseRootC * (double)G_iWoodCN + soilOrgN + litterN +
             minN + storN;
    };
    double preC = 0.0, preN = 0.0;
    if (wantDiag) {
      preC = massC();
      preN = massN();
    }
    // Full: event-log columns of the record and the events' contribution to the mass balance
    R recLeafOffComputed = 0, recEvLeafOn = 0, recEvLeafOnFromWood = 0, recEvLeafOffLitterAll = 0;
    R evInC = 0, evOutC = 0, evInN = 0, evOutN = 0;

    auto leafOnNFromC = [&](R leafOnC) -> R {  // nitrogen.c:84-86
      return rmax0(leafOnC * G_iLeafCN - leafOnC * G_iWoodCN);
    };
    auto leafOnLimit = [&](R flux) -> R {  // limitations.c:13-64
      const R cDemand = flux * len;
      if (cDemand < R(kTiny)) return flux;
      R lim = fdiv((eWood + eCoarse) * PRM_RARE(leafOnReallocFrac), cDemand);
      if (Generic && F.nitrogen) {
        const R nDemand = leafOnNFromC(cDemand);
        if (nDemand > R(kTiny)) lim = rminv(lim, fdiv(eStorN, nDemand));
      }
      lim = clip01(lim);
      return lim < R(1) ? flux * lim : flux;
    };

    // ---- 2. fluxes (sipnet.c:1256-1336) ---------------------------------------------
    const R lai = eLeaf * K_invLcsw;
    const R baseFolResp = K_rpg * lai;
    const bool frozen = tsoil < K_frozThr;
    // potPsn() + calcLightEff() + moisture(), sipnet.c:517-699: daytime only (uniform test);
    // at night par = 0 makes potGrossPsn = 0, so transpiration = 0 and GPP = 0
    R transpiration = 0, photosynthesis = 0;
    if (bits & FAST_PAR_POS) {
      const R dTemp = rmax0((K_tmax - tair) * (tair - K_tmin) * K_invDen);
      R vpdPow = recR<R>(q2.y) * recR<R>(q2.y);
      if (!PlainExp) vpdPow = (K_vexp == R(2)) ? vpdPow : fexp2(K_vexp * recR<R>(rare[2]), EC);
      const R dVpd = rmax0(R(1) - K_slope * vpdPow);
      // Simpson over 7 canopy layers: sum c_i (1 - e_i) / 18 = 1 - (sum c_i e_i) / 18,
      // c = 1 4 2 4 2 4 1.  lai = 0 needs no special case: potGrossPsn carries the factor lai.
      const R r1 = fexp2(K_attl * lai, EC);
      const R q = recR<R>(q2.x) * K_invHalf;
      const R r2 = r1 * r1, r3 = r2 * r1, r4 = r2 * r2, r5 = r4 * r1, r6 = r3 * r3;
      const R e0 = fexp2(q, EC), e1 = fexp2(q * r1, EC), e2 = fexp2(q * r2, EC), e3 = fexp2(q * r3, EC);
      const R e4 = fexp2(q * r4, EC), e5 = fexp2(q * r5, EC), e6 = fexp2(q * r6, EC);
      const R s = (e0 + e6) + R(4) * (e1 + e3 + e5) + R(2) * (e2 + e4);
      const R dLight = R(1) - s * R(1.0 / 18.0);
      const R potGrossPsn = K_g * lai * dTemp * dVpd * dLight;
      // moisture(), sipnet.c:656-699, branch-free
      const R potTrans = potGrossPsn * recR<R>(q2.y) * K_tr;
      R removable = rminv(eWater, K_whc) * K_wrf;
      removable = frozen ? removable * K_frozEff : removable;
      const bool hasPsn = potGrossPsn >= R(kTiny);
      const bool limited = removable < potTrans;
      const R dWater = fdiv(removable, potTrans);  // only used where limited && hasPsn
      transpiration = hasPsn ? (limited ? removable : potTrans) : R(0);
      photosynthesis = (hasPsn && limited) ? potGrossPsn * dWater : potGrossPsn;
    }
    STAMP(1)

    // calcPrecip(), sipnet.c:848-882 (the site's air temperature decides rain or snow)
    const bool tairPos = (bits & FAST_TAIR_POS) != 0;
    const R rate = recR<R>(q3.y);
    const R rain = tairPos ? rate : R(0), snowFall = tairPos ? R(0) : rate;
    R immedEvap = rain * K_immed;
    if (Generic) immedEvap = rminv(immedEvap, F.leafWater ? lai * G_leafPool : R(kNoCap));  // sipnet.c:872-878
    const R netRain = rain - immedEvap;

    // snowPack() sipnet.c:888-946 and bare-soil evaporation sipnet.c:984-1016: a member either
    // has a snow pack or evaporates from the soil
    R snowMelt = 0, sublimation = 0, evaporationPot = 0;
    const bool hasSnow = eSnow > R(0);
    if (hasSnow) {
      R subl = rmax0(recR<R>(q4.x) * K_invRd);
      R remaining = eSnow + snowFall * len;
      const bool allGone = remaining - subl * len < R(0);
      subl = allGone ? remaining * invLen : subl;
      remaining = allGone ? R(0) : remaining - subl * len;
      R melt = tairPos ? K_melt * tair : R(0);
      melt = (tairPos && (remaining - melt * len < R(0))) ? remaining * invLen : melt;
      sublimation = subl;
      snowMelt = melt;
    } else {
      const R wf = clip01(eWater * K_invWhc);
      const R rsoil = fexp2(K_c1l - K_c2l * wf, EC);
      evaporationPot = rmax0(fdiv(recR<R>(q4.y), K_rd * recR<R>(q5.x) + rsoil));
    }
    // calcSoilWaterFluxes(), sipnet.c:963-1031
    R evaporation, drainage, fastFlow;
    {
      R netIn = netRain + snowMelt;
      fastFlow = netIn * K_ff;
      netIn -= fastFlow;
      R remaining = eWater + netIn * len - transpiration * len;
      const bool dryOut = !hasSnow && (remaining - evaporationPot * len < R(kTiny));
      evaporation = dryOut ? (remaining - R(kTiny)) * invLen : evaporationPot;
      remaining = hasSnow ? remaining : (dryOut ? R(0) : remaining - evaporationPot * len);
      drainage = remaining > K_whc ? (remaining - K_whc) * invLen : R(0);
      if (Generic) {  // flooding, sipnet.c:1019-1027 (no cap with the flag off)
        const R excess = remaining - K_whc;
        drainage = remaining > K_whc ? rminv(excess * G_drainFrac, excess * invLen) : R(0);
      }
    }
    STAMP(2)

    const R meanNpp = (R)(ringSum * 0.2);  // runmean.c:119-121 (sum / 5)

    // vegResp(), sipnet.c:1051-1068
    const R vegQ = fexp2(recR<R>(q5.y) * K_lgVeg, EC);
    R folResp = baseFolResp * (vegQ * K_folShift);
    folResp = frozen ? folResp * K_frozFolEff : folResp;
    R rVeg = folResp + K_bvr * totalWoodC * vegQ;
    if (Generic) rVeg += rmax0(G_growthFrac * meanNpp);  // vegResp2(), sipnet.c:1084-1103 (+0 when off)

    // calcWoodAndLeafFluxes(), sipnet.c:756-782
    const R woodLitter = totalWoodC * K_wtr;
    R leafLitter = eLeaf * K_ltr;
    R leafCreation = meanNpp * K_la, woodCreation = meanNpp * K_wa;

    // calcLeafOnOffFluxes(), sipnet.c:800-842 (GDD phenology, sipnet.c:705-716): the two
    // switches fire once a year each and only feed the pools; they are handled together with
    // the events in the one rare block below
    R leafOnCreation = 0, leafOnFromWood = 0;
    if (bits & FAST_PHEN_NEW_YEAR) phenBits = 0;
    const double phenX = q6.y;   // year-to-date GDD, soil temperature or day of year: the plan's choice by flag
    const bool doOn = !(phenBits & 1) && phenX >= gddLeafOn;
    const bool doOff = !(phenBits & 2) && q7.x >= leafOffDay;

    // roots, sipnet.c:1176-1196; soil-temperature Q10 factors (depeffects.c:71-74)
    const R coarseRootLoss = K_crt * eCoarse, fineRootLoss = K_frt * eFine;
    R coarseRootCreation = K_ca * meanNpp, fineRootCreation = K_fa * meanNpp;
    if (!haveQ || !(bits & FAST_TSOIL_SAME)) {
      const R tsoil10 = recR<R>(q6.x);
      qSoil = fexp2(tsoil10 * K_lgSoil, EC);
      qFine = fexp2(tsoil10 * K_lgFine, EC);
      qCoarse = fexp2(tsoil10 * K_lgCoarse, EC);
      haveQ = true;
    }
    const R rCoarseRoot = K_bcr * eCoarse * qCoarse;
    const R rFineRoot = K_bfr * eFine * qFine;

    // calcSoilRespiration(), sipnet.c:1132-1148 with depeffects.c:23-87
    const R fWhc = clip01(eWater * K_invWhc);
    R moistEff = fWhc;
    if (!PlainExp && __builtin_amdgcn_ballot_w64(K_moistExp != R(1)) != 0)  // pow only where some member needs it
      moistEff = (K_moistExp == R(1)) ? moistEff : fpow(moistEff, K_moistExp);
    R anoxic = 0;  // anaerobic share A of depeffects.c:46-57, :89-96
    if (Generic) {
      anoxic = F.anaerobic ? clip01((fWhc - G_fAnox) * G_iOneMinusAnox) : R(0);
      const R anMoist = (R(1) - anoxic) * clip01(fWhc * G_iFAnox) + G_anDecomp * anoxic;
      moistEff = F.anaerobic ? anMoist : moistEff;
    }
    moistEff = (bits & FAST_TSOIL_NEG) ? R(1) : moistEff;   // (frozen soil, or the water_hresp flag off: the plan's bit)
    R rSoil = eSoilC * K_bsr * moistEff * qSoil * recR<R>(q3.x);
    // optional pools: calcLitterFluxes() sipnet.c:1150-1171, C:N effect depeffects.c:78-87,
    // calcMethaneFlux() sipnet.c:1201-1214
    R rLitter = 0, litterToSoil = 0, soilMethane = 0, litterMethane = 0;
    R denLitterN = 0, denSoilN = 0;
    if (Generic) {
      R cnSoil = 1, cnLitter = 1;
      if (F.nitrogen) {
        // cn = kCN / (kCN + C/N) = kCN N / (kCN N + C), N floored at TINY (util.c:72-75)
        denLitterN = eLitterN < R(kTiny) ? R(kTiny) : eLitterN;
        denSoilN = eSoilOrgN < R(kTiny) ? R(kTiny) : eSoilOrgN;
        cnSoil = fdiv(G_kCN * denSoilN, G_kCN * denSoilN + eSoilC);
        cnLitter = fdiv(G_kCN * denLitterN, G_kCN * denLitterN + eLitter);
      }
      rSoil *= cnSoil;
      // without a litter pool the pool is empty and its rate 0: both fluxes come out as 0
      const R breakdown = eLitter * G_lbr * qSoil * moistEff * recR<R>(q3.x) * cnLitter;
      rLitter = breakdown * G_flr;
      litterToSoil = breakdown * (R(1) - G_flr);
      R mMoist = anoxic * anoxic;  // pow(A, anaerobicTransExp) with the usual exponent 2
      if (__builtin_expect(F.anaerobic && __builtin_amdgcn_ballot_w64(G_anExp != R(2)) != 0, 0)) {
        const bool general = G_anExp != R(2) && (anoxic > R(0) || G_anExp <= R(0));
        mMoist = general ? fpow(anoxic, G_anExp) : (G_anExp != R(2) ? R(0) : mMoist);
      }
      soilMethane = G_soilCH4 * eSoilC * qSoil * mMoist;  // rates are 0 with the flag off
      litterMethane = G_litCH4 * eLitter * qSoil * mMoist;
    }

    // checkNegativeCreation(), limitations.c:146-182, as selects
    {
      const R leafDeficit = eLeaf * invLen + leafCreation - eLeaf * K_ltr;
      const R ld = rminv(leafDeficit, R(0));
      woodCreation += ld;
      leafCreation -= ld;
      const R fineDef = eFine * invLen + fineRootCreation - fineRootLoss;
      const R coarseDef = eCoarse * invLen + coarseRootCreation - coarseRootLoss;
      const bool fNeg = fineDef < R(0), cNeg = coarseDef < R(0);
      const R shift = (fNeg != cNeg) ? (fNeg ? fineDef : -coarseDef) : R(0);
      coarseRootCreation += shift;   // fine deficit is taken from coarse roots (shift < 0) ...
      fineRootCreation -= shift;     // ... a coarse deficit from fine roots (shift > 0)
    }
    STAMP(3)

    // ---- 1+3a. events (events.c:449-742) and their pool updates (events.c:744-790).  Event
    // fluxes only depend on the start-of-step pools and, without the N cycle, feed nothing but
    // the pools and ET, so they are evaluated here, off the common path.  Tillage is folded
    // into the plan.
    R evEvap = 0;
    R evMinN = 0, evLeafOnTotal = 0;  // event fluxes the N limitations look at (Generic)
    if (__builtin_expect(nEv > 0 || __builtin_amdgcn_ballot_w64(doOn || doOff) != 0, 0)) {
      if (doOn) {
        const R leafOn = leafOnLimit(PRM_RARE(leafGrowth) * invLen);
        leafOnCreation = leafOn;
        const R src = eWood + eCoarse;
        if (src > R(kTiny)) leafOnFromWood = fdiv(leafOn * eWood, src);
        phenBits |= 1;
      }
      if (doOff) {
        const R off = (eLeaf * PRM_RARE(fracLeafFall)) * invLen;
        leafLitter += off;
        if (Full) recLeafOffComputed = off;
        phenBits |= 2;
      }
      R evLeafC = 0, evWoodC = 0, evFineRootC = 0, evCoarseRootC = 0, evSoilWater = 0;
      R evSoilC = 0, evLeafOnCreation = 0, evLeafOnFromWood = 0, evLeafOffLitter = 0;
      R evLitterC = 0, evSoilOrgN = 0, evLitterN = 0, evLeafOffNResorp = 0;
      const bool toLitter = Generic && F.litterPool;
      const bool withN = Generic && F.nitrogen;
      const int ev0 = uni(rareI[3]);
      for (int k = 0; k < nEv; k++) {
        const EvRec& ev = a.events[evBase + ev0 + k];
        const int type = uni(ev.type);
        const R p0 = (R)ev.p[0], p1 = (R)ev.p[1], p2 = (R)ev.p[2], p3 = (R)ev.p[3];
        if (type == SIPNET_EV_IRRIG) {
          const R evapAmount = ((int)ev.p[1] == 0) ? K_immed * p0 : R(0);
          evEvap += evapAmount * invLen;
          evSoilWater += (p0 - evapAmount) * invLen;
        } else if (type == SIPNET_EV_PLANT) {
          evLeafC += p0 * invLen;
          evWoodC += p1 * invLen;
          evFineRootC += p2 * invLen;
          evCoarseRootC += p3 * invLen;
          if (Full) {  // events.c:530-541
            evInC += (p0 + p1 + p2 + p3) * invLen;
            if (withN) evInN += (p0 * G_iLeafCN + p1 * G_iWoodCN + p2 * G_iFineCN + p3 * G_iWoodCN) * invLen;
          }
        } else if (type == SIPNET_EV_HARVEST) {
          const R woodC = totalWoodC;
          if (Full) {  // events.c:582-594
            evOutC += ((woodC + eLeaf) * p0 + (eFine + eCoarse) * p1) * invLen;
            if (withN)
              evOutN += ((eWood * G_iWoodCN + eLeaf * G_iLeafCN) * p0 +
                         (eFine * G_iFineCN + eCoarse * G_iWoodCN) * p1) * invLen;
          }
          if (toLitter) {  // events.c:575-580
            evLitterC += (p2 * (eLeaf + woodC)) * invLen;
            evSoilC += (p3 * (eFine + eCoarse)) * invLen;
          } else {
            evSoilC += (p3 * (eFine + eCoarse) + p2 * (eLeaf + woodC)) * invLen;
          }
          if (withN) {  // events.c:596-620
            evSoilOrgN += (p3 * (eFine * G_iFineCN + eCoarse * G_iWoodCN)) * invLen;
            evLitterN += (p2 * (eLeaf * G_iLeafCN + eWood * G_iWoodCN)) * invLen;
          }
          evLeafC += -eLeaf * (p0 + p2) * invLen;
          evWoodC += -woodC * (p0 + p2) * invLen;
          evFineRootC += -eFine * (p1 + p3) * invLen;
          evCoarseRootC += -eCoarse * (p1 + p3) * invLen;
        } else if (type == SIPNET_EV_FERT) {
          if (toLitter) evLitterC += p1 * invLen;
          else evSoilC += p1 * invLen;
          if (withN) {  // events.c:660-672
            evLitterN += p0 * invLen;
            evMinN += p2 * invLen;
          }
          if (Full) {
            evInC += p1 * invLen;
            if (withN) evInN += (p0 + p2) * invLen;
          }
        } else if (type == SIPNET_EV_LEAFON) {
          const R flux = leafOnLimit(PRM_RARE(leafGrowth) * invLen);
          evLeafOnCreation += flux;
          const R src = eWood + eCoarse;
          if (src > R(kTiny)) evLeafOnFromWood += fdiv(flux * eWood, src);
        } else if (type == SIPNET_EV_LEAFOFF) {
          const R leafOff = eLeaf * PRM_RARE(fracLeafFall);
          evLeafOffLitter += leafOff * invLen;
          if (withN) {  // events.c:712-722
            const R leafN = leafOff * G_iLeafCN;
            const R resorb = leafN * G_resorb;
            evLeafOffNResorp += resorb * invLen;
            evLitterN += (leafN - resorb) * invLen;
          }
        }
      }
      evLeafOnTotal = evLeafOnCreation;
      if (Full) {
        recEvLeafOn = evLeafOnCreation;
        recEvLeafOnFromWood = evLeafOnFromWood;
        recEvLeafOffLitterAll = evLeafOffLitter;
      }
      plantWoodC += (double)(evWoodC * len);
      plantLeafC += (double)(evLeafC * len);
      soilC += (double)(evSoilC * len);
      plantWoodC -= (double)(evLeafOnFromWood * len);
      coarseRootC -= (double)((evLeafOnCreation - evLeafOnFromWood) * len);
      plantLeafC += (double)((evLeafOnCreation - evLeafOffLitter) * len);
      if (toLitter) {
        litterC += (double)(evLitterC * len);
        litterC += (double)(evLeafOffLitter * len);
      } else {
        soilC += (double)(evLeafOffLitter * len);
      }
      coarseRootC += (double)(evCoarseRootC * len);
      fineRootC += (double)(evFineRootC * len);
      soilWater += (double)(evSoilWater * len);
      if (withN) {  // events.c:778-789
        minN += (double)(evMinN * len);
        soilOrgN += (double)(evSoilOrgN * len);
        litterN += (double)(evLitterN * len);
        storN += (double)((evLeafOffNResorp - leafOnNFromC(evLeafOnCreation)) * len);
      }
    }

    // nitrogen cycle: nitrogen.c:15-207 with limitations.c:69-139 (after the phenology switches
    // and the events, whose leaf-on and mineral-N fluxes it looks at)
    R nVolatilization = 0, nLeaching = 0, nOrgSoil = 0, nOrgLitter = 0, nMin = 0;
    R nFixation = 0, nUptake = 0, leafOffNResorption = 0, reductionNResorption = 0;
    if (Generic && F.nitrogen) {
      auto plantNDemand = [&]() -> R {  // nitrogen.c:89-104
        return rmax0(woodCreation * G_iWoodCN + leafCreation * G_iLeafCN +
                     fineRootCreation * G_iFineCN + coarseRootCreation * G_iWoodCN);
      };
      // unclaimed storage nitrogen.c:127-134, fixation share nitrogen.c:137-152
      const R unclaimed = rmax0(eStorN - leafOnNFromC(leafOnCreation + evLeafOnTotal) * len);
      const R fixDen = G_halfNFix + eMinN;
      const R fixFrac = G_nFixMax * ((fixDen < R(kTiny)) ? R(1) : fdiv(G_halfNFix, fixDen));
      auto fixationAndUptake = [&]() {  // nitrogen.c:155-168
        const R rem = rmax0(plantNDemand() - unclaimed * invLen);
        nFixation = fixFrac * rem;
        nUptake = (R(1) - fixFrac) * rem;
      };
      // resorption, nitrogen.c:170-196
      if (woodCreation + leafCreation + fineRootCreation + coarseRootCreation < R(0)) {
        reductionNResorption -= (leafCreation * G_iLeafCN + woodCreation * G_iWoodCN +
                                 coarseRootCreation * G_iWoodCN + fineRootCreation * G_iFineCN);
      }
      leafOffNResorption = G_resorb * leafLitter * G_iLeafCN;
      // volatilisation nitrogen.c:15-26, leaching nitrogen.c:31-41
      nVolatilization = G_nVol * eMinN * qSoil * (R(0.05) + R(3.8) * anoxic * (R(1) - anoxic));
      nLeaching = eMinN * rminv(drainage * K_invWhc, R(1)) * G_nLeach;
      // pool fluxes, nitrogen.c:45-82: x / (C/N) = x * N / C
      {
        const R iLitterCN = fdiv(denLitterN, eLitter);
        const R iSoilCN = fdiv(denSoilN, eSoilC);
        const R litterMin = rLitter * iLitterCN;
        const R soilMin = rSoil * iSoilCN;
        const R soilNInputs = litterToSoil * iLitterCN + fineRootLoss * G_iFineCN +
                              coarseRootLoss * G_iWoodCN;
        const R sat = F.carbonSat ? clip01(eSoilC * G_iSoilCSat) : R(0);
        nOrgLitter = leafLitter * G_iLeafCN - leafOffNResorption + woodLitter * G_iWoodCN -
                     litterMin - litterToSoil * iLitterCN + (soilNInputs * sat);
        nOrgSoil = soilNInputs * (R(1) - sat) - soilMin;
        nMin = litterMin + soilMin;
      }
      fixationAndUptake();
      // checkMineralNLimitation, limitations.c:119-129
      {
        const R pool = eMinN + (nMin + evMinN) * len;
        const R loss = (nLeaching + nVolatilization) * len;
        const R red = (loss > R(kTiny) && loss > pool) ? fdiv(pool, loss) : R(1);
        nLeaching *= red;
        nVolatilization *= red;
      }
      // checkNitrogenLimitation, limitations.c:69-114
      {
        const R uptakeDemand = nUptake * len;
        const R availableMinN = eMinN + (nMin - nVolatilization - nLeaching) * len;
        const bool limited = uptakeDemand > R(kTiny) && uptakeDemand > availableMinN;
        if (__builtin_amdgcn_ballot_w64(limited) != 0) {
          const R demand = plantNDemand() * len;
          const R red = limited ? fdiv(fdiv(availableMinN, R(1) - fixFrac) + unclaimed, demand) : R(1);
          woodCreation *= red;
          leafCreation *= red;
          fineRootCreation *= red;
          coarseRootCreation *= red;
          fixationAndUptake();  // unchanged where red = 1
        }
      }
      // updateNitrogenPools(), nitrogen.c:210-239 (the creation fluxes are final here)
      {
        const R storageDemand = plantNDemand() - nUptake - nFixation;
        storN += (double)((leafOffNResorption + reductionNResorption - storageDemand -
                           leafOnNFromC(leafOnCreation)) * len);
        minN += (double)(((nMin - nVolatilization - nLeaching) - nUptake) * len);
        soilOrgN += (double)(nOrgSoil * len);
        litterN += (double)(nOrgLitter * len);
      }
    }
    // ---- 3. pools (sipnet.c:1769-1806) ------------------------------------------------
    {
      const R r_a = rVeg + rFineRoot + rCoarseRoot;
      const R alloc = leafCreation + woodCreation + fineRootCreation + coarseRootCreation;
      delta += (double)(((photosynthesis - r_a) - alloc) * len);
      plantWoodC += (double)((woodCreation - woodLitter - leafOnFromWood) * len);
      plantLeafC += (double)((leafCreation + leafOnCreation - leafLitter) * len);
      soilWater += (double)((rain + snowMelt - immedEvap - fastFlow - evaporation -
                             transpiration - drainage) * len);
      snow += (double)((snowFall - snowMelt - sublimation) * len);
      if (Generic) {  // updatePoolsForSoil(), sipnet.c:1645-1668: both forms, one select
        const R soilInputs = coarseRootLoss + fineRootLoss + litterToSoil;
        // the soil carbon the reference looks at here already holds this step's event fluxes
        const R sat = F.carbonSat ? clip01((R)soilC * G_iSoilCSat) : R(0);
        const R dLitter = (woodLitter + leafLitter + (soilInputs * sat) - litterToSoil -
                           rLitter - litterMethane) * len;
        const R dSoilTwo = (soilInputs * (R(1) - sat) - rSoil - soilMethane) * len;
        const R dSoilOne = (coarseRootLoss + fineRootLoss + woodLitter + leafLitter - rSoil -
                            soilMethane) * len;
        litterC += (double)(F.litterPool ? dLitter : R(0));
        soilC += (double)(F.litterPool ? dSoilTwo : dSoilOne);
      } else {
        soilC += (double)((coarseRootLoss + fineRootLoss + woodLitter + leafLitter - rSoil) * len);
      }
      coarseRootC += (double)((coarseRootCreation - coarseRootLoss -
                               (leafOnCreation - leafOnFromWood)) * len);
      fineRootC += (double)((fineRootCreation - fineRootLoss) * len);
    }

    double postC = 0.0, postN = 0.0;
    if (wantDiag) {
      postC = massC();
      postN = massN();
    }
    double deathWood = 0.0, deathRoot = 0.0, diedNowRec = 0.0;  // record columns 41..43
    // checkForMortality(), sipnet.c:1688-1767: only a change of the alive flag does anything
    bool alive = alive0;
    {
      const bool sufficient = (plantWoodC > kTiny) && (plantWoodC + delta > kTiny) &&
                              (fineRootC + coarseRootC > kTiny);
      if (__builtin_expect(sufficient != alive0, 0)) {
        if (!alive0) {
          alive = true;  // it is back (planting)
        } else {
          alive = false;
          if (diedAt < 0) diedAt = t;
          if (Full) {
            deathWood = plantWoodC + delta;
            deathRoot = fineRootC + coarseRootC;
            diedNowRec = 1.0;
          }
          soilC += fineRootC + coarseRootC;
          if (Generic && F.litterPool) litterC += plantWoodC + plantLeafC + delta;
          else soilC += plantWoodC + plantLeafC + delta;
          if (Generic && F.nitrogen) {  // sipnet.c:1735-1746
            soilOrgN += fineRootC * (double)G_iFineCN + coarseRootC * (double)G_iWoodCN;
            litterN += plantWoodC * (double)G_iWoodCN + plantLeafC * (double)G_iLeafCN + storN;
            storN = 0.0;
          }
          plantWoodC = 0.0;
          plantLeafC = 0.0;
          coarseRootC = 0.0;
          fineRootC = 0.0;
          delta = 0.0;
          ringSum = 0.0;
        }
      }
    }
    // ensureNonNegativeStocks(), sipnet.c:1368-1397
    if (wantDiag) {  // with the reference's warning count (|v| > EPS), sipnet.c:1346-1356
      auto clampW = [&](double& v, double minVal) {
        if (v < minVal) {
          if (fabs(v) > kEps) clampWarn++;
          v = 0.0;
        }
      };
      clampW(plantWoodC, 0.0);
      clampW(plantLeafC, 0.0);
      if (Generic && F.litterPool) clampW(litterC, 0.0);
      clampW(soilC, 0.0);
      clampW(coarseRootC, 0.0);
      clampW(fineRootC, 0.0);
      clampW(soilWater, 0.0);
      clampW(snow, kTiny);
      if (Generic) {
        clampW(minN, 0.0);
        clampW(soilOrgN, 0.0);
        clampW(litterN, 0.0);
        clampW(storN, 0.0);
        if (!F.litterPool) litterC = rmax0(litterC);
      }
    } else {
      plantWoodC = rmax0(plantWoodC);
      plantLeafC = rmax0(plantLeafC);
      soilC = rmax0(soilC);
      coarseRootC = rmax0(coarseRootC);
      fineRootC = rmax0(fineRootC);
      soilWater = rmax0(soilWater);
      snow = snow < kTiny ? 0.0 : snow;
      if (Generic) {
        litterC = rmax0(litterC);
        minN = rmax0(minN);
        soilOrgN = rmax0(soilOrgN);
        litterN = rmax0(litterN);
        storN = rmax0(storN);
      }
    }
    if (wantDiag) {  // updateBalanceTrackerPostClamp() + checkBalance(), balance.c:40-169
      const double finC = massC(), finN = massN();
      double clampedC = finC - postC, clampedN = finN - postN;
      if (clampedC < kEps) clampedC = 0.0;
      if (clampedN < kEps) clampedN = 0.0;
      double inC = (double)photosynthesis + (double)evInC;
      double outC = (double)rVeg + (double)rFineRoot + (double)rCoarseRoot + (double)rSoil +
                    (double)soilMethane + (double)evOutC;
      if (Generic && F.litterPool) outC += (double)rLitter + (double)litterMethane;
      inC *= (double)len;
      outC *= (double)len;
      double inN = 0.0, outN = 0.0;
      if (Generic && F.nitrogen) {
        inN = ((double)nFixation + (double)evInN) * (double)len;
        outN = ((double)nLeaching + (double)nVolatilization + (double)evOutN) * (double)len;
      }
      inC += clampedC;
      if (Generic && F.nitrogen) inN += clampedN;
      const double dC = (finC - preC) - (inC - outC);
      const double dN = (finN - preN) + (outN - inN);
      maxDC = fmax(maxDC, fabs(dC));
      maxDN = fmax(maxDN, fabs(dN));
      if (!(fabs(dC) < kEps)) balanceWarn++;
      if (!(fabs(dN) < kEps)) balanceWarn++;
    }
    STAMP(4)

    // ---- 4. outputs: updateTrackers(), sipnet.c:1420-1496 ---------------------------
    const R tGpp = photosynthesis * len;
    const R tRh = Generic ? (rLitter + rSoil) * len : rSoil * len;
    const R tRa = (rCoarseRoot + rFineRoot) * len + rVeg * len;
    const R tNee = R(-1.0) * ((tGpp - tRa) - tRh);
    const R tEt = (transpiration + immedEvap + evaporation + sublimation + evEvap) * len;
    totGpp += (double)tGpp;
    totNee += (double)tNee;
    if (!Full) pfNee += (double)tNee;   // the launch's own NEE sum, from zero, in step order (FastArgs::pfLogw)
    R tRAbove = 0, tRRoot = 0, tRSoil = 0, tRtot = 0, tNpp = 0;
    if (Full) {  // the rest of updateTrackers(), sipnet.c:1420-1496
      if (bits & FAST_TRACK_NEW_YEAR) yGpp = yRtot = yRa = yRh = yNpp = yNee = 0.0;
      tRAbove = rVeg * len;
      tRRoot = (rCoarseRoot + rFineRoot) * len;
      tRSoil = tRRoot + tRh;
      tRtot = tRa + tRh;
      tNpp = tGpp - tRa;
      yGpp += (double)tGpp;
      yRa += (double)tRa;
      yRh += (double)tRh;
      yRtot += (double)tRtot;
      yNpp += (double)tNpp;
      yNee += (double)tNee;
      totRa += (double)tRa;
      totRh += (double)tRh;
      totRtot += (double)tRtot;
      totNpp += (double)tNpp;
      yLitter += (double)(leafLitter + recEvLeafOffLitterAll);
    }

    // ---- 5. running mean of NPP (sipnet.c:1546-1570, runmean.c:61-116 via the plan) ----
    const double npp = (double)(photosynthesis - rVeg - rCoarseRoot - rFineRoot);
    const double recMeanNpp = Full ? ringSum / 5.0 : 0.0;  // trackers.meanNPP: the mean BEFORE this step's insert
    STAMP(5)
    {
      const double v0 = useLast0 ? lastNpp : (double)rv0;
      const int nOps = bits >> 16;
      // regular step: every lane alive with an untouched ring epoch, one or two evictions, plain
      // insert.  TWO evictions is the steady state of half-hourly forcing (240 x 1/48 is not
      // exactly 5 in floating point: every step evicts a 2.9e-15-day residue of the oldest entry
      // and all but that of the next one); with one eviction w1 is 0 and its term an exact no-op.
      // Same three fused multiply-adds as the general branch below performs for such a step.
      const bool irregular = __builtin_amdgcn_ballot_w64(!alive || ringValidFrom > 0) != 0 ||
                             insSlot < 0 || nOps > 2;
      if (__builtin_expect(!irregular, 1)) {
        const double v1 = useLast1 ? lastNpp : (double)rv1;
        ringSum = ffma(-q7.y, v0, ringSum);
        ringSum = ffma(-rare[0], v1, ringSum);
        ringSum = ffma(npp, (double)len, ringSum);
      } else if (alive) {
        if (insSlot < 0) {
          ringSum = npp * 5.0;
        } else {
          double w0v = v0, w1v = useLast1 ? lastNpp : (double)rv1;
          if (ringValidFrom > 0) {  // a member that died earlier: older slots count as zero
            if (uni(rareI[0]) < ringValidFrom) w0v = 0.0;
            if (uni(rareI[1]) < ringValidFrom) w1v = 0.0;
          }
          ringSum = ffma(-q7.y, w0v, ringSum);
          ringSum = ffma(-rare[0], w1v, ringSum);  // w1 = 0 when there is no second eviction
          for (int k = 2; k < nOps; k++) {
            const RingOp& op = a.ringOps[opBase + uni(rareI[2]) + k];
            const double v = (uni(op.insStep) >= ringValidFrom)
                                 ? (double)ringp[(uint32_t)uni(op.slot) * ncu] : 0.0;
            ringSum = ffma(-op.w, v, ringSum);
          }
          ringSum = ffma(npp, (double)len, ringSum);
        }
      } else {
        ringValidFrom = t + 1;
      }
    }
    STAMP(6)
    if (Full && recp) {  // the strict kernel's record row (include/sipnet_amd.h), sipnet.c:453-473
      double* __restrict__ r = recp;
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
      r[12 * L] = (oldSoilWater + soilWater) / K_whc2;
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
      r[25 * L] = storN;
      r[26 * L] = delta;
      r[27 * L] = (double)(nVolatilization * len);
      r[28 * L] = (double)(nLeaching * len);
      r[29 * L] = (double)(nFixation * len);
      r[30 * L] = (double)(nUptake * len);
      r[31 * L] = (double)((soilMethane + litterMethane) * len);
      r[32 * L] = recMeanNpp;
      r[33 * L] = rare[3];  // gddAfter
      r[34 * L] = rare[4];  // tillAfter
      r[35 * L] = totGpp;
      r[36 * L] = (double)(leafOnCreation * len);
      r[37 * L] = (double)(leafOnFromWood * len);
      r[38 * L] = (double)(recLeafOffComputed * len);
      r[39 * L] = (double)(recEvLeafOn * len);
      r[40 * L] = (double)(recEvLeafOnFromWood * len);
      r[41 * L] = deathWood;
      r[42 * L] = deathRoot;
      r[43 * L] = diedNowRec;
      recp += (int64_t)SIPNET_NREC * L;
    }
    // request the values the NEXT step evicts, then store: loads ahead of stores in the queue
    const int insEff = insSlot < 0 ? 0 : insSlot;
    const int pfSlot0 = (slots >> 16) & 255, pfSlot1 = (slots >> 24) & 255;
    rv0 = ringp[(uint32_t)pfSlot0 * ncu];
    rv1 = ringp[(uint32_t)pfSlot1 * ncu];
    useLast0 = (pfSlot0 == insEff);  // the slot being written right now (uniform test);
    useLast1 = (pfSlot1 == insEff);  // consumed a whole step later, no wait here
    lastNpp = npp;
    *oNee = tNee;
    *oGpp = tGpp;
    *oEt = tEt;
    oNee += ldNee;
    oGpp += ldGpp;
    oEt += ldEt;
    // a dead member's slot is never read as live data again (ringValidFrom), so the insert
    // needs no alive test
    ringp[(uint32_t)insEff * ncu] = (R)npp;   // (npp is an R-typed difference: nothing is lost)
    STAMP(7)
  }  // steps of this tile
  }  // tiles

#ifdef SIPNET_STAMPS
  if (blockIdx.x == 0 && lane == 0) {
    g_stamps[0] = stampAcc0; g_stamps[1] = stampAcc1; g_stamps[2] = stampAcc2; g_stamps[3] = stampAcc3;
    g_stamps[4] = stampAcc4; g_stamps[5] = stampAcc5; g_stamps[6] = stampAcc6; g_stamps[7] = stampAcc7;
  }
#endif
  // ---- state back to HBM ----------------------------------------------------------
  if (act) {
    ST(plantWoodC) = plantWoodC;
    ST(plantLeafC) = plantLeafC;
    ST(soilC) = soilC;
    ST(soilWater) = soilWater;
    ST(snow) = snow;
    ST(coarseRootC) = coarseRootC;
    ST(fineRootC) = fineRootC;
    ST(plantCAccountingDelta) = delta;
    if (Generic) {
      ST(litterC) = litterC;
      ST(minN) = minN;
      ST(soilOrgN) = soilOrgN;
      ST(litterN) = litterN;
      ST(plantStorageN) = storN;
    }
    ST(ringSum) = ringSum;
    ST(totNee) = totNee;
    ST(totGpp) = totGpp;
    ST(phenBits) = (double)phenBits;
    ST(ringValidFrom) = (double)ringValidFrom;
    ST(diedAt) = (double)diedAt;
    if (Full) {
      ST(totRtot) = totRtot;
      ST(totRa) = totRa;
      ST(totRh) = totRh;
      ST(totNpp) = totNpp;
      ST(yearlyGpp) = yGpp;
      ST(yearlyRtot) = yRtot;
      ST(yearlyRa) = yRa;
      ST(yearlyRh) = yRh;
      ST(yearlyNpp) = yNpp;
      ST(yearlyNee) = yNee;
      ST(yearlyLitter) = yLitter;
    }
    if (wantDiag) {
      double* __restrict__ dg = a.diag + col;
      dg[0 * nc] += (double)clampWarn;
      dg[1 * nc] += (double)balanceWarn;
      dg[2 * nc] = fmax(dg[2 * nc], maxDC);
      dg[3 * nc] = fmax(dg[3 * nc], maxDN);
    }
  }
  // a particle filter's forecast: the log-weights the analysis would compute in a pass over the plane (pf.hip logWeightOf:
  // the same operations in the same order -- no contraction here either) and the maximum of this wavefront's 64
  if (!Full && a.pfLogw) {
#pragma clang fp contract(off)
    const double z = (pfNee - a.pfObs) * a.pfInvSigma;
    const double lw = skip ? -INFINITY : -0.5 * z * z;
    if (live) a.pfLogw[col] = lw;
    double mx = live ? lw : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (threadIdx.x == 0) a.pfBlockMax[blockIdx.x] = mx;
  }
#undef ST
#undef PRM
#undef PRM_RARE
}

#ifdef SIPNET_STAMPS
extern "C" int sipnet_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long));
}
#endif

namespace {
template <class R, bool Plain, int Mode>
void launchFastOne(const FastArgs& a, int grid, bool twoWaves, hipStream_t stream, LaunchInfo* info) {
  // a second, 256-VGPR build exists where it costs at most a few spilled registers: the fp64
  // default-flag kernel (257 -> 256) and the fp32 run-time-flag kernel (260-266 -> 256); the other
  // fp32 kernels fit two waves as they are, the fp64 optional-flag kernels (378-400) do not.
  // The Full variants exist in the one-wave-per-SIMD register budget only.
  constexpr int kOcc2 = ((Mode == kFlagsDefault && sizeof(R) == 8) ||
                         (Mode == kFlagsRuntime && sizeof(R) == 4)) ? 2 : 1;
  const bool occ2 = kOcc2 == 2 && twoWaves && !a.full;
  if (a.full)
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, 1, true>), dim3(grid), dim3(64), 0, stream, a);
  else if (occ2)
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, kOcc2, false>), dim3(grid), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, 1, false>), dim3(grid), dim3(64), 0, stream, a);
  if (info) {
    snprintf(info->kernel, sizeof info->kernel, "stepFastKernel<%s, %s, %d, %d, %s>",
             sizeof(R) == 8 ? "double" : "float", Plain ? "true" : "false", Mode, occ2 ? 2 : 1,
             a.full ? "true" : "false");
    info->grid = grid;
    info->block = 64;
    // the fp32 default-flag / N-cycle builds need < 256 VGPRs: two waves fit as they are
    info->wavesPerSimd = (occ2 || (sizeof(R) == 4 && Mode != kFlagsRuntime && !a.full)) ? 2 : 1;
    info->ldsBytes = 2 * kFastTile * (int)sizeof(FastRec);
  }
}
template <int Mode>
void launchFastMode(const FastArgs& a, int precision, int grid, bool twoWaves, hipStream_t stream,
                    LaunchInfo* info) {
  if (precision == SIPNET_F64) {
    if (a.plainExp) launchFastOne<double, true, Mode>(a, grid, twoWaves, stream, info);
    else launchFastOne<double, false, Mode>(a, grid, twoWaves, stream, info);
  } else {
    if (a.plainExp) launchFastOne<float, true, Mode>(a, grid, twoWaves, stream, info);
    else launchFastOne<float, false, Mode>(a, grid, twoWaves, stream, info);
  }
}
}  // namespace

void launchStepFast(const FastArgs& a, int precision, int options, hipStream_t stream, LaunchInfo* info) {
  const int chunksPerSite = (a.n_members + 63) / 64;
  const int grid = a.n_sites * chunksPerSite;
  // more chunks than SIMDs: two resident wavefronts per SIMD pay (SIPNET_KOPT_ONE_WAVE_PER_SIMD
  // keeps the full register budget anyway)
  const bool twoWaves = grid > 4 * a.numCUs && !(options & SIPNET_KOPT_ONE_WAVE_PER_SIMD);
  // SIPNET_KOPT_RUNTIME_FLAGS: always the run-time-flag instantiation
  const bool forceRuntime = (options & SIPNET_KOPT_RUNTIME_FLAGS) != 0;
  if (isDefaultFlagSet(a.flags) && !forceRuntime) launchFastMode<kFlagsDefault>(a, precision, grid, twoWaves, stream, info);
  else if (isNCycleFlagSet(a.flags) && !forceRuntime) launchFastMode<kFlagsNCycle>(a, precision, grid, twoWaves, stream, info);
  else launchFastMode<kFlagsRuntime>(a, precision, grid, twoWaves, stream, info);
}

}  // namespace sipnet

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
#include <type_traits>

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
// (the sums kernel's accumulators: fast_body.inc; never touched -- and so never materialised -- in the plain kernel)
struct FastSums {
  double nee = 0.0, gpp = 0.0, et = 0.0;
  int left = 0;
};
#ifndef SIPNET_FAST_SUMS_TU
bool isNCycleFlagSet(const int32_t* f) {
  for (int i = 0; i < SIPNET_NFLAGS; i++) {
    if (i == SIPNET_F_SNOW || isPhenologyOrEventsFlag(i)) continue;  // (data, not code: see isDefaultFlagSet)
    const bool want = i == SIPNET_F_WATER_HRESP || i == SIPNET_F_LITTER_POOL || i == SIPNET_F_NITROGEN_CYCLE ||
                      i == SIPNET_F_ANAEROBIC;
    if ((f[i] != 0) != want) return false;
  }
  return true;
}
#endif

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
  constexpr bool Sums = false;
#include "fast_body.inc"
}
#ifdef SIPNET_FAST_SUMS_TU
template <class R, bool PlainExp, int Mode, int Occ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(Occ)))
void stepFastSumsKernel(FastArgs a) {
  constexpr bool Sums = true, Full = false;
#include "fast_body.inc"
}
#endif

#if defined(SIPNET_STAMPS) && !defined(SIPNET_FAST_SUMS_TU)
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
#ifdef SIPNET_FAST_SUMS_TU
  if (occ2)
    hipLaunchKernelGGL((stepFastSumsKernel<R, Plain, Mode, kOcc2>), dim3(grid), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((stepFastSumsKernel<R, Plain, Mode, 1>), dim3(grid), dim3(64), 0, stream, a);
#else
  if (a.full)
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, 1, true>), dim3(grid), dim3(64), 0, stream, a);
  else if (occ2)
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, kOcc2, false>), dim3(grid), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((stepFastKernel<R, Plain, Mode, 1, false>), dim3(grid), dim3(64), 0, stream, a);
#endif
  if (info) {
#ifdef SIPNET_FAST_SUMS_TU
    snprintf(info->kernel, sizeof info->kernel, "stepFastSumsKernel<%s, %s, %d, %d>",
             sizeof(R) == 8 ? "double" : "float", Plain ? "true" : "false", Mode, occ2 ? 2 : 1);
#else
    snprintf(info->kernel, sizeof info->kernel, "stepFastKernel<%s, %s, %d, %d, %s>",
             sizeof(R) == 8 ? "double" : "float", Plain ? "true" : "false", Mode, occ2 ? 2 : 1,
             a.full ? "true" : "false");
#endif
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

#ifdef SIPNET_FAST_SUMS_TU
namespace sums2 {
void launchStepFastSums(const FastArgs& a, int precision, int options, hipStream_t stream, LaunchInfo* info) {
#else
void launchStepFast(const FastArgs& a, int precision, int options, hipStream_t stream, LaunchInfo* info) {
#endif
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
#ifdef SIPNET_FAST_SUMS_TU
}  // namespace sums2
#endif

}  // namespace sipnet

// step_coop.hip -- the cooperative throughput kernel: three (or four) wavefronts per 64 members.
//
// Why: with <= 1 wavefront per SIMD (every BASELINE configuration up to 64 k members) the step
// loop is bound by what ONE wavefront can issue -- one instruction per ~4.3 cycles whatever its
// kind -- while three quarters of the chip idle.  A member-step is a
// chain of three blocks with thin interfaces:
//
//   L  light      lai(t)            -> potGrossPsn(t)          (dTemp, dVpd, 7-layer Simpson)
//   W  water      potGrossPsn(t)    -> photosynthesis(t), ET(t), GPP(t), soilWater(t+1), snow(t+1);
//                 also the soil-moisture effect on C's heterotrophic respiration (its own state)
//   C  carbon     photosynthesis(t), factors(t) -> pools(t+1), ring, NEE(t), lai(t+1)
//   F  factors    every factor of C's respiration terms that depends on climate and parameters only
//                 (Q10 terms, frozen-soil effect, tillage) -- a wave of its own when the workgroup
//                 has a CU to itself (its fourth SIMD is free), else part of L
//
// so a workgroup is three or four wavefronts on as many SIMDs of one CU, each running its OWN
// time loop over the same 64 members and the same site records, and passing a few doubles per
// member and step through LDS mailboxes guarded by sequence flags (DS operations of a wave execute
// in order: value then flag on the producer side, flag then value in one round trip on the
// consumer side).  The results are those of stepFastKernel to rounding (tests/test_gpu_batch.py),
// and every layout of this kernel gives the same bits (tests/test_gpu_configs.py).
//
// Layouts (coopBody<..., NP>): one chunk per workgroup, ring in LDS, with wave F (stepCoopKernel<..,
// true, ..>: batches of at most one chunk per CU) or ring in HBM without it; two chunks per
// eight-wave workgroup (stepCoopPairKernel: up to two chunks per CU); four chunks per twelve-wave
// workgroup (stepCoopQuadKernel: up to four).  See the comment above coopBody.
//
// lai(t+1) only depends on photosynthesis(t) through plant death (the leaf pool update has no
// photosynthesis term, sipnet.c:1579-1626), so C posts it BEFORE it waits for photosynthesis(t)
// and confirms it after the mortality test; W zeroes potGrossPsn(t+1) of a member that died in
// step t (its leaf pool was zeroed: lai = 0, hence potGrossPsn = 0, sipnet.c:590-641).  This takes
// the L -> W -> C chain off C's critical path.
// Mailbox slots are indexed by step & 1; the wait-for graph keeps every producer at most one
// step ahead of its consumer:  L(t) waits lai(t) [C past the pools of t-1];  W(t) ends only
// when C is past the pools of t-1 and, by day, waits potGrossPsn(t);  C(t) waits the factors
// W posts when it STARTS step t and, by day, psn(t).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

#include "fast_math.h"
#include "step_kernel.h"

namespace sipnet {
#ifdef SIPNET_COOP_BOUNDED
namespace bounded {   // (step_coop_bounded.hip: the same kernels with bounded waits, under names of their own)
#endif
#ifdef SIPNET_COOP_SUMS_TU
namespace sums2 {     // (step_coop_sums.hip: the in-launch sums of the layouts and precisions this file's own launcher does not carry)
#endif
namespace {

constexpr int kTileBytes = kFastTile * (int)sizeof(FastRec);

// ---- hand-over waits.  Every one of them is a spin on a sequence flag in LDS with no exit: the protocol guarantees the
// producer's progress (see the file header), and a back-off hook on the exit path of every poll costs the c10k step 11 %.
// A protocol bug therefore shows as a hung GPU.  The BOUNDED build of this file (step_coop_bounded.hip: the same code,
// -DSIPNET_COOP_BOUNDED, lean instantiations only; SIPNET_KOPT_BOUNDED_WAITS selects it -- fuzz campaigns and the
// fixed-seed slices, never the shape policy) gives every wait a budget of polls; a wait that exhausts it reports which one
// (the numbers below) and at which step, poisons its workgroup -- every later wait of the workgroup gives up at once, so the
// launch ends, with garbage -- and sipnet_batch_run answers SIPNET_ERR_INTERNAL naming the wait.
//    1  take
//    2  take
//    3  takePgp
//    4  takePgp
//    5  takeFactors
//    6  takeFactors
//    7  takeFactorsN
//    8  takeFactorsN
//    9  takeFactorsRing
//   10  takeFactorsRing
//   11  takeFactors7
//   12  takeFactors7
//   13  takeFactorsRing7
//   14  takeFactorsRing7
//   15  takeR2
//   16  takeR2
//   17  takeD1
//   18  takeD3
//   19  takeD4
//   20  takeD6
//   21  takeD8
//   22  awaitAtLeast
//   23  wave S: factors + moisture take
//   24  wave S: verdict word + GPP - R_a
//   25  wave C general step: record + factor take (Opt)
//   26  wave C general step: record + factor take (NCyc)
//   27  wave C general step: record + factor take
#ifdef SIPNET_COOP_BOUNDED
__device__ unsigned long long g_coopStuck[2];   // [0]: 1 << 63 | wait id << 32 | step of the FIRST wait that gave up; [1]: its workgroup
__shared__ int g_coopPoison;
constexpr int kSpinBudget = 1 << 22;            // polls (~0.2 s of a lone wave)
__device__ __forceinline__ bool spinOut(int& n, int id, int step) {
  n++;
  if ((n & 255) != 1) return false;             // look at the poison word on the 1st, 257th, ... failed poll
  int p;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"((unsigned)(size_t)&g_coopPoison) : "memory");
  if (__builtin_amdgcn_readfirstlane(p)) return true;
  if (n < kSpinBudget) return false;
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long rep = (1ull << 63) | ((unsigned long long)(unsigned)id << 32) | (unsigned)step;
    if (atomicCAS(&g_coopStuck[0], 0ull, rep) == 0ull) g_coopStuck[1] = blockIdx.x;
  }
  asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)&g_coopPoison), "v"(1) : "memory");
  return true;
}
#define WAIT_DO for (int spin_ = 0, once_ = 1; once_; once_ = 0) do
#define WAIT_WHILE(cond, id, step) while ((cond) && !spinOut(spin_, id, step))
#else
#define WAIT_DO do
#define WAIT_WHILE(cond, id, step) while (cond)
#endif

// Hand-over through LDS.  DS instructions of one wavefront are executed in issue order, so a
// producer needs no wait between the value and the flag (two plain ds_write), and a consumer
// that issues the flag read BEFORE the value read and finds the flag current has the published
// value: one LDS round trip per hand-over when the data is already there.
__device__ __forceinline__ void post(double* slot, int* flag, double v, int step) {
  asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3"
               :: "v"((unsigned)(size_t)slot), "v"(v), "v"((unsigned)(size_t)flag), "v"(step) : "memory");
}
__device__ __forceinline__ void post(float* slot, int* flag, float v, int step) {
  asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %2, %3"
               :: "v"((unsigned)(size_t)slot), "v"(v), "v"((unsigned)(size_t)flag), "v"(step) : "memory");
}
__device__ __forceinline__ double take(const double* slot, const int* flag, int step) {
  double v;
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v) : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)slot) : "memory");
  } WAIT_WHILE(uni(f) < step, 1, step);
  return v;
}
__device__ __forceinline__ float take(const float* slot, const int* flag, int step) {
  float v;
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v) : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)slot) : "memory");
  } WAIT_WHILE(uni(f) < step, 2, step);
  return v;
}
// Site-record reads and ring-value loads are issued from inline assembly on purpose: hipcc
// (ROCm 7.2) drains ALL vector-memory traffic -- `s_waitcnt vmcnt(0)`, i.e. the previous step's
// stores, ~800 cycles -- in front of every DS read that might alias a pending LDS-DMA and of
// every use of an ordinary global load while an LDS-DMA is in flight (cdna_hip_programming.md,
// glds note).  Here every wait of the hot path is placed by hand, and the hot loop of a wave that
// stores issues no vector-memory load at all (the ring lives in LDS).
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ldsAddr(const void* p) { return (unsigned)(size_t)p; }
// Q10 exponent (temperature / 10) x log2(Q10), rounded as a product of its own: the factor block runs
// on the light wave in some layouts and on a wave of its own (uniform temperature operand) in
// others, and the compiler must not fuse the product into exp2's range reduction in one of them only
template <class R>
__device__ __forceinline__ R q10Arg(R t10, R lg) {
  R x = t10 * lg;
  asm volatile("" : "+v"(x));
  return x;
}
// The "alive" confirmation of wave C is ONE per-lane word: magnitude = step + 2 (so that 0 / 1 are
// "nothing yet" whatever the first step of a launch is), negative when the member died in the
// step before (its posted leaf area is void).  Sequence and value travel in one DS operation.
__device__ __forceinline__ int aliveWord(int step, bool died) { return died ? -(step + 2) : (step + 2); }
__device__ __forceinline__ void postAlive(int* slot, int step, bool died) {
  asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)slot), "v"(aliveWord(step, died)) : "memory");
}
// wave W: potential photosynthesis of wave L (flag, then value) and wave C's alive word, one round trip
__device__ __forceinline__ void takePgp(const double* slot, const int* flag, const int* alive, int step,
                                        double& pgp, bool& died) {
  int f, w;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(pgp), "=&v"(w)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)slot), "v"((unsigned)(size_t)alive) : "memory");
  } WAIT_WHILE(uni(f) < step || uni(w < 0 ? -w : w) < step + 2, 3, step);
  died = w < 0;
}
__device__ __forceinline__ void takePgp(const float* slot, const int* flag, const int* alive, int step,
                                        float& pgp, bool& died) {
  int f, w;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(pgp), "=&v"(w)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)slot), "v"((unsigned)(size_t)alive) : "memory");
  } WAIT_WHILE(uni(f) < step || uni(w < 0 ? -w : w) < step + 2, 4, step);
  died = w < 0;
}
// wave C: this step's six factors -- rows 0..4 of the block from wave F (or L), row 5 (the moisture
// effect) from wave W -- and both producers' sequence flags: the two flags in one ds_read2_b32
// (they are neighbours), the six values in three ds_read2st64 (rows are 64 elements apart).
// Flags are read before the values (DS reads return in order): current flags vouch for them.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void takeFactors(const double* block, const int* flags2, int step, double& g1,
                                            double& g2, double& qSoilT, double& gFine, double& gCoarse,
                                            double& moist) {
  i2v f;
  d2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2st64_b64 %1, %5 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %5 offset0:2 offset1:3\n\tds_read2st64_b64 %3, %5 offset0:4 offset1:5\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 5, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
__device__ __forceinline__ void takeFactors(const float* block, const int* flags2, int step, float& g1,
                                            float& g2, float& qSoilT, float& gFine, float& gCoarse, float& moist) {
  i2v f;
  f2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2st64_b32 %1, %5 offset1:1\n\t"
                 "ds_read2st64_b32 %2, %5 offset0:2 offset1:3\n\tds_read2st64_b32 %3, %5 offset0:4 offset1:5\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 6, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
// NCyc, wave C: rows 0 1 | 3 4 | 6 of the factor block behind wave L's flag, and -- in the same round trip,
// if it is there already -- wave S's mineral nitrogen with its flag
__device__ __forceinline__ void takeFactorsN(const double* block, const int* facFlag, const double* minNSlot, const int* minNFlag,
                                             int step, double& g1, double& g2, double& gFine, double& gCoarse, double& qSoil,
                                             double& minN, int& minNSeq) {
  int f0, f1;
  d2v a, b;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %6\n\tds_read2st64_b64 %1, %7 offset1:1\n\tds_read2st64_b64 %2, %7 offset0:3 offset1:4\n\t"
                 "ds_read_b64 %3, %7 offset:3072\n\tds_read_b32 %4, %8\n\tds_read_b64 %5, %9\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f0), "=&v"(a), "=&v"(b), "=&v"(qSoil), "=&v"(f1), "=&v"(minN)
                 : "v"((unsigned)(size_t)facFlag), "v"((unsigned)(size_t)block), "v"((unsigned)(size_t)minNFlag),
                   "v"((unsigned)(size_t)minNSlot) : "memory");
  } WAIT_WHILE(uni(f0) < step, 7, step);   // (the mineral nitrogen is looked at, not waited for: see plantSideN)
  minNSeq = uni(f1);
  g1 = a.x; g2 = a.y; gFine = b.x; gCoarse = b.y;
}
__device__ __forceinline__ void takeFactorsN(const float* block, const int* facFlag, const double* minNSlot, const int* minNFlag,
                                             int step, float& g1, float& g2, float& gFine, float& gCoarse, float& qSoil,
                                             double& minN, int& minNSeq) {
  int f0, f1;
  f2v a, b;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %6\n\tds_read2st64_b32 %1, %7 offset1:1\n\tds_read2st64_b32 %2, %7 offset0:3 offset1:4\n\t"
                 "ds_read_b32 %3, %7 offset:1536\n\tds_read_b32 %4, %8\n\tds_read_b64 %5, %9\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f0), "=&v"(a), "=&v"(b), "=&v"(qSoil), "=&v"(f1), "=&v"(minN)
                 : "v"((unsigned)(size_t)facFlag), "v"((unsigned)(size_t)block), "v"((unsigned)(size_t)minNFlag),
                   "v"((unsigned)(size_t)minNSlot) : "memory");
  } WAIT_WHILE(uni(f0) < step, 8, step);   // (the mineral nitrogen is looked at, not waited for: see plantSideN)
  minNSeq = uni(f1);
  g1 = a.x; g2 = a.y; gFine = b.x; gCoarse = b.y;
}
// the same with the ring value the step will evict riding in the same round trip (LDS ring, regular
// tiles: the slot is known at the top of the step, the value is consumed at its end)
__device__ __forceinline__ void takeFactorsRing(const double* block, const int* flags2, unsigned ringAddr,
                                                int step, double& g1, double& g2, double& qSoilT,
                                                double& gFine, double& gCoarse, double& moist, double& ringV) {
  i2v f;
  d2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2st64_b64 %1, %6 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %6 offset0:2 offset1:3\n\tds_read2st64_b64 %3, %6 offset0:4 offset1:5\n\t"
                 "ds_read_b64 %4, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(ringV)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block), "v"(ringAddr) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 9, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
__device__ __forceinline__ void takeFactorsRing(const float* block, const int* flags2, unsigned ringAddr,
                                                int step, float& g1, float& g2, float& qSoilT, float& gFine,
                                                float& gCoarse, float& moist, double& ringV) {
  i2v f;
  f2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2st64_b32 %1, %6 offset1:1\n\t"
                 "ds_read2st64_b32 %2, %6 offset0:2 offset1:3\n\tds_read2st64_b32 %3, %6 offset0:4 offset1:5\n\t"
                 "ds_read_b64 %4, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(ringV)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block), "v"(ringAddr) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 10, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
// five values + flag.  DS writes of one wave execute in issue order, so the flag lands after the
// values.  Written as DS instructions by hand: the compiler's version of the flag store is a FLAT
// store followed by a full `s_waitcnt vmcnt(0)`, and its value stores wait for the LDS-DMA tile
// in flight.
__device__ __forceinline__ void post5(double* base, int* flag, double v0, double v1, double v2, double v3,
                                      double v4, int step) {
  asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:1\n\tds_write2st64_b64 %0, %3, %4 offset0:2 offset1:3\n\t"
               "ds_write_b64 %0, %5 offset:2048\n\tds_write_b32 %6, %7"
               :: "v"(ldsAddr(base)), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(ldsAddr(flag)), "v"(step)
               : "memory");
}
__device__ __forceinline__ void post5(float* base, int* flag, float v0, float v1, float v2, float v3,
                                      float v4, int step) {
  asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:1\n\tds_write2st64_b32 %0, %3, %4 offset0:2 offset1:3\n\t"
               "ds_write_b32 %0, %5 offset:1024\n\tds_write_b32 %6, %7"
               :: "v"(ldsAddr(base)), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(ldsAddr(flag)), "v"(step)
               : "memory");
}
// Ext (optional physics without the nitrogen cycle), wave W: rows 5 and 6 of the factor block -- the moisture effect
// on heterotrophic respiration and the methane moisture term -- behind its flag
__device__ __forceinline__ void post2(double* base, int* flag, double v0, double v1, int step) {
  asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:1\n\tds_write_b32 %3, %4"
               :: "v"(ldsAddr(base)), "v"(v0), "v"(v1), "v"(ldsAddr(flag)), "v"(step) : "memory");
}
__device__ __forceinline__ void post2(float* base, int* flag, float v0, float v1, int step) {
  asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:1\n\tds_write_b32 %3, %4"
               :: "v"(ldsAddr(base)), "v"(v0), "v"(v1), "v"(ldsAddr(flag)), "v"(step) : "memory");
}
// ... and wave C's take of all seven rows (takeFactors / takeFactorsRing with row 6)
__device__ __forceinline__ void takeFactors7(const double* block, const int* flags2, int step, double& g1, double& g2,
                                             double& qSoilT, double& gFine, double& gCoarse, double& moist, double& mK) {
  i2v f;
  d2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2st64_b64 %1, %6 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %6 offset0:2 offset1:3\n\tds_read2st64_b64 %3, %6 offset0:4 offset1:5\n\t"
                 "ds_read_b64 %4, %6 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(mK)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 11, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
__device__ __forceinline__ void takeFactors7(const float* block, const int* flags2, int step, float& g1, float& g2,
                                             float& qSoilT, float& gFine, float& gCoarse, float& moist, float& mK) {
  i2v f;
  f2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2st64_b32 %1, %6 offset1:1\n\t"
                 "ds_read2st64_b32 %2, %6 offset0:2 offset1:3\n\tds_read2st64_b32 %3, %6 offset0:4 offset1:5\n\t"
                 "ds_read_b32 %4, %6 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(mK)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 12, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
__device__ __forceinline__ void takeFactorsRing7(const double* block, const int* flags2, unsigned ringAddr, int step,
                                                 double& g1, double& g2, double& qSoilT, double& gFine, double& gCoarse,
                                                 double& moist, double& mK, double& ringV) {
  i2v f;
  d2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %6 offset1:1\n\tds_read2st64_b64 %1, %7 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %7 offset0:2 offset1:3\n\tds_read2st64_b64 %3, %7 offset0:4 offset1:5\n\t"
                 "ds_read_b64 %4, %7 offset:3072\n\tds_read_b64 %5, %8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(mK), "=&v"(ringV)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block), "v"(ringAddr) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 13, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
__device__ __forceinline__ void takeFactorsRing7(const float* block, const int* flags2, unsigned ringAddr, int step,
                                                 float& g1, float& g2, float& qSoilT, float& gFine, float& gCoarse,
                                                 float& moist, float& mK, double& ringV) {
  i2v f;
  f2v a, b, c;
  WAIT_DO {
    asm volatile("ds_read2_b32 %0, %6 offset1:1\n\tds_read2st64_b32 %1, %7 offset1:1\n\t"
                 "ds_read2st64_b32 %2, %7 offset0:2 offset1:3\n\tds_read2st64_b32 %3, %7 offset0:4 offset1:5\n\t"
                 "ds_read_b32 %4, %7 offset:1536\n\tds_read_b64 %5, %8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(mK), "=&v"(ringV)
                 : "v"((unsigned)(size_t)flags2), "v"((unsigned)(size_t)block), "v"(ringAddr) : "memory");
  } WAIT_WHILE(uni(f.x < f.y ? f.x : f.y) < step, 14, step);
  g1 = a.x; g2 = a.y; qSoilT = b.x; gFine = b.y; gCoarse = c.x; moist = c.y;
}
// Blocks of doubles (rows 64 apart) + a sequence flag, for the hand-overs of the nitrogen-cycle layout:
// values first, flag last (DS writes of a wave execute in order); a reader takes the flag first and
// everything in one round trip.  One asm statement per take: every value is defined by it.
__device__ __forceinline__ void postRaw(double* p, double v) {
  asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void postRaw(float* p, float v) {
  asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}
// two rows of an R-typed block behind one flag (wave W: the tillage-scaled and the plain soil Q10 factor)
__device__ __forceinline__ void takeR2(const double* a, const double* b, const int* flag, int step, double& va, double& vb) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(va), "=&v"(vb)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)a), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 15, step);
}
__device__ __forceinline__ void takeR2(const float* a, const float* b, const int* flag, int step, float& va, float& vb) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(va), "=&v"(vb)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)a), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 16, step);
}
__device__ __forceinline__ void postD(double* row0, int k, double v) {
  asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)(row0 + 64 * k)), "v"(v) : "memory");
}
__device__ __forceinline__ void postFlag(int* flag, int step) {
  asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)flag), "v"(step) : "memory");
}
__device__ __forceinline__ void takeD1(const double* b, const int* flag, int step, double& v0) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v0) : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 17, step);
}
__device__ __forceinline__ void takeD3(const double* b, const int* flag, int step, double& v0, double& v1, double& v2) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %5 offset:512\n\t"
                 "ds_read_b64 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v0), "=&v"(v1), "=&v"(v2)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 18, step);
}
__device__ __forceinline__ void takeD4(const double* b, const int* flag, int step, double& v0, double& v1, double& v2,
                                       double& v3) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %5\n\tds_read_b64 %1, %6\n\tds_read_b64 %2, %6 offset:512\n\t"
                 "ds_read_b64 %3, %6 offset:1024\n\tds_read_b64 %4, %6 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 19, step);
}
__device__ __forceinline__ void takeD6(const double* b, const int* flag, int step, double& v0, double& v1, double& v2,
                                       double& v3, double& v4, double& v5) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %7\n\tds_read_b64 %1, %8\n\tds_read_b64 %2, %8 offset:512\n\t"
                 "ds_read_b64 %3, %8 offset:1024\n\tds_read_b64 %4, %8 offset:1536\n\t"
                 "ds_read_b64 %5, %8 offset:2048\n\tds_read_b64 %6, %8 offset:2560\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 20, step);
}
__device__ __forceinline__ void takeD8(const double* b, const int* flag, int step, double& v0, double& v1, double& v2,
                                       double& v3, double& v4, double& v5, double& v6, double& v7) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %9\n\tds_read_b64 %1, %10\n\tds_read_b64 %2, %10 offset:512\n\t"
                 "ds_read_b64 %3, %10 offset:1024\n\tds_read_b64 %4, %10 offset:1536\n\t"
                 "ds_read_b64 %5, %10 offset:2048\n\tds_read_b64 %6, %10 offset:2560\n\t"
                 "ds_read_b64 %7, %10 offset:3072\n\tds_read_b64 %8, %10 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                 : "v"((unsigned)(size_t)flag), "v"((unsigned)(size_t)b) : "memory");
  } WAIT_WHILE(uni(f) < step, 21, step);
}
// The "mineral nitrogen is plentiful" test both wave C and wave W evaluate for a step (they must
// come to the SAME wave-uniform answer, so it is one function, compiled without contraction): a lower
// bound of what checkNitrogenLimitation() (limitations.c:69-114) calls availableMinN --
// minN + (nMin - nVolatilization - nLeaching) * len >= minN * (1 - (nVolFrac * qSoil + nLeachFrac) * len),
// since nMin >= 0, the volatilisation moisture term is at most 0.05 + 3.8 / 4 = 1 (nitrogen.c:15-26) and
// the leached share at most 1 (nitrogen.c:31-41) -- against the plants' whole demand, of which the
// uptake is a part.  Where it holds for all 64 members nobody is limited, and neither wave waits for
// the other's exact numbers.
__device__ __forceinline__ bool nPlentiful(double minN, double qSoil, double len, double nVolFrac, double nLeachFrac,
                                           double demand) {
#pragma clang fp contract(off)
  const double lossShare = (nVolFrac * qSoil + nLeachFrac) * len;
  const double lower = minN * (1.0 - lossShare);
  return !(demand * len > lower);
}
// progress-only wait (no value)
__device__ __forceinline__ void awaitAtLeast(const int* flag, int step) {
  int f;
  WAIT_DO {
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"((unsigned)(size_t)flag) : "memory");
  } WAIT_WHILE(uni(f) < step, 22, step);
}

// pool += x * len, the forward-Euler update (sipnet.c:1579-1680): one fma in fp64; in fp32-mixed the
// product is formed in fp32 and added to the fp64 pool
__device__ __forceinline__ void accum(double& pool, double x, double len) { pool = __builtin_fma(x, len, pool); }
__device__ __forceinline__ void accum(double& pool, float x, float len) { pool += (double)(x * len); }

#include "coop_probes.h"   // measurement scaffolding (-DSIPNET_PROBES builds only; empty macros otherwise)
}  // namespace

// RingLds: the running-mean ring of the 64 members stays in LDS for the whole launch (one
// workgroup per CU); otherwise it stays in HBM and up to four workgroups share a CU.
// Full: see stepFastKernel -- every accumulator of the restart schema, the optional 44-column record
// (the carbon wave writes the carbon / tracker columns, the water wave columns 1, 2, 12, 13, 17, 19, 35)
// and the optional per-member diagnostics (clamp and carbon-balance warnings; default flags have no
// nitrogen balance).  Same flux arithmetic and hand-overs as the lean variant.
// NP = 2 (stepCoopPairKernel): one workgroup of eight wavefronts carries TWO chunks (ring in HBM).
// Waves go to the CU's four SIMDs round-robin, so with the roles laid out as  C0 C1 W0 W1 | L1 L0 F0 F1
// a carbon wave shares its SIMD with the OTHER chunk's light wave (idle at night, when the carbon wave is
// the step; by day neither waits for the other) and a water wave with its chunk's factor wave, whatever
// SIMD the workgroup starts on.  (Round 2's layout C0 C1 W0 W1 -- -- L0 L1 kept the carbon waves alone and
// put a chunk's water and light waves -- factors included -- on one SIMD: 1 741 cycles per day step
// against 1 185 by night; this one 1 426 / 1 260, c4 10.5 -> 9.6 ms.)
// Two separate three-wave workgroups on a CU put the second one's water wave on the first one's
// carbon SIMD (tools/coop_placement.py), which costs the carbon wave a third of its issue rate.
// NP = 4 (stepCoopQuadKernel): twelve wavefronts carry FOUR chunks, C0..C3 W0..W3 L0..L3: every SIMD
// runs the three waves of one chunk, which fill each other's dependency and hand-over gaps -- for
// batches of up to four chunks per CU, where the one-wave kernel leaves every SIMD with a lone wave.
// NCyc (stepCoopNKernel): the nitrogen-cycle flag set (litter pool + anaerobic + nitrogen cycle on top
// of the defaults: what nitrogen-cycle requires, context.c:203-212) compiled in.  The soil side of the
// model becomes a wave of its own, S, on the fourth SIMD (it also does wave F's job, one step ahead):
// S keeps soil carbon, the litter pool and the four nitrogen pools and computes heterotrophic
// respiration, litter breakdown, methane and nitrogen.c's fluxes; C keeps the plants, W the water.
// Per step W hands S the anaerobic moisture terms and the leached share, C hands S the plants' litter
// fluxes and nitrogen demand and, at the end of its step, GPP - R_a (S has R_h: it forms, stores and totals
// NEE); S hands C the mineral nitrogen the limitation test needs; rare things (events, plant death, a nitrogen-limited step) travel in blocks of their own.
// (The first version had this block on wave W: 2 400 cycles per night step there against C's 1 800;
// c10kn 20.3 ms.)  One chunk per workgroup, ring in HBM (the new mailboxes take the LDS the ring
// would), lean state only.
// Ext: the reference's other optional physics under RUN-TIME flags (a.flags; wave-uniform, and written so that a
// flag that is off costs an exactly neutral operand -- rate 0, cap "infinite" -- rather than a branch):
//   growth respiration (vegResp2, sipnet.c:1084-1103)            wave C: rVeg += max(0, growthRespFrac * mean NPP)
//   leaf-water interception (calcPrecip, sipnet.c:848-882)       wave W: immediate evaporation capped by lai x
//                                                                leafPoolDepth -- it takes lai(t) (and C's alive
//                                                                word) on the steps with rain only
//   flooding (calcSoilWaterFluxes, sipnet.c:1019-1027)           wave W: drainage capped by waterDrainFrac
//   carbon saturation (updatePoolsForSoil, sipnet.c:1645-1668)   whoever owns soil carbon (C; NCyc: S)
// and, without the nitrogen cycle (NCyc = false: "the optional-physics layouts", russell_3's flag family):
//   litter pool (calcLitterFluxes, sipnet.c:1150-1171)           wave C keeps litterC next to soilC: breakdown =
//                                                                litterC x (litterBreakdownRate / baseSoilResp) x
//                                                                fSoil, R_h = rSoil + rLitter
//   anaerobic moisture effect + methane (depeffects.c:46-96,     wave W posts the anaerobic form of the moisture
//   sipnet.c:1201-1214)                                          effect and, as row 6 of the factor block, the methane
//                                                                moisture term / (baseSoilResp x (1 + tillage)), so
//                                                                that C's methane = rate x pool x qSoilT x row 6
// Same waves, same hand-overs as the default layouts (one more factor row); lean state only.
// the code phase of an instantiation: s_nop count after the 32-byte boundary in coopBody's prologue (see there)
// (role: 0 carbon, 1 water, 2 light, 3 factor / soil wave -- each wave's code starts at a boundary of its own --
// 4 the common prologue)
template <class R, bool PlainExp, bool RingLds, bool Full, int NP, bool NCyc, bool Ext, bool Sums = false>
__device__ constexpr int coopCodePhase(int role) {
#ifdef SIPNET_PH_C
  if (role == 0) return SIPNET_PH_C;
#endif
#ifdef SIPNET_PH_W
  if (role == 1) return SIPNET_PH_W;
#endif
#ifdef SIPNET_PH_L
  if (role == 2) return SIPNET_PH_L;
#endif
#ifdef SIPNET_PH_F
  if (role == 3) return SIPNET_PH_F;
#endif
  // (the in-kernel sums build of the one-chunk LDS-ring layout, swept in round 6 at c10k's shape, tools/sums_time.py,
  // profiles/r06_sums_phase_sweep.txt: carbon 8.65 ... 8.46 ms at phase 3, then water 8.52 ... 8.40 at phase 3; its
  // relatives take their plain family's values -- the fp32-mixed build of the same layout too: 8.57 ms with its family's phases
  // against 8.90 with these, 16 384 members)
  if (Sums && sizeof(R) == 8 && NP == 1 && RingLds && !NCyc && !Ext && (role == 0 || role == 1)) return 3;
  if (role == 0) {   // the carbon wave (profiles/r04_phase_sweep_roles.txt)
    if (NCyc) return 0;
    if (Ext) return 7;
    if (NP == 2) return 2;
    if (NP == 1 && RingLds) return Full ? 4 : 6;
    return 0;
  }
  if (role == 1) {   // the water wave
    if (NCyc) return 0;
    if (Ext) return 1;                            // X (LDS ring) f64 9.65 -> 9.49
    if (NP == 2) return 4;
    if (NP == 1 && RingLds) return Full ? 5 : 6;
    return 0;
  }
  if (role == 2) return (NP == 1 && RingLds && !Full && !NCyc && !Ext) ? 3 : 0;   // the light wave has slack: +-0.3 %
  if (role == 3) return (NCyc && NP == 1) ? 4 : 0;   // the soil wave: N f64 12.78 -> 12.69 (13.0 at its worst phases)
  if (role != 4) return 0;
#ifdef SIPNET_PAD_NOPS
  return SIPNET_PAD_NOPS;
#else
  // measured on the instantiations the workloads launch (tools/gpu_phase_sweep.sh, profiles/r04_phase_sweep.txt:
  // best against worst phase 1.5-5 %); relatives that were not measured take their family's value
  constexpr bool f64 = sizeof(R) == 8;
  if (NCyc) return 4;                          // N f64 12.98 -> 12.64 ms, NPair f64 15.19 -> 15.10
  if (Ext) return 4;                           // X (LDS ring) f64 9.68 -> 9.54
  if (NP == 4) return f64 ? 6 : 3;             // quad f32 9.26 -> 9.23 (9.36 at the worst phase), quad f64 18.44 -> 17.44
  if (NP == 2) return Full ? 0 : f64 ? 4 : 7;  // pair f64 9.76 -> 9.72 (9.91 at the worst), pair f32 8.90 -> 8.85 (9.17)
  if (!RingLds) return 0;                      // HBM ring f64 9.34 (9.59 at the worst)
  return (Full || f64) ? 3 : 4;                // LDS ring f64 8.51 -> 8.32, its full-state build 14.71 -> 14.21, f32 8.15 -> 8.08
#endif
}

// ---- coopBody by role (round 6) ------------------------------------------------------------------------------------------
// coopBody is ONE function -- a wavefront takes one of its branches by its index in the workgroup -- but it is written in
// one file per role, #included where the code stands:
//   coop_mailboxes.inc     the workgroup's LDS: record tiles per wave, two-slot mailboxes + sequence flags, the ring; this wave's
//                          views of them (per chunk on the two- / four-chunk layouts)
//   coop_stats.inc         sipnet_batch_run_stats inside the launch: wave F (or L) sums the plane tiles C and W have stored
//   coop_wave_factor.inc   F  climate-only factors of the respiration terms        (layouts with a spare wavefront)
//   coop_wave_soil.inc     S  soil carbon, litter, the nitrogen cycle, NEE         (nitrogen-cycle layouts)
//   coop_wave_light.inc    L  potential photosynthesis: the canopy's seven layers  (+ the factors where there is no F)
//   coop_wave_water.inc    W  moisture, GPP, transpiration, snow, soil water, ET
//   coop_wave_carbon.inc   C  carbon fluxes and pools, phenology, mortality, the running mean, NEE
// What stays here: the hand-over primitives (post / take / await, the bounded-wait build), the code-phase table, coopBody's
// prologue (who am I, which chunk, which members; the flags' initialisation and the one workgroup barrier) and the dispatch,
// the kernels and the launcher.  A function per role with the mailboxes passed as a struct was the alternative; the split by
// FILE keeps the token stream -- and with it the instruction stream and the measured code placement, which is worth up to
// 2.7 % of the headline -- exactly what it was: step_coop.o is byte-identical before and after the split.
// (the Sums instantiations' accumulators; an empty type otherwise, so that the other instantiations' code is what it was)
template <bool On>
struct CoopSums {
  double nee = 0.0, et = 0.0, gpp = 0.0;
  int left = 0;
};
template <>
struct CoopSums<false> {};
// Sums (round 6; lean, every physics family, layout and arithmetic -- this file's launcher carries fp64 on one and two chunks per
// workgroup, step_coop_sums.hip the rest; the sums are doubles whatever R is): the three output planes receive every member's SUMS over groups of
// a.sumEvery consecutive steps of the launch instead of the steps themselves -- [groups][ld] each, a row per group, the last
// group as long as the launch leaves it -- accumulated in step order by the wave that computes the value (one add per value and
// step, a store per group: 1 / sumEvery of the planes' HBM writes; sipnet_batch_run_sums).  Sums = false compiles to the code
// it was before the parameter existed (tests/test_code_placement.py holds the measured instantiations' loop heads in place).
// PairDiag (round 6): the two-chunk nitrogen-cycle layouts' full-state builds WITH the diagnostics counters (one slot of the
// mass-total rows per chunk: coop_mailboxes.inc) -- instantiations of their own, because the counters' code costs the plain
// full-state build its last registers (230 -> 256 VGPRs + 20 B of scratch: c4's shape with the record 17.9 -> 18.8 ms)
template <class R, bool PlainExp, bool RingLds, bool Full, int NP, bool NCyc = false, bool Ext = false, bool Sums = false, bool PairDiag = false>
__device__ __forceinline__ void coopBody(const FastArgs& a) {
  static_assert(!PairDiag || (NCyc && Full && NP == 2), "PairDiag: the nitrogen cycle's two-chunk full-state layout");
  static_assert(!Sums || !Full, "in-kernel sums: lean launches");
  using OutT = typename std::conditional<Sums, double, R>::type;   // what the output rows hold: the steps' values, or sums of them (always doubles)
  static_assert(!(NP > 1 && RingLds), "two chunks' rings do not fit one CU's LDS");
  static_assert(!NCyc || (NP <= 2 && !RingLds), "nitrogen-cycle layout: one or two chunks, ring in HBM");
  constexpr bool Opt = Ext && !NCyc;   // the optional pools live on wave C
  // the run-time flags (all false without Ext: dead code then)
  const bool F_growthResp = Ext && a.flags[SIPNET_F_GROWTH_RESP] != 0, F_leafWater = Ext && a.flags[SIPNET_F_LEAF_WATER] != 0;
  const bool F_flooding = Ext && a.flags[SIPNET_F_FLOODING] != 0, F_carbonSat = Ext && a.flags[SIPNET_F_CARBON_SATURATION] != 0;
  const bool F_litterPool = Opt && a.flags[SIPNET_F_LITTER_POOL] != 0, F_anaerobic = Opt && a.flags[SIPNET_F_ANAEROBIC] != 0;
  constexpr double kNoCap = 3.0e38;  // finite in fp32 too
  constexpr bool Pair = NP == 2;
  // a fourth wavefront computes the climate-only factors when the workgroup has a CU to itself
  // two chunks per workgroup: the two spare wavefronts of the eight are the chunks' factor waves, and the
  // layout is  C0 C1 W0 W1 | L1 L0 F0 F1 : a carbon wave shares its SIMD with the OTHER chunk's light wave
  // (idle at night, when the carbon wave is the step), a water wave with its chunk's factor wave
  // (NCyc: the fourth role is the soil wave S, the factors stay with L; its two-chunk layout is
  // C0 C1 S0 S1 | W1 W0 L1 L0 : the two busiest waves of a chunk, C and S, each share a SIMD with a water or light
  // wave of the OTHER chunk)
  constexpr bool FacWave = RingLds || (Pair && !NCyc);
#include "coop_mailboxes.inc"
#ifdef SIPNET_NO_STATS
  const bool stageOn = false;
#else
  const bool stageOn = Staged && a.statsPart != nullptr;
#endif
  [[maybe_unused]] const bool firstChunk = blockIdx.x == 0 && sub == 0;  // diagnostics builds report this one
  PROBE_HWID(blockIdx.x * NP + sub, role)
  unsigned char* lds = ldsTilesAll[sub][(role < 0 || role > (NCyc ? 3 : 2)) ? 0 : role];

  const int chunksPerSite = (a.n_members + 63) >> 6;
  int site, chunk;
  {
    const int pb = (int)blockIdx.x;
    if ((a.n_sites & 7) == 0) {  // keep a site's chunks on one XCD group (speed only)
      const int g = pb & 7, j = (pb >> 3) * NP + sub;
      site = g + 8 * (j / chunksPerSite);
      chunk = j % chunksPerSite;
    } else {
      const int b = pb * NP + sub;
      site = b / chunksPerSite;
      chunk = b % chunksPerSite;
    }
  }
  // an odd number of chunks leaves the last workgroup's second half empty
  bool present = uni((int)(site < a.n_sites)) != 0 && role >= 0;
  if (!present) site = 0, chunk = 0;
  int m = (chunk << 6) + lane;
  const bool live = m < a.n_members;
  if (!live) m = a.n_members - 1;  // clamped lanes recompute the last member, never store state
  const int64_t col = (int64_t)site * a.n_members + m;
  // ring-eviction and event indices in the site's records are local to the site
  const int opBase = uni(a.siteBase[3 * site]), evBase = uni(a.siteBase[3 * site + 1]);
  const int64_t nc = a.ncol;
  double* __restrict__ stp = a.state + col;
  const bool skip = stp[(int64_t)ST_status * nc] != 0.0;
  const bool act = live && !skip;
  const double* __restrict__ pp = a.prm + col;
#define PRM(name) (pp[(int64_t)SP_##name * nc])
#define PRM_RARE(name) ((R)pp[(int64_t)SP_##name * nc])
#define ST(name) stp[(int64_t)ST_##name * nc]

  // (sites of a batch may differ in length: a chunk runs to the end of ITS site's records, and not at all when they
  // ended before the launch's range)
  const int siteSteps = uni(a.siteBase[3 * site + 2]);
  const int tBegin = a.step0, tEnd = a.step0 + a.n_steps < siteSteps ? a.step0 + a.n_steps : siteSteps;
  present = present && tBegin < tEnd;
  // Where the loops lie in the instruction cache's 32-byte fetch windows is worth +-1.5 % of the step (a lone wave
  // pays every taken branch with a fetch; NOTES.md "Round 4: code placement"): everything from here on starts at a
  // 32-byte boundary plus a per-instantiation number of s_nop (4 bytes each), measured -- not at wherever the
  // prologue happens to end.  -DSIPNET_PAD_NOPS=k overrides it for all instantiations (tools/build_variants.py).
// (`s_nop 8 + role` in front: a marker the compiler never emits, executed once per wave and launch, by which
// tests/test_code_placement.py finds each pinned point in the DISASSEMBLY of the built library and checks that the code
// behind it starts where the sweeps measured it -- a compiler bump that moves a loop head fails a CPU test instead of
// silently costing up to 2.7 %)
#define COOP_CODE_PHASE(ROLE) \
  asm volatile("s_nop %1\n .p2align 5\n .rept %0\n s_nop 0\n .endr" ::"n"(coopCodePhase<R, PlainExp, RingLds, Full, NP, NCyc, Ext, Sums>(ROLE)), "n"(8 + (ROLE)))
  COOP_CODE_PHASE(4);
  if (role == 0) {
    if (lane == 0) {
      seqLai = tBegin - 1;
      seqPgp = tBegin - 1;
      seqPsn = tBegin - 1;
      seqFac = tBegin - 1;
      seqMoist = tBegin - 1;
      seqDone[0] = 0;
      seqDone[1] = 0;
      if (NCyc) {
        seqPlant = tBegin - 1;
        seqMinN = tBegin - 1;
        seqStorN = tBegin - 1;
        seqEvent = tBegin - 1;
        seqSupply = tBegin - 1;
        seqDemand = tBegin - 1;
        seqWat = tBegin - 1;
        seqLeach = tBegin - 1;
      }
    }
    mailAlive[0][lane] = 0;
    mailAlive[1][lane] = 0;
  }
#ifdef SIPNET_COOP_BOUNDED
  if (threadIdx.x == 0) g_coopPoison = 0;
#endif
  __syncthreads();  // the only workgroup barrier: flags initialised before anyone spins
  if (!present) return;
  // (wave priorities by role -- s_setprio for the carbon wave, or carbon > water > light, or the light wave first --
  // were measured on the layouts whose waves share a SIMD and are worth nothing: profiles/r05_wave_priority_probe.txt)

  const unsigned char* __restrict__ planBytes =
      (const unsigned char*)(a.fast + (int64_t)site * a.n_steps_total);
  const Exp2Coef EC = loadExp2Coef();

#include "coop_stats.inc"
#include "coop_wave_factor.inc"
  auto tileFirst = [&](int tile) -> int64_t {
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
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  int curTile = tBegin / kFastTile;
  stageTile(curTile, curTile & 1);
  __builtin_amdgcn_s_waitcnt(0);

#include "coop_wave_soil.inc"

#include "coop_wave_light.inc"

#include "coop_wave_water.inc"

#include "coop_wave_carbon.inc"
#undef ST
#undef PRM
#undef PRM_RARE
#undef seqFac
#undef seqMoist
}

template <class R, bool PlainExp, bool RingLds, bool Full>
__global__ __launch_bounds__(RingLds ? 256 : 192) void stepCoopKernel(FastArgs a) {
  coopBody<R, PlainExp, RingLds, Full, 1>(a);
}

template <class R, bool PlainExp, bool Full>
__global__ __launch_bounds__(512) void stepCoopPairKernel(FastArgs a) {
  coopBody<R, PlainExp, false, Full, 2>(a);
}

template <class R, bool PlainExp>
__global__ __launch_bounds__(768) void stepCoopQuadKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 4>(a);
}
#ifndef SIPNET_COOP_BOUNDED
// every member's sums over groups of a.sumEvery steps instead of the steps (coopBody, Sums): fp64, default physics, lean;
// one chunk per workgroup (ring in LDS or HBM) or two
template <bool PlainExp, bool RingLds>
__global__ __launch_bounds__(RingLds ? 256 : 192) void stepCoopSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, RingLds, false, 1, false, false, true>(a);
}
template <bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopPairSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, false, false, 2, false, false, true>(a);
}
// ... of the optional-physics instantiations (run-time flags) and of the nitrogen-cycle layouts (NEE summed by the soil wave)
template <bool PlainExp, bool RingLds>
__global__ __launch_bounds__(RingLds ? 256 : 192) void stepCoopXSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, RingLds, false, 1, false, true, true>(a);
}
template <bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopXPairSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, false, false, 2, false, true, true>(a);
}
template <bool PlainExp, bool Ext>
__global__ __launch_bounds__(256) void stepCoopNSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, false, false, 1, true, Ext, true>(a);
}
template <bool PlainExp, bool Ext>
__global__ __launch_bounds__(512) void stepCoopNPairSumsKernel(FastArgs a) {
  coopBody<double, PlainExp, false, false, 2, true, Ext, true>(a);
}
#endif

// (a full-state build of the four-chunk layout was probed in round 4: under its 168-register budget -- twelve
// wavefronts per CU -- the carbon wave's record columns and accumulators spill, fp64 284 bytes of scratch per lane, fp32-mixed
// 136; such batches take the one-wave kernel's Full build instead)

// the nitrogen-cycle flag set: four wavefronts per chunk (L W C S), soil + nitrogen on wave S
template <class R, bool PlainExp>
__global__ __launch_bounds__(256) void stepCoopNKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 1, true>(a);
}
// ... and two chunks per eight-wave workgroup, for batches of up to two chunks per CU
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNPairKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 2, true>(a);
}

// ... and their full-state builds (record, every accumulator; no diagnostics counters: the pools of a member are
// spread over two wavefronts)
template <class R, bool PlainExp>
__global__ __launch_bounds__(256) void stepCoopNFullKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 1, true>(a);
}
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNPairFullKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 2, true>(a);
}
#ifndef SIPNET_COOP_BOUNDED
// ... with the diagnostics counters (coopBody, PairDiag)
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNPairDiagKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 2, true, false, false, true>(a);
}
#endif

// ---- Ext: the optional-physics instantiations (run-time flags; see coopBody) -----------------------------------
// default pools + growth respiration / leaf water / flooding / litter pool / carbon saturation / anaerobic + methane
template <class R, bool PlainExp, bool RingLds, bool Full>
__global__ __launch_bounds__(RingLds ? 256 : 192) void stepCoopXKernel(FastArgs a) {
  coopBody<R, PlainExp, RingLds, Full, 1, false, true>(a);
}
template <class R, bool PlainExp, bool Full>
__global__ __launch_bounds__(512) void stepCoopXPairKernel(FastArgs a) {
  coopBody<R, PlainExp, false, Full, 2, false, true>(a);
}
// the nitrogen-cycle flag set + growth respiration / leaf water / flooding / carbon saturation
// four chunks per twelve-wave workgroup (round 5): fp32-mixed only -- the fp64 build needs 198 registers per lane, the
// layout's three wavefronts per SIMD leave 168 (tools/kernel_resources.py: it would spill in the carbon wave's loop),
// so fp64 batches of optional physics beyond two chunks per CU stay on the one-wave kernel
template <class R, bool PlainExp>
__global__ __launch_bounds__(768) void stepCoopXQuadKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 4, false, true>(a);
}
#if defined(SIPNET_PROBES) && defined(SIPNET_PROBE_XQUAD_F64)   // (tools/kernel_resources.py step_coop.hip -DSIPNET_PROBE_XQUAD_F64)
template __global__ void stepCoopXQuadKernel<double, true>(FastArgs);
#endif
template <class R, bool PlainExp>
__global__ __launch_bounds__(256) void stepCoopNXKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 1, true, true>(a);
}
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNXPairKernel(FastArgs a) {
  coopBody<R, PlainExp, false, false, 2, true, true>(a);
}
// ... with the record and every accumulator ("everything" + the 44-column record: round 5)
template <class R, bool PlainExp>
__global__ __launch_bounds__(256) void stepCoopNXFullKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 1, true, true>(a);
}
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNXPairFullKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 2, true, true>(a);
}
#ifndef SIPNET_COOP_BOUNDED
template <class R, bool PlainExp>
__global__ __launch_bounds__(512) void stepCoopNXPairDiagKernel(FastArgs a) {
  coopBody<R, PlainExp, false, true, 2, true, true, false, true>(a);
}
#endif

#ifdef SIPNET_COOP_SUMS_TU
// every member's sums over groups of a.sumEvery steps (coopBody, Sums) on ANY layout and precision: the output rows are doubles
// whatever the arithmetic type (an fp32-mixed batch adds its float values up in double, in step order)
template <class R, bool PlainExp, int Layout, bool Ext>
__global__ __launch_bounds__(Layout == COOP_RING_LDS ? 256 : Layout == COOP_RING_HBM ? 192 : Layout == COOP_QUAD ? 768
                             : Layout == COOP_NCYCLE ? 256 : 512) void stepCoopSumsAtKernel(FastArgs a) {
  constexpr bool N = Layout == COOP_NCYCLE || Layout == COOP_NCYCLE_PAIR;
  constexpr int NP = (Layout == COOP_PAIR || Layout == COOP_NCYCLE_PAIR) ? 2 : Layout == COOP_QUAD ? 4 : 1;
  coopBody<R, PlainExp, Layout == COOP_RING_LDS, false, NP, N, Ext, true>(a);
}
#endif

#if defined(SIPNET_HWID) && !defined(SIPNET_COOP_SUMS_TU)
extern "C" int sipnet_debug_read_coop_hwid(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopHwId), sizeof(unsigned) * 4096 * 4 * 2);
}
#endif
#if defined(SIPNET_WAITS) && !defined(SIPNET_COOP_SUMS_TU)
extern "C" int sipnet_debug_read_coop_waits(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopWaits), 16 * sizeof(unsigned long long));
}
#endif
#if defined(SIPNET_STAMPS) && !defined(SIPNET_COOP_SUMS_TU)
extern "C" int sipnet_debug_read_coop_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopStamps), 8 * sizeof(unsigned long long));
}
#endif

#ifdef SIPNET_COOP_BOUNDED
// what the first wait that gave up reported (0 0: none), after synchronising the stream
int readCoopStuck(unsigned long long out[2], hipStream_t stream) {
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopStuck), 2 * sizeof(unsigned long long));
}
#endif
#ifdef SIPNET_COOP_SUMS_TU
// the sums launches launchStepCoop does not carry itself: fp32-mixed batches on every layout, fp64 on the four-chunk layout
void launchStepCoopSums(const FastArgs& a, int precision, int layout, hipStream_t stream, LaunchInfo* info) {
  const int chunksPerSite = (a.n_members + 63) / 64, chunks = a.n_sites * chunksPerSite;
  const bool nFamily = layout == COOP_NCYCLE || layout == COOP_NCYCLE_PAIR;
  const bool ext = nFamily ? !isNCycleFlagSet(a.flags) : !isDefaultFlagSet(a.flags);
  const bool x8 = (a.n_sites & 7) == 0;   // (the XCD-grouped mapping: every group of eight workgroups carries 16 / 32 chunks)
  const int per = (layout == COOP_PAIR || layout == COOP_NCYCLE_PAIR) ? 2 : layout == COOP_QUAD ? 4 : 1;
  const int groups = per == 1 ? chunks : x8 ? 8 * ((chunks / 8 + per - 1) / per) : (chunks + per - 1) / per;
  const int threads = layout == COOP_RING_LDS ? 256 : layout == COOP_RING_HBM ? 192 : layout == COOP_QUAD ? 768 : layout == COOP_NCYCLE ? 256 : 512;
  const dim3 grid(groups), block(threads);
#define SUMS_AT(R, L)                                                                                         \
  {                                                                                                           \
    if (a.plainExp) { if (ext) hipLaunchKernelGGL((stepCoopSumsAtKernel<R, true, L, true>), grid, block, 0, stream, a);     \
                      else hipLaunchKernelGGL((stepCoopSumsAtKernel<R, true, L, false>), grid, block, 0, stream, a); }      \
    else { if (ext) hipLaunchKernelGGL((stepCoopSumsAtKernel<R, false, L, true>), grid, block, 0, stream, a);               \
           else hipLaunchKernelGGL((stepCoopSumsAtKernel<R, false, L, false>), grid, block, 0, stream, a); }                \
  }
  if (precision == SIPNET_F64) {   // (four chunks, default physics: the engine sends nothing else here)
    if (a.plainExp) hipLaunchKernelGGL((stepCoopSumsAtKernel<double, true, COOP_QUAD, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((stepCoopSumsAtKernel<double, false, COOP_QUAD, false>), grid, block, 0, stream, a);
  } else {
    switch (layout) {
      case COOP_RING_LDS: SUMS_AT(float, COOP_RING_LDS) break;
      case COOP_RING_HBM: SUMS_AT(float, COOP_RING_HBM) break;
      case COOP_PAIR: SUMS_AT(float, COOP_PAIR) break;
      case COOP_QUAD: SUMS_AT(float, COOP_QUAD) break;
      case COOP_NCYCLE: SUMS_AT(float, COOP_NCYCLE) break;
      default: SUMS_AT(float, COOP_NCYCLE_PAIR) break;
    }
  }
#undef SUMS_AT
  if (info) {
    snprintf(info->kernel, sizeof info->kernel, "stepCoopSumsAtKernel<%s, %s, %d, %s>", precision == SIPNET_F64 ? "double" : "float",
             a.plainExp ? "true" : "false", layout, ext ? "true" : "false");
    info->grid = (int32_t)grid.x;
    info->block = threads;
    info->wavesPerSimd = per == 4 ? 3 : per;
    const int elem = precision == SIPNET_F64 ? 8 : 4;
    info->ldsBytes = nFamily ? per * (3 * 2 * kTileBytes + (2 * 64 * 3 + 2 * 7 * 64) * elem + 2 * 64 * 4 + 16 * 4 + 64 * 8 +
                                      (16 + 2 + 2 + 2 + 12 + 4 + 3 + 1 + 12) * 64 * 8 + 2 * kTileBytes)
                             : per * (3 * 2 * kTileBytes + (2 * 64 * 3 + 2 * (ext ? 7 : 6) * 64) * elem + 2 * 64 * 4 + 7 * 4) +
                                   (layout == COOP_RING_LDS ? SIPNET_RING_SLOTS * 64 : 64) * 8;
  }
}
#else
void launchStepCoop(const FastArgs& a, int precision, int layout, hipStream_t stream, LaunchInfo* info) {
#ifdef SIPNET_COOP_BOUNDED
  {
    static const unsigned long long zero[2] = {0ull, 0ull};
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_coopStuck), zero, sizeof zero, 0, hipMemcpyHostToDevice, stream);
  }
#endif
  const int chunksPerSite = (a.n_members + 63) / 64;
  const bool ringInLds = layout == COOP_RING_LDS, pair = layout == COOP_PAIR, quad = layout == COOP_QUAD;
  // flags beyond the compiled-in set of the layout's family: the optional-physics instantiations (run-time flags)
  const bool nFamily = layout == COOP_NCYCLE || layout == COOP_NCYCLE_PAIR;
  const bool ext = nFamily ? !isNCycleFlagSet(a.flags) : !isDefaultFlagSet(a.flags);
  if (nFamily) {
    const bool pairN = layout == COOP_NCYCLE_PAIR;
    const int chunksN = a.n_sites * chunksPerSite;
    const int groupsN = (a.n_sites & 7) == 0 ? 8 * ((chunksN / 8 + 1) / 2) : (chunksN + 1) / 2;   // (see pairGroups below)
    const dim3 gridN(pairN ? groupsN : chunksN), blockN(pairN ? 512 : 256);
#define NCYC_LAUNCH(K)                                                                          \
    if (precision == SIPNET_F64) {                                                                \
      if (a.plainExp) hipLaunchKernelGGL((K<double, true>), gridN, blockN, 0, stream, a);         \
      else hipLaunchKernelGGL((K<double, false>), gridN, blockN, 0, stream, a);                   \
    } else {                                                                                      \
      if (a.plainExp) hipLaunchKernelGGL((K<float, true>), gridN, blockN, 0, stream, a);          \
      else hipLaunchKernelGGL((K<float, false>), gridN, blockN, 0, stream, a);                    \
    }
#ifndef SIPNET_COOP_BOUNDED
    const bool pairDiag = pairN && a.full && a.diag != nullptr;
    if (a.sumEvery > 0) {   // (fp64, lean: the engine asks for nothing else)
#define NSUMS(K, P, E) hipLaunchKernelGGL((K<P, E>), gridN, blockN, 0, stream, a)
      if (pairN) {
        if (a.plainExp) { if (ext) NSUMS(stepCoopNPairSumsKernel, true, true); else NSUMS(stepCoopNPairSumsKernel, true, false); }
        else { if (ext) NSUMS(stepCoopNPairSumsKernel, false, true); else NSUMS(stepCoopNPairSumsKernel, false, false); }
      } else {
        if (a.plainExp) { if (ext) NSUMS(stepCoopNSumsKernel, true, true); else NSUMS(stepCoopNSumsKernel, true, false); }
        else { if (ext) NSUMS(stepCoopNSumsKernel, false, true); else NSUMS(stepCoopNSumsKernel, false, false); }
      }
#undef NSUMS
    } else if (pairDiag) {
      if (ext) { NCYC_LAUNCH(stepCoopNXPairDiagKernel) } else { NCYC_LAUNCH(stepCoopNPairDiagKernel) }
    } else if (ext && a.full) {
      if (pairN) { NCYC_LAUNCH(stepCoopNXPairFullKernel) } else { NCYC_LAUNCH(stepCoopNXFullKernel) }
    } else
#endif
    if (ext) {
      if (pairN) { NCYC_LAUNCH(stepCoopNXPairKernel) } else { NCYC_LAUNCH(stepCoopNXKernel) }
#ifndef SIPNET_COOP_BOUNDED
    } else if (a.full) {
      if (pairN) { NCYC_LAUNCH(stepCoopNPairFullKernel) } else { NCYC_LAUNCH(stepCoopNFullKernel) }
#endif
    } else {
      if (pairN) { NCYC_LAUNCH(stepCoopNPairKernel) } else { NCYC_LAUNCH(stepCoopNKernel) }
    }
#undef NCYC_LAUNCH
    if (info) {
      if (a.sumEvery > 0)
        snprintf(info->kernel, sizeof info->kernel, "%s<%s, %s>", pairN ? "stepCoopNPairSumsKernel" : "stepCoopNSumsKernel",
                 a.plainExp ? "true" : "false", ext ? "true" : "false");
      else
      snprintf(info->kernel, sizeof info->kernel, "%s<%s, %s>",
               (pairN && a.full && a.diag) ? (ext ? "stepCoopNXPairDiagKernel" : "stepCoopNPairDiagKernel")
               : (ext && a.full) ? (pairN ? "stepCoopNXPairFullKernel" : "stepCoopNXFullKernel")
               : ext ? (pairN ? "stepCoopNXPairKernel" : "stepCoopNXKernel")
                   : a.full ? (pairN ? "stepCoopNPairFullKernel" : "stepCoopNFullKernel") : (pairN ? "stepCoopNPairKernel" : "stepCoopNKernel"),
               precision == SIPNET_F64 ? "double" : "float", a.plainExp ? "true" : "false");
      info->grid = (int32_t)gridN.x;
      info->block = pairN ? 512 : 256;
      info->wavesPerSimd = pairN ? 2 : 1;
      const int elem = precision == SIPNET_F64 ? 8 : 4;
      info->ldsBytes = (pairN ? 2 : 1) * (3 * 2 * kTileBytes + (2 * 64 * 3 + 2 * 7 * 64) * elem + 2 * 64 * 4 + 16 * 4 + 64 * 8 +
                                          (16 + 2 + 2 + 2 + 12 + 4 + 3 + 1 + 12) * 64 * 8 + 2 * kTileBytes);
    }
    return;
  }
  const int chunks = a.n_sites * chunksPerSite;
  // paired chunks: with the XCD-grouped mapping every group of eight workgroups carries 16 chunks
  const int pairGroups = (a.n_sites & 7) == 0 ? 8 * ((chunks / 8 + 1) / 2) : (chunks + 1) / 2;
  const int quadGroups = (a.n_sites & 7) == 0 ? 8 * ((chunks / 8 + 3) / 4) : (chunks + 3) / 4;
  const dim3 grid(pair ? pairGroups : quad ? quadGroups : chunks), block(pair ? 512 : quad ? 768 : ringInLds ? 256 : 192);
#ifdef SIPNET_COOP_BOUNDED   // (lean instantiations only: the engine does not send full-state launches here)
#define COOP_LAUNCH(R, P, L) { hipLaunchKernelGGL((stepCoopKernel<R, P, L, false>), grid, block, 0, stream, a); }
#define PAIR_LAUNCH(R, P) { hipLaunchKernelGGL((stepCoopPairKernel<R, P, false>), grid, block, 0, stream, a); }
#else
#define COOP_LAUNCH(R, P, L)                                                                        \
  {                                                                                                 \
    if (a.full) hipLaunchKernelGGL((stepCoopKernel<R, P, L, true>), grid, block, 0, stream, a);      \
    else hipLaunchKernelGGL((stepCoopKernel<R, P, L, false>), grid, block, 0, stream, a);           \
  }
#define PAIR_LAUNCH(R, P)                                                                           \
  {                                                                                                 \
    if (a.full) hipLaunchKernelGGL((stepCoopPairKernel<R, P, true>), grid, block, 0, stream, a);     \
    else hipLaunchKernelGGL((stepCoopPairKernel<R, P, false>), grid, block, 0, stream, a);          \
  }
#endif
#ifndef SIPNET_COOP_BOUNDED
  if (a.sumEvery > 0 && ext) {   // (the engine sends fp64, lean launches of these three layouts only)
    if (pair) {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopXPairSumsKernel<true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopXPairSumsKernel<false>), grid, block, 0, stream, a);
    } else if (ringInLds) {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopXSumsKernel<true, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopXSumsKernel<false, true>), grid, block, 0, stream, a);
    } else {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopXSumsKernel<true, false>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopXSumsKernel<false, false>), grid, block, 0, stream, a);
    }
  } else if (a.sumEvery > 0) {
    if (pair) {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopPairSumsKernel<true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopPairSumsKernel<false>), grid, block, 0, stream, a);
    } else if (ringInLds) {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopSumsKernel<true, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopSumsKernel<false, true>), grid, block, 0, stream, a);
    } else {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopSumsKernel<true, false>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopSumsKernel<false, false>), grid, block, 0, stream, a);
    }
  } else
#endif
  if (ext) {   // (one or two chunks per workgroup, lean: the engine does not ask for anything else)
#define X_LAUNCH2(R, P, F)                                                                                    \
  {                                                                                                           \
    if (quad) hipLaunchKernelGGL((stepCoopXQuadKernel<float, P>), grid, block, 0, stream, a);                 \
    else if (pair) hipLaunchKernelGGL((stepCoopXPairKernel<R, P, F>), grid, block, 0, stream, a);             \
    else if (ringInLds) hipLaunchKernelGGL((stepCoopXKernel<R, P, true, F>), grid, block, 0, stream, a);      \
    else hipLaunchKernelGGL((stepCoopXKernel<R, P, false, F>), grid, block, 0, stream, a);                    \
  }
#ifdef SIPNET_COOP_BOUNDED
#define X_LAUNCH(R, P) { X_LAUNCH2(R, P, false) }
#else
#define X_LAUNCH(R, P) { if (a.full) X_LAUNCH2(R, P, true) else X_LAUNCH2(R, P, false) }
#endif
    if (precision == SIPNET_F64) { if (a.plainExp) X_LAUNCH(double, true) else X_LAUNCH(double, false) }
    else { if (a.plainExp) X_LAUNCH(float, true) else X_LAUNCH(float, false) }
#undef X_LAUNCH
#undef X_LAUNCH2
  } else if (quad) {
    if (precision == SIPNET_F64) {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopQuadKernel<double, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopQuadKernel<double, false>), grid, block, 0, stream, a);
    } else {
      if (a.plainExp) hipLaunchKernelGGL((stepCoopQuadKernel<float, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((stepCoopQuadKernel<float, false>), grid, block, 0, stream, a);
    }
  } else if (pair) {
    if (precision == SIPNET_F64) { if (a.plainExp) PAIR_LAUNCH(double, true) else PAIR_LAUNCH(double, false) }
    else { if (a.plainExp) PAIR_LAUNCH(float, true) else PAIR_LAUNCH(float, false) }
  } else if (precision == SIPNET_F64) {
    if (a.plainExp) { if (ringInLds) COOP_LAUNCH(double, true, true) else COOP_LAUNCH(double, true, false) }
    else { if (ringInLds) COOP_LAUNCH(double, false, true) else COOP_LAUNCH(double, false, false) }
  } else {
    if (a.plainExp) { if (ringInLds) COOP_LAUNCH(float, true, true) else COOP_LAUNCH(float, true, false) }
    else { if (ringInLds) COOP_LAUNCH(float, false, true) else COOP_LAUNCH(float, false, false) }
  }
#undef COOP_LAUNCH
#undef PAIR_LAUNCH
  if (info) {
    const char* r = precision == SIPNET_F64 ? "double" : "float";
    const char* pe = a.plainExp ? "true" : "false";
    const char* fu = a.full ? "true" : "false";
    if (a.sumEvery > 0 && pair) snprintf(info->kernel, sizeof info->kernel, "stepCoop%sPairSumsKernel<%s>", ext ? "X" : "", pe);
    else if (a.sumEvery > 0) snprintf(info->kernel, sizeof info->kernel, "stepCoop%sSumsKernel<%s, %s>", ext ? "X" : "", pe, ringInLds ? "true" : "false");
    else if (ext && quad) snprintf(info->kernel, sizeof info->kernel, "stepCoopXQuadKernel<float, %s>", pe);
    else if (ext && pair) snprintf(info->kernel, sizeof info->kernel, "stepCoopXPairKernel<%s, %s, %s>", r, pe, fu);
    else if (ext) snprintf(info->kernel, sizeof info->kernel, "stepCoopXKernel<%s, %s, %s, %s>", r, pe, ringInLds ? "true" : "false", fu);
    else if (quad) snprintf(info->kernel, sizeof info->kernel, "stepCoopQuadKernel<%s, %s>", r, pe);
    else if (pair) snprintf(info->kernel, sizeof info->kernel, "stepCoopPairKernel<%s, %s, %s>", r, pe, fu);
    else snprintf(info->kernel, sizeof info->kernel, "stepCoopKernel<%s, %s, %s, %s>", r, pe,
                  ringInLds ? "true" : "false", fu);
    info->grid = (int32_t)grid.x;
    info->block = (int32_t)block.x;
    info->wavesPerSimd = pair ? 2 : quad ? 3 : 1;
    const int elem = precision == SIPNET_F64 ? 8 : 4;
    info->ldsBytes = (pair ? 2 : quad ? 4 : 1) * (3 * 2 * kTileBytes + (2 * 64 * 3 + 2 * (ext ? 7 : 6) * 64) * elem + 2 * 64 * 4 + 7 * 4) +
                     (ringInLds ? SIPNET_RING_SLOTS * 64 : 64) * 8;
  }
}

#endif   // !SIPNET_COOP_SUMS_TU
#ifdef SIPNET_COOP_BOUNDED
}  // namespace bounded
#endif
#ifdef SIPNET_COOP_SUMS_TU
}  // namespace sums2
#endif
}  // namespace sipnet

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
template <class R, bool PlainExp, bool RingLds, bool Full, int NP, bool NCyc, bool Ext>
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

// (the Sums instantiations' accumulators; an empty type otherwise, so that the other instantiations' code is what it was)
template <bool On>
struct CoopSums {
  double nee = 0.0, et = 0.0, gpp = 0.0;
  int left = 0;
};
template <>
struct CoopSums<false> {};
// Sums (round 6; fp64, default physics, lean): the three output planes receive every member's SUMS over groups of
// a.sumEvery consecutive steps of the launch instead of the steps themselves -- [groups][ld] each, a row per group, the last
// group as long as the launch leaves it -- accumulated in step order by the wave that computes the value (one add per value and
// step, a store per group: 1 / sumEvery of the planes' HBM writes; sipnet_batch_run_sums).  Sums = false compiles to the code
// it was before the parameter existed (tests/test_code_placement.py holds the measured instantiations' loop heads in place).
template <class R, bool PlainExp, bool RingLds, bool Full, int NP, bool NCyc = false, bool Ext = false, bool Sums = false>
__device__ __forceinline__ void coopBody(const FastArgs& a) {
  static_assert(!Sums || (!NCyc && !Ext && !Full && sizeof(R) == 8), "in-kernel sums: fp64, default physics, lean launches");
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
  // per-wave private record tiles (each wave stages and awaits its own DMA) + mailboxes
  __shared__ alignas(16) unsigned char ldsTilesAll[NP][NCyc ? 4 : 3][2 * kTileBytes];
  __shared__ R mailLaiAll[NP][2][64], mailPgpAll[NP][2][64], mailPsnAll[NP][2][64];
  // rows 0..4: g1 g2 qSoilT gFine gCoarse of a step (wave L: climate x parameters only); row 5: the
  // soil-moisture effect on heterotrophic respiration (wave W: its state)
  // (NCyc: row 6 = the plain soil-temperature Q10 factor, for methane, volatilisation, litter breakdown)
  // (Opt: row 6 = wave W's methane moisture term, behind W's flag like row 5)
  __shared__ alignas(16) R mailFacAll[NP][2][(NCyc || Ext) ? 7 : 6][64];
  __shared__ int mailAliveAll[NP][2][64];  // aliveWord(): wave C's confirmation of the leaf area it posted
  __shared__ int seqLaiAll[NP], seqPgpAll[NP], seqPsnAll[NP];
  __shared__ alignas(8) int seqFacMoistAll[NP][2];  // [0] wave F's / L's factor rows, [1] wave W's moisture row
  __shared__ int seqDoneAll[NP][2];  // statistics: wave C's / wave W's plane stores of the whole launch have completed
  // Two- and four-chunk layouts (no wave F; the planes do not stay in L2 there: the HBM rings of 512-1 024
  // chunks stream through it): with sipnet_batch_run_stats waves C and W also put every value they store
  // into an LDS staging block -- two halves of kStageR steps -- and wave L sums a half, one plane per
  // step, once both are past it: the planes are never read back.  (Four chunks in fp64 have no LDS left
  // for it; such a launch is followed by the reduction passes.)
  constexpr bool Staged = NP >= 2 && !Full && !NCyc && !(NP == 4 && sizeof(R) == 8);
  constexpr int kStageR = NP == 4 ? 4 : 8;
  // (rows kStageLpr elements longer than the 64 members: the summing lanes of neighbouring rows then hit different banks)
  constexpr int kStageLpr = kStageR == 4 ? 4 : 2;   // lanes that share the sum of a row
  __shared__ alignas(16) R stageAll[NP][3][Staged ? 2 * kStageR : 1][64 + kStageLpr];
  // NCyc hand-overs (doubles whatever R is).  C -> W per step: leafLitter woodLitter fineRootLoss
  // coarseRootLoss nDemand reductionNResorption leafOnN(all) leafOnN(computed switch) [rates] early in its
  // step and GPP - R_a of the step at its end (S has R_h: it forms, stores and totals NEE); S -> C per
  // step: the mineral N at the start of the step; after the mortality hand-over: storage N.  Rare: C -> W the soil-side
  // increments of events [litterC soilC minN soilOrgN litterN storN, rates] and of plant death
  // [to soilC, to litterC, to soilOrgN, to litterN]; a nitrogen-limited step: W -> C {availableMinN,
  // fixation share, unclaimed storage}, C -> W the final demand.
  // (plant fluxes and event increments in two slots: C posts a step's before it has S's mineral nitrogen
  // of that step, i.e. possibly before S has consumed the step before)
  // (one chunk: plain arrays, as the one-chunk kernel was tuned -- the per-chunk ones cost it 1.3 %; two chunks: per chunk)
  constexpr int NPN = (NCyc && NP > 1) ? NP : 1;
  // (mailPend: GPP - R_a of the step; a full-state launch adds R_a and the root respiration -- wave S writes the
  // record's R_soil / R_tot columns and carries those accumulators)
  // (one-chunk full-state launches with the diagnostics counters: eleven more rows -- the plant side's mass totals before
  // the step, after the pool updates and after the clamps [carbon, nitrogen], its carbon input and output rates, the
  // events' carbon output and nitrogen in / out: what wave S needs for checkBalance(), balance.c:122-169.  The
  // two-chunk layout has no LDS for them: such launches take the one-wave kernel)
  constexpr int kDiagRows = 11;
  constexpr int kPendRows = (NCyc && Full) ? (NP == 1 ? 3 + kDiagRows : 3) : 1;
  __shared__ alignas(16) double mailPlant1[NCyc ? 2 : 1][NCyc ? 8 : 1][64], mailPend1[NCyc ? 2 : 1][kPendRows][64], mailMinN1[NCyc ? 2 : 1][64];
  __shared__ alignas(16) double mailStorN1[NCyc ? 2 : 1][64], mailEvent1[NCyc ? 2 : 1][NCyc ? 6 : 1][64], mailDeath1[NCyc ? 4 : 1][64];
  __shared__ alignas(16) double mailSupply1[NCyc ? 3 : 1][64], mailDemand1[1][64];
  __shared__ alignas(16) double mailPlantAll[NPN][NCyc ? 2 : 1][NCyc ? 8 : 1][64], mailPendAll[NPN][NCyc ? 2 : 1][kPendRows][64],
      mailMinNAll[NPN][NCyc ? 2 : 1][64];
  __shared__ alignas(16) double mailStorNAll[NPN][NCyc ? 2 : 1][64], mailEventAll[NPN][NCyc ? 2 : 1][NCyc ? 6 : 1][64],
      mailDeathAll[NPN][NCyc ? 4 : 1][64];
  __shared__ alignas(16) double mailSupplyAll[NPN][NCyc ? 3 : 1][64], mailDemandAll[NPN][1][64];
  // wave W -> wave S per step: [anaerobic moisture effect, anoxic share] at its start (seqWat), the
  // leached share of the mineral nitrogen once the drainage is known (seqLeach)
  // (four slots: W runs at most one step ahead of C, and C at most two ahead of S's consumption)
  __shared__ alignas(16) double mailWat1[NCyc ? 4 : 1][NCyc ? 3 : 1][64];
  __shared__ int seqPlant1, seqMinN1, seqStorN1, seqEvent1, seqSupply1, seqDemand1, seqWat1, seqLeach1;
  __shared__ alignas(16) double mailWatAll[NPN][NCyc ? 4 : 1][NCyc ? 3 : 1][64];
  __shared__ int seqNAll[NPN][8];   // seqPlant seqMinN seqStorN seqEvent seqSupply seqDemand seqWat seqLeach
#define seqFac seqFacMoist[0]
#define seqMoist seqFacMoist[1]
  // The running-mean ring of the 64 members lives in LDS for the whole launch (250 x 64 x 8 B =
  // 125 KB; one workgroup per CU).  A wave that stores to HBM every step must not also load
  // from HBM every step: vector-memory operations complete in issue order, so each step's ring
  // loads would queue behind the previous step's output stores (~1800 cycles to their ack).
  __shared__ double ringL[RingLds ? SIPNET_RING_SLOTS * 64 : 64];

  const int wave = uni((int)threadIdx.x >> 6);
  // 0 carbon, 1 water, 2 light; -1: a placeholder wave that only keeps the SIMD rotation
  // which of the workgroup's chunks; NP == 4: C0..C3 W0..W3 L0..L3, a chunk's three waves on one SIMD
  // pairs: the second four wavefronts serve the OTHER chunk of their SIMD's first one
  const int sub = Pair ? ((NCyc ? (wave >> 2) == 1 : (wave >> 1) == 2) ? ((wave & 1) ^ 1) : (wave & 1)) : NP == 4 ? (wave & 3) : 0;
  // (NCyc pair: C0 C1 S0 S1 | W1 W0 L1 L0 -> roles 0 0 3 3 1 1 2 2)
  const int role = Pair ? (NCyc ? ((wave >> 1) == 0 ? 0 : (wave >> 1) == 1 ? 3 : (wave >> 1) == 2 ? 1 : 2) : (wave >> 1))
                        : NP == 4 ? (wave >> 2) : wave;
  const int lane = (int)threadIdx.x & 63;
  auto& mailLai = mailLaiAll[sub];
  auto& mailPgp = mailPgpAll[sub];
  auto& mailPsn = mailPsnAll[sub];
  auto& mailFac = mailFacAll[sub];
  auto& mailAlive = mailAliveAll[sub];
  int& seqLai = seqLaiAll[sub];
  int& seqPgp = seqPgpAll[sub];
  int& seqPsn = seqPsnAll[sub];
  auto& seqFacMoist = seqFacMoistAll[sub];
  auto& seqDone = seqDoneAll[sub];
  auto& stage = stageAll[sub];
  constexpr bool OneN = !(NCyc && NP > 1);
  const int subN = OneN ? 0 : sub;
  auto& mailPlant = OneN ? mailPlant1 : mailPlantAll[subN];
  auto& mailPend = OneN ? mailPend1 : mailPendAll[subN];
  auto& mailMinN = OneN ? mailMinN1 : mailMinNAll[subN];
  auto& mailStorN = OneN ? mailStorN1 : mailStorNAll[subN];
  auto& mailEvent = OneN ? mailEvent1 : mailEventAll[subN];
  auto& mailDeath = OneN ? mailDeath1 : mailDeathAll[subN];
  auto& mailSupply = OneN ? mailSupply1 : mailSupplyAll[subN];
  auto& mailDemand = OneN ? mailDemand1 : mailDemandAll[subN];
  auto& mailWat = OneN ? mailWat1 : mailWatAll[subN];
  int& seqPlant = OneN ? seqPlant1 : seqNAll[subN][0];
  int& seqMinN = OneN ? seqMinN1 : seqNAll[subN][1];
  int& seqStorN = OneN ? seqStorN1 : seqNAll[subN][2];
  int& seqEvent = OneN ? seqEvent1 : seqNAll[subN][3];
  int& seqSupply = OneN ? seqSupply1 : seqNAll[subN][4];
  int& seqDemand = OneN ? seqDemand1 : seqNAll[subN][5];
  int& seqWat = OneN ? seqWat1 : seqNAll[subN][6];
  int& seqLeach = OneN ? seqLeach1 : seqNAll[subN][7];
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
  asm volatile("s_nop %1\n .p2align 5\n .rept %0\n s_nop 0\n .endr" ::"n"(coopCodePhase<R, PlainExp, RingLds, Full, NP, NCyc, Ext>(ROLE)), "n"(8 + (ROLE)))
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

  // =============================================================================================
  // ---- ensemble statistics of the output planes (a.statsPart; sipnet_batch_run_stats) ------------
  // One wavefront of the workgroup -- wave F where there is one (it runs up to two steps ahead of C and
  // is busy a quarter of the time), else wave L -- sums the plane tiles waves C and W have stored,
  // three tiles behind them, while the lines are still in L2: lane 4r + q takes 16 of the chunk's 64
  // columns of step 16k + r, two quad permutations join the four quarters, lane 4r stores the row's
  // (sum, sum of squares) into the per-chunk block that launchFinishStats adds up.  Why the data is
  // there: C and W drain their memory queue at every tile start down to the last step's stores (the
  // s_waitcnt in front of their tile DMA; memory operations complete in issue order), and C takes W's
  // moisture factor of a step before it posts the next leaf area -- so once C has posted the leaf
  // area of the SECOND step of tile k + 2, both waves' stores of tile k have reached L2.  Ordinary
  // loads: every line read here was stored through THIS compute unit's vector L1 (write-through) by
  // this workgroup and is read once, so a hit there is current and a miss goes to that L2.
  // Which 16 columns a lane takes is chosen for the vector L1, which serves one 128-byte line per
  // cycle to the whole compute unit (C's and W's stores and tile DMAs queue behind these loads): the
  // four lanes of a row read NEIGHBOURING 16-byte pieces (fp64: columns 8k + 2q, 8k + 2q + 1 on load
  // k; fp32: 16k + 4q .. + 3), 16 lines per load instruction -- with 16 consecutive columns per lane
  // every lane of every load hit a line of its own (1.3 us of L1 time per tile at c10k, 3 us at c4).
  // One plane per turn, on steps 4, 9 and 14 of a tile (never right behind this wave's own tile
  // loads), load + wait + sum + store in one go.  Measured alternatives: loads issued five
  // steps before their sums, an L1 prefetch two steps before the turn, half planes per turn, the turn
  // after the factor post -- none cheaper; what costs is the loads themselves (they take C's and W's
  // place in the memory pipeline), not the wait for them.
#ifdef SIPNET_NO_STATS   // (A/B probe: what the statistics machinery costs a launch that does not use it)
  const bool statsHere = false;
#else
  const bool statsHere = a.statsPart != nullptr && role == (FacWave ? 3 : 2);
#endif
  int statTile = tBegin / kFastTile;                    // tile being summed
  int statPlane = 0;                                    // its next plane
  int statNext = (statTile + 3) * kFastTile + 4;        // step of this wave on which that plane is due
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  constexpr int kPer16 = 16 / (int)sizeof(R);          // elements per 16-byte load
  constexpr int kLoads = 16 / kPer16;                  // 16-byte loads per lane and plane
  // 16-byte loads need a full chunk and 16-byte aligned row segments; the ragged last chunk of a
  // site and odd layouts go element by element (column 4e + q) with clamped addresses
  const bool statFast = uni((int)((chunk << 6) + 64 <= a.n_members && (((a.ld | a.n_members) & (kPer16 - 1)) == 0) &&
                                  ((((uintptr_t)a.nee | (uintptr_t)a.gpp | (uintptr_t)a.et) & 15) == 0))) != 0;
  const int statRow = lane >> 2, statQ = lane & 3;
  // per-lane cursors: this lane's piece of row 16 * statTile + statRow of each plane and of the
  // partial-sum block, advanced by one tile per round (no multiplications in the loop)
  const int64_t statRow0 = (int64_t)statTile * kFastTile + statRow - tBegin;   // < 0 for rows before a launch that starts inside a tile
  const int64_t statOff0 = statRow0 * a.ld + ((int64_t)site * a.n_members + (chunk << 6) + (statFast ? kPer16 * statQ : 0));
  const R* statPtr0 = (const R*)a.nee + statOff0;
  const R* statPtr1 = (const R*)a.gpp + statOff0;
  const R* statPtr2 = (const R*)a.et + statOff0;
  const int64_t statTileStride = (int64_t)kFastTile * a.ld;
  const int64_t statPlaneStride = (int64_t)a.statsChunks * a.n_steps * 2;
  double* statDst = a.statsPart + (((int64_t)site * chunksPerSite + chunk) * a.n_steps + statRow0) * 2;
  auto quadSum = [](double v) -> double {  // over the four lanes of a row: two quad permutations (DPP)
    auto perm = [](double x, auto ctrl) -> double {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), decltype(ctrl)::value, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), decltype(ctrl)::value, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    v += perm(v, std::integral_constant<int, 0xB1>{});   // quad_perm:[1,0,3,2]
    v += perm(v, std::integral_constant<int, 0x4E>{});   // quad_perm:[2,3,0,1]
    return v;
  };
  auto statPlaneTurn = [&](const R* ptr, double* dst) {
    const int t = statTile * kFastTile + statRow;
    bool rowOk = true;
    // rows outside the launch (a first or last, partial tile) are read from the nearest row inside
    // and not stored
    if (__builtin_expect(statTile * kFastTile < tBegin || (statTile + 1) * kFastTile > tEnd, 0)) {
      const int tc = t < tBegin ? tBegin : t >= tEnd ? tEnd - 1 : t;
      ptr += (int64_t)(tc - t) * a.ld;
      rowOk = tc == t;
    }
    double s1 = 0.0, s2 = 0.0;
    if (__builtin_expect(statFast, 1)) {
      const u64x2* src = (const u64x2*)ptr;
      u64x2 v[kLoads];
#pragma unroll
      for (int k = 0; k < kLoads; k++) v[k] = src[4 * k];
#pragma unroll
      for (int k = 0; k < kLoads; k++) {
        if (sizeof(R) == 8) {
          const double x0 = __longlong_as_double((long long)v[k].x), x1 = __longlong_as_double((long long)v[k].y);
          s1 += x0;
          s2 = __builtin_fma(x0, x0, s2);
          s1 += x1;
          s2 = __builtin_fma(x1, x1, s2);
        } else {
#pragma unroll
          for (int h = 0; h < 4; h++) {
            const unsigned long long w = (h & 2) ? v[k].y : v[k].x;
            const double x = (double)__uint_as_float((unsigned)(w >> (32 * (h & 1))));
            s1 += x;
            s2 = __builtin_fma(x, x, s2);
          }
        }
      }
    } else {
      const int nLeft = a.n_members - (chunk << 6);   // >= 1: the chunk exists
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int c = 4 * e + statQ;
        const double x = c < nLeft ? (double)ptr[c < nLeft ? c : 0] : 0.0;
        s1 += x;
        s2 = __builtin_fma(x, x, s2);
      }
    }
    s1 = quadSum(s1);
    s2 = quadSum(s2);
    if (statQ == 0 && rowOk) {
      typedef double d2s __attribute__((ext_vector_type(2)));
      *(d2s*)dst = d2s{s1, s2};
    }
  };
  // the plane that is due; `settled`: C and W have finished (nothing of theirs is in flight any more)
  auto statAct = [&](bool settled) {
    if (statPlane == 0) {
      if (!settled) awaitAtLeast(&seqLai, (statTile + 2) * kFastTile + 1);
      statPlaneTurn(statPtr0, statDst);
    } else if (statPlane == 1) {
      statPlaneTurn(statPtr1, statDst + statPlaneStride);
    } else {
      statPlaneTurn(statPtr2, statDst + 2 * statPlaneStride);
      statPtr0 += statTileStride;
      statPtr1 += statTileStride;
      statPtr2 += statTileStride;
      statDst += 2 * kFastTile;
    }
    const bool last = statPlane == 2;
    statPlane = last ? 0 : statPlane + 1;
    statTile += last ? 1 : 0;
    statNext += last ? 6 : 5;
  };
  // ---- the staged variant (two- / four-chunk layouts): once C and W are past a half of kStageR steps (C has
  // posted the leaf area of the step after it) the light wave sums the half's 3 x kStageR rows in ONE action:
  // kStageLpr lanes per row (48 lanes busy), each 64 / kStageLpr interleaved columns, a DPP permutation or two
  // to join them.  (The first version took a plane per step with 64 / kStageR lanes per row: 52 instructions
  // per step on the four-chunk layout, where the three waves of a chunk share a SIMD; this one 22.)
  int stageBlock = 0;          // half being summed: steps [tBegin + stageBlock * kStageR, + kStageR)
  auto stagedHalf = [&](int tLimit) {
    constexpr int LPR = kStageLpr, NV = 64 / LPR;
    const int row = lane / LPR, c0 = lane % LPR;           // row = plane * kStageR + step of the half
    const bool rowOk = row < 3 * kStageR;
    const int p = rowOk ? row / kStageR : 0, r = row % kStageR;
    const int rowIdx = (stageBlock & 1) * kStageR + r;
    const R* src = &stage[p][rowIdx][c0];
    double s1 = 0.0, s2 = 0.0;
    if (__builtin_expect((chunk << 6) + 64 > a.n_members, 0)) {
      // (rare: the ragged last chunk) columns past the site's last member hold copies of that member
      // (clamped lanes) and are not part of the sums: element by element, with the mask
      for (int i = 0; i < NV; i++) {
        const int c = c0 + LPR * i;
        const double x = ((chunk << 6) + c < a.n_members) ? (double)stage[p][rowIdx][c] : 0.0;
        s1 += x;
        s2 = __builtin_fma(x, x, s2);
      }
    } else {
#pragma unroll
      for (int g = 0; g < NV; g += 8) {   // eight reads in flight; columns c0 + LPR (g + k)
        R v0, v1, v2, v3, v4, v5, v6, v7;
        const unsigned base = ldsAddr(src) + (unsigned)(g * LPR * sizeof(R));
        if (sizeof(R) == 8)        // LPR == 2: 16 bytes apart
          asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:32\n\t"
                       "ds_read_b64 %3, %8 offset:48\n\tds_read_b64 %4, %8 offset:64\n\tds_read_b64 %5, %8 offset:80\n\t"
                       "ds_read_b64 %6, %8 offset:96\n\tds_read_b64 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                       : "v"(base) : "memory");
        else if (LPR == 4)         // 16 bytes apart
          asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:16\n\tds_read_b32 %2, %8 offset:32\n\t"
                       "ds_read_b32 %3, %8 offset:48\n\tds_read_b32 %4, %8 offset:64\n\tds_read_b32 %5, %8 offset:80\n\t"
                       "ds_read_b32 %6, %8 offset:96\n\tds_read_b32 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                       : "v"(base) : "memory");
        else                       // floats, LPR == 2: 8 bytes apart
          asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:8\n\tds_read_b32 %2, %8 offset:16\n\t"
                       "ds_read_b32 %3, %8 offset:24\n\tds_read_b32 %4, %8 offset:32\n\tds_read_b32 %5, %8 offset:40\n\t"
                       "ds_read_b32 %6, %8 offset:48\n\tds_read_b32 %7, %8 offset:56\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                       : "v"(base) : "memory");
        const R v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const double x = (double)v[i];
          s1 += x;
          s2 = __builtin_fma(x, x, s2);
        }
      }
    }
    auto dppAdd = [](double v, auto ctrl) -> double {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), decltype(ctrl)::value, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), decltype(ctrl)::value, 0xf, 0xf, false);
      return v + __hiloint2double(hi, lo);
    };
    auto rowSum = [&](double v) -> double {
      v = dppAdd(v, std::integral_constant<int, 0xB1>{});                 // quad_perm:[1,0,3,2]
      if (LPR == 4) v = dppAdd(v, std::integral_constant<int, 0x4E>{});   // quad_perm:[2,3,0,1]
      return v;
    };
    s1 = rowSum(s1);
    s2 = rowSum(s2);
    const int t = tBegin + stageBlock * kStageR + r;
    if (c0 == 0 && rowOk && t < tLimit) {
      const int64_t gChunk = (int64_t)site * chunksPerSite + chunk;
      double* dst = a.statsPart + (((int64_t)p * a.statsChunks + gChunk) * a.n_steps + (t - tBegin)) * 2;
      typedef double d2s __attribute__((ext_vector_type(2)));
      *(d2s*)dst = d2s{s1, s2};
    }
  };
  // the staged action that is due (statNext): the half C and W have just left.  They cannot come back to its
  // rows before this wave has posted the factors of the step after next.
  auto stagedAct = [&]() {
    const int blockEnd = tBegin + (stageBlock + 1) * kStageR;
    awaitAtLeast(&seqLai, blockEnd + 1);
    stagedHalf(tEnd);
    stageBlock++;
    statNext = tBegin + (stageBlock + 1) * kStageR + 1;
  };
  auto stagedFinish = [&]() {  // after the loop: what is left, once C and W have finished
    awaitAtLeast(&seqDone[0], 1);
    awaitAtLeast(&seqDone[1], 1);
    while (tBegin + stageBlock * kStageR < tEnd) {
      stagedHalf(tEnd);
      stageBlock++;
    }
  };
  if (Staged) statNext = tBegin + kStageR + 1;
  auto statFinish = [&]() {  // after the wave's loop: the last tiles, once C and W have drained their stores
    awaitAtLeast(&seqDone[0], 1);
    awaitAtLeast(&seqDone[1], 1);
    while (statTile * kFastTile < tEnd) statAct(true);
  };

  // =============================================================================================
  // ---- F (one workgroup per CU only: the CU's fourth SIMD is free): the climate / parameter part
  // of wave C's respiration terms (vegResp sipnet.c:1051-1068, calcRootResp :1073,
  // calcSoilRespiration :1132-1148 with depeffects.c:71-74):  folResp = leafC * g1,
  // rVeg = folResp + totalWoodC * g2,  rSoil = soilC * (qSoilT * moistEff[wave W]),
  // rFineRoot = fineRootC * gFine,  rCoarseRoot = coarseRootC * gCoarse.  Nothing here depends on
  // member state.  By day the light wave is the busiest of the three (seven exp2 of the canopy
  // layers on top of these one to four); with the factors on a wave of their own the day step is
  // the carbon wave's again.  No room for a fourth record tile in LDS (ring 125 KB + 3 tiles
  // + mailboxes = 158.5 of 160 KB): lane k loads the five fields of step 16j + k one tile ahead
  // and the step's values are read back with v_readlane.
  if (FacWave && role == 3) {
#pragma clang fp contract(off)  // same bits as the light wave's copy of this block (see there)
    COOP_CODE_PHASE(3);
    const R K_frozThr = (R)PRM(frozenSoilThreshold);
    const R K_lgVeg = (R)log2(PRM(vegRespQ10)), K_lgSoil = (R)log2(PRM(soilRespQ10));
    const R K_lgFine = (R)log2(PRM(fineRootQ10)), K_lgCoarse = (R)log2(PRM(coarseRootQ10));
    const R K_fol = (R)((PRM(baseFolRespFrac) * PRM(aMax)) *
                        (kCWeight * (1.0 / kTen9) * (PRM(leafCSpWt) / PRM(cFracLeaf)) * kSecPerDay) *
                        (1.0 / PRM(leafCSpWt)) * exp2(-(PRM(psnTOpt) / 10.0) * log2(PRM(vegRespQ10))));
    const R K_frozFolEff = (R)PRM(frozenSoilFolREff);
    const R K_bvr = (R)PRM(baseVegResp), K_bsr = (R)PRM(baseSoilResp);
    const R K_bfr = (R)PRM(baseFineRootResp), K_bcr = (R)PRM(baseCoarseRootResp);
    const FastRec* __restrict__ recs = (const FastRec*)planBytes;
    const int lastStep = a.n_steps_total - 1;
    struct TileFields { double tair10, tsoil, tsoil10, tillP1; int bits; };
    auto loadFields = [&](int tileStart) {
      int t = tileStart + (lane & (kFastTile - 1));
      t = t > lastStep ? lastStep : t;
      const FastRec* r = recs + t;
      return TileFields{r->tair10, r->tsoil, r->tsoil10, r->tillP1, r->bitsOps};
    };
    // (narrow record fields, fast_math.h recR: an fp32-mixed batch has the float in the low word)
    auto laneR = [](double v, int l) -> R {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
      if (sizeof(R) == 4) return recR<R>(__hiloint2double(0, lo));
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
      return recR<R>(__hiloint2double(hi, lo));
    };
    R qSoil = 0, gFine = 0, gCoarse = 0;
    bool haveQ = false;
    int fTile = tBegin / kFastTile;
    TileFields cur = loadFields(fTile * kFastTile);
    for (int tileStart = fTile * kFastTile; tileStart < tEnd; tileStart += kFastTile) {
      const TileFields nxt = loadFields(tileStart + kFastTile);
      const int tFirst = tileStart > tBegin ? tileStart : tBegin;
      const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
      for (int t = tFirst; t < tLast; t++) {
        const int j = t - tileStart;
        const R tair10 = laneR(cur.tair10, j), tsoil = laneR(cur.tsoil, j);
        const R tillP1 = laneR(cur.tillP1, j);
        const int bits = __builtin_amdgcn_readlane(cur.bits, j);
        // the slot of step t was last used for step t-2, which C is past once it has posted the
        // leaf area of step t-1
        awaitAtLeast(&seqLai, t - 1);
        if (!Staged && statsHere && t == statNext) statAct(false);
        const R vegQ = fexp2(q10Arg(tair10, K_lgVeg), EC);
        R g1 = K_fol * vegQ;
        g1 = (tsoil < K_frozThr) ? g1 * K_frozFolEff : g1;
        const R g2 = K_bvr * vegQ;
        if (!haveQ || !(bits & FAST_TSOIL_SAME)) {
          const R tsoil10 = laneR(cur.tsoil10, j);
          qSoil = fexp2(q10Arg(tsoil10, K_lgSoil), EC);
          gFine = K_bfr * fexp2(q10Arg(tsoil10, K_lgFine), EC);
          gCoarse = K_bcr * fexp2(q10Arg(tsoil10, K_lgCoarse), EC);
          haveQ = true;
        }
        const R qSoilT = K_bsr * qSoil * tillP1;
        post5(&mailFac[t & 1][0][lane], &seqFac, g1, g2, qSoilT, gFine, gCoarse, t);
      }
      cur = nxt;
    }
    if (!Staged && statsHere) statFinish();
    return;
  }
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

  // =============================================================================================
  // ---- S (NCyc): the soil -- heterotrophic respiration, litter breakdown, methane (sipnet.c:1132-1171,
  // :1201-1214, depeffects.c:23-96), nitrogen.c:15-239 with limitations.c:69-139, the litter / soil
  // carbon and nitrogen pools (sipnet.c:1645-1668, nitrogen.c:210-239).  step_fast.hip's Generic
  // block with the nitrogen-cycle flag set, same conventions: reciprocal C:N ratios, x / (C/N) = x N / C,
  // divisions through v_rcp + Newton.  Of the site record it needs the step length and the event
  // count only.
  if (NCyc && role == 3) {
    COOP_CODE_PHASE(3);
    const R K_bsr = (R)PRM(baseSoilResp);
    const R G_lbr = (R)PRM(litterBreakdownRate), G_flr = (R)PRM(fracLitterRespired);
    const R G_nVol = (R)PRM(nVolatilizationFrac), G_nLeach = (R)PRM(nLeachingFrac);
    const double G_nVolD = PRM(nVolatilizationFrac), G_nLeachD = PRM(nLeachingFrac);
    const R G_iLeafCN = (R)(1.0 / PRM(leafCN)), G_iWoodCN = (R)(1.0 / PRM(woodCN)), G_iFineCN = (R)(1.0 / PRM(fineRootCN));
    const R G_kCN = (R)PRM(kCN), G_nFixMax = (R)PRM(nFixationFracMax), G_halfNFix = (R)PRM(halfNFixationMax);
    const R G_resorb = (R)PRM(leafNResorptionFrac), G_anExp = (R)PRM(anaerobicTransExp);
    const R G_soilCH4 = (R)PRM(soilMethaneRate), G_litCH4 = (R)PRM(litterMethaneRate);
    const R X_iSoilCSat = F_carbonSat ? (R)(1.0 / PRM(soilCSaturation)) : R(0);   // Ext: carbon saturation (share 0 when off)
    double soilC = ST(soilC), litterC = ST(litterC), minN = ST(minN);
    double soilOrgN = ST(soilOrgN), litterN = ST(litterN), storN = ST(plantStorageN);
    double totNee = ST(totNee);
    // Full: the heterotrophic side of updateTrackers() (sipnet.c:1420-1496) and the record columns this wave owns
    double totRh = Full ? ST(totRh) : 0.0, totRtot = Full ? ST(totRtot) : 0.0;
    double yRh = Full ? ST(yearlyRh) : 0.0, yRtot = Full ? ST(yearlyRtot) : 0.0, yNee = Full ? ST(yearlyNee) : 0.0;
    double* __restrict__ recs = Full && a.rec ? a.rec + col : nullptr;
    R* __restrict__ oNee = (R*)(a.nee ? a.nee : a.scratchRow) + col;
    const int64_t ldNee = a.nee ? a.ld : 0;
    // the diagnostics counters (sipnet_batch_enable_diagnostics): this wave has the soil's pools and, at the end of a
    // step, the plant side's totals from wave C -- it runs checkBalance() (balance.c:122-169) and counts its own clamps
    const bool wantDiagS = Full && NP == 1 && a.diag != nullptr;
    int clampWarnS = 0, balanceWarnS = 0;
    double maxDCS = 0.0, maxDNS = 0.0;
    // what C needs of these pools at the start of the first step
    postD(&mailMinN[tBegin & 1][lane], 0, minN);
    postFlag(&seqMinN, tBegin);
    postD(&mailStorN[tBegin & 1][lane], 0, storN);
    postFlag(&seqStorN, tBegin);

    WAIT_DECL()
    for (int tileStart = curTile * kFastTile; tileStart < tEnd; tileStart += kFastTile, curTile++) {
      if (tileStart > tBegin) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // DMA of 16 steps ago; all but the last NEE store
      stageTile(curTile + 1, (curTile + 1) & 1);
      const int tFirst = tileStart > tBegin ? tileStart : tBegin;
      const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
      const unsigned char* recB = lds + (curTile & 1) * kTileBytes +
                                  (int)(tFirst - tileFirst(curTile)) * (int)sizeof(FastRec);
      for (int t = tFirst; t < tLast; t++, recB += sizeof(FastRec)) {
        d2 q0;
        int nEvV;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %2 offset:140\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(nEvV) : "v"(ldsAddr(recB)) : "memory");
        const R len = (R)q0.x, invLen = (R)q0.y;
        const double lenD = q0.x;
        const int nEv = uni(nEvV);

        const R eSoilC = (R)soilC, eLitter = (R)litterC, eMinN = (R)minN, eSoilOrgN = (R)soilOrgN;
        const R eLitterN = (R)litterN, eStorN = (R)storN;
        const double minN0 = minN;   // the value C has been given for this step's limitation test
        // getMassTotals() before the step (balance.c:13-36): this wave's pools, before the step's events
        const double dgSoilC0 = soilC, dgLitterC0 = litterC, dgSoilOrgN0 = soilOrgN, dgLitterN0 = litterN, dgStorN0 = storN;
        // wave L's soil-temperature factors of this step (the tillage-scaled one and the plain one) and
        // wave W's moisture terms, each pair behind its flag, one round trip
        // In the same round trip, looked at but not waited for: C's plant-side block of this step and W's
        // leached share (each behind its own flag) -- when they are there already, the nitrogen block below
        // needs no round trip of its own, and the mineral nitrogen of the next step reaches C before C asks.
        double wMoist, wAnoxic;
        R qSoilT, qSoil;
        double pLeafLitter, pWoodLitter, pFineLoss, pCoarseLoss, pDemand, pReduction, pLeafOnAll, pLeafOn, wLeach;
        int plantSeq, leachSeq;
        {
          WAIT_BEGIN()
          int fSeq, wSeq;
          WAIT_DO {
            // (one statement: the compiler may put instructions of its own between two, and does)
#define SIPNET_S_TAKE(RD_R, OFF_R)                                                                                   \
  asm volatile("ds_read_b32 %0, %17\n\t" RD_R " %1, %18\n\t" RD_R " %2, %18 offset:" OFF_R "\n\t"                   \
               "ds_read_b32 %3, %19\n\tds_read_b64 %4, %20\n\tds_read_b64 %5, %20 offset:512\n\t"                    \
               "ds_read_b32 %6, %21\n\tds_read_b64 %7, %22\n\tds_read_b64 %8, %22 offset:512\n\t"                    \
               "ds_read_b64 %9, %22 offset:1024\n\tds_read_b64 %10, %22 offset:1536\n\t"                             \
               "ds_read_b64 %11, %22 offset:2048\n\tds_read_b64 %12, %22 offset:2560\n\t"                            \
               "ds_read_b64 %13, %22 offset:3072\n\tds_read_b64 %14, %22 offset:3584\n\t"                            \
               "ds_read_b32 %15, %23\n\tds_read_b64 %16, %24\n\ts_waitcnt lgkmcnt(0)"                                \
               : "=&v"(fSeq), "=&v"(qSoilT), "=&v"(qSoil), "=&v"(wSeq), "=&v"(wMoist), "=&v"(wAnoxic),              \
                 "=&v"(plantSeq), "=&v"(pLeafLitter), "=&v"(pWoodLitter), "=&v"(pFineLoss), "=&v"(pCoarseLoss),      \
                 "=&v"(pDemand), "=&v"(pReduction), "=&v"(pLeafOnAll), "=&v"(pLeafOn), "=&v"(leachSeq), "=&v"(wLeach) \
               : "v"(ldsAddr(&seqFac)), "v"(ldsAddr(&mailFac[t & 1][2][lane])), "v"(ldsAddr(&seqWat)),                \
                 "v"(ldsAddr(&mailWat[t & 3][0][lane])), "v"(ldsAddr(&seqPlant)),                                    \
                 "v"(ldsAddr(&mailPlant[t & 1][0][lane])), "v"(ldsAddr(&seqLeach)),                                  \
                 "v"(ldsAddr(&mailWat[t & 3][2][lane]))                                                              \
               : "memory")
            if (sizeof(R) == 8)
              SIPNET_S_TAKE("ds_read_b64", "2048");
            else
              SIPNET_S_TAKE("ds_read_b32", "1024");
#undef SIPNET_S_TAKE
          } WAIT_WHILE(uni(fSeq) < t || uni(wSeq) < t, 23, t);
          WAIT_END(0)
        }
        const R moistEff = (R)wMoist, anoxic = (R)wAnoxic;
        // cn = kCN / (kCN + C/N) = kCN N / (kCN N + C), N floored at TINY (util.c:72-75)
        const R denLitterN = eLitterN < R(kTiny) ? R(kTiny) : eLitterN;
        const R denSoilN = eSoilOrgN < R(kTiny) ? R(kTiny) : eSoilOrgN;
        const R cnSoil = fdiv(G_kCN * denSoilN, G_kCN * denSoilN + eSoilC);
        const R cnLitter = fdiv(G_kCN * denLitterN, G_kCN * denLitterN + eLitter);
        const R rSoil = eSoilC * (qSoilT * moistEff) * cnSoil;
        const R breakdown = eLitter * G_lbr * (qSoilT * (R(1) / K_bsr)) * moistEff * cnLitter;
        const R rLitter = breakdown * G_flr;
        const R litterToSoil = breakdown * (R(1) - G_flr);
        R mMoist = anoxic * anoxic;  // pow(A, anaerobicTransExp) with the usual exponent 2
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(G_anExp != R(2)) != 0, 0)) {
          const bool general = G_anExp != R(2) && (anoxic > R(0) || G_anExp <= R(0));
          mMoist = general ? fpow(anoxic, G_anExp) : (G_anExp != R(2) ? R(0) : mMoist);
        }
        const R soilMethane = G_soilCH4 * eSoilC * qSoil * mMoist;
        const R litterMethane = G_litCH4 * eLitter * qSoil * mMoist;
        const R rHet = (R)(double)(rLitter + rSoil);   // R_h: NEE is formed here, at the end of the step

        // ---- the plants' side of the step (C posts it early in its step) and the nitrogen block
        if (uni(plantSeq) < t) {   // not there yet at the start of the step
          WAIT_BEGIN()
          takeD8(&mailPlant[t & 1][0][lane], &seqPlant, t, pLeafLitter, pWoodLitter, pFineLoss, pCoarseLoss, pDemand,
                 pReduction, pLeafOnAll, pLeafOn);
          WAIT_END(1)
        }
        const R leafLitter = (R)pLeafLitter, woodLitter = (R)pWoodLitter, fineRootLoss = (R)pFineLoss;
        const R coarseRootLoss = (R)pCoarseLoss, reductionN = (R)pReduction, leafOnN = (R)pLeafOn;
        R nDemand = (R)pDemand, evMinN = 0;
        if (__builtin_expect(nEv > 0, 0)) {  // the soil side of this step's events, worked out by C (it has the plants)
          double eLit, eSoil, eMin, eOrg, eLitN, eStor;
          takeD6(&mailEvent[t & 1][0][lane], &seqEvent, t, eLit, eSoil, eMin, eOrg, eLitN, eStor);
          evMinN = (R)eMin;
          litterC += (double)((R)eLit * len);
          soilC += (double)((R)eSoil * len);
          minN += (double)(evMinN * len);
          soilOrgN += (double)((R)eOrg * len);
          litterN += (double)((R)eLitN * len);
          storN += (double)((R)eStor * len);
        }
        // unclaimed storage nitrogen.c:127-134, fixation share nitrogen.c:137-152
        const R unclaimed = rmax0(eStorN - (R)pLeafOnAll * len);
        const R fixDen = G_halfNFix + eMinN;
        const R fixFrac = G_nFixMax * ((fixDen < R(kTiny)) ? R(1) : fdiv(G_halfNFix, fixDen));
        const R leafOffNResorption = G_resorb * leafLitter * G_iLeafCN;
        // pool fluxes, nitrogen.c:45-82: x / (C/N) = x * N / C
        const R iLitterCN = fdiv(denLitterN, eLitter), iSoilCN = fdiv(denSoilN, eSoilC);
        const R litterMin = rLitter * iLitterCN, soilMin = rSoil * iSoilCN;
        const R soilNInputs = litterToSoil * iLitterCN + fineRootLoss * G_iFineCN + coarseRootLoss * G_iWoodCN;
        R nOrgLitter = leafLitter * G_iLeafCN - leafOffNResorption + woodLitter * G_iWoodCN - litterMin -
                       litterToSoil * iLitterCN;
        R nOrgSoil = soilNInputs - soilMin;
        if (Ext) {   // carbon saturation, nitrogen.c:60-82: the saturated share of the soil's inputs stays in the litter
          const R sat = clip01(eSoilC * X_iSoilCSat);
          nOrgLitter += soilNInputs * sat;
          nOrgSoil = soilNInputs * (R(1) - sat) - soilMin;
        }
        const R nMin = litterMin + soilMin;
        // volatilisation nitrogen.c:15-26, leaching nitrogen.c:31-41 (the leached share is wave W's:
        // it has the drainage -- by day only after the photosynthesis hand-over, unless the soil cannot
        // fill up in this step)
        R nVolatilization = G_nVol * eMinN * qSoil * (R(0.05) + R(3.8) * anoxic * (R(1) - anoxic));
        if (uni(leachSeq) < t) {   // (by day, in a soil that may fill up: W knows it only after the photosynthesis hand-over)
          WAIT_BEGIN()
          takeD1(&mailWat[t & 3][2][lane], &seqLeach, t, wLeach);
          WAIT_END(2)
        }
        R nLeaching = eMinN * (R)wLeach * G_nLeach;
        // checkMineralNLimitation, limitations.c:119-129
        {
          const R pool = eMinN + (nMin + evMinN) * len;
          const R loss = (nLeaching + nVolatilization) * len;
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(loss > R(kTiny) && loss > pool) != 0, 0)) {
            const R red = (loss > R(kTiny) && loss > pool) ? fdiv(pool, loss) : R(1);
            nLeaching *= red;
            nVolatilization *= red;
          }
        }
        // checkNitrogenLimitation, limitations.c:69-114: nobody is limited where the cheap test both
        // waves make holds; otherwise C gets the exact supply, scales its creation fluxes and answers
        // with the demand that is left
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!nPlentiful(minN0, (double)qSoil, lenD, G_nVolD, G_nLeachD, pDemand)) != 0, 0)) {
          const R availableMinN = eMinN + (nMin - nVolatilization - nLeaching) * len;
          postD(&mailSupply[0][lane], 0, (double)availableMinN);
          postD(&mailSupply[0][lane], 1, (double)fixFrac);
          postD(&mailSupply[0][lane], 2, (double)unclaimed);
          postFlag(&seqSupply, t);
          double dFinal;
          takeD1(&mailDemand[0][lane], &seqDemand, t, dFinal);
          nDemand = (R)dFinal;
        }
        // fixation and uptake, nitrogen.c:155-168
        const R rem = rmax0(nDemand - unclaimed * invLen);
        const R nFixation = fixFrac * rem, nUptake = (R(1) - fixFrac) * rem;
        // updateNitrogenPools(), nitrogen.c:210-239
        const R storageDemand = nDemand - nUptake - nFixation;
        storN += (double)((leafOffNResorption + reductionN - storageDemand - leafOnN) * len);
        minN += (double)(((nMin - nVolatilization - nLeaching) - nUptake) * len);
        soilOrgN += (double)(nOrgSoil * len);
        litterN += (double)(nOrgLitter * len);
        const double dgMinNPost = minN, dgSoilOrgNPost = soilOrgN, dgLitterNPost = litterN, dgStorNPost = storN;
        if (wantDiagS && minN < 0.0 && fabs(minN) > kEps) clampWarnS++;
        minN = rmax0(minN);   // (plant death, which comes later in the step, does not touch this pool)
        postD(&mailMinN[(t + 1) & 1][lane], 0, minN);
        postFlag(&seqMinN, t + 1);
        // updatePoolsForSoil(), sipnet.c:1645-1668 (litter pool on, no carbon saturation)
        const R soilInputs = coarseRootLoss + fineRootLoss + litterToSoil;
        if (Ext) {   // (the soil carbon the reference looks at here already holds this step's event fluxes)
          const R sat = clip01((R)soilC * X_iSoilCSat);
          litterC += (double)((woodLitter + leafLitter + (soilInputs * sat) - litterToSoil - rLitter - litterMethane) * len);
          soilC += (double)((soilInputs * (R(1) - sat) - rSoil - soilMethane) * len);
        } else {
          litterC += (double)((woodLitter + leafLitter - litterToSoil - rLitter - litterMethane) * len);
          soilC += (double)((soilInputs - rSoil - soilMethane) * len);
        }
        const double dgSoilCPost = soilC, dgLitterCPost = litterC;

        // the end of C's step: its mortality verdict (one word per lane) and, where a stand died, what its
        // biomass adds to these pools (sipnet.c:1688-1767); then ensureNonNegativeStocks() for them
        int w;
        double pend;   // GPP - R_a of this step (C posts it right before the verdict word)
        WAIT_DO {
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w), "=&v"(pend)
                       : "v"(ldsAddr(&mailAlive[(t + 1) & 1][lane])), "v"(ldsAddr(&mailPend[t & 1][0][lane])) : "memory");
        } WAIT_WHILE(uni(w < 0 ? -w : w) < t + 3, 24, t);
        double pendRa = 0.0, pendRRoot = 0.0;   // Full: R_a and the root respiration of the step (posted with GPP - R_a)
        int bitsS = 0;
        if (Full) {
          asm volatile("ds_read_b64 %0, %3 offset:512\n\tds_read_b64 %1, %3 offset:1024\n\tds_read_b32 %2, %4 offset:128\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(pendRa), "=&v"(pendRRoot), "=&v"(bitsS)
                       : "v"(ldsAddr(&mailPend[t & 1][0][lane])), "v"(ldsAddr(recB)) : "memory");
          bitsS = uni(bitsS);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(w < 0) != 0, 0)) {
          double d0, d1, d2, d3;
          asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\t"
                       "ds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ldsAddr(&mailDeath[0][lane])) : "memory");
          if (w < 0) {
            soilC += d0;
            litterC += d1;
            soilOrgN += d2;
            litterN += d3 + storN;
            storN = 0.0;
          }
        }
        if (wantDiagS)   // ensureNonNegativeStocks()' warnings, sipnet.c:1346-1356, for the pools this wave owns
          clampWarnS += (soilC < 0.0 && fabs(soilC) > kEps) + (litterC < 0.0 && fabs(litterC) > kEps) +
                        (soilOrgN < 0.0 && fabs(soilOrgN) > kEps) + (litterN < 0.0 && fabs(litterN) > kEps) +
                        (storN < 0.0 && fabs(storN) > kEps);
        soilC = rmax0(soilC);
        litterC = rmax0(litterC);
        soilOrgN = rmax0(soilOrgN);
        litterN = rmax0(litterN);
        storN = rmax0(storN);
        if (kPendRows > 3 && wantDiagS) {   // updateBalanceTrackerPostClamp() + checkBalance(), balance.c:40-169
          double pc0, pn0, pc1, pn1, pc2, pn2, dInC, plantOut, evOutC, evInN, evOutN;
          const unsigned base = ldsAddr(&mailPend[t & 1][0][lane]);   // rows 3 .. 13, behind the verdict word like rows 0 .. 2
          asm volatile("ds_read_b64 %0, %11 offset:1536\n\tds_read_b64 %1, %11 offset:2048\n\tds_read_b64 %2, %11 offset:2560\n\t"
                       "ds_read_b64 %3, %11 offset:3072\n\tds_read_b64 %4, %11 offset:3584\n\tds_read_b64 %5, %11 offset:4096\n\t"
                       "ds_read_b64 %6, %11 offset:4608\n\tds_read_b64 %7, %11 offset:5120\n\tds_read_b64 %8, %11 offset:5632\n\t"
                       "ds_read_b64 %9, %11 offset:6144\n\tds_read_b64 %10, %11 offset:6656\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(pc0), "=&v"(pn0), "=&v"(pc1), "=&v"(pn1), "=&v"(pc2), "=&v"(pn2), "=&v"(dInC), "=&v"(plantOut),
                         "=&v"(evOutC), "=&v"(evInN), "=&v"(evOutN)
                       : "v"(base) : "memory");
          // mass totals in the reference's order of summation: plants (wave C's partial sum), soil, litter | plants,
          // soil organic, litter, mineral, storage
          const double preC = (pc0 + dgSoilC0) + dgLitterC0, postC = (pc1 + dgSoilCPost) + dgLitterCPost;
          const double finC = (pc2 + soilC) + litterC;
          const double preN = (((pn0 + dgSoilOrgN0) + dgLitterN0) + minN0) + dgStorN0;
          const double postN = (((pn1 + dgSoilOrgNPost) + dgLitterNPost) + dgMinNPost) + dgStorNPost;
          const double finN = (((pn2 + soilOrgN) + litterN) + minN) + storN;
          double clampedC = finC - postC, clampedN = finN - postN;
          if (clampedC < kEps) clampedC = 0.0;
          if (clampedN < kEps) clampedN = 0.0;
          double outC = plantOut + (double)rSoil + (double)soilMethane + evOutC;
          outC += (double)rLitter + (double)litterMethane;
          const double inC = dInC * (double)len + clampedC;
          outC *= (double)len;
          const double inN = ((double)nFixation + evInN) * (double)len + clampedN;
          const double outN = ((double)nLeaching + (double)nVolatilization + evOutN) * (double)len;
          const double dC = (finC - preC) - (inC - outC);
          const double dN = (finN - preN) + (outN - inN);
          maxDCS = fmax(maxDCS, fabs(dC));
          maxDNS = fmax(maxDNS, fabs(dN));
          if (!(fabs(dC) < kEps)) balanceWarnS++;
          if (!(fabs(dN) < kEps)) balanceWarnS++;
        }
        postD(&mailStorN[(t + 1) & 1][lane], 0, storN);
        postFlag(&seqStorN, t + 1);
        {  // NEE = -(NPP - R_h), sipnet.c:1433-1450: GPP - R_a from C, R_h = (litter + soil respiration) here
#pragma clang fp contract(off)
          const R tRh = rHet * len;
          const R tNee = R(-1.0) * ((R)pend - tRh);
          totNee += (double)tNee;
          *oNee = tNee;
          oNee += ldNee;
          if (Full) {
            if (bitsS & FAST_TRACK_NEW_YEAR) yRh = yRtot = yNee = 0.0;
            const R tRa = (R)pendRa, tRRoot = (R)pendRRoot;
            const R tRtot = tRa + tRh;
            yRh += (double)tRh;
            yRtot += (double)tRtot;
            yNee += (double)tNee;
            totRh += (double)tRh;
            totRtot += (double)tRtot;
            if (recs) {
              double* __restrict__ r = recs;
              const int64_t L = a.ld;
              r[0 * L] = (double)tNee;
              r[3 * L] = totNee;
              r[6 * L] = (double)(tRRoot + tRh);
              r[9 * L] = (double)tRh;
              r[10 * L] = (double)tRtot;
              r[16 * L] = soilC;
              r[18 * L] = litterC;
              r[22 * L] = minN;
              r[23 * L] = soilOrgN;
              r[24 * L] = litterN;
              r[25 * L] = storN;
              r[27 * L] = (double)(nVolatilization * len);
              r[28 * L] = (double)(nLeaching * len);
              r[29 * L] = (double)(nFixation * len);
              r[30 * L] = (double)(nUptake * len);
              r[31 * L] = (double)((soilMethane + litterMethane) * len);
              recs += (int64_t)SIPNET_NREC * L;
            }
          }
        }
      }
    }
    WAIT_STORE(12)
    if (act) {
      ST(soilC) = soilC;
      ST(litterC) = litterC;
      ST(minN) = minN;
      ST(soilOrgN) = soilOrgN;
      ST(litterN) = litterN;
      ST(plantStorageN) = storN;
      ST(totNee) = totNee;
      if (Full) {
        ST(totRh) = totRh;
        ST(totRtot) = totRtot;
        ST(yearlyRh) = yRh;
        ST(yearlyRtot) = yRtot;
        ST(yearlyNee) = yNee;
      }
      if (wantDiagS) {
        double* __restrict__ dg = a.diag + col;
        if (clampWarnS) atomicAdd(dg, (double)clampWarnS);   // (waves C and W add their pools' counts)
        dg[1 * nc] += (double)balanceWarnS;
        dg[2 * nc] = fmax(dg[2 * nc], maxDCS);
        dg[3 * nc] = fmax(dg[3 * nc], maxDNS);
      }
    }
    return;
  }

  // =============================================================================================
  if (role == 2) {
    // Every layout of this wave must produce the same bits (they are tested against each other),
    // and what fuses into an FMA under -ffp-contract=fast depends on which neighbouring
    // expressions share a product (the factor block is here in some layouts and on wave F in
    // others): so no implicit fusion in this wave, constants included; the FMAs that matter are
    // written out below
#pragma clang fp contract(off)
    COOP_CODE_PHASE(2);
    // ---- L: potPsn() + calcLightEff(), sipnet.c:517-641 -------------------------------------
    const double leafCSpWt = PRM(leafCSpWt);
    const double convK = kCWeight * (1.0 / kTen9) * (leafCSpWt / PRM(cFracLeaf)) * kSecPerDay;
    const double respPerGram = PRM(baseFolRespFrac) * PRM(aMax);
    const R K_g = (R)((PRM(aMax) * PRM(aMaxFrac) + respPerGram) * convK);
    const R K_tmin = (R)PRM(psnTMin), K_tmax = (R)PRM(psnTMax);
    const R K_invDen = (R)(1.0 / (((PRM(psnTMax) - PRM(psnTMin)) / 2.0) * ((PRM(psnTMax) - PRM(psnTMin)) / 2.0)));
    const R K_slope = (R)PRM(dVpdSlope), K_vexp = (R)PRM(dVpdExp);
    const R K_attl = (R)(-PRM(attenuation) * (1.0 / 6.0) * kLog2e);
    const R K_invHalf = (R)(1.0 / PRM(halfSatPar));
    // for wave C's respiration terms: everything that depends on climate and parameters only
    const R K_frozThr = (R)PRM(frozenSoilThreshold);
    const R K_lgVeg = (R)log2(PRM(vegRespQ10)), K_lgSoil = (R)log2(PRM(soilRespQ10));
    const R K_lgFine = (R)log2(PRM(fineRootQ10)), K_lgCoarse = (R)log2(PRM(coarseRootQ10));
    const R K_fol = (R)((PRM(baseFolRespFrac) * PRM(aMax)) *
                        (kCWeight * (1.0 / kTen9) * (PRM(leafCSpWt) / PRM(cFracLeaf)) * kSecPerDay) *
                        (1.0 / PRM(leafCSpWt)) * exp2(-(PRM(psnTOpt) / 10.0) * log2(PRM(vegRespQ10))));
    const R K_frozFolEff = (R)PRM(frozenSoilFolREff);
    const R K_bvr = (R)PRM(baseVegResp), K_bsr = (R)PRM(baseSoilResp);
    const R K_bfr = (R)PRM(baseFineRootResp), K_bcr = (R)PRM(baseCoarseRootResp);
    R qSoil = 0, gFine = 0, gCoarse = 0;
    bool haveQ = false;
    WAIT_DECL()
    // Every parameter load above has landed before the loop starts, and the compiler is told so (a
    // wait it can see): its wait-count bookkeeping would otherwise put a full `s_waitcnt vmcnt(0)` in
    // front of the first in-loop use of each of these constants -- in the middle of the day step, where
    // it would sit out the statistics loads (and the tile DMA) on every iteration
    __builtin_amdgcn_s_waitcnt(0);
    for (int tileStart = curTile * kFastTile; tileStart < tEnd; tileStart += kFastTile, curTile++) {
      if (tileStart > tBegin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMA of 16 steps ago
      stageTile(curTile + 1, (curTile + 1) & 1);
      const int tFirst = tileStart > tBegin ? tileStart : tBegin;
      const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
      const unsigned char* recB = lds + (curTile & 1) * kTileBytes +
                                  (int)(tFirst - tileFirst(curTile)) * (int)sizeof(FastRec);
      for (int t = tFirst; t < tLast; t++, recB += sizeof(FastRec)) {
        d2 q1, q2, q3, q5;
        double q6x;
        int bitsV;
        asm volatile("ds_read_b32 %0, %6 offset:128\n\tds_read_b128 %1, %6 offset:16\n\t"
                     "ds_read_b128 %2, %6 offset:32\n\tds_read_b128 %3, %6 offset:48\n\t"
                     "ds_read_b128 %4, %6 offset:80\n\tds_read_b64 %5, %6 offset:96\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bitsV), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q5), "=&v"(q6x)
                     : "v"(ldsAddr(recB)) : "memory");
        const int bits = uni(bitsV);
#ifdef SIPNET_COOP_BOUNDED
        // the error path's own test (SIPNET_KOPT_WAIT_SELFTEST): this wave stops posting after 100 steps, the others'
        // waits must give up and the launch must end with the report
        if ((a.options & SIPNET_KOPT_WAIT_SELFTEST) && t == tBegin + 100) return;
#endif
        if (!Staged && statsHere && t == statNext) statAct(false);
        // ---- for wave C: the climate / parameter part of its respiration terms of THIS step
        // (vegResp sipnet.c:1051-1068, calcRootResp :1073, calcSoilRespiration :1132-1148 with
        // depeffects.c:71-74):  folResp = leafC * g1,  rVeg = folResp + totalWoodC * g2,
        // rSoil = soilC * (qSoilT * moistEff[wave W]),  rFineRoot = fineRootC * gFine,
        // rCoarseRoot = coarseRootC * gCoarse.  Nothing here depends on member state, so this
        // wave -- idle at night and while it waits for the leaf area by day -- runs it ahead of
        // C: the slot of step t was last used for step t-2, which C is past once it has posted
        // the leaf area of step t-1
        if (!FacWave) {  // (wave F's job when there is one)
          WAIT_BEGIN()
          awaitAtLeast(&seqLai, t - 1);
          // NCyc: wave S reads two rows of the block too, at the start of ITS step -- which C's progress
          // does not vouch for (C(t-1) only needs S's nitrogen block of step t-2 done): the slot is free
          // once S has posted the mineral nitrogen of step t-1, which it does after that read (found by the fuzzer:
          // one trial in 600 had S take the soil factors of step t+2 for step t, 2e-5 off on NEE)
          if (NCyc) awaitAtLeast(&seqMinN, t - 1);
          WAIT_END(1)
          const R vegQ = fexp2(q10Arg(recR<R>(q5.y), K_lgVeg), EC);
          R g1 = K_fol * vegQ;
          g1 = (recR<R>(q1.y) < K_frozThr) ? g1 * K_frozFolEff : g1;
          const R g2 = K_bvr * vegQ;
          if (!haveQ || !(bits & FAST_TSOIL_SAME)) {
            const R tsoil10 = recR<R>(q6x);
            qSoil = fexp2(q10Arg(tsoil10, K_lgSoil), EC);
            gFine = K_bfr * fexp2(q10Arg(tsoil10, K_lgFine), EC);
            gCoarse = K_bcr * fexp2(q10Arg(tsoil10, K_lgCoarse), EC);
            haveQ = true;
          }
          const R qSoilT = K_bsr * qSoil * recR<R>(q3.x);
          if (NCyc) postRaw(&mailFac[t & 1][6][lane], qSoil);   // before the flag post5 sets
          post5(&mailFac[t & 1][0][lane], &seqFac, g1, g2, qSoilT, gFine, gCoarse, t);
        }
        // (staged statistics: after the factor post, so that C has what it waits for.  C and W cannot be back
        // at the half's rows before this wave is done with them: they are kStageR - 1 steps away from it when
        // the action starts, and by day they need this wave's potential photosynthesis to go on)
        // (round 5 moved this to the factor wave of the two-chunk layout -- it shares its SIMD with a water wave, the
        // light wave with the other chunk's carbon wave -- and measured nothing: c4 run_stats 12.0 -> 11.9 ms.  What the
        // staged statistics cost there, 1.7 ms, is C's and W's own stage writes, not where the half is summed.)
        if (Staged && stageOn && t == statNext) stagedAct();
        if (!(bits & FAST_PAR_POS)) continue;  // night: potGrossPsn = 0, nobody waits for it
        {
        const R tair = recR<R>(q1.x);
        // climate-only factors first, then the leaf area of this step
        const R dTemp = rmax0((K_tmax - tair) * (tair - K_tmin) * K_invDen);
        R vpdPow = recR<R>(q2.y) * recR<R>(q2.y);
        if (!PlainExp)
          vpdPow = (K_vexp == R(2)) ? vpdPow : fexp2(K_vexp * recR<R>(((const double*)(recB + 144))[2]), EC);
        const R dVpd = rmax0(ffma(-K_slope, vpdPow, R(1)));
        const R q = recR<R>(q2.x) * K_invHalf;
        const R e0 = fexp2(q, EC);
        WAIT_BEGIN()
        const R lai = take(&mailLai[t & 1][lane], &seqLai, t);
        WAIT_END(0)
        const R r1 = fexp2(K_attl * lai, EC);
        const R r2 = r1 * r1, r3 = r2 * r1, r4 = r2 * r2, r5 = r4 * r1, r6 = r3 * r3;
        const R e1 = fexp2(q * r1, EC), e2 = fexp2(q * r2, EC), e3 = fexp2(q * r3, EC);
        const R e4 = fexp2(q * r4, EC), e5 = fexp2(q * r5, EC), e6 = fexp2(q * r6, EC);
        const R s = ffma(R(2), e2 + e4, ffma(R(4), (e1 + e3) + e5, e0 + e6));
        const R dLight = ffma(-s, R(1.0 / 18.0), R(1));
        post(&mailPgp[t & 1][lane], &seqPgp, K_g * lai * dTemp * dVpd * dLight, t);
        }
      }
    }
    WAIT_STORE(0)
    if (Staged && stageOn) stagedFinish();
    else if (statsHere) statFinish();
    return;
  }

  // =============================================================================================
  if (role == 1) {
    COOP_CODE_PHASE(1);
    // ---- W: moisture(), calcPrecip(), snowPack(), calcSoilWaterFluxes(), sipnet.c:656-1031 ----
    const R K_tr = (R)(1000.0 * (44.0 / 12.0) * (1.0 / 10000.0) / PRM(wueConst));
    const R K_whc = (R)PRM(soilWHC), K_invWhc = (R)(1.0 / PRM(soilWHC));
    const R K_wrf = (R)PRM(waterRemoveFrac);
    const R K_frozThr = (R)PRM(frozenSoilThreshold), K_frozEff = (R)PRM(frozenSoilEff);
    const R K_immed = (R)PRM(immedEvapFrac), K_ff = (R)PRM(fastFlowFrac);
    const R K_invRd = (R)(1.0 / PRM(rdConst)), K_rd = (R)PRM(rdConst), K_melt = (R)PRM(snowMelt);
    const R K_c1l = (R)(PRM(rSoilConst1) * kLog2e), K_c2l = (R)(PRM(rSoilConst2) * kLog2e);
    const R K_moistExp = (R)PRM(soilRespMoistEffect);
    double soilWater = ST(soilWater), snow = ST(snow);
    double totGpp = ST(totGpp);  // GPP is this wave's own product: it stores the plane and keeps the total
    // NCyc: what wave S needs of the soil water -- the anaerobic moisture effect and the anoxic share
    // (depeffects.c:46-57, :89-96) at the start of a step, the leached share of the mineral nitrogen
    // (nitrogen.c:31-41) once the drainage is known
    const R G_fAnox = NCyc ? (R)PRM(fAnoxia) : R(0), G_iFAnox = NCyc ? (R)(1.0 / PRM(fAnoxia)) : R(0);
    const R G_iOneMinusAnox = NCyc ? (R)(1.0 / (1.0 - PRM(fAnoxia))) : R(0);
    const R G_anDecomp = NCyc ? (R)PRM(anaerobicDecompRate) : R(0);
    // Ext: leaf-water interception, flooding; Opt: the anaerobic moisture effect and the methane moisture term
    const R X_leafPool = Ext ? (R)PRM(leafPoolDepth) : R(0);
    const R X_drainFrac = F_flooding ? (R)PRM(waterDrainFrac) : R(kNoCap);
    const R X_fAnox = Opt ? (R)PRM(fAnoxia) : R(0), X_iFAnox = Opt ? (R)(1.0 / PRM(fAnoxia)) : R(0);
    const R X_iOneMinusAnox = Opt ? (R)(1.0 / (1.0 - PRM(fAnoxia))) : R(0), X_anDecomp = Opt ? (R)PRM(anaerobicDecompRate) : R(0);
    const R X_anExp = Opt ? (R)PRM(anaerobicTransExp) : R(0), X_iBsr = Opt ? (R)(1.0 / PRM(baseSoilResp)) : R(0);
    R* __restrict__ oEt = (R*)(a.et ? a.et : a.scratchRow) + col;
    R* __restrict__ oGpp = (R*)(a.gpp ? a.gpp : a.scratchRow) + col;
    const int64_t ldEt = a.et ? a.ld : 0, ldGpp = a.gpp ? a.ld : 0;
    CoopSums<Sums> sumsW;   // Sums: the group's sums so far, and the steps it still lacks
    if constexpr (Sums) sumsW.left = a.sumEvery;
    const bool wantDiagW = Full && a.diag != nullptr;
    const double K_whc2 = Full ? 2.0 * PRM(soilWHC) : 0.0;
    double* __restrict__ recw = Full && a.rec ? a.rec + col : nullptr;
    int clampWarnW = 0;
    WAIT_DECL()

    for (int tileStart = curTile * kFastTile; tileStart < tEnd; tileStart += kFastTile, curTile++) {
      // the DMA of 16 steps ago has landed: all but the two youngest operations (the last step's
      // ET and GPP stores)
      if constexpr (Sums) {
        if (tileStart > tBegin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no store per step to leave in flight)
      } else {
        if (tileStart > tBegin) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      stageTile(curTile + 1, (curTile + 1) & 1);
      const int tFirst = tileStart > tBegin ? tileStart : tBegin;
      const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
      const unsigned char* recB = lds + (curTile & 1) * kTileBytes +
                                  (int)(tFirst - tileFirst(curTile)) * (int)sizeof(FastRec);
      for (int t = tFirst; t < tLast; t++, recB += sizeof(FastRec)) {
        // Explicit fused multiply-adds only, as on the carbon wave: what the compiler chose to fuse differed
        // between the layouts' instantiations (fp32: the last sublimation of a snow pack, one ulp of ET).
#pragma clang fp contract(off)
        MARK("W step begin")
        // only the fields this wave uses (80 of the record's 144 hot bytes: a lone wave pays LDS
        // reads by the byte): len invLen | tair tsoil | vpd | rainRate | sublW evapNum | invWspd | bits | evCount
        d2 q0, q1, q2, q3, q4, q5;
        i4 j0;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:40\n\t"
                     "ds_read_b64 %3, %8 offset:56\n\tds_read_b128 %4, %8 offset:64\n\t"
                     "ds_read_b64 %5, %8 offset:80\n\tds_read_b32 %6, %8 offset:128\n\t"
                     "ds_read_b32 %7, %8 offset:140\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2.y), "=&v"(q3.y), "=&v"(q4), "=&v"(q5.x), "=&v"(j0.x), "=&v"(j0.w)
                     : "v"(ldsAddr(recB)) : "memory");
        const int32_t* rareI = (const int32_t*)(recB + 184);
        const R len = (R)q0.x, invLen = (R)q0.y, tair = recR<R>(q1.x), tsoil = recR<R>(q1.y);
        const int bits = uni(j0.x);
        const int nEv = uni(j0.w);
        const R eWater = (R)soilWater, eSnow = (R)snow;
        const double oldSoilWater = soilWater;  // before this step's irrigation, sipnet.c:1470
        const bool frozen = tsoil < K_frozThr;

        // ---- for wave C: the soil-moisture effect on heterotrophic respiration of THIS step
        // (depeffects.c:23-57; the Q10 / tillage part comes from wave F or L), posted before anything
        // else so that C never waits for it
        if (!NCyc) {
          R moistEff = clip01(eWater * K_invWhc);
          if (!PlainExp && __builtin_amdgcn_ballot_w64(K_moistExp != R(1)) != 0)  // pow only where some member needs it
            moistEff = (K_moistExp == R(1)) ? moistEff : fpow(moistEff, K_moistExp);
          R mK = 0;   // Opt: the methane moisture term over baseSoilResp x (1 + tillage) (C multiplies by qSoilT)
          if (Opt && F_anaerobic) {   // depeffects.c:46-57, :89-96 (replaces the aerobic form; the pow above then ran idle)
            const R fWhc = clip01(eWater * K_invWhc);
            const R anoxic = clip01((fWhc - X_fAnox) * X_iOneMinusAnox);
            moistEff = ffma(X_anDecomp, anoxic, (R(1) - anoxic) * clip01(fWhc * X_iFAnox));
            R mMoist = anoxic * anoxic;  // pow(A, anaerobicTransExp) with the usual exponent 2
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(X_anExp != R(2)) != 0, 0)) {
              const bool general = X_anExp != R(2) && (anoxic > R(0) || X_anExp <= R(0));
              mMoist = general ? fpow(anoxic, X_anExp) : (X_anExp != R(2) ? R(0) : mMoist);
            }
            mK = mMoist * X_iBsr;
            if (__builtin_expect(bits & FAST_HAS_TILL, 0)) mK = fdiv(mK, recR<R>(((const double*)recB)[6]));   // FastRec::tillP1
          }
          moistEff = (bits & FAST_TSOIL_NEG) ? R(1) : moistEff;
          if (Opt) post2(&mailFac[t & 1][5][lane], &seqMoist, moistEff, mK, t);
          else post(&mailFac[t & 1][5][lane], &seqMoist, moistEff, t);
        }
        if (NCyc) {
          const R fWhc = clip01(eWater * K_invWhc);
          const R anoxic = clip01((fWhc - G_fAnox) * G_iOneMinusAnox);
          R moistEff = ffma(G_anDecomp, anoxic, (R(1) - anoxic) * clip01(fWhc * G_iFAnox));
          moistEff = (bits & FAST_TSOIL_NEG) ? R(1) : moistEff;
          postD(&mailWat[t & 3][0][lane], 0, (double)moistEff);
          postD(&mailWat[t & 3][0][lane], 1, (double)anoxic);
          postFlag(&seqWat, t);
        }

        // everything that does not need the light block first
        const bool tairPos = (bits & FAST_TAIR_POS) != 0;
        const R rate = recR<R>(q3.y);
        const R rain = tairPos ? rate : R(0), snowFall = tairPos ? R(0) : rate;
        // calcPrecip(), sipnet.c:848-882.  Ext: with the leaf-water flag the immediate evaporation is capped by what
        // the canopy holds, lai(t) x leafPoolDepth -- asked for on steps with rain only (the site's: wave-uniform);
        // lai(t) is C's post of the step before, void (0) for a member that died in it
        R immedEvap = rain * K_immed;
        if (Ext && F_leafWater && __builtin_amdgcn_ballot_w64(rain > R(0)) != 0) {
          R laiNow;
          bool diedBefore;
          takePgp(&mailLai[t & 1][lane], &seqLai, &mailAlive[t & 1][lane], t, laiNow, diedBefore);
          immedEvap = rminv(immedEvap, diedBefore ? R(0) : laiNow * X_leafPool);
        }
        const R netRain = Ext ? rain - immedEvap : ffma(-rain, K_immed, rain);   // (beside the product, not behind it)
        R snowMelt = 0, sublimation = 0, evaporationPot = 0;
        const bool hasSnow = eSnow > R(0);
        if (hasSnow) {
          R subl = rmax0(recR<R>(q4.x) * K_invRd);
          R remaining = ffma(snowFall, len, eSnow);
          const R afterSubl = ffma(-subl, len, remaining);
          const bool allGone = afterSubl < R(0);
          subl = allGone ? remaining * invLen : subl;
          remaining = allGone ? R(0) : afterSubl;
          R melt = tairPos ? K_melt * tair : R(0);
          melt = (tairPos && (ffma(-melt, len, remaining) < R(0))) ? remaining * invLen : melt;
          sublimation = subl;
          snowMelt = melt;
        } else {
          const R wf = clip01(eWater * K_invWhc);
          const R rsoil = fexp2(ffma(-K_c2l, wf, K_c1l), EC);
          evaporationPot = rmax0(fdiv(recR<R>(q4.y), ffma(K_rd, recR<R>(q5.x), rsoil)));
        }
        R removable = rminv(eWater, K_whc) * K_wrf;
        removable = frozen ? removable * K_frozEff : removable;

        // NCyc: the leached share needs the drainage, which by day is known only after the photosynthesis
        // hand-over -- unless the soil cannot reach its holding capacity in this step whatever the plants
        // take (the common case, decided for the wavefront): then it is zero and wave S gets it now
        bool leachPosted = false;
        if (NCyc) {
          R netIn0 = netRain + snowMelt;
          netIn0 = ffma(-netIn0, K_ff, netIn0);
          leachPosted = __builtin_amdgcn_ballot_w64(!(ffma(netIn0, len, eWater) <= K_whc)) == 0;
          if (leachPosted) {
            postD(&mailWat[t & 3][0][lane], 2, 0.0);
            postFlag(&seqLeach, t);
          }
        }

        // moisture(), sipnet.c:656-699, with the potential photosynthesis of wave L
        R transpiration = 0, photosynthesis = 0;
        if (bits & FAST_PAR_POS) {
          R pgpSpec;
          bool diedBefore;
          WAIT_BEGIN()
          takePgp(&mailPgp[t & 1][lane], &seqPgp, &mailAlive[t & 1][lane], t, pgpSpec, diedBefore);
          WAIT_END(0)
          const R potGrossPsn = diedBefore ? R(0) : pgpSpec;
          const R potTrans = potGrossPsn * recR<R>(q2.y) * K_tr;
          const bool hasPsn = potGrossPsn >= R(kTiny);
          const bool limited = hasPsn && removable < potTrans;
          transpiration = hasPsn ? (limited ? removable : potTrans) : R(0);
          photosynthesis = potGrossPsn;
          // (the division only where the water limits some member of the wavefront: this wave's day step is
          // as long as the carbon wave's, and the carbon wave waits for what is posted next)
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(limited) != 0, 0))
            photosynthesis = limited ? potGrossPsn * fdiv(removable, potTrans) : potGrossPsn;
          post(&mailPsn[t & 1][lane], &seqPsn, photosynthesis, t);
        }
        const R tGpp = photosynthesis * len;
        totGpp += (double)tGpp;   // the total is the sum of the ROUNDED per-step values (sipnet.c:1433-1450)

        R evaporation, drainage, wetting;   // wetting: rain + melt - immediate evaporation - fast flow
        {
          R netIn = netRain + snowMelt;
          wetting = Ext ? ffma(-netIn, K_ff, (rain + snowMelt) - immedEvap) : ffma(-netIn, K_ff, ffma(-rain, K_immed, rain + snowMelt));
          netIn = ffma(-netIn, K_ff, netIn);
          R remaining = ffma(-transpiration, len, ffma(netIn, len, eWater));
          const R afterEvap = ffma(-evaporationPot, len, remaining);
          const bool dryOut = !hasSnow && (afterEvap < R(kTiny));
          evaporation = dryOut ? (remaining - R(kTiny)) * invLen : evaporationPot;
          remaining = hasSnow ? remaining : (dryOut ? R(0) : afterEvap);
          drainage = remaining > K_whc ? (remaining - K_whc) * invLen : R(0);
          if (Ext) {   // flooding, sipnet.c:1019-1027 (no cap with the flag off)
            const R excess = remaining - K_whc;
            drainage = remaining > K_whc ? rminv(excess * X_drainFrac, excess * invLen) : R(0);
          }
        }
        // irrigation, events.c:484-543
        R evEvap = 0;
        if (__builtin_expect(nEv > 0, 0)) {
          R evSoilWater = 0;
          const int ev0 = uni(rareI[3]);
          for (int k = 0; k < nEv; k++) {
            const EvRec& ev = a.events[evBase + ev0 + k];
            if (uni(ev.type) == SIPNET_EV_IRRIG) {
              const R p0 = (R)ev.p[0];
              const R evapAmount = ((int)ev.p[1] == 0) ? K_immed * p0 : R(0);
              evEvap = ffma(evapAmount, invLen, evEvap);
              evSoilWater = ffma(p0 - evapAmount, invLen, evSoilWater);
            }
          }
          accum(soilWater, evSoilWater, len);
        }
        if (NCyc && !leachPosted) {
          postD(&mailWat[t & 3][0][lane], 2, (double)rminv(drainage * K_invWhc, R(1)));
          postFlag(&seqLeach, t);
        }
        accum(soilWater, wetting - evaporation - transpiration - drainage, len);
        accum(snow, snowFall - snowMelt - sublimation, len);
        if (wantDiagW) {  // clamp warnings of the two water pools, sipnet.c:1346-1356
          if (soilWater < 0.0 && fabs(soilWater) > kEps) clampWarnW++;
          if (snow < kTiny && fabs(snow) > kEps) clampWarnW++;
        }
        soilWater = rmax0(soilWater);
        snow = snow < kTiny ? 0.0 : snow;

        // never run more than one step ahead of C (the mailboxes have two slots): C is past the
        // pools of step t-1 once it has posted lai(t).  By day that is implied: this wave has taken
        // pgp(t), which wave L computed from lai(t)
        if (!(bits & FAST_PAR_POS)) {
          WAIT_BEGIN()
          awaitAtLeast(&seqLai, t);
          WAIT_END(1)
        }

        const R tEt = ((Ext ? transpiration + immedEvap : ffma(rain, K_immed, transpiration)) + evaporation + sublimation + evEvap) * len;
        if constexpr (Sums) {
          sumsW.et += (double)tEt;
          sumsW.gpp += (double)tGpp;
          if (--sumsW.left == 0) {
            *oEt = (R)sumsW.et;
            *oGpp = (R)sumsW.gpp;
            oEt += ldEt;
            oGpp += ldGpp;
            sumsW.et = sumsW.gpp = 0.0;
            sumsW.left = a.sumEvery;
          }
        } else {
          *oEt = tEt;
          *oGpp = tGpp;
          oEt += ldEt;
          oGpp += ldGpp;
        }
        if (__builtin_expect(stageOn, 0)) {   // sipnet_batch_run_stats on a two- / four-chunk layout: for wave L
          postRaw(&stage[1][(t - tBegin) & (2 * kStageR - 1)][lane], tGpp);
          postRaw(&stage[2][(t - tBegin) & (2 * kStageR - 1)][lane], tEt);
        }
        MARK("W step end")
        if (Full && recw) {
          const int64_t L = a.ld;
          recw[1 * L] = (double)tGpp;
          recw[35 * L] = totGpp;
          recw[2 * L] = (double)tEt;
          recw[12 * L] = (oldSoilWater + soilWater) / K_whc2;
          recw[13 * L] = (double)transpiration;
          recw[17 * L] = soilWater;
          recw[19 * L] = snow;
          recw += (int64_t)SIPNET_NREC * L;
        }
      }
    }
    if constexpr (Sums) {
      if (sumsW.left != a.sumEvery) {   // the launch's last, shorter group
        *oEt = (R)sumsW.et;
        *oGpp = (R)sumsW.gpp;
      }
    }
    WAIT_STORE(4)
    if (a.statsPart) {  // every ET / GPP store of the launch has reached L2: wave L may sum the last tiles
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("ds_write_b32 %0, %1" :: "v"(ldsAddr(&seqDone[1])), "v"(1) : "memory");
    }
    if (act) {
      ST(soilWater) = soilWater;
      ST(snow) = snow;
      ST(totGpp) = totGpp;

      if (wantDiagW && clampWarnW) atomicAdd(a.diag + col, (double)clampWarnW);
    }
    return;
  }

  // =============================================================================================
  // ---- C: carbon fluxes, pools, trackers, running mean (sipnet.c:756-842, 1051-1196, 1420-1806)
  COOP_CODE_PHASE(0);
  const R K_invLcsw = (R)(1.0 / PRM(leafCSpWt));
  const R K_wtr = (R)PRM(woodTurnoverRate), K_ltr = (R)PRM(leafTurnoverRate);
  const R K_frt = (R)PRM(fineRootTurnoverRate), K_crt = (R)PRM(coarseRootTurnoverRate);
  const R K_la = (R)PRM(leafAllocation), K_wa = (R)PRM(woodAllocation);
  const R K_fa = (R)PRM(fineRootAllocation), K_ca = (R)PRM(coarseRootAllocation);
  // leaf-on threshold of the variable the plan put into the record: year-to-date GDD -- or, with the gdd flag
  // off, soil temperature / day of year, whose threshold convertParamsKernel has then put into this row
  // (choosing among the three parameters here, by flag or by a row index from the host, cost the
  // one-chunk kernel 0.6 - 3 %: register allocation, not work)
  const double gddLeafOn = PRM(gddLeafOn);
  const double leafOffDay = PRM(leafOffDay) > 0 ? PRM(leafOffDay) : 1e300;

  // NCyc: reciprocal C:N ratios for the plants' nitrogen demand (nitrogen.c:89-104) and the test both
  // waves make (nPlentiful); soil carbon lives on wave W then
  const R G_iLeafCN = NCyc ? (R)(1.0 / PRM(leafCN)) : R(0), G_iWoodCN = NCyc ? (R)(1.0 / PRM(woodCN)) : R(0);
  const R G_iFineCN = NCyc ? (R)(1.0 / PRM(fineRootCN)) : R(0), G_resorbC = NCyc ? (R)PRM(leafNResorptionFrac) : R(0);
  const double G_nVolD = NCyc ? PRM(nVolatilizationFrac) : 0.0, G_nLeachD = NCyc ? PRM(nLeachingFrac) : 0.0;
  double plantWoodC = ST(plantWoodC), plantLeafC = ST(plantLeafC), soilC = NCyc ? 0.0 : ST(soilC);
  double coarseRootC = ST(coarseRootC), fineRootC = ST(fineRootC);
  // Ext: growth respiration; Opt: the litter pool, methane and carbon saturation on this wave (rates 0 / share 0
  // with their flags off: the same few instructions run to an exactly unchanged result)
  const R X_growthFrac = F_growthResp ? (R)PRM(growthRespFrac) : R(0);
  const R X_lbrK = F_litterPool ? (R)(PRM(litterBreakdownRate) / PRM(baseSoilResp)) : R(0);
  const R X_flr = Opt ? (R)PRM(fracLitterRespired) : R(0), X_1mFlr = Opt ? (R)(1.0 - PRM(fracLitterRespired)) : R(0);
  const R X_soilCH4 = F_anaerobic ? (R)PRM(soilMethaneRate) : R(0);
  const R X_litCH4 = (F_anaerobic && F_litterPool) ? (R)PRM(litterMethaneRate) : R(0);
  const R X_iSoilCSat = (Opt && F_carbonSat) ? (R)(1.0 / PRM(soilCSaturation)) : R(0);
  double litterC = Opt ? ST(litterC) : 0.0;
  // the soil side of a step with the optional pools (Opt), ONE piece of code for the regular-tile path and the
  // general step (a wavefront's path depends on its neighbours: same bits): calcSoilRespiration / calcLitterFluxes /
  // calcMethaneFlux (sipnet.c:1132-1214) and updatePoolsForSoil (sipnet.c:1645-1668, both forms, one select).
  // soilCNow: the soil carbon the reference looks at for the saturation share -- it already holds the step's events.
  auto optSoilSide = [&](R eSoilC, R eLitter, R soilCNow, R fSoil, R qSoilT, R mK, R rootLoss, R aboveLitter, R len,
                         R& rSoil, R& rHet, double& soilGain, double& litterGain, R& methane) {
#pragma clang fp contract(off)
    rSoil = eSoilC * fSoil;
    const R breakdown = eLitter * (X_lbrK * fSoil);
    const R rLitter = breakdown * X_flr, litterToSoil = breakdown * X_1mFlr;
    const R qm = qSoilT * mK;
    const R soilMethane = X_soilCH4 * eSoilC * qm, litterMethane = X_litCH4 * eLitter * qm;
    const R sat = clip01(soilCNow * X_iSoilCSat);          // 0 without carbon saturation
    const R soilInputs = rootLoss + litterToSoil;
    const R dLitter = aboveLitter + soilInputs * sat - litterToSoil - rLitter - litterMethane;
    const R dSoilTwo = soilInputs * (R(1) - sat) - rSoil - soilMethane;
    const R dSoilOne = rootLoss + aboveLitter - rSoil - soilMethane;
    soilGain = (double)((F_litterPool ? dSoilTwo : dSoilOne) * len);
    litterGain = (double)((F_litterPool ? dLitter : R(0)) * len);
    rHet = rSoil + rLitter;
    methane = soilMethane + litterMethane;   // (Full: record column 31, and the carbon balance)
  };
  double delta = ST(plantCAccountingDelta);
  double ringSum = ST(ringSum), totNee = ST(totNee);
  int phenBits = (int)ST(phenBits);
  int ringValidFrom = (int)ST(ringValidFrom);
  int diedAt = (int)ST(diedAt);
  // Full: the other accumulators of updateTrackers(), the record's constant columns (pools the
  // default flag set never touches) and the diagnostics counters
  double totRtot = Full ? ST(totRtot) : 0.0, totRa = Full ? ST(totRa) : 0.0;
  double totRh = Full ? ST(totRh) : 0.0, totNpp = Full ? ST(totNpp) : 0.0;
  double yGpp = Full ? ST(yearlyGpp) : 0.0, yRtot = Full ? ST(yearlyRtot) : 0.0;
  double yRa = Full ? ST(yearlyRa) : 0.0, yRh = Full ? ST(yearlyRh) : 0.0;
  double yNpp = Full ? ST(yearlyNpp) : 0.0, yNee = Full ? ST(yearlyNee) : 0.0;
  double yLitter = Full ? ST(yearlyLitter) : 0.0;
  const double cLitterC = Full ? ST(litterC) : 0.0, cMinN = Full ? ST(minN) : 0.0;
  const double cSoilOrgN = Full ? ST(soilOrgN) : 0.0, cLitterN = Full ? ST(litterN) : 0.0;
  const double cStorN = Full ? ST(plantStorageN) : 0.0;
  const bool wantDiag = Full && !NCyc && a.diag != nullptr;
  // NCyc, one chunk per workgroup: the pools are spread over two wavefronts -- this one sends the plant side's totals to
  // wave S (mailPend rows 3 .. 13), which runs the balance check; the clamp warnings of the plant pools are counted here
  const bool wantDiagN = Full && NCyc && NP == 1 && a.diag != nullptr;
  int clampWarn = 0, balanceWarn = 0;
  double maxDC = 0.0;
  double* __restrict__ recp = Full && a.rec ? a.rec + col : nullptr;

  // the ring in HBM holds NPP values of type R (fp32-mixed batches: fp32 numbers, stored as such); the LDS copy is fp64
  R* __restrict__ ringp = (R*)a.ring + col;
  R* __restrict__ oNee = (R*)(a.nee ? a.nee : a.scratchRow) + col;
  const int64_t ldNee = a.nee ? a.ld : 0;
  const uint32_t ncu = (uint32_t)nc;
  CoopSums<Sums> sumsC;   // Sums: the group's sum so far, and the steps it still lacks
  if constexpr (Sums) sumsC.left = a.sumEvery;

  post(&mailLai[tBegin & 1][lane], &seqLai, (R)plantLeafC * K_invLcsw, tBegin);
  postAlive(&mailAlive[tBegin & 1][lane], tBegin, false);  // lai(tBegin) is not speculative
  // The two phenology switches fire once a year per member; whether ANY member of the wave can
  // fire in a step is decided with two wave-uniform compares against the smallest thresholds
  // of the wave, and not at all once every member has fired
  double minGddOn = gddLeafOn, minOffDay = leafOffDay;
  for (int off = 32; off > 0; off >>= 1) {
    minGddOn = fmin(minGddOn, __shfl_xor(minGddOn, off, 64));
    minOffDay = fmin(minOffDay, __shfl_xor(minOffDay, off, 64));
  }
  bool allOn = __builtin_amdgcn_ballot_w64((phenBits & 1) == 0) == 0;
  bool allOff = __builtin_amdgcn_ballot_w64((phenBits & 2) == 0) == 0;
  // carried: this member's alive flag (sipnet.c:1530-1544) and, for the whole wave, "every
  // member alive with an untouched ring epoch" (what the regular ring update needs)
  bool aliveC = (plantWoodC > kTiny) && (plantWoodC + delta > kTiny) && (fineRootC + coarseRootC > kTiny);
  bool ringClean = __builtin_amdgcn_ballot_w64(!aliveC || ringValidFrom > 0) == 0;
  if (RingLds) {
    for (int k = 0; k < SIPNET_RING_SLOTS; k++) ringL[k * 64 + lane] = (double)ringp[(uint32_t)k * ncu];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  WAIT_DECL()
  // HBM ring: the value written by the previous step is forwarded from a register
  double lastNpp = 0.0;
  int lastIns = -1;
  CSTAMP_DECL()

  // NCyc: what wave S needs of a step's plant side (the litter fluxes, the nitrogen demand of the creation
  // fluxes nitrogen.c:89-104, the resorption of a negative total creation :170-196, the leaf-on nitrogen
  // :84-86), posted as soon as the creation fluxes are known -- S is working on the soil side of the same
  // step -- and checkNitrogenLimitation() (limitations.c:69-114): both waves test "plentiful" with the
  // same numbers; only where it fails for some member does C wait for S's exact supply, scale its
  // creation fluxes and answer with the demand that is left
  auto plantSideN = [&](int t, R len, R invLen, double minNStep, int minNSeq, double qSoilD, double lenD, R leafLitter, R woodLitter,
                        R fineRootLoss, R coarseRootLoss, R leafOnCreation, R evLeafOnAll, R& leafCreation,
                        R& woodCreation, R& fineRootCreation, R& coarseRootCreation) {
#pragma clang fp contract(off)
    auto leafOnN = [&](R leafOnC) -> R { return rmax0(leafOnC * G_iLeafCN - leafOnC * G_iWoodCN); };
    auto plantNDemand = [&]() -> R {
      return rmax0(woodCreation * G_iWoodCN + leafCreation * G_iLeafCN + fineRootCreation * G_iFineCN +
                   coarseRootCreation * G_iWoodCN);
    };
    R reductionN = 0;
    if (woodCreation + leafCreation + fineRootCreation + coarseRootCreation < R(0))
      reductionN -= (leafCreation * G_iLeafCN + woodCreation * G_iWoodCN + coarseRootCreation * G_iWoodCN +
                     fineRootCreation * G_iFineCN);
    const R nDemand = plantNDemand();
    postD(&mailPlant[t & 1][0][lane], 0, (double)leafLitter);
    postD(&mailPlant[t & 1][0][lane], 1, (double)woodLitter);
    postD(&mailPlant[t & 1][0][lane], 2, (double)fineRootLoss);
    postD(&mailPlant[t & 1][0][lane], 3, (double)coarseRootLoss);
    postD(&mailPlant[t & 1][0][lane], 4, (double)nDemand);
    postD(&mailPlant[t & 1][0][lane], 5, (double)reductionN);
    postD(&mailPlant[t & 1][0][lane], 6, (double)leafOnN(leafOnCreation + evLeafOnAll));
    postD(&mailPlant[t & 1][0][lane], 7, (double)leafOnN(leafOnCreation));
    postFlag(&seqPlant, t);
    // S's mineral nitrogen of this step: normally it came with the factors; if S was not that far yet
    // it is waited for only NOW, after the post S itself is waiting for (C -> S -> C would otherwise be
    // one latency chain per step)
    if (__builtin_expect(minNSeq < t, 0)) {
      WAIT_BEGIN()
      takeD1(&mailMinN[t & 1][lane], &seqMinN, t, minNStep);
      WAIT_END(2)
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!nPlentiful(minNStep, qSoilD, lenD, G_nVolD, G_nLeachD, (double)nDemand)) != 0, 0)) {
      double sAvail, sFixFrac, sUnclaimed;
      takeD3(&mailSupply[0][lane], &seqSupply, t, sAvail, sFixFrac, sUnclaimed);
      const R availableMinN = (R)sAvail, fixFrac = (R)sFixFrac, unclaimed = (R)sUnclaimed;
      const R nUptake = (R(1) - fixFrac) * rmax0(nDemand - unclaimed * invLen);
      const R uptakeDemand = nUptake * len;
      const bool limited = uptakeDemand > R(kTiny) && uptakeDemand > availableMinN;
      const R red = limited ? fdiv(fdiv(availableMinN, R(1) - fixFrac) + unclaimed, nDemand * len) : R(1);
      woodCreation *= red;
      leafCreation *= red;
      fineRootCreation *= red;
      coarseRootCreation *= red;
      postD(&mailDemand[0][lane], 0, (double)plantNDemand());
      postFlag(&seqDemand, t);
    }
  };

  for (int tileStart = curTile * kFastTile; tileStart < tEnd; tileStart += kFastTile, curTile++) {
    // the DMA of this tile was issued a tile ago; only the last step's two stores may still be
    // in flight behind it
    if constexpr (Sums) {   // (no NEE store per step: nothing, or the ring store alone, may stay in flight)
      if (tileStart > tBegin) {
        if (RingLds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      }
    } else {
    if (tileStart > tBegin) {
      if (RingLds) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // + the ring store
    }
    }
    stageTile(curTile + 1, (curTile + 1) & 1);
    const int tFirst = tileStart > tBegin ? tileStart : tBegin;
    const int tLast = (tileStart + kFastTile) < tEnd ? (tileStart + kFastTile) : tEnd;
    const unsigned char* recB = lds + (curTile & 1) * kTileBytes +
                                (int)(tFirst - tileFirst(curTile)) * (int)sizeof(FastRec);
    int t = tFirst;
    // ======== regular tile (plan.h, FastRec::tileBits) ==========================================
    // Every step of the tile has the same length and the same one or two ring evictions with all
    // slots advancing by one, no events, no year roll-over; and, for THIS wavefront, every member
    // is alive with an untouched ring epoch and no phenology switch can fire before the tile ends
    // (year-to-date GDD and the day of year only grow inside a tile).  Then the carbon block needs
    // no per-step record at all: step length and eviction weights are hoisted, the slots counted,
    // the day / night flags come from the tile's mask, and the oldest ring value is the one read
    // as "second eviction" the step before.  Same arithmetic as the general step below; left at
    // the first step on which a member dies.  (Lean instantiation only: records, all accumulators
    // and diagnostics take the general step.)
    if (!Full && ringClean && !(a.options & SIPNET_KOPT_NO_REGULAR_TILES)) {
      d2 h0, h7;
      i4 hj;
      double hW1, hEndGdd, hEndDay;
      int hTile;
      asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:112\n\tds_read_b128 %2, %7 offset:128\n\t"
                   "ds_read_b64 %3, %7 offset:144\n\tds_read_b32 %4, %7 offset:208\n\tds_read_b64 %5, %7 offset:216\n\t"
                   "ds_read_b64 %6, %7 offset:224\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(h0), "=&v"(h7), "=&v"(hj), "=&v"(hW1), "=&v"(hTile), "=&v"(hEndGdd), "=&v"(hEndDay)
                   : "v"(ldsAddr(recB)) : "memory");
      const int tileBits = uni(hTile);
      const bool phenSafe = (allOn || hEndGdd < minGddOn) && (allOff || hEndDay < minOffDay);
      if ((tileBits & FAST_TILE_REGULAR) && __builtin_amdgcn_ballot_w64(!phenSafe) == 0) {
        const R len = (R)h0.x, invLen = (R)h0.y;
        const int nOps = uni(hj.x) >> 16;
        const int slots0 = uni(hj.y);
        auto nextSlot = [](int s) { return s + 1 == SIPNET_RING_SLOTS ? 0 : s + 1; };
        // two evictions: the first one's value is the one read as second eviction a step earlier
        const double wA = nOps == 2 ? h7.y : 0.0, wB = nOps == 2 ? hW1 : h7.y;
        int readSlot = nOps == 2 ? ((slots0 >> 8) & 255) : (slots0 & 255);
        int insSlot = uni(hj.z);
        double vPrev = 0.0;
        if (nOps == 2) {
          const int s0 = slots0 & 255;
          vPrev = RingLds ? ringL[s0 * 64 + lane] : (s0 == lastIns ? lastNpp : (double)ringp[(uint32_t)s0 * ncu]);
        }
        // The rest of a step, written once and instantiated twice: the common case (nobody dies: no
        // mortality code at all behind ONE wave-uniform test) inside the loop, and the step on which a
        // member of the wavefront dies AFTER the loop, which that step leaves (the wavefront takes the
        // general step from then on) -- the loop body stays one straight run of code.  What the
        // tail needs from the step lives outside the loop for that.
        R photosynthesis = 0, rSoil = 0, rVeg = 0, rCoarseRoot = 0, rFineRoot = 0, rHet = 0;
        [[maybe_unused]] R methaneReg = 0;
        double soilGain = 0.0, litterGain = 0.0, ringNew = 0.0;
        R rvN = 0;
        bool rootsOk = true, useLast = false, dyingStep = false;
        auto finishStep = [&](auto mayDie) {
#pragma clang fp contract(off)
          constexpr bool MayDie = decltype(mayDie)::value;
          bool diedNow = false;
          double deathToSoil0 = 0.0, deathToSoil1 = 0.0;
          // wave-uniform (this instantiation runs only when some member dies): the whole wavefront
          // takes the general step from now on -- a per-lane flag would send the survivors down the
          // regular tiles and the dead member down the general step of the SAME tile afterwards
          if (MayDie) ringClean = false;
          if (MayDie && !(rootsOk && (plantWoodC + delta > kTiny))) {  // every member was alive before
            aliveC = false;
            diedNow = true;
            if (diedAt < 0) diedAt = t;
            deathToSoil0 = fineRootC + coarseRootC;
            deathToSoil1 = plantWoodC + plantLeafC + delta;
            if (NCyc) {  // to wave S's pools (sipnet.c:1735-1746), before the verdict word that announces it
              postD(&mailDeath[0][lane], 0, deathToSoil0);
              postD(&mailDeath[0][lane], 1, deathToSoil1);
              postD(&mailDeath[0][lane], 2, fineRootC * (double)G_iFineCN + coarseRootC * (double)G_iWoodCN);
              postD(&mailDeath[0][lane], 3, plantWoodC * (double)G_iWoodCN + plantLeafC * (double)G_iLeafCN);
            }
            plantWoodC = 0.0;
            plantLeafC = 0.0;
            coarseRootC = 0.0;
            fineRootC = 0.0;
            delta = 0.0;
            ringSum = 0.0;
          }
          plantWoodC = rmax0(plantWoodC);
          plantLeafC = rmax0(plantLeafC);
          coarseRootC = rmax0(coarseRootC);
          fineRootC = rmax0(fineRootC);
          const R tGpp = photosynthesis * len;
          const R tRa = ffma(rVeg, len, (rCoarseRoot + rFineRoot) * len);
          if (NCyc) postD(&mailPend[t & 1][0][lane], 0, (double)(tGpp - tRa));   // S forms NEE (it has R_h); before the verdict word
          postAlive(&mailAlive[(t + 1) & 1][lane], t + 1, diedNow);
          if (!NCyc) {
            soilC += soilGain;
            if (Opt) litterC += litterGain;
            if (MayDie && diedNow) {
              soilC += deathToSoil0;
              if (Opt && F_litterPool) litterC += deathToSoil1;   // sipnet.c:1735-1746: above-ground biomass to the litter pool
              else soilC += deathToSoil1;
            }
            soilC = rmax0(soilC);
            if (Opt) litterC = rmax0(litterC);
          }
          const R tRh = (Opt ? rHet : rSoil) * len;
          const R tNee = R(-1.0) * ((tGpp - tRa) - tRh);
          if (!NCyc) totNee += (double)tNee;
          const double npp = (double)(photosynthesis - rVeg - rCoarseRoot - rFineRoot);
          if (!RingLds) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rvN) :: "memory");
          const double vNew = RingLds ? ringNew : (useLast ? lastNpp : (double)rvN);
          if (!(MayDie && diedNow)) {
            ringSum = ffma(-wA, vPrev, ringSum);
            ringSum = ffma(-wB, vNew, ringSum);
            ringSum = ffma(npp, (double)len, ringSum);
          } else {  // its ring epoch starts over; the other members carry on
            ringValidFrom = t + 1;
          }
          vPrev = vNew;
          if (RingLds) {
            ringL[insSlot * 64 + lane] = npp;
          } else {
            ringp[(uint32_t)insSlot * ncu] = (R)npp;   // (npp is an R-typed difference: nothing is lost)
            lastIns = insSlot;
            lastNpp = npp;
          }
          if constexpr (Sums) {
            sumsC.nee += (double)tNee;
            if (--sumsC.left == 0) {
              *oNee = (R)sumsC.nee;
              oNee += ldNee;
              sumsC.nee = 0.0;
              sumsC.left = a.sumEvery;
            }
          } else {
          if (!NCyc) {
            *oNee = tNee;
            oNee += ldNee;
          }
          }
          if (__builtin_expect(stageOn, 0)) postRaw(&stage[0][(t - tBegin) & (2 * kStageR - 1)][lane], tNee);
        };
        unsigned dayMask = ((unsigned)tileBits >> 16) >> (t - tileStart);
        for (; t < tLast; t++, dayMask >>= 1) {
          // The carbon wave's arithmetic is written out with explicit fused multiply-adds and
          // compiled with contraction off (here and in the general step below): which of the two
          // paths a wavefront takes depends on its 63 neighbours, so they must not differ by what
          // the compiler happens to fuse in one context and not in the other.
#pragma clang fp contract(off)
          MARK("C regular step begin")
          // this step's factors: five from wave F / L, the moisture effect from wave W (each flag read
          // before its values, one LDS round trip when both are current)
          R g1, g2, qSoilT, gFine, gCoarse, moistEff;
          R mK = 0;       // Opt: row 6, wave W's methane moisture term
          ringNew = 0.0;  // LDS ring: the value this step evicts, read in the same round trip
          double minNStep = 0.0;   // NCyc: wave S's mineral nitrogen at the start of this step
          int minNSeq = 0;
          {
            WAIT_BEGIN()
            if (NCyc)   // rows 0 1 3 4 and the plain soil Q10 factor (row 6, carried in `moistEff`'s place) + the mineral N
              takeFactorsN(&mailFac[t & 1][0][lane], &seqFac, &mailMinN[t & 1][lane], &seqMinN, t, g1, g2, gFine, gCoarse,
                           moistEff, minNStep, minNSeq);
            else if (Opt && RingLds)
              takeFactorsRing7(&mailFac[t & 1][0][lane], seqFacMoist, ldsAddr(&ringL[readSlot * 64 + lane]), t, g1,
                               g2, qSoilT, gFine, gCoarse, moistEff, mK, ringNew);
            else if (Opt)
              takeFactors7(&mailFac[t & 1][0][lane], seqFacMoist, t, g1, g2, qSoilT, gFine, gCoarse, moistEff, mK);
            else if (RingLds)
              takeFactorsRing(&mailFac[t & 1][0][lane], seqFacMoist, ldsAddr(&ringL[readSlot * 64 + lane]), t, g1,
                              g2, qSoilT, gFine, gCoarse, moistEff, ringNew);
            else
              takeFactors(&mailFac[t & 1][0][lane], seqFacMoist, t, g1, g2, qSoilT, gFine, gCoarse, moistEff);
            WAIT_END(0)
          }
          const R fSoil = NCyc ? R(0) : qSoilT * moistEff;
          rvN = 0;
          if (!RingLds) {
            if (sizeof(R) == 8)
              asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(rvN) : "v"(ringp + (uint32_t)readSlot * ncu) : "memory");
            else
              asm volatile("global_load_dword %0, %1, off" : "=&v"(rvN) : "v"(ringp + (uint32_t)readSlot * ncu) : "memory");
          }
          useLast = readSlot == lastIns;
          const bool isDay = (dayMask & 1u) != 0;

          const R eLeaf = (R)plantLeafC, eSoilC = (R)soilC;
          const R eCoarse = (R)coarseRootC, eFine = (R)fineRootC;
          const R totalWoodC = (R)(plantWoodC + delta);
          const R meanNpp = (R)(ringSum * 0.2);
          const R folResp = eLeaf * g1;
          rVeg = ffma(totalWoodC, g2, folResp);
          if (Ext) rVeg += rmax0(X_growthFrac * meanNpp);   // vegResp2(), sipnet.c:1084-1103 (+0 with the flag off)
          rCoarseRoot = eCoarse * gCoarse;
          rFineRoot = eFine * gFine;
          if (!Opt) rSoil = eSoilC * fSoil;
          const R woodLitter = totalWoodC * K_wtr;
          const R leafLitter = eLeaf * K_ltr;
          R leafCreation = meanNpp * K_la, woodCreation = meanNpp * K_wa;
          const R coarseRootLoss = K_crt * eCoarse, fineRootLoss = K_frt * eFine;
          R coarseRootCreation = K_ca * meanNpp, fineRootCreation = K_fa * meanNpp;
          {  // checkNegativeCreation(), limitations.c:146-182, as selects
            const R leafDeficit = ffma(eLeaf, invLen, leafCreation) - leafLitter;
            const R ld = rminv(leafDeficit, R(0));
            woodCreation += ld;
            leafCreation -= ld;
            const R fineDef = ffma(eFine, invLen, fineRootCreation) - fineRootLoss;
            const R coarseDef = ffma(eCoarse, invLen, coarseRootCreation) - coarseRootLoss;
            const bool fNeg = fineDef < R(0), cNeg = coarseDef < R(0);
            const R shift = (fNeg != cNeg) ? (fNeg ? fineDef : -coarseDef) : R(0);
            coarseRootCreation += shift;
            fineRootCreation -= shift;
          }
          if (NCyc)   // the plants' side of the step for wave S, and checkNitrogenLimitation() (see the general step)
            plantSideN(t, len, invLen, minNStep, minNSeq, (double)moistEff, (double)h0.x, leafLitter, woodLitter, fineRootLoss,
                       coarseRootLoss, R(0), R(0), leafCreation, woodCreation, fineRootCreation, coarseRootCreation);
          accum(plantLeafC, leafCreation - leafLitter, len);
          post(&mailLai[(t + 1) & 1][lane], &seqLai, (R)rmax0(plantLeafC) * K_invLcsw, t + 1);
          accum(plantWoodC, woodCreation - woodLitter, len);
          accum(coarseRootC, coarseRootCreation - coarseRootLoss, len);
          accum(fineRootC, fineRootCreation - fineRootLoss, len);
          if (Opt)
            optSoilSide(eSoilC, (R)litterC, eSoilC, fSoil, qSoilT, mK, coarseRootLoss + fineRootLoss, woodLitter + leafLitter, len,
                        rSoil, rHet, soilGain, litterGain, methaneReg);
          else if (!NCyc) soilGain = (double)((coarseRootLoss + fineRootLoss + woodLitter + leafLitter - rSoil) * len);
          const R r_a = rVeg + rFineRoot + rCoarseRoot;
          const R alloc = leafCreation + woodCreation + fineRootCreation + coarseRootCreation;
          rootsOk = (plantWoodC > kTiny) && (fineRootC + coarseRootC > kTiny);
          photosynthesis = 0;
          if (__builtin_expect(isDay, 0)) {  // laid out for the night step: by night this wave is the critical one
            WAIT_BEGIN()
            photosynthesis = take(&mailPsn[t & 1][lane], &seqPsn, t);
            WAIT_END(1)
          }
          accum(delta, (photosynthesis - r_a) - alloc, len);
          const bool dies = !(rootsOk && (plantWoodC + delta > kTiny));
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(dies) != 0, 0)) {
            dyingStep = true;
            break;
          }
          finishStep(std::false_type{});
          readSlot = nextSlot(readSlot);
          insSlot = nextSlot(insSlot);
          MARK("C regular step end")
        }
        if (dyingStep) {  // finish that step with the mortality code; the general step from the next one on
          finishStep(std::true_type{});
          t++;
        }
        recB += (int)(t - tFirst) * (int)sizeof(FastRec);
      }
    }
  for (; t < tLast; t++, recB += sizeof(FastRec)) {
#pragma clang fp contract(off)  // explicit fused multiply-adds only: see the regular-tile path
    // record fields of the carbon block (len invLen | tsoil10 cumGdd | dayTime w0 | ints) and the
    // five factors of wave F / L and wave W's moisture effect for this step, in ONE LDS round trip; each flag is read before
    // the values (DS reads return in order), so a current flag vouches for what follows it
    d2 q0, q6, q7;
    i4 j0;
    R g1, g2, qSoilT, gFine, gCoarse, moistEff;
    R mK = 0;   // Opt: row 6 of the factor block, wave W's methane moisture term
    int facSeq, moistSeq;
    double minNStep = 0.0;   // NCyc: wave S's mineral nitrogen at the start of this step
    if (Opt) {
      // as the default take below, with row 6 (behind wave W's flag, like row 5)
      WAIT_BEGIN()
      const unsigned fac = ldsAddr(&mailFac[t & 1][0][lane]);
      const unsigned mst = ldsAddr(&mailFac[t & 1][5][lane]);
      WAIT_DO {
        if (sizeof(R) == 8) {
          asm volatile("ds_read_b128 %0, %13\n\tds_read_b128 %1, %13 offset:96\n\tds_read_b128 %2, %13 offset:112\n\t"
                       "ds_read_b128 %3, %13 offset:128\n\tds_read_b32 %4, %14\n\t"
                       "ds_read_b64 %5, %15\n\tds_read_b64 %6, %15 offset:512\n\tds_read_b64 %7, %15 offset:1024\n\t"
                       "ds_read_b64 %8, %15 offset:1536\n\tds_read_b64 %9, %15 offset:2048\n\t"
                       "ds_read_b32 %10, %16\n\tds_read_b64 %11, %17\n\tds_read_b64 %12, %17 offset:512\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(qSoilT), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(moistEff), "=&v"(mK)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMoist)), "v"(mst)
                       : "memory");
        } else {
          asm volatile("ds_read_b128 %0, %13\n\tds_read_b128 %1, %13 offset:96\n\tds_read_b128 %2, %13 offset:112\n\t"
                       "ds_read_b128 %3, %13 offset:128\n\tds_read_b32 %4, %14\n\t"
                       "ds_read_b32 %5, %15\n\tds_read_b32 %6, %15 offset:256\n\tds_read_b32 %7, %15 offset:512\n\t"
                       "ds_read_b32 %8, %15 offset:768\n\tds_read_b32 %9, %15 offset:1024\n\t"
                       "ds_read_b32 %10, %16\n\tds_read_b32 %11, %17\n\tds_read_b32 %12, %17 offset:256\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(qSoilT), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(moistEff), "=&v"(mK)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMoist)), "v"(mst)
                       : "memory");
        }
      } WAIT_WHILE(uni(facSeq) < t || uni(moistSeq) < t, 25, t);
      WAIT_END(0)
    } else if (NCyc) {
      // record fields, wave F's factors (rows 0 1 3 4 and the plain soil Q10 factor, row 6 -- carried in
      // `moistEff`'s place) and wave S's mineral nitrogen, each behind its flag, one round trip
      WAIT_BEGIN()
      const unsigned fac = ldsAddr(&mailFac[t & 1][0][lane]);
      const unsigned mnn = ldsAddr(&mailMinN[t & 1][lane]);
      WAIT_DO {
        if (sizeof(R) == 8) {
          asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:96\n\tds_read_b128 %2, %12 offset:112\n\t"
                       "ds_read_b128 %3, %12 offset:128\n\tds_read_b32 %4, %13\n\t"
                       "ds_read_b64 %5, %14\n\tds_read_b64 %6, %14 offset:512\n\tds_read_b64 %7, %14 offset:3072\n\t"
                       "ds_read_b64 %8, %14 offset:1536\n\tds_read_b64 %9, %14 offset:2048\n\t"
                       "ds_read_b32 %10, %15\n\tds_read_b64 %11, %16\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(moistEff), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(minNStep)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMinN)), "v"(mnn)
                       : "memory");
        } else {
          asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:96\n\tds_read_b128 %2, %12 offset:112\n\t"
                       "ds_read_b128 %3, %12 offset:128\n\tds_read_b32 %4, %13\n\t"
                       "ds_read_b32 %5, %14\n\tds_read_b32 %6, %14 offset:256\n\tds_read_b32 %7, %14 offset:1536\n\t"
                       "ds_read_b32 %8, %14 offset:768\n\tds_read_b32 %9, %14 offset:1024\n\t"
                       "ds_read_b32 %10, %15\n\tds_read_b64 %11, %16\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(moistEff), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(minNStep)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMinN)), "v"(mnn)
                       : "memory");
        }
      } WAIT_WHILE(uni(facSeq) < t, 26, t);   // (`moistSeq`: the mineral nitrogen's flag, looked at in plantSideN)
      qSoilT = 0;
      WAIT_END(0)
    } else {
      WAIT_BEGIN()
      const unsigned fac = ldsAddr(&mailFac[t & 1][0][lane]);
      const unsigned mst = ldsAddr(&mailFac[t & 1][5][lane]);
      // (re-reading the record while spinning is harmless; one asm statement defines every value,
      // so no copies are needed when the first look already finds the flags current)
      WAIT_DO {
        if (sizeof(R) == 8) {
          asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:96\n\tds_read_b128 %2, %12 offset:112\n\t"
                       "ds_read_b128 %3, %12 offset:128\n\tds_read_b32 %4, %13\n\t"
                       "ds_read_b64 %5, %14\n\tds_read_b64 %6, %14 offset:512\n\tds_read_b64 %7, %14 offset:1024\n\t"
                       "ds_read_b64 %8, %14 offset:1536\n\tds_read_b64 %9, %14 offset:2048\n\t"
                       "ds_read_b32 %10, %15\n\tds_read_b64 %11, %16\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(qSoilT), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(moistEff)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMoist)), "v"(mst)
                       : "memory");
        } else {
          asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:96\n\tds_read_b128 %2, %12 offset:112\n\t"
                       "ds_read_b128 %3, %12 offset:128\n\tds_read_b32 %4, %13\n\t"
                       "ds_read_b32 %5, %14\n\tds_read_b32 %6, %14 offset:256\n\tds_read_b32 %7, %14 offset:512\n\t"
                       "ds_read_b32 %8, %14 offset:768\n\tds_read_b32 %9, %14 offset:1024\n\t"
                       "ds_read_b32 %10, %15\n\tds_read_b32 %11, %16\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(q0), "=&v"(q6), "=&v"(q7), "=&v"(j0), "=&v"(facSeq), "=&v"(g1), "=&v"(g2),
                         "=&v"(qSoilT), "=&v"(gFine), "=&v"(gCoarse), "=&v"(moistSeq), "=&v"(moistEff)
                       : "v"(ldsAddr(recB)), "v"(ldsAddr(&seqFac)), "v"(fac), "v"(ldsAddr(&seqMoist)), "v"(mst)
                       : "memory");
        }
      } WAIT_WHILE(uni(facSeq) < t || uni(moistSeq) < t, 27, t);
      WAIT_END(0)
    }
    const R fSoil = qSoilT * moistEff;
    const double* rare = (const double*)(recB + 144);
    const int32_t* rareI = (const int32_t*)(recB + 184);
    CSTAMP(0)
    const R len = (R)q0.x, invLen = (R)q0.y;
    const int bits = uni(j0.x);
    const int slots = uni(j0.y);
    const int insSlot = uni(j0.z);
    const int nEv = uni(j0.w);
    const int evSlot0 = slots & 255, evSlot1 = (slots >> 8) & 255;
    // HBM ring: the values this step evicts are requested now (asm: see the note on waits) and
    // awaited once, right before they are needed at the end of the step
    R rv0 = 0, rv1 = 0;
    if (!RingLds) {
      if (sizeof(R) == 8)
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off"
                     : "=&v"(rv0), "=&v"(rv1)
                     : "v"(ringp + (uint32_t)evSlot0 * ncu), "v"(ringp + (uint32_t)evSlot1 * ncu) : "memory");
      else
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off"
                     : "=&v"(rv0), "=&v"(rv1)
                     : "v"(ringp + (uint32_t)evSlot0 * ncu), "v"(ringp + (uint32_t)evSlot1 * ncu) : "memory");
    }
    const bool useLast0 = evSlot0 == lastIns, useLast1 = evSlot1 == lastIns;

    const bool alive0 = aliveC;
    const R eWood = (R)plantWoodC, eLeaf = (R)plantLeafC, eSoilC = (R)soilC;
    const R eCoarse = (R)coarseRootC, eFine = (R)fineRootC;
    const R eLitter = (R)litterC;   // (Opt) before this step's events, like every pool the fluxes look at
    const R totalWoodC = (R)(plantWoodC + delta);
    // getMassTotals() before the pool updates, balance.c:13-36 (carbon; default flags)
    double preC = 0.0;
    if (wantDiag) preC = (plantWoodC + delta) + plantLeafC + fineRootC + coarseRootC + soilC + (Opt && F_litterPool ? litterC : 0.0);
    R recLeafOffComputed = 0, recEvLeafOn = 0, recEvLeafOnFromWood = 0, recEvLeafOffLitter = 0;
    R evInC = 0, evOutC = 0;
    [[maybe_unused]] R evInN = 0, evOutN = 0;   // NCyc: the events' nitrogen input / output (events.c:530-541, :582-594, :660-672)
    // the plant side's share of getMassTotals() (balance.c:13-36): carbon, and nitrogen through the fixed C:N ratios
    auto plantMassC = [&]() -> double { return (plantWoodC + delta) + plantLeafC + fineRootC + coarseRootC; };
    auto plantMassN = [&]() -> double {
      return plantWoodC * (double)G_iWoodCN + plantLeafC * (double)G_iLeafCN + fineRootC * (double)G_iFineCN + coarseRootC * (double)G_iWoodCN;
    };
    double dgPreC = 0.0, dgPreN = 0.0, dgPostC = 0.0, dgPostN = 0.0;
    if (wantDiagN) {
      dgPreC = plantMassC();
      dgPreN = plantMassN();
    }

    auto leafOnNFromC = [&](R leafOnC) -> R {  // nitrogen.c:84-86
      return rmax0(leafOnC * G_iLeafCN - leafOnC * G_iWoodCN);
    };
    auto leafOnLimit = [&](R flux) -> R {  // limitations.c:13-64
      const R cDemand = flux * len;
      if (cDemand < R(kTiny)) return flux;
      R lim = fdiv((eWood + eCoarse) * PRM_RARE(leafOnReallocFrac), cDemand);
      if (NCyc) {  // the storage nitrogen is wave W's: its value at the start of this step
        double sN;
        takeD1(&mailStorN[t & 1][lane], &seqStorN, t, sN);
        const R nDemand = leafOnNFromC(cDemand);
        if (nDemand > R(kTiny)) lim = rminv(lim, fdiv((R)sN, nDemand));
      }
      lim = clip01(lim);
      return lim < R(1) ? flux * lim : flux;
    };

    const R meanNpp = (R)(ringSum * 0.2);

    // vegResp(), calcRootResp(), calcSoilRespiration() with wave W's factors
    const R folResp = eLeaf * g1;
    R rVeg = ffma(totalWoodC, g2, folResp);
    if (Ext) rVeg += rmax0(X_growthFrac * meanNpp);   // vegResp2(), sipnet.c:1084-1103 (+0 with the flag off)
    const R rCoarseRoot = eCoarse * gCoarse;
    const R rFineRoot = eFine * gFine;
    R rSoil = (NCyc || Opt) ? R(0) : eSoilC * fSoil;   // (Opt: optSoilSide, after the events)
    R rHet = 0;

    const R woodLitter = totalWoodC * K_wtr;
    R leafLitter = eLeaf * K_ltr;
    R leafCreation = meanNpp * K_la, woodCreation = meanNpp * K_wa;

    R leafOnCreation = 0, leafOnFromWood = 0;
    R evLeafOnAll = 0;   // NCyc: leaf-on by event (its nitrogen is part of the step's claim on the storage)
    const bool phenMay = (!allOn && q6.y >= minGddOn) || (!allOff && q7.x >= minOffDay);

    const R coarseRootLoss = K_crt * eCoarse, fineRootLoss = K_frt * eFine;
    R coarseRootCreation = K_ca * meanNpp, fineRootCreation = K_fa * meanNpp;

    // checkNegativeCreation(), limitations.c:146-182, as selects
    {
      const R leafDeficit = ffma(eLeaf, invLen, leafCreation) - eLeaf * K_ltr;
      const R ld = rminv(leafDeficit, R(0));
      woodCreation += ld;
      leafCreation -= ld;
      const R fineDef = ffma(eFine, invLen, fineRootCreation) - fineRootLoss;
      const R coarseDef = ffma(eCoarse, invLen, coarseRootCreation) - coarseRootLoss;
      const bool fNeg = fineDef < R(0), cNeg = coarseDef < R(0);
      const R shift = (fNeg != cNeg) ? (fNeg ? fineDef : -coarseDef) : R(0);
      coarseRootCreation += shift;
      fineRootCreation -= shift;
    }

    CSTAMP(1)

    // events (carbon side; irrigation belongs to wave W) and the yearly phenology switches
    if (__builtin_expect(nEv > 0 || (bits & FAST_PHEN_NEW_YEAR) ||
                         __builtin_amdgcn_ballot_w64(phenMay) != 0, 0)) {
      if (bits & FAST_PHEN_NEW_YEAR) phenBits = 0;
      const bool doOn = !(phenBits & 1) && q6.y >= gddLeafOn;
      const bool doOff = !(phenBits & 2) && q7.x >= leafOffDay;
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
      allOn = __builtin_amdgcn_ballot_w64((phenBits & 1) == 0) == 0;
      allOff = __builtin_amdgcn_ballot_w64((phenBits & 2) == 0) == 0;
      R evLeafC = 0, evWoodC = 0, evFineRootC = 0, evCoarseRootC = 0;
      R evSoilC = 0, evLeafOnCreation = 0, evLeafOnFromWood = 0, evLeafOffLitter = 0;
      // NCyc: the soil side of the events (events.c:575-620, :660-672, :712-722, :778-789), for wave W
      R evLitterC = 0, evMinN = 0, evSoilOrgN = 0, evLitterN = 0, evLeafOffNResorp = 0;
      const bool toLitter = Opt && F_litterPool;   // Opt: above-ground transfers and organic carbon go to the litter pool
      const int ev0 = uni(rareI[3]);
      for (int k = 0; k < nEv; k++) {
        const EvRec& ev = a.events[evBase + ev0 + k];
        const int type = uni(ev.type);
        const R p0 = (R)ev.p[0], p1 = (R)ev.p[1], p2 = (R)ev.p[2], p3 = (R)ev.p[3];
        if (type == SIPNET_EV_PLANT) {
          evLeafC += p0 * invLen;
          evWoodC += p1 * invLen;
          evFineRootC += p2 * invLen;
          evCoarseRootC += p3 * invLen;
          if (Full) evInC += (p0 + p1 + p2 + p3) * invLen;  // events.c:530-541
          if (Full && NCyc) evInN += (p0 * G_iLeafCN + p1 * G_iWoodCN + p2 * G_iFineCN + p3 * G_iWoodCN) * invLen;
        } else if (type == SIPNET_EV_HARVEST) {
          const R woodC = totalWoodC;
          if (Full) evOutC += ((woodC + eLeaf) * p0 + (eFine + eCoarse) * p1) * invLen;  // events.c:582-594
          if (Full && NCyc)
            evOutN += ((eWood * G_iWoodCN + eLeaf * G_iLeafCN) * p0 + (eFine * G_iFineCN + eCoarse * G_iWoodCN) * p1) * invLen;
          if (NCyc) {
            evLitterC += (p2 * (eLeaf + woodC)) * invLen;
            evSoilC += (p3 * (eFine + eCoarse)) * invLen;
            evSoilOrgN += (p3 * (eFine * G_iFineCN + eCoarse * G_iWoodCN)) * invLen;
            evLitterN += (p2 * (eLeaf * G_iLeafCN + eWood * G_iWoodCN)) * invLen;
          } else if (toLitter) {   // events.c:575-580
            evLitterC += (p2 * (eLeaf + woodC)) * invLen;
            evSoilC += (p3 * (eFine + eCoarse)) * invLen;
          } else {
            evSoilC += (p3 * (eFine + eCoarse) + p2 * (eLeaf + woodC)) * invLen;
          }
          evLeafC += -eLeaf * (p0 + p2) * invLen;
          evWoodC += -woodC * (p0 + p2) * invLen;
          evFineRootC += -eFine * (p1 + p3) * invLen;
          evCoarseRootC += -eCoarse * (p1 + p3) * invLen;
        } else if (type == SIPNET_EV_FERT) {
          if (NCyc) {
            evLitterC += p1 * invLen;
            evLitterN += p0 * invLen;
            evMinN += p2 * invLen;
          } else if (toLitter) {
            evLitterC += p1 * invLen;
          } else {
            evSoilC += p1 * invLen;
          }
          if (Full) evInC += p1 * invLen;
          if (Full && NCyc) evInN += (p0 + p2) * invLen;
        } else if (type == SIPNET_EV_LEAFON) {
          const R flux = leafOnLimit(PRM_RARE(leafGrowth) * invLen);
          evLeafOnCreation += flux;
          const R src = eWood + eCoarse;
          if (src > R(kTiny)) evLeafOnFromWood += fdiv(flux * eWood, src);
        } else if (type == SIPNET_EV_LEAFOFF) {
          const R leafOff = eLeaf * PRM_RARE(fracLeafFall);
          evLeafOffLitter += leafOff * invLen;
          if (NCyc) {
            const R leafN = leafOff * G_iLeafCN;
            const R resorb = leafN * G_resorbC;
            evLeafOffNResorp += resorb * invLen;
            evLitterN += (leafN - resorb) * invLen;
          }
        }
      }
      if (NCyc) {
        evLeafOnAll = evLeafOnCreation;
        if (nEv > 0) {  // (W reads this block on every step whose record carries events)
          postD(&mailEvent[t & 1][0][lane], 0, (double)(evLitterC + evLeafOffLitter));
          postD(&mailEvent[t & 1][0][lane], 1, (double)evSoilC);
          postD(&mailEvent[t & 1][0][lane], 2, (double)evMinN);
          postD(&mailEvent[t & 1][0][lane], 3, (double)evSoilOrgN);
          postD(&mailEvent[t & 1][0][lane], 4, (double)evLitterN);
          postD(&mailEvent[t & 1][0][lane], 5, (double)(evLeafOffNResorp - leafOnNFromC(evLeafOnCreation)));
          postFlag(&seqEvent, t);
        }
      }
      if (Full) {
        recEvLeafOn = evLeafOnCreation;
        recEvLeafOnFromWood = evLeafOnFromWood;
        recEvLeafOffLitter = evLeafOffLitter;
      }
      plantWoodC += (double)(evWoodC * len);
      plantLeafC += (double)(evLeafC * len);
      if (!NCyc) soilC += (double)(evSoilC * len);
      plantWoodC -= (double)(evLeafOnFromWood * len);
      coarseRootC -= (double)((evLeafOnCreation - evLeafOnFromWood) * len);
      plantLeafC += (double)((evLeafOnCreation - evLeafOffLitter) * len);
      if (toLitter) {
        litterC += (double)(evLitterC * len);
        litterC += (double)(evLeafOffLitter * len);
      } else if (!NCyc) {
        soilC += (double)(evLeafOffLitter * len);
      }
      coarseRootC += (double)(evCoarseRootC * len);
      fineRootC += (double)(evFineRootC * len);
    }

    // ---- NCyc: the plants' side of the step for wave S, and checkNitrogenLimitation() (plantSideN)
    if (NCyc)
      plantSideN(t, len, invLen, minNStep, uni(moistSeq), (double)moistEff /* the plain soil Q10 factor */, q0.x, leafLitter, woodLitter,
                 fineRootLoss, coarseRootLoss, leafOnCreation, evLeafOnAll, leafCreation, woodCreation, fineRootCreation,
                 coarseRootCreation);

    CSTAMP(2)
    // the leaf pool of the next step does not involve this step's photosynthesis: update it
    // now and let wave L start on step t+1 (speculative only with respect to plant death)
    accum(plantLeafC, leafCreation + leafOnCreation - leafLitter, len);
    post(&mailLai[(t + 1) & 1][lane], &seqLai, (R)rmax0(plantLeafC) * K_invLcsw, t + 1);

    // plant pools that do not involve this step's photosynthesis (sipnet.c:1579-1626)
    accum(plantWoodC, woodCreation - woodLitter - leafOnFromWood, len);
    accum(coarseRootC, coarseRootCreation - coarseRootLoss - (leafOnCreation - leafOnFromWood), len);
    accum(fineRootC, fineRootCreation - fineRootLoss, len);
    double soilGain = 0.0, litterGain = 0.0;
    R methane = 0;
    if (Opt)
      optSoilSide(eSoilC, eLitter, (R)soilC, fSoil, qSoilT, mK, coarseRootLoss + fineRootLoss, woodLitter + leafLitter, len,
                  rSoil, rHet, soilGain, litterGain, methane);
    else if (!NCyc) soilGain = (double)((coarseRootLoss + fineRootLoss + woodLitter + leafLitter - rSoil) * len);
    const R r_a = rVeg + rFineRoot + rCoarseRoot;
    const R alloc = leafCreation + woodCreation + fineRootCreation + coarseRootCreation;
    const bool rootsOk = (plantWoodC > kTiny) && (fineRootC + coarseRootC > kTiny);

    // photosynthesis of this step (wave W after wave L); nights need no hand-over
    R photosynthesis = 0;
    if (bits & FAST_PAR_POS) {
      WAIT_BEGIN()
      photosynthesis = take(&mailPsn[t & 1][lane], &seqPsn, t);
      WAIT_END(1)
    }

    CSTAMP(3)
    // ---- pools (sipnet.c:1769-1806): plant pools first, so that the next step's leaf area
    // can leave for wave L as early as possible
    accum(delta, (photosynthesis - r_a) - alloc, len);
    double postC = 0.0;  // getMassTotals() after the pool updates (the soil pool's is still pending here)
    if (wantDiag)
      postC = (plantWoodC + delta) + plantLeafC + fineRootC + coarseRootC + (soilC + soilGain) +
              (Opt && F_litterPool ? litterC + litterGain : 0.0);
    if (wantDiagN) {
      dgPostC = plantMassC();
      dgPostN = plantMassN();
    }
    double deathWood = 0.0, deathRoot = 0.0;  // record columns 41, 42
    // checkForMortality(), sipnet.c:1688-1767
    bool alive = alive0;
    double deathToSoil0 = 0.0, deathToSoil1 = 0.0;
    bool diedNow = false;
    {
      const bool sufficient = rootsOk && (plantWoodC + delta > kTiny);
      // a ring epoch stays behind: the whole wavefront takes the general path from now on.  The flag
      // must stay wave-uniform -- set per lane it sent the survivors through the regular tiles and
      // the dead member through the general step of the same tile afterwards (found by the fuzzer:
      // a single member killed by a harvest, regular tiles following)
      if (__builtin_amdgcn_ballot_w64(sufficient != alive0) != 0) ringClean = false;
      if (__builtin_expect(sufficient != alive0, 0)) {
        if (!alive0) {
          alive = true;
        } else {
          alive = false;
          diedNow = true;
          if (diedAt < 0) diedAt = t;
          deathToSoil0 = fineRootC + coarseRootC;
          deathToSoil1 = plantWoodC + plantLeafC + delta;
          if (NCyc) {  // sipnet.c:1735-1746: to wave W's pools; posted before the verdict word that announces it
            postD(&mailDeath[0][lane], 0, deathToSoil0);
            postD(&mailDeath[0][lane], 1, deathToSoil1);
            postD(&mailDeath[0][lane], 2, fineRootC * (double)G_iFineCN + coarseRootC * (double)G_iWoodCN);
            postD(&mailDeath[0][lane], 3, plantWoodC * (double)G_iWoodCN + plantLeafC * (double)G_iLeafCN);
          }
          if (Full) {
            deathWood = plantWoodC + delta;
            deathRoot = deathToSoil0;
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
    aliveC = alive;
    if (wantDiag || wantDiagN) {  // clamp warnings, sipnet.c:1346-1356
      clampWarn += (plantWoodC < 0.0 && fabs(plantWoodC) > kEps) + (plantLeafC < 0.0 && fabs(plantLeafC) > kEps) +
                   (coarseRootC < 0.0 && fabs(coarseRootC) > kEps) + (fineRootC < 0.0 && fabs(fineRootC) > kEps);
    }
    plantWoodC = rmax0(plantWoodC);
    plantLeafC = rmax0(plantLeafC);
    coarseRootC = rmax0(coarseRootC);
    fineRootC = rmax0(fineRootC);
    // NCyc: GPP - R_a of this step for wave S, which has R_h and forms NEE; before the verdict word
    if (NCyc) {
      postD(&mailPend[t & 1][0][lane], 0, (double)(photosynthesis * len - ffma(rVeg, len, (rCoarseRoot + rFineRoot) * len)));
      if (Full) {
        postD(&mailPend[t & 1][0][lane], 1, (double)ffma(rVeg, len, (rCoarseRoot + rFineRoot) * len));
        postD(&mailPend[t & 1][0][lane], 2, (double)((rCoarseRoot + rFineRoot) * len));
      }
      if (kPendRows > 3 && wantDiagN) {   // for wave S's balance check: totals before / after the updates / after the clamps, rates
        postD(&mailPend[t & 1][0][lane], 3, dgPreC);
        postD(&mailPend[t & 1][0][lane], 4, dgPreN);
        postD(&mailPend[t & 1][0][lane], 5, dgPostC);
        postD(&mailPend[t & 1][0][lane], 6, dgPostN);
        postD(&mailPend[t & 1][0][lane], 7, plantMassC());
        postD(&mailPend[t & 1][0][lane], 8, plantMassN());
        postD(&mailPend[t & 1][0][lane], 9, (double)photosynthesis + (double)evInC);
        postD(&mailPend[t & 1][0][lane], 10, (double)rVeg + (double)rFineRoot + (double)rCoarseRoot);
        postD(&mailPend[t & 1][0][lane], 11, (double)evOutC);
        postD(&mailPend[t & 1][0][lane], 12, (double)evInN);
        postD(&mailPend[t & 1][0][lane], 13, (double)evOutN);
      }
    }
    // confirms the lai(t+1) posted above, or revokes it when the stand died in this step (its
    // leaf pool was just zeroed); a stand that was never alive keeps its leaves and its lai
    postAlive(&mailAlive[(t + 1) & 1][lane], t + 1, diedNow);

    CSTAMP(4)
    soilC += soilGain;
    if (Opt) litterC += litterGain;
    if (!NCyc && __builtin_expect(__builtin_amdgcn_ballot_w64(diedNow) != 0, 0)) {
      if (diedNow) {
        soilC += deathToSoil0;
        if (Opt && F_litterPool) litterC += deathToSoil1;   // sipnet.c:1735-1746
        else soilC += deathToSoil1;
      }
    }
    if (wantDiag && soilC < 0.0 && fabs(soilC) > kEps) clampWarn++;
    soilC = rmax0(soilC);
    if (Opt && wantDiag && F_litterPool && litterC < 0.0 && fabs(litterC) > kEps) clampWarn++;
    if (Opt) litterC = rmax0(litterC);
    if (wantDiag) {  // updateBalanceTrackerPostClamp() + checkBalance(), balance.c:40-169
      const double finC = (plantWoodC + delta) + plantLeafC + fineRootC + coarseRootC + soilC + (Opt && F_litterPool ? litterC : 0.0);
      double clampedC = finC - postC;
      if (clampedC < kEps) clampedC = 0.0;
      double inC = ((double)photosynthesis + (double)evInC) * (double)len;
      const double outC = ((double)rVeg + (double)rFineRoot + (double)rCoarseRoot + (double)(Opt ? rHet : rSoil) +
                           (double)methane + (double)evOutC) * (double)len;
      inC += clampedC;
      const double dC = (finC - preC) - (inC - outC);
      maxDC = fmax(maxDC, fabs(dC));
      if (!(fabs(dC) < kEps)) balanceWarn++;
    }

    // ---- outputs: updateTrackers(), sipnet.c:1420-1496 ---------------------------------------
    const R tGpp = photosynthesis * len;
    const R tRh = (Opt ? rHet : rSoil) * len;
    const R tRa = ffma(rVeg, len, (rCoarseRoot + rFineRoot) * len);
    const R tNee = R(-1.0) * ((tGpp - tRa) - tRh);
    if (!NCyc) totNee += (double)tNee;
    R tRAbove = 0, tRRoot = 0, tRSoil = 0, tRtot = 0, tNpp = 0;
    if (Full) {
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
      yLitter += (double)(leafLitter + recEvLeafOffLitter);
    }

    // ---- running mean of NPP (sipnet.c:1546-1570, runmean.c:61-116 via the plan) -------------
    const double npp = (double)(photosynthesis - rVeg - rCoarseRoot - rFineRoot);
    const double recMeanNpp = Full ? ringSum / 5.0 : 0.0;  // trackers.meanNPP: the mean BEFORE this step's insert
    CSTAMP(5)
    {
      if (!RingLds) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rv0), "+v"(rv1) :: "memory");
      const double v0 = RingLds ? ringL[evSlot0 * 64 + lane] : (useLast0 ? lastNpp : (double)rv0);
      const int nOps = bits >> 16;
      // one or two evictions and a plain insert (two is the steady state of half-hourly forcing,
      // see the regular-tile path); with one eviction w1 is 0 and its term an exact no-op
      if (__builtin_expect(ringClean && insSlot >= 0 && nOps <= 2, 1)) {
        const double v1 = RingLds ? ringL[evSlot1 * 64 + lane] : (useLast1 ? lastNpp : (double)rv1);
        ringSum = ffma(-q7.y, v0, ringSum);
        ringSum = ffma(-rare[0], v1, ringSum);
        ringSum = ffma(npp, (double)len, ringSum);
      } else if (alive) {
        if (insSlot < 0) {
          ringSum = npp * 5.0;
        } else {
          double w0v = v0, w1v = RingLds ? ringL[evSlot1 * 64 + lane] : (useLast1 ? lastNpp : (double)rv1);
          if (ringValidFrom > 0) {
            if (uni(rareI[0]) < ringValidFrom) w0v = 0.0;
            if (uni(rareI[1]) < ringValidFrom) w1v = 0.0;
          }
          ringSum = ffma(-q7.y, w0v, ringSum);
          ringSum = ffma(-rare[0], w1v, ringSum);
          for (int k = 2; k < nOps; k++) {
            const RingOp& op = a.ringOps[opBase + uni(rareI[2]) + k];
            const int os = uni(op.slot);
            const double rvk = RingLds ? ringL[os * 64 + lane]
                                       : (os == lastIns ? lastNpp : (double)ringp[(uint32_t)os * ncu]);
            const double v = (uni(op.insStep) >= ringValidFrom) ? rvk : 0.0;
            ringSum = ffma(-op.w, v, ringSum);
          }
          ringSum = ffma(npp, (double)len, ringSum);
        }
      } else {
        ringValidFrom = t + 1;
      }
    }
    const int insEff = insSlot < 0 ? 0 : insSlot;
    if (RingLds) {
      ringL[insEff * 64 + lane] = npp;
    } else {
      ringp[(uint32_t)insEff * ncu] = (R)npp;
      lastIns = insEff;
      lastNpp = npp;
    }
    if (Full && recp) {  // the carbon / tracker columns of the strict kernel's record row
      double* __restrict__ r = recp;
      const int64_t L = a.ld;
      if (!NCyc) {   // (NCyc: wave S has the heterotrophic side and the soil / nitrogen pools, and writes these)
        r[0 * L] = (double)tNee;
        r[3 * L] = totNee;
        r[6 * L] = (double)tRSoil;
        r[9 * L] = (double)tRh;
        r[10 * L] = (double)tRtot;
        r[16 * L] = soilC;
        r[18 * L] = Opt ? litterC : cLitterC;
        r[22 * L] = cMinN;
        r[23 * L] = cSoilOrgN;
        r[24 * L] = cLitterN;
        r[25 * L] = cStorN;
        r[27 * L] = 0.0;
        r[28 * L] = 0.0;
        r[29 * L] = 0.0;
        r[30 * L] = 0.0;
        r[31 * L] = (double)(methane * len);
      }
      r[4 * L] = (double)tNpp;
      r[5 * L] = (double)tRAbove;
      r[7 * L] = (double)tRRoot;
      r[8 * L] = (double)tRa;
      r[11 * L] = (double)(woodCreation * len);
      r[14 * L] = plantWoodC;
      r[15 * L] = plantLeafC;
      r[20 * L] = coarseRootC;
      r[21 * L] = fineRootC;
      r[26 * L] = delta;
      r[32 * L] = recMeanNpp;
      r[33 * L] = rare[3];  // gddAfter
      r[34 * L] = rare[4];  // tillAfter
      r[36 * L] = (double)(leafOnCreation * len);
      r[37 * L] = (double)(leafOnFromWood * len);
      r[38 * L] = (double)(recLeafOffComputed * len);
      r[39 * L] = (double)(recEvLeafOn * len);
      r[40 * L] = (double)(recEvLeafOnFromWood * len);
      r[41 * L] = deathWood;
      r[42 * L] = deathRoot;
      r[43 * L] = diedNow ? 1.0 : 0.0;
      recp += (int64_t)SIPNET_NREC * L;
    }
    if constexpr (Sums) {
      sumsC.nee += (double)tNee;
      if (--sumsC.left == 0) {
        *oNee = (R)sumsC.nee;
        oNee += ldNee;
        sumsC.nee = 0.0;
        sumsC.left = a.sumEvery;
      }
    } else {
    if (!NCyc) {
      *oNee = tNee;
      oNee += ldNee;
    }
    }
    if (__builtin_expect(stageOn, 0)) postRaw(&stage[0][(t - tBegin) & (2 * kStageR - 1)][lane], tNee);
    CSTAMP(6)
  }  // steps of this tile
  }  // tiles
  if constexpr (Sums) {
    if (sumsC.left != a.sumEvery) *oNee = (R)sumsC.nee;   // the launch's last, shorter group
  }

  CSTAMP_STORE()
  WAIT_STORE(8)
  if (a.statsPart) {  // every NEE store of the launch has reached L2 (see wave L's statistics)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("ds_write_b32 %0, %1" :: "v"(ldsAddr(&seqDone[0])), "v"(1) : "memory");
  }
  if (act) {
    if (RingLds)
      for (int k = 0; k < SIPNET_RING_SLOTS; k++) ringp[(uint32_t)k * ncu] = (R)ringL[k * 64 + lane];
    ST(plantWoodC) = plantWoodC;
    ST(plantLeafC) = plantLeafC;
    if (!NCyc) ST(soilC) = soilC;
    if (Opt) ST(litterC) = litterC;
    ST(coarseRootC) = coarseRootC;
    ST(fineRootC) = fineRootC;
    ST(plantCAccountingDelta) = delta;
    ST(ringSum) = ringSum;
    if (!NCyc) ST(totNee) = totNee;
    ST(phenBits) = (double)phenBits;
    ST(ringValidFrom) = (double)ringValidFrom;
    ST(diedAt) = (double)diedAt;
    if (Full) {
      ST(totRa) = totRa;
      ST(totNpp) = totNpp;
      ST(yearlyGpp) = yGpp;
      ST(yearlyRa) = yRa;
      ST(yearlyNpp) = yNpp;
      ST(yearlyLitter) = yLitter;
      if (!NCyc) {   // (NCyc: the accumulators with R_h in them are wave S's)
        ST(totRtot) = totRtot;
        ST(totRh) = totRh;
        ST(yearlyRtot) = yRtot;
        ST(yearlyRh) = yRh;
        ST(yearlyNee) = yNee;
      }
    }
    if (wantDiag) {
      double* __restrict__ dg = a.diag + col;
      if (clampWarn) atomicAdd(dg, (double)clampWarn);  // the water wave adds its two pools' count
      dg[1 * nc] += (double)balanceWarn;
      dg[2 * nc] = fmax(dg[2 * nc], maxDC);
    }
    if (wantDiagN && clampWarn) atomicAdd(a.diag + col, (double)clampWarn);   // (the balance counters are wave S's)
  }
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

#ifdef SIPNET_HWID
extern "C" int sipnet_debug_read_coop_hwid(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopHwId), sizeof(unsigned) * 4096 * 4 * 2);
}
#endif
#ifdef SIPNET_WAITS
extern "C" int sipnet_debug_read_coop_waits(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coopWaits), 16 * sizeof(unsigned long long));
}
#endif
#ifdef SIPNET_STAMPS
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
    if (ext && a.full) {
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
      snprintf(info->kernel, sizeof info->kernel, "%s<%s, %s>",
               (ext && a.full) ? (pairN ? "stepCoopNXPairFullKernel" : "stepCoopNXFullKernel")
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
  if (a.sumEvery > 0) {   // (the engine sends fp64, default-physics, lean launches of these three layouts only)
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
    if (a.sumEvery > 0 && pair) snprintf(info->kernel, sizeof info->kernel, "stepCoopPairSumsKernel<%s>", pe);
    else if (a.sumEvery > 0) snprintf(info->kernel, sizeof info->kernel, "stepCoopSumsKernel<%s, %s>", pe, ringInLds ? "true" : "false");
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

#ifdef SIPNET_COOP_BOUNDED
}  // namespace bounded
#endif
}  // namespace sipnet

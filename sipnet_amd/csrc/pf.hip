// pf.hip -- particle-filter analysis step for an ensemble batch (BASELINE config C5,
// SURVEY.md 8(e) "PF extra exchange"): likelihood weights from an output plane, systematic
// resampling over the GLOBAL particle set, and the column gathers that move member state
// (carried state vector + running-mean ring, optionally the converted parameters) inside a
// GPU and into / out of the packed blocks that travel between GPUs.
//
// The reference has no particle filter (PEcAn drives one process per particle and copies
// restart files between cycles, docs/developer-guide/restart-checkpoint.md); what a cycle
// exchanges there is exactly a member's checkpoint, which is what these kernels move.
//
// All kernels are HBM-bound column gathers over SoA matrices [rows][ncol]: thread = column,
// blockIdx.y = row, so reads of the index vector and writes are coalesced; after systematic
// resampling ancestors are non-decreasing, so the gathered reads are near-coalesced too.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <unistd.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"
#include "batch_impl.h"

namespace sipnet {
namespace {

constexpr int kMaxBlocks = 16;  // source ranks whose packed blocks one gather can read

// Where the received columns live: block s holds [rows][n[s]] doubles at off[s]; received
// column k (0-based over all blocks) is in the block with start[s] <= k < start[s+1].
struct RecvMap {
  int32_t nBlocks;
  int64_t start[kMaxBlocks + 1];
  int64_t off[kMaxBlocks];
  int64_t n[kMaxBlocks];
};

// dst[row][j] = (src[j] < ncol) ? own[row][src[j]] : recv(row + recvRow0, src[j] - ncol)
// One thread moves kGatherRows rows of its column: the source index is read once and
// kGatherRows independent loads are in flight per thread (HBM-bound streaming copy).
constexpr int kGatherRows = 8;
// rows of 8-byte words the ring takes in a packed block (a row of floats = half a row of words; 250 is even)
static inline int ringWords(bool ringF32) { return ringF32 ? SIPNET_RING_SLOTS / 2 : SIPNET_RING_SLOTS; }
static_assert(SIPNET_RING_SLOTS % 2 == 0, "a packed block keeps the parameter rows 8-byte aligned");
// One launch for the three matrices of a member's checkpoint (state rows, ring rows,
// then parameter rows: the row order of a packed block): blockIdx.y walks the row groups of all three,
// so a resampling or a pack is one kernel instead of three (launch gaps were a third of the analysis
// step's GPU time, profiles/r02_c5.md)
struct GatherPart {
  const void* own;     // [rows][ownPitch] of 8-byte (doubles) or 4-byte (floats: the ring of an fp32-mixed batch) elements
  void* dst;           // [rows][dstPitch]
  int32_t rows, group0;   // group0: first row group (blockIdx.y) of this part
  int32_t recvRow0;       // where the part starts inside a packed block, in rows of 8-byte words
  int32_t elem4;          // 4-byte elements
  const int32_t* remap;   // null, or: an OWN source column s is read from column remap[s] (the parameter bank's index)
};
struct GatherParts {
  GatherPart p[3];
  int32_t n;
};
template <typename T>
__device__ __forceinline__ void gatherRows(const GatherPart& part, int row0, int nr, int64_t s, int64_t j, int64_t ownPitch,
                                           int64_t ncol, const double* __restrict__ recv, const RecvMap& map,
                                           int64_t dstPitch) {
  T v[kGatherRows];
  if (s < ncol) {
    const T* __restrict__ p = (const T*)part.own + (int64_t)row0 * ownPitch + (part.remap ? (int64_t)part.remap[s] : s);
    if (nr == kGatherRows) {
#pragma unroll
      for (int r = 0; r < kGatherRows; r++) v[r] = p[(int64_t)r * ownPitch];
    } else {
      for (int r = 0; r < nr; r++) v[r] = p[(int64_t)r * ownPitch];
    }
  } else {
    const int64_t kk = s - ncol;
    int blk = 0;
    for (int q = 1; q < map.nBlocks; q++)
      if (kk >= map.start[q]) blk = q;
    // (the part's first word inside the block, then rows of the part's own element type)
    const T* __restrict__ p = (const T*)(recv + map.off[blk] + (int64_t)part.recvRow0 * map.n[blk]) +
                              (int64_t)row0 * map.n[blk] + (kk - map.start[blk]);
    for (int r = 0; r < nr; r++) v[r] = p[(int64_t)r * map.n[blk]];
  }
  T* __restrict__ q = (T*)part.dst + (int64_t)row0 * dstPitch + j;
  if (nr == kGatherRows) {
#pragma unroll
    for (int r = 0; r < kGatherRows; r++) q[(int64_t)r * dstPitch] = v[r];
  } else {
    for (int r = 0; r < nr; r++) q[(int64_t)r * dstPitch] = v[r];
  }
}
__global__ __launch_bounds__(256) void gatherMemberKernel(GatherParts parts, int64_t ownPitch, int64_t ncol,
                                                          const double* __restrict__ recv, RecvMap map,
                                                          const int32_t* __restrict__ src, int64_t nOut,
                                                          int64_t dstPitch) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nOut) return;
  int k = 0;
  for (int q = 1; q < parts.n; q++)
    if ((int)blockIdx.y >= parts.p[q].group0) k = q;
  const GatherPart part = parts.p[k];
  const int row0 = ((int)blockIdx.y - part.group0) * kGatherRows;
  const int nr = part.rows - row0 < kGatherRows ? part.rows - row0 : kGatherRows;
  // (clamped: the index vector of an analysis whose launch was void -- pfFusedKernel's barrier gave up -- is whatever the
  // caller's buffer held; the results are void either way, the reads must stay inside the matrices)
  int64_t s = src[j];
  const int64_t sMax = ncol + (map.nBlocks > 0 ? map.start[map.nBlocks] : 0) - 1;
  s = s < 0 ? 0 : s > sMax ? sMax : s;
  if (part.elem4) gatherRows<float>(part, row0, nr, s, j, ownPitch, ncol, recv, map, dstPitch);
  else gatherRows<double>(part, row0, nr, s, j, ownPitch, ncol, recv, map, dstPitch);
}

// logw[col] = -0.5 * ((sum_t plane[t][col] - obs) / sigma)^2, -inf for members that did not run
// (AgentStore: the log-weight is written through to device scope -- pfFusedKernel's phase 3 reads it from other workgroups,
// possibly on another XCD, inside the same launch)
template <typename T, bool AgentStore = false>
__device__ __forceinline__ double logWeightOf(const T* __restrict__ plane, int32_t nSteps, int64_t ld, int64_t c,
                                              const double* __restrict__ status, double obs, double invSigma,
                                              double* __restrict__ logw) {
  // the sum in step order; eight loads in flight at a time (one dependent load per step left the kernel
  // latency-bound: 25 MB in 14.7 us at C5's shape)
  double acc = 0.0;
  int t = 0;
  for (; t + 8 <= nSteps; t += 8) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = plane[(int64_t)(t + k) * ld + c];
#pragma unroll
    for (int k = 0; k < 8; k++) acc += (double)v[k];
  }
  for (; t < nSteps; t++) acc += (double)plane[(int64_t)t * ld + c];
  const double z = (acc - obs) * invSigma;
  const double lw = (status[c] != 0.0) ? -INFINITY : -0.5 * z * z;
  if (AgentStore) __hip_atomic_store(&logw[c], lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else logw[c] = lw;
  return lw;
}
// (part, if given: the block's maximum -- what maxPartialKernel would compute in a launch of its own)
template <typename T>
__global__ __launch_bounds__(256) void logWeightKernel(const T* __restrict__ plane, int32_t nSteps,
                                                       int64_t ld, int64_t ncol,
                                                       const double* __restrict__ status,
                                                       double obs, double invSigma,
                                                       double* __restrict__ logw, double* __restrict__ part,
                                                       int64_t npad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double mine = -INFINITY;
  if (c < ncol) mine = logWeightOf(plane, nSteps, ld, c, status, obs, invSigma, logw);
  else if (c < npad) logw[c] = -INFINITY;   // slots of a rank's block no particle fills (ragged shards): weight zero
  if (part) {
    __shared__ double sm[256];
    sm[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
  }
}
// the same block (256-wide maxima behind the log-weights, -inf in the slots no particle fills) when the forecast's own launch
// has left the log-weights in place and one maximum per 64 columns (FastArgs::pfLogw): four of those per entry
__global__ __launch_bounds__(256) void blockFromWaveMaximaKernel(const double* __restrict__ waveMax, int64_t nWaves, int64_t ncol,
                                                               int64_t npad, double* __restrict__ logw, double* __restrict__ part) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nPart = (npad + 255) / 256;
  if (i < nPart) {
    double m = -INFINITY;
    for (int k = 0; k < 4; k++)
      if (4 * i + k < nWaves) m = fmax(m, waveMax[4 * i + k]);
    part[i] = m;
  }
  if (ncol + i < npad) logw[ncol + i] = -INFINITY;
}
// (the scratch blocks are freed by sipnet_pf_release_scratch, not by a thread-exit destructor: that
// may run after the HIP runtime has shut down)

// max of the log-weights: per-block partial maxima (the consumer, fixedWeightKernel, takes their maximum)
__global__ __launch_bounds__(256) void maxPartialKernel(const double* __restrict__ x, int64_t n,
                                                        double* __restrict__ part) {
  __shared__ double sm[256];
  double m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, x[i]);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// THE fixed-point weight of a log-weight lw under the maximum m: rint(2^30 exp(lw - m)), 0 for a particle that did not run
// (-inf) or a filter none of whose particles did.  Every device path takes it from here -- the separate-launch kernels, the
// one-launch analysis, one rank or many -- so that they agree to the bit (the numpy oracle's glibc exp may round a weight
// to the neighbouring integer; the tests allow that one unit and resample the DEVICE's integers exactly).
// 2^30 e^x = 2^(30 + x log2 e): n = rint(y), 2^(y - n) by the degree-11 interpolant of fast_math.h (|rel err| <= 1.7e-16),
// one v_ldexp -- a fifth of OCML's exp() + llrint(), which was what grew with the number of ranks: 8 x 131 072 slots cost
// every wavefront eight of them (7.9 us of the analysis launch, profiles/r06_pf_analysis_phases.txt).
__device__ __forceinline__ long long pfFixedWeight(double lw, double m) {
  const double x = lw - m;                  // <= 0 (NaN when both are -inf)
  if (!(x >= -21.5)) return 0;              // 2^30 e^x < 0.5 below that; also lw = -inf, m = -inf (NaN), NaN weights
  const double y = x * 1.4426950408889634074;
  const double n = __builtin_rint(y), f = y - n;
  double p = 4.4549605981865186e-10;
  p = __builtin_fma(p, f, 7.072585949269223e-09);
  p = __builtin_fma(p, f, 1.0178062445845774e-07);
  p = __builtin_fma(p, f, 1.321544258792169e-06);
  p = __builtin_fma(p, f, 1.525273382983612e-05);
  p = __builtin_fma(p, f, 0.0001540353044173605);
  p = __builtin_fma(p, f, 0.0013333558146416936);
  p = __builtin_fma(p, f, 0.009618129107606888);
  p = __builtin_fma(p, f, 0.0555041086648216);
  p = __builtin_fma(p, f, 0.24022650695910097);
  p = __builtin_fma(p, f, 0.6931471805599453);
  p = __builtin_fma(p, f, 1.0);
  return (long long)(int)__builtin_rint(__builtin_amdgcn_ldexp(p, (int)n + 30));   // (<= 2^30: an int)
}

// fixed-point weights: w = rint(exp(logw - max) * 2^30).  Integer weights make the prefix
// sum exact, so every rank computes bit-identical ancestors from the same gathered logw.
// (every block first takes the maximum of the `parts` partial maxima itself: one launch less)
__global__ __launch_bounds__(256) void fixedWeightKernel(const double* __restrict__ logw,
                                                         int64_t n, const double* __restrict__ part, int parts,
                                                         int64_t* __restrict__ w) {
  __shared__ double sm[256];
  double pm = -INFINITY;
  for (int k = threadIdx.x; k < parts; k += 256) pm = fmax(pm, part[k]);
  sm[threadIdx.x] = pm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  w[i] = pfFixedWeight(logw[i], sm[0]);
}

// ancestor[j] = first slot i with cdf[i] > p_j, p_j = ((j0 + j + u0) * S) / nTotal  (S = cdf[nSlots-1] < 2^53)
// for the nOut particles j0 .. j0 + nOut - 1 of a filter of nTotal particles whose weights sit in nSlots >= nTotal
// slots (one rank: nSlots = nTotal = nOut, j0 = 0; several ranks: a rank resamples its own particles over the
// gathered weights of all, and slots no particle fills weigh nothing)
// (total, if wanted: the total integer weight, for the caller's "a particle survived" check)
__global__ __launch_bounds__(256) void ancestorKernel(const int64_t* __restrict__ cdf, int64_t nSlots, int64_t j0,
                                                      int64_t nOut, int64_t nTotal, double u0,
                                                      int32_t* __restrict__ anc, int64_t* __restrict__ total) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nOut) return;
  if (j == 0 && total) *total = cdf[nSlots - 1];
  const double S = (double)cdf[nSlots - 1];
  // S - 1 keeps the search inside the support when (j + u0) rounds up to n
  const double p = fmin((((double)(j0 + j) + u0) * S) / (double)nTotal, S - 1.0);
  int64_t lo = 0, hi = nSlots - 1;  // invariant: answer in [lo, hi]
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((double)cdf[mid] > p) {
      hi = mid;
    } else {
      lo = mid + 1;
    }
  }
  anc[j] = (int32_t)lo;
}

// ---- a filter spread over ranks, without an all-to-all: every rank reads the ancestors it needs straight out
// of its peers' checkpoint matrices (peer-mapped HBM over xGMI; sipnet_batch_pf_publish / _connect) -------------
// What the ranks all-gather is one block per rank: [nmax log-weights (slots past the rank's own particles: -inf) |
// P = ceil(nmax / 256) block maxima of them], stride = nmax + P doubles.  Slot s * nmax + c = particle c of rank s.
constexpr int kMaxPeers = 16;
struct PeerPtrs {            // kernel argument: where rank s keeps its particles' checkpoint matrices
  int32_t world, nmax;
  const double* state[kMaxPeers];
  const void* ring[kMaxPeers];
  const void* third[kMaxPeers];   // the converted parameter rows [NPARAMS][pitch] -- or, with all ranks' parameters replicated
                                  // on every rank (sipnet_batch::d_prmBank), the particles' index into that bank [pitch] int32
  int32_t pitch[kMaxPeers];  // particles of rank s = the leading dimension of its matrices
  int32_t rank;                   // the reading rank
  unsigned long long* crossing;   // += particles read from another rank's matrices (null: not counted)
};
// fixed-point weights of all slots (fixedWeightKernel over the gathered blocks)
__global__ __launch_bounds__(256) void fixedWeightGatheredKernel(const double* __restrict__ gathered, int32_t world,
                                                                 int32_t nmax, int64_t stride, int64_t* __restrict__ w) {
  __shared__ double sm[256];
  const int P = (nmax + 255) / 256;
  double pm = -INFINITY;
  for (int k = threadIdx.x; k < world * P; k += 256) pm = fmax(pm, gathered[(int64_t)(k / P) * stride + nmax + k % P]);
  sm[threadIdx.x] = pm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)world * nmax) return;
  w[i] = pfFixedWeight(gathered[(i / nmax) * stride + i % nmax], sm[0]);
}
// ---- the analysis in ONE launch (round 5; geometry, barrier and phases reworked in round 6) ------------------------
// Log-weights + maximum | fixed-point weights + prefix sum | ancestors were five launches plus hipCUB's two (and
// its temporaries' fills and copies): 40 us of a 200 us cycle at C5's shape for 15 us of work.  Here they are the
// phases of one kernel whose workgroups are all resident and meet at barriers in device memory.  Workgroup b owns the
// contiguous slots [b * chunk, (b + 1) * chunk) and, inside it, thread t the CONSECUTIVE slots [t * per, (t + 1) * per)
// (per = chunk / 256): a thread sums its own weights serially, ONE block scan per workgroup places the threads' sums
// (round 5 scanned every 256 slots with two __syncthreads: a term that grew with the number of ranks, 8 tiles per
// workgroup at 8 x 131 072 slots), the chunks' totals are summed by every workgroup for itself (<= 512 values), and
// every slot writes the run of particles that take it as their ancestor.  Integer weights: the result does not depend on
// the order of the additions, so the ancestors are those of fixedWeightKernel + DeviceScan + ancestorKernel bit for bit
// (tests/test_gpu_pf.py holds both paths to the same oracle).
// RESIDENCY.  A workgroup that spins at a device-memory barrier holds its CU slot: if not every workgroup of the grid
// is resident the launch never ends.  The grid is therefore sized by the host from what the device can hold
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x the CUs, divided by the number of shards a node has put on the
// device: fusedBudget below) -- at most kFusedBlocks, and the multi-launch path when next to nothing fits -- and the
// barrier's poll has a budget: a workgroup that gives up marks the launch void (kPfVoid in the totals, the stuck word),
// poisons the barrier so that the others leave too, and exits.
#ifndef SIPNET_PF_BLOCKS
#define SIPNET_PF_BLOCKS 512
#endif
#ifndef SIPNET_PF_SLEEP
#define SIPNET_PF_SLEEP 2
#endif
#ifndef SIPNET_PF_SPIN_BUDGET
#define SIPNET_PF_SPIN_BUDGET (1 << 19)   // polls of ~0.5-1 us each: a few tenths of a second
#endif
constexpr int kFusedBlocks = SIPNET_PF_BLOCKS;   // (<= 512: phase 3 scans the chunk totals two per thread)
constexpr int kFusedMinBlocks = 8;               // fewer resident workgroups than this: the multi-launch path
constexpr int kFusedMaxPer = 16;                 // ... or more slots per thread than this (phase 3 adds a thread's weights up again)
constexpr long long kPfVoid = LLONG_MIN;         // "total weight" of a launch whose barrier gave up
#ifdef SIPNET_PF_STAMPS   // (probe, tools/pf_analysis_time.py: where the launch spends its time -- workgroup 0's clock at every phase)
__device__ unsigned long long g_pfStamps[8];
#define PF_STAMP(k)                                                                  \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
    unsigned long long now_;                                                         \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");  \
    g_pfStamps[k] = now_;                                                            \
  }
#else
#define PF_STAMP(k)
#endif
struct FusedArgs {
  // phase 1 (the one-batch analysis): log-weights from the forecast's plane
  const void* plane;
  int32_t nSteps;
  int64_t ld, ncol;
  const double* status;
  double obs, invSigma;
  double* logw;              // [nSlots] written by phase 1, read by phase 2
  // phase 2 over gathered blocks instead (the filter across ranks): slot i = gathered[(i / nmax) * stride + i % nmax],
  // the blocks' maxima behind each rank's nmax log-weights
  const double* gathered;
  int32_t world, nmax;
  int64_t stride;
  int64_t nSlots, chunk;
  double* blockMax;          // [gridDim.x]
  // phase 1 done already by the forecast's own launch (FastArgs::pfLogw): logw is filled, preMax[nPre] are partial maxima
  const double* preMax;
  int32_t nPre;
  const double* logwIn;      // [nSlots] phase 2's input in the one-batch analysis (= logw)
  int64_t* threadIncl;       // [gridDim.x][256] every thread's inclusive sum of weights inside its chunk
  int64_t* blockSum;         // [gridDim.x]
  unsigned long long* barrier;   // THIS launch's barrier set (kBarSetWords words, all zero when the launch starts)
  unsigned long long* barrierAhead;   // the set of the launch kBarAhead launches from now: zeroed by this one
  unsigned long long* stuck;     // diagnostics: 1 << 63 | barrier number << 32 | workgroup of the first poll that gave up
  int32_t spinBudget;
  int32_t absent;            // test hook (sipnet_debug_pf_barrier): this workgroup leaves at once, without arriving; -1: nobody
  // phase 3
  int64_t j0, nOut, nTotal;
  double u0;
  int32_t* anc;
  int64_t* total;            // may be null
  int64_t* totalScratch;     // always written
};
// What the workgroups exchange (chunk maxima, chunk sums) is written and read with agent-scope relaxed atomics: such
// accesses are coherent across the chip's eight XCDs (each has an L2 of its own) without cache maintenance.  The first
// version used plain accesses and release / acquire fences at the barriers: an agent-scope release is an L2 write-back,
// an acquire an L2 invalidate -- 2 048 waves x 2 barriers of them made the launch 131 us.  Ordering: a workgroup's
// stores have been acknowledged (vmcnt(0)) before it arrives.
__device__ __forceinline__ void stAgent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ldAgent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stAgent(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ldAgent(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The barrier itself.  Atomics of one address are served one after the other, ~18 ns each on this chip: 512 workgroups
// arriving at ONE counter made a barrier 9 us (measured with s_memrealtime stamps, profiles/r05_pf_analysis_variants.txt)
// -- more than a launch boundary.  Hence two levels: workgroups arrive at their GROUP's counter (kBarGroup of them per
// address); a group's last arrival goes on to the top counter; the top's last arrival releases every group through the
// group's own flag, which is what the group's workgroups poll (32 pollers per address instead of 512), 64 bytes apart.
// Round 6: every launch has a barrier SET of its own out of a ring of kBarSets (two barriers each, all words zero when
// the launch starts: launch L clears the set of launch L + kBarAhead, which nothing uses in between -- the launches of one
// scratch block are ordered by their stream).  Round 5's counters only ever grew across launches, which made every later
// launch depend on every earlier one having completed its barriers: a launch that failed to start, or a grid that was not
// co-resident, left the host's epoch ahead of the counters and the NEXT analysis spinning for ever.  Now a void launch
// spoils its own set only.
constexpr int kBarGroup = 32;
constexpr int kBarStride = 8;   // unsigned long longs between two counters: a line of their own
constexpr int kBarGroupsMax = (kFusedBlocks + kBarGroup - 1) / kBarGroup;
constexpr int kBarWords = kBarStride * (2 + 2 * kBarGroupsMax);   // one barrier: top counter, poison word, G counters, G flags
constexpr int kBarSetWords = 2 * kBarWords;                       // a launch passes at most two
constexpr int kBarSets = 64, kBarAhead = 32;
// false: the barrier gave up (this workgroup's poll ran out of budget, or another's did and poisoned the barrier) -- the
// launch is void and the caller returns; every thread of the workgroup gets the same answer
__device__ __forceinline__ bool gridBarrier(unsigned long long* bar, int which, const FusedArgs& a, int* smOk) {
  // bar[0]: top counter; bar[kBarStride]: poison; bar[kBarStride * (2 + g)]: group g's counter; bar[kBarStride * (2 + G + g)]: its flag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nb = (int)gridDim.x, G = (nb + kBarGroup - 1) / kBarGroup, g = (int)blockIdx.x / kBarGroup;
    const int inGroup = (g == G - 1) ? nb - g * kBarGroup : kBarGroup;
    unsigned long long* flag = bar + kBarStride * (2 + G + g);
    unsigned long long* poison = bar + kBarStride;
    const unsigned long long arrived = __hip_atomic_fetch_add(bar + kBarStride * (2 + g), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived + 1 == (unsigned long long)inGroup) {
      const unsigned long long t = __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 == (unsigned long long)G)
        for (int k = 0; k < G; k++)
          __hip_atomic_store(bar + kBarStride * (2 + G + k), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int ok = 1, polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
      __builtin_amdgcn_s_sleep(SIPNET_PF_SLEEP);
      if ((++polls & 63) == 0) {
        if (__hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { ok = 0; break; }
        if (polls >= a.spinBudget) {
          __hip_atomic_store(poison, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long rep = (1ull << 63) | ((unsigned long long)(unsigned)which << 32) | (unsigned)blockIdx.x;
          atomicCAS(a.stuck, 0ull, rep);
          ok = 0;
          break;
        }
      }
    }
    if (!ok) {   // the launch is void: say so where the host looks for the total weight
      if (a.total) __hip_atomic_store((long long*)a.total, kPfVoid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((long long*)a.totalScratch, kPfVoid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *smOk = ok;
  }
  __syncthreads();
  return *smOk != 0;
}
__device__ __forceinline__ double blockMax256(double v, double* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}
// inclusive sum over the 256 threads of a workgroup; *totalOut = the sum of all
__device__ __forceinline__ long long blockScan256(long long v, long long* smWave, long long* totalOut) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const long long o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  if (lane == 63) smWave[wave] = v;
  __syncthreads();
  long long before = 0;
  for (int k = 0; k < wave; k++) before += smWave[k];
  *totalOut = smWave[0] + smWave[1] + smWave[2] + smWave[3];
  __syncthreads();
  return v + before;
}
template <typename T, bool Gathered>
__global__ __launch_bounds__(256) void pfFusedKernel(FusedArgs a) {
  __shared__ double smD[256];
  __shared__ long long smWave[4];
  __shared__ long long prefix[kFusedBlocks + 1];
  __shared__ int smOk;
  const int tid = (int)threadIdx.x, nb = (int)gridDim.x, b = (int)blockIdx.x;
  const int64_t lo = (int64_t)b * a.chunk, hi = lo + a.chunk < a.nSlots ? lo + a.chunk : a.nSlots;
  // this thread's own consecutive slots [t0, t1)
  const int per = (int)(a.chunk >> 8);
  const int64_t t0 = lo + (int64_t)tid * per < hi ? lo + (int64_t)tid * per : hi, t1 = t0 + per < hi ? t0 + per : hi;
  int nBarrier = 0;
  double m = -INFINITY;
  PF_STAMP(0)
  // the barrier set of the launch kBarAhead launches from now (gridBarrier)
  if (b == 0)
    for (int k = tid; k < kBarSetWords; k += 256) a.barrierAhead[k] = 0ull;
  if (b == a.absent) return;
  if (!Gathered && a.preMax) {
    // ---- phase 1 was the forecast kernel's epilogue: only the maximum is left to take ----
    double pm = -INFINITY;
    for (int k = tid; k < a.nPre; k += 256) pm = fmax(pm, a.preMax[k]);
    m = blockMax256(pm, smD);
  } else if (!Gathered) {
    // ---- phase 1: this chunk's log-weights and their maximum (lanes on neighbouring columns of the plane) ----
    double mine = -INFINITY;
    for (int64_t i = lo + tid; i < hi; i += 256)
      mine = fmax(mine, logWeightOf<T, true>((const T*)a.plane, a.nSteps, a.ld, i, a.status, a.obs, a.invSigma, a.logw));
    mine = blockMax256(mine, smD);
    if (tid == 0) stAgent(&a.blockMax[b], mine);
    PF_STAMP(1)
    if (!gridBarrier(a.barrier + kBarWords * nBarrier, nBarrier, a, &smOk)) return;
    nBarrier++;
    PF_STAMP(2)
    double pm = -INFINITY;
    for (int k = tid; k < nb; k += 256) pm = fmax(pm, ldAgent(&a.blockMax[k]));
    m = blockMax256(pm, smD);
  } else {
    // the maximum over every rank's block maxima (512 workgroups read the same world x P doubles at the same time: each starts
    // with another rank's, and no division in the index -- 6.5 us at 8 x 512 maxima before, profiles/r06_pf_analysis_phases.txt)
    const int P = (a.nmax + 255) / 256;
    double pm = -INFINITY;
    for (int q = 0; q < a.world; q += 4) {   // (four ranks' loads in flight: one after the other they were 16 L2 round trips)
      const double* mx[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int r = (q + u < a.world ? q + u : q) + b % a.world;
        r = r >= a.world ? r - a.world : r;
        mx[u] = a.gathered + (int64_t)r * a.stride + a.nmax;
      }
      for (int k = tid; k < P; k += 256) {
        const double v0 = mx[0][k], v1 = mx[1][k], v2 = mx[2][k], v3 = mx[3][k];
        pm = fmax(fmax(pm, fmax(v0, v1)), fmax(v2, v3));
      }
    }
    m = blockMax256(pm, smD);
    PF_STAMP(2)
  }
  // ---- phase 2: fixed-point weights (pfFixedWeight) of this thread's slots, summed; ONE block scan ----
  // slot i's log-weight, wherever it lies: the one-batch analysis' own vector, or rank (i / nmax)'s gathered block
  auto slotLogw = [&](int64_t i) -> double {
    if (!Gathered) return a.logwIn[i];
    const int64_t r = i / a.nmax;
    return a.gathered[r * a.stride + (i - r * a.nmax)];
  };
  long long mySum = 0;
  {
    int64_t r = 0, c = 0;   // (gathered blocks: slot i sits in rank r's block at column c)
    if (Gathered) { r = t0 / a.nmax; c = t0 - r * a.nmax; }
    for (int64_t i = t0; i < t1; i += 8) {
      double lw[8];
      const int n8 = t1 - i < 8 ? (int)(t1 - i) : 8;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        lw[k] = -INFINITY;
        if (k < n8) {
          if (Gathered) {
            lw[k] = a.gathered[r * a.stride + c];
            if (++c == a.nmax) { c = 0; r++; }
          } else {
            lw[k] = a.logwIn[i + k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) mySum += pfFixedWeight(lw[k], m);
    }
  }
  long long chunkTotal;
  const long long myIncl = blockScan256(mySum, smWave, &chunkTotal);
  // (the threads' inclusive sums inside the chunk: what another workgroup needs to place a particle in this chunk)
  stAgent((long long*)&a.threadIncl[(int64_t)b * 256 + tid], myIncl);
  if (tid == 0) stAgent((long long*)&a.blockSum[b], chunkTotal);
  PF_STAMP(3)
  if (!gridBarrier(a.barrier + kBarWords * nBarrier, nBarrier, a, &smOk)) return;
  nBarrier++;
  PF_STAMP(4)
  // ---- phase 3: the chunks' offsets (every workgroup for itself), then the ancestors of this workgroup's PARTICLES ----
  {
    // two entries per thread (nb <= 512), scanned as pairs
    const int k0 = 2 * tid, k1 = 2 * tid + 1;
    const long long s0 = k0 < nb ? ldAgent((const long long*)&a.blockSum[k0]) : 0, s1 = k1 < nb ? ldAgent((const long long*)&a.blockSum[k1]) : 0;
    long long all;
    const long long inc = blockScan256(s0 + s1, smWave, &all);
    if (tid == 0) prefix[0] = 0;
    if (k0 < nb) prefix[k0 + 1] = inc - s1;
    if (k1 < nb) prefix[k1 + 1] = inc;
    __syncthreads();
  }
  const long long Sll = prefix[nb];
  PF_STAMP(5)
  // ancestorKernel's rule, particle by particle: particle g takes the first slot i with cdf[i] > P(g), P(g) = min(((g + u0) S) /
  // nTotal, S - 1).  The launch's particles [j0, j0 + nOut) are dealt to the workgroups in equal contiguous shares -- round 5 and
  // the first round-6 version went slot by slot, every slot writing the run of particles that take it: across ranks a launch
  // writes only ITS rank's particles, whose slots sit in 1 / world of the chunks, so 64 of 512 workgroups did all the divisions and
  // stores of the phase (8 x 131 072 slots: ~15 us against 2.7 us for one rank's).  A particle finds its slot in three steps that
  // read nothing but sums: its CHUNK by bisection of the chunks' offsets (LDS), the THREAD of phase 2 whose slots hold it by
  // bisection of that chunk's 256 inclusive sums (LDS for the two chunks the workgroup's particles start in, device memory
  // for a particle further on), the SLOT by adding up that thread's <= 16 weights again (pfFixedWeight of log-weights that were
  // there before the launch, or came from phase 1 through agent-scope stores).  cdf is non-decreasing, so "first slot with
  // cdf > p" never lands on a slot that weighs nothing; S = 0 puts every particle on slot 0, as ancestorKernel does.
  const double S = (double)Sll, nTot = (double)a.nTotal;
  __shared__ long long inclA[256], inclB[256];
  __shared__ int chunkA;
  const int64_t share = (a.nOut + nb - 1) / nb;           // particles per workgroup
  const int64_t jLo = (int64_t)b * share, jHi = jLo + share < a.nOut ? jLo + share : a.nOut;
  auto chunkOf = [&](double p) -> int {                   // first c with (double)prefix[c + 1] > p
    int lo2 = 0, hi2 = nb - 1;
    while (lo2 < hi2) {
      const int mid = (lo2 + hi2) >> 1;
      if ((double)prefix[mid + 1] > p) hi2 = mid; else lo2 = mid + 1;
    }
    return lo2;
  };
  auto pOf = [&](int64_t j) -> double { return fmin((((double)(a.j0 + j) + a.u0) * S) / nTot, S - 1.0); };
  if (jLo < jHi) {
    if (tid == 0) chunkA = chunkOf(pOf(jLo));
    __syncthreads();
    const int cA = chunkA, cB = cA + 1 < nb ? cA + 1 : cA;
    inclA[tid] = ldAgent((const long long*)&a.threadIncl[(int64_t)cA * 256 + tid]);
    inclB[tid] = ldAgent((const long long*)&a.threadIncl[(int64_t)cB * 256 + tid]);
    __syncthreads();
    for (int64_t j = jLo + tid; j < jHi; j += 256) {
      const double p = pOf(j);
      const int c = chunkOf(p);
      const long long base = prefix[c];
      // the thread of phase 2: first t with (double)(base + incl[c][t]) > p  (incl[c][255] = the chunk's total: exists)
      int tl = 0, th = 255;
      if (c == cA || c == cB) {
        const long long* incl = c == cA ? inclA : inclB;
        while (tl < th) {
          const int mid = (tl + th) >> 1;
          if ((double)(base + incl[mid]) > p) th = mid; else tl = mid + 1;
        }
      } else {
        while (tl < th) {
          const int mid = (tl + th) >> 1;
          if ((double)(base + ldAgent((const long long*)&a.threadIncl[(int64_t)c * 256 + mid])) > p) th = mid; else tl = mid + 1;
        }
      }
      long long run = base;
      if (tl > 0) run += (c == cA) ? inclA[tl - 1] : (c == cB) ? inclB[tl - 1] : ldAgent((const long long*)&a.threadIncl[(int64_t)c * 256 + tl - 1]);
      // the slot: that thread's weights once more, until the sum passes p (eight log-weights requested at a time: one after
      // the other they were up to `per` L2 round trips per particle)
      const int64_t i0 = (int64_t)c * a.chunk + (int64_t)tl * per;
      const int64_t iEnd = i0 + per < a.nSlots ? i0 + per : a.nSlots;
      int64_t found = -1;
      for (int64_t i = i0; i < iEnd && found < 0; i += 8) {
        double lw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) lw[k] = i + k < iEnd ? slotLogw(i + k) : -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          run += pfFixedWeight(lw[k], m);
          if (found < 0 && (double)run > p) found = i + k;
        }
      }
      a.anc[j] = (int32_t)(found < 0 ? iEnd - 1 : found);
    }
  }
  // the total weight, last: a workgroup that gave up at a barrier has written kPfVoid there, and a launch is void as soon as
  // one did (gridBarrier) -- its poison word says so even if this workgroup was released in the same instant
  if (b == 0 && tid == 0) {
    bool spoilt = false;
    for (int k = 0; k < nBarrier; k++)
      spoilt = spoilt || __hip_atomic_load(a.barrier + kBarWords * k + kBarStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    const long long tot = spoilt ? kPfVoid : Sll;
    if (a.total) *a.total = tot;
    *a.totalScratch = tot;
  }
  PF_STAMP(6)
}

// dst[row][j] = matrix of rank (anc[j] / nmax)[row][anc[j] % nmax] for the three matrices of a checkpoint
struct PeerPart {
  void* dst;
  int32_t rows, group0, elem4;
};
struct PeerParts {
  PeerPart p[3];
  int32_t n;
};
template <typename T>
__device__ __forceinline__ void gatherPeerRows(const T* __restrict__ p, int64_t srcPitch, T* __restrict__ q,
                                               int64_t dstPitch, int nr) {
  T v[kGatherRows];
  if (nr == kGatherRows) {
#pragma unroll
    for (int r = 0; r < kGatherRows; r++) v[r] = p[(int64_t)r * srcPitch];
#pragma unroll
    for (int r = 0; r < kGatherRows; r++) q[(int64_t)r * dstPitch] = v[r];
  } else {
    for (int r = 0; r < nr; r++) v[r] = p[(int64_t)r * srcPitch];
    for (int r = 0; r < nr; r++) q[(int64_t)r * dstPitch] = v[r];
  }
}
__global__ __launch_bounds__(256) void gatherPeerKernel(PeerParts parts, PeerPtrs peers,
                                                        const int32_t* __restrict__ anc, int64_t nOut,
                                                        int64_t dstPitch) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nOut) return;
  int k = 0;
  for (int q = 1; q < parts.n; q++)
    if ((int)blockIdx.y >= parts.p[q].group0) k = q;
  const PeerPart part = parts.p[k];
  const int row0 = ((int)blockIdx.y - part.group0) * kGatherRows;
  const int nr = part.rows - row0 < kGatherRows ? part.rows - row0 : kGatherRows;
  int32_t a = anc[j];   // (clamped like gatherMemberKernel's: a void analysis leaves the caller's buffer as it was)
  a = a < 0 ? 0 : a;
  int s = a / peers.nmax;
  s = s >= peers.world ? peers.world - 1 : s;
  const int64_t pitch = peers.pitch[s];
  int64_t c = a - s * peers.nmax;
  c = c >= pitch ? pitch - 1 : c;
  const void* base = k == 0 ? (const void*)peers.state[s] : k == 1 ? peers.ring[s] : peers.third[s];
  if (blockIdx.y == 0 && peers.crossing) {   // how many of this rank's particles crossed a link (sipnet_batch_pf_info)
    const unsigned long long far = __ballot(s != peers.rank);
    if ((threadIdx.x & 63) == 0 && far) atomicAdd(peers.crossing, (unsigned long long)__popcll(far));
  }
  if (part.elem4)
    gatherPeerRows<float>((const float*)base + (int64_t)row0 * pitch + c, pitch, (float*)part.dst + (int64_t)row0 * dstPitch + j,
                          dstPitch, nr);
  else
    gatherPeerRows<double>((const double*)base + (int64_t)row0 * pitch + c, pitch,
                           (double*)part.dst + (int64_t)row0 * dstPitch + j, dstPitch, nr);
}

// 1 when any member needs the generic-exponent kernel variant (dVpdExp != 2 or
// soilRespMoistEffect != 1), see engine.hip set_params
__global__ __launch_bounds__(256) void exponentCheckKernel(const double* __restrict__ prm,
                                                           int64_t ncol, int32_t* __restrict__ flag) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  if (prm[(int64_t)SP_dVpdExp * ncol + c] != 2.0 ||
      prm[(int64_t)SP_soilRespMoistEffect * ncol + c] != 1.0)
    atomicOr(flag, 1);
}

// ---- exchange plan of a resampling over `world` ranks with n particles each ----------------
// The global ancestor vector (identical on every rank, non-decreasing) is cut into destination
// blocks [d*n, (d+1)*n); inside a block the ancestors owned by source rank s (anc / n == s) are
// contiguous.  A particle that has to cross ranks (s != d) travels ONCE per destination: the
// first of a run of equal ancestors inside a destination block is its "head".
//   first[d][s]  first index of block d whose ancestor belongs to rank >= s   (binary search)
//   head[i]      1 when entry i is a cross-rank head                          (exclusive scan -> P)
//   count[d][s]  = P[first[d][s+1]] - P[first[d][s]]                          (columns d receives from s)
constexpr int kMaxWorld = 64;
__global__ void planFirstKernel(const int32_t* __restrict__ anc, int64_t n, int32_t world,
                                int64_t* __restrict__ first) {
  const int d = blockIdx.x, s = threadIdx.x;  // s in 0..world
  if (s > world) return;
  const int64_t target = (int64_t)s * n;      // first ancestor value owned by rank s
  int64_t lo = (int64_t)d * n, hi = lo + n;   // lower bound of `target` in anc[lo, hi)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)anc[mid] < target) lo = mid + 1; else hi = mid;
  }
  first[(int64_t)d * (world + 1) + s] = lo;
}
// (also validates the vector: every ancestor inside [0, total) and non-decreasing -- anything else
// (NaN weights upstream, a caller's bug) raises *bad and the plan's indices are never used)
__global__ __launch_bounds__(256) void planHeadKernel(const int32_t* __restrict__ anc, int64_t n,
                                                      int64_t total, int32_t* __restrict__ head,
                                                      int32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t a = anc[i];
  if (a < 0 || a >= total || (i > 0 && anc[i - 1] > a)) {
    atomicOr(bad, 1);
    head[i] = 0;
    return;
  }
  const int64_t d = i / n, s = a / n;
  const bool newRun = (i % n == 0) || anc[i - 1] != a;
  head[i] = (newRun && s != d) ? 1 : 0;
}
// one block: counts, the bases of this rank's send / receive blocks (in rank order, self skipped)
__global__ void planCountKernel(const int64_t* __restrict__ first, const int32_t* __restrict__ P,
                                const int32_t* __restrict__ head, int64_t total, int32_t world,
                                int32_t rank, int64_t* __restrict__ counts /* [2][world]: send, recv */,
                                int64_t* __restrict__ bases /* [2][world] */) {
  if (threadIdx.x != 0) return;
  if (counts[2 * kMaxWorld * 2] != 0) {   // (the validity flag lives behind the counts and bases)
    for (int q = 0; q < 2 * world; q++) counts[q] = 0, bases[q] = 0;
    return;
  }
  auto Pat = [&](int64_t i) -> int64_t { return i < total ? (int64_t)P[i] : (int64_t)P[total - 1] + head[total - 1]; };
  int64_t sb = 0, rb = 0;
  for (int q = 0; q < world; q++) {
    const int64_t* fs = first + (int64_t)q * (world + 1);       // destination q, what I (rank) send it
    const int64_t send = q == rank ? 0 : Pat(fs[rank + 1]) - Pat(fs[rank]);
    const int64_t* fr = first + (int64_t)rank * (world + 1);    // my block, what comes from source q
    const int64_t recv = q == rank ? 0 : Pat(fr[q + 1]) - Pat(fr[q]);
    counts[q] = send;
    counts[world + q] = recv;
    bases[q] = sb;
    bases[world + q] = rb;
    sb += send;
    rb += recv;
  }
}
__global__ __launch_bounds__(256) void planFillKernel(const int32_t* __restrict__ anc, int64_t n,
                                                      int64_t total, int32_t world, int32_t rank,
                                                      const int64_t* __restrict__ first,
                                                      const int32_t* __restrict__ P,
                                                      const int32_t* __restrict__ head,
                                                      const int64_t* __restrict__ bases,
                                                      int32_t* __restrict__ sendCols,
                                                      int32_t* __restrict__ src, const int32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (*bad) return;
  const int64_t a = anc[i];
  const int64_t d = i / n, s = a / n, lo = (int64_t)rank * n;
  if (s == rank && d != rank && head[i]) {   // a column of mine that rank d needs (once)
    const int64_t f = first[d * (world + 1) + rank];
    sendCols[bases[d] + ((int64_t)P[i] - (int64_t)P[f])] = (int32_t)(a - lo);
  }
  if (d == rank) {                           // where my new column j comes from
    const int64_t j = i - lo;
    if (s == rank) {
      src[j] = (int32_t)(a - lo);
    } else {
      const int64_t f = first[(int64_t)rank * (world + 1) + s];
      const int64_t k = ((int64_t)P[i] + head[i] - 1) - (int64_t)P[f];   // index among the heads from s
      src[j] = (int32_t)(n + bases[world + s] + k);
    }
  }
}

// state + ring (+ parameters) of the columns src[0..nOut) in one launch
// (ringF32: the ring rows are floats, in the batch and in a packed block, where they take SIPNET_RING_SLOTS / 2 rows of words)
// prmRemap: the batch's parameter index (null: parameters in column order) -- the parameter rows are read through it.
// idOld / idNew (resampling with an index instead of the parameter rows; then prm must be null): idNew[j] = idOld[src[j]].
void launchGatherMember(const double* state, const void* ring, bool ringF32, const double* prm, int64_t ncol,
                        const double* recv, const RecvMap& map, const int32_t* src, int64_t nOut,
                        double* dState, void* dRing, double* dPrm, int64_t dstPitch, hipStream_t stream,
                        const int32_t* prmRemap = nullptr, const int32_t* idOld = nullptr, int32_t* idNew = nullptr) {
  // (ncol: the leading dimension of the source matrices AND the number of own source columns)
  if (nOut <= 0) return;
  auto groups = [](int rows) { return (rows + kGatherRows - 1) / kGatherRows; };
  GatherParts parts{};
  parts.n = 0;
  int total = 0;
  if (state) {
    parts.p[parts.n++] = GatherPart{state, dState, SIPNET_NSTATE, total, 0, 0, nullptr};
    total += groups(SIPNET_NSTATE);
    parts.p[parts.n++] = GatherPart{ring, dRing, SIPNET_RING_SLOTS, total, SIPNET_NSTATE, ringF32 ? 1 : 0, nullptr};
    total += groups(SIPNET_RING_SLOTS);
  }
  if (prm) {
    parts.p[parts.n++] = GatherPart{prm, dPrm, SIPNET_NPARAMS, total, SIPNET_NSTATE + ringWords(ringF32), 0, prmRemap};
    total += groups(SIPNET_NPARAMS);
  } else if (idOld) {   // one row of 4-byte elements
    parts.p[parts.n++] = GatherPart{idOld, idNew, 1, total, 0, 1, nullptr};
    total += 1;
  }
  dim3 grid((unsigned)((nOut + 255) / 256), (unsigned)total);
  hipLaunchKernelGGL(gatherMemberKernel, grid, dim3(256), 0, stream, parts, ncol, ncol, recv, map, src, nOut, dstPitch);
}

__global__ __launch_bounds__(256) void iotaKernel(int32_t* p, int64_t n, int32_t first = 0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = first + (int32_t)i;
}
// rows of doubles from a peer's matrix (pitch srcPitch) into columns col0.. of the bank (sipnet_batch_pf_connect)
__global__ __launch_bounds__(256) void copyRowsKernel(double* __restrict__ dst, int64_t dstPitch, const double* __restrict__ src,
                                                      int64_t srcPitch, int64_t width, int32_t rows) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) dst[(int64_t)r * dstPitch + c] = src[(int64_t)r * srcPitch + c];
}

}  // namespace
}  // namespace sipnet

using namespace sipnet;

// Host side of PeerPtrs: for each of the two buffers a batch's checkpoint matrices alternate between (a resampling
// gathers into the spare and swaps), where every rank keeps them.  All ranks resample in lockstep, so the local
// parity says which buffer is current everywhere.
struct PfPeers {
  int32_t world = 1, rank = 0, nmax = 0, withParams = 0;
  int64_t nTotal = 0, first = 0;       // particles of all ranks; global index of this rank's first particle
  int32_t count[kMaxPeers] = {};
  const double* state[2][kMaxPeers] = {};
  const void* ring[2][kMaxPeers] = {};
  const double* prm[2][kMaxPeers] = {};
  const int32_t* ids[2][kMaxPeers] = {};   // byIndex: the particles' index into the replicated parameter bank
  bool byIndex = false;                // all ranks' parameters are in sipnet_batch::d_prmBank; particles carry an index
  bool bankLost = false;               // ... and this batch has since changed its parameters or moved rows: connect again
  std::vector<void*> opened;           // hipIpcOpenMemHandle mappings, closed on release
  int parity = 0;
};

// the connection's parameter bank is no longer what the particles' indices mean (new parameters were set, or a resampling
// moved parameter ROWS): the batch goes back to its own column-order block
void pfDropBank(sipnet_batch* b) {
  if (!b->d_prmBank) return;
  (void)hipDeviceSynchronize();
  (void)hipFree(b->d_prmBank);
  b->d_prmBank = nullptr;
  b->prmBankPitch = 0;
  b->prmIndexed = false;
  if (b->pfPeers) b->pfPeers->bankLost = true;
}

// The parameter bank back in column order (batch_impl.h): gather through the index into the spare, swap.
int materializeParams(sipnet_batch* b, hipStream_t stream) {
  if (!b->prmIndexed) return SIPNET_OK;
  const size_t nc = (size_t)b->ncol;
  {   // (the last launch may have run on another stream)
    int rcO = orderBehindBusy(b, stream);
    if (rcO) return rcO;
  }
  RecvMap none{};
  if (b->d_prmBank) {   // a connected filter: the rows out of the bank of all ranks' parameters; the index stays what it is
    launchGatherMember(nullptr, nullptr, false, b->d_prmBank, b->prmBankPitch, nullptr, none, b->d_prmId, b->ncol, nullptr, nullptr,
                       b->d_prm, b->ncol, stream);
    HIP_TRY(hipGetLastError());
    b->prmIndexed = false;
    return markBusy(b, stream);
  }
  if (!b->d_prm2) HIP_TRY(hipMalloc(&b->d_prm2, nc * SIPNET_NPARAMS * sizeof(double)));
  launchGatherMember(nullptr, nullptr, false, b->d_prm, b->ncol, nullptr, none, b->d_prmId, b->ncol, nullptr, nullptr, b->d_prm2,
                     b->ncol, stream);   // dst column j <- bank column d_prmId[j]
  HIP_TRY(hipGetLastError());
  std::swap(b->d_prm, b->d_prm2);
  b->prmIndexed = false;
  return markBusy(b, stream);
}

#ifdef SIPNET_PF_STAMPS
extern "C" int sipnet_debug_read_pf_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfStamps), 8 * sizeof(unsigned long long));
}
#endif

extern "C" {

int32_t sipnet_pf_member_words(int32_t with_params) {
  return SIPNET_NSTATE + SIPNET_RING_SLOTS + (with_params ? SIPNET_NPARAMS : 0);
}

int32_t sipnet_batch_member_words(const sipnet_batch* b, int32_t with_params) {
  if (!b) return -1;
  return SIPNET_NSTATE + ringWords(b->precision == SIPNET_F32_MIXED) + (with_params ? SIPNET_NPARAMS : 0);
}

// d_part (optional, DEVICE, one double per 256 columns): the blocks' maxima, for the resampling that follows
// (npad >= ncol: log-weight slots to fill, the ones past the batch's particles with -inf)
static int logWeights(sipnet_batch* b, const void* d_plane, int32_t elem_is_f32, int32_t n_steps, int64_t ld,
                      double obs, double sigma, double* d_logw, double* d_part, void* hip_stream, int64_t npad = 0) {
  if (!b || !d_plane || !d_logw || n_steps <= 0 || ld < b->ncol || !(sigma > 0)) {
    setError("sipnet_batch_pf_log_weights: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  if (npad < b->ncol) npad = b->ncol;
  const int grid = (int)((npad + 255) / 256);
  const double* status = b->d_state + (size_t)ST_status * b->ncol;
  if (elem_is_f32) {
    hipLaunchKernelGGL(logWeightKernel<float>, dim3(grid), dim3(256), 0, stream,
                       (const float*)d_plane, n_steps, ld, b->ncol, status, obs, 1.0 / sigma, d_logw, d_part, npad);
  } else {
    hipLaunchKernelGGL(logWeightKernel<double>, dim3(grid), dim3(256), 0, stream,
                       (const double*)d_plane, n_steps, ld, b->ncol, status, obs, 1.0 / sigma, d_logw, d_part, npad);
  }
  HIP_TRY(hipGetLastError());
  return SIPNET_OK;
}

extern "C" int sipnet_batch_pf_log_weights(sipnet_batch* b, const void* d_plane, int32_t elem_is_f32,
                                           int32_t n_steps, int64_t ld, double obs, double sigma,
                                           double* d_logw, void* hip_stream) {
  return logWeights(b, d_plane, elem_is_f32, n_steps, ld, obs, sigma, d_logw, nullptr, hip_stream);
}

// scratch of the resampling, kept between calls: one per batch for the sipnet_batch_pf_* entry points (freed by
// sipnet_batch_destroy), one per host thread for the batch-less sipnet_pf_* ones (sipnet_pf_release_scratch)
struct PfScratch {
  int device = -1;
  int64_t cap = 0;
  double* d_max = nullptr;  // partial maxima of the log-weights (one per 256 weights at most)
  int64_t* d_w = nullptr;
  int64_t* d_cdf = nullptr;
  void* d_tmp = nullptr;
  size_t tmpBytes = 0;
  // the one-launch analysis (pfFusedKernel): chunk totals + the total weight, the ring of per-launch barrier sets (all
  // zero at allocation; launch L uses set L % kBarSets and clears set (L + kBarAhead) % kBarSets), the stuck report
  int64_t* d_blockSum = nullptr;      // [kFusedBlocks] + 1: the total
  int64_t* d_threadIncl = nullptr;    // [kFusedBlocks][256]
  unsigned long long* d_barrier = nullptr;   // [kBarSets][kBarSetWords] + 1: the stuck word
  unsigned long long launches = 0;      // fused launches that were accepted by the runtime
  int occ[3] = {-1, -1, -1};            // resident workgroups per CU of pfFusedKernel<float,false> / <double,false> / <double,true>
  void release() {
    if (d_max) (void)hipFree(d_max);
    if (d_w) (void)hipFree(d_w);
    if (d_cdf) (void)hipFree(d_cdf);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_blockSum) (void)hipFree(d_blockSum);
    if (d_threadIncl) (void)hipFree(d_threadIncl);
    if (d_barrier) (void)hipFree(d_barrier);
    d_threadIncl = nullptr;
    d_max = nullptr; d_w = d_cdf = nullptr; d_tmp = nullptr; cap = 0; tmpBytes = 0;
    d_blockSum = nullptr; d_barrier = nullptr; launches = 0;
    occ[0] = occ[1] = occ[2] = -1;
  }
  // no destructor: a thread_local's would run at thread exit, possibly after the HIP runtime is gone
};
namespace {
constexpr int kMaxParts = 256;
constexpr size_t kBarrierWords = (size_t)kBarSets * kBarSetWords + 1;   // + the stuck word
thread_local PfScratch g_pf;
}  // namespace

// scratch for n weights (the partial maxima: one per 256 of them at most)
static int pfScratchFor(PfScratch& sc, int64_t n, hipStream_t stream) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (sc.device != dev || sc.cap < n) {
    sc.release();
    sc.device = dev;
    HIP_TRY(hipMalloc(&sc.d_max, (size_t)((n + 255) / 256 + kMaxParts) * sizeof(double)));
    HIP_TRY(hipMalloc(&sc.d_w, (size_t)n * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&sc.d_cdf, (size_t)n * sizeof(int64_t)));
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, sc.tmpBytes, sc.d_w, sc.d_cdf, (int)n, stream));
    HIP_TRY(hipMalloc(&sc.d_tmp, sc.tmpBytes));
    HIP_TRY(hipMalloc(&sc.d_blockSum, (size_t)(kFusedBlocks + 1) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&sc.d_threadIncl, (size_t)kFusedBlocks * 256 * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&sc.d_barrier, kBarrierWords * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(sc.d_barrier, 0, kBarrierWords * sizeof(unsigned long long), stream));
    sc.launches = 0;
    sc.cap = n;
  }
  return SIPNET_OK;
}
// How many workgroups of the one-launch analysis may spin at its barriers at once: what the device holds
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x its CUs -- the batch's numCUs: a partitioned device reports its own) over
// the number of filters that may analyse on this device at the same time (sipnet_batch_set_device_share: a node's shards on
// one device), at most kFusedBlocks.  which: 0 pfFusedKernel<float, false>, 1 <double, false>, 2 <double, true>.
static int fusedBudget(PfScratch& sc, const sipnet_batch* b, int which) {
  if (sc.occ[which] < 0) {
    int occ = 0;
    hipError_t e = which == 0   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pfFusedKernel<float, false>, 256, 0)
                   : which == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pfFusedKernel<double, false>, 256, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pfFusedKernel<double, true>, 256, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      occ = 0;   // (unknown: the multi-launch path)
    }
    sc.occ[which] = occ;
  }
  const int64_t share = b->deviceShare > 0 ? b->deviceShare : 1;
  const int64_t fit = (int64_t)sc.occ[which] * b->numCUs / share;
  return (int)(fit < kFusedBlocks ? fit : kFusedBlocks);
}
// may the analysis over nSlots weights be ONE launch of at most `budget` resident workgroups?
static bool fusable(int64_t nSlots, int budget) {
  return budget >= kFusedMinBlocks && (nSlots + 255) / 256 <= (int64_t)budget * kFusedMaxPer;
}
// geometry of the one-launch analysis over nSlots weights with at most `budget` workgroups: contiguous chunks of whole tiles
static void fusedGeometry(int64_t nSlots, int budget, int* grid, int64_t* chunk) {
  const int64_t tiles = (nSlots + 255) / 256;
  const int64_t nb = tiles < budget ? tiles : budget;
  const int64_t tilesPer = (tiles + nb - 1) / nb;
  *chunk = tilesPer * 256;
  *grid = (int)((nSlots + *chunk - 1) / *chunk);
}
// this launch's barrier set and the one it clears for a later launch; fusedLaunched() once the runtime has accepted the launch
static void fusedBarrier(PfScratch& sc, FusedArgs* fa, int spinBudget) {
  fa->barrier = sc.d_barrier + (size_t)(sc.launches % kBarSets) * kBarSetWords;
  fa->barrierAhead = sc.d_barrier + (size_t)((sc.launches + kBarAhead) % kBarSets) * kBarSetWords;
  fa->stuck = sc.d_barrier + (size_t)kBarSets * kBarSetWords;
  fa->spinBudget = spinBudget > 0 ? spinBudget : SIPNET_PF_SPIN_BUDGET;
}
// a launch the runtime refused has not touched its set (nor cleared the one ahead): the next one takes the same set
static int fusedLaunched(PfScratch& sc) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    setError(std::string("pfFusedKernel: ") + hipGetErrorString(e));
    return SIPNET_ERR_INTERNAL;
  }
  sc.launches++;
  return SIPNET_OK;
}
// the report a void launch left (gridBarrier), for the error message; clears it
static std::string fusedStuckReport(PfScratch& sc, hipStream_t stream) {
  unsigned long long rep = 0;
  unsigned long long* d = sc.d_barrier + (size_t)kBarSets * kBarSetWords;
  if (hipMemcpyAsync(&rep, d, sizeof rep, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
    (void)hipGetLastError();
    return "the analysis kernel's grid barrier gave up";
  }
  (void)hipMemsetAsync(d, 0, sizeof rep, stream);
  return "the analysis kernel's grid barrier gave up (barrier " + std::to_string((unsigned)((rep >> 32) & 0x7fffffffu)) + ", workgroup " +
         std::to_string((unsigned)(rep & 0xffffffffu)) + "): its workgroups were not all resident -- another kernel held the device, or more filters "
         "analyse on it at once than sipnet_batch_set_device_share says; the launch's results are void";
}
static PfScratch& scratchOf(sipnet_batch* b) {
  if (!b->pfScratch) b->pfScratch = new PfScratch();
  return *b->pfScratch;
}

// partsGiven > 0: the partial maxima of d_logw are in the scratch block already (logWeights put them there)
// (partPtr: where those maxima are, when not in the scratch block)
static int ancestorsImpl(PfScratch& sc, const double* d_logw, int64_t n, double u0, int32_t* d_ancestors,
                         int64_t* d_fixed_weights, int64_t* d_total, int partsGiven, void* hip_stream,
                         const double* partPtr = nullptr) {
  if (!d_logw || !d_ancestors || n <= 0 || n > (int64_t)1 << 22 || !(u0 >= 0.0) || !(u0 < 1.0)) {
    setError("sipnet_pf_systematic_ancestors: bad argument (n <= 4194304, 0 <= u0 < 1)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  int rc = pfScratchFor(sc, n, stream);
  if (rc) return rc;
  const int grid = (int)((n + 255) / 256);
  int parts = partsGiven;
  if (parts <= 0) {
    parts = grid < kMaxParts ? grid : kMaxParts;
    hipLaunchKernelGGL(maxPartialKernel, dim3(parts), dim3(256), 0, stream, d_logw, n, sc.d_max);
  }
  hipLaunchKernelGGL(fixedWeightKernel, dim3(grid), dim3(256), 0, stream, d_logw, n, partPtr ? partPtr : sc.d_max, parts, sc.d_w);
  size_t tmpBytes = sc.tmpBytes;
  HIP_TRY(hipcub::DeviceScan::InclusiveSum(sc.d_tmp, tmpBytes, sc.d_w, sc.d_cdf, (int)n, stream));
  hipLaunchKernelGGL(ancestorKernel, dim3(grid), dim3(256), 0, stream, sc.d_cdf, n, (int64_t)0, n, n, u0, d_ancestors,
                     d_total);
  HIP_TRY(hipGetLastError());
  if (d_fixed_weights)
    HIP_TRY(hipMemcpyAsync(d_fixed_weights, sc.d_w, (size_t)n * sizeof(int64_t),
                           hipMemcpyDeviceToDevice, stream));
  return SIPNET_OK;
}

extern "C" int sipnet_pf_systematic_ancestors_async(const double* d_logw, int64_t n, double u0,
                                                    int32_t* d_ancestors, int64_t* d_fixed_weights,
                                                    int64_t* d_total, void* hip_stream) {
  return ancestorsImpl(g_pf, d_logw, n, u0, d_ancestors, d_fixed_weights, d_total, 0, hip_stream);
}

int sipnet_pf_systematic_ancestors(const double* d_logw, int64_t n, double u0,
                                   int32_t* d_ancestors, int64_t* d_fixed_weights,
                                   void* hip_stream) {
  int rc = sipnet_pf_systematic_ancestors_async(d_logw, n, u0, d_ancestors, d_fixed_weights,
                                                nullptr, hip_stream);
  if (rc) return rc;
  // the one host round trip: a filter with no surviving particle must be reported
  hipStream_t stream = (hipStream_t)hip_stream;
  int64_t total = 0;
  HIP_TRY(hipMemcpyAsync(&total, g_pf.d_cdf + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (total <= 0) {
    setError("sipnet_pf_systematic_ancestors: every particle has zero weight");
    return SIPNET_ERR_BAD_PARAMETER;
  }
  return SIPNET_OK;
}

namespace {
struct PlanScratch {
  int device = -1;
  int64_t cap = 0;
  int32_t *d_head = nullptr, *d_P = nullptr;
  int64_t *d_first = nullptr, *d_counts = nullptr;  // [(kMaxWorld+1)*kMaxWorld], [4*kMaxWorld]
  void* d_tmp = nullptr;
  size_t tmpBytes = 0;
  void release() {
    if (d_head) (void)hipFree(d_head);
    if (d_P) (void)hipFree(d_P);
    if (d_first) (void)hipFree(d_first);
    if (d_counts) (void)hipFree(d_counts);
    if (d_tmp) (void)hipFree(d_tmp);
    d_head = d_P = nullptr; d_first = d_counts = nullptr; d_tmp = nullptr; cap = 0; tmpBytes = 0;
  }
  // (no destructor, see PfScratch)
};
thread_local PlanScratch g_plan;
}  // namespace

int sipnet_pf_exchange_plan(const int32_t* d_ancestors, int64_t n_local, int32_t world, int32_t rank,
                            int32_t* d_send_cols, int32_t* d_src, int64_t* send_counts,
                            int64_t* recv_counts, void* hip_stream) {
  if (!d_ancestors || !d_send_cols || !d_src || !send_counts || !recv_counts || n_local <= 0 ||
      world < 1 || world > kMaxWorld || rank < 0 || rank >= world ||
      n_local * world > (int64_t)1 << 30) {
    setError("sipnet_pf_exchange_plan: bad argument (world <= 64)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const int64_t total = n_local * world;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  PlanScratch& sc = g_plan;
  if (sc.device != dev || sc.cap < total) {
    sc.release();
    sc.device = dev;
    HIP_TRY(hipMalloc(&sc.d_head, (size_t)total * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&sc.d_P, (size_t)total * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&sc.d_first, (size_t)(kMaxWorld + 1) * kMaxWorld * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&sc.d_counts, (size_t)(4 * kMaxWorld + 1) * sizeof(int64_t)));   // counts, bases, validity flag
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, sc.tmpBytes, sc.d_head, sc.d_P, (int)total, stream));
    HIP_TRY(hipMalloc(&sc.d_tmp, sc.tmpBytes));
    sc.cap = total;
  }
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(planFirstKernel, dim3(world), dim3(kMaxWorld + 1), 0, stream, d_ancestors, n_local,
                     world, sc.d_first);
  int32_t* d_bad = (int32_t*)(sc.d_counts + 4 * kMaxWorld);
  HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(int64_t), stream));
  hipLaunchKernelGGL(planHeadKernel, dim3(grid), dim3(256), 0, stream, d_ancestors, n_local, total, sc.d_head, d_bad);
  size_t tmpBytes = sc.tmpBytes;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(sc.d_tmp, tmpBytes, sc.d_head, sc.d_P, (int)total, stream));
  int64_t* d_bases = sc.d_counts + 2 * kMaxWorld;
  hipLaunchKernelGGL(planCountKernel, dim3(1), dim3(64), 0, stream, sc.d_first, sc.d_P, sc.d_head, total,
                     world, rank, sc.d_counts, d_bases);
  hipLaunchKernelGGL(planFillKernel, dim3(grid), dim3(256), 0, stream, d_ancestors, n_local, total, world,
                     rank, sc.d_first, sc.d_P, sc.d_head, d_bases, d_send_cols, d_src, d_bad);
  HIP_TRY(hipGetLastError());
  // the one host round trip of the plan: the split sizes of the all-to-all
  int64_t h[4 * kMaxWorld + 1];
  HIP_TRY(hipMemcpyAsync(h, sc.d_counts, (size_t)(4 * kMaxWorld + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (h[4 * kMaxWorld] != 0) {
    setError("sipnet_pf_exchange_plan: the ancestor vector is not non-decreasing inside [0, world * n_local)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  for (int q = 0; q < world; q++) {
    send_counts[q] = h[q];
    recv_counts[q] = h[world + q];
  }
  return SIPNET_OK;
}

void sipnet_pf_release_scratch(void) {
  g_pf.release();
  g_plan.release();
}

int sipnet_batch_pack_members(sipnet_batch* b, const int32_t* d_cols, int64_t n,
                              int32_t with_params, double* d_buf, void* hip_stream) {
  if (!b || n < 0 || (n > 0 && (!d_cols || !d_buf))) {
    setError("sipnet_batch_pack_members: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  rc = flushParams(b, stream);
  if (rc) return rc;
  RecvMap none{};
  // block layout: [NSTATE rows | RING_SLOTS rows (fp32-mixed batches: of floats) | NPARAMS rows] x n columns
  const bool rf = b->precision == SIPNET_F32_MIXED;
  launchGatherMember(b->d_state, b->d_ring, rf, with_params ? b->d_prm : nullptr, b->ncol, nullptr, none, d_cols, n, d_buf,
                     d_buf + (size_t)SIPNET_NSTATE * n, d_buf + (size_t)(SIPNET_NSTATE + ringWords(rf)) * n, n, stream,
                     b->prmIndexed ? b->d_prmId : nullptr);   // (a resampled index: the rows are read through it)
  HIP_TRY(hipGetLastError());
  return SIPNET_OK;
}

int sipnet_batch_resample(sipnet_batch* b, const int32_t* d_src, const double* d_recv,
                          int32_t n_blocks, const int64_t* block_cols, int32_t with_params,
                          void* hip_stream) {
  if (!b || !d_src || n_blocks < 0 || n_blocks > kMaxBlocks || (n_blocks > 0 && !block_cols)) {
    setError("sipnet_batch_resample: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->n_sites != 1) {
    setError("sipnet_batch_resample: particles of different sites must not mix (n_sites must be 1)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  b->pfPre.valid = false;   // (a forecast's log-weights belong to the particles as they were)
  b->pfArm.set = false;
  rc = flushParams(b, stream);
  if (rc) return rc;
  const size_t nc = (size_t)b->ncol;
  if (!b->d_state2) HIP_TRY(hipMalloc(&b->d_state2, nc * SIPNET_NSTATE * sizeof(double)));
  if (!b->d_ring2) HIP_TRY(hipMalloc(&b->d_ring2, nc * SIPNET_RING_SLOTS * ringElemBytes(b)));
  // every particle is here and carries its parameters: they stay where set_params put them, an index is resampled
  // (4 bytes per particle instead of 640; the one-wave forecast kernel reads through it).  With blocks received from
  // other ranks the rows themselves travel, as before.
  const bool byIndex = with_params && n_blocks == 0;
  if (b->d_prmBank && !with_params) {
    setError("sipnet_batch_resample: this batch is connected to a filter whose particles carry their parameters "
             "(sipnet_batch_pf_connect): resample with_params");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (with_params && !byIndex) {
    rc = materializeParams(b, stream);
    if (rc) return rc;
    pfDropBank(b);   // (parameter ROWS are about to move: the connection's index means nothing afterwards)
    if (!b->d_prm2) HIP_TRY(hipMalloc(&b->d_prm2, nc * SIPNET_NPARAMS * sizeof(double)));
  }
  if (byIndex) {
    if (!b->d_prmId) HIP_TRY(hipMalloc(&b->d_prmId, nc * sizeof(int32_t)));
    if (!b->d_prmId2) HIP_TRY(hipMalloc(&b->d_prmId2, nc * sizeof(int32_t)));
    if (b->d_prmBank) {
      b->prmIndexed = true;   // (a connected filter's index is always current: slots of the bank)
    } else if (!b->prmIndexed) {
      hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, b->d_prmId, (int64_t)nc);
      b->prmIndexed = true;
    }
  }
  const int words = sipnet_batch_member_words(b, with_params);
  RecvMap map{};
  map.nBlocks = n_blocks;
  int64_t start = 0, off = 0;
  for (int s = 0; s < n_blocks; s++) {
    if (block_cols[s] < 0) {
      setError("sipnet_batch_resample: negative block size");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    map.start[s] = start;
    map.off[s] = off;
    map.n[s] = block_cols[s];
    start += block_cols[s];
    off += block_cols[s] * words;
  }
  map.start[n_blocks] = start;
  if (start > 0 && !d_recv) {
    setError("sipnet_batch_resample: received columns announced but no buffer given");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  launchGatherMember(b->d_state, b->d_ring, b->precision == SIPNET_F32_MIXED, (with_params && !byIndex) ? b->d_prm : nullptr, b->ncol,
                     d_recv, map, d_src, b->ncol, b->d_state2, b->d_ring2, b->d_prm2, b->ncol, stream, nullptr,
                     byIndex ? b->d_prmId : nullptr, byIndex ? b->d_prmId2 : nullptr);
  HIP_TRY(hipGetLastError());
  std::swap(b->d_state, b->d_state2);
  std::swap(b->d_ring, b->d_ring2);
  if (byIndex) std::swap(b->d_prmId, b->d_prmId2);
  else if (with_params) std::swap(b->d_prm, b->d_prm2);
  if (b->pfPeers) b->pfPeers->parity ^= 1;   // (connected ranks resample in lockstep, whichever entry point they use)
  rc = markBusy(b, stream);                  // (an upload of new parameters waits for the gather that is writing them)
  if (rc) return rc;
  if (with_params && start > 0) {
    // parameters that arrived from other ranks may change which kernel variant the batch needs
    int32_t* d_flag = nullptr;
    HIP_TRY(hipMalloc(&d_flag, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int32_t), stream));
    hipLaunchKernelGGL(exponentCheckKernel, dim3((unsigned)((b->ncol + 255) / 256)), dim3(256), 0,
                       stream, b->d_prm, b->ncol, d_flag);
    int32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipFree(d_flag));
    if (flag != 0) b->genericExponents = true;  // only ever widened: plain-exponent kernels must never see a general exponent
  }
  return SIPNET_OK;
}

int sipnet_batch_pf_analysis(sipnet_batch* b, const void* d_plane, int32_t elem_is_f32, int32_t n_steps,
                             int64_t ld, double obs, double sigma, double u0, int32_t with_params,
                             double* d_logw, int32_t* d_ancestors, int64_t* d_total, void* hip_stream) {
  if (!b || !d_logw || !d_ancestors) {
    setError("sipnet_batch_pf_analysis: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  PfScratch& sc = scratchOf(b);
  rc = pfScratchFor(sc, b->ncol, (hipStream_t)hip_stream);
  if (rc) return rc;
  if (!d_plane || n_steps <= 0 || ld < b->ncol || !(sigma > 0) || !(u0 >= 0.0) || !(u0 < 1.0) || b->ncol > (int64_t)1 << 22) {
    setError("sipnet_batch_pf_analysis: bad argument (sigma > 0, 0 <= u0 < 1, at most 4194304 particles)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  // (the forecast's launch has left the log-weights of exactly this plane, observation and sigma: sipnet_batch_pf_arm)
  const sipnet_batch::PfPre pre = b->pfPre;
  const bool havePre = pre.valid && pre.plane == d_plane && pre.nSteps == n_steps && pre.ld == ld && pre.obs == obs &&
                       pre.sigma == sigma && pre.d_logw == d_logw && elem_is_f32 == (b->precision == SIPNET_F32_MIXED);
  b->pfPre.valid = false;
  const int budget = (b->kernelOptions & SIPNET_KOPT_PF_MULTI_LAUNCH) ? 0 : fusedBudget(sc, b, elem_is_f32 ? 0 : 1);
  b->pfInfo.fused = fusable(b->ncol, budget) ? 1 : 0;
  b->pfInfo.budget = budget;
  if (b->pfInfo.fused) {
    // log-weights, fixed-point weights, prefix sum and ancestors: ONE launch (pfFusedKernel)
    FusedArgs fa{};
    fa.plane = d_plane;
    fa.nSteps = n_steps;
    fa.ld = ld;
    fa.ncol = b->ncol;
    fa.status = b->d_state + (size_t)ST_status * b->ncol;
    fa.obs = obs;
    fa.invSigma = 1.0 / sigma;
    fa.logw = d_logw;
    fa.nSlots = b->ncol;
    int grid;
    fusedGeometry(fa.nSlots, budget, &grid, &fa.chunk);
    b->pfInfo.grid = grid;
    fa.blockMax = sc.d_max;
    fa.logwIn = d_logw;
    fa.threadIncl = sc.d_threadIncl;
    fa.blockSum = sc.d_blockSum;
    fusedBarrier(sc, &fa, b->pfSpinBudget);
    fa.absent = b->pfDebugAbsent;
    b->pfDebugAbsent = -1;
    fa.preMax = havePre ? b->d_pfPreMax : nullptr;
    fa.nPre = havePre ? pre.nMax : 0;
    fa.j0 = 0;
    fa.nOut = fa.nTotal = b->ncol;
    fa.u0 = u0;
    fa.anc = d_ancestors;
    fa.total = d_total;
    fa.totalScratch = sc.d_blockSum + kFusedBlocks;
    if (elem_is_f32) hipLaunchKernelGGL((pfFusedKernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, fa);
    else hipLaunchKernelGGL((pfFusedKernel<double, false>), dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, fa);
    rc = fusedLaunched(sc);
    if (rc) return rc;
  } else {
    // next to nothing of the device is ours to spin on (a sliver of a partitioned device, many filters sharing it): the
    // phases as launches of their own -- log-weights + 256-wide maxima | fixed-point weights | prefix sum | ancestors
    b->pfInfo.grid = 0;
    int parts;
    const double* partPtr = nullptr;
    if (havePre) {
      parts = pre.nMax;
      partPtr = b->d_pfPreMax;
    } else {
      rc = logWeights(b, d_plane, elem_is_f32, n_steps, ld, obs, sigma, d_logw, sc.d_max, hip_stream);
      if (rc) return rc;
      parts = (int)((b->ncol + 255) / 256);
    }
    rc = ancestorsImpl(sc, d_logw, b->ncol, u0, d_ancestors, nullptr, sc.d_blockSum + kFusedBlocks, parts, hip_stream, partPtr);
    if (rc) return rc;
    if (d_total)
      HIP_TRY(hipMemcpyAsync(d_total, sc.d_blockSum + kFusedBlocks, sizeof(int64_t), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  }
  if (!d_total) {   // the synchronous check of sipnet_pf_systematic_ancestors
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, sc.d_blockSum + kFusedBlocks, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    if (total == kPfVoid) {
      setError("sipnet_batch_pf_analysis: " + fusedStuckReport(sc, (hipStream_t)hip_stream));
      return SIPNET_ERR_INTERNAL;
    }
    if (total <= 0) {
      setError("sipnet_batch_pf_analysis: every particle has zero weight");
      return SIPNET_ERR_BAD_PARAMETER;
    }
  }
  return sipnet_batch_resample(b, d_ancestors, nullptr, 0, nullptr, with_params, hip_stream);
}

}  // extern "C"

// ---- the filter across ranks by peer reads --------------------------------------------------------------------
void pfRelease(sipnet_batch* b) {
  if (b->pfScratch) {
    b->pfScratch->release();
    delete b->pfScratch;
    b->pfScratch = nullptr;
  }
  if (b->pfPeers) {
    for (void* p : b->pfPeers->opened) (void)hipIpcCloseMemHandle(p);
    delete b->pfPeers;
    b->pfPeers = nullptr;
  }
}

static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(((sipnet_pf_peer*)nullptr)->ipc[0]), "sipnet_pf_peer::ipc holds a hipIpcMemHandle_t");

static int ensureSpares(sipnet_batch* b, bool withParams) {
  const size_t nc = (size_t)b->ncol;
  if (!b->d_state2) HIP_TRY(hipMalloc(&b->d_state2, nc * SIPNET_NSTATE * sizeof(double)));
  if (!b->d_ring2) HIP_TRY(hipMalloc(&b->d_ring2, nc * SIPNET_RING_SLOTS * ringElemBytes(b)));
  if (withParams && !b->d_prm2) HIP_TRY(hipMalloc(&b->d_prm2, nc * SIPNET_NPARAMS * sizeof(double)));
  return SIPNET_OK;
}

extern "C" {

int sipnet_batch_pf_publish(sipnet_batch* b, int32_t with_params, sipnet_pf_peer* out) {
  if (!b || !out) {
    setError("sipnet_batch_pf_publish: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->n_sites != 1) {
    setError("sipnet_batch_pf_publish: particles of different sites must not mix (n_sites must be 1)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  rc = ensureSpares(b, with_params != 0);
  if (rc) return rc;
  rc = waitIdle(b);                      // (whatever stream the batch last ran on: the spares and the parameters settle)
  if (rc) return rc;
  rc = flushParams(b, nullptr);          // (peers will read the converted block)
  if (rc) return rc;
  rc = materializeParams(b, nullptr);    // (... in column order: peers address a particle's rows by its column)
  if (rc) return rc;
  pfDropBank(b);                         // (an earlier connection's bank: the next connect builds a new one)
  rc = waitIdle(b);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  out->process_id = (int64_t)getpid();
  out->device = b->device;
  out->n_particles = (int32_t)b->ncol;
  out->precision = b->precision;
  out->with_params = with_params ? 1 : 0;
  out->generic_exponents = b->genericExponents ? 1 : 0;
  const bool byIndex = with_params && !(b->kernelOptions & SIPNET_KOPT_PF_MOVE_PARAMS);
  out->params_by_index = byIndex ? 1 : 0;
  if (byIndex) {   // the particles' index into the bank of all ranks' parameters (filled by connect), double-buffered like the state
    const size_t nc = (size_t)b->ncol;
    if (!b->d_prmId) HIP_TRY(hipMalloc(&b->d_prmId, nc * sizeof(int32_t)));
    if (!b->d_prmId2) HIP_TRY(hipMalloc(&b->d_prmId2, nc * sizeof(int32_t)));
  }
  void* ptr[8] = {b->d_state, b->d_state2, b->d_ring, b->d_ring2, with_params ? b->d_prm : nullptr,
                  with_params ? b->d_prm2 : nullptr, byIndex ? b->d_prmId : nullptr, byIndex ? b->d_prmId2 : nullptr};
  out->ipc_valid = 1;
  for (int k = 0; k < 8; k++) {
    out->address[k] = (uint64_t)(uintptr_t)ptr[k];
    if (!ptr[k]) continue;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, ptr[k]) != hipSuccess) {   // peers inside this process do not need it
      (void)hipGetLastError();
      out->ipc_valid = 0;
      continue;
    }
    memcpy(out->ipc[k], &h, sizeof h);
  }
  return SIPNET_OK;
}

int sipnet_batch_pf_connect(sipnet_batch* b, int32_t world, int32_t rank, const sipnet_pf_peer* peers) {
  if (!b || !peers || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) {
    setError("sipnet_batch_pf_connect: bad argument (world <= 16)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  const sipnet_pf_peer& me = peers[rank];
  if (me.process_id != (int64_t)getpid() || me.address[0] != (uint64_t)(uintptr_t)b->d_state ||
      me.address[1] != (uint64_t)(uintptr_t)b->d_state2 || me.n_particles != b->ncol) {
    setError("sipnet_batch_pf_connect: peers[rank] is not what this batch published (publish, then connect, with no "
             "resampling in between)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->pfPeers) {
    for (void* p : b->pfPeers->opened) (void)hipIpcCloseMemHandle(p);
    delete b->pfPeers;
    b->pfPeers = nullptr;
  }
  PfPeers* pp = new PfPeers();
  pp->world = world;
  pp->rank = rank;
  pp->withParams = me.with_params;
  bool generic = false;
  auto fail = [&](const std::string& why, int code) {
    for (void* p : pp->opened) (void)hipIpcCloseMemHandle(p);
    delete pp;
    setError("sipnet_batch_pf_connect: " + why);
    return code;
  };
  for (int s = 0; s < world; s++) {
    const sipnet_pf_peer& q = peers[s];
    if (q.precision != me.precision || q.with_params != me.with_params || q.params_by_index != me.params_by_index || q.n_particles <= 0)
      return fail("rank " + std::to_string(s) + " published another precision / parameter mode", SIPNET_ERR_BAD_ARGUMENT);
    pp->count[s] = q.n_particles;
    if (q.n_particles > pp->nmax) pp->nmax = q.n_particles;
    if (s < rank) pp->first += q.n_particles;
    pp->nTotal += q.n_particles;
    generic = generic || q.generic_exponents != 0;
    void* ptr[8];
    if (q.process_id == me.process_id) {   // same process (the node object): the addresses themselves
      for (int k = 0; k < 8; k++) ptr[k] = (void*)(uintptr_t)q.address[k];
      if (q.device != b->device) {
        hipError_t e = hipDeviceEnablePeerAccess(q.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e), SIPNET_ERR_NO_DEVICE);
        (void)hipGetLastError();
      }
    } else {                               // another process: map its allocations (dmabuf IPC)
      if (!q.ipc_valid) return fail("rank " + std::to_string(s) + " could not export IPC handles", SIPNET_ERR_NO_DEVICE);
      for (int k = 0; k < 8; k++) {
        ptr[k] = nullptr;
        if (!q.address[k]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, q.ipc[k], sizeof h);
        hipError_t e = hipIpcOpenMemHandle(&ptr[k], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e), SIPNET_ERR_NO_DEVICE);
        pp->opened.push_back(ptr[k]);
      }
    }
    for (int par = 0; par < 2; par++) {
      pp->state[par][s] = (const double*)ptr[0 + par];
      pp->ring[par][s] = ptr[2 + par];
      pp->prm[par][s] = (const double*)ptr[4 + par];
      pp->ids[par][s] = (const int32_t*)ptr[6 + par];
    }
  }
  if ((int64_t)world * pp->nmax > (int64_t)1 << 22) return fail("more than 4 194 304 weight slots", SIPNET_ERR_BAD_ARGUMENT);
  // particles carry their parameters between ranks: every rank runs the kernel variant the most general
  // parameter set anywhere needs (decided here, once -- not by a device -> host check after every exchange)
  if (generic && me.with_params) b->genericExponents = true;
  pp->byIndex = me.params_by_index != 0;
  if (pp->byIndex) {
    // Every rank's converted parameters, once: [NPARAMS][world * nmax], rank s's particle c in column s * nmax + c (the slot
    // numbering of the weights).  Parameters are constants of a particle; what a resampling moves from now on is this column
    // number -- 4 bytes instead of 640, and the forecast's parameter reads stay in local HBM.  (The peers' blocks are read
    // where they are: peer-mapped HBM, as every later gather reads state and ring.)
    const int64_t pitch = (int64_t)world * pp->nmax;
    pfDropBank(b);
    if (hipMalloc(&b->d_prmBank, (size_t)pitch * SIPNET_NPARAMS * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      b->d_prmBank = nullptr;
      return fail("no memory for the bank of all ranks' parameters (SIPNET_KOPT_PF_MOVE_PARAMS does without)", SIPNET_ERR_INTERNAL);
    }
    b->prmBankPitch = pitch;
    for (int s = 0; s < world; s++) {
      const int64_t cnt = pp->count[s];
      hipLaunchKernelGGL(copyRowsKernel, dim3((unsigned)((cnt + 255) / 256), 8), dim3(256), 0, nullptr, b->d_prmBank + (int64_t)s * pp->nmax,
                         pitch, pp->prm[0][s], cnt, cnt, (int32_t)SIPNET_NPARAMS);
    }
    hipLaunchKernelGGL(iotaKernel, dim3((unsigned)((b->ncol + 255) / 256)), dim3(256), 0, nullptr, b->d_prmId, b->ncol, (int32_t)(rank * pp->nmax));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      pfDropBank(b);
      return fail(std::string("filling the parameter bank: ") + hipGetErrorString(e), SIPNET_ERR_INTERNAL);
    }
    b->prmIndexed = false;   // (d_prm is current too: nothing has been resampled yet)
  }
  if ((!b->d_pfCrossing && hipMalloc(&b->d_pfCrossing, sizeof(unsigned long long)) != hipSuccess) ||
      hipMemset(b->d_pfCrossing, 0, sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    pfDropBank(b);
    return fail("no memory for the crossing counter", SIPNET_ERR_INTERNAL);
  }
  b->pfInfo.cycles = 0;
  b->pfPeers = pp;
  pp->bankLost = false;
  return SIPNET_OK;
}

int64_t sipnet_batch_pf_block_len(const sipnet_batch* b) {
  if (!b) return -1;
  const int64_t nmax = b->pfPeers ? b->pfPeers->nmax : b->ncol;
  return nmax + (nmax + 255) / 256;
}

int sipnet_batch_pf_local_weights(sipnet_batch* b, const void* d_plane, int32_t elem_is_f32, int32_t n_steps,
                                  int64_t ld, double obs, double sigma, double* d_block, void* hip_stream) {
  if (!b || !d_block) {
    setError("sipnet_batch_pf_local_weights: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int64_t nmax = b->pfPeers ? b->pfPeers->nmax : b->ncol;
  // the forecast's launch was told of this analysis (sipnet_batch_pf_arm with d_logw = this block) and has left the
  // log-weights in it: only the 256-wide maxima and the empty slots remain (wavefront k covers columns 64 k .. 64 k + 63
  // when the one site's members are a multiple of 64)
  const sipnet_batch::PfPre& pre = b->pfPre;
  const bool havePre = pre.valid && pre.plane == d_plane && pre.nSteps == n_steps && pre.ld == ld && pre.obs == obs &&
                       pre.sigma == sigma && pre.d_logw == d_block && b->n_sites == 1 && b->ncol % 64 == 0 &&
                       elem_is_f32 == (b->precision == SIPNET_F32_MIXED);
  b->pfPre.valid = false;
  if (havePre) {
    int rc = useDevice(b);
    if (rc) return rc;
    const int64_t work = std::max<int64_t>((nmax + 255) / 256, nmax - b->ncol);
    hipLaunchKernelGGL(blockFromWaveMaximaKernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       b->d_pfPreMax, b->ncol / 64, b->ncol, nmax, d_block, d_block + nmax);
    HIP_TRY(hipGetLastError());
    return SIPNET_OK;
  }
  return logWeights(b, d_plane, elem_is_f32, n_steps, ld, obs, sigma, d_block, d_block + nmax, hip_stream, nmax);
}

int sipnet_batch_pf_resample_peers(sipnet_batch* b, const double* d_gathered, double u0, int32_t* d_ancestors,
                                   int64_t* d_total, void* hip_stream) {
  if (!b || !d_gathered || !d_ancestors || !(u0 >= 0.0) || !(u0 < 1.0)) {
    setError("sipnet_batch_pf_resample_peers: bad argument (0 <= u0 < 1)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (b->n_sites != 1) {
    setError("sipnet_batch_pf_resample_peers: particles of different sites must not mix (n_sites must be 1)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  rc = flushParams(b, stream);
  if (rc) return rc;
  const bool byIndex = b->pfPeers && b->pfPeers->byIndex;
  if (byIndex && (b->pfPeers->bankLost || !b->d_prmBank)) {
    setError("sipnet_batch_pf_resample_peers: this rank's parameters were set anew (or moved as rows by sipnet_batch_resample) after "
             "sipnet_batch_pf_connect replicated them on every rank: publish and connect again");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (!byIndex) {
    rc = materializeParams(b, stream);   // (peers read a particle's parameter rows by its column)
    if (rc) return rc;
  }
  PeerPtrs tab{};
  int64_t nTotal = b->ncol, first = 0;
  bool withParams;
  if (b->pfPeers) {
    const PfPeers& pp = *b->pfPeers;
    tab.world = pp.world;
    tab.nmax = pp.nmax;
    tab.rank = pp.rank;
    for (int s = 0; s < pp.world; s++) {
      tab.state[s] = pp.state[pp.parity][s];
      tab.ring[s] = pp.ring[pp.parity][s];
      tab.third[s] = byIndex ? (const void*)pp.ids[pp.parity][s] : (const void*)pp.prm[pp.parity][s];
      tab.pitch[s] = pp.count[s];
    }
    nTotal = pp.nTotal;
    first = pp.first;
    withParams = pp.withParams != 0;
    if (tab.state[pp.rank] != b->d_state || (byIndex && tab.third[pp.rank] != (const void*)b->d_prmId)) {
      setError("sipnet_batch_pf_resample_peers: the batch was resampled behind the peers' back");
      return SIPNET_ERR_INTERNAL;
    }
    tab.crossing = b->d_pfCrossing;
  } else {   // not connected: a filter of this batch alone (parameters travel with the particles)
    tab.world = 1;
    tab.nmax = (int32_t)b->ncol;
    tab.rank = 0;
    tab.state[0] = b->d_state;
    tab.ring[0] = b->d_ring;
    tab.third[0] = b->d_prm;
    tab.pitch[0] = (int32_t)b->ncol;
    withParams = true;
  }
  rc = ensureSpares(b, withParams && !byIndex);
  if (rc) return rc;
  const int64_t nSlots = (int64_t)tab.world * tab.nmax, stride = tab.nmax + (tab.nmax + 255) / 256;
  PfScratch& sc = scratchOf(b);
  rc = pfScratchFor(sc, nSlots, stream);
  if (rc) return rc;
  const int64_t n = b->ncol;
  const int budget = (b->kernelOptions & SIPNET_KOPT_PF_MULTI_LAUNCH) ? 0 : fusedBudget(sc, b, 2);
  b->pfInfo.fused = fusable(nSlots, budget) ? 1 : 0;
  b->pfInfo.budget = budget;
  b->pfInfo.nSlots = nSlots;
  b->pfInfo.grid = 0;
  if (b->pfInfo.fused) {   // weights over all slots, prefix sum, the ancestors of MY particles: one launch (pfFusedKernel)
    FusedArgs fa{};
    fa.gathered = d_gathered;
    fa.world = tab.world;
    fa.nmax = tab.nmax;
    fa.stride = stride;
    fa.nSlots = nSlots;
    int grid;
    fusedGeometry(fa.nSlots, budget, &grid, &fa.chunk);
    b->pfInfo.grid = grid;
    fa.blockMax = sc.d_max;
    fa.threadIncl = sc.d_threadIncl;
    fa.blockSum = sc.d_blockSum;
    fusedBarrier(sc, &fa, b->pfSpinBudget);
    fa.absent = b->pfDebugAbsent;
    b->pfDebugAbsent = -1;
    fa.j0 = first;
    fa.nOut = n;
    fa.nTotal = nTotal;
    fa.u0 = u0;
    fa.anc = d_ancestors;
    fa.total = d_total;
    fa.totalScratch = sc.d_blockSum + kFusedBlocks;
    hipLaunchKernelGGL((pfFusedKernel<double, true>), dim3(grid), dim3(256), 0, stream, fa);
    rc = fusedLaunched(sc);
    if (rc) return rc;
  } else {   // the same as launches of their own (see sipnet_batch_pf_analysis)
    const int gridW = (int)((nSlots + 255) / 256);
    hipLaunchKernelGGL(fixedWeightGatheredKernel, dim3(gridW), dim3(256), 0, stream, d_gathered, tab.world, tab.nmax, stride, sc.d_w);
    size_t tmpBytes = sc.tmpBytes;
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(sc.d_tmp, tmpBytes, sc.d_w, sc.d_cdf, (int)nSlots, stream));
    hipLaunchKernelGGL(ancestorKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sc.d_cdf, nSlots, first, n, nTotal, u0,
                       d_ancestors, sc.d_blockSum + kFusedBlocks);
    HIP_TRY(hipGetLastError());
    if (d_total) HIP_TRY(hipMemcpyAsync(d_total, sc.d_blockSum + kFusedBlocks, sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  }
  auto groups = [](int rows) { return (rows + kGatherRows - 1) / kGatherRows; };
  PeerParts parts{};
  parts.p[0] = PeerPart{b->d_state2, SIPNET_NSTATE, 0, 0};
  parts.p[1] = PeerPart{b->d_ring2, SIPNET_RING_SLOTS, groups(SIPNET_NSTATE), b->precision == SIPNET_F32_MIXED ? 1 : 0};
  parts.n = 2;
  int total = groups(SIPNET_NSTATE) + groups(SIPNET_RING_SLOTS);
  if (byIndex) {          // the particle's column in the bank of all ranks' parameters: one row of 4-byte elements
    parts.p[2] = PeerPart{b->d_prmId2, 1, total, 1};
    parts.n = 3;
    total += 1;
  } else if (withParams) {
    parts.p[2] = PeerPart{b->d_prm2, SIPNET_NPARAMS, total, 0};
    parts.n = 3;
    total += groups(SIPNET_NPARAMS);
  }
  hipLaunchKernelGGL(gatherPeerKernel, dim3((unsigned)((n + 255) / 256), (unsigned)total), dim3(256), 0, stream, parts, tab,
                     d_ancestors, n, n);
  HIP_TRY(hipGetLastError());
  std::swap(b->d_state, b->d_state2);
  std::swap(b->d_ring, b->d_ring2);
  if (byIndex) {
    std::swap(b->d_prmId, b->d_prmId2);
    b->prmIndexed = true;   // (d_prm, the column-order copy, is behind the index now: materializeParams)
  } else if (withParams) {
    std::swap(b->d_prm, b->d_prm2);
  }
  if (b->pfPeers) {
    b->pfPeers->parity ^= 1;
    b->pfInfo.cycles++;
  }
  return markBusy(b, stream);
}

// what the last analysis did, and how many of this rank's particles have crossed ranks (see sipnet_amd.h)
int sipnet_batch_pf_info(sipnet_batch* b, sipnet_pf_info* out, void* hip_stream) {
  if (!b || !out) {
    setError("sipnet_batch_pf_info: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  int rc = useDevice(b);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  out->fused = b->pfInfo.fused;
  out->grid = b->pfInfo.grid;
  out->budget = b->pfInfo.budget;
  out->world = b->pfPeers ? b->pfPeers->world : 1;
  out->n_slots = b->pfInfo.nSlots;
  out->cycles = b->pfInfo.cycles;
  out->params_by_index = b->d_prmBank ? 1 : 0;
  out->device_share = b->deviceShare;
  if (b->d_pfCrossing) {
    unsigned long long c = 0;
    HIP_TRY(hipMemcpyAsync(&c, b->d_pfCrossing, sizeof c, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    out->crossing = (int64_t)c;
  }
  return SIPNET_OK;
}

}  // extern "C"

// plan_device.hip -- the site plan built on the device (see plan_device.h for the decomposition).
// Compiled with -ffp-contract=off like plan.cpp: every operation below is the IEEE operation the host performs.
#include "plan_device.h"

#include <climits>

namespace sipnet {

namespace {
constexpr double kMeanNppDays = 5.0;  // sipnet.c:39
constexpr double kEStarSnow = 0.6;    // sipnet.c:890-891
constexpr int kSlots = SIPNET_RING_SLOTS;

// What the ring walk and the record kernel want prepared: the step lengths as a compact array, per 256 steps "a step length
// changes in here" (the walk finds the end of a run of equal lengths with it) and the block's largest year
// (planExpandKernel's prefix maximum).
__global__ __launch_bounds__(256) void planPrepKernel(DevPlanArgs a) {
  const int d = blockIdx.y;
  const DevPlanSite S = a.sites[d];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= S.n) return;
  const bool v = t < S.n;
  const int tc = v ? t : S.n - 1;
  const double* r = S.clim + (size_t)SIPNET_NCLIM * tc;
  const double len = r[0];
  if (v) a.lenC[(size_t)d * a.nT + t] = len;
  const int yr = S.year[tc];
  const bool chg = v && t > 0 && len != S.clim[(size_t)SIPNET_NCLIM * (t - 1)];
  const int anyChg = __syncthreads_or(chg ? 1 : 0);
  __shared__ int waveMax[4];
  int x = v ? yr : INT_MIN;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = max(x, __shfl_xor(x, off));
  if ((threadIdx.x & 63) == 0) waveMax[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.blockInfo[((size_t)d * a.nBlk + blockIdx.x) * 2] = anyChg;
    a.blockInfo[((size_t)d * a.nBlk + blockIdx.x) * 2 + 1] = max(max(waveMax[0], waveMax[1]), max(waveMax[2], waveMax[3]));
  }
}

// ---- the ring's eviction schedule ----------------------------------------------------------------------------------
// RingSched::advance (plan.cpp) restated on (front entry j, its remaining weight): entries are the steps since the
// last ring reset r0 (slot (u - r0) mod 250, weight len[u]; the reset / initial entry: weight 5), evicted from the front.
constexpr int kWin = 2048, kBatch = 1024;   // lengths held in LDS (by step & (kWin - 1)); steps walked per refill

struct RingState {
  int t, j, r0, opCount, runStart, noFF, ffT0, ffNOps, status, statusAt, nRuns;
  double wFront, ffLen;
};

__device__ void ringWalk(const DevPlanArgs& a, const DevPlanSite& S, int d) {
  __shared__ double win[kWin];
  __shared__ double pre[256];   // the live entries the walk starts from (a fresh ring's one entry; a checkpoint's ring)
  __shared__ RingState st;
  const int lane = threadIdx.x, n = S.n;
  const int K = S.preK, preIns = S.preIns, opCap = S.opCap;
  for (int i = lane; i < K; i += 64) pre[i] = S.preW[i];
  const double* len = a.lenC + (size_t)d * a.nT;
  DevPlanSeq* seq = a.seq + (size_t)d * a.nT;
  RingOp* ops = a.ringOps + S.opBase;
  DevPlanRun* runs = a.runs + (size_t)d * a.runCap;
  constexpr int M = kWin - 1;
  int loaded = 0;
  if (lane == 0) {
    // entries before the first record are "steps" -K .. -1, the front one in slot preStart
    st.t = 0; st.j = -K; st.r0 = -K - S.preStart; st.opCount = 0; st.runStart = 0; st.noFF = 0; st.ffT0 = -1; st.ffNOps = 0;
    st.status = 0; st.statusAt = -1; st.nRuns = 0; st.wFront = S.preW[0]; st.ffLen = 0.0;
  }
  __syncthreads();
  while (true) {
    const int tBegin = st.t;
    if (tBegin >= n) break;
    const int want = min(n, tBegin + kBatch);
    // (after a jump over a run only the last 320 lengths before the new position are still of interest)
    const int floorLoaded = (tBegin - 320) & ~63;
    if (loaded < floorLoaded) loaded = floorLoaded;
    while (loaded < want) {
      const int u = loaded + lane;
      if (u < n) win[u & M] = len[u];
      loaded += 64;
    }
    __syncthreads();
    if (lane == 0) {
      int t = tBegin, j = st.j, r0 = st.r0, opCount = st.opCount, runStart = st.runStart, status = st.status, statusAt = st.statusAt;
      const int noFF = st.noFF;
      double wF = st.wFront;
      int ffT0 = -1, ffNOps = 0;
      for (; t < want; t++) {
        const double L = win[t & M];
        if (t > 0 && L != win[(t - 1) & M]) runStart = t;
        const int opFirst = opCount, jB = j;
        const double wB = wF;
        int nOps = 0, insSlot, s0 = 0, s1 = 0, i0 = -1, i1 = -1;
        double w0 = 0.0, w1 = 0.0;
        if (L >= kMeanNppDays) {   // runmean.c:67-69: one entry carrying the whole window
          r0 = t; j = t; wF = kMeanNppDays; insSlot = -1;
        } else {
          double left = L;
          while (left > 0) {   // runmean.c:76-86
            if (j >= t) {      // the ring ran empty: not reachable while the live weights sum to the window
              if (!status) { status = 2; statusAt = t; }
              break;
            }
            const int slot = (j - r0) % kSlots, ins = j >= 0 ? j : preIns;
            double wop;
            if (wF > left) {
              wop = left; wF -= left; left = 0;
            } else {
              wop = wF; left -= wF; j++;
              wF = j >= t ? 0.0 : j >= 0 ? win[j & M] : pre[j + K];
            }
            RingOp op; op.w = wop; op.slot = slot; op.insStep = ins;
            if (opCount < opCap) ops[opCount] = op;   // (the room is the bound 2 n + preK: never false for a ring that
            else if (status != 3) { status = 3; statusAt = t; }   //  carries the window; a neighbour's list is not ours to write)
            opCount++;
            if (nOps == 0) { s0 = slot; i0 = ins; w0 = wop; }
            else if (nOps == 1) { s1 = slot; i1 = ins; w1 = wop; }
            nOps++;
          }
          insSlot = (t - r0) % kSlots;
          if (t - j >= kSlots && !status) { status = 1; statusAt = t; }   // runmean.c:93-95 (excluded by kDevPlanMinLen)
        }
        // fillFastRec's defaults for missing evictions (plan.cpp)
        const int safeSlot = insSlot >= 0 ? insSlot : 0;
        if (nOps == 0) { s0 = safeSlot; i0 = -1; w0 = 0.0; }
        if (nOps <= 1) { s1 = s0; i1 = i0; w1 = 0.0; }
        DevPlanSeq* q = seq + t;
        *(double2*)&q->w0 = make_double2(w0, w1);
        *(int4*)&q->ins0 = make_int4(i0, i1, opFirst, nOps);
        *(int4*)&q->packed = make_int4(s0 | (s1 << 8) | ((insSlot + 1) << 16), 0, 0, 0);
        // fixed point inside a run of equal lengths: every entry from the front on has this length, the front moved by
        // one and its remaining weight did not change -- the following steps of the run repeat this one, shifted
        if (t >= noFF && L < kMeanNppDays && jB >= runStart && j == jB + 1 &&
            __double_as_longlong(wF) == __double_as_longlong(wB) && t + 1 < n && t + 1 < loaded && win[(t + 1) & M] == L) {
          ffT0 = t; ffNOps = nOps; st.ffLen = L;
          t++;
          break;
        }
      }
      st.t = t; st.j = j; st.r0 = r0; st.opCount = opCount; st.runStart = runStart; st.status = status; st.statusAt = statusAt;
      st.wFront = wF; st.ffT0 = ffT0; st.ffNOps = ffNOps;
    }
    __syncthreads();
    if (st.ffT0 >= 0) {
      // first step from st.t on whose length differs: the rest of st.t's 256-step block looked at directly, then the
      // blocks' "a length changes in here" flags (64 blocks a round), then the first such block looked at directly
      const double L = st.ffLen;
      const int* info = a.blockInfo + (size_t)d * a.nBlk * 2;
      auto firstDifferent = [&](int from, int to) -> int {   // first u in [from, to) with len[u] != L, or -1; to - from <= 256
        int found = -1;
        unsigned long long bad[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int u = from + 64 * k + lane;
          bad[k] = __ballot(u < to && len[u] != L);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (found < 0 && bad[k]) found = from + 64 * k + __builtin_ctzll(bad[k]);
        return found;
      };
      const int nBlkSite = (n + 255) >> 8;
      int blk = st.t >> 8;
      int runEnd = firstDifferent(st.t, min(n, (blk + 1) << 8));
      blk++;
      while (runEnd < 0 && blk < nBlkSite) {
        const int bb = blk + lane;
        const unsigned long long m = __ballot(bb < nBlkSite && info[2 * bb] != 0);
        if (m) {
          blk += __builtin_ctzll(m);
          runEnd = firstDifferent(blk << 8, min(n, (blk + 1) << 8));   // (a change AT the block's first step counts: len[u] != L there)
          blk++;
        } else {
          blk += 64;
        }
      }
      if (runEnd < 0) runEnd = n;
      if (lane == 0) {
        const int K = runEnd - st.t;   // steps st.t .. runEnd-1 repeat the template step
        if (K >= kDevPlanMinRun) {
          DevPlanRun r; r.t0 = st.ffT0; r.count = K; r.nOps = st.ffNOps; r.pad = 0;
          runs[st.nRuns++] = r;
          st.t += K; st.j += K; st.opCount += st.ffNOps * K;
          if (st.opCount > opCap && st.status != 3) { st.status = 3; st.statusAt = st.ffT0; }
        } else {
          st.noFF = runEnd;
        }
        st.ffT0 = -1;
      }
      __syncthreads();
    }
  }
  if (lane == 0) {
    int32_t* o = a.siteOut + 8 * d;
    o[0] = st.nRuns; o[1] = st.opCount; o[2] = st.status; o[3] = st.statusAt; o[5] = opCap;
  }
}

__global__ __launch_bounds__(64) void planSeqKernel(DevPlanArgs a) {
  const int d = blockIdx.x;
  const DevPlanSite S = a.sites[d];
  const unsigned long long c0 = wall_clock64();   // (100 MHz: how long the walk took, for the profiles)
  ringWalk(a, S, d);
  if (threadIdx.x == 0) a.siteOut[8 * d + 4] = (int32_t)(wall_clock64() - c0);
}

__device__ inline int wrapSlot(int s) { return s % kSlots; }

__global__ __launch_bounds__(256) void planRunsKernel(DevPlanArgs a) {
  const int d = blockIdx.y;
  const DevPlanSite S = a.sites[d];
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int nRuns = a.siteOut[8 * d];
  if (u >= S.n || nRuns == 0) return;
  const DevPlanRun* runs = a.runs + (size_t)d * a.runCap;
  int lo = 0, hi = nRuns;   // last descriptor with t0 < u
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (runs[mid].t0 < u) lo = mid + 1; else hi = mid;
  }
  if (lo == 0) return;
  const DevPlanRun r = runs[lo - 1];
  const int i = u - r.t0;
  if (i > r.count) return;
  DevPlanSeq* seq = a.seq + (size_t)d * a.nT;
  const DevPlanSeq T = seq[r.t0];
  const int s0 = T.packed & 0xff, s1 = (T.packed >> 8) & 0xff, insSlot = (T.packed >> 16) - 1;
  DevPlanSeq* q = seq + u;
  *(double2*)&q->w0 = make_double2(T.w0, T.w1);
  *(int4*)&q->ins0 = make_int4(T.ins0 + i, T.ins1 + i, T.opFirst + r.nOps * i, r.nOps);
  *(int4*)&q->packed = make_int4(wrapSlot(s0 + i) | (wrapSlot(s1 + i) << 8) | ((wrapSlot(insSlot + i) + 1) << 16), 0, 0, 0);
  RingOp* ops = a.ringOps + S.opBase;
  for (int k = 0; k < r.nOps; k++) {
    const int dst = T.opFirst + r.nOps * i + k;
    if (T.opFirst + k >= S.opCap || dst >= S.opCap) break;   // (the walk has set status 3)
    RingOp op = ops[T.opFirst + k];
    op.slot = wrapSlot(op.slot + i);
    op.insStep += i;
    ops[dst] = op;
  }
}

// ---- the records ---------------------------------------------------------------------------------------------------
__device__ inline double narrowSlotDev(double v) {   // plan.cpp narrowSlot: the rounded float under a quiet-NaN tag
  return __hiloint2double(0x7FF80000, __float_as_int(__double2float_rn(v)));
}

__global__ __launch_bounds__(256) void planExpandKernel(DevPlanArgs a) {
  const int d = blockIdx.y;
  const DevPlanSite S = a.sites[d];
  const int n = S.n;
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int tileBase = u & ~(kFastTile - 1);
  if (blockIdx.x * 256 >= n) return;   // (whole workgroups leave together: barriers below)
  const bool valid = u < n;
  const int uc = valid ? u : n - 1;
  const double* r = S.clim + (size_t)SIPNET_NCLIM * uc;
  const double len = r[0], tair = r[1], tsoil = r[2], par = r[3], precip = r[4], vpd = r[5], vpdSoil = r[6], vPress = r[7],
               wspd = r[8], gdd = r[9], tod = r[10];
  const int yr = S.year[uc], dy = S.day[uc];
  const int yrPrev = uc > 0 ? S.year[uc - 1] : S.trackInit;
  // phenology new year (sipnet.c:811-815): this record's year exceeds the largest seen before it, starting from the first
  // record's (sipnet.c:1524) -- the earlier blocks' maxima (planPrepKernel), the earlier wavefronts', the earlier lanes'
  bool phenNew;
  {
    __shared__ int waveMax[4], prevBlocks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = valid ? yr : INT_MIN;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x = max(x, y);
    }
    int before = __shfl_up(x, 1);
    if (lane == 0) before = INT_MIN;
    if (lane == 63) waveMax[w] = x;
    if (w == 0) {
      const int* info = a.blockInfo + (size_t)d * a.nBlk * 2;
      int p = S.phenInit;
      for (int bb = lane; bb < (int)blockIdx.x; bb += 64) p = max(p, info[2 * bb + 1]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p = max(p, __shfl_xor(p, off));
      if (lane == 0) prevBlocks = p;
    }
    __syncthreads();
    before = max(before, prevBlocks);
    for (int k = 0; k < w; k++) before = max(before, waveMax[k]);
    phenNew = yr > before;
  }
  const DevPlanSeq* seq = a.seq + (size_t)d * a.nT;
  const DevPlanSeq Q = seq[uc];
  const int nOps = Q.nOps, s0 = Q.packed & 0xff, s1 = (Q.packed >> 8) & 0xff, insSlot = (Q.packed >> 16) - 1;
  int ns0 = s0, ns1 = s1;   // the next step's eviction slots (its own on the site's last record)
  if (uc + 1 < n) {
    const int np = seq[uc + 1].packed;
    ns0 = np & 0xff;
    ns1 = (np >> 8) & 0xff;
  }
  const bool newTrack = yr != yrPrev;
  const double dayTime = (double)dy + tod / 24.0;
  const double tsoil10 = tsoil / 10.0;
  const bool tsoilSame = uc > 0 && (S.clim[(size_t)SIPNET_NCLIM * (uc - 1) + 2] / 10.0) == tsoil10;
  // year-to-date GDD pastLeafGrowth() sees (sipnet.c:706-716): this record's alone in a new year, else the sum so far
  const double gddAfter = a.flagGdd ? a.gddAfter[(size_t)d * a.nT + uc] : 0.0;   // trackers.gdd after this record (the host's chain)
  const double cum = newTrack ? gdd : gddAfter;

  // events on this record and the tillage modifier (the host's pass; none: zeros)
  int evFirst = 0, evCount = 0;
  double dTill = 0.0, tillAfter = 0.0;
  if (S.hasEvents) {
    const size_t k = (size_t)d * a.nT + uc;
    evFirst = a.evFirst[k];
    evCount = a.evCount[k];
    dTill = a.dTill[k];
    tillAfter = a.tillAfter[k];
  }

  FastRec f;
  f.len = len;
  f.invLen = 1.0 / len;
  f.tair = tair;
  f.tsoil = tsoil;
  f.negPar = -par;
  f.vpd = vpd;
  f.tillP1 = 1.0 + dTill;
  f.rainRate = precip / len;
  f.sublW = (a.convS * (kEStarSnow - vPress)) * wspd;
  f.evapNum = a.convE * vpdSoil;
  f.invWspd = 1.0 / wspd;
  f.tair10 = tair / 10.0;
  f.tsoil10 = tsoil10;
  f.cumGdd = a.phenMode == 0 ? cum : a.phenMode == 1 ? tsoil : dayTime;
  f.dayTime = dayTime;
  f.w0 = Q.w0;
  const int bits = (phenNew ? FAST_PHEN_NEW_YEAR : 0) | (newTrack ? FAST_TRACK_NEW_YEAR : 0) | (tair > 0 ? FAST_TAIR_POS : 0) |
                   (par > 0 ? FAST_PAR_POS : 0) | ((tsoil < 0 || !a.moistHResp) ? FAST_TSOIL_NEG : 0) |
                   (Q.w1 != 0.0 ? FAST_HAS_W1 : 0) | (dTill != 0.0 ? FAST_HAS_TILL : 0) | (tsoilSame ? FAST_TSOIL_SAME : 0) |
                   (nOps == 1 && insSlot >= 0 ? FAST_RING_REGULAR : 0);
  f.bitsOps = bits | (nOps << 16);
  f.slots = s0 | (s1 << 8) | (ns0 << 16) | (ns1 << 24);
  f.insSlot = insSlot;
  f.evCount = evCount;
  f.w1 = Q.w1;
  f.spareD = 0.0;
  f.log2vpd = 0.0;     // filled from the host's values when a member reads it (plan_device.h)
  f.gddAfter = gddAfter;
  f.tillAfter = tillAfter;
  f.ins0 = Q.ins0;
  f.ins1 = Q.ins1;
  f.opFirst = Q.opFirst;
  f.evFirst = evFirst;
  f.year = yr;
  f.day = dy;

  // the tile's summary (summariseTile, plan.cpp): 16 consecutive lanes
  const int lane = threadIdx.x & 63, g0 = lane & ~(kFastTile - 1), li = lane - g0;
  const int cnt = max(1, min(kFastTile, n - tileBase));   // (lanes of tiles past the site's end compute on a clamped record and store nothing)
  const unsigned validMask = cnt == 16 ? 0xffffu : ((1u << cnt) - 1u);
  auto first = [&](auto x) { return __shfl(x, g0); };
  auto nextSlot = [](int s) { return s + 1 == kSlots ? 0 : s + 1; };
  const int nOpsB = first(nOps);
  bool ok = !(bits & (FAST_PHEN_NEW_YEAR | FAST_TRACK_NEW_YEAR)) && evCount == 0 && insSlot >= 0 && nOps == nOpsB && f.len == first(f.len) &&
            f.invLen == first(f.invLen) && f.w0 == first(f.w0) && f.w1 == first(f.w1);
  {
    const int ps0 = __shfl_up(s0, 1), ps1 = __shfl_up(s1, 1), pIns = __shfl_up(insSlot, 1);
    if (li > 0 && (s0 != nextSlot(ps0) || s1 != nextSlot(ps1) || insSlot != nextSlot(pIns))) ok = false;
  }
  const unsigned okMask = (unsigned)(__ballot(ok && valid) >> g0) & 0xffffu;
  const unsigned dayMask = (unsigned)(__ballot((bits & FAST_PAR_POS) && valid) >> g0) & 0xffffu;
  const bool regular = okMask == validMask && nOpsB >= 1 && nOpsB <= 2;
  double maxGdd = first(f.cumGdd), maxDay = first(f.dayTime);
  for (int k = 1; k < cnt; k++) {
    const double cg = __shfl(f.cumGdd, g0 + k), dt = __shfl(f.dayTime, g0 + k);
    if (cg > maxGdd) maxGdd = cg;
    if (dt > maxDay) maxDay = dt;
  }
  f.tileBits = (int32_t)((regular ? (unsigned)FAST_TILE_REGULAR : 0u) | (dayMask << 16));
  f.tilePad = 0;
  f.tileEndCumGdd = maxGdd;
  f.tileEndDayTime = maxDay;
  for (int k = 0; k < 6; k++) f.pad[k] = 0;
  if (a.narrow) {
    f.tair = narrowSlotDev(f.tair); f.tsoil = narrowSlotDev(f.tsoil); f.negPar = narrowSlotDev(f.negPar); f.vpd = narrowSlotDev(f.vpd);
    f.tillP1 = narrowSlotDev(f.tillP1); f.rainRate = narrowSlotDev(f.rainRate); f.sublW = narrowSlotDev(f.sublW);
    f.evapNum = narrowSlotDev(f.evapNum); f.invWspd = narrowSlotDev(f.invWspd); f.tair10 = narrowSlotDev(f.tair10);
    f.tsoil10 = narrowSlotDev(f.tsoil10); f.log2vpd = narrowSlotDev(f.log2vpd);
  }
  if (valid) a.fast[(size_t)S.site * a.nT + u] = f;
}

__global__ __launch_bounds__(256) void planLog2Kernel(const DevPlanSite* sites, int32_t nT, FastRec* fast, const double* log2vpd,
                                                      int32_t narrow) {
  const DevPlanSite S = sites[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= S.n) return;
  const double v = log2vpd[(size_t)blockIdx.y * nT + t];
  fast[(size_t)S.site * nT + t].log2vpd = narrow ? narrowSlotDev(v) : v;
}
}  // namespace

void launchDevicePlan(const DevPlanArgs& a, int32_t maxSteps, hipStream_t stream) {
  if (a.nDev <= 0 || maxSteps <= 0) return;
  const dim3 wide((maxSteps + 255) / 256, a.nDev);
  hipLaunchKernelGGL(planPrepKernel, wide, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(planSeqKernel, dim3(a.nDev), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(planRunsKernel, wide, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(planExpandKernel, wide, dim3(256), 0, stream, a);
}

void launchDevicePlanLog2(const DevPlanSite* sites, int32_t nDev, int32_t nT, int32_t maxSteps, FastRec* fast, const double* log2vpd,
                          int32_t narrow, hipStream_t stream) {
  if (nDev <= 0 || maxSteps <= 0) return;
  hipLaunchKernelGGL(planLog2Kernel, dim3((maxSteps + 255) / 256, nDev), dim3(256), 0, stream, sites, nT, fast, log2vpd, narrow);
}

}  // namespace sipnet

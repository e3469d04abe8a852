// coop_probes.h -- measurement scaffolding of step_coop.hip (tools/build_variants.py, tools/coop_waits.py,
// tools/coop_placement.py, tools/gpu_phase_sweep.sh, tools/kernel_resources.py).  NOT part of the product: every
// probe needs -DSIPNET_PROBES next to its own switch (tools/build_variants.py adds it), a product build that defines
// a probe switch alone does not compile, and without the switches every macro below is empty.
//   SIPNET_HWID      where each wavefront of the first 4096 workgroups ran (HW_ID, XCC_ID)
//   SIPNET_STAMPS    s_memtime stamps along the carbon wave's step
//   SIPNET_MARKERS   comments around the hot loops in the assembly
//   SIPNET_WAITS     cycles each wave spends inside its hand-over waits
//   SIPNET_NO_STATS  A/B: what the statistics machinery costs a launch that does not use it
//   SIPNET_PH_C / _W / _L / _F, SIPNET_PAD_NOPS   code-phase sweeps (coopCodePhase)
// Included inside namespace sipnet { namespace { ... } } of step_coop.hip.
#if !defined(SIPNET_PROBES) && (defined(SIPNET_HWID) || defined(SIPNET_STAMPS) || defined(SIPNET_MARKERS) || defined(SIPNET_WAITS) || \
                                defined(SIPNET_NO_STATS) || defined(SIPNET_PH_C) || defined(SIPNET_PH_W) || defined(SIPNET_PH_L) ||     \
                                defined(SIPNET_PH_F) || defined(SIPNET_PAD_NOPS))
#error "step_coop.hip: measurement probes need -DSIPNET_PROBES (they are not part of the product build)"
#endif
// -DSIPNET_HWID (diagnostic build): where each wavefront of the first 4096 workgroups ran --
// HW_ID (wave slot, SIMD, CU, shader array / engine) and XCC_ID -- to check that the three waves of
// a workgroup sit on three different SIMDs and where the waves of co-resident workgroups land
#ifdef SIPNET_HWID
__device__ unsigned g_coopHwId[4096 * 4 * 2];  // [chunk][carbon, water, light, factors][HW_ID, XCC_ID]
#define PROBE_HWID(chunkIdx, role)                                                                         \
  if (lane == 0 && (role) >= 0 && (chunkIdx) < 4096) {                                                     \
    unsigned hw, xcc;                                                                                      \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); \
    g_coopHwId[((chunkIdx) * 4 + (role)) * 2] = hw;                                                        \
    g_coopHwId[((chunkIdx) * 4 + (role)) * 2 + 1] = xcc;                                                   \
  }
#else
#define PROBE_HWID(chunkIdx, role)
#endif
#ifdef SIPNET_STAMPS
__device__ unsigned long long g_coopStamps[16];
#define CSTAMP(k)                                                                    \
  {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                               \
    unsigned long long now_;                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);                                               \
    cAcc[k] += now_ - cLast;                                                         \
    cLast = now_;                                                                    \
  }
#define CSTAMP_DECL()                                                                \
  unsigned long long cAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cLast;                      \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cLast)::"memory");
#define CSTAMP_STORE()                                                               \
  if (firstChunk && lane == 0)                                                       \
    for (int k = 0; k < 8; k++) g_coopStamps[k] = cAcc[k];
#else
#define CSTAMP(k)
#define CSTAMP_DECL()
#define CSTAMP_STORE()
#endif
// -DSIPNET_MARKERS (reading the assembly, tools/kernel_resources.py): comments around the hot loops
#ifdef SIPNET_MARKERS
#define MARK(text) asm volatile("; ##### " text);
#else
#define MARK(text)
#endif
// -DSIPNET_WAITS (diagnostic build): cycles each wave spends inside its hand-over waits
#ifdef SIPNET_WAITS
__device__ unsigned long long g_coopWaits[16];
#define WAIT_BEGIN()                                                                 \
  unsigned long long w0_;                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                 \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0_)::"memory");
#define WAIT_END(k)                                                                  \
  {                                                                                  \
    unsigned long long w1_;                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1_)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                               \
    wAcc[k] += w1_ - w0_;                                                            \
  }
#define WAIT_DECL() unsigned long long wAcc[4] = {0, 0, 0, 0}; unsigned long long wT0_; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wT0_)::"memory");
#define WAIT_STORE(base)                                                             \
  if (firstChunk && lane == 0) {                                                     \
    unsigned long long wT1_;                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wT1_)::"memory");     \
    for (int k = 0; k < 3; k++) g_coopWaits[base + k] = wAcc[k];                     \
    g_coopWaits[base + 3] = wT1_ - wT0_;                                             \
  }
#else
#define WAIT_BEGIN()
#define WAIT_END(k)
#define WAIT_DECL()
#define WAIT_STORE(base)
#endif

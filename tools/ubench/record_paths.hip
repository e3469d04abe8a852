// Microbenchmark (gfx950, one lone 64-thread wave per CU): what it costs per step to bring the
// site-uniform record of that step to a wavefront, by path.  Every variant ends in a wait, as the
// step does (the values feed the step's first instructions), so latency is included.
//   bcast128 xK    K broadcast ds_read_b128 (all lanes the same address)        -- round-1 path
//   lanes+readlane one ds_read_b32 with lane i reading dword i of the 256-B record, then N
//                  v_readlane_b32 into SGPRs
//   reg+readlane   the record fields of 64 steps held in VGPRs (lane = step), N v_readlane_b32
//                  with the step's lane index in an SGPR: no LDS at all
//   mailbox        flag + five per-lane ds_read_b64 (the W -> C hand-over)
// Build + run: hipcc --offload-arch=gfx950 -O3 record_paths.hip -o /tmp/rp && /tmp/rp
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 20000

template <int K>
__global__ void k_bcast(double* out) {
  __shared__ alignas(16) double buf[16 * 32];
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = i * 0.25;
  __syncthreads();
  typedef double d2 __attribute__((ext_vector_type(2)));
  double acc = 0;
  for (int i = 0; i < N_ITER; i++) {
    const unsigned a = (unsigned)(size_t)(buf + (i & 15) * 32);
    d2 q[8];
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[k]) : "v"(a), "n"(k * 16));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < K; k++) { asm volatile("" : "+v"(q[k])); acc = __builtin_fma(q[k].x, q[k].y, acc); }
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}

template <int N>
__global__ void k_lanes(double* out) {
  __shared__ alignas(16) int buf[16 * 64];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = i;
  __syncthreads();
  double acc = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
    const unsigned a = (unsigned)(size_t)(buf + (i & 15) * 64 + threadIdx.x);
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
#pragma unroll
    for (int k = 0; k < N; k += 2) {
      const int lo = __builtin_amdgcn_readlane(v, k), hi = __builtin_amdgcn_readlane(v, k + 1);
      const double s = __hiloint2double(hi & 0x3ff, lo);   // a (denormal-free) double in an SGPR pair
      acc = __builtin_fma(acc, 0.999, s);
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}

template <int N>
__global__ void k_regs(double* out, const int* src) {
  int f[N];
#pragma unroll
  for (int k = 0; k < N; k++) f[k] = src[k * 64 + threadIdx.x];
  double acc = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
    const int lane = i & 63;   // uniform: lives in an SGPR
#pragma unroll
    for (int k = 0; k < N; k += 2) {
      const int lo = __builtin_amdgcn_readlane(f[k], lane), hi = __builtin_amdgcn_readlane(f[k + 1], lane);
      const double s = __hiloint2double(hi & 0x3ff, lo);
      acc = __builtin_fma(acc, 0.999, s);
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}

__global__ void k_mailbox(double* out) {
  __shared__ double box[2][5][64];
  __shared__ int flag;
  for (int i = threadIdx.x; i < 640; i += 64) (&box[0][0][0])[i] = i;
  if (threadIdx.x == 0) flag = 1 << 30;
  __syncthreads();
  double acc = 0;
  for (int i = 0; i < N_ITER; i++) {
    const unsigned a = (unsigned)(size_t)&box[i & 1][0][threadIdx.x], fa = (unsigned)(size_t)&flag;
    double g0, g1, g2, g3, g4;
    int f;
    asm volatile("ds_read_b32 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b64 %2, %7 offset:512\n\t"
                 "ds_read_b64 %3, %7 offset:1024\n\tds_read_b64 %4, %7 offset:1536\n\t"
                 "ds_read_b64 %5, %7 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(g4) : "v"(fa), "v"(a) : "memory");
    if (__builtin_amdgcn_readfirstlane(f) < i) break;
    acc += g0 + g1 + g2 + g3 + g4;
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}

template <class F> void timeit(F f, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f ms  ~%7.1f cycles (at 2.4 GHz) per step\n", name, ms, ms * 1e-3 * 2.4e9 / N_ITER);
  fflush(stdout);
}
int main() {
  double* d; hipMalloc(&d, 256 * 64 * 8);
  int* src; hipMalloc(&src, 64 * 64 * 4); hipMemset(src, 1, 64 * 64 * 4);
  const int G = 160;
  timeit([&] { k_bcast<4><<<G, 64>>>(d); }, "bcast128 x4 + wait + 4 fma        (C today)");
  timeit([&] { k_bcast<8><<<G, 64>>>(d); }, "bcast128 x8 + wait + 8 fma        (W today)");
  timeit([&] { k_lanes<14><<<G, 64>>>(d); }, "lanes b32 + wait + 14 readlane + 7 fma");
  timeit([&] { k_lanes<28><<<G, 64>>>(d); }, "lanes b32 + wait + 28 readlane + 14 fma");
  timeit([&] { k_regs<14><<<G, 64>>>(d, src); }, "regs: 14 readlane + 7 fma");
  timeit([&] { k_regs<28><<<G, 64>>>(d, src); }, "regs: 28 readlane + 14 fma");
  timeit([&] { k_mailbox<<<G, 64>>>(d); }, "mailbox: flag + 5 x ds_read_b64 + wait + 5 add");
  return 0;
}

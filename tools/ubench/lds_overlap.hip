// Can a lone wavefront hide broadcast LDS reads behind fp64 VALU work?  (dev microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_ITER 20000
typedef double d2 __attribute__((ext_vector_type(2)));

// 64 dependent FMAs per iteration, no LDS
__global__ void k_base(double* out, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 64; k++) x = __builtin_fma(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
// 9 broadcast ds_read_b128 issued back to back at the top, consumed in the NEXT iteration
template <int MODE>
__global__ void k_lds(double* out, double a, double b, int off) {
  __shared__ double buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 1e-9 * i;
  __syncthreads();
  double x = threadIdx.x;
  d2 c0 = {0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0, c8 = c0;
  for (int i = 0; i < N_ITER; i++) {
    const unsigned addr = (unsigned)(size_t)(buf + ((i + off) & 15) * 32);
    d2 n0, n1, n2, n3, n4, n5, n6, n7, n8;
    const double s = c0.x + c1.y + c2.x + c3.y + c4.x + c5.y + c6.x + c7.y + c8.x;  // 8 adds
    x += s;
    if (MODE == 0) {  // all reads at the top
      asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:16\n\tds_read_b128 %2, %9 offset:32\n\t"
                   "ds_read_b128 %3, %9 offset:48\n\tds_read_b128 %4, %9 offset:64\n\tds_read_b128 %5, %9 offset:80\n\t"
                   "ds_read_b128 %6, %9 offset:96\n\tds_read_b128 %7, %9 offset:112\n\tds_read_b128 %8, %9 offset:128"
                   : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5), "=&v"(n6), "=&v"(n7), "=&v"(n8)
                   : "v"(addr));
#pragma unroll
      for (int k = 0; k < 64; k++) x = __builtin_fma(x, a, b);
    } else {  // one read every 7 FMAs
#define RD(n, o) asm volatile("ds_read_b128 %0, %1 offset:" #o : "=v"(n) : "v"(addr));
#define F7 _Pragma("unroll") for (int k = 0; k < 7; k++) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)); }
      RD(n0, 0) F7 RD(n1, 16) F7 RD(n2, 32) F7 RD(n3, 48) F7 RD(n4, 64) F7 RD(n5, 80) F7 RD(n6, 96) F7
      RD(n7, 112) F7 RD(n8, 128) F7
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4; c5 = n5; c6 = n6; c7 = n7; c8 = n8;
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
// the whole 256-byte record in ONE ds_read_b32 (lane i reads dword i), 36 dwords fanned out
// with v_readlane into SGPRs, consumed as scalar operands
__global__ void k_readlane(double* out, double a, double b, int off) {
  __shared__ double buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 1e-9 * i;
  __syncthreads();
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
    const unsigned addr = (unsigned)(size_t)(buf + ((i + off) & 15) * 32) + threadIdx.x * 4;
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 18; k++) {
      const unsigned lo = __builtin_amdgcn_readlane(v, 2 * k), hi = __builtin_amdgcn_readlane(v, 2 * k + 1);
      const double s = __hiloint2double((int)hi, (int)lo);
      acc += s;   // 18 adds with a scalar operand
    }
    x += acc;
#pragma unroll
    for (int k = 0; k < 46; k++) x = __builtin_fma(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
// scalar loads from global memory: 2 x s_load_dwordx16 + 1 x s_load_dwordx4 per iteration,
// issued at the top, waited at the bottom
__global__ void k_sload(double* out, const double* g, double a, double b, int off) {
  double x = threadIdx.x;
  double carry = 0;
  for (int i = 0; i < N_ITER; i++) {
    const double* p = g + ((i + off) & 1023) * 32;
    typedef double d8 __attribute__((ext_vector_type(8)));
    d8 r0, r1;
    d2 r2;
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x80"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2) : "s"(p));
    x += carry;
#pragma unroll
    for (int k = 0; k < 64; k++) x = __builtin_fma(x, a, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    carry = r0[0] + r0[7] + r1[0] + r1[7] + r2[1];
  }
  out[threadIdx.x + blockIdx.x * 64] = x + carry;
}

template <class F> void timeit(F f, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.3f ms  %7.1f cycles/iter\n", name, ms, ms * 1e-3 * 2.4e9 / N_ITER); fflush(stdout);
}
int main(int argc, char** argv) {
  int sel = argc > 1 ? atoi(argv[1]) : -1; int id = 0;
#define RUN(x) if (sel < 0 || sel == id) { x; } id++;
  double* d; hipMalloc(&d, 256 * 64 * 8 * 4);
  double* g; hipMalloc(&g, 1024 * 32 * 8); hipMemset(g, 0, 1024 * 32 * 8);
  const int G = 160;
  RUN(timeit([&] { k_base<<<G, 64>>>(d, 0.999, 0.001); }, "64 dependent v_fma_f64"));
  RUN(timeit([&] { k_lds<0><<<G, 64>>>(d, 0.999, 0.001, 1); }, "9 ds_read_b128 at top + 8 add + 64 fma, used next iter"));
  RUN(timeit([&] { k_lds<1><<<G, 64>>>(d, 0.999, 0.001, 1); }, "9 ds_read_b128 interleaved (1 per 7 fma) + 8 add + 64 fma"));
  RUN(timeit([&] { k_readlane<<<G, 64>>>(d, 0.999, 0.001, 1); }, "1 ds_read_b32 + 36 v_readlane + 18 add + 46 fma"));
  RUN(timeit([&] { k_sload<<<G, 64>>>(d, g, 0.999, 0.001, 1); }, "2 s_load_x16 + 1 x4 at top + 64 fma + 5 add, used next"));
  return 0;
}

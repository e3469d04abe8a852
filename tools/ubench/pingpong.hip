// Cost of handing 64 doubles between two wavefronts of one workgroup through LDS + s_barrier
// (the per-step exchange a cooperative step kernel would need).  dev microbenchmark
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_ITER 20000

// serial chain: wave0 does NF fmas, hands x to wave1, which does NF fmas, hands back
template <int NF, int NVAL>
__global__ __launch_bounds__(128) void k_pingpong(double* out, double a, double b) {
  __shared__ double box[2][NVAL][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x = lane;
  for (int i = 0; i < N_ITER; i++) {
    if (w == 0) {
#pragma unroll
      for (int k = 0; k < NF; k++) x = __builtin_fma(x, a, b);
#pragma unroll
      for (int v = 0; v < NVAL; v++) box[0][v][lane] = x + v;
      __syncthreads();           // hand over to wave 1
      __syncthreads();           // wait for wave 1
      double s = 0;
#pragma unroll
      for (int v = 0; v < NVAL; v++) s += box[1][v][lane];
      x = s;
    } else {
      __syncthreads();
      double s = 0;
#pragma unroll
      for (int v = 0; v < NVAL; v++) s += box[0][v][lane];
      x = s;
#pragma unroll
      for (int k = 0; k < NF; k++) x = __builtin_fma(x, a, b);
#pragma unroll
      for (int v = 0; v < NVAL; v++) box[1][v][lane] = x + v;
      __syncthreads();
    }
  }
  out[threadIdx.x + blockIdx.x * 128] = x;
}
// same chain in ONE wave, for reference
template <int NF, int NVAL>
__global__ __launch_bounds__(64) void k_single(double* out, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < NF; k++) x = __builtin_fma(x, a, b);
    double s = 0;
#pragma unroll
    for (int v = 0; v < NVAL; v++) s += x + v;
    x = s;
#pragma unroll
    for (int k = 0; k < NF; k++) x = __builtin_fma(x, a, b);
    s = 0;
#pragma unroll
    for (int v = 0; v < NVAL; v++) s += x + v;
    x = s;
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
// parallel halves: both waves do NF fmas at the same time, exchange, repeat (what a split step
// gains when the two halves are independent)
template <int NF, int NVAL>
__global__ __launch_bounds__(128) void k_parallel(double* out, double a, double b) {
  __shared__ double box[2][NVAL][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x = lane;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < NF; k++) x = __builtin_fma(x, a, b);
#pragma unroll
    for (int v = 0; v < NVAL; v++) box[w][v][lane] = x + v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int v = 0; v < NVAL; v++) s += box[w ^ 1][v][lane];
    x += s;
    __syncthreads();   // the box is rewritten next iteration
  }
  out[threadIdx.x + blockIdx.x * 128] = x;
}
__global__ void k_fma_dep(double* out, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 64; k++) x = __builtin_fma(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
template <class F> void timeit(F f, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-66s %8.3f ms  %7.1f cycles/iter\n", name, ms, ms * 1e-3 * 2.4e9 / N_ITER); fflush(stdout);
}
int main(int argc, char** argv) {
  int sel = argc > 1 ? atoi(argv[1]) : -1; int id = 0;
#define RUN(x) if (sel < 0 || sel == id) { x; } id++;
  double* d; hipMalloc(&d, 256 * 128 * 8 * 4);
  const int G = 160;
  RUN(timeit([&] { k_single<50, 2><<<G, 64>>>(d, 0.999, 0.001); }, "one wave: 2 x (50 fma + 2 values)"));
  RUN(timeit([&] { k_pingpong<50, 2><<<G, 128>>>(d, 0.999, 0.001); }, "two waves serial chain: 2 x (50 fma, hand over 2 values)"));
  RUN(timeit([&] { k_pingpong<50, 4><<<G, 128>>>(d, 0.999, 0.001); }, "two waves serial chain: 2 x (50 fma, hand over 4 values)"));
  RUN(timeit([&] { k_pingpong<0, 2><<<G, 128>>>(d, 0.999, 0.001); }, "two waves: pure hand-over of 2 values, both directions"));
  RUN(timeit([&] { k_parallel<100, 2><<<G, 128>>>(d, 0.999, 0.001); }, "two waves parallel: 100 fma each + exchange 2 + 2 barriers"));
  RUN(timeit([&] { k_single<100, 2><<<G, 64>>>(d, 0.999, 0.001); }, "one wave: 2 x (100 fma + 2 values)"));
  RUN(timeit([&] { k_fma_dep<<<G, 64>>>(d, 0.999, 0.001); }, "64 dependent fma, 1 wave per workgroup"));
  RUN(timeit([&] { k_fma_dep<<<G, 192>>>(d, 0.999, 0.001); }, "64 dependent fma, 3 waves per workgroup (same SIMD?)"));
  RUN(timeit([&] { k_fma_dep<<<G, 256>>>(d, 0.999, 0.001); }, "64 dependent fma, 4 waves per workgroup"));
  RUN(timeit([&] { k_fma_dep<<<G, 512>>>(d, 0.999, 0.001); }, "64 dependent fma, 8 waves per workgroup"));
  return 0;
}

// Does a wave64 fp64 / fp32 instruction get cheaper when fewer lanes are active?  (lone wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 20000
__global__ void k64(double* out, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = __builtin_fma(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k32(float* out, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = __builtin_fmaf(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
int main() {
  double* d; (void)hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int lanes : {64, 32, 16, 8}) {
    float ms;
    k64<<<160, lanes>>>(d, 0.999, 0.001); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k64<<<160, lanes>>>(d, 0.999, 0.001); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("fp64 fma, %2d active lanes: %.3f ms -> %.2f cycles/instr\n", lanes, ms, ms * 1e-3 * 2.4e9 / N_ITER / 16); fflush(stdout);
    k32<<<160, lanes>>>((float*)d, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k32<<<160, lanes>>>((float*)d, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("fp32 fma, %2d active lanes: %.3f ms -> %.2f cycles/instr\n", lanes, ms, ms * 1e-3 * 2.4e9 / N_ITER / 16); fflush(stdout);
  }
  return 0;
}

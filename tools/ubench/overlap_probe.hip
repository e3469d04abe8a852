// dev probe: what can the host do on stream B while a long kernel occupies the device on stream A?
// (pinned / pageable H2D copies, a small kernel, synchronous hipMemcpy on the null stream, event sync)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
__global__ void tiny(double* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.0;
}
static double nowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
  const int fill = argc > 1 ? atoi(argv[1]) : 256;      // workgroups of the long kernel
  const int ldsKB = argc > 2 ? atoi(argv[2]) : 0;
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const size_t N = 16 << 20;   // 128 MB of doubles
  double *d, *pinned;
  hipMalloc(&d, N * 8);
  hipHostMalloc(&pinned, N * 8);
  std::vector<double> pageable(N, 1.0);
  int* flag;
  hipMalloc(&flag, 4);
  spin<<<1, 64, 0, a>>>(1000, flag);
  hipDeviceSynchronize();
  auto longKernel = [&]() { hipLaunchKernelGGL(spin, dim3(fill), dim3(512), ldsKB * 1024, a, 100000000LL /* ~ 40-50 ms */, flag); };
  struct Test { const char* name; int kind; };
  const Test tests[] = {{"hipMemcpyAsync pinned 128 MB on B + sync B", 0}, {"hipMemcpyAsync pageable 128 MB on B + sync B", 1},
                        {"tiny kernel on B + sync B", 2}, {"hipMemcpy (sync, null stream) pinned 128 MB", 3},
                        {"hipMemcpy (sync, null stream) pageable 128 MB", 4}, {"tiny kernel on the NULL stream + sync null", 5},
                        {"hipEventRecord on B + hipEventSynchronize", 6}};
  for (const Test& t : tests) {
    hipDeviceSynchronize();
    longKernel();
    const double t0 = nowMs();
    hipEvent_t ev;
    switch (t.kind) {
      case 0: hipMemcpyAsync(d, pinned, N * 8, hipMemcpyHostToDevice, b); hipStreamSynchronize(b); break;
      case 1: hipMemcpyAsync(d, pageable.data(), N * 8, hipMemcpyHostToDevice, b); hipStreamSynchronize(b); break;
      case 2: hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, b, d, 16384); hipStreamSynchronize(b); break;
      case 3: hipMemcpy(d, pinned, N * 8, hipMemcpyHostToDevice); break;
      case 4: hipMemcpy(d, pageable.data(), N * 8, hipMemcpyHostToDevice); break;
      case 5: hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, 0, d, 16384); hipStreamSynchronize(0); break;
      case 6: hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventRecord(ev, b); hipEventSynchronize(ev); hipEventDestroy(ev); break;
    }
    const double t1 = nowMs();
    hipStreamSynchronize(a);
    const double t2 = nowMs();
    printf("%-55s host returned after %7.2f ms; long kernel (fill %d x 512 threads, %d KB LDS) ended at %7.2f ms\n", t.name, t1 - t0, fill, ldsKB, t2 - t0);
  }
  return 0;
}

// dev (GPU box): what a pinned staging buffer costs -- hipHostMalloc of N MB, first touch, H2D copy
// rate from it, against malloc + first touch + H2D from pageable memory.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/pin_cost.hip -o /tmp/pin_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double nowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 143;
  const size_t n = mb << 20;
  void* d;
  hipMalloc(&d, n);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    double t0 = nowMs();
    void* h;
    hipHostMalloc(&h, n, hipHostMallocDefault);
    double t1 = nowMs();
    memset(h, 1, n);
    double t2 = nowMs();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    double t3 = nowMs();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    double t4 = nowMs();
    hipHostFree(h);
    double t5 = nowMs();
    printf("pinned   %zu MB: alloc %.2f ms, first touch %.2f, copy %.2f, copy again %.2f (%.1f GB/s), free %.2f\n", mb,
           t1 - t0, t2 - t1, t3 - t2, t4 - t3, n / (t4 - t3) * 1e-6, t5 - t4);
    t0 = nowMs();
    h = malloc(n);
    t1 = nowMs();
    memset(h, 1, n);
    t2 = nowMs();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    t3 = nowMs();
    memset(h, 2, n);
    t4 = nowMs();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    t5 = nowMs();
    free(h);
    printf("pageable %zu MB: alloc %.2f ms, first touch %.2f, copy %.2f, touch again %.2f, copy again %.2f (%.1f GB/s)\n", mb,
           t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, n / (t5 - t4) * 1e-6);
  }
  return 0;
}

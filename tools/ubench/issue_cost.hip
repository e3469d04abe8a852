// Microbenchmarks of single-wavefront issue costs on gfx950 (one 64-thread block per CU):
// what one lone wave pays per instruction kind.  Build+run: tools/ubench/run.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define N_ITER 20000

__global__ void k_fma_dep(double* out, double a, double b) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = __builtin_fma(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_fma_indep(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b);
      x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3;
}
__global__ void k_fma32_dep(float* out, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = __builtin_fmaf(x, a, b);
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_fma32_indep(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b);
      x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3;
}
// 16 taken scalar branches per iteration (each jumps over one never-executed instruction)
__global__ void k_branch(double* out, int flag) {
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      asm volatile(
          "s_cmp_eq_u32 %1, 0\n\t"
          "s_cbranch_scc1 1f\n\t"
          "v_add_f64 %0, %0, 1.0\n\t"
          "1:\n\t"
          : "+v"(x) : "s"(flag));
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_branch_nt(double* out, int flag) {  // same, branch NOT taken (flag != 0)
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      asm volatile(
          "s_cmp_eq_u32 %1, 0\n\t"
          "s_cbranch_scc1 1f\n\t"
          "v_add_f64 %0, %0, 1.0\n\t"
          "1:\n\t"
          : "+v"(x) : "s"(flag));
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_execz(double* out, double thr) {  // divergent-style skip: saveexec + execz taken
  double x = threadIdx.x;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      asm volatile(
          "v_cmp_gt_f64 vcc, %1, %0\n\t"
          "s_and_saveexec_b64 s[20:21], vcc\n\t"
          "s_cbranch_execz 1f\n\t"
          "v_add_f64 %0, %0, 1.0\n\t"
          "1:\n\t"
          "s_or_b64 exec, exec, s[20:21]\n\t"
          : "+v"(x) : "v"(thr) : "vcc", "s20", "s21");
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_salu(double* out, int a) {
  int s = a;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
  }
  out[threadIdx.x + blockIdx.x * 64] = s;
}
__global__ void k_exp2pieces(double* out, double a) {  // rndne + cvt + ldexp per iteration x4
  double x = threadIdx.x * 0.001 + a;
  double acc = 0;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double n = __builtin_rint(x);
      int e = (int)n;
      acc += __builtin_amdgcn_ldexp(x - n, e & 3);
      x += 0.37;
    }
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}
__global__ void k_rcp(double* out, double a) {
  double x = threadIdx.x + a;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = __builtin_amdgcn_rcp(x) + 1.5;
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_cndmask(double* out, double a) {
  double x = threadIdx.x + a;
  for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) { x = x > a ? x * 0.5 : x + 1.0; }
  }
  out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void k_lds(double* out, int off) {
  __shared__ double buf[512];
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = i;
  __syncthreads();
  typedef double d2 __attribute__((ext_vector_type(2)));
  double acc = 0;
  for (int i = 0; i < N_ITER; i++) {
    const d2* p = (const d2*)(buf + ((i + off) & 15) * 32);
    d2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5], q6 = p[6], q7 = p[7];
    d2 q8 = p[8], q9 = p[9], q10 = p[10], q11 = p[11];
    acc += q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y + q8.x + q9.y + q10.x + q11.y;
  }
  out[threadIdx.x + blockIdx.x * 64] = acc;
}

template <class F> double timeit(F f, const char* name, double perIter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / N_ITER / perIter;
  printf("%-34s %8.3f ms  ~%6.2f cycles (at 2.4 GHz) per item\n", name, ms, cyc); fflush(stdout);
  return cyc;
}
int main(int argc, char** argv) {
  int sel = argc > 1 ? atoi(argv[1]) : -1; int id = 0;
#define RUN(x) if (sel < 0 || sel == id) { x; } id++;
  double* d; hipMalloc(&d, 256 * 64 * 8 * 4);
  const int G = 160;  // one wave per CU on 160 CUs, like the c10k launch
  RUN(timeit([&] { k_fma_dep<<<G, 64>>>(d, 0.999, 0.001); }, "v_fma_f64 dependent chain", 16));
  RUN(timeit([&] { k_fma_indep<<<G, 64>>>(d, 0.999, 0.001); }, "v_fma_f64 4 independent chains", 16));
  RUN(timeit([&] { k_fma32_dep<<<G, 64>>>((float*)d, 0.999f, 0.001f); }, "v_fma_f32 dependent chain", 16));
  RUN(timeit([&] { k_fma32_indep<<<G, 64>>>((float*)d, 0.999f, 0.001f); }, "v_fma_f32 4 independent chains", 16));
  RUN(timeit([&] { k_branch<<<G, 64>>>(d, 0); }, "s_cmp+s_cbranch TAKEN", 16));
  RUN(timeit([&] { k_branch_nt<<<G, 64>>>(d, 1); }, "s_cmp+s_cbranch not taken +v_add", 16));
  RUN(timeit([&] { k_execz<<<G, 64>>>(d, -1.0); }, "v_cmp+saveexec+execz TAKEN+s_or", 16));
  RUN(timeit([&] { k_execz<<<G, 64>>>(d, 1e300); }, "v_cmp+saveexec+execz fallthrough", 16));
  RUN(timeit([&] { k_salu<<<G, 64>>>(d, 1); }, "s_add_i32", 16));
  RUN(timeit([&] { k_exp2pieces<<<G, 64>>>(d, 0.1); }, "rndne+cvt+and+ldexp+sub+2add group", 4));
  RUN(timeit([&] { k_rcp<<<G, 64>>>(d, 0.1); }, "v_rcp_f64 + v_add_f64 (dependent)", 16));
  RUN(timeit([&] { k_cndmask<<<G, 64>>>(d, 0.1); }, "cmp+mul+add+2cndmask (dependent)", 16));
  RUN(timeit([&] { k_lds<<<G, 64>>>(d, 1); }, "12 x ds_read_b128 bcast + 12 adds", 1));
  // two waves per SIMD for comparison
  RUN(timeit([&] { k_fma_dep<<<2048, 64>>>(d, 0.999, 0.001); }, "v_fma_f64 dep, 2 waves/SIMD", 16));
  RUN(timeit([&] { k_fma32_dep<<<2048, 64>>>((float*)d, 0.999f, 0.001f); }, "v_fma_f32 dep, 2 waves/SIMD", 16));
  return 0;
}

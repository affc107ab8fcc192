// How many cycles does one wave64 VALU instruction occupy a SIMD, and what do SQ_INSTS_VALU / SQ_WAVES count?
// Every wave runs ITER x 8 independent v_fma_f32 (or v_add_u32, or v_fma_f64); the grid puts W waves on every SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate;  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -- ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  double d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
  for (int i = 0; i < ITER; i++) {
    if (KIND == 0)
      asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                   "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (KIND == 1)
      asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n"
                   "v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (KIND == 2)
      asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                   "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.f) out[0] = 1;
}
template <int KIND>
static void run(const char *name, int waves_per_simd, float *out, int cus, double ghz) {
  int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double inst_per_simd = (double)waves_per_simd * ITER * 8;
  printf("%-10s %d waves/SIMD: %.3f ms -> %.2f cycles per wave64 instruction at %.2f GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz);
}
int main() {
  float *out;
  hipMalloc(&out, 4);
  int cus = 0, khz = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  double ghz = khz * 1e-6;
  printf("CUs %d, clock %.3f GHz\n", cus, ghz);
  for (int w : {1, 2, 8}) {
    run<0>("v_fma_f32", w, out, cus, ghz);
    run<1>("v_add_u32", w, out, cus, ghz);
    run<2>("v_fma_f64", w, out, cus, ghz);
  }
  return 0;
}

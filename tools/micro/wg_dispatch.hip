// How fast does the chip START workgroups?  Kernels that do (almost) nothing, 256 threads, with 0 / 18 / 35 KiB of
// static LDS, grids of 16384 and 65536 workgroups -- the tile kernels of dt_hydro.hip / dt_tiles.hip launch one
// workgroup per 64 x 64 tile and all run at ~55 M workgroups/s whatever they do.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wg_dispatch.hip -o wg_dispatch && ./wg_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_WORDS>
__global__ __launch_bounds__(256) void k_touch(float *out, int work) {
  __shared__ float s[LDS_WORDS > 0 ? LDS_WORDS : 1];
  float a = (float)threadIdx.x;
  if (LDS_WORDS > 0) {
    s[threadIdx.x] = a;
    __syncthreads();
    a += s[(threadIdx.x + 1) & 255];
  }
  for (int i = 0; i < work; i++) a = a * 1.0001f + 0.5f;  // a dependent chain of `work` FMAs
  if (a == 12345.678f) out[blockIdx.x] = a;               // (never true: nothing is written)
}
template <int LDS_WORDS>
static void run(const char *name, float *d, int blocks, int work) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_touch<LDS_WORDS>, dim3(blocks), dim3(256), 0, 0, d, work);
  hipEventRecord(e0);
  const int R = 20;
  for (int rep = 0; rep < R; rep++) hipLaunchKernelGGL(k_touch<LDS_WORDS>, dim3(blocks), dim3(256), 0, 0, d, work);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= R;
  printf("%-10s %6d workgroups, %5d dependent FMAs per thread: %.3f ms = %.1f M workgroups/s\n", name, blocks, work, ms,
         blocks / ms / 1e3);
}
int main() {
  float *d;
  hipMalloc(&d, 1 << 20);
  for (int work : {0, 1000, 4000}) {
    for (int blocks : {16384, 65536}) {
      run<0>("no LDS", d, blocks, work);
      run<4608>("18 KiB", d, blocks, work);
      run<8960>("35 KiB", d, blocks, work);
    }
  }
  return 0;
}

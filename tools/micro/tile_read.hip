// What does it cost to READ a raster tile by tile?  One workgroup of 256 threads per tile of 4096 floats, all of a
// thread's loads issued before the first is used, a trivial reduction, one store per workgroup.  Tile shapes 64 x 64
// (the hydro / flow kernels), 128 x 32, 256 x 16 (the stencil's), with and without the one-cell halo; raster 16384^2.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_read.hip -o tile_read && ./tile_read
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TW, int TH, int HALO, int LDSPAD>
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ src, int W, int H, float *__restrict__ out) {
  constexpr int WW = TW + 2 * HALO, WH = TH + 2 * HALO, N = (WW * WH + 255) / 256;
  __shared__ float s[LDSPAD > 0 ? LDSPAD : 1];
  const int tiles_x = W / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TH - HALO, x0 = tx * TW - HALO;
  float v[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int i = threadIdx.x + 256 * k;
    v[k] = 0.0f;
    if (i < WW * WH) {
      const int r = i / WW, c = i - r * WW;
      const int y = min(max(y0 + r, 0), H - 1), x = min(max(x0 + c, 0), W - 1);
      v[k] = src[(long long)y * W + x];
    }
  }
  float a = 0.0f;
#pragma unroll
  for (int k = 0; k < N; k++) a += v[k];
  if (LDSPAD > 0) {
    s[threadIdx.x] = a;
    __syncthreads();
    a += s[(threadIdx.x + 7) & 255];
  }
  if (a == 12345.678f) out[blockIdx.x] = a;
}
template <int TW, int TH, int HALO, int LDSPAD>
static void run(const char *name, const float *d, int W, int H, float *o) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = (W / TW) * (H / TH);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_tile<TW, TH, HALO, LDSPAD>), dim3(blocks), dim3(256), 0, 0, d, W, H, o);
  hipEventRecord(e0);
  const int R = 10;
  for (int rep = 0; rep < R; rep++) hipLaunchKernelGGL((k_tile<TW, TH, HALO, LDSPAD>), dim3(blocks), dim3(256), 0, 0, d, W, H, o);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= R;
  printf("%-28s %.3f ms = %.2f TB/s of raster bytes\n", name, ms, (double)W * H * 4 / ms / 1e9);
}
int main() {
  const int W = 16384, H = 16384;
  float *d, *o;
  hipMalloc(&d, (size_t)W * H * 4);
  hipMalloc(&o, 1 << 22);
  hipMemset(d, 0, (size_t)W * H * 4);
  run<64, 64, 0, 0>("64 x 64", d, W, H, o);
  run<64, 64, 1, 0>("64 x 64 + halo", d, W, H, o);
  run<64, 64, 1, 8960>("64 x 64 + halo, 35 KiB LDS", d, W, H, o);
  run<128, 32, 0, 0>("128 x 32", d, W, H, o);
  run<128, 32, 1, 0>("128 x 32 + halo", d, W, H, o);
  run<256, 16, 0, 0>("256 x 16", d, W, H, o);
  run<256, 16, 1, 0>("256 x 16 + halo", d, W, H, o);
  run<256, 16, 1, 8960>("256 x 16 + halo, 35 KiB LDS", d, W, H, o);
  return 0;
}

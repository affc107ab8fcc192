// Latency of a chain of dependent returning 64-bit atomics on global memory, by scope (one lane, one workgroup):
// what one hop of k_fa_reduce's countdown costs.  hipcc --offload-arch=gfx950 -O3 atomic_chain.hip -o atomic_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SCOPE>
__global__ void chain(unsigned long long *p, int hops, int stride, long long *out) {
  unsigned long long a = 1, idx = 0;
  long long t0 = wall_clock64();
  for (int i = 0; i < hops; i++) {
    unsigned long long old = __hip_atomic_fetch_add(&p[idx], a, __ATOMIC_RELAXED, SCOPE);
    idx = (idx + stride + (old & 1)) % ((size_t)hops * stride);  // depends on the returned value
    a += old & 1;
  }
  long long t1 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = (long long)a;
}
__global__ void loads(const unsigned long long *p, int hops, int stride, long long *out) {
  unsigned long long idx = 0, a = 0;
  long long t0 = wall_clock64();
  for (int i = 0; i < hops; i++) {
    unsigned long long v = __builtin_nontemporal_load(&p[idx]);
    idx = (idx + stride + (v & 1)) % ((size_t)hops * stride);
    a += v;
  }
  long long t1 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = (long long)a;
}
int main() {
  const int hops = 2000, stride = 4096 / 8 + 16;
  unsigned long long *p;
  long long *out, h[2];
  hipMalloc(&p, (size_t)hops * stride * 8);
  hipMalloc(&out, 16);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(p, 0, (size_t)hops * stride * 8);
    hipLaunchKernelGGL(chain<__HIP_MEMORY_SCOPE_AGENT>, dim3(1), dim3(1), 0, 0, p, hops, stride, out);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("agent-scope atomic chain:     %.1f ns / hop\n", (double)h[0] / hops * 1e6 / rate);
    hipMemset(p, 0, (size_t)hops * stride * 8);
    hipLaunchKernelGGL(chain<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(1), dim3(1), 0, 0, p, hops, stride, out);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("workgroup-scope atomic chain: %.1f ns / hop\n", (double)h[0] / hops * 1e6 / rate);
    hipMemset(p, 0, (size_t)hops * stride * 8);
    hipLaunchKernelGGL(chain<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(1), dim3(1), 0, 0, p, hops, stride, out);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("system-scope atomic chain:    %.1f ns / hop\n", (double)h[0] / hops * 1e6 / rate);
    hipLaunchKernelGGL(loads, dim3(1), dim3(1), 0, 0, p, hops, stride, out);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("dependent load chain:         %.1f ns / hop\n", (double)h[0] / hops * 1e6 / rate);
  }
  return 0;
}

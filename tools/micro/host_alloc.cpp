// How fast can page-locked host memory for the output rasters be had, and how fast do device-to-host copies into it
// run?  (a) hipHostMalloc; (b) anonymous mmap + MADV_HUGEPAGE + first touch by T threads + hipHostRegister.
//   hipcc -O2 tools/micro/host_alloc.cpp -o /tmp/host_alloc -lpthread && /tmp/host_alloc [GiB per block] [blocks] [threads]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n, int threads) {
  std::vector<std::thread> th;
  size_t per = (n / threads + 4095) & ~(size_t)4095;
  for (int t = 0; t < threads; t++)
    th.emplace_back([=] { for (size_t o = per * t; o < n && o < per * (t + 1); o += 4096) p[o] = 0; });
  for (auto &t : th) t.join();
}
int main(int argc, char **argv) {
  size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 1.0) * (1ull << 30);
  int blocks = argc > 2 ? atoi(argv[2]) : 4, threads = argc > 3 ? atoi(argv[3]) : 8;
  void *dev;
  CK(hipMalloc(&dev, bytes));
  CK(hipMemset(dev, 1, bytes));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  auto d2h = [&](void *h, const char *what) {
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      CK(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      printf("   %s d2h #%d: %.1f GB/s\n", what, rep, bytes / (now() - t0) / 1e9);
    }
  };
  for (int b = 0; b < blocks; b++) {
    double t0 = now();
    void *h;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    printf("hipHostMalloc %.2f GiB: %.1f ms\n", bytes / 1073741824.0, (now() - t0) * 1e3);
    if (b == 0) d2h(h, "hipHostMalloc");
    t0 = now();
    CK(hipHostFree(h));
    printf("   hipHostFree: %.1f ms\n", (now() - t0) * 1e3);
  }
  for (int huge = 0; huge < 2; huge++)
    for (int b = 0; b < blocks; b++) {
      double t0 = now();
      char *p = (char *)mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      char *a = (char *)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
      if (huge) madvise(a, bytes, MADV_HUGEPAGE);
      double t1 = now();
      touch(a, bytes, threads);
      double t2 = now();
      CK(hipHostRegister(a, bytes, hipHostRegisterDefault));
      double t3 = now();
      printf("mmap%s %.2f GiB: map %.1f ms, touch (%d threads) %.1f ms, hipHostRegister %.1f ms, total %.1f ms\n",
             huge ? " + MADV_HUGEPAGE" : "", bytes / 1073741824.0, (t1 - t0) * 1e3, threads, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
      if (b == 0) d2h(a, huge ? "registered huge" : "registered 4K");
      t0 = now();
      CK(hipHostUnregister(a));
      munmap(p, bytes + (2 << 20));
      printf("   unregister + munmap: %.1f ms\n", (now() - t0) * 1e3);
    }
  // pageable target, fresh and touched
  for (int pre = 0; pre < 2; pre++) {
    char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (pre) touch(p, bytes, threads);
    double t0 = now();
    CK(hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost));
    printf("pageable (%s) d2h: %.1f GB/s\n", pre ? "touched" : "fresh", bytes / (now() - t0) / 1e9);
    munmap(p, bytes);
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char buf[128] = {0};
  if (f) { if (fgets(buf, 127, f)) printf("THP: %s", buf); fclose(f); }
  return 0;
}

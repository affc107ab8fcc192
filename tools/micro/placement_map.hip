// placement_map -- what defines a "write-conflict class" of device memory on MI355X.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/placement_map.hip -o tools/micro/placement_map
//   tools/micro/placement_map [n_handles=160] [handle_MiB=1024]
//
// The product (descriptools_amd/placement.py) only measures that two rasters written concurrently at equal offsets run
// at one of two speeds and that "slow" is an equivalence relation.  This experiment separates VIRTUAL from PHYSICAL:
// physical memory is created with hipMemCreate (one handle per block) and mapped wherever we like with hipMemMap.
//   (1) label every handle's class in creation order (pair-write timing against one representative per class);
//   (2) unmap everything, map the SAME handles at new virtual addresses in a permuted order, label again:
//       class follows the handle (physical) or the address (virtual)?
//   (3) the same at 64 MiB granularity over a window that contains a class change: how sharp is the boundary and at
//       what multiple of the handle size does it fall?
//   (4) a raster assembled from chunks of two classes (alternating 64 MiB handles): is the conflict per-chunk?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// n_writes streams written in lock-step at equal offsets, 16 bytes per lane, non-temporal, grid-stride
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_write4(vf4 *w0, vf4 *w1, long long n4, int n_writes) {
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    __builtin_nontemporal_store(v, &w0[i]);
    if (n_writes > 1) __builtin_nontemporal_store(v, &w1[i]);
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  // mean ms of `reps` launches writing n_writes streams of `bytes` bytes
  float run(void *p0, void *p1, size_t bytes, int n_writes, int reps = 4) {
    const long long n4 = (long long)(bytes / 16);
    hipLaunchKernelGGL(k_write4, dim3(8192), dim3(256), 0, 0, (vf4 *)p0, (vf4 *)p1, n4, n_writes);
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL(k_write4, dim3(8192), dim3(256), 0, 0, (vf4 *)p0, (vf4 *)p1, n4, n_writes);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
  }
};

struct Labeller {
  Timer t;
  size_t bytes;
  float single = 0.f;
  std::vector<void *> reps;
  std::vector<float> ratios;
  explicit Labeller(size_t b) : bytes(b) {}
  int label(void *p) {
    if (single == 0.f) single = std::min(t.run(p, p, bytes, 1), t.run(p, p, bytes, 1));
    for (size_t k = 0; k < reps.size(); k++) {
      float r = t.run(reps[k], p, bytes, 2) / single;
      if (r > 1.86f && r < 2.02f) r = t.run(reps[k], p, bytes, 2, 8) / single;  // near the threshold: measure again
      ratios.push_back(r);
      if (r > 1.94f) return (int)k;
    }
    reps.push_back(p);
    return (int)reps.size() - 1;
  }
};

// (6) ONE large allocation, slices at chosen offsets: conflict ratio of the pair (slice 0, slice at +delta) for every
// delta -- which bits of the address difference decide?  kind 0: hipMalloc, 1: one hipMemCreate handle
static void delta_scan(int kind, size_t total, size_t slice, size_t step) {
  void *base = nullptr;
  hipMemGenericAllocationHandle_t hd;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (kind == 0) {
    if (hipMalloc(&base, total) != hipSuccess) { (void)hipGetLastError(); printf("(6) hipMalloc(%zu GiB) failed\n", total >> 30); return; }
  } else {
    if (hipMemCreate(&hd, total, &prop, 0) != hipSuccess) { (void)hipGetLastError(); printf("(6) hipMemCreate(%zu GiB) failed\n", total >> 30); return; }
    CK(hipMemAddressReserve(&base, total, 0, nullptr, 0));
    CK(hipMemMap(base, total, 0, hd, 0));
    CK(hipMemSetAccess(base, total, &acc, 1));
  }
  Timer t;
  const float single = std::min(t.run(base, base, slice, 1, 6), t.run(base, base, slice, 1, 6));
  printf("(6) %s of %zu GiB at %p, slices of %zu MiB, pair (0, +delta) / single (%.4f ms), delta in steps of %zu MiB:\n   ",
         kind ? "one hipMemCreate handle" : "one hipMalloc", total >> 30, base, slice >> 20, single, step >> 20);
  std::string bits;
  for (size_t d = slice > step ? slice : step; d + slice <= total; d += step) {
    const float r = t.run(base, (char *)base + d, slice, 2, 4) / single;
    printf(" %.2f", r);
    bits += r > 1.94f ? 'X' : (r > 1.7f ? 'x' : '.');
  }
  printf("\n    conflict map by delta (X > 1.94, x > 1.7): %s\n", bits.c_str());
  // the same relative to a slice in the middle (is it a function of the DIFFERENCE or of each address?)
  const size_t mid = (total / 2) / step * step;
  std::string bits2;
  for (size_t d = 0; d + slice <= total; d += step) {
    if (d + slice > mid && d < mid + slice) { bits2 += '-'; continue; }
    const float r = t.run((char *)base + mid, (char *)base + d, slice, 2, 4) / single;
    bits2 += r > 1.94f ? 'X' : (r > 1.7f ? 'x' : '.');
  }
  printf("    conflict map against the slice at +%zu MiB, by absolute offset: %s\n", mid >> 20, bits2.c_str());
  CK(hipDeviceSynchronize());
  if (kind == 0) CK(hipFree(base));
  else { CK(hipMemUnmap(base, total)); CK(hipMemAddressFree(base, total)); CK(hipMemRelease(hd)); }
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "delta") {
    CK(hipSetDevice(0));
    const size_t total = (size_t)(argc > 2 ? atoi(argv[2]) : 48) << 30;
    delta_scan(0, total, 1ull << 30, 256ull << 20);
    delta_scan(1, total, 1ull << 30, 256ull << 20);
    delta_scan(0, 8ull << 30, 256ull << 20, 32ull << 20);  // fine: which low bits matter
    return 0;
  }
  const int n = argc > 1 ? atoi(argv[1]) : 160;
  const size_t hbytes = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t freeb = 0, totalb = 0;
  CK(hipMemGetInfo(&freeb, &totalb));
  printf("granularity %zu KiB, free %.1f GiB of %.1f GiB, %d handles of %zu MiB\n", gran >> 10, freeb / 1073741824.0,
         totalb / 1073741824.0, n, hbytes >> 20);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;

  // ---- (1) handles in creation order ----
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  int made = 0;
  for (int i = 0; i < n; i++) {
    if (hipMemCreate(&h[i], hbytes, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    made++;
  }
  printf("created %d handles\n", made);
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, hbytes * made, 0, nullptr, 0));
  for (int i = 0; i < made; i++) {
    CK(hipMemMap((char *)va + hbytes * i, hbytes, 0, h[i], 0));
  }
  CK(hipMemSetAccess(va, hbytes * made, &acc, 1));
  Labeller L1(hbytes);
  std::vector<int> cls(made);
  std::string seq;
  for (int i = 0; i < made; i++) {
    cls[i] = L1.label((char *)va + hbytes * i);
    seq += (char)('A' + cls[i]);
  }
  printf("(1) classes by creation order (single stream %.4f ms):\n    %s\n", L1.single, seq.c_str());
  printf("    run lengths:");
  for (int i = 0, j; i < made; i = j) {
    for (j = i; j < made && cls[j] == cls[i]; j++) {}
    printf(" %c%d", 'A' + cls[i], j - i);
  }
  printf("\n    pair ratios min %.3f max %.3f; none in (1.86, 2.02): %s\n",
         *std::min_element(L1.ratios.begin(), L1.ratios.end()), *std::max_element(L1.ratios.begin(), L1.ratios.end()),
         std::none_of(L1.ratios.begin(), L1.ratios.end(), [](float r) { return r > 1.86f && r < 2.02f; }) ? "yes" : "no");

  // ---- (2) the same handles at new virtual addresses, permuted ----
  CK(hipDeviceSynchronize());
  CK(hipMemUnmap(va, hbytes * made));
  CK(hipMemAddressFree(va, hbytes * made));
  std::vector<int> perm(made);
  for (int i = 0; i < made; i++) perm[i] = (int)(((long long)i * 37 + 11) % made);  // 37 coprime to made?  fixed below
  {  // a permutation for any `made`: reverse + interleave halves
    std::vector<int> p2;
    for (int i = 0; i < (made + 1) / 2; i++) { p2.push_back(made - 1 - i); if (i != made - 1 - i) p2.push_back(i); }
    perm = p2;
  }
  void *vb = nullptr;
  CK(hipMemAddressReserve(&vb, hbytes * made, 0, nullptr, 0));
  for (int i = 0; i < made; i++) CK(hipMemMap((char *)vb + hbytes * i, hbytes, 0, h[perm[i]], 0));
  CK(hipMemSetAccess(vb, hbytes * made, &acc, 1));
  // label with the OLD representatives' handles: find where they went
  Labeller L2(hbytes);
  std::vector<int> cls2(made);
  // label in the order of the original creation so that class ids line up (first of each class = same handle)
  std::vector<int> where(made);
  for (int i = 0; i < made; i++) where[perm[i]] = i;
  int follow_phys = 0;
  for (int i = 0; i < made; i++) {
    cls2[i] = L2.label((char *)vb + hbytes * where[i]);
    follow_phys += cls2[i] == cls[i];
  }
  std::string seq2;
  for (int i = 0; i < made; i++) seq2 += (char)('A' + cls2[i]);
  printf("(2) same handles, remapped in permuted order at %p (was %p), labelled by HANDLE in creation order:\n    %s\n"
         "    %d of %d handles keep their class -> the class is a property of the %s\n",
         vb, va, seq2.c_str(), follow_phys, made, follow_phys == made ? "PHYSICAL block" : "mapping (or unstable)");
  // and read by VIRTUAL order, to show the virtual sequence is scrambled accordingly
  std::string seqv;
  for (int i = 0; i < made; i++) seqv += (char)('A' + cls2[perm[i]]);
  printf("    by virtual address order: %s\n", seqv.c_str());

  // ---- (4) a raster assembled from chunks of two classes ----
  CK(hipDeviceSynchronize());
  int ia = -1, ib = -1, ia2 = -1;
  for (int i = 0; i < made; i++) {
    if (cls[i] == 0 && ia < 0) ia = i;
    else if (cls[i] == 0 && ia2 < 0) ia2 = i;
    if (cls[i] == 1 && ib < 0) ib = i;
  }
  if (ia >= 0 && ib >= 0 && ia2 >= 0) {
    Timer t;
    float same = t.run((char *)vb + hbytes * where[ia], (char *)vb + hbytes * where[ia2], hbytes, 2, 6);
    float diff = t.run((char *)vb + hbytes * where[ia], (char *)vb + hbytes * where[ib], hbytes, 2, 6);
    printf("(4) pair A+A %.4f ms, pair A+B %.4f ms (single %.4f)\n", same, diff, L1.single);
  }
  CK(hipMemUnmap(vb, hbytes * made));
  CK(hipMemAddressFree(vb, hbytes * made));

  // ---- (3) finer handles: release everything, create 64 MiB handles, label, look at the run lengths ----
  for (int i = 0; i < made; i++) CK(hipMemRelease(h[i]));
  const size_t fb = 64u << 20;
  const int nf = (int)std::min<size_t>((size_t)made * (hbytes / fb), 1536);  // <= 96 GiB of 64 MiB handles
  std::vector<hipMemGenericAllocationHandle_t> f(nf);
  int fm = 0;
  for (int i = 0; i < nf; i++) {
    if (hipMemCreate(&f[i], fb, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    fm++;
  }
  void *vc = nullptr;
  CK(hipMemAddressReserve(&vc, fb * fm, 0, nullptr, 0));
  for (int i = 0; i < fm; i++) CK(hipMemMap((char *)vc + fb * i, fb, 0, f[i], 0));
  CK(hipMemSetAccess(vc, fb * fm, &acc, 1));
  Labeller L3(fb);
  std::vector<int> c3(fm);
  for (int i = 0; i < fm; i++) c3[i] = L3.label((char *)vc + fb * i);
  printf("(3) %d handles of 64 MiB in creation order, run lengths (x 64 MiB):", fm);
  for (int i = 0, j; i < fm; i = j) {
    for (j = i; j < fm && c3[j] == c3[i]; j++) {}
    printf(" %c%d", 'A' + c3[i], j - i);
  }
  printf("\n    classes found %zu; pair ratios min %.3f max %.3f\n", L3.reps.size(),
         L3.ratios.empty() ? 0.f : *std::min_element(L3.ratios.begin(), L3.ratios.end()),
         L3.ratios.empty() ? 0.f : *std::max_element(L3.ratios.begin(), L3.ratios.end()));
  // (5) a 1 GiB raster assembled from 16 chunks alternating between two classes vs one of a single class
  {
    std::vector<int> a_idx, b_idx;
    for (int i = 0; i < fm; i++) (c3[i] == 0 ? a_idx : b_idx).push_back(i);
    if (a_idx.size() >= 40 && b_idx.size() >= 8) {
      CK(hipDeviceSynchronize());
      CK(hipMemUnmap(vc, fb * fm));
      CK(hipMemAddressFree(vc, fb * fm));
      const size_t rb = 16 * fb;
      void *vd = nullptr;
      CK(hipMemAddressReserve(&vd, rb * 3, 0, nullptr, 0));
      // raster 0: 16 A chunks; raster 1: 16 more A chunks; raster 2: alternating A / B chunks
      for (int k = 0; k < 16; k++) CK(hipMemMap((char *)vd + fb * k, fb, 0, f[a_idx[k]], 0));
      for (int k = 0; k < 16; k++) CK(hipMemMap((char *)vd + rb + fb * k, fb, 0, f[a_idx[16 + k]], 0));
      for (int k = 0; k < 16; k++)
        CK(hipMemMap((char *)vd + 2 * rb + fb * k, fb, 0, (k & 1) ? f[b_idx[k / 2]] : f[a_idx[32 + k / 2]], 0));
      CK(hipMemSetAccess(vd, rb * 3, &acc, 1));
      Timer t;
      float s = t.run(vd, vd, rb, 1, 6);
      float aa = t.run(vd, (char *)vd + rb, rb, 2, 6);
      float am = t.run(vd, (char *)vd + 2 * rb, rb, 2, 6);
      printf("(5) 1 GiB rasters from 64 MiB chunks: single %.4f ms; A-chunks + A-chunks %.4f (x%.3f); A-chunks + "
             "alternating A/B chunks %.4f (x%.3f)\n", s, aa, aa / s, am, am / s);
      CK(hipMemUnmap(vd, rb * 3));
      CK(hipMemAddressFree(vd, rb * 3));
    } else {
      printf("(5) skipped: classes at 64 MiB: %zu A, %zu B\n", a_idx.size(), b_idx.size());
      CK(hipMemUnmap(vc, fb * fm));
      CK(hipMemAddressFree(vc, fb * fm));
    }
  }
  for (int i = 0; i < fm; i++) CK(hipMemRelease(f[i]));
  return 0;
}

// placement_study.hip -- why does the same 1 GB write stream run at 5.7 TB/s into one allocation and at 6.9 into the
// next (profiles/r01_nich1_variants.txt, fourth table)?  Stand-alone (no library, no torch):
//   A. NBUF hipMalloc'ed buffers of the C2 score matrix's size; per buffer: VA, hipMemsetAsync rate, the rate of the
//      k_score_nich1 store pattern (4-row blocks, 2 visits per wave, non-temporal 1 KiB rows) and of a flat
//      grid-stride fill;
//   B. for the slowest and the fastest buffer of A: the same pattern on every 64 MiB piece on its own (is the
//      slowness spread evenly or does it sit in parts of the allocation?);
//   C. the same size through the other allocators a library could offer: a slice of one large hipMalloc at 2 MiB /
//      1 GiB aligned offsets, hipMallocAsync (stream-ordered pool), and the virtual-memory API (hipMemCreate +
//      hipMemMap with the recommended granularity);
//   D. time series: 40 x 50 launches into one buffer (does the rate drift, i.e. is it a clock / power state?).
// Build: hipcc -O3 --offload-arch=gfx950 placement_study.hip -o placement_study
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("err %s at line %d\n", hipGetErrorString(e), __LINE__);           \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// the store pattern of k_score_nich1 at K = 256: a wave owns a slot and visits blocks of Q rows nslots * Q rows apart
template <int Q>
__global__ __launch_bounds__(256) void k_fill_slots(f4 *out, size_t nrows, size_t nslots, float v) {
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nslots) return;
  for (size_t rb = slot * Q; rb < nrows; rb += nslots * Q)
#pragma unroll
    for (int r = 0; r < Q; r++)
      if (rb + r < nrows) {
        f4 x = {v + r, v, v, v};
        __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane);
      }
}
__global__ __launch_bounds__(256) void k_fill_flat(f4 *out, size_t n16, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    f4 x = {v, v, v, v};
    __builtin_nontemporal_store(x, out + i);
  }
}

template <typename F>
static float timeit(F f, int reps, int warm = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < warm; i++) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms / reps;
}

static double rate_slots(f4 *buf, size_t nrows, int visits, int reps = 20) {
  const size_t nslots = (nrows / 4 + visits - 1) / visits;
  const float ms = timeit([&] { k_fill_slots<4><<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, nslots, 1.f); }, reps);
  return nrows * 1024.0 / ms / 1e9;
}
static double rate_flat(f4 *buf, size_t bytes, int reps = 20) {
  const float ms = timeit([&] { k_fill_flat<<<256 * 8, 256>>>(buf, bytes / 16, 1.f); }, reps);
  return bytes / ms / 1e9;
}
static double rate_memset(void *buf, size_t bytes, int reps = 20) {
  const float ms = timeit([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }, reps);
  return bytes / ms / 1e9;
}

static void report(const char *tag, void *p, size_t nrows) {
  const size_t bytes = nrows * 1024;
  const unsigned long long va = (unsigned long long)p;
  printf("%-28s va %#014llx  mod2M %7llu KiB  mod1G %5llu MiB | memset %5.2f  flat %5.2f  slots v1 %5.2f  v2 %5.2f  v4 %5.2f TB/s\n",
         tag, va, (va & ((2ull << 20) - 1)) >> 10, (va & ((1ull << 30) - 1)) >> 20, rate_memset(p, bytes), rate_flat((f4 *)p, bytes),
         rate_slots((f4 *)p, nrows, 1), rate_slots((f4 *)p, nrows, 2), rate_slots((f4 *)p, nrows, 4));
}

int main(int argc, char **argv) {
  const int NBUF = argc > 1 ? atoi(argv[1]) : 16;
  const size_t nrows = 1000000, bytes = nrows * 1024;
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("device memory: %.1f GB free of %.1f GB\n", fr / 1e9, tot / 1e9);

  // ---- A ----
  printf("== A: %d hipMalloc buffers of %zu bytes\n", NBUF, bytes);
  std::vector<f4 *> buf(NBUF);
  std::vector<double> r2(NBUF);
  for (int i = 0; i < NBUF; i++) CK(hipMalloc(&buf[i], bytes));
  for (int i = 0; i < NBUF; i++) {
    char tag[64];
    snprintf(tag, sizeof tag, "hipMalloc #%d", i);
    report(tag, buf[i], nrows);
    r2[i] = rate_slots(buf[i], nrows, 2);
  }
  const int slow = (int)(std::min_element(r2.begin(), r2.end()) - r2.begin());
  const int fast = (int)(std::max_element(r2.begin(), r2.end()) - r2.begin());
  printf("slowest #%d %.2f TB/s, fastest #%d %.2f TB/s (slots, 2 visits)\n", slow, r2[slow], fast, r2[fast]);

  // ---- B ----
  for (int which : {slow, fast}) {
    printf("== B: buffer #%d in 64 MiB pieces (slots v2 | flat), TB/s\n", which);
    const size_t piece_rows = 65536;   // 64 MiB
    for (size_t r0 = 0; r0 + piece_rows <= nrows; r0 += piece_rows) {
      f4 *p = buf[which] + r0 * 64;
      printf("  piece %2zu: %5.2f | %5.2f\n", r0 / piece_rows, rate_slots(p, piece_rows, 2, 40), rate_flat(p, piece_rows * 1024, 40));
    }
  }

  // ---- D ----
  {
    printf("== D: time series on buffer #%d (slots v2), 40 x 50 launches, TB/s:\n  ", slow);
    for (int i = 0; i < 40; i++) printf("%.2f ", rate_slots(buf[slow], nrows, 2, 50));
    printf("\n");
    printf("   and on buffer #%d:\n  ", fast);
    for (int i = 0; i < 40; i++) printf("%.2f ", rate_slots(buf[fast], nrows, 2, 50));
    printf("\n");
  }
  for (int i = 0; i < NBUF; i++) CK(hipFree(buf[i]));

  // ---- C ----
  printf("== C: other allocators, same size\n");
  {
    char *big = nullptr;
    const size_t bigbytes = 8ull << 30;
    CK(hipMalloc(&big, bigbytes));
    const unsigned long long va = (unsigned long long)big;
    const size_t to1g = (size_t)(((va + (1ull << 30) - 1) & ~((1ull << 30) - 1)) - va);
    report("8 GiB slab + 0", big, nrows);
    report("8 GiB slab, 1 GiB aligned", big + to1g, nrows);
    report("8 GiB slab, 1 GiB + 2 MiB", big + to1g + (2u << 20), nrows);
    report("8 GiB slab, 1 GiB + 1 KiB", big + to1g + 1024, nrows);
    report("8 GiB slab, +3 GiB", big + to1g + (3ull << 30), nrows);
    report("8 GiB slab, +5 GiB", big + to1g + (5ull << 30), nrows);
    CK(hipFree(big));
  }
  {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 3; i++) {
      void *p = nullptr;
      if (hipMallocAsync(&p, bytes, s) != hipSuccess) { printf("hipMallocAsync unavailable\n"); (void)hipGetLastError(); break; }
      CK(hipStreamSynchronize(s));
      char tag[64];
      snprintf(tag, sizeof tag, "hipMallocAsync #%d", i);
      report(tag, p, nrows);
      // (not freed: the next one must land elsewhere)
    }
    CK(hipStreamDestroy(s));
  }
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    if (e == hipSuccess) e = hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) {
      printf("virtual memory API unavailable: %s\n", hipGetErrorString(e));
      (void)hipGetLastError();
    } else {
      printf("hipMem granularity: minimum %zu, recommended %zu\n", gmin, grec);
      for (int i = 0; i < 3; i++) {
        const size_t g = grec ? grec : (2u << 20);
        const size_t sz = (bytes + g - 1) / g * g;
        hipMemGenericAllocationHandle_t h;
        void *va = nullptr;
        if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); (void)hipGetLastError(); break; }
        if (hipMemAddressReserve(&va, sz, 1ull << 30, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); (void)hipGetLastError(); break; }
        CK(hipMemMap(va, sz, 0, h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, sz, &acc, 1));
        char tag[64];
        snprintf(tag, sizeof tag, "hipMemCreate+Map #%d", i);
        report(tag, va, nrows);
      }
    }
  }
  return 0;
}

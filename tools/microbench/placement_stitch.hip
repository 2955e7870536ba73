// placement_stitch.hip -- can a user-mode allocation be MADE fast?  The 1 GB score matrix is mapped from separately
// created physical chunks (hipMemCreate) of 2 / 32 / 256 MiB, in allocation order or shuffled, into one VA range
// (hipMemAddressReserve + hipMemMap), and the C2 store pattern is timed on each.  If the fast regime of
// profiles/r02_placement_study.txt is "pages spread over the channels", a shuffled map of small chunks should find it.
// Build: hipcc -O3 --offload-arch=gfx950 placement_stitch.hip -o placement_stitch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                        \
  do {                                                               \
    hipError_t e = (x);                                              \
    if (e != hipSuccess) {                                           \
      printf("err %s at line %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

__global__ __launch_bounds__(256) void k_fill_slots(f4 *out, size_t nrows, size_t nslots, float v) {
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nslots) return;
  for (size_t rb = slot * 4; rb < nrows; rb += nslots * 4)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (rb + r < nrows) {
        f4 x = {v + r, v, v, v};
        __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane);
      }
}
static double rate(f4 *buf, size_t nrows, int visits) {
  const size_t nslots = (nrows / 4 + visits - 1) / visits;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) k_fill_slots<<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, nslots, 1.f);
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; i++) k_fill_slots<<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, nslots, 1.f);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return nrows * 1024.0 / (ms / 20) / 1e9;
}

struct Mapped {
  void *va = nullptr;
  size_t size = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
static Mapped map_chunks(size_t bytes, size_t chunk, bool shuffle, unsigned seed) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  Mapped m;
  const size_t n = (bytes + chunk - 1) / chunk;
  m.size = n * chunk;
  CK(hipMemAddressReserve(&m.va, m.size, 2u << 20, nullptr, 0));
  m.handles.resize(n);
  for (size_t i = 0; i < n; i++) CK(hipMemCreate(&m.handles[i], chunk, &prop, 0));
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  if (shuffle) {
    std::mt19937 g(seed);
    std::shuffle(order.begin(), order.end(), g);
  }
  for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)m.va + i * chunk, chunk, 0, m.handles[order[i]], 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(m.va, m.size, &acc, 1));
  return m;
}
static void unmap(Mapped &m) {
  CK(hipDeviceSynchronize());
  CK(hipMemUnmap(m.va, m.size));
  for (auto h : m.handles) CK(hipMemRelease(h));
  CK(hipMemAddressFree(m.va, m.size));
}

int main() {
  const size_t nrows = 1000000, bytes = nrows * 1024;
  for (int rep = 0; rep < 2; rep++) {
    f4 *plain = nullptr;
    CK(hipMalloc(&plain, bytes));
    printf("hipMalloc                        : v1 %.2f  v2 %.2f  v4 %.2f TB/s\n", rate(plain, nrows, 1), rate(plain, nrows, 2), rate(plain, nrows, 4));
    for (size_t chunk : {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20})
      for (int shuffle = 0; shuffle < 2; shuffle++) {
        Mapped m = map_chunks(bytes, chunk, shuffle != 0, 7u + rep);
        printf("chunks of %3zu MiB, %-10s : v1 %.2f  v2 %.2f  v4 %.2f TB/s\n", chunk >> 20, shuffle ? "shuffled" : "in order", rate((f4 *)m.va, nrows, 1),
               rate((f4 *)m.va, nrows, 2), rate((f4 *)m.va, nrows, 4));
        unmap(m);
      }
    // (the plain buffer stays allocated for the second round, so that the chunks come from elsewhere)
  }
  return 0;
}

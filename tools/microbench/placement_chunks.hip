// placement_chunks.hip -- is the write-rate regime of a large buffer (profiles/r02_placement_study.txt: 5.6 vs 7.0 TB/s for
// the same store stream) a property of the individual 32 MiB physical chunks it is mapped from?  If it is, a library can
// build a well-placed buffer instead of hoping for one: over-allocate chunks, profile them, keep the good ones.
//   1. NC chunks (hipMemCreate, 32 MiB) mapped side by side into one VA range.
//   2. One fill launch over the whole range, C2 store pattern with one visit per wave (a workgroup = 16 consecutive
//      1 KiB rows), every workgroup stamping its end time (s_memrealtime, 100 MHz): the write front moves through the
//      range in address order, so the time the front needs for a chunk = 32 MiB / the chunk's rate.  Repeated, rank
//      correlation between repeats printed.
//   3. The chunks are re-mapped into 1 GB buffers made of the fastest / slowest / first / random 32 and the real pattern
//      (two visits, 20 launches, HIP events) is timed on each, beside a hipMalloc'ed buffer.
// Build: hipcc -O3 --offload-arch=gfx950 placement_chunks.hip -o placement_chunks
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
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

static const size_t kChunk = 32u << 20;

__global__ __launch_bounds__(256) void k_fill_timed(f4 *out, size_t nrows, unsigned long long *t_end) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t rb = (size_t)blockIdx.x * 16 + wave * 4;
  f4 x = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (rb + r < nrows) __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane);
  __builtin_amdgcn_s_waitcnt(0);                        // (stores issued; vmcnt covers them until written back to L2)
  __syncthreads();
  if (threadIdx.x == 0) t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
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
  for (int i = 0; i < 5; i++) k_fill_slots<<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, nslots, 1.f);
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; i++) k_fill_slots<<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, nslots, 1.f);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return nrows * 1024.0 / (ms / 20) / 1e9;
}

static hipMemAllocationProp g_prop;
static void *map_list(const std::vector<hipMemGenericAllocationHandle_t> &hs) {
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, hs.size() * kChunk, 2u << 20, nullptr, 0));
  for (size_t i = 0; i < hs.size(); i++) CK(hipMemMap((char *)va + i * kChunk, kChunk, 0, hs[i], 0));
  hipMemAccessDesc acc = {};
  acc.location = g_prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, hs.size() * kChunk, &acc, 1));
  return va;
}
static void unmap_list(void *va, size_t n) {
  CK(hipDeviceSynchronize());
  CK(hipMemUnmap(va, n * kChunk));
  CK(hipMemAddressFree(va, n * kChunk));
}

// per-chunk seconds from one timed launch: the front's arrival at the end of chunk c minus its arrival at the end of c - 1
static std::vector<double> profile(f4 *base, size_t nc, unsigned long long *t_dev, std::vector<unsigned long long> &t_host) {
  const size_t nrows = nc * kChunk / 1024, nwg = nrows / 16, per = kChunk / 1024 / 16;
  k_fill_timed<<<(unsigned)nwg, 256>>>(base, nrows, t_dev);
  CK(hipMemcpy(t_host.data(), t_dev, nwg * 8, hipMemcpyDeviceToHost));
  std::vector<double> end(nc);
  for (size_t c = 0; c < nc; c++) {
    std::vector<unsigned long long> v(t_host.begin() + c * per, t_host.begin() + (c + 1) * per);
    std::nth_element(v.begin(), v.begin() + per * 9 / 10, v.end());          // 90th percentile: robust "front has passed"
    end[c] = (double)v[per * 9 / 10];
  }
  std::vector<double> sec(nc, 0.0);
  for (size_t c = 1; c < nc; c++) sec[c] = (end[c] - end[c - 1]) * 1e-8;      // 100 MHz
  sec[0] = sec[1];
  return sec;
}
static double spearman(const std::vector<double> &a, const std::vector<double> &b) {
  const size_t n = a.size();
  auto ranks = [&](const std::vector<double> &x) {
    std::vector<size_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::sort(o.begin(), o.end(), [&](size_t i, size_t j) { return x[i] < x[j]; });
    std::vector<double> r(n);
    for (size_t i = 0; i < n; i++) r[o[i]] = (double)i;
    return r;
  };
  auto ra = ranks(a), rb = ranks(b);
  double ma = (n - 1) / 2.0, num = 0, da = 0, db = 0;
  for (size_t i = 0; i < n; i++) {
    num += (ra[i] - ma) * (rb[i] - ma);
    da += (ra[i] - ma) * (ra[i] - ma);
    db += (rb[i] - ma) * (rb[i] - ma);
  }
  return num / std::sqrt(da * db);
}

int main(int argc, char **argv) {
  const size_t nc = argc > 1 ? (size_t)atoi(argv[1]) : 128;
  const size_t take = 32;                                 // 1 GB = 32 chunks (rows of 1 KiB: 1,048,576)
  g_prop = {};
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  f4 *plain = nullptr;
  CK(hipMalloc(&plain, take * kChunk));
  std::vector<hipMemGenericAllocationHandle_t> hs(nc);
  for (size_t i = 0; i < nc; i++) CK(hipMemCreate(&hs[i], kChunk, &g_prop, 0));
  void *pool = map_list(hs);
  const size_t nwg = nc * kChunk / 1024 / 16;
  unsigned long long *t_dev = nullptr;
  CK(hipMalloc(&t_dev, nwg * 8));
  std::vector<unsigned long long> t_host(nwg);
  for (int w = 0; w < 3; w++) profile((f4 *)pool, nc, t_dev, t_host);          // warm
  const int reps = 6;
  std::vector<std::vector<double>> runs;
  for (int r = 0; r < reps; r++) runs.push_back(profile((f4 *)pool, nc, t_dev, t_host));
  std::vector<double> avg(nc, 0.0);
  for (auto &v : runs)
    for (size_t c = 0; c < nc; c++) avg[c] += v[c] / reps;
  printf("# per-chunk rate (TB/s), %zu chunks of 32 MiB, mean of %d timed launches; rank correlation run0-run1 %.3f, run0-run5 %.3f\n", nc, reps,
         spearman(runs[0], runs[1]), spearman(runs[0], runs[5]));
  for (size_t c = 0; c < nc; c++) printf("%s%.2f", c % 16 ? " " : "\n  ", kChunk / avg[c] / 1e12);
  printf("\n");
  // the same chunks mapped in REVERSE order: does the rate follow the chunk or the position?
  {
    unmap_list(pool, nc);
    std::vector<hipMemGenericAllocationHandle_t> rev(hs.rbegin(), hs.rend());
    void *p2 = map_list(rev);
    for (int w = 0; w < 2; w++) profile((f4 *)p2, nc, t_dev, t_host);
    std::vector<double> r2(nc, 0.0);
    for (int r = 0; r < reps; r++) {
      auto v = profile((f4 *)p2, nc, t_dev, t_host);
      for (size_t c = 0; c < nc; c++) r2[nc - 1 - c] += v[c] / reps;            // back to handle order
    }
    printf("# reversed mapping: rank correlation with the in-order profile, by chunk %.3f, by position %.3f\n", spearman(avg, r2),
           spearman(avg, std::vector<double>(r2.rbegin(), r2.rend())));
    unmap_list(p2, nc);
  }
  std::vector<size_t> order(nc);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](size_t i, size_t j) { return avg[i] < avg[j]; });    // fastest first
  auto pick = [&](const std::vector<size_t> &idx) {
    std::vector<hipMemGenericAllocationHandle_t> v;
    for (size_t i : idx) v.push_back(hs[i]);
    return v;
  };
  std::vector<size_t> best(order.begin(), order.begin() + take), worst(order.end() - take, order.end()), first(take), rnd(order);
  std::iota(first.begin(), first.end(), 0);
  std::mt19937 g(3);
  std::shuffle(rnd.begin(), rnd.end(), g);
  rnd.resize(take);
  std::sort(best.begin(), best.end());                    // (address order inside the buffer = allocation order)
  std::sort(worst.begin(), worst.end());
  const size_t nrows = 1000000;
  printf("# 1 GB buffers, C2 store pattern (two visits), TB/s, three measurements each\n");
  printf("hipMalloc          : %.2f %.2f %.2f\n", rate(plain, nrows, 2), rate(plain, nrows, 2), rate(plain, nrows, 2));
  struct { const char *name; std::vector<size_t> *idx; } sets[] = {{"fastest 32 chunks ", &best}, {"slowest 32 chunks ", &worst},
                                                                  {"first 32 chunks   ", &first}, {"random 32 chunks  ", &rnd}};
  for (auto &s : sets) {
    auto v = pick(*s.idx);
    void *va = map_list(v);
    double mean_prof = 0;
    for (size_t i : *s.idx) mean_prof += kChunk / avg[i] / 1e12 / take;
    printf("%s : %.2f %.2f %.2f   (v1 %.2f; mean profiled chunk rate %.2f)\n", s.name, rate((f4 *)va, nrows, 2), rate((f4 *)va, nrows, 2),
           rate((f4 *)va, nrows, 2), rate((f4 *)va, nrows, 1), mean_prof);
    unmap_list(va, take);
  }
  return 0;
}

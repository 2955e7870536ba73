// placement_counters.hip -- second half of the placement study (placement_study.hip found the two regimes: most 1 GB
// allocations take the C2 store stream at ~5.6 TB/s, a few at ~7.0, stable over time, whatever the launch shape).
// This program finds one buffer of each kind, then
//   1. runs the same store kernel under two NAMES (k_fill_slow / k_fill_fast) so that one `rocprofv3 --pmc` pass of
//      this binary gives per-kernel counters for the two placements (TCC_EA0_WRREQ_STALL, TCC_TOO_MANY_EA_WRREQS_STALL,
//      TCP_UTCL1_*, ...);
//   2. measures address-translation reach directly: a single lane walks a random cycle over one 64-byte line per
//      2 MiB (or 64 KiB / 4 KiB) of the buffer -- the lines fit L2, so what is left is the translation path -- and
//      prints ns per dependent load for both buffers;
//   3. measures the read stream too (is the slow placement slow for loads as well?).
// Build: hipcc -O3 --offload-arch=gfx950 placement_counters.hip -o placement_counters
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

template <int TAG>
__device__ __forceinline__ void fill_body(f4 *out, size_t nrows, size_t nslots, float v) {
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nslots) return;
  for (size_t rb = slot * 4; rb < nrows; rb += nslots * 4)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (rb + r < nrows) {
        f4 x = {v + r + TAG, v, v, v};
        __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane);
      }
}
__global__ __launch_bounds__(256) void k_fill_probe(f4 *out, size_t nrows, size_t nslots, float v) { fill_body<0>(out, nrows, nslots, v); }
__global__ __launch_bounds__(256) void k_fill_slow(f4 *out, size_t nrows, size_t nslots, float v) { fill_body<1>(out, nrows, nslots, v); }
__global__ __launch_bounds__(256) void k_fill_fast(f4 *out, size_t nrows, size_t nslots, float v) { fill_body<2>(out, nrows, nslots, v); }

template <int TAG>
__device__ __forceinline__ void read_body(const f4 *in, size_t n16, float *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  f4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) a += __builtin_nontemporal_load(in + i);
  if (a.x + a.y + a.z + a.w == 12345.678f + TAG) *sink = 1.f;
}
__global__ __launch_bounds__(256) void k_read_slow(const f4 *in, size_t n16, float *sink) { read_body<1>(in, n16, sink); }
__global__ __launch_bounds__(256) void k_read_fast(const f4 *in, size_t n16, float *sink) { read_body<2>(in, n16, sink); }

// one lane, dependent loads: next = buf[next * stride_words]
__global__ void k_chase(const unsigned *buf, size_t stride_words, unsigned start, int steps, unsigned *out, long long *cycles) {
  unsigned i = start;
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) i = __builtin_nontemporal_load(buf + (size_t)i * stride_words);
  const long long t1 = wall_clock64();
  *out = i;
  *cycles = t1 - t0;
}

static float time_fill(void (*k)(f4 *, size_t, size_t, float), f4 *buf, size_t nrows, int reps) {
  const size_t nslots = (nrows / 4 + 1) / 2;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, 0, buf, nrows, nslots, 1.f);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, 0, buf, nrows, nslots, 1.f);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

static void chase(const char *tag, void *buf, size_t bytes) {
  int wall_khz = 0;
  CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  unsigned *out;
  long long *cyc;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&cyc, 8));
  for (size_t stride : {(size_t)4096, (size_t)65536, (size_t)(2u << 20), (size_t)(32u << 20)}) {
    const size_t n = std::min<size_t>(bytes / stride, 8192);   // lines visited (<= 512 KiB of lines: L2-resident)
    if (n < 8) continue;
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 g(7);
    std::shuffle(perm.begin(), perm.end(), g);
    std::vector<unsigned> next(n);
    for (size_t i = 0; i < n; i++) next[perm[i]] = perm[(i + 1) % n];
    for (size_t i = 0; i < n; i++) CK(hipMemcpy((char *)buf + i * stride, &next[i], 4, hipMemcpyHostToDevice));
    const int steps = (int)std::min<size_t>(4 * n, 20000);
    long long best = 1ll << 62;
    for (int rep = 0; rep < 4; rep++) {                         // (first pass warms the lines into L2)
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, (const unsigned *)buf, stride / 4, 0u, steps, out, cyc);
      long long c;
      CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      if (rep) best = std::min(best, c);
    }
    printf("  %-6s chase stride %8zu B over %5zu lines: %7.1f ns per dependent load\n", tag, stride, n,
           (double)best / steps * 1e6 / wall_khz);
  }
  CK(hipFree(out));
  CK(hipFree(cyc));
}

int main(int argc, char **argv) {
  const int maxbuf = argc > 1 ? atoi(argv[1]) : 48;
  const size_t nrows = 1000000, bytes = nrows * 1024;
  std::vector<f4 *> bufs;
  std::vector<double> rate;
  int slow = -1, fast = -1;
  for (int i = 0; i < maxbuf; i++) {
    f4 *p;
    CK(hipMalloc(&p, bytes));
    bufs.push_back(p);
    rate.push_back(bytes / time_fill(k_fill_probe, p, nrows, 10) / 1e9);
    if (rate[i] < 5.9 && slow < 0) slow = i;
    if (rate[i] > 6.6 && fast < 0) fast = i;
    if (slow >= 0 && fast >= 0) break;
  }
  printf("probed %zu buffers, TB/s:", bufs.size());
  for (double r : rate) printf(" %.2f", r);
  printf("\n");
  if (slow < 0 || fast < 0) {
    // keep going with the two extremes so that the counters are still comparable
    slow = (int)(std::min_element(rate.begin(), rate.end()) - rate.begin());
    fast = (int)(std::max_element(rate.begin(), rate.end()) - rate.begin());
    printf("both regimes were not found; using the extremes\n");
  }
  printf("slow = #%d (%.2f TB/s) at %p, fast = #%d (%.2f TB/s) at %p\n", slow, rate[slow], (void *)bufs[slow], fast, rate[fast],
         (void *)bufs[fast]);
  // 1. named kernels for the counter pass
  const float ms_s = time_fill(k_fill_slow, bufs[slow], nrows, 20), ms_f = time_fill(k_fill_fast, bufs[fast], nrows, 20);
  printf("k_fill_slow %.1f us (%.2f TB/s)   k_fill_fast %.1f us (%.2f TB/s)\n", ms_s * 1e3, bytes / ms_s / 1e9, ms_f * 1e3,
         bytes / ms_f / 1e9);
  // 3. reads
  {
    float *sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int which = 0; which < 2; which++) {
      f4 *p = bufs[which ? fast : slow];
      auto go = [&] {
        if (which) hipLaunchKernelGGL(k_read_fast, dim3(256 * 8), dim3(256), 0, 0, (const f4 *)p, bytes / 16, sink);
        else hipLaunchKernelGGL(k_read_slow, dim3(256 * 8), dim3(256), 0, 0, (const f4 *)p, bytes / 16, sink);
      };
      go();
      go();
      CK(hipEventRecord(a));
      for (int i = 0; i < 20; i++) go();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("read stream of the %s buffer: %.2f TB/s\n", which ? "fast" : "slow", bytes / (ms / 20) / 1e9);
    }
  }
  // 2. translation reach
  chase("slow", bufs[slow], bytes);
  chase("fast", bufs[fast], bytes);
  return 0;
}

// Does the write rate depend on which 1 GB allocation is written?  Several hipMalloc'ed buffers in one process,
// the same fill kernel on each (1 KiB per wave instruction, one 4-row quad per wave, two fronts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
template <int ITERS>
__global__ __launch_bounds__(256) void k_fill(f4 *out, size_t nrows, float v) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nslots = (nrows / 4 + ITERS - 1) / ITERS;
  if (wave >= nslots) return;
  for (size_t rb = wave * 4; rb < nrows; rb += nslots * 4)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (rb + r < nrows) { f4 x = {v + r, v, v, v}; __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane); }
}
// Q rows per visit, ITERS visits per wave (stride = total waves * Q rows)
template <int Q, int ITERS>
__global__ __launch_bounds__(256) void k_fill_q(f4 *out, size_t nrows, float v) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nslots = (nrows / Q + ITERS - 1) / ITERS;
  if (wave >= nslots) return;
#pragma unroll
  for (int k = 0; k < ITERS; k++) {
    const size_t rb = (wave + (size_t)k * nslots) * Q;
#pragma unroll
    for (int r = 0; r < Q; r++)
      if (rb + r < nrows) { f4 x = {v + r, v, v, v}; __builtin_nontemporal_store(x, out + (rb + r) * 64 + lane); }
  }
}
template <int Q, int ITERS> void runq(f4 *buf, size_t nrows, size_t bytes);
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int Q, int ITERS> void runq(f4 *buf, size_t nrows, size_t bytes) {
  const size_t nslots = (nrows / Q + ITERS - 1) / ITERS;
  const float ms = timeit([&] { k_fill_q<Q, ITERS><<<(unsigned)((nslots + 3) / 4), 256>>>(buf, nrows, 1.f); }, 20);
  printf("  Q=%d ITERS=%-2d %.2f", Q, ITERS, bytes / ms / 1e9);
}
int main() {
  const size_t nrows = 1000000, bytes = nrows * 1024;
  f4 *buf[8];
  for (int i = 0; i < 8; i++) CK(hipMalloc(&buf[i], bytes));
  for (int pass = 0; pass < 2; pass++) {
    printf("fill TB/s:");
    runq<1, 2>(buf[0], nrows, bytes); runq<1, 3>(buf[0], nrows, bytes); runq<1, 4>(buf[0], nrows, bytes); runq<1, 5>(buf[0], nrows, bytes);
    runq<1, 6>(buf[0], nrows, bytes); runq<1, 8>(buf[0], nrows, bytes); runq<2, 2>(buf[0], nrows, bytes); runq<2, 4>(buf[0], nrows, bytes);
    printf("\n");
  }
  for (int pass = 0; pass < 1; pass++)
    for (int i = 0; i < 2; i++) {
      const float m1 = timeit([&] { k_fill<1><<<(nrows / 4 + 3) / 4, 256>>>(buf[i], nrows, 1.f); }, 20);
      const float m2 = timeit([&] { k_fill<2><<<(nrows / 8 + 3) / 4, 256>>>(buf[i], nrows, 1.f); }, 20);
      const float m4 = timeit([&] { k_fill<4><<<(nrows / 16 + 3) / 4, 256>>>(buf[i], nrows, 1.f); }, 20);
      const float ms = timeit([&] { CK(hipMemsetAsync(buf[i], 0, bytes, 0)); }, 20);
      printf("buffer %d @%p: 1 quad %.2f  2 quads %.2f  4 quads %.2f  memset %.2f TB/s\n", i, (void *)buf[i], bytes / m1 / 1e9,
             bytes / m2 / 1e9, bytes / m4 / 1e9, bytes / ms / 1e9);
    }
  return 0;
}

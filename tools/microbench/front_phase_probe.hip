// How does the rate of a two-front fill depend on (a) where the 1 GB region starts and (b) how far apart the fronts are?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
// wave w writes the 4 KiB quad w of front 0 and the quad w of front 1 (dist_quads further on)
__global__ __launch_bounds__(256) void k_two_fronts(f4 *out, size_t nquads_per_front, size_t dist_quads, float v) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nquads_per_front) return;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    f4 *p = out + (wave + k * dist_quads) * 256 + lane;
#pragma unroll
    for (int r = 0; r < 4; r++) { f4 x = {v + r, v, v, v}; __builtin_nontemporal_store(x, p + r * 64); }
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const size_t GB = 1ull << 30, MB = 1ull << 20;
  char *big; CK(hipMalloc(&big, 6 * GB));
  printf("base %p\n", (void *)big);
  const size_t nq = 125000;                       // quads per front: 2 x 125000 x 4 KiB = 1.024 GB written
  printf("(a) start offset of the region, fronts 512,000,000 B apart:\n");
  for (size_t off_mb : {0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 6, 10, 34, 66, 130, 258, 514, 1026, 100, 300, 700}) {
    f4 *p = (f4 *)(big + off_mb * MB);
    const float ms = timeit([&] { k_two_fronts<<<(unsigned)((nq + 3) / 4), 256>>>(p, nq, nq, 1.f); }, 10);
    printf("  +%4zu MiB: %.2f TB/s\n", off_mb, 2 * nq * 4096.0 / ms / 1e9);
  }
  printf("(b) distance between the fronts (region at +0):\n");
  for (size_t dist_mb : {488, 489, 490, 492, 496, 500, 504, 512, 520, 528, 544, 576, 640, 768, 1024, 1536, 2048}) {
    const size_t dq = dist_mb * MB / 4096;
    const float ms = timeit([&] { k_two_fronts<<<(unsigned)((nq + 3) / 4), 256>>>((f4 *)big, nq, dq, 1.f); }, 10);
    printf("  %4zu MiB apart: %.2f TB/s\n", dist_mb, 2 * nq * 4096.0 / ms / 1e9);
  }
  return 0;
}

// write-bandwidth ceiling probe for the C2 store pattern (1 KiB per wave instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

// each wave writes ROWS consecutive 1-KiB rows (same as k_score_nich1: lane <-> 16 B of a 1 KiB row)
template <int NT, int ROWS>
__global__ __launch_bounds__(256) void k_fill_rows(f4* out, size_t nrows, float v) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nw = (size_t)gridDim.x * 4;
  for (size_t r0 = wave * ROWS; r0 < nrows; r0 += nw * ROWS) {
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      if (r0 + r < nrows) {
        f4 x = {v + r, v, v, v};
        f4* p = out + (r0 + r) * 64 + lane;
        if (NT) __builtin_nontemporal_store(x, p); else *p = x;
      }
    }
  }
}
template <int NT>
__global__ __launch_bounds__(256) void k_fill_flat(f4* out, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) { f4 x = {v, v, v, v}; if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x; }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const size_t nrows = 1000000; const size_t bytes = nrows * 1024;
  f4* buf; CK(hipMalloc(&buf, bytes));
  auto rep = [&](const char* name, float ms) { printf("%-40s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9); };
  rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }, 20));
  for (int grid : {1024, 2048, 4096, 8192, 31250}) {
    char nm[64];
    snprintf(nm, 64, "rows32 nt grid=%d", grid);   rep(nm, timeit([&] { k_fill_rows<1, 32><<<grid, 256>>>(buf, nrows, 1.f); }, 20));
    snprintf(nm, 64, "rows32 plain grid=%d", grid); rep(nm, timeit([&] { k_fill_rows<0, 32><<<grid, 256>>>(buf, nrows, 1.f); }, 20));
    snprintf(nm, 64, "rows8 nt grid=%d", grid);    rep(nm, timeit([&] { k_fill_rows<1, 8><<<grid, 256>>>(buf, nrows, 1.f); }, 20));
    snprintf(nm, 64, "flat nt grid=%d", grid);     rep(nm, timeit([&] { k_fill_flat<1><<<grid, 256>>>(buf, bytes / 16, 1.f); }, 20));
    snprintf(nm, 64, "flat plain grid=%d", grid);  rep(nm, timeit([&] { k_fill_flat<0><<<grid, 256>>>(buf, bytes / 16, 1.f); }, 20));
  }
  return 0;
}

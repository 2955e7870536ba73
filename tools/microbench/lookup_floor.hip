// Floor for the lookup families of the tile kernel: F features, each a ROWS-row table of 256 groups
// (float), score[n][k] = sum_f table_f[value_f[n]][k].  All tables of the tile live in LDS at once
// (one barrier per chunk), lane <-> 4 groups, wave <-> 8 rows: the inner loop is readlane + address
// add + ds_read_b128 + 4 adds per (row, feature).  Compare with bb x16 / dd32 x16 of
// profiles/r01_c3_stage_costs.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template <int F, int ROWS, int R, int W>
__global__ __launch_bounds__(W * 64) void k_lookup(const float *__restrict__ tabs /*[F][ROWS][256]*/,
                                                    const int *__restrict__ vals /*[F][N]*/, uint64_t nrows,
                                                    float *__restrict__ out) {
  extern __shared__ f4 lds[];                       // [F*ROWS][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < F * ROWS * 64; i += W * 64) lds[i] = ((const f4 *)tabs)[i];
  __syncthreads();
  const uint64_t rows_per_wg = (uint64_t)W * R, nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * R;
    f4 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (f4){0, 0, 0, 0};
#pragma unroll 2
    for (int f = 0; f < F; f++) {
      const int v = (lane < R && rb + lane < nrows) ? vals[(size_t)f * nrows + rb + lane] : 0;
      const f4 *t = lds + f * ROWS * 64 + lane;
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] += t[__builtin_amdgcn_readlane(v, r) * 64];
    }
#pragma unroll
    for (int r = 0; r < R; r++)
      if (rb + r < nrows) __builtin_nontemporal_store(acc[r], (f4 *)(out + (rb + r) * 256) + lane);
  }
}
template <typename Fn> float timeit(Fn f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int F, int ROWS, int R, int W> void run(const char *name) {
  const uint64_t N = 1000000;
  std::vector<float> ht((size_t)F * ROWS * 256); for (auto &x : ht) x = -(float)rand() / RAND_MAX;
  std::vector<int> hv((size_t)F * N); for (auto &x : hv) x = rand() % ROWS;
  float *tabs, *out; int *vals;
  CK(hipMalloc(&tabs, ht.size() * 4)); CK(hipMalloc(&vals, hv.size() * 4)); CK(hipMalloc(&out, N * 1024));
  CK(hipMemcpy(tabs, ht.data(), ht.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vals, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)F * ROWS * 1024;
  CK(hipFuncSetAttribute((const void *)k_lookup<F, ROWS, R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = 256 * (lds > 80 * 1024 ? 1 : 2);
  const float ms = timeit([&] { k_lookup<F, ROWS, R, W><<<grid, W * 64, lds>>>(tabs, vals, N, out); }, 10);
  printf("%-34s %7.3f ms  (%.0f cycles per feature and 128-row chunk at 2.4 GHz)\n", name, ms,
         (ms - 0.17) * 1e-3 * 2.4e9 / F / (N / (float)(W * R) / 256));
  CK(hipFree(tabs)); CK(hipFree(vals)); CK(hipFree(out));
}
int main() {
  run<16, 2, 8, 16>("bb-like  x16, 16 waves x 8 rows");
  run<16, 2, 16, 8>("bb-like  x16,  8 waves x 16 rows");
  run<4, 32, 8, 16>("dd32-like x4, 16 waves x 8 rows");
  run<48, 2, 8, 16>("bb-like  x48, 16 waves x 8 rows");
  return 0;
}

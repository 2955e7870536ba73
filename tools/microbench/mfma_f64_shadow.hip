// What does a vector instruction cost in the shadow of v_mfma_f64_16x16x4_f64 on gfx950?  (C4: the squares, lane sums
// and logarithms between two groups' matrix instructions -- do they run under the 64-cycle matrix instruction or beside it?)
// A loop of 16 matrix instructions on 4 accumulators, each followed by N independent vector instructions of one kind,
// 2 waves per SIMD, the whole chip; printed: cycles (at the measured rate of the bare loop = 64) per matrix instruction.
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_shadow.hip -o mfma_f64_shadow && ./mfma_f64_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

// KIND 0 none, 1 v_fma_f64, 2 v_add_f64, 3 v_fma_f32, 4 v_mov_b32, 5 v_cndmask_b32, 6 v_permlane16_swap, 7 v_mul_f64
template <int KIND, int N>
__global__ __launch_bounds__(256) void k_shadow(double *out, int iters, double a0, double b0) {
  f64x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double d[8];
  float f[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { d[i] = a0 * (i + 1); f[i] = (float)b0 * (i + 1); u[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; n++) {
        const int j = (r * N + n) & 7;
        if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[j]) : "v"(a));
        if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(a));
        if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[j]) : "v"(f[(j + 1) & 7]));
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        if (KIND == 6) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[j]), "+v"(u[(j + 1) & 7]));
        if (KIND == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(a));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; i++) s += d[i] + f[i] + u[i];
  if (s == 12345.678) out[0] = s;
}

static double g_bare = 0;
template <int KIND, int N> void run(double *d, const char *name) {
  const int iters = 1000, wgs = 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_shadow<KIND, N><<<256 * wgs, 256>>>(d, 10, 1.0, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_shadow<KIND, N><<<256 * wgs, 256>>>(d, iters, 1.0, 1.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double per_simd = (double)iters * 16 * wgs;
  const double ns = ms * 1e6 / per_simd;
  if (KIND == 0) g_bare = ns;
  printf("%-22s x %2d per matrix instruction: %6.1f ns = %6.1f cycles a matrix instruction (bare = 64), +%.1f cycles per vector instruction\n",
         name, N, ns, ns / g_bare * 64.0, N ? (ns / g_bare * 64.0 - 64.0) / N : 0.0);
}
template <int KIND> void kinds(double *d, const char *name) {
  run<KIND, 1>(d, name); run<KIND, 2>(d, name); run<KIND, 4>(d, name); run<KIND, 8>(d, name);
}
int main() {
  double *d; CK(hipMalloc(&d, 64));
  run<0, 0>(d, "none");
  kinds<1>(d, "v_fma_f64"); kinds<2>(d, "v_add_f64"); kinds<7>(d, "v_mul_f64"); kinds<3>(d, "v_fma_f32");
  kinds<4>(d, "v_mov_b32"); kinds<5>(d, "v_cndmask_b32"); kinds<6>(d, "v_permlane16_swap_b32");
  return 0;
}

// What rate does v_mfma_f64_16x16x4_f64 reach on gfx950?  (C4's roofline prices the f64 matrix pipe at 78.6 TFLOP/s = one
// instruction per 64 cycles and SIMD at 2.4 GHz; the kernel's SQ_VALU_MFMA_BUSY_CYCLES counts 64 cycles an instruction.)
// Bare loops of the instruction, NACC independent accumulators a wave, 1 / 2 waves per SIMD, the whole chip:
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
  f64x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

template <int NACC> void run(double *d, int wgs_per_cu) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_mfma<NACC><<<256 * wgs_per_cu, 256>>>(d, 10, 1.0, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_mfma<NACC><<<256 * wgs_per_cu, 256>>>(d, iters, 1.0, 1.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)iters * 16 * NACC;                 // instructions per wave
  const double per_simd = n * wgs_per_cu;                      // one wave of each workgroup per SIMD
  const double flops = per_simd * 1024.0 * 2048.0;
  printf("%d accumulators, %d wave(s)/SIMD: %.3f ms, %.1f ns per instruction and SIMD (= %.1f cycles at 2.4 GHz), %.1f TFLOP/s\n",
         NACC, wgs_per_cu, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, flops / (ms * 1e-3) / 1e12);
}
int main() {
  double *d; CK(hipMalloc(&d, 64));
  run<1>(d, 1); run<2>(d, 1); run<4>(d, 1); run<8>(d, 1);
  run<1>(d, 2); run<4>(d, 2); run<8>(d, 2);
  return 0;
}

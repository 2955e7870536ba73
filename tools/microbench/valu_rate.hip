// What rate do plain vector instructions reach on a gfx950 SIMD with 1, 2, 4, 8 waves on it?  (MI355X_MICROARCH.md: a
// SIMD is 32 lanes wide -- a wave64 v_fma_f32 takes 2 cycles -- but ONE wave alone issues one vector instruction per 4
// cycles; bench.py's valu_issue roof was the 4-cycle figure.)  Loops of independent instructions of one kind, 16
// accumulators a wave, W waves per SIMD (workgroups of 64 threads, 4 W per CU), whole chip; printed: cycles per wave
// instruction and SIMD at the clock the run held (s_memtime / wall clock).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

// KIND 0 v_fma_f32, 1 v_add_f32, 2 v_exp_f32, 3 v_fma_f64, 4 v_mov_b32, 5 v_pk_fma_f32, 6 v_log_f32, 7 v_rcp_f32
template <int KIND>
__global__ __launch_bounds__(64) void k_valu(float *out, int iters, float a0) {
  float f[16];
  double d[8];
#pragma unroll
  for (int i = 0; i < 16; i++) f[i] = a0 * (i + 1) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = a0 * (i + 1) + threadIdx.x;
  const float a = a0;
  const double ad = a0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(a));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i & 7]) : "v"(ad));
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
        if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i & 7]) : "v"(ad));
        if (KIND == 6) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
        if (KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += f[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (float)d[i];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND> void run(float *d, const char *name) {
  printf("%-14s", name);
  for (int W : {1, 2, 4, 8}) {
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_valu<KIND><<<256 * 4 * W, 64>>>(d, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_valu<KIND><<<256 * 4 * W, 64>>>(d, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)iters * 64 * W;           // wave instructions a SIMD issued (if the waves spread evenly)
    printf("  W=%d: %6.2f ns/instr/SIMD = %5.2f cyc @2.4GHz", W, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
  printf("\n");
}
int main() {
  float *d; CK(hipMalloc(&d, 64));
  run<0>(d, "v_fma_f32"); run<1>(d, "v_add_f32"); run<4>(d, "v_mov_b32"); run<5>(d, "v_pk_fma_f32");
  run<2>(d, "v_exp_f32"); run<6>(d, "v_log_f32"); run<7>(d, "v_rcp_f32"); run<3>(d, "v_fma_f64");
  return 0;
}

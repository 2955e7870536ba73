// Issue cost of the vector instructions the nich blocks and the lookup runs are made of, one wave's stream on one SIMD
// (and two / four waves a SIMD): cycles per wave-instruction by s_memtime around an unrolled run of INDEPENDENT instructions.
//   hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue && ./valu_issue
// Settles what a v_pk_*_f32 costs on gfx950 (the guide prices plain f32 at 4 cycles and the transcendentals at 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(unsigned long long *out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  f2 p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
  int s0 = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; it++) {
    if (KIND == 0) {   // v_fma_f32, 8 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                        "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 1) {   // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (KIND == 2) {   // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n"
                        "v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (KIND == 3) {   // v_add_f32
      REP8(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                        "v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) {   // v_log_f32
      REP8(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                        "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 5) {   // v_readlane_b32 (to distinct sgprs) interleaved with nothing
      REP8(asm volatile("v_readlane_b32 s40, %0, 1\n v_readlane_b32 s41, %1, 2\n v_readlane_b32 s42, %2, 3\n v_readlane_b32 s43, %3, 4\n"
                        "v_readlane_b32 s44, %4, 5\n v_readlane_b32 s45, %5, 6\n v_readlane_b32 s46, %6, 7\n v_readlane_b32 s47, %7, 8"
                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
                        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");)
    } else if (KIND == 6) {   // v_lshl_add_u32 with an sgpr operand
      REP8(asm volatile("v_lshl_add_u32 %0, s40, 10, %0\n v_lshl_add_u32 %1, s40, 10, %1\n v_lshl_add_u32 %2, s40, 10, %2\n v_lshl_add_u32 %3, s40, 10, %3\n"
                        "v_lshl_add_u32 %4, s40, 10, %4\n v_lshl_add_u32 %5, s40, 10, %5\n v_lshl_add_u32 %6, s40, 10, %6\n v_lshl_add_u32 %7, s40, 10, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s40");)
    } else if (KIND == 7) {   // plain v_fma interleaved with s_add (does the scalar unit issue beside the vector pipe?)
      REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %2, %2\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %8, %8, 1\n"
                        "v_fma_f32 %4, %4, %4, %4\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %5, %5\n s_add_u32 %8, %8, 1\n v_fma_f32 %6, %6, %6, %6\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %7, %7\n s_add_u32 %8, %8, 1"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0));)
    } else if (KIND == 8) {   // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                        "v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_mul_f32 %7, %7, %7"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (KIND == 9) {   // v_pk_fma_f32 with an SGPR-pair first operand broadcast (op_sel_hi:[0,1,1]), the nich blocks' form
      REP8(asm volatile("v_pk_fma_f32 %0, s[40:41], %0, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, s[40:41], %1, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, s[40:41], %2, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, s[40:41], %3, %3 op_sel_hi:[0,1,1]\n"
                        "v_pk_fma_f32 %4, s[40:41], %4, %4 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5, s[40:41], %5, %5 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %6, s[40:41], %6, %6 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7, s[40:41], %7, %7 op_sel_hi:[0,1,1]"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) :: "s40", "s41");)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + (float)s0;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink == 12345.678f) out[0] = 0;
}

template <int KIND> void run(const char *name, unsigned long long *d) {
  for (int waves = 1; waves <= 4; waves *= 2) {
    // one workgroup per CU of `waves` x 4 waves: waves per SIMD = waves
    k_issue<KIND><<<256, 256 * waves>>>(d, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    k_issue<KIND><<<256, 256 * waves>>>(d, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    const double n = 64.0 * 64.0;   // instructions per wave
    printf("%-44s %d waves/SIMD: %6.2f counter ticks per wave-instruction (wave 0 of CU 0: %llu ticks / %.0f), kernel %.1f us -> %.2f ns per instruction and SIMD\n",
           name, waves, (double)h[0] / n, h[0], n, ms * 1e3, ms * 1e6 / (n * waves));
  }
}
int main() {
  unsigned long long *d; CK(hipMalloc(&d, 256 * 8));
  run<0>("v_fma_f32", d);
  run<3>("v_add_f32", d);
  run<1>("v_pk_fma_f32", d);
  run<9>("v_pk_fma_f32 sgpr-pair op_sel_hi:[0,1,1]", d);
  run<2>("v_pk_add_f32", d);
  run<8>("v_pk_mul_f32", d);
  run<4>("v_log_f32", d);
  run<5>("v_readlane_b32", d);
  run<6>("v_lshl_add_u32 (sgpr operand)", d);
  run<7>("v_fma_f32 + s_add_u32 alternating", d);
  return 0;
}

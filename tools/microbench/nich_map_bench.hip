// mapping experiments for the C2 scoring pass (N=1M rows x 256 groups, one NICH feature)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
#define DEV __device__ __forceinline__
constexpr float kLn2f = 0.69314718055994530942f;
DEV float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// shipped formulation: 10 VALU + log + rcp
DEV float eval_a(float x, float mh, float ml, float c0, float c1l, float c1, float c2) {
  const float d = (x - mh) - ml;
  const float t = c2 * d * d;
  const float u = 1.0f + t;
  const float l2 = __builtin_amdgcn_logf(u);
  const float r = (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u);
  return fmaf(-c1, r, fmaf(-c1l, l2, c0));
}
// scaled formulation: a = s*x - (m_hi + m_lo), t = a*a  (9 VALU + log + rcp); here mh,ml hold s*mu, c2 holds s
DEV float eval_b(float x, float mh, float ml, float c0, float c1l, float c1, float s) {
  const float a = fmaf(x, s, -mh) - ml;
  const float t = a * a;
  const float u = 1.0f + t;
  const float l2 = __builtin_amdgcn_logf(u);
  const float r = (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u);
  return fmaf(-c1, r, fmaf(-c1l, l2, c0));
}
struct Tab { const float *mh, *ml, *c0, *c1l, *c1, *c2; };
#define LOADC                                                                               \
  const int lane = threadIdx.x & 63;                                                         \
  const f4 mh = *(const f4 *)(T.mh + lane * 4), ml = *(const f4 *)(T.ml + lane * 4),         \
           c0 = *(const f4 *)(T.c0 + lane * 4), c1l = *(const f4 *)(T.c1l + lane * 4),       \
           c1 = *(const f4 *)(T.c1 + lane * 4), c2 = *(const f4 *)(T.c2 + lane * 4);
#define EVAL4(E, x)                                                                          \
  f4 s;                                                                                      \
  s.x = E(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x); s.y = E(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y); \
  s.z = E(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z); s.w = E(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);

// V0: replica of the shipped loop (dynamic nr, 32-row chunks per wave, grid-stride)
template <int CH, int FORM>
__global__ __launch_bounds__(256) void k_v0(Tab T, const float *__restrict__ xcol, uint64_t nrows, float *__restrict__ out) {
  LOADC
  const uint64_t nchunks = (nrows + CH - 1) / CH;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * CH;
    const int nr = (int)((nrows - rb) < (uint64_t)CH ? (nrows - rb) : (uint64_t)CH);
    const float xv = lane < nr ? xcol[rb + lane] : 0.f;
    if (nr == CH) {
#pragma unroll
      for (int r = 0; r < CH; r++) {
        const float x = bcast(xv, r);
        if (FORM == 0) { EVAL4(eval_a, x) __builtin_nontemporal_store(s, (f4 *)(out + (rb + r) * 256) + lane); }
        else           { EVAL4(eval_b, x) __builtin_nontemporal_store(s, (f4 *)(out + (rb + r) * 256) + lane); }
      }
    } else {
      for (int r = 0; r < nr; r++) {
        const float x = bcast(xv, r);
        EVAL4(eval_a, x) __builtin_nontemporal_store(s, (f4 *)(out + (rb + r) * 256) + lane);
      }
    }
  }
}
// V1: dense front. wave w handles row quads {4w .. 4w+3} + i * 4*NW (one-shot grid covers nrows/ (4*ITER) waves)
template <int Q, int NT>
__global__ __launch_bounds__(256) void k_v1(Tab T, const float *__restrict__ xcol, uint64_t nrows, float *__restrict__ out) {
  LOADC
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t rb = wave_id * Q; rb < nrows; rb += nwaves * Q) {
    const float xv = (lane < Q && rb + lane < nrows) ? xcol[rb + lane] : 0.f;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      if (rb + r < nrows) {
        const float x = bcast(xv, r);
        EVAL4(eval_a, x)
        if (NT) __builtin_nontemporal_store(s, (f4 *)(out + (rb + r) * 256) + lane);
        else *((f4 *)(out + (rb + r) * 256) + lane) = s;
      }
    }
  }
}
// V2: workgroup-interleaved: block handles 128-row chunk, wave wv takes rows wv, wv+4, ...
__global__ __launch_bounds__(256) void k_v2(Tab T, const float *__restrict__ xcol, uint64_t nrows, float *__restrict__ out) {
  LOADC
  const int wv = threadIdx.x >> 6;
  const uint64_t nchunks = (nrows + 127) / 128;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * 128;
    // lane l holds x[rb + 4*l + wv] for l < 32
    const float xv = (lane < 32 && rb + 4 * lane + wv < nrows) ? xcol[rb + 4 * lane + wv] : 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; r++) {
      const uint64_t row = rb + 4 * r + wv;
      if (row < nrows) {
        const float x = bcast(xv, r);
        EVAL4(eval_a, x) __builtin_nontemporal_store(s, (f4 *)(out + row * 256) + lane);
      }
    }
  }
}
// V3: compute 4 rows, then issue the 4 stores back to back (4 KiB burst per wave)
__global__ __launch_bounds__(256) void k_v3(Tab T, const float *__restrict__ xcol, uint64_t nrows, float *__restrict__ out) {
  LOADC
  const uint64_t nchunks = nrows / 32;   // bench sizes are multiples of 32
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * 32;
    const float xv = lane < 32 ? xcol[rb + lane] : 0.f;
#pragma unroll 2
    for (int r = 0; r < 32; r += 4) {
      f4 o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const float x = bcast(xv, r + j); EVAL4(eval_a, x) o[j] = s; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) __builtin_nontemporal_store(o[j], (f4 *)(out + (rb + r + j) * 256) + lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const uint64_t nrows = 1000000; const size_t bytes = nrows * 1024;
  float *out, *x, *tab; CK(hipMalloc(&out, bytes)); CK(hipMalloc(&x, nrows * 4)); CK(hipMalloc(&tab, 6 * 256 * 4));
  std::vector<float> hx(nrows), ht(6 * 256);
  srand(1);
  for (auto &v : hx) v = 6.f * rand() / RAND_MAX - 3.f;
  for (int k = 0; k < 256; k++) {
    const double mu = 6.0 * rand() / RAND_MAX - 3.0, c1 = 1950 + k, c2 = 1.0 / (2 * c1);
    ht[k] = (float)mu; ht[256 + k] = (float)(mu - (double)(float)mu); ht[512 + k] = -1.f;
    ht[768 + k] = (float)(c1 * 0.6931471805599453); ht[1024 + k] = (float)c1; ht[1280 + k] = (float)c2;
  }
  CK(hipMemcpy(x, hx.data(), nrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tab, ht.data(), 6 * 256 * 4, hipMemcpyHostToDevice));
  Tab T{tab, tab + 256, tab + 512, tab + 768, tab + 1024, tab + 1280};
  auto rep = [&](const char *name, float ms) { printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, (bytes + nrows * 4) / ms / 1e9); fflush(stdout); };
  const int R = 30;
  char nm[96];
  for (int pass = 0; pass < 2; pass++) {
    rep("v0 ch32 formA one-shot(7813)", timeit([&] { k_v0<32, 0><<<7813, 256>>>(T, x, nrows, out); }, R));
    rep("v0 ch32 formB one-shot(7813)", timeit([&] { k_v0<32, 1><<<7813, 256>>>(T, x, nrows, out); }, R));
    rep("v0 ch16 formA one-shot(15625)", timeit([&] { k_v0<16, 0><<<15625, 256>>>(T, x, nrows, out); }, R));
    rep("v0 ch64 formA one-shot(3907)", timeit([&] { k_v0<64, 0><<<3907, 256>>>(T, x, nrows, out); }, R));
    for (int g : {2048, 4096}) { snprintf(nm, 96, "v0 ch32 formA grid=%d", g); rep(nm, timeit([&] { k_v0<32, 0><<<g, 256>>>(T, x, nrows, out); }, R)); }
    for (int g : {2048, 7813, 15625, 31250}) {
      snprintf(nm, 96, "v1 dense quad4 nt grid=%d", g); rep(nm, timeit([&] { k_v1<4, 1><<<g, 256>>>(T, x, nrows, out); }, R));
      snprintf(nm, 96, "v1 dense quad4 plain grid=%d", g); rep(nm, timeit([&] { k_v1<4, 0><<<g, 256>>>(T, x, nrows, out); }, R));
      snprintf(nm, 96, "v1 dense quad8 nt grid=%d", g); rep(nm, timeit([&] { k_v1<8, 1><<<g, 256>>>(T, x, nrows, out); }, R));
      snprintf(nm, 96, "v1 dense quad1 nt grid=%d", g); rep(nm, timeit([&] { k_v1<1, 1><<<g, 256>>>(T, x, nrows, out); }, R));
    }
    for (int g : {2048, 7813}) { snprintf(nm, 96, "v2 wg-interleaved grid=%d", g); rep(nm, timeit([&] { k_v2<<<g, 256>>>(T, x, nrows, out); }, R)); }
    for (int g : {2048, 7813}) { snprintf(nm, 96, "v3 burst4 grid=%d", g); rep(nm, timeit([&] { k_v3<<<g, 256>>>(T, x, nrows, out); }, R)); }
  }
  // sanity: checksum of a few outputs
  std::vector<float> ho(1024); CK(hipMemcpy(ho.data(), out + 999999ull * 256, 1024, hipMemcpyDeviceToHost));
  double cs = 0; for (int i = 0; i < 256; i++) cs += ho[i]; printf("checksum last row %.6f\n", cs);
  return 0;
}

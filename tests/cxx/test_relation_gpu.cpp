// GPU: the C++ relation dataviews' bridge to the device (include/microscopes_amd/relation.hpp):
// dense and compressed relations as cell dataviews, cell -> block indices, block suff-stats.
#include <microscopes/common/relation/dataview.hpp>
#include <microscopes/models/distributions.hpp>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "audit.hpp"

using namespace microscopes;
using namespace microscopes::common;
using namespace microscopes::common::relation;

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

template <typename T> static T *to_dev(const std::vector<T> &h) {
  T *d = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(T) * (h.size() ? h.size() : 1)) == hipSuccess);
  CHECK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) == hipSuccess);
  return d;
}

int main() {
  msc_context *ctx = hip::default_context();
  std::mt19937 r(11);
  const size_t n = 23, m = 31;
  const uint32_t K0 = 3, K1 = 4;
  std::vector<uint8_t> data(n * m), mask(n * m);
  for (size_t i = 0; i < n * m; i++) { data[i] = r() & 1; mask[i] = (r() % 10) < 4; }
  std::vector<int32_t> z0(n), z1(m);
  for (auto &z : z0) z = int32_t(r() % K0);
  for (auto &z : z1) z = int32_t(r() % K1);
  int32_t *z0d = to_dev(z0), *z1d = to_dev(z1);
  // expected block counts from the host-side slices
  std::vector<uint32_t> heads(K0 * K1, 0), tails(K0 * K1, 0);
  row_major_dense_dataview dense(data.data(), reinterpret_cast<const bool *>(mask.data()), {n, m}, runtime_type(TYPE_B));
  for (size_t i = 0; i < n; i++)
    for (const auto &p : dense.slice(0, i)) {
      const uint32_t b = uint32_t(z0[p.first[0]]) * K1 + uint32_t(z1[p.first[1]]);
      (p.second.get<bool>(0) ? heads : tails)[b]++;
    }
  const msc_feature_spec spec = {MSC_BB, 0};
  // dense: cells in row-major order, masked cells skipped by the kernels
  {
    msc_dataview *cells = dense.to_device_cells(ctx);
    int32_t *zc = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&zc), 4 * n * m) == hipSuccess);
    cell_blocks(ctx, dense.shape(), {z0d, z1d}, {K0, K1}, nullptr, n * m, zc);
    msc_state *st = nullptr;
    hip::check(msc_state_create(ctx, &spec, 1, K0 * K1, &st));
    hip::check(msc_accumulate(st, cells, nullptr, 0, n * m, zc, MSC_ACC_RESET));
    std::vector<uint32_t> rec(2 * K0 * K1);
    hip::check(msc_state_get_ss(st, 0, 0, K0 * K1, rec.data(), rec.size() * 4));
    for (uint32_t b = 0; b < K0 * K1; b++) CHECK(rec[2 * b] == heads[b] && rec[2 * b + 1] == tails[b]);
    msc_state_destroy(st);
    msc_dataview_destroy(cells);
    (void)hipFree(zc);
  }
  // compressed: only the stored entries are cells, their positions travel with them
  {
    std::vector<uint8_t> csr_d, csc_d;
    std::vector<uint32_t> csr_i, csr_p(1, 0), csc_i, csc_p(1, 0);
    for (size_t i = 0; i < n; i++) {
      for (size_t j = 0; j < m; j++) if (!mask[i * m + j]) { csr_d.push_back(data[i * m + j]); csr_i.push_back(uint32_t(j)); }
      csr_p.push_back(uint32_t(csr_i.size()));
    }
    for (size_t j = 0; j < m; j++) {
      for (size_t i = 0; i < n; i++) if (!mask[i * m + j]) { csc_d.push_back(data[i * m + j]); csc_i.push_back(uint32_t(i)); }
      csc_p.push_back(uint32_t(csc_i.size()));
    }
    compressed_2darray sparse(csr_d.data(), csr_i.data(), csr_p.data(), csc_d.data(), csc_i.data(), csc_p.data(), n, m,
                              runtime_type(TYPE_B));
    std::vector<uint32_t> pos;
    msc_dataview *cells = sparse.to_device_cells(ctx, &pos);
    uint32_t *posd = to_dev(pos);
    int32_t *zc = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&zc), 4 * (sparse.nnz() + 1)) == hipSuccess);
    cell_blocks(ctx, sparse.shape(), {z0d, z1d}, {K0, K1}, posd, sparse.nnz(), zc);
    msc_state *st = nullptr;
    hip::check(msc_state_create(ctx, &spec, 1, K0 * K1, &st));
    hip::check(msc_accumulate(st, cells, nullptr, 0, sparse.nnz(), zc, MSC_ACC_RESET));
    std::vector<uint32_t> rec(2 * K0 * K1);
    hip::check(msc_state_get_ss(st, 0, 0, K0 * K1, rec.data(), rec.size() * 4));
    for (uint32_t b = 0; b < K0 * K1; b++) CHECK(rec[2 * b] == heads[b] && rec[2 * b + 1] == tails[b]);
    // irm's slice reduction over the compressed relation, both dimensions: the device sums the per-cell scores of each
    // slice per candidate cluster; the host does the same walk through compressed_2darray::slice
    // (test/cxx/test_relation.cpp:281-323 is how the reference exercises those iterators)
    {
      const uint64_t nnz = sparse.nnz(), nblk = K0 * K1;
      float *sc = nullptr;
      CHECK(hipMalloc(reinterpret_cast<void **>(&sc), 4 * nnz * nblk) == hipSuccess);
      hip::check(msc_score_value(st, cells, nullptr, 0, nnz, nullptr, 0, sc, nblk));
      std::vector<float> sch(nnz * nblk);
      hip::check(msc_device_download(ctx, sch.data(), sc, 4 * nnz * nblk));
      // the stored entries are the cells in CSR order: cell id of (i, j) = position in csr_i
      std::vector<int32_t> zeros0(n, 0), zeros1(m, 0);
      int32_t *zero0d = to_dev(zeros0), *zero1d = to_dev(zeros1), *off = nullptr;
      CHECK(hipMalloc(reinterpret_cast<void **>(&off), 4 * (nnz + 1)) == hipSuccess);
      for (size_t dim = 0; dim < 2; dim++) {
        cell_blocks(ctx, sparse.shape(), {dim == 0 ? zero0d : z0d, dim == 1 ? zero1d : z1d}, {K0, K1}, posd, nnz, off);
        // rows of the CSR (dim 0) or of its transpose (dim 1), as cell ids
        std::vector<uint32_t> seg(1, 0), ids;
        const size_t nent = dim == 0 ? n : m;
        for (size_t e = 0; e < nent; e++) {
          if (dim == 0) for (uint32_t p = csr_p[e]; p < csr_p[e + 1]; p++) ids.push_back(p);
          else for (size_t i = 0; i < n; i++)
            for (uint32_t p = csr_p[i]; p < csr_p[i + 1]; p++) if (csr_i[p] == e) ids.push_back(p);
          seg.push_back(uint32_t(ids.size()));
        }
        uint32_t *segd = to_dev(seg), *idsd = to_dev(ids);
        const uint32_t ncand = dim == 0 ? K0 : K1;
        float *outd = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&outd), 4 * nent * ncand) == hipSuccess);
        slice_scores(ctx, sc, nblk, sparse.shape(), dim, segd, idsd, off, {K0, K1}, nent, outd, ncand);
        std::vector<float> got(nent * ncand);
        hip::check(msc_device_download(ctx, got.data(), outd, 4 * nent * ncand));
        for (size_t e = 0; e < nent; e++)
          for (uint32_t g = 0; g < ncand; g++) {
            double want = 0.0;
            size_t cells_seen = 0;
            for (const auto &pv : sparse.slice(dim, e)) {
              const size_t i = pv.first[0], j = pv.first[1];
              uint32_t p = csr_p[i];
              while (csr_i[p] != j) p++;                              // the cell id of (i, j)
              const uint32_t b = dim == 0 ? g * K1 + uint32_t(z1[j]) : uint32_t(z0[i]) * K1 + g;
              want += double(sch[size_t(p) * nblk + b]);
              cells_seen++;
            }
            CHECK(cells_seen == seg[e + 1] - seg[e]);
            // the kernel adds the cells' float scores in double, in a fixed order, and rounds the sum to float once: half an
            // ulp of the result, inside the plain gate (rounds 1-3: 1e-5)
            CHECK(audit::score("relation.slice_scores.compressed_2darray", double(got[e * ncand + g]), want));
          }
        (void)hipFree(segd); (void)hipFree(idsd); (void)hipFree(outd);
      }
      (void)hipFree(sc); (void)hipFree(off); (void)hipFree(zero0d); (void)hipFree(zero1d);
    }
    msc_state_destroy(st);
    msc_dataview_destroy(cells);
    (void)hipFree(zc);
    (void)hipFree(posd);
  }
  (void)hipFree(z0d);
  (void)hipFree(z1d);
  audit::dump();
  std::printf("test_relation_gpu ok\n");
  return 0;
}

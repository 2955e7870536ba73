// CPU-only checks of the host side of the plugin surface (no device needed): the runtime
// type system, the packed-record accessors, the CRP group manager and the wire format.
// Data of the first three blocks is the data the reference's own tests hold
// (test/test_dataview.py:31-75, test/cxx/test_group_manager.cpp:22-66).
#include <microscopes/common/relation/dataview.hpp>
#include <microscopes/common/group_manager.hpp>
#include <microscopes/common/recarray/dataview.hpp>
#include <microscopes/models/noop.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace microscopes::common;
using namespace microscopes::common::recarray;

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

static void test_types_and_offsets() {
  CHECK(runtime_type(TYPE_B).size() == 1 && runtime_type(TYPE_F64).size() == 8);
  CHECK(runtime_type(TYPE_F32, 3).size() == 12 && runtime_type(TYPE_F32, 3).vec());
  CHECK(runtime_type(TYPE_I32) == runtime_type(TYPE_I32) && runtime_type(TYPE_I32) != runtime_type(TYPE_I32, 1));
  CHECK(runtime_type(TYPE_U16, 4).str() == "TYPE_U16[4]");
  // (bool, float64) records: offsets 0, 1; rowsize 9; mask row 2
  std::vector<runtime_type> t1 = {runtime_type(TYPE_B), runtime_type(TYPE_F64)};
  auto r = runtime_type::GetOffsetsAndSize(t1);
  CHECK(r.offsets_[0] == 0 && r.offsets_[1] == 1 && r.rowsize_ == 9 && r.maskrowsize_ == 2);
  // (int32, float32[2]) records: offsets 0, 4; rowsize 12; mask row 3
  std::vector<runtime_type> t2 = {runtime_type(TYPE_I32), runtime_type(TYPE_F32, 2)};
  r = runtime_type::GetOffsetsAndSize(t2);
  CHECK(r.offsets_[1] == 4 && r.rowsize_ == 12 && r.maskrowsize_ == 3);
  // casts follow the implicit C++ conversion
  const double dv = -2.75;
  CHECK(runtime_cast::cast<int32_t>(reinterpret_cast<const uint8_t *>(&dv), TYPE_F64) == -2);
  CHECK(runtime_cast::cast<bool>(reinterpret_cast<const uint8_t *>(&dv), TYPE_F64) == true);
  uint8_t u = 200;
  CHECK(runtime_cast::cast<float>(&u, TYPE_U8) == 200.f);
  float f = 0;
  runtime_cast::uncast<int>(reinterpret_cast<uint8_t *>(&f), TYPE_F32, 7);
  CHECK(f == 7.f);
}

static void test_row_accessor_over_reference_rows() {
  // [(False, 32.), (True, 943.), (False, -32.)] with dtype (bool, float64)
  uint8_t buf[27];
  const bool b[3] = {false, true, false};
  const double d[3] = {32., 943., -32.};
  for (int i = 0; i < 3; i++) {
    buf[9 * i] = b[i];
    std::memcpy(buf + 9 * i + 1, &d[i], 8);
  }
  std::vector<runtime_type> types = {runtime_type(TYPE_B), runtime_type(TYPE_F64)};
  row_major_dataview view(buf, nullptr, 3, types);
  CHECK(view.size() == 3);
  int acc = 0, rows = 0;
  for (view.reset(); !view.end(); view.next(), rows++) {
    row_accessor a = view.get();
    CHECK(a.nfeatures() == 2 && !a.anymasked());
    CHECK(a.get().get<bool>(0) == b[view.index()]);
    acc += a.get().get<int>(0);
    a.bump();
    CHECK(a.get().get<double>(0) == d[view.index()]);
    CHECK(a.get().get<float>(0) == float(d[view.index()]));   // cross-type read
    a.bump();
    CHECK(a.end());
  }
  CHECK(rows == 3 && acc == 1);
  // permutation visits every row exactly once
  rng_t rng(73);
  view.permute(rng);
  std::vector<int> seen(3, 0);
  for (view.reset(); !view.end(); view.next()) seen[view.index()]++;
  CHECK(seen[0] == 1 && seen[1] == 1 && seen[2] == 1);
  // masked record: one row of five bools, last three masked
  const uint8_t vals[5] = {1, 0, 1, 1, 1};
  const bool mask[5] = {false, false, true, true, true};
  std::vector<runtime_type> t5(5, runtime_type(TYPE_B));
  row_accessor m(vals, mask, &t5);
  for (int i = 0; i < 5; i++, m.bump()) {
    CHECK(m.ismasked(0) == mask[i]);
    if (!mask[i]) CHECK(m.get().get<bool>(0) == (vals[i] != 0));
  }
  // row_mutator copies with conversion
  uint8_t out[12] = {0};
  std::vector<runtime_type> tf = {runtime_type(TYPE_F32), runtime_type(TYPE_F64)};
  row_mutator mut(out, &tf);
  row_accessor src(buf + 9, nullptr, &types);   // (True, 943.)
  mut.set(src); mut.bump(); src.bump();
  mut.set(src);
  float f0; double d1;
  std::memcpy(&f0, out, 4); std::memcpy(&d1, out + 4, 8);
  CHECK(f0 == 1.f && d1 == 943.);
}

// group_manager.hpp:320-386: ids in creation order, never reused; lookups of a deleted id fail
static void test_simple_group_manager() {
  simple_group_manager<std::string> g;
  CHECK(g.ngroups() == 0 && g.groups().empty() && g.begin() == g.end());
  auto a = g.create_group();
  a.second = "a";
  auto b = g.create_group();
  b.second = "b";
  auto c = g.create_group();
  c.second = "c";
  CHECK(a.first == 0 && b.first == 1 && c.first == 2 && g.ngroups() == 3);
  g.delete_group(1);
  CHECK(g.ngroups() == 2 && g.groups() == std::vector<size_t>({0, 2}));
  CHECK(g.create_group().first == 3);                              // 1 is not handed out again
  CHECK(g.group(2) == "c");
  g.group(0) += "!";
  const simple_group_manager<std::string> &cg = g;
  CHECK(cg.group(0) == "a!");
  bool threw = false;
  try { g.group(1); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  threw = false;
  try { g.delete_group(7); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  size_t n = 0;
  for (auto it = g.begin(); it != g.end(); ++it) n += it->first;
  CHECK(n == 0 + 2 + 3);
}

static void test_group_manager_bookkeeping_and_serialization() {
  typedef group_manager<size_t> gm;
  gm g(10);
  g.get_hp_mutator("alpha").set<float>(2.0f, 0);
  const ssize_t assignment[10] = {-1, 2, 1, 0, 6, 1, 2, -1, -1, 5};
  for (int i = 0; i < 7; i++) g.create_group();
  g.delete_group(3);
  for (size_t i = 0; i < 10; i++)
    if (assignment[i] != -1) g.add_value(size_t(assignment[i]), i)++;
  CHECK(g.ngroups() == 6 && g.nentities() == 10);
  CHECK(g.groupsize(1) == 2 && g.groupsize(2) == 2 && g.groupsize(4) == 0 && g.groupsize(6) == 1);
  CHECK(g.empty_groups().size() == 1 && g.empty_groups().count(4) == 1);
  CHECK(g.pseudocount(1, g.group(1)) == 2.f && g.pseudocount(4, g.group(4)) == 2.f);   // alpha / 1 empty group
  {  // the whole partition at once (what a batched sweep hands back): sizes and the empty set follow
    gm h(10);
    for (int i = 0; i < 7; i++) h.create_group();
    h.delete_group(3);
    const std::vector<ssize_t> a = {4, 4, 4, 0, 6, 6, -1, 0, 4, 4};
    h.reassign_all(a);
    CHECK(h.assignments() == a && h.groupsize(4) == 5 && h.groupsize(0) == 2 && h.groupsize(6) == 2 && h.groupsize(1) == 0);
    CHECK(h.empty_groups() == (std::set<size_t>{1, 2, 5}));
    bool bad = false;
    try { h.reassign_all({3, 0, 0, 0, 0, 0, 0, 0, 0, 0}); } catch (const std::runtime_error &) { bad = true; }
    CHECK(bad && h.groupsize(4) == 5);                    // gid 3 was deleted; nothing changed
  }
  bool threw = false;
  try { g.delete_group(1); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  const auto blob = g.serialize([](size_t v) { return std::to_string(v); });
  gm g1(blob, [](const std::string &s) { return size_t(std::strtoul(s.c_str(), nullptr, 10)); });
  CHECK(g.get_hp_mutator("alpha").accessor().get<float>(0) == g1.get_hp_mutator("alpha").accessor().get<float>(0));   // (a float field of the message: bit for bit; test_group_manager.cpp:22-66 asks for 1e-5)
  CHECK(g.assignments() == g1.assignments() && g.ngroups() == g1.ngroups());
  for (auto gid : g.groups()) CHECK(g.group(gid) == g1.group(gid));
  // remove / re-add moves the empty set
  auto rem = g.remove_value(9);   // entity 9 was the only member of group 5
  CHECK(rem.first == 5 && g.empty_groups().count(5) == 1 && g.assignments()[9] == -1);
  CHECK(std::fabs(g.pseudocount(5, g.group(5)) - 1.0f) < 1e-6f);   // alpha / 2 empty groups
  g.add_value(4, 9);
  CHECK(g.empty_groups().count(4) == 0);
  // sequential CRP probability
  gm h(4);
  h.get_hp_mutator("alpha").set<float>(1.5f, 0);
  for (int i = 0; i < 2; i++) h.create_group();
  h.add_value(0, 0); h.add_value(0, 1); h.add_value(1, 2); h.add_value(0, 3);
  const float want = std::log(1.f / 2.5f) + std::log(1.5f / 3.5f) + std::log(2.f / 4.5f);
  CHECK(std::fabs(h.score_assignment() - want) < 1e-6f);
}

static void test_sampling_helpers() {
  std::vector<float> s = {-1.f, -2.f, -0.5f, -30.f};
  util::scores_to_probs(s);
  float tot = 0;
  for (float p : s) tot += p;
  CHECK(std::fabs(tot - 1.f) < 1e-6f && s[2] > s[0] && s[0] > s[1] && s[3] < 1e-10f);
  rng_t rng(5);
  std::vector<int> hist(4, 0);
  for (int i = 0; i < 20000; i++) hist[util::sample_discrete(s, rng)]++;
  CHECK(std::fabs(hist[2] / 20000.0 - s[2]) < 0.02 && hist[3] == 0);
}

static void test_noop_model_and_wire() {
  microscopes::models::noop_model m;
  rng_t rng(1);
  auto h = m.create_hypers();
  auto g = h->create_group(rng);
  const bool v = true;
  CHECK(g->score_value(*h, value_accessor(&v), rng) == 0.f && g->score_data(*h, rng) == 0.f);
  CHECK(m.get_runtime_type() == runtime_type(TYPE_B));
  microscopes::wire::writer w;
  w.put_float_field(1, 2.5f);
  w.put_varint_field(2, 300);
  w.put_bytes_field(3, "abc");
  const auto fs = microscopes::wire::parse(w.str());
  CHECK(fs.size() == 3 && fs[0].f32 == 2.5f && fs[1].varint == 300 && fs[2].bytes == "abc");
}

// the checks of test/cxx/test_relation.cpp:19-70, 120-190: every slice shows exactly the present cells of that
// row / column / plane with their values and positions
static void check_2d(const std::vector<int32_t> &data, const std::vector<uint8_t> &mask, size_t n, size_t m,
                     const relation::dataview &d) {
  CHECK(d.dims() == 2 && d.shape()[0] == n && d.shape()[1] == m);
  for (size_t i = 0; i < n; i++) {
    std::vector<bool> seen(m, false);
    for (const auto &p : d.slice(0, i)) {
      CHECK(p.first.size() == 2 && p.first[0] == i && p.first[1] < m);
      CHECK(!mask[i * m + p.first[1]] && !seen[p.first[1]]);
      CHECK(p.second.get<int32_t>(0) == data[i * m + p.first[1]]);
      seen[p.first[1]] = true;
    }
    for (size_t j = 0; j < m; j++) CHECK(seen[j] == !mask[i * m + j]);
  }
  for (size_t j = 0; j < m; j++) {
    std::vector<bool> seen(n, false);
    for (const auto &p : d.slice(1, j)) {
      CHECK(p.first[1] == j && p.first[0] < n && !mask[p.first[0] * m + j] && !seen[p.first[0]]);
      CHECK(p.second.get<int32_t>(0) == data[p.first[0] * m + j]);
      seen[p.first[0]] = true;
    }
    for (size_t i = 0; i < n; i++) CHECK(seen[i] == !mask[i * m + j]);
  }
}

static void test_relation_dataviews() {
  using namespace relation;
  std::mt19937 r(7);
  const size_t n = 5, m = 7;
  std::vector<int32_t> data(n * m);
  std::vector<uint8_t> mask(n * m);
  for (size_t i = 0; i < n * m; i++) {
    data[i] = int32_t(r() % 100) + 1;
    mask[i] = (r() % 10) < 3;
  }
  mask[0] = 0;
  static_assert(sizeof(bool) == 1, "bool masks");
  row_major_dense_dataview dense(reinterpret_cast<const uint8_t *>(data.data()), reinterpret_cast<const bool *>(mask.data()),
                                 {n, m}, runtime_type(TYPE_I32));
  check_2d(data, mask, n, m, dense);
  CHECK(dense.get({0, 0}).get<int32_t>(0) == data[0] && !dense.get({0, 0}).anymasked());
  bool threw = false;
  try { dense.get({n, 0}); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  // everything masked but one cell (test_relation.cpp:99-117)
  std::vector<uint8_t> one(n * m, 1);
  one[0] = 0;
  row_major_dense_dataview lonely(reinterpret_cast<const uint8_t *>(data.data()), reinterpret_cast<const bool *>(one.data()),
                                  {n, m}, runtime_type(TYPE_I32));
  check_2d(data, one, n, m, lonely);
  // the same matrix as csr + csc: absent entries are missing, not zero
  std::vector<int32_t> csr_d, csc_d;
  std::vector<uint32_t> csr_i, csr_p(1, 0), csc_i, csc_p(1, 0);
  for (size_t i = 0; i < n; i++) {
    for (size_t j = 0; j < m; j++)
      if (!mask[i * m + j]) { csr_d.push_back(data[i * m + j]); csr_i.push_back(uint32_t(j)); }
    csr_p.push_back(uint32_t(csr_i.size()));
  }
  for (size_t j = 0; j < m; j++) {
    for (size_t i = 0; i < n; i++)
      if (!mask[i * m + j]) { csc_d.push_back(data[i * m + j]); csc_i.push_back(uint32_t(i)); }
    csc_p.push_back(uint32_t(csc_i.size()));
  }
  compressed_2darray sparse(reinterpret_cast<const uint8_t *>(csr_d.data()), csr_i.data(), csr_p.data(),
                            reinterpret_cast<const uint8_t *>(csc_d.data()), csc_i.data(), csc_p.data(), n, m,
                            runtime_type(TYPE_I32));
  check_2d(data, mask, n, m, sparse);
  CHECK(sparse.nnz() == csr_d.size());
  // three dimensions: a slice is a plane, positions enumerate the other two indices
  const size_t a = 3, b = 4, c = 2;
  std::vector<float> cube(a * b * c);
  for (size_t i = 0; i < cube.size(); i++) cube[i] = float(i);
  row_major_dense_dataview d3(reinterpret_cast<const uint8_t *>(cube.data()), nullptr, {a, b, c}, runtime_type(TYPE_F32));
  for (size_t dim = 0; dim < 3; dim++)
    for (size_t idx = 0; idx < d3.shape()[dim]; idx++) {
      size_t count = 0;
      for (const auto &p : d3.slice(dim, idx)) {
        CHECK(p.first[dim] == idx);
        CHECK(p.second.get<float>(0) == cube[(p.first[0] * b + p.first[1]) * c + p.first[2]]);
        count++;
      }
      CHECK(count == a * b * c / d3.shape()[dim]);
    }
  threw = false;
  try { row_major_dense_dataview bad(reinterpret_cast<const uint8_t *>(cube.data()), nullptr, {}, runtime_type(TYPE_F32)); }
  catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
}

int main() {
  test_relation_dataviews();
  test_types_and_offsets();
  test_row_accessor_over_reference_rows();
  test_group_manager_bookkeeping_and_serialization();
  test_simple_group_manager();
  test_sampling_helpers();
  test_noop_model_and_wire();
  std::puts("test_host_api ok");
  return 0;
}

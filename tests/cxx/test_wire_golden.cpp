// The bytes the C++ plugin layer writes for the in-tree protobuf messages (microscopes/io/schema.proto:3-46), printed
// as "message<TAB>index<TAB>hex" for tests/test_cxx.py to hold against tests/golden/wire.json (Google's protobuf
// runtime): CRP (group_manager::get_hp), BetaBernoulliNonConj / DirichletMultinomial Shared + Group (hypers::get_hp,
// group::get_ss of bbnc_model / dm_model), GroupManager (group_manager::serialize on the scenario of the reference's
// test/cxx/test_group_manager.cpp:22-66).  Host-side only: no device call.
#include <cstdio>
#include <string>
#include <vector>

#include <microscopes/common/group_manager.hpp>
#include <microscopes/models/bbnc.hpp>
#include <microscopes/models/dm.hpp>

using namespace microscopes;
using namespace microscopes::common;

static void emit(const char *name, int idx, const std::string &bytes) {
  std::printf("%s\t%d\t", name, idx);
  for (unsigned char c : bytes) std::printf("%02x", c);
  std::printf("\n");
}

int main() {
  rng_t rng(1);
  {  // CRP: alpha 2.0, 0.5, 1.5
    const float alphas[] = {2.0f, 0.5f, 1.5f};
    for (int i = 0; i < 3; i++) {
      group_manager<size_t> g(4);
      g.get_hp_mutator("alpha").set<float>(alphas[i], 0);
      emit("CRP", i, g.get_hp());
    }
  }
  {  // bbnc Shared (alpha, beta) and Group (p, heads, tails)
    const float sh[][2] = {{1.0f, 1.0f}, {1.5f, 0.25f}, {2.0f, 7.5f}};
    struct { float p; uint32_t h, t; } gr[] = {{0.25f, 3, 300}, {0.5f, 0, 0}, {0.8125f, 70000, 1}};
    models::bbnc_model m;
    for (int i = 0; i < 3; i++) {
      auto h = m.create_hypers();
      h->get_hp_mutator("alpha").set<float>(sh[i][0], 0);
      h->get_hp_mutator("beta").set<float>(sh[i][1], 0);
      emit("BetaBernoulliNonConj.Shared", i, h->get_hp());
      auto g = h->create_group(rng);
      auto *bg = static_cast<models::bbnc_group *>(g.get());
      bg->repr_.p = gr[i].p;
      bg->repr_.heads = gr[i].h;
      bg->repr_.tails = gr[i].t;
      emit("BetaBernoulliNonConj.Group", i, g->get_ss());
      // and back: a fresh group loaded from the bytes writes the same bytes
      auto g2 = h->create_group(rng);
      g2->set_ss(g->get_ss());
      if (g2->get_ss() != g->get_ss()) return 1;
    }
  }
  {  // dm Shared (alphas) and Group (counts, ratio)
    const std::vector<std::vector<float>> sh = {{1.0f, 2.0f}, {0.5f, 1.5f, 2.5f, 0.125f}, {1.f, 1.f, 1.f, 1.f, 1.f}};
    const std::vector<std::vector<uint32_t>> cn = {{1, 128}, {0, 0, 0}, {5, 300, 70000, 2}};
    const float ratio[] = {1.0f, 0.0f, 12.75f};   // (a log multinomial coefficient: never negative, dm.hpp:49)
    for (int i = 0; i < 3; i++) {
      models::dm_model ms(unsigned(sh[i].size()));
      auto h = ms.create_hypers();
      for (size_t j = 0; j < sh[i].size(); j++) h->get_hp_mutator("alphas").set<float>(sh[i][j], j);
      emit("DirichletMultinomial.Shared", i, h->get_hp());
      models::dm_model mg(unsigned(cn[i].size()));
      auto hg = mg.create_hypers();
      auto g = hg->create_group(rng);
      auto *dg = static_cast<models::dm_group *>(g.get());
      dg->repr_.counts = cn[i];
      dg->repr_.ratio = ratio[i];
      emit("DirichletMultinomial.Group", i, g->get_ss());
      auto g2 = hg->create_group(rng);
      g2->set_ss(g->get_ss());
      if (g2->get_ss() != g->get_ss()) return 1;
    }
  }
  {  // GroupManager: test/cxx/test_group_manager.cpp:22-66
    group_manager<size_t> g(10);
    g.get_hp_mutator("alpha").set<float>(2.0, 0);
    const std::vector<ssize_t> assignment_vec({-1, 2, 1, 0, 6, 1, 2, -1, -1, 5});
    for (size_t i = 0; i < 7; i++) g.create_group();
    g.delete_group(3);
    for (size_t i = 0; i < assignment_vec.size(); i++) {
      if (assignment_vec[i] == -1) continue;
      g.add_value(assignment_vec[i], i)++;
    }
    const auto blob = g.serialize([](size_t i) { return std::to_string(i); });
    emit("GroupManager", 0, blob);
    group_manager<size_t> g1(blob, [](const std::string &s) { return size_t(std::strtoul(s.c_str(), nullptr, 10)); });
    if (g1.serialize([](size_t i) { return std::to_string(i); }) != blob) return 1;
  }
  return 0;
}

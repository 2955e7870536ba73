// perf_group_cpu.cpp -- CPU baseline driver (TEST/BENCH INFRASTRUCTURE ONLY).
//
// Times the oracle's float restatement (the reference's own precision) through a
// virtual model/hypers/group API of the reference's shape
// (include/microscopes/models/base.hpp:21-62) with the loop shape of
// bin/perf_group.cpp:76-125 generalised from (1 row x D features) to
// (N rows x K groups x D features):
//   perf_group pass: per row, per group: add_value over features, remove_value
//                    over features, score += score_value over features
//   score pass:      per row, per group: score += score_value over features
//   noop pass:       the perf_group pass through noop groups (API overhead control,
//                    models/noop.hpp:13-53)
// The real reference cannot be built here, so this is kind "port".
// Prints one JSON line.  Usage:
//   perf_group_cpu <config> <N> <K> <threads> [seed] [dump-file]
//     config: c1 (bb x8) | c2 (nich x1) | c3 (bb,gp,dd32,nich x16) | nich | bb | gp | dd
//     dump-file: the generated inputs are written there (int32 z[N], then every feature's column, raw) so that a
//     test can score the same data through another entry of the oracle; "score_sum" in the JSON is the double sum
//     of every float score of the single-thread score pass
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "msc_oracle.h"

namespace {

struct value_ref { const void *p; };   // stands in for common::value_accessor

class hypers;
class group {
public:
  virtual ~group() {}
  virtual void add_value(const hypers &h, const value_ref &v) = 0;
  virtual void remove_value(const hypers &h, const value_ref &v) = 0;
  virtual float score_value(const hypers &h, const value_ref &v) const = 0;
};
class hypers {
public:
  virtual ~hypers() {}
  virtual std::shared_ptr<group> create_group() const = 0;
};

class noop_group : public group {
public:
  void add_value(const hypers &, const value_ref &) override {}
  void remove_value(const hypers &, const value_ref &) override {}
  float score_value(const hypers &, const value_ref &) const override { return 0.f; }
};
class noop_hypers : public hypers {
public:
  std::shared_ptr<group> create_group() const override { return std::make_shared<noop_group>(); }
};

class orc_hypers : public hypers {
public:
  orc_hypers(int family, unsigned dim, std::vector<float> hp) : family_(family), dim_(dim), hp_(std::move(hp)) {}
  std::shared_ptr<group> create_group() const override;
  int family_;
  unsigned dim_;
  std::vector<float> hp_;
};
class orc_group : public group {
public:
  explicit orc_group(const orc_hypers &h) : ss_(orc_f32_ss_size(h.family_, h.dim_)) {
    orc_f32_init(h.family_, h.dim_, h.hp_.data(), ss_.data());
  }
  void add_value(const hypers &m, const value_ref &v) override {
    const orc_hypers &h = static_cast<const orc_hypers &>(m);   // unchecked downcast, distributions.hpp:523-528
    orc_f32_add_value(h.family_, h.dim_, h.hp_.data(), ss_.data(), v.p);
  }
  void remove_value(const hypers &m, const value_ref &v) override {
    const orc_hypers &h = static_cast<const orc_hypers &>(m);
    orc_f32_remove_value(h.family_, h.dim_, h.hp_.data(), ss_.data(), v.p);
  }
  float score_value(const hypers &m, const value_ref &v) const override {
    const orc_hypers &h = static_cast<const orc_hypers &>(m);
    return orc_f32_score_value(h.family_, h.dim_, h.hp_.data(), ss_.data(), v.p);
  }
  std::vector<uint8_t> ss_;
};
std::shared_ptr<group> orc_hypers::create_group() const { return std::make_shared<orc_group>(*this); }

struct feature {
  int family; unsigned dim; size_t vsize;
  std::vector<uint8_t> column;              // N values
  std::shared_ptr<hypers> hp;
  std::vector<std::shared_ptr<group>> groups;  // K
};

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <config> <N> <K> <threads> [seed]\n", argv[0]);
    return 2;
  }
  const std::string config = argv[1];
  const size_t N = std::strtoull(argv[2], nullptr, 10), K = std::strtoull(argv[3], nullptr, 10);
  const unsigned threads = (unsigned)std::strtoul(argv[4], nullptr, 10);
  const uint64_t seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 73;   // perf_group.cpp:19
  const char *dump = argc > 6 ? argv[6] : nullptr;

  std::vector<std::pair<int, unsigned>> spec;
  if (config == "c1") spec.assign(8, {ORC_BB, 0});
  else if (config == "c2" || config == "nich") spec.assign(1, {ORC_NICH, 0});
  else if (config == "c3") for (int i = 0; i < 16; i++) { spec.push_back({ORC_BB, 0}); spec.push_back({ORC_GP, 0}); spec.push_back({ORC_DD, 32}); spec.push_back({ORC_NICH, 0}); }
  else if (config == "bb") spec.assign(1, {ORC_BB, 0});
  else if (config == "gp") spec.assign(1, {ORC_GP, 0});
  else if (config == "dd") spec.assign(1, {ORC_DD, 32});
  else { std::fprintf(stderr, "unknown config %s\n", config.c_str()); return 2; }
  const size_t D = spec.size();

  std::mt19937_64 rng(seed);
  std::vector<int32_t> z(N);
  for (auto &g : z) g = (int32_t)(rng() % K);
  std::vector<feature> feats(D);
  std::vector<std::shared_ptr<hypers>> noop_hp(D);
  std::vector<std::vector<std::shared_ptr<group>>> noop_groups(D);
  for (size_t f = 0; f < D; f++) {
    feature &ft = feats[f];
    ft.family = spec[f].first; ft.dim = spec[f].second;
    ft.vsize = orc_value_size(ft.family, ft.dim);
    ft.column.resize(N * ft.vsize);
    std::vector<float> hp;
    switch (ft.family) {   // default hyper-parameters, microscopes/models.pyx:189,211,223,238
      case ORC_BB: hp = {config == "c1" ? 2.f : 1.f, config == "c1" ? 2.f : 1.f}; break;   // perf_group.cpp:43-44
      case ORC_GP: hp = {1.f, 1.f}; break;
      case ORC_DD: hp.assign(ft.dim, 1.f); break;
      default: hp = {0.f, 1.f, 1.f, 1.f}; break;
    }
    std::vector<double> par(K);
    for (auto &p : par) {
      if (ft.family == ORC_BB) p = std::uniform_real_distribution<double>(0, 1)(rng);
      else if (ft.family == ORC_GP) p = std::gamma_distribution<double>(2.0, 2.0)(rng);
      else p = std::normal_distribution<double>(0, 10)(rng);
    }
    for (size_t n = 0; n < N; n++) {
      void *dst = &ft.column[n * ft.vsize];
      const double p = par[z[n]];
      if (ft.family == ORC_BB) { uint8_t v = std::bernoulli_distribution(p)(rng); std::memcpy(dst, &v, 1); }
      else if (ft.family == ORC_GP) { uint32_t v = (uint32_t)std::poisson_distribution<int>(p)(rng); std::memcpy(dst, &v, 4); }
      else if (ft.family == ORC_DD) { int32_t v = (int32_t)((rng() + (uint64_t)z[n]) % ft.dim); std::memcpy(dst, &v, 4); }
      else { float v = (float)(p + std::normal_distribution<double>(0, 1)(rng)); std::memcpy(dst, &v, 4); }
    }
    ft.hp = std::make_shared<orc_hypers>(ft.family, ft.dim, hp);
    noop_hp[f] = std::make_shared<noop_hypers>();
    for (size_t k = 0; k < K; k++) {
      ft.groups.push_back(ft.hp->create_group());
      noop_groups[f].push_back(noop_hp[f]->create_group());
    }
    // suff-stats from the true assignment
    for (size_t n = 0; n < N; n++) ft.groups[z[n]]->add_value(*ft.hp, value_ref{&ft.column[n * ft.vsize]});
  }

  if (dump) {
    FILE *fh = std::fopen(dump, "wb");
    if (!fh) { std::fprintf(stderr, "cannot write %s\n", dump); return 2; }
    std::fwrite(z.data(), sizeof(int32_t), N, fh);
    for (size_t f = 0; f < D; f++) std::fwrite(feats[f].column.data(), 1, feats[f].column.size(), fh);
    std::fclose(fh);
  }

  const double evals = (double)N * (double)K * (double)D;
  float sink = 0.f;

  // noop pass (API overhead), single thread
  double t0 = now();
  for (size_t n = 0; n < N; n++)
    for (size_t k = 0; k < K; k++) {
      for (size_t f = 0; f < D; f++) noop_groups[f][k]->add_value(*noop_hp[f], value_ref{&feats[f].column[n * feats[f].vsize]});
      for (size_t f = 0; f < D; f++) noop_groups[f][k]->remove_value(*noop_hp[f], value_ref{&feats[f].column[n * feats[f].vsize]});
      for (size_t f = 0; f < D; f++) sink += noop_groups[f][k]->score_value(*noop_hp[f], value_ref{&feats[f].column[n * feats[f].vsize]});
    }
  const double t_noop = now() - t0;

  // perf_group pass, single thread (mutates and restores the groups, so it cannot be row-parallel)
  t0 = now();
  for (size_t n = 0; n < N; n++)
    for (size_t k = 0; k < K; k++) {
      for (size_t f = 0; f < D; f++) feats[f].groups[k]->add_value(*feats[f].hp, value_ref{&feats[f].column[n * feats[f].vsize]});
      for (size_t f = 0; f < D; f++) feats[f].groups[k]->remove_value(*feats[f].hp, value_ref{&feats[f].column[n * feats[f].vsize]});
      for (size_t f = 0; f < D; f++) sink += feats[f].groups[k]->score_value(*feats[f].hp, value_ref{&feats[f].column[n * feats[f].vsize]});
    }
  const double t_pg = now() - t0;

  // score pass: 1 thread, then `threads` threads over row blocks
  double score_sum = 0.0;                 // of the single-thread pass (a test compares it with the batch oracle)
  auto score_rows = [&](size_t lo, size_t hi, float *out, double *exact) {
    float s = 0.f;
    double e = 0.0;
    for (size_t n = lo; n < hi; n++)
      for (size_t k = 0; k < K; k++)
        for (size_t f = 0; f < D; f++) {
          const float v = feats[f].groups[k]->score_value(*feats[f].hp, value_ref{&feats[f].column[n * feats[f].vsize]});
          s += v;
          e += (double)v;
        }
    *out = s;
    if (exact) *exact = e;
  };
  t0 = now();
  float s1 = 0.f;
  score_rows(0, N, &s1, &score_sum);
  const double t_s1 = now() - t0;
  sink += s1;

  double t_sn = 0;
  if (threads > 1) {
    std::vector<std::thread> pool;
    std::vector<float> part(threads, 0.f);
    t0 = now();
    for (unsigned t = 0; t < threads; t++)
      pool.emplace_back(score_rows, N * t / threads, N * (t + 1) / threads, &part[t], (double *)nullptr);
    for (auto &th : pool) th.join();
    t_sn = now() - t0;
    for (float p : part) sink += p;
  }

  std::printf("{\"config\": \"%s\", \"N\": %zu, \"K\": %zu, \"D\": %zu, \"evals\": %.0f, "
              "\"noop_s\": %.6f, \"perf_group_s\": %.6f, \"score_1core_s\": %.6f, "
              "\"score_ncore_s\": %.6f, \"threads\": %u, \"perf_group_evals_per_s\": %.6g, "
              "\"score_evals_per_s_1core\": %.6g, \"score_evals_per_s_ncore\": %.6g, \"score_sum\": %.17g, "
              "\"ignore\": %g}\n",
              config.c_str(), N, K, D, evals, t_noop, t_pg, t_s1, t_sn, threads, evals / t_pg,
              evals / t_s1, threads > 1 ? evals / t_sn : 0.0, score_sum, (double)sink);
  return 0;
}

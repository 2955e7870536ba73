// perf_group_hip.cpp -- the in-tree microbenchmark of the scoring hot path, for the HIP
// backend.  Same measurement as the reference's bin/perf_group.cpp (1 row x D boolean
// features, per iteration add_value / remove_value / score_value over every feature's
// group, seed 73, alpha = beta = 2), run four ways:
//   1. noop groups through the virtual API            (API overhead control)
//   2. Beta-Bernoulli groups through the virtual API  (each call = one device launch: latency)
//   3. the same iteration as three batched C-ABI calls over a D-feature state
//   4. a batched scoring pass at a realistic shape (N rows x K groups x D features)
// Prints sec/iter like the reference, plus evals/s for the batched runs.
#include <microscopes/common/recarray/dataview.hpp>
#include <microscopes/common/timer.hpp>
#include <microscopes/models/distributions.hpp>
#include <microscopes/models/noop.hpp>
#include <microscopes_hip.h>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <vector>

using namespace distributions;
using namespace microscopes;
using namespace microscopes::common;
using namespace microscopes::common::recarray;

static void ok(int rc) {
  if (rc != MSC_OK) {
    std::fprintf(stderr, "microscopes_hip: %s\n", msc_last_error());
    std::exit(1);
  }
}

int main(int argc, char **argv) {
  const size_t D = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1000;
  const size_t api_iters = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 20;
  rng_t r(73);

  std::vector<uint8_t> data(D);
  for (auto &b : data) b = std::bernoulli_distribution(0.5)(r);
  std::vector<runtime_type> types(D, runtime_type(TYPE_B));
  row_accessor acc(data.data(), nullptr, &types);

  std::vector<std::shared_ptr<models::hypers>> noop_shares, shares;
  std::vector<std::shared_ptr<models::group>> noop_groups, groups;
  for (size_t i = 0; i < D; i++) {
    noop_shares.emplace_back(models::noop_model().create_hypers());
    noop_groups.emplace_back(noop_shares.back()->create_group(r));
    shares.emplace_back(models::distributions_model<BetaBernoulli>().create_hypers());
    shares.back()->get_hp_mutator("alpha").set<float>(2.0);
    shares.back()->get_hp_mutator("beta").set<float>(2.0);
    groups.emplace_back(shares.back()->create_group(r));
  }

  auto one_iter = [&](std::vector<std::shared_ptr<models::group>> &gs,
                      std::vector<std::shared_ptr<models::hypers>> &hs) {
    float score = 0.f;
    acc.reset();
    for (size_t i = 0; i < acc.nfeatures(); i++, acc.bump()) gs[i]->add_value(*hs[i], acc.get(), r);
    acc.reset();
    for (size_t i = 0; i < acc.nfeatures(); i++, acc.bump()) gs[i]->remove_value(*hs[i], acc.get(), r);
    acc.reset();
    for (size_t i = 0; i < acc.nfeatures(); i++, acc.bump()) score += gs[i]->score_value(*hs[i], acc.get(), r);
    return score;
  };

  float sink = 0.f;
  {
    const size_t niters = 100000;
    timer tt;
    for (size_t n = 0; n < niters; n++) sink += one_iter(noop_groups, noop_shares);
    std::cout << "noop virtual API      sec/iter: " << (tt.lap_ms() / 1000.0 / double(niters)) << std::endl;
  }
  {
    one_iter(groups, shares);   // creates the device context
    timer tt;
    for (size_t n = 0; n < api_iters; n++) sink += one_iter(groups, shares);
    const double s = tt.lap_ms() / 1000.0 / double(api_iters);
    std::cout << "bb virtual API (HIP)  sec/iter: " << s << "   (" << s / double(3 * D) * 1e6
              << " us per call, one launch each)" << std::endl;
  }

  // 3. the same iteration, batched: D features, 1 group, 1 row
  msc_context *ctx = hip::default_context();
  {
    std::vector<msc_feature_spec> spec(D, msc_feature_spec{MSC_BB, 0});
    msc_state *st = nullptr;
    ok(msc_state_create(ctx, spec.data(), uint32_t(D), 1, &st));
    const float hp[2] = {2.f, 2.f};
    for (size_t f = 0; f < D; f++) ok(msc_state_set_hp(st, uint32_t(f), hp, 2));
    row_major_dataview hostview(data.data(), nullptr, 1, types);
    msc_dataview *view = hostview.to_device(ctx);
    int32_t *z = nullptr;
    float *out = nullptr;
    hipMalloc(reinterpret_cast<void **>(&z), 4);
    hipMemset(z, 0xff, 4);                                   // (row 0 starts unassigned: -1)
    hipMalloc(reinterpret_cast<void **>(&out), 4);
    const size_t niters = 2000;
    // the iteration as downstream code would write it against the batched entry points: the row joins the group, leaves
    // it again, and is scored -- msc_entity_op x 2 + msc_score_value = three launches for all D features, then the one
    // float comes back (the copy is the only wait)
    float score_host = 0.f;
    for (int warm = 0; warm < 3; warm++) {
      ok(msc_entity_op(st, view, nullptr, 0, 0, +1, z));
      ok(msc_entity_op(st, view, nullptr, 0, 0, -1, z));
      ok(msc_score_value(st, view, nullptr, 0, 1, nullptr, 0, out, 1));
      ok(msc_device_download(ctx, &score_host, out, 4));
    }
    timer tt;
    for (size_t n = 0; n < niters; n++) {
      ok(msc_entity_op(st, view, nullptr, 0, 0, +1, z));
      ok(msc_entity_op(st, view, nullptr, 0, 0, -1, z));
      ok(msc_score_value(st, view, nullptr, 0, 1, nullptr, 0, out, 1));
      ok(msc_device_download(ctx, &score_host, out, 4));
      sink += score_host;
    }
    std::cout << "bb batched C ABI      sec/iter: " << (tt.lap_ms() / 1000.0 / double(niters))
              << "   (add + remove + score of all " << D << " features per iteration: 3 launches + the score copied back)" << std::endl;
    float s = 0.f;
    hipMemcpy(&s, out, 4, hipMemcpyDeviceToHost);
    sink += s;
    msc_dataview_destroy(view);
    msc_state_destroy(st);
    hipFree(z);
    hipFree(out);
  }

  // 4. realistic batched shape: config C1 of BASELINE.json (N = 10k, K = 16, D = 8 bb)
  {
    const size_t N = 10000, K = 16, D1 = 8;
    std::vector<uint8_t> rows(N * D1);
    for (auto &b : rows) b = std::bernoulli_distribution(0.5)(r);
    std::vector<runtime_type> t8(D1, runtime_type(TYPE_B));
    row_major_dataview hostview(rows.data(), nullptr, N, t8);
    msc_dataview *view = hostview.to_device(ctx);
    std::vector<msc_feature_spec> spec(D1, msc_feature_spec{MSC_BB, 0});
    msc_state *st = nullptr;
    ok(msc_state_create(ctx, spec.data(), uint32_t(D1), uint32_t(K), &st));
    const float hp[2] = {2.f, 2.f};
    for (size_t f = 0; f < D1; f++) ok(msc_state_set_hp(st, uint32_t(f), hp, 2));
    std::vector<int32_t> zh(N);
    for (auto &g : zh) g = int32_t(std::uniform_int_distribution<int>(0, int(K) - 1)(r));
    int32_t *z = nullptr;
    float *out = nullptr;
    hipMalloc(reinterpret_cast<void **>(&z), 4 * N);
    hipMemcpy(z, zh.data(), 4 * N, hipMemcpyHostToDevice);
    hipMalloc(reinterpret_cast<void **>(&out), 4 * N * K);
    ok(msc_accumulate(st, view, nullptr, 0, N, z, MSC_ACC_RESET));
    for (int warm = 0; warm < 3; warm++) ok(msc_score_value(st, view, nullptr, 0, N, nullptr, 0, out, K));
    ok(msc_context_synchronize(ctx));
    const size_t niters = 200;
    timer tt;
    for (size_t n = 0; n < niters; n++) ok(msc_score_value(st, view, nullptr, 0, N, nullptr, 0, out, K));
    ok(msc_context_synchronize(ctx));
    const double s = tt.lap_ms() / 1000.0 / double(niters);
    std::cout << "C1 scoring pass (N=10k,K=16,D=8 bb) sec/pass: " << s << "   "
              << double(N * K * D1) / s << " score_value evals/s" << std::endl;
    msc_dataview_destroy(view);
    msc_state_destroy(st);
    hipFree(z);
    hipFree(out);
  }
  std::cout << "ignore: " << sink << std::endl;
  return 0;
}

// The C-level collective (microscopes_hip.h msc_comm_*): a communicator of one rank on the box's one GPU -- the id, the
// RCCL communicator, ncclAllReduce of both additive tables on the context's stream -- and the sharded step built on it
// against the single-process step.  (Ranks > 1 need one GPU each; the Python driver's 2-rank test covers the exchange
// semantics over gloo, this covers the RCCL plumbing.)
#include <microscopes_hip.h>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d) %s\n", #cond, __FILE__, __LINE__, msc_last_error()); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

int main() {
  msc_context *ctx = nullptr;
  CHECK(msc_context_create(0, nullptr, &ctx) == MSC_OK);
  std::vector<char> id(msc_comm_unique_id_bytes());
  CHECK(id.size() == 128);
  CHECK(msc_comm_unique_id(id.data(), id.size()) == MSC_OK);
  msc_comm *comm = nullptr;
  CHECK(msc_comm_create(ctx, id.data(), id.size(), 1, 0, &comm) == MSC_OK);
  int nr = 0, rk = -1;
  CHECK(msc_comm_size(comm, &nr, &rk) == MSC_OK && nr == 1 && rk == 0);
  CHECK(msc_comm_create(ctx, id.data(), 5, 1, 0, &comm) == MSC_EINVAL);          // a short id is refused

  const uint64_t N = 50000;
  const uint32_t K = 40;
  std::mt19937 gen(3);
  std::vector<float> x(N);
  std::vector<int32_t> z(N);
  for (uint64_t i = 0; i < N; i++) {
    z[i] = int32_t(gen() % K);
    x[i] = float(std::normal_distribution<double>(3.0 * (z[i] % 7), 1.0)(gen));
  }
  const msc_runtime_type ty = {MSC_TYPE_F32, 1};
  msc_dataview *view = nullptr;
  CHECK(msc_dataview_from_records(ctx, x.data(), nullptr, N, &ty, 1, nullptr, &view) == MSC_OK);
  const msc_feature_spec spec = {MSC_NICH, 0};
  msc_state *a = nullptr, *b = nullptr;
  CHECK(msc_state_create(ctx, &spec, 1, K, &a) == MSC_OK && msc_state_create(ctx, &spec, 1, K, &b) == MSC_OK);
  CHECK(msc_state_set_alpha(a, 1.5f) == MSC_OK && msc_state_set_alpha(b, 1.5f) == MSC_OK);
  int32_t *za = nullptr, *zb = nullptr;
  CHECK(msc_device_alloc(ctx, 4 * N, reinterpret_cast<void **>(&za)) == MSC_OK);
  CHECK(msc_device_alloc(ctx, 4 * N, reinterpret_cast<void **>(&zb)) == MSC_OK);
  CHECK(msc_device_upload(ctx, za, z.data(), 4 * N) == MSC_OK && msc_device_upload(ctx, zb, z.data(), 4 * N) == MSC_OK);

  // a: the sharded entry points over the one-rank communicator, with the exchange spelled out once
  CHECK(msc_accumulate(a, view, nullptr, 0, N, za, MSC_ACC_RESET | MSC_ACC_NO_COMMIT) == MSC_OK);
  CHECK(msc_state_allreduce(a, comm) == MSC_OK);                                  // ncclAllReduce x 2 in one group
  CHECK(msc_state_commit_reduce(a) == MSC_OK);
  // b: the plain single-process calls
  CHECK(msc_accumulate(b, view, nullptr, 0, N, zb, MSC_ACC_RESET) == MSC_OK);
  for (uint64_t sweep = 0; sweep < 3; sweep++) {
    if (sweep == 1) {                                                             // once through begin / exchange / commit by hand
      CHECK(msc_sweep_step_begin(a, view, nullptr, 0, N, 0, za, 11, sweep) == MSC_OK);
      CHECK(msc_state_allreduce(a, comm) == MSC_OK);
      CHECK(msc_state_commit_reduce(a) == MSC_OK);
    } else {
      CHECK(msc_sweep_step_sharded(a, view, nullptr, 0, N, 0, za, 11, sweep, comm) == MSC_OK);
    }
    CHECK(msc_sweep_step(b, view, nullptr, 0, N, 0, zb, 11, sweep) == MSC_OK);
  }
  std::vector<int32_t> ha(N), hb(N);
  CHECK(msc_device_download(ctx, ha.data(), za, 4 * N) == MSC_OK && msc_device_download(ctx, hb.data(), zb, 4 * N) == MSC_OK);
  size_t moved = 0;
  for (uint64_t i = 0; i < N; i++) {
    CHECK(ha[i] == hb[i]);
    moved += ha[i] != z[i];
  }
  CHECK(moved > 0);
  std::vector<uint32_t> ca(K), cb(K);
  CHECK(msc_state_get_group_counts(a, ca.data(), K) == MSC_OK && msc_state_get_group_counts(b, cb.data(), K) == MSC_OK);
  CHECK(ca == cb);
  CHECK(msc_accumulate_sharded(a, view, nullptr, 0, N, za, comm) == MSC_OK);      // rebuild from the final assignment
  CHECK(msc_state_get_group_counts(a, ca.data(), K) == MSC_OK && ca == cb);

  msc_state_destroy(a);
  msc_state_destroy(b);
  msc_dataview_destroy(view);
  msc_device_free(ctx, za);
  msc_device_free(ctx, zb);
  CHECK(msc_comm_destroy(comm) == MSC_OK);
  msc_context_destroy(ctx);
  std::printf("test_comm_gpu ok\n");
  return 0;
}

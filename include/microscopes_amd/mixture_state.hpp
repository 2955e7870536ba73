// mixture_state.hpp -- an entity_based_state_object (entity_state.hpp) whose component models live in
// the device tables of microscopes_hip.h: the state object a Dirichlet-process mixture model hands to the
// sampling kernels, with the same per-entity contract (add_value / remove_value / score_value /
// score_likelihood / create_group / delete_group, entity_state.hpp:57-89) plus the batched calls the
// device is built for (assign_all, gibbs_sweep).
//
// What is where:
//   host    the partition (group_manager: assignment vector, group sizes, the set of empty groups, alpha),
//           the hypers objects of the component models, the group id <-> device slot map
//   device  the data (columnar, uploaded once), the assignment vector as slots, every group's suff-stats
// A group id is what the caller sees (never reused, as group_manager hands them out); a slot is the group's
// column in the device tables (0 .. max_groups-1, reused after delete_group).
//
// Per-entity calls are the latency path -- each is a handful of launches plus, for scores, one copy back
// (~30-100 us) -- and exist so that kernels written against the reference's interface run unchanged.
// gibbs_sweep() is the throughput path: one synchronous sweep over all entities on the device
// (msc_sweep_step), after which the host partition is rebuilt from the new assignment vector.
#pragma once

#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <vector>

#include "../microscopes_hip.h"
#include "entity_state.hpp"
#include "group_manager.hpp"
#include "hip_models.hpp"
#include "recarray.hpp"

namespace microscopes {
namespace hip {

class mixture_state : public common::entity_based_state_object {
public:
  typedef common::entity_based_state_object::scores_t scores_t;

  // models[f] scores column f of `data`; max_groups = slots reserved in the device tables (groups alive at once)
  mixture_state(const std::vector<models::model_shared_ptr> &models, const common::recarray::row_major_dataview &data,
                size_t max_groups, msc_context *ctx = default_context())
      : ctx_(ctx), n_(data.size()), kmax_(max_groups), gm_(data.size()) {
    if (models.empty()) throw std::runtime_error("no component models");
    if (models.size() != data.types().size()) throw std::runtime_error("one model per column expected");
    if (!max_groups) throw std::runtime_error("max_groups must be positive");
    std::vector<msc_feature_spec> specs(models.size());
    std::vector<int32_t> col_types;
    for (size_t f = 0; f < models.size(); f++) {
      hypers_.push_back(models[f]->create_hypers());
      std::vector<float> hp;
      if (!hypers_[f]->device_spec(specs[f], hp)) throw std::runtime_error("component model has no device family");
      col_types.push_back(int32_t(models[f]->get_runtime_type().t()));     // convert at upload (runtime_type.hpp:153-161)
      hp_pushed_.push_back(std::vector<float>());
      nonconj_.push_back(specs[f].family == MSC_BBNC);   // the group carries a parameter of its own (bbnc.cpp:22-73)
      any_nonconj_ = any_nonconj_ || nonconj_.back();
    }
    view_ = data.to_device(ctx_, &col_types);
    check(msc_state_create(ctx_, specs.data(), uint32_t(specs.size()), uint32_t(kmax_), &st_));
    check(msc_device_alloc(ctx_, 4 * (n_ ? n_ : 1), reinterpret_cast<void **>(&z_dev_)));
    check(msc_pinned_alloc(ctx_, 4 * kmax_, reinterpret_cast<void **>(&row_host_), reinterpret_cast<void **>(&row_dev_)));   // the score row lands in host memory
    check(msc_device_alloc(ctx_, 4 * kmax_ * hypers_.size(), reinterpret_cast<void **>(&sd_dev_)));
    z_host_.assign(n_, -1);
    check(msc_device_upload(ctx_, z_dev_, z_host_.data(), 4 * n_));
    for (size_t s = kmax_; s-- > 0;) free_slots_.push_back(s);
  }
  ~mixture_state() override {
    if (st_) msc_state_destroy(st_);
    if (view_) msc_dataview_destroy(view_);
    msc_device_free(ctx_, z_dev_);
    msc_pinned_free(ctx_, row_host_);
    msc_device_free(ctx_, sd_dev_);
  }
  mixture_state(const mixture_state &) = delete;
  mixture_state &operator=(const mixture_state &) = delete;

  // ---- sizes and the partition ----
  size_t nentities() const override { return n_; }
  size_t ngroups() const override { sync(); return gm_.ngroups(); }
  size_t ncomponents() const override { return hypers_.size(); }
  std::vector<ssize_t> assignments() const override { sync(); return gm_.assignments(); }
  std::vector<size_t> groups() const override { sync(); return gm_.groups(); }
  size_t groupsize(size_t gid) const override { sync(); return gm_.groupsize(gid); }
  std::vector<size_t> empty_groups() const override {
    sync();
    return std::vector<size_t>(gm_.empty_groups().begin(), gm_.empty_groups().end());
  }

  // ---- parameters ----
  common::hyperparam_bag_t get_cluster_hp() const override { return gm_.get_hp(); }
  void set_cluster_hp(const common::hyperparam_bag_t &hp) override { gm_.set_hp(hp); }
  common::value_mutator get_cluster_hp_mutator(const std::string &key) override { return gm_.get_hp_mutator(key); }
  common::hyperparam_bag_t get_component_hp(size_t c) const override { return hypers_.at(c)->get_hp(); }
  void set_component_hp(size_t c, const common::hyperparam_bag_t &hp) override { hypers_.at(c)->set_hp(hp); }
  void set_component_hp(size_t c, const models::hypers &proto) override { hypers_.at(c)->set_hp(proto); }
  common::value_mutator get_component_hp_mutator(size_t c, const std::string &key) override {
    return hypers_.at(c)->get_hp_mutator(key);       // (written through a raw pointer: pushed to the device before the next device call)
  }

  // ---- sufficient statistics (identifier = group id) ----
  std::vector<common::ident_t> suffstats_identifiers(size_t) const override { sync(); return gm_.groups(); }
  common::suffstats_bag_t get_suffstats(size_t c, common::ident_t gid) const override { return fetch_group(c, gid)->get_ss(); }
  void set_suffstats(size_t c, common::ident_t gid, const common::suffstats_bag_t &ss) override {
    common::rng_t rng;
    auto g = hypers_.at(c)->create_group(rng);
    g->set_ss(ss);
    store_group(c, gid, *g);
  }
  // the device owns the numbers: a mutator would be a pointer into a copy (the reference's own dm model has no
  // mutators either, dm.cpp:120-124); read with get_suffstats, write with set_suffstats
  common::value_mutator get_suffstats_mutator(size_t, common::ident_t, const std::string &) override {
    throw std::runtime_error("suff-stats live on the device: use get_suffstats / set_suffstats");
  }

  // ---- membership ----
  void add_value(size_t gid, size_t eid, common::rng_t &) override {
    sync();
    const size_t slot = gm_.add_value(gid, eid);
    push_params();
    z_host_[eid] = int32_t(slot);
    // the group goes by value: one launch updates the sums, the group's fields and score constants, the counts and
    // the device's copy of the assignment -- nothing to upload, nothing to wait for
    check(msc_entity_op(st_, view_, nullptr, eid, uint32_t(slot), +1, z_dev_));
  }
  size_t remove_value(size_t eid, common::rng_t &) override {
    sync();
    const auto r = gm_.remove_value(eid);                 // (throws if the entity is not assigned); r = (gid, its slot)
    check(msc_entity_op(st_, view_, nullptr, eid, uint32_t(r.second), -1, z_dev_));
    z_host_[eid] = -1;
    return r.first;
  }

  // score[i] = log pseudocount(group i) + sum over components of score_value(group i, entity's value): the
  // unnormalised log-probability of the entity joining each group, empty groups included (alpha / n_empty each).
  // The likelihood terms come from one device pass over all slots; the prior is added here from the host partition.
  void inplace_score_value(scores_t &scores, size_t eid, common::rng_t &) const override {
    sync();
    if (gm_.assignments().at(eid) != -1) throw std::runtime_error("entity must be removed before it is scored");
    mixture_state *self = const_cast<mixture_state *>(this);
    self->push_params(true);
    check(msc_score_value(st_, view_, nullptr, eid, 1, nullptr, 0, row_dev_, kmax_));
    check(msc_context_synchronize(ctx_));                 // the only wait of a move; the row was written into pinned host memory
    const float *row = row_host_;
    scores.first.clear();
    scores.second.clear();
    for (auto it = gm_.begin(); it != gm_.end(); ++it) {
      scores.first.push_back(it->first);
      scores.second.push_back(std::log(gm_.pseudocount(it->first, it->second)) + row[it->second.data_]);
    }
  }

  float score_assignment() const override { sync(); return gm_.score_assignment(); }
  using common::entity_based_state_object::score_likelihood;
  float score_likelihood(size_t c, common::ident_t gid, common::rng_t &) const override {
    sync();
    if (c >= hypers_.size()) throw std::runtime_error("invalid component");
    const size_t slot = gm_.group(gid).data_;
    const_cast<mixture_state *>(this)->push_params();
    check(msc_score_data(st_, sd_dev_));
    float v = 0.f;
    check(msc_device_download(ctx_, &v, sd_dev_ + c * kmax_ + slot, 4));
    return v;
  }
  // every group of a component with one device pass and one copy (the per-id form above costs a pass each)
  float score_likelihood(size_t c, common::rng_t &) const override {
    sync();
    if (c >= hypers_.size()) throw std::runtime_error("invalid component");
    const_cast<mixture_state *>(this)->push_params();
    check(msc_score_data(st_, sd_dev_));
    std::vector<float> v(kmax_);
    check(msc_device_download(ctx_, v.data(), sd_dev_ + c * kmax_, 4 * kmax_));
    float total = 0.f;
    for (auto it = gm_.begin(); it != gm_.end(); ++it) total += v[it->second.data_];
    return total;
  }

  // ---- the supply of empty groups ----
  size_t create_group(common::rng_t &rng) override {
    sync();
    return create_group_unsynced(rng);
  }
  void delete_group(size_t gid) override {
    sync();
    const size_t slot = gm_.group(gid).data_;
    gm_.delete_group(gid);                                // (throws unless the group is empty: its device column is all zero then)
    free_slots_.push_back(slot);
  }

  // ---- batched calls (extension) ----
  // the whole partition at once: entity e joins group gids[e] (groups are created as needed); replaces
  // N add_value calls by one accumulate pass
  void assign_all(const std::vector<size_t> &labels, common::rng_t &rng) {
    sync();
    if (labels.size() != n_) throw std::runtime_error("one label per entity expected");
    for (size_t e = 0; e < n_; e++)
      if (gm_.assignments()[e] != -1) throw std::runtime_error("assign_all wants every entity unassigned");
    std::map<size_t, size_t> label_gid;
    for (size_t e = 0; e < n_; e++) {
      auto it = label_gid.find(labels[e]);
      if (it == label_gid.end()) it = label_gid.emplace(labels[e], create_group(rng)).first;
      z_host_[e] = int32_t(gm_.add_value(it->second, e));
    }
    push_params();
    check(msc_device_upload(ctx_, z_dev_, z_host_.data(), 4 * n_));
    check(msc_accumulate(st_, view_, nullptr, 0, n_, z_dev_, MSC_ACC_RESET));
  }

  // One synchronous Gibbs sweep over all entities on the device: every entity is scored leave-one-out against the
  // tables as they stand, with the CRP prior (every free slot is an empty group on offer and they share alpha,
  // which is the prior of "a new group" however many empty groups the host has created), re-drawn with the
  // counter-based uniform Philox(seed, sweep, entity), and the tables are rebuilt.  The host partition follows the new
  // assignment vector the next time something looks at it (sync(): slots that gained their first member get a group
  // id, groups that lost every member stay as empty groups -- delete_group them if unwanted), so a run of sweeps costs
  // the host nothing.
  void gibbs_sweep(uint64_t seed, uint64_t sweep, common::rng_t &rng) {
    if (any_nonconj_) {
      sync();                                           // (which slots are free is host knowledge)
      refresh_free_slots(rng);
    }
    if (!stale_)                                        // (after a sweep every entity is assigned)
      for (size_t e = 0; e < n_; e++)
        if (gm_.assignments()[e] == -1) throw std::runtime_error("gibbs_sweep wants every entity assigned");
    push_params(true);
    check(msc_sweep_step(st_, view_, nullptr, 0, n_, 0, z_dev_, seed, sweep));
    stale_ = true;      // the host partition follows when somebody looks at it: sweep after sweep costs the host nothing
  }

  // the device handles, for callers that mix in calls of microscopes_hip.h
  msc_state *device_state() const { return st_; }
  msc_dataview *device_view() const { return view_; }
  const int32_t *device_assignments() const { return z_dev_; }
  size_t slot_of(size_t gid) const { sync(); return gm_.group(gid).data_; }

private:
  // After batched sweeps the partition lives in z_dev_ alone; anything that reads or edits the host's view of it
  // calls this first: one download and an O(n + groups) rebuild (slots that gained their first member become
  // groups, groups that lost every member stay as empty groups).
  void sync() const {
    if (!stale_) return;
    mixture_state *self = const_cast<mixture_state *>(this);
    self->stale_ = false;
    check(msc_device_download(ctx_, self->z_host_.data(), z_dev_, 4 * n_));
    std::vector<ssize_t> slot_gid(kmax_, -1);
    for (auto it = gm_.begin(); it != gm_.end(); ++it) slot_gid[it->second.data_] = ssize_t(it->first);
    std::vector<ssize_t> a(n_);
    for (size_t e = 0; e < n_; e++) {
      const size_t slot = size_t(z_host_[e]);
      if (slot >= kmax_) throw std::runtime_error("device drew a slot outside the table");
      if (slot_gid[slot] < 0) {                          // a free slot was drawn: it becomes a group
        const auto pos = std::find(self->free_slots_.begin(), self->free_slots_.end(), slot);
        if (pos == self->free_slots_.end()) throw std::runtime_error("device drew a slot the host does not know");
        std::swap(*pos, self->free_slots_.back());
        common::rng_t rng;
        slot_gid[slot] = ssize_t(self->create_group_unsynced(rng, true));
      }
      a[e] = slot_gid[slot];
    }
    self->gm_.reassign_all(a);
  }
  // keep_record: the slot was on offer during a sweep and rows were scored against (and added to) what it holds --
  // its record already is the group's
  size_t create_group_unsynced(common::rng_t &rng, bool keep_record = false) {
    if (free_slots_.empty()) throw std::runtime_error("all max_groups device slots are in use");
    auto r = gm_.create_group();
    r.second = free_slots_.back();
    free_slots_.pop_back();
    if (!keep_record) init_slot(r.second, rng);
    return r.first;
  }
  // What mixturemodel's create_group does per component (hypers::create_group, base.hpp:48): the model's own initial
  // group goes into the slot.  For the conjugate families that record is all zero, which is what a slot holds once its
  // group has emptied (and what it is created with), so nothing is written; a non-conjugate model draws its
  // parameter here (bbnc: p ~ Beta(alpha, beta), bbnc.cpp:129-133) and the slot must not keep the previous occupant's.
  void init_slot(size_t slot, common::rng_t &rng) {
    for (size_t c = 0; c < hypers_.size(); c++) {
      if (!nonconj_[c]) continue;
      auto g = hypers_[c]->create_group(rng);
      std::vector<uint8_t> rec;
      g->device_record_get(*hypers_[c], rec);
      if (!rec.empty()) check(msc_state_set_ss(st_, uint32_t(c), uint32_t(slot), 1, rec.data(), rec.size()));
    }
  }
  // Every free slot is an empty group on offer in a sweep; for a non-conjugate component each gets a fresh draw of
  // its parameter first (one read-modify-write of the component's table, only for such components).
  void refresh_free_slots(common::rng_t &rng) {
    if (free_slots_.empty()) return;
    for (size_t c = 0; c < hypers_.size(); c++) {
      if (!nonconj_[c]) continue;
      msc_feature_spec spec;
      std::vector<float> hp;
      hypers_[c]->device_spec(spec, hp);
      const size_t rb = msc_ss_bytes(spec.family, spec.dim);
      std::vector<uint8_t> all(rb * kmax_), rec;
      check(msc_state_get_ss(st_, uint32_t(c), 0, uint32_t(kmax_), all.data(), all.size()));
      for (size_t slot : free_slots_) {
        auto g = hypers_[c]->create_group(rng);
        g->device_record_get(*hypers_[c], rec);
        if (rec.size() == rb) std::copy(rec.begin(), rec.end(), all.begin() + rb * slot);
      }
      check(msc_state_set_ss(st_, uint32_t(c), 0, uint32_t(kmax_), all.data(), all.size()));
    }
  }
  mutable bool stale_ = false;

  // hyper-parameters can change behind our back (mutators are raw pointers), so what the device holds is
  // compared with the hypers objects before every device call; a handful of floats per component
  void push_params(bool need_alpha = false) {
    for (size_t f = 0; f < hypers_.size(); f++) {
      msc_feature_spec spec;
      std::vector<float> hp;
      hypers_[f]->device_spec(spec, hp);
      if (hp != hp_pushed_[f]) {
        if (!hp.empty()) check(msc_state_set_hp(st_, uint32_t(f), hp.data(), hp.size()));
        hp_pushed_[f] = hp;
      }
    }
    const float alpha = gm_.get_hp_mutator("alpha").accessor().get<float>(0);
    // group_manager.hpp:78 asserts alpha > 0 where it is used; the device would otherwise keep its default of 1 and
    // draw under another concentration than the host's pseudocount() / score_assignment()
    if (!(alpha > 0.f)) {
      if (need_alpha) throw std::runtime_error("cluster hyper-parameter alpha must be set (> 0) before scoring or sweeping");
      return;                                           // (membership edits and likelihoods do not read it)
    }
    if (alpha != alpha_pushed_) {
      check(msc_state_set_alpha(st_, alpha));
      alpha_pushed_ = alpha;
    }
  }
  std::shared_ptr<models::group> fetch_group(size_t c, size_t gid) const {
    sync();
    const size_t slot = gm_.group(gid).data_;
    common::rng_t rng;
    auto g = hypers_.at(c)->create_group(rng);
    msc_feature_spec spec;
    std::vector<float> hp;
    hypers_[c]->device_spec(spec, hp);
    std::vector<uint8_t> rec(msc_ss_bytes(spec.family, spec.dim));
    if (!rec.empty()) check(msc_state_get_ss(st_, uint32_t(c), uint32_t(slot), 1, rec.data(), rec.size()));
    g->device_record_set(*hypers_[c], rec);
    return g;
  }
  void store_group(size_t c, size_t gid, const models::group &g) {
    sync();
    const size_t slot = gm_.group(gid).data_;
    std::vector<uint8_t> rec;
    g.device_record_get(*hypers_.at(c), rec);
    if (!rec.empty()) check(msc_state_set_ss(st_, uint32_t(c), uint32_t(slot), 1, rec.data(), rec.size()));
  }

  msc_context *ctx_;
  size_t n_, kmax_;
  common::group_manager<size_t> gm_;                   // group data = the group's device slot
  std::vector<models::hypers_shared_ptr> hypers_;
  std::vector<std::vector<float>> hp_pushed_;
  std::vector<bool> nonconj_;                          // per component: does a new group start from a draw?
  bool any_nonconj_ = false;
  float alpha_pushed_ = -1.f;
  std::vector<size_t> free_slots_;
  std::vector<int32_t> z_host_;                        // the assignment vector as slots, mirror of z_dev_
  msc_dataview *view_ = nullptr;
  msc_state *st_ = nullptr;
  int32_t *z_dev_ = nullptr;
  float *row_dev_ = nullptr, *row_host_ = nullptr, *sd_dev_ = nullptr;
};

}  // namespace hip
}  // namespace microscopes

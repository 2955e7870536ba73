// group_manager.hpp -- CRP bookkeeping on the host (the integer side of a mixture state):
// assignment vector, per-group entity counts, the set of empty groups, pseudocounts and
// the sequential CRP probability.  Interface of include/microscopes/common/
// group_manager.hpp:49-316 and the sampling helpers of util.hpp:85-156, re-authored.
// The device keeps the same counts per state (msc_state_set_group_counts) and applies
// the same pseudocount rule inside the score / sweep kernels.
#pragma once

#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <numeric>
#include <set>
#include <sys/types.h>

#include "recarray.hpp"
#include "types.hpp"
#include "wire.hpp"

namespace microscopes {
namespace common {

struct util {
  static void inplace_range(std::vector<std::size_t> &a, std::size_t n) {
    a.resize(n);
    std::iota(a.begin(), a.end(), std::size_t(0));
  }
  static std::vector<std::size_t> range(std::size_t n) {
    std::vector<std::size_t> r;
    inplace_range(r, n);
    return r;
  }
  static void inplace_permute(std::vector<std::size_t> &pi, std::size_t n, rng_t &rng) {
    inplace_range(pi, n);
    for (std::size_t i = n; i-- > 1;) std::swap(pi[std::uniform_int_distribution<std::size_t>(0, i)(rng)], pi[i]);
  }
  static std::vector<std::size_t> permute(std::size_t n, rng_t &rng) {
    std::vector<std::size_t> r;
    inplace_permute(r, n, rng);
    return r;
  }
  // softmax in place: shift by the max, exponentiate, normalise
  static void scores_to_probs(std::vector<float> &scores) {
    const float m = *std::max_element(scores.begin(), scores.end());
    float acc = 0.f;
    for (float &s : scores) acc += (s = std::exp(s - m));
    for (float &s : scores) s /= acc;
  }
  // inverse CDF with one uniform; falls back to the last index if rounding leaves a remainder
  static std::size_t sample_discrete(const std::vector<float> &probs, rng_t &rng) {
    float dart = std::uniform_real_distribution<float>(0.f, 1.f)(rng);
    for (std::size_t i = 0; i < probs.size(); i++)
      if ((dart -= probs[i]) <= 0.f) return i;
    return probs.size() - 1;
  }
  static std::size_t sample_discrete_log(std::vector<float> &scores, rng_t &rng) {
    scores_to_probs(scores);
    return sample_discrete(scores, rng);
  }
  template <typename T>
  static T sample_choice(const std::vector<T> &choices, rng_t &rng) {
    return choices[sample_discrete(std::vector<float>(choices.size(), 1.f / float(choices.size())), rng)];
  }
  static std::vector<std::size_t> random_assignment_vector(std::size_t n, rng_t &rng, std::size_t maxgroups = 100) {
    const auto groups = range(std::min(maxgroups, n));
    std::vector<std::size_t> r(n);
    for (auto &g : r) g = sample_choice(groups, rng);
    return r;
  }
};

template <typename T>
struct gd {
  gd() : count_(), data_() {}
  gd(std::size_t count, const T &data) : count_(count), data_(data) {}
  bool operator==(const gd &o) const { return count_ == o.count_ && data_ == o.data_; }
  bool operator!=(const gd &o) const { return !(*this == o); }
  std::size_t count_;
  T data_;
};

template <typename T>
class group_manager {
public:
  typedef typename std::map<std::size_t, gd<T>>::const_iterator const_iterator;

  group_manager() {}
  explicit group_manager(std::size_t n) : assignments_(n, -1) {}
  // inverse of serialize()
  group_manager(const serialized_t &repr, std::function<T(const std::string &)> load_group) {
    std::map<std::size_t, std::size_t> counts;
    std::vector<std::pair<std::size_t, std::string>> blobs;
    for (const auto &f : wire::parse(repr)) {
      if (f.number == 1) alpha_ = f.f32;
      else if (f.number == 2) {
        std::vector<uint64_t> v;
        wire::collect_varints(f, v);
        for (uint64_t a : v) {
          assignments_.push_back(ssize_t(int32_t(uint32_t(a))));   // int32: -1 is sign-extended on the wire
          if (assignments_.back() != -1) counts[std::size_t(assignments_.back())]++;
        }
      } else if (f.number == 3) {
        std::size_t id = 0;
        std::string data;
        for (const auto &g : wire::parse(f.bytes)) {
          if (g.number == 1) id = std::size_t(g.varint);
          if (g.number == 2) data = g.bytes;
        }
        blobs.emplace_back(id, data);
      }
    }
    for (const auto &b : blobs) {
      const std::size_t c = counts.count(b.first) ? counts[b.first] : 0;
      groups_[b.first] = gd<T>(c, load_group(b.second));
      if (!c) gempty_.insert(b.first);
    }
    if (!groups_.empty()) gcount_ = groups_.rbegin()->first + 1;
  }

  hyperparam_bag_t get_hp() const {
    wire::writer w;
    w.put_float_field(1, alpha_);
    return w.str();
  }
  void set_hp(const hyperparam_bag_t &hp) {
    for (const auto &f : wire::parse(hp))
      if (f.number == 1) alpha_ = f.f32;
  }
  value_mutator get_hp_mutator(const std::string &key) {
    if (key == "alpha") return value_mutator(&alpha_);
    throw std::runtime_error("unknown key: " + key);
  }

  const std::vector<ssize_t> &assignments() const { return assignments_; }
  const std::set<std::size_t> &empty_groups() const { return gempty_; }
  std::size_t nentities() const { return assignments_.size(); }
  std::size_t ngroups() const { return groups_.size(); }
  bool isactivegroup(std::size_t gid) const { return groups_.count(gid) != 0; }
  std::size_t groupsize(std::size_t gid) const { return group(gid).count_; }
  const gd<T> &group(std::size_t gid) const { return find(gid)->second; }
  gd<T> &group(std::size_t gid) { return const_cast<gd<T> &>(find(gid)->second); }
  std::vector<std::size_t> groups() const {
    std::vector<std::size_t> r;
    for (const auto &g : groups_) r.push_back(g.first);
    return r;
  }
  const_iterator begin() const { return groups_.begin(); }
  const_iterator end() const { return groups_.end(); }

  std::pair<std::size_t, T &> create_group() {
    const std::size_t gid = gcount_++;
    gempty_.insert(gid);
    return std::pair<std::size_t, T &>(gid, groups_[gid].data_);
  }
  void delete_group(std::size_t gid) {
    const auto it = find(gid);
    if (it->second.count_) throw std::runtime_error("group not empty");
    groups_.erase(gid);
    gempty_.erase(gid);
  }
  T &add_value(std::size_t gid, std::size_t eid) {
    if (assignments_.at(eid) != -1) throw std::runtime_error("entity already assigned");
    gd<T> &g = group(gid);
    if (g.count_++ == 0) gempty_.erase(gid);
    assignments_[eid] = ssize_t(gid);
    return g.data_;
  }
  std::pair<std::size_t, T &> remove_value(std::size_t eid) {
    if (assignments_.at(eid) == -1) throw std::runtime_error("entity not assigned");
    const std::size_t gid = std::size_t(assignments_[eid]);
    gd<T> &g = group(gid);
    if (--g.count_ == 0) gempty_.insert(gid);
    assignments_[eid] = -1;
    return std::pair<std::size_t, T &>(gid, g.data_);
  }

  // the whole assignment vector at once (every gid must exist): sizes and the empty set follow in O(n + groups)
  void reassign_all(const std::vector<ssize_t> &a) {
    if (a.size() != assignments_.size()) throw std::runtime_error("wrong number of entities");
    {                                                    // check before touching anything
      ssize_t seen = -2;
      for (ssize_t gid : a)
        if (gid != -1 && gid != seen) {
          (void)find(std::size_t(gid));                  // throws "invalid gid"
          seen = gid;
        }
    }
    for (auto &g : groups_) g.second.count_ = 0;
    gd<T> *last = nullptr;                               // consecutive entities mostly share few groups: skip the map when they repeat
    ssize_t last_gid = -2;
    for (ssize_t gid : a) {
      if (gid == -1) continue;
      if (gid != last_gid) {
        last = &group(std::size_t(gid));
        last_gid = gid;
      }
      last->count_++;
    }
    assignments_ = a;
    gempty_.clear();
    for (const auto &g : groups_)
      if (!g.second.count_) gempty_.insert(g.first);
  }

  // sequential CRP probability of the assignment vector
  float score_assignment() const {
    std::map<ssize_t, std::size_t> seen;
    float sum = 0.f;
    for (std::size_t i = 0; i < assignments_.size(); i++) {
      const ssize_t g = assignments_[i];
      if (g == -1) throw std::runtime_error("not assigned");
      std::size_t &c = seen[g];
      if (i) sum += std::log((c ? float(c) : alpha_) / (float(i) + alpha_));
      c++;
    }
    return sum;
  }
  float pseudocount(std::size_t, const gd<T> &g) const {
    return g.count_ ? float(g.count_) : alpha_ / float(gempty_.size());
  }

  // GroupManager message of microscopes/io/schema.proto:38-46
  serialized_t serialize(std::function<serialized_t(const T &)> dump_group) const {
    wire::writer w;
    w.put_float_field(1, alpha_);
    for (ssize_t a : assignments_) w.put_varint_field(2, uint64_t(int64_t(a)));
    for (const auto &g : groups_) {
      wire::writer gw;
      gw.put_varint_field(1, g.first);
      gw.put_bytes_field(2, dump_group(g.second.data_));
      w.put_bytes_field(3, gw.str());
    }
    return w.str();
  }

protected:
  typename std::map<std::size_t, gd<T>>::const_iterator find(std::size_t gid) const {
    const auto it = groups_.find(gid);
    if (it == groups_.end()) throw std::runtime_error("invalid gid");
    return it;
  }
  float alpha_ = 0.f;
  std::size_t gcount_ = 0;
  std::set<std::size_t> gempty_;
  std::vector<ssize_t> assignments_;
  std::map<std::size_t, gd<T>> groups_;
};

// Manages groups and nothing else (group_manager.hpp:320-386): ids are handed out in increasing order and never
// reused, O(log N) create / delete / lookup, iteration in id order.
template <typename T>
class simple_group_manager {
public:
  typedef typename std::map<std::size_t, T>::const_iterator const_iterator;

  simple_group_manager() = default;
  std::size_t ngroups() const { return groups_.size(); }
  std::vector<std::size_t> groups() const {
    std::vector<std::size_t> ret;
    ret.reserve(groups_.size());
    for (const auto &g : groups_) ret.push_back(g.first);
    return ret;
  }
  std::pair<std::size_t, T &> create_group() {
    const std::size_t gid = gcount_++;
    return std::pair<std::size_t, T &>(gid, groups_[gid]);
  }
  void delete_group(std::size_t gid) {
    const auto it = groups_.find(gid);
    if (it == groups_.end()) throw std::runtime_error("invalid gid");
    groups_.erase(it);
  }
  const T &group(std::size_t gid) const {
    const auto it = groups_.find(gid);
    if (it == groups_.end()) throw std::runtime_error("invalid gid");
    return it->second;
  }
  T &group(std::size_t gid) {
    const auto it = groups_.find(gid);
    if (it == groups_.end()) throw std::runtime_error("invalid gid");
    return it->second;
  }
  const_iterator begin() const { return groups_.begin(); }
  const_iterator end() const { return groups_.end(); }

private:
  std::size_t gcount_ = 0;
  std::map<std::size_t, T> groups_;
};

}  // namespace common
}  // namespace microscopes

// entity_state.hpp -- the contract between a clustered state and the sampling kernels that drive it:
// the abstract class of include/microscopes/common/entity_state.hpp:25-90 (same names, signatures and
// defaults, so downstream kernels written against it compile unchanged), re-authored.
//
// A state is N entities, a partition of them into groups (the clustering), and per component one
// likelihood model whose sufficient statistics are kept per group.  The Gibbs assignment kernel is
//   for eid in entities: remove_value(eid); score_value(eid) -> sample a group; add_value(gid, eid)
// with create_group / delete_group keeping at least one empty group on offer.
// mixture_state.hpp implements it on the device tables of microscopes_hip.h.
#pragma once

#include <string>
#include <sys/types.h>
#include <utility>
#include <vector>

#include "plugin.hpp"
#include "types.hpp"

namespace microscopes {
namespace common {

class entity_based_state_object {
public:
  typedef std::pair<std::vector<size_t>, std::vector<float>> scores_t;   // (group ids, one score each)

  virtual ~entity_based_state_object() {}

  // ---- sizes and the partition ----
  virtual size_t nentities() const = 0;
  virtual size_t ngroups() const = 0;
  virtual size_t ncomponents() const = 0;
  virtual std::vector<ssize_t> assignments() const = 0;      // -1 = unassigned
  virtual std::vector<size_t> groups() const = 0;
  virtual size_t groupsize(size_t gid) const = 0;

  // ---- parameters: the clustering prior, one likelihood model per component ----
  virtual hyperparam_bag_t get_cluster_hp() const = 0;
  virtual void set_cluster_hp(const hyperparam_bag_t &hp) = 0;
  virtual value_mutator get_cluster_hp_mutator(const std::string &key) = 0;

  virtual hyperparam_bag_t get_component_hp(size_t component) const = 0;
  virtual void set_component_hp(size_t component, const hyperparam_bag_t &hp) = 0;
  virtual void set_component_hp(size_t component, const models::hypers &proto) = 0;
  virtual value_mutator get_component_hp_mutator(size_t component, const std::string &key) = 0;

  // ---- sufficient statistics, one set per (component, identifier) ----
  virtual std::vector<ident_t> suffstats_identifiers(size_t component) const = 0;
  virtual suffstats_bag_t get_suffstats(size_t component, ident_t id) const = 0;
  virtual void set_suffstats(size_t component, ident_t id, const suffstats_bag_t &ss) = 0;
  virtual value_mutator get_suffstats_mutator(size_t component, ident_t id, const std::string &key) = 0;

  // ---- membership ----
  virtual void add_value(size_t gid, size_t eid, rng_t &rng) = 0;
  virtual size_t remove_value(size_t eid, rng_t &rng) = 0;     // -> the group the entity was in

  // log-score of the (unassigned) entity joining each group; the by-value form forwards to the in-place one
  virtual scores_t score_value(size_t eid, rng_t &rng) const {
    scores_t r;
    inplace_score_value(r, eid, rng);
    return r;
  }
  virtual void inplace_score_value(scores_t &scores, size_t eid, rng_t &rng) const = 0;

  virtual float score_assignment() const = 0;
  virtual float score_likelihood(size_t component, ident_t id, rng_t &rng) const = 0;
  virtual float score_likelihood(size_t component, rng_t &rng) const {
    float total = 0.f;
    for (ident_t id : suffstats_identifiers(component)) total += score_likelihood(component, id, rng);
    return total;
  }

  // ---- the supply of empty groups ----
  virtual std::vector<size_t> empty_groups() const = 0;
  virtual size_t create_group(rng_t &rng) = 0;
  virtual void delete_group(size_t gid) = 0;
};

}  // namespace common
}  // namespace microscopes

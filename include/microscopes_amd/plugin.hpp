// plugin.hpp -- the component-model plugin API downstream state objects program
// against: abstract model / hypers / group, same names, signatures, ownership and
// error convention as include/microscopes/models/base.hpp:21-72 (the caller passes the
// hypers on every call; a group keeps no pointer to it; errors are C++ exceptions).
#pragma once

#include <memory>

#include "recarray.hpp"
#include "types.hpp"

namespace microscopes {
namespace models {

class hypers;

// sufficient statistics of one group of one feature
class group {
public:
  virtual ~group() {}
  virtual void add_value(const hypers &m, const common::value_accessor &value, common::rng_t &rng) = 0;
  virtual void remove_value(const hypers &m, const common::value_accessor &value, common::rng_t &rng) = 0;
  virtual float score_value(const hypers &m, const common::value_accessor &value, common::rng_t &rng) const = 0;
  virtual float score_data(const hypers &m, common::rng_t &rng) const = 0;
  virtual void sample_value(const hypers &m, common::value_mutator &value, common::rng_t &rng) const = 0;
  virtual common::suffstats_bag_t get_ss() const = 0;
  virtual void set_ss(const common::suffstats_bag_t &ss) = 0;
  virtual void set_ss(const group &g) = 0;
  virtual common::value_mutator get_ss_mutator(const std::string &key) = 0;
  virtual std::string debug_str() const = 0;

  // -- HIP backend extension (nothing like it upstream): the suff-stats as the packed record of
  //    msc_state_set_ss / msc_state_get_ss, so a batched device state (mixture_state.hpp) can host this group.
  //    false = this model has no device family.
  virtual bool device_record_get(const hypers &, std::vector<uint8_t> &) const { return false; }
  virtual bool device_record_set(const hypers &, const std::vector<uint8_t> &) { return false; }
};

// hyper-parameters of one feature
class hypers {
public:
  virtual ~hypers() {}
  virtual common::hyperparam_bag_t get_hp() const = 0;
  virtual void set_hp(const common::hyperparam_bag_t &hp) = 0;
  virtual void set_hp(const hypers &s) = 0;
  virtual common::value_mutator get_hp_mutator(const std::string &key) = 0;
  virtual std::shared_ptr<group> create_group(common::rng_t &rng) const = 0;
  virtual std::string debug_str() const = 0;

  // -- HIP backend extension: the kernel family + dimension and the hyper-parameters as msc_state_set_hp takes them
  virtual bool device_spec(msc_feature_spec &, std::vector<float> &) const { return false; }
};

class model {
public:
  virtual ~model() {}
  virtual std::shared_ptr<hypers> create_hypers() const = 0;
  virtual common::runtime_type get_runtime_type() const = 0;
};

typedef group *group_raw_ptr;
typedef std::shared_ptr<group> group_shared_ptr;
typedef hypers *hypers_raw_ptr;
typedef std::shared_ptr<hypers> hypers_shared_ptr;
typedef model *model_raw_ptr;
typedef std::shared_ptr<model> model_shared_ptr;

// ---- zero-work model: measures the cost of the API itself (models/noop.hpp:13-53) ----
class noop_group : public group {
public:
  void add_value(const hypers &, const common::value_accessor &, common::rng_t &) override {}
  void remove_value(const hypers &, const common::value_accessor &, common::rng_t &) override {}
  float score_value(const hypers &, const common::value_accessor &, common::rng_t &) const override { return 0.f; }
  float score_data(const hypers &, common::rng_t &) const override { return 0.f; }
  void sample_value(const hypers &, common::value_mutator &, common::rng_t &) const override {}
  common::suffstats_bag_t get_ss() const override { return ""; }
  void set_ss(const common::suffstats_bag_t &) override {}
  void set_ss(const group &) override {}
  common::value_mutator get_ss_mutator(const std::string &) override { throw std::runtime_error("noop"); }
  std::string debug_str() const override { return "<noop>"; }
  bool device_record_get(const hypers &, std::vector<uint8_t> &rec) const override {
    rec.clear();
    return true;
  }
  bool device_record_set(const hypers &, const std::vector<uint8_t> &) override { return true; }
};

class noop_hypers : public hypers {
public:
  common::hyperparam_bag_t get_hp() const override { return ""; }
  void set_hp(const common::hyperparam_bag_t &) override {}
  void set_hp(const hypers &) override {}
  common::value_mutator get_hp_mutator(const std::string &) override { throw std::runtime_error("noop"); }
  std::shared_ptr<group> create_group(common::rng_t &) const override { return std::make_shared<noop_group>(); }
  std::string debug_str() const override { return "<noop>"; }
  bool device_spec(msc_feature_spec &spec, std::vector<float> &hp) const override {
    spec.family = MSC_NOOP;
    spec.dim = 0;
    hp.clear();
    return true;
  }
};

class noop_model : public model {
public:
  std::shared_ptr<hypers> create_hypers() const override { return std::make_shared<noop_hypers>(); }
  common::runtime_type get_runtime_type() const override { return common::runtime_type(TYPE_B); }
};

}  // namespace models
}  // namespace microscopes

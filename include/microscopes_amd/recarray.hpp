// recarray.hpp -- host-side view of a packed row-major record array: the iteration API
// downstream code uses (recarray::row_accessor / row_mutator / dataview /
// row_major_dataview of include/microscopes/common/recarray/dataview.hpp), re-authored.
// The device sees the same data through msc_dataview_from_records (columnar in HBM);
// row_major_dataview::to_device() performs that upload.
#pragma once

#include <algorithm>
#include <memory>
#include <random>
#include <utility>

#include "../microscopes_hip.h"
#include "types.hpp"

namespace microscopes {
namespace common {

typedef std::default_random_engine rng_t;   // microscopes/common/_random_fwd_h.pxd:1-8

namespace recarray {

class row_mutator;

// cursor over the fields of one record
class row_accessor {
  friend class row_mutator;

public:
  row_accessor() = default;
  row_accessor(const uint8_t *data, const bool *mask, const std::vector<runtime_type> *types)
      : data_(data), mask_(mask), types_(types), cur_(data), mcur_(mask) {}

  std::size_t tell() const { return pos_; }
  std::size_t nfeatures() const { return types_->size(); }
  const runtime_type &curtype() const { return (*types_)[pos_]; }
  unsigned curshape() const { return curtype().n(); }
  value_accessor get() const { return value_accessor(cur_, mcur_, curtype()); }
  bool ismasked(std::size_t idx) const { return get().ismasked(idx); }
  bool anymasked() const { return get().anymasked(); }
  void bump() {
    cur_ += curtype().size();
    if (mask_) mcur_ += curtype().n();
    ++pos_;
  }
  bool end() const { return pos_ == nfeatures(); }
  void reset() {
    cur_ = data_;
    mcur_ = mask_;
    pos_ = 0;
  }
  std::string debug_str() const {
    row_accessor a(data_, mask_, types_);
    std::string s = "{";
    for (; !a.end(); a.bump()) s += (a.tell() ? ", " : "") + (a.anymasked() ? std::string("--") : a.get().debug_str());
    return s + "}";
  }

private:
  const uint8_t *data_ = nullptr;
  const bool *mask_ = nullptr;
  const std::vector<runtime_type> *types_ = nullptr;
  const uint8_t *cur_ = nullptr;
  const bool *mcur_ = nullptr;
  std::size_t pos_ = 0;
};

class row_mutator {
public:
  row_mutator() = default;
  row_mutator(uint8_t *data, const std::vector<runtime_type> *types) : data_(data), types_(types), cur_(data) {}

  std::size_t tell() const { return pos_; }
  std::size_t nfeatures() const { return types_->size(); }
  const runtime_type &curtype() const { return (*types_)[pos_]; }
  unsigned curshape() const { return curtype().n(); }
  value_mutator set() const { return value_mutator(cur_, curtype()); }
  template <typename T>
  void set(T t, std::size_t idx) { set().set<T>(t, idx); }
  // copy the accessor's current field into ours, converting element-wise
  void set(const row_accessor &acc) {
    if (curshape() != acc.curshape()) throw std::runtime_error("shapes do not match");
    for (unsigned i = 0; i < curshape(); i++)
      runtime_cast::copy(cur_ + i * curtype().psize(), curtype().t(),
                         acc.cur_ + i * acc.curtype().psize(), acc.curtype().t());
  }
  void bump() {
    cur_ += curtype().size();
    ++pos_;
  }
  bool end() const { return pos_ == nfeatures(); }
  void reset() {
    cur_ = data_;
    pos_ = 0;
  }

private:
  uint8_t *data_ = nullptr;
  const std::vector<runtime_type> *types_ = nullptr;
  uint8_t *cur_ = nullptr;
  std::size_t pos_ = 0;
};

class dataview {
public:
  virtual ~dataview() {}
  virtual row_accessor get() const = 0;
  virtual std::size_t index() const = 0;
  virtual void next() = 0;
  virtual void reset() = 0;
  virtual bool end() const = 0;
  virtual row_accessor get(std::size_t idx) const = 0;
  std::size_t size() const { return n_; }
  const std::vector<runtime_type> &types() const { return types_; }

protected:
  dataview(std::size_t n, const std::vector<runtime_type> &types) : n_(n), types_(types) {
    const auto r = runtime_type::GetOffsetsAndSize(types);
    offsets_ = r.offsets_;
    rowsize_ = r.rowsize_;
    maskrowsize_ = r.maskrowsize_;
  }
  const std::vector<std::size_t> &offsets() const { return offsets_; }
  std::size_t rowsize() const { return rowsize_; }
  std::size_t maskrowsize() const { return maskrowsize_; }

private:
  std::size_t n_;
  std::vector<runtime_type> types_;
  std::vector<std::size_t> offsets_;
  std::size_t rowsize_ = 0, maskrowsize_ = 0;
};

class row_major_dataview : public dataview {
public:
  row_major_dataview(const uint8_t *data, const bool *mask, std::size_t n, const std::vector<runtime_type> &types)
      : dataview(n, types), data_(data), mask_(mask) {}

  row_accessor get() const override { return get(index()); }
  std::size_t index() const override { return pi_.empty() ? pos_ : pi_[pos_]; }
  void next() override { ++pos_; }
  void reset() override { pos_ = 0; }
  bool end() const override { return pos_ == size(); }
  row_accessor get(std::size_t i) const override {
    return row_accessor(data_ + rowsize() * i, mask_ ? mask_ + maskrowsize() * i : nullptr, &types());
  }
  void reset_permutation() { pi_.clear(); }
  // Fisher-Yates over 0..n-1 (util::inplace_permute, util.hpp:85-94)
  void permute(rng_t &rng) {
    pi_.resize(size());
    for (std::size_t i = 0; i < pi_.size(); i++) pi_[i] = i;
    for (std::size_t i = pi_.size(); i-- > 1;) std::swap(pi_[std::uniform_int_distribution<std::size_t>(0, i)(rng)], pi_[i]);
  }

  // upload as a columnar device view; col_types: target primitive type per feature (or null)
  msc_dataview *to_device(msc_context *ctx, const std::vector<int32_t> *col_types = nullptr) const {
    std::vector<msc_runtime_type> rt;
    for (const runtime_type &t : types()) rt.push_back(msc_runtime_type{int32_t(t.t()), t.n()});
    msc_dataview *out = nullptr;
    if (msc_dataview_from_records(ctx, data_, reinterpret_cast<const uint8_t *>(mask_), size(), rt.data(),
                                  uint32_t(rt.size()), col_types ? col_types->data() : nullptr, &out) != MSC_OK)
      throw std::runtime_error(msc_last_error());
    return out;
  }

private:
  const uint8_t *data_;
  const bool *mask_;
  std::size_t pos_ = 0;
  std::vector<std::size_t> pi_;
};

}  // namespace recarray
}  // namespace common
}  // namespace microscopes

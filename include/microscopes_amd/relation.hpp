// relation.hpp -- relation (N-d array) dataviews: the host-side surface of
// include/microscopes/common/relation/dataview.hpp:25-578, re-authored, plus the bridge that lets
// the scoring kernels serve per-cell models (irm): a relation becomes a one-feature device dataview
// whose rows are the relation's cells, and msc_relation_blocks turns the per-dimension cluster
// assignments into the cell's block (= group) index.
//
// Same two operations as the reference (dataview.hpp:15-24): get(indices), and slice(dim, idx), which
// iterates the present (unmasked / stored) cells whose dim-th index is idx.
#pragma once

#include <cstring>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../microscopes_hip.h"
#include "types.hpp"

namespace microscopes {
namespace common {
namespace relation {

class dataview {
public:
  typedef std::pair<std::vector<size_t>, value_accessor> value_with_position_t;

  // One concrete cursor serves both implementations: a dense slice walks a strided index product,
  // a compressed slice walks one row of a CSR / CSC triple.  (The reference type-erases an
  // implementation object per view class, dataview.hpp:35-112; the iteration protocol -- input
  // iterator over value_with_position_t, begin()/end() of a slice_iterable -- is the same.)
  class slice_iterator {
  public:
    typedef std::input_iterator_tag iterator_category;
    typedef const value_with_position_t value_type;
    typedef std::ptrdiff_t difference_type;
    typedef const value_with_position_t *pointer;
    typedef const value_with_position_t &reference;

    slice_iterator() = default;
    const value_with_position_t &operator*() const { return cur_; }
    const value_with_position_t *operator->() const { return &cur_; }
    slice_iterator &operator++() {
      advance();
      return *this;
    }
    slice_iterator operator++(int) {
      slice_iterator t(*this);
      advance();
      return t;
    }
    bool operator==(const slice_iterator &o) const { return view_ == o.view_ && pos_ == o.pos_ && end_ == o.end_; }
    bool operator!=(const slice_iterator &o) const { return !(*this == o); }

  private:
    friend class row_major_dense_dataview;
    friend class compressed_2darray;
    void advance();
    void settle();            // dense: skip masked cells; load cur_
    const dataview *view_ = nullptr;
    size_t dim_ = 0, idx_ = 0;
    size_t pos_ = 0, end_ = 0;   // dense: ordinal within the slice; compressed: position in indices[]
    value_with_position_t cur_;
  };

  class slice_iterable {
  public:
    slice_iterable() = default;
    slice_iterable(const slice_iterator &b, const slice_iterator &e) : begin_(b), end_(e) {}
    slice_iterator begin() const { return begin_; }
    slice_iterator end() const { return end_; }

  private:
    slice_iterator begin_, end_;
  };

  dataview(const std::vector<size_t> &shape, const runtime_type &type) : shape_(shape), type_(type) {
    if (shape_.empty()) throw std::runtime_error("zero-d array not allowed");            // dataview.hpp:138
    for (size_t s : shape_)
      if (!s) throw std::runtime_error("empty-dimesion not allowed");                    // dataview.hpp:141
  }
  virtual ~dataview() {}

  size_t dims() const { return shape_.size(); }
  const std::vector<size_t> &shape() const { return shape_; }
  const runtime_type &type() const { return type_; }
  size_t ncells() const {
    size_t n = 1;
    for (size_t s : shape_) n *= s;
    return n;
  }

  virtual value_accessor get(const std::vector<size_t> &indices) const = 0;
  virtual slice_iterable slice(size_t dim, size_t idx) const = 0;

protected:
  friend class slice_iterator;
  virtual bool is_dense() const = 0;
  std::vector<size_t> shape_;
  runtime_type type_;
};

// a dense numpy.ndarray (+ optional mask, one bool per cell); row-major
class row_major_dense_dataview : public dataview {
public:
  row_major_dense_dataview(const uint8_t *data, const bool *mask, const std::vector<size_t> &shape,
                           const runtime_type &type)
      : dataview(shape, type), data_(data), mask_(mask), step_(type.size()) {
    if (!data) throw std::runtime_error("data cannot be null");
    mult_.assign(shape.size(), 1);
    for (size_t i = shape.size() - 1; i-- > 0;) mult_[i] = mult_[i + 1] * shape[i + 1];
  }

  value_accessor get(const std::vector<size_t> &indices) const override {
    if (indices.size() != dims()) throw std::runtime_error("invalid # of indices");
    for (size_t i = 0; i < dims(); i++)
      if (indices[i] >= shape_[i]) throw std::runtime_error("index out of bounds");
    const size_t off = offset(indices);
    return value_accessor(data_ + off * step_, mask_ ? mask_ + off * type_.n() : nullptr, type_);
  }

  slice_iterable slice(size_t dim, size_t idx) const override {
    if (dim >= dims()) throw std::runtime_error("invalid dimension");
    if (idx >= shape_[dim]) throw std::runtime_error("invalid index");
    slice_iterator b, e;
    b.view_ = e.view_ = this;
    b.dim_ = e.dim_ = dim;
    b.idx_ = e.idx_ = idx;
    b.end_ = e.end_ = e.pos_ = ncells() / shape_[dim];
    b.pos_ = 0;
    b.settle();
    return slice_iterable(b, e);
  }

  // the relation's cells as a one-feature device dataview (row = cell, row-major order), mask included
  msc_dataview *to_device_cells(msc_context *ctx) const {
    const msc_runtime_type t = {int32_t(type_.t()), type_.n()};
    msc_dataview *out = nullptr;
    if (msc_dataview_from_records(ctx, data_, reinterpret_cast<const uint8_t *>(mask_), ncells(), &t, 1, nullptr, &out) != MSC_OK)
      throw std::runtime_error(msc_last_error());
    return out;
  }

  size_t offset(const std::vector<size_t> &indices) const {
    size_t off = 0;
    for (size_t i = 0; i < dims(); i++) off += indices[i] * mult_[i];
    return off;
  }
  bool masked(size_t off) const {
    if (!mask_) return false;
    for (unsigned e = 0; e < type_.n(); e++)
      if (mask_[off * type_.n() + e]) return true;
    return false;
  }
  // position of the ord-th cell of slice (dim, idx): the other dimensions count in row-major order
  void position(size_t dim, size_t idx, size_t ord, std::vector<size_t> &pos) const {
    pos.resize(dims());
    for (size_t i = dims(); i-- > 0;) {
      if (i == dim) {
        pos[i] = idx;
        continue;
      }
      pos[i] = ord % shape_[i];
      ord /= shape_[i];
    }
  }

protected:
  bool is_dense() const override { return true; }

private:
  friend class dataview::slice_iterator;
  const uint8_t *data_;
  const bool *mask_;
  size_t step_;
  std::vector<size_t> mult_;
};

// scipy.sparse csr + csc of the same 2-d matrix; entries that are not stored are MISSING, not zero
// (dataview.hpp:420-424).  A slice is linear in the entries of that row / column.
class compressed_2darray : public dataview {
public:
  compressed_2darray(const uint8_t *csr_data, const uint32_t *csr_indices, const uint32_t *csr_indptr,
                     const uint8_t *csc_data, const uint32_t *csc_indices, const uint32_t *csc_indptr,
                     size_t rows, size_t cols, const runtime_type &type)
      : dataview({rows, cols}, type), data_{csr_data, csc_data}, indices_{csr_indices, csc_indices},
        indptr_{csr_indptr, csc_indptr} {}

  value_accessor get(const std::vector<size_t> &indices) const override {
    if (indices.size() != 2) throw std::runtime_error("bad size given");
    if (indices[0] >= shape_[0] || indices[1] >= shape_[1]) throw std::runtime_error("index out of bounds");
    // (unimplemented upstream, dataview.hpp:517-521; a linear probe of the row is all it takes)
    for (uint32_t p = indptr_[0][indices[0]]; p < indptr_[0][indices[0] + 1]; p++)
      if (indices_[0][p] == indices[1]) return value_accessor(data_[0] + size_t(p) * type_.size(), nullptr, type_);
    throw std::runtime_error("entry not present");
  }

  slice_iterable slice(size_t dim, size_t idx) const override {
    if (dim >= 2) throw std::runtime_error("invalid dimension");
    if (idx >= shape_[dim]) throw std::runtime_error("invalid index");
    slice_iterator b, e;
    b.view_ = e.view_ = this;
    b.dim_ = e.dim_ = dim;
    b.idx_ = e.idx_ = idx;
    b.pos_ = indptr_[dim][idx];
    b.end_ = e.end_ = e.pos_ = indptr_[dim][idx + 1];
    b.settle();
    return slice_iterable(b, e);
  }

  size_t nnz() const { return indptr_[0][shape_[0]]; }

  // the stored entries (csr order) as a one-feature device dataview; positions[2 * c] = (row, col) of cell c
  msc_dataview *to_device_cells(msc_context *ctx, std::vector<uint32_t> *positions) const {
    const msc_runtime_type t = {int32_t(type_.t()), type_.n()};
    msc_dataview *out = nullptr;
    if (msc_dataview_from_records(ctx, data_[0], nullptr, nnz(), &t, 1, nullptr, &out) != MSC_OK)
      throw std::runtime_error(msc_last_error());
    if (positions) {
      positions->resize(2 * nnz());
      for (size_t r = 0; r < shape_[0]; r++)
        for (uint32_t p = indptr_[0][r]; p < indptr_[0][r + 1]; p++) {
          (*positions)[2 * size_t(p)] = uint32_t(r);
          (*positions)[2 * size_t(p) + 1] = indices_[0][p];
        }
    }
    return out;
  }

protected:
  bool is_dense() const override { return false; }

private:
  friend class dataview::slice_iterator;
  const uint8_t *data_[2];
  const uint32_t *indices_[2];
  const uint32_t *indptr_[2];
};

inline void dataview::slice_iterator::settle() {
  if (view_->is_dense()) {
    const auto *v = static_cast<const row_major_dense_dataview *>(view_);
    for (; pos_ < end_; pos_++) {
      v->position(dim_, idx_, pos_, cur_.first);
      const size_t off = v->offset(cur_.first);
      if (v->masked(off)) continue;                 // a slice only shows the cells that are present
      cur_.second = value_accessor(v->data_ + off * v->step_, nullptr, v->type());
      return;
    }
  } else if (pos_ < end_) {
    const auto *v = static_cast<const compressed_2darray *>(view_);
    cur_.first.resize(2);
    cur_.first[dim_] = idx_;
    cur_.first[1 - dim_] = v->indices_[dim_][pos_];
    cur_.second = value_accessor(v->data_[dim_] + pos_ * v->type().size(), nullptr, v->type());
  }
}
inline void dataview::slice_iterator::advance() {
  if (pos_ < end_) {
    pos_++;
    settle();
  }
}

// cell -> block index on the device: block = sum_d z_d[index_d] * prod_{e > d} ngroups[e] (the last
// dimension's cluster varies fastest), -1 when a dimension's entity is unassigned.  Dense relations
// pass positions_dev = null (cells are in row-major order); compressed ones pass their (row, col) pairs.
inline void cell_blocks(msc_context *ctx, const std::vector<size_t> &shape, const std::vector<const int32_t *> &z_dev,
                        const std::vector<uint32_t> &ngroups, const uint32_t *positions_dev, uint64_t ncells,
                        int32_t *z_cell_dev) {
  std::vector<uint64_t> sh(shape.begin(), shape.end());
  if (msc_relation_blocks(ctx, uint32_t(shape.size()), sh.data(), z_dev.data(), ngroups.data(), positions_dev, ncells,
                          z_cell_dev) != MSC_OK)
    throw std::runtime_error(msc_last_error());
}

// irm's slice reduction on the device (msc_relation_slice_scores): out[e][g] = sum over the cells of slice (dim, e) of
// the cell's score against block (g, the cell's other clusters).  scores_dev = what msc_score_value wrote for the
// relation's cells; off_dev = cell_blocks() with an all-zero assignment vector for `dim`; seg / ids non-null for a
// compressed relation (the rows of its CSR, or of its transpose for dim 1).
inline void slice_scores(msc_context *ctx, const float *scores_dev, uint64_t ld, const std::vector<size_t> &shape, size_t dim,
                         const uint32_t *seg_dev, const uint32_t *ids_dev, const int32_t *off_dev, const std::vector<uint32_t> &ngroups,
                         uint64_t nent, float *out_dev, uint64_t ld_out) {
  std::vector<uint64_t> sh(shape.begin(), shape.end());
  uint32_t stride = 1;
  for (size_t d = dim + 1; d < ngroups.size(); d++) stride *= ngroups[d];
  if (msc_relation_slice_scores(ctx, scores_dev, ld, uint32_t(shape.size()), sh.data(), uint32_t(dim), seg_dev, ids_dev, off_dev,
                                ngroups.at(dim), stride, nent, out_dev, ld_out) != MSC_OK)
    throw std::runtime_error(msc_last_error());
}

}  // namespace relation
}  // namespace common
}  // namespace microscopes

// types.hpp -- runtime type system of the plugin surface: the eleven primitive
// types, runtime_type (scalar or fixed-length vector of a primitive), typed
// accessors over raw bytes.  Own implementation of the interface downstream code
// uses from the reference (include/microscopes/common/{type_info.h,runtime_type.hpp,
// runtime_value.hpp}); names and semantics kept, code re-authored.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

// The enumerators live at global scope, as downstream code writes runtime_type(TYPE_B).
// Numbering is shared with msc_primitive_type (include/microscopes_hip.h).
enum primitive_type {
  TYPE_B = 0, TYPE_I8, TYPE_U8, TYPE_I16, TYPE_U16, TYPE_I32, TYPE_U32, TYPE_I64, TYPE_U64,
  TYPE_F32, TYPE_F64, TYPE_NELEMS
};

namespace microscopes {
namespace common {

namespace detail {
constexpr std::size_t k_primitive_bytes[TYPE_NELEMS] = {1, 1, 1, 2, 2, 4, 4, 8, 8, 4, 8};
constexpr const char *k_primitive_names[TYPE_NELEMS] = {
    "TYPE_B", "TYPE_I8", "TYPE_U8", "TYPE_I16", "TYPE_U16", "TYPE_I32",
    "TYPE_U32", "TYPE_I64", "TYPE_U64", "TYPE_F32", "TYPE_F64"};

// call fn with a value-initialised object of the C type behind `t`
template <typename Fn>
inline void with_ctype(primitive_type t, Fn &&fn) {
  switch (t) {
    case TYPE_B: fn(bool()); break;
    case TYPE_I8: fn(int8_t()); break;
    case TYPE_U8: fn(uint8_t()); break;
    case TYPE_I16: fn(int16_t()); break;
    case TYPE_U16: fn(uint16_t()); break;
    case TYPE_I32: fn(int32_t()); break;
    case TYPE_U32: fn(uint32_t()); break;
    case TYPE_I64: fn(int64_t()); break;
    case TYPE_U64: fn(uint64_t()); break;
    case TYPE_F32: fn(float()); break;
    case TYPE_F64: fn(double()); break;
    default: throw std::runtime_error("bad primitive type");
  }
}
}  // namespace detail

template <typename T> struct static_type_to_primitive_type;
template <> struct static_type_to_primitive_type<bool> { static const primitive_type value = TYPE_B; };
template <> struct static_type_to_primitive_type<int8_t> { static const primitive_type value = TYPE_I8; };
template <> struct static_type_to_primitive_type<uint8_t> { static const primitive_type value = TYPE_U8; };
template <> struct static_type_to_primitive_type<int16_t> { static const primitive_type value = TYPE_I16; };
template <> struct static_type_to_primitive_type<uint16_t> { static const primitive_type value = TYPE_U16; };
template <> struct static_type_to_primitive_type<int32_t> { static const primitive_type value = TYPE_I32; };
template <> struct static_type_to_primitive_type<uint32_t> { static const primitive_type value = TYPE_U32; };
template <> struct static_type_to_primitive_type<int64_t> { static const primitive_type value = TYPE_I64; };
template <> struct static_type_to_primitive_type<uint64_t> { static const primitive_type value = TYPE_U64; };
template <> struct static_type_to_primitive_type<float> { static const primitive_type value = TYPE_F32; };
template <> struct static_type_to_primitive_type<double> { static const primitive_type value = TYPE_F64; };

struct primitive_type_traits {
  static std::size_t PrimitiveTypeSize(primitive_type t) { return detail::k_primitive_bytes[t]; }
  static const char *PrimitiveTypeStr(primitive_type t) { return detail::k_primitive_names[t]; }
  static std::string ToString(primitive_type t, const uint8_t *px) {
    std::ostringstream os;
    detail::with_ctype(t, [&](auto tag) {
      decltype(tag) v;
      std::memcpy(&v, px, sizeof v);
      os << v;
    });
    return os.str();
  }
};

class runtime_type {
public:
  runtime_type() = default;
  runtime_type(primitive_type t) : t_(t), psize_(unsigned(detail::k_primitive_bytes[t])), n_(1), vec_(false) {}
  runtime_type(primitive_type t, unsigned n)
      : t_(t), psize_(unsigned(detail::k_primitive_bytes[t])), n_(n), vec_(true) {}

  primitive_type t() const { return t_; }
  unsigned psize() const { return psize_; }   // bytes of one element
  unsigned n() const { return n_; }           // elements
  unsigned size() const { return n_ * psize_; }
  bool vec() const { return vec_; }
  bool operator==(const runtime_type &o) const { return t_ == o.t_ && n_ == o.n_ && vec_ == o.vec_; }
  bool operator!=(const runtime_type &o) const { return !(*this == o); }
  std::string str() const {
    std::string s = detail::k_primitive_names[t_];
    return vec_ ? s + "[" + std::to_string(n_) + "]" : s;
  }

  struct offsets_ret_t {
    std::vector<std::size_t> offsets_;
    std::size_t rowsize_ = 0, maskrowsize_ = 0;
  };
  // byte offset of every field of a packed record, record size, mask-record size
  static offsets_ret_t GetOffsetsAndSize(const std::vector<runtime_type> &types) {
    offsets_ret_t r;
    r.offsets_.reserve(types.size());
    for (const runtime_type &t : types) {
      r.offsets_.push_back(r.rowsize_);
      r.rowsize_ += t.size();
      r.maskrowsize_ += t.n();
    }
    return r;
  }

private:
  primitive_type t_ = TYPE_B;
  unsigned psize_ = 0, n_ = 0;
  bool vec_ = false;
};

struct runtime_cast {
  // read the element stored as `t` at px and convert it to T as C++ would implicitly
  template <typename T>
  static T cast(const uint8_t *px, primitive_type t) {
    if (static_type_to_primitive_type<T>::value == t) {
      T v;
      std::memcpy(&v, px, sizeof v);
      return v;
    }
    T out = T();
    detail::with_ctype(t, [&](auto tag) {
      decltype(tag) v;
      std::memcpy(&v, px, sizeof v);
      out = static_cast<T>(v);
    });
    return out;
  }
  template <typename T>
  static void uncast(uint8_t *px, primitive_type t, T value) {
    detail::with_ctype(t, [&](auto tag) {
      const decltype(tag) v = static_cast<decltype(tag)>(value);
      std::memcpy(px, &v, sizeof v);
    });
  }
  static void copy(uint8_t *dst, primitive_type dst_t, const uint8_t *src, primitive_type src_t) {
    detail::with_ctype(src_t, [&](auto tag) {
      decltype(tag) v;
      std::memcpy(&v, src, sizeof v);
      uncast(dst, dst_t, v);
    });
  }
};

// non-owning typed view of one value (scalar or vector) and its optional mask
class value_accessor {
public:
  value_accessor() = default;
  template <typename T>
  value_accessor(const T *data)
      : data_(reinterpret_cast<const uint8_t *>(data)), type_(static_type_to_primitive_type<T>::value) {}
  value_accessor(const uint8_t *data, const bool *mask, const runtime_type &type)
      : data_(data), mask_(mask), type_(type) {}

  const runtime_type &type() const { return type_; }
  unsigned shape() const { return type_.n(); }
  bool ismasked(std::size_t idx) const { return mask_ ? mask_[idx] : false; }
  bool anymasked() const {
    if (mask_)
      for (unsigned i = 0; i < shape(); i++)
        if (mask_[i]) return true;
    return false;
  }
  template <typename T>
  T get(std::size_t idx = 0) const { return runtime_cast::cast<T>(data_ + idx * type_.psize(), type_.t()); }
  const uint8_t *raw() const { return data_; }
  std::string debug_str() const {
    std::string s = "[";
    for (unsigned i = 0; i < shape(); i++)
      s += (i ? ", " : "") + primitive_type_traits::ToString(type_.t(), data_ + i * type_.psize());
    return s + "]";
  }

private:
  const uint8_t *data_ = nullptr;
  const bool *mask_ = nullptr;
  runtime_type type_;
};

class value_mutator {
public:
  value_mutator() = default;
  template <typename T>
  value_mutator(T *data)
      : data_(reinterpret_cast<uint8_t *>(data)), type_(static_type_to_primitive_type<T>::value) {}
  value_mutator(uint8_t *data, const runtime_type &type) : data_(data), type_(type) {}

  const runtime_type &type() const { return type_; }
  unsigned shape() const { return type_.n(); }
  template <typename T>
  void set(T v, std::size_t idx = 0) { runtime_cast::uncast<T>(data_ + idx * type_.psize(), type_.t(), v); }
  value_accessor accessor() const { return value_accessor(data_, nullptr, type_); }

private:
  uint8_t *data_ = nullptr;
  runtime_type type_;
};

typedef std::size_t ident_t;
typedef void *opaque_t;
typedef std::string hyperparam_bag_t;
typedef std::string suffstats_bag_t;
typedef std::string serialized_t;

}  // namespace common
}  // namespace microscopes

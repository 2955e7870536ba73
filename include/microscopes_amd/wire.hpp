// wire.hpp -- minimal proto2 wire-format writer/reader for the string "bags" the plugin
// API exchanges (suff-stats / hyper-parameters, base.hpp:31-33,44-46) and for
// group_manager::serialize (microscopes/io/schema.proto:3-46).  The component-model
// messages are those of the absent `distributions` library's schema, restated from its
// published distributions/io/schema.proto; they cannot be verified in this container
// (SURVEY 8c), so byte-compatibility of those is unpinned.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace microscopes {
namespace wire {

class writer {
public:
  void put_varint_field(unsigned field, uint64_t v) {
    varint(uint64_t(field) << 3 | 0);
    varint(v);
  }
  void put_float_field(unsigned field, float v) {
    varint(uint64_t(field) << 3 | 5);
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    for (int i = 0; i < 4; i++) out_.push_back(char(bits >> (8 * i)));
  }
  void put_bytes_field(unsigned field, const std::string &b) {
    varint(uint64_t(field) << 3 | 2);
    varint(b.size());
    out_ += b;
  }
  const std::string &str() const { return out_; }

private:
  void varint(uint64_t v) {
    while (v >= 0x80) {
      out_.push_back(char(v | 0x80));
      v >>= 7;
    }
    out_.push_back(char(v));
  }
  std::string out_;
};

struct field {
  unsigned number = 0;
  unsigned wire_type = 0;   // 0 varint, 2 length-delimited, 5 fixed32
  uint64_t varint = 0;
  float f32 = 0.f;
  std::string bytes;
};

namespace detail {
inline uint64_t read_varint(const std::string &s, std::size_t &i) {
  uint64_t v = 0;
  for (int shift = 0;; shift += 7) {
    if (i >= s.size() || shift > 63) throw std::runtime_error("malformed varint");
    const uint8_t b = uint8_t(s[i++]);
    v |= uint64_t(b & 0x7f) << shift;
    if (!(b & 0x80)) return v;
  }
}
inline float read_f32(const std::string &s, std::size_t &i) {
  if (i + 4 > s.size()) throw std::runtime_error("truncated fixed32");
  uint32_t bits = 0;
  for (int k = 0; k < 4; k++) bits |= uint32_t(uint8_t(s[i + k])) << (8 * k);
  i += 4;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}
}  // namespace detail

inline std::vector<field> parse(const std::string &s) {
  std::vector<field> out;
  for (std::size_t i = 0; i < s.size();) {
    const uint64_t key = detail::read_varint(s, i);
    field f;
    f.number = unsigned(key >> 3);
    f.wire_type = unsigned(key & 7);
    if (f.wire_type == 0) f.varint = detail::read_varint(s, i);
    else if (f.wire_type == 5) f.f32 = detail::read_f32(s, i);
    else if (f.wire_type == 2) {
      const uint64_t n = detail::read_varint(s, i);
      if (n > s.size() - i) throw std::runtime_error("truncated bytes field");   // (i <= size; no wrap for a huge n)
      f.bytes = s.substr(i, n);
      i += n;
    } else throw std::runtime_error("unsupported wire type");
    out.push_back(f);
  }
  return out;
}

// repeated scalars arrive either one per field or packed in a length-delimited field
inline void collect_floats(const field &f, std::vector<float> &out) {
  if (f.wire_type == 5) out.push_back(f.f32);
  else if (f.wire_type == 2)
    for (std::size_t i = 0; i < f.bytes.size();) out.push_back(detail::read_f32(f.bytes, i));
}
inline void collect_varints(const field &f, std::vector<uint64_t> &out) {
  if (f.wire_type == 0) out.push_back(f.varint);
  else if (f.wire_type == 2)
    for (std::size_t i = 0; i < f.bytes.size();) out.push_back(detail::read_varint(f.bytes, i));
}

}  // namespace wire
}  // namespace microscopes

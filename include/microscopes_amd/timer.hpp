// timer.hpp -- wall-clock microsecond timer with the interface of
// include/microscopes/common/timer.hpp (lap, lap_ms, scoped_timer).
#pragma once

#include <chrono>
#include <cstdint>
#include <iostream>
#include <string>

#define compiler_barrier() asm volatile("" ::: "memory")

namespace microscopes {
namespace common {

class timer {
public:
  timer() { lap(); }
  // microseconds since the previous lap (or construction)
  uint64_t lap() {
    const auto now = std::chrono::steady_clock::now();
    const uint64_t us = uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(now - start_).count());
    start_ = now;
    return us;
  }
  double lap_ms() { return double(lap()) / 1000.0; }
  static uint64_t cur_usec() {
    return uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(
                        std::chrono::steady_clock::now().time_since_epoch()).count());
  }

private:
  std::chrono::steady_clock::time_point start_;
};

class scoped_timer {
public:
  explicit scoped_timer(const char *region, bool enabled = true) : region_(region), enabled_(enabled) {}
  ~scoped_timer() {
    if (enabled_) std::cerr << "timed region " << region_ << " took " << t_.lap_ms() << " ms" << std::endl;
  }

private:
  const char *region_;
  bool enabled_;
  timer t_;
};

}  // namespace common
}  // namespace microscopes

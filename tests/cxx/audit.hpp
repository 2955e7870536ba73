// audit.hpp -- the gates of the C++ boundary tests, stated the way tests/gpu_helpers.py states the Python ones
// (north_star: float log-scores within 1e-6 relative of the double evaluation; integer suff-stats bit-exact):
//   * a score is compared with the double twin FED THE DEVICE'S OWN FLOAT STATE at 1e-6 max(1, |twin|) -- the reference
//     keeps its suff-stats in float too (distributions.hpp:47-56), so a twin that carried doubles through the same
//     updates answers another question;
//   * a float suff-stat field after n per-value updates is compared with the all-double chain at n half-ulps of the field
//     (each update rounds the field once; the per-value kernel computes in double and stores floats);
//   * a sum of D feature scores at 1e-6 * sum_f max(1, |score_f|) (the per-feature tolerances add).
// Every check prints "AUDIT <name> <max error / gate as measured> <gate>"; tests/test_cxx.py folds the lines into the
// session's tolerance audit (profiles/r04_tolerance_audit.json is such a run).
#pragma once
#include <cmath>
#include <cstdio>
#include <map>
#include <string>

namespace audit {

constexpr double kTol = 1e-6;

struct Rec { double err = 0, gate = 0; int checks = 0; };
inline std::map<std::string, Rec> &table() { static std::map<std::string, Rec> t; return t; }

// record |err| against `gate`; false (and a line on stderr) when it is exceeded
inline bool check(const std::string &name, double err, double gate) {
  Rec &r = table()[name];
  r.err = std::fmax(r.err, err);
  r.gate = std::fmax(r.gate, gate);
  r.checks++;
  if (!(err <= gate)) {
    std::fprintf(stderr, "  AUDIT FAIL %s: error %.4g > gate %.4g\n", name.c_str(), err, gate);
    return false;
  }
  return true;
}
// a score against its twin: relative for |twin| >= 1, absolute below
inline bool score(const std::string &name, double got, double want) {
  return check(name, std::fabs(got - want) / std::fmax(1.0, std::fabs(want)), kTol);
}
// a sum of feature scores: `mag` = sum_f max(1, |score_f|) (+ the prior's max(1, |log pseudocount|) when it is in the sum)
inline bool sum(const std::string &name, double got, double want, double mag) {
  return check(name, std::fabs(got - want) / std::fmax(mag, std::fabs(want)), kTol);
}
// a float field after `n` per-value updates against the all-double chain: n half-ulps of the field (2^-24 relative each)
inline bool field(const std::string &name, double got, double want, int n) {
  const double half_ulp = std::ldexp(std::fmax(std::fabs(want), 1.0), -24);
  return check(name, std::fabs(got - want) / (half_ulp * double(n < 1 ? 1 : n)), 1.0);            // (a fraction of the budget)
}
inline void dump() {
  for (const auto &kv : table())
    std::printf("AUDIT %s %.6g %.6g %d\n", kv.first.c_str(), kv.second.err, kv.second.gate, kv.second.checks);
}

}  // namespace audit

// Environment knobs of libsfb.so: A/B experiments, measurements and tests only -- none of them
// changes a result (the parity tests run the variants against each other), several select another kernel or launch
// shape.  Every read goes through knob(): the first time a knob is found SET, the library says so on stderr, so a
// production run cannot be steered silently by a stale environment.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>

namespace sfb {

inline const char *knob(const char *name)
{
  const char *v = std::getenv(name);
  if (v != nullptr) {
    static std::mutex mu;
    static std::set<std::string> seen;
    std::lock_guard<std::mutex> lk(mu);
    if (seen.insert(name).second)
      std::fprintf(stderr, "[sfb] tuning knob %s=%s is set (experiments and tests only; results do not depend on it)\n", name, v);
  }
  return v;
}

}  // namespace sfb

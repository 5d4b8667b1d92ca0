#include "knobs.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace sfb {

namespace {
const char *const kKnobs[] = {"SFB_SP_GRID",  "SFB_SP_SLICE",  "SFB_SP_PAUSE",      "SFB_SP_PREDICT",   "SFB_SP_LAT",
                              "SFB_SP_FORCE_LAT", "SFB_SP_LAT_WAVES", "SFB_SP_POLISHERS", "SFB_SP_LAT_HELP", "SFB_SP_PHASED", "SFB_SP_LEAN_WAVES", "SFB_MID_GRID", "SFB_MID_SLICE",
                              "SFB_QP4_MAX_WAVES", "SFB_QP_DENSE_BIG", "SFB_PLAN_UNITS", "SFB_PLAN_DEBUG", "SFB_MPC_TIMING"};
std::mutex g_mu;
// (node-based: the value strings stay where they are until their knob is set again or cleared)
std::map<std::string, std::string> &table()
{
  static std::map<std::string, std::string> t;
  return t;
}
}  // namespace

const char *knob(const char *name)
{
  std::lock_guard<std::mutex> lk(g_mu);
  const auto &t = table();
  if (t.empty()) return nullptr;
  const auto it = t.find(name);
  return it == t.end() ? nullptr : it->second.c_str();
}

int knob_set(const char *name, const char *value)
{
  if (name == nullptr) return 1;
  bool known = false;
  for (const char *k : kKnobs) known = known || std::strcmp(k, name) == 0;
  if (!known) return 1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (value == nullptr) {
    table().erase(name);
  } else {
    table()[name] = value;
    std::fprintf(stderr, "[sfb] debug knob %s=%s set through sfb_debug_set (tests and measurements only; results do not depend on it)\n",
                 name, value);
  }
  return 0;
}

}  // namespace sfb

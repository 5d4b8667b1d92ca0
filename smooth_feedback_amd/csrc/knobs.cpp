#include "knobs.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>

namespace sfb {

namespace {
const char *const kKnobs[] = {"SFB_SP_GRID",  "SFB_SP_SLICE",  "SFB_SP_PAUSE",      "SFB_SP_PREDICT",   "SFB_SP_LAT",
                              "SFB_SP_FORCE_LAT", "SFB_SP_LAT_WAVES", "SFB_SP_POLISHERS", "SFB_SP_LAT_HELP", "SFB_SP_PHASED", "SFB_SP_LEAN_WAVES", "SFB_MID_GRID", "SFB_MID_SLICE",
                              "SFB_QP4_MAX_WAVES", "SFB_QP_DENSE_BIG", "SFB_EKF_PERSISTENT", "SFB_PLAN_UNITS", "SFB_PLAN_DEBUG", "SFB_MPC_TIMING"};
std::mutex g_mu;
std::atomic<int> g_set{0};  // knobs set right now: a production process never sets one and never takes the lock
// Values are INTERNED: a string handed out by knob() stays valid for the life of the process, whatever a concurrent
// sfb_debug_set does to the knob afterwards (sets are rare: tests and measurements).
std::map<std::string, const char *> &table()
{
  static std::map<std::string, const char *> t;
  return t;
}
const char *intern(const char *value)
{
  static std::deque<std::string> pool;  // (never shrinks; a deque does not move its elements)
  for (const auto &v : pool)
    if (v == value) return v.c_str();
  pool.emplace_back(value);
  return pool.back().c_str();
}
}  // namespace

const char *knob(const char *name)
{
  if (g_set.load(std::memory_order_acquire) == 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  const auto &t = table();
  const auto it = t.find(name);
  return it == t.end() ? nullptr : it->second;
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
    g_set.store((int)table().size(), std::memory_order_release);
  } else {
    table()[name] = intern(value);
    g_set.store((int)table().size(), std::memory_order_release);
    std::fprintf(stderr, "[sfb] debug knob %s=%s set through sfb_debug_set (tests and measurements only; results do not depend on it)\n",
                 name, value);
  }
  return 0;
}

}  // namespace sfb

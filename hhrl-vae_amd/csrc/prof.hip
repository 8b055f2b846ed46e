// Per-launch HIP-event timing behind the C ABI (diagnostics; off by default).
#include "common.h"

#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace hhrl {

struct ProfRec {
  const char* name;
  hipEvent_t beg, end;
  double flops, bytes;
};

static bool g_prof = false;
static std::mutex g_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free;

bool prof_on() { return g_prof; }

// HHRL_PROF_SHAPES=1: launch sites that serve many shapes (the product kernels) report per shape
bool prof_shapes() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("HHRL_PROF_SHAPES");
    on = (e && atoi(e) == 1) ? 1 : 0;
  }
  return g_prof && on == 1;
}

// A name that stays valid for the life of the process (records keep the pointer)
const char* prof_intern(const char* name) {
  static std::set<std::string> names;
  std::lock_guard<std::mutex> lk(g_mu);
  return names.insert(name).first->c_str();
}

static hipEvent_t get_event() {
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes, bool own_launch)
    : slot(-1), st(s), own(own_launch) {
  if (!g_prof) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{name, get_event(), get_event(), flops, bytes};
  if (!r.beg || !r.end) return;
  if (!own) (void)hipEventRecord(r.beg, st);
  slot = (int)g_recs.size();
  g_recs.push_back(r);
}

ProfScope::~ProfScope() {
  if (slot < 0 || own) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[slot].end, st);
}

hipEvent_t ProfScope::beg() const {
  if (slot < 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  return g_recs[slot].beg;
}

hipEvent_t ProfScope::end() const {
  if (slot < 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  return g_recs[slot].end;
}

}  // namespace hhrl

using namespace hhrl;

extern "C" {

int hhrl_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof = on != 0;
  return HHRL_OK;
}

int hhrl_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.end);
    g_free.push_back(r.beg);
    g_free.push_back(r.end);
  }
  g_recs.clear();
  return HHRL_OK;
}

long hhrl_prof_report(char* buf, long cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.end) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.beg, r.end) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1;
    a.ms += ms;
    a.flops += r.flops;
    a.bytes += r.bytes;
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
  }
  if (buf && cap > 0) {
    long n = (long)out.size() < cap - 1 ? (long)out.size() : cap - 1;
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (long)out.size() + 1;
}

}  // extern "C"

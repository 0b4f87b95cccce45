// Opt-in per-kernel timing with HIP events on the launch stream (bench.py's live roofline measurement).
// Disabled by default: the marks are no-ops and the library then holds no mutable state at all.
#include <mutex>
#include <vector>
#include "common.h"
#include "prof.h"
#include "../../include/acvae_hip.h"

namespace {
struct Pair { hipEvent_t a, b; };
bool g_enabled = false;
std::vector<Pair> g_pending[ACVAE_PROF_NTAGS];
std::vector<Pair> g_free;
hipEvent_t g_open[ACVAE_PROF_NTAGS];
bool g_is_open[ACVAE_PROF_NTAGS] = {false};
hipEvent_t g_open_b[ACVAE_PROF_NTAGS];
std::mutex g_mu;   // the encoder backward may run on the autograd thread while the main thread reads / toggles

Pair get_pair() {
  if (!g_free.empty()) { Pair p = g_free.back(); g_free.pop_back(); return p; }
  Pair p;
  (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b);
  return p;
}
}  // namespace

namespace acvae {
void prof_begin(int tag, hipStream_t st) {
  if (!g_enabled || tag < 0 || tag >= ACVAE_PROF_NTAGS) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Pair p = get_pair();
  g_open[tag] = p.a; g_open_b[tag] = p.b; g_is_open[tag] = true;
  (void)hipEventRecord(p.a, st);
}
void prof_end(int tag, hipStream_t st) {
  if (!g_enabled || tag < 0 || tag >= ACVAE_PROF_NTAGS) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_is_open[tag]) return;
  (void)hipEventRecord(g_open_b[tag], st);
  g_pending[tag].push_back(Pair{g_open[tag], g_open_b[tag]});
  g_is_open[tag] = false;
}
}  // namespace acvae

extern "C" int acvae_prof_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = enable != 0;
  if (!g_enabled)
    for (auto& v : g_pending) { for (auto& p : v) g_free.push_back(p); v.clear(); }
  return ACVAE_OK;
}

extern "C" int acvae_prof_pause(int paused) {   // stop / resume marking without dropping what was recorded
  g_enabled = paused == 0;
  return ACVAE_OK;
}

extern "C" int acvae_prof_read(int tag, double* total_ms_host, int64_t* launches_host) {
  if (tag < 0 || tag >= ACVAE_PROF_NTAGS || !total_ms_host || !launches_host) return ACVAE_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  double tot = 0.0;
  for (auto& p : g_pending[tag]) {
    if (hipEventSynchronize(p.b) != hipSuccess) return (int)hipGetLastError();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.a, p.b);
    tot += ms;
    g_free.push_back(p);
  }
  *total_ms_host = tot;
  *launches_host = (int64_t)g_pending[tag].size();
  g_pending[tag].clear();
  return ACVAE_OK;
}

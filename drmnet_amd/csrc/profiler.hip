#include "profiler.h"

#include <vector>

namespace drm {

namespace {
struct Rec {
  hipEvent_t e0, e1;
  int kind;
  double flops, bytes;
};
bool g_on = false;
std::vector<Rec> g_pool;   // event pairs, reused after each collect
size_t g_used = 0;
double g_ms[PROF_KINDS], g_fl[PROF_KINDS], g_by[PROF_KINDS];
int64_t g_n[PROF_KINDS];
}  // namespace

void prof_enable(int on) { g_on = on != 0; }
bool prof_enabled() { return g_on; }

ProfScope::ProfScope(int k, double flops, double bytes, hipStream_t st) : kind(k), slot(-1), s(st) {
  if (!g_on) return;
  if (g_used == g_pool.size()) {
    Rec r{};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    g_pool.push_back(r);
  }
  slot = (int)g_used++;
  g_pool[slot].kind = k;
  g_pool[slot].flops = flops;
  g_pool[slot].bytes = bytes;
  (void)hipEventRecord(g_pool[slot].e0, s);
}

ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_pool[slot].e1, s);
}

void prof_collect(double ms[PROF_KINDS], double flops[PROF_KINDS], double bytes[PROF_KINDS], int64_t launches[PROF_KINDS]) {
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_pool[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
    g_ms[r.kind] += t;
    g_fl[r.kind] += r.flops;
    g_by[r.kind] += r.bytes;
    g_n[r.kind] += 1;
  }
  g_used = 0;
  for (int k = 0; k < PROF_KINDS; ++k) {
    ms[k] = g_ms[k];
    flops[k] = g_fl[k];
    bytes[k] = g_by[k];
    launches[k] = g_n[k];
  }
}

void prof_reset() {
  g_used = 0;
  for (int k = 0; k < PROF_KINDS; ++k) {
    g_ms[k] = g_fl[k] = g_by[k] = 0.0;
    g_n[k] = 0;
  }
}

}  // namespace drm

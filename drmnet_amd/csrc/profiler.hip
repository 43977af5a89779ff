#include "profiler.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <tuple>
#include <vector>

namespace drm {

namespace {
struct Rec {
  hipEvent_t e0, e1;
  int kind;
  double flops, bytes;
  int tag[5];
  const char* variant;
};
struct VarAcc {
  int kind = 0;
  int64_t n = 0;
  double ms = 0, flops = 0, bytes = 0;
};
const char* g_variant = nullptr;
std::map<std::string, VarAcc> g_var;
int g_tag[5] = {0, 0, 0, 0, 0};
int g_on = 0;  // 0 off, 1 every kernel family, 2 the dominant family (fused 3x3 convs) only: least perturbation of a timed region
std::vector<Rec> g_pool;   // event pairs, reused after each collect
size_t g_used = 0;
double g_ms[PROF_KINDS], g_fl[PROF_KINDS], g_by[PROF_KINDS];
int64_t g_n[PROF_KINDS];
}  // namespace

void prof_tag(int n, int h, int w, int cin, int cout) {
  g_tag[0] = n; g_tag[1] = h; g_tag[2] = w; g_tag[3] = cin; g_tag[4] = cout;
}
void prof_variant(const char* name) { g_variant = name; }
void prof_enable(int on) { g_on = on < 0 ? 0 : on; }
bool prof_enabled() { return g_on != 0; }

ProfScope::ProfScope(int k, double flops, double bytes, hipStream_t st) : kind(k), slot(-1), s(st) {
  const char* variant = g_variant;
  g_variant = nullptr;
  if (!g_on || k < 0 || (g_on == 2 && k != PROF_CONV3)) return;  // k < 0: the launch sits inside an enclosing scope
  if (g_used == g_pool.size()) {
    Rec r{};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    g_pool.push_back(r);
  }
  slot = (int)g_used++;
  g_pool[slot].kind = k;
  g_pool[slot].flops = flops;
  g_pool[slot].bytes = bytes;
  for (int i = 0; i < 5; ++i) g_pool[slot].tag[i] = g_tag[i];
  for (int i = 0; i < 5; ++i) g_tag[i] = 0;
  g_pool[slot].variant = variant;
  (void)hipEventRecord(g_pool[slot].e0, s);
}

ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_pool[slot].e1, s);
}

void prof_collect(double ms[PROF_KINDS], double flops[PROF_KINDS], double bytes[PROF_KINDS], int64_t launches[PROF_KINDS]) {
  static const bool dump = getenv("DRM_PROF_DUMP") != nullptr;
  std::map<std::tuple<int, int, int, int, int, int>, std::tuple<double, double, int>> table;
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_pool[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
    g_ms[r.kind] += t;
    g_fl[r.kind] += r.flops;
    g_by[r.kind] += r.bytes;
    g_n[r.kind] += 1;
    if (r.variant) {
      VarAcc& v = g_var[r.variant];
      v.kind = r.kind; v.n += 1; v.ms += t; v.flops += r.flops; v.bytes += r.bytes;
    }
    if (dump) {
      auto& e = table[std::make_tuple(r.kind, r.tag[0], r.tag[1], r.tag[2], r.tag[3], r.tag[4])];
      std::get<0>(e) += t;
      std::get<1>(e) += r.flops;
      std::get<2>(e) += 1;
    }
  }
  if (dump && !table.empty()) {
    std::vector<std::pair<double, std::string>> rows;
    double total = 0;
    for (auto& kv : table) {
      char buf[256];
      const double t = std::get<0>(kv.second), f = std::get<1>(kv.second);
      const int n = std::get<2>(kv.second);
      snprintf(buf, sizeof buf, "kind %d N%-3d %3dx%-3d %4d->%-4d : %4d launches %9.3f ms total %8.4f ms/launch %7.1f TF", std::get<0>(kv.first),
               std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), std::get<4>(kv.first), std::get<5>(kv.first), n, t, t / n,
               t > 0 ? f / t / 1e9 : 0.0);
      rows.emplace_back(t, buf);
      total += t;
    }
    std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.first > b.first; });
    fprintf(stderr, "[drm profile] %zu shapes, %.3f ms total\n", rows.size(), total);
    for (auto& r : rows) fprintf(stderr, "[drm profile] %s\n", r.second.c_str());
  }
  g_used = 0;
  for (int k = 0; k < PROF_KINDS; ++k) {
    ms[k] = g_ms[k];
    flops[k] = g_fl[k];
    bytes[k] = g_by[k];
    launches[k] = g_n[k];
  }
}

size_t prof_variants_text(char* buf, size_t cap) {
  std::string out;
  for (auto& kv : g_var) {
    char line[384];
    snprintf(line, sizeof line, "%s\t%d\t%lld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.kind, (long long)kv.second.n, kv.second.ms, kv.second.flops,
             kv.second.bytes);
    out += line;
  }
  if (buf && cap) {
    const size_t n = std::min(out.size(), cap - 1);
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

void prof_reset() {
  g_var.clear();
  g_used = 0;
  for (int k = 0; k < PROF_KINDS; ++k) {
    g_ms[k] = g_fl[k] = g_by[k] = 0.0;
    g_n[k] = 0;
  }
}

}  // namespace drm

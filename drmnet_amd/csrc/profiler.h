// Optional in-library launch profiler: HIP events recorded on the launch stream around each kernel family,
// so bench.py can report the dominant kernel's average launch duration and achieved FLOP/s measured live in
// the timed region (torch.cuda.Event would only see torch's current stream, and cannot bracket single launches
// issued from C++).  Off by default; when off the cost is one branch per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drm {

enum ProfKind { PROF_CONV3 = 0, PROF_CONV1 = 1, PROF_ATTN = 2, PROF_GNSTATS = 3, PROF_MISC = 4, PROF_KINDS = 5 };

struct ProfScope {
  int kind;
  int slot;
  hipStream_t s;
  ProfScope(int kind, double flops, double bytes, hipStream_t s);
  ~ProfScope();
};

// optional shape tag (N,H,W,Cin,Cout) attached to the next ProfScope; DRM_PROF_DUMP=1 prints a per-shape table at collect
void prof_tag(int n, int h, int w, int cin, int cout);
// optional kernel-variant name (a string with static storage, the name rocprofv3 prints for the instantiation) attached to the next
// ProfScope: per-variant totals next to the per-family ones, so that a measured per-launch figure (HBM traffic of ONE instantiation)
// is compared with the algorithmic bytes of the same launches
void prof_variant(const char* name);
// "name\tkind\tlaunches\tms\tflops\tbytes\n" per variant seen since the last reset (after prof_collect); returns the bytes needed
size_t prof_variants_text(char* buf, size_t cap);
void prof_enable(int on);
bool prof_enabled();
// synchronises the recorded events and accumulates; returns per-kind totals since the last reset
void prof_collect(double ms[PROF_KINDS], double flops[PROF_KINDS], double bytes[PROF_KINDS], int64_t launches[PROF_KINDS]);
void prof_reset();

}  // namespace drm

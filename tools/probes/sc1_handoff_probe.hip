// Probe for the fused split-K hand-off of conv_split2.hip: do agent-scope (sc1) stores + loads, ordered only by s_waitcnt vmcnt(0) and a relaxed
// agent-scope ticket, carry a slab from one workgroup to another (other XCD) without buffer_wbl2 / buffer_inv?  KS workgroups per tile write their
// slab; the last arriver reads all KS slabs and checks them.  The buffers are reused every iteration with new values (a stale line shows up).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/sc1_handoff_probe.hip -o /tmp/sc1_probe && /tmp/sc1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int MODE, bool XSPLIT>  // XSPLIT: the splits of a tile are neighbours in the grid = on DIFFERENT XCDs (else same XCD).  0: sc1 accesses, no fences   1: plain accesses + release / acquire fences (the old form)   2: plain accesses, no fences (must fail)
__global__ __launch_bounds__(256) void handoff(float* slabs, unsigned* ticket, unsigned* errors, int tiles, int ks, int iter, float* dbg) {
  const int tile = XSPLIT ? blockIdx.y : blockIdx.x, split = XSPLIT ? blockIdx.x : blockIdx.y, tid = threadIdx.x;
  __shared__ int s_last;
  float* mine = slabs + ((size_t)split * tiles + tile) * 4096;
  for (int j = 0; j < 4; ++j) {
    const float base = (float)(iter * 7 + tile * 3 + split * 1000 + j);
    f32x4 v = {base, base + 1, base + 2, (float)tid};
    float* p = mine + (j * 256 + tid) * 4;
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");  // (s_nop: the >8-byte store-data hazard the compiler cannot see in asm)
    else *reinterpret_cast<f32x4*>(p) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    s_last = __hip_atomic_fetch_add(ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(iter * ks + ks - 1);
  }
  __syncthreads();
  if (!s_last) return;
  if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  unsigned bad = 0;
  for (int k = 0; k < ks; ++k)
    for (int j = 0; j < 4; ++j) {
      const float* p = slabs + ((size_t)k * tiles + tile) * 4096 + (j * 256 + tid) * 4;
      f32x4 v;
      if (MODE == 0) { asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); }
      else v = *reinterpret_cast<const f32x4*>(p);
      const float base = (float)(iter * 7 + tile * 3 + k * 1000 + j);
      const unsigned b = (v.x != base) + (v.y != base + 1) + (v.z != base + 2) + (v.w != (float)tid);
      if (b && dbg) {
        const unsigned slot = atomicAdd(errors + 1, 1u);
        if (slot < 16) { float* d = dbg + slot * 12; d[0] = iter; d[1] = tile; d[2] = split; d[3] = k; d[4] = j; d[5] = tid; d[6] = v.x; d[7] = v.y; d[8] = v.z; d[9] = v.w; d[10] = base; }
      }
      bad += b;
    }
  if (bad) atomicAdd(errors, bad);
}
template <int MODE, bool XSPLIT>
int run(const char* name, float* slabs, unsigned* ticket, unsigned* errors, int tiles, int ks, int iters) {
  float* dbg; CK(hipMalloc(&dbg, 16 * 12 * 4)); CK(hipMemset(dbg, 0, 16 * 12 * 4));
  CK(hipMemset(ticket, 0, tiles * 4)); CK(hipMemset(errors, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((handoff<MODE, XSPLIT>), XSPLIT ? dim3(ks, tiles) : dim3(tiles, ks), dim3(256), 0, 0, slabs, ticket, errors, tiles, ks, it, dbg);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h = 0; CK(hipMemcpy(&h, errors, 4, hipMemcpyDeviceToHost));
  printf("%-40s %s tiles %d ks %d: %u wrong values in %d launches, %.2f us per launch\n", name, XSPLIT ? "[splits on different XCDs]" : "[splits on one XCD]      ", tiles, ks, h, iters, 1000.0 * ms / iters);
  if (h) {
    std::vector<float> d(16 * 12); CK(hipMemcpy(d.data(), dbg, 16 * 12 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) printf("   iter %g tile %g reader-split %g slab %g j %g tid %g: got %g %g %g %g, base %g\n", d[i*12], d[i*12+1], d[i*12+2], d[i*12+3], d[i*12+4], d[i*12+5], d[i*12+6], d[i*12+7], d[i*12+8], d[i*12+9], d[i*12+10]);
  }
  CK(hipFree(dbg));
  return 0;
}
int main() {
  const int tiles = 256, ks = 4, iters = 400;
  float* slabs; unsigned *ticket, *errors;
  CK(hipMalloc(&slabs, (size_t)ks * tiles * 4096 * 4)); CK(hipMalloc(&ticket, tiles * 4)); CK(hipMalloc(&errors, 8));
  CK(hipMemset(slabs, 0, (size_t)ks * tiles * 4096 * 4));
  for (int rep = 0; rep < 2; ++rep) {
    if (run<0, false>("sc1 stores + loads, no fences", slabs, ticket, errors, tiles, ks, iters)) return 1;
    if (run<0, true>("sc1 stores + loads, no fences", slabs, ticket, errors, tiles, ks, iters)) return 1;
    if (run<1, false>("plain accesses + release/acquire fences", slabs, ticket, errors, tiles, ks, iters)) return 1;
    if (run<1, true>("plain accesses + release/acquire fences", slabs, ticket, errors, tiles, ks, iters)) return 1;
    if (run<2, false>("plain accesses, no fences (control)", slabs, ticket, errors, tiles, ks, iters)) return 1;
    if (run<2, true>("plain accesses, no fences (control)", slabs, ticket, errors, tiles, ks, iters)) return 1;
  }
  return 0;
}

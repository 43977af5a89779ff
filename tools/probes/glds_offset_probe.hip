// Where does `global_load_lds_dwordx4 v, s[..] offset:IMM` put its bytes?  (Does the instruction offset move the LDS destination as well as the
// global source?)  Build: hipcc -O2 --offload-arch=gfx950 tools/probes/glds_offset_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* src, float4* dump) {
  __shared__ float4 lds[512];  // 8 KiB
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds);
  const unsigned voff = threadIdx.x * 16u;
  unsigned keep;
  const unsigned dst = lds0 + 2048u;  // M0 = LDS byte 2048
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) dump[i] = lds[i];
}
int main() {
  float4 *src, *dump;
  hipMalloc(&src, 8192); hipMalloc(&dump, 8192);
  float4 h[512];
  for (int i = 0; i < 512; ++i) h[i] = make_float4((float)i, 0.f, 0.f, 0.f);  // src float4 index
  hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dump);
  hipMemcpy(h, dump, 8192, hipMemcpyDeviceToHost);
  int first = -1, firstval = -1;
  for (int i = 0; i < 512; ++i) if (h[i].x >= 0.f) { first = i; firstval = (int)h[i].x; break; }
  printf("M0 = LDS byte 2048 (float4 128), global offset:1024 (float4 64): first written LDS float4 index %d holds source float4 %d\n", first, firstval);
  printf("=> the instruction offset %s the LDS destination\n", first == 128 ? "does NOT move" : (first == 192 ? "ALSO moves" : "?? moves"));
  return 0;
}

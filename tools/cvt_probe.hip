#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale) {
  const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  s16x2 z = {0, 0};
  s16x2 q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, a, b, scale, false);
  out[2 * threadIdx.x] = (unsigned)p & 0xffff;
  out[2 * threadIdx.x + 1] = (unsigned)(unsigned short)q[0];
}
static float dec(unsigned char c) {
  int s = c >> 7, E = (c >> 3) & 15, M = c & 7;
  float v = E == 0 ? ldexpf(M / 8.f, -6) : ldexpf(1.f + M / 8.f, E - 7);
  if (E == 15 && M == 7) v = NAN;
  return s ? -v : v;
}
int main() {
  float h[16] = {1.0f, -0.3f, 447.f, 449.f, 1000.f, -5000.f, 0.001f, 0.01f, 3.3f, 17.f, 1e-5f, 240.f, 460.f, 480.f, 65504.f, 1e9f};
  float* d; unsigned* o; hipMalloc(&d, 64); hipMalloc(&o, 64);
  hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, scale);
    unsigned r[16]; hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    printf("scale %g\n", scale);
    for (int i = 0; i < 8; ++i)
      printf("  in (%g, %g): cvt_pk -> (%g, %g) [%04x]; cvt_scalef32_pk(scale) -> (%g, %g) [%04x]\n", h[2 * i], h[2 * i + 1], dec(r[2 * i] & 255), dec(r[2 * i] >> 8), r[2 * i],
             dec(r[2 * i + 1] & 255), dec((r[2 * i + 1] >> 8) & 255), r[2 * i + 1]);
  }
  return 0;
}

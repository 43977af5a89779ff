// Sustained-MFMA microbenchmark for gfx950: registers only, no memory traffic.  Prints the achieved TFLOP/s of
// v_mfma_f32_32x32x16_f16 and v_mfma_f32_32x32x2_f32 at 1 / 2 waves per SIMD so the conv kernels can be priced against what
// the chip sustains under its power limit rather than the data-sheet clock.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int ACCS>
__global__ __launch_bounds__(512) void k_f16(float* out, int iters, float seed) {
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(seed + threadIdx.x * 1e-3f);
    b[e] = (_Float16)(seed * 0.5f + e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < ACCS; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACCS>
__global__ __launch_bounds__(512) void k_f32(float* out, int iters, float seed) {
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < ACCS; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Dependency-distance variants of the split-precision inner product: 4 accumulators x 3 MFMAs each, issue order pinned.
// DIST = 1: AAA BBB CCC DDD, 2: ABABAB CDCDCD (what hipcc emits for conv_split2), 4: ABCD ABCD ABCD.
template <int DIST>
__global__ __launch_bounds__(512) void k_dep(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a[3], b[3];
  for (int k = 0; k < 3; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (_Float16)(seed + threadIdx.x * 1e-3f + k);
      b[k][e] = (_Float16)(seed * 0.5f + e - k);
    }
#define MF(i, k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[i], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
  for (int it = 0; it < iters; ++it) {
    if (DIST == 1) {
      MF(0, 0); MF(0, 1); MF(0, 2); MF(1, 0); MF(1, 1); MF(1, 2); MF(2, 0); MF(2, 1); MF(2, 2); MF(3, 0); MF(3, 1); MF(3, 2);
    } else if (DIST == 2) {
      MF(0, 0); MF(1, 0); MF(0, 1); MF(1, 1); MF(0, 2); MF(1, 2); MF(2, 0); MF(3, 0); MF(2, 1); MF(3, 1); MF(2, 2); MF(3, 2);
    } else {
      MF(0, 0); MF(1, 0); MF(2, 0); MF(3, 0); MF(0, 1); MF(1, 1); MF(2, 1); MF(3, 1); MF(0, 2); MF(1, 2); MF(2, 2); MF(3, 2);
    }
  }
#undef MF
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The same 12-MFMA product pattern on operands that look like real data: every lane and register holds different
// pseudo-random fp16 values (uniform in [-2, 2), random mantissas), 8 fragment registers as in the conv kernels.  Constant
// operands toggle almost no datapath bits; this variant shows what the chip sustains under its power limit on real data.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <int MODE>  // 0: random data, fixed registers; 1: random data, operands rotate every iteration
__global__ __launch_bounds__(512) void k_rand(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 fr[8];
  for (int k = 0; k < 8; ++k)
    for (int e = 0; e < 8; ++e) {
      const unsigned hsh = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 64u + k * 8u + e + (unsigned)seed);
      fr[k][e] = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
#define MF(i, ka, kb) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ka], fr[kb], acc[i], 0, 0, 0)
  for (int it = 0; it < iters; ++it) {
    // al/ah = fr[0..3], bh/bl = fr[4..7]
    MF(0, 2, 4); MF(0, 0, 6); MF(0, 0, 4);
    MF(1, 2, 5); MF(1, 0, 7); MF(1, 0, 5);
    MF(2, 3, 4); MF(2, 1, 6); MF(2, 1, 4);
    MF(3, 3, 5); MF(3, 1, 7); MF(3, 1, 5);
    if (MODE == 1) {
      const f16x8 t = fr[0];
#pragma unroll
      for (int k = 0; k < 7; ++k) fr[k] = fr[k + 1];
      fr[7] = t;
    }
  }
#undef MF
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Operand-toggle experiment: the same 12 products per iteration (4 accumulators x {lo.hi, hi.lo, hi.hi}) on random data,
// issued in different orders.  ORDER 0: per accumulator (al,bh)(ah,bl)(ah,bh) -- both operands change at almost every MFMA;
// ORDER 1: B-stationary -- consecutive MFMAs share an operand wherever the dependency structure allows.
template <int ORDER>
__global__ __launch_bounds__(512) void k_order(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 fr[8];  // 0,1 = ah0,ah1; 2,3 = al0,al1; 4,5 = bh0,bh1; 6,7 = bl0,bl1
  for (int k = 0; k < 8; ++k)
    for (int e = 0; e < 8; ++e) {
      const unsigned hsh = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 64u + k * 8u + e + (unsigned)seed);
      fr[k][e] = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
#define MF(i, ka, kb) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ka], fr[kb], acc[i], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {  // acc index = i*2 + c
      MF(0, 2, 4); MF(0, 0, 6); MF(0, 0, 4);
      MF(1, 2, 5); MF(1, 0, 7); MF(1, 0, 5);
      MF(2, 3, 4); MF(2, 1, 6); MF(2, 1, 4);
      MF(3, 3, 5); MF(3, 1, 7); MF(3, 1, 5);
    } else if (ORDER == 1) {
      MF(0, 2, 4); MF(0, 0, 4); MF(2, 1, 4); MF(2, 3, 4);   // bh0 stationary: al0 ah0 ah1 al1
      MF(3, 3, 5); MF(3, 1, 5); MF(1, 0, 5); MF(1, 2, 5);   // bh1 stationary: al1 ah1 ah0 al0
      MF(1, 0, 7); MF(3, 1, 7);                             // bl1: ah0 ah1
      MF(2, 1, 6); MF(0, 0, 6);                             // bl0: ah1 ah0
    } else {  // what hipcc emits for conv_split2 today: pairs share A
      MF(0, 2, 4); MF(1, 2, 5); MF(0, 0, 6); MF(1, 0, 7); MF(0, 0, 4); MF(1, 0, 5);
      MF(2, 3, 4); MF(3, 3, 5); MF(2, 1, 6); MF(3, 1, 7); MF(2, 1, 4); MF(3, 1, 5);
    }
    // fresh-looking data every iteration, as in the conv kernel: rotate the fragment registers
    const f16x8 t = fr[0];
    fr[0] = fr[1]; fr[1] = t;
    const f16x8 t2 = fr[4];
    fr[4] = fr[5]; fr[5] = t2;
  }
#undef MF
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bf16 twin of k_rand<1>: 8 x 8-bit multipliers instead of 11 x 11 -- does the power limit leave it a higher rate?
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k_rand_bf16(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 fr[8];
  for (int k = 0; k < 8; ++k)
    for (int e = 0; e < 8; ++e) {
      const unsigned hsh = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 64u + k * 8u + e + (unsigned)seed);
      fr[k][e] = (__bf16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
#define MF(i, ka, kb) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ka], fr[kb], acc[i], 0, 0, 0)
  for (int it = 0; it < iters; ++it) {
    MF(0, 2, 4); MF(0, 0, 6); MF(0, 0, 4);
    MF(1, 2, 5); MF(1, 0, 7); MF(1, 0, 5);
    MF(2, 3, 4); MF(2, 1, 6); MF(2, 1, 4);
    MF(3, 3, 5); MF(3, 1, 7); MF(3, 1, 5);
    const bf16x8 t = fr[0];
#pragma unroll
    for (int k = 0; k < 7; ++k) fr[k] = fr[k + 1];
    fr[7] = t;
  }
#undef MF
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// 16x16x32 twin of k_rand<1>: the same 64x64 per-wave product (4 A fragments x 4 B fragments, hi and lo planes, three MFMAs
// per 16x16 output block = 48 MFMAs per 32-deep slab) on rotating pseudo-random fp16 operands.  MI355X_MICROARCH.md "DVFS
// give-back" item 7 reports the 16x16x32 bf16 shape holding a higher clock than 32x32x16 at equal cycles per FLOP.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_rand16(float* out, int iters, float seed) {
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  f16x8 fr[16];  // 0-3 ah, 4-7 al, 8-11 bh, 12-15 bl
  for (int k = 0; k < 16; ++k)
    for (int e = 0; e < 8; ++e) {
      const unsigned hsh = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 128u + k * 8u + e + (unsigned)seed);
      fr[k][e] = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[i * 4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[4 + i], fr[8 + c], acc[i * 4 + c], 0, 0, 0);
        acc[i * 4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i], fr[12 + c], acc[i * 4 + c], 0, 0, 0);
        acc[i * 4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i], fr[8 + c], acc[i * 4 + c], 0, 0, 0);
      }
    const f16x8 t = fr[0];
#pragma unroll
    for (int k = 0; k < 15; ++k) fr[k] = fr[k + 1];
    fr[15] = t;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 32x32x16 with the same 64x64x32 slab per iteration (2 slabs x 12 MFMAs) for a like-for-like comparison
__global__ __launch_bounds__(512) void k_rand32(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 fr[16];  // slab s: 8s + {0,1 ah; 2,3 al; 4,5 bh; 6,7 bl}
  for (int k = 0; k < 16; ++k)
    for (int e = 0; e < 8; ++e) {
      const unsigned hsh = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 128u + k * 8u + e + (unsigned)seed);
      fr[k][e] = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[i * 2 + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[8 * s2 + 2 + i], fr[8 * s2 + 4 + c], acc[i * 2 + c], 0, 0, 0);
          acc[i * 2 + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[8 * s2 + i], fr[8 * s2 + 6 + c], acc[i * 2 + c], 0, 0, 0);
          acc[i * 2 + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[8 * s2 + i], fr[8 * s2 + 4 + c], acc[i * 2 + c], 0, 0, 0);
        }
    const f16x8 t = fr[0];
#pragma unroll
    for (int k = 0; k < 15; ++k) fr[k] = fr[k + 1];
    fr[15] = t;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, int blocks, int iters, double flop_per_mfma, int accs, float* out, double secs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters / 10, 1.0f);
  hipDeviceSynchronize();
  // repeat launches for `secs` seconds so the power controller reaches steady state; report the last launch
  double tf = 0, total_ms = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms;
    const double waves = (double)blocks * threads / 64;
    tf = waves * iters * accs * flop_per_mfma / (ms * 1e-3) / 1e12;
  }
  printf("%-28s blocks=%d threads=%d  last launch: %.1f TFLOP/s\n", name, blocks, threads, tf);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  float* out;
  hipMalloc(&out, 4096 * 512 * 4);
  const double F16 = 2.0 * 32 * 32 * 16, F32 = 2.0 * 32 * 32 * 2;
  if (argc > 2 && atoi(argv[2]) == 16) {  // MFMA shape comparison only: 64x64x32 slab per iteration in both shapes
    const double F1616 = 2.0 * 16 * 16 * 32;
    for (int rep = 0; rep < 3; ++rep) {
      run("f16 32x32x16 slab, 2 w/SIMD", k_rand32, 512, 256, 15000, F16, 24, out, secs);
      run("f16 16x16x32 slab, 2 w/SIMD", k_rand16, 512, 256, 15000, F1616, 48, out, secs);
      run("f16 32x32x16 slab, 1 w/SIMD", k_rand32, 256, 256, 30000, F16, 24, out, secs);
      run("f16 16x16x32 slab, 1 w/SIMD", k_rand16, 256, 256, 30000, F1616, 48, out, secs);
    }
    return 0;
  }
  run("f16 32x32x16, 1 wave/SIMD", k_f16<4>, 256, 256, 200000, F16, 4, out, secs);
  run("f16 32x32x16, 2 waves/SIMD", k_f16<4>, 512, 256, 100000, F16, 4, out, secs);
  run("f16 dep distance 1, 1 w/SIMD", k_dep<1>, 256, 256, 60000, F16, 12, out, secs);
  run("f16 dep distance 1, 2 w/SIMD", k_dep<1>, 512, 256, 30000, F16, 12, out, secs);
  run("f16 dep distance 2, 1 w/SIMD", k_dep<2>, 256, 256, 60000, F16, 12, out, secs);
  run("f16 dep distance 2, 2 w/SIMD", k_dep<2>, 512, 256, 30000, F16, 12, out, secs);
  run("f16 dep distance 4, 1 w/SIMD", k_dep<4>, 256, 256, 60000, F16, 12, out, secs);
  run("f16 dep distance 4, 2 w/SIMD", k_dep<4>, 512, 256, 30000, F16, 12, out, secs);
  run("f16 random data, 1 w/SIMD", k_rand<0>, 256, 256, 60000, F16, 12, out, secs);
  run("f16 random data, 2 w/SIMD", k_rand<0>, 512, 256, 30000, F16, 12, out, secs);
  run("f16 random rotating, 2 w/SIMD", k_rand<1>, 512, 256, 30000, F16, 12, out, secs);
  run("order 0 (per-acc)        2 w/SIMD", k_order<0>, 512, 256, 30000, F16, 12, out, secs);
  run("order 1 (B-stationary)   2 w/SIMD", k_order<1>, 512, 256, 30000, F16, 12, out, secs);
  run("order 2 (hipcc today)    2 w/SIMD", k_order<2>, 512, 256, 30000, F16, 12, out, secs);
  run("order 1 (B-stationary)   2 w/SIMD", k_order<1>, 512, 256, 30000, F16, 12, out, secs);
  run("order 0 (per-acc)        2 w/SIMD", k_order<0>, 512, 256, 30000, F16, 12, out, secs);
  run("bf16 random rotating, 2 w/SIMD", k_rand_bf16, 512, 256, 30000, F16, 12, out, secs);
  run("f32 32x32x2, 1 wave/SIMD", k_f32<4>, 256, 256, 100000, F32, 4, out, secs);
  run("f32 32x32x2, 2 waves/SIMD", k_f32<4>, 512, 256, 50000, F32, 4, out, secs);
  return 0;
}

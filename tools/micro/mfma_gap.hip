// Micro-benchmark: what does an MFMA gap cost when it carries the softmax fillers of the attention kernels?
// One wave per SIMD (4 waves per workgroup, one workgroup per CU via a large LDS request), every CU busy; the body is
// a chain of v_mfma_f32_32x32x16_f16 on four accumulators with the chosen fillers (volatile asm, as in
// attention_glob.hip / attention_win.hip) between consecutive MFMAs.  Prints s_memtime ticks per gap.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_gap.hip -o tools/micro/mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_IT 256
#define GAPS 16

__device__ __forceinline__ float fma_at(float s, float c, float a) {
  float t;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(s), "v"(c), "v"(a));
  return t;
}
__device__ __forceinline__ float exp2_at(float t) {
  float d;
  asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(t));
  return d;
}
__device__ __forceinline__ uint32_t cvt_pk_at(float a, float b) {
  uint32_t d;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float max3_at(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// MODE: 0 bare MFMAs | 1 +2 fma | 2 +2 exp | 3 +2 exp 2 fma 1 cvt (the kernels' gap) | 4 = 3 + one ds_read_b128
//       5 +1 exp 2 fma 1 cvt | 6 +4 fma | 7 +2 max3 | 8 = 3 without MFMAs (VALU only) | 9 +1 exp | 10 +3 exp
//       11 = 3 + two ds_read_b128 | 12 = bare + two ds_read_b128
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = seed;
  float e[GAPS][2];
  uint32_t pk[GAPS];
#pragma unroll
  for (int g = 0; g < GAPS; ++g) { e[g][0] = seed * g + lane; e[g][1] = seed - g; pk[g] = 0; }
  const float c = 0.5f, nm = -1.0f;
  f32x4 ld[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  const char* lp = lds + threadIdx.x * 16;
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int g = 0; g < GAPS; ++g) {
      if (MODE != 8) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
      float x0 = e[g][0], x1 = e[g][1];
      if (MODE == 1 || MODE == 6) { x0 = fma_at(x0, c, nm); x1 = fma_at(x1, c, nm); }
      if (MODE == 6) { x0 = fma_at(x0, c, nm); x1 = fma_at(x1, c, nm); }
      if (MODE == 2 || MODE == 10) { x0 = exp2_at(x0); x1 = exp2_at(x1); }
      if (MODE == 10) { e[(g + 1) % GAPS][0] = exp2_at(e[(g + 1) % GAPS][0]); }
      if (MODE == 9) { x0 = exp2_at(x0); }
      if (MODE == 3 || MODE == 4 || MODE == 8 || MODE == 11) {
        // the kernels' software pipeline: fma of step k, exp of step k - 1, cvt of step k - 2 (independent registers)
        const int g1 = (g + GAPS - 1) % GAPS, g2 = (g + GAPS - 2) % GAPS;
        e[g1][0] = exp2_at(e[g1][0]);
        e[g1][1] = exp2_at(e[g1][1]);
        x0 = fma_at(x0, c, nm);
        x1 = fma_at(x1, c, nm);
        pk[g2] = cvt_pk_at(e[g2][0], e[g2][1]);
      }
      if (MODE == 5) {
        const int g1 = (g + GAPS - 1) % GAPS, g2 = (g + GAPS - 2) % GAPS;
        e[g1][0] = exp2_at(e[g1][0]);
        x0 = fma_at(x0, c, nm);
        x1 = fma_at(x1, c, nm);
        pk[g2] = cvt_pk_at(e[g2][0], e[g2][1]);
      }
      if (MODE == 7) { x0 = max3_at(x0, x1, c); x1 = max3_at(x1, x0, nm); }
      e[g][0] = x0;
      e[g][1] = x1;
      if (MODE == 4 || MODE == 11 || MODE == 12) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(lp + (g & 7) * 4096)));
        ld[g & 1] = v;
      }
      if (MODE == 11 || MODE == 12) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(v) : "v"((uint32_t)(uintptr_t)(lp + (g & 7) * 4096)));
        ld[(g + 1) & 1] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4 || MODE == 11 || MODE == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[j][i];
#pragma unroll
  for (int g = 0; g < GAPS; ++g) s += e[g][0] + e[g][1] + (float)pk[g];
  s += ld[0][0] + ld[1][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name) {
  float* d;
  unsigned long long* t;
  hipMalloc(&d, 256 * 256 * 4);
  hipMalloc(&t, 256 * 4 * 8);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 96 * 1024, 0, d, t, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 96 * 1024, 0, d, t, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024];
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 1024; ++i) sum += (double)h[i];
  const double per_gap = sum / 1024 / (N_IT * GAPS);
  printf("%-44s %7.1f ticks per gap   (%.3f ms: %.1f ns per gap)\n", name, per_gap, ms, ms * 1e6 / (N_IT * GAPS));
  hipFree(d);
  hipFree(t);
}
int main() {
  run<0>("bare MFMA 32x32x16 chain");
  run<1>("+ 2 fma");
  run<6>("+ 4 fma");
  run<7>("+ 2 max3");
  run<9>("+ 1 exp");
  run<2>("+ 2 exp");
  run<10>("+ 3 exp");
  run<5>("+ 1 exp 2 fma 1 cvt");
  run<3>("+ 2 exp 2 fma 1 cvt (kernel gap)");
  run<8>("  2 exp 2 fma 1 cvt, no MFMA");
  run<4>("+ 2 exp 2 fma 1 cvt + 1 ds_read_b128");
  run<11>("+ 2 exp 2 fma 1 cvt + 2 ds_read_b128");
  run<12>("+ 2 ds_read_b128 only");
  return 0;
}

// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs transcendentals on gfx950, VALU-only waves
// (2 waves per SIMD, every CU busy).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float seed) {
  float a[8];
  f2 b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = (f2){a[i], a[i] + 0.5f}; }
  const float m = 0.999f, c = 0.001f;
  const f2 m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);                       // 8 independent chains of v_fma_f32
      if (MODE == 1) b[i] = __builtin_elementwise_fma(b[i], m2, c2);          // 8 chains of v_pk_fma_f32 (16 elements)
      if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.0001f);           // v_mul + v_exp
      if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.0f;               // v_rcp + v_add
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, double elems_per_inst, int inst_per_it) {
  float* d;
  hipMalloc(&d, 1024 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, d, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double waves = 1024.0 * 8, insts = waves * N_IT * 8.0 * inst_per_it;
  // per SIMD: 1024 SIMDs, each sees waves/1024 waves' instructions
  const double ns_per_inst_per_simd = ms * 1e6 / (insts / 1024.0);
  printf("%-28s %8.3f ms  %.2f ns per wave-instruction per SIMD (= %.1f cyc at 2.4 GHz), %.1f Gelem/s\n", name, ms,
         ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4, insts * 64 * elems_per_inst / inst_per_it / ms / 1e6);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 1, 1);
  run<1>("v_pk_fma_f32", 2, 1);
  run<2>("v_mul_f32 + v_exp_f32", 1, 2);
  run<3>("v_rcp_f32 + v_add_f32", 1, 2);
  return 0;
}

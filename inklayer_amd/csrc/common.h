// Shared device helpers for the InkLayer gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; all reductions below are wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define INK_OK 0
#define INK_ERR_ARG 1
#define INK_ERR_LAUNCH 2

#define INK_CHECK_ARG(cond)            \
  do {                                 \
    if (!(cond)) return INK_ERR_ARG;   \
  } while (0)

static inline int ink_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? INK_OK : INK_ERR_LAUNCH;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-form GELU, as torch.nn.GELU() default.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e.
// f32-rounding level for GELU's use) with v_rcp / v_exp: branch-free, ~12 VALU ops instead of libm's erff.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = pl * t * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);   // 1 - erf(|z|)
  const float erf_abs = 1.0f - e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin
// dispatch); give every XCD a contiguous chunk of logical tiles so neighbouring
// tiles (sharing an operand panel) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

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
// f32-rounding level for GELU's use) with v_rcp / v_exp: branch-free, 12 VALU + 2 transcendental ops instead of
// libm's erff.  With e = 1 - erf(|x| / sqrt 2):  gelu(x) = max(x, 0) - (|x| / 2) e  for either sign of x, and the
// exponent's log2(e) is folded into the argument: u = |x| sqrt(log2(e) / 2), e = poly(t) t 2^(-u u),
// t = 1 / (1 + p |x| / sqrt 2) = 1 / (1 + (p / sqrt(log2 e)) u).
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = fabsf(x) * 0.84932180028801904272f;                  // sqrt(log2(e) / 2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.27273748087922250f, u, 1.0f));   // 0.3275911 / sqrt(log2 e)
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = pl * t * __builtin_amdgcn_exp2f(-(u * u));           // 1 - erf(|x| / sqrt 2)
  return fmaf(-0.5f * fabsf(x), e, fmaxf(x, 0.0f));
}

// bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin
// dispatch); give every XCD a contiguous chunk of logical tiles so neighbouring
// tiles (sharing an operand panel) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

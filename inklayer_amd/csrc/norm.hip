// Normalisation / conversion kernels (HBM-bound; one wave64 per row, 16-B lanes).
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// Split-f16 operands: an f32 value v travels into an f16 MFMA GEMM as THREE K-segments
//   [ hi | (v - hi) * 64 | hi / 64 ],  hi = f16(v),
// against weights laid out as [ W_hi | W_hi / 64 | (W - W_hi) * 64 ], so the accumulator receives
// hi*W_hi + lo*W_hi + hi*W_lo = v*W to ~2^-21 relative.  The powers of two keep both low parts inside f16's
// NORMAL range for |v|, |W| >= 4e-3 (no reliance on how the MFMA unit treats f16 subnormals).
__device__ __forceinline__ void split_store(f16* o, int C, const f32x4& y) {
  f16x4 hi, lo, hs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (f16)y[e];
    lo[e] = (f16)((y[e] - (float)hi[e]) * 64.0f);
    hs[e] = (f16)((float)hi[e] * 0.015625f);
  }
  *(f16x4*)o = hi;
  *(f16x4*)(o + C) = lo;
  *(f16x4*)(o + 2 * C) = hs;
}

// one wave per output row; row kept in registers (C <= 2048 -> <= 8 float4 per lane)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, const int32_t* __restrict__ gather, int rows_out,
    int C, f16* __restrict__ out_h, float* __restrict__ out_f, int64_t ldo, int act, int split,
    const float* __restrict__ add, int64_t ld_add, const int32_t* __restrict__ add_batch_rows, int rows_per_batch) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_out) return;
  const int nv = C >> 2;
  const int64_t ldf = split ? ldo / 3 : ldo;      // f32 copy next to a split operand: rows of ldo / 3
  const int src = gather ? gather[row] : row;
  if (src < 0) {  // padded token: zeros (the reference pads AFTER the norm)
    for (int v = lane; v < nv; v += 64) {
      if (out_h && split) split_store(out_h + row * ldo + v * 4, C, (f32x4){0.f, 0.f, 0.f, 0.f});
      else if (out_h) *(f16x4*)(out_h + row * ldo + v * 4) = (f16x4){0, 0, 0, 0};
      if (out_f) *(f32x4*)(out_f + row * ldf + v * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const float* xr = x + (int64_t)src * ldx;
  // optional addend row: add[add_batch_rows[b] + i] for row = b * rows_per_batch + i (a per-batch-entry gather of a
  // shared tensor: the mask decoder's "keys + attn_out" where the keys are still one copy per IMAGE, sam.py)
  const float* ar = nullptr;
  if (add) {
    const int b = row / rows_per_batch, i = row - b * rows_per_batch;
    ar = add + ((int64_t)(add_batch_rows ? add_batch_rows[b] : b * rows_per_batch) + i) * ld_add;
  }
  f32x4 r[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    r[j] = v < nv ? *(const f32x4*)(xr + v * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ar && v < nv) r[j] += *(const f32x4*)(ar + v * 4);
    s += (r[j][0] + r[j][1]) + (r[j][2] + r[j][3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    if (v < nv) {
      const f32x4 d = r[j] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    if (v < nv) {
      f32x4 y = (r[j] - mean) * rstd;
      if (gamma) y *= *(const f32x4*)(gamma + v * 4);
      if (beta) y += *(const f32x4*)(beta + v * 4);
      if (act == INK_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = gelu_erf(y[e]);
      }
      if (out_h && split) split_store(out_h + row * ldo + v * 4, C, y);
      else if (out_h) *(f16x4*)(out_h + row * ldo + v * 4) = (f16x4){(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
      if (out_f) *(f32x4*)(out_f + row * ldf + v * 4) = y;
    }
  }
}

// out[r, :] = split3(a[r, :] + b[(r*C + c) % n_b]) for contiguous f32 a [rows, C]; out f16 [rows, 3C]
__global__ __launch_bounds__(256) void add_split_f16_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ b,
                                                            f16* __restrict__ o, int64_t n4, int64_t nb4,
                                                            int C) {
  const int c4n = C >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = *(const f32x4*)(a + i * 4);
    if (b) v += *(const f32x4*)(b + (i % nb4) * 4);
    const int64_t row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    split_store(o + row * 3 * C + c, C, v);
  }
}

__global__ __launch_bounds__(256) void add_cvt_f16_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          f16* __restrict__ o,
                                                          float* __restrict__ of, int64_t n4,
                                                          int64_t nb4) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = *(const f32x4*)(a + i * 4);
    if (b) v += *(const f32x4*)(b + (i % nb4) * 4);
    if (o) *(f16x4*)(o + i * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    if (of) *(f32x4*)(of + i * 4) = v;
  }
}

// f32 rows -> the split-f16 residual stream (hi = f16(x), lo = f16(x - hi)) + the per-chunk (sum, sum of squares)
// partials the folded LayerNorm of the next projection reads (gemm.hip ln_rows_prologue).  One wave per row.
template <int NV>
__global__ __launch_bounds__(256) void hilo_split_stats_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C,
                                                               f16* __restrict__ hi, f16* __restrict__ lo, int64_t ldo,
                                                               float* __restrict__ stats, int chunk) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = C >> 2, parts = C / chunk;
  f32x4 r[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    r[j] = v < nv ? *(const f32x4*)(x + (int64_t)row * ldx + v * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (v < nv) {
      f16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (f16)r[j][e];
        l[e] = (f16)(r[j][e] - (float)h[e]);
      }
      *(f16x4*)(hi + (int64_t)row * ldo + v * 4) = h;
      *(f16x4*)(lo + (int64_t)row * ldo + v * 4) = l;
    }
  }
  for (int p = 0; p < parts; ++p) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c >= p * chunk && c < (p + 1) * chunk) {            // chunk % 4 == 0: a lane's four values share a chunk
        s1 += (r[j][0] + r[j][1]) + (r[j][2] + r[j][3]);
        s2 += (r[j][0] * r[j][0] + r[j][1] * r[j][1]) + (r[j][2] * r[j][2] + r[j][3] * r[j][3]);
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) ((float2*)stats)[(int64_t)row * parts + p] = make_float2(s1, s2);
  }
}

__global__ __launch_bounds__(256) void hilo_join_kernel(const f16* __restrict__ hi, const f16* __restrict__ lo,
                                                        float* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f16x4 h = *(const f16x4*)(hi + i * 4), l = *(const f16x4*)(lo + i * 4);
    *(f32x4*)(out + i * 4) = (f32x4){(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                                     (float)h[3] + (float)l[3]};
  }
}

}  // namespace

extern "C" int ink_hilo_split_stats(const float* x, int64_t ldx, int32_t rows, int32_t C, void* hi_f16, void* lo_f16,
                                    int64_t ldo, float* stats, int32_t chunk, void* stream) {
  INK_CHECK_ARG(x && hi_f16 && lo_f16 && stats && rows > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  INK_CHECK_ARG(chunk > 0 && chunk % 4 == 0 && C % chunk == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= C && ldo >= C);
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nv = (C / 4 + 63) / 64;
#define INK_HL(NV) hipLaunchKernelGGL(hilo_split_stats_kernel<NV>, grid, block, 0, s, x, ldx, rows, C, (f16*)hi_f16, \
                                      (f16*)lo_f16, ldo, stats, chunk)
  switch (nv) {
    case 1: INK_HL(1); break;
    case 2: INK_HL(2); break;
    case 3: INK_HL(3); break;
    case 4: INK_HL(4); break;
    case 5: INK_HL(5); break;
    default: INK_HL(8); break;
  }
#undef INK_HL
  return ink_launch_status();
}

extern "C" int ink_hilo_join(const void* hi_f16, const void* lo_f16, int64_t n, float* out, void* stream) {
  INK_CHECK_ARG(hi_f16 && lo_f16 && out && n > 0 && n % 4 == 0);
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(hilo_join_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)hi_f16,
                     (const f16*)lo_f16, out, n4);
  return ink_launch_status();
}

extern "C" int ink_layernorm_rows(const float* x, int64_t ldx, const float* gamma,
                                  const float* beta, float eps, const int32_t* gather,
                                  int32_t rows_out, int32_t C, void* out_f16, float* out_f32,
                                  int64_t ldo, int32_t act, int32_t split, const float* add, int64_t ld_add,
                                  const int32_t* add_batch_rows, int32_t rows_per_batch, void* stream) {
  INK_CHECK_ARG(x && (out_f16 || out_f32));
  INK_CHECK_ARG(!add || (!gather && rows_per_batch > 0 && ld_add >= C && ld_add % 4 == 0));
  INK_CHECK_ARG(split == 0 || (split == 1 && out_f16 && ldo >= 3 * (int64_t)C && (!out_f32 || ldo % 12 == 0)));
  INK_CHECK_ARG(act == INK_ACT_NONE || act == INK_ACT_GELU);
  INK_CHECK_ARG(rows_out > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  INK_CHECK_ARG(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= C && ldo >= C);
  const dim3 grid((rows_out + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nv = (C / 4 + 63) / 64;
#define INK_LN(NV)                                                                          \
  hipLaunchKernelGGL(layernorm_rows_kernel<NV>, grid, block, 0, s, x, ldx, gamma, beta, eps, \
                     gather, rows_out, C, (f16*)out_f16, out_f32, ldo, act, split, add, ld_add, add_batch_rows, \
                     rows_per_batch)
  switch (nv) {
    case 1: INK_LN(1); break;
    case 2: INK_LN(2); break;
    case 3: INK_LN(3); break;
    case 4: INK_LN(4); break;
    case 5: INK_LN(5); break;
    default: INK_LN(8); break;
  }
#undef INK_LN
  return ink_launch_status();
}

extern "C" int ink_add_cvt_f16(const float* a, const float* b, int64_t n_b, void* out_f16,
                               int64_t n, void* stream) {
  INK_CHECK_ARG(a && out_f16 && n > 0 && n % 4 == 0);
  INK_CHECK_ARG(!b || (n_b > 0 && n_b % 4 == 0 && n % n_b == 0));
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_cvt_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b,
                     (f16*)out_f16, (float*)nullptr, n4, b ? n_b / 4 : 1);
  return ink_launch_status();
}

extern "C" int ink_add_split_f16(const float* a, const float* b, int64_t n_b, void* out_f16, int64_t n,
                                 int32_t C, void* stream) {
  INK_CHECK_ARG(a && out_f16 && n > 0 && C > 0 && C % 4 == 0 && n % C == 0);
  INK_CHECK_ARG(!b || (n_b > 0 && n_b % 4 == 0 && n % n_b == 0));
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add_split_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b,
                     (f16*)out_f16, n4, b ? n_b / 4 : 1, C);
  return ink_launch_status();
}

extern "C" int ink_add_f32(const float* a, const float* b, int64_t n_b, float* out, int64_t n,
                           void* stream) {
  INK_CHECK_ARG(a && out && n > 0 && n % 4 == 0);
  INK_CHECK_ARG(!b || (n_b > 0 && n_b % 4 == 0 && n % n_b == 0));
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_cvt_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b,
                     (f16*)nullptr, out, n4, b ? n_b / 4 : 1);
  return ink_launch_status();
}

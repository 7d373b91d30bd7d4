// Tail of SAM's mask decoder for gfx950, one kernel instead of three passes over per-box tensors:
//     low[b, 4y + .., 4x + ..] = hyper[b, :] . GELU( ConvT2( GELU( LayerNorm2d( u0 ) ) ) )
// = output_upscaling[1..4] + the hyper-network product (SA/modeling/mask_decoder.py:54-60, 138-145) for mask token 0.
// u0 f32 [n*T*4, 64] is the first transposed convolution's output (one row per (box, token, sub-pixel s1)); the second
// transposed convolution (k2 s2) is a [64 -> 4 x 32] projection per row, and the mask logit of sub-pixel s2 is the dot
// product of its 32 channels with the box's hyper-network vector.  As three kernels (LayerNorm + GELU writing split-f16
// operands, the GEMM writing [n*T*16, 32] f32 = 1.07 GB, mask_logits reading it back) this moved 3.2 GB per 128 boxes;
// here u0 is read once (537 MB) and 4 floats per row are written.
//
// One wave = tiles of 32 rows (mfma_f32_32x32x16_f16, swapped form: the lane (m = lane & 31, hh) holds its row's
// channels).  LayerNorm statistics: half a row per lane + one cross-half exchange.  The projection runs on split-f16
// operands like every other layer of the decoder ([hi | lo*64 | hi/64] against
// [W_hi | W_hi/64 | W_lo*64], K' = 192 = 12 k-steps): the B operand is built in registers, the 48 KB weight blob sits
// in LDS as lane-linear 1-KiB MFMA operand blocks (packed at load time).  The kernel is bound by VALU issue: 96 erf-GELUs
// per lane and tile against 48 MFMAs.
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

constexpr int CI = 64, CO = 128, CG = 32;      // channels in, channels out (4 sub-pixels x 32), channels per sub-pixel
constexpr int KS = 3 * CI / 16;                // 12 k-steps of the split operand
constexpr int BLK = 1024;
constexpr int W_BYTES = (CO / 32) * KS * BLK;  // 48 KiB
constexpr int LDS_BYTES = W_BYTES + CO * 4 + 2 * CI * 4;

__device__ __forceinline__ f32x16 mfma32(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ws f16 [128, 192] (ops.split_weight layout) -> blob: block (jt, s) = 64 lanes x 8 halves, lane (l, hh) holds
// ws[32 jt + l][16 s + 8 hh .. + 8]
__global__ __launch_bounds__(256) void upscale_pack_kernel(const f16* __restrict__ ws, f16* __restrict__ blob) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // one 16-B piece
  if (idx >= (CO / 32) * KS * 64) return;
  const int lane = idx & 63, q = idx >> 6, jt = q / KS, s = q % KS;
  *(f16x8*)(blob + (int64_t)idx * 8) = *(const f16x8*)(ws + (int64_t)(32 * jt + (lane & 31)) * (3 * CI) + 16 * s + 8 * (lane >> 5));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void upscale_tail_kernel(const float* __restrict__ u0, int64_t ld_tok, int64_t n_tiles,
                                                           const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                           float eps, const f16* __restrict__ blob,
                                                           const float* __restrict__ b3, const float* __restrict__ hyper,
                                                           int g, float* __restrict__ low) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sB3 = (float*)(smem + W_BYTES);
  float* sG = sB3 + CO;
  float* sBt = sG + CI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = lane & 31, hh = lane >> 5;
  for (int i = tid; i < W_BYTES / 16; i += 256) ((f32x4*)smem)[i] = ((const f32x4*)blob)[i];
  if (tid < CO) sB3[tid] = b3[tid];
  if (tid < CI) { sG[tid] = ln_g[tid]; sBt[tid] = ln_b[tid]; }
  __syncthreads();
  const char* wl = smem + lane * 16;
  const int T = g * g;

  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");      // the 48 weight fragments are re-read from LDS per tile: hoisted out of the loop they take 192 registers
    const int64_t row = t * 32 + l;
    // ---- the lane's half of its row: channels 16 s' + 8 hh + 0..7, s' = 0..3
    const float* xp = u0 + (row >> 2) * ld_tok + (row & 3) * CI + 8 * hh;      // row = (token, sub-pixel s1)
    f32x4 x[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      x[2 * s] = *(const f32x4*)(xp + 16 * s);
      x[2 * s + 1] = *(const f32x4*)(xp + 16 * s + 4);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / CI);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 d = x[i] - mean;
      sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / CI) + eps);
    // ---- LayerNorm2d + GELU -> the three K-segments of the split operand, already in B-operand layout
    f16x8 bf[KS];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int c = 16 * s + 8 * hh + 4 * h2;
        const f32x4 gm = *(const f32x4*)(sG + c), bt = *(const f32x4*)(sBt + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = gelu_erf((x[2 * s + h2][e] - mean) * rstd * gm[e] + bt[e]);
          const f16 hi = (f16)v;
          bf[s][4 * h2 + e] = hi;
          bf[4 + s][4 * h2 + e] = (f16)((v - (float)hi) * 64.0f);
          bf[8 + s][4 * h2 + e] = (f16)((float)hi * 0.015625f);
        }
      }
    // ---- [64 -> 4 x 32] projection: D^T[j, m], tile jt = sub-pixel s2; accumulators start from the bias
    f32x16 acc[CO / 32];
#pragma unroll
    for (int jt = 0; jt < CO / 32; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *(const f32x4*)(sB3 + 32 * jt + 8 * q + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jt][4 * q + e] = b4[e];
      }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int jt = 0; jt < CO / 32; ++jt)
        acc[jt] = mfma32(*(const f16x8*)(wl + (jt * KS + s) * BLK), bf[s], acc[jt]);
    // ---- GELU + hyper-network product: the lane holds channels 8 q + 4 hh + e of every sub-pixel, lane ^ 32 the others
    const int64_t tok = row >> 2;
    const int s1 = (int)(row & 3);
    const int b = (int)(tok / T), yx = (int)(tok - (int64_t)b * T), y = yx / g, xx = yx - y * g;
    const float* hp = hyper + (int64_t)b * CG + 4 * hh;
    f32x4 hv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = *(const f32x4*)(hp + 8 * q);
    float dot[CO / 32];
#pragma unroll
    for (int jt = 0; jt < CO / 32; ++jt) {
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) d = fmaf(hv[q][e], gelu_erf(acc[jt][4 * q + e]), d);
      dot[jt] = d + __shfl_xor(d, 32, 64);
    }
    // lane hh stores sub-pixels s2 = 2 hh, 2 hh + 1 (a horizontal pair): Y = 4y + 2 (s1 >> 1) + (s2 >> 1), X = 4x + 2 (s1 & 1) + (s2 & 1)
    const int Y = 4 * y + 2 * (s1 >> 1) + hh, X = 4 * xx + 2 * (s1 & 1);
    *(f32x2*)(low + ((int64_t)b * 4 * g + Y) * 4 * g + X) = hh ? (f32x2){dot[2], dot[3]} : (f32x2){dot[0], dot[1]};
  }
}

}  // namespace

extern "C" int ink_sam_upscale_pack(const void* ws_f16, void* blob_f16, void* stream) {
  INK_CHECK_ARG(ws_f16 && blob_f16 && ((((uintptr_t)ws_f16 | (uintptr_t)blob_f16) & 15) == 0));
  hipLaunchKernelGGL(upscale_pack_kernel, dim3(((CO / 32) * KS * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)ws_f16, (f16*)blob_f16);
  return ink_launch_status();
}

extern "C" int ink_sam_upscale_tail(const float* u0, int64_t ld_tok, int32_t n, int32_t g, const float* ln_g, const float* ln_b, float eps,
                                    const void* blob_f16, const float* b3, const float* hyper, float* low, void* stream) {
  INK_CHECK_ARG(u0 && ln_g && ln_b && blob_f16 && b3 && hyper && low && n > 0 && g > 0 && (g * g * 4) % 32 == 0);
  INK_CHECK_ARG(ld_tok >= 4 * CI && ld_tok % 4 == 0);
  INK_CHECK_ARG((((uintptr_t)u0 | (uintptr_t)blob_f16 | (uintptr_t)hyper | (uintptr_t)low) & 15) == 0);
  static bool attr = ((void)hipFuncSetAttribute((const void*)upscale_tail_kernel,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), true);
  (void)attr;
  const int64_t n_tiles = (int64_t)n * g * g * 4 / 32;
  const int grid = (int)(n_tiles / 4 < 1024 ? (n_tiles + 3) / 4 : 1024);
  hipLaunchKernelGGL(upscale_tail_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, u0, ld_tok, n_tiles, ln_g, ln_b,
                     eps, (const f16*)blob_f16, b3, hyper, g, low);
  return ink_launch_status();
}

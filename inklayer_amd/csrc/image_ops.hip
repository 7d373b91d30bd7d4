// Pixel-side kernels of the SAM path (HBM-bound gathers / elementwise).
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// Sam.preprocess (normalise + zero-pad to LxL) fused with the 16x16/s16 patch gather of
// PatchEmbed: writes the im2col matrix A[token, c*P*P + ky*P + kx] in f16.
// One thread = 8 consecutive kx of one (token, c, ky) -> one 16-B store; pixel loads are
// 24 contiguous bytes of the HWC row.
__global__ __launch_bounds__(256) void sam_patchify_kernel(const uint8_t* __restrict__ img, int h,
                                                           int w, int L, int P, f32x4 mean_is,
                                                           f32x4 istd, f16* __restrict__ out,
                                                           int chan_reverse, int split) {
  const int g = L / P;
  const int KP = 3 * P * P;
  const int chunks_per_tok = KP / 8;
  const int64_t total = (int64_t)g * g * chunks_per_tok;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int tok = (int)(i / chunks_per_tok);
    const int col = (int)(i % chunks_per_tok) * 8;
    const int c = col / (P * P), ky = (col / P) % P, kx0 = col % P;
    const int y = (tok / g) * P + ky, x0 = (tok % g) * P + kx0;
    const int cs = chan_reverse ? 2 - c : c;
    f16x8 v, lo, hs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int x = x0 + j;
      float f = 0.f;  // padded area is zero AFTER normalisation (SA/modeling/sam.py:167-173)
      if (y < h && x < w) f = ((float)img[((int64_t)y * w + x) * 3 + cs] - mean_is[c]) * istd[c];
      v[j] = (f16)f;
      lo[j] = (f16)((f - (float)v[j]) * 64.0f);      // split-f16 operand segments, see ink_add_split_f16
      hs[j] = (f16)((float)v[j] * 0.015625f);
    }
    if (split) {
      f16* o = out + (int64_t)tok * 3 * KP + col;
      *(f16x8*)o = v;
      *(f16x8*)(o + KP) = lo;
      *(f16x8*)(o + 2 * KP) = hs;
    } else {
      *(f16x8*)(out + (int64_t)tok * KP + col) = v;
    }
  }
}

// 3x3 / pad 1 im2col on an NHWC f16 map: out[b,y,x][(ky*3+kx)*C + c] = in[b,y+ky-1,x+kx-1][c].
__global__ __launch_bounds__(256) void im2col3x3_kernel(const f16* __restrict__ in, int B, int H,
                                                        int W, int C, f16* __restrict__ out) {
  const int cv = C / 8;
  const int64_t total = (int64_t)B * H * W * 9 * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    const int tap = (int)((i / cv) % 9);
    const int64_t pix = i / (9 * cv);
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W)
      v = *(const f16x8*)(in + ((b * H + yy) * W + xx) * C + c8 * 8);
    *(f16x8*)(out + (pix * 9 + tap) * C + c8 * 8) = v;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Pillow's antialiased bilinear resize of 8-bit images, bit for bit (third-party algorithm: Pillow 12.2,
// src/libImaging/Resample.c ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc): every output sample is
//   clip8((2^21 + sum_x in[xmin + x] * k[x]) >> 22)
// with the 22-bit fixed-point coefficients and [xmin, xmin + n) bounds computed on the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do (inklayer_amd/resize.py).  The horizontal pass runs first and its
// result is rounded to u8 before the vertical pass - that intermediate rounding is part of the result.
// Reference call sites: torchvision F.resize on a PIL image in GD/datasets/transforms.py:87-117 (detector, 800
// shorter side) and SA/utils/transforms.py:26-31 (SAM, 1024 longest side).  axis 0: along x, axis 1: along y.
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* __restrict__ in, int in_h, int in_w,
                                                           uint8_t* __restrict__ out, int out_h, int out_w,
                                                           const int32_t* __restrict__ bounds,
                                                           const int32_t* __restrict__ coef, int ksize, int axis) {
  const int64_t total = (int64_t)out_h * out_w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int oy = (int)(i / out_w), ox = (int)(i % out_w);
    const int o = axis == 0 ? ox : oy;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int32_t* k = coef + (int64_t)o * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < n; ++x) {
      const uint8_t* px = axis == 0 ? in + ((int64_t)oy * in_w + lo + x) * 3 : in + ((int64_t)(lo + x) * in_w + ox) * 3;
      const int kk = k[x];
      s0 += px[0] * kk;
      s1 += px[1] * kk;
      s2 += px[2] * kk;
    }
    uint8_t* q = out + i * 3;
    q[0] = (uint8_t)min(max(s0 >> 22, 0), 255);
    q[1] = (uint8_t)min(max(s1 >> 22, 0), 255);
    q[2] = (uint8_t)min(max(s2 >> 22, 0), 255);
  }
}
}  // namespace

extern "C" int ink_sam_patchify(const void* image_u8, int32_t h, int32_t w, int32_t L, int32_t P,
                                const float* mean3, const float* std3, int32_t chan_reverse,
                                int32_t split, void* out_f16, void* stream) {
  INK_CHECK_ARG(image_u8 && out_f16 && mean3 && std3 && (split == 0 || split == 1));
  INK_CHECK_ARG(h > 0 && w > 0 && h <= L && w <= L && P % 8 == 0 && L % P == 0);
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f};
  const f32x4 is = {1.f / std3[0], 1.f / std3[1], 1.f / std3[2], 0.f};
  const int64_t total = (int64_t)(L / P) * (L / P) * (3 * P * P / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sam_patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)image_u8, h, w, L, P, m, is, (f16*)out_f16, chan_reverse, split);
  return ink_launch_status();
}

extern "C" int ink_im2col3x3_f16(const void* in_f16, int32_t B, int32_t H, int32_t W, int32_t C,
                                 void* out_f16, void* stream) {
  INK_CHECK_ARG(in_f16 && out_f16 && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
  const int64_t total = (int64_t)B * H * W * 9 * (C / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)in_f16, B, H, W, C, (f16*)out_f16);
  return ink_launch_status();
}

extern "C" int ink_resize_bilinear_u8(const void* src_u8, int32_t h, int32_t w, void* dst_u8, int32_t oh, int32_t ow,
                                      const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                                      const int32_t* ybounds, const int32_t* ycoef, int32_t ky, void* tmp_u8,
                                      void* stream) {
  INK_CHECK_ARG(src_u8 && dst_u8 && h > 0 && w > 0 && oh > 0 && ow > 0);
  const bool need_h = ow != w, need_v = oh != h;
  INK_CHECK_ARG(need_h || need_v);
  INK_CHECK_ARG(!need_h || (xbounds && xcoef && kx > 0));
  INK_CHECK_ARG(!need_v || (ybounds && ycoef && ky > 0));
  INK_CHECK_ARG(!(need_h && need_v) || tmp_u8);
  hipStream_t s = (hipStream_t)stream;
  auto grid = [](int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); };
  const uint8_t* cur = (const uint8_t*)src_u8;
  if (need_h) {                       // [h, w] -> [h, ow]
    uint8_t* o = need_v ? (uint8_t*)tmp_u8 : (uint8_t*)dst_u8;
    hipLaunchKernelGGL(resize_pass_kernel, dim3(grid((int64_t)h * ow)), dim3(256), 0, s, cur, h, w, o, h, ow, xbounds,
                       xcoef, kx, 0);
    cur = o;
  }
  if (need_v) {                       // [h, ow] -> [oh, ow]
    hipLaunchKernelGGL(resize_pass_kernel, dim3(grid((int64_t)oh * ow)), dim3(256), 0, s, cur, h, ow,
                       (uint8_t*)dst_u8, oh, ow, ybounds, ycoef, ky, 1);
  }
  return ink_launch_status();
}

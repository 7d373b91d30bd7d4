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
                                                           int chan_reverse) {
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
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int x = x0 + j;
      float f = 0.f;  // padded area is zero AFTER normalisation (SA/modeling/sam.py:167-173)
      if (y < h && x < w) f = ((float)img[((int64_t)y * w + x) * 3 + cs] - mean_is[c]) * istd[c];
      v[j] = (f16)f;
    }
    *(f16x8*)(out + (int64_t)tok * KP + col) = v;
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

}  // namespace

extern "C" int ink_sam_patchify(const void* image_u8, int32_t h, int32_t w, int32_t L, int32_t P,
                                const float* mean3, const float* std3, int32_t chan_reverse,
                                void* out_f16, void* stream) {
  INK_CHECK_ARG(image_u8 && out_f16 && mean3 && std3);
  INK_CHECK_ARG(h > 0 && w > 0 && h <= L && w <= L && P % 8 == 0 && L % P == 0);
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f};
  const f32x4 is = {1.f / std3[0], 1.f / std3[1], 1.f / std3[2], 0.f};
  const int64_t total = (int64_t)(L / P) * (L / P) * (3 * P * P / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sam_patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)image_u8, h, w, L, P, m, is, (f16*)out_f16, chan_reverse);
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

// Pixel-side kernels of the Depth-Anything-V2 path: HBM / latency bound, tiny next to the ViT GEMMs.
//   reference: DA/dpt.py:189-221 (image2tensor: cv2.INTER_CUBIC resize, normalise), DA/dinov2_layers/patch_embed.py
//              (14x14 / s14 conv), DA/util/blocks.py + DA/dpt.py:118-150 (bilinear interpolate, align_corners=True;
//              3x3 convs with ReLU'd inputs, a 3x3 / s2 conv)
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// OpenCV interpolateCubic (A = -0.75) at fractional offset fx: the four tap weights
__device__ __forceinline__ void cubic_w(double fx, double (&c)[4]) {
  const double A = -0.75;
  c[0] = ((A * (fx + 1) - 5 * A) * (fx + 1) + 8 * A) * (fx + 1) - 4 * A;
  c[1] = ((A + 2) * fx - (A + 3)) * fx * fx + 1;
  c[2] = ((A + 2) * (1 - fx) - (A + 3)) * (1 - fx) * (1 - fx) + 1;
  c[3] = 1.0 - c[0] - c[1] - c[2];
}

struct Norm3 { double mean[3], std[3]; };

// image2tensor + PatchEmbed gather in one pass: out[token, seg*KP + c*P*P + ky*P + kx] (split-f16 segments
// hi | lo*64 | hi/64, KP = 3*P*P rounded up to a multiple of 32, pad columns zero) of
//   ((cubic_resize(img / 255))[y, x, c] - mean[c]) / std[c],  y = ty*P + ky, x = tx*P + kx,
// cv2.resize INTER_CUBIC semantics: half-pixel centres, 4 taps, replicated border, float64 arithmetic.
// One thread per (token, column).
__global__ __launch_bounds__(256) void depth_patchify_kernel(const uint8_t* __restrict__ img, int H, int W, int nh,
                                                             int nw, int P, int KP, Norm3 nrm,
                                                             int chan_reverse, f16* __restrict__ out) {
  const int gw = nw / P;
  const int64_t total = (int64_t)(nh / P) * gw * KP;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % KP);
  const int64_t tok = i / KP;
  f16* o = out + tok * 3 * KP + col;
  if (col >= 3 * P * P) {
    o[0] = (f16)0.f; o[KP] = (f16)0.f; o[2 * KP] = (f16)0.f;
    return;
  }
  const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
  const int y = (int)(tok / gw) * P + ky, x = (int)(tok % gw) * P + kx;
  const int cs = chan_reverse ? 2 - c : c;
  double v;
  if (nh == H && nw == W) {
    v = (double)img[((int64_t)y * W + x) * 3 + cs] / 255.0;
  } else {
    double fy = (y + 0.5) * ((double)H / nh) - 0.5, fx = (x + 0.5) * ((double)W / nw) - 0.5;
    const int sy = (int)floor(fy), sx = (int)floor(fx);
    double cy[4], cx[4];
    cubic_w(fy - sy, cy);
    cubic_w(fx - sx, cx);
    // horizontal pass first, then vertical (the order of cv2's separable resize)
    v = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(sy - 1 + a, 0), H - 1);
      double r = 0.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(sx - 1 + b, 0), W - 1);
        r += ((double)img[((int64_t)yy * W + xx) * 3 + cs] / 255.0) * cx[b];
      }
      v += r * cy[a];
    }
  }
  const float f = (float)((v - nrm.mean[c]) / nrm.std[c]);
  const f16 hi = (f16)f;
  o[0] = hi;
  o[KP] = (f16)((f - (float)hi) * 64.0f);
  o[2 * KP] = (f16)((float)hi * 0.015625f);
}

// F.interpolate(mode="bilinear", align_corners=True) on NHWC f32 maps; output f32 and/or f16.
// torch: scale = (in - 1) / (out - 1) (0 if out == 1) in float; src = scale * dst; i0 = (int)src; l1 = src - i0.
__global__ __launch_bounds__(256) void resize_bilinear_ac_kernel(const float* __restrict__ in, int B, int h, int w,
                                                                 int C, int H, int W, float* __restrict__ out_f,
                                                                 f16* __restrict__ out_h) {
  const int cv = C >= 4 ? C / 4 : 1;
  const int64_t total = (int64_t)B * H * W * cv;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % cv);
    const int64_t pix = i / cv;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* base = in + b * h * w * C;
    if (C >= 4) {
      const f32x4 v00 = *(const f32x4*)(base + ((int64_t)y0 * w + x0) * C + c4 * 4);
      const f32x4 v01 = *(const f32x4*)(base + ((int64_t)y0 * w + x1) * C + c4 * 4);
      const f32x4 v10 = *(const f32x4*)(base + ((int64_t)y1 * w + x0) * C + c4 * 4);
      const f32x4 v11 = *(const f32x4*)(base + ((int64_t)y1 * w + x1) * C + c4 * 4);
      const f32x4 v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
      if (out_f) *(f32x4*)(out_f + pix * C + c4 * 4) = v;
      if (out_h) *(f16x4*)(out_h + pix * C + c4 * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    } else {
      for (int c = 0; c < C; ++c) {
        const float v00 = base[((int64_t)y0 * w + x0) * C + c], v01 = base[((int64_t)y0 * w + x1) * C + c];
        const float v10 = base[((int64_t)y1 * w + x0) * C + c], v11 = base[((int64_t)y1 * w + x1) * C + c];
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        if (out_f) out_f[pix * C + c] = v;
        if (out_h) out_h[pix * C + c] = (f16)v;
      }
    }
  }
}

// 3x3 / pad 1 im2col on an NHWC f16 map with a stride and an optional ReLU on the way:
// out[b, oy, ox][(ky*3+kx)*C + c] = act(in[b, oy*stride + ky - 1, ox*stride + kx - 1][c]).
__global__ __launch_bounds__(256) void im2col3x3_ex_kernel(const f16* __restrict__ in, int B, int H, int W, int C,
                                                           int stride, int relu, int OH, int OW, f16* __restrict__ out) {
  const int cv = C / 8;
  const int64_t total = (int64_t)B * OH * OW * 9 * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    const int tap = (int)((i / cv) % 9);
    const int64_t pix = i / (9 * cv);
    const int x = (int)(pix % OW), y = (int)((pix / OW) % OH);
    const int64_t b = pix / ((int64_t)OW * OH);
    const int yy = y * stride + tap / 3 - 1, xx = x * stride + tap % 3 - 1;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      v = *(const f16x8*)(in + ((b * H + yy) * W + xx) * C + c8 * 8);
      if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > (f16)0 ? v[j] : (f16)0;
      }
    }
    *(f16x8*)(out + (pix * 9 + tap) * C + c8 * 8) = v;
  }
}

}  // namespace

extern "C" int ink_depth_patchify(const void* image_u8, int32_t H, int32_t W, int32_t nh, int32_t nw, int32_t P,
                                  int32_t KP, const double* mean3, const double* std3, int32_t chan_reverse,
                                  void* out_f16, void* stream) {
  INK_CHECK_ARG(image_u8 && out_f16 && mean3 && std3 && H > 0 && W > 0 && P > 0);
  INK_CHECK_ARG(nh > 0 && nw > 0 && nh % P == 0 && nw % P == 0 && KP >= 3 * P * P && KP % 32 == 0);
  Norm3 nrm;
  for (int c = 0; c < 3; ++c) { nrm.mean[c] = mean3[c]; nrm.std[c] = std3[c]; }
  const int64_t total = (int64_t)(nh / P) * (nw / P) * KP;
  hipLaunchKernelGGL(depth_patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)image_u8, H, W, nh, nw, P, KP, nrm, chan_reverse, (f16*)out_f16);
  return ink_launch_status();
}

extern "C" int ink_resize_bilinear_ac_nhwc(const float* in, int32_t B, int32_t h, int32_t w, int32_t C, int32_t H,
                                           int32_t W, float* out_f32, void* out_f16, void* stream) {
  INK_CHECK_ARG(in && (out_f32 || out_f16) && B > 0 && h > 0 && w > 0 && C > 0 && H > 0 && W > 0);
  INK_CHECK_ARG(C < 4 || C % 4 == 0);
  const int64_t total = (int64_t)B * H * W * (C >= 4 ? C / 4 : 1);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, B, h, w, C, H, W,
                     out_f32, (f16*)out_f16);
  return ink_launch_status();
}

extern "C" int ink_im2col3x3_ex_f16(const void* in_f16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                                    int32_t relu, void* out_f16, void* stream) {
  INK_CHECK_ARG(in_f16 && out_f16 && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (stride == 1 || stride == 2));
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  const int64_t total = (int64_t)B * OH * OW * 9 * (C / 8);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(im2col3x3_ex_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)in_f16, B, H, W,
                     C, stride, relu, OH, OW, (f16*)out_f16);
  return ink_launch_status();
}

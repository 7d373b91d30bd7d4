// SAM prompt-encoder / mask-decoder tail kernels (small, HBM/latency bound).
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// PositionEmbeddingRandom._pe_encoding: out = [sin(2*pi*((2c-1) @ G)), cos(..)]
__global__ __launch_bounds__(256) void pe_encode_kernel(const float* __restrict__ coords,
                                                        const float* __restrict__ G, int N, int F,
                                                        const float* __restrict__ add, int n_add,
                                                        float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * F) return;
  const int n = i / F, f = i % F;
  const float cx = 2.f * coords[2 * n] - 1.f, cy = 2.f * coords[2 * n + 1] - 1.f;
  float v = cx * G[f] + cy * G[F + f];
  v = 6.283185307179586f * v;
  float sv = sinf(v), cv = cosf(v);
  if (add) {
    const float* ar = add + (int64_t)(n % n_add) * 2 * F;
    sv += ar[f];
    cv += ar[F + f];
  }
  out[(int64_t)n * 2 * F + f] = sv;
  out[(int64_t)n * 2 * F + F + f] = cv;
}

// masks[n, 4y+2dy1+dy2, 4x+2dx1+dx2] = hyper[n,:] . up[((n*g*g + y*g + x)*4 + s1)*4 + s2, :]
template <int C>
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ up,
                                                          const float* __restrict__ hyper, int n,
                                                          int g, float* __restrict__ out) {
  const int64_t total = (int64_t)n * g * g * 16;
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= total) return;
  const int s2 = (int)(r & 3), s1 = (int)((r >> 2) & 3);
  const int64_t tok = r >> 4;
  const int x = (int)(tok % g), y = (int)((tok / g) % g);
  const int b = (int)(tok / ((int64_t)g * g));
  const float* u = up + r * C;
  const float* h = hyper + (int64_t)b * C;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const f32x4 uv = *(const f32x4*)(u + c);
    const f32x4 hv = *(const f32x4*)(h + c);
    acc = fmaf(hv[0], uv[0], acc);
    acc = fmaf(hv[1], uv[1], acc);
    acc = fmaf(hv[2], uv[2], acc);
    acc = fmaf(hv[3], uv[3], acc);
  }
  const int Y = 4 * y + 2 * (s1 >> 1) + (s2 >> 1), X = 4 * x + 2 * (s1 & 1) + (s2 & 1);
  out[((int64_t)b * 4 * g + Y) * 4 * g + X] = acc;
}

// torch bilinear (align_corners=False) source index + weights
__device__ __forceinline__ void bil(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 < in_size - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}

// Sam.postprocess_masks + threshold, fused: low [n, S, S] -> (virtual) [L, L] -> crop
// [in_h, in_w] -> [out_h, out_w] -> (> thr) as uint8.  Nothing but the bool mask is written.
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ low, int n, int S,
                                                          int L, int in_h, int in_w, int out_h,
                                                          int out_w, float thr,
                                                          uint8_t* __restrict__ out,
                                                          float* __restrict__ out_logits) {
  const int64_t total = (int64_t)n * out_h * out_w;
  const float sA = (float)S / (float)L;
  const float sBh = (float)in_h / (float)out_h, sBw = (float)in_w / (float)out_w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int X = (int)(i % out_w), Y = (int)((i / out_w) % out_h);
    const int b = (int)(i / ((int64_t)out_w * out_h));
    const float* lp = low + (int64_t)b * S * S;
    int y0, y1, x0, x1;
    float ly, lx;
    bil(Y, sBh, in_h, y0, y1, ly);
    bil(X, sBw, in_w, x0, x1, lx);
    auto stageA = [&](int yy, int xx) {
      int a0, a1, c0, c1;
      float la, lc;
      bil(yy, sA, S, a0, a1, la);
      bil(xx, sA, S, c0, c1, lc);
      const float v00 = lp[a0 * S + c0], v01 = lp[a0 * S + c1];
      const float v10 = lp[a1 * S + c0], v11 = lp[a1 * S + c1];
      return (1.f - la) * ((1.f - lc) * v00 + lc * v01) + la * ((1.f - lc) * v10 + lc * v11);
    };
    const float v00 = stageA(y0, x0), v01 = stageA(y0, x1), v10 = stageA(y1, x0), v11 = stageA(y1, x1);
    const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    if (out) out[i] = v > thr ? 1 : 0;
    if (out_logits) out_logits[i] = v;
  }
}

// Same arithmetic, four consecutive output pixels of one row per thread, workgroups walking ROWS: the mask goes out
// as one packed 4-byte store per lane instead of four one-byte stores (the one-pixel form is bound by store
// INSTRUCTIONS: 2.1 M of them for 128 masks of 1024^2, 710 us), and everything that depends on the column only -
// the second-stage source columns and weights and, for each of them, the first-stage columns and weights - is
// computed once per thread instead of once per pixel (the index/weight arithmetic was ~half of the instructions).
__global__ __launch_bounds__(256) void postprocess_rows_kernel(const float* __restrict__ low, int n, int S, int L,
                                                               int in_h, int in_w, int out_h, int out_w, float thr,
                                                               uint8_t* __restrict__ out,
                                                               float* __restrict__ out_logits) {
  constexpr int PX = 4;
  const float sA = (float)S / (float)L;
  const float sBh = (float)in_h / (float)out_h, sBw = (float)in_w / (float)out_w;
  const int wq = out_w / PX;                         // out_w % 4 == 0 (checked by the launcher)
  for (int xq = threadIdx.x; xq < wq; xq += 256) {
    float lx[PX], lc[PX][2];
    int c0[PX][2], c1[PX][2];
#pragma unroll
    for (int px = 0; px < PX; ++px) {
      int x0, x1;
      bil(xq * PX + px, sBw, in_w, x0, x1, lx[px]);
      bil(x0, sA, S, c0[px][0], c1[px][0], lc[px][0]);
      bil(x1, sA, S, c0[px][1], c1[px][1], lc[px][1]);
    }
    for (int row = blockIdx.x; row < n * out_h; row += gridDim.x) {
      const int b = row / out_h, Y = row - b * out_h;
      const float* lp = low + (int64_t)b * S * S;
      int y0, y1;
      float ly;
      bil(Y, sBh, in_h, y0, y1, ly);
      int a0[2], a1[2];
      float la[2];
      bil(y0, sA, S, a0[0], a1[0], la[0]);
      bil(y1, sA, S, a0[1], a1[1], la[1]);
      const float* r00 = lp + a0[0] * S;
      const float* r01 = lp + a1[0] * S;
      const float* r10 = lp + a0[1] * S;
      const float* r11 = lp + a1[1] * S;
      float vals[PX];
#pragma unroll
      for (int px = 0; px < PX; ++px) {
        auto stageA = [&](const float* ra, const float* rb, float lav, int k) {
          const float v00 = ra[c0[px][k]], v01 = ra[c1[px][k]];
          const float v10 = rb[c0[px][k]], v11 = rb[c1[px][k]];
          return (1.f - lav) * ((1.f - lc[px][k]) * v00 + lc[px][k] * v01) + lav * ((1.f - lc[px][k]) * v10 + lc[px][k] * v11);
        };
        const float v00 = stageA(r00, r01, la[0], 0), v01 = stageA(r00, r01, la[0], 1);
        const float v10 = stageA(r10, r11, la[1], 0), v11 = stageA(r10, r11, la[1], 1);
        vals[px] = (1.f - ly) * ((1.f - lx[px]) * v00 + lx[px] * v01) + ly * ((1.f - lx[px]) * v10 + lx[px] * v11);
      }
      const int64_t o = (int64_t)row * out_w + (int64_t)xq * PX;
      if (out)
        *(uint32_t*)(out + o) = (uint32_t)(vals[0] > thr) | ((uint32_t)(vals[1] > thr) << 8) |
                                ((uint32_t)(vals[2] > thr) << 16) | ((uint32_t)(vals[3] > thr) << 24);
      if (out_logits) *(f32x4*)(out_logits + o) = (f32x4){vals[0], vals[1], vals[2], vals[3]};
    }
  }
}

}  // namespace

extern "C" int ink_sam_pe_encode(const float* coords01, const float* gauss, int32_t N, int32_t F,
                                 const float* add, int32_t n_add, float* out, void* stream) {
  INK_CHECK_ARG(coords01 && gauss && out && N > 0 && F > 0 && (!add || n_add > 0));
  hipLaunchKernelGGL(pe_encode_kernel, dim3((N * F + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     coords01, gauss, N, F, add, n_add, out);
  return ink_launch_status();
}

extern "C" int ink_sam_mask_logits(const float* up, const float* hyper, int32_t n, int32_t g,
                                   int32_t C, float* out, void* stream) {
  INK_CHECK_ARG(up && hyper && out && n > 0 && g > 0);
  const int64_t total = (int64_t)n * g * g * 16;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (C == 32) {
    hipLaunchKernelGGL(mask_logits_kernel<32>, grid, block, 0, (hipStream_t)stream, up, hyper, n, g, out);
  } else if (C == 8) {
    hipLaunchKernelGGL(mask_logits_kernel<8>, grid, block, 0, (hipStream_t)stream, up, hyper, n, g, out);
  } else {
    return INK_ERR_ARG;
  }
  return ink_launch_status();
}

extern "C" int ink_sam_postprocess(const float* low, int32_t n, int32_t S, int32_t L, int32_t in_h,
                                   int32_t in_w, int32_t out_h, int32_t out_w, float thr,
                                   void* out_u8, float* out_logits, void* stream) {
  INK_CHECK_ARG(low && (out_u8 || out_logits) && n > 0 && S > 0 && L >= S);
  INK_CHECK_ARG(in_h > 0 && in_w > 0 && in_h <= L && in_w <= L && out_h > 0 && out_w > 0);
  const int64_t total = (int64_t)n * out_h * out_w;
  if (out_w % 4 == 0 && (int64_t)n * out_h < (int64_t)1 << 30 && ((uintptr_t)out_u8 & 3) == 0 &&
      ((uintptr_t)out_logits & 15) == 0) {
    const int64_t rows = (int64_t)n * out_h;
    const int blocks = (int)(rows < 8192 ? rows : 8192);
    hipLaunchKernelGGL(postprocess_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, low, n, S, L, in_h,
                       in_w, out_h, out_w, thr, (uint8_t*)out_u8, out_logits);
    return ink_launch_status();
  }
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(postprocess_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, low, n, S,
                     L, in_h, in_w, out_h, out_w, thr, (uint8_t*)out_u8, out_logits);
  return ink_launch_status();
}

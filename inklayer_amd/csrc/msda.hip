// Multi-scale deformable attention forward for gfx950 — replaces the reference's only native op
// (GD/models/GroundingDINO/csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:237-299, host
// ms_deform_attn_cuda.cu:21-81).
//
// The reference launches one thread per output ELEMENT (b,q,head,channel) and re-derives the 16
// sample positions per channel.  Here 4 lanes share one (query, head): each lane owns 8 of the 32
// channels (one 16-B f16 / two 16-B f32 gathers per corner), the sample geometry is computed once
// per 4-lane group, and a wave64 covers 2 queries x 8 heads.  Everything is a gather from an
// L2/Infinity-Cache resident value map (13294 x 256 f16 = 6.8 MB per image), so the kernel is
// bound by gather issue + L2 latency, not HBM: high occupancy (few VGPRs), no LDS.
//
// Two entry points:
//   ink_ms_deform_attn_forward  — the reference's argument list (f32 value, int64 shapes, explicit
//                                 sampling locations + weights) for operator-level parity;
//   ink_msda_fused              — what the pipeline uses: f16 value map, raw sampling_offsets /
//                                 attention_weights projections (one f32 GEMM output [.., 384]),
//                                 reference points; softmax over the 16 (level,point) logits and the
//                                 location arithmetic of ms_deform_attn.py:296-322 are done in-kernel
//                                 and the result is written in f16 for the output_proj GEMM.
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

constexpr int MAXL = 8;

struct LevelInfo {
  int H[MAXL], W[MAXL], start[MAXL];
};

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const f16* p, float (&v)[8]) {
  const f16x8 a = *(const f16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}

// bilinear sample of 8 channels with zero padding (ms_deform_attn_im2col_bilinear, :33-84)
template <typename VT>
__device__ __forceinline__ void sample_acc(const VT* __restrict__ vbase, int64_t row_stride, int H, int W,
                                           float h_im, float w_im, float aw, float (&acc)[8]) {
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;
  const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
  const float lh = h_im - (float)h0, lw = w_im - (float)w0;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const float w00 = hh * hw * aw, w01 = hh * lw * aw, w10 = lh * hw * aw, w11 = lh * lw * aw;
  float v[8];
  if (h0 >= 0 && w0 >= 0) {
    load8(vbase + ((int64_t)h0 * W + w0) * row_stride, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(w00, v[i], acc[i]);
  }
  if (h0 >= 0 && w0 + 1 <= W - 1) {
    load8(vbase + ((int64_t)h0 * W + w0 + 1) * row_stride, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(w01, v[i], acc[i]);
  }
  if (h0 + 1 <= H - 1 && w0 >= 0) {
    load8(vbase + ((int64_t)(h0 + 1) * W + w0) * row_stride, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(w10, v[i], acc[i]);
  }
  if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) {
    load8(vbase + ((int64_t)(h0 + 1) * W + w0 + 1) * row_stride, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(w11, v[i], acc[i]);
  }
}

// One sample for the fused kernel, branch-free: every corner is loaded from a clamped (always valid) address and an
// out-of-range corner / sample gets weight 0, so that the four gathers of the sample are in flight together (the
// conditional form above waits for each corner's load inside its own branch).  fma(0, v, acc) == acc for the finite f16
// values of the map and the accumulation order is unchanged.  (Two samples / all 16 gathers of a level in flight were tried:
// 16 / 48 more registers cost occupancy - 245 -> 394 / 370 us.)
__device__ __forceinline__ void sample_acc_bf(const f16* __restrict__ vbase, int64_t row_stride, int H, int W, float h_im,
                                              float w_im, float aw, float (&acc)[8]) {
  const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h0 = in ? (int)hf : 0, w0 = in ? (int)wf : 0;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const bool t = h0 >= 0, b = h0 + 1 <= H - 1, l = w0 >= 0, r = w0 + 1 <= W - 1;
  const float w00 = (in && t && l) ? hh * hw * aw : 0.f, w01 = (in && t && r) ? hh * lw * aw : 0.f;
  const float w10 = (in && b && l) ? lh * hw * aw : 0.f, w11 = (in && b && r) ? lh * lw * aw : 0.f;
  const int ht = max(h0, 0), hb = min(h0 + 1, H - 1), wl = max(w0, 0), wr = min(w0 + 1, W - 1);
  const f16x8 v00 = *(const f16x8*)(vbase + ((int64_t)ht * W + wl) * row_stride);
  const f16x8 v01 = *(const f16x8*)(vbase + ((int64_t)ht * W + wr) * row_stride);
  const f16x8 v10 = *(const f16x8*)(vbase + ((int64_t)hb * W + wl) * row_stride);
  const f16x8 v11 = *(const f16x8*)(vbase + ((int64_t)hb * W + wr) * row_stride);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w00, (float)v00[i], acc[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w01, (float)v01[i], acc[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w10, (float)v10[i], acc[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w11, (float)v11[i], acc[i]);
}

// ---- reference-ABI form: explicit locations + weights, f32 value, C == 32
__global__ __launch_bounds__(256) void msda_ref_kernel(const float* __restrict__ value,
                                                       const float* __restrict__ loc,
                                                       const float* __restrict__ aw, LevelInfo li,
                                                       int B, int S, int M, int Q, int L, int P,
                                                       float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c8 = (int)(gid & 3);                    // which 8-channel slice of the 32
  const int64_t qm = gid >> 2;                      // (b, q, m) flat
  if (qm >= (int64_t)B * Q * M) return;
  const int m = (int)(qm % M);
  const int64_t bq = qm / M;
  const int b = (int)(bq / Q);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float* lp = loc + qm * L * P * 2;
  const float* wp = aw + qm * L * P;
  const int64_t rs = (int64_t)M * 32;
  for (int l = 0; l < L; ++l) {
    const int H = li.H[l], W = li.W[l];
    const float* vb = value + ((int64_t)b * S + li.start[l]) * rs + m * 32 + c8 * 8;
    for (int p = 0; p < P; ++p) {
      const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
      sample_acc<float>(vb, rs, H, W, ly * H - 0.5f, lx * W - 0.5f, wp[l * P + p], acc);
    }
  }
  float* op = out + qm * 32 + c8 * 8;
  *(f32x4*)op = (f32x4){acc[0], acc[1], acc[2], acc[3]};
  *(f32x4*)(op + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
}

// ---- fused form: L == 4, P == 4, M == 8, C == 32 (GroundingDINO_SwinT_OGC.py)
template <int REFD>
__global__ __launch_bounds__(256) void msda_fused_kernel(const f16* __restrict__ value,
                                                         const float* __restrict__ proj, int64_t ldp,
                                                         const float* __restrict__ ref,
                                                         int64_t ref_q_stride, int64_t ref_b_stride,
                                                         LevelInfo li, int B, int S, int Q,
                                                         f16* __restrict__ out) {
  constexpr int M = 8, L = 4, P = 4, C = 32;
  // XCD-contiguous work order: workgroups are dealt round-robin to the 8 XCDs, so without the remap every XCD's
  // 4 MiB L2 would gather from the value maps of ALL images (6.8 MB each); with it XCD x works on one contiguous
  // eighth of the (image, query) space - one image's map at batch 8 - and consecutive queries sample nearby rows
  const int64_t gid = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  const int c8 = (int)(gid & 3);
  const int64_t qm = gid >> 2;
  if (qm >= (int64_t)B * Q * M) return;
  const int m = (int)(qm % M);
  const int64_t bq = qm / M;
  const int b = (int)(bq / Q), q = (int)(bq % Q);
  const float* pr = proj + bq * ldp;
  // sampling_offsets: cols [0, 256) as (m, l, p, xy); attention logits: cols [256, 384) as (m, l, p)
  float off[L * P * 2], lg[L * P];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = *(const f32x4*)(pr + m * 32 + 4 * i);
    off[4 * i] = v[0]; off[4 * i + 1] = v[1]; off[4 * i + 2] = v[2]; off[4 * i + 3] = v[3];
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *(const f32x4*)(pr + 256 + m * 16 + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { lg[4 * i + j] = v[j]; mx = fmaxf(mx, v[j]); }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
  const float inv = 1.f / sum;
  const float* rp = ref + (int64_t)b * ref_b_stride + (int64_t)q * ref_q_stride;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t rs = (int64_t)M * C;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = li.H[l], W = li.W[l];
    const f16* vb = value + ((int64_t)b * S + li.start[l]) * rs + m * C + c8 * 8;
    // reference points are per level in the 2-d (encoder) form, shared across levels in the 4-d form
    float rx, ry, rw = 0.f, rh = 0.f;
    if (REFD == 2) {
      rx = rp[0]; ry = rp[1];
    } else {
      rx = rp[0]; ry = rp[1]; rw = rp[2]; rh = rp[3];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float ox = off[(l * P + p) * 2], oy = off[(l * P + p) * 2 + 1];
      float lx, ly;
      if (REFD == 2) {       // ref + off / (W_l, H_l)              (ms_deform_attn.py:309-314)
        lx = rx + ox / (float)W;
        ly = ry + oy / (float)H;
      } else {               // ref_xy + off / P * ref_wh * 0.5     (ms_deform_attn.py:315-322)
        lx = rx + ox / (float)P * rw * 0.5f;
        ly = ry + oy / (float)P * rh * 0.5f;
      }
      sample_acc_bf(vb, rs, H, W, ly * H - 0.5f, lx * W - 0.5f, lg[l * P + p] * inv, acc);
    }
  }
  f16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (f16)acc[i];
  *(f16x8*)(out + qm * C + c8 * 8) = o;
}

}  // namespace

extern "C" int ink_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes_host,
                                          const int64_t* level_start_index_host,
                                          const float* sampling_loc, const float* attn_weight,
                                          int32_t B, int32_t S, int32_t M, int32_t C, int32_t Q,
                                          int32_t L, int32_t P, int32_t im2col_step, float* out,
                                          void* stream) {
  INK_CHECK_ARG(value && spatial_shapes_host && level_start_index_host && sampling_loc && attn_weight && out);
  INK_CHECK_ARG(B > 0 && S > 0 && M > 0 && Q > 0 && L > 0 && L <= MAXL && P > 0 && C == 32);
  // the reference requires batch % min(batch, im2col_step) == 0 (ms_deform_attn_cuda.cu:51-53)
  const int step = B < im2col_step ? B : im2col_step;
  INK_CHECK_ARG(im2col_step > 0 && B % step == 0);
  LevelInfo li;
  int64_t total = 0;
  for (int l = 0; l < L; ++l) {
    li.H[l] = (int)spatial_shapes_host[2 * l];
    li.W[l] = (int)spatial_shapes_host[2 * l + 1];
    li.start[l] = (int)level_start_index_host[l];
    INK_CHECK_ARG(li.H[l] > 0 && li.W[l] > 0 && li.start[l] == total);
    total += (int64_t)li.H[l] * li.W[l];
  }
  INK_CHECK_ARG(total == S);
  const int64_t threads = (int64_t)B * Q * M * 4;
  hipLaunchKernelGGL(msda_ref_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, value, sampling_loc, attn_weight, li, B, S, M, Q, L, P, out);
  return ink_launch_status();
}

extern "C" int ink_msda_fused(const void* value_f16, const float* proj, int64_t ldp, const float* ref,
                              int32_t ref_dim, int64_t ref_q_stride, int64_t ref_b_stride,
                              const int32_t* shapes_host, int32_t B, int32_t S, int32_t Q,
                              void* out_f16, void* stream) {
  INK_CHECK_ARG(value_f16 && proj && ref && shapes_host && out_f16);
  INK_CHECK_ARG(B > 0 && S > 0 && Q > 0 && ldp >= 384 && ldp % 4 == 0 && (ref_dim == 2 || ref_dim == 4));
  LevelInfo li;
  int total = 0;
  for (int l = 0; l < 4; ++l) {
    li.H[l] = shapes_host[2 * l];
    li.W[l] = shapes_host[2 * l + 1];
    li.start[l] = total;
    INK_CHECK_ARG(li.H[l] > 0 && li.W[l] > 0);
    total += li.H[l] * li.W[l];
  }
  INK_CHECK_ARG(total == S);
  const int64_t threads = (int64_t)B * Q * 8 * 4;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (ref_dim == 2) {
    hipLaunchKernelGGL(msda_fused_kernel<2>, grid, block, 0, s, (const f16*)value_f16, proj, ldp, ref,
                       ref_q_stride, ref_b_stride, li, B, S, Q, (f16*)out_f16);
  } else {
    hipLaunchKernelGGL(msda_fused_kernel<4>, grid, block, 0, s, (const f16*)value_f16, proj, ldp, ref,
                       ref_q_stride, ref_b_stride, li, B, S, Q, (f16*)out_f16);
  }
  return ink_launch_status();
}

// GroundingDINO's image <-> text fusion layer with the 4-token caption folded through it
// (GD/models/GroundingDINO/fuse_modules.py:146-295, BiAttentionBlock / BiMultiHeadAttention; called from
// GD/models/GroundingDINO/transformer.py:535-548 once per encoder layer).
//
// The reference projects every image token to q and value_v (two 256 -> 1024 linears on 13 294 tokens per image), builds
// the [tokens x 4 heads x T] score matrix, takes one softmax over the T text tokens (image update) and one over the image
// tokens (text update), and projects the 1024-wide image result back (1024 -> 256).  InkLayer always prompts with the
// caption "object" (InkLayer/detector/gdino.py:18): T = 4, H T = 16.  With so few text tokens every per-token matrix
// product collapses ALGEBRAICALLY (no approximation) onto 16-wide ones:
//     score[s,h,t] = scale (vn_s Wq_h^T + bq_h) . k[t,h]            = vn_s . U[h,t] + c[h,t],   U[h,t] = scale Wq_h^T k[t,h]
//     out_v[s]     = (sum_{h,t} p_v[s,h,t] vl[t,h]) Wo^T + bo        = sum_{h,t} p_v[s,h,t] Z[h,t] + bo,   Z[h,t] = Wo_h vl[t,h]
//     out_l[t,h]   = sum_s p_l[h,t,s] (vn_s Wvv_h^T + bvv_h)          = Wvv_h m[h,t] + bvv_h,   m[h,t] = sum_s p_l[h,t,s] vn_s
// (vn = LayerNorm_v(v); k, vl = the text projections; p_v = softmax over t, p_l = softmax over s).  So the image side is:
// LayerNorm + a 256 -> 16 product (scores), a 16 -> 256 product (update), and a weighted mean of vn rows per (h, t) - the
// [tokens, 2048] q / value_v projection (435 MB of f16 per layer at batch 8) and the 1024 -> 256 output projection are
// never formed.  Everything is f32 (the old path rounded q, k, values and p to f16).  HBM-bound: the token rows are read
// and written twice per layer (2 x 1 KiB in, 2 x 1 KiB out) instead of ~10 KiB per token.
//   fuse_prep_kernel       U, c, Z from the text projections            (tiny: 16 x 256 x 256 MACs per image)
//   fuse_ln_scores_kernel  vn = LN(v) (in place), scores = vn U^T + c   (one wave per token)
//   biattn_colstats_*      column max / sum-exp over the image tokens   (detector_ops.hip, unchanged)
//   fuse_apply_kernel      v = vn + gamma_v (p_v Z + bo) (in place); partial weighted sums of vn for the text side
//   fuse_text_kernel       m = sum of the partials; out_l = Wvv_h m + bvv_h  -> the text-side output projection input
// The reference's global max subtraction / +-50000 clamps (fuse_modules.py:181-202) are no-ops for a softmax and are not
// reproduced (as before).
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

constexpr int FD = 256;        // d_model of the fusion layer's two inputs
constexpr int FH = 4;          // heads
constexpr int FHD = 256;       // head dim (embed_dim 1024 / 4)
constexpr int FJ = 16;         // H * T with T = 4

// 16 per-lane partial sums -> lane group g = lane >> 2 ends up with the wave total of value g (17 shuffles instead of
// 16 x 6): four halving exchange steps (xor 32, 16, 8, 4), then two plain butterfly steps.
__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane) {
  float a8[8], a4[4], a2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool up = lane & 32;
    const float keep = up ? v[8 + i] : v[i], give = up ? v[i] : v[8 + i];
    a8[i] = keep + __shfl_xor(give, 32, 64);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool up = lane & 16;
    const float keep = up ? a8[4 + i] : a8[i], give = up ? a8[i] : a8[4 + i];
    a4[i] = keep + __shfl_xor(give, 16, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool up = lane & 8;
    const float keep = up ? a4[2 + i] : a4[i], give = up ? a4[i] : a4[2 + i];
    a2[i] = keep + __shfl_xor(give, 8, 64);
  }
  const bool up = lane & 4;
  const float keep = up ? a2[1] : a2[0], give = up ? a2[0] : a2[1];
  float r = keep + __shfl_xor(give, 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;          // value index (lane >> 2) & 15 = (b5 b4 b3 b2)
}

// U[b, j, i], c[b, j], Z[b, j, o] for j = h * 4 + t.  K, VL: f32 [B*T, E] text projections (l_proj, values_l_proj),
// Wq f16 [E, 256] (v_proj weight), bq f32 [E], Wo f16 [256, E] (out_v_proj weight).  grid (16, B), 256 threads.
__global__ __launch_bounds__(256) void fuse_prep_kernel(const float* __restrict__ K, const float* __restrict__ VL,
                                                        int64_t ldk, const f16* __restrict__ Wq,
                                                        const float* __restrict__ bq, const f16* __restrict__ Wo,
                                                        int T, float scale, float* __restrict__ U,
                                                        float* __restrict__ c, float* __restrict__ Z) {
  __shared__ float kk[FHD], vv[FHD], red[4];
  const int j = blockIdx.x, b = blockIdx.y, h = j >> 2, t = j & 3, i = threadIdx.x;
  const bool live = t < T;
  kk[i] = live ? K[((int64_t)b * T + t) * ldk + h * FHD + i] : 0.f;
  vv[i] = live ? VL[((int64_t)b * T + t) * ldk + h * FHD + i] : 0.f;
  __syncthreads();
  float u = 0.f, z = 0.f;
  for (int d = 0; d < FHD; ++d) u = fmaf((float)Wq[(int64_t)(h * FHD + d) * FD + i], kk[d], u);
  const f16* wo = Wo + (int64_t)i * (FH * FHD) + h * FHD;
  for (int d = 0; d < FHD; d += 8) {
    const f16x8 w8 = *(const f16x8*)(wo + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) z = fmaf((float)w8[e], vv[d + e], z);
  }
  U[((int64_t)b * FJ + j) * FD + i] = u * scale;
  Z[((int64_t)b * FJ + j) * FD + i] = z;
  float cb = wave_sum(bq[h * FHD + i] * kk[i]);
  if ((i & 63) == 0) red[i >> 6] = cb;
  __syncthreads();
  // a text token that does not exist (T < 4) gets a score no softmax notices
  if (i == 0) c[b * FJ + j] = live ? (red[0] + red[1] + red[2] + red[3]) * scale : -1.0e30f;
}

// one wave per token: vn = LN(x) * g + be (written over x), scores[s, j] = vn . U[b, j] + c[b, j].
// grid (ceil(S / 4 / TPW), B); every wave keeps ITS four channels of the 16 U rows in registers (64 VGPRs: an LDS copy
// costs 16 ds_read_b128 per token and made the kernel LDS-issue-bound) and walks TPW tokens.
constexpr int TPW = 16;
__global__ __launch_bounds__(256) void fuse_ln_scores_kernel(float* __restrict__ x, int S, const float* __restrict__ g,
                                                             const float* __restrict__ be, float eps,
                                                             const float* __restrict__ U, const float* __restrict__ c,
                                                             float* __restrict__ scores) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 u[FJ];
#pragma unroll
  for (int j = 0; j < FJ; ++j) u[j] = *(const f32x4*)(U + ((int64_t)b * FJ + j) * FD + lane * 4);
  const f32x4 gv = *(const f32x4*)(g + lane * 4), bv = *(const f32x4*)(be + lane * 4);
  const float cj = c[b * FJ + (lane >> 2)];
  const int s0 = (blockIdx.x * 4 + wave) * TPW, s1 = min(S, s0 + TPW);
  if (s0 >= S) return;
  f32x4 nxt = *(const f32x4*)(x + ((int64_t)b * S + s0) * FD + lane * 4);
  for (int s = s0; s < s1; ++s) {
    float* row = x + ((int64_t)b * S + s) * FD + lane * 4;
    f32x4 v = nxt;
    if (s + 1 < s1) nxt = *(const f32x4*)(row + FD);          // the next token's row is in flight during this one's math
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / FD);
    const f32x4 d = v - mean;
    const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / FD);
    const float rstd = 1.0f / sqrtf(var + eps);
    v = d * rstd * gv + bv;
    *(f32x4*)row = v;
    float part[16];
#pragma unroll
    for (int j = 0; j < FJ; ++j) part[j] = (v[0] * u[j][0] + v[1] * u[j][1]) + (v[2] * u[j][2] + v[3] * u[j][3]);
    const float sc = wave_reduce16(part, lane) + cj;
    if ((lane & 3) == 0) scores[((int64_t)b * S + s) * FJ + (lane >> 2)] = sc;
  }
}

// one wave per token: p_v = softmax_t(scores[s, h, :]);  x[s] = vn[s] + gamma * (p_v Z[b] + bo)  (in place);
// text side: w[j] = exp(scores[s, j] - max_j) (normalised later), partial[b, chunk, j, :] = sum over the workgroup's
// tokens of w[j] vn[s].  grid (nchunk, B): the workgroup owns tokens [chunk * TOK, (chunk + 1) * TOK).  Z rows live in
// registers (the lane's four channels of the 16 rows), the 16 probabilities / weights of a token reach every lane as
// scalars (v_readlane of the lane group that holds them) - no LDS in the token loop.
__global__ __launch_bounds__(256) void fuse_apply_kernel(float* __restrict__ x, int S, int tok_per_wg,
                                                         const float* __restrict__ scores,
                                                         const float* __restrict__ stats, const float* __restrict__ Z,
                                                         const float* __restrict__ gamma, const float* __restrict__ bo,
                                                         float* __restrict__ partial, const float* __restrict__ pos,
                                                         f16* __restrict__ out16_pos, f16* __restrict__ out16) {
  __shared__ __attribute__((aligned(16))) float red[FJ * FD];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 z[FJ];
#pragma unroll
  for (int q = 0; q < FJ; ++q) z[q] = *(const f32x4*)(Z + ((int64_t)b * FJ + q) * FD + lane * 4);
  const f32x4 gv = *(const f32x4*)(gamma + lane * 4), ov = *(const f32x4*)(bo + lane * 4);
  const int j = lane >> 2;
  const float cmax = stats[((int64_t)b * FJ + j) * 2];
  f32x4 acc[FJ];
#pragma unroll
  for (int q = 0; q < FJ; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int s_lo = chunk * tok_per_wg, s_hi = min(S, s_lo + tok_per_wg);
  float sc_n = 0.f;
  f32x4 v_n = {0.f, 0.f, 0.f, 0.f};
  if (s_lo + wave < s_hi) {
    sc_n = scores[((int64_t)b * S + s_lo + wave) * FJ + j];
    v_n = *(const f32x4*)(x + ((int64_t)b * S + s_lo + wave) * FD + lane * 4);
  }
  for (int s = s_lo + wave; s < s_hi; s += 4) {
    const float sc = sc_n;
    const f32x4 v = v_n;
    if (s + 4 < s_hi) {                                        // next token of this wave: in flight during the math below
      sc_n = scores[((int64_t)b * S + s + 4) * FJ + j];
      v_n = *(const f32x4*)(x + ((int64_t)b * S + s + 4) * FD + lane * 4);
    }
    // softmax over the 4 text tokens of head j >> 2: the lanes that differ in bits 2, 3
    float mx = fmaxf(sc, __shfl_xor(sc, 4, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
    const float e = expf(sc - mx);
    float sum = e + __shfl_xor(e, 4, 64);
    sum += __shfl_xor(sum, 8, 64);
    const float pv = e / sum, wl = expf(sc - cmax);
    float* row = x + ((int64_t)b * S + s) * FD + lane * 4;
    f32x4 o = ov;
#pragma unroll
    for (int q = 0; q < FJ; ++q) {
      const float p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv), 4 * q));
      const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), 4 * q));
      o += p * z[q];
      acc[q] += w * v;
    }
    const f32x4 nv = v + gv * o;
    *(f32x4*)row = nv;
    // the f16 operands of the deformable attention's two projections (query = x + pos, value = x), written here instead
    // of by two add_cvt passes over the tokens
    if (out16) *(f16x4*)(out16 + ((int64_t)b * S + s) * FD + lane * 4) = (f16x4){(f16)nv[0], (f16)nv[1], (f16)nv[2], (f16)nv[3]};
    if (out16_pos) {
      const f32x4 pv4 = nv + *(const f32x4*)(pos + (int64_t)s * FD + lane * 4);
      *(f16x4*)(out16_pos + ((int64_t)b * S + s) * FD + lane * 4) = (f16x4){(f16)pv4[0], (f16)pv4[1], (f16)pv4[2], (f16)pv4[3]};
    }
  }
  // the four waves' accumulators -> one partial per workgroup
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int q = 0; q < FJ; ++q) {
        f32x4* dst = (f32x4*)(red + q * FD + lane * 4);
        *dst = wv == 0 ? acc[q] : *dst + acc[q];
      }
    }
    __syncthreads();
  }
  float* out = partial + ((int64_t)b * nchunk + chunk) * FJ * FD;
  for (int i = threadIdx.x; i < FJ * FD / 4; i += 256) ((f32x4*)out)[i] = ((const f32x4*)red)[i];
}

// m[b, j, :] = sum_chunk partial / sumexp_j;  out_l[b*T + t, h*256 + e] = Wvv[h*256 + e, :] . m[b, j, :] + bvv[h*256 + e]
// (f16, the operand of the text-side output projection).  grid (16, B), 1024 threads: four thread groups sum every
// fourth chunk each (the sum over ~100 chunks was one dependent chain per thread), combined in a fixed order.
__global__ __launch_bounds__(1024) void fuse_text_kernel(const float* __restrict__ partial, int nchunk,
                                                         const float* __restrict__ stats, const f16* __restrict__ Wvv,
                                                         const float* __restrict__ bvv, int T, f16* __restrict__ out_l) {
  __shared__ __attribute__((aligned(16))) float part[4][FD];
  __shared__ __attribute__((aligned(16))) float m[FD];
  const int j = blockIdx.x, b = blockIdx.y, h = j >> 2, t = j & 3, i = threadIdx.x & 255, grp = threadIdx.x >> 8;
  if (t >= T) return;
  float s0 = 0.f, s1 = 0.f;
  int c = grp;
  for (; c + 4 < nchunk; c += 8) {
    s0 += partial[(((int64_t)b * nchunk + c) * FJ + j) * FD + i];
    s1 += partial[(((int64_t)b * nchunk + c + 4) * FJ + j) * FD + i];
  }
  if (c < nchunk) s0 += partial[(((int64_t)b * nchunk + c) * FJ + j) * FD + i];
  part[grp][i] = s0 + s1;
  __syncthreads();
  if (grp == 0) m[i] = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) / stats[((int64_t)b * FJ + j) * 2 + 1];
  __syncthreads();
  if (grp != 0) return;
  const f16* wr = Wvv + (int64_t)(h * FHD + i) * FD;
  float o = bvv[h * FHD + i];
  for (int d = 0; d < FD; d += 8) {
    const f16x8 w8 = *(const f16x8*)(wr + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o = fmaf((float)w8[e], m[d + e], o);
  }
  out_l[((int64_t)b * T + t) * (FH * FHD) + h * FHD + i] = (f16)o;
}

}  // namespace

extern "C" int ink_fusion_fold_workspace(int32_t B, int32_t S, int64_t* out_floats) {
  INK_CHECK_ARG(out_floats && B > 0 && S > 0);
  const int64_t nchunk = (S + 127) / 128;
  // U, Z [B,16,256] | c [B,16] | scores [B,S,16] | stats [B,16,2] | column-stat partials [B,64,16,2] | partial [B,nchunk,16,256]
  *out_floats = 2 * (int64_t)B * FJ * FD + (int64_t)B * FJ + (int64_t)B * S * FJ + (int64_t)B * FJ * 2 +
                (int64_t)B * 64 * FJ * 2 + (int64_t)B * nchunk * FJ * FD;
  return INK_OK;
}

// forward declarations of the column-statistics passes shared with the unfolded path (detector_ops.hip)
extern "C" int ink_biattn_colstats(const float* scores, int32_t B, int32_t S, int32_t HT, float* part_ws, float* stats,
                                   void* stream);

extern "C" int ink_fusion_fold(float* v_f32, int32_t B, int32_t S, const float* lnv_g, const float* lnv_b, float eps,
                               const float* text_k_f32, const float* text_vl_f32, int64_t ld_text, int32_t T,
                               const void* Wq_f16, const float* bq, const void* Wvv_f16, const float* bvv,
                               const void* Wo_f16, const float* bo, const float* gamma_v, float scale, float* ws,
                               void* out_l_f16, const float* pos, void* out16_pos, void* out16, void* stream) {
  INK_CHECK_ARG(v_f32 && lnv_g && lnv_b && text_k_f32 && text_vl_f32 && Wq_f16 && bq && Wvv_f16 && bvv && Wo_f16 && bo &&
                gamma_v && ws && out_l_f16);
  INK_CHECK_ARG(B > 0 && S > 0 && S <= 64 * 512 && T >= 1 && T <= 4 && ld_text >= FH * FHD);
  INK_CHECK_ARG(!out16_pos || pos);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (S + 127) / 128;
  float* U = ws;
  float* Z = U + (int64_t)B * FJ * FD;
  float* c = Z + (int64_t)B * FJ * FD;
  float* scores = c + (int64_t)B * FJ;
  float* stats = scores + (int64_t)B * S * FJ;
  float* cpart = stats + (int64_t)B * FJ * 2;
  float* partial = cpart + (int64_t)B * 64 * FJ * 2;
  hipLaunchKernelGGL(fuse_prep_kernel, dim3(FJ, B), dim3(256), 0, s, text_k_f32, text_vl_f32, ld_text,
                     (const f16*)Wq_f16, bq, (const f16*)Wo_f16, T, scale, U, c, Z);
  hipLaunchKernelGGL(fuse_ln_scores_kernel, dim3((S + 4 * TPW - 1) / (4 * TPW), B), dim3(256), 0, s, v_f32, S, lnv_g,
                     lnv_b, eps, (const float*)U, (const float*)c, scores);
  if (ink_biattn_colstats(scores, B, S, FJ, cpart, stats, stream) != INK_OK) return INK_ERR_LAUNCH;
  hipLaunchKernelGGL(fuse_apply_kernel, dim3(nchunk, B), dim3(256), 0, s, v_f32, S, 128, (const float*)scores,
                     (const float*)stats, (const float*)Z, gamma_v, bo, partial, pos, (f16*)out16_pos, (f16*)out16);
  hipLaunchKernelGGL(fuse_text_kernel, dim3(FJ, B), dim3(1024), 0, s, (const float*)partial, nchunk, (const float*)stats,
                     (const f16*)Wvv_f16, bvv, T, (f16*)out_l_f16);
  return ink_launch_status();
}

// Small GroundingDINO-side kernels (HBM / latency bound): Swin patch gather, patch-merge LayerNorm,
// GroupNorm, bi-directional image<->text fusion attention, few-key attention, top-k query selection,
// sine position embedding of boxes, iterative box refinement.
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// load_image normalisation + 4x4/s4 PatchEmbed gather: u8 HWC -> f16 [tokens, 64]
// (48 real columns c*16 + ky*4 + kx, 16 zero columns so that K % 32 == 0)
__global__ __launch_bounds__(256) void swin_patchify_kernel(const uint8_t* __restrict__ img, int h, int w,
                                                            int gh, int gw, f32x4 mean, f32x4 stdv,
                                                            f16* __restrict__ out) {
  const int tok = blockIdx.x * 256 + threadIdx.x;
  if (tok >= gh * gw) return;
  const int ty = tok / gw, tx = tok % gw;
  f16 v[64];
#pragma unroll
  for (int i = 48; i < 64; ++i) v[i] = (f16)0;
#pragma unroll
  for (int ky = 0; ky < 4; ++ky) {
    const int y = ty * 4 + ky;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const int x = tx * 4 + kx;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float f = 0.f;  // PatchEmbed pads the NORMALISED image with zeros (swin_transformer.py:483-487)
        if (y < h && x < w) f = ((float)img[((int64_t)y * w + x) * 3 + c] / 255.0f - mean[c]) / stdv[c];
        v[c * 16 + ky * 4 + kx] = (f16)f;
      }
    }
  }
  f16x8* o = (f16x8*)(out + (int64_t)tok * 64);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    o[i] = (f16x8){v[8 * i], v[8 * i + 1], v[8 * i + 2], v[8 * i + 3], v[8 * i + 4], v[8 * i + 5], v[8 * i + 6], v[8 * i + 7]};
}

// ---------------------------------------------------------------------------------------------
// PatchMerging: out[r] = LN(concat(x[g[r][0]], x[g[r][1]], x[g[r][2]], x[g[r][3]])) in f16; -1 = zero row.
// One wave per output row; 4C <= 4096.
template <int NV>
__global__ __launch_bounds__(256) void ln_merge4_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const int32_t* __restrict__ g4, int rows, int C,
                                                        f16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cv = C >> 2, nv = cv * 4;
  f32x4 r[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    r[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (v < nv) {
      const int src = g4[row * 4 + v / cv];
      if (src >= 0) r[j] = *(const f32x4*)(x + (int64_t)src * ldx + (v % cv) * 4);
    }
    s += (r[j][0] + r[j][1]) + (r[j][2] + r[j][3]);
  }
  const float n = (float)(4 * C);
  const float mean = wave_sum(s) / n;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (lane + 64 * j < nv) {
      const f32x4 d = r[j] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / n + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = lane + 64 * j;
    if (v < nv) {
      f32x4 y = (r[j] - mean) * rstd * *(const f32x4*)(gamma + v * 4) + *(const f32x4*)(beta + v * 4);
      *(f16x4*)(out + (int64_t)row * 4 * C + v * 4) = (f16x4){(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(G, C) on NHWC tokens [B, T, C]: stats over (T x C/G) per (b, group).
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, int T, int C,
                                                              int G, float eps, float* __restrict__ stats) {
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;                     // 8 for GroupNorm(32, 256)
  const float* xb = x + (int64_t)b * T * C + g * cg;
  __shared__ float red[4];
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256)
    for (int c = 0; c < cg; c += 4) {
      const f32x4 v = *(const f32x4*)(xb + (int64_t)t * C + c);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  const float n = (float)T * cg;
  const float mean = block_sum(s) / n;
  float q = 0.f;
  for (int t = threadIdx.x; t < T; t += 256)
    for (int c = 0; c < cg; c += 4) {
      const f32x4 d = *(const f32x4*)(xb + (int64_t)t * C + c) - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  const float var = block_sum(q) / n;
  if (threadIdx.x == 0) {
    stats[blockIdx.x * 2] = mean;
    stats[blockIdx.x * 2 + 1] = 1.0f / sqrtf(var + eps);
  }
}
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, int T, int C,
                                                              int G, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ out, int64_t out_bstride,
                                                              int B) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * T * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % cv);
    const int64_t bt = i / cv;
    const int b = (int)(bt / T), t = (int)(bt % T);
    const int g = (c4 * 4) / (C / G);
    const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
    const f32x4 v = (*(const f32x4*)(x + bt * C + c4 * 4) - mean) * rstd * *(const f32x4*)(gamma + c4 * 4) +
                    *(const f32x4*)(beta + c4 * 4);
    *(f32x4*)(out + (int64_t)b * out_bstride + (int64_t)t * C + c4 * 4) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// f32 rows -> f16 with a row gather (-1 -> zeros): masked_fill of invalid proposals
// (utils.py:111-113) and the top-k gathers (transformer.py:302-316).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const int32_t* __restrict__ idx,
                                                          int64_t idx_bstride, int64_t x_bstride_rows,
                                                          int rows_per_b, int B, int C,
                                                          f16* __restrict__ out_h, float* __restrict__ out_f) {
  const int cv = C / 4;
  const int64_t total = (int64_t)B * rows_per_b * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % cv);
    const int64_t r = i / cv;
    const int b = (int)(r / rows_per_b), rr = (int)(r % rows_per_b);
    const int src = idx[(int64_t)b * idx_bstride + rr];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0) v = *(const f32x4*)(x + ((int64_t)b * x_bstride_rows + src) * ldx + c4 * 4);
    if (out_h) *(f16x4*)(out_h + r * C + c4 * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    if (out_f) *(f32x4*)(out_f + r * C + c4 * 4) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// BiMultiHeadAttention, image side (fuse_modules.py:168-225): per image token s and head h
//   score[s,h,t] = scale * q[s,h,:] . k[t,h,:]          (written out for the text side)
//   out_v[s,h,:] = sum_t softmax_t(score)[t] * values_l[t,h,:]
// QV f16 [B*S, 2E] = [v_proj(v) | values_v_proj(v)], KL f16 [B*T, 2E] = [l_proj(l) | values_l_proj(l)],
// E = 1024 = 4 heads x 256.  One wave per token: lane owns 16 of the 1024 dims (head = lane/16).
// The global max subtraction / +-50000 clamps of the reference are no-ops for a softmax unless
// |score| > 5e4 (never for sane weights) and are not reproduced.
template <int E>
__global__ __launch_bounds__(256) void biattn_image_kernel(const f16* __restrict__ QV, const f16* __restrict__ KL,
                                                           int B, int S, int T, float scale,
                                                           float* __restrict__ scores, f16* __restrict__ out_v) {
  constexpr int H = 4, HD = E / H, LD = 2 * E;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* kl = (f16*)smem;                                   // [T][2E] of this image
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < T * LD / 8; i += 256)
    ((f16x8*)kl)[i] = ((const f16x8*)(KL + (int64_t)b * T * LD))[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane / 16;
  const int stride = gridDim.x * 4;
  int s = blockIdx.x * 4 + wv;
  f16x8 q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0;
  if (s < S) {
    const f16* qp = QV + ((int64_t)b * S + s) * LD + lane * 16;
    q0 = *(const f16x8*)qp;
    q1 = *(const f16x8*)(qp + 8);
  }
  for (; s < S; s += stride) {
    // the next token's query row is fetched before this token's reductions (one token in flight per wave)
    f16x8 n0 = q0, n1 = q1;
    if (s + stride < S) {
      const f16* np = QV + ((int64_t)b * S + s + stride) * LD + lane * 16;
      n0 = *(const f16x8*)np;
      n1 = *(const f16x8*)(np + 8);
    }
    float sc[16];
    float mx = -3.0e38f;
    for (int t = 0; t < T; ++t) {
      const f16x8 k0 = *(const f16x8*)(kl + t * LD + lane * 16), k1 = *(const f16x8*)(kl + t * LD + lane * 16 + 8);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf((float)q0[j], (float)k0[j], d);
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf((float)q1[j], (float)k1[j], d);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);   // reduce over the head's 16 lanes
      sc[t] = d * scale;
      mx = fmaxf(mx, sc[t]);
    }
    if ((lane & 15) == 0)
      for (int t = 0; t < T; ++t) scores[(((int64_t)b * S + s) * H + h) * T + t] = sc[t];
    float sum = 0.f;
    for (int t = 0; t < T; ++t) { sc[t] = expf(sc[t] - mx); sum += sc[t]; }
    const float inv = 1.f / sum;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int t = 0; t < T; ++t) {
      const f16* vp = kl + t * LD + E + lane * 16;
      const f16x8 v0 = *(const f16x8*)vp, v1 = *(const f16x8*)(vp + 8);
      const float pw = sc[t] * inv;
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[j] = fmaf(pw, (float)v0[j], acc[j]); acc[8 + j] = fmaf(pw, (float)v1[j], acc[8 + j]); }
    }
    f16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = (f16)acc[j]; o1[j] = (f16)acc[8 + j]; }
    f16* op = out_v + ((int64_t)b * S + s) * E + lane * 16;
    *(f16x8*)op = o0;
    *(f16x8*)(op + 8) = o1;
    q0 = n0;
    q1 = n1;
  }
}

// text side, pass 1a: per (b, row-chunk): partial max / sum-exp of every (h, t) column over the chunk's rows
__global__ __launch_bounds__(256) void biattn_colstats_partial_kernel(const float* __restrict__ scores, int S,
                                                                      int HT, int rows_per_chunk,
                                                                      float* __restrict__ part) {
  const int c = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
  const float* sp = scores + (int64_t)b * S * HT;
  const int s0 = c * rows_per_chunk, s1 = min(S, s0 + rows_per_chunk);
  __shared__ float red[4][2];
  float mx[16], sm[16];
  for (int k = 0; k < HT; ++k) { mx[k] = -3.0e38f; sm[k] = 0.f; }
  for (int s = s0 + threadIdx.x; s < s1; s += 256)
    for (int k = 0; k < HT; ++k) {
      const float v = sp[(int64_t)s * HT + k];
      if (v > mx[k]) { sm[k] = sm[k] * expf(mx[k] - v) + 1.f; mx[k] = v; } else { sm[k] += expf(v - mx[k]); }
    }
  for (int k = 0; k < HT; ++k) {
    const float gm = wave_max(mx[k]);
    const float gs = wave_sum(sm[k] * expf(mx[k] - gm));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = gm; red[threadIdx.x >> 6][1] = gs; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float M = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
      float Ssum = 0.f;
      for (int w = 0; w < 4; ++w) Ssum += red[w][1] * expf(red[w][0] - M);
      part[(((int64_t)b * nchunk + c) * HT + k) * 2] = M;
      part[(((int64_t)b * nchunk + c) * HT + k) * 2 + 1] = Ssum;
    }
  }
}
// pass 1b: combine the chunk partials -> stats[b][col] = (max, sum-exp)
__global__ __launch_bounds__(64) void biattn_colstats_combine_kernel(const float* __restrict__ part, int nchunk,
                                                                     int HT, float* __restrict__ stats) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= HT) return;
  float M = -3.0e38f;
  for (int c = 0; c < nchunk; ++c) M = fmaxf(M, part[(((int64_t)b * nchunk + c) * HT + k) * 2]);
  float Ssum = 0.f;
  for (int c = 0; c < nchunk; ++c)
    Ssum += part[(((int64_t)b * nchunk + c) * HT + k) * 2 + 1] * expf(part[(((int64_t)b * nchunk + c) * HT + k) * 2] - M);
  stats[((int64_t)b * HT + k) * 2] = M;
  stats[((int64_t)b * HT + k) * 2 + 1] = Ssum;
}

// text side, pass 2: partial[b,h,chunk,t,d] = sum_{s in chunk} exp(score[s,h,t]-max) * values_v[s,h,d]
// One workgroup per (chunk, batch): 128 threads x 8 consecutive value dims = the whole 1024-wide value half of a
// row as ONE coalesced 2-KiB read (16 B per lane); the chunk's exp() values of all 4 heads are computed once into
// LDS.  (The first version read one f16 per lane per row: 0.9 TB/s.)
template <int E>
__global__ __launch_bounds__(E / 8) void biattn_text_partial_kernel(const float* __restrict__ scores,
                                                                    const float* __restrict__ stats,
                                                                    const f16* __restrict__ QV, int S, int T,
                                                                    int chunk, float* __restrict__ partial) {
  constexpr int H = 4, HD = E / H, LD = 2 * E, NTH = E / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pe = (float*)smem;                                 // [chunk][H][T]
  const int nchunk = gridDim.x;
  const int c = blockIdx.x, b = blockIdx.y;
  const int d0 = threadIdx.x * 8, h = d0 / HD;              // this thread's 8 dims sit inside head h
  const int s0 = c * chunk, s1 = min(S, s0 + chunk);
  const int HT = H * T;
  for (int i = threadIdx.x; i < (s1 - s0) * HT; i += NTH) {
    const int r = i / HT, k = i % HT;                       // k = head * T + t: the layout of a scores row
    pe[i] = expf(scores[((int64_t)b * S + s0 + r) * HT + k] - stats[((int64_t)b * HT + k) * 2]);
  }
  __syncthreads();
  float acc[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const f16* vp = QV + ((int64_t)b * S + s0) * LD + E + d0;
  const int n = s1 - s0;
  auto accum = [&](int s, const f16x8& v) {
    const float* pr = pe + s * HT + h * T;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < T) {
        const float w = pr[t];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = fmaf(w, (float)v[j], acc[t][j]);
      }
    }
  };
  int s = 0;
  for (; s + 8 <= n; s += 8) {                              // eight rows in flight per thread
    f16x8 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f16x8*)(vp + (int64_t)(s + u) * LD);
#pragma unroll
    for (int u = 0; u < 8; ++u) accum(s + u, v[u]);
  }
  for (; s < n; ++s) accum(s, *(const f16x8*)(vp + (int64_t)s * LD));
  const int dh = d0 % HD;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < T) {
      float* o = partial + ((((int64_t)b * H + h) * nchunk + c) * T + t) * HD + dh;
      *(f32x4*)o = (f32x4){acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
      *(f32x4*)(o + 4) = (f32x4){acc[t][4], acc[t][5], acc[t][6], acc[t][7]};
    }
  }
}
// pass 3: out_l[b*T + t, h*HD + d] = sum_chunks partial / sumexp
template <int E>
__global__ __launch_bounds__(256) void biattn_text_reduce_kernel(const float* __restrict__ partial,
                                                                 const float* __restrict__ stats, int T,
                                                                 int nchunk, f16* __restrict__ out_l) {
  constexpr int H = 4, HD = E / H;
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += partial[((((int64_t)b * H + h) * nchunk + c) * T + t) * HD + d];
  out_l[((int64_t)b * T + t) * E + h * HD + d] = (f16)(s / stats[((int64_t)b * H * T + h * T + t) * 2 + 1]);
}

// ---------------------------------------------------------------------------------------------
// Attention against a handful of keys (n_k <= 16): text self-attention (4x4, block-diagonal mask,
// transformer_vanilla.py:114-116) and decoder text cross-attention (900 x 4, transformer.py:893-900).
// One thread per (batch, query, head).
// 8 consecutive head-dim elements of an f16 or f32 row, as f32
__device__ __forceinline__ void load8(const f16* p, float* d) {
  const f16x8 v = *(const f16x8*)p;
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = (float)v[j];
}
__device__ __forceinline__ void load8(const float* p, float* d) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { d[j] = a[j]; d[4 + j] = b[j]; }
}
__device__ __forceinline__ void store8(f16* p, const float* d) {
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (f16)d[j];
  *(f16x8*)p = v;
}
__device__ __forceinline__ void store8(float* p, const float* d) {
  *(f32x4*)p = (f32x4){d[0], d[1], d[2], d[3]};
  *(f32x4*)(p + 4) = (f32x4){d[4], d[5], d[6], d[7]};
}

template <int HD, typename T>
__global__ __launch_bounds__(256) void attn_fewkeys_kernel(const T* __restrict__ Q, int64_t ldq,
                                                           const T* __restrict__ K, int64_t ldk,
                                                           const T* __restrict__ V, int64_t ldv, int B,
                                                           int n_q, int n_k, int n_heads, float scale,
                                                           const uint8_t* __restrict__ blocked,
                                                           const int32_t* __restrict__ q_rows,
                                                           const float* __restrict__ q_add,
                                                           T* __restrict__ O, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * n_q * n_heads) return;
  const int h = (int)(gid % n_heads);
  const int64_t bq = gid / n_heads;
  const int b = (int)(bq / n_q), q = (int)(bq % n_q);
  float qv[HD];
  const T* qp = Q + (q_rows ? (int64_t)q_rows[b] + q : bq) * ldq + h * HD;
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) load8(qp + 8 * i, qv + 8 * i);
  if (q_add) {   // per-position constant of the query projection (the (x + pe) W = x W + pe W split, sam.py)
    const float* ap = q_add + ((int64_t)q * n_heads + h) * HD;
#pragma unroll
    for (int i = 0; i < HD; ++i) qv[i] += ap[i];
  }
  float sc[16];
  float mx = -3.0e38f;
  for (int t = 0; t < n_k; ++t) {
    const T* kp = K + ((int64_t)b * n_k + t) * ldk + h * HD;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      float kv[8];
      load8(kp + 8 * i, kv);
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qv[8 * i + j], kv[j], d);
    }
    d *= scale;
    if (blocked && blocked[q * n_k + t]) d = -3.0e38f;
    sc[t] = d;
    mx = fmaxf(mx, d);
  }
  float sum = 0.f;
  for (int t = 0; t < n_k; ++t) { sc[t] = sc[t] <= -1.0e38f ? 0.f : expf(sc[t] - mx); sum += sc[t]; }
  const float inv = 1.f / sum;
  float acc[HD];
#pragma unroll
  for (int i = 0; i < HD; ++i) acc[i] = 0.f;
  for (int t = 0; t < n_k; ++t) {
    const T* vp = V + ((int64_t)b * n_k + t) * ldv + h * HD;
    const float pw = sc[t] * inv;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      float vv[8];
      load8(vp + 8 * i, vv);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[8 * i + j] = fmaf(pw, vv[j], acc[8 * i + j]);
    }
  }
  T* op = O + bq * ldo + h * HD;
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) store8(op + 8 * i, acc + 8 * i);
}


// head_dim 16, f32 rows (the SAM decoder's image->token attention: 4096 queries x 7 keys per box, 8 heads): FOUR lanes
// per (query, head), 4 of the 16 dimensions each, so that every load and store instruction of a wave covers one
// contiguous KiB (a thread per (query, head) reads 64 B at a 64-B stride: every cache line is touched by four
// instructions); the four partial dot products meet through two shuffles.  Base-2 softmax (log2 e folded into the scale).
template <int NK>
__global__ __launch_bounds__(256) void attn_fewkeys16_f32_kernel(const float* __restrict__ Q, int64_t ldq,
                                                                 const float* __restrict__ K, int64_t ldk,
                                                                 const float* __restrict__ V, int64_t ldv, int B,
                                                                 int n_q, int n_heads, float scale,
                                                                 const int32_t* __restrict__ q_rows,
                                                                 const float* __restrict__ q_add,
                                                                 float* __restrict__ O, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * n_q * n_heads * 4) return;      // (whole 4-lane groups: the total is a multiple of 4)
  const int part = (int)(gid & 3);
  const int h = (int)((gid >> 2) % n_heads);
  const int64_t bq = (gid >> 2) / n_heads;
  const int b = (int)(bq / n_q), q = (int)(bq % n_q);
  const int col = h * 16 + 4 * part;
  f32x4 qv = *(const f32x4*)(Q + (q_rows ? (int64_t)q_rows[b] + q : bq) * ldq + col);
  if (q_add) qv += *(const f32x4*)(q_add + (int64_t)q * n_heads * 16 + col);
  qv *= scale * 1.44269504088896340736f;
  const float* kp = K + (int64_t)b * NK * ldk + col;
  const float* vp = V + (int64_t)b * NK * ldv + col;
  f32x4 kv[NK], vv[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    kv[t] = *(const f32x4*)(kp + (int64_t)t * ldk);
    vv[t] = *(const f32x4*)(vp + (int64_t)t * ldv);
  }
  float sc[NK];
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    float d = (qv[0] * kv[t][0] + qv[1] * kv[t][1]) + (qv[2] * kv[t][2] + qv[3] * kv[t][3]);
    // the four lanes of a (query, head) group are a DPP quad: xor 1 = quad_perm [1,0,3,2], xor 2 = [2,3,0,1] (no LDS trip)
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, false));
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, false));
    sc[t] = d;
    mx = fmaxf(mx, d);
  }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NK; ++t) { sc[t] = __builtin_amdgcn_exp2f(sc[t] - mx); sum += sc[t]; }
  const float inv = 1.f / sum;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NK; ++t) acc += (sc[t] * inv) * vv[t];
  *(f32x4*)(O + bq * ldo + col) = acc;
}

// ---------------------------------------------------------------------------------------------
// Attention of a FEW queries (n_q <= 8) against many keys: SAM decoder token->image attention
// (7 tokens x 4096 image keys, 8 heads x 16; SA/modeling/transformer.py:163-168, 101-103).
// One workgroup per (batch, head); thread (qi = tid/32, kl = tid%32) streams keys kl, kl+32, ... with an
// online softmax, then the 32 key-lanes of each query are merged with half-wave shuffles.
template <int HD>
__global__ __launch_bounds__(256) void attn_fewq_kernel(const f16* __restrict__ Q, int64_t ldq,
                                                        const f16* __restrict__ K, int64_t ldk,
                                                        const f16* __restrict__ V, int64_t ldv, int n_q,
                                                        int n_k, int n_heads, float scale,
                                                        const int32_t* __restrict__ q_rows,
                                                        const int32_t* __restrict__ kv_rows,
                                                        f16* __restrict__ O, int64_t ldo) {
  const int b = blockIdx.x / n_heads, h = blockIdx.x % n_heads;
  const int qi = threadIdx.x >> 5, kl = threadIdx.x & 31;
  const bool active = qi < n_q;
  const int64_t q0 = q_rows ? (int64_t)q_rows[b] : (int64_t)b * n_q;
  const int64_t k0 = kv_rows ? (int64_t)kv_rows[b] : (int64_t)b * n_k;
  float qv[HD];
  const f16* qp = Q + (q0 + (active ? qi : 0)) * ldq + h * HD;
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) {
    const f16x8 v = *(const f16x8*)(qp + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[8 * i + j] = (float)v[j] * scale;
  }
  float m = -3.0e38f, l = 0.f, acc[HD];
#pragma unroll
  for (int i = 0; i < HD; ++i) acc[i] = 0.f;
  for (int key = kl; key < n_k; key += 32) {
    const f16* kp = K + (k0 + key) * ldk + h * HD;
    const f16* vp = V + (k0 + key) * ldv + h * HD;
    float d = 0.f;
    f16x8 vv[HD / 8];
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      const f16x8 kk = *(const f16x8*)(kp + 8 * i);
      vv[i] = *(const f16x8*)(vp + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qv[8 * i + j], (float)kk[j], d);
    }
    const float mn = fmaxf(m, d);
    const float a = expf(m - mn), pw = expf(d - mn);
    l = l * a + pw;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[8 * i + j] = fmaf(pw, (float)vv[i][j], acc[8 * i + j] * a);
    m = mn;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {       // merge the 32 key-lanes (stays inside one half-wave)
    const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, mo);
    const float a = expf(m - mn), bsc = expf(mo - mn);
    l = l * a + lo * bsc;
#pragma unroll
    for (int i = 0; i < HD; ++i) acc[i] = acc[i] * a + __shfl_xor(acc[i], o, 64) * bsc;
    m = mn;
  }
  if (active && kl == 0) {
    const float inv = 1.f / l;
    f16* op = O + ((int64_t)b * n_q + qi) * ldo + h * HD;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      f16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (f16)(acc[8 * i + j] * inv);
      *(f16x8*)(op + 8 * i) = v;
    }
  }
}

// LDS-tiled form for head_dim 16 and n_heads % 4 == 0 (the SAM decoder: 8 heads x 16).  The kernel above reads
// 32-byte head slices at the row stride, once per query group.  Here one workgroup serves (batch entry, 4 heads):
// a key row's 4-head slice is exactly one 128-B line, 64-key K/V tiles are staged through LDS with 16-byte loads
// (each byte fetched once per workgroup), wave w owns queries 2w and 2w+1, lane (kl = lane / 4, hl = lane % 4)
// streams keys kl, kl+16, ... of head hl, and the 16 key-lanes of a head are merged with in-wave shuffles.
template <typename T, int KSPLIT>
__global__ __launch_bounds__(256 * KSPLIT) void attn_fewq16_kernel(const T* __restrict__ Q, int64_t ldq,
                                                          const T* __restrict__ K, int64_t ldk,
                                                          const T* __restrict__ V, int64_t ldv, int n_q,
                                                          int n_k, int n_heads, float scale,
                                                          const int32_t* __restrict__ q_rows,
                                                          const int32_t* __restrict__ kv_rows,
                                                          const float* __restrict__ k_add,
                                                          T* __restrict__ O, int64_t ldo) {
  // KSPLIT groups of four waves share the key range (group g streams keys [g, g + 1) * n_k / KSPLIT through its own LDS
  // tiles; the partial softmax states meet in LDS at the end): the launch is only n_batch x n_heads / 4 workgroups - one
  // per CU for 128 boxes - and one wave per SIMD leaves every LDS and memory latency exposed (round 3: 268 us for
  // 4096 keys x 128 boxes at KSPLIT = 1, whatever the tile size or prefetch depth).
  constexpr int HD = 16, HB = 4, TK = 64;
  constexpr int ROWE = HB * HD;                      // elements of K (or V) per key and 4-head group (one 128-B line in f16)
  constexpr int CH = 16 / (int)sizeof(T);            // elements per 16-byte chunk
  constexpr int NCH = ROWE / CH;                     // chunks per row: 8 (f16) / 16 (f32)
  constexpr int NU = TK * NCH / 256;                 // chunks per thread and operand: 2 / 4
  extern __shared__ __attribute__((aligned(16))) char fewq_smem[];
  const int grp = threadIdx.x >> 8;                  // key-range group of this wave
  T* sk = (T*)fewq_smem + grp * 2 * TK * ROWE;
  T* sv = sk + TK * ROWE;
  const int hgroups = n_heads / HB;
  const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * HB;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int per = ((n_k + KSPLIT - 1) / KSPLIT + TK - 1) / TK * TK;      // keys per group, whole tiles
  const int k_lo = grp * per, k_hi = min(n_k, k_lo + per);
  const int kl = lane >> 2, hl = lane & 3;
  const int64_t q0 = q_rows ? (int64_t)q_rows[b] : (int64_t)b * n_q;
  const int64_t k0 = kv_rows ? (int64_t)kv_rows[b] : (int64_t)b * n_k;
  const int qa = 2 * wave, qb = 2 * wave + 1;
  float qA[HD], qB[HD];
  {
    const T* pa = Q + (q0 + (qa < n_q ? qa : 0)) * ldq + (h0 + hl) * HD;
    const T* pb = Q + (q0 + (qb < n_q ? qb : 0)) * ldq + (h0 + hl) * HD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      load8(pa + 8 * i, qA + 8 * i);
      load8(pb + 8 * i, qB + 8 * i);
    }
    const float sc2 = scale * 1.44269504088896340736f;       // scores in log2 units: exp2 below
#pragma unroll
    for (int j = 0; j < HD; ++j) { qA[j] *= sc2; qB[j] *= sc2; }
  }
  float mA = -3.0e38f, lA = 0.f, mB = -3.0e38f, lB = 0.f, aA[HD], aB[HD];
#pragma unroll
  for (int i = 0; i < HD; ++i) aA[i] = aB[i] = 0.f;
  // staging: TK rows x NCH chunks of 16 B per operand; thread t moves chunks t, t + 256, ...  The NEXT tile's chunks are
  // loaded into registers before the current tile is computed (round 3: without that every one of the n_k / 64 tiles
  // exposed a full memory round trip - 268 us for 4096 keys, of which the arithmetic is ~100).
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rk[NU], rv[NU];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = tid + 256 * u, row = c / NCH, col = c % NCH;
      rk[u] = rv[u] = (u32x4){0u, 0u, 0u, 0u};
      if (t0 + row < k_hi) {
        rk[u] = *(const u32x4*)(K + (k0 + t0 + row) * ldk + h0 * HD + col * CH);
        rv[u] = *(const u32x4*)(V + (k0 + t0 + row) * ldv + h0 * HD + col * CH);
        if constexpr (sizeof(T) == 4) {
          if (k_add) {   // per-key constant of the key projection, [n_k, n_heads*HD] f32
            const f32x4 ad = *(const f32x4*)(k_add + (int64_t)(t0 + row) * n_heads * HD + h0 * HD + col * CH);
            f32x4 kv4 = __builtin_bit_cast(f32x4, rk[u]);
            kv4 += ad;
            rk[u] = __builtin_bit_cast(u32x4, kv4);
          }
        }
      }
    }
  };
  fetch(k_lo);
  for (int t0 = k_lo; t0 < k_lo + per; t0 += TK) {       // every group runs the same number of tiles: shared barriers
    __syncthreads();                                   // previous tile consumed
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = tid + 256 * u, row = c / NCH, col = c % NCH;
      *(u32x4*)(sk + row * ROWE + col * CH) = rk[u];
      *(u32x4*)(sv + row * ROWE + col * CH) = rv[u];
    }
    __syncthreads();
    if (t0 + TK < k_hi) fetch(t0 + TK);                // in flight under this tile's arithmetic
    // scores of the lane's four keys of this tile (base-2 exponent units: log2(e) is folded into the query scale), ONE
    // rescale of the running sums per tile and query
    float dA[TK / 16], dB[TK / 16];
#pragma unroll
    for (int kk = 0; kk < TK / 16; ++kk) {
      const int key = kk * 16 + kl;
      float kf[HD];
      load8(sk + key * ROWE + hl * HD, kf);
      load8(sk + key * ROWE + hl * HD + 8, kf + 8);
      float a0 = 0.f, b0 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = fmaf(qA[j], kf[j], a0); a0 = fmaf(qA[8 + j], kf[8 + j], a0);
        b0 = fmaf(qB[j], kf[j], b0); b0 = fmaf(qB[8 + j], kf[8 + j], b0);
      }
      const bool in = t0 + key < k_hi;
      dA[kk] = in ? a0 : -3.0e38f;
      dB[kk] = in ? b0 : -3.0e38f;
    }
    float nA = mA, nB = mB;
#pragma unroll
    for (int kk = 0; kk < TK / 16; ++kk) { nA = fmaxf(nA, dA[kk]); nB = fmaxf(nB, dB[kk]); }
    const float sA = __builtin_amdgcn_exp2f(mA - nA), sB = __builtin_amdgcn_exp2f(mB - nB);
    lA *= sA;
    lB *= sB;
#pragma unroll
    for (int j = 0; j < HD; ++j) { aA[j] *= sA; aB[j] *= sB; }
    mA = nA;
    mB = nB;
#pragma unroll
    for (int kk = 0; kk < TK / 16; ++kk) {
      const int key = kk * 16 + kl;
      float vf[HD];
      load8(sv + key * ROWE + hl * HD, vf);
      load8(sv + key * ROWE + hl * HD + 8, vf + 8);
      const bool in = t0 + key < k_hi;
      const float pA = in ? __builtin_amdgcn_exp2f(dA[kk] - nA) : 0.f, pB = in ? __builtin_amdgcn_exp2f(dB[kk] - nB) : 0.f;
      lA += pA;
      lB += pB;
#pragma unroll
      for (int j = 0; j < HD; ++j) { aA[j] = fmaf(pA, vf[j], aA[j]); aB[j] = fmaf(pB, vf[j], aB[j]); }
    }
  }
  auto merge = [&](float& m, float& l, float (&acc)[HD]) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {               // the 16 key-lanes of this head: lane bits 2..5
      const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
      const float mn = fmaxf(m, mo);
      const float a = __builtin_amdgcn_exp2f(m - mn), bsc = __builtin_amdgcn_exp2f(mo - mn);
      l = l * a + lo * bsc;
#pragma unroll
      for (int i = 0; i < HD; ++i) acc[i] = acc[i] * a + __shfl_xor(acc[i], o, 64) * bsc;
      m = mn;
    }
  };
  merge(mA, lA, aA);
  merge(mB, lB, aB);
  if constexpr (KSPLIT > 1) {
    // groups 1.. park their states (m, l, acc[16]) per (wave, query slot, head lane) in LDS; group 0 folds them in
    __syncthreads();                                     // the last tiles are consumed: the K/V tiles can be overwritten
    float* st = (float*)fewq_smem;
    if (grp > 0 && kl == 0) {
      float* pa = st + ((((grp - 1) * 4 + wave) * 2 + 0) * 4 + hl) * 18;
      float* pb = st + ((((grp - 1) * 4 + wave) * 2 + 1) * 4 + hl) * 18;
      pa[0] = mA; pa[1] = lA; pb[0] = mB; pb[1] = lB;
#pragma unroll
      for (int i = 0; i < HD; ++i) { pa[2 + i] = aA[i]; pb[2 + i] = aB[i]; }
    }
    __syncthreads();
    if (grp > 0) return;
    if (kl == 0) {
#pragma unroll
      for (int g2 = 1; g2 < KSPLIT; ++g2) {
        auto fold = [&](int slot, float& m, float& l, float (&acc)[HD]) {
          const float* ps = st + ((((g2 - 1) * 4 + wave) * 2 + slot) * 4 + hl) * 18;
          const float mo = ps[0], mn = fmaxf(m, mo);
          const float a = __builtin_amdgcn_exp2f(m - mn), bsc = __builtin_amdgcn_exp2f(mo - mn);
          l = l * a + ps[1] * bsc;
#pragma unroll
          for (int i = 0; i < HD; ++i) acc[i] = acc[i] * a + ps[2 + i] * bsc;
          m = mn;
        };
        fold(0, mA, lA, aA);
        fold(1, mB, lB, aB);
      }
    }
  }
  if (kl == 0) {
    auto put = [&](int q, float l, float (&acc)[HD]) {
      if (q < n_q) {
        const float inv = 1.f / l;
#pragma unroll
        for (int i = 0; i < HD; ++i) acc[i] *= inv;
        T* op = O + ((int64_t)b * n_q + q) * ldo + (h0 + hl) * HD;
        store8(op, acc);
        store8(op + 8, acc + 8);
      }
    };
    put(qa, lA, aA);
    put(qb, lB, aB);
  }
}

// ---------------------------------------------------------------------------------------------
// Two-stage query selection: key[s] = max_t logits[b,s,t]; indices of the K largest, in descending
// order, ties -> lower index (torch.topk leaves tie order unspecified).  One 1024-thread workgroup
// per image, bitonic sort of 64-bit (ordered-value, index) keys in LDS (S <= 16384 -> 128 KiB).
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int NP) {
  for (int size = 2; size <= NP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < NP / 2; i += 1024) {
        const int lo = 2 * i - (i & (stride - 1));         // index with the `stride` bit clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}
__device__ __forceinline__ void emit_topk(const unsigned long long* keys, int K, int b, int32_t* out_idx,
                                          float* out_val) {
  for (int i = threadIdx.x; i < K; i += 1024) {
    const unsigned long long k = keys[i];
    out_idx[(int64_t)b * K + i] = (int)(k & 0xffffffffu);
    if (out_val) {
      unsigned u = ~(unsigned)(k >> 32);
      u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
      out_val[(int64_t)b * K + i] = __uint_as_float(u);
    }
  }
}
// pass 1: grid (B, nchunk); each workgroup sorts one chunk of <= 16384 tokens and keeps its K best keys
// (nchunk == 1: writes the final result directly)
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ logits, int S, int T, int K,
                                                    int chunk, int NP, unsigned long long* __restrict__ cand,
                                                    int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  const int b = blockIdx.x, c = blockIdx.y, nchunk = gridDim.y;
  const float* lp = logits + (int64_t)b * S * T;
  const int s0 = c * chunk, s1 = min(S, s0 + chunk);
  for (int i = threadIdx.x; i < NP; i += 1024) {
    unsigned long long k = ~0ull;
    const int sidx = s0 + i;
    if (sidx < s1) {
      float v = lp[(int64_t)sidx * T];
      for (int t = 1; t < T; ++t) v = fmaxf(v, lp[(int64_t)sidx * T + t]);
      unsigned u = __float_as_uint(v);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone float -> uint
      k = ((unsigned long long)(~u) << 32) | (unsigned)sidx;  // ascending key == descending value, then index
    }
    keys[i] = k;
  }
  __syncthreads();
  bitonic_sort_lds(keys, NP);
  if (nchunk == 1) {
    emit_topk(keys, K, b, out_idx, out_val);
  } else {
    for (int i = threadIdx.x; i < K; i += 1024) cand[((int64_t)b * nchunk + c) * K + i] = keys[i];
  }
}
// pass 2: merge the nchunk * K candidates of one image
__global__ __launch_bounds__(1024) void topk_merge_kernel(const unsigned long long* __restrict__ cand, int n,
                                                          int K, int NP, int32_t* __restrict__ out_idx,
                                                          float* __restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < NP; i += 1024) keys[i] = i < n ? cand[(int64_t)b * n + i] : ~0ull;
  __syncthreads();
  bitonic_sort_lds(keys, NP);
  emit_topk(keys, K, b, out_idx, out_val);
}

// ---------------------------------------------------------------------------------------------
// gen_sineembed_for_position for 4-d boxes (utils.py:204-230): ref [N,4] (cx,cy,w,h) -> f16 [N,512]
// ordered (y, x, w, h) x 128, element i = sin / cos (i even / odd) of 2*pi*v / dim_t[i].
__global__ __launch_bounds__(256) void sine_embed4_kernel(const float* __restrict__ ref,
                                                          const float* __restrict__ dim_t, int N,
                                                          f16* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)N * 512) return;
  const int n = (int)(gid >> 9), c = (int)(gid & 511);
  const int j = c >> 7, i = c & 127;
  const int src = j == 0 ? 1 : (j == 1 ? 0 : j);
  const float v = ref[n * 4 + src] * 6.283185307179586f / dim_t[i];
  out[gid] = (f16)((i & 1) ? cosf(v) : sinf(v));
}

// new_ref = sigmoid(delta + inverse_sigmoid(ref))  (transformer.py:716-722, misc.py:704-708)
__global__ __launch_bounds__(256) void box_refine_kernel(const float* __restrict__ delta, int64_t ldd,
                                                         const float* __restrict__ ref, int N,
                                                         int ref_is_logit, float* __restrict__ out) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= N * 4) return;
  const int n = gid >> 2, c = gid & 3;
  float x = ref[gid];
  float lg = x;
  if (!ref_is_logit) {
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-3f), x2 = fmaxf(1.f - x, 1e-3f);
    lg = logf(x1 / x2);
  }
  const float z = delta[(int64_t)n * ldd + c] + lg;
  out[gid] = 1.f / (1.f + expf(-z));
}

}  // namespace

extern "C" int ink_swin_patchify(const void* image_u8, int32_t h, int32_t w, const float* mean3,
                                 const float* std3, void* out_f16, void* stream) {
  INK_CHECK_ARG(image_u8 && mean3 && std3 && out_f16 && h > 0 && w > 0);
  const int gh = (h + 3) / 4, gw = (w + 3) / 4;
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
  hipLaunchKernelGGL(swin_patchify_kernel, dim3((gh * gw + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)image_u8, h, w, gh, gw, m, sd, (f16*)out_f16);
  return ink_launch_status();
}

extern "C" int ink_layernorm_merge4(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                    float eps, const int32_t* gather4, int32_t rows, int32_t C,
                                    void* out_f16, void* stream) {
  INK_CHECK_ARG(x && gamma && beta && gather4 && out_f16 && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024);
  INK_CHECK_ARG(ldx % 4 == 0 && ldx >= C);
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nv = (C + 63) / 64;  // float4 per lane over 4C values
#define INK_M4(NV) hipLaunchKernelGGL(ln_merge4_kernel<NV>, grid, block, 0, s, x, ldx, gamma, beta, eps, gather4, rows, C, (f16*)out_f16)
  if (nv <= 2) { INK_M4(2); } else if (nv <= 3) { INK_M4(3); } else if (nv <= 6) { INK_M4(6); } else if (nv <= 12) { INK_M4(12); } else { INK_M4(16); }
#undef INK_M4
  return ink_launch_status();
}

extern "C" int ink_groupnorm_nhwc(const float* x, int32_t B, int32_t T, int32_t C, int32_t G,
                                  const float* gamma, const float* beta, float eps, float* stats_ws,
                                  float* out, int64_t out_batch_stride, void* stream) {
  INK_CHECK_ARG(x && gamma && beta && stats_ws && out && B > 0 && T > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(B * G), dim3(256), 0, s, x, T, C, G, eps, stats_ws);
  const int64_t total = (int64_t)B * T * C / 4;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(blocks), dim3(256), 0, s, x, T, C, G, stats_ws, gamma, beta,
                     out, out_batch_stride, B);
  return ink_launch_status();
}

extern "C" int ink_gather_rows(const float* x, int64_t ldx, int64_t x_batch_rows, const int32_t* idx,
                               int64_t idx_batch_stride, int32_t rows_per_batch, int32_t B, int32_t C,
                               void* out_f16, float* out_f32, void* stream) {
  INK_CHECK_ARG(x && idx && (out_f16 || out_f32) && rows_per_batch > 0 && B > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0);
  const int64_t total = (int64_t)B * rows_per_batch * C / 4;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, idx,
                     idx_batch_stride, x_batch_rows, rows_per_batch, B, C, (f16*)out_f16, out_f32);
  return ink_launch_status();
}

extern "C" int ink_biattn_fusion(const void* QV_f16, const void* KL_f16, int32_t B, int32_t S, int32_t T,
                                 int32_t E, float scale, float* scores_ws, float* stats_ws,
                                 float* partial_ws, int32_t chunk, void* out_v_f16, void* out_l_f16,
                                 void* stream) {
  INK_CHECK_ARG(QV_f16 && KL_f16 && scores_ws && stats_ws && partial_ws && out_v_f16 && out_l_f16);
  INK_CHECK_ARG(B > 0 && S > 0 && T > 0 && T <= 4 && E == 1024 && chunk > 0);   // H*T <= 16
  hipStream_t s = (hipStream_t)stream;
  const int lds = T * 2 * E * 2;
  const int bx = (S + 3) / 4 < 512 ? (S + 3) / 4 : 512;
  hipLaunchKernelGGL(biattn_image_kernel<1024>, dim3(bx, B), dim3(256), lds, s, (const f16*)QV_f16,
                     (const f16*)KL_f16, B, S, T, scale, scores_ws, (f16*)out_v_f16);
  const int nchunk = (S + chunk - 1) / chunk;
  // column statistics: chunk partials are parked at the head of partial_ws (consumed before pass 2 rewrites it)
  const int rows_cs = 512, ncs = (S + rows_cs - 1) / rows_cs;
  INK_CHECK_ARG((int64_t)ncs * 4 * T * 2 <= (int64_t)4 * nchunk * T * 256);
  hipLaunchKernelGGL(biattn_colstats_partial_kernel, dim3(ncs, B), dim3(256), 0, s, scores_ws, S, 4 * T, rows_cs,
                     partial_ws);
  hipLaunchKernelGGL(biattn_colstats_combine_kernel, dim3(B), dim3(64), 0, s, partial_ws, ncs, 4 * T, stats_ws);
  hipLaunchKernelGGL(biattn_text_partial_kernel<1024>, dim3(nchunk, B), dim3(128), chunk * 4 * T * 4, s, scores_ws,
                     stats_ws, (const f16*)QV_f16, S, T, chunk, partial_ws);
  hipLaunchKernelGGL(biattn_text_reduce_kernel<1024>, dim3(T, 4, B), dim3(256), 0, s, partial_ws, stats_ws, T,
                     nchunk, (f16*)out_l_f16);
  return ink_launch_status();
}

// column (max, sum-exp) over the S rows of every image's [S, HT] score matrix (the text-side softmax statistics), also
// used by the folded fusion layer (fusion_fold.hip).  part_ws: B * ceil(S / 512) * HT * 2 floats.
extern "C" int ink_biattn_colstats(const float* scores, int32_t B, int32_t S, int32_t HT, float* part_ws, float* stats,
                                   void* stream) {
  INK_CHECK_ARG(scores && part_ws && stats && B > 0 && S > 0 && HT > 0 && HT <= 16);
  hipStream_t s = (hipStream_t)stream;
  const int rows_cs = 512, ncs = (S + rows_cs - 1) / rows_cs;
  hipLaunchKernelGGL(biattn_colstats_partial_kernel, dim3(ncs, B), dim3(256), 0, s, scores, S, HT, rows_cs, part_ws);
  hipLaunchKernelGGL(biattn_colstats_combine_kernel, dim3(B), dim3(64), 0, s, (const float*)part_ws, ncs, HT, stats);
  return ink_launch_status();
}

extern "C" int ink_attn_fewkeys(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V,
                                int64_t ldv, int32_t B, int32_t n_q, int32_t n_k, int32_t n_heads,
                                int32_t head_dim, float scale, const uint8_t* blocked,
                                const int32_t* q_batch_rows, const float* q_add, int32_t io_f32, void* O,
                                int64_t ldo, void* stream) {
  INK_CHECK_ARG(Q && K && V && O && B > 0 && n_q > 0 && n_k > 0 && n_k <= 16 && n_heads > 0);
  INK_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  INK_CHECK_ARG(io_f32 == 0 || io_f32 == 1);
  INK_CHECK_ARG(!q_add || (((uintptr_t)q_add & 15) == 0));
  const int64_t total = (int64_t)B * n_q * n_heads;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define INK_FEWKEYS(HD, T)                                                                                     \
  hipLaunchKernelGGL((attn_fewkeys_kernel<HD, T>), grid, block, 0, s, (const T*)Q, ldq, (const T*)K, ldk,      \
                     (const T*)V, ldv, B, n_q, n_k, n_heads, scale, blocked, q_batch_rows, q_add, (T*)O, ldo)
  if (head_dim == 32 && !io_f32) INK_FEWKEYS(32, f16);
  else if (head_dim == 64 && !io_f32) INK_FEWKEYS(64, f16);
  else if (head_dim == 16 && !io_f32) INK_FEWKEYS(16, f16);
  else if (head_dim == 32) INK_FEWKEYS(32, float);
  else if (head_dim == 16 && !blocked && n_k == 7) {        // SAM's 5 output tokens + 2 box corners
    const int64_t t4 = total * 4;
    hipLaunchKernelGGL(attn_fewkeys16_f32_kernel<7>, dim3((unsigned)((t4 + 255) / 256)), block, 0, s, (const float*)Q, ldq,
                       (const float*)K, ldk, (const float*)V, ldv, B, n_q, n_heads, scale, q_batch_rows, q_add,
                       (float*)O, ldo);
  } else if (head_dim == 16) INK_FEWKEYS(16, float);
  else return INK_ERR_ARG;
#undef INK_FEWKEYS
  return ink_launch_status();
}

extern "C" int ink_attn_fewq(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                             int32_t n_batch, int32_t n_q, int32_t n_k, int32_t n_heads, int32_t head_dim,
                             float scale, const int32_t* q_batch_rows, const int32_t* kv_batch_rows,
                             const float* k_add, int32_t io_f32, void* O, int64_t ldo, void* stream) {
  INK_CHECK_ARG(Q && K && V && O && n_batch > 0 && n_q > 0 && n_q <= 8 && n_k > 0 && n_heads > 0);
  INK_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  INK_CHECK_ARG(io_f32 == 0 || (io_f32 == 1 && head_dim == 16 && n_heads % 4 == 0));
  INK_CHECK_ARG(!k_add || (io_f32 == 1 && ((uintptr_t)k_add & 15) == 0));
  const dim3 grid(n_batch * n_heads), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 16 && n_heads % 4 == 0 && io_f32) {
    constexpr int lds = 2 * 2 * 64 * 64 * 4;           // two key-range groups x (K, V) tiles of 64 keys x 4 heads x 16 f32
    static bool attr = ((void)hipFuncSetAttribute((const void*)attn_fewq16_kernel<float, 2>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
    (void)attr;
    hipLaunchKernelGGL((attn_fewq16_kernel<float, 2>), dim3(n_batch * (n_heads / 4)), dim3(512), lds, s, (const float*)Q, ldq,
                       (const float*)K, ldk, (const float*)V, ldv, n_q, n_k, n_heads, scale, q_batch_rows,
                       kv_batch_rows, k_add, (float*)O, ldo);
  } else if (head_dim == 16 && n_heads % 4 == 0) {
    hipLaunchKernelGGL((attn_fewq16_kernel<f16, 1>), dim3(n_batch * (n_heads / 4)), block, 2 * 64 * 64 * 2, s, (const f16*)Q, ldq,
                       (const f16*)K, ldk, (const f16*)V, ldv, n_q, n_k, n_heads, scale, q_batch_rows, kv_batch_rows,
                       (const float*)nullptr, (f16*)O, ldo);
  } else if (head_dim == 16) {
    hipLaunchKernelGGL(attn_fewq_kernel<16>, grid, block, 0, s, (const f16*)Q, ldq, (const f16*)K, ldk,
                       (const f16*)V, ldv, n_q, n_k, n_heads, scale, q_batch_rows, kv_batch_rows, (f16*)O, ldo);
  } else if (head_dim == 32) {
    hipLaunchKernelGGL(attn_fewq_kernel<32>, grid, block, 0, s, (const f16*)Q, ldq, (const f16*)K, ldk,
                       (const f16*)V, ldv, n_q, n_k, n_heads, scale, q_batch_rows, kv_batch_rows, (f16*)O, ldo);
  } else {
    return INK_ERR_ARG;
  }
  return ink_launch_status();
}

extern "C" int ink_topk_rowmax(const float* logits, int32_t B, int32_t S, int32_t T, int32_t K,
                               int32_t* out_idx, float* out_val, void* cand_ws, void* stream) {
  constexpr int CHUNK = 16384;
  INK_CHECK_ARG(logits && out_idx && B > 0 && S > 0 && T > 0 && K > 0 && K <= S);
  const int nchunk = (S + CHUNK - 1) / CHUNK;
  INK_CHECK_ARG(nchunk == 1 || (cand_ws && K <= CHUNK / 2 && S - (nchunk - 1) * CHUNK >= 0 && nchunk * K <= 16384));
  INK_CHECK_ARG(nchunk == 1 || S / nchunk >= K);   // every chunk must hold at least K tokens
  static bool attr = ((void)hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                CHUNK * 8),
                      (void)hipFuncSetAttribute((const void*)topk_merge_kernel,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK * 8), true);
  (void)attr;
  hipStream_t s = (hipStream_t)stream;
  // equal chunks so that each one has >= K tokens
  const int chunk = (S + nchunk - 1) / nchunk;
  int NP = 2;
  while (NP < chunk) NP <<= 1;
  hipLaunchKernelGGL(topk_kernel, dim3(B, nchunk), dim3(1024), NP * 8, s, logits, S, T, K, chunk, NP,
                     (unsigned long long*)cand_ws, out_idx, out_val);
  if (nchunk > 1) {
    const int n = nchunk * K;
    int NP2 = 2;
    while (NP2 < n) NP2 <<= 1;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(1024), NP2 * 8, s, (const unsigned long long*)cand_ws, n, K,
                       NP2, out_idx, out_val);
  }
  return ink_launch_status();
}

extern "C" int ink_sine_embed4(const float* ref, const float* dim_t, int32_t N, void* out_f16, void* stream) {
  INK_CHECK_ARG(ref && dim_t && out_f16 && N > 0);
  const int64_t total = (int64_t)N * 512;
  hipLaunchKernelGGL(sine_embed4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, ref, dim_t, N, (f16*)out_f16);
  return ink_launch_status();
}

extern "C" int ink_box_refine(const float* delta, int64_t ldd, const float* ref, int32_t N,
                              int32_t ref_is_logit, float* out, void* stream) {
  INK_CHECK_ARG(delta && ref && out && N > 0 && ldd >= 4);
  hipLaunchKernelGGL(box_refine_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, delta,
                     ldd, ref, N, ref_is_logit, out);
  return ink_launch_status();
}

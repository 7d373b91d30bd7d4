/*
 * inklayer_hip.h — C ABI of libinklayer_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the InkLayer detector→segmentor hot path.  The
 * reference (ooowedyn/InkLayer) is Python on torch.nn modules plus ONE native
 * op (groundingdino._C.ms_deform_attn_forward).  Every entry point below
 * replaces one reference call site (cited as file:line under
 * /root/reference; GD/ = InkLayer/third_party/GroundingDINO/groundingdino/,
 * SA/ = InkLayer/third_party/segment-anything/segment_anything/).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     unless a parameter says "host";
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *     on it (no allocation, no synchronisation: safe under hipGraph capture);
 *   - return value: 0 = ok, 1 = bad argument (nothing was launched),
 *     2 = the HIP launch failed;
 *   - "f16" is IEEE binary16 (_Float16); accumulation is always f32;
 *   - inputs are borrowed, outputs are caller-allocated.
 */
#ifndef INKLAYER_HIP_H
#define INKLAYER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INK_ABI_VERSION 4
int ink_abi_version(void);

/* ------------------------------------------------------------------------
 * Dense projection:  C[row_map[m], n] = residual[row_map[m], n]
 *                       + col_scale[n] * act( sum_k A[m,k] * W[n,k] + bias[n] )
 * W is in nn.Linear layout [N,K] (K contiguous).  MFMA 16x16x32 f16, f32 acc.
 * Replaces every nn.Linear / 1x1-conv / im2col-conv on the path, e.g.
 *   SA/modeling/image_encoder.py:231 (qkv), :237 (proj), SA/modeling/common.py:25
 *   (MLPBlock lin1+GELU+lin2), GD/.../swin_transformer.py:140,170 (qkv/proj),
 *   GD/.../transformer.py:789-795 (encoder FFN), SA/modeling/mask_decoder.py:53-59
 *   (ConvTranspose2d k2s2 as a [256 -> 4*64] projection).
 * Constraints: K % 32 == 0, N % 4 == 0, lda/ldw % 8 == 0, ldc/ldr % 4 == 0,
 * A/W 16-byte aligned.
 * --------------------------------------------------------------------- */
typedef struct InkGemm {
  const void* A;            /* f16 [M, lda] */
  const void* W;            /* f16 [N, ldw] */
  const float* bias;        /* [N] or NULL */
  const float* col_scale;   /* [N] or NULL */
  const float* residual;    /* f32 [*, ldr] or NULL; indexed by OUTPUT row */
  const int32_t* row_map;   /* [M]: output row of input row m, <0 drops the row; NULL = identity */
  void* C;                  /* f32 or f16 [*, ldc] */
  int32_t M, N, K;
  int32_t lda, ldw, ldr, ldc;
  int32_t act;              /* 0 none, 1 GELU(erf), 2 ReLU */
  int32_t c_f16;            /* 0: C is f32, 1: C is f16, 2: split f16 - C = hi plane, C_lo = lo plane, value = hi + lo */
  /* ABI 4: the residual stream as two f16 planes + LayerNorm folded into the consuming projection (SAM ViT-H blocks,
   * SA/modeling/image_encoder.py:166-182: x = x + attn(norm1(x)); x = x + mlp(norm2(x))) */
  void* C_lo;               /* c_f16 == 2: f16 [*, ldc], lo = f16(v - f16(v)) */
  const void* res_hi;       /* split residual (instead of `residual`): f16 [*, ldr] planes, residual = hi + lo; */
  const void* res_lo;       /*   only in the linear form (no activation, no col_scale) */
  float* stats_out;         /* c_f16 == 2: [rows, stats_parts, 2] (sum, sum of squares) of every output row over each
                               column chunk of ink_gemm_query_stats_chunk(M, N, K) columns; stats_parts = N / chunk */
  const float* ln_stats;    /* folded LayerNorm over the rows of A: [M, ln_parts, 2] partial sums over the ln_dim source
                               columns; then C = rstd_m * (A W^T - mean_m * ln_colsum) + bias (W carries gamma, bias
                               carries beta W^T + b) before the activation */
  const float* ln_colsum;   /* [N]: sum_k W[n, k] */
  int32_t stats_parts, ln_parts, ln_dim;
  float ln_eps;
} InkGemm;
#define INK_ACT_NONE 0
#define INK_ACT_GELU 1
#define INK_ACT_RELU 2
int ink_gemm_f16(const InkGemm* p, void* stream);
/* Tuning knob (tools/gemm_sweep.py): force tile variant v >= 0 (gm * 100 + variant, gm = tile-order group size) for
 * every following ink_gemm_f16 call; -1 restores the built-in shape heuristic (-2: the same heuristic restricted to
 * one-tile-per-workgroup kernels, for whole-step A/B runs).  Results are identical across variants up to f32
 * summation order.  A variant that cannot take a call's form (e.g. 54 / 55: plain epilogues of the 256x320 tile only)
 * makes ink_gemm_f16 return INK_ERR_ARG.  Process-wide, not thread-safe, never set by the product path. */
int ink_gemm_set_variant(int32_t v);
/* Which tile variant the built-in heuristic picks for (M,N,K): 10 = 256x256x64 / 16 waves (the dominant kernel),
 * 0 = 128x128x64 / 4 waves, 32 = 128x128x32 (K % 64 != 0).  Pure host function, used by bench.py's roofline. */
int ink_gemm_query_variant(int32_t M, int32_t N, int32_t K);
int ink_gemm_query_stats_chunk(int32_t M, int32_t N, int32_t K);

/* ------------------------------------------------------------------------
 * Row LayerNorm with optional row gather (fuses window-partition / pad /
 * cyclic shift into the normalisation pass).
 *   out[r, :] = LN(x[gather[r], :]) * gamma + beta      (gather[r] < 0 -> zeros)
 * Replaces nn.LayerNorm + window_partition (SA/modeling/image_encoder.py:168-172,
 * :243-264) and norm1 + pad + roll + window_partition
 * (GD/.../swin_transformer.py:246-265).  x is f32; out is f16 and/or f32.
 * C % 4 == 0, C <= 2048.  act: INK_ACT_NONE or INK_ACT_GELU applied after the affine
 * (LayerNorm2d + GELU of SA/modeling/mask_decoder.py:54-56).
 * split = 1: out_f16 rows are SPLIT-f16 operands [hi | lo*64 | hi/64] of 3*C columns (ldo >= 3*C), see
 * ink_add_split_f16; an f32 copy may be written in the same pass: out_f32 rows then have stride ldo / 3.
 * add (f32 or NULL, not together with gather): the row normalised is x[r] + add[add_batch_rows[r / rows_per_batch] +
 * r % rows_per_batch] (add_batch_rows NULL: add[r]) - the residual add of SA/modeling/transformer.py:180-181 when the
 * image keys are still shared by all boxes of an image (no per-box copy of them is ever made).
 * --------------------------------------------------------------------- */
int ink_layernorm_rows(const float* x, int64_t ldx, const float* gamma, const float* beta,
                       float eps, const int32_t* gather, int32_t rows_out, int32_t C,
                       void* out_f16, float* out_f32, int64_t ldo, int32_t act, int32_t split,
                       const float* add, int64_t ld_add, const int32_t* add_batch_rows, int32_t rows_per_batch,
                       void* stream);

/* f32 -> f16 conversion with optional broadcast addend:  out[i] = f16(a[i] + b[i % n_b])
 * (b may be NULL).  n % 4 == 0, n_b % 4 == 0, n % n_b == 0.  Used for the "x + pos" operands of
 * GD/.../transformer.py:783 and SA/modeling/transformer.py:164-165,178-179 (keys + key_pe). */
int ink_add_cvt_f16(const float* a, const float* b, int64_t n_b, void* out_f16, int64_t n,
                    void* stream);
/* Split-f16 GEMM operand: v = a[i] + b[i % n_b] (f32, rows of C columns, contiguous) is written as three
 * f16 K-segments  out[r, 0:C] = hi = f16(v),  out[r, C:2C] = f16((v - hi) * 64),  out[r, 2C:3C] = f16(hi / 64)
 * (out is [n / C, 3*C]).  Multiplied by ink_gemm_f16 against weights laid out [W_hi | W_hi/64 | (W - W_hi)*64]
 * the f32 accumulator receives hi*W_hi + lo*W_hi + hi*W_lo, i.e. an fp32-grade product (~2^-21 relative) on the
 * f16 MFMA pipe.  Used for the layers whose f16 rounding dominates the mask error (SAM neck, prompt/mask decoder,
 * upscaler, hyper-network: SA/modeling/image_encoder.py:88-104, mask_decoder.py:112-149, transformer.py:62-240);
 * the reference computes these in fp32. */
int ink_add_split_f16(const float* a, const float* b, int64_t n_b, void* out_f16, int64_t n, int32_t C,
                      void* stream);
/* Same with an f32 result (src = image_embeddings + dense_prompt, SA/modeling/mask_decoder.py:124-125). */
int ink_add_f32(const float* a, const float* b, int64_t n_b, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * Fused softmax attention  O = softmax(scale * Q K^T + bias) V   (f16 in/out, f32 math),
 * never materialising the score matrix.  Rows of batch b start at b*n_q (Q, O) and
 * b*n_k (K, V); head h occupies columns [h*head_dim, (h+1)*head_dim) of each row, so the
 * packed qkv projection output is consumed in place (ldq = ldk = ldv = 3*dim).
 * bias_mode 0: none.
 * bias_mode 1: SAM global blocks (SA/modeling/image_encoder.py:231-237,325-361) on a 64x64
 *              token grid: bias[q,k] = scale*(rel_h[q, k/64] + rel_w[q, k%64]); rel_h/rel_w are
 *              f32 [n_batch*n_heads, n_q, 64] as produced by ink_relpos_bias.
 * bias_mode 2: SAM 14x14 windows: rel_aug f16 [n_batch*n_heads, n_q, 32] from ink_relpos_bias
 *              (cols 0..S-1 = rel_h, S..2S-1 = rel_w); grid_w = S <= 16.  At SAM's own size
 *              (193 <= n_k <= 208, i.e. S = 14) a dedicated kernel runs (csrc/attention_win.hip); it addresses
 *              rows through 32-bit byte offsets: the O rows it writes must lie within 2 GiB of O, the
 *              q / k / v rows it gathers within 4 GiB of Q / K / V.
 * bias_mode 3: Swin windows (GD/.../swin_transformer.py:148-167), n_q, n_k <= 64: dense_bias f32
 *              [n_heads, n_q, 64] (relative_position_bias_table gathered by relative_position_index)
 *              + optional dense_mask f32 [n_mask, n_q, 64] (the 0/-100 SW-MSA mask; batch entry b
 *              uses mask b % n_mask).  Both are PRE-DIVIDED by `scale` and row-padded to 64.
 * Supported head_dim: 80 (modes 0,1,2), 32 (modes 0,3) and 16 (mode 0).
 * q_batch_rows / kv_batch_rows (int32 [n_batch], optional): first row of batch entry b in Q/O and
 * in K/V; NULL means b*n_q and b*n_k.  Lets many batch entries share one K/V (or Q) block, e.g.
 * SAM decoder layer 0 where the image keys are identical for all boxes of an image.
 * tok_rows (int32 [n_batch, n_q], optional, bias_mode 2 only, n_q == n_k): token i of batch entry
 *   (window) b lives in row tok_rows[b*n_q + i] of Q, K, V AND O, or is window padding (-1).  This
 *   is window_partition / window_unpartition (SA/modeling/image_encoder.py:243-289) folded into
 *   the attention: a padding token is skipped as a query and contributes the rows pad_k / pad_v
 *   (f16 [n_heads*head_dim]: the k and v slices of the qkv bias, i.e. qkv(0)) as a key, exactly
 *   what the reference computes for its zero-padded rows - without ever projecting them.
 * --------------------------------------------------------------------- */
typedef struct InkAttn {
  const void* Q; const void* K; const void* V;   /* f16 */
  void* O;                                        /* f16 */
  int64_t ldq, ldk, ldv, ldo;                     /* row strides in elements */
  int32_t n_batch, n_heads, n_q, n_k, head_dim;
  float scale;
  int32_t bias_mode;
  int32_t grid_w;
  const int32_t* q_batch_rows;
  const int32_t* kv_batch_rows;
  const float* rel_h; const float* rel_w;         /* mode 1 (f16 tables when rel_f16 != 0: ink_relpos_bias64_f16) */
  const void* rel_aug;                            /* mode 2 */
  const float* dense_bias; const float* dense_mask; /* mode 3 */
  int32_t n_mask; int32_t rel_f16;                /* rel_f16: 0 / 1, SAM's own shape only (was padding: 0 in older callers) */
  const int32_t* tok_rows;                        /* mode 2, optional */
  const void* pad_k; const void* pad_v;           /* f16 rows used for tok_rows == -1 keys */
} InkAttn;
int ink_flash_attn(const InkAttn* p, void* stream);

/* ink_relpos_bias for the 64 x 64 grid with f16 output tables [n_batch*n_heads*4096, 64] (InkAttn.rel_f16 = 1). */
int ink_relpos_bias64_f16(const void* Q, int64_t ldq, const float* rel_pos_h, const float* rel_pos_w, int32_t n_batch,
                          int32_t n_heads, int32_t head_dim, float scale, void* out_h_f16, void* out_w_f16, void* stream);

/* Decomposed relative-position terms of SA/modeling/image_encoder.py:292-361
 * (get_rel_pos + the two einsums of add_decomposed_rel_pos), divided by `scale`:
 *   rel_h[bh, q, j] = (Q[b, q, h, :] . rel_pos_h[q_h - j + S - 1, :]) / scale   (same for w).
 * rel_pos_h / rel_pos_w: f32 [2S-1, head_dim].  S == 64 writes out_h/out_w (f32 [.., 64]);
 * S <= 16 writes out_aug_f16 ([.., 32], rel_h then rel_w then zeros).
 * tok_rows (optional, S <= 16 only): as in InkAttn - query i of window b is row tok_rows[b*S*S+i]
 * of Q (-1: padding, its output row is left untouched). */
int ink_relpos_bias(const void* Q, int64_t ldq, const float* rel_pos_h, const float* rel_pos_w,
                    int32_t S, int32_t n_batch, int32_t n_heads, int32_t head_dim, float scale,
                    const int32_t* tok_rows, float* out_h, float* out_w, void* out_aug_f16,
                    void* stream);

/* Sam.preprocess + PatchEmbed gather (SA/modeling/sam.py:164-174, image_encoder.py:364-395):
 * image_u8 is the ResizeLongestSide output, HWC uint8 [h, w, 3] (h, w <= L); writes the f16
 * im2col matrix [ (L/P)^2, 3*P*P ] (column = c*P*P + ky*P + kx, matching proj.weight.view(D,-1)),
 * with (x - mean[c]) / std[c] applied and the bottom/right padding left at 0.
 * mean3 / std3 are HOST pointers.  chan_reverse != 0 reads channel 2-c (the BGR/RGB quirk of
 * InkLayer/segmentor/sam.py:24-26 without an extra host copy).  split = 1: rows are split-f16 operands
 * [hi | lo*64 | hi/64] of 3*(3*P*P) columns (see ink_add_split_f16): the patch embedding is the one ViT-H
 * projection whose rounding error stays in the residual stream of all 32 blocks. */
int ink_sam_patchify(const void* image_u8, int32_t h, int32_t w, int32_t L, int32_t P,
                     const float* mean3, const float* std3, int32_t chan_reverse, int32_t split,
                     void* out_f16, void* stream);

/* Pillow's antialiased bilinear resize of an HWC uint8 RGB image, bit for bit: the `F.resize` of
 * load_image (GD/util/inference.py:39-50, GD/datasets/transforms.py:87-117: shorter side 800, max 1333) and
 * ResizeLongestSide.apply_image (SA/utils/transforms.py:26-31, 93-102), both of which end in PIL
 * Image.resize(BILINEAR).  src [h, w, 3] -> dst [oh, ow, 3].  xbounds/ybounds: int32 [ow|oh, 2] = (first
 * input sample, number of samples); xcoef/ycoef: int32 [ow|oh, kx|ky] 22-bit fixed-point weights, both
 * exactly as Pillow's precompute_coeffs + normalize_coeffs_8bpc produce them (inklayer_amd/resize.py).
 * Horizontal pass first, rounded to u8 into tmp [h, ow, 3] (needed only when both sizes change), then
 * the vertical pass.  At least one of the two sizes must differ. */
int ink_resize_bilinear_u8(const void* src_u8, int32_t h, int32_t w, void* dst_u8, int32_t oh, int32_t ow,
                           const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                           const int32_t* ybounds, const int32_t* ycoef, int32_t ky, void* tmp_u8,
                           void* stream);

/* 3x3 / pad-1 im2col of an NHWC f16 map [B,H,W,C] -> [B*H*W, 9*C] with column (ky*3+kx)*C + c
 * (neck conv SA/modeling/image_encoder.py:96-103; GD input_proj 3x3 s2 uses the strided form). */
int ink_im2col3x3_f16(const void* in_f16, int32_t B, int32_t H, int32_t W, int32_t C, void* out_f16,
                      void* stream);

/* PositionEmbeddingRandom._pe_encoding (SA/modeling/prompt_encoder.py:186-193):
 * out[n] = [sin(2*pi*((2c-1) @ G)), cos(2*pi*((2c-1) @ G))], coords01 f32 [N,2] in [0,1]^2,
 * G f32 [2,F], out f32 [N,2F]; if `add` (f32 [n_add, 2F]) is given, out[n] += add[n % n_add]
 * (the learned corner embeddings).  Serves _embed_boxes (:93-100) and get_dense_pe (:62-71). */
int ink_sam_pe_encode(const float* coords01, const float* gauss, int32_t N, int32_t F,
                      const float* add, int32_t n_add, float* out, void* stream);

/* masks = hyper_in @ upscaled_embedding for ONE mask token (SA/modeling/mask_decoder.py:139-144),
 * reading the ConvTranspose output in its un-shuffled GEMM layout
 * up[((b*g*g + y*g + x)*4 + (dy1*2+dx1))*4 + (dy2*2+dx2), C] and writing the pixel-shuffled
 * low-res logits out[b, 4y+2dy1+dy2, 4x+2dx1+dx2] (f32 [n, 4g, 4g]).  C in {32, 8}. */
int ink_sam_mask_logits(const float* up, const float* hyper, int32_t n, int32_t g, int32_t C,
                        float* out, void* stream);

/* Sam.postprocess_masks + `> mask_threshold` (SA/modeling/sam.py:133-162, SA/predictor.py:238-241)
 * fused: bilinear S->L (align_corners=False), crop to [in_h,in_w], bilinear to [out_h,out_w],
 * compare.  out_u8 [n,out_h,out_w] gets 0/1; out_logits (optional, f32) the un-thresholded value. */
int ink_sam_postprocess(const float* low, int32_t n, int32_t S, int32_t L, int32_t in_h,
                        int32_t in_w, int32_t out_h, int32_t out_w, float thr, void* out_u8,
                        float* out_logits, void* stream);

/* ------------------------------------------------------------------------
 * Multi-scale deformable attention forward — the reference's ONLY native op.
 * ink_ms_deform_attn_forward mirrors groundingdino._C.ms_deform_attn_forward
 * (GD/models/GroundingDINO/csrc/vision.cpp:53-56, MsDeformAttn/ms_deform_attn.h:21-40,
 * ms_deform_attn_cuda.cu:21-81): value f32 [B,S,M,C], spatial_shapes int64 [L,2] (h,w),
 * level_start_index int64 [L], sampling_loc f32 [B,Q,M,L,P,2] (x,y in [0,1]), attn_weight f32
 * [B,Q,M,L,P], im2col_step (only validated: B % min(B, step) == 0) -> out f32 [B,Q,M*C].
 * The two int64 arrays are HOST pointers here (the reference reads them on the device; they are
 * 8 numbers known to the caller).  C must be 32.
 * --------------------------------------------------------------------- */
int ink_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes_host,
                               const int64_t* level_start_index_host, const float* sampling_loc,
                               const float* attn_weight, int32_t B, int32_t S, int32_t M, int32_t C,
                               int32_t Q, int32_t L, int32_t P, int32_t im2col_step, float* out,
                               void* stream);

/* Pipeline form of the same op (MultiScaleDeformableAttention.forward, ms_deform_attn.py:282-352,
 * minus the three Linear layers): value f16 [B,S,8,32]; proj f32 [B*Q, ldp] with columns
 * [0,256) = sampling_offsets(query) as (head,level,point,xy) and [256,384) = attention_weights(query)
 * as (head,level,point); softmax over the 16 (level,point) logits, the sampling-location arithmetic
 * (2-d refs: ref + off/(W_l,H_l); 4-d refs: ref_xy + off/4 * ref_wh * 0.5) and the bilinear
 * gather are fused; out f16 [B*Q, 256].  ref: f32, element (b,q) at b*ref_b_stride + q*ref_q_stride,
 * the same for every level (valid_ratios == 1).  shapes_host: int32 [4,2] (h,w), HOST pointer. */
int ink_msda_fused(const void* value_f16, const float* proj, int64_t ldp, const float* ref,
                   int32_t ref_dim, int64_t ref_q_stride, int64_t ref_b_stride,
                   const int32_t* shapes_host, int32_t B, int32_t S, int32_t Q, void* out_f16,
                   void* stream);

/* load_image normalisation (GD/util/inference.py:40-49: /255, -mean, /std) + Swin PatchEmbed 4x4/s4
 * gather (swin_transformer.py:480-489): u8 HWC [h,w,3] -> f16 [ceil(h/4)*ceil(w/4), 64]
 * (48 real columns c*16+ky*4+kx, 16 zero columns).  mean3/std3 are HOST pointers. */
int ink_swin_patchify(const void* image_u8, int32_t h, int32_t w, const float* mean3,
                      const float* std3, void* out_f16, void* stream);

/* PatchMerging (swin_transformer.py:314-340): out[r] = LN(concat of the 4 source rows gather4[r][0..3])
 * in f16 [rows, 4C]; index -1 = zero row (odd-size padding).  C <= 1024. */
int ink_layernorm_merge4(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                         const int32_t* gather4, int32_t rows, int32_t C, void* out_f16, void* stream);

/* nn.GroupNorm(G, C) on NHWC tokens x f32 [B,T,C] (input_proj, groundingdino.py:121-151);
 * stats_ws: f32 [B*G*2] scratch; out f32, batch b written at out + b*out_batch_stride. */
int ink_groupnorm_nhwc(const float* x, int32_t B, int32_t T, int32_t C, int32_t G, const float* gamma,
                       const float* beta, float eps, float* stats_ws, float* out,
                       int64_t out_batch_stride, void* stream);

/* Row gather of f32 rows (batch b reads row idx[b*idx_batch_stride + r] of x + b*x_batch_rows rows;
 * -1 -> zeros) into f16 and/or f32 [B*rows_per_batch, C]: masked_fill of invalid proposals
 * (GD/.../utils.py:111-113) and the top-k gathers of transformer.py:302-316. */
int ink_gather_rows(const float* x, int64_t ldx, int64_t x_batch_rows, const int32_t* idx,
                    int64_t idx_batch_stride, int32_t rows_per_batch, int32_t B, int32_t C,
                    void* out_f16, float* out_f32, void* stream);

/* BiMultiHeadAttention core (GD/.../fuse_modules.py:168-240), both directions, no score matrix in
 * HBM beyond [B,S,4,T] f32:  QV f16 [B*S, 2E] = [v_proj(v) | values_v_proj(v)],
 * KL f16 [B*T, 2E] = [l_proj(l) | values_l_proj(l)], E = 1024 (4 heads x 256), T <= 4.
 * out_v f16 [B*S, E] (softmax over text), out_l f16 [B*T, E] (softmax over image tokens).
 * Workspaces: scores_ws f32 [B*S*4*T], stats_ws f32 [B*4*T*2], partial_ws f32 [B*4*ceil(S/chunk)*T*256]. */
int ink_biattn_fusion(const void* QV_f16, const void* KL_f16, int32_t B, int32_t S, int32_t T, int32_t E,
                      float scale, float* scores_ws, float* stats_ws, float* partial_ws, int32_t chunk,
                      void* out_v_f16, void* out_l_f16, void* stream);

/* softmax(scale q k^T [+ blocked -> -inf]) v against n_k <= 16 keys; head h at columns
 * [h*hd,(h+1)*hd), hd in {16,32,64}; blocked: u8 [n_q, n_k] (1 = not allowed) or NULL.  io_f32 = 0: Q/K/V/O are
 * f16 rows, 1: f32 rows (ld* in elements either way; the math is f32 in both).  q_batch_rows (int32 [B] or NULL):
 * first Q row of batch entry b (default b*n_q); keys and the output are dense.  q_add (f32 [n_q, n_heads*hd] or NULL) is
 * added to the query rows by position (the per-position constant pe.W of a projection of x + pe).
 * Text self-attention (transformer_vanilla.py:114-116), decoder text cross-attention (transformer.py:893-900) and,
 * with f32 rows, the SAM mask decoder's token self-attention and image -> token attention
 * (SA/modeling/transformer.py:151-182: 7 x 7 and 4096 x 7). */
int ink_attn_fewkeys(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                     int32_t B, int32_t n_q, int32_t n_k, int32_t n_heads, int32_t head_dim, float scale,
                     const uint8_t* blocked, const int32_t* q_batch_rows, const float* q_add, int32_t io_f32, void* O,
                     int64_t ldo, void* stream);

/* softmax(scale q k^T) v for n_q <= 8 queries per batch entry against MANY keys (SAM decoder tokens ->
 * image: 7 x 4096, SA/modeling/transformer.py:163-168): head h at columns [h*hd,(h+1)*hd),
 * hd in {16,32}; q_batch_rows / kv_batch_rows as in InkAttn; O dense [n_batch*n_q, ..].  io_f32 = 0: f16 rows,
 * 1: f32 rows (head_dim 16 with n_heads % 4 == 0 only).  k_add (f32 [n_k, n_heads*hd] or NULL, f32 rows only) is
 * added to the key rows by key position. */
int ink_attn_fewq(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  int32_t n_batch, int32_t n_q, int32_t n_k, int32_t n_heads, int32_t head_dim, float scale,
                  const int32_t* q_batch_rows, const int32_t* kv_batch_rows, const float* k_add, int32_t io_f32, void* O,
                  int64_t ldo, void* stream);

/* Column (max, sum-exp) over the S rows of B score matrices [S, HT] f32 (HT <= 16): stats f32 [B, HT, 2];
 * part_ws f32 [B * ceil(S / 512) * HT * 2].  The text-side softmax statistics of the fusion layer. */
int ink_biattn_colstats(const float* scores, int32_t B, int32_t S, int32_t HT, float* part_ws, float* stats, void* stream);

/* BiAttentionBlock (GD/.../fuse_modules.py:146-295) with the <= 4 text tokens of InkLayer's fixed caption FOLDED through
 * it: the image tokens v f32 [B*S, 256] are updated IN PLACE to  LN_v(v) + gamma_v * out_v_proj(attention over text)  and
 * out_l f16 [B*T, 1024] receives the text-side attention output (operand of out_l_proj), without ever forming the per-token
 * q / value_v projections (256 -> 2 x 1024) or the 1024 -> 256 image output projection - see csrc/fusion_fold.hip for
 * the algebra.  text_k / text_vl: f32 [B*T, >= 1024] (row stride ld_text) = l_proj / values_l_proj of LN_l(l);
 * Wq / Wvv f16 [1024, 256] (v_proj / values_v_proj weights), bq / bvv f32 [1024]; Wo f16 [256, 1024], bo f32 [256]
 * (out_v_proj); scale = 256^-0.5.  ws: f32[ink_fusion_fold_workspace(B, S)].  S <= 32768.  Optional (NULL to skip): out16 f16
 * [B*S, 256] = f16(updated v) and out16_pos = f16(updated v + pos[s]) with pos f32 [S, 256] - the operands of the
 * deformable attention's value / sampling projections (transformer.py:780-789), written in the same pass. */
int ink_fusion_fold_workspace(int32_t B, int32_t S, int64_t* out_floats);
int ink_fusion_fold(float* v_f32, int32_t B, int32_t S, const float* lnv_g, const float* lnv_b, float eps,
                    const float* text_k_f32, const float* text_vl_f32, int64_t ld_text, int32_t T, const void* Wq_f16,
                    const float* bq, const void* Wvv_f16, const float* bvv, const void* Wo_f16, const float* bo,
                    const float* gamma_v, float scale, float* ws, void* out_l_f16, const float* pos, void* out16_pos,
                    void* out16, void* stream);

/* Tail of the SAM mask decoder in one kernel (csrc/upscale_tail.hip): LayerNorm2d + GELU + the second ConvTranspose2d
 * (k2 s2 = a [64 -> 4 x 32] projection per row) + GELU + the hyper-network product of mask token 0
 * (SA/modeling/mask_decoder.py:54-60, 138-145).  u0 f32: the first transposed convolution's output, 4 x 64 floats per
 * (box, token) at token stride ld_tok (>= 256 floats: it may be a column block of a wider projection), row (token, s1); ln_g / ln_b f32 [64] (output_upscaling.1), eps 1e-6; blob: output_upscaling.3.weight as the
 * split-f16 matrix [128, 192] (rows (s2, c), see ink_add_split_f16) packed by ink_sam_upscale_pack (48 KiB); b3 f32 [128]
 * (the bias repeated per sub-pixel); hyper f32 [n, 32]; low f32 [n, 4g, 4g].  (g*g*4) % 32 == 0. */
int ink_sam_upscale_pack(const void* ws_f16, void* blob_f16, void* stream);
int ink_sam_upscale_tail(const float* u0, int64_t ld_tok, int32_t n, int32_t g, const float* ln_g, const float* ln_b, float eps,
                         const void* blob_f16, const float* b3, const float* hyper, float* low, void* stream);

/* Image-side tail of a layer of SAM's two-way transformer in one kernel (csrc/proj_ln.hip):
 *     out = LayerNorm(res + a W^T + bias)         a f32 [rows, 128] (the image->token attention's output), 256 columns
 * = cross_attn_image_to_token.out_proj + the residual + norm4 (SA/modeling/transformer.py:175-182).  The projection runs
 * on split-f16 operands (built in registers from a; blob = the weight as the split-f16 matrix [256, 384] packed by
 * ink_proj256_ln_pack, 192 KiB).  res f32 [.., 256]: row r, or - res_batch_rows given - row res_batch_rows[r /
 * rows_per_batch] + r % rows_per_batch of a tensor shared by the boxes of an image.  Outputs (either may be NULL): out_f32
 * [rows, 256] and out_split_f16 [rows, 768], the split-f16 operand [hi | lo*64 | hi/64] of the next projection. */
int ink_proj256_ln_pack(const void* ws_f16, void* blob_f16, void* stream);
int ink_proj256_ln(const float* a_f32, const void* blob_f16, const float* bias, const float* res_f32,
                   const int32_t* res_batch_rows, int32_t rows_per_batch, const float* ln_g, const float* ln_b, float eps,
                   int64_t rows, float* out_f32, void* out_split_f16, void* stream);

/* Feed-forward block of the deformable encoder layer, fused (csrc/ffn_fused.hip):
 *     out = LayerNorm(res + linear2(relu(linear1(x))))      d_model 256, d_ffn = hid (multiple of 64, <= 2048)
 * = DeformableTransformerEncoderLayer.forward_ffn + norm2 (GD/models/GroundingDINO/transformer.py:780-799).  x f16 [M, 256]
 * (row stride ldx) is the GEMM operand (f16 of the layer's input), res f32 [M, 256] the same rows in f32 (the residual);
 * out f32 [M, 256] may alias res.  The hidden activations are rounded to f16 between the two products, as in the two-GEMM
 * form.  blob: the weights packed by ink_ffn256_pack (ink_ffn256_pack_bytes(hid) bytes, 16-B aligned) from linear1.weight
 * f16 [hid, 256], linear1.bias f32 [hid] (carried as an f16 hi + lo pair in a 17th k-step, i.e. to 2^-22) and
 * linear2.weight f16 [256, hid]: per 64 hidden units the LDS image the kernel streams - 1-KiB MFMA operand blocks,
 * linear2's columns permuted inside 16-blocks to the order phase A leaves them in registers.  b2: linear2.bias. */
int ink_ffn256_pack_bytes(int32_t hid, int32_t with_pre, int64_t* out_bytes);
int ink_ffn256_pack(const void* w1_f16, const float* b1, const void* w2_f16, int32_t hid, const void* wpre_f16, void* blob,
                    void* stream);
int ink_ffn256_fused(const void* x_f16, int64_t ldx, const float* res_f32, const void* blob, const float* b2,
                     const float* ln_g, const float* ln_b, float eps, int32_t M, int32_t hid, const float* pre_bias,
                     const float* pre_ln_g, const float* pre_ln_b, float* out_f32, void* stream);
/* With a PRECEDING projection (wpre_f16 f16 [256, 256] given to the pack, pre_bias / pre_ln_g / pre_ln_b f32 [256] to the
 * call; all NULL = the plain form above): x is the input of that projection and the kernel computes
 *     s = LayerNorm_pre(res + x wpre^T + pre_bias);   out = LayerNorm(s + linear2(relu(linear1(f16(s)))))
 * = the deformable attention's output projection + residual + norm1 + the feed-forward block + norm2 of
 * DeformableTransformerEncoderLayer.forward (transformer.py:780-799) in one launch; s never leaves the registers. */

/* Two-stage query selection (transformer.py:293-300): indices of the K largest max_t logits[b,s,t],
 * descending, ties -> lower index.  logits f32 [B,S,T]; out_idx int32 [B,K].  S <= 16384 is one LDS
 * bitonic sort per image; larger S (800x1333 inputs give 22223 tokens) sorts ceil(S/16384) equal chunks and
 * merges their K best: cand_ws must then hold B * nchunk * K uint64. */
int ink_topk_rowmax(const float* logits, int32_t B, int32_t S, int32_t T, int32_t K, int32_t* out_idx,
                    float* out_val, void* cand_ws, void* stream);

/* gen_sineembed_for_position (GD/.../utils.py:204-230) for boxes ref f32 [N,4] -> f16 [N,512];
 * dim_t f32 [128] = 10000^(2*(i//2)/128). */
int ink_sine_embed4(const float* ref, const float* dim_t, int32_t N, void* out_f16, void* stream);

/* out = sigmoid(delta[:, :4] + inverse_sigmoid(ref))  (transformer.py:716-722, util/misc.py:704-708);
 * ref_is_logit != 0: ref is already un-sigmoided (two-stage anchors, transformer.py:296-305). */
int ink_box_refine(const float* delta, int64_t ldd, const float* ref, int32_t N, int32_t ref_is_logit,
                   float* out, void* stream);

/* ------------------------------------------------------------------------
 * Mask hand-off stage of the refinement, on resident masks.  Integer / byte work, bit-exact with the
 * reference's cv2 results.
 *
 * ink_mask_cleanup = clean_up_mask of InkLayer/refinement/mask_cleaner.py:11-36 for n masks at once:
 *   cv2.threshold(m, 127, 255) -> cv2.morphologyEx(MORPH_CLOSE, k x k rect, k odd = calculate_kernel_size(shape),
 *   mask_cleaner.py:6-9) -> cv2.connectedComponentsWithStats(connectivity=8) -> keep components with
 *   area > area_threshold (500) OR max(w,h) / (min(w,h) + 1e-5) > aspect_threshold (1.1).
 * masks_u8 / out_u8: [n, H, W] uint8 (any value > 127 is foreground on input; 0 / 255 on output), the masks stay in
 * HBM instead of travelling through masks/mask_i.png -> masks_cleaned/mask_i.png (runner.py:57-60,69).
 * tmp_a / tmp_b: [n, H, W] uint8 scratch; workspace: int32[ink_mask_cleanup_workspace_ints(n, H, W, k)].
 * H, W <= 16383. */
int ink_mask_cleanup_workspace_ints(int32_t n, int32_t H, int32_t W, int32_t k, int64_t* out_ints);
int ink_mask_cleanup(const void* masks_u8, int32_t n, int32_t H, int32_t W, int32_t k, int32_t area_threshold,
                     double aspect_threshold, void* tmp_a_u8, void* tmp_b_u8, int32_t* workspace, void* out_u8,
                     void* stream);

/* Pair table of the sketch NMS (content_iou, InkLayer/refinement/nms_sketch.py:186-234, which re-opens the sketch and
 * two mask PNGs PER PAIR): refined_m = (mask_m > 0) AND (PIL-luma(sketch) < 250) (refine_mask_to_sketch_regions,
 * nms_sketch.py:62-78); counts[i, j] = (|refined_i AND refined_j|, |refined_i OR refined_j|) as int32 [n, n, 2].
 * sketch_rgb_u8: [H, W, 3] (the masks have the sketch's size on this path); bits_ws: uint64[n * ceil(H*W/64)]. */
int ink_mask_sketch_iou_counts(const void* masks_u8, const void* sketch_rgb_u8, int32_t n, int32_t H, int32_t W,
                               void* bits_ws_u64, int32_t* counts, void* stream);

/* The SAM ViT-H residual stream as two f16 planes (x = hi + lo, hi = f16(x), lo = f16(x - hi): ~22 significant bits),
 * so that the hi plane IS the f16 operand of the next projection and LayerNorm folds into that projection
 * (InkGemm.ln_stats; SA/modeling/image_encoder.py:166-182).
 * ink_hilo_split_stats: f32 rows [rows, C] -> hi / lo planes (row stride ldo) + stats [rows, C / chunk, 2] = (sum, sum of
 * squares) per chunk of `chunk` columns, the layout InkGemm.stats_out / ln_stats use.  ink_hilo_join: out = hi + lo. */
int ink_hilo_split_stats(const float* x, int64_t ldx, int32_t rows, int32_t C, void* hi_f16, void* lo_f16, int64_t ldo,
                         float* stats, int32_t chunk, void* stream);
int ink_hilo_join(const void* hi_f16, const void* lo_f16, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------
 * Refinement stage on resident masks: depth ordering support, disjoint parsing, mask growth, box
 * assignment support, the "unlabeled" extra mask.  Integer / bit work, bit-exact with the reference.
 * Binary images are ROW-ALIGNED BIT PLANES: Wp = ceil(W / 64) uint64 words per row, bit b of word w of row y = pixel
 * (y, 64 w + b), tail bits 0; n planes are [n, H, Wp].  Mask sets after the disjoint parsing are ONE uint8 label image
 * [H, W] (label l = mask index l - 1, 0 = no mask; at most 254 masks).  H, W <= 16383.
 *
 * ink_refine_sketch_planes: sketch RGB u8 [H, W, 3] -> 4 planes
 *   0: sketch_to_01binary of the cv2 BGR image (InkLayer/refinement/utils.py:3-9): blue <= max(all bytes) / 2
 *   1: PIL luma <  250 (refiner.py:106-110)        2: PIL luma <= 250 (refiner.py:134, 246)
 *   3: cv2 gray <  250 (refiner.py:302-303)        max_ws: int32[1] scratch. */
int ink_refine_sketch_planes(const void* rgb_u8, int32_t H, int32_t W, void* planes4_u64, int32_t* max_ws, void* stream);

/* uint8 images [n, H, W] -> planes (pixel > thresh). */
int ink_bitplane_pack(const void* img_u8, int32_t n, int32_t H, int32_t W, int32_t thresh, void* planes_u64, void* stream);

/* get_mask_depth_score's inputs (depth_sort.py:72-89): vals[p] = depth[pts[p]] and inside[m, p] = mask m covers sample p
 * (uint8 [n, P]); pts_yx int32 [P, 2] (y, x).  The binned mode itself is a few hundred numbers: host. */
int ink_refine_depth_samples(const void* mask_planes, int32_t n, const float* depth, const int32_t* pts_yx, int32_t P,
                             int32_t H, int32_t W, float* vals, void* inside_u8, void* stream);

/* All pair / per-mask counts the ordering and the disjoint parsing need, in one pass over the planes:
 *   D_m = cv2.dilate(mask_m & plane 0, 3x3 ellipse) (depth_sort.py:193-197)
 *   pair[i, j] = (|M_i & M_j| over the image (refiner.py:71-74), |D_i & D_j| inside rect[i, j] (depth_sort.py:222-231))
 *   per_mask[m] = (|M_m| (refiner.py:41), |D_m| (depth_sort.py:200), |M_m & plane 1| (refiner.py:108-110))
 *   sketch_area = |plane 1| (refiner.py:106)
 * rect int32 [n, n, 4] = (y0, y1, x0, x1): the pair's box intersection as RESOLVED numpy slice bounds (half open,
 * inside the image; the host applies numpy's rules for negative / oversized indices).  dil_ws: n planes of scratch.
 * pair int32 [n, n, 2], per_mask int32 [n, 3]. */
int ink_refine_pair_tables(const void* mask_planes, const void* sketch_planes4, int32_t n, int32_t H, int32_t W,
                           const int32_t* rect, void* dil_ws_planes, int32_t* pair, int32_t* per_mask,
                           int32_t* sketch_area, void* stream);

/* composite_and_parse_masks' layering (refiner.py:44-47): label = 1 + the first rank r whose mask order[r] covers the
 * pixel (order[r] < 0: rank emptied by the "covers the whole sketch" rule, refiner.py:104-112); hist256[l] = pixels. */
int ink_refine_composite(const void* mask_planes, const int32_t* order, int32_t n_ranks, int32_t H, int32_t W,
                         void* label_u8, int32_t* hist256, void* stream);

/* out = clean_delicate_mask (refiner.py:21-33) of every mask of map256[label]: a pixel with at most one 8-neighbour of
 * its own (mapped) label is cleared.  map256: uint8[256] on the device (0 drops a label). */
int ink_refine_relabel_clean(const void* label_u8, const void* map256_u8, int32_t H, int32_t W, void* out_label_u8,
                             void* stream);

/* refine_masks_with_watershed (refiner.py:129-196) on the label image: unlabeled stroke pixels (plane 2, label 0),
 * their closing by disk(3), its 4-connected components of more than 50 pixels ("large regions"), masks within disk(3)
 * of one grow by disk(3) (flags256[l] = 1), the others by disk(2); where several masks reach a pixel the largest label
 * wins; everything restricted to plane 2.  Also returns what refine_masks_with_boxes needs: bbox256x4[l] = (xmin, ymin,
 * xmax, ymax) of grown mask l ((W, H, -1, -1) if empty) and the still unlabeled stroke pixels in raster order
 * (unl_yx int32 [unl_cap, 2], *unl_count = their number, possibly > unl_cap).
 * planes_ws / cc_ws sizes: ink_refine_grow_workspace. */
int ink_refine_grow_workspace(int32_t H, int32_t W, int64_t* plane_words, int64_t* cc_ints);
int ink_refine_grow(const void* label_u8, const void* sketch_planes4, int32_t H, int32_t W, void* planes_ws,
                    int32_t* cc_ws, int32_t* flags256, void* out_label_u8, int32_t* bbox256x4, int32_t* unl_yx,
                    int32_t unl_cap, int32_t* unl_count, void* stream);

/* d2[q, l] = squared distance from pixel q_yx[q] to the nearest pixel of label l for the labels in cand (4 uint64 per
 * query = a 256-bit set), 0x7fffffff where the label has no pixel (`np.min(distances)` of refiner.py:277-281). */
int ink_refine_query_dists(const void* label_u8, const int32_t* q_yx, const void* cand256_bits, int32_t Q, int32_t H,
                           int32_t W, int32_t* d2_Qx256, void* stream);

/* Applies the assignments (y, x, label) of the raster-order loop to the label image (in place), then
 * create_unlabeled_mask (refiner.py:301-337): plane 3 minus the labelled pixels, cv2 MORPH_OPEN 3x3, cv2.dilate 2x2
 * -> extra_plane, *extra_count = its pixel count.  planes_ws3: 3 planes of scratch. */
int ink_refine_finalize(void* label_u8, const int32_t* assign_yxl, int32_t A, const void* sketch_planes4, int32_t H,
                        int32_t W, void* planes_ws3, void* extra_plane, int32_t* extra_count, void* stream);

/* HOST functions (host pointers, no GPU work): the two sequential algorithms of the stage.
 * ink_host_sparse_sample = sparse_sketch_sample (depth_sort.py:49-68) of the sketch (RGB u8 [H, W, 3]): greedy
 * row-major thinning of the plane-0 pixels with radius 0.01 H; out_yx int32 [cap, 2], *count = samples found.
 * ink_host_assign_unlabeled = the pixel loop of refine_masks_with_boxes (refiner.py:262-295): q_yx the unlabeled stroke
 * pixels in raster order, boxes int32 [nb, 4] (inclusive), box2mask[nb] (-1: unmatched), d2 int32 [Q, 256] from
 * ink_refine_query_dists, nonempty uint8 [n_masks]; out_label[q] = 1 + mask index, 0 = stays unlabeled. */
int ink_host_sparse_sample(const uint8_t* rgb_u8, int32_t H, int32_t W, int32_t* out_yx, int32_t cap, int32_t* count);
int ink_host_assign_unlabeled(const int32_t* q_yx, int32_t Q, const int32_t* boxes, int32_t nb, const int32_t* box2mask,
                              const int32_t* d2, const uint8_t* nonempty, int32_t n_masks, int32_t* out_label);

/* ------------------------------------------------------------------------
 * Depth-Anything-V2 ViT-B pixel-side ops (the dense layers reuse ink_gemm_f16 / ink_flash_attn /
 * ink_layernorm_rows).
 *
 * ink_depth_patchify = DepthAnythingV2.image2tensor (DA/dpt.py:199-221: BGR->RGB, /255, cv2.resize INTER_CUBIC to
 * (nw, nh), (x - mean) / std) fused with the 14x14 / s14 patch gather of DA/dinov2_layers/patch_embed.py:
 * image_u8 [H, W, 3] -> split-f16 im2col rows [ (nh/P)*(nw/P), 3*KP ], KP = 3*P*P rounded up to a multiple of 32
 * (zero columns), column = c*P*P + ky*P + kx.  chan_reverse != 0 reads channel 2-c (cv2.imread's BGR order).
 * mean3 / std3 are HOST pointers.  float64 arithmetic like cv2 on a CV_64F image. */
int ink_depth_patchify(const void* image_u8, int32_t H, int32_t W, int32_t nh, int32_t nw, int32_t P, int32_t KP,
                       const double* mean3, const double* std3, int32_t chan_reverse, void* out_f16, void* stream);

/* F.interpolate(mode="bilinear", align_corners=True) on an NHWC f32 map [B, h, w, C] -> [B, H, W, C] (f32 and/or
 * f16 copy): the up-samplings of FeatureFusionBlock (DA/util/blocks.py:136-146), of DPTHead.forward
 * (DA/dpt.py:146) and of infer_image (DA/dpt.py:195).  C % 4 == 0 or C < 4. */
int ink_resize_bilinear_ac_nhwc(const float* in, int32_t B, int32_t h, int32_t w, int32_t C, int32_t H, int32_t W,
                                float* out_f32, void* out_f16, void* stream);

/* 3x3 / pad 1 im2col of an NHWC f16 map with stride 1 or 2 and an optional ReLU applied to the gathered values
 * (ResidualConvUnit's activation(x) before conv1, DA/util/blocks.py:69-70; the stride-2 resize conv of
 * DA/dpt.py:72-77): [B*H*W, C] -> [B*OH*OW, 9*C], OH = (H - 1) / stride + 1. */
int ink_im2col3x3_ex_f16(const void* in_f16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, int32_t relu,
                         void* out_f16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INKLAYER_HIP_H */

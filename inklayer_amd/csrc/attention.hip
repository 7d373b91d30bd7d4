// Fused (flash-style) attention for gfx950, f16 in / f32 accumulate / f16 out.
//
// One wave64 owns 32 query rows; a workgroup of NW waves shares 64-key K/V tiles in LDS.
//   S^T = K Q^T      mfma_f32_32x32x16_f16(A = K rows, B = Q^T): the lane (q = lane&31) ends up
//                    with 16 keys of ITS query per 32-key sub-tile -> softmax max/sum are
//                    in-lane (+ one cross-half shuffle), no LDS round trip.
//   O^T = V^T P^T    the S^T accumulator is used directly as the B operand (k order permuted
//                    as cdna_hip_programming.md §3 "accumulator tile as the next MFMA's
//                    operand"); V^T fragments come from the row-major V tile with
//                    ds_read_b64_tr_b16 (hardware transpose), so O^T has the query on the
//                    lane: alpha/l rescaling is lane-local.
// Bias modes (SAM decomposed relative position, SA/modeling/image_encoder.py:325-361):
//   1 "row tile"  : global attention on a 64-wide token grid; a 64-key tile is exactly one
//                   key row, so rel_w[q, kw] sits in 32 registers for the whole kernel and
//                   rel_h[q, kh] is one scalar per tile; both are folded into the accumulator
//                   INIT (no per-score VALU work).
//   2 "augmented" : 14x14 windows; the bias rides in the MFMA: Q' = [q | rel(q,.)/scale],
//                   K' = [k | onehot(kh), onehot(kw)] (two extra 16-wide k-steps).  This instance (ALLKV) keeps
//                   all 196 keys of a window in LDS, is persistent over (window, head) blocks with register
//                   prefetch of the next block, and can gather/scatter token rows (tok_rows) so that SAM's
//                   window partition / padding never materialises.
// Streamed K/V (all other instances): 64-key tiles alternate between two LDS buffers, one barrier per tile.
#include <stdlib.h>
#include "common.h"
#include "../../include/inklayer_hip.h"

int ink_win4_attn_launch(const InkAttn& p, int n_cus, hipStream_t s);   // attention_win.hip
int ink_glob4_attn_launch(const InkAttn& p, hipStream_t s);              // attention_glob.hip

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ f16x4 tr_read(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  return __builtin_bit_cast(f16x4, v);
}

__device__ __forceinline__ f16x8 cvt8(const f32x16& s, int base) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (f16)s[base + j];
  return r;
}

__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int HD, int MODE, int NW, bool ALLKV = false>
__global__ __launch_bounds__(NW * 64) void flash_attn_kernel(InkAttn p) {
  constexpr int NQKB = HD / 16;
  constexpr int NQK = NQKB + (MODE == 2 ? 2 : 0);
  constexpr int NB = (HD + 31) / 32;
  constexpr int DVP = NB * 32;
  constexpr bool LSUM_MFMA = DVP >= HD + 8;     // room for a ones-column in V's padding -> l = P @ 1 for free
  constexpr int CH = HD / 8;                    // 16-B chunks of real data per K/V row
  constexpr int KROW = ((NQK * 2) | 1) * 16;    // odd chunk count -> conflict-free b128 reads
  constexpr int VROW = DVP * 2;                 // 64 or 192 B: conflict-free tr_b16 reads
  constexpr int NT = NW * 64;
  constexpr int KIT = (64 * CH + NT - 1) / NT;
  constexpr float NEG = -1e30f;
  constexpr int TILEB = 64 * KROW + 64 * VROW;   // LDS bytes of one 64-key K/V tile
  constexpr int MAXT = ALLKV ? 4 : 2;             // ALLKV: every tile (n_k <= 256) resident at once; else a double buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 64 * KROW;
  char* sR = smem + MAXT * TILEB;                 // MODE 1: rel_w rows, [8 float4][NT lanes]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  const int nqb = (p.n_q + NW * 32 - 1) / (NW * 32);
  // ALLKV (SAM windows) is PERSISTENT: workgroup w walks blocks [w*total/G, (w+1)*total/G) - consecutive heads of
  // the same few windows - and loads the K/V rows and the Q / rel-pos fragments of block i+1 into registers while
  // it computes block i, so the load -> barrier -> compute -> store chain of a (window, head) no longer runs
  // serially at one workgroup per CU.  Everything else: one block per workgroup.
  const int total = p.n_batch * p.n_heads * nqb;
  // Global attention (MODE 1): the nqb = 16 query-tile workgroups of one (image, head) all stream the same 1.3 MB of
  // K/V.  Hardware dispatch deals consecutive workgroups round-robin to the 8 XCDs, which put every head's K/V through
  // every L2 (measured round 1: 4.4 GB fetched for 335 MB algorithmic, L2 hit 52 %); xcd_remap gives each XCD a
  // contiguous range of logical blocks, so the 16 tiles of a head run on ONE XCD, adjacent in time, and its K/V is
  // fetched from HBM once.
  int blk = ALLKV ? (int)((int64_t)blockIdx.x * total / gridDim.x)
                  : (MODE == 1 ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x);
  const int blk_end = ALLKV ? (int)((int64_t)(blockIdx.x + 1) * total / gridDim.x) : blk + 1;

  // Q^T fragments (B operand): lane (col = q, half hh) holds Q[q][16 s + 8 hh + j]; + the rel-pos columns (mode 2)
  struct QState { bool ok; int64_t row; };
  // ALLKV + tok_rows: the token rows of a window depend on the window only, not on the head, and a workgroup walks
  // consecutive heads of the same window, so they are cached in registers (wrow_q, wrow_k) and re-read only when
  // the window changes - the prefetch of the next block then has no dependent load in it
  constexpr int KALL = ALLKV ? (MAXT * 64 * CH + NT - 1) / NT : 1;
  int wrow_q = 0, wrow_k[KALL], wrow_b = -1;
  auto load_wrows = [&](int b_) {
    if (MODE == 2 && ALLKV && p.tok_rows && b_ != wrow_b) {
      const int qi = wave * 32 + lq;
      wrow_q = p.tok_rows[(int64_t)b_ * p.n_q + (qi < p.n_q ? qi : p.n_q - 1)];
#pragma unroll
      for (int it = 0; it < KALL; ++it) {
        const int key = (tid + it * NT) / CH;
        wrow_k[it] = key < p.n_k ? p.tok_rows[(int64_t)b_ * p.n_k + key] : -1;
      }
      wrow_b = b_;
    }
  };
  auto load_q = [&](int blk_, f16x8 (&dst)[NQK]) -> QState {
    const int bh_ = blk_ / nqb, qb_ = blk_ % nqb;
    const int b_ = bh_ / p.n_heads, h_ = bh_ % p.n_heads;
    const int qi = qb_ * NW * 32 + wave * 32 + lq;
    bool ok = qi < p.n_q;
    const int qc = ok ? qi : p.n_q - 1;
    const int64_t qrow0 = p.q_batch_rows ? (int64_t)p.q_batch_rows[b_] : (int64_t)b_ * p.n_q;
    int64_t row = qrow0 + qc;                      // row of this query in Q (and, with tok_rows, in O)
    if constexpr (MODE == 2 && ALLKV) {
      if (p.tok_rows) {
        ok = ok && wrow_q >= 0;                    // window padding: nothing to compute, nothing to store
        row = wrow_q >= 0 ? wrow_q : 0;
      }
    }
    const f16* Qrow = (const f16*)p.Q + row * p.ldq + h_ * HD;
#pragma unroll
    for (int s = 0; s < NQKB; ++s) dst[s] = *(const f16x8*)(Qrow + 16 * s + 8 * hh);
    if constexpr (MODE == 2) {
      const f16* R = (const f16*)p.rel_aug + ((int64_t)bh_ * p.n_q + qc) * 32;
      dst[NQKB] = *(const f16x8*)(R + 8 * hh);
      dst[NQKB + 1] = *(const f16x8*)(R + 16 + 8 * hh);
    }
    return QState{ok, row};
  };
  // all K/V rows of a (batch, head) into registers (ALLKV): tok_rows gathers them, padded keys take qkv(0)
  f16x8 ka[KALL], va[KALL];
  (void)ka; (void)va;
  auto load_kv_all = [&](int blk_) {
    const int bh_ = blk_ / nqb;
    const int b_ = bh_ / p.n_heads, h_ = bh_ % p.n_heads;
    const int64_t kvb_ = p.kv_batch_rows ? (int64_t)p.kv_batch_rows[b_] : (int64_t)b_ * p.n_k;
#pragma unroll
    for (int it = 0; it < KALL; ++it) {
      const int ci = tid + it * NT;
      const int key = ci / CH, cc = ci % CH;
      ka[it] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      va[it] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (ci < MAXT * 64 * CH && key < p.n_k) {
        int64_t r = kvb_ + key;
        if (MODE == 2 && p.tok_rows) r = wrow_k[it];
        if (r >= 0) {
          ka[it] = *(const f16x8*)((const f16*)p.K + r * p.ldk + h_ * HD + cc * 8);
          va[it] = *(const f16x8*)((const f16*)p.V + r * p.ldv + h_ * HD + cc * 8);
        } else {                                   // padded key: qkv(0) = the bias rows
          ka[it] = *(const f16x8*)((const f16*)p.pad_k + h_ * HD + cc * 8);
          va[it] = *(const f16x8*)((const f16*)p.pad_v + h_ * HD + cc * 8);
        }
      }
    }
  };
  load_wrows((blk / nqb) / p.n_heads);
  f16x8 qf[NQK];
  QState qs = load_q(blk, qf);
  if constexpr (ALLKV) load_kv_all(blk);
  // V pad columns: zero, except a ones-column (d = HD for the lower half-wave, HD+4 for the upper) so that the
  // PV MFMA also yields l = sum_k P[q,k] in o[NB-1][..] of BOTH halves
  if constexpr (DVP > HD) {
    constexpr int PCH = (DVP - HD) / 8;
    for (int i = tid; i < MAXT * 64 * PCH; i += NT) {
      const int row = (i / PCH) % 64, pc = i % PCH, tl = i / (64 * PCH);
      f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (LSUM_MFMA && pc == 0) { z[0] = (f16)1; z[4] = (f16)1; }
      *(f16x8*)(sV + tl * TILEB + row * VROW + (CH + pc) * 16) = z;
    }
  }
  // per-thread byte offsets of its K/V chunks inside a 64-key tile (32-bit; the tile base is wave-uniform)
  int koff_g[KIT], voff_g[KIT], loff_k[KIT], loff_v[KIT];
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int ci = tid + it * NT;
    const int row = (ci / CH) & 63, cc = ci % CH;
    koff_g[it] = (row * (int)p.ldk + cc * 8) * 2;
    voff_g[it] = (row * (int)p.ldv + cc * 8) * 2;
    loff_k[it] = row * KROW + cc * 16;
    loff_v[it] = row * VROW + cc * 16;
  }
  if constexpr (MODE == 2 && ALLKV) {   // one-hot (kh, kw) columns of K': the same for every block, written once
    for (int i = tid; i < MAXT * 256; i += NT) {
      const int key = i >> 2, c4 = i & 3;
      const int kh = key / p.grid_w, kw = key - kh * p.grid_w + p.grid_w;
      f16x8 e;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = c4 * 8 + j;
        e[j] = (key < p.n_k && (col == kh || col == kw)) ? (f16)1 : (f16)0;
      }
      *(f16x8*)(sK + (key >> 6) * TILEB + (key & 63) * KROW + (CH + c4) * 16) = e;
    }
  }
  const float c = p.scale * 1.44269504088896340736f;
  const int ntiles = (p.n_k + 63) / 64;
  // the last tile of a 14x14 window holds 4 of its 64 key slots: only its first 32-key half is computed
  const bool half_tail = ALLKV && (p.n_k - (ntiles - 1) * 64) <= 32;
  // per-lane LDS read offsets
  const int koff0 = lq * KROW + hh * 16;
  const int koff1 = (32 + lq) * KROW + hh * 16;
  const int voff = (4 * hh + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  // register index / half that holds row d = HD (+4 for hh = 1) of O^T
  constexpr int LI = HD / 32, LR = (((HD % 32) / 8) * 4);

  for (; blk < blk_end; ++blk) {
  const int bh = blk / nqb, qb = blk % nqb;
  const int b = bh / p.n_heads, h = bh % p.n_heads;
  const int q_idx = qb * NW * 32 + wave * 32 + lq;
  const int q_c = q_idx < p.n_q ? q_idx : p.n_q - 1;
  const bool q_ok = qs.ok;
  const int64_t q_row = qs.row;
  const int64_t kvb = p.kv_batch_rows ? (int64_t)p.kv_batch_rows[b] : (int64_t)b * p.n_k;
  const char* Kb = (const char*)((const f16*)p.K + kvb * p.ldk + h * HD);
  const char* Vb = (const char*)((const f16*)p.V + kvb * p.ldv + h * HD);
  float bias3[2][16];   // mode 3 only: the whole (bias + mask) row of this query (single tile)
  const float* RH = nullptr;
  if constexpr (MODE == 3) {
    const float* bp = p.dense_bias + ((int64_t)h * p.n_q + q_c) * 64;
    const float* mp = p.dense_mask ? p.dense_mask + ((int64_t)(b % p.n_mask) * p.n_q + q_c) * 64 : nullptr;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = *(const f32x4*)(bp + sub * 32 + 8 * g + 4 * hh);
        if (mp) v += *(const f32x4*)(mp + sub * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias3[sub][4 * g + r] = v[r];
      }
  }
  f32x16 rw0, rw1;      // mode 1: rel_w[q, kw] of this lane's query for the 2 x 16 key slots of a tile; they are
                        // the C operand of the first QK^T MFMA of every tile (C != D: no copies, no LDS, no VALU)
  if constexpr (MODE == 1) {
    const float* RW = p.rel_w + ((int64_t)bh * p.n_q + q_c) * 64;
    RH = p.rel_h + ((int64_t)bh * p.n_q + q_c) * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = *(const f32x4*)(RW + 8 * g + 4 * hh), c2 = *(const f32x4*)(RW + 32 + 8 * g + 4 * hh);
#pragma unroll
      for (int r = 0; r < 4; ++r) { rw0[4 * g + r] = a[r]; rw1[4 * g + r] = c2[r]; }
    }
  }

  f16x8 kreg[KIT], vreg[KIT];
  auto load_tile = [&](int t) {
    const char* kt = Kb + (int64_t)t * 64 * p.ldk * 2;
    const char* vt = Vb + (int64_t)t * 64 * p.ldv * 2;
    const bool full = (t + 1) * 64 <= p.n_k;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int ci = tid + it * NT;
      const bool ok = ci < 64 * CH && (full || t * 64 + ci / CH < p.n_k);
      if (ok) {
        kreg[it] = *(const f16x8*)(kt + koff_g[it]);
        vreg[it] = *(const f16x8*)(vt + voff_g[it]);
      } else {
        kreg[it] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        vreg[it] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  auto store_tile = [&](int t) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      if (tid + it * NT < 64 * CH) {
        *(f16x8*)(sK + (t & 1) * TILEB + loff_k[it]) = kreg[it];
        *(f16x8*)(sV + (t & 1) * TILEB + loff_v[it]) = vreg[it];
      }
    }
    if constexpr (MODE == 2 && !ALLKV) {  // one-hot (kh, kw) columns of K'
      for (int i = tid; i < 256; i += NT) {
        const int row = i >> 2, c4 = i & 3;
        const int key = t * 64 + row;
        const int kh = key / p.grid_w, kw = key - kh * p.grid_w + p.grid_w;
        f16x8 e;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = c4 * 8 + j;
          e[j] = (key < p.n_k && (col == kh || col == kw)) ? (f16)1 : (f16)0;
        }
        *(f16x8*)(sK + (t & 1) * TILEB + row * KROW + (CH + c4) * 16) = e;
      }
    }
  };

  f32x16 o[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = NEG, l_run = 0.f;

  float rh_next = 0.f;
  if constexpr (MODE == 1) rh_next = RH[0];
  if constexpr (ALLKV) {
    // the rows of this block are in ka/va (issued one block ago): hand them to LDS between two barriers (the first
    // one: every wave has finished reading the previous block's tiles), then immediately issue the loads of the
    // next block so that they fly during the tile loop, which runs without any synchronisation
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KALL; ++it) {
      const int ci = tid + it * NT;
      const int key = ci / CH, cc = ci % CH;
      if (ci < MAXT * 64 * CH) {
        *(f16x8*)(sK + (key >> 6) * TILEB + (key & 63) * KROW + cc * 16) = ka[it];
        *(f16x8*)(sV + (key >> 6) * TILEB + (key & 63) * VROW + cc * 16) = va[it];
      }
    }
    __syncthreads();
    if (blk + 1 < blk_end) {
      load_wrows(((blk + 1) / nqb) / p.n_heads);   // a (rare) window change costs one exposed round trip
      load_kv_all(blk + 1);
    }
  } else {
    // streamed K/V (everything but the windows): tiles alternate between two LDS buffers, so ONE barrier per tile
    // is enough - it publishes tile t+1 and retires the reads of tile t - and the LDS store of tile t+1 and the
    // global loads of tile t+2 follow the MFMAs of tile t (issuing them first was measured slower: 1232 vs 1188 us)
    load_tile(0);
    store_tile(0);
    if (ntiles > 1) load_tile(1);
    __syncthreads();
  }
  // (a wave whose 32 queries are all window padding - bottom-row windows - has nothing to compute)
  const int nt_wave = (ALLKV && !__any(q_ok)) ? 0 : ntiles;
  for (int t = 0; t < nt_wave; ++t) {
    const char* tK = sK + (ALLKV ? t : (t & 1)) * TILEB;
    const char* tV = sV + (ALLKV ? t : (t & 1)) * TILEB;
    const float rh = rh_next;               // rel_h[q, kh = t]: a per-tile constant of this lane
    if constexpr (MODE == 1) {
      if (t + 1 < ntiles) rh_next = RH[t + 1];   // prefetch: never a dependent load at the top of a tile
    }

    f32x16 s0, s1;
    // K fragments of the whole tile up front where the registers allow it (everything but the persistent window
    // kernel): the S-phase MFMA chain then never waits on an LDS read issued just before it
    constexpr bool KPRE = !ALLKV;
    f16x8 kfa[KPRE ? NQK : 1], kfb[KPRE ? NQK : 1];
    if constexpr (KPRE) {
#pragma unroll
      for (int s = 0; s < NQK; ++s) {
        kfa[s] = *(const f16x8*)(tK + koff0 + s * 32);
        kfb[s] = *(const f16x8*)(tK + koff1 + s * 32);
      }
      __builtin_amdgcn_sched_barrier(0);     // (the scheduler would sink the reads back next to their MFMAs)
    }
    if constexpr (MODE == 1) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfa[0], qf[0], rw0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfb[0], qf[0], rw1, 0, 0, 0);
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = bias3[0][r];
        s1[r] = bias3[1][r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
    }
    const bool half = half_tail && t == ntiles - 1;     // wave-uniform: keys 32..63 of this tile do not exist
    if (half) {
#pragma unroll
      for (int s = 0; s < NQK; ++s) {
        const f16x8 k0 = *(const f16x8*)(tK + koff0 + s * 32);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = (MODE == 1 ? 1 : 0); s < NQK; ++s) {
        const f16x8 k0 = KPRE ? kfa[s] : *(const f16x8*)(tK + koff0 + s * 32);
        const f16x8 k1 = KPRE ? kfb[s] : *(const f16x8*)(tK + koff1 + s * 32);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
      }
    }
    if ((t + 1) * 64 > p.n_k) {  // ragged last tile: mask keys >= n_k (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 64 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= p.n_k) s0[r] = NEG;
        if (key + 32 >= p.n_k) s1[r] = NEG;
      }
    }
    // row max (this half's 32 keys, then the other half); rel_h is added AFTER the max (it is constant)
    float mx = max3(s0[0], s0[1], s1[0]);
    mx = max3(mx, s1[1], s0[2]);
#pragma unroll
    for (int r = 3; r < 16; r += 2) mx = max3(mx, s0[r], s0[r + 1 < 16 ? r + 1 : r]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = max3(mx, s1[r], s1[r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, (mx + rh) * c);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      m_run = m_new;
    }
    const float koff = rh * c - m_run;      // p = 2^((s + rh) * c - m)
    f16x8 pf[4];
    auto pv = [&](int ks, int i) {
      const char* base = tV + voff + (16 * ks) * VROW + i * 64;
      const f16x4 a0 = tr_read(base);
      const f16x4 a1 = tr_read(base + 8 * VROW);
      const f16x8 vf = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      o[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], o[i], 0, 0, 0);
    };
    if constexpr (LSUM_MFMA && !ALLKV) {
      // keys 0..31 first, and their P.V MFMAs go out while the exps of keys 32..63 are still being computed: an
      // MFMA 32x32x16 blocks vector issue for only 8 of its 32 cycles (MI355X_MICROARCH.md, issue-cost table), so
      // ~3 exp+fma per MFMA gap ride along for free instead of forming a VALU-only phase
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, koff));
      pf[0] = cvt8(s0, 0);
      pf[1] = cvt8(s0, 8);
      if (half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < NB; ++i) pv(ks, i);
      } else {
        constexpr int NG = 2 * NB;               // MFMA gaps available for the 16 exps of the second half
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          pv(g / NB, g % NB);
#pragma unroll
          for (int r = (16 * g) / NG; r < (16 * (g + 1)) / NG; ++r) s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, koff));
        }
        pf[2] = cvt8(s1, 0);
        pf[3] = cvt8(s1, 8);
#pragma unroll
        for (int ks = 2; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < NB; ++i) pv(ks, i);
      }
    } else {
      // (the persistent window kernel has no registers to spare for the interleave; and without the ones-column
      // the row sum is accumulated on the VALU)
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, koff));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, koff));
        ps += s0[r] + s1[r];
      }
      if constexpr (!LSUM_MFMA) l_run += ps;
      pf[0] = cvt8(s0, 0);
      pf[1] = cvt8(s0, 8);
      pf[2] = cvt8(s1, 0);
      pf[3] = cvt8(s1, 8);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= 2 && half) break;                        // P of keys 32..63 is exactly 0 there
#pragma unroll
        for (int i = 0; i < NB; ++i) pv(ks, i);
      }
    }
    if constexpr (!ALLKV) {
      if (t + 1 < ntiles) {
        store_tile(t + 1);                       // into the buffer tile t-1 was read from
        if (t + 2 < ntiles) load_tile(t + 2);
      }
      __syncthreads();                           // publishes tile t+1, retires the reads of tile t
    }
  }

  float ltot;
  if constexpr (LSUM_MFMA) {
    ltot = o[LI][LR];                        // row d = HD (hh = 0) / HD + 4 (hh = 1) of O^T: sum_k P
  } else {
    ltot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / ltot;
  if (q_ok) {
    // O is dense per batch entry unless tok_rows scatters it back to token order
    const int64_t o_row = (MODE == 2 && ALLKV && p.tok_rows) ? q_row : (int64_t)b * p.n_q + q_idx;
    f16* Orow = (f16*)p.O + o_row * p.ldo + h * HD;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * i + 8 * g + 4 * hh;
        if (d0 < HD) {
          const f16x4 v = {(f16)(o[i][4 * g] * inv), (f16)(o[i][4 * g + 1] * inv),
                           (f16)(o[i][4 * g + 2] * inv), (f16)(o[i][4 * g + 3] * inv)};
          *(f16x4*)(Orow + d0) = v;
        }
      }
  }
  if constexpr (ALLKV) {
    if (blk + 1 < blk_end) qs = load_q(blk + 1, qf);   // flies during the barriers / LDS hand-off of the next block
  }
  }   // block loop
}

// Decomposed relative-position terms, pre-divided by the softmax scale so that they can sit
// next to q.k inside the accumulator:  rel_x[q, j] = (q . R_x[q_x - j + S - 1]) / scale.
// SA/modeling/image_encoder.py:292-361.
//
// MFMA form: D[r][q] = sum_d Tab[r][d] * Q[q][d] for the stacked tables
// Tab = [rel_pos_h (2S-1 rows, padded to RT*32) ; rel_pos_w (same)], then a scattered store
// out[q][j] = D[q_x - j + S - 1][q].  One workgroup per (batch*head, block of QB*32 queries); the table is
// converted to f16 once into LDS (A operand via ds_read_b128), Q^T fragments come straight from HBM.
template <int HD, int S, bool AUG, bool F16T = false>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const f16* __restrict__ Q, int64_t ldq,
                                                          const float* __restrict__ Rh,
                                                          const float* __restrict__ Rw, int n_heads,
                                                          float inv_scale, const int32_t* __restrict__ tok_rows,
                                                          float* __restrict__ out_h, float* __restrict__ out_w,
                                                          f16* __restrict__ out_aug) {
  constexpr int NQ = S * S;
  constexpr int RT = (2 * S - 1 + 31) / 32;       // 32-row tiles per table: 1 (S=14) or 4 (S=64)
  constexpr int ROWS = 2 * RT * 32;
  constexpr int TROW = (HD / 8 + 1) * 16;         // 176 B: odd number of 16-B chunks -> conflict-free b128
  constexpr int NQG = (NQ + 31) / 32;             // 32-query groups per (batch, head)
  constexpr int GPB = AUG ? NQG : 8;              // groups per workgroup (all 7 for a window; 8 of 128 global)
  // per-wave output tile: [32 queries][AUG ? 32 f16 : 64 f32] (+16 B row pad), so that the scattered D elements
  // land in LDS and HBM only ever sees whole-row 16-B stores
  constexpr int OROW = (AUG ? 64 : 256) + 16;
  __shared__ __attribute__((aligned(16))) char tab[ROWS * TROW + 4 * 32 * OROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  char* ot = tab + ROWS * TROW + wave * 32 * OROW;
  const int nblk = (NQG + GPB - 1) / GPB;
  const int bh = blockIdx.x / nblk, gb = blockIdx.x % nblk;
  const int b = bh / n_heads, h = bh % n_heads;
  for (int i = tid; i < ROWS * (HD / 8); i += 256) {
    const int r = i / (HD / 8), c8 = i % (HD / 8);
    const int tr = r % (RT * 32);
    const float* src = (r < RT * 32 ? Rh : Rw) + (int64_t)tr * HD + c8 * 8;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tr < 2 * S - 1) {
      const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
      v = (f16x8){(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
    }
    *(f16x8*)(tab + r * TROW + c8 * 16) = v;
  }
  __syncthreads();
  for (int g = gb * GPB + wave; g < min(NQG, (gb + 1) * GPB); g += 4) {
    const int q = g * 32 + lq;
    bool q_ok = q < NQ;
    const int qc = q_ok ? q : NQ - 1;
    int64_t qrow = (int64_t)b * NQ + qc;
    if (AUG && tok_rows) {
      const int r = tok_rows[qrow];
      q_ok = q_ok && r >= 0;
      qrow = r >= 0 ? r : 0;
    }
    const f16* qp = Q + qrow * ldq + h * HD;
    f16x8 qf[HD / 16];
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) qf[s] = *(const f16x8*)(qp + 16 * s + 8 * hh);
    const int qh = qc / S, qw = qc % S;
    if (AUG && hh == 0) *(f16x4*)(ot + lq * OROW + 56) = (f16x4){0, 0, 0, 0};   // cols 28..31
#pragma unroll
    for (int tt = 0; tt < 2 * RT; ++tt) {
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
      for (int s = 0; s < HD / 16; ++s) {
        const f16x8 a = *(const f16x8*)(tab + (tt * 32 + lq) * TROW + (2 * s + hh) * 16);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], d, 0, 0, 0);
      }
      const bool is_w = tt >= RT;
      const int base = (is_w ? qw : qh) + S - 1 - (tt % RT) * 32;   // j = base - (row within this tile)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = base - ((r & 3) + 8 * (r >> 2) + 4 * hh);
        if (j >= 0 && j < S) {
          const float v = d[r] * inv_scale;
          if (AUG) {
            *(f16*)(ot + lq * OROW + ((is_w ? S : 0) + j) * 2) = (f16)v;
          } else {
            *(float*)(ot + lq * OROW + j * 4) = v;
          }
        }
      }
      if (!AUG && (tt % RT) == RT - 1) {
        // one table done for these 32 queries: rows of 64 f32 -> lane (q, hh) stores its 128-B half row
        if (F16T) {
          // f16 tables (out_h / out_w carry f16 pointers): half the bytes written here and read by the attention kernel
          f16* dst = (f16*)(is_w ? out_w : out_h) + ((int64_t)bh * NQ + q) * S + hh * 32;
          if (q_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 a = *(const f32x4*)(ot + lq * OROW + hh * 128 + i * 32), c = *(const f32x4*)(ot + lq * OROW + hh * 128 + i * 32 + 16);
              *(f16x8*)(dst + 8 * i) = (f16x8){(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
            }
          }
        } else {
          float* dst = (is_w ? out_w : out_h) + ((int64_t)bh * NQ + q) * S + hh * 32;
          if (q_ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(f32x4*)(dst + 4 * i) = *(const f32x4*)(ot + lq * OROW + hh * 128 + i * 16);
          }
        }
      }
    }
    if (AUG && q_ok) {
      f16* dst = out_aug + ((int64_t)bh * NQ + q) * 32 + hh * 16;
      *(f16x8*)dst = *(const f16x8*)(ot + lq * OROW + hh * 32);
      *(f16x8*)(dst + 8) = *(const f16x8*)(ot + lq * OROW + hh * 32 + 16);
    }
  }
}

}  // namespace

extern "C" int ink_flash_attn(const InkAttn* pp, void* stream) {
  INK_CHECK_ARG(pp != nullptr);
  const InkAttn& p = *pp;
  INK_CHECK_ARG(p.Q && p.K && p.V && p.O);
  INK_CHECK_ARG(p.n_batch > 0 && p.n_heads > 0 && p.n_q > 0 && p.n_k > 0);
  INK_CHECK_ARG(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0);
  INK_CHECK_ARG((((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V) & 15) == 0);
  INK_CHECK_ARG(((uintptr_t)p.O & 7) == 0);
  INK_CHECK_ARG(!p.tok_rows || p.bias_mode == 2);
  static const int n_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  constexpr bool persist = true;      // windows: persistent walk over (window, head) blocks
  hipStream_t s = (hipStream_t)stream;
  const int bhn = p.n_batch * p.n_heads;
#define INK_FA_X(HD, MODE, NW, ALL)                                                                      \
  {                                                                                                        \
    constexpr int nqk_ = HD / 16 + (MODE == 2 ? 2 : 0);                                                    \
    constexpr int lds_ = (ALL ? 4 : 2) * (64 * (((nqk_ * 2) | 1) * 16) + 64 * (((HD + 31) / 32) * 64)) +   \
                         0;                                                                            \
    static bool attr_ = ((void)hipFuncSetAttribute((const void*)flash_attn_kernel<HD, MODE, NW, ALL>,      \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_), true); \
    (void)attr_;                                                                                           \
    const int nqb = (p.n_q + NW * 32 - 1) / (NW * 32);                                                     \
    const int grid_ = (ALL && persist) ? (bhn * nqb < n_cus ? bhn * nqb : n_cus) : bhn * nqb;                \
    hipLaunchKernelGGL((flash_attn_kernel<HD, MODE, NW, ALL>), dim3(grid_), dim3(NW * 64), lds_, s, p);      \
  }
#define INK_FA(HD, MODE, NW) INK_FA_X(HD, MODE, NW, false)
  if (p.head_dim == 80 && p.bias_mode == 1) {
    INK_CHECK_ARG(p.rel_h && p.rel_w && p.grid_w == 64 && p.n_k % 64 == 0);
    // SAM's own shape (64 x 64 tokens): the one-wave-per-SIMD kernel of attention_glob.hip
    if (p.n_q % 256 == 0 && p.n_k % 128 == 0 && p.n_k >= 256 && p.n_k <= 4096) return ink_glob4_attn_launch(p, s);
    INK_CHECK_ARG(!p.rel_f16);          // f16 tables: the one-wave-per-SIMD kernel only
    INK_FA(80, 1, 8);
  } else if (p.head_dim == 80 && p.bias_mode == 2) {
    INK_CHECK_ARG(p.rel_aug && p.grid_w > 0 && p.grid_w <= 16 && p.n_k <= p.grid_w * p.grid_w && p.n_k <= 256);
    INK_CHECK_ARG(!p.tok_rows || (p.n_q == p.n_k && p.pad_k && p.pad_v &&
                                  (((uintptr_t)p.pad_k | (uintptr_t)p.pad_v) & 15) == 0));
    // SAM's own window size (14 x 14 = 196 keys): the one-wave-per-SIMD kernel of attention_win.hip.  Its output
    // stores go through a 2 GiB buffer descriptor (invalid rows are dropped by the bounds check).
#ifndef INK_EXP_NO_WIN4        // (experiment: the round-1 window kernel instead, tools/race_variants.sh)
    if (p.n_k >= 193 && p.n_k <= 208 && p.n_q <= 256 &&
        (p.tok_rows || (int64_t)p.n_batch * p.n_q * p.ldo * 2 < 0x80000000LL))
      return ink_win4_attn_launch(p, n_cus, s);
#endif
    INK_FA_X(80, 2, 7, true);
  } else if (p.head_dim == 80 && p.bias_mode == 0) {
    INK_FA(80, 0, 4);
  } else if (p.head_dim == 64 && p.bias_mode == 0) {
    INK_FA(64, 0, 4);
  } else if (p.head_dim == 32 && p.bias_mode == 0) {
    if (p.n_q <= 32) { INK_FA(32, 0, 1) } else { INK_FA(32, 0, 4) }
  } else if (p.head_dim == 32 && p.bias_mode == 3) {
    INK_CHECK_ARG(p.dense_bias && p.n_k <= 64 && p.n_q <= 64 && (!p.dense_mask || p.n_mask > 0));
    INK_FA(32, 3, 2);
  } else if (p.head_dim == 16 && p.bias_mode == 0) {
    if (p.n_q <= 32) { INK_FA(16, 0, 1) } else { INK_FA(16, 0, 4) }
  } else {
    return INK_ERR_ARG;
  }
#undef INK_FA
#undef INK_FA_X
  return ink_launch_status();
}

extern "C" int ink_relpos_bias64_f16(const void* Q, int64_t ldq, const float* rel_pos_h, const float* rel_pos_w,
                                     int32_t n_batch, int32_t n_heads, int32_t head_dim, float scale, void* out_h_f16,
                                     void* out_w_f16, void* stream) {
  INK_CHECK_ARG(Q && rel_pos_h && rel_pos_w && head_dim == 80 && ldq % 8 == 0 && out_h_f16 && out_w_f16);
  INK_CHECK_ARG(n_batch > 0 && n_heads > 0 && scale > 0.f && ((((uintptr_t)out_h_f16 | (uintptr_t)out_w_f16) & 15) == 0));
  hipLaunchKernelGGL((relpos_mfma_kernel<80, 64, false, true>), dim3(n_batch * n_heads * 16), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)Q, ldq, rel_pos_h, rel_pos_w, n_heads, 1.0f / scale, (const int32_t*)nullptr,
                     (float*)out_h_f16, (float*)out_w_f16, (f16*)nullptr);
  return ink_launch_status();
}

extern "C" int ink_relpos_bias(const void* Q, int64_t ldq, const float* rel_pos_h,
                               const float* rel_pos_w, int32_t S, int32_t n_batch,
                               int32_t n_heads, int32_t head_dim, float scale, const int32_t* tok_rows,
                               float* out_h, float* out_w, void* out_aug_f16, void* stream) {
  INK_CHECK_ARG(Q && rel_pos_h && rel_pos_w && head_dim == 80 && ldq % 8 == 0);
  INK_CHECK_ARG(n_batch > 0 && n_heads > 0 && S > 0 && scale > 0.f);
  if (out_aug_f16) {
    INK_CHECK_ARG(S <= 16);
  } else {
    INK_CHECK_ARG(S == 64 && out_h && out_w && !tok_rows);
  }
  const f16* q = (const f16*)Q;
  hipStream_t st = (hipStream_t)stream;
  const int nbh = n_batch * n_heads;
  if (out_aug_f16) {
    INK_CHECK_ARG(S == 14);
    hipLaunchKernelGGL((relpos_mfma_kernel<80, 14, true>), dim3(nbh), dim3(256), 0, st, q, ldq, rel_pos_h,
                       rel_pos_w, n_heads, 1.0f / scale, tok_rows, out_h, out_w, (f16*)out_aug_f16);
  } else {
    hipLaunchKernelGGL((relpos_mfma_kernel<80, 64, false>), dim3(nbh * 16), dim3(256), 0, st, q, ldq, rel_pos_h,
                       rel_pos_w, n_heads, 1.0f / scale, (const int32_t*)nullptr, out_h, out_w, (f16*)nullptr);
  }
  return ink_launch_status();
}

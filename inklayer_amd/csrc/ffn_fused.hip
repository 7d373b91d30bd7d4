// Fused feed-forward block of GroundingDINO's deformable encoder layer for gfx950:
//     out = LayerNorm(src + linear2(relu(linear1(src))))            d_model 256, d_ffn 2048
// (DeformableTransformerEncoderLayer.forward_ffn + norm2, GD/models/GroundingDINO/transformer.py:780-799).
// As two GEMMs the [rows, 2048] f16 hidden tensor is written and re-read (2 x 435 MB per layer at B = 8) and at
// K = 256 / N = 256 the GEMM tiles are all prologue and epilogue (192 us each + 33 us LayerNorm).  Here the hidden
// activations never leave the registers.
//
// One workgroup = 128 token rows = FOUR waves, one per SIMD with the whole 512-register file; a wave owns 32 rows.
//   * d_ffn is walked in chunks of 64 hidden units.  Phase A: H^T[j, m] = W1c X^T (mfma_f32_32x32x16_f16, the wave's X
//     fragments - 32 rows x 256 - stay in 64 registers for the whole tile), accumulators initialised with b1.
//     relu + convert in registers: in the swapped form the lane (m = lane & 31, hh) holds 8 hidden units of ITS row per
//     16-wide k-step, which IS the B-operand layout of phase B up to a fixed permutation of the hidden index inside
//     16-blocks - a contraction index, so the permutation is applied to W2's columns when the weights are packed and
//     costs nothing here (no shuffles, no LDS round trip).  Phase B: Y^T[n, m] += W2c H^T into 8 accumulator tiles
//     (all 256 output columns of the wave's 32 rows: 128 registers), initialised with src + b2.
//   * The weights are PRE-PACKED (ink_ffn256_pack) into the exact LDS image: per chunk 64 blocks of 1 KiB, each one
//     MFMA A-operand (32 rows x 16 k, lane-linear).  Staging is a linear LDS-DMA copy (fully coalesced) and every
//     fragment read is a lane-linear ds_read_b128 (conflict-free by construction).  Two 64-KiB buffers: chunk c + 1
//     streams in while chunk c is computed; one barrier per chunk.  Only DMA is in flight inside the loop, so the one
//     vmcnt(0) per chunk is exact.
//   * Epilogue: a lane holds half of its row (128 values), the other half sits in lane ^ 32: LayerNorm statistics are
//     in-lane sums + one cross-half exchange; rows leave through LDS (the weight buffers are free by then) as whole
//     1-KiB rows.
#include <type_traits>
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 256, HC = 64, BM = 128;
constexpr int BLK = 1024;                       // one A-operand block: 32 rows x 16 k of f16, lane-linear
constexpr int KA = C / 16 + 1;                  // k-steps of phase A: 16 of x . W1 + one that carries b1 (x' = [x | 1 1 0 ..])
constexpr int W1_BLKS = (HC / 32) * KA;         // 34: (jt, s)
constexpr int W2_BLKS = (C / 32) * (HC / 16);   // 32: (nt, s)
constexpr int NBLK = W1_BLKS + W2_BLKS;         // 66 blocks = MFMAs = fragment reads per chunk and wave
constexpr int CHUNK = NBLK * BLK;               // 66 KiB per chunk of 64 hidden units
constexpr int MAX_HID = 2048;
constexpr int OROW = C * 4 + 16;                // epilogue staging row (f32) + pad
constexpr int PAR_BYTES = 5 * C * 4;             // LayerNorm / bias vectors next to the weight buffers: pre_g, pre_be, b2, ln_g, ln_b
constexpr int LDS_BYTES = 2 * CHUNK + PAR_BYTES;
#ifndef INK_FFN_DEPTH
#define INK_FFN_DEPTH 6
#endif
constexpr int DEPTH = INK_FFN_DEPTH;            // fragment reads in flight ahead of the MFMA that consumes them
constexpr int NPRE = 2;                         // chunks of the optional preceding [256 -> 256] projection (128 blocks + padding)
constexpr int NSLOT = (NBLK + 3) / 4;           // LDS-DMA pieces per wave and chunk (17; piece index clamped to NBLK - 1)
static_assert(4 * 32 * OROW <= 2 * CHUNK, "epilogue staging fits the (then free) weight buffers");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) char* lds_char_ptr;
template <int I> using ic = std::integral_constant<int, I>;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<I + 1, N>(f);
  }
}
// Fragment reads are volatile asm so that they KEEP their distance to the MFMA that consumes them (left to itself hipcc
// sinks every ds_read next to its use and waits lgkmcnt(0) in front of each MFMA: 6000 instead of 2100 cycles per chunk);
// the counted wait is tied to the fragment it releases, which orders the MFMA behind it.  Nothing else in the loop uses
// lgkmcnt, and LDS returns in order.
template <int OFF>
__device__ __forceinline__ f16x8 lds_frag(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_frag(f16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}
// four parameter vectors (2 x gamma, 2 x beta of one 8-column group pair) through the same pinned path: as plain loads
// hipcc puts all 64 of a LayerNorm in flight at once (256 registers) and spills - and a spilled fragment register of the
// asm reads above would be stored before its data has arrived
template <int OFF>
__device__ __forceinline__ f32x4 lds_f4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wait_f4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __forceinline__ f32x16 mfma32(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// position p = 8 hh + e of a 16-wide k-step of phase B  <->  hidden unit (within the 16-block) the lane holds there
__host__ __device__ constexpr int hid_perm(int p) { return (p & 7) < 4 ? 4 * (p >> 3) + (p & 7) : 8 + 4 * (p >> 3) + (p & 7) - 4; }

__global__ __launch_bounds__(256) void ffn256_pack_kernel(const f16* __restrict__ W1, const float* __restrict__ b1,
                                                          const f16* __restrict__ W2, int HID, const f16* __restrict__ Wpre,
                                                          f16* __restrict__ blob) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 16-B piece of the blob
  const int npre = Wpre ? NPRE : 0;
  const int64_t total = (int64_t)(npre + HID / HC) * NBLK * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int q = (int)((idx >> 6) % NBLK);
  const int cc = (int)((idx >> 6) / NBLK);
  const int l = lane & 31, hh = lane >> 5;
  f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cc < npre) {
    // the preceding [256 -> 256] projection: pre-chunk cc holds k-steps 8 cc .. 8 cc + 7 of all eight column tiles, block
    // 8 nt + j = (nt, s = 8 cc + j) - one body of code serves both pre-chunks with compile-time register indices
    if (q < 64) {
      const int nt = q / 8, s = 8 * cc + q % 8;
      v = *(const f16x8*)(Wpre + (int64_t)(32 * nt + l) * C + 16 * s + 8 * hh);
    }
  } else {
    const int c = cc - npre;
    if (q < W1_BLKS) {
      const int jt = q / KA, s = q % KA;
      const int j = c * HC + 32 * jt + l;
      if (s < C / 16) {
        const f16* src = W1 + (int64_t)j * C + 16 * s;
        if (Wpre) {                    // x comes out of accumulators: its channels sit in hid_perm order inside 16-blocks
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[hid_perm(8 * hh + e)];
        } else {
          v = *(const f16x8*)(src + 8 * hh);
        }
      } else if (hh == 0) {              // the bias k-step: b1 as hi + lo against x' = (1, 1, 0, ...)
        const f16 hi = (f16)b1[j];
        v[0] = hi;
        v[1] = (f16)(b1[j] - (float)hi);
      }
    } else {
      const int nt = (q - W1_BLKS) / (HC / 16), s = (q - W1_BLKS) % (HC / 16);
      const f16* src = W2 + (int64_t)(32 * nt + l) * HID + c * HC + 16 * s;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[hid_perm(8 * hh + e)];
    }
  }
  *(f16x8*)(blob + idx * 8) = v;
}

// PRE: the kernel starts one step earlier - x is the INPUT of a preceding projection (here: the deformable attention's
// output), src <- LayerNorm1(src + out_proj(x)) is formed in the accumulators first (transformer.py:790-793: the
// attention's output projection, residual and norm1) and its f16 rounding becomes the feed-forward block's operand
// without leaving the registers: the accumulator layout is the B-operand layout up to hid_perm, which the packed W1
// absorbs.  Saves a GEMM launch, a LayerNorm pass, and one write + two reads of the token matrix per layer.
template <bool PRE>
__global__ __launch_bounds__(256) void ffn256_fused_kernel(const f16* __restrict__ X, int64_t ldx,
                                                           const float* __restrict__ res, const f16* __restrict__ blob,
                                                           const float* __restrict__ b2, const float* __restrict__ ln_g,
                                                           const float* __restrict__ ln_b, float eps, int M, int HID,
                                                           const float* __restrict__ pre_b, const float* __restrict__ pre_g,
                                                           const float* __restrict__ pre_be, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the LDS-DMA's M0 and base address are SALU work
  const int l = lane & 31, hh = lane >> 5;
  const int ntiles = (M + BM - 1) / BM;
  const int m0 = xcd_remap(blockIdx.x, ntiles) * BM + wave * 32;
  const int row = min(m0 + l, M - 1);
  const int nchunk = (PRE ? NPRE : 0) + HID / HC;

  // X' fragments of the wave's 32 rows (B operand): 16 k-steps of x + the bias step (1, 1, 0, ...)
  f16x8 xf[KA];
  const f16* xp = X + (int64_t)row * ldx + 8 * hh;
#pragma unroll
  for (int s = 0; s < (PRE ? 8 : C / 16); ++s) xf[s] = *(const f16x8*)(xp + 16 * s);   // PRE: k-steps 8..15 follow after pre-chunk 0
  xf[C / 16] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  if (hh == 0) xf[C / 16][0] = xf[C / 16][1] = (f16)1.0f;
  // Y^T accumulators = src + bias : lane (m = l, hh), tile nt, reg r <-> column 32 nt + 8 (r >> 2) + 4 hh + (r & 3)
  f32x16 y[C / 32];
  const float* rp = res + (int64_t)row * C + 4 * hh;
  const float* bias0 = PRE ? pre_b : b2;
#pragma unroll
  for (int nt = 0; nt < C / 32; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 r4 = *(const f32x4*)(rp + 32 * nt + 8 * g);
      const f32x4 c4 = *(const f32x4*)(bias0 + 32 * nt + 8 * g + 4 * hh);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[nt][4 * g + e] = r4[e] + c4[e];
    }
  {
    float* sP = (float*)(smem + 2 * CHUNK);
    const int i = tid;                                   // 256 threads, 256 columns
    sP[i] = PRE ? pre_g[i] : 0.f;
    sP[C + i] = PRE ? pre_be[i] : 0.f;
    sP[2 * C + i] = b2[i];
    sP[3 * C + i] = ln_g[i];
    sP[4 * C + i] = ln_b[i];
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on only the LDS-DMA is in flight

  // LDS-DMA piece k of this wave for chunk c: block min(4 k + wave, NBLK - 1) (the two waves without a 17th piece
  // re-copy the last block: same bytes).  One piece per four MFMAs of the previous chunk: a piece costs ~60 cycles of
  // issue, which one wave per SIMD cannot hide behind more than the MFMAs already in flight.
  auto stage = [&](int buf, int c, int k) {
    const int q = min(4 * k + wave, NBLK - 1);
    __builtin_amdgcn_global_load_lds((gptr_t)((const char*)blob + (int64_t)c * CHUNK + q * BLK + lane * 16),
                                     (lptr_t)(smem + buf * CHUNK + q * BLK), 16, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) stage(0, 0, k);

  // LayerNorm over the 256 columns of the lane's row, in the accumulators: 128 values here, 128 in lane ^ 32; gamma / beta
  // from the LDS strip (which = 0: pre_g / pre_be, 3: ln_g / ln_b), two 4-column groups at a time
  const uint32_t par0 = (uint32_t)(uintptr_t)(lds_char_ptr)smem + 2 * CHUNK + 16 * hh;
  auto layernorm = [&](int which) {
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < C / 32; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += y[nt][r];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int nt = 0; nt < C / 32; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = y[nt][r] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + eps);
    const uint32_t pg = par0 + which * (C * 4);       // gamma; beta follows C floats later
    static_for<0, C / 32 * 2>([&](auto ii) {
      constexpr int nt = decltype(ii)::value / 2, g = 2 * (decltype(ii)::value % 2);
      constexpr int o0 = (32 * nt + 8 * g) * 4;
      f32x4 g0 = lds_f4<o0>(pg), g1 = lds_f4<o0 + 32>(pg), b0 = lds_f4<C * 4 + o0>(pg), b1v = lds_f4<C * 4 + o0 + 32>(pg);
      wait_f4(g0, g1, b0, b1v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[nt][4 * g + e] = (y[nt][4 * g + e] - mean) * rstd * g0[e] + b0[e];
        y[nt][4 * g + 4 + e] = (y[nt][4 * g + 4 + e] - mean) * rstd * g1[e] + b1v[e];
      }
    });
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_ptr)smem + lane * 16;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) {
    // a loop of its own (as a branch inside the chunk loop below it made hipcc park half of the main loop's state in
    // AGPRs: 470 us instead of ~290)
    for (int c = 0; c < NPRE; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of chunk c has landed
    __builtin_amdgcn_s_barrier();                        // everyone's has; everyone is done reading chunk c - 1
    asm volatile("" ::: "memory");
    const int cn = min(c + 1, nchunk - 1);               // the last chunk re-stages itself into the idle buffer: no branch
    const uint32_t wa = lds0 + (c & 1) * CHUNK;          // phase-A blocks; phase-B blocks follow at + W1_BLKS * BLK
    f16x8 fr[DEPTH + 1];
      // ---- the preceding projection: Y^T[n, m] += Wpre X^T, k-steps 8 c .. 8 c + 7 (held in xf[0..7]) of all eight
      // column tiles: 64 blocks, block 8 nt + j
      auto readp = [&](auto ii) {
        constexpr int i = decltype(ii)::value;
        fr[i % (DEPTH + 1)] = lds_frag<i * BLK>(wa);
      };
      static_for<0, DEPTH>(readp);
      static_for<0, 64>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if constexpr (i + DEPTH < 64) readp(ic<i + DEPTH>{});
        wait_frag<(63 - i < DEPTH ? 63 - i : DEPTH)>(fr[i % (DEPTH + 1)]);
        y[i / 8] = mfma32(fr[i % (DEPTH + 1)], xf[i % 8], y[i / 8]);
        if constexpr (i % 4 == 1) stage((c + 1) & 1, cn, i / 4);
      });
      stage((c + 1) & 1, cn, NSLOT - 1);
      if (c == 0) {
        // the second half of x's k-steps: in flight until the next chunk's vmcnt(0)
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[s] = *(const f16x8*)(xp + 16 * (8 + s));
      }
      if (c == NPRE - 1) {
        // src <- LayerNorm1(src + out_proj(x)); its f16 rounding is the feed-forward operand (hid_perm order, absorbed
        // by the packed W1), the f32 value + b2 the accumulator start of the second projection
        layernorm(0);
#pragma unroll
        for (int s = 0; s < C / 16; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) xf[s][e] = (f16)y[s >> 1][8 * (s & 1) + e];
        static_for<0, C / 32>([&](auto ii) {
          constexpr int nt = decltype(ii)::value;
          constexpr int ob = 2 * (C * 4) + (32 * nt) * 4;
          f32x4 c0 = lds_f4<ob>(par0), c1 = lds_f4<ob + 32>(par0), c2 = lds_f4<ob + 64>(par0), c3 = lds_f4<ob + 96>(par0);
          wait_f4(c0, c1, c2, c3);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[nt][e] += c0[e];
            y[nt][4 + e] += c1[e];
            y[nt][8 + e] += c2[e];
            y[nt][12 + e] += c3[e];
          }
        });
      }
    }
  }
  for (int c = PRE ? NPRE : 0; c < nchunk; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of chunk c has landed
    __builtin_amdgcn_s_barrier();                        // everyone's has; everyone is done reading chunk c - 1
    asm volatile("" ::: "memory");
    const int cn = min(c + 1, nchunk - 1);               // the last chunk re-stages itself into the idle buffer: no branch
    const uint32_t wa = lds0 + (c & 1) * CHUNK;          // phase-A blocks; phase-B blocks follow at + W1_BLKS * BLK
    const uint32_t wb = wa + W1_BLKS * BLK;
    f16x8 fr[DEPTH + 1];

    // step i < 34: phase A, k-step i / 2 of hidden tile i % 2;  i >= 34: phase B, k-step (i - 34) / 8 of column tile % 8
    auto read = [&](auto ii) {
      constexpr int i = decltype(ii)::value;
      if constexpr (i < W1_BLKS) fr[i % (DEPTH + 1)] = lds_frag<((i % 2) * KA + i / 2) * BLK>(wa);
      else fr[i % (DEPTH + 1)] = lds_frag<(((i - W1_BLKS) % 8) * (HC / 16) + (i - W1_BLKS) / 8) * BLK>(wb);
    };
    static_for<0, DEPTH>(read);
    f32x16 h[HC / 32];
    f16x8 hf[HC / 16];
    static_for<0, NBLK>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
#ifndef INK_FFN_NOREAD
      if constexpr (i + DEPTH < NBLK) read(ic<i + DEPTH>{});
#endif
#ifndef INK_FFN_NOREAD
      wait_frag<(NBLK - 1 - i < DEPTH ? NBLK - 1 - i : DEPTH)>(fr[i % (DEPTH + 1)]);
#else
      wait_frag<0>(fr[i % (DEPTH + 1)]);
#endif
      if constexpr (i < W1_BLKS) {
        constexpr int s = i / 2, jt = i % 2;
        h[jt] = mfma32(fr[i % (DEPTH + 1)], xf[s], s == 0 ? zero : h[jt]);
      } else {
        constexpr int s = (i - W1_BLKS) / 8, nt = (i - W1_BLKS) % 8;
        if constexpr (nt == 0) {
          // relu + f16: registers 8 (s & 1) .. + 7 of hidden tile s >> 1 are the lane's B operand of k-step s (hid_perm)
#pragma unroll
          for (int e = 0; e < 8; ++e) hf[s][e] = (f16)fmaxf(h[s >> 1][8 * (s & 1) + e], 0.0f);
        }
        y[nt] = mfma32(fr[i % (DEPTH + 1)], hf[s], y[nt]);
      }
#ifndef INK_FFN_NODMA          // (timing experiments of tools/ffn_variants.sh: wrong results)
      if constexpr (i % 4 == 1) stage((c + 1) & 1, cn, i / 4);
#endif
    });
  }

  layernorm(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the idle re-stage of the last chunk
  __syncthreads();                                       // all waves are done with the weight buffers
  char* ot = smem + wave * 32 * OROW;                    // wave-private tile: 32 rows of 256 f32
#pragma unroll
  for (int nt = 0; nt < C / 32; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = y[nt][4 * g + e];
      *(f32x4*)(ot + l * OROW + (32 * nt + 8 * g + 4 * hh) * 4) = v;
    }
  // (wave-private tile: the wave's own LDS writes are ordered before its reads by the lgkmcnt wait)
  // rows leave as 1-KiB segments: one row per instruction (64 lanes x 16 B)
#pragma unroll 4
  for (int r = 0; r < 32; ++r) {
    const int m = m0 + r;
    if (m < M) *(f32x4*)(out + (int64_t)m * C + lane * 4) = *(const f32x4*)(ot + r * OROW + lane * 16);
  }
}

}  // namespace

extern "C" int ink_ffn256_pack_bytes(int32_t hid, int32_t with_pre, int64_t* out_bytes) {
  INK_CHECK_ARG(out_bytes && hid > 0 && hid % HC == 0 && hid <= MAX_HID);
  *out_bytes = (int64_t)((with_pre ? NPRE : 0) + hid / HC) * CHUNK;
  return INK_OK;
}

extern "C" int ink_ffn256_pack(const void* w1_f16, const float* b1, const void* w2_f16, int32_t hid, const void* wpre_f16,
                               void* blob, void* stream) {
  INK_CHECK_ARG(w1_f16 && b1 && w2_f16 && blob && hid > 0 && hid % HC == 0 && hid <= MAX_HID);
  INK_CHECK_ARG((((uintptr_t)w1_f16 | (uintptr_t)blob | (uintptr_t)wpre_f16) & 15) == 0);
  const int64_t pieces = (int64_t)((wpre_f16 ? NPRE : 0) + hid / HC) * NBLK * 64;
  hipLaunchKernelGGL(ffn256_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)w1_f16, b1, (const f16*)w2_f16, hid, (const f16*)wpre_f16, (f16*)blob);
  return ink_launch_status();
}

extern "C" int ink_ffn256_fused(const void* x_f16, int64_t ldx, const float* res_f32, const void* blob, const float* b2,
                                const float* ln_g, const float* ln_b, float eps, int32_t M, int32_t hid,
                                const float* pre_bias, const float* pre_ln_g, const float* pre_ln_b, float* out_f32,
                                void* stream) {
  INK_CHECK_ARG(x_f16 && res_f32 && blob && b2 && ln_g && ln_b && out_f32);
  INK_CHECK_ARG(M > 0 && hid > 0 && hid % HC == 0 && hid <= MAX_HID && ldx >= C && ldx % 8 == 0);
  INK_CHECK_ARG((!pre_bias && !pre_ln_g && !pre_ln_b) || (pre_bias && pre_ln_g && pre_ln_b));
  INK_CHECK_ARG((((uintptr_t)x_f16 | (uintptr_t)res_f32 | (uintptr_t)blob | (uintptr_t)b2 | (uintptr_t)ln_g |
                  (uintptr_t)ln_b | (uintptr_t)out_f32 | (uintptr_t)pre_bias | (uintptr_t)pre_ln_g | (uintptr_t)pre_ln_b) & 15) == 0);
  static bool attr = ((void)hipFuncSetAttribute((const void*)ffn256_fused_kernel<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                      (void)hipFuncSetAttribute((const void*)ffn256_fused_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), true);
  (void)attr;
  const int ntiles = (M + BM - 1) / BM;
  if (pre_bias) {
    hipLaunchKernelGGL(ffn256_fused_kernel<true>, dim3(ntiles), dim3(256), LDS_BYTES, (hipStream_t)stream, (const f16*)x_f16,
                       ldx, res_f32, (const f16*)blob, b2, ln_g, ln_b, eps, M, hid, pre_bias, pre_ln_g, pre_ln_b, out_f32);
  } else {
    hipLaunchKernelGGL(ffn256_fused_kernel<false>, dim3(ntiles), dim3(256), LDS_BYTES, (hipStream_t)stream, (const f16*)x_f16,
                       ldx, res_f32, (const f16*)blob, b2, ln_g, ln_b, eps, M, hid, pre_bias, pre_ln_g, pre_ln_b, out_f32);
  }
  return ink_launch_status();
}

// Dense projection kernels for gfx950 (MI355X): C = epilogue(A[M,K] * W[N,K]^T), f16 operands, f32 accumulate.
//
// Two kernel families share one accumulator set-up / epilogue (init_wave_tile, store_wave_tile):
//   gemm_f16_nt_pp  ping-pong: 256x256 or 256x320 tile, 8 waves in two groups staggered by one barrier, ring of
//                   K32 granules with counted vmcnt - the SAM ViT-H projections (see its comment);
//   gemm_f16_nt     generic BMxBN tile, WMxWN waves of (BM/WM)x(BN/WN), two LDS stages with one drain + barrier per
//                   K-tile - 16-wave 256x256 for the other large shapes, 128x128 (K step 64 or 32) for the rest.
// Common to both (cdna_hip_programming.md §5):
//   - operands staged HBM -> LDS by LDS-DMA (global_load_lds_dwordx4); the LDS image is lane-linear (DMA
//     constraint), so the bank-conflict swizzle is applied to the per-lane SOURCE address and to the ds_read address:
//     16-B chunk c of row r lives at chunk c ^ swz(r);
//   - MFMA 16x16x32 f16 issued "swapped" (W fragment as the A operand) so each lane ends up with 4 CONSECUTIVE
//     output columns of one row;
//   - workgroup ids are remapped so each XCD (private 4 MiB L2) works on a contiguous band of tiles, GROUP_M
//     M-tiles x all N-tiles at a time, and re-reads its A / W panels from L2;
//   - epilogue (fused, f32): +bias, GELU/ReLU, *col_scale, +residual (preloaded into the accumulators for linear
//     GEMMs), optional row scatter (window-unpartition / crop / un-shift), f32 or f16 store as whole row segments.
// Variant choice: ink_gemm_query_variant; everything else in the switch of ink_gemm_f16 is a measurement build.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

template <int BK> struct Swz;
template <> struct Swz<64> {  // 128-B rows, 8 chunks
  static __device__ __forceinline__ int f(int row) { return row & 7; }
};
template <> struct Swz<32> {  // 64-B rows, 4 chunks
  static __device__ __forceinline__ int f(int row) { return (-(row >> 2)) & 3; }
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------------
// Accumulator set-up and epilogue of one wave tile (TM x TN MFMA tiles of 16x16; the lane holds
// C[m = mw + 16*ti + fr][n = nw + 16*j + 4*fq + 0..3]), shared by every kernel in this file.
//
// Loads and stores share vmcnt and retire in issue order in the counter, so a load issued AFTER a store can only
// be waited for with vmcnt(0) - i.e. by waiting for that store to reach memory (and the compiler has to assume
// the worst over all paths, so "counted" waits degrade to 0 as soon as a store may be skipped).  Measured with
// s_memrealtime stamps (tools/gemm_stamps.py): with the row_map / residual / bias loads inside the slab loop each
// of the 8 slabs of a 256x256 tile paid a full store round trip, 7.5 us per tile against a 31 us main loop.
// So the slab loop contains NO load:
//   * the residual of the linear case (no activation, no layer scale - every large GEMM of the pipeline) is
//     loaded straight into the accumulators before the K loop, row-mapped, and the MFMAs accumulate on top of it
//     (the four j-loads of a row cover 256 contiguous bytes; they overlap the pipeline fill);
//   * the output rows (wave_rows, one per accumulator row, redistributed with ds_bpermute) and the bias are
//     loaded before the first store;
//   * the rare non-linear residual / layer-scale loads stay in the loop, each used inside its own branch.
__device__ __forceinline__ bool residual_preloaded(const InkGemm& p) {
  return (p.residual || p.res_hi) && p.act == INK_ACT_NONE && !p.col_scale;
}

// Output row of every accumulator row of the wave tile (lane & 15 = row within the 16-row slab), -1 = outside M or
// dropped by the row map.  Loaded ONCE, before the first K-tile DMA: the residual preload indexes the residual
// with it and the epilogue redistributes it across lanes with ds_bpermute (no load next to the stores).
template <int TM>
__device__ __forceinline__ void wave_rows(int (&rows)[TM], const InkGemm& p, int mw, int lane) {
#pragma unroll
  for (int ti = 0; ti < TM; ++ti) {
    const int m = mw + ti * 16 + (lane & 15);
    int r = -1;
    if (m < p.M) r = p.row_map ? p.row_map[m] : m;
    rows[ti] = r;
  }
}

template <int TM, int TN, bool SPLIT = false>
__device__ __forceinline__ void init_wave_tile(f32x4 (&acc)[TM][TN], const InkGemm& p, const int (&rows)[TM], int nw,
                                               int lane) {
  const int fq = lane >> 4;
  const bool pre = residual_preloaded(p);
  if (SPLIT) {
    // residual stream kept as two f16 planes (value = hi + lo, ~22 significant bits): the hi plane doubles as the
    // f16 operand of the next projection, so no conversion pass ever reads the stream
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
      const int r = pre ? rows[ti] : -1;
      const size_t off = (size_t)max(r, 0) * p.ldr + nw + fq * 4;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[ti][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r >= 0 && nw + j * 16 + fq * 4 < p.N) {
          const f16x4 h = *(const f16x4*)((const f16*)p.res_hi + off + j * 16);
          const f16x4 l = *(const f16x4*)((const f16*)p.res_lo + off + j * 16);
          acc[ti][j] = (f32x4){(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                               (float)h[3] + (float)l[3]};
        }
      }
    }
    return;
  }
#pragma unroll
  for (int ti = 0; ti < TM; ++ti) {
    const int r = pre ? rows[ti] : -1;
    const float* rp = p.residual + (size_t)max(r, 0) * p.ldr + nw + fq * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[ti][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r >= 0 && nw + j * 16 + fq * 4 < p.N) acc[ti][j] = *(const f32x4*)(rp + j * 16);
    }
  }
}

// LayerNorm fold: the A operand of the projection is the RAW residual stream (its f16 hi plane) and W
// carries the norm's gamma, so  LN(x) W^T + b = rstd_m (acc_mn - mean_m s_n) + c_n  with s_n = sum_k W'[n,k] and
// c_n = beta W^T + b (passed as the bias).  mean / rstd of the wave tile's rows come from the partial (sum, sum of
// squares) the PRODUCER of the stream wrote next to it (stats_out of the previous projection): lane L reduces rows
// L, L + 64, ... of the wave tile and parks (rstd, rstd * mean) in a wave-private LDS strip; the epilogue reads them
// back per accumulator row.  Called before the K loop (its loads are ordinary loads: they must not sit between DMAs).
template <int TM, int TN>
__device__ __forceinline__ void ln_rows_prologue(const InkGemm& p, float2* strip, int mw, int nw, int lane) {
  // columns of the wave tile: (colsum, bias) pairs behind the row strip - the epilogue then reads everything it needs
  // from LDS (short latency, nothing to keep in registers across the K loop)
  float2* cols = strip + TM * 16;
  for (int c = lane; c < TN * 16; c += 64) {
    const int n = nw + c;
    cols[c] = n < p.N ? make_float2(p.ln_colsum[n], p.bias ? p.bias[n] : 0.f) : make_float2(0.f, 0.f);
  }
  // a row's partials are ln_parts (<= 20, even) contiguous float2: ten 16-B loads, all in flight before the first use
  // (a run-time-bounded scalar loop would serialise ln_parts dependent round trips per row)
  constexpr int RPL = (TM * 16 + 63) / 64;                  // rows per lane
  f32x4 v[RPL][10];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int r = lane + 64 * q, m = mw + r;
    const f32x4* sp = (const f32x4*)((const float2*)p.ln_stats + (size_t)min(m, p.M - 1) * p.ln_parts);
#pragma unroll
    for (int k = 0; k < 10; ++k) v[q][k] = 2 * k < p.ln_parts ? sp[k] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int r = lane + 64 * q;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      s1 += (double)v[q][k][0] + (double)v[q][k][2];
      s2 += (double)v[q][k][1] + (double)v[q][k][3];
    }
    const double mean = s1 / (double)p.ln_dim;
    const double var = fmax(s2 / (double)p.ln_dim - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)p.ln_eps));
    if (r < TM * 16) strip[r] = mw + r < p.M ? make_float2(rstd, rstd * (float)mean) : make_float2(0.f, 0.f);
  }
}

// Each 16-row slab goes through a wave-private LDS patch so that HBM sees whole row segments (16 B per lane,
// 128 B (f16) / 256 B (f32) contiguous per row); bias / activation / layer-scale are applied on the way in.
// MODE: -1 = activation / layer scale / late residual decided at run time (the generic kernels);  0, 1, 2 = compile-time
// "no activation" / GELU / ReLU with no layer scale and no late residual - the ping-pong kernel dispatches on it ONCE per
// workgroup, so that the 20 activations of a slab form one basic block (the per-group run-time branches cut the GELU
// into 4-element dependent chains: transcendental latency instead of throughput).
// OUT: 0 = f32 C, 1 = f16 C, 2 = split f16 (C = hi plane, C_lo = lo plane: hi = f16(v), lo = f16(v - hi)).
// p.stats_out: per output row and per wave-tile column chunk the partial (sum, sum of squares) of the final values -
// the LayerNorm statistics of the NEXT projection (ln_rows_prologue), computed here for free.
// p.ln_stats: this projection is itself a folded LayerNorm + Linear (see ln_rows_prologue); `strip` holds its rows.
// BIAS = false: the caller has added the bias already (add_bias_wave_tile) - the persistent kernel puts the next tile's
// pipeline fill between the bias loads and the stores, and then nothing in here loads from global memory.
template <int TM, int TN>
__device__ __forceinline__ void add_bias_wave_tile(f32x4 (&acc)[TM][TN], const InkGemm& p, int nw, int lane) {
  const int fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 16 + fq * 4;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) acc[ti][j] += bv;     // unconditional: no copy of the array at a join
  }
}

template <int TM, int TN, int OUT, int MODE = -1, bool LNF = false, bool BIAS = true>
__device__ __forceinline__ void store_wave_tile(f32x4 (&acc)[TM][TN], const InkGemm& p, char* er,
                                                const int (&rows)[TM], int nw, int lane, const float2* strip = nullptr) {
  constexpr bool F16O = OUT == 1;
  constexpr int WNC = TN * 16, EP = WNC * 4 + 16;
  constexpr int ES = OUT == 0 ? 4 : 8;             // elements per 16-B chunk of the output row(s)
  constexpr int CPRW = WNC / ES;                   // chunks per patch row
  constexpr int NIT = (16 * CPRW + 63) / 64;       // chunk rounds per slab
  const int fr = lane & 15, fq = lane >> 4;
  const bool wide16 = OUT != 0 && (p.ldc % 8 == 0);
  const bool res_late = MODE < 0 && p.residual && !residual_preloaded(p);
  const int act = MODE < 0 ? p.act : MODE;
  const bool scaled = MODE < 0 && p.col_scale != nullptr;
  // chunk c = it*64 + lane of a slab: patch row c / CPRW, chunk column c % CPRW (the same for every slab)
  int row_of[NIT], n_of[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    row_of[it] = c < 16 * CPRW ? c / CPRW : 16;
    n_of[it] = nw + (c % CPRW) * ES;
  }

  // the only loads of the epilogue, before the first store: bias (and the column sums of a folded LayerNorm);
  // LNF is a compile-time switch: a run-time branch around two versions of the accumulator update makes hipcc keep
  // both copies of the array alive at the join (spills)
  if (LNF) {
    const float2* cols = strip + TM * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 c01 = *(const f32x4*)(cols + j * 16 + fq * 4);       // (s, b) of columns 0, 1
      const f32x4 c23 = *(const f32x4*)(cols + j * 16 + fq * 4 + 2);   // ... of columns 2, 3
      const f32x4 sv = (f32x4){c01[0], c01[2], c23[0], c23[2]}, bv = (f32x4){c01[1], c01[3], c23[1], c23[3]};
#pragma unroll
      for (int ti = 0; ti < TM; ++ti) {
        const float2 mr = strip[ti * 16 + fr];               // (rstd, rstd * mean) of this accumulator row
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ti][j][e] = fmaf(mr.x, acc[ti][j][e], fmaf(-mr.y, sv[e], bv[e]));
      }
      __builtin_amdgcn_sched_barrier(0);                       // one column group at a time: 8 + 16 live values
    }
  } else if (BIAS) {
    add_bias_wave_tile<TM, TN>(acc, p, nw, lane);
  }

#pragma unroll
  for (int ti = 0; ti < TM; ++ti) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x4 v = acc[ti][j];
      if (act == INK_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (act == INK_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (scaled) {
        const int n = nw + j * 16 + fq * 4;
        if (n < p.N) v *= *(const f32x4*)(p.col_scale + n);
      }
      if (OUT == 2 && nw + j * 16 + fq * 4 < p.N) {
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
      if (F16O) {
        *(f16x4*)(er + fr * EP + (j * 16 + fq * 4) * 2) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      } else {
        *(f32x4*)(er + fr * EP + (j * 16 + fq * 4) * 4) = v;
      }
    }
    if (OUT == 2 && p.stats_out) {
      // the four lanes fr, fr + 16, fr + 32, fr + 48 hold the columns of accumulator row fr
      s1 += __shfl_xor(s1, 16, 64);
      s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (fq == 0 && rows[ti] >= 0)
        ((float2*)p.stats_out)[(size_t)rows[ti] * p.stats_parts + nw / WNC] = make_float2(s1, s2);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      // output row of patch row row_of[it]: held by the lanes with (lane & 15) == that row
      int r = __builtin_amdgcn_ds_bpermute((row_of[it] & 15) << 2, rows[ti]);
      if (row_of[it] >= 16 || n_of[it] >= p.N) r = -1;
      if (r >= 0) {
        const int c_n = n_of[it];
        const char* src = er + row_of[it] * EP + (c_n - nw) * (F16O ? 2 : 4);
        if (F16O) {
          f16x8 d = *(const f16x8*)src;
          f16* dst = (f16*)p.C + (size_t)r * p.ldc + c_n;
          if (res_late) {
            const float* rp = p.residual + (size_t)r * p.ldr + c_n;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c_n + e < p.N) d[e] = (f16)((float)d[e] + rp[e]);
          }
          if (wide16 && c_n + 8 <= p.N) {
            *(f16x8*)dst = d;
          } else {
            *(f16x4*)dst = (f16x4){d[0], d[1], d[2], d[3]};
            if (c_n + 8 <= p.N) *(f16x4*)(dst + 4) = (f16x4){d[4], d[5], d[6], d[7]};
          }
        } else if (OUT == 2) {
          const f32x4 d0 = *(const f32x4*)src, d1 = *(const f32x4*)(src + 16);
          f16x8 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = (f16)d0[e];
            lo[e] = (f16)(d0[e] - (float)hi[e]);
            hi[e + 4] = (f16)d1[e];
            lo[e + 4] = (f16)(d1[e] - (float)hi[e + 4]);
          }
          f16* dh = (f16*)p.C + (size_t)r * p.ldc + c_n;
          f16* dl = (f16*)p.C_lo + (size_t)r * p.ldc + c_n;
          if (wide16 && c_n + 8 <= p.N) {
            *(f16x8*)dh = hi;
            *(f16x8*)dl = lo;
          } else {
            *(f16x4*)dh = (f16x4){hi[0], hi[1], hi[2], hi[3]};
            *(f16x4*)dl = (f16x4){lo[0], lo[1], lo[2], lo[3]};
            if (c_n + 8 <= p.N) {
              *(f16x4*)(dh + 4) = (f16x4){hi[4], hi[5], hi[6], hi[7]};
              *(f16x4*)(dl + 4) = (f16x4){lo[4], lo[5], lo[6], lo[7]};
            }
          }
        } else {
          f32x4 d = *(const f32x4*)src;
          if (res_late) d += *(const f32x4*)(p.residual + (size_t)r * p.ldr + c_n);
          *(f32x4*)((float*)p.C + (size_t)r * p.ldc + c_n) = d;
        }
      }
    }
  }
}

// Generic tile: BM x BN output per workgroup, WM x WN waves (each (BM/WM) x (BN/WN)), K step BK,
// NS LDS stages.  NS == 2: one K-tile in flight, plain __syncthreads (drains the DMA).
// NS >= 3: NS-1 K-tiles in flight behind a COUNTED s_waitcnt vmcnt(N) + raw s_barrier
// (cdna_hip_programming.md §5 "Pipelining across barriers"): the wait that retires tile kt comes
// before the barrier, the reads of tile kt after it, and the buffer that is re-filled is the one read
// in the previous iteration (every wave has passed this iteration's barrier, i.e. finished those reads).
// EXT: the kernel also carries the ABI-4 forms (split-f16 output + row statistics, folded LayerNorm); only the 128x128
// tiles are built with it - the 16-wave 256x256 tile has 128 VGPRs per lane and no room for them.
template <int BM, int BN, int BK, int WM, int WN, int NS, int ABL = 0, bool EXT = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f16_nt(InkGemm p, int group_m) {
  constexpr int NT = WM * WN * 64;
  constexpr int CPR = BK / 8;          // 16-B chunks per tile row
  constexpr int ROWB = BK * 2;         // bytes per tile row
  constexpr int TILE_A = BM * ROWB, TILE_W = BN * ROWB;
  constexpr int STAGE = TILE_A + TILE_W;
  constexpr int IT_A = (BM * CPR) / NT, IT_W = (BN * CPR) / NT;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int LOADS = IT_A + IT_W;   // DMA instructions per thread per stage
  static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  // grouped order inside the XCD-contiguous id space: GM consecutive M-tiles x all N-tiles, M fastest, so the
  // ~32 tiles an XCD runs concurrently share GM A-panels and 32/GM W-panels (L2 = 4 MiB per XCD)
  int mt, nt;
  if (group_m > 1) {
    const int per = group_m * ntn;
    const int first = (id / per) * group_m;
    const int gsz = min(ntm - first, group_m);
    mt = first + (id % per) % gsz;
    nt = (id % per) / gsz;
  } else {
    mt = id / ntn;
    nt = id % ntn;
  }
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;

  const f16* srcA[IT_A];
  const f16* srcW[IT_W];
#pragma unroll
  for (int it = 0; it < IT_A; ++it) {
    const int pch = it * NT + tid;
    const int row = pch / CPR;
    const int lch = (pch % CPR) ^ Swz<BK>::f(row);
    srcA[it] = A + (size_t)min(m0 + row, p.M - 1) * p.lda + lch * 8;
  }
#pragma unroll
  for (int it = 0; it < IT_W; ++it) {
    const int pch = it * NT + tid;
    const int row = pch / CPR;
    const int lch = (pch % CPR) ^ Swz<BK>::f(row);
    srcW[it] = W + (size_t)min(n0 + row, p.N - 1) * p.ldw + lch * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int it = 0; it < IT_A; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[it] + kt * BK), (lptr_t)(base + (it * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < IT_W; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[it] + kt * BK), (lptr_t)(base + TILE_A + (it * NT + wave * 64) * 16), 16, 0, 0);
  };

  int rows[TM];
  wave_rows<TM>(rows, p, m0 + wm * (BM / WM), lane);
  f32x4 acc[TM][TN];
  // wave-private strip behind the staging buffers: (rstd, rstd * mean) of the wave tile's rows (folded LayerNorm)
  float2* strip = (float2*)(smem + NS * STAGE) + wave * (TM * 16 + TN * 16);
  if (EXT && p.ln_stats) ln_rows_prologue<TM, TN>(p, strip, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);

  const int fr = lane & 15, fq = lane >> 4;
  const int offA = (wm * (BM / WM) + fr) * ROWB;
  const int offW = (wn * (BN / WN) + fr) * ROWB;
  const int swz = Swz<BK>::f(fr);      // tile-row bases are multiples of 16 -> swz depends on fr only

  const int nk = p.K / BK;
  auto compute = [&](int cur) {
    const char* bA = smem + cur * STAGE;
    const char* bW = bA + TILE_A;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      constexpr int AG = TM > 4 ? 4 : TM;    // A fragments live at a time (big wave tiles: registers)
      f16x8 a[AG], w[TN];
      const int co = ((kk * 4 + fq) ^ swz) << 4;
#pragma unroll
      for (int j = 0; j < TN; ++j) w[j] = *(const f16x8*)(bW + offW + j * 16 * ROWB + co);
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += AG) {
#pragma unroll
        for (int i = 0; i < AG; ++i) a[i] = *(const f16x8*)(bA + offA + (i0 + i) * 16 * ROWB + co);
#pragma unroll
        for (int i = 0; i < AG; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], acc[i0 + i][j], 0, 0, 0);
      }
    }
  };

  if constexpr (NS == 2) {
    stage(0, 0);
    if (EXT && p.res_hi) {
      init_wave_tile<TM, TN, true>(acc, p, rows, n0 + wn * (BN / WN), lane);
    } else {
      init_wave_tile<TM, TN>(acc, p, rows, n0 + wn * (BN / WN), lane);
    }
    for (int kt = 0; kt < nk; ++kt) {
      // the LDS-DMA of tile kt is tracked by vmcnt only: drain it EXPLICITLY before the barrier (whether
      // __syncthreads() alone emits the vmcnt wait depends on what else the compiler sees in flight)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk && (ABL != 1 || kt == 0)) stage((kt + 1) & 1, kt + 1);
      if (ABL != 2 || kt + 1 == nk) compute(kt & 1);
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) stage(s, s);
    init_wave_tile<TM, TN>(acc, p, rows, n0 + wn * (BN / WN), lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the counted waits below assume only DMA in flight)
    int cur = 0, nxt = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
      // tiles issued after kt and still wanted in flight: min(NS-2, nk-1-kt)
      if (kt + NS - 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NS - 1 < nk) stage(nxt, kt + NS - 1);
      compute(cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
  }

  // ---- epilogue: lane holds C[m = .. + fr][n = .. + 4*fq + 0..3] for each (i,j)
  if (ABL == 3) {   // ablation: no epilogue (keep the accumulators live)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) ((float*)p.C)[0] = sum;
    return;
  }
  constexpr int WNC = BN / WN;                 // columns of the wave tile
  constexpr int EP = WNC * 4 + 16;             // patch row pitch in bytes (f32 worst case + pad)
  static_assert(WM * WN * 16 * EP <= NS * STAGE, "epilogue patch must fit in the staging LDS");
  __syncthreads();                             // every wave is done reading the last K-tile
  char* er = smem + wave * (16 * EP);
  if (EXT && p.ln_stats) {
    if (p.c_f16 == 1) {
      store_wave_tile<TM, TN, 1, -1, true>(acc, p, er, rows, n0 + wn * WNC, lane, strip);
    } else {
      store_wave_tile<TM, TN, 0, -1, true>(acc, p, er, rows, n0 + wn * WNC, lane, strip);
    }
  } else if (EXT && p.c_f16 == 2) {
    store_wave_tile<TM, TN, 2>(acc, p, er, rows, n0 + wn * WNC, lane);
  } else if (p.c_f16 == 1) {
    store_wave_tile<TM, TN, 1>(acc, p, er, rows, n0 + wn * WNC, lane);
  } else {
    store_wave_tile<TM, TN, 0>(acc, p, er, rows, n0 + wn * WNC, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong kernel: 256x256 tile, 8 waves = two groups of 4 (group = wave / 4 owns 128 rows; one wave of each
// group per SIMD), K streamed as a ring of RING granules of 32 (32 KiB each: A 256x32 + W 256x32 f16).
// The groups run the same program staggered by ONE barrier, so in every slot one group issues its 32 MFMAs
// while the other reads its 12 fragments from LDS and issues the LDS-DMA of the granule RING-1 ahead:
//     slot      2g          2g+1        2g+2
//     group 0   LOAD(g)     MFMA(g)     LOAD(g+1)
//     group 1   MFMA(g-1)   LOAD(g)     MFMA(g)
// Granule g sits in ring slot g % RING; both groups have read it by the end of slot 2g+1, and the DMA that
// overwrites it (granule g+RING) is issued in slots 2g+2 / 2g+3.  Every wave ends its LOAD slot with a COUNTED
// vmcnt (RING-2 granules stay in flight) and every slot ends with a raw s_barrier, so a granule is only read after
// the wait + barrier that retire it (cdna_hip_programming.md §5 "Read a staged buffer one phase AFTER the wait").
// 8 waves x <=256 VGPRs leave room for the load-free epilogue (store_wave_tile), which the 16-wave tiles lack.
//
// ABL (ablation / instrumentation builds, reachable through ink_gemm_set_variant only): 1 no MFMA, 2 no LDS
// reads / barriers, 4 every tile reads tile (0,0) (all L2 hits), 8 every workgroup writes (HW_ID, XCC_ID,
// t_entry, t_filled, t_loop_end, t_stores_issued) in 100 MHz ticks through p.residual (tools/gemm_stamps.py).
// TN = 4: 256x256 tile, TN = 5: 256x320 (wave tile 128 x 16*TN; 160 accumulator VGPRs).  LATE_A keeps only four of
// the eight A fragments live: rows 4-7 are read DURING the MFMA slot into the registers of rows 0-3 as soon as
// those have issued their MFMAs.  A granule is then still being read one slot later, so its ring slot may only be
// refilled one granule later: the DMA runs RING-2 granules ahead instead of RING-1.
// EPI: 0 = the epilogue is chosen at run time among the ABI-3 forms (bias / activation / layer scale / f32 residual,
// f32 or f16 C); 1, 2, 3 = ONE compiled epilogue each for the SAM ViT-H block on the split-f16 residual stream:
// 1 = folded LayerNorm -> f16 (qkv), 2 = folded LayerNorm + GELU -> f16 (lin1), 3 = split residual in, split C + row
// statistics out (proj, lin2).  Separate kernels rather than more branches: with every form inside one kernel the
// register allocator spilled around the dispatch.
// EPI == 4 of the ping-pong kernel: an asm global load with a scalar base and a 32-bit lane offset (asm: hipcc turns
// its own vmcnt waits into vmcnt(0) while an LDS-DMA is in flight; the loop's counted waits cover this load, see there)
__device__ __forceinline__ void rmf_global_load(f32x4& dst, const float* base, uint32_t byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for_pp(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_pp<I + 1, N>(f);
  }
}

template <int RING, int TN, int ABL = 0, bool LATE_A = (TN > 4), int EPI = 0>
__global__ __launch_bounds__(512) void gemm_f16_nt_pp(InkGemm p, int group_m) {
  constexpr int BM = 256, BN = 64 * TN, BK = 32, NT = 512;
  constexpr int CPR = BK / 8, ROWB = BK * 2;
  constexpr int TILE_A = BM * ROWB, TILE_W = BN * ROWB, GRAN = TILE_A + TILE_W;   // 32 KiB
  constexpr int IT_A = (BM * CPR) / NT, IT_W = (BN * CPR) / NT;                    // 2 + 2 DMA per thread
  constexpr bool W_TAIL = (BN * CPR) % NT != 0;       // 320 rows: a third, half-populated round (waves 0-3 = group 0)
  static_assert(!W_TAIL || (BN * CPR) % NT == NT / 2, "tail is exactly the first four waves");
  constexpr int LOADS = IT_A + IT_W;                  // per wave of group 1; group 0 issues one more with W_TAIL
  constexpr int TM = 8, WNC = 16 * TN;
  constexpr int AHEAD = LATE_A ? RING - 2 : RING - 1;   // granules the DMA runs ahead of the LOAD slot
  constexpr int AG = LATE_A ? 4 : TM;                   // A fragments live at a time
  static_assert(AHEAD >= 1, "ring too small");
  constexpr int EP = WNC * 4 + 16;
  static_assert(8 * 16 * EP <= RING * GRAN, "patch fits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;         // grp = wave row (128 rows each), wn = wave column (64 cols)
  const int fr = lane & 15, fq = lane >> 4;

  unsigned wg_rt[4] = {0, 0, 0, 0}, wg_cy[4] = {0, 0, 0, 0};
  auto stamp_rt = [&](int k) {
    if ((ABL & 8) && wave == 0) {
      wg_rt[k] = (unsigned)__builtin_amdgcn_s_memrealtime();
      wg_cy[k] = (unsigned)__builtin_readcyclecounter();
    }
  };
  stamp_rt(0);
  const float* dbg_out = EPI ? p.col_scale : p.residual;        // (stamp builds: the timeline leaves through a spare pointer)
  if (ABL & 8) {
    if (EPI) p.col_scale = nullptr; else p.residual = nullptr;
  }

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  int mt, nt;
  const int gm_abs = group_m < 0 ? -group_m : group_m;
  if (gm_abs > 1) {
    const int per = gm_abs * ntn;
    const int first = (id / per) * gm_abs;
    const int gsz = min(ntm - first, gm_abs);
    mt = first + (id % per) % gsz;
    nt = (id % per) / gsz;
  } else {
    mt = id / ntn;
    nt = id % ntn;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int sm0 = (ABL & 4) ? 0 : m0, sn0 = (ABL & 4) ? 0 : n0;
  const int G = p.K / BK;
  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  const f16* srcA[IT_A];
  const f16* srcW[IT_W + (W_TAIL ? 1 : 0)];
#pragma unroll
  for (int it = 0; it < IT_A; ++it) {
    const int pch = it * NT + tid, row = pch / CPR, lch = (pch % CPR) ^ Swz<BK>::f(row);
    srcA[it] = A + (size_t)min(sm0 + row, p.M - 1) * p.lda + lch * 8;
  }
#pragma unroll
  for (int it = 0; it < IT_W + (W_TAIL ? 1 : 0); ++it) {
    const int pch = it * NT + tid, row = min(pch / CPR, BN - 1), lch = (pch % CPR) ^ Swz<BK>::f(row);
    srcW[it] = W + (size_t)min(sn0 + row, p.N - 1) * p.ldw + lch * 8;
  }
  auto dma = [&](int g, int slot) {
    char* base = smem + slot * GRAN;
#pragma unroll
    for (int it = 0; it < IT_A; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[it] + g * BK), (lptr_t)(base + (it * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < IT_W; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[it] + g * BK), (lptr_t)(base + TILE_A + (it * NT + wave * 64) * 16), 16, 0, 0);
    if (W_TAIL && grp == 0)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[IT_W] + g * BK), (lptr_t)(base + TILE_A + (IT_W * NT + wave * 64) * 16), 16, 0, 0);
  };
  // counted wait that leaves the `ahead` most recent granules of THIS wave in flight
  auto wait_ahead = [&](auto ahead_c) {
    constexpr int ahead = decltype(ahead_c)::value;
    if (W_TAIL && grp == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ahead * (LOADS + 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ahead * LOADS) : "memory");
    }
  };
  auto slot_end = [&]() {
    if (ABL & 2) return;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: residual row indices, then RING-1 granules in flight (the launcher guarantees G >= RING-1), then
  // the residual preload behind them; with a preload everything is drained once (the counted waits of the loop
  // assume only DMA in flight), without one only granule 0 is waited for
  int rows[TM];
  wave_rows<TM>(rows, p, m0 + grp * 128, lane);
  float2* strip = (float2*)(smem + RING * GRAN) + wave * (128 + WNC);   // behind the ring: rows + columns of the fold
#pragma unroll
  for (int g = 0; g < AHEAD; ++g) dma(g, g);
  if (EPI == 1 || EPI == 2) ln_rows_prologue<TM, TN>(p, strip, m0 + grp * 128, n0 + wn * WNC, lane);   // (drains the fill, like a preload)
  f32x4 acc[TM][TN];
  if constexpr (EPI == 4) {
    // the f32 residual enters THROUGH THE MFMA PIPE during the K loop (below) instead of being preloaded into the
    // accumulators: nothing to wait for here but granule 0
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[ti][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wait_ahead(std::integral_constant<int, AHEAD - 1>{});
  } else {
    init_wave_tile<TM, TN, EPI == 3>(acc, p, rows, n0 + wn * WNC, lane);
    if (residual_preloaded(p) || EPI == 1 || EPI == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      wait_ahead(std::integral_constant<int, AHEAD - 1>{});
    }
  }
  slot_end();
  if (grp == 1) slot_end();                         // the stagger: group 1 idles through slot 0
  stamp_rt(1);

  const int offA = (grp * 128 + fr) * ROWB;
  const int offW = (wn * WNC + fr) * ROWB;
  const int co = (fq ^ Swz<BK>::f(fr)) << 4;        // one k-step of 32 per granule: logical chunk = fq
  f16x8 a[AG], w[TN];
  int cslot = 0, islot = AHEAD % RING;
  auto iter = [&](int g) __attribute__((always_inline)) {
    const char* bA = smem + cslot * GRAN;
    const char* bW = bA + TILE_A;
    // ---- LOAD slot
    if (!(ABL & 2) || g == 0) {
#pragma unroll
      for (int i = 0; i < AG; ++i) a[i] = *(const f16x8*)(bA + offA + i * 16 * ROWB + co);
#pragma unroll
      for (int j = 0; j < TN; ++j) w[j] = *(const f16x8*)(bW + offW + j * 16 * ROWB + co);
    }
    if (g + AHEAD < G) {
      dma(g + AHEAD, islot);
      wait_ahead(std::integral_constant<int, AHEAD - 1>{});   // granule g+1 of this wave landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    slot_end();
    // ---- MFMA slot
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (!(ABL & 1)) {
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += AG) {
#pragma unroll
        for (int i = 0; i < AG; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], acc[i0 + i][j], 0, 0, 0);
          if (LATE_A && i0 + AG < TM) {     // this fragment register is free: fetch the row AG further down
            a[i] = *(const f16x8*)(bA + offA + (i0 + AG + i) * 16 * ROWB + co);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AG; ++i) acc[i][0][0] += (float)a[i][0] + (float)w[i % TN][1];
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto iter_end = [&]() __attribute__((always_inline)) {
    slot_end();
    cslot = (cslot + 1 == RING) ? 0 : cslot + 1;
    islot = (islot + 1 == RING) ? 0 : islot + 1;
  };
  if constexpr (EPI != 4) {
    for (int g = 0; g < G; ++g) {
      iter(g);
      iter_end();
    }
  } else {
    // ---- residual through the MFMA pipe.  The 40 accumulator tiles (ti, j) of the wave take their residual block
    // R[16 rows, 16 columns] as  acc += I16 . hi(R)^T + I16 . lo(R)^T  (two v_mfma_f32_16x16x16_f16 with a 16 x 16
    // identity as the first operand; hi = f16(R), lo = f16(R - hi): R to 2^-22), one tile every G / 40 granules.  The
    // lane's f32x4 of R (the address the preload read) is fetched one step earlier by an asm global load issued in
    // the MFMA slot, i.e. AFTER that granule's LDS-DMA: it is older than the next granule's DMA pieces, so the loop's
    // counted vmcnt waits - unchanged - also cover it.  (With the preload all 256 workgroups of a round read their
    // 328-KB residual tiles at once, before any MFMA: 17 us per round, 34 of proj's 135-146 us.)  The K loop is split
    // into TM static blocks so that the accumulator row ti is a compile-time index and only j is a 5-way branch.
    typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
    f32x4 rraw = {0.f, 0.f, 0.f, 0.f};
    f16x4v idv;
#pragma unroll
    for (int e = 0; e < 4; ++e) idv[e] = (fr == 4 * fq + e) ? (f16)1.0f : (f16)0.0f;
    const int gpb = G / TM, rstride = gpb / TN;          // granules per block / per residual step
    const uint32_t roff0 = (uint32_t)(n0 + wn * WNC + fq * 4) * 4u;
    auto rmf_apply = [&](auto tti, auto jj) __attribute__((always_inline)) {
      constexpr int ti = decltype(tti)::value, j = decltype(jj)::value;
      f16x4v hi, lo;
      asm volatile("" : "+v"(rraw));       // (stays behind the loop's vmcnt waits, like every volatile asm)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (f16)rraw[e];
        lo[e] = (f16)(rraw[e] - (float)hi[e]);
      }
      if (rows[ti] < 0) { hi = (f16x4v){0, 0, 0, 0}; lo = hi; }
      acc[ti][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(idv, hi, acc[ti][j], 0, 0, 0);
      acc[ti][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(idv, lo, acc[ti][j], 0, 0, 0);
    };
    auto rmf_load = [&](auto tti, int j) __attribute__((always_inline)) {
      constexpr int ti = decltype(tti)::value;
      rmf_global_load(rraw, p.residual, (uint32_t)max(rows[ti], 0) * (uint32_t)p.ldr * 4u + roff0 + (uint32_t)j * 64u);
    };
    static_for_pp<0, TM>([&](auto tti) {
      constexpr int ti = decltype(tti)::value;
      for (int gi = 0; gi < gpb; ++gi) {
        iter(ti * gpb + gi);
        if (gi % rstride == 0) {
          const int j = gi / rstride;          // 0 .. TN - 1: apply the block loaded one step ago, then load (ti, j)
          if (j == 0) {
            if constexpr (ti > 0) rmf_apply(std::integral_constant<int, (ti > 0 ? ti - 1 : 0)>{}, std::integral_constant<int, TN - 1>{});
          } else if (j == 1) {
            rmf_apply(tti, std::integral_constant<int, 0>{});
          } else if (j == 2) {
            rmf_apply(tti, std::integral_constant<int, 1>{});
          } else if (j == 3) {
            rmf_apply(tti, std::integral_constant<int, 2>{});
          } else {
            rmf_apply(tti, std::integral_constant<int, (TN > 4 ? 3 : 0)>{});
          }
          rmf_load(tti, j);
        }
        iter_end();
      }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the last residual block
    rmf_apply(std::integral_constant<int, TM - 1>{}, std::integral_constant<int, TN - 1>{});
  }
  if (grp == 0) slot_end();                         // group 0 idles through the last slot (same barrier count)
  stamp_rt(2);

  // ---- epilogue: the drained ring is the patch space
  char* er = smem + wave * (16 * EP);
  // (group_m < 0: A/B switch of tools/gemm_res_ab.py - take the run-time-dispatch epilogue everywhere)
  const bool plain = group_m > 0 && !p.col_scale && !(p.residual && !residual_preloaded(p));
  if constexpr (EPI == 1) {
    store_wave_tile<TM, TN, 1, INK_ACT_NONE, true>(acc, p, er, rows, n0 + wn * WNC, lane, strip);
  } else if constexpr (EPI == 2) {
    store_wave_tile<TM, TN, 1, INK_ACT_GELU, true>(acc, p, er, rows, n0 + wn * WNC, lane, strip);
  } else if constexpr (EPI == 3) {
    store_wave_tile<TM, TN, 2, INK_ACT_NONE>(acc, p, er, rows, n0 + wn * WNC, lane);
  } else if constexpr (EPI == 4) {
    store_wave_tile<TM, TN, 0, INK_ACT_NONE>(acc, p, er, rows, n0 + wn * WNC, lane);
  } else if (p.c_f16) {
    if (plain && p.act == INK_ACT_GELU) {          // lin1 of the ViT-H MLP
      store_wave_tile<TM, TN, 1, INK_ACT_GELU>(acc, p, er, rows, n0 + wn * WNC, lane);
    } else if (plain && p.act == INK_ACT_NONE) {   // qkv
      store_wave_tile<TM, TN, 1, INK_ACT_NONE>(acc, p, er, rows, n0 + wn * WNC, lane);
    } else {
      store_wave_tile<TM, TN, 1>(acc, p, er, rows, n0 + wn * WNC, lane);
    }
  } else {
    if (plain && p.act == INK_ACT_NONE) {          // proj, lin2 (residual preloaded into the accumulators)
      store_wave_tile<TM, TN, 0, INK_ACT_NONE>(acc, p, er, rows, n0 + wn * WNC, lane);
    } else {
      store_wave_tile<TM, TN, 0>(acc, p, er, rows, n0 + wn * WNC, lane);
    }
  }
  if (ABL & 8) {
    stamp_rt(3);                                    // stores issued (not retired)
    if (wave == 0 && lane == 0) {
      unsigned* out = (unsigned*)dbg_out + (size_t)blockIdx.x * 8;
      out[0] = __builtin_amdgcn_s_getreg(63492);    // HW_ID
      out[1] = __builtin_amdgcn_s_getreg(6164);     // XCC_ID
      out[2] = wg_rt[0]; out[3] = wg_rt[1]; out[4] = wg_rt[2]; out[5] = wg_rt[3];
      out[6] = wg_cy[2] - wg_cy[1];                 // shader cycles of the main loop
      out[7] = wg_cy[3] - wg_cy[2];                 // ... of the epilogue
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of the 256x320 ping-pong kernel (ring of 4, LATE_A) for the three plain epilogues of the ViT-H
// block: FORM 0 = f16 C (qkv), 1 = GELU -> f16 C (lin1), 2 = f32 C with an optional preloaded f32 residual (proj, lin2).
// One workgroup per CU walks tiles b, b + grid, b + 2 grid, ... (the order the hardware dispatches the one-tile
// kernel's workgroups in, so xcd_remap keeps its meaning).  The K loop of a tile is the one-tile kernel's.  What changes
// is the tile boundary: a K = 1280 tile of the one-tile kernel is 2.2-3.9 us of pipeline fill + 37 us of loop + 4.6-7.4 us
// of epilogue + ~1 us until the next workgroup starts on the CU, and the fill cannot start before the previous
// workgroup's stores have retired.  Here the first two granules of tile t+1 are requested BEFORE the stores of tile t
// are issued (the bias loads come first and are waited for, so that no load of the epilogue sits behind the DMA), and
// one vmcnt(0) at the top of tile t+1 retires stores, fill and residual preload together.  The ring simply continues
// (granule g of the next tile goes to the slot after the last one); the epilogue's patches live in the two slots the
// fill does not use (every wave has left the K loop when the fill is issued: the groups' barrier counts are equal).
// The earlier persistent attempt (round 1) kept the COUNTED waits running across tiles and over-waited on the stores.
template <int TN, int FORM>
__global__ __launch_bounds__(512) void gemm_f16_nt_pp_persist(InkGemm p, int group_m) {
  constexpr int RING = 4, BM = 256, BN = 64 * TN, BK = 32, NT = 512;
  constexpr int CPR = BK / 8, ROWB = BK * 2;
  constexpr int TILE_A = BM * ROWB, TILE_W = BN * ROWB, GRAN = TILE_A + TILE_W;
  constexpr int IT_A = (BM * CPR) / NT, IT_W = (BN * CPR) / NT;
  constexpr bool W_TAIL = (BN * CPR) % NT != 0;
  static_assert(!W_TAIL || (BN * CPR) % NT == NT / 2, "tail is exactly the first four waves");
  constexpr int LOADS = IT_A + IT_W;
  constexpr int TM = 8, WNC = 16 * TN, AHEAD = RING - 2, AG = 4;
  constexpr int EP = WNC * 4 + 16;
  static_assert(4 * 16 * EP <= GRAN, "the patches of four waves fit in one ring slot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM, ntiles = ntm * ntn;      // (launcher: N % BN == 0)
  const int G = p.K / BK;
  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  // DMA source rows: pointers formed as scalar base + 32-bit byte offset (the launcher checks that both operands are
  // below 4 GiB) - with 64-bit index arithmetic hipcc keeps copies of the two bases in VGPRs across the K loop
  const char* srcA[IT_A];
  const char* srcW[IT_W + (W_TAIL ? 1 : 0)];
  int m0, n0;
  auto set_tile = [&](int v) {       // tile coordinates + DMA source rows of virtual workgroup v
    const int id = xcd_remap(v, ntiles);
    int mt, nt;
    if (group_m > 1) {
      const int per = group_m * ntn;
      const int first = (id / per) * group_m;
      const int gsz = min(ntm - first, group_m);
      mt = first + (id % per) % gsz;
      nt = (id % per) / gsz;
    } else {
      mt = id / ntn;
      nt = id % ntn;
    }
    m0 = mt * BM;
    n0 = nt * BN;
#pragma unroll
    for (int it = 0; it < IT_A; ++it) {
      const int pch = it * NT + tid, row = pch / CPR, lch = (pch % CPR) ^ Swz<BK>::f(row);
      srcA[it] = (const char*)A + ((uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.lda + (uint32_t)lch * 8u) * 2u;
    }
#pragma unroll
    for (int it = 0; it < IT_W + (W_TAIL ? 1 : 0); ++it) {
      const int pch = it * NT + tid, row = min(pch / CPR, BN - 1), lch = (pch % CPR) ^ Swz<BK>::f(row);
      srcW[it] = (const char*)W + ((uint32_t)(n0 + row) * (uint32_t)p.ldw + (uint32_t)lch * 8u) * 2u;
    }
  };
  auto dma = [&](int g, int slot) {
    char* base = smem + slot * GRAN;
#pragma unroll
    for (int it = 0; it < IT_A; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[it] + g * (BK * 2)), (lptr_t)(base + (it * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < IT_W; ++it)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[it] + g * (BK * 2)), (lptr_t)(base + TILE_A + (it * NT + wave * 64) * 16), 16, 0, 0);
    if (W_TAIL && grp == 0)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[IT_W] + g * (BK * 2)), (lptr_t)(base + TILE_A + (IT_W * NT + wave * 64) * 16), 16, 0, 0);
  };
  auto wait_one_granule = [&]() {      // leaves the most recent granule of THIS wave in flight (AHEAD - 1 = 1)
    if (W_TAIL && grp == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    }
  };
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int offA = (grp * 128 + fr) * ROWB;
  const int offW = (wn * WNC + fr) * ROWB;
  const int co = (fq ^ Swz<BK>::f(fr)) << 4;
  int cslot = 0;                      // ring slot of the next granule to be consumed
  int v = blockIdx.x;
  set_tile(v);
  // output rows of the wave tile at m0 (no row map here: the launcher sends those to the one-tile kernel); recomputed
  // where they are needed instead of living across the K loop
  auto tile_rows = [&](int (&rows)[TM], int mbase) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
      const int m = mbase + grp * 128 + ti * 16 + fr;
      rows[ti] = m < p.M ? m : -1;
    }
  };
#pragma unroll
  for (int g = 0; g < AHEAD; ++g) dma(g, g);

  for (;;) {
    // ---- top of a tile: its first AHEAD granules are in flight (behind the previous tile's stores)
    f32x4 acc[TM][TN];
    if constexpr (FORM == 2) {
      int rows[TM];
      tile_rows(rows, m0);
      init_wave_tile<TM, TN>(acc, p, rows, n0 + wn * WNC, lane);
    } else {
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[ti][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    slot_end();
    if (grp == 1) slot_end();                       // the stagger: group 1 idles through slot 0
    f16x8 a[AG], w[TN];
    int islot = (cslot + AHEAD) & (RING - 1);
    for (int g = 0; g < G; ++g) {
      const char* bA = smem + cslot * GRAN;
      const char* bW = bA + TILE_A;
      // ---- LOAD slot
#pragma unroll
      for (int i = 0; i < AG; ++i) a[i] = *(const f16x8*)(bA + offA + i * 16 * ROWB + co);
#pragma unroll
      for (int j = 0; j < TN; ++j) w[j] = *(const f16x8*)(bW + offW + j * 16 * ROWB + co);
      if (g + AHEAD < G) {
        dma(g + AHEAD, islot);
        wait_one_granule();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      slot_end();
      // ---- MFMA slot
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += AG) {
#pragma unroll
        for (int i = 0; i < AG; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], acc[i0 + i][j], 0, 0, 0);
          if (i0 + AG < TM) {
            a[i] = *(const f16x8*)(bA + offA + (i0 + AG + i) * 16 * ROWB + co);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      slot_end();
      cslot = (cslot + 1) & (RING - 1);
      islot = (islot + 1) & (RING - 1);
    }
    if (grp == 0) slot_end();                       // group 0 idles through the last slot (same barrier count)

    // ---- tile boundary.  Every wave has issued its last ring read (group 1's LATE_A reads precede the barrier
    // group 0 has just passed).  Bias first: the epilogue's only loads.  The empty asm USES the loaded registers, so the
    // compiler's wait for them sits here, in front of the DMA (behind it, it would be a vmcnt(0) that waits for the fill)
    f32x4 bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv[j] = *(const f32x4*)(p.bias + n0 + wn * WNC + j * 16 + fq * 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bv[j]) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ... then the next tile's fill into slots cslot, cslot + 1 ...
    const int n0_cur = n0, m0_cur = m0;
    v += gridDim.x;
    const bool more = v < ntiles;
    if (more) {
      set_tile(v);
      __builtin_amdgcn_sched_barrier(0);            // (all address arithmetic in front of the first DMA piece)
#pragma unroll
      for (int g = 0; g < AHEAD; ++g) dma(g, (cslot + g) & (RING - 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ... then the stores, patches in slots cslot + 2 (group 0) and cslot + 3 (group 1)
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[ti][j] += bv[j];
    int rows[TM];
    tile_rows(rows, m0_cur);
    char* er = smem + ((cslot + 2 + grp) & (RING - 1)) * GRAN + wn * (16 * EP);
    if constexpr (FORM == 0) {
      store_wave_tile<TM, TN, 1, INK_ACT_NONE, false, false>(acc, p, er, rows, n0_cur + wn * WNC, lane);
    } else if constexpr (FORM == 1) {
      store_wave_tile<TM, TN, 1, INK_ACT_GELU, false, false>(acc, p, er, rows, n0_cur + wn * WNC, lane);
    } else {
      store_wave_tile<TM, TN, 0, INK_ACT_NONE, false, false>(acc, p, er, rows, n0_cur + wn * WNC, lane);
    }
    if (!more) break;
  }
}

template <int TN, int FORM>
static int launch_gemm_pp_persist(const InkGemm& p, hipStream_t s, int group_m) {
  constexpr int BN = 64 * TN;
  constexpr int lds = 4 * (256 + BN) * 32 * 2;
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (p.K / 32 < 4 || p.N % BN != 0 || p.ln_stats || p.c_f16 == 2 || p.res_hi || p.col_scale || p.row_map) return INK_ERR_ARG;
  if (p.residual && p.act != INK_ACT_NONE) return INK_ERR_ARG;      // (a late residual: the one-tile kernel)
  if ((int64_t)p.M * p.lda * 2 >= ((int64_t)1 << 32) || (int64_t)p.N * p.ldw * 2 >= ((int64_t)1 << 32)) return INK_ERR_ARG;
  static int n_cu = [] {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipFuncSetAttribute((const void*)gemm_f16_nt_pp_persist<TN, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return n > 0 ? n : 256;
  }();
  const int ntiles = ((p.M + 255) / 256) * (p.N / BN);
  hipLaunchKernelGGL((gemm_f16_nt_pp_persist<TN, FORM>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), lds, s, p, group_m);
  return ink_launch_status();
}

template <int RING, int TN = 4, int ABL = 0, bool LATE_A = (TN > 4), int EPI = 0>
static int launch_gemm_pp(const InkGemm& p, hipStream_t s, int group_m) {
  if (p.K / 32 < RING) return 1;
  if (EPI == 0 && (p.ln_stats || p.c_f16 == 2 || p.res_hi)) return INK_ERR_ARG;
  if (EPI == 4 && !(TN == 5 && p.residual && !p.c_f16 && p.act == INK_ACT_NONE && !p.col_scale && (p.K / 32) % 40 == 0 &&
                    (int64_t)p.M * p.ldr * 4 < ((int64_t)1 << 32))) return INK_ERR_ARG;
  constexpr int BN = 64 * TN;
  constexpr int lds = RING * (256 + BN) * 32 * 2 + 8 * (128 + 16 * TN) * 8;   // ring + the LayerNorm-fold strips of the 8 waves
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr = ((void)hipFuncSetAttribute((const void*)gemm_f16_nt_pp<RING, TN, ABL, LATE_A, EPI>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)attr;
  const int ntiles = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_f16_nt_pp<RING, TN, ABL, LATE_A, EPI>), dim3(ntiles), dim3(512), lds, s, p, group_m);
  return ink_launch_status();
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int ABL = 0, bool EXT = false>
static int launch_gemm(const InkGemm& p, hipStream_t s, int group_m = 1) {
  constexpr int lds = NS * (BM + BN) * BK * 2 + (EXT ? (BM * WN + BN * WM) * 8 : 0);   // staging (+ the LayerNorm-fold strips)
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (!EXT && (p.ln_stats || p.c_f16 == 2 || p.res_hi)) return INK_ERR_ARG;
  static bool attr = ((void)hipFuncSetAttribute((const void*)gemm_f16_nt<BM, BN, BK, WM, WN, NS, ABL, EXT>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)attr;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_f16_nt<BM, BN, BK, WM, WN, NS, ABL, EXT>), dim3(ntm * ntn), dim3(WM * WN * 64), lds, s, p, group_m);
  return ink_launch_status();
}

}  // namespace

static int g_variant = -1;
// variant 55 (persistent workgroups) for the f16-output projections: 3-4 % faster per launch and 0.6 ms per encoder pass,
// but the overlapped step is 1.2 ms SLOWER with it - a persistent grid holds every CU for the whole launch, and the
// detector stream's kernels can no longer slip onto CUs between tiles (tools/stage_times.py A/B, profiles/r03_gemm_persist_ab.txt)
constexpr bool PERSIST_DEFAULT = false;
constexpr bool RMF_DEFAULT = false;   // variant 54 (residual through the MFMA pipe) for the eligible in-place f32 projections
// shape heuristic (tools/gemm_sweep.py on MI355X): the ping-pong 256x320 tile when N is a multiple of 320 and the
// launch is at least ~1.5 rounds of 256 CUs (SAM ViT-H: N = 1280 / 3840 / 5120, where batch 8 gives exact round
// counts and 10 % fewer staged bytes per flop than 256x256); else the 16-wave 256x256 tile whenever it fills the
// chip (>= ~200 tiles) and N does not waste a large part of a 256-wide tile; else the 128x128 tile.
extern "C" int ink_gemm_query_variant(int32_t M, int32_t N, int32_t K) {
  if (K % 64 != 0) return 32;      // 128x128x32 tile
  if (N % 320 == 0 && K >= 128 && (long)((M + 255) / 256) * (N / 320) >= 384) return 45;
  const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  const bool n_fits = (N % 256 == 0) || N >= 1024;
  return (tiles256 >= 200 && n_fits) ? 10 : 0;
}
// columns per statistics chunk (= wave-tile width) of the kernel ink_gemm_f16 picks for this shape: what
// InkGemm.stats_parts = N / chunk has to be computed from
extern "C" int ink_gemm_query_stats_chunk(int32_t M, int32_t N, int32_t K) {
  return (K % 64 == 0 && ink_gemm_query_variant(M, N, K) == 45) ? 80 : 64;
}
extern "C" int ink_abi_version(void) { return INK_ABI_VERSION; }
extern "C" int ink_gemm_set_variant(int32_t v) {
  // A process-wide override for the sweep / debugging tools (not thread-safe, not used by the product path).
  // v = gm * 100 + variant; + 10000: the ping-pong kernel takes its run-time-dispatch epilogue (A/B of the
  // compile-time activation modes, tools/gemm_res_ab.py)
  const int vv = v >= 10000 ? v - 10000 : v;
  const int base = vv >= 100 ? vv % 100 : vv;
  bool ok = v == -2 || base == -1 || base == 0 || base == 10 || base == 11 || base == 12 || base == 14 || base == 16 ||
            base == 32 || base == 40 || base == 42 || base == 45 || base == 47 || base == 53 || base == 54 || base == 55;
#ifdef INK_ABLATION
  ok = ok || (base >= 21 && base <= 23) || base == 43 || base == 44 || base == 46 || (base >= 48 && base <= 52) ||
       (base >= 61 && base <= 63);
#endif
  INK_CHECK_ARG(ok);
  g_variant = v;
  return INK_OK;
}

extern "C" int ink_gemm_f16(const InkGemm* pp, void* stream) {
  INK_CHECK_ARG(pp != nullptr);
  const InkGemm& p = *pp;
  INK_CHECK_ARG(p.A && p.W && p.C);
  INK_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0);
  INK_CHECK_ARG(p.K % 32 == 0 && p.N % 4 == 0);
  INK_CHECK_ARG(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.lda >= p.K && p.ldw >= p.K);
  INK_CHECK_ARG(p.ldc % 4 == 0 && p.ldc >= p.N);
  INK_CHECK_ARG(!p.residual || (p.ldr % 4 == 0 && p.ldr >= p.N));
  INK_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0);
  INK_CHECK_ARG(((uintptr_t)p.C & 15) == 0);
  INK_CHECK_ARG(p.act >= 0 && p.act <= 2);
  INK_CHECK_ARG(p.c_f16 >= 0 && p.c_f16 <= 2 && (p.c_f16 != 2 || (p.C_lo && ((uintptr_t)p.C_lo & 15) == 0)));
#ifndef INK_ABLATION      // (the stamp builds pass their debug buffer through col_scale)
  INK_CHECK_ARG(!p.res_hi || !p.col_scale);
#endif
  INK_CHECK_ARG(!p.res_hi || (p.res_lo && !p.residual && p.ldr % 4 == 0 && p.ldr >= p.N && p.act == INK_ACT_NONE));
  INK_CHECK_ARG(!p.ln_stats || (p.ln_colsum && p.ln_parts > 0 && p.ln_parts <= 20 && p.ln_parts % 2 == 0 &&
                                p.ln_dim > 0 && !p.row_map && ((uintptr_t)p.ln_stats & 15) == 0));
  hipStream_t s = (hipStream_t)stream;
  int v = g_variant;           // -1 (default): shape heuristic.  No environment variable reaches this function.
  const bool one_tile_only = v == -2;      // -2: the heuristic without the persistent form (A/B of whole steps)
  if (one_tile_only) v = -1;
  int gm = 1;
  bool generic_epilogue = false;
  if (v >= 10000) { generic_epilogue = true; v -= 10000; }
  if (v >= 100) { gm = v / 100; v = v % 100; }
  const bool ext = p.ln_stats || p.c_f16 == 2 || p.res_hi;      // ABI-4 forms: ping-pong or 128x128 tiles only
  if (p.K % 64 != 0) return launch_gemm<128, 128, 32, 2, 2, 2, 0, true>(p, s);
  if (v < 0) {
    v = ink_gemm_query_variant(p.M, p.N, p.K);
    gm = 4;
    if (ext && v == 10) { v = 0; gm = 1; }
  }
  if (p.stats_out) {      // row statistics are per wave-tile chunk: the split output only, whole chunks only
    const bool pp3 = (v == 45 && !p.ln_stats && !p.col_scale && !p.row_map && p.res_hi && p.act == INK_ACT_NONE) || v == 63;
    const int chunk = pp3 ? 80 : 64;
    INK_CHECK_ARG(p.c_f16 == 2 && p.N % chunk == 0 && p.stats_parts == p.N / chunk && (g_variant < 0 || v == 63));
  }
  // Production variants: 0 / 32 (128x128 tiles, K step 64 / 32), 10 (16-wave 256x256), 45 (ping-pong 256x320).
  // 40/42/47/53/54/55/16/12/14/11 are alternative CORRECT tilings kept for tools/gemm_sweep.py (ink_gemm_set_variant).
  // The ablation / instrumentation kernels DESIGN.md's measurements come from (they skip MFMAs, loads or the
  // epilogue and return garbage) exist only in a library built with -DINK_ABLATION (`python -m inklayer_amd.build
  // --ablation`, tools/gemm_stamps.py); the shipped library rejects their numbers in ink_gemm_set_variant.
  const bool persist_ok = !ext && !generic_epilogue && !p.col_scale && !p.row_map && (int64_t)p.M * p.lda * 2 < ((int64_t)1 << 32) &&
                          (int64_t)p.N * p.ldw * 2 < ((int64_t)1 << 32) && p.N % 320 == 0 && p.K >= 128 && p.act != INK_ACT_RELU &&
                          (p.c_f16 == 1 ? !p.residual : (p.act == INK_ACT_NONE));
  const bool rmf_ok = !ext && !generic_epilogue && p.residual && !p.c_f16 && p.act == INK_ACT_NONE && !p.col_scale &&
                      !p.row_map && (p.K / 32) % 40 == 0 && (int64_t)p.M * p.ldr * 4 < ((int64_t)1 << 32);
  switch (v) {
    case 10: return launch_gemm<256, 256, 64, 4, 4, 2>(p, s, gm);       // 16 waves x (64x64), 2 x 64 KB stages
    case 45: {                                                          // ping-pong 256x320, ring of 4 (144 KB)
      if (ext) {       // the ViT-H block forms on the split-f16 stream: one specialised kernel each
        const bool simple = !p.col_scale && !p.row_map;
        if (p.ln_stats && p.c_f16 == 1 && simple && !p.residual && !p.res_hi && p.act == INK_ACT_NONE)
          return launch_gemm_pp<4, 5, 0, true, 1>(p, s, gm);
        if (p.ln_stats && p.c_f16 == 1 && simple && !p.residual && !p.res_hi && p.act == INK_ACT_GELU)
          return launch_gemm_pp<4, 5, 0, true, 2>(p, s, gm);
        if (!p.ln_stats && p.c_f16 == 2 && simple && p.res_hi && p.act == INK_ACT_NONE)
          return launch_gemm_pp<4, 5, 0, true, 3>(p, s, gm);
        return launch_gemm<128, 128, 64, 2, 2, 2, 0, true>(p, s);      // any other combination: the general tile
      }
      // f16 C (qkv, lin1 of the ViT-H blocks): the persistent form is 3-4 % faster at K = 1280 (tools/gemm_persist_ab.py;
      // with an f32 C + residual 1-2 % slower) - see PERSIST_DEFAULT for why it is not dispatched
      if (PERSIST_DEFAULT && g_variant < 0 && !one_tile_only && persist_ok && p.c_f16 == 1)
        return p.act == INK_ACT_GELU ? launch_gemm_pp_persist<5, 1>(p, s, gm) : launch_gemm_pp_persist<5, 0>(p, s, gm);
      if (RMF_DEFAULT && g_variant < 0 && rmf_ok) return launch_gemm_pp<4, 5, 0, true, 4>(p, s, gm);
      return launch_gemm_pp<4, 5>(p, s, generic_epilogue ? -gm : gm);
    }
    case 55: {                                                          // persistent 256x320 (plain epilogues only)
      INK_CHECK_ARG(persist_ok);
      if (p.c_f16) return p.act == INK_ACT_GELU ? launch_gemm_pp_persist<5, 1>(p, s, gm) : launch_gemm_pp_persist<5, 0>(p, s, gm);
      return launch_gemm_pp_persist<5, 2>(p, s, gm);
    }
    case 54:                                                            // 256x320, f32 residual through the MFMA pipe
      INK_CHECK_ARG(rmf_ok);
      return launch_gemm_pp<4, 5, 0, true, 4>(p, s, gm);
    case 40: return launch_gemm_pp<4>(p, s, gm);                        // ping-pong 256x256, ring of 4 (128 KB)
    case 42: return launch_gemm_pp<3>(p, s, gm);                        // ... ring of 3 (96 KB)
    case 47: return launch_gemm_pp<3, 5>(p, s, gm);                     // 256x320, ring of 3 (DMA 1 granule ahead)
    case 53: return launch_gemm_pp<4, 5, 0, false>(p, s, gm);           // 256x320, 8 A fragments live, 3 ahead
    case 16: return launch_gemm<128, 128, 64, 2, 2, 2>(p, s, gm);       // variant 0 with grouped tile order
    case 12: return launch_gemm<128, 256, 64, 2, 4, 2>(p, s);           // 8 waves x (64x64), 96 KB
    case 14: return launch_gemm<256, 128, 32, 4, 2, 2>(p, s, gm);       // 48 KB: 3 workgroups / CU
    case 11: return launch_gemm<256, 256, 32, 4, 4, 4>(p, s);           // 16 waves, counted-vmcnt ring of 4 x K32
#ifdef INK_ABLATION
    case 21: return launch_gemm<256, 256, 64, 4, 4, 2, 1>(p, s, gm);    // variant 10 ablations: no DMA after tile 1
    case 22: return launch_gemm<256, 256, 64, 4, 4, 2, 2>(p, s, gm);    // ... no MFMA
    case 23: return launch_gemm<256, 256, 64, 4, 4, 2, 3>(p, s, gm);    // ... no epilogue
    case 43: return launch_gemm_pp<4, 4, 1>(p, s, gm);                  // ping-pong ablations (see the kernel comment)
    case 44: return launch_gemm_pp<4, 4, 3>(p, s, gm);
    case 46: return launch_gemm_pp<4, 4, 7>(p, s, gm);
    case 48: return launch_gemm_pp<4, 4, 8>(p, s, gm);                  // per-workgroup timeline, 256x256
    case 49: return launch_gemm_pp<4, 5, 8>(p, s, gm);                  // ... 256x320
    case 50: return launch_gemm_pp<4, 5, 9>(p, s, gm);                  // ... without MFMA
    case 51: return launch_gemm_pp<4, 5, 11>(p, s, gm);                 // ... pure DMA stream
    case 52: return launch_gemm_pp<4, 5, 15>(p, s, gm);                 // ... pure DMA stream, all L2 hits
    case 61: return launch_gemm_pp<4, 5, 8, true, 1>(p, s, gm);         // per-workgroup timeline of the ABI-4 forms
    case 62: return launch_gemm_pp<4, 5, 8, true, 2>(p, s, gm);         //   (stamps leave through col_scale)
    case 63: return launch_gemm_pp<4, 5, 8, true, 3>(p, s, gm);
#endif
    default: return launch_gemm<128, 128, 64, 2, 2, 2, 0, true>(p, s);  // variant 0
  }
}

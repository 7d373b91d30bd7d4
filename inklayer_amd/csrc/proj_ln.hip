// Image-side tail of a layer of SAM's two-way transformer for gfx950, one kernel:
//     keys <- LayerNorm( keys + out_proj(attn_out) )        (+ the split-f16 operand of the next projection)
// = cross_attn_image_to_token's output projection, residual and norm4 (SA/modeling/transformer.py:175-182) on the per-box
// image tokens [n_boxes * 4096, 256].  As separate kernels (add_split of the attention output, the split-f16 GEMM with
// residual, layernorm_rows writing f32 and split rows) the 537-MB per-box tensors crossed HBM ten times; here the
// attention output (f32 [R, 128]) and the residual are read once and the normalised rows written once.
//
// Structure of ffn_fused.hip's pre-phase: 128 rows per workgroup, four waves (one per SIMD), a wave owns 32 rows x all 256
// output columns in 128 accumulator registers (mfma_f32_32x32x16_f16, swapped form).  The operand is built in registers
// as split-f16 segments [hi | lo*64 | hi/64] (K' = 384 = 24 k-steps) against the packed [W_hi | W_hi/64 | W_lo*64]
// weight: 192 one-KiB operand blocks = three 64-KiB chunks, one per segment, streamed by LDS-DMA through two buffers
// (straight-line: no chunk loop).  LayerNorm in the accumulators; rows leave through a wave-private LDS tile as whole
// 1-KiB f32 rows and 512-B f16 segments.
#include <type_traits>
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr;

constexpr int CI = 128, C = 256, BM = 128;
constexpr int BLK = 1024;
constexpr int NSEG = 3, KSEG = CI / 16;              // 3 segments of 8 k-steps
constexpr int NBLK = (C / 32) * KSEG;                // 64 blocks per chunk (= segment): block 8 nt + j = (nt, s = 8 seg + j)
constexpr int CHUNK = NBLK * BLK;                    // 64 KiB
constexpr int PAR_BYTES = 3 * C * 4;                 // bias, gamma, beta
constexpr int OROW = C * 4 + 16;
constexpr int PAR_OFF = 4 * 32 * OROW;               // the parameter strip sits behind the epilogue's staging tiles (133 120 B > 2 chunks)
constexpr int LDS_BYTES = PAR_OFF + PAR_BYTES;
constexpr int DEPTH = 4;
static_assert(PAR_OFF >= 2 * CHUNK && LDS_BYTES <= 160 * 1024, "LDS layout");

template <int I> using ic = std::integral_constant<int, I>;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ f32x16 mfma32(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// pinned LDS reads with counted waits tied to the value they release (see ffn_fused.hip)
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
template <int OFF>
__device__ __forceinline__ f32x4 lds_f4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wait_f4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// ws f16 [256, 384] (ops.split_weight layout) -> blob [3 segments][64 blocks][64 lanes][8]
__global__ __launch_bounds__(256) void proj_ln_pack_kernel(const f16* __restrict__ ws, f16* __restrict__ blob) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= NSEG * NBLK * 64) return;
  const int lane = idx & 63, q = (idx >> 6) % NBLK, seg = (idx >> 6) / NBLK;
  const int nt = q / KSEG, s = KSEG * seg + q % KSEG;
  *(f16x8*)(blob + (int64_t)idx * 8) = *(const f16x8*)(ws + (int64_t)(32 * nt + (lane & 31)) * (NSEG * CI) + 16 * s + 8 * (lane >> 5));
}

__global__ __launch_bounds__(256) void proj256_ln_kernel(const float* __restrict__ A, const f16* __restrict__ blob,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         const int32_t* __restrict__ res_batch_rows, int rows_per_batch,
                                                         const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                         float eps, int64_t R, float* __restrict__ out_f32,
                                                         f16* __restrict__ out_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & 31, hh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM + wave * 32;
  const int64_t row = min(m0 + l, R - 1);
  {
    float* sP = (float*)(smem + PAR_OFF);
    sP[tid] = bias[tid];
    sP[C + tid] = ln_g[tid];
    sP[2 * C + tid] = ln_b[tid];
  }
  // the lane's half of its attention-output row (channels 16 s + 8 hh + 0..7) as the three split-f16 segments
  f16x8 xf[NSEG * KSEG];
  {
    const float* ap = A + row * CI + 8 * hh;
#pragma unroll
    for (int s = 0; s < KSEG; ++s) {
      const f32x4 a0 = *(const f32x4*)(ap + 16 * s), a1 = *(const f32x4*)(ap + 16 * s + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? a0[e] : a1[e - 4];
        const f16 hi = (f16)v;
        xf[s][e] = hi;
        xf[KSEG + s][e] = (f16)((v - (float)hi) * 64.0f);
        xf[2 * KSEG + s][e] = (f16)((float)hi * 0.015625f);
      }
    }
  }
  // accumulators = residual row (a per-box gather of a shared tensor while the keys are still one copy per image)
  f32x16 y[C / 32];
  {
    int64_t rrow = row;
    if (res_batch_rows) {
      const int64_t b = row / rows_per_batch;
      rrow = (int64_t)res_batch_rows[b] + (row - b * rows_per_batch);
    }
    const float* rp = res + rrow * C + 4 * hh;
#pragma unroll
    for (int nt = 0; nt < C / 32; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 r4 = *(const f32x4*)(rp + 32 * nt + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[nt][4 * g + e] = r4[e];
      }
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on only the LDS-DMA is in flight

  auto stage = [&](int buf, int seg, int k) {             // piece k (0..15) of this wave: block 4 k + wave
    const int q = 4 * k + wave;
    __builtin_amdgcn_global_load_lds((gptr_t)((const char*)blob + (int64_t)seg * CHUNK + q * BLK + lane * 16),
                                     (lptr_t)(smem + buf * CHUNK + q * BLK), 16, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < NBLK / 4; ++k) stage(0, 0, k);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_ptr)smem + lane * 16;
  const uint32_t par0 = (uint32_t)(uintptr_t)(lds_char_ptr)smem + PAR_OFF + 16 * hh;

  static_for<0, NSEG>([&](auto cc) {
    constexpr int seg = decltype(cc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of the segment has landed
    __builtin_amdgcn_s_barrier();                        // everyone's has; everyone is done reading the previous one
    asm volatile("" ::: "memory");
    const uint32_t wa = lds0 + (seg & 1) * CHUNK;
    f16x8 fr[DEPTH + 1];
    auto read = [&](auto ii) {
      constexpr int i = decltype(ii)::value;
      fr[i % (DEPTH + 1)] = lds_frag<i * BLK>(wa);
    };
    static_for<0, DEPTH>(read);
    static_for<0, NBLK>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      if constexpr (i + DEPTH < NBLK) read(ic<i + DEPTH>{});
      wait_frag<(NBLK - 1 - i < DEPTH ? NBLK - 1 - i : DEPTH)>(fr[i % (DEPTH + 1)]);
      y[i / KSEG] = mfma32(fr[i % (DEPTH + 1)], xf[KSEG * seg + i % KSEG], y[i / KSEG]);
      if constexpr (seg + 1 < NSEG && i % 4 == 1) stage((seg + 1) & 1, seg + 1, i / 4);
    });
  });

  // ---- + bias, LayerNorm over the 256 columns of the lane's row (128 values here, 128 in lane ^ 32)
  static_for<0, C / 32>([&](auto ii) {
    constexpr int nt = decltype(ii)::value;
    constexpr int ob = (32 * nt) * 4;
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
  __syncthreads();                                       // all waves are done with the weight buffers
  char* ot = smem + wave * 32 * OROW;                    // wave-private tile: 32 rows of 256 f32
  static_for<0, C / 32 * 2>([&](auto ii) {
    constexpr int nt = decltype(ii)::value / 2, g = 2 * (decltype(ii)::value % 2);
    constexpr int o0 = C * 4 + (32 * nt + 8 * g) * 4;    // gamma; beta C floats later
    f32x4 g0 = lds_f4<o0>(par0), g1 = lds_f4<o0 + 32>(par0), b0 = lds_f4<C * 4 + o0>(par0), b1 = lds_f4<C * 4 + o0 + 32>(par0);
    wait_f4(g0, g1, b0, b1);
    f32x4 v0, v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = (y[nt][4 * g + e] - mean) * rstd * g0[e] + b0[e];
      v1[e] = (y[nt][4 * g + 4 + e] - mean) * rstd * g1[e] + b1[e];
    }
    *(f32x4*)(ot + l * OROW + (32 * nt + 8 * g + 4 * hh) * 4) = v0;
    *(f32x4*)(ot + l * OROW + (32 * nt + 8 * g + 8 + 4 * hh) * 4) = v1;
  });
  // rows leave as whole segments: one row per instruction (64 lanes x 4 columns)
#pragma unroll 2
  for (int r = 0; r < 32; ++r) {
    const int64_t m = m0 + r;
    if (m >= R) break;
    const f32x4 v = *(const f32x4*)(ot + r * OROW + lane * 16);
    if (out_f32) *(f32x4*)(out_f32 + m * C + lane * 4) = v;
    if (out_split) {
      f16x4 hi, lo, hs;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (f16)v[e];
        lo[e] = (f16)((v[e] - (float)hi[e]) * 64.0f);
        hs[e] = (f16)((float)hi[e] * 0.015625f);
      }
      f16* o = out_split + m * (3 * C) + lane * 4;
      *(f16x4*)o = hi;
      *(f16x4*)(o + C) = lo;
      *(f16x4*)(o + 2 * C) = hs;
    }
  }
}

}  // namespace

extern "C" int ink_proj256_ln_pack(const void* ws_f16, void* blob_f16, void* stream) {
  INK_CHECK_ARG(ws_f16 && blob_f16 && ((((uintptr_t)ws_f16 | (uintptr_t)blob_f16) & 15) == 0));
  hipLaunchKernelGGL(proj_ln_pack_kernel, dim3((NSEG * NBLK * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)ws_f16, (f16*)blob_f16);
  return ink_launch_status();
}

extern "C" int ink_proj256_ln(const float* a_f32, const void* blob_f16, const float* bias, const float* res_f32,
                              const int32_t* res_batch_rows, int32_t rows_per_batch, const float* ln_g, const float* ln_b,
                              float eps, int64_t rows, float* out_f32, void* out_split_f16, void* stream) {
  INK_CHECK_ARG(a_f32 && blob_f16 && bias && res_f32 && ln_g && ln_b && rows > 0 && (out_f32 || out_split_f16));
  INK_CHECK_ARG(!res_batch_rows || (rows_per_batch > 0 && rows % rows_per_batch == 0));
  INK_CHECK_ARG((((uintptr_t)a_f32 | (uintptr_t)blob_f16 | (uintptr_t)res_f32 | (uintptr_t)out_f32 |
                  (uintptr_t)out_split_f16) & 15) == 0);
  static bool attr = ((void)hipFuncSetAttribute((const void*)proj256_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                LDS_BYTES), true);
  (void)attr;
  const int64_t ntiles = (rows + BM - 1) / BM;
  INK_CHECK_ARG(ntiles < (int64_t)1 << 31);
  hipLaunchKernelGGL(proj256_ln_kernel, dim3((unsigned)ntiles), dim3(256), LDS_BYTES, (hipStream_t)stream, a_f32,
                     (const f16*)blob_f16, bias, res_f32, res_batch_rows, rows_per_batch, ln_g, ln_b, eps, rows, out_f32,
                     (f16*)out_split_f16);
  return ink_launch_status();
}

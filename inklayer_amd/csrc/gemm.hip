// Dense projection kernel for gfx950 (MI355X): C = epilogue(A[M,K] * W[N,K]^T).
//
// Structure (cdna_hip_programming.md §5, "minimum 2-phase" form of T3):
//   - 128x128 output tile per 256-thread workgroup (4 waves as 2(M) x 2(N),
//     each wave 64x64 = 4x4 MFMA 16x16x32 f16 tiles, f32 accumulate);
//   - operands staged HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), two LDS
//     stages; the DMA for K-tile t+1 is in flight while tile t feeds the MFMAs;
//   - LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is
//     applied to the per-lane SOURCE address and to the ds_read address
//     (rule 21): 16-B chunk c of row r lives at chunk c ^ swz(r);
//   - MFMA is issued "swapped" (W fragment as the A operand) so each lane ends
//     up with 4 CONSECUTIVE output columns of one row -> 16-B epilogue accesses;
//   - workgroup ids are remapped so each XCD (private 4 MiB L2) works on a
//     contiguous band of M-tiles and re-reads its A panel from L2.
// Epilogue (fused, f32): +bias, GELU/ReLU, *col_scale, +residual, optional row
// scatter (window-unpartition / crop / un-shift), f32 or f16 store.
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

template <int BK>
__global__ __launch_bounds__(256, 2) void gemm_f16_nt_128(InkGemm p) {
  constexpr int BM = 128, BN = 128;
  constexpr int CPR = BK / 8;          // 16-B chunks per tile row
  constexpr int ROWB = BK * 2;         // bytes per tile row
  constexpr int TILE = BM * ROWB;      // bytes per operand tile
  constexpr int STAGE = 2 * TILE;      // A tile + W tile
  constexpr int ITERS = (BM * CPR) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM;
  const int n0 = (id % ntn) * BN;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;

  // per-thread DMA source rows (constant over the K loop)
  const f16* srcA[ITERS];
  const f16* srcW[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int pch = it * 256 + tid;      // linear chunk index inside the tile image
    const int row = pch / CPR;
    const int lch = (pch % CPR) ^ Swz<BK>::f(row);  // logical chunk stored at this slot
    const int ra = min(m0 + row, p.M - 1);
    const int rw = min(n0 + row, p.N - 1);
    srcA[it] = A + (size_t)ra * p.lda + lch * 8;
    srcW[it] = W + (size_t)rw * p.ldw + lch * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      // wave-uniform LDS base; the DMA adds lane*16 itself
      char* la = base + (it * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[it] + kt * BK), (lptr_t)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(srcW[it] + kt * BK), (lptr_t)(la + TILE), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) within a tile, per k-step
  const int fr = lane & 15, fq = lane >> 4;
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + fr;
    const int rw = wn * 64 + i * 16 + fr;
    offA[i] = ra * ROWB;
    offW[i] = rw * ROWB;
  }
  const int swzA = Swz<BK>::f(wm * 64 + fr);  // swz depends on row & 15 only (tile bases are multiples of 16)
  const int swzW = Swz<BK>::f(wn * 64 + fr);

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    __syncthreads();  // tile kt has landed (vmcnt(0) + barrier); everyone is done reading buffer cur^1
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* bA = smem + cur * STAGE;
    const char* bW = bA + TILE;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      f16x8 a[4], w[4];
      const int lc = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *(const f16x8*)(bA + offA[i] + ((lc ^ swzA) << 4));
        w[i] = *(const f16x8*)(bW + offW[i] + ((lc ^ swzW) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds C[m = .. + fr][n = .. + 4*fq + 0..3] for each (i,j)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
    const int orow = p.row_map ? p.row_map[m] : m;
    if (orow < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) v += *(const f32x4*)(p.bias + n);
      if (p.act == INK_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == INK_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.col_scale) v *= *(const f32x4*)(p.col_scale + n);
      if (p.residual) v += *(const f32x4*)(p.residual + (size_t)orow * p.ldr + n);
      if (p.c_f16) {
        f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *(f16x4*)((f16*)p.C + (size_t)orow * p.ldc + n) = h;
      } else {
        *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + n) = v;
      }
    }
  }
}

}  // namespace

extern "C" int ink_abi_version(void) { return INK_ABI_VERSION; }

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
  const int ntm = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
  const dim3 grid(ntm * ntn), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (p.K % 64 == 0) {
    constexpr int lds = 2 * 2 * 128 * 64 * 2;
    static bool attr = ((void)hipFuncSetAttribute((const void*)gemm_f16_nt_128<64>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
    (void)attr;
    hipLaunchKernelGGL(gemm_f16_nt_128<64>, grid, block, lds, s, p);
  } else {
    constexpr int lds = 2 * 2 * 128 * 32 * 2;
    hipLaunchKernelGGL(gemm_f16_nt_128<32>, grid, block, lds, s, p);
  }
  return ink_launch_status();
}

// SAM ViT-H GLOBAL attention for gfx950 (4 of the 32 blocks; SA/modeling/image_encoder.py:206-237 with the decomposed
// relative-position bias of :325-361 on the 64x64 token grid).  head_dim 80, n_q % 256 == 0, n_k % 128 == 0, grid_w 64.
//
// Same structure as the window kernel (attention_win.hip; see there for what the compiler needs to be told and why):
// one workgroup = FOUR waves, one per SIMD with the whole 512-register file, = one 256-query tile of one (image,
// head); a wave owns TWO 32-query subtiles and runs them half a key tile out of phase in one instruction stream
//      PV_B(t-1) || softmax_A(t) first half  |  S_B(t) || softmax_A(t) second half  |  barrier  |
//      PV_A(t)   || softmax_B(t) first half  |  S_A(t+1) || softmax_B(t) second half
// so that one subtile's softmax (VALU, volatile asm pinned between sched_barriers) issues in the gaps of the other's
// MFMAs; K and V^T fragments are read from LDS once per wave and used for both subtiles.
//   * S^T = K Q^T (mfma_f32_32x32x16_f16): the lane (q = lane & 31) holds 16 keys of ITS query per 32-key half.
//     A 64-key tile is one key row of the grid, so rel_w[q, kw] is the accumulator INIT of every tile - read from an
//     LDS table straight into the S registers while the other subtile computes (held in 64 registers it pushed the
//     scores into AGPRs: 110 v_accvgpr_read per tile) - and rel_h[q, kh = t] one scalar per tile, fetched a tile ahead
//     and folded into the exponent: p = 2^(s c + (rel_h c - m)).
//   * V rows carry a ones-column: l = sum_k P comes out of the PV MFMA.  Deferred running max (rescale only when a
//     row's max grew by more than 2^12 since it was set).
//   * K/V tiles stream through TWO LDS buffers with ONE raw barrier per tile: tile t+3 is loaded into registers during
//     tile t (5 chunks per thread, straight-line, counted waits), handed to LDS during tile t+1 (into the buffer tile
//     t+1... see the loop), published by the barrier of tile t+2.
//   * O rows leave through LDS as 160-B row segments.
// The 16 query tiles of an (image, head) run on ONE XCD, adjacent in time (xcd_remap): K/V is fetched from HBM once.
// Measured (same-box A/B, tools/attn_time.py): 997 us per launch at B = 8 against 1058 us for the 2-waves-per-SIMD form
// (692 vs 650 TFLOP/s).  Stamps (tools/glob_stamps.py): a key tile takes ~4300 cycles, of which the MFMA + LDS + load
// skeleton alone takes 2050 (the softmax fillers compiled out) - the fillers (est. 1250-1750 cycles of VALU issue per
// tile, two v_exp per MFMA gap) ADD to the skeleton instead of hiding in it.  The guide's rule is one transcendental
// per gap; 64 exps per tile do not fit 44 gaps, so the kernel is bound by VALU/transcendental issue, not by the MFMA
// pipe (1408 cycles per tile).  Removing the next tile's global loads (timing experiment) gives -500 cycles.
// Further timing experiments (results wrong on purpose, stamps only): v_exp replaced by v_add -240 cycles per tile; no
// fma and no cvt -340; no max phase -300.  No single class of filler is the cost - each gap that carries ANY VALU work
// runs ~50 cycles longer than a bare MFMA gap; with none the tile takes 2050 (46 per MFMA, the LDS / load skeleton).
// I.e. in this instruction stream the wave's own VALU does not hide under its MFMAs the way the guide's single-wave
// measurements suggest; why (operand-port conflicts of 512-register waves? the volatile-asm pinning?) is open.
#include <type_traits>
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x4 tr_read(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  return __builtin_bit_cast(f16x4, v);
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma_at(float s, float c, float add) {
  float t;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(s), "v"(c), "v"(add));
  return t;
}
__device__ __forceinline__ float exp2_at(float t) {
  float d;
  asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(t));
  return d;
}
__device__ __forceinline__ uint32_t cvt_pk_at(float a, float b) {
  uint32_t d;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x16 mfma(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <int I> using ic = std::integral_constant<int, I>;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int HD = 80, NT = 256, NQK = 5, NB = 3, CH = 10;
constexpr int KROW = 176, VROW = 192;                  // K rows: k | pad (odd chunk count); V rows: v | ones-column | pad
constexpr int TILE = 64 * KROW + 64 * VROW;            // one 64-key tile
constexpr int RWROW = 272;                             // rel_w rows of the tile's 256 queries: 64 f32 + pad (68 words:
                                                       // a half-wave's b128 reads of one column block hit all banks)
constexpr int LDS_BYTES = 2 * TILE + 256 * RWROW;
constexpr int OROW = 176;                              // output staging rows (reuses the tile buffers)
static_assert(4 * 64 * OROW <= 2 * TILE, "the output tiles fit the K/V buffers");
constexpr float NEG = -1e30f;
constexpr float THR = 12.0f;

#ifdef INK_ABLATION
// measurement build only (tools/glob_stamps.py): s_memtime stamps of workgroup 0, [wave][tile 8..23][8]
__device__ unsigned long long g_glob_stamps[4 * 16 * 8];
#define GSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (blockIdx.x == 0 && lane == 0 && t >= 8 && t < 24)                                           \
      g_glob_stamps[(wave * 16 + (t - 8)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                \
  } while (0)
#else
#define GSTAMP(i)
#endif

struct Sub {
  f16x8 qf[NQK];     // Q^T fragments (B operand)
  f32x16 s0, s1;
  uint32_t pw[16];
  f32x16 o[NB];
  float m, nm;       // running (deferred) max in log2 units; nm = rel_h c - m of the current tile
  float mx, mx2, ea, eb, fa, fb;
  __device__ __forceinline__ f16x8 pfrag(int ks) const {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f16x8, (u32x4){pw[4 * ks], pw[4 * ks + 1], pw[4 * ks + 2], pw[4 * ks + 3]});
  }
};

// j-th MFMA of an S unit (10): k-step j >> 1 on the lower (even j) / upper (odd j) 32 keys
__device__ __forceinline__ void s_mfma(int j, const f16x8 (&kfa)[NQK], const f16x8 (&kfb)[NQK], Sub& u) {
  const int s = j >> 1;
  if (j & 1) u.s1 = mfma(kfb[s], u.qf[s], u.s1);     // (s0 / s1 were preloaded with rel_w[q, kw])
  else u.s0 = mfma(kfa[s], u.qf[s], u.s0);
}
__device__ __forceinline__ void pv_mfma(int j, const f16x8 (&vf)[12], Sub& u) {
  u.o[j % 3] = mfma(vf[j], u.pfrag(j / 3), u.o[j % 3]);
}
__device__ __forceinline__ void sm_max(int g, Sub& u) {
  if (g == 0) {
    u.mx = max3(u.s0[0], u.s0[1], u.s0[2]);
    u.mx2 = max3(u.s0[3], u.s0[4], u.s0[5]);
    u.mx = max3(u.mx, u.s0[6], u.s0[7]);
    u.mx2 = max3(u.mx2, u.s0[8], u.s0[9]);
    u.mx = max3(u.mx, u.s0[10], u.s0[11]);
    u.mx2 = max3(u.mx2, u.s0[12], u.s0[13]);
  } else if (g == 1) {
    u.mx = max3(u.mx, u.s0[14], u.s0[15]);
    u.mx2 = max3(u.mx2, u.s1[0], u.s1[1]);
    u.mx = max3(u.mx, u.s1[2], u.s1[3]);
    u.mx2 = max3(u.mx2, u.s1[4], u.s1[5]);
    u.mx = max3(u.mx, u.s1[6], u.s1[7]);
  } else {
    u.mx2 = max3(u.mx2, u.s1[8], u.s1[9]);
    u.mx = max3(u.mx, u.s1[10], u.s1[11]);
    u.mx2 = max3(u.mx2, u.s1[12], u.s1[13]);
    u.mx = max3(u.mx, u.s1[14], u.s1[15]);
    u.mx = fmaxf(u.mx, u.mx2);
  }
}
// join the half-waves; adopt the max (first tile) or check the deferred-max threshold; set this tile's exponent offset
template <bool FIRST>
__device__ __forceinline__ void sm_decide(Sub& u, float c, float rh) {
  float x = u.mx, y = u.mx;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  const float mc = (fmaxf(x, y) + rh) * c;
  if constexpr (FIRST) {
    u.m = mc;
  } else {
    if (__any(mc - u.m > THR)) {          // rare: every earlier P.V of this subtile is complete at this point
      const float m_new = fmaxf(u.m, mc);
      const float alpha = __builtin_amdgcn_exp2f(u.m - m_new);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) u.o[i][r2] *= alpha;
      u.m = m_new;
    }
  }
  u.nm = rh * c - u.m;
}
__device__ __forceinline__ void sm_exp(int i, Sub& u, float c) {
  float pa = 0.f, pb = 0.f;
  if (i >= 2 && i < 18) { pa = u.ea; pb = u.eb; }
  if (i >= 1 && i < 17) {
    u.ea = exp2_at(u.fa);
    u.eb = exp2_at(u.fb);
  }
  if (i < 16) {
    float a, b;
    if (i < 8) { a = u.s0[2 * i]; b = u.s0[2 * i + 1]; }
    else { a = u.s1[2 * i - 16]; b = u.s1[2 * i - 15]; }
    u.fa = fma_at(a, c, u.nm);
    u.fb = fma_at(b, c, u.nm);
  }
  if (i >= 2 && i < 18) u.pw[i - 2] = cvt_pk_at(pa, pb);
}
// first half: 12 gap fillers (3 max + decide + exp slots 0..7); second half: 10 (slots 8..17)
template <bool FIRST>
__device__ __forceinline__ void sm_first(int g, Sub& u, float c, float rh) {
  if (g < 3) sm_max(g, u);
  else if (g == 3) sm_decide<FIRST>(u, c, rh);
  else sm_exp(g - 4, u, c);
}
__device__ __forceinline__ void sm_second(int g, Sub& u, float c) { sm_exp(8 + g, u, c); }

// RF16: rel_h / rel_w are f16 tables (ink_relpos_bias64_f16): converted when rel_w enters the LDS table / when the tile's
// rel_h scalar is fetched - half the table bytes written and read per launch
template <bool RF16>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void glob4_attn_kernel(InkAttn p) {
  using RT = std::conditional_t<RF16, f16, float>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // claims the SIMD's whole register file like the window kernel (attention_win.hip: waves of other kernels sharing a
  // SIMD with that kernel read corrupted registers; this one, same structure, showed nothing in tools/coresidency_matrix.py
  // at its 453 registers - the 56 spare ones are not worth the exposure)
  asm volatile("v_accvgpr_write_b32 a255, %0" ::"v"(0) : "a255");
  char* sRW = smem + 2 * TILE;                            // [256 queries][RWROW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hh = lane >> 5;
  const int nqb = p.n_q >> 8;
  const int blk = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bh = blk / nqb, qb = blk - bh * nqb;
  const int b = bh / p.n_heads, h = bh - b * p.n_heads;
  const int ntiles = p.n_k >> 6;
  const int64_t qrow0 = p.q_batch_rows ? (int64_t)p.q_batch_rows[b] : (int64_t)b * p.n_q;
  const int64_t kvb = p.kv_batch_rows ? (int64_t)p.kv_batch_rows[b] : (int64_t)b * p.n_k;
  const int qlA = wave * 32 + lq, qlB = 128 + wave * 32 + lq;     // query index inside the 256-query tile

  Sub A, B;
  auto load_q = [&](Sub& u, int ql) {
    const int64_t q = (int64_t)qb * 256 + ql;
    const f16* Qrow = (const f16*)p.Q + (qrow0 + q) * p.ldq + h * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < NQK; ++s) u.qf[s] = *(const f16x8*)(Qrow + 16 * s);
  };
  load_q(A, qlA);
  load_q(B, qlB);
  // rel_w rows of the tile's 256 queries -> LDS
  {
    const RT* RW = (const RT*)p.rel_w + ((int64_t)bh * p.n_q + (int64_t)qb * 256) * 64;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int ci = it * NT + tid, q = ci >> 4, c4 = ci & 15;
      if constexpr (RF16) {
        const f16x4 v = *(const f16x4*)(RW + q * 64 + c4 * 4);
        *(f32x4*)(sRW + q * RWROW + c4 * 16) = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      } else {
        *(f32x4*)(sRW + q * RWROW + c4 * 16) = *(const f32x4*)(RW + q * 64 + c4 * 4);
      }
    }
  }
  const RT* RHA = (const RT*)p.rel_h + ((int64_t)bh * p.n_q + (int64_t)qb * 256 + qlA) * 64;
  const RT* RHB = (const RT*)p.rel_h + ((int64_t)bh * p.n_q + (int64_t)qb * 256 + qlB) * 64;
  // V pad columns of both tile buffers (ones at d = 80 / 84, then zeros), written once
  for (int i = tid; i < 2 * 64 * 2; i += NT) {
    const int buf = i >> 7, row = (i >> 1) & 63, pc = i & 1;
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pc == 0) { z[0] = (f16)1; z[4] = (f16)1; }
    *(f16x8*)(smem + buf * TILE + 64 * KROW + row * VROW + (CH + pc) * 16) = z;
  }

  // ---- K/V staging: chunk ci = it * 256 + tid of a tile's 1280 (640 K, then 640 V); 5 per thread ----
  const char* Kg = (const char*)p.K + (kvb * p.ldk + h * HD) * 2;
  const char* Vg = (const char*)p.V + (kvb * p.ldv + h * HD) * 2;
  uint32_t goff[5];      // byte offset of the chunk inside its tile (relative to Kg / Vg + tile * 64 * ld * 2)
  int loff[5];           // LDS byte offset inside a tile buffer
  bool isv[5];
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const int ci = it * NT + tid;
    const bool v = ci >= 640;
    const int cj = v ? ci - 640 : ci, row = cj / CH, cc = cj - row * CH;
    isv[it] = v;
    goff[it] = (uint32_t)(row * (v ? p.ldv : p.ldk) * 2 + cc * 16);
    loff[it] = v ? 64 * KROW + row * VROW + cc * 16 : row * KROW + cc * 16;
  }
  f16x8 stg[2][5];
  // the five chunk addresses walk from tile to tile by a stride instead of being recomputed (the address arithmetic
  // of a load sits in an MFMA gap like everything else; measured neutral against recomputing them, same-box A/B)
  const char* gp[5];
#pragma unroll
  for (int it = 0; it < 5; ++it) gp[it] = (isv[it] ? Vg : Kg) + goff[it];
  const int64_t kstride = (int64_t)64 * p.ldk * 2, vstride = (int64_t)64 * p.ldv * 2;
  auto g_load = [&](int tile, auto SET, auto IT) {           // tiles are requested in order 0, 1, 2, 3, ...
    constexpr int set = decltype(SET)::value, it = decltype(IT)::value;
    stg[set][it] = *(const f16x8*)gp[it];
    const int64_t step = tile + 1 < ntiles ? (isv[it] ? vstride : kstride) : 0;     // (the last tile is re-read)
    gp[it] += step;
  };
  auto l_write = [&](int buf, auto SET, auto IT) {
    constexpr int set = decltype(SET)::value, it = decltype(IT)::value;
    *(f16x8*)(smem + buf * TILE + loff[it]) = stg[set][it];
  };
  // prologue: tiles 0 and 1 -> LDS, tile 2 -> register set 0
  static_for<0, 5>([&](auto it) { g_load(0, ic<0>{}, it); });
  static_for<0, 5>([&](auto it) { g_load(1, ic<1>{}, it); });
  static_for<0, 5>([&](auto it) { l_write(0, ic<0>{}, it); });
  static_for<0, 5>([&](auto it) { l_write(1, ic<1>{}, it); });
  static_for<0, 5>([&](auto it) { g_load(2, ic<0>{}, it); });
  __syncthreads();

  const float c = p.scale * 1.44269504088896340736f;
  const int koff0 = lq * KROW + hh * 16;
  const int koff1 = (32 + lq) * KROW + hh * 16;
  const int voff = 64 * KROW + (4 * hh + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) A.o[i][r] = B.o[i][r] = 0.f;
  A.m = B.m = NEG;
  A.nm = B.nm = 0.f;
  f16x8 kfa[NQK], kfb[NQK], vf[12];
  auto read_k = [&](int buf, int s) {
    kfa[s] = *(const f16x8*)(smem + buf * TILE + koff0 + s * 32);
    kfb[s] = *(const f16x8*)(smem + buf * TILE + koff1 + s * 32);
  };
  auto read_v = [&](int buf, int j) {
    const char* base = smem + buf * TILE + voff + (16 * (j / 3)) * VROW + (j % 3) * 64;
    const f16x4 a0 = tr_read(base);
    const f16x4 a1 = tr_read(base + 8 * VROW);
    vf[j] = (f16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  };
#define GAP() __builtin_amdgcn_sched_barrier(0)

  // rel_w[q, kw] of the lane's 2 x 16 key slots -> the S accumulators (group g of 4: kw = 8 g + 4 hh .. + 3, + 32)
  auto read_rw = [&](Sub& u, int ql, int g) {
    const f32x4 a = *(const f32x4*)(sRW + ql * RWROW + (8 * g + 4 * hh) * 4);
    const f32x4 c2 = *(const f32x4*)(sRW + ql * RWROW + (32 + 8 * g + 4 * hh) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { u.s0[4 * g + r] = a[r]; u.s1[4 * g + r] = c2[r]; }
  };
  float rhA = (float)RHA[0], rhB = (float)RHB[0];             // rel_h of tile 0
#pragma unroll
  for (int g = 0; g < 4; ++g) { read_rw(A, qlA, g); read_rw(B, qlB, g); }
#pragma unroll
  for (int s = 0; s < NQK; ++s) read_k(0, s);
#pragma unroll
  for (int j = 0; j < 10; ++j) s_mfma(j, kfa, kfb, A);
  GAP();

  // one key tile.  PAR = t & 1: tile t lives in buffer PAR, tile t+1 in 1 - PAR; register set PAR holds tile t+2 (loaded
  // one iteration ago, handed to buffer PAR after this tile's barrier), set 1 - PAR receives tile t+3.
  auto tile_iter = [&](int t, auto PAR_, auto FIRST_) {
    constexpr int PAR = decltype(PAR_)::value;
    constexpr bool FIRST = decltype(FIRST_)::value;
    float rhA1 = 0.f, rhB1 = 0.f;
    GSTAMP(0);
    if constexpr (FIRST) {
      // pipeline fill: S_B(0) || softmax_A(0) first half, then its second half alone
      static_for<0, 10>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        s_mfma(g, kfa, kfb, B);
        sm_first<true>(g, A, c, rhA);
        read_v(PAR, g);
        if constexpr (g == 9) { read_v(PAR, 10); read_v(PAR, 11); }
        if constexpr (g % 4 == 1) g_load(t + 3, ic<1 - PAR>{}, ic<g / 4>{});
        GAP();
      });
      static_for<10, 12>([&](auto g_) { sm_first<true>(decltype(g_)::value, A, c, rhA); GAP(); });
      static_for<0, 10>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        sm_second(g, A, c);
        if constexpr (g == 2) g_load(t + 3, ic<1 - PAR>{}, ic<3>{});
        if constexpr (g == 6) g_load(t + 3, ic<1 - PAR>{}, ic<4>{});
        GAP();
      });
    } else {
      // PV_B(t-1)  ||  softmax_A(t), first half; loads 0..2 of tile t+3
      static_for<0, 12>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        pv_mfma(g, vf, B);
        sm_first<false>(g, A, c, rhA);
        if constexpr (g % 4 == 2) g_load(t + 3, ic<1 - PAR>{}, ic<g / 4>{});
        if constexpr (g >= 8) read_rw(B, qlB, g - 8);             // (B's scores of tile t-1 are consumed: unit 4)
        GAP();
      });
      GSTAMP(1);
      // S_B(t)  ||  softmax_A(t), second half; V^T fragments of tile t; loads 3, 4
      static_for<0, 10>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        s_mfma(g, kfa, kfb, B);
        sm_second(g, A, c);
        read_v(PAR, g);
        if constexpr (g == 9) { read_v(PAR, 10); read_v(PAR, 11); }
        if constexpr (g == 3) g_load(t + 3, ic<1 - PAR>{}, ic<3>{});
        if constexpr (g == 7) g_load(t + 3, ic<1 - PAR>{}, ic<4>{});
        GAP();
      });
    }
    // every wave has read tile t out of buffer PAR (K fragments one tile ago, V^T just now), and tile t+1 - written to
    // buffer 1 - PAR during the previous tile - is complete
    GSTAMP(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GSTAMP(3);
    // PV_A(t)  ||  softmax_B(t), first half; K fragments of tile t+1; tile t+2: registers -> buffer PAR
    static_for<0, 12>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      pv_mfma(g, vf, A);
      sm_first<FIRST>(g, B, c, rhB);
      if constexpr (g < NQK) read_k(1 - PAR, g);
      if constexpr (g == 5) { rhA1 = (float)RHA[(t + 1) & 63]; rhB1 = (float)RHB[(t + 1) & 63]; }
      if constexpr (g >= 6 && g % 2 == 0) l_write(PAR, ic<PAR>{}, ic<(g - 6) / 2>{});
      if constexpr (g >= 8) read_rw(A, qlA, g - 8);               // (A's scores of tile t are consumed: unit 2)
      GAP();
    });
    GSTAMP(4);
    // S_A(t+1)  ||  softmax_B(t), second half
    static_for<0, 10>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      s_mfma(g, kfa, kfb, A);
      sm_second(g, B, c);
      if constexpr (g == 2) l_write(PAR, ic<PAR>{}, ic<3>{});
      if constexpr (g == 6) l_write(PAR, ic<PAR>{}, ic<4>{});
      GAP();
    });
    rhA = rhA1;
    rhB = rhB1;
    GSTAMP(5);
  };

  tile_iter(0, ic<0>{}, std::true_type{});
  tile_iter(1, ic<1>{}, std::false_type{});
  for (int t = 2; t < ntiles; t += 2) {
    tile_iter(t, ic<0>{}, std::false_type{});
    tile_iter(t + 1, ic<1>{}, std::false_type{});
  }
  // drain: PV_B of the last tile
  static_for<0, 12>([&](auto g_) { pv_mfma(decltype(g_)::value, vf, B); });
#undef GAP

  // O^T -> LDS (the tile buffers are free once every wave is here) -> O as 160-B row segments
  __syncthreads();
  char* myO = smem + wave * 64 * OROW;
  auto stage_o = [&](const Sub& u, int row0) {
    const float inv = 1.0f / u.o[2][8];
    char* dst = myO + (row0 + lq) * OROW + 8 * hh;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (32 * i + 8 * g < HD) {
          const f16x4 v = {(f16)(u.o[i][4 * g] * inv), (f16)(u.o[i][4 * g + 1] * inv),
                           (f16)(u.o[i][4 * g + 2] * inv), (f16)(u.o[i][4 * g + 3] * inv)};
          *(f16x4*)(dst + (32 * i + 8 * g) * 2) = v;
        }
      }
  };
  stage_o(A, 0);
  stage_o(B, 32);
  // (wave-private tile: the same wave's LDS operations are in order)
  f16* Ob = (f16*)p.O + ((int64_t)b * p.n_q + (int64_t)qb * 256) * p.ldo + h * HD;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const int idx = j * 64 + lane, row = idx / CH, ch = idx - row * CH;
    const f16x8 v = *(const f16x8*)(myO + row * OROW + ch * 16);
    const int ql = row < 32 ? wave * 32 + row : 128 + wave * 32 + row - 32;
    *(f16x8*)(Ob + (int64_t)ql * p.ldo + ch * 8) = v;
  }
}

}  // namespace

// Launcher used by ink_flash_attn (attention.hip) for bias_mode 1.
__attribute__((visibility("hidden"))) int ink_glob4_attn_launch(const InkAttn& p, hipStream_t s) {
  static bool attr = ((void)hipFuncSetAttribute((const void*)glob4_attn_kernel<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                      (void)hipFuncSetAttribute((const void*)glob4_attn_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), true);
  (void)attr;
  const int grid = p.n_batch * p.n_heads * (p.n_q >> 8);
  if (p.rel_f16) {
    hipLaunchKernelGGL(glob4_attn_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, s, p);
  } else {
    hipLaunchKernelGGL(glob4_attn_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, s, p);
  }
  return ink_launch_status();
}

#ifdef INK_ABLATION
extern "C" int ink_glob4_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_glob_stamps), sizeof(g_glob_stamps)) == hipSuccess ? INK_OK : INK_ERR_LAUNCH;
}
#endif

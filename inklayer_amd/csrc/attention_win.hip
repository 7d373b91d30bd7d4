// SAM 14x14-window attention for gfx950 (28 of the 32 ViT-H blocks; SA/modeling/image_encoder.py:206-237 with the
// window partition of :243-289 folded in through tok_rows).  head_dim 80, 193 <= n_k <= 208 (S = 14: 196), n_q <= 256.
//
// One workgroup = FOUR waves, one per SIMD, each with the whole 512-register file, = one (window, head) block at a
// time, persistent over a contiguous range of blocks (consecutive heads of the same few windows).
//   * A wave owns TWO 32-query subtiles (A = subtile w, B = subtile w + 4) and runs them half a tile out of phase in
//     ONE instruction stream:  S_A(t+1) | PV_B(t-1)->(t) | S_B | PV_A | ... so that the softmax of one subtile (VALU:
//     max, 32 x exp2, convert) is issued in the gaps of the other subtile's MFMAs.  An MFMA 32x32x16 holds vector
//     issue for 8 of its 32 cycles; each gap takes ~24 cycles of VALU issue, and the order is pinned per gap
//     (sched_barrier): left to itself hipcc gathers the exps in front of the MFMAs and the phases serialise, which is
//     what the 2-waves-per-SIMD form of this kernel did (7.4-8.2 us per block for 2.4 us of MFMA).  K' and V fragments
//     are read from LDS once per wave and used for both subtiles.
//   * S^T = K' Q'^T with K' = [k | onehot(kh), onehot(kw)], Q' = [q | rel(q,.)/scale]: the decomposed rel-pos bias
//     rides in the MFMA (two extra 16-wide k-steps); V rows carry a ones-column, so l = sum_k P comes out of the PV
//     MFMA.  The lane (q = lane & 31) holds 16 keys of ITS query per 32-key half: max / exp are in-lane, one
//     v_permlane32_swap joins the halves.
//   * The running max is DEFERRED (cdna_hip_programming.md T13): tile 0 sets m, a later tile rescales only if some
//     row's max grew by more than 2^12; P is f16 (10-bit mantissa at any magnitude below 65504) and l, O accumulate in
//     f32, so nothing is lost; the rescale, when taken, happens with every earlier P.V of that subtile complete.
//   * While block i is computed the operands of block i+1 arrive, all issued by STRAIGHT-LINE code (block indices
//     clamped to the last block, padded keys by an address select, invalid output rows dropped by the buffer bounds
//     check instead of a branch), so every wait hipcc inserts is a counted vmcnt(N), and SPREAD over the MFMA gaps of
//     the block (`vm` below): issued back to back, 4 waves x 19 loads overflow the CU's vector-memory queue and
//     every wave stands still for 4000-6000 cycles.
//         K and V rows of block i+1 -> 72 registers (handed to LDS between the two barriers of block i+1); the token
//         rows of block i+2's window -> 1 register (-> the LDS row table, same place); after the last S MFMA the Q'
//         fragments of block i+1 -> the (now dead) fragment registers; the O rows of block i-1 leave from a
//         wave-private LDS tile as 160-B row segments (a row-per-lane store touches 64 cache lines per instruction).
//     Tried and dropped: LDS-DMA for V (with a DMA in flight hipcc turns every vmcnt wait it inserts into
//     vmcnt(0)); Q' through the staging tile as coalesced row segments (needs 7 pieces = 28 registers in flight to
//     cover ~3000-4000 cycles of loaded memory latency; at that pressure hipcc's AGPR-copy rewrite pass crashes,
//     with 4 rotating registers every piece waits out its latency: 297 us).
//   * staging map: thread t < 250 owns chunk t % 10 of keys t / 10 + 25 it (it < 9): its LDS and table addresses
//     differ by constants.
// Measured (tools/attn_time.py, tools/win_stamps.py): 122 us per launch at B = 8 (336 MB of q,k,v,o: 2.75 TB/s = 34 % of
// 8 TB/s) against 172 us for the 2-waves-per-SIMD form; per block ~17 000 cycles of which 3 x 2450 are the tiles
// (52 MFMAs each: 47 cycles per gap, the gap's VALU + LDS + MFMA issue, not the 32 of the MFMA pipe).
#include <type_traits>
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x4 tr_read(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  return __builtin_bit_cast(f16x4, v);
}
// The softmax arithmetic is issued through VOLATILE asm statements: they keep their place between the sched_barriers of
// the MFMA gaps.  As plain (pure) operations hipcc's instruction selection hoists them - e.g. all of a tile's second-half
// exps into the gap that computes the row max - before the machine scheduler ever sees the barriers.
__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma_at(float s, float c, float neg_m) {       // s c - m, pinned
  float t;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(s), "v"(c), "v"(neg_m));
  return t;
}
__device__ __forceinline__ float exp2_at(float t) {                             // 2^t, pinned
  float d;
  asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(t));
  return d;
}
__device__ __forceinline__ uint32_t cvt_pk_at(float a, float b) {              // two f32 -> packed f16 (RNE), pinned
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

constexpr int HD = 80, NT = 256, NQKB = 5, NQK = 7, NB = 3, CH = 10;
constexpr int KROW = 240, VROW = 192, ROWS = 232;     // K' rows: k | one-hot(kh, kw) | pad; V rows: v | ones-column | pad
constexpr int SKEYS = 25, SIT = 9, STHR = SKEYS * CH;
constexpr int XROW = 176;                             // wave-private staging tile: 64 O rows of 160 B
#ifndef INK_EXP_LDS_SLACK
#define INK_EXP_LDS_SLACK 0          // (experiment, tools/race_variants.sh: unused bytes requested after the kernel's own)
#endif
constexpr int LDS_BYTES = ROWS * KROW + ROWS * VROW + 256 * 4 + 4 * 64 * XROW + 4 * 64 * 4 + INK_EXP_LDS_SLACK;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
#ifndef INK_WIN_DROP
#define INK_WIN_DROP 0x80000000u       // offset of a store the buffer bounds check drops (= the descriptor's num_records)
#endif
constexpr float NEG = -1e30f;
constexpr float THR = 12.0f;
// schedule knobs of `vm` (tools/win_variants.sh builds alternatives for same-box A/B runs)
#ifndef INK_WIN_KV_STEP
#define INK_WIN_KV_STEP 4
#endif
#ifndef INK_WIN_ST_STEP
#define INK_WIN_ST_STEP 3
#endif
#ifndef INK_WIN_QA_EARLY
#define INK_WIN_QA_EARLY 0
#endif          // deferred max: rescale when a row's max grew by more than 2^THR

#ifdef INK_ABLATION
// measurement build only (tools/win_stamps.py): s_memtime stamps of workgroup 0's waves, [wave][block][16]
__device__ unsigned long long g_win_stamps[4 * 16 * 16];
#define STAMP(i)                                                                              \
  do {                                                                                        \
    if (blockIdx.x == 0 && lane == 0 && nstamp < 16)                                          \
      g_win_stamps[(wave * 16 + nstamp) * 16 + (i)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define STAMP(i)
#endif

// One 32-query subtile's softmax state and operands.
struct Sub {
  f16x8 qf[NQK];     // Q'^T fragments (B operand): lane (col = q, half hh) holds Q'[q][16 s + 8 hh + j]
  f32x16 s0, s1;     // S^T of the current tile: keys 0..31 / 32..63 (16 of them per lane)
  uint32_t pw[16];   // P^T as the PV B operand, packed f16 pairs: 16 keys (4 words) per 16-key step
  f32x16 o[NB];      // O^T accumulators (d = 0..95; d = 80 / 84 = l)
  float m, nm;       // running (deferred) max, log2 units, and its negative
  bool ok;           // this lane's query exists (not window padding): only such lanes vote for a rescale
  float mx, mx2;     // scratch of the max phase (two independent chains)
  float ea, eb, fa, fb;   // exp pipeline: the fma results of step k and the exps of step k - 1
  __device__ __forceinline__ f16x8 pfrag(int ks) const {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f16x8, (u32x4){pw[4 * ks], pw[4 * ks + 1], pw[4 * ks + 2], pw[4 * ks + 3]});
  }
};

// j-th MFMA of a full-tile S unit (14): k-step j >> 1 on the lower (even j) / upper (odd j) 32 keys
__device__ __forceinline__ void s_mfma(int j, const f16x8 (&kfa)[NQK], const f16x8 (&kfb)[NQK], Sub& u) {
  const int s = j >> 1;
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j & 1) u.s1 = mfma(kfb[s], u.qf[s], s == 0 ? z : u.s1);
  else u.s0 = mfma(kfa[s], u.qf[s], s == 0 ? z : u.s0);
}
// k-step s of the tail tile's S unit (7): only the lower 32 keys exist
__device__ __forceinline__ void s_mfma_tail(int s, const f16x8 (&kfa)[NQK], Sub& u) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  u.s0 = mfma(kfa[s], u.qf[s], s == 0 ? z : u.s0);
}
// j-th MFMA of a full-tile PV unit (12): 16-key step j / 3, d-block j % 3
__device__ __forceinline__ void pv_mfma(int j, const f16x8 (&vf)[12], Sub& u) {
  u.o[j % 3] = mfma(vf[j], u.pfrag(j / 3), u.o[j % 3]);
}

// ---- softmax, cut into MFMA-gap sized steps --------------------------------------------------------------------
// max phase, gaps 0..2: 16 v_max3 over the 32 scores of the lane, as two independent chains
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
// join the half-waves, then either adopt the max (first tile) or check the deferred-max threshold
template <bool FIRST>
__device__ __forceinline__ void sm_decide(Sub& u, float c) {
  // x.hi <-> y.lo: afterwards the lane holds its own value in one of the two and its partner's (lane ^ 32) in the other.
  // (asm: through __builtin_amdgcn_permlane32_swap this hipcc folds max(r[0], r[1]) to r[0].)
  float x = u.mx, y = u.mx;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  const float mc = fmaxf(x, y) * c;
  if constexpr (FIRST) {
    u.m = mc;
    u.nm = -mc;
  } else {
    // rare: every earlier P.V of this subtile is complete at this point.  Padding lanes compute on a stand-in row
    // (row 0 of Q) and must not vote, or a window's rounding would depend on what else is in the batch.
    if (__any(u.ok && mc - u.m > THR)) {
      const float m_new = fmaxf(u.m, mc);
      const float alpha = __builtin_amdgcn_exp2f(u.m - m_new);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) u.o[i][r2] *= alpha;
      u.m = m_new;
      u.nm = -m_new;
    }
  }
}
// exp slot i (0..17) of a full tile, a three-stage pipeline over the 16 steps so that no instruction in an MFMA gap
// waits for its neighbour: fma of step i (scores 2i, 2i+1 of [s0 | s1]: s c - m), exp2 of step i - 1, f16 pair of
// step i - 2 -> word i - 2 of P
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
// first half of a tile's softmax as 12 gap fillers: 3 max + decide + exp slots 0..7; second half: slots 8..17
template <bool FIRST>
__device__ __forceinline__ void sm_first(int g, Sub& u, float c) {
  if (g < 3) sm_max(g, u);
  else if (g == 3) sm_decide<FIRST>(u, c);
  else sm_exp(g - 4, u, c);
}
__device__ __forceinline__ void sm_second(int g, Sub& u, float c) {
  if (g < 10) sm_exp(8 + g, u, c);
}
// tail tile (keys 192 .. 207 are all a window of n_k <= 208 can hold): 8 scores per lane, keys >= n_k masked
__device__ __forceinline__ void sm_tail(Sub& u, float c, int n_k, int hh) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int key = 192 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (key >= n_k) u.s0[r] = NEG;
  }
  u.mx = max3(u.s0[0], u.s0[1], u.s0[2]);
  u.mx = max3(u.mx, u.s0[3], u.s0[4]);
  u.mx = max3(u.mx, u.s0[5], u.s0[6]);
  u.mx = fmaxf(u.mx, u.s0[7]);
  sm_decide<false>(u, c);
  float f[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) f[r] = fma_at(u.s0[r], c, u.nm);
#pragma unroll
  for (int r = 0; r < 8; ++r) f[r] = exp2_at(f[r]);
#pragma unroll
  for (int k = 0; k < 4; ++k) u.pw[k] = cvt_pk_at(f[2 * k], f[2 * k + 1]);
}

template <bool TOK>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void win4_attn_kernel(InkAttn p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The kernel CLAIMS the SIMD's whole register file (512 = 256 VGPRs + 256 AGPRs; it uses 437): a wave of another
  // kernel must never share a SIMD with one of these.  Found in round 3 (tools/coresidency_matrix.py):
  // with this kernel on one stream and small kernels of another stream co-resident in the 72 registers it left over,
  // those kernels read zeros in one register of one quarter-wave (groupnorm_apply_kernel: gamma.z of lanes 48..63 in
  // 48 of 48 runs; the detector's boxes wrong in a third of back-to-back two-stream steps).  No store of this kernel
  // leaves its output (tools/win_canary.py), its own results never change, the buffer-store bounds trick and the LDS
  // size are not involved (variants built by tools/race_variants.sh); with the full claim: 0 of 80.
#ifndef INK_EXP_CLAIM_AGPR         // (experiment switch of tools/race_variants.sh: claim up to another AGPR; 180 = old behaviour)
#define INK_EXP_CLAIM_AGPR 255
#endif
#define INK_STR2(x) #x
#define INK_STR(x) INK_STR2(x)
  asm volatile("v_accvgpr_write_b32 a" INK_STR(INK_EXP_CLAIM_AGPR) ", %0" ::"v"(0) : "a" INK_STR(INK_EXP_CLAIM_AGPR));
  char* sK = smem;
  char* sV = smem + ROWS * KROW;
  int* sT = (int*)(smem + ROWS * KROW + ROWS * VROW);    // token rows of the window being fetched, [256]
  char* sX = smem + ROWS * KROW + ROWS * VROW + 256 * 4;   // per-wave staging tiles (wave-private: no barrier)
  uint32_t* sOoff = (uint32_t*)(sX + 4 * 64 * XROW);       // byte offset in O of each staged row, or 0x80000000

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hh = lane >> 5;
  const int total = p.n_batch * p.n_heads;
  int blk = (int)((int64_t)blockIdx.x * total / gridDim.x);
  const int blk_end = (int)((int64_t)(blockIdx.x + 1) * total / gridDim.x);
  const int last = blk_end - 1;
  const int qiA = wave * 32 + lq, qiB = (wave + 4) * 32 + lq;
  const int qcA = qiA < p.n_q ? qiA : p.n_q - 1, qcB = qiB < p.n_q ? qiB : p.n_q - 1;
  const int st = tid < STHR ? tid : STHR - 1;          // (the 6 spare threads repeat thread 249's chunks)
  const int key0 = st / CH, cc = st - key0 * CH;
  const int tkey = tid < p.n_k ? tid : p.n_k - 1;      // the table entry this thread fetches

  // row (in K/V) of key `tkey` of block blk_'s window, or -1 (window padding)
  auto fetch_row = [&](int blk_) -> int {
    const int b_ = blk_ / p.n_heads;
    if constexpr (TOK) {
      return p.tok_rows[(int64_t)b_ * p.n_k + tkey];
    } else {
      return (p.kv_batch_rows ? p.kv_batch_rows[b_] : b_ * p.n_k) + tkey;
    }
  };
  // row (in Q, and in O with tok_rows) of a query; reads the table of block blk_'s window
  auto query_row = [&](int blk_, int qc) -> int {
    if constexpr (TOK) {
      return sT[qc];                                    // (n_q == n_k with tok_rows)
    } else {
      const int b_ = blk_ / p.n_heads;
      return (p.q_batch_rows ? p.q_batch_rows[b_] : b_ * p.n_q) + qc;
    }
  };
  f16x8 ka[SIT], va[SIT];
  // K and V rows of block blk_ -> registers; reads the table of blk_'s window
  int krow[SIT];
  auto read_rows = [&]() {                              // all table reads at once: one LDS round trip, not nine
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int key = key0 + SKEYS * it;                // keys >= n_k: any finite row (they are masked / P = 0)
      krow[it] = sT[key < p.n_k ? key : p.n_k - 1];
    }
  };
  // load j of the 18 of a block: the K (even j) / V (odd j) chunk of key key0 + 25 (j / 2).  The loads are spread over
  // the MFMA gaps of the block: issued back to back (4 waves x 19) they overflow the CU's vector-memory queue and every
  // wave stands still for 4000-6000 cycles until its last load has been accepted.
  auto fetch_kv_one = [&](int blk_, int j) {
    const uint32_t hc = (uint32_t)((blk_ % p.n_heads) * HD + cc * 8) * 2u;     // byte offset inside a row
    const int it = j >> 1, r = krow[it];
    // padded key (r < 0): qkv(0) = the bias rows.  Both arms of the select are uniform pointers and the row offset is
    // computed for max(r, 0), so that hipcc emits v_cndmask, not a branch per load
    const char* base = (j & 1) ? (const char*)p.V : (const char*)p.K;
    if constexpr (TOK) base = r >= 0 ? base : ((j & 1) ? (const char*)p.pad_v : (const char*)p.pad_k);
    const uint32_t rp = (uint32_t)(r > 0 ? r : 0);
    const uint32_t ld2 = (uint32_t)(((j & 1) ? p.ldv : p.ldk) * 2);
    const f16x8 v = *(const f16x8*)(base + ((uint64_t)rp * ld2 + hc));
    if (j & 1) va[it] = v; else ka[it] = v;
  };
  auto fetch_kv = [&](int blk_) {
    read_rows();
#pragma unroll
    for (int j = 0; j < 2 * SIT; ++j) fetch_kv_one(blk_, j);
  };
  Sub A, B;
  auto load_q = [&](int blk_, int qrow_, int qc, Sub& u) {
    const int64_t row = (TOK && qrow_ < 0) ? 0 : qrow_;
    const f16* Qrow = (const f16*)p.Q + row * p.ldq + (blk_ % p.n_heads) * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < NQKB; ++s) u.qf[s] = *(const f16x8*)(Qrow + 16 * s);
    const f16* R = (const f16*)p.rel_aug + ((int64_t)blk_ * p.n_q + qc) * 32 + 8 * hh;
    u.qf[NQKB] = *(const f16x8*)R;
    u.qf[NQKB + 1] = *(const f16x8*)(R + 16);
  };

  // ---- the wave's staging tile X (64 rows: subtile A's 32 queries, then subtile B's): the O rows of block i are
  // staged at its end and leave during the first tile of block i+1 as 10 coalesced 1 KiB stores (a row-per-lane store
  // touches 64 cache lines per instruction: measured ~175 cycles of issue each, 20 of them per block)
  char* myX = sX + wave * 64 * XROW;
  uint32_t* myOoff = sOoff + wave * 64;
  auto x_query = [&](int row) { return row < 32 ? wave * 32 + row : (wave + 4) * 32 + row - 32; };
  int lane_x = lane;                                    // (made opaque per block: the piece addresses below are cheap
                                                        // to recompute and must not be hoisted into registers)
  // one fragment (k-step s2 of 7) of block blk_'s Q'^T for a subtile: row-per-lane, straight into the fragment register
  auto load_q_piece = [&](int blk_, int qrow_, int qc, Sub& u, int s2) {
    if (s2 < NQKB) {
      const int64_t row = (TOK && qrow_ < 0) ? 0 : qrow_;
      u.qf[s2] = *(const f16x8*)((const f16*)p.Q + row * p.ldq + (blk_ % p.n_heads) * HD + 8 * hh + 16 * s2);
    } else {
      u.qf[s2] = *(const f16x8*)((const f16*)p.rel_aug + ((int64_t)blk_ * p.n_q + qc) * 32 + 8 * hh + 16 * (s2 - NQKB));
    }
  };

  // constant parts of the LDS image, written once: the one-hot (kh, kw) columns of K' (+ its pad chunk), the pad
  // columns of V (ones at d = 80 for the lower half-wave, d = 84 for the upper) and zero data for the key slots that
  // are never staged
  for (int i = tid; i < ROWS * 2; i += NT) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((i & 1) == 0) { z[0] = (f16)1; z[4] = (f16)1; }
    *(f16x8*)(sV + (i >> 1) * VROW + (CH + (i & 1)) * 16) = z;
  }
  for (int i = tid; i < ROWS * 5; i += NT) {
    const int key = i / 5, c5 = i - key * 5;
    const int kh = key / p.grid_w, kw = key - kh * p.grid_w + p.grid_w;
    f16x8 e;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = c5 * 8 + j;
      e[j] = (c5 < 4 && key < p.n_k && (col == kh || col == kw)) ? (f16)1 : (f16)0;
    }
    *(f16x8*)(sK + key * KROW + (CH + c5) * 16) = e;
  }
  for (int i = tid; i < (ROWS - SKEYS * SIT) * CH; i += NT) {
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(sK + (SKEYS * SIT + i / CH) * KROW + (i % CH) * 16) = z;
    *(f16x8*)(sV + (SKEYS * SIT + i / CH) * VROW + (i % CH) * 16) = z;
  }
  // prologue: table <- window of the first block; its Q, K, V; then the row of the second block's window
  sT[tid] = fetch_row(blk);
  __syncthreads();
  int qrowA = query_row(blk, qcA), qrowB = query_row(blk, qcB);
  fetch_kv(blk);
  int rt = fetch_row(blk + 1 < blk_end ? blk + 1 : last);
  load_q(blk, qrowA, qcA, A);
  load_q(blk, qrowB, qcB, B);
  myOoff[lane] = INK_WIN_DROP;                          // (no output rows staged yet: the first block's stores are dropped)

  const float c = p.scale * 1.44269504088896340736f;
  const int koff0 = lq * KROW + hh * 16;
  const int koff1 = (32 + lq) * KROW + hh * 16;
  const int voff = (4 * hh + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  char* wK = sK + key0 * KROW + cc * 16;               // hand-off destinations of this thread
  char* wV = sV + key0 * VROW + cc * 16;
  // invalid output rows (window padding, q >= n_q) get an offset beyond the descriptor's range: the store is dropped
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.O, 0, 0x80000000u, 0x00020000);
  // (hipcc merges the memory-counter state of the loop entry with the back edge's and waits for the smaller count:
  // ten dropped stores here give the entry the block loop's own issue order - rows, Q', stores - so the waits at
  // the loop top count the stores of the previous block instead of draining them)
#ifndef INK_EXP_GLOBAL_STORES
#pragma unroll
  for (int i = 0; i < 10; ++i)
    __builtin_amdgcn_raw_buffer_store_b64((i32x2){0, 0}, orsrc, INK_WIN_DROP, 0, 0);
#endif

  int nstamp = 0;
  (void)nstamp;
  for (; blk < blk_end; ++blk) {
    const int b = blk / p.n_heads, h = blk - b * p.n_heads;
    STAMP(0);
    const bool okA = qiA < p.n_q && (!TOK || qrowA >= 0);    // window padding: nothing to compute, nothing to store
    const bool okB = qiB < p.n_q && (!TOK || qrowB >= 0);
    __syncthreads();                                          // every wave has finished reading block blk - 1
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      *(f16x8*)(wK + it * SKEYS * KROW) = ka[it];
      *(f16x8*)(wV + it * SKEYS * VROW) = va[it];
    }
    sT[tid] = rt;
    STAMP(1);
    __syncthreads();
    STAMP(2);
    const int n1 = blk + 1 < blk_end ? blk + 1 : last, n2 = blk + 2 < blk_end ? blk + 2 : last;
    const int qrow1A = query_row(n1, qcA), qrow1B = query_row(n1, qcB);
    read_rows();
    lane_x = lane;
    asm volatile("" : "+v"(lane_x));
    STAMP(3);

#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) A.o[i][r] = B.o[i][r] = 0.f;
    A.m = B.m = NEG;
    A.ok = okA;
    B.ok = okB;
    A.nm = B.nm = -NEG;
    const char* bV = sV;
    f16x8 kfa[NQK], kfb[NQK], vf[12], vt[NB];
    auto read_k = [&](int t, int s) {          // K' fragments (A operand) of k-step s of tile t
      kfa[s] = *(const f16x8*)(sK + t * 64 * KROW + koff0 + s * 32);
      kfb[s] = *(const f16x8*)(sK + t * 64 * KROW + koff1 + s * 32);
    };
    auto read_k_tail = [&](int s) { kfa[s] = *(const f16x8*)(sK + 192 * KROW + koff0 + s * 32); };
    auto read_v = [&](int t, int j) {          // V^T fragment j = (16-key step j / 3, d-block j % 3) of tile t
      const char* base = bV + t * 64 * VROW + voff + (16 * (j / 3)) * VROW + (j % 3) * 64;
      const f16x4 a0 = tr_read(base);
      const f16x4 a1 = tr_read(base + 8 * VROW);
      vf[j] = (f16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    auto read_v_tail = [&](int i) {            // keys 192..207, d-block i
      const char* base = bV + 192 * VROW + voff + i * 64;
      const f16x4 a0 = tr_read(base);
      const f16x4 a1 = tr_read(base + 8 * VROW);
      vt[i] = (f16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    f16x8 ov[2];
    uint32_t ooff[2];
    auto xo_read = [&](int j) {                // piece j (0..9) of the staged O rows of the previous block: LDS -> regs
      const int idx = j * 64 + lane_x, row = idx / CH, ch = idx - row * CH;
      ov[j & 1] = *(const f16x8*)(myX + row * XROW + ch * 16);
      const uint32_t off = myOoff[row];
      ooff[j & 1] = off == INK_WIN_DROP ? off : off + (uint32_t)(ch * 16);
    };
    auto xo_issue = [&](int j) {               // ... -> O (read one slot earlier: no LDS round trip inside a gap)
#ifdef INK_EXP_GLOBAL_STORES       // (experiment of tools/race_variants.sh: masked global stores instead of buffer stores)
      if (ooff[j & 1] < 0x80000000u) *(i32x4*)((char*)p.O + ooff[j & 1]) = __builtin_bit_cast(i32x4, ov[j & 1]);
#else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ov[j & 1]), orsrc, ooff[j & 1], 0, 0);
#endif
    };
    // The block's vector-memory instructions, spread over its 166 MFMA gaps (G = gap number in the block, a
    // compile-time constant after unrolling).  Back to back they overflow the CU's vector-memory queue (4 waves x 19
    // loads: every wave stood still for 4000-6000 cycles); a store whose operand is produced in the same gap stalls the
    // wave for the producer's latency, hence the one-slot-earlier reads.
    //   G = 2, 5 .. 29      O rows of the previous block leave (read from the staging tile one slot earlier)
    //   G = 31, 35 .. 99    the next block's 18 K / V chunks -> registers
    //   G = 128             the table row of block + 2
    //   G = 154 .. 165      (after the last S MFMA) its Q' fragments -> the fragment registers, the last two after the tail
    auto vm = [&](auto G_) {
      constexpr int G = decltype(G_)::value;
      if constexpr (G == 0) xo_read(0);
      constexpr int ST = INK_WIN_ST_STEP, KS = INK_WIN_KV_STEP, KV0 = 2 + ST * 9 + 2;
      static_assert(KV0 + KS * 17 < 140, "the K / V loads end before the table row load");
      if constexpr (G >= 2 && G <= 2 + ST * 9 && (G - 2) % ST == 0) {
        constexpr int i = (G - 2) / ST;
        xo_issue(i);
        if constexpr (i < 9) xo_read(i + 1);
      }
      if constexpr (G >= KV0 && G <= KV0 + KS * 17 && (G - KV0) % KS == 0) fetch_kv_one(n1, (G - KV0) / KS);
      if constexpr (G == 140) rt = fetch_row(n2);
      if constexpr (INK_WIN_QA_EARLY) {       // A's fragments are dead after its tail S unit (G 140..146)
        if constexpr (G >= 147 && G <= 153) load_q_piece(n1, qrow1A, qcA, A, G - 147);
        if constexpr (G >= 154 && G <= 160) load_q_piece(n1, qrow1B, qcB, B, G - 154);
      } else if constexpr (G >= 154 && G <= 165) {
        constexpr int k = G - 154;
        if constexpr (k < NQK) load_q_piece(n1, qrow1A, qcA, A, k); else load_q_piece(n1, qrow1B, qcB, B, k - NQK);
      }
    };
#define GAP() __builtin_amdgcn_sched_barrier(0)

    {
      // ---- both subtiles, half a tile out of phase ----
      // (static_for: the gap numbers must be constants when the arrays are scalarised - with plain unrolled loops the
      // K / V staging registers ended up in scratch memory)
#pragma unroll
      for (int s = 0; s < NQK; ++s) read_k(0, s);
#pragma unroll
      for (int j = 0; j < 14; ++j) s_mfma(j, kfa, kfb, A);
      GAP();
      STAMP(4);
      static_for<0, 3>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        if constexpr (t == 0) {
          // pipeline fill: S_B(0)  ||  softmax_A(0), first half; V fragments of tile 0; then the second half alone.
          // (Every volatile-asm VALU read of an MFMA result below sits at least two MFMAs after the MFMA that wrote it:
          // hipcc does not pad MFMA -> inline-asm hazards, and CDNA does not interlock them.)
          static_for<0, 14>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            s_mfma(g, kfa, kfb, B);
            if constexpr (g < 12) { sm_first<true>(g, A, c); read_v(0, g); }
            vm(ic<g>{});
            GAP();
          });
          static_for<0, 10>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            sm_second(g, A, c);
            vm(ic<14 + g>{});
            GAP();
          });
        } else {
          // PV_B(t-1)  ||  softmax_A(t), first half
          static_for<0, 12>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            pv_mfma(g, vf, B);
            sm_first<false>(g, A, c);
            vm(ic<t * 52 - 2 + g>{});
            GAP();
          });
          // S_B(t)  ||  softmax_A(t), second half; V fragments of tile t
          static_for<0, 14>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            s_mfma(g, kfa, kfb, B);
            sm_second(g, A, c);
            if constexpr (g < 12) read_v(t, g);
            vm(ic<t * 52 + 10 + g>{});
            GAP();
          });
        }
        // PV_A(t)  ||  softmax_B(t), first half; K' fragments of tile t + 1 (the tail tile after the third)
        static_for<0, 12>([&](auto g_) {
          constexpr int g = decltype(g_)::value;
          pv_mfma(g, vf, A);
          sm_first<t == 0>(g, B, c);
          if constexpr (g < NQK) {
            if constexpr (t < 2) read_k(t + 1, g); else read_k_tail(g);
          }
          vm(ic<t * 52 + 24 + g>{});
          GAP();
        });
        // S_A(t+1)  ||  softmax_B(t), second half   (after the third tile: both tail S units)
        static_for<0, 14>([&](auto g_) {
          constexpr int g = decltype(g_)::value;
          if constexpr (t < 2) s_mfma(g, kfa, kfb, A);
          else if constexpr (g < NQK) s_mfma_tail(g, kfa, A);
          else s_mfma_tail(g - NQK, kfa, B);
          sm_second(g, B, c);
          vm(ic<t * 52 + 36 + g>{});
          GAP();
        });
        STAMP(5 + t);
      });
    }
    STAMP(8);
    {
      // PV_B(2)  ||  tail softmax of A; tail V fragments
      static_for<0, 12>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        pv_mfma(g, vf, B);
        if constexpr (g == 0) sm_tail(A, c, p.n_k, hh);
        if constexpr (g >= 1 && g <= NB) read_v_tail(g - 1);
        vm(ic<154 + g>{});
        GAP();
      });
#pragma unroll
      for (int i = 0; i < NB; ++i) A.o[i] = mfma(vt[i], A.pfrag(0), A.o[i]);
      sm_tail(B, c, p.n_k, hh);
#pragma unroll
      for (int i = 0; i < NB; ++i) B.o[i] = mfma(vt[i], B.pfrag(0), B.o[i]);
    }
#undef GAP

    // O^T -> staging tile (after the Q' fragments have been read out of it): the lane holds 4 consecutive d of ITS
    // query per accumulator group; the rows leave during the next block's first tile.  O is dense per batch entry
    // unless tok_rows scatters it back to token order; invalid rows get an offset the buffer bounds check drops.
    auto stage_o = [&](const Sub& u, bool ok, int qrow, int qi, int row0) {
      const float inv = 1.0f / u.o[2][8];      // row d = 80 (hh = 0) / 84 (hh = 1) of O^T: sum_k P
      char* dst = myX + (row0 + lq) * XROW + 8 * hh;
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
      const uint32_t o_row = TOK ? (uint32_t)qrow : (uint32_t)(b * p.n_q + qi);
      if (hh == 0) myOoff[row0 + lq] = ok ? (o_row * (uint32_t)p.ldo + (uint32_t)(h * HD)) * 2u : INK_WIN_DROP;
    };
    STAMP(9);
    if constexpr (!INK_WIN_QA_EARLY) {
      load_q_piece(n1, qrow1B, qcB, B, 5);
      load_q_piece(n1, qrow1B, qcB, B, 6);
    }
    stage_o(A, okA, qrowA, qiA, 0);
    stage_o(B, okB, qrowB, qiB, 32);
    STAMP(10);
    ++nstamp;
    qrowA = qrow1A;
    qrowB = qrow1B;
  }
  // the last block's output rows
  typedef int i32x4b __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const int idx = j * 64 + lane, row = idx / CH, ch = idx - row * CH;
    const f16x8 v = *(const f16x8*)(myX + row * XROW + ch * 16);
    const uint32_t off = myOoff[row];
#ifdef INK_EXP_GLOBAL_STORES
    if (off != INK_WIN_DROP) *(i32x4b*)((char*)p.O + off + (uint32_t)(ch * 16)) = __builtin_bit_cast(i32x4b, v);
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4b, v), orsrc,
                                           off == INK_WIN_DROP ? off : off + (uint32_t)(ch * 16), 0, 0);
#endif
  }
}

}  // namespace

// Launcher used by ink_flash_attn (attention.hip) for bias_mode 2 at SAM's window size.
__attribute__((visibility("hidden"))) int ink_win4_attn_launch(const InkAttn& p, int n_cus, hipStream_t s) {
  static bool attr = ((void)hipFuncSetAttribute((const void*)win4_attn_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                      (void)hipFuncSetAttribute((const void*)win4_attn_kernel<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), true);
  (void)attr;
  const int bhn = p.n_batch * p.n_heads;
  const int grid = bhn < n_cus ? bhn : n_cus;            // persistent walk over (window, head) blocks
  if (p.tok_rows) {
    hipLaunchKernelGGL(win4_attn_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, s, p);
  } else {
    hipLaunchKernelGGL(win4_attn_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, s, p);
  }
  return ink_launch_status();
}

#ifdef INK_ABLATION
extern "C" int ink_win4_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_stamps), sizeof(g_win_stamps)) == hipSuccess ? INK_OK : INK_ERR_LAUNCH;
}
#endif

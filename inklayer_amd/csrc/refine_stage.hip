// Refinement stage of InkLayer on resident masks: depth ordering support, disjoint parsing, mask growth
// over unlabeled stroke pixels, per-pixel box assignment support, the "unlabeled" extra mask.  Integer / bit work on
// ROW-ALIGNED BIT PLANES (bitplane.h) and ONE uint8 LABEL IMAGE instead of n separate HxW masks: after the disjoint
// parsing every pixel belongs to at most one mask, so the n masks of the reference ARE a label image, and every step
// that loops over masks there is one stencil / histogram pass here.
//   reference: InkLayer/refinement/depth_sort.py:72-89 (get_mask_depth_score), :177-240 (compute_major_overlap_matrix),
//              InkLayer/refinement/refiner.py:21-33 (clean_delicate_mask), :35-88 (composite_and_parse_masks),
//              :91-126 (parse_masks_to_disjoint_masks), :129-196 (refine_masks_with_watershed), :228-297
//              (refine_masks_with_boxes: only its distance queries; the raster-order assignment itself is sequential
//              and runs on the host, ink_host_assign_unlabeled), :301-337 (create_unlabeled_mask),
//              InkLayer/refinement/utils.py:3-9 (sketch_to_01binary).
// What stays on the host and why: the greedy stroke thinning of sparse_sketch_sample (depth_sort.py:49-68) and the
// raster-order pixel assignment (every assignment changes the distances of the pixels after it) are sequential by
// construction; they are plain C++ below (ink_host_*), fed by / feeding the kernels.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "bitplane.h"
#include "../../include/inklayer_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// sketch -> four bit planes
//   0: sketch_to_01binary of the cv2 BGR image: blue <= max(all channels) / 2   (utils.py:3-9)
//   1: PIL luma <  250   (SKETCH_THRESHOLD tests `sketch_array < SKETCH_THRESHOLD`, refiner.py:108-110)
//   2: PIL luma <= 250   (`~(sketch_image > SKETCH_THRESHOLD)`, refiner.py:134,246)
//   3: cv2 gray <  250   (cv2.imread(IMREAD_GRAYSCALE) < SKETCH_THRESHOLD, refiner.py:302-303)
__global__ __launch_bounds__(256) void rs_max_kernel(const uint8_t* __restrict__ p, int64_t n, int* __restrict__ out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, (int)p[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void rs_sketch_planes_kernel(const uint8_t* __restrict__ rgb, int H, int W, int Wp,
                                                               const int* __restrict__ maxv, u64* __restrict__ planes) {
  const int word = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (word >= H * Wp) return;
  const int y = word / Wp, w = word - y * Wp, x = w * 64 + lane;
  bool b0 = false, b1 = false, b2 = false, b3 = false;
  if (x < W) {
    const uint8_t* c = rgb + ((int64_t)y * W + x) * 3;
    const unsigned r = c[0], g = c[1], b = c[2];
    const unsigned luma = (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;      // PIL convert("L")
    const unsigned gray = (r * 9798u + g * 19235u + b * 3735u + 16384u) >> 15;        // libpng rgb_to_gray (cv2.imread)
    b0 = 2 * (int)b <= *maxv;
    b1 = luma < 250u;
    b2 = luma <= 250u;
    b3 = gray < 250u;
  }
  const u64 m0 = __ballot(b0), m1 = __ballot(b1), m2 = __ballot(b2), m3 = __ballot(b3);
  if (lane == 0) {
    const int64_t ps = (int64_t)H * Wp;
    planes[word] = m0;
    planes[ps + word] = m1;
    planes[2 * ps + word] = m2;
    planes[3 * ps + word] = m3;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// depth scores: the depth value at every sample point and, per mask, whether the point lies inside it
__global__ __launch_bounds__(256) void rs_depth_samples_kernel(const u64* __restrict__ planes, int n,
                                                               const float* __restrict__ depth,
                                                               const int* __restrict__ pts, int P, int H, int W, int Wp,
                                                               float* __restrict__ vals, uint8_t* __restrict__ inside) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int y = pts[2 * p], x = pts[2 * p + 1];
  vals[p] = depth[(int64_t)y * W + x];
  const int64_t ps = (int64_t)H * Wp;
  for (int m = 0; m < n; ++m) inside[(int64_t)m * P + p] = (planes[m * ps + (int64_t)y * Wp + (x >> 6)] >> (x & 63)) & 1ull;
}

// ------------------------------------------------------------------------------------------------------------------
// pair tables.  D_m = dilate_cross(mask_m & binary) (cv2.dilate with the 3x3 ellipse = cross, depth_sort.py:193-197)
__global__ __launch_bounds__(256) void rs_stroke_dilate_kernel(const u64* __restrict__ planes,
                                                               const u64* __restrict__ binary, int H, int W, int Wp,
                                                               u64* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * Wp) return;
  const int y = idx / Wp, w = idx - y * Wp;
  const u64* src = planes + (int64_t)blockIdx.y * H * Wp;
  auto S = [&](int yy, int ww) -> u64 {
    if (yy < 0 || yy >= H || ww < 0 || ww >= Wp) return 0ull;
    return src[(int64_t)yy * Wp + ww] & binary[(int64_t)yy * Wp + ww];
  };
  const u64 c = S(y, w);
  u64 r = c | (c << 1) | (S(y, w - 1) >> 63) | (c >> 1) | (S(y, w + 1) << 63) | S(y - 1, w) | S(y + 1, w);
  out[(int64_t)blockIdx.y * H * Wp + idx] = r & bp_tail_mask(w, W);
}

// block (i, j), i <= j:  pair[i][j] = (|M_i & M_j| over the image, |D_i & D_j| inside rect[i][j] = rows [y0, y1) x
// columns [x0, x1));  i == j also:  per[i] = (|M_i|, |D_i|, |M_i & stroke250|);  block (0, 0) also: sketch_area.
__global__ __launch_bounds__(256) void rs_pair_counts_kernel(const u64* __restrict__ M, const u64* __restrict__ D,
                                                             const u64* __restrict__ stroke, int n, int H, int Wp,
                                                             const int* __restrict__ rect, int* __restrict__ pair,
                                                             int* __restrict__ per, int* __restrict__ sketch_area) {
  const int i = blockIdx.x / n, j = blockIdx.x % n;
  if (j < i) return;
  const int64_t ps = (int64_t)H * Wp;
  const u64 *mi = M + i * ps, *mj = M + j * ps, *di = D + i * ps, *dj = D + j * ps;
  const int* rc = rect + ((int64_t)i * n + j) * 4;
  const int y0 = rc[0], y1 = rc[1], x0 = rc[2], x1 = rc[3];
  int c_m = 0, c_d = 0, c_dd = 0, c_s = 0, c_sk = 0;
  for (int64_t k = threadIdx.x; k < ps; k += 256) {
    const int y = (int)(k / Wp), w = (int)(k - (int64_t)y * Wp);
    const u64 a = mi[k];
    c_m += __builtin_popcountll(a & mj[k]);
    const u64 d = di[k];
    if (y >= y0 && y < y1) {
      const int lo = max(x0 - w * 64, 0), hi = min(x1 - w * 64, 64);
      if (hi > lo) {
        const u64 xm = ((hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) >> lo) << lo;
        c_d += __builtin_popcountll(d & dj[k] & xm);
      }
    }
    if (i == j) {
      c_dd += __builtin_popcountll(d);
      const u64 s = stroke[k];
      c_s += __builtin_popcountll(a & s);
      if (i == 0) c_sk += __builtin_popcountll(s);
    }
  }
  __shared__ int red[4][5];
  int v[5] = {c_m, c_d, c_dd, c_s, c_sk};
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int t[5];
    for (int q = 0; q < 5; ++q) t[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
    pair[((int64_t)i * n + j) * 2] = pair[((int64_t)j * n + i) * 2] = t[0];
    pair[((int64_t)i * n + j) * 2 + 1] = pair[((int64_t)j * n + i) * 2 + 1] = t[1];
    if (i == j) {
      per[i * 3] = t[0];
      per[i * 3 + 1] = t[2];
      per[i * 3 + 2] = t[3];
      if (i == 0) *sketch_area = t[4];
    }
  }
}

// with no masks at all the sketch area is still wanted
__global__ __launch_bounds__(256) void rs_plane_count_kernel(const u64* __restrict__ plane, int64_t nwords,
                                                             int* __restrict__ out) {
  int c = 0;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nwords; k += (int64_t)gridDim.x * 256)
    c += __builtin_popcountll(plane[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------------------------------------
// composite (refiner.py:44-47): label = 1 + the first rank whose mask covers the pixel (earlier masks win), 0 if none.
// order[r] = mask index of rank r, < 0: the rank's mask was emptied (refiner.py:104-112).  hist[l] = pixels of label l.
__global__ __launch_bounds__(256) void rs_composite_kernel(const u64* __restrict__ planes, const int* __restrict__ order,
                                                           int nr, int H, int W, int Wp, uint8_t* __restrict__ label,
                                                           int* __restrict__ hist) {
  __shared__ int lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int word = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (word < H * Wp) {
    const int y = word / Wp, w = word - y * Wp, x = w * 64 + lane;
    int lab = 0;
    const int64_t ps = (int64_t)H * Wp;
    for (int r = 0; r < nr; ++r) {
      const int m = order[r];
      if (m < 0) continue;
      const u64 X = planes[m * ps + word];                    // one address per wave: a broadcast load
      if (lab == 0 && ((X >> lane) & 1ull)) lab = r + 1;
    }
    if (x < W) {
      label[(int64_t)y * W + x] = (uint8_t)lab;
      if (lab) atomicAdd(&lh[lab], 1);
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// relabel (dropped / merged ranks -> 0, survivors -> their final index + 1) and clean_delicate_mask (refiner.py:21-33):
// a pixel of mask L with at most one 8-neighbour in the SAME mask is removed.
__global__ __launch_bounds__(256) void rs_relabel_clean_kernel(const uint8_t* __restrict__ label,
                                                               const uint8_t* __restrict__ map, int H, int W,
                                                               uint8_t* __restrict__ out) {
  __shared__ uint8_t lm[256];
  lm[threadIdx.x] = map[threadIdx.x];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  int L = lm[label[p]];
  if (L) {
    int cnt = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        if ((dy | dx) == 0) continue;
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        cnt += lm[label[(int64_t)yy * W + xx]] == L;
      }
    if (cnt <= 1) L = 0;
  }
  out[p] = (uint8_t)L;
}

// ------------------------------------------------------------------------------------------------------------------
// growth (refine_masks_with_watershed, refiner.py:129-196; the watershed call never floods: every pixel inside the
// sketch mask is already a marker)
// plane of pixels with (pl bit set) AND label == 0
__global__ __launch_bounds__(256) void rs_unlabeled_plane_kernel(const uint8_t* __restrict__ label,
                                                                 const u64* __restrict__ pl, int H, int W, int Wp,
                                                                 u64* __restrict__ out) {
  const int word = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (word >= H * Wp) return;
  const int y = word / Wp, w = word - y * Wp, x = w * 64 + lane;
  const bool lab = x < W && label[(int64_t)y * W + x] != 0;
  const u64 m = __ballot(lab);
  if (lane == 0) out[word] = pl[word] & ~m;
}

// flags[L] |= 1 where a pixel of label L lies in `near` (= the large unlabeled regions dilated by disk(3))
__global__ __launch_bounds__(256) void rs_near_flags_kernel(const uint8_t* __restrict__ label,
                                                            const u64* __restrict__ near_pl, int H, int W, int Wp,
                                                            int* __restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  const int L = label[p];
  if (L && ((near_pl[(int64_t)y * Wp + (x >> 6)] >> (x & 63)) & 1ull)) atomicOr(&flags[L], 1);
}

// label3: mask pixels keep their label where the sketch plane is set; an unlabeled stroke pixel takes the LARGEST label
// L with a pixel of L inside disk(3 if flags[L] else 2) around it (later masks overwrite earlier ones, refiner.py:160-167)
__global__ __launch_bounds__(256) void rs_grow_kernel(const uint8_t* __restrict__ label, const u64* __restrict__ sk,
                                                      const int* __restrict__ flags, int H, int W, int Wp,
                                                      uint8_t* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  int L = label[p];
  const bool s = (sk[(int64_t)y * Wp + (x >> 6)] >> (x & 63)) & 1ull;
  if (!s) {
    L = 0;
  } else if (L == 0) {
    for (int dy = -3; dy <= 3; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -3; dx <= 3; ++dx) {
        const int xx = x + dx, d2 = dy * dy + dx * dx;
        if (xx < 0 || xx >= W || d2 > 9) continue;
        const int Lq = label[(int64_t)yy * W + xx];
        if (Lq > L && d2 <= (flags[Lq] ? 9 : 4)) L = Lq;
      }
    }
  }
  out[p] = (uint8_t)L;
}

// bounding box of every label: bbox[L] = (xmin, ymin, xmax, ymax), initialised to (W, H, -1, -1)
__global__ __launch_bounds__(256) void rs_bbox_init_kernel(int H, int W, int* __restrict__ bbox) {
  bbox[threadIdx.x * 4] = W;
  bbox[threadIdx.x * 4 + 1] = H;
  bbox[threadIdx.x * 4 + 2] = -1;
  bbox[threadIdx.x * 4 + 3] = -1;
}
__global__ __launch_bounds__(256) void rs_label_bbox_kernel(const uint8_t* __restrict__ label, int H, int W,
                                                            int* __restrict__ bbox) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int L = label[p];
  if (!L) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  atomicMin(&bbox[L * 4], x);
  atomicMin(&bbox[L * 4 + 1], y);
  atomicMax(&bbox[L * 4 + 2], x);
  atomicMax(&bbox[L * 4 + 3], y);
}

// raster-order list of the set pixels of a plane: row counts -> exclusive scan (one workgroup) -> (y, x) pairs
__global__ __launch_bounds__(256) void rs_row_counts_kernel(const u64* __restrict__ pl, int H, int Wp,
                                                            int* __restrict__ rowoff) {
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (y >= H) return;
  int c = 0;
  for (int w = lane; w < Wp; w += 64) c += __builtin_popcountll(pl[(int64_t)y * Wp + w]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) rowoff[y] = c;
}
__global__ __launch_bounds__(1024) void rs_row_scan_kernel(int* __restrict__ rowoff, int H, int* __restrict__ total) {
  __shared__ int carry, wt[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int y0 = 0; y0 < H; y0 += 1024) {
    const int y = y0 + threadIdx.x;
    const int v = y < H ? rowoff[y] : 0;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) wt[wave] = inc;
    __syncthreads();
    int base = carry;
    for (int k = 0; k < wave; ++k) base += wt[k];
    if (y < H) rowoff[y] = base + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void rs_list_pixels_kernel(const u64* __restrict__ pl, int H, int Wp,
                                                             const int* __restrict__ rowoff, int cap,
                                                             int* __restrict__ yx) {
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (y >= H) return;
  int off = rowoff[y];
  for (int w0 = 0; w0 < Wp; w0 += 64) {
    const int w = w0 + lane;
    u64 X = w < Wp ? pl[(int64_t)y * Wp + w] : 0ull;
    const int c = __builtin_popcountll(X);
    const int inc = wave_incl_scan(c, lane);
    int o = off + inc - c;
    while (X) {
      const int b = __builtin_ctzll(X);
      X &= X - 1;
      if (o < cap) {
        yx[2 * o] = y;
        yx[2 * o + 1] = w * 64 + b;
      }
      ++o;
    }
    off += __shfl(inc, 63, 64);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// squared distance from query pixel q to the nearest pixel of every CANDIDATE label (cand: 256-bit set per query).
// One workgroup per query scans rows outward (y, y-1, y+1, y-2, ...) and stops once dy^2 exceeds every candidate's
// best distance (an absent candidate label forces the full scan; its distance stays INT_MAX).
__global__ __launch_bounds__(256) void rs_query_dist_kernel(const uint8_t* __restrict__ label,
                                                            const int* __restrict__ qyx, const u64* __restrict__ cand,
                                                            int H, int W, int* __restrict__ d2) {
  __shared__ int best[256];
  __shared__ int bound;
  const int q = blockIdx.x;
  const int qy = qyx[2 * q], qx = qyx[2 * q + 1];
  best[threadIdx.x] = 0x7fffffff;
  __syncthreads();
  const bool mine = (cand[(int64_t)q * 4 + (threadIdx.x >> 6)] >> (threadIdx.x & 63)) & 1ull;
  const int maxdy = max(qy, H - 1 - qy);
  for (int dy = 0; dy <= maxdy; ++dy) {
    // bound = max over candidate labels of their best distance so far
    if (threadIdx.x == 0) bound = 0;
    __syncthreads();
    if (mine) atomicMax(&bound, best[threadIdx.x]);
    __syncthreads();
    if ((int64_t)dy * dy >= (int64_t)bound) break;            // uniform: `bound` is shared
    for (int side = 0; side < (dy ? 2 : 1); ++side) {
      const int yy = side ? qy + dy : qy - dy;
      if (yy < 0 || yy >= H) continue;
      const uint8_t* row = label + (int64_t)yy * W;
      for (int x = threadIdx.x; x < W; x += 256) {
        const int L = row[x];
        if (L) atomicMin(&best[L], dy * dy + (x - qx) * (x - qx));
      }
    }
    __syncthreads();
  }
  __syncthreads();
  d2[(int64_t)q * 256 + threadIdx.x] = best[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rs_apply_assign_kernel(uint8_t* __restrict__ label, int W,
                                                              const int* __restrict__ assign, int A) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= A) return;
  label[(int64_t)assign[3 * a] * W + assign[3 * a + 1]] = (uint8_t)assign[3 * a + 2];
}

// cv2.dilate with a 2x2 kernel, anchor (1, 1): a pixel also takes its upper, left and upper-left neighbours
__global__ __launch_bounds__(256) void bp_dilate2x2_kernel(const u64* __restrict__ in, u64* __restrict__ out, int H,
                                                           int W, int Wp) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * Wp) return;
  const int y = idx / Wp, w = idx - y * Wp;
  auto S = [&](int yy) -> u64 {
    if (yy < 0) return 0ull;
    const u64 c = in[(int64_t)yy * Wp + w], p = w > 0 ? in[(int64_t)yy * Wp + w - 1] : 0ull;
    return c | (c << 1) | (p >> 63);
  };
  out[idx] = (S(y) | S(y - 1)) & bp_tail_mask(w, W);
}

static inline dim3 words_grid(int H, int Wp, int n = 1) { return dim3((H * Wp + 3) / 4, n); }
static inline dim3 wordthreads_grid(int H, int Wp, int n = 1) { return dim3((H * Wp + 255) / 256, n); }
static inline dim3 pixels_grid(int H, int W) { return dim3((unsigned)(((int64_t)H * W + 255) / 256)); }

}  // namespace

// ====================================================================================================================
// C ABI
// ====================================================================================================================
extern "C" int ink_refine_sketch_planes(const void* rgb_u8, int32_t H, int32_t W, void* planes4_u64, int32_t* max_ws,
                                        void* stream) {
  INK_CHECK_ARG(rgb_u8 && planes4_u64 && max_ws && H > 0 && W > 0 && H <= 16383 && W <= 16383);
  hipStream_t s = (hipStream_t)stream;
  const int Wp = (W + 63) / 64;
  if (hipMemsetAsync(max_ws, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  hipLaunchKernelGGL(rs_max_kernel, dim3(256), dim3(256), 0, s, (const uint8_t*)rgb_u8, (int64_t)H * W * 3, max_ws);
  hipLaunchKernelGGL(rs_sketch_planes_kernel, words_grid(H, Wp), dim3(256), 0, s, (const uint8_t*)rgb_u8, H, W, Wp,
                     max_ws, (u64*)planes4_u64);
  return ink_launch_status();
}

extern "C" int ink_bitplane_pack(const void* img_u8, int32_t n, int32_t H, int32_t W, int32_t thresh, void* planes_u64,
                                 void* stream) {
  INK_CHECK_ARG(img_u8 && planes_u64 && n > 0 && H > 0 && W > 0 && H <= 16383 && W <= 16383);
  const int Wp = (W + 63) / 64;
  hipLaunchKernelGGL(bp_pack_kernel, words_grid(H, Wp, n), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img_u8, H,
                     W, Wp, thresh, (u64*)planes_u64);
  return ink_launch_status();
}

extern "C" int ink_refine_depth_samples(const void* mask_planes, int32_t n, const float* depth, const int32_t* pts_yx,
                                        int32_t P, int32_t H, int32_t W, float* vals, void* inside_u8, void* stream) {
  INK_CHECK_ARG(depth && pts_yx && vals && P > 0 && H > 0 && W > 0 && n >= 0 && (n == 0 || (mask_planes && inside_u8)));
  hipLaunchKernelGGL(rs_depth_samples_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const u64*)mask_planes, n, depth, pts_yx, P, H, W, (W + 63) / 64, vals, (uint8_t*)inside_u8);
  return ink_launch_status();
}

extern "C" int ink_refine_pair_tables(const void* mask_planes, const void* sketch_planes4, int32_t n, int32_t H,
                                      int32_t W, const int32_t* rect, void* dil_ws_planes, int32_t* pair,
                                      int32_t* per_mask, int32_t* sketch_area, void* stream) {
  INK_CHECK_ARG(sketch_planes4 && sketch_area && n >= 0 && H > 0 && W > 0);
  INK_CHECK_ARG(n == 0 || (mask_planes && rect && dil_ws_planes && pair && per_mask));
  hipStream_t s = (hipStream_t)stream;
  const int Wp = (W + 63) / 64;
  const int64_t ps = (int64_t)H * Wp;
  const u64* sk = (const u64*)sketch_planes4;
  if (n == 0) {
    if (hipMemsetAsync(sketch_area, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
    hipLaunchKernelGGL(rs_plane_count_kernel, dim3(64), dim3(256), 0, s, sk + ps, ps, sketch_area);
    return ink_launch_status();
  }
  hipLaunchKernelGGL(rs_stroke_dilate_kernel, wordthreads_grid(H, Wp, n), dim3(256), 0, s, (const u64*)mask_planes, sk,
                     H, W, Wp, (u64*)dil_ws_planes);
  hipLaunchKernelGGL(rs_pair_counts_kernel, dim3(n * n), dim3(256), 0, s, (const u64*)mask_planes,
                     (const u64*)dil_ws_planes, sk + ps, n, H, Wp, rect, pair, per_mask, sketch_area);
  return ink_launch_status();
}

extern "C" int ink_refine_composite(const void* mask_planes, const int32_t* order, int32_t n_ranks, int32_t H, int32_t W,
                                    void* label_u8, int32_t* hist256, void* stream) {
  INK_CHECK_ARG(label_u8 && hist256 && H > 0 && W > 0 && n_ranks >= 0 && n_ranks <= 254);
  INK_CHECK_ARG(n_ranks == 0 || (mask_planes && order));
  hipStream_t s = (hipStream_t)stream;
  const int Wp = (W + 63) / 64;
  if (hipMemsetAsync(hist256, 0, 256 * sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  hipLaunchKernelGGL(rs_composite_kernel, words_grid(H, Wp), dim3(256), 0, s, (const u64*)mask_planes, order, n_ranks, H,
                     W, Wp, (uint8_t*)label_u8, hist256);
  return ink_launch_status();
}

extern "C" int ink_refine_relabel_clean(const void* label_u8, const void* map256_u8, int32_t H, int32_t W,
                                        void* out_label_u8, void* stream) {
  INK_CHECK_ARG(label_u8 && map256_u8 && out_label_u8 && label_u8 != out_label_u8 && H > 0 && W > 0);
  hipLaunchKernelGGL(rs_relabel_clean_kernel, pixels_grid(H, W), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)label_u8, (const uint8_t*)map256_u8, H, W, (uint8_t*)out_label_u8);
  return ink_launch_status();
}

extern "C" int ink_refine_grow_workspace(int32_t H, int32_t W, int64_t* plane_words, int64_t* cc_ints) {
  INK_CHECK_ARG(plane_words && cc_ints && H > 0 && W > 0);
  const int Wp = (W + 63) / 64;
  *plane_words = 4 * (int64_t)H * Wp;                                  // U, tmp, closed / near, large
  *cc_ints = 1 + cc_ws_ints_per_plane(H, W / 2 + 1) + H + 1;           // overflow flag, CC workspace, row offsets, total
  return INK_OK;
}

extern "C" int ink_refine_grow(const void* label_u8, const void* sketch_planes4, int32_t H, int32_t W, void* planes_ws,
                               int32_t* cc_ws, int32_t* flags256, void* out_label_u8, int32_t* bbox256x4,
                               int32_t* unl_yx, int32_t unl_cap, int32_t* unl_count, void* stream) {
  INK_CHECK_ARG(label_u8 && sketch_planes4 && planes_ws && cc_ws && flags256 && out_label_u8 && bbox256x4 && unl_yx &&
                unl_count && unl_cap > 0 && H > 0 && W > 0 && H <= 16383 && W <= 16383 && label_u8 != out_label_u8);
  hipStream_t s = (hipStream_t)stream;
  const int Wp = (W + 63) / 64, RM = W / 2 + 1;
  const int64_t ps = (int64_t)H * Wp;
  const u64* sk_le = (const u64*)sketch_planes4 + 2 * ps;
  u64 *U = (u64*)planes_ws, *T = U + ps, *Cn = T + ps, *L = Cn + ps;
  const uint8_t* lab = (const uint8_t*)label_u8;
  uint8_t* out = (uint8_t*)out_label_u8;
  int* overflow = cc_ws;
  int* ccw = cc_ws + 1;
  int* rowoff = ccw + cc_ws_ints_per_plane(H, RM);
  if (hipMemsetAsync(overflow, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  if (hipMemsetAsync(flags256, 0, 256 * sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  // unlabeled stroke pixels; their closing by disk(3) (skimage: dilation ignores the outside, erosion counts it as set)
  hipLaunchKernelGGL(rs_unlabeled_plane_kernel, words_grid(H, Wp), dim3(256), 0, s, lab, sk_le, H, W, Wp, U);
  hipLaunchKernelGGL(bp_morph_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)U, T, H, W, Wp,
                     bp_shape_disk3(), 0, ps);
  hipLaunchKernelGGL(bp_morph_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)T, Cn, H, W, Wp,
                     bp_shape_disk3(), 1, ps);
  // 4-connected components of the closed image (scipy.ndimage.label's default structure) with more than 50 pixels
  if (cc_run(Cn, ps, 1, H, W, Wp, RM, 0, 50, 0.0, ccw, overflow, nullptr, L, ps, s) != INK_OK) return INK_ERR_LAUNCH;
  // masks that come within disk(3) of a large region grow by disk(3), the others by disk(2)
  hipLaunchKernelGGL(bp_morph_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)L, Cn, H, W, Wp,
                     bp_shape_disk3(), 0, ps);
  hipLaunchKernelGGL(rs_near_flags_kernel, pixels_grid(H, W), dim3(256), 0, s, lab, (const u64*)Cn, H, W, Wp, flags256);
  hipLaunchKernelGGL(rs_grow_kernel, pixels_grid(H, W), dim3(256), 0, s, lab, sk_le, (const int*)flags256, H, W, Wp, out);
  // what the box assignment needs: bounding boxes of the grown masks, the still unlabeled stroke pixels in raster order
  hipLaunchKernelGGL(rs_bbox_init_kernel, dim3(1), dim3(256), 0, s, H, W, bbox256x4);
  hipLaunchKernelGGL(rs_label_bbox_kernel, pixels_grid(H, W), dim3(256), 0, s, (const uint8_t*)out, H, W, bbox256x4);
  hipLaunchKernelGGL(rs_unlabeled_plane_kernel, words_grid(H, Wp), dim3(256), 0, s, (const uint8_t*)out, sk_le, H, W, Wp, U);
  hipLaunchKernelGGL(rs_row_counts_kernel, dim3((H + 3) / 4), dim3(256), 0, s, (const u64*)U, H, Wp, rowoff);
  hipLaunchKernelGGL(rs_row_scan_kernel, dim3(1), dim3(1024), 0, s, rowoff, H, unl_count);
  hipLaunchKernelGGL(rs_list_pixels_kernel, dim3((H + 3) / 4), dim3(256), 0, s, (const u64*)U, H, Wp,
                     (const int*)rowoff, unl_cap, unl_yx);
  return ink_launch_status();
}

extern "C" int ink_refine_query_dists(const void* label_u8, const int32_t* q_yx, const void* cand256_bits, int32_t Q,
                                      int32_t H, int32_t W, int32_t* d2_Qx256, void* stream) {
  INK_CHECK_ARG(label_u8 && q_yx && cand256_bits && d2_Qx256 && Q > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(rs_query_dist_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)label_u8, q_yx,
                     (const u64*)cand256_bits, H, W, d2_Qx256);
  return ink_launch_status();
}

extern "C" int ink_refine_finalize(void* label_u8, const int32_t* assign_yxl, int32_t A, const void* sketch_planes4,
                                   int32_t H, int32_t W, void* planes_ws3, void* extra_plane, int32_t* extra_count,
                                   void* stream) {
  INK_CHECK_ARG(label_u8 && sketch_planes4 && planes_ws3 && extra_plane && extra_count && H > 0 && W > 0 && A >= 0 &&
                (A == 0 || assign_yxl));
  hipStream_t s = (hipStream_t)stream;
  const int Wp = (W + 63) / 64;
  const int64_t ps = (int64_t)H * Wp;
  u64 *U = (u64*)planes_ws3, *T = U + ps, *O = T + ps;
  if (A > 0)
    hipLaunchKernelGGL(rs_apply_assign_kernel, dim3((A + 255) / 256), dim3(256), 0, s, (uint8_t*)label_u8, W, assign_yxl, A);
  // create_unlabeled_mask (refiner.py:301-337): stroke pixels (cv2 gray < 250) no mask claims, opened with a 3x3
  // square (cv2 borders: erosion counts the outside as set, dilation as clear), dilated with the anchored 2x2 square
  hipLaunchKernelGGL(rs_unlabeled_plane_kernel, words_grid(H, Wp), dim3(256), 0, s, (const uint8_t*)label_u8,
                     (const u64*)sketch_planes4 + 3 * ps, H, W, Wp, U);
  hipLaunchKernelGGL(bp_morph_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)U, T, H, W, Wp,
                     bp_shape_square3(), 1, ps);
  hipLaunchKernelGGL(bp_morph_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)T, O, H, W, Wp,
                     bp_shape_square3(), 0, ps);
  hipLaunchKernelGGL(bp_dilate2x2_kernel, wordthreads_grid(H, Wp), dim3(256), 0, s, (const u64*)O, (u64*)extra_plane, H, W, Wp);
  if (hipMemsetAsync(extra_count, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  hipLaunchKernelGGL(rs_plane_count_kernel, dim3(64), dim3(256), 0, s, (const u64*)extra_plane, ps, extra_count);
  return ink_launch_status();
}

// ====================================================================================================================
// host-side sequential pieces (HOST pointers; no GPU work)
// ====================================================================================================================
// sparse_sketch_sample (depth_sort.py:49-68) on the cv2 BGR view of an RGB sketch: take the first remaining stroke pixel
// in row-major order, drop every stroke pixel within radius = 0.01 H of it (inclusive), repeat.  The reference walks a
// Python set of point indices - for a set built from range(n) `next(iter(...))` is the smallest remaining index - and a
// KD-tree ball query; on a pixel grid that is: scan in raster order, a live stroke pixel becomes a sample and clears the
// disk around it.  out_yx: (y, x) pairs; *count = number of samples (may exceed cap: then only cap were written).
extern "C" int ink_host_sparse_sample(const uint8_t* rgb_u8, int32_t H, int32_t W, int32_t* out_yx, int32_t cap,
                                      int32_t* count) {
  INK_CHECK_ARG(rgb_u8 && out_yx && count && H > 0 && W > 0 && cap >= 0);
  const int64_t npx = (int64_t)H * W;
  int mx = 0;
  for (int64_t i = 0; i < npx * 3; ++i) mx = std::max(mx, (int)rgb_u8[i]);
  std::vector<uint8_t> live((size_t)npx);
  for (int64_t p = 0; p < npx; ++p) live[p] = 2 * (int)rgb_u8[p * 3 + 2] <= mx;       // blue <= max / 2
  const double radius = H * 0.01;
  const int r = (int)std::floor(radius);
  std::vector<int> half(2 * r + 1);                       // per row offset dy: largest |dx| with dy^2 + dx^2 <= radius^2
  for (int dy = -r; dy <= r; ++dy) {
    int h = -1;
    for (int dx = 0; dx <= r; ++dx)
      if ((double)(dy * dy + dx * dx) <= radius * radius) h = dx;
    half[dy + r] = h;
  }
  int n = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      if (!live[(int64_t)y * W + x]) continue;
      if (n < cap) { out_yx[2 * n] = y; out_yx[2 * n + 1] = x; }
      ++n;
      for (int dy = -r; dy <= r; ++dy) {
        const int yy = y + dy, h = half[dy + r];
        if (yy < 0 || yy >= H || h < 0) continue;
        const int a = std::max(0, x - h), b = std::min(W - 1, x + h);
        std::memset(&live[(int64_t)yy * W + a], 0, (size_t)(b - a + 1));
      }
    }
  *count = n;
  return INK_OK;
}

// The raster-order loop of refine_masks_with_boxes (refiner.py:262-295).  q_yx: the unlabeled stroke pixels in raster
// order; boxes: [nb, 4] (x1, y1, x2, y2), inclusive on both sides; box2mask[bi]: index of the mask matched to box bi or
// -1; d2: [Q, 256] squared distance from pixel q to the nearest ORIGINAL pixel of label l = mask index + 1 (INT_MAX: the
// mask has none; only the candidates of q need to be valid); nonempty[mi] != 0: mask mi has pixels before the loop.
// out_label[q] = label (mask index + 1) the pixel is assigned to, 0 = left unlabeled.
extern "C" int ink_host_assign_unlabeled(const int32_t* q_yx, int32_t Q, const int32_t* boxes, int32_t nb,
                                         const int32_t* box2mask, const int32_t* d2, const uint8_t* nonempty,
                                         int32_t n_masks, int32_t* out_label) {
  INK_CHECK_ARG(Q >= 0 && nb >= 0 && n_masks >= 0 && n_masks <= 255);
  INK_CHECK_ARG(Q == 0 || (q_yx && out_label && (nb == 0 || (boxes && box2mask && d2 && nonempty))));
  std::vector<std::vector<int32_t>> added((size_t)n_masks);          // pixels given to each mask so far (y, x pairs)
  std::vector<uint8_t> has(nonempty ? std::vector<uint8_t>(nonempty, nonempty + n_masks) : std::vector<uint8_t>());
  std::vector<int> inside;
  for (int q = 0; q < Q; ++q) {
    const int y = q_yx[2 * q], x = q_yx[2 * q + 1];
    out_label[q] = 0;
    inside.clear();
    for (int bi = 0; bi < nb; ++bi) {
      const int32_t* b = boxes + 4 * bi;
      if (b[0] <= x && x <= b[2] && b[1] <= y && y <= b[3]) inside.push_back(bi);
    }
    if (inside.empty()) continue;
    int mi = -1;
    if (inside.size() > 1) {
      int64_t best_d = -1;                                            // -1 = infinity
      for (int bi : inside) {
        const int m = box2mask[bi];
        if (m < 0 || m >= n_masks || !has[m]) continue;
        int64_t d = d2[(int64_t)q * 256 + m + 1];
        if (d == 0x7fffffff) d = -1;
        const std::vector<int32_t>& ad = added[m];
        for (size_t k = 0; k + 1 < ad.size(); k += 2) {
          const int64_t dy = ad[k] - y, dx = ad[k + 1] - x, dd = dy * dy + dx * dx;
          if (d < 0 || dd < d) d = dd;
        }
        if (d >= 0 && (best_d < 0 || d < best_d)) { best_d = d; mi = m; }     // strict <: the first nearest box wins
      }
    } else {
      mi = box2mask[inside[0]];
      if (mi >= n_masks) mi = -1;
    }
    if (mi >= 0) {
      out_label[q] = mi + 1;
      added[mi].push_back(y);
      added[mi].push_back(x);
      has[mi] = 1;
    }
  }
  return INK_OK;
}

// Bit-plane image primitives for the refinement stage: binary images live as ROW-ALIGNED
// bit planes - Wp = ceil(W / 64) uint64 words per row, bit b of word w of row y = pixel (y, 64 w + b), bits >= W of a
// row's last word are always 0 - so that morphology is word logic (64 pixels per lane-operation, 1/8 of the bytes of a
// uint8 mask) and pair statistics are popcounts.
//   * bp_morph_kernel: binary dilation / erosion by a small structuring element given as per-row half widths
//     (cross, 3x3 square, skimage disk(2), disk(3)); dilation ignores the outside of the image, erosion treats it as
//     foreground (skimage's binary_erosion / cv2.erode's default border);
//   * connected components with statistics, RUN based and MULTI-workgroup: runs per row (wave per row, start / end
//     bits by word logic) -> union of touching runs of adjacent rows inside blocks of CC_BR rows (one workgroup per
//     block and plane) -> union across the block seams (one workgroup per plane) -> area / bounding box per root by
//     atomics over runs -> keep / drop per component (stored per run) -> paint.  Every phase is its own launch: no workgroup ever reads
//     what another workgroup of the SAME launch wrote (per-XCD L2s are not coherent inside a launch).
// Included by refine.hip (mask cleanup) and refine_stage.hip (disjoint parsing / growth).
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 bp_tail_mask(int w, int W) {       // valid bits of word w in a row of W pixels
  const int rem = W - w * 64;
  return rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
}

// a row spread by +-h pixels; prev / next = the neighbouring words of the row (0 outside it)
__device__ __forceinline__ u64 bp_hspread(u64 prev, u64 cur, u64 next, int h) {
  u64 r = cur;
  for (int s = 1; s <= h; ++s) r |= (cur << s) | (prev >> (64 - s)) | (cur >> s) | (next << (64 - s));
  return r;
}

struct BpShape {          // structuring element: rows dy = -R .. R, half width hw[dy + R] (< 0: row not part of it)
  int R;
  int hw[7];
};
static inline BpShape bp_shape_cross() { return BpShape{1, {0, 1, 0, -1, -1, -1, -1}}; }          // disk(1) / cv2 3x3 ellipse
static inline BpShape bp_shape_square3() { return BpShape{1, {1, 1, 1, -1, -1, -1, -1}}; }
static inline BpShape bp_shape_disk2() { return BpShape{2, {0, 1, 2, 1, 0, -1, -1}}; }           // x^2 + y^2 <= 4
static inline BpShape bp_shape_disk3() { return BpShape{3, {0, 2, 2, 3, 2, 2, 0}}; }             // x^2 + y^2 <= 9

// out = dilate(in) (erode == 0) or erode(in) (erode != 0, computed as ~dilate(~in) with the outside of the image and
// the tail bits of ~in cleared, i.e. the outside of `in` counted as foreground).  One thread per output word.
__global__ __launch_bounds__(256) void bp_morph_kernel(const u64* __restrict__ in, u64* __restrict__ out, int H, int W,
                                                       int Wp, BpShape sh, int erode, int64_t plane_stride) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * Wp) return;
  const int y = idx / Wp, w = idx - y * Wp;
  const u64* src = in + (int64_t)blockIdx.y * plane_stride;
  u64 acc = 0;
  for (int dy = -sh.R; dy <= sh.R; ++dy) {
    const int hh = sh.hw[dy + sh.R];
    const int yy = y + dy;
    if (hh < 0 || yy < 0 || yy >= H) continue;
    const u64* row = src + (int64_t)yy * Wp;
    u64 p = w > 0 ? row[w - 1] : 0ull, c = row[w], n = w + 1 < Wp ? row[w + 1] : 0ull;
    if (erode) {
      p = w > 0 ? ~p : 0ull;                                   // word w - 1 is never a row's last word
      c = ~c & bp_tail_mask(w, W);
      n = w + 1 < Wp ? (~n & bp_tail_mask(w + 1, W)) : 0ull;
    }
    acc |= bp_hspread(p, c, n, hh);
  }
  out[(int64_t)blockIdx.y * plane_stride + idx] = (erode ? ~acc : acc) & bp_tail_mask(w, W);
}

// uint8 images [n, H, W] -> bit planes (pixel > thresh)
__global__ __launch_bounds__(256) void bp_pack_kernel(const uint8_t* __restrict__ img, int H, int W, int Wp, int thresh,
                                                      u64* __restrict__ planes) {
  const int word = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (word >= H * Wp) return;
  const int y = word / Wp, w = word - y * Wp, x = w * 64 + lane;
  const bool on = x < W && (int)img[((int64_t)blockIdx.y * H + y) * W + x] > thresh;
  const u64 m = __ballot(on);
  if (lane == 0) planes[(int64_t)blockIdx.y * H * Wp + word] = m;
}

// ------------------------------------------------------------------------------------------------------------------
// connected components on bit planes
// Workspace per plane (int32): nruns[H] | run[H*RM] (start | end << 16) | parent[H*RM] | area[H*RM] | xmin | xmax | ymin
// | ymax (each [H*RM]); run id = y * RM + index in row.
#define CC_BR 32                               // rows per block of the intra-block union launch
static inline int64_t cc_ws_ints_per_plane(int H, int RM) { return (int64_t)H + 7 * (int64_t)H * RM; }

struct CcWs {
  int* nruns;
  int* run;
  int* parent;
  int* area;
  int* xmin;
  int* xmax;
  int* ymin;
  int* ymax;
};
__device__ __forceinline__ CcWs cc_ws(int* base, int64_t stride, int plane, int H, int RM) {
  CcWs w;
  const int64_t NR = (int64_t)H * RM;
  w.nruns = base + (int64_t)plane * stride;
  w.run = w.nruns + H;
  w.parent = w.run + NR;
  w.area = w.parent + NR;
  w.xmin = w.area + NR;
  w.xmax = w.xmin + NR;
  w.ymin = w.xmax + NR;
  w.ymax = w.ymin + NR;
  return w;
}

__device__ __forceinline__ int cc_find(int* parent, int x) {
  int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    x = p;
    p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}
__device__ __forceinline__ void cc_union(int* parent, int a, int b) {
  while (true) {
    a = cc_find(parent, a);
    b = cc_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }           // hook the larger root under the smaller one
    const int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;
  }
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// A. runs of every row: one wave per row; lane l owns word w0 + l.  A run starts at a set bit whose left neighbour is
// clear and ends at a set bit whose right neighbour is clear; the k-th start and the k-th end of a row belong to the
// same run, so starts and ends are written independently (16-bit halves of run[]).
__global__ __launch_bounds__(256) void cc_runs_kernel(const u64* __restrict__ planes, int64_t plane_stride, int H,
                                                      int W, int Wp, int RM, int* __restrict__ ws_all,
                                                      int64_t ws_stride, int* __restrict__ overflow) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= H) return;
  const CcWs ws = cc_ws(ws_all, ws_stride, blockIdx.y, H, RM);
  const u64* row = planes + (int64_t)blockIdx.y * plane_stride + (int64_t)y * Wp;
  unsigned short* rr = (unsigned short*)(ws.run + (int64_t)y * RM);
  int ns = 0, ne = 0;
  for (int w0 = 0; w0 < Wp; w0 += 64) {
    const int w = w0 + lane;
    const u64 X = w < Wp ? row[w] : 0ull;
    const u64 P = (w > 0 && w <= Wp) ? row[w - 1] : 0ull;
    const u64 N = w + 1 < Wp ? row[w + 1] : 0ull;
    u64 starts = X & ~((X << 1) | (P >> 63));
    u64 ends = X & ~((X >> 1) | (N << 63));
    const int cs = __builtin_popcountll(starts), ce = __builtin_popcountll(ends);
    const int is = wave_incl_scan(cs, lane), ie = wave_incl_scan(ce, lane);
    int os = ns + is - cs, oe = ne + ie - ce;
    while (starts) {
      const int b = __builtin_ctzll(starts);
      starts &= starts - 1;
      if (os < RM) rr[2 * os] = (unsigned short)(w * 64 + b);
      ++os;
    }
    while (ends) {
      const int b = __builtin_ctzll(ends);
      ends &= ends - 1;
      if (oe < RM) rr[2 * oe + 1] = (unsigned short)(w * 64 + b);
      ++oe;
    }
    ns += __shfl(is, 63, 64);
    ne += __shfl(ie, 63, 64);
  }
  if (ns > RM && lane == 0) atomicOr(overflow, 1);
  const int n = ns < RM ? ns : RM;
  if (lane == 0) ws.nruns[y] = n;
  for (int i = lane; i < n; i += 64) {
    const int64_t id = (int64_t)y * RM + i;
    ws.parent[id] = (int)id;
    ws.area[id] = 0;
    ws.xmin[id] = W;
    ws.xmax[id] = -1;
    ws.ymin[id] = H;
    ws.ymax[id] = -1;
  }
}

// B. union of touching runs of rows y and y - 1.  conn8: [s, e] touches [s', e'] iff s' <= e + 1 and e' >= s - 1;
// 4-connectivity: s' <= e and e' >= s.  One wave per row, lanes over the row's runs, binary search into the sorted
// runs of the row above.  seams == 0: rows inside block blockIdx.x (y % CC_BR != 0); seams != 0: the rows y = k CC_BR.
__global__ __launch_bounds__(256) void cc_link_kernel(int H, int RM, int conn8, int seams, int* __restrict__ ws_all,
                                                      int64_t ws_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const CcWs ws = cc_ws(ws_all, ws_stride, blockIdx.y, H, RM);
  const int c = conn8 ? 1 : 0;
  int y, ystep, yend;
  if (seams) {
    y = (1 + wave) * CC_BR;
    ystep = 4 * CC_BR;
    yend = H;
  } else {
    y = blockIdx.x * CC_BR + 1 + wave;
    ystep = 4;
    yend = (blockIdx.x + 1) * CC_BR < H ? (blockIdx.x + 1) * CC_BR : H;
  }
  for (; y < yend; y += ystep) {
    const int na = ws.nruns[y], nb = ws.nruns[y - 1];
    if (nb == 0) continue;
    const int* ra = ws.run + (int64_t)y * RM;
    const int* rb = ws.run + (int64_t)(y - 1) * RM;
    for (int i = lane; i < na; i += 64) {
      const int r = ra[i];
      const int s = r & 0xffff, e = (r >> 16) & 0xffff;
      int lo = 0, hi = nb;                                   // first j with end_j >= s - c
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (((rb[mid] >> 16) & 0xffff) < s - c) lo = mid + 1; else hi = mid;
      }
      for (int j = lo; j < nb; ++j) {
        if ((rb[j] & 0xffff) > e + c) break;
        cc_union(ws.parent, y * RM + i, (y - 1) * RM + j);
      }
    }
  }
}

// C. statistics per root: area, bounding box (atomics over runs).  Parents are only READ here.
__global__ __launch_bounds__(256) void cc_stats_kernel(int H, int RM, int* __restrict__ ws_all, int64_t ws_stride) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= H) return;
  const CcWs ws = cc_ws(ws_all, ws_stride, blockIdx.y, H, RM);
  const int n = ws.nruns[y];
  for (int i = lane; i < n; i += 64) {
    const int id = y * RM + i;
    const int root = cc_find(ws.parent, id);
    const int r = ws.run[id];
    const int s = r & 0xffff, e = (r >> 16) & 0xffff;
    atomicAdd(&ws.area[root], e - s + 1);
    atomicMin(&ws.xmin[root], s);
    atomicMax(&ws.xmax[root], e);
    atomicMin(&ws.ymin[root], y);
    atomicMax(&ws.ymax[root], y);
  }
}

// D. keep / drop per component: keep iff area > area_thr, or (aspect_thr > 0 and max(w, h) / (min(w, h) + 1e-5) >
// aspect_thr) (mask_cleaner.py:27-33).  Every run stores the verdict of its component in bit 31 of its OWN run word
// (coordinates are < 2^14), so the paint pass needs no find.
__global__ __launch_bounds__(256) void cc_decide_kernel(int H, int RM, int area_thr, double aspect_thr,
                                                        int* __restrict__ ws_all, int64_t ws_stride) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= H) return;
  const CcWs ws = cc_ws(ws_all, ws_stride, blockIdx.y, H, RM);
  const int n = ws.nruns[y];
  for (int i = lane; i < n; i += 64) {
    const int id = y * RM + i;
    const int root = cc_find(ws.parent, id);
    bool keep = ws.area[root] > area_thr;
    if (!keep && aspect_thr > 0.0) {
      const int bw = ws.xmax[root] - ws.xmin[root] + 1, bh = ws.ymax[root] - ws.ymin[root] + 1;
      keep = (double)(bw > bh ? bw : bh) / ((double)(bw < bh ? bw : bh) + 1e-5) > aspect_thr;
    }
    if (keep) ws.run[id] |= (int)0x80000000;
  }
}

// E. paint the kept runs: into a uint8 image (0 / 255; the wave walks the row's runs and paints each cooperatively) when
// out_u8 is given, else into a bit plane (lanes own runs and OR their bit ranges into the row's words in LDS).
__global__ __launch_bounds__(256) void cc_paint_kernel(int H, int W, int Wp, int RM, const int* __restrict__ ws_all,
                                                       int64_t ws_stride, uint8_t* __restrict__ out_u8,
                                                       u64* __restrict__ out_planes, int64_t out_plane_stride) {
  extern __shared__ u64 roww[];                              // [4][Wp] (plane output only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = blockIdx.x * 4 + wave;
  const CcWs ws = cc_ws(const_cast<int*>(ws_all), ws_stride, blockIdx.y, H, RM);
  const int n = y < H ? ws.nruns[y] : 0;
  const int* rr = ws.run + (int64_t)y * RM;
  if (out_u8) {
    if (y >= H) return;
    uint8_t* op = out_u8 + ((int64_t)blockIdx.y * H + y) * W;
    for (int x = lane; x < W; x += 64) op[x] = 0;
    for (int i = 0; i < n; ++i) {                            // uniform over the wave
      const int r = rr[i];
      if (r >= 0) continue;
      const int s = r & 0xffff, e = (r >> 16) & 0x7fff;
      for (int x = s + lane; x <= e; x += 64) op[x] = 255;
    }
    return;
  }
  u64* mine = roww + wave * Wp;
  for (int w = lane; w < Wp; w += 64) mine[w] = 0ull;
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const int r = rr[i];
    if (r >= 0) continue;
    const int s = r & 0xffff, e = (r >> 16) & 0x7fff;
    for (int w = s >> 6; w <= (e >> 6); ++w) {
      const int a = s > w * 64 ? s - w * 64 : 0, b = e < w * 64 + 63 ? e - w * 64 : 63;
      atomicOr(&mine[w], ((b == 63 ? ~0ull : ((1ull << (b + 1)) - 1ull)) >> a) << a);
    }
  }
  __syncthreads();
  if (y < H)
    for (int w = lane; w < Wp; w += 64) out_planes[(int64_t)blockIdx.y * out_plane_stride + (int64_t)y * Wp + w] = mine[w];
}

// host-side launch sequence of the six phases for `np` planes
static inline int cc_run(const u64* planes, int64_t plane_stride, int np, int H, int W, int Wp, int RM, int conn8,
                         int area_thr, double aspect_thr, int* ws, int* overflow, uint8_t* out_u8, u64* out_planes,
                         int64_t out_plane_stride, hipStream_t s) {
  const int64_t stride = cc_ws_ints_per_plane(H, RM);
  const dim3 rows((H + 3) / 4, np);
  hipLaunchKernelGGL(cc_runs_kernel, rows, dim3(256), 0, s, planes, plane_stride, H, W, Wp, RM, ws, stride, overflow);
  hipLaunchKernelGGL(cc_link_kernel, dim3((H + CC_BR - 1) / CC_BR, np), dim3(256), 0, s, H, RM, conn8, 0, ws, stride);
  if (H > CC_BR) hipLaunchKernelGGL(cc_link_kernel, dim3(1, np), dim3(256), 0, s, H, RM, conn8, 1, ws, stride);
  hipLaunchKernelGGL(cc_stats_kernel, rows, dim3(256), 0, s, H, RM, ws, stride);
  hipLaunchKernelGGL(cc_decide_kernel, rows, dim3(256), 0, s, H, RM, area_thr, aspect_thr, ws, stride);
  hipLaunchKernelGGL(cc_paint_kernel, rows, dim3(256), out_u8 ? 0 : 4 * Wp * sizeof(u64), s, H, W, Wp, RM,
                     (const int*)ws, stride, out_u8, out_planes, out_plane_stride);
  return ink_launch_status();
}

}  // namespace

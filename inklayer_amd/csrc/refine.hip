// Mask hand-off stage of InkLayer's refinement on resident masks: morphological cleanup of the SAM
// masks and the pairwise "sketch IoU" table of the sketch NMS.  Integer / byte work, HBM-bound; bit-exact against the
// reference's cv2 results (oracle/refine_ref.py, pinned by the reference's own committed outputs).
//   reference: InkLayer/refinement/mask_cleaner.py:11-36  (threshold -> MORPH_CLOSE with a k x k rect -> 8-connected
//              components with stats -> keep area > 500 or aspect ratio > 1.1)
//              InkLayer/refinement/nms_sketch.py:62-78,186-234 (masks AND stroke pixels, |A&B| / |A|B| per pair)
#include "common.h"
#include "../../include/inklayer_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// MORPH_CLOSE with a k x k rectangle = dilate then erode, separable: four box passes on 0/1 bytes.
// A box "any" (dilate) / "all" (erode) over [x-r, x+r] clipped to the image (pixels outside never win, OpenCV's default
// morphology border) is a window COUNT: any <=> count > 0, all <=> count == number of in-image pixels in the window.
// Horizontal pass: one workgroup per row, inclusive prefix sum of the row in LDS.  mode 0: dilate, 1: erode.
// `thresh`: input pixels are compared with > thresh (127 for the raw mask, 0 for the 0/1 intermediates).
__global__ __launch_bounds__(256) void box_rows_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       int W, int r, int mode, int thresh) {
  extern __shared__ int pre[];                     // [W + 1]: pre[i] = number of set pixels in [0, i)
  __shared__ int wave_tot[4];
  const int64_t row = blockIdx.x;
  const uint8_t* ip = in + row * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (W + 255) / 256;                 // consecutive pixels per thread
  const int x0 = tid * per;
  int cnt = 0;
  for (int j = 0; j < per; ++j) {
    const int x = x0 + j;
    if (x < W) cnt += ip[x] > thresh ? 1 : 0;
  }
  // exclusive scan of the per-thread counts: wave scan + wave totals
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = inc - cnt;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  if (tid == 0) pre[0] = 0;
  int run = base;
  for (int j = 0; j < per; ++j) {
    const int x = x0 + j;
    if (x < W) {
      run += ip[x] > thresh ? 1 : 0;
      pre[x + 1] = run;
    }
  }
  __syncthreads();
  uint8_t* op = out + row * W;
  for (int x = tid; x < W; x += 256) {
    const int lo = x - r < 0 ? 0 : x - r, hi = x + r >= W ? W - 1 : x + r;
    const int c = pre[hi + 1] - pre[lo];
    op[x] = mode == 0 ? (c > 0) : (c == hi - lo + 1);
  }
}

// Vertical pass: one thread per column (coalesced across the wave), walking down a segment of rows with a running
// window count.  Input is 0/1 bytes.
__global__ __launch_bounds__(256) void box_cols_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       int H, int W, int r, int mode, int seg) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int64_t m = blockIdx.z;
  const int y0 = blockIdx.y * seg, y1 = y0 + seg < H ? y0 + seg : H;
  const uint8_t* ip = in + m * H * W + x;
  uint8_t* op = out + m * H * W + x;
  int c = 0;
  {
    const int lo = y0 - r < 0 ? 0 : y0 - r, hi = y0 + r >= H ? H - 1 : y0 + r;
    for (int y = lo; y <= hi; ++y) c += ip[(int64_t)y * W];
  }
  for (int y = y0; y < y1; ++y) {
    const int lo = y - r < 0 ? 0 : y - r, hi = y + r >= H ? H - 1 : y + r;
    op[(int64_t)y * W] = mode == 0 ? (c > 0) : (c == hi - lo + 1);
    if (y + 1 + r < H) c += ip[(int64_t)(y + 1 + r) * W];      // row entering the window of y + 1
    if (y - r >= 0) c -= ip[(int64_t)(y - r) * W];              // row leaving it
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 8-connected components with stats + the area / aspect-ratio filter, RUN based, one workgroup per mask.
// Workspace per mask (int32): nruns[H] | run[H*RM] (start | end << 16) | parent[H*RM] | area | xmin | xmax | ymin | ymax
// (each [H*RM]); run id = y * RM + index in row.  After a k x k closing every background gap in a row is at least
// (k+1)/2 pixels long, so RM = W / ((k+1)/2 + 1) + 2 bounds the runs of a row (ink_mask_cleanup sizes it so).
__device__ __forceinline__ int cc_find(volatile int* parent, int x) {
  int p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
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

__global__ __launch_bounds__(1024) void components_filter_kernel(const uint8_t* __restrict__ closed,
                                                                 uint8_t* __restrict__ out, int H, int W, int RM,
                                                                 int area_thr, double aspect_thr,
                                                                 int* __restrict__ ws_all, int64_t ws_stride,
                                                                 int* __restrict__ overflow) {
  const int64_t m = blockIdx.x;
  const uint8_t* img = closed + m * H * W;
  uint8_t* dst = out + m * H * W;
  int* ws = ws_all + m * ws_stride;
  int* nruns = ws;
  int* run = nruns + H;
  const int64_t NR = (int64_t)H * RM;
  int* parent = run + NR;
  int* area = parent + NR;
  int* xmin = area + NR;
  int* xmax = xmin + NR;
  int* ymin = xmax + NR;
  int* ymax = ymin + NR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;

  // A. runs of every row: one wave per row, 64 pixels per step, transitions from the ballot of the foreground bits
  for (int y = wave; y < H; y += nwave) {
    const uint8_t* rp = img + (int64_t)y * W;
    int n = 0, open_start = -1;
    for (int x0 = 0; x0 < W; x0 += 64) {
      const int x = x0 + lane;
      const bool fg = x < W && rp[x] != 0;
      unsigned long long bits = __ballot(fg);
      // every lane walks the same transitions (uniform control flow, no divergence)
      int pos = 0;
      while (pos < 64) {
        if (open_start < 0) {
          const unsigned long long rest = pos < 64 ? (bits >> pos) : 0ull;
          if (rest == 0ull) break;
          pos += __builtin_ctzll(rest);
          open_start = x0 + pos;
        } else {
          const unsigned long long rest = ~bits >> pos;        // first background bit at or after pos
          const int z = pos < 64 && rest != 0ull ? __builtin_ctzll(rest) : 64;
          if (pos + z >= 64) { pos = 64; break; }              // run continues into the next 64-pixel group
          pos += z;
          if (n < RM) {
            if (lane == 0) run[(int64_t)y * RM + n] = open_start | ((x0 + pos - 1) << 16);
          } else if (lane == 0) {
            atomicOr(overflow, 1);
          }
          n = n < RM ? n + 1 : n;
          open_start = -1;
        }
      }
    }
    if (open_start >= 0) {                                     // run touching the right border
      if (n < RM) {
        if (lane == 0) run[(int64_t)y * RM + n] = open_start | ((W - 1) << 16);
        ++n;
      } else if (lane == 0) {
        atomicOr(overflow, 1);
      }
    }
    if (lane == 0) nruns[y] = n;
    for (int i = lane; i < n; i += 64) {
      const int64_t id = (int64_t)y * RM + i;
      parent[id] = (int)id;
      area[id] = 0;
      xmin[id] = W;
      xmax[id] = -1;
      ymin[id] = H;
      ymax[id] = -1;
    }
  }
  __syncthreads();
  // B. union of 8-connected runs of adjacent rows: [s, e] touches [s', e'] of the row above iff s' <= e + 1 and
  //    e' >= s - 1.  One thread per row, two-pointer walk over the two sorted run lists.
  for (int y = 1 + tid; y < H; y += blockDim.x) {
    const int na = nruns[y], nb = nruns[y - 1];
    int j = 0;
    for (int i = 0; i < na; ++i) {
      const int ra = run[(int64_t)y * RM + i];
      const int s = ra & 0xffff, e = ra >> 16;
      while (j < nb && (run[(int64_t)(y - 1) * RM + j] >> 16) < s - 1) ++j;
      int jj = j;
      while (jj < nb) {
        const int rb = run[(int64_t)(y - 1) * RM + jj];
        if ((rb & 0xffff) > e + 1) break;
        cc_union(parent, y * RM + i, (y - 1) * RM + jj);
        ++jj;
      }
    }
  }
  __syncthreads();
  // C + D. flatten, accumulate area and bounding box per root
  for (int y = wave; y < H; y += nwave) {
    const int n = nruns[y];
    for (int i = lane; i < n; i += 64) {
      const int id = y * RM + i;
      const int root = cc_find(parent, id);
      parent[id] = root;
      const int ra = run[id];
      const int s = ra & 0xffff, e = ra >> 16;
      atomicAdd(&area[root], e - s + 1);
      atomicMin(&xmin[root], s);
      atomicMax(&xmax[root], e);
      atomicMin(&ymin[root], y);
      atomicMax(&ymax[root], y);
    }
  }
  __syncthreads();
  // E. decision per root (in place: area[root] becomes 1 = keep / 0 = drop)
  for (int y = wave; y < H; y += nwave) {
    const int n = nruns[y];
    for (int i = lane; i < n; i += 64) {
      const int id = y * RM + i;
      if (parent[id] == id) {
        const int w = xmax[id] - xmin[id] + 1, h = ymax[id] - ymin[id] + 1;
        const double ar = (double)(w > h ? w : h) / ((double)(w < h ? w : h) + 1e-5);
        xmin[id] = (area[id] > area_thr || ar > aspect_thr) ? 1 : 0;     // xmin doubles as the keep flag from here on
      }
    }
  }
  __syncthreads();
  // F. output: one wave per row, coalesced; zero the row, then paint the kept runs with 255
  for (int y = wave; y < H; y += nwave) {
    uint8_t* op = dst + (int64_t)y * W;
    for (int x = lane; x < W; x += 64) op[x] = 0;
    const int n = nruns[y];
    for (int i = 0; i < n; ++i) {
      const int id = y * RM + i;
      if (xmin[parent[id]] != 0) {
        const int ra = run[id];
        const int s = ra & 0xffff, e = ra >> 16;
        for (int x = s + lane; x <= e; x += 64) op[x] = 255;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Sketch NMS table.  refined_m = (mask_m > 0) AND (luma(sketch) < 250)  (refine_mask_to_sketch_regions), bit-packed
// 64 pixels per word by wave ballot; then counts[i, j] = (popcount(r_i & r_j), popcount(r_i | r_j)).
__global__ __launch_bounds__(256) void pack_refined_kernel(const uint8_t* __restrict__ masks,
                                                           const uint8_t* __restrict__ rgb, int64_t npix,
                                                           int64_t nwords, unsigned long long* __restrict__ bits) {
  const int64_t m = blockIdx.y;
  const int64_t word = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (word >= nwords) return;
  const int64_t p = word * 64 + (threadIdx.x & 63);
  bool on = false;
  if (p < npix && masks[m * npix + p] != 0) {
    const uint8_t* c = rgb + p * 3;
    // PIL Image.convert("L"): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
    const unsigned l = ((unsigned)c[0] * 19595u + (unsigned)c[1] * 38470u + (unsigned)c[2] * 7471u + 0x8000u) >> 16;
    on = l < 250u;
  }
  const unsigned long long b = __ballot(on);
  if ((threadIdx.x & 63) == 0) bits[m * nwords + word] = b;
}

__global__ __launch_bounds__(256) void pair_counts_kernel(const unsigned long long* __restrict__ bits, int n,
                                                          int64_t nwords, int32_t* __restrict__ counts) {
  const int i = blockIdx.x / n, j = blockIdx.x % n;
  if (j < i) return;                                  // symmetric: the (i <= j) half is computed, then mirrored
  const unsigned long long* a = bits + (int64_t)i * nwords;
  const unsigned long long* b = bits + (int64_t)j * nwords;
  int inter = 0, uni = 0;
  for (int64_t w = threadIdx.x; w < nwords; w += 256) {
    const unsigned long long x = a[w], y = b[w];
    inter += __builtin_popcountll(x & y);
    uni += __builtin_popcountll(x | y);
  }
  __shared__ int si[4], su[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_xor(inter, o, 64);
    uni += __shfl_xor(uni, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { si[threadIdx.x >> 6] = inter; su[threadIdx.x >> 6] = uni; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ti = si[0] + si[1] + si[2] + si[3], tu = su[0] + su[1] + su[2] + su[3];
    counts[((int64_t)i * n + j) * 2] = ti;
    counts[((int64_t)i * n + j) * 2 + 1] = tu;
    counts[((int64_t)j * n + i) * 2] = ti;
    counts[((int64_t)j * n + i) * 2 + 1] = tu;
  }
}

}  // namespace

extern "C" int ink_mask_cleanup_workspace_ints(int32_t n, int32_t H, int32_t W, int32_t k, int64_t* out_ints) {
  INK_CHECK_ARG(out_ints && n > 0 && H > 0 && W > 0 && k >= 1 && k % 2 == 1);
  const int gap = (k + 1) / 2;                       // shortest background gap of a row after a k x k closing
  const int64_t RM = W / (gap + 1) + 2;
  *out_ints = 1 + (int64_t)n * ((int64_t)H + 7 * (int64_t)H * RM);    // [0] = overflow flag, then n per-mask blocks
  return INK_OK;
}

extern "C" int ink_mask_cleanup(const void* masks_u8, int32_t n, int32_t H, int32_t W, int32_t k,
                                int32_t area_threshold, double aspect_threshold, void* tmp_a_u8, void* tmp_b_u8,
                                int32_t* workspace, void* out_u8, void* stream) {
  INK_CHECK_ARG(masks_u8 && tmp_a_u8 && tmp_b_u8 && workspace && out_u8);
  INK_CHECK_ARG(n > 0 && H > 0 && W > 0 && H <= 16383 && W <= 16383 && k >= 1 && k % 2 == 1);
  hipStream_t s = (hipStream_t)stream;
  const int r = k / 2;
  uint8_t* a = (uint8_t*)tmp_a_u8;
  uint8_t* b = (uint8_t*)tmp_b_u8;
  const size_t lds = (size_t)(W + 1) * sizeof(int);
  INK_CHECK_ARG(lds <= 64 * 1024);
  const int seg = 128;
  const dim3 cgrid((W + 255) / 256, (H + seg - 1) / seg, n);
  // dilate: rows (threshold > 127, cv2.threshold) then columns; erode: rows then columns
  hipLaunchKernelGGL(box_rows_kernel, dim3(n * H), dim3(256), lds, s, (const uint8_t*)masks_u8, a, W, r, 0, 127);
  hipLaunchKernelGGL(box_cols_kernel, cgrid, dim3(256), 0, s, a, b, H, W, r, 0, seg);
  hipLaunchKernelGGL(box_rows_kernel, dim3(n * H), dim3(256), lds, s, b, a, W, r, 1, 0);
  hipLaunchKernelGGL(box_cols_kernel, cgrid, dim3(256), 0, s, a, b, H, W, r, 1, seg);
  const int gap = (k + 1) / 2;
  const int RM = W / (gap + 1) + 2;
  const int64_t stride = (int64_t)H + 7 * (int64_t)H * RM;
  // workspace[0] = overflow flag (stays 0 by construction; checked by the host wrapper in debug runs)
  if (hipMemsetAsync(workspace, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  hipLaunchKernelGGL(components_filter_kernel, dim3(n), dim3(1024), 0, s, b, (uint8_t*)out_u8, H, W, RM,
                     area_threshold, aspect_threshold, workspace + 1, stride, workspace);
  return ink_launch_status();
}

extern "C" int ink_mask_sketch_iou_counts(const void* masks_u8, const void* sketch_rgb_u8, int32_t n, int32_t H,
                                          int32_t W, void* bits_ws_u64, int32_t* counts, void* stream) {
  INK_CHECK_ARG(masks_u8 && sketch_rgb_u8 && bits_ws_u64 && counts && n > 0 && H > 0 && W > 0);
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)H * W, nwords = (npix + 63) / 64;
  hipLaunchKernelGGL(pack_refined_kernel, dim3((unsigned)((nwords + 3) / 4), n), dim3(256), 0, s,
                     (const uint8_t*)masks_u8, (const uint8_t*)sketch_rgb_u8, npix, nwords,
                     (unsigned long long*)bits_ws_u64);
  hipLaunchKernelGGL(pair_counts_kernel, dim3(n * n), dim3(256), 0, s, (const unsigned long long*)bits_ws_u64, n,
                     nwords, counts);
  return ink_launch_status();
}

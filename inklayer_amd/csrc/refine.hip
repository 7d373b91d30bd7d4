// Mask hand-off stage of InkLayer's refinement on resident masks: morphological cleanup of the SAM
// masks and the pairwise "sketch IoU" table of the sketch NMS.  Integer / byte work, HBM-bound; bit-exact against the
// reference's cv2 results (oracle/refine_ref.py, pinned by the reference's own committed outputs).
//   reference: InkLayer/refinement/mask_cleaner.py:11-36  (threshold -> MORPH_CLOSE with a k x k rect -> 8-connected
//              components with stats -> keep area > 500 or aspect ratio > 1.1)
//              InkLayer/refinement/nms_sketch.py:62-78,186-234 (masks AND stroke pixels, |A&B| / |A|B| per pair)
#include "bitplane.h"
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

// After a k x k closing every background gap of a row is at least (k+1)/2 pixels long, so a row has at most
// RM = W / ((k+1)/2 + 1) + 2 runs: that sizes the components workspace (an overflow flag guards it all the same).
static inline int cleanup_rm(int W, int k) { return W / ((k + 1) / 2 + 1) + 2; }

extern "C" int ink_mask_cleanup_workspace_ints(int32_t n, int32_t H, int32_t W, int32_t k, int64_t* out_ints) {
  INK_CHECK_ARG(out_ints && n > 0 && H > 0 && W > 0 && k >= 1 && k % 2 == 1);
  *out_ints = 1 + (int64_t)n * cc_ws_ints_per_plane(H, cleanup_rm(W, k));   // [0] = overflow flag, then n per-mask blocks
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
  // 8-connected components of the closed masks with area / aspect filter: the closed 0/1 bytes are packed into bit
  // planes (they live in the first scratch image, which is free again: 8 ceil(W/64) <= W bytes per row from W = 8 on)
  // and labelled by the multi-workgroup run-based passes of bitplane.h; the kept runs are painted straight into
  // out_u8 as 0 / 255.
  const int Wp = (W + 63) / 64;
  INK_CHECK_ARG((int64_t)Wp * 8 <= (int64_t)W);
  u64* planes = (u64*)a;
  INK_CHECK_ARG(((uintptr_t)planes & 7) == 0);
  hipLaunchKernelGGL(bp_pack_kernel, dim3((H * Wp + 3) / 4, n), dim3(256), 0, s, (const uint8_t*)b, H, W, Wp, 0, planes);
  if (hipMemsetAsync(workspace, 0, sizeof(int32_t), s) != hipSuccess) return INK_ERR_LAUNCH;
  return cc_run(planes, (int64_t)H * Wp, n, H, W, Wp, cleanup_rm(W, k), 1, area_threshold, aspect_threshold,
                workspace + 1, workspace, (uint8_t*)out_u8, nullptr, 0, s);
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

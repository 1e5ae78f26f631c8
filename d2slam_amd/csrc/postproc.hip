// postproc.hip -- SuperPoint post-processing kept on the device (gfx950).
//
// Replaces the CPU code the reference runs after copying the whole score and descriptor maps back to
// the host: SuperPoint::processOutput and helpers,
// d2frontend/src/CNN/superpoint_tensorrt.cpp:201-350, plus the in-graph tail of the network
// (softmax / dustbin drop / 8x8 depth-to-space, d2frontend/superpoint.ipynb:355-364).
// All kernels here are HBM/latency bound integer+fp32 work; the arithmetic follows the oracle
// (oracle/d2fe_oracle.c) operation by operation so that scores and indices compare exactly.
#include <algorithm>

#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fma-only expf: SAME operation sequence as orc_expf (oracle/d2fe_oracle.c) -> bitwise-equal results.
__device__ __forceinline__ float d2fe_expf(float x) {
  if (x < -87.0f) x = -87.0f;
  const float t = x * 1.44269504088896341f;
  const float n = __builtin_rintf(t);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = __builtin_fmaf(p, r, 1.3981999507E-3f);
  p = __builtin_fmaf(p, r, 8.3334519073E-3f);
  p = __builtin_fmaf(p, r, 4.1665795894E-2f);
  p = __builtin_fmaf(p, r, 1.6666665459E-1f);
  p = __builtin_fmaf(p, r, 5.0000001201E-1f);
  const float r2 = r * r;
  const float y = __builtin_fmaf(p, r2, r) + 1.0f;
  return __builtin_ldexpf(y, (int)n);
}

// -----------------------------------------------------------------------------------------------------
// softmax over 65 logits per 8x8 cell, sequential sum c = 0..64 (oracle order), scores e_c / s.
// Block = 256 threads = 64 consecutive cells x 4 lanes: the logits are staged through LDS with coalesced loads; the four
// lanes of a cell own the channel quarters [0,17) [17,33) [33,49) [49,65).  max is order-independent; the exps and the
// divisions are independent per channel; only the SUM has a prescribed order, and it is kept: the running sum is handed
// from quarter to quarter (3 shuffles), so the chain is still ((e0 + e1) + e2) + ... + e64 bit for bit.
// (First version: one lane per cell, 65 serial exps + 64 serial divisions per lane at 2 waves/SIMD: 69 us per 32 images.)
// Variant-B candidates (superpoint_tensorrt.cpp:201-230: score > thr, inside [border, dim-border)) are appended as
// keys (score_bits << 32) | (0xFFFFFFFF - raster_idx): descending key order == descending score, ties by ascending
// raster index (the oracle's tie-break).  The list order is irrelevant (keys are unique, select_b sorts).
// -----------------------------------------------------------------------------------------------------
// Bound by VALU issue, not by HBM (rocprofv3, 64 images: 2.0e7 wave-level VALU instructions -- 65 correctly rounded exponentials and divisions per cell, ~1030
// per wave -- are 32 us of the launch's 54 at the chip's full issue rate; the 80 MB of logits would take 13 us at 6 TB/s).  The exponential and the division are
// the oracle's (bitwise scores), so the instruction count is the arithmetic's.
__global__ __launch_bounds__(256) void softmax_cand_kernel(const float* __restrict__ logits, int lstride, int Hc, int Wc,
                                                           float thr, int border, float* __restrict__ semi,
                                                           unsigned long long* __restrict__ cand,
                                                           int* __restrict__ cand_count, long cand_cap) {
  __shared__ float sl[64 * 65];
  const int img = blockIdx.y;
  const int ncell = Hc * Wc;
  const int cell0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const float* lg = logits + ((size_t)img * ncell + cell0) * lstride;
  const int nvalid = min(64, ncell - cell0);
  if (lstride == 65) {   // the cells of a block are one contiguous run of floats
    for (int i = tid; i < nvalid * 65; i += 256) sl[i] = lg[i];
  } else {
    for (int i = tid; i < nvalid * 65; i += 256) {
      const int c = i / 65, k = i % 65;
      sl[c * 65 + k] = lg[(size_t)c * lstride + k];
    }
  }
  __syncthreads();
  const int cl = tid >> 2, q = tid & 3;
  const bool live = cl < nvalid;
  const int c0 = q == 0 ? 0 : 1 + 16 * q;          // 0, 17, 33, 49
  const int nc = q == 0 ? 17 : 16;
  const float* l = sl + (live ? cl : 0) * 65 + c0;
  float v[17];
  float m = l[0];
#pragma unroll
  for (int i = 0; i < 17; ++i) {
    v[i] = i < nc ? l[i] : l[0];
    m = v[i] > m ? v[i] : m;
  }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
#pragma unroll
  for (int i = 0; i < 17; ++i) v[i] = d2fe_expf(v[i] - m);
  float s = 0.f;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    if (q == qq) {
#pragma unroll
      for (int i = 0; i < 17; ++i)
        if (i < nc) s += v[i];
    }
    s = __shfl(s, (lane & ~3) | qq, 64);
  }
  const int cell = cell0 + cl;
  const int cy = cell / Wc, cx = cell % Wc;
  const int W = Wc * 8, H = Hc * 8;
  unsigned long long* cd = cand + (size_t)img * cand_cap;
  int mine = 0;
  unsigned pass_mask = 0;
#pragma unroll
  for (int i = 0; i < 17; ++i) {
    const int c = c0 + i;
    if (i < nc && c < 64) {
      const float p = v[i] / s;
      v[i] = p;
      const int y = cy * 8 + (c >> 3), x = cx * 8 + (c & 7);
      if (semi && live) semi[(size_t)img * H * W + y * W + x] = p;
      if (live && p > thr && y >= border && y < H - border && x >= border && x < W - border) { ++mine; pass_mask |= 1u << i; }
    }
  }
  // one reservation per BLOCK (same-address atomics serialise in L2 at ~5 ns each; measured): exclusive prefix of the
  // per-lane counts inside the wave, wave totals through LDS, a single atomic for the block's total
  __shared__ int wtot[4];
  __shared__ int s_base;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const int wave = tid >> 6;
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  if (tid == 0) {
    const int t = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    s_base = t > 0 ? atomicAdd(cand_count + img, t) : 0;
  }
  __syncthreads();
  int base = s_base;
#pragma unroll
  for (int w = 0; w < 3; ++w) base += w < wave ? wtot[w] : 0;
  int slot = base + incl - mine;
#pragma unroll
  for (int i = 0; i < 17; ++i) {
    if (pass_mask & (1u << i)) {
      const int c = c0 + i;
      const int y = cy * 8 + (c >> 3), x = cx * 8 + (c & 7);
      const unsigned idx = (unsigned)(y * W + x);
      if (slot < cand_cap)
        cd[slot] = ((unsigned long long)__float_as_uint(v[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
      ++slot;
    }
  }
}

hipError_t launch_softmax_cand(const float* logits, int lstride, int Hc, int Wc, int n_img, float thr, int border,
                               float* semi, unsigned long long* cand, int* cand_count, long cand_cap, bool zero_counts, hipStream_t s) {
  if (zero_counts) {      // false: the caller zeroed the counters together with the work counters at the start of the network pass
    hipError_t e = hipMemsetAsync(cand_count, 0, sizeof(int) * n_img, s);
    if (e != hipSuccess) return e;
  }
  dim3 grid((Hc * Wc + 63) / 64, n_img), block(256);
  hipLaunchKernelGGL(softmax_cand_kernel, grid, block, 0, s, logits, lstride, Hc, Wc, thr, border, semi, cand,
                     cand_count, cand_cap);
  return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------
// Variant-B selection (topKeypoints, superpoint_tensorrt.cpp:241-253): one 1024-thread block per image.
//   count <= K : keep all, in RASTER order (the reference does not sort in this case)
//   count >  K : the K largest keys, in descending key order (score desc, raster asc on ties)
// Exact selection by MSB-first radix select on the unique 64-bit keys, then an in-LDS bitonic sort.
// -----------------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAXSORT = 16384;     // keys the in-LDS bitonic sort takes (128 KiB of the 160 KiB LDS)

// sort_n: power of two in [1024, SEL_MAXSORT], >= min(K, SEL_MAXSORT) -- the size of the LDS key array (dynamic shared memory).
// semi/H/thr/border: the dense score map and the candidate predicate of softmax_cand_kernel; only read when everything is kept
// (count <= K, raster order) and the count exceeds sort_n: the keypoints then come from an ordered compaction of the map itself.
__global__ __launch_bounds__(SEL_THREADS) void select_b_kernel(const unsigned long long* __restrict__ cand,
                                                               const int* __restrict__ cand_count, long cand_cap,
                                                               int W, int max_kp, int cap, int always_sort, int sort_n,
                                                               const float* __restrict__ semi, int H, float thr, int border,
                                                               float* __restrict__ kps_xy,
                                                               float* __restrict__ scores, int32_t* __restrict__ kps_idx,
                                                               int32_t* __restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [sort_n]
  __shared__ int hist[256];
  __shared__ int s_digit, s_need, s_cnt;
  __shared__ int s_wcnt[SEL_THREADS / 64];
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const unsigned long long* c = cand + (size_t)img * cand_cap;
  long n = cand_count[img];
  if (n > cand_cap) n = cand_cap;
  int K = (max_kp < 0) ? cap : min(max_kp, cap);
  const bool take_all = (n <= K) && !always_sort;   // variant A sorts by confidence even when everything is kept
  if (take_all && n > sort_n && semi) {
    // keep-all with more keypoints than the LDS sort takes: raster order straight from the dense score map (same predicate as the
    // candidate list: score > thr, inside the borders), ordered compaction by ballot + prefix
    const float* sm = semi + (size_t)img * H * W;
    const int lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (long i0 = 0; i0 < (long)H * W; i0 += SEL_THREADS) {
      const long i = i0 + tid;
      bool ok = false;
      float p = 0.f;
      if (i < (long)H * W) {
        p = sm[i];
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        ok = p > thr && y >= border && y < H - border && x >= border && x < W - border;
      }
      const unsigned long long m = __ballot(ok);
      if (lane == 0) s_wcnt[wv] = __popcll(m);
      __syncthreads();
      int off = base, tot = 0;
      for (int w = 0; w < SEL_THREADS / 64; ++w) { if (w < wv) off += s_wcnt[w]; tot += s_wcnt[w]; }
      if (ok) {
        const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < cap) {
          const size_t o = (size_t)img * cap + slot;
          kps_xy[2 * o] = (float)(i % W); kps_xy[2 * o + 1] = (float)(i / W);
          scores[o] = p;
          if (kps_idx) kps_idx[o] = (int32_t)i;
        }
      }
      base += tot;
      __syncthreads();
    }
    if (tid == 0) n_out[img] = base < cap ? base : cap;
    return;
  }
  if (K > sort_n) K = sort_n;
  // When everything is kept the output order is raster (the reference does not sort): sort on the low word
  // (0xFFFFFFFF - idx, descending == idx ascending) and let the score bits ride along in the low half.
  unsigned long long thresh = 0;  // keep keys >= thresh
  int nk = (int)(n < K ? n : K);
  if (!(take_all && n <= sort_n) && n > K) {
    // radix select: find the K-th largest key
    unsigned long long prefix = 0, mask = 0;
    int need = K;
    for (int pass = 7; pass >= 0; --pass) {
      const int shift = pass * 8;
      for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0;
      __syncthreads();
      for (long i = tid; i < n; i += SEL_THREADS) {
        const unsigned long long k = c[i];
        if ((k & mask) == prefix) atomicAdd(&hist[(int)((k >> shift) & 0xFF)], 1);
      }
      __syncthreads();
      if (tid < 64) {
        // the digit d = the largest one whose suffix count S(d) = sum_{x >= d} hist[x] reaches `need` (0 if none does): lane l owns digits
        // 4l .. 4l+3, suffix sums by a 6-step shuffle scan -- what used to be a 256-iteration serial loop of one thread per pass
        const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        int above = h0 + h1 + h2 + h3;              // becomes the sum over the lanes ABOVE this one
        int incl = above;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(incl, o, 64); if (tid + o < 64) incl += t; }
        above = incl - above;
        const int S3 = above + h3, S2 = S3 + h2, S1 = S2 + h1, S0 = S1 + h0;
        int best = -1;                              // the largest digit of this lane with S >= need
        if (S3 >= need) best = 4 * tid + 3; else if (S2 >= need) best = 4 * tid + 2; else if (S1 >= need) best = 4 * tid + 1; else if (S0 >= need) best = 4 * tid;
        int gb = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gb = max(gb, __shfl_xor(gb, o, 64));
        const int d = gb < 0 ? 0 : gb;
        if (4 * tid <= d && d < 4 * tid + 4) {      // the lane that owns d publishes
          const int k = d - 4 * tid;
          const int Sd = k == 3 ? S3 : k == 2 ? S2 : k == 1 ? S1 : S0;      // S(d)
          const int hd = k == 3 ? h3 : k == 2 ? h2 : k == 1 ? h1 : h0;
          const int acc = Sd - hd;                  // S(d + 1): keys with a larger digit
          s_digit = d;
          s_need = need - acc;
          s_cnt = (K - need) + acc + hd;            // keys >= the prefix chosen so far
        }
      }
      __syncthreads();
      prefix |= ((unsigned long long)s_digit) << shift;
      mask |= 0xFFull << shift;
      need = s_need;
      const int above = s_cnt;
      __syncthreads();
      if (above <= sort_n) break;   // all of them fit in LDS: sort them there and keep the first K
    }
    thresh = prefix;  // at least K (and <= sort_n) keys are >= thresh; the sort below keeps the first K
  }
  const bool by_index = take_all && n <= sort_n;
  // gather survivors into LDS
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < sort_n; i += SEL_THREADS) keys[i] = 0;
  __syncthreads();
  for (long i = tid; i < n; i += SEL_THREADS) {
    const unsigned long long k = c[i];
    if (k >= thresh) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < sort_n) keys[slot] = by_index ? ((k << 32) | (k >> 32)) : k;
    }
  }
  __syncthreads();
  // bitonic sort, descending, sort_n elements (zeros sink to the end); only as many stages as the occupied power of two needs
  int used = s_cnt < sort_n ? s_cnt : sort_n;
  int sn = 1024;
  while (sn < used) sn <<= 1;
  for (int k2 = 2; k2 <= sn; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < sn; i += SEL_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          const bool desc = ((i & k2) == 0);
          if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  if (tid == 0) n_out[img] = nk;
  for (int t = tid; t < nk; t += SEL_THREADS) {
    unsigned long long k = keys[t];
    if (by_index) k = (k << 32) | (k >> 32);
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
    const float sc = __uint_as_float((unsigned)(k >> 32));
    const size_t o = (size_t)img * cap + t;
    kps_xy[2 * o] = (float)(idx % (unsigned)W);
    kps_xy[2 * o + 1] = (float)(idx / (unsigned)W);
    scores[o] = sc;
    if (kps_idx) kps_idx[o] = (int32_t)idx;
  }
}

hipError_t launch_select_b(const unsigned long long* cand, const int* cand_count, long cand_cap, int n_img, int W,
                           int max_kp, int cap, int always_sort, const float* semi, int H, float thr, int border, float* kps_xy,
                           float* scores, int32_t* kps_idx, int32_t* n_out, hipStream_t s) {
  int K = max_kp < 0 ? cap : (max_kp < cap ? max_kp : cap);
  if (K > SEL_MAXSORT) K = SEL_MAXSORT;
  int sort_n = 1024;
  while (sort_n < K) sort_n <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)sort_n;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned long long) * SEL_MAXSORT));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(select_b_kernel, dim3(n_img), dim3(SEL_THREADS), lds, s, cand, cand_count, cand_cap, W, max_kp, cap,
                     always_sort, sort_n, semi, H, thr, border, kps_xy, scores, kps_idx, n_out);
  return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------
// Variant-B descriptor sampling: normalize_keypoints (:255-265), grid_sample (:272-310) and
// normalize_descriptors (:312-317) of superpoint_tensorrt.cpp.  One wave per keypoint, 4 channels per lane
// (256 = 64 x 4): each corner is one coalesced 1 KiB read of the NHWC descriptor map.  The dense channel-L2
// normalisation of the network's `desc` output (superpoint.ipynb:352-353) is applied to the four corner
// cells on the fly, so the normalised 4.9 MB map is never written.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int clipi(int v, int mx) { return v < 0 ? 0 : (v < mx - 1 ? v : mx - 1); }

// The four corner cells and weights of one keypoint -- literal transcription of the reference's mixed float/double arithmetic
// (normalize_keypoints + grid_sample, superpoint_tensorrt.cpp:255-310; see orc_sample_b).  Shared by the sampler and by the
// kernel that marks which cells of the descriptor map are needed at all.
struct SampleBCorners { int ix[4], iy[4]; float w[4]; };     // order: nw, ne, sw, se
__device__ __forceinline__ SampleBCorners sample_b_corners(float x, float y, int Hc, int Wc) {
  const int s = 8;
  float k0 = (float)((double)(x - (float)(s / 2)) + 0.5);
  float k1 = (float)((double)(y - (float)(s / 2)) + 0.5);
  k0 = (float)((double)k0 / ((double)(Wc * s - s / 2) - 0.5));
  k1 = (float)((double)k1 / ((double)(Hc * s - s / 2) - 0.5));
  k0 = k0 * 2.0f - 1.0f;
  k1 = k1 * 2.0f - 1.0f;
  const float ix = ((k0 + 1.0f) / 2.0f) * (float)(Wc - 1);
  const float iy = ((k1 + 1.0f) / 2.0f) * (float)(Hc - 1);
  SampleBCorners c;
  c.ix[0] = clipi((int)__builtin_floorf(ix), Wc); c.iy[0] = clipi((int)__builtin_floorf(iy), Hc);
  c.ix[1] = clipi(c.ix[0] + 1, Wc); c.iy[1] = clipi(c.iy[0], Hc);
  c.ix[2] = clipi(c.ix[0], Wc);     c.iy[2] = clipi(c.iy[0] + 1, Hc);
  c.ix[3] = clipi(c.ix[0] + 1, Wc); c.iy[3] = clipi(c.iy[0] + 1, Hc);
  c.w[0] = ((float)c.ix[3] - ix) * ((float)c.iy[3] - iy);
  c.w[1] = (ix - (float)c.ix[2]) * ((float)c.iy[2] - iy);
  c.w[2] = ((float)c.ix[1] - ix) * (iy - (float)c.iy[1]);
  c.w[3] = (ix - (float)c.ix[0]) * (iy - (float)c.iy[0]);
  return c;
}

// slotmap == nullptr: desc_raw is the dense map [img][Hc*Wc][dstride] (+dcoff).  slotmap != nullptr: desc_raw is the SPARSE
// descriptor store [img][max_slots][256] and slotmap[img][cell] the slot of a cell (desc_head_sparse_kernel).
__global__ __launch_bounds__(256) void sample_b_kernel(const float* __restrict__ desc_raw, int dstride, int dcoff, int Hc,
                                                       int Wc, const float* __restrict__ kps_xy,
                                                       const int32_t* __restrict__ n_kp, int cap,
                                                       const int32_t* __restrict__ slotmap, int max_slots,
                                                       float* __restrict__ desc_out) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_kp[img] || k >= cap) return;
  const size_t o = (size_t)img * cap + k;
  const SampleBCorners c = sample_b_corners(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc);
  f32x4 v[4];
  float nrm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cell = c.iy[q] * Wc + c.ix[q];
    const float* row = slotmap ? desc_raw + ((size_t)img * max_slots + slotmap[(size_t)img * Hc * Wc + cell]) * 256
                               : desc_raw + ((size_t)img * Hc * Wc + cell) * dstride + dcoff;
    v[q] = *reinterpret_cast<const f32x4*>(row + lane * 4);
    nrm[q] = __builtin_sqrtf(wave_sum(v[q][0] * v[q][0] + v[q][1] * v[q][1] + v[q][2] * v[q][2] + v[q][3] * v[q][3]));
  }
  f32x4 d;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = (v[0][j] / nrm[0]) * c.w[0];
    t = t + (v[1][j] / nrm[1]) * c.w[1];
    t = t + (v[2][j] / nrm[2]) * c.w[2];
    t = t + (v[3][j] / nrm[3]) * c.w[3];
    d[j] = t;
    ss += t * t;
  }
  ss = wave_sum(ss);
  const float ninv = (float)(1.0 / (double)__builtin_sqrtf(ss));
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = d[j] * ninv;
  *reinterpret_cast<f32x4*>(desc_out + o * 256 + lane * 4) = d;
}

// -----------------------------------------------------------------------------------------------------
// Sparse descriptor head.  SuperPoint's descriptor branch (convDa 3x3 128->256 + ReLU, convDb 1x1 256->256: 3.46 of the
// 52.1 GFLOP per 640x480 image) is only ever READ at the <= 4 corner cells of each selected keypoint
// (superpoint_tensorrt.cpp:272-310), i.e. at <= 800 of 4800 cells for 200 keypoints.  The reference computes it densely because
// its network is one TensorRT graph; here the detector head runs first, the needed cells are marked and compacted, and the
// descriptor head is evaluated for those cells only -- same fmaf chains per output (bias, (ky,kx,ci) order), so the result is
// bit-identical to the dense map at every cell that is read.
//   desc_mark_kernel    : keypoints -> flags[img][cell]
//   desc_compact_kernel : flags -> slotmap[img][cell], cell list, count  (cells in raster order; one block per image)
//   desc_head_sparse_kernel : 32 cells x 256 channels per workgroup: convDa (A tile of one tap staged in LDS per step),
//                         ReLU, tile kept in LDS, convDb, raw descriptors to desc_sparse[img][slot][256]
// -----------------------------------------------------------------------------------------------------
// variant A corners (computeDescriptors / ATen grid_sampler, align_corners = false, zeros padding): only in-bounds cells exist
__device__ __forceinline__ void sample_a_origin(float x, float y, int Hc, int Wc, int img_w, int img_h, float& ix, float& iy) {
  const float gx = 2.0f * x / (float)img_w - 1.0f;
  const float gy = 2.0f * y / (float)img_h - 1.0f;
  ix = ((gx + 1.0f) * (float)Wc - 1.0f) / 2.0f;
  iy = ((gy + 1.0f) * (float)Hc - 1.0f) / 2.0f;
}

// img_w > 0 selects the variant-A corner rule
__global__ __launch_bounds__(256) void desc_mark_kernel(const float* __restrict__ kps_xy, const int32_t* __restrict__ n_kp, int cap,
                                                        int Hc, int Wc, int img_w, int img_h, uint8_t* __restrict__ flags) {
  const int img = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_kp[img] || k >= cap) return;
  const size_t o = (size_t)img * cap + k;
  uint8_t* f = flags + (size_t)img * Hc * Wc;
  if (img_w > 0) {
    float ix, iy;
    sample_a_origin(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc, img_w, img_h, ix, iy);
    const int x0 = (int)__builtin_floorf(ix), y0 = (int)__builtin_floorf(iy);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
      if (yy >= 0 && yy < Hc && xx >= 0 && xx < Wc) f[yy * Wc + xx] = 1;
    }
    return;
  }
  const SampleBCorners c = sample_b_corners(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc);
#pragma unroll
  for (int q = 0; q < 4; ++q) f[c.iy[q] * Wc + c.ix[q]] = 1;
}

__global__ __launch_bounds__(1024) void desc_compact_kernel(const uint8_t* __restrict__ flags, int ncell, int max_slots,
                                                            int32_t* __restrict__ slotmap, int32_t* __restrict__ cells,
                                                            int32_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < ncell; c0 += 1024) {
    const int cell = c0 + tid;
    const bool f = cell < ncell && flags[(size_t)img * ncell + cell];
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = wsum[k]; pre += k < wv ? c : 0; tot += c; }
    const int slot = s_base + pre + __popcll(bal & ((1ull << lane) - 1ull));
    if (cell < ncell) slotmap[(size_t)img * ncell + cell] = (f && slot < max_slots) ? slot : -1;
    if (f && slot < max_slots) cells[(size_t)img * max_slots + slot] = cell;
    __syncthreads();
    if (tid == 0) s_base += tot;
    __syncthreads();
  }
  if (tid == 0) count[img] = s_base < max_slots ? s_base : max_slots;
}

// desc_mark_kernel + desc_compact_kernel in ONE launch per image batch (one workgroup per image, the cell flags in LDS): the form every map of up to
// 64 Ki cells takes -- two launches and a memset less in front of the sparse head, which is what a one- or two-image call notices
__global__ __launch_bounds__(1024) void desc_mark_compact_kernel(const float* __restrict__ kps_xy, const int32_t* __restrict__ n_kp, int cap, int Hc, int Wc,
                                                                 int img_w, int img_h, int max_slots, int32_t* __restrict__ slotmap,
                                                                 int32_t* __restrict__ cells, int32_t* __restrict__ count) {
  extern __shared__ uint8_t mc_flags[];      // [ncell rounded up to 4]
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ncell = Hc * Wc;
  for (int i = tid; i < (ncell + 3) / 4; i += 1024) reinterpret_cast<uint32_t*>(mc_flags)[i] = 0u;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const int nk = min(n_kp[img], cap);
  for (int k = tid; k < nk; k += 1024) {
    const size_t o = (size_t)img * cap + k;
    if (img_w > 0) {
      float ix, iy;
      sample_a_origin(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc, img_w, img_h, ix, iy);
      const int x0 = (int)__builtin_floorf(ix), y0 = (int)__builtin_floorf(iy);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
        if (yy >= 0 && yy < Hc && xx >= 0 && xx < Wc) mc_flags[yy * Wc + xx] = 1;
      }
    } else {
      const SampleBCorners c = sample_b_corners(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc);
#pragma unroll
      for (int q = 0; q < 4; ++q) mc_flags[c.iy[q] * Wc + c.ix[q]] = 1;
    }
  }
  __syncthreads();
  for (int c0 = 0; c0 < ncell; c0 += 1024) {
    const int cell = c0 + tid;
    const bool f = cell < ncell && mc_flags[cell];
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = wsum[k]; pre += k < wv ? c : 0; tot += c; }
    const int slot = s_base + pre + __popcll(bal & ((1ull << lane) - 1ull));
    if (cell < ncell) slotmap[(size_t)img * ncell + cell] = (f && slot < max_slots) ? slot : -1;
    if (f && slot < max_slots) cells[(size_t)img * max_slots + slot] = cell;
    __syncthreads();
    if (tid == 0) s_base += tot;
    __syncthreads();
  }
  if (tid == 0) count[img] = s_base < max_slots ? s_base : max_slots;
}

struct SparseHeadArgs {
  const float* x; int x_cstride; long x_img_stride;      // conv4b output, NHWC [img][cell][128]
  const void* w_da; const float* b_da;                   // convDa packed fp32 fragments (8 n-tiles x 9 taps x 16 k-octets), bias[256]
  const void* w_db; const float* b_db;                   // convDb packed fp32 fragments (8 n-tiles x 1 tap x 32 k-octets), bias[256]
  const int32_t* cells; const int32_t* count; int max_slots;
  int Hc, Wc;
  float* out;                                            // [img][max_slots][256]
  float* mid;                                            // [img][max_slots][256]: ReLU(convDa) between the two stages of the split form (may be null: whole-head launches)
};

// MT: 32-cell row tiles per workgroup.  Every workgroup streams ALL of convDa's and convDb's weights (1.44 MB) from L2 once, whatever its number of cells:
// at a batch the launch is L2-bandwidth-bound on them (64 images x 25 workgroups x 1.44 MB = 2.3 GB per step at MT = 1), so a batch runs 64 cells per
// workgroup -- each weight fragment feeds two MFMAs.  A stereo pair (a few dozen workgroups on 256 CUs) is latency-bound instead and keeps MT = 1: half the
// MFMAs per wave.  The fmaf chain of an output does not depend on MT: bit-identical either way.
// NW / STAGE (round 5): a one- or two-image pass has ~20 tiles per image -- 42 workgroups of 8 waves, each CU-bound on 32 cells x (9 x 128 x 256 + 256 x 256)
// MACs = 38 us of matrix-pipe time with two waves per SIMD, 68 us measured, on a sixth of the chip.  STAGE 1 / 2 split the launch in two over TWICE the
// workgroups of FOUR waves (one per SIMD): blockIdx.z picks 128 of the 256 output channels; STAGE 1 = convDa + ReLU into `mid` [img][slot][256] (global, L2-resident),
// STAGE 2 = convDb from `mid`.  Every output keeps its chain (bias, taps, ci ascending): bit-identical to STAGE 0.
template <int MT, int NW, int STAGE>
__global__ __launch_bounds__(64 * NW) void desc_head_sparse_kernel(SparseHeadArgs a) {
  constexpr int CIN = 128, CPA = CIN + 1, CMID = 256, CPD = CMID + 1, G = 4, M = 32 * MT, NT = 64 * NW;
  static_assert((NW == 8 && STAGE == 0) || (NW == 4 && STAGE != 0 && MT == 1), "whole head: 8 waves x 32 channels; split stages: 4 waves per channel half");
  // A: one tap of the M cells' inputs [M][CPA]; D: ReLU(convDa) of the M cells = convDb's input [M][CPD] -- D overlays A (A is dead once the last tap is through)
  extern __shared__ __attribute__((aligned(16))) float sparse_lds[];       // [M * CPD] floats + [M] cell indices (MT = 2: 66 KB, dynamic)
  float* A = sparse_lds;
  float* D = sparse_lds;
  int* s_cell = reinterpret_cast<int*>(sparse_lds + M * CPD);
  const int img = blockIdx.y, mt = blockIdx.x;
  const int n = a.count[img];
  if (mt * M >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < M) s_cell[tid] = mt * M + tid < n ? a.cells[(size_t)img * a.max_slots + mt * M + tid] : -1;
  __syncthreads();
  const float* x = a.x + (size_t)img * a.x_img_stride;
  const f32x4* wda = reinterpret_cast<const f32x4*>(a.w_da);
  const f32x4* wdb = reinterpret_cast<const f32x4*>(a.w_db);
  const int nt0 = (STAGE == 0 ? 0 : (int)blockIdx.z * NW) + wave;      // this wave's 32 output channels
  float* mid = a.mid + ((size_t)img * a.max_slots + mt * M) * 256;     // STAGE 1 writes, STAGE 2 reads

  f32x16 acc[MT];
  if constexpr (STAGE != 2) {
    // ---- convDa: acc = bias; for tap (ky,kx): for ci ascending: fmaf -- the dense kernels' chain
    {
      const float b = a.b_da[nt0 * 32 + (lane & 31)];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = b;
    }
    const int arow = (lane & 31) * CPA + (lane >> 5);
    constexpr int NG = M * (CIN / 4) / NT;                  // float4 gathers per thread and tap
    for (int tap = 0; tap < 9; ++tap) {
      __syncthreads();                                      // previous tap's A tile fully consumed
      // gather: M cells x 32 float4, all loads issued before the first LDS store
      f32x4 gv[NG];
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int i = tid + NT * j;
        const int row = i / (CIN / 4), c4 = i % (CIN / 4);
        const int cell = s_cell[row];
        const int cy = cell / a.Wc + tap / 3 - 1, cx = cell % a.Wc + tap % 3 - 1;
        const bool ok = cell >= 0 && cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (ok ? ((size_t)cy * a.Wc + cx) * a.x_cstride + c4 * 4 : 0));
        gv[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int i = tid + NT * j;
        const int row = i / (CIN / 4), c4 = i % (CIN / 4);
        float* d = A + row * CPA + c4 * 4;
        d[0] = gv[j][0]; d[1] = gv[j][1]; d[2] = gv[j][2]; d[3] = gv[j][3];
      }
      __syncthreads();
#pragma unroll 1
      for (int g = 0; g < (CIN / 8) / G; ++g) {
        f32x4 bq[G];
#pragma unroll
        for (int j = 0; j < G; ++j) bq[j] = wda[((size_t)(nt0 * 9 + tap) * (CIN / 8) + g * G + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const float* ap = A + arow + (g * G + j) * 8;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m * 32 * CPA + 2 * q], bq[j][q], acc[m], 0, 0, 0);
        }
      }
    }
    if constexpr (STAGE == 1) {
      // ReLU -> mid (rows beyond the image's cell count are never read).  C layout: col = lane & 31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (cell)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[0][r];
        if (mt * M + row < n) mid[(size_t)row * 256 + nt0 * 32 + (lane & 31)] = v > 0.f ? v : 0.f;
      }
      return;
    }
    __syncthreads();                                        // everybody is through with A: D overlays it
    // ReLU -> D tile
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[m][r];
        D[row * CPD + nt0 * 32 + (lane & 31)] = v > 0.f ? v : 0.f;
      }
    __syncthreads();
  } else {
    // STAGE 2: the D tile from `mid` (rows beyond the count: zeros; their outputs are not stored)
    constexpr int ND = M * (CMID / 4) / NT;
    f32x4 dv[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int i = tid + NT * j;
      const int row = i / (CMID / 4), c4 = i % (CMID / 4);
      dv[j] = mt * M + row < n ? *reinterpret_cast<const f32x4*>(mid + (size_t)row * 256 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int i = tid + NT * j;
      const int row = i / (CMID / 4), c4 = i % (CMID / 4);
      float* d = D + row * CPD + c4 * 4;
      d[0] = dv[j][0]; d[1] = dv[j][1]; d[2] = dv[j][2]; d[3] = dv[j][3];
    }
    __syncthreads();
  }
  // ---- convDb (1x1): acc = bias; ci ascending
  {
    const float b = a.b_db[nt0 * 32 + (lane & 31)];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = b;
  }
  const int drow = (lane & 31) * CPD + (lane >> 5);
#pragma unroll 1
  for (int g = 0; g < (CMID / 8) / G; ++g) {
    f32x4 bq[G];
#pragma unroll
    for (int j = 0; j < G; ++j) bq[j] = wdb[((size_t)nt0 * (CMID / 8) + g * G + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const float* ap = D + drow + (g * G + j) * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m * 32 * CPD + 2 * q], bq[j][q], acc[m], 0, 0, 0);
    }
  }
  float* out = a.out + ((size_t)img * a.max_slots + mt * M) * 256;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (mt * M + row < n) out[(size_t)row * 256 + nt0 * 32 + (lane & 31)] = acc[m][r];
    }
}

hipError_t launch_desc_head_sparse(const float* kps_xy, const int32_t* n_kp, int cap, int Hc, int Wc, int n_img, const float* x,
                                   int x_cstride, long x_img_stride, const void* w_da, const float* b_da, const void* w_db,
                                   const float* b_db, uint8_t* flags, int32_t* slotmap, int32_t* cells, int32_t* count,
                                   int max_slots, float* out, float* mid, int mid_imgs, int img_w, int img_h, hipStream_t s) {
  const int ncell = Hc * Wc;
  hipError_t e = hipSuccess;
  if (ncell <= 60 * 1024) {
    hipLaunchKernelGGL(desc_mark_compact_kernel, dim3(n_img), dim3(1024), (size_t)(ncell + 3) / 4 * 4, s, kps_xy, n_kp, cap, Hc, Wc, img_w, img_h, max_slots, slotmap, cells, count);
  } else {
    e = hipMemsetAsync(flags, 0, (size_t)n_img * ncell, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(desc_mark_kernel, dim3((cap + 255) / 256, n_img), dim3(256), 0, s, kps_xy, n_kp, cap, Hc, Wc, img_w, img_h, flags);
    hipLaunchKernelGGL(desc_compact_kernel, dim3(n_img), dim3(1024), 0, s, flags, ncell, max_slots, slotmap, cells, count);
  }
  SparseHeadArgs a;
  a.x = x; a.x_cstride = x_cstride; a.x_img_stride = x_img_stride; a.w_da = w_da; a.b_da = b_da; a.w_db = w_db; a.b_db = b_db;
  a.cells = cells; a.count = count; a.max_slots = max_slots; a.Hc = Hc; a.Wc = Wc; a.out = out; a.mid = mid;
  const int slots = std::min(max_slots, 4 * cap);
  if (n_img >= 8) {
    constexpr size_t lds = sizeof(float) * (64 * 257 + 64);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(desc_head_sparse_kernel<2, 8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((desc_head_sparse_kernel<2, 8, 0>), dim3((slots + 63) / 64, n_img), dim3(512), lds, s, a);
  } else if (mid && n_img <= mid_imgs && n_img <= 4) {
    // a one- to four-image pass: the split form, twice the workgroups with one wave per SIMD (see the kernel)
    const dim3 grid((slots + 31) / 32, n_img, 2);
    hipLaunchKernelGGL((desc_head_sparse_kernel<1, 4, 1>), grid, dim3(256), sizeof(float) * (32 * 257 + 32), s, a);
    hipLaunchKernelGGL((desc_head_sparse_kernel<1, 4, 2>), grid, dim3(256), sizeof(float) * (32 * 257 + 32), s, a);
  } else {
    hipLaunchKernelGGL((desc_head_sparse_kernel<1, 8, 0>), dim3((slots + 31) / 32, n_img), dim3(512), sizeof(float) * (32 * 257 + 32), s, a);
  }
  return hipGetLastError();
}

hipError_t launch_sample_b(const float* desc_raw, int dstride, int dcoff, int Hc, int Wc, int n_img, const float* kps_xy,
                           const int32_t* n_kp, int cap, const int32_t* slotmap, int max_slots, float* desc_out, hipStream_t s) {
  dim3 grid((cap + 3) / 4, n_img), block(256);
  hipLaunchKernelGGL(sample_b_kernel, grid, block, 0, s, desc_raw, dstride, dcoff, Hc, Wc, kps_xy, n_kp, cap, slotmap, max_slots,
                     desc_out);
  return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------
// Variant A: getKeyPoints + NMS2 (d2frontend/src/CNN/superpoint_common.cpp:12-40,107-177).
// NMS2 is a SEQUENTIAL raster-order sweep, not greedy score-ordered NMS.  Its result is the unique
// solution of   active(P)  <=> no raster-earlier active Q within Chebyshev distance d with conf(Q) > conf(P)
//               survive(P) <=> active(P) and no raster-later active R within d with conf(R) > conf(P)
// (well-founded recursion over raster order), so it can be reached by chaotic in-place iteration from any start:
// after k sweeps the first k candidates in raster order are final.  One 1024-thread block per image iterates until a
// sweep changes nothing.  `aconf` holds conf for currently-active candidates and 0 elsewhere, so a neighbour test is
// one float compare.  Survivors are emitted as (conf, raster) keys for select_b_kernel (always_sort = 1).
// The reference's CV_16UC1 index map (:115,128: `inds` holds the candidate's raster rank modulo 65536, and the output point is
// pts_raw[inds(v,u)], :160-163) is reproduced by nms2_wrap_fix_kernel after the selection: above 65 536 candidates a survivor
// of rank r is reported at the coordinates of candidate r mod 65536, exactly as the reference (and the oracle) do.
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void nms2_a_kernel(const float* __restrict__ semi, int H, int W, float thr, int d,
                                                      float* __restrict__ aconf, int* __restrict__ clist,
                                                      unsigned long long* __restrict__ cand, int* __restrict__ cand_count,
                                                      long cand_cap, int* __restrict__ iters_out,
                                                      int* __restrict__ ncand_out) {
  __shared__ int s_n, s_changed;
  const int img = blockIdx.x, tid = threadIdx.x;
  const size_t hw = (size_t)H * W;
  const float* sm = semi + img * hw;
  float* ac = aconf + img * hw;
  int* cl = clist + img * hw;
  // candidate list in RASTER order (cv::findNonZero order, :17-19): wave w owns the contiguous pixel segment
  // [w*seg, (w+1)*seg); pass 1 counts per segment, pass 2 writes at the segment's prefix with ballot/mbcnt offsets.
  // The order itself does not matter to the fixpoint below; it is what lets nms2_wrap_fix_kernel reproduce the
  // reference's CV_16UC1 index map (rank -> pixel, pixel -> rank by binary search).
  __shared__ int s_wcnt[16];
  const int lane = tid & 63, wv = tid >> 6;
  const size_t seg = ((hw + 15) / 16 + 63) / 64 * 64;
  const size_t seg0 = wv * seg, seg1 = (seg0 + seg < hw) ? seg0 + seg : hw;
  int wcnt = 0;
  for (size_t i0 = seg0; i0 < seg1; i0 += 64) {
    const size_t i = i0 + lane;
    const float p = i < seg1 ? sm[i] : 0.f;
    const bool c = i < seg1 && p > thr;
    if (i < seg1) ac[i] = c ? p : 0.f;
    wcnt += __popcll(__ballot(c));
  }
  if (lane == 0) s_wcnt[wv] = wcnt;
  __syncthreads();
  int wbase = 0, n = 0;
  for (int w = 0; w < 16; ++w) { if (w < wv) wbase += s_wcnt[w]; n += s_wcnt[w]; }
  for (size_t i0 = seg0; i0 < seg1; i0 += 64) {
    const size_t i = i0 + lane;
    const bool c = i < seg1 && sm[i] > thr;          // the SAME predicate as pass 1 (a negative threshold admits p == 0, whose aconf is 0)
    const unsigned long long m = __ballot(c);
    if (c) cl[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (int)i;
    wbase += __popcll(m);
  }
  if (tid == 0) { s_n = n; if (ncand_out) ncand_out[img] = n; }
  __syncthreads();
  int iters = 0;
  for (;;) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int k = tid; k < n; k += 1024) {
      const int i = cl[k];
      const int x = i % W, y = i / W;
      const float p = sm[i];
      bool sup = false;
      for (int yy = max(0, y - d); yy <= y; ++yy) {
        const int x1 = (yy == y) ? x - 1 : min(W - 1, x + d);
        for (int xx = max(0, x - d); xx <= x1; ++xx) sup |= (ac[(size_t)yy * W + xx] > p);
      }
      const float want = sup ? 0.f : p;
      if (ac[i] != want) { ac[i] = want; s_changed = 1; }
    }
    __syncthreads();   // workgroup-scope release/acquire: the block's own global stores are visible to all its waves (shared L1)
    ++iters;
    const int ch = s_changed;
    __syncthreads();
    if (!ch) break;
  }
  if (tid == 0 && iters_out) iters_out[img] = iters;
  // survivors
  unsigned long long* cd = cand + (size_t)img * cand_cap;
  for (int k = tid; k < n; k += 1024) {
    const int i = cl[k];
    const float p = sm[i];
    if (ac[i] == 0.f) continue;
    const int x = i % W, y = i / W;
    bool dead = false;
    for (int yy = y; yy <= min(H - 1, y + d); ++yy) {
      const int x0 = (yy == y) ? x + 1 : max(0, x - d);
      for (int xx = x0; xx <= min(W - 1, x + d); ++xx) dead |= (ac[(size_t)yy * W + xx] > p);
    }
    if (!dead) {
      const int slot = atomicAdd(cand_count + img, 1);
      if (slot < cand_cap) cd[slot] = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
  }
}

hipError_t launch_nms2_a(const float* semi, int H, int W, int n_img, float thr, int dist, float* aconf, int* clist,
                         unsigned long long* cand, int* cand_count, long cand_cap, int* ncand, hipStream_t s) {
  hipError_t e = hipMemsetAsync(cand_count, 0, sizeof(int) * n_img, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nms2_a_kernel, dim3(n_img), dim3(1024), 0, s, semi, H, W, thr, dist, aconf, clist, cand, cand_count,
                     cand_cap, (int*)nullptr, ncand);
  return hipGetLastError();
}

// CV_16UC1 wrap of NMS2's index map (superpoint_common.cpp:115,128,160-163), applied to the selected keypoints of an image
// with more than 65 536 candidates: keypoint at pixel i (rank r in the raster-ordered candidate list) -> pixel clist[r & 0xFFFF].
__global__ __launch_bounds__(256) void nms2_wrap_fix_kernel(const int* __restrict__ clist, const int* __restrict__ ncand, int H, int W,
                                                            float* __restrict__ kps_xy, int32_t* __restrict__ kps_idx,
                                                            const int32_t* __restrict__ n_kp, int cap) {
  const int img = blockIdx.y;
  const int n = ncand[img];
  if (n <= 65536) return;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_kp[img] || k >= cap) return;
  const int* cl = clist + (size_t)img * H * W;
  float* kp = kps_xy + ((size_t)img * cap + k) * 2;
  const int i = (int)kp[1] * W + (int)kp[0];
  int lo = 0, hi = n;                       // lower_bound: the list is ascending in pixel index
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (cl[mid] < i) lo = mid + 1; else hi = mid; }
  const int j = cl[lo & 0xFFFF];
  kp[0] = (float)(j % W); kp[1] = (float)(j / W);
  if (kps_idx) kps_idx[(size_t)img * cap + k] = j;       // the raster index output follows the reported coordinates
}
hipError_t launch_nms2_wrap_fix(const int* clist, const int* ncand, int H, int W, int n_img, float* kps_xy, int32_t* kps_idx,
                                const int32_t* n_kp, int cap, hipStream_t s) {
  hipLaunchKernelGGL(nms2_wrap_fix_kernel, dim3((cap + 255) / 256, n_img), dim3(256), 0, s, clist, ncand, H, W, kps_xy, kps_idx, n_kp, cap);
  return hipGetLastError();
}

// Variant-A descriptor sampling: computeDescriptors (superpoint_common.cpp:42-99).  Three small kernels, because the reference
// normalises every CHANNEL over the image's keypoints before it normalises the rows (`torch::norm(desc, 2, 1)` on the
// [256, N] tensor, :68-69 -- see the oracle's note and tests/golden/reference_notebook.npz):
//   1. sample_a_kernel   : grid = 2*x/W - 1, torch::grid_sampler(bilinear, zeros padding, align_corners = false) -> S[k][256]
//   2. chan_norm_a_kernel: cn[c] = sqrt(sum_k S[k][c]^2), keypoints in list order (one thread per channel)
//   3. finish_a_kernel   : T = S / cn; optional PCA (T - mean) * comp^T; row L2.  One wave per keypoint.
// comp_t: [256][pca_dims] (transposed CSV layout, superpoint_onnx.cpp:47-53) or null.
__global__ __launch_bounds__(256) void sample_a_kernel(const float* __restrict__ desc_raw, int dstride, int dcoff, int Hc,
                                                       int Wc, int img_w, int img_h, const float* __restrict__ kps_xy,
                                                       const int32_t* __restrict__ n_kp, int cap, int scap,
                                                       const int32_t* __restrict__ slotmap, int max_slots,
                                                       float* __restrict__ samp) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wv;
  if (k >= n_kp[img] || k >= cap || k >= scap) return;
  const size_t o = (size_t)img * cap + k;
  float ix, iy;
  sample_a_origin(kps_xy[2 * o], kps_xy[2 * o + 1], Hc, Wc, img_w, img_h, ix, iy);
  const int x0 = (int)__builtin_floorf(ix), y0 = (int)__builtin_floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
  const float nw = ((float)x1 - ix) * ((float)y1 - iy);
  const float ne = (ix - (float)x0) * ((float)y1 - iy);
  const float sw = ((float)x1 - ix) * (iy - (float)y0);
  const float se = (ix - (float)x0) * (iy - (float)y0);
  const float* base = desc_raw + (size_t)img * Hc * Wc * dstride + dcoff + lane * 4;
  auto corner = [&](int yy, int xx, float wgt, f32x4& acc) {
    if (yy < 0 || yy >= Hc || xx < 0 || xx >= Wc) return;   // zeros padding (wave-uniform branch)
    const float* row = slotmap ? desc_raw + ((size_t)img * max_slots + slotmap[(size_t)img * Hc * Wc + yy * Wc + xx]) * 256 + lane * 4
                               : base + ((size_t)yy * Wc + xx) * dstride;
    const f32x4 v = *reinterpret_cast<const f32x4*>(row);
    const float nrm = __builtin_sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc[j] + (v[j] / nrm) * wgt;
  };
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  corner(y0, x0, nw, d); corner(y0, x1, ne, d); corner(y1, x0, sw, d); corner(y1, x1, se, d);
  *reinterpret_cast<f32x4*>(samp + ((size_t)img * scap + k) * 256 + lane * 4) = d;
}

__global__ __launch_bounds__(256) void chan_norm_a_kernel(const float* __restrict__ samp, const int32_t* __restrict__ n_kp, int cap,
                                                          int scap, float* __restrict__ cn) {
  const int img = blockIdx.x, c = threadIdx.x;
  const int n = min(min(n_kp[img], cap), scap);
  const float* s = samp + (size_t)img * scap * 256 + c;
  float ss = 0.f;
  for (int k = 0; k < n; ++k) { const float v = s[(size_t)k * 256]; ss += v * v; }
  cn[img * 256 + c] = __builtin_sqrtf(ss);
}

__global__ __launch_bounds__(256) void finish_a_kernel(const float* __restrict__ samp, const float* __restrict__ cn,
                                                       const int32_t* __restrict__ n_kp, int cap, int scap,
                                                       const float* __restrict__ comp_t, const float* __restrict__ mean,
                                                       int pca_dims, float* __restrict__ desc_out) {
  __shared__ float sd[4][256];
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wv;
  if (k >= n_kp[img] || k >= cap || k >= scap) return;
  const size_t o = (size_t)img * cap + k;
  f32x4 d = *reinterpret_cast<const f32x4*>(samp + ((size_t)img * scap + k) * 256 + lane * 4);
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(cn + img * 256 + lane * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = d[j] / c4[j];
  if (!comp_t) {
    const float ss = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    const float n2 = __builtin_sqrtf(ss);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = d[j] / n2;
    *reinterpret_cast<f32x4*>(desc_out + o * 256 + lane * 4) = d;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sd[wv][lane * 4 + j] = d[j] - mean[lane * 4 + j];
  __builtin_amdgcn_wave_barrier();
  // lane j (and j+64, ...) computes output component j
  for (int j0 = 0; j0 < pca_dims; j0 += 64) {
    const int j = j0 + lane;
    float a = 0.f;
    if (j < pca_dims)
      for (int c = 0; c < 256; ++c) a += sd[wv][c] * comp_t[(size_t)c * pca_dims + j];
    float s2 = wave_sum(j < pca_dims ? a * a : 0.f);
    // pca_dims <= 64 in every shipped config; for larger dims the norm needs all chunks: accumulate first
    if (pca_dims <= 64) {
      const float n2 = __builtin_sqrtf(s2);
      if (j < pca_dims) desc_out[o * pca_dims + j] = a / n2;
    } else {
      if (j < pca_dims) desc_out[o * pca_dims + j] = a;
    }
  }
  if (pca_dims > 64) {
    __builtin_amdgcn_wave_barrier();
    float s2 = 0.f;
    for (int j = lane; j < pca_dims; j += 64) { const float a = desc_out[o * pca_dims + j]; s2 += a * a; }
    s2 = wave_sum(s2);
    const float n2 = __builtin_sqrtf(s2);
    for (int j = lane; j < pca_dims; j += 64) desc_out[o * pca_dims + j] = desc_out[o * pca_dims + j] / n2;
  }
}

// samp: scratch [n_img][scap][256]; cn: scratch [n_img][256]
hipError_t launch_sample_a(const float* desc_raw, int dstride, int dcoff, int Hc, int Wc, int img_w, int img_h, int n_img,
                           const float* kps_xy, const int32_t* n_kp, int cap, const float* comp_t, const float* mean,
                           int pca_dims, float* samp, int scap, float* cn, const int32_t* slotmap, int max_slots, float* desc_out,
                           hipStream_t s) {
  const int kmax = cap < scap ? cap : scap;
  dim3 grid((kmax + 3) / 4, n_img), block(256);
  hipLaunchKernelGGL(sample_a_kernel, grid, block, 0, s, desc_raw, dstride, dcoff, Hc, Wc, img_w, img_h, kps_xy, n_kp, cap, scap, slotmap,
                     max_slots, samp);
  hipLaunchKernelGGL(chan_norm_a_kernel, dim3(n_img), block, 0, s, samp, n_kp, cap, scap, cn);
  hipLaunchKernelGGL(finish_a_kernel, grid, block, 0, s, samp, cn, n_kp, cap, scap, comp_t, mean, pca_dims, desc_out);
  return hipGetLastError();
}

}  // namespace d2fe

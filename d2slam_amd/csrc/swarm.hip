// swarm.hip -- the cross-agent exchange block and the NetVLAD gate.
//
// Reference analogue: every agent broadcasts its keyframe's {landmark positions, scores, descriptors, NetVLAD descriptor} over LCM
// (d2frontend/src/loop_net.cpp:24-87); a receiver tracks a remote frame only if the NetVLAD similarity with one of its own keyframes
// reaches track_remote_netvlad_thres (D2FeatureTracker::getMatchedPrevKeyframe, d2frontend/src/d2featuretracker.cpp:185-203) and then
// runs matchKNN on the descriptors (trackRemoteFrames, :237-310).  Here the broadcast is ONE all-gather of fixed-capacity blocks:
//
//   block (float words) = desc[cap][256] | kps[cap][2] | scores[cap] | netvlad[G] | n (int32) | zero padding to a multiple of 256
//
// descriptors first, block size a multiple of 256 words: a gathered block's descriptors are directly addressable by the batched
// matcher, whose offsets count rows of `dim` floats (d2fe_match_batch).
#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// NetVLAD rows inside a block start at word cap*259, 16-byte aligned only when cap is a multiple of 4: the gate kernels take a flag and
// fall back to four scalar loads when a base pointer or a stride is not (the C ABI computes it)
__device__ __forceinline__ f32x4 ld4(const float* p, bool vec) {
  if (vec) return *reinterpret_cast<const f32x4*>(p);
  return f32x4{p[0], p[1], p[2], p[3]};
}

// one workgroup per frame; `row0 + f * row_step` is the frame's row in the dense extract outputs ([rows][cap][...]),
// f its row in the NetVLAD output [F][G] (null: the block's NetVLAD part is zero-filled)
__global__ __launch_bounds__(256) void pack_blocks_kernel(const float* __restrict__ desc, const float* __restrict__ kps,
                                                          const float* __restrict__ scores, const int32_t* __restrict__ n_kp,
                                                          const float* __restrict__ gdesc, int row0, int row_step, int cap, int G,
                                                          int blk_words, float* __restrict__ blocks) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const int row = row0 + f * row_step;
  float* b = blocks + (size_t)f * blk_words;
  const int n = max(0, min(n_kp[row], cap));
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // descriptors: n valid rows, the rest zero (so that a block is a pure function of the frame, whatever the buffers held before)
  const f32x4* dsrc = reinterpret_cast<const f32x4*>(desc + (size_t)row * cap * 256);
  f32x4* ddst = reinterpret_cast<f32x4*>(b);
  for (int i = tid; i < cap * 64; i += 256) ddst[i] = (i >> 6) < n ? dsrc[i] : z;
  float* bk = b + (size_t)cap * 256;
  for (int i = tid; i < cap * 2; i += 256) bk[i] = (i >> 1) < n ? kps[(size_t)row * cap * 2 + i] : 0.f;
  float* bs = bk + cap * 2;
  for (int i = tid; i < cap; i += 256) bs[i] = i < n ? scores[(size_t)row * cap + i] : 0.f;
  float* bg = bs + cap;
  for (int i = tid; i < G; i += 256) bg[i] = gdesc ? gdesc[(size_t)f * G + i] : 0.f;
  const int used = cap * 259 + G;
  if (tid == 0) reinterpret_cast<int32_t*>(b)[used] = n;
  for (int i = used + 1 + tid; i < blk_words; i += 256) b[i] = 0.f;
}

// NetVLAD gate of a pair list: pair p passes iff dot(q[pair_q[p]], db[pair_db[p]]) >= thres (the reference rejects `< thres`,
// d2featuretracker.cpp:189-190; the dot product is accumulated in fp32 like Eigen's VectorXf::dot, the comparison is in double).
// pass[p] = 1/0; *n_pass += number passing; if cnt_inout != null a rejected pair's count is set to 0 (the batched matcher then
// returns no matches for it).  One wave per pair.
__global__ __launch_bounds__(256) void gate_pairs_kernel(const float* __restrict__ q, long q_stride, const float* __restrict__ db,
                                                         long db_stride, int dim, const int32_t* __restrict__ pair_q,
                                                         const int32_t* __restrict__ pair_db, int npairs, double thres,
                                                         int32_t* __restrict__ cnt_inout, int32_t* __restrict__ pass,
                                                         float* __restrict__ sims, int32_t* __restrict__ n_pass, bool vec) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= npairs) return;
  const float* a = q + (size_t)pair_q[p] * q_stride;
  const float* b = db + (size_t)pair_db[p] * db_stride;
  float s = 0.f;
  for (int j = lane * 4; j < dim; j += 256) {
    const f32x4 x = ld4(a + j, vec), y = ld4(b + j, vec);
#pragma unroll
    for (int e = 0; e < 4; ++e) s = __builtin_fmaf(x[e], y[e], s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    const bool ok = !((double)s < thres);
    if (pass) pass[p] = ok ? 1 : 0;
    if (sims) sims[p] = s;
    if (!ok && cnt_inout) cnt_inout[p] = 0;
    if (ok && n_pass) atomicAdd(n_pass, 1);
  }
}

// FOURCORNER_FISHEYE form of the gate: D2FeatureTracker::getMatchedPrevKeyframe (d2frontend/src/d2featuretracker.cpp:212-233) compares
// view 2 of the REMOTE quad frame (dir_a = 2) with the views dirs = {2, 3, 0, 1} of the local keyframe, in that order, and stops at the
// first whose NetVLAD similarity is not below the threshold: dir_b = dirs[j].  trackRemoteFrames (:282-297) then tracks the four view
// pairs (remote view dir_a = (2 + k) % 4, local view (dir_b - 2 + dir_a) % 4), k = 0..3 -- the relative yaw between the two drones, in
// quarter turns.  One wave per job = (local quad frame, remote quad frame): the four dot products (fp32, as Eigen's VectorXf::dot;
// compared in double), dir_prev[job] = dir_b or -1, sims[job][j] for dirs[j], and -- view pairs laid out as 16 consecutive matcher
// problems per job, index local_view * 4 + remote_view -- cnt_inout[job*16 + p] = 0 for every pair the reference would not track.
__global__ __launch_bounds__(256) void quad_gate_kernel(const float* __restrict__ loc, long loc_stride, const float* __restrict__ rem,
                                                        long rem_stride, int dim, const int32_t* __restrict__ job_loc_row0,
                                                        const int32_t* __restrict__ job_rem_row0, int loc_view_step, int rem_view_step,
                                                        int njobs, double thres, int32_t* __restrict__ dir_prev,
                                                        float* __restrict__ sims, int32_t* __restrict__ cnt_inout,
                                                        int32_t* __restrict__ n_pass, bool vec) {
  const int job = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (job >= njobs) return;
  const float* r2 = rem + ((size_t)job_rem_row0[job] + 2 * (size_t)rem_view_step) * rem_stride;
  const float* l0 = loc + (size_t)job_loc_row0[job] * loc_stride;
  float s[4] = {0.f, 0.f, 0.f, 0.f};      // s[j]: local view dirs[j] = (2 + j) & 3
  for (int e = lane * 4; e < dim; e += 256) {
    const f32x4 y = ld4(r2 + e, vec);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 x = ld4(l0 + (size_t)((2 + j) & 3) * loc_view_step * loc_stride + e, vec);
#pragma unroll
      for (int c = 0; c < 4; ++c) s[j] = __builtin_fmaf(x[c], y[c], s[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
  int db = -1;
#pragma unroll
  for (int j = 3; j >= 0; --j) if (!((double)s[j] < thres)) db = (2 + j) & 3;     // the FIRST j that passes wins
  if (lane == 0) {
    if (dir_prev) dir_prev[job] = db;
    if (sims) { sims[job * 4 + 0] = s[0]; sims[job * 4 + 1] = s[1]; sims[job * 4 + 2] = s[2]; sims[job * 4 + 3] = s[3]; }
    if (db >= 0 && n_pass) atomicAdd(n_pass, 1);
  }
  if (cnt_inout && lane < 16) {
    const int lv = lane >> 2, rv = lane & 3;
    const bool tracked = db >= 0 && lv == ((db - 2 + rv + 4) & 3);
    if (!tracked) cnt_inout[(size_t)job * 16 + lane] = 0;
  }
}

hipError_t launch_quad_gate(const float* loc, long loc_stride, const float* rem, long rem_stride, int dim, const int32_t* job_loc_row0,
                            const int32_t* job_rem_row0, int loc_view_step, int rem_view_step, int njobs, double thres, int32_t* dir_prev,
                            float* sims, int32_t* cnt_inout, int32_t* n_pass, hipStream_t s) {
  const bool vec = !(((uintptr_t)loc | (uintptr_t)rem) & 15) && !((loc_stride | rem_stride) & 3);
  hipLaunchKernelGGL(quad_gate_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, loc, loc_stride, rem, rem_stride, dim, job_loc_row0,
                     job_rem_row0, loc_view_step, rem_view_step, njobs, thres, dir_prev, sims, cnt_inout, n_pass, vec);
  return hipGetLastError();
}

// ---- int8 wire format of the exchange block ------------------------------------------------------------------------------------------
// The reference ships descriptors as int8 (VisualImageDesc::toLCM, d2common/include/d2common/d2frontend_types.h:228-237: q = (int8)(x / max|x| * 127)
// with ONE float maximum over the frame's whole landmark_descriptor vector; :260-268 the NetVLAD vector with a double maximum; scores are not
// sent) and decodes them in the LCM constructor (:319-338: x = q / 127.0, then `desc0.segment(i*32, 32).normalize()` for i < landmark_num --
// the hard-coded 32: only the first landmark_num*32 floats are re-normalised, in 32-float pieces -- and the NetVLAD vector normalised as a whole).
// int8 block (bytes) = desc_q[cap][256] | netvlad_q[G] | kps f32[cap][2] | n int32 | zero padding to a multiple of 64: 3.9x smaller than the fp32
// block.  pack = the reference's quantisation; unpack expands a gathered int8 block into the fp32 block layout (scores = 0) with the
// reference's decode arithmetic, so that the gate and the matcher see exactly the descriptors a receiving agent of the reference would hold.
__global__ __launch_bounds__(256) void pack_blocks_int8_kernel(const float* __restrict__ desc, const float* __restrict__ kps,
                                                               const int32_t* __restrict__ n_kp, const float* __restrict__ gdesc, int row0,
                                                               int row_step, int cap, int G, int blk_bytes, int8_t* __restrict__ blocks) {
  __shared__ float red[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = row0 + f * row_step;
  int8_t* b = blocks + (size_t)f * blk_bytes;
  const int n = max(0, min(n_kp[row], cap));
  const f32x4* src = reinterpret_cast<const f32x4*>(desc + (size_t)row * cap * 256);
  float m = 0.f;
  for (int i = tid; i < n * 64; i += 256) { const f32x4 v = src[i]; m = fmaxf(fmaxf(m, __builtin_fabsf(v[0])), fmaxf(__builtin_fabsf(v[1]), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])))); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  uint32_t* dq = reinterpret_cast<uint32_t*>(b);
  for (int i = tid; i < cap * 64; i += 256) {
    uint32_t w = 0;
    if (i < n * 64) {
      const f32x4 v = src[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) w |= (uint32_t)(uint8_t)(int8_t)(int)(m > 0.f ? v[e] / m * 127.0f : 0.f) << (8 * e);      // C cast: truncation toward zero
                                                                                                         // (an all-zero frame: 0/0 in the reference; zeros here)
    }
    dq[i] = w;
  }
  __syncthreads();
  // NetVLAD: `double max` (d2frontend_types.h:265): x / max * 127 evaluated in double
  float gm = 0.f;
  if (gdesc) for (int i = tid; i < G; i += 256) gm = fmaxf(gm, __builtin_fabsf(gdesc[(size_t)f * G + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
  if (lane == 0) red[wv] = gm;
  __syncthreads();
  const double gmd = (double)fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int8_t* gq = b + (size_t)cap * 256;
  for (int i = tid; i < G; i += 256) gq[i] = (gdesc && gmd > 0.0) ? (int8_t)(int)((double)gdesc[(size_t)f * G + i] / gmd * 127.0) : (int8_t)0;
  float* bk = reinterpret_cast<float*>(gq + G);
  for (int i = tid; i < cap * 2; i += 256) bk[i] = (i >> 1) < n ? kps[(size_t)row * cap * 2 + i] : 0.f;
  int32_t* bn = reinterpret_cast<int32_t*>(bk + cap * 2);
  if (tid == 0) *bn = n;
  const int used = cap * 256 + G + cap * 8 + 4;
  for (int i = used + tid; i < blk_bytes; i += 256) b[i] = 0;
}

// renorm 0 = the reference's decode (32-float segments, the first n of them); 1 = every descriptor row re-normalised over its 256 floats
__global__ __launch_bounds__(256) void unpack_blocks_int8_kernel(const int8_t* __restrict__ blocks, int cap, int G, int blk_bytes, int blk_words,
                                                                 int renorm, float* __restrict__ out) {
  __shared__ float red[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int8_t* b = blocks + (size_t)f * blk_bytes;
  float* o = out + (size_t)f * blk_words;
  const int8_t* gq = b + (size_t)cap * 256;
  const float* bk = reinterpret_cast<const float*>(gq + G);
  const int n = max(0, min(*reinterpret_cast<const int32_t*>(bk + cap * 2), cap));
  const uint32_t* dq = reinterpret_cast<const uint32_t*>(b);
  f32x4* od = reinterpret_cast<f32x4*>(o);
  // thread i holds 4 consecutive values; 8 consecutive threads = one 32-float segment, 64 = one descriptor row
  for (int i0 = 0; i0 < cap * 64; i0 += 256) {
    const int i = i0 + tid;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n * 64) {
      const uint32_t w = dq[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)((double)(int8_t)(w >> (8 * e)) / 127.0);
    }
    float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    const int nsh = renorm ? 32 : 4;
#pragma unroll
    for (int sh = 1; sh <= 32; sh <<= 1) if (sh <= nsh) s += __shfl_xor(s, sh, 64);
    const bool norm = renorm ? (i < n * 64) : ((i >> 3) < n);        // reference: segment index < landmark_num (= n)
    if (norm && s > 0.f) { const float r = __builtin_sqrtf(s); v[0] = v[0] / r; v[1] = v[1] / r; v[2] = v[2] / r; v[3] = v[3] / r; }
    if (i < cap * 64) od[i] = v;
  }
  float* ok = o + (size_t)cap * 256;
  for (int i = tid; i < cap * 2; i += 256) ok[i] = bk[i];
  float* os = ok + cap * 2;
  for (int i = tid; i < cap; i += 256) os[i] = 0.f;                  // scores are not on the wire (d2frontend_types.h:229 "Not send scores currently")
  float* og = os + cap;
  float s = 0.f;
  for (int i = tid; i < G; i += 256) { const float v = (float)((double)gq[i] / 127.0); s += v * v; }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float z = (red[0] + red[1]) + (red[2] + red[3]);
  const float r = __builtin_sqrtf(z);
  for (int i = tid; i < G; i += 256) { const float v = (float)((double)gq[i] / 127.0); og[i] = z > 0.f ? v / r : v; }
  const int used = cap * 259 + G;
  if (tid == 0) reinterpret_cast<int32_t*>(o)[used] = n;
  for (int i = used + 1 + tid; i < blk_words; i += 256) o[i] = 0.f;
}

hipError_t launch_pack_blocks_int8(const float* desc, const float* kps, const int32_t* n_kp, const float* gdesc, int row0, int row_step,
                                   int nframes, int cap, int G, int blk_bytes, int8_t* blocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_blocks_int8_kernel, dim3(nframes), dim3(256), 0, s, desc, kps, n_kp, gdesc, row0, row_step, cap, G, blk_bytes, blocks);
  return hipGetLastError();
}
hipError_t launch_unpack_blocks_int8(const int8_t* blocks, int nblocks, int cap, int G, int blk_bytes, int blk_words, int renorm, float* out,
                                     hipStream_t s) {
  hipLaunchKernelGGL(unpack_blocks_int8_kernel, dim3(nblocks), dim3(256), 0, s, blocks, cap, G, blk_bytes, blk_words, renorm, out);
  return hipGetLastError();
}

hipError_t launch_pack_blocks(const float* desc, const float* kps, const float* scores, const int32_t* n_kp, const float* gdesc,
                              int row0, int row_step, int nframes, int cap, int G, int blk_words, float* blocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_blocks_kernel, dim3(nframes), dim3(256), 0, s, desc, kps, scores, n_kp, gdesc, row0, row_step, cap, G, blk_words, blocks);
  return hipGetLastError();
}
hipError_t launch_gate_pairs(const float* q, long q_stride, const float* db, long db_stride, int dim, const int32_t* pair_q,
                             const int32_t* pair_db, int npairs, double thres, int32_t* cnt_inout, int32_t* pass, float* sims,
                             int32_t* n_pass, hipStream_t s) {
  const bool vec = !(((uintptr_t)q | (uintptr_t)db) & 15) && !((q_stride | db_stride) & 3);
  hipLaunchKernelGGL(gate_pairs_kernel, dim3((npairs + 3) / 4), dim3(256), 0, s, q, q_stride, db, db_stride, dim, pair_q, pair_db, npairs,
                     thres, cnt_inout, pass, sims, n_pass, vec);
  return hipGetLastError();
}

// ---- quadcam neighbour matching (A12): getFeatureHalfImg + the +-move_cols shift, and the index remap, on the device --------------------
// Reference: getFeatureHalfImg (d2frontend/src/d2featuretracker.cpp:1051-1075) keeps the keypoints of one half of the undistorted view
// (x < W_u - move_cols for the left set, x >= move_cols for the right set, move_cols = W_u * 90 / fov as a float) and copies their
// descriptors; matchLocalFeatures (:1161-1170) shifts the a-side x by +-move_cols before the radius gate and (:1178-1181) maps the
// match indices back.  One workgroup per job: ordered compaction (ballot + prefix over <= 1024 points), descriptor rows copied as
// float4, `map[c] = original index`.
__global__ __launch_bounds__(256) void half_compact_kernel(const float* __restrict__ desc, const float* __restrict__ pts,
                                                           const int32_t* __restrict__ n_kp, const int32_t* __restrict__ job_row,
                                                           const int32_t* __restrict__ job_left, const float* __restrict__ job_shift,
                                                           int cap, int dim, float width_undistort, float move_cols,
                                                           float* __restrict__ out_desc, float* __restrict__ out_pts,
                                                           int32_t* __restrict__ out_map, int32_t* __restrict__ out_n) {
  __shared__ int s_wcnt[4], s_base;
  __shared__ int s_src[1024];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = job_row[j];
  const bool left = job_left[j] != 0;
  const float shift = job_shift[j];
  const int n = max(0, min(n_kp[row], cap));
  const float* p = pts + (size_t)row * cap * 2;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    bool keep = false;
    if (i < n) { const float x = p[2 * i]; keep = left ? (x < width_undistort - move_cols) : (x >= move_cols); }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wv; ++w) off += s_wcnt[w];
    if (keep) {
      const int c = off + __popcll(m & ((1ull << lane) - 1ull));
      s_src[c] = i;
      out_map[(size_t)j * cap + c] = i;
      out_pts[((size_t)j * cap + c) * 2] = p[2 * i] + shift;
      out_pts[((size_t)j * cap + c) * 2 + 1] = p[2 * i + 1];
    }
    __syncthreads();
    if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  const int c_n = s_base;
  if (tid == 0) out_n[j] = c_n;
  const int d4 = dim >> 2;
  const f32x4* src = reinterpret_cast<const f32x4*>(desc + (size_t)row * cap * dim);
  f32x4* dst = reinterpret_cast<f32x4*>(out_desc + (size_t)j * cap * dim);
  for (int i = tid; i < c_n * d4; i += 256) { const int c = i / d4, q = i - c * d4; dst[i] = src[(size_t)s_src[c] * d4 + q]; }
}

__global__ __launch_bounds__(256) void remap_matches_kernel(int32_t* __restrict__ q_idx, int32_t* __restrict__ t_idx,
                                                            const int32_t* __restrict__ n_match, const int32_t* __restrict__ map_a_job,
                                                            const int32_t* __restrict__ map_b_job, const int32_t* __restrict__ maps,
                                                            int cap_match, int cap_map) {
  const int p = blockIdx.x;
  const int n = min(n_match[p], cap_match);
  const int32_t* ma = maps + (size_t)map_a_job[p] * cap_map;
  const int32_t* mb = maps + (size_t)map_b_job[p] * cap_map;
  for (int i = threadIdx.x; i < n; i += 256) {
    q_idx[(size_t)p * cap_match + i] = ma[q_idx[(size_t)p * cap_match + i]];
    t_idx[(size_t)p * cap_match + i] = mb[t_idx[(size_t)p * cap_match + i]];
  }
}

hipError_t launch_half_compact(const float* desc, const float* pts, const int32_t* n_kp, const int32_t* job_row, const int32_t* job_left,
                               const float* job_shift, int njobs, int cap, int dim, float width_undistort, float move_cols,
                               float* out_desc, float* out_pts, int32_t* out_map, int32_t* out_n, hipStream_t s) {
  hipLaunchKernelGGL(half_compact_kernel, dim3(njobs), dim3(256), 0, s, desc, pts, n_kp, job_row, job_left, job_shift, cap, dim, width_undistort,
                     move_cols, out_desc, out_pts, out_map, out_n);
  return hipGetLastError();
}
hipError_t launch_remap_matches(int32_t* q_idx, int32_t* t_idx, const int32_t* n_match, const int32_t* map_a_job, const int32_t* map_b_job,
                                const int32_t* maps, int npairs, int cap_match, int cap_map, hipStream_t s) {
  hipLaunchKernelGGL(remap_matches_kernel, dim3(npairs), dim3(256), 0, s, q_idx, t_idx, n_match, map_a_job, map_b_job, maps, cap_match, cap_map);
  return hipGetLastError();
}

}  // namespace d2fe

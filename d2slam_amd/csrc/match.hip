// match.hip -- brute-force descriptor matching on gfx950: matchKNN and the cross-check matcher.
//
// Replaces D2FrontEnd::matchKNN (d2frontend/src/feature_matcher.cpp:4-42: cv::BFMatcher(NORM_L2).knnMatch
// both ways, Lowe ratio, mutual check, optional pixel-radius gate) and cv::BFMatcher(NORM_L2, true).match
// (loop_cam.cpp:167-170, d2featuretracker.cpp:1141-1142).
//
// Two kernels per batch of pairs:
//  (1) match_prefilter: per (pair, direction, 32-query tile) the Gram tile  T . Q^T  on fp32 MFMA
//      (v_mfma_f32_32x32x2_f32), d2 = |t|^2 + |q|^2 - 2 t.q, and a per-query top-4 of candidate train indices
//      kept in registers (queries sit on the MFMA column axis, so the scan over a lane's 16 train rows is
//      lane-local; halves and waves merge through shuffles / LDS).
//  (2) match_finalize: per pair, re-evaluates the <= 4 candidates of every row of both directions with the
//      ORACLE's arithmetic (orc_l2_dist: OpenCV normL2Sqr_ accumulation order, then sqrt), takes the exact
//      2-NN, applies ratio / mutual / radius tests in double exactly as feature_matcher.cpp:16-37, and emits
//      matches in ascending query order.
// Exact by construction: candidates are chosen on the Gram-trick distance, whose error against the exact one is bounded by
// GRAM_ERR * (|q|^2 + |t|^2).  Eight candidates per query are tracked; the first four are re-ranked exactly.  A row outside the
// first four has an approximate d2 >= the 4th candidate's: if that bound cannot rule out that such a row beats the exact 2nd
// neighbour, the query is SATURATED at level 1 (about 1 % of the queries on SuperPoint descriptors, whose 2nd..4th neighbours lie
// ~3e-3 apart in d2) and candidates 5..8 are re-ranked exactly as well, now against the 8th candidate's bound; only if that fails
// too (more than eight rows within round-off of each other: repeated texture, a frame matched against a near-copy, all-equal sets)
// is the 2-NN recomputed by an exact scan of every train row.  Indices and distances equal the oracle's bit for bit for any input.
#include "kernels.h"

namespace d2fe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MQ = 32;         // queries per block
constexpr int TB = 128;        // train rows per staged block (4 waves x 32)
constexpr int KCH = 64;        // K chunk staged per pass (64: 67 KB of LDS per block -> two blocks per CU)
constexpr int QS = 257;        // LDS row stride of the query tile (odd -> conflict-free column reads)
constexpr int TS = KCH + 1;    // LDS row stride of the train chunk
constexpr int MAXDIM = 256;
// |d2_gram - d2_exact| <= GRAM_ERR * (|q|^2 + |t|^2): three fp32 sums of <= 256 products (gamma_257 = 257 * 2^-24 = 1.53e-5 each, and
// sum |t_k q_k| <= (|q|^2 + |t|^2) / 2) give 2 * gamma_257 = 3.1e-5; 4e-5 leaves room for the final subtraction's rounding
constexpr float GRAM_ERR = 4.0e-5f;

struct Cand { float d; int i; };
__device__ __forceinline__ bool cand_less(float d, int i, const Cand& c) { return d < c.d || (d == c.d && i < c.i); }
constexpr int NC = 8;          // candidates tracked per query (the first four are always re-ranked, the rest on demand)
__device__ __forceinline__ void cand_insert(Cand (&top)[NC], float d, int i) {
  if (!cand_less(d, i, top[NC - 1])) return;
  top[NC - 1].d = d; top[NC - 1].i = i;
#pragma unroll
  for (int k = NC - 1; k > 0; --k) {
    if (cand_less(top[k].d, top[k].i, top[k - 1])) {
      const Cand t = top[k]; top[k] = top[k - 1]; top[k - 1] = t;
    }
  }
}

// exact distance of one (q, t) row pair with 16 lanes: lane slot s accumulates elements j = 16*i + s
// (slot s = 4*v + l of OpenCV's four 4-lane accumulators), then the oracle's reduction order.
__device__ __forceinline__ float exact_dist16(const float* __restrict__ q, const float* __restrict__ t, int dim, int slot,
                                              int lane) {
  float acc = 0.f;
  const int nfull = dim & ~15;
  // the train row comes from global memory: eight loads are issued before the first use (one L2 round trip per batch instead
  // of one per element); the accumulation itself stays in ascending j, the oracle's order
  for (int j0 = slot; j0 < nfull; j0 += 128) {
    float tv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) tv[u] = j0 + 16 * u < nfull ? t[j0 + 16 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + 16 * u < nfull) {
        const float d = q[j0 + 16 * u] - tv[u];
        const float dd = d * d;
        acc = acc + dd;
      }
  }
  // r[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l]   with slot = 4*v + l
  const int base = lane & ~15;
  const int l = slot & 3;
  const float a0 = __shfl(acc, base + 0 + l, 64), a1 = __shfl(acc, base + 4 + l, 64);
  const float a2 = __shfl(acc, base + 8 + l, 64), a3 = __shfl(acc, base + 12 + l, 64);
  const float r = ((a0 + a1) + a2) + a3;  // valid in every lane for its l
  const float r0 = __shfl(r, base + 0, 64), r1 = __shfl(r, base + 1, 64);
  const float r2 = __shfl(r, base + 2, 64), r3 = __shfl(r, base + 3, 64);
  float d = (r0 + r2) + (r1 + r3);
  for (int j = nfull; j < dim; ++j) { const float e = q[j] - t[j]; d += e * e; }
  return __builtin_sqrtf(d);
}

__global__ __launch_bounds__(256) void match_prefilter_kernel(MatchArgs m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                       // [MQ][QS]
  float* Ts = Qs + MQ * QS;               // [TB][TS]
  float* qn = Ts + TB * TS;               // [MQ]
  float* tn = qn + MQ;                    // [TB]
  Cand* merge = reinterpret_cast<Cand*>(tn + TB);  // [4 waves][MQ][NC]
  float* aux = reinterpret_cast<float*>(merge + 4 * MQ * NC);  // [4] per-wave max |t|^2
  float* m4th = aux + 4;                                        // [MQ] approximate d2 of the 4th candidate
  float* m8th = m4th + MQ;                                      // [MQ] ... of the 8th
  int* mcand2 = reinterpret_cast<int*>(m8th + MQ);              // [MQ][4] candidates 5..8
  float* mdist2 = reinterpret_cast<float*>(mcand2 + MQ * 4);    // [MQ][4] their exact distances (computed for saturated queries only)
  int* satn = reinterpret_cast<int*>(mdist2 + MQ * 4);          // [2]: saturated at level 1 / still saturated after level 2
  int* sat = satn + 2;                                          // [MQ] tile-local rows saturated at level 1
  int* sat2 = sat + MQ;                                         // [MQ] ... at level 2: exact scan
  Cand* scan = reinterpret_cast<Cand*>(sat2 + MQ + 2);          // [16 groups][2] partial 2-NN of the exact scan (8-byte aligned)

  const int pair = blockIdx.z, dir = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int na = min(m.a_cnt[pair], m.max_n), nb = min(m.b_cnt[pair], m.max_n);
  const int nq = dir == 0 ? na : nb, nt = dir == 0 ? nb : na;
  const int q0 = blockIdx.x * MQ;
  if (q0 >= nq || q0 >= m.max_n) return;
  const float* Q = dir == 0 ? m.a + (size_t)m.a_off[pair] * m.dim : m.b + (size_t)m.b_off[pair] * m.dim;
  const float* T = dir == 0 ? m.b + (size_t)m.b_off[pair] * m.dim : m.a + (size_t)m.a_off[pair] * m.dim;
  const int dim = m.dim;
  const int d4 = dim / 4;

  // stage the query tile (zero padded), coalesced float4 reads
  for (int i = tid; i < MQ * (MAXDIM / 4); i += 256) {
    const int r = i / (MAXDIM / 4), c4 = i % (MAXDIM / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q0 + r < nq && c4 < d4) v = *reinterpret_cast<const f32x4*>(Q + (size_t)(q0 + r) * dim + c4 * 4);
    float* d = Qs + r * QS + c4 * 4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  __syncthreads();
  {   // |q|^2 with 8 threads per row (the prefilter distance is approximate by design: any summation order will do)
    const int r = tid >> 3, part = tid & 7;
    float s = 0.f;
    for (int k = part; k < dim; k += 8) s = __builtin_fmaf(Qs[r * QS + k], Qs[r * QS + k], s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) qn[r] = s;
  }

  Cand top[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { top[k].d = __builtin_inff(); top[k].i = 0x7FFFFFFF; }

  const int nkc = (dim + KCH - 1) / KCH;
  // train chunks (TB rows x KCH columns) go global -> registers -> LDS; the loads of chunk c + 1 are in flight during the norms and the
  // MFMAs of chunk c (the kernel is latency-bound: staged synchronously, every chunk exposed one L2 round trip)
  constexpr int NLD = TB * (KCH / 4) / 256;
  static_assert(TB * (KCH / 4) % 256 == 0, "staging assumes whole passes");
  f32x4 stg[NLD];
  auto gload = [&](int c) {
    const int t0 = (c / nkc) * TB, kc = c % nkc;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + 256 * u;
      const int r = i / (KCH / 4), c4 = i % (KCH / 4);
      const int col = kc * KCH + c4 * 4;
      stg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t0 + r < nt && col < dim) stg[u] = *reinterpret_cast<const f32x4*>(T + (size_t)(t0 + r) * dim + col);
    }
  };
  const int nchunks = ((nt + TB - 1) / TB) * nkc;
  if (nchunks > 0) gload(0);
  f32x16 acc;
  float tnorm = 0.f;
  float tnmax = 0.f;                     // largest |t|^2 this lane has seen (the error bound of the rows it dropped)
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = (c / nkc) * TB, kc = c % nkc;
    if (kc == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      tnorm = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + 256 * u;
      const int r = i / (KCH / 4), c4 = i % (KCH / 4);
      float* d = Ts + r * TS + c4 * 4;
      d[0] = stg[u][0]; d[1] = stg[u][1]; d[2] = stg[u][2]; d[3] = stg[u][3];
    }
    __syncthreads();
    if (c + 1 < nchunks) gload(c + 1);
    {   // |t|^2 of the staged chunk, 2 threads per row
      const int r = tid >> 1, part = tid & 1;
      float s = 0.f;
      for (int k = part; k < KCH; k += 2) s = __builtin_fmaf(Ts[r * TS + k], Ts[r * TS + k], s);
      s += __shfl_xor(s, 1, 64);
      tnorm += s;
    }
    // A = train rows of this wave (row = lane&31), B = queries (col = lane&31); k = 2*step + (lane>>5)
    const float* ap = Ts + (wave * 32 + (lane & 31)) * TS + (lane >> 5);
    const float* bp = Qs + (lane & 31) * QS + kc * KCH + (lane >> 5);
#pragma unroll 8
    for (int ks = 0; ks < KCH / 2; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
    if (kc == nkc - 1) {
      if ((tid & 1) == 0) tn[tid >> 1] = tnorm;
      __syncthreads();
      // acc[r]: train row i = (r&3) + 8*(r>>2) + 4*(lane>>5) of this wave's 32, query j = lane&31
      const float qq = qn[lane & 31];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int li = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int gi = t0 + li;
        if (gi < nt) {
          const float d2 = (tn[li] + qq) - 2.0f * acc[r];
          tnmax = fmaxf(tnmax, tn[li]);
          cand_insert(top, d2, gi);
        }
      }
    }
  }
  // merge the two lane halves (same query, different train rows)
  {
    Cand other[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      other[k].d = __shfl_xor(top[k].d, 32, 64);
      other[k].i = __shfl_xor(top[k].i, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) cand_insert(top, other[k].d, other[k].i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tnmax = fmaxf(tnmax, __shfl_xor(tnmax, o, 64));
  __syncthreads();
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < NC; ++k) merge[(wave * MQ + lane) * NC + k] = top[k];
  }
  if (lane == 0) aux[wave] = tnmax;
  __syncthreads();
  int* mcand = reinterpret_cast<int*>(Ts);          // [MQ][4] candidate indices (the train chunk buffer is free now)
  float* mdist = reinterpret_cast<float*>(Ts) + MQ * 4;  // [MQ][4] exact distances
  if (wave == 0 && lane < 32) {
    for (int w = 1; w < 4; ++w)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const Cand c = merge[(w * MQ + lane) * NC + k];
        cand_insert(top, c.d, c.i);
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mcand[lane * 4 + k] = top[k].i == 0x7FFFFFFF ? -1 : top[k].i;
      mcand2[lane * 4 + k] = top[4 + k].i == 0x7FFFFFFF ? -1 : top[4 + k].i;
    }
    m4th[lane] = top[3].d;                          // every row outside the first four has an approximate d2 of at least this
    m8th[lane] = top[NC - 1].d;                     // ... and every row outside the eight of at least this
  }
  if (tid < 2) satn[tid] = 0;
  __syncthreads();
  // exact re-rank: every (query, candidate) distance re-evaluated in the oracle's order, 16 lanes per pair.  A 16-lane group owns 8 of
  // the 128 (query, candidate) pairs; for dim = 256 the train-row elements of FOUR pairs (4 x 16 loads per lane) are requested before
  // the first is used, so the loop costs two L2 round trips instead of sixteen (this kernel is latency-bound: ~60 cycles per instruction)
  if (dim == 256) {
    const int slot = tid & 15;
#pragma unroll 1
    for (int r0 = 0; r0 < (MQ * 4 * 16) / 256; r0 += 4) {
      float tv[4][16];
      int cis[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pr = (r0 + u) * 16 + (tid >> 4);
        cis[u] = mcand[pr];
        const float* trow = T + (size_t)(cis[u] >= 0 ? cis[u] : 0) * 256 + slot;
#pragma unroll
        for (int i = 0; i < 16; ++i) tv[u][i] = trow[16 * i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pr = (r0 + u) * 16 + (tid >> 4);
        const float* q = Qs + (pr >> 2) * QS + slot;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {              // ascending j = slot + 16 i: the oracle's order within an accumulator lane
          const float d = q[16 * i] - tv[u][i];
          const float dd = d * d;
          acc = acc + dd;
        }
        const int base = lane & ~15, l = slot & 3;
        const float a0 = __shfl(acc, base + 0 + l, 64), a1 = __shfl(acc, base + 4 + l, 64);
        const float a2 = __shfl(acc, base + 8 + l, 64), a3 = __shfl(acc, base + 12 + l, 64);
        const float rr = ((a0 + a1) + a2) + a3;
        const float r0v = __shfl(rr, base + 0, 64), r1v = __shfl(rr, base + 1, 64);
        const float r2v = __shfl(rr, base + 2, 64), r3v = __shfl(rr, base + 3, 64);
        const float dd = __builtin_sqrtf((r0v + r2v) + (r1v + r3v));
        if (slot == 0) mdist[pr] = cis[u] >= 0 ? dd : __builtin_inff();
      }
    }
  } else {
    for (int r = 0; r < (MQ * 4 * 16) / 256; ++r) {
      const int pr = r * 16 + (tid >> 4);
      const int q = pr >> 2;
      const int ci = mcand[pr];
      const float* trow = T + (size_t)(ci >= 0 ? ci : 0) * dim;
      const float dd = exact_dist16(Qs + q * QS, trow, dim, tid & 15, lane);
      if ((tid & 15) == 0) mdist[pr] = ci >= 0 ? dd : __builtin_inff();
    }
  }
  __syncthreads();
  // the exact 2-NN among a query's re-ranked candidates: (distance, index) lexicographic = the oracle's insertion order
  auto best2 = [&](int q, int ncand, float& bd0, int& bi0, float& bd1, int& bi1) {
    bd0 = __builtin_inff(); bd1 = __builtin_inff(); bi0 = -1; bi1 = -1;
    for (int k = 0; k < ncand; ++k) {
      const float dk = k < 4 ? mdist[q * 4 + k] : mdist2[q * 4 + k - 4];
      const int ik = k < 4 ? mcand[q * 4 + k] : mcand2[q * 4 + k - 4];
      if (ik < 0) continue;
      if (dk < bd0 || (dk == bd0 && ik < bi0)) { bd1 = bd0; bi1 = bi0; bd0 = dk; bi0 = ik; }
      else if (dk < bd1 || (dk == bd1 && ik < bi1)) { bd1 = dk; bi1 = ik; }
    }
  };
  auto emit = [&](int q, int bi0, float bd0, float bd1) {
    int32_t* out = m.cand4 + (((size_t)pair * 2 + dir) * m.max_n + q0 + q) * 4;   // {nn index, d0 bits, d1 bits, (finalize: inverse dictionary)}
    out[0] = bi0; out[1] = __float_as_int(bd0); out[2] = __float_as_int(bd1);
  };
  const float tn_all = fmaxf(fmaxf(aux[0], aux[1]), fmaxf(aux[2], aux[3]));
  if (tid < MQ && q0 + tid < nq) {
    float bd0, bd1; int bi0, bi1;
    best2(tid, 4, bd0, bi0, bd1, bi1);
    // saturation test (header): can a row outside the first four still beat the exact 2nd neighbour?
    const float slack = GRAM_ERR * (qn[tid] + tn_all);
    const bool saturated = !m.no_fallback && nt > 4 && !(m4th[tid] - slack > bd1 * bd1 * 1.00001f);
    if (saturated) sat[atomicAdd(&satn[0], 1)] = tid; else emit(tid, bi0, bd0, bd1);
  }
  __syncthreads();
  const int ns = satn[0];
  if (ns > 0) {
    // level 2: candidates 5..8 of the saturated queries, 16 lanes per (query, candidate)
    for (int p0 = 0; p0 < ns * 4; p0 += 16) {
      const int pr = p0 + (tid >> 4);
      if (pr < ns * 4) {
        const int q = sat[pr >> 2], k = pr & 3;
        const int ci = mcand2[q * 4 + k];
        const float dd = exact_dist16(Qs + q * QS, T + (size_t)(ci >= 0 ? ci : 0) * dim, dim, tid & 15, lane);
        if ((tid & 15) == 0) mdist2[q * 4 + k] = ci >= 0 ? dd : __builtin_inff();
      }
    }
    __syncthreads();
    if (tid < ns) {
      const int q = sat[tid];
      float bd0, bd1; int bi0, bi1;
      best2(q, 8, bd0, bi0, bd1, bi1);
      const float slack = GRAM_ERR * (qn[q] + tn_all);
      const bool still = nt > NC && !(m8th[q] - slack > bd1 * bd1 * 1.00001f);
      if (still) sat2[atomicAdd(&satn[1], 1)] = q; else emit(q, bi0, bd0, bd1);
      if (m.stats) atomicAdd(m.stats + 1, 1);
    }
    __syncthreads();
  }
  // exact scan of the queries that are saturated even with eight candidates (degenerate inputs): 16 lanes per train row, 16 rows in
  // flight, every row of the pair's train set
  const int ns2 = ns > 0 ? satn[1] : 0;
  for (int si = 0; si < ns2; ++si) {
    const int sq = sat2[si];
    const int grp = tid >> 4, slot = tid & 15;
    Cand b0{__builtin_inff(), 0x7FFFFFFF}, b1{__builtin_inff(), 0x7FFFFFFF};
    for (int j = grp; j < nt; j += 16) {
      const float dj = exact_dist16(Qs + sq * QS, T + (size_t)j * dim, dim, slot, lane);
      if (cand_less(dj, j, b0)) { b1 = b0; b0.d = dj; b0.i = j; }
      else if (cand_less(dj, j, b1)) { b1.d = dj; b1.i = j; }
    }
    if (slot == 0) { scan[grp * 2] = b0; scan[grp * 2 + 1] = b1; }
    __syncthreads();
    if (tid == 0) {
      Cand r0{__builtin_inff(), 0x7FFFFFFF}, r1{__builtin_inff(), 0x7FFFFFFF};
      for (int g = 0; g < 32; ++g) {
        const Cand c = scan[g];
        if (c.i == 0x7FFFFFFF) continue;
        if (cand_less(c.d, c.i, r0)) { r1 = r0; r0 = c; }
        else if (cand_less(c.d, c.i, r1)) { r1 = c; }
      }
      emit(sq, r0.i == 0x7FFFFFFF ? -1 : r0.i, r0.d, r1.d);
      if (m.stats) atomicAdd(m.stats, 1);
    }
    __syncthreads();
  }
}

constexpr int FIN_THREADS = 1024;
constexpr int MATCH_MAXN = 16384;       // rows per side (the candidate scratch is 2 x max_n x 16 bytes per pair)

// One workgroup per pair.  The exact 2-NN records {nn index, d0, d1, -} of every row of both directions sit in the candidate scratch
// (written by the prefilter blocks); nothing here is sized by the row count: the inverse dictionary goes into the records' fourth
// word, the forward test walks the queries in chunks of 1024 with an ordered compaction.
__global__ __launch_bounds__(FIN_THREADS) void match_finalize_kernel(MatchArgs m) {
  __shared__ int wsum[FIN_THREADS / 64];
  const int pair = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int na = min(m.a_cnt[pair], m.max_n), nb = min(m.b_cnt[pair], m.max_n);
  int32_t* fwd = m.cand4 + ((size_t)pair * 2 + 0) * m.max_n * 4;
  int32_t* inv = m.cand4 + ((size_t)pair * 2 + 1) * m.max_n * 4;
  // phase 1: inverse dictionary (feature_matcher.cpp:16-25): inv[j][3] = the a-row b-row j names, or -1
  for (int j = tid; j < nb; j += FIN_THREADS) {
    int v = -1;
    const int g0 = inv[4 * j];
    if (m.mode == 0) {
      if (na >= 2 && (double)__int_as_float(inv[4 * j + 1]) < m.ratio * (double)__int_as_float(inv[4 * j + 2])) v = g0;
    } else {
      v = g0;
    }
    inv[4 * j + 3] = v;
  }
  __syncthreads();       // workgroup-scope release/acquire: the records are read back below by other threads of this workgroup
  // phase 2: forward test + ordered compaction (ascending query index)
  int base = 0;
  for (int i0 = 0; i0 < na; i0 += FIN_THREADS) {
    const int i = i0 + tid;
    bool ok = false;
    int j = -1;
    float d0 = 0.f;
    if (i < na) {
      j = fwd[4 * i];
      d0 = __int_as_float(fwd[4 * i + 1]);
      const float d1 = __int_as_float(fwd[4 * i + 2]);
      if (m.mode == 0) {
        ok = nb >= 2 && j >= 0 && (double)d0 < m.ratio * (double)d1 && inv[4 * j + 3] == i;
        if (ok && m.radius > 0 && m.pts_a && m.pts_b) {
          const float* pa = m.pts_a + 2 * ((size_t)m.a_off[pair] + i);
          const float* pb = m.pts_b + 2 * ((size_t)m.b_off[pair] + j);
          const float dx = pa[0] - pb[0], dy = pa[1] - pb[1];
          const double nr = __builtin_sqrt((double)dx * dx + (double)dy * dy);
          if (nr > m.radius) ok = false;
        }
      } else {
        ok = j >= 0 && inv[4 * j + 3] == i;
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    int total = 0;
    for (int w = 0; w < FIN_THREADS / 64; ++w) total += wsum[w];
    if (ok) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < m.max_n) {
        const size_t o = (size_t)pair * m.max_n + pos;
        m.q_idx[o] = i; m.t_idx[o] = j; m.dist[o] = d0;
      }
    }
    base += total;
    __syncthreads();
  }
  if (tid == 0) m.n_out[pair] = base < m.max_n ? base : m.max_n;
}

hipError_t launch_match(const MatchArgs& m_in, hipStream_t s) {
  MatchArgs m = m_in;
  { static const int nf = [] { const char* e = getenv("D2FE_MATCH_NOFALLBACK"); return e ? atoi(e) : 0; }(); m.no_fallback = nf; }
  if (m.dim > MAXDIM || (m.dim & 3) || m.max_n > MATCH_MAXN || m.max_n < 1) return hipErrorInvalidValue;
  const size_t lds = sizeof(float) * (MQ * QS + TB * TS + MQ + TB) + sizeof(Cand) * 4 * MQ * NC + sizeof(float) * (4 + 2 * MQ + 8 * MQ) +
                     sizeof(int) * (2 * MQ + 4) + sizeof(Cand) * 32;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(match_prefilter_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  dim3 grid((m.max_n + MQ - 1) / MQ, 2, m.npairs);
  hipLaunchKernelGGL(match_prefilter_kernel, grid, dim3(256), lds, s, m);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(match_finalize_kernel, dim3(m.npairs), dim3(FIN_THREADS), 0, s, m);
  return hipGetLastError();
}

}  // namespace d2fe

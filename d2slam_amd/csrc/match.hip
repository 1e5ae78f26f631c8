// match.hip -- brute-force descriptor matching on gfx950: matchKNN and the cross-check matcher, ONE launch per batch of pairs.
//
// Replaces D2FrontEnd::matchKNN (d2frontend/src/feature_matcher.cpp:4-42: cv::BFMatcher(NORM_L2).knnMatch
// both ways, Lowe ratio, mutual check, optional pixel-radius gate) and cv::BFMatcher(NORM_L2, true).match
// (loop_cam.cpp:167-170, d2featuretracker.cpp:1141-1142).
//
// match_kernel, one workgroup (2 waves; 4 when the whole launch is resident with them: single pairs) per (pair, direction, 32-query tile):
//  (1) distance strip on the fp32 matrix pipe.  Each wave keeps the whole query tile as MFMA B fragments in registers (32 queries x 256
//      floats = 128 VGPRs) and streams the train rows 16 at a time straight from L2 into A fragments -- no LDS staging: the contraction
//      index may be visited in any order as long as both operands use the same one, so lane (row r, k-group g) loads the float4 at
//      k = 16 s + 4 g and feeds its four elements to four v_mfma_f32_16x16x4_f32.  d2 = |t|^2 + |q|^2 - 2 t.q goes into an LDS strip
//      S[query][train row] (up to 256 train rows per pass; longer train sets are walked in passes).
//  (2) threshold instead of top-k lists.  |d2_gram - d2_exact| <= GRAM_ERR (|q|^2 + |t|^2) (below).  With m2 = the second smallest
//      strip value of a query and slack = GRAM_ERR (|q|^2 + max |t|^2), every row with S <= tau = (m2 + slack) 1.00002 + slack is a
//      CANDIDATE (the two nearest are always among them; on SuperPoint descriptors 2.0-2.3 rows per query); candidates are re-evaluated
//      in the ORACLE's arithmetic (orc_l2_dist: OpenCV normL2Sqr_ accumulation order, then sqrt) by 16 lanes each, and the exact
//      2-NN (distance, then index: the reference's insertion order) is taken among them.  Every other row has an exact squared
//      distance > tau - slack, which is checked against the exact second neighbour found (always true by construction for one pass;
//      the check also covers multi-pass train sets with unequal norms).  A query with more than CMAX candidates in a pass (more than
//      sixteen rows within round-off of each other: repeated texture, a frame matched against a near-copy, all-equal sets) or a failed
//      check takes an exact scan of every train row.  Indices and distances equal the oracle's bit for bit for any input.
//  (3) the workgroup that finishes LAST for a pair (one device counter per pair) applies ratio / mutual / radius tests in double exactly
//      as feature_matcher.cpp:16-37 and emits the matches in ascending query order: no second launch.  Hand-off between workgroups
//      follows the agent-scope recipe (write-through record stores, vmcnt drain, barrier, relaxed agent ticket; the last arriver
//      acquires once, then plain loads).
// One distance matrix would serve both directions, but the per-query selection wants the queries on the lanes: the two directions
// are two strips (2 x 20.5 MFLOP per 200 x 200 pair; at 64 pairs 3.3 GFLOP = 21 us of the fp32 matrix pipe).
#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MQ = 32;           // queries per workgroup (two 16-wide MFMA column tiles)
constexpr int SC = 256;          // train rows per pass (width of the LDS strip)
constexpr int SPITCH = SC + 4;   // strip row pitch (floats): the 4-threads-per-query scan below is bank-conflict-free
constexpr int CMAX = 16;         // candidate slots per query and pass
constexpr int SHMAX = 8;         // ... of which one scan thread's share of the strip row may hold
constexpr int MAXDIM = 256;
constexpr int KST = MAXDIM / 16; // k-steps of 16 floats (4 k-groups x float4)
// |d2_gram - d2_exact| <= GRAM_ERR * (|q|^2 + |t|^2): three fp32 sums of <= 256 products (gamma_257 = 257 * 2^-24 = 1.53e-5 each, and
// sum |t_k q_k| <= (|q|^2 + |t|^2) / 2) give 2 * gamma_257 = 3.1e-5; 4e-5 leaves room for the final subtraction's rounding
constexpr float GRAM_ERR = 4.0e-5f;
constexpr int MATCH_MAXN = 16384;       // rows per side (the record scratch is 2 x max_n x 16 bytes per pair)

struct Cand { float d; int i; };
__device__ __forceinline__ bool cand_less(float d, int i, float cd, int ci) { return d < cd || (d == cd && i < ci); }

template <int NW>
struct MatchSmem {
  float S[MQ * SPITCH];
  float qn[MQ], rm1[MQ], rm2[MQ], lb[MQ], bd0[MQ], bd1[MQ];
  int bi0[MQ], bi1[MQ], ccnt[MQ], ctot[MQ], over[MQ], fb[MQ];
  unsigned short wslot[MQ * CMAX];   // work lists, one region per wave: q * CMAX + slot of every candidate of the pass (4 workgroups per CU: <= 40 KiB each)
  int ct[MQ * CMAX];             // [q][slot] train row
  float cd[MQ * CMAX];           // [q][slot] exact distance
  float tnmax_w[NW];
  int nwork_w[NW], wsum[NW];
  int nfb, last;
  Cand scan[2 * NW * 4];
};

// exact distance of one (q, t) row pair with 16 lanes: lane slot s accumulates elements j = 16*i + s
// (slot s = 4*v + l of OpenCV's four 4-lane accumulators), then the oracle's reduction order.
__device__ __forceinline__ float exact_reduce16(float acc, int slot, int lane) {
  // r[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l]   with slot = 4*v + l
  const int base = lane & ~15;
  const int l = slot & 3;
  const float a0 = __shfl(acc, base + 0 + l, 64), a1 = __shfl(acc, base + 4 + l, 64);
  const float a2 = __shfl(acc, base + 8 + l, 64), a3 = __shfl(acc, base + 12 + l, 64);
  const float r = ((a0 + a1) + a2) + a3;  // valid in every lane for its l
  const float r0 = __shfl(r, base + 0, 64), r1 = __shfl(r, base + 1, 64);
  const float r2 = __shfl(r, base + 2, 64), r3 = __shfl(r, base + 3, 64);
  return (r0 + r2) + (r1 + r3);
}
__device__ __forceinline__ float exact_dist16(const float* __restrict__ q, const float* __restrict__ t, int dim, int slot,
                                              int lane) {
  float acc = 0.f;
  const int nfull = dim & ~15;
  // both rows come from global memory: eight element pairs are requested before the first use (one L2 round trip per batch instead
  // of one per element); the accumulation itself stays in ascending j, the oracle's order
  for (int j0 = slot; j0 < nfull; j0 += 128) {
    float tv[8], qv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const bool in = j0 + 16 * u < nfull; tv[u] = in ? t[j0 + 16 * u] : 0.f; qv[u] = in ? q[j0 + 16 * u] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + 16 * u < nfull) {
        const float d = qv[u] - tv[u];
        const float dd = d * d;
        acc = acc + dd;
      }
  }
  float d = exact_reduce16(acc, slot, lane);
  for (int j = nfull; j < dim; ++j) { const float e = q[j] - t[j]; d += e * e; }
  return __builtin_sqrtf(d);
}

struct PairView { const float* a; const float* b; const float* pa; const float* pb; int na, nb; double radius; };
__device__ __forceinline__ PairView pair_view(const MatchArgs& m, int pair) {
  PairView v;
  if (m.pairs) {
    const MatchPairDesc d = m.pairs[pair];
    v.a = d.a; v.b = d.b; v.pa = d.pts_a; v.pb = d.pts_b; v.na = min(*d.na, m.max_n); v.nb = min(*d.nb, m.max_n); v.radius = d.radius;
  } else {
    v.a = m.a + (size_t)m.a_off[pair] * m.dim; v.b = m.b + (size_t)m.b_off[pair] * m.dim;
    v.pa = m.pts_a ? m.pts_a + 2 * (size_t)m.a_off[pair] : nullptr; v.pb = m.pts_b ? m.pts_b + 2 * (size_t)m.b_off[pair] : nullptr;
    v.na = min(m.a_cnt[pair], m.max_n); v.nb = min(m.b_cnt[pair], m.max_n); v.radius = m.radius;
  }
  return v;
}

__device__ __forceinline__ void best2_insert(float d, int i, float& bd0, int& bi0, float& bd1, int& bi1) {
  if (cand_less(d, i, bd0, bi0)) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = i; }
  else if (cand_less(d, i, bd1, bi1)) { bd1 = d; bi1 = i; }
}

// tail of the workgroup that arrives last for its pair (feature_matcher.cpp:16-37).  The exact 2-NN records {nn index, d0, d1, -} of
// every row of both directions sit in the record scratch; nothing here is sized by the row count: the inverse dictionary goes into
// the records' fourth word, the forward test walks the queries in chunks of one workgroup with an ordered compaction.
template <int NW>
__device__ void match_finalize_pair(const MatchArgs& m, int pair, MatchSmem<NW>& sm) {
  constexpr int NTHR = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const PairView pv = pair_view(m, pair);
  const int na = pv.na, nb = pv.nb;
  int32_t* fwd = m.cand4 + ((size_t)pair * 2 + 0) * m.max_n * 4;
  int32_t* inv = m.cand4 + ((size_t)pair * 2 + 1) * m.max_n * 4;
  // phase 1: inverse dictionary (feature_matcher.cpp:16-25): inv[j][3] = the a-row b-row j names, or -1
  for (int j = tid; j < nb; j += NTHR) {
    int v = -1;
    const int g0 = inv[4 * j];
    if (m.mode == 0) {
      if (na >= 2 && (double)__int_as_float(inv[4 * j + 1]) < m.ratio * (double)__int_as_float(inv[4 * j + 2])) v = g0;
    } else {
      v = g0;
    }
    __hip_atomic_store(inv + 4 * j + 3, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through: re-read below by other lanes
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // phase 2: forward test + ordered compaction (ascending query index)
  int base = 0;
  for (int i0 = 0; i0 < na; i0 += NTHR) {
    const int i = i0 + tid;
    bool ok = false;
    int j = -1;
    float d0 = 0.f;
    if (i < na) {
      j = fwd[4 * i];
      d0 = __int_as_float(fwd[4 * i + 1]);
      const float d1 = __int_as_float(fwd[4 * i + 2]);
      if (m.mode == 0) {
        ok = nb >= 2 && j >= 0 && (double)d0 < m.ratio * (double)d1 &&
             __hip_atomic_load(inv + 4 * j + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i;
        if (ok && pv.radius > 0 && pv.pa && pv.pb) {
          const float* pa = pv.pa + 2 * (size_t)i;
          const float* pb = pv.pb + 2 * (size_t)j;
          const float dx = pa[0] - pb[0], dy = pa[1] - pb[1];
          const double nr = __builtin_sqrt((double)dx * dx + (double)dy * dy);
          if (nr > pv.radius) ok = false;
        }
      } else {
        ok = j >= 0 && __hip_atomic_load(inv + 4 * j + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i;
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) sm.wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = sm.wsum[w]; off += w < wave ? c : 0; total += c; }
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

template <int NST, bool FULL, int NW>
__global__ __launch_bounds__(64 * NW, 2) void match_kernel(MatchArgs m, int tiles) {
  constexpr int NTHR = 64 * NW;
  constexpr int TPQ = NTHR / MQ;             // threads per query in the scan: 4 or 8 adjacent lanes
  constexpr int QPW = 64 / TPQ;              // queries per wave in the scan
  constexpr int VPT = SC / TPQ;              // strip values per thread (64 or 32), held in registers
  constexpr int WREG = MQ * CMAX / NW;       // work-list region of a wave
  __shared__ __attribute__((aligned(16))) MatchSmem<NW> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup b runs on XCD b % 8 (observed; a speed assumption only): give every XCD a contiguous run of (pair, direction, tile) items
  // so that all tiles of a pair read its descriptors through one L2
  const int nwg = gridDim.x;
  int item;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, qn8 = nwg >> 3, r8 = nwg & 7;
    item = xcd * qn8 + (xcd < r8 ? xcd : r8) + slot;
  }
  const int per_pair = 2 * tiles;
  const int pair = item / per_pair, dir = (item % per_pair) / tiles, tile = item % tiles;
  const PairView pv = pair_view(m, pair);
  const int nq = dir == 0 ? pv.na : pv.nb, nt = dir == 0 ? pv.nb : pv.na;
  const int q0 = tile * MQ;
  const int dim = m.dim;
  const float* Q = dir == 0 ? pv.a : pv.b;
  const float* T = dir == 0 ? pv.b : pv.a;

  D2FE_STAMP(m.stamps, blockIdx.x, 0);
  if (q0 < nq) {
    const int c = lane & 15, g = lane >> 4;
    const int nst = (dim + 15) >> 4;
    if (tid < MQ) {
      sm.rm1[tid] = __builtin_inff(); sm.rm2[tid] = __builtin_inff(); sm.lb[tid] = __builtin_inff();
      sm.bd0[tid] = __builtin_inff(); sm.bd1[tid] = __builtin_inff(); sm.bi0[tid] = -1; sm.bi1[tid] = -1;
      sm.ctot[tid] = 0; sm.over[tid] = 0;
    }
    if (tid == 0) sm.nfb = 0;
    for (int t_base = 0; t_base < nt; t_base += SC) {
      const int ntc = min(SC, nt - t_base);
      const int ntiles = (ntc + 15) >> 4;
      // ---- (1) the strip.  Query fragments are (re)loaded per pass so that they are not live across the exact phase below
      f32x4 qf[2][NST];
      float qq[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // rows past the end are clamped to a valid row: their strip entries are never read (branch-free loads from one base register)
        const int row = q0 + 16 * j + c;
        const float* qp = Q + (size_t)(row < nq ? row : q0) * dim + 4 * g;
        float s = 0.f;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          f32x4 v;
          if (FULL) {
            v = *reinterpret_cast<const f32x4*>(qp + 16 * st);
          } else {       // dim < 256: columns past the end read column 0 and count as zero
            const bool kv = 16 * st + 4 * g < dim;
            v = *reinterpret_cast<const f32x4*>(qp + (kv ? 16 * st : -4 * g));
            if (!kv) v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          qf[j][st] = v;
          s = __builtin_fmaf(v[0], v[0], s); s = __builtin_fmaf(v[1], v[1], s); s = __builtin_fmaf(v[2], v[2], s); s = __builtin_fmaf(v[3], v[3], s);
        }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        qq[j] = s;
      }
      D2FE_STAMP(m.stamps, blockIdx.x, 1);
      __syncthreads();       // the previous pass is done with the strip and the work lists (first pass: the initial values are visible)
      if (wave == 0 && g == 0) { sm.qn[c] = qq[0]; sm.qn[16 + c] = qq[1]; }
      float tnmax = 0.f;
      // train tiles of this wave: ti = wave, wave + NW, ...  The A fragments of a tile are two halves of NST / 2 k-steps; the loads of the
      // next tile's half are issued right behind the MFMAs that consumed this tile's same half, so they travel under the other half's
      // MFMAs (one wave per SIMD when a single pair is matched: nothing else hides the L2 round trip)
      constexpr int NH = NST / 2;
      auto tile_ptr = [&](int ti) {
        const int trow = t_base + ti * 16 + c;
        return T + (size_t)(trow < nt ? trow : t_base) * dim + 4 * g;
      };
      auto load_half = [&](f32x4 (&dst)[NH], const float* tp, int st0) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
          const int st = st0 + i;
          f32x4 v;
          if (FULL) {
            v = *reinterpret_cast<const f32x4*>(tp + 16 * st);
          } else {
            const bool kv = 16 * st + 4 * g < dim;
            v = *reinterpret_cast<const f32x4*>(tp + (kv ? 16 * st : -4 * g));
            if (!kv) v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          dst[i] = v;
        }
      };
      f32x4 fa[NH], fb[NH];
      if (wave < ntiles) { const float* tp0 = tile_ptr(wave); load_half(fa, tp0, 0); load_half(fb, tp0, NH); }
      for (int ti = wave; ti < ntiles; ti += NW) {
        const float* tpn = tile_ptr(ti + NW < ntiles ? ti + NW : ti);      // past the last tile: the same tile again (never used)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float tn = 0.f;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
          if (FULL || i < nst) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], qf[0][i][e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], qf[1][i][e], acc1, 0, 0, 0);
              tn = __builtin_fmaf(fa[i][e], fa[i][e], tn);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_half(fa, tpn, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
          if (FULL || NH + i < nst) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i][e], qf[0][NH + i][e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i][e], qf[1][NH + i][e], acc1, 0, 0, 0);
              tn = __builtin_fmaf(fb[i][e], fb[i][e], tn);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_half(fb, tpn, NH);
        __builtin_amdgcn_sched_barrier(0);
        tn += __shfl_xor(tn, 16, 64); tn += __shfl_xor(tn, 32, 64);     // |t|^2 of train row c of this tile, in every lane with lane & 15 == c
        if (t_base + ti * 16 + c < nt) tnmax = fmaxf(tnmax, tn);
        // acc[r]: train row 4 g + r of the tile, query (lane & 15) of the column tile; rows past the end of the train set: +inf
        f32x4 d0, d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float tr = __shfl(tn, 4 * g + r, 64);
          const bool in = ti * 16 + 4 * g + r < ntc;
          d0[r] = in ? (tr + qq[0]) - 2.0f * acc0[r] : __builtin_inff();
          d1[r] = in ? (tr + qq[1]) - 2.0f * acc1[r] : __builtin_inff();
        }
        *reinterpret_cast<f32x4*>(&sm.S[c * SPITCH + ti * 16 + 4 * g]) = d0;
        *reinterpret_cast<f32x4*>(&sm.S[(16 + c) * SPITCH + ti * 16 + 4 * g]) = d1;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tnmax = fmaxf(tnmax, __shfl_xor(tnmax, o, 64));
      if (lane == 0) sm.tnmax_w[wave] = tnmax;
      D2FE_STAMP(m.stamps, blockIdx.x, 2);
      __syncthreads();
      D2FE_STAMP(m.stamps, blockIdx.x, 3);
      // ---- (2) per query: the two smallest strip values, the threshold, the candidates.  TPQ adjacent lanes per query, each owns a
      // contiguous share of the strip row, read once as float4s into registers (entries past the pass's last tile are not read: +inf).
      // No atomics: a thread keeps its hits as a bit mask, the slots of a query are a prefix over its lanes, the work list of a wave a
      // prefix over its queries
      {
        const int q = tid / TPQ, p = tid % TPQ;
        const bool qvalid = q0 + q < nq;
        const int L = ((ntiles * 16 + 4 * TPQ - 1) / (4 * TPQ)) * 4;      // share length, a multiple of 4: TPQ * L >= 16 * ntiles
        const int tq0 = p * L;
        const float* srow = sm.S + q * SPITCH + tq0;
        const int lim = ntiles * 16 - tq0;                    // entries of this share that exist in the strip (may be <= 0)
        float v[VPT];
#pragma unroll
        for (int i = 0; i < VPT; i += 4) {
          f32x4 v4 = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
          if (i < L && i < lim) v4 = *reinterpret_cast<const f32x4*>(srow + i);        // whole float4s: tq0, L and 16 * ntiles are multiples of 4
          v[i] = v4[0]; v[i + 1] = v4[1]; v[i + 2] = v4[2]; v[i + 3] = v4[3];
        }
        float m1 = __builtin_inff(), m2 = __builtin_inff();
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
          m2 = __builtin_amdgcn_fmed3f(m1, m2, v[i]);        // (m1 <= m2): the second smallest of {m1, m2, v}
          m1 = fminf(m1, v[i]);
        }
#pragma unroll
        for (int o = 1; o < TPQ; o <<= 1) {
          const float o1 = __shfl_xor(m1, o, 64), o2 = __shfl_xor(m2, o, 64);
          const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
          m2 = fminf(hi, fminf(m2, o2)); m1 = lo;
        }
        // running over the passes (all threads of a query compute the same values; thread p == 0 stores them)
        const float r1 = sm.rm1[q], r2 = sm.rm2[q];
        const float n1 = fminf(m1, r1), n2 = fminf(fmaxf(m1, r1), fminf(m2, r2));
        float tnall = sm.tnmax_w[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tnall = fmaxf(tnall, sm.tnmax_w[w]);
        const float slack = GRAM_ERR * (sm.qn[q] + tnall);
        const float tau = (n2 + slack) * 1.00002f + slack;          // +inf while fewer than two rows have been seen: everything is a candidate
        // (the threads of a query are adjacent lanes of one wave: program order separates these reads from the stores below)
        if (p == 0) { sm.rm1[q] = n1; sm.rm2[q] = n2; sm.lb[q] = fminf(sm.lb[q], tau - slack); }
        unsigned long long hits = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) hits |= (v[i] <= tau && v[i] < __builtin_inff()) ? (1ull << i) : 0ull;
        if (!qvalid) hits = 0;
        const int cnt = __popcll(hits);
        // slots of the query: exclusive prefix of its lanes' stored counts; more than SHMAX hits in one share or more than CMAX in the row:
        // the query takes the exact scan (ccnt > CMAX -> sm.over)
        const int sc = cnt < SHMAX ? cnt : SHMAX;
        int ofq = cnt > SHMAX ? 1 : 0;
        int incq = sc;
#pragma unroll
        for (int o = 1; o < TPQ; o <<= 1) {
          ofq |= __shfl_xor(ofq, o, 64);
          const int up = __shfl_up(incq, o, TPQ);
          if (p >= o) incq += up;
        }
        const int base = incq - sc;
        const int total = __shfl(incq, (lane & ~(TPQ - 1)) + TPQ - 1, 64);
        const int kept = total <= CMAX ? total : CMAX;
        // work list of the wave: exclusive prefix of `kept` over its queries (lanes with p == 0 contribute)
        int incl = p == 0 ? kept : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
        const int woff = __shfl(incl, lane & ~(TPQ - 1), 64) - kept;       // entries of the wave's queries before this one
        if (p == 0) sm.ccnt[q] = ofq ? CMAX + 1 : total;
#pragma unroll
        for (int k = 0; k < SHMAX; ++k) {
          if (k < sc && base + k < CMAX) {
            const int j = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            sm.ct[q * CMAX + base + k] = t_base + tq0 + j;
            sm.wslot[wave * WREG + woff + base + k] = (unsigned short)(q * CMAX + base + k);
          }
        }
        if (lane == 63) sm.nwork_w[wave] = incl;
      }
      __syncthreads();
      D2FE_STAMP(m.stamps, blockIdx.x, 4);
      // ---- exact re-evaluation of the candidates: 16 lanes per (query, candidate), EU candidates per 16-lane group in flight (the
      // loop costs one L2 round trip per EU * NTHR / 16 candidates)
      {
        constexpr int EU = 4;
        int wstart[NW + 1];
        wstart[0] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) wstart[w + 1] = wstart[w] + sm.nwork_w[w];
        const int E = wstart[NW];
        auto work = [&](int e) {
          int r = sm.wslot[e];
#pragma unroll
          for (int w = 1; w < NW; ++w) if (e >= wstart[w]) r = sm.wslot[w * WREG + e - wstart[w]];
          return r;
        };
        const int slot = tid & 15, grp = tid >> 4;
        for (int e0 = 0; e0 < E; e0 += EU * (NTHR / 16)) {
          if (FULL) {
            float tv[EU][16], qv[EU][16];
            int ws[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
              const int e = e0 + EU * grp + u;
              ws[u] = e < E ? work(e) : -1;
              const int w = ws[u] >= 0 ? ws[u] : 0;
              const float* trow = T + (size_t)(ws[u] >= 0 ? sm.ct[w] : 0) * MAXDIM + slot;
              const float* qrow = Q + (size_t)(q0 + (w / CMAX)) * MAXDIM + slot;
#pragma unroll
              for (int i = 0; i < 16; ++i) { tv[u][i] = trow[16 * i]; qv[u][i] = qrow[16 * i]; }
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
              float acc = 0.f;
#pragma unroll
              for (int i = 0; i < 16; ++i) {              // ascending j = slot + 16 i: the oracle's order within an accumulator lane
                const float d = qv[u][i] - tv[u][i];
                const float dd = d * d;
                acc = acc + dd;
              }
              const float dd = __builtin_sqrtf(exact_reduce16(acc, slot, lane));
              if (slot == 0 && ws[u] >= 0) sm.cd[ws[u]] = dd;
            }
          } else {
#pragma unroll 1
            for (int u = 0; u < EU; ++u) {
              const int e = e0 + EU * grp + u;
              const int w = e < E ? work(e) : 0;
              const int trw = e < E ? sm.ct[w] : 0;
              const float dd = exact_dist16(Q + (size_t)(q0 + (w / CMAX)) * dim, T + (size_t)trw * dim, dim, slot, lane);
              if (slot == 0 && e < E) sm.cd[w] = dd;
            }
          }
        }
      }
      __syncthreads();
      D2FE_STAMP(m.stamps, blockIdx.x, 5);
      if (tid < MQ && q0 + tid < nq) {
        const int n = sm.ccnt[tid];
        float bd0 = sm.bd0[tid], bd1 = sm.bd1[tid];
        int bi0 = sm.bi0[tid], bi1 = sm.bi1[tid];
        for (int k = 0; k < (n < CMAX ? n : CMAX); ++k) best2_insert(sm.cd[tid * CMAX + k], sm.ct[tid * CMAX + k], bd0, bi0, bd1, bi1);
        sm.bd0[tid] = bd0; sm.bd1[tid] = bd1; sm.bi0[tid] = bi0; sm.bi1[tid] = bi1;
        sm.ctot[tid] += n;
        if (n > CMAX) sm.over[tid] = 1;
        if (m.stats && n > 2) atomicAdd(m.stats + 1, n - 2);
      }
    }
    __syncthreads();
    // ---- the records: {nn index, d0 bits, d1 bits, (finalize: inverse dictionary)}, write-through (read by the pair's last workgroup)
    auto emit = [&](int q, int bi0, float bd0, float bd1) {
      int32_t* out = m.cand4 + (((size_t)pair * 2 + dir) * m.max_n + q0 + q) * 4;
      __hip_atomic_store(out + 0, bi0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(out + 1, __float_as_int(bd0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(out + 2, __float_as_int(bd1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (tid < MQ && q0 + tid < nq) {
      const float bd1 = sm.bd1[tid];
      // every row that was not a candidate has an exact squared distance > lb; it cannot be (or tie with) one of the two nearest if
      // lb exceeds the exact second neighbour's (1.00001: sqrt rounding and the re-squaring)
      const bool proven = !sm.over[tid] && (sm.ctot[tid] >= nt || sm.lb[tid] > bd1 * bd1 * 1.00001f);
      if (proven) emit(tid, sm.bi0[tid], sm.bd0[tid], bd1);
      else sm.fb[atomicAdd(&sm.nfb, 1)] = tid;
    }
    __syncthreads();
    // exact scan of the queries that could not be proven (degenerate inputs): 16 lanes per train row, NTHR / 16 rows in flight, every row
    const int nfb = sm.nfb;
    for (int si = 0; si < nfb; ++si) {
      const int sq = sm.fb[si];
      const int grp = tid >> 4, slot = tid & 15;
      float b0d = __builtin_inff(), b1d = __builtin_inff();
      int b0i = 0x7FFFFFFF, b1i = 0x7FFFFFFF;
      if (FULL) {
        // the query row once, four train rows per 16-lane group in flight: 4 * NTHR / 16 rows per L2 round trip
        constexpr int EU = 4;
        float qv[16];
        { const float* qrow = Q + (size_t)(q0 + sq) * MAXDIM + slot;
#pragma unroll
          for (int i = 0; i < 16; ++i) qv[i] = qrow[16 * i]; }
        for (int j0 = 0; j0 < nt; j0 += EU * (NTHR / 16)) {
          float tv[EU][16];
#pragma unroll
          for (int u = 0; u < EU; ++u) {
            const int j = j0 + EU * grp + u;
            const float* trow = T + (size_t)(j < nt ? j : 0) * MAXDIM + slot;
#pragma unroll
            for (int i = 0; i < 16; ++i) tv[u][i] = trow[16 * i];
          }
#pragma unroll
          for (int u = 0; u < EU; ++u) {
            const int j = j0 + EU * grp + u;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float d = qv[i] - tv[u][i];
              const float dd = d * d;
              acc = acc + dd;
            }
            const float dj = __builtin_sqrtf(exact_reduce16(acc, slot, lane));
            if (j < nt) best2_insert(dj, j, b0d, b0i, b1d, b1i);
          }
        }
      } else {
        for (int j = grp; j < nt; j += NTHR / 16) {
          const float dj = exact_dist16(Q + (size_t)(q0 + sq) * dim, T + (size_t)j * dim, dim, slot, lane);
          best2_insert(dj, j, b0d, b0i, b1d, b1i);
        }
      }
      if (slot == 0) { sm.scan[grp * 2] = Cand{b0d, b0i}; sm.scan[grp * 2 + 1] = Cand{b1d, b1i}; }
      __syncthreads();
      if (tid == 0) {
        float r0d = __builtin_inff(), r1d = __builtin_inff();
        int r0i = 0x7FFFFFFF, r1i = 0x7FFFFFFF;
        for (int k = 0; k < 2 * (NTHR / 16); ++k) {
          const Cand cnd = sm.scan[k];
          if (cnd.i == 0x7FFFFFFF) continue;
          best2_insert(cnd.d, cnd.i, r0d, r0i, r1d, r1i);
        }
        emit(sq, r0i == 0x7FFFFFFF ? -1 : r0i, r0d, r1d);
        if (m.stats) atomicAdd(m.stats, 1);
      }
      __syncthreads();
    }
  }
  // ---- (3) ticket: the workgroup that arrives last for this pair finalizes it
  D2FE_STAMP(m.stamps, blockIdx.x, 6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave: its write-through record stores have left
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(m.ticket + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == per_pair - 1;
    if (last) {
      __hip_atomic_store(m.ticket + pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch on this scratch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    sm.last = last;
  }
  __syncthreads();
  D2FE_STAMP(m.stamps, blockIdx.x, 7);
  if (sm.last) { match_finalize_pair<NW>(m, pair, sm); D2FE_STAMP(m.stamps, blockIdx.x, 8); }
}

template <int NW>
static hipError_t launch_match_nw(const MatchArgs& m, int tiles, long nwg, hipStream_t s) {
  // 256-D: no column checks; shorter descriptors (PCA'd SuperPoint descriptors, superpoint_onnx.cpp:47-53): fragment arrays sized for the dimension
  const dim3 grid((unsigned)nwg), block(64 * NW);
  if (m.dim == MAXDIM) hipLaunchKernelGGL((match_kernel<KST, true, NW>), grid, block, 0, s, m, tiles);
  else if (m.dim <= 64) hipLaunchKernelGGL((match_kernel<4, false, NW>), grid, block, 0, s, m, tiles);
  else if (m.dim <= 128) hipLaunchKernelGGL((match_kernel<8, false, NW>), grid, block, 0, s, m, tiles);
  else hipLaunchKernelGGL((match_kernel<KST, false, NW>), grid, block, 0, s, m, tiles);
  return hipGetLastError();
}

hipError_t launch_match(const MatchArgs& m, hipStream_t s) {
  if (m.dim > MAXDIM || (m.dim & 3) || m.max_n > MATCH_MAXN || m.max_n < 1 || m.npairs < 1 || !m.ticket) return hipErrorInvalidValue;
  const int tiles = (m.max_n + MQ - 1) / MQ;
  const long nwg = (long)tiles * 2 * m.npairs;
  if (nwg > 0x7FFFFFFF) return hipErrorInvalidValue;
  // four waves per workgroup split a tile's train rows four ways (half the latency of a workgroup) but only two such workgroups fit a
  // CU (256 registers per lane): used while the whole launch is resident at once
  const int ncu = m.ncu > 0 ? m.ncu : 256;
  return nwg <= 2L * ncu ? launch_match_nw<4>(m, tiles, nwg, s) : launch_match_nw<2>(m, tiles, nwg, s);
}

}  // namespace d2fe

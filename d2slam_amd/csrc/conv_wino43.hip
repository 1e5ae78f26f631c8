// conv_wino43.hip -- EXPERIMENTAL (development library only): 3x3 / stride 1 / pad 1 convolutions as Winograd F(4,3) down the columns x F(2,3) along the
// rows on the fp32 matrix pipe: output tiles of 4 x 2 pixels, 6 x 4 = 24 transform positions, 3 multiplies per output where F(2x2,3x3) (conv_wino.hip)
// has 4 and the direct form 9.  Why it is worth building: tools/wino4_study.py / profiles/r04_wino4_study.json -- the larger transforms do not cost index
// parity (0 of 160 000 keypoints against an fp64 truth, score error 1.4e-6 against 1.06e-6 for a direct fp32 convolution).
//
// Same replacement as conv_wino.hip (the TensorRT engine execution at d2frontend/src/CNN/superpoint_tensorrt.cpp:150 for the 3x3 layers with Cin >= 64);
// the evaluation order is fixed here and restated by orc_conv3x3_wino43 (oracle/d2fe_oracle.c): outputs are bit-identical to that restatement.
//
// STATUS (round 4): bit-identical to the restatement for every layer shape of the network (tests/test_wino43.py).  Per layer at 64 images (conv2a / conv3b / conv4a):
// 1.66 / 1.51 / 0.40 ms -- 3-9 % faster than conv_wino.hip with a static split of the work items, 7-18 % behind its tuned production path (dynamic claiming, two
// phase-shifted workgroups per CU: 1.36-1.43 / 1.36-1.39 / 0.37-0.38).  D2FE_ABLATE (tools/gpu_w43.sh): without the epilogue 1.40 / 1.42 / 0.36 ms -- all twelve
// waves of a CU reach the epilogue together, so the matrix pipe idles through it; K loop alone 1.06 ms.  What it needs is in DESIGN.md section 7.6.
//
// Work split (what changes against conv_wino.hip, whose row transform, accumulator layout, LDS slot pattern and U stream are kept):
//   * a wave owns ONE ROW i of the 6 x 4 transform domain: the 4 positions (i, 0..3) x 32 tiles x 32 output channels = 64 accumulators; a workgroup is
//     TWELVE waves -- the rows i = 0..5 for two 32-channel groups -- sharing one staged patch: 16 x 16 output pixels (4 x 8 tiles) x 64 channels, one workgroup
//     per CU = exactly three waves on every SIMD (six-wave workgroups, two per CU, land 4 / 2 / 3 / 3 on the SIMDs of three CUs out of four -- measured with
//     D2FE_ABLATE=256 -- and the SIMD with four waves sets the pace)
//   * row i of B^T d B (F(4,3)) needs three or four of the six window rows:  t0 = 4 d0 - 5 d2 + d4,  t1 = -4 (d1 + d2) + (d3 + d4),  t2 = 4 (d1 - d2) + (d4 - d3),
//     t3 = 2 (d3 - d1) + (d4 - d2),  t4 = 2 (d1 - d3) + (d4 - d2),  t5 = 4 d1 - 5 d3 + d5  (one fma each on top of the sums; see the oracle for the exact nesting)
//   * LDS chunk = 8 channels of the 18 x 18 patch as [channel quad 2][phase plane 8 = (row mod 4, column mod 2)][5 rows x 12 (9 used)][4 channels]: window
//     position (dy, dx) of all 32 tiles sits in ONE plane at 12 ty + tx -- the conflict-free pattern of conv_wino.hip; ring of three chunk buffers (LDS-DMA)
//   * output transform: along the row in registers (s = M A, F(2,3)), then the six rows of a channel group meet in LDS (48 KB per group) and its six waves
//     share the tiles: y = A4^T s, bias, ReLU, 2x2 max-pool (a 4 x 2 tile holds two pool windows) and the stores
#include "conv_common.h"

#include <cstdio>

#ifdef D2FE_DEVTOOLS      /* the whole file: an experiment of the development library, not part of libd2fe_hip.so */

namespace d2fe {

namespace {

constexpr int QCHUNK = 2 * 8 * 64 * 4;      // floats per chunk buffer (16 KiB)
constexpr int QWR = 3;                      // ring depth
constexpr int QROW = 12;                    // plane row stride in slots (9 used)
constexpr int QXCH = 6 * 16 * 64 * 2;       // floats of a channel group's exchange area: [row 6][accumulator register 16][lane 64] x (b0, b1) (48 KiB)

struct QItem { int img, by, bx, cb; };
__device__ __forceinline__ QItem q_decode(int t, int nbx, int nby, int ncb) {
  QItem r;
  r.cb = t % ncb; t /= ncb;
  r.bx = t % nbx; t /= nbx;
  r.by = t % nby;
  r.img = t / nby;
  return r;
}
typedef __attribute__((address_space(3))) void qlvoid_t;
__device__ __forceinline__ f32x4 q_buf_load_f32x4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  f32x4 o;
  __builtin_memcpy(&o, &v, 16);
  return o;
}
__device__ __forceinline__ void q_buf_load_lds16(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (qlvoid_t*)lds, 16, voff, soff, 0, 0);
}

}  // namespace

#ifdef D2FE_DEVTOOLS
__device__ unsigned g_w43_hw[1024][12][2];      // D2FE_ABLATE bit 256: HW_ID / XCC_ID of every wave of the first 1024 workgroups (which SIMDs do the waves get?)
#endif

template <int CIN, bool POOL, int I, int NU>
__device__ __forceinline__ void w43_body(const ConvArgs& a, int nbx, int nby, int ncb, int total, float* lds, const int wv) {
  constexpr int NCH = CIN / 8, KSTEPS = CIN / 2;
  // wave wv = 6 G + I: row I of the transform domain for the 32-channel group G of the item's 64 channels; it copies the (quad, plane) units wv, wv + 12 (< 16): NU of them
  const int G = wv >= 6 ? 1 : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int aH = a.H, aW = a.W, in_cs = a.in_cstride;
  const int tstride = gridDim.x;

  // ---- LDS-DMA descriptors of an item: unit u = quad (u >> 3), plane (u & 7) = (row phase << 1) | column phase; lane -> slot (Y = lane / 12, X = lane % 12)
  struct DmaItem { __amdgpu_buffer_rsrc_t rsrc; int off[NU]; };
  const int in_bytes = aH * aW * in_cs * 4;
  auto dma_prepare = [&](const QItem& T) {
    DmaItem d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)T.img * a.in_img_stride), 0, in_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int pl = (wv + 12 * j) & 7;
      const int Y = lane / QROW, X = lane % QROW;
      const int pr = 4 * Y + (pl >> 1), pc = 2 * X + (pl & 1);            // patch row / column, 0 .. 17
      const int gy = T.by * 16 - 1 + pr, gx = T.bx * 16 - 1 + pc;
      const bool ok = Y < 5 && X < 9 && pr < 18 && pc < 18 && gy >= 0 && gy < aH && gx >= 0 && gx < aW;
      d.off[j] = ok ? (gy * aW + gx) * in_cs * 4 : (int)0x80000000;       // beyond num_records: the load returns 0
    }
    return d;
  };
  auto dma_issue = [&](const DmaItem& d, int ch, int buf) {
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = wv + 12 * j;
      q_buf_load_lds16(d.rsrc, lds + buf * QCHUNK + u * 256, d.off[j], (a.in_coff + (u >> 3) * 4) * 4 + ch * 32);
    }
  };

  // ---- per-lane constants: tile <-> MFMA row as in conv_wino.hip
  const int trow = lane & 31, q8 = trow >> 2, hh = lane >> 5;
  const int ty = 2 * (q8 >> 2) + (__builtin_popcount(q8) & 1), tx = 4 * ((q8 >> 1) & 1) + (trow & 3);
  const int rd_off = hh * (8 * 256) + (ty * QROW + tx) * 4;     // quad hh, plane 0, this lane's tile origin
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, ncb * 2 * KSTEPS * 6144, 0x00020000);
  auto u_ptr = [&](const QItem& T) { return (T.cb * 2 + G) * KSTEPS * 6144; };      // the item's two 32-channel streams
  const int lane16 = lane * 16 + I * 1024;       // row I of the 6 KiB k-step record

  f32x16 acc[4];
  f32x4 t[4];              // t[dx]: row I of B4^T d for window column dx, the quad's 4 channels
  f32x4 ub[4];             // U fragments of k-step (slot): one float per position (I, 0..3)
  f32x2 vp[4];             // the four positions' A operands for a pair of k-steps

  auto win = [&](const float* p, int dy, int dx) {
    return *reinterpret_cast<const f32x4*>(p + ((dy & 3) * 2 + (dx & 1)) * 256 + ((dy >> 2) * QROW + (dx >> 1)) * 4);
  };
  // packed fp32 arithmetic by name (the compiler scalarises <2 x float> additions): the column transform is the wave's largest block of vector work
  auto padd = [](f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
  auto psub = [](f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y)); return d; };
  auto pfma = [](float c, f32x2 x, f32x2 y) { return __builtin_elementwise_fma(f32x2{c, c}, x, y); };
  auto lo2 = [](f32x4 v) { return f32x2{v[0], v[1]}; };
  auto hi2 = [](f32x4 v) { return f32x2{v[2], v[3]}; };
  auto cat = [](f32x2 a, f32x2 b) { return f32x4{a[0], a[1], b[0], b[1]}; };
  auto read_t = [&](int buf) {
    const float* p = lds + buf * QCHUNK + rd_off;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      if constexpr (I == 0 || I == 5) {
        const f32x4 da = win(p, I == 0 ? 0 : 1, dx), db = win(p, I == 0 ? 2 : 3, dx), dc = win(p, I == 0 ? 4 : 5, dx);      // t = 4 da - 5 db + dc
        t[dx] = cat(pfma(4.f, lo2(da), pfma(-5.f, lo2(db), lo2(dc))), pfma(4.f, hi2(da), pfma(-5.f, hi2(db), hi2(dc))));
      } else {
        const f32x4 d1 = win(p, 1, dx), d2 = win(p, 2, dx), d3 = win(p, 3, dx), d4 = win(p, 4, dx);
        if constexpr (I == 1) t[dx] = cat(pfma(-4.f, padd(lo2(d1), lo2(d2)), padd(lo2(d3), lo2(d4))), pfma(-4.f, padd(hi2(d1), hi2(d2)), padd(hi2(d3), hi2(d4))));
        else if constexpr (I == 2) t[dx] = cat(pfma(4.f, psub(lo2(d1), lo2(d2)), psub(lo2(d4), lo2(d3))), pfma(4.f, psub(hi2(d1), hi2(d2)), psub(hi2(d4), hi2(d3))));
        else if constexpr (I == 3) t[dx] = cat(pfma(2.f, psub(lo2(d3), lo2(d1)), psub(lo2(d4), lo2(d2))), pfma(2.f, psub(hi2(d3), hi2(d1)), psub(hi2(d4), hi2(d2))));
        else t[dx] = cat(pfma(2.f, psub(lo2(d1), lo2(d3)), psub(lo2(d4), lo2(d2))), pfma(2.f, psub(hi2(d1), hi2(d3)), psub(hi2(d4), hi2(d2))));
      }
    }
  };
  auto load_u = [&](int slot, int up, int ks) { ub[slot] = q_buf_load_f32x4(u_rsrc, lane16, up + ks * 6144); };

  float* xch = lds + QWR * QCHUNK + G * QXCH;

  int icur = blockIdx.x, inxt = icur + tstride;
  QItem cur = q_decode(icur, nbx, nby, ncb);
  QItem nxt = inxt < total ? q_decode(inxt, nbx, nby, ncb) : cur;
  DmaItem dcur = dma_prepare(cur), dnxt = dma_prepare(nxt);
  int ucur = u_ptr(cur), unxt = u_ptr(nxt);
  dma_issue(dcur, 0, 0);
  dma_issue(dcur, 1, 1);
  load_u(0, ucur, 0);
  load_u(1, ucur, 1);
  load_u(2, ucur, 2);

  // chunk 0 of the walk: landed, visible, its window rows read and reduced (the loop below does the same for chunk g + 1 in front of the last k-step of chunk g)
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  __syncthreads();
  if (NCH > 2) dma_issue(dcur, 2, 2);
  read_t(0);

  int g = 0;               // chunks walked so far (ring position)
#pragma unroll 1
  for (; icur < total;) {
    const bool has_next = inxt < total;
    const float bias = a.bias[(cur.cb * 2 + G) * 32 + (lane & 31)];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch, ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kt = ch * 4 + j + 3;                          // U runs three k-steps ahead, across the item boundary
        if (kt < KSTEPS) load_u((j + 3) & 3, ucur, kt); else load_u((j + 3) & 3, unxt, kt - KSTEPS);
        // row transform (F(2,3)) for the k-step pair (j, j + 1) at the even one: four packed instructions instead of eight
        if ((j & 1) == 0) {
          const f32x2 t0 = j ? hi2(t[0]) : lo2(t[0]), t1 = j ? hi2(t[1]) : lo2(t[1]), t2 = j ? hi2(t[2]) : lo2(t[2]), t3 = j ? hi2(t[3]) : lo2(t[3]);
          vp[0] = psub(t0, t2); vp[1] = padd(t1, t2); vp[2] = psub(t2, t1); vp[3] = psub(t1, t3);
        }
        const float v0 = vp[0][j & 1], v1 = vp[1][j & 1], v2 = vp[2][j & 1], v3 = vp[3][j & 1];
        // the scheduling fences of this loop are load-bearing: the compiler does not see the LDS-DMA copies as writes to LDS, and without the fences it moved the
        // window reads of the next chunk (results wrong on every layer; measured no faster either)
        __builtin_amdgcn_sched_barrier(0);
        if (j == 3) {
          // the last k-step's operands are in registers: t is free.  Chunk g + 1 (copied two iterations ago) must have landed: younger than its copies are the
          // copies of chunk g + 2 (NU) and this iteration's four U loads -- loads complete in order
          const bool more = ch + 1 < NCH || has_next;
          if (more) {
            if constexpr (NU == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            __syncthreads();     // every wave's part of chunk g + 1 is in LDS; every wave has read chunk g out of its buffer (one iteration ago)
            const int c3 = ch + 3;         // chunk g + 3 of the walk -> the buffer of chunk g
            if (D2FE_ABL(a, 4)) {}
            else if (c3 < NCH) dma_issue(dcur, c3, g % QWR);
            else if (has_next && c3 - NCH < NCH) dma_issue(dnxt, c3 - NCH, g % QWR);
            if (!D2FE_ABL(a, 2)) read_t((g + 1) % QWR);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, ub[j][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, ub[j][1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, ub[j][2], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, ub[j][3], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: s = M A (F(2,3), in registers), the six rows meet in LDS, waves 0..3 finish -------------------------------------------
    float* out = a.out + (size_t)cur.img * a.out_img_stride + a.out_coff;
    const int cs = a.out_cstride;
    const int co = (cur.cb * 2 + G) * 32 + (lane & 31);
    const bool cok = co < a.cout_real;
    if (!D2FE_ABL(a, 1)) {
      // every wave: its row of s for all 16 accumulator registers, (b0, b1) as one ds_write_b64
      f32x2* xw = reinterpret_cast<f32x2*>(xch) + (I * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
        xw[r * 64] = f32x2{(m0 + m1) + m2, (m1 - m2) - m3};
      }
      __syncthreads();
      // the six waves of a channel group share the 16 registers: wave I finishes registers I, I + 6, I + 12 (< 16).  No barrier behind the reads: the
      // next writer of the exchange area is the next item's epilogue, NCH chunk barriers away
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int r = I + 6 * k;
        if (r < 16) {
          const int q = 2 * (r >> 2) + hh;                                  // tile = MFMA row (r & 3) + 8 (r >> 2) + 4 hh
          const int tyr = 2 * (q >> 2) + (__builtin_popcount(q) & 1), txr = 4 * ((q >> 1) & 1) + (r & 3);
          f32x2 sv[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) sv[i] = (reinterpret_cast<const f32x2*>(xch) + (i * 16 + r) * 64)[lane];
          float y[4][2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            y[0][b] = ((sv[0][b] + sv[1][b]) + sv[2][b]) + (sv[3][b] + sv[4][b]);
            y[1][b] = __builtin_fmaf(2.f, sv[3][b] - sv[4][b], sv[1][b] - sv[2][b]);
            y[2][b] = __builtin_fmaf(4.f, sv[3][b] + sv[4][b], sv[1][b] + sv[2][b]);
            y[3][b] = __builtin_fmaf(8.f, sv[3][b] - sv[4][b], sv[1][b] - sv[2][b]) + sv[5][b];
          }
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int b = 0; b < 2; ++b) { const float v = y[p][b] + bias; y[p][b] = v > 0.f ? v : 0.f; }
          const int oy = cur.by * 16 + 4 * tyr, ox = cur.bx * 16 + 2 * txr;
          if constexpr (POOL) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const float v = fmaxf(fmaxf(y[2 * pp][0], y[2 * pp][1]), fmaxf(y[2 * pp + 1][0], y[2 * pp + 1][1]));
              if (cok && !D2FE_ABL(a, 8) && oy + 2 * pp + 1 < aH && ox + 1 < aW) out[((size_t)((oy >> 1) + pp) * (aW >> 1) + (ox >> 1)) * cs + co] = v;
            }
          } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                if (cok && !D2FE_ABL(a, 8) && oy + p < aH && ox + b < aW) out[((size_t)(oy + p) * aW + ox + b) * cs + co] = y[p][b];
          }
        }
      }
    }

    cur = nxt; dcur = dnxt; ucur = unxt;
    icur = inxt; inxt += tstride;
    if (inxt < total) { nxt = q_decode(inxt, nbx, nby, ncb); dnxt = dma_prepare(nxt); unxt = u_ptr(nxt); }
  }
}

template <int CIN, bool POOL>
__global__ __launch_bounds__(768, 3) void conv_wino43_kernel(ConvArgs a, int nbx, int nby, int ncb, int total) {
  extern __shared__ __attribute__((aligned(16))) float q_lds[];
  if ((int)blockIdx.x >= total) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef D2FE_DEVTOOLS
  if (D2FE_ABL(a, 256) && (threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_w43_hw[blockIdx.x][wv][0] = hwid; g_w43_hw[blockIdx.x][wv][1] = xcc;
  }
#endif
  // the six rows of the transform domain run different (compile-time) column arithmetic; waves 0..3 copy two units of a chunk, the others one
  switch (wv) {
    case 0: w43_body<CIN, POOL, 0, 2>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 1: w43_body<CIN, POOL, 1, 2>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 2: w43_body<CIN, POOL, 2, 2>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 3: w43_body<CIN, POOL, 3, 2>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 4: case 10: w43_body<CIN, POOL, 4, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 5: case 11: w43_body<CIN, POOL, 5, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 6: w43_body<CIN, POOL, 0, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 7: w43_body<CIN, POOL, 1, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
    case 8: w43_body<CIN, POOL, 2, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
    default: w43_body<CIN, POOL, 3, 1>(a, nbx, nby, ncb, total, q_lds, wv); break;
  }
}

hipError_t launch_conv_wino43(int cin, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  if (!relu || cout_pad % 64 || (a.in_cstride & 3) || (a.in_coff & 3) || (cin != 64 && cin != 128)) return hipErrorInvalidValue;
  if ((long)a.H * a.W * a.in_cstride * 4 >= (1l << 31) || (long)a.H * a.W * a.out_cstride * 4 >= (1l << 31)) return hipErrorInvalidValue;
  const int nbx = (a.W + 15) / 16, nby = (a.H + 15) / 16, ncb = cout_pad / 64;
  const int total = nbx * nby * ncb * a.n_img;
  const int ncu = a.ncu > 0 ? a.ncu : 256;
  const int grid = total < ncu ? total : ncu;
  const size_t lds = (size_t)(QWR * QCHUNK + 2 * QXCH) * sizeof(float);
#define D2FE_W43_K(K)                                                                                        \
  do {                                                                                                      \
    auto k = K;                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                          \
    hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, s, a, nbx, nby, ncb, total);                          \
  } while (0)
  if (cin == 64 && pool) D2FE_W43_K((conv_wino43_kernel<64, true>));
  else if (cin == 64) D2FE_W43_K((conv_wino43_kernel<64, false>));
  else if (pool) D2FE_W43_K((conv_wino43_kernel<128, true>));
  else D2FE_W43_K((conv_wino43_kernel<128, false>));
#undef D2FE_W43_K
#ifdef D2FE_DEVTOOLS
  if (a.ablate & 256) {
    static int dumped = 0;
    static unsigned hw[1024][12][2];
    if (!dumped++ && hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_w43_hw), sizeof(hw)) == hipSuccess) {
      // per (xcc, se, sh, cu): how many waves landed on each SIMD
      int hist[5][5] = {};       // [waves on the fullest SIMD][waves on the emptiest SIMD] over CUs
      const int nwg = grid < 1024 ? grid : 1024;
      for (int b = 0; b < nwg; ++b) {
        int simd[4] = {0, 0, 0, 0}, peers = 0;
        for (int c = 0; c < nwg; ++c) {
          if ((hw[b][0][0] & 0xff00) != (hw[c][0][0] & 0xff00) || (hw[b][0][1] & 15) != (hw[c][0][1] & 15)) continue;
          ++peers;
          for (int w = 0; w < 12; ++w) ++simd[(hw[c][w][0] >> 4) & 3];
        }
        if (b < 4) fprintf(stderr, "wg %d: %d workgroups on its CU, waves per SIMD %d %d %d %d; own waves on SIMDs %u %u %u %u %u %u\n", b, peers, simd[0], simd[1], simd[2], simd[3],
                           (hw[b][0][0] >> 4) & 3, (hw[b][1][0] >> 4) & 3, (hw[b][2][0] >> 4) & 3, (hw[b][3][0] >> 4) & 3, (hw[b][4][0] >> 4) & 3, (hw[b][5][0] >> 4) & 3);
        int mx = 0, mn = 99;
        for (int i = 0; i < 4; ++i) { mx = simd[i] > mx ? simd[i] : mx; mn = simd[i] < mn ? simd[i] : mn; }
        if (mx < 5 && mn < 5) ++hist[mx][mn];
      }
      for (int mx = 0; mx < 5; ++mx) for (int mn = 0; mn < 5; ++mn) if (hist[mx][mn]) fprintf(stderr, "CUs (counted per workgroup) with max %d / min %d waves per SIMD: %d\n", mx, mn, hist[mx][mn]);
    }
  }
#endif
  return hipGetLastError();
}

// host-side weight transform + packing:  [32-channel group][k-step][row i of the 6 x 4 domain][lane] float4,
//   float4[e] = U[xi = 4 i + e][ci = 8 (ks / 4) + 4 (lane >> 5) + ks % 4][co = group * 32 + (lane & 31)],  U = (float)(G4 g G2^T) evaluated in double
size_t packed_weight_floats_wino43(int cout_pad, int cin) { return (size_t)24 * cout_pad * cin; }

void pack_weights_wino43(const float* w, int cout, int cin, int cout_pad, float* dst) {
  static const double G4[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  static const double G2[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int ksteps = cin / 2;
  for (int grp = 0; grp < cout_pad / 32; ++grp)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int i = 0; i < 6; ++i)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int co = grp * 32 + (lane & 31), ci = 8 * (ks / 4) + 4 * (lane >> 5) + (ks % 4);
            double sum = 0.0;
            if (co < cout) {
              const float* g = w + ((size_t)co * cin + ci) * 9;
              for (int p = 0; p < 3; ++p)
                for (int r = 0; r < 3; ++r) sum += G4[i][p] * G2[e][r] * (double)g[p * 3 + r];
            }
            dst[((((size_t)grp * ksteps + ks) * 6 + i) * 64 + lane) * 4 + e] = (float)sum;
          }
}

}  // namespace d2fe

#endif  // D2FE_DEVTOOLS

// conv_wino.hip -- 3x3 / stride 1 / pad 1 convolutions as Winograd F(2x2,3x3) on the fp32 matrix pipe (precision mode 2).
//
// Replaces the TensorRT engine execution at d2frontend/src/CNN/superpoint_tensorrt.cpp:150 for the eight 3x3 layers with
// Cin >= 64 (network: d2frontend/superpoint.ipynb:300-374).  TensorRT itself picks Winograd kernels for such layers; the
// reference fixes no accumulation order, so this mode fixes one (below) and the oracle restates it (orc_conv3x3_wino):
// outputs are bit-identical to that restatement and within ~1e-6 of the direct-convolution chain of the exact mode.
//
// Arithmetic per 2x2 output tile and (ci, co):  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A  -- 16 multiplies instead of 36.
// The sum over ci for each of the 16 transform positions xi = (i,j) is a GEMM  M_xi[tile][co] = V_xi[tile][ci] U_xi[ci][co]:
// v_mfma_f32_32x32x2_f32 with A = 32 tiles x 2 channels, B = 2 channels x 32 output channels.
//
// Fixed evaluation order (what the oracle restates):
//   U = (float)(G g G^T) evaluated in double, term order (a, b) ascending;
//   B^T d B: rows first (t0 = d0-d2, t1 = d1+d2, t2 = d2-d1, t3 = d1-d3 down each column), then the same along each row;
//   M: one fmaf chain from +0, channels in the order 0,4,1,5,2,6,3,7 of every block of 8 (lanes 0-31 / 32-63 of the MFMA
//      read the two 16-byte channel quads of a pixel);
//   A^T M A: s_i0 = (m_i0 + m_i1) + m_i2, s_i1 = (m_i1 - m_i2) - m_i3; y_0b = (s_0b + s_1b) + s_2b, y_1b = (s_1b - s_2b) - s_3b;
//   y + bias, ReLU, max-pool.
//
// Work split.  Measured on gfx950 (tools/ubench/mfma_f32_valu.hip): with ONE wave on a SIMD a VALU instruction next to
// v_mfma_f32_32x32x2_f32 is not hidden (64.6 cycles per MFMA bare, 77.5 with two VALU per gap, 87.5 with four); with TWO waves the
// partner's MFMAs cover it (64.2 bare, 69.8 with two or four per gap).  So a wave must fit 2 per SIMD: it owns ONE ROW i of the 4x4
// transform domain: 32 tiles (4 x 8) x 64 output channels x the 4 positions (i, 0..3) = 128 accumulator registers, two
// waves per SIMD.  Row i of B^T d B needs two rows of the input window and 4 + 4 subtractions per channel, which feed 8 MFMAs
// (1 VALU per MFMA and 8 instead of 12 or 16 LDS reads per chunk; a wave with all 16 positions needs 256 accumulators, i.e. one
// wave per SIMD).  A workgroup is the 4 waves i = 0..3 and handles 8 x 16 output pixels x 64 channels, two workgroups per CU.  The
// output transform needs all four rows: after the K loop each wave forms its row of s = M A, the waves swap rows through
// LDS (each keeps a quarter of the tiles, 48 floats per lane go each way) and finish A^T s, bias, ReLU, the 2x2 max-pool (the
// Winograd tile IS the pool window) and the stores for their 8 of the 32 tiles.  The first k-step of an item multiplies into
// a zero C operand instead of zeroing 128 registers.
//
// Data movement.  Persistent workgroups walk a list of work items; the K loop runs over chunks of 8 input channels.  A chunk of
// the 10 x 18 input patch is brought in by LDS-DMA (buffer_load_dwordx4 ... lds: 16 bytes = 4 channels of one pixel per lane,
// no VGPR round trip; out-of-image pixels are out-of-range buffer offsets, which read as 0) into a ring of three 8 KiB
// buffers that runs on across work items.  LDS layout of a chunk: [channel quad 2][pixel parity plane 4][5 rows x 12 (9 used)]
// [4 channels]: the 32 tiles of a wave read the same (dy,dx) of their 4x4 input window from ONE parity plane at positions
// 12*ty + tx, and the tile -> MFMA-row assignment (below) makes that conflict-free for ds_read_b128's lane groups.  U streams
// from L2 in MFMA lane order (8 values per lane per k-step, requested three k-steps ahead, running on across work items).
// With FUSE (conv1b, D2FE_FUSE1A=1) the copies are replaced by conv1a evaluated on the matrix pipe into a whole-K patch.
#include "conv_common.h"

#include <cstdio>
#include <type_traits>

namespace d2fe {

namespace {

constexpr int WR = 3;                       // ring depth (four buffers measure the same and would fill the LDS to the last byte)
constexpr int WCHUNK = 2 * 4 * 64 * 4;      // floats per chunk buffer (8 KiB): [quad 2][plane 4][64 slots][4 channels]
constexpr int WROW = 12;                    // plane row stride in positions (9 used)
constexpr int WXCH = 4 * 12 * 64 * 4;       // floats of the epilogue exchange area: [wave 4][12 float4][64 lanes]

struct WItem { int img, by, bx, cb; };
template <int V> using IC = std::integral_constant<int, V>;

__device__ __forceinline__ WItem w_decode(int t, int nbx, int nby, int ncb) {
  WItem r;
  r.cb = t % ncb; t /= ncb;
  r.bx = t % nbx; t /= nbx;
  r.by = t % nby;
  r.img = t / nby;
  return r;
}

typedef __attribute__((address_space(3))) void lvoid_t;

// buffer-addressed loads (SGPR resource + 32-bit lane offset + SGPR offset), kept in plain device functions
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  f32x4 o;
  __builtin_memcpy(&o, &v, 16);
  return o;
}
// Cache policy of the STREAMED data (aux bit 1 = nt on gfx940+).  The ablation sweep (profiles/r06_wino_ring_ablation.txt) shows the U loads as the largest exposed cost
// of the K loop (13 %), so non-temporal patch copies / output stores were tried to leave the L2 ways to U: measured SLOWER (same-box A/B of bench.py --single-mode, round 6: value 2327 -> 2098 with
// both, 2309 -> 2032 with the loads alone, 2317 -> 2308 with the stores alone: the second 64-channel walk and the neighbouring items re-read the patches from L2).  Kept as compile-time A/B switches, default 0
#ifndef D2FE_WINO_LOAD_AUX
#define D2FE_WINO_LOAD_AUX 0
#endif
#ifndef D2FE_WINO_STORE_AUX
#define D2FE_WINO_STORE_AUX 0
#endif
__device__ __forceinline__ void buf_store_f32(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, D2FE_WINO_STORE_AUX);
}
__device__ __forceinline__ void buf_load_lds16(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {   // LDS-DMA, 16 B per lane
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lvoid_t*)lds, 16, voff, soff, 0, D2FE_WINO_LOAD_AUX);
}

}  // namespace

#ifdef D2FE_DEVTOOLS
// development library, ABL 256: workgroup 0 / wave 0 records shader-clock marks (s_memtime) for its first 4 items
__device__ long long g_wino_trace[4][16][6];
__device__ long long g_wino_pair[2][64][3];       // ABL 256: the two workgroups of ONE CU: [wave slot parity][item][K start, epilogue start, epilogue end]
__device__ unsigned g_wino_hw[1024][2];      // ABL 256: HW_ID / XCC_ID of wave 0 of every workgroup (which two share a CU?)
#endif

// Tile <-> MFMA row.  Row t (= lane & 31 of the A operand, = (r&3) + 8*(r>>2) + 4*(lane>>5) of an accumulator register r)
// holds tile  ty = 2*(q>>2) + parity(q),  tx = 4*((q>>1)&1) + (t&3),  q = t>>2.  ds_read_b128 serves a wave in the lane groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31} (+32): with this assignment a group holds tile rows {0,2} or {1,3}, whose
// positions 12*ty + tx cover every residue mod 16 exactly once -- one 16-byte slot per lane, no bank conflict.
// NT: 32-channel halves of the output channels a wave carries.  NT = 2 (default): the 8 x 16 pixel x 64 channel items described above (128
// accumulators, two workgroups per CU).  NT = 1 (ring kernels only, D2FE_WINO_NT=1): 8 x 16 pixels x 32 channels per item -- 64 accumulators,
// 138 registers and 48 KB of LDS per workgroup, so THREE workgroups share a CU and two waves of a SIMD are in their K loops while the third is
// in its MFMA-free epilogue.  The price is the input transform, the LDS window reads and the patch copies once per 32 instead of per 64
// output channels -- measured (round 3, profiles/r03_wino_nt_ab.txt): 6-8 % slower on every layer, bit-identical.  Kept as the record of
// the structural experiment VERDICT r02 asked for; same arithmetic per output, so the oracle does not change.
template <int CIN, bool POOL, bool RELU, int ABL, int I, bool FUSE, int NT>
__device__ __forceinline__ void wino_body(const ConvArgs& a, int nbx, int nby, int ncb, int total, float* wlds) {
  static_assert(NT == 1 || NT == 2, "a wave carries one or two 32-channel halves");
  static_assert(!FUSE || NT == 2, "the fused conv1b kernel keeps 64 channels per item (its LDS patch allows two workgroups per CU either way)");
  constexpr int WXCHN = 4 * 6 * NT * 64 * 4;  // floats of the exchange area: [wave 4][3 other waves x 2 tile pairs x NT float4][64 lanes]
  constexpr int NCH = CIN / 8;               // chunks per work item
  constexpr int KSTEPS = CIN / 2;
  // FUSE (conv1b): the whole 64-channel patch is produced in LDS by conv1a on the matrix pipe instead of being copied in; its
  // planes are 272 floats apart (16 floats of padding keep the four parity planes on different banks for the float4 writes)
  constexpr int PL = FUSE ? 272 : 256;       // floats per parity plane
  constexpr int CHF = 2 * 4 * PL;            // floats per 8-channel chunk
  static_assert(!FUSE || CIN == 64, "the fused prologue is conv1a -> conv1b");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int aH = a.H, aW = a.W, in_cs = a.in_cstride;

  // Work items: a workgroup starts with items blockIdx and blockIdx + gridDim and then either strides on (a.work_ctr == null) or CLAIMS
  // the next index from a device counter.  Claiming matters: of the two workgroups of a CU the one with the older waves gets ~20 % more
  // of the matrix pipe (issue arbitration is by age; measured with D2FE_ABLATE=256), so with an equal split it is done early and its
  // partner runs the tail alone, every MFMA-free phase exposed.
  const int tstride = gridDim.x;
  const bool dyn = a.work_ctr != nullptr && total >= 48 * tstride;      // short walks (10-40 items per workgroup) gain nothing from claiming (measured)
  int icur = blockIdx.x, inxt = icur + tstride;                  // indices >= total: no such item
  int* claim_slot = reinterpret_cast<int*>(FUSE ? wlds + 8 * (2 * 4 * (FUSE ? 272 : 256)) + 64 : wlds + WR * WCHUNK + WXCHN);

  // ---- LDS-DMA: wave w copies channel quad (w>>1), parity planes 2(w&1), 2(w&1)+1 (64 slots = one instruction each) ---------
  struct DmaItem { __amdgpu_buffer_rsrc_t rsrc; int off[2]; };
  const int in_bytes = aH * aW * in_cs * 4;
  auto dma_prepare = [&](const WItem& T) {
    DmaItem d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)T.img * a.in_img_stride), 0, in_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int plane = 2 * (wave & 1) + j;
      const int Y = lane / WROW, X = lane % WROW;
      const int gy = T.by * 8 - 1 + 2 * Y + (plane >> 1), gx = T.bx * 16 - 1 + 2 * X + (plane & 1);
      const bool ok = Y < 5 && X < 9 && gy >= 0 && gy < aH && gx >= 0 && gx < aW;
      d.off[j] = ok ? (gy * aW + gx) * in_cs * 4 : (int)0x80000000;     // beyond num_records: the load returns 0
    }
    return d;
  };
  const int dma_soff0 = (a.in_coff + (wave >> 1) * 4) * 4;
  auto dma_issue = [&](const DmaItem& d, int ch, int buf) {
    if constexpr ((ABL & 1) != 0) return;
    float* dst = wlds + buf * WCHUNK + ((wave >> 1) * 4 + 2 * (wave & 1)) * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) buf_load_lds16(d.rsrc, dst + j * 256, d.off[j], dma_soff0 + ch * 32);
  };

  // ---- per-lane constants ----------------------------------------------------------------------------------------------------
  const int trow = lane & 31, q8 = trow >> 2, hh = lane >> 5;
  const int ty = 2 * (q8 >> 2) + (__builtin_popcount(q8) & 1), tx = 4 * ((q8 >> 1) & 1) + (trow & 3);
  const int rd_off = hh * 4 * PL + (ty * WROW + tx) * 4;     // floats: quad hh, plane 0, this lane's tile origin
  // U: one buffer resource over the packed weights; wave-uniform byte offset of a 32-channel group's stream + lane * 16
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, ncb * NT * KSTEPS * 4096, 0x00020000);
  auto u_ptr = [&](const WItem& T) { return T.cb * NT * KSTEPS * 4096; };      // first of the item's NT 32-channel streams
  const int lane16 = lane * 16 + I * 1024;       // xi quad I (row I) of the 4 KiB k-step record

  f32x16 acc[4 * NT];      // acc[nt*4 + j]: position (I, j), 32-channel half nt
  f32x4 dq[8];             // dq[r*4+dx] = the 4 channels of this lane's quad at window position (dy = DY[r], dx)
  f32x4 ub[4][NT];         // [k-step of the chunk][nt]: the fragments of k-step k+3 are requested before the MFMAs of k-step k
  // the two window rows row I of B^T d B is made of:  t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3
  constexpr int DYA = I == 0 ? 0 : 1, DYB = I == 3 ? 3 : 2;

  auto read_d = [&](int buf) {
    const float* p = wlds + buf * CHF + rd_off;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int dy = r == 0 ? DYA : DYB;
        dq[r * 4 + dx] = *reinterpret_cast<const f32x4*>(p + ((dy & 1) * 2 + (dx & 1)) * PL + ((dy >> 1) * WROW + (dx >> 1)) * 4);
      }
  };
  auto load_u = [&](int slot, int up, int ks) {
    if constexpr ((ABL & 2) != 0) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ub[slot][nt] = buf_load_f32x4(u_rsrc, lane16, up + (nt * KSTEPS + ks) * 4096);
  };
  // row I of B^T d B for channel j of the quad, then the row transform
  auto transform = [&](auto j_c, float (&v)[4]) {
    constexpr int j = decltype(j_c)::value;
    float t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float ra = dq[c][j], rb = dq[4 + c][j];
      if constexpr (I == 0 || I == 3) t[c] = ra - rb;
      else if constexpr (I == 1) t[c] = ra + rb;
      else t[c] = rb - ra;
    }
    v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
  };
  auto mfmas = [&](auto slot_c, auto first_c, const float (&v)[4]) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr bool first = decltype(first_c)::value != 0;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[nt * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], ub[slot][nt][j], first ? zero : acc[nt * 4 + j], 0, 0, 0);
  };

#ifdef D2FE_DEVTOOLS
  const bool tracing = (ABL & 256) && blockIdx.x == 0 && tid == 0;
  int pair_slot = -1;       // ABL 256: this workgroup sits on (xcc 0, se 0, sh 0, cu 0): record its item timeline beside its CU partner's
  if constexpr ((ABL & 256) != 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0 && (hwid & 0xff00) == 0 && (xcc & 15) == 0) pair_slot = hwid & 1;
  }
  auto mark = [&](int item, int ch, int slot) {
    if constexpr ((ABL & 256) != 0) {
      if (tracing && item < 4) g_wino_trace[item][ch][slot] = clock64();
      if (pair_slot >= 0 && item < 64) {
        if (ch == 0 && slot == 0) g_wino_pair[pair_slot][item][0] = clock64();
        if (ch == 0 && slot == 4) g_wino_pair[pair_slot][item][1] = clock64();
        if (ch == 0 && slot == 5) g_wino_pair[pair_slot][item][2] = clock64();
      }
    }
  };
#else
  auto mark = [&](int, int, int) {};
#endif
  int trace_item = 0;
  int inn = 0, inn_claim = 0;            // the item after `nxt`: its index, and thread 0's claim in flight
  float bias0 = 0.f, bias1 = 0.f;        // the current item's biases (channel lane & 31 of its one or two 32-channel halves), requested at the item's start
  // FUSE: the frame bytes of the next item (see the staging below)
  unsigned char* u8p = reinterpret_cast<unsigned char*>(wlds + 8 * CHF);     // [12][20] bytes behind the patch
  const int fr = tid / 20, fc = tid - fr * 20;
  int fbyte = 0;
  auto load_frame = [&](const WItem& T) {
    const int gy = T.by * 8 - 2 + fr, gx = T.bx * 16 - 2 + fc;
    fbyte = 0;
    if (tid < 240 && gy >= 0 && gy < aH && gx >= 0 && gx < aW) fbyte = a.img[(size_t)T.img * a.img_istride + (size_t)gy * a.img_stride + gx];
  };
  auto store_frame = [&]() { if (tid < 240) u8p[tid] = (unsigned char)fbyte; };
  // ---- FUSE: conv1a for the 10 x 18 patch on the matrix pipe, straight into the chunk layout ------------------------------------
  // D[channel][pixel] = sum_k W[channel][k] * tap[k][pixel]: the weights are the A operand, so that a lane (= pixel) holds 4
  // consecutive channels in registers 4q..4q+3 -- one 16-byte LDS write per channel quad.  K = 10: k = 0 carries the bias against a
  // tap of 1.0 (fmaf(1, b, +0) = b exactly), k = 1..9 the taps in (ky,kx) order -- the same chain as conv1a_kernel, bit for bit.
  // unit u = wave + 4*uu covers m-tile u >> 1 and channel half u & 1 = wave & 1: a wave only ever needs one half of the weights
  float c1a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (FUSE) {
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int k = 2 * st + (lane >> 5);
      c1a[st] = k == 0 ? a.b1a[(wave & 1) * 32 + (lane & 31)] : a.w1a[(k - 1) * 64 + (wave & 1) * 32 + (lane & 31)];
    }
  }
  // The frame bytes under the patch (12 x 20: the 10 x 18 patch plus conv1a's own ring) are staged through LDS: thread t < 240 fetches
  // ONE byte per item, one item ahead (out-of-image bytes are 0 = conv1a's zero padding), and writes it at the start of the running
  // item's epilogue, in front of a barrier that is there anyway.  Every address below is a per-lane constant of the kernel, so what is
  // left per item and unit is 5 byte reads + conversions, the in-image test of the patch pixel and the MFMA chain.
  int tb[3], dsto[3], pyx[3];            // unit uu: byte index of tap (0,0), float index of the patch pixel in the chunk layout, (py, px)
  int koff[5];                           // k-step st: this lane half's tap (ky, kx) as a byte offset
  if constexpr (FUSE) {
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) {
      const int pidx = ((wave >> 1) + 2 * uu) * 32 + (lane & 31);
      const int py = pidx / 18, px = pidx - py * 18;
      tb[uu] = pidx < 180 ? py * 20 + px : 0;
      pyx[uu] = pidx < 180 ? (py << 8) | px : (int)0x8080;         // 0x8080 marks the 12 lanes of the last m-tile beyond the patch: computed, not stored
      dsto[uu] = ((py & 1) * 2 + (px & 1)) * PL + ((py >> 1) * WROW + (px >> 1)) * 4;
    }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int t = 2 * st + (lane >> 5) - 1;                      // tap index; -1 is the bias slot (k = 0)
      koff[st] = t < 0 ? 0 : (t / 3) * 20 + t % 3;
    }
  }
  // The staging is split: stage_taps (byte reads + conversions), stage_step (one k-step of the three conv1a MFMA chains) and stage_store (ReLU +
  // the patch writes).  A wave issues in order, so its MFMAs only run under its own VALU / LDS / store work when they are INTERLEAVED with it:
  // the epilogue issues the next item's taps and the five chain steps between the four quarters of its finish phase (exchange reads, A^T s, bias,
  // ReLU, pool, stores), which touch neither the frame bytes nor the chain registers; stage_store follows the barrier that ends the epilogue.
  auto stage_taps = [&](const WItem& T, float (&tap)[3][5]) {
    const float scale = (float)(1.0 / 255.0);
    const int gy0 = T.by * 8 - 1, gx0 = T.bx * 16 - 1;
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) {
      // a patch pixel outside the image is conv1b's zero padding: all its taps AND the bias slot are 0, so the chain gives +0
      const bool pvalid = (unsigned)(gy0 + (pyx[uu] >> 8)) < (unsigned)aH && (unsigned)(gx0 + (pyx[uu] & 255)) < (unsigned)aW;
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        float v = (float)u8p[tb[uu] + koff[st]] * scale;
        if (st == 0) v = hh ? v : 1.0f;                           // k = 0 (lanes 0-31 of the first step) carries the bias against 1.0
        tap[uu][st] = pvalid ? v : 0.f;
      }
    }
  };
  // 12 units = 6 m-tiles of 32 patch pixels x 2 halves of the channels; unit u = wave + 4*uu covers m-tile u >> 1, half wave & 1
  auto stage_step = [&](auto st_c, const float (&tap)[3][5], f32x16 (&d)[3]) {
    constexpr int st = decltype(st_c)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) d[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1a[st], tap[uu][st], st == 0 ? zero : d[uu], 0, 0, 0);
  };
  auto stage_store = [&](const f32x16 (&d)[3]) {
    const int nt = wave & 1;
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) {
      // rows (channels) of this lane: 8*q + 4*hh + (0..3) of half nt -> channel quad Q = nt*8 + 2*q + hh; column = patch pixel
      if (pyx[uu] != (int)0x8080) {
        float* dst = wlds + dsto[uu];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int Q = nt * 8 + 2 * q + hh;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)      // ReLU as one v_max (the chain never yields -0 or a NaN; the builtins add a canonicalising second one)
            asm("v_max_f32 %0, 0, %1" : "=v"(o[e]) : "v"(d[uu][4 * q + e]));
          *reinterpret_cast<f32x4*>(dst + (Q >> 1) * CHF + (Q & 1) * 4 * PL) = o;
        }
      }
    }
  };

  // ---- epilogue: s = M A for this wave's row, swap rows between the four waves, A^T s, bias, ReLU, pool, store ----------------
  float* xch = FUSE ? wlds : wlds + WR * WCHUNK;      // FUSE: the patch is dead once the K loop is through, the exchange area reuses it
  auto epilogue = [&](const WItem& T, bool stage_next, const WItem& TN) {
    float* out = a.out + (size_t)T.img * a.out_img_stride + a.out_coff;
    const int cs = a.out_cstride, cs4 = cs * 4;
    const int n32 = T.cb * NT;
    const int co0 = n32 * 32 + (lane & 31);
    // fast path (item inside the image, all of its channels real): buffer stores, wave-uniform offsets on the SALU
    const bool full = T.by * 8 + 8 <= aH && T.bx * 16 + 16 <= aW && (n32 + NT) * 32 <= a.cout_real && !(ABL & 4);
    const int Wo = POOL ? (aW >> 1) : aW, Ho = POOL ? (aH >> 1) : aH;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, Ho * Wo * cs4, 0x00020000);
    const int rowpair = (POOL ? Wo : 2 * Wo) * cs4;               // bytes between tile rows ty and ty + 1
    const int vo0 = (lane & 31) * 4 + hh * rowpair, vo1 = (lane & 31) * 4 + (1 - hh) * rowpair;
    const int obase = ((POOL ? T.by * 4 * Wo + T.bx * 8 : T.by * 8 * Wo + T.bx * 16)) * cs4 + n32 * 128;
    if constexpr (FUSE) { store_frame(); __syncthreads(); }   // the next item's frame bytes; every wave is through with the patch before the exchange overwrites it
    mark(trace_item, 8, 0);
    // Everything below works on PAIRS of tiles (accumulator registers r, r + 1: neighbours in x) as the halves of v_pk_add_f32 -- the
    // epilogue runs while this wave has no MFMAs in flight, so every VALU instruction saved is time.  The compiler scalarises
    // <2 x float> subtractions, hence the instruction is asked for by name.  Same additions in the same order as one tile at a time.
    auto padd = [](f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto psub = [](f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y)); return d; };
    // kept pairs: sk[p * 2 + nt] = (s_b0(r), s_b0(r+1), s_b1(r), s_b1(r+1)) of row I for the tile pair r = 4I + 2p and channel half nt
    f32x4 sk[2 * NT];
    f32x4* xw = reinterpret_cast<f32x4*>(xch) + (I * 6 * NT) * 64 + lane;
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      const int r = 2 * rp, o = rp >> 1, p = rp & 1;        // o: the wave that finishes these two tiles
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x2 m0 = {acc[nt * 4 + 0][r], acc[nt * 4 + 0][r + 1]}, m1 = {acc[nt * 4 + 1][r], acc[nt * 4 + 1][r + 1]};
        const f32x2 m2 = {acc[nt * 4 + 2][r], acc[nt * 4 + 2][r + 1]}, m3 = {acc[nt * 4 + 3][r], acc[nt * 4 + 3][r + 1]};
        const f32x2 b0 = padd(padd(m0, m1), m2), b1 = psub(psub(m1, m2), m3);
        const f32x4 q = {b0[0], b0[1], b1[0], b1[1]};
        if (o == I) sk[p * NT + nt] = q; else xw[((o < I ? o : o - 1) * 2 * NT + p * NT + nt) * 64] = q;
      }
    }
    mark(trace_item, 8, 1);
    if (dyn && tid == 0) *claim_slot = 2 * tstride + inn_claim;          // the index claimed at the start of this item (one barrier serves both)
    __syncthreads();
    if (dyn) inn = *claim_slot;
    // both biases count as arrived from here on: first used between the stores of the finish phase, the second one would otherwise cost a vmcnt(0)
    // there -- loads and stores share one in-order counter, so that wait also sits out the stores just issued
    asm volatile("" : "+v"(bias0), "+v"(bias1));
    mark(trace_item, 8, 2);
    // bias, ReLU, pool and the stores of one tile (register r) and channel half nt; y[pp][b] = output pixel (2 ty + pp, 2 tx + b)
    auto emit_tile = [&](auto r_c, int nt, float (&y)[2][2]) {
      constexpr int r = decltype(r_c)::value;
      constexpr int par = ((r >> 2) ^ (r >> 3)) & 1;
      const int txr = 4 * ((r >> 2) & 1) + (r & 3);
      if constexpr (RELU) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int b = 0; b < 2; ++b) y[pp][b] = y[pp][b] > 0.f ? y[pp][b] : 0.f;
      }
      if (full) {
        // interior item: uniform byte offset (SALU) + one of two per-lane offsets (channel, and the tile row this lane half holds)
        const int vo = par ? vo1 : vo0;
        if constexpr (POOL) {
          const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
          buf_store_f32(v, orsrc, vo, obase + ((2 * (r >> 3)) * (aW >> 1) + txr) * cs4 + nt * 128);
        } else {
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              buf_store_f32(y[pp][b], orsrc, vo, obase + ((4 * (r >> 3) + pp) * aW + 2 * txr + b) * cs4 + nt * 128);
        }
      } else {
        const int co = co0 + nt * 32;
        const bool cok = co < a.cout_real && !(ABL & 4);
        const int tyr = 2 * (r >> 3) + (par ^ hh);
        const int oy = T.by * 8 + 2 * tyr, ox = T.bx * 16 + 2 * txr;
        if constexpr (POOL) {
          const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
          if (cok && oy + 1 < aH && ox + 1 < aW) out[((size_t)(oy >> 1) * (aW >> 1) + (ox >> 1)) * cs + co] = v;
        } else {
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (cok && oy + pp < aH && ox + b < aW) out[((size_t)(oy + pp) * aW + ox + b) * cs + co] = y[pp][b];
        }
      }
    };
    auto finish_q = [&](auto p_c, auto nt_c) {
      constexpr int p = decltype(p_c)::value;
      constexpr int nt = decltype(nt_c)::value;
      constexpr int r = 4 * I + 2 * p;
      if constexpr (nt < NT) {
        f32x2 lo[4], hi[4];      // rows 0..3 of s: lo = (b0(r), b0(r+1)), hi = (b1(r), b1(r+1))
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          f32x4 sv;
          if (w == I) sv = sk[p * NT + nt];
          else sv = (reinterpret_cast<const f32x4*>(xch) + (w * 6 * NT + (I < w ? I : I - 1) * 2 * NT + p * NT + nt) * 64)[lane];
          lo[w] = f32x2{sv[0], sv[1]}; hi[w] = f32x2{sv[2], sv[3]};
        }
        const float bias = nt ? bias1 : bias0;
        const f32x2 bb = {bias, bias};
        // y[pp][b] of tile r in element 0, of tile r + 1 in element 1
        const f32x2 y00 = padd(padd(padd(lo[0], lo[1]), lo[2]), bb), y01 = padd(padd(padd(hi[0], hi[1]), hi[2]), bb);
        const f32x2 y10 = padd(psub(psub(lo[1], lo[2]), lo[3]), bb), y11 = padd(psub(psub(hi[1], hi[2]), hi[3]), bb);
        float ya[2][2] = {{y00[0], y01[0]}, {y10[0], y11[0]}}, yb[2][2] = {{y00[1], y01[1]}, {y10[1], y11[1]}};
        emit_tile(IC<r>{}, nt, ya);
        emit_tile(IC<r + 1>{}, nt, yb);
      }
    };
    if constexpr (FUSE) {
      // the next item's conv1a chains, one k-step between two quarters of the finish phase (see stage_step); the fences pin the interleaving
      float tap[3][5];
      f32x16 dstage[3];          // local to the epilogue: no value is carried around the item loop
      if (stage_next) { stage_taps(TN, tap); stage_step(IC<0>{}, tap, dstage); }
      __builtin_amdgcn_sched_barrier(0);
      finish_q(IC<0>{}, IC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      if (stage_next) stage_step(IC<1>{}, tap, dstage);
      __builtin_amdgcn_sched_barrier(0);
      finish_q(IC<0>{}, IC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      if (stage_next) stage_step(IC<2>{}, tap, dstage);
      __builtin_amdgcn_sched_barrier(0);
      finish_q(IC<1>{}, IC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      if (stage_next) stage_step(IC<3>{}, tap, dstage);
      __builtin_amdgcn_sched_barrier(0);
      finish_q(IC<1>{}, IC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      if (stage_next) stage_step(IC<4>{}, tap, dstage);
      mark(trace_item, 8, 3);
      // the staging overwrites the patch = the exchange area: every wave must be through with it
      __syncthreads();
      if (stage_next) stage_store(dstage);
      return;
    } else {
      finish_q(IC<0>{}, IC<0>{}); finish_q(IC<0>{}, IC<1>{}); finish_q(IC<1>{}, IC<0>{}); finish_q(IC<1>{}, IC<1>{});
    }
    mark(trace_item, 8, 3);
    // FUSE: the next item's staging overwrites the patch = the exchange area, so every wave must be through with it.  Otherwise the
    // next writer of the exchange area is the next item's epilogue, eight chunk barriers away: no barrier needed here.
  };

  // ---- prologue ---------------------------------------------------------------------------------------------------------------
  WItem cur = w_decode(icur, nbx, nby, ncb);
  WItem nxt = inxt < total ? w_decode(inxt, nbx, nby, ncb) : cur;
  DmaItem dcur{}, dnxt{};
  if constexpr (!FUSE) {
    dcur = dma_prepare(cur); dnxt = dma_prepare(nxt);
    // chunk c of the walk belongs to item c / NCH: the DMA cursor is at most WR chunks (< NCH) ahead, i.e. in `cur` or `nxt`
#pragma unroll
    for (int c = 0; c < WR; ++c) dma_issue(dcur, c, c);      // WR < NCH: all in the first item
  }
  if constexpr (FUSE) {
    float tap0[3][5]; f32x16 d0[3];
    load_frame(cur); store_frame(); __syncthreads();
    stage_taps(cur, tap0);
    stage_step(IC<0>{}, tap0, d0); stage_step(IC<1>{}, tap0, d0); stage_step(IC<2>{}, tap0, d0); stage_step(IC<3>{}, tap0, d0); stage_step(IC<4>{}, tap0, d0);
    stage_store(d0);
  }
  int ucur = u_ptr(cur), unxt = u_ptr(nxt);
  load_u(0, ucur, 0);
  load_u(1, ucur, 1);
  load_u(2, ucur, 2);
  if constexpr (!FUSE) {
    // chunk 0 landed (younger: 2 x 2 copies + 3 x NT U loads)
    if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __syncthreads();
    read_d(0);
  }


  // ---- walk: items x chunks; g counts chunks across items (ring position) ---------------------------------------------------
  int g = 0;
  auto chunk = [&](auto first_c, int item, int ch) {
    const bool last_ch = ch == NCH - 1;
    float v[4];
    mark(item, ch, 0);
    // U fragments run three k-steps ahead through four register slots: slot (k+3)&3 held k-step k-1, whose MFMAs are issued
    load_u(3, ucur, ch * 4 + 3);
    transform(IC<0>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(IC<0>{}, first_c, v);
    __builtin_amdgcn_sched_barrier(0);
    // k-steps 0..2 of the next chunk (the next item's first chunk at an item boundary; wraps harmlessly at the very end)
    if (last_ch) load_u(0, unxt, 0); else load_u(0, ucur, ch * 4 + 4);
    transform(IC<1>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(IC<1>{}, IC<0>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    if (last_ch) load_u(1, unxt, 1); else load_u(1, ucur, ch * 4 + 5);
    transform(IC<2>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(IC<2>{}, IC<0>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    if (last_ch) load_u(2, unxt, 2); else load_u(2, ucur, ch * 4 + 6);
    // the last k-step: its transform frees dq, then chunk g+1 is made visible and read while its MFMAs run
    transform(IC<3>{}, v);
    __builtin_amdgcn_sched_barrier(0);
    mark(item, ch, 1);
    if constexpr (FUSE) {
      // the whole patch is in LDS: read the next chunk (after the last one: a harmless re-read of chunk 0)
      read_d((ch + 1) & (NCH - 1));
    } else {
      const bool has_next = inxt < total;
      if (ch + 1 < NCH || has_next) {
        // chunk g+1 (copied WR-1 iterations ago) must have landed: the loads younger than it are this iteration's 4 x NT U loads
        // and the 2 copies of chunk g+2 -- loads complete in order, so "at most 4 NT + 2 outstanding" implies it is complete (a claim
        // in flight in wave 0 only makes the wait stricter)
        if (ch + 2 < NCH || has_next) { if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      mark(item, ch, 2);
      if constexpr (!(ABL & 8)) __syncthreads();   // every wave's part of chunk g+1 is in LDS; every wave has read chunk g out of its buffer
      mark(item, ch, 3);
      {
        const int c3 = ch + WR;      // chunk g+WR of the walk, relative to the current item
        if (c3 < NCH) dma_issue(dcur, c3, g % WR); else if (has_next) dma_issue(dnxt, c3 - NCH, g % WR);
      }
      if constexpr (!(ABL & 16)) read_d((g + 1) % WR);        // unconditional (a stale buffer after the very last chunk): no phi copies
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(IC<3>{}, IC<0>{}, v);
    ++g;
  };
#pragma unroll 1
  for (int item = 0; icur < total; ++item) {
    inn = inxt + tstride;
    // in flight during the K loop, published in the epilogue.  The raw return value is not touched before that (an add here would wait for the
    // round trip on the spot), and this file is compiled with the AMDGPU atomic optimizer off: its wave-wide reduction reads the result back with
    // v_readfirstlane right behind the atomic -- a 1-2 us stall of wave 0 at the start of every item
    if (dyn && tid == 0) inn_claim = atomicAdd(a.work_ctr, 1);
    {      // the biases of this item: older than every load of the K loop, so they have arrived long before the epilogue asks (requested there, they were
           // the YOUNGEST loads in flight and their wait also sat out the next item's prefetch)
      const int co0 = cur.cb * NT * 32 + (lane & 31);
      bias0 = a.bias[co0]; bias1 = NT == 2 ? a.bias[co0 + 32] : 0.f;
    }
    if constexpr (FUSE) {
      __syncthreads();                                           // the patch (staged by the prologue / the previous epilogue) is complete
      read_d(0);
      if (inxt < total && !D2FE_ABL(a, 64)) load_frame(nxt);     // the next item's frame bytes, in flight during this item's K loop
    }
    chunk(IC<1>{}, item, 0);       // the first k-step multiplies into C = 0: no accumulator clearing
#pragma unroll 1
    for (int ch = 1; ch < NCH; ++ch) chunk(IC<0>{}, item, ch);
    mark(item, 0, 4);
    trace_item = item;
    // D2FE_ABLATE=64: timing experiment, the staging runs for the first item only
    if constexpr (!(ABL & 32)) epilogue(cur, FUSE && inxt < total && !D2FE_ABL(a, 64), nxt);
    mark(item, 0, 5);
    cur = nxt; dcur = dnxt; ucur = unxt;
    icur = inxt; inxt = inn;
    if (inxt < total) {
      nxt = w_decode(inxt, nbx, nby, ncb);
      if constexpr (!FUSE) dnxt = dma_prepare(nxt);
      unxt = u_ptr(nxt);
    }
  }
}

template <int CIN, bool POOL, bool RELU, int ABL = 0, int TAG = 0, bool FUSE = false, int NT = 2>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void conv_wino_kernel(ConvArgs a, int nbx, int nby, int ncb, int total) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  if ((int)blockIdx.x >= total) return;
#ifdef D2FE_DEVTOOLS
  if constexpr ((ABL & 256) != 0) {
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_wino_hw[blockIdx.x][0] = hwid; g_wino_hw[blockIdx.x][1] = xcc;
    }
  }
#endif
  // the four rows of the transform domain run different (compile-time) row arithmetic; the branch is wave-uniform
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: wino_body<CIN, POOL, RELU, ABL, 0, FUSE, NT>(a, nbx, nby, ncb, total, wlds); break;
    case 1: wino_body<CIN, POOL, RELU, ABL, 1, FUSE, NT>(a, nbx, nby, ncb, total, wlds); break;
    case 2: wino_body<CIN, POOL, RELU, ABL, 2, FUSE, NT>(a, nbx, nby, ncb, total, wlds); break;
    default: wino_body<CIN, POOL, RELU, ABL, 3, FUSE, NT>(a, nbx, nby, ncb, total, wlds); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
hipError_t launch_conv_wino(int cin, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  if (cout_pad % 64 || (a.in_cstride & 3) || (a.in_coff & 3)) return hipErrorInvalidValue;
  // buffer addressing: 32-bit byte offsets inside one image's tensor, 0x80000000 must stay out of range
  if ((long)a.H * a.W * a.in_cstride * 4 >= (1l << 31) || (long)a.H * a.W * a.out_cstride * 4 >= (1l << 31)) return hipErrorInvalidValue;
  // 64-channel items (NT = 2, two workgroups per CU) are 6-8 % faster per unit of work than 32-channel items (NT = 1, three per CU;
  // profiles/r03_wino_nt_ab.txt), so every launch that keeps the chip busy for several rounds of workgroups uses them.  A SMALL launch -- the
  // 60 x 80 and 120 x 160 layers of a one- or two-image call, how the reference calls infer (loop_cam.cpp:609-616) -- is a handful of items
  // per CU at most: its duration is (rounds of resident workgroups) x (one item's latency), and halving the item while raising the resident
  // workgroups from 2 to 3 per CU wins there (conv4a at two images: 160 items -> 320 half-items in ONE round of 768 slots).
  // D2FE_WINO_NT=1 / 2 forces one form (A/B measurements); bit-identical either way.
  static const int nt_env = d2fe_dev_env("D2FE_WINO_NT", 0);
  int NTsel = 2;
  {
    const int ncu0 = a.ncu > 0 ? a.ncu : 256;
    const long total2 = (long)((a.W + 15) / 16) * ((a.H + 7) / 8) * (cout_pad / 64) * a.n_img;
    if (total2 <= 4l * 2 * ncu0) {
      const long rounds2 = (total2 + 2 * ncu0 - 1) / (2 * ncu0), rounds1 = (2 * total2 + 3 * ncu0 - 1) / (3 * ncu0);
      if (rounds1 * 0.5 * 1.07 < (double)rounds2) NTsel = 1;
    }
    if (nt_env == 1) NTsel = 1;
    if (nt_env == 2 || a.ablate) NTsel = 2;
  }
  const int nbx = (a.W + 15) / 16, nby = (a.H + 7) / 8, ncb = cout_pad / (32 * NTsel);
  const int total = nbx * nby * ncb * a.n_img;
  const int ncu = a.ncu > 0 ? a.ncu : 256;     // ConvArgs::ncu: the handle's device (no process-wide cache: one process may drive several GPUs)
  const int wg_per_cu = NTsel == 1 ? 3 : 2;
  const int grid = total < wg_per_cu * ncu ? total : wg_per_cu * ncu;
  const size_t lds = (size_t)(WR * WCHUNK + (NTsel == 1 ? WXCH / 2 : WXCH)) * sizeof(float) + 16;      // ring + exchange area + the claim slot
#define D2FE_WINO_K(K)                                                                                       \
  do {                                                                                                      \
    auto k = K;                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                          \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a, nbx, nby, ncb, total);                          \
  } while (0)
  if (NTsel == 1) {
    if (cin == 64 && pool && relu) { D2FE_WINO_K((conv_wino_kernel<64, true, true, 0, 0, false, 1>)); return hipGetLastError(); }
    if (cin == 64 && !pool && relu) { D2FE_WINO_K((conv_wino_kernel<64, false, true, 0, 0, false, 1>)); return hipGetLastError(); }
    if (cin == 128 && pool && relu) { D2FE_WINO_K((conv_wino_kernel<128, true, true, 0, 0, false, 1>)); return hipGetLastError(); }
    if (cin == 128 && !pool && relu) { D2FE_WINO_K((conv_wino_kernel<128, false, true, 0, 0, false, 1>)); return hipGetLastError(); }
    return hipErrorInvalidValue;
  }
#ifdef D2FE_DEVTOOLS
  if (a.ablate && cin == 64 && !pool && relu) {     // timing experiments through d2fe_debug_conv3x3_wino (D2FE_ABLATE)
    switch (a.ablate) {
      case 1: D2FE_WINO_K((conv_wino_kernel<64, false, true, 1>)); return hipGetLastError();
      case 2: D2FE_WINO_K((conv_wino_kernel<64, false, true, 2>)); return hipGetLastError();
      case 8: D2FE_WINO_K((conv_wino_kernel<64, false, true, 8>)); return hipGetLastError();
      case 16: D2FE_WINO_K((conv_wino_kernel<64, false, true, 16>)); return hipGetLastError();
      case 19: D2FE_WINO_K((conv_wino_kernel<64, false, true, 19>)); return hipGetLastError();
      case 256: {
        D2FE_WINO_K((conv_wino_kernel<64, false, true, 256>));
        static int dumped = 0;
        long long tr[4][16][6];
        static unsigned hw[1024][2];
        if (!dumped && hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wino_hw), sizeof(hw)) == hipSuccess) {
          for (int b = 0; b < grid && b < 1024; b += (b < 8 || (b >= 256 && b < 264)) ? 1 : 37)
            fprintf(stderr, "wg %4d: hw_id %08x wave_id %u simd %u cu %u sh %u se %u | xcc_id %08x\n", b, hw[b][0], hw[b][0] & 15, (hw[b][0] >> 4) & 3,
                    (hw[b][0] >> 8) & 15, (hw[b][0] >> 12) & 1, (hw[b][0] >> 13) & 7, hw[b][1]);
          // how many workgroups share (xcc, se, sh, cu), and do partners differ in wave_id parity?
          int pairs = 0, parity_ok = 0, half_ok = 0;
          for (int b = 0; b < grid && b < 1024; ++b)
            for (int c = b + 1; c < grid && c < 1024; ++c)
              if ((hw[b][0] & 0xff00) == (hw[c][0] & 0xff00) && (hw[b][1] & 15) == (hw[c][1] & 15)) {
                ++pairs; parity_ok += ((hw[b][0] ^ hw[c][0]) & 1); half_ok += (b < grid / 2) != (c < grid / 2);
              }
          fprintf(stderr, "co-resident workgroup pairs: %d; wave_id parity differs in %d; in different halves of the grid in %d\n", pairs, parity_ok, half_ok);
        }
        static long long pr[2][64][3];
        if (!dumped && hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_wino_pair), sizeof(pr)) == hipSuccess) {
          const long long base = pr[0][0][0] < pr[1][0][0] ? pr[0][0][0] : pr[1][0][0];
          for (int it = 20; it < 28; ++it)
            fprintf(stderr, "CU (0,0,0,0) item %2d: wg A  K %8lld  epilogue %8lld..%8lld | wg B  K %8lld  epilogue %8lld..%8lld | B.epi - A.epi = %6lld\n", it,
                    pr[0][it][0] - base, pr[0][it][1] - base, pr[0][it][2] - base, pr[1][it][0] - base, pr[1][it][1] - base, pr[1][it][2] - base,
                    pr[1][it][1] - pr[0][it][1]);
        }
        if (!dumped++ && hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_wino_trace), sizeof(tr)) == hipSuccess) {
          for (int it = 0; it < 3; ++it) {
            for (int c = 0; c < 8; ++c)
              fprintf(stderr, "item %d chunk %d: 3 k-steps+transform %5lld | vmcnt wait %5lld | barrier %5lld | to next chunk start %5lld\n", it, c,
                      tr[it][c][1] - tr[it][c][0], tr[it][c][2] - tr[it][c][1], tr[it][c][3] - tr[it][c][2],
                      (c < 7 ? tr[it][c + 1][0] : tr[it][0][4]) - tr[it][c][3]);
            fprintf(stderr, "item %d epilogue %5lld (s-transform + exchange writes %5lld | barrier %5lld | finish + stores %5lld), item period %5lld\n", it,
                    tr[it][0][5] - tr[it][0][4], tr[it][8][1] - tr[it][8][0], tr[it][8][2] - tr[it][8][1], tr[it][8][3] - tr[it][8][2],
                    tr[it + 1][0][0] - tr[it][0][0]);
          }
        }
        return hipGetLastError();
      }
      default: break;
    }
  }
#endif
  if (cin == 64 && pool && relu && a.tag == 1) { D2FE_WINO_K((conv_wino_kernel<64, true, true, 0, 1>)); return hipGetLastError(); }   // conv1b
  if (cin == 64 && pool && relu) { D2FE_WINO_K((conv_wino_kernel<64, true, true>)); return hipGetLastError(); }
  if (cin == 64 && !pool && relu) { D2FE_WINO_K((conv_wino_kernel<64, false, true>)); return hipGetLastError(); }
  if (cin == 128 && pool && relu) { D2FE_WINO_K((conv_wino_kernel<128, true, true>)); return hipGetLastError(); }
  if (cin == 128 && !pool && relu) { D2FE_WINO_K((conv_wino_kernel<128, false, true>)); return hipGetLastError(); }
#undef D2FE_WINO_K
  return hipErrorInvalidValue;
}

// conv1a (from the u8 frame) fused into the Winograd conv1b: a.img / a.w1a / a.b1a instead of a.in
hipError_t launch_conv_wino_fused1b(int cout_pad, const ConvArgs& a, hipStream_t s) {
  if (cout_pad != 64 || !a.img || !a.w1a || !a.b1a) return hipErrorInvalidValue;
  if ((long)a.H * a.W * a.out_cstride >= (1l << 29)) return hipErrorInvalidValue;
  const int nbx = (a.W + 15) / 16, nby = (a.H + 7) / 8, ncb = 1;
  const int total = nbx * nby * a.n_img;
  const int ncu = a.ncu > 0 ? a.ncu : 256;     // ConvArgs::ncu: the handle's device (no process-wide cache: one process may drive several GPUs)
  const int grid = total < 2 * ncu ? total : 2 * ncu;
  constexpr size_t lds = (size_t)8 * 2 * 4 * 272 * sizeof(float) + 256 + 16;      // the 64-channel patch (the exchange area, 48 KiB, reuses it) + the 12 x 20 frame bytes
  static_assert(lds >= (size_t)WXCH * sizeof(float), "exchange area must fit into the patch buffer");
  auto k = conv_wino_kernel<64, true, true, 0, 1, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a, nbx, nby, ncb, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side weight transform + packing:  [32-channel group][k-step][xi quad][lane] float4,
//   float4[e] = U[xi = 4q+e][ci = 8*(ks/4) + 4*(lane>>5) + ks%4][co = group*32 + (lane&31)]
// ---------------------------------------------------------------------------------------------------------------------
size_t packed_weight_floats_wino(int cout_pad, int cin) { return (size_t)16 * cout_pad * cin; }

void pack_weights_wino(const float* w, int cout, int cin, int cout_pad, float* dst) {
  static const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int ksteps = cin / 2;
  for (int grp = 0; grp < cout_pad / 32; ++grp)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int xi = 4 * q + e, i = xi >> 2, j = xi & 3;
            const int co = grp * 32 + (lane & 31), ci = 8 * (ks / 4) + 4 * (lane >> 5) + (ks % 4);
            double sum = 0.0;
            if (co < cout) {
              const float* g = w + ((size_t)co * cin + ci) * 9;
              for (int p = 0; p < 3; ++p)
                for (int r = 0; r < 3; ++r) sum += Gm[i][p] * Gm[j][r] * (double)g[p * 3 + r];
            }
            dst[((((size_t)grp * ksteps + ks) * 4 + q) * 64 + lane) * 4 + e] = (float)sum;
          }
}

}  // namespace d2fe

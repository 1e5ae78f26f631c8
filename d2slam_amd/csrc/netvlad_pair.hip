// netvlad_pair.hip -- the stride-1 MobileNetV2 inverted-residual blocks (pw expand -> dw 3x3 -> pw project (+ residual)) of the NetVLAD trunk and its
// first block (conv 3x3 from the u8 frame -> dw 3x3 -> pw project), one kernel per block, organised around the LDS traffic of the depthwise stage (gfx950).  Reference boundary: MobileNetVLADONNX::inference,
// d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74 (one ONNX Runtime session run); the layer list is whatever d2fe_load_netvlad() got.
//
// Same data flow as nv_xblock_kernel (netvlad_fused.hip): block input in registers, the expanded tensor 16 hidden channels at a time through LDS,
// project accumulators in registers.  What the phase stamps of that kernel showed (tools/nv_stamps.py, profiles/r03_netvlad_stamps.txt): a chunk costs
// 3.1 us, 2.0 of them in depthwise + project, where every wave issues 128 ds_read_b32 (72 for the 3x3 windows, 40 depthwise weights, 16 project
// B fragments) with a third of the LDS cycles lost to bank conflicts -- the LDS pipe of the CU, shared by ~10 waves, is the limit, not the matrix
// or vector pipes.  Here:
//   * a lane owns two horizontally adjacent output pixels (2 px, 2 px + 1) of one hidden channel: the 4 input columns both windows cover are two
//     aligned ds_read_b64 per window row (24 per chunk and wave instead of 72 ds_read_b32), and the pair is the two halves of v_pk_fma_f32;
//     even pixels feed the wave's first project m-tile, odd pixels the second
//   * E rows (one per hidden channel of the chunk, 256-float pitch) start at bank 4 (c & 7) + 32 (c >> 3): the expand stage's ds_write_b128 (8-lane
//     groups = 8 channels) lands on 8 distinct 16-byte slots, and the two channels c, c + 8 that the lane groups of one 32-lane half read
//     are exactly half a bank row (32 of 64 banks) apart -- hidden channel of k-step ks for lane group lq: ks + 4 (lq >> 1) + 8 (lq & 1)
//   * depthwise weights [lane group][k-step][12] and project B fragments [k-step][lane][n-tiles] are read as ds_read_b128 / b64
//   * the expand GEMM walks exactly Cin / 4 k-steps (lane group lq holds input channels lq Cin/4 ..): no zero rows for Cin = 8, 24, 56; its bias
//     is the accumulator's initial value (bias x in-image mask in the C layout) instead of one more k-step
//   * the residual is read first thing in the kernel, not in the epilogue
#include <algorithm>

#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ float nvp_lo(int act) { return act >= 1 ? 0.f : -__builtin_inff(); }
__device__ __forceinline__ float nvp_hi(int act) { return act == 2 ? 6.f : __builtin_inff(); }
__device__ __forceinline__ float nvp_clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
}  // namespace

constexpr int NVP_EROW = 256, NVP_EBUF = 16 * NVP_EROW, NVP_PATCH = 192;      // patch pixels = 12 m-tiles, 3 per wave
constexpr int NVP_U8_PITCH = 40, NVP_U8_ROWS = 24, NVP_U8_DUMMY = 2 * NVP_U8_PITCH + 8;      // first block: u8 copy of the patch's receptive field, (10 - 1) * 2 + 3 = 21 rows x 37 bytes at stride 2
__host__ __device__ constexpr int nvp_we_rec(int nk) { return (nk * 64 + 16 + 255) / 256 * 256; }      // [nk/4][lane][4] (+ [lane][2]) + bias[16]
// project B fragments of one k-step: the n-tiles in groups that ONE LDS read fetches per lane -- nt / 4 groups of four ([lane][4], ds_read_b128), then the remaining
// 1 / 2 / 3 n-tiles as [lane][1] (b32), [lane][2] (b64) or [lane][4] with the last float unused (b128).  nt = 1, 2, 4, 8 are the layouts of rounds 3-5
__host__ __device__ constexpr int nvp_rem_width(int nt) { return nt % 4 == 3 ? 4 : nt % 4; }
__host__ __device__ constexpr int nvp_ks_floats(int nt) { return 64 * (4 * (nt / 4) + nvp_rem_width(nt)); }
__host__ __device__ constexpr int nvp_wd_rec(int nt) { return (256 + 4 * nvp_ks_floats(nt) + 255) / 256 * 256; }      // [lq][ks][12] + pad, [ks][groups as above]
__host__ __device__ constexpr int nvp_row(int c) { return c * NVP_EROW + 4 * (c & 7) + 32 * (c >> 3); }

// waves per SIMD the register allocation must leave room for (unified VGPR + AGPR file, 512 per lane): what the launches of this network need to be
// resident in ONE round -- 640 workgroups of <2, 8> on 256 CUs need 3 per CU, 1280 of <1, 4> need 5; left alone the compiler settles for 2 and 4
__host__ __device__ constexpr int nvp_min_waves(int nt, int nk, bool merge = false) { return (merge && nt * 8 + nk * 3 > 72) ? 2 : nt == 1 ? (nk <= 4 ? 5 : nk <= 8 ? 4 : 3) : nt == 2 ? (nk == 6 ? 4 : nk <= 8 ? 3 : 2) : (nt <= 5 && nk <= 12) ? 3 : 2; }
// Cin >= 72 (18 / 30 k-steps: 54 / 90 input registers per lane): no second register set for a producer's partial slabs -- the input must be ONE slab (run_netvlad sums
// first; these layers' producers split into >= 3 groups and are summed anyway).  That, and the residual read in the epilogue from Cin 48 on, is what lets the 48-wide
// blocks run three workgroups per CU instead of two (166 registers, no spills; the 72-wide ones spill at 168 and measured slower: two; tools/kernel_resources.py)
__host__ __device__ constexpr bool nvp_multi_in(int nk) { return nk < 18; }
// project stage of one k-step: the pair's two depthwise outputs (even pixel -> m-tile 0, odd pixel -> m-tile 1) against the NT n-tiles of B fragments at `wks`
template <int NT>
__device__ __forceinline__ void nvp_project(const float* wks, int lane, float d0, float d1, f32x4 (&acc)[2][NT]) {
  constexpr int NG4 = NT / 4, REM = NT % 4;
#pragma unroll
  for (int hf = 0; hf < NG4; ++hf) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(wks + (hf * 64 + lane) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0][hf * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0, v[e], acc[0][hf * 4 + e], 0, 0, 0);
      acc[1][hf * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, v[e], acc[1][hf * 4 + e], 0, 0, 0);
    }
  }
  if constexpr (REM != 0) {
    float wv[REM];
    const float* wr = wks + NG4 * 256;
    if constexpr (REM == 3) { const f32x4 v = *reinterpret_cast<const f32x4*>(wr + lane * 4); wv[0] = v[0]; wv[1] = v[1]; wv[2] = v[2]; }
    else if constexpr (REM == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(wr + lane * 2); wv[0] = v[0]; wv[1] = v[1]; }
    else wv[0] = wr[lane];
#pragma unroll
    for (int e = 0; e < REM; ++e) {
      acc[0][NG4 * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0, wv[e], acc[0][NG4 * 4 + e], 0, 0, 0);
      acc[1][NG4 * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, wv[e], acc[1][NG4 * 4 + e], 0, 0, 0);
    }
  }
}

// MERGE: a workgroup walks a.gmerge consecutive hidden-channel groups (see NvBlockArgs::gmerge): at every group boundary the project accumulators are closed into a
// running total `tot` and cleared.  `tot` starts as the residual (group 0's workgroup; zero elsewhere), so the first boundary makes (acc + bias) + residual -- the
// value group 0 stores in the unmerged launch -- and every further boundary adds one more group's partial sum, in group order.
// Not for Cin = 120 (240-250 registers without a running total); a variant that kept the total in the workgroup's own output slab (read back, add, store at every
// boundary) measured 101 us against 80 for the 15 x 20 layers at 32 images: those launches are bound by their chunks, not by the workgroups' prologues.
template <int NT, int NK, int NBUF, int MERGE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(nvp_min_waves(NT, NK, MERGE == 1)))) void nv_pblock_kernel(NvBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WE_N = nvp_we_rec(NK), WD_N = nvp_wd_rec(NT), WER = WE_N / 256, WDR = WD_N / 256;
  constexpr int NK4 = NK / 4, NK2 = (NK % 4) / 2, KSF = nvp_ks_floats(NT);
  static_assert(NK % 2 == 0, "Cin must be a multiple of 8");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  float* E = lds;                         // [NBUF][16 rows, NVP_EROW pitch, swizzled starts]
  float* WE = E + NBUF * NVP_EBUF;        // [2][WE_N]
  float* WD = WE + 2 * WE_N;              // [2][WD_N]
  const int Cin = a.Cin, Cout = a.Cout, Cv = a.Cv, th = a.th, tw = a.tw, pw = tw >> 1;        // this launch computes output channels co0 .. co0 + Cv - 1 of the Cout
  const int iw = tw + 2, npx = (th + 2) * iw;
  const int tiles_x = (a.Wo + tw - 1) / tw;
  const int n = blockIdx.y;
  const int ty = (int)blockIdx.x / tiles_x;
  const int oy0 = ty * th, ox0 = ((int)blockIdx.x - ty * tiles_x) * tw;
  const int iy0 = oy0 - a.pt, ix0 = ox0 - a.pl;
  const int nchunk = a.Chid >> 4;
  const int wspan = MERGE ? a.gmerge * a.cpg : a.cpg;          // chunks of this workgroup
  const int ch0 = blockIdx.z * wspan, ch1 = min(nchunk, ch0 + wspan);
  const bool lead = blockIdx.z == 0;
  unsigned long long* stamp = a.stamps ? a.stamps + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 : nullptr;
  int stamp_i = 0;
  auto STAMP = [&]() { if (stamp && tid == 0 && stamp_i < 32) stamp[stamp_i] = wall_clock64(); ++stamp_i; };
  STAMP();

  // ---- residual (hidden-channel group 0 only): C layout of the project accumulators, row = pair 16 wave + 4 lq + r, pixel 2 px + m2 ----------
  const float* rp = (a.res && lead) ? a.res + (size_t)n * a.Ho * a.Wo * Cout : nullptr;
  constexpr bool RES_EARLY = !MERGE && NT <= 4 && NK < 12;
  float resv[2][4][RES_EARLY ? NT : 1];
  int obase[4];                 // element offset of (pixel of pair r, m2 = 0, channel lp), or -1
  bool ok2[4];                  // the pair's odd pixel is inside the output too
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int P = wave * 16 + lq * 4 + r;
    const int oy = (int)(((unsigned)P * a.inv_tw) >> 20), px = P - oy * pw;
    const int gy = oy0 + oy, gx = ox0 + 2 * px;
    const bool okp = P < th * pw && gy < a.Ho && gx < a.Wo;
    obase[r] = okp ? (gy * a.Wo + gx) * Cout + a.co0 + lp : -1;
    ok2[r] = okp && gx + 1 < a.Wo;
  }
  // every global load of the prologue is issued before anything waits on one: batches per slab, no load inside a run-time loop (a loop over the
  // slabs around each load costs one memory round trip per iteration -- 6 us of the old kernel's prologue + epilogue)
  float bvv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bvv[t] = a.bp[t * 16 + lp];           // padded to the n-tiles by the host
  f32x4 tot[2][MERGE == 1 ? NT : 1];
  if constexpr (MERGE == 1) {                 // the residual's slabs are summed first, in slab order (as the unmerged epilogue does)
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool okr = rp && obase[r] >= 0 && (m2 == 0 || ok2[r]) && t * 16 + lp < Cv;
          tot[m2][t][r] = okr ? rp[(unsigned)(obase[r] + m2 * Cout + t * 16)] : 0.f;
        }
    for (int sl = 1; sl < (rp ? a.res_slabs : 0); ++sl)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool okr = obase[r] >= 0 && (m2 == 0 || ok2[r]) && t * 16 + lp < Cv;
            tot[m2][t][r] += okr ? (rp + (size_t)sl * a.res_slab_stride)[(unsigned)(obase[r] + m2 * Cout + t * 16)] : 0.f;
          }
  }
  float r1[2][4][RES_EARLY ? NT : 1];
  if (RES_EARLY && rp) {
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) {
          const bool okr = obase[r] >= 0 && (m2 == 0 || ok2[r]) && t * 16 + lp < Cv;
          resv[m2][r][t] = rp[okr ? (unsigned)(obase[r] + m2 * Cout + t * 16) : 0u];
        }
    if (a.res_slabs > 1) {
      const float* rp1 = rp + a.res_slab_stride;          // uniform bases + 32-bit lane offsets: no 64-bit address pair per load
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) {
            const bool okr = obase[r] >= 0 && (m2 == 0 || ok2[r]) && t * 16 + lp < Cv;
            r1[m2][r][t] = rp1[okr ? (unsigned)(obase[r] + m2 * Cout + t * 16) : 0u];
          }
      for (int sl = 2; sl < a.res_slabs; ++sl)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) {
              const bool okr = obase[r] >= 0 && (m2 == 0 || ok2[r]) && t * 16 + lp < Cv;
              r1[m2][r][t] += (rp + (size_t)sl * a.res_slab_stride)[okr ? (unsigned)(obase[r] + m2 * Cout + t * 16) : 0u];
            }
    }
  }

  // ---- weights: global -> registers -> LDS double buffers, two chunks ahead (as nv_xblock_kernel) ---------------------------------------
  float wes[WER], wds[WDR];
  auto fetch_we = [&](int ch) {
    const float* wb = a.we + (size_t)ch * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wes[i] = wb[256 * i];
  };
  auto fetch_wd = [&](int ch) {
    const float* wc = a.wp + (size_t)ch * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wds[i] = wc[256 * i];
  };
  auto store_we = [&](int buf) {
    float* wl = WE + buf * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wl[256 * i] = wes[i];
  };
  auto store_wd = [&](int buf) {
    float* wm = WD + buf * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wm[256 * i] = wds[i];
  };
  if (ch0 < ch1) { fetch_we(ch0); fetch_wd(ch0); }

  // ---- input: the wave's 3 patch m-tiles, pixel mt*16 + lp, channels lq*NK .. lq*NK + NK - 1, summed over the producer's partial slabs ------
  float xr[3][NK];
  f32x4 maskc[3];               // C layout: pixel mt*16 + 4 lq + r is inside the image
  constexpr int LW = NK % 4 == 0 ? 4 : 2;          // lq * NK floats is 16-byte aligned only when NK is a multiple of 4
  auto ldx = [&](float* dst, const float* base, unsigned off) {        // uniform base + 32-bit lane offset
    if constexpr (LW == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(base + off); dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    else { const f32x2 v = *reinterpret_cast<const f32x2*>(base + off); dst[0] = v[0]; dst[1] = v[1]; }
  };
  bool okm[3]; unsigned offm[3];
  const float* ip = a.in + (size_t)n * a.H * a.W * Cin;
  constexpr bool MULTI_IN = nvp_multi_in(NK);
  float x1[3][MULTI_IN ? NK : 1];
  {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int p = (wave + 4 * m) * 16 + lp;
      const int iy = (int)(((unsigned)p * a.inv_iw) >> 20), ix = p - iy * iw;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = p < npx && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      okm[m] = ok; offm[m] = (unsigned)((ok ? (gy * a.W + gx) * Cin : 0) + lq * NK);
#pragma unroll
      for (int s = 0; s < NK; s += LW) ldx(xr[m] + s, ip, offm[m] + s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pc = (wave + 4 * m) * 16 + lq * 4 + r;
        const int cy = (int)(((unsigned)pc * a.inv_iw) >> 20), cx = pc - cy * iw;
        const int hy = iy0 + cy, hx = ix0 + cx;
        maskc[m][r] = (pc < npx && hy >= 0 && hy < a.H && hx >= 0 && hx < a.W) ? 1.f : 0.f;
      }
    }
    if constexpr (MULTI_IN) if (a.in_slabs > 1) {
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int s = 0; s < NK; s += LW) ldx(x1[m] + s, ip + a.in_slab_stride, offm[m] + s);
      for (int sl = 2; sl < a.in_slabs; ++sl)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int s = 0; s < NK; s += LW) { float t[LW]; ldx(t, ip + (size_t)sl * a.in_slab_stride, offm[m] + s); for (int e = 0; e < LW; ++e) x1[m][s + e] += t[e]; }
    }
  }
  // ---- depthwise: pair P = 16 wave + lp of the th x (tw / 2) pair grid -> patch pixel of tap (0, 0) of its even pixel --------------------------
  int ebase;
  {
    const int P = wave * 16 + lp;
    const int oy = (int)(((unsigned)P * a.inv_tw) >> 20), px = P - oy * pw;
    ebase = P < th * pw ? oy * iw + 2 * px : 0;
  }
  const int c0 = 4 * (lq >> 1) + 8 * (lq & 1);           // hidden channel of k-step 0 for this lane group; k-step ks: c0 + ks, row start + 260 ks
  const int rd0 = nvp_row(0) + c0 * NVP_EROW + 16 * (lq >> 1) + 32 * (lq & 1) + ebase;
  const int wr0 = lp * NVP_EROW + 4 * (lp & 7) + 32 * (lp >> 3) + lq * 4;
  STAMP();
  if (ch0 < ch1) { store_we(0); store_wd(0); }
  if (ch0 + 1 < ch1) { fetch_we(ch0 + 1); fetch_wd(ch0 + 1); }
  STAMP();
  __syncthreads();
  STAMP();
  // the sums over the partial slabs, now that every load has been issued (loads return in order: nothing here waits longer than the input would)
  if (RES_EARLY && rp && a.res_slabs > 1) {
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) resv[m2][r][t] += r1[m2][r][t];
  }
  if constexpr (MULTI_IN) if (a.in_slabs > 1) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int s = 0; s < NK; ++s) xr[m][s] += x1[m][s];
  }
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int s = 0; s < NK; ++s) xr[m][s] = okm[m] ? xr[m][s] : 0.f;

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float lo_e = nvp_lo(a.act_e), hi_e = nvp_hi(a.act_e), lo_d = nvp_lo(a.act_d), hi_d = nvp_hi(a.act_d);
  int gleft = a.cpg;                     // MERGE: chunks left in the current group
  bool gfirst = true;

  for (int ch = ch0; ch < ch1; ++ch) {
    const int wb_i = (ch - ch0) & 1;
    float* Eb = E + (NBUF == 2 ? wb_i : 0) * NVP_EBUF;
    if (NBUF == 1 && ch > ch0) __syncthreads();            // one E buffer (more workgroups per CU): everybody is done reading the previous chunk
    {
      const float* wl = WE + wb_i * WE_N;
      float wk[NK];
#pragma unroll
      for (int j = 0; j < NK4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wl + (j * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) wk[j * 4 + e] = v[e];
      }
      if (NK2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(wl + NK4 * 256 + lane * 2);
        wk[NK4 * 4] = v[0]; wk[NK4 * 4 + 1] = v[1];
      }
      const float bias = wl[NK * 64 + lp];
      f32x4 c[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) c[m] = maskc[m] * bias;
#pragma unroll
      for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int m = 0; m < 3; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[m][s], wk[s], c[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = nvp_clamp(c[m][r], lo_e, hi_e);
        *reinterpret_cast<f32x4*>(Eb + wr0 + (wave + 4 * m) * 16) = o;
      }
    }
    STAMP();
    if (ch + 1 < ch1) store_we(wb_i ^ 1);              // WE[wb_i ^ 1] was last read by the previous chunk's expand stage
    if (ch + 2 < ch1) fetch_we(ch + 2);
    STAMP();
    __syncthreads();
    STAMP();
    const float* wd = WD + wb_i * WD_N;
    const float* r0 = Eb + rd0;
    const float* r1 = r0 + iw;
    const float* r2 = r1 + iw;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 wa = *reinterpret_cast<const f32x4*>(wd + (lq * 4 + ks) * 12);
      const f32x4 wb = *reinterpret_cast<const f32x4*>(wd + (lq * 4 + ks) * 12 + 4);
      const f32x4 wc = *reinterpret_cast<const f32x4*>(wd + (lq * 4 + ks) * 12 + 8);       // tap 8, bias, 0, 0
      const float tap[9] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3], wc[0]};
      f32x2 d = {wc[1], wc[1]};
      const float* rr[3] = {r0, r1, r2};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x2 A = *reinterpret_cast<const f32x2*>(rr[ky] + ks * 260);
        const f32x2 B = *reinterpret_cast<const f32x2*>(rr[ky] + ks * 260 + 2);
        d = __builtin_elementwise_fma(A, f32x2{tap[ky * 3], tap[ky * 3]}, d);
        d = __builtin_elementwise_fma(f32x2{A[1], B[0]}, f32x2{tap[ky * 3 + 1], tap[ky * 3 + 1]}, d);
        d = __builtin_elementwise_fma(B, f32x2{tap[ky * 3 + 2], tap[ky * 3 + 2]}, d);
      }
      d[0] = nvp_clamp(d[0], lo_d, hi_d); d[1] = nvp_clamp(d[1], lo_d, hi_d);
      nvp_project<NT>(wd + 256 + ks * KSF, lane, d[0], d[1], acc);
    }
    if constexpr (MERGE == 1) {
      if (--gleft == 0 || ch + 1 == ch1) {               // the group's partial sum is complete (uniform branch)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv = (gfirst && lead) ? bvv[t] : 0.f;
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[m2][t][r] = (acc[m2][t][r] + bv) + tot[m2][t][r];
            acc[m2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        gleft = a.cpg; gfirst = false;
      }
    }
    STAMP();
    if (ch + 1 < ch1) store_wd(wb_i ^ 1);              // everybody passed this chunk's barrier, so chunk ch - 1 is done with WD[wb_i ^ 1]
    if (ch + 2 < ch1) fetch_wd(ch + 2);
    STAMP();
  }

  // ---- epilogue: hidden-channel group g > 0 stores its bare partial sum into slab g; group 0 adds the bias and the residual -------------------
  float* op = a.out + (size_t)blockIdx.z * a.out_slab_stride + (size_t)n * a.Ho * a.Wo * Cout;
  const float lo_p = nvp_lo(a.act_p), hi_p = nvp_hi(a.act_p);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t * 16 + lp >= Cv) continue;
    const float bv = lead ? bvv[t] : 0.f;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (obase[r] < 0 || (m2 == 1 && !ok2[r])) continue;
        const int o = obase[r] + m2 * Cout + t * 16;
        if constexpr (MERGE == 1) { op[o] = nvp_clamp(tot[m2][t][r], lo_p, hi_p); continue; }
        float v = acc[m2][t][r] + bv;
        if constexpr (RES_EARLY) { if (rp) v += resv[m2][r][t]; }
        else if (rp) { float rs = rp[o]; for (int sl = 1; sl < a.res_slabs; ++sl) rs += rp[(size_t)sl * a.res_slab_stride + o]; v += rs; }
        op[o] = nvp_clamp(v, lo_p, hi_p);
      }
  }
  STAMP();
}

// ---- the network's first block: conv 3x3 from the u8 frame -> dw 3x3 -> pw project, same depthwise organisation ---------------------------------
// One chunk (the first conv has 16 output channels = the block's hidden channels), so no weight pipeline: the u8 receptive field of the patch,
// the conv's B fragments and the depthwise + project record are fetched in ONE batch of loads at the top (the old form, nv_block_kernel MODE 1,
// walked the u8 patch in a run-time loop: four memory round trips per workgroup, and was bound by the LDS reads of its per-pixel depthwise stage).
//   conv: M = patch pixels (12 m-tiles), K = 9 taps + 1 bias row (+ 2 zero rows), N = 16; A = (u8 - 128) / 128 read from the u8 copy in LDS; a pixel
//   outside the conv's output map zeroes its whole A row, bias included, so the depthwise stage sees its zero padding.
// NCH = chunks of 16 hidden channels: 1 for a first conv of <= 16 output channels (MobileNetV2 x 0.35 / 0.5), 2 for <= 32 (x 0.75: 24, x 1.0: 32; the
// second chunk's missing channels are zero weights, whose ReLU6(0) contributes exact zeros).  With two chunks the conv stage runs once per chunk over the
// same u8 copy, into the same E rows, and the project accumulators run over both.
template <int NT, int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT == 1 ? (NCH == 1 ? 7 : 5) : NT == 2 ? (NCH == 1 ? 5 : 4) : NT <= 4 ? 4 : 2))) void nv_fpair_kernel(NvBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WD_N = nvp_wd_rec(NT), WDR = NCH * WD_N / 256, KSF = nvp_ks_floats(NT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  float* E = lds;                                   // [16 rows], as nv_pblock_kernel
  float* WD = E + NVP_EBUF;                         // [NCH][WD_N]
  float* W0 = WD + NCH * WD_N;                      // [NCH][3][64] B fragments of the conv (n-tile chunk of pack_nv_conv0's [3][2][64])
  uint8_t* U8 = reinterpret_cast<uint8_t*>(W0 + NCH * 192);      // [NVP_U8_ROWS][NVP_U8_PITCH]
  const int Cout = a.Cout, th = a.th, tw = a.tw, pw = tw >> 1;
  const int iw = tw + 2, npx = (th + 2) * iw;
  const int tiles_x = (a.Wo + tw - 1) / tw, ntiles = tiles_x * ((a.Ho + th - 1) / th);
  const int n = blockIdx.y;
  const int cs = a.c0_stride;
  const int UH = (th + 1) * cs + 3, UW = (iw - 1) * cs + 3;
  // a workgroup walks a.tpw consecutive tiles: the u8 patch of the next one is in flight while this one is computed (a tile is ~3.5 us of work
  // behind ~3.3 us of load latency), and the weights are fetched once
  const int tpw = a.tpw > 0 ? a.tpw : 1;
  const int t_begin = (int)blockIdx.x * tpw, t_end = min(ntiles, t_begin + tpw);
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;
  int stamp_i = 0;
  auto STAMP = [&]() { if (stamp && tid == 0 && stamp_i < 32) stamp[stamp_i] = wall_clock64(); ++stamp_i; };
  STAMP();
  const uint8_t* ipx = a.img + (size_t)n * a.img_istride;
  // tile-independent index arithmetic: this thread's 4 bytes of the u8 copy, its 4 output pairs, its 3 conv rows, its depthwise window
  // (y, x) pairs packed as y << 8 | x, one register each; y = 0xffff (far outside any frame) marks an entry that does not exist
  int u_yx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + tid;
    const int uy = idx / NVP_U8_PITCH, ux = idx - uy * NVP_U8_PITCH;
    u_yx[i] = (idx < NVP_U8_ROWS * NVP_U8_PITCH && uy < UH && ux < UW) ? (uy << 8 | ux) : (0xffff << 8);
  }
  int p_yx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int P = wave * 16 + lq * 4 + r;
    const int py = (int)(((unsigned)P * a.inv_tw) >> 20);
    p_yx[r] = P < th * pw ? (py << 8 | 2 * (P - py * pw)) : (0xffff << 8);
  }
  int c_yx[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int p = (wave + 4 * m) * 16 + lp;
    const int cy = (int)(((unsigned)p * a.inv_iw) >> 20);
    c_yx[m] = p < npx ? (cy << 8 | (p - cy * iw)) : (0xffff << 8);
  }
  int ebase;
  {
    const int P = wave * 16 + lp;
    const int oy = (int)(((unsigned)P * a.inv_tw) >> 20), px = P - oy * pw;
    ebase = P < th * pw ? oy * iw + 2 * px : 0;
  }
  const int c0 = 4 * (lq >> 1) + 8 * (lq & 1);
  const int rd0 = nvp_row(0) + c0 * NVP_EROW + 16 * (lq >> 1) + 32 * (lq & 1) + ebase;
  const int wr0 = lp * NVP_EROW + 4 * (lp & 7) + 32 * (lp >> 3) + lq * 4;
  int toff[3];                   // byte offset of this lane group's tap in k-step ks (k = 4 ks + lq; 0 where k is not a tap)
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) { const int k = ks * 4 + lq; toff[ks] = k < 9 ? (k / 3) * NVP_U8_PITCH + (k % 3) : 0; }
  uint8_t ub[4];
  unsigned ub_in = 0;            // bit i: byte i lies inside the frame.  The loaded values are not touched before they are stored to LDS one tile later:
                                 // a select right after the load would wait for it on the spot and expose the latency the prefetch is there to hide
  auto load_u8 = [&](int t) {
    const int ty = t / tiles_x;
    const int uy0 = (ty * th - a.pt) * cs - a.c0_pt, ux0 = ((t - ty * tiles_x) * tw - a.pl) * cs - a.c0_pl;
    ub_in = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = uy0 + (u_yx[i] >> 8), xx = ux0 + (u_yx[i] & 255);
      const bool in = yy >= 0 && yy < a.H0 && xx >= 0 && xx < a.W0;
      ub[i] = ipx[in ? (unsigned)(yy * a.img_stride + xx) : 0u];
      ub_in |= in ? 1u << i : 0u;
    }
  };
  // ---- one batch of loads: first u8 patch, conv fragments, depthwise + project record, project bias ---------------------------------------------
  if (t_begin < t_end) load_u8(t_begin);
  float w0v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) w0v[c] = tid < 192 ? a.w0[(tid >> 6) * 128 + c * 64 + (tid & 63)] : 0.f;
  float wds[WDR];
#pragma unroll
  for (int i = 0; i < WDR; ++i) wds[i] = a.wp[i * 256 + tid];
  float bvv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bvv[t] = a.bp[t * 16 + lp];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (tid < 192) W0[c * 192 + tid] = (tid >> 6) * 4 + ((tid & 63) >> 4) < 9 ? w0v[c] * 0.0078125f : w0v[c];       // tap rows carry the 1 / 128
  if (tid < NVP_U8_DUMMY) U8[NVP_U8_ROWS * NVP_U8_PITCH + tid] = 128;                               // the all-128 window of pixels outside the map
#pragma unroll
  for (int i = 0; i < WDR; ++i) WD[i * 256 + tid] = wds[i];
  // the bias counts as arrived from here on: first used inside the tile loop, it would otherwise make the compiler wait for EVERY outstanding memory
  // operation (vmcnt(0): the prefetched bytes and the previous pair's stores) before each pair of output stores
#pragma unroll
  for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(bvv[t]));
  auto store_u8 = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i * 256 + tid < NVP_U8_ROWS * NVP_U8_PITCH) U8[i * 256 + tid] = (ub_in >> i) & 1 ? ub[i] : (uint8_t)128;       // 128 -> exactly 0 after x - 128
  };
  if (t_begin < t_end) store_u8();
  const float lo0 = nvp_lo(a.act0), hi0 = nvp_hi(a.act0), lo_d = nvp_lo(a.act_d), hi_d = nvp_hi(a.act_d), lo_p = nvp_lo(a.act_p), hi_p = nvp_hi(a.act_p);
  float* op = a.out + (size_t)n * a.Ho * a.Wo * Cout;

  for (int t = t_begin; t < t_end; ++t) {
    const int ty = t / tiles_x;
    const int oy0 = ty * th, ox0 = (t - ty * tiles_x) * tw;
    const int iy0 = oy0 - a.pt, ix0 = ox0 - a.pl;
    STAMP();
    __syncthreads();          // this tile's u8 copy (and, first time round, the weights) are in LDS; everybody is done with the previous tile's E
    STAMP();
    if (t + 1 < t_end) load_u8(t + 1);

    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[m][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int chn = 0; chn < NCH; ++chn) {
      if (chn > 0) __syncthreads();          // everybody is done reading the previous chunk's E rows
      // ---- first conv on the matrix pipe -> E rows (channel lp of this chunk, pixels 4 lq .. + 3 of m-tile mt) ---------------------------------
      // A = u8 - 128 (the 1/128 of the normalisation sits in the tap rows of B: a power of two, so every product and the result are bit-identical to
      // ((u8 - 128) / 128) * w); a pixel outside the conv's output reads the all-128 dummy window, i.e. zeros, and gets no bias
      {
        float wf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wf[ks] = W0[chn * 192 + ks * 64 + lane];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int cy = c_yx[m] >> 8, cx = c_yx[m] & 255;
          const int gy = iy0 + cy, gx = ix0 + cx;
          const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          const uint8_t* up = U8 + (ok ? (cy * cs) * NVP_U8_PITCH + cx * cs : NVP_U8_ROWS * NVP_U8_PITCH);
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) {
            float av = (float)up[toff[ks]] - 128.0f;           // k = 4 ks + lq: a tap for k < 9 (toff), the bias row for k == 9, nothing beyond
            if (ks == 2) av = lq == 0 ? av : (lq == 1 && ok ? 1.f : 0.f);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[ks], c, 0, 0, 0);
          }
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = nvp_clamp(c[r], lo0, hi0);
          *reinterpret_cast<f32x4*>(E + wr0 + (wave + 4 * m) * 16) = o;
        }
      }
      STAMP();
      __syncthreads();
      STAMP();

      // ---- depthwise (pixel pairs) + project, as one chunk of nv_pblock_kernel -------------------------------------------------------------------
      {
        const float* wdc = WD + chn * WD_N;
        const float* r0 = E + rd0;
        const float* r1 = r0 + iw;
        const float* r2 = r1 + iw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(wdc + (lq * 4 + ks) * 12);
          const f32x4 wb = *reinterpret_cast<const f32x4*>(wdc + (lq * 4 + ks) * 12 + 4);
          const f32x4 wc = *reinterpret_cast<const f32x4*>(wdc + (lq * 4 + ks) * 12 + 8);
          const float tap[9] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3], wc[0]};
          f32x2 d = {wc[1], wc[1]};
          const float* rr[3] = {r0, r1, r2};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const f32x2 A = *reinterpret_cast<const f32x2*>(rr[ky] + ks * 260);
            const f32x2 B = *reinterpret_cast<const f32x2*>(rr[ky] + ks * 260 + 2);
            d = __builtin_elementwise_fma(A, f32x2{tap[ky * 3], tap[ky * 3]}, d);
            d = __builtin_elementwise_fma(f32x2{A[1], B[0]}, f32x2{tap[ky * 3 + 1], tap[ky * 3 + 1]}, d);
            d = __builtin_elementwise_fma(B, f32x2{tap[ky * 3 + 2], tap[ky * 3 + 2]}, d);
          }
          d[0] = nvp_clamp(d[0], lo_d, hi_d); d[1] = nvp_clamp(d[1], lo_d, hi_d);
          nvp_project<NT>(wdc + 256 + ks * KSF, lane, d[0], d[1], acc);
        }
      }
    }
    STAMP();
    // the next tile's bytes go to LDS BEFORE this tile's output stores are issued (the conv stage, the last reader of the u8 copy, is behind every wave's
    // second barrier): loads and stores share one in-order counter, so a wait for the prefetched bytes placed after the stores would also wait for them
    store_u8();                                 // (unconditional: behind the last tile it rewrites the same bytes; a condition equal to the loop's own lets the
    __builtin_amdgcn_sched_barrier(0);          //  compiler sink the LDS writes onto the back edge, behind the global stores)
    // ---- store: C layout row = pair 16 wave + 4 lq + r, pixel 2 px + m2, column = channel lp of n-tile t -------------------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gy = oy0 + (p_yx[r] >> 8), gx = ox0 + (p_yx[r] & 255);
      if (gy >= a.Ho || gx >= a.Wo) continue;
      const unsigned ob = (unsigned)((gy * a.Wo + gx) * Cout + lp);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        if (tt * 16 + lp >= Cout) continue;
        op[ob + tt * 16] = nvp_clamp(acc[0][tt][r] + bvv[tt], lo_p, hi_p);
        if (gx + 1 < a.Wo) op[ob + Cout + tt * 16] = nvp_clamp(acc[1][tt][r] + bvv[tt], lo_p, hi_p);
      }
    }
    STAMP();
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
// n-tiles of 16 output channels one launch of the pixel-pair kernels computes: any count up to 8 (128 channels); wider outputs are computed in two
// launches over channel halves (NvBlockArgs::co0 / Cv), each of which repeats the expand + depthwise stages
int nv_pblock_ntiles(int cout) { const int nt = (cout + 15) / 16; return nt >= 1 && nt <= 8 ? nt : -1; }
int nv_pblock_halves(int cout) { return cout <= 128 ? 1 : cout <= 256 ? 2 : -1; }
int nv_pblock_half_cout(int cout, int half) { if (cout <= 128) return cout; const int c0 = (cout / 2 + 15) / 16 * 16; return half == 0 ? c0 : cout - c0; }
static bool nvp_shape_exists(int nk, int nt);
bool nv_pblock_single_input(int cin) { return !nvp_multi_in(cin / 4); }
bool nv_pblock_supported(int cin, int chid, int cout, int stride) {
  if (stride != 1 || (cin & 7) || (chid & 15) || nv_pblock_halves(cout) < 0) return false;
  for (int hf = 0; hf < nv_pblock_halves(cout); ++hf)
    if (!nvp_shape_exists(cin / 4, nv_pblock_ntiles(nv_pblock_half_cout(cout, hf)))) return false;
  return true;
}
// tile: th x tw output pixels, tw even, th * tw / 2 <= 64 pairs, (th + 2)(tw + 2) <= 192 patch pixels; maximise the useful fraction of the MFMA rows
// (c0_stride > 0: the first block -- the u8 receptive field of the patch must fit its LDS copy)
void nv_pblock_tile(int Ho, int Wo, int* th_out, int* tw_out, int c0_stride) {
  double best = -1; int bth = 8, btw = 16;
  for (int th = 1; th <= 32; ++th)
    for (int tw = 4; tw <= 64; tw += 2) {
      if (th * tw > 128 || (th + 2) * (tw + 2) > NVP_PATCH) continue;
      if (c0_stride > 0 && ((th + 1) * c0_stride + 3 > NVP_U8_ROWS || (tw + 1) * c0_stride + 3 > NVP_U8_PITCH)) continue;
      const long tiles = (long)((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
      const double eff = (double)Ho * Wo / (double)(tiles * 128);
      const double halo = (double)(th * tw) / (double)((th + 2) * (tw + 2));
      const double score = eff * (0.75 + 0.25 * halo);
      if (score > best + 1e-9) { best = score; bth = th; btw = tw; }
    }
  *th_out = bth; *tw_out = btw;
}
size_t pack_nv_expand_pair_floats(int chid, int cin) { return (size_t)(chid / 16) * nvp_we_rec(cin / 4); }
// expand record: lane (n = lane & 15, lq = lane >> 4) holds W[chunk*16 + n][lq*nk + s] for k-step s: [s / 4][lane][4], a [lane][2] remainder, bias[16]
void pack_nv_expand_pair(const float* w /*[chid][cin]*/, const float* b, int chid, int cin, float* dst) {
  const int nk = cin / 4, rec = nvp_we_rec(nk), nk4 = nk / 4;
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < rec; ++i) d[i] = 0.f;
    for (int l = 0; l < 64; ++l) {
      const float* wr = w + (size_t)(ch * 16 + (l & 15)) * cin + (l >> 4) * nk;
      for (int s = 0; s < nk4 * 4; ++s) d[((s >> 2) * 64 + l) * 4 + (s & 3)] = wr[s];
      for (int s = nk4 * 4; s < nk; ++s) d[nk4 * 256 + l * 2 + (s - nk4 * 4)] = wr[s];
    }
    for (int c = 0; c < 16; ++c) d[nk * 64 + c] = b[ch * 16 + c];
  }
}
size_t pack_nv_dwproj_pair_floats(int chid, int nt) { return (size_t)((chid + 15) / 16) * nvp_wd_rec(nt); }
// depthwise + project record: [lq][ks][12] = 9 taps, bias, 0, 0 of hidden channel c(ks, lq) = ks + 4 (lq >> 1) + 8 (lq & 1); then at 256 per k-step ks
// (nvp_ks_floats(nt) floats each) the groups of n-tiles nvp_project reads: Wp[co0 + t*16 + (lane & 15)][chunk*16 + c(ks, lane >> 4)].  `cout` output channels
// starting at row `co0` of wp; hidden channels beyond chid (a last chunk of fewer than 16: the first block's 24) are zero weights
void pack_nv_dwproj_pair(const float* wd /*[chid][9]*/, const float* bd, const float* wp /*[..][chid]*/, int cout, int chid, int nt, float* dst, int co0) {
  const int rec = nvp_wd_rec(nt), ksf = nvp_ks_floats(nt), ng4 = nt / 4, rem = nt % 4, remw = nvp_rem_width(nt);
  for (int ch = 0; ch < (chid + 15) / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < rec; ++i) d[i] = 0.f;
    for (int lq = 0; lq < 4; ++lq)
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ch * 16 + ks + 4 * (lq >> 1) + 8 * (lq & 1);
        if (c >= chid) continue;
        for (int t = 0; t < 9; ++t) d[(lq * 4 + ks) * 12 + t] = wd[(size_t)c * 9 + t];
        d[(lq * 4 + ks) * 12 + 9] = bd[c];
      }
    for (int ks = 0; ks < 4; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int lq = l >> 4, k = ch * 16 + ks + 4 * (lq >> 1) + 8 * (lq & 1);
        auto W = [&](int t) { const int co = t * 16 + (l & 15); return (co < cout && k < chid) ? wp[(size_t)(co0 + co) * chid + k] : 0.f; };
        float* dk = d + 256 + ks * ksf;
        for (int hf = 0; hf < ng4; ++hf)
          for (int e = 0; e < 4; ++e) dk[(hf * 64 + l) * 4 + e] = W(hf * 4 + e);
        for (int e = 0; e < rem; ++e) dk[ng4 * 256 + l * remw + e] = W(ng4 * 4 + e);
      }
  }
}

template <int NT, int NK, int NBUF, int MERGE = 0>
static hipError_t launch_pblock_b(const NvBlockArgs& a, int n, int groups, hipStream_t s) {      // groups: grid.z (workgroup groups of this launch)
  const size_t lds = sizeof(float) * (NBUF * (size_t)NVP_EBUF + 2 * nvp_we_rec(NK) + 2 * nvp_wd_rec(NT));
  auto k = nv_pblock_kernel<NT, NK, NBUF, MERGE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long tiles = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th);
  hipLaunchKernelGGL(k, dim3((unsigned)tiles, n, groups), dim3(256), lds, s, a);
  return hipGetLastError();
}
static size_t nvp_lds_bytes(int nt, int nk, int nbuf) { return sizeof(float) * ((size_t)nbuf * NVP_EBUF + 2 * nvp_we_rec(nk) + 2 * nvp_wd_rec(nt)); }
// workgroups of this block shape that can be resident at once (registers: nvp_min_waves; LDS with `nbuf` E buffers)
long nv_pblock_slots(int cin, int cout, int ncu, int nbuf) {
  const int nt = nv_pblock_ntiles(nv_pblock_half_cout(cout, 0)), nk = cin / 4;
  const long per_cu = std::min<long>(nvp_min_waves(nt, nk), (long)((160 * 1024) / nvp_lds_bytes(nt, nk, nbuf)));
  return (long)(ncu > 0 ? ncu : 256) * per_cu;
}
// E double-buffered (one barrier per chunk) when every workgroup of the launch is resident at once anyway; one buffer (two barriers per chunk,
// 16 KB less LDS: 5-6 workgroups per CU instead of 3-4) when the launch would otherwise run in rounds.  The dispatch is never perfectly even:
// "fits" means 85 % of the slots.  D2FE_NV_NBUF=1|2 (NvBlockArgs::nbuf) forces.
template <int NT, int NK, int MERGE = 0>
static hipError_t launch_pblock_t(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  const int force = a.nbuf;            // D2FE_NV_NBUF, read when the network was loaded
  const long wgs = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th) * n * groups;
  const bool one = force ? force == 1 : wgs * 100 > nv_pblock_slots(a.Cin, a.Cout, a.ncu, 2) * 85;
  return one ? launch_pblock_b<NT, NK, 1, MERGE>(a, n, groups, s) : launch_pblock_b<NT, NK, 2, MERGE>(a, n, groups, s);
}
// the (k-steps of the expand GEMM = Cin / 4, n-tiles of the project GEMM) pairs that exist as kernels: MobileNetV2 x 0.35 (rounds 3-5) and x 0.75 (round 6:
// Cin 48 -> 48 / 72, 72 -> 72, 120 -> 120 and the two halves 128 + 112 of 120 -> 240) with their neighbours
#define NVP_SHAPES(X) \
  X(2, 1) X(2, 2) X(2, 4) X(2, 8) X(4, 1) X(4, 2) X(4, 4) X(4, 8) X(6, 1) X(6, 2) X(6, 3) X(6, 4) X(6, 8) X(8, 1) X(8, 2) X(8, 4) X(8, 8) \
  X(12, 3) X(12, 4) X(12, 5) X(14, 1) X(14, 2) X(14, 4) X(14, 7) X(14, 8) X(18, 5) X(18, 8) X(30, 7) X(30, 8)
// the shapes whose layers split into six or more groups per image (30 x 40 and 15 x 20 at alpha = 0.75): also as kernels that walk several groups per workgroup
#define NVP_MERGE_SHAPES(X) X(12, 3, 1) X(18, 5, 1)
bool nv_pblock_can_merge(int cin, int cv) {
  const int nk = cin / 4, nt = nv_pblock_ntiles(cv);
#define X(K, T, M) if (nk == K && nt == T) return true;
  NVP_MERGE_SHAPES(X)
#undef X
  return false;
}
static bool nvp_shape_exists(int nk, int nt) {
#define X(K, T) if (nk == K && nt == T) return true;
  NVP_SHAPES(X)
#undef X
  return false;
}
bool nv_fpair_supported(int c0_cout, int c0_stride, int dw_stride, int cout) {
  const int nt = nv_pblock_ntiles(cout);
  return (c0_cout == 16 || c0_cout == 24 || c0_cout == 32) && (c0_stride == 1 || c0_stride == 2) && dw_stride == 1 && (nt == 1 || nt == 2 || nt == 4 || nt == 8);
}
template <int NT, int NCH>
static hipError_t launch_fpair_t(const NvBlockArgs& a, int n, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)NVP_EBUF + NCH * (nvp_wd_rec(NT) + 192)) + NVP_U8_ROWS * NVP_U8_PITCH + NVP_U8_DUMMY;
  auto k = nv_fpair_kernel<NT, NCH>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long tiles = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th);
  const int tpw = a.tpw > 0 ? a.tpw : 1;
  hipLaunchKernelGGL(k, dim3((unsigned)((tiles + tpw - 1) / tpw), n, 1), dim3(256), lds, s, a);
  return hipGetLastError();
}
// the first block (conv from the u8 frame -> dw -> pw): a.we unused, a.wp = one pack_nv_dwproj_pair record, a.w0 = pack_nv_conv0
hipError_t launch_nv_fpair(const NvBlockArgs& a_in, int n, hipStream_t s) {
  NvBlockArgs a = a_in;
  {
    // tiles per workgroup: 3 once the launch is several rounds of workgroups deep anyway (measured 1..5: 684, 668, 663, 665, 666 us for the whole
    // 32-image NetVLAD call; D2FE_NV_FRONT_TPW forces)
    const int force = a.tpw;           // D2FE_NV_FRONT_TPW, read when the network was loaded
    const long tiles = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th) * n;
    a.tpw = force > 0 ? force : (tiles >= 16l * (a.ncu > 0 ? a.ncu : 256) ? 3 : 1);
  }
  if (a.stride != 1 || a.tw < 2 || (a.tw & 1) || a.th < 1 || a.th * a.tw > 128 || (a.th + 2) * (a.tw + 2) > NVP_PATCH) return hipErrorInvalidValue;
  if ((a.th + 1) * a.c0_stride + 3 > NVP_U8_ROWS || (a.tw + 1) * a.c0_stride + 3 > NVP_U8_PITCH) return hipErrorInvalidValue;
  const int iw = a.tw + 2, pw = a.tw / 2;
  a.inv_iw = ((1u << 20) + iw - 1) / iw; a.inv_tw = ((1u << 20) + pw - 1) / pw;
  if ((long)a.H0 * a.img_stride >= (1l << 31) || (long)a.Ho * a.Wo * a.Cout >= (1l << 31)) return hipErrorInvalidValue;
  const bool two = a.Chid > 16;        // hidden channels = the first conv's outputs: one or two chunks of 16
  switch (nv_pblock_ntiles(a.Cout)) {
    case 1: return two ? launch_fpair_t<1, 2>(a, n, s) : launch_fpair_t<1, 1>(a, n, s);
    case 2: return two ? launch_fpair_t<2, 2>(a, n, s) : launch_fpair_t<2, 1>(a, n, s);
    case 4: return two ? launch_fpair_t<4, 2>(a, n, s) : launch_fpair_t<4, 1>(a, n, s);
    case 8: return two ? launch_fpair_t<8, 2>(a, n, s) : launch_fpair_t<8, 1>(a, n, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_nv_pblock(const NvBlockArgs& a_in, int n, int groups, hipStream_t s) {
  NvBlockArgs a = a_in;
  if (a.stride != 1 || a.tw < 2 || (a.tw & 1) || a.th < 1 || a.th * a.tw > 128 || (a.th + 2) * (a.tw + 2) > NVP_PATCH) return hipErrorInvalidValue;
  const int iw = a.tw + 2, pw = a.tw / 2;
  a.inv_iw = ((1u << 20) + iw - 1) / iw; a.inv_tw = ((1u << 20) + pw - 1) / pw;       // exact for n < 2^20 / d: n < 256 here
  if ((long)a.H * a.W * a.Cin >= (1l << 31) || (long)a.Ho * a.Wo * a.Cout >= (1l << 31)) return hipErrorInvalidValue;
  if (a.Cv <= 0) { a.Cv = a.Cout; a.co0 = 0; }           // one launch computes every output channel
  if (a.co0 < 0 || a.co0 + a.Cv > a.Cout || a.Cv > 128) return hipErrorInvalidValue;
  const int nk = a.Cin / 4, nt = nv_pblock_ntiles(a.Cv);
  if (a.in_slabs > 1 && !nvp_multi_in(nk)) return hipErrorInvalidValue;
  if (a.gmerge > 1) {                    // `groups` stays the number of groups of the summation order; the launch has ceil(groups / gmerge) workgroup groups
    const int wg = (groups + a.gmerge - 1) / a.gmerge;
#define X(K, T, M) if (nk == K && nt == T) return launch_pblock_t<T, K, M>(a, n, wg, s);
    NVP_MERGE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
  }
#define X(K, T) if (nk == K && nt == T) return launch_pblock_t<T, K>(a, n, groups, s);
  NVP_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace d2fe

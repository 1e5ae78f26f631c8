// kernels.h -- launch interfaces of the gfx950 kernels behind include/d2fe.h (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Development instrumentation is compiled out of the product library: `python -m d2slam_amd.build --dev` builds
// lib/libd2fe_hip_dev.so with -DD2FE_DEVTOOLS (phase stamps, ablation switches, the d2fe_debug_* exports of include/d2fe_debug.h).
#ifdef D2FE_DEVTOOLS
#include <cstdlib>
// development switches come from the environment in the development library ONLY; the product library reads no environment variable
static inline int d2fe_dev_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define D2FE_ABL(a, bits) ((a).ablate & (bits))    /* timing experiments that switch parts of a kernel off (results are wrong with any bit set) */
#define D2FE_STAMP(buf, wg, i) do { if ((buf) && threadIdx.x == 0) (buf)[(size_t)(wg) * 16 + (i)] = wall_clock64(); } while (0)
#else
static inline int d2fe_dev_env(const char*, int dflt) { return dflt; }
#define D2FE_ABL(a, bits) (0)
#define D2FE_STAMP(buf, wg, i) do {} while (0)
#endif

namespace d2fe {

// ---- conv stack -----------------------------------------------------------------------------------
// Activations: NHWC fp32 in HBM, addressed as base + ((img*H + y)*W + x)*cstride + coff + c.
struct ConvArgs {
  const float* in;  int in_cstride;  int in_coff;
  float* out;       int out_cstride; int out_coff;
  int cout_real;            // channels actually stored (<= padded cout)
  const void* wpack;        // packed weights (layout depends on kernel family)
  const float* bias;        // padded to a multiple of 32
  int H, W;                 // conv input == output spatial size (before the optional 2x2 pool)
  int n_img;
  long in_img_stride;       // floats between images
  long out_img_stride;
  // fused conv1a prologue (CONV1B_FUSED only): the patch is computed from the u8 frame instead of being loaded
  const uint8_t* img; int img_stride; long img_istride; const float* w1a; const float* b1a;
  const float* zeros;       // >= 256 zero floats in HBM (source of the LDS-DMA copies of out-of-image patch pixels)
  int tag;                  // 1: conv1b (the dominant launch gets its own kernel instantiation so that rocprofv3 --stats lists it by itself)
  int ablate;               // development library only (D2FE_ABL): 1 skip patch loads, 2 skip B reloads, 4 skip stores; always 0 in the product library
  int* work_ctr = nullptr;  // Winograd kernels: zeroed device counter -> work items are claimed dynamically (null: static round-robin split)
  int ncu = 0;              // compute units of the handle's device (d2fe_create reads it once): the persistent kernels size their grids on it
};

enum ConvShape {
  CONV_64_T8x32 = 0,    // Cin 64, 3x3, tile 8x32, BN 64           (conv1b, conv2a, conv2b, conv3a)
  CONV_128_T4x32,       // Cin 128, 3x3, tile 4x32, BN 128          (conv3b)
  CONV_128_T4x16,       // Cin 128, 3x3, tile 4x16, BN 128          (conv4a, conv4b, convPa|convDa)
  CONV_256_1x1_T4x16,   // Cin 256, 1x1, tile 4x16, BN 128          (convPb, convDb)
  CONV1B_FUSED,         // conv1a (1->64, from the u8 frame) fused into conv1b's patch staging, + ReLU + 2x2 pool
};

// precision: 0 = fp32 exact MFMA, 1 = fp16 hi/lo split MFMA
hipError_t launch_conv(ConvShape shape, int precision, bool pool, bool relu, int cout_pad, const ConvArgs& a,
                       hipStream_t s);

// convPb (1x1, 256 -> 65) in the fp32 modes: 64 cell channels on the matrix pipe + the dustbin on the vector pipe (conv1x1.hip); same bits as
// the generic kernels; hipErrorNotSupported when the tensors are not flat pixel lists
hipError_t launch_conv1x1_256_65(const ConvArgs& a, hipStream_t s);

// persistent producer/consumer kernels (conv_pc.hip); same arithmetic, selected by D2FE_CONV_PC (default 2: every layer but the fused conv1a+conv1b)
hipError_t launch_conv_pc(ConvShape shape, int precision, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s);

// Winograd F(2x2,3x3) fp32 kernels (conv_wino.hip, precision mode 2): 3x3 layers with Cin 64 / 128, cout_pad a multiple of 64
hipError_t launch_conv_wino(int cin, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s);
hipError_t launch_conv_wino_fused1b(int cout_pad, const ConvArgs& a, hipStream_t s);   // conv1a (u8 frame) + conv1b, ReLU, 2x2 pool
size_t packed_weight_floats_wino(int cout_pad, int cin);
void pack_weights_wino(const float* w /*[cout][cin][3][3]*/, int cout, int cin, int cout_pad, float* dst);

// conv1a: u8 gray [n][H][stride] -> NHWC fp32 [n][H][W][64], fused (float)u8 * (1/255), bias, ReLU.
hipError_t launch_conv1a(const uint8_t* img, int stride, long img_stride_bytes, int H, int W, int n,
                         const float* w9x64 /*[9][64]*/, const float* bias, float* out, hipStream_t s);

// host-side packing helpers (fragment order of the kernels above)
size_t packed_weight_floats_f32(int cout_pad, int cin, int ks);
void pack_weights_f32(const float* w /*[cout][cin][k][k]*/, int cout, int cin, int ks, int cout_pad, float* dst);
size_t packed_weight_halfs_f16x2(int cout_pad, int cin, int ks);
void pack_weights_f16x2(const float* w, int cout, int cin, int ks, int cout_pad, uint16_t* dst);

// ---- post-processing ------------------------------------------------------------------------------
// softmax(65) -> drop dustbin -> 8x8 unfold; writes dense semi (optional) and appends variant-B candidates
// (score > thr, inside borders) as u64 keys (score_bits << 32 | ~raster_idx) with one counter per image.
hipError_t launch_softmax_cand(const float* logits, int lstride, int Hc, int Wc, int n_img, float thr, int border,
                               float* semi, unsigned long long* cand, int* cand_count, long cand_cap, bool zero_counts, hipStream_t s);
// exact top-K (score desc, raster asc) / raster-ordered pass-through when count <= K (variant B).
hipError_t launch_select_b(const unsigned long long* cand, const int* cand_count, long cand_cap, int n_img, int W,
                           int max_kp, int cap, int always_sort, const float* semi, int H, float thr, int border, float* kps_xy,
                           float* scores, int32_t* kps_idx, int32_t* n_out, hipStream_t s);
// variant-B descriptor sampling (normalize_keypoints + grid_sample + normalize_descriptors).
hipError_t launch_sample_b(const float* desc_raw, int dstride, int dcoff, int Hc, int Wc, int n_img, const float* kps_xy,
                           const int32_t* n_kp, int cap, const int32_t* slotmap, int max_slots, float* desc_out, hipStream_t s);
// sparse descriptor head (postproc.hip): convDa + convDb only at the corner cells of the selected keypoints
hipError_t launch_desc_head_sparse(const float* kps_xy, const int32_t* n_kp, int cap, int Hc, int Wc, int n_img, const float* x,
                                   int x_cstride, long x_img_stride, const void* w_da, const float* b_da, const void* w_db,
                                   const float* b_db, uint8_t* flags, int32_t* slotmap, int32_t* cells, int32_t* count,
                                   int max_slots, float* out, float* mid, int mid_imgs, int img_w, int img_h, hipStream_t s);

// variant A (SuperPointONNX path): NMS2-exact and grid_sampler(align_corners=false) sampling with optional PCA
hipError_t launch_nms2_a(const float* semi, int H, int W, int n_img, float thr, int dist, float* aconf, int* clist,
                         unsigned long long* cand, int* cand_count, long cand_cap, int* ncand, hipStream_t s);
hipError_t launch_nms2_wrap_fix(const int* clist, const int* ncand, int H, int W, int n_img, float* kps_xy, int32_t* kps_idx,
                                const int32_t* n_kp, int cap, hipStream_t s);
hipError_t launch_sample_a(const float* desc_raw, int dstride, int dcoff, int Hc, int Wc, int img_w, int img_h, int n_img,
                           const float* kps_xy, const int32_t* n_kp, int cap, const float* comp_t, const float* mean,
                           int pca_dims, float* samp, int scap, float* cn, const int32_t* slotmap, int max_slots, float* desc_out,
                           hipStream_t s);

// ---- NetVLAD (netvlad.hip) -------------------------------------------------------------------------------------------
hipError_t launch_nv_conv0(const uint8_t* img, int stride_b, long img_stride, int H, int W, int Ho, int Wo, int cstride,
                           int cout, int act, const float* w, const float* b, float* out, int n, hipStream_t s);
hipError_t launch_nv_dw(const float* in, int H, int W, int C, int Ho, int Wo, int cstride, int act, const float* w, const float* b,
                        float* out, int n, hipStream_t s);
hipError_t launch_nv_pw(const float* in, long P, int Cin, int Cout, int CoutPad, int act, const float* w, const float* b,
                        const float* res, float* out, hipStream_t s);
hipError_t launch_nv_vlad(const float* x, int slabs, long slab_stride, int np, int D, int K, const float* aw, const float* aw_pack,
                          const float* ab, const float* cen, float* part, float* out, int n, hipStream_t s);
void pack_nv_assign(const float* aw /*[K][D]*/, int K, int D, float* dst);
size_t nv_vlad_part_floats(int np, int D, int K);
hipError_t launch_nv_pca(const float* x, int nfeat, const float* comp, const float* mean, int m, float* y, int n, hipStream_t s);

// fused MobileNetV2 blocks (netvlad_fused.hip): [pw expand ->] dw 3x3 -> pw project (+ residual) in one launch, the expanded
// tensor only ever in LDS; `front`: the block input is the first 3x3 convolution of the u8 frame, evaluated inside the staging
struct NvBlockArgs {
  const float* in;                 // NHWC [n][H][W][Cin]                     (unused when front)
  int in_slabs; long in_slab_stride;     // the input is the sum of `in_slabs` partial tensors, `in_slab_stride` floats apart
  const uint8_t* img; int img_stride; long img_istride; int H0, W0;           // front: the u8 frames
  int c0_stride, c0_pt, c0_pl, act0; const float* w0;                         // front: first conv as B fragments (pack_nv_conv0)
  float* out; long out_slab_stride;      // NHWC [n][Ho][Wo][Cout]; hidden-channel group g writes slab g
  const float* res; int res_slabs; long res_slab_stride;   // residual: same shape as out (sum of res_slabs slabs) or null
  int H, W, Cin, Chid, Cout, stride, Ho, Wo, pt, pl;
  int co0, Cv;                     // nv_pblock_kernel: this launch computes output channels co0 .. co0 + Cv - 1 of the Cout (Cv = 0: all of them)
  long P;                          // mode 2: number of pixels in the flat list
  int th, tw;                      // nv_xblock_kernel: output tile (th x tw <= 128 pixels)
  unsigned inv_iw, inv_tw;         // ceil(2^20 / d) for d = patch width, tile width: n / d = (n * inv) >> 20 for n < 1024 (filled in by the launcher)
  int cpg;                         // chunks of 16 hidden channels per workgroup group
  int gmerge;                      // nv_pblock_kernel: consecutive groups ONE workgroup walks (0 / 1: one).  The groups stay the unit of the fp32 summation order: a
                                   // workgroup closes each group's partial sum and adds it to a running total, so slab j of the launch holds (s[j gm] + s[j gm + 1]) + ... --
                                   // exactly what launch_nv_slab_sum(..., tree = gm) makes of the unmerged slabs (a batch merges, one image does not: same bits)
  const float* we;                    // expand weights + bias, one record per chunk (pack_nv_expand)
  const float* wp;                    // depthwise weights + bias + project weights, one record per chunk (pack_nv_dwproj)
  const float* bp;                    // project bias [Cout padded to the n-tiles]
  int act_e, act_d, act_p;
  int ncu;                            // compute units of the device (launch heuristics; 0: 256)
  int tpw;                            // nv_fpair_kernel: consecutive tiles per workgroup (0: the launcher decides)
  int nbuf;                           // nv_pblock_kernel: E buffers (0: the launcher decides)
  unsigned long long* stamps;         // diagnostics (D2FE_NV_STAMP_STEP): [workgroup][32] wall_clock64() phase stamps, or null
};
bool nv_block_supported(int cin, int chid, int cout, int stride, bool expand, int mode);
int nv_block_ntiles(int cout);
hipError_t launch_nv_block(const NvBlockArgs& a, bool expand, int mode, int n, int groups, hipStream_t s);
void pack_nv_expand(const float* w /*[chid][cin]*/, const float* b, int chid, int cin, float* dst);
size_t pack_nv_expand_floats(int chid, int cin);
void pack_nv_dwproj(const float* wd /*[chid][9] or null*/, const float* bd, const float* wp /*[cout][chid]*/, int cout, int chid, int nt, float* dst);
size_t pack_nv_dwproj_floats(int chid, int nt);
void pack_nv_proj_t(const float* wp, int cout, int chid, int nt, float* dst);                 // nv_tail_kernel's project fragments (transposed expand GEMM)
void pack_nv_dwproj_x(const float* wd, const float* bd, const float* wp, int cout, int chid, int nt, float* dst);     // nv_xblock_kernel's layout of the depthwise part
void pack_nv_expand_tail(const float* w, const float* b, int chid, int cin, float* dst);    // K order of nv_tail_kernel
bool nv_tail_supported(int cin, int cout);
hipError_t launch_nv_tail(const NvBlockArgs& a, int groups, hipStream_t s);
// expand blocks with the input in registers and a run-time tile shape (nv_xblock_kernel)
bool nv_xblock_supported(int cin, int chid, int cout, int stride);
void nv_xblock_tile(int Ho, int Wo, int stride, int* th, int* tw);
size_t pack_nv_expand_perm_floats(int chid, int cin);
void pack_nv_expand_perm(const float* w, const float* b, int chid, int cin, float* dst);
hipError_t launch_nv_xblock(const NvBlockArgs& a, int n, int groups, hipStream_t s);
// stride-1 expand blocks, depthwise stage on horizontal pixel pairs (netvlad_pair.hip)
bool nv_pblock_supported(int cin, int chid, int cout, int stride);
void nv_pblock_tile(int Ho, int Wo, int* th, int* tw, int c0_stride = 0);
size_t pack_nv_expand_pair_floats(int chid, int cin);
void pack_nv_expand_pair(const float* w, const float* b, int chid, int cin, float* dst);
size_t pack_nv_dwproj_pair_floats(int chid, int nt);
void pack_nv_dwproj_pair(const float* wd, const float* bd, const float* wp, int cout, int chid, int nt, float* dst, int co0 = 0);     // `cout` output channels from row co0 of wp
int nv_pblock_ntiles(int cout);           // n-tiles of one launch (1..8), -1 beyond 128 channels
int nv_pblock_halves(int cout);           // launches a block of `cout` output channels takes (1, or 2 beyond 128), -1 beyond 256
int nv_pblock_half_cout(int cout, int half);
bool nv_pblock_single_input(int cin);     // the kernel of this input width reads ONE input slab (its producer's partial slabs are summed first)
hipError_t launch_nv_pblock(const NvBlockArgs& a, int n, int groups, hipStream_t s);
bool nv_fpair_supported(int c0_cout, int c0_stride, int dw_stride, int cout);      // the first block (conv from u8 -> dw -> pw) in the same form
hipError_t launch_nv_fpair(const NvBlockArgs& a, int n, hipStream_t s);
long nv_pblock_slots(int cin, int cout, int ncu, int nbuf);     // resident workgroups of that block shape on the whole device
hipError_t launch_nv_slab_sum(float* t, int slabs, long slab_stride, long count, hipStream_t s, int tree = 1);   // slab 0 = sum of the slabs, in slab order; tree > 1: runs of `tree` slabs are summed first
bool nv_pblock_can_merge(int cin, int cv);   // the block shape exists with a running total (NvBlockArgs::gmerge > 1)
void pack_nv_conv0(const float* w /*[cout][9]*/, const float* b, int cout, float* dst /*[384]*/);

// ---- SURVEY 8(f) next rows (next.hip) ------------------------------------------------------------------------------------
hipError_t launch_gen_map(const double* cam9, const double* q4, int mode, int width, int height, double f, float* mapx,
                          float* mapy, hipStream_t s);
hipError_t launch_prep_gray(const uint8_t* src, int ch, int sw, int sh, int sstride, long src_istride, int n, int dw, int dh,
                            uint8_t* dst, hipStream_t s);
hipError_t launch_undistort(const uint8_t* src, int sh, int sw, int sstride, long src_istride, const float* mapx,
                            const float* mapy, const float* gain, int dh, int dw, int n, uint8_t* dst, hipStream_t s);
hipError_t launch_db_search(const float* db, int ntotal, int dim, const float* q, int nq, int k, float* sims_scratch,
                            int32_t* labels, float* out_sims, hipStream_t s);
hipError_t launch_quant_int8(const float* x, int n, int double_max, int8_t* out, hipStream_t s);
hipError_t launch_dequant_int8(const int8_t* q, int n, int landmark_num, float* out, hipStream_t s);

// ---- cross-agent exchange (swarm.hip) -----------------------------------------------------------------------------------------
hipError_t launch_pack_blocks(const float* desc, const float* kps, const float* scores, const int32_t* n_kp, const float* gdesc,
                              int row0, int row_step, int nframes, int cap, int G, int blk_words, float* blocks, hipStream_t s);
hipError_t launch_gate_pairs(const float* q, long q_stride, const float* db, long db_stride, int dim, const int32_t* pair_q,
                             const int32_t* pair_db, int npairs, double thres, int32_t* cnt_inout, int32_t* pass, float* sims,
                             int32_t* n_pass, hipStream_t s);
hipError_t launch_pack_blocks_int8(const float* desc, const float* kps, const int32_t* n_kp, const float* gdesc, int row0, int row_step,
                                   int nframes, int cap, int G, int blk_bytes, int8_t* blocks, hipStream_t s);
hipError_t launch_unpack_blocks_int8(const int8_t* blocks, int nblocks, int cap, int G, int blk_bytes, int blk_words, int renorm, float* out,
                                     hipStream_t s);
hipError_t launch_quad_gate(const float* loc, long loc_stride, const float* rem, long rem_stride, int dim, const int32_t* job_loc_row0,
                            const int32_t* job_rem_row0, int loc_view_step, int rem_view_step, int njobs, double thres, int32_t* dir_prev,
                            float* sims, int32_t* cnt_inout, int32_t* n_pass, hipStream_t s);

hipError_t launch_half_compact(const float* desc, const float* pts, const int32_t* n_kp, const int32_t* job_row, const int32_t* job_left,
                               const float* job_shift, int njobs, int cap, int dim, float width_undistort, float move_cols,
                               float* out_desc, float* out_pts, int32_t* out_map, int32_t* out_n, hipStream_t s);
hipError_t launch_remap_matches(int32_t* q_idx, int32_t* t_idx, const int32_t* n_match, const int32_t* map_a_job, const int32_t* map_b_job,
                                const int32_t* maps, int npairs, int cap_match, int cap_map, hipStream_t s);

// ---- matcher --------------------------------------------------------------------------------------
// optional per-pair view (pipe.hip): absolute device pointers instead of pool + row offset, counts anywhere, a radius per pair
struct MatchPairDesc { const float* a; const float* b; const float* pts_a; const float* pts_b; const int32_t* na; const int32_t* nb; double radius; };
struct MatchArgs {
  const MatchPairDesc* pairs = nullptr;     // [npairs] in device memory; null: the pool form below
  const float* a; const float* b; const float* pts_a; const float* pts_b;
  const int32_t* a_off; const int32_t* b_off; const int32_t* a_cnt; const int32_t* b_cnt;
  int npairs, dim, max_n, mode;
  double ratio, radius;
  int32_t* q_idx; int32_t* t_idx; float* dist; int32_t* n_out;
  // scratch: per pair, per direction, per row one record {nn index, d0 bits, d1 bits, inverse dictionary}
  int32_t* cand4;   // [npairs][2][max_n][4]
  int32_t* ticket;  // [npairs] arrival counters, ZERO before the first launch on this scratch (the kernel leaves them zero)
  int32_t* stats = nullptr;   // optional: [0] += queries whose 2-NN came from the exact scan of every row, [1] += candidates re-ranked beyond two per query (match.hip)
  int ncu = 0;                // compute units of the device (0: 256): launch shape
  unsigned long long* stamps = nullptr;   // development builds (-DD2FE_DEVTOOLS) only: [workgroup][16] wall_clock64() phase stamps
};
hipError_t launch_match(const MatchArgs& m, hipStream_t s);
inline size_t match_scratch_bytes(int npairs, int max_n) { return sizeof(int32_t) * (((size_t)npairs + 15) / 16 * 16 + 8 * (size_t)max_n * npairs); }
// carves the zero-initialised scratch [tickets | records]
inline void match_scratch_carve(void* base, int npairs, MatchArgs* m) { m->ticket = static_cast<int32_t*>(base); m->cand4 = m->ticket + ((size_t)npairs + 15) / 16 * 16; }

}  // namespace d2fe

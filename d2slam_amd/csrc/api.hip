// api.hip -- C ABI (include/d2fe.h) of libd2fe_hip.so: context, device buffers, weight packing and the
// launch sequence of the SuperPoint / matcher kernels.  Host code only; kernels live in conv.hip,
// conv_f16.hip, postproc.hip and match.hip.  There is deliberately NO CPU fallback in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <array>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "context.h"
#ifdef D2FE_DEVTOOLS
#include "../../include/d2fe_debug.h"
#endif

using namespace d2fe;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace

// accessors for the translation units that keep their own extern "C" entry points (lk.hip)
namespace d2fe {
int ctx_fail(int code, const std::string& msg) { return fail(code, msg); }
int ctx_device(d2fe_handle h) { return h->cfg.device_id; }
hipStream_t ctx_stream(d2fe_handle h) { return h->stream; }
// grow-only device scratch owned by the handle: no hipMalloc/hipFree (a device-wide sync) per tracker call.  Not re-entrant,
// like the extract calls: one tracker thread per handle (the reference's D2FeatureTracker is single-threaded, d2frontend.cpp:155-169)
int ctx_scratch(d2fe_handle h, size_t bytes, void** out) {
  if (bytes > h->lk_scratch_bytes) {
    if (h->lk_scratch) { hipStreamSynchronize(h->stream); hipFree(h->lk_scratch); h->lk_scratch = nullptr; h->lk_scratch_bytes = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipMalloc(&h->lk_scratch, want) != hipSuccess) return fail(D2FE_ERR_HIP, "hipMalloc scratch");
    h->lk_scratch_bytes = want;
  }
  *out = h->lk_scratch;
  return D2FE_OK;
}
}  // namespace d2fe

namespace {

int alloc_f(Tensor& t, size_t per_img, int batch) {
  t.per_img = per_img;
  HIP_TRY(hipMalloc(&t.p, per_img * batch * sizeof(float)));
  return D2FE_OK;
}

int upload(const void* src, size_t bytes, void** dst) {
  HIP_TRY(hipMalloc(dst, bytes));
  HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return D2FE_OK;
}

// pack one conv (or a channel-concatenation of convs sharing the input) into the kernel's fragment order
int pack_layer(d2fe_context* h, Layer& L, const std::vector<const d2fe_conv_params*>& parts, int cout_pad, bool force_f32 = false) {
  const int cin = parts[0]->cin, ks = parts[0]->ksize;
  int cout = 0;
  for (auto* p : parts) cout += p->cout;
  std::vector<float> w((size_t)cout * cin * ks * ks), b(cout_pad, 0.f);
  size_t wo = 0;
  int bo = 0;
  for (auto* p : parts) {
    const size_t n = (size_t)p->cout * cin * ks * ks;
    memcpy(w.data() + wo, p->weight, n * sizeof(float));
    memcpy(b.data() + bo, p->bias, p->cout * sizeof(float));
    wo += n;
    bo += p->cout;
  }
  L.cout = cout; L.cout_pad = cout_pad; L.cin = cin; L.ks = ks;
  if (L.wpack) { hipFree(L.wpack); L.wpack = nullptr; }
  if (L.bias) { hipFree(L.bias); L.bias = nullptr; }
  int rc;
  if (h->cfg.precision == D2FE_PREC_F32_WINO && ks == 3 && cin >= 64 && !force_f32) {
    std::vector<float> pk(packed_weight_floats_wino(cout_pad, cin));
    pack_weights_wino(w.data(), cout, cin, cout_pad, pk.data());
    rc = upload(pk.data(), pk.size() * sizeof(float), &L.wpack);
  } else if (h->cfg.precision != D2FE_PREC_F16X2 || force_f32) {
    std::vector<float> pk(packed_weight_floats_f32(cout_pad, cin, ks));
    pack_weights_f32(w.data(), cout, cin, ks, cout_pad, pk.data());
    rc = upload(pk.data(), pk.size() * sizeof(float), &L.wpack);
  } else {
    std::vector<uint16_t> pk(packed_weight_halfs_f16x2(cout_pad, cin, ks));
    pack_weights_f16x2(w.data(), cout, cin, ks, cout_pad, pk.data());
    rc = upload(pk.data(), pk.size() * sizeof(uint16_t), &L.wpack);
  }
  if (rc) return rc;
  return upload(b.data(), b.size() * sizeof(float), reinterpret_cast<void**>(&L.bias));
}

int check_layer(const d2fe_conv_params& p, int cout, int cin, int ks, const char* name) {
  if (!p.weight || !p.bias || p.cout != cout || p.cin != cin || p.ksize != ks)
    return fail(D2FE_ERR_INVALID, std::string("superpoint layer ") + name + ": unexpected shape");
  return D2FE_OK;
}

struct ProfScope {
  d2fe_context* h; int stage; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on = false;
  ProfScope(d2fe_context* h_, int stage_, hipStream_t s_) : h(h_), stage(stage_), s(s_) {
    on = h->prof_mode == 2 || (h->prof_mode == 1 && (stage == D2FE_PROF_CONV1B || stage == D2FE_PROF_NETVLAD));
    if (!on) return;
    if (h->prof_used + 2 > h->prof_pool.size()) {
      for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } h->prof_pool.push_back(e); }
    }
    a = h->prof_pool[h->prof_used++]; b = h->prof_pool[h->prof_used++];
    (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    h->prof_recs.push_back({stage, a, b});
  }
};

// weights / PCA matrices were (re)loaded: the captured launch sequences hold the old device pointers
void graphs_clear(d2fe_context* h) {
  if (h->graphs.empty()) return;
  (void)hipStreamSynchronize(h->stream);
  for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  h->graphs.clear();
}

// Runs `fn(s)` -- a launch sequence on `s` whose arguments are a pure function of `key` (handle-owned buffers only) -- directly the
// first time a key is seen (module loads, function attributes, lazy allocations happen there), captures it into a hipGraph the second
// time and replays the instantiated graph from then on.  Anything that cannot be captured marks the key bad and runs directly.
void host_state_save(const d2fe_context* h, d2fe_context::HostState& st) {
  st.last_w = h->last_w; st.last_h = h->last_h; st.last_n = h->last_n; st.last_set = h->last_set;
  st.last_gray = h->last_gray; st.last_stride = h->last_stride; st.last_istride = h->last_istride;
  st.nv_slabs.clear();
  for (const auto& l : h->nv) st.nv_slabs.emplace_back(l.slabs, l.slab_stride);
  st.nv_feat_slabs = h->nv_feat_slabs; st.nv_feat_slab_stride = h->nv_feat_slab_stride; st.nv_stamp_wgs = h->nv_stamp_wgs;
}
void host_state_restore(d2fe_context* h, const d2fe_context::HostState& st, bool netvlad) {
  if (netvlad) {
    for (size_t i = 0; i < h->nv.size() && i < st.nv_slabs.size(); ++i) { h->nv[i].slabs = st.nv_slabs[i].first; h->nv[i].slab_stride = st.nv_slabs[i].second; }
    h->nv_feat_slabs = st.nv_feat_slabs; h->nv_feat_slab_stride = st.nv_feat_slab_stride; h->nv_stamp_wgs = st.nv_stamp_wgs;
  } else {
    h->last_w = st.last_w; h->last_h = st.last_h; h->last_n = st.last_n; h->last_set = st.last_set;
    h->last_gray = st.last_gray; h->last_stride = st.last_stride; h->last_istride = st.last_istride;
  }
}

template <class F>
int run_cached(d2fe_context* h, const std::array<long, 6>& key, hipStream_t s, F&& fn) {
  if (!h->use_graphs || h->prof_mode != 0) return fn(s);
  auto& e = h->graphs[key];
  if (e.bad) return fn(s);
  const bool netvlad = key[0] != 1;       // key[0]: 1 = the SuperPoint sequence, 2 / 3 = NetVLAD (own frames / the frames SuperPoint reads)
  if (e.exec) { HIP_TRY(hipGraphLaunch(e.exec, s)); host_state_restore(h, e.st, netvlad); return D2FE_OK; }
  if (e.seen++ < 1) return fn(s);
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); e.bad = true; return fn(s); }
  const int rc = fn(s);
  hipGraph_t g = nullptr;
  const hipError_t er = hipStreamEndCapture(s, &g);
  if (rc != D2FE_OK || er != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    e.bad = true;
    return rc != D2FE_OK ? rc : fn(s);
  }
  hipGraphExec_t ex = nullptr;
  const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ei != hipSuccess || !ex) { (void)hipGetLastError(); e.bad = true; return fn(s); }
  e.exec = ex;
  host_state_save(h, e.st);               // fn(s) ran its host code during the capture: this is the state a direct run leaves
  HIP_TRY(hipGraphLaunch(e.exec, s));
  return D2FE_OK;
}

}  // namespace

namespace d2fe {
// the launch sequence == one TensorRT executeV2 + processOutput of the reference
int run_superpoint(d2fe_context* h, const uint8_t* d_gray, int n, int W, int H, int stride, size_t image_stride,
                   float* d_kps, float* d_scores, float* d_desc, int32_t* d_idx, int cap, int32_t* d_n, hipStream_t s,
                   hipStream_t s_tail, int bs) {
  // s: the convolutions ("trunk"); s_tail (default: s): softmax, selection, descriptor head, sampling ("tail"); bs: buffer set
  Tensor& a4b = bs ? h->a4b2 : h->a4b;
  Tensor& logits = bs ? h->logits2 : h->logits;
  Tensor& draw = bs ? h->draw2 : h->draw;
  h->last_set = bs;
  const int prec = h->cfg.precision;
  // One memset per pass: the Winograd work counters and (when the post-processing runs on this same stream) the per-image candidate
  // counters behind them.  async_tail: the previous call's tail may still be reading ITS counts, so the tail zeroes them itself, in stream order.
  const bool tail_elsewhere = s_tail && s_tail != s;
  const bool wctr = prec == D2FE_PREC_F32_WINO && h->wino_dynamic;
  if (wctr || !tail_elsewhere)
    // the cleared range is padded to a multiple of 64 ints (256 B; the allocation is): the runtime splits a memset whose size is not a multiple of its fill width
    // into an aligned fill and a tail fill -- two ~5 us kernels in front of every pass instead of one (kernel trace of a one-frame pass, round 5)
    HIP_TRY(hipMemsetAsync(wctr ? h->work_ctrs : h->cand_count, 0, ((wctr ? 64 : 0) + (tail_elsewhere ? 0 : (n + 63) / 64 * 64)) * sizeof(int), s));
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, Hc = H / 8, Wc = W / 8;
  auto conv = [&](ConvShape shape, const Layer& L, const float* in, int ics, int ico, long iis, float* out, int ocs,
                  long ois, int hh, int ww, bool pool, bool relu, int oco = 0, bool direct = false) -> hipError_t {
    ConvArgs a;
    a.in = in; a.in_cstride = ics; a.in_coff = ico;
    a.out = out; a.out_cstride = ocs; a.out_coff = oco;
    a.cout_real = L.cout; a.wpack = L.wpack; a.bias = L.bias;
    a.img = d_gray; a.img_stride = stride; a.img_istride = (long)image_stride; a.w1a = h->w1a; a.b1a = h->b1a;
    a.H = hh; a.W = ww; a.n_img = n; a.in_img_stride = iis; a.out_img_stride = ois; a.zeros = h->zeros; a.ncu = h->ncu;
    a.tag = (&L == &h->L[L_1B]) ? 1 : 0;
    a.work_ctr = (prec == D2FE_PREC_F32_WINO && h->wino_dynamic) ? h->work_ctrs + (int)(&L - &h->L[0]) : nullptr;
    { static const int ab = d2fe_dev_env("D2FE_ABLATE", 0); a.ablate = ab; }
    if (prec != D2FE_PREC_F16X2 && shape == CONV_256_1x1_T4x16 && L.cout == 65 && !pool && !relu) {      // convPb
      static const int on = d2fe_dev_env("D2FE_CONV1X1", 1);
      if (on) { const hipError_t e = launch_conv1x1_256_65(a, s); if (e != hipErrorNotSupported) return e; }
    }
    if (prec == D2FE_PREC_F32_WINO)   // 3x3 layers: Winograd kernels; the 1x1 heads (and `direct`: convDa of the dense head): the exact fp32 kernels
      return shape == CONV1B_FUSED ? launch_conv_wino_fused1b(L.cout_pad, a, s)
             : (L.ks == 3 && !direct) ? launch_conv_wino(L.cin, pool, relu, L.cout_pad, a, s) : launch_conv(shape, D2FE_PREC_F32, pool, relu, L.cout_pad, a, s);
    return launch_conv(shape, prec, pool, relu, L.cout_pad, a, s);
  };
  if (h->fuse1a) {
    ProfScope ps(h, D2FE_PROF_CONV1B, s);
    HIP_TRY(conv(CONV1B_FUSED, h->L[L_1B], nullptr, 64, 0, 0, h->a1b.p, 64, (long)H2 * W2 * 64, H, W, true, true));
  } else {
    { ProfScope ps(h, D2FE_PROF_CONV1A, s); HIP_TRY(launch_conv1a(d_gray, stride, (long)image_stride, H, W, n, h->w1a, h->b1a, h->a1a.p, s)); }
    { ProfScope ps(h, D2FE_PROF_CONV1B, s); HIP_TRY(conv(CONV_64_T8x32, h->L[L_1B], h->a1a.p, 64, 0, (long)H * W * 64, h->a1b.p, 64, (long)H2 * W2 * 64, H, W, true, true)); }
  }
  { ProfScope ps(h, D2FE_PROF_CONV2A, s); HIP_TRY(conv(CONV_64_T8x32, h->L[L_2A], h->a1b.p, 64, 0, (long)H2 * W2 * 64, h->a2a.p, 64, (long)H2 * W2 * 64, H2, W2, false, true)); }
  { ProfScope ps(h, D2FE_PROF_CONV2B, s); HIP_TRY(conv(CONV_64_T8x32, h->L[L_2B], h->a2a.p, 64, 0, (long)H2 * W2 * 64, h->a2b.p, 64, (long)H4 * W4 * 64, H2, W2, true, true)); }
  { ProfScope ps(h, D2FE_PROF_CONV3A, s); HIP_TRY(conv(CONV_64_T8x32, h->L[L_3A], h->a2b.p, 64, 0, (long)H4 * W4 * 64, h->a3a.p, 128, (long)H4 * W4 * 128, H4, W4, false, true)); }
  { ProfScope ps(h, D2FE_PROF_CONV3B, s); HIP_TRY(conv(CONV_128_T4x32, h->L[L_3B], h->a3a.p, 128, 0, (long)H4 * W4 * 128, h->a3b.p, 128, (long)Hc * Wc * 128, H4, W4, true, true)); }
  { ProfScope ps(h, D2FE_PROF_CONV4A, s); HIP_TRY(conv(CONV_128_T4x16, h->L[L_4A], h->a3b.p, 128, 0, (long)Hc * Wc * 128, h->a4a.p, 128, (long)Hc * Wc * 128, Hc, Wc, false, true)); }
  { ProfScope ps(h, D2FE_PROF_CONV4B, s); HIP_TRY(conv(CONV_128_T4x16, h->L[L_4B], h->a4a.p, 128, 0, (long)Hc * Wc * 128, a4b.p, 128, (long)Hc * Wc * 128, Hc, Wc, false, true)); }
  // both paths give identical bits; with fewer than 4 images per call the dense head is quicker (the sparse kernels are
  // latency-bound with so few 32-cell workgroups in flight; measured per step, exact mode: 2 images 1.079 ms dense / 1.096 sparse,
  // 4 images 1.877 / 1.862, 8 images 3.42 / 3.28, 16 images 6.48 / 6.20)
  const bool sparse = h->sparse_desc && n >= h->sp_min_batch && 4 * (long)cap <= h->sp_slots;      // a larger capacity than the sparse store was sized for: dense head
  if (sparse) {
    // detector head only (convPa 128->256, convPb); the descriptor head is evaluated after keypoint selection, at the needed cells
    { ProfScope ps(h, D2FE_PROF_CONVPADA, s); HIP_TRY(conv(CONV_128_T4x16, h->L[L_PA], a4b.p, 128, 0, (long)Hc * Wc * 128, h->aPD.p, 256, (long)Hc * Wc * 256, Hc, Wc, false, true)); }
    { ProfScope ps(h, D2FE_PROF_CONVPB, s); HIP_TRY(conv(CONV_256_1x1_T4x16, h->L[L_PB], h->aPD.p, 256, 0, (long)Hc * Wc * 256, logits.p, 65, (long)Hc * Wc * 65, Hc, Wc, false, false)); }
  } else {
    if (prec == D2FE_PREC_F32_WINO && h->sparse_desc) {
      // Winograd mode, dense head (calls with fewer images than the sparse head pays for): the detector branch convPa as a Winograd layer, the
      // descriptor branch convDa as DIRECT fp32 chains -- the arithmetic of the sparse head -- so that a frame's descriptors are the same bits
      // whichever head a call takes (1-image call, 64-image batch, any pass of the frames-in-flight pipe)
      ProfScope ps(h, D2FE_PROF_CONVPADA, s);
      HIP_TRY(conv(CONV_128_T4x16, h->L[L_PA], a4b.p, 128, 0, (long)Hc * Wc * 128, h->aPD.p, 512, (long)Hc * Wc * 512, Hc, Wc, false, true));
      HIP_TRY(conv(CONV_128_T4x16, h->L[L_DA32], a4b.p, 128, 0, (long)Hc * Wc * 128, h->aPD.p, 512, (long)Hc * Wc * 512, Hc, Wc, false, true, 256, true));
    } else
    { ProfScope ps(h, D2FE_PROF_CONVPADA, s); HIP_TRY(conv(CONV_128_T4x16, h->L[L_PADA], a4b.p, 128, 0, (long)Hc * Wc * 128, h->aPD.p, 512, (long)Hc * Wc * 512, Hc, Wc, false, true)); }
    { ProfScope ps(h, D2FE_PROF_CONVPB, s); HIP_TRY(conv(CONV_256_1x1_T4x16, h->L[L_PB], h->aPD.p, 512, 0, (long)Hc * Wc * 512, logits.p, 65, (long)Hc * Wc * 65, Hc, Wc, false, false)); }
    { ProfScope ps(h, D2FE_PROF_CONVDB, s); HIP_TRY(conv(CONV_256_1x1_T4x16, h->L[L_DB], h->aPD.p, 512, 256, (long)Hc * Wc * 512, draw.p, 256, (long)Hc * Wc * 256, Hc, Wc, false, false)); }
  }
  const bool varA = h->cfg.postproc == D2FE_POSTPROC_A;
  if (s_tail && s_tail != s) {       // trunk done -> tail may start; from here on everything is issued on the tail stream
    HIP_TRY(hipEventRecord(h->ev_trunk[bs], s));
    HIP_TRY(hipStreamWaitEvent(s_tail, h->ev_trunk[bs], 0));
    s = s_tail;
  }
  { ProfScope ps(h, D2FE_PROF_SOFTMAX, s);
  HIP_TRY(launch_softmax_cand(logits.p, 65, Hc, Wc, n, h->cfg.keypoint_threshold, h->cfg.remove_borders,
                              (h->cfg.keep_score_map || varA || h->cfg.max_keypoints < 0) ? h->semi.p : nullptr,
                              h->cand, h->cand_count, varA ? 0 : h->cand_cap, /*zero_counts=*/tail_elsewhere, s)); }
  if (varA) {
    // getKeyPoints + NMS2 (superpoint_common.cpp:12-40,107-177): border = 0, sorted by confidence, max_num
    ProfScope ps(h, D2FE_PROF_SELECT, s);
    HIP_TRY(launch_nms2_a(h->semi.p, H, W, n, h->cfg.keypoint_threshold, h->cfg.nms_dist, h->aconf, h->clist, h->cand,
                          h->cand_count, h->cand_cap, h->a_ncand, s));
    // variant A's sampling scratch holds a_scap keypoints per image: keep-all (max_keypoints = -1) means "up to a_scap" here
    HIP_TRY(launch_select_b(h->cand, h->cand_count, h->cand_cap, n, W, h->cfg.max_keypoints < 0 ? h->a_scap : h->cfg.max_keypoints, cap, 1, nullptr, H,
                            h->cfg.keypoint_threshold, 0, d_kps, d_scores, d_idx, d_n, s));
    if ((long)H * W > 65536)     // the reference's CV_16UC1 index map wraps above 65 536 candidates; reproduced (no-op below that)
      HIP_TRY(launch_nms2_wrap_fix(h->clist, h->a_ncand, H, W, n, d_kps, d_idx, d_n, cap, s));
  } else {
    ProfScope ps(h, D2FE_PROF_SELECT, s);
    // raster indices and keypoints are in score-map coordinates: (W/8)*8 wide.  Keep-all handles also hand over the dense score map:
    // more keypoints than the in-LDS sort takes are compacted from it in raster order (any count up to the call's capacity)
    HIP_TRY(launch_select_b(h->cand, h->cand_count, h->cand_cap, n, Wc * 8, h->cfg.max_keypoints, cap, 0,
                            (h->cfg.keep_score_map || h->cfg.max_keypoints < 0) ? h->semi.p : nullptr, Hc * 8, h->cfg.keypoint_threshold,
                            h->cfg.remove_borders, d_kps, d_scores, d_idx, d_n, s));
  }
  if (!sparse) {
    ProfScope ps(h, D2FE_PROF_SAMPLE, s);
    if (varA)
      HIP_TRY(launch_sample_a(draw.p, 256, 0, Hc, Wc, W, H, n, d_kps, d_n, cap, h->pca_dims ? h->pca_comp_t : nullptr,
                              h->pca_mean, h->pca_dims, h->a_samp, h->a_scap, h->a_cn, nullptr, 0, d_desc, s));
    else
      HIP_TRY(launch_sample_b(draw.p, 256, 0, Hc, Wc, n, d_kps, d_n, cap, nullptr, 0, d_desc, s));
  } else {
    { ProfScope ps(h, D2FE_PROF_CONVDB, s);
      HIP_TRY(launch_desc_head_sparse(d_kps, d_n, cap, Hc, Wc, n, a4b.p, 128, (long)Hc * Wc * 128, h->L[L_DA32].wpack, h->L[L_DA32].bias,
                                      h->L[L_DB32].wpack, h->L[L_DB32].bias, h->sp_flags, h->sp_slotmap, h->sp_cells, h->sp_count,
                                      h->sp_slots, h->sp_desc, h->sp_mid, h->sp_mid_imgs, varA ? W : 0, varA ? H : 0, s)); }
    ProfScope ps(h, D2FE_PROF_SAMPLE, s);
    if (varA)
      HIP_TRY(launch_sample_a(h->sp_desc, 256, 0, Hc, Wc, W, H, n, d_kps, d_n, cap, h->pca_dims ? h->pca_comp_t : nullptr,
                              h->pca_mean, h->pca_dims, h->a_samp, h->a_scap, h->a_cn, h->sp_slotmap, h->sp_slots, d_desc, s));
    else
      HIP_TRY(launch_sample_b(h->sp_desc, 256, 0, Hc, Wc, n, d_kps, d_n, cap, h->sp_slotmap, h->sp_slots, d_desc, s));
  }
  h->last_w = W; h->last_h = H; h->last_n = n;
  h->last_gray = d_gray; h->last_stride = stride; h->last_istride = image_stride;
  return D2FE_OK;
}

int check_geometry(d2fe_context* h, int n, int W, int H, int stride, int cap) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  if (!h->sp_loaded) return fail(D2FE_ERR_NOT_READY, "superpoint weights not loaded");
  if (n < 1 || n > h->cfg.max_batch) return fail(D2FE_ERR_INVALID, "batch size out of range");
  if (W < 16 || H < 16 || W > h->cfg.max_width || H > h->cfg.max_height || (long)W * H > (long)h->cfg.max_width * h->cfg.max_height)
    return fail(D2FE_ERR_INVALID, "image size must be at least 16x16 and within the configured maximum");
  // Sizes that are not multiples of 8 (the reference's TensorRT profile admits 100x100 .. 1500x1500, superpoint_tensorrt.cpp:50-55): the
  // three 2x2 max-pools floor, the score map is (H/8)*8 x (W/8)*8 and keypoints live in ITS coordinates, as processOutput reads
  // semi_dims_ (:331-336).  Variant A sizes its cv::Mat views from the configured width/height (superpoint_onnx.cpp:86-94), which only
  // describes the network output for multiples of 8.
  if (((W | H) & 7) && h->cfg.postproc == D2FE_POSTPROC_A)
    return fail(D2FE_ERR_INVALID, "post-processing variant A needs image sizes that are multiples of 8");
  if (stride < W) return fail(D2FE_ERR_INVALID, "stride < width");
  if (cap < 1) return fail(D2FE_ERR_INVALID, "cap < 1");
  return D2FE_OK;
}

}  // namespace d2fe

extern "C" {

const char* d2fe_last_error(void) { return g_err.c_str(); }
const char* d2fe_version(void) { return "d2fe-hip 0.1 (gfx950)"; }

void d2fe_default_config(d2fe_config* c) {
  memset(c, 0, sizeof(*c));
  c->struct_size = (int32_t)sizeof(d2fe_config);
  c->device_id = 0;
  c->max_width = 640;
  c->max_height = 480;
  c->max_batch = 2;
  c->max_keypoints = 100;       // SuperPointConfig default, superpoint_tensorrt.h:18
  c->remove_borders = 1;        // :19
  c->keypoint_threshold = 0.015f;  // :24
  c->postproc = D2FE_POSTPROC_B;
  c->nms_dist = 10;
  c->precision = D2FE_PREC_F32;
  c->dense_descriptors = 0;
  c->async_tail = 0;
}

}  // extern "C"
namespace d2fe {
// lane = a context cloned for a pipe (clone_lane): no host-pointer staging (frame upload buffer, output block, pinned mirrors) -- a lane is only ever
// driven through run_superpoint / run_netvlad on device buffers of the pipe
static int create_context(const d2fe_config* cfg, d2fe_handle* out, bool lane, hipStream_t adopt = nullptr) {
  if (!cfg || !out) return fail(D2FE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(d2fe_config)) return fail(D2FE_ERR_INVALID, "d2fe_config size mismatch");
  if (!(cfg->keypoint_threshold >= 0.f)) return fail(D2FE_ERR_INVALID, "keypoint_threshold must be >= 0 (scores are probabilities)");
  if (cfg->max_width < 16 || cfg->max_height < 16) return fail(D2FE_ERR_INVALID, "max_width/max_height must be at least 16");
  if (cfg->max_batch < 1) return fail(D2FE_ERR_INVALID, "max_batch < 1");
  // -1 = keep every keypoint above the threshold (SuperPoint::topKeypoints only truncates when k != -1, superpoint_tensorrt.cpp:241-253;
  // NMS2's `i < max_num` is an unsigned compare, superpoint_common.cpp:173): variant B then returns EVERY keypoint above the threshold, in
  // raster order, up to the call's capacity (any capacity up to H*W; D2FE_ERR_TRUNCATED with the strongest kept if an image had more);
  // variant A up to 1024.  A sorted top-K selection takes K <= 16384 (one workgroup's in-LDS bitonic sort); the reference's TensorRT
  // profile (1500 x 1500, superpoint_tensorrt.cpp:50-55) with max_keypoints in the thousands is inside that.
  if ((cfg->max_keypoints < 1 && cfg->max_keypoints != -1) || cfg->max_keypoints > 16384) return fail(D2FE_ERR_INVALID, "max_keypoints must be -1 (keep all) or in 1..16384");
  if (cfg->precision != D2FE_PREC_F32 && cfg->precision != D2FE_PREC_F16X2 && cfg->precision != D2FE_PREC_F32_WINO) return fail(D2FE_ERR_INVALID, "bad precision");
  if (cfg->postproc != D2FE_POSTPROC_B && cfg->postproc != D2FE_POSTPROC_A) return fail(D2FE_ERR_INVALID, "bad postproc");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(D2FE_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(D2FE_ERR_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(cfg->device_id));
  d2fe_context* h = new d2fe_context();
  h->cfg = *cfg;
  const int rc_alloc = [&]() -> int {
  // conv1a is evaluated inside conv1b's staging in every mode (the Winograd kernel runs it on the matrix pipe): the 78.6 MB/image
  // activation never exists.  D2FE_FUSE1A=0 falls back to a stand-alone conv1a kernel (bit-identical; kept for A/B measurements).
  h->fuse1a = d2fe_dev_env("D2FE_FUSE1A", 1) != 0;
    // (a lane that is handed its stream does not create one first: no stream is created that is not needed, see place_streams in pipe.hip)
    if (adopt) h->stream = adopt; else HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t H = cfg->max_height, W = cfg->max_width;
    const int B = cfg->max_batch;
    int rc = 0;
    if (!h->fuse1a) rc |= alloc_f(h->a1a, H * W * 64, B);   // the fused path never materialises conv1a (78.6 MB per image); debug reads allocate it on demand
    rc |= alloc_f(h->a1b, H * W * 16, B);
    rc |= alloc_f(h->a2a, H * W * 16, B);
    rc |= alloc_f(h->a2b, H * W * 4, B);
    rc |= alloc_f(h->a3a, H * W * 8, B);
    rc |= alloc_f(h->a3b, H * W * 2, B);
    rc |= alloc_f(h->a4a, H * W * 2, B);
    rc |= alloc_f(h->a4b, H * W * 2, B);
    rc |= alloc_f(h->aPD, H * W * 8, B);
    rc |= alloc_f(h->logits, (H / 8) * (W / 8) * 65, B);
    rc |= alloc_f(h->draw, H * W * 4, B);
    rc |= alloc_f(h->semi, H * W, B);
    if (rc) return D2FE_ERR_HIP;
    h->cand_cap = (long)(H * W);
    HIP_TRY(hipMalloc(&h->cand, sizeof(unsigned long long) * h->cand_cap * B));
    // the per-image candidate counters sit directly behind the 64 Winograd work counters: ONE memset clears both at the start of a pass
    HIP_TRY(hipMalloc(&h->work_ctrs, (64 + ((size_t)B + 63) / 64 * 64) * sizeof(int)));      // counts padded to 64 ints: see the memset of a pass (run_superpoint)
    HIP_TRY(hipMemset(h->work_ctrs, 0, (64 + ((size_t)B + 63) / 64 * 64) * sizeof(int)));
    h->cand_count = h->work_ctrs + 64;
    if (cfg->async_tail) {
      HIP_TRY(hipStreamCreateWithFlags(&h->tail_stream, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_trunk[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_tail[i], hipEventDisableTiming));
      }
      if (alloc_f(h->a4b2, H * W * 2, B) || alloc_f(h->logits2, (H / 8) * (W / 8) * 65, B) || alloc_f(h->draw2, H * W * 4, B)) return D2FE_ERR_HIP;
    }
    h->sparse_desc = !cfg->dense_descriptors;
    // Winograd mode: both heads evaluate the descriptor branch as direct fp32 chains (run_superpoint), so the choice is free of consequences for the
    // bits; measured per call (host to host, 640x480): 1 image 0.44 ms dense / 0.47 sparse, 2 images 0.69 dense / 0.67 sparse
    if (cfg->precision == D2FE_PREC_F32_WINO) h->sp_min_batch = 2;
    h->sp_min_batch = d2fe_dev_env("D2FE_SPARSE_MIN_BATCH", h->sp_min_batch);
    if (h->sparse_desc) {
      const size_t ncell = (H / 8) * (W / 8);
      h->sp_slots = 4 * (cfg->max_keypoints < 0 || cfg->max_keypoints > 1024 ? 1024 : cfg->max_keypoints);   // <= 4 corner cells per keypoint; larger calls take the dense head
      HIP_TRY(hipMalloc(&h->sp_flags, ncell * B));
      HIP_TRY(hipMalloc(&h->sp_slotmap, sizeof(int32_t) * ncell * B));
      HIP_TRY(hipMalloc(&h->sp_cells, sizeof(int32_t) * (size_t)h->sp_slots * B));
      HIP_TRY(hipMalloc(&h->sp_count, sizeof(int32_t) * B));
      HIP_TRY(hipMalloc(&h->sp_desc, sizeof(float) * 256 * (size_t)h->sp_slots * B));
      h->sp_mid_imgs = B < 4 ? B : 4;
      HIP_TRY(hipMalloc(&h->sp_mid, sizeof(float) * 256 * (size_t)h->sp_slots * h->sp_mid_imgs));
    }
    HIP_TRY(hipMalloc(&h->zeros, 1024));
    HIP_TRY(hipMemset(h->zeros, 0, 1024));
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device_id) == hipSuccess && v > 0) h->ncu = h->ncu_dev = v; }
    HIP_TRY(hipMalloc(&h->match_stats, 4 * sizeof(int32_t)));
    HIP_TRY(hipMemset(h->match_stats, 0, 4 * sizeof(int32_t)));
    h->wino_dynamic = d2fe_dev_env("D2FE_WINO_DYNAMIC", 1) != 0;
    if (cfg->postproc == D2FE_POSTPROC_A) {
      HIP_TRY(hipMalloc(&h->aconf, sizeof(float) * H * W * B));
      HIP_TRY(hipMalloc(&h->clist, sizeof(int) * H * W * B));
      HIP_TRY(hipMalloc(&h->a_ncand, sizeof(int) * B));
      h->a_scap = h->cfg.max_keypoints > 0 ? h->cfg.max_keypoints : 1024;   // variant A: select keeps at most min(max_keypoints, call capacity); keep-all: 1024
      HIP_TRY(hipMalloc(&h->a_samp, sizeof(float) * 256 * (size_t)h->a_scap * B));
      HIP_TRY(hipMalloc(&h->a_cn, sizeof(float) * 256 * B));
    }
    h->use_graphs = d2fe_dev_env("D2FE_GRAPH", 1) != 0;
    if (lane) { h->use_pinned = false; return D2FE_OK; }
    h->s_cap = h->cfg.max_keypoints > 1024 ? h->cfg.max_keypoints : 1024;      // initial staging capacity per image; grows with the calls (ensure_staging)
    HIP_TRY(hipMalloc(&h->s_img, H * W * B));
    h->s_out_bytes = (sizeof(float) * (size_t)h->s_cap * 260 + sizeof(int32_t)) * B;
    HIP_TRY(hipMalloc(&h->s_out, h->s_out_bytes));
    h->use_pinned = d2fe_dev_env("D2FE_PINNED", 1) != 0;
    if (h->use_pinned) {
      h->pin_in_bytes = (size_t)H * W * B;
      h->pin_out_bytes = h->s_out_bytes;
      if (h->pin_out_bytes < sizeof(float) * 8192 * (size_t)B) h->pin_out_bytes = sizeof(float) * 8192 * (size_t)B;      // NetVLAD descriptors
      if (hipHostMalloc(&h->pin_in, h->pin_in_bytes, hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(&h->pin_out, h->pin_out_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (h->pin_in) { (void)hipHostFree(h->pin_in); h->pin_in = nullptr; }
        h->pin_out = nullptr; h->use_pinned = false;        // not fatal: the pageable path remains
      }
    }
    return D2FE_OK;
  }();
  // (an adopted stream stays the caller's on failure)
  if (rc_alloc != D2FE_OK) { if (adopt) h->stream = nullptr; d2fe_destroy(h); return rc_alloc; }   // everything allocated so far is released
  // the hipMemsets above ran on the null stream; the handle's work runs on non-blocking streams that do not wait for it
  if (hipDeviceSynchronize() != hipSuccess) { if (adopt) h->stream = nullptr; d2fe_destroy(h); return fail(D2FE_ERR_HIP, "hipDeviceSynchronize"); }
  *out = h;
  return D2FE_OK;
}
}  // namespace d2fe
extern "C" {
int d2fe_create(const d2fe_config* cfg, d2fe_handle* out) { return d2fe::create_context(cfg, out, false); }

}  // extern "C"
namespace { void nv_free(d2fe_context* h); }
extern "C" {

void d2fe_destroy(d2fe_handle h) {
  if (!h) return;
  // pipes created from this handle run on ITS packed weights: destroying it under them would leave every lane with dangling pointers.  The destruction is
  // DEFERRED: the handle is marked and the last d2fe_pipe_destroy releases it (a silent refusal leaked the handle when a wrapper destroyed in the
  // wrong order).  Until then the handle must not be used for anything else; d2fe_last_error records what happened
  if (h->live_pipes.load() > 0) { h->doomed.store(true); fail(D2FE_ERR_INVALID, "d2fe_destroy: the handle still has live pipes; it will be released when the last of them is destroyed"); return; }
  hipSetDevice(h->cfg.device_id);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->tail_stream) { hipStreamSynchronize(h->tail_stream); (void)hipStreamDestroy(h->tail_stream); }
  for (int i = 0; i < 2; ++i) { if (h->ev_trunk[i]) (void)hipEventDestroy(h->ev_trunk[i]); if (h->ev_tail[i]) (void)hipEventDestroy(h->ev_tail[i]); }
  for (Tensor* t : {&h->a1a, &h->a1b, &h->a2a, &h->a2b, &h->a3a, &h->a3b, &h->a4a, &h->a4b, &h->aPD, &h->logits, &h->draw, &h->semi, &h->a4b2, &h->logits2, &h->draw2})
    if (t->p) hipFree(t->p);
  if (!h->borrowed) {
    for (auto& L : h->L) { if (L.wpack) hipFree(L.wpack); if (L.bias) hipFree(L.bias); }
    for (void* p : {(void*)h->w1a, (void*)h->b1a, (void*)h->pca_comp_t, (void*)h->pca_mean, (void*)h->nv_pca_comp, (void*)h->nv_pca_mean}) if (p) hipFree(p);
  }
  for (void* p : {(void*)h->cand, (void*)h->s_img, (void*)h->aconf, (void*)h->clist, (void*)h->a_ncand, (void*)h->a_samp, (void*)h->a_cn, h->lk_scratch, (void*)h->zeros, (void*)h->work_ctrs, (void*)h->match_stats, (void*)h->match_stamps, (void*)h->sp_flags, (void*)h->sp_slotmap, (void*)h->sp_cells, (void*)h->sp_count, (void*)h->sp_desc, (void*)h->sp_mid})
    if (p) hipFree(p);
  nv_free(h);
  for (void* p : {(void*)h->nv_s_img, (void*)h->nv_s_out}) if (p) hipFree(p);
  for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  if (h->nv_stream) { hipStreamSynchronize(h->nv_stream); (void)hipStreamDestroy(h->nv_stream); }
  if (h->ev_up) (void)hipEventDestroy(h->ev_up);
  if (h->pin_nv) (void)hipHostFree(h->pin_nv);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  if (h->s_out) hipFree(h->s_out);
  for (auto& sc : h->m_scratch) if (sc.cand4) hipFree(sc.cand4);
  for (auto& ms : h->match_slots) { if (ms.stream) { hipStreamSynchronize(ms.stream); (void)hipStreamDestroy(ms.stream); } if (ms.buf) hipFree(ms.buf); if (ms.pin) (void)hipHostFree(ms.pin); }
  for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int d2fe_load_superpoint(d2fe_handle h, const d2fe_superpoint_weights* w) {
  if (h && h->live_pipes.load() > 0) return fail(D2FE_ERR_INVALID, "the handle has live pipes whose lanes read its packed weights: destroy them before loading weights or PCA matrices");
  if (h) graphs_clear(h);
  if (!h || !w) return fail(D2FE_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  static const struct { const char* name; int cout, cin, ks; } spec[D2FE_SP_NUM_LAYERS] = {
      {"conv1a", 64, 1, 3},    {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},   {"conv2b", 64, 64, 3},
      {"conv3a", 128, 64, 3},  {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3}, {"conv4b", 128, 128, 3},
      {"convPa", 256, 128, 3}, {"convPb", 65, 256, 1},  {"convDa", 256, 128, 3}, {"convDb", 256, 256, 1}};
  for (int i = 0; i < D2FE_SP_NUM_LAYERS; ++i) {
    int rc = check_layer(w->layer[i], spec[i].cout, spec[i].cin, spec[i].ks, spec[i].name);
    if (rc) return rc;
  }
  h->sp_loaded = false;
  // conv1a: [64][1][3][3] -> [9][64]
  {
    std::vector<float> t(9 * 64);
    for (int co = 0; co < 64; ++co)
      for (int k = 0; k < 9; ++k) t[k * 64 + co] = w->layer[0].weight[co * 9 + k];
    if (h->w1a) { hipFree(h->w1a); h->w1a = nullptr; }
    if (h->b1a) { hipFree(h->b1a); h->b1a = nullptr; }
    int rc = upload(t.data(), t.size() * sizeof(float), reinterpret_cast<void**>(&h->w1a));
    if (rc) return rc;
    rc = upload(w->layer[0].bias, 64 * sizeof(float), reinterpret_cast<void**>(&h->b1a));
    if (rc) return rc;
  }
  const d2fe_conv_params* l = w->layer;
  int rc = 0;
  rc = rc ? rc : pack_layer(h, h->L[L_1B], {&l[1]}, 64);
  rc = rc ? rc : pack_layer(h, h->L[L_2A], {&l[2]}, 64);
  rc = rc ? rc : pack_layer(h, h->L[L_2B], {&l[3]}, 64);
  rc = rc ? rc : pack_layer(h, h->L[L_3A], {&l[4]}, 128);
  rc = rc ? rc : pack_layer(h, h->L[L_3B], {&l[5]}, 128);
  rc = rc ? rc : pack_layer(h, h->L[L_4A], {&l[6]}, 128);
  rc = rc ? rc : pack_layer(h, h->L[L_4B], {&l[7]}, 128);
  rc = rc ? rc : pack_layer(h, h->L[L_PADA], {&l[8], &l[10]}, 512);  // convPa | convDa share their input
  rc = rc ? rc : pack_layer(h, h->L[L_PB], {&l[9]}, 128);
  rc = rc ? rc : pack_layer(h, h->L[L_DB], {&l[11]}, 256);
  if (h->sparse_desc) {   // detector head alone in the mode's precision; descriptor head always as exact fp32 fragments
    rc = rc ? rc : pack_layer(h, h->L[L_PA], {&l[8]}, 256);
    rc = rc ? rc : pack_layer(h, h->L[L_DA32], {&l[10]}, 256, true);
    rc = rc ? rc : pack_layer(h, h->L[L_DB32], {&l[11]}, 256, true);
  }
  if (rc) return rc;
  h->sp_loaded = true;
  return D2FE_OK;
}

int d2fe_set_superpoint_pca(d2fe_handle h, const float* comp, const float* mean, int pca_dims) {
  if (h && h->live_pipes.load() > 0) return fail(D2FE_ERR_INVALID, "the handle has live pipes whose lanes read its packed weights: destroy them before loading weights or PCA matrices");
  if (h) graphs_clear(h);
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  if (pca_dims < 0 || pca_dims > 256 || (pca_dims > 0 && (!comp || !mean))) return fail(D2FE_ERR_INVALID, "bad PCA arguments");
  if (pca_dims > 0 && h->cfg.postproc != D2FE_POSTPROC_A)
    return fail(D2FE_ERR_UNSUPPORTED, "PCA is only applied by post-processing variant A (the reference's variant B never applies it)");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->pca_comp_t) { hipFree(h->pca_comp_t); h->pca_comp_t = nullptr; }
  if (h->pca_mean) { hipFree(h->pca_mean); h->pca_mean = nullptr; }
  h->pca_dims = 0;
  if (pca_dims == 0) return D2FE_OK;
  std::vector<float> t((size_t)256 * pca_dims);
  for (int j = 0; j < pca_dims; ++j)
    for (int c = 0; c < 256; ++c) t[(size_t)c * pca_dims + j] = comp[(size_t)j * 256 + c];
  int rc = upload(t.data(), t.size() * sizeof(float), reinterpret_cast<void**>(&h->pca_comp_t));
  if (rc) return rc;
  rc = upload(mean, 256 * sizeof(float), reinterpret_cast<void**>(&h->pca_mean));
  if (rc) return rc;
  h->pca_dims = pca_dims;
  return D2FE_OK;
}

int d2fe_desc_dim(d2fe_handle h) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  return (h->cfg.postproc == D2FE_POSTPROC_A && h->pca_dims) ? h->pca_dims : 256;
}

int d2fe_superpoint_extract_device(d2fe_handle h, const uint8_t* d_gray, int n, int width, int height, int stride,
                                   size_t image_stride, float* d_kps_xy, float* d_scores, float* d_desc,
                                   int32_t* d_kps_idx, int cap, int32_t* d_n_out, void* stream) {
  int rc = check_geometry(h, n, width, height, stride, cap);
  if (rc) return rc;
  if (!d_gray || !d_kps_xy || !d_scores || !d_desc || !d_n_out) return fail(D2FE_ERR_INVALID, "null device pointer");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (!h->cfg.async_tail)
    return run_superpoint(h, d_gray, n, width, height, stride, image_stride, d_kps_xy, d_scores, d_desc, d_kps_idx, cap,
                          d_n_out, s);
  // async_tail: convolutions on `s`, post-processing on the handle's tail stream.  Buffer set p was last read by the tail of
  // the call before the previous one: the trunk may only overwrite it once that tail is done.
  const int p = h->parity;
  HIP_TRY(hipStreamWaitEvent(s, h->ev_tail[p], 0));       // never-recorded event: no-op
  rc = run_superpoint(h, d_gray, n, width, height, stride, image_stride, d_kps_xy, d_scores, d_desc, d_kps_idx, cap, d_n_out, s,
                      h->tail_stream, p);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->ev_tail[p], h->tail_stream));
  h->parity = p ^ 1;
  return D2FE_OK;
}

void* d2fe_tail_stream(d2fe_handle h) { return h ? (void*)h->tail_stream : nullptr; }

int d2fe_superpoint_wait_tail(d2fe_handle h, void* stream) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  if (!h->cfg.async_tail) return D2FE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipStreamWaitEvent(stream ? (hipStream_t)stream : h->stream, h->ev_tail[h->parity ^ 1], 0));
  return D2FE_OK;
}

// host image(s) -> the handle's device staging, tight rows: through the pinned staging buffer (CPU row copy + ONE DMA) or, when that
// is off, with pageable 2D copies
static int upload_frames(d2fe_context* h, uint8_t* d_dst, const uint8_t* gray, int n, int width, int height, int stride, size_t image_stride,
                         hipStream_t s) {
  const size_t bytes = (size_t)width * height * n;
  if (h->use_pinned && bytes <= h->pin_in_bytes) {
    HIP_TRY(hipStreamSynchronize(s));            // the previous call's DMA out of pin_in has completed (calls are synchronous: a no-op)
    for (int i = 0; i < n; ++i) {
      const uint8_t* src = gray + i * image_stride;
      uint8_t* dst = h->pin_in + (size_t)i * width * height;
      if (stride == width) memcpy(dst, src, (size_t)width * height);
      else for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * width, src + (size_t)y * stride, width);
    }
    HIP_TRY(hipMemcpyAsync(d_dst, h->pin_in, bytes, hipMemcpyHostToDevice, s));
    return D2FE_OK;
  }
  if (image_stride == (size_t)stride * height) {
    HIP_TRY(hipMemcpy2DAsync(d_dst, width, gray, stride, width, (size_t)height * n, hipMemcpyHostToDevice, s));
  } else {
    for (int i = 0; i < n; ++i)
      HIP_TRY(hipMemcpy2DAsync(d_dst + (size_t)i * width * height, width, gray + i * image_stride, stride, width, height,
                               hipMemcpyHostToDevice, s));
  }
  return D2FE_OK;
}


// host-pointer extract of n images; n_netvlad > 0: the NetVLAD descriptors of the first n_netvlad of them as well, from the SAME uploaded frames,
// on a second stream beside SuperPoint (d2fe_extract_all*)
static int extract_host(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride, size_t image_stride, float* kps_xy,
                        float* scores, float* desc, int cap, int* n_out, int n_netvlad, float* netvlad_out) {
  if (n_out) for (int i = 0; i < (n > 0 ? n : 0); ++i) n_out[i] = 0;
  int rc = check_geometry(h, n, width, height, stride, cap);
  if (rc) return rc;
  if (!gray || !kps_xy || !scores || !desc || !n_out) return fail(D2FE_ERR_INVALID, "null pointer");
  if (n_netvlad < 0 || n_netvlad > n || (n_netvlad > 0 && !netvlad_out)) return fail(D2FE_ERR_INVALID, "bad NetVLAD image count / null output");
  if (n_netvlad > 0) { rc = nv_check(h, n_netvlad, width, height, stride); if (rc) return rc; }
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  hipStream_t s = h->stream;
  // the call's capacity, bounded by what can exist: H*W candidates (keep-all) or the configured maximum
  const long most = h->cfg.max_keypoints > 0 ? h->cfg.max_keypoints : (long)height * width;
  const int dcap = cap < most ? cap : (int)most;
  const size_t D = (h->cfg.postproc == D2FE_POSTPROC_A && h->pca_dims) ? (size_t)h->pca_dims : 256;
  // outputs of the call, contiguous in one device block: kps [n][dcap][2] | scores [n][dcap] | desc [n][dcap][D] | counts [n] | idx [n][dcap]
  const size_t out_bytes = sizeof(float) * (size_t)n * dcap * (3 + D) + sizeof(int32_t) * n;
  const size_t need = out_bytes + sizeof(int32_t) * (size_t)n * dcap;
  if (need > h->s_out_bytes) {       // grow-only staging: a keep-all call may ask for far more than the initial 1024 per image
    HIP_TRY(hipStreamSynchronize(s));
    graphs_clear(h);                 // captured launch sequences hold the old staging pointers
    if (h->s_out) { hipFree(h->s_out); h->s_out = nullptr; h->s_out_bytes = 0; }
    HIP_TRY(hipMalloc(&h->s_out, need));
    h->s_out_bytes = need;
    if (h->use_pinned && out_bytes > h->pin_out_bytes) {
      if (h->pin_out) { (void)hipHostFree(h->pin_out); h->pin_out = nullptr; h->pin_out_bytes = 0; }
      if (hipHostMalloc(&h->pin_out, out_bytes, hipHostMallocDefault) == hipSuccess) h->pin_out_bytes = out_bytes;
      else { (void)hipGetLastError(); h->pin_out = nullptr; }     // the pageable path below serves calls the pinned buffer cannot
    }
  }
  rc = upload_frames(h, h->s_img, gray, n, width, height, stride, image_stride, s);
  if (rc) return rc;
  size_t nv_bytes = 0;
  // once NetVLAD work is queued on nv_stream, EVERY exit waits for it: its D2H may target the caller's netvlad_out, and the next call's upload
  // overwrites the frames it reads
  struct NvGuard { d2fe_context* h; bool armed = false; ~NvGuard() { if (armed && h->nv_stream) (void)hipStreamSynchronize(h->nv_stream); } } nv_guard{h};
  if (n_netvlad > 0) {
    // NetVLAD reads the frames SuperPoint reads (one upload), on its own stream: at one or two images per call both launch sequences are
    // latency-bound and leave most of the chip idle, so they overlap almost completely (the reference calls infer and then inference
    // for the same image, loop_cam.cpp:609-616)
    // (a stream on another hardware pipe than the handle's own: two busy streams on one pipe take turns, pipe.hip)
    if (!h->nv_stream) { HIP_TRY(create_stream_beside(h->cfg.device_id, h->stream, &h->nv_stream)); HIP_TRY(hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming)); }
    const int G = d2fe_netvlad_dim(h);
    nv_bytes = sizeof(float) * (size_t)G * n_netvlad;
    if (h->use_pinned && nv_bytes > h->pin_nv_bytes) {
      if (h->pin_nv) { (void)hipHostFree(h->pin_nv); h->pin_nv = nullptr; h->pin_nv_bytes = 0; }
      const size_t want = sizeof(float) * 8192 * (size_t)h->cfg.max_batch;
      if (hipHostMalloc(&h->pin_nv, want > nv_bytes ? want : nv_bytes, hipHostMallocDefault) == hipSuccess) h->pin_nv_bytes = want > nv_bytes ? want : nv_bytes;
      else { (void)hipGetLastError(); h->pin_nv = nullptr; }
    }
    HIP_TRY(hipEventRecord(h->ev_up, s));
    HIP_TRY(hipStreamWaitEvent(h->nv_stream, h->ev_up, 0));
    nv_guard.armed = true;
    rc = run_cached(h, {3, n_netvlad, width, height, (long)h->nv_pca_m, 0}, h->nv_stream, [&](hipStream_t st) {
      return run_netvlad(h, h->s_img, n_netvlad, width, height, width, (size_t)width * height, h->nv_s_out, st);
    });
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->pin_nv ? (void*)h->pin_nv : (void*)netvlad_out, h->nv_s_out, nv_bytes, hipMemcpyDeviceToHost, h->nv_stream));
  }
  // async_tail handles: a preceding d2fe_superpoint_extract_device call may still have its post-processing running on the tail
  // stream against the single-buffered scratch (candidates, score map, sparse-head slots) that this run uses too
  if (h->cfg.async_tail)
    for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamWaitEvent(s, h->ev_tail[i], 0));
  float* o_kps = h->s_out;
  float* o_sc = o_kps + (size_t)n * dcap * 2;
  float* o_desc = o_sc + (size_t)n * dcap;
  int32_t* o_n = reinterpret_cast<int32_t*>(o_desc + (size_t)n * dcap * D);
  int32_t* o_idx = o_n + n;
  rc = run_cached(h, {1, n, width, height, dcap, (long)D}, s, [&](hipStream_t st) {
    return run_superpoint(h, h->s_img, n, width, height, width, (size_t)width * height, o_kps, o_sc, o_desc, o_idx, dcap, o_n, st);
  });
  if (rc) return rc;
  std::vector<int32_t> ncand(h->cfg.max_keypoints < 0 ? n : 0);
  if (!ncand.empty()) HIP_TRY(hipMemcpyAsync(ncand.data(), h->cand_count, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
  if (h->use_pinned && h->pin_out && out_bytes <= h->pin_out_bytes) {
    // ONE D2H of the whole block into pinned memory, one synchronisation, then the rows that exist go to the caller's arrays
    HIP_TRY(hipMemcpyAsync(h->pin_out, h->s_out, out_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const float* p_kps = h->pin_out;
    const float* p_sc = p_kps + (size_t)n * dcap * 2;
    const float* p_desc = p_sc + (size_t)n * dcap;
    const int32_t* p_n = reinterpret_cast<const int32_t*>(p_desc + (size_t)n * dcap * D);
    for (int i = 0; i < n; ++i) {
      const int k = p_n[i] < dcap ? p_n[i] : dcap;
      n_out[i] = k;
      if (k > 0) {
        memcpy(kps_xy + (size_t)i * cap * 2, p_kps + (size_t)i * dcap * 2, sizeof(float) * 2 * k);
        memcpy(scores + (size_t)i * cap, p_sc + (size_t)i * dcap, sizeof(float) * k);
        memcpy(desc + (size_t)i * cap * D, p_desc + (size_t)i * dcap * D, sizeof(float) * D * k);
      }
    }
  } else {
    std::vector<int32_t> cnt(n);
    HIP_TRY(hipMemcpyAsync(cnt.data(), o_n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
      const int k = cnt[i];
      n_out[i] = k;
      if (k > 0) {
        HIP_TRY(hipMemcpyAsync(kps_xy + (size_t)i * cap * 2, o_kps + (size_t)i * dcap * 2, sizeof(float) * 2 * k, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(scores + (size_t)i * cap, o_sc + (size_t)i * dcap, sizeof(float) * k, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(desc + (size_t)i * cap * D, o_desc + (size_t)i * dcap * D, sizeof(float) * D * k, hipMemcpyDeviceToHost, s));
      }
    }
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (n_netvlad > 0) {
    HIP_TRY(hipStreamSynchronize(h->nv_stream));
    nv_guard.armed = false;
    if (h->pin_nv) memcpy(netvlad_out, h->pin_nv, nv_bytes);
  }
  for (int32_t c : ncand)
    if (c > dcap) return fail(D2FE_ERR_TRUNCATED, "max_keypoints = -1: an image has more keypoints than the call's capacity; the strongest were kept");
  return D2FE_OK;
}

int d2fe_superpoint_extract_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride,
                                  size_t image_stride, float* kps_xy, float* scores, float* desc, int cap, int* n_out) {
  return extract_host(h, gray, n, width, height, stride, image_stride, kps_xy, scores, desc, cap, n_out, 0, nullptr);
}

int d2fe_extract_all_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride, size_t image_stride, float* kps_xy,
                           float* scores, float* desc, int cap, int* n_out, int n_netvlad, float* netvlad_out) {
  return extract_host(h, gray, n, width, height, stride, image_stride, kps_xy, scores, desc, cap, n_out, n_netvlad, netvlad_out);
}

int d2fe_extract_all(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, float* kps_xy, float* scores, float* desc, int cap,
                     int* n_out, float* netvlad_out) {
  return extract_host(h, gray, 1, width, height, stride, (size_t)stride * height, kps_xy, scores, desc, cap, n_out, 1, netvlad_out);
}

int d2fe_superpoint_extract(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, float* kps_xy,
                            float* scores, float* desc, int cap, int* n_out) {
  return d2fe_superpoint_extract_batch(h, gray, 1, width, height, stride, (size_t)stride * height, kps_xy, scores, desc,
                                       cap, n_out);
}

}  // extern "C"
// ---- NetVLAD -----------------------------------------------------------------------------------------------------------
namespace {
void nv_free(d2fe_context* h) {
  const bool own = !h->borrowed;      // a pipeline lane owns its activations only
  for (auto& l : h->nv) { if (own && l.w) hipFree(l.w); if (own && l.b) hipFree(l.b); if (l.out) hipFree(l.out); }
  h->nv.clear();
  if (own) for (auto& st : h->nv_plan) { if (st.we) hipFree(st.we); if (st.wp) hipFree(st.wp); if (st.bp) hipFree(st.bp); if (st.w0) hipFree(st.w0); if (st.wp2) hipFree(st.wp2); if (st.bp2) hipFree(st.bp2); }
  h->nv_plan.clear();
  for (float** p : {&h->nv_pre_w, &h->nv_pre_b, &h->nv_aw, &h->nv_aw_pack, &h->nv_ab, &h->nv_cen})
    if (*p) { if (own) hipFree(*p); *p = nullptr; }
  for (float** p : {&h->nv_feat_buf, &h->nv_raw, &h->nv_pca_out, &h->nv_part})
    if (*p) { hipFree(*p); *p = nullptr; }
  if (h->nv_stamps) { hipFree(h->nv_stamps); h->nv_stamps = nullptr; }
  h->nv_loaded = false;
}
inline int same_out(int in, int stride) { return (in + stride - 1) / stride; }
inline int same_pad_begin(int in, int stride, int out) { const int t = (out - 1) * stride + 3 - in; return t > 0 ? t / 2 : 0; }   // TF "SAME", 3x3

// hidden-channel groups for a fused step: enough workgroups to fill the chip (2 per CU), at least 3 chunks of 16 per group (every
// group stages the whole input patch again, and its consumer reads one more partial slab).  Three or more groups (two, when the consumer reads a
// single slab: `sum_at_2`) cost a slab-sum launch of ~6 us behind the block; a chunk costs ~2.3 us of a workgroup's latency (tools/nv_stamps.py):
// the split goes past two groups only when the chunks it takes off every workgroup are worth more than that launch (30 x 40 layers: 9-12 chunks,
// two groups; 15 x 20 layers: 60 chunks, seven)
inline void nv_groups(long base_blocks, int nchunk, int gmax, int* groups, int* cpg, int target, long cap = 0, bool sum_at_2 = false, bool rule = true) {
  int g = (int)((target + base_blocks - 1) / base_blocks);
  if (cap > 0 && g > 1 && g * base_blocks > cap) --g;      // a second round of workgroups costs more than one more chunk per group
  if (g > gmax) g = gmax;
  if (g > nchunk / 3) g = nchunk / 3;
  if (g < 1) g = 1;
  auto per = [&](int gg) { return (nchunk + gg - 1) / gg; };
  const double chunk_us = 2.3, launch_us = 6.0;
  if (rule && g >= 3 && (per(2) - per(g)) * chunk_us < launch_us) g = 2;
  if (rule && g == 2 && sum_at_2 && (per(1) - per(2)) * chunk_us < launch_us) g = 1;
  // round 6 (MobileNetV2-0.75: 9 chunks at 120 x 160 and 60 x 80): with 32 or more tiles per image a second group halves a workgroup's chunks (~10 us of ONE image's
  // latency) but makes every batch stage each input patch twice, write and re-read a second slab and, where the consumer wants one slab, launch the slab sum:
  // measured at 32 images 194 + 26 us with two groups against 160 us with one (profiles/r06_netvlad_timeline.txt).  The split stays per IMAGE (batch invariance)
  if (rule && g == 2 && base_blocks >= 32 && nchunk <= 12) g = 1;
  *cpg = per(g);
  *groups = (nchunk + *cpg - 1) / *cpg;
}

}  // namespace
namespace d2fe {
int run_netvlad(d2fe_context* h, const uint8_t* d_gray, int n, int W, int H, int stride, size_t image_stride, float* d_out,
                hipStream_t s) {
  ProfScope ps(h, D2FE_PROF_NETVLAD, s);
  int ch = H, cw = W;
  bool feat_done = false;
  for (size_t si = 0; si < h->nv_plan.size(); ++si) {
    const auto& st = h->nv_plan[si];
    const bool next_fused = si + 1 < h->nv_plan.size() && h->nv_plan[si + 1].fused;
    // nv_xblock_kernel (stride-2 blocks) and nv_tail_kernel read ONE input slab: their producer's partial slabs are summed first
    const bool next_single = next_fused && ((h->nv_plan[si + 1].xblock && !h->nv_plan[si + 1].pblock) || h->nv_plan[si + 1].tail ||
                                            (h->nv_plan[si + 1].pblock && !h->nv_plan[si + 1].front && nv_pblock_single_input(h->nv[h->nv_plan[si + 1].l0].cin)));
    if (!st.fused) {
      auto& l = h->nv[st.l0];
      const float* in = st.l0 ? h->nv[st.l0 - 1].out : nullptr;
      const int ho = same_out(ch, l.stride), wo = same_out(cw, l.stride);
      if (l.kind == D2FE_NV_CONV) {
        HIP_TRY(launch_nv_conv0(d_gray, stride, (long)image_stride, ch, cw, ho, wo, l.stride, l.cout, l.act, l.w, l.b, l.out, n, s));
      } else if (l.kind == D2FE_NV_DW) {
        HIP_TRY(launch_nv_dw(in, ch, cw, l.cin, ho, wo, l.stride, l.act, l.w, l.b, l.out, n, s));
      } else {
        HIP_TRY(launch_nv_pw(in, (long)n * ch * cw, l.cin, l.cout, l.cout_pad, l.act, l.w, l.b, l.res >= 0 ? h->nv[l.res].out : nullptr,
                             l.out, s));
      }
      l.slabs = 1; l.slab_stride = 0;
      ch = ho; cw = wo;
      continue;
    }
    NvBlockArgs a{};
    int li = st.l0;
    if (st.front) {
      const auto& c0 = h->nv[li++];
      a.img = d_gray; a.img_stride = stride; a.img_istride = (long)image_stride; a.H0 = ch; a.W0 = cw;
      const int ho = same_out(ch, c0.stride), wo = same_out(cw, c0.stride);
      a.c0_stride = c0.stride; a.c0_pt = same_pad_begin(ch, c0.stride, ho); a.c0_pl = same_pad_begin(cw, c0.stride, wo);
      a.act0 = c0.act; a.w0 = st.w0;
      ch = ho; cw = wo;
    } else {
      const auto& pin = h->nv[st.l0 - 1];
      a.in = pin.out; a.in_slabs = pin.slabs; a.in_slab_stride = pin.slab_stride;
    }
    int groups = 1, cpg = 0;
    if (st.tail) {
      // the trunk's last 1x1 (expand: feat_dim hidden channels) chained with the NetVLAD pre-projection, over the flat pixel list
      const auto& e = h->nv[li];
      a.we = st.we; a.act_e = e.act;
      a.H = ch; a.W = cw; a.Ho = ch; a.Wo = cw; a.Cin = e.cin; a.Chid = e.cout; a.Cout = h->nv_proj; a.stride = 1;
      a.P = (long)n * ch * cw;
      a.wp = st.wp; a.bp = st.bp; a.act_p = 0;
      // three workgroups per CU fit (registers), and the MFMA pipe is the limit: ~768 workgroups of equal length load every SIMD alike
      // (the hidden-channel split is decided on ONE image's pixel count whatever the batch: see the block steps below)
      nv_groups(((long)ch * cw + 127) / 128, a.Chid / 16, h->nv_feat_gmax, &groups, &cpg, h->nv_tail_blocks);
      a.cpg = cpg; a.out = h->nv_feat_buf; a.out_slab_stride = a.P * a.Cout;
      h->nv_feat_slabs = groups; h->nv_feat_slab_stride = a.out_slab_stride;
      if (nv_tail_supported(a.Cin, a.Cout)) HIP_TRY(launch_nv_tail(a, groups, s));     // st.we was packed in that kernel's K order
      else HIP_TRY(launch_nv_block(a, true, 2, n, groups, s));
      // no slab sum here: the VLAD stage reads every feature exactly once and adds the slabs, in slab order, while it stages them
      feat_done = true;
      continue;
    }
    if (st.expand) { const auto& e = h->nv[li++]; a.we = st.we; a.act_e = e.act; }
    const auto& d = h->nv[li++];
    auto& pj = h->nv[li];
    a.H = ch; a.W = cw; a.Cin = st.expand ? h->nv[st.l0].cin : d.cin; a.Chid = d.cin; a.Cout = pj.cout; a.stride = d.stride;
    a.Ho = same_out(ch, d.stride); a.Wo = same_out(cw, d.stride);
    a.pt = same_pad_begin(ch, d.stride, a.Ho); a.pl = same_pad_begin(cw, d.stride, a.Wo);
    a.act_d = d.act;
    a.wp = st.wp; a.bp = st.bp; a.act_p = pj.act;
    if (pj.res >= 0) { const auto& r = h->nv[pj.res]; a.res = r.out; a.res_slabs = r.slabs; a.res_slab_stride = r.slab_stride; }
    a.th = 8; a.tw = 16;
    if (st.pblock) nv_pblock_tile(a.Ho, a.Wo, &a.th, &a.tw, st.front ? a.c0_stride : 0);
    else if (st.xblock) nv_xblock_tile(a.Ho, a.Wo, a.stride, &a.th, &a.tw);
    const long tiles1 = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th);
    const long tiles = tiles1 * n;
    // partial slabs are summed by the consumer's staging: only when that consumer is a fused step
    // pixel-pair kernel: no more workgroups than 85 % of what the device holds at once (registers / LDS of that block shape).
    // The split of the hidden channels over workgroup groups fixes the fp32 summation order of the block's output, so it is decided on ONE
    // image's tile count and the DEVICE's compute units (not the batch, not a pipeline lane's share): an image's descriptor is the same bits
    // in a 1-image call, a 32-image batch and any pass of the frames-in-flight pipe.  A batch then runs with more groups than it needs to fill
    // the device (15 x 20 layers at 32 images: 7 slabs instead of 4) -- a few MB of partial-slab traffic
    // a consumer that is NOT a fused step (a generic per-layer launch: the stride-2 block 72 -> 432 -> 120 of the 0.75-wide trunk) reads one plain tensor: the split is
    // still worth it (27 chunks in ONE workgroup per tile ran 115 us at 32 images and 70 us for one image; nine groups + the slab sum: 55 + 17 us), the slabs are summed below
    const bool sum_for_plain_consumer = !next_fused;
    nv_groups(tiles1, a.Chid / 16, pj.gmax, &groups, &cpg, h->nv_blocks_target,
              st.pblock ? nv_pblock_slots(a.Cin, a.Cout, h->ncu_dev, 1) * 85 / 100 : 0, next_single || sum_for_plain_consumer, h->nv_group_rule);
    a.cpg = cpg; a.out = pj.out; a.out_slab_stride = (long)n * a.Ho * a.Wo * a.Cout;
    // Six or more groups per image (30 x 40 and 15 x 20 layers: what ONE image needs to reach 60-135 workgroups) are 2-3 rounds of short workgroups for a batch, each
    // staging its input patch again and writing its own slab.  The summation order of such a layer is a two-level tree -- runs of `tree` groups, then the runs in order --
    // and a batch lets one workgroup walk a whole run (NvBlockArgs::gmerge): same bits as one image's unmerged launch + tree-ordered slab sum, a third of the
    // workgroups, patch loads and slabs.  `tree` depends on the layer alone, merging on the batch
    int tree = 1, wgroups = groups;
    const int half0 = nv_pblock_half_cout(a.Cout, 0);
    if (st.pblock && !st.front && groups >= 6 && nv_pblock_can_merge(a.Cin, half0) &&
        (!st.wp2 || nv_pblock_can_merge(a.Cin, a.Cout - half0))) {
      tree = 3;
      const long slots = nv_pblock_slots(a.Cin, a.Cout, h->ncu_dev, 1);
      if (h->nv_merge && tiles * groups > slots && tiles * ((groups + tree - 1) / tree) * 2 >= (h->ncu_dev > 0 ? h->ncu_dev : 256)) { a.gmerge = tree; wgroups = (groups + tree - 1) / tree; }
    }
    pj.slabs = wgroups; pj.slab_stride = a.out_slab_stride;
    a.ncu = h->ncu; a.tpw = h->nv_front_tpw; a.nbuf = h->nv_nbuf;
    if ((int)si == h->nv_stamp_step && h->nv_stamps && (tiles * wgroups <= 32768)) {
      HIP_TRY(hipMemsetAsync(h->nv_stamps, 0, sizeof(unsigned long long) * 32 * 32768, s));
      a.stamps = h->nv_stamps; h->nv_stamp_wgs = (int)(tiles * wgroups);
    }
    if (st.pblock && st.front) HIP_TRY(launch_nv_fpair(a, n, s));
    else if (st.pblock && st.wp2) {
      // more than 128 output channels: two launches over channel halves, each with its own project record (the expand + depthwise stages run in both)
      NvBlockArgs h0 = a, h1 = a;
      h0.co0 = 0; h0.Cv = nv_pblock_half_cout(a.Cout, 0);
      h1.co0 = h0.Cv; h1.Cv = a.Cout - h0.Cv; h1.wp = st.wp2; h1.bp = st.bp2; h1.stamps = nullptr;
      HIP_TRY(launch_nv_pblock(h0, n, groups, s));
      HIP_TRY(launch_nv_pblock(h1, n, groups, s));
    }
    else if (st.pblock) HIP_TRY(launch_nv_pblock(a, n, groups, s));
    else if (st.xblock) HIP_TRY(launch_nv_xblock(a, n, groups, s));
    else HIP_TRY(launch_nv_block(a, st.expand, st.front ? 1 : 0, n, groups, s));
    // three or more partial slabs: sum them once instead of in every consumer workgroup (and in every residual read)
    // (decided on `groups`, the layer's own count: a consumer sees one slab or several whatever the batch merged)
    // (a tree-ordered layer is always summed here: a consumer adding the slabs itself would do so in slab order, i.e. differently for merged and unmerged launches)
    if (tree > 1 || (h->nv_slabsum > 0 && groups >= h->nv_slabsum) || ((next_single || sum_for_plain_consumer) && groups > 1)) {
      HIP_TRY(launch_nv_slab_sum(pj.out, wgroups, pj.slab_stride, pj.slab_stride, s, a.gmerge > 1 ? 1 : tree));
      pj.slabs = 1;
    }
    ch = a.Ho; cw = a.Wo;
  }
  const int np = ch * cw;
  if (!feat_done) {
    const int pp = (h->nv_proj + 31) / 32 * 32;
    HIP_TRY(launch_nv_pw(h->nv.back().out, (long)n * np, h->nv_feat, h->nv_proj, pp, 0, h->nv_pre_w, h->nv_pre_b, nullptr, h->nv_feat_buf, s));
    h->nv_feat_slabs = 1; h->nv_feat_slab_stride = 0;
  }
  float* raw = h->nv_pca_m ? h->nv_raw : d_out;
  HIP_TRY(launch_nv_vlad(h->nv_feat_buf, h->nv_feat_slabs, h->nv_feat_slab_stride, np, h->nv_proj, h->nv_k, h->nv_aw, h->nv_aw_pack, h->nv_ab, h->nv_cen,
                         h->nv_part, raw, n, s));
  if (h->nv_pca_m) HIP_TRY(launch_nv_pca(raw, h->nv_k * h->nv_proj, h->nv_pca_comp, h->nv_pca_mean, h->nv_pca_m, d_out, n, s));
  return D2FE_OK;
}
}  // namespace d2fe
extern "C" {

int d2fe_load_netvlad(d2fe_handle h, const d2fe_netvlad_weights* w) {
  if (h && h->live_pipes.load() > 0) return fail(D2FE_ERR_INVALID, "the handle has live pipes whose lanes read its packed weights: destroy them before loading weights or PCA matrices");
  if (h) graphs_clear(h);
  if (!h || !w || !w->layers || w->n_layers < 1) return fail(D2FE_ERR_INVALID, "null argument");
  if (!w->pre_w || !w->pre_b || !w->assign_w || !w->assign_b || !w->centroids) return fail(D2FE_ERR_INVALID, "null head weights");
  if (w->n_clusters < 1 || w->n_clusters > 64 || w->proj_dim < 4 || w->proj_dim > 256 || (w->proj_dim & 3) ||
      w->n_clusters * w->proj_dim > 8192 || (w->feat_dim & 3))
    return fail(D2FE_ERR_INVALID, "unsupported NetVLAD head shape");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipStreamSynchronize(h->stream));
  nv_free(h);
  const int B = h->cfg.max_batch;
  int ch = h->cfg.max_height, cw = h->cfg.max_width, cprev = 1;
  // any failure below leaves the handle without a NetVLAD network and frees what was uploaded so far
  struct Guard { d2fe_context* h; bool ok = false; ~Guard() { if (!ok) nv_free(h); } } guard{h};
  for (int i = 0; i < w->n_layers; ++i) {
    const d2fe_nv_layer& L = w->layers[i];
    d2fe_context::NvLayer l;
    l.kind = L.kind; l.cin = L.cin; l.cout = L.cout; l.stride = L.stride; l.act = L.act; l.res = L.res;
    if (!L.weight || !L.bias || L.stride < 1 || L.stride > 2 || L.cin != cprev || L.res >= i || L.act < 0 || L.act > 2)
      return fail(D2FE_ERR_INVALID, "netvlad layer " + std::to_string(i) + ": bad descriptor");
    std::vector<float> wt, bt;
    if (L.kind == D2FE_NV_CONV) {
      if (i != 0 || L.cin != 1 || L.cout > 32) return fail(D2FE_ERR_INVALID, "conv layer must be first, 1 -> <=32 channels");
      wt.assign(9 * 32, 0.f); bt.assign(32, 0.f);
      for (int co = 0; co < L.cout; ++co) { bt[co] = L.bias[co]; for (int t = 0; t < 9; ++t) wt[t * 32 + co] = L.weight[co * 9 + t]; }
      l.cout_pad = 32;
    } else if (L.kind == D2FE_NV_DW) {
      if (L.cin != L.cout || (L.cin & 3)) return fail(D2FE_ERR_INVALID, "depthwise layer: channels must match and be a multiple of 4");
      wt.resize(9 * (size_t)L.cin); bt.assign(L.bias, L.bias + L.cin);
      for (int c = 0; c < L.cin; ++c) for (int t = 0; t < 9; ++t) wt[(size_t)t * L.cin + c] = L.weight[c * 9 + t];
      l.cout_pad = L.cout;
    } else if (L.kind == D2FE_NV_PW) {
      if ((L.cin & 3) || L.stride != 1) return fail(D2FE_ERR_INVALID, "pointwise layer: cin must be a multiple of 4, stride 1");
      if (L.cin & 7) return fail(D2FE_ERR_INVALID, "pointwise layer: cin must be a multiple of 8");
      l.cout_pad = (L.cout + 31) / 32 * 32;
      wt.resize(packed_weight_floats_f32(l.cout_pad, L.cin, 1)); bt.assign(l.cout_pad, 0.f);
      pack_weights_f32(L.weight, L.cout, L.cin, 1, l.cout_pad, wt.data());
      for (int co = 0; co < L.cout; ++co) bt[co] = L.bias[co];
    } else {
      return fail(D2FE_ERR_INVALID, "unknown layer kind");
    }
    ch = same_out(ch, L.stride); cw = same_out(cw, L.stride);
    l.oh = ch; l.ow = cw;
    if (L.res >= 0) {
      // a skip connection adds two tensors of the SAME shape: channels and spatial size (a stride-2 layer in between would make
      // the 1x1 kernel read past the smaller buffer)
      const auto& r = h->nv[L.res];
      if (L.kind != D2FE_NV_PW || r.cout != L.cout || r.oh != ch || r.ow != cw)
        return fail(D2FE_ERR_INVALID, "netvlad layer " + std::to_string(i) + ": residual source has a different shape");
    }
    h->nv.push_back(l);                 // pushed before the uploads: nv_free() releases whatever made it to the device
    auto& dl = h->nv.back();
    int rc = upload(wt.data(), wt.size() * sizeof(float), reinterpret_cast<void**>(&dl.w));
    rc = rc ? rc : upload(bt.data(), bt.size() * sizeof(float), reinterpret_cast<void**>(&dl.b));
    if (rc) return rc;
    cprev = L.cout;
  }
  // ---- plan: fuse [conv0 ->] [pw expand ->] dw -> pw project where the kernels support the shape (D2FE_NV_LEGACY=1: one launch per layer)
  {
    // schedule switches of the development library (A/B measurements; the product library always takes the defaults: the measured best)
    const bool legacy = d2fe_dev_env("D2FE_NV_LEGACY", 0) != 0;
    { const int v = d2fe_dev_env("D2FE_NV_BLOCKS", 0); if (v > 0) h->nv_blocks_target = v; }
    { const int v = d2fe_dev_env("D2FE_NV_TAIL_BLOCKS", 0); if (v > 0) h->nv_tail_blocks = v; }
    h->nv_slabsum = d2fe_dev_env("D2FE_NV_SLABSUM", h->nv_slabsum);
    h->nv_group_rule = d2fe_dev_env("D2FE_NV_GROUP_RULE", 1) != 0;
    h->nv_merge = d2fe_dev_env("D2FE_NV_MERGE", 1) != 0;
    h->nv_front_tpw = d2fe_dev_env("D2FE_NV_FRONT_TPW", 0); h->nv_nbuf = d2fe_dev_env("D2FE_NV_NBUF", 0);
    h->nv_stamp_step = d2fe_dev_env("D2FE_NV_STAMP_STEP", -1);
    if (h->nv_stamp_step >= 0 && !h->nv_stamps) HIP_TRY(hipMalloc(&h->nv_stamps, sizeof(unsigned long long) * 32 * 32768));
    const int nl = w->n_layers;
    auto K = [&](int i) { return i < nl ? h->nv[i].kind : -1; };
    std::vector<char> materialised(nl, 0);
    int i = 0;
    while (i < nl) {
      d2fe_context::NvStep st;
      st.l0 = st.l1 = i;
      if (!legacy) {
        const bool pair_on = d2fe_dev_env("D2FE_NV_PAIR", 1) != 0;
        // pixel-pair forms (netvlad_pair.hip): the first block with a first conv of 16 / 24 / 32 channels, stride-1 expand blocks of the widths in NVP_SHAPES
        const bool fp_ok = pair_on && K(i) == D2FE_NV_CONV && K(i + 1) == D2FE_NV_DW && K(i + 2) == D2FE_NV_PW &&
                           nv_fpair_supported(h->nv[i].cout, h->nv[i].stride, h->nv[i + 1].stride, h->nv[i + 2].cout);
        const bool pb_ok = pair_on && i > 0 && K(i) == D2FE_NV_PW && K(i + 1) == D2FE_NV_DW && K(i + 2) == D2FE_NV_PW &&
                           nv_pblock_supported(h->nv[i].cin, h->nv[i].cout, h->nv[i + 2].cout, h->nv[i + 1].stride);
        if (K(i) == D2FE_NV_CONV && K(i + 1) == D2FE_NV_DW && K(i + 2) == D2FE_NV_PW && h->nv[i + 1].res < 0 && h->nv[i + 2].res < 0 &&
            (fp_ok || nv_block_supported(h->nv[i + 1].cin, h->nv[i + 1].cin, h->nv[i + 2].cout, h->nv[i + 1].stride, false, 1))) {
          st.fused = true; st.front = true; st.l1 = i + 2;
          st.pblock = fp_ok;
        } else if (i > 0 && K(i) == D2FE_NV_PW && K(i + 1) == D2FE_NV_DW && K(i + 2) == D2FE_NV_PW && h->nv[i].res < 0 &&
                   (pb_ok || nv_block_supported(h->nv[i].cin, h->nv[i].cout, h->nv[i + 2].cout, h->nv[i + 1].stride, true, 0))) {
          st.fused = true; st.expand = true; st.l1 = i + 2;
          {       // input-in-registers form of the block where the shape allows it (otherwise the LDS-resident form)
            st.xblock = d2fe_dev_env("D2FE_NV_XBLOCK", 1) != 0 && nv_xblock_supported(h->nv[i].cin, h->nv[i].cout, h->nv[i + 2].cout, h->nv[i + 1].stride);
            // stride 1: the pixel-pair form of the same block (netvlad_pair.hip; D2FE_NV_PAIR=0: nv_xblock_kernel / nv_block_kernel)
            st.pblock = pb_ok; }
        } else if (i > 0 && K(i) == D2FE_NV_DW && K(i + 1) == D2FE_NV_PW &&
                   nv_block_supported(h->nv[i].cin, h->nv[i].cin, h->nv[i + 1].cout, h->nv[i].stride, false, 0)) {
          st.fused = true; st.l1 = i + 1;
        } else if (i > 0 && i == nl - 1 && K(i) == D2FE_NV_PW && h->nv[i].res < 0 &&
                   (nv_tail_supported(h->nv[i].cin, w->proj_dim) || nv_block_supported(h->nv[i].cin, h->nv[i].cout, w->proj_dim, 1, true, 2))) {
          st.fused = true; st.tail = true; st.expand = true;        // last 1x1 of the trunk + the NetVLAD pre-projection in one launch
        }
        // a residual must read a tensor that exists in HBM: the output of an earlier step
        if (st.fused && !st.tail && h->nv[st.l1].res >= 0 && !materialised[h->nv[st.l1].res]) { st = d2fe_context::NvStep(); st.l0 = st.l1 = i; }
      }
      if (!st.fused && h->nv[i].res >= 0 && !materialised[h->nv[i].res])
        return fail(D2FE_ERR_UNSUPPORTED, "netvlad layer " + std::to_string(i) + ": residual source is internal to a fused block");
      if (st.fused) {
        int li = st.l0 + (st.front ? 1 : 0);
        if (st.front) {
          std::vector<float> pk(384);
          pack_nv_conv0(w->layers[st.l0].weight, w->layers[st.l0].bias, w->layers[st.l0].cout, pk.data());
          const int rc = upload(pk.data(), pk.size() * sizeof(float), reinterpret_cast<void**>(&st.w0));
          if (rc) { h->nv_plan.push_back(st); return rc; }
        }
        if (st.expand) {
          const d2fe_nv_layer& E = w->layers[li++];
          std::vector<float> pk(pack_nv_expand_floats(E.cout, E.cin));
          if (st.tail && nv_tail_supported(E.cin, w->proj_dim)) pack_nv_expand_tail(E.weight, E.bias, E.cout, E.cin, pk.data());
          else if (st.pblock) { pk.assign(pack_nv_expand_pair_floats(E.cout, E.cin), 0.f); pack_nv_expand_pair(E.weight, E.bias, E.cout, E.cin, pk.data()); }
          else if (st.xblock) { pk.assign(pack_nv_expand_perm_floats(E.cout, E.cin), 0.f); pack_nv_expand_perm(E.weight, E.bias, E.cout, E.cin, pk.data()); }
          else pack_nv_expand(E.weight, E.bias, E.cout, E.cin, pk.data());
          const int rc = upload(pk.data(), pk.size() * sizeof(float), reinterpret_cast<void**>(&st.we));
          if (rc) { h->nv_plan.push_back(st); return rc; }
        }
        // depthwise + project record: the block's dw 3x3 and last 1x1, or (tail) no dw and the NetVLAD pre-projection [proj_dim][feat_dim]
        const float* pwt = st.tail ? w->pre_w : w->layers[li + 1].weight;
        const float* pbs = st.tail ? w->pre_b : w->layers[li + 1].bias;
        const int pco = st.tail ? w->proj_dim : w->layers[li + 1].cout, pci = st.tail ? w->feat_dim : w->layers[li + 1].cin;
        // pixel-pair kernels: any n-tile count up to 8 per launch, wider outputs as two channel halves (each with its own project record and bias)
        const int halves = st.pblock ? nv_pblock_halves(pco) : 1, pco0 = st.pblock ? nv_pblock_half_cout(pco, 0) : pco;
        const int nt = st.pblock ? nv_pblock_ntiles(pco0) : nv_block_ntiles(pco);
        std::vector<float> pk(st.pblock ? pack_nv_dwproj_pair_floats(pci, nt) : pack_nv_dwproj_floats(pci, nt)), pb(nt * 16, 0.f);
        if (st.pblock) pack_nv_dwproj_pair(w->layers[li].weight, w->layers[li].bias, pwt, pco0, pci, nt, pk.data(), 0);
        else if (st.tail && nv_tail_supported(w->layers[st.l0].cin, w->proj_dim)) pack_nv_proj_t(pwt, pco, pci, nt, pk.data());
        else if (st.xblock && !st.tail) pack_nv_dwproj_x(w->layers[li].weight, w->layers[li].bias, pwt, pco, pci, nt, pk.data());
        else pack_nv_dwproj(st.tail ? nullptr : w->layers[li].weight, st.tail ? nullptr : w->layers[li].bias, pwt, pco, pci, nt, pk.data());
        for (int co = 0; co < pco0; ++co) pb[co] = pbs[co];
        h->nv_plan.push_back(st);
        auto& ds = h->nv_plan.back();
        int rc = upload(pk.data(), pk.size() * sizeof(float), reinterpret_cast<void**>(&ds.wp));
        rc = rc ? rc : upload(pb.data(), pb.size() * sizeof(float), reinterpret_cast<void**>(&ds.bp));
        if (rc) return rc;
        if (halves == 2) {
          const int pco1 = pco - pco0, nt1 = nv_pblock_ntiles(pco1);
          std::vector<float> pk1(pack_nv_dwproj_pair_floats(pci, nt1)), pb1(nt1 * 16, 0.f);
          pack_nv_dwproj_pair(w->layers[li].weight, w->layers[li].bias, pwt, pco1, pci, nt1, pk1.data(), pco0);
          for (int co = 0; co < pco1; ++co) pb1[co] = pbs[pco0 + co];
          rc = upload(pk1.data(), pk1.size() * sizeof(float), reinterpret_cast<void**>(&ds.wp2));
          rc = rc ? rc : upload(pb1.data(), pb1.size() * sizeof(float), reinterpret_cast<void**>(&ds.bp2));
          if (rc) return rc;
        }
        if (st.tail) {
          h->nv_feat_gmax = std::max(1, std::min(16, w->feat_dim / 32));
        } else if (h->nv[st.l1].act == 0) {
          // a linear bottleneck output may be written as partial slabs (hidden channels split over workgroup groups)
          const int nchunk = h->nv[st.l1].cin / 16;
          h->nv[st.l1].gmax = std::max(1, std::min(16, nchunk / 2));
        }
      } else {
        h->nv_plan.push_back(st);
      }
      if (!st.tail) materialised[st.l1] = 1;
      i = st.l1 + 1;
    }
    for (int li = 0; li < nl; ++li)
      if (materialised[li]) HIP_TRY(hipMalloc(&h->nv[li].out, sizeof(float) * (size_t)h->nv[li].gmax * B * h->nv[li].oh * h->nv[li].ow * h->nv[li].cout));
  }
  if (cprev != w->feat_dim) return fail(D2FE_ERR_INVALID, "feat_dim does not match the last layer");
  h->nv_feat = w->feat_dim; h->nv_proj = w->proj_dim; h->nv_k = w->n_clusters;
  const int pp = (w->proj_dim + 31) / 32 * 32;
  if (w->feat_dim & 7) return fail(D2FE_ERR_INVALID, "feat_dim must be a multiple of 8");
  std::vector<float> pw(packed_weight_floats_f32(pp, w->feat_dim, 1)), pb(pp, 0.f);
  pack_weights_f32(w->pre_w, w->proj_dim, w->feat_dim, 1, pp, pw.data());
  for (int co = 0; co < w->proj_dim; ++co) pb[co] = w->pre_b[co];
  int rc = upload(pw.data(), pw.size() * sizeof(float), reinterpret_cast<void**>(&h->nv_pre_w));
  rc = rc ? rc : upload(pb.data(), pb.size() * sizeof(float), reinterpret_cast<void**>(&h->nv_pre_b));
  rc = rc ? rc : upload(w->assign_w, sizeof(float) * w->n_clusters * w->proj_dim, reinterpret_cast<void**>(&h->nv_aw));
  if (w->n_clusters % 16 == 0 && w->proj_dim % 4 == 0) {
    std::vector<float> ap((size_t)w->n_clusters * w->proj_dim);
    pack_nv_assign(w->assign_w, w->n_clusters, w->proj_dim, ap.data());
    rc = rc ? rc : upload(ap.data(), ap.size() * sizeof(float), reinterpret_cast<void**>(&h->nv_aw_pack));
  }
  rc = rc ? rc : upload(w->assign_b, sizeof(float) * w->n_clusters, reinterpret_cast<void**>(&h->nv_ab));
  rc = rc ? rc : upload(w->centroids, sizeof(float) * w->n_clusters * w->proj_dim, reinterpret_cast<void**>(&h->nv_cen));
  if (rc) return rc;
  HIP_TRY(hipMalloc(&h->nv_feat_buf, sizeof(float) * (size_t)h->nv_feat_gmax * B * ch * cw * w->proj_dim));
  HIP_TRY(hipMalloc(&h->nv_raw, sizeof(float) * (size_t)B * w->n_clusters * w->proj_dim));
  HIP_TRY(hipMalloc(&h->nv_part, sizeof(float) * (size_t)B * nv_vlad_part_floats(ch * cw, w->proj_dim, w->n_clusters)));
  if (!h->nv_s_img) HIP_TRY(hipMalloc(&h->nv_s_img, (size_t)h->cfg.max_width * h->cfg.max_height * B));
  if (!h->nv_s_out) HIP_TRY(hipMalloc(&h->nv_s_out, sizeof(float) * 8192 * B));
  h->nv_loaded = true;
  guard.ok = true;
  return D2FE_OK;
}

#ifdef D2FE_DEVTOOLS      /* development library only: include/d2fe_debug.h */
/* diagnostics: with D2FE_NV_STAMP_STEP=<plan step> set at d2fe_load_netvlad() time, the wall_clock64() phase stamps [workgroup][32] that step's
 * nv_xblock_kernel wrote during the last d2fe_netvlad* call; returns the number of workgroups (tools/nv_stamps.py). */
long d2fe_debug_netvlad_stamps(d2fe_handle h, unsigned long long* dst, long max_wgs) {
  if (!h || !dst || !h->nv_stamps) return fail(D2FE_ERR_NOT_READY, "D2FE_NV_STAMP_STEP was not set when the network was loaded");
  hipSetDevice(h->cfg.device_id);
  const long nw = std::min<long>(max_wgs, h->nv_stamp_wgs);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(dst, h->nv_stamps, sizeof(unsigned long long) * 32 * nw, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(D2FE_ERR_HIP, "D2H");
  return nw;
}

/* test hook: the output of layer `layer` of the loaded network for the last d2fe_netvlad* call (NHWC fp32), if the execution plan
 * materialises it (the last layer of every fused block and every unfused layer); D2FE_ERR_NOT_READY otherwise. */
long d2fe_debug_netvlad_layer(d2fe_handle h, int layer, int n_images, void* dst, size_t max_bytes) {
  if (!h || !dst || !h->nv_loaded || layer < 0 || layer >= (int)h->nv.size() || n_images < 1 || n_images > h->cfg.max_batch)
    return fail(D2FE_ERR_INVALID, "bad argument");
  const auto& l = h->nv[layer];
  if (!l.out) return fail(D2FE_ERR_NOT_READY, "layer output lives inside a fused block");
  // spatial size of the LAST call: the plan works for any size up to the maximum; the caller passes images of the handle's maximum size here
  const size_t bytes = sizeof(float) * (size_t)n_images * l.oh * l.ow * l.cout;
  if (bytes > max_bytes) return fail(D2FE_ERR_TRUNCATED, "destination too small");
  hipSetDevice(h->cfg.device_id);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(dst, l.out, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(D2FE_ERR_HIP, "D2H");
  if (l.slabs > 1) {      // hidden-channel groups wrote partial slabs: the tensor is their sum (what the consumer's staging forms)
    if ((size_t)l.slab_stride * sizeof(float) != bytes) return fail(D2FE_ERR_INVALID, "n_images differs from the last call");
    std::vector<float> tmp(bytes / sizeof(float));
    float* o = static_cast<float*>(dst);
    for (int sl = 1; sl < l.slabs; ++sl) {
      if (hipMemcpy(tmp.data(), l.out + (size_t)sl * l.slab_stride, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(D2FE_ERR_HIP, "D2H");
      for (size_t i = 0; i < tmp.size(); ++i) o[i] += tmp[i];
    }
  }
  return (long)bytes;
}

#endif  // D2FE_DEVTOOLS

int d2fe_set_netvlad_pca(d2fe_handle h, const float* comp, const float* mean, int m) {
  if (h && h->live_pipes.load() > 0) return fail(D2FE_ERR_INVALID, "the handle has live pipes whose lanes read its packed weights: destroy them before loading weights or PCA matrices");
  if (h) graphs_clear(h);
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  if (!h->nv_loaded) return fail(D2FE_ERR_NOT_READY, "netvlad weights not loaded");
  const int G = h->nv_k * h->nv_proj;
  if (m < 0 || m > 8192 || (m > 0 && (!comp || !mean)) || (G & 3)) return fail(D2FE_ERR_INVALID, "bad PCA arguments");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->nv_pca_comp) { hipFree(h->nv_pca_comp); h->nv_pca_comp = nullptr; }
  if (h->nv_pca_mean) { hipFree(h->nv_pca_mean); h->nv_pca_mean = nullptr; }
  h->nv_pca_m = 0;
  if (m == 0) return D2FE_OK;
  int rc = upload(comp, sizeof(float) * (size_t)m * G, reinterpret_cast<void**>(&h->nv_pca_comp));
  rc = rc ? rc : upload(mean, sizeof(float) * G, reinterpret_cast<void**>(&h->nv_pca_mean));
  if (rc) return rc;
  h->nv_pca_m = m;
  return D2FE_OK;
}

int d2fe_netvlad_dim(d2fe_handle h) {
  if (!h || !h->nv_loaded) return fail(D2FE_ERR_NOT_READY, "netvlad weights not loaded");
  return h->nv_pca_m ? h->nv_pca_m : h->nv_k * h->nv_proj;
}

}  // extern "C"
namespace d2fe {
int nv_check(d2fe_context* h, int n, int W, int H, int stride) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  if (!h->nv_loaded) return fail(D2FE_ERR_NOT_READY, "netvlad weights not loaded");
  if (n < 1 || n > h->cfg.max_batch) return fail(D2FE_ERR_INVALID, "batch size out of range");
  if (W < 32 || H < 32 || W > h->cfg.max_width || H > h->cfg.max_height) return fail(D2FE_ERR_INVALID, "image size out of range");
  if (stride < W) return fail(D2FE_ERR_INVALID, "stride < width");
  return D2FE_OK;
}
}  // namespace d2fe
extern "C" {

int d2fe_netvlad_device(d2fe_handle h, const uint8_t* d_gray, int n, int width, int height, int stride, size_t image_stride,
                        float* d_out, void* stream) {
  int rc = nv_check(h, n, width, height, stride);
  if (rc) return rc;
  if (!d_gray || !d_out) return fail(D2FE_ERR_INVALID, "null device pointer");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  return run_netvlad(h, d_gray, n, width, height, stride, image_stride, d_out, stream ? (hipStream_t)stream : h->stream);
}

int d2fe_netvlad_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride, size_t image_stride,
                       float* out) {
  int rc = nv_check(h, n, width, height, stride);
  if (rc) return rc;
  if (!gray || !out) return fail(D2FE_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  hipStream_t s = h->stream;
  rc = upload_frames(h, h->nv_s_img, gray, n, width, height, stride, image_stride, s);
  if (rc) return rc;
  rc = run_cached(h, {2, n, width, height, (long)h->nv_pca_m, 0}, s, [&](hipStream_t st) {
    return run_netvlad(h, h->nv_s_img, n, width, height, width, (size_t)width * height, h->nv_s_out, st);
  });
  if (rc) return rc;
  const int G = d2fe_netvlad_dim(h);
  const size_t bytes = sizeof(float) * (size_t)G * n;
  if (h->use_pinned && h->pin_out && bytes <= h->pin_out_bytes) {
    HIP_TRY(hipMemcpyAsync(h->pin_out, h->nv_s_out, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(out, h->pin_out, bytes);
  } else {
    HIP_TRY(hipMemcpyAsync(out, h->nv_s_out, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return D2FE_OK;
}

int d2fe_netvlad(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, float* out) {
  return d2fe_netvlad_batch(h, gray, 1, width, height, stride, (size_t)stride * height, out);
}

}  // extern "C"
namespace d2fe {
int clone_lane(d2fe_context* p, int max_batch, d2fe_context** out, hipStream_t stream, int ncu, bool with_netvlad) {
  *out = nullptr;
  d2fe_config cfg = p->cfg;
  cfg.max_batch = max_batch;
  cfg.async_tail = 0;
  d2fe_handle c = nullptr;
  int rc = create_context(&cfg, &c, true, stream);       // on failure the stream stays the caller's (create_context un-adopts it below)
  if (rc) return rc;
  c->borrowed = true;
  if (ncu > 0) c->ncu = ncu;
  c->w1a = p->w1a; c->b1a = p->b1a;
  for (int i = 0; i < L_COUNT; ++i) c->L[i] = p->L[i];
  c->sp_loaded = p->sp_loaded;
  c->pca_comp_t = p->pca_comp_t; c->pca_mean = p->pca_mean; c->pca_dims = p->pca_dims;
  c->fuse1a = p->fuse1a; c->wino_dynamic = p->wino_dynamic; c->sp_min_batch = p->sp_min_batch;
  if (p->nv_loaded && with_netvlad) {
    const int B = max_batch;
    c->nv = p->nv;
    for (auto& l : c->nv) { l.out = nullptr; l.slabs = 1; l.slab_stride = 0; }
    c->nv_plan = p->nv_plan;
    c->nv_feat_gmax = p->nv_feat_gmax; c->nv_blocks_target = p->nv_blocks_target; c->nv_tail_blocks = p->nv_tail_blocks; c->nv_slabsum = p->nv_slabsum; c->nv_group_rule = p->nv_group_rule; c->nv_merge = p->nv_merge;
    c->nv_front_tpw = p->nv_front_tpw; c->nv_nbuf = p->nv_nbuf;
    c->nv_feat = p->nv_feat; c->nv_proj = p->nv_proj; c->nv_k = p->nv_k;
    c->nv_pre_w = p->nv_pre_w; c->nv_pre_b = p->nv_pre_b; c->nv_aw = p->nv_aw; c->nv_aw_pack = p->nv_aw_pack; c->nv_ab = p->nv_ab; c->nv_cen = p->nv_cen;
    c->nv_pca_comp = p->nv_pca_comp; c->nv_pca_mean = p->nv_pca_mean; c->nv_pca_m = p->nv_pca_m;
    auto alloc = [&]() -> int {
      for (size_t li = 0; li < c->nv.size(); ++li)
        if (p->nv[li].out) HIP_TRY(hipMalloc(&c->nv[li].out, sizeof(float) * (size_t)c->nv[li].gmax * B * c->nv[li].oh * c->nv[li].ow * c->nv[li].cout));
      const int ch = c->nv.back().oh, cw = c->nv.back().ow;
      HIP_TRY(hipMalloc(&c->nv_feat_buf, sizeof(float) * (size_t)c->nv_feat_gmax * B * ch * cw * c->nv_proj));
      HIP_TRY(hipMalloc(&c->nv_raw, sizeof(float) * (size_t)B * c->nv_k * c->nv_proj));
      HIP_TRY(hipMalloc(&c->nv_part, sizeof(float) * (size_t)B * nv_vlad_part_floats(ch * cw, c->nv_proj, c->nv_k)));
      return D2FE_OK;       // no nv_s_img / nv_s_out: the host-pointer NetVLAD calls never run on a lane
    };
    rc = alloc();
    if (rc) { if (stream) c->stream = nullptr; d2fe_destroy(c); return rc; }       // the stream stays the caller's
    c->nv_loaded = true;
  }
  *out = c;
  return D2FE_OK;
}
}  // namespace d2fe
extern "C" {

// ---- SURVEY 8(f) next rows -------------------------------------------------------------------------------------------------
struct d2fe_db {
  d2fe_context* h = nullptr;
  int dim = 0, cap = 0, ntotal = 0;
  float* vecs = nullptr;
  float* sims = nullptr;   // [maxq][cap] scratch
  float* q = nullptr;      // staged queries
  float* osims = nullptr; int32_t* olabels = nullptr;
  std::mutex mu;
};
namespace { constexpr int DB_MAXQ = 8, DB_MAXK = 1024; }

int d2fe_undistort_device(d2fe_handle h, const uint8_t* d_src, int n, int sw, int sh, int sstride, size_t src_image_stride,
                          const float* d_mapx, const float* d_mapy, const float* d_gain, int dw, int dh, uint8_t* d_dst,
                          void* stream) {
  if (!h || !d_src || !d_mapx || !d_mapy || !d_dst) return fail(D2FE_ERR_INVALID, "null argument");
  if (n < 1 || sw < 1 || sh < 1 || sstride < sw || dw < 1 || dh < 1) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_undistort(d_src, sh, sw, sstride, (long)src_image_stride, d_mapx, d_mapy, d_gain, dh, dw, n, d_dst,
                           stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

int d2fe_undistort(d2fe_handle h, const uint8_t* src, int sw, int sh, int sstride, const float* mapx, const float* mapy,
                   const float* gain, int dw, int dh, uint8_t* dst) {
  if (!h || !src || !mapx || !mapy || !dst) return fail(D2FE_ERR_INVALID, "null argument");
  if (sw < 1 || sh < 1 || sstride < sw || dw < 1 || dh < 1) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  const size_t sb = (size_t)sstride * sh, mb = sizeof(float) * (size_t)dw * dh, db = (size_t)dw * dh;
  void* raw = nullptr;
  { const int rc0 = d2fe::ctx_scratch(h, sb + 3 * mb + db + 64, &raw); if (rc0 != D2FE_OK) return rc0; }
  char* buf = static_cast<char*>(raw);
  uint8_t* d_src = reinterpret_cast<uint8_t*>(buf);
  float* d_mx = reinterpret_cast<float*>(buf + ((sb + 15) / 16) * 16);
  float* d_my = d_mx + db; float* d_g = d_my + db;
  uint8_t* d_dst = reinterpret_cast<uint8_t*>(d_g + db);
  hipStream_t s = h->stream;
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == D2FE_OK) rc = fail(D2FE_ERR_HIP, std::string(w) + ": " + hipGetErrorString(e)); };
  chk(hipMemcpyAsync(d_src, src, sb, hipMemcpyHostToDevice, s), "H2D src");
  chk(hipMemcpyAsync(d_mx, mapx, mb, hipMemcpyHostToDevice, s), "H2D mapx");
  chk(hipMemcpyAsync(d_my, mapy, mb, hipMemcpyHostToDevice, s), "H2D mapy");
  if (gain) chk(hipMemcpyAsync(d_g, gain, mb, hipMemcpyHostToDevice, s), "H2D gain");
  if (rc == D2FE_OK) chk(launch_undistort(d_src, sh, sw, sstride, 0, d_mx, d_my, gain ? d_g : nullptr, dh, dw, 1, d_dst, s), "undistort");
  if (rc == D2FE_OK) chk(hipMemcpyAsync(dst, d_dst, db, hipMemcpyDeviceToHost, s), "D2H");
  chk(hipStreamSynchronize(s), "sync");
  return rc;
}

int d2fe_prepare_gray_device(d2fe_handle h, const uint8_t* d_src, int n, int channels, int sw, int sh, int sstride,
                             size_t src_image_stride, int dw, int dh, uint8_t* d_dst, void* stream) {
  if (!h || !d_src || !d_dst) return fail(D2FE_ERR_INVALID, "null argument");
  if (n < 1 || (channels != 1 && channels != 3) || sw < 2 || sh < 2 || sstride < sw * channels || dw < 1 || dh < 1)
    return fail(D2FE_ERR_INVALID, "bad geometry (channels must be 1 or 3)");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_prep_gray(d_src, channels, sw, sh, sstride, (long)src_image_stride, n, dw, dh, d_dst, stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

int d2fe_prepare_gray(d2fe_handle h, const uint8_t* src, int channels, int sw, int sh, int sstride, int dw, int dh, uint8_t* dst) {
  if (!h || !src || !dst) return fail(D2FE_ERR_INVALID, "null argument");
  if ((channels != 1 && channels != 3) || sw < 2 || sh < 2 || sstride < sw * channels || dw < 1 || dh < 1)
    return fail(D2FE_ERR_INVALID, "bad geometry (channels must be 1 or 3)");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  const size_t sb = (size_t)sstride * sh, db = (size_t)dw * dh;
  void* raw = nullptr;
  { const int rc0 = d2fe::ctx_scratch(h, sb + db + 64, &raw); if (rc0 != D2FE_OK) return rc0; }
  uint8_t* d_src = static_cast<uint8_t*>(raw);
  uint8_t* d_dst = d_src + ((sb + 15) / 16) * 16;
  hipStream_t s = h->stream;
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == D2FE_OK) rc = fail(D2FE_ERR_HIP, std::string(w) + ": " + hipGetErrorString(e)); };
  chk(hipMemcpyAsync(d_src, src, sb, hipMemcpyHostToDevice, s), "H2D");
  if (rc == D2FE_OK) chk(launch_prep_gray(d_src, channels, sw, sh, sstride, 0, 1, dw, dh, d_dst, s), "prep_gray");
  if (rc == D2FE_OK) chk(hipMemcpyAsync(dst, d_dst, db, hipMemcpyDeviceToHost, s), "D2H");
  chk(hipStreamSynchronize(s), "sync");
  return rc;
}

static int gen_map(d2fe_handle h, const d2fe_mei_camera* cam, const double* q, int mode, int width, int height, double f,
                   float* mapx, float* mapy, bool device, void* stream) {
  if (!h || !cam || !mapx || !mapy || (mode == 1 && !q)) return fail(D2FE_ERR_INVALID, "null argument");
  if (width < 2 || height < 2 || (size_t)width * height > (1u << 28) || !(f > 0)) return fail(D2FE_ERR_INVALID, "bad map geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  const double c9[9] = {cam->xi, cam->k1, cam->k2, cam->p1, cam->p2, cam->gamma1, cam->gamma2, cam->u0, cam->v0};
  hipStream_t s = device && stream ? (hipStream_t)stream : h->stream;
  if (device) {
    HIP_TRY(launch_gen_map(c9, q, mode, width, height, f, mapx, mapy, s));
    return D2FE_OK;
  }
  const size_t n = (size_t)width * height;
  void* raw = nullptr;
  { const int rc0 = d2fe::ctx_scratch(h, sizeof(float) * 2 * n, &raw); if (rc0 != D2FE_OK) return rc0; }
  float* buf = static_cast<float*>(raw);
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == D2FE_OK) rc = fail(D2FE_ERR_HIP, std::string(w) + ": " + hipGetErrorString(e)); };
  chk(launch_gen_map(c9, q, mode, width, height, f, buf, buf + n, s), "gen_map");
  chk(hipMemcpyAsync(mapx, buf, sizeof(float) * n, hipMemcpyDeviceToHost, s), "D2H mapx");
  chk(hipMemcpyAsync(mapy, buf + n, sizeof(float) * n, hipMemcpyDeviceToHost, s), "D2H mapy");
  chk(hipStreamSynchronize(s), "sync");
  
  return rc;
}
static double cyl_focal(int width, double fov_deg) { return (double)(unsigned)width / (fov_deg * (M_PI / 180.0)); }

int d2fe_gen_cylinder_map(d2fe_handle h, const d2fe_mei_camera* cam, int width, int height, double fov_deg, float* mapx, float* mapy) {
  if (!(fov_deg > 0)) return fail(D2FE_ERR_INVALID, "fov must be positive");
  return gen_map(h, cam, nullptr, 0, width, height, cyl_focal(width, fov_deg), mapx, mapy, false, nullptr);
}
int d2fe_gen_cylinder_map_device(d2fe_handle h, const d2fe_mei_camera* cam, int width, int height, double fov_deg, float* d_mapx,
                                 float* d_mapy, void* stream) {
  if (!(fov_deg > 0)) return fail(D2FE_ERR_INVALID, "fov must be positive");
  return gen_map(h, cam, nullptr, 0, width, height, cyl_focal(width, fov_deg), d_mapx, d_mapy, true, stream);
}
int d2fe_gen_pinhole_map(d2fe_handle h, const d2fe_mei_camera* cam, const double* q_wxyz, int width, int height, double f,
                         float* mapx, float* mapy) {
  return gen_map(h, cam, q_wxyz, 1, width, height, f, mapx, mapy, false, nullptr);
}
int d2fe_gen_pinhole_map_device(d2fe_handle h, const d2fe_mei_camera* cam, const double* q_wxyz, int width, int height, double f,
                                float* d_mapx, float* d_mapy, void* stream) {
  return gen_map(h, cam, q_wxyz, 1, width, height, f, d_mapx, d_mapy, true, stream);
}

void d2fe_db_destroy(d2fe_db_handle db);
int d2fe_db_create(d2fe_handle h, int dim, int capacity, d2fe_db_handle* out) {
  if (!h || !out) return fail(D2FE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (dim < 4 || (dim & 3) || dim > 8192 || capacity < 1) return fail(D2FE_ERR_INVALID, "dim must be a multiple of 4 in 4..8192, capacity >= 1");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  d2fe_db* db = new d2fe_db();
  db->h = h; db->dim = dim; db->cap = capacity;
  const int rc = [&]() -> int {
    HIP_TRY(hipMalloc(&db->vecs, sizeof(float) * (size_t)dim * capacity));
    HIP_TRY(hipMalloc(&db->sims, sizeof(float) * (size_t)DB_MAXQ * capacity));
    HIP_TRY(hipMalloc(&db->q, sizeof(float) * (size_t)DB_MAXQ * dim));
    HIP_TRY(hipMalloc(&db->osims, sizeof(float) * DB_MAXQ * DB_MAXK));
    HIP_TRY(hipMalloc(&db->olabels, sizeof(int32_t) * DB_MAXQ * DB_MAXK));
    return D2FE_OK;
  }();
  if (rc != D2FE_OK) { d2fe_db_destroy(db); return rc; }
  *out = db;
  return D2FE_OK;
}

void d2fe_db_destroy(d2fe_db_handle db) {
  if (!db) return;
  hipSetDevice(db->h->cfg.device_id);
  for (void* p : {(void*)db->vecs, (void*)db->sims, (void*)db->q, (void*)db->osims, (void*)db->olabels}) if (p) hipFree(p);
  delete db;
}

int d2fe_db_ntotal(d2fe_db_handle db) { return db ? db->ntotal : fail(D2FE_ERR_INVALID, "null db"); }

int d2fe_db_add(d2fe_db_handle db, const float* vecs, int n) {
  if (!db || !vecs || n < 1) return fail(D2FE_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(db->mu);
  if (db->ntotal + n > db->cap) return fail(D2FE_ERR_TRUNCATED, "database capacity exceeded");
  HIP_TRY(hipSetDevice(db->h->cfg.device_id));
  HIP_TRY(hipMemcpy(db->vecs + (size_t)db->ntotal * db->dim, vecs, sizeof(float) * (size_t)n * db->dim, hipMemcpyHostToDevice));
  const int first = db->ntotal;
  db->ntotal += n;
  return first;
}

int d2fe_db_search(d2fe_db_handle db, const float* q, int nq, int k, float* sims, int32_t* labels) {
  if (!db || !q || !sims || !labels) return fail(D2FE_ERR_INVALID, "null argument");
  if (nq < 1 || nq > DB_MAXQ || k < 1 || k > DB_MAXK || (size_t)nq * db->dim * 4 > 65536) return fail(D2FE_ERR_INVALID, "nq/k out of range");
  std::lock_guard<std::mutex> lk(db->mu);
  for (int i = 0; i < nq * k; ++i) { labels[i] = -1; sims[i] = 0.f; }
  if (db->ntotal == 0) return D2FE_OK;
  HIP_TRY(hipSetDevice(db->h->cfg.device_id));
  hipStream_t s = db->h->stream;
  const int kk = k < db->ntotal ? k : db->ntotal;
  HIP_TRY(hipMemcpyAsync(db->q, q, sizeof(float) * (size_t)nq * db->dim, hipMemcpyHostToDevice, s));
  HIP_TRY(launch_db_search(db->vecs, db->ntotal, db->dim, db->q, nq, kk, db->sims, db->olabels, db->osims, s));
  std::vector<float> hs((size_t)nq * kk); std::vector<int32_t> hl((size_t)nq * kk);
  HIP_TRY(hipMemcpyAsync(hs.data(), db->osims, sizeof(float) * hs.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hl.data(), db->olabels, sizeof(int32_t) * hl.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int qi = 0; qi < nq; ++qi)
    for (int r = 0; r < kk; ++r) { sims[(size_t)qi * k + r] = hs[(size_t)qi * kk + r]; labels[(size_t)qi * k + r] = hl[(size_t)qi * kk + r]; }
  return D2FE_OK;
}

int d2fe_db_query_gated(d2fe_db_handle db, const float* q, int max_index, double thres, int32_t* label, float* sim) {
  if (!db || !q || !label || !sim || max_index < 0) return fail(D2FE_ERR_INVALID, "bad argument");
  *label = -1; *sim = 0.f;
  const int ntotal = db->ntotal;
  int k = 5 + max_index;                      // SEARCH_NEAREST_NUM, d2frontend_params.h:22
  if (k > ntotal) k = ntotal;
  if (k <= 0) return D2FE_OK;
  if (k > DB_MAXK) k = DB_MAXK;
  std::vector<float> s(k); std::vector<int32_t> l(k);
  int rc = d2fe_db_search(db, q, 1, k, s.data(), l.data());
  if (rc) return rc;
  for (int i = 0; i < k; ++i) {
    if (l[i] < 0) continue;
    if (l[i] <= ntotal - max_index && (double)s[i] > thres) { *label = l[i]; *sim = s[i]; return D2FE_OK; }
  }
  return D2FE_OK;
}

static int codec_run(d2fe_handle h, const void* in, size_t in_bytes, void* out, size_t out_bytes, int n, int arg, bool quant) {
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  void* raw = nullptr;
  { const int rc0 = d2fe::ctx_scratch(h, in_bytes + out_bytes + 64, &raw); if (rc0 != D2FE_OK) return rc0; }
  char* buf = static_cast<char*>(raw);
  char* d_out = buf + ((in_bytes + 15) / 16) * 16;
  hipStream_t s = h->stream;
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == D2FE_OK) rc = fail(D2FE_ERR_HIP, std::string(w) + ": " + hipGetErrorString(e)); };
  chk(hipMemcpyAsync(buf, in, in_bytes, hipMemcpyHostToDevice, s), "H2D");
  if (rc == D2FE_OK) {
    if (quant) chk(launch_quant_int8(reinterpret_cast<const float*>(buf), n, arg, reinterpret_cast<int8_t*>(d_out), s), "quant");
    else chk(launch_dequant_int8(reinterpret_cast<const int8_t*>(buf), n, arg, reinterpret_cast<float*>(d_out), s), "dequant");
  }
  if (rc == D2FE_OK) chk(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s), "D2H");
  chk(hipStreamSynchronize(s), "sync");
  return rc;
}

int d2fe_quantize_int8(d2fe_handle h, const float* x, int n, int double_max, int8_t* out) {
  if (!h || !x || !out || n < 1) return fail(D2FE_ERR_INVALID, "bad argument");
  return codec_run(h, x, sizeof(float) * (size_t)n, out, (size_t)n, n, double_max ? 1 : 0, true);
}

int d2fe_dequantize_int8(d2fe_handle h, const int8_t* q, int n, int landmark_num, float* out) {
  if (!h || !q || !out || n < 1) return fail(D2FE_ERR_INVALID, "bad argument");
  if (landmark_num >= 0 && (n & 31)) return fail(D2FE_ERR_INVALID, "landmark descriptors: n must be a multiple of 32");
  return codec_run(h, q, (size_t)n, out, sizeof(float) * (size_t)n, n, landmark_num, false);
}

int d2fe_block_field_offset(int cap, int netvlad_dim, int field) {
  if (cap < 1 || netvlad_dim < 0 || field < 0 || field > 4) return fail(D2FE_ERR_INVALID, "bad argument");
  const int off[5] = {0, cap * 256, cap * 258, cap * 259, cap * 259 + netvlad_dim};
  return off[field];
}
int d2fe_block_words(int cap, int netvlad_dim) {
  if (cap < 1 || netvlad_dim < 0 || (netvlad_dim & 3)) return fail(D2FE_ERR_INVALID, "bad argument");
  return (cap * 259 + netvlad_dim + 1 + 255) / 256 * 256;
}
int d2fe_pack_blocks_device(d2fe_handle h, const float* d_desc, const float* d_kps_xy, const float* d_scores, const int32_t* d_n,
                            const float* d_netvlad, int row0, int row_step, int nframes, int cap, int netvlad_dim, float* d_blocks,
                            void* stream) {
  if (!h || !d_desc || !d_kps_xy || !d_scores || !d_n || !d_blocks) return fail(D2FE_ERR_INVALID, "null argument");
  if (nframes < 1 || cap < 1 || row0 < 0 || row_step < 1 || netvlad_dim < 0 || (netvlad_dim & 3)) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_pack_blocks(d_desc, d_kps_xy, d_scores, d_n, d_netvlad, row0, row_step, nframes, cap, netvlad_dim,
                             d2fe_block_words(cap, netvlad_dim), d_blocks, stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}
int d2fe_gate_pairs_device(d2fe_handle h, const float* d_q, size_t q_stride, const float* d_db, size_t db_stride, int dim,
                           const int32_t* d_pair_q, const int32_t* d_pair_db, int npairs, double thres, int32_t* d_cnt_inout,
                           int32_t* d_pass, float* d_sims, int32_t* d_n_pass, void* stream) {
  if (!h || !d_q || !d_db || !d_pair_q || !d_pair_db) return fail(D2FE_ERR_INVALID, "null argument");
  if (npairs < 1 || dim < 4 || (dim & 3)) return fail(D2FE_ERR_INVALID, "dim must be a multiple of 4");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_gate_pairs(d_q, (long)q_stride, d_db, (long)db_stride, dim, d_pair_q, d_pair_db, npairs, thres, d_cnt_inout, d_pass,
                            d_sims, d_n_pass, stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

int d2fe_block_bytes_int8(int cap, int netvlad_dim) {
  if (cap < 1 || netvlad_dim < 0 || (netvlad_dim & 3)) return fail(D2FE_ERR_INVALID, "bad argument");
  return (cap * 256 + netvlad_dim + cap * 8 + 4 + 63) / 64 * 64;
}
int d2fe_pack_blocks_int8_device(d2fe_handle h, const float* d_desc, const float* d_kps_xy, const int32_t* d_n, const float* d_netvlad,
                                 int row0, int row_step, int nframes, int cap, int netvlad_dim, int8_t* d_blocks, void* stream) {
  if (!h || !d_desc || !d_kps_xy || !d_n || !d_blocks) return fail(D2FE_ERR_INVALID, "null argument");
  if (nframes < 1 || cap < 1 || row0 < 0 || row_step < 1 || netvlad_dim < 0 || (netvlad_dim & 3)) return fail(D2FE_ERR_INVALID, "bad geometry");
  if (((uintptr_t)d_desc & 15) || ((uintptr_t)d_blocks & 3)) return fail(D2FE_ERR_INVALID, "d_desc must be 16-byte aligned, d_blocks 4-byte aligned");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_pack_blocks_int8(d_desc, d_kps_xy, d_n, d_netvlad, row0, row_step, nframes, cap, netvlad_dim,
                                  d2fe_block_bytes_int8(cap, netvlad_dim), d_blocks, stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}
int d2fe_unpack_blocks_int8_device(d2fe_handle h, const int8_t* d_blocks_int8, int nblocks, int cap, int netvlad_dim, int renorm,
                                   float* d_blocks, void* stream) {
  if (!h || !d_blocks_int8 || !d_blocks) return fail(D2FE_ERR_INVALID, "null argument");
  if (nblocks < 1 || cap < 1 || netvlad_dim < 0 || (netvlad_dim & 3) || (renorm != 0 && renorm != 1)) return fail(D2FE_ERR_INVALID, "bad geometry");
  if (((uintptr_t)d_blocks & 15) || ((uintptr_t)d_blocks_int8 & 3)) return fail(D2FE_ERR_INVALID, "d_blocks must be 16-byte aligned, d_blocks_int8 4-byte aligned");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_unpack_blocks_int8(d_blocks_int8, nblocks, cap, netvlad_dim, d2fe_block_bytes_int8(cap, netvlad_dim),
                                    d2fe_block_words(cap, netvlad_dim), renorm, d_blocks, stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

int d2fe_quad_gate_device(d2fe_handle h, const float* d_local, size_t local_stride, const float* d_remote, size_t remote_stride, int dim,
                          const int32_t* d_job_local_row0, const int32_t* d_job_remote_row0, int local_view_step, int remote_view_step,
                          int njobs, double thres, int32_t* d_dir_prev, float* d_sims, int32_t* d_cnt_inout, int32_t* d_n_pass,
                          void* stream) {
  if (!h || !d_local || !d_remote || !d_job_local_row0 || !d_job_remote_row0) return fail(D2FE_ERR_INVALID, "null argument");
  if (njobs < 1 || dim < 4 || (dim & 3) || local_view_step < 1 || remote_view_step < 1) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_quad_gate(d_local, (long)local_stride, d_remote, (long)remote_stride, dim, d_job_local_row0, d_job_remote_row0,
                           local_view_step, remote_view_step, njobs, thres, d_dir_prev, d_sims, d_cnt_inout, d_n_pass,
                           stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

float d2fe_half_move_cols(int width_undistort, double undistort_fov) { return (float)((double)width_undistort * 90.0 / undistort_fov); }
int d2fe_half_image_compact_device(d2fe_handle h, const float* d_desc, const float* d_pts_xy, const int32_t* d_n, const int32_t* d_job_row,
                                   const int32_t* d_job_left, const float* d_job_shift_x, int njobs, int cap, int dim, int width_undistort,
                                   double undistort_fov, float* d_out_desc, float* d_out_pts, int32_t* d_out_map, int32_t* d_out_n,
                                   void* stream) {
  if (!h || !d_desc || !d_pts_xy || !d_n || !d_job_row || !d_job_left || !d_job_shift_x || !d_out_desc || !d_out_pts || !d_out_map || !d_out_n)
    return fail(D2FE_ERR_INVALID, "null argument");
  if (njobs < 1 || cap < 1 || cap > 1024 || dim < 4 || (dim & 3) || width_undistort < 1 || !(undistort_fov > 0)) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_half_compact(d_desc, d_pts_xy, d_n, d_job_row, d_job_left, d_job_shift_x, njobs, cap, dim, (float)width_undistort,
                              d2fe_half_move_cols(width_undistort, undistort_fov), d_out_desc, d_out_pts, d_out_map, d_out_n,
                              stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}
int d2fe_remap_matches_device(d2fe_handle h, int32_t* d_q_idx, int32_t* d_t_idx, const int32_t* d_n_match, const int32_t* d_map_a_job,
                              const int32_t* d_map_b_job, const int32_t* d_maps, int npairs, int cap_match, int cap_map, void* stream) {
  if (!h || !d_q_idx || !d_t_idx || !d_n_match || !d_map_a_job || !d_map_b_job || !d_maps) return fail(D2FE_ERR_INVALID, "null argument");
  if (npairs < 1 || cap_match < 1 || cap_map < 1) return fail(D2FE_ERR_INVALID, "bad geometry");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(launch_remap_matches(d_q_idx, d_t_idx, d_n_match, d_map_a_job, d_map_b_job, d_maps, npairs, cap_match, cap_map,
                               stream ? (hipStream_t)stream : h->stream));
  return D2FE_OK;
}

int d2fe_match_batch_device(d2fe_handle h, const d2fe_match_batch* mb, void* stream) {
  if (!h || !mb) return fail(D2FE_ERR_INVALID, "null argument");
  if (mb->npairs < 1 || mb->max_n < 1 || mb->max_n > 16384) return fail(D2FE_ERR_INVALID, "npairs/max_n out of range (max_n <= 16384)");
  if (mb->dim < 4 || mb->dim > 256 || (mb->dim & 3)) return fail(D2FE_ERR_INVALID, "dim must be a multiple of 4 in 4..256");
  if (mb->mode != 0 && mb->mode != 1) return fail(D2FE_ERR_INVALID, "bad mode");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  std::lock_guard<std::mutex> lk(h->match_mu);
  // the candidate scratch is keyed by the caller's stream: two calls on different streams (main and tail, two threads) can
  // overlap on the GPU and must not share cand4[pair][dir][row]; calls on one stream are ordered by the stream itself
  d2fe_context::MatchScratch* sc = nullptr;
  for (auto& e : h->m_scratch) if (e.stream == s) { sc = &e; break; }
  if (!sc) { h->m_scratch.emplace_back(); sc = &h->m_scratch.back(); sc->stream = s; }
  // [arrival tickets | records]; the tickets are zero between launches (the kernel's last workgroup per pair resets its own), so a scratch
  // sized for more pairs serves any smaller batch
  const int tp = mb->npairs > sc->npairs ? mb->npairs : sc->npairs;
  const size_t need = match_scratch_bytes(tp, 0) + sizeof(int32_t) * 8 * (size_t)mb->max_n * mb->npairs;
  if (need > sc->bytes || mb->npairs > sc->npairs) {
    HIP_TRY(hipStreamSynchronize(s));      // the only work that can still read the old scratch is on this stream
    if (sc->cand4) hipFree(sc->cand4);
    sc->cand4 = nullptr; sc->bytes = 0; sc->npairs = 0;
    const size_t want = need > sc->bytes ? need : sc->bytes;
    HIP_TRY(hipMalloc(&sc->cand4, want));
    HIP_TRY(hipMemsetAsync(sc->cand4, 0, want, s));      // ON the caller's stream: hipMemset runs on the null stream, which a non-blocking stream does not wait for
    sc->bytes = want; sc->npairs = tp;
  }
  MatchArgs m;
  m.a = mb->d_a; m.b = mb->d_b; m.pts_a = mb->d_pts_a; m.pts_b = mb->d_pts_b;
  m.a_off = mb->d_a_off; m.b_off = mb->d_b_off; m.a_cnt = mb->d_a_cnt; m.b_cnt = mb->d_b_cnt;
  m.npairs = mb->npairs; m.dim = mb->dim; m.max_n = mb->max_n; m.mode = mb->mode;
  m.ratio = mb->ratio; m.radius = mb->radius;
  m.q_idx = mb->d_q_idx; m.t_idx = mb->d_t_idx; m.dist = mb->d_dist; m.n_out = mb->d_n_out;
  match_scratch_carve(sc->cand4, sc->npairs, &m); m.stats = h->match_stats; m.ncu = h->ncu;
#ifdef D2FE_DEVTOOLS
  if (d2fe_dev_env("D2FE_MATCH_STAMPS", 0) && (long)((mb->max_n + 31) / 32) * 2 * mb->npairs <= 4096) {
    if (!h->match_stamps) HIP_TRY(hipMalloc(&h->match_stamps, sizeof(unsigned long long) * 16 * 4096));
    HIP_TRY(hipMemsetAsync(h->match_stamps, 0, sizeof(unsigned long long) * 16 * 4096, s));
    m.stamps = h->match_stamps;
  }
#endif
  { ProfScope ps(h, D2FE_PROF_MATCH, s); HIP_TRY(launch_match(m, s)); }
  return D2FE_OK;
}

static int match_host(d2fe_handle h, int mode, const float* a, int na, const float* b, int nb, int dim, double ratio,
                      const float* pts_a, const float* pts_b, double radius, int32_t* q_idx, int32_t* t_idx, float* dist,
                      int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!h || !n_out) return fail(D2FE_ERR_INVALID, "null argument");
  if (na < 0 || nb < 0 || cap < 0) return fail(D2FE_ERR_INVALID, "negative size");
  if (na == 0 || nb == 0) return D2FE_OK;
  if (!a || !b || !q_idx || !t_idx || !dist) return fail(D2FE_ERR_INVALID, "null pointer");
  const int max_n = na > nb ? na : nb;
  if (max_n > 16384) return fail(D2FE_ERR_UNSUPPORTED, "more than 16384 descriptors per side");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  const bool use_pts = mode == 0 && radius > 0 && pts_a && pts_b;
  const size_t fa = (size_t)na * dim, fb = (size_t)nb * dim;
  if (dim < 4 || dim > 256 || (dim & 3)) return fail(D2FE_ERR_INVALID, "dim must be a multiple of 4 in 4..256");
  // a private (stream, scratch) slot per call in flight: re-entrant; slots are created on demand and reused
  // a slot's buffers are sized for 1024 rows per side at first (every shipped configuration) and grow with the calls
  const size_t rows = (size_t)(max_n > 1024 ? max_n : 1024);
  const size_t slot_bytes = sizeof(float) * (2 * rows * 256 + 4 * rows + rows) + sizeof(int32_t) * (2 * rows + 16 + 8 * rows) + 128;
  int slot = -1;
  hipStream_t s = nullptr;
  char* buf = nullptr;
  char* pin = nullptr;
  d2fe_context::MatchSlot* msp = nullptr;       // taken under the lock: a deque never relocates its elements, but indexing it races with an append
  {
    std::lock_guard<std::mutex> lk(h->match_mu);
    for (size_t i = 0; i < h->match_slots.size(); ++i)
      if (!h->match_slots[i].busy) { slot = (int)i; break; }
    if (slot < 0) {
      d2fe_context::MatchSlot ms;
      if (hipStreamCreateWithFlags(&ms.stream, hipStreamNonBlocking) != hipSuccess) return fail(D2FE_ERR_HIP, "hipStreamCreate (match slot)");
      h->match_slots.push_back(ms);
      slot = (int)h->match_slots.size() - 1;
    }
    d2fe_context::MatchSlot& ms = h->match_slots[slot];
    ms.busy = true;      // the slot is this call's from here on: the (re)allocation below happens OUTSIDE the lock (hipFree / hipHostFree synchronise the device)
    s = ms.stream;      // read under the lock: another thread may be appending a slot right now
    msp = &ms;
  }
  struct Release { d2fe_context* h; int slot; ~Release() { std::lock_guard<std::mutex> lk(h->match_mu); h->match_slots[slot].busy = false; } } release{h, slot};
  {
    d2fe_context::MatchSlot& ms = *msp;
    if (ms.bytes < slot_bytes) {
      if (ms.buf) { hipFree(ms.buf); ms.buf = nullptr; }
      if (ms.pin) { (void)hipHostFree(ms.pin); ms.pin = nullptr; }
      ms.bytes = 0;
      if (hipMalloc(&ms.buf, slot_bytes) != hipSuccess) return fail(D2FE_ERR_HIP, "hipMalloc match scratch");
      if (h->use_pinned && hipHostMalloc(&ms.pin, slot_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ms.pin = nullptr; }
      ms.bytes = slot_bytes;
    }
    buf = ms.buf;
    pin = ms.pin;
  }
  // device layout (words): a | b | pts_a | pts_b | meta {a_off, b_off, a_cnt, b_cnt, arrival ticket (0), pad x3} | n_out, pad x7 | dist | q | t | cand4
  // -- the inputs are one contiguous run (ONE H2D from the slot's pinned mirror), the outputs another (ONE D2H, one synchronisation)
  const size_t w_in = fa + fb + (use_pts ? 2 * (size_t)(na + nb) : 0) + 8;
  float* d_a = reinterpret_cast<float*>(buf);
  float* d_b = d_a + fa;
  float* d_pa = d_b + fb;
  float* d_pb = d_pa + (use_pts ? 2 * (size_t)na : 0);
  int32_t* d_meta = reinterpret_cast<int32_t*>(d_pb + (use_pts ? 2 * (size_t)nb : 0));
  int32_t* d_nout = d_meta + 8;
  float* d_dist = reinterpret_cast<float*>(d_nout + 8);
  int32_t* d_q = reinterpret_cast<int32_t*>(d_dist + max_n);
  int32_t* d_t = d_q + max_n;
  int32_t* d_c4 = d_t + max_n;
  const size_t w_out = 8 + 3 * (size_t)max_n;
  const int32_t meta[8] = {0, 0, na, nb, 0, 0, 0, 0};
  int rc = D2FE_OK;
  auto chk = [&](hipError_t er, const char* what) { if (er != hipSuccess && rc == D2FE_OK) rc = fail(D2FE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(er)); };
  if (pin) {
    float* p = reinterpret_cast<float*>(pin);
    memcpy(p, a, sizeof(float) * fa); memcpy(p + fa, b, sizeof(float) * fb);
    float* pm = p + fa + fb;
    if (use_pts) { memcpy(pm, pts_a, sizeof(float) * 2 * na); memcpy(pm + 2 * (size_t)na, pts_b, sizeof(float) * 2 * nb); pm += 2 * (size_t)(na + nb); }
    memcpy(pm, meta, sizeof(meta));
    chk(hipMemcpyAsync(d_a, pin, sizeof(float) * w_in, hipMemcpyHostToDevice, s), "H2D inputs");
  } else {
    chk(hipMemcpyAsync(d_a, a, sizeof(float) * fa, hipMemcpyHostToDevice, s), "H2D a");
    chk(hipMemcpyAsync(d_b, b, sizeof(float) * fb, hipMemcpyHostToDevice, s), "H2D b");
    if (use_pts) {
      chk(hipMemcpyAsync(d_pa, pts_a, sizeof(float) * 2 * na, hipMemcpyHostToDevice, s), "H2D pts_a");
      chk(hipMemcpyAsync(d_pb, pts_b, sizeof(float) * 2 * nb, hipMemcpyHostToDevice, s), "H2D pts_b");
    }
    chk(hipMemcpyAsync(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, s), "H2D meta");
  }
  MatchArgs m;
  m.a = d_a; m.b = d_b; m.pts_a = use_pts ? d_pa : nullptr; m.pts_b = use_pts ? d_pb : nullptr;
  m.a_off = d_meta; m.b_off = d_meta + 1; m.a_cnt = d_meta + 2; m.b_cnt = d_meta + 3;
  m.npairs = 1; m.dim = dim; m.max_n = max_n; m.mode = mode; m.ratio = ratio; m.radius = use_pts ? radius : -1.0;
  m.q_idx = d_q; m.t_idx = d_t; m.dist = d_dist; m.n_out = d_nout; m.cand4 = d_c4; m.stats = h->match_stats;
  m.ncu = h->ncu;
  m.ticket = d_meta + 4;      // a zero word of the meta block: uploaded with the inputs, so zero at every launch
  if (rc == D2FE_OK) chk(launch_match(m, s), "launch_match");
  if (pin) {
    // pinned mirror of the output run, behind the inputs' mirror
    int32_t* po = reinterpret_cast<int32_t*>(pin) + ((w_in + 15) & ~(size_t)15);
    if (rc == D2FE_OK) chk(hipMemcpyAsync(po, d_nout, sizeof(int32_t) * w_out, hipMemcpyDeviceToHost, s), "D2H results");
    if (rc == D2FE_OK) chk(hipStreamSynchronize(s), "sync");
    if (rc == D2FE_OK) {
      const int cnt = po[0];
      const int k = cnt < cap ? cnt : cap;
      if (k > 0) {
        memcpy(dist, po + 8, sizeof(float) * k);
        memcpy(q_idx, po + 8 + max_n, sizeof(int32_t) * k);
        memcpy(t_idx, po + 8 + 2 * (size_t)max_n, sizeof(int32_t) * k);
      }
      *n_out = k;
      if (cnt > cap) rc = fail(D2FE_ERR_TRUNCATED, "match output capacity too small");
    }
    return rc;
  }
  int32_t cnt = 0;
  if (rc == D2FE_OK) chk(hipMemcpyAsync(&cnt, d_nout, sizeof(int32_t), hipMemcpyDeviceToHost, s), "D2H n");
  if (rc == D2FE_OK) chk(hipStreamSynchronize(s), "sync");
  if (rc == D2FE_OK) {
    const int k = cnt < cap ? cnt : cap;
    if (k > 0) {
      chk(hipMemcpyAsync(q_idx, d_q, sizeof(int32_t) * k, hipMemcpyDeviceToHost, s), "D2H q");
      chk(hipMemcpyAsync(t_idx, d_t, sizeof(int32_t) * k, hipMemcpyDeviceToHost, s), "D2H t");
      chk(hipMemcpyAsync(dist, d_dist, sizeof(float) * k, hipMemcpyDeviceToHost, s), "D2H d");
      chk(hipStreamSynchronize(s), "sync2");
    }
    *n_out = k;
    if (rc == D2FE_OK && cnt > cap) rc = fail(D2FE_ERR_TRUNCATED, "match output capacity too small");
  }
  return rc;
}

int d2fe_match_knn(d2fe_handle h, const float* a, int na, const float* b, int nb, int dim, double ratio,
                   const float* pts_a, const float* pts_b, double radius, int32_t* q_idx, int32_t* t_idx, float* dist,
                   int cap, int* n_out) {
  return match_host(h, 0, a, na, b, nb, dim, ratio, pts_a, pts_b, radius, q_idx, t_idx, dist, cap, n_out);
}

int d2fe_match_crosscheck(d2fe_handle h, const float* a, int na, const float* b, int nb, int dim, int32_t* q_idx,
                          int32_t* t_idx, float* dist, int cap, int* n_out) {
  return match_host(h, 1, a, na, b, nb, dim, 0.0, nullptr, nullptr, -1.0, q_idx, t_idx, dist, cap, n_out);
}

int d2fe_half_image_filter(const float* pts_xy, int n, int require_left, int width_undistort, double undistort_fov,
                           int32_t* map, int* n_out) {
  if (!n_out) return fail(D2FE_ERR_INVALID, "null argument");
  *n_out = 0;
  if (n < 0 || (n > 0 && (!pts_xy || !map))) return fail(D2FE_ERR_INVALID, "null pointer");
  // host bookkeeping on <= N points (d2featuretracker.cpp:1058-1071); float move_cols as in the reference
  const float move_cols = (float)((double)width_undistort * 90.0 / undistort_fov);
  int c = 0;
  for (int i = 0; i < n; ++i) {
    const float x = pts_xy[2 * i];
    if ((require_left && x < (float)width_undistort - move_cols) || (!require_left && x >= move_cols)) map[c++] = i;
  }
  *n_out = c;
  return D2FE_OK;
}

#ifdef D2FE_DEVTOOLS      /* development library only: include/d2fe_debug.h */
long d2fe_debug_read(d2fe_handle h, const char* name, void* dst, size_t max_bytes) {
  if (!h || !name || !dst) return fail(D2FE_ERR_INVALID, "null argument");
  if (h->last_n == 0) return fail(D2FE_ERR_NOT_READY, "no extract call yet");
  hipSetDevice(h->cfg.device_id);
  const size_t H = h->last_h, W = h->last_w, n = h->last_n;
  struct { const char* nm; Tensor* t; size_t per; } tab[] = {
      // pooled sizes floor (sizes need not be multiples of 8)
      {"conv1a", &h->a1a, H * W * 64},           {"conv1b", &h->a1b, (H / 2) * (W / 2) * 64},   {"conv2a", &h->a2a, (H / 2) * (W / 2) * 64},
      {"conv2b", &h->a2b, (H / 4) * (W / 4) * 64},   {"conv3a", &h->a3a, (H / 4) * (W / 4) * 128},  {"conv3b", &h->a3b, (H / 8) * (W / 8) * 128},
      {"conv4a", &h->a4a, (H / 8) * (W / 8) * 128},  {"conv4b", h->last_set ? &h->a4b2 : &h->a4b, (H / 8) * (W / 8) * 128},  {"convPaDa", &h->aPD, (H / 8) * (W / 8) * 512},
      {"logits", h->last_set ? &h->logits2 : &h->logits, (H / 8) * (W / 8) * 65}, {"desc_raw", h->last_set ? &h->draw2 : &h->draw, (H / 8) * (W / 8) * 256}, {"semi", &h->semi, (H / 8) * (W / 8) * 64}};
  if (!strcmp(name, "conv1a") && h->fuse1a) {
    // fused mode never materialises conv1a: evaluate it on demand from the last input frame(s)
    if (!h->a1a.p && alloc_f(h->a1a, (size_t)h->cfg.max_height * h->cfg.max_width * 64, h->cfg.max_batch) != 0)
      return fail(D2FE_ERR_HIP, "hipMalloc conv1a debug buffer");
    if (launch_conv1a(h->last_gray, h->last_stride, (long)h->last_istride, (int)H, (int)W, (int)n, h->w1a, h->b1a, h->a1a.p, h->stream) != hipSuccess)
      return fail(D2FE_ERR_HIP, "conv1a debug launch");
  }
  if (!strcmp(name, "semi") && !h->cfg.keep_score_map) return fail(D2FE_ERR_NOT_READY, "score map not kept (set keep_score_map)");
  if (h->sparse_desc && h->last_n >= h->sp_min_batch && (!strcmp(name, "desc_raw") || !strcmp(name, "convPaDa")))
    return fail(D2FE_ERR_NOT_READY, "the dense descriptor map does not exist with the sparse descriptor head (set dense_descriptors)");
  for (auto& e : tab)
    if (!strcmp(e.nm, name)) {
      const size_t bytes = e.per * n * sizeof(float);
      if (bytes > max_bytes) return fail(D2FE_ERR_TRUNCATED, "destination too small");
      if (hipStreamSynchronize(h->stream) != hipSuccess || (h->tail_stream && hipStreamSynchronize(h->tail_stream) != hipSuccess)) return fail(D2FE_ERR_HIP, "sync");
      if (hipMemcpy(dst, e.t->p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(D2FE_ERR_HIP, "D2H");
      return (long)bytes;
    }
  return fail(D2FE_ERR_INVALID, "unknown tensor name");
}

// Host-side half of the Winograd mode, callable without a GPU: U = G g G^T in the kernels' fragment order (see pack_weights_wino).
long d2fe_debug_pack_wino(const float* weight, int cout, int cin, float* out, long max_floats) {
  if (!weight || !out || cout < 1 || cin < 8 || (cin & 7)) return fail(D2FE_ERR_INVALID, "bad argument");
  const int cout_pad = (cout + 63) / 64 * 64;
  const size_t n = packed_weight_floats_wino(cout_pad, cin);
  if ((long)n > max_floats) return fail(D2FE_ERR_TRUNCATED, "destination too small");
  pack_weights_wino(weight, cout, cin, cout_pad, out);
  return (long)n;
}

// Tile shape the NetVLAD block launchers pick for an Ho x Wo output map (no GPU needed; tests/test_netvlad_pack_cpu.py sweeps it against the kernels' limits):
// kind 0 nv_pblock_kernel (stride 1), kind 1 nv_fpair_kernel (first block; c0_stride = stride of the first conv), kind 2 nv_xblock_kernel (stride = 1 or 2).
int d2fe_debug_netvlad_tile(int kind, int Ho, int Wo, int stride, int* th, int* tw) {
  if (!th || !tw || Ho < 1 || Wo < 1 || stride < 1 || stride > 2) return fail(D2FE_ERR_INVALID, "bad argument");
  if (kind == 0) nv_pblock_tile(Ho, Wo, th, tw, 0);
  else if (kind == 1) nv_pblock_tile(Ho, Wo, th, tw, stride);
  else if (kind == 2) nv_xblock_tile(Ho, Wo, stride, th, tw);
  else return fail(D2FE_ERR_INVALID, "kind");
  return D2FE_OK;
}

// Host-side weight packing of the NetVLAD block kernels, callable without a GPU (tests/test_netvlad_pack_cpu.py):
//   kind 0 expand record of nv_pblock_kernel (pack_nv_expand_pair)      kind 1 depthwise + project record of nv_pblock / nv_fpair (pack_nv_dwproj_pair)
//   kind 2 expand record of nv_xblock_kernel (pack_nv_expand_perm)      kind 3 depthwise + project record of nv_xblock_kernel (pack_nv_dwproj_x)
//   kind 4 expand record of nv_tail_kernel (pack_nv_expand_tail)        kind 5 project record of nv_tail_kernel (pack_nv_proj_t)
// we [chid][cin], be [chid], wd [chid][9], bd [chid], wp [cout][chid]; returns the number of floats written or <0.
long d2fe_debug_pack_netvlad(int kind, const float* we, const float* be, const float* wd, const float* bd, const float* wp, int cin, int chid,
                             int cout, float* out, long max_floats) {
  if (!out || cin < 8 || (cin & 7) || chid < 16 || (chid & 15) || cout < 1) return fail(D2FE_ERR_INVALID, "bad argument");
  const int nt = kind <= 1 ? nv_pblock_ntiles(cout) : nv_block_ntiles(cout);
  if (nt < 0) return fail(D2FE_ERR_UNSUPPORTED, "cout");
  size_t n = 0;
  switch (kind) {
    case 0: n = pack_nv_expand_pair_floats(chid, cin); break;
    case 1: n = pack_nv_dwproj_pair_floats(chid, nt); break;
    case 2: if (!nv_xblock_supported(cin, chid, cout, 1)) return fail(D2FE_ERR_UNSUPPORTED, "shape"); n = pack_nv_expand_perm_floats(chid, cin); break;
    case 3: case 5: n = pack_nv_dwproj_floats(chid, nt); break;
    case 4: if (!nv_tail_supported(cin, cout)) return fail(D2FE_ERR_UNSUPPORTED, "shape"); n = pack_nv_expand_floats(chid, cin); break;
    default: return fail(D2FE_ERR_INVALID, "kind");
  }
  if ((long)n > max_floats) return fail(D2FE_ERR_TRUNCATED, "destination too small");
  if (((kind == 0 || kind == 2 || kind == 4) && (!we || !be)) || ((kind == 1 || kind == 3) && (!wd || !bd || !wp)) || (kind == 5 && !wp))
    return fail(D2FE_ERR_INVALID, "null weights");
  switch (kind) {
    case 0: pack_nv_expand_pair(we, be, chid, cin, out); break;
    case 1: pack_nv_dwproj_pair(wd, bd, wp, cout, chid, nt, out); break;
    case 2: pack_nv_expand_perm(we, be, chid, cin, out); break;
    case 3: pack_nv_dwproj_x(wd, bd, wp, cout, chid, nt, out); break;
    case 4: pack_nv_expand_tail(we, be, chid, cin, out); break;
    case 5: pack_nv_proj_t(wp, cout, chid, nt, out); break;
  }
  return (long)n;
}

// One 3x3 layer through the Winograd kernels, host buffers in and out (layer-level parity tests and timing; not a product path).
int d2fe_debug_conv3x3_wino(d2fe_handle h, const float* in, int n, int H, int W, int cin, const float* weight, const float* bias,
                            int cout, int pool, int relu, float* out, int iters, float* ms_per_launch) {
  if (!h || !in || !weight || !bias || !out) return fail(D2FE_ERR_INVALID, "null argument");
  if ((cin != 64 && cin != 128) || cout < 1 || n < 1 || H < 2 || W < 2 || (pool && ((H | W) & 1)) || !relu)
    return fail(D2FE_ERR_INVALID, "unsupported layer shape");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  const int cout_pad = (cout + 63) / 64 * 64;
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const size_t in_fl = (size_t)n * H * W * cin, out_fl = (size_t)n * Ho * Wo * cout;
  if ((size_t)H * W * cin * 4 >= (1ull << 31)) return fail(D2FE_ERR_INVALID, "image too large");
  std::vector<float> pk(packed_weight_floats_wino(cout_pad, cin)), bp(cout_pad, 0.f);
  pack_weights_wino(weight, cout, cin, cout_pad, pk.data());
  memcpy(bp.data(), bias, sizeof(float) * cout);
  auto launch = [&](const ConvArgs& ca) { return launch_conv_wino(cin, pool != 0, relu != 0, cout_pad, ca, h->stream); };
  float *d_in = nullptr, *d_out = nullptr, *d_w = nullptr, *d_b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = [&]() -> int {
    HIP_TRY(hipMalloc(&d_in, in_fl * 4));
    HIP_TRY(hipMalloc(&d_out, out_fl * 4));
    HIP_TRY(hipMalloc(&d_w, pk.size() * 4));
    HIP_TRY(hipMalloc(&d_b, bp.size() * 4));
    HIP_TRY(hipMemcpy(d_in, in, in_fl * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_out, 0xff, out_fl * 4));
    HIP_TRY(hipDeviceSynchronize());      // the null-stream memset is not ordered with the handle's non-blocking stream
    ConvArgs a{};
    a.in = d_in; a.in_cstride = cin; a.in_coff = 0; a.out = d_out; a.out_cstride = cout; a.out_coff = 0;
    a.cout_real = cout; a.wpack = d_w; a.bias = d_b; a.H = H; a.W = W; a.n_img = n;
    a.in_img_stride = (long)H * W * cin; a.out_img_stride = (long)Ho * Wo * cout; a.zeros = h->zeros; a.ncu = h->ncu;
    a.ablate = d2fe_dev_env("D2FE_ABLATE", 0);
    HIP_TRY(launch(a));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, d_out, out_fl * 4, hipMemcpyDeviceToHost));
    if (iters > 0 && ms_per_launch) {
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, h->stream));
      for (int i = 0; i < iters; ++i) HIP_TRY(launch(a));
      HIP_TRY(hipEventRecord(e1, h->stream));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      *ms_per_launch = ms / iters;
    }
    return D2FE_OK;
  }();
  for (void* p : {(void*)d_in, (void*)d_out, (void*)d_w, (void*)d_b}) if (p) hipFree(p);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return rc;
}

#endif  // D2FE_DEVTOOLS

int d2fe_profile_enable(d2fe_handle h, int mode) {
  if (!h || mode < 0 || mode > 2) return fail(D2FE_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipDeviceSynchronize());
  h->prof_mode = mode;
  h->prof_used = 0;
  h->prof_recs.clear();
  return D2FE_OK;
}

int d2fe_profile_read(d2fe_handle h, float* ms, int32_t* launches) {
  if (!h || !ms || !launches) return fail(D2FE_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < D2FE_PROF_COUNT; ++i) { ms[i] = 0.f; launches[i] = 0; }
  for (auto& r : h->prof_recs) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.stage] += t;
    launches[r.stage] += 1;
  }
  h->prof_used = 0;
  h->prof_recs.clear();
  return D2FE_OK;
}

#ifdef D2FE_DEVTOOLS
int d2fe_debug_graph_count(d2fe_handle h, int* rejected) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  int n = 0, bad = 0;
  for (auto& kv : h->graphs) { n += kv.second.exec != nullptr; bad += kv.second.bad; }
  if (rejected) *rejected = bad;
  return n;
}

#endif

#ifdef D2FE_DEVTOOLS
/* development builds: the wall_clock64() phase stamps [workgroup][16] of the last d2fe_match_batch_device launch made with D2FE_MATCH_STAMPS set */
D2FE_API long d2fe_debug_match_stamps(d2fe_handle h, unsigned long long* dst, long max_wgs) {
  if (!h || !dst || !h->match_stamps) return fail(D2FE_ERR_NOT_READY, "no stamped launch yet (D2FE_MATCH_STAMPS)");
  hipSetDevice(h->cfg.device_id);
  const long nw = max_wgs < 4096 ? max_wgs : 4096;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(dst, h->match_stamps, sizeof(unsigned long long) * 16 * nw, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(D2FE_ERR_HIP, "D2H");
  return nw;
}
#endif

long d2fe_match_fallback_rows(d2fe_handle h, int reset, long* full_scans) {
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipDeviceSynchronize());
  int32_t v[2] = {0, 0};
  HIP_TRY(hipMemcpy(v, h->match_stats, sizeof(v), hipMemcpyDeviceToHost));
  if (reset) { HIP_TRY(hipMemset(h->match_stats, 0, sizeof(v))); HIP_TRY(hipDeviceSynchronize()); }
  if (full_scans) *full_scans = v[0];
  return v[1];
}

int d2fe_sync(d2fe_handle h) {
  if (h && h->tail_stream) (void)hipStreamSynchronize(h->tail_stream);
  if (!h) return fail(D2FE_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return D2FE_OK;
}

}  // extern "C"

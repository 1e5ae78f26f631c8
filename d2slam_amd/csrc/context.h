// context.h -- the device context behind d2fe_handle and the launch sequences shared by the translation units that implement
// include/d2fe.h (api.hip: the entry points; pipe.hip: the frames-in-flight pipeline).  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <array>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/d2fe.h"
#include "kernels.h"

namespace d2fe {
int ctx_fail(int code, const std::string& msg);
}

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return d2fe::ctx_fail(D2FE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + __FILE__ + ":" + \
                                              std::to_string(__LINE__));                                    \
  } while (0)

namespace d2fe {

struct Layer {
  void* wpack = nullptr;
  float* bias = nullptr;
  int cout = 0, cout_pad = 0, cin = 0, ks = 0;
};

enum { L_1B = 0, L_2A, L_2B, L_3A, L_3B, L_4A, L_4B, L_PADA, L_PB, L_DB, L_PA, L_DA32, L_DB32, L_COUNT };   // the last three: sparse descriptor head

struct Tensor {
  float* p = nullptr;
  size_t per_img = 0;  // floats per image at max size
};

}  // namespace d2fe

struct d2fe_context {
  d2fe_config cfg;
  hipStream_t stream = nullptr;
  bool sp_loaded = false;
  bool borrowed = false;       // a pipeline lane (clone_lane): the packed weights belong to the parent context
  std::atomic<bool> doomed{false};    // d2fe_destroy was called while pipes were alive: the last d2fe_pipe_destroy releases the handle
  std::atomic<int> live_pipes{0};   // pipes created from this handle and not destroyed yet: their lanes read THIS handle's packed weights, so d2fe_destroy,
                               // d2fe_load_* and d2fe_set_*_pca refuse (D2FE_ERR_INVALID) while it is non-zero
  float* w1a = nullptr;  // [9][64]
  float* b1a = nullptr;
  d2fe::Layer L[d2fe::L_COUNT];
  // activations (NHWC fp32), separate buffer per layer so that d2fe_debug_read can inspect any of them
  d2fe::Tensor a1a, a1b, a2a, a2b, a3a, a3b, a4a, a4b, aPD, logits, draw, semi;
  // async_tail: the post-processing (softmax .. descriptors) of call k runs on tail_stream under the convolutions of call k+1,
  // so the three tensors the tail reads exist twice (buffer set = call parity) and events order trunk / tail / reuse
  d2fe::Tensor a4b2, logits2, draw2;
  hipStream_t tail_stream = nullptr;
  hipEvent_t ev_trunk[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
  int parity = 0, last_set = 0;
  unsigned long long* cand = nullptr;
  int* cand_count = nullptr;
  long cand_cap = 0;
  // staging for the host-pointer API
  uint8_t* s_img = nullptr;
  int s_cap = 0;
  // last call geometry (for debug reads)
  int last_w = 0, last_h = 0, last_n = 0;
  const uint8_t* last_gray = nullptr; int last_stride = 0; size_t last_istride = 0;
  float* aconf = nullptr; int* clist = nullptr; int* a_ncand = nullptr;     // variant A scratch
  float* zeros = nullptr;      // 1 KiB of zeros (ConvArgs::zeros)
  // host-pointer calls: pinned host staging (one DMA in, one DMA out per call) and cached hipGraphs of the launch sequences.
  // The reference calls infer / inference with ONE image at 15-30 Hz (loop_cam.cpp:609-616): at that batch the ~10 us the command
  // processor spends between two dependent launches and the per-copy latency of pageable D2H copies are a third of a call.
  uint8_t* pin_in = nullptr; size_t pin_in_bytes = 0;
  float* pin_out = nullptr; size_t pin_out_bytes = 0;
  float* s_out = nullptr;      // device: [kps | scores | desc | n | idx] of a host-pointer extract call, contiguous -> ONE D2H of the first four
  size_t s_out_bytes = 0;
  bool use_graphs = true, use_pinned = true;       // D2FE_GRAPH=0 / D2FE_PINNED=0 switch them off (A/B measurements)
  // d2fe_extract_all*: NetVLAD of the same uploaded frame(s) on a second stream, beside SuperPoint
  hipStream_t nv_stream = nullptr; hipEvent_t ev_up = nullptr; float* pin_nv = nullptr; size_t pin_nv_bytes = 0;
  // host-side bookkeeping a launch sequence leaves behind (what the debug reads and the next NetVLAD step consult): saved when a sequence is
  // captured, restored on every replay, so that a replay leaves the handle exactly as a direct run of the same geometry would
  struct HostState { int last_w = 0, last_h = 0, last_n = 0, last_set = 0; const uint8_t* last_gray = nullptr; int last_stride = 0; size_t last_istride = 0;
                     std::vector<std::pair<int, long>> nv_slabs; int nv_feat_slabs = 1; long nv_feat_slab_stride = 0; int nv_stamp_wgs = 0; };
  struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; bool bad = false; HostState st; };
  std::map<std::array<long, 6>, GraphEntry> graphs;
  int ncu = 256;               // compute units this context sizes its persistent grids for: the device's (d2fe_create) or a pipeline lane's share
  int ncu_dev = 256;           // compute units of cfg.device_id: decisions that fix an arithmetic order use this one
  unsigned long long* match_stamps = nullptr;   // development builds: [4096][16] phase stamps of the last d2fe_match_batch_device launch
  int32_t* match_stats = nullptr;   // [4] matcher counters: [0] queries that took the exact fallback scan (MatchArgs::stats)
  int* work_ctrs = nullptr;    // one work-item counter per Winograd layer, zeroed at the start of every network pass (ConvArgs::work_ctr)
  bool wino_dynamic = true;    // D2FE_WINO_DYNAMIC=0: static round-robin split of the work items
  // sparse descriptor head (variant B unless cfg.dense_descriptors): cell flags, cell -> slot map, slot -> cell list, counts, descriptors
  bool sparse_desc = false; int sp_slots = 0; int sp_min_batch = 4;
  uint8_t* sp_flags = nullptr; int32_t* sp_slotmap = nullptr; int32_t* sp_cells = nullptr; int32_t* sp_count = nullptr; float* sp_desc = nullptr;
  float* sp_mid = nullptr; int sp_mid_imgs = 0;      // ReLU(convDa) at the selected cells between the two stages of the split sparse head (passes of <= 4 images)
  void* lk_scratch = nullptr; size_t lk_scratch_bytes = 0;   // grow-only scratch of the LK / detector entry points (lk.hip)
  float* a_samp = nullptr; float* a_cn = nullptr; int a_scap = 0;   // variant A sampling: [batch][a_scap][256] samples, [batch][256] channel norms
  float* pca_comp_t = nullptr; float* pca_mean = nullptr; int pca_dims = 0;
  // NetVLAD
  struct NvLayer { int kind, cin, cout, cout_pad, stride, act, res; float* w = nullptr; float* b = nullptr; float* out = nullptr; int oh = 0, ow = 0;
                   int gmax = 1;                 // slabs `out` has room for (a fused block may split its hidden channels over workgroup groups)
                   int slabs = 1; long slab_stride = 0; };   // of the last call: out = sum of `slabs` partial tensors `slab_stride` floats apart
  std::vector<NvLayer> nv;
  // execution plan over the flat layer list: fused MobileNetV2 blocks (netvlad_fused.hip) where the pattern matches, single layers otherwise;
  // only the LAST layer of a step is materialised in HBM (NvLayer::out), everything inside a fused block lives in LDS
  struct NvStep { int l0 = 0, l1 = 0; bool fused = false, expand = false, front = false, tail = false, xblock = false, pblock = false; float* we = nullptr; float* wp = nullptr; float* bp = nullptr; float* w0 = nullptr;
                  float* wp2 = nullptr; float* bp2 = nullptr; };      // pblock with more than 128 output channels: the second channel half's project record / bias (netvlad_pair.hip)
  int nv_feat_gmax = 1, nv_feat_slabs = 1; long nv_feat_slab_stride = 0;
  // scheduling knobs of the fused plan, read from the environment by d2fe_load_netvlad (A/B measurements; defaults measured best):
  // workgroups per launch the hidden-channel split aims at (D2FE_NV_BLOCKS), the same for the tail kernel (D2FE_NV_TAIL_BLOCKS),
  // and the number of partial slabs from which they are summed once instead of by every consumer (D2FE_NV_SLABSUM, 0 = never)
  int nv_front_tpw = 0, nv_nbuf = 0;       // D2FE_NV_FRONT_TPW, D2FE_NV_NBUF (0: the launchers decide)
  int nv_stamp_step = -1; unsigned long long* nv_stamps = nullptr; int nv_stamp_wgs = 0;     // D2FE_NV_STAMP_STEP (diagnostics)
  int nv_blocks_target = 512, nv_tail_blocks = 30, nv_slabsum = 3;      // workgroups per IMAGE the hidden-channel split aims at (blocks; tail: 3 pixel tiles x 10 groups at 15 x 20)
       // the pre-projected features (input of the VLAD stage), same slab scheme
  std::vector<NvStep> nv_plan;
  bool nv_loaded = false;
  int nv_feat = 0, nv_proj = 0, nv_k = 0;
  float *nv_pre_w = nullptr, *nv_pre_b = nullptr, *nv_aw = nullptr, *nv_aw_pack = nullptr, *nv_ab = nullptr, *nv_cen = nullptr;
  float *nv_feat_buf = nullptr, *nv_raw = nullptr, *nv_pca_out = nullptr, *nv_part = nullptr;
  bool nv_group_rule = true;       // nv_groups(): no split past two groups when the slab-sum launch costs more than the chunks it saves (D2FE_NV_GROUP_RULE=0, development library: off)
  bool nv_merge = true;            // a batch lets one workgroup walk a run of hidden-channel groups (NvBlockArgs::gmerge; D2FE_NV_MERGE=0, development library: never) -- same bits either way
  float *nv_pca_comp = nullptr, *nv_pca_mean = nullptr; int nv_pca_m = 0;
  uint8_t* nv_s_img = nullptr; float* nv_s_out = nullptr;
  bool fuse1a = true;      // conv1a fused into conv1b's staging (D2FE_FUSE1A=0 keeps the stand-alone conv1a kernel)
  // host-pointer matcher: pool of (stream, scratch) slots so that concurrent callers (the reference calls matchKNN from three
  // threads) neither share state nor pay hipStreamCreate / hipMalloc / hipFree (a device-wide sync) per call
  struct MatchSlot { hipStream_t stream = nullptr; char* buf = nullptr; char* pin = nullptr; size_t bytes = 0; bool busy = false; };
  std::deque<MatchSlot> match_slots;     // deque: growing it never relocates the slots other threads are using
  std::mutex match_mu;
  // device-API matcher scratch (cand4), one per caller stream: calls on different streams may overlap on the GPU
  struct MatchScratch { hipStream_t stream = nullptr; int32_t* cand4 = nullptr; size_t bytes = 0; int npairs = 0; };
  std::deque<MatchScratch> m_scratch;
  // profiling (HIP events on the launch stream)
  int prof_mode = 0;
  std::vector<hipEvent_t> prof_pool;
  size_t prof_used = 0;
  struct ProfRec { int stage; hipEvent_t a, b; };
  std::vector<ProfRec> prof_recs;
};


namespace d2fe {
// launch sequences (api.hip).  run_superpoint == one TensorRT executeV2 + processOutput of the reference; run_netvlad == one
// MobileNetVLADONNX::inference.  Both only enqueue work on the given stream(s)
int run_superpoint(d2fe_context* h, const uint8_t* d_gray, int n, int W, int H, int stride, size_t image_stride, float* d_kps, float* d_scores,
                   float* d_desc, int32_t* d_idx, int cap, int32_t* d_n, hipStream_t s, hipStream_t s_tail = nullptr, int bs = 0);
int run_netvlad(d2fe_context* h, const uint8_t* d_gray, int n, int W, int H, int stride, size_t image_stride, float* d_out, hipStream_t s);
int check_geometry(d2fe_context* h, int n, int W, int H, int stride, int cap);
int nv_check(d2fe_context* h, int n, int W, int H, int stride);
// a second context on the same device that BORROWS the parent's packed weights (SuperPoint, NetVLAD, PCA matrices) and owns its own
// activations, scratch, counters and streams: one lane of the frames-in-flight pipeline.  d2fe_destroy() of a lane leaves the weights alone;
// the parent must outlive its lanes and must not reload weights while they exist
// `stream` (optional): the lane's launch stream, e.g. one created with a CU mask (the lane then owns it); `ncu` (optional): the compute units that
// stream may use -- the persistent kernels size their grids on it
// with_netvlad: the lane will run NetVLAD itself (its own activation buffers); a lane never gets the host-pointer staging of d2fe_create
int clone_lane(d2fe_context* parent, int max_batch, d2fe_context** out, hipStream_t stream = nullptr, int ncu = 0, bool with_netvlad = true);
// a new non-blocking stream that does NOT take turns with `beside` on the device (pipe.hip: the same measurement as d2fe_pipe_create's stream placement, on up to four
// candidates; the first candidate if none can be told apart).  hipSuccess or the failing call's error
hipError_t create_stream_beside(int device_id, hipStream_t beside, hipStream_t* out);
}  // namespace d2fe

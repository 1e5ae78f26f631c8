// pipe.hip -- frames in flight: the stereo-frame front end of include/d2fe.h (d2fe_pipe_*) as K lanes.
//
// The reference processes ONE stereo frame at a time on one thread (D2Frontend::processStereoframe, d2frontend.cpp:155-169): per
// image SuperPoint::infer and, for the main camera, MobileNetVLADONNX::inference (LoopCam::extractorImgDescDeepnet,
// loop_cam.cpp:589-648, called for both cameras by generateStereoImageDescriptor :440-470), then the two matchKNN calls of
// D2FeatureTracker::trackLocalFrames (left <-> right, d2featuretracker.cpp:658-695; left <-> previous left, :403-456).  A one- or
// two-image pass leaves most of a 256-CU device idle (latency-bound launches, 60 x 80 layers with a few hundred work items), so the
// throughput form of the same work keeps K frames in flight: submit() only enqueues, wait() returns the frame's results.
//
// A lane = its own context (clone_lane: activations, counters, scratch, streams of its own; the packed weights are the parent's),
// a device input buffer and TWO output blocks (alternating, so the block the NEXT frame's temporal match reads is not the one this
// lane writes on its next turn).  Per submit, on the lane's stream:
//     H2D (one DMA from pinned memory) -> [NetVLAD of the left images on the lane's second stream] -> SuperPoint of the 2F images
//     -> ONE matcher launch over {L_f <-> R_f, L_f <-> L_(f-1)} (the pair table addresses the previous frame's block directly; the
//        launch waits for the previous submit's extraction event, nothing else of it) -> ONE D2H of the block into pinned memory.
// Outputs are bit-identical to the single-call entry points (same kernels, same launch shapes for the same image count).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "context.h"
#include "stream_deal.h"

using namespace d2fe;

struct d2fe_pipe_s {
  d2fe_context* parent = nullptr;
  d2fe_pipe_config cfg{};
  // K lanes; a PASS = the g <= C consecutive submits that run as one launch sequence on one lane (C = cfg.coalesce; C > 1 needs F = 1).
  // Image / output row order of a pass: C == 1: L_0 .. L_(F-1), R_0 .. R_(F-1); C > 1: L, R of submit 0, L, R of submit 1, ... (a pass of g
  // submits is a prefix, so every address is fixed at creation whatever g turns out to be)
  int K = 0, F = 0, C = 1, NI = 0, W = 0, H = 0, cap = 0, D = 256, G = 0, npp = 0;      // NI: images per full pass; npp: matcher pairs per submit
  // block layout (float words from the block base; every array starts on a 64-word boundary)
  size_t o_desc = 0, o_kps = 0, o_scores = 0, o_nv = 0, o_cnt = 0, o_mn = 0, o_mq = 0, o_mt = 0, o_md = 0, o_idx = 0, blk_words = 0, d2h_words = 0;
  float* d_all = nullptr;            // [64 zero words | K lanes x 2 sets x block]
  MatchPairDesc* d_pairs = nullptr;  // [K][2][C variants: submits of the PREVIOUS pass][C * npp]
  int32_t* d_match_scratch = nullptr; size_t match_scratch_lane = 0;   // per lane: tickets + records
  struct Lane {
    d2fe_context* ctx = nullptr;
    hipStream_t s = nullptr, nv = nullptr;
    hipEvent_t ev_up = nullptr, ev_nv = nullptr, ev_ext[2] = {nullptr, nullptr}, ev_done = nullptr;
    // device views of the two result blocks (d2fe_pipe_device_view / _release): views handed out and not released yet; ev_rel = the consumers' last release
    hipEvent_t ev_rel[2] = {nullptr, nullptr};
    int views[2] = {0, 0};
    bool rel_pending[2] = {false, false};
    bool nv_on_side[2] = {false, false};      // the pass that wrote this block ran NetVLAD on the lane's second stream (auto mode decides per pass)
    uint8_t* d_img = nullptr;
    uint8_t* pin_in = nullptr;
    float* pin_out[2] = {nullptr, nullptr};
    long long rec = -1, synced = -1;   // the pass whose completion ev_done last recorded / the newest pass known to be complete (idle: synced >= rec)
  };
  std::vector<Lane> lanes;
  std::vector<int> first_class, second_class;   // the hardware-pipe class place_streams() measured for each lane's two streams (-1: not measured / no such stream)
  int n_classes = 0;
  long long probe_ticks = 0; double probe_turns_us = 0.0;      // the spin length and the 'takes turns' threshold of that measurement (d2fe_pipe_classify_stream)
  uint8_t* d_img_all = nullptr;      // the lanes' input buffers, one allocation: lane k at k * NI images (netvlad_group reads several lanes' left images with one stride)
  // netvlad_group = M > 1 (frames == 1, coalesce == 1, lanes % M == 0): the NetVLAD descriptors of M consecutive submits come from ONE call on the pipe's own
  // context and stream (NetVLAD at one image is ~20 launches of a few workgroups each: 0.25 ms for one image, 0.28 ms for four), while SuperPoint and the
  // matches of every submit are launched at once.  Tickets [i M, (i + 1) M) use lanes k0 .. k0 + M - 1; a wait() launches the part that is there
  int M = 1;
  d2fe_context* gctx = nullptr; hipStream_t gnv = nullptr;
  float* d_gnv = nullptr; float* pin_gnv = nullptr;      // [2 sets][K][G]
  std::vector<hipEvent_t> ev_g;                          // [2 sets][K / M]: the group's descriptors are in pinned memory
  std::vector<char> g_synced;
  long long g_first = 0;                                 // first ticket whose NetVLAD has not been launched
  std::mutex mu;                     // submit() and wait() may come from different threads (image callback / tracker); wait() drops it while it blocks
  long long next_ticket = 0;
  long long next_pass = 0;           // passes started so far
  int pend = 0;                      // submits of the newest pass that are staged but not launched yet (0: no pass open)
  int prev_g = 1;                    // submits of the last LAUNCHED pass
  bool nv_inline = false;            // cfg.netvlad_inline == 1: NetVLAD always on the lane's one stream (no second stream exists)
  bool nv_auto = false;              // cfg.netvlad_inline == 2: per pass (pipe_flush): the second stream while at most two passes are in flight, inline beyond
  int failed = D2FE_OK;              // sticky: the first error of an enqueue leaves a pass half-queued (frames copied or not, events recorded or not); the ring
                                     // bookkeeping of every later pass would build on it, so every later submit / wait returns this code instead
  std::string failed_msg;
  struct TInfo { long long pass = -1; int j = 0; bool waited = true; };
  long long oldest_unwaited = 0;     // every ticket below has been waited for (dynamic batching keeps the passes since then inside the result ring)
  std::vector<TInfo> tinfo;          // ring over tickets
  float* block(int lane, int set) const { return d_all + 64 + ((size_t)lane * 2 + set) * blk_words; }
  int left_row(int j, int f) const { return C > 1 ? 2 * j : f; }
  int right_row(int j, int f) const { return C > 1 ? 2 * j + 1 : F + f; }
};

namespace {

size_t up64(size_t w) { return (w + 63) / 64 * 64; }

int pipe_fail(int code, const std::string& msg) { return ctx_fail(code, msg); }

// ---- stream placement by measurement ---------------------------------------------------------------------------------------------------------
// The device runs the busy streams of a process side by side only when they sit on different hardware pipes (four of them); two busy streams on one pipe take turns
// (profiles/r05_pipe_one_frame.txt (8): a lone single-frame pass whose SuperPoint and NetVLAD streams shared a pipe ran at 873 stereo frames/s instead of 1400).  Which pipe a
// stream gets is the runtime's business (hardware queues are handed out from a pool that earlier streams of the process have used and returned), so it is MEASURED: chains
// of short dependent spin launches are timed on pairs of the candidate streams, candidates whose chains take turns are one class, and the lanes then take their streams
// from the classes so that a lane's two streams -- and the streams of consecutive lanes -- are in different ones (stream_deal.h).
__global__ void pipe_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

constexpr int PROBE_CHAIN = 6;         // dependent launches per stream and probe
constexpr double PROBE_SPIN_US = 20.0;

double probe_pair_us(hipStream_t a, hipStream_t b, long long ticks) {       // < 0: HIP error
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < PROBE_CHAIN; ++i) {
    hipLaunchKernelGGL(pipe_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    if (b) hipLaunchKernelGGL(pipe_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
  }
  if (hipStreamSynchronize(a) != hipSuccess || (b && hipStreamSynchronize(b) != hipSuccess) || hipGetLastError() != hipSuccess) return -1.0;
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// n_first + n_second streams out of measured classes; on any doubt (HIP error, a device that is not quiet, nothing told apart) the classes a fresh process would have
// (hardware queues in creation order, queue i on hardware pipe i mod 4) are assumed instead and *n_classes = 0 says so
int place_streams(int device_id, int n_first, int n_second, std::vector<hipStream_t>& first, std::vector<hipStream_t>& second, std::vector<int>& first_class,
                  std::vector<int>& second_class, int* n_classes, long long* probe_ticks, double* probe_turns_us) {
  const int need = n_first + n_second;
  *probe_ticks = 0; *probe_turns_us = 0.0;
  first.assign((size_t)n_first, nullptr); second.assign((size_t)n_second, nullptr);
  first_class.assign((size_t)n_first, -1); second_class.assign((size_t)n_second, -1);
  *n_classes = 0;
  // Exactly the streams the pipe needs to begin with.  A version with four spare candidates per pipe (and lanes that created and destroyed a stream of their own before
  // adopting one) measured the same in one process, but two ranks sharing ONE GPU went from 26 to 36-71 ms per step (bisected, profiles/r05_pipe_one_frame.txt (8g)) --
  // consistent with the two processes' hardware queues no longer fitting the device's queue slots.  So a spare is only created while the streams at hand cannot be dealt
  // out as wanted (never in a fresh process).
  std::vector<hipStream_t> cand;
  auto drop = [&](int rc) { for (auto c : cand) if (c) (void)hipStreamDestroy(c); first.assign(first.size(), nullptr); second.assign(second.size(), nullptr); return rc; };
  auto add_candidate = [&]() { hipStream_t q = nullptr; if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return false; cand.push_back(q); return true; };
  for (int c = 0; c < need; ++c) if (!add_candidate()) return drop(pipe_fail(D2FE_ERR_HIP, "hipStreamCreateWithFlags"));
  std::vector<int> cls, rep;        // class of every candidate; one candidate per class
  long long ticks = 0;
  double turns = 0.0;
  bool ok = true, quiet = false;
  auto takes_turns = [&](hipStream_t a, hipStream_t b) {       // a slow sample is confirmed once (a busy host or device looks the same)
    double t = probe_pair_us(a, b, ticks);
    if (t >= turns) t = std::min(t, probe_pair_us(a, b, ticks));
    if (t < 0) ok = false;
    return t >= turns;
  };
  auto classify = [&](int c) {       // appends cls[c]
    int k = -1;
    for (size_t r = 0; ok && r < rep.size() && k < 0; ++r) if (takes_turns(cand[c], cand[rep[r]])) k = (int)r;
    if (ok && k < 0) { if (rep.size() < 8) { k = (int)rep.size(); rep.push_back(c); } else { quiet = false; k = 0; } }     // more than eight classes: noise, not hardware
    cls.push_back(k);
  };
  if (need > 1 && d2fe_dev_env("D2FE_PIPE_PLACEMENT", 1) != 0) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) != hipSuccess || khz <= 0) khz = 100000;
    ticks = (long long)(PROBE_SPIN_US * 1e-3 * khz);
    ok = hipDeviceSynchronize() == hipSuccess;
    // the chain alone (the first sample pays for loading the kernel).  The measurement needs a quiet device: when the solo samples disagree by more than a third, somebody
    // else is using it (another process of a shared GPU, another handle of this process at work) and nothing below would mean anything
    double solo = -1.0, solo_max = 0.0;
    for (int attempt = 0; ok && !quiet && attempt < 2; ++attempt) {      // a device that looks busy is asked ONCE more (one stray launch of another handle
      solo = -1.0; solo_max = 0.0;                                       // must not cost the pipe its placement); the first sample of the first attempt loads the kernel
      for (int i = 0; ok && i < 5; ++i) {
        const double t = probe_pair_us(cand[(size_t)i % cand.size()], nullptr, ticks);
        if (t < 0) ok = false; else if (i > 0 || attempt > 0) { solo = solo < 0 ? t : std::min(solo, t); solo_max = std::max(solo_max, t); }
      }
      quiet = ok && solo_max <= 1.35 * solo;
      if (ok && !quiet) ok = hipDeviceSynchronize() == hipSuccess;
    }
    turns = 1.6 * std::max(solo, PROBE_CHAIN * PROBE_SPIN_US);
    for (int c = 0; ok && quiet && c < need; ++c) classify(c);
    // cross-check: two members of one class (neither its representative) take turns with each other too
    for (int c = 0; ok && quiet && c < (int)cls.size(); ++c)
      for (int d = c + 1; ok && quiet && d < (int)cls.size(); ++d)
        if (cls[c] == cls[d] && c != rep[cls[c]] && d != rep[cls[d]]) { if (!takes_turns(cand[c], cand[d])) quiet = false; break; }
  }
  bool measured = ok && quiet && rep.size() >= 2 && (int)cls.size() == need;
  std::vector<int> pick_first, pick_second;
  // (stream_deal.h: lane k's own stream from class k mod n, its second stream two classes further on, as far as the streams at hand allow; returns how far from that)
  auto deal = [&](const std::vector<int>& cl, int ncls) { return deal_streams(cl, ncls, n_first, n_second, pick_first, pick_second); };
  if (measured) {
    int bad = deal(cls, (int)rep.size());
    // (up to sixteen: a process that already holds every hardware queue of its pool gets new streams on the least-shared queues first, and those may all sit on three of
    // the four pipes -- measured in bench.py after its other legs: twelve candidates, three classes; creating them costs such a process no further queue)
    for (int extra = 0, stale = 0; bad > 0 && extra < 16 && stale < 6 && ok && quiet; ++extra) {      // (six spares in a row that change nothing: that is all there is)
      if (!add_candidate()) break;
      classify((int)cand.size() - 1);
      const int was = bad;
      if (ok && quiet) bad = deal(cls, (int)rep.size());
      stale = bad < was ? 0 : stale + 1;
    }
    measured = ok && quiet;
    (void)hipGetLastError();
  }
  if (d2fe_dev_env("D2FE_PIPE_PLACEMENT", 1) == 2) {      // development library: the classes in creation order
    fprintf(stderr, "[d2fe] place_streams need %d candidates %zu classes %zu measured %d:", need, cand.size(), rep.size(), (int)measured);
    for (int c : cls) fprintf(stderr, " %d", c);
    fprintf(stderr, "\n");
  }
  if (!measured) {      // creation order of the first `need` candidates
    cls.assign(cand.size(), 0);
    for (size_t c = 0; c < cand.size(); ++c) cls[c] = (int)(c % 4);
    (void)deal(cls, 4);
  } else {
    *probe_ticks = ticks; *probe_turns_us = turns; *n_classes = (int)rep.size();
  }
  std::vector<char> keep(cand.size(), 0);
  for (int k = 0; k < n_first; ++k) { first[k] = cand[pick_first[k]]; keep[pick_first[k]] = 1; if (measured) first_class[k] = cls[pick_first[k]]; }
  for (int k = 0; k < n_second; ++k) { second[k] = cand[pick_second[k]]; keep[pick_second[k]] = 1; if (measured) second_class[k] = cls[pick_second[k]]; }
  for (size_t c = 0; c < cand.size(); ++c) if (!keep[c]) (void)hipStreamDestroy(cand[c]);
  return D2FE_OK;
}

int lane_sync(d2fe_pipe_s::Lane& L) {       // called with the pipe's mutex held for the whole wait
  if (L.synced < L.rec) {
    HIP_TRY(hipEventSynchronize(L.ev_done));
    L.synced = L.rec;
  }
  return D2FE_OK;
}

// passes launched and not known to be complete (one hipEventQuery per busy lane); < 0: a HIP error (d2fe_last_error is set)
int passes_in_flight(d2fe_pipe_s* p) {
  int inflight = 0;
  for (auto& Lq : p->lanes) {
    if (Lq.synced >= Lq.rec) continue;
    const hipError_t q = hipEventQuery(Lq.ev_done);
    if (q == hipSuccess) Lq.synced = Lq.rec;
    else if (q == hipErrorNotReady) ++inflight;
    else { (void)ctx_fail(D2FE_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q)); return -1; }
  }
  (void)hipGetLastError();        // hipErrorNotReady is sticky in hipGetLastError
  return inflight;
}

// launches the open pass (its `pend` submits are staged in the lane's device input buffer): NetVLAD, SuperPoint, ONE matcher launch, ONE D2H
int pipe_flush(d2fe_pipe_s* p) {
  if (p->pend == 0) return D2FE_OK;
  const long long P = p->next_pass - 1;
  const int k = (int)(P % p->K), set = (int)((P / p->K) & 1), g = p->pend;
  auto& L = p->lanes[k];
  const size_t img = (size_t)p->W * p->H;
  const int F = p->F, W = p->W, H = p->H;
  hipStream_t s = L.s;
  float* B = p->block(k, set);
  const int n_left = p->C > 1 ? g : F, n_img = p->C > 1 ? 2 * g : 2 * F;
  const size_t left_stride = p->C > 1 ? 2 * img : img;
  int rc;
  // Where this pass's NetVLAD runs.  The device runs FOUR busy streams of a process side by side and makes a fifth take turns (d2fe_pipe_create), so in auto mode a pass
  // takes the lane's second stream (SuperPoint and NetVLAD beside each other: the shortest latency of one frame) only while that keeps the busy streams at four or fewer,
  // i.e. with at most one other pass in flight; beyond, NetVLAD goes in front of SuperPoint on the lane's one stream
  bool nv_side = p->cfg.netvlad && !p->nv_inline && p->M == 1;
  if (nv_side && p->nv_auto) {
    const int infl = passes_in_flight(p);
    if (infl < 0) return D2FE_ERR_HIP;
    nv_side = infl <= 1;
  }
  L.nv_on_side[set] = nv_side;
  // device views of this block (handed out 2 K passes ago): the consumers' stream must be through with it before this pass writes it.  The NetVLAD
  // stream is ordered behind this wait through ev_up
  if (L.views[set] > 0) return pipe_fail(D2FE_ERR_INVALID, "a device view of this lane's result block was not released (d2fe_pipe_device_release) within 2 * lanes passes");
  if (L.rel_pending[set]) { HIP_TRY(hipStreamWaitEvent(s, L.ev_rel[set], 0)); L.rel_pending[set] = false; }
  if (p->cfg.netvlad && p->M > 1) HIP_TRY(hipEventRecord(L.ev_up, s));       // netvlad_group: the pipe's NetVLAD stream waits for this lane's frames
  // NetVLAD of the pass's left images in ONE call (its arithmetic order does not depend on the batch: run_netvlad decides the hidden-channel
  // split per image), C > 1: the left images are every second image of the lane's input buffer
  auto netvlad = [&](hipStream_t st) -> int { return run_netvlad(L.ctx, L.d_img, n_left, W, H, W, left_stride, B + p->o_nv, st); };
  // Launch order on the host: SuperPoint FIRST.  Its 13 dependent launches are the long pole of a pass (0.58 ms of kernels for one stereo pair against NetVLAD's
  // 0.25-0.3 ms beside it), and enqueuing NetVLAD's 27 launches costs ~0.12 ms of host time: issued first (rounds 3-4) they held conv1b back by that much in a
  // pass of one frame.  NetVLAD on the lane's second stream still only waits for the frames (ev_up)
  if (nv_side) HIP_TRY(hipEventRecord(L.ev_up, s));
  else if (p->cfg.netvlad && p->M == 1) {        // inline: ONE stream per lane, NetVLAD in front of SuperPoint (behind it measured 8 % slower at four lanes: 1869 vs 2039)
    rc = netvlad(s);
    if (rc) return rc;
  }
  rc = run_superpoint(L.ctx, L.d_img, n_img, W, H, W, img, B + p->o_kps, B + p->o_scores, B + p->o_desc, reinterpret_cast<int32_t*>(B + p->o_idx), p->cap,
                      reinterpret_cast<int32_t*>(B + p->o_cnt), s);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev_ext[set], s));
  if (nv_side) {
    HIP_TRY(hipStreamWaitEvent(L.nv, L.ev_up, 0));
    rc = netvlad(L.nv);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(L.ev_nv, L.nv));
  }
  if (p->npp > 0) {
    if (p->cfg.match_prev && P > 0 && p->K > 1) {      // the first temporal pair reads the previous pass's block: wait for ITS extraction only
      const int pk = k > 0 ? k - 1 : p->K - 1, pset = k > 0 ? set : set ^ 1;
      HIP_TRY(hipStreamWaitEvent(s, p->lanes[pk].ev_ext[pset], 0));
    }
    const int npairs = g * p->npp, maxp = p->C * p->npp;
    MatchArgs m{};
    m.pairs = p->d_pairs + (((size_t)k * 2 + set) * p->C + (p->prev_g - 1)) * maxp;
    m.npairs = npairs; m.dim = p->D; m.max_n = p->cap; m.mode = 0; m.ratio = p->cfg.ratio; m.radius = -1.0;
    m.q_idx = reinterpret_cast<int32_t*>(B + p->o_mq); m.t_idx = reinterpret_cast<int32_t*>(B + p->o_mt); m.dist = B + p->o_md;
    m.n_out = reinterpret_cast<int32_t*>(B + p->o_mn);
    match_scratch_carve(reinterpret_cast<char*>(p->d_match_scratch) + p->match_scratch_lane * k, maxp, &m);
    m.stats = p->parent->match_stats; m.ncu = L.ctx->ncu;
    HIP_TRY(launch_match(m, s));
  }
  if (nv_side) HIP_TRY(hipStreamWaitEvent(s, L.ev_nv, 0));
  HIP_TRY(hipMemcpyAsync(L.pin_out[set], B, sizeof(float) * p->d2h_words, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipEventRecord(L.ev_done, s));
  L.rec = P;
  p->prev_g = g;
  p->pend = 0;
  return D2FE_OK;
}

// netvlad_group: NetVLAD of the tickets [g_first, upto) -- all inside one aligned group, i.e. consecutive lanes -- as ONE call
int pipe_flush_group(d2fe_pipe_s* p, long long upto) {
  if (p->M <= 1 || upto <= p->g_first) return D2FE_OK;
  const long long t0 = p->g_first;
  const int n = (int)(upto - t0), k0 = (int)(t0 % p->K), gset = (int)((t0 / p->K) & 1), slot = gset * (p->K / p->M) + k0 / p->M;
  const size_t img = (size_t)p->W * p->H;
  for (int i = 0; i < n; ++i) HIP_TRY(hipStreamWaitEvent(p->gnv, p->lanes[k0 + i].ev_up, 0));
  float* d_out = p->d_gnv + ((size_t)gset * p->K + k0) * p->G;
  const int rc = run_netvlad(p->gctx, p->d_img_all + (size_t)k0 * p->NI * img, n, p->W, p->H, p->W, (size_t)p->NI * img, d_out, p->gnv);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(p->pin_gnv + ((size_t)gset * p->K + k0) * p->G, d_out, sizeof(float) * (size_t)n * p->G, hipMemcpyDeviceToHost, p->gnv));
  HIP_TRY(hipEventRecord(p->ev_g[slot], p->gnv));
  p->g_synced[slot] = 0;
  p->g_first = upto;
  return D2FE_OK;
}

}  // namespace

namespace d2fe {
hipError_t create_stream_beside(int device_id, hipStream_t beside, hipStream_t* out) {
  *out = nullptr;
  // candidates one at a time (the first is almost always the one: a stream created right after `beside` sits on the next hardware pipe); the ones that took turns are
  // given back at the end, not before -- the runtime would hand the same queue out again
  hipStream_t cand[4] = {nullptr, nullptr, nullptr, nullptr};
  int n = 0, pick = -1;
  long long ticks = 0;
  double turns = 0.0;
  hipError_t e = hipSuccess;
  while (pick < 0 && n < 4) {
    if ((e = hipStreamCreateWithFlags(&cand[n], hipStreamNonBlocking)) != hipSuccess) break;
    ++n;
    if (!beside || d2fe_dev_env("D2FE_PIPE_PLACEMENT", 1) == 0) { pick = 0; break; }
    if (n == 1) {
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) != hipSuccess || khz <= 0) khz = 100000;
      ticks = (long long)(PROBE_SPIN_US * 1e-3 * khz);
      double solo = -1.0, solo_max = 0.0;
      for (int i = 0; i < 4; ++i) { const double t = probe_pair_us(cand[0], nullptr, ticks); if (i > 0 && t >= 0) { solo = solo < 0 ? t : std::min(solo, t); solo_max = std::max(solo_max, t); } }
      if (solo < 0 || solo_max > 1.35 * solo) { pick = 0; break; }      // not a quiet device: nothing to measure
      turns = 1.6 * std::max(solo, PROBE_CHAIN * PROBE_SPIN_US);
    }
    double t = probe_pair_us(cand[n - 1], beside, ticks);
    if (t >= turns) t = std::min(t, probe_pair_us(cand[n - 1], beside, ticks));
    if (t >= 0 && t < turns) pick = n - 1;
  }
  (void)hipGetLastError();
  if (e == hipSuccess && pick < 0) pick = 0;
  for (int c = 0; c < n; ++c) if (e != hipSuccess || c != pick) (void)hipStreamDestroy(cand[c]);
  if (e == hipSuccess) *out = cand[pick];
  return e;
}
}  // namespace d2fe

extern "C" {

void d2fe_pipe_destroy(d2fe_pipe p);

void d2fe_pipe_default_config(d2fe_pipe_config* c) {
  memset(c, 0, sizeof(*c));
  c->struct_size = (int32_t)sizeof(d2fe_pipe_config);
  c->lanes = 4; c->frames = 1; c->width = 640; c->height = 480; c->cap = 200;
  c->netvlad = 1; c->match_lr = 1; c->match_prev = 1; c->pinned_input = 0;
  c->ratio = 0.8; c->radius_lr = -1.0; c->radius_prev = -1.0;
  c->coalesce = 1;
  c->netvlad_inline = 2;      // auto
}

int d2fe_pipe_create(d2fe_handle h, const d2fe_pipe_config* cfg, d2fe_pipe* out) {
  if (!h || !cfg || !out) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(d2fe_pipe_config)) return pipe_fail(D2FE_ERR_INVALID, "d2fe_pipe_config size mismatch");
  if (!h->sp_loaded) return pipe_fail(D2FE_ERR_NOT_READY, "superpoint weights not loaded");
  if (cfg->netvlad && !h->nv_loaded) return pipe_fail(D2FE_ERR_NOT_READY, "netvlad weights not loaded");
  if (cfg->lanes < 1 || cfg->lanes > 16 || cfg->frames < 1 || cfg->frames > 4096) return pipe_fail(D2FE_ERR_INVALID, "lanes must be 1..16, frames >= 1");
  const int C = cfg->coalesce > 0 ? cfg->coalesce : 1;
  if (C > 16 || (C > 1 && cfg->frames != 1)) return pipe_fail(D2FE_ERR_INVALID, "coalesce must be 1..16 and needs frames == 1");
  if (cfg->coalesce_depth < 0 || cfg->coalesce_depth > cfg->lanes) return pipe_fail(D2FE_ERR_INVALID, "coalesce_depth must be 0..lanes");
  if (cfg->netvlad_inline < 0 || cfg->netvlad_inline > 2) return pipe_fail(D2FE_ERR_INVALID, "netvlad_inline must be 0 (side stream), 1 (inline) or 2 (auto)");
  const int M = cfg->netvlad && cfg->netvlad_group > 1 ? cfg->netvlad_group : 1;
  if (M > 1 && (cfg->frames != 1 || C != 1 || cfg->lanes % M != 0)) return pipe_fail(D2FE_ERR_INVALID, "netvlad_group needs frames == 1, coalesce == 1 and lanes % netvlad_group == 0");
  if (cfg->cap < 1 || cfg->cap > 16384) return pipe_fail(D2FE_ERR_INVALID, "cap out of range");
  if (h->cfg.max_keypoints < 0) return pipe_fail(D2FE_ERR_UNSUPPORTED, "keep-all handles (max_keypoints = -1) are served by the single-call entry points");
  if (cfg->width > h->cfg.max_width || cfg->height > h->cfg.max_height) return pipe_fail(D2FE_ERR_INVALID, "frame size exceeds the handle's maximum");
  HIP_TRY(hipSetDevice(h->cfg.device_id));
  d2fe_pipe_s* p = new d2fe_pipe_s();
  p->parent = h; p->cfg = *cfg;
  h->live_pipes.fetch_add(1);        // from here on d2fe_pipe_destroy (every failure path below goes through it or through `delete p` + the decrement) gives it back
  p->M = M;
  p->nv_inline = cfg->netvlad_inline == 1;
  p->nv_auto = cfg->netvlad_inline == 2 && cfg->lanes > 2;      // one or two lanes: never more than four busy streams, the second stream always
  p->K = cfg->lanes; p->F = cfg->frames; p->C = C; p->NI = 2 * cfg->frames * C; p->W = cfg->width; p->H = cfg->height;
  p->cap = cfg->cap < h->cfg.max_keypoints ? cfg->cap : h->cfg.max_keypoints;
  p->D = d2fe_desc_dim(h);
  p->G = cfg->netvlad ? d2fe_netvlad_dim(h) : 0;
  p->npp = (cfg->match_lr ? p->F : 0) + (cfg->match_prev ? p->F : 0);
  p->tinfo.resize((size_t)2 * p->K * C + C);
  {
    int rc = check_geometry(h, 1, p->W, p->H, p->W, p->cap);
    if (rc) { h->live_pipes.fetch_sub(1); delete p; return rc; }
    if (cfg->netvlad) { rc = nv_check(h, 1, p->W, p->H, p->W); if (rc) { h->live_pipes.fetch_sub(1); delete p; return rc; } }
  }
  const size_t NI = p->NI, cap = p->cap, F = p->F, NL = (size_t)F * C, MP = (size_t)p->npp * C;
  size_t o = 0;
  p->o_desc = o; o += up64(NI * cap * p->D);
  p->o_kps = o; o += up64(NI * cap * 2);
  p->o_scores = o; o += up64(NI * cap);
  p->o_nv = o; o += up64(NL * (size_t)p->G);
  p->o_cnt = o; o += up64(NI);
  p->o_mn = o; o += up64(MP);
  p->o_mq = o; o += up64(MP * cap);
  p->o_mt = o; o += up64(MP * cap);
  p->o_md = o; o += up64(MP * cap);
  p->d2h_words = o;
  p->o_idx = o; o += up64(NI * cap);
  p->blk_words = o;
  const int rc = [&]() -> int {
    const size_t all_words = 64 + (size_t)p->K * 2 * p->blk_words;
    HIP_TRY(hipMalloc(&p->d_all, sizeof(float) * all_words));
    HIP_TRY(hipMemset(p->d_all, 0, sizeof(float) * all_words));
    HIP_TRY(hipMalloc(&p->d_img_all, (size_t)p->W * p->H * p->NI * p->K));
    if (p->M > 1) {
      const int rcg = clone_lane(h, p->M, &p->gctx, nullptr, 0, true);
      if (rcg) return rcg;
      p->gnv = p->gctx->stream;
      HIP_TRY(hipMalloc(&p->d_gnv, sizeof(float) * 2 * p->K * p->G));
      HIP_TRY(hipHostMalloc(&p->pin_gnv, sizeof(float) * 2 * p->K * p->G, hipHostMallocDefault));
      p->ev_g.resize((size_t)2 * (p->K / p->M)); p->g_synced.assign(p->ev_g.size(), 1);
      for (auto& e : p->ev_g) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    p->lanes.resize(p->K);
    // Streams (round 5).  The device services FOUR busy compute streams of a process side by side -- one per hardware pipe -- and a fifth takes turns with one of them
    // (lanes x 1 frame, NetVLAD on a second stream per lane: 2 lanes 1830-1855 stereo fps, 3 lanes 1470-1515, 4 lanes 1640-1780; NetVLAD inline, one stream per lane: 3 lanes
    // 1779, FOUR lanes 2039, 5 lanes 1544; profiles/r05_pipe_one_frame.txt).  So: (1) a stream that would never be used (inline mode, netvlad_group, no NetVLAD) is not created
    // at all; (2) netvlad_inline = auto keeps the busy streams at four or fewer when it can (pipe_flush); (3) place_streams() MEASURES which candidate streams take turns and
    // hands every lane two streams of different hardware pipes, consecutive lanes' streams on different ones too.  (In a fresh process streams get hardware queues in creation
    // order and queue i sits on pipe i mod 4, and an arrangement by creation order measured 1390 / 1838 / 1789 / 2044 stereo frames/s for 1 / 2 / 3 / 4 single-frame passes in flight
    // on a 4-lane pipe -- but in a process that had created and destroyed other streams before, bench.py after its other legs, the same arrangement put the two streams of a
    // 1-lane pipe on ONE pipe: 878 instead of 1400.)
    const bool nv_streams = cfg->netvlad && !p->nv_inline && p->M == 1;
    const bool masked = cfg->cu_partition && p->K > 1;           // CU-masked streams are created per lane below
    struct Spare { std::vector<hipStream_t> first, second; ~Spare() { for (auto& v : {&first, &second}) for (hipStream_t q : *v) if (q) (void)hipStreamDestroy(q); } } spare;
    if (!masked) {
      const int rcs = place_streams(h->cfg.device_id, p->K, nv_streams ? p->K : 0, spare.first, spare.second, p->first_class, p->second_class, &p->n_classes, &p->probe_ticks, &p->probe_turns_us);
      if (rcs) return rcs;
    }
    for (int k = 0; k < p->K; ++k) {
      auto& L = p->lanes[k];
      hipStream_t ms = nullptr;
      int lane_cus = 0;
      if (cfg->cu_partition && p->K > 1) {
        // Disjoint compute units per lane.  Bit i of a HIP CU mask is CU i / 8 of XCD i % 8 on this device, and an XCD without any bit
        // set is NOT masked at all (tools/ubench/cu_mask_probe.hip), so a lane gets the same rows of CUs in every XCD: rows
        // [k R / K, (k + 1) R / K) of the R = CUs / 8 rows
        const int rows = h->ncu / 8, r0 = k * rows / p->K, r1 = (k + 1) * rows / p->K;
        if (r1 > r0) {
          std::vector<uint32_t> mask((h->ncu + 31) / 32, 0u);
          for (int b = 8 * r0; b < 8 * r1; ++b) mask[b / 32] |= 1u << (b % 32);
          HIP_TRY(hipExtStreamCreateWithCUMask(&ms, (uint32_t)mask.size(), mask.data()));
          if (nv_streams && hipExtStreamCreateWithCUMask(&L.nv, (uint32_t)mask.size(), mask.data()) != hipSuccess) {      // `ms` has no owner yet
            (void)hipStreamDestroy(ms);
            return pipe_fail(D2FE_ERR_HIP, "hipExtStreamCreateWithCUMask");
          }
          lane_cus = 8 * (r1 - r0);
        }
      }
      // lane_cus without a CU mask: the lane's persistent kernels size their grids for that many compute units but may run anywhere -- a
      // full-device persistent launch holds every CU's LDS until it ends, so another lane's small launches cannot start beside it; grids sized
      // for a share of the device leave workgroup slots on every CU to the other lanes
      if (!lane_cus && cfg->lane_cus > 0 && cfg->lane_cus < h->ncu) lane_cus = cfg->lane_cus;
      if (!masked) { ms = spare.first[k]; spare.first[k] = nullptr; if (nv_streams) { L.nv = spare.second[k]; spare.second[k] = nullptr; } }
      int rc2 = clone_lane(h, p->NI, &L.ctx, ms, lane_cus, cfg->netvlad && p->M == 1);
      if (rc2) { if (ms) (void)hipStreamDestroy(ms); return rc2; }
      L.s = L.ctx->stream;
      HIP_TRY(hipEventCreateWithFlags(&L.ev_up, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_nv, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_ext[0], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_ext[1], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_rel[0], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&L.ev_rel[1], hipEventDisableTiming));
      L.d_img = p->d_img_all + (size_t)k * p->NI * p->W * p->H;
      if (!cfg->pinned_input) HIP_TRY(hipHostMalloc(&L.pin_in, (size_t)p->W * p->H * p->NI, hipHostMallocDefault));
      for (int set = 0; set < 2; ++set) HIP_TRY(hipHostMalloc(&L.pin_out[set], sizeof(float) * p->d2h_words, hipHostMallocDefault));
    }
    if (nv_streams)      // CU-masked lanes whose second stream could not be created with the mask above
      for (auto& L : p->lanes) if (!L.nv) HIP_TRY(hipStreamCreateWithFlags(&L.nv, hipStreamNonBlocking));
    if (p->npp > 0) {
      // pair tables [lane][set][variant v = submits of the previous pass - 1][C * npp].  Submit j of a pass contributes npp consecutive pairs:
      // L_f <-> R_f for its F frames (when match_lr), then L_f <-> L_(f-1); the first left frame of a pass pairs with the LAST left frame
      // of the previous pass, whose row in the previous pass's block depends on how many submits that pass carried (the variant)
      const size_t maxp = MP;
      std::vector<MatchPairDesc> tab((size_t)p->K * 2 * C * maxp);
      for (int k = 0; k < p->K; ++k)
        for (int set = 0; set < 2; ++set)
          for (int v = 0; v < C; ++v) {
            float* B = p->block(k, set);
            // the previous pass: P - 1.  With P = i K + k the set is i & 1; P - 1 = i K + k - 1 (k > 0: same i) or (i - 1) K + K - 1
            const int pk = k > 0 ? k - 1 : p->K - 1;
            const int pset = k > 0 ? set : set ^ 1;
            float* PB = p->block(pk, pset);
            MatchPairDesc* row = tab.data() + (((size_t)k * 2 + set) * C + v) * maxp;
            int pi = 0;
            auto fill = [&](MatchPairDesc& d, float* BA, int ra, float* BB, int rb, double radius) {
              d.a = BA + p->o_desc + (size_t)ra * cap * p->D; d.b = BB + p->o_desc + (size_t)rb * cap * p->D;
              d.pts_a = BA + p->o_kps + (size_t)ra * cap * 2; d.pts_b = BB + p->o_kps + (size_t)rb * cap * 2;
              d.na = reinterpret_cast<int32_t*>(BA + p->o_cnt) + ra; d.nb = reinterpret_cast<int32_t*>(BB + p->o_cnt) + rb;
              d.radius = radius;
            };
            for (int j = 0; j < C; ++j) {
              for (int f = 0; cfg->match_lr && f < (int)F; ++f) fill(row[pi++], B, p->left_row(j, f), B, p->right_row(j, f), cfg->radius_lr);
              for (int f = 0; cfg->match_prev && f < (int)F; ++f) {
                if (f > 0) fill(row[pi++], B, p->left_row(j, f), B, p->left_row(j, f - 1), cfg->radius_prev);
                else if (j > 0) fill(row[pi++], B, p->left_row(j, 0), B, p->left_row(j - 1, (int)F - 1), cfg->radius_prev);
                else fill(row[pi++], B, p->left_row(0, 0), PB, p->left_row(v, (int)F - 1), cfg->radius_prev);        // previous pass: its last submit is v
              }
            }
          }
      HIP_TRY(hipMalloc(&p->d_pairs, sizeof(MatchPairDesc) * tab.size()));
      HIP_TRY(hipMemcpy(p->d_pairs, tab.data(), sizeof(MatchPairDesc) * tab.size(), hipMemcpyHostToDevice));
      p->match_scratch_lane = match_scratch_bytes((int)maxp, p->cap);
      HIP_TRY(hipMalloc(&p->d_match_scratch, p->match_scratch_lane * p->K));
      HIP_TRY(hipMemset(p->d_match_scratch, 0, p->match_scratch_lane * p->K));
    }
    return D2FE_OK;
  }();
  if (rc != D2FE_OK) { d2fe_pipe_destroy(p); return rc; }
  // the hipMemsets above ran on the null stream; the lanes' non-blocking streams do not wait for it
  if (hipDeviceSynchronize() != hipSuccess) { d2fe_pipe_destroy(p); return pipe_fail(D2FE_ERR_HIP, "hipDeviceSynchronize"); }
  *out = p;
  return D2FE_OK;
}

void d2fe_pipe_destroy(d2fe_pipe p) {
  if (!p) return;
  (void)hipSetDevice(p->parent->cfg.device_id);
  for (auto& L : p->lanes) {
    if (L.s) (void)hipStreamSynchronize(L.s);
    if (L.nv) { (void)hipStreamSynchronize(L.nv); (void)hipStreamDestroy(L.nv); }
    for (hipEvent_t e : {L.ev_up, L.ev_nv, L.ev_ext[0], L.ev_ext[1], L.ev_done, L.ev_rel[0], L.ev_rel[1]}) if (e) (void)hipEventDestroy(e);
    if (L.pin_in) (void)hipHostFree(L.pin_in);
    for (float* q : L.pin_out) if (q) (void)hipHostFree(q);
    if (L.ctx) d2fe_destroy(L.ctx);
  }
  if (p->gnv) (void)hipStreamSynchronize(p->gnv);
  for (auto e : p->ev_g) if (e) (void)hipEventDestroy(e);
  if (p->gctx) d2fe_destroy(p->gctx);
  if (p->d_gnv) (void)hipFree(p->d_gnv);
  if (p->pin_gnv) (void)hipHostFree(p->pin_gnv);
  if (p->d_img_all) (void)hipFree(p->d_img_all);
  if (p->d_pairs) (void)hipFree(p->d_pairs);
  if (p->d_match_scratch) (void)hipFree(p->d_match_scratch);
  if (p->d_all) (void)hipFree(p->d_all);
  d2fe_context* parent = p->parent;
  delete p;
  // a handle destroyed while this pipe was alive was only MARKED (d2fe_destroy): the last pipe to go releases it
  if (parent->live_pipes.fetch_sub(1) == 1 && parent->doomed.load()) d2fe_destroy(parent);
}

int d2fe_pipe_submit(d2fe_pipe p, const uint8_t* left, const uint8_t* right, int stride, size_t image_stride, int64_t* ticket) {
  if (!p || !left || !right || !ticket) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  if (stride < p->W) return pipe_fail(D2FE_ERR_INVALID, "stride < width");
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->failed) return pipe_fail(p->failed, "the pipe failed in an earlier call and accepts no more work (destroy it): " + p->failed_msg);
  // A device view of the block the NEXT pass will write that its consumer has not released is the CALLER's protocol error, found here before anything is staged or
  // advanced: the submit is refused (D2FE_ERR_NOT_READY, not sticky -- release the view and submit again); finished tickets stay collectable
  if (p->pend == 0) {
    const long long Pn = p->next_pass;
    const auto& Ln = p->lanes[(size_t)(Pn % p->K)];
    if (Ln.views[(int)((Pn / p->K) & 1)] > 0)
      return pipe_fail(D2FE_ERR_NOT_READY, "a device view of the result block this submit's pass would write has not been released (d2fe_pipe_device_release): nothing was queued");
  }
  // an error anywhere below leaves the open pass half-enqueued while next_pass / pend / the ticket ring may or may not have advanced: no later pass can
  // build on that (it would wait on a stale extraction event and index its pair table with a stale prev_g), so the first error is final for the pipe
  const int rc_all = [&]() -> int {
  const long long t = p->next_ticket;
  int rc;
  if (p->pend == 0) {
    // a new pass on lane P % K: the lane's previous pass (P - K) must be complete before its buffers are reused; with it every pass <= P - K
    // is (the invariant that makes the alternating output blocks sufficient, see the header)
    rc = lane_sync(p->lanes[(size_t)(p->next_pass % p->K)]);
    if (rc) return rc;
    if (p->M > 1) {
      // the NetVLAD call that read this lane's previous frame (K tickets ago; the other set) must be through with it before the H2D below
      const int k = (int)(p->next_pass % p->K), pset = (int)(((p->next_pass / p->K) & 1) ^ 1), slot = pset * (p->K / p->M) + k / p->M;
      if (t >= p->K) {
        if (p->g_first <= t - p->K) { rc = pipe_flush_group(p, std::min<long long>(((t - p->K) / p->M + 1) * p->M, t)); if (rc) return rc; }
        if (!p->g_synced[slot]) { HIP_TRY(hipEventSynchronize(p->ev_g[slot])); p->g_synced[slot] = 1; }
      }
    }
    ++p->next_pass;
  }
  const long long P = p->next_pass - 1;
  const int k = (int)(P % p->K), j = p->pend;
  auto& L = p->lanes[k];
  const size_t img = (size_t)p->W * p->H;
  const int F = p->F, W = p->W, H = p->H;
  hipStream_t s = L.s;
  // frames of this submit -> the lane's input buffer at the rows of submit j (H2D is issued at once: it travels while the pass fills).  Left and right frames that are
  // one contiguous run on the host AND in the lane's buffer (C == 1: rows [0, F) | [F, 2F); C > 1: rows 2j, 2j + 1) travel as ONE copy: a DMA costs ~10 us of
  // set-up whatever its size, and a one-frame pass waits for it
  const bool one_run = p->cfg.pinned_input && stride == W && image_stride == img && right == left + img * F &&
                       p->right_row(j, 0) == p->left_row(j, 0) + (size_t)F;
  if (one_run) HIP_TRY(hipMemcpyAsync(L.d_img + (size_t)p->left_row(j, 0) * img, left, 2 * img * F, hipMemcpyHostToDevice, s));
  for (int side = 0; side < 2 && !one_run; ++side) {
    const uint8_t* src = side ? right : left;
    const size_t row0 = side ? p->right_row(j, 0) : p->left_row(j, 0);       // the F images of a side are consecutive rows (C > 1: F = 1)
    uint8_t* dst = L.d_img + row0 * img;
    if (p->cfg.pinned_input) {
      if (stride == W && image_stride == img) HIP_TRY(hipMemcpyAsync(dst, src, img * F, hipMemcpyHostToDevice, s));
      else if (image_stride == (size_t)stride * H) HIP_TRY(hipMemcpy2DAsync(dst, W, src, stride, W, (size_t)H * F, hipMemcpyHostToDevice, s));
      else for (int f = 0; f < F; ++f) HIP_TRY(hipMemcpy2DAsync(dst + f * img, W, src + f * image_stride, stride, W, H, hipMemcpyHostToDevice, s));
    } else {
      uint8_t* stage = L.pin_in + row0 * img;
      for (int f = 0; f < F; ++f) {
        const uint8_t* sf = src + f * image_stride;
        if (stride == W) memcpy(stage + f * img, sf, img);
        else for (int y = 0; y < H; ++y) memcpy(stage + f * img + (size_t)y * W, sf + (size_t)y * stride, W);
      }
      // the staging buffer has the lane buffer's row order: after the right side was staged, both sides of a submit are one run
      if (side == 0 && p->right_row(j, 0) == p->left_row(j, 0) + (size_t)F) continue;
      if (side == 1 && p->right_row(j, 0) == p->left_row(j, 0) + (size_t)F)
        HIP_TRY(hipMemcpyAsync(L.d_img + (size_t)p->left_row(j, 0) * img, L.pin_in + (size_t)p->left_row(j, 0) * img, 2 * img * F, hipMemcpyHostToDevice, s));
      else HIP_TRY(hipMemcpyAsync(dst, stage, img * F, hipMemcpyHostToDevice, s));
    }
  }
  auto& ti = p->tinfo[(size_t)(t % (long long)p->tinfo.size())];
  ti.pass = P; ti.j = j; ti.waited = false;
  ++p->pend;
  p->next_ticket = t + 1;
  *ticket = t;
  bool launch = p->pend == p->C;
  if (!launch && p->cfg.coalesce_depth > 0) {
    // dynamic batching: the device would run dry with fewer than `coalesce_depth` passes in flight -- launch what is staged; otherwise let the pass grow
    const int inflight = passes_in_flight(p);
    if (inflight < 0) return D2FE_ERR_HIP;
    // ... unless the tickets nobody has waited for yet already span `lanes` passes: small passes use up the ring of 2 * lanes result blocks as fast as
    // full ones, so a caller with many frames outstanding gets full passes (which is what it wants anyway)
    const long long p_old = p->tinfo[(size_t)(p->oldest_unwaited % (long long)p->tinfo.size())].pass;
    launch = inflight < p->cfg.coalesce_depth && P - p_old < p->K;
  }
  if (launch) {
    rc = pipe_flush(p);
    if (rc) return rc;
    if (p->M > 1 && (t + 1) % p->M == 0) return pipe_flush_group(p, t + 1);      // the group is complete
  }
  return D2FE_OK;
  }();
  if (rc_all != D2FE_OK) { p->failed = rc_all; p->failed_msg = d2fe_last_error(); }
  return rc_all;
}

int d2fe_pipe_wait(d2fe_pipe p, int64_t ticket, d2fe_pipe_result* out) {
  if (!p || !out) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  memset(out, 0, sizeof(*out));
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->failed) return pipe_fail(p->failed, "the pipe failed in an earlier call (destroy it): " + p->failed_msg);
  if (ticket < 0 || ticket >= p->next_ticket) return pipe_fail(D2FE_ERR_INVALID, "unknown ticket");
  const auto ti = p->tinfo[(size_t)(ticket % (long long)p->tinfo.size())];      // a copy: the ring entry may be rewritten while this call blocks without the mutex
  // the ticket's result block is written again by the pass 2 K passes later
  if (ticket + (long long)p->tinfo.size() <= p->next_ticket || ti.pass < 0 || ti.pass + 2 * p->K < p->next_pass)
    return pipe_fail(D2FE_ERR_INVALID, "the ticket's result block has been reused: wait for a frame within 2 * lanes passes");
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  if (p->pend > 0 && ti.pass == p->next_pass - 1) {       // the ticket's pass is still filling: launch it with what it has
    const int rc = pipe_flush(p);
    if (rc) { p->failed = rc; p->failed_msg = d2fe_last_error(); return rc; }
  }
  const int k = (int)(ti.pass % p->K), set = (int)((ti.pass / p->K) & 1), j = ti.j;
  auto& L = p->lanes[k];
  int rc = D2FE_OK;
  if (L.synced < ti.pass) {
    // block WITHOUT the mutex, so that the other thread can go on submitting.  The event may be recorded again meanwhile (a later pass of this lane):
    // the wait then covers that record too, and everything the lane recorded up to `rec` is complete either way (one stream, in order)
    const long long rec = L.rec;
    hipEvent_t ev = L.ev_done;
    lk.unlock();
    const hipError_t e = hipEventSynchronize(ev);
    lk.lock();
    if (e != hipSuccess) return pipe_fail(D2FE_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    if (L.synced < rec) L.synced = rec;
    if (ticket + (long long)p->tinfo.size() <= p->next_ticket || ti.pass + 2 * p->K < p->next_pass)
      return pipe_fail(D2FE_ERR_INVALID, "the ticket's result block was reused while this call waited for it");
  }
  p->tinfo[(size_t)(ticket % (long long)p->tinfo.size())].waited = true;
  while (p->oldest_unwaited < p->next_ticket && p->tinfo[(size_t)(p->oldest_unwaited % (long long)p->tinfo.size())].waited) ++p->oldest_unwaited;
  const float* gdesc = nullptr;
  if (p->M > 1) {
    if (p->g_first <= ticket) { rc = pipe_flush_group(p, std::min<long long>((ticket / p->M + 1) * p->M, p->next_ticket)); if (rc) return rc; }
    const int slot = set * (p->K / p->M) + k / p->M;
    if (!p->g_synced[slot]) { HIP_TRY(hipEventSynchronize(p->ev_g[slot])); p->g_synced[slot] = 1; }
    gdesc = p->pin_gnv + ((size_t)set * p->K + k) * p->G;
  }
  const float* B = L.pin_out[set];
  const size_t F = p->F, cap = p->cap;
  const size_t r0 = p->left_row(j, 0);          // C == 1: 0; C > 1: the submit's L, R rows are 2 j, 2 j + 1 -- the layout of a frames = 1 result
  out->frames = p->F; out->cap = p->cap; out->desc_dim = p->D; out->netvlad_dim = p->G;
  out->kps_xy = B + p->o_kps + r0 * cap * 2; out->scores = B + p->o_scores + r0 * cap; out->desc = B + p->o_desc + r0 * cap * p->D;
  out->n_kp = reinterpret_cast<const int32_t*>(B + p->o_cnt) + r0;
  out->netvlad = gdesc ? gdesc : p->G ? B + p->o_nv + (p->C > 1 ? (size_t)j : 0) * p->G : nullptr;
  size_t pi = (size_t)j * p->npp;
  if (p->cfg.match_lr) {
    out->lr_q = reinterpret_cast<const int32_t*>(B + p->o_mq) + pi * cap; out->lr_t = reinterpret_cast<const int32_t*>(B + p->o_mt) + pi * cap;
    out->lr_dist = B + p->o_md + pi * cap; out->lr_n = reinterpret_cast<const int32_t*>(B + p->o_mn) + pi;
    pi += F;
  }
  if (p->cfg.match_prev) {
    out->prev_q = reinterpret_cast<const int32_t*>(B + p->o_mq) + pi * cap; out->prev_t = reinterpret_cast<const int32_t*>(B + p->o_mt) + pi * cap;
    out->prev_dist = B + p->o_md + pi * cap; out->prev_n = reinterpret_cast<const int32_t*>(B + p->o_mn) + pi;
  }
  return D2FE_OK;
}

// ---- device-side consumers of a ticket (the cross-agent exchange on a stream of its own) ----------------------------------------------------------------
namespace {
// the ticket's (lane, set, submit index) while its result block is still the one the ticket wrote; launches its pass if it is still staged
int view_locate(d2fe_pipe_s* p, int64_t ticket, bool may_flush, int* k, int* set, int* j) {
  if (p->failed) return pipe_fail(p->failed, "the pipe failed in an earlier call (destroy it): " + p->failed_msg);
  if (ticket < 0 || ticket >= p->next_ticket) return pipe_fail(D2FE_ERR_INVALID, "unknown ticket");
  const auto ti = p->tinfo[(size_t)(ticket % (long long)p->tinfo.size())];
  if (ticket + (long long)p->tinfo.size() <= p->next_ticket || ti.pass < 0 || ti.pass + 2 * p->K < p->next_pass)
    return pipe_fail(D2FE_ERR_INVALID, "the ticket's result block has been reused: take the view within 2 * lanes passes");
  if (p->pend > 0 && ti.pass == p->next_pass - 1) {
    if (!may_flush) return pipe_fail(D2FE_ERR_INVALID, "the ticket's pass has not been launched");
    const int rc = pipe_flush(p);
    if (rc) { p->failed = rc; p->failed_msg = d2fe_last_error(); return rc; }
  }
  *k = (int)(ti.pass % p->K); *set = (int)((ti.pass / p->K) & 1); *j = ti.j;
  return D2FE_OK;
}
}  // namespace

int d2fe_pipe_device_view(d2fe_pipe p, int64_t ticket, void* stream, d2fe_pipe_device_result* out) {
  if (!p || !out || !stream) return pipe_fail(D2FE_ERR_INVALID, "null argument (the consumer's stream must be a real hipStream_t)");
  memset(out, 0, sizeof(*out));
  if (p->M > 1) return pipe_fail(D2FE_ERR_UNSUPPORTED, "device views are not available with netvlad_group > 1 (the descriptors of a group live outside the result blocks)");
  std::lock_guard<std::mutex> lk(p->mu);
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  int k, set, j;
  const int rc = view_locate(p, ticket, true, &k, &set, &j);
  if (rc) return rc;
  auto& L = p->lanes[k];
  hipStream_t cs = static_cast<hipStream_t>(stream);
  // SuperPoint of the pass: ev_ext[set] (re-recorded only by the pass that rewrites this block, which view_locate has excluded).  NetVLAD on the lane's second
  // stream: ev_nv -- a later pass of the lane may have re-recorded it; waiting for that later record is merely later, never earlier
  HIP_TRY(hipStreamWaitEvent(cs, L.ev_ext[set], 0));
  if (L.nv_on_side[set]) HIP_TRY(hipStreamWaitEvent(cs, L.ev_nv, 0));
  const float* B = p->block(k, set);
  const size_t cap = p->cap, r0 = p->left_row(j, 0);
  out->frames = p->F; out->cap = p->cap; out->desc_dim = p->D; out->netvlad_dim = p->G;
  out->d_kps_xy = B + p->o_kps + r0 * cap * 2; out->d_scores = B + p->o_scores + r0 * cap; out->d_desc = B + p->o_desc + r0 * cap * p->D;
  out->d_n_kp = reinterpret_cast<const int32_t*>(B + p->o_cnt) + r0;
  out->d_netvlad = p->G ? B + p->o_nv + (p->C > 1 ? (size_t)j : 0) * p->G : nullptr;
  ++L.views[set];
  return D2FE_OK;
}

/* what a device-side consumer needs to size its buffers, and the stream it may queue behind (d2fe_exchange_*, csrc/exchange.hip) */
int d2fe_pipe_geometry(d2fe_pipe p, int32_t* frames, int32_t* cap, int32_t* desc_dim, int32_t* netvlad_dim) {
  if (!p) return pipe_fail(D2FE_ERR_INVALID, "null pipe");
  if (frames) *frames = p->F;
  if (cap) *cap = p->cap;
  if (desc_dim) *desc_dim = p->D;
  if (netvlad_dim) *netvlad_dim = p->G;
  return D2FE_OK;
}
d2fe_handle d2fe_pipe_handle(d2fe_pipe p) { return p ? p->parent : nullptr; }
int d2fe_pipe_lane_stream(d2fe_pipe p, int64_t ticket, void** stream) {
  if (!p || !stream) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  int k, set, j;
  const int rc = view_locate(p, ticket, true, &k, &set, &j);      // launches the ticket's pass if coalescing still holds it back: the stream order below needs it queued
  if (rc) return rc;
  *stream = p->lanes[k].s;
  return D2FE_OK;
}

int d2fe_pipe_device_release(d2fe_pipe p, int64_t ticket, void* stream) {
  if (!p || !stream) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  int k, set, j;
  const int rc = view_locate(p, ticket, false, &k, &set, &j);
  if (rc) return rc;
  auto& L = p->lanes[k];
  if (L.views[set] <= 0) return pipe_fail(D2FE_ERR_INVALID, "no device view of this ticket's block is outstanding");
  // several consumers may share a block (coalesced submits): the event is re-recorded by each release; the lane waits for the last record, and consumers that
  // release on DIFFERENT streams must order those streams themselves (documented: one consumer stream per pipe)
  HIP_TRY(hipEventRecord(L.ev_rel[set], static_cast<hipStream_t>(stream)));
  L.rel_pending[set] = true;
  --L.views[set];
  return D2FE_OK;
}

/* HIP-event timing of the lanes' launch sequences (d2fe_profile_enable / d2fe_profile_read of every lane, summed) */
int d2fe_pipe_profile_enable(d2fe_pipe p, int mode) {
  if (!p) return pipe_fail(D2FE_ERR_INVALID, "null pipe");
  for (auto& L : p->lanes) { const int rc = d2fe_profile_enable(L.ctx, mode); if (rc) return rc; }
  return D2FE_OK;
}
int d2fe_pipe_profile_read(d2fe_pipe p, float* ms, int32_t* launches) {
  if (!p || !ms || !launches) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  for (int i = 0; i < D2FE_PROF_COUNT; ++i) { ms[i] = 0.f; launches[i] = 0; }
  for (auto& L : p->lanes) {
    float m[D2FE_PROF_COUNT]; int32_t n[D2FE_PROF_COUNT];
    const int rc = d2fe_profile_read(L.ctx, m, n);
    if (rc) return rc;
    for (int i = 0; i < D2FE_PROF_COUNT; ++i) { ms[i] += m[i]; launches[i] += n[i]; }
  }
  return D2FE_OK;
}

int d2fe_pipe_lanes(d2fe_pipe p) { return p ? p->K : pipe_fail(D2FE_ERR_INVALID, "null pipe"); }

int d2fe_pipe_classify_stream(d2fe_pipe p, void* stream, int32_t* cls) {
  if (!p || !stream || !cls) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  *cls = -1;
  if (p->n_classes < 2 || p->probe_ticks <= 0) return D2FE_OK;            // nothing was told apart at creation
  std::lock_guard<std::mutex> lk(p->mu);
  HIP_TRY(hipSetDevice(p->parent->cfg.device_id));
  if (passes_in_flight(p) != 0) return pipe_fail(D2FE_ERR_INVALID, "d2fe_pipe_classify_stream measures on an idle pipe: wait for every ticket first");
  const hipStream_t x = (hipStream_t)stream;
  HIP_TRY(hipStreamSynchronize(x));
  std::vector<char> seen((size_t)p->n_classes, 0);
  for (int k = 0; k < p->K && *cls < 0; ++k)
    for (int w = 0; w < 2 && *cls < 0; ++w) {
      const int c = w == 0 ? (k < (int)p->first_class.size() ? p->first_class[k] : -1) : (k < (int)p->second_class.size() ? p->second_class[k] : -1);
      const hipStream_t r = w == 0 ? p->lanes[k].s : p->lanes[k].nv;
      if (c < 0 || c >= p->n_classes || !r || seen[c]) continue;
      seen[c] = 1;
      double t = probe_pair_us(x, r, p->probe_ticks);
      if (t >= p->probe_turns_us) t = std::min(t, probe_pair_us(x, r, p->probe_ticks));
      if (t < 0) return pipe_fail(D2FE_ERR_HIP, "stream probe");
      if (t >= p->probe_turns_us) *cls = c;
    }
  return D2FE_OK;
}

int d2fe_pipe_stream_placement(d2fe_pipe p, int32_t* classes, int32_t* n_classes) {
  if (!p || !classes || !n_classes) return pipe_fail(D2FE_ERR_INVALID, "null argument");
  for (int k = 0; k < p->K; ++k) {
    classes[2 * k] = k < (int)p->first_class.size() ? p->first_class[k] : -1;
    classes[2 * k + 1] = k < (int)p->second_class.size() ? p->second_class[k] : -1;
  }
  *n_classes = p->n_classes;
  return D2FE_OK;
}

}  // extern "C"

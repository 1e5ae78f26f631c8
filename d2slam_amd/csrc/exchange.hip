// exchange.hip -- d2fe_exchange_*: the cross-agent exchange of one rank behind a frames-in-flight pipe, inside the library.
// Replaces the reference's broadcast of the frame it has just extracted (LCM, d2frontend/src/loop_net.cpp:24-87; int8 wire form d2frontend_types.h:228-268) and
// D2FeatureTracker::trackRemoteFrames on the receivers (d2frontend/src/d2featuretracker.cpp:237-310: NetVLAD gate :185-203, matchKNN per remote frame) by ONE
// sequence per submitted ticket:
//
//   d2fe_pipe_device_view -> pack_blocks(_int8) -> ONE all-gather (RCCL: ncclAllGather, or the caller's collective) -> [int8: decode] -> counts -> NetVLAD gate
//   -> ONE matcher launch (a side in place in the lane's result block, b side in place in the gathered blocks) -> d2fe_pipe_device_release -> ONE D2H into a pinned slot
//
// Rounds 3-5 had this sequence in Python (d2slam_amd/swarm.py PipeExchange, torch.distributed for the collective) and as a C++ test program (tests/cpp/swarm_test.cpp);
// here it is behind the C ABI, so D2SLAM's C++ calls it and neither torch's stream wrapper nor Python is in the path (measured over one-rank RCCL: 2230-2235 stereo
// frames/s with the exchange against 2202-2211 for the Python-driven form).  Where it runs: ONE stream of the exchange's own (default) or the stream of the LANE
// that produced the ticket (d2fe_pipe_lane_stream, behind that lane's D2H).  The second form needs no further hardware pipe, but the lane's NEXT pass -- which
// a saturated pipe submits the moment the ticket has been waited for -- then queues behind 0.3 ms of latency-bound launches: measured 5-7 % per step against 1-3 %
// (profiles/r06_exchange_placement_ab.txt), so it is an option, not the default.  RCCL is loaded at run time (dlopen), only when a communicator is made or used.
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

#include "context.h"

using namespace d2fe;

namespace {
// the five RCCL entry points the exchange needs; ncclComm_t and ncclUniqueId stay opaque (a pointer; 128 bytes)
struct Rccl {
  struct Uid { char b[128]; };       // ncclUniqueId: 128 bytes, passed BY VALUE to ncclCommInitRank
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Uid, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string path;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load(const char* path) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return D2FE_OK;
  // an RCCL the process already holds (e.g. PyTorch's) first: two copies of the library in one process would each run their own proxy threads
  const char* names[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (int pass = 0; pass < 2 && !lib; ++pass)
    for (const char* n : names) {
      if (!n || !n[0]) continue;
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 && n != path ? RTLD_NOLOAD : 0));
      if (lib) { g_rccl.path = n; break; }
    }
  if (!lib) return ctx_fail(D2FE_ERR_UNSUPPORTED, std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found"));
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) { dlclose(lib); return ctx_fail(D2FE_ERR_UNSUPPORTED, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather"); }
  g_rccl.lib = lib;
  return D2FE_OK;
}
int rccl_fail(const char* what, int rc) {
  return ctx_fail(D2FE_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
}

// a_cnt[p] = n of the local frame of pair p (read from this rank's own packed blocks), b_cnt[p] = n of the remote block of pair p (read from the gathered fp32 blocks)
__global__ void exchange_counts_kernel(const int32_t* __restrict__ own_n, int own_stride_words, const int32_t* __restrict__ gath, int blk_words, int n_off,
                                       const int32_t* __restrict__ q_frame, const int32_t* __restrict__ rem_blk, int npairs, int32_t* __restrict__ a_cnt,
                                       int32_t* __restrict__ b_cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  a_cnt[p] = own_n[(size_t)q_frame[p] * own_stride_words];
  b_cnt[p] = gath[(size_t)rem_blk[p] * blk_words + n_off];
}
}  // namespace

struct d2fe_exchange_s {
  d2fe_pipe p = nullptr;
  d2fe_handle h = nullptr;
  d2fe_exchange_config cfg{};
  void* comm = nullptr;
  int F = 0, cap = 0, G = 0, BLK = 0, BLKB = 0, NR = 0, n_off = 0, g_off = 0, own_n_word = 0;
  bool int8 = false;
  int32_t *d_q_frame = nullptr, *d_rem_blk = nullptr, *d_a_off = nullptr, *d_b_off = nullptr;      // the pair layout (fixed)
  size_t out_words = 0, o_mq = 0, o_mt = 0, o_md = 0, o_mn = 0, o_pass = 0, o_sims = 0, o_np = 0;      // one result record: device copy and pinned slot share the layout
  struct Slot {
    float* d_blocks = nullptr; int8_t* d_blocks_q = nullptr; float* d_gath = nullptr; int8_t* d_gath_q = nullptr;
    int32_t *d_a_cnt = nullptr, *d_b_cnt = nullptr;
    float* d_out = nullptr; float* pin = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; hipEvent_t done = nullptr;
    bool busy = false; int64_t ticket = -1;
  };
  std::vector<Slot> slots;
  hipStream_t own = nullptr;       // cfg.own_stream: the one stream of its own (round 5's placement), else the lanes' streams
};

extern "C" {

int d2fe_rccl_load(const char* path) { return rccl_load(path); }
const char* d2fe_rccl_path(void) { return g_rccl.path.c_str(); }
int d2fe_rccl_unique_id(void* id128) {
  if (!id128) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  const int e = g_rccl.GetUniqueId(id128);
  return e ? rccl_fail("ncclGetUniqueId", e) : D2FE_OK;
}
int d2fe_rccl_comm_init_rank(const void* id128, int world, int rank, int device, void** comm_out) {
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return ctx_fail(D2FE_ERR_INVALID, "bad argument");
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  Rccl::Uid u;
  memcpy(u.b, id128, 128);
  void* c = nullptr;
  const int e = g_rccl.CommInitRank(&c, world, u, rank);
  if (e) return rccl_fail("ncclCommInitRank", e);
  *comm_out = c;
  return D2FE_OK;
}
int d2fe_rccl_comm_destroy(void* comm) {
  if (!comm) return D2FE_OK;
  if (!g_rccl.lib) return ctx_fail(D2FE_ERR_INVALID, "no RCCL library is loaded");
  const int e = g_rccl.CommDestroy(comm);
  return e ? rccl_fail("ncclCommDestroy", e) : D2FE_OK;
}

void d2fe_exchange_default_config(d2fe_exchange_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->struct_size = (int32_t)sizeof(*c);
  c->world = 1; c->rank = 0; c->wire = D2FE_WIRE_FP32; c->loopback = 0; c->slots = 4; c->gate_thres = 0.8; c->ratio = 0.8; c->own_stream = 1; c->timing = 0;
}

void d2fe_exchange_destroy(d2fe_exchange x) {
  if (!x) return;
  if (x->h) (void)hipSetDevice(x->h->cfg.device_id);
  for (auto& S : x->slots) {
    if (S.busy && S.done) (void)hipEventSynchronize(S.done);
    for (void* q : {(void*)S.d_blocks, (void*)S.d_blocks_q, (void*)S.d_gath, (void*)S.d_gath_q, (void*)S.d_a_cnt, (void*)S.d_b_cnt, (void*)S.d_out})
      if (q) (void)hipFree(q);
    if (S.pin) (void)hipHostFree(S.pin);
    for (auto& e : S.ev) if (e) (void)hipEventDestroy(e);
    if (S.done) (void)hipEventDestroy(S.done);
  }
  for (void* q : {(void*)x->d_q_frame, (void*)x->d_rem_blk, (void*)x->d_a_off, (void*)x->d_b_off})
    if (q) (void)hipFree(q);
  if (x->own) (void)hipStreamDestroy(x->own);
  delete x;
}

int d2fe_exchange_create(d2fe_pipe p, void* nccl_comm, const d2fe_exchange_config* cfg_in, d2fe_exchange* out) {
  if (!p || !cfg_in || !out) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  *out = nullptr;
  d2fe_exchange_config cfg;
  d2fe_exchange_default_config(&cfg);
  memcpy(&cfg, cfg_in, (size_t)std::min<int32_t>(cfg_in->struct_size > 0 ? cfg_in->struct_size : (int32_t)sizeof(cfg), (int32_t)sizeof(cfg)));
  if (cfg.world < 1 || cfg.rank < 0 || cfg.rank >= cfg.world || cfg.slots < 1 || cfg.slots > 64 || cfg.wire < 0 || cfg.wire > 2) return ctx_fail(D2FE_ERR_INVALID, "bad exchange configuration");
  if (cfg.world == 1 && !cfg.loopback) return ctx_fail(D2FE_ERR_INVALID, "one rank and no loopback: nothing to exchange");
  if (!nccl_comm && !cfg.all_gather) return ctx_fail(D2FE_ERR_INVALID, "neither an RCCL communicator nor an all-gather callback");
  if (nccl_comm) { const int rc = rccl_load(nullptr); if (rc) return rc; }
  int pf = 0, pcap = 0, pdim = 0, pg = 0;
  {
    const int rc = d2fe_pipe_geometry(p, &pf, &pcap, &pdim, &pg);
    if (rc) return rc;
  }
  if (pdim != 256) return ctx_fail(D2FE_ERR_UNSUPPORTED, "the exchange blocks hold 256-float descriptors (a pipe with descriptor PCA cannot be exchanged)");
  auto* x = new (std::nothrow) d2fe_exchange_s();
  if (!x) return ctx_fail(D2FE_ERR_HIP, "out of memory");
  struct Guard { d2fe_exchange_s* x; bool ok = false; ~Guard() { if (!ok) d2fe_exchange_destroy(x); } } guard{x};
  x->p = p; x->h = d2fe_pipe_handle(p); x->cfg = cfg; x->comm = nccl_comm;
  x->F = pf; x->cap = pcap; x->G = pg;
  x->int8 = cfg.wire != D2FE_WIRE_FP32;
  x->BLK = d2fe_block_words(pcap, pg); x->BLKB = d2fe_block_bytes_int8(pcap, pg);
  x->n_off = d2fe_block_field_offset(pcap, pg, 4); x->g_off = d2fe_block_field_offset(pcap, pg, 3);
  if (x->int8 && (pcap * 256 + pg + pcap * 8) % 4) return ctx_fail(D2FE_ERR_UNSUPPORTED, "int8 blocks of this capacity / NetVLAD size do not keep their count word aligned");
  x->own_n_word = x->int8 ? (pcap * 256 + pg + pcap * 8) / 4 : x->n_off;
  HIP_TRY(hipSetDevice(x->h->cfg.device_id));
  // pair layout: local left frame f against the frame with the same time index of every other rank, rank-major (swarm.remote_pair_layout)
  std::vector<int32_t> a_off, b_off, qf, rb;
  const int rows = x->BLK / 256;
  for (int r = 0; r < cfg.world; ++r) {
    if (r == cfg.rank && !cfg.loopback) continue;
    for (int f = 0; f < pf; ++f) { a_off.push_back(f * pcap); b_off.push_back((r * pf + f) * rows); qf.push_back(f); rb.push_back(r * pf + f); }
  }
  x->NR = (int)a_off.size();
  const int NR = x->NR;
  auto up = [&](const std::vector<int32_t>& v, int32_t** d) -> int {
    HIP_TRY(hipMalloc(d, sizeof(int32_t) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) HIP_TRY(hipMemcpy(*d, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
    return D2FE_OK;
  };
  int rc = up(a_off, &x->d_a_off); rc = rc ? rc : up(b_off, &x->d_b_off); rc = rc ? rc : up(qf, &x->d_q_frame); rc = rc ? rc : up(rb, &x->d_rem_blk);
  if (rc) return rc;
  auto up64 = [](size_t w) { return (w + 63) / 64 * 64; };
  size_t o = 0;
  x->o_mq = o; o += up64((size_t)NR * pcap); x->o_mt = o; o += up64((size_t)NR * pcap); x->o_md = o; o += up64((size_t)NR * pcap);
  x->o_mn = o; o += up64(NR); x->o_pass = o; o += up64(NR); x->o_sims = o; o += up64(NR); x->o_np = o; o += 64;
  x->out_words = o;
  x->slots.resize(cfg.slots);
  for (auto& S : x->slots) {
    const size_t nb = (size_t)pf * x->BLK, ng = (size_t)cfg.world * pf * x->BLK;
    HIP_TRY(hipMalloc(&S.d_blocks, sizeof(float) * nb)); HIP_TRY(hipMemset(S.d_blocks, 0, sizeof(float) * nb));
    HIP_TRY(hipMalloc(&S.d_gath, sizeof(float) * ng)); HIP_TRY(hipMemset(S.d_gath, 0, sizeof(float) * ng));
    if (x->int8) {
      HIP_TRY(hipMalloc(&S.d_blocks_q, (size_t)pf * x->BLKB)); HIP_TRY(hipMemset(S.d_blocks_q, 0, (size_t)pf * x->BLKB));
      HIP_TRY(hipMalloc(&S.d_gath_q, (size_t)cfg.world * pf * x->BLKB)); HIP_TRY(hipMemset(S.d_gath_q, 0, (size_t)cfg.world * pf * x->BLKB));
    }
    HIP_TRY(hipMalloc(&S.d_a_cnt, sizeof(int32_t) * std::max(NR, 1))); HIP_TRY(hipMalloc(&S.d_b_cnt, sizeof(int32_t) * std::max(NR, 1)));
    HIP_TRY(hipMalloc(&S.d_out, sizeof(float) * x->out_words)); HIP_TRY(hipMemset(S.d_out, 0, sizeof(float) * x->out_words));
    HIP_TRY(hipHostMalloc(&S.pin, sizeof(float) * x->out_words, hipHostMallocDefault));
    memset(S.pin, 0, sizeof(float) * x->out_words);
    if (cfg.timing) for (auto& e : S.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
  }
  if (cfg.own_stream) HIP_TRY(hipStreamCreateWithFlags(&x->own, hipStreamNonBlocking));
  HIP_TRY(hipDeviceSynchronize());
  guard.ok = true;
  *out = x;
  return D2FE_OK;
}

int d2fe_exchange_pairs(d2fe_exchange x) { return x ? x->NR : ctx_fail(D2FE_ERR_INVALID, "null exchange"); }
int d2fe_exchange_block_bytes(d2fe_exchange x) { return x ? (x->int8 ? x->BLKB : 4 * x->BLK) : ctx_fail(D2FE_ERR_INVALID, "null exchange"); }
void* d2fe_exchange_stream(d2fe_exchange x) { return x ? x->own : nullptr; }

int d2fe_exchange_enqueue(d2fe_exchange x, int64_t ticket, int slot) {
  if (!x || slot < 0 || slot >= (int)x->slots.size()) return ctx_fail(D2FE_ERR_INVALID, "bad argument");
  auto& S = x->slots[slot];
  if (S.busy) return ctx_fail(D2FE_ERR_NOT_READY, "this slot's previous exchange has not been collected");
  HIP_TRY(hipSetDevice(x->h->cfg.device_id));
  void* lane_stream = nullptr;
  if (!x->own) { const int rc = d2fe_pipe_lane_stream(x->p, ticket, &lane_stream); if (rc) return rc; }
  hipStream_t st = x->own ? x->own : static_cast<hipStream_t>(lane_stream);
  d2fe_pipe_device_result v{};
  int rc = d2fe_pipe_device_view(x->p, ticket, st, &v);
  if (rc) return rc;
  // from here on the view must be released whatever happens (a block with an outstanding view ends the pipe 2 * lanes passes later)
  auto run = [&]() -> int {
    if (v.frames != x->F || v.cap != x->cap || v.desc_dim != 256) return ctx_fail(D2FE_ERR_INVALID, "the pipe's geometry changed under the exchange");
    const bool tm = x->cfg.timing != 0;
    auto mark = [&](int i) -> int { if (tm) HIP_TRY(hipEventRecord(S.ev[i], st)); return D2FE_OK; };
    int r = mark(0); if (r) return r;
    const int F = x->F, cap = x->cap, G = x->G, W = x->cfg.world;
    const void* send; void* recv; size_t bytes;
    if (x->int8) {
      r = d2fe_pack_blocks_int8_device(x->h, v.d_desc, v.d_kps_xy, v.d_n_kp, v.d_netvlad, 0, 1, F, cap, G, S.d_blocks_q, st); if (r) return r;
      send = S.d_blocks_q; recv = S.d_gath_q; bytes = (size_t)F * x->BLKB;
    } else {
      r = d2fe_pack_blocks_device(x->h, v.d_desc, v.d_kps_xy, v.d_scores, v.d_n_kp, v.d_netvlad, 0, 1, F, cap, G, S.d_blocks, st); if (r) return r;
      send = S.d_blocks; recv = S.d_gath; bytes = sizeof(float) * (size_t)F * x->BLK;
    }
    r = mark(1); if (r) return r;
    if (x->comm) {
      const int e = g_rccl.AllGather(send, recv, bytes, /* ncclInt8 */ 0, x->comm, st);
      if (e) return rccl_fail("ncclAllGather", e);
    } else {
      r = x->cfg.all_gather(x->cfg.all_gather_user, send, recv, bytes, st);
      if (r) return ctx_fail(D2FE_ERR_HIP, "the all-gather callback failed (" + std::to_string(r) + ")");
    }
    r = mark(2); if (r) return r;
    if (x->int8) { r = d2fe_unpack_blocks_int8_device(x->h, S.d_gath_q, W * F, cap, G, x->cfg.wire == D2FE_WIRE_INT8_RENORM256 ? 1 : 0, S.d_gath, st); if (r) return r; }
    int32_t* O = reinterpret_cast<int32_t*>(S.d_out);
    if (x->NR > 0) {
      const int32_t* own = x->int8 ? reinterpret_cast<const int32_t*>(S.d_blocks_q) : reinterpret_cast<const int32_t*>(S.d_blocks);
      hipLaunchKernelGGL(exchange_counts_kernel, dim3((x->NR + 63) / 64), dim3(64), 0, st, own + x->own_n_word, x->int8 ? x->BLKB / 4 : x->BLK,
                         reinterpret_cast<const int32_t*>(S.d_gath), x->BLK, x->n_off, x->d_q_frame, x->d_rem_blk, x->NR, S.d_a_cnt, S.d_b_cnt);
      HIP_TRY(hipGetLastError());
      if (G) {
        HIP_TRY(hipMemsetAsync(O + x->o_np, 0, sizeof(int32_t), st));
        // all-to-all mode (BASELINE configs[3] / [4]): every pair is matched; the reference's gate is evaluated and counted (gate_pass / gate_n)
        r = d2fe_gate_pairs_device(x->h, v.d_netvlad, (size_t)G, S.d_gath + x->g_off, (size_t)x->BLK, G, x->d_q_frame, x->d_rem_blk, x->NR, x->cfg.gate_thres, nullptr,
                                   O + x->o_pass, S.d_out + x->o_sims, O + x->o_np, st);
        if (r) return r;
      }
    }
    r = mark(3); if (r) return r;
    if (x->NR > 0) {
      d2fe_match_batch mb{};
      mb.d_a = v.d_desc; mb.d_b = S.d_gath; mb.d_a_off = x->d_a_off; mb.d_b_off = x->d_b_off; mb.d_a_cnt = S.d_a_cnt; mb.d_b_cnt = S.d_b_cnt;
      mb.npairs = x->NR; mb.dim = 256; mb.max_n = cap; mb.mode = 0; mb.ratio = x->cfg.ratio; mb.radius = -1.0;
      mb.d_q_idx = O + x->o_mq; mb.d_t_idx = O + x->o_mt; mb.d_dist = S.d_out + x->o_md; mb.d_n_out = O + x->o_mn;
      r = d2fe_match_batch_device(x->h, &mb, st); if (r) return r;
    }
    return mark(4);
  };
  rc = run();
  const int rr = d2fe_pipe_device_release(x->p, ticket, st);
  if (rc) return rc;
  if (rr) return rr;
  HIP_TRY(hipMemcpyAsync(S.pin, S.d_out, sizeof(float) * x->out_words, hipMemcpyDeviceToHost, st));
  if (x->cfg.timing) HIP_TRY(hipEventRecord(S.ev[5], st));
  HIP_TRY(hipEventRecord(S.done, st));
  S.busy = true; S.ticket = ticket;
  return D2FE_OK;
}

int d2fe_exchange_collect(d2fe_exchange x, int slot, d2fe_exchange_result* out) {
  if (!x || !out || slot < 0 || slot >= (int)x->slots.size()) return ctx_fail(D2FE_ERR_INVALID, "bad argument");
  auto& S = x->slots[slot];
  if (!S.busy) return ctx_fail(D2FE_ERR_INVALID, "nothing was enqueued on this slot");
  HIP_TRY(hipSetDevice(x->h->cfg.device_id));
  HIP_TRY(hipEventSynchronize(S.done));
  memset(out, 0, sizeof(*out));
  const int32_t* I = reinterpret_cast<const int32_t*>(S.pin);
  out->ticket = S.ticket; out->npairs = x->NR; out->cap = x->cap;
  out->q_idx = I + x->o_mq; out->t_idx = I + x->o_mt; out->dist = S.pin + x->o_md; out->n_match = I + x->o_mn;
  out->gate_pass = x->G ? I + x->o_pass : nullptr; out->gate_sims = x->G ? S.pin + x->o_sims : nullptr; out->gate_n = x->G ? I[x->o_np] : 0;
  if (x->cfg.timing)
    for (int i = 0; i < 5; ++i) { float ms = 0.f; if (hipEventElapsedTime(&ms, S.ev[i], S.ev[i + 1]) == hipSuccess) out->phase_ms[i] = ms; }
  S.busy = false;
  return D2FE_OK;
}

}  // extern "C"

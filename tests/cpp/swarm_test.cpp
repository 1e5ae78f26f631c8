// swarm_test.cpp -- the cross-agent exchange of INTEGRATION.md section 3a as a D2SLAM maintainer would write it: plain C++ (g++), the C ABI of
// include/d2fe.h, the HIP runtime API for buffers / streams and RCCL (librccl: ncclAllGather) for the exchange -- no Python, no torch.
//   d2fe_pack_blocks_device -> ncclAllGather -> d2fe_gate_pairs_device -> d2fe_match_batch_device (b side inside the gathered blocks)
// replaces the LCM broadcast + tracking gate of the reference (loop_net.cpp:24-87, d2featuretracker.cpp:185-203,237-310).
// Runs as a ONE-rank communicator on a 1-GPU box (RCCL refuses two ranks per device); tests/test_cpp_swarm.py compares every output with the
// Python path (d2slam_amd/swarm.py) and the oracle.  usage: swarm_test <in.bin> <out.bin>
//        swarm_test --two-devices : the same sequence as TWO ranks of one process on devices 0 and 1 (ncclCommInitAll + grouped ncclAllGather), self-checked;
//                                   exit code 77 = fewer than two devices visible (a clean skip on the 1-GPU boxes)
//   in : int32 F, cap, G;  float desc[F][cap][256], kps[F][cap][2], scores[F][cap];  int32 n[F];  float netvlad[F][G];  double thres
//   out: int32-counted vectors: gathered blocks (float), pass (int32), sims (float), n_pass (int32), q, t (int32), dist (float), n_match (int32)
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "d2fe.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(e_)); return 4; } } while (0)
#define CHECK_D2FE(x) do { int e_ = (x); if (e_ != D2FE_OK) { fprintf(stderr, "%s: %d %s\n", #x, e_, d2fe_last_error()); return 5; } } while (0)

template <class T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <class T> static void wr(FILE* f, const std::vector<T>& v) { int32_t n = (int32_t)v.size(); fwrite(&n, 4, 1, f); fwrite(v.data(), sizeof(T), v.size(), f); }

// ---- two ranks, two devices, one process: what an N-agent swarm does on every agent, with a REAL (N > 1) RCCL collective in the middle ---------------------
// Inputs are generated here (seeded): both "agents" see the same scene (a shared descriptor set, permuted, with per-agent noise), so cross-agent matches exist.
// Checked: every device's gathered buffer == the concatenation of both agents' packed blocks, bit for bit; the gate similarities and the match counts are
// symmetric between the agents (sim(a,b) == sim(b,a); matchKNN's mutual ratio test gives |matches(a->b)| == |matches(b->a)|).
static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float unif(unsigned& s) { return (float)lcg(s) / 16777216.0f - 0.5f; }
static void unit_rows(std::vector<float>& v, int rows, int dim) {
  for (int r = 0; r < rows; ++r) {
    double n = 0; for (int k = 0; k < dim; ++k) n += (double)v[(size_t)r * dim + k] * v[(size_t)r * dim + k];
    const float inv = (float)(1.0 / std::sqrt(n > 0 ? n : 1.0));
    for (int k = 0; k < dim; ++k) v[(size_t)r * dim + k] *= inv;
  }
}
static int two_devices() {
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 2) { printf("swarm_test SKIP: --two-devices needs 2 visible devices, found %d\n", ndev); return 77; }
  const int world = 2, F = 3, cap = 64, G = 256;
  const int BLK = d2fe_block_words(cap, G), off_nv = d2fe_block_field_offset(cap, G, 3), off_n = d2fe_block_field_offset(cap, G, 4);
  int devs[2] = {0, 1};
  ncclComm_t comm[2];
  CHECK_NCCL(ncclCommInitAll(comm, world, devs));
  struct Agent {
    hipStream_t st = nullptr; d2fe_handle h = nullptr;
    float *d_desc = nullptr, *d_kps = nullptr, *d_scores = nullptr, *d_nv = nullptr, *d_blocks = nullptr, *d_gath = nullptr, *d_sims = nullptr, *d_dist = nullptr;
    int32_t *d_cnt = nullptr, *d_pq = nullptr, *d_pdb = nullptr, *d_pass = nullptr, *d_np = nullptr, *d_aoff = nullptr, *d_boff = nullptr, *d_acnt = nullptr, *d_bcnt = nullptr,
            *d_q = nullptr, *d_t = nullptr, *d_nm = nullptr;
    std::vector<float> blocks, gath, sims; std::vector<int32_t> cnt, nm;
  } A[2];
  // the shared scene
  unsigned seed = 12345u;
  std::vector<float> base((size_t)cap * 256), gbase(G);
  for (auto& x : base) x = unif(seed);
  for (auto& x : gbase) x = unif(seed);
  unit_rows(base, cap, 256);
  const int NP = F;                                   // local frame f against the OTHER agent's frame f
  for (int r = 0; r < world; ++r) {
    Agent& a = A[r];
    CHECK_HIP(hipSetDevice(devs[r]));
    CHECK_HIP(hipStreamCreateWithFlags(&a.st, hipStreamNonBlocking));
    d2fe_config cfg; d2fe_default_config(&cfg);
    cfg.device_id = devs[r]; cfg.max_width = 64; cfg.max_height = 64; cfg.max_batch = 1; cfg.max_keypoints = cap;
    CHECK_D2FE(d2fe_create(&cfg, &a.h));
    std::vector<float> desc((size_t)F * cap * 256), kps((size_t)F * cap * 2), scores((size_t)F * cap), nv((size_t)F * G);
    a.cnt.resize(F);
    for (int f = 0; f < F; ++f) {
      a.cnt[f] = cap - 5 * f - r;
      for (int i = 0; i < cap; ++i) {
        const int src = (i * 7 + 3 * f + 11 * r) % cap;      // a permutation of the shared rows (7 and 64 are coprime)
        for (int k = 0; k < 256; ++k) desc[((size_t)f * cap + i) * 256 + k] = base[(size_t)src * 256 + k] + 0.02f * unif(seed);
        kps[((size_t)f * cap + i) * 2] = (float)(lcg(seed) % 640); kps[((size_t)f * cap + i) * 2 + 1] = (float)(lcg(seed) % 480);
        scores[(size_t)f * cap + i] = unif(seed) + 0.5f;
      }
      for (int k = 0; k < G; ++k) nv[(size_t)f * G + k] = gbase[k] + 0.05f * unif(seed);
    }
    unit_rows(desc, F * cap, 256); unit_rows(nv, F, G);
    CHECK_HIP(hipMalloc(&a.d_desc, desc.size() * 4)); CHECK_HIP(hipMalloc(&a.d_kps, kps.size() * 4)); CHECK_HIP(hipMalloc(&a.d_scores, scores.size() * 4));
    CHECK_HIP(hipMalloc(&a.d_nv, nv.size() * 4)); CHECK_HIP(hipMalloc(&a.d_cnt, F * 4)); CHECK_HIP(hipMalloc(&a.d_blocks, (size_t)F * BLK * 4));
    CHECK_HIP(hipMalloc(&a.d_gath, (size_t)world * F * BLK * 4));
    CHECK_HIP(hipMalloc(&a.d_pq, NP * 4)); CHECK_HIP(hipMalloc(&a.d_pdb, NP * 4)); CHECK_HIP(hipMalloc(&a.d_pass, NP * 4)); CHECK_HIP(hipMalloc(&a.d_sims, NP * 4));
    CHECK_HIP(hipMalloc(&a.d_np, 4)); CHECK_HIP(hipMalloc(&a.d_aoff, NP * 4)); CHECK_HIP(hipMalloc(&a.d_boff, NP * 4)); CHECK_HIP(hipMalloc(&a.d_acnt, NP * 4));
    CHECK_HIP(hipMalloc(&a.d_bcnt, NP * 4)); CHECK_HIP(hipMalloc(&a.d_q, (size_t)NP * cap * 4)); CHECK_HIP(hipMalloc(&a.d_t, (size_t)NP * cap * 4));
    CHECK_HIP(hipMalloc(&a.d_dist, (size_t)NP * cap * 4)); CHECK_HIP(hipMalloc(&a.d_nm, NP * 4));
    CHECK_HIP(hipMemcpyAsync(a.d_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_kps, kps.data(), kps.size() * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_scores, scores.data(), scores.size() * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_nv, nv.data(), nv.size() * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_cnt, a.cnt.data(), F * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemsetAsync(a.d_np, 0, 4, a.st));
    CHECK_HIP(hipStreamSynchronize(a.st));      // the host vectors above go out of scope
    CHECK_D2FE(d2fe_pack_blocks_device(a.h, a.d_desc, a.d_kps, a.d_scores, a.d_cnt, a.d_nv, 0, 1, F, cap, G, a.d_blocks, a.st));
  }
  // ONE collective per agent and step; two ranks driven by one thread must be grouped
  CHECK_NCCL(ncclGroupStart());
  for (int r = 0; r < world; ++r) { CHECK_HIP(hipSetDevice(devs[r])); CHECK_NCCL(ncclAllGather(A[r].d_blocks, A[r].d_gath, (size_t)F * BLK, ncclFloat, comm[r], A[r].st)); }
  CHECK_NCCL(ncclGroupEnd());
  for (int r = 0; r < world; ++r) {
    Agent& a = A[r];
    CHECK_HIP(hipSetDevice(devs[r]));
    a.blocks.resize((size_t)F * BLK); a.gath.resize((size_t)world * F * BLK);
    CHECK_HIP(hipMemcpyAsync(a.blocks.data(), a.d_blocks, a.blocks.size() * 4, hipMemcpyDeviceToHost, a.st));
    CHECK_HIP(hipMemcpyAsync(a.gath.data(), a.d_gath, a.gath.size() * 4, hipMemcpyDeviceToHost, a.st));
    CHECK_HIP(hipStreamSynchronize(a.st));
  }
  for (int r = 0; r < world; ++r)
    for (int src = 0; src < world; ++src)
      if (memcmp(A[r].gath.data() + (size_t)src * F * BLK, A[src].blocks.data(), (size_t)F * BLK * 4)) { fprintf(stderr, "rank %d: gathered blocks of rank %d differ from what it packed\n", r, src); return 6; }
  for (int r = 0; r < world; ++r) {
    Agent& a = A[r];
    const int o = 1 - r;
    CHECK_HIP(hipSetDevice(devs[r]));
    std::vector<int32_t> pq(NP), pdb(NP), aoff(NP), boff(NP), acnt(NP), bcnt(NP);
    for (int f = 0; f < F; ++f) {
      pq[f] = f; pdb[f] = o * F + f; aoff[f] = f * cap; boff[f] = (o * F + f) * (BLK / 256);
      acnt[f] = a.cnt[f]; bcnt[f] = reinterpret_cast<const int32_t*>(a.gath.data())[(size_t)pdb[f] * BLK + off_n];
      if (bcnt[f] != A[o].cnt[f]) { fprintf(stderr, "count word of a gathered block is wrong\n"); return 6; }
    }
    CHECK_HIP(hipMemcpyAsync(a.d_pq, pq.data(), NP * 4, hipMemcpyHostToDevice, a.st)); CHECK_HIP(hipMemcpyAsync(a.d_pdb, pdb.data(), NP * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_aoff, aoff.data(), NP * 4, hipMemcpyHostToDevice, a.st)); CHECK_HIP(hipMemcpyAsync(a.d_boff, boff.data(), NP * 4, hipMemcpyHostToDevice, a.st));
    CHECK_HIP(hipMemcpyAsync(a.d_acnt, acnt.data(), NP * 4, hipMemcpyHostToDevice, a.st)); CHECK_HIP(hipMemcpyAsync(a.d_bcnt, bcnt.data(), NP * 4, hipMemcpyHostToDevice, a.st));
    CHECK_D2FE(d2fe_gate_pairs_device(a.h, a.d_nv, (size_t)G, a.d_gath + off_nv, (size_t)BLK, G, a.d_pq, a.d_pdb, NP, 0.5, nullptr, a.d_pass, a.d_sims, a.d_np, a.st));
    // a side: the local descriptors; b side: in place inside the gathered blocks (a separate pool: d_b = the gathered buffer)
    d2fe_match_batch mb = {a.d_desc, a.d_gath, nullptr, nullptr, a.d_aoff, a.d_boff, a.d_acnt, a.d_bcnt, NP, 256, cap, 0, 0.8, -1.0, a.d_q, a.d_t, a.d_dist, a.d_nm};
    CHECK_D2FE(d2fe_match_batch_device(a.h, &mb, a.st));
    a.sims.resize(NP); a.nm.resize(NP);
    CHECK_HIP(hipMemcpyAsync(a.sims.data(), a.d_sims, NP * 4, hipMemcpyDeviceToHost, a.st));
    CHECK_HIP(hipMemcpyAsync(a.nm.data(), a.d_nm, NP * 4, hipMemcpyDeviceToHost, a.st));
    CHECK_HIP(hipStreamSynchronize(a.st));
  }
  int total = 0;
  for (int f = 0; f < F; ++f) {
    if (std::fabs(A[0].sims[f] - A[1].sims[f]) > 1e-5f) { fprintf(stderr, "frame %d: gate similarity differs between the agents (%g vs %g)\n", f, A[0].sims[f], A[1].sims[f]); return 6; }
    if (A[0].nm[f] != A[1].nm[f]) { fprintf(stderr, "frame %d: |matches(a->b)| = %d but |matches(b->a)| = %d\n", f, A[0].nm[f], A[1].nm[f]); return 6; }
    total += A[0].nm[f];
  }
  if (total == 0) { fprintf(stderr, "no cross-agent matches although both agents see the same scene\n"); return 6; }
  for (int r = 0; r < world; ++r) { CHECK_HIP(hipSetDevice(devs[r])); d2fe_destroy(A[r].h); ncclCommDestroy(comm[r]); }
  printf("swarm_test OK: two devices, %d frames per agent, %d cross-agent matches, gathered buffers identical on both\n", F, total);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 2 && std::string(argv[1]) == "--two-devices") return two_devices();
  if (argc != 3) { fprintf(stderr, "usage: swarm_test <in.bin> <out.bin> | swarm_test --two-devices\n"); return 2; }
  FILE* fi = fopen(argv[1], "rb");
  if (!fi) return 2;
  int32_t F, cap, G;
  if (!rd(fi, &F, 1) || !rd(fi, &cap, 1) || !rd(fi, &G, 1)) return 2;
  std::vector<float> desc((size_t)F * cap * 256), kps((size_t)F * cap * 2), scores((size_t)F * cap), nv((size_t)F * G);
  std::vector<int32_t> cnt(F);
  double thres = 0;
  if (!rd(fi, desc.data(), desc.size()) || !rd(fi, kps.data(), kps.size()) || !rd(fi, scores.data(), scores.size()) || !rd(fi, cnt.data(), cnt.size()) ||
      !rd(fi, nv.data(), nv.size()) || !rd(fi, &thres, 1)) return 2;
  fclose(fi);

  CHECK_HIP(hipSetDevice(0));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  // one-rank communicator: what `ncclCommInitRank(&comm, world, id, rank)` is on every agent of a swarm
  const int world = 1, rank = 0;
  ncclUniqueId id;
  CHECK_NCCL(ncclGetUniqueId(&id));
  ncclComm_t comm;
  CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));

  d2fe_config cfg;
  d2fe_default_config(&cfg);
  cfg.max_width = 64; cfg.max_height = 64; cfg.max_batch = 1; cfg.max_keypoints = cap;
  d2fe_handle h = nullptr;
  CHECK_D2FE(d2fe_create(&cfg, &h));

  const int BLK = d2fe_block_words(cap, G);
  const int off_nv = d2fe_block_field_offset(cap, G, 3), off_n = d2fe_block_field_offset(cap, G, 4);
  // one pool of 256-float rows: [F*cap) the local descriptors, then the gathered blocks (world * F blocks, each a multiple of 256 words)
  const size_t local_rows = (size_t)F * cap, pool_words = local_rows * 256 + (size_t)world * F * BLK;
  float *d_pool, *d_kps, *d_scores, *d_nv, *d_blocks, *d_sims, *d_dist;
  int32_t *d_cnt, *d_pair_q, *d_pair_db, *d_pass, *d_npass, *d_aoff, *d_boff, *d_acnt, *d_bcnt, *d_q, *d_t, *d_nm;
  CHECK_HIP(hipMalloc(&d_pool, pool_words * 4)); CHECK_HIP(hipMalloc(&d_kps, kps.size() * 4)); CHECK_HIP(hipMalloc(&d_scores, scores.size() * 4));
  CHECK_HIP(hipMalloc(&d_nv, nv.size() * 4)); CHECK_HIP(hipMalloc(&d_blocks, (size_t)F * BLK * 4)); CHECK_HIP(hipMalloc(&d_cnt, F * 4));
  float* d_gath = d_pool + local_rows * 256;
  // pairs: local frame f against the gathered block g of every OTHER frame (with one rank the "remote" blocks are this rank's own)
  std::vector<int32_t> pq, pdb, aoff, boff;
  for (int f = 0; f < F; ++f)
    for (int g = 0; g < world * F; ++g)
      if (g != rank * F + f) { pq.push_back(f); pdb.push_back(g); aoff.push_back(f * cap); boff.push_back((int32_t)(local_rows + (size_t)g * (BLK / 256))); }
  const int NP = (int)pq.size();
  CHECK_HIP(hipMalloc(&d_pair_q, NP * 4)); CHECK_HIP(hipMalloc(&d_pair_db, NP * 4)); CHECK_HIP(hipMalloc(&d_pass, NP * 4)); CHECK_HIP(hipMalloc(&d_sims, NP * 4));
  CHECK_HIP(hipMalloc(&d_npass, 4)); CHECK_HIP(hipMalloc(&d_aoff, NP * 4)); CHECK_HIP(hipMalloc(&d_boff, NP * 4)); CHECK_HIP(hipMalloc(&d_acnt, NP * 4));
  CHECK_HIP(hipMalloc(&d_bcnt, NP * 4)); CHECK_HIP(hipMalloc(&d_q, (size_t)NP * cap * 4)); CHECK_HIP(hipMalloc(&d_t, (size_t)NP * cap * 4));
  CHECK_HIP(hipMalloc(&d_dist, (size_t)NP * cap * 4)); CHECK_HIP(hipMalloc(&d_nm, NP * 4));
  CHECK_HIP(hipMemcpyAsync(d_pool, desc.data(), desc.size() * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_kps, kps.data(), kps.size() * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_scores, scores.data(), scores.size() * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_nv, nv.data(), nv.size() * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_cnt, cnt.data(), F * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_pair_q, pq.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_pair_db, pdb.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_aoff, aoff.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_boff, boff.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemsetAsync(d_npass, 0, 4, stream));

  // ---- the sequence of INTEGRATION.md section 3a --------------------------------------------------------------------------------------
  CHECK_D2FE(d2fe_pack_blocks_device(h, d_pool, d_kps, d_scores, d_cnt, d_nv, /*row0*/0, /*row_step*/1, F, cap, G, d_blocks, stream));   // one block per frame
  CHECK_NCCL(ncclAllGather(d_blocks, d_gath, (size_t)F * BLK, ncclFloat, comm, stream));                                                  // ONE collective
  // matcher counts: a side = the local frame's, b side = the n word of the gathered block (host bookkeeping of a few ints per pair: a gather kernel
  // or the reference's message header would deliver them; here they are read back once)
  std::vector<float> gath((size_t)world * F * BLK);
  CHECK_HIP(hipMemcpyAsync(gath.data(), d_gath, gath.size() * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  std::vector<int32_t> acnt(NP), bcnt(NP);
  for (int p = 0; p < NP; ++p) { acnt[p] = cnt[pq[p]]; bcnt[p] = reinterpret_cast<const int32_t*>(gath.data())[(size_t)pdb[p] * BLK + off_n]; }
  CHECK_HIP(hipMemcpyAsync(d_acnt, acnt.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_bcnt, bcnt.data(), NP * 4, hipMemcpyHostToDevice, stream));
  CHECK_D2FE(d2fe_gate_pairs_device(h, d_nv, (size_t)G, d_gath + off_nv, (size_t)BLK, G, d_pair_q, d_pair_db, NP, thres,
                                    d_acnt /* rejected pairs -> 0 */, d_pass, d_sims, d_npass, stream));
  d2fe_match_batch mb = {d_pool, d_pool, nullptr, nullptr, d_aoff, d_boff, d_acnt, d_bcnt, NP, 256, cap, 0, 0.8, -1.0, d_q, d_t, d_dist, d_nm};
  CHECK_D2FE(d2fe_match_batch_device(h, &mb, stream));   // b-side offsets address the descriptors INSIDE the gathered blocks

  std::vector<int32_t> pass(NP), npass(1), q((size_t)NP * cap), t((size_t)NP * cap), nm(NP);
  std::vector<float> sims(NP), dist((size_t)NP * cap);
  CHECK_HIP(hipMemcpyAsync(pass.data(), d_pass, NP * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(sims.data(), d_sims, NP * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(npass.data(), d_npass, 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(q.data(), d_q, q.size() * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(t.data(), d_t, t.size() * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(dist.data(), d_dist, dist.size() * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(nm.data(), d_nm, NP * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  FILE* fo = fopen(argv[2], "wb");
  if (!fo) return 2;
  wr(fo, gath); wr(fo, pass); wr(fo, sims); wr(fo, npass); wr(fo, q); wr(fo, t); wr(fo, dist); wr(fo, nm);
  fclose(fo);
  d2fe_destroy(h);
  ncclCommDestroy(comm);
  for (void* p : {(void*)d_pool, (void*)d_kps, (void*)d_scores, (void*)d_nv, (void*)d_blocks, (void*)d_cnt, (void*)d_pair_q, (void*)d_pair_db, (void*)d_pass, (void*)d_sims,
                  (void*)d_npass, (void*)d_aoff, (void*)d_boff, (void*)d_acnt, (void*)d_bcnt, (void*)d_q, (void*)d_t, (void*)d_dist, (void*)d_nm})
    (void)hipFree(p);
  (void)hipStreamDestroy(stream);
  printf("swarm_test OK: %d frames, %d pairs, %d pass the NetVLAD gate\n", F, NP, npass[0]);
  return 0;
}

// exchange_test.cpp -- the cross-agent exchange BEHIND a frames-in-flight pipe as D2SLAM's C++ would drive it (round 6): plain C++ (g++), only the C ABI of
// include/d2fe.h -- d2fe_pipe_* for the frames, d2fe_rccl_* for the communicator (librccl is loaded by the LIBRARY with dlopen: this program does not link it),
// d2fe_exchange_* for the sequence (pack -> ncclAllGather -> gate -> remote matchKNN -> D2H, queued by the library on the producing lane's stream).  No Python,
// no torch, no HIP call of its own.  Weights come from D2FW containers (include/d2fe_weights_file.hpp), frames from a raw file.
// Replaces the LCM broadcast + trackRemoteFrames of the reference (loop_net.cpp:24-87, d2featuretracker.cpp:185-203,237-310).
// ONE rank with loopback (the rank's own blocks as the remote agent): what a 1-GPU box can run of it; tests/test_cpp_swarm.py checks every output.
//   usage: exchange_test <sp.d2fw> <nv.d2fw> <frames.bin> <out.bin> <lanes> <wire 0|1|2> <own_stream 0|1>
//   frames.bin: int32 steps, F, H, W, cap; then steps x { left[F][H][W], right[F][H][W] } u8
//   out.bin   : per step: int32 n_kp[2F], int32 npairs, int32 n_match[npairs], int32 gate_pass[npairs], int32 q[npairs][cap], int32 t[npairs][cap], float dist[npairs][cap]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "d2fe.h"
#include "d2fe_weights_file.hpp"

#define CHECK(x) do { int e_ = (x); if (e_ != D2FE_OK) { fprintf(stderr, "%s: %d %s\n", #x, e_, d2fe_last_error()); return 5; } } while (0)

int main(int argc, char** argv) {
  if (argc != 8) { fprintf(stderr, "usage: exchange_test <sp.d2fw> <nv.d2fw> <frames.bin> <out.bin> <lanes> <wire> <own_stream>\n"); return 2; }
  const int lanes = atoi(argv[5]), wire = atoi(argv[6]), own_stream = atoi(argv[7]);
  FILE* fi = fopen(argv[3], "rb");
  if (!fi) return 2;
  int32_t hd[5];
  if (fread(hd, 4, 5, fi) != 5) return 2;
  const int steps = hd[0], F = hd[1], H = hd[2], W = hd[3], cap = hd[4];
  std::vector<std::vector<uint8_t>> L(steps), R(steps);
  for (int i = 0; i < steps; ++i) {
    L[i].resize((size_t)F * H * W); R[i].resize((size_t)F * H * W);
    if (fread(L[i].data(), 1, L[i].size(), fi) != L[i].size() || fread(R[i].data(), 1, R[i].size(), fi) != R[i].size()) return 2;
  }
  fclose(fi);

  d2fe_config c;
  d2fe_default_config(&c);
  c.max_width = W; c.max_height = H; c.max_batch = 2 * F; c.max_keypoints = cap; c.precision = D2FE_PREC_F32_WINO;
  d2fe_handle h = nullptr;
  CHECK(d2fe_create(&c, &h));
  {
    d2fe_weights::File f; d2fe_superpoint_weights w; std::string err;
    if (!f.load(argv[1]) || !d2fe_weights::superpoint(f, &w, &err)) { fprintf(stderr, "%s%s\n", f.error.c_str(), err.c_str()); return 3; }
    CHECK(d2fe_load_superpoint(h, &w));
  }
  {
    d2fe_weights::File f; std::vector<d2fe_nv_layer> layers; d2fe_netvlad_weights w; std::string err;
    if (!f.load(argv[2]) || !d2fe_weights::netvlad(f, &layers, &w, &err)) { fprintf(stderr, "%s%s\n", f.error.c_str(), err.c_str()); return 3; }
    CHECK(d2fe_load_netvlad(h, &w));
  }
  d2fe_pipe_config pc;
  d2fe_pipe_default_config(&pc);
  pc.lanes = lanes; pc.frames = F; pc.width = W; pc.height = H; pc.cap = cap; pc.netvlad = 1;
  d2fe_pipe pipe = nullptr;
  CHECK(d2fe_pipe_create(h, &pc, &pipe));

  // the communicator: rank 0 makes the id; in a swarm it travels to the other agents over whatever D2SLAM has (its LCM bus); here there is one rank
  char uid[128];
  void* comm = nullptr;
  CHECK(d2fe_rccl_unique_id(uid));
  CHECK(d2fe_rccl_comm_init_rank(uid, 1, 0, 0, &comm));
  d2fe_exchange_config xc;
  d2fe_exchange_default_config(&xc);
  const int NS = lanes + 2;
  xc.world = 1; xc.rank = 0; xc.wire = wire; xc.loopback = 1; xc.slots = NS; xc.own_stream = own_stream; xc.timing = 1; xc.gate_thres = 0.8; xc.ratio = 0.8;
  d2fe_exchange x = nullptr;
  CHECK(d2fe_exchange_create(pipe, comm, &xc, &x));
  const int NP = d2fe_exchange_pairs(x);
  if (NP != F) { fprintf(stderr, "pairs %d != %d\n", NP, F); return 6; }

  FILE* fo = fopen(argv[4], "wb");
  if (!fo) return 2;
  std::vector<int64_t> tk(steps);
  int enq = 0;
  float ag_ms = 0.f;
  auto finish = [&](int j) -> int {
    d2fe_pipe_result o; d2fe_exchange_result r;
    CHECK(d2fe_pipe_wait(pipe, tk[j], &o));
    CHECK(d2fe_exchange_collect(x, j % NS, &r));
    if (r.ticket != tk[j] || r.npairs != NP || r.cap != cap) { fprintf(stderr, "slot %d holds ticket %ld\n", j % NS, (long)r.ticket); return 7; }
    ag_ms += r.phase_ms[1];
    fwrite(o.n_kp, 4, 2 * F, fo);
    fwrite(&r.npairs, 4, 1, fo);
    fwrite(r.n_match, 4, NP, fo); fwrite(r.gate_pass, 4, NP, fo);
    fwrite(r.q_idx, 4, (size_t)NP * cap, fo); fwrite(r.t_idx, 4, (size_t)NP * cap, fo); fwrite(r.dist, 4, (size_t)NP * cap, fo);
    return 0;
  };
  for (int i = 0; i < steps; ++i) {
    CHECK(d2fe_pipe_submit(pipe, L[i].data(), R[i].data(), W, (size_t)W * H, &tk[i]));
    for (; enq <= i - 1; ++enq) CHECK(d2fe_exchange_enqueue(x, tk[enq], enq % NS));        // one submit behind the pipe
    if (i >= lanes) { const int rc = finish(i - lanes); if (rc) return rc; }
  }
  for (; enq < steps; ++enq) CHECK(d2fe_exchange_enqueue(x, tk[enq], enq % NS));
  for (int j = steps > lanes ? steps - lanes : 0; j < steps; ++j) { const int rc = finish(j); if (rc) return rc; }
  fclose(fo);
  d2fe_exchange_destroy(x);
  CHECK(d2fe_rccl_comm_destroy(comm));
  d2fe_pipe_destroy(pipe);
  d2fe_destroy(h);
  printf("exchange_test OK: %d submits of %d stereo frames, %d lanes, wire %d, %s; RCCL %s; all-gather %.3f ms per submit\n", steps, F, lanes, wire,
         own_stream ? "a stream of its own" : "the lanes' streams", d2fe_rccl_path(), ag_ms / steps);
  return 0;
}

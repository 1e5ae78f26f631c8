// The stream deal of the frames-in-flight pipe (d2slam_amd/csrc/stream_deal.h) on the host: the arrangements place_streams() must produce from the class sequences
// measured on MI355X (profiles/r05_pipe_one_frame.txt (8)), and the invariants of any deal.  Exit code 0 = all cases hold.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "stream_deal.h"

using d2fe::deal_streams;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "deal_test.cpp:%d: %s\n", __LINE__, #c); ++fails; } } while (0)

struct Deal { std::vector<int> f, s; int bad; };
static Deal run(const std::vector<int>& cl, int ncls, int nf, int ns) {
  std::vector<int> pf, ps;
  Deal d;
  d.bad = deal_streams(cl, ncls, nf, ns, pf, ps);
  std::set<int> seen;
  CHECK((int)pf.size() == nf && (int)ps.size() == ns);
  for (int c : pf) { CHECK(c >= 0 && c < (int)cl.size() && seen.insert(c).second); d.f.push_back(cl[c]); }       // every candidate at most once
  for (int c : ps) { CHECK(c >= 0 && c < (int)cl.size() && seen.insert(c).second); d.s.push_back(cl[c]); }
  return d;
}

int main() {
  {  // a fresh process, four lanes: classes come 0 1 2 3 0 1 2 3 -> own streams 0 1 2 3, second streams two classes on
    Deal d = run({0, 1, 2, 3, 0, 1, 2, 3}, 4, 4, 4);
    CHECK(d.bad == 0 && d.f == std::vector<int>({0, 1, 2, 3}) && d.s == std::vector<int>({2, 3, 0, 1}));
  }
  {  // two lanes: four streams in four classes
    Deal d = run({0, 1, 2, 3}, 4, 2, 2);
    CHECK(d.bad == 0 && d.f == std::vector<int>({0, 1}) && d.s == std::vector<int>({2, 3}));
  }
  {  // one lane: two adjacent streams
    Deal d = run({0, 1}, 2, 1, 1);
    CHECK(d.bad == 0 && d.f[0] != d.s[0]);
    Deal e = run({0, 0}, 1, 1, 1);           // both candidates on one pipe: a spare is wanted
    CHECK(e.bad > 0);
  }
  {  // bench.py after its other legs (recycled queues): 0 1 2 0 -- three classes among four candidates: short, spares are wanted ...
    Deal d = run({0, 1, 2, 0}, 3, 2, 2);
    CHECK(d.bad > 0);
    // ... and with the spares that came next there (1, then 3) the deal is the fresh one
    Deal e = run({0, 1, 2, 0, 1, 3}, 4, 2, 2);
    CHECK(e.bad == 0 && e.f == std::vector<int>({0, 1}) && e.s == std::vector<int>({2, 3}));
  }
  {  // the irregular sequence measured for a four-lane pipe there
    Deal d = run({0, 1, 0, 2, 2, 1, 0, 3, 2, 1, 0, 3}, 4, 4, 4);
    CHECK(d.bad == 0 && d.f == std::vector<int>({0, 1, 2, 3}) && d.s == std::vector<int>({2, 3, 0, 1}));
  }
  {  // three lanes on six fresh streams (0 1 2 3 0 1): class 2 is gone after the own streams -- no lane gets both streams in one class, but spares are wanted ...
    Deal d = run({0, 1, 2, 3, 0, 1}, 4, 3, 3);
    for (int k = 0; k < 3; ++k) CHECK(d.f[k] != d.s[k]);
    CHECK(d.bad > 0);
    // ... and the next two fresh streams (2 3) complete it: what tools/pipe_probe.py reports for three lanes
    Deal e = run({0, 1, 2, 3, 0, 1, 2, 3}, 4, 3, 3);
    CHECK(e.bad == 0 && e.f == std::vector<int>({0, 1, 2}) && e.s == std::vector<int>({2, 3, 0}));
  }
  {  // eight lanes, sixteen fresh streams: the four-lane pattern twice
    std::vector<int> cl;
    for (int c = 0; c < 16; ++c) cl.push_back(c % 4);
    Deal d = run(cl, 4, 8, 8);
    CHECK(d.bad == 0);
    for (int k = 0; k < 8; ++k) CHECK(d.f[k] == k % 4 && d.s[k] == (k + 2) % 4);
  }
  {  // no second streams (netvlad_inline = 1, netvlad_group, no NetVLAD): four lanes want four classes
    CHECK(run({0, 1, 2, 3}, 4, 4, 0).bad == 0);
    CHECK(run({0, 1, 0, 2}, 3, 4, 0).bad > 0);
    Deal d = run({0, 1, 0, 2, 1, 3}, 4, 4, 0);
    CHECK(d.bad == 0 && std::set<int>(d.f.begin(), d.f.end()).size() == 4);
  }
  {  // any class sequence: a valid deal (distinct candidates, checked in run()), bad >= 0, and bad == 0 means what it says
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (seed >> 16) & 0x7fff; };
    for (int t = 0; t < 2000; ++t) {
      const int K = 1 + (int)(rnd() % 8), ns = (rnd() & 1) ? K : 0, extra = (int)(rnd() % 5), ncls = 1 + (int)(rnd() % 5);
      std::vector<int> cl;
      for (int c = 0; c < K + ns + extra; ++c) cl.push_back((int)(rnd() % ncls));
      int present = 0;
      for (int k = 0; k < ncls; ++k) present += std::count(cl.begin(), cl.end(), k) > 0;
      if (present < ncls) continue;              // place_streams numbers the classes it has seen: every class has a member
      Deal d = run(cl, ncls, K, ns);
      CHECK(d.bad >= 0);
      if (d.bad == 0 && ns) {
        for (int k = 0; k < K; ++k) CHECK(d.f[k] != d.s[k] || ncls < 2);
        for (int k = 0; k + 1 < K; ++k) CHECK((int)std::set<int>({d.f[k], d.s[k], d.f[k + 1], d.s[k + 1]}).size() == 4);
      }
    }
  }
  if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  std::printf("deal_test OK\n");
  return 0;
}

// stream_deal.h -- dealing measured stream classes out to the lanes of a frames-in-flight pipe (pipe.hip, place_streams).  Plain C++ (no HIP): tests/cpp/deal_test.cpp
// exercises it on the host.
#pragma once
#include <algorithm>
#include <vector>

namespace d2fe {

// cl[c] = the hardware-pipe class (0 .. ncls-1) of candidate stream c; streams of one class take turns on the device.  Picks n_first own streams and n_second second
// (NetVLAD) streams, n_first + n_second <= cl.size(): lane k's own stream from class k mod n, its second stream half the classes further on (two of four), so that the four
// streams of two consecutive lanes sit in four classes -- as far as the streams at hand allow: a class that has run out is replaced by the one that meets the fewest of the
// neighbouring lanes' streams.  Returns how far the deal is from that: streams of one class within a lane's pair or within two consecutive lanes' streams, counted against
// what four hardware pipes would allow (0 = nothing more to gain from further candidates)
inline int deal_streams(const std::vector<int>& cl, int ncls, int n_first, int n_second, std::vector<int>& pick_first, std::vector<int>& pick_second) {
  const int NC = (int)cl.size();
  std::vector<char> used((size_t)NC, 0);
  auto take = [&](int want, const std::vector<int>& avoid) -> int {
    std::vector<int> left((size_t)ncls, 0);
    for (int c = 0; c < NC; ++c) if (!used[c]) ++left[cl[c]];
    int cls_pick = -1;
    if (left[want] > 0) cls_pick = want;
    else {
      long best_score = -1;
      for (int k = 0; k < ncls; ++k) {
        if (!left[k]) continue;
        int meets = 0;
        for (int a : avoid) meets += a == k;
        const long score = (long)(16 - meets) * 1024 + left[k];        // fewest neighbours first, then the class with the most streams left
        if (score > best_score) { best_score = score; cls_pick = k; }
      }
    }
    for (int c = 0; c < NC; ++c) if (!used[c] && cl[c] == cls_pick) { used[c] = 1; return c; }
    return -1;      // not reached: the pipe never asks for more streams than there are candidates
  };
  pick_first.clear(); pick_second.clear();
  std::vector<int> fc, sc;
  for (int k = 0; k < n_first; ++k) {
    std::vector<int> avoid;
    if (k > 0) avoid.push_back(fc[k - 1]);
    if (k + 1 == n_first && n_first > 2) avoid.push_back(fc[0]);
    pick_first.push_back(take(k % ncls, avoid)); fc.push_back(cl[pick_first.back()]);
  }
  for (int k = 0; k < n_second; ++k) {
    std::vector<int> avoid = {fc[k], fc[(k + 1) % n_first], fc[(k + n_first - 1) % n_first]};
    if (k > 0) avoid.push_back(sc[k - 1]);
    if (k + 1 == n_second && n_second > 2) avoid.push_back(sc[0]);
    pick_second.push_back(take((k + (ncls + 1) / 2) % ncls, avoid)); sc.push_back(cl[pick_second.back()]);
  }
  auto clashes = [&](std::vector<int> v) {       // streams beyond the first of every class, less what fewer classes than streams force
    const int n = (int)v.size();
    std::sort(v.begin(), v.end());
    const int distinct = (int)(std::unique(v.begin(), v.end()) - v.begin());
    return std::max(0, std::min(n, std::max(ncls, 4)) - distinct);      // (four hardware pipes are there even when fewer classes have shown up so far)
  };
  int bad = 0;
  if (n_second) {
    if (n_first == 1) bad += clashes({fc[0], sc[0]});
    // pairs of consecutive lanes; the pair (last, first) only where the pattern closes (a multiple of four lanes)
    for (int k = 0; k + 1 < n_first || (k + 1 == n_first && n_first > 2 && n_first % 4 == 0); ++k) { const int m = (k + 1) % n_first; bad += clashes({fc[k], sc[k], fc[m], sc[m]}); }
  } else {
    for (int k = 0; k < n_first; k += 4) bad += clashes(std::vector<int>(fc.begin() + k, fc.begin() + std::min(k + 4, n_first)));      // blocks of four lanes
  }
  return bad;
}

}  // namespace d2fe

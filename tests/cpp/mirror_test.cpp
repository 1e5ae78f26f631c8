// Drives the C++ mirror of the reference interfaces (include/d2fe.hpp) exactly the way D2SLAM's call sites do and dumps the
// results for tests/test_cpp_mirror.py, which compares them with the oracle.  Usage: mirror_test <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "d2fe.hpp"

using namespace D2FrontEnd;

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <typename T>
static void wr(FILE* f, const std::vector<T>& v) {
  const int32_t n = (int32_t)v.size();
  fwrite(&n, 4, 1, f);
  if (n) fwrite(v.data(), sizeof(T), v.size(), f);
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* fi = fopen(argv[1], "rb");
  if (!fi) return 2;
  int32_t H, W, maxkp;
  if (!rd(fi, &H, 1) || !rd(fi, &W, 1) || !rd(fi, &maxkp, 1)) return 2;
  std::vector<std::vector<float>> ws(12), bs(12);
  d2fe_superpoint_weights w;
  for (int l = 0; l < 12; ++l) {
    int32_t dims[3];
    if (!rd(fi, dims, 3)) return 2;
    ws[l].resize((size_t)dims[0] * dims[1] * dims[2] * dims[2]); bs[l].resize(dims[0]);
    if (!rd(fi, ws[l].data(), ws[l].size()) || !rd(fi, bs[l].data(), bs[l].size())) return 2;
    w.layer[l].weight = ws[l].data(); w.layer[l].bias = bs[l].data();
    w.layer[l].cout = dims[0]; w.layer[l].cin = dims[1]; w.layer[l].ksize = dims[2];
  }
  std::vector<uint8_t> img0((size_t)H * W), img1((size_t)H * W);
  if (!rd(fi, img0.data(), img0.size()) || !rd(fi, img1.data(), img1.size())) return 2;
  if (maxkp == -1) {
    // keep-all (SuperPointConfig::max_keypoints = -1, superpoint_tensorrt.cpp:241-253): infer() must neither throw nor truncate.  The file carries the
    // threshold and a deliberately small starting capacity, so that the grow-and-run-again path of the mirror is what delivers the result
    float thr; int32_t cap0;
    if (!rd(fi, &thr, 1) || !rd(fi, &cap0, 1)) return 2;
    fclose(fi);
    SuperPointConfig kc;
    kc.max_keypoints = -1; kc.keep_all_capacity = cap0; kc.keypoint_threshold = thr; kc.input_width = W; kc.input_height = H; kc.max_batch = 1;
    SuperPoint ks(kc);
    if (!ks.build(w)) return 3;
    std::vector<Point2f> k(1, Point2f(-7.f, -7.f));        // infer() APPENDS keypoints (:172-174)
    std::vector<float> d, sc;
    bool ok = false;
    try { ok = ks.infer(ImageView(img0.data(), H, W), k, d, sc); } catch (...) { return 10; }
    if (!ok || k.empty() || k[0].x != -7.f) return 4;
    k.erase(k.begin());
    FILE* fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    std::vector<float> kf; for (auto& q : k) { kf.push_back(q.x); kf.push_back(q.y); }
    wr(fo, kf); wr(fo, sc); wr(fo, d);
    fclose(fo);
    return 0;
  }
  fclose(fi);

  SuperPointConfig cfg;
  cfg.max_keypoints = maxkp; cfg.input_width = W; cfg.input_height = H;
  SuperPoint sp(cfg);
  if (!sp.build(w)) return 3;
  std::vector<Point2f> k0, k1;
  std::vector<float> d0, d1, s0, s1;
  if (!sp.infer(ImageView(img0.data(), H, W), k0, d0, s0)) return 4;
  if (!sp.infer(ImageView(img1.data(), H, W), k1, d1, s1)) return 4;
  // failure behaviour: a wrong size empties the outputs and returns false (superpoint_tensorrt.cpp:164-170)
  std::vector<Point2f> kbad(3); std::vector<float> dbad(5), sbad(2);
  const bool bad = sp.infer(ImageView(img0.data(), H + 8, W), kbad, dbad, sbad);
  if (bad || !kbad.empty() || !dbad.empty() || !sbad.empty()) return 5;

  const DescView A(d0.data(), (int)k0.size(), 256), B(d1.data(), (int)k1.size(), 256);
  const std::vector<DMatch> m = matchKNN(sp.handle(), A, B, 0.8, k0, k1, 0.2 * W);
  const std::vector<DMatch> mc = matchCrossCheck(sp.handle(), A, B);
  std::vector<Point2f> half_pts; std::vector<int> half_idx;
  const std::vector<float> half_desc = getFeatureHalfImg(k0, A, true, W, 200.0, half_pts, half_idx);

  // LK tracker: detect on frame 0 (both detectors), track into frame 1
  LKImageInfo prev;
  prev.pyr = buildImagePyramid(sp.handle(), ImageView(img0.data(), H, W));
  std::vector<Point2f> fast_pts, gftt_pts;
  detectPoints(sp.handle(), prev.pyr, fast_pts, std::vector<Point2f>(), 150, true, 3, 4, 20.0);
  detectPoints(sp.handle(), prev.pyr, gftt_pts, std::vector<Point2f>(), 150, false, 3, 4, 20.0);
  prev.lk_pts = fast_pts;
  for (size_t i = 0; i < fast_pts.size(); ++i) { prev.lk_ids.push_back(1000 + (int64_t)i); prev.lk_local_index.push_back((int)i); prev.lk_types.push_back(0); }
  LKImageInfo cur = opticalflowTrackPyr(sp.handle(), ImageView(img1.data(), H, W), prev, WHOLE_IMG_MATCH, 200.0);

  // A8: lift the keypoints of frame 0 through the three camera models (values checked against numpy by the Python side)
  d2fe_mei_camera mei{2.2176903753419963, -0.17703529535292872, 0.7517933338735744, -0.0008911425891703079, 2.1653595535258756e-05,
                      1162.5434300524314, 1161.839362615319, 660.6393183718625, 386.1663300322095};
  d2fe_mei_camera mei_c = mei;   // same lens, principal point inside this test's small image; `mei` itself puts most of the image past the
  mei_c.u0 = W / 2 + 0.3; mei_c.v0 = H / 2 + 0.2; mei_c.gamma1 = mei_c.gamma2 = 0.9 * W;   // model's valid cone -> NaN -> skipped (:625-631)
  std::vector<double> lifts;
  auto push = [&](const std::vector<Landmark>& v) { for (auto& l : v) { lifts.push_back(l.pt3d_norm.x); lifts.push_back(l.pt3d_norm.y); lifts.push_back(l.pt3d_norm.z); } };
  push(fillLandmarks(k0, [&](const Point2f& p) { return liftProjectivePinhole(385.0, 386.0, 322.5, 241.0, RadTan{0.01, -0.02, 0.001, -0.0005}, p); }));
  push(fillLandmarks(k0, [&](const Point2f& p) { return liftProjectiveMEI(mei_c, p); }));
  push(fillLandmarks(k0, [&](const Point2f& p) { return liftProjectiveMEI(mei, p); }));
  push(fillLandmarks(k0, [&](const Point2f& p) { return liftProjectiveCylindrical(W / 3.4906585039886591, W / 3.4906585039886591, W / 2, H / 2, p); }));

  // the frames-in-flight pipe behind the reference's containers (StereoPipe): three frames in flight over the two images and their mirror order, two threads'
  // worth of calls made from one; every result must equal infer() + matchKNN() on the same frames, bit for bit (the same kernels)
  {
    d2fe_pipe_config pc;
    d2fe_pipe_default_config(&pc);
    pc.lanes = 3; pc.width = W; pc.height = H; pc.cap = maxkp; pc.netvlad = 0; pc.match_lr = 1; pc.match_prev = 1; pc.ratio = 0.8; pc.coalesce = 2; pc.coalesce_depth = 1;
    StereoPipe pipe(sp.handle(), pc);
    if (!pipe.ok()) return 6;
    {       // every lane's two streams sit in different hardware-pipe classes when the placement could be measured (pc.netvlad = 0 here: no second streams, class -1)
      std::vector<std::pair<int, int>> pl;
      const int ncl = pipe.streamPlacement(pl);
      if ((int)pl.size() != pc.lanes || ncl < 0 || ncl > 8) return 6;
      for (auto& q : pl) if (q.second != -1 || (ncl >= 2 && (q.first < 0 || q.first >= ncl))) return 6;
    }
    const uint8_t* L[4] = {img0.data(), img1.data(), img0.data(), img1.data()};
    const uint8_t* R[4] = {img1.data(), img0.data(), img1.data(), img0.data()};
    int64_t t[4];
    for (int i = 0; i < 4; ++i) { t[i] = pipe.submit(ImageView(L[i], H, W), ImageView(R[i], H, W)); if (t[i] < 0) return 6; }
    const std::vector<Point2f>* K[2] = {&k0, &k1};
    const std::vector<float>* D[2] = {&d0, &d1};
    for (int i = 0; i < 4; ++i) {
      StereoFrameResult fr;
      if (!pipe.wait(t[i], fr)) return 6;
      const int li = i & 1, ri = li ^ 1;
      if (fr.kps_left.size() != K[li]->size() || fr.kps_right.size() != K[ri]->size()) return 7;
      if (std::memcmp(fr.kps_left.data(), K[li]->data(), K[li]->size() * sizeof(Point2f)) || std::memcmp(fr.desc_left.data(), D[li]->data(), D[li]->size() * 4)) return 7;
      if (std::memcmp(fr.desc_right.data(), D[ri]->data(), D[ri]->size() * 4)) return 7;
      const DescView dl(D[li]->data(), (int)K[li]->size(), 256), dr(D[ri]->data(), (int)K[ri]->size(), 256);
      const std::vector<DMatch> mlr = matchKNN(sp.handle(), dl, dr, 0.8);
      if (mlr.size() != fr.left_right.size()) return 8;
      for (size_t j = 0; j < mlr.size(); ++j)
        if (mlr[j].queryIdx != fr.left_right[j].queryIdx || mlr[j].trainIdx != fr.left_right[j].trainIdx || mlr[j].distance != fr.left_right[j].distance) return 8;
      if (i == 0) { if (!fr.left_prev.empty()) return 9; }
      else {
        // the previous left frame is the other image, i.e. this frame's right image: same match list
        if (fr.left_prev.size() != mlr.size()) return 9;
        for (size_t j = 0; j < mlr.size(); ++j) if (mlr[j].trainIdx != fr.left_prev[j].trainIdx || mlr[j].distance != fr.left_prev[j].distance) return 9;
      }
    }
  }

  FILE* fo = fopen(argv[2], "wb");
  if (!fo) return 2;
  auto flat = [](const std::vector<Point2f>& p) { std::vector<float> o; for (auto& q : p) { o.push_back(q.x); o.push_back(q.y); } return o; };
  auto mflat = [](const std::vector<DMatch>& mm) { std::vector<float> o; for (auto& q : mm) { o.push_back((float)q.queryIdx); o.push_back((float)q.trainIdx); o.push_back(q.distance); } return o; };
  wr(fo, flat(k0)); wr(fo, s0); wr(fo, d0); wr(fo, flat(k1)); wr(fo, s1); wr(fo, d1);
  wr(fo, mflat(m)); wr(fo, mflat(mc));
  wr(fo, half_idx); wr(fo, half_desc);
  wr(fo, flat(fast_pts)); wr(fo, flat(gftt_pts)); wr(fo, flat(cur.lk_pts)); wr(fo, cur.lk_ids);
  wr(fo, lifts);
  fclose(fo);
  d2fe_lk_frame_destroy(prev.pyr);
  d2fe_lk_frame_destroy(cur.pyr);
  return 0;
}

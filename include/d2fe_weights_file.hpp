// d2fe_weights_file.hpp -- reader of the "D2FW" weight container (d2slam_amd/weights.py: save_superpoint_d2fw / save_netvlad_d2fw), header-only.
// D2SLAM names model FILES in its configuration (superpoint_model / netvlad_model, d2frontend/src/d2frontend_params.cpp:86-106) where the C ABI of
// include/d2fe.h takes plain arrays (d2fe_superpoint_weights, d2fe_netvlad_weights); this is what include/d2fe_adapter.cpp puts in between.
// Layout (little endian): "D2FW" | u32 version = 1 | u32 n | n x { u32 name_len | name | u32 ndim | ndim x i64 dims | f32 data in C order }.
#ifndef D2FE_WEIGHTS_FILE_HPP_
#define D2FE_WEIGHTS_FILE_HPP_
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "d2fe.h"

namespace d2fe_weights {
struct Tensor { std::vector<int64_t> dims; std::vector<float> data; };
struct File {
  std::map<std::string, Tensor> t;
  std::string error;
  bool load(const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { error = "cannot open " + path; return false; }
    auto rd = [&](void* p, size_t n) { return std::fread(p, 1, n, f) == n; };
    char magic[4]; uint32_t ver = 0, n = 0;
    bool ok = rd(magic, 4) && rd(&ver, 4) && rd(&n, 4) && !std::memcmp(magic, "D2FW", 4) && ver == 1 && n < 100000;
    for (uint32_t i = 0; ok && i < n; ++i) {
      uint32_t nl = 0, nd = 0;
      ok = rd(&nl, 4) && nl < 4096;
      std::string name(nl, '\0');
      ok = ok && rd(&name[0], nl) && rd(&nd, 4) && nd <= 8;
      Tensor x; x.dims.resize(nd);
      int64_t cnt = 1;
      for (uint32_t d = 0; ok && d < nd; ++d) { ok = rd(&x.dims[d], 8) && x.dims[d] >= 0 && x.dims[d] < (1ll << 32); cnt *= x.dims[d]; }
      ok = ok && cnt < (1ll << 31);
      if (ok) { x.data.resize((size_t)cnt); ok = rd(x.data.data(), sizeof(float) * (size_t)cnt); }
      if (ok) t[name] = std::move(x);
    }
    std::fclose(f);
    if (!ok) error = path + " is not a D2FW version 1 container";
    return ok;
  }
  const Tensor* get(const std::string& name) const { auto it = t.find(name); return it == t.end() ? nullptr : &it->second; }
};

// The 12 SuperPoint layers (state_dict names of d2frontend/superpoint.ipynb:306-321) -> d2fe_superpoint_weights; pointers stay valid as long as `f` lives
inline bool superpoint(const File& f, d2fe_superpoint_weights* w, std::string* err) {
  static const char* names[D2FE_SP_NUM_LAYERS] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"};
  for (int i = 0; i < D2FE_SP_NUM_LAYERS; ++i) {
    const Tensor* W = f.get(std::string(names[i]) + ".weight");
    const Tensor* b = f.get(std::string(names[i]) + ".bias");
    if (!W || !b || W->dims.size() != 4 || W->dims[2] != W->dims[3] || b->dims.size() != 1 || b->dims[0] != W->dims[0]) { if (err) *err = std::string("layer ") + names[i] + " missing or of the wrong shape"; return false; }
    w->layer[i] = d2fe_conv_params{W->data.data(), b->data.data(), (int32_t)W->dims[0], (int32_t)W->dims[1], (int32_t)W->dims[2]};
  }
  return true;
}

// The MobileNetVLAD layer list + head -> d2fe_netvlad_weights (`layers` backs w->layers)
inline bool netvlad(const File& f, std::vector<d2fe_nv_layer>* layers, d2fe_netvlad_weights* w, std::string* err) {
  const Tensor* arch = f.get("arch");
  if (!arch || arch->dims.size() != 2 || arch->dims[1] != 6) { if (err) *err = "no `arch` table"; return false; }
  const int n = (int)arch->dims[0];
  layers->resize(n);
  for (int i = 0; i < n; ++i) {
    const float* a = arch->data.data() + 6 * i;
    const Tensor* W = f.get("layer." + std::to_string(i) + ".weight");
    const Tensor* b = f.get("layer." + std::to_string(i) + ".bias");
    if (!W || !b) { if (err) *err = "layer " + std::to_string(i) + " missing"; return false; }
    (*layers)[i] = d2fe_nv_layer{(int32_t)a[0], (int32_t)a[1], (int32_t)a[2], (int32_t)a[3], (int32_t)a[4], (int32_t)a[5], W->data.data(), b->data.data()};
  }
  const Tensor *pw = f.get("head.pre_w"), *pb = f.get("head.pre_b"), *aw = f.get("head.assign_w"), *ab = f.get("head.assign_b"), *ce = f.get("head.centroids");
  if (!pw || !pb || !aw || !ab || !ce || pw->dims.size() != 2 || aw->dims.size() != 2) { if (err) *err = "head arrays missing"; return false; }
  w->n_layers = n; w->layers = layers->data();
  w->feat_dim = (int32_t)pw->dims[1]; w->proj_dim = (int32_t)pw->dims[0]; w->n_clusters = (int32_t)aw->dims[0];
  w->pre_w = pw->data.data(); w->pre_b = pb->data.data(); w->assign_w = aw->data.data(); w->assign_b = ab->data.data(); w->centroids = ce->data.data();
  return true;
}
}  // namespace d2fe_weights
#endif

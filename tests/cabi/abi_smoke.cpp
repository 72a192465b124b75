// The same consumer in C++17, the reference's own language (CMakeLists.txt:17): the handle in a
// unique_ptr, arrays in vectors, OrbFrameBounds brace-initialised and passed by value - the way the
// shim classes of a host build would hold them.  Reads the directory tests/test_cabi_gpu.py wrote.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbhip.h"

namespace {

std::string g_dir;

template <typename T>
std::vector<T> load(const std::string& name) {
  std::ifstream f(g_dir + "/" + name, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + name);
  const std::streamsize bytes = f.tellg();
  if (bytes % static_cast<std::streamsize>(sizeof(T))) throw std::runtime_error(name + ": odd size");
  std::vector<T> v(static_cast<size_t>(bytes) / sizeof(T));
  f.seekg(0);
  f.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}

void ok(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + std::to_string(rc) + " (" + orb_last_error() + ")");
}

struct ExtractorDeleter { void operator()(OrbExtractor* h) const { orb_extractor_destroy(h); } };
using ExtractorPtr = std::unique_ptr<OrbExtractor, ExtractorDeleter>;

struct Desc { uint8_t b[32]; };

void test_extract() {
  const auto par = load<int>("extract_params.raw");
  const auto img = load<uint8_t>("image.raw");
  const auto want_kp = load<OrbKeyPoint>("expect_kps.raw");
  const auto want_desc = load<Desc>("expect_desc.raw");
  OrbExtractor* raw = nullptr;
  ok(orb_extractor_create(par.at(2), 1.2f, par.at(3), par.at(4), par.at(5), 0, &raw), "orb_extractor_create");
  ExtractorPtr ex(raw);
  std::vector<float> sf(par[3]), inv(par[3]), s2(par[3]), is2(par[3]);
  if (orb_extractor_get_scale_tables(ex.get(), sf.data(), inv.data(), s2.data(), is2.data()) != par[3])
    throw std::runtime_error("scale tables");
  for (int l = 1; l < par[3]; l++)
    if (!(sf[l] > sf[l - 1]) || std::fabs(sf[l] * inv[l] - 1.f) > 1e-6f || s2[l] != sf[l] * sf[l]) throw std::runtime_error("scale table values");
  const int cap = orb_extractor_max_keypoints(ex.get(), par[0], par[1]);
  ok(cap, "orb_extractor_max_keypoints");
  std::vector<OrbKeyPoint> kps(cap);
  std::vector<Desc> desc(cap);
  // two frames in one call, the second the same image: both must equal the expected arrays
  std::vector<uint8_t> two(img);
  two.insert(two.end(), img.begin(), img.end());
  std::vector<OrbKeyPoint> kps2(2 * static_cast<size_t>(cap));
  std::vector<Desc> desc2(2 * static_cast<size_t>(cap));
  int n = -1, n2[2] = {-1, -1};
  ok(orb_extract(ex.get(), img.data(), par[0], par[1], static_cast<size_t>(par[1]), cap, kps.data(), desc[0].b, &n), "orb_extract");
  ok(orb_extract_batch(ex.get(), 2, two.data(), par[0], par[1], static_cast<size_t>(par[1]), img.size(), cap, kps2.data(),
                       desc2[0].b, n2), "orb_extract_batch");
  if (static_cast<size_t>(n) != want_kp.size() || n2[0] != n || n2[1] != n) throw std::runtime_error("keypoint count " + std::to_string(n));
  size_t bad = 0;
  for (size_t i = 0; i < want_kp.size(); i++) {
    const OrbKeyPoint &a = kps[i], &b = want_kp[i];
    float da = std::fabs(a.angle - b.angle);
    if (da > 180.f) da = 360.f - da;
    if (a.x != b.x || a.y != b.y || a.size != b.size || a.response != b.response || a.octave != b.octave || da > 1e-4f * 57.29578f)
      throw std::runtime_error("keypoint " + std::to_string(i) + " differs");
    bad += std::memcmp(desc[i].b, want_desc[i].b, 32) != 0;
    for (int f = 0; f < 2; f++)
      if (std::memcmp(&kps2[f * static_cast<size_t>(cap) + i], &a, sizeof a) != 0 ||
          std::memcmp(desc2[f * static_cast<size_t>(cap) + i].b, desc[i].b, 32) != 0)
        throw std::runtime_error("batch frame " + std::to_string(f) + " differs from the single call at " + std::to_string(i));
  }
  if (bad * 1000 > want_kp.size()) throw std::runtime_error(std::to_string(bad) + " descriptors differ");
  std::cout << "extract: " << want_kp.size() << " keypoints equal (single call and batch of 2), " << bad << " descriptors differ\n";
}

void test_match() {
  const auto ip = load<int>("match_ints.raw");
  const auto fp = load<float>("match_floats.raw");
  const auto tcw = load<float>("match_tcw.raw");
  const auto sf = load<float>("match_scale.raw");
  const auto l_obs = load<int>("match_last_obs.raw");
  const auto l_xw = load<float>("match_last_xw.raw");
  const auto l_desc = load<Desc>("match_last_desc.raw");
  const auto l_oct = load<int>("match_last_octave.raw");
  const auto l_ang = load<float>("match_last_angle.raw");
  const auto c_x = load<float>("match_cur_x.raw");
  const auto c_y = load<float>("match_cur_y.raw");
  const auto c_oct = load<int>("match_cur_octave.raw");
  const auto c_ang = load<float>("match_cur_angle.raw");
  const auto c_desc = load<Desc>("match_cur_desc.raw");
  const auto c_obs = load<int>("match_cur_obs.raw");
  const auto want = load<int>("expect_match.raw");
  const auto want_n = load<int>("expect_nmatch.raw");
  const int n_last = ip.at(0), n_cur = ip.at(1);
  const OrbFrameBounds bounds{fp.at(4), fp.at(5), fp.at(6), fp.at(7)};
  std::vector<int> got(n_cur, 12345);
  int nm = -1;
  ok(orb_search_by_projection_frames(0, 1, n_last, n_cur, tcw.data(), fp[0], fp[1], fp[2], fp[3], bounds, sf.data(), ip.at(2),
                                     &n_last, l_obs.data(), l_xw.data(), l_desc[0].b, l_oct.data(), l_ang.data(), &n_cur,
                                     c_x.data(), c_y.data(), c_oct.data(), c_ang.data(), c_desc[0].b, c_obs.data(), fp.at(8),
                                     ip.at(3), got.data(), &nm),
     "orb_search_by_projection_frames");
  if (nm != want_n.at(0) || got != want) throw std::runtime_error("SearchByProjection differs: " + std::to_string(nm));
  std::cout << "SearchByProjection: " << nm << " matches, " << got.size() << " entries equal\n";
}

void test_knn() {
  const auto q = load<Desc>("knn_q.raw");
  const auto db = load<Desc>("knn_db.raw");
  const float ratio = load<float>("knn_ratio.raw").at(0);
  const auto wb = load<int>("expect_knn_best.raw"), ws = load<int>("expect_knn_second.raw"), wi = load<int>("expect_knn_idx.raw");
  const auto wa = load<uint8_t>("expect_knn_accept.raw");
  std::vector<int> best(q.size()), second(q.size()), idx(q.size());
  std::vector<uint8_t> acc(q.size());
  ok(orb_knn2_bruteforce(0, q[0].b, static_cast<int>(q.size()), db[0].b, static_cast<int>(db.size()), ratio, best.data(),
                         second.data(), idx.data(), acc.data()),
     "orb_knn2_bruteforce");
  if (best != wb || second != ws || idx != wi || acc != wa) throw std::runtime_error("kNN-2 differs");
  std::cout << "kNN-2: " << q.size() << " queries x " << db.size() << " rows equal\n";
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 2) { std::cerr << "usage: " << argv[0] << " <directory with the raw arrays>\n"; return 2; }
  g_dir = argv[1];
  static_assert(sizeof(OrbKeyPoint) == 24 && sizeof(OrbFrameBounds) == 16, "record layouts of the C ABI");
  try {
    if (orb_device_count() < 1) { std::cerr << "no GPU: " << orb_last_error() << "\n"; return 3; }
    test_extract();
    test_match();
    test_knn();
  } catch (const std::exception& e) {
    std::cerr << "abi_smoke: " << e.what() << "\n";
    return 1;
  }
  std::cout << "abi_smoke ok\n";
  return 0;
}

/*
 * A plain C99 consumer of include/orbhip.h: what a C or C++ host that links liborbhip.so sees,
 * with no Python, ctypes or torch in the process.  Reads raw arrays from a directory (written by
 * tests/test_cabi_gpu.py from the committed golden vectors and the CPU oracle), calls
 *   orb_extractor_create -> orb_extractor_max_keypoints -> orb_extract
 *   orb_search_by_projection_frames   (OrbFrameBounds passed by value)
 *   orb_knn2_bruteforce
 * and compares every output with the expected arrays.  Exit code 0 and "abi_smoke ok" on success.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orbhip.h"

static char g_dir[1024];

static void* read_file(const char* name, size_t elem, size_t* count) {
  char path[1200];
  FILE* f;
  long bytes;
  void* buf;
  snprintf(path, sizeof path, "%s/%s", g_dir, name);
  f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf = malloc(bytes > 0 ? (size_t)bytes : 1);
  if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  fclose(f);
  if ((size_t)bytes % elem != 0) { fprintf(stderr, "%s: size is not a multiple of %zu\n", path, elem); exit(2); }
  if (count) *count = (size_t)bytes / elem;
  return buf;
}

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ < 0) {                                                                   \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, orb_last_error());         \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static int test_extract(void) {
  size_t n_par, n_img, n_kp, n_desc;
  int* par = (int*)read_file("extract_params.raw", sizeof(int), &n_par); /* rows cols nfeatures nlevels iniTh minTh */
  unsigned char* img = (unsigned char*)read_file("image.raw", 1, &n_img);
  OrbKeyPoint* want_kp = (OrbKeyPoint*)read_file("expect_kps.raw", sizeof(OrbKeyPoint), &n_kp);
  unsigned char* want_desc = (unsigned char*)read_file("expect_desc.raw", 32, &n_desc);
  OrbExtractor* ex = NULL;
  OrbKeyPoint* kps;
  unsigned char* desc;
  float scale[16];
  int cap, n = -1, levels, bad_desc = 0, w = 0, h = 0;
  size_t i;
  if (n_par != 6 || n_img != (size_t)par[0] * (size_t)par[1] || n_kp != n_desc) { fprintf(stderr, "extract inputs malformed\n"); return 1; }
  CHECK(orb_extractor_create(par[2], 1.2f, par[3], par[4], par[5], 0, &ex));
  levels = orb_extractor_get_scale_tables(ex, scale, NULL, NULL, NULL);
  if (levels != par[3] || scale[0] != 1.0f) { fprintf(stderr, "scale tables: %d levels\n", levels); return 1; }
  cap = orb_extractor_max_keypoints(ex, par[0], par[1]);
  CHECK(cap);
  kps = (OrbKeyPoint*)calloc((size_t)cap, sizeof(OrbKeyPoint));
  desc = (unsigned char*)calloc((size_t)cap, 32);
  CHECK(orb_extract(ex, img, par[0], par[1], (size_t)par[1], cap, kps, desc, &n));
  if ((size_t)n != n_kp) { fprintf(stderr, "extract: %d keypoints, expected %zu\n", n, n_kp); return 1; }
  for (i = 0; i < n_kp; i++) {
    const OrbKeyPoint *a = &kps[i], *b = &want_kp[i];
    float da = fabsf(a->angle - b->angle);
    if (da > 180.f) da = 360.f - da;
    if (a->x != b->x || a->y != b->y || a->size != b->size || a->response != b->response || a->octave != b->octave) {
      fprintf(stderr, "keypoint %zu differs: (%g, %g, %d) vs (%g, %g, %d)\n", i, a->x, a->y, a->octave, b->x, b->y, b->octave);
      return 1;
    }
    if (da > 1e-4f * 57.29578f) { fprintf(stderr, "keypoint %zu: angle %g vs %g\n", i, a->angle, b->angle); return 1; }
    if (memcmp(desc + 32 * i, want_desc + 32 * i, 32) != 0) bad_desc++;
  }
  if ((double)bad_desc > 0.001 * (double)n_kp) { fprintf(stderr, "%d of %zu descriptors differ\n", bad_desc, n_kp); return 1; }
  /* mvImagePyramid of the call stays readable (include/ORBextractor.h:166) */
  CHECK(orb_get_pyramid_level(ex, 0, 1, 0, NULL, 0, &w, &h));
  if (w <= 0 || h <= 0 || w >= par[1]) { fprintf(stderr, "level 1 is %d x %d\n", w, h); return 1; }
  /* an empty image: OK and zero keypoints (src/ORBextractor.cc:1227) */
  n = -1;
  CHECK(orb_extract(ex, img, 0, 0, 0, cap, kps, desc, &n));
  if (n != 0) { fprintf(stderr, "empty image gave %d keypoints\n", n); return 1; }
  /* a null output is an argument error, not a crash */
  if (orb_extract(ex, img, par[0], par[1], (size_t)par[1], cap, NULL, desc, &n) != ORBHIP_ERR_ARG || orb_last_error()[0] == 0) {
    fprintf(stderr, "null keypoint array was not refused\n");
    return 1;
  }
  orb_extractor_destroy(ex);
  printf("extract: %zu keypoints equal, %d descriptors differ\n", n_kp, bad_desc);
  free(par); free(img); free(want_kp); free(want_desc); free(kps); free(desc);
  return 0;
}

static int test_match(void) {
  size_t n_i, n_f, nl, nc, k;
  int* ip = (int*)read_file("match_ints.raw", sizeof(int), &n_i);       /* n_last n_cur nlevels check_ori */
  float* fp = (float*)read_file("match_floats.raw", sizeof(float), &n_f); /* fx fy cx cy minx maxx miny maxy th */
  float* tcw = (float*)read_file("match_tcw.raw", sizeof(float), NULL);
  float* sf = (float*)read_file("match_scale.raw", sizeof(float), NULL);
  int* l_obs = (int*)read_file("match_last_obs.raw", sizeof(int), &nl);
  float* l_xw = (float*)read_file("match_last_xw.raw", sizeof(float), NULL);
  unsigned char* l_desc = (unsigned char*)read_file("match_last_desc.raw", 32, NULL);
  int* l_oct = (int*)read_file("match_last_octave.raw", sizeof(int), NULL);
  float* l_ang = (float*)read_file("match_last_angle.raw", sizeof(float), NULL);
  float* c_x = (float*)read_file("match_cur_x.raw", sizeof(float), &nc);
  float* c_y = (float*)read_file("match_cur_y.raw", sizeof(float), NULL);
  int* c_oct = (int*)read_file("match_cur_octave.raw", sizeof(int), NULL);
  float* c_ang = (float*)read_file("match_cur_angle.raw", sizeof(float), NULL);
  unsigned char* c_desc = (unsigned char*)read_file("match_cur_desc.raw", 32, NULL);
  int* c_obs = (int*)read_file("match_cur_obs.raw", sizeof(int), NULL);
  int* want = (int*)read_file("expect_match.raw", sizeof(int), NULL);
  int* want_n = (int*)read_file("expect_nmatch.raw", sizeof(int), NULL);
  OrbFrameBounds b;
  int n_last, n_cur, nm = -1, *got;
  if (n_i != 4 || n_f != 9 || nl != (size_t)ip[0] || nc != (size_t)ip[1]) { fprintf(stderr, "match inputs malformed\n"); return 1; }
  n_last = ip[0]; n_cur = ip[1];
  b.min_x = fp[4]; b.max_x = fp[5]; b.min_y = fp[6]; b.max_y = fp[7];
  got = (int*)malloc(sizeof(int) * (size_t)n_cur);
  CHECK(orb_search_by_projection_frames(0, 1, n_last, n_cur, tcw, fp[0], fp[1], fp[2], fp[3], b, sf, ip[2], &n_last, l_obs, l_xw,
                                        l_desc, l_oct, l_ang, &n_cur, c_x, c_y, c_oct, c_ang, c_desc, c_obs, fp[8], ip[3], got,
                                        &nm));
  if (nm != want_n[0]) { fprintf(stderr, "SearchByProjection returned %d, expected %d\n", nm, want_n[0]); return 1; }
  for (k = 0; k < nc; k++)
    if (got[k] != want[k]) { fprintf(stderr, "cur_match[%zu] = %d, expected %d\n", k, got[k], want[k]); return 1; }
  printf("SearchByProjection: %d matches, %zu entries equal\n", nm, nc);
  free(ip); free(fp); free(tcw); free(sf); free(l_obs); free(l_xw); free(l_desc); free(l_oct); free(l_ang);
  free(c_x); free(c_y); free(c_oct); free(c_ang); free(c_desc); free(c_obs); free(want); free(want_n); free(got);
  return 0;
}

static int test_knn(void) {
  size_t nq, ndb, i;
  unsigned char* q = (unsigned char*)read_file("knn_q.raw", 32, &nq);
  unsigned char* db = (unsigned char*)read_file("knn_db.raw", 32, &ndb);
  float* ratio = (float*)read_file("knn_ratio.raw", sizeof(float), NULL);
  int* wb = (int*)read_file("expect_knn_best.raw", sizeof(int), NULL);
  int* ws = (int*)read_file("expect_knn_second.raw", sizeof(int), NULL);
  int* wi = (int*)read_file("expect_knn_idx.raw", sizeof(int), NULL);
  unsigned char* wa = (unsigned char*)read_file("expect_knn_accept.raw", 1, NULL);
  int* best = (int*)malloc(sizeof(int) * nq);
  int* second = (int*)malloc(sizeof(int) * nq);
  int* idx = (int*)malloc(sizeof(int) * nq);
  unsigned char* acc = (unsigned char*)malloc(nq);
  int d = -1;
  CHECK(orb_knn2_bruteforce(0, q, (int)nq, db, (int)ndb, ratio[0], best, second, idx, acc));
  for (i = 0; i < nq; i++)
    if (best[i] != wb[i] || second[i] != ws[i] || idx[i] != wi[i] || acc[i] != wa[i]) {
      fprintf(stderr, "kNN-2 query %zu: (%d, %d, %d, %d), expected (%d, %d, %d, %d)\n", i, best[i], second[i], idx[i], acc[i],
              wb[i], ws[i], wi[i], wa[i]);
      return 1;
    }
  /* ORBmatcher::DescriptorDistance on the first query against its best row */
  if (idx[0] >= 0) {
    CHECK(orb_hamming_distance_pairs(0, q, db + 32 * (size_t)idx[0], 1, &d));
    if (d != best[0]) { fprintf(stderr, "DescriptorDistance %d, kNN-2 best %d\n", d, best[0]); return 1; }
  }
  printf("kNN-2: %zu queries x %zu rows equal\n", nq, ndb);
  free(q); free(db); free(ratio); free(wb); free(ws); free(wi); free(wa); free(best); free(second); free(idx); free(acc);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s <directory with the raw arrays>\n", argv[0]); return 2; }
  snprintf(g_dir, sizeof g_dir, "%s", argv[1]);
  if (orb_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", orb_last_error()); return 3; }
  if (test_extract() || test_match() || test_knn()) return 1;
  printf("abi_smoke ok\n");
  return 0;
}

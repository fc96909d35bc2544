/*
 * o3d_oracle.c -- CPU restatement of the PolyCam/open3d ICP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline.  Nothing under open3d_amd/ may import, link or call it; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED by the reference's own tests: the reference fork ships no
 * tests, fixtures or golden vectors, and Eigen3 / FLANN
 * (un-vendored, version unpinned in the reference's CMakeLists.txt) are absent
 * from the image, so the real library cannot be built.  The restatement is
 * pinned instead by (i) the reference's own call sites cited per function,
 * (ii) independent numpy/scipy cross-checks (tests/test_oracle_*.py) and
 * (iii) committed fixtures made by tests/golden/make_golden.py.
 *
 * All file:line citations are relative to /root/reference/open3d/.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 * Plain C99, double precision everywhere, no FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Threads                                                                    */
/* ------------------------------------------------------------------------- */
static double orc_wall_seconds(void) { /* (the serial build has no omp_get_wtime) */
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
ORC_EXPORT int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
ORC_EXPORT void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* Distance: flann::L2<double> on dim 3 (third-party FLANN 1.9.x, absent from  */
/* the image; call sites geometry/KDTreeFlann.cpp:106,126,148).  The tail loop */
/* accumulates result += diff*diff for x,y,z from 0.0:  ((dx*dx)+dy*dy)+dz*dz. */
/* ------------------------------------------------------------------------- */
static inline double l2sq3(const double *a, const double *b) {
  double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  double r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* Radius as the reference hands it to FLANN: float(radius*radius), widened
 * back to double (geometry/KDTreeFlann.cpp:126,148). */
ORC_EXPORT double orc_radius_threshold(double radius) { return (double)(float)(radius * radius); }

/* ------------------------------------------------------------------------- */
/* Result set: (d2, idx)-lexicographic k-best within a strict threshold.       */
/* FLANN's KNNSimpleResultSet / KNNRadiusResultSet / RadiusResultSet keep     */
/* first-visited among exact ties; the project contract (BASELINE.json         */
/* north_star) is "ties broken by lowest target index", which is what this     */
/* set implements.  Strict accept: d2 < thr (KNNRadiusResultSet::addPoint      */
/* rejects dist >= worst_dist_).                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
  int cap;      /* <=0: unbounded (RadiusResultSet) */
  int count;
  double thr;   /* strict upper bound on d2 (INFINITY for pure KNN) */
  int *idx;
  double *d2;
  int alloc;    /* capacity of idx/d2 arrays when unbounded (growable) */
  int owns;
} rset;

static inline int lex_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

static inline double rset_worst(const rset *s) {
  if (s->cap > 0 && s->count == s->cap) return s->d2[s->count - 1];
  return s->thr;
}

static void rset_add(rset *s, double d, int id) {
  if (!(d < s->thr)) return;
  if (s->cap > 0) {
    if (s->count == s->cap) {
      if (!lex_less(d, id, s->d2[s->count - 1], s->idx[s->count - 1])) return;
    } else {
      s->count++;
    }
    int i = s->count - 1;
    while (i > 0 && lex_less(d, id, s->d2[i - 1], s->idx[i - 1])) {
      s->d2[i] = s->d2[i - 1];
      s->idx[i] = s->idx[i - 1];
      i--;
    }
    s->d2[i] = d;
    s->idx[i] = id;
  } else {
    if (s->count == s->alloc) {
      int na = s->alloc ? s->alloc * 2 : 64;
      s->idx = (int *)realloc(s->idx, sizeof(int) * (size_t)na);
      s->d2 = (double *)realloc(s->d2, sizeof(double) * (size_t)na);
      s->alloc = na;
    }
    s->idx[s->count] = id;
    s->d2[s->count] = d;
    s->count++;
  }
}

static void rset_sort_unbounded(rset *s) {
  /* insertion sort is fine for radius neighbourhoods (tens of points); use
   * shell-ish fallback for bigger sets */
  int n = s->count;
  for (int gap = n / 2; gap > 0; gap /= 2) {
    for (int i = gap; i < n; i++) {
      double d = s->d2[i];
      int id = s->idx[i];
      int j = i;
      while (j >= gap && lex_less(d, id, s->d2[j - gap], s->idx[j - gap])) {
        s->d2[j] = s->d2[j - gap];
        s->idx[j] = s->idx[j - gap];
        j -= gap;
      }
      s->d2[j] = d;
      s->idx[j] = id;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* FLANN-shaped single KD-tree (KDTreeSingleIndex, leaf_max_size 15,           */
/* reorder=true) as configured at geometry/KDTreeFlann.cpp:164.  Third-party   */
/* algorithm restated from the published FLANN 1.9.x sources (from memory;     */
/* SURVEY.md Appendix B); the result contract is exact search, verified        */
/* against orc_brute_* below.                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
  int child1, child2; /* -1 for leaf */
  int left, right;    /* leaf: [left,right) into reordered data */
  int divfeat;
  double divlow, divhigh;
} kdnode;

typedef struct orc_kdtree {
  int64_t n;
  double *data; /* reordered copy, 3*n */
  int *vind;    /* reordered position -> original index */
  kdnode *nodes;
  int n_nodes, cap_nodes;
  double bbox[6]; /* low0,high0,low1,high1,low2,high2 */
  const double *orig;
} orc_kdtree;

#define KD_LEAF 15

static int kd_new_node(orc_kdtree *t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kdnode *)realloc(t->nodes, sizeof(kdnode) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}

static void kd_minmax(const orc_kdtree *t, const int *ind, int count, int dim, double *mn, double *mx) {
  double a = t->orig[3 * (size_t)ind[0] + dim], b = a;
  for (int i = 1; i < count; i++) {
    double v = t->orig[3 * (size_t)ind[i] + dim];
    if (v < a) a = v;
    if (v > b) b = v;
  }
  *mn = a;
  *mx = b;
}

static void kd_plane_split(const orc_kdtree *t, int *ind, int count, int cutfeat, double cutval, int *lim1, int *lim2) {
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && t->orig[3 * (size_t)ind[left] + cutfeat] < cutval) ++left;
    while (left <= right && t->orig[3 * (size_t)ind[right] + cutfeat] >= cutval) --right;
    if (left > right) break;
    int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left; --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && t->orig[3 * (size_t)ind[left] + cutfeat] <= cutval) ++left;
    while (left <= right && t->orig[3 * (size_t)ind[right] + cutfeat] > cutval) --right;
    if (left > right) break;
    int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left; --right;
  }
  *lim2 = left;
}

static int kd_divide(orc_kdtree *t, int *ind_base, int left, int right, double *bbox) {
  int me = kd_new_node(t);
  int count = right - left;
  int *ind = ind_base + left;
  if (count <= KD_LEAF) {
    t->nodes[me].child1 = t->nodes[me].child2 = -1;
    t->nodes[me].left = left;
    t->nodes[me].right = right;
    for (int d = 0; d < 3; d++) {
      double mn, mx;
      kd_minmax(t, ind, count, d, &mn, &mx);
      bbox[2 * d] = mn;
      bbox[2 * d + 1] = mx;
    }
    return me;
  }
  /* middleSplit_ */
  int cutfeat = 0;
  double max_span = bbox[1] - bbox[0];
  for (int d = 1; d < 3; d++) {
    double span = bbox[2 * d + 1] - bbox[2 * d];
    if (span > max_span) { max_span = span; cutfeat = d; }
  }
  double mn, mx;
  kd_minmax(t, ind, count, cutfeat, &mn, &mx);
  double cutval = (mn + mx) / 2;
  max_span = mx - mn;
  int k = cutfeat;
  for (int d = 0; d < 3; d++) {
    if (d == k) continue;
    double span = bbox[2 * d + 1] - bbox[2 * d];
    if (span > max_span) {
      kd_minmax(t, ind, count, d, &mn, &mx);
      span = mx - mn;
      if (span > max_span) { max_span = span; cutfeat = d; cutval = (mn + mx) / 2; }
    }
  }
  int lim1, lim2, idx;
  kd_plane_split(t, ind, count, cutfeat, cutval, &lim1, &lim2);
  if (lim1 > count / 2) idx = lim1;
  else if (lim2 < count / 2) idx = lim2;
  else idx = count / 2;
  if (idx <= 0 || idx >= count) idx = count / 2; /* all-equal degenerate guard */

  double lb[6], rb[6];
  memcpy(lb, bbox, sizeof lb);
  memcpy(rb, bbox, sizeof rb);
  lb[2 * cutfeat + 1] = cutval;
  rb[2 * cutfeat] = cutval;
  int c1 = kd_divide(t, ind_base, left, left + idx, lb);
  int c2 = kd_divide(t, ind_base, left + idx, right, rb);
  kdnode *nd = &t->nodes[me];
  nd->child1 = c1;
  nd->child2 = c2;
  nd->divfeat = cutfeat;
  nd->divlow = lb[2 * cutfeat + 1];
  nd->divhigh = rb[2 * cutfeat];
  for (int d = 0; d < 3; d++) {
    bbox[2 * d] = lb[2 * d] < rb[2 * d] ? lb[2 * d] : rb[2 * d];
    bbox[2 * d + 1] = lb[2 * d + 1] > rb[2 * d + 1] ? lb[2 * d + 1] : rb[2 * d + 1];
  }
  return me;
}

/* KDTreeFlann::SetRawData, geometry/KDTreeFlann.cpp:154-167 */
ORC_EXPORT orc_kdtree *orc_kdtree_build(const double *pts, int64_t n) {
  if (n <= 0) return NULL;
  orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof *t);
  t->n = n;
  t->orig = pts;
  t->vind = (int *)malloc(sizeof(int) * (size_t)n);
  for (int64_t i = 0; i < n; i++) t->vind[i] = (int)i;
  for (int d = 0; d < 3; d++) {
    double mn = pts[d], mx = pts[d];
    for (int64_t i = 1; i < n; i++) {
      double v = pts[3 * i + d];
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
    t->bbox[2 * d] = mn;
    t->bbox[2 * d + 1] = mx;
  }
  double bb[6];
  memcpy(bb, t->bbox, sizeof bb);
  kd_divide(t, t->vind, 0, (int)n, bb);
  t->data = (double *)malloc(sizeof(double) * 3 * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    const double *p = pts + 3 * (size_t)t->vind[i];
    t->data[3 * i] = p[0];
    t->data[3 * i + 1] = p[1];
    t->data[3 * i + 2] = p[2];
  }
  t->orig = NULL;
  return t;
}

ORC_EXPORT void orc_kdtree_free(orc_kdtree *t) {
  if (!t) return;
  free(t->data);
  free(t->vind);
  free(t->nodes);
  free(t);
}

static void kd_search_level(const orc_kdtree *t, rset *rs, const double *q, int ni, double mindistsq, double *dists) {
  const kdnode *nd = &t->nodes[ni];
  if (nd->child1 < 0) {
    for (int i = nd->left; i < nd->right; i++) {
      double d = l2sq3(q, t->data + 3 * (size_t)i);
      if (d <= rset_worst(rs)) rset_add(rs, d, t->vind[i]);
    }
    return;
  }
  int f = nd->divfeat;
  double val = q[f];
  double diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
  int best, other;
  double cut;
  if (diff1 + diff2 < 0) { best = nd->child1; other = nd->child2; cut = diff2 * diff2; }
  else { best = nd->child2; other = nd->child1; cut = diff1 * diff1; }
  kd_search_level(t, rs, q, best, mindistsq, dists);
  double dst = dists[f];
  mindistsq = mindistsq + cut - dst;
  dists[f] = cut;
  if (mindistsq <= rset_worst(rs)) kd_search_level(t, rs, q, other, mindistsq, dists);
  dists[f] = dst;
}

static void kd_find(const orc_kdtree *t, rset *rs, const double *q) {
  double dists[3] = {0, 0, 0};
  double distsq = 0;
  for (int d = 0; d < 3; d++) {
    if (q[d] < t->bbox[2 * d]) dists[d] = (q[d] - t->bbox[2 * d]) * (q[d] - t->bbox[2 * d]);
    if (q[d] > t->bbox[2 * d + 1]) dists[d] = (q[d] - t->bbox[2 * d + 1]) * (q[d] - t->bbox[2 * d + 1]);
    distsq += dists[d];
  }
  if (q[0] != q[0] || q[1] != q[1] || q[2] != q[2]) return; /* NaN query: nothing passes d2 < thr */
  kd_search_level(t, rs, q, 0, distsq, dists);
}

/* KDTreeFlann::SearchHybrid, geometry/KDTreeFlann.cpp:132-152 */
ORC_EXPORT int orc_kdtree_search_hybrid(const orc_kdtree *t, const double *q, double radius, int max_nn, int *idx, double *d2) {
  if (!t || max_nn < 0) return -1;
  if (max_nn == 0) return 0;
  rset rs = {max_nn, 0, orc_radius_threshold(radius), idx, d2, 0, 0};
  kd_find(t, &rs, q);
  return rs.count;
}

/* KDTreeFlann::SearchKNN, geometry/KDTreeFlann.cpp:93-110 */
ORC_EXPORT int orc_kdtree_search_knn(const orc_kdtree *t, const double *q, int knn, int *idx, double *d2) {
  if (!t || knn < 0) return -1;
  if (knn == 0) return 0;
  rset rs = {knn, 0, INFINITY, idx, d2, 0, 0};
  kd_find(t, &rs, q);
  return rs.count;
}

/* KDTreeFlann::SearchRadius, geometry/KDTreeFlann.cpp:112-130.  Outputs are
 * malloc'ed; caller frees with orc_free. */
ORC_EXPORT int orc_kdtree_search_radius(const orc_kdtree *t, const double *q, double radius, int **idx, double **d2) {
  if (!t) return -1;
  rset rs = {0, 0, orc_radius_threshold(radius), NULL, NULL, 0, 1};
  kd_find(t, &rs, q);
  rset_sort_unbounded(&rs);
  *idx = rs.idx;
  *d2 = rs.d2;
  return rs.count;
}
ORC_EXPORT void orc_free(void *p) { free(p); }

/* Brute-force ground truth for the same contracts (small N). */
ORC_EXPORT int orc_brute_search(const double *tgt, int64_t n, const double *q, double thr, int cap, int *idx, double *d2) {
  rset rs = {cap, 0, thr, idx, d2, 0, 0};
  for (int64_t i = 0; i < n; i++) rset_add(&rs, l2sq3(q, tgt + 3 * i), (int)i);
  return rs.count;
}

/* ------------------------------------------------------------------------- */
/* GetRegistrationResultAndCorrespondences,                                    */
/* pipelines/registration/Registration.cpp:41-91.                              */
/* corr[i] = target index or -1 (the reference's CorrespondenceSet is the      */
/* ascending-i list of (i, corr[i]) with corr[i] >= 0 when run serially).      */
/* use_brute!=0 replaces the KD-tree by exhaustive search.                     */
/* ------------------------------------------------------------------------- */
ORC_EXPORT void orc_correspondences(const double *src, int64_t ns, const double *tgt, int64_t nt, const orc_kdtree *kd,
                                    double max_dist, int use_brute, int32_t *corr, int64_t *n_corr, double *sum_d2,
                                    double *fitness, double *rmse) {
  int64_t nc = 0;
  double err2 = 0.0;
  if (max_dist <= 0.0) { /* Registration.cpp:45-47 */
    for (int64_t i = 0; i < ns; i++) corr[i] = -1;
    *n_corr = 0; *sum_d2 = 0; *fitness = 0; *rmse = 0;
    return;
  }
  double thr = orc_radius_threshold(max_dist);
#ifdef _OPENMP
#pragma omp parallel
  {
    double e_priv = 0.0;
    int64_t c_priv = 0;
#pragma omp for schedule(static) nowait
    for (int64_t i = 0; i < ns; i++) {
      int id;
      double d;
      int k = use_brute ? orc_brute_search(tgt, nt, src + 3 * i, thr, 1, &id, &d)
                        : orc_kdtree_search_hybrid(kd, src + 3 * i, max_dist, 1, &id, &d);
      if (k > 0) { corr[i] = id; e_priv += d; c_priv++; } else corr[i] = -1;
    }
#pragma omp critical
    { err2 += e_priv; nc += c_priv; }
  }
  if (orc_max_threads() > 1) { /* make the sum thread-count independent: redo serially in index order */
    err2 = 0.0;
    for (int64_t i = 0; i < ns; i++)
      if (corr[i] >= 0) err2 += l2sq3(src + 3 * i, tgt + 3 * (size_t)corr[i]);
  }
#else
  for (int64_t i = 0; i < ns; i++) {
    int id;
    double d;
    int k = use_brute ? orc_brute_search(tgt, nt, src + 3 * i, thr, 1, &id, &d)
                      : orc_kdtree_search_hybrid(kd, src + 3 * i, max_dist, 1, &id, &d);
    if (k > 0) { corr[i] = id; err2 += d; nc++; } else corr[i] = -1;
  }
#endif
  *n_corr = nc;
  *sum_d2 = err2;
  if (nc == 0) { *fitness = 0.0; *rmse = 0.0; } /* Registration.cpp:82-84 */
  else { *fitness = (double)nc / (double)ns; *rmse = sqrt(err2 / (double)nc); } /* :86-88 */
}

/* ------------------------------------------------------------------------- */
/* Geometry3D::TransformPoints / TransformNormals, geometry/Geometry3D.cpp:77-89 */
/* T is column-major (Eigen::Matrix4d default).  Product order fixed as        */
/* ((T(r,0)*x + T(r,1)*y) + T(r,2)*z) + T(r,3)*w  (SURVEY.md Appendix C).       */
/* ------------------------------------------------------------------------- */
#define TM(T, r, c) ((T)[(c) * 4 + (r)])
ORC_EXPORT void orc_transform_points(const double *T, double *pts, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    double o[4];
    for (int r = 0; r < 4; r++) {
      double a = TM(T, r, 0) * x;
      a = a + TM(T, r, 1) * y;
      a = a + TM(T, r, 2) * z;
      a = a + TM(T, r, 3) * 1.0;
      o[r] = a;
    }
    pts[3 * i] = o[0] / o[3];
    pts[3 * i + 1] = o[1] / o[3];
    pts[3 * i + 2] = o[2] / o[3];
  }
}
ORC_EXPORT void orc_transform_normals(const double *T, double *nrm, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    double x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
    for (int r = 0; r < 3; r++) {
      double a = TM(T, r, 0) * x;
      a = a + TM(T, r, 1) * y;
      a = a + TM(T, r, 2) * z;
      a = a + TM(T, r, 3) * 0.0;
      nrm[3 * i + r] = a;
    }
  }
}

/* Eigen::Matrix4d::isIdentity() with the default dummy_precision 1e-12
 * (Registration.cpp:123,150; third-party Eigen, restated from memory). */
ORC_EXPORT int orc_is_identity(const double *T) {
  const double prec = 1e-12;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double v = TM(T, r, c);
      if (r == c) {
        double m = fabs(v) < 1.0 ? fabs(v) : 1.0;
        if (!(fabs(v - 1.0) <= prec * m)) return 0;
      } else {
        if (!(fabs(v) <= prec)) return 0;
      }
    }
  return 1;
}

static void mat4_mul(const double *A, const double *B, double *C) { /* C = A*B, column-major */
  double R[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double a = TM(A, r, 0) * TM(B, 0, c);
      a = a + TM(A, r, 1) * TM(B, 1, c);
      a = a + TM(A, r, 2) * TM(B, 2, c);
      a = a + TM(A, r, 3) * TM(B, 3, c);
      R[c * 4 + r] = a;
    }
  memcpy(C, R, sizeof R);
}
ORC_EXPORT void orc_mat4_mul(const double *A, const double *B, double *C) { mat4_mul(A, B, C); }

static void mat4_identity(double *T) {
  memset(T, 0, 16 * sizeof(double));
  T[0] = T[5] = T[10] = T[15] = 1.0;
}

/* ------------------------------------------------------------------------- */
/* utility::ComputeJTJandJTr with the point-to-plane lambda                    */
/* (utility/Eigen.cpp:159-201; TransformationEstimation.cpp:79-86), serial      */
/* order.  JTJ row-major 6x6 (symmetric), JTr 6, r2 = sum r^2.                 */
/* ------------------------------------------------------------------------- */
ORC_EXPORT void orc_p2plane_jtj(const double *src, const double *tgt, const double *nrm, const int32_t *corr, int64_t ns,
                                double *JTJ, double *JTr, double *r2_out) {
  memset(JTJ, 0, 36 * sizeof(double));
  memset(JTr, 0, 6 * sizeof(double));
  double r2 = 0;
  for (int64_t i = 0; i < ns; i++) {
    if (corr[i] < 0) continue;
    const double *vs = src + 3 * i, *vt = tgt + 3 * (size_t)corr[i], *nt = nrm + 3 * (size_t)corr[i];
    double r = (vs[0] - vt[0]) * nt[0];
    r = r + (vs[1] - vt[1]) * nt[1];
    r = r + (vs[2] - vt[2]) * nt[2];
    double J[6];
    J[0] = vs[1] * nt[2] - vs[2] * nt[1];
    J[1] = vs[2] * nt[0] - vs[0] * nt[2];
    J[2] = vs[0] * nt[1] - vs[1] * nt[0];
    J[3] = nt[0]; J[4] = nt[1]; J[5] = nt[2];
    for (int a = 0; a < 6; a++) {
      for (int b = 0; b < 6; b++) JTJ[a * 6 + b] += J[a] * J[b];
      JTr[a] += J[a] * r;
    }
    r2 += r * r;
  }
  *r2_out = r2;
}

/* ------------------------------------------------------------------------- */
/* A.ldlt().solve(b)  (utility/Eigen.cpp:85).  Third-party Eigen 3.3/3.4 LDLT  */
/* restated: symmetric pivoting on the largest |diagonal|, lower triangle,     */
/* zero pivots in the solve give a zero component.                             */
/* ------------------------------------------------------------------------- */
ORC_EXPORT int orc_ldlt_solve(int n, const double *A_rowmajor, const double *b, double *x) {
  double M[12 * 12];
  int tr[12];
  if (n > 12) return -1;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) M[i * n + j] = A_rowmajor[i * n + j];
#define MM(i, j) M[(i) * n + (j)]
  for (int k = 0; k < n; k++) {
    int p = k;
    double big = fabs(MM(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(MM(i, i)) > big) { big = fabs(MM(i, i)); p = i; }
    tr[k] = p;
    if (p != k) { /* symmetric swap on the lower triangle */
      for (int j = 0; j < k; j++) { double t = MM(k, j); MM(k, j) = MM(p, j); MM(p, j) = t; }
      for (int i = p + 1; i < n; i++) { double t = MM(i, k); MM(i, k) = MM(i, p); MM(i, p) = t; }
      for (int i = k + 1; i < p; i++) { double t = MM(i, k); MM(i, k) = MM(p, i); MM(p, i) = t; }
      { double t = MM(k, k); MM(k, k) = MM(p, p); MM(p, p) = t; }
    }
    if (k > 0) {
      double temp[12];
      for (int j = 0; j < k; j++) temp[j] = MM(j, j) * MM(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += MM(k, j) * temp[j];
      MM(k, k) -= s;
      for (int i = k + 1; i < n; i++) {
        double a = 0;
        for (int j = 0; j < k; j++) a += MM(i, j) * temp[j];
        MM(i, k) -= a;
      }
    }
    double akk = MM(k, k);
    if (fabs(akk) > 0)
      for (int i = k + 1; i < n; i++) MM(i, k) /= akk;
  }
  double y[12];
  for (int i = 0; i < n; i++) y[i] = b[i];
  for (int k = 0; k < n; k++)
    if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= MM(i, j) * y[j];
  for (int i = 0; i < n; i++) {
    if (fabs(MM(i, i)) > DBL_MIN) y[i] /= MM(i, i);
    else y[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= MM(j, i) * y[j];
  for (int k = n - 1; k >= 0; k--)
    if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++) x[i] = y[i];
#undef MM
  return 0;
}

/* utility::TransformVector6dToMatrix4d, utility/Eigen.cpp:89-97:
 * R = Rz(x2) * Ry(x1) * Rx(x0), t = x3..5.  Column-major output. */
ORC_EXPORT void orc_vec6_to_mat4(const double *v, double *T) {
  double sx = sin(v[0]), cx = cos(v[0]);
  double sy = sin(v[1]), cy = cos(v[1]);
  double sz = sin(v[2]), cz = cos(v[2]);
  mat4_identity(T);
  TM(T, 0, 0) = cy * cz;  TM(T, 0, 1) = sx * sy * cz - cx * sz;  TM(T, 0, 2) = cx * sy * cz + sx * sz;
  TM(T, 1, 0) = cy * sz;  TM(T, 1, 1) = sx * sy * sz + cx * cz;  TM(T, 1, 2) = cx * sy * sz - sx * cz;
  TM(T, 2, 0) = -sy;      TM(T, 2, 1) = sx * cy;                 TM(T, 2, 2) = cx * cy;
  TM(T, 0, 3) = v[3]; TM(T, 1, 3) = v[4]; TM(T, 2, 3) = v[5];
}

/* TransformationEstimationPointToPlane::ComputeTransformation,
 * TransformationEstimation.cpp:74-98 (+ SolveJacobianSystemAndObtainExtrinsicMatrix,
 * utility/Eigen.cpp:116-130: solves JTJ x = -JTr). */
ORC_EXPORT void orc_p2plane_compute(const double *src, const double *tgt, const double *nrm, const int32_t *corr, int64_t ns,
                                    double *T, double *JTJ_out, double *JTr_out, double *r2_out) {
  int64_t nc = 0;
  for (int64_t i = 0; i < ns; i++) nc += corr[i] >= 0;
  if (nc == 0 || !nrm) { mat4_identity(T); return; }
  double JTJ[36], JTr[6], r2, mb[6], x[6];
  orc_p2plane_jtj(src, tgt, nrm, corr, ns, JTJ, JTr, &r2);
  for (int i = 0; i < 6; i++) mb[i] = -JTr[i];
  orc_ldlt_solve(6, JTJ, mb, x);
  orc_vec6_to_mat4(x, T);
  if (JTJ_out) memcpy(JTJ_out, JTJ, sizeof JTJ);
  if (JTr_out) memcpy(JTr_out, JTr, sizeof JTr);
  if (r2_out) *r2_out = r2;
}

/* ------------------------------------------------------------------------- */
/* 3x3 SVD by one-sided (Hestenes) Jacobi; stands in for Eigen::JacobiSVD       */
/* inside Eigen::umeyama (TransformationEstimation.cpp:59).  A = U diag(s) V^T, */
/* singular values descending, U and V orthogonal (row-major 3x3).              */
/* ------------------------------------------------------------------------- */
static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static double det3(const double *M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
ORC_EXPORT void orc_svd3(const double *A, double *U, double *S, double *V) {
  double W[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(W, A, sizeof W);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; i++) {
          alpha += W[i * 3 + p] * W[i * 3 + p];
          beta += W[i * 3 + q] * W[i * 3 + q];
          gamma += W[i * 3 + p] * W[i * 3 + q];
        }
        if (gamma == 0.0) continue;
        double lim = 1e-32 * alpha * beta;
        if (gamma * gamma <= lim) continue;
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; i++) {
          double wp = W[i * 3 + p], wq = W[i * 3 + q];
          W[i * 3 + p] = c * wp - s * wq;
          W[i * 3 + q] = s * wp + c * wq;
          double vp = Vm[i * 3 + p], vq = Vm[i * 3 + q];
          Vm[i * 3 + p] = c * vp - s * vq;
          Vm[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3];
  for (int j = 0; j < 3; j++) sv[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; a++)
    for (int b = a + 1; b < 3; b++)
      if (sv[ord[b]] > sv[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
  double Uc[3][3];
  int have[3] = {0, 0, 0};
  for (int j = 0; j < 3; j++) {
    int o = ord[j];
    S[j] = sv[o];
    for (int i = 0; i < 3; i++) V[i * 3 + j] = Vm[i * 3 + o];
    if (sv[o] > 1e-300 && sv[o] > sv[ord[0]] * 1e-14) {
      for (int i = 0; i < 3; i++) Uc[j][i] = W[i * 3 + o] / sv[o];
      have[j] = 1;
    }
  }
  /* complete U to an orthonormal basis when rank-deficient */
  if (!have[0]) { Uc[0][0] = 1; Uc[0][1] = 0; Uc[0][2] = 0; have[0] = 1; }
  if (!have[1]) {
    double e[3] = {0, 0, 0};
    int m = fabs(Uc[0][0]) < fabs(Uc[0][1]) ? (fabs(Uc[0][0]) < fabs(Uc[0][2]) ? 0 : 2) : (fabs(Uc[0][1]) < fabs(Uc[0][2]) ? 1 : 2);
    e[m] = 1;
    double c[3];
    cross3(Uc[0], e, c);
    double nrm = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int i = 0; i < 3; i++) Uc[1][i] = c[i] / nrm;
    have[1] = 1;
  }
  if (!have[2]) {
    double c[3];
    cross3(Uc[0], Uc[1], c);
    double nrm = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int i = 0; i < 3; i++) Uc[2][i] = c[i] / nrm;
  }
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) U[i * 3 + j] = Uc[j][i];
}

/* TransformationEstimationPointToPoint::ComputeTransformation,
 * TransformationEstimation.cpp:49-60, with Eigen::umeyama restated (third-party
 * Eigen 3.3/3.4 Geometry/Umeyama.h): two-pass means,
 * sigma = (1/n) dst_demean * src_demean^T, SVD, S(2) = -1 iff det(U)det(V) < 0. */
ORC_EXPORT void orc_p2p_compute(const double *src, const double *tgt, const int32_t *corr, int64_t ns, int with_scaling,
                                double *T) {
  int64_t n = 0;
  double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
  for (int64_t i = 0; i < ns; i++) {
    if (corr[i] < 0) continue;
    n++;
    for (int d = 0; d < 3; d++) { ms[d] += src[3 * i + d]; md[d] += tgt[3 * (size_t)corr[i] + d]; }
  }
  mat4_identity(T);
  if (n == 0) return; /* :51-52 */
  double inv_n = 1.0 / (double)n;
  for (int d = 0; d < 3; d++) { ms[d] *= inv_n; md[d] *= inv_n; }
  double sig[9] = {0}, var = 0;
  for (int64_t i = 0; i < ns; i++) {
    if (corr[i] < 0) continue;
    double s[3], t[3];
    for (int d = 0; d < 3; d++) { s[d] = src[3 * i + d] - ms[d]; t[d] = tgt[3 * (size_t)corr[i] + d] - md[d]; }
    var += s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) sig[a * 3 + b] += t[a] * s[b];
  }
  var *= inv_n;
  for (int k = 0; k < 9; k++) sig[k] *= inv_n;
  double U[9], S[3], V[9];
  orc_svd3(sig, U, S, V);
  double sgn[3] = {1, 1, 1};
  if (det3(U) * det3(V) < 0) sgn[2] = -1;
  double R[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += U[a * 3 + k] * sgn[k] * V[b * 3 + k];
      R[a * 3 + b] = acc;
    }
  double c = 1.0;
  if (with_scaling) c = 1.0 / var * (S[0] * sgn[0] + S[1] * sgn[1] + S[2] * sgn[2]);
  for (int a = 0; a < 3; a++) {
    double rm = R[a * 3] * ms[0] + R[a * 3 + 1] * ms[1] + R[a * 3 + 2] * ms[2];
    TM(T, a, 3) = md[a] - c * rm;
    for (int b = 0; b < 3; b++) TM(T, a, b) = c * R[a * 3 + b];
  }
}

/* ComputeRMSE, TransformationEstimation.cpp:38-47 (p2p) and :62-72 (p2plane). */
ORC_EXPORT double orc_compute_rmse(const double *src, const double *tgt, const double *nrm, const int32_t *corr, int64_t ns,
                                   int est_type) {
  int64_t n = 0;
  double err = 0;
  for (int64_t i = 0; i < ns; i++) {
    if (corr[i] < 0) continue;
    n++;
    const double *vs = src + 3 * i, *vt = tgt + 3 * (size_t)corr[i];
    if (est_type == 2) {
      if (!nrm) return 0.0;
      const double *nt = nrm + 3 * (size_t)corr[i];
      double r = (vs[0] - vt[0]) * nt[0];
      r = r + (vs[1] - vt[1]) * nt[1];
      r = r + (vs[2] - vt[2]) * nt[2];
      err += r * r;
    } else {
      double dx = vs[0] - vt[0], dy = vs[1] - vt[1], dz = vs[2] - vt[2];
      err += dx * dx + dy * dy + dz * dz;
    }
  }
  if (n == 0) return 0.0;
  return sqrt(err / (double)n);
}

/* ------------------------------------------------------------------------- */
/* RegistrationICP, Registration.cpp:129-168.                                  */
/* est_type: 1 = PointToPoint, 2 = PointToPlane (TransformationEstimation.h:45-50) */
/* Returns 0 ok; 1 = invalid max_correspondence_distance (:134-136);           */
/* 2 = point-to-plane without target normals (:137-144).                       */
/* log (nullable): per LOOP-A pass p (0..iters) 20 doubles:                    */
/*   [0]=fitness [1]=rmse [2]=n_corr [3]=sum_d2 [4..19]=cumulative T colmajor.  */
/* ------------------------------------------------------------------------- */
ORC_EXPORT int orc_registration_icp(const double *src, int64_t ns, const double *tgt, const double *tgt_nrm, int64_t nt,
                                    double max_dist, const double *init, int est_type, int with_scaling, double rel_fitness,
                                    double rel_rmse, int max_iter, int use_brute, double *T_out, int32_t *corr_out,
                                    int64_t *n_corr_out, double *fitness_out, double *rmse_out, int *iters_out, double *log,
                                    double *time_loop_s) {
  if (max_dist <= 0.0) return 1;
  if (est_type == 2 && !tgt_nrm) return 2;
  double T[16];
  memcpy(T, init, sizeof T);
  orc_kdtree *kd = use_brute ? NULL : orc_kdtree_build(tgt, nt);
  double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
  memcpy(pcd, src, sizeof(double) * 3 * (size_t)ns);
  if (!orc_is_identity(init)) orc_transform_points(init, pcd, ns);
  int64_t nc;
  double sd2, fit, rmse;
  double t0 = orc_wall_seconds();
  orc_correspondences(pcd, ns, tgt, nt, kd, max_dist, use_brute, corr_out, &nc, &sd2, &fit, &rmse);
  int pass = 0;
  if (log) { log[0] = fit; log[1] = rmse; log[2] = (double)nc; log[3] = sd2; memcpy(log + 4, T, sizeof T); }
  int it = 0;
  for (it = 0; it < max_iter; it++) {
    double upd[16];
    if (est_type == 2) orc_p2plane_compute(pcd, tgt, tgt_nrm, corr_out, ns, upd, NULL, NULL, NULL);
    else orc_p2p_compute(pcd, tgt, corr_out, ns, with_scaling, upd);
    mat4_mul(upd, T, T);
    orc_transform_points(upd, pcd, ns);
    double bf = fit, br = rmse;
    orc_correspondences(pcd, ns, tgt, nt, kd, max_dist, use_brute, corr_out, &nc, &sd2, &fit, &rmse);
    pass++;
    if (log) {
      double *L = log + 20 * pass;
      L[0] = fit; L[1] = rmse; L[2] = (double)nc; L[3] = sd2; memcpy(L + 4, T, sizeof T);
    }
    if (fabs(bf - fit) < rel_fitness && fabs(br - rmse) < rel_rmse) { it++; break; }
  }
  if (time_loop_s) *time_loop_s = orc_wall_seconds() - t0;
  memcpy(T_out, T, sizeof T);
  *n_corr_out = nc;
  *fitness_out = fit;
  *rmse_out = rmse;
  *iters_out = it; /* number of estimation updates applied == LOOP-A passes - 1 */
  free(pcd);
  orc_kdtree_free(kd);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Colored ICP, pipelines/registration/ColoredICP.cpp (row "next", rank 2).     */
/* ------------------------------------------------------------------------- */
static inline double intensity3(const double *c) { return (c[0] + c[1] + c[2]) / 3.0; }

/* InitializePointCloudForColoredICP, ColoredICP.cpp:68-123: per target point the least-squares colour gradient in its
 * tangent plane over the hybrid neighbourhood (radius, max_nn), zero when fewer than 4 neighbours.  grad: n x 3. */
ORC_EXPORT void orc_color_gradient(const double *pts, const double *nrm, const double *col, int64_t n, double radius, int max_nn,
                                   double *grad) {
  orc_kdtree *kd = orc_kdtree_build(pts, n);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int *idx = (int *)malloc(sizeof(int) * (size_t)(max_nn > 0 ? max_nn : 1));
    double *d2 = (double *)malloc(sizeof(double) * (size_t)(max_nn > 0 ? max_nn : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t k = 0; k < n; k++) {
      const double *vt = pts + 3 * k, *nt = nrm + 3 * k;
      const double it = intensity3(col + 3 * k);
      double *g = grad + 3 * k;
      g[0] = g[1] = g[2] = 0.0;
      const int nn = orc_kdtree_search_hybrid(kd, vt, radius, max_nn, idx, d2);
      if (nn < 4) continue;
      double AtA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Atb[3] = {0, 0, 0};
      for (int i = 1; i <= nn; i++) { /* rows 0..nn-2 from neighbours 1..nn-1, row nn-1 the orthogonal constraint */
        double a[3], b;
        if (i < nn) {
          const double *va = pts + 3 * (size_t)idx[i];
          const double dot = ((va[0] - vt[0]) * nt[0] + (va[1] - vt[1]) * nt[1]) + (va[2] - vt[2]) * nt[2];
          for (int d = 0; d < 3; d++) a[d] = (va[d] - dot * nt[d]) - vt[d];
          b = intensity3(col + 3 * (size_t)idx[i]) - it;
        } else {
          for (int d = 0; d < 3; d++) a[d] = (double)(nn - 1) * nt[d];
          b = 0.0;
        }
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) AtA[r * 3 + c] += a[r] * a[c];
          Atb[r] += a[r] * b;
        }
      }
      orc_ldlt_solve(3, AtA, Atb, g); /* SolveLinearSystemPSD without checks: always "success" (utility/Eigen.cpp:85) */
    }
    free(idx);
    free(d2);
  }
  orc_kdtree_free(kd);
}

/* the two residuals and Jacobian rows of one correspondence, ColoredICP.cpp:136-165 */
static void colored_rows(const double *vs, double is, const double *vt, const double *nt, double it, const double *dit, double slg,
                         double slp, double *J0, double *r0, double *J1, double *r1) {
  double c[3];
  cross3(vs, nt, c);
  for (int d = 0; d < 3; d++) { J0[d] = slg * c[d]; J0[3 + d] = slg * nt[d]; }
  const double dot = ((vs[0] - vt[0]) * nt[0] + (vs[1] - vt[1]) * nt[1]) + (vs[2] - vt[2]) * nt[2];
  *r0 = slg * dot;
  double vp[3];
  for (int d = 0; d < 3; d++) vp[d] = vs[d] - dot * nt[d];
  const double is0 = ((dit[0] * (vp[0] - vt[0]) + dit[1] * (vp[1] - vt[1])) + dit[2] * (vp[2] - vt[2])) + it;
  double M[9], ditM[3];
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) M[r * 3 + cc] = (r == cc ? 1.0 : 0.0) - nt[r] * nt[cc];
  for (int cc = 0; cc < 3; cc++) ditM[cc] = ((-dit[0]) * M[0 * 3 + cc] + (-dit[1]) * M[1 * 3 + cc]) + (-dit[2]) * M[2 * 3 + cc];
  cross3(vs, ditM, c);
  for (int d = 0; d < 3; d++) { J1[d] = slp * c[d]; J1[3 + d] = slp * ditM[d]; }
  *r1 = slp * (is - is0);
}

/* TransformationEstimationForColoredICP::ComputeTransformation, ColoredICP.cpp:125-177.  corr[i] = target index or -1. */
ORC_EXPORT void orc_colored_compute(const double *src, const double *src_col, const double *tgt, const double *tgt_nrm,
                                    const double *tgt_col, const double *grad, const int32_t *corr, int64_t ns, double lambda_geometric,
                                    double *T, double *JTJ_out, double *JTr_out) {
  int64_t nc = 0;
  for (int64_t i = 0; i < ns; i++) nc += corr[i] >= 0;
  mat4_identity(T);
  if (nc == 0 || !tgt_nrm || !tgt_col || !src_col) return;
  const double slg = sqrt(lambda_geometric), slp = sqrt(1.0 - lambda_geometric);
  double JTJ[36], JTr[6], mb[6], x[6];
  memset(JTJ, 0, sizeof JTJ);
  memset(JTr, 0, sizeof JTr);
  for (int64_t i = 0; i < ns; i++) {
    const int32_t t = corr[i];
    if (t < 0) continue;
    double J[2][6], r[2];
    colored_rows(src + 3 * i, intensity3(src_col + 3 * i), tgt + 3 * (size_t)t, tgt_nrm + 3 * (size_t)t, intensity3(tgt_col + 3 * (size_t)t),
                 grad + 3 * (size_t)t, slg, slp, J[0], &r[0], J[1], &r[1]);
    for (int j = 0; j < 2; j++)
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) JTJ[a * 6 + b] += J[j][a] * J[j][b];
        JTr[a] += J[j][a] * r[j];
      }
  }
  for (int i = 0; i < 6; i++) mb[i] = -JTr[i];
  orc_ldlt_solve(6, JTJ, mb, x);
  orc_vec6_to_mat4(x, T);
  if (JTJ_out) memcpy(JTJ_out, JTJ, sizeof JTJ);
  if (JTr_out) memcpy(JTr_out, JTr, sizeof JTr);
}

/* TransformationEstimationForColoredICP::ComputeRMSE, ColoredICP.cpp:179-203: the SUM of the squared residuals (no mean, no root). */
ORC_EXPORT double orc_colored_rmse(const double *src, const double *src_col, const double *tgt, const double *tgt_nrm,
                                   const double *tgt_col, const double *grad, const int32_t *corr, int64_t ns, double lambda_geometric) {
  const double slg = sqrt(lambda_geometric), slp = sqrt(1.0 - lambda_geometric);
  double residual = 0.0;
  for (int64_t i = 0; i < ns; i++) {
    const int32_t t = corr[i];
    if (t < 0) continue;
    double J0[6], J1[6], r0, r1;
    colored_rows(src + 3 * i, intensity3(src_col + 3 * i), tgt + 3 * (size_t)t, tgt_nrm + 3 * (size_t)t, intensity3(tgt_col + 3 * (size_t)t),
                 grad + 3 * (size_t)t, slg, slp, J0, &r0, J1, &r1);
    residual += r0 * r0 + r1 * r1;
  }
  return residual;
}

/* RegistrationColoredICP, ColoredICP.cpp:205-211: gradients over Hybrid(2 * max_distance, 30), then RegistrationICP
 * (Registration.cpp:129-168) with the coloured estimation.  Returns 0 ok, 1 invalid distance, 2 target without normals. */
ORC_EXPORT int orc_registration_colored_icp(const double *src, const double *src_col, int64_t ns, const double *tgt,
                                            const double *tgt_nrm, const double *tgt_col, int64_t nt, double max_dist, const double *init,
                                            double lambda_geometric, double rel_fitness, double rel_rmse, int max_iter, double *T_out,
                                            int32_t *corr_out, int64_t *n_corr_out, double *fitness_out, double *rmse_out, int *iters_out,
                                            double *grad_out /* nt x 3, nullable */) {
  if (lambda_geometric < 0 || lambda_geometric > 1.0) lambda_geometric = 0.968; /* :50-53 */
  double *grad = (double *)calloc((size_t)(nt > 0 ? nt : 1) * 3, sizeof(double));
  if (tgt_nrm && tgt_col) orc_color_gradient(tgt, tgt_nrm, tgt_col, nt, max_dist * 2.0, 30, grad);
  if (grad_out) memcpy(grad_out, grad, sizeof(double) * 3 * (size_t)nt);
  if (max_dist <= 0.0) { free(grad); return 1; }
  if (!tgt_nrm) { free(grad); return 2; }
  double T[16];
  memcpy(T, init, sizeof T);
  orc_kdtree *kd = orc_kdtree_build(tgt, nt);
  double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
  memcpy(pcd, src, sizeof(double) * 3 * (size_t)ns);
  if (!orc_is_identity(init)) orc_transform_points(init, pcd, ns);
  int64_t nc;
  double sd2, fit, rmse;
  orc_correspondences(pcd, ns, tgt, nt, kd, max_dist, 0, corr_out, &nc, &sd2, &fit, &rmse);
  int it = 0;
  for (it = 0; it < max_iter; it++) {
    double upd[16];
    orc_colored_compute(pcd, src_col, tgt, tgt_nrm, tgt_col, grad, corr_out, ns, lambda_geometric, upd, NULL, NULL);
    mat4_mul(upd, T, T);
    orc_transform_points(upd, pcd, ns);
    double bf = fit, br = rmse;
    orc_correspondences(pcd, ns, tgt, nt, kd, max_dist, 0, corr_out, &nc, &sd2, &fit, &rmse);
    if (fabs(bf - fit) < rel_fitness && fabs(br - rmse) < rel_rmse) { it++; break; }
  }
  memcpy(T_out, T, sizeof T);
  *n_corr_out = nc;
  *fitness_out = fit;
  *rmse_out = rmse;
  *iters_out = it;
  free(pcd);
  free(grad);
  orc_kdtree_free(kd);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* The other KDTreeFlann consumers, geometry/PointCloud.cpp:112-132,394-468,521-544 */
/* mode 0: sqrt of the knn-th smallest d2 (0 when fewer), 1: mean of the knn       */
/* distances (-1 when none), 2: SearchRadius count.                                */
/* ------------------------------------------------------------------------- */
ORC_EXPORT void orc_neighbor_query(int mode, const double *cloud, int64_t nc, const double *q, int64_t nq, int knn, double radius,
                                   double *out, int32_t *out_count) {
  orc_kdtree *kd = orc_kdtree_build(cloud, nc);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int *idx = (int *)malloc(sizeof(int) * (size_t)(knn > 0 ? knn : 1));
    double *d2 = (double *)malloc(sizeof(double) * (size_t)(knn > 0 ? knn : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < nq; i++) {
      if (mode == 2) {
        int *ri = NULL;
        double *rd = NULL;
        const int k = nc > 0 ? orc_kdtree_search_radius(kd, q + 3 * i, radius, &ri, &rd) : 0;
        out_count[i] = k > 0 ? k : 0;
        free(ri);
        free(rd);
        continue;
      }
      const int k = nc > 0 ? orc_kdtree_search_knn(kd, q + 3 * i, knn, idx, d2) : 0;
      if (mode == 0) {
        out[i] = (k >= knn) ? sqrt(d2[knn - 1]) : 0.0;
      } else {
        double mean = -1.0;
        if (k > 0) {
          double sum = 0.0;
          for (int j = 0; j < k; j++) sum += sqrt(d2[j]);
          mean = sum / (double)k;
        }
        out[i] = mean;
      }
    }
    free(idx);
    free(d2);
  }
  orc_kdtree_free(kd);
}

/* ------------------------------------------------------------------------- */
/* FPFH, pipelines/registration/Feature.cpp:39-159.  search_type 0 = KNN,        */
/* 2 = hybrid.  feature: 33 doubles per point (Feature::data_, column-major).    */
/* ------------------------------------------------------------------------- */
static void pair_features(const double *p1, const double *n1, const double *p2, const double *n2, double *f) { /* :39-69 */
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  f[0] = f[1] = f[2] = f[3] = 0.0;
  const double len = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  if (len == 0.0) return;
  double a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  const double angle1 = ((a[0] * d[0] + a[1] * d[1]) + a[2] * d[2]) / len;
  const double angle2 = ((b[0] * d[0] + b[1] * d[1]) + b[2] * d[2]) / len;
  double f2;
  if (acos(fabs(angle1)) > acos(fabs(angle2))) {
    for (int k = 0; k < 3; k++) { a[k] = n2[k]; b[k] = n1[k]; d[k] *= -1.0; }
    f2 = -angle2;
  } else {
    f2 = angle1;
  }
  double v[3], w[3];
  cross3(d, a, v);
  const double vn = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  if (vn == 0.0) return;
  for (int k = 0; k < 3; k++) v[k] /= vn;
  cross3(a, v, w);
  f[3] = len;
  f[2] = f2;
  f[1] = (v[0] * b[0] + v[1] * b[1]) + v[2] * b[2];
  f[0] = atan2((w[0] * b[0] + w[1] * b[1]) + w[2] * b[2], (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]);
}

static int fpfh_bin(double v) {
  int h = (int)floor(v);
  return h < 0 ? 0 : (h >= 11 ? 10 : h);
}

/* KDTreeFlann::Search (KDTreeFlann.cpp:75-91) by parameter type; a radius search (type 1) returns its own arrays in
 * *ridx / *rd2 (ascending distance, to be freed by the caller), the others fill idx / d2 */
static int fpfh_search(const orc_kdtree *kd, const double *q, int search_type, int knn, double radius, int *idx, double *d2,
                       int **ridx, double **rd2) {
  *ridx = NULL; *rd2 = NULL;
  if (search_type == 1) return orc_kdtree_search_radius(kd, q, radius, ridx, rd2);
  if (knn <= 0) return 0;
  return search_type == 0 ? orc_kdtree_search_knn(kd, q, knn, idx, d2) : orc_kdtree_search_hybrid(kd, q, radius, knn, idx, d2);
}

ORC_EXPORT void orc_fpfh(const double *pts, const double *nrm, int64_t n, int search_type, int knn, double radius, double *feature,
                         double *spfh_out /* nullable */) {
  const double kPi = 3.14159265358979323846;
  orc_kdtree *kd = orc_kdtree_build(pts, n);
  double *spfh = (double *)calloc((size_t)(n > 0 ? n : 1) * 33, sizeof(double));
  memset(feature, 0, sizeof(double) * 33 * (size_t)n);
  const int cap = knn > 0 ? knn : 1;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)cap);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < n; i++) { /* ComputeSPFHFeature :78-109 */
      int *ri; double *rd;
      const int k = fpfh_search(kd, pts + 3 * i, search_type, knn, radius, idx, d2, &ri, &rd);
      const int *use = ri ? ri : idx;
      if (k > 1) {
        const double hist_incr = 100.0 / (double)(k - 1);
        double *o = spfh + 33 * i;
        for (int j = 1; j < k; j++) {
          double f[4];
          pair_features(pts + 3 * i, nrm + 3 * i, pts + 3 * (size_t)use[j], nrm + 3 * (size_t)use[j], f);
          o[fpfh_bin(11 * (f[0] + kPi) / (2.0 * kPi))] += hist_incr;
          o[fpfh_bin(11 * (f[1] + 1.0) * 0.5) + 11] += hist_incr;
          o[fpfh_bin(11 * (f[2] + 1.0) * 0.5) + 22] += hist_incr;
        }
      }
      free(ri); free(rd);
    }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < n; i++) { /* :127-157 */
      int *ri; double *rd;
      const int k = fpfh_search(kd, pts + 3 * i, search_type, knn, radius, idx, d2, &ri, &rd);
      const int *use = ri ? ri : idx;
      const double *used2 = rd ? rd : d2;
      if (k > 1) {
        double sum[3] = {0.0, 0.0, 0.0};
        double *o = feature + 33 * i;
        for (int j = 1; j < k; j++) {
          const double dist = used2[j];
          if (dist == 0.0) continue;
          for (int b = 0; b < 33; b++) {
            const double val = spfh[33 * (size_t)use[j] + b] / dist;
            sum[b / 11] += val;
            o[b] += val;
          }
        }
        for (int b = 0; b < 3; b++)
          if (sum[b] != 0.0) sum[b] = 100.0 / sum[b];
        for (int b = 0; b < 33; b++) {
          o[b] *= sum[b / 11];
          o[b] += spfh[33 * i + b];
        }
      }
      free(ri); free(rd);
    }
    free(idx);
    free(d2);
  }
  if (spfh_out) memcpy(spfh_out, spfh, sizeof(double) * 33 * (size_t)n);
  free(spfh);
  orc_kdtree_free(kd);
}

/* EvaluateRegistration, Registration.cpp:118-127 */
ORC_EXPORT int orc_evaluate_registration(const double *src, int64_t ns, const double *tgt, int64_t nt, double max_dist,
                                         const double *T, int use_brute, int32_t *corr_out, int64_t *n_corr, double *fitness,
                                         double *rmse) {
  orc_kdtree *kd = use_brute ? NULL : orc_kdtree_build(tgt, nt);
  double *pcd = (double *)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
  memcpy(pcd, src, sizeof(double) * 3 * (size_t)ns);
  if (!orc_is_identity(T)) orc_transform_points(T, pcd, ns);
  double sd2;
  orc_correspondences(pcd, ns, tgt, nt, kd, max_dist, use_brute, corr_out, n_corr, &sd2, fitness, rmse);
  free(pcd);
  orc_kdtree_free(kd);
  return 0;
}

/* GetInformationMatrixFromPointClouds, Registration.cpp:303-355 (row "next"). */
ORC_EXPORT void orc_information_matrix(const double *src, int64_t ns, const double *tgt, int64_t nt, double max_dist,
                                       const double *T, int use_brute, double *GTG /* 36 row-major */) {
  int32_t *corr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ns > 0 ? ns : 1));
  int64_t nc;
  double fit, rmse;
  orc_evaluate_registration(src, ns, tgt, nt, max_dist, T, use_brute, corr, &nc, &fit, &rmse);
  memset(GTG, 0, 36 * sizeof(double));
  for (int64_t i = 0; i < ns; i++) {
    if (corr[i] < 0) continue;
    const double *p = tgt + 3 * (size_t)corr[i];
    double x = p[0], y = p[1], z = p[2];
    double G[3][6] = {{0, z, -y, 1, 0, 0}, {-z, 0, x, 0, 1, 0}, {y, -x, 0, 0, 0, 1}};
    for (int r = 0; r < 3; r++)
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) GTG[a * 6 + b] += G[r][a] * G[r][b];
  }
  free(corr);
}

/* ------------------------------------------------------------------------- */
/* Bounds: Geometry3D::ComputeMinBound/MaxBound, geometry/Geometry3D.cpp:39-53 */
/* ------------------------------------------------------------------------- */
ORC_EXPORT void orc_bounds(const double *pts, int64_t n, double *mn, double *mx) {
  if (n == 0) { for (int d = 0; d < 3; d++) mn[d] = mx[d] = 0.0; return; }
  for (int d = 0; d < 3; d++) mn[d] = mx[d] = pts[d];
  for (int64_t i = 1; i < n; i++)
    for (int d = 0; d < 3; d++) {
      double v = pts[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
}

/* ------------------------------------------------------------------------- */
/* PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306 (+ AccumulatedPoint */
/* :185-217).  Output ordered by voxel index (z, then y, then x ascending); the   */
/* reference's order is libstdc++ unordered_map bucket order (unspecified), so    */
/* parity is as a set keyed by out_vox.  Per-voxel sums run in input order.       */
/* Returns number of voxels M, -1 if voxel_size<=0 (:275-277), -2 if too small    */
/* (:281-283).  Output arrays must hold n entries.                                */
/* ------------------------------------------------------------------------- */
typedef struct { int32_t v[3]; int64_t i; } vkey;
static int vkey_cmp(const void *a, const void *b) {
  const vkey *A = (const vkey *)a, *B = (const vkey *)b;
  if (A->v[2] != B->v[2]) return A->v[2] < B->v[2] ? -1 : 1;
  if (A->v[1] != B->v[1]) return A->v[1] < B->v[1] ? -1 : 1;
  if (A->v[0] != B->v[0]) return A->v[0] < B->v[0] ? -1 : 1;
  return A->i < B->i ? -1 : (A->i > B->i ? 1 : 0);
}
/* The same order as qsort(vkey_cmp) -- (z, y, x, input index) ascending -- by a stable LSD radix sort of the packed voxel
 * index, for the full-size parity tests (100 M points: minutes with qsort, seconds here).  Only an ordering device: the
 * keys and the sums that follow are untouched.  Returns 0 (nothing done) when the packed index would not fit 64 bits or
 * any index is negative (NaN / out-of-range input), and the caller falls back to qsort.  orc_voxel_force_qsort lets the
 * tests compare the two orders. */
static int g_voxel_force_qsort = 0;
ORC_EXPORT void orc_voxel_force_qsort(int on) { g_voxel_force_qsort = on; }
static int orc_voxel_order_radix(vkey *k, int64_t n) {
  if (g_voxel_force_qsort || n < 2) return 0;
  int32_t mx[3] = {0, 0, 0};
  for (int64_t i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) {
      if (k[i].v[d] < 0) return 0;
      if (k[i].v[d] > mx[d]) mx[d] = k[i].v[d];
    }
  int bits[3];
  for (int d = 0; d < 3; d++) {
    bits[d] = 1;
    while (bits[d] < 31 && ((int64_t)1 << bits[d]) <= (int64_t)mx[d]) bits[d]++;
  }
  const int total = bits[0] + bits[1] + bits[2];
  if (total > 64) return 0;
  typedef struct { uint64_t key; int64_t i; } rk;
  rk *a = (rk *)malloc(sizeof(rk) * (size_t)n), *b = (rk *)malloc(sizeof(rk) * (size_t)n);
  if (!a || !b) { free(a); free(b); return 0; }
  for (int64_t i = 0; i < n; i++) {
    a[i].key = ((uint64_t)k[i].v[2] << (bits[0] + bits[1])) | ((uint64_t)k[i].v[1] << bits[0]) | (uint64_t)k[i].v[0];
    a[i].i = i;  /* input order: the sort is stable, so equal keys keep ascending i */
  }
  for (int shift = 0; shift < total; shift += 11) {
    size_t cnt[2049];
    memset(cnt, 0, sizeof cnt);
    for (int64_t i = 0; i < n; i++) cnt[((a[i].key >> shift) & 2047u) + 1]++;
    for (int q = 0; q < 2048; q++) cnt[q + 1] += cnt[q];
    for (int64_t i = 0; i < n; i++) b[cnt[(a[i].key >> shift) & 2047u]++] = a[i];
    rk *t = a; a = b; b = t;
  }
  const uint64_t m0 = ((uint64_t)1 << bits[0]) - 1, m1 = ((uint64_t)1 << bits[1]) - 1;
  for (int64_t i = 0; i < n; i++) {
    k[i].v[0] = (int32_t)(a[i].key & m0);
    k[i].v[1] = (int32_t)((a[i].key >> bits[0]) & m1);
    k[i].v[2] = (int32_t)(a[i].key >> (bits[0] + bits[1]));
    k[i].i = a[i].i;
  }
  free(a); free(b);
  return 1;
}
ORC_EXPORT int64_t orc_voxel_down_sample(const double *pts, const double *nrm, const double *col, int64_t n, double voxel,
                                         double *out_pts, double *out_nrm, double *out_col, int32_t *out_vox) {
  if (voxel <= 0.0) return -1;
  double mn[3], mx[3];
  orc_bounds(pts, n, mn, mx);
  double vmin[3], vmax[3], ext = -INFINITY;
  for (int d = 0; d < 3; d++) {
    vmin[d] = mn[d] - voxel * 0.5;
    vmax[d] = mx[d] + voxel * 0.5;
    if (vmax[d] - vmin[d] > ext) ext = vmax[d] - vmin[d];
  }
  if (voxel * (double)INT32_MAX < ext) return -2;
  if (n == 0) return 0;
  vkey *k = (vkey *)malloc(sizeof(vkey) * (size_t)n);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < n; i++) {
    for (int d = 0; d < 3; d++) k[i].v[d] = (int32_t)floor((pts[3 * i + d] - vmin[d]) / voxel);
    k[i].i = i;
  }
  if (!orc_voxel_order_radix(k, n)) qsort(k, (size_t)n, sizeof(vkey), vkey_cmp);
  /* voxel boundaries in the sorted order, then one voxel at a time (each voxel's sums run in input order whatever thread
   * takes it, so the result does not depend on the number of threads) */
  int64_t *start = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
  int64_t m = 0;
  for (int64_t a = 0; a < n; a++)
    if (a == 0 || k[a].v[0] != k[a - 1].v[0] || k[a].v[1] != k[a - 1].v[1] || k[a].v[2] != k[a - 1].v[2]) start[m++] = a;
  start[m] = n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t q = 0; q < m; q++) {
    const int64_t a = start[q], e = start[q + 1];
    double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
    for (int64_t b = a; b < e; b++) {
      int64_t i = k[b].i;
      for (int d = 0; d < 3; d++) sp[d] += pts[3 * i + d];
      if (nrm && !isnan(nrm[3 * i]) && !isnan(nrm[3 * i + 1]) && !isnan(nrm[3 * i + 2]))
        for (int d = 0; d < 3; d++) sn[d] += nrm[3 * i + d];
      if (col)
        for (int d = 0; d < 3; d++) sc[d] += col[3 * i + d];
    }
    double cnt = (double)(e - a);
    for (int d = 0; d < 3; d++) {
      out_pts[3 * q + d] = sp[d] / cnt;
      if (nrm) out_nrm[3 * q + d] = sn[d] / cnt;
      if (col) out_col[3 * q + d] = sc[d] / cnt;
      if (out_vox) out_vox[3 * q + d] = k[a].v[d];
    }
  }
  free(start);
  free(k);
  return m;
}

/* ------------------------------------------------------------------------- */
/* EstimateNormals: geometry/EstimateNormals.cpp:40-128 (ComputeEigenvector0/1), */
/* :130-213 (FastEigen3x3), :215-252 (ComputeNormal), :320-351 (the loop).       */
/* ------------------------------------------------------------------------- */
static void eigvec0(const double A[3][3], double eval0, double *out) { /* :40-68 */
  double row0[3] = {A[0][0] - eval0, A[0][1], A[0][2]};
  double row1[3] = {A[0][1], A[1][1] - eval0, A[1][2]};
  double row2[3] = {A[0][2], A[1][2], A[2][2] - eval0};
  double r0xr1[3], r0xr2[3], r1xr2[3];
  cross3(row0, row1, r0xr1);
  cross3(row0, row2, r0xr2);
  cross3(row1, row2, r1xr2);
  double d0 = r0xr1[0] * r0xr1[0] + r0xr1[1] * r0xr1[1] + r0xr1[2] * r0xr1[2];
  double d1 = r0xr2[0] * r0xr2[0] + r0xr2[1] * r0xr2[1] + r0xr2[2] * r0xr2[2];
  double d2 = r1xr2[0] * r1xr2[0] + r1xr2[1] * r1xr2[1] + r1xr2[2] * r1xr2[2];
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) { dmax = d1; imax = 1; }
  if (d2 > dmax) { imax = 2; }
  const double *v = imax == 0 ? r0xr1 : (imax == 1 ? r0xr2 : r1xr2);
  double s = sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
  out[0] = v[0] / s; out[1] = v[1] / s; out[2] = v[2] / s;
}

static void eigvec1(const double A[3][3], const double *evec0, double eval1, double *out) { /* :70-128 */
  double U[3], V[3];
  if (fabs(evec0[0]) > fabs(evec0[1])) {
    double inv = 1 / sqrt(evec0[0] * evec0[0] + evec0[2] * evec0[2]);
    U[0] = -evec0[2] * inv; U[1] = 0; U[2] = evec0[0] * inv;
  } else {
    double inv = 1 / sqrt(evec0[1] * evec0[1] + evec0[2] * evec0[2]);
    U[0] = 0; U[1] = evec0[2] * inv; U[2] = -evec0[1] * inv;
  }
  cross3(evec0, U, V);
  double AU[3] = {A[0][0] * U[0] + A[0][1] * U[1] + A[0][2] * U[2], A[0][1] * U[0] + A[1][1] * U[1] + A[1][2] * U[2],
                  A[0][2] * U[0] + A[1][2] * U[1] + A[2][2] * U[2]};
  double AV[3] = {A[0][0] * V[0] + A[0][1] * V[1] + A[0][2] * V[2], A[0][1] * V[0] + A[1][1] * V[1] + A[1][2] * V[2],
                  A[0][2] * V[0] + A[1][2] * V[1] + A[2][2] * V[2]};
  double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - eval1;
  double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
  double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - eval1;
  double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    double mx = a00 > a01 ? a00 : a01;
    if (mx > 0) {
      if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
      else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
      for (int d = 0; d < 3; d++) out[d] = m01 * U[d] - m00 * V[d];
    } else {
      for (int d = 0; d < 3; d++) out[d] = U[d];
    }
  } else {
    double mx = a11 > a01 ? a11 : a01;
    if (mx > 0) {
      if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
      else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
      for (int d = 0; d < 3; d++) out[d] = m11 * U[d] - m01 * V[d];
    } else {
      for (int d = 0; d < 3; d++) out[d] = U[d];
    }
  }
}

ORC_EXPORT void orc_fast_eigen3x3(const double *cov_rowmajor, double *normal) { /* :130-213 */
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = cov_rowmajor[i * 3 + j];
  double max_coeff = A[0][0];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (A[i][j] > max_coeff) max_coeff = A[i][j];
  if (max_coeff == 0) { normal[0] = normal[1] = normal[2] = 0; return; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] /= max_coeff;
  double norm = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
  if (norm > 0) {
    double eval[3], evec0[3], evec1[3], evec2[3];
    double q = (A[0][0] + A[1][1] + A[2][2]) / 3;
    double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
    double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
    double c00 = b11 * b22 - A[1][2] * A[1][2];
    double c01 = A[0][1] * b22 - A[1][2] * A[0][2];
    double c02 = A[0][1] * A[1][2] - b11 * A[0][2];
    double det = (b00 * c00 - A[0][1] * c01 + A[0][2] * c02) / (p * p * p);
    double half_det = det * 0.5;
    half_det = fmin(fmax(half_det, -1.0), 1.0);
    double angle = acos(half_det) / (double)3;
    const double two_thirds_pi = 2.09439510239319549;
    double beta2 = cos(angle) * 2;
    double beta0 = cos(angle + two_thirds_pi) * 2;
    double beta1 = -(beta0 + beta2);
    eval[0] = q + p * beta0;
    eval[1] = q + p * beta1;
    eval[2] = q + p * beta2;
    if (half_det >= 0) {
      eigvec0(A, eval[2], evec2);
      if (eval[2] < eval[0] && eval[2] < eval[1]) { memcpy(normal, evec2, 24); return; }
      eigvec1(A, evec2, eval[1], evec1);
      if (eval[1] < eval[0] && eval[1] < eval[2]) { memcpy(normal, evec1, 24); return; }
      cross3(evec1, evec2, evec0);
      memcpy(normal, evec0, 24);
    } else {
      eigvec0(A, eval[0], evec0);
      if (eval[0] < eval[1] && eval[0] < eval[2]) { memcpy(normal, evec0, 24); return; }
      eigvec1(A, evec0, eval[1], evec1);
      if (eval[1] < eval[0] && eval[1] < eval[2]) { memcpy(normal, evec1, 24); return; }
      cross3(evec0, evec1, evec2);
      memcpy(normal, evec2, 24);
    }
  } else {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A[i][j] *= max_coeff;
    if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) { normal[0] = 1; normal[1] = 0; normal[2] = 0; }
    else if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) { normal[0] = 0; normal[1] = 1; normal[2] = 0; }
    else { normal[0] = 0; normal[1] = 0; normal[2] = 1; }
  }
}

/* fast_normal_computation = false, EstimateNormals.cpp:247-251:
 *   Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> solver; solver.compute(covariance, ComputeEigenvectors);
 *   return solver.eigenvectors().col(0);
 * Eigen is not in the container: restated FROM MEMORY of Eigen 3.4.0, parity unpinned --
 *   SelfAdjointEigenSolver::compute            lower triangle, scaled by its largest |coefficient| (a division each)
 *   tridiagonalization_inplace_selector<,3>    the closed form for 3x3 (one Householder-like reflection, Q = [1 0 0; 0 m01 m02; 0 m02 -m01])
 *   computeFromTridiagonal_impl                deflation test (|e| / eps)^2 <= |d_i| + |d_i+1|, at most 30 * 3 sweeps,
 *                                              then selection sort of the eigenvalues (first minimum), columns swapped along
 *   tridiagonal_qr_step                        Wilkinson shift (numext::hypot = p * sqrt(1 + (q/p)^2)), Givens chase,
 *                                              Q.applyOnTheRight(k, k + 1, rot)
 *   JacobiRotation::makeGivens                 the real-scalar branch.
 * The sign of the returned column is whatever this sequence of operations leaves; the caller does not normalise it. */
static void orc_make_givens(double p, double q, double *c, double *s) {
  if (q == 0.0) { *c = p < 0.0 ? -1.0 : 1.0; *s = 0.0; }
  else if (p == 0.0) { *c = 0.0; *s = q < 0.0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) {
    double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    *c = 1.0 / u;
    *s = -t * *c;
  } else {
    double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    *s = -1.0 / u;
    *c = -t * *s;
  }
}
static double orc_eigen_hypot(double x, double y) { /* numext::hypot -> positive_real_hypot(|x|, |y|) */
  x = fabs(x); y = fabs(y);
  if (isinf(x) || isinf(y)) return INFINITY;
  if (isnan(x) || isnan(y)) return NAN;
  double p = x > y ? x : y;   /* numext::maxi(x, y) */
  if (p == 0.0) return 0.0;
  double qp = (y < x ? y : x) / p;
  return p * sqrt(1.0 + qp * qp);
}
/* eivals[3] ascending, eivecs column-major (eivecs[3 * c + r]); returns 0 when the iteration did not converge */
ORC_EXPORT int orc_self_adjoint_eigen3x3(const double *cov_rowmajor, double *eivals, double *eivecs) {
  double m00 = cov_rowmajor[0], m10 = cov_rowmajor[3], m11 = cov_rowmajor[4];
  double m20 = cov_rowmajor[6], m21 = cov_rowmajor[7], m22 = cov_rowmajor[8];
  /* cwiseAbs().maxCoeff() of the lower triangle (column-major visit; the order does not matter for a maximum) */
  double scale = fabs(m00);
  if (fabs(m10) > scale) scale = fabs(m10);
  if (fabs(m20) > scale) scale = fabs(m20);
  if (fabs(m11) > scale) scale = fabs(m11);
  if (fabs(m21) > scale) scale = fabs(m21);
  if (fabs(m22) > scale) scale = fabs(m22);
  if (scale == 0.0) scale = 1.0;
  m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
  double diag[3], sub[2], Q[9]; /* Q column-major */
  const double tol = DBL_MIN;
  diag[0] = m00;
  double v1norm2 = m20 * m20;
  if (v1norm2 <= tol) {
    diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
    for (int i = 0; i < 9; i++) Q[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    double beta = sqrt(m10 * m10 + v1norm2);
    double invBeta = 1.0 / beta;
    double m01 = m10 * invBeta, m02 = m20 * invBeta;
    double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    Q[0] = 1; Q[1] = 0; Q[2] = 0;      /* column 0 */
    Q[3] = 0; Q[4] = m01; Q[5] = m02;  /* column 1 */
    Q[6] = 0; Q[7] = m02; Q[8] = -m01; /* column 2 */
  }
  const int n = 3, max_iter = 30;
  int end = n - 1, start = 0, iter = 0;
  const double precision_inv = 1.0 / DBL_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; i++) {
      if (fabs(sub[i]) < DBL_MIN) sub[i] = 0.0;
      else {
        double scaled = precision_inv * sub[i];
        if (scaled * scaled <= fabs(diag[i]) + fabs(diag[i + 1])) sub[i] = 0.0;
      }
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > max_iter * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    { /* tridiagonal_qr_step(diag, sub, start, end, Q, n) */
      double td = (diag[end - 1] - diag[end]) * 0.5;
      double e = sub[end - 1];
      double mu = diag[end];
      if (td == 0.0) mu -= fabs(e);
      else if (e != 0.0) {
        double e2 = e * e;
        double h = orc_eigen_hypot(td, e);
        if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
        else mu -= e2 / (td + (td > 0.0 ? h : -h));
      }
      double x = diag[start] - mu;
      double z = sub[start];
      for (int k = start; k < end && z != 0.0; k++) {
        double c, s;
        orc_make_givens(x, z, &c, &s);
        double sdk = s * diag[k] + c * sub[k];
        double dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1) {
          z = -s * sub[k + 1];
          sub[k + 1] = c * sub[k + 1];
        }
        if (!(c == 1.0 && s == 0.0)) /* apply_rotation_in_the_plane returns early on the identity */
          for (int r = 0; r < 3; r++) {
            double xi = Q[3 * k + r], yi = Q[3 * (k + 1) + r];
            Q[3 * k + r] = c * xi - s * yi;
            Q[3 * (k + 1) + r] = s * xi + c * yi;
          }
      }
    }
  }
  int ok = iter <= max_iter * n;
  if (ok) {
    for (int i = 0; i < n - 1; i++) {
      int k = 0;
      for (int j = 1; j < n - i; j++) if (diag[i + j] < diag[i + k]) k = j;
      if (k > 0) {
        double t = diag[i]; diag[i] = diag[k + i]; diag[k + i] = t;
        for (int r = 0; r < 3; r++) { t = Q[3 * i + r]; Q[3 * i + r] = Q[3 * (k + i) + r]; Q[3 * (k + i) + r] = t; }
      }
    }
  }
  for (int i = 0; i < 3; i++) eivals[i] = diag[i] * scale;
  memcpy(eivecs, Q, sizeof Q);
  return ok;
}

/* ComputeNormal :215-252 (fast_normal_computation=true is the reference default, PointCloud.h:192).
 * Returns the covariance too. */
static void orc_compute_normal_impl(const double *pts, const int *indices, int k, double *normal, double *cov_out, int fast);
ORC_EXPORT void orc_compute_normal(const double *pts, const int *indices, int k, double *normal, double *cov_out) {
  orc_compute_normal_impl(pts, indices, k, normal, cov_out, 1);
}
ORC_EXPORT void orc_compute_normal_eigen(const double *pts, const int *indices, int k, double *normal, double *cov_out) {
  orc_compute_normal_impl(pts, indices, k, normal, cov_out, 0);
}
static void orc_compute_normal_impl(const double *pts, const int *indices, int k, double *normal, double *cov_out, int fast) {
  if (k == 0) { normal[0] = normal[1] = normal[2] = 0; return; }
  double c[9] = {0};
  for (int i = 0; i < k; i++) {
    const double *p = pts + 3 * (size_t)indices[i];
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    c[3] += p[0] * p[0]; c[4] += p[0] * p[1]; c[5] += p[0] * p[2];
    c[6] += p[1] * p[1]; c[7] += p[1] * p[2]; c[8] += p[2] * p[2];
  }
  for (int i = 0; i < 9; i++) c[i] /= (double)k;
  double cov[9];
  cov[0] = c[3] - c[0] * c[0];
  cov[4] = c[6] - c[1] * c[1];
  cov[8] = c[8] - c[2] * c[2];
  cov[1] = cov[3] = c[4] - c[0] * c[1];
  cov[2] = cov[6] = c[5] - c[0] * c[2];
  cov[5] = cov[7] = c[7] - c[1] * c[2];
  if (cov_out) memcpy(cov_out, cov, sizeof cov);
  if (fast) orc_fast_eigen3x3(cov, normal);
  else {
    double ev[3], Q[9];
    orc_self_adjoint_eigen3x3(cov, ev, Q);
    normal[0] = Q[0]; normal[1] = Q[1]; normal[2] = Q[2];
  }
}

/* PointCloud::EstimateNormals :320-351.  search_type: 0 = KNN, 1 = Radius,
 * 2 = Hybrid (KDTreeSearchParam.h:35-107).  normals is in/out when has_normals.
 * nn_count (nullable) receives the neighbour count per point. */
static void orc_estimate_normals_impl(const double *pts, int64_t n, int search_type, int knn, double radius,
                                      double *normals, int has_normals, int32_t *nn_count, int fast);
ORC_EXPORT void orc_estimate_normals(const double *pts, int64_t n, int search_type, int knn, double radius,
                                     double *normals, int has_normals, int32_t *nn_count) {
  orc_estimate_normals_impl(pts, n, search_type, knn, radius, normals, has_normals, nn_count, 1);
}
/* ... with fast_normal_computation = false */
ORC_EXPORT void orc_estimate_normals_eigen(const double *pts, int64_t n, int search_type, int knn, double radius,
                                           double *normals, int has_normals, int32_t *nn_count) {
  orc_estimate_normals_impl(pts, n, search_type, knn, radius, normals, has_normals, nn_count, 0);
}
static void orc_estimate_normals_impl(const double *pts, int64_t n, int search_type, int knn, double radius,
                                      double *normals, int has_normals, int32_t *nn_count, int fast) {
  orc_kdtree *kd = orc_kdtree_build(pts, n);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int cap = knn > 0 ? knn : 1;
    int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)cap);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < n; i++) {
      int k;
      int *ridx = NULL;
      double *rd2 = NULL;
      const int *use;
      if (search_type == 0) { k = orc_kdtree_search_knn(kd, pts + 3 * i, knn, idx, d2); use = idx; }
      else if (search_type == 2) { k = orc_kdtree_search_hybrid(kd, pts + 3 * i, radius, knn, idx, d2); use = idx; }
      else { k = orc_kdtree_search_radius(kd, pts + 3 * i, radius, &ridx, &rd2); use = ridx; }
      if (nn_count) nn_count[i] = k;
      double nv[3];
      if (k >= 3) {
        orc_compute_normal_impl(pts, use, k, nv, NULL, fast);
        if (sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]) == 0.0) {
          if (has_normals) { nv[0] = normals[3 * i]; nv[1] = normals[3 * i + 1]; nv[2] = normals[3 * i + 2]; }
          else { nv[0] = 0; nv[1] = 0; nv[2] = 1; }
        }
        if (has_normals && nv[0] * normals[3 * i] + nv[1] * normals[3 * i + 1] + nv[2] * normals[3 * i + 2] < 0.0) {
          nv[0] *= -1.0; nv[1] *= -1.0; nv[2] *= -1.0;
        }
      } else {
        nv[0] = 0; nv[1] = 0; nv[2] = 1;
      }
      normals[3 * i] = nv[0]; normals[3 * i + 1] = nv[1]; normals[3 * i + 2] = nv[2];
      free(ridx);
      free(rd2);
    }
    free(idx);
    free(d2);
  }
  orc_kdtree_free(kd);
}

/* Neighbour lists for cross-checks (k-best by (d2, idx)). */
ORC_EXPORT int orc_neighbors(const double *pts, int64_t n, const double *q, int search_type, int knn, double radius,
                             int use_brute, int *idx, double *d2, int cap_out) {
  if (use_brute) {
    double thr = search_type == 0 ? INFINITY : orc_radius_threshold(radius);
    int cap = search_type == 1 ? cap_out : knn;
    return orc_brute_search(pts, n, q, thr, cap, idx, d2);
  }
  orc_kdtree *kd = orc_kdtree_build(pts, n);
  int k;
  if (search_type == 0) k = orc_kdtree_search_knn(kd, q, knn, idx, d2);
  else if (search_type == 2) k = orc_kdtree_search_hybrid(kd, q, radius, knn, idx, d2);
  else {
    int *ri; double *rd;
    k = orc_kdtree_search_radius(kd, q, radius, &ri, &rd);
    int m = k < cap_out ? k : cap_out;
    memcpy(idx, ri, sizeof(int) * (size_t)m);
    memcpy(d2, rd, sizeof(double) * (size_t)m);
    free(ri); free(rd);
  }
  orc_kdtree_free(kd);
  return k;
}

/* ---- RegistrationRANSACBasedOnFeatureMatching pieces (Registration.cpp:199-301) ------------------------------------
 * Feature-space nearest neighbour, Registration.cpp:223,237: KDTreeFlann over the target Feature, SearchKNN(column, 1)
 * (KDTreeFlann.cpp:93-110).  The KD-tree search is exact, so the result is the argmin of flann::L2<double> -- restated
 * here as an exhaustive scan with L2's own summation order (groups of four, then the tail) and ties to the
 * lowest index (the north-star rule; FLANN keeps the first visited). */
static double orc_flann_l2(const double *a, const double *b, int dim) {
  double result = 0.0;
  int k = 0;
  for (; k + 3 < dim; k += 4) {
    const double d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  for (; k < dim; k++) {
    const double d = a[k] - b[k];
    result += d * d;
  }
  return result;
}

ORC_EXPORT void orc_feature_nearest(const double *sf, int64_t n_src, const double *tf, int64_t n_tgt, int dim, const int32_t *qids,
                                    int64_t n_q, int32_t *out_idx, double *out_d2) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t q = 0; q < n_q; q++) {
    const int64_t id = qids ? qids[q] : q;
    int32_t bi = -1;
    double bd = 0.0;
    if (id >= 0 && id < n_src)
      for (int64_t t = 0; t < n_tgt; t++) {
        const double d = orc_flann_l2(sf + (size_t)id * dim, tf + (size_t)t * dim, dim);
        if (bi < 0 ? d < INFINITY : d < bd) { bd = d; bi = (int32_t)t; }
      }
    out_idx[q] = bi;
    if (out_d2) out_d2[q] = bd;
  }
}

/* The checkers of CorrespondenceChecker.cpp:38-82 on one correspondence set of n pairs {source index, target index}.
 * T column-major.  A negative threshold leaves a checker out; the normal checker passes without normals (:67-72). */
ORC_EXPORT int orc_ransac_check(const double *src, const double *snrm, const double *tgt, const double *tnrm, const int32_t *c, int n,
                                const double *T, double edge_sim, double dist_thr, double normal_angle) {
  if (edge_sim >= 0.0)
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) {
        const double *si = src + 3 * (size_t)c[2 * i], *sj = src + 3 * (size_t)c[2 * j];
        const double *ti = tgt + 3 * (size_t)c[2 * i + 1], *tj = tgt + 3 * (size_t)c[2 * j + 1];
        const double sx = si[0] - sj[0], sy = si[1] - sj[1], sz = si[2] - sj[2];
        const double tx = ti[0] - tj[0], ty = ti[1] - tj[1], tz = ti[2] - tj[2];
        const double ds = sqrt((sx * sx + sy * sy) + sz * sz), dt = sqrt((tx * tx + ty * ty) + tz * tz);
        if (ds < dt * edge_sim || dt < ds * edge_sim) return 0;
      }
  if (dist_thr >= 0.0)
    for (int j = 0; j < n; j++) {
      const double *sp = src + 3 * (size_t)c[2 * j], *tp = tgt + 3 * (size_t)c[2 * j + 1];
      double d[3];
      for (int r = 0; r < 3; r++) d[r] = tp[r] - (((T[r] * sp[0] + T[4 + r] * sp[1]) + T[8 + r] * sp[2]) + T[12 + r]);
      if (sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > dist_thr) return 0;
    }
  if (normal_angle >= 0.0 && snrm && tnrm) {
    const double ct = cos(normal_angle);
    for (int j = 0; j < n; j++) {
      const double *sn = snrm + 3 * (size_t)c[2 * j], *tn = tnrm + 3 * (size_t)c[2 * j + 1];
      double v[3];
      for (int r = 0; r < 3; r++) v[r] = (T[r] * sn[0] + T[4 + r] * sn[1]) + T[8 + r] * sn[2];
      if ((tn[0] * v[0] + tn[1] * v[1]) + tn[2] * v[2] < ct) return 0;
    }
  }
  return 1;
}

/* EvaluateRANSACBasedOnCorrespondence, Registration.cpp:93-116: `src` is moved by T (PointCloud::Transform,
 * Geometry3D.cpp:77-82), every listed pair with squared distance < max_dist^2 is kept.  Returns the number kept; *err2 =
 * sum of their squared distances (serial, in list order); mask[k] = 1 for kept pairs (nullable). */
ORC_EXPORT int64_t orc_ransac_corr_evaluate(const double *src, const double *tgt, const int32_t *c, int64_t n_c, const double *T,
                                            double max_dist, double *err2, int32_t *mask) {
  const double max_d2 = max_dist * max_dist;
  const int affine = (T[3] == 0.0 && T[7] == 0.0 && T[11] == 0.0 && T[15] == 1.0);
  int64_t good = 0;
  double e = 0.0;
  for (int64_t k = 0; k < n_c; k++) {
    const double *s = src + 3 * (size_t)c[2 * k], *t = tgt + 3 * (size_t)c[2 * k + 1];
    double v[4];
    for (int r = 0; r < 4; r++) v[r] = ((T[r] * s[0] + T[4 + r] * s[1]) + T[8 + r] * s[2]) + T[12 + r];
    if (!affine) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
    const double dx = v[0] - t[0], dy = v[1] - t[1], dz = v[2] - t[2];
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    const int in = d2 < max_d2;
    if (in) { good++; e += d2; }
    if (mask) mask[k] = in;
  }
  *err2 = e;
  return good;
}

// open3d_amd_registration.h -- header-only C++ adapter with the reference's own signatures over the C ABI of
// libo3d_amd.so (include/o3d_icp.h).
//
// Mirrors, name for name and default for default:
//   open3d::pipelines::registration::RegistrationICP / EvaluateRegistration   (pipelines/registration/Registration.h:136-152)
//   ICPConvergenceCriteria, RegistrationResult                                  (Registration.h:54-76,107-126)
//   TransformationEstimation{,PointToPoint,PointToPlane}, CorrespondenceSet     (TransformationEstimation.h:43-130)
//   geometry::PointCloud::VoxelDownSample / EstimateNormals, KDTreeSearchParam* (geometry/PointCloud.h:130,192;
//                                                                                 geometry/KDTreeSearchParam.h:35-107)
// Error behaviour follows utility::LogError (utility/Console.h:82-87): a std::runtime_error carrying the reference's
// "[Open3D ERROR] ..." text.
//
// Inside the reference tree, define O3D_AMD_USE_OPEN3D_TYPES before including this header: the adapter then works on
// open3d::geometry::PointCloud and Eigen types directly.  Stand-alone (no Eigen in the image) it
// uses the layout-compatible value types below: Vector3d = 3 contiguous doubles, Matrix4d = 16 doubles column-major,
// Vector2i = 2 ints -- the storage Eigen gives the reference's std::vector<Eigen::Vector3d> (geometry/PointCloud.h:330-334).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "o3d_icp.h"

#ifdef O3D_AMD_USE_OPEN3D_TYPES
#ifdef O3D_AMD_OPEN3D_TYPES_HEADER  // (tests/cpp/compile_check.cpp: a types-only stub, for images without Eigen / Open3D)
#include O3D_AMD_OPEN3D_TYPES_HEADER
#else
#include "open3d/geometry/KDTreeSearchParam.h"
#include "open3d/geometry/PointCloud.h"
#include "open3d/pipelines/registration/Registration.h"
#include "open3d/pipelines/registration/TransformationEstimation.h"
#endif
#endif

namespace open3d_amd {

// ---- context handling ---------------------------------------------------------------------------------------------
class Context {
 public:
  explicit Context(int device = 0) {
    const int st = o3d_ctx_create(device, &ctx_);
    if (st != O3D_OK) throw std::runtime_error(std::string("open3d_amd: ") + o3d_last_error(nullptr));
  }
  // One context over several GPUs of the node (o3d_ctx_create_multi): RegistrationICP shards its source over them.
  explicit Context(const std::vector<int> &devices) {
    const int st = o3d_ctx_create_multi((int)devices.size(), devices.data(), &ctx_);
    if (st != O3D_OK) throw std::runtime_error(std::string("open3d_amd: ") + o3d_last_error(nullptr));
  }
  ~Context() { o3d_ctx_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  o3d_ctx *get() const { return ctx_; }
  // One context per host thread; created on first use.  O3D_AMD_DEVICES="0,1,2,3" in the environment makes it span those
  // GPUs, so that an unchanged caller of RegistrationICP(source, target, ...) uses all of them.
  static Context &Default(int device = 0) {
    thread_local std::unique_ptr<Context> c;
    if (!c) {
      std::vector<int> ids;
      if (const char *e = std::getenv("O3D_AMD_DEVICES")) {
        int v = 0;
        bool have = false;
        for (const char *p = e;; p++) {
          if (*p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); have = true; }
          else { if (have) ids.push_back(v); v = 0; have = false; if (!*p) break; }
        }
      }
      if (ids.size() > 1) c.reset(new Context(ids));
      else c.reset(new Context(ids.empty() ? device : ids[0]));
    }
    return *c;
  }

 private:
  o3d_ctx *ctx_ = nullptr;
};

inline void Check(o3d_ctx *ctx, int status) {
  if (status == O3D_OK) return;
  const char *msg = o3d_last_error(ctx);
  throw std::runtime_error((msg && *msg) ? msg : o3d_status_string(status));
}

#ifndef O3D_AMD_USE_OPEN3D_TYPES
// ---- layout-compatible stand-ins ------------------------------------------------------------------------------------
struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Vector2i {
  int v[2];
  Vector2i() : v{0, 0} {}
  Vector2i(int a, int b) : v{a, b} {}
  int operator[](int i) const { return v[i]; }
  int operator()(int i) const { return v[i]; }
};
struct Matrix4d {  // column-major, like Eigen::Matrix4d
  double m[16];
  static Matrix4d Identity() {
    Matrix4d r;
    for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return r;
  }
  double &operator()(int r, int c) { return m[c * 4 + r]; }
  double operator()(int r, int c) const { return m[c * 4 + r]; }
  const double *data() const { return m; }
  double *data() { return m; }
};
struct Matrix6d {  // column-major, like Eigen::Matrix6d (utility/Eigen.h:38)
  double m[36];
  double &operator()(int r, int c) { return m[c * 6 + r]; }
  double operator()(int r, int c) const { return m[c * 6 + r]; }
  const double *data() const { return m; }
  double *data() { return m; }
};
static_assert(sizeof(Vector3d) == 24 && sizeof(Vector2i) == 8 && sizeof(Matrix4d) == 128 && sizeof(Matrix6d) == 288,
              "layout must match Eigen's");

namespace geometry {

class KDTreeSearchParam {  // geometry/KDTreeSearchParam.h:35-57
 public:
  enum class SearchType { Knn = 0, Radius = 1, Hybrid = 2 };
  virtual ~KDTreeSearchParam() {}
  SearchType GetSearchType() const { return search_type_; }

 protected:
  explicit KDTreeSearchParam(SearchType t) : search_type_(t) {}

 private:
  SearchType search_type_;
};
class KDTreeSearchParamKNN : public KDTreeSearchParam {  // :62-75
 public:
  explicit KDTreeSearchParamKNN(int knn = 30) : KDTreeSearchParam(SearchType::Knn), knn_(knn) {}
  int knn_;
};
class KDTreeSearchParamRadius : public KDTreeSearchParam {  // :77-90
 public:
  explicit KDTreeSearchParamRadius(double radius) : KDTreeSearchParam(SearchType::Radius), radius_(radius) {}
  double radius_;
};
class KDTreeSearchParamHybrid : public KDTreeSearchParam {  // :92-107
 public:
  KDTreeSearchParamHybrid(double radius, int max_nn) : KDTreeSearchParam(SearchType::Hybrid), radius_(radius), max_nn_(max_nn) {}
  double radius_;
  int max_nn_;
};

class PointCloud {  // the members of geometry/PointCloud.h:330-334 that the ICP path touches
 public:
  std::vector<Vector3d> points_, normals_, colors_;
  bool HasPoints() const { return !points_.empty(); }
  bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }
  bool HasColors() const { return !points_.empty() && colors_.size() == points_.size(); }
  std::shared_ptr<PointCloud> VoxelDownSample(double voxel_size) const;
  void EstimateNormals(const KDTreeSearchParam &search_param = KDTreeSearchParamKNN(), bool fast_normal_computation = true);
};

}  // namespace geometry
namespace o3d_types {
using geometry::KDTreeSearchParam;
using geometry::KDTreeSearchParamHybrid;
using geometry::KDTreeSearchParamKNN;
using geometry::KDTreeSearchParamRadius;
using geometry::PointCloud;
using Mat4 = Matrix4d;
using Mat6 = Matrix6d;
using Vec2i = Vector2i;
using Vec3 = Vector3d;
}  // namespace o3d_types
#else
namespace o3d_types {
using open3d::geometry::KDTreeSearchParam;
using open3d::geometry::KDTreeSearchParamHybrid;
using open3d::geometry::KDTreeSearchParamKNN;
using open3d::geometry::KDTreeSearchParamRadius;
using open3d::geometry::PointCloud;
using Mat4 = Eigen::Matrix4d;
using Mat6 = Eigen::Matrix6d;
using Vec2i = Eigen::Vector2i;
using Vec3 = Eigen::Vector3d;
}  // namespace o3d_types
#endif

// ---- preprocessing --------------------------------------------------------------------------------------------------
// PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306 (output ordered by voxel index; the reference's order is
// unordered_map bucket order, i.e. unspecified).
inline std::shared_ptr<o3d_types::PointCloud> VoxelDownSample(const o3d_types::PointCloud &cloud, double voxel_size,
                                                               Context &c = Context::Default()) {
  auto out = std::make_shared<o3d_types::PointCloud>();
  const int64_t n = (int64_t)cloud.points_.size();
  const bool hn = cloud.HasNormals(), hc = cloud.HasColors();
  out->points_.resize(n);
  if (hn) out->normals_.resize(n);
  if (hc) out->colors_.resize(n);
  int64_t m = 0;
  Check(c.get(), o3d_voxel_down_sample(c.get(), (const double *)cloud.points_.data(), hn ? (const double *)cloud.normals_.data() : nullptr,
                                       hc ? (const double *)cloud.colors_.data() : nullptr, n, voxel_size, O3D_MEM_HOST,
                                       (double *)out->points_.data(), hn ? (double *)out->normals_.data() : nullptr,
                                       hc ? (double *)out->colors_.data() : nullptr, nullptr, &m));
  out->points_.resize(m);
  if (hn) out->normals_.resize(m);
  if (hc) out->colors_.resize(m);
  return out;
}

// PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351
inline void EstimateNormals(o3d_types::PointCloud &cloud, const o3d_types::KDTreeSearchParam &param = o3d_types::KDTreeSearchParamKNN(),
                            bool fast_normal_computation = true, Context &c = Context::Default()) {
  Check(c.get(), o3d_set_fast_normal_computation(c.get(), fast_normal_computation ? 1 : 0));  // :245-251
  const bool has = cloud.HasNormals();
  if (!has) cloud.normals_.resize(cloud.points_.size());
  int type = (int)param.GetSearchType(), knn = 0;
  double radius = 0.0;
  if (type == O3D_SEARCH_KNN) knn = static_cast<const o3d_types::KDTreeSearchParamKNN &>(param).knn_;
  else if (type == O3D_SEARCH_RADIUS) radius = static_cast<const o3d_types::KDTreeSearchParamRadius &>(param).radius_;
  else {
    radius = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).radius_;
    knn = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).max_nn_;
  }
  Check(c.get(), o3d_estimate_normals(c.get(), (const double *)cloud.points_.data(), (int64_t)cloud.points_.size(), type, knn, radius,
                                      O3D_MEM_HOST, (double *)cloud.normals_.data(), has ? 1 : 0));
}

// The other KDTreeFlann consumers on point clouds, geometry/PointCloud.cpp:112-132, 394-468, 521-544.
inline std::vector<double> ComputePointCloudDistance(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                                     Context &c = Context::Default()) {
  std::vector<double> d(source.points_.size());
  Check(c.get(), o3d_compute_point_cloud_distance(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                                  (const double *)target.points_.data(), (int64_t)target.points_.size(), O3D_MEM_HOST,
                                                  d.data()));
  return d;
}
inline std::vector<double> ComputeNearestNeighborDistance(const o3d_types::PointCloud &cloud, Context &c = Context::Default()) {
  std::vector<double> d(cloud.points_.size(), 0.0);
  if (cloud.points_.size() < 2) return d;  // :522-524
  Check(c.get(), o3d_compute_nearest_neighbor_distance(c.get(), (const double *)cloud.points_.data(), (int64_t)cloud.points_.size(),
                                                       O3D_MEM_HOST, d.data()));
  return d;
}
// indices kept by RemoveRadiusOutliers (:394-419); the caller's SelectByIndex builds the cloud
inline std::vector<size_t> RadiusOutlierInliers(const o3d_types::PointCloud &cloud, size_t nb_points, double search_radius,
                                                Context &c = Context::Default()) {
  if (nb_points < 1 || search_radius <= 0)
    throw std::runtime_error("[Open3D ERROR] [RemoveRadiusOutliers] Illegal input parameters,number of points and radius must be positive");
  std::vector<int32_t> cnt(cloud.points_.size());
  Check(c.get(), o3d_radius_neighbor_count(c.get(), (const double *)cloud.points_.data(), (int64_t)cloud.points_.size(), search_radius,
                                           O3D_MEM_HOST, cnt.data()));
  std::vector<size_t> indices;
  for (size_t i = 0; i < cnt.size(); i++)
    if ((size_t)cnt[i] > nb_points) indices.push_back(i);
  return indices;
}
// indices kept by RemoveStatisticalOutliers (:421-468): k-nearest mean distances from the device, then the reference's
// serial statistics (:453-466)
inline std::vector<size_t> StatisticalOutlierInliers(const o3d_types::PointCloud &cloud, size_t nb_neighbors, double std_ratio,
                                                     Context &c = Context::Default()) {
  if (nb_neighbors < 1 || std_ratio <= 0)
    throw std::runtime_error("[Open3D ERROR] [RemoveStatisticalOutliers] Illegal input parameters, number of neighbors and standard deviation ratio must be positive");
  std::vector<size_t> indices;
  if (cloud.points_.empty()) return indices;
  std::vector<double> avg(cloud.points_.size());
  Check(c.get(), o3d_knn_mean_distance(c.get(), (const double *)cloud.points_.data(), (int64_t)cloud.points_.size(), (int)nb_neighbors,
                                       O3D_MEM_HOST, avg.data()));
  size_t valid = 0;
  double cloud_mean = 0.0;
  for (double a : avg) { if (a >= 0) valid++; if (a > 0) cloud_mean += a; }
  if (valid == 0) return indices;
  cloud_mean /= (double)valid;
  double sq_sum = 0.0;
  for (double a : avg) if (a > 0) sq_sum += (a - cloud_mean) * (a - cloud_mean);
  const double thr = cloud_mean + std_ratio * std::sqrt(sq_sum / (double)(valid - 1));
  for (size_t i = 0; i < avg.size(); i++)
    if (avg[i] > 0 && avg[i] < thr) indices.push_back(i);
  return indices;
}

#ifndef O3D_AMD_USE_OPEN3D_TYPES
inline std::shared_ptr<geometry::PointCloud> geometry::PointCloud::VoxelDownSample(double voxel_size) const {
  return open3d_amd::VoxelDownSample(*this, voxel_size);
}
inline void geometry::PointCloud::EstimateNormals(const KDTreeSearchParam &p, bool fast) { open3d_amd::EstimateNormals(*this, p, fast); }
#endif

// ---- registration -----------------------------------------------------------------------------------------------------
namespace pipelines {
namespace registration {

#ifndef O3D_AMD_USE_OPEN3D_TYPES
typedef std::vector<Vector2i> CorrespondenceSet;  // TransformationEstimation.h:43
#else
using open3d::pipelines::registration::CorrespondenceSet;
#endif

// TransformationEstimationPointToPoint / PointToPlane::ComputeTransformation and ComputeRMSE on the device
// (TransformationEstimation.cpp:38-98).  estimation: O3D_EST_POINT_TO_POINT / O3D_EST_POINT_TO_PLANE.  Inside the reference
// tree the four member functions become one-line calls of these two.
inline o3d_types::Mat4 ComputeTransformationOnDevice(int estimation, bool with_scaling, const o3d_types::PointCloud &source,
                                                     const o3d_types::PointCloud &target, const CorrespondenceSet &corres,
                                                     Context &c = Context::Default()) {
  o3d_types::Mat4 T;
  Check(c.get(), o3d_compute_transformation(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                            (const double *)target.points_.data(),
                                            target.HasNormals() ? (const double *)target.normals_.data() : nullptr,
                                            (int64_t)target.points_.size(), (const int32_t *)corres.data(), (int64_t)corres.size(),
                                            estimation, with_scaling ? 1 : 0, O3D_MEM_HOST, T.data()));
  return T;
}
inline double ComputeRMSEOnDevice(int estimation, const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                  const CorrespondenceSet &corres, Context &c = Context::Default()) {
  double rmse = 0.0;
  Check(c.get(), o3d_compute_rmse(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                  (const double *)target.points_.data(),
                                  target.HasNormals() ? (const double *)target.normals_.data() : nullptr,
                                  (int64_t)target.points_.size(), (const int32_t *)corres.data(), (int64_t)corres.size(), estimation,
                                  O3D_MEM_HOST, &rmse));
  return rmse;
}

#ifndef O3D_AMD_USE_OPEN3D_TYPES
enum class TransformationEstimationType { Unspecified = 0, PointToPoint = 1, PointToPlane = 2, ColoredICP = 3 };  // :45-50

class TransformationEstimation {  // :57-80
 public:
  virtual ~TransformationEstimation() {}
  virtual TransformationEstimationType GetTransformationEstimationType() const = 0;
  virtual double ComputeRMSE(const geometry::PointCloud &source, const geometry::PointCloud &target,
                             const CorrespondenceSet &corres) const = 0;
  virtual Matrix4d ComputeTransformation(const geometry::PointCloud &source, const geometry::PointCloud &target,
                                         const CorrespondenceSet &corres) const = 0;
};
class TransformationEstimationPointToPoint : public TransformationEstimation {  // :85-107
 public:
  explicit TransformationEstimationPointToPoint(bool with_scaling = false) : with_scaling_(with_scaling) {}
  TransformationEstimationType GetTransformationEstimationType() const override { return TransformationEstimationType::PointToPoint; }
  double ComputeRMSE(const geometry::PointCloud &source, const geometry::PointCloud &target,
                     const CorrespondenceSet &corres) const override {
    return ComputeRMSEOnDevice(O3D_EST_POINT_TO_POINT, source, target, corres);
  }
  Matrix4d ComputeTransformation(const geometry::PointCloud &source, const geometry::PointCloud &target,
                                 const CorrespondenceSet &corres) const override {
    return ComputeTransformationOnDevice(O3D_EST_POINT_TO_POINT, with_scaling_, source, target, corres);
  }
  bool with_scaling_ = false;
};
class TransformationEstimationPointToPlane : public TransformationEstimation {  // :112-130
 public:
  TransformationEstimationType GetTransformationEstimationType() const override { return TransformationEstimationType::PointToPlane; }
  double ComputeRMSE(const geometry::PointCloud &source, const geometry::PointCloud &target,
                     const CorrespondenceSet &corres) const override {
    return ComputeRMSEOnDevice(O3D_EST_POINT_TO_PLANE, source, target, corres);
  }
  Matrix4d ComputeTransformation(const geometry::PointCloud &source, const geometry::PointCloud &target,
                                 const CorrespondenceSet &corres) const override {
    return ComputeTransformationOnDevice(O3D_EST_POINT_TO_PLANE, false, source, target, corres);
  }
};

class ICPConvergenceCriteria {  // Registration.h:54-76
 public:
  ICPConvergenceCriteria(double relative_fitness = 1e-6, double relative_rmse = 1e-6, int max_iteration = 30)
      : relative_fitness_(relative_fitness), relative_rmse_(relative_rmse), max_iteration_(max_iteration) {}
  double relative_fitness_, relative_rmse_;
  int max_iteration_;
};

class RegistrationResult {  // Registration.h:107-126
 public:
  explicit RegistrationResult(const Matrix4d &transformation = Matrix4d::Identity())
      : transformation_(transformation), inlier_rmse_(0.0), fitness_(0.0) {}
  Matrix4d transformation_;
  CorrespondenceSet correspondence_set_;
  double inlier_rmse_, fitness_;
};
#else
using open3d::pipelines::registration::ICPConvergenceCriteria;
using open3d::pipelines::registration::RegistrationResult;
using open3d::pipelines::registration::TransformationEstimation;
using open3d::pipelines::registration::TransformationEstimationPointToPlane;
using open3d::pipelines::registration::TransformationEstimationPointToPoint;
using open3d::pipelines::registration::TransformationEstimationType;
#endif

namespace detail {
inline void ExpandCorrespondences(const std::vector<int32_t> &corr, CorrespondenceSet &out) {
  out.clear();
  for (size_t i = 0; i < corr.size(); i++)  // ascending source index: the reference's serial order (Registration.cpp:60-68)
    if (corr[i] >= 0) out.push_back(o3d_types::Vec2i((int)i, corr[i]));
}

// Column-major 4x4 helpers on raw storage (the same code serves Eigen::Matrix4d and the stand-in).
inline void MatMul4(const double *A, const double *B, double *C) {
  double r[16];
  for (int c = 0; c < 4; c++)
    for (int i = 0; i < 4; i++) {
      double v = 0.0;
      for (int k = 0; k < 4; k++) v += A[k * 4 + i] * B[c * 4 + k];
      r[c * 4 + i] = v;
    }
  for (int i = 0; i < 16; i++) C[i] = r[i];
}
inline bool IsIdentity4(const double *T) {  // Eigen isIdentity(), dummy_precision 1e-12
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      const double v = T[c * 4 + r];
      if (r == c) {
        if (!(std::fabs(v - 1.0) <= 1e-12 * std::fmin(std::fabs(v), 1.0))) return false;
      } else if (!(std::fabs(v) <= 1e-12)) {
        return false;
      }
    }
  return true;
}
// PointCloud::Transform, geometry/Geometry3D.cpp:77-89 (points: homogeneous divide; normals: rotation part only)
inline void TransformCloud(o3d_types::PointCloud &pcd, const double *T) {
  double *p = (double *)pcd.points_.data();
  for (size_t i = 0; i < pcd.points_.size(); i++, p += 3) {
    const double x = p[0], y = p[1], z = p[2];
    const double w = T[3] * x + T[7] * y + T[11] * z + T[15];
    p[0] = (T[0] * x + T[4] * y + T[8] * z + T[12]) / w;
    p[1] = (T[1] * x + T[5] * y + T[9] * z + T[13]) / w;
    p[2] = (T[2] * x + T[6] * y + T[10] * z + T[14]) / w;
  }
  if (pcd.HasNormals()) {
    double *n = (double *)pcd.normals_.data();
    for (size_t i = 0; i < pcd.normals_.size(); i++, n += 3) {
      const double x = n[0], y = n[1], z = n[2];
      n[0] = T[0] * x + T[4] * y + T[8] * z;
      n[1] = T[1] * x + T[5] * y + T[9] * z;
      n[2] = T[2] * x + T[6] * y + T[10] * z;
    }
  }
}
}  // namespace detail

// True when the whole loop runs on the device.  Any other estimation (ColoredICP, user subclasses) still gets its
// correspondence search from the device: RegistrationICP below then calls the virtual ComputeTransformation on the host.
inline bool IsAccelerated(const TransformationEstimation &e) {
  const auto t = e.GetTransformationEstimationType();
  return t == TransformationEstimationType::PointToPoint || t == TransformationEstimationType::PointToPlane;
}

// EvaluateRegistration, Registration.cpp:118-127
inline RegistrationResult EvaluateRegistration(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                               double max_correspondence_distance,
                                               const o3d_types::Mat4 &transformation = o3d_types::Mat4::Identity(),
                                               Context &c = Context::Default()) {
  RegistrationResult result(transformation);
  std::vector<int32_t> corr(source.points_.size());
  int64_t nc = 0;
  Check(c.get(), o3d_evaluate_registration(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                           (const double *)target.points_.data(), (int64_t)target.points_.size(),
                                           max_correspondence_distance, transformation.data(), O3D_MEM_HOST, corr.data(), &nc,
                                           &result.fitness_, &result.inlier_rmse_));
  detail::ExpandCorrespondences(corr, result.correspondence_set_);
  return result;
}

// Registration.cpp:129-168 for an estimation whose ComputeTransformation stays on the host: LOOP-A (the correspondence
// search, :41-91) on the device through o3d_icp_pass / o3d_icp_fetch, the virtual call on `pcd` exactly as the reference
// makes it (:157), the update handed back with o3d_icp_set_update (:158-159), convergence test (:162-165) here.
inline RegistrationResult RegistrationICPHostEstimation(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                                        double max_correspondence_distance, const o3d_types::Mat4 &init,
                                                        const TransformationEstimation &estimation,
                                                        const ICPConvergenceCriteria &criteria, Context &c = Context::Default()) {
  if (max_correspondence_distance <= 0.0) throw std::runtime_error(o3d_status_string(O3D_ERR_INVALID_DISTANCE));
  if (estimation.GetTransformationEstimationType() == TransformationEstimationType::ColoredICP && !target.HasNormals())
    throw std::runtime_error(o3d_status_string(O3D_ERR_NEED_NORMALS));
  const int64_t ns = (int64_t)source.points_.size();
  Check(c.get(), o3d_icp_setup(c.get(), (const double *)source.points_.data(), ns, (const double *)target.points_.data(), nullptr,
                               (int64_t)target.points_.size(), max_correspondence_distance, init.data(), O3D_EST_POINT_TO_POINT, 0,
                               0.0, 0.0, 1 << 20, O3D_MEM_HOST, ns));
  o3d_types::PointCloud pcd = source;
  if (!detail::IsIdentity4(init.data())) detail::TransformCloud(pcd, init.data());
  RegistrationResult result(init);
  std::vector<int32_t> corr((size_t)ns);
  int64_t nc = 0;
  auto pass = [&](RegistrationResult &r) {
    Check(c.get(), o3d_icp_pass(c.get()));
    Check(c.get(), o3d_icp_fetch(c.get(), O3D_MEM_HOST, corr.data(), &nc, &r.fitness_, &r.inlier_rmse_));
    detail::ExpandCorrespondences(corr, r.correspondence_set_);
  };
  pass(result);
  o3d_types::Mat4 T = init;
  for (int i = 0; i < criteria.max_iteration_; i++) {
    const o3d_types::Mat4 update = estimation.ComputeTransformation(pcd, target, result.correspondence_set_);
    detail::MatMul4(update.data(), T.data(), T.data());
    detail::TransformCloud(pcd, update.data());
    Check(c.get(), o3d_icp_set_update(c.get(), update.data()));
    const double pf = result.fitness_, pr = result.inlier_rmse_;
    pass(result);
    result.transformation_ = T;
    if (std::fabs(pf - result.fitness_) < criteria.relative_fitness_ && std::fabs(pr - result.inlier_rmse_) < criteria.relative_rmse_)
      break;
  }
  return result;
}

// RegistrationICP, Registration.cpp:129-168
inline RegistrationResult RegistrationICP(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                          double max_correspondence_distance,
                                          const o3d_types::Mat4 &init = o3d_types::Mat4::Identity(),
                                          const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false),
                                          const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria(),
                                          Context &c = Context::Default()) {
  if (!IsAccelerated(estimation)) return RegistrationICPHostEstimation(source, target, max_correspondence_distance, init, estimation, criteria, c);
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  int with_scaling = 0;
  if (!p2l) with_scaling = static_cast<const TransformationEstimationPointToPoint &>(estimation).with_scaling_ ? 1 : 0;
  RegistrationResult result(init);
  std::vector<int32_t> corr(source.points_.size());
  int64_t nc = 0;
  int iters = 0;
  double T[16];
  Check(c.get(), o3d_registration_icp(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                      (const double *)target.points_.data(),
                                      target.HasNormals() ? (const double *)target.normals_.data() : nullptr,
                                      (int64_t)target.points_.size(), max_correspondence_distance, init.data(),
                                      p2l ? O3D_EST_POINT_TO_PLANE : O3D_EST_POINT_TO_POINT, with_scaling,
                                      criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_, O3D_MEM_HOST, T,
                                      corr.data(), &nc, &result.fitness_, &result.inlier_rmse_, &iters));
  for (int i = 0; i < 16; i++) result.transformation_.data()[i] = T[i];
  detail::ExpandCorrespondences(corr, result.correspondence_set_);
  return result;
}

// ---- clouds that stay in HBM across calls (o3d_cloud, include/o3d_icp.h) ------------------------------------------------
// The reference passes `const PointCloud &` from VoxelDownSample to EstimateNormals to RegistrationICP in one address space;
// through host pointers every call would cross PCIe again.  A DeviceCloud is uploaded once and down-sampled, given normals
// and registered where it lies; results equal those of the PointCloud overloads bit for bit.
class DeviceCloud {
 public:
  DeviceCloud() = default;
  explicit DeviceCloud(const o3d_types::PointCloud &cloud, Context &c = Context::Default()) : ctx_(&c) {
    Check(c.get(), o3d_cloud_upload(c.get(), (const double *)cloud.points_.data(),
                                    cloud.HasNormals() ? (const double *)cloud.normals_.data() : nullptr,
                                    cloud.HasColors() ? (const double *)cloud.colors_.data() : nullptr, (int64_t)cloud.points_.size(),
                                    O3D_MEM_HOST, &h_));
  }
  ~DeviceCloud() { o3d_cloud_free(h_); }
  DeviceCloud(const DeviceCloud &) = delete;
  DeviceCloud &operator=(const DeviceCloud &) = delete;
  DeviceCloud(DeviceCloud &&o) noexcept : h_(o.h_), ctx_(o.ctx_) { o.h_ = nullptr; }
  DeviceCloud &operator=(DeviceCloud &&o) noexcept {
    if (this != &o) { o3d_cloud_free(h_); h_ = o.h_; ctx_ = o.ctx_; o.h_ = nullptr; }
    return *this;
  }
  size_t size() const { return (size_t)o3d_cloud_size(h_); }
  bool HasNormals() const { return o3d_cloud_has_normals(h_) != 0; }
  o3d_cloud *get() const { return h_; }
  // PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306
  DeviceCloud VoxelDownSample(double voxel_size) const {
    DeviceCloud out;
    out.ctx_ = ctx_;
    Check(ctx_->get(), o3d_cloud_voxel_down_sample(ctx_->get(), h_, voxel_size, &out.h_));
    return out;
  }
  // PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351
  void EstimateNormals(const o3d_types::KDTreeSearchParam &param = o3d_types::KDTreeSearchParamKNN(), bool fast_normal_computation = true) {
    Check(ctx_->get(), o3d_set_fast_normal_computation(ctx_->get(), fast_normal_computation ? 1 : 0));
    int type = (int)param.GetSearchType(), knn = 0;
    double radius = 0.0;
    if (type == O3D_SEARCH_KNN) knn = static_cast<const o3d_types::KDTreeSearchParamKNN &>(param).knn_;
    else if (type == O3D_SEARCH_RADIUS) radius = static_cast<const o3d_types::KDTreeSearchParamRadius &>(param).radius_;
    else {
      radius = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).radius_;
      knn = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).max_nn_;
    }
    Check(ctx_->get(), o3d_cloud_estimate_normals(ctx_->get(), h_, type, knn, radius));
  }
  o3d_types::PointCloud Download() const {
    o3d_types::PointCloud out;
    const size_t n = size();
    out.points_.resize(n);
    if (HasNormals()) out.normals_.resize(n);
    if (o3d_cloud_has_colors(h_)) out.colors_.resize(n);
    Check(ctx_->get(), o3d_cloud_download(ctx_->get(), h_, (double *)out.points_.data(),
                                          out.normals_.empty() ? nullptr : (double *)out.normals_.data(),
                                          out.colors_.empty() ? nullptr : (double *)out.colors_.data()));
    return out;
  }
  Context &context() const { return *ctx_; }

 private:
  o3d_cloud *h_ = nullptr;
  Context *ctx_ = nullptr;
};

// RegistrationICP, Registration.cpp:129-168, on device-resident clouds (point-to-point / point-to-plane).
inline RegistrationResult RegistrationICP(const DeviceCloud &source, const DeviceCloud &target, double max_correspondence_distance,
                                          const o3d_types::Mat4 &init = o3d_types::Mat4::Identity(),
                                          const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false),
                                          const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria()) {
  if (!IsAccelerated(estimation)) throw std::runtime_error("open3d_amd: device-resident clouds take the point-to-point and point-to-plane estimations");
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  int with_scaling = 0;
  if (!p2l) with_scaling = static_cast<const TransformationEstimationPointToPoint &>(estimation).with_scaling_ ? 1 : 0;
  RegistrationResult result(init);
  std::vector<int32_t> corr(source.size());
  int64_t nc = 0;
  int iters = 0;
  double T[16];
  o3d_ctx *c = source.context().get();
  Check(c, o3d_cloud_registration_icp(c, source.get(), target.get(), max_correspondence_distance, init.data(),
                                      p2l ? O3D_EST_POINT_TO_PLANE : O3D_EST_POINT_TO_POINT, with_scaling, criteria.relative_fitness_,
                                      criteria.relative_rmse_, criteria.max_iteration_, T, corr.data(), &nc, &result.fitness_,
                                      &result.inlier_rmse_, &iters));
  for (int i = 0; i < 16; i++) result.transformation_.data()[i] = T[i];
  detail::ExpandCorrespondences(corr, result.correspondence_set_);
  return result;
}

// Coarse-to-fine registration (BASELINE configs 3 and 4; build-defined: the reference has no multi-scale function, this is the
// loop a caller writes over its three calls).  Per scale v: VoxelDownSample(v) of both clouds, EstimateNormals(Hybrid(2 v, 30))
// of the target for point-to-plane, RegistrationICP with max distance 2 v from the previous scale's transformation.  ONE upload
// per raw cloud; everything else happens in HBM.
inline std::vector<RegistrationResult> MultiScaleICP(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                                     const std::vector<double> &voxel_sizes,
                                                     const o3d_types::Mat4 &init = o3d_types::Mat4::Identity(),
                                                     const TransformationEstimation &estimation = TransformationEstimationPointToPlane(),
                                                     const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria(),
                                                     Context &c = Context::Default()) {
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  DeviceCloud d_src(source, c), d_tgt(target, c);
  std::vector<RegistrationResult> results;
  o3d_types::Mat4 T = init;
  for (double v : voxel_sizes) {
    DeviceCloud sv = d_src.VoxelDownSample(v), tv = d_tgt.VoxelDownSample(v);
    if (p2l) tv.EstimateNormals(o3d_types::KDTreeSearchParamHybrid(2.0 * v, 30));
    results.push_back(RegistrationICP(sv, tv, 2.0 * v, T, estimation, criteria));
    T = results.back().transformation_;
  }
  return results;
}

// RegistrationColoredICP, ColoredICP.cpp:205-211 (colour gradients :68-123 and the two-row estimation :125-177 on the device)
inline RegistrationResult RegistrationColoredICP(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                                 double max_distance, const o3d_types::Mat4 &init = o3d_types::Mat4::Identity(),
                                                 const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria(),
                                                 double lambda_geometric = 0.968, Context &c = Context::Default()) {
  RegistrationResult result(init);
  std::vector<int32_t> corr(source.points_.size());
  int64_t nc = 0;
  int iters = 0;
  double T[16];
  Check(c.get(), o3d_registration_colored_icp(
                     c.get(), (const double *)source.points_.data(), source.HasColors() ? (const double *)source.colors_.data() : nullptr,
                     (int64_t)source.points_.size(), (const double *)target.points_.data(),
                     target.HasNormals() ? (const double *)target.normals_.data() : nullptr,
                     target.HasColors() ? (const double *)target.colors_.data() : nullptr, (int64_t)target.points_.size(), max_distance,
                     init.data(), lambda_geometric, criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_,
                     O3D_MEM_HOST, T, corr.data(), &nc, &result.fitness_, &result.inlier_rmse_, &iters, nullptr));
  for (int i = 0; i < 16; i++) result.transformation_.data()[i] = T[i];
  detail::ExpandCorrespondences(corr, result.correspondence_set_);
  return result;
}

// ComputeFPFHFeature, pipelines/registration/Feature.cpp:113-159.  Returns the storage of Feature::data_ (33 x n,
// column-major: 33 consecutive doubles per point); inside the reference tree, map it with
// Eigen::Map<Eigen::MatrixXd>(v.data(), 33, n).
inline std::vector<double> ComputeFPFHFeatureData(const o3d_types::PointCloud &input,
                                                  const o3d_types::KDTreeSearchParam &param = o3d_types::KDTreeSearchParamKNN(),
                                                  Context &c = Context::Default()) {
  if (!input.HasNormals()) throw std::runtime_error("[Open3D ERROR] [ComputeFPFHFeature] Failed because input point cloud has no normal.");
  int type = (int)param.GetSearchType(), knn = 0;
  double radius = 0.0;
  if (type == O3D_SEARCH_KNN) knn = static_cast<const o3d_types::KDTreeSearchParamKNN &>(param).knn_;
  else if (type == O3D_SEARCH_RADIUS) radius = static_cast<const o3d_types::KDTreeSearchParamRadius &>(param).radius_;
  else {
    radius = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).radius_;
    knn = static_cast<const o3d_types::KDTreeSearchParamHybrid &>(param).max_nn_;
  }
  std::vector<double> data(33 * input.points_.size());
  Check(c.get(), o3d_compute_fpfh_feature(c.get(), (const double *)input.points_.data(), (const double *)input.normals_.data(),
                                          (int64_t)input.points_.size(), type, knn, radius, O3D_MEM_HOST, data.data()));
  return data;
}

// RANSACConvergenceCriteria, Registration.h:87-102
class RANSACConvergenceCriteria {
 public:
  RANSACConvergenceCriteria(int max_iteration = 1000, int max_validation = 1000)
      : max_iteration_(max_iteration), max_validation_(max_validation) {}
  int max_iteration_;
  int max_validation_;
};

// The three checkers of CorrespondenceChecker.h; their Check() bodies (CorrespondenceChecker.cpp:38-82) run on the device
// inside o3d_ransac_hypotheses, so the stand-ins only carry the thresholds.  A negative threshold = checker absent.
struct CorrespondenceCheckers {
  double edge_length_similarity_ = -1.0;    // CorrespondenceCheckerBasedOnEdgeLength::similarity_threshold_
  double distance_threshold_ = -1.0;        // CorrespondenceCheckerBasedOnDistance::distance_threshold_
  double normal_angle_threshold_ = -1.0;    // CorrespondenceCheckerBasedOnNormal::normal_angle_threshold_
};

// RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:199-301.  Features are the storage of Feature::data_
// (dim x n, column-major: source_feature.data_.data()).  seed < 0: std::random_device, as the reference.
inline RegistrationResult RegistrationRANSACBasedOnFeatureMatching(
    const o3d_types::PointCloud &source, const o3d_types::PointCloud &target, const double *source_feature,
    const double *target_feature, int feature_dim, double max_correspondence_distance,
    const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false), int ransac_n = 4,
    const CorrespondenceCheckers &checkers = CorrespondenceCheckers(),
    const RANSACConvergenceCriteria &criteria = RANSACConvergenceCriteria(), int64_t seed = -1, Context &c = Context::Default()) {
  if (!IsAccelerated(estimation))
    throw std::runtime_error("open3d_amd: RegistrationRANSACBasedOnFeatureMatching needs a point-to-point or point-to-plane estimation");
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  int with_scaling = 0;
  if (!p2l) with_scaling = static_cast<const TransformationEstimationPointToPoint &>(estimation).with_scaling_ ? 1 : 0;
  RegistrationResult result;
  std::vector<int32_t> corr(source.points_.size());
  int64_t nc = 0;
  double T[16];
  Check(c.get(), o3d_registration_ransac_feature_matching(
                     c.get(), (const double *)source.points_.data(), source.HasNormals() ? (const double *)source.normals_.data() : nullptr,
                     (int64_t)source.points_.size(), (const double *)target.points_.data(),
                     target.HasNormals() ? (const double *)target.normals_.data() : nullptr, (int64_t)target.points_.size(),
                     source_feature, target_feature, feature_dim, max_correspondence_distance,
                     p2l ? O3D_EST_POINT_TO_PLANE : O3D_EST_POINT_TO_POINT, with_scaling, ransac_n, checkers.edge_length_similarity_,
                     checkers.distance_threshold_, checkers.normal_angle_threshold_, criteria.max_iteration_, criteria.max_validation_,
                     seed, O3D_MEM_HOST, T, corr.data(), &nc, &result.fitness_, &result.inlier_rmse_, nullptr));
  for (int i = 0; i < 16; i++) result.transformation_.data()[i] = T[i];
  detail::ExpandCorrespondences(corr, result.correspondence_set_);
  return result;
}

// RegistrationRANSACBasedOnCorrespondence, Registration.cpp:170-197
inline RegistrationResult RegistrationRANSACBasedOnCorrespondence(
    const o3d_types::PointCloud &source, const o3d_types::PointCloud &target, const CorrespondenceSet &corres,
    double max_correspondence_distance, const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false),
    int ransac_n = 6, const RANSACConvergenceCriteria &criteria = RANSACConvergenceCriteria(), int64_t seed = -1,
    Context &c = Context::Default()) {
  static_assert(sizeof(corres[0]) == 2 * sizeof(int32_t), "CorrespondenceSet must be pairs of 32-bit ints");
  if (!IsAccelerated(estimation))
    throw std::runtime_error("open3d_amd: RegistrationRANSACBasedOnCorrespondence needs a point-to-point or point-to-plane estimation");
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  int with_scaling = 0;
  if (!p2l) with_scaling = static_cast<const TransformationEstimationPointToPoint &>(estimation).with_scaling_ ? 1 : 0;
  RegistrationResult result;
  std::vector<int32_t> inlier(corres.size());
  int64_t n_in = 0;
  double T[16];
  Check(c.get(), o3d_registration_ransac_correspondence(
                     c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                     (const double *)target.points_.data(), target.HasNormals() ? (const double *)target.normals_.data() : nullptr,
                     (int64_t)target.points_.size(), (const int32_t *)corres.data(), (int64_t)corres.size(),
                     max_correspondence_distance, p2l ? O3D_EST_POINT_TO_PLANE : O3D_EST_POINT_TO_POINT, with_scaling, ransac_n,
                     criteria.max_iteration_, criteria.max_validation_, seed, O3D_MEM_HOST, T, inlier.data(), &n_in,
                     &result.fitness_, &result.inlier_rmse_));
  for (int i = 0; i < 16; i++) result.transformation_.data()[i] = T[i];
  result.correspondence_set_.reserve((size_t)n_in);
  for (size_t k = 0; k < corres.size(); k++)
    if (inlier[k]) result.correspondence_set_.push_back(corres[k]);   // Registration.cpp:105
  return result;
}

// GetInformationMatrixFromPointClouds, Registration.cpp:303-355
inline o3d_types::Mat6 GetInformationMatrixFromPointClouds(const o3d_types::PointCloud &source, const o3d_types::PointCloud &target,
                                                           double max_correspondence_distance,
                                                           const o3d_types::Mat4 &transformation, Context &c = Context::Default()) {
  o3d_types::Mat6 GTG;
  Check(c.get(), o3d_information_matrix(c.get(), (const double *)source.points_.data(), (int64_t)source.points_.size(),
                                        (const double *)target.points_.data(), (int64_t)target.points_.size(),
                                        max_correspondence_distance, transformation.data(), O3D_MEM_HOST, GTG.data()));
  return GTG;
}

// RegistrationICP over independent pairs (no counterpart in the reference, which would loop over Registration.cpp:129-168):
// results[p] as the single call on (sources[p], targets[p], inits[p]) would give.  inits empty: identity for every pair.
inline std::vector<RegistrationResult> RegistrationICPBatch(const std::vector<const o3d_types::PointCloud *> &sources,
                                                            const std::vector<const o3d_types::PointCloud *> &targets,
                                                            double max_correspondence_distance,
                                                            const std::vector<o3d_types::Mat4> &inits = {},
                                                            const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false),
                                                            const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria(),
                                                            int n_workers = 0, Context &c = Context::Default()) {
  if (!IsAccelerated(estimation)) throw std::runtime_error("open3d_amd: RegistrationICPBatch needs a point-to-point or point-to-plane estimation");
  const size_t n = sources.size();
  if (targets.size() != n || (!inits.empty() && inits.size() != n)) throw std::runtime_error("open3d_amd: RegistrationICPBatch: size mismatch");
  const bool p2l = estimation.GetTransformationEstimationType() == TransformationEstimationType::PointToPlane;
  int with_scaling = 0;
  if (!p2l) with_scaling = static_cast<const TransformationEstimationPointToPoint &>(estimation).with_scaling_ ? 1 : 0;
  std::vector<const double *> src(n), tgt(n), nrm(n);
  std::vector<int64_t> ns(n), nt(n), ncorr(n);
  std::vector<std::vector<int32_t>> corr(n);
  std::vector<int32_t *> corr_ptr(n);
  std::vector<double> init_flat(inits.empty() ? 0 : 16 * n), T(16 * n), fit(n), rmse(n);
  std::vector<int> iters(n), status(n);
  for (size_t p = 0; p < n; p++) {
    src[p] = (const double *)sources[p]->points_.data();
    tgt[p] = (const double *)targets[p]->points_.data();
    nrm[p] = targets[p]->HasNormals() ? (const double *)targets[p]->normals_.data() : nullptr;
    ns[p] = (int64_t)sources[p]->points_.size();
    nt[p] = (int64_t)targets[p]->points_.size();
    corr[p].resize(sources[p]->points_.size());
    corr_ptr[p] = corr[p].data();
    if (!inits.empty())
      for (int i = 0; i < 16; i++) init_flat[16 * p + i] = inits[p].data()[i];
  }
  Check(c.get(), o3d_registration_icp_batch(c.get(), (int64_t)n, src.data(), ns.data(), tgt.data(), nrm.data(), nt.data(),
                                            max_correspondence_distance, inits.empty() ? nullptr : init_flat.data(),
                                            p2l ? O3D_EST_POINT_TO_PLANE : O3D_EST_POINT_TO_POINT, with_scaling,
                                            criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_, O3D_MEM_HOST,
                                            n_workers, T.data(), corr_ptr.data(), ncorr.data(), fit.data(), rmse.data(), iters.data(),
                                            status.data()));
  std::vector<RegistrationResult> out(n);
  for (size_t p = 0; p < n; p++) {
    for (int i = 0; i < 16; i++) out[p].transformation_.data()[i] = T[16 * p + i];
    out[p].fitness_ = fit[p];
    out[p].inlier_rmse_ = rmse[p];
    detail::ExpandCorrespondences(corr[p], out[p].correspondence_set_);
  }
  return out;
}

}  // namespace registration
}  // namespace pipelines
}  // namespace open3d_amd

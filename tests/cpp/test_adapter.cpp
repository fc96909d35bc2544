// Drives the C++ adapter (open3d_amd/cpp/open3d_amd_registration.h) the way the reference's callers drive
// open3d::pipelines::registration: value types in, RegistrationResult out, exceptions for the reference's LogError cases.
// usage: test_adapter <in.bin> <out.bin>
//   in : int64 n_src, int64 n_tgt, double radius, double voxel, src[3n], tgt[3m], tgt_normals[3m]
//   out: p2plane {T[16], fitness, rmse, n_pairs, pairs[2*n_pairs] as double}, evaluate {fitness, rmse, n_pairs},
//        voxel {m, pts[3m]}, normals(knn 30) {n, nrm[3n]}, error flags {3 doubles}, information matrix {36},
//        host-estimation ICP (user subclass) {T[16], fitness, rmse, n_pairs, virtual calls}, ComputeRMSE {p2p, p2plane},
//        batch of two pairs {T[16], fitness, rmse} x 2, coloured ICP with synthetic colours {T[16], fitness, rmse}
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "open3d_amd_registration.h"

using namespace open3d_amd;
namespace reg = open3d_amd::pipelines::registration;

static void put(std::vector<double> &o, double v) { o.push_back(v); }

// A user-defined estimation (type Unspecified), as a caller of the reference may write one: the loop must call it through
// the virtual interface with the transformed source.
class MyEstimation : public reg::TransformationEstimation {
 public:
  mutable int calls = 0;
  reg::TransformationEstimationType GetTransformationEstimationType() const override {
    return reg::TransformationEstimationType::Unspecified;
  }
  double ComputeRMSE(const geometry::PointCloud &s, const geometry::PointCloud &t, const reg::CorrespondenceSet &c) const override {
    return reg::ComputeRMSEOnDevice(O3D_EST_POINT_TO_PLANE, s, t, c);
  }
  Matrix4d ComputeTransformation(const geometry::PointCloud &s, const geometry::PointCloud &t,
                                 const reg::CorrespondenceSet &c) const override {
    calls++;
    return reg::ComputeTransformationOnDevice(O3D_EST_POINT_TO_PLANE, false, s, t, c);
  }
};

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  int64_t ns, nt;
  double radius, voxel;
  if (fread(&ns, 8, 1, f) != 1 || fread(&nt, 8, 1, f) != 1 || fread(&radius, 8, 1, f) != 1 || fread(&voxel, 8, 1, f) != 1) return 4;
  geometry::PointCloud source, target;
  source.points_.resize(ns);
  target.points_.resize(nt);
  target.normals_.resize(nt);
  if (fread(source.points_.data(), 24, ns, f) != (size_t)ns || fread(target.points_.data(), 24, nt, f) != (size_t)nt ||
      fread(target.normals_.data(), 24, nt, f) != (size_t)nt)
    return 5;
  fclose(f);
  std::vector<double> out;

  reg::RegistrationResult r = reg::RegistrationICP(source, target, radius, Matrix4d::Identity(),
                                                   reg::TransformationEstimationPointToPlane(), reg::ICPConvergenceCriteria());
  for (int i = 0; i < 16; i++) put(out, r.transformation_.m[i]);
  put(out, r.fitness_); put(out, r.inlier_rmse_); put(out, (double)r.correspondence_set_.size());
  for (const auto &c : r.correspondence_set_) { put(out, c[0]); put(out, c[1]); }

  reg::RegistrationResult e = reg::EvaluateRegistration(source, target, radius, r.transformation_);
  put(out, e.fitness_); put(out, e.inlier_rmse_); put(out, (double)e.correspondence_set_.size());

  auto ds = target.VoxelDownSample(voxel);
  put(out, (double)ds->points_.size());
  for (const auto &p : ds->points_) { put(out, p(0)); put(out, p(1)); put(out, p(2)); }
  if (!ds->HasNormals()) return 6;

  geometry::PointCloud tn;
  tn.points_ = target.points_;
  tn.EstimateNormals();
  put(out, (double)tn.normals_.size());
  for (const auto &p : tn.normals_) { put(out, p(0)); put(out, p(1)); put(out, p(2)); }

  // the reference's three LogError cases surface as std::runtime_error with its message text
  double flags[3] = {0, 0, 0};
  try { reg::RegistrationICP(source, target, 0.0); } catch (const std::runtime_error &x) {
    flags[0] = std::string(x.what()).find("Invalid max_correspondence_distance") != std::string::npos;
  }
  try {
    geometry::PointCloud bare;
    bare.points_ = target.points_;
    reg::RegistrationICP(source, bare, radius, Matrix4d::Identity(), reg::TransformationEstimationPointToPlane());
  } catch (const std::runtime_error &x) {
    flags[1] = std::string(x.what()).find("require pre-computed normal vectors") != std::string::npos;
  }
  try { target.VoxelDownSample(-1.0); } catch (const std::runtime_error &x) {
    flags[2] = std::string(x.what()).find("voxel_size <= 0") != std::string::npos;
  }
  for (double v : flags) put(out, v);

  Matrix6d G = reg::GetInformationMatrixFromPointClouds(source, target, radius, r.transformation_);
  for (int i = 0; i < 36; i++) put(out, G.m[i]);

  MyEstimation mine;
  reg::RegistrationResult h = reg::RegistrationICP(source, target, radius, Matrix4d::Identity(), mine, reg::ICPConvergenceCriteria());
  for (int i = 0; i < 16; i++) put(out, h.transformation_.m[i]);
  put(out, h.fitness_); put(out, h.inlier_rmse_); put(out, (double)h.correspondence_set_.size()); put(out, (double)mine.calls);

  put(out, reg::TransformationEstimationPointToPoint().ComputeRMSE(source, target, r.correspondence_set_));
  put(out, reg::TransformationEstimationPointToPlane().ComputeRMSE(source, target, r.correspondence_set_));

  Matrix4d shifted = Matrix4d::Identity();
  shifted(0, 3) = 0.25 * radius;
  std::vector<reg::RegistrationResult> b = reg::RegistrationICPBatch({&source, &source}, {&target, &target}, radius,
                                                                      {Matrix4d::Identity(), shifted},
                                                                      reg::TransformationEstimationPointToPlane());
  if (b.size() != 2) return 8;
  for (const auto &x : b) {
    for (int i = 0; i < 16; i++) put(out, x.transformation_.m[i]);
    put(out, x.fitness_); put(out, x.inlier_rmse_);
  }

  // colours: a fixed function of the point (the Python side builds the same arrays)
  geometry::PointCloud cs = source, ct = target;
  cs.colors_.resize(ns);
  ct.colors_.resize(nt);
  for (int64_t i = 0; i < ns; i++) { const auto &p = cs.points_[i]; cs.colors_[i] = Vector3d(0.5 + 0.4 * std::sin(2.0 * p(0)), 0.5 + 0.4 * std::cos(1.5 * p(1)), 0.5); }
  for (int64_t i = 0; i < nt; i++) { const auto &p = ct.points_[i]; ct.colors_[i] = Vector3d(0.5 + 0.4 * std::sin(2.0 * p(0)), 0.5 + 0.4 * std::cos(1.5 * p(1)), 0.5); }
  reg::RegistrationResult cr = reg::RegistrationColoredICP(cs, ct, radius);
  for (int i = 0; i < 16; i++) put(out, cr.transformation_.m[i]);
  put(out, cr.fitness_); put(out, cr.inlier_rmse_);

  // global registration: FPFH on both clouds, RANSAC with a fixed seed (the Python side repeats the run)
  geometry::PointCloud gs = source, gt = target;
  gs.EstimateNormals(geometry::KDTreeSearchParamHybrid(2.0 * radius, 30));
  gt.EstimateNormals(geometry::KDTreeSearchParamHybrid(2.0 * radius, 30));
  for (auto *pc : {&gs, &gt})   // one orientation for both clouds (the height field faces +z)
    for (auto &nv : pc->normals_)
      if (nv(2) < 0.0) nv = Vector3d(-nv(0), -nv(1), -nv(2));
  std::vector<double> fs = reg::ComputeFPFHFeatureData(gs, geometry::KDTreeSearchParamHybrid(5.0 * radius, 100));
  std::vector<double> ft = reg::ComputeFPFHFeatureData(gt, geometry::KDTreeSearchParamHybrid(5.0 * radius, 100));
  reg::CorrespondenceCheckers chk;
  chk.edge_length_similarity_ = 0.9;
  chk.distance_threshold_ = 1.5 * radius;
  reg::RegistrationResult gr = reg::RegistrationRANSACBasedOnFeatureMatching(
      gs, gt, fs.data(), ft.data(), 33, 1.5 * radius, reg::TransformationEstimationPointToPoint(false), 4, chk,
      reg::RANSACConvergenceCriteria(20000, 500), 5);
  for (int i = 0; i < 16; i++) put(out, gr.transformation_.m[i]);
  put(out, gr.fitness_); put(out, gr.inlier_rmse_); put(out, (double)gr.correspondence_set_.size());

  // RANSAC over the ICP result's own correspondence set (all inliers): the estimate must keep (nearly) all of them
  reg::RegistrationResult rc = reg::RegistrationRANSACBasedOnCorrespondence(source, target, r.correspondence_set_, radius,
                                                                            reg::TransformationEstimationPointToPoint(false), 6,
                                                                            reg::RANSACConvergenceCriteria(200, 200), 11);
  for (int i = 0; i < 16; i++) put(out, rc.transformation_.m[i]);
  put(out, rc.fitness_); put(out, rc.inlier_rmse_); put(out, (double)rc.correspondence_set_.size());

  // device-resident clouds: the two-scale chain (one upload per cloud) and, step by step, what it is made of
  {
    geometry::PointCloud bare;
    bare.points_ = target.points_;
    std::vector<reg::RegistrationResult> ms = reg::MultiScaleICP(source, bare, {2.0 * voxel, voxel});
    if (ms.size() != 2) return 9;
    for (const auto &x : ms) {
      for (int i = 0; i < 16; i++) put(out, x.transformation_.m[i]);
      put(out, x.fitness_); put(out, x.inlier_rmse_); put(out, (double)x.correspondence_set_.size());
    }
    reg::DeviceCloud dt(bare);
    reg::DeviceCloud dv = dt.VoxelDownSample(voxel);
    dv.EstimateNormals(geometry::KDTreeSearchParamHybrid(2.0 * voxel, 30));
    geometry::PointCloud back = dv.Download();
    put(out, (double)back.points_.size());
    for (const auto &p : back.points_) { put(out, p(0)); put(out, p(1)); put(out, p(2)); }
    for (const auto &p : back.normals_) { put(out, p(0)); put(out, p(1)); put(out, p(2)); }
  }

  f = fopen(argv[2], "wb");
  if (!f) return 7;
  fwrite(out.data(), 8, out.size(), f);
  fclose(f);
  return 0;
}

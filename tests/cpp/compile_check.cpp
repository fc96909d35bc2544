// Parses and type-checks the O3D_AMD_USE_OPEN3D_TYPES branch of the adapter against tests/cpp/compile_stub (types only: see
// the header there).  Compiled with -fsyntax-only; never linked, never run.
#define O3D_AMD_USE_OPEN3D_TYPES
#define O3D_AMD_OPEN3D_TYPES_HEADER "compile_stub/open3d_types_stub.h"
#include "open3d_amd_registration.h"

namespace reg = open3d_amd::pipelines::registration;
using open3d::geometry::PointCloud;

// every entry point a patched Registration.cpp / PointCloud.cpp / EstimateNormals.cpp would call
void instantiate(const PointCloud &source, PointCloud &target) {
  open3d::pipelines::registration::RegistrationResult r =
      reg::RegistrationICP(source, target, 0.05, Eigen::Matrix4d::Identity(), open3d::pipelines::registration::TransformationEstimationPointToPlane(),
                           open3d::pipelines::registration::ICPConvergenceCriteria());
  r = reg::EvaluateRegistration(source, target, 0.05, r.transformation_);
  std::shared_ptr<PointCloud> down = open3d_amd::VoxelDownSample(source, 0.1);
  open3d_amd::EstimateNormals(target, open3d::geometry::KDTreeSearchParamHybrid(0.2, 30));
  Eigen::Matrix6d info = reg::GetInformationMatrixFromPointClouds(source, target, 0.05, r.transformation_);
  Eigen::Matrix4d T = reg::ComputeTransformationOnDevice(O3D_EST_POINT_TO_POINT, false, source, target, r.correspondence_set_);
  double rmse = reg::ComputeRMSEOnDevice(O3D_EST_POINT_TO_PLANE, source, target, r.correspondence_set_);
  std::vector<open3d::pipelines::registration::RegistrationResult> ms = reg::MultiScaleICP(source, target, {0.1, 0.05});
  reg::DeviceCloud d(source);
  (void)info; (void)T; (void)rmse; (void)ms; (void)down; (void)d;
}

// placeholder until the preprocessing kernels land
#include "o3d_internal.h"
int o3d_voxel_down_sample_impl(o3d_ctx *ctx, const double *, const double *, const double *, int64_t, double, int, double *,
                               double *, double *, int32_t *, int64_t *) {
  return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "VoxelDownSample: not built yet");
}
int o3d_estimate_normals_impl(o3d_ctx *ctx, const double *, int64_t, int, int, double, int, double *, int) {
  return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "EstimateNormals: not built yet");
}

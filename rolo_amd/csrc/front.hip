// Front end (K1-K4) — placeholder until the kernels land; see DESIGN.md.
#include "rolo_internal.hpp"
struct rolo_ctx;
extern "C" {
void rolo_front_destroy(rolo_ctx*) {}
void rolo_front_default_params(rolo_front_params* p) {
  p->n_scan = 32; p->horizon_scan = 1024; p->downsample_rate = 1;
  p->lidar_min_range = 2.0f; p->lidar_max_range = 1000.0f;
  p->edge_threshold = 0.8f; p->surf_threshold = 0.1f; p->odometry_surf_leaf_size = 0.4f;
}
int rolo_project_frame(rolo_ctx*, const rolo_front_params*, const float*, int, const uint16_t*, int, float*, int32_t*, float*, int32_t*, int32_t*, float*, int*) { return ROLO_EUNSUPPORTED; }
int rolo_extract_features(rolo_ctx*, const rolo_front_params*, float*, int*, float*, int*, float*, int32_t*, int32_t*) { return ROLO_EUNSUPPORTED; }
}

/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header). PARITY UNPINNED by the reference itself.
 * CPU restatement of the back end's scan-to-submap optimisation (src/backMapping.cpp:681-1058); see rolo_oracle_backend.cpp. */
#ifndef ROLO_ORACLE_BACKEND_H
#define ROLO_ORACLE_BACKEND_H
#ifdef __cplusplus
extern "C" {
#endif
/* scan2MapOptimization: corner / surf = laserCloudCornerLastDS / laserCloudSurfLastDS, map_* = laserCloud*FromMapDS (n x 4 floats: x, y, z, intensity),
 * tf6 = transformTobeMapped (roll, pitch, yaw, x, y, z), updated in place. stats5 = skipped, iterations, converged, degenerate, n_selected (last iteration).
 * selected_out[n_corner + n_surf] / coeff_out[(n_corner + n_surf) * 4]: laserCloudOri*Flag / coeffSel of the LAST iteration (optional). */
int orc_scan2map(const float* corner, int n_corner, const float* surf, int n_surf, const float* map_corner, int m_corner, const float* map_surf, int m_surf,
                 float* tf6, int edge_min, int surf_min, int threads, int* stats5, unsigned char* selected_out, float* coeff_out);
#ifdef __cplusplus
}
#endif
#endif

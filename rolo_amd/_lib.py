"""ctypes loader for rolo_amd/librolo_hip.so (the C ABI of include/rolo_hip.h).

There is no CPU fallback: importing works without a GPU (so the CPU test tier can check the exported symbols),
but every compute entry point needs a HIP device and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROLO_HIP_LIB", os.path.join(HERE, "librolo_hip.so"))  # override: A/B experiments only


class Params(C.Structure):
    _fields_ = [("k_correspondences", C.c_int), ("regularization", C.c_int), ("neighbor_search", C.c_int),
                ("voxel_type", C.c_int), ("voxel_resolution", C.c_double), ("polar_resolution", C.c_double * 3),
                ("optimizer", C.c_int), ("max_iterations", C.c_int), ("rotation_epsilon", C.c_double),
                ("transformation_epsilon", C.c_double), ("lm_max_iterations", C.c_int),
                ("lm_init_lambda_factor", C.c_double), ("fixed_iterations", C.c_int), ("q2_intended", C.c_int),
                ("overlap_knn", C.c_int), ("use_graph", C.c_int), ("fused_lm", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n_outer", C.c_int), ("converged", C.c_int), ("lm_failed", C.c_int), ("n_passes", C.c_int),
                ("n_correspondences", C.c_int), ("n_cost_only", C.c_int)]


class LmScript(C.Structure):
    """rolo_lm_script (include/rolo_hip.h): scripted pass results for the test hook rolo_debug_lm_script_*"""
    _fields_ = [("n_outer", C.c_int), ("n_trial", C.c_int), ("lin_y", C.POINTER(C.c_double)), ("lin_H", C.POINTER(C.c_double)),
                ("lin_b", C.POINTER(C.c_double)), ("lin_n", C.POINTER(C.c_int32)), ("err_y", C.POINTER(C.c_double))]


class TraceRec(C.Structure):
    _fields_ = [("stage", C.c_int), ("outer", C.c_int), ("trial", C.c_int), ("accepted", C.c_int),
                ("y0", C.c_double), ("yi", C.c_double), ("rho", C.c_double), ("lambda_", C.c_double),
                ("dnorm", C.c_double)]


class Deskew(C.Structure):
    _fields_ = [("enabled", C.c_int), ("odom_incre_rpy", C.c_float * 3), ("scan_period", C.c_float), ("odom_time_diff", C.c_double)]


class CloudLayout(C.Structure):
    _fields_ = [("point_step", C.c_int), ("off_x", C.c_int), ("off_y", C.c_int), ("off_z", C.c_int), ("off_ring", C.c_int),
                ("ring_bytes", C.c_int), ("off_time", C.c_int), ("time_kind", C.c_int)]


class FrontParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("downsample_rate", C.c_int),
                ("lidar_min_range", C.c_float), ("lidar_max_range", C.c_float), ("edge_threshold", C.c_float),
                ("surf_threshold", C.c_float), ("odometry_surf_leaf_size", C.c_float)]


class Scan2MapStats(C.Structure):
    _fields_ = [("skipped", C.c_int), ("iterations", C.c_int), ("converged", C.c_int), ("degenerate", C.c_int), ("n_selected", C.c_int)]


class EskfOptions(C.Structure):   # include/rolo_fusion.h
    _fields_ = [(k, C.c_double) for k in ("max_dt", "q_linear_jerk_std", "q_angular_jerk_std", "r_position_std", "r_rotation_std", "init_position_std",
                                          "init_rotation_std", "init_velocity_std", "init_angular_velocity_std", "init_acceleration_std",
                                          "init_angular_acceleration_std")] + [("maximum_iteration", C.c_int), ("convergence_limit", C.c_double)]


class FusionOdometry(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("velocity", C.c_double * 3), ("speed", C.c_double),
                ("path_appended", C.c_int), ("path_length", C.c_int)]


class FuturePoint(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("longitudinal_velocity_mps", C.c_double),
                ("lateral_velocity_mps", C.c_double), ("heading_rate_rps", C.c_double), ("is_final", C.c_int)]


dp, fp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p

# include/rolo_fusion.h
FUSION_SYMBOLS = {
    "rolo_eskf_default_options": (None, [C.POINTER(EskfOptions)]),
    "rolo_eskf_create": (C.c_int, [C.POINTER(EskfOptions), C.POINTER(vp)]),
    "rolo_eskf_destroy": (None, [vp]),
    "rolo_eskf_copy": (C.c_int, [vp, vp]),
    "rolo_eskf_reset": (None, [vp]),
    "rolo_eskf_initialized": (C.c_int, [vp]),
    "rolo_eskf_last_time": (C.c_double, [vp]),
    "rolo_eskf_process_measurement": (C.c_int, [vp, C.c_double, dp, dp, dp]),
    "rolo_eskf_state_predict": (C.c_int, [vp, C.c_double]),
    "rolo_eskf_get_state": (None, [vp, dp, dp, dp, dp, dp, dp]),
    "rolo_eskf_get_covariance": (None, [vp, dp]),
    "rolo_eskf_state_propagate": (C.c_int, [vp, C.c_double, C.c_double, dp, C.c_int]),
    "rolo_fusion_create": (C.c_int, [C.POINTER(EskfOptions), C.POINTER(vp)]),
    "rolo_fusion_destroy": (None, [vp]),
    "rolo_fusion_mapping_odometry": (C.c_int, [vp, C.c_double, dp, dp]),
    "rolo_fusion_lidar_odometry": (C.c_int, [vp, C.c_double, dp, dp]),
    "rolo_fusion_timer": (C.c_int, [vp, C.c_double, C.POINTER(FusionOdometry)]),
    "rolo_fusion_predict_timer": (C.c_int, [vp, C.POINTER(FuturePoint), C.c_int]),
    "rolo_fusion_filter": (vp, [vp]),
}

# every symbol include/rolo_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rolo_last_error": (C.c_char_p, []),
    "rolo_device_count": (C.c_int, []),
    "rolo_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "rolo_ctx_destroy": (None, [vp]),
    "rolo_ctx_acquire": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "rolo_ctx_release": (None, [vp]),
    "rolo_ctx_pool_clear": (None, []),
    "rolo_default_params": (None, [C.POINTER(Params)]),
    "rolo_set_params": (C.c_int, [vp, C.POINTER(Params)]),
    "rolo_ctx_stream": (vp, [vp]),
    "rolo_set_target": (C.c_int, [vp, fp, C.c_int, C.c_int]),
    "rolo_set_source": (C.c_int, [vp, fp, C.c_int, C.c_int]),
    "rolo_set_target_device": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "rolo_set_source_device": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "rolo_swap_source_and_target": (C.c_int, [vp]),
    "rolo_adopt_target_covariances": (C.c_int, [vp]),
    "rolo_clear_source": (C.c_int, [vp]),
    "rolo_clear_target": (C.c_int, [vp]),
    "rolo_compute_covariances": (C.c_int, [vp]),
    "rolo_get_source_covariances": (C.c_int, [vp, dp]),
    "rolo_get_target_covariances": (C.c_int, [vp, dp]),
    "rolo_set_source_covariances": (C.c_int, [vp, dp]),
    "rolo_set_target_covariances": (C.c_int, [vp, dp]),
    "rolo_get_knn": (C.c_int, [vp, C.c_int, ip, fp]),
    "rolo_build_voxelmap": (C.c_int, [vp]),
    "rolo_scan2map_optimize": (C.c_int, [vp, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.POINTER(Scan2MapStats),
                                          C.POINTER(C.c_ubyte), fp]),
    "rolo_scan2map_set_submap": (C.c_int, [vp, fp, C.c_int, fp, C.c_int]),
    "rolo_num_voxels": (C.c_int, [vp]),
    "rolo_num_edge_points": (C.c_int, [vp]),
    "rolo_get_voxels": (C.c_int, [vp, ip, ip, dp, dp]),
    "rolo_get_target_voxel_keys": (C.c_int, [vp, ip]),
    "rolo_so3_linearize": (C.c_int, [vp, dp, dp, dp, dp]),
    "rolo_linearize": (C.c_int, [vp, dp, dp, dp, dp]),
    "rolo_compute_error": (C.c_int, [vp, dp, dp]),
    "rolo_get_correspondences": (C.c_int, [vp, ip, ip]),
    "rolo_t3_linearize": (C.c_int, [vp, dp, dp, dp, C.c_double, C.c_double, C.c_float, dp, dp, dp]),
    "rolo_compute_t_error": (C.c_int, [vp, dp, dp, dp, C.c_double, C.c_double, C.c_float, dp]),
    "rolo_align": (C.c_int, [vp, fp, fp, dp, C.POINTER(Stats)]),
    "rolo_compute_translation": (C.c_int, [vp, dp, dp, dp, C.c_double, C.c_double, C.c_float, C.POINTER(Stats)]),
    "rolo_register_async": (C.c_int, [vp, fp, dp, dp, dp, C.c_double, C.c_double, C.c_float]),
    "rolo_register_wait": (C.c_int, [vp, fp, dp, dp, C.POINTER(Stats), C.POINTER(Stats)]),
    "rolo_batch_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(vp)]),
    "rolo_batch_destroy": (None, [vp]),
    "rolo_batch_size": (C.c_int, [vp]),
    "rolo_batch_member": (vp, [vp, C.c_int]),
    "rolo_batch_register_async": (C.c_int, [vp, fp, dp, dp, dp, C.c_double, C.c_double, C.c_float]),
    "rolo_batch_register_wait": (C.c_int, [vp, fp, dp, dp, C.POINTER(Stats), C.POINTER(Stats)]),
    "rolo_get_final_hessian": (C.c_int, [vp, dp]),
    "rolo_get_trace": (C.c_int, [vp, C.POINTER(TraceRec), C.c_int]),
    "rolo_transform_cloud": (C.c_int, [vp, fp, fp, C.c_int, C.c_int, fp]),
    "rolo_shard_range": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rolo_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rolo_set_shard_knn": (C.c_int, [vp, C.c_int]),
    "rolo_debug_lm_script_align": (C.c_int, [vp, C.POINTER(LmScript), fp, C.c_int, fp, dp, C.POINTER(Stats)]),
    "rolo_debug_lm_script_translation": (C.c_int, [vp, C.POINTER(LmScript), dp, dp, dp, C.c_double, C.c_double, C.c_float, C.c_int, C.POINTER(Stats)]),
    "rolo_set_load_hint": (C.c_int, [vp, C.c_int]),
    "rolo_set_shard": (C.c_int, [vp, C.c_int, C.c_int]),
    "rolo_comm_unique_id": (C.c_int, [vp]),
    "rolo_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "rolo_comm_destroy": (C.c_int, [vp]),
    "rolo_peer_export": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "rolo_peer_connect": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "rolo_peer_disconnect": (C.c_int, [vp]),
    "rolo_peer_selftest": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
    "rolo_peer_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p]),
    "rolo_ctx_counters": (C.c_int, [vp, C.POINTER(C.c_longlong), C.c_int]),
    "rolo_debug_chain": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rolo_alloc_count": (C.c_longlong, []),
    "rolo_prof_enable": (C.c_int, [vp, C.c_int]),
    "rolo_prof_read": (C.c_int, [vp, C.c_int, fp, C.c_int]),
    "rolo_odom_create": (C.c_int, [vp, C.c_float, C.POINTER(vp)]),
    "rolo_odom_destroy": (None, [vp]),
    "rolo_odom_backend_odometry": (C.c_int, [vp, C.c_double]),
    "rolo_odom_cloud": (C.c_int, [vp, C.c_double, fp, C.c_int, fp, C.c_int, fp, dp, dp]),
    "rolo_odom_frame": (C.c_int, [vp, C.POINTER(FrontParams), C.c_double, vp, C.c_int, vp, C.c_int, C.c_int, fp, dp, dp, C.POINTER(C.c_int)]),
    "rolo_odom_submit": (C.c_int, [vp, C.POINTER(FrontParams), C.c_double, vp, C.c_int, vp, C.c_int, C.c_int]),
    "rolo_odom_collect": (C.c_int, [vp, fp, dp, dp, C.POINTER(C.c_int)]),
    "rolo_odom_submit_msg": (C.c_int, [vp, C.POINTER(FrontParams), C.c_double, vp, C.POINTER(CloudLayout), C.c_int, C.c_int]),
    "rolo_odom_get_features": (C.c_int, [vp, fp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rolo_odom_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
    "rolo_odom_set_deskew": (C.c_int, [vp, C.POINTER(Deskew), vp, C.c_int, C.c_int]),
    "rolo_odom_increment": (None, [fp, fp, fp]),
    "rolo_affine3f_rotation": (None, [fp, fp]),
    "rolo_front_set_deskew": (C.c_int, [vp, C.POINTER(Deskew), vp, C.c_int, C.c_int]),
    "rolo_front_default_params": (None, [C.POINTER(FrontParams)]),
    "rolo_project_frame": (C.c_int, [vp, C.POINTER(FrontParams), fp, C.c_int, C.POINTER(C.c_uint16), C.c_int, fp, ip, fp,
                                     ip, ip, fp, C.POINTER(C.c_int)]),
    "rolo_front_load_projection": (C.c_int, [vp, C.POINTER(FrontParams), fp, ip, fp, ip, ip, C.c_int]),
    "rolo_extract_features": (C.c_int, [vp, C.POINTER(FrontParams), fp, C.POINTER(C.c_int), fp, C.POINTER(C.c_int), fp,
                                        ip, ip]),
}

_lib = None


def lib() -> C.CDLL:
    """Load librolo_hip.so. Raises if the extension has not been built — never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m rolo_amd.build` (needs hipcc); "
                               "rolo_amd has no CPU / PyTorch fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same soname as /opt/rocm's, other
        # version). Whichever loads first serves both; if this library pulled in /opt/rocm's first, a later
        # torch.cuda.init() fails with "No HIP GPUs are available". Callers that hand over torch device pointers
        # need torch anyway, so let it load its runtime first when it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(FUSION_SYMBOLS.items()):
            f = getattr(L, name)  # AttributeError if a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class RoloError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = lib().rolo_last_error()
        super().__init__(f"{where} failed with code {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code: int, where: str) -> int:
    if code < 0:
        raise RoloError(code, where)
    return code

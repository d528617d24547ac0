/* rolo_hip.h — C ABI of librolo_hip.so, the MI355X-native (gfx950, HIP) implementation of sdwyc/ROLO's
 * per-frame scan-matching hot path. This is the drop-in boundary: plain pointers and sizes, no C++/torch
 * types. Every entry point names the reference interface it replaces (paths relative to the ROLO repository).
 *
 * Operator API replaced: fast_gicp::RotVGICP<pcl::PointXYZI, pcl::PointXYZI> of librot_gicp.so
 *   (include/rot_gicp/gicp/rot_vgicp.hpp:72-104, include/rot_gicp/gicp/lsq_registration.hpp:51-62), as driven by
 *   LidarOdometry::scanRegeistration (src/lidarOdometry.cpp:460-494).
 * Node cores replaced: ImageProjection::projectPointCloud/cloudExtraction (src/imageProjection.cpp:399-505),
 *   FeatureExtraction::calculateSmoothness/markOccludedPoints/extractFeatures (src/featureExtraction.cpp:87-266),
 *   LidarOdometry::cloudHandler (src/lidarOdometry.cpp:503-570).
 *
 * All functions return 0 on success or a negative ROLO_E* code; nothing throws or aborts. A context is
 * single-threaded (like one RotVGICP instance); several contexts may run concurrently on one device.
 * Host pointers unless the name ends in _device. Matrices are row-major.
 */
#ifndef ROLO_HIP_H
#define ROLO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ROLO_OK 0
#define ROLO_EINVAL (-1)        /* bad argument */
#define ROLO_ETOOFEW (-2)       /* cloud has fewer than k points (reference: FLANN clamps k -> UB) */
#define ROLO_ENOCORR (-4)       /* empty correspondence set (reference: H = 0 -> NaN) */
#define ROLO_ESTATE (-5)        /* call order: a prerequisite stage has not run */
#define ROLO_EHIP (-6)          /* HIP runtime error; see rolo_last_error() */
#define ROLO_EUNSUPPORTED (-7)  /* option of the reference API that this build does not implement */
#define ROLO_EALIAS (-8)        /* reference: std::invalid_argument, rot_vgicp_impl.hpp:150-152 */
#define ROLO_ECOMM (-9)         /* RCCL error */
#define ROLO_EKEYRANGE (-10)    /* voxel coordinate outside +-2^20 (packed 3x21-bit key) */
#define ROLO_ENONFINITE (-11)   /* a non-finite point / covariance (degenerate neighbourhood) reached the voxel map's fixed-point sums. DEVIATION from the
                                   reference, on purpose: vmp_voxel.hpp:169-196 carries the NaN into that ONE voxel and on into H (a degraded or NaN pose, no
                                   error); this library fails the frame with this code instead of returning a finite but wrong pose. When a frame holds
                                   both this and ROLO_EKEYRANGE, this (more negative) code is reported. */

/* enum orders follow include/rot_gicp/gicp/gicp_settings.hpp:6-13 and lsq_registration.hpp:13 */
enum { ROLO_REG_NONE = 0, ROLO_REG_MIN_EIG, ROLO_REG_NORMALIZED_MIN_EIG, ROLO_REG_PLANE, ROLO_REG_FROBENIUS, ROLO_REG_PLANE_S };
enum { ROLO_DIRECT27 = 0, ROLO_DIRECT7, ROLO_DIRECT1 };
enum { ROLO_VOXEL_POLAR = 0, ROLO_VOXEL_UNIFORM };
enum { ROLO_OPT_GN = 0, ROLO_OPT_LM, ROLO_OPT_SO3_LM };

typedef struct rolo_params {
  int k_correspondences;         /* setCorrespondenceRandomness, rot_vgicp_impl.hpp:92-94 (20) */
  int regularization;            /* setRegularizationMethod :97-99 (PLANE) */
  int neighbor_search;           /* setNeighborSearchMethod :58-60 (DIRECT1) */
  int voxel_type;                /* setResolution -> UNIFORM :44-48 ; setPolarResolution -> POLAR :51-55 */
  double voxel_resolution;       /* setResolution (1.0) */
  double polar_resolution[3];    /* setPolarResolution(theta, phi, r); ctor leaves (1,0,0) */
  int optimizer;                 /* setOptimizerType, lsq_registration_impl.hpp:337-340 (SO3_LM) */
  int max_iterations;            /* pcl::Registration::setMaximumIterations; lsq ctor :11 (64) */
  double rotation_epsilon;       /* setRotationEpsilon :30-32 (2e-3) */
  double transformation_epsilon; /* pcl::Registration::setTransformationEpsilon; lsq ctor :13 (5e-4) */
  int lm_max_iterations;         /* lsq ctor :18 (10) */
  double lm_init_lambda_factor;  /* setInitialLambdaFactor :35-37 (1e-9) */
  int fixed_iterations;          /* harness knob: >0 runs exactly this many outer iterations of align() */
  int q2_intended;               /* 0 as written, 1 intended continuous-time term */
  int overlap_knn;               /* tuning knob (default 1): source and target go through ONE chain of search launches (every
                                    kernel works on the pair) instead of one chain per cloud — both searches start together */
  int use_graph;                 /* tuning knob (default 1): rolo_register_async captures the frame's fixed launch schedule in a
                                    hipGraph on the second frame with unchanged sizes / buffers / parameters and replays it */
  int fused_lm;                  /* how the LM chain of a registration is launched (tuning knob, results equal to rounding):
                                    0 = a pass launch + a controller launch per LM trial (rounds 1-5's default: ~60 launches per frame, predicated on the device state);
                                    1 = ONE launch per LM trial (the controller runs in the prologue of the next pass, in every workgroup): the shortest chain of rounds 1-5 for
                                        one frame at a time (the odometry driver turns it on), slower with several contexts in flight (-22 % with four);
                                    2 = ONE launch per FRAME (round 6, the default): 64 (other contexts' frames in flight) ... 256 (idle device) workgroups stay resident for the
                                        whole chain of both stages and exchange their rows through self-validating words between the trials (passes.hip lm_persist_kernel) — no
                                        kernel boundary inside the chain, no schedule that can fall short. Spin-waits are bounded: a launch that cannot get all its workgroups
                                        onto the chip within ROLO_LM_PERSIST_ADMIT_US (1000) leaves the stage untouched and the frame is finished as with 0
                                        (rolo_ctx_counters[12]); a later poll longer than ROLO_LM_PERSIST_TIMEOUT_MS (200) ends the frame with ROLO_ECOMM.
                                        Ranks that share a frame (rolo_peer_*, rolo_comm_*) always run as with 0: their sums cross the exchange between pass and controller */
} rolo_params;

typedef struct rolo_stats {
  int n_outer;        /* outer iterations run (nr_iterations_ + 1) */
  int converged;      /* pcl::Registration::hasConverged() */
  int lm_failed;      /* "lm not converged!!" (lsq_registration_impl.hpp:66-69,168-171); result still returned */
  int n_passes;       /* fused linearize/error passes over the source points */
  int n_correspondences;
  int n_cost_only;    /* of n_passes: passes that evaluated a trial's cost alone (compute_error / compute_t_error without the linearisation half) — counted on the device */
} rolo_stats;

typedef struct rolo_trace_rec { /* one LM trial; lm_debug_print_ table of lsq_registration_impl.hpp:299-305 */
  int stage, outer, trial, accepted; /* accepted: 1 yes, 0 no, 2 rejected-but-converged */
  double y0, yi, rho, lambda, dnorm;
} rolo_trace_rec;

typedef struct rolo_ctx rolo_ctx;

const char* rolo_last_error(void);
int rolo_device_count(void);

/* RotVGICP() / ~RotVGICP(): rot_vgicp_impl.hpp:20-42. `device` = HIP device ordinal. */
int rolo_ctx_create(int device, rolo_ctx** out);
void rolo_ctx_destroy(rolo_ctx* ctx);
/* The same for callers that construct one operator PER FRAME, as the reference does (src/lidarOdometry.cpp:460 — `RotVGICP rot_vgicp;` inside
 * scanRegeistration): rolo_ctx_release parks the context (streams, events, pinned and device buffers, the captured hipGraph of a frame) in a
 * small per-process pool after resetting everything a fresh object would not have — parameters back to the defaults, no clouds, no
 * covariances, no map, no correspondences — and rolo_ctx_acquire hands a parked context of that device out again (or creates one). The C++
 * drop-in class (include/rot_vgicp_hip.hpp) constructs / destructs through this pair, so lidarOdometry.cpp:460-494 runs unchanged WITHOUT a
 * hipMalloc per frame. rolo_ctx_pool_clear destroys what is parked (call before process exit if the HIP runtime should see every free). */
int rolo_ctx_acquire(int device, rolo_ctx** out);
void rolo_ctx_release(rolo_ctx* ctx);
void rolo_ctx_pool_clear(void);
void rolo_default_params(rolo_params* p);
int rolo_set_params(rolo_ctx* ctx, const rolo_params* p);
/* the HIP stream all work of this context is enqueued on (hipStream_t as void*), for event timing */
void* rolo_ctx_stream(rolo_ctx* ctx);

/* setInputTarget / setInputSource (rot_vgicp_impl.hpp:133-143 / :112-120): n records of `stride` floats with
 * x,y,z at float offsets 0,1,2 (pcl::PointXYZI: stride 8). The cloud is copied to the device; cached
 * covariances, voxel map and correspondences are dropped, as in the reference. */
int rolo_set_target(rolo_ctx* ctx, const float* pts, int n, int stride);
int rolo_set_source(rolo_ctx* ctx, const float* pts, int n, int stride);
int rolo_set_target_device(rolo_ctx* ctx, const float* d_pts, int n, int stride);
int rolo_set_source_device(rolo_ctx* ctx, const float* d_pts, int n, int stride);
/* swapSourceAndTarget :69-77, clearSource :102-105, clearTarget :107-110 */
int rolo_swap_source_and_target(rolo_ctx* ctx);
/* The source cloud is the previous target moved by a pure translation (LidarOdometry::stateLinearPropagation zeroes
 * the rotation, lidarOdometry.cpp:707): its covariances are those of the target up to the float rounding of the moved
 * points, so take them over instead of searching again. Call after rolo_set_source*, before rolo_set_target*.
 * Not what the reference does (it recomputes both, :460-466): results differ at the 1e-7 relative level of the
 * covariances; the drop-in default leaves it off. ROLO_ESTATE unless the target holds covariances and n matches. */
int rolo_adopt_target_covariances(rolo_ctx* ctx);
int rolo_clear_source(rolo_ctx* ctx);
int rolo_clear_target(rolo_ctx* ctx);

/* calculate_covariances :421-496 for whichever of source/target has none. */
int rolo_compute_covariances(rolo_ctx* ctx);
/* getSourceCovariances / getTargetCovariances (rot_vgicp.hpp:91-97): n x 16 doubles (Matrix4d, row-major). */
int rolo_get_source_covariances(rolo_ctx* ctx, double* covs);
int rolo_get_target_covariances(rolo_ctx* ctx, double* covs);
/* setSourceCovariances / setTargetCovariances :122-130 */
int rolo_set_source_covariances(rolo_ctx* ctx, const double* covs);
int rolo_set_target_covariances(rolo_ctx* ctx, const double* covs);
/* debug: the k nearest neighbours (indices, squared float distances) of every source / target point */
int rolo_get_knn(rolo_ctx* ctx, int which /*0 source, 1 target*/, int32_t* idx, float* d2);

/* VmfVoxelMap::create_voxelmap (vmp_voxel.hpp:167-197) on the target. */
int rolo_build_voxelmap(rolo_ctx* ctx);
int rolo_num_voxels(rolo_ctx* ctx);
/* target points of the last map build whose POLAR voxel coordinate lies within 1e-12 (in bins) of a bin edge — the only place where the
 * device's atan2 / acos (ulps away from glibc's) could put a point into a different voxel than the reference's CPU does. The synthetic
 * scans of the tests hold a few (the azimuth wrap, atan2(y, x) + pi ~ 1e-16); their keys are checked against the CPU path like all others.
 * UNIFORM keys are correctly rounded on both sides: always 0. */
int rolo_num_edge_points(rolo_ctx* ctx);
/* keys V x 3, counts V, means V x 4, covs V x 16 — voxel order is unspecified (match on keys). */
int rolo_get_voxels(rolo_ctx* ctx, int32_t* keys, int32_t* counts, double* means, double* covs);
/* polar_coord / voxel_coord (vmp_voxel.hpp:199-211) of the target points (n x 3). */
int rolo_get_target_voxel_keys(rolo_ctx* ctx, int32_t* keys);

/* so3_linearize :293-388 / linearize :225-290 (update_correspondences :173-222 fused in), compute_error :391-417.
 * T row-major 4x4. H/b may be NULL. */
int rolo_so3_linearize(rolo_ctx* ctx, const double* T, double* H9, double* b3, double* err);
int rolo_linearize(rolo_ctx* ctx, const double* T, double* H36, double* b6, double* err);
int rolo_compute_error(rolo_ctx* ctx, const double* T, double* err);
/* voxel_correspondences_ of the last linearize: per source point the 3-int key of its voxel, or found[i] = 0. */
int rolo_get_correspondences(rolo_ctx* ctx, int32_t* found /* n_src * n_offsets */, int32_t* keys /* n_src * n_offsets * 3 */);
/* t3_linearize :499-607 / compute_t_error :610-658 on the correspondences cached by the last linearize. */
int rolo_t3_linearize(rolo_ctx* ctx, const double* t3, const double* init_guess3, const double* last_t03,
                      double dtn, double dtn1, float ct_lambda, double* H36, double* b6, double* err);
int rolo_compute_t_error(rolo_ctx* ctx, const double* t3, const double* init_guess3, const double* last_t03,
                         double dtn, double dtn1, float ct_lambda, double* err);

/* pcl::Registration::align(out, guess) -> RotVGICP::computeTransformation :146-160 -> LsqRegistration::
 * computeTransformation (lsq_registration_impl.hpp:152-179). guess16 NULL = Identity. T_out_f16 =
 * getFinalTransformation() (float); T_out_d16 (optional) the double pose it was cast from. The transformed source the
 * reference also writes (pcl::transformPointCloud into `output`) is rolo_transform_cloud with T_out_f16. */
int rolo_align(rolo_ctx* ctx, const float* guess16, float* T_out_f16, double* T_out_d16, rolo_stats* stats);
/* RotVGICP::computeTranslation :163-169 -> lsq :55-80. trans3_io: in = start value, out = result. */
int rolo_compute_translation(rolo_ctx* ctx, double* trans3_io, const double* init_guess3, const double* last_t03,
                             double dtn, double dtn1, float ct_lambda, rolo_stats* stats);
/* Both stages back to back with no host synchronisation in between (scanRegeistration :460-500 as one enqueue).
 * rolo_register_async only enqueues; rolo_register_wait blocks and fetches results. */
int rolo_register_async(rolo_ctx* ctx, const float* guess16, const double* trans3_start, const double* init_guess3,
                        const double* last_t03, double dtn, double dtn1, float ct_lambda);
int rolo_register_wait(rolo_ctx* ctx, float* T_out_f16, double* T_out_d16, double* trans3_out,
                       rolo_stats* rot_stats, rolo_stats* trans_stats);
/* Batches of independent scan pairs (BASELINE config 5 — the reference would loop scanRegeistration over them).
 * A batch owns n member contexts that share one HIP stream; the caller configures every member through the normal
 * API (rolo_set_params, rolo_set_source/_target[_device]) and then registers all of them with ONE call: the members'
 * neighbourhood searches and voxel maps are enqueued back to back and their LM chains run as batched launches (one
 * pass / controller launch per trial for the whole batch), replayed from a hipGraph when nothing changed. Array
 * arguments hold one entry per member (guess16: n x 16 or NULL, vectors: n x 3). Members must share the optimizer
 * and fixed_iterations settings. */
typedef struct rolo_batch rolo_batch;
int rolo_batch_create(int device, int n_members, rolo_batch** out);
void rolo_batch_destroy(rolo_batch* b);
int rolo_batch_size(rolo_batch* b);
rolo_ctx* rolo_batch_member(rolo_batch* b, int i); /* owned by the batch: do not destroy */
int rolo_batch_register_async(rolo_batch* b, const float* guess16, const double* trans3_start, const double* init_guess3,
                              const double* last_t03, double dtn, double dtn1, float ct_lambda);
int rolo_batch_register_wait(rolo_batch* b, float* T_out_f16, double* T_out_d16, double* trans3_out, rolo_stats* rot_stats,
                             rolo_stats* trans_stats);
/* getFinalHessian (lsq_registration_impl.hpp:45-47): 6x6, Identity until a 6-dof LM step accepts */
int rolo_get_final_hessian(rolo_ctx* ctx, double* H36);
int rolo_get_trace(rolo_ctx* ctx, rolo_trace_rec* out, int cap); /* returns the number of records */

/* pcl::transformPointCloud float path used at lidarOdometry.cpp:459,492 and lsq_registration_impl.hpp:78,178 */
int rolo_transform_cloud(rolo_ctx* ctx, const float* in, float* out, int n, int stride, const float* T16);

/* Multi-GPU point sharding: every rank holds the full clouds; rank r evaluates source points
 * [r*n/W, (r+1)*n/W) in the passes and the per-pass sums are all-reduced (fp64, <= 32 values) with RCCL on the
 * context's stream. K5 shards by query point: each rank searches 1/W of the Morton-sorted queries and the 48-byte covariances are
 * all-gathered once per cloud and frame (ncclAllGather). unique_id = the 128-byte ncclUniqueId created by rank 0
 * (rolo_comm_unique_id) and distributed by the caller (e.g. torch.distributed broadcast). */
/* the slice of n source points rank r of `world` evaluates: [n*r/world, n*(r+1)/world) */
void rolo_shard_range(int n, int rank, int world, int* begin, int* end);
/* test hook: shard the passes WITHOUT a communicator — sums returned by the stage-level calls are then partial */
int rolo_set_shard(rolo_ctx* ctx, int rank, int world);
/* test hook: with rolo_set_shard and no communicator, also shard K5 (calculate_covariances, rot_vgicp_impl.hpp:430-496) by query point
 * as a communicator does — every rank sorts the whole cloud, searches only its slice of the Morton-sorted queries (whole 256-query
 * workgroups, equal slices) — but WITHOUT the all-gather: only the covariances of the own slice are valid afterwards (the rest of
 * rolo_get_*_covariances is stale / undefined); the union over the ranks must equal the unsharded result. */
int rolo_set_shard_knn(rolo_ctx* ctx, int on);
/* test hook: the LM drivers of lsq_registration_impl.hpp:55-179, 225-324 — the controller kernels of passes.hip — fed with SCRIPTED pass results instead of pass
 * kernels: the linearisation opened by outer iteration o returns lin_*[o], the trial cost of (o, trial t) returns err_y[o][t] (indices beyond the extents repeat the
 * last entry); slots of a pass row that the step must not read are poisoned with NaN. Every exit of the drivers (LM failure, iteration cap, rejected-but-converged,
 * NaN / infinite gain ratios, the cost-only passes' late linearisation) can be reached with inputs that are the same bits as a CPU statement's.
 * rolo_debug_lm_script_align: computeTransformation from `guess16` with the context's parameters; generic_ctrl != 0 runs the one-size-fits-all controller of the
 * batched launches instead of the specialised one. rolo_debug_lm_script_translation: computeTranslation afterwards (it needs the correspondence count the
 * rotation script left, as the reference's does). Results as rolo_align / rolo_compute_translation; rolo_get_trace reads the trace. */
typedef struct rolo_lm_script {
  int n_outer, n_trial;
  const double* lin_y;    /* [n_outer] */
  const double* lin_H;    /* [n_outer][36] row-major 6 x 6; a 3-dof optimiser reads the top-left 3 x 3 */
  const double* lin_b;    /* [n_outer][6] */
  const int32_t* lin_n;   /* [n_outer] correspondences of that linearisation */
  const double* err_y;    /* [n_outer][n_trial] */
} rolo_lm_script;
int rolo_debug_lm_script_align(rolo_ctx* ctx, const rolo_lm_script* script, const float* guess16, int generic_ctrl, float* T_out_f16, double* T_out_d16, rolo_stats* stats);
int rolo_debug_lm_script_translation(rolo_ctx* ctx, const rolo_lm_script* script, double* trans3_io, const double* init_guess3, const double* last_t03,
                                     double dtn, double dtn1, float ct_lambda, int generic_ctrl, rolo_stats* stats);
/* Which kernels rolo_register_async picks where the best one depends on whether the GPU is shared (not in the reference: its operator owns its CPU threads).
 * -1 (default): decided per frame — other contexts of the device have frames in flight when this one is enqueued => the kernels that share the chip best
 * (throughput); an idle device => the ones that finish soonest (latency). 0 / 1 pin the idle- / busy-device choice (profiling runs, callers that know their
 * load). It selects the neighbour search of large launches — 64-query packets (busy) or two lanes per query (idle): the same lists, bit for bit — and, since round 6, the
 * size of the resident LM kernel (fused_lm = 2: 64 workgroups busy, 256 idle): the sums of a pass then run over other groups of points and poses agree to rounding
 * (1e-12), not bit for bit. A caller that needs run-to-run identical bits under changing load pins the hint (or sets fused_lm = 0). */
int rolo_set_load_hint(rolo_ctx* ctx, int mode);
int rolo_comm_unique_id(void* unique_id128);
int rolo_comm_init(rolo_ctx* ctx, const void* unique_id128, int rank, int world);
int rolo_comm_destroy(rolo_ctx* ctx);
/* rank and size read back from the RCCL communicator itself (ncclCommUserRank / ncclCommCount); world = 0 without one */
int rolo_comm_info(rolo_ctx* ctx, int* rank, int* world);

/* The same exchange WITHOUT a collective library ("every GPU peer-writes its partial into a slot on each peer, fixed-rank-
 * order local sum"): every rank exports one device allocation (its "mailbox": slots for the 32 fp64 sums of an LM pass + the covariance
 * exchange area), the ranks swap the 64-byte handles by any means (torch.distributed all_gather, a file, a pipe), and connect. From then on
 *   - the LM controller kernel of every trial writes its shard's sums into its slot of EVERY rank's mailbox and adds all slots of its own
 *     mailbox in rank order (bit-identical on every rank): no reduce launch, no library call, one launch per trial as on one GPU, and the
 *     frame's launch schedule stays hipGraph-capturable;
 *   - K5's covariances of the own query slice are copied into every peer's exchange area by a kernel (replaces ncclAllGather).
 * Handles come from hipIpcGetMemHandle (processes of one node; HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver); contexts of ONE process
 * (several GPUs driven by one process, or two contexts on one device) are recognised and use each other's pointers directly. A rank that
 * does not answer within ROLO_PEER_TIMEOUT_MS (default 10 000) makes the waiting ranks end the registration with ROLO_ECOMM — a lost peer
 * never hangs the GPU. world <= 8 (one node). max_points >= source + target points of the largest frame (sizes the covariance area).
 * Mutually exclusive with rolo_comm_init (RCCL), which stays as the A/B. Reference loop that is split over the GPUs:
 * rot_vgicp_impl.hpp:313-382 (the per-thread Hs / bs / sum_errors of so3_linearize summed at :377-382). */
#define ROLO_PEER_HANDLE_BYTES 64
int rolo_peer_export(rolo_ctx* ctx, int world, int max_points, void* handle64);
int rolo_peer_connect(rolo_ctx* ctx, const void* handles /* world x 64 bytes, rank order */, int rank, int world);
int rolo_peer_disconnect(rolo_ctx* ctx);
/* Collective self-test of a connected group, to be run by EVERY rank (same reps) before the first frame: `reps` all-reduces of 32 known fp64
 * through the LM mailboxes and one covariance-segment push of known words into every peer's exchange area, verified on every rank. us_out[0] =
 * mean microseconds of one LM exchange launch (first one excluded), us_out[1] = microseconds of the covariance exchange (push + flags + wait).
 * ROLO_ECOMM with a message naming the rank and the words that were wrong or missing — the first crossing of a new transport (hipIpc mapping,
 * peer access, fine-grained memory over xGMI) fails here and not as a wrong pose inside a frame. */
int rolo_peer_selftest(rolo_ctx* ctx, int reps, double* us_out /* [2], optional */);
/* rank / world of the connection (world 0: none); mem_kind16 (optional, 16 chars): "finegrained" | "coarse" */
int rolo_peer_info(rolo_ctx* ctx, int* rank, int* world, char* mem_kind16);

/* Bookkeeping of rolo_register_async / _wait on this context since its creation (what bench.py reports next to the throughput):
 * out[0] frames registered, [1] of them replayed from the hipGraph, [2] captured, [3] enqueued eagerly, [4] frames whose first launch
 * schedule was too short (rolo_register_wait had to top up with host round trips), [5] synchronous chunks of predicated passes enqueued
 * by the drivers (top-ups + rolo_align / rolo_compute_translation), [6] / [7] passes the next frame's first schedule holds per stage, [8] lanes per query
 * of the last neighbour search enqueued (1: 64-query packets, 2 / 4: knn_walk_sub_kernel — picked by launch size, ROLO_KNN_SUB overrides),
 * [9] nanoseconds of HOST time spent inside rolo_register_async (graph launch / capture / eager enqueue) since the context was created or recycled,
 * [10] nanoseconds the host was blocked in rolo_register_wait's hipEventSynchronize, [11] nanoseconds of the rest of rolo_register_wait (std::chrono::steady_clock;
 * bench.py's host_enqueue_us_per_frame — round 5's verdict, item 2a), [12] frames whose resident LM kernel (fused_lm = 2) could not get all its workgroups onto the chip within its
 * admission time and gave the stage back to the host, which finished it with pass + controller launches, [13] what the context has LEARNED about load it cannot count (another
 * process on the GPU): 0 = its frames last what they last alone (idle-device kernels), 1 = trying the busy-device kernels, 2 = keeping them (rolo_set_load_hint -1 only). */
int rolo_ctx_counters(rolo_ctx* ctx, long long* out, int n);
/* device buffer (re)allocations made so far by the registration contexts of this process (every hipMalloc behind a rolo_ctx's buffers; a captured hipGraph is keyed on it):
 * a steady-state frame loop must stop moving it */
long long rolo_alloc_count(void);
/* experiment hook (profiles/tools/concurrency.py, round 5's verdict item 2b): enqueue `reps` replays of a captured chain of `n_pairs` launch PAIRS on the context's stream —
 * kind 0: empty kernels shaped like a controller (1 workgroup); 1: empty kernels shaped like a pass (`grid` workgroups of 256); 2: empty pass + empty controller
 * alternating (the LM chain's boundaries and dispatches without its instructions or traffic); 3: the context's REAL LM chain (frame begin + predicated pass / controller
 * pairs as rolo_register_async enqueues them, on the map and clouds of its last registration); 4: the real passes without the controller (every launch linearises at
 * the start pose). Asynchronous: synchronise through rolo_ctx_stream. */
int rolo_debug_chain(rolo_ctx* ctx, int kind, int n_pairs, int grid, int reps);

/* Per-kernel timing with HIP events on the context's stream (bench.py's "roofline" object). While enabled, every
 * launch of the listed kernels is bracketed by an event pair; rolo_prof_read synchronises the stream and returns
 * the durations (ms) of one slot in launch order, then forgets them. Returns the number of launches recorded. */
enum { ROLO_PROF_KNN_BUILD = 0, ROLO_PROF_KNN_WALK, ROLO_PROF_VOXEL_BUILD, ROLO_PROF_ROT_PASS, ROLO_PROF_TRANS_PASS, ROLO_PROF_CTRL, ROLO_PROF_KNN_TAIL, ROLO_PROF_LM_PASS /* fused trial: controller prologue + pass */, ROLO_PROF_N };
int rolo_prof_enable(rolo_ctx* ctx, int on);
int rolo_prof_read(rolo_ctx* ctx, int slot, float* ms, int cap);

/* ---- front end ------------------------------------------------------------------------------------------ */
typedef struct rolo_front_params {
  int n_scan, horizon_scan, downsample_rate;       /* include/rolo/utility.h:310-312 */
  float lidar_min_range, lidar_max_range;          /* :313-314 */
  float edge_threshold, surf_threshold;            /* :318-319 */
  float odometry_surf_leaf_size;                   /* :323 */
} rolo_front_params;
void rolo_front_default_params(rolo_front_params* p); /* config/params.yaml values */

/* ImageProjection::projectPointCloud + cloudExtraction (src/imageProjection.cpp:399-505), deskew off.
 * in: n_raw points (x,y,z at float offsets 0..2 of `stride`-float records) + ring[n_raw].
 * out (host, sized n_scan*horizon_scan): extracted[N*4] (x,y,z,intensity), point_col_ind[N], point_range[N],
 * start_ring[n_scan], end_ring[n_scan]; range_mat (optional) n_scan*horizon_scan. *n_valid = N. */
int rolo_project_frame(rolo_ctx* ctx, const rolo_front_params* P, const float* pts, int stride, const uint16_t* ring,
                       int n_raw, float* extracted, int32_t* point_col_ind, float* point_range, int32_t* start_ring,
                       int32_t* end_ring, float* range_mat, int* n_valid);
/* ImageProjection::deskewPoint (src/imageProjection.cpp:368-396; deskewCloudInfo :266-366 prepares its inputs): rotation-only
 * de-skew of every stored point by the front-end odometry increment over the scan, `rolo/deskewEnabled` (off in every
 * shipped config). Arms the NEXT rolo_project_frame / rolo_odom_submit on this context: rel_time[i] = fabs(point.time) of
 * raw point i (what :358-359 stores), odom_incre_rpy = odomIncreRoll/Pitch/Yaw, odom_time_diff = odomTimeDiff (:349-351; rolo_odom_increment does
 * the pose algebra). rel_time == NULL: the times come from the time field of the next message (rolo_odom_submit_msg) or, for
 * points without one (timeFlag == -1), are interpolated from the azimuth as deskewCloudInfo does (:270-327). Range, pixel and every
 * index still come from the raw point, as in the reference (:412-454). */
typedef struct rolo_deskew { int enabled; float odom_incre_rpy[3]; float scan_period; double odom_time_diff; } rolo_deskew;
int rolo_front_set_deskew(rolo_ctx* ctx, const rolo_deskew* d, const float* rel_time, int n_raw, int rel_time_on_device);
/* lidarOdomAffineFront.inverse() * lidarOdomAffineBack -> pcl::getTranslationAndEulerAngles (:345-351); poses and increment as
 * x, y, z, roll, pitch, yaw (float) */
void rolo_odom_increment(const float* front6, const float* back6, float* incre6);
/* Eigen::Affine3f::rotation() of the row-major 4x4 T — what `Rotation = transformation_interpolated.rotation().cast<double>()` reads
 * (src/lidarOdometry.cpp:474, :548; TransformFusion::affineToPose :130). For an Affine (not Isometry) transform this is
 * computeRotationScaling(): JacobiSVD<Matrix3f> of the linear part in float, U V^T with the determinant sign fix — the polar factor, which
 * differs from the linear part in the last float ulps. Host code (the pose chain of the odometry driver uses it); R9 row-major. */
void rolo_affine3f_rotation(const float* T16, float* R9);
/* The input of FeatureExtraction::laserCloudInfoHandler (src/featureExtraction.cpp:71-85) when that node runs as its own process: the arrays
 * of the received rolo/cloud_info — extracted[n_valid*4] = fromROSMsg(cloud_projected) as x, y, z, intensity; pointColInd, pointRange
 * (first n_valid entries), startRingIndex / endRingIndex [n_scan] — go where rolo_project_frame would have left them on the device;
 * rolo_extract_features then runs as usual. */
int rolo_front_load_projection(rolo_ctx* ctx, const rolo_front_params* P, const float* extracted, const int32_t* point_col_ind,
                               const float* point_range, const int32_t* start_ring, const int32_t* end_ring, int n_valid);
/* FeatureExtraction::calculateSmoothness + markOccludedPoints + extractFeatures (src/featureExtraction.cpp:87-266)
 * on the arrays rolo_project_frame left on the device (call order: project, then extract).
 * out (host): corner[nc*4], surface[ns*4] (sized for N points each); optional curvature/picked/label [N]. */
int rolo_extract_features(rolo_ctx* ctx, const rolo_front_params* P, float* corner, int* n_corner, float* surface,
                          int* n_surface, float* curvature, int32_t* neighbor_picked, int32_t* label);

/* ---- per-frame odometry driver -------------------------------------------------------------------------------
 * LidarOdometry (src/lidarOdometry.cpp:325-713) between fromROSMsg and publish, on feature clouds (n x 4 floats:
 * x, y, z, intensity). The driver keeps the node's state (previous features, LaserOdomPose, the last step for
 * the forward prediction) and runs both registration stages on `ctx`, whose registration parameters
 * (setPolarResolution(0.175, 0.175, 2.0) in the reference, :462) the caller sets with rolo_set_params. */
typedef struct rolo_odom rolo_odom;
int rolo_odom_create(rolo_ctx* ctx, float ct_lambda /* rolo/continuousTrajectoryWeight */, rolo_odom** out);
void rolo_odom_destroy(rolo_odom* o);
/* odometryHandler :440-446 — the back end's rolo/mapping/odometry; scan matching is gated on it (:537-541) */
int rolo_odom_backend_odometry(rolo_odom* o, double stamp);
/* cloudHandler :503-570. Returns 0 = first frame stored, 1 = gated (pose propagated, no registration),
 * 2 = registered; <0 error. pose6 = LaserOdomPose (x, y, z, roll, pitch, yaw) as published on
 * odomTopic+"_incremental"; rot9 / trans3 = Rotation / Translation of the frame-to-frame step. */
int rolo_odom_cloud(rolo_odom* o, double stamp, const float* corner, int n_corner, const float* surface, int n_surface,
                    float* pose6, double* rot9, double* trans3);
/* The three node cores fused, device-resident: raw frame (ImageProjection::cloudHandler input, imageProjection.cpp:151)
 * -> projection -> features -> cloudHandler of LidarOdometry, without the CloudInfoStamp hops: the feature clouds, the
 * propagated previous features and both registration inputs never leave HBM; the host sees three counts mid-frame
 * and the pose at the end. Same results and return codes as rolo_project_frame + rolo_extract_features +
 * rolo_odom_cloud. pts/ring may be device pointers (pts_on_device != 0). counts3 (optional) = N, n_corner, n_surface. */
int rolo_odom_frame(rolo_odom* o, const rolo_front_params* P, double stamp, const float* pts, int stride, const uint16_t* ring,
                    int n_raw, int pts_on_device, float* pose6, double* rot9, double* trans3, int* counts3);
/* The same in two halves, for throughput: rolo_odom_submit enqueues K1-K4 of a frame on the driver's own front-end
 * stream and returns at once; rolo_odom_collect finishes the oldest submitted frame (waits for its features, registers,
 * updates the pose). Up to two frames may be in flight, so the features of frame k+1 are extracted while frame k
 * registers — the overlap the reference gets from running its three nodes as separate processes. With host input the
 * caller must keep pts / ring valid until the matching collect. rolo_odom_frame = submit + collect. */
int rolo_odom_submit(rolo_odom* o, const rolo_front_params* P, double stamp, const float* pts, int stride, const uint16_t* ring,
                     int n_raw, int pts_on_device);
int rolo_odom_collect(rolo_odom* o, float* pose6, double* rot9, double* trans3, int* counts3);
/* The same from the message itself: `data` is the payload of the sensor_msgs/PointCloud2 that ImageProjection::cloudHandler
 * receives (n_points records of layout->point_step bytes), `layout` the byte offsets of the fields the node reads — what
 * pcl::moveFromROSMsg and the Ouster conversion loop of cachePointCloud (imageProjection.cpp:188-212) do on the host runs as a
 * kernel: x, y, z (FLOAT32), ring (UINT16 for Velodyne, UINT8 for Ouster), time (time_kind 1: Velodyne "time", FLOAT32 seconds;
 * 2: Ouster "t", UINT32 nanoseconds, * 1e-9f; 0: none). A de-skew armed without times (rolo_odom_set_deskew with rel_time =
 * NULL) takes fabs(time) of this message, or the azimuth-interpolated times when the message has no time field. */
typedef struct rolo_cloud_layout { int point_step, off_x, off_y, off_z, off_ring, ring_bytes, off_time, time_kind; } rolo_cloud_layout;
int rolo_odom_submit_msg(rolo_odom* o, const rolo_front_params* P, double stamp, const uint8_t* data, const rolo_cloud_layout* layout,
                         int n_points, int data_on_device);
/* The feature clouds of the last collected frame of the fused path, for the outgoing rolo/CloudInfoStamp (extracted_corner ++
 * extracted_surface as n x 4 floats: x, y, z, intensity; corners first). features may be NULL to query the counts. */
int rolo_odom_get_features(rolo_odom* o, float* features, int cap_points, int* n_corner, int* n_surface);
/* options of the fused path: ROLO_ODOM_REUSE_COVARIANCES (default 0) = rolo_adopt_target_covariances between frames */
#define ROLO_ODOM_REUSE_COVARIANCES 1
/* ROLO_ODOM_FUSED_LM (default 2; values 0 / 1 / 2 = rolo_params.fused_lm): how the driver's registrations launch their LM chain — it registers one frame at a
 * time, where the shortest chain wins: one launch per frame since round 6 (one launch per trial, 1, through round 5). The driver asserts the option on its context right
 * before each registration it enqueues; it does not depend on, and is not reverted by, the parameter block the caller hands to rolo_set_params. */
#define ROLO_ODOM_FUSED_LM 2
/* ROLO_ODOM_EARLY_SOURCE (default 0): when a frame is submitted with nothing else in flight (rolo_odom_frame, or submit / collect one frame at
 * a time), the propagated previous features — the SOURCE of the coming registration, known at submit time — are moved and searched on the
 * registration stream while K1-K4 of the new frame run on the front-end stream; collect searches the target alone. Same results (the search
 * of a cloud does not depend on what it is launched with), but MEASURED SLOWER on MI355X: the search of a 48 k-point cloud lasts as long
 * as its heaviest packets, alone nearly as long as together with the other cloud — two chains cost 0.68 ms per frame against 0.58 ms for the
 * pair chain after K1-K4. Kept as an A/B (env ROLO_ODOM_EARLY_SOURCE=1 arms it in every driver). */
#define ROLO_ODOM_EARLY_SOURCE 3
int rolo_odom_set_option(rolo_odom* o, int option, int value);
/* rolo_front_set_deskew for the next rolo_odom_submit / rolo_odom_frame of the fused path */
int rolo_odom_set_deskew(rolo_odom* o, const rolo_deskew* d, const float* rel_time, int n_raw, int rel_time_on_device);

/* ---- back end: scan-to-submap optimisation ---------------------------------------------------------------------
 * scan2MapOptimization (src/backMapping.cpp:681-711) with cornerOptimization :720-824, surfOptimization :827-901,
 * combineOptimizationCoeffs :904-925 and LMOptimization :929-1058: the down-sampled corner / surface points of the current scan
 * (laserCloudCornerLastDS / laserCloudSurfLastDS, n x 4 floats) against the corner / surface sub-maps (laserCloud*FromMapDS), starting
 * from and updating transformTobeMapped = (roll, pitch, yaw, x, y, z). Up to 30 Gauss-Newton iterations, each one kernel over the points
 * (exact 5-NN in the sub-map's BVH, line / plane fit, Jacobian row, J^T J reduction) and a 6 x 6 float solve on the host.
 * edge_min / surf_min = edgeFeatureMinValidNum / surfFeatureMinValidNum (utility.h: 10 / 100): with fewer features the call does nothing
 * (stats->skipped = 1); a sub-map with fewer than five points cannot answer the 5-NN association: also nothing (stats->skipped = 2). The context's source / target clouds are used as scratch for the sub-map trees: use a context of its own.
 * transformUpdate()'s clamps (:1060-1068; tolerances FLT_MAX in every shipped config) are left to the caller.
 * selected_out[n_corner + n_surf] / coeff_out[(n_corner + n_surf) * 4] (optional): laserCloudOri*Flag and coeffSel of the LAST iteration. */
typedef struct rolo_scan2map_stats { int skipped, iterations, converged, degenerate, n_selected; } rolo_scan2map_stats;
/* kdtreeCornerFromMap / kdtreeSurfFromMap ->setInputCloud (:690-691) as a call of its own: uploads the sub-map and builds its two search trees once; they stay
 * resident in the context until the next call. rolo_scan2map_optimize with map_corner = map_surf = NULL then registers against the resident sub-map (the
 * surrounding key-frame set changes every few scans, not every scan). */
int rolo_scan2map_set_submap(rolo_ctx* ctx, const float* map_corner, int m_corner, const float* map_surf, int m_surf);
int rolo_scan2map_optimize(rolo_ctx* ctx, const float* corner, int n_corner, const float* surf, int n_surf, const float* map_corner, int m_corner,
                           const float* map_surf, int m_surf, float* transformTobeMapped6, int edge_min, int surf_min, rolo_scan2map_stats* stats,
                           unsigned char* selected_out, float* coeff_out);

#ifdef __cplusplus
}
#endif
#endif /* ROLO_HIP_H */

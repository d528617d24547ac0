// Internal declarations shared by the HIP translation units of librolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rolo_hip.h"

// Issue priority of the SHORT kernels of a frame (LM passes, controller, the search's build chain): with several contexts in flight their wavefronts share SIMDs with
// another frame's neighbour search, whose eight older wavefronts per SIMD win the age-ordered issue arbitration; s_setprio lets the latency-bound chain go first
// and the search fill what is left (MI355X_MICROARCH.md: VALU issue is arbitrated by priority, then age): +3.5 % scans/s with four contexts in flight, levels 1 and 3 alike;
// raising the build chain and the covariance tail as well adds nothing (ROLO_BUILD_PRIO / ROLO_TAIL_PRIO, off). 0 = off (A/B build: --flag=-DROLO_SHORT_PRIO=0).
#ifndef ROLO_SHORT_PRIO
#define ROLO_SHORT_PRIO 1
#endif
#define ROLO_SHORT_KERNEL_PRIO() do { if (ROLO_SHORT_PRIO) __builtin_amdgcn_s_setprio(ROLO_SHORT_PRIO); } while (0)
#ifndef ROLO_BUILD_PRIO
#define ROLO_BUILD_PRIO 0
#endif
#ifndef ROLO_TAIL_PRIO
#define ROLO_TAIL_PRIO 0
#endif
#define ROLO_ALL_KERNEL_PRIO() do { if (ROLO_BUILD_PRIO) __builtin_amdgcn_s_setprio(ROLO_BUILD_PRIO); } while (0)
#define ROLO_TAIL_KERNEL_PRIO() do { if (ROLO_TAIL_PRIO) __builtin_amdgcn_s_setprio(ROLO_TAIL_PRIO); } while (0)

namespace rolo {

// ---- per-pass reduction layout (fp64): one row of NV_MAX values per workgroup, then one row total ----------
constexpr int NV_MAX = 32;
constexpr int V_YI = 0;   // cost at the trial pose on the cached correspondences (compute_error / compute_t_error)
constexpr int V_Y = 1;    // cost of the new linearisation
constexpr int V_N = 2;    // number of correspondences of the new linearisation
constexpr int V_H = 3;    // lower triangle of H, row-major: (0,0) (1,0) (1,1) (2,0) ... 6 or 21 values
constexpr int V_B = 24;   // b, 3 or 6 values
#ifndef ROLO_PASS_THREADS
#define ROLO_PASS_THREADS 256
#endif
constexpr int PASS_THREADS = ROLO_PASS_THREADS;

constexpr unsigned long long KEY_EMPTY = ~0ull;
constexpr int KEY_BIAS = 1 << 20;
#ifndef ROLO_REC_DOUBLES
#define ROLO_REC_DOUBLES 12
#endif
constexpr int REC_DOUBLES = ROLO_REC_DOUBLES;  // voxel record: mean xyz, cov xx xy xz yy yz zz, sqrt(n), n, pad (A/B: 16 = one 128-byte line per record)
constexpr int TRACE_CAP = 2048;
#ifndef ROLO_KNN_LEAF
#define ROLO_KNN_LEAF 16
#endif
// points per BVH leaf (A/B builds: 8 or 16; 16: one tree level and ~45 % of the dependent scalar fetches less for ~25 % more distance
// evaluations — walk 0.218 -> 0.200 ms on the 2 x 131 072-point pair); 64 / KNN_LEAF leaves seed a wavefront
constexpr int KNN_LEAF = ROLO_KNN_LEAF;
static_assert(KNN_LEAF == 8 || KNN_LEAF == 16, "leaf size");

struct VoxelTable {
  unsigned long long* keys;  // 2 words per slot (round 5): [2h] = packed key (KEY_EMPTY = free), [2h + 1] = compact voxel id in the low 32 bits — ONE 16-byte
                             // load returns both (voxel_dev.hpp slot_key / slot_id); rounds 1-4 kept the ids in an array of their own: a second dependent fetch per lookup
  double* rec;               // V x REC_DOUBLES
  unsigned long long* id_keys;  // V: packed key of each compact voxel id (for the getters)
  unsigned mask;             // capacity - 1
  int voxel_type;
  double voxel_resolution;
  double inv_voxel_resolution;   // 1.0 / voxel_resolution (host): the pass kernels' fast path of the UNIFORM key (voxel_dev.hpp voxel_coord_dev)
  double polar_res[3];
  int polar_exact;           // 1 (default): POLAR keys of target points near a bin edge through the correctly rounded atan2 / acos (polar_exact.hpp); 0: counted only (ROLO_POLAR_EXACT=0, the A/B)
};

struct CloudDev {          // one point cloud resident in HBM
  float4* xyz = nullptr;   // n  (x, y, z, 1)
  double* cov = nullptr;   // 6 x n  SoA: xx | xy | xz | yy | yz | zz
  int n = 0;
  int cap = 0;
  bool have_cov = false;
  bool cov_user = false;   // covariances handed in by the caller (rolo_set_*_covariances): entries not bounded by the regularisation
  // PLANE regularisation only (round 5): the covariance U diag(1, 1, 1e-3) V^T IS I - m m^T with m = sqrt(1 -+ 1e-3) * (third singular vector) — three doubles
  // per point instead of six, and R C R^T = R R^T - (R m)(R m)^T: the LM passes read and rotate THIS when the covariances were computed here (knn_covariance_finish)
  double* nrm = nullptr;   // 3 x n SoA; valid while have_nrm
  size_t nrm_cap = 0;
  bool have_nrm = false;
  // kNN acceleration structure (Morton-ordered implicit BVH)
  float4* sorted = nullptr;     // 8 * n_leaves, (x,y,z, bits(original index)); padding = +inf, index INT_MAX
  float4* boxes = nullptr;      // 2 * 2P entries: node h -> boxes[2h] = lo, boxes[2h+1] = hi ; leaves h in [P, 2P)
  int n_leaves = 0, P = 0;
  bool have_sorted = false;     // sorted / boxes belong to the current xyz
  const int* bbox6 = nullptr;   // bounding box of the current xyz left by the search (device, 6 order-preserving ints); nullptr: unknown
  int* bbox_part = nullptr;     // partial boxes of the current xyz left by the pack kernel: n_bbox_part x 6 order-preserving ints (0: none)
  size_t bbox_part_cap = 0; int n_bbox_part = 0;
  int32_t* knn_idx = nullptr;   // debug: n x k
  float* knn_d2 = nullptr;
};

struct LmState {
  // poses
  double x0_R[9], x0_t[3];  // pose of the current linearisation (x0)
  double xt_R[9], xt_t[3];  // trial pose (xi) evaluated by the next pass
  double tr_R[9];           // rotation of the last reference linearize call: Mahalanobis of the translation stage
  double x0_S[6], xt_S[6], tr_S[6];   // R R^T of the three rotations above (xx xy xz yy yz zz), kept next to them by whoever writes them (lm_set_rrt): the passes rotate a PLANE
                                      // covariance I - m m^T as R R^T - (R m)(R m)^T and would otherwise form the uniform product in every lane of every trial
  double H[36], b[6], y0;
  double lambda, nu;
  double d[6];
  double delta_R[9], delta_t[3];
  double final_H[36];
  // translation stage
  double t0[3], tt[3], g[3], l[3], dtn, dtn1, lam_over_n;
  double lastA_q[3], lastB_q[3], inv_dtn;   // last_transform / dt_{n-1} as compute_t_error / t3_linearize see it, 1 / dt_n: formed once per stage (trans_consts)
  float ct_lambda;
  // control
  int stage;   // 0 idle, 1 rotation / 6-dof, 2 translation
  int phase;   // 0: first pass of the stage pending (linearise only), 1: trial pending
  int outer, trial;
  int cur;     // correspondence buffer that belongs to x0
  int tr_cur;  // correspondence buffer of the last reference linearize call
  int n_corr, tr_n_corr;
  int run_trans;  // start the translation stage from the controller when the rotation stage ends
  int rot_done, rot_converged, rot_failed, rot_outer, rot_passes, rot_ncorr;
  int trans_done, trans_failed, trans_outer, trans_passes;
  int trace_count;
  int error;  // ROLO_E* raised on the device (key range, no correspondences)
  int pending;  // fused LM launches: the rows of the previous launch wait to be summed and stepped on
  // Speculation control of the fused pass (round 5). A pass evaluates a trial's cost (A) AND, betting on acceptance, the linearisation at the trial pose (B). After a
  // REJECTED trial the bet is poor — with 20 forced iterations the rotation stage repeats one rejected-but-converged trial fourteen times, the translation stage ends in
  // six rejections — so the controller marks the next pass lin_skip = 1: (A) only. If that trial is accepted after all, the controller sets phase = 0 and the next pass
  // linearises at the accepted pose: the state after it is the one the full pass would have left, one launch later. spec_lin = 0 (ROLO_LM_SPEC_LIN=0) never skips.
  int lin_skip, spec_lin;
  int rot_cost_only, trans_cost_only;   // passes of rot_passes / trans_passes that ran with lin_skip set (counted by the step that consumes them): rolo_stats::n_cost_only
  int lmp_bailed;   // the resident LM kernel (fused_lm = 2) did not get all its workgroups resident within the admission time and left WITHOUT touching the stage: the host
                    // finishes the frame with pass + controller launches (passes.hip lm_persist_kernel, api.hip run_stage)
  // parameters
  int optimizer, max_iterations, fixed_iterations, lm_max, q2_intended;
  double rot_eps, trans_eps, lm_init;
  double inv_rot_eps, inv_trans_eps;   // 1.0 / eps, formed once per frame (the convergence tests used to divide in every trial)
};

struct PassArgs {
  const float4* src;
  const double* cov;  // 6 x n_total SoA
  const double* nrm;  // 3 x n_total SoA: the PLANE covariances as I - m m^T (CloudDev::nrm), or nullptr: the six-entry covariances are used
  int n_total;        // SoA pitch
  int begin, end;     // shard of source points evaluated by this rank
  int n_off;          // 1, 7 or 27 neighbour offsets
  int* corr[2];       // n_total * n_off voxel ids (-1 = none)
  double* partials;   // gridDim.x x NV_MAX
  int xcd_map;        // 1: XCD x evaluates the x-th eighth of the point blocks (passes.hip pass_xcd_block); 0: blocks dealt round-robin (ROLO_PASS_XCD=0, the A/B)
  VoxelTable tab;
};

// ---- peer exchange (multi-GPU): every rank owns one "mailbox" allocation that all ranks of the node map (hipIpc
// handles between processes, plain pointers inside one process). A rank WRITES its 32 fp64 partial sums of an LM pass into its slot of
// every rank's mailbox and READS only its own mailbox; the sum over the slots is taken in rank order, so every rank gets the same bits
// and takes the same LM decision — no collective library call, no extra launch, and the frame's schedule stays graph-capturable.
// Words are 64-bit {epoch : 32 | half a double : 32} written / polled with relaxed system-scope 64-bit atomics (each word validates
// itself — the LL idea of NCCL — so no fence sits on the path); slots are double-buffered by epoch parity (a rank can run at most one
// exchange ahead of a peer). The same allocation carries the K5 covariance exchange buffer (peer-written, flag per rank).
constexpr int PEER_MAX = 8;                 // ranks of one node
constexpr int PEER_SLOT_WORDS = 2 * NV_MAX; // 64 words of 8 bytes per (parity, rank)
constexpr int PEER_W_LM_EPOCH = 0;          // own counter of LM exchanges done (only the local controller touches it)
constexpr int PEER_W_COV_EPOCH = 1;         // own counter of covariance exchanges done
constexpr int PEER_W_TICKET = 2;            // arrival ticket of the covariance push kernel
constexpr int PEER_W_AREA_BYTES = 3;        // bytes of ONE exchange area of this mailbox (written at export, read by the peers at connect: every rank pushes whole segments here)
constexpr int PEER_W_WORLD = 4;             // the world size this mailbox was exported for
constexpr int PEER_W_COV_FLAG = 8;          // [PEER_MAX] flag of rank r: epoch of the last covariance segment r pushed here
constexpr int PEER_W_SLOTS = 64;            // [2][PEER_MAX][PEER_SLOT_WORDS]
constexpr size_t PEER_STAGE_OFFSET = 16384; // bytes: covariance exchange area behind the header (2 * 8 * 512 B of slots end at 8704)
static_assert((PEER_W_SLOTS + 2 * PEER_MAX * PEER_SLOT_WORDS) * 8 <= PEER_STAGE_OFFSET, "mailbox header");
struct PeerArgs {
  int rank, world;                          // world <= 1: no exchange
  unsigned long long* box[PEER_MAX];        // mailbox of every rank as mapped here (box[rank] = the own one)
  unsigned long long timeout_ticks;         // wall_clock64 ticks (100 MHz) a poll may last before the rank gives up with ROLO_ECOMM
};

// one member of a batch (rolo_batch_*): the arguments of its pass kernels, its LM state and trace, its row count
struct BatchSlot { PassArgs a; LmState* st; rolo_trace_rec* trace; int grid; int pad; };

// ---- launchers (defined in the .hip files) --------------------------------------------------------------
// the clouds one chain of search launches works on (knn_cov.hip): source and target of a registration, or one cloud
// q_begin / q_end: the slice of Morton-sorted query positions this rank searches (multi-GPU: K5 shards by query point, SURVEY 8e;
// whole cloud otherwise). stage != nullptr: the covariances of the slice go to an exchange buffer in sorted order (6 doubles per
// position; position j lives in segment j / chunk at stage + (j / chunk) * seg + stage_off + (j % chunk) * 6) instead of cov[].
struct KnnCloud { const float4* xyz; float4* sorted; float4* boxes; double* cov; double* nrm /* 3 x n SoA or nullptr (CloudDev::nrm) */; int32_t* knn_idx; float* knn_d2; int32_t* nbr; int n, n_leaves, P, n_sorted;
                  int q_begin, q_end; double* stage; int chunk, stage_off; size_t seg;
                  const unsigned long long* stage_epoch; size_t stage_alt;   // peers: stage = area 0; the area of a frame = stage + (exchange number & 1) * stage_alt doubles (knn_walk.hpp stage_area)
                  const int* bpart; int n_bpart;      // partial bounding boxes to fold (the pack kernel's, or bbox_kernel's in the scratch buffer)
                  // k_correspondences > 64: the lists are found 64 at a time — round r keeps the 64 smallest keys ABOVE lower[j], the last key of
                  // round r - 1 (keys (d2, index) are unique per point), and writes slots [slot0, slot0 + 64) of the k_total per query
                  double* lower; int slot0, k_total; };
struct KnnPair { KnnCloud c[2]; int n_clouds; };
// The target's voxel map built inside the search's launches (single GPU, covariances computed here and bounded): the table is cleared by
// extra workgroups of the key kernel, the points are inserted by extra workgroups of the sort-scatter launches (a share each, on
// the CUs the 128 sort tiles leave idle: nothing on the critical path), and the tail accumulates each target point's covariance straight
// from registers, in curve order. Only voxel_finalize_kernel is left of the map build.
struct VoxelFuse {
  int enabled;        // 0: the map is built by launch_voxel_build alone
  int which;          // index of the target cloud in the pair
  VoxelTable tab;
  unsigned long long* tgt_keys; int* tgt_slot; int* counters;
  const int* bbox6;   // the target's final bounding box (written by the key kernel)
  const float4* tgt_xyz; int n_tgt;   // the target in input order (what the insert workgroups read)
};
hipError_t launch_knn_build(const KnnPair& A, void* sort_tmp, size_t sort_tmp_bytes, uint32_t* keys0, uint32_t* keys1,
                            uint32_t* vals0, uint32_t* vals1, int* bbox, const VoxelFuse& vf, hipStream_t s);
size_t knn_sort_temp_bytes(int n_total);
bool knn_voxel_fuse_supported();
size_t knn_bbox_ints();   // ints of the bounding-box buffer: 12 for the final boxes of a pair + the partial boxes behind them
constexpr int KNN_WALK_STACK = 48;
// regularization >= 0: the walk ends in the covariance tail (A.c[].cov / the exchange buffer); -1: neighbour indices -> A.c[].nbr only
// coop_budget > 0 (k = 20, own covariance launch): the cooperative walk — a packet that has scored that many leaves with sub-trees left publishes them to the
// idle wavefronts of its workgroup (knn_walk.hpp); 0: the plain walk, every wavefront for itself
// device_busy: other contexts have frames in flight on this device — large launches then take the 64-query packets (half the wavefronts: shares the chip better) instead of two lanes per query (finishes sooner alone)
// moments (k = 20, plain and sub-lane walks, regularization -1): the walk's epilogue leaves the six centred second moments of every query's neighbourhood in A.c[].cov
// (SoA, by original index) instead of the neighbour indices in A.c[].nbr; launch_knn_tail(..., moments = true) finishes them in place
hipError_t launch_knn_walk(const KnnPair& A, int k, int regularization_or_minus1, const VoxelFuse& vf, hipStream_t s, int coop_budget = 0, int* lanes_out = nullptr, bool device_busy = false,
                           bool moments = false);
hipError_t launch_knn_tail(const KnnPair& A, int k, int regularization, const VoxelFuse& vf, hipStream_t s, bool moments = false);   // covariances from A.c[].nbr (or from the moments in A.c[].cov)
// multi-GPU: exchange buffer (sorted order, all ranks' slices after the all-gather, or [q_begin, q_end) only) -> cov[] by original index
// vf.enabled: the target's points are accumulated into the voxel map by this scatter (sharded VoxelFuse)
hipError_t launch_knn_unstage(const KnnPair& A, bool own_slice_only, const VoxelFuse& vf, hipStream_t s);

// fixed_cov: the covariance sums go through 64-bit fixed point as the positions always do (entries bounded by 1); false: fp64 atomics
// bbox6: the target's bounding box as the neighbour search leaves it on the device (6 order-preserving ints), or nullptr
// prefused: clear / insert / accumulate already ran inside the search's launches (VoxelFuse): finalize only
hipError_t launch_voxel_build(const CloudDev& tgt, VoxelTable tab, unsigned long long* tgt_keys, int* tgt_slot, int* counters, bool morton_order, bool fixed_cov,
                              const int* bbox6, bool prefused, hipStream_t s, int* pub_counters = nullptr /* pinned host copy of the 4 counters, written by the finalize kernel */,
                              struct LmState* lm_state = nullptr, const struct FrameArgs* lm_args = nullptr /* the finalize kernel also starts the frame's LM state (lm_begin.hpp) */);
hipError_t launch_stamp(unsigned long long* buf, int slot, hipStream_t s);   // debug timeline
hipError_t launch_voxel_keys(const float4* pts, int n, VoxelTable tab, int32_t* keys3, hipStream_t s);

hipError_t launch_rot_pass(int dof, const PassArgs& a, const LmState* st, int grid, hipStream_t s);
hipError_t launch_trans_pass(const PassArgs& a, const LmState* st, int grid, hipStream_t s);
// fused trial (passes.hip lm_kernel): finish the pending trial (rows_in, st_in), write the new state to st_out (!= st_in unless grid 1),
// evaluate the next pass into rows_out; threads in {256, 512, 1024}; nrows = workgroups of a pass; do_body = 0: closing launch
hipError_t launch_lm(int dof, int threads, int ppt /* slabs of `threads` points per workgroup */, const PassArgs& a, const LmState* st_in, LmState* st_out, const double* rows_in, double* rows_out, int nrows,
                     rolo_trace_rec* trace, int do_body, hipStream_t s, LmState* pub = nullptr /* pinned host copy of the state written by workgroup 0 */);
// one launch per frame (passes.hip lm_persist_kernel): nrows resident workgroups of 512 threads x ppt points, rows exchanged through xbuf (lm_persist_words(nrows) 64-bit words,
// zeroed once); the state starts and ends in st; admit_ticks: wall-clock ticks (100 MHz) the workgroups wait for each other to become resident before they leave the stage to the
// host (LmState::lmp_bailed); timeout_ticks: what a poll may last after that (a bug guard: ROLO_ECOMM); max_trials: hard cap on the trials of one launch
hipError_t launch_lm_persist(int dof, int threads, int ppt, const PassArgs& a, LmState* st, unsigned long long* xbuf, int nrows, rolo_trace_rec* trace, LmState* pub, unsigned long long timeout_ticks,
                             unsigned long long admit_ticks, int max_trials, hipStream_t s);
size_t lm_persist_words(int nrows);
hipError_t launch_reduce(const double* partials, int nblocks, double* sums, const LmState* st, int stage, hipStream_t s);
// controller: sums the rows of `partials` itself (single GPU) or takes all-reduced `sums` (partials == nullptr)
// pub != nullptr: also leave the state in that (pinned host) copy, step or no step
// peer != nullptr (world > 1): the row sums go through the peer exchange before the step (every rank sums all ranks' values in rank order)
// dof: 3 / 6 = degrees of freedom of the rotation-stage optimiser (picks the specialised controller), 0 = unknown (generic kernel)
hipError_t launch_ctrl(LmState* st, const double* partials, int nblocks, const double* sums, rolo_trace_rec* trace, int stage, hipStream_t s, LmState* pub = nullptr,
                       const PeerArgs* peer = nullptr, int dof = 0);
// peer.hip: sums[NV_MAX] (device) <- sum over the ranks, in rank order (stage-level evaluations); *err_flag (device int) is set on a timeout
hipError_t launch_peer_allreduce(double* sums, const PeerArgs& peer, int* err_flag, hipStream_t s);
// peer.hip: covariance exchange of one frame — push the own segment [rank * seg_doubles, +seg_doubles) of the local exchange area into every
// peer's area and raise the own flag there (last workgroup), then wait (one wavefront) for every rank's flag of this epoch. Two areas of
// area_bytes each alternate behind PEER_STAGE_OFFSET; the kernels pick the area from the exchange's number (own epoch word + 1) on the device
hipError_t launch_peer_cov_exchange(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, int* err_flag, hipStream_t s);

// rolo_peer_selftest: fill the own segment of the next exchange's area with known words / count the wrong words per rank after the exchange (bad: PEER_MAX device counters, zeroed by the caller)
hipError_t launch_peer_selftest_fill(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, hipStream_t s);
hipError_t launch_peer_selftest_check(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, unsigned* bad, hipStream_t s);

struct RotBegin { double R[9], t[3]; int optimizer, max_iterations, fixed_iterations, lm_max, q2_intended; double rot_eps, trans_eps, lm_init; int run_trans; int spec_lin; };
struct TransBegin { double t0[3], g[3], l[3], dtn, dtn1; float ct_lambda; int direct; /* 1: start now (rotation already done) */
                    // the knobs computeTranslation reads when IT runs (lsq_registration_impl.hpp:63, :98, :142-148 read the members live): a setter called between align and
                    // computeTranslation counts — until round 6 the stage ran on what the last align had copied into the state
                    int max_iterations, lm_max, q2_intended; double trans_eps, lm_init; };
struct FrameArgs { RotBegin rot; TransBegin trans; };
hipError_t launch_rot_begin(LmState* st, const RotBegin& a, hipStream_t s);
hipError_t launch_frame_begin(LmState* st, const FrameArgs* a, hipStream_t s);
hipError_t launch_batch_pass(int stage, int dof, const BatchSlot* slots, int n_slots, int bps, hipStream_t s);
hipError_t launch_batch_ctrl(int stage, const BatchSlot* slots, int n_slots, hipStream_t s);
hipError_t launch_batch_begin(const BatchSlot* slots, const FrameArgs* args, int n_slots, hipStream_t s);
hipError_t launch_trans_begin(LmState* st, const TransBegin& a, hipStream_t s);
// single evaluations for the stage-level API (rolo_so3_linearize, rolo_compute_error, rolo_t3_linearize, ...)
hipError_t launch_eval_begin(LmState* st, const RotBegin& a, int mode, hipStream_t s);
hipError_t launch_eval_end(LmState* st, const double* sums, int mode, hipStream_t s);
hipError_t launch_t3_eval_begin(LmState* st, const TransBegin& a, int phase, hipStream_t s);

hipError_t launch_transform_cloud(const float* in, float* out, int n, int stride, const float* T16_dev_or_null,
                                  const float* T16_host, hipStream_t s);
// bbox_part != nullptr: one partial bounding box (6 order-preserving ints) per workgroup of 256 points goes there
hipError_t launch_pack_xyz(const float* in, int stride, float4* out, int n, hipStream_t s, int* bbox_part = nullptr);
// source (moved by the row-major 4x4 float transform T16 on the way, or nullptr) and target packed by one launch, partial boxes as launch_pack_xyz leaves them
hipError_t launch_pack_pair(const float* in0, int stride0, float4* out0, int n0, int* bbox0, const float* T16_host_or_null,
                            const float* in1, int stride1, float4* out1, int n1, int* bbox1, hipStream_t s);
hipError_t launch_empty(int grid, int threads, hipStream_t s);   // rolo_debug_chain: a launch that does nothing
hipError_t launch_cov_unpack(const double* soa, int n, double* m16, hipStream_t s);   // 6 SoA -> n x 16
hipError_t launch_cov_pack(const double* m16, int n, double* soa, hipStream_t s);     // n x 16 -> 6 SoA

// front.hip: fused K1-K4 for the device-resident pipeline — raw frame -> corner ++ surface in d_feat; counts3 = N, n_corner, n_surface;
// asynchronous on the context's stream (event `done` behind the read-back of the counts)
size_t front_feature_capacity(const rolo_front_params* P);
int front_frame_features_from_msg(rolo_ctx* c, const rolo_front_params* P, const unsigned char* data, const rolo_cloud_layout* L, int n_raw,
                                  bool on_device, float4* d_feat, int* h_counts3_pinned, hipEvent_t done);
int front_frame_features_enqueue(rolo_ctx* c, const rolo_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                                 bool on_device, float4* d_feat, int* h_counts3_pinned, hipEvent_t done);

}  // namespace rolo

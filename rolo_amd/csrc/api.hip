// Host side of librolo_hip.so: the C ABI of include/rolo_hip.h. Owns the device buffers of one context (one
// fast_gicp::RotVGICP instance in the reference's terms), enqueues the kernels of knn_cov.hip / voxelmap.hip /
// passes.hip on the context's stream and mirrors the call semantics of the reference class
// (include/rot_gicp/gicp/rot_vgicp.hpp:72-104, impl/rot_vgicp_impl.hpp:20-169, impl/lsq_registration_impl.hpp:55-80,
// 152-179). No CPU fallback exists: every entry point fails with ROLO_EHIP if the HIP runtime does.
#include "rolo_internal.hpp"
#include "load_learner.hpp"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <array>
#include <map>
#include <mutex>
#include <chrono>

using namespace rolo;

namespace {

thread_local std::string g_err;

int fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return ROLO_EHIP;
}
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail_hip(_e, #x); } while (0)

std::atomic<unsigned long long> g_alloc_epoch{0};  // bumped on every (re)allocation: captured graphs hold raw device pointers

template <typename T>
int ensure(T*& p, size_t& cap, size_t need) {
  if (need <= cap && p) return ROLO_OK;
  g_alloc_epoch++;
  if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return fail_hip(e, "hipFree"); p = nullptr; }
  size_t ncap = std::max<size_t>(need + need / 4, 1024);
  hipError_t e = hipMalloc((void**)&p, ncap * sizeof(T));
  if (e != hipSuccess) { cap = 0; return fail_hip(e, "hipMalloc"); }
  cap = ncap;
  return ROLO_OK;
}

// ---- RCCL through dlopen (only multi-GPU runs need it) ----
struct Uid { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ Uid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return ROLO_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { g_err = std::string("dlopen librccl: ") + dlerror(); return ROLO_ECOMM; }
  g_rccl.lib = h;
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
  g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy) { g_err = "librccl: missing symbols"; return ROLO_ECOMM; }
  return ROLO_OK;
}
constexpr int NCCL_FLOAT64 = 8;  // ncclDouble
constexpr int NCCL_SUM = 0;

}  // namespace

extern "C" void rolo_shard_range(int n, int rank, int world, int* begin, int* end) {
  const long long N = n;
  if (begin) *begin = (int)(N * rank / world);
  if (end) *end = (int)(N * (rank + 1) / world);
}

// ---- peer exchange: process-local registry of exported mailboxes (two contexts of ONE process must not go through hipIpcOpenMemHandle:
// a handle cannot be opened by the process that exported it) ----
namespace {
struct PeerExport { void* base; int device; };
std::mutex g_peer_mu;
std::map<std::array<char, ROLO_PEER_HANDLE_BYTES>, PeerExport> g_peer_exports;
}  // namespace

struct rolo_peer_state {
  void* base = nullptr;          // own mailbox + the two covariance exchange areas (one allocation, exported)
  size_t bytes = 0, area_bytes = 0;
  std::array<char, ROLO_PEER_HANDLE_BYTES> handle{};
  int export_world = 0;
  bool connected = false;
  void* mapped[PEER_MAX] = {};   // every rank's mailbox as mapped here
  bool ipc_opened[PEER_MAX] = {};
  PeerArgs args{};
  int* h_err = nullptr;          // pinned: ROLO_ECOMM written by a kernel whose poll timed out
  const char* mem_kind = "";
};

struct rolo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  rolo_params P;
  CloudDev src, tgt;
  size_t src_xyz_cap = 0, src_cov_cap = 0, tgt_xyz_cap = 0, tgt_cov_cap = 0;
  size_t src_sorted_cap = 0, src_boxes_cap = 0, tgt_sorted_cap = 0, tgt_boxes_cap = 0;
  size_t src_knn_cap = 0, src_knnd_cap = 0, tgt_knn_cap = 0, tgt_knnd_cap = 0;
  bool want_knn_lists = false;
  // kNN scratch
  // scratch set 0 serves a pair search (or a lone source), set 1 a lone target — a batch's eager path runs the two on two streams
  struct KnnScratch {
    char* sort_tmp = nullptr; size_t sort_tmp_cap = 0;
    uint32_t *keys0 = nullptr, *keys1 = nullptr, *vals0 = nullptr, *vals1 = nullptr;
    size_t keys0_cap = 0, keys1_cap = 0, vals0_cap = 0, vals1_cap = 0;
    int* bbox = nullptr; size_t bbox_cap = 0;
    int32_t* nbr = nullptr; size_t nbr_cap = 0;   // neighbour indices between the walk and the covariance kernel
    double* stage = nullptr; size_t stage_cap = 0;  // multi-GPU: covariance exchange buffer (sorted order, one segment per rank)
    double* lower = nullptr; size_t lower_cap = 0;  // k_correspondences > 64: the key the next round of 64 starts above, per sorted position
  } ks[2];
  hipEvent_t ev_done = nullptr;    // end of the frame rolo_register_async enqueued (the stream may carry other contexts' frames behind it)
  hipEvent_t ev_start = nullptr;   // its start (both with timing: the frame's duration on the DEVICE is the load signal of LoadLearner below)
  rolo::LoadLearner learn;     // load this process cannot count (another process on the GPU), learned from the frames' device time: load_learner.hpp
  bool frame_auto_idle = false;   // the frame in flight was sized by the learner (load_hint < 0, nobody else of this process in flight, not sharded)
  hipStream_t stream2 = nullptr;   // second stream for the eager (uncaptured) path of rolo_batch_*; stream and stream2 are a PAIR of the device's stream bank
                                   // (below): main streams on every other stream of a burst = two hardware queues, alternating — measured the best layout for
                                   // frames of several contexts in flight on MI355X (2930 scans/s; one queue per context 2140, three
                                   // contexts on three queues 2620, GPU_MAX_HW_QUEUES=8 1310)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int bank_slot = -1;              // >= 0: stream / stream2 belong to the device's stream bank (given back, not destroyed)
  // voxel map
  VoxelTable tab{};
  size_t tab_keys_cap = 0, tab_rec_cap = 0, tab_idk_cap = 0;
  unsigned long long* tgt_keys = nullptr; size_t tgt_keys_cap = 0;
  int* tgt_slot = nullptr; size_t tgt_slot_cap = 0;
  int* counters = nullptr; size_t counters_cap = 0;
  bool have_map = false;
  int n_voxels = 0;
  unsigned long long* stamps = nullptr; size_t stamps_cap = 0; unsigned long long* h_stamps = nullptr;   // ROLO_STAMP=1 debug timeline (pinned)
  VoxelFuse vf{};           // enqueue_frame arms it before the search when the map can be built inside the search's launches
  bool vf_done = false;     // the search just enqueued did carry the map build
  int n_edge = 0;   // target points of the last map build within 1e-12 of a POLAR bin edge
  // passes
  int* corr[2] = {nullptr, nullptr}; size_t corr_cap[2] = {0, 0};
  double* partials = nullptr; size_t partials_cap = 0;
  int lm_rows = 1;   // workgroups (= partial rows) of one fused LM launch
  unsigned long long* xbuf = nullptr; size_t xbuf_cap = 0;   // row exchange of the resident LM kernel (fused_lm = 2): header + 2 parities x workgroups x 64 words, zeroed when (re)allocated
  int lmp_rows = 1, lmp_ppt = 1, lmp_threads = 512;   // its grid, the points per thread and the workgroup size
  double* sums = nullptr; size_t sums_cap = 0;
  LmState* state = nullptr; size_t state_cap = 0;
  rolo_trace_rec* trace = nullptr; size_t trace_cap = 0;
  bool have_corr = false;
  // staging
  float* stage_in = nullptr; size_t stage_in_cap = 0;
  float* stage_out = nullptr; size_t stage_out_cap = 0;
  double* stage_d = nullptr; size_t stage_d_cap = 0;
  int32_t* stage_i = nullptr; size_t stage_i_cap = 0;
  // pinned host mirrors
  LmState* h_state = nullptr;
  double* h_sums = nullptr;
  int* h_counters = nullptr;
  // multi-GPU
  void* comm = nullptr;
  rolo_peer_state peer;     // rolo_peer_*: the exchange without a collective library
  int rank = 0, world = 1;
  bool shard_knn = false;   // rolo_set_shard_knn: K5 by query slice without a communicator (test hook)
  // async registration bookkeeping
  bool async_pending = false;
  long long n_frames = 0, n_replays = 0, n_captures = 0, n_eager = 0, n_topup_frames = 0, n_topup_chunks = 0;   // rolo_ctx_counters
  long long n_persist_bails = 0;   // frames whose resident LM kernel gave the stage back to the host (rolo_ctx_counters [12])
  long long ns_enqueue = 0, ns_wait_blocked = 0, ns_wait_other = 0;   // host time inside rolo_register_async / the event wait / the rest of rolo_register_wait (steady_clock)
  hipGraphExec_t dbg_chain_exec = nullptr; int dbg_chain_key[3] = {-1, -1, -1};   // rolo_debug_chain: the captured chain and its (kind, n_pairs, grid)
  // hipGraph of one whole frame (rolo_register_async): captured on the second frame with an unchanged key, replayed after
  FrameArgs* h_args = nullptr;   // pinned; a captured H2D copy refreshes d_args on every replay
  FrameArgs* d_args = nullptr; size_t d_args_cap = 0;
  struct GraphKey { int n_src, n_tgt; const void *src_xyz, *tgt_xyz; rolo_params P; unsigned long long epoch; int nrot, ntrans, rank, world, busy; } gkey{}, gseen{};
  bool device_busy = false;   // other contexts of this device had frames in flight when this frame was enqueued (picks the walk kernel of large launches: knn_cov.hip launch_knn_walk)
  bool counted_in_flight = false;
  int load_hint = -1;         // rolo_set_load_hint: -1 per frame from the device's load, 0 / 1 pinned
  int busy_credit = 0;        // frames this context keeps the busy-device choice after it last saw other frames in flight (a frame enqueued right after a caller's
                              // barrier would otherwise flip the choice — and with it the captured hipGraph — once per round of a multi-context loop)
  // passes the last frames needed per stage (update_hint): the next frame enqueues that many predicated pass/controller pairs up
  // front instead of a fixed worst-case chunk; rolo_register_wait tops up if a frame needs more
  int hint_rot = 0, hint_trans = 0;
  int walk_lanes = 1;   // lanes per query of the last K5 walk enqueued (rolo_ctx_counters [8])
  struct NeedWindow { int need[64] = {0}; int n = 0, pos = 0; } win_rot, win_trans;   // passes the last 64 frames needed per stage
  bool gseen_valid = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool graph_nrm_written = false;   // the captured build_clouds left the PLANE covariances as I - m m^T too (CloudDev::have_nrm after a replay)
  // the captured frame of the OTHER load regime (GraphKey::busy): a context whose load estimate flips — the learner's periodic second look, a second context that comes and
  // goes — swaps its two graphs instead of capturing again (42 captures in 11 000 frames of the two-process run before this slot existed)
  struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; GraphKey key{}; bool nrm = false; } galt;
  // per-kernel event timing (rolo_prof_*)
  bool prof_on = false;
  struct ProfEv { int slot; hipEvent_t a, b; };
  std::vector<ProfEv> prof;
  // front end (front.hip)
  void* front = nullptr;
  void* s2m = nullptr;   // scan2map.hip scratch
  unsigned long long cloud_epoch = 0;   // bumped whenever the source / target clouds (or their buffers) change hands: the resident sub-map of rolo_scan2map_set_submap lives in them
};

namespace {

// brackets the launches issued during its lifetime with a HIP event pair on the context's stream
struct ProfScope {
  rolo_ctx* c; int idx = -1; hipStream_t s;
  ProfScope(rolo_ctx* ctx, int slot, hipStream_t stream = nullptr) : c(ctx), s(stream ? stream : ctx->stream) {
    if (!c->prof_on) return;
    rolo_ctx::ProfEv e{slot, nullptr, nullptr};
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
    (void)hipEventRecord(e.a, s);
    c->prof.push_back(e);
    idx = (int)c->prof.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof[idx].b, s); }
};

inline bool peers(const rolo_ctx* c) { return c->peer.connected && c->peer.args.world > 1; }
inline const PeerArgs* peer_args(const rolo_ctx* c) { return peers(c) ? &c->peer.args : nullptr; }

// a kernel's poll of the peers' words timed out (it left ROLO_ECOMM in pinned memory): report it once the host has synchronised
inline int peer_check(rolo_ctx* c) {
  if (c->peer.h_err && *c->peer.h_err != 0) { g_err = "peer exchange timed out (a rank of the node did not answer)"; return ROLO_ECOMM; }
  return ROLO_OK;
}

inline int n_offsets(const rolo_params& P) { return P.neighbor_search == ROLO_DIRECT1 ? 1 : (P.neighbor_search == ROLO_DIRECT7 ? 7 : 27); }

int set_device(rolo_ctx* c) { HIPCHK(hipSetDevice(c->device)); return ROLO_OK; }

int upload_cloud(rolo_ctx* c, CloudDev& cl, size_t& xyz_cap, const float* pts, int n, int stride, bool on_device) {
  if (n < 0 || stride < 3 || (n > 0 && !pts)) { g_err = "bad cloud arguments"; return ROLO_EINVAL; }
  int rc = ensure(cl.xyz, xyz_cap, (size_t)std::max(n, 1));
  if (rc) return rc;
  const float* dsrc = pts;
  if (!on_device && n > 0) {
    rc = ensure(c->stage_in, c->stage_in_cap, (size_t)n * stride);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->stage_in, pts, sizeof(float) * (size_t)n * stride, hipMemcpyHostToDevice, c->stream));
    dsrc = c->stage_in;
  }
  const int nparts = (n + 255) / 256;
  if ((rc = ensure(cl.bbox_part, cl.bbox_part_cap, (size_t)std::max(nparts, 1) * 6))) return rc;
  HIPCHK(launch_pack_xyz(dsrc, stride, cl.xyz, n, c->stream, cl.bbox_part));
  cl.n_bbox_part = nparts;
  cl.n = n;
  cl.have_cov = false; cl.have_nrm = false;
  cl.have_sorted = false;
  cl.bbox6 = nullptr;
  c->cloud_epoch++;
  return ROLO_OK;
}

// search structures of one cloud: geometry + allocations
int prepare_cloud(rolo_ctx* c, CloudDev& cl, size_t& cov_cap, size_t& sorted_cap, size_t& boxes_cap, size_t& knn_cap, size_t& knnd_cap, KnnCloud& out, bool tree_only = false) {
  const int n = cl.n, k = tree_only ? 1 : c->P.k_correspondences;
  if (k < 1) { g_err = "k_correspondences must be positive"; return ROLO_EINVAL; }
  if (n < k) { g_err = "cloud has fewer points than k_correspondences"; return ROLO_ETOOFEW; }
  cl.n_leaves = (n + KNN_LEAF - 1) / KNN_LEAF;
  int P = 2; while (P < cl.n_leaves) P <<= 1;
  cl.P = P;
  int rc;
  if ((rc = ensure(cl.cov, cov_cap, 6 * (size_t)n))) return rc;
  if (!tree_only && (rc = ensure(cl.nrm, cl.nrm_cap, 3 * (size_t)n))) return rc;
  if ((rc = ensure(cl.sorted, sorted_cap, KNN_LEAF * (size_t)cl.n_leaves))) return rc;
  if ((rc = ensure(cl.boxes, boxes_cap, 4 * (size_t)P))) return rc;
  if (c->want_knn_lists) {
    if ((rc = ensure(cl.knn_idx, knn_cap, (size_t)n * k))) return rc;
    if ((rc = ensure(cl.knn_d2, knnd_cap, (size_t)n * k))) return rc;
  }
  out.xyz = cl.xyz; out.sorted = cl.sorted; out.boxes = cl.boxes; out.cov = cl.cov; out.nrm = tree_only ? nullptr : cl.nrm;
  out.knn_idx = c->want_knn_lists ? cl.knn_idx : nullptr; out.knn_d2 = c->want_knn_lists ? cl.knn_d2 : nullptr;
  out.n = n; out.n_leaves = cl.n_leaves; out.P = P; out.n_sorted = KNN_LEAF * cl.n_leaves;
  out.q_begin = 0; out.q_end = out.n_sorted; out.stage = nullptr; out.chunk = out.n_sorted; out.stage_off = 0; out.seg = 0; out.stage_epoch = nullptr; out.stage_alt = 0;
  out.bpart = cl.n_bbox_part > 0 ? cl.bbox_part : nullptr; out.n_bpart = cl.n_bbox_part;
  return ROLO_OK;
}

// Morton sort, BVH, neighbour search and covariances of the source and / or the target in ONE chain of launches.
// A pair shares the scratch set 0; a lone target uses set 1 so that it can run next to a lone source on another stream.
// ROLO_KNN_FUSE_TAIL=1: the covariance tail inside the walk kernel instead of its own launch (an A/B: slower, see knn_walk.hpp)
static bool fused_tail_env() {
  static const bool v = [] { const char* e = getenv("ROLO_KNN_FUSE_TAIL"); return e && atoi(e) != 0; }();
  return v;
}
// ROLO_KNN_BUDGET=<leaves>: the cooperative walk (knn_walk.hpp) — after that many leaves a packet starts handing sub-trees to the idle wavefronts of its workgroup.
// Default 0 = the plain walk: measured in round 4, the cooperative form is exact (all list tests pass with it) but no faster (0.224 against 0.223 ms over the
// pool at 24 leaves): what thieves can take are the far sub-trees at the bottom of a stack, which prune to nothing, and the kernel is bound by the spread of the
// work over the CUs (x 1.21 - 1.34 between the busiest CU and the mean), which nothing inside a workgroup can move.
static int knn_budget_env() {
  static const int v = [] { const char* e = getenv("ROLO_KNN_BUDGET"); const int b = e ? atoi(e) : 0; return b < 0 ? 0 : b; }();
  return v;
}
// ROLO_VOXEL_FUSE=0: the voxel map as its own launches after the search (the A/B of VoxelFuse)
static bool voxel_fuse_env() {
  static const bool v = [] { const char* e = getenv("ROLO_VOXEL_FUSE"); return !(e && atoi(e) == 0); }();
  return v;
}

int build_clouds(rolo_ctx* c, bool do_src, bool do_tgt, hipStream_t stream, bool tree_only = false, KnnPair* out_pair = nullptr) {
  KnnPair A{};
  int rc, nc = 0;
  if (do_src) { if ((rc = prepare_cloud(c, c->src, c->src_cov_cap, c->src_sorted_cap, c->src_boxes_cap, c->src_knn_cap, c->src_knnd_cap, A.c[nc], tree_only))) return rc; nc++; }
  if (do_tgt) { if ((rc = prepare_cloud(c, c->tgt, c->tgt_cov_cap, c->tgt_sorted_cap, c->tgt_boxes_cap, c->tgt_knn_cap, c->tgt_knnd_cap, A.c[nc], tree_only))) return rc; nc++; }
  if (nc == 0) return ROLO_OK;
  A.n_clouds = nc;
  const size_t n_total = (size_t)A.c[0].n + (nc > 1 ? (size_t)A.c[1].n : 0);
  rolo_ctx::KnnScratch& S = c->ks[(do_tgt && !do_src) ? 1 : 0];
  if ((rc = ensure(S.keys0, S.keys0_cap, n_total))) return rc;
  if ((rc = ensure(S.keys1, S.keys1_cap, n_total))) return rc;
  if ((rc = ensure(S.vals0, S.vals0_cap, n_total))) return rc;
  if ((rc = ensure(S.vals1, S.vals1_cap, n_total))) return rc;
  if ((rc = ensure(S.bbox, S.bbox_cap, knn_bbox_ints()))) return rc;
  const int kc = tree_only ? 1 : c->P.k_correspondences;
  const size_t kslots = kc > 64 ? 64 * (((size_t)kc + 63) / 64) : (kc > 32 ? 64 : 32);   // slot-major neighbour lists: KMAX slots per sorted position (k > 64: rounds of 64)
  if ((rc = ensure(S.nbr, S.nbr_cap, kslots * ((size_t)A.c[0].n_sorted + (nc > 1 ? (size_t)A.c[1].n_sorted : 0))))) return rc;
  A.c[0].nbr = S.nbr; if (nc > 1) A.c[1].nbr = S.nbr + kslots * (size_t)A.c[0].n_sorted;
  for (int i = 0; i < 2; i++) { A.c[i].lower = nullptr; A.c[i].slot0 = 0; A.c[i].k_total = 0; }
  if (kc > 64) {   // one key per sorted position: where the next round of 64 starts
    if ((rc = ensure(S.lower, S.lower_cap, (size_t)A.c[0].n_sorted + (nc > 1 ? (size_t)A.c[1].n_sorted : 0)))) return rc;
    A.c[0].lower = S.lower; if (nc > 1) A.c[1].lower = S.lower + A.c[0].n_sorted;
  }
  const size_t tmp = knn_sort_temp_bytes((int)n_total);
  if ((rc = ensure(S.sort_tmp, S.sort_tmp_cap, tmp + 256))) return rc;
  // Multi-GPU ("K5 shards by query point with the full cloud replicated"): every rank sorts and builds the BVH of the whole
  // cloud (cheap, identical on all ranks), searches only its slice of the Morton-sorted queries — whole 256-query workgroups, equal
  // slices — and the 48-byte covariances are all-gathered once per frame in sorted order, then scattered to cov[] by original index.
  const bool sharded = c->comm != nullptr || peers(c) || (c->world > 1 && c->shard_knn);
  if (sharded) {
    size_t seg = 0;
    for (int i = 0; i < nc; i++) {
      KnnCloud& K = A.c[i];
      const int nb = (K.n_sorted + 255) / 256;
      K.chunk = ((nb + c->world - 1) / c->world) * 256;
      K.q_begin = std::min(c->rank * K.chunk, K.n_sorted);
      K.q_end = std::min((c->rank + 1) * K.chunk, K.n_sorted);
      K.stage_off = (int)seg;
      seg += (size_t)K.chunk * 6;
    }
    double* stage;
    const unsigned long long* stage_epoch = nullptr; size_t stage_alt = 0;
    if (peers(c)) {   // the exchange area the peers write into: inside the exported mailbox allocation, two areas alternating by the parity of the
                      // exchange's number — which the kernels read from the mailbox's epoch word ON THE DEVICE: a frame replayed from its hipGraph, an eager
                      // frame and a frame of a rank that captured earlier or later all agree (a host-side counter baked into a capture did not)
      if (seg * (size_t)c->world * sizeof(double) > c->peer.area_bytes) { g_err = "peer exchange area too small for this frame: rolo_peer_export with a larger max_points"; return ROLO_EINVAL; }
      stage = reinterpret_cast<double*>(static_cast<char*>(c->peer.base) + PEER_STAGE_OFFSET);
      stage_epoch = static_cast<const unsigned long long*>(c->peer.base) + PEER_W_COV_EPOCH; stage_alt = c->peer.area_bytes / sizeof(double);
    } else {
      if ((rc = ensure(S.stage, S.stage_cap, seg * (size_t)c->world))) return rc;
      stage = S.stage;
    }
    for (int i = 0; i < nc; i++) { A.c[i].seg = seg; A.c[i].stage = stage; A.c[i].stage_epoch = stage_epoch; A.c[i].stage_alt = stage_alt; }
  }
  // the build overwrites this scratch set's bounding boxes: whoever still pointed at them (a cloud searched earlier) loses them
  if (c->src.bbox6 >= S.bbox && c->src.bbox6 < S.bbox + 12) c->src.bbox6 = nullptr;
  if (c->tgt.bbox6 >= S.bbox && c->tgt.bbox6 < S.bbox + 12) c->tgt.bbox6 = nullptr;
  VoxelFuse vf = c->vf;
  // sharded with every rank's covariances exchanged (peers / RCCL): the map is still built inside the search's launches — clear and insert ride
  // on the key / sort launches each rank runs on the whole cloud anyway, the accumulation moves from the tail to the scatter after the exchange
  const bool all_ranks = c->comm != nullptr || peers(c);
  vf.enabled = vf.enabled && do_tgt && (!sharded || all_ranks) && !tree_only && !fused_tail_env();
  if (vf.enabled) { vf.which = do_src ? 1 : 0; vf.bbox6 = S.bbox + 6 * vf.which; vf.tgt_xyz = c->tgt.xyz; vf.n_tgt = c->tgt.n; }
  c->vf_done = vf.enabled != 0;
  { ProfScope ps(c, ROLO_PROF_KNN_BUILD, stream); HIPCHK(launch_knn_build(A, S.sort_tmp, tmp, S.keys0, S.keys1, S.vals0, S.vals1, S.bbox, vf, stream)); }
  if (out_pair) *out_pair = A;
  if (tree_only) {   // Hilbert sort + BVH only (scan-to-submap association searches it with foreign queries); no covariances
    if (do_src) { c->src.have_sorted = true; c->src.have_cov = false; c->src.bbox6 = S.bbox; }
    if (do_tgt) { c->tgt.have_sorted = true; c->tgt.have_cov = false; c->tgt.bbox6 = S.bbox + (do_src ? 6 : 0); }
    return ROLO_OK;
  }
  const bool split_tail = !fused_tail_env() || kc > 64;
  // default (round 6, k = 20): the walk's epilogue leaves the six centred moments of every neighbourhood, by sorted position, where the index lists used to go, and the tail
  // finishes them (knn_walk.hpp walk_write_moments); ROLO_KNN_MOMENTS=0: the neighbour indices through A.c[].nbr and the tail's own gather (rounds 1-5, the A/B)
  static const bool moments_on = [] { const char* e = getenv("ROLO_KNN_MOMENTS"); return !(e && atoi(e) == 0); }();
  const bool moments = moments_on && kc == 20 && split_tail && knn_budget_env() == 0;
  { ProfScope ps(c, ROLO_PROF_KNN_WALK, stream); HIPCHK(launch_knn_walk(A, c->P.k_correspondences, split_tail ? -1 : c->P.regularization, vf, stream, (kc == 20 && split_tail) ? knn_budget_env() : 0, &c->walk_lanes, c->device_busy, moments)); }
  if (split_tail) { ProfScope ps(c, ROLO_PROF_KNN_TAIL, stream); HIPCHK(launch_knn_tail(A, c->P.k_correspondences, c->P.regularization, vf, stream, moments)); }
  if (sharded) {
    const size_t seg = A.c[0].seg;
    if (peers(c)) {   // every rank pushes its segment into every peer's area, flags, and waits for the others' flags (peer.hip)
      HIPCHK(launch_peer_cov_exchange(c->peer.args, c->peer.area_bytes, seg, c->peer.h_err, stream));
    } else if (c->comm) {
      int e = g_rccl.AllGather(S.stage + (size_t)c->rank * seg, S.stage, seg, NCCL_FLOAT64, c->comm, stream);
      if (e != 0) { g_err = std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?"); return ROLO_ECOMM; }
    }
    // without a communicator / peers (rolo_set_shard test hook) only the own slice is valid afterwards
    HIPCHK(launch_knn_unstage(A, c->comm == nullptr && !peers(c), vf, stream));
  }
  // (the tail leaves the PLANE covariances as I - m m^T too — CloudDev::nrm — unless the slices went through an exchange buffer: passes.hip load_pt)
  const bool nrm_written = !sharded && c->P.regularization == ROLO_REG_PLANE;
  if (do_src) { c->src.have_cov = true; c->src.have_sorted = true; c->src.cov_user = false; c->src.have_nrm = nrm_written; c->src.bbox6 = S.bbox; }
  if (do_tgt) { c->tgt.have_cov = true; c->tgt.have_sorted = true; c->tgt.cov_user = false; c->tgt.have_nrm = nrm_written; c->tgt.bbox6 = S.bbox + (do_src ? 6 : 0); }
  return ROLO_OK;
}

int build_src(rolo_ctx* c, hipStream_t s) { return build_clouds(c, true, false, s); }
int build_tgt(rolo_ctx* c, hipStream_t s) { return build_clouds(c, false, true, s); }

int ensure_covs(rolo_ctx* c) {
  if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "source/target not set"; return ROLO_ESTATE; }
  int rc;
  // overlap_knn: both clouds go through one chain of launches (every kernel works on the pair), so the two searches
  // start together; off = one chain per cloud, back to back
  if (!c->src.have_cov && !c->tgt.have_cov && c->P.overlap_knn) return build_clouds(c, true, true, c->stream);
  if (!c->src.have_cov && (rc = build_src(c, c->stream))) return rc;
  if (!c->tgt.have_cov && (rc = build_tgt(c, c->stream))) return rc;
  return ROLO_OK;
}

// Morton-ordered voxel build (voxelmap.hip) pays with many points per voxel: decide from the previous map of this context,
// before there is one from the grid type (the production POLAR grid is coarse)
bool voxel_morton_order(const rolo_ctx* c) {
  if (c->n_voxels > 0) return c->tgt.n / c->n_voxels >= 16;
  return c->P.voxel_type == ROLO_VOXEL_POLAR;
}

// covariance entries bounded by 1 (regularisations that fix the spectrum, computed here): their voxel sums can go through fixed point
bool voxel_fixed_cov(const rolo_ctx* c) {
  const int r = c->P.regularization;
  return !c->tgt.cov_user && (r == ROLO_REG_PLANE || r == ROLO_REG_NORMALIZED_MIN_EIG || r == ROLO_REG_PLANE_S);
}

void fill_table_params(rolo_ctx* c) {
  static const int polar_exact = [] { const char* e = getenv("ROLO_POLAR_EXACT"); return (e && atoi(e) == 0) ? 0 : 1; }();
  c->tab.polar_exact = polar_exact;
  c->tab.voxel_type = c->P.voxel_type;
  c->tab.voxel_resolution = c->P.voxel_resolution; c->tab.inv_voxel_resolution = 1.0 / c->P.voxel_resolution;
  for (int i = 0; i < 3; i++) c->tab.polar_res[i] = c->P.polar_resolution[i];
}

int ensure_map(rolo_ctx* c) {
  int rc = ensure_covs(c);
  if (rc) return rc;
  if (c->have_map) return ROLO_OK;
  const int n = c->tgt.n;
  size_t capslots = 1024; while (capslots < 2 * (size_t)n) capslots <<= 1;
  if ((rc = ensure(c->tab.keys, c->tab_keys_cap, 2 * capslots))) return rc;   // 16 bytes per slot: key + id
  if ((rc = ensure(c->tab.rec, c->tab_rec_cap, (size_t)n * REC_DOUBLES))) return rc;
  if ((rc = ensure(c->tab.id_keys, c->tab_idk_cap, (size_t)n))) return rc;
  if ((rc = ensure(c->tgt_keys, c->tgt_keys_cap, (size_t)n))) return rc;
  if ((rc = ensure(c->tgt_slot, c->tgt_slot_cap, (size_t)n + KNN_LEAF))) return rc;
  if ((rc = ensure(c->counters, c->counters_cap, 4))) return rc;
  c->tab.mask = (unsigned)(capslots - 1);
  fill_table_params(c);
  { ProfScope ps(c, ROLO_PROF_VOXEL_BUILD); HIPCHK(launch_voxel_build(c->tgt, c->tab, c->tgt_keys, c->tgt_slot, c->counters, voxel_morton_order(c), voxel_fixed_cov(c), c->tgt.bbox6, false, c->stream)); }
  HIPCHK(hipMemcpyAsync(c->h_counters, c->counters, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if ((rc = peer_check(c))) return rc;
  if (c->h_counters[1] != 0) { g_err = c->h_counters[1] == ROLO_ENONFINITE ? "non-finite point or covariance in the voxel map build" : "voxel coordinate outside the packed key range"; return c->h_counters[1]; }
  c->n_voxels = c->h_counters[0];
  c->n_edge = c->h_counters[2];
  c->have_map = true;
  c->have_corr = false;
  return ROLO_OK;
}

// workgroup size of the fused LM launches (tuning: ROLO_LM_THREADS = 512 | 1024)
int lm_threads() {
  static const int t = [] { const char* e = getenv("ROLO_LM_THREADS"); const int v = e ? atoi(e) : 512; return (v == 512 || v == 1024) ? v : 512; }();
  return t;
}
int lm_ppt() {
  static const int t = [] { const char* e = getenv("ROLO_LM_PPT"); const int v = e ? atoi(e) : 1; return (v >= 1 && v <= 16) ? v : 1; }();
  return t;
}
// 0: pass + controller launches; 1: one launch per LM trial (lm_kernel); 2: one launch per frame (lm_persist_kernel)
int lm_mode(const rolo_ctx* c) {
  static const int force = [] { const char* e = getenv("ROLO_LM_FUSED"); return e ? atoi(e) : -1; }();   // A/B runs: 0 / 1 / 2 overrides the parameter
  // with a communicator / peers the sums pass through the exchange between pass and controller
  if (c->comm || peers(c)) return 0;
  const int m = force >= 0 ? force : c->P.fused_lm;
  return m == 2 ? 2 : (m != 0 ? 1 : 0);
}
bool lm_fused(const rolo_ctx* c) { return lm_mode(c) == 1; }
bool lm_persist(const rolo_ctx* c) { return lm_mode(c) == 2; }
// workgroups of the resident LM kernel: on an idle device one 512-thread workgroup per CU (a point per thread at 131 072 points: the shortest trial, 7.8 us), with other
// contexts' frames in flight 64 — the kernel holds the register files it runs on for the whole chain, and four launches of 64 are what the chip takes at one workgroup per
// CU (profiles/r06/concurrency.md); ROLO_LM_PERSIST_WGS pins it (A/B)
int lm_persist_max_wgs(const rolo_ctx* c) {
  static const int v = [] { const char* e = getenv("ROLO_LM_PERSIST_WGS"); const int w = e ? atoi(e) : 0; return (w >= 8 && w <= 256) ? w : 0; }();
  return v ? v : (c->device_busy ? 64 : 256);
}
unsigned long long lm_persist_admit_ticks() {     // how long the resident kernel's workgroups wait for each other to become resident before they leave the frame to the host
  // (ROLO_LM_PERSIST_ADMIT_US=0: a test switch — no launch is ever admitted, every frame takes the bail-out path)
  static const unsigned long long v = [] { const char* e = getenv("ROLO_LM_PERSIST_ADMIT_US"); const long us = e ? atol(e) : 1000; return (unsigned long long)(us >= 0 ? us : 1000) * 100ull; }();
  return v;
}
unsigned long long lm_persist_timeout_ticks() {   // wall_clock64 runs at 100 MHz
  static const unsigned long long v = [] { const char* e = getenv("ROLO_LM_PERSIST_TIMEOUT_MS"); const long ms = e ? atol(e) : 200; return (unsigned long long)(ms > 0 ? ms : 200) * 100000ull; }();
  return v;
}

void shard(const rolo_ctx* c, int& begin, int& end) { rolo_shard_range(c->src.n, c->rank, c->world, &begin, &end); }

int prepare_pass(rolo_ctx* c, PassArgs& a, int& grid) {
  const int noff = n_offsets(c->P);
  int rc;
  for (int b = 0; b < 2; b++) if ((rc = ensure(c->corr[b], c->corr_cap[b], (size_t)c->src.n * noff))) return rc;
  int begin, end; shard(c, begin, end);
  grid = std::max(1, (end - begin + PASS_THREADS - 1) / PASS_THREADS);
  c->lm_rows = std::max(1, (end - begin + lm_threads() * lm_ppt() - 1) / (lm_threads() * lm_ppt()));
  if ((rc = ensure(c->partials, c->partials_cap, std::max((size_t)grid, 2 * (size_t)c->lm_rows) * NV_MAX))) return rc;
  if (lm_persist(c)) {
    // ROLO_LM_PERSIST_BUSY_THREADS=256 (A/B): with other frames in flight twice the workgroups of half the size — the same registers held, on twice the CUs, half of each
    static const int busy_threads = [] { const char* e = getenv("ROLO_LM_PERSIST_BUSY_THREADS"); return (e && atoi(e) == 256) ? 256 : 512; }();
    int T = c->device_busy ? busy_threads : 512;
    const int npts = std::max(end - begin, 1);
    int maxw = std::min(256, lm_persist_max_wgs(c) * (512 / T));
    int ppt = (npts + T * maxw - 1) / (T * maxw);
    if (ppt == 3) ppt = 4;   // (1, 2 and 4 points per thread have the interleaved bodies)
    if (T == 256 && !(ppt == 4 && c->P.optimizer == ROLO_OPT_SO3_LM && noff == 1)) {   // the A/B form exists for the headline's case only
      T = 512; maxw = lm_persist_max_wgs(c);
      ppt = (npts + T * maxw - 1) / (T * maxw);
      if (ppt == 3) ppt = 4;
    }
    c->lmp_ppt = ppt; c->lmp_threads = T;
    c->lmp_rows = (npts + T * c->lmp_ppt - 1) / (T * c->lmp_ppt);
    const size_t need = lm_persist_words(256);   // sized for the largest grid once: the epochs in it must survive a change of the cloud size
    if (!c->xbuf || c->xbuf_cap < need) {
      if ((rc = ensure(c->xbuf, c->xbuf_cap, need))) return rc;
      HIPCHK(hipMemsetAsync(c->xbuf, 0, c->xbuf_cap * sizeof(unsigned long long), c->stream));
    }
  }
  a.src = c->src.xyz; a.cov = c->src.cov; a.n_total = c->src.n; a.begin = begin; a.end = end; a.n_off = noff;
  // the source covariances as I - m m^T: only what the library computed itself for THIS cloud under PLANE (ROLO_PASS_NRM=0: the six-entry form always — the A/B)
  static const bool nrm_on = [] { const char* e = getenv("ROLO_PASS_NRM"); return !e || atoi(e) != 0; }();
  a.nrm = (nrm_on && c->src.have_cov && c->src.have_nrm && !c->src.cov_user && c->src.nrm && c->P.regularization == ROLO_REG_PLANE) ? c->src.nrm : nullptr;
  a.corr[0] = c->corr[0]; a.corr[1] = c->corr[1]; a.partials = c->partials; a.tab = c->tab;
  static const bool xcd_on = [] { const char* e = getenv("ROLO_PASS_XCD"); return !e || atoi(e) != 0; }();
  a.xcd_map = xcd_on ? 1 : 0;
  return ROLO_OK;
}

// one LM trial: fused pass + controller launch, both predicated on the device state
int enqueue_pass(rolo_ctx* c, const PassArgs& a, int grid, int stage, bool publish = false) {
  {
    ProfScope ps(c, stage == 1 ? ROLO_PROF_ROT_PASS : ROLO_PROF_TRANS_PASS);
    if (stage == 1) HIPCHK(launch_rot_pass(c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6, a, c->state, grid, c->stream));
    else HIPCHK(launch_trans_pass(a, c->state, grid, c->stream));
  }
  ProfScope pc(c, ROLO_PROF_CTRL);
  if (c->comm) {
    HIPCHK(launch_reduce(c->partials, grid, c->sums, c->state, stage, c->stream));
    int e = g_rccl.AllReduce(c->sums, c->sums, NV_MAX, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream);
    if (e != 0) { g_err = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?"); return ROLO_ECOMM; }
    HIPCHK(launch_ctrl(c->state, nullptr, 0, c->sums, c->trace, stage, c->stream, publish ? c->h_state : nullptr, nullptr, c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6));
  } else {
    // with peers the controller itself exchanges its row sums through the mailboxes: still ONE launch, still graph-capturable
    HIPCHK(launch_ctrl(c->state, c->partials, grid, nullptr, c->trace, stage, c->stream, publish ? c->h_state : nullptr, peer_args(c), c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6));
  }
  return ROLO_OK;
}

// k fused trials + the closing launch; the state starts and ends in c->state[0] (see passes.hip lm_kernel)
int enqueue_lm_chunk(rolo_ctx* c, const PassArgs& a, int k, bool publish = false) {
  LmState* sb[2] = {c->state, c->state + 1};
  const int nrows = c->lm_rows;
  double* rb[2] = {c->partials, c->partials + (size_t)nrows * NV_MAX};
  const int dof = c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6, T = lm_threads();
  for (int j = 0; j < k; j++) {
    ProfScope ps(c, ROLO_PROF_LM_PASS);
    HIPCHK(launch_lm(dof, T, lm_ppt(), a, sb[j & 1], sb[(j + 1) & 1], rb[(j + 1) & 1], rb[j & 1], nrows, c->trace, 1, c->stream));
  }
  ProfScope ps(c, ROLO_PROF_LM_PASS);
  HIPCHK(launch_lm(dof, T, lm_ppt(), a, sb[k & 1], sb[0], rb[(k + 1) & 1], rb[k & 1], nrows, c->trace, 0, c->stream, publish ? c->h_state : nullptr));
  return ROLO_OK;
}

// both stages (or the one the state is in) to completion in ONE launch (passes.hip lm_persist_kernel); the state starts and ends in c->state[0]
int enqueue_lm_persist(rolo_ctx* c, const PassArgs& a, bool publish = false) {
  ProfScope ps(c, ROLO_PROF_LM_PASS);
  const int cap = (std::max(c->P.max_iterations, c->P.fixed_iterations) + 2) * (std::max(c->P.lm_max_iterations, 0) + 2) * 2 + 16;
  HIPCHK(launch_lm_persist(c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6, c->lmp_threads, c->lmp_ppt, a, c->state, c->xbuf, c->lmp_rows, c->trace, publish ? c->h_state : nullptr,
                           lm_persist_timeout_ticks(), lm_persist_admit_ticks(), cap, c->stream));
  return ROLO_OK;
}

int fetch_state(rolo_ctx* c) {
  HIPCHK(hipMemcpyAsync(c->h_state, c->state, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ROLO_OK;
}

RotBegin make_rot_begin(const rolo_ctx* c, const double* R9, const double* t3, int run_trans) {
  RotBegin b{};
  for (int i = 0; i < 9; i++) b.R[i] = R9 ? R9[i] : ((i % 4 == 0) ? 1.0 : 0.0);
  for (int i = 0; i < 3; i++) b.t[i] = t3 ? t3[i] : 0.0;
  b.optimizer = c->P.optimizer; b.max_iterations = c->P.max_iterations; b.fixed_iterations = c->P.fixed_iterations;
  b.lm_max = c->P.lm_max_iterations; b.q2_intended = c->P.q2_intended; b.rot_eps = c->P.rotation_epsilon;
  b.trans_eps = c->P.transformation_epsilon; b.lm_init = c->P.lm_init_lambda_factor; b.run_trans = run_trans;
  static const int spec_lin = [] { const char* e = getenv("ROLO_LM_SPEC_LIN"); return (e && atoi(e) == 0) ? 0 : 1; }();   // 0: every pass carries both halves (the A/B, rounds 1-4)
  b.spec_lin = spec_lin;
  return b;
}

void fill_trans_knobs(const rolo_ctx* c, TransBegin& tb) {
  tb.max_iterations = c->P.max_iterations; tb.lm_max = c->P.lm_max_iterations; tb.q2_intended = c->P.q2_intended;
  tb.trans_eps = c->P.transformation_epsilon; tb.lm_init = c->P.lm_init_lambda_factor;
}

void guess_to_Rt(const float* g16, double* R, double* t) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i * 3 + j] = g16 ? (double)g16[i * 4 + j] : (i == j ? 1.0 : 0.0); t[i] = g16 ? (double)g16[i * 4 + 3] : 0.0; }
}

void fill_rot_outputs(const LmState* s, float* Tf, double* Td, rolo_stats* st) {
  double T[16];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[i * 4 + j] = s->x0_R[i * 3 + j]; T[i * 4 + 3] = s->x0_t[i]; }
  T[12] = T[13] = T[14] = 0; T[15] = 1;
  if (Td) memcpy(Td, T, sizeof(T));
  if (Tf) for (int i = 0; i < 16; i++) Tf[i] = (float)T[i];
  if (st) { st->n_outer = s->rot_outer; st->converged = s->rot_converged; st->lm_failed = s->rot_failed; st->n_passes = s->rot_passes; st->n_correspondences = s->rot_ncorr; st->n_cost_only = s->rot_cost_only; }
}

int rot_first_chunk(const rolo_ctx* c) { return c->P.fixed_iterations > 0 ? c->P.fixed_iterations + 3 : 8; }
// first chunks of a whole frame (rolo_register_async): from the hints once a frame has been seen
void frame_chunks(const rolo_ctx* c, int& nrot, int& ntrans) {
  nrot = c->hint_rot > 0 ? c->hint_rot : rot_first_chunk(c);
  ntrans = c->hint_trans > 0 ? c->hint_trans : 12;
}
// The first schedule of the next frame holds the most passes any of the last 64 frames needed as predicated pass / controller pairs (fused launches: one more,
// rounded up to an even count); it grows at once and shrinks only when the window's maximum has fallen 6 below it.
// A stream of DIFFERENT frame pairs needs different numbers of LM trials (BASELINE configs[4]: 24 ... 41 per pair). Round 2 followed the last
// frame alone: every other frame either re-captured its hipGraph (the schedule length is part of the graph's key) or topped up through
// host round trips — 351 top-ups, 297 captures and 753 eager frames in 1536. A 16-frame window still re-captured 90 times (the maximum slides
// in and out of a short window) and a capture is milliseconds of host time; a predicated no-op pair costs ~5 us of GPU time.
// Round 5: with pass + controller launches the schedule holds EXACTLY the window's maximum — through round 4 it held one pair more, rounded up to an even count (what
// the fused launches' double-buffered state needs): 22 + 12 pairs for frames that use 21 + 9..10, i.e. six to eight no-op launches of ~2.5 us on every frame's
// critical path. A frame that needs more than any of the last 64 did tops up through one host round trip and raises the hint.
void update_hint(int& hint, rolo_ctx::NeedWindow& w, int used, bool fused) {
  constexpr int WN = 64;
  w.need[w.pos] = used; w.pos = (w.pos + 1) % WN; if (w.n < WN) w.n++;
  int mx = 0;
  for (int i = 0; i < w.n; i++) mx = std::max(mx, w.need[i]);
  const int want = fused ? std::min((std::max(mx + 1, 2) + 1) & ~1, 96) : std::min(std::max(mx, 2), 96);
  if (hint == 0 || want > hint || want <= hint - 6) hint = want;
}

// drive a stage to completion: enqueue predicated passes in chunks, look at the device flags between chunks
// no_persist: the frame's resident kernel gave the stage back (admission, LmState::lmp_bailed): finish with pass + controller launches
int run_stage(rolo_ctx* c, const PassArgs& a, int grid, int stage, int first_chunk, bool no_persist = false) {
  int chunk = first_chunk;
  const int hard_cap = (c->P.max_iterations + 2) * (c->P.lm_max_iterations + 1) + 8;
  int issued = 0;
  while (true) {
    if (lm_persist(c) && !no_persist) { int rc = enqueue_lm_persist(c, a); if (rc) return rc; }   // runs until the state says the stage (and what follows it) is over
    else if (lm_fused(c)) { int rc = enqueue_lm_chunk(c, a, chunk); if (rc) return rc; }
    else for (int i = 0; i < chunk; i++) { int rc = enqueue_pass(c, a, grid, stage); if (rc) return rc; }
    issued += chunk;
    c->n_topup_chunks++;
    int rc = fetch_state(c);
    if (rc) return rc;
    const bool done = (stage == 1) ? (c->h_state->rot_done != 0) : (c->h_state->trans_done != 0);
    if (done) return ROLO_OK;
    if (c->h_state->lmp_bailed && !no_persist) { no_persist = true; c->n_persist_bails++; }
    if (issued > hard_cap) { g_err = "LM stage did not terminate"; return ROLO_ESTATE; }
    chunk = 8;
  }
}

// unmap the peers' mailboxes, free the own one
void peer_disconnect_impl(rolo_ctx* c) {
  rolo_peer_state& P = c->peer;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int r = 0; r < PEER_MAX; r++) {
    if (P.ipc_opened[r] && P.mapped[r]) (void)hipIpcCloseMemHandle(P.mapped[r]);
    P.mapped[r] = nullptr; P.ipc_opened[r] = false;
  }
  if (P.connected) { c->rank = 0; c->world = 1; c->have_corr = false; c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false; }
  P.connected = false; P.args = PeerArgs{};
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }   // a captured schedule holds the peers' pointers
  if (c->galt.exec) { (void)hipGraphExecDestroy(c->galt.exec); c->galt.exec = nullptr; }
  c->gseen_valid = false;
}
void peer_release(rolo_ctx* c) {
  rolo_peer_state& P = c->peer;
  peer_disconnect_impl(c);
  if (P.base) {
    { std::lock_guard<std::mutex> lk(g_peer_mu); g_peer_exports.erase(P.handle); }
    (void)hipFree(P.base); P.base = nullptr; P.bytes = 0;
  }
  if (P.h_err) { (void)hipHostFree(P.h_err); P.h_err = nullptr; }
}

}  // namespace

namespace rolo {
// accessors for front.hip / odometry.hip (the context layout is private to this file)
void** ctx_front_slot(rolo_ctx* c) { return &c->front; }
void** ctx_s2m_slot(rolo_ctx* c) { return &c->s2m; }
unsigned long long ctx_cloud_epoch(rolo_ctx* c) { return c->cloud_epoch; }
hipStream_t ctx_stream(rolo_ctx* c) { return c->stream; }
int ctx_device(rolo_ctx* c) { return c->device; }
void ctx_set_error(const char* msg) { g_err = msg ? msg : ""; }
void ctx_set_fused_lm(rolo_ctx* c, int mode) { c->P.fused_lm = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
// rolo_set_source_device(T * src) + rolo_set_target_device(tgt) as ONE launch (the odometry driver's per-frame hand-over; both clouds on the device)
int ctx_set_pair_device(rolo_ctx* c, const float* d_src, int n_src, int stride_src, const float* T16_host_or_null, const float* d_tgt, int n_tgt, int stride_tgt) {
  if (!c) return ROLO_EINVAL;
  if (n_src <= 0 || n_tgt <= 0 || stride_src < 3 || stride_tgt < 3 || !d_src || !d_tgt) { g_err = "bad cloud arguments"; return ROLO_EINVAL; }
  int rc = set_device(c); if (rc) return rc;
  CloudDev* cl[2] = {&c->src, &c->tgt};
  size_t* cap[2] = {&c->src_xyz_cap, &c->tgt_xyz_cap};
  const int n[2] = {n_src, n_tgt};
  for (int i = 0; i < 2; i++) {
    if ((rc = ensure(cl[i]->xyz, *cap[i], (size_t)n[i]))) return rc;
    if ((rc = ensure(cl[i]->bbox_part, cl[i]->bbox_part_cap, (size_t)((n[i] + 255) / 256) * 6))) return rc;
  }
  HIPCHK(launch_pack_pair(d_src, stride_src, c->src.xyz, n_src, c->src.bbox_part, T16_host_or_null, d_tgt, stride_tgt, c->tgt.xyz, n_tgt, c->tgt.bbox_part, c->stream));
  for (int i = 0; i < 2; i++) { cl[i]->n_bbox_part = (n[i] + 255) / 256; cl[i]->n = n[i]; cl[i]->have_cov = false; cl[i]->have_nrm = false; cl[i]->have_sorted = false; cl[i]->bbox6 = nullptr; }
  c->have_map = false; c->have_corr = false;
  c->cloud_epoch++;
  return ROLO_OK;
}
// scan2map.hip: the two sub-map clouds (corner, surface) as the context's source / target with their search trees built
int ctx_build_map_trees(rolo_ctx* c, const float* corner, int nc, const float* surf, int ns, int stride, KnnPair* out) {
  int rc = set_device(c); if (rc) return rc;
  if ((rc = upload_cloud(c, c->src, c->src_xyz_cap, corner, nc, stride, false))) return rc;
  if ((rc = upload_cloud(c, c->tgt, c->tgt_xyz_cap, surf, ns, stride, false))) return rc;
  c->have_map = false; c->have_corr = false;
  return build_clouds(c, true, true, c->stream, true, out);
}
}  // namespace rolo

extern "C" {

const char* rolo_last_error(void) { return g_err.c_str(); }

int rolo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void rolo_default_params(rolo_params* p) {
  p->k_correspondences = 20;
  p->regularization = ROLO_REG_PLANE;
  p->neighbor_search = ROLO_DIRECT1;
  p->voxel_type = ROLO_VOXEL_POLAR;
  p->voxel_resolution = 1.0;
  p->polar_resolution[0] = 1.0; p->polar_resolution[1] = 0.0; p->polar_resolution[2] = 0.0;
  p->optimizer = ROLO_OPT_SO3_LM;
  p->max_iterations = 64;
  p->rotation_epsilon = 2e-3;
  p->transformation_epsilon = 5e-4;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->fixed_iterations = 0;
  p->q2_intended = 0;
  p->overlap_knn = 1;
  p->use_graph = 1;
  p->fused_lm = 2;
}

// frames in flight per device (rolo_register_async .. rolo_register_wait): a frame enqueued while OTHER contexts of the device have frames in flight takes the
// kernels that share the chip best (throughput), a frame enqueued on an idle device the ones that finish soonest (latency) — launch_knn_walk
static std::atomic<int> g_frames_in_flight[64];
static void count_in_flight(rolo_ctx* c, bool on) {
  if (on == c->counted_in_flight || c->device < 0 || c->device >= 64) return;
  c->counted_in_flight = on;
  g_frames_in_flight[c->device].fetch_add(on ? 1 : -1, std::memory_order_relaxed);
}
static bool others_in_flight(const rolo_ctx* c) {
  if (c->device < 0 || c->device >= 64) return false;
  return g_frames_in_flight[c->device].load(std::memory_order_relaxed) - (c->counted_in_flight ? 1 : 0) > 0;
}

static int ctx_create_impl(int device, bool high_priority, rolo_ctx** out);
int rolo_ctx_create(int device, rolo_ctx** out) { return ctx_create_impl(device, false, out); }
}  // extern "C"
namespace rolo {
// A context whose main stream has HIGH priority (the odometry driver's front-end context: the short K1-K4 kernels of frame k + 1 next to the
// registration of frame k, and never behind it in a shared hardware queue). Measured neutral to +0.5 % on the pipelined rate (2055 vs 2040 frames/s).
int ctx_create_high_priority(int device, rolo_ctx** out) { return ctx_create_impl(device, true, out); }
}  // namespace rolo
extern "C" {
// ROLO_CU_PARTITION=<groups> (2 | 4 | 8; default 0 = off): the main stream of the k-th context created in this process is confined to XCD group k % groups
// (hipExtStreamCreateWithCUMask; mask bit i = CU i, which sits on XCD i % 8 on this part) — every context its own XCDs and L2s, so that one frame's kernel
// boundaries cannot write back / invalidate the L2 under another frame's kernels. An experiment switch (round 3's verdict, item 4); the measurement is in DESIGN.md.
static int cu_partition_env() {
  static const int v = [] { const char* e = getenv("ROLO_CU_PARTITION"); const int g = e ? atoi(e) : 0; return (g == 2 || g == 4 || g == 8) ? g : 0; }();
  return v;
}
static std::atomic<int> g_ctx_serial{0};

// ---- stream bank: the placement of the contexts' streams on HIP's hardware queues belongs to the library ------------------------------------------
// HIP deals a process's streams to its 4 hardware queues in CREATION order, and throughput with several frames in flight depends on the result:
// four contexts whose main streams sit two by two on two queues register 2940 scans/s, each on a queue of its own 2140. Rounds
// 1-3 got the good layout by creating an idle second stream with every context — which held only while contexts were created back to back in a fresh
// process: four more contexts next to four idle ones (bench.py's config5 leg, round 3), or a caller that creates one stream of its own between two
// contexts, landed in a bad layout and lost 27 % with nothing to detect it. Now every device has a BANK of streams created back to back, eight at a
// time, on first use; a context takes the lowest free PAIR (main stream + the eager fork of rolo_batch_*) and gives it back when it is destroyed.
// The pairs' positions relative to each other — main streams on every other stream of a burst: two hardware queues, alternating — no longer depend
// on when contexts come and go or on what else the process creates in between. bench.py's `layout_check` re-measures after foreign streams.
struct StreamBank { std::vector<hipStream_t> s; std::vector<char> used; };
static std::mutex g_bank_mu;
static std::map<int, StreamBank> g_banks;
static int bank_acquire_pair(int device, hipStream_t* main, hipStream_t* second, int* slot) {
  std::lock_guard<std::mutex> lk(g_bank_mu);
  StreamBank& B = g_banks[device];
  size_t i = 0;
  for (; i + 1 < B.s.size(); i += 2) if (!B.used[i]) break;
  if (i + 1 >= B.s.size()) {   // another burst of eight, back to back
    i = B.s.size();
    for (int k = 0; k < 8; k++) {
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return ROLO_EHIP;
      B.s.push_back(st); B.used.push_back(0);
    }
  }
  B.used[i] = B.used[i + 1] = 1;
  *main = B.s[i]; *second = B.s[i + 1]; *slot = (int)i;
  return ROLO_OK;
}
static void bank_release_pair(int device, int slot) {
  std::lock_guard<std::mutex> lk(g_bank_mu);
  auto it = g_banks.find(device);
  if (it == g_banks.end() || slot < 0 || (size_t)slot + 1 >= it->second.s.size()) return;
  it->second.used[(size_t)slot] = it->second.used[(size_t)slot + 1] = 0;
}
static hipError_t create_masked_stream(hipStream_t* st, int device, int group, int groups) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return e;
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32, per = 8 / groups;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int i = 0; i < ncu; i++) { const int xcd = i % 8; if (xcd / per == group) mask[(size_t)i / 32] |= 1u << (i % 32); }
  return hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask.data());
}

static int ctx_create_impl(int device, bool high_priority, rolo_ctx** out) {
  if (!out) return ROLO_EINVAL;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { g_err = "no HIP device (librolo_hip has no CPU fallback)"; return ROLO_EHIP; }
  if (device < 0 || device >= n) { g_err = "bad device ordinal"; return ROLO_EINVAL; }
  rolo_ctx* c = new rolo_ctx();
  c->device = device;
  rolo_default_params(&c->P);
  int prio_lo = 0, prio_hi = 0;   // (numerically lower = higher priority)
  if (high_priority && (hipSetDevice(device) != hipSuccess || hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess)) { prio_lo = prio_hi = 0; (void)hipGetLastError(); }
  const int groups = high_priority ? 0 : cu_partition_env();
  bool ok = hipSetDevice(device) == hipSuccess;
  if (ok && (groups || high_priority)) {   // experiment / front-end contexts: streams of their own
    ok = (groups ? create_masked_stream(&c->stream, device, g_ctx_serial.fetch_add(1) % groups, groups)
                 : (prio_hi != prio_lo ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking))) == hipSuccess &&
         hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess;
  } else if (ok) ok = bank_acquire_pair(device, &c->stream, &c->stream2, &c->bank_slot) == ROLO_OK;
  if (!ok || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreate(&c->ev_done) != hipSuccess || hipEventCreate(&c->ev_start) != hipSuccess) { rolo_ctx_destroy(c); g_err = "hipStreamCreate failed"; return ROLO_EHIP; }
  if (hipHostMalloc((void**)&c->h_state, sizeof(LmState)) != hipSuccess || hipHostMalloc((void**)&c->h_sums, sizeof(double) * NV_MAX) != hipSuccess ||
      hipHostMalloc((void**)&c->h_counters, sizeof(int) * 4) != hipSuccess || hipHostMalloc((void**)&c->h_args, sizeof(FrameArgs)) != hipSuccess) { rolo_ctx_destroy(c); g_err = "hipHostMalloc failed"; return ROLO_EHIP; }
  memset(c->h_state, 0, sizeof(LmState));
  int rc = ensure(c->state, c->state_cap, 2);   // [0] the state every entry point sees; [1] the other half of the fused launches' double buffer
  if (!rc) rc = ensure(c->sums, c->sums_cap, NV_MAX);
  if (!rc) rc = ensure(c->trace, c->trace_cap, TRACE_CAP);
  if (!rc) rc = ensure(c->d_args, c->d_args_cap, 1);
  if (!rc && hipMemsetAsync(c->state, 0, 2 * sizeof(LmState), c->stream) != hipSuccess) rc = ROLO_EHIP;
  if (rc) { rolo_ctx_destroy(c); return rc; }
  *out = c;
  return ROLO_OK;
}

void rolo_front_destroy(rolo_ctx* c);  // front.hip
void rolo_s2m_destroy(rolo_ctx* c);    // scan2map.hip
void rolo_s2m_forget(rolo_ctx* c);     // scan2map.hip: a context going back to the pool forgets its resident sub-map and gives its helper context back
}
namespace rolo { void front_reset_object_state(rolo_ctx* c); }   // front.hip
extern "C" {

void rolo_ctx_destroy(rolo_ctx* c) {
  if (!c) return;
  count_in_flight(c, false);
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  rolo_front_destroy(c);
  rolo_s2m_destroy(c);
  peer_release(c);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  void* bufs[] = {c->src.nrm, c->tgt.nrm, c->src.bbox_part, c->tgt.bbox_part, c->src.xyz, c->src.cov, c->src.sorted, c->src.boxes, c->src.knn_idx, c->src.knn_d2, c->tgt.xyz, c->tgt.cov, c->tgt.sorted,
                  c->tgt.boxes, c->tgt.knn_idx, c->tgt.knn_d2, c->ks[0].sort_tmp, c->ks[0].keys0, c->ks[0].keys1, c->ks[0].vals0, c->ks[0].vals1, c->ks[0].bbox,
                  c->ks[1].sort_tmp, c->ks[1].keys0, c->ks[1].keys1, c->ks[1].vals0, c->ks[1].vals1, c->ks[1].bbox, c->ks[0].nbr, c->ks[1].nbr, c->ks[0].stage, c->ks[1].stage, c->ks[0].lower, c->ks[1].lower, c->tab.keys,
                  c->tab.rec, c->tab.id_keys, c->tgt_keys, c->tgt_slot, c->counters, c->corr[0], c->corr[1], c->partials, c->sums,
                  c->state, c->trace, c->stage_in, c->stage_out, c->stage_d, c->stage_i, c->xbuf};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->galt.exec) (void)hipGraphExecDestroy(c->galt.exec);
  if (c->galt.graph) (void)hipGraphDestroy(c->galt.graph);
  if (c->dbg_chain_exec) (void)hipGraphExecDestroy(c->dbg_chain_exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->h_args) (void)hipHostFree(c->h_args);
  if (c->d_args) (void)hipFree(c->d_args);
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->h_sums) (void)hipHostFree(c->h_sums);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); if (c->bank_slot < 0) (void)hipStreamDestroy(c->stream2); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->stream && c->bank_slot < 0) (void)hipStreamDestroy(c->stream);
  if (c->bank_slot >= 0) bank_release_pair(c->device, c->bank_slot);
  delete c;
}

// ---- context pool: RotVGICP() / ~RotVGICP() per frame without per-frame allocation ------------------------------------------------------
// The reference constructs its operator inside scanRegeistration — one object per frame (src/lidarOdometry.cpp:460) — which for a HIP context
// means 2 streams, 3 events, 4 pinned and ~45 device allocations per frame. rolo_ctx_acquire hands out a released context of the same device
// instead (buffers, streams and a captured hipGraph of the frame kept; parameters, clouds and every cached result reset to a fresh object's).
namespace {
std::mutex g_pool_mu;
std::vector<rolo_ctx*> g_pool;
constexpr size_t POOL_MAX = 8;
void reset_to_fresh(rolo_ctx* c) {
  rolo_default_params(&c->P);
  c->src.n = 0; c->tgt.n = 0;
  c->cloud_epoch++;   // whatever sub-map the previous owner left resident is not the next owner's
  c->src.have_cov = c->tgt.have_cov = false; c->src.have_sorted = c->tgt.have_sorted = false; c->src.cov_user = c->tgt.cov_user = false; c->src.have_nrm = c->tgt.have_nrm = false;
  c->src.bbox6 = c->tgt.bbox6 = nullptr; c->src.n_bbox_part = c->tgt.n_bbox_part = 0;
  c->have_map = false; c->have_corr = false; c->n_voxels = 0; c->n_edge = 0;
  c->want_knn_lists = false; c->prof_on = false; c->shard_knn = false;
  c->rank = 0; c->world = 1; c->load_hint = -1; c->busy_credit = 0;
  front_reset_object_state(c);   // no projection, armed de-skew or pre-cleared arrays of the previous owner
  c->n_frames = c->n_replays = c->n_captures = c->n_eager = c->n_topup_frames = c->n_topup_chunks = 0;   // rolo_ctx_counters counts per object
  c->ns_enqueue = c->ns_wait_blocked = c->ns_wait_other = 0; c->n_persist_bails = 0;
  // (schedule hints, their windows and the captured graph stay on purpose: they are keyed on sizes, buffers and parameters, not on the object's
  // identity — a frame loop that constructs its operator per frame, src/lidarOdometry.cpp:460, keeps replaying its graph. Drivers created on the
  // context, rolo_odom_create, must be destroyed before the context is released: the pool does not track them.)
}
}  // namespace

int rolo_ctx_acquire(int device, rolo_ctx** out) {
  if (!out) return ROLO_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) {
      if (g_pool[i]->device != device) continue;
      *out = g_pool[i];
      g_pool.erase(g_pool.begin() + (long)i);
      return ROLO_OK;
    }
  }
  return rolo_ctx_create(device, out);
}

void rolo_ctx_release(rolo_ctx* c) {
  if (!c) return;
  // a context with a communicator / peers, a frame in flight or event timers is not worth keeping: destroy
  if (c->comm || c->peer.base || c->async_pending || !c->prof.empty()) { rolo_ctx_destroy(c); return; }
  rolo_s2m_forget(c);   // (before the pool's lock is taken: it releases the helper context through this same function)
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < POOL_MAX) { (void)hipSetDevice(c->device); if (c->stream) (void)hipStreamSynchronize(c->stream); reset_to_fresh(c); g_pool.push_back(c); return; }
  }
  rolo_ctx_destroy(c);
}

void rolo_ctx_pool_clear(void) {
  std::vector<rolo_ctx*> all;
  { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
  for (rolo_ctx* c : all) rolo_ctx_destroy(c);
}

void* rolo_ctx_stream(rolo_ctx* c) { return c ? (void*)c->stream : nullptr; }

int rolo_set_params(rolo_ctx* c, const rolo_params* p) {
  if (!c || !p) return ROLO_EINVAL;
  if (p->neighbor_search < ROLO_DIRECT27 || p->neighbor_search > ROLO_DIRECT1) { g_err = "unsupported neighbor search method"; return ROLO_EUNSUPPORTED; }  // vmp_voxel.hpp:16-18 aborts
  if (p->regularization < ROLO_REG_NONE || p->regularization > ROLO_REG_PLANE_S) { g_err = "bad regularization"; return ROLO_EINVAL; }
  if (p->optimizer < ROLO_OPT_GN || p->optimizer > ROLO_OPT_SO3_LM) { g_err = "bad optimizer"; return ROLO_EINVAL; }
  const bool cov_change = p->k_correspondences != c->P.k_correspondences || p->regularization != c->P.regularization;
  const bool map_change = p->voxel_type != c->P.voxel_type || p->voxel_resolution != c->P.voxel_resolution ||
                          memcmp(p->polar_resolution, c->P.polar_resolution, sizeof(p->polar_resolution)) != 0;
  c->P = *p;
  if (cov_change) { c->src.have_cov = false; c->tgt.have_cov = false; }
  if (cov_change || map_change) { c->have_map = false; c->have_corr = false; }  // setResolution / setPolarResolution: voxelmap_.reset()
  return ROLO_OK;
}

int rolo_set_target(rolo_ctx* c, const float* pts, int n, int stride) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  rc = upload_cloud(c, c->tgt, c->tgt_xyz_cap, pts, n, stride, false);
  c->have_map = false; c->have_corr = false;
  return rc;
}
int rolo_set_source(rolo_ctx* c, const float* pts, int n, int stride) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  rc = upload_cloud(c, c->src, c->src_xyz_cap, pts, n, stride, false);
  c->have_corr = false;
  return rc;
}
int rolo_set_target_device(rolo_ctx* c, const float* d_pts, int n, int stride) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  rc = upload_cloud(c, c->tgt, c->tgt_xyz_cap, d_pts, n, stride, true);
  c->have_map = false; c->have_corr = false;
  return rc;
}
int rolo_set_source_device(rolo_ctx* c, const float* d_pts, int n, int stride) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  rc = upload_cloud(c, c->src, c->src_xyz_cap, d_pts, n, stride, true);
  c->have_corr = false;
  return rc;
}

int rolo_swap_source_and_target(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  std::swap(c->src, c->tgt);
  std::swap(c->src_xyz_cap, c->tgt_xyz_cap); std::swap(c->src_cov_cap, c->tgt_cov_cap);
  std::swap(c->src_sorted_cap, c->tgt_sorted_cap); std::swap(c->src_boxes_cap, c->tgt_boxes_cap);
  std::swap(c->src_knn_cap, c->tgt_knn_cap); std::swap(c->src_knnd_cap, c->tgt_knnd_cap);
  c->have_map = false; c->have_corr = false;   // (bbox6 travels with the CloudDev)
  return ROLO_OK;
}
int rolo_adopt_target_covariances(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  if (!c->tgt.have_cov || c->tgt.n <= 0 || c->src.n != c->tgt.n) { g_err = "no target covariances of matching size to adopt"; return ROLO_ESTATE; }
  std::swap(c->src.cov, c->tgt.cov);
  std::swap(c->src_cov_cap, c->tgt_cov_cap);
  std::swap(c->src.nrm, c->tgt.nrm); std::swap(c->src.nrm_cap, c->tgt.nrm_cap);
  c->src.have_nrm = c->tgt.have_nrm; c->tgt.have_nrm = false; c->src.cov_user = c->tgt.cov_user;
  c->src.have_cov = true; c->tgt.have_cov = false;
  c->have_map = false; c->have_corr = false;
  return ROLO_OK;
}
int rolo_clear_source(rolo_ctx* c) { if (!c) return ROLO_EINVAL; c->src.n = 0; c->src.have_cov = false; c->have_corr = false; return ROLO_OK; }
int rolo_clear_target(rolo_ctx* c) { if (!c) return ROLO_EINVAL; c->tgt.n = 0; c->tgt.have_cov = false; c->have_map = false; c->have_corr = false; return ROLO_OK; }

int rolo_compute_covariances(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  if ((rc = ensure_covs(c))) return rc;
  if (peers(c)) { HIPCHK(hipStreamSynchronize(c->stream)); return peer_check(c); }   // a timed-out exchange surfaces here
  return ROLO_OK;
}

static int get_covs(rolo_ctx* c, CloudDev& cl, double* covs) {
  if (!covs) return ROLO_EINVAL;
  if (!cl.have_cov) { g_err = "covariances not computed"; return ROLO_ESTATE; }
  int rc = ensure(c->stage_d, c->stage_d_cap, (size_t)cl.n * 16);
  if (rc) return rc;
  HIPCHK(launch_cov_unpack(cl.cov, cl.n, c->stage_d, c->stream));
  HIPCHK(hipMemcpyAsync(covs, c->stage_d, sizeof(double) * 16 * (size_t)cl.n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ROLO_OK;
}
int rolo_get_source_covariances(rolo_ctx* c, double* covs) { if (!c) return ROLO_EINVAL; int rc = set_device(c); return rc ? rc : get_covs(c, c->src, covs); }
int rolo_get_target_covariances(rolo_ctx* c, double* covs) { if (!c) return ROLO_EINVAL; int rc = set_device(c); return rc ? rc : get_covs(c, c->tgt, covs); }

static int set_covs(rolo_ctx* c, CloudDev& cl, size_t& cov_cap, const double* covs) {
  if (!covs || cl.n <= 0) return ROLO_EINVAL;
  int rc = ensure(cl.cov, cov_cap, 6 * (size_t)cl.n);
  if (!rc) rc = ensure(c->stage_d, c->stage_d_cap, (size_t)cl.n * 16);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->stage_d, covs, sizeof(double) * 16 * (size_t)cl.n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(launch_cov_pack(c->stage_d, cl.n, cl.cov, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  cl.have_cov = true; cl.cov_user = true; cl.have_nrm = false;
  return ROLO_OK;
}
int rolo_set_source_covariances(rolo_ctx* c, const double* covs) { if (!c) return ROLO_EINVAL; int rc = set_device(c); if (rc) return rc; c->have_corr = false; return set_covs(c, c->src, c->src_cov_cap, covs); }
int rolo_set_target_covariances(rolo_ctx* c, const double* covs) { if (!c) return ROLO_EINVAL; int rc = set_device(c); if (rc) return rc; c->have_map = false; c->have_corr = false; return set_covs(c, c->tgt, c->tgt_cov_cap, covs); }

int rolo_get_knn(rolo_ctx* c, int which, int32_t* idx, float* d2) {
  if (!c || !idx || !d2) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  CloudDev& cl = which == 0 ? c->src : c->tgt;
  if (cl.n <= 0) return ROLO_ESTATE;
  c->want_knn_lists = true;
  cl.have_cov = false;
  rc = which == 0 ? build_src(c, c->stream) : build_tgt(c, c->stream);
  c->want_knn_lists = false;
  if (rc) return rc;
  const size_t m = (size_t)cl.n * c->P.k_correspondences;
  HIPCHK(hipMemcpyAsync(idx, cl.knn_idx, sizeof(int32_t) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(d2, cl.knn_d2, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ROLO_OK;
}

int rolo_build_voxelmap(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  c->have_map = false;
  return ensure_map(c);
}
int rolo_num_voxels(rolo_ctx* c) { return (c && c->have_map) ? c->n_voxels : ROLO_ESTATE; }
int rolo_num_edge_points(rolo_ctx* c) { return (c && c->have_map) ? c->n_edge : ROLO_ESTATE; }

static void unpack_key_host(unsigned long long key, int32_t* k3) {
  k3[0] = (int)(key & 0x1fffffu) - KEY_BIAS; k3[1] = (int)((key >> 21) & 0x1fffffu) - KEY_BIAS; k3[2] = (int)((key >> 42) & 0x1fffffu) - KEY_BIAS;
}

int rolo_get_voxels(rolo_ctx* c, int32_t* keys, int32_t* counts, double* means, double* covs) {
  if (!c) return ROLO_EINVAL;
  if (!c->have_map) { g_err = "voxel map not built"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  const int V = c->n_voxels;
  std::vector<double> rec((size_t)V * REC_DOUBLES);
  std::vector<unsigned long long> idk(V);
  HIPCHK(hipMemcpyAsync(rec.data(), c->tab.rec, sizeof(double) * rec.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(idk.data(), c->tab.id_keys, sizeof(unsigned long long) * (size_t)V, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int v = 0; v < V; v++) {
    const double* r = &rec[(size_t)v * REC_DOUBLES];
    if (keys) unpack_key_host(idk[v], keys + 3 * (size_t)v);
    if (counts) counts[v] = (int32_t)(r[10] + 0.5);
    if (means) { means[4 * (size_t)v] = r[0]; means[4 * (size_t)v + 1] = r[1]; means[4 * (size_t)v + 2] = r[2]; means[4 * (size_t)v + 3] = 1.0; }
    if (covs) {
      double* o = covs + 16 * (size_t)v;
      o[0] = r[3]; o[1] = r[4]; o[2] = r[5]; o[3] = 0; o[4] = r[4]; o[5] = r[6]; o[6] = r[7]; o[7] = 0;
      o[8] = r[5]; o[9] = r[7]; o[10] = r[8]; o[11] = 0; o[12] = o[13] = o[14] = o[15] = 0;
    }
  }
  return ROLO_OK;
}

int rolo_get_target_voxel_keys(rolo_ctx* c, int32_t* keys) {
  if (!c || !keys) return ROLO_EINVAL;
  if (c->tgt.n <= 0) return ROLO_ESTATE;
  int rc = set_device(c); if (rc) return rc;
  rc = ensure(c->stage_i, c->stage_i_cap, 3 * (size_t)c->tgt.n);
  if (rc) return rc;
  fill_table_params(c);
  HIPCHK(launch_voxel_keys(c->tgt.xyz, c->tgt.n, c->tab, c->stage_i, c->stream));
  HIPCHK(hipMemcpyAsync(keys, c->stage_i, sizeof(int32_t) * 3 * (size_t)c->tgt.n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ROLO_OK;
}

// ---- stage-level evaluations ------------------------------------------------------------------------------
static int eval_rot(rolo_ctx* c, const double* T, int dof_optimizer, int mode, double* Hout, double* bout, double* err) {
  if (!c || !T) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  if ((rc = ensure_map(c))) return rc;
  if (mode == 1 && !c->have_corr) { g_err = "no cached correspondences: call a linearize first"; return ROLO_ESTATE; }
  PassArgs a; int grid;
  if ((rc = prepare_pass(c, a, grid))) return rc;
  double R[9], t[3];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i * 3 + j] = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
  RotBegin b = make_rot_begin(c, R, t, 0);
  b.optimizer = dof_optimizer;
  const int dof = dof_optimizer == ROLO_OPT_SO3_LM ? 3 : 6;
  HIPCHK(launch_eval_begin(c->state, b, mode, c->stream));
  HIPCHK(launch_rot_pass(dof, a, c->state, grid, c->stream));
  HIPCHK(launch_reduce(c->partials, grid, c->sums, c->state, -1, c->stream));
  if (c->comm) {
    int e = g_rccl.AllReduce(c->sums, c->sums, NV_MAX, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream);
    if (e != 0) { g_err = "ncclAllReduce failed"; return ROLO_ECOMM; }
  } else if (peers(c)) HIPCHK(launch_peer_allreduce(c->sums, c->peer.args, c->peer.h_err, c->stream));
  HIPCHK(launch_eval_end(c->state, c->sums, mode, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_sums, c->sums, sizeof(double) * NV_MAX, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if ((rc = peer_check(c))) return rc;
  const double* S = c->h_sums;
  if (mode == 0) {
    c->have_corr = true;
    if (err) *err = S[V_Y];
    if (Hout) { int tt = 0; for (int i = 0; i < dof; i++) for (int j = 0; j <= i; j++) { Hout[i * dof + j] = S[V_H + tt]; Hout[j * dof + i] = S[V_H + tt]; tt++; } }
    if (bout) for (int i = 0; i < dof; i++) bout[i] = S[V_B + i];
    if ((int)(S[V_N] + 0.5) == 0) { g_err = "no correspondences"; return ROLO_ENOCORR; }
  } else {
    if (err) *err = S[V_YI];
  }
  return ROLO_OK;
}

int rolo_so3_linearize(rolo_ctx* c, const double* T, double* H9, double* b3, double* err) { return eval_rot(c, T, ROLO_OPT_SO3_LM, 0, H9, b3, err); }
int rolo_linearize(rolo_ctx* c, const double* T, double* H36, double* b6, double* err) { return eval_rot(c, T, ROLO_OPT_LM, 0, H36, b6, err); }
int rolo_compute_error(rolo_ctx* c, const double* T, double* err) { return eval_rot(c, T, ROLO_OPT_SO3_LM, 1, nullptr, nullptr, err); }

int rolo_get_correspondences(rolo_ctx* c, int32_t* found, int32_t* keys) {
  if (!c || !found) return ROLO_EINVAL;
  if (!c->have_corr) { g_err = "no cached correspondences"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  if ((rc = fetch_state(c))) return rc;
  const int noff = n_offsets(c->P);
  const size_t m = (size_t)c->src.n * noff;
  std::vector<int> ids(m);
  std::vector<unsigned long long> idk(std::max(c->n_voxels, 1));
  HIPCHK(hipMemcpyAsync(ids.data(), c->corr[c->h_state->tr_cur], sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(idk.data(), c->tab.id_keys, sizeof(unsigned long long) * (size_t)c->n_voxels, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  int b, e; shard(c, b, e);
  for (size_t i = 0; i < m; i++) {
    const int pt = (int)(i / noff);
    const bool in = pt >= b && pt < e && ids[i] >= 0;
    found[i] = in ? 1 : 0;
    if (keys) { if (in) unpack_key_host(idk[ids[i]], keys + 3 * i); else keys[3 * i] = keys[3 * i + 1] = keys[3 * i + 2] = 0; }
  }
  return ROLO_OK;
}

static int eval_t3(rolo_ctx* c, const double* t3, const double* g3, const double* l3, double dtn, double dtn1, float lam, int phase, double* H36, double* b6, double* err) {
  if (!c || !t3 || !g3 || !l3) return ROLO_EINVAL;
  if (!c->have_corr) { g_err = "no cached correspondences: run a linearize / align first"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  PassArgs a; int grid;
  if ((rc = prepare_pass(c, a, grid))) return rc;
  TransBegin tb{};
  for (int i = 0; i < 3; i++) { tb.t0[i] = t3[i]; tb.g[i] = g3[i]; tb.l[i] = l3[i]; }
  tb.dtn = dtn; tb.dtn1 = dtn1; tb.ct_lambda = lam; tb.direct = 0; fill_trans_knobs(c, tb);
  HIPCHK(launch_t3_eval_begin(c->state, tb, phase, c->stream));
  HIPCHK(launch_trans_pass(a, c->state, grid, c->stream));
  HIPCHK(launch_reduce(c->partials, grid, c->sums, c->state, -1, c->stream));
  if (c->comm) {
    int e = g_rccl.AllReduce(c->sums, c->sums, NV_MAX, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream);
    if (e != 0) { g_err = "ncclAllReduce failed"; return ROLO_ECOMM; }
  } else if (peers(c)) HIPCHK(launch_peer_allreduce(c->sums, c->peer.args, c->peer.h_err, c->stream));
  HIPCHK(launch_eval_end(c->state, c->sums, 1, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_sums, c->sums, sizeof(double) * NV_MAX, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if ((rc = peer_check(c))) return rc;
  const double* S = c->h_sums;
  if (phase == 0) {
    if (err) *err = S[V_Y];
    if (H36) { int tt = 0; for (int i = 0; i < 6; i++) for (int j = 0; j <= i; j++) { H36[i * 6 + j] = S[V_H + tt]; H36[j * 6 + i] = S[V_H + tt]; tt++; } }
    if (b6) for (int i = 0; i < 6; i++) b6[i] = S[V_B + i];
  } else if (err) *err = S[V_YI];
  return ROLO_OK;
}
int rolo_t3_linearize(rolo_ctx* c, const double* t3, const double* g3, const double* l3, double dtn, double dtn1, float lam, double* H36, double* b6, double* err) {
  return eval_t3(c, t3, g3, l3, dtn, dtn1, lam, 0, H36, b6, err);
}
int rolo_compute_t_error(rolo_ctx* c, const double* t3, const double* g3, const double* l3, double dtn, double dtn1, float lam, double* err) {
  return eval_t3(c, t3, g3, l3, dtn, dtn1, lam, 1, nullptr, nullptr, err);
}

// ---- drivers ------------------------------------------------------------------------------------------------
static int enqueue_frame(rolo_ctx* c, bool with_trans);
// One enqueue, one wait (round 4; until then: covariances + voxel map with a host round trip for its counters, then the LM chunks): the frame path of
// rolo_register_async with the rotation stage alone — the voxel map rides inside the search's launches (VoxelFuse), its finalize kernel starts the LM
// state from pinned arguments, the first schedule of predicated trials follows the last frames' need, and the host waits once.
int rolo_align(rolo_ctx* c, const float* guess16, float* Tf, double* Td, rolo_stats* stats) {
  if (!c) return ROLO_EINVAL;
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "source/target not set"; return ROLO_ESTATE; }
  c->have_map = false;  // computeTransformation: voxelmap_.reset() (rot_vgicp_impl.hpp:147)
  double R[9], t[3]; guess_to_Rt(guess16, R, t);
  c->h_args->rot = make_rot_begin(c, R, t, 0);
  c->h_args->trans = TransBegin{}; fill_trans_knobs(c, c->h_args->trans);
  if ((rc = enqueue_frame(c, false))) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  if ((rc = peer_check(c))) return rc;
  if (c->h_counters[1] != 0) { g_err = c->h_counters[1] == ROLO_ENONFINITE ? "non-finite point or covariance in the voxel map build" : "voxel coordinate outside the packed key range"; return c->h_counters[1]; }
  c->n_voxels = c->h_counters[0];
  c->n_edge = c->h_counters[2];
  c->have_map = true;
  if (!c->h_state->rot_done) {   // the first schedule was too short (or the resident kernel gave the stage back): keep feeding predicated trials
    PassArgs a; int grid;
    const bool bailed = c->h_state->lmp_bailed != 0;
    if (bailed) c->n_persist_bails++;
    if ((rc = prepare_pass(c, a, grid))) return rc;
    if ((rc = run_stage(c, a, grid, 1, 8, bailed))) return rc;
  }
  c->have_corr = true;
  if (!c->h_state->error) update_hint(c->hint_rot, c->win_rot, c->h_state->rot_passes, lm_fused(c));
  fill_rot_outputs(c->h_state, Tf, Td, stats);
  if (c->h_state->error) { g_err = c->h_state->error == ROLO_ENOCORR ? "no correspondences" : "device-side error during align"; return c->h_state->error; }
  return ROLO_OK;
}

int rolo_compute_translation(rolo_ctx* c, double* trans, const double* g3, const double* l3, double dtn, double dtn1, float lam, rolo_stats* stats) {
  if (!c || !trans || !g3 || !l3) return ROLO_EINVAL;
  if (!c->have_corr) { g_err = "computeTranslation needs the correspondences of a previous align"; return ROLO_ENOCORR; }
  int rc = set_device(c); if (rc) return rc;
  PassArgs a; int grid;
  if ((rc = prepare_pass(c, a, grid))) return rc;
  TransBegin tb{};
  for (int i = 0; i < 3; i++) { tb.t0[i] = trans[i]; tb.g[i] = g3[i]; tb.l[i] = l3[i]; }
  tb.dtn = dtn; tb.dtn1 = dtn1; tb.ct_lambda = lam; tb.direct = 1; fill_trans_knobs(c, tb);
  HIPCHK(launch_trans_begin(c->state, tb, c->stream));
  if ((rc = run_stage(c, a, grid, 2, 12))) return rc;
  const LmState* s = c->h_state;
  for (int i = 0; i < 3; i++) trans[i] = s->t0[i];
  if (stats) { stats->n_outer = s->trans_outer; stats->converged = s->trans_failed ? 0 : 1; stats->lm_failed = s->trans_failed; stats->n_passes = s->trans_passes; stats->n_correspondences = s->tr_n_corr; stats->n_cost_only = s->trans_cost_only; }
  if (s->error) { g_err = "device-side error during computeTranslation"; return s->error; }
  return ROLO_OK;
}

// ---- test hook: the controller kernels on scripted pass results (include/rolo_hip.h rolo_lm_script) -------------------------------------------
static int script_stage(rolo_ctx* c, const rolo_lm_script* S, int stage, int dof, int generic_ctrl) {
  if (!S || S->n_outer < 1 || S->n_trial < 1 || !S->lin_y || !S->lin_H || !S->lin_b || !S->lin_n || !S->err_y) { g_err = "bad LM script"; return ROLO_EINVAL; }
  int rc = ensure(c->partials, c->partials_cap, (size_t)NV_MAX);
  if (rc) return rc;
  const int hard_cap = (std::max(c->P.max_iterations, c->P.fixed_iterations) + 2) * (std::max(c->P.lm_max_iterations, 0) + 2) + 8;
  for (int it = 0; it <= hard_cap; it++) {
    if ((rc = fetch_state(c))) return rc;
    const LmState* s = c->h_state;
    if (stage == 1 ? s->rot_done != 0 : s->trans_done != 0) return ROLO_OK;
    double row[NV_MAX];
    for (double& v : row) v = __builtin_nan("");
    auto put_lin = [&](int o) {
      o = std::min(std::max(o, 0), S->n_outer - 1);
      row[V_Y] = S->lin_y[o]; row[V_N] = (double)S->lin_n[o];
      int t = 0;
      for (int i = 0; i < dof; i++) for (int j = 0; j <= i; j++) row[V_H + t++] = S->lin_H[(size_t)o * 36 + i * 6 + j];
      for (int i = 0; i < dof; i++) row[V_B + i] = S->lin_b[(size_t)o * 6 + i];
    };
    if (s->phase == 0) put_lin(s->outer);   // a linearise-only pass: the stage's first, or the one after a trial accepted on a cost-only pass
    else {
      row[V_YI] = S->err_y[(size_t)std::min(std::max(s->outer, 0), S->n_outer - 1) * S->n_trial + std::min(std::max(s->trial, 0), S->n_trial - 1)];
      if (!s->lin_skip) put_lin(s->outer + 1);   // half (B) of a full pass: the linearisation at the trial pose = the one that opens the next outer iteration
    }
    HIPCHK(hipMemcpyAsync(c->partials, row, sizeof(row), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_ctrl(c->state, c->partials, 1, nullptr, c->trace, stage, c->stream, nullptr, nullptr, generic_ctrl ? 0 : dof));
  }
  g_err = "scripted LM stage did not terminate";
  return ROLO_ESTATE;
}
extern "C" int rolo_debug_lm_script_align(rolo_ctx* c, const rolo_lm_script* S, const float* guess16, int generic_ctrl, float* Tf, double* Td, rolo_stats* stats) {
  if (!c) return ROLO_EINVAL;
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  double R[9], t[3]; guess_to_Rt(guess16, R, t);
  HIPCHK(launch_rot_begin(c->state, make_rot_begin(c, R, t, 0), c->stream));
  if ((rc = script_stage(c, S, 1, c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6, generic_ctrl))) return rc;
  c->have_corr = c->h_state->tr_n_corr > 0;   // what computeTranslation asks for: a linearisation that left correspondences
  fill_rot_outputs(c->h_state, Tf, Td, stats);
  return c->h_state->error;
}
extern "C" int rolo_debug_lm_script_translation(rolo_ctx* c, const rolo_lm_script* S, double* trans, const double* g3, const double* l3, double dtn, double dtn1, float lam,
                                                int generic_ctrl, rolo_stats* stats) {
  if (!c || !trans || !g3 || !l3) return ROLO_EINVAL;
  if (!c->have_corr) { g_err = "computeTranslation needs the correspondences of a previous align"; return ROLO_ENOCORR; }
  int rc = set_device(c); if (rc) return rc;
  TransBegin tb{};
  for (int i = 0; i < 3; i++) { tb.t0[i] = trans[i]; tb.g[i] = g3[i]; tb.l[i] = l3[i]; }
  tb.dtn = dtn; tb.dtn1 = dtn1; tb.ct_lambda = lam; tb.direct = 1; fill_trans_knobs(c, tb);
  HIPCHK(launch_trans_begin(c->state, tb, c->stream));
  if ((rc = script_stage(c, S, 2, 6, generic_ctrl))) return rc;
  const LmState* s = c->h_state;
  for (int i = 0; i < 3; i++) trans[i] = s->t0[i];
  if (stats) { stats->n_outer = s->trans_outer; stats->converged = s->trans_failed ? 0 : 1; stats->lm_failed = s->trans_failed; stats->n_passes = s->trans_passes; stats->n_correspondences = s->tr_n_corr; stats->n_cost_only = s->trans_cost_only; }
  return s->error;
}

// everything of one frame after the clouds are on the device; per-frame arguments come from c->h_args (pinned)
static bool stamp_env() { static const bool v = [] { const char* e = getenv("ROLO_STAMP"); return e && atoi(e) != 0; }(); return v; }
#define STAMP(slot) do { if (stamp_env()) HIPCHK(launch_stamp(c->stamps, slot, c->stream)); } while (0)

static int enqueue_frame(rolo_ctx* c, bool with_trans) {   // with_trans = false: the rotation stage alone (rolo_align as one enqueue)
  int rc;
  if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "source/target not set"; return ROLO_ESTATE; }
  if (stamp_env()) {
    if ((rc = ensure(c->stamps, c->stamps_cap, 8))) return rc;
    if (!c->h_stamps) HIPCHK(hipHostMalloc((void**)&c->h_stamps, sizeof(unsigned long long) * 8));
  }
  STAMP(0);
  // voxel map without the host round trip of ensure_map(): errors are picked up in rolo_register_wait. The table is sized first: when
  // the target's covariances are about to be computed (and are bounded), the search's own launches build the map (VoxelFuse).
  {
    const int n = c->tgt.n;
    size_t capslots = 1024; while (capslots < 2 * (size_t)n) capslots <<= 1;
    if ((rc = ensure(c->tab.keys, c->tab_keys_cap, 2 * capslots))) return rc;   // 16 bytes per slot: key + id
    if ((rc = ensure(c->tab.rec, c->tab_rec_cap, (size_t)n * REC_DOUBLES))) return rc;
    if ((rc = ensure(c->tab.id_keys, c->tab_idk_cap, (size_t)n))) return rc;
    if ((rc = ensure(c->tgt_keys, c->tgt_keys_cap, (size_t)n))) return rc;
    if ((rc = ensure(c->tgt_slot, c->tgt_slot_cap, (size_t)n + KNN_LEAF))) return rc;
    if ((rc = ensure(c->counters, c->counters_cap, 4))) return rc;
    c->tab.mask = (unsigned)(capslots - 1);
    fill_table_params(c);
    c->vf_done = false;
    c->vf = VoxelFuse{};
    if (!c->tgt.have_cov && voxel_fuse_env() && knn_voxel_fuse_supported()) {
      c->tgt.cov_user = false;   // about to be computed here
      if (voxel_fixed_cov(c)) { c->vf.enabled = 1; c->vf.tab = c->tab; c->vf.tgt_keys = c->tgt_keys; c->vf.tgt_slot = c->tgt_slot; c->vf.counters = c->counters; }
    }
    rc = ensure_covs(c);
    c->vf.enabled = 0;
    if (rc) return rc;
    STAMP(1);
    { ProfScope ps(c, ROLO_PROF_VOXEL_BUILD); HIPCHK(launch_voxel_build(c->tgt, c->tab, c->tgt_keys, c->tgt_slot, c->counters, voxel_morton_order(c), voxel_fixed_cov(c), c->tgt.bbox6, c->vf_done, c->stream, c->h_counters, c->state, c->h_args)); }   // the finalize kernel leaves the counters in pinned memory and starts the frame's LM state from c->h_args (pinned)
    c->vf_done = false;
  }
  PassArgs a; int grid;
  if ((rc = prepare_pass(c, a, grid))) return rc;
  // (the LM state of the frame was started by the voxel map's finalize kernel above: frame_begin_kernel was a launch of its own until round 3)
  STAMP(2);
  int nrot, ntrans; frame_chunks(c, nrot, ntrans);
  if (!with_trans) ntrans = 0;
  if (lm_persist(c)) {
    if ((rc = enqueue_lm_persist(c, a, true))) return rc;   // one launch for both stages; it leaves the state in pinned memory
  } else if (lm_fused(c)) {
    // both stages are the same launches (the device decides which pass a launch evaluates); each hint carries one spare
    if ((rc = enqueue_lm_chunk(c, a, std::max(nrot + ntrans - 1, 2), true))) return rc;   // the closing launch leaves the state in pinned memory
  } else {
    for (int i = 0; i < nrot; i++) if ((rc = enqueue_pass(c, a, grid, 1))) return rc;
    STAMP(3);
    for (int i = 0; i < ntrans; i++) if ((rc = enqueue_pass(c, a, grid, 2, i + 1 == ntrans && !c->comm))) return rc;   // the last controller publishes the state
  }
  STAMP(4);
  if (stamp_env()) HIPCHK(hipMemcpyAsync(c->h_stamps, c->stamps, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, c->stream));
  if (!lm_persist(c) && ((c->comm && !lm_fused(c)) || (!lm_fused(c) && ntrans == 0))) HIPCHK(hipMemcpyAsync(c->h_state, c->state, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
  return ROLO_OK;
}

extern "C" int rolo_debug_stamps(rolo_ctx* c, unsigned long long* out8) {   // after rolo_register_wait; zeros unless ROLO_STAMP=1
  if (!c || !out8) return ROLO_EINVAL;
  if (c->h_stamps) memcpy(out8, c->h_stamps, sizeof(unsigned long long) * 8); else memset(out8, 0, sizeof(unsigned long long) * 8);
  return ROLO_OK;
}

static int register_async_impl(rolo_ctx* c, const float* guess16, const double* trans_start, const double* g3, const double* l3, double dtn, double dtn1, float lam) {
  if (!c || !g3 || !l3) return ROLO_EINVAL;
  if (c->async_pending) { g_err = "a registration is already in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "source/target not set"; return ROLO_ESTATE; }
  double R[9], t[3]; guess_to_Rt(guess16, R, t);
  c->h_args->rot = make_rot_begin(c, R, t, 1);
  TransBegin& tb = c->h_args->trans;
  for (int i = 0; i < 3; i++) { tb.t0[i] = trans_start ? trans_start[i] : 0.0; tb.g[i] = g3[i]; tb.l[i] = l3[i]; }
  tb.dtn = dtn; tb.dtn1 = dtn1; tb.ct_lambda = lam; tb.direct = 0; fill_trans_knobs(c, tb);

  // hipGraph: the schedule of a frame is fixed (predicated launches), so with unchanged sizes / buffers / parameters
  // the ~95 launches are captured once and replayed with one hipGraphLaunch (host cost 0.35 ms -> ~0.02 ms per frame)
  const bool graphable = c->P.use_graph && !c->prof_on && !c->comm && !c->want_knn_lists && !c->src.have_cov && !c->tgt.have_cov;
  if (graphable) {
    rolo_ctx::GraphKey key{};
    key.n_src = c->src.n; key.n_tgt = c->tgt.n; key.src_xyz = c->src.xyz; key.tgt_xyz = c->tgt.xyz; key.P = c->P; key.epoch = g_alloc_epoch;
    key.rank = c->rank; key.world = c->world;   // the captured launches bake the shard range in
    key.busy = c->device_busy ? 1 : 0;          // ... and the walk kernel picked by the device's load
    frame_chunks(c, key.nrot, key.ntrans);
    auto same = [](const rolo_ctx::GraphKey& a, const rolo_ctx::GraphKey& b) {
      return a.n_src == b.n_src && a.n_tgt == b.n_tgt && a.src_xyz == b.src_xyz && a.tgt_xyz == b.tgt_xyz && a.epoch == b.epoch &&
             a.nrot == b.nrot && a.ntrans == b.ntrans && a.rank == b.rank && a.world == b.world && a.busy == b.busy && memcmp(&a.P, &b.P, sizeof(rolo_params)) == 0;
    };
    if (c->galt.exec && same(key, c->galt.key) && !(c->graph_exec && same(key, c->gkey))) {   // the other regime's frame is cached: swap
      std::swap(c->graph, c->galt.graph); std::swap(c->graph_exec, c->galt.exec); std::swap(c->gkey, c->galt.key); std::swap(c->graph_nrm_written, c->galt.nrm);
    }
    if (c->graph_exec && same(key, c->gkey)) {
      HIPCHK(hipGraphLaunch(c->graph_exec, c->stream));
      c->n_replays++;
      c->src.have_cov = true; c->tgt.have_cov = true; c->src.have_sorted = true; c->tgt.have_sorted = true;
      c->src.have_nrm = c->tgt.have_nrm = c->graph_nrm_written;   // what build_clouds decided when this graph was captured (recorded then, not re-derived: advisor, round 5)
      c->async_pending = true;
      return ROLO_OK;
    }
    if (c->gseen_valid && same(key, c->gseen)) {
      if (c->graph_exec && c->gkey.busy != key.busy) {   // keep the other regime's frame in the second slot (whatever was there goes)
        if (c->galt.exec) (void)hipGraphExecDestroy(c->galt.exec);
        if (c->galt.graph) (void)hipGraphDestroy(c->galt.graph);
        c->galt.graph = c->graph; c->galt.exec = c->graph_exec; c->galt.key = c->gkey; c->galt.nrm = c->graph_nrm_written;
        c->graph = nullptr; c->graph_exec = nullptr;
      }
      if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
      if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
      HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      rc = enqueue_frame(c, true);
      hipGraph_t gph = nullptr;
      hipError_t e = hipStreamEndCapture(c->stream, &gph);
      const bool epoch_moved = key.epoch != g_alloc_epoch;  // an allocation inside the capture would be a bug; fall back
      if (rc == ROLO_OK && e == hipSuccess && gph && !epoch_moved && hipGraphInstantiate(&c->graph_exec, gph, nullptr, nullptr, 0) == hipSuccess) {
        c->graph = gph; c->gkey = key;
        c->graph_nrm_written = c->src.have_nrm && c->tgt.have_nrm;
        HIPCHK(hipGraphLaunch(c->graph_exec, c->stream));
        c->n_captures++;
        c->async_pending = true;
        return ROLO_OK;
      }
      if (gph) (void)hipGraphDestroy(gph);
      c->graph_exec = nullptr;
      (void)hipGetLastError();
      c->src.have_cov = false; c->tgt.have_cov = false;  // nothing ran: redo eagerly below
      c->gseen_valid = false;
    } else {
      c->gseen = key; c->gseen_valid = true;
    }
  }
  if ((rc = enqueue_frame(c, true))) return rc;
  if (graphable) c->gseen.epoch = g_alloc_epoch;  // the eager frame did the allocations the capture must not do
  c->n_eager++;
  c->async_pending = true;
  return ROLO_OK;
}

int rolo_register_async(rolo_ctx* c, const float* guess16, const double* trans_start, const double* g3, const double* l3, double dtn, double dtn1, float lam) {
  if (c) {
    if (others_in_flight(c)) c->busy_credit = 8; else if (c->busy_credit > 0) c->busy_credit--;
    // (ranks that share ONE frame — peers, a communicator, a shard range — always have each other's frames "in flight": that is cooperation, not load)
    const bool sharded_ctx = c->comm != nullptr || peers(c) || c->world > 1;
    c->device_busy = c->load_hint < 0 ? (!sharded_ctx && c->busy_credit > 0) : c->load_hint != 0;
    // nobody of this process in flight: the learner decides (it may know of load this process cannot count)
    c->frame_auto_idle = c->load_hint < 0 && !sharded_ctx && c->busy_credit == 0 && !c->async_pending;
    if (c->frame_auto_idle) {
      c->learn.sizes(c->src.n, c->tgt.n);
      c->device_busy = c->learn.busy();
    }
    if (!c->async_pending && set_device(c) == ROLO_OK) (void)hipEventRecord(c->ev_start, c->stream);
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = register_async_impl(c, guess16, trans_start, g3, l3, dtn, dtn1, lam);
  if (rc == ROLO_OK && c->async_pending) { c->n_frames++; count_in_flight(c, true); HIPCHK(hipEventRecord(c->ev_done, c->stream)); }
  if (c) c->ns_enqueue += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int rolo_register_wait(rolo_ctx* c, float* Tf, double* Td, double* trans_out, rolo_stats* rs, rolo_stats* ts) {
  if (!c) return ROLO_EINVAL;
  if (!c->async_pending) { g_err = "no registration in flight"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  c->async_pending = false;
  count_in_flight(c, false);
  c->device_busy = c->load_hint == 1;   // the choice belongs to the frame that was enqueued: synchronous entry points (rolo_compute_covariances, rolo_align, ...) run alone
  const auto tw0 = std::chrono::steady_clock::now();
  HIPCHK(hipEventSynchronize(c->ev_done));
  const auto tw1 = std::chrono::steady_clock::now();
  c->ns_wait_blocked += std::chrono::duration_cast<std::chrono::nanoseconds>(tw1 - tw0).count();
  if (c->frame_auto_idle) {   // the learner's signal: how long the frame took on the device (rolo_ctx::LoadLearner)
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_done) == hipSuccess && ms > 0.f) {
      c->learn.frame((double)ms);
    } else (void)hipGetLastError();
  }
  struct WaitTimer { rolo_ctx* c; std::chrono::steady_clock::time_point t; ~WaitTimer() { c->ns_wait_other += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } } wt{c, tw1};
  if ((rc = peer_check(c))) return rc;
  if (c->h_counters[1] != 0) { g_err = c->h_counters[1] == ROLO_ENONFINITE ? "non-finite point or covariance in the voxel map build" : "voxel coordinate outside the packed key range"; return c->h_counters[1]; }
  c->n_voxels = c->h_counters[0];
  c->n_edge = c->h_counters[2];
  c->have_map = true;
  PassArgs a; int grid;
  if ((rc = prepare_pass(c, a, grid))) return rc;
  // the common case finished inside the first enqueue; otherwise keep feeding predicated passes
  if (!c->h_state->rot_done || (!c->h_state->trans_done && !c->h_state->error)) c->n_topup_frames++;   // the first schedule was too short: host round trips
  const bool bailed = c->h_state->lmp_bailed != 0;   // (read before the top-ups overwrite the host copy)
  if (bailed) { c->n_persist_bails++; c->busy_credit = 64; }   // somebody this process cannot see shares the GPU (another process, a foreign workload): the busy-device sizing for the next frames
  if (!c->h_state->rot_done) { if ((rc = run_stage(c, a, grid, 1, 8, bailed))) return rc; }
  if (!c->h_state->trans_done && !c->h_state->error) { if ((rc = run_stage(c, a, grid, 2, 8, bailed))) return rc; }
  c->have_corr = true;
  const LmState* s = c->h_state;
  if (!s->error) { update_hint(c->hint_rot, c->win_rot, s->rot_passes, lm_fused(c)); update_hint(c->hint_trans, c->win_trans, s->trans_passes, lm_fused(c)); }
  fill_rot_outputs(s, Tf, Td, rs);
  if (trans_out) for (int i = 0; i < 3; i++) trans_out[i] = s->t0[i];
  if (ts) { ts->n_outer = s->trans_outer; ts->converged = s->trans_failed ? 0 : 1; ts->lm_failed = s->trans_failed; ts->n_passes = s->trans_passes; ts->n_correspondences = s->tr_n_corr; ts->n_cost_only = s->trans_cost_only; }
  if (s->error) { g_err = "device-side error during registration"; return s->error; }
  return ROLO_OK;
}

// experiment hook (include/rolo_hip.h): a captured chain of launch pairs, replayed `reps` times
int rolo_debug_chain(rolo_ctx* c, int kind, int n_pairs, int grid, int reps) {
  if (!c || kind < 0 || kind > 4 || n_pairs < 1 || n_pairs > 256 || grid < 1 || reps < 1) return ROLO_EINVAL;
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  if (kind >= 3 && (!c->have_map || !c->src.have_cov)) { g_err = "the real LM chain needs a finished registration on this context"; return ROLO_ESTATE; }
  if (!c->dbg_chain_exec || c->dbg_chain_key[0] != kind || c->dbg_chain_key[1] != n_pairs || c->dbg_chain_key[2] != grid) {
    if (c->dbg_chain_exec) { (void)hipGraphExecDestroy(c->dbg_chain_exec); c->dbg_chain_exec = nullptr; }
    PassArgs a; int pgrid = 0;
    if (kind >= 3 && (rc = prepare_pass(c, a, pgrid))) return rc;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipSuccess;
    const int dof = c->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6;
    if (kind >= 3) e = launch_frame_begin(c->state, c->h_args, c->stream);
    for (int i = 0; i < n_pairs && e == hipSuccess; i++) {
      if (kind == 0) { e = launch_empty(1, 256, c->stream); if (e == hipSuccess) e = launch_empty(1, 256, c->stream); }
      else if (kind == 1) { e = launch_empty(grid, 256, c->stream); if (e == hipSuccess) e = launch_empty(grid, 256, c->stream); }
      else if (kind == 2) { e = launch_empty(grid, 256, c->stream); if (e == hipSuccess) e = launch_empty(1, 256, c->stream); }
      else if (kind == 3) {   // the first two thirds of the pairs belong to the rotation stage, the rest to the translation stage (a frame's 21 + 10)
        const int stage = i < (2 * n_pairs + 2) / 3 ? 1 : 2;
        e = stage == 1 ? launch_rot_pass(dof, a, c->state, pgrid, c->stream) : launch_trans_pass(a, c->state, pgrid, c->stream);
        if (e == hipSuccess) e = launch_ctrl(c->state, c->partials, pgrid, nullptr, c->trace, stage, c->stream, nullptr, nullptr, dof);
      } else { e = launch_rot_pass(dof, a, c->state, pgrid, c->stream); if (e == hipSuccess) e = launch_rot_pass(dof, a, c->state, pgrid, c->stream); }
    }
    hipGraph_t gph = nullptr;
    const hipError_t e2 = hipStreamEndCapture(c->stream, &gph);
    if (e != hipSuccess || e2 != hipSuccess || !gph || hipGraphInstantiate(&c->dbg_chain_exec, gph, nullptr, nullptr, 0) != hipSuccess) {
      if (gph) (void)hipGraphDestroy(gph);
      c->dbg_chain_exec = nullptr; (void)hipGetLastError();
      g_err = "rolo_debug_chain: capture failed"; return ROLO_EHIP;
    }
    (void)hipGraphDestroy(gph);
    c->dbg_chain_key[0] = kind; c->dbg_chain_key[1] = n_pairs; c->dbg_chain_key[2] = grid;
  }
  for (int r = 0; r < reps; r++) HIPCHK(hipGraphLaunch(c->dbg_chain_exec, c->stream));
  return ROLO_OK;
}

int rolo_get_final_hessian(rolo_ctx* c, double* H36) {
  if (!c || !H36) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  if ((rc = fetch_state(c))) return rc;
  memcpy(H36, c->h_state->final_H, sizeof(double) * 36);
  return ROLO_OK;
}

int rolo_get_trace(rolo_ctx* c, rolo_trace_rec* out, int cap) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  if ((rc = fetch_state(c))) return rc;
  const int n = std::min(c->h_state->trace_count, TRACE_CAP);
  const int m = std::min(n, cap);
  if (m > 0 && out) {
    HIPCHK(hipMemcpyAsync(out, c->trace, sizeof(rolo_trace_rec) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return n;
}

int rolo_transform_cloud(rolo_ctx* c, const float* in, float* out, int n, int stride, const float* T16) {
  if (!c || !in || !out || !T16 || n < 0 || stride < 3) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  if (n == 0) return ROLO_OK;
  if ((rc = ensure(c->stage_in, c->stage_in_cap, (size_t)n * stride))) return rc;
  if ((rc = ensure(c->stage_out, c->stage_out_cap, (size_t)n * stride))) return rc;
  HIPCHK(hipMemcpyAsync(c->stage_in, in, sizeof(float) * (size_t)n * stride, hipMemcpyHostToDevice, c->stream));
  HIPCHK(launch_transform_cloud(c->stage_in, c->stage_out, n, stride, nullptr, T16, c->stream));
  HIPCHK(hipMemcpyAsync(out, c->stage_out, sizeof(float) * (size_t)n * stride, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ROLO_OK;
}

int rolo_ctx_counters(rolo_ctx* c, long long* out, int n) {
  if (!c || !out || n < 0) return ROLO_EINVAL;
  const long long v[14] = {c->n_frames, c->n_replays, c->n_captures, c->n_eager, c->n_topup_frames, c->n_topup_chunks, c->hint_rot, c->hint_trans, c->walk_lanes,
                           c->ns_enqueue, c->ns_wait_blocked, c->ns_wait_other, c->n_persist_bails, c->learn.mode};
  for (int i = 0; i < n && i < 14; i++) out[i] = v[i];
  return ROLO_OK;
}

long long rolo_alloc_count(void) { return (long long)g_alloc_epoch.load(); }

int rolo_prof_enable(rolo_ctx* c, int on) {
  if (!c) return ROLO_EINVAL;
  c->prof_on = on != 0;
  return ROLO_OK;
}

int rolo_prof_read(rolo_ctx* c, int slot, float* ms, int cap) {
  if (!c || slot < 0 || slot >= ROLO_PROF_N) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  int n = 0;
  std::vector<rolo_ctx::ProfEv> keep;
  for (auto& e : c->prof) {
    if (e.slot != slot) { keep.push_back(e); continue; }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e.a, e.b);
    if (ms && n < cap) ms[n] = t;
    n++;
    (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
  }
  c->prof.swap(keep);
  return n;
}

// ---- batches of independent scan pairs (BASELINE config 5) -------------------------------------------------------
struct rolo_batch {
  int device = 0, n = 0;
  std::vector<rolo_ctx*> m;          // members: ordinary contexts; their own streams carry the per-member front work
  hipStream_t stream = nullptr;      // the batched LM chain runs here
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;   // member front work done (recorded inside the captured schedule)
  std::vector<hipEvent_t> ev_in;     // member stream's earlier work (cloud packing) done — recorded outside the graph
  BatchSlot* h_slots = nullptr;      // pinned
  BatchSlot* d_slots = nullptr;
  FrameArgs* h_args = nullptr;       // pinned, n entries
  FrameArgs* d_args = nullptr;
  bool pending = false;
  int bps = 0;
  // hipGraph of the whole batch
  struct Key { std::vector<rolo_ctx::GraphKey> k; } gkey, gseen;
  bool gseen_valid = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
};

static bool same_key(const rolo_ctx::GraphKey& a, const rolo_ctx::GraphKey& b) {
  return a.n_src == b.n_src && a.n_tgt == b.n_tgt && a.src_xyz == b.src_xyz && a.tgt_xyz == b.tgt_xyz && a.epoch == b.epoch &&
         memcmp(&a.P, &b.P, sizeof(rolo_params)) == 0;
}
static bool same_keys(const std::vector<rolo_ctx::GraphKey>& a, const std::vector<rolo_ctx::GraphKey>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) if (!same_key(a[i], b[i])) return false;
  return true;
}

int rolo_batch_create(int device, int n_members, rolo_batch** out) {
  if (!out || n_members < 1 || n_members > 64) return ROLO_EINVAL;
  rolo_batch* b = new rolo_batch();
  b->device = device; b->n = n_members;
  for (int i = 0; i < n_members; i++) {
    rolo_ctx* c = nullptr;
    int rc = rolo_ctx_create(device, &c);
    if (rc) { rolo_batch_destroy(b); return rc; }
    b->m.push_back(c);
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { g_err = "event creation failed"; rolo_batch_destroy(b); return ROLO_EHIP; }
    b->ev_join.push_back(e);
    e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { g_err = "event creation failed"; rolo_batch_destroy(b); return ROLO_EHIP; }
    b->ev_join.push_back(e);
    e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { g_err = "event creation failed"; rolo_batch_destroy(b); return ROLO_EHIP; }
    b->ev_in.push_back(e);
  }
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void**)&b->h_slots, sizeof(BatchSlot) * n_members) != hipSuccess || hipHostMalloc((void**)&b->h_args, sizeof(FrameArgs) * n_members) != hipSuccess ||
      hipMalloc((void**)&b->d_slots, sizeof(BatchSlot) * n_members) != hipSuccess || hipMalloc((void**)&b->d_args, sizeof(FrameArgs) * n_members) != hipSuccess) {
    g_err = "batch allocation failed"; rolo_batch_destroy(b); return ROLO_EHIP;
  }
  *out = b;
  return ROLO_OK;
}

void rolo_batch_destroy(rolo_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
  if (b->graph) (void)hipGraphDestroy(b->graph);
  for (rolo_ctx* c : b->m) rolo_ctx_destroy(c);
  for (hipEvent_t e : b->ev_join) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev_in) (void)hipEventDestroy(e);
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  if (b->h_slots) (void)hipHostFree(b->h_slots);
  if (b->h_args) (void)hipHostFree(b->h_args);
  if (b->d_slots) (void)hipFree(b->d_slots);
  if (b->d_args) (void)hipFree(b->d_args);
  delete b;
}

int rolo_batch_size(rolo_batch* b) { return b ? b->n : 0; }
rolo_ctx* rolo_batch_member(rolo_batch* b, int i) { return (b && i >= 0 && i < b->n) ? b->m[i] : nullptr; }

// everything of one batch step after the clouds are on the device
static int enqueue_batch(rolo_batch* b, bool fork) {
  rolo_ctx* c0 = b->m[0];
  hipStream_t st = b->stream;
  int rc, bps = 1;
  // Eager launches (fork): the per-member front work fans out over the members' two streams — source search on one,
  // target search + voxel map on the other; one search launch fills about half the chip — and joins the batch stream
  // before the shared LM chain. Both streams fork directly from the batch stream. Captured schedule (!fork): a single
  // branch on the batch stream. ROCm 7.2's graph runtime is not safe with forked captures here: a fork of a fork sends
  // hipStreamEndCapture into an endless recursion, and several multi-branch graphs in flight crashed hipGraphLaunch
  // (hip::Graph::UpdateStreams); overlap between frames then comes from keeping several batches in flight.
  if (fork) HIPCHK(hipEventRecord(b->ev_fork, st));
  for (int i = 0; i < b->n; i++) {
    rolo_ctx* c = b->m[i];
    const hipStream_t s1 = fork ? c->stream : st, s2 = fork ? c->stream2 : st;
    if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "source/target not set"; return ROLO_ESTATE; }
    if (fork) { HIPCHK(hipStreamWaitEvent(s1, b->ev_fork, 0)); HIPCHK(hipStreamWaitEvent(s2, b->ev_fork, 0)); }
    if (!c->src.have_cov && (rc = build_src(c, s1))) return rc;
    if (!c->tgt.have_cov && (rc = build_tgt(c, s2))) return rc;
    const int n = c->tgt.n;
    size_t capslots = 1024; while (capslots < 2 * (size_t)n) capslots <<= 1;
    if ((rc = ensure(c->tab.keys, c->tab_keys_cap, 2 * capslots))) return rc;   // 16 bytes per slot: key + id
    if ((rc = ensure(c->tab.rec, c->tab_rec_cap, (size_t)n * REC_DOUBLES))) return rc;
    if ((rc = ensure(c->tab.id_keys, c->tab_idk_cap, (size_t)n))) return rc;
    if ((rc = ensure(c->tgt_keys, c->tgt_keys_cap, (size_t)n))) return rc;
    if ((rc = ensure(c->tgt_slot, c->tgt_slot_cap, (size_t)n + KNN_LEAF))) return rc;
    if ((rc = ensure(c->counters, c->counters_cap, 4))) return rc;
    c->tab.mask = (unsigned)(capslots - 1);
    fill_table_params(c);
    HIPCHK(launch_voxel_build(c->tgt, c->tab, c->tgt_keys, c->tgt_slot, c->counters, voxel_morton_order(c), voxel_fixed_cov(c), c->tgt.bbox6, false, s2));
    HIPCHK(hipMemcpyAsync(c->h_counters, c->counters, 4 * sizeof(int), hipMemcpyDeviceToHost, s2));
    if (fork) { HIPCHK(hipEventRecord(b->ev_join[2 * i], s1)); HIPCHK(hipEventRecord(b->ev_join[2 * i + 1], s2)); }
    PassArgs a; int grid;
    if ((rc = prepare_pass(c, a, grid))) return rc;
    BatchSlot& S = b->h_slots[i];
    S.a = a; S.st = c->state; S.trace = c->trace; S.grid = grid; S.pad = 0;
    bps = std::max(bps, grid);
  }
  b->bps = bps;
  if (fork) for (hipEvent_t e : b->ev_join) HIPCHK(hipStreamWaitEvent(st, e, 0));
  HIPCHK(hipMemcpyAsync(b->d_slots, b->h_slots, sizeof(BatchSlot) * b->n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_args, b->h_args, sizeof(FrameArgs) * b->n, hipMemcpyHostToDevice, st));
  HIPCHK(launch_batch_begin(b->d_slots, b->d_args, b->n, st));
  const int dof = c0->P.optimizer == ROLO_OPT_SO3_LM ? 3 : 6;
  const int nrot = rot_first_chunk(c0);
  for (int k = 0; k < nrot; k++) { HIPCHK(launch_batch_pass(1, dof, b->d_slots, b->n, bps, st)); HIPCHK(launch_batch_ctrl(1, b->d_slots, b->n, st)); }
  for (int k = 0; k < 12; k++) { HIPCHK(launch_batch_pass(2, dof, b->d_slots, b->n, bps, st)); HIPCHK(launch_batch_ctrl(2, b->d_slots, b->n, st)); }
  for (int i = 0; i < b->n; i++) HIPCHK(hipMemcpyAsync(b->m[i]->h_state, b->m[i]->state, sizeof(LmState), hipMemcpyDeviceToHost, st));
  return ROLO_OK;
}

int rolo_batch_register_async(rolo_batch* b, const float* guess16, const double* trans_start, const double* init_guess, const double* last_t0,
                              double dtn, double dtn1, float lam) {
  if (!b || !init_guess || !last_t0) return ROLO_EINVAL;
  if (b->pending) { g_err = "a batch registration is already in flight"; return ROLO_ESTATE; }
  int rc = set_device(b->m[0]); if (rc) return rc;
  bool graphable = true;
  std::vector<rolo_ctx::GraphKey> keys(b->n);
  for (int i = 0; i < b->n; i++) {
    rolo_ctx* c = b->m[i];
    if (c->src.n <= 0 || c->tgt.n <= 0) { g_err = "batch member without source/target"; return ROLO_ESTATE; }
    if (c->comm || peers(c) || c->async_pending) { g_err = "batch members must be idle single-GPU contexts"; return ROLO_ESTATE; }
    if (c->P.optimizer != b->m[0]->P.optimizer || c->P.fixed_iterations != b->m[0]->P.fixed_iterations) { g_err = "batch members must share optimizer and iteration settings"; return ROLO_EUNSUPPORTED; }
    double R[9], t[3]; guess_to_Rt(guess16 ? guess16 + 16 * (size_t)i : nullptr, R, t);
    b->h_args[i].rot = make_rot_begin(c, R, t, 1);
    TransBegin& tb = b->h_args[i].trans;
    for (int d = 0; d < 3; d++) { tb.t0[d] = trans_start ? trans_start[3 * i + d] : 0.0; tb.g[d] = init_guess[3 * i + d]; tb.l[d] = last_t0[3 * i + d]; }
    tb.dtn = dtn; tb.dtn1 = dtn1; tb.ct_lambda = lam; tb.direct = 0; fill_trans_knobs(c, tb);
    graphable = graphable && c->P.use_graph && !c->prof_on && !c->want_knn_lists && !c->src.have_cov && !c->tgt.have_cov;
    keys[i].n_src = c->src.n; keys[i].n_tgt = c->tgt.n; keys[i].src_xyz = c->src.xyz; keys[i].tgt_xyz = c->tgt.xyz; keys[i].P = c->P; keys[i].epoch = g_alloc_epoch;
  }
  hipStream_t st = b->stream;
  // whatever the caller queued on the members' streams (rolo_set_source/_target pack the clouds there) comes first
  for (int i = 0; i < b->n; i++) { HIPCHK(hipEventRecord(b->ev_in[i], b->m[i]->stream)); HIPCHK(hipStreamWaitEvent(st, b->ev_in[i], 0)); }
  if (graphable) {
    if (b->graph_exec && same_keys(keys, b->gkey.k)) {
      HIPCHK(hipGraphLaunch(b->graph_exec, st));
      for (rolo_ctx* c : b->m) { c->src.have_cov = true; c->tgt.have_cov = true; c->src.have_sorted = true; c->tgt.have_sorted = true;
                                 c->src.have_nrm = c->tgt.have_nrm = c->graph_nrm_written; }   // as recorded when the batch's graph was captured
      b->pending = true;
      return ROLO_OK;
    }
    if (b->gseen_valid && same_keys(keys, b->gseen.k)) {
      if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
      if (b->graph) { (void)hipGraphDestroy(b->graph); b->graph = nullptr; }
      HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      rc = enqueue_batch(b, false);
      hipGraph_t gph = nullptr;
      hipError_t e = hipStreamEndCapture(st, &gph);
      const bool epoch_moved = keys[0].epoch != g_alloc_epoch;
      if (rc == ROLO_OK && e == hipSuccess && gph && !epoch_moved && hipGraphInstantiate(&b->graph_exec, gph, nullptr, nullptr, 0) == hipSuccess) {
        b->graph = gph; b->gkey.k = keys;
        for (rolo_ctx* c : b->m) c->graph_nrm_written = c->src.have_nrm && c->tgt.have_nrm;
        HIPCHK(hipGraphLaunch(b->graph_exec, st));
        b->pending = true;
        return ROLO_OK;
      }
      if (gph) (void)hipGraphDestroy(gph);
      b->graph_exec = nullptr;
      (void)hipGetLastError();
      for (rolo_ctx* c : b->m) { c->src.have_cov = false; c->tgt.have_cov = false; }
      b->gseen_valid = false;
    } else {
      b->gseen.k = keys; b->gseen_valid = true;
    }
  }
  if ((rc = enqueue_batch(b, true))) return rc;
  if (graphable) for (auto& k : b->gseen.k) k.epoch = g_alloc_epoch;
  b->pending = true;
  return ROLO_OK;
}

int rolo_batch_register_wait(rolo_batch* b, float* Tf, double* Td, double* trans_out, rolo_stats* rs, rolo_stats* ts) {
  if (!b) return ROLO_EINVAL;
  if (!b->pending) { g_err = "no batch registration in flight"; return ROLO_ESTATE; }
  int rc = set_device(b->m[0]); if (rc) return rc;
  b->pending = false;
  HIPCHK(hipStreamSynchronize(b->stream));
  int first_err = ROLO_OK;
  for (int i = 0; i < b->n; i++) {
    rolo_ctx* c = b->m[i];
    if (c->h_counters[1] != 0) { g_err = c->h_counters[1] == ROLO_ENONFINITE ? "non-finite point or covariance in the voxel map build" : "voxel coordinate outside the packed key range"; if (!first_err) first_err = c->h_counters[1]; continue; }
    c->n_voxels = c->h_counters[0];
    c->n_edge = c->h_counters[2];
    c->have_map = true;
    // a member whose data needed more trials than the fixed schedule holds is finished on its own (rare)
    if (!c->h_state->error && (!c->h_state->rot_done || !c->h_state->trans_done)) {
      PassArgs a; int grid;
      if ((rc = prepare_pass(c, a, grid))) return rc;
      if (!c->h_state->rot_done) { if ((rc = run_stage(c, a, grid, 1, 8))) return rc; }
      if (!c->h_state->trans_done && !c->h_state->error) { if ((rc = run_stage(c, a, grid, 2, 8))) return rc; }
    }
    c->have_corr = true;
    const LmState* s = c->h_state;
    fill_rot_outputs(s, Tf ? Tf + 16 * (size_t)i : nullptr, Td ? Td + 16 * (size_t)i : nullptr, rs ? rs + i : nullptr);
    if (trans_out) for (int d = 0; d < 3; d++) trans_out[3 * i + d] = s->t0[d];
    if (ts) { ts[i].n_outer = s->trans_outer; ts[i].converged = s->trans_failed ? 0 : 1; ts[i].lm_failed = s->trans_failed; ts[i].n_passes = s->trans_passes; ts[i].n_correspondences = s->tr_n_corr; ts[i].n_cost_only = s->trans_cost_only; }
    if (s->error && !first_err) { g_err = "device-side error in a batch member"; first_err = s->error; }
  }
  return first_err;
}

int rolo_set_shard(rolo_ctx* c, int rank, int world) {
  if (!c || world < 1 || rank < 0 || rank >= world) return ROLO_EINVAL;
  c->rank = rank; c->world = world; c->have_corr = false;
  if (c->shard_knn) { c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false; }
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }   // the captured schedule bakes the shard range in
  if (c->galt.exec) { (void)hipGraphExecDestroy(c->galt.exec); c->galt.exec = nullptr; }
  c->gseen_valid = false;
  return ROLO_OK;
}

int rolo_set_load_hint(rolo_ctx* c, int mode) {
  if (!c || mode < -1 || mode > 1) return ROLO_EINVAL;
  c->load_hint = mode;
  return ROLO_OK;
}
int rolo_set_shard_knn(rolo_ctx* c, int on) {
  if (!c) return ROLO_EINVAL;
  c->shard_knn = on != 0;
  c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false; c->have_corr = false;
  return ROLO_OK;
}

int rolo_comm_unique_id(void* uid128) {
  if (!uid128) return ROLO_EINVAL;
  int rc = load_rccl(); if (rc) return rc;
  int e = g_rccl.GetUniqueId(uid128);
  if (e != 0) { g_err = "ncclGetUniqueId failed"; return ROLO_ECOMM; }
  return ROLO_OK;
}

int rolo_comm_init(rolo_ctx* c, const void* uid128, int rank, int world) {
  if (!c || !uid128 || world < 1 || rank < 0 || rank >= world) return ROLO_EINVAL;
  if (c->peer.connected) { g_err = "context is connected to peers (rolo_peer_connect): disconnect first"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  // world == 1 is a real (loopback) communicator too: the single-GPU test drives the whole collective path with it
  if ((rc = load_rccl())) return rc;
  Uid id; memcpy(&id, uid128, sizeof(id));
  int e = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (e != 0) { g_err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?"); return ROLO_ECOMM; }
  c->rank = rank; c->world = world;
  c->have_corr = false; c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false;
  return ROLO_OK;
}

int rolo_comm_info(rolo_ctx* c, int* rank, int* world) {
  if (!c) return ROLO_EINVAL;
  if (!c->comm) { if (rank) *rank = 0; if (world) *world = 0; return ROLO_OK; }   // world 0: no communicator
  int r = -1, w = -1;
  if (!g_rccl.CommCount || !g_rccl.CommUserRank || g_rccl.CommCount(c->comm, &w) != 0 || g_rccl.CommUserRank(c->comm, &r) != 0) { g_err = "ncclCommCount / ncclCommUserRank failed"; return ROLO_ECOMM; }
  if (rank) *rank = r;
  if (world) *world = w;
  return ROLO_OK;
}

int rolo_comm_destroy(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  if (c->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(c->stream); g_rccl.CommDestroy(c->comm); }
  c->comm = nullptr; c->rank = 0; c->world = 1;
  c->have_corr = false; c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false;
  return ROLO_OK;
}

// ---- peer exchange without a collective library -------------------------------------------------------------------
int rolo_peer_export(rolo_ctx* c, int world, int max_points, void* handle64) {
  if (!c || !handle64 || world < 1 || world > PEER_MAX || max_points < 0) { g_err = "rolo_peer_export: bad arguments (1 <= world <= 8)"; return ROLO_EINVAL; }
  if (c->comm) { g_err = "context already holds an RCCL communicator"; return ROLO_ESTATE; }
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  peer_release(c);
  rolo_peer_state& P = c->peer;
  // per area: one segment per rank, a segment = the rank's share of whole 256-query workgroups of both clouds, 6 doubles per position
  P.area_bytes = ((size_t)max_points + (size_t)(2 * 256 + 2 * KNN_LEAF) * world + 512) * 6 * sizeof(double);
  P.area_bytes = (P.area_bytes + 4095) & ~(size_t)4095;
  P.bytes = PEER_STAGE_OFFSET + 2 * P.area_bytes;
  // Fine-grained device memory: what the peers write here while a kernel of this rank polls must not be served from a stale L2 line — the
  // memory type RCCL keeps its flags in; the words of the LM exchange are read with system-scope atomics either way. Ordinary (coarse)
  // device memory is the fall-back when the allocation or its export fails (ROLO_PEER_MEM = finegrained | coarse forces one).
  // NOT hipDeviceMallocUncached: measured on MI355X / ROCm 7.2 — after such an allocation is freed, later ordinary hipMalloc blocks of the
  // same process that land on its pages lose kernel writes (an unrelated context created afterwards read back covariances that were partly
  // zero, differently every run); fine-grained and coarse allocations do not leave that behind.
  const char* want = getenv("ROLO_PEER_MEM");
  hipError_t e = hipErrorUnknown;
  if (!want || !strcmp(want, "finegrained")) { e = hipExtMallocWithFlags(&P.base, P.bytes, hipDeviceMallocFinegrained); P.mem_kind = "finegrained"; }
  if (e != hipSuccess && (!want || !strcmp(want, "coarse"))) { (void)hipGetLastError(); e = hipMalloc(&P.base, P.bytes); P.mem_kind = "coarse"; }
  if (e != hipSuccess) { P.base = nullptr; return fail_hip(e, "peer mailbox allocation"); }
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == ROLO_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  e = hipIpcGetMemHandle(&h, P.base);
  if (e != hipSuccess && strcmp(P.mem_kind, "coarse") != 0 && !want) {   // this allocation kind cannot be exported here: ordinary device memory can
    (void)hipGetLastError(); (void)hipFree(P.base); P.base = nullptr;
    e = hipMalloc(&P.base, P.bytes); P.mem_kind = "coarse";
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, P.base);
  }
  if (e != hipSuccess) { if (P.base) { (void)hipFree(P.base); P.base = nullptr; } return fail_hip(e, "hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); }
  HIPCHK(hipMemsetAsync(P.base, 0, PEER_STAGE_OFFSET, c->stream));
  {  // what the peers check before they push anything here (rolo_peer_connect)
    const unsigned long long hdr[2] = {(unsigned long long)P.area_bytes, (unsigned long long)world};
    HIPCHK(hipMemcpyAsync(static_cast<unsigned long long*>(P.base) + PEER_W_AREA_BYTES, hdr, sizeof(hdr), hipMemcpyHostToDevice, c->stream));
  }
  if (!P.h_err) HIPCHK(hipHostMalloc((void**)&P.h_err, sizeof(int)));
  *P.h_err = 0;
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(P.handle.data(), &h, ROLO_PEER_HANDLE_BYTES);
  { std::lock_guard<std::mutex> lk(g_peer_mu); g_peer_exports[P.handle] = PeerExport{P.base, c->device}; }
  P.export_world = world;
  memcpy(handle64, &h, ROLO_PEER_HANDLE_BYTES);
  return ROLO_OK;
}

int rolo_peer_connect(rolo_ctx* c, const void* handles, int rank, int world) {
  if (!c || !handles || world < 1 || world > PEER_MAX || rank < 0 || rank >= world) return ROLO_EINVAL;
  rolo_peer_state& P = c->peer;
  if (!P.base || P.export_world != world) { g_err = "rolo_peer_connect: call rolo_peer_export with the same world first"; return ROLO_ESTATE; }
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  peer_disconnect_impl(c);
  const char* hb = static_cast<const char*>(handles);
  if (memcmp(hb + (size_t)rank * ROLO_PEER_HANDLE_BYTES, P.handle.data(), ROLO_PEER_HANDLE_BYTES) != 0) { g_err = "rolo_peer_connect: handles[rank] is not this context's export"; return ROLO_EINVAL; }
  for (int r = 0; r < world; r++) {
    std::array<char, ROLO_PEER_HANDLE_BYTES> key; memcpy(key.data(), hb + (size_t)r * ROLO_PEER_HANDLE_BYTES, ROLO_PEER_HANDLE_BYTES);
    if (r == rank) { P.mapped[r] = P.base; continue; }
    PeerExport local{nullptr, -1};
    { std::lock_guard<std::mutex> lk(g_peer_mu); auto it = g_peer_exports.find(key); if (it != g_peer_exports.end()) local = it->second; }
    if (local.base) {   // a context of this process (one process driving several GPUs, or the in-process test)
      if (local.device != c->device) {
        hipError_t e = hipDeviceEnablePeerAccess(local.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { peer_disconnect_impl(c); return fail_hip(e, "hipDeviceEnablePeerAccess"); }
        (void)hipGetLastError();
      }
      P.mapped[r] = local.base;
    } else {
      hipIpcMemHandle_t h; memcpy(&h, key.data(), ROLO_PEER_HANDLE_BYTES);
      void* ptr = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { peer_disconnect_impl(c); return fail_hip(e, "hipIpcOpenMemHandle"); }
      P.mapped[r] = ptr; P.ipc_opened[r] = true;
    }
  }
  // every rank pushes its covariance segments into every peer's exchange area and its LM sums into every peer's slots: a peer that exported a
  // smaller mailbox (another max_points, another world) would be written out of bounds — refuse before the first frame
  for (int r = 0; r < world; r++) {
    if (r == rank) continue;
    unsigned long long hdr[2] = {0, 0};
    hipError_t e = hipMemcpy(hdr, static_cast<const unsigned long long*>(P.mapped[r]) + PEER_W_AREA_BYTES, sizeof(hdr), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { peer_disconnect_impl(c); return fail_hip(e, "reading a peer's mailbox header"); }
    if (hdr[0] != (unsigned long long)P.area_bytes || hdr[1] != (unsigned long long)world) {
      peer_disconnect_impl(c);
      g_err = "rolo_peer_connect: rank " + std::to_string(r) + " exported a mailbox for another max_points / world (every rank must call rolo_peer_export with the same arguments)";
      return ROLO_EINVAL;
    }
  }
  P.args = PeerArgs{};
  P.args.rank = rank; P.args.world = world;
  for (int r = 0; r < world; r++) P.args.box[r] = static_cast<unsigned long long*>(P.mapped[r]);
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz
  const char* tm = getenv("ROLO_PEER_TIMEOUT_MS");
  const double ms = tm ? atof(tm) : 10000.0;
  P.args.timeout_ticks = (unsigned long long)(std::max(ms, 1.0) * (double)khz);
  P.connected = true;
  c->rank = rank; c->world = world;
  c->have_corr = false; c->src.have_cov = false; c->tgt.have_cov = false; c->have_map = false;
  return ROLO_OK;
}

// Collective self-test of a connected group, meant to run before the first frame (bench.py's sharded leg, a deployment's start-up): the two
// exchanges of the sharded path with KNOWN words — `reps` all-reduces of 32 fp64 through the LM mailboxes (peer_allreduce_kernel: the block the
// controller runs per trial) and one covariance-segment push into every peer's exchange area — verified on every rank. The first time the
// ranks' mailboxes are written across devices (hipIpc mapping, peer access, fine-grained memory over xGMI) fails HERE, with a named error,
// instead of as a wrong pose or a time-out inside a frame. Every rank must call it with the same reps (it advances both exchange epochs).
int rolo_peer_selftest(rolo_ctx* c, int reps, double* us2) {
  if (!c || reps < 1 || reps > 1000) return ROLO_EINVAL;
  if (!peers(c)) { g_err = "rolo_peer_selftest: context is not connected to peers"; return ROLO_ESTATE; }
  if (c->async_pending) { g_err = "a registration is in flight on this context"; return ROLO_ESTATE; }
  int rc = set_device(c); if (rc) return rc;
  const int W = c->peer.args.world, rank = c->peer.args.rank;
  hipEvent_t ea = nullptr, eb = nullptr;
  HIPCHK(hipEventCreate(&ea)); HIPCHK(hipEventCreate(&eb));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{ea, eb};
  // (1) the LM exchange: rank r contributes (r + 1)(i + 1) + rep in value i; every rank must read W (W + 1) / 2 (i + 1) + W rep
  double lm_us = 0.0; int timed = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < NV_MAX; i++) c->h_sums[i] = (double)(rank + 1) * (i + 1) + rep;
    HIPCHK(hipMemcpyAsync(c->sums, c->h_sums, sizeof(double) * NV_MAX, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(ea, c->stream));
    HIPCHK(launch_peer_allreduce(c->sums, c->peer.args, c->peer.h_err, c->stream));
    HIPCHK(hipEventRecord(eb, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_sums, c->sums, sizeof(double) * NV_MAX, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (*c->peer.h_err != 0) { g_err = "rolo_peer_selftest: LM exchange " + std::to_string(rep) + " timed out on rank " + std::to_string(rank) + " (a peer's words never arrived in this rank's mailbox)"; return ROLO_ECOMM; }
    for (int i = 0; i < NV_MAX; i++) {
      const double want = 0.5 * W * (W + 1) * (i + 1) + (double)W * rep;
      if (c->h_sums[i] != want) { g_err = "rolo_peer_selftest: LM exchange " + std::to_string(rep) + " on rank " + std::to_string(rank) + ": value " + std::to_string(i) + " = " + std::to_string(c->h_sums[i]) + ", expected " + std::to_string(want); return ROLO_ECOMM; }
    }
    if (rep > 0 || reps == 1) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ea, eb); lm_us += 1e3 * ms; timed++; }   // the first one carries every rank's start-up skew
  }
  // (2) the covariance exchange: 4 workgroups' worth of words per rank
  const size_t seg = (size_t)6 * 256 * 4;
  if (seg * (size_t)W * sizeof(double) > c->peer.area_bytes) { g_err = "rolo_peer_selftest: exchange area smaller than the test segment"; return ROLO_EINVAL; }
  unsigned* bad = reinterpret_cast<unsigned*>(c->sums);   // NV_MAX doubles of scratch: PEER_MAX counters fit
  static_assert(PEER_MAX * sizeof(unsigned) <= NV_MAX * sizeof(double), "selftest counters");
  HIPCHK(hipMemsetAsync(bad, 0, PEER_MAX * sizeof(unsigned), c->stream));
  HIPCHK(launch_peer_selftest_fill(c->peer.args, c->peer.area_bytes, seg, c->stream));
  HIPCHK(hipEventRecord(ea, c->stream));
  HIPCHK(launch_peer_cov_exchange(c->peer.args, c->peer.area_bytes, seg, c->peer.h_err, c->stream));
  HIPCHK(hipEventRecord(eb, c->stream));
  HIPCHK(launch_peer_selftest_check(c->peer.args, c->peer.area_bytes, seg, bad, c->stream));
  unsigned h_bad[PEER_MAX] = {};
  HIPCHK(hipMemcpyAsync(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (*c->peer.h_err != 0) { g_err = "rolo_peer_selftest: covariance exchange timed out on rank " + std::to_string(rank) + " (a peer's flag never arrived)"; return ROLO_ECOMM; }
  for (int r = 0; r < W; r++)
    if (h_bad[r]) { g_err = "rolo_peer_selftest: rank " + std::to_string(rank) + " read " + std::to_string(h_bad[r]) + " wrong words in the segment rank " + std::to_string(r) + " pushed"; return ROLO_ECOMM; }
  float ms = 0.f; (void)hipEventElapsedTime(&ms, ea, eb);
  if (us2) { us2[0] = timed ? lm_us / timed : 0.0; us2[1] = 1e3 * ms; }
  return ROLO_OK;
}

int rolo_peer_disconnect(rolo_ctx* c) {
  if (!c) return ROLO_EINVAL;
  int rc = set_device(c); if (rc) return rc;
  peer_release(c);
  return ROLO_OK;
}

int rolo_peer_info(rolo_ctx* c, int* rank, int* world, char* mem_kind16) {
  if (!c) return ROLO_EINVAL;
  if (rank) *rank = c->peer.connected ? c->peer.args.rank : 0;
  if (world) *world = c->peer.connected ? c->peer.args.world : 0;
  if (mem_kind16) { strncpy(mem_kind16, c->peer.base ? c->peer.mem_kind : "", 15); mem_kind16[15] = 0; }
  return ROLO_OK;
}

}  // extern "C"

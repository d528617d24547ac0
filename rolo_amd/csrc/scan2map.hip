// SURVEY §8f.4 — the back end's scan-to-submap optimisation on gfx950.
// Replaces (reference src/backMapping.cpp) scan2MapOptimization :681-711, cornerOptimization :720-824, surfOptimization :827-901,
// combineOptimizationCoeffs :904-925 and LMOptimization :929-1058: every down-sampled corner / surface point of the current scan is moved
// by the current pose estimate (pointAssociateToMap :293-299), its 5 nearest neighbours in the corner / surface sub-map are fitted with a
// line (covariance + eigen-decomposition) / a plane (least squares), and the point-to-line / point-to-plane residuals drive a 6-dof
// Gauss-Newton step; up to 30 iterations.
//
// MI355X design: the two sub-maps get the Hilbert-sorted implicit BVH of the neighbour search (knn_cov.hip) once per call; ONE kernel per
// iteration does association + fit + Jacobian row per point (a per-lane exact 5-NN walk: the queries are foreign to the tree, so there is
// no packet to share) and reduces J^T J (21 values), J^T r (6) and the number of selected points per workgroup; a second tiny kernel sums the
// rows in a fixed order. The 6 x 6 solve, the degeneracy projection and the convergence test run on the host in float, as the reference's
// cv::solve / cv::eigen do (one 232-byte read-back per iteration; the back end runs at <= 1 / 0.15 s).
// Third-party numerics restated, not copied (OpenCV / Eigen are not in the reference tree), in FLOAT with the operation order of their published
// algorithms: cv::eigen of a symmetric float matrix = OpenCV's largest-pivot Jacobi scheme, eigenvalues descending, eigenvectors as rows;
// colPivHouseholderQr().solve of the 5 x 3 plane system = Eigen's column-pivoted Householder QR. The C++ oracle (oracle/rolo_oracle_backend.cpp)
// restates the same algorithms independently: selection flags bit-identical, coefficients to float rounding (tests/test_gpu_backend.py).
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include "knn_packet.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace rolo {
int ctx_build_map_trees(rolo_ctx* c, const float* corner, int nc, const float* surf, int ns, int stride, KnnPair* out);
extern "C" int rolo_ctx_acquire(int device, rolo_ctx** out);
extern "C" void rolo_ctx_release(rolo_ctx* c);
hipStream_t ctx_stream(rolo_ctx* c);
int ctx_device(rolo_ctx* c);
void ctx_set_error(const char* msg);

namespace {

constexpr int S2M_THREADS = 128;
constexpr int S2M_STACK = 40;
constexpr int S2M_NV = 28;   // 21 (lower triangle of A^T A) + 6 (A^T b) + 1 (selected points)

struct S2mArgs {
  const float4* feat;      // n_corner corner points, then n_surf surface points (x, y, z, intensity)
  int n_corner, n_surf;
  KnnCloud qry[2];         // the same features in CURVE order (sorted: x, y, z, bits(index in its cloud); padded to whole leaves): 64 consecutive ones form a packet
  KnnCloud map[2];         // corner / surface sub-map trees
  float T[12];             // transPointAssociateToMap rows (float Affine3f of pcl::getTransformation)
  float srx, crx, sry, cry, srz, crz;   // LMOptimization :942-947
  double* partials;        // grid x S2M_NV
  int xcd_remap;
  int cap;                 // 1: the search is capped at the radius the fits accept (d2 < 1.0)
  int4* wstats;            // ROLO_S2M_STATS: per wavefront (nodes, leaves, clock ticks of the walk, block) of the LAST iteration
  int qpp;                 // features per packet (wavefront) of s2m_packet_kernel: 64, or fewer (the remaining lanes idle) for more wavefronts in flight
  unsigned char* selected; // per feature point: 1 = laserCloudOri*Flag (debug / tests)
  float4* coeff;           // per feature point: coeffSel (debug / tests)
};

ROLO_DEV float box_d2f(const float4& lo, const float4& hi, float qx, float qy, float qz) {
  const float dx = fmaxf(fmaxf(__fsub_rn(lo.x, qx), __fsub_rn(qx, hi.x)), 0.f);
  const float dy = fmaxf(fmaxf(__fsub_rn(lo.y, qy), __fsub_rn(qy, hi.y)), 0.f);
  const float dz = fmaxf(fmaxf(__fsub_rn(lo.z, qz), __fsub_rn(qz, hi.z)), 0.f);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// exact 5 nearest neighbours of a foreign query in a Hilbert-sorted implicit BVH: (d2, index) order like the oracle's search, float
// distances ((dx*dx)+(dy*dy))+(dz*dz) without contraction (the FLANN L2 functor of pcl::KdTreeFLANN::nearestKSearch)
ROLO_DEV void knn5(const KnnCloud& M, float qx, float qy, float qz, int* stk /* LDS, stride S2M_THREADS */, float (&bd)[5], int (&bi)[5]) {
#pragma unroll
  for (int u = 0; u < 5; u++) { bd[u] = INFINITY; bi[u] = INT_MAX; }
  const float4* __restrict__ sorted = M.sorted;
  const float4* __restrict__ boxes = M.boxes;
  const int P = M.P, n_leaves = M.n_leaves;
  int sp = 0, h = 1;
  while (true) {
    if (h < P) {
      const float4 llo = boxes[4 * (size_t)h], lhi = boxes[4 * (size_t)h + 1], rlo = boxes[4 * (size_t)h + 2], rhi = boxes[4 * (size_t)h + 3];
      const float bl = box_d2f(llo, lhi, qx, qy, qz), br = box_d2f(rlo, rhi, qx, qy, qz);
      const bool okl = bl <= bd[4] && bl < INFINITY, okr = br <= bd[4] && br < INFINITY;
      if (okl && okr) {
        const bool lf = bl <= br;
        if (sp < S2M_STACK) { stk[sp * S2M_THREADS] = lf ? 2 * h + 1 : 2 * h; sp++; }
        h = lf ? 2 * h : 2 * h + 1;
        continue;
      }
      if (okl) { h = 2 * h; continue; }
      if (okr) { h = 2 * h + 1; continue; }
    } else if (h - P < n_leaves) {
      const int g = h - P;
      for (int u = 0; u < KNN_LEAF; u++) {
        const float4 c = sorted[(size_t)KNN_LEAF * g + u];
        const int ci = __float_as_int(c.w);
        if (ci == INT_MAX) continue;
        const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
        const float cd = ((dx * dx) + (dy * dy)) + (dz * dz);
        if (cd < bd[4] || (cd == bd[4] && ci < bi[4])) {
          // sorted insert into the 5 slots
          float d = cd; int i = ci;
#pragma unroll
          for (int s = 0; s < 5; s++) {
            const bool less = d < bd[s] || (d == bd[s] && i < bi[s]);
            const float td = less ? bd[s] : d; const int ti = less ? bi[s] : i;
            bd[s] = less ? d : bd[s]; bi[s] = less ? i : bi[s];
            d = td; i = ti;
          }
        }
      }
    }
    if (sp == 0) break;
    sp--;
    // re-test the popped node's parent bound is implicit: children are tested when expanded
    h = stk[sp * S2M_THREADS];
  }
}

// OpenCV's own hypot (scaled; only + * / sqrt: the same bits on host and device)
__host__ __device__ inline float cv_hypotf(float a, float b) {
  a = fabsf(a); b = fabsf(b);
  if (a > b) { b /= a; return a * sqrtf(1 + b * b); }
  if (b > 0) { a /= b; return b * sqrtf(1 + a * a); }
  return 0;
}

// cv::eigen of a symmetric N x N float matrix (backMapping.cpp:771 N = 3 on the device, :1009 N = 6 on the host), restated from OpenCV's published Jacobi
// scheme (not in the reference tree): only the upper triangle is kept; each step annihilates the off-diagonal element of largest magnitude (per-row and
// per-column maxima in index arrays, strict comparisons: the first of equals wins), rotation from y = (w_l - w_k) / 2, t = |y| + hypot(p, y),
// s = hypot(p, t), c = t / s, s = p / s, t = (p / t) p; at most 30 N^2 steps, stop when |p| <= FLT_EPSILON; eigenvalues descending, eigenvectors as rows.
// The float operation order is the parity target (oracle/rolo_oracle_backend.cpp restates the same scheme independently): "selection flags bit-identical" in the tests
// and DESIGN.md means bit-identical TO THAT ORACLE. Assumption about the reference's build: cv::eigen runs OpenCV's own Jacobi (modules/core/src/lapack.cpp, JacobiImpl_),
// which is what a stock OpenCV 4.x does; an OpenCV configured WITH_EIGEN=ON / HAVE_EIGEN routes cv::eigen through Eigen::SelfAdjointEigenSolver instead — eigenvalues
// and vectors then differ in their last ulps and the borderline "D[0] > 3 * D[1]" / "s > 0.1" decisions can flip for a few features per scan (advisor, round 4). A dump
// of cornerOptimization's flags from a real ROLO build (tools/README.md) is what would settle which variant a given installation runs.
template <int N>
__host__ __device__ inline void cv_eigen_sym(float* A /* N x N, destroyed */, float* W, float* V) {
  int indR[N], indC[N];
  for (int i = 0; i < N; i++) { for (int j = 0; j < N; j++) V[i * N + j] = 0.f; V[i * N + i] = 1.f; }
  auto row_max = [&](int k) { int m = k + 1; float mv = fabsf(A[k * N + m]); for (int i = k + 2; i < N; i++) { const float v = fabsf(A[k * N + i]); if (mv < v) { mv = v; m = i; } } return m; };
  auto col_max = [&](int k) { int m = 0; float mv = fabsf(A[k]); for (int i = 1; i < k; i++) { const float v = fabsf(A[i * N + k]); if (mv < v) { mv = v; m = i; } } return m; };
  for (int k = 0; k < N; k++) {
    W[k] = A[k * N + k];
    if (k < N - 1) indR[k] = row_max(k);
    if (k > 0) indC[k] = col_max(k);
  }
  for (int it = 0; it < N * N * 30; it++) {
    int k = 0; float mv = fabsf(A[indR[0]]);
    for (int i = 1; i < N - 1; i++) { const float v = fabsf(A[i * N + indR[i]]); if (mv < v) { mv = v; k = i; } }
    int l = indR[k];
    for (int i = 1; i < N; i++) { const float v = fabsf(A[indC[i] * N + i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
    const float p = A[k * N + l];
    if (fabsf(p) <= FLT_EPSILON) break;
    const float y = (W[l] - W[k]) * 0.5f;
    float t = fabsf(y) + cv_hypotf(p, y);
    float s = cv_hypotf(p, t);
    const float c = t / s;
    s = p / s; t = (p / t) * p;
    if (y < 0) { s = -s; t = -t; }
    A[k * N + l] = 0;
    W[k] -= t; W[l] += t;
    for (int i = 0; i < k; i++) { const float a0 = A[i * N + k], b0 = A[i * N + l]; A[i * N + k] = a0 * c - b0 * s; A[i * N + l] = a0 * s + b0 * c; }
    for (int i = k + 1; i < l; i++) { const float a0 = A[k * N + i], b0 = A[i * N + l]; A[k * N + i] = a0 * c - b0 * s; A[i * N + l] = a0 * s + b0 * c; }
    for (int i = l + 1; i < N; i++) { const float a0 = A[k * N + i], b0 = A[l * N + i]; A[k * N + i] = a0 * c - b0 * s; A[l * N + i] = a0 * s + b0 * c; }
    for (int i = 0; i < N; i++) { const float a0 = V[k * N + i], b0 = V[l * N + i]; V[k * N + i] = a0 * c - b0 * s; V[l * N + i] = a0 * s + b0 * c; }
    for (int j = 0; j < 2; j++) {
      const int idx = j == 0 ? k : l;
      if (idx < N - 1) indR[idx] = row_max(idx);
      if (idx > 0) indC[idx] = col_max(idx);
    }
  }
  for (int k = 0; k < N - 1; k++) {
    int m = k;
    for (int i = k + 1; i < N; i++) if (W[m] < W[i]) m = i;
    if (k != m) { const float tw = W[m]; W[m] = W[k]; W[k] = tw; for (int i = 0; i < N; i++) { const float tv = V[m * N + i]; V[m * N + i] = V[k * N + i]; V[k * N + i] = tv; } }
  }
}

// matX0 = matA0.colPivHouseholderQr().solve(matB0) for the 5 x 3 plane system A x = -1 (:845-861), restated from Eigen's published algorithm in FLOAT as the
// reference runs it: column-pivoted Householder QR — the column of largest remaining norm first, norms down-dated as in LAPACK working note 176, rank decided
// against eps * (largest column norm) / rows —, Q^T applied to the right-hand side, back substitution on the non-zero pivots, permutation undone.
// Columns are swapped with static indices (three cases) so that everything stays in registers.
ROLO_DEV void plane_colpiv_qr(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float& xa, float& xb, float& xc) {
  float q0[5], q1[5], q2[5];
#pragma unroll
  for (int i = 0; i < 5; i++) { q0[i] = px[i]; q1[i] = py[i]; q2[i] = pz[i]; }
  auto sq = [](const float (&c)[5], int from) { float s = 0; for (int i = from; i < 5; i++) s += c[i] * c[i]; return s; };
  auto swapc = [](float (&a)[5], float (&b)[5]) {
#pragma unroll
    for (int i = 0; i < 5; i++) { const float t = a[i]; a[i] = b[i]; b[i] = t; } };
  float nU0 = sqrtf(sq(q0, 0)), nU1 = sqrtf(sq(q1, 0)), nU2 = sqrtf(sq(q2, 0)), nD0 = nU0, nD1 = nU1, nD2 = nU2;
  const float th0 = fmaxf(nU0, fmaxf(nU1, nU2)) * FLT_EPSILON / 5.0f, threshold_helper = th0 * th0, downdate = sqrtf(FLT_EPSILON);
  int nonzero = 3, t0, t1;
  float h0, h1, h2;
  auto householder = [](float (&c)[5], int k, float& tau) {   // makeHouseholderInPlace on c[k..4]: essential part left in c[k+1..4], beta in c[k]
    float tail = 0; for (int i = k + 1; i < 5; i++) tail += c[i] * c[i];
    const float c0 = c[k];
    if (tail <= FLT_MIN) { tau = 0; for (int i = k + 1; i < 5; i++) c[i] = 0; return; }
    float beta = sqrtf(c0 * c0 + tail);
    if (c0 >= 0) beta = -beta;
    for (int i = k + 1; i < 5; i++) c[i] = c[i] / (c0 - beta);
    tau = (beta - c0) / beta;
    c[k] = beta;
  };
  auto apply = [](const float (&v)[5], int k, float tau, float (&c)[5]) {   // applyHouseholderOnTheLeft on column c, rows k..4
    float tmp = 0; for (int i = k + 1; i < 5; i++) tmp += v[i] * c[i];
    tmp += c[k];
    c[k] -= tau * tmp;
    for (int i = k + 1; i < 5; i++) c[i] -= tau * v[i] * tmp;
  };
  auto downdate_norm = [&](const float (&c)[5], int k, float& nU, float& nD) {
    if (nU != 0.f) {
      float temp = fabsf(c[k]) / nU;
      temp = (1.f + temp) * (1.f - temp);
      temp = temp < 0.f ? 0.f : temp;
      const float q = nU / nD, temp2 = temp * (q * q);
      if (temp2 <= downdate) { nD = sqrtf(sq(c, k + 1)); nU = nD; }
      else nU *= sqrtf(temp);
    }
  };
  // k = 0
  { int big = 0; float bn = nU0; if (nU1 > bn) { bn = nU1; big = 1; } if (nU2 > bn) { bn = nU2; big = 2; }
    if (bn * bn < threshold_helper * 5.0f) nonzero = 0;
    t0 = big;
    if (big == 1) { swapc(q0, q1); float t = nU0; nU0 = nU1; nU1 = t; t = nD0; nD0 = nD1; nD1 = t; }
    else if (big == 2) { swapc(q0, q2); float t = nU0; nU0 = nU2; nU2 = t; t = nD0; nD0 = nD2; nD2 = t; }
    householder(q0, 0, h0);
    apply(q0, 0, h0, q1); apply(q0, 0, h0, q2);
    downdate_norm(q1, 0, nU1, nD1); downdate_norm(q2, 0, nU2, nD2); }
  // k = 1
  { int big = 1; float bn = nU1; if (nU2 > bn) { bn = nU2; big = 2; }
    if (nonzero == 3 && bn * bn < threshold_helper * 4.0f) nonzero = 1;
    t1 = big;
    if (big == 2) { swapc(q1, q2); float t = nU1; nU1 = nU2; nU2 = t; t = nD1; nD1 = nD2; nD2 = t; }
    householder(q1, 1, h1);
    apply(q1, 1, h1, q2);
    downdate_norm(q2, 1, nU2, nD2); }
  // k = 2
  { if (nonzero == 3 && nU2 * nU2 < threshold_helper * 3.0f) nonzero = 2;
    householder(q2, 2, h2); }
  // permutation: identity with the transpositions (0, t0), (1, t1) applied on the right, in that order
  int p0 = 0, p1 = 1, p2 = 2;
  if (t0 == 1) { const int t = p0; p0 = p1; p1 = t; } else if (t0 == 2) { const int t = p0; p0 = p2; p2 = t; }
  if (t1 == 2) { const int t = p1; p1 = p2; p2 = t; }
  float c[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
  if (nonzero > 0) apply(q0, 0, h0, c);
  if (nonzero > 1) apply(q1, 1, h1, c);
  if (nonzero > 2) apply(q2, 2, h2, c);
  // R = [q0[0] q1[0] q2[0]; 0 q1[1] q2[1]; 0 0 q2[2]]: back substitution on the leading nonzero x nonzero block
  if (nonzero > 2) { c[2] /= q2[2]; c[0] -= q2[0] * c[2]; c[1] -= q2[1] * c[2]; }
  if (nonzero > 1) { c[1] /= q1[1]; c[0] -= q1[0] * c[1]; }
  if (nonzero > 0) { c[0] /= q0[0]; }
  float x[3] = {0.f, 0.f, 0.f};
  if (nonzero > 0) { if (p0 == 0) x[0] = c[0]; else if (p0 == 1) x[1] = c[0]; else x[2] = c[0]; }
  if (nonzero > 1) { if (p1 == 0) x[0] = c[1]; else if (p1 == 1) x[1] = c[1]; else x[2] = c[1]; }
  if (nonzero > 2) { if (p2 == 0) x[0] = c[2]; else if (p2 == 1) x[1] = c[2]; else x[2] = c[2]; }
  xa = x[0]; xb = x[1]; xc = x[2];
}

// cornerOptimization :740-820 / surfOptimization :845-897 for one feature: (sx, sy, sz) = pointSel, po = pointOri, px / py / pz = its five nearest sub-map
// points in (d2, index) order. Float arithmetic in the reference's operation order (double where its literals make it double).
ROLO_DEV void s2m_fit(bool corner, float sx, float sy, float sz, const float4& po, const float (&px)[5], const float (&py)[5], const float (&pz)[5], bool& sel, float4& coeff) {
  if (corner) {   // cornerOptimization :740-820
    float cx = 0, cy = 0, cz = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { cx += px[j]; cy += py[j]; cz += pz[j]; }
    cx /= 5; cy /= 5; cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const float ax = px[j] - cx, ay = py[j] - cy, az = pz[j] - cz;
      a11 += ax * ax; a12 += ax * ay; a13 += ax * az; a22 += ay * ay; a23 += ay * az; a33 += az * az;
    }
    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
    float A1[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33}, D[3], V[9];
    cv_eigen_sym<3>(A1, D, V);
    if (D[0] > 3 * D[1]) {
      const float x0 = sx, y0 = sy, z0 = sz;
      // "float x1 = cx + 0.1 * matV1.at<float>(0, 0)" (:782-788): the literal is a double — the sum is formed in double and narrowed on assignment
      const float x1 = (float)((double)cx + 0.1 * (double)V[0]), y1 = (float)((double)cy + 0.1 * (double)V[1]), z1 = (float)((double)cz + 0.1 * (double)V[2]);
      const float x2 = (float)((double)cx - 0.1 * (double)V[0]), y2 = (float)((double)cy - 0.1 * (double)V[1]), z2 = (float)((double)cz - 0.1 * (double)V[2]);
      const float m1 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1), m2 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1), m3 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
      const float a012 = sqrtf(m1 * m1 + m2 * m2 + m3 * m3);
      const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
      const float la = ((y1 - y2) * m1 + (z1 - z2) * m2) / a012 / l12;
      const float lb = -((x1 - x2) * m1 - (z1 - z2) * m3) / a012 / l12;
      const float lc = -((x1 - x2) * m2 + (y1 - y2) * m3) / a012 / l12;
      const float ld2 = a012 / l12;
      const float s = (float)(1.0 - 0.9 * (double)fabsf(ld2));   // "float s = 1 - 0.9 * fabs(ld2)": double literals, narrowed on assignment
      coeff = make_float4(s * la, s * lb, s * lc, s * ld2);
      sel = (double)s > 0.1;                                       // "if (s > 0.1)": compared as doubles
    }
  } else {        // surfOptimization :845-897
    float pa, pb, pc, pd = 1.f;
    plane_colpiv_qr(px, py, pz, pa, pb, pc);
    {
      const float ps = sqrtf(pa * pa + pb * pb + pc * pc);
      pa /= ps; pb /= ps; pc /= ps; pd /= ps;
      bool planeValid = true;
#pragma unroll
      for (int j = 0; j < 5; j++) if ((double)fabsf(pa * px[j] + pb * py[j] + pc * pz[j] + pd) > 0.2) planeValid = false;   // "> 0.2": compared as doubles (a NaN plane is valid here and dies at s > 0.1, as in the reference)
      if (planeValid) {
        const float pd2 = pa * sx + pb * sy + pc * sz + pd;
        // "float s = 1 - 0.9 * fabs(pd2) / sqrt(sqrt(...))": the square roots in float, the rest in double, narrowed on assignment
        const float s = (float)(1.0 - 0.9 * (double)fabsf(pd2) / (double)sqrtf(sqrtf(po.x * po.x + po.y * po.y + po.z * po.z)));
        coeff = make_float4(s * pa, s * pb, s * pc, s * pd2);
        sel = (double)s > 0.1;
      }
    }
  }
}

// one row of matA / matB (:960-989) of a selected feature, accumulated into the 21 + 6 + 1 sums
ROLO_DEV void s2m_row(const S2mArgs& A, const float4& po, const float4& coeff, double (&acc)[S2M_NV]) {
  const float ox = po.y, oy = po.z, oz = po.x;
  const float kx = coeff.y, ky = coeff.z, kz = coeff.x;
  const float srx = A.srx, crx = A.crx, sry = A.sry, cry = A.cry, srz = A.srz, crz = A.crz;
  const float arx = (crx * sry * srz * ox + crx * crz * sry * oy - srx * sry * oz) * kx + (-srx * srz * ox - crz * srx * oy - crx * oz) * ky +
                    (crx * cry * srz * ox + crx * cry * crz * oy - cry * srx * oz) * kz;
  const float ary = ((cry * srx * srz - crz * sry) * ox + (sry * srz + cry * crz * srx) * oy + crx * cry * oz) * kx +
                    ((-cry * crz - srx * sry * srz) * ox + (cry * srz - crz * srx * sry) * oy - crx * sry * oz) * kz;
  const float arz = ((crz * srx * sry - cry * srz) * ox + (-cry * crz - srx * sry * srz) * oy) * kx + (crx * crz * ox - crx * srz * oy) * ky +
                    ((sry * srz + cry * crz * srx) * ox + (crz * sry - cry * srx * srz) * oy) * kz;
  const double row[6] = {arz, arx, ary, kz, kx, ky};
  const double b = -(double)coeff.w;
  int t = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
#pragma unroll
    for (int c = 0; c <= r; c++) acc[t++] = row[r] * row[c];
  }
#pragma unroll
  for (int r = 0; r < 6; r++) acc[21 + r] = row[r] * b;
  acc[27] = 1.0;
}

__global__ __launch_bounds__(S2M_THREADS) void s2m_kernel(S2mArgs A) {
  __shared__ int stk[S2M_STACK * S2M_THREADS];
  __shared__ double red[S2M_THREADS / 64][S2M_NV];
  const int i = blockIdx.x * S2M_THREADS + threadIdx.x;
  const int n = A.n_corner + A.n_surf;
  double acc[S2M_NV];
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) acc[v] = 0.0;
  bool sel = false;
  float4 coeff = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    const float4 po = A.feat[i];
    const float* T = A.T;
    // pointAssociateToMap :293-299 (float, left to right)
    const float sx = T[0] * po.x + T[1] * po.y + T[2] * po.z + T[3];
    const float sy = T[4] * po.x + T[5] * po.y + T[6] * po.z + T[7];
    const float sz = T[8] * po.x + T[9] * po.y + T[10] * po.z + T[11];
    const bool corner = i < A.n_corner;
    const KnnCloud& M = A.map[corner ? 0 : 1];
    float bd[5]; int bi[5];
    knn5(M, sx, sy, sz, stk + threadIdx.x, bd, bi);
    if (bi[4] != INT_MAX && bd[4] < 1.0f) {
      float px[5], py[5], pz[5];
#pragma unroll
      for (int j = 0; j < 5; j++) { const float4 p = M.xyz[bi[j]]; px[j] = p.x; py[j] = p.y; pz[j] = p.z; }
      s2m_fit(corner, sx, sy, sz, po, px, py, pz, sel, coeff);
    }
    if (A.selected) A.selected[i] = sel ? 1 : 0;
    if (A.coeff) A.coeff[i] = sel ? coeff : make_float4(0.f, 0.f, 0.f, 0.f);
    if (sel) s2m_row(A, po, coeff, acc);
  }
  // workgroup sum, fixed order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) {
    double x = acc[v];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) red[wv][v] = x;
  }
  __syncthreads();
  if (threadIdx.x < S2M_NV) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < S2M_THREADS / 64; w++) s += red[w][threadIdx.x];
    A.partials[(size_t)blockIdx.x * S2M_NV + threadIdx.x] = s;
  }
}

// ---- the same with PACKETS (round 4) ------------------------------------------------------------------------------------------------------------
// The per-lane walk above takes ~1 ms for 48.7 k features against a 252 k-point sub-map: every lane descends the tree on its own, the wavefront executes
// the union of 64 divergent walks. But the features are a point cloud too: sorted along the same Hilbert curve (a rigid motion keeps neighbours
// neighbours, so they are sorted once per call, before the first pose), 64 consecutive ones are one blob — one wavefront walks the sub-map's tree ONCE
// for its 64 queries with the packet walk of the K5 search (knn_packet.hpp: ballot-driven descent, leaves fetched once per wave, packed-key min / max
// insert, k = 5). Same neighbours in the same (d2, index) order, so everything downstream — fits, flags, rows — is the same floats.
__global__ __launch_bounds__(256) void s2m_packet_kernel(S2mArgs A, int split /* first block of the surface features */) {
  __shared__ int stk[4][WALK_STACK];
  __shared__ double red[4][S2M_NV];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool corner = (int)blockIdx.x < split;   // workgroup-uniform
  const KnnCloud& Q = A.qry[corner ? 0 : 1];
  const KnnCloud& M = A.map[corner ? 0 : 1];
  const int j = lane < A.qpp ? (((int)blockIdx.x - (corner ? 0 : split)) * 4 + wv) * A.qpp + lane : INT_MAX;
  double acc[S2M_NV];
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) acc[v] = 0.0;
  float4 qs = make_float4(0.f, 0.f, 0.f, __int_as_float(INT_MAX));
  if (j < Q.n_sorted) qs = Q.sorted[j];
  const int qidx = __float_as_int(qs.w);
  const bool active = qidx != INT_MAX;
  const float* T = A.T;
  // pointAssociateToMap :293-299 (float, left to right)
  const float sx = T[0] * qs.x + T[1] * qs.y + T[2] * qs.z + T[3];
  const float sy = T[4] * qs.x + T[5] * qs.y + T[6] * qs.z + T[7];
  const float sz = T[8] * qs.x + T[9] * qs.y + T[10] * qs.z + T[11];
  const float4 q = make_float4(active ? sx : 0.f, active ? sy : 0.f, active ? sz : 0.f, 0.f);
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[5] = {sentinel, sentinel, sentinel, sentinel, sentinel};
  float bd = active ? INFINITY : -1.0f;
  double bkey = active ? sentinel : key_pack(0.f, 0);
  {
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
    int sp = 0, ns = 0, np = 0;
    const CoopPub none{};
    packet_walk<5, false, false, false, false>(M.sorted, M.boxes, M.P, 0, 0, q, K, 5, bkey, bd, 0.0, 0.0, (lds_int*)&stk[wv][0], sp, 1, none, ns, np, s0, s1, s2, s3, s4, s5);
  }
  bool sel = false;
  float4 coeff = make_float4(0.f, 0.f, 0.f, 0.f);
  const int gi = (corner ? 0 : A.n_corner) + qidx;   // position in the caller's order
  if (active) {
    float bdv[5]; int bi[5];
#pragma unroll
    for (int u = 0; u < 5; u++) { bdv[u] = key_d2(K[u]); bi[u] = key_idx(K[u]); }
    const float4 po = qs;
    if (bi[4] != INT_MAX && bdv[4] < 1.0f) {
      float px[5], py[5], pz[5];
#pragma unroll
      for (int u = 0; u < 5; u++) { const float4 p = M.xyz[bi[u]]; px[u] = p.x; py[u] = p.y; pz[u] = p.z; }
      s2m_fit(corner, sx, sy, sz, po, px, py, pz, sel, coeff);
    }
    if (A.selected) A.selected[gi] = sel ? 1 : 0;
    if (A.coeff) A.coeff[gi] = sel ? coeff : make_float4(0.f, 0.f, 0.f, 0.f);
    if (sel) s2m_row(A, po, coeff, acc);
  }
  // workgroup sum, fixed order
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) {
    double x = acc[v];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) red[wv][v] = x;
  }
  __syncthreads();
  if (threadIdx.x < S2M_NV) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) s += red[w][threadIdx.x];
    A.partials[(size_t)blockIdx.x * S2M_NV + threadIdx.x] = s;
  }
}

// ---- the same with SUB lanes per feature (round 4, second step) -----------------------------------------------------------------------------------------
// Measured on the packet kernel above (profiles/r04/s2m_kernel_stats.txt): 0.30 ms per iteration for 48.7 k features — 764 wavefronts, not even one
// per SIMD, each scoring all 16 points of ~160 leaves in every lane: the scan's features are thinned to 0.4 m (64 of them along the curve span metres)
// while the sub-map is five key frames dense, so a 64-feature packet's frontier is wide and every lane pays for the union. Here a wavefront carries
// 64 / SUB features, SUB adjacent lanes per feature: the frontier shrinks with the packet, there are SUB times as many wavefronts to hide the fetch
// latency, and a visited leaf costs each lane 16 / SUB candidates (point u of the leaf goes to sub-lane u % SUB: every sub-lane sees an even sample
// of the neighbourhood). Each sub-lane keeps the 5 best of ITS candidates; the feature's pruning bound is shared by its SUB lanes:
//     B = min( min_s K_s[4],  max_s K_s[ceil(5 / SUB) - 1] )          (SUB = 4: the second term is max( max_s K_s[0], min_s K_s[1] ))
// — all are keys with at least five real candidates at or below them (five in one sub-lane's list; ceil(5/SUB) in each of SUB lists; two in one list and
// the best of each other one), so a candidate or a box beyond B cannot belong to the five nearest: the walk stays exact. After the walk the SUB lists are merged (sorted inserts of the partners'
// keys): the same five (d2, index) keys in the same order as the one-lane search, and the fits, flags and rows downstream are the same floats.
ROLO_DEV void insert5(double (&K)[5], double ck) {
#pragma unroll
  for (int s = 4; s >= 1; s--) insert_slot(K[s], K[s - 1], ck);
  K[0] = vmin_f64(ck, K[0]);
}
template <int ST> ROLO_DEV void merge5(double (&K)[5]) {   // K <- the five smallest of K and the partner's K (both lanes end up with the same list)
  double o[5];
#pragma unroll
  for (int u = 0; u < 5; u++) o[u] = sub_xchg<ST>(K[u]);
#pragma unroll
  for (int u = 0; u < 5; u++) insert5(K, o[u]);
}

template <int SUB, int MODE /* 4: two tree levels per step (SUB = 4); + 2: two leaves per fetch */>
__global__ __launch_bounds__(256) void s2m_sub_kernel(S2mArgs A, int split /* first block of the surface features */) {
  constexpr int QPW = 64 / SUB, PPL = KNN_LEAF / SUB, SH = SUB == 2 ? 1 : SUB == 4 ? 2 : 3, NEED = (5 + SUB - 1) / SUB;
  static_assert(SUB == 2 || SUB == 4 || SUB == 8, "sub-lanes per feature");
  __shared__ int stk_[4][WALK_STACK];
  __shared__ double red[4][S2M_NV];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, sub = lane & (SUB - 1), ql = lane >> SH;
  const int blk = A.xcd_remap ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x, 4) : (int)blockIdx.x;   // curve-adjacent features on one XCD: they read the same boxes and leaves
  const bool corner = blk < split;   // workgroup-uniform
  const KnnCloud& Q = A.qry[corner ? 0 : 1];
  const KnnCloud& M = A.map[corner ? 0 : 1];
  const int j = ((blk - (corner ? 0 : split)) * 4 + wv) * QPW + ql;
  double acc[S2M_NV];
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) acc[v] = 0.0;
  float4 qs = make_float4(0.f, 0.f, 0.f, __int_as_float(INT_MAX));
  if (j < Q.n_sorted) qs = Q.sorted[j];
  const int qidx = __float_as_int(qs.w);
  const bool active = qidx != INT_MAX;
  const float* T = A.T;
  // pointAssociateToMap :293-299 (float, left to right)
  const float sx = T[0] * qs.x + T[1] * qs.y + T[2] * qs.z + T[3];
  const float sy = T[4] * qs.x + T[5] * qs.y + T[6] * qs.z + T[7];
  const float sz = T[8] * qs.x + T[9] * qs.y + T[10] * qs.z + T[11];
  const float4 q = make_float4(active ? sx : 0.f, active ? sy : 0.f, active ? sz : 0.f, 0.f);
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[5] = {sentinel, sentinel, sentinel, sentinel, sentinel};
  // The fits use a feature's neighbours only "if (pointSearchSqDis[4] < 1.0)" (:745, :852): a neighbour at d2 >= 1 can never matter — either the fifth
  // nearest is closer than 1 m, and so are the other four, or the feature is dropped. The search therefore starts with the bound (1.0f, index 0): every
  // key it accepts is below it, a feature without five neighbours inside the ball keeps sentinels in its list and is dropped by the same test as before.
  // Without the cap a feature far from the sub-map walks until it has ANY five points: those were the launch's heaviest wavefronts (A.cap = 0: the A/B).
  double B = active ? (A.cap ? key_pack(1.0f, 0) : sentinel) : key_pack(0.f, 0);   // (no key is below (0, 0): an idle feature accepts nothing)
  float bd = active ? (A.cap ? 1.0f : INFINITY) : -1.0f;
  int n_nodes = 0, n_leaves = 0;
  const long long t_walk0 = A.wstats ? wall_clock64() : 0;
  {
    const float4* __restrict__ sorted = M.sorted;
    const float4* __restrict__ boxes = M.boxes;
    const int P = M.P;
    lds_int* stk = (lds_int*)&stk_[wv][0];
    int sp = 0, h = 1;
    // Per-wavefront counters (ROLO_S2M_STATS, profiles/r04/s2m_wave_stats.txt): 123 nodes + 40 leaves per wavefront on average, 353 + 107 in the worst one,
    // ~0.4 us per step whichever it is — the launch (every wavefront resident at once) lasts as long as its heaviest wavefront, and a step costs what it
    // ISSUES. So (MODE 4) where the tree allows it a step takes TWO levels: the four grandchild boxes of h (nodes 4h .. 4h + 3, 128 contiguous bytes) are
    // tested one per sub-lane, the grandchild most features are nearest to is entered, the other live ones are pushed, best on top — one box test per
    // lane where two binary steps cost four (nodes per wavefront 123 -> 63, heaviest wavefront 167 -> 136 us). (+2) fetches a leaf together with the
    // entry on top of the stack when that is a leaf too: no further gain, kept as the A/B.
    auto score = [&](const float4 (&c)[PPL]) {
      bool changed = false;
#pragma unroll
      for (int t = 0; t < PPL; t++) {
        const float dx = q.x - c[t].x, dy = q.y - c[t].y, dz = q.z - c[t].z;
        const float cd = ((dx * dx) + (dy * dy)) + (dz * dz);   // (-ffp-contract=off)
        const double ck = key_pack(cd, __float_as_int(c[t].w));
        if (ck < B) { insert5(K, ck); changed = true; }
      }
      if (__any(changed)) {
        if (SUB == 4) {
          // exactly five candidates instead of eight: two from the sub-lane whose second key is smallest, the best key of each of the other three
          // (max_s K_s[0] also covers that sub-lane's own first key) — never above max_s K_s[1], the bound of two from each
          B = vmin_f64(vmin_f64(B, sub_min<SUB>(K[4])), vmax_f64(sub_max<SUB>(K[0]), sub_min<SUB>(K[1])));
        } else {
          B = vmin_f64(vmin_f64(B, sub_min<SUB>(K[4])), sub_max<SUB>(K[NEED - 1]));
        }
        bd = key_d2(B);
      }
    };
    while (true) {
      h = __builtin_amdgcn_readfirstlane(h);
      if ((MODE & 4) && SUB == 4 && 2 * h < P) {
        // two levels per step with the four box tests SPREAD over the feature's four lanes: sub-lane c tests grandchild c (the walk is bound by the
        // instructions it issues, not by its fetches: this step costs one box test per lane where two binary steps cost four)
        n_nodes++;
        const float4 blo = boxes[8 * (size_t)h + 2 * sub], bhi = boxes[8 * (size_t)h + 2 * sub + 1];
        const float d = box_d2(blo, bhi, q);
        const bool ok = (d <= bd) && (d < INFINITY);
        const unsigned long long m = __ballot(ok);
        float dm = ok ? d : INFINITY;   // the feature's nearest live grandchild votes (ties: every lane at the minimum)
        dm = fminf(dm, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dm), 0xB1, 0xF, 0xF, true)));
        dm = fminf(dm, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dm), 0x4E, 0xF, 0xF, true)));
        const unsigned long long v = __ballot(ok && d == dm);
        int key[4];   // wave-uniform: votes * 4 + (3 - c) for a live grandchild, -1 for one no lane reaches
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const unsigned long long sel = 0x1111111111111111ull << c;
          key[c] = (m & sel) != 0ull ? __popcll(v & sel) * 4 + (3 - c) : -1;
        }
        auto cx = [](int& a, int& bb) { const int hi = max(a, bb), lo = min(a, bb); a = hi; bb = lo; };
        cx(key[0], key[1]); cx(key[2], key[3]); cx(key[0], key[2]); cx(key[1], key[3]); cx(key[1], key[2]);   // descending
        if (key[0] >= 0) {
#pragma unroll
          for (int r = 3; r >= 1; r--) if (key[r] >= 0 && sp < WALK_STACK) { stk[sp] = 4 * h + (3 - (key[r] & 3)); sp++; }   // worst first: the second best is popped first
          h = 4 * h + (3 - (key[0] & 3));
          continue;
        }
      } else if (h < P) {
        n_nodes++;
        float4 llo, lhi, rlo, rhi;
        sload_node<false>(boxes + 4 * (size_t)h, llo, lhi, rlo, rhi);
        const float bl = box_d2(llo, lhi, q), br = box_d2(rlo, rhi, q);
        const bool okl = (bl <= bd) && (bl < INFINITY), okr = (br <= bd) && (br < INFINITY);
        const unsigned long long ml = __ballot(okl), mr = __ballot(okr);
        if (ml != 0ull && mr != 0ull) {
          const unsigned long long pref = __ballot((okl || okr) && (bl <= br));
          const bool left_first = 2 * __popcll(pref) >= __popcll(ml | mr);
          if (sp < WALK_STACK) { stk[sp] = left_first ? 2 * h + 1 : 2 * h; sp++; }
          h = left_first ? 2 * h : 2 * h + 1;
          continue;
        }
        if (ml != 0ull) { h = 2 * h; continue; }
        if (mr != 0ull) { h = 2 * h + 1; continue; }
      } else {
        const float4* __restrict__ leaf = sorted + KNN_LEAF * (size_t)(h - P);
        int h2 = -1;
        if ((MODE & 2) && sp > 0) { h2 = __builtin_amdgcn_readfirstlane(stk[sp - 1]); if (h2 < P) h2 = -1; }
        float4 c[PPL];
#pragma unroll
        for (int t = 0; t < PPL; t++) c[t] = leaf[t * SUB + sub];
        if ((MODE & 2) && h2 >= 0) {
          const float4* __restrict__ leaf2 = sorted + KNN_LEAF * (size_t)(h2 - P);
          float4 c2[PPL];
#pragma unroll
          for (int t = 0; t < PPL; t++) c2[t] = leaf2[t * SUB + sub];
          score(c);
          score(c2);
          sp--;
          n_leaves += 2;
        } else {
          score(c);
          n_leaves++;
        }
      }
      if (sp == 0) break;
      sp--;
      h = stk[sp];
    }
  }
  if (A.wstats && lane == 0) A.wstats[blk * 4 + wv] = make_int4(n_nodes, n_leaves, (int)(wall_clock64() - t_walk0), blk);
  merge5<0>(K);
  if (SUB >= 4) merge5<1>(K);
  if (SUB >= 8) merge5<2>(K);
  bool sel = false;
  float4 coeff = make_float4(0.f, 0.f, 0.f, 0.f);
  const int gi = (corner ? 0 : A.n_corner) + qidx;   // position in the caller's order
  if (active && sub == 0) {
    float bdv[5]; int bi[5];
#pragma unroll
    for (int u = 0; u < 5; u++) { bdv[u] = key_d2(K[u]); bi[u] = key_idx(K[u]); }
    const float4 po = qs;
    if (bi[4] != INT_MAX && bdv[4] < 1.0f) {
      float px[5], py[5], pz[5];
#pragma unroll
      for (int u = 0; u < 5; u++) { const float4 p = M.xyz[bi[u]]; px[u] = p.x; py[u] = p.y; pz[u] = p.z; }
      s2m_fit(corner, sx, sy, sz, po, px, py, pz, sel, coeff);
    }
    if (A.selected) A.selected[gi] = sel ? 1 : 0;
    if (A.coeff) A.coeff[gi] = sel ? coeff : make_float4(0.f, 0.f, 0.f, 0.f);
    if (sel) s2m_row(A, po, coeff, acc);
  }
  // workgroup sum, fixed order
#pragma unroll
  for (int v = 0; v < S2M_NV; v++) {
    double x = acc[v];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) red[wv][v] = x;
  }
  __syncthreads();
  if (threadIdx.x < S2M_NV) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) s += red[w][threadIdx.x];
    A.partials[(size_t)blk * S2M_NV + threadIdx.x] = s;
  }
}

// the rows of the workgroups summed in a fixed order (deterministic for a given grid): 8 strided groups of 32 lanes with eight loads in flight each, then
// the groups in order; the 28 sums go straight to pinned host memory (one wait per iteration, no copy launch)
__global__ __launch_bounds__(256) void s2m_sum_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  __shared__ double part[8][32];
  const int v = threadIdx.x & 31, q = threadIdx.x >> 5;
  double s = 0;
  if (v < S2M_NV) {
    for (int b0 = q; b0 < nblocks; b0 += 64) {
      double r[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int b = b0 + 8 * u; const double x = partials[(size_t)min(b, nblocks - 1) * S2M_NV + v]; r[u] = b < nblocks ? x : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++) s += r[u];
    }
  }
  part[q][v] = s;
  __syncthreads();
  if (threadIdx.x < S2M_NV) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) t += part[k][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

// ---- host side: float linear algebra of LMOptimization ------------------------------------------------------------------------
// cv::solve(AtA, AtB, X, DECOMP_QR) on a 6 x 6 float system: Householder QR in float
bool solve_qr6f(const float* Ain, const float* bin, float* x) {
  float A[36], b[6];
  std::memcpy(A, Ain, sizeof(A)); std::memcpy(b, bin, sizeof(b));
  for (int k = 0; k < 6; k++) {
    float nrm = 0; for (int i = k; i < 6; i++) nrm += A[i * 6 + k] * A[i * 6 + k];
    nrm = std::sqrt(nrm);
    if (nrm == 0.f) return false;
    const float alpha = A[k * 6 + k] > 0 ? -nrm : nrm;
    float v[6] = {0, 0, 0, 0, 0, 0};
    for (int i = k; i < 6; i++) v[i] = A[i * 6 + k];
    v[k] -= alpha;
    float vv = 0; for (int i = k; i < 6; i++) vv += v[i] * v[i];
    if (vv == 0.f) continue;
    for (int j = k; j < 6; j++) { float d = 0; for (int i = k; i < 6; i++) d += v[i] * A[i * 6 + j]; d = 2 * d / vv; for (int i = k; i < 6; i++) A[i * 6 + j] -= d * v[i]; }
    { float d = 0; for (int i = k; i < 6; i++) d += v[i] * b[i]; d = 2 * d / vv; for (int i = k; i < 6; i++) b[i] -= d * v[i]; }
  }
  for (int i = 5; i >= 0; i--) {
    float s = b[i];
    for (int j = i + 1; j < 6; j++) s -= A[i * 6 + j] * x[j];
    if (A[i * 6 + i] == 0.f) return false;
    x[i] = s / A[i * 6 + i];
  }
  return true;
}
bool invert6f(const float* Ain, float* inv) {   // matV.inv() (LU with partial pivoting)
  float a[36]; std::memcpy(a, Ain, sizeof(a));
  for (int i = 0; i < 36; i++) inv[i] = (i % 7 == 0) ? 1.f : 0.f;
  for (int c = 0; c < 6; c++) {
    int p = c; float best = std::fabs(a[c * 6 + c]);
    for (int r = c + 1; r < 6; r++) if (std::fabs(a[r * 6 + c]) > best) { best = std::fabs(a[r * 6 + c]); p = r; }
    if (best == 0.f) return false;
    if (p != c) for (int j = 0; j < 6; j++) { std::swap(a[p * 6 + j], a[c * 6 + j]); std::swap(inv[p * 6 + j], inv[c * 6 + j]); }
    const float d = 1.f / a[c * 6 + c];
    for (int j = 0; j < 6; j++) { a[c * 6 + j] *= d; inv[c * 6 + j] *= d; }
    for (int r = 0; r < 6; r++) if (r != c) { const float f = a[r * 6 + c]; if (f != 0.f) for (int j = 0; j < 6; j++) { a[r * 6 + j] -= f * a[c * 6 + j]; inv[r * 6 + j] -= f * inv[c * 6 + j]; } }
  }
  return true;
}

}  // namespace
}  // namespace rolo

using namespace rolo;

#define SCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { ctx_set_error((std::string(#x) + ": " + hipGetErrorString(_e)).c_str()); return ROLO_EHIP; } } while (0)

namespace rolo {
void** ctx_s2m_slot(rolo_ctx* c);   // api.hip
unsigned long long ctx_cloud_epoch(rolo_ctx* c);   // api.hip: changes whenever the context's source / target clouds change hands
struct S2mScratch { float4* feat = nullptr; double *part = nullptr, *sum = nullptr; unsigned char* sel = nullptr; float4* coeff = nullptr;
                    size_t feat_cap = 0, part_cap = 0, sum_cap = 0, sel_cap = 0, coeff_cap = 0;
                    KnnPair maps{}; int m_corner = 0, m_surf = 0; bool have_maps = false;   // the resident sub-map (rolo_scan2map_set_submap): trees of the context's two clouds
                    bool map_set = false; unsigned long long maps_epoch = 0;   // a sub-map was handed in (possibly too small to search); the context's cloud epoch when its trees were built —
                                                                               // any later upload into the context (rolo_set_input_*, a registration, the pool) makes the raw pointers in `maps` stale
                    rolo_ctx* qctx = nullptr; hipEvent_t qev = nullptr;
                    double* h_sum = nullptr; };   // pinned: the 28 sums of an iteration, written by s2m_sum_kernel itself   // helper context (from the pool): the scan's features sorted along the curve, its two clouds
}  // namespace rolo
extern "C" void rolo_s2m_destroy(rolo_ctx* c) {   // called by rolo_ctx_destroy
  S2mScratch* W = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
  if (!W) return;
  for (void* p : {(void*)W->feat, (void*)W->part, (void*)W->sum, (void*)W->sel, (void*)W->coeff}) if (p) (void)hipFree(p);
  if (W->qev) (void)hipEventDestroy(W->qev);
  if (W->h_sum) (void)hipHostFree(W->h_sum);
  rolo_ctx* q = W->qctx;
  delete W; *ctx_s2m_slot(c) = nullptr;
  if (q) rolo_ctx_release(q);
}
extern "C" void rolo_s2m_forget(rolo_ctx* c) {   // called by rolo_ctx_release before the context goes back to the pool
  S2mScratch* W = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
  if (!W) return;
  W->have_maps = false; W->map_set = false; W->m_corner = W->m_surf = 0; W->maps = KnnPair{};
  rolo_ctx* q = W->qctx; W->qctx = nullptr;
  if (q) rolo_ctx_release(q);   // (the scratch buffers stay: they only grow, and a hipFree would stall every context's frames in flight)
}

// kdtreeCornerFromMap->setInputCloud(laserCloudCornerFromMapDS) / kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS) (:690-691) as a call of its own: the
// sub-map is uploaded and its two search trees are built ONCE and stay in the context (its source / target clouds) until the next call — the surrounding
// key-frame set changes every few scans, not every scan (extractSurroundingKeyFrames), so rolo_scan2map_optimize(..., NULL, 0, NULL, 0, ...) then costs one
// upload of the scan's features and the Gauss-Newton iterations.
extern "C" int rolo_scan2map_set_submap(rolo_ctx* c, const float* map_corner, int m_corner, const float* map_surf, int m_surf) {
  if (!c || m_corner < 0 || m_surf < 0 || (m_corner && !map_corner) || (m_surf && !map_surf)) return ROLO_EINVAL;
  S2mScratch* W = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
  if (!W) { W = new S2mScratch(); *ctx_s2m_slot(c) = W; }
  W->have_maps = false; W->map_set = false; W->m_corner = 0; W->m_surf = 0;   // nothing is resident until the build below has succeeded
  if (m_corner < 5 || m_surf < 5) { W->map_set = true; W->m_corner = m_corner; W->m_surf = m_surf; return ROLO_OK; }   // nothing a 5-NN query could be answered from: rolo_scan2map_optimize reports skipped = 2
  KnnPair built{};
  const int rc = ctx_build_map_trees(c, map_corner, m_corner, map_surf, m_surf, 4, &built);
  if (rc) return rc;
  W->maps = built; W->m_corner = m_corner; W->m_surf = m_surf; W->maps_epoch = ctx_cloud_epoch(c);
  W->have_maps = true; W->map_set = true;
  return ROLO_OK;
}

extern "C" int rolo_scan2map_optimize(rolo_ctx* c, const float* corner, int n_corner, const float* surf, int n_surf, const float* map_corner, int m_corner,
                                      const float* map_surf, int m_surf, float* transformTobeMapped, int edge_min, int surf_min, rolo_scan2map_stats* stats,
                                      unsigned char* selected_out, float* coeff_out) {
  if (!c || !transformTobeMapped || n_corner < 0 || n_surf < 0 || m_corner < 0 || m_surf < 0 || (n_corner && !corner) || (n_surf && !surf) ||
      (m_corner && !map_corner) || (m_surf && !map_surf))
    return ROLO_EINVAL;
  rolo_scan2map_stats st{};
  if (stats) *stats = st;
  const bool resident = !map_corner && !map_surf;   // the sub-map of the last rolo_scan2map_set_submap
  if (resident) {
    S2mScratch* W0 = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
    if (!W0 || !W0->map_set) { ctx_set_error("rolo_scan2map_optimize without a sub-map: call rolo_scan2map_set_submap first"); return ROLO_ESTATE; }
    // the resident trees are raw pointers into the context's own source / target clouds: anything uploaded into the context since (rolo_set_input_source /
    // _target, a registration, a trip through the pool) has overwritten or reallocated them
    if (W0->have_maps && W0->maps_epoch != ctx_cloud_epoch(c)) {
      W0->have_maps = false; W0->map_set = false;
      ctx_set_error("rolo_scan2map_optimize: the resident sub-map was overwritten by a later upload into this context: call rolo_scan2map_set_submap again");
      return ROLO_ESTATE;
    }
    m_corner = W0->m_corner; m_surf = W0->m_surf;
  }
  // :689 — "if (laserCloudCornerLastDSNum > edgeFeatureMinValidNum && laserCloudSurfLastDSNum > surfFeatureMinValidNum)"
  if (!(n_corner > edge_min && n_surf > surf_min)) { st.skipped = 1; if (stats) *stats = st; return ROLO_OK; }
  // a sub-map without five points cannot answer a 5-NN query: the reference's association then selects nothing from it (and reads
  // pointSearchSqDis[4] of a shorter result, :745 / :852) — nothing to optimise against; report it as skipped instead of an error
  if (m_corner < 5 || m_surf < 5) { st.skipped = 2; if (stats) *stats = st; return ROLO_OK; }
  if (!resident) { const int rc = rolo_scan2map_set_submap(c, map_corner, m_corner, map_surf, m_surf); if (rc) return rc; }
  if (!static_cast<S2mScratch*>(*ctx_s2m_slot(c))->have_maps) { ctx_set_error("rolo_scan2map_optimize: no searchable sub-map is resident"); return ROLO_ESTATE; }
  const KnnPair maps = static_cast<S2mScratch*>(*ctx_s2m_slot(c))->maps;
  hipStream_t s = ctx_stream(c);
  const int n = n_corner + n_surf;
  // ROLO_S2M_PACKETS=0: one tree walk per lane in the caller's order (rounds 2-3, the A/B); default: the features sorted along the curve once per call,
  // 64 consecutive ones walk the sub-map's tree as one packet (s2m_packet_kernel)
  static const bool use_packets = [] { const char* e = getenv("ROLO_S2M_PACKETS"); return !(e && atoi(e) == 0); }();
  static const int qpp = [] { const char* e = getenv("ROLO_S2M_QPP"); const int v = e ? atoi(e) : 64; return v == 8 || v == 16 || v == 32 ? v : 64; }();
  // ROLO_S2M_SUB = 2 / 4 / 8 lanes per feature (s2m_sub_kernel); 1 = the 64-feature packets (s2m_packet_kernel, the A/B)
  static const int sub = [] { const char* e = getenv("ROLO_S2M_SUB"); const int v = e ? atoi(e) : 4; return v == 1 || v == 2 || v == 8 ? v : 4; }();
  static const int wide = [] { const char* e = getenv("ROLO_S2M_WIDE"); const int v = e ? atoi(e) : 4; return v == 0 || v == 6 ? v : 4; }();
  const int fpw = sub > 1 ? 64 / sub : qpp;   // features per wavefront
  KnnPair qp{};
  int grid = (n + S2M_THREADS - 1) / S2M_THREADS, split = 0;
  if (use_packets) {
    S2mScratch* Wq = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
    if (!Wq->qctx) { if (rolo_ctx_acquire(ctx_device(c), &Wq->qctx) != ROLO_OK) return ROLO_EHIP; SCHK(hipEventCreateWithFlags(&Wq->qev, hipEventDisableTiming)); }
    const int rq = ctx_build_map_trees(Wq->qctx, corner, n_corner, surf, n_surf, 4, &qp);   // upload + Hilbert sort (+ a tree nobody walks) on the helper's stream
    if (rq) return rq;
    SCHK(hipEventRecord(Wq->qev, ctx_stream(Wq->qctx)));
    SCHK(hipStreamWaitEvent(s, Wq->qev, 0));
    split = (qp.c[0].n_sorted + 4 * fpw - 1) / (4 * fpw);
    grid = split + (qp.c[1].n_sorted + 4 * fpw - 1) / (4 * fpw);
  }
  // scratch lives with the context and only grows: hipFree is a device-wide synchronisation that would stall the frames other contexts have in flight
  S2mScratch* W = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
  if (!W) { W = new S2mScratch(); *ctx_s2m_slot(c) = W; }
  auto grow = [&](void** p, size_t& cap, size_t bytes) -> bool {
    if (bytes <= cap && *p) return true;
    if (*p) { (void)hipFree(*p); *p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(p, want) != hipSuccess) return false;
    cap = want; return true;
  };
  if (!grow((void**)&W->feat, W->feat_cap, sizeof(float4) * (size_t)n) || !grow((void**)&W->part, W->part_cap, sizeof(double) * S2M_NV * (size_t)grid) ||
      !grow((void**)&W->sum, W->sum_cap, sizeof(double) * S2M_NV) || (selected_out && !grow((void**)&W->sel, W->sel_cap, (size_t)n)) ||
      (coeff_out && !grow((void**)&W->coeff, W->coeff_cap, sizeof(float4) * (size_t)n))) { ctx_set_error("hipMalloc failed (scan2map)"); return ROLO_EHIP; }
  float4* d_feat = W->feat; double* d_part = W->part; unsigned char* d_sel = selected_out ? W->sel : nullptr; float4* d_coeff = coeff_out ? W->coeff : nullptr;
  auto cleanup = [&]() {};
  if (!use_packets &&   // (the packet kernel reads the features from the helper context's sorted clouds)
      (hipMemcpyAsync(d_feat, corner, sizeof(float4) * (size_t)n_corner, hipMemcpyHostToDevice, s) != hipSuccess ||
       hipMemcpyAsync(d_feat + n_corner, surf, sizeof(float4) * (size_t)n_surf, hipMemcpyHostToDevice, s) != hipSuccess)) { cleanup(); ctx_set_error("upload failed (scan2map)"); return ROLO_EHIP; }
  S2mArgs A{};
  A.feat = d_feat; A.n_corner = n_corner; A.n_surf = n_surf; A.map[0] = maps.c[0]; A.map[1] = maps.c[1]; A.partials = d_part; A.selected = d_sel; A.coeff = d_coeff;
  A.qry[0] = qp.c[0]; A.qry[1] = qp.c[1]; A.qpp = qpp;
  { static const int xr = [] { const char* e = getenv("ROLO_S2M_XCD"); return e ? atoi(e) : 1; }(); A.xcd_remap = xr; }
  { static const int cp = [] { const char* e = getenv("ROLO_S2M_CAP"); return e ? atoi(e) : 1; }(); A.cap = cp; }
  static const char* stats_path = getenv("ROLO_S2M_STATS");
  int4* d_wstats = nullptr;
  if (stats_path && use_packets && sub > 1) { SCHK(hipMalloc((void**)&d_wstats, sizeof(int4) * 4 * (size_t)grid)); SCHK(hipMemsetAsync(d_wstats, 0, sizeof(int4) * 4 * (size_t)grid, s)); A.wstats = d_wstats; }
  float* tf = transformTobeMapped;
  bool isDegenerate = false;
  float matP[36]; for (int i = 0; i < 36; i++) matP[i] = (i % 7 == 0) ? 1.f : 0.f;
  if (!W->h_sum) SCHK(hipHostMalloc((void**)&W->h_sum, sizeof(double) * S2M_NV));
  double* h_sum = W->h_sum;
  for (int iterCount = 0; iterCount < 30; iterCount++) {
    // trans2Affine3f :339-342 = pcl::getTransformation(x, y, z, roll, pitch, yaw), float
    {
      const float Ax = std::cos(tf[2]), Bx = std::sin(tf[2]), Cx = std::cos(tf[1]), Dx = std::sin(tf[1]), Ex = std::cos(tf[0]), Fx = std::sin(tf[0]);
      const float DE = Dx * Ex, DF = Dx * Fx;
      A.T[0] = Ax * Cx; A.T[1] = Ax * DF - Bx * Ex; A.T[2] = Bx * Fx + Ax * DE; A.T[3] = tf[3];
      A.T[4] = Bx * Cx; A.T[5] = Ax * Ex + Bx * DF; A.T[6] = Bx * DE - Ax * Fx; A.T[7] = tf[4];
      A.T[8] = -Dx; A.T[9] = Cx * Fx; A.T[10] = Cx * Ex; A.T[11] = tf[5];
    }
    A.srx = std::sin(tf[1]); A.crx = std::cos(tf[1]); A.sry = std::sin(tf[2]); A.cry = std::cos(tf[2]); A.srz = std::sin(tf[0]); A.crz = std::cos(tf[0]);
    if (use_packets && sub == 4 && wide == 4) s2m_sub_kernel<4, 4><<<grid, 256, 0, s>>>(A, split);
    else if (use_packets && sub == 4 && wide == 6) s2m_sub_kernel<4, 6><<<grid, 256, 0, s>>>(A, split);
    else if (use_packets && sub == 4) s2m_sub_kernel<4, 0><<<grid, 256, 0, s>>>(A, split);
    else if (use_packets && sub == 2) s2m_sub_kernel<2, 0><<<grid, 256, 0, s>>>(A, split);
    else if (use_packets && sub == 8) s2m_sub_kernel<8, 0><<<grid, 256, 0, s>>>(A, split);
    else if (use_packets) s2m_packet_kernel<<<grid, 256, 0, s>>>(A, split);
    else s2m_kernel<<<grid, S2M_THREADS, 0, s>>>(A);
    SCHK(hipGetLastError());
    s2m_sum_kernel<<<1, 256, 0, s>>>(d_part, grid, h_sum);
    SCHK(hipGetLastError());
    if (hipStreamSynchronize(s) != hipSuccess) { cleanup(); ctx_set_error("scan2map iteration failed"); return ROLO_EHIP; }
    st.iterations = iterCount + 1;
    st.n_selected = (int)(h_sum[27] + 0.5);
    if (st.n_selected < 50) break;   // LMOptimization returns false without touching the pose: the remaining iterations would repeat this one
    float AtA[36], AtB[6], X[6];
    { int t = 0; for (int r = 0; r < 6; r++) for (int cc = 0; cc <= r; cc++) { AtA[r * 6 + cc] = AtA[cc * 6 + r] = (float)h_sum[t]; t++; } }
    for (int r = 0; r < 6; r++) AtB[r] = (float)h_sum[21 + r];
    if (!solve_qr6f(AtA, AtB, X)) { for (int r = 0; r < 6; r++) X[r] = 0.f; }
    if (iterCount == 0) {   // degeneracy of the first linearisation :1004-1026
      float Acp[36], E[6], V[36], V2[36], Vi[36];
      std::memcpy(Acp, AtA, sizeof(Acp));
      cv_eigen_sym<6>(Acp, E, V);
      std::memcpy(V2, V, sizeof(V2));
      isDegenerate = false;
      for (int i = 5; i >= 0; i--) {
        if (E[i] < 100.f) { for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0; isDegenerate = true; } else break;
      }
      if (invert6f(V, Vi)) { for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) { float a = 0; for (int k = 0; k < 6; k++) a += Vi[r * 6 + k] * V2[k * 6 + cc]; matP[r * 6 + cc] = a; } }
    }
    if (isDegenerate) { float X2[6]; std::memcpy(X2, X, sizeof(X2)); for (int r = 0; r < 6; r++) { float a = 0; for (int k = 0; k < 6; k++) a += matP[r * 6 + k] * X2[k]; X[r] = a; } }
    for (int r = 0; r < 6; r++) tf[r] += X[r];
    const float r2d = 57.29578f;   // pcl::rad2deg(float)
    // "float deltaR = sqrt(pow(pcl::rad2deg(x), 2) + ...)" (:1041-1048): rad2deg in float, pow(float, int) and the sum in double, narrowed on assignment
    const float deltaR = (float)std::sqrt(std::pow((double)(X[0] * r2d), 2) + std::pow((double)(X[1] * r2d), 2) + std::pow((double)(X[2] * r2d), 2));
    const float deltaT = (float)std::sqrt(std::pow((double)(X[3] * 100), 2) + std::pow((double)(X[4] * 100), 2) + std::pow((double)(X[5] * 100), 2));
    if (deltaR < 0.05f && deltaT < 0.05f) { st.converged = 1; break; }
  }
  if (d_wstats) {   // one line per wavefront: nodes, leaves, walk ticks (100 MHz), block
    std::vector<int4> hw(4 * (size_t)grid);
    SCHK(hipMemcpy(hw.data(), d_wstats, sizeof(int4) * hw.size(), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stats_path, "w")) { for (const int4& r : hw) fprintf(f, "%d %d %d %d\n", r.x, r.y, r.z, r.w); fclose(f); }
    (void)hipFree(d_wstats);
  }
  st.degenerate = isDegenerate ? 1 : 0;
  if (selected_out) SCHK(hipMemcpy(selected_out, d_sel, (size_t)n, hipMemcpyDeviceToHost));
  if (coeff_out) SCHK(hipMemcpy(coeff_out, d_coeff, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
  if (stats) *stats = st;
  return ROLO_OK;
}

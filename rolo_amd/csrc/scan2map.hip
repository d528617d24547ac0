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
// Third-party numerics restated, not copied (OpenCV / Eigen are not in the reference tree): cv::eigen of a symmetric float matrix = Jacobi
// rotations, eigenvalues descending, eigenvectors as rows; colPivHouseholderQr().solve of the 5 x 3 plane system = its least-squares solution.
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace rolo {
int ctx_build_map_trees(rolo_ctx* c, const float* corner, int nc, const float* surf, int ns, int stride, KnnPair* out);
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
  KnnCloud map[2];         // corner / surface sub-map trees
  float T[12];             // transPointAssociateToMap rows (float Affine3f of pcl::getTransformation)
  float srx, crx, sry, cry, srz, crz;   // LMOptimization :942-947
  double* partials;        // grid x S2M_NV
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

// eigen-decomposition of a symmetric 3 x 3 float matrix by cyclic Jacobi rotations: eigenvalues descending, eigenvectors as rows (cv::eigen)
ROLO_DEV void eigen_sym3f(float a11, float a12, float a13, float a22, float a23, float a33, float (&D)[3], float (&V)[9]) {
  float A[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33};
#pragma unroll
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.f : 0.f;
  for (int sweep = 0; sweep < 30; sweep++) {
    const float off = fabsf(A[1]) + fabsf(A[2]) + fabsf(A[5]);
    if (off < FLT_MIN * 16) break;
#pragma unroll
    for (int pq = 0; pq < 3; pq++) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const float apq = A[p * 3 + q];
      if (fabsf(apq) < FLT_MIN) continue;
      const float app = A[p * 4], aqq = A[q * 4];
      const float theta = (aqq - app) / (2.f * apq);
      const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
      const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
      for (int k = 0; k < 3; k++) {   // A <- A J (columns p, q)
        const float akp = A[k * 3 + p], akq = A[k * 3 + q];
        A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq;
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {   // A <- J^T A (rows p, q)
        const float apk = A[p * 3 + k], aqk = A[q * 3 + k];
        A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk;
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {   // V rows = eigenvectors: V <- J^T V
        const float vpk = V[p * 3 + k], vqk = V[q * 3 + k];
        V[p * 3 + k] = c * vpk - s * vqk; V[q * 3 + k] = s * vpk + c * vqk;
      }
    }
  }
  D[0] = A[0]; D[1] = A[4]; D[2] = A[8];
  // sort descending with the rows of V
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2 - i; j++)
      if (D[j] < D[j + 1]) {
        const float td = D[j]; D[j] = D[j + 1]; D[j + 1] = td;
#pragma unroll
        for (int k = 0; k < 3; k++) { const float tv = V[j * 3 + k]; V[j * 3 + k] = V[(j + 1) * 3 + k]; V[(j + 1) * 3 + k] = tv; }
      }
}

// least-squares solution of the 5 x 3 system A x = -1 (matA0.colPivHouseholderQr().solve(matB0), :845-861): normal equations in double
// (well conditioned: 5 points within 1 m of each other, coordinates O(100 m)), result narrowed to float
ROLO_DEV bool plane_lsq(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float& pa, float& pb, float& pc) {
  double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const double x = px[j], y = py[j], z = pz[j];
    sxx += x * x; sxy += x * y; sxz += x * z; syy += y * y; syz += y * z; szz += z * z; bx -= x; by -= y; bz -= z;
  }
  const Sym3 Minv = sym3_inverse(Sym3{sxx, sxy, sxz, syy, syz, szz});
  const Vec3 x = sym3_mulv(Minv, Vec3{bx, by, bz});
  pa = (float)x.x; pb = (float)x.y; pc = (float)x.z;
  return isfinite(pa) && isfinite(pb) && isfinite(pc);
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
        float D[3], V[9];
        eigen_sym3f(a11, a12, a13, a22, a23, a33, D, V);
        if (D[0] > 3 * D[1]) {
          const float x0 = sx, y0 = sy, z0 = sz;
          const float x1 = cx + 0.1f * V[0], y1 = cy + 0.1f * V[1], z1 = cz + 0.1f * V[2];
          const float x2 = cx - 0.1f * V[0], y2 = cy - 0.1f * V[1], z2 = cz - 0.1f * V[2];
          const float m1 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1), m2 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1), m3 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
          const float a012 = sqrtf(m1 * m1 + m2 * m2 + m3 * m3);
          const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
          const float la = ((y1 - y2) * m1 + (z1 - z2) * m2) / a012 / l12;
          const float lb = -((x1 - x2) * m1 - (z1 - z2) * m3) / a012 / l12;
          const float lc = -((x1 - x2) * m2 + (y1 - y2) * m3) / a012 / l12;
          const float ld2 = a012 / l12;
          const float s = 1 - 0.9f * fabsf(ld2);
          coeff = make_float4(s * la, s * lb, s * lc, s * ld2);
          sel = s > 0.1f;
        }
      } else {        // surfOptimization :845-897
        float pa, pb, pc, pd = 1.f;
        if (plane_lsq(px, py, pz, pa, pb, pc)) {
          const float ps = sqrtf(pa * pa + pb * pb + pc * pc);
          pa /= ps; pb /= ps; pc /= ps; pd /= ps;
          bool planeValid = true;
#pragma unroll
          for (int j = 0; j < 5; j++) if (fabsf(pa * px[j] + pb * py[j] + pc * pz[j] + pd) > 0.2f) planeValid = false;
          if (planeValid) {
            const float pd2 = pa * sx + pb * sy + pc * sz + pd;
            const float s = 1 - 0.9f * fabsf(pd2) / sqrtf(sqrtf(po.x * po.x + po.y * po.y + po.z * po.z));
            coeff = make_float4(s * pa, s * pb, s * pc, s * pd2);
            sel = s > 0.1f;
          }
        }
      }
    }
    if (A.selected) A.selected[i] = sel ? 1 : 0;
    if (A.coeff) A.coeff[i] = sel ? coeff : make_float4(0.f, 0.f, 0.f, 0.f);
    if (sel) {   // one row of matA / matB (:960-989): lidar -> camera axes
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

__global__ __launch_bounds__(64) void s2m_sum_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  const int v = threadIdx.x;
  if (v >= S2M_NV) return;
  double s = 0;
  for (int b = 0; b < nblocks; b++) s += partials[(size_t)b * S2M_NV + v];
  out[v] = s;
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
// cv::eigen of a symmetric 6 x 6 float matrix: Jacobi, eigenvalues descending, eigenvectors as rows
void eigen_sym6f(const float* Ain, float* E, float* V) {
  float A[36]; std::memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.f : 0.f;
  for (int sweep = 0; sweep < 60; sweep++) {
    float off = 0; for (int p = 0; p < 6; p++) for (int q = p + 1; q < 6; q++) off += std::fabs(A[p * 6 + q]);
    if (off < 1e-30f) break;
    for (int p = 0; p < 6; p++) for (int q = p + 1; q < 6; q++) {
      const float apq = A[p * 6 + q];
      if (std::fabs(apq) < FLT_MIN) continue;
      const float theta = (A[q * 7] - A[p * 7]) / (2.f * apq);
      const float t = (theta >= 0.f ? 1.f : -1.f) / (std::fabs(theta) + std::sqrt(theta * theta + 1.f));
      const float c = 1.f / std::sqrt(t * t + 1.f), s = t * c;
      for (int k = 0; k < 6; k++) { const float akp = A[k * 6 + p], akq = A[k * 6 + q]; A[k * 6 + p] = c * akp - s * akq; A[k * 6 + q] = s * akp + c * akq; }
      for (int k = 0; k < 6; k++) { const float apk = A[p * 6 + k], aqk = A[q * 6 + k]; A[p * 6 + k] = c * apk - s * aqk; A[q * 6 + k] = s * apk + c * aqk; }
      for (int k = 0; k < 6; k++) { const float vpk = V[p * 6 + k], vqk = V[q * 6 + k]; V[p * 6 + k] = c * vpk - s * vqk; V[q * 6 + k] = s * vpk + c * vqk; }
    }
  }
  for (int i = 0; i < 6; i++) E[i] = A[i * 7];
  for (int i = 0; i < 5; i++) for (int j = 0; j < 5 - i; j++) if (E[j] < E[j + 1]) {
    std::swap(E[j], E[j + 1]);
    for (int k = 0; k < 6; k++) std::swap(V[j * 6 + k], V[(j + 1) * 6 + k]);
  }
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
struct S2mScratch { float4* feat = nullptr; double *part = nullptr, *sum = nullptr; unsigned char* sel = nullptr; float4* coeff = nullptr;
                    size_t feat_cap = 0, part_cap = 0, sum_cap = 0, sel_cap = 0, coeff_cap = 0; };
}  // namespace rolo
extern "C" void rolo_s2m_destroy(rolo_ctx* c) {   // called by rolo_ctx_destroy
  S2mScratch* W = static_cast<S2mScratch*>(*ctx_s2m_slot(c));
  if (!W) return;
  for (void* p : {(void*)W->feat, (void*)W->part, (void*)W->sum, (void*)W->sel, (void*)W->coeff}) if (p) (void)hipFree(p);
  delete W; *ctx_s2m_slot(c) = nullptr;
}

extern "C" int rolo_scan2map_optimize(rolo_ctx* c, const float* corner, int n_corner, const float* surf, int n_surf, const float* map_corner, int m_corner,
                                      const float* map_surf, int m_surf, float* transformTobeMapped, int edge_min, int surf_min, rolo_scan2map_stats* stats,
                                      unsigned char* selected_out, float* coeff_out) {
  if (!c || !transformTobeMapped || n_corner < 0 || n_surf < 0 || m_corner < 0 || m_surf < 0 || (n_corner && !corner) || (n_surf && !surf) ||
      (m_corner && !map_corner) || (m_surf && !map_surf))
    return ROLO_EINVAL;
  rolo_scan2map_stats st{};
  if (stats) *stats = st;
  // :689 — "if (laserCloudCornerLastDSNum > edgeFeatureMinValidNum && laserCloudSurfLastDSNum > surfFeatureMinValidNum)"
  if (!(n_corner > edge_min && n_surf > surf_min)) { st.skipped = 1; if (stats) *stats = st; return ROLO_OK; }
  // a sub-map without five points cannot answer a 5-NN query: the reference's association then selects nothing from it (and reads
  // pointSearchSqDis[4] of a shorter result, :745 / :852) — nothing to optimise against; report it as skipped instead of an error
  if (m_corner < 5 || m_surf < 5) { st.skipped = 2; if (stats) *stats = st; return ROLO_OK; }
  KnnPair maps{};
  int rc = ctx_build_map_trees(c, map_corner, m_corner, map_surf, m_surf, 4, &maps);
  if (rc) return rc;
  hipStream_t s = ctx_stream(c);
  const int n = n_corner + n_surf;
  const int grid = (n + S2M_THREADS - 1) / S2M_THREADS;
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
  float4* d_feat = W->feat; double *d_part = W->part, *d_sum = W->sum; unsigned char* d_sel = selected_out ? W->sel : nullptr; float4* d_coeff = coeff_out ? W->coeff : nullptr;
  auto cleanup = [&]() {};
  if (hipMemcpyAsync(d_feat, corner, sizeof(float4) * (size_t)n_corner, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_feat + n_corner, surf, sizeof(float4) * (size_t)n_surf, hipMemcpyHostToDevice, s) != hipSuccess) { cleanup(); ctx_set_error("upload failed (scan2map)"); return ROLO_EHIP; }
  S2mArgs A{};
  A.feat = d_feat; A.n_corner = n_corner; A.n_surf = n_surf; A.map[0] = maps.c[0]; A.map[1] = maps.c[1]; A.partials = d_part; A.selected = d_sel; A.coeff = d_coeff;
  float* tf = transformTobeMapped;
  bool isDegenerate = false;
  float matP[36]; for (int i = 0; i < 36; i++) matP[i] = (i % 7 == 0) ? 1.f : 0.f;
  double h_sum[S2M_NV];
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
    s2m_kernel<<<grid, S2M_THREADS, 0, s>>>(A);
    SCHK(hipGetLastError());
    s2m_sum_kernel<<<1, 64, 0, s>>>(d_part, grid, d_sum);
    SCHK(hipGetLastError());
    if (hipMemcpyAsync(h_sum, d_sum, sizeof(h_sum), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { cleanup(); ctx_set_error("scan2map iteration failed"); return ROLO_EHIP; }
    st.iterations = iterCount + 1;
    st.n_selected = (int)(h_sum[27] + 0.5);
    if (st.n_selected < 50) break;   // LMOptimization returns false without touching the pose: the remaining iterations would repeat this one
    float AtA[36], AtB[6], X[6];
    { int t = 0; for (int r = 0; r < 6; r++) for (int cc = 0; cc <= r; cc++) { AtA[r * 6 + cc] = AtA[cc * 6 + r] = (float)h_sum[t]; t++; } }
    for (int r = 0; r < 6; r++) AtB[r] = (float)h_sum[21 + r];
    if (!solve_qr6f(AtA, AtB, X)) { for (int r = 0; r < 6; r++) X[r] = 0.f; }
    if (iterCount == 0) {   // degeneracy of the first linearisation :1004-1026
      float E[6], V[36], V2[36], Vi[36];
      eigen_sym6f(AtA, E, V);
      std::memcpy(V2, V, sizeof(V2));
      isDegenerate = false;
      for (int i = 5; i >= 0; i--) {
        if (E[i] < 100.f) { for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0; isDegenerate = true; } else break;
      }
      if (invert6f(V, Vi)) { for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) { float a = 0; for (int k = 0; k < 6; k++) a += Vi[r * 6 + k] * V2[k * 6 + cc]; matP[r * 6 + cc] = a; } }
    }
    if (isDegenerate) { float X2[6]; std::memcpy(X2, X, sizeof(X2)); for (int r = 0; r < 6; r++) { float a = 0; for (int k = 0; k < 6; k++) a += matP[r * 6 + k] * X2[k]; X[r] = a; } }
    for (int r = 0; r < 6; r++) tf[r] += X[r];
    const float r2d = 180.0f / (float)M_PI;
    const float deltaR = std::sqrt(std::pow(X[0] * r2d, 2.f) + std::pow(X[1] * r2d, 2.f) + std::pow(X[2] * r2d, 2.f));
    const float deltaT = std::sqrt(std::pow(X[3] * 100, 2.f) + std::pow(X[4] * 100, 2.f) + std::pow(X[5] * 100, 2.f));
    if (deltaR < 0.05f && deltaT < 0.05f) { st.converged = 1; break; }
  }
  st.degenerate = isDegenerate ? 1 : 0;
  if (selected_out) SCHK(hipMemcpy(selected_out, d_sel, (size_t)n, hipMemcpyDeviceToHost));
  if (coeff_out) SCHK(hipMemcpy(coeff_out, d_coeff, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
  if (stats) *stats = st;
  return ROLO_OK;
}

// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header). Exact k-nearest-neighbour search shared by the registration oracle
// (rolo_oracle.cpp: pcl::search::KdTree, rot_vgicp_impl.hpp:436) and the back-end oracle (rolo_oracle_backend.cpp: pcl::KdTreeFLANN::nearestKSearch,
// src/backMapping.cpp:738, :845).
#pragma once
#include <algorithm>
#include <cfloat>
#include <numeric>
#include <vector>

namespace orc {

struct P4 { float x, y, z, w; };  // pcl::PointXYZI data[4]: x,y,z,1

// ------------------------------------------------------------------------------------------------
// Exact k-nearest-neighbour search (stands in for pcl::search::KdTree / FLANN single kd-tree,
// rot_vgicp_impl.hpp:436): 3-D float squared L2 accumulated as ((dx*dx)+(dy*dy))+(dz*dz), query point
// included, sorted ascending. FLANN's tie order is arbitrary; this restatement fixes it to (d2, index)
// lexicographic so the neighbour set is a pure function of the cloud.
// ------------------------------------------------------------------------------------------------
struct KdNode { int dim; float split; int left, right; int begin, end; };

struct KdTree {
  const std::vector<P4>* pts = nullptr;
  std::vector<int> order;
  std::vector<KdNode> nodes;
  static constexpr int LEAF = 12;

  static inline float coord(const P4& p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

  int build_rec(int b, int e) {
    KdNode nd; nd.begin = b; nd.end = e; nd.left = nd.right = -1; nd.dim = -1; nd.split = 0;
    int id = (int)nodes.size();
    nodes.push_back(nd);
    if (e - b <= LEAF) return id;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = b; i < e; i++) {
      const P4& p = (*pts)[order[i]];
      for (int d = 0; d < 3; d++) { float c = coord(p, d); mn[d] = std::min(mn[d], c); mx[d] = std::max(mx[d], c); }
    }
    int dim = 0; float ext = mx[0] - mn[0];
    for (int d = 1; d < 3; d++) if (mx[d] - mn[d] > ext) { ext = mx[d] - mn[d]; dim = d; }
    if (!(ext > 0)) return id;  // all points identical: keep as one leaf
    int mid = (b + e) / 2;
    std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](int a, int c) {
      float ca = coord((*pts)[a], dim), cc = coord((*pts)[c], dim);
      return ca < cc || (ca == cc && a < c);
    });
    float split = coord((*pts)[order[mid]], dim);
    int l = build_rec(b, mid);
    int r = build_rec(mid, e);
    nodes[id].dim = dim; nodes[id].split = split; nodes[id].left = l; nodes[id].right = r;
    return id;
  }
  void build(const std::vector<P4>& p) {
    pts = &p;
    order.resize(p.size());
    std::iota(order.begin(), order.end(), 0);
    nodes.clear();
    nodes.reserve(p.size() / 4 + 16);
    if (!p.empty()) build_rec(0, (int)p.size());
  }

  struct Cand { float d2; int idx; };
  static inline bool worse(const Cand& a, const Cand& b) { return a.d2 > b.d2 || (a.d2 == b.d2 && a.idx > b.idx); }

  // heap: max-heap under (d2, idx) order, size <= k
  void search_rec(int nid, const P4& q, int k, std::vector<Cand>& heap) const {
    const KdNode& nd = nodes[nid];
    if (nd.dim < 0) {
      for (int i = nd.begin; i < nd.end; i++) {
        int j = order[i];
        const P4& p = (*pts)[j];
        float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
        float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
        Cand c{d2, j};
        if ((int)heap.size() < k) {
          heap.push_back(c);
          std::push_heap(heap.begin(), heap.end(), [](const Cand& a, const Cand& b) { return worse(b, a); });
        } else if (worse(heap.front(), c)) {
          std::pop_heap(heap.begin(), heap.end(), [](const Cand& a, const Cand& b) { return worse(b, a); });
          heap.back() = c;
          std::push_heap(heap.begin(), heap.end(), [](const Cand& a, const Cand& b) { return worse(b, a); });
        }
      }
      return;
    }
    float diff = coord(q, nd.dim) - nd.split;
    int near = diff < 0 ? nd.left : nd.right, far = diff < 0 ? nd.right : nd.left;
    search_rec(near, q, k, heap);
    float pd2 = diff * diff;
    // left holds coord <= split (ties by index may sit on either side), so explore on <=
    if ((int)heap.size() < k || pd2 <= heap.front().d2) search_rec(far, q, k, heap);
  }
  // returns number found (min(k, n)); results sorted ascending by (d2, idx)
  int knn(const P4& q, int k, int* idx, float* d2) const {
    std::vector<Cand> heap;
    heap.reserve(k + 1);
    if (!nodes.empty()) search_rec(0, q, k, heap);
    std::sort(heap.begin(), heap.end(), [](const Cand& a, const Cand& b) { return worse(b, a); });
    for (size_t i = 0; i < heap.size(); i++) { idx[i] = heap[i].idx; if (d2) d2[i] = heap[i].d2; }
    return (int)heap.size();
  }
};


}  // namespace orc

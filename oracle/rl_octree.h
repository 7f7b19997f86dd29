// oracle/rl_octree.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's incremental pointer octree
// (reference include/fast_limo/Objects/Octree.hpp):
//   Heap            :45-88    -> KnnHeap
//   Octant          :103-132  -> Octant
//   mortonCode      :269-275
//   initialize      :282-298
//   createOctant    :301-338
//   update          :341-377   (root growth by doubling :354-374)
//   updateOctant    :380-432   (leaf split :385-395, batch-drop downsampling :399-401)
//   overlaps        :435-450
//   knn             :526-599   (descend into the query's Morton child first, then the
//                               fixed sibling order `ordered_indices` :144-153)
// Quirks kept on purpose (SURVEY.md section 8 a-notes 1, 9, 10):
//   * setBucketSize is a no-op in the reference (:178-180) => bucket_size_ stays 32.
//   * the initial build never downsamples; incremental insert drops the whole incoming
//     batch of a min-extent leaf that already holds > bucket_size_/8 points.
// PARITY UNPINNED (no reference tests exist); cross-checked against brute force and
// scipy.spatial.cKDTree in tests/.
#pragma once
#include <vector>
#include <limits>
#include <cstddef>
#include "rl_linalg.h"

namespace oracle {

struct KnnHeap {                       // Octree.hpp:45-88
  struct Entry { float dist; V3f point; };
  size_t capacity, count;
  std::vector<Entry> data;
  explicit KnnHeap(size_t cap) : capacity(cap), count(0), data(cap) {
    for (auto& e : data) e.dist = std::numeric_limits<float>::max();
  }
  bool full() const { return count == capacity; }
  float worstDist() const { return full() ? data[count - 1].dist : std::numeric_limits<float>::max(); }
  void addPoint(const V3f& p, float dist) {      // :72-87 sorted insertion, stable on ties
    if (full() && dist >= data[count - 1].dist) return;
    if (count < capacity) ++count;
    int i = static_cast<int>(count) - 1;
    while (i > 0 && data[i - 1].dist > dist) { data[i] = data[i - 1]; --i; }
    data[i].dist = dist;
    data[i].point = p;
  }
};

struct Octant {                        // Octree.hpp:103-132
  V3f centroid;
  float extent;
  std::vector<V3f> points;
  Octant** child;
  Octant() : extent(0.f), child(nullptr) {}
  ~Octant() {
    if (child) {
      for (int i = 0; i < 8; i++) delete child[i];
      delete[] child;
    }
  }
  void init_child() { child = new Octant*[8](); }
};

struct Octree {
  Octant* root_ = nullptr;
  size_t num_points_ = 0;
  size_t bucket_size_ = 32;            // Octree.hpp:155 (and :178-180: the setter is a no-op)
  float min_extent_ = 0.2f;
  bool downsample_ = true;
  mutable long long dist_evals_ = 0;   // instrumentation only: leaf-point distance evaluations

  ~Octree() { delete root_; }
  void clear() { delete root_; root_ = nullptr; }
  void setBucketSize(size_t) { /* reference bug kept: parameter shadows the member */ }
  void setMinExtent(float e) { min_extent_ = e; }
  void setDownsample(bool d) { downsample_ = d; }
  size_t size() const { return num_points_; }

  static size_t mortonCode(const V3f& p, const V3f& c) {      // :269-275
    size_t out = 0;
    if (p.x > c.x) out |= 1;
    if (p.y > c.y) out |= 2;
    if (p.z > c.z) out |= 4;
    return out;
  }

  // processPoints :235-267 (NaN filter + bounding box)
  static std::vector<V3f> processPoints(const float* xyz, size_t n, size_t stride_f, V3f& mn, V3f& mx) {
    std::vector<V3f> out;
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
      float x = xyz[i * stride_f], y = xyz[i * stride_f + 1], z = xyz[i * stride_f + 2];
      if (std::isnan(x) || std::isnan(y) || std::isnan(z)) continue;
      out.push_back(V3f(x, y, z));
      mn.x = x < mn.x ? x : mn.x; mn.y = y < mn.y ? y : mn.y; mn.z = z < mn.z ? z : mn.z;
      mx.x = x > mx.x ? x : mx.x; mx.y = y > mx.y ? y : mx.y; mx.z = z > mx.z ? z : mx.z;
    }
    return out;
  }

  void initialize(const float* xyz, size_t n, size_t stride_f) {   // :282-298
    clear();
    const float fmax = std::numeric_limits<float>::max();
    V3f mn(fmax, fmax, fmax), mx(-fmax, -fmax, -fmax);
    std::vector<V3f> pts = processPoints(xyz, n, stride_f, mn, mx);
    if (pts.empty()) return;
    V3f ext = 0.5f * (mx - mn);
    V3f cen = mn + ext;
    float me = ext.x;
    if (ext.y > me) me = ext.y;
    if (ext.z > me) me = ext.z;
    root_ = createOctant(cen, me, pts);
  }

  Octant* createOctant(const V3f& centroid, float maxextent, const std::vector<V3f>& points) {  // :301-338
    Octant* o = new Octant;
    o->centroid = centroid;
    o->extent = maxextent;
    static const float factor[] = {-0.5f, 0.5f};
    if (points.size() > bucket_size_ && maxextent > 2 * min_extent_) {
      o->init_child();
      std::vector<std::vector<V3f>> cp(8);
      for (const auto& p : points) cp[mortonCode(p, centroid)].push_back(p);
      for (int i = 0; i < 8; i++) {
        if (cp[i].empty()) continue;
        V3f cc(centroid.x + factor[(i & 1) > 0] * maxextent,
               centroid.y + factor[(i & 2) > 0] * maxextent,
               centroid.z + factor[(i & 4) > 0] * maxextent);
        o->child[i] = createOctant(cc, maxextent * 0.5f, cp[i]);
      }
    } else {
      num_points_ += points.size();
      o->points = points;
    }
    return o;
  }

  void update(const float* xyz, size_t n, size_t stride_f) {       // :341-377
    if (root_ == nullptr) { initialize(xyz, n, stride_f); return; }
    const float fmax = std::numeric_limits<float>::max();
    V3f mn(fmax, fmax, fmax), mx(-fmax, -fmax, -fmax);
    std::vector<V3f> pts = processPoints(xyz, n, stride_f, mn, mx);
    static const float factor[] = {-0.5f, 0.5f};
    auto expandTree = [&](const V3f& b) {
      for (;;) {
        V3f d = b - root_->centroid;
        float m = std::fabs(d.x);
        if (std::fabs(d.y) > m) m = std::fabs(d.y);
        if (std::fabs(d.z) > m) m = std::fabs(d.z);
        if (!(m > root_->extent)) break;
        float pe = 2 * root_->extent;
        V3f pc(root_->centroid.x + factor[b.x > root_->centroid.x] * pe,
               root_->centroid.y + factor[b.y > root_->centroid.y] * pe,
               root_->centroid.z + factor[b.z > root_->centroid.z] * pe);
        Octant* o = new Octant;
        o->centroid = pc;
        o->extent = pe;
        o->init_child();
        o->child[mortonCode(root_->centroid, pc)] = root_;
        root_ = o;
      }
    };
    expandTree(mx);
    expandTree(mn);
    updateOctant(root_, pts);
  }

  void updateOctant(Octant*& o, const std::vector<V3f>& points) {  // :380-432
    static const float factor[] = {-0.5f, 0.5f};
    if (o->child == nullptr) {
      if (o->points.size() + points.size() > bucket_size_ && o->extent > 2 * min_extent_) {
        num_points_ -= o->points.size();
        o->points.insert(o->points.end(), points.begin(), points.end());
        Octant* no = createOctant(o->centroid, o->extent, o->points);
        delete o;
        o = no;
      } else {
        if (downsample_ && o->extent <= 2 * min_extent_ && o->points.size() > bucket_size_ / 8) return;
        o->points.insert(o->points.end(), points.begin(), points.end());
        num_points_ += points.size();
      }
    } else {
      std::vector<std::vector<V3f>> cp(8);
      for (const auto& p : points) cp[mortonCode(p, o->centroid)].push_back(p);
      for (size_t i = 0; i < 8; i++) {
        if (cp[i].empty()) continue;
        if (o->child[i] == nullptr) {
          V3f cc(o->centroid.x + factor[(i & 1) > 0] * o->extent,
                 o->centroid.y + factor[(i & 2) > 0] * o->extent,
                 o->centroid.z + factor[(i & 4) > 0] * o->extent);
          o->child[i] = createOctant(cc, o->extent * 0.5f, cp[i]);
        } else {
          updateOctant(o->child[i], cp[i]);
        }
      }
    }
  }

  static bool overlaps(const Octant* o, const V3f& q, float sqr_radius) {   // :435-450
    V3f d(std::fabs(q.x - o->centroid.x) - o->extent,
          std::fabs(q.y - o->centroid.y) - o->extent,
          std::fabs(q.z - o->centroid.z) - o->extent);
    if ((d.x > 0 && d.x * d.x > sqr_radius) || (d.y > 0 && d.y * d.y > sqr_radius) ||
        (d.z > 0 && d.z * d.z > sqr_radius))
      return false;
    int num_less = (d.x < 0) + (d.y < 0) + (d.z < 0);
    if (num_less > 1) return true;
    V3f c(d.x > 0.f ? d.x : 0.f, d.y > 0.f ? d.y : 0.f, d.z > 0.f ? d.z : 0.f);
    return sqnorm3(c) < sqr_radius;
  }

  static bool inside(const Octant* o, const V3f& q, float radius) {         // :560-567
    V3f d(o->extent - std::fabs(q.x - o->centroid.x),
          o->extent - std::fabs(q.y - o->centroid.y),
          o->extent - std::fabs(q.z - o->centroid.z));
    return (d.x < 0 || d.x * d.x < radius) ? false
         : (d.y < 0 || d.y * d.y < radius) ? false
         : (d.z < 0 || d.z * d.z < radius) ? false : true;
  }

  bool knn_rec(const Octant* o, const V3f& q, KnnHeap& heap, long long& evals) const {   // :558-599
    static const size_t ordered[8][7] = {                                   // :144-153
        {1, 2, 4, 3, 5, 6, 7}, {0, 3, 5, 2, 4, 7, 6}, {0, 3, 6, 1, 4, 7, 5}, {1, 2, 7, 0, 5, 6, 4},
        {0, 5, 6, 1, 2, 7, 3}, {1, 4, 7, 0, 3, 6, 2}, {2, 4, 7, 0, 3, 5, 1}, {3, 5, 6, 1, 2, 4, 0}};
    if (o->child == nullptr) {
      for (const auto& p : o->points) {
        float sq = sqnorm3(q - p);          // (query - p).squaredNorm() :572
        heap.addPoint(p, sq);
      }
      evals += (long long)o->points.size();
      return heap.full() && inside(o, q, heap.worstDist());
    }
    size_t morton = mortonCode(q, o->centroid);
    if (o->child[morton] != nullptr) {
      if (knn_rec(o->child[morton], q, heap, evals)) return true;
    }
    for (int i = 0; i < 7; i++) {
      size_t c = ordered[morton][i];
      if (o->child[c] == nullptr) continue;
      if (heap.full() && !overlaps(o->child[c], q, heap.worstDist())) continue;
      if (knn_rec(o->child[c], q, heap, evals)) return true;
    }
    return heap.full() && inside(o, q, heap.worstDist());
  }

  // public knn :526-555.  Returns the number of neighbours found (<= k), ascending sq. distance.
  int knn(const V3f& q, int k, V3f* nbr, float* sqd, long long* evals_out = nullptr) const {
    if (root_ == nullptr) return 0;
    KnnHeap heap(k);
    long long evals = 0;
    knn_rec(root_, q, heap, evals);
    for (size_t i = 0; i < heap.count; i++) { nbr[i] = heap.data[i].point; sqd[i] = heap.data[i].dist; }
    if (evals_out) *evals_out = evals;
    return (int)heap.count;
  }

  void get_points(const Octant* o, std::vector<V3f>& out) const {           // :217-228
    if (!o) return;
    if (!o->child) out.insert(out.end(), o->points.begin(), o->points.end());
    else for (int i = 0; i < 8; i++) get_points(o->child[i], out);
  }
};

}  // namespace oracle

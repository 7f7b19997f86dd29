// fast_limo_amd/csrc/hip/flimo_insert.cpp -- see flimo_insert.h.
// Semantics follow the reference's Octree::initialize / update / updateOctant / createOctant
// (Objects/Octree.hpp:282-432) including its float32 centroid arithmetic, which fixes the lattice.
#include "flimo_insert.h"
#include <vector>
#include <cmath>
#include <cfloat>
#include <utility>

namespace flimo {

namespace {
struct P3 { float x, y, z; };

struct Cell {
  float cx, cy, cz, half;   // centre and half edge
  int32_t kid[8];           // node indices, -1 = absent
  bool split;               // true: internal node
  std::vector<P3> pts;      // leaf payload
  Cell() : cx(0), cy(0), cz(0), half(0), split(false) { for (int i = 0; i < 8; i++) kid[i] = -1; }
};

inline int octant_of(const P3& p, float cx, float cy, float cz) {   // Octree.hpp:269-275
  return (p.x > cx ? 1 : 0) | (p.y > cy ? 2 : 0) | (p.z > cz ? 4 : 0);
}
}  // namespace

struct InsertBook {
  std::vector<Cell> pool;
  std::vector<int32_t> free_list;
  int32_t root = -1;
  size_t stored = 0;
  size_t bucket = 32;
  float min_half = 0.2f;
  bool downsample = true;

  int32_t alloc() {
    if (!free_list.empty()) { int32_t i = free_list.back(); free_list.pop_back(); pool[i] = Cell(); return i; }
    pool.emplace_back();
    return (int32_t)pool.size() - 1;
  }
  void release(int32_t i) {
    if (i < 0) return;
    for (int k = 0; k < 8; k++) release(pool[i].kid[k]);
    pool[i].pts.clear();
    pool[i].pts.shrink_to_fit();
    free_list.push_back(i);
  }

  // createOctant (Octree.hpp:301-338): split while the cell holds more than `bucket` points and is
  // larger than the minimum extent; nothing is ever dropped here.
  int32_t grow(float cx, float cy, float cz, float half, std::vector<P3>& pts) {
    const int32_t id = alloc();
    pool[id].cx = cx; pool[id].cy = cy; pool[id].cz = cz; pool[id].half = half;
    if (pts.size() > bucket && half > 2 * min_half) {
      pool[id].split = true;
      std::vector<P3> part[8];
      for (const P3& p : pts) part[octant_of(p, cx, cy, cz)].push_back(p);
      static const float f[2] = {-0.5f, 0.5f};
      for (int k = 0; k < 8; k++) {
        if (part[k].empty()) continue;
        const float kx = cx + f[(k & 1) ? 1 : 0] * half, ky = cy + f[(k & 2) ? 1 : 0] * half,
                    kz = cz + f[(k & 4) ? 1 : 0] * half;
        const int32_t c = grow(kx, ky, kz, half * 0.5f, part[k]);
        pool[id].kid[k] = c;
      }
    } else {
      stored += pts.size();
      pool[id].pts = std::move(pts);
    }
    return id;
  }

  struct Item { P3 p; uint32_t src; };

  // updateOctant (Octree.hpp:380-432)
  void route(int32_t& id, std::vector<Item>& items, unsigned char* keep) {
    Cell* c = &pool[id];
    if (!c->split) {
      if (c->pts.size() + items.size() > bucket && c->half > 2 * min_half) {
        stored -= c->pts.size();
        std::vector<P3> all = std::move(c->pts);
        all.reserve(all.size() + items.size());
        for (const Item& it : items) all.push_back(it.p);
        const float cx = c->cx, cy = c->cy, cz = c->cz, half = c->half;
        release(id);
        id = grow(cx, cy, cz, half, all);
      } else {
        if (downsample && c->half <= 2 * min_half && c->pts.size() > bucket / 8) {
          for (const Item& it : items) keep[it.src] = 0;     // the whole batch for this leaf is dropped
          return;
        }
        for (const Item& it : items) c->pts.push_back(it.p);
        stored += items.size();
      }
      return;
    }
    std::vector<Item> part[8];
    for (const Item& it : items) part[octant_of(it.p, c->cx, c->cy, c->cz)].push_back(it);
    static const float f[2] = {-0.5f, 0.5f};
    for (int k = 0; k < 8; k++) {
      if (part[k].empty()) continue;
      c = &pool[id];   // pool may have been reallocated by a previous iteration
      if (c->kid[k] < 0) {
        const float kx = c->cx + f[(k & 1) ? 1 : 0] * c->half, ky = c->cy + f[(k & 2) ? 1 : 0] * c->half,
                    kz = c->cz + f[(k & 4) ? 1 : 0] * c->half;
        const float kh = c->half * 0.5f;
        std::vector<P3> pts;
        pts.reserve(part[k].size());
        for (const Item& it : part[k]) pts.push_back(it.p);
        const int32_t nc = grow(kx, ky, kz, kh, pts);
        pool[id].kid[k] = nc;
      } else {
        int32_t child = c->kid[k];
        route(child, part[k], keep);
        pool[id].kid[k] = child;
      }
    }
  }

  void update(const float* xyz, size_t n, unsigned char* keep) {
    for (size_t i = 0; i < n; i++) keep[i] = 1;
    if (n == 0) return;
    P3 mn{FLT_MAX, FLT_MAX, FLT_MAX}, mx{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; i++) {
      const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
      mn.x = x < mn.x ? x : mn.x; mn.y = y < mn.y ? y : mn.y; mn.z = z < mn.z ? z : mn.z;
      mx.x = x > mx.x ? x : mx.x; mx.y = y > mx.y ? y : mx.y; mx.z = z > mx.z ? z : mx.z;
    }
    if (root < 0) {
      // initialize (Octree.hpp:282-298): root cube from the bounding box, no down-sampling
      const float ex = 0.5f * (mx.x - mn.x), ey = 0.5f * (mx.y - mn.y), ez = 0.5f * (mx.z - mn.z);
      const float cx = mn.x + ex, cy = mn.y + ey, cz = mn.z + ez;
      float half = ex;
      if (ey > half) half = ey;
      if (ez > half) half = ez;
      std::vector<P3> pts(n);
      for (size_t i = 0; i < n; i++) pts[i] = P3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      root = grow(cx, cy, cz, half, pts);
      return;
    }
    // root growth by doubling towards out-of-bounds corners (Octree.hpp:354-374): max first, then min
    static const float f[2] = {-0.5f, 0.5f};
    const P3 corners[2] = {mx, mn};
    for (int ci = 0; ci < 2; ci++) {
      const P3 b = corners[ci];
      for (;;) {
        const Cell& r = pool[root];
        float m = std::fabs(b.x - r.cx);
        const float my = std::fabs(b.y - r.cy), mz = std::fabs(b.z - r.cz);
        if (my > m) m = my;
        if (mz > m) m = mz;
        if (!(m > r.half)) break;
        const float ph = 2 * r.half;
        const float px = r.cx + f[b.x > r.cx ? 1 : 0] * ph, py = r.cy + f[b.y > r.cy ? 1 : 0] * ph,
                    pz = r.cz + f[b.z > r.cz ? 1 : 0] * ph;
        const P3 old_c{r.cx, r.cy, r.cz};
        const int32_t nr = alloc();
        pool[nr].cx = px; pool[nr].cy = py; pool[nr].cz = pz; pool[nr].half = ph;
        pool[nr].split = true;
        pool[nr].kid[octant_of(old_c, px, py, pz)] = root;
        root = nr;
      }
    }
    std::vector<Item> items(n);
    for (size_t i = 0; i < n; i++) items[i] = Item{P3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, (uint32_t)i};
    route(root, items, keep);
  }
};

// flatten the live nodes (depth-first from the root) for the device-side book
void insert_book_export(const InsertBook* b, std::vector<float>& c4, std::vector<int>& child, std::vector<int>& cnt, int* root_out) {
  c4.clear(); child.clear(); cnt.clear();
  *root_out = -1;
  if (b->root < 0) return;
  std::vector<int> stack, newid(b->pool.size(), -1), order;
  stack.push_back(b->root);
  while (!stack.empty()) {
    const int id = stack.back();
    stack.pop_back();
    newid[id] = (int)order.size();
    order.push_back(id);
    for (int k = 0; k < 8; k++) if (b->pool[id].kid[k] >= 0) stack.push_back(b->pool[id].kid[k]);
  }
  c4.resize(order.size() * 4); child.assign(order.size() * 8, -1); cnt.resize(order.size());
  for (size_t i = 0; i < order.size(); i++) {
    const Cell& c = b->pool[order[i]];
    c4[i * 4] = c.cx; c4[i * 4 + 1] = c.cy; c4[i * 4 + 2] = c.cz; c4[i * 4 + 3] = c.half;
    for (int k = 0; k < 8; k++) if (c.kid[k] >= 0) child[i * 8 + k] = newid[c.kid[k]];
    cnt[i] = c.split ? -1 : (int)c.pts.size();
  }
  *root_out = newid[b->root];
}

InsertBook* insert_book_create() { return new InsertBook(); }
void insert_book_destroy(InsertBook* b) { delete b; }
void insert_book_config(InsertBook* b, float min_extent, bool downsample) { b->min_half = min_extent; b->downsample = downsample; }
void insert_book_clear(InsertBook* b) {
  b->pool.clear(); b->free_list.clear(); b->root = -1; b->stored = 0;
}
size_t insert_book_size(const InsertBook* b) { return b->stored; }
void insert_book_update(InsertBook* b, const float* xyz, size_t n, unsigned char* keep) { b->update(xyz, n, keep); }

}  // namespace flimo

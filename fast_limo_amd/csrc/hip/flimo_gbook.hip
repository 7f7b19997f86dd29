// fast_limo_amd/csrc/hip/flimo_gbook.hip  -- gfx950 (MI355X) only.
//
// GPU-resident bookkeeping of WHICH points the map stores = the reference's incremental octree insert
// rule (Objects/Octree.hpp:341-432: update / updateOctant / createOctant), reproduced exactly, on the
// device.  The k-NN index stays the uniform grid; this structure only
// decides keep / drop and therefore needs, per octree node, its cube (centre, half edge in float32,
// computed with the reference's arithmetic), its 8 child links and -- for leaves -- the point count;
// every stored map point remembers its leaf (pt_leaf) so that a splitting leaf can be re-partitioned.
//
// One batch (Octree::update) on the device:
//   route    every incoming point descends from the root to a leaf or to a missing child slot
//   sort     stable radix sort of the batch by destination (rocPRIM)
//   decide   one thread per destination group applies updateOctant's rule:
//              leaf:  n + g > 32 and half > 2*min_extent  -> SPLIT (rebuild that subtree, keep all)
//                     down-sampling and half <= 2*min_extent and n > 4 -> DROP the whole group
//                     else APPEND
//              missing child -> CREATE a subtree from the group (keep all)
//   compact  kept points are appended to the map in batch order
//   gather   points already stored in splitting leaves are collected (one pass over pt_leaf)
//   build    one thread per SPLIT / CREATE item runs createOctant (split while count > 32 and
//            half > 2*min_extent) on its index list, allocating nodes from the pool
// The initial batch (Octree::initialize) is built on the host once (flimo_insert.cpp) and exported.
#include <hip/hip_runtime.h>
#include "flimo_prims.h"
#include <float.h>
#include <vector>
#include "flimo_types.h"
#include "flimo_kernels.h"
#include "flimo_gbook.h"

#pragma clang fp contract(off)

namespace flimo {

static const int kBucket = 32;     // effective bucket size of the reference (Octree.hpp:155,178-180)
static const int kBigItem = 2048;  // subtree builds with more points than this get a whole 1024-thread block

__device__ __forceinline__ int octant_of(float px, float py, float pz, const float4 c) {   // Octree.hpp:269-275
  return (px > c.x ? 1 : 0) | (py > c.y ? 2 : 0) | (pz > c.z ? 4 : 0);
}

// ---- route ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gb_route_kernel(const float4* __restrict__ pts, int n, const float4* __restrict__ node_c,
                                                       const int* __restrict__ node_child, const int* __restrict__ node_cnt,
                                                       int root, uint32_t nan_key, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int node = root;
  uint32_t key;
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {     // Octree::processPoints drops NaNs (:243-244)
    keys[i] = nan_key;                                          // above every (node, child) key: sorts last
    vals[i] = (uint32_t)i;
    return;
  }
  for (;;) {
    if (node_cnt[node] >= 0) { key = (uint32_t)node * 9u + 8u; break; }        // leaf
    const int k = octant_of(p.x, p.y, p.z, node_c[node]);
    const int ch = node_child[(size_t)node * 8 + k];
    if (ch < 0) { key = (uint32_t)node * 9u + (uint32_t)k; break; }            // missing child
    node = ch;
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// leaf of every stored point after the host-built tree was exported
__global__ __launch_bounds__(256) void gb_leafof_kernel(const float4* __restrict__ pts, int n, const float4* __restrict__ node_c,
                                                        const int* __restrict__ node_child, const int* __restrict__ node_cnt,
                                                        int root, int* __restrict__ pt_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int node = root;
  while (node_cnt[node] < 0) {
    const int ch = node_child[(size_t)node * 8 + octant_of(p.x, p.y, p.z, node_c[node])];
    if (ch < 0) { node = -1; break; }       // cannot happen for points the tree was built from
    node = ch;
  }
  pt_leaf[i] = node;
}

// ---- decide -----------------------------------------------------------------------------------
struct GbItem {            // one subtree build
  int node;                // SPLIT: the leaf to rebuild in place; CREATE: parent node
  int slot;                // CREATE: child slot (0..7); SPLIT: 8
  int g_begin, g_len;      // group range in the sorted batch
  int n_old;               // SPLIT: points already in the leaf
  int seg;                 // start of its index segment in the scratch lists
};

// One thread per destination GROUP HEAD (the first sorted element of a run of equal keys): the group size comes from a
// binary search for the end of the run, the decision is stored at the head's sorted position (dec_keep / dec_assign);
// gb_apply_kernel hands it to every member.  No per-member loops: a scan entering fresh territory routes thousands of
// points to one missing child.
__device__ __forceinline__ int run_end(const uint32_t* __restrict__ keys, int i, int n, uint32_t key) {   // first j > i with keys[j] != key
  int lo = i + 1, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] == key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int run_begin(const uint32_t* __restrict__ keys, int i, uint32_t key) {        // first j <= i with keys[j] == key
  int lo = 0, hi = i;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void gb_decide_kernel(const uint32_t* __restrict__ keys, int n,
                                                        const float4* __restrict__ node_c, int* __restrict__ node_cnt,
                                                        float min_half, int downsample, uint32_t* __restrict__ dec_keep,
                                                        int* __restrict__ dec_assign /* leaf, or -1-item */,
                                                        GbItem* __restrict__ items, int* __restrict__ counters /* [0] items, [1] seg cursor, [3] big items */,
                                                        int* __restrict__ node_item, int* __restrict__ big, uint32_t nan_key,
                                                        int* __restrict__ cursor /* per item: members gathered so far */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;            // not a group head
  const int g = run_end(keys, i, n, key) - i;
  if (key == nan_key) { dec_keep[i] = 0u; dec_assign[i] = -1; return; }      // non-finite points: never stored
  const int node = (int)(key / 9u), slot = (int)(key % 9u);
  if (slot == 8) {
    const int cnt = node_cnt[node];
    const float half = node_c[node].w;
    if (cnt + g > kBucket && half > 2 * min_half) {                      // SPLIT (Octree.hpp:385-395)
      const int it = atomicAdd(&counters[0], 1);
      const int seg = atomicAdd(&counters[1], cnt + g);
      items[it] = GbItem{node, 8, i, g, cnt, seg};
      cursor[it] = 0;
      node_item[node] = it;
      if (cnt + g > kBigItem) big[atomicAdd(&counters[3], 1)] = it;
      dec_keep[i] = 1u; dec_assign[i] = -1 - it;
    } else if (downsample && half <= 2 * min_half && cnt > kBucket / 8) {   // DROP (:399-401)
      dec_keep[i] = 0u; dec_assign[i] = node;
    } else {                                                             // APPEND (:403-404)
      node_cnt[node] = cnt + g;
      dec_keep[i] = 1u; dec_assign[i] = node;
    }
  } else {                                                               // CREATE (:418-426)
    const int it = atomicAdd(&counters[0], 1);
    const int seg = atomicAdd(&counters[1], g);
    items[it] = GbItem{node, slot, i, g, 0, seg};
    cursor[it] = 0;
    if (g > kBigItem) big[atomicAdd(&counters[3], 1)] = it;
    dec_keep[i] = 1u; dec_assign[i] = -1 - it;
  }
}

// every sorted element copies its group's decision to its batch point
__global__ __launch_bounds__(256) void gb_apply_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int n,
                                                       const uint32_t* __restrict__ dec_keep, const int* __restrict__ dec_assign,
                                                       unsigned char* __restrict__ keep, int* __restrict__ assign,
                                                       uint32_t* __restrict__ keep_word /* the same as a 0/1 word, for the rank scan */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int h = run_begin(keys, i, keys[i]);
  const uint32_t pt = perm[i];
  const uint32_t k = dec_keep[h];
  keep[pt] = (unsigned char)k;
  keep_word[pt] = k ? 1u : 0u;
  assign[pt] = dec_assign[h];
}

// ---- compact: append the kept batch points to the map in batch order ---------------------------
__global__ __launch_bounds__(256) void gb_flags_kernel(const unsigned char* __restrict__ keep, int n, uint32_t* __restrict__ f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = keep[i] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void gb_append_kernel(const float4* __restrict__ batch, const unsigned char* __restrict__ keep,
                                                        const uint32_t* __restrict__ rank, const int* __restrict__ assign, int n,
                                                        int map_n, float4* __restrict__ map_raw, int* __restrict__ pt_leaf,
                                                        int* __restrict__ new_index /* per batch point or -1 */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!keep[i]) { new_index[i] = -1; return; }
  const int gi = map_n + (int)rank[i];
  float4 p = batch[i];
  p.w = __uint_as_float((uint32_t)gi);
  map_raw[gi] = p;
  pt_leaf[gi] = assign[i] >= 0 ? assign[i] : -1;      // SPLIT / CREATE members get their leaf from the build
  new_index[i] = gi;
}

// ---- gather the members of every build item into its index segment -----------------------------
// (one launch: the first `old_blocks` blocks look at the stored points, the rest at the batch -- the two halves do not depend on
//  each other)
__global__ __launch_bounds__(256) void gb_gather_kernel(const int* __restrict__ pt_leaf, int map_n_old, const int* __restrict__ node_item,
                                                        const GbItem* __restrict__ items, int* __restrict__ cursor, int* __restrict__ lists,
                                                        int old_blocks, const uint32_t* __restrict__ perm, const int* __restrict__ assign,
                                                        const int* __restrict__ new_index, int n) {
  if ((int)blockIdx.x >= old_blocks) {
    const int i = ((int)blockIdx.x - old_blocks) * blockDim.x + threadIdx.x;        // sorted position
    if (i >= n) return;
    const uint32_t pt = perm[i];
    const int a = assign[pt];
    if (a >= 0 || new_index[pt] < 0) return;                    // not a member of a build item (or dropped / non-finite)
    const GbItem I = items[-1 - a];
    lists[I.seg + I.n_old + (i - I.g_begin)] = new_index[pt];
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= map_n_old) return;
  const int leaf = pt_leaf[i];
  if (leaf < 0) return;
  const int it = node_item[leaf];
  if (it < 0) return;
  const int pos = atomicAdd(&cursor[it], 1);
  lists[items[it].seg + pos] = i;
}

// ---- build: createOctant (Octree.hpp:301-338), ONE WAVE per item ---------------------------------
// The wave walks the item's subtree with an explicit stack (LDS); at every node that splits, the lanes load
// the node's points in parallel, octant counts and stable ranks come from ballots, the index list is
// partitioned through `tmp`, and lane 0 allocates the children with one atomic.
constexpr int GB_STACK = 128;      // >= 7 * max depth + 1

__global__ __launch_bounds__(256) void gb_build_kernel(const GbItem* __restrict__ items, const int* __restrict__ n_items_dev, const float4* __restrict__ map_raw,
                                                       int* __restrict__ lists, int* __restrict__ tmp, float4* __restrict__ node_c,
                                                       int* __restrict__ node_child, int* __restrict__ node_cnt, int* __restrict__ node_n,
                                                       int node_cap, float min_half, int* __restrict__ pt_leaf, int* __restrict__ node_item,
                                                       int* __restrict__ overflow) {
  __shared__ int s_node[4][GB_STACK], s_b[4][GB_STACK], s_e[4][GB_STACK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int it = blockIdx.x * 4 + wave;
  if (it >= *n_items_dev) return;             // the big-item kernel appends the small subtrees it cuts off
  const GbItem I = items[it];
  const int total = I.n_old + I.g_len;
  if (total > kBigItem) return;               // gb_build_big_kernel
  int root_node;
  if (I.slot == 8) {
    root_node = I.node;                       // rebuilt in place (`delete octant; octant = newOctant`)
    if (lane == 0) node_item[I.node] = -1;
  } else {
    int rn = 0;
    if (lane == 0) rn = atomicAdd(node_n, 1);
    root_node = __shfl(rn, 0, 64);
    if (root_node >= node_cap) { if (lane == 0) atomicExch(overflow, 1); return; }
    if (lane == 0) {
      const float4 pc = node_c[I.node];
      const float f0 = (I.slot & 1) ? 0.5f : -0.5f, f1 = (I.slot & 2) ? 0.5f : -0.5f, f2 = (I.slot & 4) ? 0.5f : -0.5f;
      node_c[root_node] = make_float4(pc.x + f0 * pc.w, pc.y + f1 * pc.w, pc.z + f2 * pc.w, pc.w * 0.5f);
      node_child[(size_t)I.node * 8 + I.slot] = root_node;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  int sp = 1;
  if (lane == 0) { s_node[wave][0] = root_node; s_b[wave][0] = I.seg; s_e[wave][0] = I.seg + total; }
  __builtin_amdgcn_wave_barrier();
  while (sp > 0) {
    sp--;
    const int nd = s_node[wave][sp], b = s_b[wave][sp], e = s_e[wave][sp];
    __builtin_amdgcn_wave_barrier();
    const float4 c = node_c[nd];
    const int cnt = e - b;
    if (cnt > kBucket && c.w > 2 * min_half) {
      // ---- octant histogram (uniform across the wave) ----
      int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int j0 = b; j0 < e; j0 += 64) {
        const int j = j0 + lane;
        int oct = -1;
        if (j < e) { const float4 p = map_raw[lists[j]]; oct = octant_of(p.x, p.y, p.z, c); }
#pragma unroll
        for (int k = 0; k < 8; k++) hist[k] += __popcll(__ballot(oct == k));
      }
      int start[8], run[8];
      int acc = b;
#pragma unroll
      for (int k = 0; k < 8; k++) { start[k] = acc; run[k] = acc; acc += hist[k]; }
      // ---- stable partition through tmp ----
      for (int j0 = b; j0 < e; j0 += 64) {
        const int j = j0 + lane;
        int oct = -1, id = 0;
        if (j < e) { id = lists[j]; const float4 p = map_raw[id]; oct = octant_of(p.x, p.y, p.z, c); }
        int dest = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const unsigned long long m = __ballot(oct == k);
          if (oct == k) dest = run[k] + __popcll(m & ((1ull << lane) - 1ull));
          run[k] += __popcll(m);
        }
        if (j < e) tmp[dest] = id;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int j = b + lane; j < e; j += 64) lists[j] = tmp[j];
      // ---- children ----
      int nch = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) nch += hist[k] > 0 ? 1 : 0;
      int base = 0;
      if (lane == 0) base = atomicAdd(node_n, nch);
      base = __shfl(base, 0, 64);
      if (base + nch > node_cap || sp + nch > GB_STACK) { if (lane == 0) atomicExch(overflow, 1); return; }
      if (lane == 0) node_cnt[nd] = -1;
      int ord = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int ch = hist[k] > 0 ? base + ord : -1;
        if (lane == 0) node_child[(size_t)nd * 8 + k] = ch;
        if (hist[k] > 0) {
          if (lane == 0) {
            const float f0 = (k & 1) ? 0.5f : -0.5f, f1 = (k & 2) ? 0.5f : -0.5f, f2 = (k & 4) ? 0.5f : -0.5f;
            node_c[ch] = make_float4(c.x + f0 * c.w, c.y + f1 * c.w, c.z + f2 * c.w, c.w * 0.5f);
            s_node[wave][sp + ord] = ch; s_b[wave][sp + ord] = start[k]; s_e[wave][sp + ord] = start[k] + hist[k];
          }
          ord++;
        }
      }
      sp += nch;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    } else {
      if (lane == 0) node_cnt[nd] = cnt;
      if (lane < 8) node_child[(size_t)nd * 8 + lane] = -1;
      for (int j = b + lane; j < e; j += 64) pt_leaf[lists[j]] = nd;
    }
  }
}

// ---- the same build for a BIG item (a scan entering fresh territory routes thousands of points to one missing
//      child): one 1024-thread block per item, octant counts and partition slots through LDS atomics (the order of
//      the points inside a leaf is irrelevant: leaves are sets) ----
__global__ __launch_bounds__(1024) void gb_build_big_kernel(GbItem* __restrict__ items, int* __restrict__ n_items_dev, int items_cap,
                                                            const int* __restrict__ big, int n_big,
                                                            const float4* __restrict__ map_raw, int* __restrict__ lists,
                                                            int* __restrict__ tmp, float4* __restrict__ node_c,
                                                            int* __restrict__ node_child, int* __restrict__ node_cnt,
                                                            int* __restrict__ node_n, int node_cap, float min_half,
                                                            int* __restrict__ pt_leaf, int* __restrict__ node_item,
                                                            int* __restrict__ overflow) {
  __shared__ int s_node[GB_STACK], s_b[GB_STACK], s_e[GB_STACK];
  __shared__ int s_hist[8], s_cur[8], s_base, s_root, s_fail;
  if ((int)blockIdx.x >= n_big) return;
  const int t = threadIdx.x, lane = t & 63;
  const GbItem I = items[big[blockIdx.x]];
  const int total = I.n_old + I.g_len;
  if (t == 0) {
    s_fail = 0;
    int root_node;
    if (I.slot == 8) {
      root_node = I.node;
      node_item[I.node] = -1;
    } else {
      root_node = atomicAdd(node_n, 1);
      if (root_node >= node_cap) { atomicExch(overflow, 1); s_fail = 1; }
      else {
        const float4 pc = node_c[I.node];
        const float f0 = (I.slot & 1) ? 0.5f : -0.5f, f1 = (I.slot & 2) ? 0.5f : -0.5f, f2 = (I.slot & 4) ? 0.5f : -0.5f;
        node_c[root_node] = make_float4(pc.x + f0 * pc.w, pc.y + f1 * pc.w, pc.z + f2 * pc.w, pc.w * 0.5f);
        node_child[(size_t)I.node * 8 + I.slot] = root_node;
      }
    }
    s_root = root_node;
    s_node[0] = root_node; s_b[0] = I.seg; s_e[0] = I.seg + total;
  }
  __threadfence_block();
  __syncthreads();
  if (s_fail) return;
  int sp = 1;
  while (sp > 0) {
    sp--;
    const int nd = s_node[sp], b = s_b[sp], e = s_e[sp];
    __syncthreads();
    const float4 c = node_c[nd];
    const int cnt = e - b;
    if (cnt > kBucket && c.w > 2 * min_half) {
      if (t < 8) s_hist[t] = 0;
      __syncthreads();
      for (int j0 = b; j0 < e; j0 += 1024) {
        const int j = j0 + t;
        int oct = -1;
        if (j < e) { const float4 p = map_raw[lists[j]]; oct = octant_of(p.x, p.y, p.z, c); }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int w = __popcll(__ballot(oct == k));
          if (lane == 0 && w) atomicAdd(&s_hist[k], w);
        }
      }
      __syncthreads();
      if (t == 0) {
        int acc = b, nch = 0;
        for (int k = 0; k < 8; k++) { s_cur[k] = acc; acc += s_hist[k]; nch += s_hist[k] > 0 ? 1 : 0; }
        const int base = atomicAdd(node_n, nch);
        if (base + nch > node_cap || sp + nch > GB_STACK) { atomicExch(overflow, 1); s_fail = 1; }
        s_base = base;
      }
      __syncthreads();
      if (s_fail) return;
      int start[8];
#pragma unroll
      for (int k = 0; k < 8; k++) start[k] = s_cur[k];
      __syncthreads();
      for (int j0 = b; j0 < e; j0 += 1024) {
        const int j = j0 + t;
        int oct = -1, id = 0;
        if (j < e) { id = lists[j]; const float4 p = map_raw[id]; oct = octant_of(p.x, p.y, p.z, c); }
        int dest = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const unsigned long long m = __ballot(oct == k);
          int wb = 0;
          if (lane == 0 && m) wb = atomicAdd(&s_cur[k], __popcll(m));
          wb = __shfl(wb, 0, 64);
          if (oct == k) dest = wb + __popcll(m & ((1ull << lane) - 1ull));
        }
        if (j < e) tmp[dest] = id;
      }
      __threadfence_block();
      __syncthreads();
      for (int j = b + t; j < e; j += 1024) lists[j] = tmp[j];
      if (t == 0) {
        node_cnt[nd] = -1;
        int ord = 0, cidx = 0;
        for (int k = 0; k < 8; k++) {
          const int h = s_hist[k];
          const int ch = h > 0 ? s_base + cidx : -1;
          node_child[(size_t)nd * 8 + k] = ch;
          if (h > 0) {
            cidx++;
            const float f0 = (k & 1) ? 0.5f : -0.5f, f1 = (k & 2) ? 0.5f : -0.5f, f2 = (k & 4) ? 0.5f : -0.5f;
            node_c[ch] = make_float4(c.x + f0 * c.w, c.y + f1 * c.w, c.z + f2 * c.w, c.w * 0.5f);
            if (h > kBigItem) {                                   // still big: stays with this block
              s_node[sp + ord] = ch; s_b[sp + ord] = start[k]; s_e[sp + ord] = start[k] + h;
              ord++;
            } else {                                              // small subtree: one wave of gb_build_kernel finishes it
              const int slot = atomicAdd(n_items_dev, 1);
              if (slot >= items_cap) { atomicExch(overflow, 1); s_fail = 1; }
              else items[slot] = GbItem{ch, 8, 0, 0, h, start[k]};
            }
          }
        }
        s_base = ord;
      }
      __threadfence_block();
      __syncthreads();
      if (s_fail) return;
      sp += s_base;
      __syncthreads();
    } else {
      if (t == 0) node_cnt[nd] = cnt;
      if (t < 8) node_child[(size_t)nd * 8 + t] = -1;
      for (int j = b + t; j < e; j += 1024) pt_leaf[lists[j]] = nd;
      __syncthreads();
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------
#define GBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

template <typename T>
static hipError_t grow(T*& p, size_t& cap, size_t need, size_t keep_n, hipStream_t st) {
  if (need <= cap) return hipSuccess;
  const size_t ncap = need + need / 2 + 1024;
  T* np = nullptr;
  GBCHK(hipMalloc(&np, ncap * sizeof(T)));
  if (p && keep_n) {
    GBCHK(hipMemcpyAsync(np, p, keep_n * sizeof(T), hipMemcpyDeviceToDevice, st));
    GBCHK(hipStreamSynchronize(st));
  }
  if (p) (void)hipFree(p);
  p = np;
  cap = ncap;
  return hipSuccess;
}

// node arrays grow together; the first node_n entries survive, node_item (per-batch scratch) is re-armed
hipError_t GBook::reserve_nodes(hipStream_t st, size_t want) {
  if (want <= node_cap) return hipSuccess;
  const size_t keep_n = (size_t)node_n;
  size_t c1 = node_cap, c2 = node_cap * 8, c3 = node_cap, c4 = 0;
  const size_t ncap = want + want / 2 + 4096;
  GBCHK(grow(node_c, c1, ncap, keep_n, st));
  GBCHK(grow(node_child, c2, ncap * 8, keep_n * 8, st));
  GBCHK(grow(node_cnt, c3, ncap, keep_n, st));
  if (node_item) { (void)hipFree(node_item); node_item = nullptr; }
  GBCHK(grow(node_item, c4, ncap, 0, st));
  node_cap = std::min(std::min(c1, c2 / 8), std::min(c3, c4));
  return hipMemsetAsync(node_item, 0xff, node_cap * sizeof(int), st);
}

void GBook::release() {
  (void)hipFree(node_c); (void)hipFree(node_child); (void)hipFree(node_cnt); (void)hipFree(node_item); (void)hipFree(pt_leaf);
  (void)hipFree(keep); (void)hipFree(assign); (void)hipFree(new_index); (void)hipFree(items); (void)hipFree(lists); (void)hipFree(tmp);
  (void)hipFree(cursor); (void)hipFree(counters); (void)hipFree(big_items); (void)hipFree(flags); (void)hipFree(rank); (void)hipFree(node_n_dev);
  *this = GBook();
}

hipError_t GBook::import_host(hipStream_t st, const std::vector<float>& c4, const std::vector<int>& child, const std::vector<int>& cnt,
                              int root_id, const float4* map_raw, int map_n, float min_half_, bool downsample_) {
  min_half = min_half_;
  downsample = downsample_;
  const size_t nn = cnt.size();
  node_n = 0;                                           // nothing to carry over
  GBCHK(reserve_nodes(st, nn + 4096));
  GBCHK(hipMemcpyAsync(node_c, c4.data(), nn * sizeof(float4), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(node_child, child.data(), nn * 8 * sizeof(int), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(node_cnt, cnt.data(), nn * sizeof(int), hipMemcpyHostToDevice, st));
  GBCHK(hipMemsetAsync(node_item, 0xff, node_cap * sizeof(int), st));
  node_n = (int)nn;
  node_n_on_dev = -1;
  finish_pending = false;
  root = root_id;
  const float* r = &c4[(size_t)root_id * 4];
  root_c[0] = r[0]; root_c[1] = r[1]; root_c[2] = r[2]; root_half = r[3];
  GBCHK(grow(pt_leaf, pt_cap, (size_t)map_n + 1024, 0, st));
  if (!node_n_dev) GBCHK(hipMalloc(&node_n_dev, sizeof(int)));
  if (!counters) GBCHK(hipMalloc(&counters, 4 * sizeof(int)));
  GBCHK(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));
  if (map_n > 0)
    hipLaunchKernelGGL(gb_leafof_kernel, dim3((map_n + 255) / 256), dim3(256), 0, st, map_raw, map_n, node_c, node_child, node_cnt, root, pt_leaf);
  GBCHK(hipStreamSynchronize(st));
  active = true;
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void gb_finite_kernel(const float4* __restrict__ pts, int n, unsigned char* __restrict__ keep,
                                                        int* __restrict__ assign) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  keep[i] = (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) ? 1 : 0;     // Octree::processPoints (:243-244)
  assign[i] = -1;
}
__global__ __launch_bounds__(256) void gb_iota_kernel(int* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// Octree::initialize (Octree.hpp:282-298) on the device: the first batch is stored completely (finite points, batch
// order); the root cube comes from its bounding box; the tree is ONE build item over all stored points.
hipError_t GBook::init(hipStream_t st, const float4* batch, int m, const float bb[6], float4* map_raw, int* kept_out,
                       float min_half_, bool downsample_, MapBuildScratch& S) {
  *kept_out = 0;
  active = false;
  if (m <= 0) return hipSuccess;
  min_half = min_half_;
  downsample = downsample_;
  node_n = 0;
  GBCHK(reserve_nodes(st, (size_t)m + 4096));
  GBCHK(grow(pt_leaf, pt_cap, (size_t)m + 1024, 0, st));
  {
    size_t c1 = batch_cap, c2 = batch_cap, c3 = batch_cap, c4 = batch_cap, c5 = batch_cap, c6 = batch_cap, c7 = batch_cap;
    GBCHK(grow(keep, c1, m, 0, st)); GBCHK(grow(assign, c2, m, 0, st)); GBCHK(grow(new_index, c3, m, 0, st));
    GBCHK(grow(items, c4, (size_t)m + m / 16 + 64, 0, st));
    GBCHK(grow(cursor, c5, m, 0, st)); GBCHK(grow(flags, c6, m, 0, st)); GBCHK(grow(rank, c7, m, 0, st));
    batch_cap = std::min(std::min(std::min(c1, c2), std::min(c3, c4)), std::min(std::min(c5, c6), c7));
    GBCHK(grow(big_items, big_cap, (size_t)m / kBigItem + 16, 0, st));
    size_t l1 = lists_cap, l2 = lists_cap;
    GBCHK(grow(lists, l1, (size_t)m + 1024, 0, st)); GBCHK(grow(tmp, l2, (size_t)m + 1024, 0, st));
    lists_cap = std::min(l1, l2);
  }
  if (!node_n_dev) GBCHK(hipMalloc(&node_n_dev, sizeof(int)));
  if (!counters) GBCHK(hipMalloc(&counters, 4 * sizeof(int)));
  const int blocks = (m + 255) / 256;
  hipLaunchKernelGGL(gb_finite_kernel, dim3(blocks), dim3(256), 0, st, batch, m, keep, assign);
  hipLaunchKernelGGL(gb_flags_kernel, dim3(blocks), dim3(256), 0, st, keep, m, flags);
  size_t scan_bytes = 0;
  GBCHK(exclusive_sum(nullptr, scan_bytes, flags, rank, m, st));
  if (scan_bytes > S.cub_tmp_bytes) {
    GBCHK(hipStreamSynchronize(st));
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    GBCHK(hipMalloc(&S.cub_tmp, scan_bytes + 1024));
    S.cub_tmp_bytes = scan_bytes + 1024;
  }
  GBCHK(exclusive_sum(S.cub_tmp, scan_bytes, flags, rank, m, st));
  hipLaunchKernelGGL(gb_append_kernel, dim3(blocks), dim3(256), 0, st, batch, keep, rank, assign, m, 0, map_raw, pt_leaf, new_index);
  uint32_t last_rank = 0, last_flag = 0;
  GBCHK(hipMemcpyAsync(&last_rank, rank + (m - 1), 4, hipMemcpyDeviceToHost, st));
  GBCHK(hipMemcpyAsync(&last_flag, flags + (m - 1), 4, hipMemcpyDeviceToHost, st));
  GBCHK(hipStreamSynchronize(st));
  const int kept = (int)(last_rank + last_flag);
  if (kept == 0) return hipSuccess;
  // root cube (Octree.hpp:287-296), float32 like the reference
  const float ex = 0.5f * (bb[3] - bb[0]), ey = 0.5f * (bb[4] - bb[1]), ez = 0.5f * (bb[5] - bb[2]);
  root_c[0] = bb[0] + ex; root_c[1] = bb[1] + ey; root_c[2] = bb[2] + ez;
  root_half = ex;
  if (ey > root_half) root_half = ey;
  if (ez > root_half) root_half = ez;
  root = 0;
  const float4 rc = make_float4(root_c[0], root_c[1], root_c[2], root_half);
  int child8[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  const int zero = 0, one = 1;
  GBCHK(hipMemcpyAsync(node_c, &rc, sizeof(float4), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(node_child, child8, sizeof(child8), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(node_cnt, &zero, sizeof(int), hipMemcpyHostToDevice, st));
  GBCHK(hipMemsetAsync(node_item, 0xff, node_cap * sizeof(int), st));
  GBCHK(hipMemcpyAsync(node_n_dev, &one, sizeof(int), hipMemcpyHostToDevice, st));
  const GbItem I0{0, 8, 0, 0, kept, 0};
  const int n_big = kept > kBigItem ? 1 : 0;
  const int cnt4[4] = {1, kept, 0, n_big};
  GBCHK(hipMemcpyAsync(items, &I0, sizeof(GbItem), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(counters, cnt4, sizeof(cnt4), hipMemcpyHostToDevice, st));
  GBCHK(hipMemcpyAsync(big_items, &zero, sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(gb_iota_kernel, dim3((kept + 255) / 256), dim3(256), 0, st, lists, kept);
  const int items_cap = m + m / 16 + 64;
  int bound = 1;
  if (n_big) {
    hipLaunchKernelGGL(gb_build_big_kernel, dim3(1), dim3(1024), 0, st, items, counters, items_cap, big_items, 1, map_raw, lists, tmp, node_c,
                       node_child, node_cnt, node_n_dev, (int)node_cap, min_half, pt_leaf, node_item, counters + 2);
    bound = items_cap;
  }
  hipLaunchKernelGGL(gb_build_kernel, dim3((bound + 3) / 4), dim3(256), 0, st, items, counters, map_raw, lists, tmp, node_c, node_child,
                     node_cnt, node_n_dev, (int)node_cap, min_half, pt_leaf, node_item, counters + 2);
  int ovf = 0;
  GBCHK(hipMemcpyAsync(&node_n, node_n_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  GBCHK(hipMemcpyAsync(&ovf, counters + 2, sizeof(int), hipMemcpyDeviceToHost, st));
  GBCHK(hipStreamSynchronize(st));
  if (ovf) return hipErrorOutOfMemory;
  GBCHK(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));      // update() expects them zero (its final mail re-zeroes them)
  node_n_on_dev = node_n;
  finish_pending = false;
  *kept_out = kept;
  active = true;
  return hipGetLastError();
}

// batch: m NaN-free device points.  Appends the kept ones to *map_raw (capacity ensured by the caller:
// map_cap >= map_n + m) and returns the number kept.
hipError_t GBook::update(hipStream_t st, const float4* batch, int m, const float bb[6], float4* map_raw, int map_n, int* kept_out,
                         MapBuildScratch& S) {
  *kept_out = 0;
  if (m <= 0) return hipSuccess;
  GBCHK(finish(st, S));
  // ---- root growth on the host mirror (Octree.hpp:354-374): max corner first, then min ----
  {
    std::vector<float> nc;   // new root nodes (c4) oldest first
    std::vector<int> nchild;
    const float corners[2][3] = {{bb[3], bb[4], bb[5]}, {bb[0], bb[1], bb[2]}};
    for (int ci = 0; ci < 2; ci++) {
      const float* b = corners[ci];
      for (;;) {
        float mm = fabsf(b[0] - root_c[0]);
        const float my = fabsf(b[1] - root_c[1]), mz = fabsf(b[2] - root_c[2]);
        if (my > mm) mm = my;
        if (mz > mm) mm = mz;
        if (!(mm > root_half)) break;
        const float ph = 2 * root_half;
        const float px = root_c[0] + (b[0] > root_c[0] ? 0.5f : -0.5f) * ph;
        const float py = root_c[1] + (b[1] > root_c[1] ? 0.5f : -0.5f) * ph;
        const float pz = root_c[2] + (b[2] > root_c[2] ? 0.5f : -0.5f) * ph;
        const int slot = (root_c[0] > px ? 1 : 0) | (root_c[1] > py ? 2 : 0) | (root_c[2] > pz ? 4 : 0);
        nc.push_back(px); nc.push_back(py); nc.push_back(pz); nc.push_back(ph);
        for (int k = 0; k < 8; k++) nchild.push_back(k == slot ? root : -1);
        root = node_n + (int)(nc.size() / 4) - 1;
        root_c[0] = px; root_c[1] = py; root_c[2] = pz; root_half = ph;
      }
    }
    const int add = (int)(nc.size() / 4);
    if (add > 0) {
      GBCHK(reserve_nodes(st, (size_t)node_n + add + 16));
      std::vector<int> ncnt(add, -1);
      GBCHK(hipMemcpyAsync(node_c + node_n, nc.data(), (size_t)add * sizeof(float4), hipMemcpyHostToDevice, st));
      GBCHK(hipMemcpyAsync(node_child + (size_t)node_n * 8, nchild.data(), (size_t)add * 8 * sizeof(int), hipMemcpyHostToDevice, st));
      GBCHK(hipMemcpyAsync(node_cnt + node_n, ncnt.data(), (size_t)add * sizeof(int), hipMemcpyHostToDevice, st));
      GBCHK(hipStreamSynchronize(st));
      node_n += add;
    }
  }
  // ---- capacity for this batch: worst case every kept point creates a short chain of nodes ----
  {
    GBCHK(reserve_nodes(st, (size_t)node_n + (size_t)m * 16 + 4096));
    GBCHK(grow(pt_leaf, pt_cap, (size_t)map_n + m + 1024, map_n, st));
    size_t c1 = batch_cap, c2 = batch_cap, c3 = batch_cap, c4 = batch_cap, c5 = batch_cap, c6 = batch_cap, c7 = batch_cap;
    GBCHK(grow(keep, c1, m, 0, st)); GBCHK(grow(assign, c2, m, 0, st)); GBCHK(grow(new_index, c3, m, 0, st));
    // items: + room for the small subtrees the big-item kernel cuts off
    GBCHK(grow(items, c4, (size_t)m + m / 16 + 64, 0, st));
    GBCHK(grow(cursor, c5, m, 0, st)); GBCHK(grow(flags, c6, m, 0, st)); GBCHK(grow(rank, c7, m, 0, st));
    batch_cap = std::min(std::min(std::min(c1, c2), std::min(c3, c4)), std::min(std::min(c5, c6), c7));
    GBCHK(grow(big_items, big_cap, (size_t)m / kBigItem + 16, 0, st));          // a batch of m points holds at most m / kBigItem big items
    size_t l1 = lists_cap, l2 = lists_cap;
    const size_t lw = (size_t)m * (kBucket + 1) + 1024;      // every group may pull in a full leaf
    GBCHK(grow(lists, l1, lw, 0, st)); GBCHK(grow(tmp, l2, lw, 0, st));
    lists_cap = std::min(l1, l2);
  }
  const int blocks = (m + 255) / 256;
  // scratch for the sort: reuse the map builder's key / value buffers
  {
    if (S.cap_pts < (size_t)m) {
      if (S.keys_in) { (void)hipFree(S.keys_in); (void)hipFree(S.keys_out); (void)hipFree(S.vals_in); (void)hipFree(S.vals_out); }
      const size_t cap = (size_t)m + m / 4 + 1024;
      GBCHK(hipMalloc(&S.keys_in, cap * 4)); GBCHK(hipMalloc(&S.keys_out, cap * 4));
      GBCHK(hipMalloc(&S.vals_in, cap * 4)); GBCHK(hipMalloc(&S.vals_out, cap * 4));
      S.cap_pts = cap;
    }
  }
  // keys = node * 9 + slot; non-finite points get the key above all of them; only the bits in use are sorted
  const uint32_t nan_key = (uint32_t)node_n * 9u + 9u;
  int key_bits = 1;
  while (key_bits < 32 && (1ull << key_bits) <= (unsigned long long)nan_key) key_bits++;
  hipLaunchKernelGGL(gb_route_kernel, dim3(blocks), dim3(256), 0, st, batch, m, node_c, node_child, node_cnt, root, nan_key, S.keys_in, S.vals_in);
  size_t tmp_bytes = 0, scan_bytes = 0;
  GBCHK(sort_pairs_u32(nullptr, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, m, 0, key_bits, st));
  GBCHK(exclusive_sum(nullptr, scan_bytes, flags, rank, m, st));
  const size_t need = std::max(tmp_bytes, scan_bytes);
  if (need > S.cub_tmp_bytes) {
    GBCHK(hipStreamSynchronize(st));
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    GBCHK(hipMalloc(&S.cub_tmp, need + 1024));
    S.cub_tmp_bytes = need + 1024;
  }
  GBCHK(sort_pairs_u32(S.cub_tmp, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, m, 0, key_bits, st));
  // (counters: zero since the last batch's final mail, see below -- unless that batch left early on an error: cleared here then)
  if (counters_dirty) GBCHK(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));
  counters_dirty = true;
  hipLaunchKernelGGL(gb_decide_kernel, dim3(blocks), dim3(256), 0, st, S.keys_out, m, node_c, node_cnt, min_half,
                     downsample ? 1 : 0, flags /* dec_keep */, reinterpret_cast<int*>(rank) /* dec_assign */, items, counters, node_item, big_items, nan_key,
                     cursor);
  // keep flags as words for the rank scan: into the sort's (now free) key input buffer
  hipLaunchKernelGGL(gb_apply_kernel, dim3(blocks), dim3(256), 0, st, S.keys_out, S.vals_out, m, flags, reinterpret_cast<const int*>(rank),
                     keep, assign, S.keys_in);
  GBCHK(exclusive_sum(S.cub_tmp, scan_bytes, S.keys_in, rank, m, st));
  hipLaunchKernelGGL(gb_append_kernel, dim3(blocks), dim3(256), 0, st, batch, keep, rank, assign, m, map_n, map_raw, pt_leaf, new_index);
  // counts back through the mail words (one small kernel + one wait, no staged 4-byte copies)
  {
    const MailPart parts[3] = {{counters, 4, MAIL_BOOK}, {rank + (m - 1), 1, MAIL_BOOK + 4}, {S.keys_in + (m - 1), 1, MAIL_BOOK + 5}};
    GBCHK(mail_words(st, S, parts, 3));
  }
  GBCHK(mail_wait(st, S));
  int h_cnt[4];
  for (int k = 0; k < 4; k++) h_cnt[k] = (int)S.mail_host[MAIL_BOOK + k];
  const uint32_t last_rank = S.mail_host[MAIL_BOOK + 4], last_flag = S.mail_host[MAIL_BOOK + 5];
  const int n_items = h_cnt[0];
  const int kept = (int)(last_rank + last_flag);
  if ((size_t)h_cnt[1] > lists_cap) return hipErrorOutOfMemory;
  if (n_items > 0) {
    {
      const int old_blocks = map_n > 0 ? (map_n + 255) / 256 : 0;
      hipLaunchKernelGGL(gb_gather_kernel, dim3(old_blocks + blocks), dim3(256), 0, st, pt_leaf, map_n, node_item, items, cursor, lists,
                         old_blocks, S.vals_out, assign, new_index, m);
    }
    if (node_n != node_n_on_dev) {                                  // (root growth on the host, or first use)
      GBCHK(hipMemcpyAsync(node_n_dev, &node_n, sizeof(int), hipMemcpyHostToDevice, st));
      node_n_on_dev = node_n;
    }
    const int n_big = h_cnt[3];                      // (counters[2], the overflow mark, is zero like the others)
    const int items_cap = m + m / 16 + 64;
    int bound = n_items;
    if (n_big > 0) {
      hipLaunchKernelGGL(gb_build_big_kernel, dim3(n_big), dim3(1024), 0, st, items, counters, items_cap, big_items, n_big, map_raw, lists,
                         tmp, node_c, node_child, node_cnt, node_n_dev, (int)node_cap, min_half, pt_leaf, node_item, counters + 2);
      bound = items_cap;                               // the final item count stays on the device (counters[0])
    }
    hipLaunchKernelGGL(gb_build_kernel, dim3((bound + 3) / 4), dim3(256), 0, st, items, counters, map_raw, lists, tmp, node_c, node_child,
                       node_cnt, node_n_dev, (int)node_cap, min_half, pt_leaf, node_item, counters + 2);
    // the node count and the overflow mark travel by mail; the caller's next wait on the stream delivers them (finish)
    // ... and the same small kernel zeroes the four counters for the next batch
    const MailPart parts[2] = {{node_n_dev, 1, MAIL_BOOK_END}, {counters + 2, 1, MAIL_BOOK_END + 1}};
    GBCHK(mail_words(st, S, parts, 2, false, counters, 4));
    counters_dirty = false;
    finish_pending = true;
  } else {
    GBCHK(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));       // nothing to build: [1] / [3] may still be set
    counters_dirty = false;
  }
  *kept_out = kept;
  last_items = n_items;
  return hipGetLastError();
}

// Second half of update(): waits for the stream and takes over the node count the subtree builds left on the device.
hipError_t GBook::finish(hipStream_t st, MapBuildScratch& S) {
  if (!finish_pending) return hipSuccess;
  finish_pending = false;
  GBCHK(hipStreamSynchronize(st));
  node_n = (int)S.mail_host[MAIL_BOOK_END];
  node_n_on_dev = node_n;
  if (S.mail_host[MAIL_BOOK_END + 1]) return hipErrorOutOfMemory;
  return hipSuccess;
}

}  // namespace flimo

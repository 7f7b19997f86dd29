// fast_limo_amd/csrc/hip/flimo_gbook.h -- device-resident octree bookkeeping of the map's insert rule
// (see flimo_gbook.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "flimo_kernels.h"

namespace flimo {

struct GbItem;

struct GBook {
  bool active = false;
  // octree nodes (SoA)
  float4* node_c = nullptr;      // centre xyz + half edge
  int* node_child = nullptr;     // [8 * cap], -1 = absent
  int* node_cnt = nullptr;       // leaf: number of points (>= 0); internal: -1
  int* node_item = nullptr;      // scratch: build item of a splitting leaf, -1 otherwise
  size_t node_cap = 0;
  int node_n = 0;
  int* node_n_dev = nullptr;
  int node_n_on_dev = -1;        // the value *node_n_dev is known to hold (-1: unknown, upload before use)
  bool finish_pending = false;   // update() left its node count in the mail words: finish() takes it over
  int root = -1;
  float root_c[3] = {0, 0, 0}, root_half = 0.f;   // host mirror of the root cube
  float min_half = 0.2f;
  bool downsample = true;
  // per stored point: its leaf
  int* pt_leaf = nullptr;
  size_t pt_cap = 0;
  // per batch scratch
  unsigned char* keep = nullptr;
  int* assign = nullptr;
  int* new_index = nullptr;
  GbItem* items = nullptr;
  int* cursor = nullptr;
  uint32_t* flags = nullptr;
  uint32_t* rank = nullptr;
  size_t batch_cap = 0;
  int* lists = nullptr;
  int* tmp = nullptr;
  size_t lists_cap = 0;
  int* big_items = nullptr;      // item ids whose subtree build gets a whole block (at most batch / 2048 of them)
  size_t big_cap = 0;
  int* counters = nullptr;       // [0] items, [1] list cursor, [2] overflow
  bool counters_dirty = false;   // an update() that did not reach its trailing mail (an error return) left them non-zero: the next one clears them
  int last_items = 0;

  // take over a tree built on the host for the first batch
  hipError_t import_host(hipStream_t st, const std::vector<float>& c4, const std::vector<int>& child, const std::vector<int>& cnt,
                         int root_id, const float4* map_raw, int map_n, float min_half, bool downsample);
  // Octree::initialize for the first batch (m device points, NaNs ignored): stores every finite point in batch order
  hipError_t init(hipStream_t st, const float4* batch, int m, const float bb[6], float4* map_raw, int* kept_out, float min_half,
                  bool downsample, MapBuildScratch& S);
  // Octree::update for a batch of m device points (NaN points are ignored); bb = bounding box of the
  // finite points of the batch; appends the kept points to map_raw[map_n ...] and returns their count
  // (update() ends with its subtree builds still queued: finish() -- after any later wait on the stream, or with its own --
  //  completes it; update() itself calls it first when the caller did not)
  hipError_t finish(hipStream_t st, MapBuildScratch& S);
  hipError_t update(hipStream_t st, const float4* batch, int m, const float bb[6], float4* map_raw, int map_n, int* kept_out,
                    MapBuildScratch& S);
  void release();
  hipError_t reserve_nodes(hipStream_t st, size_t want);
};

// flimo_map.hip: bounding box of the finite points (host result); *any = false when there is none
hipError_t batch_bbox(hipStream_t st, const float4* pts, size_t n, MapBuildScratch& S, float bb[6], bool* any);

}  // namespace flimo

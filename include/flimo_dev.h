/* include/flimo_dev.h -- developer instrumentation of libflimo_hip.so: timing, statistics and the A/B switches bench.py and the
 * tests use.  Nothing here is part of the drop-in boundary (include/flimo_c.h): no entry replaces a reference interface, none
 * changes a result.  Same library, same context handle. */
#ifndef FLIMO_DEV_H
#define FLIMO_DEV_H
#include "flimo_c.h"
#ifdef __cplusplus
extern "C" {
#endif

/* GPU time [ms] of the stages of the last flimo_match_reduce, from HIP events on the ctx stream:
 * k-NN fast path, ring widening of the worklist, fit + reductions.  flimo_set_timing level:
 * 0 off, 1 on: every dispatch of a pass carries its own begin / end events (no extra packets on the stream). */
int flimo_set_timing(flimo_ctx* ctx, int level);
/* level 1 only: time every `every`-th pass (default 1 = all); the totals count the timed passes only. */
int flimo_set_timing_stride(flimo_ctx* ctx, int every);
/* on != 0: the events of flimo_match_reduce's timed passes are read when the totals are asked for (flimo_timing_totals /
 * flimo_timing_split; at the latest after 64 timed passes) instead of right behind each pass.  Reading costs the host tens of
 * microseconds per pass; a series in which every pass is timed then leaves the GPU idle half of the time, and on some boxes its
 * clocks follow (the same kernels read 10-15 % long).  flimo_last_kernel_ms reports the last pass READ, not the last run. */
int flimo_set_timing_deferred(flimo_ctx* ctx, int on);
/* number of flimo_match_reduce passes launched on this context so far */
unsigned long long flimo_pass_count(const flimo_ctx* ctx);
/* ... of which ran as ONE launch (k-NN + in-kernel widening + fit + reduction); the others used separate dispatches (first pass
 * of a scan with a poor prior, records / caps / debug, non-default lanes per query, gates wider than 3 rings) */
unsigned long long flimo_fused_pass_count(const flimo_ctx* ctx);
/* exact float32 distance ties (Objects/Octree.hpp:72-87,558-599: the reference keeps the candidate its recursion meets first):
 * out[0] = passes whose rows were rebuilt after settling ties in a launch of their own (records / caps / debug path), out[1] =
 * queries settled so far -- there and inside the reducing launches of the per-pass fast paths, which settle a tied query where
 * they build its row */
int flimo_tie_stats(flimo_ctx* ctx, unsigned long long out[2]);   /* (enters the context and drains its stream: owner's thread only) */
/* second level over crowded regions (cells holding > 64 points get a grid with a quarter of the cell edge and a pre-pass):
 * out[0] = active now, out[1] = map points copied into it, out[2] = times it was (re)built, out[3] = passes that ran the pre-pass */
int flimo_fine_stats(const flimo_ctx* ctx, unsigned long long out[4]);
int flimo_last_kernel_ms(const flimo_ctx* ctx, float* knn_ms, float* widen_ms, float* fit_ms);
/* sums over every pass since the last reset (timing must be on): per-stage GPU ms, passes, k-NN queries */
int flimo_timing_totals(flimo_ctx* ctx, double* knn_ms, double* widen_ms, double* fit_ms, long long* passes,
                        long long* queries, int reset);
/* level-1 totals by kind of timed pass since the last reset: out[0] ms of the one-launch passes (k-NN + in-kernel widening + fit +
 * reduction), out[1] their count; out[2..4] ms of the k-NN, widening and fit dispatches of the passes that ran them separately,
 * out[5] their count */
int flimo_timing_split(flimo_ctx* ctx, double out[6], int reset);
/* A/B switches of the pass layout (each: 1 on, 0 off, negative = leave): `tail` finishes pending queries inside the k-NN launch,
 * `fuse` runs the whole pass as one launch.  Both on by default; the benchmark switches `fuse` off for a short series to time the
 * k-NN stage (fast path + widening) on its own. */
int flimo_set_path_switches(flimo_ctx* ctx, int tail, int fuse);
/* number of scan points of the last pass that needed more than the 3x3x3 cell block */
int flimo_last_widen_count(const flimo_ctx* ctx);
/* the same count as published by the pass itself with its result (fast path), -1 when the last pass took a path that does
 * not report it (records / caps / timing level 2) */
int flimo_last_stragglers(const flimo_ctx* ctx);
/* ... by the pass's position within its scan, as last reported: out[0] first pass .. out[3] fourth and later (what decides the
 * layout of the pass at the same position of the next scan) */
int flimo_stragglers_by_pass(const flimo_ctx* ctx, int out[4]);
/* mean number of candidate map points examined per query in the last pass */
double flimo_last_candidates_per_query(const flimo_ctx* ctx);

/* The cell-sorted copy of the map is maintained incrementally: points appended by an insert go into their rows in place, a map
 * that outgrows its grid has the grid grown around it; the whole map is sorted only at the first layout and when a tile shape,
 * the point array or the tile pool no longer does.  Debug check: sorts the whole map again with the current geometry and counts
 * where the maintained index differs (by meaning: a row's points in order, a row's position at every column) -- 0 by construction.
 * stats = {inserts in place, full layouts} so far. */
int flimo_map_grid_selfcheck(flimo_ctx* ctx, uint64_t* mismatches, uint64_t stats[2]);

/* out[0] = GPU ms of the algebra launches timed so far (timing level 1), out[1] = their number,
 * out[2] = chains run, out[3] = chains that came back before the final iteration, out[4] = chains declined */
int flimo_chain_stats(flimo_ctx* ctx, double out[5], int reset);
/* bytes held by the map: out[0] = the stored points (16 B each), out[1] = its index (the tiles that exist, directory, row
 * starts, the rows' room and first positions, escape pool as allocated), out[2] = the second level over crowded regions (points +
 * index), 0 when none is active; out[3] = tiles of the index that exist (with the shared empty one), out[4] = inserts that found
 * the tile pool too small and had the index laid out afresh, out[5] = the cell-sorted array AS ALLOCATED (three times the raw
 * buffer's capacity: its rows keep room behind their last point, so that an insert touches only the rows it adds to) */
int flimo_map_index_bytes(const flimo_ctx* ctx, uint64_t out[6]);
/* pipelined host loop (flimo_set_pass_pipeline): {passes that found their launch waiting, queued passes nobody asked for, passes
 * whose waiting launch was found too old to publish to (told to leave, launched the usual way), passes whose launch had left as a
 * whole before the publish reached it (launched again)} */
int flimo_pass_pipeline_stats(const flimo_ctx* ctx, unsigned long long out[4]);

/* ---- host-side evaluation, no GPU (round 6: out of the boundary header): what the host C++ mirror's Plane / calculate_H objects and
 * the CPU tests call ----
 * Replays the map's insert rule (Octree::initialize / update, Objects/Octree.hpp:282-432) over a
 * sequence of batches of packed NaN-free points: keep[i] = 1 if point i is stored.  Host only. */
/* Host-side evaluation of the plane routines of the fit kernel (same source, compiled for the host): Plane::estimate_plane
 * (Objects/Plane.cpp:80-105) for exactly 5 points (xyz packed) and Plane::plane_eval (:107-114).  They back the
 * fast_limo::Plane object of the host C++ mirror; the registration path never calls them. */
void flimo_plane_fit5_host(const float xyz[15], float n_out[4]);
int  flimo_plane_eval5_host(const float n[4], const float xyz[15], float threshold);

/* Localizer::calculate_H (Localizer.cpp:537-577) for M given matches on the host, with the fit kernel's own row routine:
 * p_global [M][3], n [M][4] (plane.get_normal()), dist [M] (Match::dist); H [M][12] row-major, h [M] = -dist.  Backs the
 * Localizer::calculate_H method of the host C++ mirror; the registration path computes the same rows on the GPU. */
int flimo_calculate_H_host(const double x26[26], const float* p_global, const float* n, const float* dist, size_t M,
                           int estimate_extrinsics, double* H, double* h);

int flimo_insert_rule_replay(float min_extent, int downsample, const float* xyz, const size_t* batch_sizes,
                             size_t n_batches, unsigned char* keep, size_t* stored);

#ifdef __cplusplus
}
#endif
#endif /* FLIMO_DEV_H */

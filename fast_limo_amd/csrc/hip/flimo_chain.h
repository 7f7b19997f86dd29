// fast_limo_amd/csrc/hip/flimo_chain.h
// The whole iterated update of a scan enqueued at once ("chain"): esekf::update_iterated_dyn_share_modified
// (IKFoM_toolkit/esekfom/esekfom.hpp:1620-1823) as  pass_1 -> algebra -> pass_2 -> algebra -> ...  on ONE HIP stream, no host
// round trip between the passes.  A pass is the measurement plug-in (h_share_model, use-ikfom.cpp:10-31: the k-NN / fit /
// reduction launches of flimo_kernels.hip); the algebra is the rest of one outer iteration (:1652-1760, and :1764-1820 on the
// last one) run by a one-block kernel (flimo_ieskf.hip) that leaves the next pass's float32 pose constants in device memory.
// Every launch of a later pass reads its pose from there and leaves at once when the chain has ended (converged, or handed back
// to the host filter: M < 23, exact distance ties, a degenerate H^T H).
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"

namespace flimo {

constexpr int CH_MAX_PASSES = 12;      // MAX_NUM_ITERS + 1 <= this, else the host loop runs the update

// What the pass kernels read (first bytes of the device filter state)
struct ChainHead {
  PoseMats pose;           // float32 constants of the NEXT pass (State casts, get_RT / get_RT_inv / get_extr_RT_inv, calculate_H's rotations)
  float prev_RT[16];       // body -> world of the pass just completed: the next pass's pruning bound is relative to it
  int status;              // 0: the chain goes on; 1: finished (x, P final); 2: handed back to the host filter
  int pad[3];
};

// The prior of a scan's update, written by the host into mapped memory before the chain is enqueued (read once, by the first
// algebra kernel)
struct ChainPrior {
  double x[26];            // flat state_ikfom (use-ikfom.hpp:12-21)
  double P[529];           // 23 x 23 row-major
  double limit[23];        // convergence limits (esekfom.hpp:1757-1763)
  double R, D;             // measurement noise, degeneracy threshold
  int max_iter;            // MAX_NUM_ITERS: passes it = -1 .. max_iter - 1
  int pad;
  float RT0[16];           // body -> world the first pass ran with
};

// Results: 16-byte granules {value, tag} in mapped host memory (data and "ready" travel together, like a pass's sums)
constexpr int CH_STATUS = 0, CH_BAIL = 1, CH_PASSES = 2, CH_IT = 3, CH_T = 4;
constexpr int CH_X = 5;                                   // x[26] after the last completed algebra
constexpr int CH_XMEAS = CH_X + 26;                       // x[26] the last executed pass measured at
constexpr int CH_PASSINFO = CH_XMEAS + 26;                // per pass: M, stragglers, ties
constexpr int CH_P = CH_PASSINFO + 3 * CH_MAX_PASSES;     // P[529] (status 1)
constexpr int CH_RES = CH_P + 529;
// optional per-pass log (tests): HTH[144], HTh[12], dx[23], x_after[26]
constexpr int CH_LOGN = 144 + 12 + 23 + 26;

struct ChainState;                                        // flimo_ieskf.hip
size_t chain_state_size();
// One outer iteration's algebra after a pass: `gran` = the pass's granules in DEVICE memory ([FIT_GROUPS][FIT_LIVE_PAD] x {sum, seq});
// prior != nullptr: first pass of the chain (the state is loaded from it).  res / log: mapped host memory.
void launch_ieskf(hipStream_t st, ChainState* S, const void* gran, unsigned long long seq, const ChainPrior* prior, void* res, void* log,
                  unsigned long long tag, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

}  // namespace flimo

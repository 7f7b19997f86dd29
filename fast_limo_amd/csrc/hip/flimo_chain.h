// fast_limo_amd/csrc/hip/flimo_chain.h
// The whole iterated update of a scan enqueued at once ("chain"): esekf::update_iterated_dyn_share_modified
// (IKFoM_toolkit/esekfom/esekfom.hpp:1620-1823) as  pass_1 -> algebra -> pass_2 -> algebra -> ...  on ONE HIP stream, no host round
// trip between the passes.  A pass is the measurement plug-in (h_share_model, use-ikfom.cpp:10-31: the k-NN / fit / reduction launches
// of flimo_kernels.hip).  The rest of one outer iteration (:1652-1760) is split where the loop allows it (flimo_ieskf.h):
//   * one extra workgroup of the pass's reducing launch computes the half that does not depend on the measurement (x boxminus x_prop,
//     the covariance through the manifold blocks, :1652-1697) while the other workgroups search and fit;
//   * a one-workgroup launch behind the pass (ieskf_kernel, flimo_ieskf.hip) goes on from the pass's 91 sums to the gain, the step,
//     boxplus and the convergence test (:1722-1764) and leaves the next pass's float32 pose constants in device memory.  (Rounds 4
//     also ran that half inside the reducing launch and in one resident workgroup beside the chain; neither beat the launch of its
//     own by more than 1 %, profiles/r04/withdrawn_experiments.md, and both were removed in round 5.)
// Every launch of a later pass reads its pose from there and leaves at once when the chain has ended.  The chain ends by handing the
// loop back to the host filter: at the iteration whose covariance update is due (:1764-1820: the host runs that one iteration from
// the sums the device hands over -- no further pass), or earlier at a branch the device does not run (M < 23, a degenerate H^T H,
// a failure).
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"

namespace flimo {

constexpr int CH_MAX_PASSES = 12;      // MAX_NUM_ITERS + 1 <= this, else the host loop runs the update

// What the pass kernels read (first bytes of the device filter state)
struct ChainHead {
  PoseMats pose;           // float32 constants of the NEXT pass (State casts, get_RT / get_RT_inv / get_extr_RT_inv, calculate_H's rotations)
  float prev_RT[16];       // body -> world of the pass just completed: the next pass's pruning bound is relative to it
  int status;              // 0: the chain goes on; 2: handed back to the host filter
  unsigned int epoch;      // pipelined host loop: number (low 32 bits) of the pass these constants are FOR -- written last, behind them
  unsigned int decision;   // a launch that waits for `epoch`: workgroup 0's verdict when the wait runs out (chain_enter, flimo_kernels.hip); written by the device only
  int pad;
};
static_assert(sizeof(ChainHead) % 4 == 0 && sizeof(ChainHead) / 4 <= 128, "a pass workgroup reads the head with one load per thread");

// The prior of a scan's update, written by the host into mapped memory before the chain is enqueued (copied to the device filter
// by the first pass's extra workgroup, beside the pass).  The measurement-independent half of iteration -1 (x == x_prop: no
// transcendental function is evaluated) comes with it.
struct ChainPrior {
  double x[26];            // flat state_ikfom (use-ikfom.hpp:12-21)
  double P[529];           // 23 x 23 row-major
  double limit[23];        // convergence limits (esekfom.hpp:1757-1763)
  double R, D;             // measurement noise, degeneracy threshold
  int max_iter;            // MAX_NUM_ITERS: iterations it = -1 .. max_iter - 1
  int pad;
  double dxn[23];          // dx_new of iteration -1
  double AG[276];          // of iteration -1: A11^-1 (144) and G2 = A21 A11^-1 (132) of A = (P_ through the blocks) / R
};

// Filter state of one scan's update in device memory.  flat state x26: pos3 rot4(xyzw) offR4 offT3 vel3 bg3 ba3 grav3.
struct ChainState {
  ChainHead head;          // what the pass kernels read
  double x[26];
  double x_prop[26];
  double P_prop[529];
  double limit[23];
  double R, D;
  int max_iter;            // MAX_NUM_ITERS: iterations it = -1 .. max_iter - 1
  int it;                  // next iteration: it = -1 + iterations done
  int t;                   // iterations that met the limits so far
  int passes;              // iterations completed by the device
  double info[3 * CH_MAX_PASSES];   // per pass: M, stragglers, ties
  double pre_dxn[23];      // the measurement-independent half of the CURRENT iteration (extra workgroup of its pass):
  double pre_AG[276];      // dx_new; A11^-1 (144) and G2 = A21 A11^-1 (132) of A = P_ / R
};

// Arguments of the algebra inside a pass's reducing launch (S == nullptr: a host-driven pass)
struct ChainCtl {
  ChainState* S;
  const ChainPrior* prior;     // first pass of the chain: mapped host memory; later passes: nullptr
  double2* gran;               // DEVICE copy of the granule slots: [FIT_GROUPS][FIT_LIVE_PAD] x {sum, pass number}
  double2* res;                // mapped host memory: CH_RES result granules {value, tag}
  double2* log;                // mapped host memory: per-pass log (or nullptr)
  unsigned long long tag;      // tag of this scan's chain
  unsigned int end_code;       // a pass queued ahead of its pose (pipelined host loop): head.epoch takes this value (top bit set) when it is told to leave
};

// Results: 16-byte granules {value, tag} in mapped host memory (data and "ready" travel together, like a pass's sums)
constexpr int CH_STATUS = 0, CH_BAIL = 1, CH_PASSES = 2, CH_IT = 3, CH_T = 4;
constexpr int CH_X = 5;                                   // x[26]: the state the handed-back iteration measured at
constexpr int CH_PASSINFO = CH_X + 26;                    // per pass: M, stragglers, ties
constexpr int CH_SUMS = CH_PASSINFO + 3 * CH_MAX_PASSES;  // the handed-back iteration's 91 sums (upper triangle of H^T H, H^T h, M)
constexpr int CH_RES = CH_SUMS + 91;
// reasons (CH_BAIL); TIES no longer occurs (exact ties are settled inside the reducing launch)
constexpr int CH_R_FEW = 1, CH_R_TIES = 2, CH_R_DEGENERATE = 3, CH_R_FAILED = 4, CH_R_FINAL = 5;
// optional per-pass log (tests): HTH[144], HTh[12], dx[23], x_after[26]
constexpr int CH_LOGN = 144 + 12 + 23 + 26;

size_t chain_state_size();
// the algebra as a launch of its own behind the pass.  gran: the pass's granules in device memory.
void launch_ieskf(hipStream_t st, const ChainCtl& ch, unsigned long long seq, const float* used_RT_host_or_null,
                  hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void launch_ieskf_extra(hipStream_t st, const ChainCtl& ch, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);   // developer tool
// head.epoch of the pass numbered seq: never zero (zero means "do not wait"), top bit clear (set: an end code)
__host__ __device__ inline unsigned int ch_epoch_of(unsigned long long seq) { return (unsigned int)(seq & 0x3fffffffull) | 0x40000000u; }
constexpr int CH_POLL_MS = 50;           // a pass's workgroups give up waiting for their constants after this long (status FAILED)

}  // namespace flimo

// fast_limo_amd/csrc/hip/flimo_ieskf.hip  -- gfx950 device code.
//
// A/B form of the chained update (flimo_chain.h): the measurement-dependent half of an iteration (flimo_ieskf.h: ik_final_stage) as
// a one-workgroup launch of its own behind the pass, instead of inside the pass's reducing launch (FLIMO_CHAIN_INLINE=0; also what
// tools/ieskf_bench.hip times and checks against the host filter).  Same routine, same results; it costs a dispatch boundary, the
// launch itself and one more round trip for the sums per iteration.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "flimo_types.h"
#include "flimo_kernels.h"
#include "flimo_chain.h"
#include "flimo_ieskf.h"

#pragma clang fp contract(off)

namespace flimo {

static_assert(IK_LIVE == FIT_LIVE && IK_LIVE_PAD == FIT_LIVE_PAD && IK_GROUPS == FIT_GROUPS, "flimo_ieskf.h mirrors the reduction's constants");

size_t chain_state_size() { return sizeof(ChainState); }

struct RT16 { float v[16]; };

__global__ __launch_bounds__(256) void ieskf_kernel(ChainCtl ch, unsigned long long seq, RT16 rt0, int use_rt0) {
  __shared__ double lds[IKL_END];
  __shared__ int s_i[32];
  if (!ch.prior && ch.S->head.status != 0) return;       // the chain ended in an earlier iteration
  ik_final_stage<true>(ch, seq, use_rt0 ? rt0.v : ch.S->head.pose.RT, lds, s_i, (int)threadIdx.x);
}

// The resident form (flimo_chain.h): one workgroup runs the algebra of every iteration of a chain.  It is launched before the chain's
// passes on a stream of its own, so it holds its place on the GPU while they run, and learns that pass i has delivered -- eight
// group sums and the extra workgroup's half, all performed -- from the arrival count at ticket3.  No dispatch boundary and no
// kernel start between a pass and its algebra, none between the algebra and the next pass either: that pass's workgroups are
// already placed and poll head.epoch (flimo_kernels.hip: chain_enter).
__global__ __launch_bounds__(256) void ieskf_resident_kernel(ChainCtl ch, unsigned long long seq0, int n_pass, RT16 rt0, unsigned long long wait_ticks) {
  __shared__ double lds[IKL_END];
  __shared__ int s_i[32];
  __shared__ float s_rt[16];
  __shared__ int s_go;
  const int tid = (int)threadIdx.x;
  if (tid < 16) s_rt[tid] = rt0.v[tid];
  __syncthreads();
  for (int i = 0; i < n_pass; i++) {
    const unsigned long long seq = seq0 + 1ull + (unsigned long long)i;
    if (tid == 0) {
      const unsigned int want = (unsigned int)(FIT_GROUPS + 1) * (unsigned int)(i + 1);
      const unsigned long long t0 = wall_clock64();
      int go = 1;
      while (__hip_atomic_load(ch.ticket3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() - t0 > wait_ticks) { go = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      s_go = go;
      ch.S->stamps[ch_epoch_of(seq + 1ull) % CH_MAX_PASSES][2] = wall_clock64();
    }
    __syncthreads();
    if (!s_go) {
      // the pass never arrived (a launch that failed, a hung kernel): the loop goes back to the host with reason FAILED
      if (tid < 26) lds[IKL_XC + tid] = ch.S->x[tid];
      if (tid >= 32 && tid < 32 + IK_LIVE_PAD) lds[IKL_LIVE + tid - 32] = 0.0;
      __syncthreads();
      ik_hand_back(ch, lds, CH_R_FAILED, 0, 0, 0, 0, 0, 0, tid);
      break;
    }
    const bool on = ik_final_stage<true>(ch, seq, s_rt, lds, s_i, tid);
    if (!on) break;
    if (tid == 0) ch.S->stamps[ch_epoch_of(seq + 1ull) % CH_MAX_PASSES][3] = wall_clock64();
    // the pose the NEXT pass runs with (its pruning bound's reference once it has delivered): as stored in the head
    __syncthreads();
    if (tid < 16) s_rt[tid] = __hip_atomic_load(&ch.S->head.pose.RT[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  // (ik_hand_back re-armed ticket3 before it published the result)
}

// A developer's way to run the extra workgroup's half on its own (tools/ieskf_bench.hip)
__global__ __launch_bounds__(256) void ieskf_extra_kernel(ChainCtl ch) {
  __shared__ double lds[IKL_END];
  ik_extra_block(ch, lds, (int)threadIdx.x);
}

// (first pass: the pose it ran with is the host's, handed over as an argument; later passes: the filter's head)
void launch_ieskf(hipStream_t st, const ChainCtl& ch, unsigned long long seq, const float* used_RT_host_or_null, hipEvent_t e0, hipEvent_t e1) {
  RT16 rt{};
  if (used_RT_host_or_null) for (int i = 0; i < 16; i++) rt.v[i] = used_RT_host_or_null[i];
  hipExtLaunchKernelGGL(ieskf_kernel, dim3(1), dim3(256), 0, st, e0, e1, 0, ch, seq, rt, used_RT_host_or_null ? 1 : 0);
}
void launch_ieskf_resident(hipStream_t st, const ChainCtl& ch, unsigned long long seq0, int n_pass, const float* first_RT_host, int wait_ms) {
  RT16 rt{};
  for (int i = 0; i < 16; i++) rt.v[i] = first_RT_host[i];
  const unsigned long long ticks = (unsigned long long)(wait_ms > 0 ? wait_ms : 2000) * 100000ull;      // the wall clock counts at 100 MHz
  hipLaunchKernelGGL(ieskf_resident_kernel, dim3(1), dim3(256), 0, st, ch, seq0, n_pass, rt, ticks);
}
void launch_ieskf_extra(hipStream_t st, const ChainCtl& ch, hipEvent_t e0, hipEvent_t e1) {
  hipExtLaunchKernelGGL(ieskf_extra_kernel, dim3(1), dim3(256), 0, st, e0, e1, 0, ch);
}

}  // namespace flimo

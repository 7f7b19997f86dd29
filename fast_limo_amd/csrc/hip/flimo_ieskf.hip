// fast_limo_amd/csrc/hip/flimo_ieskf.hip  -- gfx950 device code.
//
// The chained update (flimo_chain.h): the measurement-dependent half of an iteration (flimo_ieskf.h: ik_final_stage) as a
// one-workgroup launch behind the pass (also what tools/ieskf_bench.hip times and checks against the host filter).
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
void launch_ieskf_extra(hipStream_t st, const ChainCtl& ch, hipEvent_t e0, hipEvent_t e1) {
  hipExtLaunchKernelGGL(ieskf_extra_kernel, dim3(1), dim3(256), 0, st, e0, e1, 0, ch);
}

}  // namespace flimo

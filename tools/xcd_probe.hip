// Developer probe (GPU box): can the pass's grid reduction keep a group's hand-offs inside ONE XCD's L2?
//   fit_reduce_publish (flimo_kernels.hip) groups the launch's workgroups as b & 7 and moves partial sums between them at AGENT scope
//   (write-through stores, an acknowledged store before the ticket, loads past the L2).  Workgroup b of a dispatch is placed on XCD
//   b % 8 (each XCD's dispatcher takes its share of the packet), so a group's workgroups share an L2 -- if that rule holds, the
//   same hand-offs can run at the L2's scope (sc0 only: stores acknowledged by the L2, loads that miss only the per-CU cache).
// This probe  (1) records XCC_ID and CU of every workgroup of 512-workgroup launches and checks the rule,
//             (2) runs the hand-off (91 partial sums per workgroup -> ticket -> the group's last workgroup adds 64 partials) in both
//                 scopes, timed inside the kernel (100 MHz wall clock), and checks every sum (a stale read shows as a mismatch).
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int LIVE = 91, PAD = 96, GROUPS = 8;

template <bool LOCAL>
__device__ __forceinline__ void st_f64(double* p, double v) {
  if (LOCAL) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
template <bool LOCAL>
__device__ __forceinline__ double ld_f64(const double* p) {
  return LOCAL ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stamps[b][0..5]: start, partials acknowledged, ticket back, gather done, published (0 when not the last), xcc | cu << 8
template <bool LOCAL>
__global__ __launch_bounds__(256) void handoff(double* __restrict__ partials, unsigned int* __restrict__ ticket, double* __restrict__ out,
                                               unsigned long long* __restrict__ stamps, unsigned int iter, int stagger,
                                               unsigned int* __restrict__ bad) {
  __shared__ unsigned int s_last;
  __shared__ double s_a[2][128];
  const int b = blockIdx.x, t = threadIdx.x;
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  const int group = b & (GROUPS - 1), nb_g = gridDim.x / GROUPS;
  if (stagger) {                               // workgroups of a real pass arrive spread over microseconds
    const unsigned long long t0 = wall_clock64();
    const unsigned long long d = (unsigned long long)((b * 2654435761u) >> 24) * (unsigned long long)stagger / 256ull;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long ts0 = wall_clock64();
  if (t < LIVE) st_f64<LOCAL>(&partials[(size_t)b * PAD + t], (double)(b + 1) * (double)(t + 1) + (double)iter);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long ts1 = wall_clock64();
  if (t == 0) {
    const unsigned int old = __hip_atomic_fetch_add(ticket + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == (unsigned int)nb_g - 1u ? 1u : 0u;
  }
  __syncthreads();
  const unsigned long long ts2 = wall_clock64();
  unsigned long long ts3 = 0, ts4 = 0;
  if (s_last) {
    const int c = t & 127, part = t >> 7;
    const int per = (nb_g + 1) >> 1, k0 = part * per, k1 = min(nb_g, k0 + per);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (c < LIVE) {
      const double* base = partials + (size_t)group * PAD + c;
      const size_t stride = (size_t)GROUPS * PAD;
      int k = k0;
      for (; k + 31 < k1; k += 32) {
        double w[32];
#pragma unroll
        for (int u = 0; u < 32; u++) w[u] = ld_f64<LOCAL>(base + (size_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 32; u += 4) { s0 += w[u]; s1 += w[u + 1]; s2 += w[u + 2]; s3 += w[u + 3]; }
      }
      for (; k < k1; k++) s0 += ld_f64<LOCAL>(base + (size_t)k * stride);
    }
    s_a[part][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    ts3 = wall_clock64();
    if (t < LIVE) {
      const double sum = s_a[0][t] + s_a[1][t];
      // expected: sum over the group's blocks of (blk + 1)(t + 1) + iter
      double want = 0.0;
      for (int k = 0; k < nb_g; k++) want += (double)(k * GROUPS + group + 1) * (double)(t + 1) + (double)iter;
      if (sum != want) atomicAdd(bad, 1u);
      __hip_atomic_store(&out[(size_t)group * PAD + t], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t == 0) __hip_atomic_store(ticket + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts4 = wall_clock64();
  }
  if (t == 0) {
    unsigned long long* s = stamps + (size_t)b * 6;
    s[0] = ts0; s[1] = ts1; s[2] = ts2; s[3] = ts3; s[4] = ts4; s[5] = (unsigned long long)xcc | ((unsigned long long)__smid() << 8);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  const int NB = 512;
  double *partials, *out;
  unsigned int *ticket, *bad;
  unsigned long long* stamps;
  CHECK(hipMalloc(&partials, (size_t)NB * PAD * sizeof(double)));
  CHECK(hipMemset(partials, 0, (size_t)NB * PAD * sizeof(double)));
  CHECK(hipMalloc(&ticket, 64 * sizeof(unsigned int)));
  CHECK(hipMemset(ticket, 0, 64 * sizeof(unsigned int)));
  CHECK(hipMalloc(&bad, sizeof(unsigned int)));
  CHECK(hipMemset(bad, 0, sizeof(unsigned int)));
  CHECK(hipHostMalloc((void**)&out, GROUPS * PAD * sizeof(double), hipHostMallocMapped));
  CHECK(hipMalloc(&stamps, (size_t)NB * 6 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h((size_t)NB * 6);
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  for (int stagger : {0, 400}) {                                   // all at once / spread over 4 us
    for (int local = 0; local < 2; local++) {
      const int ITERS = 400;
      double acc[4] = {0, 0, 0, 0}, worst[4] = {0, 0, 0, 0};
      long n_blocks = 0, n_last = 0, off_rule = 0;
      std::vector<double> kernel_span;
      for (int it = 0; it < ITERS; it++) {
        if (local) hipLaunchKernelGGL(handoff<true>, dim3(NB), dim3(256), 0, st, partials, ticket, out, stamps, (unsigned)it, stagger, bad);
        else hipLaunchKernelGGL(handoff<false>, dim3(NB), dim3(256), 0, st, partials, ticket, out, stamps, (unsigned)it, stagger, bad);
        CHECK(hipMemcpyAsync(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        if (it < 20) continue;
        unsigned long long first = ~0ull, last = 0;
        for (int b = 0; b < NB; b++) {
          const unsigned long long* s = &h[(size_t)b * 6];
          const double ack = (double)(s[1] - s[0]) * 0.01, tick = (double)(s[2] - s[1]) * 0.01;
          acc[0] += ack; acc[1] += tick; worst[0] = std::max(worst[0], ack); worst[1] = std::max(worst[1], tick);
          n_blocks++;
          first = std::min(first, s[0]);
          last = std::max(last, std::max(s[2], s[4]));
          if (s[4]) {
            const double gat = (double)(s[3] - s[2]) * 0.01, pub = (double)(s[4] - s[3]) * 0.01;
            acc[2] += gat; acc[3] += pub; worst[2] = std::max(worst[2], gat); worst[3] = std::max(worst[3], pub);
            n_last++;
          }
          if ((int)(s[5] & 0xff) != (b & 7)) off_rule++;
        }
        kernel_span.push_back((double)(last - first) * 0.01);
      }
      std::sort(kernel_span.begin(), kernel_span.end());
      unsigned int nbad = 0;
      CHECK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
      printf("stagger %3d  scope %-6s  store ack %.2f us (worst %.2f)  ticket %.2f (worst %.2f)  gather %.2f (worst %.2f)  publish %.2f (worst %.2f)  "
             "first start -> last end: median %.2f us  | workgroups off the b%%8 rule: %ld of %ld, wrong sums: %u\n",
             stagger, local ? "L2" : "agent", acc[0] / n_blocks, worst[0], acc[1] / n_blocks, worst[1], acc[2] / std::max(1l, n_last), worst[2],
             acc[3] / std::max(1l, n_last), worst[3], kernel_span[kernel_span.size() / 2], off_rule, n_blocks, nbad);
    }
  }
  // the placement rule with other launch shapes (the pass also runs with 513 workgroups -- a chain's extra one -- and 2048)
  for (int nb : {8, 64, 513, 2048, 4096}) {
    double* p2; unsigned long long* s2;
    CHECK(hipMalloc(&p2, (size_t)(nb + 8) * PAD * sizeof(double)));
    CHECK(hipMalloc(&s2, (size_t)nb * 6 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h2((size_t)nb * 6);
    long off = 0, tot = 0;
    for (int it = 0; it < 20; it++) {
      CHECK(hipMemsetAsync(ticket, 0, 64 * sizeof(unsigned int), st));
      hipLaunchKernelGGL(handoff<false>, dim3(nb), dim3(256), 0, st, p2, ticket + 16, out, s2, (unsigned)it, 0, bad + 0);
      CHECK(hipMemcpyAsync(h2.data(), s2, h2.size() * 8, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      for (int b = 0; b < nb; b++) { tot++; if ((int)(h2[(size_t)b * 6 + 5] & 0xff) != (b & 7)) off++; }
    }
    printf("launch of %4d workgroups: %ld of %ld off the b%%8 rule\n", nb, off, tot);
    (void)hipFree(p2); (void)hipFree(s2);
  }
  return 0;
}

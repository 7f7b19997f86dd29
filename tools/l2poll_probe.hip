// Developer probe (GPU box): do loads at L2 scope (sc0: past the CU's cache, served by the XCD's L2) see the results of agent-scope
// atomic adds by OTHER workgroups of the same XCD?  512 workgroups; each adds 1 to the word of its XCD (b % 8) and one thread polls
// that word until it reads 64 -- with sc0 loads, then with sc1 loads (past the L2) -- and the launch's duration says what 512 pollers
// of 8 words cost either way.  (Asked by the shared tail: its lists' counts and cursors are polled by thousands of waves.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2poll_probe.hip -o tools/l2poll_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>   // 0: sc0 loads, 1: sc1 loads
__global__ __launch_bounds__(256) void poll(unsigned int* __restrict__ words, unsigned int* __restrict__ out, unsigned int target, int spread_us) {
  const int b = blockIdx.x, x = b & 7;
  unsigned int* w = words + x * 64;                 // one line per XCD
  if (threadIdx.x == 0) {
    // arrivals spread over spread_us
    const unsigned long long t0 = wall_clock64();
    const unsigned long long d = (unsigned long long)((b * 2654435761u) >> 24) * (unsigned long long)(spread_us * 100) / 256ull;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(2);
    __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int v = 0, looks = 0;
    const unsigned long long t1 = wall_clock64();
    for (;;) {
      if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(w) : "memory");
      else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(w) : "memory");
      looks++;
      if (v >= target) break;
      if (wall_clock64() - t1 > 500000ull) break;      // 5 ms
    }
    out[b * 4] = v; out[b * 4 + 1] = looks; out[b * 4 + 2] = (unsigned int)(wall_clock64() - t1); out[b * 4 + 3] = (unsigned int)(wall_clock64() - t0);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  const int NB = 512;
  unsigned int *words, *out;
  CHECK(hipMalloc(&words, 8 * 64 * 4));
  CHECK(hipMalloc(&out, NB * 16));
  std::vector<unsigned int> h(NB * 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; mode++) {
    for (int spread : {0, 10}) {
      std::vector<double> dur, wait;
      long stale = 0, looks = 0;
      for (int it = 0; it < 30; it++) {
        CHECK(hipMemset(words, 0, 8 * 64 * 4));
        CHECK(hipDeviceSynchronize());
        if (mode == 0) hipExtLaunchKernelGGL(poll<0>, dim3(NB), dim3(256), 0, 0, e0, e1, 0, words, out, 64u, spread);
        else hipExtLaunchKernelGGL(poll<1>, dim3(NB), dim3(256), 0, 0, e0, e1, 0, words, out, 64u, spread);
        CHECK(hipMemcpy(h.data(), out, NB * 16, hipMemcpyDeviceToHost));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it < 5) continue;
        dur.push_back(ms * 1e3);
        for (int b = 0; b < NB; b++) { if (h[b * 4] < 64) stale++; looks += h[b * 4 + 1]; wait.push_back(h[b * 4 + 2] * 0.01); }
      }
      std::sort(dur.begin(), dur.end()); std::sort(wait.begin(), wait.end());
      printf("%s loads, arrivals over %2d us: launch %.1f us (median of 25); a poller's wait median %.2f / max %.2f us, %.0f looks each; pollers that gave up: %ld\n",
             mode == 0 ? "sc0 (L2)" : "sc1 (past L2)", spread, dur[dur.size() / 2], wait[wait.size() / 2], wait.back(), (double)looks / (25.0 * NB), stale);
    }
  }
  return 0;
}

// Developer probe (GPU box): where do the four waves of a 256-thread workgroup run?  HW_REG_HW_ID of every wave of 512-workgroup
// launches (the pass's shape: 2 workgroups per CU): are a workgroup's waves on four different SIMDs, and which workgroups share a CU?
// (Asked by the idea of packing a workgroup's 128 fit rows into two full waves: the two fitting waves of the two workgroups of a CU
// should sit on different SIMDs.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/simd_probe.hip -o tools/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void where(unsigned* __restrict__ out, int spin) {
  __shared__ float pad[10240];                      // 40 KB: the pass's footprint (at most three workgroups per CU)
  pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);      // stay resident: co-residents overlap
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (xcc & 0xf) | ((unsigned)(pad[threadIdx.x] != 0.f) << 31);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  for (int nb : {512, 513, 1024}) {
    unsigned* d;
    CHECK(hipMalloc(&d, (size_t)nb * 8 * sizeof(unsigned)));
    std::vector<unsigned> h((size_t)nb * 8);
    long wg_four_simds = 0, wgs = 0, rule_ok = 0, pairs = 0, pair_rule = 0;
    std::map<int, int> first_simd_hist;
    for (int it = 0; it < 20; it++) {
      hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, 0, d, 500);      // 5 us resident
      CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
      std::map<unsigned, std::vector<int>> by_cu;                         // (xcc, se, sh, cu) -> workgroups
      for (int b = 0; b < nb; b++) {
        std::set<unsigned> simds;
        unsigned cu_key = 0;
        for (int w = 0; w < 4; w++) {
          const unsigned hw = h[(size_t)(b * 4 + w) * 2], xcc = h[(size_t)(b * 4 + w) * 2 + 1] & 0xf;
          const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
          simds.insert(simd);
          cu_key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
          if (w == 0) first_simd_hist[(int)simd]++;
          if ((int)simd == w) rule_ok++;
        }
        wgs++;
        if (simds.size() == 4) wg_four_simds++;
        by_cu[cu_key].push_back(b);
      }
      if (it == 0) {
        printf("launch of %d workgroups: %zu distinct CUs; first CUs:", nb, by_cu.size());
        int shown = 0;
        for (auto& kv : by_cu) { if (shown++ >= 6) break; printf("  [%05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("]"); }
        printf("\n");
      }
      for (auto& kv : by_cu)
        if (kv.second.size() == 2) { pairs++; if ((((kv.second[0] >> 3) >> 5) & 1) != (((kv.second[1] >> 3) >> 5) & 1)) pair_rule++; }
    }
    printf("  workgroups whose four waves sit on four different SIMDs: %ld of %ld; waves with simd == wave index: %ld of %ld; first wave's SIMD histogram:", wg_four_simds, wgs, rule_ok, wgs * 4);
    for (auto& kv : first_simd_hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n  CUs holding exactly two workgroups: %ld; of those with different ((b >> 3) >> 5) & 1: %ld\n", pairs, pair_rule);
    (void)hipFree(d);
  }
  return 0;
}

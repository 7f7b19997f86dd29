// Developer probe (GPU box): how long a launch of the pass's shape takes to get all of its workgroups started ("ramp"), by
// workgroup size and shared memory per workgroup.  The one-launch pass: 512 workgroups x 256 threads, 42 KB of shared memory,
// 168 VGPRs -- its last workgroup starts 1.3 us after the first (profiles/r06/withdrawn_experiments.md section 0).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ramp_probe.hip -o tools/ramp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void stamp_kernel(unsigned long long* __restrict__ t, int spin_us) {
  extern __shared__ float lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = t0;
  lds[threadIdx.x] = (float)t0;
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(4);      // stay resident like a pass does
  if (lds[(threadIdx.x + 1) % blockDim.x] == -1.f) t[blockIdx.x] = 0;
}

int main() {
  CHECK(hipSetDevice(0));
  unsigned long long* d;
  CHECK(hipMalloc(&d, 8192 * 8));
  std::vector<unsigned long long> h(8192);
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  struct Shape { int blocks, threads, lds_kb; };
  for (Shape s : {Shape{512, 256, 0}, Shape{512, 256, 42}, Shape{512, 256, 64}, Shape{256, 512, 84}, Shape{256, 512, 0}, Shape{1024, 128, 21}, Shape{256, 1024, 0},
                  Shape{576, 256, 42}, Shape{2048, 256, 42}}) {
    CHECK(hipFuncSetAttribute((const void*)stamp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    std::vector<double> med, last;
    for (int it = 0; it < 60; it++) {
      hipLaunchKernelGGL(stamp_kernel, dim3(s.blocks), dim3(s.threads), (size_t)s.lds_kb * 1024, st, d, 10);
      CHECK(hipMemcpyAsync(h.data(), d, s.blocks * 8, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      if (it < 10) continue;
      std::vector<unsigned long long> v(h.begin(), h.begin() + s.blocks);
      std::sort(v.begin(), v.end());
      med.push_back((double)(v[v.size() / 2] - v[0]) * 0.01);
      last.push_back((double)(v.back() - v[0]) * 0.01);
    }
    std::sort(med.begin(), med.end()); std::sort(last.begin(), last.end());
    printf("%5d workgroups x %4d threads, %3d KB shared: median workgroup starts %.2f us after the first, last %.2f us (medians of 50 launches)\n",
           s.blocks, s.threads, s.lds_kb, med[med.size() / 2], last[last.size() / 2]);
  }
  return 0;
}

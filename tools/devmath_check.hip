// Developer check (GPU box): the float32 device math of flimo_math.h is bit-identical to the same
// source compiled for the host (x86-64, no FMA).  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -I fast_limo_amd/csrc/hip tools/devmath_check.hip -o /tmp/devmath_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "flimo_math.h"
using namespace flimo;

struct In { float px[5], py[5], pz[5]; float a, b; };
struct Out { float n[4]; int ok; float s, d, sq; };

__host__ __device__ void eval(const In& in, Out& o) {
  plane_fit5(in.px, in.py, in.pz, o.n);
  o.ok = plane_eval5(o.n, in.px, in.py, in.pz, 0.05f) ? 1 : 0;
  o.s = fl_sqrt(fabsf(in.a));
  o.d = fl_div(in.a, in.b);
  o.sq = sqdist3(in.a, in.b, in.px[0], in.py[0], in.pz[0], in.px[1]);
}
__global__ void k(const In* in, Out* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) eval(in[i], out[i]);
}
__global__ void k_sincos(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) libm_sincosf(x[i], s[i], c[i]);
}
// libm_sincosf on the device against the host's libm (sinf / cosf), three argument ranges
static int check_sincos() {
  const int N = 1 << 22;
  std::vector<float> x(N), hs(N), hc(N), gs(N), gc(N);
  srand(11);
  for (int t = 0; t < N; t++) {
    const double u = (rand() % 2000001 - 1000000) * 1e-6;
    x[t] = (float)(t % 4 == 0 ? u * 100.0 : (t % 4 == 1 ? u * 1e-3 : u * 0.785));
    hs[t] = sinf(x[t]); hc[t] = cosf(x[t]);
  }
  float *dx, *ds, *dc;
  hipMalloc(&dx, N * 4); hipMalloc(&ds, N * 4); hipMalloc(&dc, N * 4);
  hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sincos, dim3(N / 256), dim3(256), 0, 0, dx, ds, dc, N);
  hipMemcpy(gs.data(), ds, N * 4, hipMemcpyDeviceToHost);
  hipMemcpy(gc.data(), dc, N * 4, hipMemcpyDeviceToHost);
  int small_bad = 0, large_bad = 0, n_small = 0;
  for (int t = 0; t < N; t++) {
    const bool bad = memcmp(&hs[t], &gs[t], 4) || memcmp(&hc[t], &gc[t], 4);
    if (fabsf(x[t]) < 0.785f) { n_small++; small_bad += bad; } else large_bad += bad;
  }
  printf("DEVMATH sincos mismatches vs host libm: |x|<pi/4: %d of %d   up to 100 rad: %d of %d\n", small_bad, n_small, large_bad, N - n_small);
  return small_bad != 0 || large_bad > N / 100000;
}
__global__ void k_atan2(const float* y, const float* x, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = libm_atan2f(y[i], x[i]);
}
// libm_atan2f on the device against the host's libm (atan2f): lidar-like points, every breakpoint region of the reduction, the
// axes, tiny / huge ratios, signed zeros, infinities
static int check_atan2() {
  const int N = 1 << 24;
  std::vector<float> y(N), x(N), h(N), g(N);
  srand(23);
  auto ru = []() { return (rand() % 2000001 - 1000000) * 1e-6; };
  const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-30f, -1e-30f, 1e30f, -1e30f, INFINITY, -INFINITY, 0.4375f, 0.6875f, 1.1875f, 2.4375f,
                           3.0e-9f, 33554432.0f, 1.17549435e-38f, 1e-42f};
  const int ns = (int)(sizeof(special) / sizeof(special[0]));
  for (int t = 0; t < N; t++) {
    const int mode = t % 8;
    double a = ru(), b = ru();
    if (mode == 0) { y[t] = (float)(a * 100.0); x[t] = (float)(b * 100.0); }
    else if (mode == 1) { y[t] = (float)(a * 100.0); x[t] = (float)(b * 1e-3); }             // around +-pi/2
    else if (mode == 2) { y[t] = (float)(a * 1e-3); x[t] = (float)(b * 100.0); }             // around 0 and +-pi
    else if (mode == 3) { const double r = 0.3 + 2.5 * fabs(a); x[t] = (float)(b * 50.0); y[t] = (float)(r * x[t]); }   // ratios across the breakpoints
    else if (mode == 4) { y[t] = (float)(a * pow(10.0, 30.0 * b)); x[t] = (float)(ru() * pow(10.0, 30.0 * ru())); }
    else if (mode == 5) { y[t] = special[rand() % ns]; x[t] = special[rand() % ns]; }
    else if (mode == 6) { y[t] = (float)a; x[t] = 1.0f; }
    else { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(), v = ((unsigned)rand() << 16) ^ (unsigned)rand(); memcpy(&y[t], &u, 4); memcpy(&x[t], &v, 4); }
    h[t] = atan2f(y[t], x[t]);
  }
  float *dy, *dx, *dout;
  hipMalloc(&dy, (size_t)N * 4); hipMalloc(&dx, (size_t)N * 4); hipMalloc(&dout, (size_t)N * 4);
  hipMemcpy(dy, y.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_atan2, dim3(N / 256), dim3(256), 0, 0, dy, dx, dout, N);
  hipMemcpy(g.data(), dout, (size_t)N * 4, hipMemcpyDeviceToHost);
  int bad = 0, shown = 0;
  for (int t = 0; t < N; t++) {
    const bool nan_both = (h[t] != h[t]) && (g[t] != g[t]);
    if (!nan_both && memcmp(&h[t], &g[t], 4)) {
      bad++;
      if (shown++ < 8) printf("  atan2f(%a, %a): host %a device %a (mode %d)\n", y[t], x[t], h[t], g[t], t % 8);
    }
  }
  printf("DEVMATH atan2f mismatches vs host libm: %d of %d\n", bad, N);
  hipFree(dy); hipFree(dx); hipFree(dout);
  return bad != 0;
}
int main() {
  if (check_sincos()) return 2;
  if (check_atan2()) return 3;
  const int N = 1 << 20;
  std::vector<In> in(N);
  std::vector<Out> ho(N), go(N);
  srand(7);
  auto rf = []() { return (rand() % 200001 - 100000) * 1e-5f; };
  for (int t = 0; t < N; t++) {
    int mode = t % 4;
    float ox = rf() * 100.f, oy = rf() * 100.f, oz = rf() * 10.f;
    for (int i = 0; i < 5; i++) {
      float a = rf() * 0.5f, b = rf() * 0.5f, c = rf() * (mode == 0 ? 0.01f : (mode == 1 ? 0.5f : 0.f));
      if (mode == 2) { in[t].px[i] = ox + a; in[t].py[i] = oy + c; in[t].pz[i] = oz + b; }
      else if (mode == 3) { in[t].px[i] = ox + a; in[t].py[i] = oy + a * 0.5f + c; in[t].pz[i] = oz + rf() * 1e-3f; }
      else { in[t].px[i] = ox + a; in[t].py[i] = oy + b; in[t].pz[i] = oz + c; }
    }
    in[t].a = rf() * (t % 7 == 0 ? 1e-30f : 3.f);
    in[t].b = rf() * (t % 11 == 0 ? 1e-25f : 7.f) + 1e-9f;
  }
  for (int t = 0; t < N; t++) eval(in[t], ho[t]);
  In* din; Out* dout;
  hipMalloc(&din, N * sizeof(In)); hipMalloc(&dout, N * sizeof(Out));
  hipMemcpy(din, in.data(), N * sizeof(In), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, 0, din, dout, N);
  hipMemcpy(go.data(), dout, N * sizeof(Out), hipMemcpyDeviceToHost);
  int bn = 0, bok = 0, bs = 0, bd = 0, bq = 0;
  for (int t = 0; t < N; t++) {
    if (memcmp(ho[t].n, go[t].n, 16)) { if (bn < 5) printf("n mismatch t=%d mode=%d: %.9g %.9g %.9g %.9g | %.9g %.9g %.9g %.9g\n", t, t % 4, ho[t].n[0], ho[t].n[1], ho[t].n[2], ho[t].n[3], go[t].n[0], go[t].n[1], go[t].n[2], go[t].n[3]); bn++; }
    if (ho[t].ok != go[t].ok) bok++;
    if (memcmp(&ho[t].s, &go[t].s, 4)) { if (bs < 3) printf("sqrt mismatch a=%.9g: %.9g %.9g\n", in[t].a, ho[t].s, go[t].s); bs++; }
    if (memcmp(&ho[t].d, &go[t].d, 4)) { if (bd < 3) printf("div mismatch %.9g/%.9g: %.9g %.9g\n", in[t].a, in[t].b, ho[t].d, go[t].d); bd++; }
    if (memcmp(&ho[t].sq, &go[t].sq, 4)) bq++;
  }
  printf("DEVMATH mismatches: plane_n=%d plane_ok=%d sqrt=%d div=%d sqdist=%d of %d\n", bn, bok, bs, bd, bq, N);
  return (bn || bok || bs || bd || bq) ? 1 : 0;
}

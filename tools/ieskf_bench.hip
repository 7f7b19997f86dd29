// tools/ieskf_bench.hip -- developer tool (GPU box): the device filter's algebra (fast_limo_amd/csrc/hip/flimo_ieskf.h) on its own,
// as the two one-workgroup launches of the A/B form (flimo_ieskf.hip): duration per launch (HIP events on the dispatch), phase
// stamps, bit-reproducibility, and parity with the host filter (csrc/host/flimo_ikfom.cpp) on synthetic sums -- the same H^T H /
// H^T h in every iteration, like flimo_eskf_update_fixed.   Build: tools/ieskf_bench.sh
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "../fast_limo_amd/csrc/hip/flimo_ieskf.hip"
#include "../fast_limo_amd/csrc/host/flimo_ikfom.hpp"

using namespace flimo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  const int max_iter = 3;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> N01(0.0, 1.0);
  double x[26] = {0};
  x[0] = 0.3; x[1] = -0.2; x[2] = 0.1;
  { const double a = 0.02; x[3] = 0; x[4] = 0; x[5] = sin(a / 2); x[6] = cos(a / 2); }
  x[10] = 1.0;
  x[23] = 0.05; x[24] = -0.03; x[25] = -9.8088; { double n = sqrt(x[23]*x[23]+x[24]*x[24]+x[25]*x[25]); for (int i = 23; i < 26; i++) x[i] *= 9.809 / n; }
  double P[529];
  {
    double A[529];
    for (int i = 0; i < 529; i++) A[i] = 0.02 * N01(rng);
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) { double s = 0; for (int k = 0; k < 23; k++) s += A[i * 23 + k] * A[j * 23 + k]; P[i * 23 + j] = 1e-3 * s + (i == j ? 1e-4 : 0.0); }
  }
  const int M = 5000;
  double HTH[144] = {0}, HTh[12] = {0};
  for (int m = 0; m < M; m++) {
    double h[12];
    for (int i = 0; i < 3; i++) h[i] = N01(rng);
    { double n = sqrt(h[0]*h[0]+h[1]*h[1]+h[2]*h[2]); for (int i = 0; i < 3; i++) h[i] /= n; }
    for (int i = 3; i < 12; i++) h[i] = 3.0 * N01(rng);
    const double r = 0.02 * N01(rng) + 0.05 * h[0];
    for (int i = 0; i < 12; i++) { for (int j = 0; j < 12; j++) HTH[i * 12 + j] += h[i] * h[j]; HTh[i] += h[i] * r; }
  }
  double limits[23]; for (int i = 0; i < 23; i++) limits[i] = 1e-4;
  // ---- host filter ----
  flimo_host::Esekf f;
  flimo_host::StateIkfom s; s.from_flat(x); f.change_x(s);
  flimo_host::Esekf::Cov C; memcpy(&C.a[0][0], P, sizeof(P)); f.change_P(C);
  f.init(max_iter, limits); f.keep_log = true;
  f.h_reduced = [&](const flimo_host::StateIkfom&, flimo_host::ReducedMeas& o) { o.M = M; memcpy(o.HTH, HTH, sizeof(HTH)); memcpy(o.HTh, HTh, sizeof(HTh)); };
  f.update_iterated_dyn_share_modified(0.001, 5.0);
  // ---- device ----
  ChainState* S; CK(hipMalloc(&S, sizeof(ChainState))); CK(hipMemset(S, 0, sizeof(ChainState)));
  double2* gran; CK(hipMalloc(&gran, FIT_GROUPS * FIT_LIVE_PAD * sizeof(double2)));
  const unsigned long long seq = 42, tag = 0x4000000000000001ull;
  {
    std::vector<double2> g(FIT_GROUPS * FIT_LIVE_PAD);
    const double tagd = *(const double*)&seq;
    for (auto& e : g) { e.x = 0.0; e.y = tagd; }
    int k = 0;
    for (int i = 0; i < 12; i++) for (int j = i; j < 12; j++) g[k++].x = HTH[i * 12 + j];
    for (int i = 0; i < 12; i++) g[k++].x = HTh[i];
    g[k++].x = (double)M;
    g[FIT_LIVE].x = 17.0; g[FIT_LIVE + 1].x = 0.0;
    CK(hipMemcpy(gran, g.data(), g.size() * sizeof(double2), hipMemcpyHostToDevice));
  }
  ChainPrior* pr; CK(hipHostMalloc((void**)&pr, sizeof(ChainPrior), hipHostMallocMapped));
  memcpy(pr->x, x, sizeof(x)); memcpy(pr->P, P, sizeof(P)); memcpy(pr->limit, limits, sizeof(limits));
  pr->R = 0.001; pr->D = 5.0; pr->max_iter = max_iter; pr->pad = 0;
  ik_pre_serial(x, x, P, 0.001, pr->dxn, pr->AG, pr->AG + 144);
  PoseMats P0; pose_from_x26(x, P0);
  double* res; CK(hipHostMalloc((void**)&res, CH_RES * 2 * sizeof(double), hipHostMallocMapped)); memset(res, 0, CH_RES * 2 * sizeof(double));
  double* lg; CK(hipHostMalloc((void**)&lg, CH_MAX_PASSES * CH_LOGN * 2 * sizeof(double), hipHostMallocMapped));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e[16]; for (auto& v : e) CK(hipEventCreate(&v));
  ChainCtl ctl{};
  ctl.S = S; ctl.gran = gran; ctl.res = (double2*)res; ctl.log = (double2*)lg; ctl.tag = tag;
  auto chain = [&](unsigned long long tg, bool log, bool ev) {
    for (int i = 0; i <= max_iter; i++) {
      ChainCtl c2 = ctl; c2.tag = tg; c2.log = log ? ctl.log : nullptr; c2.prior = i == 0 ? pr : nullptr;
      launch_ieskf_extra(st, c2, ev ? e[4 * i] : nullptr, ev ? e[4 * i + 1] : nullptr);
      launch_ieskf(st, c2, seq, i == 0 ? P0.RT : nullptr, ev ? e[4 * i + 2] : nullptr, ev ? e[4 * i + 3] : nullptr);
    }
    CK(hipStreamSynchronize(st));
  };
  double tx[4] = {0, 0, 0, 0}, tf[4] = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++) {
    chain(tag + r, r == 0, true);
    for (int i = 0; i <= max_iter; i++) {
      float a = 0, b = 0;
      CK(hipEventElapsedTime(&a, e[4 * i], e[4 * i + 1])); CK(hipEventElapsedTime(&b, e[4 * i + 2], e[4 * i + 3]));
      if (r >= reps / 4) { tx[i] += a; tf[i] += b; }
    }
  }
  const int nt = reps - reps / 4;
  const int passes_dev = (int)llround(res[2 * CH_PASSES]);
  printf("device: status %d reason %d iterations completed %d, it_next %d t %d (host filter logged %zu passes)\n", (int)llround(res[2 * CH_STATUS]),
         (int)llround(res[2 * CH_BAIL]), passes_dev, (int)llround(res[2 * CH_IT]), (int)llround(res[2 * CH_T]), f.log.size());
  printf("extra workgroup (measurement-independent half) [us]: first (copies the prior) %.2f, then %.2f %.2f %.2f\n", 1e3 * tx[0] / nt, 1e3 * tx[1] / nt, 1e3 * tx[2] / nt, 1e3 * tx[3] / nt);
  printf("final stage (sums -> next state) [us]: %.2f %.2f %.2f, handing back %.2f\n", 1e3 * tf[0] / nt, 1e3 * tf[1] / nt, 1e3 * tf[2] / nt, 1e3 * tf[3] / nt);
  double dlog = 0, dxa = 0;
  for (int p = 0; p < passes_dev && p < (int)f.log.size(); p++) {
    for (int k = 0; k < 23; k++) dlog = fmax(dlog, fabs(lg[2 * (p * CH_LOGN + 156 + k)] - f.log[p].dx[k]));
    for (int k = 0; k < 26; k++) dxa = fmax(dxa, fabs(lg[2 * (p * CH_LOGN + 179 + k)] - f.log[p].x_after[k]));
  }
  double dxm = 0;
  if (passes_dev >= 1) for (int k = 0; k < 26; k++) dxm = fmax(dxm, fabs(res[2 * (CH_X + k)] - f.log[passes_dev - 1].x_after[k]));
  printf("vs host filter: per-pass step %.3e, state after each pass %.3e, handed-back state %.3e; sums handed back: M %.0f HTH[0] %.17g (in %.17g)\n", dlog, dxa, dxm,
         res[2 * (CH_SUMS + 90)], res[2 * CH_SUMS], HTH[0]);
  {
    std::vector<double> ref(CH_RES);
    for (int k = 0; k < CH_RES; k++) ref[k] = res[2 * k];
    long bad = 0;
    const int nrep = reps * 10;
    for (int r = 0; r < nrep; r++) {
      chain(tag + 100000 + r, false, false);
      for (int k = 0; k < CH_RES; k++) if (memcmp(&ref[k], &res[2 * k], 8) != 0) { bad++; break; }
    }
    printf("reproducibility: %ld of %d chains differ from the first\n", bad, nrep);
  }
#ifdef IESKF_STAMPS
  {
    const char* names[12] = {"start", "loaded", "", "", "T, v built", "GJ done", "", "u, dx_ (+ Cholesky test)", "boxplus + conv", "", "END (goes on)", "END (hands back)"};
    for (int which = 0; which < 2; which++) {
      const int upto = which ? max_iter : 1;
      for (int i = 0; i <= upto; i++) {
        ChainCtl c2 = ctl; c2.tag = tag + 5000; c2.log = nullptr; c2.prior = i == 0 ? pr : nullptr;
        launch_ieskf_extra(st, c2);
        launch_ieskf(st, c2, seq, i == 0 ? P0.RT : nullptr);
      }
      CK(hipStreamSynchronize(st));
      unsigned long long t[32];
      CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ik_stamps), sizeof(t)));
      printf("%s iteration:", which ? "handing-back" : "middle");
      for (int k = 1; k < 12; k++) if (names[k][0] && (which ? k != 10 : k != 11)) printf("  %s +%.2f", names[k], 0.01 * (double)(long long)(t[k] - t[0]));
      printf(" us\n");
    }
  }
#endif
  CK(hipEventRecord(e[0], st));
  for (int r = 0; r < reps; r++)
    for (int i = 0; i <= max_iter; i++) { ChainCtl c2 = ctl; c2.tag = tag + 9000 + r; c2.log = nullptr; c2.prior = i == 0 ? pr : nullptr; launch_ieskf(st, c2, seq, i == 0 ? P0.RT : nullptr); }
  CK(hipEventRecord(e[1], st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e[0], e[1]));
  printf("back to back: %.2f us per final-stage launch (kernel + boundary)\n", 1e3 * ms / (reps * (max_iter + 1)));
  return 0;
}

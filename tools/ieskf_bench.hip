// tools/ieskf_bench.hip -- developer tool (GPU box): the device filter's algebra kernel (fast_limo_amd/csrc/hip/flimo_ieskf.hip) on
// its own: duration per launch (HIP events on the dispatch) and parity with the host filter (csrc/host/flimo_ikfom.cpp) on
// synthetic sums -- the same H^T H / H^T h in every iteration, like flimo_eskf_update_fixed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ifast_limo_amd/csrc/hip -Ifast_limo_amd/csrc/host -Iinclude \
//         tools/ieskf_bench.hip fast_limo_amd/csrc/host/flimo_ikfom.cpp -o tools/ieskf_bench
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
  // a plausible prior: attitude near identity, P = diag-dominant SPD
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
  // sums of M random measurement rows
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
  double xh[26]; f.get_x().to_flat(xh);
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
  { PoseMats P0; pose_from_x26(x, P0); memcpy(pr->RT0, P0.RT, sizeof(pr->RT0)); }
  double* res; CK(hipHostMalloc((void**)&res, CH_RES * 2 * sizeof(double), hipHostMallocMapped)); memset(res, 0, CH_RES * 2 * sizeof(double));
  double* lg; CK(hipHostMalloc((void**)&lg, CH_MAX_PASSES * CH_LOGN * 2 * sizeof(double), hipHostMallocMapped));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e[8]; for (auto& v : e) CK(hipEventCreate(&v));
  double tsum[4] = {0, 0, 0, 0}; int passes_dev = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = 0; i <= max_iter; i++)
      launch_ieskf(st, S, gran, seq, i == 0 ? pr : nullptr, res, r == 0 ? lg : nullptr, tag + r, e[2 * i], e[2 * i + 1]);
    CK(hipStreamSynchronize(st));
    passes_dev = (int)llround(res[2 * CH_PASSES]);
    for (int i = 0; i < passes_dev && i < 4; i++) { float ms = 0; CK(hipEventElapsedTime(&ms, e[2 * i], e[2 * i + 1])); if (r >= reps / 4) tsum[i] += ms; }
  }
  // bit-reproducibility of the result granules, chain after chain (same inputs)
  {
    std::vector<double> ref(CH_RES);
    for (int k = 0; k < CH_RES; k++) ref[k] = res[2 * k];
#ifdef IESKF_STAMPS
    static double dref[8][529], dcur[8][529];
    CK(hipMemcpyFromSymbol(dref, HIP_SYMBOL(g_ik_dbg), sizeof(dref)));
    int shown = 0;
#endif
    long bad_chains = 0, bad_vals = 0; int first_bad = -1;
    const int nrep = reps * 20;
    for (int r = 0; r < nrep; r++) {
      for (int i = 0; i <= max_iter; i++) launch_ieskf(st, S, gran, seq, i == 0 ? pr : nullptr, res, nullptr, tag + 100000 + r);
      CK(hipStreamSynchronize(st));
      int nb = 0;
      for (int k = 0; k < CH_RES; k++) if (memcmp(&ref[k], &res[2 * k], 8) != 0) { nb++; if (first_bad < 0) first_bad = k; }
      if (nb) { bad_chains++; bad_vals += nb; }
#ifdef IESKF_STAMPS
      if (nb && shown < 4) {
        shown++;
        CK(hipMemcpyFromSymbol(dcur, HIP_SYMBOL(g_ik_dbg), sizeof(dcur)));
        const char* nm[6] = {"L (end)", "P_ (end)", "K_x (end)", "J (end)", "P_ (cov start)", "J (cov start)"};
        for (int a = 0; a < 6; a++) {
          int cnt = 0, first = -1;
          for (int k = 0; k < 529; k++) if (memcmp(&dref[a][k], &dcur[a][k], 8) != 0) { cnt++; if (first < 0) first = k; }
          if (cnt) printf("  chain %d: %s differs in %d values, first %d (row %d col %d)\n", r, nm[a], cnt, first, first / 23, first % 23);
        }
      }
#endif
    }
    printf("reproducibility: %ld of %d chains differ from the first (%ld values; first differing slot %d, CH_P = %d)\n", bad_chains, nrep, bad_vals, first_bad, CH_P);
  }
  const int nt = reps - reps / 4;
  printf("device: status %d bail %d passes %d (host %zu)\n", (int)llround(res[2 * CH_STATUS]), (int)llround(res[2 * CH_BAIL]), passes_dev, f.log.size());
  printf("algebra kernel [us]: first (prior from mapped memory) %.2f, middle %.2f %.2f, last (covariance) %.2f\n", 1e3 * tsum[0] / nt, 1e3 * tsum[1] / nt,
         1e3 * tsum[2] / nt, 1e3 * tsum[passes_dev - 1 < 4 ? passes_dev - 1 : 3] / nt);
  double dxm = 0, dPm = 0, dPrel = 0;
  for (int i = 0; i < 26; i++) dxm = fmax(dxm, fabs(res[2 * (CH_X + i)] - xh[i]));
  for (int i = 0; i < 529; i++) { const double a = res[2 * (CH_P + i)], b = f.get_P().a[i / 23][i % 23]; dPm = fmax(dPm, fabs(a - b)); dPrel = fmax(dPrel, fabs(a - b) / (fabs(b) + 1e-30)); }
  double dlog = 0;
  for (size_t p = 0; p < f.log.size() && (int)p < passes_dev; p++)
    for (int k = 0; k < 23; k++) dlog = fmax(dlog, fabs(lg[2 * (p * CH_LOGN + 156 + k)] - f.log[p].dx[k]));
  printf("vs host filter: max |dx state| %.3e, per-pass step %.3e, |dP| %.3e (rel %.3e)\n", dxm, dlog, dPm, dPrel);
#ifdef IESKF_STAMPS
  {
    // phase stamps of a middle iteration and of the last one (10 ns resolution)
    const char* names[12] = {"start", "loaded", "pre: chains done", "P re-projected", "T built", "LU done", "solved", "dx_", "boxplus", "cov: start", "END (middle)", "END (last)"};
    for (int which = 0; which < 2; which++) {
      launch_ieskf(st, S, gran, seq, pr, res, nullptr, tag + 5000);
      launch_ieskf(st, S, gran, seq, nullptr, res, nullptr, tag + 5000);
      if (which) { launch_ieskf(st, S, gran, seq, nullptr, res, nullptr, tag + 5000); launch_ieskf(st, S, gran, seq, nullptr, res, nullptr, tag + 5000); }
      CK(hipStreamSynchronize(st));
      unsigned long long t[32];
      CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ik_stamps), sizeof(t)));
      printf("%s iteration:", which ? "last" : "middle");
      for (int k = 1; k < 12; k++) if ((which ? k != 10 : k < 9 || k == 10)) printf("  %s +%.2f", names[k], 0.01 * (double)(long long)(t[k] - t[0]));
      printf(" us\n");
    }
  }
#endif
  // back-to-back chain of 4 algebra kernels with nothing between: wall time per kernel incl. the dispatch boundary
  CK(hipEventRecord(e[0], st));
  for (int r = 0; r < reps; r++)
    for (int i = 0; i <= max_iter; i++) launch_ieskf(st, S, gran, seq, i == 0 ? pr : nullptr, res, nullptr, tag + 1000 + r);
  CK(hipEventRecord(e[1], st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e[0], e[1]));
  printf("back to back: %.2f us per launch (kernel + boundary)\n", 1e3 * ms / (reps * (max_iter + 1)));
  return 0;
}

// fast_limo_amd/csrc/hip/flimo_ieskf.hip  -- gfx950 device code.
//
// The 23-dof algebra of ONE outer iteration of esekf::update_iterated_dyn_share_modified (IKFoM_toolkit/esekfom/esekfom.hpp:
// 1620-1823) as a one-block kernel queued behind the pass's launches (flimo_chain.h): from the pass's 91 sums (H^T H, H^T h, M) and
// the filter state kept in device memory to the next state, the convergence decision, the next pass's float32 pose constants and
// -- on the last iteration -- the covariance.  The host enqueues the <= MAX_NUM_ITERS + 1 passes of a scan back to back and reads
// the result once (flimo_update_chain, flimo_capi.hip).
//
// It is the SAME arithmetic as the host filter (csrc/host/flimo_ikfom.cpp: boxminus :1652, SO(3) / S2 re-projection of the
// covariance :1659-1697, gain through the matrix-inversion-lemma form of :1722-1729, step :1733, boxplus :1747, convergence
// :1757-1764, covariance :1766-1820), element by element in the same order.  What the device does NOT do: the M < 23 branch (dense
// H, :1701-1709), the eigen-decomposition of a degenerate H^T H (:1736-1744) and the repair of exactly tied distances -- rare
// branches: the chain stops (`bail`), and the host filter goes on from the state the device hands back.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "flimo_types.h"
#include "flimo_kernels.h"
#include "flimo_chain.h"
#include "flimo_ieskf.h"

#pragma clang fp contract(off)

namespace flimo {

// Filter state of one scan's update in device memory.  flat state x26: pos3 rot4(xyzw) offR4 offT3 vel3 bg3 ba3 grav3.
struct ChainState {
  ChainHead head;          // what the pass kernels read
  double x[26];
  double x_prop[26];
  double P_prop[529];
  double limit[IK_N];
  double R, D;
  int max_iter;            // MAX_NUM_ITERS: iterations it = -1 .. max_iter - 1
  int it;                  // next iteration: it = -1 + iterations done
  int t;                   // iterations that met the limits so far
  int passes;              // iterations completed by the device
  double info[3 * CH_MAX_PASSES];   // per pass: M, stragglers, ties
};
size_t chain_state_size() { return sizeof(ChainState); }

// more shared memory beyond the matrices of flimo_ieskf.h
constexpr int IKL_XC = IESKF_LDS_DOUBLES;        // x       26
constexpr int IKL_XP = IKL_XC + 26;              // x_prop  26
constexpr int IKL_LIM = IKL_XP + 26;             // limit   23
constexpr int IKL_LIVE = IKL_LIM + 23;           // the pass's sums 96
constexpr int IKL_END = IKL_LIVE + FIT_LIVE_PAD;

// (a) rows idx .. idx+B of M(.., 0:ncols) <- J * rows;  (b) cols idx .. idx+B of M <- cols * J^T    (host left_block / right_block_T)
template <int B>
__device__ inline void ik_left_block(double* M, int ld, int idx, const double* J, int ncols, int tid) {
  if (tid >= 0 && tid < ncols) {
    const int c = tid;
    double t[B];
#pragma unroll
    for (int r = 0; r < B; r++) { double s = 0; for (int k = 0; k < B; k++) s += J[r * B + k] * M[(idx + k) * ld + c]; t[r] = s; }
#pragma unroll
    for (int r = 0; r < B; r++) M[(idx + r) * ld + c] = t[r];
  }
}
template <int B>
__device__ inline void ik_right_block_T(double* M, int ld, int idx, const double* J, int nrows, int tid) {
  if (tid >= 0 && tid < nrows) {
    const int r = tid;
    double t[B];
#pragma unroll
    for (int c = 0; c < B; c++) { double s = 0; for (int k = 0; k < B; k++) s += M[r * ld + idx + k] * J[c * B + k]; t[c] = s; }
#pragma unroll
    for (int c = 0; c < B; c++) M[r * ld + idx + c] = t[c];
  }
}

// Developer-only phase stamps (tools/ieskf_bench.hip builds this file with -DIESKF_STAMPS): thread 0 stores the 100 MHz wall clock
#ifdef IESKF_STAMPS
__device__ unsigned long long g_ik_stamps[32];
#define IK_STAMP(k) do { if (threadIdx.x == 0) g_ik_stamps[k] = wall_clock64(); } while (0)
__device__ double g_ik_dbg[8][529];
#define IK_DUMP(slot, arr, cnt) do { for (int q_ = threadIdx.x; q_ < (cnt); q_ += 256) g_ik_dbg[slot][q_] = (arr)[q_]; } while (0)
#else
#define IK_DUMP(slot, arr, cnt) do {} while (0)
#define IK_STAMP(k) do {} while (0)
#endif

typedef double v2d_t __attribute__((ext_vector_type(2)));
// (s_nop 1: a VMEM store of more than 64 bits must be followed by two wait states before a VALU instruction may overwrite its data
//  registers on gfx940+ -- the compiler's hazard recognizer inserts them for its own stores and cannot see into inline assembly.
//  Without them the low dword of a stored double was, now and then, the NEXT value's: 1e-6 relative, run-to-run different.)
__device__ __forceinline__ void put_granule(double2* base, int slot, double value, unsigned long long tag) {
  v2d_t g;
  g.x = value;
  g.y = __longlong_as_double((long long)tag);
  double2* o = base + slot;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
}

// ---- the measurement-independent half of an iteration (esekfom.hpp:1652-1697): x boxminus x_prop, the SO(3) / S2 blocks, P_prop
//      through them, PR = P_[:, 0:12] / R.  In: xc, xp, P_ (= P_prop) in shared memory.  Out (shared memory): dx, dxn, Jb, P_, PR.
//      Block-wide (256 threads); ends with a barrier. ----
__device__ __forceinline__ void ik_pre_block(double* lds, double R, int tid) {
  const int n = IK_N;
  double* P_ = lds + IKL_P;
  double* PR = lds + IKL_PR;
  double* dx = lds + IKL_DX;
  double* dxn = lds + IKL_DXN;
  double* Jb = lds + IKL_J;
  const double* xc = lds + IKL_XC;
  const double* xp = lds + IKL_XP;
  // independent chains: rot and offset_R_L_I run the same code in two lanes of one wave; the S2 chain on another wave
  if (tid < 2) {
    const int o = (tid == 0) ? 3 : 7;                        // rot / offset_R_L_I in the flat state
    const int idx = (tid == 0) ? 3 : 6;
    const Q4 a{xc[o], xc[o + 1], xc[o + 2], xc[o + 3]};
    const Q4 b{-xp[o], -xp[o + 1], -xp[o + 2], xp[o + 3]};   // conj(other)
    double r[3];
    ik_so3_log(ik_qmul(b, a), r);
    for (int i = 0; i < 3; i++) dx[idx + i] = r[i];
    ik_A_T(r, Jb + 9 * tid);
  } else if (tid == 64) {
    double d[2];
    ik_s2_boxminus(xc + 23, xp + 23, d);
    dx[21] = d[0]; dx[22] = d[1];
    ik_s2_J(xc + 23, xp + 23, d, Jb + 18);
  } else if (tid >= 128 && tid < 128 + 15) {
    const int e = tid - 128, seg = e / 3, i = e % 3;         // pos, offset_T_L_I, vel, bg, ba
    const int xo = seg == 0 ? 0 : 8 + 3 * seg, dxo = seg == 0 ? 0 : 6 + 3 * seg;
    dx[dxo + i] = xc[xo + i] - xp[xo + i];
  }
  __syncthreads();
  // dx_new = dx with the three manifold segments through their blocks; P_ <- J P_ J^T block by block, in the reference's order
  if (tid >= 64 && tid < 64 + n) {
    const int i = tid - 64;
    double v = dx[i];
    if (i >= 3 && i < 9) {
      const int idx = i < 6 ? 3 : 6;
      const double* J = Jb + (i < 6 ? 0 : 9);
      const int r = i - idx;
      v = J[r * 3 + 0] * dx[idx] + J[r * 3 + 1] * dx[idx + 1] + J[r * 3 + 2] * dx[idx + 2];
    } else if (i >= 21) {
      const double* J = Jb + 18;
      const int r = i - 21;
      v = J[r * 2 + 0] * dx[21] + J[r * 2 + 1] * dx[22];
    }
    dxn[i] = v;
  }
  ik_left_block<3>(P_, n, 3, Jb, n, tid);
  __syncthreads();
  ik_right_block_T<3>(P_, n, 3, Jb, n, tid);
  __syncthreads();
  ik_left_block<3>(P_, n, 6, Jb + 9, n, tid);
  __syncthreads();
  ik_right_block_T<3>(P_, n, 6, Jb + 9, n, tid);
  __syncthreads();
  ik_left_block<2>(P_, n, 21, Jb + 18, n, tid);
  __syncthreads();
  ik_right_block_T<2>(P_, n, 21, Jb + 18, n, tid);
  __syncthreads();
  for (int e = tid; e < n * 12; e += 256) PR[e] = P_[(e / 12) * n + (e % 12)] / R;
  __syncthreads();
}

// ---- the measurement-dependent half (:1722-1764): gain through the matrix-inversion lemma, step, boxplus, convergence.
//      In (shared memory): live[91] sums, PR, dxn, xc, lim.  Out: HTH, HTh, KX, dxu, xn; s_i[18] = last iteration, s_i[21] = t.
//      Returns (block-uniform) false when H^T H needs the eigen-decomposition or the solve met a zero pivot: nothing is updated. ----
__device__ __forceinline__ bool ik_post_block(double* lds, int* s_i, double D, int it, int t_in, int max_iter, int tid) {
  const int n = IK_N;
  double* PR = lds + IKL_PR;
  double* W = lds + IKL_W;
  double* KX = lds + IKL_KX;
  double* HTH = lds + IKL_HTH;
  double* T = lds + IKL_T;
  double* X = lds + IKL_X;
  double* HTh = lds + IKL_HTh;
  double* dxn = lds + IKL_DXN;
  double* Kh = lds + IKL_KH;
  double* dxu = lds + IKL_DXU;
  double* xn = lds + IKL_XN;
  const double* xc = lds + IKL_XC;
  const double* lim = lds + IKL_LIM;
  const double* live = lds + IKL_LIVE;
  // the pass's sums: upper triangle -> full H^T H, H^T h
  if (tid < 144) {
    const int i = tid / 12, j = tid % 12;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    HTH[tid] = live[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];       // row lo of the triangle starts at 12 lo - lo (lo - 1) / 2
  } else if (tid >= 160 && tid < 172) {
    HTh[tid - 160] = live[78 + tid - 160];
  }
  __syncthreads();
  // T = H^T H PR[0:12] + I
  if (tid < 144) {
    const int i = tid / 12, j = tid % 12;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) acc += HTH[i * 12 + k] * PR[k * 12 + j];
    T[tid] = acc + (i == j ? 1.0 : 0.0);
  }
  // degeneracy (:1736-1744): when H^T H[0:6,0:6] - D I is positive definite every eigenvalue is >= D and the projector is the
  // identity (the host's shortcut); otherwise the host does the eigen-decomposition.  One lane of another wave, beside the solve.
  if (tid == 192) {
    double Lc[21];                                            // lower triangle, row-major packed
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double s = 0.5 * (HTH[i * 12 + j] + HTH[j * 12 + i]) - (i == j ? D : 0.0);
#pragma unroll
        for (int k = 0; k < j; k++) s -= Lc[i * (i + 1) / 2 + k] * Lc[j * (j + 1) / 2 + k];
        if (i == j) {
          if (!(s > 1e-9 * D)) ok = false;
          Lc[i * (i + 1) / 2 + i] = sqrt(ok ? s : 1.0);
        } else {
          Lc[i * (i + 1) / 2 + j] = s / Lc[j * (j + 1) / 2 + j];
        }
      }
    s_i[19] = ok ? 0 : 1;
  }
  __syncthreads();
  // S = T^-1: Gauss-Jordan in the registers of wave 0
  if (tid < 64) {
    const bool ok = ik_gj12_wave(T, X, tid);
    if (tid == 0) s_i[17] = ok ? 0 : 1;
  }
  __syncthreads();
  if (s_i[17] != 0 || s_i[19] != 0) return false;
  for (int e = tid; e < n * 12; e += 256) {                   // W = PR S
    const int i = e / 12, j = e % 12;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) acc += PR[i * 12 + k] * X[k * 12 + j];
    W[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < n * 12 + n; e += 256) {
    if (e < n * 12) {                                         // K_x[:, 0:12] = W H^T H
      const int i = e / 12, j = e % 12;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 12; k++) acc += W[i * 12 + k] * HTH[k * 12 + j];
      KX[e] = acc;
    } else {                                                  // K_h = W H^T h
      const int i = e - n * 12;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 12; k++) s += W[i * 12 + k] * HTh[k];
      Kh[i] = s;
    }
  }
  __syncthreads();
  if (tid < n) {                                              // dx_ = K_h + (K_x - I) dx_new   (:1733)
    const int i = tid;
    double s = 0;
#pragma unroll
    for (int k = 0; k < n; k++) s += ((k < 12 ? KX[i * 12 + k] : 0.0) - (i == k ? 1.0 : 0.0)) * dxn[k];
    dxu[i] = Kh[i] + s;
  }
  __syncthreads();
  // ---- x boxplus dx_ (:1747), convergence (:1757-1764) ----
  if (tid < 2) {
    const int o = (tid == 0) ? 3 : 7, idx = (tid == 0) ? 3 : 6;
    const Q4 a{xc[o], xc[o + 1], xc[o + 2], xc[o + 3]};
    const double v[3] = {dxu[idx], dxu[idx + 1], dxu[idx + 2]};
    const Q4 r = ik_qmul(a, ik_exp_quat(v, 0.5));              // SO3::boxplus: exp(v, scale / 2)
    xn[o] = r.x; xn[o + 1] = r.y; xn[o + 2] = r.z; xn[o + 3] = r.w;
  } else if (tid == 64) {
    double B[6], Bu[3], Rm[9], g[3];
    ik_s2_Bx(xc + 23, B);
    for (int i = 0; i < 3; i++) Bu[i] = B[i * 2 + 0] * dxu[21] + B[i * 2 + 1] * dxu[22];
    ik_q2r(ik_exp_quat(Bu, 0.5), Rm);
    ik_mv3(Rm, xc + 23, g);
    for (int i = 0; i < 3; i++) xn[23 + i] = g[i];
  } else if (tid >= 128 && tid < 128 + 15) {
    const int e = tid - 128, seg = e / 3, i = e % 3;         // pos, offset_T_L_I, vel, bg, ba
    const int xo = seg == 0 ? 0 : 8 + 3 * seg, dxo = seg == 0 ? 0 : 6 + 3 * seg;
    xn[xo + i] = xc[xo + i] + dxu[dxo + i];
  } else if (tid >= 192 && tid < 256) {
    const int l = tid - 192;
    const bool over = l < n && fabs(dxu[l]) > lim[l];
    const bool conv = __ballot(over) == 0ull;
    if (l == 0) {
      const int t = t_in + (conv ? 1 : 0);
      s_i[18] = (t > 1 || it == max_iter - 1) ? 1 : 0;        // last iteration?
      s_i[21] = t;
    }
  }
  __syncthreads();
  return true;
}

// The next pass's float32 constants from the new state (Objects/State.cpp:38-55,136-172, Localizer.cpp:554-555), three independent
// pieces on three waves, written straight to the device filter's head
__device__ __forceinline__ void ik_store_pose(const double* xn, ChainHead* H, int tid) {
  if (tid == 0) {
    const float p[3] = {(float)xn[0], (float)xn[1], (float)xn[2]};
    const float q[4] = {(float)xn[3], (float)xn[4], (float)xn[5], (float)xn[6]};
    float T[16];
    se3_from(q, p, T);
#pragma unroll
    for (int i = 0; i < 16; i++) H->pose.RT[i] = T[i];
  } else if (tid == 64 || tid == 65) {
    const int o = tid == 64 ? 3 : 7, po = tid == 64 ? 0 : 11;    // (rot, pos) / (offset_R_L_I, offset_T_L_I)
    const float p[3] = {(float)xn[po], (float)xn[po + 1], (float)xn[po + 2]};
    const float q[4] = {(float)xn[o], (float)xn[o + 1], (float)xn[o + 2], (float)xn[o + 3]};
    float T[16];
    se3_inv_from(q, p, T);
    float* dst = tid == 64 ? H->pose.RT_inv : H->pose.TLI_inv;
#pragma unroll
    for (int i = 0; i < 16; i++) dst[i] = T[i];
  } else if (tid == 128 || tid == 129) {
    const int o = tid == 128 ? 3 : 7;
    const double qc[4] = {-xn[o], -xn[o + 1], -xn[o + 2], xn[o + 3]};
    double Rd[9];
    quat_to_rot_d(qc, Rd);
    float* dst = tid == 128 ? H->pose.R_inv : H->pose.RLI_inv;
#pragma unroll
    for (int i = 0; i < 9; i++) dst[i] = (float)Rd[i];
  }
}

__global__ __launch_bounds__(256) void ieskf_kernel(ChainState* __restrict__ S, const double2* __restrict__ gran, unsigned long long seq,
                                                    const ChainPrior* __restrict__ prior, double2* __restrict__ res,
                                                    double2* __restrict__ logp, unsigned long long tag) {
  __shared__ double lds[IKL_END];
  __shared__ int s_i[32];                          // [17] singular, [18] last, [19] degenerate, [20..23] it, t, passes, max_iter
  const int tid = threadIdx.x;
  const int n = IK_N;
  if (!prior && S->head.status != 0) return;       // the chain ended in an earlier iteration
  IK_STAMP(0);
  double* P_ = lds + IKL_P;
  double* Lm = lds + IKL_L;
  double* KX = lds + IKL_KX;
  double* HTH = lds + IKL_HTH;
  double* HTh = lds + IKL_HTh;
  double* dxu = lds + IKL_DXU;
  double* Jb = lds + IKL_J;
  double* xn = lds + IKL_XN;
  double* xc = lds + IKL_XC;
  double* xp = lds + IKL_XP;
  double* lim = lds + IKL_LIM;
  double* live = lds + IKL_LIVE;

  // ---- everything this iteration reads, in one round trip: the pass's sums (groups added in slot order, as the host adds them),
  //      the state, the propagated state and covariance ----
  double R, D;
  bool pass_ok = true;
  if (tid < FIT_LIVE + 2) {
    // granule k of group g: {sum, pass number}; granules FIT_LIVE / FIT_LIVE + 1 of group 0: stragglers, ties
    const int groups = tid < FIT_LIVE ? FIT_GROUPS : 1;
    double2 g[FIT_GROUPS];
#pragma unroll
    for (int q = 0; q < FIT_GROUPS; q++) g[q] = gran[(size_t)(q < groups ? q : 0) * FIT_LIVE_PAD + tid];
    double r = g[0].x;
    pass_ok = (unsigned long long)__double_as_longlong(g[0].y) == seq;
#pragma unroll
    for (int q = 1; q < FIT_GROUPS; q++)
      if (q < groups) { r += g[q].x; pass_ok = pass_ok && (unsigned long long)__double_as_longlong(g[q].y) == seq; }
    live[tid] = r;
  }
  if (prior) {
    for (int i = tid; i < 529; i += 256) { const double v = prior->P[i]; P_[i] = v; S->P_prop[i] = v; }
    if (tid >= 128 && tid < 128 + 26) { const double v = prior->x[tid - 128]; xc[tid - 128] = v; xp[tid - 128] = v; S->x_prop[tid - 128] = v; }
    if (tid >= 160 && tid < 160 + n) { const double v = prior->limit[tid - 160]; lim[tid - 160] = v; S->limit[tid - 160] = v; }
    R = prior->R; D = prior->D;
    if (tid == 0) { s_i[20] = -1; s_i[21] = 0; s_i[22] = 0; s_i[23] = prior->max_iter; S->R = R; S->D = D; S->max_iter = prior->max_iter; }
  } else {
    for (int i = tid; i < 529; i += 256) P_[i] = S->P_prop[i];                    // :1655
    if (tid >= 128 && tid < 128 + 26) { xc[tid - 128] = S->x[tid - 128]; xp[tid - 128] = S->x_prop[tid - 128]; }
    if (tid >= 160 && tid < 160 + n) lim[tid - 160] = S->limit[tid - 160];
    R = S->R; D = S->D;
    if (tid == 0) { s_i[20] = S->it; s_i[21] = S->t; s_i[22] = S->passes; s_i[23] = S->max_iter; }
  }
  const int all_ok = __syncthreads_and(pass_ok ? 1 : 0);
  IK_STAMP(1);
  const int M = (int)llrint(live[FIT_LIVE - 1]);
  const int n_strag = (int)llrint(live[FIT_LIVE]), n_ties = (int)llrint(live[FIT_LIVE + 1]);
  const int it = s_i[20], t_in = s_i[21], passes = s_i[22], max_iter = s_i[23];

  // ---- the end of the chain, wherever it happens: head of the result + state words ----
  auto finish = [&](int status, int bail, int it_next, int t_next, int passes_done, const double* x_now, const double* x_meas) {
    // (block-uniform call; x_now / x_meas in shared memory)
    if (tid < 26) { put_granule(res, CH_X + tid, x_now[tid], tag); S->x[tid] = x_now[tid]; }
    if (tid >= 32 && tid < 32 + 26) put_granule(res, CH_XMEAS + tid - 32, x_meas[tid - 32], tag);
    if (tid >= 64 && tid < 64 + 3 * CH_MAX_PASSES) {
      const int k = tid - 64, p = k / 3;
      double v = 0.0;
      if (p < passes) v = S->info[k];
      else if (p == passes) v = (k % 3 == 0) ? (double)M : (k % 3 == 1 ? (double)n_strag : (double)n_ties);
      put_granule(res, CH_PASSINFO + k, v, tag);
    }
    if (tid == 128) {
      put_granule(res, CH_BAIL, (double)bail, tag); put_granule(res, CH_PASSES, (double)passes_done, tag);
      put_granule(res, CH_IT, (double)it_next, tag); put_granule(res, CH_T, (double)t_next, tag);
      put_granule(res, CH_STATUS, (double)status, tag);
      S->head.status = status; S->it = it_next; S->t = t_next; S->passes = passes_done;
    }
  };

  // ---- branches the host filter takes over: a pass that did not publish, M < 23, exact distance ties ----
  if (!all_ok || M < n || n_ties > 0) {
    finish(2, !all_ok ? 4 : (M < n ? 1 : 2), it, t_in, passes, xc, xc);
    return;
  }
  ik_pre_block(lds, R, tid);
  IK_STAMP(3);
  if (!ik_post_block(lds, s_i, D, it, t_in, max_iter, tid)) {
    finish(2, 3, it, t_in, passes, xc, xc);
    return;
  }
  IK_STAMP(8);
  const bool last = s_i[18] != 0;
  const int t_out = s_i[21];
  // per-pass bookkeeping + the optional log
  if (tid == 200 && passes < CH_MAX_PASSES) { S->info[3 * passes] = (double)M; S->info[3 * passes + 1] = (double)n_strag; S->info[3 * passes + 2] = (double)n_ties; }
  if (logp && passes < CH_MAX_PASSES) {
    double2* lg = logp + (size_t)passes * CH_LOGN;
    if (tid < 144) put_granule(lg, tid, HTH[tid], tag);
    if (tid >= 144 && tid < 156) put_granule(lg, tid, HTh[tid - 144], tag);
    if (tid >= 160 && tid < 160 + n) put_granule(lg, 156 + tid - 160, dxu[tid - 160], tag);
    if (tid >= 192 && tid < 192 + 26) put_granule(lg, 179 + tid - 192, xn[tid - 192], tag);
  }
  if (!last) {
    // the next pass's constants and the bound's reference pose (the pose the pass just done ran with)
    if (tid >= 224 && tid < 240) {
      const float* src = prior ? prior->RT0 : S->head.pose.RT;
      S->head.prev_RT[tid - 224] = src[tid - 224];
    }
    __syncthreads();                                              // (pose.RT is read above before it is overwritten)
    ik_store_pose(xn, &S->head, tid);
    if (tid == 32) { S->head.status = 0; S->it = it + 1; S->t = t_out; S->passes = passes + 1; }
    if (tid >= 96 && tid < 96 + 26) S->x[tid - 96] = xn[tid - 96];
    IK_STAMP(10);
    return;
  }
  // ---- covariance (:1766-1820) ----
  IK_STAMP(9);
  if (tid < 2) {
    const int idx = (tid == 0) ? 3 : 6;
    const double v[3] = {dxu[idx], dxu[idx + 1], dxu[idx + 2]};
    ik_A_T(v, Jb + 9 * tid);
  } else if (tid == 64) {
    const double d[2] = {dxu[21], dxu[22]};
    ik_s2_J(xn + 23, xp + 23, d, Jb + 18);
  }
  for (int i = tid; i < 529; i += 256) Lm[i] = P_[i];
  __syncthreads();
  IK_DUMP(4, P_, 529); IK_DUMP(5, Jb, 22);
  for (int s = 0; s < 2; s++) {
    const int idx = s == 0 ? 3 : 6;
    const double* J = Jb + 9 * s;
    if (tid < n) {
      const int c = tid;
      for (int r = 0; r < 3; r++) Lm[(idx + r) * n + c] = J[r * 3 + 0] * P_[idx * n + c] + J[r * 3 + 1] * P_[(idx + 1) * n + c] + J[r * 3 + 2] * P_[(idx + 2) * n + c];
    }
    ik_left_block<3>(KX, 12, idx, J, 12, tid - 32);             // threads 32..43
    __syncthreads();
    ik_right_block_T<3>(Lm, n, idx, J, n, tid);
    ik_right_block_T<3>(P_, n, idx, J, n, tid - 32);            // threads 32..54
    __syncthreads();
  }
  {
    const double* J = Jb + 18;
    if (tid < n) {
      const int c = tid;
      for (int r = 0; r < 2; r++) Lm[(21 + r) * n + c] = J[r * 2 + 0] * P_[21 * n + c] + J[r * 2 + 1] * P_[22 * n + c];
    }
    ik_left_block<2>(KX, 12, 21, J, 12, tid - 32);
    __syncthreads();
    ik_right_block_T<2>(Lm, n, 21, J, n, tid);
    ik_right_block_T<2>(P_, n, 21, J, n, tid - 32);
    __syncthreads();
  }
  IK_DUMP(0, Lm, 529); IK_DUMP(1, P_, 529); IK_DUMP(2, KX, 276); IK_DUMP(3, Jb, 22);
  for (int e = tid; e < 529; e += 256) {
    const int i = e / n, j = e % n;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) acc += KX[i * 12 + k] * P_[k * n + j];
    put_granule(res, CH_P + e, Lm[e] - acc, tag);
  }
  finish(1, 0, it + 1, t_out, passes + 1, xn, xc);
  IK_STAMP(11);
}

void launch_ieskf(hipStream_t st, ChainState* S, const void* gran, unsigned long long seq, const ChainPrior* prior, void* res, void* log,
                  unsigned long long tag, hipEvent_t e0, hipEvent_t e1) {
  hipExtLaunchKernelGGL(ieskf_kernel, dim3(1), dim3(256), 0, st, e0, e1, 0, S, (const double2*)gran, seq, prior, (double2*)res,
                        (double2*)log, tag);
}

}  // namespace flimo

// Strip walk for the batched Chebyshev term (BASELINE configs[4]: panel X[i * b + s], lane = state) on a lattice
// operator H = H_a (x) 1 + 1 (x) H_c: near distances +-d_i (d < 64) and far distances +-m g, m = 1..K.
//
// The wave-per-row kernels (kernels_spmm.hip: spmm_rows_smem_kernel) send 16 gathered rows of X (16 KiB) per row through
// the compute unit's L1; what bounds them is that path, not HBM (profiles/r02/batched_c5_pmc_diag.txt: texture
// addresser busy 87 %, 90 L1 -> L2 requests per row, 64 of them the eight far gathers that no neighbouring row shares).
// Here a wavefront walks down one inner index c -- rows c + a g, a = a0, a0 + 1, ... -- and keeps the far rows of X in a
// REGISTER ring (x[r + m g], m = -K..K: nine double2 per lane at K = 4; one new row per step), so the far gathers never
// touch memory again; the near rows are the rows its neighbours in the workgroup (consecutive c) stream at the same
// step and come out of the L1; the row's matrix entries arrive as one 256-byte vector load (one entry per lane,
// broadcast with v_readlane); and everything a step needs from memory is requested one step ahead.  No LDS, so the
// occupancy stays at that of the kernels it replaces (the LDS-ring form of round 2 lost there: 16 rows in flight per
// CU instead of 32).  Rows whose +-K g neighbours wrap around, and anything that is not this lattice shape, stay with
// the wave-per-row kernel (an `order` list of those rows, second launch).
//
// Sums per (row, state) in the order of csr_spmm_kernel -- entries in storage (column) order alternating between two
// partial sums -- hence bit-identical to the other batched kernels (tests/test_gpu_parity.py).
#include <type_traits>

#include "kernel_common.h"

namespace qp {

constexpr int kSpmmWalkWaves = 8;

__device__ __forceinline__ double readlane_d(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int NN>
struct SpmmStep {
  double2 nlo[NN], nhi[NN];   // X[(r - d_i) b + s], X[(r + d_i) b + s]
  double2 xnew;               // X[(r + K g) b + s]
  double2 v0, acc;            // row-local operands of the epilogue
  double2 mv;                 // lane k < z: entry k of the row
};

// entries of a row in storage order: [far -K..-1] [near -d_NN..-d_1] [diagonal] [near d_1..d_NN] [far 1..K]
template <class Op, int NN, int K, int DIAG>
__global__ __launch_bounds__(64 * kSpmmWalkWaves) void spmm_walk_kernel(const double2* __restrict__ vals,
                                                                         const double2* __restrict__ X, SpmmWalkPlan P,
                                                                         int L, int nseg, int b, Op op) {
  constexpr int Z = 2 * (NN + K) + DIAG;
  static_assert(Z % 8 == 0 || Z == 2 * (NN + K) + DIAG, "");
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t task = (int64_t)wg * kSpmmWalkWaves + wave;
  const int64_t g = P.g;
  const int seg = (int)(task / g);
  const int64_t c = task - (int64_t)seg * g;
  if (seg >= nseg) return;
  const int a0 = P.a_lo + seg * L, a1 = min(a0 + L, P.a_hi);
  if (a0 >= a1) return;
  const int st = blockIdx.y * 64 + lane;
  const bool active = st < b;
  const int stc = active ? st : b - 1;
  const double2* __restrict__ Xs = X + stc;
  typename std::remove_const<Op>::type opl = op;
  opl.e.mirror = nullptr;
  opl.e.check_partials = nullptr;
  const double2* __restrict__ v0p = op.e.v0;
  const double2* __restrict__ accp = op.e.acc_in;
  const int64_t gb = g * (int64_t)b;
  int64_t row = (int64_t)a0 * g + c;
  auto load_step = [&](int64_t r, SpmmStep<NN>& w) __attribute__((always_inline)) {
    const int64_t e = r * (int64_t)b + stc;
#pragma unroll
    for (int i = 0; i < NN; ++i) {
      w.nlo[i] = Xs[(r - P.near[i]) * (int64_t)b];
      w.nhi[i] = Xs[(r + P.near[i]) * (int64_t)b];
    }
    w.xnew = ld_stream<Op::kStream>(Xs + (r + (int64_t)K * g) * (int64_t)b);
    w.v0 = ld_stream<Op::kStream>(v0p ? v0p + e : X + lane);
    w.acc = ld_stream<Op::kStream>(accp ? accp + e : X + lane);
    w.mv = ld_val<Op::kStream>(vals + r * (int64_t)Z + min(lane, Z - 1));
  };
  double2 xr[2 * K + 1];
#pragma unroll
  for (int m = -K; m < K; ++m) xr[K + m] = Xs[(row + m * g) * (int64_t)b];
  SpmmStep<NN> wa, wb;
  load_step(row, wa);
  auto step = [&](const SpmmStep<NN>& cu, SpmmStep<NN>& nx, auto has_next) __attribute__((always_inline)) {
    xr[2 * K] = cu.xnew;
    if constexpr (decltype(has_next)::value) load_step(row + g, nx);
    double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < Z; ++k) {
      const double2 a = make_double2(readlane_d(cu.mv.x, k), readlane_d(cu.mv.y, k));
      double2 xv;
      if (k < K) xv = xr[k];                                        // far -(K - k) g
      else if (k < K + NN) xv = cu.nlo[NN - 1 - (k - K)];           // near -d
      else if (DIAG && k == K + NN) xv = xr[K];
      else if (k < K + 2 * NN + DIAG) xv = cu.nhi[k - K - NN - DIAG];
      else xv = xr[K + 1 + (k - K - 2 * NN - DIAG)];               // far +m g
      // csr_spmm_kernel's order: groups of eight (then four) alternate between the two sums, a remainder joins the first
      const bool second = (k < (Z & ~7)) ? (k & 1) : ((k < (Z & ~3)) ? (k & 1) : false);
      if (second) cfma(s1, a, xv);
      else cfma(s0, a, xv);
    }
    typename Op::Pre pre;
    pre.xi = xr[K];
    pre.v0 = v0p ? cu.v0 : make_double2(0.0, 0.0);
    pre.acc = accp ? cu.acc : make_double2(0.0, 0.0);
    double2 chk = make_double2(0.0, 0.0);
    double nrm = 0.0;
    const int64_t e = row * (int64_t)b + stc;
    if (active) opl.row(e, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, e);
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) xr[i] = xr[i + 1];
    row += g;
  };
  int n = a1 - a0;
  for (; n > 2; n -= 2) {
    step(wa, wb, std::true_type());
    step(wb, wa, std::true_type());
  }
  if (n == 2) {
    step(wa, wb, std::true_type());
    step(wb, wa, std::false_type());
  } else {
    step(wa, wb, std::false_type());
  }
  (void)gb;
}

template <class Op, int NN, int K, int DIAG>
static void launch_inst(hipStream_t s, dim3 grid, const double2* vals, const double2* X, const SpmmWalkPlan& P, int L,
                        int nseg, int b, const Op& op) {
  hipLaunchKernelGGL((spmm_walk_kernel<Op, NN, K, DIAG>), grid, dim3(64 * kSpmmWalkWaves), 0, s, vals, X, P, L, nseg, b, op);
}

template <class Op>
static bool launch_shape(hipStream_t s, dim3 grid, const double2* vals, const double2* X, const SpmmWalkPlan& P, int L,
                         int nseg, int b, const Op& op) {
  switch (P.nn * 100 + P.K * 10 + P.diag) {
    case 440: launch_inst<Op, 4, 4, 0>(s, grid, vals, X, P, L, nseg, b, op); return true;
    case 441: launch_inst<Op, 4, 4, 1>(s, grid, vals, X, P, L, nseg, b, op); return true;
    case 220: launch_inst<Op, 2, 2, 0>(s, grid, vals, X, P, L, nseg, b, op); return true;
    case 221: launch_inst<Op, 2, 2, 1>(s, grid, vals, X, P, L, nseg, b, op); return true;
    default: return false;
  }
}

int launch_spmm_walk_cheby(hipStream_t s, const double2* vals, const double2* X, const SpmmWalkPlan& P, int b,
                           const ChebyEpi& e, const Tuning& tun, bool nt, bool* launched) {
  *launched = false;
  if (!P.valid) return QP_OK;
  const int steps = P.a_hi - P.a_lo;
  if (steps < 8) return QP_OK;
  const int64_t target = tun.spmm_walk_waves > 0 ? tun.spmm_walk_waves : 3072;
  int nseg = (int)std::max<int64_t>(1, (target + P.g / 2) / P.g);
  nseg = std::min(nseg, std::max(1, steps / 8));
  const int L = (steps + nseg - 1) / nseg;
  nseg = (steps + L - 1) / L;
  const int64_t ntask = (int64_t)nseg * P.g;
  const dim3 grid((unsigned)((ntask + kSpmmWalkWaves - 1) / kSpmmWalkWaves), (unsigned)((b + 63) / 64));
  bool ok;
  if (nt) {
    ChebyOpT<true> op{e};
    ok = launch_shape(s, grid, vals, X, P, L, nseg, b, op);
  } else {
    ChebyOp op{e};
    ok = launch_shape(s, grid, vals, X, P, L, nseg, b, op);
  }
  if (!ok) return QP_OK;
  QP_HIP(hipGetLastError());
  *launched = true;
  return QP_OK;
}

}  // namespace qp

// Strip-walk kernel for the fused Chebyshev term of a Hermitian-packed LATTICE operator (src/cheby.jl:171-211 with the
// mat-vec of src/generators.jl:634-645 inside), gfx950 / wave64.
//
// Why: the per-block kernel (kernels.hip: hrb_spmv_kernel) sends 34 KiB per 64-row block through the compute unit's
// vector L1 -- 8 KiB of upper values, 8 KiB of conj-transposed values and 16 KiB of gathered vector elements, of which
// only the first 8 KiB and two of the vector streams come from HBM.  Beyond the Infinity Cache the L1's miss queue, not
// HBM, is what is full (profiles/r03/hrb_n22_pmc_diag.txt: texture addresser busy 85 %, the L1 stalled on its pending
// misses 58 % of the launch, 209 L1 -> L2 requests per block, 83 in flight per CU).  On a lattice all of that re-read
// data is data the SAME wavefront would load anyway if it walked down a strip column (see WalkPlan in device.h):
//   * x[r + m g]              = the row-local element of the block m steps away           -> register ring, 1 load/step
//   * conj H[r - m g, r]      = the far upper value streamed m steps ago                  -> register FIFO, 0 loads
//   * x[r +- d], conj H[r - d, r] (d <= 16) = lane shifts of the block's own element / near values
//                                                                                         -> per-wavefront LDS window
// so that a block costs 8 value loads + 3 vector loads + a few one-line halo loads: ~95 instead of 209 L1 -> L2
// requests, nearly all of them HBM streams, prefetched one step ahead.
//
// Summation order per row is that of the per-block kernel (lower slots then upper slots in storage order, two
// interleaved partial sums), so the two kernels agree bit for bit (tests/test_gpu_parity.py).
#include <type_traits>

#include "kernel_common.h"

namespace qp {

struct HrbArrays {   // what the per-block path of the edge blocks reads
  const int64_t* uptr;
  const int64_t* ucmeta;
  const char* ucolbytes;
  const int64_t* lptr;
  const int64_t* lcmeta;
  const char* lcolbytes;
  const int4* lpos4;
};

struct WalkGeom {
  int L = 0;           // steps per wavefront
  int nseg = 0;        // segments of L steps per strip column
  int n_edge_wg = 0;   // leading workgroups: eight edge blocks each, per-block code path
  int n_walk_wg = 0;
};

constexpr int kWalkWaves = 8;   // wavefronts (adjacent strip columns) per workgroup

// one row block by the per-block rules of hrb_spmv_kernel (loop form: same sums as its straight-line form)
template <class VT>
__device__ __forceinline__ void hrb_edge_block(const HrbArrays& H, const VT* __restrict__ uvals,
                                               const double2* __restrict__ x, int64_t b, int lane, int64_t nrows,
                                               const ChebyOp& op) {
  const int64_t ubase = H.uptr[b], lbase = H.lptr[b];
  const int nuq = (int)((H.uptr[b + 1] - ubase) >> 8);
  const int nlq = (int)((H.lptr[b + 1] - lbase) >> 8);
  const VT* __restrict__ v = uvals + ubase + lane;
  const int64_t ucm = H.ucmeta[b], lcm = H.lcmeta[b];
  const int4* __restrict__ lp4 = H.lpos4 + (lbase >> 2) + lane;
  const int64_t row = b * kRB + lane;
  const int64_t rowc = row < nrows ? row : nrows - 1;
  const ChebyOp::Pre pre = op.pre(rowc);
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
  if ((lcm & 3) == 2) {
    const LowerStencilSlot* __restrict__ ls = reinterpret_cast<const LowerStencilSlot*>(H.lcolbytes + (lcm >> 2));
#pragma unroll 2
    for (int k = 0; k < 4 * nlq; k += 2) {
      const LowerStencilSlot e0 = ls[k], e1 = ls[k + 1];
      const int c0 = (int)rowc + e0.delta, c1 = (int)rowc + e1.delta;
      const double2 a0 = ld_val<false>(uvals + (((c0 >> 6) == e0.cb0 ? e0.pb0 : e0.pb1) + (c0 & 63)));
      const double2 a1 = ld_val<false>(uvals + (((c1 >> 6) == e1.cb0 ? e1.pb0 : e1.pb1) + (c1 & 63)));
      const double2 x0 = x[c0];
      const double2 x1 = x[c1];
      cfma_conj(s0, a0, x0);
      cfma_conj(s1, a1, x1);
    }
  } else {
    for (int q = 0; q < nlq; ++q) {
      const int4 c = ld_cols<true>(H.lcolbytes, lcm, q, lane, (int)rowc);
      const int4 p = ld_col<true>(lp4 + (size_t)q * 64);
      const double2 a0 = ld_tr(uvals, p.x);
      const double2 a1 = ld_tr(uvals, p.y);
      const double2 a2 = ld_tr(uvals, p.z);
      const double2 a3 = ld_tr(uvals, p.w);
      const double2 x0 = x[c.x];
      const double2 x1 = x[c.y];
      const double2 x2 = x[c.z];
      const double2 x3 = x[c.w];
      cfma_conj(s0, a0, x0);
      cfma_conj(s1, a1, x1);
      cfma_conj(s0, a2, x2);
      cfma_conj(s1, a3, x3);
    }
  }
  for (int q = 0; q < nuq; ++q) {
    const int4 c = ld_cols<true>(H.ucolbytes, ucm, q, lane, (int)rowc);
    const double2 a0 = ld_val<false>(v + (size_t)(4 * q + 0) * 64);
    const double2 a1 = ld_val<false>(v + (size_t)(4 * q + 1) * 64);
    const double2 a2 = ld_val<false>(v + (size_t)(4 * q + 2) * 64);
    const double2 a3 = ld_val<false>(v + (size_t)(4 * q + 3) * 64);
    const double2 x0 = x[c.x];
    const double2 x1 = x[c.y];
    const double2 x2 = x[c.z];
    const double2 x3 = x[c.w];
    cfma(s0, a0, x0);
    cfma(s1, a1, x1);
    cfma(s0, a2, x2);
    cfma(s1, a3, x3);
  }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, 0);
}

// (by value: a conditional expression over two lvalues selects an ADDRESS and would force both objects into scratch)
__device__ __forceinline__ double2 sel2(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }

// what a step needs from memory (everything else is carried over from the steps before)
template <int NU>
struct WalkStep {
  double2 ua[NU];      // the block's upper values (pads skipped)
  double2 xnew;        // x[row + K g]
  double2 v0, acc;     // row-local operands of the epilogue
  // halos of the near windows, sixteen lanes each (lane = 16 q + t, only t < distance is used):
  double2 hx;          //   q = 0: x[r0 - dmax + t]         q = 1: x[r0 + 64 + t]
  double2 ha;          //   q = i: value (64 - d_i + t) of slot z0 + i of block b - 1
};

// LDS of one wavefront, in double2 elements: the near window of x (16 + 64 + 16), NN near value windows (16 + 64) and the
// FIFOs of the far upper values (slot m: m entries of 64)
template <int NN, int K>
struct WalkLds {
  static constexpr int XW = kRB + 2 * kWalkHalo, AW = kRB + kWalkHalo;
  static constexpr int kHist = XW + NN * AW;
  static constexpr int kPerWave = kHist + kRB * (K * (K + 1) / 2);
  static constexpr size_t kBytes = sizeof(double2) * (size_t)kPerWave * kWalkWaves;
};

template <class VT, int NN, int K, int Z0>
__global__ __launch_bounds__(64 * kWalkWaves) void hrb_walk_kernel(const VT* __restrict__ uvals,
                                                                    const double2* __restrict__ x, WalkPlan P,
                                                                    WalkGeom G, HrbArrays H, int64_t nrows, ChebyOp op) {
  constexpr int NL = NN + K;         // lower slots: [-K g .. -g] [-d_NN .. -d_1]
  constexpr int NU = Z0 + NN + K;    // upper slots that carry entries: [0] [d_1 .. d_NN] [g .. K g]
  using Lds = WalkLds<NN, K>;
  constexpr int XW = Lds::XW, AW = Lds::AW;
  static_assert(NL % 4 == 0, "a stencil lower section has no pad slots");
  static_assert(NN <= 4, "the near value halos share one register: sixteen lanes each");
  extern __shared__ double2 walk_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x < G.n_edge_wg) {
    const int64_t idx = (int64_t)blockIdx.x * kWalkWaves + wave;
    if (idx < P.n_edge) hrb_edge_block<VT>(H, uvals, x, (int64_t)P.edge_map[idx], lane, nrows, op);
    return;
  }
  const unsigned wg = xcd_remap(blockIdx.x - G.n_edge_wg, G.n_walk_wg);
  const int task = (int)wg * kWalkWaves + wave;
  const int S = P.S;
  const int seg = task / S, col = task - seg * S;
  const int64_t nW = P.R1 - P.W0;
  if (seg >= G.nseg || col >= nW) return;
  const int Jc = (int)((nW - col + S - 1) / S);
  const int j0 = seg * G.L, j1 = min(j0 + G.L, Jc);
  if (j0 >= j1) return;
  const int64_t g = (int64_t)kRB * S;
  const int dmax = P.near[NN - 1];
  double2* __restrict__ xwin = walk_lds + (size_t)wave * Lds::kPerWave;
  double2* __restrict__ hring = xwin + Lds::kHist;
  auto ubase = [&](int64_t blk) __attribute__((always_inline)) -> int64_t { return P.U0 + (blk - P.R0) * (int64_t)P.ustride; };
  // No branch inside the walk: a join in the control flow makes the compiler wait for EVERY outstanding load (the
  // prefetch included).  Operands that a term does not have are loaded from a line that stays in the L1 and ignored
  // by the epilogue; halo lanes beyond the halo repeat its last element.
  ChebyOp opl = op;
  opl.e.mirror = nullptr;           // (the launcher takes this kernel only without them)
  opl.e.check_partials = nullptr;
  const double2* __restrict__ v0p = op.e.v0;
  const double2* __restrict__ accp = op.e.acc_in;
  // halo lanes: group q = lane / 16, element t = lane % 16
  const int hq = lane >> 4, ht = lane & 15;
  const int hd = P.near[hq < NN ? hq : NN - 1];               // distance of this lane's near value halo
  const int hoff_x = (hq & 1) ? kRB + min(ht, dmax - 1) : -dmax + min(ht, dmax - 1);
  const int hoff_a = (Z0 + (hq < NN ? hq : NN - 1)) * kRB + kRB - hd + min(ht, hd - 1);
  auto load_step = [&](int64_t blk, WalkStep<NU>& w) __attribute__((always_inline)) {
    const int64_t r = blk * kRB + lane;
    const VT* __restrict__ v = uvals + ubase(blk) + lane;
#pragma unroll
    for (int u = 0; u < NU; ++u) w.ua[u] = ld_val<false>(v + (size_t)u * 64);
    w.xnew = x[r + K * g];
    w.v0 = *(v0p ? v0p + r : x + lane);
    w.acc = *(accp ? accp + r : x + lane);
    w.hx = x[blk * kRB + hoff_x];
    w.ha = ld_val<false>(uvals + ubase(blk - 1) + hoff_a);
  };
  // where this lane's halo elements go in the windows (lanes that carry none rewrite their own main element)
  const bool hx_on = hq < 2 && ht < dmax;
  const int hx_pos = hx_on ? ((hq & 1) ? kWalkHalo + kRB + ht : kWalkHalo - dmax + ht) : kWalkHalo + lane;
  const bool ha_on = hq < NN && ht < hd;
  const int ha_pos = XW + (hq < NN ? hq : 0) * AW + (ha_on ? kWalkHalo - hd + ht : kWalkHalo + lane);

  int64_t b = P.W0 + col + (int64_t)S * j0;
  int64_t row = b * kRB + lane;
  // the ring of gathered elements x[row + m g], m = -K .. K (the last one arrives with each step's loads) ...
  double2 xr[2 * K + 1];
#pragma unroll
  for (int m = -K; m < K; ++m) xr[K + m] = x[row + m * g];
  // ... and the far upper values of the K blocks behind, FIFO m in LDS: the value of t steps ago sits at entry
  // (step - t) mod m, so the entry read at a step (the value of m steps ago) is the one overwritten at that step
#pragma unroll
  for (int m = 1; m <= K; ++m)
#pragma unroll
    for (int a = 1; a <= m; ++a)
      hring[(m * (m - 1) / 2 + (m - a)) * kRB + lane] =
          ld_val<false>(uvals + ubase(b - (int64_t)a * S) + (size_t)(Z0 + NN + m - 1) * 64 + lane);
  int hpos[K];   // (wave-uniform) entry of FIFO m that this step reads and then overwrites: step mod m
#pragma unroll
  for (int m = 1; m <= K; ++m) hpos[m - 1] = 0;
  // Two register sets that swap roles every step: while the arithmetic of a block runs out of one, the next
  // block's streams land in the other (no copies, and the wait for them sits at their first use, a whole step later).
  WalkStep<NU> wa, wb;
  load_step(b, wa);
  auto step = [&](const WalkStep<NU>& cu, WalkStep<NU>& nx, auto has_next) __attribute__((always_inline)) {
    xr[2 * K] = cu.xnew;
    if constexpr (decltype(has_next)::value) load_step(b + S, nx);
    // ---- near windows through LDS: element e of the block's window sits at [kWalkHalo + e], e = -16 .. 79
    xwin[kWalkHalo + lane] = xr[K];
    xwin[hx_pos] = sel2(hx_on, cu.hx, xr[K]);
#pragma unroll
    for (int i = 0; i < NN; ++i) xwin[XW + i * AW + kWalkHalo + lane] = cu.ua[Z0 + i];
    {
      double2 own = cu.ua[Z0];
#pragma unroll
      for (int i = 1; i < NN; ++i) own = sel2(hq == i, cu.ua[Z0 + i], own);
      xwin[ha_pos] = sel2(ha_on, cu.ha, own);
    }
    // the lanes of this wavefront exchange data through its own window: order the writes before the reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every operand that comes out of LDS first (independent reads, one wait), then the arithmetic
    double2 fa[K], na[NN], nxl[NN], nxu[NN];
#pragma unroll
    for (int m = 1; m <= K; ++m) fa[m - 1] = hring[(m * (m - 1) / 2 + hpos[m - 1]) * kRB + lane];
#pragma unroll
    for (int i = 0; i < NN; ++i) {
      const int d = P.near[i];
      na[i] = xwin[XW + i * AW + kWalkHalo + lane - d];
      nxl[i] = xwin[kWalkHalo + lane - d];
      nxu[i] = xwin[kWalkHalo + lane + d];
    }
    double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
    // lower section, storage order: far -K g .. -g, then near -d_NN .. -d_1
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      double2 a, xv;
      if (l < K) {
        const int m = K - l;
        a = fa[m - 1];
        xv = xr[K - m];
      } else {
        const int i = NN - 1 - (l - K);
        a = na[i];
        xv = nxl[i];
      }
      if (l & 1) cfma_conj(s1, a, xv);
      else cfma_conj(s0, a, xv);
    }
    // upper section: the diagonal, near d_1 .. d_NN, far g .. K g
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      double2 xv;
      if (u < Z0) xv = xr[K];
      else if (u < Z0 + NN) xv = nxu[u - Z0];
      else xv = xr[K + (u - Z0 - NN + 1)];
      if (u & 1) cfma(s1, cu.ua[u], xv);
      else cfma(s0, cu.ua[u], xv);
    }
    ChebyOp::Pre pre;
    pre.xi = xr[K];
    pre.v0 = v0p ? cu.v0 : make_double2(0.0, 0.0);
    pre.acc = accp ? cu.acc : make_double2(0.0, 0.0);
    double2 chk = make_double2(0.0, 0.0);
    double nrm = 0.0;
    opl.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, 0);
    // ---- one step down the strip column
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this step's window / FIFO reads before the writes below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 1; m <= K; ++m) {
      hring[(m * (m - 1) / 2 + hpos[m - 1]) * kRB + lane] = cu.ua[Z0 + NN + m - 1];
      hpos[m - 1] = (hpos[m - 1] + 1 == m) ? 0 : hpos[m - 1] + 1;
    }
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) xr[i] = xr[i + 1];
    b += S;
    row += g;
  };
  int n = j1 - j0;
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
}

template <class VT, int NN, int K, int Z0>
static bool launch_instance(hipStream_t s, dim3 grid, const VT* uvals, const double2* x, const WalkPlan& P,
                            const WalkGeom& G, const HrbArrays& H, int64_t nrows, const ChebyOp& op) {
  constexpr size_t lds = WalkLds<NN, K>::kBytes;
  auto kern = &hrb_walk_kernel<VT, NN, K, Z0>;
  // more than the 64 KB a launch gets without asking: opt in once per kernel instance (and device)
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!attr_ok) return false;
  hipLaunchKernelGGL(kern, grid, dim3(64 * kWalkWaves), lds, s, uvals, x, P, G, H, nrows, op);
  return true;
}

template <class VT>
static bool launch_shape(hipStream_t s, dim3 grid, const VT* uvals, const double2* x, const WalkPlan& P,
                         const WalkGeom& G, const HrbArrays& H, int64_t nrows, const ChebyOp& op) {
  const int key = P.nn * 100 + P.K * 10 + P.z0;
  switch (key) {
    case 440: return launch_instance<VT, 4, 4, 0>(s, grid, uvals, x, P, G, H, nrows, op);
    case 441: return launch_instance<VT, 4, 4, 1>(s, grid, uvals, x, P, G, H, nrows, op);
    case 220: return launch_instance<VT, 2, 2, 0>(s, grid, uvals, x, P, G, H, nrows, op);
    case 221: return launch_instance<VT, 2, 2, 1>(s, grid, uvals, x, P, G, H, nrows, op);
    case 310: return launch_instance<VT, 3, 1, 0>(s, grid, uvals, x, P, G, H, nrows, op);
    case 130: return launch_instance<VT, 1, 3, 0>(s, grid, uvals, x, P, G, H, nrows, op);
    default: return false;
  }
}

bool walk_shape_supported(int nn, int K, int z0) {
  return (nn == 4 && K == 4 && z0 <= 1) || (nn == 2 && K == 2 && z0 <= 1) || (nn == 3 && K == 1 && z0 == 0) ||
         (nn == 1 && K == 3 && z0 == 0);
}

int launch_hrb_walk_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, const Tuning& tun,
                          bool* launched) {
  *launched = false;
  const WalkPlan* P = A.walk;
  if (!P || !P->valid || A.format != QP_FMT_HRB) return QP_OK;
  const int64_t nW = P->R1 - P->W0;
  if (nW < tun.walk_min_blocks || nW < P->S) return QP_OK;
  WalkGeom G;
  const int64_t J = (nW + P->S - 1) / P->S;                       // steps of the longest strip column
  const int64_t nseg_target = std::max<int64_t>(1, tun.walk_waves / P->S);
  G.L = (int)std::max<int64_t>(1, (J + nseg_target - 1) / nseg_target);
  G.nseg = (int)((J + G.L - 1) / G.L);
  const int64_t ntask = (int64_t)G.nseg * P->S;
  G.n_walk_wg = (int)((ntask + kWalkWaves - 1) / kWalkWaves);
  G.n_edge_wg = (int)((P->n_edge + kWalkWaves - 1) / kWalkWaves);
  HrbArrays H{A.bptr, A.cmeta, reinterpret_cast<const char*>(A.cols), A.lptr, A.lcmeta,
              reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos)};
  ChebyOp op{e};
  const dim3 grid((unsigned)(G.n_edge_wg + G.n_walk_wg));
  const bool ok = A.vals_r ? launch_shape<double>(s, grid, A.vals_r, x, *P, G, H, A.nrows, op)
                           : launch_shape<double2>(s, grid, A.vals, x, *P, G, H, A.nrows, op);
  if (!ok) return QP_OK;
  QP_HIP(hipGetLastError());
  *launched = true;
  return QP_OK;
}

}  // namespace qp

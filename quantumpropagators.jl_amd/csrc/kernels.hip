// HIP kernels of the prop_step! hot path, written for CDNA4 / gfx950 (wave64).
//
//  K2/K3/K4  fused Chebyshev term  = SpMV + shift/scale + three-term recurrence + axpy
//            (+ final phase)                    src/cheby.jl:171-211
//  K7        plain SpMV   y = beta y + alpha H x       src/generators.jl:634-645
//  K8/K9     fused "axpy -> dot" modified Gram-Schmidt pass, norm + scale
//                                                       src/arnoldi.jl:82-96
//  K11/K12   tall-skinny combine  out = s0 out + sum_i coef_i q_i (+ |out|^2)
//                                                       src/newton.jl:346-367
//  K6        operator value planes  vals = sum_l c_l plane_l   src/generators.jl:757-766
//
// All kernels are HBM-bandwidth bound (AI < 0.4 flop/B); the design rules are
// 16-B-per-lane fully coalesced streams, no atomics (bitwise run-to-run determinism is
// required by check_propagator's reinit test), reductions finished in the *next*
// kernel's prologue instead of an extra launch or an in-launch fence.
#include <cstring>
#include <type_traits>

#include "kernel_common.h"

namespace qp {


// VAR bit 0: nt matrix loads; bit 1: row-local operands prefetched before the loop;
// bit 2: unroll 4 quads (16 value loads in flight per lane) instead of 2
template <class Op, int VAR, class VT, int WS = kThreads / 64>   // WS wavefronts (row blocks) per workgroup
__global__ __launch_bounds__(64 * WS) void rbcsr_spmv_kernel(const int64_t* __restrict__ bptr,
                                                              const int64_t* __restrict__ cmeta,
                                                              const char* __restrict__ colbytes,
                                                              const VT* __restrict__ vals,
                                                              const double2* __restrict__ x,
                                                              int64_t nblocks, int64_t nrows, Op op,
                                                              const int32_t* __restrict__ block_map, SyncArgs sy) {
  constexpr bool NT = (VAR & 1) != 0;
  constexpr bool PRE = (VAR & 2) != 0;
  constexpr int UNR = (VAR & 4) ? 4 : 2;
  static_assert(WS == kThreads / 64 || std::is_same<Op, ChebyOp>::value,
                "block_sum (Op::begin of the folded norm, finish_check) sums kThreads / 64 wavefronts: only the fused term, "
                "launched without its per-workgroup check, may run with another workgroup width");
  __shared__ double2 lds[WS];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  sync_wait(sy, wg);
  op.begin_issue();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)wg * WS + wave;  // position in the row set
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  int64_t row = nrows;
  typename Op::Pre pre;
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
  if (idx < nblocks) {
    const int64_t b = block_map ? (int64_t)block_map[idx] : idx;
    const int64_t base = bptr[b];
    const int nq = (int)((bptr[b + 1] - base) >> 8);  // width / 4
    const VT* __restrict__ v = vals + base + lane;
    const int64_t cm = cmeta[b];
    row = b * kRB + lane;
    const int64_t rowc = row < nrows ? row : nrows - 1;
    if (PRE) pre = op.pre(rowc);
#pragma unroll UNR
    for (int q = 0; q < nq; ++q) {
      const int4 c = ld_cols<NT>(colbytes, cm, q, lane, (int)rowc);
      const double2 a0 = ld_val<NT>(v + (size_t)(4 * q + 0) * 64);
      const double2 a1 = ld_val<NT>(v + (size_t)(4 * q + 1) * 64);
      const double2 a2 = ld_val<NT>(v + (size_t)(4 * q + 2) * 64);
      const double2 a3 = ld_val<NT>(v + (size_t)(4 * q + 3) * 64);
      const double2 x0 = x[c.x];
      const double2 x1 = x[c.y];
      const double2 x2 = x[c.z];
      const double2 x3 = x[c.w];
      cfma(s0, a0, x0);
      cfma(s1, a1, x1);
      cfma(s0, a2, x2);
      cfma(s1, a3, x3);
    }
    if (!PRE) pre = op.pre(rowc);
  }
  op.begin(lds);
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, idx * kRB + lane);
  finish_check(op, chk, nrm, lds);
  sync_signal(sy);
}

// ---------------------------------------------------------------------------
// HRB SpMV (Hermitian-packed row blocks).  Upper section exactly as RBCSR.  For a lower
// entry (r, c), c < r, the lane loads the value stored for (c, r) -- a line that the wave
// owning row c streamed shortly before on the same XCD, i.e. an L2 hit -- and uses its
// complex conjugate.  HBM sees 20 B per upper entry but only 8 B (column + position) per
// lower entry.  The upper value loads keep the default cache policy (they are re-read
// through L2); the lower index streams are read-once.
// ---------------------------------------------------------------------------
template <class Op, int VAR, class VT, int WS = kThreads / 64>   // WS wavefronts (row blocks) per workgroup
__global__ __launch_bounds__(64 * WS) void hrb_spmv_kernel(const int64_t* __restrict__ uptr,
                                                            const int64_t* __restrict__ ucmeta,
                                                            const char* __restrict__ ucolbytes,
                                                            const VT* __restrict__ uvals,
                                                            const int64_t* __restrict__ lptr,
                                                            const int64_t* __restrict__ lcmeta,
                                                            const char* __restrict__ lcolbytes,
                                                            const int4* __restrict__ lpos4,
                                                            const double2* __restrict__ x, int64_t nblocks,
                                                            int64_t nrows, Op op,
                                                            const int32_t* __restrict__ block_map, SyncArgs sy) {
  constexpr bool NT = (VAR & 1) != 0;
  constexpr bool PRE = (VAR & 2) != 0;
  constexpr int UNR = (VAR & 4) ? 2 : 1;
  constexpr bool DEEP = (VAR & 8) != 0;
  constexpr bool NEAR = (VAR & 16) != 0;
  static_assert(WS == kThreads / 64 || std::is_same<Op, ChebyOp>::value,
                "block_sum sums kThreads / 64 wavefronts: only the fused term without its check may use another width");
  __shared__ double2 lds[WS];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  sync_wait(sy, wg);
  op.begin_issue();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)wg * WS + wave;  // position in the row set
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  int64_t row = nrows;
  typename Op::Pre pre;
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
  if (idx < nblocks) {
    const int64_t b = block_map ? (int64_t)block_map[idx] : idx;
    const int64_t ubase = uptr[b], lbase = lptr[b];
    const int nuq = (int)((uptr[b + 1] - ubase) >> 8);
    const int nlq = (int)((lptr[b + 1] - lbase) >> 8);
    const VT* __restrict__ v = uvals + ubase + lane;
    const int64_t ucm = ucmeta[b], lcm = lcmeta[b];
    const int4* __restrict__ lp4 = lpos4 + (lbase >> 2) + lane;
    row = b * kRB + lane;
    const int64_t rowc = row < nrows ? row : nrows - 1;
    if (PRE) pre = op.pre(rowc);
    auto lower_stencil = [&]() {
      const LowerStencilSlot* __restrict__ ls = reinterpret_cast<const LowerStencilSlot*>(lcolbytes + (lcm >> 2));
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
    };
    auto lower = [&]() {
      if ((lcm & 3) == 2) {
        lower_stencil();
        return;
      }
#pragma unroll UNR
      for (int q = 0; q < nlq; ++q) {
        const int4 c = ld_cols<NT>(lcolbytes, lcm, q, lane, (int)rowc);
        const int4 p = ld_col<NT>(lp4 + (size_t)q * 64);
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
    };
    auto upper = [&]() {
#pragma unroll UNR
      for (int q = 0; q < nuq; ++q) {
        const int4 c = ld_cols<NT>(ucolbytes, ucm, q, lane, (int)rowc);
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
    };
    // The common shape of a lattice / tensor-product H -- both sections stencil-encoded, two quads
    // each (z = 16) -- as straight-line code: all 16 value loads and 16 gathers of the row block are
    // issued before the first FMA (32 KiB in flight per wave instead of 4-8), which is what the
    // kernel needs once the working set no longer sits in the Infinity Cache (N >= 2^22: HBM
    // latency).  Same FMA order as the two loops below: bit-identical.
    if (DEEP && nlq == 2 && nuq == 2 && (lcm & 3) == 2 && (ucm & 3) == 2) {
      const LowerStencilSlot* __restrict__ ls = reinterpret_cast<const LowerStencilSlot*>(lcolbytes + (lcm >> 2));
      const int4* __restrict__ ud = reinterpret_cast<const int4*>(ucolbytes + (ucm >> 2));
      // NEAR (variant bit 4): a gathered element x[row + d] with |d| < 64 is the row-local element x_i of the
      // lane d places away in this very wavefront, which the epilogue has loaded anyway: take it through the
      // cross-lane network (ds_bpermute, no memory access) and load only the |d| lanes whose neighbour lives in
      // the next row block -- half of the gathers of a lattice H never reach the L1.  Same values, same FMA order.
      const bool near_ok = NEAR && Op::kHasXi && PRE && op.xloc() == x && (b + 1) * kRB <= nrows;
      const double2 xi = Op::xi_of(pre);
      auto gather = [&](int d) -> double2 {
        const int c = (int)rowc + d;
        if (near_ok && d > -64 && d < 64) {   // wave-uniform
          const int src = lane + d;
          double2 v = make_double2(__shfl(xi.x, src & 63, 64), __shfl(xi.y, src & 63, 64));
          if ((unsigned)src >= 64u) v = x[c];
          return v;
        }
        return x[c];
      };
      double2 la[8], lx[8], ua[8], ux[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const LowerStencilSlot e = ls[k];
        const int c = (int)rowc + e.delta;
        la[k] = ld_val<false>(uvals + (((c >> 6) == e.cb0 ? e.pb0 : e.pb1) + (c & 63)));
        lx[k] = gather(e.delta);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int4 d = ud[q];
        ua[4 * q + 0] = ld_val<false>(v + (size_t)(4 * q + 0) * 64);
        ua[4 * q + 1] = ld_val<false>(v + (size_t)(4 * q + 1) * 64);
        ua[4 * q + 2] = ld_val<false>(v + (size_t)(4 * q + 2) * 64);
        ua[4 * q + 3] = ld_val<false>(v + (size_t)(4 * q + 3) * 64);
        ux[4 * q + 0] = gather(d.x);
        ux[4 * q + 1] = gather(d.y);
        ux[4 * q + 2] = gather(d.z);
        ux[4 * q + 3] = gather(d.w);
      }
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        cfma_conj(s0, la[k], lx[k]);
        cfma_conj(s1, la[k + 1], lx[k + 1]);
      }
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        cfma(s0, ua[k], ux[k]);
        cfma(s1, ua[k + 1], ux[k + 1]);
      }
    } else
    // the lower section first: its conj-transposed values are the lines the waves of the rows above streamed shortly before
    // on the same XCD.  (The summation order is part of the result's bits: the strip walk follows the same order.)
    {
      lower();
      upper();
    }
    if (!PRE) pre = op.pre(rowc);
  }
  op.begin(lds);
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, idx * kRB + lane);
  finish_check(op, chk, nrm, lds);
  sync_signal(sy);
}

// ---------------------------------------------------------------------------
// CSR SpMV, T lanes per row (sub-wave segmented reduction by shuffles).  General
// fallback for matrices whose row lengths vary too much for RBCSR padding.
// ---------------------------------------------------------------------------
template <int T, class Op, class VT>
__global__ __launch_bounds__(kThreads) void csr_spmv_kernel(const int64_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ cols,
                                                            const VT* __restrict__ vals,
                                                            const double2* __restrict__ x, int64_t nrows,
                                                            Op op) {
  __shared__ double2 lds[kThreads / 64];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  op.begin_issue();
  const int64_t row = ((int64_t)wg * kThreads + threadIdx.x) / T;
  const int tl = threadIdx.x % T;
  double2 s = make_double2(0.0, 0.0);
  if (row < nrows) {
    const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
    int64_t p = p0 + tl;
    double2 s1 = make_double2(0.0, 0.0);
    for (; p + 3 * T < p1; p += 4 * T) {   // four independent load chains in flight
      const int32_t c0 = cols[p], c1 = cols[p + T], c2 = cols[p + 2 * T], c3 = cols[p + 3 * T];
      const double2 a0 = ld_val<false>(vals + p), a1 = ld_val<false>(vals + p + T), a2 = ld_val<false>(vals + p + 2 * T),
                    a3 = ld_val<false>(vals + p + 3 * T);
      const double2 x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
      cfma(s, a0, x0);
      cfma(s1, a1, x1);
      cfma(s, a2, x2);
      cfma(s1, a3, x3);
    }
    for (; p < p1; p += T) cfma(s, ld_val<false>(vals + p), x[cols[p]]);
    s.x += s1.x;
    s.y += s1.y;
  }
#pragma unroll
  for (int o = T / 2; o > 0; o >>= 1) {
    s.x += __shfl_down(s.x, o, T);
    s.y += __shfl_down(s.y, o, T);
  }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  op.begin(lds);
  if (row < nrows && tl == 0) op.row(row, s, op.pre(row), chk, nrm, row);
  finish_check(op, chk, nrm, lds);
}

int spmv_grid_size(const DevMatrix& A) {
  // (dense: one wavefront per R rows -- the grid launch_dense_gemv uses; the per-workgroup check triples of check_normalization
  // are sized and reduced with THIS count, so the two must be one expression)
  if (A.format == QP_FMT_DENSE) return dense_gemv_grid(A.nrows);
  if (A.format == QP_FMT_RBCSR || A.format == QP_FMT_HRB) return (int)((A.nblocks + kThreads / 64 - 1) / (kThreads / 64));
  const int64_t threads = A.nrows * A.lanes_per_row;
  return (int)((threads + kThreads - 1) / kThreads);
}

template <class Op>
static int launch_spmv(hipStream_t s, const DevMatrix& A, const double2* x, const Op& op, Stats* st,
                       const RowSet* rs = nullptr) {
  if (A.nrows == 0) return QP_OK;
  int grid = spmv_grid_size(A);
  const int32_t* bmap = nullptr;
  int64_t nblk = A.nblocks;
  const SyncArgs sy = rs ? rs->sync : SyncArgs();
  static const Tuning kDefaults;
  const Tuning& tun = A.tun ? *A.tun : kDefaults;
  // Eight instead of four row blocks per workgroup for the fused Chebyshev term wherever nothing counts
  // workgroups of four: not with the per-workgroup check partials, and of the two launches of a split term only for the
  // interior one (no completion signal, no mirror map; its wait threshold, given in workgroups of four, is halved and
  // rounded down: the workgroup that straddles the threshold waits as well)
  // an operator with irregular columns: its column-blocked mirror (kernels_colblock.hip), whole-operator launches only
  if (!rs && A.cb && A.cb->valid && tun.colblock != 0) {
    bool launched = false;
    int rcb;
    if constexpr (std::is_same<Op, ChebyOp>::value) rcb = launch_colblock_cheby(s, A, x, op.e, tun, &launched);
    else rcb = launch_colblock_plain(s, A, x, op.e, tun, &launched);
    if (rcb != QP_OK) return rcb;
    if (launched) {
      if (st) {
        st->n_launch++;
        st->n_matvec++;
      }
      return QP_OK;
    }
  }
  bool wide_ok = false;
  SyncArgs sy8 = sy;
  if constexpr (std::is_same<Op, ChebyOp>::value) {
    wide_ok = !op.e.check_partials && (!rs || (!rs->sync.signal && !op.e.mirror));
    sy8.wait_from_wg = sy.wait_from_wg / 2;
  }
  if (rs && rs->block_map) {
    if (A.format != QP_FMT_RBCSR && A.format != QP_FMT_HRB) return fail(QP_E_BAD_ARG, "row sets need a row-block format");
    bmap = rs->block_map;
    nblk = rs->nmap;
    if (nblk == 0) return QP_OK;
    grid = (int)((nblk + kThreads / 64 - 1) / (kThreads / 64));
  }
  if (A.format == QP_FMT_RBCSR && A.cv && A.cv->valid && tun.value_dict != 0) {
    // few distinct values per block: one byte + a cached table line per entry instead of the value (kernels_coded.hip)
    int rcc;
    if constexpr (std::is_same<Op, ChebyOp>::value) rcc = launch_rbcsr_coded_cheby(s, A, x, op.e, nblk, bmap, sy, wide_ok);
    else rcc = launch_rbcsr_coded_plain(s, A, x, op.e, nblk, bmap, sy);
    if (rcc != QP_OK) return rcc;
    if (st) {
      st->n_launch++;
      if (!rs || rs->count) st->n_matvec++;
    }
    return QP_OK;
  }
  if (A.format == QP_FMT_RBCSR) {
#define QP_RB_CASE(VV)                                                                                   \
  case VV:                                                                                               \
    if (A.vals_r)                                                                                        \
      hipLaunchKernelGGL((rbcsr_spmv_kernel<Op, VV, double>), dim3(grid), dim3(kThreads), 0, s, A.bptr, A.cmeta, \
                         reinterpret_cast<const char*>(A.cols), A.vals_r, x, nblk, A.nrows, op, bmap, sy); \
    else                                                                                                 \
      hipLaunchKernelGGL((rbcsr_spmv_kernel<Op, VV, double2>), dim3(grid), dim3(kThreads), 0, s, A.bptr, A.cmeta, \
                         reinterpret_cast<const char*>(A.cols), A.vals, x, nblk, A.nrows, op, bmap, sy);   \
    break;
    if constexpr (std::is_same<Op, ChebyOp>::value) {
      // as for the Hermitian-packed kernel below: eight row blocks per workgroup for the plain fused term of a whole operator
      if (wide_ok && (tun.rbcsr_variant & 7) == 7 && A.stored > A.nblocks * (int64_t)(kRB * 8)) {
        const int g8 = (int)((nblk + 7) / 8);
        if (A.vals_r)
          hipLaunchKernelGGL((rbcsr_spmv_kernel<Op, 7, double, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals_r, x, nblk, A.nrows, op, bmap, sy8);
        else
          hipLaunchKernelGGL((rbcsr_spmv_kernel<Op, 7, double2, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals, x, nblk, A.nrows, op, bmap, sy8);
        QP_HIP(hipGetLastError());
        if (st) {
          st->n_launch++;
          if (!rs || rs->count) st->n_matvec++;
        }
        return QP_OK;
      }
    }
    // the deeper unroll (bit 2: 156-160 VGPRs, 3 wavefronts per SIMD) pays from three quads per row on; blocks of
    // at most two quads (8 entries per row: the Liouvillian of config C3) take the shallow one (88 VGPRs, 5 per
    // SIMD) -- at N = 2^18 that is one round of wavefronts instead of one and a third.  Same sums either way.
    int variant = tun.rbcsr_variant & 7;
    if (A.stored <= A.nblocks * (int64_t)(kRB * 8)) variant &= ~4;
    switch (variant) {
      QP_RB_CASE(0)
      QP_RB_CASE(1)
      QP_RB_CASE(2)
      QP_RB_CASE(3)
      QP_RB_CASE(4)
      QP_RB_CASE(5)
      QP_RB_CASE(6)
      QP_RB_CASE(7)
    }
#undef QP_RB_CASE
  } else if (A.format == QP_FMT_HRB) {
#define QP_HRB_CASE(VV)                                                                                  \
  case VV:                                                                                               \
    if (A.vals_r)                                                                                        \
      hipLaunchKernelGGL((hrb_spmv_kernel<Op, VV, double>), dim3(grid), dim3(kThreads), 0, s, A.bptr, A.cmeta, \
                         reinterpret_cast<const char*>(A.cols), A.vals_r, A.lptr, A.lcmeta,              \
                         reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, \
                         nblk, A.nrows, op, bmap, sy);                                 \
    else                                                                                                 \
      hipLaunchKernelGGL((hrb_spmv_kernel<Op, VV, double2>), dim3(grid), dim3(kThreads), 0, s, A.bptr, A.cmeta, \
                         reinterpret_cast<const char*>(A.cols), A.vals, A.lptr, A.lcmeta,                \
                         reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, \
                         nblk, A.nrows, op, bmap, sy);                                 \
    break;
    if constexpr (std::is_same<Op, ChebyOp>::value) {
      // a lattice operator (one stencil repeated down the row blocks): the strip walk (kernels_walk.hip) -- whole operator,
      // no normalisation check, the gathered vector's own rows being the row-local operand; same sums as the kernels below
      // ... or the interior launch of a split term whose row set carries a plan of its own (RowSet::walk)
      const bool whole = !rs && A.walk && A.walk->valid;
      const bool set_walk = rs && rs->walk && rs->walk->valid && !rs->sync.signal;
      if (tun.hrb_walk && (whole || set_walk) && wide_ok && (tun.rbcsr_variant & 31) == 15 &&
          op.e.xloc == x && !op.e.mirror) {
        bool launched = false;
        const int rcw = launch_hrb_walk_cheby(s, A, x, op.e, tun, &launched, rs);
        if (rcw != QP_OK) return rcw;
        if (launched) {
          if (st) {
            st->n_launch++;
            if (!rs || rs->count) st->n_matvec++;
          }
          return QP_OK;
        }
      }
      // eight row blocks per workgroup instead of four (see wide_ok above): half as many workgroups to dispatch, 36.3 ->
      // 35.2 us per term at N = 2^20 (profiles/r02/kbench_banded.txt); the same sums
      if (wide_ok && (tun.rbcsr_variant & 31) == 15) {
        const int g8 = (int)((nblk + 7) / 8);
        if (A.vals_r)
          hipLaunchKernelGGL((hrb_spmv_kernel<Op, 15, double, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals_r, A.lptr, A.lcmeta,
                             reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, nblk, A.nrows, op,
                             bmap, sy8);
        else
          hipLaunchKernelGGL((hrb_spmv_kernel<Op, 15, double2, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals, A.lptr, A.lcmeta,
                             reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, nblk, A.nrows, op,
                             bmap, sy8);
        QP_HIP(hipGetLastError());
        if (st) {
          st->n_launch++;
          if (!rs || rs->count) st->n_matvec++;
        }
        return QP_OK;
      }
    }
    switch (tun.rbcsr_variant & 31) {
      QP_HRB_CASE(0)
      QP_HRB_CASE(1)
      QP_HRB_CASE(2)
      QP_HRB_CASE(3)
      QP_HRB_CASE(4)
      QP_HRB_CASE(5)
      QP_HRB_CASE(6)
      QP_HRB_CASE(7)
      QP_HRB_CASE(8)
      QP_HRB_CASE(15)
      QP_HRB_CASE(31)
      default: return fail(QP_E_BAD_ARG, "rbcsr_variant %d has no Hermitian-packed kernel (0-8, 15, 31)", tun.rbcsr_variant);
    }
#undef QP_HRB_CASE
  } else {
#define QP_CSR_CASE(TT)                                                                                  \
  case TT:                                                                                               \
    if (A.vals_r)                                                                                        \
      hipLaunchKernelGGL((csr_spmv_kernel<TT, Op, double>), dim3(grid), dim3(kThreads), 0, s, A.rowptr, A.cols, \
                         A.vals_r, x, A.nrows, op);                                                      \
    else                                                                                                 \
      hipLaunchKernelGGL((csr_spmv_kernel<TT, Op, double2>), dim3(grid), dim3(kThreads), 0, s, A.rowptr, A.cols, \
                         A.vals, x, A.nrows, op);                                                        \
    break;
    switch (A.lanes_per_row) {
      QP_CSR_CASE(2)
      QP_CSR_CASE(4)
      QP_CSR_CASE(8)
      QP_CSR_CASE(16)
      QP_CSR_CASE(32)
      QP_CSR_CASE(64)
      default:
        return fail(QP_E_INTERNAL, "bad lanes_per_row %d", A.lanes_per_row);
    }
#undef QP_CSR_CASE
  }
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    if (!rs || rs->count) st->n_matvec++;
  }
  return QP_OK;
}

// Chebyshev term for an operator without stored entries: s = A x by the owner's apply, then
// the same row epilogue as the fused kernels
__global__ __launch_bounds__(kThreads) void cheby_epilogue_kernel(const double2* __restrict__ s, int64_t n, ChebyOp op) {
  __shared__ double2 lds[kThreads / 64];
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) op.row(i, s[i], op.pre(i), chk, nrm, i);
  finish_check(op, chk, nrm, lds);
}

int launch_spmv_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, Stats* st,
                      const RowSet* rs) {
  if (A.format == QP_FMT_MATFREE) {
    if (rs) return fail(QP_E_BAD_ARG, "row sets need a row-block format");
    if (e.check_partials) return fail(QP_E_BAD_ARG, "check_normalization is not available for a matrix-free operator");
    if (A.matfree_cheby && e.xloc == x && !e.mirror) return A.matfree_cheby(s, A.matfree, x, e, st);
    double2* tmp = A.matfree_scratch(A.matfree);
    int rc = A.matfree_apply(s, A.matfree, x, tmp, make_double2(1.0, 0.0), make_double2(0.0, 0.0), st);
    if (rc != QP_OK) return rc;
    ChebyOp op{e};
    const int grid = (int)((A.nrows + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(cheby_epilogue_kernel, dim3(grid), dim3(kThreads), 0, s, tmp, A.nrows, op);
    QP_HIP(hipGetLastError());
    if (st) st->n_launch++;
    return QP_OK;
  }
  if (A.format == QP_FMT_DENSE) {
    if (rs && rs->block_map) return fail(QP_E_BAD_ARG, "row sets need a row-block format");
    return launch_dense_gemv_cheby(s, A, x, e, st);
  }
  ChebyOp op{e};
  int rc = launch_spmv(s, A, x, op, st, rs);
  // algorithmic bytes, SURVEY 8d: z (V + 4) N + 4 (N + 1) + 5 * 16 N
  if (st && rc == QP_OK && (!rs || rs->count))
    st->spmv_bytes += 20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 80.0 * (double)A.nrows;
  return rc;
}

int launch_spmv_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, Stats* st) {
  if (A.format == QP_FMT_MATFREE)
    return A.matfree_apply(s, A.matfree, x, e.y, e.alpha, e.beta_zero ? make_double2(0.0, 0.0) : e.beta, st);
  if (A.format == QP_FMT_DENSE) return launch_dense_gemv_plain(s, A, x, e, st);
  PlainOp op{e};
  int rc = launch_spmv(s, A, x, op, st);
  // plain SpMV: matrix + read x + write y  ((20 z + 36) N)
  if (st && rc == QP_OK) st->spmv_bytes += 20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 32.0 * (double)A.nrows;
  return rc;
}

}  // namespace qp

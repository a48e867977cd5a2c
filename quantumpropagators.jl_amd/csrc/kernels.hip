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
                                                            const int32_t* __restrict__ block_map, SyncArgs sy,
                                                            int lower_last) {
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
    if (DEEP && !lower_last && nlq == 2 && nuq == 2 && (lcm & 3) == 2 && (ucm & 3) == 2) {
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
    // order of the two sections (tuning key "hrb_lower_last"): the conj-transposed values of
    // the lower section are found in L2 only if the wave that owns them has already fetched
    // them; summation order changes with it, bitwise reproducibility per setting is kept
    if (lower_last) {
      upper();
      lower();
    } else {
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

static inline int ew_grid(int64_t n);

// ---------------------------------------------------------------------------
// Batched states (BASELINE configs[4]): b states as a panel X[i*b + s] (state index
// contiguous).  One wavefront per row, lane = state: the matrix entry is wave-uniform
// (scalar loads, broadcast for free), every gather of X[col, :] is a contiguous 16*b-byte
// burst, and the matrix traffic is amortised over the b states (20 z + 80 b bytes per row).
// There is no dense contraction to feed MFMA: H has scalar entries, so per row this is z
// AXPYs of length b (0.4 flop/B at b = 64, far below the fp64 ridge).
// ---------------------------------------------------------------------------
// TS = states per tile.  A workgroup covers 256/TS rows x TS states; gridDim.y walks the state
// tiles, so the chip streams all rows for one tile of states before the next: the gather
// window of a banded H (+-4096 rows) is 4096 * 16 * TS bytes per direction and must stay
// inside an XCD's 4 MiB L2 -- with all 64 states per pass it does not (measured 3.1 TB/s
// algorithmic, the far gathers spill to HBM), with TS = 16 it does, at the price of streaming
// the matrix 64/TS times.
template <class Op, int TS>
__global__ __launch_bounds__(kThreads) void csr_spmm_kernel(const int64_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ cols,
                                                            const double2* __restrict__ vals,
                                                            const double2* __restrict__ X, int64_t nrows, int b,
                                                            Op op) {
  constexpr int RPW = kThreads / TS;   // rows per workgroup
  constexpr int CH = TS;               // matrix entries staged per row and chunk (one per lane of the row)
  // (value, column) of the workgroup's rows, staged through LDS so that the TS lanes of a row
  // read each entry as an LDS broadcast instead of TS redundant global loads; +1 pads the
  // row stride off the bank period
  __shared__ double2 s_val[RPW][CH + 1];
  __shared__ int s_col[RPW][CH + 1];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int rl = threadIdx.x / TS, sl = threadIdx.x % TS;
  const int64_t row = (int64_t)wg * RPW + rl;
  const int st = blockIdx.y * TS + sl;
  const bool rvalid = row < nrows;
  const bool active = rvalid && st < b;
  const int64_t p0 = rvalid ? rowptr[row] : 0, p1 = rvalid ? rowptr[row + 1] : 0;
  // longest row of the workgroup (uniform loop bound)
  int len = (int)(p1 - p0);
  __shared__ int s_maxlen;
  if (threadIdx.x == 0) s_maxlen = 0;
  __syncthreads();
  if (sl == 0) atomicMax(&s_maxlen, len);
  __syncthreads();
  const int maxlen = s_maxlen;
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  const int64_t e = active ? row * (int64_t)b + st : 0;
  typename Op::Pre pre;
  if (active) pre = op.pre(e);
  double2 acc0 = make_double2(0.0, 0.0), acc1 = make_double2(0.0, 0.0);
  for (int k0 = 0; k0 < maxlen; k0 += CH) {
    if (k0 > 0) __syncthreads();
    if (k0 + sl < len) {
      s_val[rl][sl] = ld_stream<Op::kStream>(vals + p0 + k0 + sl);
      s_col[rl][sl] = Op::kStream ? __builtin_nontemporal_load(cols + p0 + k0 + sl) : cols[p0 + k0 + sl];
    }
    __syncthreads();
    const int cnt = min(CH, len - k0);
    if (active) {
      int k = 0;
      for (; k + 3 < cnt; k += 4) {
        const double2 x0 = X[(int64_t)s_col[rl][k] * b + st];
        const double2 x1 = X[(int64_t)s_col[rl][k + 1] * b + st];
        const double2 x2 = X[(int64_t)s_col[rl][k + 2] * b + st];
        const double2 x3 = X[(int64_t)s_col[rl][k + 3] * b + st];
        cfma(acc0, s_val[rl][k], x0);
        cfma(acc1, s_val[rl][k + 1], x1);
        cfma(acc0, s_val[rl][k + 2], x2);
        cfma(acc1, s_val[rl][k + 3], x3);
      }
      for (; k < cnt; ++k) cfma(acc0, s_val[rl][k], X[(int64_t)s_col[rl][k] * b + st]);
    }
  }
  if (active) op.row(e, make_double2(acc0.x + acc1.x, acc0.y + acc1.y), pre, chk, nrm, e);
}

// ---------------------------------------------------------------------------
// Batched states, one wavefront per row, lane = state (the default for panels of more than 32
// states).  A matrix entry is the same for all 64 lanes: the wave loads the row's (value, column)
// pairs once, one entry per lane in a single coalesced burst, and broadcasts them through SGPRs
// (v_readlane), so the matrix is streamed ONCE for all states (the tiled kernel above streams it
// 64 / TS times) and every gather of X[col, :] is one line-aligned 1-KiB wave access.
//
// What decides the speed is how often a row of X comes from HBM: row r is gathered by every row
// i with H[i, r] != 0.  For H = H_a (x) 1 + 1 (x) H_c -- the lattice / tensor-product operators of
// BASELINE's workloads: offsets +-1..4 and +-1024 k -- those rows are a span of 8192 rows apart in
// natural order, 8 MiB of X at 64 states, twice an XCD's L2.  `order` (operator_spmm_order in
// engine_core.hip) lists the rows strip by strip -- for a strip of `sw` inner indices c, all outer
// indices a in turn, i = a g + c -- so that the +-k g neighbours are the rows visited just before and
// just after, and the window a wave can hit in L2 shrinks to (2 a_max + 1) sw rows.  Any
// permutation gives the same values bit for bit (rows are independent); it only moves traffic.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Scalar-memory variant (knob spmm_rw = 0): the row's entries are wave-uniform, so they can be fetched by the
// scalar unit (s_load: value and column straight into SGPRs, which the FMAs and the gather addresses take as
// operands) instead of one entry per lane + v_readlane broadcasts -- five VALU instructions per entry less.
// WS wavefronts (consecutive walk positions) per workgroup (knob spmm_wg): 8 measured 282 us per term of config C5
// against 302 with 4 and 290 with 16; 2-D tiles of walk positions per workgroup instead of runs: no difference
// (profiles/r02/batched_c5_sweep.txt)
template <class Op, int WS>
__global__ __launch_bounds__(64 * WS) void spmm_rows_smem_kernel(const int64_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ cols,
                                                                  const double2* __restrict__ vals,
                                                                  const double2* __restrict__ X, int64_t nrows, int b, Op op,
                                                                  const int32_t* __restrict__ order) {
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t pos = (int64_t)wg * WS + wave;
  if (pos >= nrows) return;
  const int64_t row = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[pos]) : pos;
  const int st = blockIdx.y * 64 + lane;
  const bool active = st < b;
  const int stc = active ? st : b - 1;
  const int64_t p0 = __builtin_amdgcn_readfirstlane((int)rowptr[row]) ;
  const int len = __builtin_amdgcn_readfirstlane((int)(rowptr[row + 1] - rowptr[row]));
  const int64_t e = row * (int64_t)b + stc;
  const typename Op::Pre pre = op.pre(e);
  const double2* __restrict__ Xs = X + stc;
  const double2* __restrict__ rv = vals + p0;    // wave-uniform addresses: scalar loads
  const int32_t* __restrict__ rc = cols + p0;
  double2 acc0 = make_double2(0.0, 0.0), acc1 = make_double2(0.0, 0.0);
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  int k = 0;
  // sums in the order of csr_spmm_kernel (groups of four alternating between two partial sums, remainder into the first)
  for (; k + 7 < len; k += 8) {
    double2 x[8], a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x[u] = Xs[(int64_t)rc[k + u] * b];
      a[u] = rv[k + u];
    }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      cfma(acc0, a[u], x[u]);
      cfma(acc1, a[u + 1], x[u + 1]);
    }
  }
  if (k + 3 < len) {
    double2 x[4], a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = Xs[(int64_t)rc[k + u] * b];
      a[u] = rv[k + u];
    }
    cfma(acc0, a[0], x[0]);
    cfma(acc1, a[1], x[1]);
    cfma(acc0, a[2], x[2]);
    cfma(acc1, a[3], x[3]);
    k += 4;
  }
  for (; k < len; ++k) cfma(acc0, rv[k], Xs[(int64_t)rc[k] * b]);
  if (active) op.row(e, make_double2(acc0.x + acc1.x, acc0.y + acc1.y), pre, chk, nrm, e);
}

// One wavefront walks RW consecutive positions of the row walk.  The dependent loads in front of a
// row's gathers (walk position -> row, row pointers, the row's entries) are issued for all RW rows
// together, so a row costs one round of up to 16 gathers instead of a chain of four memory latencies.
template <class Op, int RW, int G>
__global__ __launch_bounds__(kThreads) void spmm_rows_kernel(const int64_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ cols,
                                                             const double2* __restrict__ vals,
                                                             const double2* __restrict__ X, int64_t nrows, int b, Op op,
                                                             const int32_t* __restrict__ order) {
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t pos0 = ((int64_t)wg * (kThreads / 64) + wave) * RW;
  if (pos0 >= nrows) return;
  const int nr = (int)min((int64_t)RW, nrows - pos0);   // wave-uniform
  const int st = blockIdx.y * 64 + lane;
  const bool active = st < b;
  const int stc = active ? st : b - 1;
  const double2* __restrict__ Xs = X + stc;
  // lanes 0 .. nr-1: row of walk position pos0 + lane and its pointer pair
  int rv = 0;
  int64_t pv0 = 0, pv1 = 0;
  if (lane < nr) {
    rv = order ? order[pos0 + lane] : (int)(pos0 + lane);
    pv0 = rowptr[rv];
    pv1 = rowptr[rv + 1];
  }
  int64_t rowi[RW], p0[RW];
  int len[RW], mc[RW];
  double2 mv[RW];
#pragma unroll
  for (int l = 0; l < RW; ++l) {
    rowi[l] = __builtin_amdgcn_readlane(rv, l);
    const int lo = __builtin_amdgcn_readlane((int)(pv0 & 0xffffffff), l), hi = __builtin_amdgcn_readlane((int)(pv0 >> 32), l);
    p0[l] = ((int64_t)hi << 32) | (uint32_t)lo;
    len[l] = __builtin_amdgcn_readlane((int)(pv1 - pv0), l);
  }
  // entries 0 .. 63 of every row: one coalesced load of values and one of columns per row, all in flight
#pragma unroll
  for (int l = 0; l < RW; ++l) {
    mv[l] = make_double2(0.0, 0.0);
    mc[l] = 0;
    if (l < nr && lane < len[l]) {
      mv[l] = ld_stream<Op::kStream>(vals + p0[l] + lane);
      mc[l] = Op::kStream ? __builtin_nontemporal_load(cols + p0[l] + lane) : cols[p0[l] + lane];
    }
  }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  // row-local operands of all RW rows first, their stores last: the wave's accesses to each of the
  // streamed vectors (v0, the accumulator, the new term) come as one burst of RW KiB
  typename Op::Pre pre[RW];
  double2 res[RW];
#pragma unroll
  for (int l = 0; l < RW; ++l)
    if (l < nr) pre[l] = op.pre(rowi[l] * (int64_t)b + stc);
#pragma unroll
  for (int l = 0; l < RW; ++l) {
    if (l >= nr) break;
    double2 acc0 = make_double2(0.0, 0.0), acc1 = make_double2(0.0, 0.0);
    double2 cv = mv[l];
    int cc = mc[l];
    for (int k0 = 0; k0 < len[l]; k0 += 64) {
      const int cnt = min(64, len[l] - k0);   // wave-uniform
      if (k0 > 0) {
        cv = make_double2(0.0, 0.0);
        cc = 0;
        if (lane < cnt) {
          cv = ld_stream<Op::kStream>(vals + p0[l] + k0 + lane);
          cc = Op::kStream ? __builtin_nontemporal_load(cols + p0[l] + k0 + lane) : cols[p0[l] + k0 + lane];
        }
      }
      // the sums run in the order of csr_spmm_kernel: within groups of four, entries alternate between
      // two partial sums; the remainder goes to the first.  Up to G gathers in flight.
      int k = 0;
#define QP_SPMM_GROUP(GG)                                                                              \
  {                                                                                                    \
    double2 x[GG];                                                                                     \
    _Pragma("unroll") for (int u = 0; u < GG; ++u) x[u] = Xs[(int64_t)__builtin_amdgcn_readlane(cc, k + u) * b]; \
    _Pragma("unroll") for (int u = 0; u < GG; u += 2) {                                                \
      cfma(acc0, make_double2(readlane_f64(cv.x, k + u), readlane_f64(cv.y, k + u)), x[u]);            \
      cfma(acc1, make_double2(readlane_f64(cv.x, k + u + 1), readlane_f64(cv.y, k + u + 1)), x[u + 1]); \
    }                                                                                                  \
    k += GG;                                                                                           \
  }
      if (G >= 16)
        while (k + 15 < cnt) QP_SPMM_GROUP(16)
      while (k + 7 < cnt) QP_SPMM_GROUP(8)
      if (k + 3 < cnt) QP_SPMM_GROUP(4)
#undef QP_SPMM_GROUP
      for (; k < cnt; ++k)
        cfma(acc0, make_double2(readlane_f64(cv.x, k), readlane_f64(cv.y, k)), Xs[(int64_t)__builtin_amdgcn_readlane(cc, k) * b]);
    }
    res[l] = make_double2(acc0.x + acc1.x, acc0.y + acc1.y);
  }
#pragma unroll
  for (int l = 0; l < RW; ++l)
    if (l < nr && active) {
      const int64_t e = rowi[l] * (int64_t)b + stc;
      op.row(e, res[l], pre[l], chk, nrm, e);
    }
}

// knob spmm_nt -- nontemporal matrix and row-local streams in the batched kernel: 0 never, 2 always,
// 1 when one panel vector is larger than what the caches could keep until the next launch anyway
template <int TS>
static void launch_spmm_cheby_t(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals,
                                const double2* X, int64_t nrows, int b, const ChebyEpi& e, int spmm_nt) {
  const int rpw = kThreads / TS;
  dim3 grid((unsigned)((nrows + rpw - 1) / rpw), (unsigned)((b + TS - 1) / TS));
  const bool nt = spmm_nt == 2 || (spmm_nt == 1 && (double)nrows * b * sizeof(double2) >= 128.0 * 1024 * 1024);
  if (nt) {
    ChebyOpT<true> op{e};
    hipLaunchKernelGGL((csr_spmm_kernel<ChebyOpT<true>, TS>), grid, dim3(kThreads), 0, s, rowptr, cols, vals, X, nrows, b, op);
  } else {
    ChebyOp op{e};
    hipLaunchKernelGGL((csr_spmm_kernel<ChebyOp, TS>), grid, dim3(kThreads), 0, s, rowptr, cols, vals, X, nrows, b, op);
  }
}

int launch_spmm_cheby(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals,
                      const double2* X, int64_t nrows, int64_t nnz, int b, const ChebyEpi& e, const Tuning& tun,
                      bool rows_kernel, const int32_t* order, Stats* st) {
  if (nrows == 0) return QP_OK;
  if (rows_kernel) {
    const bool nt = tun.spmm_nt == 2 || (tun.spmm_nt == 1 && (double)nrows * b * sizeof(double2) >= 128.0 * 1024 * 1024);
#define QP_SPMM_ROWS(RW)                                                                                         \
  {                                                                                                              \
    const int64_t per_wg = (int64_t)(kThreads / 64) * RW;                                                        \
    dim3 grid((unsigned)((nrows + per_wg - 1) / per_wg), (unsigned)((b + 63) / 64));                             \
    if (nt) {                                                                                                    \
      ChebyOpT<true> op{e};                                                                                      \
      hipLaunchKernelGGL((spmm_rows_kernel<ChebyOpT<true>, RW, 8>), grid, dim3(kThreads), 0, s, rowptr, cols, vals, X, nrows, b, op, order); \
    } else {                                                                                                     \
      ChebyOp op{e};                                                                                             \
      hipLaunchKernelGGL((spmm_rows_kernel<ChebyOp, RW, 8>), grid, dim3(kThreads), 0, s, rowptr, cols, vals, X, nrows, b, op, order); \
    }                                                                                                            \
  }
    if (tun.spmm_rw == 0 && nnz <= (int64_t)INT32_MAX) {   // (the scalar-entry kernel broadcasts a 32-bit row pointer)
#define QP_SPMM_SMEM(WS)                                                                                          \
  {                                                                                                               \
    dim3 grid((unsigned)((nrows + (WS) - 1) / (WS)), (unsigned)((b + 63) / 64));                                  \
    if (nt) {                                                                                                     \
      ChebyOpT<true> op{e};                                                                                       \
      hipLaunchKernelGGL((spmm_rows_smem_kernel<ChebyOpT<true>, WS>), grid, dim3(64 * (WS)), 0, s, rowptr, cols, vals, X, nrows, b, op, order); \
    } else {                                                                                                      \
      ChebyOp op{e};                                                                                              \
      hipLaunchKernelGGL((spmm_rows_smem_kernel<ChebyOp, WS>), grid, dim3(64 * (WS)), 0, s, rowptr, cols, vals, X, nrows, b, op, order); \
    }                                                                                                             \
  }
      if (tun.spmm_wg == 16) QP_SPMM_SMEM(16)
      else if (tun.spmm_wg == 8) QP_SPMM_SMEM(8)
      else QP_SPMM_SMEM(4)
#undef QP_SPMM_SMEM
    } else
    switch (tun.spmm_rw) {
      case 2: QP_SPMM_ROWS(2) break;
      case 4: QP_SPMM_ROWS(4) break;
      case 8: QP_SPMM_ROWS(8) break;
      default: QP_SPMM_ROWS(1) break;
    }
#undef QP_SPMM_ROWS
  } else
  switch ((b <= 8 && tun.spmm_tile == 16) ? 8 : tun.spmm_tile) {   // a panel of at most eight states (one GPU's share of 64 over 8): no idle lanes
    case 8: launch_spmm_cheby_t<8>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt); break;
    case 32: launch_spmm_cheby_t<32>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt); break;
    case 64: launch_spmm_cheby_t<64>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt); break;
    default: launch_spmm_cheby_t<16>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt); break;
  }
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
    st->spmv_bytes += 20.0 * (double)nnz + 4.0 * (double)(nrows + 1) + 80.0 * (double)nrows * b;
  }
  return QP_OK;
}

// CSR-ordered copy of the current operator values: out[p] = map[p] >= 0 ? vals[map[p]]
//                                                          : conj(vals[-map[p]-1])
__global__ __launch_bounds__(kThreads) void gather_csr_vals_kernel(double2* __restrict__ out,
                                                                   const double2* __restrict__ vals,
                                                                   const int64_t* __restrict__ map, int64_t nnz) {
  for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * kThreads) {
    const int64_t m = map[p];
    double2 v = vals[m >= 0 ? m : -m - 1];
    if (m < 0) v.y = -v.y;
    out[p] = v;
  }
}

int launch_gather_csr_vals(hipStream_t s, double2* out, const double2* vals, const int64_t* map, int64_t nnz,
                           Stats* st) {
  if (nnz == 0) return QP_OK;
  hipLaunchKernelGGL(gather_csr_vals_kernel, dim3(ew_grid(nnz)), dim3(kThreads), 0, s, out, vals, map, nnz);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

// ---------------------------------------------------------------------------
// Small systems (the reference's own test sizes: N = 2 ... about a thousand): a launch per
// Chebychev term is bound by the launch itself, not by the matrix.  One persistent
// workgroup runs the WHOLE time grid -- evaluate!(G, tlist, n) per interval, the
// three-term recurrence (src/cheby.jl:171-211), the observables and the state storage of
// propagate (src/propagate.jl:283-344) -- with a workgroup barrier where the multi-launch
// path has a kernel boundary.  The vectors live in LDS and every lane keeps its share of
// the matrix in registers: a group of `lanes` lanes owns rows g, g + G, ... (`rows_per_group`
// of them), each lane `ent` entries of each row; rows_per_group * ent <= kSmallEpt.
// ---------------------------------------------------------------------------
// (the single-workgroup kernels keep their cross-lane sums on the LDS crossbar: their eight wavefronts are VALU-bound in
// step with one another, and the extra vector instructions of the DPP forms cost more than the crossbar's latency --
// config C1 8.65 ms per 200 steps against 9.1 with DPP wavefront sums and 10.0 with DPP row butterflies too)
__device__ __forceinline__ double2 small_block_sum(double2 v, double2* red) {
  v.x = wave_sum_lds(v.x);
  v.y = wave_sum_lds(v.y);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) red[w] = v;
  __syncthreads();
  double2 r = red[0];
#pragma unroll
  for (int i = 1; i < kSmallThreads / 64; ++i) {
    r.x += red[i].x;
    r.y += red[i].y;
  }
  __syncthreads();
  return r;
}

template <int E, int R>
__global__ __launch_bounds__(kSmallThreads) void cheby_propagate_small_kernel(SmallArgs s) {
  constexpr int NS = E * R;   // register slots in use
  extern __shared__ double2 small_lds[];
  double2* red = small_lds;                        // [16] reduction scratch
  double2* coef = small_lds + kSmallThreads / 64;  // [nops] effective coefficients of the step
  double2* vec = coef + s.nops;
  const int64_t n = s.n;
  double2* A = vec;
  double2* B = vec + n;
  double2* ACC = vec + 2 * n;
  const int T = s.lanes;
  const int tid = threadIdx.x;
  const int lane = tid & (T - 1);
  const int64_t grp = tid / T, ngrp = kSmallThreads / T;
  const int nterms = s.n_coeffs - 1;
  const int drift = s.nops - s.ncoeffs;

  for (int64_t i = tid; i < n; i += kSmallThreads) A[i] = s.psi[i];
  // register-resident share of the matrix: slot e <-> (row grp + (e / E) ngrp, entry lane + (e % E) T)
  // More than 16 slots per lane: column (12 bits), plane position + 1 (19 bits) and the conjugation
  // flag (sign bit) share one register; the values of the slots from 16 on live in LDS, [slot][thread];
  // evaluate! and the mat-vec go through the slots 8 at a time.
  constexpr bool PACK = NS > 16;
  constexpr int CHK = PACK ? 8 : NS;
  constexpr int NR = PACK ? 16 : NS;
  double2* vlds = vec + 3 * n;         // (NS - NR) * kSmallThreads values (the launcher sizes the allocation)
  int32_t rc[NS];
  int32_t rm[PACK ? 1 : NS];   // 0: no entry; +(m+1): plane[m]; -(m+1): conj(plane[m])
  double2 rv[NR];
#pragma unroll
  for (int e = 0; e < NS; ++e) {
    rc[e] = 0;
    if (!PACK) rm[e] = 0;
    if (e < NR) rv[e] = make_double2(0.0, 0.0);
    const int64_t r = grp + (int64_t)(e / E) * ngrp;
    if (r < n) {
      const int64_t k = s.rowptr[r] + lane + (int64_t)(e % E) * T;
      if (k < s.rowptr[r + 1]) {
        const int64_t m = s.map[k];                // m < 0 encodes -(pos + 1): conj(plane[pos])
        if (PACK) {
          const int64_t pos1 = m >= 0 ? m + 1 : -m;
          rc[e] = (int32_t)((uint32_t)s.cols[k] | ((uint32_t)pos1 << 12) | (m < 0 ? 0x80000000u : 0u));
        } else {
          rc[e] = s.cols[k];
          rm[e] = (int32_t)(m >= 0 ? m + 1 : m);
        }
      }
    }
    if (PACK && (e % CHK) == CHK - 1) __builtin_amdgcn_sched_barrier(0);
  }
  auto col_of = [&](int e) -> int { return PACK ? (rc[e] & 0xfff) : rc[e]; };
  __syncthreads();

  // <psi|O|psi> for every observable and the state history, at storage row `row`
  const int TO = s.obs_lanes;
  const int olane = tid & (TO - 1);
  const int64_t ogrp = tid / TO, ongrp = kSmallThreads / TO;
  auto record = [&](const double2* psi, int row) {
    for (int o = 0; o < s.nobs; ++o) {
      const SmallObs ob = s.obs[o];
      double2 part = make_double2(0.0, 0.0);
      for (int64_t r = ogrp; r < n; r += ongrp) {
        double2 sum = make_double2(0.0, 0.0);
        for (int64_t k = ob.rowptr[r] + olane; k < ob.rowptr[r + 1]; k += TO) cfma(sum, ob.vals[k], psi[ob.cols[k]]);
        for (int off = TO >> 1; off > 0; off >>= 1) {
          sum.x += __shfl_xor(sum.x, off);
          sum.y += __shfl_xor(sum.y, off);
        }
        if (olane == 0) {
          const double2 d = cconj_mul(psi[r], sum);
          part.x += d.x;
          part.y += d.y;
        }
      }
      part = small_block_sum(part, red);
      if (tid == 0) s.expvals[(size_t)row * s.nobs + o] = part;
    }
    if (s.states)
      for (int64_t i = tid; i < n; i += kSmallThreads) s.states[(size_t)row * n + i] = psi[i];
  };
  record(A, 0);

  double2* v0 = A;   // holds Psi at the start of every step
  double2* v1 = B;
  for (int step = 0; step < s.nsteps; ++step) {
    // evaluate!(G, tlist, n): values of this interval      src/pwc_utils.jl:86-92
    if (s.ncoeffs > 0 || step == 0) {
      if (tid < s.nops) {
        double2 cl = s.scale;
        if (tid >= drift) cl = cmul(cl, s.table[(size_t)step * s.ncoeffs + (tid - drift)]);
        coef[tid] = cl;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NR; ++e) rv[e] = make_double2(0.0, 0.0);
      for (int l = 0; l < s.nops; ++l) {
        const double2* pl = s.planes[l];
        const double2 cl = coef[l];
#pragma unroll
        for (int e0 = 0; e0 < NS; e0 += CHK) {
          double2 v[CHK];
#pragma unroll
          for (int u = 0; u < CHK; ++u) {   // independent loads, all in flight together
            const int e = e0 + u;
            int pos1;
            if (PACK) pos1 = (rc[e] >> 12) & 0x7ffff;
            else pos1 = rm[e] > 0 ? rm[e] : -rm[e];
            v[u] = pos1 != 0 ? pl[pos1 - 1] : make_double2(0.0, 0.0);
          }
#pragma unroll
          for (int u = 0; u < CHK; ++u) {
            const int e = e0 + u;
            if (PACK ? (rc[e] < 0) : (rm[e] < 0)) v[u].y = -v[u].y;
            if (e < NR) {
              cfma(rv[e], cl, v[u]);
            } else {   // own slot of this thread only: no barrier needed
              double2 acc = (l == 0) ? make_double2(0.0, 0.0) : vlds[(size_t)(e - NR) * kSmallThreads + tid];
              cfma(acc, cl, v[u]);
              vlds[(size_t)(e - NR) * kSmallThreads + tid] = acc;
            }
          }
          if (PACK) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    double2 c = s.c;
    double2* x = v0;    // gathered vector (v1 of the recurrence; Psi for the first term)
    double2* ob = v1;   // holds v0 of the recurrence, overwritten in place by v2
    for (int m = 1; m <= nterms; ++m) {
      const bool last = (m == nterms);
      const double am = s.a[m];
      double2 chk = make_double2(0.0, 0.0);
      double nrm = 0.0;
      double2 sum = make_double2(0.0, 0.0);
#pragma unroll
      for (int e = 0; e < NS; ++e) {
        {
          cfma(sum, e < NR ? rv[e] : vlds[(size_t)(e - NR) * kSmallThreads + tid], x[col_of(e)]);
          if (PACK && (e % CHK) == CHK - 1) __builtin_amdgcn_sched_barrier(0);   // at most CHK gathers in flight
          if ((e + 1) % E == 0) {   // the row is complete
            for (int off = T >> 1; off > 0; off >>= 1) {
              sum.x += __shfl_xor(sum.x, off);
              sum.y += __shfl_xor(sum.y, off);
            }
            const int64_t r = grp + (int64_t)(e / E) * ngrp;
            if (lane == 0 && r < n) {
              const double2 xi = x[r];
              // t = c (H x - beta x) [+ v0]                  src/cheby.jl:178-179, :192-193, :202
              double2 t = make_double2(fma(-s.beta, xi.x, sum.x), fma(-s.beta, xi.y, sum.y));
              t = cmul(c, t);
              if (s.check && m >= 2) {                        // :194-200
                const double2 d = cconj_mul(xi, t);
                chk.x += d.x;
                chk.y += d.y;
                nrm += xi.x * xi.x + xi.y * xi.y;
              }
              double2 acc;
              if (m == 1) {
                acc = make_double2(s.a[0] * xi.x, s.a[0] * xi.y);   // lmul!(a[1], Psi)  :172
              } else {
                const double2 o = ob[r];
                t.x += o.x;
                t.y += o.y;
                acc = ACC[r];
              }
              acc.x = fma(am, t.x, acc.x);                    // axpy!(a[i], v, Psi)  :182, :205
              acc.y = fma(am, t.y, acc.y);
              if (last) {
                ob[r] = cmul(s.phase, acc);                   // lmul!(exp(-i beta dt), Psi)  :211
              } else {
                ob[r] = t;
                ACC[r] = acc;
              }
            }
            sum = make_double2(0.0, 0.0);
          }
        }
      }
      if (s.check && m >= 2) {
        const double2 cs = small_block_sum(chk, red);
        const double2 ns = small_block_sum(make_double2(nrm, 0.0), red);
        if (tid == 0 && !(hypot(cs.x, cs.y) / (2 * ns.x) <= 1.0 + s.limit) && s.fail[0] == 0) {
          s.fail[0] = 1;
          s.fail[1] = step;
          s.fail[2] = m;
        }
      }
      __syncthreads();
      if (m == 1) {
        c.x *= 2.0;                                            // :184
        c.y *= 2.0;
      }
      double2* tmp = x;
      x = ob;
      ob = tmp;
    }
    // the new Psi was written to the last `ob`, which the swap above left in `x`
    v1 = ob;
    v0 = x;
    record(v0, step + 1);
  }
  __syncthreads();
  for (int64_t i = tid; i < n; i += kSmallThreads) s.psi[i] = v0[i];
}

// ---------------------------------------------------------------------------
// arnoldi! for small systems: m columns = m mat-vecs + m (m + 1) / 2 projections + m norms,
// about 5 m launches on the general path, each bound by its launch.  Here: one workgroup,
// the operator in registers, the Krylov basis and the work vector in LDS, modified
// Gram-Schmidt in the reference's order (src/arnoldi.jl:82-97).
// ---------------------------------------------------------------------------
template <int E, int R>
__global__ __launch_bounds__(kSmallThreads) void arnoldi_small_kernel(SmallArnoldiArgs s) {
  constexpr int NS = E * R;
  extern __shared__ double2 small_lds[];
  double2* red = small_lds;
  double2* QL = small_lds + kSmallThreads / 64;   // [m + 1][n]
  const int64_t n = s.n;
  double2* W = QL + (size_t)(s.m + 1) * n;
  const int T = s.lanes;
  const int tid = threadIdx.x;
  const int lane = tid & (T - 1);
  const int64_t grp = tid / T, ngrp = kSmallThreads / T;

  int32_t rc[NS];
  double2 rv[NS];
#pragma unroll
  for (int e = 0; e < NS; ++e) {
    rc[e] = 0;
    rv[e] = make_double2(0.0, 0.0);
    const int64_t r = grp + (int64_t)(e / E) * ngrp;
    if (r < n) {
      const int64_t k = s.rowptr[r] + lane + (int64_t)(e % E) * T;
      if (k < s.rowptr[r + 1]) {
        rc[e] = s.cols[k];
        const int64_t mp = s.map[k];
        double2 v = s.vals[mp >= 0 ? mp : -mp - 1];
        if (mp < 0) v.y = -v.y;
        rv[e] = v;
      }
    }
  }
  // fill!(Hess, 0) :78 (the caller reads the whole matrix back)
  for (int i = tid; i < s.ldd * s.ldd; i += kSmallThreads) s.hess[i] = make_double2(0.0, 0.0);
  for (int i = tid; i < s.ldd; i += kSmallThreads) s.norms[i] = 0.0;
  double inv0 = 1.0;
  if (s.normalize_start) {   // newton! :271-272: beta = |Psi|, v = Psi / beta
    double nrm = 0.0;
    for (int64_t i = tid; i < n; i += kSmallThreads) {
      const double2 v = s.start[i];
      nrm += v.x * v.x + v.y * v.y;
    }
    const double beta0 = sqrt(small_block_sum(make_double2(nrm, 0.0), red).x);
    inv0 = 1.0 / beta0;
    if (tid == 0) s.norms[s.ldd - 1] = beta0;
  }
  for (int64_t i = tid; i < n; i += kSmallThreads) {   // q_0 = start   :79
    double2 v = s.start[i];
    v.x *= inv0;
    v.y *= inv0;
    QL[i] = v;
    s.Q[i] = v;
  }
  __syncthreads();

  for (int j = 0; j < s.m; ++j) {
    const double2* x = QL + (size_t)j * n;
    double2* hcol = s.hess + (size_t)j * s.ldd;
    // W = H q_j                                             :82
    double2 sum = make_double2(0.0, 0.0);
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      cfma(sum, rv[e], x[rc[e]]);
      if ((e + 1) % E == 0) {
        for (int off = T >> 1; off > 0; off >>= 1) {
          sum.x += __shfl_xor(sum.x, off);
          sum.y += __shfl_xor(sum.y, off);
        }
        const int64_t r = grp + (int64_t)(e / E) * ngrp;
        if (lane == 0 && r < n) W[r] = sum;
        sum = make_double2(0.0, 0.0);
      }
    }
    __syncthreads();
    // Hess[i,j] = dt <q_i|W>;  W -= (Hess[i,j] / dt) q_i     :84-87
    for (int i = 0; i <= j; ++i) {
      const double2* qi = QL + (size_t)i * n;
      double2 part = make_double2(0.0, 0.0);
      for (int64_t e = tid; e < n; e += kSmallThreads) {
        const double2 d = cconj_mul(qi[e], W[e]);
        part.x += d.x;
        part.y += d.y;
      }
      const double2 h = small_block_sum(part, red);
      const double2 hd = make_double2(s.dt * h.x, s.dt * h.y);
      if (tid == 0) hcol[i] = hd;
      const double2 coef = make_double2(-hd.x / s.dt, -hd.y / s.dt);
      for (int64_t e = tid; e < n; e += kSmallThreads) {
        double2 r = W[e];
        cfma(r, coef, qi[e]);
        W[e] = r;
      }
    }
    double nrm = 0.0;
    for (int64_t e = tid; e < n; e += kSmallThreads) {
      const double2 r = W[e];
      nrm += r.x * r.x + r.y * r.y;
    }
    const double hn = sqrt(small_block_sum(make_double2(nrm, 0.0), red).x);   // :88
    bool stop = false;
    double inv = 1.0;
    if ((j + 1 < s.m) || s.extended) {                        // :88-97
      if (tid == 0) {
        hcol[j + 1] = make_double2(s.dt * hn, 0.0);
        s.norms[j] = hn;
      }
      if (hn < s.norm_min) {
        stop = true;                                          // dimensionality exhausted  :91-95
      } else {
        inv = 1.0 / hn;
      }
    }
    double2* qn = QL + (size_t)(j + 1) * n;
    for (int64_t e = tid; e < n; e += kSmallThreads) {
      double2 r = W[e];
      r.x *= inv;
      r.y *= inv;
      qn[e] = r;
      s.Q[(size_t)(j + 1) * n + e] = r;
    }
    if (stop) break;
    __syncthreads();
  }
}

int launch_arnoldi_small(hipStream_t s, const SmallArnoldiArgs& a, Stats* st) {
  const size_t lds = sizeof(double2) * (kSmallThreads / 64 + (size_t)(a.m + 2) * (size_t)a.n);
  void (*kern)(SmallArnoldiArgs) = nullptr;
  switch (a.ent * 32 + a.rows_per_group) {
    case 1 * 32 + 1: kern = arnoldi_small_kernel<1, 1>; break;
    case 2 * 32 + 1: kern = arnoldi_small_kernel<2, 1>; break;
    case 4 * 32 + 1: kern = arnoldi_small_kernel<4, 1>; break;
    case 8 * 32 + 1: kern = arnoldi_small_kernel<8, 1>; break;
    case 1 * 32 + 2: kern = arnoldi_small_kernel<1, 2>; break;
    case 2 * 32 + 2: kern = arnoldi_small_kernel<2, 2>; break;
    case 4 * 32 + 2: kern = arnoldi_small_kernel<4, 2>; break;
    case 1 * 32 + 4: kern = arnoldi_small_kernel<1, 4>; break;
    case 2 * 32 + 4: kern = arnoldi_small_kernel<2, 4>; break;
    case 1 * 32 + 8: kern = arnoldi_small_kernel<1, 8>; break;
    case 16 * 32 + 1: kern = arnoldi_small_kernel<16, 1>; break;
    case 8 * 32 + 2: kern = arnoldi_small_kernel<8, 2>; break;
    case 4 * 32 + 4: kern = arnoldi_small_kernel<4, 4>; break;
    case 2 * 32 + 8: kern = arnoldi_small_kernel<2, 8>; break;
    case 1 * 32 + 16: kern = arnoldi_small_kernel<1, 16>; break;
    case 32 * 32 + 1: kern = arnoldi_small_kernel<32, 1>; break;
    case 16 * 32 + 2: kern = arnoldi_small_kernel<16, 2>; break;
    case 8 * 32 + 4: kern = arnoldi_small_kernel<8, 4>; break;
    case 4 * 32 + 8: kern = arnoldi_small_kernel<4, 8>; break;
    case 2 * 32 + 16: kern = arnoldi_small_kernel<2, 16>; break;
    case 1 * 32 + 32: kern = arnoldi_small_kernel<1, 32>; break;
    default: return fail(QP_E_BAD_ARG, "small plan (%d entries, %d rows per group) has no kernel", a.ent, a.rows_per_group);
  }
  if (lds > 48 * 1024)
    QP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(1), dim3(kSmallThreads), lds, s, a);
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec += a.m;
  }
  return QP_OK;
}

static_assert(kSmallThreads * kSmallEpt == 8192, "Tuning::small_nnz default = one register slot set (x2 for the 32-slot variants)");

int* tuning_field(Tuning& t, const char* key) {
  struct Entry {
    const char* name;
    int Tuning::*field;
  };
  static const Entry table[] = {
      {"rbcsr_variant", &Tuning::rbcsr_variant}, {"hrb_lower_last", &Tuning::hrb_lower_last},
      {"arnoldi_mode", &Tuning::arnoldi_mode},   {"split_mode", &Tuning::split_mode},
      {"arnoldi_fold", &Tuning::arnoldi_fold},   {"spmm_wg", &Tuning::spmm_wg},   {"hrb_wg", &Tuning::hrb_wg},   {"arnoldi_solve", &Tuning::arnoldi_solve},   {"arnoldi_fuse_dots", &Tuning::arnoldi_fuse_dots},   {"lattice_fill", &Tuning::lattice_fill},   {"sparse_controls", &Tuning::sparse_controls},
      {"liouville_fused_n", &Tuning::liouville_fused_n}, {"liouville_tile32_n", &Tuning::liouville_tile32_n}, {"liouville_tile32_min_n", &Tuning::liouville_tile32_min_n}, {"real_vals", &Tuning::real_vals},
      {"stencil", &Tuning::stencil}, {"block_map", &Tuning::block_map},             {"acc_defer", &Tuning::acc_defer},
      {"cheby_graph", &Tuning::cheby_graph},     {"small_nnz", &Tuning::small_nnz},
      {"roctx", &Tuning::roctx}, {"newton_graph", &Tuning::newton_graph}, {"arnoldi_l2_order", &Tuning::arnoldi_l2_order}, {"arnoldi_nt", &Tuning::arnoldi_nt},
      {"colblock", &Tuning::colblock}, {"cb_log2w", &Tuning::cb_log2w}, {"cb_min_log2n", &Tuning::cb_min_log2n}, {"cb_waves", &Tuning::cb_waves}, {"cb_rpt", &Tuning::cb_rpt},
      {"dense_auto", &Tuning::dense_auto},       {"dense_min_density_pct", &Tuning::dense_min_density_pct}, {"dense_panel_mfma", &Tuning::dense_panel_mfma},
      {"newton_pipeline", &Tuning::newton_pipeline}, {"spmm_tile", &Tuning::spmm_tile},
      {"spmm_nt", &Tuning::spmm_nt},             {"spmm_rows", &Tuning::spmm_rows},
      {"spmm_strip", &Tuning::spmm_strip},       {"spmm_rw", &Tuning::spmm_rw},
      {"hrb_walk", &Tuning::hrb_walk},           {"walk_waves", &Tuning::walk_waves},
      {"walk_min_blocks", &Tuning::walk_min_blocks}, {"walk_dbg", &Tuning::walk_dbg}, {"walk_nt", &Tuning::walk_nt}, {"walk_wg", &Tuning::walk_wg}, {"walk_reserve_cu", &Tuning::walk_reserve_cu}, {"walk_edge_steps", &Tuning::walk_edge_steps}, {"split_spin_log2", &Tuning::split_spin_log2}, {"split_dbg", &Tuning::split_dbg}, {"spmm_walk", &Tuning::spmm_walk}, {"spmm_walk_waves", &Tuning::spmm_walk_waves},
  };
  for (const Entry& e : table)
    if (std::strcmp(e.name, key) == 0) return &(t.*(e.field));
  return nullptr;
}

// lanes per row, entries per lane and rows per lane group such that the whole matrix is
// register-resident; false when the system does not fit (the caller then runs the general loop)
bool small_plan(int64_t n, int64_t maxrow, SmallArgs* a, int max_slots) {
  if (n < 1 || n > kSmallLdsRows) return false;
  for (int t = 1; t <= 64; t <<= 1) {
    const int64_t ngrp = kSmallThreads / t;
    const int64_t rows = (n + ngrp - 1) / ngrp;
    int64_t ent = 1;
    while (ent * t < maxrow) ent <<= 1;   // compile-time variants: 1, 2, 4, 8, 16 (Arnoldi: also 32)
    int64_t rows_p2 = 1;
    while (rows_p2 < rows) rows_p2 <<= 1;
    if (rows_p2 * ent <= max_slots) {   // smallest t: fewest cross-lane reduction levels
      a->lanes = t;
      a->ent = (int)ent;
      a->rows_per_group = (int)rows_p2;
      int to = 1;
      while (to < 64 && (int64_t)kSmallThreads / (2 * to) >= n) to <<= 1;
      a->obs_lanes = to;
      return true;
    }
  }
  return false;
}

int launch_cheby_propagate_small(hipStream_t s, const SmallArgs& a, Stats* st) {
  const int slots = a.ent * a.rows_per_group;
  const size_t lds = sizeof(double2) * (kSmallThreads / 64 + (size_t)a.nops + 3 * (size_t)a.n +
                                        (slots > 16 ? (size_t)(slots - 16) * kSmallThreads : 0));
  void (*kern)(SmallArgs) = nullptr;
  switch (a.ent * 32 + a.rows_per_group) {
    case 1 * 32 + 1: kern = cheby_propagate_small_kernel<1, 1>; break;
    case 2 * 32 + 1: kern = cheby_propagate_small_kernel<2, 1>; break;
    case 4 * 32 + 1: kern = cheby_propagate_small_kernel<4, 1>; break;
    case 8 * 32 + 1: kern = cheby_propagate_small_kernel<8, 1>; break;
    case 1 * 32 + 2: kern = cheby_propagate_small_kernel<1, 2>; break;
    case 2 * 32 + 2: kern = cheby_propagate_small_kernel<2, 2>; break;
    case 4 * 32 + 2: kern = cheby_propagate_small_kernel<4, 2>; break;
    case 1 * 32 + 4: kern = cheby_propagate_small_kernel<1, 4>; break;
    case 2 * 32 + 4: kern = cheby_propagate_small_kernel<2, 4>; break;
    case 1 * 32 + 8: kern = cheby_propagate_small_kernel<1, 8>; break;
    case 16 * 32 + 1: kern = cheby_propagate_small_kernel<16, 1>; break;
    case 8 * 32 + 2: kern = cheby_propagate_small_kernel<8, 2>; break;
    case 4 * 32 + 4: kern = cheby_propagate_small_kernel<4, 4>; break;
    case 2 * 32 + 8: kern = cheby_propagate_small_kernel<2, 8>; break;
    case 1 * 32 + 16: kern = cheby_propagate_small_kernel<1, 16>; break;
    case 32 * 32 + 1: kern = cheby_propagate_small_kernel<32, 1>; break;
    case 16 * 32 + 2: kern = cheby_propagate_small_kernel<16, 2>; break;
    case 8 * 32 + 4: kern = cheby_propagate_small_kernel<8, 4>; break;
    case 4 * 32 + 8: kern = cheby_propagate_small_kernel<4, 8>; break;
    case 2 * 32 + 16: kern = cheby_propagate_small_kernel<2, 16>; break;
    case 1 * 32 + 32: kern = cheby_propagate_small_kernel<1, 32>; break;
    default: return fail(QP_E_BAD_ARG, "small plan (%d entries, %d rows per group) has no kernel", a.ent, a.rows_per_group);
  }
  if (lds > 48 * 1024)
    QP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(1), dim3(kSmallThreads), lds, s, a);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

int spmv_grid_size(const DevMatrix& A) {
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
  // Eight instead of four row blocks per workgroup (knob hrb_wg) for the fused Chebyshev term wherever nothing counts
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
      if (tun.hrb_wg == 8 && wide_ok && (tun.rbcsr_variant & 7) == 7 && A.stored > A.nblocks * (int64_t)(kRB * 8)) {
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
                         nblk, A.nrows, op, bmap, sy, tun.hrb_lower_last);                                 \
    else                                                                                                 \
      hipLaunchKernelGGL((hrb_spmv_kernel<Op, VV, double2>), dim3(grid), dim3(kThreads), 0, s, A.bptr, A.cmeta, \
                         reinterpret_cast<const char*>(A.cols), A.vals, A.lptr, A.lcmeta,                \
                         reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, \
                         nblk, A.nrows, op, bmap, sy, tun.hrb_lower_last);                                 \
    break;
    if constexpr (std::is_same<Op, ChebyOp>::value) {
      // a lattice operator (one stencil repeated down the row blocks): the strip walk (kernels_walk.hip) -- whole operator,
      // no normalisation check, the gathered vector's own rows being the row-local operand; same sums as the kernels below
      // ... or the interior launch of a split term whose row set carries a plan of its own (RowSet::walk)
      const bool whole = !rs && A.walk && A.walk->valid;
      const bool set_walk = rs && rs->walk && rs->walk->valid && !rs->sync.signal;
      if (tun.hrb_walk && (whole || set_walk) && wide_ok && !tun.hrb_lower_last && (tun.rbcsr_variant & 31) == 15 &&
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
      if (tun.hrb_wg == 8 && wide_ok && (tun.rbcsr_variant & 31) == 15) {
        const int g8 = (int)((nblk + 7) / 8);
        if (A.vals_r)
          hipLaunchKernelGGL((hrb_spmv_kernel<Op, 15, double, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals_r, A.lptr, A.lcmeta,
                             reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, nblk, A.nrows, op,
                             bmap, sy8, tun.hrb_lower_last);
        else
          hipLaunchKernelGGL((hrb_spmv_kernel<Op, 15, double2, 8>), dim3(g8), dim3(512), 0, s, A.bptr, A.cmeta,
                             reinterpret_cast<const char*>(A.cols), A.vals, A.lptr, A.lcmeta,
                             reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos), x, nblk, A.nrows, op,
                             bmap, sy8, tun.hrb_lower_last);
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

// ---------------------------------------------------------------------------
// elementwise / BLAS-1
// ---------------------------------------------------------------------------
static inline int ew_grid(int64_t n) {
  int64_t g = (n + kThreads - 1) / kThreads;
  if (g > 256 * 8) g = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

// coefficients travel as kernel arguments: no staging buffer that a later
// set_coeffs() could overwrite while an earlier combine is still queued
__global__ __launch_bounds__(kThreads) void combine_planes_kernel(double2* __restrict__ vals,
                                                                  const double2* const* __restrict__ planes,
                                                                  CoefBlock coefs, int first, int nplanes,
                                                                  int accumulate, int64_t n, double* __restrict__ vals_r) {
  for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < n; p += (int64_t)gridDim.x * kThreads) {
    double2 acc = accumulate ? vals[p] : make_double2(0.0, 0.0);
    for (int l = 0; l < nplanes; ++l) cfma(acc, coefs.c[l], planes[first + l][p]);
    vals[p] = acc;
    if (vals_r) vals_r[p] = acc.x;   // real copy for the mat-vec kernels of an all-real operator
  }
}

// evaluate! for sparse trailing control terms: only the positions they touch are rewritten, in the summation order of
// combine_planes_kernel (the sum over the earlier planes is `base`)
__global__ __launch_bounds__(kThreads) void sparse_planes_update_kernel(double2* __restrict__ vals, const double2* __restrict__ base,
                                                                        const int32_t* __restrict__ support, int64_t n_support,
                                                                        const double2* __restrict__ support_vals, CoefBlock coefs,
                                                                        int nplanes, double* __restrict__ vals_r) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_support; i += (int64_t)gridDim.x * kThreads) {
    const int32_t p = support[i];
    double2 acc = base[p];
    for (int l = 0; l < nplanes; ++l) cfma(acc, coefs.c[l], support_vals[(size_t)l * n_support + i]);
    vals[p] = acc;
    if (vals_r) vals_r[p] = acc.x;
  }
}

int launch_sparse_planes_update(hipStream_t s, double2* vals, const double2* base, const int32_t* support, int64_t n_support,
                                const double2* support_vals, int nplanes, const double2* coefs, double* vals_r, Stats* st) {
  if (n_support == 0) return QP_OK;
  if (nplanes > kCoefBlock) return fail(QP_E_BAD_ARG, "more than %d sparse control terms", kCoefBlock);
  CoefBlock cb;
  for (int l = 0; l < nplanes; ++l) cb.c[l] = coefs[l];
  hipLaunchKernelGGL(sparse_planes_update_kernel, dim3(ew_grid(n_support)), dim3(kThreads), 0, s, vals, base, support, n_support,
                     support_vals, cb, nplanes, vals_r);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

__global__ __launch_bounds__(kThreads) void real_part_kernel(double* __restrict__ out, const double2* __restrict__ v,
                                                             int64_t n) {
  for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < n; p += (int64_t)gridDim.x * kThreads) out[p] = v[p].x;
}

int launch_real_part(hipStream_t s, double* out, const double2* v, int64_t n, Stats* st) {
  if (n == 0) return QP_OK;
  hipLaunchKernelGGL(real_part_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, out, v, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

int launch_combine_planes(hipStream_t s, double2* vals, const double2* const* planes_dev, const double2* coefs,
                          int nplanes, int64_t n, double* vals_r, Stats* st) {
  if (n == 0) return QP_OK;
  for (int first = 0; first < nplanes; first += kCoefBlock) {
    CoefBlock cb;
    const int cnt = (nplanes - first < kCoefBlock) ? nplanes - first : kCoefBlock;
    for (int l = 0; l < cnt; ++l) cb.c[l] = coefs[first + l];
    const bool last_chunk = first + cnt >= nplanes;
    hipLaunchKernelGGL(combine_planes_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, vals, planes_dev, cb, first,
                       cnt, first > 0 ? 1 : 0, n, last_chunk ? vals_r : nullptr);
    QP_HIP(hipGetLastError());
    if (st) st->n_launch++;
  }
  return QP_OK;
}

__global__ __launch_bounds__(kThreads) void fill_kernel(double2* __restrict__ x, double2 a, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) x[i] = a;
}
__global__ __launch_bounds__(kThreads) void scal_kernel(double2* __restrict__ x, double2 a, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    x[i] = cmul(a, x[i]);
}
__global__ __launch_bounds__(kThreads) void axpy_kernel(double2 a, const double2* __restrict__ x,
                                                        double2* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    double2 r = y[i];
    cfma(r, a, x[i]);
    y[i] = r;
  }
}

int launch_fill(hipStream_t s, double2* x, double2 a, int64_t n, Stats* st) {
  if (n == 0) return QP_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, x, a, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}
int launch_scal(hipStream_t s, double2* x, double2 a, int64_t n, Stats* st) {
  if (n == 0) return QP_OK;
  hipLaunchKernelGGL(scal_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, x, a, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}
int launch_axpy(hipStream_t s, double2 a, const double2* x, double2* y, int64_t n, Stats* st) {
  if (n == 0) return QP_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, a, x, y, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

// reductions run on a fixed grid of kRedBlocks workgroups: partial b covers the
// elements i with (i / kThreads) % kRedBlocks == b, summed in a fixed order
__global__ __launch_bounds__(kThreads) void dot_partials_kernel(const double2* __restrict__ x,
                                                                const double2* __restrict__ y,
                                                                double2* __restrict__ partials, int64_t n) {
  __shared__ double2 lds[kThreads / 64];
  double2 acc = make_double2(0.0, 0.0);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)kRedBlocks * kThreads) {
    const double2 d = cconj_mul(x[i], y[i]);
    acc.x += d.x;
    acc.y += d.y;
  }
  acc = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

int launch_dot_partials(hipStream_t s, const double2* x, const double2* y, double2* partials, int64_t n,
                        Stats* st) {
  hipLaunchKernelGGL(dot_partials_kernel, dim3(kRedBlocks), dim3(kThreads), 0, s, x, y, partials, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

// every workgroup re-reduces the previous kernel's kRedBlocks partials (one per
// thread, kRedBlocks == kThreads) -- "combine in the next kernel's prologue"
__device__ __forceinline__ double2 reduce_partials(const double2* __restrict__ part, double2* lds) {
  static_assert(kRedBlocks == kThreads, "one partial per thread");
  return block_sum(part[threadIdx.x], lds);
}

__global__ __launch_bounds__(kThreads) void mgs_pass_kernel(MgsArgs a) {
  __shared__ double2 lds[kThreads / 64];
  double2 coef = make_double2(0.0, 0.0);
  if (a.q_prev) {
    const double2 h = reduce_partials(a.part_in, lds);
    // Hess[i,j] = dt <q_i|q_j+1>;  axpy!(-Hess[i,j]/dt, q_i, q_j+1)   src/arnoldi.jl:85-86
    const double2 hd = make_double2(a.dt * h.x, a.dt * h.y);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.hess_prev) *a.hess_prev = hd;
    coef = make_double2(-hd.x / a.dt, -hd.y / a.dt);
  }
  double2 acc = make_double2(0.0, 0.0);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < a.n; i += (int64_t)kRedBlocks * kThreads) {
    double2 w = a.w[i];
    if (a.q_prev) {
      cfma(w, coef, a.q_prev[i]);
      a.w[i] = w;
    }
    if (a.q_cur) {
      const double2 d = cconj_mul(a.q_cur[i], w);
      acc.x += d.x;
      acc.y += d.y;
    } else {
      acc.x += w.x * w.x + w.y * w.y;
    }
  }
  acc = block_sum(acc, lds);
  if (threadIdx.x == 0) a.part_out[blockIdx.x] = acc;
}

int launch_mgs_pass(hipStream_t s, const MgsArgs& a, Stats* st) {
  hipLaunchKernelGGL(mgs_pass_kernel, dim3(kRedBlocks), dim3(kThreads), 0, s, a);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

__global__ __launch_bounds__(kThreads) void norm_scale_kernel(double2* __restrict__ w,
                                                              const double2* __restrict__ part_in,
                                                              double2* hess_slot, double dt, int64_t n) {
  __shared__ double2 lds[kThreads / 64];
  const double2 s2 = reduce_partials(part_in, lds);
  const double h = sqrt(s2.x);  // h = norm(q[j+1])          src/arnoldi.jl:89
  if (blockIdx.x == 0 && threadIdx.x == 0 && hess_slot) *hess_slot = make_double2(dt * h, 0.0);  // :90
  const double inv = 1.0 / h;   // lmul!(1 / h, q[j+1])       :96
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    double2 v = w[i];
    v.x *= inv;
    v.y *= inv;
    w[i] = v;
  }
}

int launch_norm_scale(hipStream_t s, double2* w, const double2* part_in, double2* hess_slot, double dt,
                      int64_t n, Stats* st) {
  hipLaunchKernelGGL(norm_scale_kernel, dim3(ew_grid(n)), dim3(kThreads), 0, s, w, part_in, hess_slot, dt, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

__global__ __launch_bounds__(kThreads) void combine_vecs_kernel(double2* __restrict__ out, int use_out, double2 s0,
                                                                const double2* __restrict__ Q, int64_t ldq, int m,
                                                                CoefBlock coefs,
                                                                double2* __restrict__ norm_partials, int64_t n) {
  __shared__ double2 lds[kThreads / 64];
  double nrm = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)kRedBlocks * kThreads) {
    double2 r = make_double2(0.0, 0.0);
    if (use_out) r = cmul(s0, out[i]);
    for (int k = 0; k < m; ++k) cfma(r, coefs.c[k], Q[(size_t)k * ldq + i]);
    out[i] = r;
    nrm += r.x * r.x + r.y * r.y;
  }
  if (norm_partials) {
    const double2 t = block_sum(make_double2(nrm, 0.0), lds);
    if (threadIdx.x == 0) norm_partials[blockIdx.x] = t;
  }
}

// Two combinations of the same basis in one pass over Q (newton!: Psi += sum_i P_i q_i and the next
// restart vector v = sum_i R_i q_i, src/newton.jl:346-367): out1 = (use_out1 ? out1 : 0) + sum_{k<m1}
// c1_k q_k with |out1|^2 partials, out2 = sum_{k<m2} c2_k q_k; each output sees its terms in the
// order of the single-output kernel.  Two elements per lane and four basis vectors per round in flight.
__global__ __launch_bounds__(kThreads) void combine2_vecs_kernel(double2* __restrict__ out1, int use_out1, int m1,
                                                                 CoefBlock c1, double2* __restrict__ out2, int m2,
                                                                 CoefBlock c2, const double2* __restrict__ Q, int64_t ldq,
                                                                 double2* __restrict__ norm_partials, int64_t n) {
  __shared__ double2 lds[kThreads / 64];
  double nrm = 0.0;
  const int mm = m1 > m2 ? m1 : m2;
  const int64_t stride = (int64_t)kRedBlocks * kThreads;
  const double2 zero = make_double2(0.0, 0.0);
  for (int64_t e0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; e0 < n; e0 += 2 * stride) {
    const int64_t e1 = e0 + stride;
    const bool two = e1 < n;
    double2 a0 = use_out1 ? out1[e0] : zero, a1 = (use_out1 && two) ? out1[e1] : zero;
    double2 b0 = zero, b1 = zero;
    int k = 0;
    for (; k + 3 < mm; k += 4) {
      double2 q0[4], q1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        q0[t] = Q[(size_t)(k + t) * ldq + e0];
        q1[t] = two ? Q[(size_t)(k + t) * ldq + e1] : zero;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (k + t < m1) {
          cfma(a0, c1.c[k + t], q0[t]);
          cfma(a1, c1.c[k + t], q1[t]);
        }
        if (k + t < m2) {
          cfma(b0, c2.c[k + t], q0[t]);
          cfma(b1, c2.c[k + t], q1[t]);
        }
      }
    }
    for (; k < mm; ++k) {
      const double2 q0 = Q[(size_t)k * ldq + e0];
      const double2 q1 = two ? Q[(size_t)k * ldq + e1] : zero;
      if (k < m1) {
        cfma(a0, c1.c[k], q0);
        cfma(a1, c1.c[k], q1);
      }
      if (k < m2) {
        cfma(b0, c2.c[k], q0);
        cfma(b1, c2.c[k], q1);
      }
    }
    out1[e0] = a0;
    out2[e0] = b0;
    nrm += a0.x * a0.x + a0.y * a0.y;
    if (two) {
      out1[e1] = a1;
      out2[e1] = b1;
      nrm += a1.x * a1.x + a1.y * a1.y;
    }
  }
  const double2 t = block_sum(make_double2(nrm, 0.0), lds);
  if (threadIdx.x == 0) norm_partials[blockIdx.x] = t;
}

// false when one of the coefficient lists does not fit one launch (the caller then combines one by one)
bool launch_combine2_vecs(hipStream_t s, double2* out1, int use_out1, int m1, const double2* coefs1, double2* out2, int m2,
                          const double2* coefs2, const double2* Q, int64_t ldq, double2* norm_partials, int64_t n,
                          Stats* st) {
  if (m1 > kCoefBlock || m2 > kCoefBlock || m1 < 1 || m2 < 1) return false;
  CoefBlock c1, c2;
  for (int l = 0; l < m1; ++l) c1.c[l] = coefs1[l];
  for (int l = 0; l < m2; ++l) c2.c[l] = coefs2[l];
  hipLaunchKernelGGL(combine2_vecs_kernel, dim3(kRedBlocks), dim3(kThreads), 0, s, out1, use_out1, m1, c1, out2, m2, c2, Q, ldq,
                     norm_partials, n);
  if (st) st->n_launch++;
  return hipGetLastError() == hipSuccess;
}

int launch_combine_vecs(hipStream_t s, double2* out, int use_out, double2 s0, const double2* Q, int64_t ldq,
                        int m, const double2* coefs, double2* norm_partials, int64_t n, Stats* st) {
  for (int first = 0; first < m || first == 0; first += kCoefBlock) {
    CoefBlock cb;
    const int cnt = (m - first < kCoefBlock) ? m - first : kCoefBlock;
    for (int l = 0; l < cnt; ++l) cb.c[l] = coefs[first + l];
    const bool lastc = (first + cnt >= m);
    hipLaunchKernelGGL(combine_vecs_kernel, dim3(kRedBlocks), dim3(kThreads), 0, s, out, first > 0 ? 1 : use_out,
                       first > 0 ? make_double2(1.0, 0.0) : s0, Q + (size_t)first * ldq, ldq, cnt, cb,
                       lastc ? norm_partials : nullptr, n);
    QP_HIP(hipGetLastError());
    if (st) st->n_launch++;
    if (lastc) break;
  }
  return QP_OK;
}

// ---------------------------------------------------------------------------
// Low-synchronisation modified Gram-Schmidt (one column in 3 launches instead of j+2).
// MGS computes h_i = <q_i | w - sum_{k<i} h_k q_k> = c_i - sum_{k<i} <q_i|q_k> h_k with
// c = Q^H w: given the classical inner products c and the Gram rows <q_i|q_k> of the
// (not exactly orthogonal) basis, a triangular solve reproduces the MGS coefficients
// (exactly in exact arithmetic, to rounding in floating point), and the projections are
// then subtracted in the MGS order.  Kernel 1 (multidot) forms c and the new Gram row in
// one pass over Q; kernel 2 reduces the partials (one workgroup per value) and its last
// workgroup solves; kernel 3 subtracts and accumulates |w|^2.  (Row-partitioned runs need an
// all-reduce between the sums and the solve: there the solve is a launch of its own.)
// ---------------------------------------------------------------------------
constexpr int kTI = 8;  // basis vectors per multidot tile (16 complex accumulators per lane)

#include "mgs_common.h"   // tri_index, mgs_solve_wave, mgs_stage_gram

// LDS of the finishing workgroup: red[2(j+1)] | h[j+1] | Gt[j(j+1)/2]
__host__ __device__ inline size_t mgs_solve_lds(int j) {
  return sizeof(double2) * (size_t)(3 * (j + 1) + j * (j + 1) / 2);
}

// partials are stored value-major: partials[v * kRedBlocks + workgroup].
// BS threads per workgroup, EPL elements per lane and round (shipped: 256 x 2.  Tried: 1024-thread workgroups
// with one element per lane, for 16 instead of 4 wavefronts per CU while the basis is one tile wide -- slower,
// 1.31 instead of 1.06 ms per Arnoldi sweep at config C3: profiles/r02/newton_c3_notes.txt)
template <int BS, int EPL>
__global__ __launch_bounds__(BS) void multidot_kernel(const double2* __restrict__ Q, int64_t ldq, int j,
                                                      const double2* __restrict__ w,
                                                      double2* __restrict__ partials, int64_t n) {
  __shared__ double2 wsum[BS / 64][2 * kTI];
  const int i0 = blockIdx.y * kTI;
  double2 ac[kTI], ag[kTI];
#pragma unroll
  for (int t = 0; t < kTI; ++t) ac[t] = ag[t] = make_double2(0.0, 0.0);
  const double2* __restrict__ qj = Q + (size_t)j * ldq;
  // EPL (kTI + 2) loads in flight per lane; each accumulator adds its elements in ascending order
  const int64_t stride = (int64_t)kRedBlocks * BS;
  for (int64_t e0 = (int64_t)blockIdx.x * BS + threadIdx.x; e0 < n; e0 += EPL * stride) {
    const int64_t e1 = e0 + stride;
    const bool two = EPL == 2 && e1 < n;
    const double2 zero = make_double2(0.0, 0.0);
    const double2 wv0 = w[e0], qv0 = qj[e0];
    const double2 wv1 = two ? w[e1] : zero, qv1 = two ? qj[e1] : zero;
    double2 qa[kTI], qb[kTI];
#pragma unroll
    for (int t = 0; t < kTI; ++t) {
      if (i0 + t <= j) {
        qa[t] = Q[(size_t)(i0 + t) * ldq + e0];
        qb[t] = two ? Q[(size_t)(i0 + t) * ldq + e1] : zero;
      }
    }
#pragma unroll
    for (int t = 0; t < kTI; ++t) {
      if (i0 + t <= j) {
        const double2 a0 = cconj_mul(qa[t], wv0), b0 = cconj_mul(qa[t], qv0);
        ac[t].x += a0.x;
        ac[t].y += a0.y;
        ag[t].x += b0.x;
        ag[t].y += b0.y;
        if (two) {
          const double2 a1 = cconj_mul(qb[t], wv1), b1 = cconj_mul(qb[t], qv1);
          ac[t].x += a1.x;
          ac[t].y += a1.y;
          ag[t].x += b1.x;
          ag[t].y += b1.y;
        }
      }
    }
  }
  // workgroup sums in the order of block_sum, with one barrier for all 2 kTI values
  const int wvid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < kTI; ++t) {
    if (i0 + t <= j) {
      ac[t].x = wave_sum(ac[t].x);
      ac[t].y = wave_sum(ac[t].y);
      ag[t].x = wave_sum(ag[t].x);
      ag[t].y = wave_sum(ag[t].y);
      if (lane == 0) {
        wsum[wvid][t] = ac[t];
        wsum[wvid][kTI + t] = ag[t];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * kTI) {
    const int t = threadIdx.x % kTI;
    if (i0 + t <= j) {
      double2 r = wsum[0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < BS / 64; ++k) {
        r.x += wsum[k][threadIdx.x].x;
        r.y += wsum[k][threadIdx.x].y;
      }
      const int v = (threadIdx.x < kTI ? 0 : j + 1) + (i0 + t);
      partials[(size_t)v * kRedBlocks + blockIdx.x] = r;
    }
  }
}

// One workgroup per value: sum the kRedBlocks multidot partials in a fixed order.  In a
// row-partitioned run these are the sums over the local rows; the caller all-reduces
// `reduced` over the ranks before the solve consumes it (ticket == NULL).  On one GPU the
// workgroup that finishes last (agent-scope release / acquire around one counter) goes on to
// solve for the MGS coefficients; which workgroup that is does not influence any value.  Every
// workgroup starts by pulling the older Gram rows into LDS so that the finishing one has them.
__global__ __launch_bounds__(kThreads) void multidot_reduce_kernel(const double2* __restrict__ partials, int j,
                                                                   double2* __restrict__ reduced, unsigned* ticket,
                                                                   double2* __restrict__ G, int ldg,
                                                                   double2* __restrict__ hess_col,
                                                                   double2* __restrict__ coef, double dt) {
  extern __shared__ double2 dyn[];
  __shared__ double2 lds[kThreads / 64];
  static_assert(kRedBlocks == kThreads, "one partial per thread");
  const int nv = 2 * (j + 1);
  const int v = blockIdx.x;
  double2* red = dyn;
  double2* h = dyn + nv;
  double2* Gt = h + (j + 1);
  if (ticket) {
    const int older = j * (j - 1) / 2;   // rows 1 .. j-1
    for (int idx = threadIdx.x; idx < older; idx += kThreads) {
      int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)idx)) * 0.5f);
      while (i * (i - 1) / 2 > idx) --i;
      while ((i + 1) * i / 2 <= idx) ++i;
      Gt[idx] = G[(size_t)i * ldg + (idx - i * (i - 1) / 2)];
    }
  }
  const double2 s = block_sum(partials[(size_t)v * kRedBlocks + threadIdx.x], lds);
  if (threadIdx.x == 0) reduced[v] = s;
  if (!ticket) return;
  __shared__ unsigned s_last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t + 1 == gridDim.x) ? 1u : 0u;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!s_last) return;
  for (int k = threadIdx.x; k < nv; k += kThreads) red[k] = reduced[k];
  __syncthreads();
  for (int k = threadIdx.x; k < j; k += kThreads) {   // Gram row j = conj of the fresh <q_k|q_j>
    const double2 r = red[(j + 1) + k];
    const double2 g = make_double2(r.x, -r.y);
    Gt[tri_index(j, k)] = g;
    G[(size_t)j * ldg + k] = g;
  }
  __syncthreads();
  if (threadIdx.x < 64) mgs_solve_wave(j, red, Gt, h, hess_col, coef, dt);
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the solve as its own single-workgroup launch (row-partitioned runs: after the all-reduce)
__global__ __launch_bounds__(kThreads) void mgs_solve_kernel(int j, const double2* __restrict__ reduced,
                                                             double2* __restrict__ G, int ldg,
                                                             double2* __restrict__ hess_col, double2* __restrict__ coef,
                                                             double dt) {
  extern __shared__ double2 dyn[];
  const int nv = 2 * (j + 1);
  double2* red = dyn;
  double2* h = dyn + nv;
  double2* Gt = h + (j + 1);
  for (int v = threadIdx.x; v < nv; v += kThreads) red[v] = reduced[v];
  __syncthreads();
  mgs_stage_gram(j, red, Gt, G, ldg);
  __syncthreads();
  if (threadIdx.x < 64) mgs_solve_wave(j, red, Gt, h, hess_col, coef, dt);
}

// w += sum_i coef_i q_i in MGS order (coef_i = -h_i) and |w|^2 partials; EPL elements per lane and four
// basis vectors per round in flight; BS threads per workgroup (see multidot_kernel)
// SOLVE (one GPU, knob arnoldi_solve): every workgroup first sums the kRedBlocks multidot partials of all 2 (j + 1)
// values itself -- lane l adds partials l, l + 64, l + 128, l + 192, then the wavefront tree: a fixed order -- and
// solves for the MGS coefficients, redundantly but without the reduction launch in between; workgroup 0 records the
// Hessenberg column, every workgroup writes the (identical) new Gram row.
struct MgsSolveArgs {
  const double2* partials;
  double2* G;
  int ldg;
  double2* hess_col;
  double dt;
  unsigned* early_flag = nullptr;   // host-visible: the column's MGS coefficients (Hess[0..j, j]) are written
  unsigned flag_value = 0;
};

// ORD (knob arnoldi_l2_order): the elements a workgroup owns and the order in which it reads the basis are chosen for the
// XCD's L2.  The column's mat-vec with the dot products in its epilogue (kernels_arnoldi.hip) has just read the basis
// vectors q_0 .. q_j, ascending, on the rows of ITS workgroups -- rows [t * 512 grid + 512 wg, + 512) in round t, wg =
// xcd_remap(blockIdx) -- so each XCD's L2 holds the share of the LAST vectors of the LAST round.  ORD = true gives the
// projection the same rows per (remapped) workgroup and walks rounds and basis vectors back to front: what the dots pass
// read last is read first, out of L2 instead of the Infinity Cache; and it ends on q_0 of round 0, which is where the next
// column's dots pass begins.  The coefficients are the solved ones either way (the sum w - sum_i h_i q_i in another order:
// a rounding-level difference, deterministic).  ORD = false: the round-2 layout (element = blockIdx * BS + thread + k * 65536).
template <int BS, int EPL, bool SOLVE, bool ORD>
__global__ __launch_bounds__(BS) void mgs_update_kernel(double2* __restrict__ w, const double2* __restrict__ Q,
                                                        int64_t ldq, int j, const double2* __restrict__ coef,
                                                        double2* __restrict__ norm_partials, int64_t n, MgsSolveArgs sv) {
  extern __shared__ double2 sm[];  // [0, j+1): coefficients; [j+1, j+1+BS/64): reduction scratch; SOLVE: + red | hs | Gt | dummy
  double2* h = sm;
  double2* lds = sm + (j + 1);
  // the first round of the streams (this lane's elements of w and of the first four basis vectors) is requested BEFORE
  // the prologue below: the reduction + solve is a chain of L2 round trips and barriers (2-3 us) that needs no memory
  // pipe, and the coefficients are not needed before the first FMA
  static_assert(!ORD || EPL == 2, "the ordered form takes two elements per lane and round");
  const int64_t stride = (int64_t)kRedBlocks * BS;
  // ORD: rounds of gridDim.x * 2 BS elements, this workgroup's 2 BS of the LAST round first
  const int64_t per_round = (int64_t)gridDim.x * 2 * BS;
  const int64_t nrounds = ORD ? (n + per_round - 1) / per_round : 0;
  const int64_t ef0 = ORD ? (nrounds - 1) * per_round + (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 2 * BS + threadIdx.x
                          : (int64_t)blockIdx.x * BS + threadIdx.x;
  const int64_t ef1 = ORD ? ef0 + BS : ef0 + stride;
  const int pq0 = ORD ? j - 3 : 0;   // first of the four basis vectors requested ahead of the prologue
  const bool pre_on = ef0 < n, pre_two = EPL == 2 && ef1 < n, pre_q = j >= 3;
  double2 pr0 = make_double2(0.0, 0.0), pr1 = make_double2(0.0, 0.0), pa[4], pb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pa[t] = pb[t] = make_double2(0.0, 0.0);
  if (pre_on) {
    pr0 = w[ef0];
    if (pre_two) pr1 = w[ef1];
    if (pre_q) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pa[t] = Q[(size_t)(pq0 + t) * ldq + ef0];
        if (pre_two) pb[t] = Q[(size_t)(pq0 + t) * ldq + ef1];
      }
    }
  }
  if (SOLVE) {
    static_assert(!SOLVE || (BS == kThreads && kRedBlocks == 256), "four partials per lane");
    const int nv = 2 * (j + 1);
    double2* red = lds + BS / 64;
    double2* hs = red + nv;
    double2* Gt = hs + (j + 1);
    double2* dummy = Gt + j * (j + 1) / 2;   // hess column of the workgroups that do not record it
    const int wv0 = threadIdx.x >> 6, l0 = threadIdx.x & 63;
    constexpr int NW = BS / 64, UV = 4;   // UV values (16 loads per lane) in flight per wavefront and round
    for (int v0 = wv0; v0 < nv; v0 += NW * UV) {
      double2 q[UV][4];
#pragma unroll
      for (int u = 0; u < UV; ++u) {
        const int v = min(v0 + u * NW, nv - 1);
        const double2* __restrict__ pp = sv.partials + (size_t)v * kRedBlocks + l0;
        q[u][0] = pp[0];
        q[u][1] = pp[64];
        q[u][2] = pp[128];
        q[u][3] = pp[192];
      }
#pragma unroll
      for (int u = 0; u < UV; ++u) {
        const int v = v0 + u * NW;
        double2 r = make_double2(((q[u][0].x + q[u][1].x) + q[u][2].x) + q[u][3].x,
                                 ((q[u][0].y + q[u][1].y) + q[u][2].y) + q[u][3].y);
        r.x = wave_sum(r.x);
        r.y = wave_sum(r.y);
        if (l0 == 0 && v < nv) red[v] = r;
      }
    }
    __syncthreads();
    mgs_stage_gram(j, red, Gt, sv.G, sv.ldg);
    __syncthreads();
    if (threadIdx.x < 64) mgs_solve_wave(j, red, Gt, hs, blockIdx.x == 0 ? sv.hess_col : dummy, h, sv.dt);
    // (lane 0 of the wavefront that stored the column: its release covers those stores)
    if (sv.early_flag && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(sv.early_flag, sv.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
  } else {
    for (int i = threadIdx.x; i <= j; i += BS) h[i] = coef[i];
    __syncthreads();
  }
  double nrm = 0.0;
  if constexpr (ORD) {
    for (int64_t e0 = ef0; e0 >= 0; e0 -= per_round) {   // rounds back to front
      const int64_t e1 = e0 + BS;
      const bool on = e0 < n, two = e1 < n;
      const bool first = e0 == ef0;   // (the same for every lane of the workgroup)
      if (!on) continue;              // (only in the last round, which comes first: lanes past the end)
      double2 r0 = first ? pr0 : w[e0];
      double2 r1 = first ? pr1 : (two ? w[e1] : make_double2(0.0, 0.0));
      int i = j;
      for (; i >= 3; i -= 4) {        // q_i, q_{i-1}, q_{i-2}, q_{i-3}: loaded as [i-3 .. i], applied from i downwards
        double2 a[4], b[4];
        if (first && i == j) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            a[t] = pa[t];
            b[t] = pb[t];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            a[t] = Q[(size_t)(i - 3 + t) * ldq + e0];
            b[t] = two ? Q[(size_t)(i - 3 + t) * ldq + e1] : make_double2(0.0, 0.0);
          }
        }
#pragma unroll
        for (int t = 3; t >= 0; --t) {
          cfma(r0, h[i - 3 + t], a[t]);
          cfma(r1, h[i - 3 + t], b[t]);
        }
      }
      for (; i >= 0; --i) {
        const double2 a = Q[(size_t)i * ldq + e0];
        const double2 b = two ? Q[(size_t)i * ldq + e1] : make_double2(0.0, 0.0);
        cfma(r0, h[i], a);
        cfma(r1, h[i], b);
      }
      w[e0] = r0;
      nrm += r0.x * r0.x + r0.y * r0.y;
      if (two) {
        w[e1] = r1;
        nrm += r1.x * r1.x + r1.y * r1.y;
      }
    }
  } else {
  for (int64_t e0 = ef0; e0 < n; e0 += EPL * stride) {
    const int64_t e1 = e0 + stride;
    const bool two = EPL == 2 && e1 < n;
    const bool first = e0 == ef0;   // (the same for every lane of the workgroup)
    double2 r0 = first ? pr0 : w[e0];
    double2 r1 = first ? pr1 : (two ? w[e1] : make_double2(0.0, 0.0));
    int i = 0;
    for (; i + 3 <= j; i += 4) {
      double2 a[4], b[4];
      if (first && i == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[t] = pa[t];
          b[t] = pb[t];
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[t] = Q[(size_t)(i + t) * ldq + e0];
          b[t] = two ? Q[(size_t)(i + t) * ldq + e1] : make_double2(0.0, 0.0);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        cfma(r0, h[i + t], a[t]);
        cfma(r1, h[i + t], b[t]);
      }
    }
    for (; i <= j; ++i) {
      const double2 a = Q[(size_t)i * ldq + e0];
      const double2 b = two ? Q[(size_t)i * ldq + e1] : make_double2(0.0, 0.0);
      cfma(r0, h[i], a);
      cfma(r1, h[i], b);
    }
    w[e0] = r0;
    nrm += r0.x * r0.x + r0.y * r0.y;
    if (two) {
      w[e1] = r1;
      nrm += r1.x * r1.x + r1.y * r1.y;
    }
  }
  }
  // block sum over BS / 64 wavefronts in wave order
  double v = wave_sum(nrm);
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[wv] = make_double2(v, 0.0);
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = lds[0].x;
    for (int k = 1; k < BS / 64; ++k) t += lds[k].x;
    norm_partials[blockIdx.x] = make_double2(t, 0.0);
  }
}

static int launch_multidot(hipStream_t s, const double2* Q, int64_t ldq, int j, const double2* w, double2* md_partials,
                           int64_t n, Stats* st) {
  const int ntiles = (j + 1 + kTI - 1) / kTI;
  hipLaunchKernelGGL((multidot_kernel<kThreads, 2>), dim3(kRedBlocks, ntiles), dim3(kThreads), 0, s, Q, ldq, j, w, md_partials, n);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

int launch_mgs_multidot(hipStream_t s, const double2* Q, int64_t ldq, int j, const double2* w, double2* md_partials,
                        double2* reduced, int64_t n, Stats* st) {
  int rc = launch_multidot(s, Q, ldq, j, w, md_partials, n, st);
  if (rc != QP_OK) return rc;
  hipLaunchKernelGGL(multidot_reduce_kernel, dim3(2 * (j + 1)), dim3(kThreads), 0, s, md_partials, j, reduced,
                     (unsigned*)nullptr, (double2*)nullptr, 0, (double2*)nullptr, (double2*)nullptr, 0.0);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

static int launch_mgs_update(hipStream_t s, const double2* Q, int64_t ldq, int j, double2* w, const double2* coef,
                             double2* norm_partials, int64_t n, Stats* st) {
  const size_t shmem = sizeof(double2) * (size_t)(j + 1 + kThreads / 64);
  hipLaunchKernelGGL((mgs_update_kernel<kThreads, 2, false, false>), dim3(kRedBlocks), dim3(kThreads), shmem, s, w, Q, ldq, j, coef,
                     norm_partials, n, MgsSolveArgs{});
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

int launch_mgs_project(hipStream_t s, const double2* Q, int64_t ldq, int j, double2* w, const double2* reduced,
                       double2* G, int ldg, double2* hess_col, double2* coef, double2* norm_partials, double dt,
                       int64_t n, Stats* st) {
  if (!mgs_lowsync_fits(j)) return fail(QP_E_BAD_ARG, "Krylov basis of %d vectors is too long for the low-synchronisation projection", j + 1);
  hipLaunchKernelGGL(mgs_solve_kernel, dim3(1), dim3(kThreads), mgs_solve_lds(j), s, j, reduced, G, ldg, hess_col, coef, dt);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return launch_mgs_update(s, Q, ldq, j, w, coef, norm_partials, n, st);
}

int launch_mgs_lowsync(hipStream_t s, const double2* Q, int64_t ldq, int j, double2* w, double2* md_partials,
                       double2* G, int ldg, double2* hess_col, double2* reduced, double2* coef, unsigned* ticket,
                       double2* norm_partials, double dt, int64_t n, Stats* st, bool solve_in_update,
                       unsigned* early_flag, unsigned flag_value, bool* early_armed, bool dots_done, bool l2_order) {
  if (!dots_done) {   // (else: the mat-vec left the partials, kernels_arnoldi.hip)
    int rc = launch_multidot(s, Q, ldq, j, w, md_partials, n, st);
    if (rc != QP_OK) return rc;
  }
  if (early_armed) *early_armed = false;
  if (solve_in_update && mgs_solve_lds(j) <= 12 * 1024) {   // j <= 35: reduction + solve in the projection's prologue
    if (early_armed) *early_armed = early_flag != nullptr;
    const size_t shmem = sizeof(double2) * (size_t)(j + 1 + kThreads / 64 + j + 1) + mgs_solve_lds(j);
    if (l2_order)
      hipLaunchKernelGGL((mgs_update_kernel<kThreads, 2, true, true>), dim3(kRedBlocks), dim3(kThreads), shmem, s, w, Q, ldq, j, coef,
                         norm_partials, n, MgsSolveArgs{md_partials, G, ldg, hess_col, dt, early_flag, flag_value});
    else
      hipLaunchKernelGGL((mgs_update_kernel<kThreads, 2, true, false>), dim3(kRedBlocks), dim3(kThreads), shmem, s, w, Q, ldq, j, coef,
                         norm_partials, n, MgsSolveArgs{md_partials, G, ldg, hess_col, dt, early_flag, flag_value});
    QP_HIP(hipGetLastError());
    if (st) st->n_launch++;
    return QP_OK;
  }
  hipLaunchKernelGGL(multidot_reduce_kernel, dim3(2 * (j + 1)), dim3(kThreads), mgs_solve_lds(j), s, md_partials, j,
                     reduced, ticket, G, ldg, hess_col, coef, dt);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return launch_mgs_update(s, Q, ldq, j, w, coef, norm_partials, n, st);
}

__global__ __launch_bounds__(kThreads) void reduce_triples_kernel(const double* __restrict__ partials, int nwg,
                                                                  double* __restrict__ out3) {
  __shared__ double2 lds[kThreads / 64];
  double a = 0, b = 0, c = 0;
  for (int i = threadIdx.x; i < nwg; i += kThreads) {
    a += partials[3 * (size_t)i + 0];
    b += partials[3 * (size_t)i + 1];
    c += partials[3 * (size_t)i + 2];
  }
  const double2 ab = block_sum(make_double2(a, b), lds);
  const double2 cc = block_sum(make_double2(c, 0.0), lds);
  if (threadIdx.x == 0) {
    out3[0] = ab.x;
    out3[1] = ab.y;
    out3[2] = cc.x;
  }
}

int launch_reduce_triples(hipStream_t s, const double* partials, int nwg, double* out3, Stats* st) {
  hipLaunchKernelGGL(reduce_triples_kernel, dim3(1), dim3(kThreads), 0, s, partials, nwg, out3);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

}  // namespace qp

// Batched states (BASELINE configs[4]): the panel kernels of the fused Chebyshev term and the CSR-ordered mirror's gather
// (split out of kernels.hip in round 4; kernel notes below and in DESIGN 4).
#include <atomic>
#include <cstring>
#include <type_traits>

#include "kernel_common.h"

namespace qp {

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
// WS wavefronts (consecutive walk positions) per workgroup: 8 measured 282 us per term of config C5
// against 302 with 4 and 290 with 16; 2-D tiles of walk positions per workgroup instead of runs: no difference
// (profiles/r02/batched_c5_sweep.txt)
// (the row's work as a device function: the tile kernel below sends the rows outside its tiles through the same code)
template <class Op>
__device__ __forceinline__ void spmm_row_scalar_entries(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                        const double2* __restrict__ vals, const double2* __restrict__ X, int b,
                                                        const Op& op, int64_t row, int lane) {
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
  spmm_row_scalar_entries(rowptr, cols, vals, X, b, op, row, lane);
}

// ---------------------------------------------------------------------------
// LDS-staged 4 x 4 tiles of a lattice operator's interior rows (SpmmTiles, device.h; plan: operator_spmm_tiles in engine_core.hip).
// Sixteen wavefronts = the sixteen rows r0 + i g + j of the tile, lane = state.  The workgroup first loads the 16 + 8 K + 8 NN
// distinct panel rows its rows read -- slots [(4 + 2 K) strip steps][4 columns], then [4 strip steps][2 NN near-halo columns] --
// five per wavefront, issued before the row-local streams so that the barrier waits on them as little as possible; after the
// barrier every wavefront sums its row exactly as spmm_rows_smem_kernel does (same entries, same order, same two partial sums:
// the same bits), with ds_read_b128 in place of the gathers.  Measured on the bare loop (profiles/r06/panel_tile_probe.txt):
// larger tiles or tiles held in registers lose to the occupancy they cost; two workgroups of 80 KiB per compute unit do not.
// ---------------------------------------------------------------------------
template <class Op>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmm_tile_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                         const double2* __restrict__ vals, const double2* __restrict__ X,
                                                         int64_t nrows, int b, Op op, const int32_t* __restrict__ tiles,
                                                         const int32_t* __restrict__ tab, int nd, int T, int ntiles,
                                                         const int32_t* __restrict__ rest, int nrest) {
  extern __shared__ double2 tile_lds[];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // the rows outside the tiles, sixteen per workgroup, by the row kernel's code: those workgroups are spread evenly over the launch
  // (every `stride`-th position of the walk; at its end they would all run on the last XCD, after its tiles)
  const unsigned nrwg = (unsigned)(nrest + 15) >> 4;
  unsigned tile = wg;
  if (nrwg > 0) {
    const unsigned stride = gridDim.x / nrwg;
    const unsigned before = min((wg + 1) / stride, nrwg);      // rest workgroups at positions <= wg
    if (wg % stride == stride - 1 && wg / stride < nrwg) {
      const int pos = (int)(wg / stride) * 16 + wave;
      if (pos < nrest) spmm_row_scalar_entries(rowptr, cols, vals, X, b, op, (int64_t)__builtin_amdgcn_readfirstlane(rest[pos]), lane);
      return;
    }
    tile = wg - before;
  }
  const int64_t r0 = (int64_t)__builtin_amdgcn_readfirstlane(tiles[tile]);
  const int st = blockIdx.y * 64 + lane;
  const bool active = st < b;
  const int stc = active ? st : b - 1;
  const double2* __restrict__ Xs = X + stc;
  // everything the row needs from memory is requested before the barrier: the wavefront's share of the staged rows (tab[slot] = that
  // slot's row relative to r0), the row's entries one per lane (broadcast later with v_readlane: after the barrier no load is left
  // whose latency only two resident workgroups per compute unit would have to hide), the LDS byte offset of every entry's operand
  // (tab[kSpmmTileSlots + 24 wave + k]: host arithmetic, SpmmTiles), the row-local streams
  double2 stg[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int slot = min(wave + 16 * q, T - 1);      // wave-uniform; past the last slot the load is repeated and not stored
    int64_t gr = r0 + tab[slot];
    gr = gr < 0 ? 0 : (gr >= nrows ? nrows - 1 : gr);   // (a slot no row of the tile reads: a pattern with gaps at the lattice's edge)
    stg[q] = Xs[gr * b];
  }
  const int64_t row = r0 + tab[kSpmmTileSlots + kSpmmTileMaxEntries * 16 + wave];
  const int64_t e = row * (int64_t)b + stc;
  const int lk = lane < nd ? lane : nd - 1;
  const double2 mv = vals[rowptr[row] + lk];
  const int mo = tab[kSpmmTileSlots + kSpmmTileMaxEntries * wave + lk];
  const int own = __builtin_amdgcn_readfirstlane(tab[kSpmmTileSlots + kSpmmTileMaxEntries * 16 + 16 + wave]);
  typename Op::Pre pre = op.pre_streams(e);
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int slot = wave + 16 * q;
    if (slot < T) tile_lds[slot * 64 + lane] = stg[q];
  }
  __syncthreads();
  const char* __restrict__ L = (const char*)(tile_lds + lane);
  pre.xi = *(const double2*)(L + own);
  const int len = nd;
  auto xk = [&](int k) -> double2 { return *(const double2*)(L + __builtin_amdgcn_readlane(mo, k)); };      // k: compile-time
  auto ak = [&](int k) -> double2 { return make_double2(readlane_f64(mv.x, k), readlane_f64(mv.y, k)); };
  double2 acc0 = make_double2(0.0, 0.0), acc1 = make_double2(0.0, 0.0);
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  // sums in the order of spmm_rows_smem_kernel / csr_spmm_kernel: groups of eight alternating between the two partial sums while
  // eight entries are left, then one group of four, then the last one to three entries into the first sum
#pragma unroll
  for (int kk = 0; kk < kSpmmTileMaxEntries; kk += 8) {
    if (kk + 7 < len) {      // (operands four at a time: the order of the sums is that of a group of eight, the registers are half)
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        double2 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = xk(kk + h + u);
        cfma(acc0, ak(kk + h), x[0]);
        cfma(acc1, ak(kk + h + 1), x[1]);
        cfma(acc0, ak(kk + h + 2), x[2]);
        cfma(acc1, ak(kk + h + 3), x[3]);
      }
    } else if (kk < len) {
      if (kk + 3 < len) {
        double2 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = xk(kk + u);
        cfma(acc0, ak(kk), x[0]);
        cfma(acc1, ak(kk + 1), x[1]);
        cfma(acc0, ak(kk + 2), x[2]);
        cfma(acc1, ak(kk + 3), x[3]);
#pragma unroll
        for (int u = 4; u < 7; ++u)
          if (kk + u < len) cfma(acc0, ak(kk + u), xk(kk + u));
      } else {
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (kk + u < len) cfma(acc0, ak(kk + u), xk(kk + u));
      }
    }
  }
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

static bool spmm_streams_nt(const Tuning& tun, int64_t nrows, int b) {
  return tun.spmm_nt == 2 || (tun.spmm_nt == 1 && (double)nrows * b * sizeof(double2) >= 128.0 * 1024 * 1024);
}

template <class Op>
static void launch_rows_smem(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals, const double2* X,
                             int64_t nrows, int b, const Op& op, const int32_t* order) {
  constexpr int WS = 8;
  dim3 grid((unsigned)((nrows + WS - 1) / WS), (unsigned)((b + 63) / 64));
  hipLaunchKernelGGL((spmm_rows_smem_kernel<Op, WS>), grid, dim3(64 * WS), 0, s, rowptr, cols, vals, X, nrows, b, op, order);
}

template <class Op>
static int launch_tiles_t(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals, const double2* X,
                          int64_t nrows, int b, const Op& op, const SpmmTiles& P) {
  const size_t lds = (size_t)P.T * 64 * sizeof(double2);
  // more than the 64 KiB a launch may ask for by default: a property of the function on a device, set once per device (the call
  // is a slow one: hundreds of microseconds on the host)
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  QP_HIP(hipGetDevice(&dev));
  if (dev >= 64 || !(done.load(std::memory_order_acquire) >> dev & 1)) {
    QP_HIP(hipFuncSetAttribute((const void*)spmm_tile_kernel<Op>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    if (dev < 64) done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  dim3 grid((unsigned)(P.ntiles + (P.nrest + 15) / 16), (unsigned)((b + 63) / 64));
  hipLaunchKernelGGL((spmm_tile_kernel<Op>), grid, dim3(1024), lds, s, rowptr, cols, vals, X, nrows, b, op, P.tiles, P.tab, P.shape.nd,
                     P.T, (int)P.ntiles, P.rest, (int)P.nrest);
  return QP_OK;
}

int launch_spmm_tile_cheby(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals, const double2* X,
                           int64_t nrows, int64_t nnz, int b, const ChebyEpi& e, const Tuning& tun, const SpmmTiles& P, Stats* st) {
  if (nrows == 0) return QP_OK;
  if (spmm_streams_nt(tun, nrows, b)) {
    ChebyOpT<true> op{e};
    if (const int rc = launch_tiles_t(s, rowptr, cols, vals, X, nrows, b, op, P)) return rc;
  } else {
    ChebyOp op{e};
    if (const int rc = launch_tiles_t(s, rowptr, cols, vals, X, nrows, b, op, P)) return rc;
  }
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
    st->spmv_bytes += 20.0 * (double)nnz + 4.0 * (double)(nrows + 1) + 80.0 * (double)nrows * b;
  }
  return QP_OK;
}

int launch_spmm_cheby(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals,
                      const double2* X, int64_t nrows, int64_t nnz, int b, const ChebyEpi& e, const Tuning& tun,
                      bool rows_kernel, const int32_t* order, Stats* st) {
  if (nrows == 0) return QP_OK;
  if (rows_kernel) {
    const bool nt = spmm_streams_nt(tun, nrows, b);
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
    if (tun.spmm_rw <= 0 && nnz <= (int64_t)INT32_MAX) {   // (the scalar-entry kernel broadcasts a 32-bit row pointer)
      if (nt) {
        ChebyOpT<true> op{e};
        launch_rows_smem(s, rowptr, cols, vals, X, nrows, b, op, order);
      } else {
        ChebyOp op{e};
        launch_rows_smem(s, rowptr, cols, vals, X, nrows, b, op, order);
      }
    } else
    switch (tun.spmm_rw) {
      case 2: QP_SPMM_ROWS(2) break;
      case 4: QP_SPMM_ROWS(4) break;
      case 8: QP_SPMM_ROWS(8) break;
      default: QP_SPMM_ROWS(1) break;
    }
#undef QP_SPMM_ROWS
  } else
  // states per pass of the tiled kernel: 16 (the gather window of a banded H stays inside an XCD's L2; 32 and 64 measured slower,
  // profiles/r02/batched_c5_sweep.txt); a panel of at most eight states (one GPU's share of 64 over 8): 8, no idle lanes
  if (b <= 8) launch_spmm_cheby_t<8>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt);
  else launch_spmm_cheby_t<16>(s, rowptr, cols, vals, X, nrows, b, e, tun.spmm_nt);
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

}  // namespace qp

// Row-block mat-vec through the value-dictionary mirror (device.h: CodedVals) -- gfx950, wave64.
//
// The operator's values are a few distinct numbers (a spin chain's couplings and its handful of diagonal energies, a grid's
// hopping amplitudes; the reference's generators are sums of such terms, src/generators.jl:634-645): every 64-row block carries
// one byte per stored entry and a table of the distinct values it holds.  One wavefront per row block, lane = row, exactly as
// rbcsr_spmv_kernel (kernels.hip): per quad of slots one 4-byte load of the four codes (256 B per wavefront instead of 4 KiB of
// complex values), the column section as stored, then four table reads -- from the block's table of at most 256 entries, which the
// wavefront copies into its LDS window first -- and the four gathers.  The partial sums
// alternate between two accumulators in slot order as in rbcsr_spmv_kernel and the table holds the numbers the value plane
// would hold, so the result is bit-identical to the uncoded path.
#include <type_traits>

#include "kernel_common.h"

namespace qp {

template <class Op, class TT, int WS>   // TT: double2, or double for an all-real operator; WS row blocks per workgroup
__global__ __launch_bounds__(64 * WS) void rbcsr_coded_spmv_kernel(const int64_t* __restrict__ bptr,
                                                                    const int64_t* __restrict__ cmeta,
                                                                    const char* __restrict__ colbytes,
                                                                    const unsigned* __restrict__ codes4,
                                                                    const int64_t* __restrict__ tptr,
                                                                    const TT* __restrict__ tab,
                                                                    const double2* __restrict__ x, int64_t nblocks,
                                                                    int64_t nrows, Op op,
                                                                    const int32_t* __restrict__ block_map, SyncArgs sy) {
  static_assert(WS == kThreads / 64 || std::is_same<Op, ChebyOp>::value,
                "block_sum (Op::begin of the folded norm, finish_check) sums kThreads / 64 wavefronts");
  __shared__ double2 lds[WS];
  __shared__ TT tabs[WS][256];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  sync_wait(sy, wg);
  op.begin_issue();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)wg * WS + wave;   // position in the row set
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  int64_t row = nrows;
  typename Op::Pre pre;
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
  if (idx < nblocks) {
    const int64_t b = block_map ? (int64_t)block_map[idx] : idx;
    const int64_t base = bptr[b];
    const int nq = (int)((bptr[b + 1] - base) >> 8);   // slots / 4
    const unsigned* __restrict__ cq = codes4 + (base >> 2) + lane;
    const int64_t cm = cmeta[b];
    row = b * kRB + lane;
    const int64_t rowc = row < nrows ? row : nrows - 1;
    pre = op.pre(rowc);
    // the block's table into this wavefront's LDS window (at most 256 entries; a spin chain's block has a few dozen): the
    // look-ups then go through the LDS crossbar, not through the vector L1 that the gathers of x keep busy
    // (profiles/r05/value_dictionary.txt: 21 table reads per row through the L1 cost 8.5 of 45 us per term)
    TT* __restrict__ tw = tabs[wave];
    {
      const int64_t tp = tptr[b];
      const TT* __restrict__ tb = tab + (tp >> 9);
      const int tlen = (int)(tp & 511);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i * 64 < tlen) tw[i * 64 + lane] = tb[min(i * 64 + lane, tlen - 1)];   // (wave-uniform condition)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll 4
    for (int q = 0; q < nq; ++q) {
      const unsigned cw = __builtin_nontemporal_load(cq + (size_t)q * 64);
      const int4 c = ld_cols<true>(colbytes, cm, q, lane, (int)rowc);
      const double2 x0 = x[c.x];
      const double2 x1 = x[c.y];
      const double2 x2 = x[c.z];
      const double2 x3 = x[c.w];
      const double2 a0 = ld_val<false>(tw + (cw & 255u));
      const double2 a1 = ld_val<false>(tw + ((cw >> 8) & 255u));
      const double2 a2 = ld_val<false>(tw + ((cw >> 16) & 255u));
      const double2 a3 = ld_val<false>(tw + (cw >> 24));
      cfma(s0, a0, x0);
      cfma(s1, a1, x1);
      cfma(s0, a2, x2);
      cfma(s1, a3, x3);
    }
  }
  op.begin(lds);
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, idx * kRB + lane);
  finish_check(op, chk, nrm, lds);
  sync_signal(sy);
}

template <class Op, int WS>
static int launch_coded(hipStream_t s, const DevMatrix& A, const double2* x, const Op& op, int64_t nblk, const int32_t* bmap,
                        const SyncArgs& sy) {
  const CodedVals& C = *A.cv;
  const dim3 grid((unsigned)((nblk + WS - 1) / WS));
  const unsigned* codes4 = reinterpret_cast<const unsigned*>(C.codes);
  if (C.use_real)
    hipLaunchKernelGGL((rbcsr_coded_spmv_kernel<Op, double, WS>), grid, dim3(64 * WS), 0, s, A.bptr, A.cmeta,
                       reinterpret_cast<const char*>(A.cols), codes4, C.tptr, C.tab_r, x, nblk, A.nrows, op, bmap, sy);
  else
    hipLaunchKernelGGL((rbcsr_coded_spmv_kernel<Op, double2, WS>), grid, dim3(64 * WS), 0, s, A.bptr, A.cmeta,
                       reinterpret_cast<const char*>(A.cols), codes4, C.tptr, C.tab, x, nblk, A.nrows, op, bmap, sy);
  QP_HIP(hipGetLastError());
  return QP_OK;
}

// wide: eight row blocks per workgroup (the plain fused term of a whole operator or of an interior launch: no per-workgroup
// check partials, no completion signal, no mirror map -- launch_spmv decides); the wait threshold is given in workgroups of four
int launch_rbcsr_coded_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, int64_t nblk,
                             const int32_t* bmap, const SyncArgs& sy, bool wide) {
  ChebyOp op{e};
  if (wide) {
    SyncArgs sy8 = sy;
    sy8.wait_from_wg = sy.wait_from_wg / 2;
    return launch_coded<ChebyOp, 8>(s, A, x, op, nblk, bmap, sy8);
  }
  return launch_coded<ChebyOp, kThreads / 64>(s, A, x, op, nblk, bmap, sy);
}

int launch_rbcsr_coded_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, int64_t nblk,
                             const int32_t* bmap, const SyncArgs& sy) {
  PlainOp op{e};
  return launch_coded<PlainOp, kThreads / 64>(s, A, x, op, nblk, bmap, sy);
}

}  // namespace qp

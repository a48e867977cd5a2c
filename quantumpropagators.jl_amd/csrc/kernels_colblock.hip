// Column-blocked mat-vec for operators with IRREGULAR columns (device.h: ColBlockPlan).
//
// The reference places no structure on a generator's terms (src/generators.jl:634-645: any sparse H_l; src/cheby.jl:177,
// :191 and src/arnoldi.jl:81 are `mul!(v, H, Psi)` with whatever H is).  Lattices and bands take the row-block formats
// and the strip walk; an operator whose columns are spread over the whole vector (a random graph, a molecular
// Hamiltonian in a configuration basis) defeats them: every gathered x[col] is its own 128-byte line, the vector
// (16 MB at N = 2^20) does not fit the 4 MB of L2 an XCD owns, and the term runs at the rate the Infinity Cache serves
// single lines -- 233 us per term for 16 random columns per row at N = 2^20 (profiles/r03), 0.21 of the HBM roofline.
// Measured with the mirror (profiles/r04/colblock.txt, colblock_pmc.txt): 174 us at 2^20, 354 (569) at 2^21, 835 (1265) at
// 2^22; the L2 hit rate goes from 0.30 to 0.76-0.89 and the mean request latency from 794 to 440 cycles.  The floor is the
// L2's request rate: a 16-byte gather costs a whole line request, no CU can reuse a line, and ~20 M requests per term pass
// at ~7 per clock and XCD.
//
// Here the SAME entries are grouped by (row tile, column block) and the launch walks the column blocks in its OUTER loop:
//   * every wavefront of the launch is resident from the first cycle (the grid is sized to the chip) and owns its row tiles
//     -- 64 rpt rows each, `tpw` of them -- for the whole launch, row sums in registers;
//   * phase c: every wavefront processes the segments (own tile, block c).  All gathers of the whole chip fall into ONE
//     window of 2^log2w elements of x (1-4 MB: about sixteen blocks, engine_plans.hip build_colblock): every XCD's L2 loads
//     the window once and serves the rest as hits;
//   * inside a segment the entries are read one per lane (coalesced, nontemporal: the matrix is read exactly once per
//     launch), multiplied with the gathered element, and the products staged in the wavefront's own LDS buffer; then lane r
//     adds up the products of its rows in storage order (ascending column, as the CSR kernels do).  No workgroup barrier:
//     the buffer is private to the wavefront and the LDS executes a wavefront's accesses in order.
// Algorithmic bytes per term: the contract's 20 z N + 84 N; the mirror moves 20 z N + 2 P N of row offsets +
// 8 x 16 N of vector windows (one per XCD) + the row-local streams.
#include "kernel_common.h"

namespace qp {

template <class Op, class VT, int RPT, int MAXT>
__global__ __launch_bounds__(kThreads) void colblock_spmv_kernel(const int32_t* __restrict__ segptr,
                                                                 const uint16_t* __restrict__ rowoff,
                                                                 const uint32_t* __restrict__ cols, const VT* __restrict__ vals,
                                                                 const double2* __restrict__ x, int64_t nrows, int64_t ntiles,
                                                                 int P, int tpw, int cap, Op op) {
  extern __shared__ double2 cb_lds[];   // [kThreads / 64][cap] products, then kThreads / 64 slots for the workgroup sums
  constexpr int TR = 64 * RPT;
  op.begin_issue();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t nW = (int64_t)gridDim.x * (kThreads / 64);
  const int64_t w = (int64_t)blockIdx.x * (kThreads / 64) + wave;
  double2* __restrict__ my = cb_lds + (size_t)wave * cap;
  double2 acc[MAXT][RPT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int q = 0; q < RPT; ++q) acc[i][q] = make_double2(0.0, 0.0);

  for (int c = 0; c < P; ++c) {
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int64_t t = w + (int64_t)i * nW;
      if (i < tpw && t < ntiles) {   // (wave-uniform)
        const int64_t sg = t * P + c;
        const int e0 = __builtin_amdgcn_readfirstlane(segptr[sg]);
        const int e1 = __builtin_amdgcn_readfirstlane(segptr[sg + 1]);
        // where this lane's rows begin and end inside the segment (issued first: needed last)
        const uint16_t* __restrict__ ro = rowoff + (size_t)sg * (TR + 1) + lane;
        int o0[RPT], o1[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          o0[q] = ro[q * 64];
          o1[q] = ro[q * 64 + 1];
        }
        // one entry per lane, four in flight: value and column (streams), gathered element (the L2-resident window)
        for (int e = e0 + lane; e < e1; e += 256) {
          double2 v[4];
          uint32_t cc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ee = e + 64 * u;
            const int ec = ee < e1 ? ee : e1 - 1;
            v[u] = ld_val<true>(vals + ec);
            cc[u] = __builtin_nontemporal_load(cols + ec);
          }
          double2 xg[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) xg[u] = x[cc[u]];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ee = e + 64 * u;
            if (ee < e1) my[ee - e0] = cmul(v[u], xg[u]);
          }
        }
        __builtin_amdgcn_wave_barrier();   // (ordering for the compiler; the LDS serves one wavefront's accesses in order)
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          double2 s = acc[i][q];
          for (int j = o0[q]; j < o1[q]; ++j) {
            const double2 p = my[j];
            s.x += p.x;
            s.y += p.y;
          }
          acc[i][q] = s;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }

  double2* lds4 = cb_lds + (size_t)(kThreads / 64) * cap;
  op.begin(lds4);
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int64_t t = w + (int64_t)i * nW;
    if (i < tpw && t < ntiles) {
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int64_t row = t * TR + q * 64 + lane;
        if (row < nrows) op.row(row, acc[i][q], op.pre(row), chk, nrm, row);
      }
    }
  }
}

// (Tried and dropped, round 4: the same walk with the next segment's values, columns and row offsets requested before the
// current segment is worked on, segment pointers in registers -- 177 vs 175 us per term at N = 2^20, 410 vs 354 at 2^21.  The
// dependent chain of a segment is not what bounds the kernel; the L2's request rate is: DESIGN 4.)

// mirror values from the operator's current values
__global__ __launch_bounds__(kThreads) void colblock_gather_kernel(double2* __restrict__ out, double* __restrict__ out_r,
                                                                   const double2* __restrict__ vals,
                                                                   const int64_t* __restrict__ map, int64_t nnz) {
  for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * kThreads) {
    const int64_t m = map[p];
    double2 v = vals[m >= 0 ? m : -m - 1];
    if (m < 0) v.y = -v.y;
    out[p] = v;
    if (out_r) out_r[p] = v.x;
  }
}

int launch_colblock_gather(hipStream_t s, const ColBlockPlan& P, const double2* src, Stats* st) {
  if (!P.valid || P.nnz == 0) return QP_OK;
  int64_t g = (P.nnz + kThreads - 1) / kThreads;
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL(colblock_gather_kernel, dim3((unsigned)g), dim3(kThreads), 0, s, P.vals, P.use_real ? P.vals_r : nullptr, src,
                     P.map, P.nnz);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

template <class Op, class VT, int RPT, int MAXT>
static int launch_cb_instance(hipStream_t s, const ColBlockPlan& P, const VT* vals, const double2* x, int64_t nrows, int grid, int tpw,
                              int cap, size_t lds, const Op& op) {
  auto kern = colblock_spmv_kernel<Op, VT, RPT, MAXT>;
  if (lds > 48 * 1024)
    QP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, P.segptr, P.rowoff, P.cols, vals, x, nrows, P.ntiles, P.P, tpw, cap, op);
  QP_HIP(hipGetLastError());
  return QP_OK;
}

template <class Op>
static int launch_colblock(hipStream_t s, const DevMatrix& A, const double2* x, const Op& op, const Tuning& tun, bool* launched) {
  *launched = false;
  const ColBlockPlan* Pp = A.cb;
  if (!Pp || !Pp->valid || tun.colblock == 0 || A.nrows == 0) return QP_OK;
  const ColBlockPlan& P = *Pp;
  // every wavefront resident from the start: the grid is what the chip holds, a wavefront takes `tpw` tiles
  const int64_t resident = (int64_t)std::max(device_cu_count(), 1) * kCbWavesPerCu;
  int64_t tpw = (P.ntiles + resident - 1) / resident;
  if (tpw < 1) tpw = 1;
  if (tpw > kCbMaxTilesPerWave) return QP_OK;
  const int64_t nW = (P.ntiles + tpw - 1) / tpw;
  const int grid = (int)((nW + kThreads / 64 - 1) / (kThreads / 64));
  const int cap = (P.max_seg + 31) & ~31;
  const size_t lds = sizeof(double2) * ((size_t)(kThreads / 64) * (size_t)cap + (size_t)(kThreads / 64));
  const bool real = P.use_real && A.vals_r != nullptr;
#define QP_CB_LAUNCH(RPT, MAXT)                                                                                                      \
  (real ? launch_cb_instance<Op, double, RPT, MAXT>(s, P, P.vals_r, x, A.nrows, grid, (int)tpw, cap, lds, op)                        \
        : launch_cb_instance<Op, double2, RPT, MAXT>(s, P, P.vals, x, A.nrows, grid, (int)tpw, cap, lds, op))
  int rc;
  if (P.rpt == 2) rc = (tpw <= 2) ? QP_CB_LAUNCH(2, 2) : QP_CB_LAUNCH(2, kCbMaxTilesPerWave);
  else if (P.rpt == 1) rc = (tpw <= 2) ? QP_CB_LAUNCH(1, 2) : QP_CB_LAUNCH(1, kCbMaxTilesPerWave);
  else return QP_OK;
#undef QP_CB_LAUNCH
  if (rc != QP_OK) return rc;
  *launched = true;
  return QP_OK;
}

int launch_colblock_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, const Tuning& tun, bool* launched) {
  *launched = false;
  if (e.check_partials || e.mirror) return QP_OK;   // (per-workgroup check triples / the exchange pack count row blocks)
  ChebyOp op{e};
  return launch_colblock(s, A, x, op, tun, launched);
}

int launch_colblock_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const Tuning& tun, bool* launched) {
  PlainOp op{e};
  return launch_colblock(s, A, x, op, tun, launched);
}

}  // namespace qp

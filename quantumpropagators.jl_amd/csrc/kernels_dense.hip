// QP_FMT_DENSE: the generator is a dense matrix (the reference's own test and benchmark operators:
// test/test_cheby.jl:24-47 and test/test_newton.jl:53-65 use N = 1000 `Hermitian(rand(ComplexF64, N, N))`; BASELINE
// configs[0] is N = 128 dense).  Stored row-major, 16 B per entry (8 B when every term and coefficient is real) and NO
// index bytes -- as plain CSR the same operator costs 20 B per entry plus the column decode in front of every gather.
//
//   one state    dense_gemv_kernel<Op>: one wavefront per row (R rows per wavefront for large operators, so that x is
//                loaded once per R rows), lanes stride the row in 1-KiB bursts, four independent accumulators, a DPP
//                wavefront sum, the row epilogue of the sparse kernels (fused Chebyshev term / plain mat-vec with the
//                folded Arnoldi normalisation).  HBM / L2 bound: 16 N^2 bytes per term.
//   b states     dense_zgemm_cheby_kernel: Y[N x b] = epilogue(H X) -- here H [psi_1 .. psi_b] IS a dense panel
//                contraction (the case BASELINE's north_star reserves the matrix cores for): arithmetic intensity b / 2
//                flop per byte, beyond the fp64 ridge (9.8) from b = 20.  v_mfma_f64_16x16x4_f64, a complex product as four
//                real MFMAs (two when H is real); the k loop is built like engine_liouville.hip's 32 x 32 kernel, whose
//                measured rule it follows: on this chip the fp64 MFMA shares the issue of the fp64 vector unit, so the loop
//                carries no vector-ALU work but two sign flips -- operands go from L2 into the MFMA lane layout through
//                buffer loads (scalar base advanced by the scalar unit + a constant lane offset), software-pipelined D
//                deep, and the recurrence + accumulate of the Chebyshev term run in the epilogue of the tile.
// gfx950 only.  Build with -mllvm -amdgpu-mfma-vgpr-form (accumulators stay in VGPRs across the k loop).
#include <type_traits>

#include "kernel_common.h"

namespace qp {

// ---------------------------------------------------------------------------
// one state: y_i = sum_k H[i, k] x_k, then Op's row epilogue
// ---------------------------------------------------------------------------
template <class Op, class VT, int R, bool NT>
__global__ __launch_bounds__(kThreads) void dense_gemv_kernel(const VT* __restrict__ H, const double2* __restrict__ x,
                                                              int64_t nrows, int64_t ncols, Op op) {
  __shared__ double2 lds[kThreads / 64];
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);   // every XCD a contiguous range of rows: its share of H stays in its L2
  op.begin_issue();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)wg * (kThreads / 64) + wave) * R;
  double2 s[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r][0] = s[r][1] = make_double2(0.0, 0.0);
  typename Op::Pre pre;
  if (lane < R && row0 + lane < nrows) pre = op.pre(row0 + lane);   // the row-local operands of the epilogue, ahead of the row sums
  if (row0 < nrows) {
    const VT* __restrict__ h[R];
#pragma unroll
    for (int r = 0; r < R; ++r) h[r] = H + (size_t)min(row0 + r, nrows - 1) * (size_t)ncols;   // (rows past the end repeat the last: not stored)
    int64_t k = lane;
    for (; k + 192 < ncols; k += 256) {   // four 1-KiB bursts per row in flight
      const double2 x0 = x[k], x1 = x[k + 64], x2 = x[k + 128], x3 = x[k + 192];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double2 a0 = ld_val<NT>(h[r] + k), a1 = ld_val<NT>(h[r] + k + 64), a2 = ld_val<NT>(h[r] + k + 128),
                      a3 = ld_val<NT>(h[r] + k + 192);
        cfma(s[r][0], a0, x0);
        cfma(s[r][1], a1, x1);
        cfma(s[r][0], a2, x2);
        cfma(s[r][1], a3, x3);
      }
    }
    for (; k < ncols; k += 64) {
      const double2 xk = x[k];
#pragma unroll
      for (int r = 0; r < R; ++r) cfma(s[r][0], ld_val<NT>(h[r] + k), xk);
    }
  }
  double2 tot = make_double2(0.0, 0.0);   // lane r keeps the sum of row row0 + r
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const double re = wave_sum(s[r][0].x + s[r][1].x), im = wave_sum(s[r][0].y + s[r][1].y);
    if (lane == r) tot = make_double2(re, im);
  }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  op.begin(lds);
  if (lane < R && row0 + lane < nrows) op.row(row0 + lane, tot, pre, chk, nrm, row0 + lane);
  finish_check(op, chk, nrm, lds);
}

template <class Op>
static int launch_dense_gemv(hipStream_t s, const DevMatrix& A, const double2* x, const Op& op, Stats* st) {
  if (A.nrows == 0) return QP_OK;
  // R rows per wavefront once there are rows to spare (x is then read once per R rows: the wave's x loads go through the
  // same L1 queue as its matrix loads); nontemporal matrix loads once the operator is beyond what the caches can hold
  // between two terms anyway
  const int R = dense_gemv_rows_per_wave(A.nrows);
  const bool nt = (double)A.nrows * (double)A.ncols * (A.vals_r ? 8.0 : 16.0) > 224.0 * 1024 * 1024;
  const dim3 grid((unsigned)dense_gemv_grid(A.nrows));
#define QP_DENSE_GEMV(RR, NTF)                                                                                         \
  do {                                                                                                                 \
    if (A.vals_r)                                                                                                      \
      hipLaunchKernelGGL((dense_gemv_kernel<Op, double, RR, NTF>), grid, dim3(kThreads), 0, s, A.vals_r, x, A.nrows, A.ncols, op); \
    else                                                                                                               \
      hipLaunchKernelGGL((dense_gemv_kernel<Op, double2, RR, NTF>), grid, dim3(kThreads), 0, s, A.vals, x, A.nrows, A.ncols, op);  \
  } while (0)
  if (R == 4) {
    if (nt) QP_DENSE_GEMV(4, true); else QP_DENSE_GEMV(4, false);
  } else {
    if (nt) QP_DENSE_GEMV(1, true); else QP_DENSE_GEMV(1, false);
  }
#undef QP_DENSE_GEMV
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
  }
  return QP_OK;
}

int launch_dense_gemv_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, Stats* st) {
  ChebyOp op{e};
  const int rc = launch_dense_gemv(s, A, x, op, st);
  if (st && rc == QP_OK) st->spmv_bytes += (A.vals_r ? 8.0 : 16.0) * (double)A.nrows * (double)A.ncols + 80.0 * (double)A.nrows;
  return rc;
}

int launch_dense_gemv_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, Stats* st) {
  PlainOp op{e};
  const int rc = launch_dense_gemv(s, A, x, op, st);
  if (st && rc == QP_OK) st->spmv_bytes += (A.vals_r ? 8.0 : 16.0) * (double)A.nrows * (double)A.ncols + 32.0 * (double)A.nrows;
  return rc;
}

// ---------------------------------------------------------------------------
// b states: the fused Chebyshev term of the panel on the fp64 matrix cores
// ---------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// 16 (or 8) bytes at (wave-uniform base) + (32-bit lane offset): a buffer load whose descriptor the scalar unit builds from
// the base -- no vector-ALU address arithmetic in the k loop
__device__ __forceinline__ double2 ld_off(const double2* base, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(base), (short)0, -1, 0x00020000);
  const u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  double2 d;
  __builtin_memcpy(&d, &v, 16);
  return d;
}
__device__ __forceinline__ double2 ld_off(const double* base, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), (short)0, -1, 0x00020000);
  const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
  double d;
  __builtin_memcpy(&d, &v, 8);
  return make_double2(d, 0.0);
}

// One workgroup = one (16 TA) x (16 TB) tile of the panel (rows i0 .., states s0 ..); each of its four wavefronts owns a quarter
// of the inner dimension and keeps the whole tile (TA x TB MFMA tiles, real and imaginary part: 8 TA TB accumulator
// registers); a k-step of 4 is TA + TB loads for 4 TA TB MFMAs (2 TA TB when H is real).  TA = TB = 2 (32 x 32) for panels of
// more than 16 states: 16 MFMAs per four 1-KiB loads, the MFMA rate decides (N = 4096, b = 64: 61 TFLOP/s); TA = TB = 1
// (16 x 16) for panels of at most 16 states, where the step is HBM-bound (arithmetic intensity b / 2 < 9.8 flop/B): twice the
// workgroups, no MFMA spent on columns past the panel's width.  MFMA operand layout (as in engine_liouville.hip, verified there
// against the library GEMM): lane l = (li = l & 15, lk = l >> 4) feeds A[i = li][k = lk] and B[k = lk][j = li] and receives
// C[i = lk + 4 r][j = li] in accumulator register r.
//   A fragment: H[i0 + 16 a + li][k + lk]            (row-major H: 16 rows x 64 contiguous bytes per load)
//   B fragment: X[(k + lk) b + s0 + 16 c + li]       (panel, state index contiguous: 4 rows x 256 contiguous bytes)
// The four partial tiles are summed through LDS in wave order (deterministic), then the 4 TA TB (tile, register) pairs are dealt
// to the four wavefronts, which apply the row epilogue of the fused term: element e = row * b + state, exactly as the sparse
// panel kernels do.
template <class Op, class VT, int D, int TA, int TB>
__global__ __launch_bounds__(256) void dense_zgemm_cheby_kernel(const VT* __restrict__ H, const double2* __restrict__ X, int n,
                                                                int ncols, int b, Op op) {
  constexpr bool CPLX = std::is_same<VT, double2>::value;
  constexpr int NT = TA * TB;
  __shared__ double red[4][NT][2][4][64];   // the partial tiles of the four wavefronts (16 KB per MFMA tile)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * (16 * TA), col0 = blockIdx.y * (16 * TB);
  const int ksteps = ncols >> 2;   // whole k-steps; the ncols & 3 inner indices left over: one masked step after the loop
  const int per = (ksteps + 3) / 4;
  const int sbeg = wave * per;
  const int total = max(min(ksteps, sbeg + per) - sbeg, 0);
  const int li = lane & 15, lk = lane >> 4;
  constexpr unsigned ES = sizeof(VT);
  unsigned oa[TA], ob[TB];
  int ra[TA], cb[TB];
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    ra[a] = min(row0 + 16 * a + li, n - 1) - row0;     // rows past the end repeat the last (not stored)
    oa[a] = ((unsigned)ra[a] * (unsigned)ncols + (unsigned)lk) * ES;
  }
#pragma unroll
  for (int c = 0; c < TB; ++c) {
    cb[c] = min(col0 + 16 * c + li, b - 1);
    ob[c] = ((unsigned)lk * (unsigned)b + (unsigned)cb[c]) * 16u;
  }
  const VT* baseA = H + (size_t)row0 * (size_t)ncols + (size_t)sbeg * 4;
  const double2* baseB = X + (size_t)sbeg * 4 * (size_t)b;
  const size_t strideB = (size_t)4 * (size_t)b;

  v4d cr[TA][TB], ci[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int c = 0; c < TB; ++c) cr[a][c] = ci[a][c] = v4d{0.0, 0.0, 0.0, 0.0};

  double2 fa[D][TA], fb[D][TB];
  auto load = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TA; ++a) fa[slot][a] = ld_off(baseA, oa[a]);
#pragma unroll
    for (int c = 0; c < TB; ++c) fb[slot][c] = ld_off(baseB, ob[c]);
    baseA += 4;
    baseB += strideB;
  };
  auto mfma = [&](int slot) {   // 4 TA TB MFMAs (half for a real H); consecutive ones never share an accumulator where TA TB > 1
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        cr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].x, fb[slot][c].x, cr[a][c], 0, 0, 0);
        ci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].x, fb[slot][c].y, ci[a][c], 0, 0, 0);
      }
    if (CPLX) {
      double nai[TA];
#pragma unroll
      for (int a = 0; a < TA; ++a) nai[a] = -fa[slot][a].y;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          cr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai[a], fb[slot][c].y, cr[a][c], 0, 0, 0);
          ci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].y, fb[slot][c].x, ci[a][c], 0, 0, 0);
        }
    }
  };
  constexpr int NM = (CPLX ? 4 : 2) * NT;          // MFMAs per k-step
  constexpr int NL = TA + TB;                      // refills per k-step
  int s = 0;
  if (total >= 2 * D - 1) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
      load(d);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; s + 2 * D - 1 <= total; s += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load((j + D - 1) % D);
        mfma(j);
#pragma unroll
        for (int g = 0; g < NM; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                      // at most one vector-ALU instruction
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);                      // scalar work of the cursors
          if (NM >= NL ? (g % (NM / NL) == (NM / NL) / 2) : true)
            __builtin_amdgcn_sched_group_barrier(0x020, NM >= NL ? 1 : NL / NM, 0);   // the step's refills, spread over it
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // here steps s .. s + D - 2 are loaded or in flight, in slots 0 .. D - 2
    for (; s < total; s += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if (s + j < total) {
          if (s + j + D - 1 < total) load((j + D - 1) % D);
          mfma(j);
        }
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < total) load(d);
    for (; s < total; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s + d < total) {
          mfma(d);
          if (s + d + D < total) load(d);
        }
      }
    }
  }
  // ncols not a multiple of 4: one more k-step for the ncols & 3 inner indices left over (wavefront 0), the lanes past the
  // end masked out of the A fragment
  if ((ncols & 3) && wave == 0) {
    const int k = ksteps * 4 + lk;
    const bool kin = k < ncols;
    const int kc = min(k, ncols - 1);
    const double2 zero = make_double2(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      const double2 p = ld_val<false>(H + ((size_t)(row0 + ra[a]) * (size_t)ncols + kc));
      fa[0][a] = kin ? p : zero;
    }
#pragma unroll
    for (int c = 0; c < TB; ++c) fb[0][c] = X[(size_t)kc * b + cb[c]];
    mfma(0);
  }
  // the 4 NT (tile, accumulator register) pairs of the workgroup's tile, NT per wavefront: pair p = wave NT + i is register
  // r = p & 3 of MFMA tile p >> 2 (for the 32 x 32 tile: wavefront w finishes MFMA tile w).  Their row-local operands are
  // requested before the partial tiles go through LDS.
  typename Op::Pre pre[NT];
  int64_t el[NT];
  bool live[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int p = wave * NT + i, tile = p >> 2, r = p & 3;
    const int row = row0 + (tile / TB) * 16 + lk + 4 * r, col = col0 + (tile % TB) * 16 + li;
    live[i] = row < n && col < b;
    el[i] = live[i] ? (int64_t)row * b + col : 0;
    if (live[i]) pre[i] = op.pre(el[i]);
  }
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int c = 0; c < TB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wave][a * TB + c][0][r][lane] = cr[a][c][r];
        red[wave][a * TB + c][1][r][lane] = ci[a][c][r];
      }
  __syncthreads();
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int p = wave * NT + i, tile = p >> 2, r = p & 3;
    double sr = red[0][tile][0][r][lane], si = red[0][tile][1][r][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      sr += red[w][tile][0][r][lane];
      si += red[w][tile][1][r][lane];
    }
    if (live[i]) op.row(el[i], make_double2(sr, si), pre[i], chk, nrm, el[i]);
  }
}

int launch_dense_zgemm_cheby(hipStream_t s, const DevMatrix& A, const double2* X, int batch, const ChebyEpi& e, Stats* st) {
  if (A.nrows == 0) return QP_OK;
  if (e.check_partials) return fail(QP_E_BAD_ARG, "check_normalization is not available for the batched step");
  if (A.nrows > (int64_t)INT32_MAX / 64 || A.ncols > (int64_t)INT32_MAX / 64 || (int64_t)32 * A.ncols * 16 > (int64_t)UINT32_MAX ||
      (int64_t)4 * batch * 16 > (int64_t)UINT32_MAX)
    return fail(QP_E_BAD_ARG, "dense panel step: operator %lld x %lld too large for the 32-bit lane offsets", (long long)A.nrows,
                (long long)A.ncols);
  // Tile per workgroup: 32 x 32 where that still gives every compute unit a workgroup (each of its wavefronts runs at the
  // MFMA issue rate alone: more of them per CU add nothing, fewer workgroups than CUs leave matrix cores idle); else 16 x 32,
  // else 16 x 16 -- also for panels of at most 16 states, which are HBM-bound and would spend half of a wider tile's MFMAs
  // on columns past the panel's width.  (N = 4096: b = 8 126 -> 46 us per term, b = 32 126 -> 16 x 32 tiles;
  // profiles/r04/dense_narrow_panels.txt)
  auto wgs = [&](int ta, int tb) { return ((A.nrows + 16 * ta - 1) / (16 * ta)) * (int64_t)((batch + 16 * tb - 1) / (16 * tb)); };
  int ta = 1, tb = 1;
  if (batch > 16) {
    if (wgs(2, 2) >= 192) ta = 2, tb = 2;
    else if (wgs(1, 2) >= 192) tb = 2;
  }
  const dim3 grid((unsigned)((A.nrows + 16 * ta - 1) / (16 * ta)), (unsigned)((batch + 16 * tb - 1) / (16 * tb)));
  // the panel streams (v0, accumulator, new term) of a large panel are read / written once per term: nontemporal, as in the
  // sparse panel kernels
  const bool nt = (double)A.nrows * batch * sizeof(double2) >= 128.0 * 1024 * 1024;
#define QP_DENSE_ZGEMM(OP, TA_, TB_)                                                                                     \
  do {                                                                                                                   \
    OP op{e};                                                                                                            \
    if (A.vals_r)                                                                                                        \
      hipLaunchKernelGGL((dense_zgemm_cheby_kernel<OP, double, 6, TA_, TB_>), grid, dim3(256), 0, s, A.vals_r, X, (int)A.nrows, (int)A.ncols, batch, op); \
    else                                                                                                                 \
      hipLaunchKernelGGL((dense_zgemm_cheby_kernel<OP, double2, 6, TA_, TB_>), grid, dim3(256), 0, s, A.vals, X, (int)A.nrows, (int)A.ncols, batch, op);  \
  } while (0)
  if (ta == 2) {
    if (nt) QP_DENSE_ZGEMM(ChebyOpT<true>, 2, 2); else QP_DENSE_ZGEMM(ChebyOp, 2, 2);
  } else if (tb == 2) {
    if (nt) QP_DENSE_ZGEMM(ChebyOpT<true>, 1, 2); else QP_DENSE_ZGEMM(ChebyOp, 1, 2);
  } else {
    if (nt) QP_DENSE_ZGEMM(ChebyOpT<true>, 1, 1); else QP_DENSE_ZGEMM(ChebyOp, 1, 1);
  }
#undef QP_DENSE_ZGEMM
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
    st->spmv_bytes += (A.vals_r ? 8.0 : 16.0) * (double)A.nrows * (double)A.ncols + 80.0 * (double)A.nrows * batch;
  }
  return QP_OK;
}

}  // namespace qp

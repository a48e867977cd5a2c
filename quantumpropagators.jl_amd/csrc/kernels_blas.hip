// Elementwise / BLAS-1 kernels, the operator's plane combination (evaluate!), the Gram-Schmidt passes of arnoldi! and the
// Newton combines (split out of kernels.hip in round 4).
#include <cstring>
#include <type_traits>

#include "kernel_common.h"

namespace qp {

// ---------------------------------------------------------------------------
// elementwise / BLAS-1
// ---------------------------------------------------------------------------

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
// SOLVE (one GPU): every workgroup first sums the kRedBlocks multidot partials of all 2 (j + 1)
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

// ORD: the elements a workgroup owns and the order in which it reads the basis are chosen for the
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

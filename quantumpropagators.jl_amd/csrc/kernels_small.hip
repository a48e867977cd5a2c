// Small systems: the whole time grid of `propagate` / a whole Arnoldi sweep as ONE persistent single-workgroup launch
// (split out of kernels.hip in round 4).
#include <cstring>
#include <type_traits>

#include "kernel_common.h"

namespace qp {

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

}  // namespace qp

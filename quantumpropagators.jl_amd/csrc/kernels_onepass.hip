// One pass over the Krylov basis from memory per Arnoldi column (knob arnoldi_onepass) -- gfx950, wave64.
//
// The low-synchronisation sweep (kernels_arnoldi.hip + kernels_blas.hip) reads the basis twice per column, in two launches:
// for the dot products c_k = <q_k | H q_j> in the mat-vec's epilogue, and again for the projection w - sum_k h_k q_k once the
// dot products are reduced.  The two cannot be merged as they stand: the next mat-vec gathers from the PROJECTED vector, which
// no row knows before the reduction.  They can with the mat-vec moved to the other side of the projection:
//
//   a_t = H qh_t  is kept as a vector of its own (two buffers in rotation).  With the projection coefficients h of a_t known,
//   qh_{t+1} = s (a_t - sum_k h_k qh_k),   and by linearity   a_{t+1} = H qh_{t+1} = s (H a_t - sum_k h_k a_k),
//   where every a_k is itself a combination of basis vectors -- the Arnoldi relation a_k = sum_{i<=k} Hh_ik qh_i + qh_{k+1} / s_{k+1}
//   holds by construction of qh_{k+1} -- so that sum_k h_k a_k = sum_{i<=t+1} gamma_i qh_i with gamma = Hh h, a small product.
//
// ONE kernel per column therefore gathers from a_t (complete since the previous launch) and, row block by row block, forms the
// new basis row and the new a row from the block's basis rows, then -- with those rows read again while they are still in the
// caches -- the dot products d_k = <qh_k | a_{t+1}> of the NEXT column and the Gram row <qh_k | qh_{t+1}>, measured directly:
// the solve turns d and the Gram matrix into the reference's sequential modified-Gram-Schmidt coefficients
// (src/arnoldi.jl:84-87), as the low-synchronisation sweep does.  The solve of column t - 1 (reduction of the 256 partials per
// value, forward substitution, gamma) is the PROLOGUE of column kernel t: every workgroup repeats it in LDS (same numbers, same
// order), workgroup 0 also records it -- Gram row, Hessenberg column, the host's copy and the column's flag.
//
// The scale s is only an ESTIMATE of 1 / |a_t - sum h q| (Pythagoras, from reduced quantities): the stored vectors qh_k have
// norm nu_k close to, not exactly, one; nu_k is measured in the same pass (the Gram diagonal) and every quantity the host sees
// is converted to the orthonormal basis q_k = qh_k / nu_k: Hess_ij = dt Hh_ij nu_i / nu_j, the breakdown test on the true norm
// nu_{j+1} / (s nu_j), the combination coefficients divided by nu (engine_krylov.hip).  All of it is linear algebra on exact
// identities; what changes against the reference is rounding (tests/test_gpu_onepass.py: newton! to 1e-10 of the oracle after
// every step, the oracle's restart counts).
#include <type_traits>

#include "kernel_common.h"

namespace qp {

constexpr int kOpWavesMax = 8;         // most wavefronts per workgroup of the column kernel (template parameter WS: 8 or 4)
constexpr int kOnePassMaxVec = 20;     // most basis vectors a kernel instance streams (columns 0 .. 20 of a sweep: m <= 20)
constexpr int kOpLd = kOnePassMaxVec + 2;   // leading dimension of the LDS copies of the small matrices

// value slots of a column's partials: d_k at k, Gram g_k at (nvec + 1) + k, |a|^2 at 2 (nvec + 1)
int op_part_slots(int nvec) { return 2 * (nvec + 1) + 1; }

struct OpSolveArgs {
  const double2* partials;   // of column kernel t
  int t, m, nvec;
  double2* gram;             // device, nvec x nvec: G[i][k] = <qh_i | qh_k> at [i nvec + k]
  double2* hhat;             // device, nvec x nvec column major: Hh[i][k] at [k nvec + i]
  double* svals;             // device: s_k
  double* nu_dev;            // device: nu_k
  double dt;
  double2* hess_map;         // host (mapped): Hess column major, leading dimension nvec
  double* norms_map;         // host (mapped): norm of the projected vector per column
  double* nu_map;            // host (mapped): nu_k
  unsigned* flags_map;       // host (mapped): column j complete <=> flags[j] == flag_value
  unsigned flag_value;
};

struct OpSolveLds {
  double2 red[2 * kOpLd + 1];
  double2 Gs[kOpLd * kOpLd];   // G[i][k] at [i kOpLd + k]
  double2 Hs[kOpLd * kOpLd];   // Hh[i][k] at [k kOpLd + i]
  double2 hs[kOpLd], rs[kOpLd], gam[kOpLd];
  double nus[kOpLd];
  double s_next;
};

// The solve after column kernel t (it wrote qh_t and, unless it was the last, a_t), by a whole workgroup of NW wavefronts:
// reduces the kernel's partials, completes row / column t of the Gram matrix, and
//   * (t < m) solves for the coefficients of a_t against qh_0 .. qh_t in the reference's sequential order
//       h_i = (d_i - sum_{k<i} G_ik h_k) / G_ii            (src/arnoldi.jl:84-87 on vectors of norm nu, not one)
//     estimates the scale s_{t+1} = 1 / sqrt(|a_t|^2 - sum |h_k|^2 G_kk) and forms gamma = Hh h: left in L.hs (zero beyond t),
//     L.gam (zero beyond t + 1), L.s_next;
//   * writer (one workgroup per launch): records the Gram row, column t of Hh, s, nu_t, and (t >= 1) converts column t - 1 to the
//     orthonormal basis for the host: Hess[i, t-1] = dt Hh[i, t-1] nu_i / nu_{t-1}, i <= t, the norm of the projected vector
//     nu_t / (s_t nu_{t-1}), then raises the column's flag.
// Every thread of the workgroup calls it; the results are valid after the __syncthreads() the CALLER does next.
template <int NW>
__device__ __forceinline__ void onepass_solve(OpSolveLds& L, const OpSolveArgs& A, bool writer) {
  static_assert(kRedBlocks == 256, "four partials per lane");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int t = A.t, nvec = A.nvec;
  const bool has_d = t < A.m;
  // the earlier rows of the Gram matrix, the earlier columns of Hh and the norms: one parallel copy into LDS (the serial part
  // below must not wait for memory once per entry)
  for (int e = threadIdx.x; e < t * t; e += 64 * NW) {
    const int i = e / t, k = e - i * t;
    L.Gs[i * kOpLd + k] = A.gram[(size_t)i * nvec + k];
  }
  for (int e = threadIdx.x; e < t * (t + 1); e += 64 * NW) {
    const int k = e / (t + 1), i = e - k * (t + 1);   // column k < t, row i <= t
    L.Hs[k * kOpLd + i] = A.hhat[(size_t)k * nvec + i];
  }
  if ((int)threadIdx.x < t) L.nus[threadIdx.x] = A.nu_dev[threadIdx.x];
  // the values of this column: d_0 .. d_t (has_d), g_0 .. g_t, |a|^2 (has_d); one wavefront per value, round robin -- every
  // load of the wavefront's values is issued before the first sum (a loop that loads, sums and stores one value at a time
  // pays the memory latency once per value)
  const int nval = 2 * (t + 1) + 1;
  constexpr int kPerWave = (2 * (kOnePassMaxVec + 1) + 1 + NW - 1) / NW;
  double2 acc[kPerWave];
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int v = wave + NW * i;
    acc[i] = make_double2(0.0, 0.0);
    if (v < nval) {      // (wave-uniform)
      const int slot = v <= t ? v : (v <= 2 * t + 1 ? (nvec + 1) + (v - (t + 1)) : 2 * (nvec + 1));
      const bool need = has_d || (v > t && v <= 2 * t + 1);
      if (need) {
        const double2* p = A.partials + (size_t)slot * kRedBlocks;
        const double2 x0 = p[lane], x1 = p[64 + lane], x2 = p[128 + lane], x3 = p[192 + lane];
        acc[i] = make_double2((x0.x + x1.x) + (x2.x + x3.x), (x0.y + x1.y) + (x2.y + x3.y));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int v = wave + NW * i;
    if (v < nval) {
      const double re = wave_sum(acc[i].x), im = wave_sum(acc[i].y);
      if (lane == 0) L.red[v] = make_double2(re, im);
    }
  }
  __syncthreads();
  if (wave != 0) return;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // Gram row / column t: g_k = <qh_k | qh_t>
  if (lane <= t) {
    const double2 gk = L.red[(t + 1) + lane];
    L.Gs[lane * kOpLd + t] = gk;                                   // G[k][t]
    L.Gs[t * kOpLd + lane] = make_double2(gk.x, -gk.y);            // G[t][k]
    if (writer) {
      A.gram[(size_t)lane * nvec + t] = gk;
      A.gram[(size_t)t * nvec + lane] = make_double2(gk.x, -gk.y);
    }
  }
  if (lane < kOpLd) L.hs[lane] = L.gam[lane] = make_double2(0.0, 0.0);
  wsync();
  const double gtt = L.Gs[t * kOpLd + t].x;
  const double nu_t = sqrt(gtt > 0.0 ? gtt : 0.0);
  if (lane == 0) L.nus[t] = nu_t;
  if (has_d) {
    // forward substitution, column by column: once h_k is final every later row subtracts G_ik h_k (row i accumulates in
    // ascending k: the order of the sequential projections, src/arnoldi.jl:84-87)
    if (lane <= t) L.rs[lane] = L.red[lane];
    wsync();
    for (int k = 0; k <= t; ++k) {
      const double gkk = L.Gs[k * kOpLd + k].x;
      const double2 rk = L.rs[k];
      const double2 hk = gkk > 0.0 ? make_double2(rk.x / gkk, rk.y / gkk) : make_double2(0.0, 0.0);   // (a zero vector: past a breakdown)
      if (lane == 0) L.hs[k] = hk;
      if (lane > k && lane <= t) {
        const double2 gik = L.Gs[lane * kOpLd + k];
        double2 r = L.rs[lane];
        r.x = fma(-gik.x, hk.x, r.x);
        r.x = fma(gik.y, hk.y, r.x);
        r.y = fma(-gik.x, hk.y, r.y);
        r.y = fma(-gik.y, hk.x, r.y);
        L.rs[lane] = r;
      }
      wsync();
    }
    // scale estimate: |a - sum h q|^2 ~ |a|^2 - sum |h_k|^2 G_kk
    double sub = 0.0;
    if (lane <= t) {
      const double2 h = L.hs[lane];
      sub = (h.x * h.x + h.y * h.y) * L.Gs[lane * kOpLd + lane].x;
    }
    sub = wave_sum(sub);
    const double aa = L.red[2 * (t + 1)].x;
    const double est = aa - sub;
    const double s_next = est > 1e-20 * aa ? 1.0 / sqrt(est) : (aa > 0.0 ? 1.0 / sqrt(aa) : 1.0);
    if (lane == 0) L.s_next = s_next;
    // column t of Hh: h_0 .. h_t, then 1 / s_{t+1}
    if (lane <= t) {
      L.Hs[t * kOpLd + lane] = L.hs[lane];
      if (writer) A.hhat[(size_t)t * nvec + lane] = L.hs[lane];
    }
    if (lane == t + 1) {
      L.Hs[t * kOpLd + lane] = make_double2(1.0 / s_next, 0.0);
      if (writer) {
        A.hhat[(size_t)t * nvec + lane] = make_double2(1.0 / s_next, 0.0);
        A.svals[t + 1] = s_next;
      }
    }
    wsync();
    // gamma_i = sum_{k >= i-1, k <= t} Hh[i][k] h_k, i <= t + 1 (lane i)
    if (lane <= t + 1) {
      double2 g = make_double2(0.0, 0.0);
      for (int k = (lane > 0 ? lane - 1 : 0); k <= t; ++k) {
        const double2 e = L.Hs[k * kOpLd + lane], hk = L.hs[k];
        g.x = fma(e.x, hk.x, g.x);
        g.x = fma(-e.y, hk.y, g.x);
        g.y = fma(e.x, hk.y, g.y);
        g.y = fma(e.y, hk.x, g.y);
      }
      L.gam[lane] = g;
    }
  }
  if (!writer) return;
  if (lane == 0) {
    A.nu_map[t] = nu_t;      // (host-mapped: for the host; the device keeps its own copy)
    A.nu_dev[t] = nu_t;
  }
  if (t >= 1) {
    // column t - 1 in the orthonormal basis, for the host
    const double nu_p = L.nus[t - 1];
    const double s_t = A.svals[t];
    const double inv = nu_p > 0.0 ? 1.0 / nu_p : 0.0;
    double2* hc = A.hess_map + (size_t)(t - 1) * nvec;
    const double hn = nu_t * inv / s_t;                            // |projected vector| in the orthonormal basis   src/arnoldi.jl:89
    if (lane < t) {
      const double2 e = L.Hs[(t - 1) * kOpLd + lane];
      const double f = A.dt * L.nus[lane] * inv;
      hc[lane] = make_double2(e.x * f, e.y * f);                   // Hess[i, t-1] = dt <q_i | H q_{t-1}>           :85
    } else if (lane == t) {
      hc[t] = make_double2(A.dt * hn, 0.0);                        // Hess[t, t-1] = dt h                           :90
      A.norms_map[t - 1] = hn;
    }
    // every lane's stores to the host buffer are complete before the flag goes up
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(A.flags_map + (t - 1), A.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the solve on its own: after the LAST column kernel of a sweep (no kernel follows whose prologue it could be)
__global__ __launch_bounds__(256) void arnoldi_onepass_solve_kernel(OpSolveArgs A) {
  __shared__ OpSolveLds L;
  onepass_solve<4>(L, A, true);
}

// Column kernel.  nb: basis vectors that exist (qh_0 .. qh_{nb-1}); the kernel writes qh_nb into q_out and (MV) a_nb = H qh_nb
// into a_out, and leaves the partials of d_k = <qh_k | a_nb> (k <= nb), g_k = <qh_k | qh_nb> (k <= nb), |a_nb|^2.
// nb >= 1: S = the solve of column nb - 1 (its prologue); nb = 0: the start of a sweep -- qh_0 = s0 psi, a_0 = H qh_0.
// WS wavefronts per workgroup: 8 while the accumulators leave room for two wavefronts per SIMD (256 registers each), else 4 with
// the register file of a SIMD to one wavefront (no spills: the 2 (JT + 1) complex accumulators alone are 168 registers at JT = 20)
template <int JT, class VT, bool NT, bool MV, int WS>
__global__ __launch_bounds__(64 * WS) void arnoldi_onepass_kernel(
    const int64_t* __restrict__ bptr, const int64_t* __restrict__ cmeta, const char* __restrict__ colbytes,
    const VT* __restrict__ vals, const double2* __restrict__ a_in, int64_t nblocks, int64_t nrows,
    const double2* __restrict__ Q, int64_t ldq, int nb, OpSolveArgs S, double s0, double2* __restrict__ q_out,
    double2* __restrict__ a_out, double2* __restrict__ partials) {
  static_assert(JT % 4 == 0 && JT <= kOnePassMaxVec, "chunks of four basis vectors");
  constexpr int NV = 4 * (JT + 2);     // doubles to reduce: d (JT + 1 complex), g (JT + 1 complex), |a|^2, three zeros
  static_assert(NV % 8 == 0, "the lane transpose takes eight values at a time");
  __shared__ OpSolveLds L;
  __shared__ double red_tile[WS][64 * 9];
  __shared__ double red_parts[WS][4][NV];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  if (nb > 0) {
    onepass_solve<WS>(L, S, blockIdx.x == 0);
  } else if (threadIdx.x < kOpLd) {
    L.hs[threadIdx.x] = L.gam[threadIdx.x] = make_double2(0.0, 0.0);
    if (threadIdx.x == 0) L.s_next = s0;
  }
  __syncthreads();
  const double s = L.s_next;
  const double2 gam_own = L.gam[nb];   // gamma_nb: multiplies the row of the new basis vector
  // (the start of a sweep has no basis yet: the streams then repeat rows of a_in -- finite numbers, zero weights)
  const double2* __restrict__ Qs = nb > 0 ? Q : a_in;
  const int64_t ldqs = nb > 0 ? ldq : 0;
  const int nbl = nb > 0 ? nb - 1 : 0;   // last basis vector that is loaded (the slots beyond repeat it: a line that is in the L1)
  double2 d[JT + 1], g[JT + 1];
#pragma unroll
  for (int k = 0; k <= JT; ++k) d[k] = g[k] = make_double2(0.0, 0.0);
  double aa = 0.0;
  const int rounds = (int)((nblocks + (int64_t)gridDim.x * WS - 1) / ((int64_t)gridDim.x * WS));
  // NBK row blocks in flight per wavefront.  The accumulators are per WAVEFRONT, not per block, so a second block costs only its
  // streams' registers -- and with the register file of a SIMD to one wavefront (WS = 4) there is room: every phase of a round
  // (the gathers of the mat-vec, the first sweep over the basis rows, the second) then has two blocks' loads in flight, and a
  // column takes half as many rounds of dependent memory latencies (N = 2^18: 2 instead of 4).  Block order inside a wavefront and
  // the order of every sum are unchanged: the same bits as with one block at a time.
  // (JT <= 12 only: with 2 (JT + 1) = 34 / 42 complex accumulators the second block's streams no longer fit the 512 registers of a
  // wavefront that has the SIMD to itself -- 60 / 580 bytes of scratch per lane at JT = 16 / 20)
  constexpr int NBK = (WS == 4 && JT <= 12) ? 2 : 1;
  for (int t = 0; t < rounds; t += NBK) {
    bool valid[NBK];
    int64_t rows[NBK], bcs[NBK];
    unsigned ro[NBK];
    double2 ar[NBK], qa[NBK][4], qb[NBK][4], z[NBK], u[NBK], y[NBK], qn[NBK], an[NBK];
    bool act[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
      const int64_t b = ((int64_t)(t + j) * gridDim.x + wg) * WS + wave;
      act[j] = (t + j) < rounds && b < nblocks;   // (wave-uniform)
      bcs[j] = act[j] ? b : nblocks - 1;
      rows[j] = bcs[j] * kRB + lane;
      valid[j] = act[j] && rows[j] < nrows;
      const int64_t rowc = rows[j] < nrows ? rows[j] : nrows - 1;
      ro[j] = (unsigned)rowc;   // (the launcher takes this kernel only below 2^28 rows)
      ar[j] = a_in[ro[j]];
      // first chunk of the basis: in flight during the mat-vec
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[j][e] = (Qs + (size_t)min(e, nbl) * ldqs)[ro[j]];
      z[j] = make_double2(0.0, 0.0);
    }
    if constexpr (MV) {
      int64_t base[NBK], cm[NBK];
      int nq[NBK];
      const VT* __restrict__ v[NBK];
      double2 s0a[NBK], s1a[NBK];
      int nqm = 0;
#pragma unroll
      for (int j = 0; j < NBK; ++j) {
        base[j] = bptr[bcs[j]];
        nq[j] = act[j] ? (int)((bptr[bcs[j] + 1] - base[j]) >> 8) : 0;
        v[j] = vals + base[j] + lane;
        cm[j] = cmeta[bcs[j]];
        s0a[j] = s1a[j] = make_double2(0.0, 0.0);
        nqm = max(nqm, nq[j]);
      }
      (void)nqm;
      // (one block after the other: the gathers of a block are four to eight lines, and holding both blocks' values and operands at
      // once spilled 520 bytes per lane next to the 2 (JT + 1) accumulators; the sweeps over the basis rows below are interleaved)
#pragma unroll
      for (int j = 0; j < NBK; ++j) {
#pragma unroll 2
        for (int q = 0; q < nq[j]; ++q) {
          const int4 cc = ld_cols<NT>(colbytes, cm[j], q, lane, (int)ro[j]);
          const double2 a0 = ld_val<NT>(v[j] + (size_t)(4 * q + 0) * 64);
          const double2 a1 = ld_val<NT>(v[j] + (size_t)(4 * q + 1) * 64);
          const double2 a2 = ld_val<NT>(v[j] + (size_t)(4 * q + 2) * 64);
          const double2 a3 = ld_val<NT>(v[j] + (size_t)(4 * q + 3) * 64);
          const double2 x0 = a_in[cc.x];
          const double2 x1 = a_in[cc.y];
          const double2 x2 = a_in[cc.z];
          const double2 x3 = a_in[cc.w];
          cfma(s0a[j], a0, x0);
          cfma(s1a[j], a1, x1);
          cfma(s0a[j], a2, x2);
          cfma(s1a[j], a3, x3);
        }
      }
#pragma unroll
      for (int j = 0; j < NBK; ++j) z[j] = make_double2(s0a[j].x + s1a[j].x, s0a[j].y + s1a[j].y);
    }
    // ---- first sweep over the blocks' basis rows (from memory, chunks of four, one chunk ahead): the projection of a_t
    // (ascending k, as the sequential axpys of src/arnoldi.jl:86) and the same sum with gamma for H a_t
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
      u[j] = ar[j];
      y[j] = z[j];
    }
    auto project = [&](const double2 (&qq)[NBK][4], int k0) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double2 hk = L.hs[k0 + e], gk = (k0 + e) < nb ? L.gam[k0 + e] : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
          u[j].x = fma(-hk.x, qq[j][e].x, u[j].x);
          u[j].x = fma(hk.y, qq[j][e].y, u[j].x);
          u[j].y = fma(-hk.x, qq[j][e].y, u[j].y);
          u[j].y = fma(-hk.y, qq[j][e].x, u[j].y);
          if constexpr (MV) {
            y[j].x = fma(-gk.x, qq[j][e].x, y[j].x);
            y[j].x = fma(gk.y, qq[j][e].y, y[j].x);
            y[j].y = fma(-gk.x, qq[j][e].y, y[j].y);
            y[j].y = fma(-gk.y, qq[j][e].x, y[j].y);
          }
        }
      }
    };
#pragma unroll
    for (int kb = 0; kb < JT; kb += 8) {
      if (kb + 4 < JT) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) qb[j][e] = (Qs + (size_t)min(kb + 4 + e, nbl) * ldqs)[ro[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
      project(qa, kb);
      if (kb + 8 < JT) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) qa[j][e] = (Qs + (size_t)min(kb + 8 + e, nbl) * ldqs)[ro[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 4 < JT) project(qb, kb + 4);
    }
    // the second sweep's first chunk: requested before the new rows are formed and stored.  (Scheduling barriers around the
    // sweeps: the loads of the second one depend on nothing, and hoisted above the first they would hold every basis row in
    // registers at once -- the accumulators below leave no room for that.)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[j][e] = (Qs + (size_t)min(e, nbl) * ldqs)[ro[j]];
      qn[j] = make_double2(s * u[j].x, s * u[j].y);
      an[j] = make_double2(0.0, 0.0);
      if constexpr (MV) {
        y[j].x = fma(-gam_own.x, qn[j].x, y[j].x);
        y[j].x = fma(gam_own.y, qn[j].y, y[j].x);
        y[j].y = fma(-gam_own.x, qn[j].y, y[j].y);
        y[j].y = fma(-gam_own.y, qn[j].x, y[j].y);
        an[j] = make_double2(s * y[j].x, s * y[j].y);
      }
      if (valid[j]) {
        q_out[rows[j]] = qn[j];
        if constexpr (MV) a_out[rows[j]] = an[j];
      } else {
        qn[j] = an[j] = make_double2(0.0, 0.0);
      }
    }
    // ---- second sweep over the same rows (the lines are in the L1 / L2 now): d_k += conj(qh_k) a_new, g_k += conj(qh_k) qh_new;
    // accumulators beyond nb collect numbers nobody reads; slot JT: the new vector with itself.  (Block 0 then block 1 into the same
    // accumulator: the order in which one block at a time adds them.)
    auto dots = [&](const double2 (&qq)[NBK][4], int k0) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
          if constexpr (MV) {
            const double2 pd = cconj_mul(qq[j][e], an[j]);
            d[k0 + e].x += pd.x;
            d[k0 + e].y += pd.y;
          }
          const double2 pg = cconj_mul(qq[j][e], qn[j]);
          g[k0 + e].x += pg.x;
          g[k0 + e].y += pg.y;
        }
      }
    };
#pragma unroll
    for (int kb = 0; kb < JT; kb += 8) {
      if (kb + 4 < JT) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) qb[j][e] = (Qs + (size_t)min(kb + 4 + e, nbl) * ldqs)[ro[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
      dots(qa, kb);
      if (kb + 8 < JT) {
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) qa[j][e] = (Qs + (size_t)min(kb + 8 + e, nbl) * ldqs)[ro[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 4 < JT) dots(qb, kb + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
      if constexpr (MV) {
        const double2 pd = cconj_mul(qn[j], an[j]);
        d[JT].x += pd.x;
        d[JT].y += pd.y;
        aa += an[j].x * an[j].x + an[j].y * an[j].y;
      }
      g[JT].x += qn[j].x * qn[j].x + qn[j].y * qn[j].y;
    }
  }
  // NV sums over the lanes of every wavefront, then over the wavefronts (the transpose of kernels_arnoldi.hip: eight values
  // at a time through a wavefront-private LDS tile, row = lane, nine doubles wide; fixed order, no atomics)
  {
    double* __restrict__ tile = red_tile[wave];
    const int tv = lane & 7, tp = lane >> 3;
#pragma unroll
    for (int ch = 0; ch < NV / 8; ++ch) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int id = 8 * ch + i;   // d_k.re, d_k.im at 2 k, 2 k + 1 (k <= JT); g_k at 2 (JT + 1) + 2 k (+ 1); |a|^2 at 4 (JT + 1)
        double val = 0.0;
        if (id < 2 * (JT + 1)) val = (id & 1) ? d[id / 2].y : d[id / 2].x;
        else if (id < 4 * (JT + 1)) val = (id & 1) ? g[(id - 2 * (JT + 1)) / 2].y : g[(id - 2 * (JT + 1)) / 2].x;
        else if (id == 4 * (JT + 1)) val = aa;
        tile[lane * 9 + i] = val;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double sum = tile[(tp * 8) * 9 + tv];
#pragma unroll
      for (int i = 1; i < 8; ++i) sum += tile[(tp * 8 + i) * 9 + tv];
      sum += dpp_take<0x118, 0xf>(sum);   // row_shr:8: part 2 r + 1 += part 2 r
      if (tp & 1) red_parts[wave][tp >> 1][8 * ch + tv] = sum;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
      const int id = threadIdx.x;
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < WS; ++w)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) r += red_parts[w][pp][id];
      // slot of the value in the partials (by basis index, not by instance width): accumulator k < nb -> k, JT -> nb
      const int nvec = S.nvec;
      int vslot = -1;
      if (id < 2 * (JT + 1)) {
        const int k = id >> 1;
        if (k < nb) vslot = k;
        else if (k == JT) vslot = nb;
      } else if (id < 4 * (JT + 1)) {
        const int k = (id - 2 * (JT + 1)) >> 1;
        if (k < nb) vslot = (nvec + 1) + k;
        else if (k == JT) vslot = (nvec + 1) + nb;
      } else if (id == 4 * (JT + 1)) {
        vslot = 2 * (nvec + 1);
      }
      if (vslot >= 0) reinterpret_cast<double*>(partials + (size_t)vslot * kRedBlocks + blockIdx.x)[id & 1] = r;
    }
  }
}

template <int JT, bool NT, bool MV>
static void launch_op_instance(hipStream_t s, const DevMatrix& A, const double2* a_in, const double2* Q, int64_t ldq, int nb,
                               const OpSolveArgs& S, double s0, double2* q_out, double2* a_out, double2* partials) {
  constexpr int WS = JT <= 4 ? 8 : 4;
  if (A.vals_r)
    hipLaunchKernelGGL((arnoldi_onepass_kernel<JT, double, NT, MV, WS>), dim3(kRedBlocks), dim3(64 * WS), 0, s, A.bptr, A.cmeta,
                       reinterpret_cast<const char*>(A.cols), A.vals_r, a_in, A.nblocks, A.nrows, Q, ldq, nb, S, s0, q_out, a_out,
                       partials);
  else
    hipLaunchKernelGGL((arnoldi_onepass_kernel<JT, double2, NT, MV, WS>), dim3(kRedBlocks), dim3(64 * WS), 0, s, A.bptr, A.cmeta,
                       reinterpret_cast<const char*>(A.cols), A.vals, a_in, A.nblocks, A.nrows, Q, ldq, nb, S, s0, q_out, a_out,
                       partials);
}

bool arnoldi_onepass_fits(const DevMatrix& A, int m, int nvec) {
  return A.format == QP_FMT_RBCSR && (A.vals || A.vals_r) && A.nblocks >= 1 && A.nrows < (1 << 28) && m >= 1 && m <= kOnePassMaxVec &&
         nvec <= kOnePassMaxVec + 1 && !(A.cb && A.cb->valid && A.tun && A.tun->colblock != 0);
}

// One sweep of m columns: column kernels t = 0 .. m (the last without a mat-vec: no a_m is needed), the solve of column t - 1 in
// the prologue of kernel t, and the solve of column m on its own.  The partials ping-pong between two buffers (kernel t reads
// those of kernel t - 1 in its prologue while it writes its own).
int launch_arnoldi_onepass_sweep(hipStream_t s, const DevMatrix& A, const double2* start, double s0, double2* Q, int64_t ldq,
                                 double2* const a_buf[2], int m, int nvec, double2* const part[2], double2* gram, double2* hhat,
                                 double* svals, double* nu_dev, double dt, double2* hess_map, double* norms_map, double* nu_map,
                                 unsigned* flags_map, unsigned flag_value, Stats* st) {
  const bool nt = (double)A.stored * (A.vals_r ? 8.0 : 16.0) > 8.0 * 1024 * 1024;
  OpSolveArgs S;
  S.m = m;
  S.nvec = nvec;
  S.gram = gram;
  S.hhat = hhat;
  S.svals = svals;
  S.nu_dev = nu_dev;
  S.dt = dt;
  S.hess_map = hess_map;
  S.norms_map = norms_map;
  S.nu_map = nu_map;
  S.flags_map = flags_map;
  S.flag_value = flag_value;
  for (int t = 0; t <= m; ++t) {
    const double2* a_in = t == 0 ? start : a_buf[(t - 1) & 1];
    double2* q_out = Q + (size_t)t * ldq;
    double2* a_out = a_buf[t & 1];
    S.t = t - 1;                       // the prologue solves column t - 1 from the partials kernel t - 1 left
    S.partials = part[(t + 1) & 1];
    double2* pout = part[t & 1];
    const bool mv = t < m;
#define QP_OP_CASE(JT_)                                                                                       \
  do {                                                                                                        \
    if (!mv) launch_op_instance<JT_, false, false>(s, A, a_in, Q, ldq, t, S, s0, q_out, a_out, pout);          \
    else if (nt) launch_op_instance<JT_, true, true>(s, A, a_in, Q, ldq, t, S, s0, q_out, a_out, pout);        \
    else launch_op_instance<JT_, false, true>(s, A, a_in, Q, ldq, t, S, s0, q_out, a_out, pout);               \
  } while (0)
    if (t <= 4) QP_OP_CASE(4);
    else if (t <= 8) QP_OP_CASE(8);
    else if (t <= 12) QP_OP_CASE(12);
    else if (t <= 16) QP_OP_CASE(16);
    else QP_OP_CASE(20);
#undef QP_OP_CASE
    QP_HIP(hipGetLastError());
    if (st) {
      st->n_launch++;
      if (mv) {
        st->n_matvec++;
        st->spmv_bytes += 20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 32.0 * (double)A.nrows;
      }
    }
  }
  S.t = m;
  S.partials = part[m & 1];
  hipLaunchKernelGGL(arnoldi_onepass_solve_kernel, dim3(1), dim3(256), 0, s, S);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

}  // namespace qp

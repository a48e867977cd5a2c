// Device helpers of the low-synchronisation modified Gram-Schmidt (kernels_blas.hip: mgs_update_kernel and friends).  Included
// INSIDE namespace qp.
__device__ __forceinline__ int tri_index(int i, int k) { return i * (i - 1) / 2 + k; }  // k < i

// Forward substitution  h_i = c_i - sum_{k<i} <q_i|q_k> h_k  (the MGS coefficients, see above) by
// one wavefront, column by column: once h_k is final every later row subtracts its <q_i|q_k> h_k
// (no reduction; row i accumulates in ascending k).  `red` = [c_0..c_j | <q_0|q_j> .. <q_j|q_j>]
// and the packed strict lower triangle `Gt` of the Gram matrix (rows 1..j), both in LDS; h starts
// as a copy of c.  Leaves Hess[i,j] = dt h_i in hess_col and the axpy coefficients -Hess[i,j]/dt
// (src/arnoldi.jl:85-86) in coef.
__device__ __forceinline__ void mgs_solve_wave(int j, const double2* red, const double2* Gt, double2* h,
                                               double2* __restrict__ hess_col, double2* __restrict__ coef, double dt) {
  const int lane = threadIdx.x;
  for (int i = lane; i <= j; i += 64) h[i] = red[i];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int k = 0; k < j; ++k) {
    const double2 hk = h[k];
    for (int i = k + 1 + lane; i <= j; i += 64) {
      const double2 g = Gt[tri_index(i, k)];
      double2 v = h[i];
      v.x = fma(-g.x, hk.x, v.x);
      v.x = fma(g.y, hk.y, v.x);
      v.y = fma(-g.x, hk.y, v.y);
      v.y = fma(-g.y, hk.x, v.y);
      h[i] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i <= j; i += 64) {
    const double2 hd = make_double2(dt * h[i].x, dt * h[i].y);
    hess_col[i] = hd;
    coef[i] = make_double2(-hd.x / dt, -hd.y / dt);
  }
}

// Gram rows into LDS (packed lower triangle): rows 1..j-1 from G, row j = conj of the fresh
// <q_k|q_j> in red[(j+1)+k]; row j is also stored to G for the later columns.
__device__ __forceinline__ void mgs_stage_gram(int j, const double2* red, double2* Gt, double2* __restrict__ G, int ldg, int bs = kThreads) {
  const int total = j * (j + 1) / 2;
  for (int idx = threadIdx.x; idx < total; idx += bs) {
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)idx)) * 0.5f);
    while (i * (i - 1) / 2 > idx) --i;
    while ((i + 1) * i / 2 <= idx) ++i;
    const int k = idx - i * (i - 1) / 2;
    double2 g;
    if (i == j) {
      const double2 r = red[(j + 1) + k];
      g = make_double2(r.x, -r.y);
      G[(size_t)j * ldg + k] = g;
    } else {
      g = G[(size_t)i * ldg + k];
    }
    Gt[idx] = g;
  }
}


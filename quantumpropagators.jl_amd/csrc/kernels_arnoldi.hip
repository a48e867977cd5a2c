// One Arnoldi column's mat-vec with the dot products of the orthogonalisation in its epilogue (gfx950, wave64).
//
//   w = H q_j                       src/arnoldi.jl:82
//   c_k = <q_k | w>,  k <= j        src/arnoldi.jl:84-85 (all of them against the SAME w: the low-synchronisation form
//   g_k = <q_k | q_j>, k <= j        of modified Gram-Schmidt -- the projection kernel turns c and the Gram rows g into the
//                                   reference's sequential coefficients, kernels_blas.hip, mgs_common.h: mgs_solve_wave)
//
// The separate multidot launch read w and q_j again, once per tile of eight basis vectors, and cost a launch boundary per
// column; the lane that owns row i has w_i and (q_j)_i in registers when its row sum is complete.  Here every wavefront
// walks its row blocks in sequence and carries 2 JT complex accumulators per lane across them (JT = basis vectors the
// instance has room for); the loads of q_k[row], k < j, do not depend on the mat-vec and are issued in chunks of four,
// one chunk ahead of the arithmetic.  256 workgroups of eight wavefronts = the kRedBlocks partial sums per value that
// the projection kernel's prologue reduces in a fixed order (no atomics: run-to-run identical bits).
#include <type_traits>

#include "kernel_common.h"

namespace qp {

constexpr int kFusedWaves = 8;

// NT: the matrix values and column sections are loaded nontemporal -- the matrix is read once per
// column and does not fit an XCD's L2 next to the basis; streamed, it leaves the L2 to the basis vectors that the
// projection kernel reads next (kernels_blas.hip: mgs_update_kernel<.., ORD = true>)
// CODED: the operator has a value-dictionary mirror (device.h: CodedVals; kernels_coded.hip) -- `vals` is then the combined TABLE,
// a quad of slots costs one dword of four codes instead of four value loads, and the block's table (at most 256 entries) is staged
// in the wavefront's reduction tile, which is idle until the epilogue.  Same values in the same order: bit-identical sums.
template <int JT, class VT, bool NT, bool CODED = false>   // VT: double2, or double for the real copy of an all-real operator (kernel_common.h: ld_val)
__global__ __launch_bounds__(64 * kFusedWaves) void arnoldi_matvec_dots_kernel(
    const int64_t* __restrict__ bptr, const int64_t* __restrict__ cmeta, const char* __restrict__ colbytes,
    const VT* __restrict__ vals, const double2* __restrict__ x, int64_t nblocks, int64_t nrows, PlainEpi e,
    const double2* __restrict__ Q, int64_t ldq, int j, double2* __restrict__ partials,
    const unsigned* __restrict__ codes4 = nullptr, const int64_t* __restrict__ tptr = nullptr) {
  static_assert(!CODED || sizeof(VT) * 256 <= sizeof(double) * 64 * 9, "a block's table fits the wavefront's reduction tile");
  static_assert(JT % 4 == 0, "chunks of four basis vectors");
  static_assert(kRedBlocks == 256, "the folded norm: 256 partials, four wavefronts' worth");
  __shared__ double2 lds4[4];
  __shared__ double red_tile[kFusedWaves][64 * 9];
  __shared__ double red_parts[kFusedWaves][4][4 * JT];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  // the folded "norm + scale" of the previous column (PlainEpi): the partials' load first, the reduction after the first
  // row sums (a barrier every wavefront reaches: the loop below runs the same number of rounds in all of them)
  double2 np = make_double2(0.0, 0.0);
  if (e.norm_part && threadIdx.x < kRedBlocks) np = e.norm_part[threadIdx.x];
  double inv = 1.0;
  double2 c[JT], g[JT];
#pragma unroll
  for (int k = 0; k < JT; ++k) c[k] = g[k] = make_double2(0.0, 0.0);
  const int jl = j > 0 ? j - 1 : 0;   // last basis vector that is loaded (k < j; q_j itself is in registers)
  const int rounds = (int)((nblocks + (int64_t)gridDim.x * kFusedWaves - 1) / ((int64_t)gridDim.x * kFusedWaves));
  for (int t = 0; t < rounds; ++t) {
    const int64_t b = ((int64_t)t * gridDim.x + wg) * kFusedWaves + wave;
    const bool active = b < nblocks;   // (wave-uniform)
    const int64_t bc = active ? b : nblocks - 1;
    const int64_t base = bptr[bc];
    const int nq = active ? (int)((bptr[bc + 1] - base) >> 8) : 0;
    const VT* __restrict__ v = vals + (CODED ? 0 : base + lane);
    const int64_t cm = cmeta[bc];
    const int64_t row = bc * kRB + lane;
    const bool valid = active && row < nrows;
    const int64_t rowc = row < nrows ? row : nrows - 1;
    const unsigned ro = (unsigned)rowc;   // (the launcher takes this kernel only below 2^28 rows: 32-bit lane offsets)
    const double2 xi = x[ro];
    // first chunk of the basis: in flight during the mat-vec
    double2 qa[4], qb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) qa[u] = (Q + (size_t)min(u, jl) * ldq)[ro];
    double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
    if constexpr (CODED) {
      VT* __restrict__ tw = reinterpret_cast<VT*>(red_tile[wave]);
      const int64_t tp = tptr[bc];
      const VT* __restrict__ tb = vals + (tp >> 9);
      const int tlen = (int)(tp & 511);
      if (t > 0) {      // the previous block's table reads are done before this block's table overwrites them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i * 64 < tlen) tw[i * 64 + lane] = tb[min(i * 64 + lane, tlen - 1)];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned* __restrict__ cq = codes4 + (base >> 2) + lane;
#pragma unroll 2
      for (int q = 0; q < nq; ++q) {
        const unsigned cw = __builtin_nontemporal_load(cq + (size_t)q * 64);
        const int4 cc = ld_cols<true>(colbytes, cm, q, lane, (int)rowc);
        const double2 x0 = x[cc.x];
        const double2 x1 = x[cc.y];
        const double2 x2 = x[cc.z];
        const double2 x3 = x[cc.w];
        const double2 a0 = ld_val<false>(tw + (cw & 255u));
        const double2 a1 = ld_val<false>(tw + ((cw >> 8) & 255u));
        const double2 a2 = ld_val<false>(tw + ((cw >> 16) & 255u));
        const double2 a3 = ld_val<false>(tw + (cw >> 24));
        cfma(s0, a0, x0);
        cfma(s1, a1, x1);
        cfma(s0, a2, x2);
        cfma(s1, a3, x3);
      }
    } else {
#pragma unroll 2
    for (int q = 0; q < nq; ++q) {
      const int4 cc = ld_cols<NT>(colbytes, cm, q, lane, (int)rowc);
      const double2 a0 = ld_val<NT>(v + (size_t)(4 * q + 0) * 64);
      const double2 a1 = ld_val<NT>(v + (size_t)(4 * q + 1) * 64);
      const double2 a2 = ld_val<NT>(v + (size_t)(4 * q + 2) * 64);
      const double2 a3 = ld_val<NT>(v + (size_t)(4 * q + 3) * 64);
      const double2 x0 = x[cc.x];
      const double2 x1 = x[cc.y];
      const double2 x2 = x[cc.z];
      const double2 x3 = x[cc.w];
      cfma(s0, a0, x0);
      cfma(s1, a1, x1);
      cfma(s0, a2, x2);
      cfma(s1, a3, x3);
    }
    }
    if (t == 0 && e.norm_part) {
      np.x = wave_sum(np.x);
      np.y = wave_sum(np.y);
      if (lane == 0 && wave < 4) lds4[wave] = np;
      __syncthreads();
      double2 r = lds4[0];   // (the order of block_sum: the same h as the unfused mat-vec computes)
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        r.x += lds4[i].x;
        r.y += lds4[i].y;
      }
      const double h = sqrt(r.x);                        // h = norm(q[j])              src/arnoldi.jl:89
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (e.hess_slot) *e.hess_slot = make_double2(e.dt * h, 0.0);   // :90
        if (e.norm_slot) *e.norm_slot = h;
        if (e.flag) __hip_atomic_store(e.flag, e.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      inv = (h < e.norm_min) ? 1.0 : 1.0 / h;            // lmul!(1 / h, q[j])          :96  (not past a breakdown :91-95)
    }
    double2 wv = make_double2((s0.x + s1.x) * inv, (s0.y + s1.y) * inv);
    double2 qv = make_double2(xi.x * inv, xi.y * inv);
    if (valid) {
      e.y[row] = wv;
      if (e.qn_out) e.qn_out[row] = qv;
    } else {
      wv = qv = make_double2(0.0, 0.0);
    }
    // c_k += conj(q_k) w, g_k += conj(q_k) q_j over this lane's row; k == j takes the lane's own element.  Accumulators
    // beyond j collect numbers nobody reads (their loads repeat a line that is in the L1).
#pragma unroll
    for (int kb = 0; kb < JT; kb += 8) {
      if (kb + 4 < JT) {
#pragma unroll
        for (int u = 0; u < 4; ++u) qb[u] = (Q + (size_t)min(kb + 4 + u, jl) * ldq)[ro];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + u;
        const double2 qq = (k == j) ? qv : qa[u];
        const double2 pc = cconj_mul(qq, wv), pg = cconj_mul(qq, qv);
        c[k].x += pc.x;
        c[k].y += pc.y;
        g[k].x += pg.x;
        g[k].y += pg.y;
      }
      if (kb + 8 < JT) {
#pragma unroll
        for (int u = 0; u < 4; ++u) qa[u] = (Q + (size_t)min(kb + 8 + u, jl) * ldq)[ro];
      }
      if (kb + 4 < JT) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = kb + 4 + u;
          const double2 qq = (k == j) ? qv : qb[u];
          const double2 pc = cconj_mul(qq, wv), pg = cconj_mul(qq, qv);
          c[k].x += pc.x;
          c[k].y += pc.y;
          g[k].x += pg.x;
          g[k].y += pg.y;
        }
      }
    }
  }
  // 4 JT sums over the lanes of every wavefront, then over the wavefronts.  One cross-lane tree per value would be 80
  // dependent chains at JT = 20; instead the lanes transpose eight values at a time through a wavefront-private LDS tile
  // (row = lane, nine doubles wide: conflict-free both ways): lane l then owns value l % 8 and adds the entries of the
  // eight lanes 8 (l / 8) .. 8 (l / 8) + 7 in order -- all 64 lanes busy, reads independent of one another --, one row
  // shift folds the eight parts into four, and the workgroup's last stage adds 8 wavefronts x 4 parts per value in a
  // fixed order.
  {
    constexpr int NV = 4 * JT;
    double* __restrict__ tile = red_tile[wave];
    if constexpr (CODED) {      // the tile held the last block's table until here
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int tv = lane & 7, tp = lane >> 3;
#pragma unroll
    for (int ch = 0; ch < NV / 8; ++ch) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int id = 8 * ch + i;                       // value ids: c_k.re, c_k.im at 2 k, 2 k + 1; g_k at 2 JT + 2 k (+ 1)
        const double2 a = id < 2 * JT ? c[id / 2] : g[(id - 2 * JT) / 2];
        tile[lane * 9 + i] = (id & 1) ? a.y : a.x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double sum = tile[(tp * 8) * 9 + tv];
#pragma unroll
      for (int i = 1; i < 8; ++i) sum += tile[(tp * 8 + i) * 9 + tv];
      sum += dpp_take<0x118, 0xf>(sum);                  // row_shr:8: part 2 r + 1 (lanes 8 .. 15 of a row) += part 2 r
      if (tp & 1) red_parts[wave][tp >> 1][8 * ch + tv] = sum;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this chunk's reads before the next chunk's writes
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
      const int id = threadIdx.x;
      const bool is_g = id >= 2 * JT;
      const int k = (is_g ? id - 2 * JT : id) >> 1;
      if (k <= j) {
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < kFusedWaves; ++w)
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) r += red_parts[w][pp][id];
        const int vslot = (is_g ? j + 1 : 0) + k;          // the multidot's layout: c_0 .. c_j, g_0 .. g_j
        reinterpret_cast<double*>(partials + (size_t)vslot * kRedBlocks + blockIdx.x)[id & 1] = r;
      }
    }
  }
}

template <int JT, bool NT>
static void launch_instance_nt(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const double2* Q,
                               int64_t ldq, int j, double2* partials) {
  if (A.vals_r)
    hipLaunchKernelGGL((arnoldi_matvec_dots_kernel<JT, double, NT>), dim3(kRedBlocks), dim3(64 * kFusedWaves), 0, s, A.bptr,
                       A.cmeta, reinterpret_cast<const char*>(A.cols), A.vals_r, x, A.nblocks, A.nrows, e, Q, ldq, j, partials, nullptr, nullptr);
  else
    hipLaunchKernelGGL((arnoldi_matvec_dots_kernel<JT, double2, NT>), dim3(kRedBlocks), dim3(64 * kFusedWaves), 0, s, A.bptr,
                       A.cmeta, reinterpret_cast<const char*>(A.cols), A.vals, x, A.nblocks, A.nrows, e, Q, ldq, j, partials, nullptr, nullptr);
}
template <int JT>
static void launch_instance(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const double2* Q,
                            int64_t ldq, int j, double2* partials) {
  // an operator with a value dictionary: codes + the block's table instead of the value plane
  if (A.cv && A.cv->valid && A.tun && A.tun->value_dict != 0) {
    const CodedVals& C = *A.cv;
    const unsigned* codes4 = reinterpret_cast<const unsigned*>(C.codes);
    if (C.use_real)
      hipLaunchKernelGGL((arnoldi_matvec_dots_kernel<JT, double, true, true>), dim3(kRedBlocks), dim3(64 * kFusedWaves), 0, s, A.bptr,
                         A.cmeta, reinterpret_cast<const char*>(A.cols), C.tab_r, x, A.nblocks, A.nrows, e, Q, ldq, j, partials, codes4, C.tptr);
    else
      hipLaunchKernelGGL((arnoldi_matvec_dots_kernel<JT, double2, true, true>), dim3(kRedBlocks), dim3(64 * kFusedWaves), 0, s, A.bptr,
                         A.cmeta, reinterpret_cast<const char*>(A.cols), C.tab, x, A.nblocks, A.nrows, e, Q, ldq, j, partials, codes4, C.tptr);
    return;
  }
  // (only where the operator is large enough for the question to exist: a small one sits in the L2 with its basis)
  const bool nt = (double)A.stored * (A.vals_r ? 8.0 : 16.0) > 8.0 * 1024 * 1024;
  if (nt) launch_instance_nt<JT, true>(s, A, x, e, Q, ldq, j, partials);
  else launch_instance_nt<JT, false>(s, A, x, e, Q, ldq, j, partials);
}

// the largest j (basis vectors 0 .. j) with a kernel instance
constexpr int kFusedMaxJ = 19;

// *launched = false: no instance for this operator / column (the caller then takes mat-vec + multidot)
int launch_arnoldi_matvec_dots(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const double2* Q,
                               int64_t ldq, int j, double2* partials, bool* launched, Stats* st) {
  *launched = false;
  if (A.format != QP_FMT_RBCSR || (!A.vals && !A.vals_r) || j < 0 || j > kFusedMaxJ || A.nblocks < 1 || A.nrows >= (1 << 28)) return QP_OK;
  // an operator with a column-blocked mirror: its plain mat-vec (kernels_colblock.hip) + the separate multidot beat the fused
  // kernel, whose gathers are what the mirror exists to fix
  if (A.cb && A.cb->valid && A.tun && A.tun->colblock != 0) return QP_OK;
  if (!e.beta_zero || e.alpha.x != 1.0 || e.alpha.y != 0.0) return QP_OK;
  if (e.xloc && e.xloc != x) return QP_OK;
  if (j < 4) launch_instance<4>(s, A, x, e, Q, ldq, j, partials);
  else if (j < 8) launch_instance<8>(s, A, x, e, Q, ldq, j, partials);
  else if (j < 12) launch_instance<12>(s, A, x, e, Q, ldq, j, partials);
  else if (j < 16) launch_instance<16>(s, A, x, e, Q, ldq, j, partials);
  else launch_instance<20>(s, A, x, e, Q, ldq, j, partials);
  QP_HIP(hipGetLastError());
  *launched = true;
  if (st) {
    st->n_launch++;
    st->n_matvec++;
    st->spmv_bytes += 20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 32.0 * (double)A.nrows;
  }
  return QP_OK;
}

}  // namespace qp

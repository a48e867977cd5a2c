// A whole Arnoldi sweep (src/arnoldi.jl:60-100: m columns of mat-vec + modified Gram-Schmidt + norm) as ONE persistent
// launch with the Krylov basis RESIDENT ON THE CHIP -- gfx950 / wave64, systems of at most 2^18 rows (config C3's size).
//
// Why.  The two kernels of a column (kernels_arnoldi.hip: mat-vec + dot products; kernels.hip: mgs_update_kernel) each read
// the basis q_0 .. q_j once -- the global reduction between the dots and the projection IS the orthogonalisation -- and the
// basis (21 x 4 MiB at C3) does not fit the 32 MiB of L2: 65 % of a column's traffic is basis re-read out of the Infinity
// Cache at ~4.4 TB/s (a CU's L1 holds ~94 line requests against ~700 ns of latency; profiles/r04/newton_c3_notes.txt).
// But a lane only ever needs ITS OWN rows of the basis (dots and projection are row-local), and the chip's register files
// hold 128 MiB and its LDS 40 MiB: one workgroup of eight wavefronts per CU, two rows per lane = 2^18 rows, the first 10 basis
// vectors in registers (80 VGPRs -- the mat-vec and the reductions need the rest: with 12 the compiler spills), the next 7 in
// LDS (112 KiB), the last ones (17 .. 20 at m = 20) re-read from memory: 48 MB of basis reads per sweep instead of 1680.  A
// column then moves the matrix, the gathered vector and the two vectors it publishes -- 40 % of the bytes -- and the kernel
// boundaries become grid barriers (all 256 workgroups are co-resident: hipLaunchCooperativeKernel).
//
// Per column j (the arithmetic and its order per row are those of the two-kernel path; reductions in fixed orders: results
// are run-to-run identical bits):
//   phase A   h = |u_j| from the previous phase's 256 partials (every workgroup, the order of block_sum); w = H u_j / h (the
//             folded "norm + scale", src/arnoldi.jl:89-96), q_j = u_j / h into its basis slot and out to Q[j] (the restart
//             combination reads the basis from memory afterwards); c_k = <q_k|w>, g_k = <q_k|q_j>, k <= j, over this
//             workgroup's rows -> partials
//   barrier
//   phase B   every workgroup sums the 256 partials of the 2 (j + 1) values, stages the Gram rows and solves for the MGS
//             coefficients (mgs_common.h, as mgs_update_kernel's prologue does); u_{j+1} = w - sum_k h_k q_k out of registers /
//             LDS, published (unnormalised) for the next column's gathers; |u|^2 partial
//   barrier
// Hessenberg entries, norms and the per-column flags go to the host-mapped buffers exactly as the two-kernel path writes
// them (engine_krylov.hip: arnoldi_impl polls the same flags).
#include <atomic>
#include <type_traits>

#include "kernel_common.h"

namespace qp {

#include "mgs_common.h"

constexpr int kSweepWaves = 8;                 // one workgroup of eight wavefronts per compute unit
constexpr int kSweepThreads = 64 * kSweepWaves;
constexpr int kSweepRegs = 10;                 // basis vectors 0 .. 9 live in registers (2 rows per lane: 80 VGPRs; 233 in all, no spills)
constexpr int kSweepLds = 7;                   // basis vectors 10 .. 16 live in LDS (7 x 2 x 512 x 16 B = 112 KiB)
constexpr int kSweepRes = kSweepRegs + kSweepLds;      // the later ones (17 .. m) are re-read from Q: 48 MB per sweep at m = 20, 2^18 rows (the two-kernel path: 1680 MB)
constexpr int kSweepMaxVec = 21;               // m + 1 <= 21 (the solve's LDS scratch)


// all workgroups of the (cooperative) launch; `target` = arrivals expected so far
__device__ __forceinline__ void sweep_grid_barrier(unsigned* counter, unsigned target, unsigned* error) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores are out
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) {      // a few seconds: never hang the queue for good (with a cooperative launch this cannot happen)
        __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <class VT, bool NT>
__global__ __launch_bounds__(kSweepThreads) void arnoldi_sweep_resident_kernel(SweepArgs a) {
  extern __shared__ double2 dyn[];
  // LDS: basis vectors 16 .. 20 | reduction tiles of the eight wavefronts | reduced parts | solve scratch
  double2* Ql = dyn;                                                        // [kSweepLds][2][512]
  double* red_tile = reinterpret_cast<double*>(Ql + (size_t)kSweepLds * 2 * kSweepThreads);     // [8][64 * 9]
  double* red_parts = red_tile + kSweepWaves * 64 * 9;                      // [8][4][16]   (one chunk of 4 basis vectors at a time)
  double2* lds4 = reinterpret_cast<double2*>(red_parts + kSweepWaves * 4 * 16);                  // [8]
  double2* red = lds4 + kSweepWaves;                                        // [2 nvec]
  double2* hs = red + 2 * kSweepMaxVec;                                     // [nvec] Hess column (dt h)
  double2* hc = hs + kSweepMaxVec;                                          // [nvec] axpy coefficients (-h)
  double2* Gt = hc + kSweepMaxVec;                                          // packed lower triangle of the Gram matrix
  __shared__ double2 dummy_col[kSweepMaxVec];

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const VT* __restrict__ vals = static_cast<const VT*>(a.vals);
  // this lane's two rows (the ownership of kernels_arnoldi.hip): round t covers blocks (t grid + wg) 8 + wave
  int64_t rowc[2];
  bool valid[2], active[2];
  int64_t blk[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int64_t b = ((int64_t)t * gridDim.x + wg) * kSweepWaves + wave;
    active[t] = b < a.nblocks;
    blk[t] = active[t] ? b : a.nblocks - 1;
    const int64_t row = blk[t] * kRB + lane;
    valid[t] = active[t] && row < a.nrows;
    rowc[t] = row < a.nrows ? row : a.nrows - 1;
  }
  double2 Qr[kSweepRegs][2];
#pragma unroll
  for (int k = 0; k < kSweepRegs; ++k) Qr[k][0] = Qr[k][1] = make_double2(0.0, 0.0);
  double2 u[2];      // this lane's elements of the current unnormalised vector u_j
#pragma unroll
  for (int t = 0; t < 2; ++t) u[t] = valid[t] ? a.Q[rowc[t]] : make_double2(0.0, 0.0);      // u_0 = q_0 (normalised by the caller)
  unsigned nbar = 0;

  // the sums of one chunk of four basis vectors (16 real values per lane: c.re, c.im, g.re, g.im each) over the 64 lanes of every
  // wavefront -- through the wavefront's LDS tile, eight values at a time, as kernels_arnoldi.hip does -- and over the eight
  // wavefronts; thread id < 16 of the workgroup stores value id of the chunk (basis vectors k0 .. k0 + 3, those <= j only)
  // element of q_k, k >= kSweepRegs (wave-uniform k): LDS for the next kSweepLds vectors, Q (this lane's own earlier store) beyond
  auto basis_mem = [&](int k, int t) -> double2 {
    if (k < kSweepRes) return Ql[((size_t)(k - kSweepRegs) * 2 + t) * kSweepThreads + threadIdx.x];
    return a.Q[(size_t)k * a.ldq + rowc[t]];
  };
  // one chunk of four basis vectors q_{k0} .. q_{k0+3} (elements q[i][t] of this lane): c = conj(q) w and g = conj(q) q_j summed over
  // the lane's two rows, then over the 64 lanes -- two basis vectors (eight real values) at a time through the wavefront's LDS
  // tile, as kernels_arnoldi.hip does -- and over the eight wavefronts; thread id < 16 stores value id of the chunk
  auto dots_chunk = [&](int k0, int j, const double2& q00, const double2& q01, const double2& q10, const double2& q11, const double2& q20,
                        const double2& q21, const double2& q30, const double2& q31, const double2 (&wv)[2], const double2 (&qv)[2]) {
    double* __restrict__ tile = red_tile + wave * 64 * 9;
    const int tv = lane & 7, tp = lane >> 3;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double2 e0 = ch == 0 ? (i == 0 ? q00 : q10) : (i == 0 ? q20 : q30);
        const double2 e1 = ch == 0 ? (i == 0 ? q01 : q11) : (i == 0 ? q21 : q31);
        const double2 c0 = cconj_mul(e0, wv[0]), c1 = cconj_mul(e1, wv[1]);
        const double2 g0 = cconj_mul(e0, qv[0]), g1 = cconj_mul(e1, qv[1]);
        tile[lane * 9 + 4 * i + 0] = c0.x + c1.x;
        tile[lane * 9 + 4 * i + 1] = c0.y + c1.y;
        tile[lane * 9 + 4 * i + 2] = g0.x + g1.x;
        tile[lane * 9 + 4 * i + 3] = g0.y + g1.y;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double sum = tile[(tp * 8) * 9 + tv];
#pragma unroll
      for (int i = 1; i < 8; ++i) sum += tile[(tp * 8 + i) * 9 + tv];
      sum += dpp_take<0x118, 0xf>(sum);                  // row_shr:8: part 2 r + 1 += part 2 r
      if (tp & 1) red_parts[(wave * 4 + (tp >> 1)) * 16 + 8 * ch + tv] = sum;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      const int id = threadIdx.x, k = k0 + (id >> 2);
      if (k <= j) {
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < kSweepWaves; ++w)
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) r += red_parts[(w * 4 + pp) * 16 + id];
        const bool is_g = (id & 2) != 0;
        const int vslot = (is_g ? j + 1 : 0) + k;        // the multidot's layout: c_0 .. c_j, g_0 .. g_j
        reinterpret_cast<double*>(a.partials + (size_t)vslot * kRedBlocks + blockIdx.x)[id & 1] = r;
      }
    }
    __syncthreads();
  };

  for (int j = 0; j < a.m; ++j) {
    // ---------------- phase A: h = |u_j|, w = H u_j / h, q_j = u_j / h, the column's dot products ----------------
    const double2* __restrict__ x = (j == 0) ? a.Q : ((j & 1) ? a.raw1 : a.raw0);
    double inv = 1.0;
    if (j > 0) {
      double2 np = make_double2(0.0, 0.0);
      if (threadIdx.x < kRedBlocks) np = a.norm_part[(size_t)(j & 1) * kRedBlocks + threadIdx.x];
      np.x = wave_sum(np.x);
      np.y = wave_sum(np.y);
      if (lane == 0 && wave < 4) lds4[wave] = np;
      __syncthreads();
      double2 r = lds4[0];   // (the order of block_sum over 256 threads: the same h as the two-kernel path computes)
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        r.x += lds4[i].x;
        r.y += lds4[i].y;
      }
      const double h = sqrt(r.x);                                   // h = norm(q[j])              src/arnoldi.jl:89
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.hess[(size_t)(j - 1) * a.ldh + j] = make_double2(a.dt * h, 0.0);   // Hess[j, j-1]      :90
        a.norms[j - 1] = h;
        __hip_atomic_store(a.flags + (j - 1), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // column j - 1 is complete on the host
      }
      inv = (h < a.norm_min) ? 1.0 : 1.0 / h;                       // lmul!(1 / h, q[j])          :96  (not past a breakdown :91-95)
      __syncthreads();
    }
    double2 wv[2], qv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t base = a.bptr[blk[t]];
      const int nq = active[t] ? (int)((a.bptr[blk[t] + 1] - base) >> 8) : 0;
      const VT* __restrict__ v = vals + base + lane;
      const int64_t cm = a.cmeta[blk[t]];
      double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll 2
      for (int q = 0; q < nq; ++q) {
        const int4 cc = ld_cols<NT>(a.colbytes, cm, q, lane, (int)rowc[t]);
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
      wv[t] = make_double2((s0.x + s1.x) * inv, (s0.y + s1.y) * inv);
      qv[t] = make_double2(u[t].x * inv, u[t].y * inv);
      __builtin_amdgcn_sched_barrier(0);      // (the two rows one after the other: their loads' registers are reused)
      if (valid[t]) {
        if (j > 0) a.Q[(size_t)j * a.ldq + rowc[t]] = qv[t];       // (q_0 is in memory already)
      } else {
        wv[t] = qv[t] = make_double2(0.0, 0.0);
      }
    }
    // q_j into its basis slot (beyond the resident ones: it is in Q[j], written above)
    if (j >= kSweepRes) {
    } else if (j >= kSweepRegs) {
#pragma unroll
      for (int t = 0; t < 2; ++t) Ql[((size_t)(j - kSweepRegs) * 2 + t) * kSweepThreads + threadIdx.x] = qv[t];
    } else {
#pragma unroll
      for (int K = 0; K < kSweepRegs; ++K)
        if (K == j) {
          Qr[K][0] = qv[0];
          Qr[K][1] = qv[1];
        }
    }
    // c_k = <q_k|w>, g_k = <q_k|q_j>, k <= j, in chunks of four basis vectors: the register-resident ones by compile-time index
    // (slots beyond j hold zeros: their sums are not stored), then the LDS-resident ones
#pragma unroll
    for (int C = 0; C < kSweepRegs / 4; ++C) {
      if (4 * C <= j)
        dots_chunk(4 * C, j, Qr[4 * C][0], Qr[4 * C][1], Qr[4 * C + 1][0], Qr[4 * C + 1][1], Qr[4 * C + 2][0], Qr[4 * C + 2][1],
                   Qr[4 * C + 3][0], Qr[4 * C + 3][1], wv, qv);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int k0 = kSweepRegs; k0 <= j; k0 += 4) {
      double2 qq[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = min(k0 + i, j);      // (slots beyond j repeat q_j: their sums are not stored)
#pragma unroll
        for (int t = 0; t < 2; ++t) qq[i][t] = (k == j) ? qv[t] : basis_mem(k, t);
      }
      dots_chunk(k0, j, qq[0][0], qq[0][1], qq[1][0], qq[1][1], qq[2][0], qq[2][1], qq[3][0], qq[3][1], wv, qv);
    }
    nbar += gridDim.x;
    sweep_grid_barrier(a.barrier, nbar, a.error);

    // ---------------- phase B: sum the partials, solve for the MGS coefficients, project, |u|^2 ----------------
    {
      const int nv = 2 * (j + 1);
      for (int v = wave; v < nv; v += kSweepWaves) {       // one value per wavefront and round: lane l adds partials l, l + 64, l + 128, l + 192
        const double2* __restrict__ pp = a.partials + (size_t)v * kRedBlocks + lane;
        const double2 q0 = pp[0], q1 = pp[64], q2 = pp[128], q3 = pp[192];
        double2 r = make_double2(((q0.x + q1.x) + q2.x) + q3.x, ((q0.y + q1.y) + q2.y) + q3.y);
        r.x = wave_sum(r.x);
        r.y = wave_sum(r.y);
        if (lane == 0) red[v] = r;
      }
      __syncthreads();
      mgs_stage_gram(j, red, Gt, a.G, a.ldg, kSweepThreads);
      __syncthreads();
      if (threadIdx.x < 64) mgs_solve_wave(j, red, Gt, hs, blockIdx.x == 0 ? a.hess + (size_t)j * a.ldh : dummy_col, hc, a.dt);
      // (lane 0 of the wavefront that stored the column: its release covers those stores)
      if (j + 1 == a.m && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.flags + a.nvec + j, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the last column's early flag
      __syncthreads();
    }
    double nrm = 0.0;
    double2* __restrict__ uout = ((j + 1) & 1) ? a.raw1 : a.raw0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double2 r = wv[t];
#pragma unroll
      for (int K = 0; K < kSweepRegs; ++K)
        if (K <= j) cfma(r, hc[K], Qr[K][t]);                        // MGS order (hc = -h: the axpy coefficients)
      for (int k = kSweepRegs; k <= j; ++k) cfma(r, hc[k], basis_mem(k, t));
      u[t] = r;
      if (valid[t]) {
        uout[rowc[t]] = r;
        nrm += r.x * r.x + r.y * r.y;
      }
    }
    {
      const double v = wave_sum(nrm);
      if (lane == 0) lds4[wave] = make_double2(v, 0.0);
      __syncthreads();
      if (threadIdx.x == 0) {
        double tsum = lds4[0].x;
        for (int k = 1; k < kSweepWaves; ++k) tsum += lds4[k].x;
        a.norm_part[(size_t)((j + 1) & 1) * kRedBlocks + blockIdx.x] = make_double2(tsum, 0.0);
      }
    }
    nbar += gridDim.x;
    sweep_grid_barrier(a.barrier, nbar, a.error);
  }
  // ---------------- the last vector: h = |u_m|, Hess[m, m-1], q_m = u_m / h (extended sweep) ----------------
  {
    const int j = a.m;
    double2 np = make_double2(0.0, 0.0);
    if (threadIdx.x < kRedBlocks) np = a.norm_part[(size_t)(j & 1) * kRedBlocks + threadIdx.x];
    np.x = wave_sum(np.x);
    np.y = wave_sum(np.y);
    if (lane == 0 && wave < 4) lds4[wave] = np;
    __syncthreads();
    double2 r = lds4[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      r.x += lds4[i].x;
      r.y += lds4[i].y;
    }
    const double h = sqrt(r.x);
    const double inv = (h < a.norm_min) ? 1.0 : 1.0 / h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (valid[t]) a.Q[(size_t)j * a.ldq + rowc[t]] = make_double2(u[t].x * inv, u[t].y * inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.hess[(size_t)(j - 1) * a.ldh + j] = make_double2(a.dt * h, 0.0);
      a.norms[j - 1] = h;
      __hip_atomic_store(a.flags + (j - 1), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static size_t sweep_lds_bytes() {
  return sizeof(double2) * (size_t)kSweepLds * 2 * kSweepThreads + sizeof(double) * (kSweepWaves * 64 * 9 + kSweepWaves * 4 * 16) +
         sizeof(double2) * (kSweepWaves + 4 * kSweepMaxVec + kSweepMaxVec * (kSweepMaxVec + 1) / 2);
}

template <class VT, bool NT>
static int launch_sweep_instance(hipStream_t s, const SweepArgs& a, int grid, bool* launched) {
  auto kern = &arnoldi_sweep_resident_kernel<VT, NT>;
  const size_t lds = sweep_lds_bytes();
  static std::atomic<unsigned char> opted[64];   // per device: 0 = not tried, 1 = granted, 2 = refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return QP_OK;
  unsigned char st = opted[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : 2;
    if (st == 1) {   // ... and every workgroup of the grid must be resident at once
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kSweepThreads, lds) != hipSuccess || per_cu < 1) st = 2;
    }
    if (st == 2) (void)hipGetLastError();
    opted[dev].store(st, std::memory_order_release);
  }
  if (st != 1) return QP_OK;
  SweepArgs args = a;
  void* params[] = {&args};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3((unsigned)grid), dim3(kSweepThreads), params,
                                                  (unsigned)lds, s);
  if (e != hipSuccess) {      // not co-resident on this device / partition: the caller takes the two-kernel path
    (void)hipGetLastError();
    opted[dev].store(2, std::memory_order_release);
    return QP_OK;
  }
  *launched = true;
  return QP_OK;
}

// *launched = false: no instance for this operator / size / device (the caller then enqueues the columns one by one)
int launch_arnoldi_sweep_resident(hipStream_t s, const DevMatrix& A, const SweepArgs& a, int n_cu, bool* launched, Stats* st) {
  *launched = false;
  if (A.format != QP_FMT_RBCSR || (!A.vals && !A.vals_r) || a.m < 1 || a.m + 1 > kSweepMaxVec) return QP_OK;
  if (n_cu < kRedBlocks || A.nblocks > (int64_t)2 * kRedBlocks * kSweepWaves || A.nblocks < 1) return QP_OK;   // two rows per lane, 256 workgroups
  const bool nt = A.tun && A.tun->arnoldi_nt != 0 && (double)A.stored * (A.vals_r ? 8.0 : 16.0) > 8.0 * 1024 * 1024;
  int rc;
  if (A.vals_r) rc = nt ? launch_sweep_instance<double, true>(s, a, kRedBlocks, launched) : launch_sweep_instance<double, false>(s, a, kRedBlocks, launched);
  else rc = nt ? launch_sweep_instance<double2, true>(s, a, kRedBlocks, launched) : launch_sweep_instance<double2, false>(s, a, kRedBlocks, launched);
  if (rc != QP_OK) return rc;
  if (*launched && st) {
    st->n_launch++;
    st->n_matvec += (uint64_t)a.m;
    st->spmv_bytes += (double)a.m * (20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 32.0 * (double)A.nrows);
  }
  return QP_OK;
}

}  // namespace qp

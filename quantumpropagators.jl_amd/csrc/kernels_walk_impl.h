// (kernels_walk_impl.h: the kernel template and its instantiating launchers, included by the four translation units
// kernels_walk.hip / _b / _c / _d so that `make -j` compiles the 70-odd kernel shapes in parallel)
//
// Strip-walk kernel for the fused Chebyshev term of a Hermitian-packed LATTICE operator (src/cheby.jl:171-211 with the
// mat-vec of src/generators.jl:634-645 inside), gfx950 / wave64.
//
// Why: the per-block kernel (kernels.hip: hrb_spmv_kernel) sends 34 KiB per 64-row block through the compute unit's
// vector L1 -- 8 KiB of upper values, 8 KiB of conj-transposed values and 16 KiB of gathered vector elements, of which
// only the first 8 KiB and two of the vector streams come from HBM.  Beyond the Infinity Cache the L1's miss queue, not
// HBM, is what is full (profiles/r03/hrb_n22_pmc_diag.txt: texture addresser busy 85 %, the L1 stalled on its pending
// misses 58 % of the launch, 209 L1 -> L2 requests per block, 83 in flight per CU).  On a lattice all of that re-read
// data is data the SAME wavefront would load anyway if it walked down a strip column (see WalkPlan in device.h):
//   * x[r + m g]              = the row-local element of the block m steps away           -> register ring, 1 load/step
//   * conj H[r - m g, r]      = the far upper value streamed m steps ago                  -> FIFO in LDS, 0 loads
//   * x[r +- d], conj H[r - d, r] (d <= 16) = lane shifts of the block's own element / near values
//                                                                                         -> per-wavefront LDS window
// so that a block costs 8 value loads + 3 vector loads + 2 packed halo loads: 94 instead of 209 L1 -> L2 requests,
// nearly all of them HBM streams, requested one step ahead.  N = 2^22: 174 -> 126 us per term (0.56 -> 0.79 of 8 TB/s).
//
// Summation order per row is that of the per-block kernel (lower slots then upper slots in storage order, two
// interleaved partial sums), so the two kernels agree bit for bit (tests/test_gpu_parity.py).
#pragma once
#include <atomic>
#include <type_traits>

#include "kernel_common.h"

namespace qp {

struct HrbArrays {   // what the per-block path of the edge blocks reads
  const int64_t* uptr;
  const int64_t* ucmeta;
  const char* ucolbytes;
  const int64_t* lptr;
  const int64_t* lcmeta;
  const char* lcolbytes;
  const int4* lpos4;
};

struct WalkGeom {
  int L = 0;           // steps per wavefront
  int nseg = 0;        // segments of L steps per strip column
  int n_walk_wg = 0;
  int ntask = 0;       // wavefronts of the walk (n_walk_wg x wavefronts per workgroup)
  // Edge blocks (outside the walkable run), two schemes:
  //  * beside the walk (n_edge_wg > 0): workgroups of their own at the head of the grid, one block per wavefront, while
  //    every workgroup of the launch still finds room on the chip at once -- the walk is cut so that it does (768
  //    wavefronts inside the Infinity Cache, 8 per CU on all but the CUs the edge workgroups take beyond it);
  //  * inside the walk (n_edge_wg == 0; knob walk_waves / walk_dbg): edge block i goes to wavefront i, BEFORE its walk
  //    (edge_last: after), and the segments of those wavefronts are `edge_steps` steps shorter -- a block on the per-block
  //    path is three dependent rounds of loads, a step of the walk about one -- so that every wavefront finishes at
  //    about the same time.  (As leading workgroups of a launch that fills every CU they cost 5-6 us: whichever compute
  //    units ran them started their walk that much later.)
  int edge_segs = 0;   // segments 0 .. edge_segs - 1 are the shorter ones
  int edge_steps = 0;
  int edge_last = 0;
  int64_t xlast = 0;   // last element of x (columns of a row-partitioned operator run beyond its rows: the halo slabs)
  int n_edge_wg = 0;
};

constexpr int kWalkWaves = 8;   // most wavefronts (adjacent strip columns) per workgroup; the launch may use fewer (knob walk_wg)

// One row block by the per-block rules of hrb_spmv_kernel (same sums), arranged for LATENCY: a wavefront of the walk
// takes its edge block alone, so the block is three dependent rounds of loads -- block pointers; column sections,
// upper values and row-local operands; conj-transposed values and gathers -- and not one round per quad.  Sections of
// up to three quads each (12 + 12 entries per row: what the wrap-around blocks of a 16-entry lattice have); wider blocks take the loop form below.
template <class VT>
__device__ __forceinline__ void hrb_edge_block_loop(const HrbArrays& H, const VT* __restrict__ uvals,
                                                    const double2* __restrict__ x, int64_t b, int lane, int64_t nrows,
                                                    const ChebyOp& op, int64_t ubase, int64_t lbase, int nuq, int nlq,
                                                    int64_t ucm, int64_t lcm) {
  const VT* __restrict__ v = uvals + ubase + lane;
  const int4* __restrict__ lp4 = H.lpos4 + (lbase >> 2) + lane;
  const int64_t row = b * kRB + lane;
  const int64_t rowc = row < nrows ? row : nrows - 1;
  const ChebyOp::Pre pre = op.pre(rowc);
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
  if ((lcm & 3) == 2) {
    const LowerStencilSlot* __restrict__ ls = reinterpret_cast<const LowerStencilSlot*>(H.lcolbytes + (lcm >> 2));
    for (int k = 0; k < 4 * nlq; k += 2) {
      const LowerStencilSlot e0 = ls[k], e1 = ls[k + 1];
      const int c0 = (int)rowc + e0.delta, c1 = (int)rowc + e1.delta;
      const double2 a0 = ld_val<false>(uvals + (((c0 >> 6) == e0.cb0 ? e0.pb0 : e0.pb1) + (c0 & 63)));
      const double2 a1 = ld_val<false>(uvals + (((c1 >> 6) == e1.cb0 ? e1.pb0 : e1.pb1) + (c1 & 63)));
      cfma_conj(s0, a0, x[c0]);
      cfma_conj(s1, a1, x[c1]);
    }
  } else {
    for (int q = 0; q < nlq; ++q) {
      const int4 c = ld_cols<true>(H.lcolbytes, lcm, q, lane, (int)rowc);
      const int4 p = ld_col<true>(lp4 + (size_t)q * 64);
      cfma_conj(s0, ld_tr(uvals, p.x), x[c.x]);
      cfma_conj(s1, ld_tr(uvals, p.y), x[c.y]);
      cfma_conj(s0, ld_tr(uvals, p.z), x[c.z]);
      cfma_conj(s1, ld_tr(uvals, p.w), x[c.w]);
    }
  }
  for (int q = 0; q < nuq; ++q) {
    const int4 c = ld_cols<true>(H.ucolbytes, ucm, q, lane, (int)rowc);
    cfma(s0, ld_val<false>(v + (size_t)(4 * q + 0) * 64), x[c.x]);
    cfma(s1, ld_val<false>(v + (size_t)(4 * q + 1) * 64), x[c.y]);
    cfma(s0, ld_val<false>(v + (size_t)(4 * q + 2) * 64), x[c.z]);
    cfma(s1, ld_val<false>(v + (size_t)(4 * q + 3) * 64), x[c.w]);
  }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, 0);
}

template <class VT>
__device__ __forceinline__ void hrb_edge_block(const HrbArrays& H, const VT* __restrict__ uvals,
                                               const double2* __restrict__ x, int64_t b, int lane, int64_t nrows,
                                               const ChebyOp& op) {
  constexpr int MQ = 3;
  // round 1 (b is wave-uniform: scalar loads)
  const int64_t ubase = H.uptr[b], lbase = H.lptr[b];
  const int nuq = (int)((H.uptr[b + 1] - ubase) >> 8);
  const int nlq = (int)((H.lptr[b + 1] - lbase) >> 8);
  const int64_t ucm = H.ucmeta[b], lcm = H.lcmeta[b];
  if (nuq > MQ || nlq > MQ) {
    hrb_edge_block_loop<VT>(H, uvals, x, b, lane, nrows, op, ubase, lbase, nuq, nlq, ucm, lcm);
    return;
  }
  const VT* __restrict__ v = uvals + ubase + lane;
  const int4* __restrict__ lp4 = H.lpos4 + (lbase >> 2) + lane;
  const int64_t row = b * kRB + lane;
  const int64_t rowc = row < nrows ? row : nrows - 1;
  // round 2: everything whose address the block pointers give
  const ChebyOp::Pre pre = op.pre(rowc);
  const bool lst = (lcm & 3) == 2;
  const LowerStencilSlot* __restrict__ ls = reinterpret_cast<const LowerStencilSlot*>(H.lcolbytes + (lcm >> 2));
  int4 uc[MQ], lc[MQ], lp[MQ];
  double2 ua[4 * MQ];
#pragma unroll
  for (int q = 0; q < MQ; ++q) {
    if (q < nuq) {
      uc[q] = ld_cols<true>(H.ucolbytes, ucm, q, lane, (int)rowc);
#pragma unroll
      for (int k = 0; k < 4; ++k) ua[4 * q + k] = ld_val<false>(v + (size_t)(4 * q + k) * 64);
    }
    if (q < nlq) {
      if (lst) {   // column = row + delta, position of the transposed value = pb(column block) + column % 64
        const LowerStencilSlot e0 = ls[4 * q], e1 = ls[4 * q + 1], e2 = ls[4 * q + 2], e3 = ls[4 * q + 3];
        const int c0 = (int)rowc + e0.delta, c1 = (int)rowc + e1.delta, c2 = (int)rowc + e2.delta, c3 = (int)rowc + e3.delta;
        lc[q] = make_int4(c0, c1, c2, c3);
        lp[q] = make_int4((int)(((c0 >> 6) == e0.cb0 ? e0.pb0 : e0.pb1) + (c0 & 63)),
                          (int)(((c1 >> 6) == e1.cb0 ? e1.pb0 : e1.pb1) + (c1 & 63)),
                          (int)(((c2 >> 6) == e2.cb0 ? e2.pb0 : e2.pb1) + (c2 & 63)),
                          (int)(((c3 >> 6) == e3.cb0 ? e3.pb0 : e3.pb1) + (c3 & 63)));
      } else {
        lc[q] = ld_cols<true>(H.lcolbytes, lcm, q, lane, (int)rowc);
        lp[q] = ld_col<true>(lp4 + (size_t)q * 64);
      }
    }
  }
  // round 3: the conj-transposed values and every gathered element
  double2 la[4 * MQ], lx[4 * MQ], ux[4 * MQ];
#pragma unroll
  for (int q = 0; q < MQ; ++q) {
    if (q < nlq) {
      la[4 * q + 0] = ld_tr(uvals, lp[q].x);
      la[4 * q + 1] = ld_tr(uvals, lp[q].y);
      la[4 * q + 2] = ld_tr(uvals, lp[q].z);
      la[4 * q + 3] = ld_tr(uvals, lp[q].w);
      lx[4 * q + 0] = x[lc[q].x];
      lx[4 * q + 1] = x[lc[q].y];
      lx[4 * q + 2] = x[lc[q].z];
      lx[4 * q + 3] = x[lc[q].w];
    }
    if (q < nuq) {
      ux[4 * q + 0] = x[uc[q].x];
      ux[4 * q + 1] = x[uc[q].y];
      ux[4 * q + 2] = x[uc[q].z];
      ux[4 * q + 3] = x[uc[q].w];
    }
  }
  double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll
  for (int q = 0; q < MQ; ++q)
    if (q < nlq) {
      cfma_conj(s0, la[4 * q + 0], lx[4 * q + 0]);
      cfma_conj(s1, la[4 * q + 1], lx[4 * q + 1]);
      cfma_conj(s0, la[4 * q + 2], lx[4 * q + 2]);
      cfma_conj(s1, la[4 * q + 3], lx[4 * q + 3]);
    }
#pragma unroll
  for (int q = 0; q < MQ; ++q)
    if (q < nuq) {
      cfma(s0, ua[4 * q + 0], ux[4 * q + 0]);
      cfma(s1, ua[4 * q + 1], ux[4 * q + 1]);
      cfma(s0, ua[4 * q + 2], ux[4 * q + 2]);
      cfma(s1, ua[4 * q + 3], ux[4 * q + 3]);
    }
  double2 chk = make_double2(0.0, 0.0);
  double nrm = 0.0;
  if (row < nrows) op.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, 0);
}

// (by value: a conditional expression over two lvalues selects an ADDRESS and would force both objects into scratch)
__device__ __forceinline__ double2 sel2(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }

// what a step needs from memory (everything else is carried over from the steps before)
template <int NU, int XL = 0, int FD = 0>
struct WalkStep {
  double2 ua[NU];      // the block's upper values (pads skipped)
  double2 xnew;        // x[row + K g]
  double2 v0, acc;     // row-local operands of the epilogue
  // halos of the near windows, sixteen lanes each (lane = 16 q + t, only t < distance is used):
  double2 hx;          //   q = 0: x[r0 - dmax + t]         q = 1: x[r0 + 64 + t]
  double2 ha;          //   q = i: value (64 - d_i + t) of slot z0 + i of block b - 1
  // the long pairs (XL of them): x[row + L_p], x[row - L_p] and the conj-transposed value of the lower entry (row, row - L_p)
  double2 xlu[XL > 0 ? XL : 1], xll[XL > 0 ? XL : 1], al[XL > 0 ? XL : 1];
  // diagonal far neighbours (FD): the elements just outside the wavefront's 64 rows at the ring's far steps -- lane e < 4 K:
  // m = e / 4 + 1, x[r0 - 1 + m g], x[r0 + 64 + m g], x[r0 - 1 - m g], x[r0 + 64 - m g] -- and the two conj-transposed values
  // per m that the neighbouring strip columns streamed -- lane e < 2 K: m = e / 2 + 1, rows r0 - 1 - m g (slot m g + 1) and
  // r0 + 64 - m g (slot m g - 1)
  double2 hxf, hvf;
};

// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double2 bcast2(double2 v, int src) {
  return make_double2(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.x), src), __builtin_amdgcn_readlane(__double2loint(v.x), src)),
                      __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.y), src), __builtin_amdgcn_readlane(__double2loint(v.y), src)));
}
// the element of lane + 1 (UP = true) or lane - 1 in every lane (one DPP wavefront shift per dword); the lane without a
// source gets `edge`
template <bool UP>
__device__ __forceinline__ double lane_shift1(double v) {
  constexpr int CTRL = UP ? 0x130 : 0x138;   // wave_shl:1 (dst[i] = src[i + 1]) / wave_shr:1 (dst[i] = src[i - 1])
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <bool UP>
__device__ __forceinline__ double2 lane_shift2(double2 v, double2 edge, int lane) {
  const double2 t = make_double2(lane_shift1<UP>(v.x), lane_shift1<UP>(v.y));
  const bool e = UP ? lane == 63 : lane == 0;
  return make_double2(e ? edge.x : t.x, e ? edge.y : t.y);
}

// LDS of one wavefront, in double2 elements: the near window of x (16 + 64 + 16), NN near value windows (16 + 64) and the
// FIFOs of the far upper values (slot m: m entries of 64; with diagonal far neighbours, FD = 1, three slots per m)
template <int NN, int K, int FD = 0>
struct WalkLds {
  static constexpr int XW = kRB + 2 * kWalkHalo, AW = kRB + kWalkHalo;
  static constexpr int kHist = XW + NN * AW;
  static constexpr int kPerWave = kHist + kRB * (1 + 2 * FD) * (K * (K + 1) / 2);
  static constexpr size_t kBytesPerWave = sizeof(double2) * (size_t)kPerWave;
};

// One wavefront waits for the other launch of a split term (the per-wavefront form of kernel_common.h: sync_wait): lane 0
// polls the completion counter with a bounded spin, the wavefront then acquires at agent scope.  Only wavefronts that
// are about to take an EDGE block call it (the walk of a row set never reads a row of the other launch).
__device__ __forceinline__ void wave_sync_wait(const SyncArgs& sy) {
  if (!sy.wait) return;
  if ((threadIdx.x & 63) == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(sy.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.wait_target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > sy.spin_limit) {
        if (sy.timeout_flag) __hip_atomic_store(sy.timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// NTM: nontemporal accesses (bit 0: the matrix values, bit 1: the vector loads, bit 2: the stores)
// XL = 1: one more pair of distances +-L beyond the ring's reach (L > K g, any number of rows: the plane distance nx ny of
// a three-dimensional grid walked in steps of g = nx).  Its operands are not in the ring -- they would be ny steps away --
// and are loaded directly, one step ahead like everything else: x[row + L], x[row - L], the upper value of slot
// z0 + nn + K and the value stored for (row - L, row).  Storage order of the sections: the long entry is the first of
// the lower and the last of the upper one.
// XL = 2: two such pairs, K g < L_0 < L_1 (P.glong1, P.glong) -- the fourth-order Laplacian of a three-dimensional grid
// (+-1, +-2; +-nx, +-2 nx; +-nx ny, +-2 nx ny: near 2, far 2, long 2), next-nearest planes, a four-dimensional grid.  Lower
// section [-L_1] [-L_0] [far] [near], upper section [diag] [near] [far] [L_0] [L_1].
// FD = 1: DIAGONAL far neighbours -- the far distances of strip step m are m g - 1, m g, m g + 1 (the nine-point stencil of a
// two-dimensional grid with next-nearest hopping: +-1, +-(g - 1), +-g, +-(g + 1)).  The gathered elements x[r + m g +- 1] are
// the ring's elements of the NEIGHBOURING lanes (one DPP wavefront shift per dword; the lane at the edge takes the element
// just outside the wavefront's rows from a packed halo load), the conj-transposed values of the lower entries are read from
// the FIFO of their slot one lane over (the edge lane: a value the neighbouring strip column streamed, from the same halo load).
template <class VT, int NN, int K, int Z0, int NTM, int XL = 0, int FD = 0>
__global__ __launch_bounds__(64 * kWalkWaves) void hrb_walk_kernel(const VT* __restrict__ uvals,
                                                                    const double2* __restrict__ x, WalkPlan P,
                                                                    WalkGeom G, HrbArrays H, int64_t nrows, ChebyOp op,
                                                                    SyncArgs sy) {
  constexpr int FS = 1 + 2 * FD;          // far slots per strip step: [m g] or [m g - 1] [m g] [m g + 1]
  constexpr int KF = K * FS;
  constexpr int NL = XL + NN + KF;        // lower slots: [-L] [-K g .. -g] [-d_NN .. -d_1]
  constexpr int NU = Z0 + NN + KF + XL;   // upper slots that carry entries: [0] [d_1 .. d_NN] [g .. K g] [L]
  static_assert(FD == 0 || XL <= 1, "diagonal far neighbours come with at most one long pair");
  using Lds = WalkLds<NN, K, FD>;
  constexpr int XW = Lds::XW, AW = Lds::AW;
  static_assert(NN <= 4, "the near value halos share one register: sixteen lanes each");
  extern __shared__ double2 walk_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x < G.n_edge_wg) {
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (idx < P.n_edge) {
      wave_sync_wait(sy);
      hrb_edge_block<VT>(H, uvals, x, (int64_t)__builtin_amdgcn_readfirstlane(P.edge_map[idx]), lane, nrows, op);
    }
    return;
  }
  const unsigned wg = xcd_remap(blockIdx.x - G.n_edge_wg, G.n_walk_wg);
  const int task = (int)wg * (int)(blockDim.x >> 6) + wave;
  const int S = P.S;
  const int seg = task / S, col = task - seg * S;
  // rows of this wavefront at step j: W0 * 64 + j g + 64 col + lane; the strip step g need not be a multiple of 64 rows:
  // the last of the S = ceil(g / 64) column chunks is then partly filled (lanes beyond g are idle)
  const int64_t g = P.g;
  const int64_t wrows = (P.R1 - P.W0) * (int64_t)kRB;
  const int Jc = (int64_t)col * kRB < wrows ? (int)((wrows - (int64_t)col * kRB + g - 1) / g) : 0;
  const int j0 = seg * G.L - min(seg, G.edge_segs) * G.edge_steps;
  const int j1 = min((seg + 1) * G.L - min(seg + 1, G.edge_segs) * G.edge_steps, Jc);
  // this wavefront's edge block(s) first (see WalkGeom)
  if (!G.edge_last && G.n_edge_wg == 0 && task < P.n_edge) {
    wave_sync_wait(sy);
    for (int64_t idx = task; idx < P.n_edge; idx += G.ntask)
      hrb_edge_block<VT>(H, uvals, x, (int64_t)__builtin_amdgcn_readfirstlane(P.edge_map[idx]), lane, nrows, op);
  }
  if (seg < G.nseg && j0 < j1) {
  const int dmax = P.near[NN - 1];
  double2* __restrict__ xwin = walk_lds + (size_t)wave * Lds::kPerWave;
  double2* __restrict__ hring = xwin + Lds::kHist;
  // position of slot 0 of row r in the upper value array (r inside the run; rows of one 64-row block are contiguous)
  auto vpos = [&](int64_t r) __attribute__((always_inline)) -> int64_t {
    return P.U0 + ((r >> 6) - P.R0) * (int64_t)P.ustride + (r & 63);
  };
  const int64_t rmax = G.xlast;
  const int64_t vmax = P.R1 * (int64_t)kRB - 1;               // last row whose values sit at the run's strides
  // No branch inside the walk: a join in the control flow makes the compiler wait for EVERY outstanding load (the
  // prefetch included).  Operands that a term does not have are loaded from a line that stays in the L1 and ignored
  // by the epilogue; halo lanes beyond the halo repeat its last element.
  ChebyOpT<(NTM & 4) != 0> opl{op.e};
  opl.e.mirror = nullptr;           // (the launcher takes this kernel only without them)
  opl.e.check_partials = nullptr;
  const double2* __restrict__ v0p = op.e.v0;
  const double2* __restrict__ accp = op.e.acc_in;
  // halo lanes: group q = lane / 16, element t = lane % 16
  const int hq = lane >> 4, ht = lane & 15;
  const int hd = P.near[hq < NN ? hq : NN - 1];               // distance of this lane's near value halo
  const int hoff_x = (hq & 1) ? kRB + min(ht, dmax - 1) : -dmax + min(ht, dmax - 1);
  const int hslot = Z0 + (hq < NN ? hq : NN - 1);             // ... of slot z0 + q: rows r0 - d .. r0 - 1
  // r0 = first row of the wavefront at that step.  Lanes beyond the strip (or the run) still load real data -- their
  // elements of x are the near neighbours of the last active lanes -- with the row clamped into the matrix.
  auto load_step = [&](int64_t r0, WalkStep<NU, XL, FD>& w) __attribute__((always_inline)) {
    const int64_t r = min(r0 + lane, vmax);
    const VT* __restrict__ v = uvals + vpos(r);
#pragma unroll
    for (int u = 0; u < NU; ++u) w.ua[u] = ld_val<(NTM & 1) != 0>(v + (size_t)u * 64);
    // (the row of the gathered element is NOT clamped into the run: with diagonal far neighbours the last row of the run reads
    // the ring element of the lane beside it, which must be the true x[row + 1 + K g] -- only the matrix' last column clamps)
    w.xnew = ld_stream<(NTM & 2) != 0>(x + min(r0 + lane + K * g, rmax));
    w.v0 = ld_stream<(NTM & 2) != 0>(v0p ? v0p + r : x + lane);
    w.acc = ld_stream<(NTM & 2) != 0>(accp ? accp + r : x + lane);
    w.hx = x[min(r0 + hoff_x, rmax)];
    w.ha = ld_val<false>(uvals + vpos(r0 - hd + min(ht, hd - 1)) + (size_t)hslot * 64);
    if constexpr (FD != 0) {
      const int em = min(lane >> 2, K - 1) + 1, ew = lane & 3;               // x halos: lane e -> (m, which)
      const int64_t xrow = r0 + ((ew & 1) ? kRB : -1) + ((ew & 2) ? -(int64_t)em * g : (int64_t)em * g);
      w.hxf = x[min(max(xrow, (int64_t)0), rmax)];
      const int vm = min(lane >> 1, K - 1) + 1, vs = lane & 1;               // value halos: lane e -> (m, side)
      // side 0: row r0 - 1 - m g, its entry at distance m g + 1 (slot d = 2); side 1: row r0 + 64 - m g, distance m g - 1 (d = 0)
      const int64_t vrow = min(max(r0 + (vs ? kRB : -1) - (int64_t)vm * g, P.R0 * (int64_t)kRB), vmax);
      w.hvf = ld_val<false>(uvals + vpos(vrow) + (size_t)(Z0 + NN + (vm - 1) * FS + (vs ? 0 : 2)) * 64);
    }
#pragma unroll
    for (int p = 0; p < XL; ++p) {
      const int64_t Lp = (p == XL - 1) ? P.glong : P.glong1;
      w.xlu[p] = ld_stream<(NTM & 2) != 0>(x + min(r + Lp, rmax));
      w.xll[p] = ld_stream<(NTM & 2) != 0>(x + (r - Lp));
      w.al[p] = ld_val<false>(uvals + vpos(r - Lp) + (size_t)(Z0 + NN + KF + p) * 64);
    }
  };
  // where this lane's halo elements go in the windows (lanes that carry none rewrite their own main element)
  const bool hx_on = hq < 2 && ht < dmax;
  const int hx_pos = hx_on ? ((hq & 1) ? kWalkHalo + kRB + ht : kWalkHalo - dmax + ht) : kWalkHalo + lane;
  const bool ha_on = hq < NN && ht < hd;
  const int ha_pos = XW + (hq < NN ? hq : 0) * AW + (ha_on ? kWalkHalo - hd + ht : kWalkHalo + lane);

  int64_t row0 = P.W0 * (int64_t)kRB + (int64_t)j0 * g + (int64_t)col * kRB;
  const int64_t rend = P.R1 * (int64_t)kRB;
  const bool in_strip = (int64_t)col * kRB + lane < g;
  // the ring of gathered elements x[row + m g], m = -K .. K (the last one arrives with each step's loads) ...
  double2 xr[2 * K + 1];
#pragma unroll
  for (int m = -K; m < K; ++m) xr[K + m] = x[min(row0 + lane + m * g, rmax)];
  // ... and the far upper values of the K blocks behind, FIFO m in LDS: the value of t steps ago sits at entry
  // (step - t) mod m, so the entry read at a step (the value of m steps ago) is the one overwritten at that step
  // (FIFO of far slot (m, d), d < FS: entries FS m (m - 1) / 2 + d m ... + m - 1)
#pragma unroll
  for (int m = 1; m <= K; ++m)
#pragma unroll
    for (int d = 0; d < FS; ++d)
#pragma unroll
      for (int a = 1; a <= m; ++a)
        hring[(FS * (m * (m - 1) / 2) + d * m + (m - a)) * kRB + lane] =
            ld_val<false>(uvals + vpos(min(row0 + lane - (int64_t)a * g, vmax)) + (size_t)(Z0 + NN + (m - 1) * FS + d) * 64);   // (the HISTORY row is clamped into the run, not the lane's own: with diagonal neighbours the last row of the run reads the lane beside it)
  int hpos[K];   // (wave-uniform) entry of FIFO m that this step reads and then overwrites: step mod m
#pragma unroll
  for (int m = 1; m <= K; ++m) hpos[m - 1] = 0;
  // Two register sets that swap roles every step: while the arithmetic of a block runs out of one, the next
  // block's streams land in the other (no copies, and the wait for them sits at their first use, a whole step later).
  WalkStep<NU, XL, FD> wa, wb;
  load_step(row0, wa);
  auto step = [&](const WalkStep<NU, XL, FD>& cu, WalkStep<NU, XL, FD>& nx, auto has_next) __attribute__((always_inline)) {
    xr[2 * K] = cu.xnew;
    if constexpr (decltype(has_next)::value) load_step(row0 + g, nx);
    // ---- near windows through LDS: element e of the block's window sits at [kWalkHalo + e], e = -16 .. 79
    xwin[kWalkHalo + lane] = xr[K];
    xwin[hx_pos] = sel2(hx_on, cu.hx, xr[K]);
#pragma unroll
    for (int i = 0; i < NN; ++i) xwin[XW + i * AW + kWalkHalo + lane] = cu.ua[Z0 + i];
    {
      double2 own = cu.ua[Z0];
#pragma unroll
      for (int i = 1; i < NN; ++i) own = sel2(hq == i, cu.ua[Z0 + i], own);
      xwin[ha_pos] = sel2(ha_on, cu.ha, own);
    }
    // the lanes of this wavefront exchange data through its own window: order the writes before the reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every operand that comes out of LDS first (independent reads, one wait), then the arithmetic
    double2 fa[KF], na[NN], nxl[NN], nxu[NN];
#pragma unroll
    for (int m = 1; m <= K; ++m)
#pragma unroll
      for (int d = 0; d < FS; ++d) {
        // the value streamed m steps ago for the row that is this row's partner: the same lane (d = middle), or one lane over
        // for a diagonal partner (row - m g - delta, delta = d - 1: lane - delta; the edge lane is fixed up below)
        const int sl = FD ? min(max(lane - (d - 1), 0), kRB - 1) : lane;
        fa[(m - 1) * FS + d] = hring[(FS * (m * (m - 1) / 2) + d * m + hpos[m - 1]) * kRB + sl];
      }
    if constexpr (FD != 0) {
#pragma unroll
      for (int m = 1; m <= K; ++m) {
        const double2 v0h = bcast2(cu.hvf, 2 * (m - 1));       // row r0 - 1 - m g, distance m g + 1: partner of lane 0
        const double2 v1h = bcast2(cu.hvf, 2 * (m - 1) + 1);   // row r0 + 64 - m g, distance m g - 1: partner of lane 63
        fa[(m - 1) * FS + 2] = sel2(lane == 0, v0h, fa[(m - 1) * FS + 2]);
        fa[(m - 1) * FS + 0] = sel2(lane == kRB - 1, v1h, fa[(m - 1) * FS + 0]);
      }
    }
#pragma unroll
    for (int i = 0; i < NN; ++i) {
      const int d = P.near[i];
      na[i] = xwin[XW + i * AW + kWalkHalo + lane - d];
      nxl[i] = xwin[kWalkHalo + lane - d];
      nxu[i] = xwin[kWalkHalo + lane + d];
    }
    double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
    // lower section, storage order: (the long entries -L_1, -L_0,) far -K g .. -g, then near -d_NN .. -d_1
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      double2 a, xv;
      if (l < XL) {
        a = cu.al[XL - 1 - l];
        xv = cu.xll[XL - 1 - l];
      } else if (l < XL + KF) {
        // storage order: distance descending -- m = K .. 1 and inside a strip step m g + 1, m g, m g - 1 (d = 2, 1, 0)
        const int f = l - XL;
        const int m = K - f / FS;
        const int d = FS - 1 - f % FS;
        a = fa[(m - 1) * FS + d];
        if constexpr (FD != 0) {
          // column r - m g - delta, delta = d - 1: the ring element m steps back, one lane to the left (delta = +1) / right
          if (d == 2) xv = lane_shift2<false>(xr[K - m], bcast2(cu.hxf, 4 * (m - 1) + 2), lane);       // x[r0 - 1 - m g] for lane 0
          else if (d == 0) xv = lane_shift2<true>(xr[K - m], bcast2(cu.hxf, 4 * (m - 1) + 3), lane);   // x[r0 + 64 - m g] for lane 63
          else xv = xr[K - m];
        } else {
          xv = xr[K - m];
        }
      } else {
        const int i = NN - 1 - (l - XL - KF);
        a = na[i];
        xv = nxl[i];
      }
      if (l & 1) cfma_conj(s1, a, xv);
      else cfma_conj(s0, a, xv);
    }
    // upper section: the diagonal, near d_1 .. d_NN, far g .. K g
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      double2 xv;
      if (u < Z0) xv = xr[K];
      else if (u < Z0 + NN) xv = nxu[u - Z0];
      else if (u < Z0 + NN + KF) {
        const int f = u - Z0 - NN;
        const int m = f / FS + 1;
        if constexpr (FD != 0) {
          const int d = f % FS;   // distance m g + (d - 1)
          if (d == 0) xv = lane_shift2<false>(xr[K + m], bcast2(cu.hxf, 4 * (m - 1) + 0), lane);      // x[r0 - 1 + m g] for lane 0
          else if (d == 2) xv = lane_shift2<true>(xr[K + m], bcast2(cu.hxf, 4 * (m - 1) + 1), lane);   // x[r0 + 64 + m g] for lane 63
          else xv = xr[K + m];
        } else {
          xv = xr[K + m];
        }
      } else xv = cu.xlu[u - (Z0 + NN + KF) < 0 ? 0 : u - (Z0 + NN + KF)];
      if (u & 1) cfma(s1, cu.ua[u], xv);
      else cfma(s0, cu.ua[u], xv);
    }
    typename ChebyOpT<(NTM & 4) != 0>::Pre pre;
    pre.xi = xr[K];
    pre.v0 = v0p ? cu.v0 : make_double2(0.0, 0.0);
    pre.acc = accp ? cu.acc : make_double2(0.0, 0.0);
    double2 chk = make_double2(0.0, 0.0);
    double nrm = 0.0;
    const int64_t row = row0 + lane;
    if (in_strip && row < rend) opl.row(row, make_double2(s0.x + s1.x, s0.y + s1.y), pre, chk, nrm, 0);
    // ---- one step down the strip column
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this step's window / FIFO reads before the writes below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 1; m <= K; ++m) {
#pragma unroll
      for (int d = 0; d < FS; ++d) hring[(FS * (m * (m - 1) / 2) + d * m + hpos[m - 1]) * kRB + lane] = cu.ua[Z0 + NN + (m - 1) * FS + d];
      hpos[m - 1] = (hpos[m - 1] + 1 == m) ? 0 : hpos[m - 1] + 1;
    }
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) xr[i] = xr[i + 1];
    row0 += g;
  };
  int n = j1 - j0;
  for (; n > 2; n -= 2) {
    step(wa, wb, std::true_type());
    step(wb, wa, std::true_type());
  }
  if (n == 2) {
    step(wa, wb, std::true_type());
    step(wb, wa, std::false_type());
  } else {
    step(wa, wb, std::false_type());
  }
  }
  if (G.edge_last && G.n_edge_wg == 0 && task < P.n_edge) {
    wave_sync_wait(sy);
    for (int64_t idx = task; idx < P.n_edge; idx += G.ntask)
      hrb_edge_block<VT>(H, uvals, x, (int64_t)__builtin_amdgcn_readfirstlane(P.edge_map[idx]), lane, nrows, op);
  }
}

template <class VT, int NN, int K, int Z0, int NTM = 0, int XL = 0, int FD = 0>
static bool launch_instance(hipStream_t s, dim3 grid, const VT* uvals, const double2* x, const WalkPlan& P,
                            const WalkGeom& G, const HrbArrays& H, int64_t nrows, const ChebyOp& op, const SyncArgs& sy) {
  const int ws = G.ntask / std::max(G.n_walk_wg, 1);      // wavefronts per workgroup of this launch
  const size_t lds = WalkLds<NN, K, FD>::kBytesPerWave * (size_t)ws;
  constexpr size_t lds_max = WalkLds<NN, K, FD>::kBytesPerWave * kWalkWaves;
  auto kern = &hrb_walk_kernel<VT, NN, K, Z0, NTM, XL, FD>;
  // more than the 64 KB a launch gets without asking: opt in once per kernel instance AND device (a process may hold
  // contexts on several GPUs); 0 = not tried, 1 = granted, 2 = refused (the caller then takes the per-block kernel)
  static std::atomic<unsigned char> opted[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  unsigned char st = opted[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    opted[dev].store(st, std::memory_order_release);
  }
  if (st != 1) return false;
  hipLaunchKernelGGL(kern, grid, dim3(64 * ws), lds, s, uvals, x, P, G, H, nrows, op, sy);
  return true;
}

// The kernel shapes are spread over translation units: PART 0 = near 3-4, PART 1 = near 1-2 and the shapes with one long pair and
// one far distance, PART 2 = the other long-pair shapes (two far distances, two long pairs), PART 3 = the shapes with diagonal far neighbours, each
// for complex values (double2) and for the real copy (double).  The measurement variants of the headline shape (matrix loads
// temporal / nontemporal in other mixes: ntm 3, 5, 7) exist in developer builds only (-DQP_DEVELOPER).
template <class VT, int PART>
static bool launch_shape(hipStream_t s, dim3 grid, const VT* uvals, const double2* x, const WalkPlan& P,
                         const WalkGeom& G, const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  const int key = P.nn * 100 + P.K * 10 + P.z0;
  if (P.fd) {   // diagonal far neighbours (nine-point stencils): near 1 or 2, one strip step, with or without ONE long pair
                // (layers of such planes: +-nx ny) -- PART 3
    if constexpr (PART == 3) {
#define QP_WALK_FD(NN_, Z0_, XL_)                                                                                       \
  return (ntm & 1) ? launch_instance<VT, NN_, 1, Z0_, 1, XL_, 1>(s, grid, uvals, x, P, G, H, nrows, op, sy)              \
                   : launch_instance<VT, NN_, 1, Z0_, 0, XL_, 1>(s, grid, uvals, x, P, G, H, nrows, op, sy);
      if (P.xl > 1 || P.K != 1) return false;
      switch (P.xl * 1000 + key) {
        case 110: QP_WALK_FD(1, 0, 0)
        case 111: QP_WALK_FD(1, 1, 0)
        case 210: QP_WALK_FD(2, 0, 0)
        case 211: QP_WALK_FD(2, 1, 0)
        case 1110: QP_WALK_FD(1, 0, 1)
        case 1111: QP_WALK_FD(1, 1, 1)
        case 1210: QP_WALK_FD(2, 0, 1)
        case 1211: QP_WALK_FD(2, 1, 1)
        default: return false;
      }
#undef QP_WALK_FD
    }
    return false;
  }
  if constexpr (PART == 3) return false;
  if (P.xl) {   // long pairs: near 1 or 2, far 1 or 2, one or two pairs (three-dimensional grids; PART 1: one pair and one far distance, PART 2: the rest)
#define QP_WALK_XL(NN_, K_, Z0_, XL_)                                                                              \
  return (ntm & 1) ? launch_instance<VT, NN_, K_, Z0_, 1, XL_>(s, grid, uvals, x, P, G, H, nrows, op, sy)           \
                   : launch_instance<VT, NN_, K_, Z0_, 0, XL_>(s, grid, uvals, x, P, G, H, nrows, op, sy);
    if constexpr (PART == 1) {
      if (P.xl != 1) return false;
      switch (key) {
        case 110: QP_WALK_XL(1, 1, 0, 1)
        case 111: QP_WALK_XL(1, 1, 1, 1)
        case 210: QP_WALK_XL(2, 1, 0, 1)
        case 211: QP_WALK_XL(2, 1, 1, 1)
        default: return false;
      }
    }
    if constexpr (PART == 2) {
      switch (P.xl * 1000 + key) {
        case 1120: QP_WALK_XL(1, 2, 0, 1)
        case 1121: QP_WALK_XL(1, 2, 1, 1)
        case 1220: QP_WALK_XL(2, 2, 0, 1)
        case 1221: QP_WALK_XL(2, 2, 1, 1)
        case 2110: QP_WALK_XL(1, 1, 0, 2)
        case 2111: QP_WALK_XL(1, 1, 1, 2)
        case 2210: QP_WALK_XL(2, 1, 0, 2)
        case 2211: QP_WALK_XL(2, 1, 1, 2)
        case 2120: QP_WALK_XL(1, 2, 0, 2)
        case 2121: QP_WALK_XL(1, 2, 1, 2)
        case 2220: QP_WALK_XL(2, 2, 0, 2)
        case 2221: QP_WALK_XL(2, 2, 1, 2)
        default: return false;
      }
    }
#undef QP_WALK_XL
    return false;
  }
  if constexpr (PART == 2) return false;
#define QP_WALK_SHAPE(NN_, K_, Z0_)                                                                       \
  return (ntm & 1) ? launch_instance<VT, NN_, K_, Z0_, 1>(s, grid, uvals, x, P, G, H, nrows, op, sy)            \
                   : launch_instance<VT, NN_, K_, Z0_, 0>(s, grid, uvals, x, P, G, H, nrows, op, sy);
#ifdef QP_DEVELOPER
  if constexpr (PART == 0) {
    if (key == 440 && (ntm == 3 || ntm == 5 || ntm == 7)) {   // (measurement variants of the headline shape)
      if (ntm == 3) return launch_instance<VT, 4, 4, 0, 3>(s, grid, uvals, x, P, G, H, nrows, op, sy);
      if (ntm == 5) return launch_instance<VT, 4, 4, 0, 5>(s, grid, uvals, x, P, G, H, nrows, op, sy);
      return launch_instance<VT, 4, 4, 0, 7>(s, grid, uvals, x, P, G, H, nrows, op, sy);
    }
  }
#endif
#define QP_WALK_NN(NN_)                                    \
  case NN_ * 100 + 10: QP_WALK_SHAPE(NN_, 1, 0)            \
  case NN_ * 100 + 11: QP_WALK_SHAPE(NN_, 1, 1)            \
  case NN_ * 100 + 20: QP_WALK_SHAPE(NN_, 2, 0)            \
  case NN_ * 100 + 21: QP_WALK_SHAPE(NN_, 2, 1)            \
  case NN_ * 100 + 30: QP_WALK_SHAPE(NN_, 3, 0)            \
  case NN_ * 100 + 31: QP_WALK_SHAPE(NN_, 3, 1)            \
  case NN_ * 100 + 40: QP_WALK_SHAPE(NN_, 4, 0)            \
  case NN_ * 100 + 41: QP_WALK_SHAPE(NN_, 4, 1)
  if constexpr (PART == 0) {
    switch (key) {
      QP_WALK_NN(3)
      QP_WALK_NN(4)
      default: return false;
    }
  } else if constexpr (PART == 1) {
    switch (key) {
      QP_WALK_NN(1)
      QP_WALK_NN(2)
      default: return false;
    }
  }
  return false;
#undef QP_WALK_NN
#undef QP_WALK_SHAPE
}

// the eight translation units' entry points (uvals: double2* for _c128_*, double* for _f64_*)
bool walk_launch_c128_hi(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_c128_lo(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_f64_hi(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                        const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_f64_lo(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                        const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_c128_xl(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_f64_xl(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                        const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_c128_fd(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);
bool walk_launch_f64_fd(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                        const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy);

}  // namespace qp

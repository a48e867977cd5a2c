// (kernels_walk2_impl.h: the kernel template and its instantiating launchers, included by the translation units kernels_walk2.hip / _b / _c / _d --
// one per number of near distances -- so that `make -j` compiles the 128 kernel instances in parallel)
//
// Two-term strip walk: TWO fused Chebyshev terms per pass over the matrix values (temporal blocking of the three-term
// recurrence, src/cheby.jl:186-209), gfx950 / wave64.  For Hermitian-packed LATTICE operators beyond the Infinity Cache, where the
// one-term walk (kernels_walk_impl.h) is bound by the requests it keeps in flight: 128 of the 186 bytes it moves per row and
// term are matrix values, read once per term.  profiles/r06/two_term_probe.txt measured the ceiling of this form before it
// was built (bare loop, 1024 wavefronts: N = 2^22 104 us per term against 121, 2^24 406 - 411 against 539 - 585).
//
// A launch takes x = v_{m-1} and p = v_{m-2} and forms   y = v_m     = c (H x - beta x) + p     (term m,     epilogue op1)
//                                                          z = v_{m+1} = c (H y - beta y) + x     (term m + 1, epilogue op2)
// A wavefront walks down a strip column as in the one-term walk.  At step t it forms y for block t (phase Y: the one-term
// walk's step, unchanged: 8 value lines + 3 vector lines + 2 packed halo loads, one step ahead) and z for block t - K (phase
// Z), whose row sums need
//   * y of the blocks t - 2K .. t of its own column  -> a second register ring (the newest element is the y just formed);
//   * the matrix values of block t - K a second time: its diagonal / near values from a register history of K + 1 blocks, its far
//     values -- upper (streamed K steps ago) and conj-transposed lower (streamed K + m steps ago) -- from the far-value FIFOs in
//     LDS, now K + m entries deep for far slot m instead of m;
//   * y of the rows just outside its 64: they belong to the neighbouring strip columns, which run on their own -- so a chunk
//     forms y on all 64 lanes but z (and the stored y) only on the W = 64 - 2 d_max lanes in the middle; the chunks overlap by
//     2 d_max rows (W = 56 for the headline lattice: 19 chunks per 1024-row strip step instead of 16).
// A segment of L steps runs in 2 K steps before its first z (K of y alone, then K more until y is K blocks ahead).  One wavefront
// per SIMD (the rings and the history need ~370 registers: 256 VGPRs + 116 AGPRs at (4, 4); LDS: 33 KB per wavefront).  y and z go to two fresh vectors: writing in
// place would race with the neighbouring columns' halo reads of x and p.
//
// Blocks outside the region where BOTH phases have their operands inside the lattice run -- the one-term walk's edge blocks and K strip
// steps at either end of the run: the plan's edge list -- get term m from the per-block path inside this launch (before the walk,
// they only read x and p) and term m + 1 from a per-block launch over the same list afterwards (engine_cheby.hip).
//
// Both row sums are the one-term walk's (lower slots then upper slots in storage order, two interleaved partial sums) and both
// epilogues are ChebyOp's: bit-identical to one-term launches (tests/test_gpu_walk2.py).
#pragma once
#include <atomic>

#include "kernels_walk_impl.h"

namespace qp {

constexpr int kWalk2Waves = 4;   // wavefronts per workgroup = per compute unit: one per SIMD

template <int NN, int K>
struct Walk2Lds {
  static constexpr int XW = kRB + 2 * kWalkHalo, AW = kRB + kWalkHalo;
  static constexpr int kWin = XW + NN * AW;                    // the near windows (phase Y, then phase Z)
  static constexpr int kFifo = K * K + K * (K + 1) / 2;        // far slot m: K + m entries of 64
  static constexpr int kPerWave = kWin + kRB * kFifo;
  static constexpr size_t kBytesPerWave = sizeof(double2) * (size_t)kPerWave;
};

struct Walk2Geom {
  int L = 0, nseg = 0, ntask = 0, n_walk_wg = 0;
  int S2 = 0;          // column chunks per strip step: ceil(g / W)
  int W = 0;           // rows of a chunk that form z: 64 - 2 d_max
  int64_t xlast = 0;   // last element of x
  int64_t vend = 0;    // first row beyond the lattice run (values at the run's strides exist below it)
};

template <int NU>
struct Walk2Step {
  double2 ua[NU];
  double2 xnew;        // x[row + K g]
  double2 p, acc1;     // row-local operands of term m (block t)
  double2 acc2;        // accumulator of term m + 1 (block t - K)
  double2 hx, ha;      // packed near halos of phase Y (see WalkStep)
};

template <class VT, int NN, int K, int Z0, int NTM>
__global__ __launch_bounds__(64 * kWalk2Waves) void hrb_walk2_kernel(const VT* __restrict__ uvals, const double2* __restrict__ x,
                                                                      WalkPlan P, Walk2Geom G, HrbArrays H, int64_t nrows,
                                                                      ChebyOp op1, ChebyOp op2) {
  constexpr int NL = NN + K;
  constexpr int NU = Z0 + NN + K;
  constexpr int NH = Z0 + NN;             // diagonal + near values per row: the register history
  using Lds = Walk2Lds<NN, K>;
  constexpr int XW = Lds::XW, AW = Lds::AW;
  static_assert(NN <= 4, "the near value halos share one register: sixteen lanes each");
  extern __shared__ double2 walk2_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, G.n_walk_wg);
  const int task = (int)wg * kWalk2Waves + wave;
  // term m of the blocks outside the two-term region: per-block path, before the walk
  for (int64_t idx = task; idx < P.n_edge; idx += G.ntask)
    hrb_edge_block<VT>(H, uvals, x, (int64_t)__builtin_amdgcn_readfirstlane(P.edge_map[idx]), lane, nrows, op1);
  const int seg = task / G.S2, col = task - seg * G.S2;
  if (seg >= G.nseg) return;
  const int64_t g = P.g;
  const int64_t zb = P.W0 * (int64_t)kRB, ze = P.R1 * (int64_t)kRB;      // rows of the two-term region
  const int Jz = (int)((ze - zb + g - 1) / g);
  const int j0 = seg * G.L, j1 = min(j0 + G.L, Jz);
  if (j0 >= j1) return;
  const int dmax = P.near[NN - 1];
  const int64_t coff = (int64_t)col * G.W - dmax;      // first lane's offset inside the strip step
  const bool owner = lane >= dmax && lane < kRB - dmax && coff + lane < g;
  double2* __restrict__ xwin = walk2_lds + (size_t)wave * Lds::kPerWave;
  double2* __restrict__ fifo = xwin + Lds::kWin;
  const int64_t rlo = P.R0 * (int64_t)kRB, vmax = G.vend - 1, rmax = G.xlast, nlast = nrows - 1;
  auto vpos = [&](int64_t r) __attribute__((always_inline)) -> int64_t {
    return P.U0 + ((r >> 6) - P.R0) * (int64_t)P.ustride + (r & 63);
  };
  auto clampv = [&](int64_t r) __attribute__((always_inline)) -> int64_t { return min(max(r, rlo), vmax); };
  ChebyOp opl1{op1.e}, opl2{op2.e};
  opl1.e.mirror = opl2.e.mirror = nullptr;
  opl1.e.check_partials = opl2.e.check_partials = nullptr;
  const double2* __restrict__ pp = op1.e.v0;            // v_{m-2}: always there (the launcher takes this kernel for m >= 2 only)
  const double2* __restrict__ acc1p = op1.e.acc_in;
  const double2* __restrict__ acc2p = op2.e.acc_in;
  const int hq = lane >> 4, ht = lane & 15;
  const int hd = P.near[hq < NN ? hq : NN - 1];
  const int hoff_x = (hq & 1) ? kRB + min(ht, dmax - 1) : -dmax + min(ht, dmax - 1);
  const int hslot = Z0 + (hq < NN ? hq : NN - 1);
  // first row of the chunk's lanes at Y step t (t may be negative: the run-in reaches K steps below the region)
  auto row_of = [&](int t) __attribute__((always_inline)) -> int64_t { return zb + (int64_t)t * g + coff; };
  auto load_step = [&](int t, Walk2Step<NU>& w) __attribute__((always_inline)) {
    const int64_t r0 = row_of(t);
    const int64_t r = clampv(r0 + lane);
    const VT* __restrict__ v = uvals + vpos(r);
#pragma unroll
    for (int u = 0; u < NU; ++u) w.ua[u] = ld_val<(NTM & 1) != 0>(v + (size_t)u * 64);
    w.xnew = x[min(max(r0 + lane + K * g, (int64_t)0), rmax)];
    const int64_t rc = min(max(r0 + lane, (int64_t)0), nlast);
    w.p = pp[rc];
    w.acc1 = acc1p ? acc1p[rc] : x[lane];
    w.acc2 = acc2p ? acc2p[min(max(r0 + lane - K * g, (int64_t)0), nlast)] : x[lane];
    w.hx = x[min(max(r0 + hoff_x, (int64_t)0), rmax)];
    w.ha = ld_val<false>(uvals + vpos(clampv(r0 - hd + min(ht, hd - 1))) + (size_t)hslot * 64);
  };
  const bool hx_on = hq < 2 && ht < dmax;
  const int hx_pos = hx_on ? ((hq & 1) ? kWalkHalo + kRB + ht : kWalkHalo - dmax + ht) : kWalkHalo + lane;
  const bool ha_on = hq < NN && ht < hd;
  const int ha_pos = XW + (hq < NN ? hq : 0) * AW + (ha_on ? kWalkHalo - hd + ht : kWalkHalo + lane);

  const int t0 = j0 - K;                                // first Y step
  const int nsteps = (j1 - j0) + 2 * K;                 // Y steps t0 .. t0 + nsteps - 1; z of block t - K from step 2 K on
  // rings: x[block t + i] at xr[K + i]; y[block t - K + i] at yr[K + i] (the newest, yr[2 K], is this step's y)
  double2 xr[2 * K + 1], yr[2 * K + 1];
  {
    const int64_t r0 = row_of(t0);
#pragma unroll
    for (int m = -K; m < K; ++m) xr[K + m] = x[min(max(r0 + lane + m * g, (int64_t)0), rmax)];
  }
#pragma unroll
  for (int i = 0; i <= 2 * K; ++i) yr[i] = make_double2(0.0, 0.0);
  double2 nh[K + 1][NH];                                // diagonal / near values of blocks t - K .. t
#pragma unroll
  for (int a = 0; a <= K; ++a)
#pragma unroll
    for (int u = 0; u < NH; ++u) nh[a][u] = make_double2(0.0, 0.0);
  // far-value FIFOs: slot m holds the far upper value m of the last K + m blocks; the block of step tau (counted from the
  // segment's first step) sits at entry tau mod (K + m).  History of the blocks before the first step: ages 1 .. m (phase Y of the
  // first steps reads them; phase Z reads ages up to K + m only from step 2 K on, when every such block is the segment's own)
  auto foff = [](int m) __attribute__((always_inline)) -> int { return (m - 1) * K + (m - 1) * m / 2; };
#pragma unroll
  for (int m = 1; m <= K; ++m)
#pragma unroll
    for (int a = 1; a <= m; ++a)
      fifo[(foff(m) + (K + m - a)) * kRB + lane] =
          ld_val<false>(uvals + vpos(clampv(row_of(t0) + lane - (int64_t)a * g)) + (size_t)(Z0 + NN + (m - 1)) * 64);
  int hp[K];   // (wave-uniform) entry of FIFO m that this step's block goes to
#pragma unroll
  for (int m = 1; m <= K; ++m) hp[m - 1] = 0;
  auto fpos = [&](int m, int age) __attribute__((always_inline)) -> int {      // entry of the block `age` steps back
    const int q = hp[m - 1] - age;
    return foff(m) + (q < 0 ? q + K + m : q);
  };

  Walk2Step<NU> wa, wb;
  load_step(t0, wa);
  int tau = 0;
  auto step = [&](const Walk2Step<NU>& cu, Walk2Step<NU>& nx, auto has_next) __attribute__((always_inline)) {
    const int t = t0 + tau;
    xr[2 * K] = cu.xnew;
    if constexpr (decltype(has_next)::value) load_step(t + 1, nx);
#pragma unroll
    for (int u = 0; u < NH; ++u) nh[K][u] = cu.ua[u];
    // ---- phase Y: near windows of x and of the block's near values through LDS (the one-term walk's step)
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double2 fy[K], fzl[K], fzu[K], na[NN], nxl[NN], nxu[NN];
#pragma unroll
    for (int m = 1; m <= K; ++m) {
      fy[m - 1] = fifo[fpos(m, m) * kRB + lane];          // block t - m           (lower far entry of block t)
      fzu[m - 1] = fifo[fpos(m, K) * kRB + lane];         // block t - K           (upper far entry of block t - K)
      fzl[m - 1] = fifo[fpos(m, K + m) * kRB + lane];     // block t - K - m       (lower far entry of block t - K)
    }
#pragma unroll
    for (int i = 0; i < NN; ++i) {
      const int d = P.near[i];
      na[i] = xwin[XW + i * AW + kWalkHalo + lane - d];
      nxl[i] = xwin[kWalkHalo + lane - d];
      nxu[i] = xwin[kWalkHalo + lane + d];
    }
    double2 yv;
    {
      double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        double2 a, xv;
        if (l < K) {
          const int m = K - l;
          a = fy[m - 1];
          xv = xr[K - m];
        } else {
          const int i = NN - 1 - (l - K);
          a = na[i];
          xv = nxl[i];
        }
        if (l & 1) cfma_conj(s1, a, xv);
        else cfma_conj(s0, a, xv);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double2 xv;
        if (u < Z0) xv = xr[K];
        else if (u < Z0 + NN) xv = nxu[u - Z0];
        else xv = xr[K + (u - Z0 - NN + 1)];
        if (u & 1) cfma(s1, cu.ua[u], xv);
        else cfma(s0, cu.ua[u], xv);
      }
      ChebyOp::Pre pre;
      pre.xi = xr[K];
      pre.v0 = cu.p;
      pre.acc = acc1p ? cu.acc1 : make_double2(0.0, 0.0);
      yv = opl1.term(make_double2(s0.x + s1.x, s0.y + s1.y), pre);
      const int64_t row = row_of(t) + lane;
      if (owner && t >= j0 && t < j1 && row < ze) opl1.finish(row, yv, pre);
    }
    yr[2 * K] = yv;
    // ---- this step's far values into the FIFOs, the windows of phase Z (block t - K: its y and its near values; no halos -- the
    //      lanes that form z have their near neighbours inside the chunk)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // phase Y's window / FIFO reads before the writes below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 1; m <= K; ++m) {
      fifo[(foff(m) + hp[m - 1]) * kRB + lane] = cu.ua[Z0 + NN + (m - 1)];
      hp[m - 1] = (hp[m - 1] + 1 == K + m) ? 0 : hp[m - 1] + 1;
    }
    xwin[kWalkHalo + lane] = yr[K];
#pragma unroll
    for (int i = 0; i < NN; ++i) xwin[XW + i * AW + kWalkHalo + lane] = nh[0][Z0 + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      double2 za[NN], zyl[NN], zyu[NN];
#pragma unroll
      for (int i = 0; i < NN; ++i) {
        const int d = P.near[i];
        za[i] = xwin[XW + i * AW + kWalkHalo + lane - d];
        zyl[i] = xwin[kWalkHalo + lane - d];
        zyu[i] = xwin[kWalkHalo + lane + d];
      }
      double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        double2 a, xv;
        if (l < K) {
          const int m = K - l;
          a = fzl[m - 1];
          xv = yr[K - m];
        } else {
          const int i = NN - 1 - (l - K);
          a = za[i];
          xv = zyl[i];
        }
        if (l & 1) cfma_conj(s1, a, xv);
        else cfma_conj(s0, a, xv);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double2 a, xv;
        if (u < Z0) {
          a = nh[0][u];
          xv = yr[K];
        } else if (u < Z0 + NN) {
          a = nh[0][u];
          xv = zyu[u - Z0];
        } else {
          const int m = u - Z0 - NN + 1;
          a = fzu[m - 1];
          xv = yr[K + m];
        }
        if (u & 1) cfma(s1, a, xv);
        else cfma(s0, a, xv);
      }
      ChebyOp::Pre pre;
      pre.xi = yr[K];
      pre.v0 = xr[0];                                   // x of block t - K: v_{m-1}, the "v0" of term m + 1
      pre.acc = acc2p ? cu.acc2 : make_double2(0.0, 0.0);
      const double2 zv = opl2.term(make_double2(s0.x + s1.x, s0.y + s1.y), pre);
      const int c = t - K;
      const int64_t row = row_of(c) + lane;
      if (owner && c >= j0 && c < j1 && row < ze) opl2.finish(row, zv, pre);
    }
    // ---- one step down the strip column
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // phase Z's window reads before the next step's window writes
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) {
      xr[i] = xr[i + 1];
      yr[i] = yr[i + 1];
    }
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
      for (int u = 0; u < NH; ++u) nh[a][u] = nh[a + 1][u];
    ++tau;
  };
  int n = nsteps;
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

template <class VT, int NN, int K, int Z0, int NTM>
static bool launch2_instance(hipStream_t s, const VT* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G,
                             const HrbArrays& H, int64_t nrows, const ChebyOp& op1, const ChebyOp& op2) {
  constexpr size_t lds = Walk2Lds<NN, K>::kBytesPerWave * kWalk2Waves;
  static_assert(lds <= 160 * 1024, "four wavefronts' windows and FIFOs fit the compute unit's LDS");
  auto kern = &hrb_walk2_kernel<VT, NN, K, Z0, NTM>;
  static std::atomic<unsigned char> opted[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  unsigned char st = opted[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    opted[dev].store(st, std::memory_order_release);
  }
  if (st != 1) return false;
  hipLaunchKernelGGL(kern, dim3((unsigned)G.n_walk_wg), dim3(64 * kWalk2Waves), lds, s, uvals, x, P, G, H, nrows, op1, op2);
  return true;
}

// every far reach K = 1 .. 4, with and without a diagonal, for NN near distances; both cache policies of the value stream
template <class VT, int NN>
static bool launch2_shape(hipStream_t s, const VT* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G,
                          const HrbArrays& H, int64_t nrows, const ChebyOp& op1, const ChebyOp& op2, int ntm) {
  if (P.nn != NN) return false;
#define QP_WALK2(K_, Z0_)                                                                              \
  return (ntm & 1) ? launch2_instance<VT, NN, K_, Z0_, 1>(s, uvals, x, P, G, H, nrows, op1, op2)         \
                   : launch2_instance<VT, NN, K_, Z0_, 0>(s, uvals, x, P, G, H, nrows, op1, op2);
  switch (P.K * 10 + P.z0) {
    case 10: QP_WALK2(1, 0)
    case 11: QP_WALK2(1, 1)
    case 20: QP_WALK2(2, 0)
    case 21: QP_WALK2(2, 1)
    case 30: QP_WALK2(3, 0)
    case 31: QP_WALK2(3, 1)
    case 40: QP_WALK2(4, 0)
    case 41: QP_WALK2(4, 1)
    default: return false;
  }
#undef QP_WALK2
}

// the translation units' entry points (uvals: double2*, or double* for the real copy)
#define QP_WALK2_DECL(NAME)                                                                                                   \
  bool NAME##_c128(hipStream_t s, const double2* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G, const HrbArrays& H, \
                   int64_t nrows, const ChebyOp& op1, const ChebyOp& op2, int ntm);                                                  \
  bool NAME##_f64(hipStream_t s, const double* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G, const HrbArrays& H,    \
                  int64_t nrows, const ChebyOp& op1, const ChebyOp& op2, int ntm);
QP_WALK2_DECL(walk2_launch_nn1)
QP_WALK2_DECL(walk2_launch_nn2)
QP_WALK2_DECL(walk2_launch_nn3)
QP_WALK2_DECL(walk2_launch_nn4)
#undef QP_WALK2_DECL
#define QP_WALK2_DEFINE(NAME, NN_)                                                                                            \
  bool NAME##_c128(hipStream_t s, const double2* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G, const HrbArrays& H, \
                   int64_t nrows, const ChebyOp& op1, const ChebyOp& op2, int ntm) {                                                 \
    return launch2_shape<double2, NN_>(s, uvals, x, P, G, H, nrows, op1, op2, ntm);                                                  \
  }                                                                                                                                   \
  bool NAME##_f64(hipStream_t s, const double* uvals, const double2* x, const WalkPlan& P, const Walk2Geom& G, const HrbArrays& H,    \
                  int64_t nrows, const ChebyOp& op1, const ChebyOp& op2, int ntm) {                                                   \
    return launch2_shape<double, NN_>(s, uvals, x, P, G, H, nrows, op1, op2, ntm);                                                    \
  }

}  // namespace qp

// Probe (round 6): the batched-states Chebyshev term of BASELINE configs[4] (N = 2^18 rows, 64 states as a panel
// X[row * 64 + state], lattice operator: near +-1..4, far +-g..4g with g = 1024) spends 18-21 wave-wide 1-KiB requests per matrix row
// in today's kernel (one wavefront per row, lane = state: 16 gathers, x_i, v_{t-1}, the store, the accumulator every third term).
// What does it cost when a wavefront owns a TILE of RA x RC rows (RA along the far direction i = a g + c, RC consecutive c) and keeps
// the tile, its near halo (RA x 8 rows) and its far halo (8 x RC rows) in registers -- (RA RC + 8 RA + 8 RC) / (RA RC) loads per row
// instead of 17, the same sums in the same order?  Bare kernels, synthetic values, boundaries clamped (both variants the same way, so
// their outputs can be compared bit for bit).
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/panel_tile_probe.hip -o tools/probe/panel_tile_probe
//   panel_tile_probe [log2n = 18] [strip width = 64] [waves per workgroup = 4]     -> one JSON line per variant
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int NS = 64;      // states = lanes
constexpr int NE = 16;      // entries per row: -4g..-g, -4..-1, 1..4, g..4g (sorted by column)

struct Args {
  const d2* vals;           // [row][16]
  const d2* x;
  const d2* p;
  const d2* acc;
  d2* y;
  d2* acc_out;
  long long n;
  int g, sw, with_acc, wpw;
  int gmode;                // rows_perm_kernel: 2 = the gathers as they are, 1 = all sixteen from the row's own line, 0 = none (x_i only)
  int ablate;               // lds_tile_kernel: 1 = no staging loads (LDS read as it is), 2 = staged but every operand read is the row's own slot, 3 = both
  int persist;              // rows_perm_kernel: > 0 = that many wavefronts in all, each walking positions wave, wave + persist, ...
};

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, j = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + j;
}
__device__ __forceinline__ void cfma(d2& a, d2 v, d2 x) {
  a.x = fma(v.x, x.x, a.x);
  a.x = fma(-v.y, x.y, a.x);
  a.y = fma(v.x, x.y, a.y);
  a.y = fma(v.y, x.x, a.y);
}
__host__ __device__ constexpr int cl(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ long long clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void finish2(const Args& A, long long row, int lane, d2 s0, d2 s1, d2 xi, d2 v0, d2 r) {
  const long long e = row * NS + lane;
  d2 t;
  t.x = 2.0 * (s0.x + s1.x) - 0.1 * xi.x + v0.x;
  t.y = 2.0 * (s0.y + s1.y) - 0.1 * xi.y + v0.y;
  __builtin_nontemporal_store(t, A.y + e);
  if (A.with_acc) {
    r.x = fma(0.3, t.x, r.x);
    r.y = fma(0.3, t.y, r.y);
    __builtin_nontemporal_store(r, A.acc_out + e);
  }
}
__device__ __forceinline__ void finish(const Args& A, long long row, int lane, d2 s0, d2 s1, d2 xi) {
  const long long e = row * NS + lane;
  const d2 v0 = __builtin_nontemporal_load(A.p + e);
  d2 t;
  t.x = 2.0 * (s0.x + s1.x) - 0.1 * xi.x + v0.x;
  t.y = 2.0 * (s0.y + s1.y) - 0.1 * xi.y + v0.y;
  __builtin_nontemporal_store(t, A.y + e);
  if (A.with_acc) {
    d2 r = __builtin_nontemporal_load(A.acc + e);
    r.x = fma(0.3, t.x, r.x);
    r.y = fma(0.3, t.y, r.y);
    __builtin_nontemporal_store(r, A.acc_out + e);
  }
}

// today: one wavefront per row; walk order = strips of sw consecutive c, all a in turn, c fastest
// VM: how the wave-uniform matrix entries reach the FMAs -- 0: scalar loads (today: s_load through the scalar cache), 1: one entry per
// lane in one coalesced vector load + v_readlane broadcasts, 2: vector loads of a uniform address (a broadcast in the texture path)
// GM: 2 = the gathers as they are, 0 = none (x_i stands in for all sixteen)
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int VM, int GM>
__global__ __launch_bounds__(512) void rows_kernel(Args A) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long pos = (long long)wg * A.wpw + wave;
  if (pos >= A.n) return;
  const long long na = A.n / A.g;
  const long long strip = pos / (na * A.sw), r = pos - strip * na * A.sw;
  const long long a = r / A.sw, c = strip * A.sw + (r - a * A.sw);
  const long long row = a * A.g + c;
  const d2* __restrict__ rv = A.vals + row * NE;
  const d2* __restrict__ Xs = A.x + lane;
  d2 mv = {0.0, 0.0};
  if (VM == 1) mv = rv[lane & 15];
  d2 x[NE];
  const d2 xi = Xs[row * NS];
  if (GM == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[k] = Xs[(clampll(a - 4 + k, 0, na - 1) * A.g + c) * NS];
      x[12 + k] = Xs[(clampll(a + 1 + k, 0, na - 1) * A.g + c) * NS];
      x[4 + k] = Xs[(a * A.g + clampll(c - 4 + k, 0, A.g - 1)) * NS];
      x[8 + k] = Xs[(a * A.g + clampll(c + 1 + k, 0, A.g - 1)) * NS];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NE; ++k) x[k] = xi;
  }
  d2 v[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    if (VM == 0) v[k] = rv[k];
    else if (VM == 1) v[k] = d2{readlane_f64(mv.x, k), readlane_f64(mv.y, k)};
    else v[k] = *(const d2*)((const char*)(rv + k) + (lane >> 6));      // lane >> 6 == 0: a vector address the compiler cannot prove uniform
  }
  d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NE; k += 2) {
    cfma(s0, v[k], x[k]);
    cfma(s1, v[k + 1], x[k + 1]);
  }
  finish(A, row, lane, s0, s1, xi);
}

// one wavefront per row as today, but the workgroup's wavefronts form a WA x WC patch of the lattice (today: 1 x 8): the patch's
// operands overlap (WA WC + 8 WA + 8 WC distinct rows for 17 WA WC requests) and the requests are issued within one memory latency of
// each other, so the vector L1 can merge them
template <int WA, int WC>
__global__ __launch_bounds__(64 * WA * WC) void rows_patch_kernel(Args A) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long na = A.n / A.g, nta = na / WA, ntc = A.sw / WC;
  const long long strip = wg / (nta * ntc), r = wg - strip * nta * ntc;
  const long long ta = r / ntc, tc = r - ta * ntc;
  const long long a = ta * WA + wave / WC, c = strip * A.sw + tc * WC + wave % WC;
  const long long row = a * A.g + c;
  const d2* __restrict__ rv = A.vals + row * NE;
  const d2* __restrict__ Xs = A.x + lane;
  d2 x[NE];
  const d2 xi = Xs[row * NS];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = Xs[(clampll(a - 4 + k, 0, na - 1) * A.g + c) * NS];
    x[12 + k] = Xs[(clampll(a + 1 + k, 0, na - 1) * A.g + c) * NS];
    x[4 + k] = Xs[(a * A.g + clampll(c - 4 + k, 0, A.g - 1)) * NS];
    x[8 + k] = Xs[(a * A.g + clampll(c + 1 + k, 0, A.g - 1)) * NS];
  }
  d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NE; k += 2) {
    cfma(s0, rv[k], x[k]);
    cfma(s1, rv[k + 1], x[k + 1]);
  }
  finish(A, row, lane, s0, s1, xi);
}

// the panel stored in the ORDER OF THE WALK (physical row = walk position): every stream is contiguous, the far neighbours are
// +-k sw positions away, the near ones +-d (strip edges ignored here: a bandwidth probe)
__global__ __launch_bounds__(512) void rows_perm_kernel(Args A) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  long long pos = (long long)wg * A.wpw + wave;
  const long long step = A.persist > 0 ? A.persist : A.n;
  const long long swf = A.gmode == 2 ? A.sw : 0, nf = A.gmode == 2 ? 1 : 0;
  for (; pos < A.n; pos += step) {
    const d2* __restrict__ rv = A.vals + pos * NE;
    const d2* __restrict__ Xs = A.x + lane;
    d2 x[NE];
    const d2 xi = Xs[pos * NS];
    if (A.gmode == 0) {
#pragma unroll
      for (int k = 0; k < NE; ++k) x[k] = xi;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[k] = Xs[clampll(pos + (long long)(k - 4) * swf, 0, A.n - 1) * NS];
        x[12 + k] = Xs[clampll(pos + (long long)(k + 1) * swf, 0, A.n - 1) * NS];
        x[4 + k] = Xs[clampll(pos + (-4 + k) * nf, 0, A.n - 1) * NS];
        x[8 + k] = Xs[clampll(pos + (1 + k) * nf, 0, A.n - 1) * NS];
      }
    }
    d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NE; k += 2) {
      cfma(s0, rv[k], x[k]);
      cfma(s1, rv[k + 1], x[k + 1]);
    }
    finish(A, pos, lane, s0, s1, xi);
  }
}

// the row-local streams alone as a plain elementwise kernel (y = 2 x - p in place of p, accumulator every third launch): what the
// memory system gives this read : write mix at this footprint, whatever the kernel structure
template <bool NT, int U>
__global__ __launch_bounds__(256) void stream_kernel(Args A) {
  const long long total = A.n * NS;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e0 = (long long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += stride * U) {
    d2 xv[U], pv[U], av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = e0 + u * stride;
      if (e < total) {
        xv[u] = A.x[e];
        pv[u] = NT ? __builtin_nontemporal_load(A.p + e) : A.p[e];
        if (A.with_acc) av[u] = NT ? __builtin_nontemporal_load(A.acc + e) : A.acc[e];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = e0 + u * stride;
      if (e < total) {
        d2 t;
        t.x = 2.0 * xv[u].x + pv[u].x;
        t.y = 2.0 * xv[u].y + pv[u].y;
        if (NT) __builtin_nontemporal_store(t, A.y + e); else A.y[e] = t;
        if (A.with_acc) {
          d2 r = av[u];
          r.x = fma(0.3, t.x, r.x);
          r.y = fma(0.3, t.y, r.y);
          if (NT) __builtin_nontemporal_store(r, A.acc_out + e); else A.acc_out[e] = r;
        }
      }
    }
  }
}

// a wavefront owns the tile a0 .. a0+RA-1  x  c0 .. c0+RC-1; tiles in the order of the row walk (strip, a-tile, c-tile fastest)
template <int RA, int RC>
__global__ __launch_bounds__(256) void tile_kernel(Args A) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long task = (long long)wg * A.wpw + wave;
  const long long na = A.n / A.g, nta = na / RA, ntc = A.sw / RC;
  if (task >= nta * ntc * (A.g / A.sw)) return;
  const long long strip = task / (nta * ntc), r = task - strip * nta * ntc;
  const long long ta = r / ntc, tc = r - ta * ntc;
  const long long a0 = ta * RA, c0 = strip * A.sw + tc * RC;
  const d2* __restrict__ Xs = A.x + lane;
  d2 xc[RA][RC], xn[RA][8], xf[8][RC];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
#pragma unroll
    for (int j = 0; j < RC; ++j) xc[i][j] = Xs[((a0 + i) * A.g + c0 + j) * NS];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xn[i][j] = Xs[((a0 + i) * A.g + clampll(c0 - 4 + j, 0, A.g - 1)) * NS];
      xn[i][4 + j] = Xs[((a0 + i) * A.g + clampll(c0 + RC + j, 0, A.g - 1)) * NS];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < RC; ++j) {
      xf[k][j] = Xs[(clampll(a0 - 4 + k, 0, na - 1) * A.g + c0 + j) * NS];
      xf[4 + k][j] = Xs[(clampll(a0 + RA + k, 0, na - 1) * A.g + c0 + j) * NS];
    }
  // the row-local streams of the whole tile before its first store (the new term overwrites v_{t-1} in place: the compiler
  // cannot move these loads above a store by itself, and one at a time they are RA RC memory latencies in a chain)
  d2 pv[RA][RC], pa[RA][RC];
#pragma unroll
  for (int i = 0; i < RA; ++i)
#pragma unroll
    for (int j = 0; j < RC; ++j) {
      const long long e = ((a0 + i) * A.g + c0 + j) * NS + lane;
      pv[i][j] = __builtin_nontemporal_load(A.p + e);
      pa[i][j] = A.with_acc ? __builtin_nontemporal_load(A.acc + e) : d2{0.0, 0.0};
    }
#pragma unroll
  for (int i = 0; i < RA; ++i)
#pragma unroll
    for (int j = 0; j < RC; ++j) {
      const long long row = (a0 + i) * A.g + c0 + j;
      const d2* __restrict__ rv = A.vals + row * NE;
      d2 x[NE];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // far: a0 + i - 4 + k  (inside the tile when >= a0) ...  a0 + i + 1 + k
        const int lo = i - 4 + k, hi = i + 1 + k;
        x[k] = lo >= 0 ? xc[cl(lo, 0, RA - 1)][j] : xf[cl(4 + lo, 0, 7)][j];
        x[12 + k] = hi < RA ? xc[cl(hi, 0, RA - 1)][j] : xf[cl(4 + hi - RA, 0, 7)][j];
        const int nl = j - 4 + k, nh = j + 1 + k;
        x[4 + k] = nl >= 0 ? xc[i][cl(nl, 0, RC - 1)] : xn[i][cl(4 + nl, 0, 7)];
        x[8 + k] = nh < RC ? xc[i][cl(nh, 0, RC - 1)] : xn[i][cl(4 + nh - RC, 0, 7)];
      }
      d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
      for (int k = 0; k < NE; k += 2) {
        cfma(s0, rv[k], x[k]);
        cfma(s1, rv[k + 1], x[k + 1]);
      }
      finish2(A, row, lane, s0, s1, xc[i][j], pv[i][j], pa[i][j]);
    }
}

// a WORKGROUP owns the tile: its rows, the far halo and the near halo are staged ONCE in LDS (RA RC + 8 RC + 8 RA rows of 1 KiB),
// then one wavefront per row reads its seventeen operands from LDS (lane = state: conflict-free 16-byte reads)
template <int RA, int RC, int V = 0, int PD = 0>
__global__ __launch_bounds__(64 * RA * RC) __attribute__((amdgpu_waves_per_eu((64 * RA * RC == 1024 ? 8 : 1), 8))) void lds_tile_kernel(Args A) {
  extern __shared__ d2 lds[];
  constexpr int NW = RA * RC, T = (RA + 8) * RC + RA * 8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long na = A.n / A.g, nta = na / RA, ntc = A.sw / RC;
  const long long strip = wg / (nta * ntc), r = wg - strip * nta * ntc;
  const long long ta = (V & 2) ? r % nta : r / ntc, tc = (V & 2) ? r / nta : r - ta * ntc;
  const long long a0 = ta * RA, c0 = strip * A.sw + tc * RC;
  const int i = wave / RC, j = wave % RC;
  const long long row = (a0 + i) * A.g + c0 + j;
  const long long e = row * NS + lane;
  d2 v0, av = {0.0, 0.0};
  if (!(V & 1)) {
    v0 = __builtin_nontemporal_load(A.p + e);
    if (A.with_acc) av = __builtin_nontemporal_load(A.acc + e);
  }
  constexpr int PER = (T + NW - 1) / NW;
  d2 st[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int slot = wave + q * NW;      // wave-uniform
    if (slot < T) {
      long long gr;
      if (slot < (RA + 8) * RC) {
        const int ai = slot / RC, jj = slot % RC;
        gr = clampll(a0 - 4 + ai, 0, na - 1) * A.g + c0 + jj;
      } else {
        const int s2 = slot - (RA + 8) * RC, ii = s2 / 8, jj = s2 % 8;
        gr = (a0 + ii) * A.g + clampll(jj < 4 ? c0 - 4 + jj : c0 + RC + jj - 4, 0, A.g - 1);
      }
      if (!((V & 4) && slot < (RA + 8) * RC && slot / RC >= RA + 4)) st[q] = A.x[gr * NS + lane];
    }
  }
  // V & 4: the far halo BELOW the tile (rows a0 + RA .. a0 + RA + 3: nobody has touched them yet -- the tiles of the next strip step
  // come later --, so they are the staged rows that come from HBM) is not staged: the wavefronts that need those rows gather them
  // straight into the operand registers (i + 1 of them for strip step i of the tile), after the staging loads, so that the barrier
  // waits for rows other tiles have fetched already and the HBM latency passes behind it
  d2 dg[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dg[k] = d2{0.0, 0.0};
    if ((V & 4) && i + 1 + k >= RA) dg[k] = A.x[(clampll(a0 + i + 1 + k, 0, na - 1) * A.g + c0 + j) * NS + lane];
  }
  if (V & 1) {
    v0 = __builtin_nontemporal_load(A.p + e);
    if (A.with_acc) av = __builtin_nontemporal_load(A.acc + e);
  }
  d2 pf = {0.0, 0.0};
  if (PD > 0) {
    // software prefetch into L2: the row of x this wavefront's position will have PD tiles further down the strip -- the YOUNGEST load
    // of the wavefront (loads return in order: an older miss would hold up the staging loads in front of the barrier), consumed by
    // a comparison that never holds after the last store
    const long long pr = clampll(a0 + 4 + 4 * PD + i, 0, na - 1) * A.g + c0 + j;
    pf = A.x[pr * NS + lane];
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int slot = wave + q * NW;
    if (slot < T) lds[slot * NS + lane] = st[q];
  }
  __syncthreads();
  const d2* __restrict__ rv = A.vals + row * NE;
  d2 x[NE];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = lds[((i + k) * RC + j) * NS + lane];
    x[12 + k] = ((V & 4) && i + 1 + k >= RA) ? dg[k] : lds[((i + 5 + k) * RC + j) * NS + lane];
    const int nl = j - 4 + k, nh = j + 1 + k;      // wave-uniform
    x[4 + k] = nl >= 0 ? lds[((i + 4) * RC + nl) * NS + lane] : lds[((RA + 8) * RC + i * 8 + 4 + nl) * NS + lane];
    x[8 + k] = nh < RC ? lds[((i + 4) * RC + nh) * NS + lane] : lds[((RA + 8) * RC + i * 8 + 4 + nh - RC) * NS + lane];
  }
  const d2 xi = lds[((i + 4) * RC + j) * NS + lane];
  d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NE; k += 2) {
    cfma(s0, rv[k], x[k]);
    cfma(s1, rv[k + 1], x[k + 1]);
  }
  finish2(A, row, lane, s0, s1, xi, v0, av);
  if (PD > 0 && pf.x == 1.2345678e-300 && pf.y == -9.87654321e-301) A.acc_out[row * NS + lane] = pf;
}

// the workgroup tile, several tiles per workgroup in turn (a persistent workgroup: no relaunch, whose granularity -- sixteen wavefront
// slots and 80 KiB at once -- leaves slots idle until a workgroup's slowest wavefront has retired).  LATE = 0: the loads of tile t + 1
// are issued before tile t is computed (they overlap its LDS phase but must be held in registers through it: 120 VGPRs, one workgroup
// per compute unit); LATE = 1: they are issued after the sums of tile t, in front of its epilogue and stores (<= 64 VGPRs)
template <int RA, int RC, int LATE>
__global__ __launch_bounds__(64 * RA * RC) __attribute__((amdgpu_waves_per_eu(LATE ? 8 : 4, 8))) void lds_tile_loop_kernel(Args A) {
  extern __shared__ d2 lds[];
  constexpr int NW = RA * RC, T = (RA + 8) * RC + RA * 8, PER = (T + NW - 1) / NW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long na = A.n / A.g, nta = na / RA, ntc = A.sw / RC;
  const long long ntiles = nta * ntc * (A.g / A.sw);
  const int tpw = A.persist;
  const int i = wave / RC, j = wave % RC;
  d2 st[PER], v0, av = {0.0, 0.0};
  long long row = 0;
  const unsigned g32 = (unsigned)A.g, na32 = (unsigned)na, ntc32 = (unsigned)ntc, per_strip = (unsigned)(nta * ntc), sw32 = (unsigned)A.sw;
  auto issue_x = [&](long long tile64) {      // (32-bit index arithmetic: a 64-bit division in the loop costs dozens of registers while the loop's state is live)
    const unsigned tile = (unsigned)tile64;
    const unsigned strip = tile / per_strip, r = tile - strip * per_strip;
    const unsigned ta = r / ntc32, tc = r - ta * ntc32;
    const int a0 = (int)(ta * RA), c0 = (int)(strip * sw32 + tc * RC);
    row = (long long)((unsigned)(a0 + i) * g32 + (unsigned)(c0 + j));
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int slot = wave + q * NW < T ? wave + q * NW : T - 1;
      unsigned gr;
      if (slot < (RA + 8) * RC) {
        const int ai = slot / RC, jj = slot % RC;
        const int aa = a0 - 4 + ai;
        gr = (unsigned)(aa < 0 ? 0 : (aa > (int)na32 - 1 ? (int)na32 - 1 : aa)) * g32 + (unsigned)(c0 + jj);
      } else {
        const int s2 = slot - (RA + 8) * RC, ii = s2 / 8, jj = s2 % 8;
        const int cc = jj < 4 ? c0 - 4 + jj : c0 + RC + jj - 4;
        gr = (unsigned)(a0 + ii) * g32 + (unsigned)(cc < 0 ? 0 : (cc > (int)g32 - 1 ? (int)g32 - 1 : cc));
      }
      st[q] = A.x[(size_t)gr * NS + lane];
    }
  };
  auto issue_streams = [&]() {
    v0 = __builtin_nontemporal_load(A.p + row * NS + lane);
    if (A.with_acc) av = __builtin_nontemporal_load(A.acc + row * NS + lane);
  };
  long long tile = (long long)wg * tpw;
  const long long tend = tile + tpw < ntiles ? tile + tpw : ntiles;
  if (tile >= tend) return;
  issue_x(tile);
  issue_streams();
  for (; tile < tend; ++tile) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int slot = wave + q * NW;
      if (slot < T) lds[slot * NS + lane] = st[q];
    }
    const long long crow = row;
    __syncthreads();
    if (!LATE && tile + 1 < tend) issue_x(tile + 1);
    const d2* __restrict__ rv = A.vals + crow * NE;
    const d2 xi = lds[((i + 4) * RC + j) * NS + lane];
    d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 4; ++h) {      // entries 4 h .. 4 h + 3: far below, near below, near above, far above
      d2 x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (h == 0) x[k] = lds[((i + k) * RC + j) * NS + lane];
        if (h == 3) x[k] = lds[((i + 5 + k) * RC + j) * NS + lane];
        const int nl = j - 4 + k, nh = j + 1 + k;
        if (h == 1) x[k] = nl >= 0 ? lds[((i + 4) * RC + nl) * NS + lane] : lds[((RA + 8) * RC + i * 8 + 4 + nl) * NS + lane];
        if (h == 2) x[k] = nh < RC ? lds[((i + 4) * RC + nh) * NS + lane] : lds[((RA + 8) * RC + i * 8 + 4 + nh - RC) * NS + lane];
      }
      cfma(s0, rv[4 * h], x[0]);
      cfma(s1, rv[4 * h + 1], x[1]);
      cfma(s0, rv[4 * h + 2], x[2]);
      cfma(s1, rv[4 * h + 3], x[3]);
    }
    __syncthreads();
    if (LATE && tile + 1 < tend) issue_x(tile + 1);      // (the row-local streams of the next tile after this tile's epilogue: their registers are this tile's until then)
    finish2(A, crow, lane, s0, s1, xi, v0, av);
    if (tile + 1 < tend) issue_streams();
  }
}

// the same with the panel's 64 states cut into slices of S: a workgroup owns tile x slice, a wavefront 64 / S rows of the tile; the
// LDS image shrinks by 64 / S, so more workgroups fit a compute unit (fewer launch bubbles) or the tile grows (fewer loads per row)
template <int RA, int RC, int S>
__global__ __launch_bounds__(RA * RC * S) void lds_tile_s_kernel(Args A) {
  extern __shared__ d2 lds[];
  constexpr int NT = RA * RC * S, T = (RA + 8) * RC + RA * 8, NSL = NS / S;
  const int tid = threadIdx.x;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = wg / NSL, slice = wg % NSL;
  const long long na = A.n / A.g, nta = na / RA, ntc = A.sw / RC;
  const long long strip = tile / (nta * ntc), r = tile - strip * nta * ntc;
  const long long ta = r / ntc, tc = r - ta * ntc;
  const long long a0 = ta * RA, c0 = strip * A.sw + tc * RC;
  const int q = tid / S, stl = tid % S, st = slice * S + stl;
  const int i = q / RC, j = q % RC;
  const long long row = (a0 + i) * A.g + c0 + j;
  const long long e = row * NS + st;
  const d2 v0 = __builtin_nontemporal_load(A.p + e);
  d2 av = {0.0, 0.0};
  if (A.with_acc) av = __builtin_nontemporal_load(A.acc + e);
  constexpr int PER = (T * S + NT - 1) / NT;
  d2 stg[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int el = tid + u * NT;
    const int slot = el / S, sx = el % S;
    if (slot < T) {
      long long gr;
      if (slot < (RA + 8) * RC) {
        const int ai = slot / RC, jj = slot % RC;
        gr = clampll(a0 - 4 + ai, 0, na - 1) * A.g + c0 + jj;
      } else {
        const int s2 = slot - (RA + 8) * RC, ii = s2 / 8, jj = s2 % 8;
        gr = (a0 + ii) * A.g + clampll(jj < 4 ? c0 - 4 + jj : c0 + RC + jj - 4, 0, A.g - 1);
      }
      stg[u] = A.x[gr * NS + slice * S + sx];
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int el = tid + u * NT;
    if (el < T * S) lds[el] = stg[u];
  }
  __syncthreads();
  const d2* __restrict__ rv = A.vals + row * NE;
  d2 v[NE];
  if (S == 32) {
    // two rows per wavefront: each half's lanes 0 .. 15 hold their row's entries (one coalesced load of 2 x 256 B, issued before the
    // barrier would be better still), broadcast by v_readlane from both halves and selected per half
    const d2 mv = rv[stl & 15];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const d2 lo = d2{readlane_f64(mv.x, k), readlane_f64(mv.y, k)}, hi = d2{readlane_f64(mv.x, 32 + k), readlane_f64(mv.y, 32 + k)};
      v[k] = (tid & 32) ? hi : lo;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NE; ++k) v[k] = rv[k];
  }
  d2 x[NE];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = lds[((i + k) * RC + j) * S + stl];
    x[12 + k] = lds[((i + 5 + k) * RC + j) * S + stl];
    const int nl = j - 4 + k, nh = j + 1 + k;
    x[4 + k] = lds[(nl >= 0 ? (i + 4) * RC + nl : (RA + 8) * RC + i * 8 + 4 + nl) * S + stl];
    x[8 + k] = lds[(nh < RC ? (i + 4) * RC + nh : (RA + 8) * RC + i * 8 + 4 + nh - RC) * S + stl];
  }
  const d2 xi = lds[((i + 4) * RC + j) * S + stl];
  d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NE; k += 2) {
    cfma(s0, v[k], x[k]);
    cfma(s1, v[k + 1], x[k + 1]);
  }
  finish2(A, row, st, s0, s1, xi, v0, av);
}

static double rnd(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return ((double)(s >> 11) / 9007199254740992.0) - 0.5;
}

template <class F>
static int time_it(const char* name, int ra, int rc, Args A, d2* bufs[4], F launch, int reps, double bytes, std::vector<d2>* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.0f;
  for (int it = 0; it < reps + 3; ++it) {
    // rotation of the three-term recurrence: the new term overwrites v_{t-1} in place (row-local); accumulator every third launch
    A.x = bufs[it % 2];
    A.p = bufs[(it + 1) % 2];
    A.y = bufs[(it + 1) % 2];
    A.acc = A.acc_out = bufs[3];
    A.with_acc = (it % 3 == 2);
    CK(hipEventRecord(e0, 0));
    launch(A);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) {
      best = std::min(best, ms);
      sum += ms;
    }
  }
  CK(hipGetLastError());
  const double avg = sum / reps;
  printf("{\"variant\": \"%s\", \"ra\": %d, \"rc\": %d, \"us_avg\": %.1f, \"us_best\": %.1f, \"bytes\": %.0f, \"tbs_avg\": %.2f, \"sw\": %d, \"wpw\": %d}\n", name, ra, rc,
         avg * 1e3, best * 1e3, bytes, bytes / (avg * 1e-3) / 1e12, A.sw, A.wpw);
  if (out) {
    out->resize((size_t)A.n * NS);
    CK(hipMemcpy(out->data(), bufs[3], out->size() * sizeof(d2), hipMemcpyDeviceToHost));
  }
  return 0;
}

int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 18;
  const int sw = argc > 2 ? atoi(argv[2]) : 64;
  const int wpw = argc > 3 ? atoi(argv[3]) : 4;
  const long long n = 1ll << log2n;
  const int g = 1024;
  const size_t pe = (size_t)n * NS;
  d2 *vals, *bufs[4];
  CK(hipMalloc(&vals, (size_t)n * NE * sizeof(d2)));
  std::vector<d2> h((size_t)n * NE);
  unsigned long long s = 7;
  for (auto& v : h) v = d2{0.05 * rnd(s), 0.05 * rnd(s)};
  CK(hipMemcpy(vals, h.data(), h.size() * sizeof(d2), hipMemcpyHostToDevice));
  std::vector<d2> hx(pe);
  for (auto& v : hx) v = d2{rnd(s), rnd(s)};
  auto reset = [&]() -> int {
    for (int i = 0; i < 4; ++i) CK(hipMemcpy(bufs[i], hx.data(), pe * sizeof(d2), hipMemcpyHostToDevice));
    return 0;
  };
  for (int i = 0; i < 4; ++i) CK(hipMalloc(&bufs[i], pe * sizeof(d2)));
  Args A{};
  A.vals = vals;
  A.n = n;
  A.g = g;
  A.sw = sw;
  A.wpw = wpw;
  // bytes of the layout per term: matrix 16 x 16 B per row + x, p, y + (acc in and out) / 3
  const double bytes = (double)n * (NE * 16.0 + NS * 16.0 * (3.0 + 2.0 / 3.0));
  const int reps = 30;
  std::vector<d2> ref, got;
  if (reset()) return 1;
  {
    Args B = A;
    B.wpw = 8;
    if (time_it("rows", 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL((rows_kernel<0, 2>), dim3((unsigned)((n + 7) / 8)), dim3(512), 0, 0, a); }, reps, bytes, &ref)) return 1;
#define ROWS(VM, GM, NAME)                                                                                                              \
    if (reset()) return 1;                                                                                                              \
    if (time_it(NAME, 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL((rows_kernel<VM, GM>), dim3((unsigned)((n + 7) / 8)), dim3(512), 0, 0, a); }, reps, bytes, &got)) return 1; \
    if (GM == 2) printf("{\"variant\": \"%s\", \"bit_identical_to_rows\": %s}\n", NAME, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false");
#define PATCH(WA, WC)                                                                                                             \
    if (reset()) return 1;                                                                                                        \
    if (time_it("rows_patch", WA, WC, B, bufs, [&](const Args& a) {                                                               \
          hipLaunchKernelGGL((rows_patch_kernel<WA, WC>), dim3((unsigned)(n / (WA * WC))), dim3(64 * WA * WC), 0, 0, a);           \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"rows_patch\": \"%dx%d\", \"bit_identical_to_rows\": %s}\n", WA, WC, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false");
    PATCH(1, 8)
    PATCH(2, 4)
    PATCH(4, 2)
    PATCH(8, 1)
    PATCH(4, 4)
    PATCH(2, 8)
    PATCH(8, 2)
    PATCH(1, 16)
    PATCH(16, 1)
    PATCH(2, 2)
    PATCH(1, 4)
    PATCH(4, 1)
    if (argc > 9) return 0;
    ROWS(1, 2, "rows_values_by_readlane")
    ROWS(2, 2, "rows_values_by_uniform_vector_loads")
    ROWS(0, 0, "rows_no_gathers")
    ROWS(1, 0, "rows_no_gathers_values_by_readlane")
    ROWS(2, 0, "rows_no_gathers_values_by_uniform_vector_loads")
    if (argc > 6 && argv[6][0] == 'z' && argc < 8) return 0;
  }
  {
    if (reset()) return 1;
    Args B = A;
    B.wpw = 8;
    B.gmode = 2;
    if (time_it("rows_walk_order_layout", 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL(rows_perm_kernel, dim3((unsigned)((n + 7) / 8)), dim3(512), 0, 0, a); }, reps, bytes, nullptr)) return 1;
    B.gmode = 1;
    if (time_it("rows_walk_order_layout_gathers_from_own_line", 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL(rows_perm_kernel, dim3((unsigned)((n + 7) / 8)), dim3(512), 0, 0, a); }, reps, bytes, nullptr)) return 1;
    B.gmode = 0;
    if (time_it("rows_walk_order_layout_no_gathers", 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL(rows_perm_kernel, dim3((unsigned)((n + 7) / 8)), dim3(512), 0, 0, a); }, reps, bytes, nullptr)) return 1;
    for (int wgs : {2048, 8192, 65536}) {
      char nm[96];
      snprintf(nm, sizeof nm, "plain_stream_nt_%d_workgroups", wgs);
      if (time_it(nm, 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL((stream_kernel<true, 4>), dim3((unsigned)wgs), dim3(256), 0, 0, a); }, reps, bytes - (double)n * NE * 16.0, nullptr)) return 1;
      snprintf(nm, sizeof nm, "plain_stream_%d_workgroups", wgs);
      if (time_it(nm, 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL((stream_kernel<false, 4>), dim3((unsigned)wgs), dim3(256), 0, 0, a); }, reps, bytes - (double)n * NE * 16.0, nullptr)) return 1;
    }
    if (argc > 5 && argv[5][0] == 'y') return 0;
    for (int wgs : {256, 512, 1024, 2048}) {
      for (int gm : {2, 0}) {
        B.gmode = gm;
        B.persist = wgs * 8;
        char nm[96];
        snprintf(nm, sizeof nm, "rows_walk_order_layout_persistent_%d_workgroups_gmode_%d", wgs, gm);
        if (time_it(nm, 1, 1, B, bufs, [&](const Args& a) { hipLaunchKernelGGL(rows_perm_kernel, dim3((unsigned)wgs), dim3(512), 0, 0, a); }, reps, bytes, nullptr)) return 1;
      }
    }
  }
  if (argc > 4 && argv[4][0] == 'x') return 0;
#define LDST(RA, RC)                                                                                                              \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / RA) * (long long)(g / RC);                                                                    \
    const size_t ldsb = (size_t)((RA + 8) * RC + RA * 8) * NS * sizeof(d2);                                                        \
    CK(hipFuncSetAttribute((const void*)lds_tile_kernel<RA, RC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));          \
    if (time_it("lds_tile", RA, RC, A, bufs, [&](const Args& a) {                                                                 \
          hipLaunchKernelGGL((lds_tile_kernel<RA, RC>), dim3((unsigned)ntask), dim3(64 * RA * RC), ldsb, 0, a);                    \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"lds_tile\": \"%dx%d\", \"lds_kib\": %zu, \"bit_identical_to_rows\": %s}\n", RA, RC, ldsb / 1024, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
  LDST(4, 4)
#define LDSV(V)                                                                                                                   \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / 4) * (long long)(g / 4);                                                                      \
    const size_t ldsb = (size_t)80 * NS * sizeof(d2);                                                                              \
    CK(hipFuncSetAttribute((const void*)lds_tile_kernel<4, 4, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));         \
    if (time_it("lds_tile_variant_" #V, 4, 4, A, bufs, [&](const Args& a) {                                                       \
          hipLaunchKernelGGL((lds_tile_kernel<4, 4, V>), dim3((unsigned)ntask), dim3(1024), ldsb, 0, a);                           \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"lds_tile_variant\": %d, \"bit_identical_to_rows\": %s}\n", V, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
  LDSV(1)
  LDSV(5)
  LDSV(7)
#define LDSP(PD)                                                                                                                  \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / 4) * (long long)(g / 4);                                                                      \
    const size_t ldsb = (size_t)80 * NS * sizeof(d2);                                                                              \
    CK(hipFuncSetAttribute((const void*)lds_tile_kernel<4, 4, 1, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));     \
    if (time_it("lds_tile_prefetch_" #PD, 4, 4, A, bufs, [&](const Args& a) {                                                     \
          hipLaunchKernelGGL((lds_tile_kernel<4, 4, 1, PD>), dim3((unsigned)ntask), dim3(1024), ldsb, 0, a);                       \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"lds_tile_prefetch\": %d, \"bit_identical_to_rows\": %s}\n", PD, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
  LDSP(2)
  if (argc > 8) return 0;
#define LDSS(RA, RC, S)                                                                                                           \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / RA) * (long long)(g / RC) * (NS / S);                                                         \
    const size_t ldsb = (size_t)((RA + 8) * RC + RA * 8) * S * sizeof(d2);                                                         \
    CK(hipFuncSetAttribute((const void*)lds_tile_s_kernel<RA, RC, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));     \
    if (time_it("lds_tile_slice_" #S, RA, RC, A, bufs, [&](const Args& a) {                                                       \
          hipLaunchKernelGGL((lds_tile_s_kernel<RA, RC, S>), dim3((unsigned)ntask), dim3(RA * RC * S), ldsb, 0, a);                \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"lds_tile_slice\": \"%dx%dx%d\", \"lds_kib\": %zu, \"bit_identical_to_rows\": %s}\n", RA, RC, S, ldsb / 1024, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
#define LDSL(RA, RC, TPW, LATE)                                                                                                         \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / RA) * (long long)(g / RC);                                                                    \
    const size_t ldsb = (size_t)((RA + 8) * RC + RA * 8) * NS * sizeof(d2);                                                        \
    CK(hipFuncSetAttribute((const void*)lds_tile_loop_kernel<RA, RC, LATE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
    Args B2 = A;                                                                                                                  \
    B2.persist = TPW;                                                                                                             \
    if (time_it("lds_tile_loop_" #TPW "_late_" #LATE, RA, RC, B2, bufs, [&](const Args& a) {                                      \
          hipLaunchKernelGGL((lds_tile_loop_kernel<RA, RC, LATE>), dim3((unsigned)((ntask + TPW - 1) / TPW)), dim3(64 * RA * RC), ldsb, 0, a); \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"lds_tile_loop\": \"%dx%d\", \"tiles_per_workgroup\": %d, \"bit_identical_to_rows\": %s}\n", RA, RC, TPW, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
  LDSL(4, 4, 1, 1)
  LDSL(4, 4, 2, 1)
  LDSL(4, 4, 4, 1)
  LDSL(4, 4, 8, 1)
  LDSL(4, 4, 16, 1)
  LDSL(4, 4, 31, 1)
  LDSL(4, 4, 62, 1)
  LDSL(4, 4, 8, 0)
  if (argc > 9) return 0;
  LDSS(4, 4, 64)
  LDSS(4, 4, 32)
  LDSS(4, 4, 16)
  LDSS(4, 8, 32)
  LDSS(8, 4, 32)
  LDSS(4, 8, 16)
  LDSS(8, 4, 16)
  LDSS(8, 8, 16)
  LDSS(2, 4, 32)
  LDSS(4, 2, 32)
  if (argc > 7) return 0;
#define TILE(RA, RC)                                                                                                              \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const long long ntask = (n / g / RA) * (long long)(g / RC);                                                                    \
    if (time_it("tile", RA, RC, A, bufs, [&](const Args& a) {                                                                     \
          hipLaunchKernelGGL((tile_kernel<RA, RC>), dim3((unsigned)((ntask + wpw - 1) / wpw)), dim3(64 * wpw), 0, 0, a);          \
        }, reps, bytes, &got)) return 1;                                                                                          \
    printf("{\"tile\": \"%dx%d\", \"bit_identical_to_rows\": %s}\n", RA, RC, memcmp(ref.data(), got.data(), pe * sizeof(d2)) == 0 ? "true" : "false"); \
  }
  TILE(1, 4)
  TILE(1, 8)
  TILE(2, 2)
  TILE(2, 4)
  TILE(4, 2)
  TILE(2, 8)
  TILE(4, 4)
  TILE(8, 2)
  TILE(4, 8)
  return 0;
}
